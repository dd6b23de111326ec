#!/bin/bash
OUT=gpurun_out/r05_ninth
mkdir -p $OUT
python -m pytest tests/test_runners.py tests/test_full_size.py tests/test_bn.py tests/test_priors.py -q -m gpu -x 2>&1 | tail -15 > $OUT/tests.log
for L in 1 2 3 4; do for G in 1 auto; do
  SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1
done; done > $OUT/exact_pass_lanes_groups_sweep.txt
python bench.py --steps 50 --warmup 10 --cpu-budget 0 --other-workloads 0 --sweep-log2 0 --no-kernel-timing --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/tests.log $OUT/exact_pass_lanes_groups_sweep.txt $OUT/bench_line.json
