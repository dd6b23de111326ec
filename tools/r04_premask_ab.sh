#!/bin/bash
# A/B of bnlink.PREMASK (gradients of BatchNorm + ReLU outputs stored masked by their producers): tests, then GPU-busy
# time per googleresnet step under rocprofv3 with the switch off / on / off, then the three quick bench lines each way.
OUT=gpurun_out/r04_premask
mkdir -p $OUT
python -m pytest tests/test_resblock.py tests/test_bn.py tests/test_conv.py tests/test_graphed.py tests/test_full_size.py -q -m gpu -x 2>&1 | tail -8 > $OUT/tests.log
tail -3 $OUT/tests.log
for v in 0 1 0 1; do
  SGMCMC_PREMASK=$v bash tools/prof_workload.sh googleresnet $OUT/prof_$v 60 20 --other-workloads 0 > /dev/null 2>&1
  echo "PREMASK=$v: $(head -1 $OUT/prof_$v/steady_state_summary.txt)  $(python -c "import json;print(json.loads(open('$OUT/prof_$v/bench.json').read().strip().splitlines()[-1])['value'])")"
done
grep "bwd_dx\|conv3x3_bwd" $OUT/prof_0/steady_state_summary.txt | cut -c1-110
echo ---
grep "bwd_dx\|conv3x3_bwd" $OUT/prof_1/steady_state_summary.txt | cut -c1-110
for v in 0 1; do
  echo "PREMASK=$v"
  SGMCMC_PREMASK=$v bash tools/r04_quick_bench.sh 2>&1 | tail -3
done
