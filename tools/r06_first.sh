#!/bin/bash
# round 6, first full check on the GPU box: GPU tests, smoke, the headline bench line (driver's arguments), the 2-rank
# gloo plumbing run through --gpus 2
OUT=gpurun_out/r06_first
mkdir -p $OUT
python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -25 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_googleresnet_driver_args_detail.json > $OUT/bench_googleresnet_driver_args.json 2> $OUT/bench_googleresnet.err
timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --samples 0 --other-workloads 0 --no-kernel-timing --detail $OUT/bench_2rank_detail.json > $OUT/bench_2rank_gloo_plumbing.json 2> $OUT/bench_2rank.err
tail -5 $OUT/gputests.log; tail -1 $OUT/smoke.log
for f in bench_googleresnet_driver_args bench_2rank_gloo_plumbing; do echo "== $f: $(wc -c < $OUT/$f.json) bytes"; cat $OUT/$f.json; done
tail -3 $OUT/bench_googleresnet.err | cut -c1-300
