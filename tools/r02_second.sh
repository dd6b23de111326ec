#!/bin/bash
# round-2 second GPU pass: two-phase staging + multi-workgroup finalize
OUT=gpurun_out/r02b
mkdir -p $OUT
python -m pytest tests/test_conv.py tests/test_hip_parity.py tests/test_runners.py tests/test_hip_reference_tests.py tests/test_bn.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 > $OUT/bench_resnet.json 2> $OUT/bench_resnet.err; tail -c 600 $OUT/bench_resnet.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02b/bench_resnet.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], d["ms_per_step"])
for r in d["roofline_kernels"]: print(r["kernel"], r["avg_kernel_us"], r["frac"])
print(d["roofline_sampler"]["avg_kernel_us"])
PY
tools/prof_workload.sh googleresnet $OUT/kt_resnet 60 20 > $OUT/kt_resnet.log 2>&1; tail -45 $OUT/kt_resnet.log
tools/prof_workload.sh convnet $OUT/kt_convnet 100 20 > $OUT/kt_convnet.log 2>&1; tail -40 $OUT/kt_convnet.log
tools/prof_workload.sh densenet $OUT/kt_densenet 400 50 > $OUT/kt_densenet.log 2>&1; tail -12 $OUT/kt_densenet.log
tools/pmc_hbm.sh $OUT googleresnet_step 0.3 -- bench.py --workload googleresnet --eager --steps 6 --warmup 3 --min-seconds 0.01 --cpu-budget 0 --sweep-log2 0 --samples 0 --no-kernel-timing --cudnn-benchmark 0 > $OUT/pmc_resnet.log 2>&1; tail -30 $OUT/pmc_resnet.log
for it in 6 20 24; do python tools/flat_arena.py --log2 28 --iters $it; done > $OUT/flat_arena_repeat.jsonl 2>&1
python tools/flat_arena.py --sweep >> $OUT/flat_arena_repeat.jsonl 2>&1; cat $OUT/flat_arena_repeat.jsonl | cut -c1-220
