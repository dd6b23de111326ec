#!/bin/bash
# round-2 third GPU pass: fused residual block
OUT=gpurun_out/r02c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_resblock.py tests/test_conv.py tests/test_priors.py -m gpu -x -q > $OUT/pytest_block.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_block.log; tail -25 $OUT/pytest_block.log
timeout 900 python -m pytest tests/test_runners.py tests/test_hip_parity.py tests/test_bn.py tests/test_pool.py tests/test_fused_dense.py -m gpu -x -q > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_rest.log; tail -8 $OUT/pytest_rest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 > $OUT/bench_resnet.json 2> $OUT/bench_resnet.err; tail -3 $OUT/bench_resnet.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02c/bench_resnet.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], d["ms_per_step"])
for r in d["roofline_kernels"]: print(r["kernel"], r["avg_kernel_us"], r["frac"])
PY
SGMCMC_BLOCK=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet_layered.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02c/bench_resnet_layered.json').read().strip().splitlines()[-1]); print('LAYERED', d['value'])"
tools/prof_workload.sh googleresnet $OUT/kt_resnet 60 20 > $OUT/kt_resnet.log 2>&1; tail -45 $OUT/kt_resnet.log
tools/prof_workload.sh convnet $OUT/kt_convnet 100 20 > $OUT/kt_convnet.log 2>&1; head -12 $OUT/kt_convnet/steady_state_summary.txt; tail -c 400 $OUT/kt_convnet/bench.json
