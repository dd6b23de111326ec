#!/bin/bash
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_resblock.py tests/test_conv.py tests/test_hip_parity.py -m gpu -x -q > $OUT/pytest_block.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_block.log; tail -25 $OUT/pytest_block.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet.json 2> $OUT/bench_resnet.err; tail -3 $OUT/bench_resnet.err
python -c "
import json; d=json.loads(open('gpurun_out/r02d/bench_resnet.json').read().strip().splitlines()[-1]); print('FUSED', d['value'])"
SGMCMC_BLOCK=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet_layered.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02d/bench_resnet_layered.json').read().strip().splitlines()[-1]); print('LAYERED', d['value'])"
tools/prof_workload.sh googleresnet $OUT/kt_resnet 60 20 > $OUT/kt_resnet.log 2>&1; tail -45 $OUT/kt_resnet.log
