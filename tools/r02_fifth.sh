#!/bin/bash
OUT=gpurun_out/r02e
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -25 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet.json 2> $OUT/bench_resnet.err; tail -3 $OUT/bench_resnet.err
python -c "
import json; d=json.loads(open('gpurun_out/r02e/bench_resnet.json').read().strip().splitlines()[-1]); print('DEFAULT', d['value'])"
SGMCMC_BLOCK_FUSED=16x32,32x16,64x8 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet_allfused.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02e/bench_resnet_allfused.json').read().strip().splitlines()[-1]); print('ALLFUSED', d['value'])"
SGMCMC_BLOCK=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_resnet_layered.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02e/bench_resnet_layered.json').read().strip().splitlines()[-1]); print('LAYERED', d['value'])"
