#!/bin/bash
OUT=gpurun_out/r02f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv.py tests/test_pool.py tests/test_hip_parity.py tests/test_runners.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -25 $OUT/pytest.log
tools/prof_workload.sh convnet $OUT/kt_convnet 100 20 > $OUT/kt_convnet.log 2>&1; tail -32 $OUT/kt_convnet.log | cut -c1-200
python bench.py --workload convnet --gpus 1 --steps 200 --warmup 20 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_convnet.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02f/bench_convnet.json').read().strip().splitlines()[-1]); print('CONVNET', d['value'])"
