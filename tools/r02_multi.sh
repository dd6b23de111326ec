#!/bin/bash
OUT=gpurun_out/r02h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_fused_dense.py tests/test_runners.py tests/test_evaluation.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -25 $OUT/pytest.log
python bench.py --workload densenet --steps 2000 --warmup 200 --cpu-budget 0 --samples 0 --sweep-log2 0 > $OUT/bench_densenet.json 2> $OUT/bench_densenet.err; tail -3 $OUT/bench_densenet.err; python -c "
import json; d=json.loads(open('gpurun_out/r02h/bench_densenet.json').read().strip().splitlines()[-1]); print('DENSENET', d['value'], d.get('chains_per_gpu'))"
