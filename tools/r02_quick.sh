#!/bin/bash
# quick confidence run: the convolution / block / pooling / runner tests, then the bench values
OUT=${1:-gpurun_out/r02q}
TESTS=${2:-"tests/test_conv.py tests/test_resblock.py tests/test_pool.py tests/test_bn.py tests/test_runners.py tests/test_augment.py"}
mkdir -p $OUT
timeout 1500 python -m pytest $TESTS -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
for w in googleresnet convnet; do
python bench.py --workload $w --gpus 1 --steps 100 --warmup 20 --cpu-budget 0 --samples 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_$w.json 2>$OUT/bench_$w.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'])"
done
