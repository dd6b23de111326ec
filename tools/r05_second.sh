#!/bin/bash
# round 5, second GPU call: new GPU tests, conditioning probe of the exact pass, chains per GPU on streams K = 1..4
OUT=gpurun_out/r05_second
mkdir -p $OUT
python -m pytest tests/test_priors.py tests/test_bench_line.py tests/test_evaluation.py tests/test_storage.py tests/test_full_size.py tests/test_augment.py -q -m gpu -x 2>&1 | tail -8 > $OUT/tests.log
timeout 900 python tools/conditioning_probe.py --steps 0,50,200,400,800 > $OUT/conditioning.txt 2> $OUT/conditioning.err
Q="--steps 50 --warmup 10 --samples 0 --cpu-budget 0 --other-workloads 0 --sweep-log2 0 --no-kernel-timing"
python bench.py $Q --stream-chains 1,2,3,4 --detail $OUT/chains_default.json > $OUT/chains_default.line 2> $OUT/chains_default.err
GPU_MAX_HW_QUEUES=8 python bench.py $Q --stream-chains 1,2,3,4 --detail $OUT/chains_q8.json > $OUT/chains_q8.line 2> $OUT/chains_q8.err
python bench.py $Q --workload convnet --stream-chains 1,2,3,4 --detail $OUT/chains_convnet.json > $OUT/chains_convnet.line 2> $OUT/chains_convnet.err
cat $OUT/tests.log; cat $OUT/conditioning.txt; tail -3 $OUT/conditioning.err
python - <<'PY'
import json
for f in ("chains_default","chains_q8","chains_convnet"):
    try:
        d=json.load(open(f"gpurun_out/r05_second/{f}.json")); print(f, d["value"], json.dumps(d.get("chains_per_gpu")))
    except Exception as e: print(f, "ERR", e)
PY
