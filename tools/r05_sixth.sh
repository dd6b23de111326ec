#!/bin/bash
OUT=gpurun_out/r05_sixth
mkdir -p $OUT
python tools/lab/mixture_debug.py > $OUT/mixture.txt 2>&1
Q="--steps 50 --warmup 10 --samples 0 --cpu-budget 0 --other-workloads 0 --sweep-log2 0 --no-kernel-timing"
GPU_MAX_HW_QUEUES=8 python bench.py $Q --stream-chains 1,2,3,4,6 --detail $OUT/chains_q8.json > $OUT/chains_q8.line 2> $OUT/chains_q8.err
GPU_MAX_HW_QUEUES=8 python bench.py $Q --workload convnet --stream-chains 1,2,3,4,6 --detail $OUT/chains_convnet_q8.json > $OUT/chains_convnet_q8.line 2> $OUT/chains_convnet_q8.err
tail -40 $OUT/mixture.txt
python - <<'PY'
import json
for f in ("chains_q8","chains_convnet_q8"):
    try:
        d=json.load(open(f"gpurun_out/r05_sixth/{f}.json")); print(f, d["value"], json.dumps(d.get("chains_per_gpu")))
    except Exception as e: print(f, "ERR", e)
PY
