#!/bin/bash
# samples/s of the googleresnet sample cycle against the number of streams the exact full-data pass uses
for L in 1 2 3 4; do
  SGMCMC_EXACT_LANES=$L python bench.py --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing \
    --other-workloads 0 --eval-rows 0 --stream-chains "" 2>/dev/null > /tmp/lanes_$L.json
  python - $L <<'PY'
import sys, json
L = sys.argv[1]
d = json.loads(open(f"/tmp/lanes_{L}.json").read().strip().splitlines()[-1])
print("lanes", L, "steps/s", d["value"], "samples/s", d["samples_per_sec"]["value"])
PY
done
