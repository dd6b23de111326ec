#!/bin/bash
# round 6: K chains on min(K, 4) streams (multichain.chain_streams)
OUT=gpurun_out/r06_chains2
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 \
  --stream-chains 1,2,4,6,8,12 --detail $OUT/detail.json > $OUT/line.json 2> $OUT/err.txt
python - <<PY
import json
d=json.load(open('$OUT/detail.json'))['chains_per_gpu']
print("distinct", d.get('distinct_hw_queues'), {k:(v['aggregate_steps_per_s'], v['us_per_lockstep']) for k,v in d.items() if isinstance(v,dict)})
PY
