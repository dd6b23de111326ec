#!/bin/bash
# round 6: where does the several-chains-per-GPU aggregate fall? K = 3 .. 8 streams, and the same with 16 hardware queues
OUT=gpurun_out/r06_chains
mkdir -p $OUT
for Q in 8 16; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --steps 20 --warmup 5 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 \
    --stream-chains 3,4,5,6,8 --detail $OUT/detail_q$Q.json > $OUT/line_q$Q.json 2> $OUT/err_q$Q.txt
  python - <<PY
import json
d=json.load(open('$OUT/detail_q$Q.json'))['chains_per_gpu']
print("GPU_MAX_HW_QUEUES=$Q distinct", d.get('distinct_hw_queues'), {k:(v['aggregate_steps_per_s'], v['us_per_lockstep']) for k,v in d.items() if isinstance(v,dict)})
PY
done | tee $OUT/summary.txt
