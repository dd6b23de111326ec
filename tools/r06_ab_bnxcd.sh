#!/bin/bash
# round 6: XCD-aware block order of the BatchNorm kernels x store policy of the activation-sized outputs, in the step
OUT=gpurun_out/r06_ab_bnxcd
mkdir -p $OUT
bash tools/ab_table.sh $OUT/tab tools/_ab/x1w15.so tools/_ab/down1.so > $OUT/table.txt 2>&1
cut -c1-170 $OUT/table.txt | head -40
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for v in x1w15 down1 x1w15 down1 x1w15 down1; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python bench.py --steps 200 --warmup 30 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json
d=json.loads(open('$OUT/q.json').read().strip().splitlines()[-1])
print("$v", d['value'], d.get('ms_per_step'))
PY
done | tee $OUT/steps.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
