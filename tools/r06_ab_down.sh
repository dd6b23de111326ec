#!/bin/bash
# round 6: the down-sampling backward launch -- items per weight-gradient workgroup (2 / 1) x role order, in the step
export SGMCMC_ALLOW_STALE_LIB=1
OUT=gpurun_out/r06_ab_down
mkdir -p $OUT
V=${AB_V:-"dn_base dn_it1 dn_df dn_it1df"}
L=""; for v in $V; do L="$L tools/_ab/$v.so"; done
bash tools/ab_table.sh $OUT/tab $L > $OUT/table.txt 2>&1
cut -c1-170 $OUT/table.txt | grep -i "busy\|n/step\|convdown\|reduce"
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for v in $V $V; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python bench.py --steps 200 --warmup 30 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json
d=json.loads(open('$OUT/q.json').read().strip().splitlines()[-1])
print("$v", d['value'], d.get('ms_per_step'))
PY
done | tee $OUT/steps.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
