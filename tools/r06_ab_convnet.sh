#!/bin/bash
# round 6: conv::xcd_remap in the convolutional classifier's kernels (convfirst, conv50: cn1) against the previous tree
# (down1), then the same order for the linear head's rows and the minibatch gather (cn2), in both convolutional steps
export SGMCMC_ALLOW_STALE_LIB=1
OUT=gpurun_out/r06_ab_convnet
A=${AB_A:-down1}; B=${AB_B:-cn1}
mkdir -p $OUT
AB_WORKLOAD=convnet bash tools/ab_table.sh $OUT/tab_$B tools/_ab/$A.so tools/_ab/$B.so > $OUT/table_$B.txt 2>&1
cut -c1-150 $OUT/table_$B.txt | head -30
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for w in convnet googleresnet; do
for v in $A $B $A $B; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python bench.py --workload $w --steps 300 --warmup 50 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json
d=json.loads(open('$OUT/q.json').read().strip().splitlines()[-1])
print("$w $v", d['value'], d.get('ms_per_step'))
PY
done
done | tee $OUT/steps_$B.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
