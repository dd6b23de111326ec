#!/bin/bash
# round 6: uniform backward convolution (tools/_ab/uni1.so) against the merged two-role launch (uni0.so) IN THE STEP:
# per-kernel table under rocprofv3, then unprofiled steps/s of each build (alternating, twice)
OUT=gpurun_out/r06_ab_uni
mkdir -p $OUT
bash tools/ab_table.sh $OUT/tab tools/_ab/uni0.so tools/_ab/uni1.so > $OUT/table.txt 2>&1
cat $OUT/table.txt | cut -c1-200
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for v in uni0 uni1 uni0 uni1; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python bench.py --steps 200 --warmup 30 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json
d=json.loads(open('$OUT/q.json').read().strip().splitlines()[-1])
print("$v", d['value'], d.get('ms_per_step'))
PY
done | tee $OUT/steps.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
