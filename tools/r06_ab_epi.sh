#!/bin/bash
# round 6: EPI as a template value (no run-time branch around the e_out load) against the previous build -- per-kernel
# in-step table, conv bits, steps/s alternating, the exact pass (persistent kernels) alternating
export SGMCMC_ALLOW_STALE_LIB=1
OUT=gpurun_out/r06_ab_epi
mkdir -p $OUT
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep0.so
for v in pre_prune epi_int; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python -W ignore tools/lab/conv_bits.py > $OUT/bits_$v.txt 2>&1
done
cmp -s $OUT/bits_pre_prune.txt $OUT/bits_epi_int.txt && echo "conv bits IDENTICAL" || { echo "conv bits DIFFER"; diff $OUT/bits_pre_prune.txt $OUT/bits_epi_int.txt | head -5; }
cp /tmp/keep0.so bnn_priors_amd/_build/libsgmcmc_hip.so
bash tools/ab_table.sh $OUT/tab tools/_ab/pre_prune.so tools/_ab/epi_int.so > $OUT/table.txt 2>&1
grep -v "^ *[0-9]* *[0-9.]* *[0-9.]* *[0-9.]*  \(void \)\?bn::\|augment\|head::\|convstem\|convdown::fwd" $OUT/table.txt | cut -c1-150
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for v in pre_prune epi_int pre_prune epi_int; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  python bench.py --steps 200 --warmup 30 --samples 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json
d=json.loads(open('$OUT/q.json').read().strip().splitlines()[-1])
print("$v", d['value'], d.get('ms_per_step'))
PY
  echo "$v exact pass: $(python -W ignore tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1 | cut -c1-120)"
done | tee $OUT/steps.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
