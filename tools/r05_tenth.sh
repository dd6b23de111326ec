#!/bin/bash
OUT=gpurun_out/r05_tenth
mkdir -p $OUT
python -m pytest tests -q -m gpu -x --durations=5 > $OUT/gputests_full.log 2>&1
tail -12 $OUT/gputests_full.log
AB_WORKLOAD=convnet bash tools/ab_table.sh $OUT/ab_convnet tools/_ab/wt15.so tools/_ab/wt31.so > $OUT/ab_convnet.txt 2>&1
cat $OUT/ab_convnet.txt
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep2.so
for v in wt15 wt47 wt15 wt47; do
  cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
  echo "$v: $(python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1)"
done | tee $OUT/exact_pass_wt.txt
cp /tmp/keep2.so bnn_priors_amd/_build/libsgmcmc_hip.so
