#!/bin/bash
# A/B of library builds by the exact pass's wall time (tools/exact_pass_probe.py): tools/ab_pass.sh a.so b.so ... (first again at the end)
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep.so
for so in "$@" "$1"; do
  cp $so bnn_priors_amd/_build/libsgmcmc_hip.so
  echo "$so: $(python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1 | cut -c1-150)"
done
cp /tmp/keep.so bnn_priors_amd/_build/libsgmcmc_hip.so
