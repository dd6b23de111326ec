#!/bin/bash
# tools/ab_bits.sh a.so b.so ...: tools/lab/conv_bits.py under every build (same lines = same bits)
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep_bits.so
for so in "$@"; do
  cp $so bnn_priors_amd/_build/libsgmcmc_hip.so
  echo "== $so"; python tools/lab/conv_bits.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/keep_bits.so bnn_priors_amd/_build/libsgmcmc_hip.so
