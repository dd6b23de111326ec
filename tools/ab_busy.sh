#!/bin/bash
# A/B of library builds by the GPU-busy time per googleresnet step under rocprofv3 (steadier than steps/s across a noisy box):
#   tools/ab_busy.sh tools/_ab/a.so tools/_ab/b.so ...   (each variant profiled once, then the first again)
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep.so
for so in "$@" "$1"; do
  cp $so bnn_priors_amd/_build/libsgmcmc_hip.so
  bash tools/prof_workload.sh googleresnet gpurun_out/ab_busy 60 20 --other-workloads 0 > /dev/null 2>&1
  echo "$so: $(head -1 gpurun_out/ab_busy/steady_state_summary.txt)"
  grep "bn::bwd_dx_kernel<true, false\|bn::apply_kernel<true, false\|conv3x3_kernel<16" gpurun_out/ab_busy/steady_state_summary.txt | cut -c1-100
done
cp /tmp/keep.so bnn_priors_amd/_build/libsgmcmc_hip.so
