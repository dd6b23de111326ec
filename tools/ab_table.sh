#!/bin/bash
# A/B of library builds, per kernel: tools/ab_table.sh <out-dir> a.so b.so ...   (each variant profiled once under rocprofv3
# --kernel-trace in the googleresnet step, the first again at the end; one table of mean in-step durations per kernel)
OUT=$1; shift
mkdir -p $OUT
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep.so
i=0
for so in "$@" "$1"; do
  cp $so bnn_priors_amd/_build/libsgmcmc_hip.so
  name=$(basename $so .so); [ $i -ge $# ] && name=${name}_again
  bash tools/prof_workload.sh ${AB_WORKLOAD:-googleresnet} $OUT/run 60 20 --other-workloads 0 > /dev/null 2>&1
  cp $OUT/run/steady_state_summary.txt $OUT/$name.txt
  i=$((i+1))
done
cp /tmp/keep.so bnn_priors_amd/_build/libsgmcmc_hip.so
rm -rf $OUT/run
python tools/ab_table.py $OUT/*.txt
