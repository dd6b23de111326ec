#!/bin/bash
# A/B of an environment switch, per kernel: tools/ab_env.sh <out-dir> VAR a b [a b ...]  (googleresnet step under rocprofv3)
OUT=$1; VAR=$2; shift 2
mkdir -p $OUT
i=0
for v in "$@"; do
  env $VAR=$v bash tools/prof_workload.sh ${AB_WORKLOAD:-googleresnet} $OUT/run 60 20 --other-workloads 0 > /dev/null 2>&1
  cp $OUT/run/steady_state_summary.txt $OUT/${VAR}_${v}_$i.txt
  i=$((i+1))
done
rm -rf $OUT/run
python tools/ab_table.py $OUT/*.txt
