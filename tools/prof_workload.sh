#!/bin/bash
# steady-state kernel breakdown of one bench workload: tools/prof_workload.sh <workload> <out-dir> <steps> <warmup> [extra bench flags]
set -e
WL=$1; OUT=$(realpath -m $2); STEPS=${3:-60}; WARM=${4:-20}; shift 4 || true
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- \
  python $REPO/bench.py --workload $WL --steps $STEPS --warmup $WARM --min-seconds 0.01 --cpu-budget 0 --sweep-log2 0 \
  --samples 0 --no-kernel-timing --stream-chains "" "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $REPO && python tools/step_summary.py $OUT/kt_kernel_trace.csv --steps 40 > $OUT/steady_state_summary.txt
[ -n "$KEEP_TRACE" ] || rm -f $OUT/kt_kernel_trace.csv
cat $OUT/steady_state_summary.txt; cat $OUT/bench.json
