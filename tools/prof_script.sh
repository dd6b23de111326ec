#!/bin/bash
# kernel durations of an arbitrary python script: tools/prof_script.sh <out-dir> <filter-regex> <script> [args...]
set -e
OUT=$(realpath -m $1); FILT=$2; shift 2
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $REPO && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python "$@" > $OUT/run.log 2>&1 ) || true
cd $REPO
python - "$OUT/kt_kernel_stats.csv" "$FILT" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Name"])]
for r in rows:
    print(f'{float(r["AverageNs"]) / 1e3:9.2f} us avg  {int(r["Calls"]):6d} calls  min {float(r["MinNs"]) / 1e3:8.2f}  {r["Name"][:100]}')
PY
rm -f $OUT/kt_kernel_trace.csv
