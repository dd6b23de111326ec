#!/bin/bash
# HBM-side traffic of the kernels of one command, from the PMC counters FETCH_SIZE and WRITE_SIZE collected in
# SEPARATE rocprofv3 passes (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), counter-only
# (--pmc with --kernel-trace; no sys/hip/hsa traces).  Usage on the GPU box:
#   tools/pmc_hbm.sh <out-dir> <summary-name> <tail-fraction> -- <python script + args>
# Writes <out-dir>/<summary-name>_pmc_hbm.txt (per-kernel averages over the last <tail-fraction> of the dispatches).
set -e
OUT=$(realpath -m $1); NAME=$2; TAIL=$3; shift 4
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd $REPO && timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $C --kernel-trace --output-format csv \
      -d $OUT/$NAME.$C -o pmc -- python "$@" > $OUT/$NAME.$C.log 2>&1 ) || echo "pass $C exited non-zero (see log)"
done
cd $REPO
F=$(find $OUT/$NAME.FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/$NAME.WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_hbm_summary.py "$F" "$W" --tail $TAIL $PMC_SUMMARY_ARGS > $OUT/${NAME}_pmc_hbm.txt
find $OUT/$NAME.FETCH_SIZE $OUT/$NAME.WRITE_SIZE -name '*.csv' -size +8M -delete   # keep scratch small
cat $OUT/${NAME}_pmc_hbm.txt
