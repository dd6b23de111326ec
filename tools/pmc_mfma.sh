#!/bin/bash
# MFMA utilisation of the gradient kernels (reported separately from the HBM line).
# Counter pass only (--pmc with --kernel-trace; no sys/hip/hsa traces).  Usage on the GPU box:
#   tools/pmc_mfma.sh <workload> <out-dir> [extra bench flags]
set -e   # every pass is bounded: a counter set the hardware rejects aborts the queue and rocprofv3 then waits forever
WL=$1; OUT=$(realpath -m $2); shift 2
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc ${PMC:-MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} \
  --kernel-trace --output-format csv -d $OUT -o pmc -- \
  python $REPO/bench.py --workload $WL --steps 12 --warmup 4 --cpu-budget 0 --sweep-log2 0 --samples 0 \
  --no-kernel-timing "$@" > $OUT/bench.log 2>&1
ls -la $OUT | head
cd $REPO && python tools/pmc_summary.py $OUT/pmc_counter_collection.csv > $OUT/mfma_summary.txt
rm -f $OUT/pmc_counter_collection.csv $OUT/pmc_kernel_trace.csv    # hundreds of MB with MIOpen's find mode
cat $OUT/mfma_summary.txt
