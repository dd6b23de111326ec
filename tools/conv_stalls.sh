#!/bin/bash
# Where a convolution launch's wave-cycles go: counter-only rocprofv3 passes (--pmc with --kernel-trace, nothing else)
# over tools/conv_pmc.py -- the six convolutions of the googleresnet step (forward with statistics, both gradients with
# the sums epilogue; three trunk shapes) at batch 128 through the C ABI.  8 SQ slots per pass (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots"), so three passes; the summary joins them per kernel.
#   tools/conv_stalls.sh <out-dir> [conv_pmc.py args]   -> <out-dir>/conv_stalls.txt
set -e
OUT=$(realpath -m ${1:-gpurun_out/stalls}); shift || true
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  ( cd $REPO && timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $P --kernel-trace --output-format csv \
      -d $OUT/pass$i -o pmc -- python tools/conv_pmc.py --iters 4 "$@" > $OUT/pass$i.log 2>&1 ) || echo "pass $i exited non-zero (see $OUT/pass$i.log)"
done
cd $REPO
python tools/conv_stalls_summary.py $(for i in 1 2 3; do find $OUT/pass$i -name '*counter_collection.csv' | head -1; done) \
    $(find $OUT/pass1 -name '*kernel_trace.csv' | head -1) > $OUT/conv_stalls.txt
find $OUT -name '*.csv' -size +4M -delete
cat $OUT/conv_stalls.txt
