#!/bin/bash
# round 6, after the knob pruning: the trunk convolutions' bits before / after (tools/lab/conv_bits.py on the pre-prune
# library and on the tree's), then the GPU tests
OUT=gpurun_out/r06_second
mkdir -p $OUT
cp bnn_priors_amd/_build/libsgmcmc_hip.so /tmp/keep.so
cp tools/_ab/pre_prune.so bnn_priors_amd/_build/libsgmcmc_hip.so
SGMCMC_ALLOW_STALE_LIB=1 python -W ignore tools/lab/conv_bits.py > $OUT/bits_before.txt 2>&1
cp /tmp/keep.so bnn_priors_amd/_build/libsgmcmc_hip.so
python -W ignore tools/lab/conv_bits.py > $OUT/bits_after.txt 2>&1
if cmp -s $OUT/bits_before.txt $OUT/bits_after.txt; then echo "conv bits: IDENTICAL before and after the pruning ($(wc -l < $OUT/bits_after.txt) lines)"; else echo "conv bits DIFFER"; diff $OUT/bits_before.txt $OUT/bits_after.txt | head; fi | tee $OUT/bits_verdict.txt
python -m pytest tests -q -m gpu -x --durations=6 2>&1 | tail -22 > $OUT/gputests.log
tail -9 $OUT/gputests.log | cut -c1-300
