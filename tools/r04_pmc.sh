#!/bin/bash
# round 4: counter-only HBM passes (FETCH_SIZE / WRITE_SIZE separately, --kernel-trace only) of the trunk's convolution
# kernels (tools/conv_pmc.py) and of the BatchNorm kernels stage by stage (tools/bn_pmc.py) -> profiles/pmc_traffic.json
OUT=gpurun_out/r04pmc
mkdir -p $OUT
tools/pmc_hbm.sh $OUT conv_kernels 1.0 -- tools/conv_pmc.py --iters 4 > $OUT/pmc_conv.log 2>&1; tail -20 $OUT/pmc_conv.log
for S in 16x32 32x16 64x8; do
  C=${S%x*}; HW=${S#*x}
  PMC_SUMMARY_ARGS="--bn-shape 128,$C,$HW" tools/pmc_hbm.sh $OUT bn_$S 1.0 -- tools/bn_pmc.py --shape $S --iters 4 > $OUT/pmc_bn_$S.log 2>&1
  grep "bn::" $OUT/bn_${S}_pmc_hbm.txt
done
python tools/pmc_to_json.py $OUT/conv_kernels_pmc_hbm.txt $OUT/bn_16x32_pmc_hbm.txt $OUT/bn_32x16_pmc_hbm.txt $OUT/bn_64x8_pmc_hbm.txt > $OUT/pmc_traffic.json
find $OUT -name '*.csv' -size +2M -delete
