#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
for L in 2 3 4 5; do for G in auto 4 8; do
  echo "lanes=$L group=$G: $(SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G python -W ignore tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1)"
done; done | tee $OUT/sweep.txt
