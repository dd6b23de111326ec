#!/bin/bash
# round 3: counter-only passes of the trunk convolution kernels launched alone (tools/conv_pmc.py) -- HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), MfmaUtil, and the kernel-trace stats of the same script
OUT=gpurun_out/r03pmc
mkdir -p $OUT
tools/pmc_hbm.sh $OUT conv_kernels 1.0 -- tools/conv_pmc.py --iters 4 > $OUT/pmc_conv.log 2>&1; tail -30 $OUT/pmc_conv.log
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/$OUT/mfma -o pmc -- python /root/repo/tools/conv_pmc.py --iters 4 > /root/repo/$OUT/mfma.log 2>&1
cd /root/repo
python tools/pmc_summary.py $(find $OUT/mfma -name '*counter_collection.csv' | head -1) --tail 1.0 > $OUT/conv_kernels_mfma_util.txt; cat $OUT/conv_kernels_mfma_util.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/kt -o kt -- python /root/repo/tools/conv_pmc.py --iters 20 > /root/repo/$OUT/kt.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r03pmc/kt/**/kt_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=open("gpurun_out/r03pmc/conv_kernels_kernel_stats.txt","w")
for r in rows:
    line=f'{float(r["AverageNs"])/1e3:9.2f} us avg {int(r["Calls"]):5d} calls  min {float(r["MinNs"])/1e3:8.2f}  {r["Name"][:110]}'
    print(line); out.write(line+"\n")
PY
find $OUT -name '*.csv' -size +4M -delete
