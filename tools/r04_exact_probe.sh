# exact pass alone: lanes x group sweep (wall); optional kernel stats of one configuration:  bash tools/r04_exact_probe.sh [trace]
cd /root/repo; mkdir -p gpurun_out/r04
for L in 1 2 3; do for G in 1 4 8; do
  SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1
done; done | tee gpurun_out/r04/exact_probe_sweep.txt
SGMCMC_WRW_GROUP_MULT=0 SGMCMC_EXACT_LANES=2 SGMCMC_EXACT_GROUP=4 python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1 | sed 's/^/wrw_mult off: /' | tee -a gpurun_out/r04/exact_probe_sweep.txt
if [ "$1" = trace ]; then
cd /tmp && export TMPDIR=/tmp
for G in 4; do
  SGMCMC_EXACT_LANES=1 SGMCMC_EXACT_GROUP=$G rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_G$G -o p -- python /root/repo/tools/exact_pass_probe.py --passes 3 > /tmp/prof_G$G.log 2>&1
  cp /tmp/prof_G$G/p_kernel_stats.csv /root/repo/gpurun_out/r04/exact_G${G}_kernel_stats.csv
done
fi
