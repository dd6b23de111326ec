cd /root/repo
for V in "" "64x8" "32x16,64x8" "16x32,32x16,64x8" ""; do
  SGMCMC_CONV_PERSISTENT_BWD=$V bash tools/prof_workload.sh googleresnet gpurun_out/ab_busy 60 20 --other-workloads 0 > /dev/null 2>&1
  echo "persistent bwd at [$V]: $(head -1 gpurun_out/ab_busy/steady_state_summary.txt)  bench $(python -c "import json;print(json.loads(open('gpurun_out/ab_busy/bench.json').read().strip().splitlines()[-1])['value'])")"
done
