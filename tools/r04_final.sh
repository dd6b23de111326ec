#!/bin/bash
# round-4 end-of-round verification on the GPU box: GPU tests, smoke, the bench lines of all workloads (the driver's
# arguments for the headline), the 8-rank gloo plumbing run, steady-state rocprofv3 summaries + in-step durations, the
# exact-pass sweep, the evaluation probe.  Everything lands under gpurun_out/r04_final/.
OUT=gpurun_out/r04_final
mkdir -p $OUT
python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_googleresnet_driver_args.json 2> $OUT/bench_googleresnet.err
python bench.py --workload densenet --other-workloads 0 --eval-rows 0 > $OUT/bench_densenet.json 2> $OUT/bench_densenet.err
python bench.py --workload convnet --stream-chains 1,2 --other-workloads 0 > $OUT/bench_convnet.json 2> $OUT/bench_convnet.err
python bench.py --inference HMCReject --trajectory 50 --temperature 0.1 --other-workloads 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_googleresnet_hmc_L50_T0.1.json 2> $OUT/bench_hmc.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 8 --backend gloo --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --samples 0 --other-workloads 0 --no-kernel-timing > $OUT/bench_8rank_gloo_plumbing.json 2> $OUT/bench_8rank.err
for WL in googleresnet convnet densenet; do
  KEEP_TRACE=$([ $WL = googleresnet ] && echo 1) bash tools/prof_workload.sh $WL $OUT/prof_$WL 60 20 --other-workloads 0 > /dev/null 2>&1
done
python tools/step_summary.py $OUT/prof_googleresnet/kt_kernel_trace.csv --steps 40 --json $OUT/in_step_us.json --source profiles/r04_googleresnet_steady_state_summary.txt > /dev/null
rm -f $OUT/prof_googleresnet/kt_kernel_trace.csv
for L in 1 2 3; do for G in 1 4 auto; do
  SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1
done; done > $OUT/exact_pass_lanes_groups_sweep.txt
SGMCMC_EXACT_PERSISTENT=0 python tools/exact_pass_probe.py --passes 4 2>/dev/null | tail -1 | sed 's/^/default convolutions in the grouped bodies: /' >> $OUT/exact_pass_lanes_groups_sweep.txt
for W in googleresnet convnet densenet; do python tools/eval_probe.py --workload $W 2>&1 | grep "rows=" ; done > $OUT/eval_probe.txt
tail -3 $OUT/gputests.log; tail -1 $OUT/smoke.log
python - <<'PY'
import json
for f in ("bench_googleresnet_driver_args","bench_densenet","bench_convnet","bench_googleresnet_hmc_L50_T0.1"):
    try:
        d=json.loads(open(f"gpurun_out/r04_final/{f}.json").read().strip().splitlines()[-1])
        s=d.get("samples_per_sec") or {}; e=d.get("samples_per_sec_with_eval") or {}
        print(f, d["value"], s.get("per_chain"), e.get("per_chain"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("frac_in_step"))
    except Exception as ex: print(f, "ERR", ex)
PY
