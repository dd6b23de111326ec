#!/bin/bash
# round-3 end-of-round verification on the GPU box: GPU tests, smoke, the bench lines of all workloads, the two-rank
# plumbing run, steady-state rocprofv3 summaries.  Everything lands under gpurun_out/r03_final/.
OUT=gpurun_out/r03_final
mkdir -p $OUT
python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_googleresnet_driver_args.json 2> $OUT/bench_googleresnet.err
python bench.py --workload densenet --other-workloads 0 --eval-rows 0 > $OUT/bench_densenet.json 2> $OUT/bench_densenet.err
python bench.py --workload convnet --stream-chains 1,2 --other-workloads 0 --eval-rows 0 > $OUT/bench_convnet.json 2> $OUT/bench_convnet.err
python bench.py --inference HMCReject --trajectory 50 --temperature 0.1 --other-workloads 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_googleresnet_hmc_L50_T0.1.json 2> $OUT/bench_hmc.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --samples 0 > $OUT/bench_2rank_gloo_plumbing.json 2> $OUT/bench_2rank.err
for WL in googleresnet convnet densenet; do
  bash tools/prof_workload.sh $WL $OUT/prof_$WL 60 20 > /dev/null 2>&1
done
tail -3 $OUT/gputests.log; cat $OUT/smoke.log | tail -2
