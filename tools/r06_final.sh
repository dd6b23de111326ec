#!/bin/bash
# round-6 end-of-round verification on the GPU box: GPU tests, smoke, the bench lines of all workloads (the driver's
# arguments for the headline), the 2-rank gloo plumbing run through --gpus 2, steady-state rocprofv3 summaries + in-step
# durations, counter passes (HBM bytes per launch) of the convolution and BatchNorm kernels, the exact-pass sweep.
OUT=gpurun_out/r06_final
mkdir -p $OUT
python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
for WL in googleresnet convnet densenet; do
  KEEP_TRACE=$([ $WL = googleresnet ] && echo 1) bash tools/prof_workload.sh $WL $OUT/prof_$WL 60 20 --other-workloads 0 > /dev/null 2>&1
done
python tools/step_summary.py $OUT/prof_googleresnet/kt_kernel_trace.csv --steps 40 --json $OUT/in_step_us.json --source profiles/r06_googleresnet_steady_state_summary.txt > /dev/null
rm -f $OUT/prof_googleresnet/kt_kernel_trace.csv
cp $OUT/in_step_us.json profiles/in_step_us.json
# counter passes -> pmc_traffic.json (as tools/r04_pmc.sh)
P=$OUT/pmc; mkdir -p $P
tools/pmc_hbm.sh $P conv_kernels 1.0 -- tools/conv_pmc.py --iters 4 > $P/pmc_conv.log 2>&1
for S in 16x32 32x16 64x8; do
  C=${S%x*}; HW=${S#*x}
  PMC_SUMMARY_ARGS="--bn-shape 128,$C,$HW" tools/pmc_hbm.sh $P bn_$S 1.0 -- tools/bn_pmc.py --shape $S --iters 4 > $P/pmc_bn_$S.log 2>&1
done
python tools/pmc_to_json.py $P/conv_kernels_pmc_hbm.txt $P/bn_16x32_pmc_hbm.txt $P/bn_32x16_pmc_hbm.txt $P/bn_64x8_pmc_hbm.txt > $P/pmc_traffic.json
cp $P/pmc_traffic.json profiles/pmc_traffic.json
find $P -name '*.csv' -size +2M -delete
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_googleresnet_driver_args_detail.json > $OUT/bench_googleresnet_driver_args.json 2> $OUT/bench_googleresnet.err
python bench.py --workload densenet --other-workloads 0 --eval-rows 0 --detail $OUT/bench_densenet_detail.json > $OUT/bench_densenet.json 2> $OUT/bench_densenet.err
python bench.py --workload convnet --stream-chains 1,2,4,8 --other-workloads 0 --detail $OUT/bench_convnet_detail.json > $OUT/bench_convnet.json 2> $OUT/bench_convnet.err
python bench.py --inference HMCReject --trajectory 50 --temperature 0.1 --other-workloads 0 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --detail $OUT/bench_hmc_detail.json > $OUT/bench_googleresnet_hmc_L50_T0.1.json 2> $OUT/bench_hmc.err
timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --samples 0 --other-workloads 0 --no-kernel-timing --detail $OUT/bench_2rank_detail.json > $OUT/bench_2rank_gloo_plumbing.json 2> $OUT/bench_2rank.err
tail -3 $OUT/gputests.log; tail -1 $OUT/smoke.log
head -3 $OUT/prof_googleresnet/steady_state_summary.txt
for f in bench_googleresnet_driver_args bench_densenet bench_convnet bench_googleresnet_hmc_L50_T0.1 bench_2rank_gloo_plumbing; do echo "== $f: $(wc -c < $OUT/$f.json) bytes"; cat $OUT/$f.json; done
