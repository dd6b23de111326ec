#!/bin/bash
# round-2 first GPU pass: tests, bench at the driver's arguments and at the defaults, PMC traffic, kernel trace
OUT=gpurun_out/r02a
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; tail -c 3000 $OUT/bench_driver_args.json
python bench.py --cpu-budget 0 --samples 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
tools/pmc_hbm.sh $OUT flat_arena_2p28 0.6 -- tools/flat_arena.py --log2 28 --iters 6 > $OUT/pmc_flat.log 2>&1; tail -8 $OUT/pmc_flat.log
tools/pmc_hbm.sh $OUT googleresnet_step 0.4 -- bench.py --workload googleresnet --steps 10 --warmup 4 --min-seconds 0.01 --cpu-budget 0 --sweep-log2 0 --samples 0 --no-kernel-timing > $OUT/pmc_resnet.log 2>&1; tail -30 $OUT/pmc_resnet.log
tools/prof_workload.sh googleresnet $OUT/kt_resnet 60 20 60 > $OUT/kt_resnet.log 2>&1; tail -30 $OUT/kt_resnet.log
