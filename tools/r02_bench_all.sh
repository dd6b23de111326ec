#!/bin/bash
# the three workloads' bench lines (driver arguments for the headline), HMC configs[4], saved for profiles/
OUT=${1:-gpurun_out/r02k}
mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_googleresnet_driver_args.json 2> $OUT/bench_googleresnet.err
python bench.py --workload convnet --cpu-budget 8 > $OUT/bench_convnet.json 2> $OUT/bench_convnet.err
python bench.py --workload densenet --cpu-budget 8 > $OUT/bench_densenet.json 2> $OUT/bench_densenet.err
python bench.py --inference HMCReject --trajectory 50 --temperature 0.1 --cpu-budget 0 --sweep-log2 0 > $OUT/bench_googleresnet_hmc_L50_T0.1.json 2> $OUT/bench_hmc.err
python bench.py --augment 0 --cpu-budget 0 --sweep-log2 0 --samples 0 --no-kernel-timing > $OUT/bench_googleresnet_noaugment.json 2>/dev/null
python - <<PY
import json
for f in ("bench_googleresnet_driver_args","bench_convnet","bench_densenet","bench_googleresnet_hmc_L50_T0.1","bench_googleresnet_noaugment"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("samples_per_sec",{}).get("value"), d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), d.get("cpu_baseline",{}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -n 3 $OUT/*.err
