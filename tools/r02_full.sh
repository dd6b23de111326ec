#!/bin/bash
OUT=gpurun_out/r02g
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02g/bench_driver.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], d["ms_per_step"], d.get("samples_per_sec"), d.get("speedup_vs_cpu"))
print("roofline", {k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_kernel_us")})
PY
python bench.py --inference HMCReject --trajectory 50 --temperature 0.1 --steps 20 --warmup 5 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing > $OUT/bench_hmc_L50_T0.1.json 2> $OUT/bench_hmc.err; tail -c 900 $OUT/bench_hmc_L50_T0.1.json; tail -3 $OUT/bench_hmc.err
