#!/bin/bash
# round 5, fourth GPU call (first of the re-created container): the whole GPU suite, smoke, the driver's bench command,
# the steady-state kernel summary of the googleresnet step, chains per GPU on streams K = 1..4
OUT=gpurun_out/r05_fourth
mkdir -p $OUT
python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err
wc -c $OUT/bench_line.json
KEEP_TRACE=1 bash tools/prof_workload.sh googleresnet $OUT/prof_googleresnet 60 20 --other-workloads 0 > /dev/null 2>&1
python tools/step_summary.py $OUT/prof_googleresnet/kt_kernel_trace.csv --steps 40 --json $OUT/in_step_us.json --source profiles/r05_googleresnet_steady_state_summary.txt > /dev/null
rm -f $OUT/prof_googleresnet/kt_kernel_trace.csv
Q="--steps 50 --warmup 10 --samples 0 --cpu-budget 0 --other-workloads 0 --sweep-log2 0 --no-kernel-timing"
python bench.py $Q --stream-chains 1,2,3,4 --detail $OUT/chains_default.json > $OUT/chains_default.line 2> $OUT/chains_default.err
python bench.py $Q --workload convnet --stream-chains 1,2,3,4 --detail $OUT/chains_convnet.json > $OUT/chains_convnet.line 2> $OUT/chains_convnet.err
tail -4 $OUT/gputests.log; tail -1 $OUT/smoke.log; cat $OUT/bench_line.json
head -60 $OUT/prof_googleresnet/steady_state_summary.txt
python - <<'PY'
import json
for f in ("chains_default","chains_convnet"):
    try:
        d=json.load(open(f"gpurun_out/r05_fourth/{f}.json")); print(f, d["value"], json.dumps(d.get("chains_per_gpu")))
    except Exception as e: print(f, "ERR", e)
PY
