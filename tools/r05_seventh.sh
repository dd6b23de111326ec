#!/bin/bash
OUT=$(realpath -m gpurun_out/r05_seventh)
mkdir -p $OUT
REPO=$(pwd)
python -m pytest tests/test_priors.py -q -m gpu 2>&1 | tail -5 > $OUT/priors.log
Q="--steps 50 --warmup 10 --samples 0 --cpu-budget 0 --other-workloads 0 --sweep-log2 0 --no-kernel-timing"
python bench.py $Q --stream-chains 1,2,3,4 --detail $OUT/chains.json > $OUT/chains.line 2> $OUT/chains.err
for K in 2 4; do
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_k$K -o kt -- python $REPO/bench.py $Q --stream-chains $K --detail $OUT/prof_k$K.json > /dev/null 2> $OUT/prof_k$K.err
  cd $REPO
  python tools/union_busy.py $OUT/prof_k$K/kt_kernel_trace.csv --frac 0.25 > $OUT/union_busy_k$K.txt
  rm -f $OUT/prof_k$K/kt_kernel_trace.csv
done
cat $OUT/priors.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_seventh/chains.json")); print(json.dumps(d.get("chains_per_gpu"), indent=0))
PY
cat $OUT/union_busy_k2.txt $OUT/union_busy_k4.txt
