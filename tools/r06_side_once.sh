#!/bin/bash
# round 6 lab: merged backward launch / one fork per convolution / one fork per pass (timing bound only): tools/lab/side_once.py
OUT=gpurun_out/r06_side_once
mkdir -p $OUT
for m in merged fork once merged once; do
  SIDE_MODE=$m timeout 300 python tools/lab/side_once.py > $OUT/$m.json 2> $OUT/$m.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/$m.json').read().strip().splitlines()[-1]); print("$m", d['value'], d.get('ms_per_step'))
except Exception as e:
    print("$m failed", e); print(open('$OUT/$m.err').read()[-1500:])
PY
done | tee $OUT/steps.txt
