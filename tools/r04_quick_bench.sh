# the three conv-net bench lines, samples/s included (no rooflines, no CPU baseline, no sub-runs)
cd /root/repo; mkdir -p gpurun_out/r04
run() { name=$1; shift
  python bench.py --steps 20 --warmup 5 --samples 10 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" "$@" > gpurun_out/r04/q_$name.json 2> gpurun_out/r04/q_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/q_$name.json').read().strip().splitlines()[-1])
s=d.get('samples_per_sec') or {}; e=d.get('samples_per_sec_with_eval') or {}
print("$name", d['value'], s.get('per_chain'), e.get('per_chain'))
PY
}
run googleresnet
run hmc_T0.1 --inference HMCReject --trajectory 50 --temperature 0.1
run convnet --workload convnet --steps 100 --warmup 20
