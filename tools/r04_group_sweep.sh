# bench samples/s for a few (lanes, group) settings of the exact pass: googleresnet VerletSGLDReject and HMC L = 50
cd /root/repo; mkdir -p gpurun_out/r04
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --samples 10 --cpu-budget 0 --sweep-log2 0 --no-kernel-timing --other-workloads 0 --stream-chains "" $EXTRA > gpurun_out/r04/$name.json 2> gpurun_out/r04/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/$name.json').read().strip().splitlines()[-1])
s=d.get('samples_per_sec') or {}; e=d.get('samples_per_sec_with_eval') or {}
print("$name", d['value'], s.get('per_chain'), e.get('per_chain'))
PY
}
EXTRA=""
for L in 2 3; do for G in 4 8; do run g_L${L}_G$G SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G; done; done
EXTRA="--inference HMCReject --trajectory 50 --temperature 0.1 --weight-prior student-t"
for L in 2 3; do for G in 4 8; do run hmc_L${L}_G$G SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G; done; done
EXTRA="--workload convnet"
for L in 2 3; do for G in 1 4 8; do run convnet_L${L}_G$G SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=$G; done; done
