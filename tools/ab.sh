#!/bin/bash
# A/B of two builds of libsgmcmc_hip.so ON ONE GPU BOX (box-to-box spread is ~1 %, larger than most single changes):
#   tools/ab.sh prepare old|new     (here, after building the variant: stashes a copy under tools/_ab/)
#   gpurun -- 'bash tools/ab.sh run [micro-script.py]'   (alternates old/new twice: micro timings + a 1 s bench)
# tools/_ab/ is scratch: remove it before committing.
set -e
case "$1" in
  prepare) mkdir -p tools/_ab; cp bnn_priors_amd/_build/libsgmcmc_hip.so tools/_ab/$2.so;;
  run)
    for v in old new old new; do
      cp tools/_ab/$v.so bnn_priors_amd/_build/libsgmcmc_hip.so
      echo "== $v"
      [ -n "$2" ] && python $2 2>&1 | grep -v amdgpu.ids
      python bench.py ${AB_BENCH_FLAGS} --cpu-budget 0 --sweep-log2 0 --samples 0 --no-kernel-timing --min-seconds 1.0 2>/dev/null \
        | python -c "import sys,json; print('steps/s', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
    done;;
esac
