#!/bin/bash
# end-of-round verification: full GPU suite, smoke, the bench lines, steady-state profiles
OUT=${1:-gpurun_out/r02z}
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
bash tools/r02_bench_all.sh $OUT 2>&1 | tail -8
tools/prof_workload.sh googleresnet $OUT/kt_resnet 60 20 > /dev/null 2>&1
tools/prof_workload.sh convnet $OUT/kt_convnet 100 20 >/dev/null 2>&1
tools/prof_workload.sh densenet $OUT/kt_densenet 2000 200 --chain-sweep "" > /dev/null 2>&1
head -2 $OUT/kt_*/steady_state_summary.txt
