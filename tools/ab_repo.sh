#!/bin/bash
# A/B of two whole checkouts ON ONE GPU BOX: tools/_ab/old (e.g. `git archive <commit> | tar -x -C tools/_ab/old`, library
# built there) against this tree; alternates old / new three times, 1 s of timed steps each.
#   gpurun -- 'bash tools/ab_repo.sh [bench flags]'
FLAGS="$@ --cpu-budget 0 --sweep-log2 0 --samples 0 --no-kernel-timing --min-seconds 1.0 --other-workloads 0 --stream-chains ''"
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then d=tools/_ab/old; else d=.; fi
    ( cd $d && eval python bench.py $FLAGS 2>/dev/null | python -c "import sys,json; print('$v steps/s', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" )
  done
done
