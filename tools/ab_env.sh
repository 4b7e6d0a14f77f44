#!/bin/bash
# A/B of an environment switch of the engine (e.g. BEOM_NO_PITCH=1) on tools/bench_case.py cases, same box, alternating
R=${GRAFT_REPO_ROOT:-$PWD}
VAR=${1:-BEOM_NO_PITCH}
for rep in 1 2; do
  for c in ${CASES:-sill jet soliton}; do
    for which in off on; do
      if [ $which = on ]; then export $VAR=1; else unset $VAR; fi
      echo "$VAR=$which $(python3 $R/tools/bench_case.py $c 100 2>&1 | tail -1 | cut -c1-170)" | tee -a $R/gpurun_out/ab_env.txt
    done
  done
done
unset $VAR
