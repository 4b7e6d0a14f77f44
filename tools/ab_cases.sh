#!/bin/bash
# A/B of ab/prev.so against the in-tree library on tools/bench_case.py cases (same box)
R=${GRAFT_REPO_ROOT:-$PWD}
for c in ${CASES:-soliton headline_land sill}; do
  for which in prev new; do
    if [ $which = prev ]; then export BEOM_HIP_LIB=$R/ab/prev.so; else unset BEOM_HIP_LIB; fi
    echo "$which $(python3 $R/tools/bench_case.py $c 100 2>&1 | tail -1 | cut -c1-150)" | tee -a $R/gpurun_out/ab_cases.txt
  done
done
