#!/bin/bash
# The engine of another source tree (ab/<name>tree: e.g. `git archive <commit> beom_amd include tools | tar -x -C ab/r01tree`,
# built there) against this tree on tools/bench_case.py cases, same box, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}
OTHER=${OTHER:-ab/r01tree}
for rep in 1 2; do
  for c in ${CASES:-soliton stommel sill}; do
    for which in other this; do
      if [ $which = other ]; then T=$R/$OTHER; else T=$R; fi
      echo "$which $(cd $T && python3 tools/bench_case.py $c ${STEPS:-200} 2>&1 | tail -1 | cut -c1-230)" | tee -a $R/gpurun_out/ab_trees.txt
    done
  done
done
