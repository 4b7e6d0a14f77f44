#!/bin/bash
# Several source trees (ab/t_<commit>, each built in place) on tools/bench_case.py cases, same box, alternating: which commit
# changed a configuration's step time?  TREES="ab/r01tree ab/t_133d62f ." CASES="soliton stommel" tools/ab_bisect.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for c in ${CASES:-soliton stommel}; do
    for t in ${TREES:-.}; do
      echo "$t $(cd $R/$t && python3 tools/bench_case.py $c ${STEPS:-300} 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["case"], round(d["us_per_step"],1), d["per_kernel_us"])')" | tee -a $R/gpurun_out/ab_bisect.txt
    done
  done
done
