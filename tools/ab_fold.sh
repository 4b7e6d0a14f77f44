#!/bin/bash
# the Leith fold on / off (option "fold_leith" through BEOM_NO_FOLD_LEITH), same box, alternating.  BENCH_ARGS="--case sill4" ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})'
for rep in 1 2 3; do
  for which in fold nofold; do
    unset BEOM_NO_FOLD_LEITH
    [ $which = nofold ] && export BEOM_NO_FOLD_LEITH=1
    echo "$which ${BENCH_ARGS} $(python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "$fmt")" | tee -a $R/gpurun_out/r03_ab_fold.txt
  done
done
