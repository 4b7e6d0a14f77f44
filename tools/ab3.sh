#!/bin/bash
# A/B/C on the SAME box, alternating: the in-tree library (Leith fold, rvor/dive straight from global u, v), ab/lfstage.so (u, v
# staged in LDS first) and the in-tree library with the fold off.
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})'
for rep in 1 2; do
  for which in direct staged nofold; do
    unset BEOM_HIP_LIB BEOM_NO_FOLD_LEITH
    [ $which = staged ] && export BEOM_HIP_LIB=$R/ab/lfstage.so
    [ $which = nofold ] && export BEOM_NO_FOLD_LEITH=1
    echo "$which $(python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "$fmt")" | tee -a $R/gpurun_out/r03_ab3.txt
  done
done
