#!/bin/bash
# what three workgroups per CU instead of four cost the fused u+v sweep by themselves: ab/pad.so = the in-tree kernels with
# 14 KB of unused LDS per workgroup; the Leith fold off in both
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})'
export BEOM_NO_FOLD_LEITH=1
for rep in 1 2; do
  for which in occ4 occ3; do
    unset BEOM_HIP_LIB
    [ $which = occ3 ] && export BEOM_HIP_LIB=$R/ab/pad.so
    echo "$which $(python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$fmt")" | tee -a $R/gpurun_out/r03_ab_pad.txt
  done
done
