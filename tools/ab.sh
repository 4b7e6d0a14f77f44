#!/bin/bash
# A/B of two engine builds on the SAME box: ab/prev.so against the in-tree library, alternating
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export BEOM_HIP_LIB=$R/ab/prev.so; else unset BEOM_HIP_LIB; fi
    python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/ab_$which.log 2>&1
    echo "$which $(tail -1 $R/gpurun_out/ab_$which.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})')" | tee -a $R/gpurun_out/ab.txt
  done
done
