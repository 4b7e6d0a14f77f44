#!/bin/bash
# run-to-run spread of the headline bench on ONE box: N separate processes of the same build
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-8}); do
  python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/rep.log 2>&1
  echo "rep $rep $(grep '^{"metric"' $R/gpurun_out/rep.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})')" | tee -a $R/gpurun_out/repeat.txt
done
