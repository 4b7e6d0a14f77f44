#!/bin/bash
# times (and optionally PMC traffic of) the headline bench under a list of BEOM_DBG values
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for dbg in ${DBG_LIST:-0 4 8 12}; do
  export BEOM_DBG=$dbg
  O=$R/gpurun_out/dbg_$dbg; rm -rf $O; mkdir -p $O
  python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/b.log 2>&1
  echo "dbg=$dbg $(tail -1 $O/b.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})')" | tee -a $R/gpurun_out/dbgsweep.txt
  if [ -n "$WITH_PMC" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/w.log 2>&1
  python3 $R/tools/pmc_traffic.py $O/f $O/w | python3 -c '
import json,sys; t=json.load(sys.stdin); n=16785409*4
for k,v in t.items():
    if not k.endswith("_detail"): print("   ",k, round(v/n,1), {kk:round(vv/n,1) for kk,vv in t[k+"_detail"].items()})' | tee -a $R/gpurun_out/dbgsweep.txt
  rm -rf $O/f $O/w
  fi
done
