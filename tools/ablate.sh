#!/bin/bash
# traffic attribution of the fused u+v sweep: BEOM_DBG bit0 = no ring evaluations, bit1 = no second update
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for dbg in 0 1 2 3; do
  export BEOM_DBG=$dbg
  O=$R/gpurun_out/abl_$dbg; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/w.log 2>&1
  echo "dbg=$dbg $(tail -1 $O/f.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()})')"
  python3 $R/tools/pmc_traffic.py $O/f $O/w | python3 -c '
import json,sys; t=json.load(sys.stdin); n=16785409*4
for k,v in t.items():
    if not k.endswith("_detail"): print("   ",k, round(v/n,1), {kk:round(vv/n,1) for kk,vv in t[k+"_detail"].items()})'
  rm -rf $O
done
