#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/a -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/a/**/*_counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "k_" not in n: continue
    agg[n.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: "%.3g"%(sum(x[len(x)//2:])/len(x[len(x)//2:])) for c,x in v.items()})
PY
rm -rf $O/a
