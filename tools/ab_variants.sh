#!/bin/bash
# Several engine builds (ab/<name>.so, e.g. other tile geometries: -DUV_WAVES=8, -DMV_Q=1) against the in-tree library on
# tools/bench_case.py cases, same box, alternating.  VARIANTS="base w8 q1" CASES="soliton jet" tools/ab_variants.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for c in ${CASES:-soliton jet stommel}; do
    for v in ${VARIANTS:-base}; do
      if [ $v = base ]; then unset BEOM_HIP_LIB; else export BEOM_HIP_LIB=$R/ab/$v.so; fi
      echo "$v $(python3 $R/tools/bench_case.py $c ${STEPS:-300} 2>&1 | tail -1 | cut -c1-220)" | tee -a $R/gpurun_out/ab_variants.txt
    done
  done
done
