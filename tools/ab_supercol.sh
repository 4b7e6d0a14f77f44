#!/bin/bash
# A/B of the TileMap strip width (BEOM_SUPERCOL, tiles; 0 = whole frame width) on the bench line, same box, alternating
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for w in ${WIDTHS:-0 4 8 16 32}; do
    export BEOM_SUPERCOL=$w
    python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print('supercol=$w ms/step %.3f  h %.3f  mont+visc %.3f  u+v %.3f' % (d['ms_per_step'], r['update_h']['avg_ms'], r['update_mont+update_viscosity']['avg_ms'], r['update_u+update_v']['avg_ms']))" | tee -a $R/gpurun_out/ab_supercol.txt
  done
done
