#!/usr/bin/env python3
"""What ONE band of the headline frame cut N ways costs per step on this GPU, with the whole exchange machinery
around it (rows from the recipe, beom_multi_create_local, pack -> transport -> unpack on the second stream,
split steps): the band's exchange is looped back (BEOM_XCHG_LOOPBACK: it receives its own edge rows), over the
shared-memory transport and over a one-rank RCCL communicator.  Beside it: the same rows as a plain slab handle
without any exchange, and a closed frame of the band's size.  Implied parallel efficiency at N bands =
t(whole frame) / (N * t(band)) — the part of the scaling curve that does not depend on the wire.

python tools/band_cost.py [N=8] [band=3] [steps=300] [lm=4096] [mm=4096] [nlay=4]"""
import json
import os
import sys
import time
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beom_amd import capi, inputs as I, slab
from beom_amd.grid import read_input_data

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lm = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
mm = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
nlay = int(sys.argv[6]) if len(sys.argv) > 6 else 4
whole_us = float(os.environ.get("BEOM_WHOLE_US", "0")) or None


def timed(eng, K, reps=int(os.environ.get("BEOM_COST_REPS", "3"))):
    eng.step(1, 30)
    best = 1e30
    for r in range(reps):
        t = time.perf_counter()
        eng.step(31 + r * K, K)
        best = min(best, (time.perf_counter() - t) / K * 1e6)
    return best


recipe = I.recipe_headline(lm, mm, nlay)
p = recipe.p
out = {"frame": "%dx%dx%d" % (lm, mm, nlay), "bands": N, "band": B, "steps": K}

if whole_us is None:
    e = capi.Engine(read_input_data(p, files=recipe.rows(0, p.mm + 1)))
    whole_us = timed(e, max(K // 4, 30))
    e.close()
out["whole_frame_us_per_step"] = round(whole_us, 1)
print(json.dumps(out), flush=True)

f, g, orphan = slab.build_band(recipe, N, B)
out["band_rows"] = {"owned": g.own1 - g.own0 + 1, "window": g.rows}

# the band's rows as an ordinary slab handle, no exchange at all (ghost rows go stale: timing only)
if not os.environ.get("BEOM_COST_SKIP_BASE"):
    e = capi.Engine(f, **slab.engine_slab_args(g))
    out["slab_handle_no_exchange_us"] = round(timed(e, K), 1)
    e.close()
    print(json.dumps(out), flush=True)

for transport in os.environ.get("BEOM_COST_TRANSPORTS", "shm,rccl").split(","):
    for overlap in [int(x) for x in os.environ.get("BEOM_COST_OVERLAP", "1,0").split(",")]:
        kw = dict(shm_name="/beom_cost_%d_%s" % (os.getpid(), uuid.uuid4().hex[:8])) if transport == "shm" else dict(rccl_id=capi.rccl_unique_id())
        try:
            eng = capi.BandEngine(f, p, N, B, device=0, orphan=orphan, loopback=True, **kw)
        except capi.BeomError as exc:
            out["%s_overlap%d" % (transport, overlap)] = "unavailable: %s" % str(exc)[:200]
            continue
        eng.set_option("overlap", overlap)
        us = timed(eng, K)
        st = eng.download(("hlay",))
        out["%s_%s_us" % (transport, "split" if overlap else "plain")] = round(us, 1)
        out["%s_%s_stats" % (transport, "split" if overlap else "plain")] = eng.stats()
        out["%s_%s_finite" % (transport, "split" if overlap else "plain")] = bool(np.isfinite(st["hlay"]).all())
        out["%s_%s_implied_efficiency_N%d" % (transport, "split" if overlap else "plain", N)] = round(whole_us / (N * us), 3)
        eng.close()
        print(json.dumps(out), flush=True)

if os.environ.get("BEOM_COST_SKIP_BASE"):
    sys.exit(0)
# a closed frame with as many rows as the band owns
pc, files = I.case_headline(lm, g.own1 - g.own0 + 1, nlay)
e = capi.Engine(read_input_data(pc, files=files))
out["closed_frame_of_band_size_us"] = round(timed(e, K), 1)
e.close()
print(json.dumps(out))
