#!/usr/bin/env python3
"""Soak of the overlapped band exchange: many steps, several band counts, against the single handle
(bitwise), at sizes where phase 1 of a step really runs while the previous step's ghosts are copied."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600


def with_land(pf, sponge_obc=False):
    """an island across the band seams and a ragged coast in one corner: bands of packed rows, dealt by cell count"""
    p, files = pf
    files = {k: np.array(v, dtype=np.float64) for k, v in files.items()}
    h = files["h_bo"]
    x = np.arange(p.lm + 2)[:, None]; y = np.arange(p.mm + 2)[None, :]
    land = ((x - 0.3 * p.lm) ** 2 + (y - 0.55 * p.mm) ** 2) < (0.2 * p.mm) ** 2
    land = land | ((x > 0.8 * p.lm) & (y < 0.3 * p.mm) & ((x + y) % 7 != 0))
    h[land] = 0.0
    if "init" in files:
        files["init"][land] = 0.0
    return p.replace(ndeg=I.get_nbr_deg_freedom(h)), files


for name, mk in (("headline 4096x1024x4", lambda: I.case_headline(4096, 1024, 4)),
                 ("headline 2048x1024x4 with land (bands of packed rows)", lambda: with_land(I.case_headline(2048, 1024, 4))),
                 ("sill 1024x1024x3 with land (nudging, ocrp)", lambda: with_land(I.case_sill_exchange3d(lm=1024, mm=1024, nlay=3, dt_s=30.0, npts=15, sill_halfwidth=50.0))),
                 ("sill 2048x1024x4 (nudging, ocrp)", lambda: I.case_sill_exchange3d(lm=2048, mm=1024, nlay=4, dt_s=30.0, npts=15, sill_halfwidth=50.0)),
                 ("beach 4096x512x2 (no Leith)", lambda: I.case_carrier_beach(lm=4096, mm=512, nlay=2, dt_s=0.08))):
    p, files = mk()
    f = read_input_data(p, files=files)
    one = capi.Engine(f); one.step(1, N); ref = one.download(("hlay", "u", "v", "h_u", "h_v")); one.close()
    for nb in (2, 5, 8):
        m = capi.MultiEngine(f, devices=[0] * nb)
        t = time.perf_counter()
        for k in range(0, N, 97):                       # uneven call lengths: pending exchanges cross calls
            m.step(1 + k, min(97, N - k), sync=False)
        got = m.download(("hlay", "u", "v", "h_u", "h_v")); dt = time.perf_counter() - t
        ok = all(np.array_equal(ref[k], got[k]) for k in ref)
        print(json.dumps({"case": name, "bands": nb, "steps": N, "bitwise_equal": ok, "stats": m.stats(), "finite": bool(np.isfinite(got["hlay"]).all())}), flush=True)
        m.close()
        assert ok
