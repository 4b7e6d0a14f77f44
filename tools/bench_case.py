#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (not the bench.py line):
python tools/bench_case.py soliton|jet|sill|stommel [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
case = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
def _case():
    return {"soliton": lambda: I.case_soliton(lm=2048, mm=256, dt_s=60.0),
            "jet": lambda: I.case_unstable_jet(lm=2048, mm=2048, nlay=2, dt_s=50.0),
            "sill": lambda: I.case_sill_exchange3d(lm=4096, mm=512, nlay=4, dt_s=30.0, npts=15, sill_halfwidth=50.0),
            "stommel": lambda: I.case_stommel(lm=128, mm=128, dl=100.0e3, dt_s=400.0),
            "beach": lambda: I.case_carrier_beach(lm=8192, mm=1024, nlay=8, dt_s=0.08),   # one GPU's share of config 5
            "wind": lambda: I.case_mixed_open_bc(lm=4096, mm=2048, npts=15),   # wind-driven, nudged, 2 layers
            "headline": lambda: I.case_headline(4096, 4096, 4),
            "headline_gather": lambda: I.case_headline(4096, 4096, 4),     # same frame through the neig tables
            "headline_land": lambda: with_land(I.case_headline(4096, 4096, 4)),
            "band8": lambda: I.case_headline(4096, 512, 4),                # what one of 8 / 4 / 2 bands of the headline frame holds
            "band4": lambda: I.case_headline(4096, 1024, 4),
            "band2": lambda: I.case_headline(4096, 2048, 4),
            "closed_small": lambda: I.case_headline(2048, 256, 1),         # the soliton's frame as a closed basin (no periodic seam)
            "closed_small4": lambda: I.case_headline(1024, 128, 4),        # the same number of cell-layers in 4 layers
            }[case]()


def with_land(pf):
    """Headline basin with a land mass (5 % of the cells): not a dense frame any more."""
    import numpy as np
    p, files = pf
    h = files["h_bo"]
    lm, mm = p.lm, p.mm
    x = np.arange(lm + 2)[:, None]; y = np.arange(mm + 2)[None, :]
    h[((x - 0.3 * lm) ** 2 + (y - 0.6 * mm) ** 2) < (0.126 * lm) ** 2] = 0.0
    files["init"][h == 0.0] = 0.0
    return p.replace(ndeg=I.get_nbr_deg_freedom(h)), files


p, files = _case()
f = read_input_data(p, files=files)
e = capi.Engine(f, dense_hint=0 if case == "headline_gather" else 1)
e.step(1, 10)
t = time.perf_counter(); e.step(11, K); dt = time.perf_counter() - t
ms, nl = e.profile_steps(11 + K, 50)
names = ("h", "mont", "visc", "u", "v", "mont+visc", "u+v", "mont+visc+u+v")
print(json.dumps({"case": case, "per_kernel_us": {names[k]: round(ms[k] / 50 * 1e3, 1) for k in range(len(nl)) if nl[k]}, "lm": p.lm, "mm": p.mm, "nlay": p.nlay, "steps": K, "us_per_step": dt / K * 1e6,
                  "cell_layer_updates_per_s": p.ndeg * p.nlay * K / dt, "dense": e.is_dense,
                  "kernel_us_per_step": round(sum(ms) / 50 * 1e3, 1)}))
