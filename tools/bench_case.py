#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (not the bench.py line):
python tools/bench_case.py soliton|jet|sill|stommel [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
case = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p, files = {"soliton": lambda: I.case_soliton(lm=2048, mm=256, dt_s=60.0),
            "jet": lambda: I.case_unstable_jet(lm=2048, mm=2048, nlay=2, dt_s=50.0),
            "sill": lambda: I.case_sill_exchange3d(lm=4096, mm=512, nlay=4, dt_s=30.0, npts=15, sill_halfwidth=50.0),
            "stommel": lambda: I.case_stommel(lm=128, mm=128, dl=100.0e3, dt_s=400.0)}[case]()
f = read_input_data(p, files=files)
e = capi.Engine(f)
e.step(1, 10)
t = time.perf_counter(); e.step(11, K); dt = time.perf_counter() - t
ms, nl = e.profile_steps(11 + K, 50)
print(json.dumps({"case": case, "lm": p.lm, "mm": p.mm, "nlay": p.nlay, "steps": K, "us_per_step": dt / K * 1e6,
                  "cell_layer_updates_per_s": p.ndeg * p.nlay * K / dt, "dense": e.is_dense,
                  "kernel_us_per_step": round(sum(ms) / 50 * 1e3, 1)}))
