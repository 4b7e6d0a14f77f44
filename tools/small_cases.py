#!/usr/bin/env python3
"""Native-size reference testcases: microseconds per time step on the GPU (engine) against the
C restatement of the reference on the host (1 thread and all threads) — the small 2-D cases are
launch-latency-bound on a GPU."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
import oracle_lib
cases = {"tide_ridge 501x1x7": lambda: I.case_tide_ridge(), "lock_exchange 160x1x2": lambda: I.case_lock_exchange(),
         "wave_sponge 91x91x2": lambda: I.case_wave_sponge(), "conservation 61x61x2": lambda: I.case_conservation(),
         "stommel 100x63x1": lambda: I.case_stommel(), "sill_exchange2D 2001x1x2": lambda: I.case_sill_exchange2d(),
         "soliton 307x153x1": lambda: I.case_soliton()}
for name, mk in cases.items():
    p, files = mk()
    f = read_input_data(p, files=files)
    e = capi.Engine(f)
    e.step(1, 20)
    t = time.perf_counter(); e.step(21, 2000); g = (time.perf_counter() - t) / 2000
    e.close()
    res = {"case": name, "gpu_us_per_step": round(g * 1e6, 1)}
    th = int(os.environ.get("OMP_NUM_THREADS", "1"))         # read once when the oracle library loads
    o = oracle_lib.Oracle(f, per_layer_scratch=False)
    o.step(1, 20)
    t = time.perf_counter(); o.step(21, 300); c = (time.perf_counter() - t) / 300
    res["cpu_us_per_step_%dthr" % th] = round(c * 1e6, 1)
    print(json.dumps(res), flush=True)
