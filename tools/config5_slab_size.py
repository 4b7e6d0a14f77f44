#!/usr/bin/env python3
"""BASELINE config 5 (8192 x 8192 x 8 layers on 8 GPUs, carrier-beach bathymetry, ocrp = 1, western
sponge) at the size ONE of its GPUs holds: by default 8192 x 4096 x 8 = four bands of 8192 x 1024 x 8,
all on the one GPU of the box through the single-process form (beom_multi_*).  Checks, at that size:
  * the banded run (ghost exchange + overlapped split steps) == the single handle, bit for bit;
  * the volume of every layer is conserved to rounding away from the sponge (closed basin + sponge:
    reported, not asserted);
and reports time per step.  Usage: python tools/config5_slab_size.py [mm] [bands] [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
mm = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t0 = time.time()
p, files = I.case_carrier_beach(lm=8192, mm=mm, nlay=8, dt_s=0.08)
print("recipe %.0f s" % (time.time() - t0), flush=True)
f = read_input_data(p, files=files)
del files
import resource
def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
if rss_gb() > 215.0:
    raise SystemExit("host state needs %.0f GB: too close to the box's memory cap" % rss_gb())
print("host state %.0f s, ndeg=%d nlay=%d (%.1f M cell-layers)" % (time.time() - t0, p.ndeg, p.nlay, p.ndeg * p.nlay / 1e6), flush=True)
print("host peak RSS %.1f GB" % (__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
one = capi.Engine(f)
assert one.is_dense
one.step(1, 3)
t = time.perf_counter(); one.step(4, K); ms1 = (time.perf_counter() - t) / K * 1e3
ref = one.download(("hlay", "u", "v"))
one.close()
print("single handle: %.2f ms per step (%.3g cell-layer updates/s)  %.0f s" % (ms1, p.ndeg * p.nlay / ms1 * 1e3, time.time() - t0), flush=True)
many = capi.MultiEngine(f, devices=[0] * nb)
many.step(1, 3)
t = time.perf_counter(); many.step(4, K); msn = (time.perf_counter() - t) / K * 1e3
got = many.download(("hlay", "u", "v"))
st = many.stats()
many.close()
same = {k: bool(np.array_equal(ref[k], got[k])) for k in ref}
print(json.dumps({"frame": "8192x%dx8 carrier beach, ocrp=1" % mm, "bands": nb, "steps": 3 + K,
                  "single_ms_per_step": round(ms1, 2), "banded_ms_per_step_one_gpu": round(msn, 2),
                  "bitwise_equal": same, "band_steps": st, "finite": bool(np.isfinite(got["hlay"]).all()),
                  "max_abs_u": float(np.max(np.abs(got["u"]))), "wall_s": round(time.time() - t0),
                  "host_peak_rss_GB": round(rss_gb(), 1)}), flush=True)
assert all(same.values())
