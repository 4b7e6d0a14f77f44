#!/usr/bin/env python3
"""Rigid lid (rgld = 1) at a size beyond the toy frames: a wind-driven closed basin 1024 x 1024 x 2.  GPU: ms per step of the
whole lid step and of what surf_pressure's Gauss-Seidel iteration costs in it (sweeps per solve, launches per solve — the
pipeline of wavefronts of beom_engine.hip lid_solve); CPU: the reference Fortran itself (oracle/_ref/lid_1024x1024x2, built
by oracle/build_ref_baseline.py where /root/reference exists) on this host's cores, the same steps.  The lid pressure after
the last step must equal the oracle's serial sweeps bit for bit.

python tools/lid_cost.py [steps=12]"""
import json
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import build_ref_baseline as brb
import oracle_lib
import ref_build
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data

N = int(sys.argv[1]) if len(sys.argv) > 1 else brb.LID_NSTEPS
lm, mm, nlay = brb.LID_SAMPLE
out = {"frame": "%dx%dx%d" % (lm, mm, nlay), "steps": N}

# the reference first: nothing in this process has touched the GPU yet
exe = os.path.join(ROOT, "oracle", "_ref", "lid_%dx%dx%d" % brb.LID_SAMPLE, "beom_ref")
_, files = I.case_headline(lm, mm, nlay)
if os.path.exists(exe):
    cores = oracle_lib.host_cores()
    wd = tempfile.mkdtemp(prefix="beom_lid_")
    try:
        I.write_inputs(wd, files)
        txt = ref_build.run(exe, wd, t_from=3, t_to=brb.LID_NSTEPS, threads=cores, timeout=1500)
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    m = re.search(r"ORACLE_TIMER\s+(\d+)\s+(\d+)\s+([0-9.Ee+-]+)", txt)
    out["reference_fortran_ms_per_step"] = round(float(m.group(3)) / (brb.LID_NSTEPS - 3) * 1e3, 1)
    out["reference_threads"] = cores
print(json.dumps(out), flush=True)

p = brb.lid_params(nsteps=N)
f = read_input_data(p, files=files)
e = capi.Engine(f)
e.step(1, 3)
s0, l0, n0 = e.info("lid_sweeps"), e.info("lid_launches"), e.info("lid_solves")
t = time.perf_counter(); e.step(4, N - 3); dt = time.perf_counter() - t
solves = e.info("lid_solves") - n0
out.update(gpu_ms_per_step=round(dt / (N - 3) * 1e3, 2), sweeps_per_solve=round((e.info("lid_sweeps") - s0) / solves, 1),
           launches_per_solve=round((e.info("lid_launches") - l0) / solves, 1), sweep_distance=e.info("lid_sweep_distance"))
print(json.dumps(out), flush=True)
# the same steps of a free-surface run of this frame, for scale
ef = capi.Engine(read_input_data(p.replace(rgld="0.", g_fb="0."), files=files))
ef.step(1, 3)
t = time.perf_counter(); ef.step(4, N - 3); out["free_surface_ms_per_step"] = round((time.perf_counter() - t) / (N - 3) * 1e3, 3)
ef.close()
# parity at this size: the oracle's serial sweeps
o = oracle_lib.Oracle(f)
t = time.perf_counter(); o.step(1, N); out["oracle_c_serial_ms_per_step"] = round((time.perf_counter() - t) / N * 1e3, 1)
pi = e.download_pressure()
out["pi_s_bitwise_equal_to_oracle"] = bool(np.array_equal(pi.view(np.uint64), o.rgld["pi_s"].view(np.uint64)))
out["pi_s_max_abs"] = float(np.abs(pi).max())
st = e.download(("hlay", "u", "v"))
out["state_bitwise_equal_to_oracle"] = all(bool(np.array_equal(st[k].view(np.uint64), o.state()[k].view(np.uint64))) for k in st)
print(json.dumps(out))
