#!/usr/bin/env python3
"""PCIe-inclusive figures for the boundary: time of beom_upload_state, of a full and of
an output-sized beom_download_state, and of beom_download_outputs at the headline size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
p, files = I.case_headline(4096, 4096, 4)
f = read_input_data(p, files=files)
e = capi.Engine(f, upload=False)
st = {k: getattr(f, k) for k in capi.STATE_NAMES}
nbytes = sum(v.nbytes for v in st.values())
t = time.perf_counter(); e.upload(**st); t_up = time.perf_counter() - t
e.step(1, 10)
t = time.perf_counter(); full = e.download(); t_dn = time.perf_counter() - t
t = time.perf_counter(); part = e.download(("hlay", "u", "v")); t_dn3 = time.perf_counter() - t
h0r4 = np.ascontiguousarray(f.h_0[:, 1:].astype(np.float32))
t = time.perf_counter(); out = e.download_outputs(h0r4); t_out = time.perf_counter() - t
t = time.perf_counter(); out = e.download_outputs(None); t_out2 = time.perf_counter() - t
t = time.perf_counter(); e.step(11, 100); t_step = (time.perf_counter() - t) / 100
n = p.ndeg * p.nlay
print(json.dumps({"state_bytes": nbytes, "upload_s": round(t_up, 3), "upload_GBs": round(nbytes / t_up / 1e9, 1),
                  "download_full_s": round(t_dn, 3), "download_full_GBs": round(nbytes / t_dn / 1e9, 1),
                  "download_hlay_u_v_s": round(t_dn3, 3), "download_outputs_first_s": round(t_out, 3),
                  "download_outputs_s": round(t_out2, 3), "ms_per_step": round(t_step * 1e3, 3),
                  "updates_per_s_resident": n / t_step,
                  "updates_per_s_with_outputs_every_100_steps": n * 100 / (100 * t_step + t_out2),
                  "updates_per_s_with_full_state_round_trip_every_100_steps": n * 100 / (100 * t_step + t_up + t_dn)}))
