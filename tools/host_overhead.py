#!/usr/bin/env python3
"""How much HOST time does one band-step of the multi-GPU loop cost now that the loop runs inside the library
(beom_multi_step; round 1 drove it from Python: ~100 us per rank-step)?

Two bands of one frame in ONE process on one GPU (peer copies; the RCCL form adds one grouped send/recv per
step).  (a) tiny frame: the GPU is never the limit, so wall time per band-step of an asynchronous call = host
enqueue time (launches, events, the exchange); (b) an N=8-sized band pair (4096 x ~520 rows x 4): wall time per
step against the single handle = what the exchange machinery costs on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data

for (lm, mm, nlay, n, bands) in ((256, 128, 4, 2000, (1, 2)), (4096, 1024, 4, 100, (1, 2, 4))):
    p, files = I.case_headline(lm, mm, nlay)
    f = read_input_data(p, files=files)
    for nb in bands:
        for transport in (("peer", capi.XCHG_PEER),) + ((("rccl-self-ring", capi.XCHG_RCCL),) if False else ()):
            m = capi.MultiEngine(f, devices=[0] * nb, transport=transport[1])
            m.step(1, 10)
            t0 = time.perf_counter()
            m.step(11, n, sync=False)                 # returns when everything is enqueued
            t1 = time.perf_counter()
            m.sync()
            t2 = time.perf_counter()
            print("%dx%dx%d, %d band(s), %s: host enqueue %.1f us per band-step, wall %.3f ms per step, %s"
                  % (lm, mm, nlay, nb, transport[0], (t1 - t0) / n / nb * 1e6, (t2 - t0) / n * 1e3, m.stats()), flush=True)
            m.close()

# the ring form over RCCL with one rank (the only RCCL form a one-GPU box can run): jet frame, one band
p, files = I.case_unstable_jet(lm=256, mm=128, nlay=2, dt_s=1.0)
f = read_input_data(p, files=files)
for tr, name in ((capi.XCHG_PEER, "peer"), (capi.XCHG_RCCL, "RCCL")):
    m = capi.MultiEngine(f, devices=[0], transport=tr, ring1=True)
    m.step(1, 10)
    t0 = time.perf_counter(); m.step(11, 2000, sync=False); t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
    print("jet 256x128x2 as a ring of ONE band + companion frame, %s: host enqueue %.1f us per step, wall %.3f ms per step"
          % (name, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e3), flush=True)
    m.close()
