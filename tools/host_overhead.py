#!/usr/bin/env python3
"""How much host time does one overlapped slab step cost?  Two slabs of one frame in ONE
process on one GPU (the loopback form of tests/test_gpu_parity.py), transfer = device copy.
Prints: host enqueue time per rank-step (tiny grid: the GPU is never the limit), and for an
N=8-sized slab pair (4096 x ~520 rows x 4) the wall time per step against the same two
slabs stepped without any exchange."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beom_amd import capi, inputs as I, slab
from beom_amd.grid import read_input_data


def make(lm, mm, nlay):
    p, files = I.case_headline(lm, mm, nlay)
    f = read_input_data(p, files=files)
    runs = []
    for g in slab.decompose(p.mm, p.lm, 2):
        e = capi.Engine(slab.slice_fields(f, g), slab_row0=g.row0, slab_mm=p.mm)
        r = slab.SlabRunner(e, g, p.nlay, dist=None, overlap=False)
        r.overlap = True
        r.main, r.comm = torch.cuda.Stream(), torch.cuda.Stream()
        r.main.wait_stream(torch.cuda.current_stream())
        e.set_stream(r.main.cuda_stream)
        runs.append(r)
    return runs


def loop(runs, t0, n, exchange=True):
    a, b = runs

    def xfer(dst, src_runner, src):
        def go():
            torch.cuda.current_stream().wait_event(src_runner._packed)
            dst.copy_(src, non_blocking=True)
        return go
    for t in range(t0, t0 + n):
        for r in runs:
            with torch.cuda.stream(r.main):
                if exchange and r._pending is not None and r.engine.step_phase(t, 1):
                    r._exchange_end()
                    r.engine.step_phase(t, 2)
                else:
                    r._exchange_end()
                    r.engine.step(t, 1, sync=False)
        if exchange:
            for r in runs:
                r._begin_pack()
            a._begin_transfer(xfer(a.recv_n, b, b.send_s))
            b._begin_transfer(xfer(b.recv_s, a, a.send_n))


for (lm, mm, nlay, n) in ((256, 128, 4, 400), (4096, 1024, 4, 60)):
    runs = make(lm, mm, nlay)
    loop(runs, 1, 10); [r.finish() for r in runs]; torch.cuda.synchronize()
    t1 = time.perf_counter(); loop(runs, 11, n); t2 = time.perf_counter()
    [r.finish() for r in runs]; torch.cuda.synchronize(); t3 = time.perf_counter()
    loop(runs, 11 + n, n, exchange=False); [r.finish() for r in runs]; torch.cuda.synchronize()
    t4 = time.perf_counter(); loop(runs, 11 + 2 * n, n, exchange=False); [r.finish() for r in runs]
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print("%dx%dx%d two slabs: host enqueue %.1f us per rank-step; wall %.3f ms per step pair with "
          "exchange, %.3f ms without" % (lm, mm, nlay, (t2 - t1) / n / 2 * 1e6, (t3 - t1) / n * 1e3,
                                         (t5 - t4) / n * 1e3), flush=True)
    for r in runs:
        r.engine.close()

# ---- the single-process form (beom_multi_*): same frame, 1 handle against 2 and 4 bands on this GPU
import time as _t
p, files = I.case_headline(4096, 1024, 4)
f = read_input_data(p, files=files)
for nb in (1, 2, 4):
    m = capi.MultiEngine(f, devices=[0] * nb)
    m.step(1, 10)
    t0 = _t.perf_counter(); m.step(11, 60); t1 = _t.perf_counter()
    print("beom_multi 4096x1024x4, %d band(s) on one GPU: %.3f ms per step, %s" % (nb, (t1 - t0) / 60 * 1e3, m.stats()), flush=True)
    m.close()
