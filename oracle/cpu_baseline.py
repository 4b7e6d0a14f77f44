"""MEASUREMENT INFRASTRUCTURE for bench.py's cpu_baseline leg: times the reference
Fortran binary (kind 'reference') — or, if it is absent, the C restatement (kind 'port')
— on the host cores over a bounded sample of the headline workload."""
from __future__ import annotations

import os
import re
import shutil
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (ROOT, HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def run(sample: str = "1024x1024x4") -> dict:
    import numpy as np
    from beom_amd import inputs
    import build_ref_baseline as brb
    lm, mm, nlay = (int(x) for x in sample.split("x"))
    import oracle_lib
    cores = oracle_lib.host_cores()
    cores = int(os.environ.get("BEOM_CPU_BASELINE_THREADS", cores))
    nsteps = brb.NSTEPS
    p = brb.params(lm, mm, nlay, nsteps)
    _, files = inputs.case_headline(lm, mm, nlay)
    exe = os.path.join(HERE, "_ref", "baseline_%dx%dx%d" % (lm, mm, nlay), "beom_ref")
    t_from, t_to = 4, nsteps
    units = float(p.ndeg) * nlay * (t_to - t_from)
    if os.path.exists(exe):
        import ref_build
        wd = tempfile.mkdtemp(prefix="beom_cpu_baseline_")
        try:
            inputs.write_inputs(wd, files)
            out = ref_build.run(exe, wd, t_from=t_from, t_to=t_to, threads=cores, timeout=1500)
        finally:
            shutil.rmtree(wd, ignore_errors=True)
        m = re.search(r"ORACLE_TIMER\s+(\d+)\s+(\d+)\s+([0-9.Ee+-]+)", out)
        if not m:
            raise RuntimeError("no ORACLE_TIMER line in reference output")
        secs = float(m.group(3))
        return {"value": units / secs, "unit": "cell-layer updates/s", "cores": cores, "kind": "reference",
                "sample": "reference Fortran (flang -O3 -fopenmp, %d threads), %dx%dx%d layers, time steps %d..%d "
                          "(%.1f s)" % (cores, lm, mm, nlay, t_from + 1, t_to, secs),
                "GBs_at_416B": 416.0 * units / secs / 1e9}
    # fallback: the C restatement, OpenMP
    import oracle_lib
    from beom_amd.grid import read_input_data
    f = read_input_data(p, files=files)
    o = oracle_lib.Oracle(f, per_layer_scratch=False)
    o.step(1, t_from)
    t0 = time.perf_counter()
    o.step(t_from + 1, t_to - t_from)
    secs = time.perf_counter() - t0
    return {"value": units / secs, "unit": "cell-layer updates/s", "cores": cores, "kind": "port",
            "sample": "C restatement (gcc -O2 -fopenmp, %d threads), %dx%dx%d layers, time steps %d..%d (%.1f s)"
                      % (cores, lm, mm, nlay, t_from + 1, t_to, secs),
            "GBs_at_416B": 416.0 * units / secs / 1e9}


if __name__ == "__main__":
    import json
    print(json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "1024x1024x4")))
