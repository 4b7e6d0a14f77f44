#!/usr/bin/env python3
"""TEST/MEASUREMENT INFRASTRUCTURE — builds the reference Fortran (OpenMP, -O3) for the
bounded CPU-baseline sample of bench.py into oracle/_ref/baseline_<sample>/beom_ref.
Runs only where /root/reference exists; the binary then travels to the GPU box."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

SAMPLE = (1024, 1024, 4)
NSTEPS = 1200        # ~15-25 s of reference CPU work on a 16-core share


def params(lm, mm, nlay, nsteps=NSTEPS):
    from beom_amd import inputs
    p, _ = inputs.case_headline(lm, mm, nlay)
    dt_s = (nsteps + 0.2) * float(p.dt) / 86400.0
    return p.replace(dt_s="%.9f" % dt_s, dt_o="1000.")


def main():
    import ref_build
    lm, mm, nlay = SAMPLE
    out = os.path.join(HERE, "_ref", "baseline_%dx%dx%d" % SAMPLE)
    exe = os.path.join(out, "beom_ref")
    stamp = os.path.join(out, "stamp")
    src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("ref_build.py", "build_ref_baseline.py"))
    if os.path.exists(exe) and os.path.exists(stamp) and os.path.getmtime(stamp) >= src_m:
        return
    p = params(lm, mm, nlay)
    assert p.nstp == NSTEPS, p.nstp
    ref_build.build(p, out, "private_mod.f95", openmp=True, opt="-O3")
    open(stamp, "w").write("ok\n")
    print("built", exe)


if __name__ == "__main__":
    main()
