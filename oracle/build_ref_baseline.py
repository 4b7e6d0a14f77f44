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


# rigid lid (rgld = 1; tools/lid_cost.py): a wind-driven closed basin, 1024 x 1024 x 2, a dozen steps
LID_SAMPLE = (1024, 1024, 2)
LID_NSTEPS = 12


def lid_params(lm=LID_SAMPLE[0], mm=LID_SAMPLE[1], nlay=LID_SAMPLE[2], nsteps=LID_NSTEPS, tauw=("0.5", "0.2")):
    from beom_amd import inputs
    p, _ = inputs.case_headline(lm, mm, nlay)
    dt_s = (nsteps + 0.2) * float(p.dt) / 86400.0
    return p.replace(dt_s="%.9f" % dt_s, dt_o="1000.", rgld="1.", ocrp="1.", g_fb="0.", bdrg="2.e-4", tauw=list(tauw))


def build_lid():
    import ref_build
    out = os.path.join(HERE, "_ref", "lid_%dx%dx%d" % LID_SAMPLE)
    exe, stamp = os.path.join(out, "beom_ref"), os.path.join(out, "stamp")
    src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("ref_build.py", "build_ref_baseline.py"))
    if os.path.exists(exe) and os.path.exists(stamp) and os.path.getmtime(stamp) >= src_m:
        return
    ref_build.build(lid_params(), out, "private_mod.f95", openmp=True, opt="-O3")
    open(stamp, "w").write("ok\n")
    print("built", exe)


def main():
    import ref_build
    build_lid()
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
