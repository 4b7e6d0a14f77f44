#!/usr/bin/env python3
"""End-to-end drop-in check with timing: the reference program (its own main.f95 + shared_mod.f95 +
private_mod.f95, flang -O3 -fopenmp) against the SAME main.f95/shared_mod.f95 with
beom_amd/host/beom_host_mod.f95 + libbeom_hip.so, both reading idir/*.bin and writing odir/*.

  python tools/host_e2e.py prepare      # here (needs /root/reference): builds both programs under oracle/_ref/e2e/
  python tools/host_e2e.py run          # GPU box: runs both, compares every output file byte for byte, prints JSON
"""
import filecmp, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.join(ROOT, "oracle", "_ref", "e2e")
NSTEPS, NOUT = 4000, 1000


WHICH = os.environ.get("BEOM_E2E_CASE", "soliton")       # soliton (BASELINE config 2) | jet (config 3)
if WHICH == "jet":
    NSTEPS, NOUT = 400, 200
    OUT = os.path.join(ROOT, "oracle", "_ref", "e2e_jet")


def case():
    from beom_amd import inputs as I
    if WHICH == "jet":
        p, files = I.case_unstable_jet(lm=2048, mm=2048, nlay=2, dt_s=50.0)
    else:
        p, files = I.case_soliton(lm=2048, mm=256, dt_s=1.0)
    dt = float(p.dt)
    p = p.replace(dt_s="%.9f" % ((NSTEPS + 0.2) * dt / 86400.0), dt_o="%.9f" % ((NOUT + 0.01) * dt / 86400.0))
    assert p.nstp == NSTEPS and p.notp == NOUT, (p.nstp, p.notp)
    return p, files


def prepare():
    import ref_build
    from beom_amd.host import build_host
    p, _ = case()
    os.makedirs(OUT, exist_ok=True)
    ref_build.build(p, os.path.join(OUT, "ref"), "private_mod.f95", openmp=True, opt="-O3")
    build_host.build(p, os.path.join(OUT, "gpu", "beom_gpu"), opt="-O2")
    print("built", os.listdir(OUT))


def run():
    import ref_build, oracle_lib
    from beom_amd import inputs as I
    p, files = case()
    threads = oracle_lib.host_cores()
    res = {"case": "%s %dx%dx%d, %d steps, output every %d" % (WHICH, p.lm, p.mm, p.nlay, NSTEPS, NOUT), "cpu_threads": threads}
    dirs = {}
    for who, exe in (("gpu", os.path.join(OUT, "gpu", "beom_gpu")), ("ref", os.path.join(OUT, "ref", "beom_ref"))):
        wd = tempfile.mkdtemp(prefix="beom_e2e_%s_" % who)
        dirs[who] = wd
        I.write_inputs(wd, files)
        t0 = time.perf_counter()
        if who == "ref":
            ref_build.run(exe, wd, threads=threads, timeout=1500)
        else:
            r = subprocess.run([exe], cwd=wd, capture_output=True, text=True, timeout=1500)
            assert r.returncode == 0 and "ERROR CODE" not in r.stdout + r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
        res[who + "_wall_s"] = round(time.perf_counter() - t0, 2)
        print(who, res[who + "_wall_s"], "s", flush=True)
    names = sorted(f for f in os.listdir(dirs["ref"]) if f.endswith(".bin") and f[:-4] not in files
                   and not f.startswith("oracle_"))
    res["identical_files"] = [f for f in names if filecmp.cmp(os.path.join(dirs["ref"], f), os.path.join(dirs["gpu"], f), shallow=False)]
    res["different_files"] = [f for f in names if f not in res["identical_files"]]
    res["speedup_wall"] = round(res["ref_wall_s"] / res["gpu_wall_s"], 1)
    for d in dirs.values():
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    {"prepare": prepare, "run": run}[sys.argv[1]]()
