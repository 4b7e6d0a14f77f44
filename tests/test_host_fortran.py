"""The Fortran-95 host (beom_amd/host): `module private_mod` exporting `run`, calling the
HIP engine through iso_c_binding.

CPU: it compiles and links (a) against a generated shared_mod + a 6-line main, and
(b) — where /root/reference exists — against the reference's UNCHANGED main.f95 and
shared_mod.f95: the drop-in.
GPU: the built executable is run on the golden cases' input files and every output file
of the reference's contract (grid.bin, h_0.bin, param_basin.txt, time.txt, eta_, u___,
v___ and the diag fields) is compared byte for byte with the file the real reference
wrote (tide case: real*4 values within 2e-6 relative — device cos())."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from beom_amd import inputs
from beom_amd.host import build_host
from helpers import Golden, golden_names

REF = "/root/reference"
HAVE_FLANG = os.path.exists(build_host.FLANG)


@pytest.mark.skipif(not HAVE_FLANG, reason="flang not present")
def test_host_compiles_with_generated_shared_mod(tmp_path):
    p, _ = inputs.case_stommel(lm=24, mm=16, dt_s=0.2)
    exe = build_host.build(p, str(tmp_path / "beom_gpu"))
    assert os.path.exists(exe)
    nm = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    for sym in ("beom_create", "beom_upload_state", "beom_step", "beom_download_outputs", "beom_sync"):
        assert sym in nm, sym          # the time loop goes through the C-ABI
    # (beom_download_state is only referenced when shared_mod's `diag` parameter is > 0.5: flang
    #  folds the constant and drops the dead branch.)


@pytest.mark.skipif(not (HAVE_FLANG and os.path.isdir(REF)), reason="needs flang and /root/reference")
def test_drop_in_under_unchanged_reference_main_and_shared_mod(tmp_path):
    """main.f95 and shared_mod.f95 are taken from the reference as they are."""
    exe = build_host.build(None, str(tmp_path / "beom_gpu_ref"),
                           shared_mod_path=os.path.join(REF, "shared_mod.f95"),
                           main_path=os.path.join(REF, "main.f95"))
    assert os.path.exists(exe)


def _run_host(g, work, ngpu=1):
    exe = build_host.build(g.p, os.path.join(work, "beom_gpu"), variant=g.variant)
    inputs.write_inputs(work, g.files)
    for fn, data in g.pre.items():       # a restarted run (rsta = 1) continues in the directory of the run before it
        with open(os.path.join(work, fn), "w" if isinstance(data, str) else "wb") as fh:
            fh.write(data)
    env = dict(os.environ)
    if ngpu > 1:       # bands of rows on "several" devices: all of them the one GPU of the box
        env.update(BEOM_NGPU=str(ngpu), BEOM_MULTI_WRAP_DEVICES="1")
    r = subprocess.run([exe], cwd=work, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ERROR CODE" not in r.stderr, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_FLANG, reason="flang not present")
@pytest.mark.parametrize("name,ngpu", [(n, 1) for n in golden_names()]
                         + [("stommel_24x16", 2), ("sill_4l_ocrp", 3), ("tide_sponge", 2), ("variant3d_3l", 2),
                            ("jet_2l_xyper", 2), ("tc_conservation_xyper_stdfb", 3),      # periodic in y: a ring of bands
                            ("obc_mcbc0_2l", 2), ("obc_mcbc0_2l", 3),                       # open-boundary segments dealt to the bands
                            ("obc_mcbc0_yper_2l", 2), ("obc_mcbc0_yper_2l", 3),             # ... of a ring of bands (+ the companion frame)
                            ("island_3l_forced", 2), ("tc_outcrop_seamount_3d_3l", 2), ("topdrag_topo_2l", 3)])   # land: bands of packed rows
def test_fortran_host_reproduces_reference_output_files(name, ngpu):
    """ngpu > 1: the same program with BEOM_NGPU set — the library cuts the frame into row bands
    (beom_multi_*), the Fortran side stays one process."""
    g = Golden(name)
    work = tempfile.mkdtemp(prefix="beom_host_")
    try:
        out = _run_host(g, work, ngpu)
        assert "MI355X engine" in out
        assert ngpu == 1 or "row bands" in out
        for key in [k for k in g.z.files if k.startswith("file_")]:
            fn = key[5:].replace("_bin", ".bin").replace("_txt", ".txt")
            path = os.path.join(work, fn)
            assert os.path.exists(path), fn
            if fn.endswith(".txt"):
                assert open(path).read() == str(g.z[key]), fn
                continue
            mine = np.fromfile(path, dtype=np.uint8)
            ref = g.z[key]
            if g.uses_cos() and fn in ("eta_.bin", "u___.bin", "v___.bin"):
                a = mine.view("<f4").astype(np.float64); b = ref.view("<f4").astype(np.float64)
                assert a.shape == b.shape and np.max(np.abs(a - b)) <= 2e-6 * max(np.max(np.abs(b)), 1e-30), fn
            else:
                assert np.array_equal(mine, ref), fn
    finally:
        shutil.rmtree(work, ignore_errors=True)
