#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — builds the *real* reference (zhazorken/beom Fortran) into
``oracle/_ref/<name>/beom_ref`` so that the C restatement (oracle/beom_oracle.c) and
the golden fixtures (tests/golden) can be pinned against it.

Nothing under ``beom_amd/`` may import or execute this.  Only usable where
``/root/reference`` exists (this container); the GPU box receives the built binary
(oracle/_ref is git-ignored but not gpurun-ignored) and no reference source.

Recipe: the three reference files are read where they lie, patched
*in memory*, written to a throw-away scratch directory outside the repo, compiled
with AMD flang, and the scratch directory is deleted.  Patches:

  P0  the user block of shared_mod.f95:38-79 is replaced by the configuration
      (that block is what a reference user edits by hand for every run).
  P1  adds the three parameters the fork uses but never declares
      (``svis, tdrg, topt`` — private_mod.f95:780,1043,1267 vs shared_mod.f95:41-77).
      Without this the reference does not compile with any compiler.
  P2  private_mod.f95:1654 indexes hlay(:,2) in dead rigid-lid code; with nlay=1
      flang rejects it.  Subscript clamped to min(2,nlay); never executed (rgld=0).
  P3  private_mod.f95:2866-2869 reads ior4(ipnt, 2) when nlay=1 (out of bounds;
      eta_.bin is garbage).  Guarded with ``if (nlay > 1)``.
  P4  FP64 dump hooks.  The module is ``private`` with the single export ``run`` and
      writes real*4 only, so parity needs a hook: ``oracle_dump_static`` after
      read_input_data (:101) and ``oracle_dump(tstp)`` after each time step
      (:1867,1871,1875,1906).  Controlled at run time by ``<odir>oracle_ctl.txt``
      (absent → no dumps): ``dump_upto dump_every t_from t_to``.
No arithmetic statement of the reference is altered by P1-P4.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

REF = os.environ.get("BEOM_REFERENCE", "/root/reference")
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")

DUMP_CODE = r"""
subroutine oracle_ctl( dump_upto, dump_every, t_from, t_to )
  implicit none
  integer, intent(out) :: dump_upto, dump_every, t_from, t_to
  logical :: is_e
  integer :: unum, ios
  dump_upto = -1; dump_every = 0; t_from = -1; t_to = -1
  inquire( exist = is_e, file = trim(odir) // 'oracle_ctl.txt' )
  if ( .not. is_e ) return
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'oracle_ctl.txt', action = 'read', status = 'old' )
  read( unum, *, iostat = ios ) dump_upto, dump_every, t_from, t_to
  close( unum )
end subroutine oracle_ctl

subroutine oracle_dump_static()
  implicit none
  integer :: dump_upto, dump_every, t_from, t_to, unum
  call oracle_ctl( dump_upto, dump_every, t_from, t_to )
  if ( dump_upto < 0 .and. dump_every <= 0 ) return
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'oracle_static.bin', access = 'stream', &
        form = 'unformatted', status = 'replace', action = 'write' )
  write( unum ) int(lm, 4), int(mm, 4), int(nlay, 4), int(ndeg, 4)
  write( unum ) neig, subc
  write( unum ) mk_u, mk_v, mk_n, mkpe, mkpi, fcor, h_th, h_to
  write( unum ) nudg, fnud, hdot, tide, w_ti, bodf, taus
  write( unum ) real(invf, 8), real(dt, 8)
  write( unum ) hlay, u, v
  if ( rgld > 0.5_rw ) write( unum ) pi_s, Ow, Os, Osum_
  close( unum )
end subroutine oracle_dump_static

subroutine oracle_dump( tstp )
  implicit none
  integer, intent(in) :: tstp
  integer, save :: dump_upto = -2, dump_every, t_from, t_to
  integer(8), save :: c_from
  integer(8) :: c_now, c_rate
  integer :: unum
  character(len = 6) :: tag
  if ( dump_upto == -2 ) call oracle_ctl( dump_upto, dump_every, t_from, t_to )
  if ( tstp == t_from ) call system_clock( c_from )
  if ( tstp == t_to ) then
    call system_clock( c_now, c_rate )
    write( ioso, * ) 'ORACLE_TIMER', t_from, t_to, real(c_now - c_from, 8) / real(c_rate, 8)
  end if
  if ( tstp <= dump_upto .or. ( dump_every > 0 .and. mod(tstp, max(dump_every,1)) == 0 ) ) then
    write( tag, '(i6.6)' ) tstp
    unum = get_un()
    open( unit = unum, file = trim(odir) // 'oracle_step_' // tag // '.bin', access = 'stream', &
          form = 'unformatted', status = 'replace', action = 'write' )
    write( unum ) hlay, u, v, h_u, h_v
    write( unum ) rs_h, dmdx, dmdy
    write( unum ) v_cc, v_ll
    write( unum ) mont, rvor, pvor, dive, d2hx, d2hy
    write( unum ) tt3d, tb3d, tu3d
    write( unum ) real(ctim, 8), real(ramp, 8), real(gene, 8)
    if ( svis > 0._rw ) write( unum ) delu, delv, UU4, VV4
    if ( rgld > 0.5_rw ) write( unum ) pi_s
    close( unum )
  end if
end subroutine oracle_dump
"""


def patch_shared(text: str, block: str) -> str:
    beg = text.index("!<=============BEGINNING OF USER-MODIFIABLE SECTION")
    end = text.index("!<=============END OF USER-MODIFIABLE SECTION")
    beg_eol = text.index("\n", beg) + 1
    return text[:beg_eol] + block + text[end:]          # P0 (+P1 via block)


def patch_private(text: str) -> str:
    # P2
    old = "hlay(ipnt, 2) = hlay(ipnt,2)-0.5*real"
    assert text.count(old) == 1, "P2 anchor"
    text = text.replace(old, "hlay(ipnt, min(2,nlay)) = hlay(ipnt,min(2,nlay))-0.5*real")
    # P3
    pat = re.compile(
        r"(    ilay = 1\n    if \(rgld < 0\.5_rw\) then\n       do ipnt=1, ndeg\n)"
        r"(          ior4\( ipnt, ilay \)\s*&\n"
        r"            = real\(\s*hlay\( ipnt, ilay\s*\)\s*&\n"
        r"                  - real\( h_0\(\s*ipnt, ilay\s*\), r8 \) &\n)"
        r"(                  \+ real\( ior4\( ipnt, ilay \+ 1 \), r8 \), r4 \)\n)")
    m = pat.search(text)
    assert m, "P3 anchor"
    guarded = (m.group(1) + "        if ( nlay > 1 ) then\n" + m.group(2)
               + m.group(3).replace("ilay + 1", "min(ilay + 1, nlay)")
               + "        else\n"
               + "          ior4( ipnt, ilay ) = real( hlay( ipnt, ilay ) - real( h_0( ipnt, ilay ), r8 ), r4 )\n"
               + "        end if\n")
    text = text[:m.start()] + guarded + text[m.end():]
    # P4
    n = text.count("  call first_three_timesteps( tstp )\n")
    assert n == 3, "P4 anchor a"
    text = text.replace("  call first_three_timesteps( tstp )\n",
                        "  call first_three_timesteps( tstp )\n  call oracle_dump( tstp )\n")
    old = "    call gener_forward_backward( tstp, upst )\n"
    assert text.count(old) == 1, "P4 anchor b"
    text = text.replace(old, old + "    call oracle_dump( tstp )\n")
    old = "  call read_input_data()\n  call integrate_time ()\n"
    assert text.count(old) == 1, "P4 anchor c"
    text = text.replace(old, "  call read_input_data()\n  call oracle_dump_static()\n  call integrate_time ()\n")
    old = "end module private_mod"
    assert text.count(old) == 1
    text = text.replace(old, DUMP_CODE + "\n" + old)
    return text


def build(params, out_dir: str, variant: str = "private_mod.f95", openmp: bool = False,
          opt: str = "-O2") -> str:
    """params: beom_amd.params.Params.  Returns path of the binary."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree %s not present (GPU box?)" % REF)
    out_dir = os.path.abspath(out_dir)
    os.makedirs(out_dir, exist_ok=True)
    scratch = tempfile.mkdtemp(prefix="beom_refbuild_", dir="/tmp")
    try:
        with open(os.path.join(REF, "shared_mod.f95")) as f:
            shared = patch_shared(f.read(), params.fortran_block())
        with open(os.path.join(REF, variant)) as f:
            private = patch_private(f.read())
        for name, txt in (("shared_mod.f95", shared), ("private_mod.f95", private)):
            with open(os.path.join(scratch, name), "w") as f:
                f.write(txt)
        exe = os.path.join(out_dir, "beom_ref")
        cmd = [FLANG, opt, "-ffp-contract=off"] + os.environ.get("BEOM_FLANG_EXTRA", "").split()
        if openmp:
            cmd.append("-fopenmp")
        cmd += ["shared_mod.f95", "private_mod.f95", os.path.join(REF, "main.f95"), "-o", exe]
        r = subprocess.run(cmd, cwd=scratch, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("flang failed:\n" + r.stdout + r.stderr)
        with open(os.path.join(out_dir, "params.json"), "w") as f:
            json.dump(dict(params=params.to_json(), variant=variant, openmp=openmp, opt=opt), f, indent=1)
        return exe
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def run(exe: str, cwd: str, dump_upto: int = -1, dump_every: int = 0, t_from: int = -1,
        t_to: int = -1, threads: int | None = None, timeout: int = 3600) -> str:
    """Runs the reference binary in ``cwd`` (idir/odir must be './')."""
    with open(os.path.join(cwd, "oracle_ctl.txt"), "w") as f:
        f.write("%d %d %d %d\n" % (dump_upto, dump_every, t_from, t_to))
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    env.setdefault("OMP_STACKSIZE", "1G")
    cmd = "ulimit -s unlimited 2>/dev/null || ulimit -s $(ulimit -H -s) 2>/dev/null; exec %s" % os.path.abspath(exe)
    r = subprocess.run(["bash", "-c", cmd], cwd=cwd, capture_output=True, text=True, env=env,
                       timeout=timeout)
    if r.returncode != 0 or "ERROR CODE" in r.stderr:
        raise RuntimeError("reference run failed (%d):\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-2000:]))
    return r.stdout


if __name__ == "__main__":
    from beom_amd.params import Params
    ap = argparse.ArgumentParser()
    ap.add_argument("params_json")
    ap.add_argument("out_dir")
    ap.add_argument("--variant", default="private_mod.f95")
    ap.add_argument("--openmp", action="store_true")
    ap.add_argument("--opt", default="-O2")
    a = ap.parse_args()
    with open(a.params_json) as f:
        p = Params.from_json(json.load(f))
    print(build(p, a.out_dir, a.variant, a.openmp, a.opt))
