"""Builds the Fortran host executable: <shared_mod> + beom_cabi + beom_host_mod + <main>,
linked against libbeom_hip.so.  `shared_mod_path` / `main_path` default to a generated
shared_mod (gen_shared_mod.py) and the tests' 6-line main; pass the reference's own
files to show the drop-in."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
from typing import Optional

from ..params import Params
from .gen_shared_mod import shared_mod_source

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")

MAIN_SRC = """program main
  use shared_mod,  only: errc, errm, quit
  use private_mod, only: run
  implicit none
  errc = 0
  errm = 'In main,'
  call run()
  call quit()
end program main
"""


def build(p: Optional[Params], out_exe: str, shared_mod_path: Optional[str] = None,
          main_path: Optional[str] = None, opt: str = "-O2", variant: int = 0) -> str:
    """variant = 1 selects the update_h epilogue of private_mod3d.f95 (:1635-1683) — the
    reference picks it by compiling that file instead of private_mod.f95."""
    work = tempfile.mkdtemp(prefix="beom_hostbuild_")
    try:
        host_src = os.path.join(HERE, "beom_host_mod.f95")
        if variant:
            txt = open(host_src).read()
            assert txt.count("prm%variant = 0") == 1
            host_src = os.path.join(work, "beom_host_mod.f95")
            with open(host_src, "w") as f:
                f.write(txt.replace("prm%variant = 0", "prm%variant = " + str(int(variant))))
        if shared_mod_path is None:
            shared_mod_path = os.path.join(work, "shared_mod.f95")
            with open(shared_mod_path, "w") as f:
                f.write(shared_mod_source(p))
        if main_path is None:
            main_path = os.path.join(work, "main.f95")
            with open(main_path, "w") as f:
                f.write(MAIN_SRC)
        out_exe = os.path.abspath(out_exe)
        os.makedirs(os.path.dirname(out_exe), exist_ok=True)
        # the fork's extra switches exist only in a shared_mod.f95 that declares them
        fork = ["-DBEOM_FORK_SWITCHES"] if re.search(r"^[^!]*\bsvis\s*=", open(shared_mod_path).read(), re.M) else []
        cmd = [FLANG, opt, "-cpp", *fork, "-ffp-contract=off", *os.environ.get("BEOM_FLANG_EXTRA", "").split(), shared_mod_path, os.path.join(HERE, "beom_cabi.f95"),
               host_src, main_path, "-L" + CSRC, "-lbeom_hip",
               "-Wl,-rpath," + CSRC, "-o", out_exe]
        r = subprocess.run(cmd, cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("flang failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
        return out_exe
    finally:
        shutil.rmtree(work, ignore_errors=True)
