"""Mirror of the reference's compile-time parameter block.

The reference configures a run by editing Fortran ``parameter`` declarations in
``shared_mod.f95:41-77`` (user block) from which ``shared_mod.f95:83-99`` derives
``dt, hsal, hdry, tole, pi, grav, rho0, beta, epsi, gamm, del1, del2``.  All of
them are ``real(rw=r8)`` *initialised from default-real (single precision)
literals* (SURVEY F4), e.g. ``grav = 9.8`` is ``dble(9.8e0)``, not ``9.8d0``.
Bit parity therefore needs the literal text, not a Python float: a `Params`
object stores the literals as strings and evaluates them with Fortran's rules.

``format_like_print_params`` restates the number formatting of
``testcases/print_params.m:6-94`` (the text a user pastes into shared_mod.f95),
so that a testcase recipe here yields the same literals a reference user gets.
"""
from __future__ import annotations

import dataclasses
import re
from typing import Dict, List, Sequence

import numpy as np

_INT_RE = re.compile(r"^[+-]?\d+$")


def lit(text: str | float | int) -> np.float64:
    """Value of a Fortran numeric literal assigned to a ``real(r8)`` parameter.

    integer literal  -> exact;  default-real literal -> rounded to r4 first;
    literal with ``_rw``/``_r8``/``d`` exponent -> double.
    """
    if isinstance(text, (int, np.integer)):
        return np.float64(text)
    if isinstance(text, (float, np.floating)):
        return np.float64(np.float32(text))
    t = text.strip().lower()
    if _INT_RE.match(t):
        return np.float64(int(t))
    if t.endswith("_rw") or t.endswith("_r8"):
        return np.float64(float(t[:-3]))
    if "d" in t:
        return np.float64(float(t.replace("d", "e")))
    if t.endswith("_r4"):
        t = t[:-3]
    return np.float64(np.float32(float(t)))


# print_params.m:6-94 — formatting of each value as the user pastes it.
def _fmt_hash0(x: float) -> str:  # '%#0.0f'
    return "%#0.0f" % x


def format_like_print_params(name: str, x) -> str:
    if name in ("lm", "mm", "nlay", "ndeg"):
        return "%d" % int(x)
    if name == "dl":
        if x < 1.0e3:
            return _fmt_hash0(x)
        if (x - np.floor(x / 1.0e3) * 1.0e3) > 1.0:
            return "%g" % (x / 1.0e3) + "e3"
        return _fmt_hash0(x / 1.0e3) + "e3"
    if name == "cext":
        return "%0.1f" % x
    if name == "f0":
        if abs(x) < 1.0e-4:
            return "%e" % x
        if abs(x) > 1.001e-4:
            return "%0.3f" % (x / 1.0e-4) + "e-4"
        return _fmt_hash0(x / 1.0e-4) + "e-4"
    if name == "rhon":
        r = np.atleast_1d(np.asarray(x, dtype=np.float64))
        if np.all((r * 1.0e3 - np.floor(r) * 1.0e3) < 1.0):
            return "(/" + ",".join(_fmt_hash0(v) for v in r) + "/)"
        return "(/" + ",".join("%0.3f" % v for v in r) + "/)"
    if name == "topl":
        r = np.atleast_1d(np.asarray(x, dtype=np.float64))
        return "(/" + ",".join("%f" % v for v in r) + "/)"
    if name in ("dt_s", "dt_o", "hmin"):
        return "%#f" % x
    if name in ("dt_r", "dt3d", "bvis"):
        return ("%#f" % x) if x > 0.0 else _fmt_hash0(x)
    if name == "dvis":
        return ("%#0.3f" % x) if x > 0.0 else _fmt_hash0(x)
    if name == "bdrg":
        return ("%e" % x) if x > 0.0 else _fmt_hash0(x)
    if name in ("hsbl", "hbbl", "g_fb", "uadv", "qdrg", "ocrp", "rsta", "xper",
                "yper", "diag"):
        return _fmt_hash0(x)
    raise KeyError(name)


_REAL_SCALARS = ("dl", "cext", "f0", "dt_s", "dt_o", "dt_r", "dt3d", "bvis",
                 "dvis", "bdrg", "hmin", "hsbl", "hbbl", "g_fb", "uadv", "qdrg",
                 "ocrp", "rsta", "xper", "yper", "diag", "rgld", "mcbc",
                 # fork additions the root shared_mod.f95 forgot to declare
                 "svis", "tdrg", "topt")


@dataclasses.dataclass
class Params:
    """One configuration = the user block of shared_mod.f95:41-77, as literals."""
    lm: int
    mm: int
    nlay: int
    ndeg: int
    lits: Dict[str, str]                 # real scalars, literal text
    rhon: List[str]
    topl: List[str]
    tauw: Sequence[str] = ("0.00", "0.0")
    idir: str = "./"
    odir: str = "./"
    desc: str = "beom_amd"

    def __post_init__(self):
        defaults = dict(rgld="0.", mcbc="1.", svis="0.", tdrg="0.", topt="0.",
                        dt_r="0.", dt3d="0.", bvis="0.", rsta="0.", xper="0.",
                        yper="0.", diag="0.", f0="0.")
        for k, v in defaults.items():
            self.lits.setdefault(k, v)
        missing = [k for k in _REAL_SCALARS if k not in self.lits]
        if missing:
            raise ValueError("missing parameters: %s" % missing)
        if len(self.rhon) != self.nlay or len(self.topl) != self.nlay:
            raise ValueError("rhon/topl must have nlay entries")

    # -- values with Fortran literal semantics ------------------------------
    def __getattr__(self, name):
        lits = self.__dict__.get("lits", {})
        if name in lits:
            return lit(lits[name])
        raise AttributeError(name)

    @property
    def rhon_v(self) -> np.ndarray:
        return np.array([lit(x) for x in self.rhon], dtype=np.float64)

    @property
    def topl_v(self) -> np.ndarray:
        return np.array([lit(x) for x in self.topl], dtype=np.float64)

    @property
    def tauw_v(self):
        return lit(self.tauw[0]), lit(self.tauw[1])

    # -- derived constants, shared_mod.f95:83-99 ----------------------------
    @property
    def dt(self):    return np.float64(0.5) * self.dl / self.cext          # :84
    @property
    def hsal(self):  return np.float64(10.0) * self.hmin                   # :85
    @property
    def hdry(self):  return lit("1.e-3")                                   # :86
    @property
    def tole(self):  return lit("1.e-6")                                   # :87
    @property
    def pi(self):    return lit("3.1415927")                               # :88
    @property
    def grav(self):  return lit("9.8")                                     # :89
    @property
    def rho0(self):  return self.rhon_v[self.nlay - 1]                     # :90
    @property
    def beta(self):  return lit("0.281105")                                # :91
    @property
    def epsi(self):  return lit("0.013")                                   # :92
    @property
    def gamm(self):  return lit("0.088")                                   # :93
    @property
    def del1(self):                                                        # :94
        return np.float64(0.5) + self.gamm + np.float64(2.0) * self.epsi
    @property
    def del2(self):                                                        # :95
        return np.float64(1.0) - self.del1 - self.gamm - self.epsi
    @property
    def sor(self):   return lit("1.9")                                     # :99
    itmx = 99999                                                           # :104
    nsal = 4                                                               # :105

    # -- step-count arithmetic of integrate_time, private_mod.f95:1853-1856 --
    @property
    def dtd8(self):
        return np.float64(self.dt) / np.float64(24.0) / np.float64(3600.0)

    @staticmethod
    def _nint(x):
        return int(np.floor(abs(x) + 0.5) * np.sign(x))

    @property
    def nstp(self): return self._nint(self.dt_s / self.dtd8)
    @property
    def notp(self): return max(self._nint(self.dt_o / self.dtd8), 1)
    @property
    def n_3d(self): return max(self._nint(self.dt3d / self.dtd8), 1)

    # -- the text of the user block ------------------------------------------
    def fortran_block(self) -> str:
        """Declarations with the public names of shared_mod.f95:41-77 (+F2 fix)."""
        L = self.lits
        rl = ",".join(self.rhon)
        tl = ",".join(self.topl)
        lines = [
            "  integer,     parameter, public :: &",
            "    lm   = %d, mm   = %d, nlay = %d, ndeg = %d" % (self.lm, self.mm, self.nlay, self.ndeg),
            "  real( rw ),  parameter, public :: &",
            "    dl = %s, cext = %s, f0 = %s, &" % (L["dl"], L["cext"], L["f0"]),
            "    rhon(nlay) = (/%s/), &" % rl,
            "    topl(nlay) = (/%s/), &" % tl,
        ]
        rest = [k for k in _REAL_SCALARS if k not in ("dl", "cext", "f0")]
        for i, k in enumerate(rest):
            lines.append("    %-4s = %s%s" % (k, L[k], ", &" if i < len(rest) - 1 else ""))
        lines += [
            "  complex(rw), parameter, public :: &",
            "    tauw = (%s, %s)" % (self.tauw[0], self.tauw[1]),
            "  character(len = sstr), public :: &",
            "    idir = '%s', &" % self.idir,
            "    odir = '%s', &" % self.odir,
            "    desc = '%s'" % self.desc,
        ]
        return "\n".join(lines) + "\n"

    def to_json(self) -> dict:
        d = dataclasses.asdict(self)
        d["tauw"] = list(self.tauw)
        return d

    @classmethod
    def from_json(cls, d: dict) -> "Params":
        return cls(**d)

    def replace(self, **kw) -> "Params":
        d = self.to_json()
        lits = dict(d["lits"])
        for k, v in kw.items():
            if k in d and k != "lits":
                d[k] = v
            else:
                lits[k] = v
        d["lits"] = lits
        return Params.from_json(d)


def make_params(lm, mm, nlay, ndeg, dl, cext, f0, rhon, topl, dt_s, dt_o, dt_r,
                dt3d, bvis, dvis, bdrg, hmin, hsbl, hbbl, g_fb, uadv, qdrg,
                ocrp, rsta, xper, yper, diag, tauw=(0.0, 0.0), idir="./",
                odir="./", desc="beom_amd", **extra) -> Params:
    """Same argument list as testcases/print_params.m:1-4; numbers are turned
    into the literals that script would print."""
    f = format_like_print_params
    vals = dict(dl=dl, cext=cext, f0=f0, dt_s=dt_s, dt_o=dt_o, dt_r=dt_r,
                dt3d=dt3d, bvis=bvis, dvis=dvis, bdrg=bdrg, hmin=hmin,
                hsbl=hsbl, hbbl=hbbl, g_fb=g_fb, uadv=uadv, qdrg=qdrg,
                ocrp=ocrp, rsta=rsta, xper=xper, yper=yper, diag=diag)
    lits = {k: f(k, float(v)) for k, v in vals.items()}
    for k, v in extra.items():
        lits[k] = v if isinstance(v, str) else repr(float(v))
    rl = f("rhon", rhon)[2:-2].split(",")
    tl = f("topl", np.atleast_1d(topl))[2:-2].split(",")
    tw = ["%#0.2f" % tauw[0], "%#0.2f" % tauw[1]]
    return Params(lm=int(lm), mm=int(mm), nlay=int(nlay), ndeg=int(ndeg),
                  lits=lits, rhon=rl, topl=tl, tauw=tw, idir=idir, odir=odir,
                  desc=desc)
