"""TEST INFRASTRUCTURE — ctypes driver of oracle/libbeom_oracle.so (the C restatement
of the reference hot path).  Imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py."""
from __future__ import annotations

import copy
import ctypes as C
import os
import subprocess

import numpy as np

from beom_amd.capi import BeomParams, make_params_struct, STATE_NAMES, SCRATCH_NAMES

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libbeom_oracle.so")

_STATIC = ("neig", "subc", "mk_u", "mk_v", "mk_n", "mkpe", "mkpi", "fcor", "h_th", "h_to",
           "nudg", "fnud", "hdot", "tide", "bodf", "taus")
_SCR_L = tuple(k + "_l" for k in SCRATCH_NAMES)


class OracleState(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in _STATIC] + [(k, C.c_void_p) for k in STATE_NAMES]
                + [(k, C.c_void_p) for k in SCRATCH_NAMES] + [(k, C.c_void_p) for k in _SCR_L]
                + [("segm", C.c_void_p), ("nseg", C.c_int64)]
                + [(k, C.c_void_p) for k in ("delu", "delv", "uu4", "vv4")]
                + [(k, C.c_void_p) for k in ("pi_s", "Ow", "Os", "Osum_")])


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


_lib = None


def host_cores() -> int:
    """CPUs this process may really use: cgroup quota if any, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def load():
    global _lib
    if _lib is None:
        # small fixtures: a big OpenMP team only adds barrier cost (256-way on the GPU box)
        os.environ.setdefault("OMP_NUM_THREADS", str(min(8, host_cores())))
        src = os.path.join(HERE, "beom_oracle.c")
        stale = os.path.exists(LIB) and os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB) + 1.0
        if not os.path.exists(LIB) or stale:
            build()                    # raises if the C restatement does not compile: never run a stale checker
        _lib = C.CDLL(LIB)
        cd, ci = C.c_double, C.c_int
        PP, SP = C.POINTER(BeomParams), C.POINTER(OracleState)
        _lib.oracle_update_h.argtypes = [PP, SP, cd, cd, cd]
        _lib.oracle_update_mont.argtypes = [PP, SP, ci]
        _lib.oracle_update_viscosity.argtypes = [PP, SP, ci]
        _lib.oracle_update_u.argtypes = [PP, SP, ci, cd, cd, cd]
        _lib.oracle_update_v.argtypes = [PP, SP, ci, cd, cd, cd]
        _lib.oracle_rebuild_fluxes.argtypes = [PP, SP]
        _lib.oracle_distribute_stress.argtypes = [PP, SP]
        _lib.oracle_no_gradient_obc.argtypes = [PP, SP, ci]
        for nm in ("oracle_rgld_h_epilogue", "oracle_rgld_upstream_fluxes", "oracle_surf_pressure"):
            getattr(_lib, nm).argtypes = [PP, SP]
        _lib.oracle_step.argtypes = [PP, SP, ci, ci, cd, cd, cd, cd, ci]
        _lib.oracle_step.restype = ci
        assert _lib.oracle_sizeof_params() == C.sizeof(BeomParams)
        assert _lib.oracle_sizeof_state() == C.sizeof(OracleState)
    return _lib


class Oracle:
    """Holds private copies of a Fields object's arrays and steps them on the CPU."""

    def __init__(self, f, variant: int = 0, per_layer_scratch: bool = True):
        self.lib = load()
        self.p = f.p
        self.f = f
        self.prm = make_params_struct(f.p, f, variant)
        n1 = f.p.ndeg + 1
        self.a = {}
        for k in _STATIC:
            self.a[k] = np.ascontiguousarray(getattr(f, k))
        for k in STATE_NAMES:
            self.a[k] = np.array(getattr(f, k), dtype=np.float64, order="C", copy=True)
        for k in SCRATCH_NAMES:
            self.a[k] = np.zeros(n1)
        for k in _SCR_L:
            self.a[k] = np.zeros((f.p.nlay, n1)) if per_layer_scratch else None
        self.st = OracleState()
        self.segm = np.ascontiguousarray(f.segm, dtype=np.int32) if getattr(f, "segm", None) is not None else None
        for k, _ in OracleState._fields_:
            if k in ("segm", "nseg", "delu", "delv", "uu4", "vv4", "pi_s", "Ow", "Os", "Osum_"):
                continue
            arr = self.a[k]
            present = arr is not None and (k not in ("hdot", "tide", "bodf") or f.has.get(k, True))
            setattr(self.st, k, arr.ctypes.data if present else None)

        self.biharm = {k: np.zeros((f.p.nlay, n1)) for k in ("delu", "delv", "uu4", "vv4")}
        for k, a in self.biharm.items():
            setattr(self.st, k, a.ctypes.data)
        self.rgld = {}
        if getattr(f, "pi_s", None) is not None:                     # rigid lid (rgld = 1)
            self.rgld = {"pi_s": np.array(f.pi_s, dtype=np.float64, copy=True),
                         "Ow": np.ascontiguousarray(f.Ow), "Os": np.ascontiguousarray(f.Os), "Osum_": np.ascontiguousarray(f.Osum_)}
            for k, a in self.rgld.items():
                setattr(self.st, k, a.ctypes.data)
        self.st.segm = self.segm.ctypes.data if self.segm is not None else None
        self.st.nseg = self.segm.shape[1] if self.segm is not None else 0

    def step(self, tstp_first: int, nsteps: int, tres=None):
        p = self.p
        tres = float(getattr(self.f, "tres", 0.0)) if tres is None else tres      # restarted runs (rsta = 1, :1311-1325)
        rc = self.lib.oracle_step(C.byref(self.prm), C.byref(self.st), tstp_first, nsteps, tres,
                                  float(p.dtd8), float(p.dt_r), float(p.rsta), p.n_3d)
        if rc != 0:
            raise RuntimeError("oracle_step rc=%d" % rc)

    def update_h(self, gene, ramp, ctim): self.lib.oracle_update_h(C.byref(self.prm), C.byref(self.st), gene, ramp, ctim)
    def update_mont(self, ilay): self.lib.oracle_update_mont(C.byref(self.prm), C.byref(self.st), ilay)
    def update_viscosity(self, ilay): self.lib.oracle_update_viscosity(C.byref(self.prm), C.byref(self.st), ilay)
    def update_u(self, ilay, gene, ramp, ctim): self.lib.oracle_update_u(C.byref(self.prm), C.byref(self.st), ilay, gene, ramp, ctim)
    def update_v(self, ilay, gene, ramp, ctim): self.lib.oracle_update_v(C.byref(self.prm), C.byref(self.st), ilay, gene, ramp, ctim)
    def no_gradient_obc(self, ilay): self.lib.oracle_no_gradient_obc(C.byref(self.prm), C.byref(self.st), ilay)
    def rebuild_fluxes(self): self.lib.oracle_rebuild_fluxes(C.byref(self.prm), C.byref(self.st))
    def rgld_h_epilogue(self): self.lib.oracle_rgld_h_epilogue(C.byref(self.prm), C.byref(self.st))
    def rgld_upstream_fluxes(self): self.lib.oracle_rgld_upstream_fluxes(C.byref(self.prm), C.byref(self.st))
    def surf_pressure(self): self.lib.oracle_surf_pressure(C.byref(self.prm), C.byref(self.st))
    def distribute_stress(self): self.lib.oracle_distribute_stress(C.byref(self.prm), C.byref(self.st))

    def state(self) -> dict:
        return {k: self.a[k] for k in STATE_NAMES}

    def scratch(self) -> dict:
        return {k: self.a[k + "_l"] for k in SCRATCH_NAMES}
