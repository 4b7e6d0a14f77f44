"""ctypes binding of the C-ABI declared in include/beom_hip.h (libbeom_hip.so).

This is the Python host's FFI stub — the same calls the Fortran host makes through
iso_c_binding (beom_amd/host/beom_cabi.f95).  There is no fallback: if the HIP
library is missing or no GPU is usable, `load()`/`Engine()` raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .grid import Fields
from .params import Params

BEOM_MAX_LAYERS = 16
BEOM_ABI_VERSION = 2
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEOM_HIP_LIB", os.path.join(_HERE, "csrc", "libbeom_hip.so"))


class BeomParams(C.Structure):
    """struct beom_params of include/beom_hip.h."""
    _fields_ = (
        [(n, C.c_int32) for n in ("abi_version", "lm", "mm", "nlay", "ndeg", "nsal", "variant",
                                  "flag_nudging", "dense_hint", "slab_row0", "slab_mm")]
        + [(n, C.c_double) for n in ("dl", "dt", "grav", "rho0", "beta", "epsi", "gamm", "del1",
                                     "del2", "hmin", "hsal", "bvis", "dvis", "svis", "bdrg",
                                     "tdrg", "qdrg", "hsbl", "hbbl", "g_fb", "uadv", "ocrp",
                                     "rgld", "mcbc", "invf", "w_ti")]
        + [("rhon", C.c_double * BEOM_MAX_LAYERS)]
    )


STATICS_NAMES = ("fcor", "h_th", "h_to", "nudg", "fnud", "hdot", "tide", "bodf", "taus")


class BeomStatics(C.Structure):
    """struct beom_statics of include/beom_hip.h."""
    _fields_ = [(n, C.POINTER(C.c_double)) for n in STATICS_NAMES]


class BeomState(C.Structure):
    """struct beom_state of include/beom_hip.h."""
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy",
                                                     "v_cc", "v_ll", "tt3d", "tb3d", "tu3d")]


XCHG_PEER, XCHG_RCCL, XCHG_SHM, XCHG_RING1, XCHG_LOOPBACK = 0, 1, 2, 0x100, 0x200


def make_params_struct(p: Params, f: Optional[Fields] = None, variant: int = 0,
                       dense_hint: int = 1, slab_row0: int = 0, slab_mm: int = 0) -> BeomParams:
    s = BeomParams()
    s.abi_version = BEOM_ABI_VERSION
    s.lm, s.mm, s.nlay, s.ndeg = p.lm, p.mm, p.nlay, p.ndeg
    s.nsal = p.nsal
    s.variant = variant
    s.flag_nudging = int(bool(f.flag_nudging)) if f is not None else 0
    s.dense_hint = dense_hint
    s.slab_row0, s.slab_mm = slab_row0, slab_mm
    for n in ("dl", "dt", "grav", "rho0", "beta", "epsi", "gamm", "del1", "del2", "hmin", "hsal",
              "bvis", "dvis", "svis", "bdrg", "tdrg", "qdrg", "hsbl", "hbbl", "g_fb", "uadv",
              "ocrp", "rgld", "mcbc"):
        setattr(s, n, float(getattr(p, n)))
    s.invf = float(f.invf) if f is not None else 0.0
    s.w_ti = float(f.w_ti[0]) if f is not None else 0.0
    if p.nlay > BEOM_MAX_LAYERS:
        raise ValueError("nlay > BEOM_MAX_LAYERS")
    for i, r in enumerate(p.rhon_v):
        s.rhon[i] = float(r)
    return s


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


_lib = None
ERRLEN = 999                       # lstr of shared_mod.f95:34


def source_hash() -> str:
    """The hash the Makefile embeds: sha1 of the library's sources in the Makefile's order."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha1()
    for fn in ("beom_engine.hip", "beom_multi.hip", "beom_dev.h", "beom_kernels.h", "beom_launch_tiled.h", "beom_dense_host.h",
               os.path.join("..", "..", "include", "beom_hip.h")):
        with open(os.path.join(csrc, fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load(path: Optional[str] = None) -> C.CDLL:
    """Loads libbeom_hip.so and declares every prototype of include/beom_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, and if this
    # library pulls in /opt/rocm's copy first, torch later finds "No HIP GPUs".  Let torch
    # (when present) load its runtime first; libbeom_hip.so then binds to the same one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError("HIP engine library %s not built (run __graft_entry__.build()); "
                           "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    if path == LIB_PATH and os.environ.get("BEOM_HIP_LIB") is None:      # in-tree library: must match the in-tree sources
        lib.beom_source_hash.restype = C.c_char_p
        built, tree = lib.beom_source_hash().decode(), source_hash()
        if built != tree:
            raise RuntimeError("libbeom_hip.so was built from other sources (%s, tree %s): run "
                               "__graft_entry__.build()" % (built, tree))
    dpp, ipp, cp, ci, cd = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_char_p, C.c_int, C.c_double
    H = C.c_void_p
    lib.beom_abi_version.restype = ci
    lib.beom_device_count.argtypes = [cp, ci]
    lib.beom_device_pci_bus_id.argtypes = [ci, cp, ci]
    lib.beom_device_pci_bus_id.restype = ci
    lib.beom_create.argtypes = [C.POINTER(BeomParams), ci, ipp, ipp] + [dpp] * 14 + [C.POINTER(H), cp, ci]
    lib.beom_destroy.argtypes = [H]
    lib.beom_set_rigid_lid.argtypes = [H, dpp, dpp, dpp, dpp, cp, ci]
    lib.beom_download_pressure.argtypes = [H, dpp, cp, ci]
    lib.beom_upload_state.argtypes = [H] + [dpp] * 13 + [cp, ci]
    lib.beom_download_state.argtypes = [H] + [dpp] * 13 + [cp, ci]
    lib.beom_download_scratch.argtypes = [H] + [dpp] * 6 + [cp, ci]
    lib.beom_step.argtypes = [H, ci, ci, cd, cd, cd, cd, ci, cp, ci]
    lib.beom_sync.argtypes = [H, cp, ci]
    lib.beom_set_stream.argtypes = [H, C.c_void_p, ci]
    lib.beom_set_option.argtypes = [H, cp, ci]
    lib.beom_set_open_boundaries.argtypes = [H, ci, ipp, cp, ci]
    lib.beom_multi_set_open_boundaries.argtypes = [H, ci, ipp, cp, ci]
    lib.beom_download_outputs.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dpp, C.POINTER(ci), cp, ci]
    lib.beom_download_diag.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, cp, ci]
    lib.beom_download_diag.restype = ci
    lib.beom_step_phase.argtypes = [H, ci, cd, cd, cd, cd, ci, ci, cp, ci]
    lib.beom_pack_rows.argtypes = [H, ci, ci, C.c_void_p]
    lib.beom_unpack_rows.argtypes = [H, ci, ci, C.c_void_p]
    lib.beom_pack_rows2.argtypes = [H, ci, ci, C.c_void_p, ci, C.c_void_p]
    lib.beom_unpack_rows2.argtypes = [H, ci, ci, C.c_void_p, ci, C.c_void_p]
    lib.beom_profile_start.argtypes = [H]
    lib.beom_profile_stop.argtypes = [H, dpp, C.POINTER(ci), cp, ci]
    lib.beom_update_h.argtypes = [H, cd, cd, cd]
    lib.beom_update_mont_rvor_pvor_dive_kine.argtypes = [H, ci]
    lib.beom_update_viscosity.argtypes = [H, ci]
    lib.beom_update_u.argtypes = [H, ci, cd, cd, cd]
    lib.beom_update_v.argtypes = [H, ci, cd, cd, cd]
    lib.beom_rebuild_fluxes.argtypes = [H]
    lib.beom_distribute_stress.argtypes = [H]
    lib.beom_info.argtypes = [H, cp]
    lib.beom_info.restype = ci
    lib.beom_device_field.argtypes = [H, cp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.beom_is_dense.argtypes = [H]
    lib.beom_profile_steps.argtypes = [H, ci, ci, cd, cd, cd, cd, ci, dpp, C.POINTER(ci), cp, ci]
    MH = C.c_void_p
    lib.beom_multi_create.argtypes = [C.POINTER(BeomParams), ci, C.POINTER(ci), ipp, ipp] + [dpp] * 14 + [C.POINTER(MH), cp, ci]
    lib.beom_multi_destroy.argtypes = [MH]
    lib.beom_multi_count.argtypes = [MH]
    lib.beom_multi_band.argtypes = [MH, ci] + [C.POINTER(ci)] * 5
    lib.beom_multi_upload_state.argtypes = [MH] + [dpp] * 13 + [cp, ci]
    lib.beom_multi_download_state.argtypes = [MH] + [dpp] * 13 + [cp, ci]
    lib.beom_multi_step.argtypes = [MH, ci, ci, cd, cd, cd, cd, ci, cp, ci]
    lib.beom_multi_sync.argtypes = [MH, cp, ci]
    lib.beom_multi_stats.argtypes = [MH, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.beom_multi_create_ex.argtypes = [C.POINTER(BeomParams), ci, C.POINTER(ci), ci, ipp, ipp] + [dpp] * 14 + [C.POINTER(MH), cp, ci]
    lib.beom_multi_download_outputs.argtypes = [MH, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dpp, C.POINTER(ci), cp, ci]
    lib.beom_multi_download_diag.argtypes = [MH, C.c_void_p, C.c_void_p, C.c_void_p, cp, ci]
    lib.beom_multi_describe.argtypes = [MH] + [C.POINTER(ci)] * 5
    lib.beom_multi_set_option.argtypes = [MH, cp, ci]
    lib.beom_multi_engine.argtypes = [MH, ci, C.POINTER(H)]
    lib.beom_multi_profile_start.argtypes = [MH]
    lib.beom_multi_profile_stop.argtypes = [MH, dpp, C.POINTER(ci), cp, ci]
    lib.beom_rccl_unique_id.argtypes = [C.c_void_p, cp, ci]
    lib.beom_rccl_version.argtypes = [cp, ci]
    lib.beom_multi_window.argtypes = [C.POINTER(BeomParams), ci, ci, ci] + [C.POINTER(ci)] * 4
    lib.beom_multi_create_local.argtypes = [C.POINTER(BeomParams), ci, ci, ci, ci, ci, C.c_void_p,
                                            C.POINTER(BeomStatics), C.POINTER(BeomStatics), C.POINTER(MH), cp, ci]
    lib.beom_multi_create_local_ex.argtypes = [C.POINTER(BeomParams), ci, ci, ci, ci, ci, ci, C.c_void_p,
                                               C.POINTER(BeomStatics), C.POINTER(BeomStatics), C.POINTER(MH), cp, ci]
    lib.beom_multi_set_open_boundaries_local.argtypes = [MH, ci, C.POINTER(ci), ci, C.POINTER(ci), cp, ci]
    lib.beom_multi_upload_local.argtypes = [MH, C.POINTER(BeomState), C.POINTER(BeomState), cp, ci]
    lib.beom_multi_download_local.argtypes = [MH, C.POINTER(BeomState), C.POINTER(BeomState), cp, ci]
    for name in ("beom_multi_create", "beom_multi_destroy", "beom_multi_count", "beom_multi_band",
                 "beom_multi_upload_state", "beom_multi_download_state", "beom_multi_step", "beom_multi_sync",
                 "beom_multi_stats", "beom_multi_create_ex", "beom_multi_describe", "beom_multi_engine",
                 "beom_multi_download_outputs", "beom_multi_download_diag", "beom_multi_set_open_boundaries",
                 "beom_multi_set_option", "beom_multi_profile_start", "beom_multi_profile_stop", "beom_rccl_unique_id", "beom_rccl_version",
                 "beom_multi_window", "beom_multi_create_local", "beom_multi_create_local_ex", "beom_multi_set_open_boundaries_local",
                 "beom_multi_upload_local", "beom_multi_download_local"):
        getattr(lib, name).restype = ci
    for name in ("beom_device_count", "beom_create", "beom_destroy", "beom_upload_state",
                 "beom_download_state", "beom_download_scratch", "beom_step", "beom_sync",
                 "beom_update_h", "beom_update_mont_rvor_pvor_dive_kine", "beom_update_viscosity",
                 "beom_update_u", "beom_update_v", "beom_rebuild_fluxes", "beom_distribute_stress",
                 "beom_device_field", "beom_is_dense", "beom_profile_steps", "beom_set_stream",
                 "beom_profile_start", "beom_profile_stop", "beom_set_option", "beom_step_phase",
                 "beom_pack_rows", "beom_unpack_rows", "beom_pack_rows2", "beom_unpack_rows2", "beom_download_outputs", "beom_set_open_boundaries"):
        getattr(lib, name).restype = ci
    if lib.beom_abi_version() != BEOM_ABI_VERSION:
        raise RuntimeError("ABI mismatch")
    _lib = lib
    return lib


EXPORTS = ("beom_abi_version", "beom_device_count", "beom_device_pci_bus_id", "beom_info", "beom_download_diag", "beom_create", "beom_destroy", "beom_set_rigid_lid", "beom_download_pressure",
           "beom_upload_state", "beom_download_state", "beom_download_scratch", "beom_step",
           "beom_sync", "beom_update_h", "beom_update_mont_rvor_pvor_dive_kine",
           "beom_update_viscosity", "beom_update_u", "beom_update_v", "beom_rebuild_fluxes",
           "beom_distribute_stress", "beom_device_field", "beom_is_dense", "beom_profile_steps",
           "beom_set_stream", "beom_profile_start", "beom_profile_stop", "beom_set_option",
           "beom_step_phase", "beom_pack_rows", "beom_unpack_rows", "beom_pack_rows2", "beom_unpack_rows2", "beom_download_outputs",
           "beom_set_open_boundaries",
           "beom_multi_create", "beom_multi_destroy", "beom_multi_count", "beom_multi_band",
           "beom_multi_upload_state", "beom_multi_download_state", "beom_multi_step", "beom_multi_sync",
           "beom_multi_stats", "beom_multi_create_ex", "beom_multi_describe", "beom_multi_engine",
           "beom_multi_download_outputs", "beom_multi_download_diag", "beom_multi_set_open_boundaries",
           "beom_multi_set_option", "beom_multi_profile_start", "beom_multi_profile_stop", "beom_rccl_unique_id", "beom_rccl_version",
           "beom_multi_window", "beom_multi_create_local", "beom_multi_create_local_ex", "beom_multi_set_open_boundaries_local",
           "beom_multi_upload_local", "beom_multi_download_local")

STATE_NAMES = ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "v_cc", "v_ll",
               "tt3d", "tb3d", "tu3d")
SCRATCH_NAMES = ("mont", "rvor", "pvor", "dive", "d2hx", "d2hy")


class BeomError(RuntimeError):
    pass


class Engine:
    """One handle = one GPU's copy of the engine state (mirror of the Fortran module)."""

    def __init__(self, f: Fields, device: int = 0, variant: int = 0, dense_hint: int = 1,
                 upload: bool = True, slab_row0: int = 0, slab_mm: int = 0):
        self.lib = load()
        self.f = f
        self.p = f.p
        self.device = device
        self.prm = make_params_struct(f.p, f, variant, dense_hint, slab_row0, slab_mm)
        self._err = C.create_string_buffer(ERRLEN + 1)
        self.h = C.c_void_p()
        opt = lambda k: _dp(getattr(f, k)) if f.has.get(k, True) else None
        rc = self.lib.beom_create(
            C.byref(self.prm), device, _ip(f.neig), _ip(f.subc),
            _dp(f.mk_u), _dp(f.mk_v), _dp(f.mk_n), _dp(f.mkpe), _dp(f.mkpi),
            _dp(f.fcor), _dp(f.h_th), _dp(f.h_to), _dp(f.nudg), _dp(f.fnud),
            opt("hdot"), opt("tide"), opt("bodf"), _dp(f.taus),
            C.byref(self.h), self._err, ERRLEN)
        self._check(rc)
        if f.flag_nudging and float(f.p.mcbc) < 0.5 and f.segm is not None:     # no_gradient_obc (:2613)
            seg = np.ascontiguousarray(f.segm, dtype=np.int32)
            self._check(self.lib.beom_set_open_boundaries(self.h, seg.shape[1], _ip(seg), self._err, ERRLEN))
        if float(f.p.rgld) > 0.5:                                                # rigid lid: the Poisson operators (:505-563)
            self._check(self.lib.beom_set_rigid_lid(self.h, _dp(f.Ow), _dp(f.Os), _dp(f.Osum_), _dp(f.pi_s),
                                                    self._err, ERRLEN))
        if upload:
            self.upload(**{k: getattr(f, k) for k in STATE_NAMES})

    def upload_pressure(self, pi_s):
        """The lid pressure pi_s(0:ndeg) of a rigid-lid handle (rgld = 1)."""
        self._check(self.lib.beom_set_rigid_lid(self.h, None, None, None, _dp(np.ascontiguousarray(pi_s, dtype=np.float64)),
                                                self._err, ERRLEN))

    def download_pressure(self):
        out = np.zeros(self.p.ndeg + 1)
        self._check(self.lib.beom_download_pressure(self.h, _dp(out), self._err, ERRLEN))
        return out

    def _check(self, rc):
        if rc != 0:
            raise BeomError("beom_hip error %d: %s" % (rc, self._err.value.decode(errors="replace")))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.beom_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, **arrays):
        args = [_dp(arrays.get(k)) for k in STATE_NAMES]
        self._check(self.lib.beom_upload_state(self.h, *args, self._err, ERRLEN))

    def download(self, names=STATE_NAMES) -> dict:
        f = self.f
        out = {k: np.zeros_like(getattr(f, k)) for k in names}
        args = [_dp(out.get(k)) for k in STATE_NAMES]
        self._check(self.lib.beom_download_state(self.h, *args, self._err, ERRLEN))
        return out

    def download_scratch(self) -> dict:
        n1 = self.p.ndeg + 1
        out = {k: np.zeros((self.p.nlay, n1)) for k in SCRATCH_NAMES}
        self._check(self.lib.beom_download_scratch(self.h, *[_dp(out[k]) for k in SCRATCH_NAMES],
                                                   self._err, ERRLEN))
        return out

    def step(self, tstp_first: int, nsteps: int, tres: Optional[float] = None, sync: bool = True):
        p = self.p
        tres = float(getattr(self.f, "tres", 0.0)) if tres is None else tres      # a restarted run continues from its record's time
        self._check(self.lib.beom_step(self.h, tstp_first, nsteps, tres, float(p.dtd8),
                                       float(p.dt_r), float(p.rsta), p.n_3d, self._err, ERRLEN))
        if sync:
            self.sync()

    def sync(self):
        self._check(self.lib.beom_sync(self.h, self._err, ERRLEN))

    def set_stream(self, hip_stream: Optional[int]):
        """hip_stream: integer handle (torch.cuda.current_stream().cuda_stream; 0 = the default
        stream) — or None to go back to the handle's own stream."""
        if hip_stream is None:
            self._check(self.lib.beom_set_stream(self.h, None, 1))
        else:
            self._check(self.lib.beom_set_stream(self.h, C.c_void_p(hip_stream), 0))

    def step_phase(self, tstp: int, phase: int, tres: Optional[float] = None) -> bool:
        """Split step (beom_step_phase).  False if not available for this step."""
        p = self.p
        tres = float(getattr(self.f, "tres", 0.0)) if tres is None else tres
        rc = self.lib.beom_step_phase(self.h, tstp, tres, float(p.dtd8), float(p.dt_r),
                                      float(p.rsta), p.n_3d, phase, self._err, ERRLEN)
        if rc == -20:
            return False
        self._check(rc)
        return True

    def pack_rows(self, jlo: int, nrows: int, tensor):
        self._check(self.lib.beom_pack_rows(self.h, jlo, nrows, C.c_void_p(tensor.data_ptr())))

    def unpack_rows(self, jlo: int, nrows: int, tensor):
        self._check(self.lib.beom_unpack_rows(self.h, jlo, nrows, C.c_void_p(tensor.data_ptr())))

    def download_outputs(self, h0r4: Optional[np.ndarray]):
        """(eta, u, v) real*4 records [nlay, ndeg], minmax [nlay, 6], thin_layer — formed on the device."""
        n = (self.p.nlay, self.p.ndeg)
        eta, u4, v4 = (np.zeros(n, dtype=np.float32) for _ in range(3))
        mm = np.zeros((self.p.nlay, 6))
        thin = C.c_int(0)
        hp = h0r4.ctypes.data_as(C.c_void_p) if h0r4 is not None else None
        self._check(self.lib.beom_download_outputs(self.h, hp, eta.ctypes.data_as(C.c_void_p),
                                                   u4.ctypes.data_as(C.c_void_p), v4.ctypes.data_as(C.c_void_p),
                                                   _dp(mm), C.byref(thin), self._err, ERRLEN))
        return eta, u4, v4, mm, thin.value

    def download_diag(self):
        """(pvor, mont, v_cc) real*4 records [nlay, ndeg] of write_array's `diag` branch, formed on the device."""
        n = (self.p.nlay, self.p.ndeg)
        out = [np.zeros(n, dtype=np.float32) for _ in range(3)]
        self._check(self.lib.beom_download_diag(self.h, *[a.ctypes.data_as(C.c_void_p) for a in out], self._err, ERRLEN))
        return tuple(out)

    def set_option(self, name: str, value: int):
        self._check(self.lib.beom_set_option(self.h, name.encode(), int(value)))

    def profile_start(self):
        self._check(self.lib.beom_profile_start(self.h))

    def profile_stop(self):
        ms = (C.c_double * 8)()
        nl = (C.c_int * 8)()
        self._check(self.lib.beom_profile_stop(self.h, ms, nl, self._err, ERRLEN))
        return list(ms)[:8], list(nl)[:8]

    def info(self, what: str) -> int:
        """beom_info: "stress_folded" (the last step formed its stress inside the momentum sweep: tt3d, tb3d, tu3d are
        then not kept current), "tile_rows"."""
        v = self.lib.beom_info(self.h, what.encode())
        if v < 0:
            raise BeomError("beom_info(%s) = %d" % (what, v))
        return v

    def field_tensors(self, names=("hlay", "u", "v", "h_u", "h_v")):
        """Zero-copy torch views [nlay, layer stride] of the device-resident prognostic fields.  Handles on the
        table path keep the packed layout (layer stride ndeg+1); dense handles use a padded row pitch — cell
        (i, j) at i + (j-1)*pitch (`self.row_pitch`) — so move rows with pack_rows/unpack_rows, not by slicing."""
        import torch
        out = {}
        cache = self.__dict__.setdefault("_tensor_cache", {})
        for k in names:
            ptr, sl, sr, r0 = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
            self._check(self.lib.beom_device_field(self.h, k.encode(), C.byref(ptr), C.byref(sl),
                                                   C.byref(sr), C.byref(r0)))
            n1 = sl.value
            self.row_pitch = sr.value
            t = cache.get(ptr.value)          # the fused sweeps ping-pong buffers: key by address
            if t is None:
                class _Iface:
                    __cuda_array_interface__ = {"shape": (self.p.nlay, n1), "typestr": "<f8",
                                                "data": (ptr.value, False), "version": 3}
                t = cache[ptr.value] = torch.as_tensor(_Iface(), device=torch.device("cuda", self.device))
            out[k] = t
        return out

    def profile_steps(self, tstp_first: int, nsteps: int, tres: Optional[float] = None):
        p = self.p
        tres = float(getattr(self.f, "tres", 0.0)) if tres is None else tres
        ms = (C.c_double * 8)()
        nl = (C.c_int * 8)()
        self._check(self.lib.beom_profile_steps(self.h, tstp_first, nsteps, tres, float(p.dtd8),
                                                float(p.dt_r), float(p.rsta), p.n_3d, ms, nl,
                                                self._err, ERRLEN))
        return list(ms)[:8], list(nl)[:8]

    @property
    def is_dense(self) -> bool:
        """The tiled fast path is active (frames without land, or with land embedded in the rectangle)."""
        return bool(self.lib.beom_is_dense(self.h))

    @property
    def is_embedded(self) -> bool:
        """A frame with land running on the rectangle (masks from arrays in the tiles that touch land)."""
        return self.lib.beom_is_dense(self.h) == 2

    # per-sweep entry points (parity tests)
    def update_h(self, gene, ramp, ctim): self._check(self.lib.beom_update_h(self.h, gene, ramp, ctim))
    def update_mont(self, ilay=0): self._check(self.lib.beom_update_mont_rvor_pvor_dive_kine(self.h, ilay))
    def update_viscosity(self, ilay=0): self._check(self.lib.beom_update_viscosity(self.h, ilay))
    def update_u(self, ilay, gene, ramp, ctim): self._check(self.lib.beom_update_u(self.h, ilay, gene, ramp, ctim))
    def update_v(self, ilay, gene, ramp, ctim): self._check(self.lib.beom_update_v(self.h, ilay, gene, ramp, ctim))
    def rebuild_fluxes(self): self._check(self.lib.beom_rebuild_fluxes(self.h))
    def distribute_stress(self): self._check(self.lib.beom_distribute_stress(self.h))


class MultiEngine:
    """beom_multi_*: the whole frame on several HIP devices from ONE process (row bands with ghost
    exchange inside the library) — what the Fortran host uses with BEOM_NGPU > 1.  `devices` may
    name a device more than once (tests: three bands on the one GPU of the box)."""

    def __init__(self, f: Fields, devices=(0,), variant: int = 0, upload: bool = True,
                 transport: int = XCHG_PEER, ring1: bool = False):
        self.lib = load()
        self.f, self.p = f, f.p
        self.prm = make_params_struct(f.p, f, variant, 1, 0, 0)
        self._err = C.create_string_buffer(ERRLEN + 1)
        self.h = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        opt = lambda k: _dp(getattr(f, k)) if f.has.get(k, True) else None
        rc = self.lib.beom_multi_create_ex(
            C.byref(self.prm), len(devices), dev, transport | (XCHG_RING1 if ring1 else 0), _ip(f.neig), _ip(f.subc),
            _dp(f.mk_u), _dp(f.mk_v), _dp(f.mk_n), _dp(f.mkpe), _dp(f.mkpi),
            _dp(f.fcor), _dp(f.h_th), _dp(f.h_to), _dp(f.nudg), _dp(f.fnud),
            opt("hdot"), opt("tide"), opt("bodf"), _dp(f.taus),
            C.byref(self.h), self._err, ERRLEN)
        self._check(rc)
        if f.flag_nudging and float(f.p.mcbc) < 0.5 and f.segm is not None:     # no_gradient_obc (:2613), dealt to the bands
            seg = np.ascontiguousarray(f.segm, dtype=np.int32)
            self._check(self.lib.beom_multi_set_open_boundaries(self.h, seg.shape[1], _ip(seg), self._err, ERRLEN))
        if upload:
            self.upload(**{k: getattr(f, k) for k in STATE_NAMES})

    _check = Engine._check

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.beom_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def count(self) -> int:
        return self.lib.beom_multi_count(self.h)

    def band(self, k: int) -> dict:
        v = [C.c_int() for _ in range(5)]
        self._check(self.lib.beom_multi_band(self.h, k, *[C.byref(x) for x in v]))
        return dict(zip(("own0", "own1", "win0", "win1", "device"), (x.value for x in v)))

    def upload(self, **arrays):
        args = [_dp(arrays.get(k)) for k in STATE_NAMES]
        self._check(self.lib.beom_multi_upload_state(self.h, *args, self._err, ERRLEN))

    def download(self, names=STATE_NAMES) -> dict:
        out = {k: np.zeros_like(getattr(self.f, k)) for k in names}
        args = [_dp(out.get(k)) for k in STATE_NAMES]
        self._check(self.lib.beom_multi_download_state(self.h, *args, self._err, ERRLEN))
        return out

    def step(self, tstp_first: int, nsteps: int, tres: Optional[float] = None, sync: bool = True):
        p = self.p
        tres = float(getattr(self.f, "tres", 0.0)) if tres is None else tres      # a restarted run continues from its record's time
        self._check(self.lib.beom_multi_step(self.h, tstp_first, nsteps, tres, float(p.dtd8), float(p.dt_r),
                                             float(p.rsta), p.n_3d, self._err, ERRLEN))
        if sync:
            self.sync()

    def sync(self):
        self._check(self.lib.beom_multi_sync(self.h, self._err, ERRLEN))

    def stats(self) -> dict:
        a, b = C.c_longlong(), C.c_longlong()
        self.lib.beom_multi_stats(self.h, C.byref(a), C.byref(b))
        return {"split": a.value, "plain": b.value}

    def download_outputs(self, h0r4: np.ndarray):
        """(eta, u, v) real*4 records [nlay, ndeg], minmax [nlay, 6], thin_layer — every band forms its rows on its device."""
        n = (self.p.nlay, self.p.ndeg)
        eta, u4, v4 = (np.zeros(n, dtype=np.float32) for _ in range(3))
        mm = np.zeros((self.p.nlay, 6))
        thin = C.c_int(0)
        self._check(self.lib.beom_multi_download_outputs(self.h, h0r4.ctypes.data_as(C.c_void_p), eta.ctypes.data_as(C.c_void_p),
                                                         u4.ctypes.data_as(C.c_void_p), v4.ctypes.data_as(C.c_void_p),
                                                         _dp(mm), C.byref(thin), self._err, ERRLEN))
        return eta, u4, v4, mm, thin.value

    def download_diag(self):
        n = (self.p.nlay, self.p.ndeg)
        out = [np.zeros(n, dtype=np.float32) for _ in range(3)]
        self._check(self.lib.beom_multi_download_diag(self.h, *[a.ctypes.data_as(C.c_void_p) for a in out], self._err, ERRLEN))
        return tuple(out)

    def describe(self) -> dict:
        v = [C.c_int() for _ in range(5)]
        self._check(self.lib.beom_multi_describe(self.h, *[C.byref(x) for x in v]))
        d = dict(zip(("bands_total", "bands_local", "transport", "ring", "rccl_version"), (x.value for x in v)))
        d["transport"] = {XCHG_PEER: "hipMemcpyPeerAsync", XCHG_RCCL: "RCCL ncclSend/ncclRecv",
                          XCHG_SHM: "POSIX shared memory (host-staged)"}.get(d["transport"], "?")
        return d

    def profile_start(self):
        self._check(self.lib.beom_multi_profile_start(self.h))

    def profile_stop(self):
        ms = (C.c_double * 8)()
        nl = (C.c_int * 8)()
        self._check(self.lib.beom_multi_profile_stop(self.h, ms, nl, self._err, ERRLEN))
        return list(ms)[:8], list(nl)[:8]

    def profile_steps(self, tstp_first: int, nsteps: int, tres: Optional[float] = None):
        self.profile_start()
        self.step(tstp_first, nsteps, tres, sync=False)
        return self.profile_stop()

    def set_option(self, name: str, value: int):
        """"overlap": split steps around the exchange in flight; other names go to every band."""
        self._check(self.lib.beom_multi_set_option(self.h, name.encode(), int(value)))

    def info(self, what: str) -> int:
        """beom_info of band 0's handle (all bands of a frame run the same launches)."""
        v = self.lib.beom_info(self.band_engine_handle(0), what.encode())
        if v < 0:
            raise BeomError("beom_info(%s) = %d" % (what, v))
        return v

    def band_engine_handle(self, k: int) -> C.c_void_p:
        h = C.c_void_p()
        self._check(self.lib.beom_multi_engine(self.h, k, C.byref(h)))
        return h


def device_pci_bus_id(device: int) -> Optional[str]:
    buf = C.create_string_buffer(64)
    return buf.value.decode().lower() if load().beom_device_pci_bus_id(device, buf, 64) == 0 else None


def rccl_unique_id() -> bytes:
    """128 bytes of ncclGetUniqueId (one rank calls it, every rank of the job gets the bytes)."""
    lib = load()
    buf = C.create_string_buffer(128)
    err = C.create_string_buffer(ERRLEN + 1)
    rc = lib.beom_rccl_unique_id(buf, err, ERRLEN)
    if rc != 0:
        raise BeomError("beom_rccl_unique_id %d: %s" % (rc, err.value.decode(errors="replace")))
    return buf.raw


def multi_window(p: Params, nb: int, band: int, yper: bool) -> dict:
    """Rows of band `band` of `nb`: owned global rows own0..own1 and the ghost rows on either side."""
    lib = load()
    prm = make_params_struct(p)
    v = [C.c_int() for _ in range(4)]
    rc = lib.beom_multi_window(C.byref(prm), nb, band, int(bool(yper)), *[C.byref(x) for x in v])
    if rc != 0:
        raise BeomError("beom_multi_window %d" % rc)
    return dict(zip(("own0", "own1", "ghost_s", "ghost_n"), (x.value for x in v)))


class BandEngine(MultiEngine):
    """beom_multi_create_local[_ex]: ONE band of a frame cut over `nb` processes, built from that band's window
    only (bench.py under torchrun; exchange over RCCL).  `f` = the window's Fields (rows: south ghosts,
    owned rows, north ghosts — beom_amd.slab.build_window), `p_global` the global frame's parameters,
    `orphan` = Fields of row mm+1 (band 0 of a frame periodic in y).  `shm_name` ("/...", the same on every
    rank) selects the shared-memory transport instead of RCCL: ranks of one node that may share a device;
    `loopback`: the band receives its own sends (one band alone with the whole exchange machinery: timing)."""

    def __init__(self, f: Fields, p_global: Params, nb: int, band: int, device: int = 0, variant: int = 0,
                 rccl_id: Optional[bytes] = None, orphan: Optional[Fields] = None, upload: bool = True,
                 shm_name: Optional[str] = None, loopback: bool = False):
        self.lib = load()
        self.f, self.p, self.pg = f, f.p, p_global
        self.orphan = orphan
        self.prm = make_params_struct(p_global, f, variant, 1, 0, 0)
        self._err = C.create_string_buffer(ERRLEN + 1)
        self.h = C.c_void_p()
        if shm_name is not None:
            idbuf, transport = C.create_string_buffer(shm_name.encode()), XCHG_SHM
        else:
            idbuf, transport = (C.create_string_buffer(rccl_id, 128) if rccl_id is not None else None), XCHG_RCCL
        st = self._statics(f)
        so = self._statics(orphan) if orphan is not None else None
        rc = self.lib.beom_multi_create_local_ex(
            C.byref(self.prm), nb, band, device, int(float(p_global.xper) > 0.5), int(float(p_global.yper) > 0.5),
            transport | (XCHG_LOOPBACK if loopback else 0), idbuf, C.byref(st), C.byref(so) if so is not None else None,
            C.byref(self.h), self._err, ERRLEN)
        self._check(rc)
        if bool(f.flag_nudging) and float(p_global.mcbc) < 0.5:       # no_gradient_obc (:2613): the segments of this window's rows
            seg = lambda x: np.ascontiguousarray(x.segm, dtype=np.int32) if (x is not None and x.segm is not None) else None
            sw, so_ = seg(f), seg(orphan)
            self._check(self.lib.beom_multi_set_open_boundaries_local(
                self.h, 0 if sw is None else sw.shape[1], None if sw is None else _ip(sw),
                0 if so_ is None else so_.shape[1], None if so_ is None else _ip(so_), self._err, ERRLEN))
        if upload:
            self.upload()

    @staticmethod
    def _statics(f: Fields) -> BeomStatics:
        s = BeomStatics()
        for k in STATICS_NAMES:
            present = f.has.get(k, True) if k in ("hdot", "tide", "bodf") else True
            setattr(s, k, _dp(getattr(f, k)) if present else None)
        return s

    @staticmethod
    def _state(arrays: dict) -> BeomState:
        s = BeomState()
        for k in STATE_NAMES:
            setattr(s, k, _dp(arrays.get(k)))
        return s

    def upload(self, **arrays):
        src = arrays or {k: getattr(self.f, k) for k in STATE_NAMES}
        sw = self._state(src)
        so = self._state({k: getattr(self.orphan, k) for k in STATE_NAMES}) if self.orphan is not None else None
        self._check(self.lib.beom_multi_upload_local(self.h, C.byref(sw), C.byref(so) if so is not None else None,
                                                     self._err, ERRLEN))

    def download(self, names=STATE_NAMES, orphan: bool = False) -> dict:
        out = {k: np.zeros_like(getattr(self.f, k)) for k in names}
        sw = self._state(out)
        oo, so = None, None
        if orphan and self.orphan is not None:
            oo = {k: np.zeros_like(getattr(self.orphan, k)) for k in names}
            so = self._state(oo)
        self._check(self.lib.beom_multi_download_local(self.h, C.byref(sw), C.byref(so) if so is not None else None,
                                                       self._err, ERRLEN))
        return (out, oo) if orphan else out
