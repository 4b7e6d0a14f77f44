"""Multi-GPU: 1-D j-slab decomposition with deep ghost rows.

The reference has no distributed mode (OpenMP only).  Packing is j-major
(private_mod.f95:588-602), so rows j0..j1 of a dense frame are one contiguous ipnt range
per layer and a slab is itself a dense frame.  One process per GPU (torchrun); rank r owns
global rows own0..own1 and keeps G ghost rows on each side that touches a neighbour.

Deep-halo scheme: every sweep of a time step runs on the whole local window (owned +
ghost rows).  Results in the outermost ghost rows are wrong (their N or S neighbour is
outside the window and reads the land sentinel) and the error front moves inwards by at
most one row per dependent sweep:
    rebuild_fluxes (steps 1-3) 1 row S | update_h 1 row N | update_mont 1 N + 1 S |
    update_viscosity 1 N + 1 S | first momentum sweep 1 | second momentum sweep 1
=> at most 4 rows per side per step, so G = 4 keeps every OWNED row bit-identical to the
single-domain run, and ONE exchange per step of the five prognostic fields
(hlay, u, v, h_u, h_v; histories are recomputed consistently in the ghosts) replaces the
five per-sweep exchanges of a one-row halo.  Redundant work: 2G rows per rank.

The exchange is point-to-point only (each rank talks to <= 2 neighbours, 2 of the 7 xGMI
links), via torch.distributed batch_isend_irecv: backend "nccl" = RCCL on GPUs, "gloo" in
the CPU tests.  ~2.6 MB per direction per step at 4096 columns x 4 layers — latency-bound.

Domains periodic in y are supported on one GPU only (the wrap would need the exchange to
close the ring and the orphan row mm+1 to be special-cased); x periodicity is local.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

from .grid import Fields
from .params import Params

GHOST = 4
EXCHANGED = ("hlay", "u", "v", "h_u", "h_v")


@dataclasses.dataclass
class SlabGeom:
    rank: int
    world: int
    L: int              # columns = lm + 1
    Mg: int             # global rows = mm + 1
    own0: int           # first owned global row (1-based)
    own1: int           # last owned global row (inclusive)
    win0: int           # first global row of the local window
    win1: int           # last global row of the local window

    @property
    def rows(self): return self.win1 - self.win0 + 1
    @property
    def row0(self): return self.win0 - 1           # beom_params.slab_row0
    @property
    def ghost_s(self): return self.own0 - self.win0
    @property
    def ghost_n(self): return self.win1 - self.own1

    def local_rows(self, g0: int, g1: int) -> Tuple[int, int]:
        """Packed index range [a, b) of global rows g0..g1 (inclusive) inside the window."""
        return 1 + (g0 - self.win0) * self.L, 1 + (g1 - self.win0 + 1) * self.L


def decompose(mm: int, lm: int, world: int, ghost: int = GHOST) -> List[SlabGeom]:
    """Equal row counts (dense frames: equal work); remainders go to the first ranks."""
    Mg, L = mm + 1, lm + 1
    if world < 1 or Mg < world * (ghost + 1):
        raise ValueError("too few rows (%d) for %d slabs with %d ghost rows" % (Mg, world, ghost))
    base, rem = divmod(Mg, world)
    out, j = [], 1
    for r in range(world):
        n = base + (1 if r < rem else 0)
        own0, own1 = j, j + n - 1
        j += n
        win0 = own0 - ghost if r > 0 else own0
        win1 = own1 + ghost if r < world - 1 else own1
        out.append(SlabGeom(r, world, L, Mg, own0, own1, win0, win1))
    return out


def slice_fields(f: Fields, g: SlabGeom) -> Fields:
    """Local Fields of a slab: the window's rows of every packed array, sentinel first,
    neighbours re-indexed locally (0 outside the window)."""
    p = f.p
    if p.ndeg != g.L * g.Mg:
        raise ValueError("slab decomposition needs a dense frame (ndeg = (lm+1)(mm+1))")
    if float(p.yper) > 0.5 and g.world > 1:
        raise NotImplementedError("y-periodic domains run on one GPU only (see beom_amd/slab.py)")
    a, b = 1 + (g.win0 - 1) * g.L, 1 + g.win1 * g.L          # global packed range [a, b)
    n_loc = b - a

    def cut(x):                                              # last axis = packed index
        z = np.zeros(x.shape[:-1] + (n_loc + 1,), dtype=x.dtype)
        z[..., 0] = x[..., 0]
        z[..., 1:] = x[..., a:b]
        return np.ascontiguousarray(z)

    def cut_hist(x):                                         # [nlay, n1, K]
        z = np.zeros((x.shape[0], n_loc + 1, x.shape[2]), dtype=x.dtype)
        z[:, 1:, :] = x[:, a:b, :]
        return z

    neig = np.zeros((n_loc + 1, 8), dtype=np.int32)
    gl = f.neig[a:b].astype(np.int64)
    inside = (gl >= a) & (gl < b)
    neig[1:] = np.where(inside, gl - a + 1, 0).astype(np.int32)
    tide = np.zeros((3, n_loc + 1, 1, 2), dtype=np.float64)
    tide[:, 1:] = f.tide[:, a:b]
    lp = dataclasses.replace(p) if False else Params.from_json(p.to_json())
    lp.mm = g.rows - 1
    lp.ndeg = n_loc
    kw = {}
    for k in ("mk_u", "mk_v", "mk_n", "mkpe", "mkpi", "fcor", "h_th", "h_to", "h_0", "hlay", "u", "v",
              "h_u", "h_v", "v_cc", "v_ll", "tt3d", "tb3d", "tu3d", "taus", "fnud", "nudg", "hdot", "subc"):
        kw[k] = cut(getattr(f, k))
    for k in ("rs_h", "dmdx", "dmdy"):
        kw[k] = cut_hist(getattr(f, k))
    kw["fcor"][0] = f.fcor[0]
    return Fields(p=lp, neig=neig, posc=f.posc[a - 1:b - 1].copy(), tide=tide, w_ti=f.w_ti.copy(),
                  bodf=f.bodf.copy(), invf=f.invf, flag_nudging=f.flag_nudging, has=dict(f.has), **kw)


class SlabRunner:
    """Steps one slab and exchanges ghost rows with its neighbours once per time step.

    `engine` needs: step(tstp_first, nsteps, sync=False), field_tensors() -> {name:
    tensor[nlay, n_local+1]} (views of the live state), sync(); optional profile_start/
    profile_stop, pack_rows/unpack_rows (one-launch packing) and step_phase (split step).
    beom_amd.capi.Engine provides them on a GPU; tests plug in a CPU adapter over the oracle
    to exercise exactly this decomposition / exchange logic under gloo.

    Overlap (GPU, `overlap=True`): the exchange of step n runs on a second HIP stream while
    the main stream already computes the interior rows of step n+1 (beom_step_phase 1); the
    rows next to the ghost zones follow once the ghosts have landed (phase 2)."""

    def __init__(self, engine, geom: SlabGeom, nlay: int, dist=None, overlap: bool = False):
        import torch
        self.torch = torch
        self.engine = engine
        self.g = geom
        self.nlay = nlay
        self.dist = dist
        ref = engine.field_tensors(EXCHANGED)["hlay"]
        G = GHOST
        n = len(EXCHANGED) * nlay * G * geom.L
        mk = lambda: torch.empty(n, dtype=ref.dtype, device=ref.device)
        self.has_s = geom.rank > 0
        self.has_n = geom.rank < geom.world - 1
        self.send_s, self.recv_s = (mk(), mk()) if self.has_s else (None, None)
        self.send_n, self.recv_n = (mk(), mk()) if self.has_n else (None, None)
        # local row numbers (1-based): what I send = my outermost OWNED rows; what I receive = ghosts
        loc = lambda g0: g0 - geom.win0 + 1
        if self.has_s:
            self.j_send_s, self.j_recv_s = loc(geom.own0), loc(geom.win0)
        if self.has_n:
            self.j_send_n, self.j_recv_n = loc(geom.own1 - G + 1), loc(geom.own1 + 1)
        self.fast_pack = hasattr(engine, "pack_rows")
        self.can_overlap = hasattr(engine, "step_phase") and ref.is_cuda and geom.world > 1
        self.overlap = bool(overlap) and self.can_overlap
        self._pending = None
        self.main = self.comm = None
        if ref.is_cuda and hasattr(engine, "set_stream"):
            # the engine, the packing and the exchange share ONE explicit stream (`main`); the
            # overlapped form adds `comm` for the transfer + unpack
            self.main = torch.cuda.Stream(device=ref.device)
            self.comm = torch.cuda.Stream(device=ref.device)
            self.main.wait_stream(torch.cuda.current_stream(ref.device))
            engine.set_stream(self.main.cuda_stream)

    # -- construction helpers ---------------------------------------------------------
    @classmethod
    def from_global_case(cls, p: Params, files: Dict[str, np.ndarray], rank: int, world: int,
                         device: int = 0, variant: int = 0, overlap: bool = True):
        """GPU path used by bench.py: build the global state on the host, keep this rank's
        window, create the HIP engine on `device` and run it on a torch stream."""
        import torch
        import torch.distributed as dist
        from . import capi
        from .grid import read_input_data
        f = read_input_data(p, files=files)
        geom = decompose(p.mm, p.lm, world)[rank]
        lf = slice_fields(f, geom)
        del f
        eng = capi.Engine(lf, device=device, variant=variant, slab_row0=geom.row0, slab_mm=p.mm)
        run = cls(eng, geom, p.nlay, dist=dist, overlap=overlap)
        run.local_fields = lf
        return run

    def reset_state(self):
        """Back to the initial state of this slab (bench: verification and timed runs start equal)."""
        from .capi import STATE_NAMES
        self.finish()
        self.engine.sync()
        self.engine.upload(**{k: getattr(self.local_fields, k) for k in STATE_NAMES})
        self._pending = None

    def _backend(self) -> str:
        try:
            return str(self.dist.get_backend()) if self.dist is not None and self.dist.is_initialized() else "none"
        except Exception:
            return "unknown"

    def describe(self) -> dict:
        g = self.g
        return {"ghost_rows": GHOST, "exchanges_per_step": 1, "fields": list(EXCHANGED),
                "rows_owned": g.own1 - g.own0 + 1, "rows_local": g.rows,
                "bytes_per_direction_per_step": len(EXCHANGED) * self.nlay * GHOST * g.L * 8,
                "backend": "torch.distributed P2P (%s)" % ("RCCL" if self._backend() == "nccl" else self._backend()),
                "overlap_with_interior": self.overlap}

    # -- packing ------------------------------------------------------------------------
    @property
    def t(self):
        # the engine may ping-pong its buffers between steps: always ask for the live views
        return self.engine.field_tensors(EXCHANGED)

    def _rows(self, j):
        return 1 + (j - 1) * self.g.L, 1 + (j - 1 + GHOST) * self.g.L

    def _pack(self, buf, j):
        if self.fast_pack:
            self.engine.pack_rows(j, GHOST, buf)
            return
        a, b = self._rows(j)
        t, k, m = self.t, 0, (b - a) * self.nlay
        for name in EXCHANGED:
            buf[k:k + m].view(self.nlay, b - a).copy_(t[name][:, a:b])
            k += m

    def _unpack(self, buf, j):
        if self.fast_pack:
            self.engine.unpack_rows(j, GHOST, buf)
            return
        a, b = self._rows(j)
        t, k, m = self.t, 0, (b - a) * self.nlay
        for name in EXCHANGED:
            t[name][:, a:b].copy_(buf[k:k + m].view(self.nlay, b - a))
            k += m

    def pack_all(self):
        if self.has_s:
            self._pack(self.send_s, self.j_send_s)
        if self.has_n:
            self._pack(self.send_n, self.j_send_n)

    def unpack_all(self):
        if self.has_s:
            self._unpack(self.recv_s, self.j_recv_s)
        if self.has_n:
            self._unpack(self.recv_n, self.j_recv_n)

    def _p2p(self):
        dist, torch = self.dist, self.torch
        buf = self.recv_s if self.has_s else self.recv_n
        # RCCL orders the transfer after the CURRENT stream's work and wait() orders the current
        # stream after the transfer.  gloo (one-GPU rehearsals) stages CUDA tensors through the
        # host outside any stream order: fence the device on both sides of it.
        fence = buf.is_cuda and dist.get_backend() != "nccl"
        if fence:
            torch.cuda.synchronize()
        ops = []
        # same order of peers on both sides of a link: lower neighbour first
        if self.has_s:
            ops.append(dist.P2POp(dist.irecv, self.recv_s, self.g.rank - 1))
            ops.append(dist.P2POp(dist.isend, self.send_s, self.g.rank - 1))
        if self.has_n:
            ops.append(dist.P2POp(dist.isend, self.send_n, self.g.rank + 1))
            ops.append(dist.P2POp(dist.irecv, self.recv_n, self.g.rank + 1))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if fence:
            torch.cuda.synchronize()

    def exchange(self):
        """Blocking form (in stream order): pack, send/recv, unpack."""
        if self.g.world == 1:
            return
        self.pack_all()
        self._p2p()
        self.unpack_all()

    # -- overlapped form ------------------------------------------------------------------
    def _begin_pack(self):
        """After a step: pack the outermost owned rows on the main stream."""
        self.pack_all()
        self._packed = self.torch.cuda.Event()
        self._packed.record(self.main)

    def _begin_transfer(self, transfer=None):
        """Send/recv + unpack on the comm stream; `transfer` replaces the P2P (tests)."""
        torch = self.torch
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self._packed)
            (transfer or self._p2p)()
            self.engine.set_stream(self.comm.cuda_stream)
            self.unpack_all()
            self.engine.set_stream(self.main.cuda_stream)
            done = torch.cuda.Event()
            done.record(self.comm)
        self._pending = done

    def _exchange_begin(self):
        self._begin_pack()
        self._begin_transfer()

    def _exchange_end(self):
        if self._pending is not None:
            self.main.wait_event(self._pending)
            self._pending = None

    # -- stepping -----------------------------------------------------------------------
    def step(self, tstp_first: int, nsteps: int):
        torch = self.torch
        if not self.overlap:
            import contextlib
            ctx = torch.cuda.stream(self.main) if self.main is not None else contextlib.nullcontext()
            with ctx:
                for t in range(tstp_first, tstp_first + nsteps):
                    self.engine.step(t, 1, sync=False)
                    self.exchange()
            return
        with torch.cuda.stream(self.main):
            for t in range(tstp_first, tstp_first + nsteps):
                if self._pending is not None and self.engine.step_phase(t, 1):
                    self._exchange_end()                 # ghosts of the previous step have landed
                    self.engine.step_phase(t, 2)
                else:
                    self._exchange_end()
                    self.engine.step(t, 1, sync=False)
                self._exchange_begin()

    def finish(self):
        """Drain the exchange in flight (before reading ghost rows or leaving the timed region)."""
        if self.main is not None:
            self._exchange_end()
            self.main.synchronize()

    def profile_steps(self, tstp_first: int, nsteps: int):
        self.engine.profile_start()
        self.step(tstp_first, nsteps)
        self.finish()
        return self.engine.profile_stop()

    def sync(self):
        self.finish()
        self.engine.sync()

    def owned(self, arr: np.ndarray) -> np.ndarray:
        """Owned rows of a local [.., n_local+1] array (for gathering / comparison)."""
        a, b = self.g.local_rows(self.g.own0, self.g.own1)
        return arr[..., a:b]
