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

Frames periodic in y (private_mod.f95:642-668): rows 1..mm are dealt out as a RING of bands (the
first band's south ghosts are the last band's top owned rows); a band of a ring is presented to
the engine as a slab deep inside a taller frame, so every row mask is 1 and nothing wraps
locally.  Row mm+1 keeps its slot in every array although nothing points to it; its own
neighbours are cells of rows 1 and mm, so it is carried by a small companion frame next to band 0
(`mini_rows`).  x periodicity is local to every band.

This module is the host-side description of the decomposition (geometry, window builds from a
recipe without any array of global size, the CPU/gloo rehearsal of the exchange).  On GPUs the
time loop itself runs inside the library (beom_multi_step, beom_amd/csrc/beom_multi.hip; RCCL
or peer copies); `SlabRunner` drives the same scheme from Python for the CPU tests.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

from .grid import Fields
from .params import Params

GHOST = 4
MINI_LO = 6                     # rows 1..6 of a y-periodic frame in the companion frame (beom_multi.hip: kMiniLo)
FAKE_PAD = 8                    # a band of a ring is presented as rows 9.. of a frame 16 rows taller (kFakePad)
EXCHANGED = ("hlay", "u", "v", "h_u", "h_v")


@dataclasses.dataclass
class SlabGeom:
    rank: int
    world: int
    L: int              # columns = lm + 1
    Mg: int             # global rows = mm + 1
    own0: int           # first owned global row (1-based)
    own1: int           # last owned global row (inclusive)
    win0: int           # first global row of the local window (<= 0 in a ring: wrapped ghost rows)
    win1: int           # last global row of the local window (> mm in a ring: wrapped ghost rows)
    ring: bool = False  # frame periodic in y: the bands form a ring over rows 1..mm

    @property
    def rows(self): return self.win1 - self.win0 + 1
    @property
    def row0(self): return self.win0 - 1           # beom_params.slab_row0 (chain)
    @property
    def ghost_s(self): return self.own0 - self.win0
    @property
    def ghost_n(self): return self.win1 - self.own1
    @property
    def south(self): return (self.rank - 1) % self.world if (self.ring or self.rank > 0) else None
    @property
    def north(self): return (self.rank + 1) % self.world if (self.ring or self.rank < self.world - 1) else None

    def global_rows(self) -> List[int]:
        """Global row of every local row (ghost rows of a ring wrap over rows 1..mm)."""
        mr = self.Mg - 1
        return [((g - 1) % mr) + 1 if self.ring else g for g in range(self.win0, self.win1 + 1)]

    def local_rows(self, g0: int, g1: int) -> Tuple[int, int]:
        """Packed index range [a, b) of rows g0..g1 (inclusive, window coordinates) inside the window."""
        return 1 + (g0 - self.win0) * self.L, 1 + (g1 - self.win0 + 1) * self.L

    def pieces(self) -> List[Tuple[int, int]]:
        """The window as runs of consecutive global rows, south to north."""
        out, rows = [], self.global_rows()
        a = rows[0]
        for k in range(1, len(rows) + 1):
            if k == len(rows) or rows[k] != rows[k - 1] + 1:
                out.append((a, rows[k - 1]))
                if k < len(rows):
                    a = rows[k]
        return out


def decompose(mm: int, lm: int, world: int, ghost: int = GHOST, yper: bool = False) -> List[SlabGeom]:
    """Equal row counts (dense frames: equal work); remainders go to the first ranks.  Frames periodic
    in y: rows 1..mm are dealt out as a ring (every band has ghosts on both sides; row mm+1, which
    nothing points to, is carried by the companion frame — mini_rows)."""
    Mg, L = mm + 1, lm + 1
    nrows = mm if yper else Mg
    if world < 1 or nrows < world * (max(ghost, MINI_LO) if yper else ghost + 1):
        raise ValueError("too few rows (%d) for %d slabs with %d ghost rows" % (nrows, world, ghost))
    base, rem = divmod(nrows, world)
    out, j = [], 1
    for r in range(world):
        n = base + (1 if r < rem else 0)
        own0, own1 = j, j + n - 1
        j += n
        win0 = own0 - ghost if (r > 0 or yper) else own0
        win1 = own1 + ghost if (r < world - 1 or yper) else own1
        out.append(SlabGeom(r, world, L, Mg, own0, own1, win0, win1, ring=bool(yper)))
    return out


def mini_rows(mm: int) -> List[int]:
    """Rows of the companion frame of a ring: 1..6, mm-3..mm and the orphan row mm+1, as ONE small
    y-periodic frame (the orphan's E/W neighbours are cells of row 1, its S neighbours cells of row mm,
    private_mod.f95:642-668)."""
    return list(range(1, MINI_LO + 1)) + list(range(mm - GHOST + 1, mm + 1)) + [mm + 1]


def dense_tables(L: int, M: int, joff: int, Mg: int, slab: bool, xper: bool, yper: bool) -> Dict[str, np.ndarray]:
    """Connectivity and masks of a dense frame in closed form (numpy twin of
    beom_amd/csrc/beom_dense_host.h): local rows 1..M are global rows joff+1..joff+M of an Mg-row frame."""
    n1 = L * M + 1
    i = np.tile(np.arange(1, L + 1), M)
    j = np.repeat(np.arange(1, M + 1), L)
    jg = j + joff
    ywrap = bool(yper) and not slab

    def wrap_at(a, b):
        # both wraps act on the TARGET coordinate, evaluated on the unwrapped values
        a2 = a.copy(); b2 = b.copy()
        if xper:
            a2 = np.where(a == 0, L - 1, np.where(a == L, 1, a))
        if ywrap:
            b2 = np.where(b == 0, M - 1, np.where(b == M, 1, b))
        ok = (a2 >= 1) & (a2 <= L) & (b2 >= 1) & (b2 <= M)
        return np.where(ok, a2 + (b2 - 1) * L, 0)

    neig = np.zeros((n1, 8), dtype=np.int32)
    for k, (di, dj) in enumerate(((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))):
        neig[1:, k] = wrap_at(i + di, j + dj)
    subc = np.zeros((2, n1), dtype=np.int32)
    subc[0, 1:] = i; subc[1, 1:] = jg
    inn = (i <= L - 1) & (jg <= Mg - 1)
    ux = (i >= 2) | bool(xper)
    vy = (jg >= 2) | bool(yper)
    mk = lambda c: np.concatenate([[0.0], np.where(c, 1.0, 0.0)])
    return dict(neig=neig, subc=subc, mk_n=mk(inn), mk_u=mk(inn & ux), mk_v=mk(inn & vy), mkpe=mk(inn & ux & vy),
                mkpi=np.concatenate([[0.0], np.ones(L * M)]))


_PER_CELL = ("fcor", "h_th", "h_to", "h_0", "hlay", "u", "v", "h_u", "h_v", "v_cc", "v_ll", "tt3d", "tb3d", "tu3d",
             "taus", "fnud", "nudg", "hdot")


def _take_rows(x: np.ndarray, rows: List[int], L: int, axis: int = -1) -> np.ndarray:
    """[.., 0:n1, ..] -> sentinel + the listed (1-based) rows of L cells each, along `axis`."""
    idx = np.concatenate([[0]] + [np.arange(1 + (r - 1) * L, 1 + r * L) for r in rows])
    return np.ascontiguousarray(np.take(x, idx, axis=axis))


def _assemble(f: Fields, rows: List[int], L: int, tables: Dict[str, np.ndarray], mm_local: int) -> Fields:
    """Fields of the frame made of the listed rows of `f` (any order), with the given tables."""
    p = f.p
    lp = Params.from_json(p.to_json())
    lp.mm = mm_local
    lp.ndeg = L * len(rows)
    kw = {k: _take_rows(getattr(f, k), rows, L) for k in _PER_CELL}
    for k in ("rs_h", "dmdx", "dmdy"):
        kw[k] = _take_rows(getattr(f, k), rows, L, axis=1)
    tide = _take_rows(f.tide, rows, L, axis=1)
    posc = np.concatenate([f.posc[(r - 1) * L:r * L] for r in rows]).astype(np.int32)
    return Fields(p=lp, posc=posc, tide=tide, w_ti=f.w_ti.copy(), bodf=f.bodf.copy(), invf=f.invf,
                  flag_nudging=f.flag_nudging, has=dict(f.has), **tables, **kw)


def _band_tables(g: SlabGeom, xper: bool) -> Dict[str, np.ndarray]:
    if g.ring:        # deep inside a taller fake frame: every row mask is 1, no wrap of its own
        return dense_tables(g.L, g.rows, FAKE_PAD, g.rows + 2 * FAKE_PAD, True, xper, False)
    if g.world == 1:
        return dense_tables(g.L, g.rows, 0, g.Mg, False, xper, False)
    return dense_tables(g.L, g.rows, g.row0, g.Mg, True, xper, False)


def _check_dense(f: Fields, L: int, Mg: int):
    p = f.p
    if p.ndeg != L * Mg:
        raise ValueError("slab decomposition needs a dense frame (ndeg = (lm+1)(mm+1))")
    t = dense_tables(L, Mg, 0, Mg, False, float(p.xper) > 0.5, float(p.yper) > 0.5)
    for k in ("neig", "mk_n", "mk_u", "mk_v", "mkpe"):
        if not np.array_equal(getattr(f, k), t[k]):
            raise ValueError("slab decomposition needs a dense frame (interior entirely wet): %s differs" % k)


def slice_fields(f: Fields, g: SlabGeom) -> Fields:
    """Local Fields of a slab cut from the whole frame's: the window's rows of every packed array,
    sentinel first; connectivity and masks in closed form (a band is a dense frame)."""
    _check_dense(f, g.L, g.Mg)
    if g.ring != (float(f.p.yper) > 0.5):
        raise ValueError("ring geometry and yper disagree")
    return _assemble(f, g.global_rows(), g.L, _band_tables(g, float(f.p.xper) > 0.5), g.rows - 1)


def slice_mini(f: Fields) -> Fields:
    """The companion frame of a ring cut from the whole frame's Fields (mini_rows)."""
    L, mm = f.p.lm + 1, f.p.mm
    rows = mini_rows(mm)
    M = len(rows)
    return _assemble(f, rows, L, dense_tables(L, M, 0, M, False, float(f.p.xper) > 0.5, True), M - 1)


# ---- windows built from a recipe: nothing of global size on any rank ------------------------------------
def recipe_global_info(recipe, chunk: int = 512) -> dict:
    """What init derives from the WHOLE frame (grid.read_input_data, window=): deepest / shallowest wet
    depth after read_input_file('h_bo') (private_mod.f95:827-839, :132-134), the Coriolis mean fcor(0)
    (:933) and invf (:223-229) — found by walking the recipe in chunks of rows."""
    from .grid import f4, f8
    p = recipe.p
    if "fcor" in recipe.keys or "h_to" in recipe.keys:
        raise NotImplementedError("window builds of recipes with fcor.bin / h_to.bin")
    dmax, dmin, nudged = 0.0, np.inf, False
    for ja in range(0, p.mm + 2, chunk):
        jb = min(ja + chunk - 1, p.mm + 1)
        fr = recipe.rows(ja, jb)
        nudged = nudged or ("nudg" in fr and bool(np.any(np.asarray(fr["nudg"]).astype(f4) > 1e-9)))
        h = np.asarray(fr["h_bo"]).astype(f4).astype(f8)
        h[h < p.hdry] = 0.0
        h[0, :] = 0.0; h[-1, :] = 0.0
        if ja == 0:
            h[:, 0] = 0.0
        if jb == p.mm + 1:
            h[:, -1] = 0.0
        wet = h[h > p.hdry]
        if wet.size:
            dmin = min(dmin, float(wet.min()))
        dmax = max(dmax, float(h.max()))
    fc = np.full(p.ndeg + 1, p.f0, dtype=f8)                      # :301 (no fcor.bin)
    invf = f8(np.add.reduce(fc) / f8(fc.size))
    invf = float(f8(1.0) / invf) if abs(invf) > 1.25e-5 else 0.0
    return dict(dmax=dmax, dmin=dmin if np.isfinite(dmin) else 0.0, fcor0=float(p.f0), invf=invf,
                flag_nudging=nudged)


def build_rows(recipe, glob: dict, j0: int, j1: int) -> Fields:
    """Fields of packed rows j0..j1 of the recipe's frame, from the recipe's rows j0-1..j1 only."""
    from .grid import read_input_data
    p = recipe.p
    lp = Params.from_json(p.to_json())
    lp.mm = j1 - j0
    lp.ndeg = (p.lm + 1) * (j1 - j0 + 1)
    lp.lits["yper"] = "0."                                        # a band never wraps by itself
    win = dict(glob, bottom_margin=(j0 - 1 == 0), top_margin=(j1 == p.mm + 1), yper=p.lits["yper"])
    return read_input_data(lp, files=recipe.rows(j0 - 1, j1), window=win)


def _join_segm(pieces: List[Fields]) -> Optional[np.ndarray]:
    """Open-boundary segments (Fields.segm, int32 [18, nseg]) of pieces stacked south to north: cell indices and row numbers
    shifted by the rows below the piece (segments run along rows: both ends of one lie in the same piece or the next row)."""
    out, cells, rows = [], 0, 0
    for q in pieces:
        if q.segm is not None and q.segm.size:
            t = np.array(q.segm, dtype=np.int64)
            for c in (0, 6, 9, 12, 15):                  # columns 1, 7, 10, 13, 16 of segm(nseg, 18): cell, then its (i, j)
                t[c] = np.where(t[c] > 0, t[c] + cells, t[c])
                t[c + 2] = t[c + 2] + rows
            out.append(t)
        cells += q.p.ndeg
        rows += q.p.mm + 1
    return np.ascontiguousarray(np.concatenate(out, axis=1).astype(np.int32)) if out else None


def _join(pieces: List[Fields]) -> Fields:
    """Rows of several Fields stacked south to north into one (tables are set by the caller)."""
    f0 = pieces[0]
    if len(pieces) == 1:
        return f0
    cat = lambda k, ax: np.ascontiguousarray(np.concatenate(
        [np.take(getattr(q, k), np.arange(0 if n == 0 else 1, getattr(q, k).shape[ax]), axis=ax) for n, q in enumerate(pieces)], axis=ax))
    kw = {k: cat(k, -1) for k in _PER_CELL + ("subc", "mk_u", "mk_v", "mk_n", "mkpe", "mkpi")}
    kw["neig"] = cat("neig", 0)
    for k in ("rs_h", "dmdx", "dmdy", "tide"):
        kw[k] = cat(k, 1)
    lp = Params.from_json(f0.p.to_json())
    lp.mm = sum(q.p.mm + 1 for q in pieces) - 1
    lp.ndeg = sum(q.p.ndeg for q in pieces)
    return Fields(p=lp, posc=np.concatenate([q.posc for q in pieces]), w_ti=f0.w_ti.copy(), bodf=f0.bodf.copy(),
                  invf=f0.invf, flag_nudging=any(q.flag_nudging for q in pieces), has=dict(f0.has), segm=_join_segm(pieces), **kw)


def _with_tables(f: Fields, t: Dict[str, np.ndarray]) -> Fields:
    for k, v in t.items():
        setattr(f, k, v)
    f.hlay = np.ascontiguousarray(f.hlay)        # (hlay = h_0 * mk_n with the piece's own mk_n: the wet mask itself)
    return f


def build_band(recipe, world: int, rank: int, glob: Optional[dict] = None):
    """(window Fields, geometry, orphan-row Fields or None) of band `rank` of `world`, every array from the
    recipe's rows for that band only.  Rows: south ghosts, owned rows, north ghosts (beom_multi_window)."""
    p = recipe.p
    yper, xper = float(p.yper) > 0.5, float(p.xper) > 0.5
    if p.ndeg != (p.lm + 1) * (p.mm + 1):
        raise ValueError("slab decomposition needs a dense frame (ndeg = (lm+1)(mm+1))")
    glob = glob or recipe_global_info(recipe)
    g = decompose(p.mm, p.lm, world, yper=yper)[rank]
    f = _with_tables(_join([build_rows(recipe, glob, a, b) for a, b in g.pieces()]), _band_tables(g, xper))
    f.p.lits["yper"] = p.lits["yper"]
    ii = np.arange(1, g.L + 1)
    f.posc = np.concatenate([ii + 1 + j * (p.lm + 2) for j in g.global_rows()]).astype(np.int32)      # :716
    f.flag_nudging = bool(glob.get("flag_nudging", f.flag_nudging))
    orphan = None
    if yper and rank == 0:
        orphan = build_rows(recipe, glob, p.mm + 1, p.mm + 1)
    return f, g, orphan


def engine_slab_args(g: SlabGeom) -> dict:
    """slab_row0 / slab_mm of beom_params for a band's handle (capi.Engine keywords)."""
    if g.ring:
        return dict(slab_row0=FAKE_PAD, slab_mm=g.rows + 2 * FAKE_PAD - 1)
    if g.world == 1:
        return {}
    return dict(slab_row0=g.row0, slab_mm=g.Mg - 1)


class SlabRunner:
    """Steps one slab and exchanges ghost rows with its neighbours once per time step.

    `engine` needs: step(tstp_first, nsteps, sync=False), field_tensors() -> {name:
    tensor[nlay, n_local+1]} (views of the live state), sync(); optional profile_start/
    profile_stop, pack_rows/unpack_rows (one-launch packing) and step_phase (split step).
    beom_amd.capi.Engine provides them on a GPU; tests plug in a CPU adapter over the oracle
    to exercise exactly this decomposition / exchange logic under gloo.

    Overlap (GPU, `overlap=True`): boundary first — the main stream runs a step up to the momentum
    sweeps on all rows (beom_step_phase 1) and then the momentum sweep on the rows in between
    (part 3); a second HIP stream runs that sweep on the strips next to the ghost zones (part 2),
    packs the outermost owned rows, moves them and unpacks the neighbours' — inside the interior
    sweep.  The main stream waits for its ghost rows once, before the next step."""

    def __init__(self, engine, geom: SlabGeom, nlay: int, dist=None, overlap: bool = False, mini=None):
        import torch
        self.torch = torch
        self.engine = engine
        self.g = geom
        self.nlay = nlay
        self.dist = dist
        self.mini = mini              # companion frame of a ring (rank 0 only): carries the orphan row mm+1
        if geom.ring and geom.rank == 0 and mini is None:
            raise ValueError("rank 0 of a ring needs the companion frame (slice_mini / build_band's orphan)")
        ref = engine.field_tensors(EXCHANGED)["hlay"]
        G = GHOST
        n = len(EXCHANGED) * nlay * G * geom.L
        mk = lambda: torch.empty(n, dtype=ref.dtype, device=ref.device)
        self.has_s = geom.south is not None
        self.has_n = geom.north is not None
        self.send_s, self.recv_s = (mk(), mk()) if self.has_s else (None, None)
        self.send_n, self.recv_n = (mk(), mk()) if self.has_n else (None, None)
        # local row numbers (1-based): what I send = my outermost OWNED rows; what I receive = ghosts
        loc = lambda g0: g0 - geom.win0 + 1
        if self.has_s:
            self.j_send_s, self.j_recv_s = loc(geom.own0), loc(geom.win0)
        if self.has_n:
            self.j_send_n, self.j_recv_n = loc(geom.own1 - G + 1), loc(geom.own1 + 1)
        self.fast_pack = hasattr(engine, "pack_rows")
        self.can_overlap = hasattr(engine, "step_phase") and ref.is_cuda and geom.world > 1 and not geom.ring
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
        geom = decompose(p.mm, p.lm, world, yper=float(p.yper) > 0.5)[rank]
        lf = slice_fields(f, geom)
        mini = capi.Engine(slice_mini(f), device=device, variant=variant) if (geom.ring and rank == 0) else None
        del f
        eng = capi.Engine(lf, device=device, variant=variant, **engine_slab_args(geom))
        run = cls(eng, geom, p.nlay, dist=dist, overlap=overlap, mini=mini)
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
        if self.g.world == 1:         # a ring of one band closes on itself
            self.recv_s.copy_(self.send_n)
            self.recv_n.copy_(self.send_s)
            return
        ops = []
        # tag 0: rows that become the receiver's SOUTH ghosts, tag 1: its north ghosts (in a ring of two
        # bands both neighbours are the same peer); RCCL has no tags and matches a pair's messages in issue
        # order, hence: sends (south, north), receives (north, south) — the order beom_multi.hip uses
        S, N = self.g.south, self.g.north
        tagged = dist.get_backend() != "nccl"
        kw = lambda t: {"tag": t} if tagged else {}
        if self.has_s:
            ops.append(dist.P2POp(dist.isend, self.send_s, S, **kw(1)))
        if self.has_n:
            ops.append(dist.P2POp(dist.isend, self.send_n, N, **kw(0)))
        if self.has_n:
            ops.append(dist.P2POp(dist.irecv, self.recv_n, N, **kw(1)))
        if self.has_s:
            ops.append(dist.P2POp(dist.irecv, self.recv_s, S, **kw(0)))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if fence:
            torch.cuda.synchronize()

    def _mini_step(self, t: int):
        """Companion frame: refresh its rows 1..6 and mm-3..mm from this band (its first owned rows and
        its south ghosts, as they stand BEFORE step t), then its own step t."""
        g, L, G = self.g, self.g.L, GHOST
        if self.fast_pack:            # device handles: the row pitch is the library's business
            if not hasattr(self, "_mini_buf"):
                ref = self.send_s
                self._mini_buf = self.torch.empty(len(EXCHANGED) * self.nlay * MINI_LO * L, dtype=ref.dtype, device=ref.device)
            self.engine.pack_rows(g.ghost_s + 1, MINI_LO, self._mini_buf)
            self.mini.unpack_rows(1, MINI_LO, self._mini_buf)
            self.engine.pack_rows(1, G, self._mini_buf)
            self.mini.unpack_rows(MINI_LO + 1, G, self._mini_buf)
        else:
            mt = self.mini.field_tensors(EXCHANGED)
            bt = self.t
            lo_a, lo_b = 1 + g.ghost_s * L, 1 + (g.ghost_s + MINI_LO) * L
            for k in EXCHANGED:
                mt[k][:, 1:1 + MINI_LO * L].copy_(bt[k][:, lo_a:lo_b])
                mt[k][:, 1 + MINI_LO * L:1 + (MINI_LO + G) * L].copy_(bt[k][:, 1:1 + G * L])
        self.mini.step(t, 1, sync=False)

    def exchange(self):
        """Blocking form (in stream order): pack, send/recv, unpack."""
        if self.g.world == 1 and not self.g.ring:
            return
        self.pack_all()
        self._p2p()
        self.unpack_all()

    # -- overlapped form ------------------------------------------------------------------
    def _begin_pack(self, stream=None):
        """After a step (or after part 2 of a cut step, on the second stream): pack the outermost owned rows."""
        stream = stream or self.main
        with self.torch.cuda.stream(stream):
            self.engine.set_stream(stream.cuda_stream)
            self.pack_all()
            self.engine.set_stream(self.main.cuda_stream)
            self._packed = self.torch.cuda.Event()
            self._packed.record(stream)

    def _advance(self, t: int) -> bool:
        """One step with its rows to send packed (event `_packed`); True if the step was cut boundary first."""
        torch = self.torch
        with torch.cuda.stream(self.main):
            self._exchange_end()                         # the ghost rows of the step before have landed
            if not self.engine.step_phase(t, 1):         # up to the momentum sweeps, all rows
                self.engine.step(t, 1, sync=False)
                self._begin_pack()
                return False
            front = torch.cuda.Event()
            front.record(self.main)
        self.comm.wait_event(front)
        self.engine.set_stream(self.comm.cuda_stream)
        self.engine.step_phase(t, 2)                     # the momentum sweep on the strips next to the ghost zones
        self.engine.set_stream(self.main.cuda_stream)
        self.engine.step_phase(t, 3)                     # ... on the rows in between, then the pointer rotations
        self._begin_pack(self.comm)
        return True

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
                    if self.mini is not None:
                        self._mini_step(t)
                    self.engine.step(t, 1, sync=False)
                    self.exchange()
            return
        for t in range(tstp_first, tstp_first + nsteps):
            self._advance(t)
            self._begin_transfer()

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
