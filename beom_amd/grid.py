"""Host-side initialisation: the packed field layout of the reference.

Python/numpy mirror of the init half of the engine, used by tests and bench.py to
build what the C-ABI consumes.  The Fortran host (beom_amd/host/private_mod.f95) does
the same work in Fortran for the drop-in build.  Follows, routine by routine:

  read_input_data            private_mod.f95:105-250
  initialize_variables       private_mod.f95:252-307
  get_equilibrium_thickness  private_mod.f95:309-502   (ocrp=1: Newton + SOR)
  index_grid_points          private_mod.f95:567-764
  read_input_file            private_mod.f95:766-967

Memory layout (identical bytes to the Fortran module arrays, private_mod.f95:27-93):
a Fortran ``X(0:ndeg, nlay)`` is a C-ordered numpy ``X[nlay, ndeg+1]``;
``neig(8, 0:ndeg)`` is ``neig[ndeg+1, 8]``; ``rs_h(2,0:ndeg,nlay)`` is
``rs_h[nlay, ndeg+1, 2]``; ``dmdx(3,0:ndeg,nlay)`` is ``[nlay, ndeg+1, 3]``;
``fnud(0:ndeg,nlay,3)`` is ``[3, nlay, ndeg+1]``; ``nudg(0:ndeg,3)`` is
``[3, ndeg+1]``; ``tide(2,1,0:ndeg,3)`` is ``[3, ndeg+1, 1, 2]``;
``tt3d(0:ndeg,2,nlay)`` is ``[nlay, 2, ndeg+1]``; ``bodf(nlay,2)`` is ``[2, nlay]``.
Index 0 of every packed array is the land sentinel.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Dict, Optional

import numpy as np

from .inputs import read_input
from .params import Params

IX_N, IX_U, IX_V = 0, 1, 2          # shared_mod.f95:106-108 (1-based there)
f8 = np.float64
f4 = np.float32


@dataclasses.dataclass
class Fields:
    """Everything the engine owns (module state of private_mod.f95:27-93)."""
    p: Params
    neig: np.ndarray      # int32 [ndeg+1, 8]
    subc: np.ndarray      # int32 [2, ndeg+1]
    posc: np.ndarray      # int32 [ndeg]    (grid.bin record 1)
    mk_u: np.ndarray; mk_v: np.ndarray; mk_n: np.ndarray; mkpe: np.ndarray; mkpi: np.ndarray
    fcor: np.ndarray; h_th: np.ndarray; h_to: np.ndarray
    h_0: np.ndarray       # f8 [nlay, ndeg+1]
    hlay: np.ndarray; u: np.ndarray; v: np.ndarray; h_u: np.ndarray; h_v: np.ndarray
    rs_h: np.ndarray; dmdx: np.ndarray; dmdy: np.ndarray
    v_cc: np.ndarray; v_ll: np.ndarray
    tt3d: np.ndarray; tb3d: np.ndarray; tu3d: np.ndarray; taus: np.ndarray
    fnud: np.ndarray; nudg: np.ndarray; hdot: np.ndarray
    tide: np.ndarray; w_ti: np.ndarray; bodf: np.ndarray
    invf: float
    flag_nudging: bool
    has: Dict[str, bool]  # which optional input files were present
    segm: Optional[np.ndarray] = None   # int32 [18, nseg] = Fortran segm(nseg, 18): nudged open-boundary segments
    # rigid lid (rgld = 1, private_mod.f95:505-563): surface pressure and the operators of its Poisson equation, f8 [ndeg+1]
    pi_s: Optional[np.ndarray] = None
    Ow: Optional[np.ndarray] = None
    Os: Optional[np.ndarray] = None
    Osum_: Optional[np.ndarray] = None
    tres: float = 0.0     # time of the record a restarted run (rsta = 1) continues from, days (:1311-1325); ctim = tres + dtd8*tstp

    @property
    def ndeg(self): return self.p.ndeg
    @property
    def nlay(self): return self.p.nlay


# ---------------------------------------------------------------------------
def _frame(lm, mm):
    """h_2d(-1:lm+2, -1:mm+2) as a [lm+4, mm+4] array with offset 1."""
    return np.zeros((lm + 4, mm + 4), dtype=f8)


def default_depth(p: Params) -> np.ndarray:
    h_2d = _frame(p.lm, p.mm)
    # :121  cext**2._rw / grav   (real power 2.0 → exact square)
    h_2d[2:p.lm + 2, 2:p.mm + 2] = (p.cext * p.cext) / p.grav
    return h_2d


def apply_h_bo(p: Params, h_2d: np.ndarray, h_bo_r4: np.ndarray, h_to_r4=None, window=None) -> np.ndarray:
    """read_input_file('h_bo'), private_mod.f95:827-839.  window: the frame is a band of rows of a
    taller frame — its first / last row is forced dry only where it is the taller frame's margin."""
    lm, mm = p.lm, p.mm
    h_2d[:, :] = 0.0
    if h_to_r4 is not None and p.topt > 0.5:
        h_2d[1:lm + 3, 1:mm + 3] = (h_bo_r4.astype(f4) - h_to_r4.astype(f4)).astype(f8)
    else:
        h_2d[1:lm + 3, 1:mm + 3] = h_bo_r4.astype(f8)
    h_2d[h_2d < p.hdry] = 0.0
    h_2d[1, :] = 0.0; h_2d[lm + 2, :] = 0.0
    if window is None or window["bottom_margin"]:
        h_2d[:, 1] = 0.0
    if window is None or window["top_margin"]:
        h_2d[:, mm + 2] = 0.0
    return h_2d


def index_grid_points(p: Params, h_2d: np.ndarray, window=None):
    """private_mod.f95:567-764.  Returns dict with neig, subc, posc, masks, h_th.
    window: frame row 0 holds the (wet) row below a band of rows; it is looked at, not packed."""
    lm, mm, ndeg = p.lm, p.mm, p.ndeg
    wet = h_2d > p.hdry                              # [lm+4, mm+4], offset 1
    # frame cells i=0..lm+1, j=0..mm+1 → array index +1
    W = lambda di, dj: wet[1 + di:lm + 3 + di, 1 + dj:mm + 3 + dj]
    w00, wm0, w0m, wmm = W(0, 0), W(-1, 0), W(0, -1), W(-1, -1)
    incl = w00 | wm0 | w0m | wmm                     # :594-597
    if window is not None:
        incl = incl.copy()
        incl[:, 0] = False
    order = np.flatnonzero(incl.ravel(order="F"))    # j outer, i inner
    i_c = order.size
    if i_c != ndeg:                                  # :604-610
        raise ValueError("wrong input parameter! Please set ndeg = %d" % i_c)
    indc = np.zeros((lm + 4, mm + 4), dtype=np.int64)
    sub = indc[1:lm + 3, 1:mm + 3]
    flat = np.zeros(incl.size, dtype=np.int64)
    flat[order] = np.arange(1, ndeg + 1)
    sub[:, :] = flat.reshape(incl.shape, order="F")
    I = lambda i: i + 1                              # frame index → array index
    mk = {k: np.zeros(ndeg + 1, dtype=f8) for k in ("mk_u", "mk_v", "mk_n", "mkpe", "mkpi")}
    h = lambda i, j: h_2d[I(i), I(j)] > p.hdry

    if p.xper > 0.5:                                 # :614-640
        i = 1
        for j in range(1, mm + 1):
            if h(i, j) and h(lm, j):
                indc[I(0), I(j)] = indc[I(lm), I(j)]
                indc[I(lm + 1), I(j)] = indc[I(1), I(j)]
                mk["mk_u"][indc[I(i), I(j)]] = 1.0
            if j > 1:
                if h(i, j - 1) and h(i, j) and h(lm, j - 1) and h(lm, j):
                    mk["mkpe"][indc[I(i), I(j)]] = 1.0
            if j == mm:
                if h(i, j) and h(lm, j):
                    indc[I(0), I(mm + 1)] = indc[I(lm), I(mm + 1)]
                    indc[I(lm + 1), I(mm + 1)] = indc[I(1), I(mm + 1)]
    if p.yper > 0.5:                                 # :642-668
        j = 1
        for i in range(1, lm + 1):
            if h(i, j) and h(i, mm):
                indc[I(i), I(0)] = indc[I(i), I(mm)]
                indc[I(i), I(mm + 1)] = indc[I(i), I(1)]
                mk["mk_v"][indc[I(i), I(j)]] = 1.0
            if i > 1:
                if h(i - 1, j) and h(i, j) and h(i - 1, mm) and h(i, mm):
                    mk["mkpe"][indc[I(i), I(j)]] = 1.0
            if i == lm:
                if h(i, mm) and h(i, j):
                    indc[I(lm + 1), I(0)] = indc[I(lm + 1), I(mm)]
                    indc[I(lm + 1), I(mm + 1)] = indc[I(lm + 1), I(1)]
    if p.xper > 0.5 and p.yper > 0.5:                # :672-685
        if h(1, 1) and h(lm, 1) and h(1, mm):
            indc[I(0), I(0)] = indc[I(lm), I(mm)]
            mk["mkpe"][indc[I(1), I(1)]] = 1.0
            indc[I(0), I(mm + 1)] = indc[I(lm), I(1)]
        if h(lm, mm) and h(1, mm) and h(lm, 1):
            indc[I(lm + 1), I(0)] = indc[I(1), I(mm)]
            indc[I(lm + 1), I(mm + 1)] = indc[I(1), I(1)]
    mk["mk_u"][0] = 0.0; mk["mk_v"][0] = 0.0; mk["mkpe"][0] = 0.0   # indc==0 writes hit the sentinel in
    # the reference too only if the cell is not packed, which cannot happen for wet (i,j).

    ii, jj = np.unravel_index(order, incl.shape, order="F")  # frame i, j of each packed cell
    sel = (ii, jj)
    idx = np.arange(1, ndeg + 1)
    mk["mk_n"][idx] = np.where(w00[sel], 1.0, mk["mk_n"][idx])                       # :701
    mk["mk_u"][idx] = np.where(w00[sel] & wm0[sel], 1.0, mk["mk_u"][idx])            # :703-705
    mk["mk_v"][idx] = np.where(w00[sel] & w0m[sel], 1.0, mk["mk_v"][idx])            # :706-708
    mk["mkpe"][idx] = np.where(w00[sel] & wm0[sel] & w0m[sel] & wmm[sel], 1.0, mk["mkpe"][idx])
    mk["mkpi"][idx] = np.where(incl[sel], 1.0, 0.0)                                  # :712-714
    posc = (ii + 1 + jj * (lm + 2)).astype(np.int32)                                 # :716
    subc = np.zeros((2, ndeg + 1), dtype=np.int32)
    subc[0, 1:] = ii; subc[1, 1:] = jj
    neig = np.zeros((ndeg + 1, 8), dtype=np.int32)
    offs = ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))    # :719-726
    for k, (di, dj) in enumerate(offs):
        neig[1:, k] = indc[ii + 1 + di, jj + 1 + dj]
    h_th = np.zeros(ndeg + 1, dtype=f8)
    h_th[1:] = h_2d[ii + 1, jj + 1]                                                  # :753-757
    h_th[0] = h_2d[I(0), I(0)]
    return dict(neig=neig, subc=subc, posc=posc, h_th=h_th, **mk)


def index_boundary_points(p: Params, nd: np.ndarray, h_2d: np.ndarray, band: bool = False) -> np.ndarray:
    """private_mod.f95:1060-1240 — the table of nudged open-boundary segments used by
    no_gradient_obc (:2613-2679).  nd = nudg.bin content [lm+2, mm+2, 3] (real*4).
    Returns int32 [18, nseg] (the bytes of Fortran segm(nseg, 18)); raises like the reference
    when no segment exists.  band: the frame is a band of rows of a taller frame (read_input_data's
    window): its row 0 is the (wet) row below the band — looked at, neither packed nor searched."""
    lm, mm = p.lm, p.mm
    wet = h_2d > p.hdry                                   # offset 1: frame (i, j) -> [i+1, j+1]
    W = lambda i, j: bool(wet[i + 1, j + 1])
    # packed numbering exactly as in index_grid_points, WITHOUT the periodic overwrites (:1088-1097)
    incl = wet[1:lm + 3, 1:mm + 3] | wet[0:lm + 2, 1:mm + 3] | wet[1:lm + 3, 0:mm + 2] | wet[0:lm + 2, 0:mm + 2]
    if band:
        incl[:, 0] = False
    indc = np.zeros((lm + 4, mm + 4), dtype=np.int64)
    flat = np.zeros(incl.size, dtype=np.int64)
    order = np.flatnonzero(incl.ravel(order="F"))
    flat[order] = np.arange(1, order.size + 1)
    indc[1:lm + 3, 1:mm + 3] = flat.reshape(incl.shape, order="F")
    I = lambda i, j: int(indc[i + 1, j + 1])
    tiny = np.finfo(np.float32).tiny
    xopen, yopen = float(p.xper) < 0.5, float(p.yper) < 0.5
    rows = []
    for j in range(1 if band else 0, mm + 2):
        for i in range(0, lm + 2):
            if W(i, j) and not W(i - 1, j) and xopen and nd[i, j, IX_U] > tiny and nd[i - 1, j, IX_U] > tiny:       # west
                rows.append([I(i, j), i, j, 1, 0, 1, I(i - 1, j), i - 1, j, I(i, j), i, j, I(i + 1, j), i + 1, j, I(i + 1, j), i + 1, j])
            if (not W(i, j)) and W(i - 1, j) and xopen and nd[i - 1, j, IX_U] > tiny and nd[i, j, IX_U] > tiny:     # east
                rows.append([I(i, j), i, j, 1, 0, -1, I(i, j), i, j, I(i - 1, j), i - 1, j, I(i - 1, j), i - 1, j, I(i - 2, j), i - 2, j])
            if W(i, j) and not W(i, j - 1) and yopen and nd[i, j, IX_V] > tiny and nd[i, j - 1, IX_V] > tiny:       # south
                rows.append([I(i, j), i, j, 0, 1, 1, I(i, j - 1), i, j - 1, I(i, j), i, j, I(i, j + 1), i, j + 1, I(i, j + 1), i, j + 1])
            if (not W(i, j)) and W(i, j - 1) and yopen and nd[i, j - 1, IX_V] > tiny and nd[i, j, IX_V] > tiny:     # north
                rows.append([I(i, j), i, j, 0, 1, -1, I(i, j), i, j, I(i, j - 1), i, j - 1, I(i, j - 1), i, j - 1, I(i, j - 2), i, j - 2])
    if not rows:
        raise ValueError("the nudged open boundary segments could not be identified.")
    return np.ascontiguousarray(np.array(rows, dtype=np.int32).T)


def _seq_sum(a: np.ndarray, axis: int) -> np.ndarray:
    """Left-to-right sum along axis (the order flang emits for SUM of a short section)."""
    a = np.moveaxis(a, axis, 0)
    s = np.zeros(a.shape[1:], dtype=a.dtype)
    for k in range(a.shape[0]):
        s = s + a[k]
    return s


def equilibrium_h0_noocrp(p: Params, g, h_2d, dmax=None) -> np.ndarray:
    """private_mod.f95:154-175."""
    nlay, ndeg = p.nlay, p.ndeg
    dmax = f8(h_2d.max()) if dmax is None else f8(dmax)
    h_0 = np.zeros((nlay, ndeg + 1), dtype=f8)
    wetn = g["mk_n"] > 0.5
    hb = g["h_th"]                                     # = h_2d(i,j)
    topl = p.topl_v
    for il in range(nlay - 1, -1, -1):
        habv = dmax * topl[il] if il > 0 else f8(0.0)
        hbel = _seq_sum(h_0[il + 1:], 0) if il < nlay - 1 else 0.0
        val = hb - habv - hbel
        h_0[il] = np.where(wetn, val, 0.0)
    h_0[:, 0] = 0.0
    return h_0


def _powi(x, n):
    """x**n for a CONSTANT integer n as flang emits it: the left-to-right product ((x*x)*x)*...
    (nsal is a parameter of shared_mod.f95; probed with flang 22 for n = 4..8 — a run-time
    exponent would use repeated squaring, which differs in the last bit from n = 4 on)."""
    r = x
    for _ in range(n - 1):
        r = r * x
    return r


def equilibrium_h0_ocrp(p: Params, g, h_2d, dmax=None) -> np.ndarray:
    """private_mod.f95:309-476 — per-cell Newton with Gaussian elimination and SOR,
    vectorised over cells (every cell performs the reference's scalar operations in
    the reference's order; converged cells are frozen)."""
    nlay, ndeg, nsal = p.nlay, p.ndeg, p.nsal
    thre = f8(p.tole)
    dmax = f8(h_2d.max()) if dmax is None else f8(dmax)
    rho8 = p.rhon_v
    topl = p.topl_v
    hsal = f8(p.hsal)
    sor = f8(p.sor)
    gues0 = np.zeros(nlay)
    for il in range(nlay):                                         # :339-345
        gues0[il] = dmax * (1.0 - topl[il])
        if il < nlay - 1:
            gues0[il] = gues0[il] - dmax * (1.0 - topl[il + 1])
    cons = np.zeros(nlay)
    for il in range(nlay):                                         # :349-355
        cons[il] = dmax * (-1.0) + _seq_sum(gues0, 0)
        for k in range(il):
            cons[il] = cons[il] - (rho8[il] - rho8[k]) * gues0[k] / rho8[il]
    cells_all = np.flatnonzero(g["mk_n"] > 0.5)
    # the column solve depends on the cell only through its depth: solve once per distinct depth
    hbot, inverse = np.unique(g["h_th"][cells_all].astype(f8), return_inverse=True)
    cells = np.arange(hbot.size)                  # columns of the reduced problem
    n = cells.size
    gues = np.zeros((nlay, n))
    for il in range(nlay - 1, -1, -1):                             # :370-380
        habv = dmax * topl[il]
        hbel = _seq_sum(gues[il + 1:], 0) if il < nlay - 1 else 0.0
        gues[il] = np.maximum(hbot - habv - hbel, hsal)
    h_u = np.zeros((nlay, n), dtype=f8)           # solution per distinct depth
    active = np.arange(n)
    for it in range(1, p.itmx + 1):
        G = gues[:, active]
        hb = hbot[active]
        func = np.zeros_like(G)
        sg = _seq_sum(G, 0)
        for i in range(nlay):                                      # :383-393
            f = (hb - sg) + 1.0 / f8(nsal - 1) * hsal * _powi(hsal / G[i], nsal - 1) + cons[i]
            f = f * (-1.0)
            for j in range(i):
                f = f - (rho8[i] - rho8[j]) * G[j] / rho8[i]
            func[i] = f
        if it == p.itmx:
            raise RuntimeError("calculation of h_layers did not converge")
        conv = np.all(np.abs(func) < thre, axis=0)                 # :403-408
        if conv.any():
            h_u[:, cells[active[conv]]] = G[:, conv]
        keep = ~conv
        if not keep.any():
            break
        active = active[keep]
        G = G[:, keep]; func = func[:, keep]
        m = active.size
        maug = np.zeros((nlay, nlay + 1, m))
        for i in range(nlay):                                      # :410-419
            for j in range(nlay):
                maug[i, j] = min(rho8[i], rho8[j]) / rho8[i]
                if i == j:
                    maug[i, j] = maug[i, j] + _powi(hsal / G[j], nsal)
        maug[:, nlay] = func * (-1.0)
        ar = np.arange(m)
        for k in range(nlay):                                      # :426-455
            sub = np.abs(maug[k:, k])
            maxv = np.zeros(m); imax = np.full(m, -1)
            for r in range(sub.shape[0]):
                better = sub[r] > maxv
                maxv = np.where(better, sub[r], maxv)
                imax = np.where(better, k + r, imax)
            sw = (imax != k) & (imax >= 0)
            if sw.any():
                rows = imax[sw]
                tmp = maug[k][:, sw].copy()
                maug[k][:, sw] = maug[rows, :, ar[sw]].T
                maug[rows, :, ar[sw]] = tmp.T
            for il in range(k + 1, nlay):
                for l in range(k, nlay + 1):
                    # the factor is re-read inside the l loop (:449-450): after l == k it is
                    # built from the already-eliminated maug(il,k).  Reference behaviour, kept.
                    maug[il, l] = maug[il, l] - maug[k, l] * (maug[il, k] / maug[k, k])
                maug[il, k] = 0.0
        for il in range(nlay - 1, -1, -1):                         # :459-466
            resu = np.zeros(m)
            for j in range(il + 1, nlay):
                resu = resu + maug[il, j] * maug[j, nlay]
            maug[il, nlay] = (maug[il, nlay] - resu) / maug[il, il]
        G = (1.0 - sor) * G + sor * (maug[:, nlay] + G)            # :468
        low = np.any(G <= thre, axis=0)                            # :470-472
        if low.any():
            G[:, low] = np.maximum(G[:, low], thre)
        gues[:, active] = G
    h_0 = np.zeros((nlay, ndeg + 1), dtype=f8)
    h_0[:, cells_all] = h_u[:, inverse]
    return h_0


def rigid_lid_operators(p: Params, g, h_0: np.ndarray):
    """private_mod.f95:505-563 — start value of the lid pressure and the operators Ow, Os, 1/Osum of its Poisson
    equation.  In the reference this block sits at the end of get_equilibrium_thickness_h_0, i.e. it only runs
    with ocrp = 1; otherwise all four stay zero (and the pressure solve is a no-op)."""
    n1 = p.ndeg + 1
    z = lambda: np.zeros(n1, dtype=f8)
    pi_s, Ow, Os, Osum, Osum_ = z(), z(), z(), z(), z()
    if p.ocrp < 0.5:
        return pi_s, Ow, Os, Osum_
    lm, mm = p.lm, p.mm
    h_th, neig = g["h_th"], g["neig"].astype(np.int64)
    i, j = g["subc"][0].astype(np.int64), g["subc"][1].astype(np.int64)
    c1, c3, c5, c7 = neig[:, 0], neig[:, 2], neig[:, 4], neig[:, 6]
    pi_s[:] = (_seq_sum(h_0, 0) - h_th) * p.grav                          # :509, index 0 included
    dl2 = p.dl * p.dl
    hw = 0.5 * (h_th + h_th[c5])
    hs = 0.5 * (h_th + h_th[c7])
    a = (1 < i) & (i < lm + 1) & (1 < j) & (j < mm + 1)
    b = (i == 1) & (1 < j) & (j < mm + 1)
    c = (1 < i) & (i < lm + 1) & (j == 1)
    Ow[:] = np.where(a | c, hw / dl2, 0.0)
    Os[:] = np.where(a | b, hs / dl2, 0.0)
    Ow[0] = 0.0; Os[0] = 0.0
    s1 = ((Ow + Ow[c1]) + Os) + Os[c3]
    s2 = (Ow + Os) + Os[c3]
    s3 = (Ow + Os) + Ow[c1]
    s4 = Ow + Os
    Osum[:] = np.where((i < lm) & (j < mm), s1, np.where((i == lm) & (j < mm), s2, np.where((j == mm) & (i < lm), s3, s4)))
    inside = (i > 0) & (i < lm + 1) & (j > 0) & (j < mm + 1)
    with np.errstate(divide="ignore"):
        Osum_[:] = np.where(inside, 1.0 / Osum, 0.0)
    Osum_[0] = 0.0
    return pi_s, Ow, Os, Osum_


def _unpack(arr2d: np.ndarray, subc: np.ndarray) -> np.ndarray:
    out = np.zeros(subc.shape[1], dtype=arr2d.dtype)
    out[1:] = arr2d[subc[0, 1:], subc[1, 1:]]
    return out


def fcor_mean_r4(ior4: np.ndarray) -> np.float32:
    """fcor(0) = sum(ior4)/size in real*4 (private_mod.f95:933), column-major order."""
    return np.float32(np.add.reduce(ior4.ravel(order="F"), dtype=np.float32) / np.float32(ior4.size))


def apply_restart_record(f: Fields, eta: np.ndarray, u4: np.ndarray, v4: np.ndarray, tres: float) -> Fields:
    """read_restart_record / read_array (private_mod.f95:1299-1420): the state a run with rsta = 1 starts from is
    the LAST complete output record of the previous run — real*4 `u___`, `v___` widened, and the interface
    elevations `eta_` turned back into thicknesses with the real*4 h_0 of h_0.bin (which init has just rewritten,
    :185-193): hlay = (h_0 + eta_k) - eta_k+1, times mk_n.  eta, u4, v4: [nlay, ndeg] real*4; tres = the last
    entry of time.txt (days)."""
    nlay = f.p.nlay
    eta = np.asarray(eta, dtype=f4).reshape(nlay, f.p.ndeg)
    h0 = f.h_0[:, 1:].astype(f4)
    for il in range(nlay):
        h = h0[il].astype(f8) + eta[il].astype(f8)
        if il < nlay - 1:
            h = h - eta[il + 1].astype(f8)
        f.hlay[il, 1:] = h * f.mk_n[1:]
    f.u[:, 1:] = np.asarray(u4, dtype=f4).reshape(nlay, f.p.ndeg).astype(f8)
    f.v[:, 1:] = np.asarray(v4, dtype=f4).reshape(nlay, f.p.ndeg).astype(f8)
    f.tres = float(tres)
    return f


def restart_from_files(f: Fields, odir_files: Dict[str, bytes]) -> Fields:
    """The same from the previous run's output files: {"time.txt": text, "eta_.bin" | "u___.bin" | "v___.bin": bytes}."""
    stamps = []
    for line in str(odir_files["time.txt"]).split("\n"):       # list-directed reads until the first failure (:1313-1322)
        try:
            stamps.append(float(line.split()[0].replace("D", "E").replace("d", "e")))
        except (ValueError, IndexError):
            break
    irec, n = len(stamps), f.p.nlay * f.p.ndeg
    rec = lambda k: np.frombuffer(bytes(odir_files[k]), dtype="<f4")[(irec - 1) * n: irec * n]
    return apply_restart_record(f, rec("eta_.bin"), rec("u___.bin"), rec("v___.bin"), stamps[-1])


def read_input_data(p: Params, idir: Optional[str] = None,
                    files: Optional[Dict[str, np.ndarray]] = None, window: Optional[dict] = None) -> Fields:
    """private_mod.f95:105-250 (without the output calls).  Inputs come from
    ``idir/*.bin`` or from an in-memory dict of arrays (rounded to real*4 here).
    A restarted run (rsta = 1) then takes its state from the previous run's last record: restart_from_files.

    window (multi-GPU, beom_amd/slab.py): the frame described by `p` and `files` is the band of rows
    j0..j1 of a taller DENSE frame — frame row 0 is the taller frame's row j0-1 (looked at by the wet
    tests and the (j-1) averages, never packed), frame rows 1..mm+1 are packed.  What init derives from
    the WHOLE frame comes in through the dict: bottom_margin/top_margin (is row 0 / row mm+1 the taller
    frame's margin), dmax/dmin (deepest / shallowest wet depth, :132-144,154-183), fcor0 and invf
    (:933, :223-229).  No open-boundary segment table (:1060-1240; bands refuse mcbc = 0)."""
    lm, mm, nlay, ndeg = p.lm, p.mm, p.nlay, p.ndeg

    def get(key, shape):
        if files is not None:
            if key not in files:
                return None
            a = np.asarray(files[key]).astype(f4)
            assert a.shape == tuple(shape), (key, a.shape, shape)
            return a
        return read_input(idir, key, shape)

    h_2d = default_depth(p)
    hb = get("h_bo", (lm + 2, mm + 2))
    has = {}
    has["h_bo"] = hb is not None
    if window is not None and hb is None:
        raise ValueError("a window of rows needs an h_bo array")
    if hb is not None:
        h_to = get("h_to", (lm + 2, mm + 2)) if p.topt > 0.5 else None
        apply_h_bo(p, h_2d, hb, h_to, window)
    g = index_grid_points(p, h_2d, window)
    subc = g["subc"]
    wet = h_2d[h_2d > p.hdry]
    dmin = f8(wet.min()) if wet.size else f8(0)
    dmax = f8(h_2d.max())
    if window is not None:
        dmin, dmax = f8(window["dmin"]), f8(window["dmax"])
    if p.ocrp < 0.5 and nlay > 1:                                   # :137-144
        if p.topl_v[nlay - 1] * dmax + 10.0 * p.hmin >= dmin:
            raise ValueError("Please modify topl so that bathymetry is contained within lower layer.")
    elif p.ocrp < 0.5 and nlay == 1:
        if dmin <= 10.0 * p.hmin:
            raise ValueError("Please adjust h_bo or hmin so that min(h_bo) > 10. * hmin.")
    if p.ocrp < 0.5:
        h_0 = equilibrium_h0_noocrp(p, g, h_2d, dmax)
    else:
        h_0 = equilibrium_h0_ocrp(p, g, h_2d, dmax)
    rl = rigid_lid_operators(p, g, h_0) if p.rgld > 0.5 else (None, None, None, None)
    z2 = lambda: np.zeros((nlay, ndeg + 1), dtype=f8)
    hlay = h_0 * g["mk_n"][None, :]                                 # :198-200
    u, v = z2(), z2()
    fnud = np.zeros((3, nlay, ndeg + 1), dtype=f8)
    nudg = np.zeros((3, ndeg + 1), dtype=f8)
    flag_nudging = False
    segm = None
    nd = get("nudg", (lm + 2, mm + 2, 3))
    has["nudg"] = nd is not None
    ii, jj = subc[0, 1:], subc[1, 1:]
    if nd is not None:                                              # :843-881
        nudg[IX_N, 1:] = nd[ii, jj, IX_N].astype(f8)
        im, jm = np.maximum(ii - 1, 0), np.maximum(jj - 1, 0)       # i-1 ≥ 0 for packed cells with i≥1
        # cells with i=0 (or j=0) index ior4(-1,…) in the reference: out of bounds there;
        # they are never u (v) points, so the value is irrelevant and we take 0.
        ok_u = (nd[ii, jj, IX_U] > f4(1e-9)) & (nd[im, jj, IX_U] > f4(1e-9)) & (ii > 0)
        nudg[IX_U, 1:] = np.where(ok_u, nd[ii, jj, IX_U].astype(f8) * 0.5 + nd[im, jj, IX_U].astype(f8) * 0.5, 0.0)
        ok_v = (nd[ii, jj, IX_V] > f4(1e-9)) & (nd[ii, jm, IX_V] > f4(1e-9)) & (jj > 0)
        nudg[IX_V, 1:] = np.where(ok_v, nd[ii, jj, IX_V].astype(f8) * 0.5 + nd[ii, jm, IX_V].astype(f8) * 0.5, 0.0)
        flag_nudging = bool(np.any(nudg > 1e-9))
        if flag_nudging and window is None:
            segm = index_boundary_points(p, nd, h_2d)               # :868-871
        elif window is not None and float(p.mcbc) < 0.5:
            # a band of rows finds the segments of its own rows (the finder looks at a cell and its four neighbours; the rows
            # below and above the band are in the frame's margins, dry only where they are the whole frame's margins)
            pseg = p
            if "yper" in window:                                    # (a band never wraps by itself, but the frame may: no N/S segments then)
                pseg = Params.from_json(p.to_json())
                pseg.lits["yper"] = window["yper"]
            try:
                segm = index_boundary_points(pseg, nd, h_2d, band=True)
            except ValueError:
                segm = None                                         # none in these rows
        fnud[IX_N, :, 1:] = hlay[:, 1:]                             # :874-881
    it = get("init", (lm + 2, mm + 2, nlay, 3))
    has["init"] = it is not None
    if it is not None:                                              # :882-910
        for il in range(nlay):
            e0 = it[ii, jj, il, IX_N].astype(f8)
            if il < nlay - 1:
                e1 = it[ii, jj, il + 1, IX_N].astype(f8)
                fn = hlay[il, 1:] + e0 - e1
            else:
                fn = hlay[il, 1:] + e0
            fnud[IX_N, il, 1:] = fn * g["mk_n"][1:]
            fnud[IX_U, il, 1:] = it[ii, jj, il, IX_U].astype(f8)
            fnud[IX_V, il, 1:] = it[ii, jj, il, IX_V].astype(f8)
            if p.rsta < 0.5:
                hlay[il, 1:] = fnud[IX_N, il, 1:] * g["mk_n"][1:]
                u[il, 1:] = fnud[IX_U, il, 1:]
                v[il, 1:] = fnud[IX_V, il, 1:]
    bodf = np.zeros((2, nlay), dtype=f8)
    bf = get("bodf", (nlay, 2))
    has["bodf"] = bf is not None
    if bf is not None:
        bodf[:, :] = bf.astype(f8).T
    hdot = z2()
    hd = get("hdot", (lm + 2, mm + 2, nlay))
    has["hdot"] = hd is not None
    if hd is not None:
        for il in range(nlay):
            hdot[il, 1:] = hd[ii, jj, il].astype(f8)
    taus = np.zeros((2, ndeg + 1), dtype=f8)
    tw = p.tauw_v
    taus[0, :] = tw[0]; taus[1, :] = tw[1]                          # :302-303
    ts = get("taus", (lm + 2, mm + 2, 2))
    has["taus"] = ts is not None
    if ts is not None:
        taus[:, :] = 0.0
        taus[0, 1:] = ts[ii, jj, 0].astype(f8)
        taus[1, 1:] = ts[ii, jj, 1].astype(f8)
    tide = np.zeros((3, ndeg + 1, 1, 2), dtype=f8)
    w_ti = np.zeros(1, dtype=f8)
    td = get("tide", (2, 1, lm + 2, mm + 2, 3))
    has["tide"] = td is not None
    if td is not None:                                              # :951-964
        w_ti[0] = f8(td[0, 0, 0, 0, 0])
        for var in range(3):
            tide[var, 1:, 0, 0] = td[0, 0, ii, jj, var].astype(f8)
            tide[var, 1:, 0, 1] = td[1, 0, ii, jj, var].astype(f8)
    fcor = np.full(ndeg + 1, p.f0, dtype=f8)                        # :301
    fc = get("fcor", (lm + 2, mm + 2))
    has["fcor"] = fc is not None
    if fc is not None:                                              # :932-950
        fcor[0] = f8(fcor_mean_r4(fc))
        q = f4(0.25)
        im, jm = np.maximum(ii - 1, 0), np.maximum(jj - 1, 0)
        interp = ((fc[ii, jj] * q + fc[im, jj] * q) + fc[ii, jm] * q) + fc[im, jm] * q
        fcor[1:] = np.where((ii > 0) & (jj > 0), interp.astype(f8), fc[ii, jj].astype(f8))
    invf = f8(np.add.reduce(fcor) / f8(fcor.size))                  # :223 (summation order: see DESIGN.md)
    invf = f8(1.0) / invf if abs(invf) > 1.25e-5 else f8(0.0)
    if window is not None:
        if fc is not None:
            fcor[0] = f8(window["fcor0"])
        invf = f8(window["invf"])
    bv = f8(p.bvis)
    return Fields(
        p=p, neig=g["neig"], subc=subc, posc=g["posc"],
        mk_u=g["mk_u"], mk_v=g["mk_v"], mk_n=g["mk_n"], mkpe=g["mkpe"], mkpi=g["mkpi"],
        fcor=fcor, h_th=g["h_th"], h_to=np.zeros(ndeg + 1, dtype=f8), h_0=h_0,
        hlay=hlay, u=u, v=v, h_u=z2(), h_v=z2(),
        rs_h=np.zeros((nlay, ndeg + 1, 2), dtype=f8),
        dmdx=np.zeros((nlay, ndeg + 1, 3), dtype=f8), dmdy=np.zeros((nlay, ndeg + 1, 3), dtype=f8),
        v_cc=np.full((nlay, ndeg + 1), bv), v_ll=np.full((nlay, ndeg + 1), bv),
        tt3d=np.zeros((nlay, 2, ndeg + 1), dtype=f8), tb3d=np.zeros((nlay, 2, ndeg + 1), dtype=f8),
        tu3d=np.zeros((nlay, 2, ndeg + 1), dtype=f8), taus=taus,
        fnud=fnud, nudg=nudg, hdot=hdot, tide=tide, w_ti=w_ti, bodf=bodf,
        invf=float(invf), flag_nudging=flag_nudging, has=has, segm=segm,
        pi_s=rl[0], Ow=rl[1], Os=rl[2], Osum_=rl[3])
