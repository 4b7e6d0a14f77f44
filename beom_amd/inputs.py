"""Input half of the testcases/*.m file contract, restated (no Octave in the image).

Files are little-endian real*4, Fortran (column-major) order on the frame
``(0:lm+1, 0:mm+1)`` exactly as ``read_input_file`` expects them
(private_mod.f95:766-967):

  h_bo.bin, fcor.bin  [lm+2, mm+2]          (:778)
  taus.bin            [lm+2, mm+2, 2]       (:800)
  nudg.bin            [lm+2, mm+2, 3]       (:785)   eta,u,v relaxation rates
  hdot.bin            [lm+2, mm+2, nlay]    (:794)
  init.bin            [lm+2, mm+2, nlay, 3] (:788)   interface eta, u, v
  bodf.bin            [nlay, 2]             (:791)
  tide.bin            [2, ncon, lm+2, mm+2, 3] (:797), omega at (1,k,0,0,1) (:953)

Each ``case_*`` function follows the cited script's arithmetic on a grid size of the
caller's choice and returns ``(Params, files)`` where ``files`` maps keyword →
numpy array indexed ``[i, j, ...]`` (first index = x).  Arrays are float64 here and
rounded to real*4 only when written, as Octave's ``fwrite(...,'real*4')`` does.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

from .params import Params, make_params

HDRY = 1.0e-3


def write_inputs(idir: str, files: Dict[str, np.ndarray]) -> None:
    os.makedirs(idir, exist_ok=True)
    for key, arr in files.items():
        a = np.asarray(arr)
        a.astype("<f4").ravel(order="F").tofile(os.path.join(idir, key + ".bin"))


def read_input(idir: str, key: str, shape) -> np.ndarray | None:
    path = os.path.join(idir, key + ".bin")
    if not os.path.exists(path):
        return None
    a = np.fromfile(path, dtype="<f4")
    return a.reshape(shape, order="F")


def get_nbr_deg_freedom(h_bo: np.ndarray) -> int:
    """testcases/get_nbr_deg_freedom.m:1-61 — count of packed cells."""
    h = np.array(h_bo, dtype=np.float64)
    h[h < 2.0 * HDRY] = 0.0
    h[0, :] = 0.0; h[-1, :] = 0.0; h[:, 0] = 0.0; h[:, -1] = 0.0
    lm, mm = h.shape[0] - 2, h.shape[1] - 2
    hext = np.zeros((lm + 4, mm + 4))
    hext[1:-1, 1:-1] = h
    mask = (hext > HDRY).astype(np.int64)
    neig = mask[1:-1, 1:-1] + mask[:-2, 1:-1] + mask[1:-1, :-2] + mask[:-2, :-2]
    return int(np.count_nonzero(neig > 0))


def _edge_replicate(inner: np.ndarray) -> np.ndarray:
    """stommel1948.m:57-62 — embed [lm,mm,...] in [lm+2,mm+2,...] copying edges."""
    shp = (inner.shape[0] + 2, inner.shape[1] + 2) + inner.shape[2:]
    t = np.full(shp, np.nan)
    t[1:-1, 1:-1] = inner
    t[0, :] = t[1, :]
    t[:, 0] = t[:, 1]
    t[-1, :] = t[-2, :]
    t[:, -1] = t[:, -2]
    return t


GRAV = 9.8


# ---------------------------------------------------------------------------
def case_stommel(lm: int = 100, mm: int = 63, dl: float = 100.0e3, dt_s: float = 40.0,
                 dt_o: float = 1.0) -> Tuple[Params, Dict[str, np.ndarray]]:
    """testcases/stommel1948.m:17-88 (BASELINE config 1).  The script's own size is
    lm=round(1e7/dl)=100, mm=round(2*pi*1e6/dl)=63; other sizes keep lambda=lm*dl,
    b=mm*dl."""
    b = mm * dl if (lm, mm) != (100, 63) else 2.0 * np.pi * 1.0e6
    rhon = 1027.0
    hfla = 200.0
    F = 0.1 / rhon
    R = 2.0e-4
    beta, fmin = 1.0e-11, 0.0
    xx, yy = np.meshgrid((np.arange(lm) + 0.5) * dl, (np.arange(mm) + 0.5) * dl, indexing="ij")
    h_bo = np.zeros((lm + 2, mm + 2))
    h_bo[1:-1, 1:-1] = hfla
    ndeg = get_nbr_deg_freedom(h_bo)
    fcor = np.empty((lm, mm))
    for j in range(1, mm + 1):
        fcor[:, j - 1] = fmin + (j - 0.5) * dl * beta
    tausx = -rhon * F * np.cos(np.pi * yy / b)
    taus = np.stack([tausx, np.zeros_like(tausx)], axis=2)
    cext = np.sqrt(GRAV * h_bo.max())
    p = make_params(lm, mm, 1, ndeg, dl, cext, 0.0, [rhon], [0.0], dt_s, dt_o, 0.0, 0.0, 0.0,
                    0.0, R, 1.0, 10.0, 10.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for Stommel 1948")
    # no h_bo.bin: the script relies on the default flat depth cext**2/grav (:121)
    return p, {"fcor": _edge_replicate(fcor), "taus": _edge_replicate(taus)}


def case_soliton(lm: int = 307, mm: int = 153, dt_s: float = 60.0, dt_o: float = 2.0
                 ) -> Tuple[Params, Dict[str, np.ndarray]]:
    """testcases/soliton.m:8-89 (BASELINE config 2).  The script has lx=6.12e6,
    ly=3.06e6, dl=20e3 → 307×153; other sizes keep lx, ly and set dl=lx/lm."""
    rhon, nlay, H = 1029.0, 1, 1.0
    lx, ly = 0.5 * 12.24e6, 1.5 * 2.04e6
    dl = 20.0e3 if (lm, mm) == (307, 153) else lx / lm
    a_rd = 6371.0e3
    omeg = 2.0 * np.pi / (24.0 * 3600.0)
    x_0, parB = 0.0, 0.394
    parA = 0.772 * parB ** 2
    Elam = 4.0 * omeg ** 2 * a_rd ** 2 / GRAV / H
    L_ls = a_rd / Elam ** 0.25
    h_bo = np.zeros((lm + 2, mm + 2))
    h_bo[1:-1, 1:-1] = H
    ndeg = get_nbr_deg_freedom(h_bo)
    xx, yy = np.meshgrid((np.arange(1, lm + 3) - 1.5) * dl, (np.arange(1, mm + 3) - 1.5) * dl,
                         indexing="ij")
    xx = xx - xx.mean()
    yy = yy - yy.mean()
    sech2 = 1.0 / np.cosh(parB * (xx - x_0) / L_ls) ** 2
    ex = np.exp(-yy ** 2 / (2.0 * L_ls ** 2))
    n = parA * H * sech2 * (6.0 * yy ** 2 + 3.0 * L_ls ** 2) / (4.0 * L_ls ** 2) * ex
    u = parA * np.sqrt(GRAV * H) * sech2 * (6.0 * yy ** 2 - 9.0 * L_ls ** 2) / (4.0 * L_ls ** 2) * ex
    v = (-2.0 * parA * parB * np.sqrt(GRAV * H) * np.tanh(parB * (xx - x_0) / L_ls) * sech2
         * 2.0 * yy / L_ls * ex)
    cext = np.sqrt(GRAV * (h_bo + n).max())
    deld = 0.25 * ly / 40.0e6
    beta = 2.0 * omeg * np.sin(np.deg2rad(deld)) - 2.0 * omeg * np.sin(np.deg2rad(-deld))
    beta = beta / (2.0 * deld * 40.0e6 / 360.0)
    fcor = beta * yy
    init = np.stack([n[:, :, None], u[:, :, None], v[:, :, None]], axis=3)
    p = make_params(lm, mm, nlay, ndeg, dl, cext, 0.0, [rhon], [0.0], dt_s, dt_o, 0.0, 0.0, 0.0,
                    0.0, 0.0, 0.05, 10.0, 10.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0,
                    desc="Test-case for equatorial soliton")
    return p, {"h_bo": h_bo, "fcor": fcor, "init": init}


def _ndeg_y_uniform(col: np.ndarray, mm: int) -> int:
    """get_nbr_deg_freedom of a frame whose interior rows 1..mm all hold the depth profile `col`
    (margin rows dry): every packed row j = 1..mm+1 then has the same number of cells."""
    f = np.zeros((col.size, 5))
    f[:, 1:4] = np.asarray(col, dtype=np.float64)[:, None]
    return get_nbr_deg_freedom(f) // 4 * (mm + 1)


class Recipe:
    """A testcase recipe that can be evaluated on any band of frame rows (multi-GPU: every rank
    builds its own rows only, beom_amd/slab.py).  ``p`` = the global frame's parameters;
    ``rows(ja, jb)`` = the input arrays restricted to frame rows ja..jb (0 <= ja <= jb <= mm+1,
    inclusive; second axis) — element for element what the whole-frame arrays hold there."""

    def __init__(self, p: Params, rows_fn, keys):
        self.p, self._rows, self.keys = p, rows_fn, tuple(keys)

    def rows(self, ja: int, jb: int) -> Dict[str, np.ndarray]:
        assert 0 <= ja <= jb <= self.p.mm + 1, (ja, jb)
        return self._rows(ja, jb)

    def whole(self) -> Tuple[Params, Dict[str, np.ndarray]]:
        return self.p, self.rows(0, self.p.mm + 1)


def recipe_unstable_jet(lm: int = 201, mm: int = 267, nlay: int = 1, dt_s: float = 50.0,
                        dt_o: float = 2.0) -> Recipe:
    """testcases/unstable_jet.m:10-66 (BASELINE config 3).  Script: 1 layer, dl=15e3,
    lx=3000e3, ly=4000e3 → 201×267.  Other sizes keep lx and set dl=lx/lm.  For
    nlay=2 (BASELINE asks for the multi-layer path; the script itself is 1-layer) the
    column is split at topl=[0,0.5] with rhon=[1029,1030]; the jet is a surface
    anomaly carried by every interface in proportion to its depth fraction."""
    hshf = 5.0
    lx = 3000.0e3
    dl = 15.0e3 if (lm, mm) == (201, 267) else lx / lm
    fcor = 0.5e-4
    xv = np.arange(1, lm + 3) - 1.5
    yv = np.arange(1, mm + 3) - 1.5
    # the script subtracts the mean of the 2-D meshgrid arrays: keep that summation (last bits)
    xm = np.broadcast_to(xv[:, None], (lm + 2, mm + 2)).mean()
    ym = np.broadcast_to(yv[None, :], (lm + 2, mm + 2)).mean()
    xx = (xv - xm)[:, None]
    hcol = hshf + 0.1 * hshf * np.cos(4.0 * np.pi * xx / lm)
    hcol[hcol < 1.0] = 0.0
    hcol[0] = 0.0; hcol[-1] = 0.0
    if nlay > 1:
        hcol = np.where(hcol > 0, hshf, 0.0)       # layered split needs flat bottom (ocrp=0 check :137-144)
    cext = np.sqrt(GRAV * hcol.max())
    if nlay == 1:
        rhon, topl = [1030.0], [0.0]
    else:
        rhon = [1029.0 + k for k in range(nlay)]
        topl = [k / nlay for k in range(nlay)]

    def n0(ja, jb):                                 # surface anomaly on frame rows ja..jb
        yy = (yv[ja:jb + 1] - ym)[None, :]
        return 1.0 * np.exp(-yy ** 2 / (0.1 * mm) ** 2) * np.ones((lm + 2, 1))

    def rows(ja, jb):
        nj = jb - ja + 1
        h_bo = np.repeat(hcol, nj, axis=1)
        for j in (0, mm + 1):
            if ja <= j <= jb:
                h_bo[:, j - ja] = 0.0
        n = np.zeros((lm + 2, nj, nlay)); u = np.zeros((lm + 2, nj, nlay)); v = np.zeros((lm + 2, nj, nlay))
        n[:, :, 0] = n0(ja, jb)
        ka, kb = max(ja, 1), min(jb, mm)            # rows 1..mm carry the geostrophic u
        if kb >= ka:
            ne = n0(ka - 1, kb + 1)
            u[:, ka - ja:kb - ja + 1, 0] = (ne[:, 2:] - ne[:, :-2]) / (2.0 * dl) * GRAV / abs(fcor) * (-1.0)
        for k in range(1, nlay):
            n[:, :, k] = n[:, :, 0] * (1.0 - topl[k])
            u[:, :, k] = u[:, :, 0]
        return {"h_bo": h_bo, "init": np.stack([n, u, v], axis=3)}

    ndeg = _ndeg_y_uniform(hcol[:, 0], mm)
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s, dt_o, 0.0, 0.0, 0.0,
                    0.2, 0.0, 0.1, 10.0, 10.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0,
                    desc="Test-case for barotropic instability")
    return Recipe(p, rows, ("h_bo", "init"))


def case_unstable_jet(lm: int = 201, mm: int = 267, nlay: int = 1, dt_s: float = 50.0,
                      dt_o: float = 2.0) -> Tuple[Params, Dict[str, np.ndarray]]:
    return recipe_unstable_jet(lm, mm, nlay, dt_s, dt_o).whole()


def recipe_sill_exchange3d(lm: int = 125, mm: int = 501, nlay: int = 2, dt_s: float = 30.0,
                           dt_o: float = 0.01, npts: int = 15, sill_halfwidth: float = 50.0) -> Recipe:
    """testcases/sill_exchange3D.m:6-155 (BASELINE config 4).  Script: 2 layers,
    dl=400, 125×501, rhon=[1027.47,1027.75], topl=[0,0.1428].  For nlay>2 extra
    interfaces are inserted (rhon linear, topl = 0.1428·k), the initial anomaly goes
    on the deepest interface as in the script."""
    hmax, hsill = 700.0, 400.0
    fcor = 0.00014087
    dl = 400.0
    if nlay == 2:
        rhon, topl = [1027.47, 1027.75], [0.0, 0.1428]
    else:
        rhon = list(np.linspace(1027.47, 1027.75, nlay))
        topl = [0.1428 * k for k in range(nlay)]
    yv = (np.arange(1, mm + 3) - 1.5) * dl
    ym = np.broadcast_to(yv[None, :], (lm + 2, mm + 2)).mean()     # the script's mean of the 2-D meshgrid array
    yv = yv - ym
    hrow = hmax - hsill * np.exp(-(yv / (sill_halfwidth * dl)) ** 2)    # depth depends on y only
    ndeg = get_nbr_deg_freedom(np.repeat(hrow[None, :], 3, axis=0)) // 2 * (lm + 1)   # every row is wet across: (lm+1) cells per packed row
    cext = np.sqrt(GRAV * hrow.max())
    hsal = 5.0
    hmin = hsal / 10.0
    j0 = int(np.floor(0.6 * (mm + 2) + 0.5))          # Octave round()
    dt = 0.5 * dl / cext
    widt = npts * dl

    def rows(ja, jb):
        nj = jb - ja + 1
        h_bo = np.repeat(hrow[None, ja:jb + 1], lm + 2, axis=0)
        n = np.zeros((lm + 2, nj, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
        ka = max(ja, j0)
        if jb >= ka:
            n[:, ka - ja:, nlay - 1] = np.minimum(-h_bo[:, ka - ja:] + hmax - 100.0 + 4.0 * hsal, 0.0)
        nort = np.zeros((lm + 2, nj, 3)); sout = np.zeros_like(nort)
        for j in range(ja + 1, jb + 2):
            xpos = min(max(j - 1.5 + npts - mm, 0.0), npts - 0.5)
            nort[:, j - 1 - ja, 1:3] = dt * cext / widt * xpos / (npts - xpos)
            nort[:, j - 1 - ja, 0] = dt / (31.0 * 24.0 * 3600.0) * xpos / npts
            xpos = min(max(npts - (j - 1.5), 0.0), npts - 0.5)
            sout[:, j - 1 - ja, 1:3] = dt * cext / widt * xpos / (npts - xpos)
            sout[:, j - 1 - ja, 0] = dt / (31.0 * 24.0 * 3600.0) * xpos / npts
        nudg = np.maximum(nort, sout)
        nudg[0, :, :] = 0.0
        nudg[-1, :, :] = 0.0
        return {"init": np.stack([n, u, v], axis=3), "h_bo": h_bo, "nudg": nudg}

    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s, dt_o, 0.0, 0.0, 0.0,
                    0.9, 0.0, hmin, 5.0, 5.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case: 3D sill exchange")
    return Recipe(p, rows, ("init", "h_bo", "nudg"))


def case_sill_exchange3d(lm: int = 125, mm: int = 501, nlay: int = 2, dt_s: float = 30.0,
                         dt_o: float = 0.01, npts: int = 15, sill_halfwidth: float = 50.0
                         ) -> Tuple[Params, Dict[str, np.ndarray]]:
    return recipe_sill_exchange3d(lm, mm, nlay, dt_s, dt_o, npts, sill_halfwidth).whole()


def recipe_carrier_beach(lm: int | None = None, mm: int = 1, nlay: int = 1, dt_s: float = 0.08,
                         dt_o: float = 5.0e-4, npts: int = 15) -> Recipe:
    """testcases/carrier_beach.m:11-167 + its print_params call (wetting/drying,
    ocrp=1; the recipe BASELINE config 5 extends in y).  For mm>1 the 1-D profile is
    replicated along y.  For nlay>1 the column is split evenly in density."""
    alph, l_0, epsi = 1.0e-3, 3.0e3, 0.1
    dl = l_0 / 50.0
    hhti = 3.0 * epsi * alph * l_0
    l_x = 10.0 * l_0 + hhti / alph
    lm0 = int(np.floor(l_x / dl + 0.5))
    if lm is None:
        lm = lm0
    else:
        dl = l_x / lm
    hsal = 0.2 * alph * dl
    hmin = hsal / 10.0
    p_ = 1.0 / 8.0 / (1.0 + epsi)
    prof = np.arange(lm + 1, -1, -1, dtype=np.float64) * dl * alph
    col = prof.copy()                               # one interior row of h_bo
    col[:npts] = col[npts - 1]
    col[-1] = 0.0; col[0] = 0.0
    ndeg = _ndeg_y_uniform(col, mm)
    d = np.abs(col - hhti)
    i_sl = int(np.where(d == d.min())[0][0])
    hhti = col[i_sl]
    topl0 = hhti / col.max()
    cext = np.sqrt(GRAV * col.max())
    xref = np.arange(1, lm + 3, dtype=np.float64) * dl
    xref = xref - xref[i_sl]
    sigm = np.arange(10.0, -1e-9, -0.1)
    x = 0.25 * epsi * np.exp(2) * p_ ** 2 * sigm ** 4 * np.exp(-sigm ** 2 * p_) - sigm ** 2 / 16.0
    eta = 0.25 * epsi * p_ ** 2 * np.exp(2) * sigm ** 4 * np.exp(-sigm ** 2 * p_)
    xs, es = x * l_0, eta * alph * l_0
    order = np.argsort(xs)
    n1 = np.interp(xref, xs[order], es[order], left=np.nan, right=np.nan)
    n1[np.isnan(n1)] = 0.0
    # western sponge on eta,u (carrier_beach.m:121-147)
    ncol = np.zeros((lm + 2, 3))
    dt = 0.5 * dl / cext
    widt = npts * dl
    for i in range(1, lm + 3):
        xpos = min(max(npts - (i - 1.5), 0.0), npts - 0.5)
        ncol[i - 1, 0:2] = dt * cext / widt * xpos / (npts - xpos)
    if nlay == 1:
        rhon, topl = [1030.0], [topl0]
    else:
        rhon = [1030.0 + 0.5 * k for k in range(nlay)]
        topl = [topl0 + (1.0 - topl0) * k / nlay for k in range(nlay)]

    def rows(ja, jb):
        nj = jb - ja + 1
        h_bo = np.repeat(col[:, None], nj, axis=1)
        for j in (0, mm + 1):
            if ja <= j <= jb:
                h_bo[:, j - ja] = 0.0
        n = np.zeros((lm + 2, nj, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
        n[:, :, 0] = n1[:, None]
        nudg = np.repeat(ncol[:, None, :], nj, axis=1)
        return {"h_bo": h_bo, "init": np.stack([n, u, v], axis=3), "nudg": nudg}

    p = make_params(lm, mm, nlay, ndeg, dl, cext, 0.0, rhon, topl, dt_s, dt_o, 0.0, 0.0, 0.0,
                    0.0, 0.0, hmin, 10.0, 10.0, 1.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for wave on sloping beach")
    return Recipe(p, rows, ("h_bo", "init", "nudg"))


def case_carrier_beach(lm: int | None = None, mm: int = 1, nlay: int = 1, dt_s: float = 0.08,
                       dt_o: float = 5.0e-4, npts: int = 15) -> Tuple[Params, Dict[str, np.ndarray]]:
    return recipe_carrier_beach(lm, mm, nlay, dt_s, dt_o, npts).whole()


def recipe_headline(lm: int = 4096, mm: int = 4096, nlay: int = 4, dvis: float = 0.2,
                    nsteps_days: float | None = None) -> Recipe:
    """SURVEY.md §8(d) headline: closed flat 4000 m basin, dl=1000, f0=1e-4,
    rhon=1026+0.5k, topl=k/nlay, 1 m Gaussian surface mound of radius lm/12
    (as conservation.m:75), g_fb=1, uadv=1, dvis=0.2, no forcing."""
    dl, hfla, f0 = 1000.0, 4000.0, 1.0e-4
    ndeg = (lm + 1) * (mm + 1)
    cext = np.sqrt(GRAV * hfla)
    rhon = [1026.0 + 0.5 * k for k in range(nlay)]
    topl = [k / nlay for k in range(nlay)]
    x = (np.arange(lm + 2, dtype=np.float64) - 0.5 * (lm + 1))[:, None]
    yall = np.arange(mm + 2, dtype=np.float64) - 0.5 * (mm + 1)
    rad = lm / 12.0

    def rows(ja, jb):
        nj = jb - ja + 1
        h_bo = np.zeros((lm + 2, nj), dtype=np.float32)
        ka, kb = max(ja, 1), min(jb, mm)
        if kb >= ka:
            h_bo[1:-1, ka - ja:kb - ja + 1] = hfla
        y = yall[None, ja:jb + 1]
        mound = np.exp(-(x * x + y * y) / rad ** 2).astype(np.float32)
        init = np.zeros((lm + 2, nj, nlay, 3), dtype=np.float32)
        for k in range(nlay):
            init[:, :, k, 0] = mound * np.float32(1.0 - topl[k])
        return {"h_bo": h_bo, "init": init}

    dt = 0.5 * dl / cext
    dt_s = nsteps_days if nsteps_days is not None else 110.0 * dt / 86400.0
    p = make_params(lm, mm, nlay, ndeg, dl, cext, f0, rhon, topl, dt_s, 1.0e3, 0.0, 0.0, 0.0,
                    dvis, 0.0, 1.0, 10.0, 10.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Headline closed basin")
    return Recipe(p, rows, ("h_bo", "init"))


def case_headline(lm: int = 4096, mm: int = 4096, nlay: int = 4, dvis: float = 0.2,
                  nsteps_days: float | None = None) -> Tuple[Params, Dict[str, np.ndarray]]:
    return recipe_headline(lm, mm, nlay, dvis, nsteps_days).whole()


# ---- further recipes of testcases/*.m (input halves), sizes reducible for fixtures -----------------

def case_upwelling_seaward_wind(lm: int = 200, mm: int = 1, dt_s: float = 6.0, dt_o: float = 0.16667,
                                dt_r: float = 4.0) -> Tuple[Params, Dict[str, np.ndarray]]:
    """upwelling_seaward_wind.m:10-37 — flat 40 m channel, two layers, constant seaward wind
    ``tauw = [0.1, 0]`` ramped over dt_r days, periodic in y, no input file but the parameters."""
    nlay, dl, f0, hfla = 2, 1.0e3, 1.0e-4, 40.0
    h_bo = np.zeros((lm + 2, mm + 2))
    h_bo[1:-1, 1:-1] = hfla
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    p = make_params(lm, mm, nlay, ndeg, dl, cext, f0, [1028.95, 1030.0], [0.0, 0.5], dt_s,
                    dt_o, dt_r, 0.0, 0.0, 0.0, 0.0, 1.0, 10.0, 10.0, 1.0,
                    0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, tauw=(0.1, 0.0),
                    desc="Test-case for upwelling seaward wind")
    return p, {}


def case_lock_exchange(lx: float = 64.0e3, mm: int = 1, dt_s: float = 5.0, dt_o: float = 1.0 / 24.0
                       ) -> Tuple[Params, Dict[str, np.ndarray]]:
    """lock_exchange.m:11-64 — closed 20 m flume, two layers, the interface 4 hsal below the surface
    in the left half and 4 hsal above the bottom in the right half; f0 = 0, dvis = 0.03."""
    hmax, nlay, dl, hmin = 20.0, 2, 400.0, 0.005
    hsal = 10.0 * hmin
    lm = int(round(lx / dl))
    h_bo = hmax * np.ones((lm + 2, mm + 2))
    h_bo[:, 0] = 0.0; h_bo[:, -1] = 0.0; h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    n = np.zeros((lm + 2, mm + 2, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
    half = int(round(0.5 * (lm + 2)))                     # Octave round(): half away from zero
    n[:half, :, 1] = 0.5 * hmax - 4.0 * hsal
    n[half:, :, 1] = -0.5 * hmax + 4.0 * hsal
    p = make_params(lm, mm, nlay, ndeg, dl, cext, 0.0, [1025.0, 1030.0], [0.0, 0.5], dt_s,
                    dt_o, 0.0, 0.0, 0.0, 0.03, 0.0, hmin, 10.0, 10.0, 1.0,
                    1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for lock-exchange")
    return p, {"init": np.stack([n, u, v], axis=3)}


def case_morel_upwelling(ly_in_rext: float = 1.0, lm: int = 1) -> Tuple[Params, Dict[str, np.ndarray]]:
    """morel_upwelling.m:14-65 — x-periodic strip, coast at y_max, two layers with outcropping
    (ocrp = 1), the wind as a body force on the top layer (bodf.bin), diag = 1."""
    dl, nlay, hfla, hsal, f0 = 1.0e3, 2, 50.0, 0.5, 1.0e-4
    topl, rhon = [0.0, 0.5], [1015.0, 1030.0]
    rext = np.sqrt(GRAV * hfla) / abs(f0)
    mm = int(round(ly_in_rext * rext / dl))
    h_bo = np.zeros((lm + 2, mm + 2))
    h_bo[1:-1, 1:-1] = hfla
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    hmin = hsal / 10.0
    h_1, h_2 = topl[1] * hfla, (1.0 - topl[1]) * hfla
    r_d = np.sqrt(GRAV * (rhon[1] - rhon[0]) / rhon[1] * h_1 * h_2 / hfla) / abs(f0)
    delt = h_1 / h_2
    t_w = 0.05 / rhon[0] / h_1
    t_o = abs(f0) * r_d * (1.0 + delt) / t_w
    dt_s = float(np.floor(3.0 * t_o / 3600.0 / 24.0 + 0.5))
    bodf = np.zeros((nlay, 2)); bodf[0, 0] = t_w
    p = make_params(lm, mm, nlay, ndeg, dl, cext, f0, rhon, topl, dt_s,
                    0.05 * dt_s, 0.0, 0.0, 0.0, 0.0, 0.0, hmin, 10.0, 10.0, 1.0,
                    0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0,
                    desc="Test-case: Upwelling in presence of outcrop")
    return p, {"bodf": bodf}                     # no h_bo.bin: the flat depth comes from cext (:105-160)


def case_outcrop_seamount(lx: float = 600.0e3, dl: float = 5.0e3, nlay: int = 5, three_d: bool = False,
                          dt_s: float = 15.0, dt_o: float = 0.2) -> Tuple[Params, Dict[str, np.ndarray]]:
    """outcrop_seamount.m:9-105 — closed basin deepening from the rim to 300 m with a Gaussian
    seamount; up to five layers outcrop on the slopes (ocrp = 1); a state of rest that must stay so.
    three_d = False is the script's x-z plane (mm = 1)."""
    fcor, hmax = 1.0e-4, 300.0
    if nlay == 1:
        rhon, topl = [1000.0], [0.0]
    else:
        rhon = [1000.0 + 30.0 * k / (nlay - 1) for k in range(nlay)]
        topl = [k / nlay for k in range(nlay)]
    lm = int(np.floor(lx / dl + 0.5))
    if lm % 2 == 0:
        lm += 1
    mm = lm if three_d else 1
    xx = np.arange(1, lm + 3, dtype=np.float64)[:, None] * np.ones((1, mm + 2))
    yy = np.ones((lm + 2, 1)) * np.arange(1, mm + 3, dtype=np.float64)[None, :]
    xx = (xx - xx.mean()) * dl
    yy = (yy - yy.mean()) * dl
    ix_0 = int(np.ceil(0.5 * (mm + 2))) - 1                   # centre column (0-based)
    h_bo = np.sqrt(xx ** 2 + yy ** 2) / dl
    h_bo = h_bo / h_bo[:, ix_0].max() * hmax
    h_bo = h_bo[:, ix_0].max() - h_bo
    smnt = hmax - np.exp(-(xx ** 2 + yy ** 2) / (50.0e3) ** 2) * 0.75 * hmax
    h_bo = np.minimum(h_bo, smnt)
    h_bo[h_bo < h_bo[:, ix_0].min()] = 0.0
    dhdx = np.zeros_like(h_bo); dhdy = np.zeros_like(h_bo)
    dhdx[1:-1, :] = (h_bo[2:, :] - h_bo[:-2, :]) / (2.0 * dl)
    dhdy[:, 1:-1] = (h_bo[:, 2:] - h_bo[:, :-2]) / (2.0 * dl)
    slop = np.sqrt(dhdx ** 2 + dhdy ** 2)
    hsal = 1.0 * slop[h_bo > 1.0e-3].max() * dl * (rhon[1] - rhon[0]) / rhon[1] if nlay > 1 else 1.0
    hmin = hsal / 10.0
    dryd = 10.0 * hsal + (nlay - 1) * hsal
    h_bo[h_bo < dryd] = 0.0
    h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0; h_bo[:, -1] = 0.0; h_bo[:, 0] = 0.0
    cext = np.sqrt(GRAV * h_bo.max())
    ndeg = get_nbr_deg_freedom(h_bo)
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s,
                    dt_o, 0.0, 0.0, 0.0, 0.0, 0.0, hmin, 10.0, 10.0, 1.0,
                    1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for state of rest allowing isopycnal outcrop")
    return p, {"h_bo": h_bo}


def _frs_coefficients(n: int, extent: int, npts: int, dt: float, cext: float, dl: float, high_side: bool) -> np.ndarray:
    """Flow-relaxation coefficients along one axis (index 1..n+2 of the frame), as every sponge
    recipe computes them: ``dt·cext/widt · xpos/(npts − xpos)`` with xpos clipped to [0, npts − ½]
    (Modave et al. 2010, Eq. 29; wave_sponge.m:60-88).  high_side: the E or N boundary."""
    widt = npts * dl
    out = np.zeros(n + 2)
    for i in range(1, n + 3):
        xpos = (i - 1.5 + npts - extent) if high_side else (npts - (i - 1.5))
        xpos = min(max(xpos, 0.0), npts - 0.5)
        out[i - 1] = dt * cext / widt * xpos / (npts - xpos)
    return out


def case_wave_sponge(lx: float = 600.0e3, ly: float = 600.0e3, dl: float = 10.0e3, npts: int = 15,
                     dt_s: float = 0.25, dt_o: float = 4.17e-3) -> Tuple[Params, Dict[str, np.ndarray]]:
    """wave_sponge.m:11-124 — flat two-layer basin, a 1 m Gaussian mound radiating into flow-relaxation
    sponges on all four sides (eta and the normal velocity are relaxed on each side)."""
    nlay, fcor, hfla = 2, 1.0e-4, 200.0
    lm = int(np.floor(lx / dl + 0.5)); mm = int(np.floor(ly / dl + 0.5))
    if lm % 2 == 0: lm += 1
    if mm % 2 == 0: mm += 1
    lm += 2 * npts; mm += 2 * npts
    h_bo = np.zeros((lm + 2, mm + 2)); h_bo[1:-1, 1:-1] = hfla
    cext = np.sqrt(GRAV * hfla)
    ndeg = get_nbr_deg_freedom(h_bo)
    xi = np.arange(1, lm + 3) - 1.5; yj = np.arange(1, mm + 3) - 1.5
    xx = ((xi - np.median(xi)) * dl)[:, None] * np.ones((1, mm + 2))
    yy = np.ones((lm + 2, 1)) * ((yj - np.median(yj)) * dl)[None, :]
    n = np.zeros((lm + 2, mm + 2, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
    n[:, :, 0] = 1.0 * np.exp(-(xx ** 2 + yy ** 2) / (50.0e3) ** 2)
    dt = 0.5 * dl / cext
    ce = _frs_coefficients(lm, lm, npts, dt, cext, dl, True)[:, None]
    cw = _frs_coefficients(lm, lm, npts, dt, cext, dl, False)[:, None]
    cn = _frs_coefficients(mm, mm, npts, dt, cext, dl, True)[None, :]
    cs = _frs_coefficients(mm, mm, npts, dt, cext, dl, False)[None, :]
    one = np.ones((lm + 2, mm + 2))
    nudg = np.zeros((lm + 2, mm + 2, 3))
    nudg[:, :, 0] = np.maximum.reduce([ce * one, cw * one, cn * one, cs * one])   # eta: every side
    nudg[:, :, 1] = np.maximum(ce * one, cw * one)                                # u: E and W
    nudg[:, :, 2] = np.maximum(cn * one, cs * one)                                # v: N and S
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, [1000.0, 1030.0], [0.0, 0.5], dt_s,
                    dt_o, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 10.0, 10.0, 1.0,
                    1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for wave sponge")
    return p, {"init": np.stack([n, u, v], axis=3), "nudg": nudg}


def case_tide_ridge(lm: int = 500, ocrp: int = 1, dt_s: float = 3.0, dt_o: float = 0.01, npts: int = 15,
                    ridge_halfwidth: float = 75.0) -> Tuple[Params, Dict[str, np.ndarray]]:
    """tide_ridge.m:6-156 — semidiurnal (M2) barotropic current over a Gaussian ridge in a 30 m deep
    x-z plane, seven outcropping layers (three without ocrp) from an exponential density profile,
    E/W sponges carrying the tidal u, quadratic drag switch, dvis = 0.5."""
    hmax, bdrg, fcor, dl, mm = 30.0, 0.0, 0.0, 100.0, 1
    dt_r = 0.75 * dt_s
    nlay = 7 if ocrp == 1 else 3
    if lm % 2 == 0: lm += 1
    z = np.arange(0.0, hmax + 0.5, 1.0)
    rhop = 1028.0 - 6.0 * np.exp(-z / (0.2 * hmax))
    drho = (rhop.max() - rhop.min()) / nlay
    rhon = [float(rhop.min() + (k + 0.5) * drho) for k in range(nlay)]
    xi = (np.arange(1, lm + 3) - 1.5) * dl
    xx = (xi - np.mean(np.repeat(xi, mm + 2)))[:, None] * np.ones((1, mm + 2))
    h_bo = hmax - 0.75 * hmax * np.exp(-(xx / (ridge_halfwidth * dl)) ** 2)
    h_bo[:, 0] = 0.0; h_bo[:, -1] = 0.0; h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    dhdx = np.zeros((lm + 2, mm + 2))
    dhdx[2:-2, :] = h_bo[3:-1, :] - h_bo[1:-3, :]
    dhdx = dhdx / (2.0 * dl)
    hsal = 20.0 * dhdx.max() * dl * (rhon[-1] - rhon[0]) / rhon[0]
    hmin = hsal / 10.0
    topl = np.interp(rhop.min() + np.arange(nlay) * drho, rhop, z)
    if topl[1] < 10.0 * hsal:
        topl[1:] = topl[1:] + (10.0 * hsal - topl[1])
    topl = topl / hmax
    tide = np.zeros((2, 1, lm + 2, mm + 2, 3))
    tide[0, 0, :, :, 1] = 0.1
    tide[1, 0, :, :, 1] = np.pi / 2.0
    tide[1, 0, :, :, 2] = np.pi / 2.0
    tide[0, 0, 0, 0, 0] = 2.0 * np.pi / (12.4206012 / 24.0)       # M2, rad/day, stored in the first element
    dt = 0.5 * dl / cext
    ce = _frs_coefficients(lm, lm, npts, dt, cext, dl, True)[:, None] * np.ones((1, mm + 2))
    cw = _frs_coefficients(lm, lm, npts, dt, cext, dl, False)[:, None] * np.ones((1, mm + 2))
    nudg = np.zeros((lm + 2, mm + 2, 3))
    nudg[:, :, 0] = np.maximum(ce, cw); nudg[:, :, 1] = np.maximum(ce, cw)
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, [float(t) for t in topl], dt_s,
                    dt_o, dt_r, 0.0, 0.0, 0.5, bdrg, hmin, 5.0, 5.0, 1.0,
                    1.0, 1.0, float(ocrp), 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for tidal flow over a ridge")
    return p, {"h_bo": h_bo, "nudg": nudg, "tide": tide}


def case_baines_ridge(domain_in_lros: float = 150.0, npts: int = 15, dt_s: float = 10.0, dt_o: float = 0.2
                      ) -> Tuple[Params, Dict[str, np.ndarray]]:
    """baines_ridge.m:4-163 — uniform 1.2 m/s two-layer flow over a cosine ridge (Baines & Leonard
    1989), y-periodic with mm = 1, the cross-stream pressure gradient as a body force, E/W sponges."""
    hmax, u_0, nlay, fcor = 110.0, 1.2, 2, 1.0e-4
    rhon, topl = [1025.0, 1030.0], [0.0, 1.0 / 1.1]
    gp = GRAV * (rhon[1] - rhon[0]) / rhon[1]
    d_0 = hmax * (1.0 - topl[1])
    lros = np.sqrt(gp * d_0) / abs(fcor)
    dl = lros / 5.0
    lm = int(np.floor(domain_in_lros * lros / dl + 0.5))
    if lm % 2 == 0: lm += 1
    mm = 1
    xi = (np.arange(1, lm + 3) - 1.5) * dl
    xx = (xi - np.mean(np.repeat(xi, mm + 2)))[:, None] * np.ones((1, mm + 2))
    h_bo = 0.1 * d_0 * np.cos(np.pi * xx / (10.0 * lros))
    h_bo[(xx < -5.0 * lros) | (xx > 5.0 * lros)] = 0.0
    h_bo = hmax - h_bo
    h_bo[:, 0] = 0.0; h_bo[:, -1] = 0.0; h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    n = np.zeros((lm + 2, mm + 2, nlay)); u = np.ones_like(n) * u_0; v = np.zeros_like(n)
    bodf = np.zeros((nlay, 2)); bodf[:, 1] = fcor * u_0
    dt = 0.5 * dl / cext
    ce = _frs_coefficients(lm, lm, npts, dt, cext, dl, True)[:, None] * np.ones((1, mm + 2))
    cw = _frs_coefficients(lm, lm, npts, dt, cext, dl, False)[:, None] * np.ones((1, mm + 2))
    nudg = np.zeros((lm + 2, mm + 2, 3))
    nudg[:, :, 0] = np.maximum(ce, cw); nudg[:, :, 1] = np.maximum(ce, cw)
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s,
                    dt_o, 0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 10.0, 10.0, 1.0,
                    1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0,
                    desc="Test-case for flow over a ridge")
    return p, {"h_bo": h_bo, "init": np.stack([n, u, v], axis=3), "nudg": nudg, "bodf": bodf}


def case_mixed_open_bc(lm: int = 200, mm: int = 100, npts: int = 15, dt_s: float = 6.0, dt_o: float = 0.16667,
                       dt_r: float = 4.0) -> Tuple[Params, Dict[str, np.ndarray]]:
    """mixed_open_bc.m:4-113 — the seaward-wind upwelling with a coast in the west and open boundaries
    elsewhere: flow relaxation of eta,u (one-month relaxation of v) in the east, one-month relaxation
    of everything in the north and south."""
    nlay, dl, f0, hfla = 2, 1.0e3, 1.0e-4, 40.0
    h_bo = np.zeros((lm + 2, mm + 2)); h_bo[1:-1, 1:-1] = hfla
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    dt = 0.5 * dl / cext
    month = 31.0 * 24.0 * 3600.0
    def slow(n, extent, high):
        out = np.zeros(n + 2)
        for i in range(1, n + 3):
            xpos = (i - 1.5 + npts - extent) if high else (npts - (i - 1.5))
            out[i - 1] = dt / month * min(max(xpos, 0.0), npts - 0.5) / npts
        return out
    one = np.ones((lm + 2, mm + 2))
    ce = _frs_coefficients(lm, lm, npts, dt, cext, dl, True)[:, None] * one
    se = slow(lm, lm, True)[:, None] * one
    sn = slow(mm, mm, True)[None, :] * one
    ss = slow(mm, mm, False)[None, :] * one
    nudg = np.zeros((lm + 2, mm + 2, 3))
    nudg[:, :, 0] = np.maximum.reduce([ce, sn, ss])
    nudg[:, :, 1] = np.maximum.reduce([ce, sn, ss])
    nudg[:, :, 2] = np.maximum.reduce([se, sn, ss])
    nudg[0, :, :] = 0.0                                       # the western edge is the coast
    p = make_params(lm, mm, nlay, ndeg, dl, cext, f0, [1028.95, 1030.0], [0.0, 0.5], dt_s,
                    dt_o, dt_r, 0.0, 0.0, 0.0, 0.0, 0.001, 1.0, 1.0, 1.0,
                    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, tauw=(0.1, 0.0),
                    desc="Test-case: Upwelling seaward wind with mixed open boundary conditions")
    return p, {"nudg": nudg}


def case_conservation(lx: float = 600.0e3, dl: float = 10.0e3, nlay: int = 2, outc: int = 0, topo: bool = True,
                      xper: int = 1, yper: int = 1, dt_s: float = 50.0, dt_o: float = 0.5
                      ) -> Tuple[Params, Dict[str, np.ndarray]]:
    """conservation.m:4-98 — doubly periodic (by default) basin with a Gaussian seamount and a 1 m
    surface mound of radius lx/12; standard forward-backward stepping (g_fb = 0), diag = 1;
    outc = 1 lets the seamount pierce the deepest interfaces."""
    fcor, hfla = 1.0e-4, 200.0
    rhon = [1000.0 + 30.0 * (k + 1) / nlay for k in range(nlay)]
    topl = [k / nlay for k in range(nlay)]
    lm = int(np.floor(lx / dl + 0.5))
    if lm % 2 == 0: lm += 1
    mm = lm
    xi = np.arange(1, lm + 3) - 1.5; yj = np.arange(1, mm + 3) - 1.5
    xx = ((xi - xi.mean()) * dl)[:, None] * np.ones((1, mm + 2))
    yy = np.ones((lm + 2, 1)) * ((yj - yj.mean()) * dl)[None, :]
    h_bo = hfla * np.ones((lm + 2, mm + 2))
    if topo:
        amp = 0.5 * (2.0 - topl[-1] - topl[-2]) * hfla if outc else 0.5 * (1.0 - topl[-1]) * hfla
        h_bo = hfla - amp * np.exp(-(xx ** 2 + yy ** 2) / (0.25 * lx) ** 2)
    h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0; h_bo[:, 0] = 0.0; h_bo[:, -1] = 0.0
    ndeg = get_nbr_deg_freedom(h_bo)
    n = np.zeros((lm + 2, mm + 2, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
    n[:, :, 0] = 1.0 * np.exp(-(xx ** 2 + yy ** 2) / (lx / 12.0) ** 2)
    cext = np.sqrt(GRAV * h_bo.max())
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s,
                    dt_o, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 10.0, 10.0, 0.0,
                    1.0, 0.0, float(outc), 0.0, float(xper), float(yper), 1.0,
                    desc="Test-case for integral conservation of properties")
    files = {"init": np.stack([n, u, v], axis=3)}
    if topo:
        files["h_bo"] = h_bo
    return p, files


def case_sill_exchange2d(lx: float | None = None, dl: float = 100.0, npts: int | None = None, tides: bool = False,
                         sill_halfwidth: float = 200.0, dt_s: float | None = None
                         ) -> Tuple[Params, Dict[str, np.ndarray]]:
    """sill_exchange2D.m:6-139 and, with tides = True, sill_exchange2Dtides.m:6-165 — the x-z plane
    form of the sill exchange: Gaussian sill of 400 m in 700 m, two outcropping layers, E/W sponges;
    the tidal variant adds an M2 current in the sponges, starts from a different interface and uses
    dvis = 0.5, topl = [0, 0.5]."""
    ocrp, hmax, hsill, fcor, mm, nlay, bdrg, dt_r = 1, 700.0, 400.0, 0.00014087, 1, 2, 0.0, 0.0
    if lx is None: lx = 100.0e3 if tides else 200.0e3
    if npts is None: npts = 15 if tides else 200
    if dt_s is None: dt_s = 10.0 if tides else 3.0
    rhon = [1027.47, 1027.75]
    topl = [0.0, 0.5] if tides else [0.0, 0.1428]
    lm = int(np.floor(lx / dl + 0.5))
    if lm % 2 == 0: lm += 1
    xi = (np.arange(1, lm + 3) - 1.5) * dl
    xx = (xi - np.mean(np.repeat(xi, mm + 2)))[:, None] * np.ones((1, mm + 2))
    h_bo = hmax - hsill * np.exp(-(xx / (sill_halfwidth * dl)) ** 2)
    h_bo[:, 0] = 0.0; h_bo[:, -1] = 0.0; h_bo[0, :] = 0.0; h_bo[-1, :] = 0.0
    ndeg = get_nbr_deg_freedom(h_bo)
    cext = np.sqrt(GRAV * h_bo.max())
    dhdx = np.zeros((lm + 2, mm + 2))
    dhdx[2:-2, :] = h_bo[3:-1, :] - h_bo[1:-3, :]
    dhdx = dhdx / (2.0 * dl)
    hsal = 20.0 * dhdx.max() * dl * (rhon[-1] - rhon[0]) / rhon[0]
    hmin = hsal / 10.0
    n = np.zeros((lm + 2, mm + 2, nlay)); u = np.zeros_like(n); v = np.zeros_like(n)
    if tides:
        half = int(np.floor(0.5 * (lm + 2) + 0.5))
        n[:half, :, 1] = 0.5 * hmax - 4.0 * hsal - 100.0
        right = -0.5 * hmax + 4.0 * hsal + 450.0 + (hmax - h_bo[half:, 1])
        n[half:, :, 1] = np.minimum(right, 0.5 * hmax - 4.0 * hsal - 100.0)[:, None]
    else:
        j0 = int(np.floor(0.6 * (lm + 2) + 0.5))
        n[j0:, :, 1] = (-h_bo[j0:, 1] + 500.0 + 4.0 * hsal)[:, None]
    dt = 0.5 * dl / cext
    ce = _frs_coefficients(lm, lm, npts, dt, cext, dl, True)[:, None] * np.ones((1, mm + 2))
    cw = _frs_coefficients(lm, lm, npts, dt, cext, dl, False)[:, None] * np.ones((1, mm + 2))
    nudg = np.zeros((lm + 2, mm + 2, 3))
    nudg[:, :, 0] = np.maximum(ce, cw); nudg[:, :, 1] = np.maximum(ce, cw)
    files = {"init": np.stack([n, u, v], axis=3), "h_bo": h_bo, "nudg": nudg}
    if tides:
        tide = np.zeros((2, 1, lm + 2, mm + 2, 3))
        tide[0, 0, :, :, 1] = 0.1
        tide[1, 0, :, :, 1] = np.pi / 2.0
        tide[1, 0, :, :, 2] = np.pi / 2.0
        tide[0, 0, 0, 0, 0] = 2.0 * np.pi / (12.4206012 / 24.0)
        files["tide"] = tide
    p = make_params(lm, mm, nlay, ndeg, dl, cext, fcor, rhon, topl, dt_s,
                    0.01, dt_r, 0.0, 0.0, 0.5 if tides else 0.9, bdrg, hmin, 5.0, 5.0, 1.0,
                    1.0, 1.0, float(ocrp), 0.0, 0.0, 0.0, 0.0,
                    desc="Test-case for tidal flow over a ridge" if tides else "Test-case: 2D sill exchange")
    return p, files
