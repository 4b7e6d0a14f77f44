"""CPU: the open-boundary segments (index_boundary_points, private_mod.f95:1060-1240) a rank finds from its own rows of a
recipe (slab.build_band -> grid.read_input_data(window=...)) are exactly the whole frame's table restricted to those rows —
what beom_multi_set_open_boundaries_local hands to the band's engine."""
import numpy as np
import pytest

from beom_amd import inputs as I, slab
from beom_amd.grid import read_input_data


def _with_sponges(base, yper):
    p = base.p
    nudg = np.zeros((p.lm + 2, p.mm + 2, 3))
    for i in range(0, 9):
        nudg[i, :, 0:2] = 0.3 * (9 - i) / 9.0
    for i in range(p.lm + 1, p.lm - 7, -1):
        w = 0.25 * (i - (p.lm - 7)) / 9.0
        nudg[i, :, 0] = np.maximum(nudg[i, :, 0], w); nudg[i, :, 1] = np.maximum(nudg[i, :, 1], w)
    if not yper:
        for j in range(p.mm + 1, p.mm - 7, -1):
            w = 0.2 * (j - (p.mm - 7)) / 9.0
            nudg[:, j, 0] = np.maximum(nudg[:, j, 0], w); nudg[:, j, 2] = np.maximum(nudg[:, j, 2], w)
    return I.Recipe(p.replace(xper="0.", mcbc="0."), lambda ja, jb: dict(base.rows(ja, jb), nudg=nudg[:, ja:jb + 1, :]),
                    tuple(base.keys) + ("nudg",))


def _passes(T, tocell):
    """(pass, updated cell, source cell, east/west flag, north/south flag, sign) of every active pass of a table [18, nseg]"""
    out = set()
    for k in range(T.shape[1]):
        for ps, (cu, cs) in enumerate(((9, 15), (0, 12))):
            if T[cu, k] >= 1:
                out.add((ps, tocell(int(T[cu, k])), tocell(int(T[cs, k])) if T[cs, k] > 0 else 0, int(T[3, k]), int(T[4, k]), int(T[5, k])))
    return out


@pytest.mark.parametrize("case", ["jet_yper", "closed"])
def test_segments_of_a_window_are_the_frame_s_segments_of_its_rows(case):
    yper = case == "jet_yper"
    base = I.recipe_unstable_jet(lm=61, mm=83, nlay=1, dt_s=1.5) if yper else I.recipe_headline(70, 75, 2)
    r = _with_sponges(base, yper)
    p, L = r.p, r.p.lm + 1
    whole = _passes(read_input_data(p, files=r.rows(0, p.mm + 1)).segm, lambda q: q)
    assert whole
    glob = slab.recipe_global_info(r)
    for world in (2, 3):
        seen = set()
        for rank in range(world):
            f, g, orphan = slab.build_band(r, world, rank, glob)
            rows = g.global_rows()
            tocell = lambda q: (q - 1) % L + 1 + (rows[(q - 1) // L] - 1) * L
            mine = _passes(f.segm, tocell) if f.segm is not None else set()
            want = {e for e in whole if ((e[1] - 1) // L + 1) in rows and (e[2] == 0 or ((e[2] - 1) // L + 1) in rows)}
            assert mine == want, (case, world, rank)
            seen |= {e for e in mine if g.own0 <= (e[1] - 1) // L + 1 <= g.own1}
            if orphan is not None and orphan.segm is not None:
                seen |= _passes(orphan.segm, lambda q: (q - 1) % L + 1 + p.mm * L)
        assert seen == whole, (case, world)                # every pass of the frame is some rank's own
