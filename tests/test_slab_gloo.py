"""Multi-rank path on CPU: world_size 2 and 3 over gloo.  Each rank runs beom_amd.slab's
SlabRunner (the same decomposition / pack / exchange / unpack code the GPU path uses)
around a CPU adapter of the oracle, and checks that its OWNED rows equal the
single-domain oracle run bit for bit after 12 steps (steps 1-3 plain FB, both U/V
orders, G = 4 ghost rows, one exchange per step).  Frames periodic in y: the bands form a ring
and rank 0 also carries the companion frame that owns the orphan row mm+1 (checked in the sign of
zero too).  The windows are built both ways: cut from the whole frame's Fields, and from the
recipe's rows for the band alone (what bench.py does on N GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in with the engine protocol SlabRunner needs (tests only)."""

    def __init__(self, fields, variant=0):
        import oracle_lib
        self.o = oracle_lib.Oracle(fields, variant=variant)

    def step(self, tstp_first, nsteps, sync=False):
        self.o.step(tstp_first, nsteps)

    def field_tensors(self, names):
        import torch
        return {k: torch.from_numpy(self.o.a[k]) for k in names}

    def sync(self):
        pass


def _case(name):
    """(Params, files) or a Recipe (then the windows are built from the recipe's rows)."""
    from beom_amd import inputs as I
    if name == "closed_3l":
        return I.case_headline(30, 44, 3)
    if name == "closed_3l_recipe":
        return I.recipe_headline(30, 44, 3)
    if name == "sill_ocrp_nudg":
        return I.case_sill_exchange3d(lm=15, mm=47, nlay=2, dt_s=0.01, npts=5, sill_halfwidth=6.0)
    if name == "sill_ocrp_nudg_recipe":
        return I.recipe_sill_exchange3d(lm=15, mm=47, nlay=2, dt_s=0.01, npts=5, sill_halfwidth=6.0)
    if name == "beach_ocrp_recipe":
        return I.recipe_carrier_beach(lm=40, mm=50, nlay=3, dt_s=0.08)
    if name == "soliton_xper":
        return I.case_soliton(lm=31, mm=39, dt_s=5.0)
    if name == "jet_xyper":
        return I.case_unstable_jet(lm=21, mm=47, nlay=2, dt_s=1.0)
    if name == "jet_xyper_recipe":
        return I.recipe_unstable_jet(lm=21, mm=47, nlay=2, dt_s=1.0)
    raise KeyError(name)


def _worker(rank, world, port, case, nsteps):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch
    import torch.distributed as dist
    import oracle_lib
    from beom_amd import slab
    from beom_amd.grid import read_input_data
    from helpers import same, same_bits
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = _case(case)
        recipe = c if hasattr(c, "rows") else None
        p, files = recipe.whole() if recipe else c
        f = read_input_data(p, files=files)          # the whole frame: the reference run (and, without a recipe, the source of the windows)
        yper = float(p.yper) > 0.5
        geom = slab.decompose(p.mm, p.lm, world, yper=yper)[rank]
        mini = None
        if recipe:
            lf, g2, orphan = slab.build_band(recipe, world, rank)
            assert g2 == geom and (orphan is not None) == (yper and rank == 0)
            if orphan is not None:                   # companion frame = this band's rows + the orphan row, as the library builds it
                rows = slab.mini_rows(p.mm)
                mf = slab.slice_mini(f)
                L = geom.L
                for k in ("hlay", "u", "v", "fnud", "fcor", "h_th", "nudg"):
                    assert same(getattr(mf, k)[..., -L:], getattr(orphan, k)[..., 1:]), k
                mini = OracleEngine(mf)
        else:
            lf = slab.slice_fields(f, geom)
            if yper and rank == 0:
                mini = OracleEngine(slab.slice_mini(f))
        run = slab.SlabRunner(OracleEngine(lf), geom, p.nlay, dist=dist, mini=mini)
        run.step(1, nsteps)
        ref = oracle_lib.Oracle(f)
        ref.step(1, nsteps)
        a, b = 1 + (geom.own0 - 1) * geom.L, 1 + geom.own1 * geom.L
        la, lb = geom.local_rows(geom.own0, geom.own1)
        for k in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "v_cc", "v_ll"):
            loc = run.engine.o.a[k]
            if loc.ndim == 3:
                ok = same(loc[:, la:lb, :], ref.a[k][:, a:b, :])
            else:
                ok = same(loc[..., la:lb], ref.a[k][..., a:b])
            assert ok, (case, rank, k)
            if mini is not None:                     # the orphan row mm+1, the sign of zero included
                L = geom.L
                mo = mini.o.a[k]
                ok = same_bits(mo[:, -L:, :], ref.a[k][:, -L:, :]) if mo.ndim == 3 else same_bits(mo[..., -L:], ref.a[k][..., -L:])
                assert ok, (case, "orphan row", k)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,case", [(2, "closed_3l"), (3, "closed_3l_recipe"), (2, "sill_ocrp_nudg"),
                                        (3, "sill_ocrp_nudg_recipe"), (2, "beach_ocrp_recipe"), (2, "soliton_xper"),
                                        (2, "jet_xyper"), (3, "jet_xyper_recipe"), (2, "jet_xyper_recipe")])
def test_slab_runner_matches_single_domain(world, case):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case, 12), nprocs=world, join=True)


def test_decompose_covers_rows_once():
    from beom_amd import slab
    for mm, world in ((4096, 8), (47, 3), (100, 7)):
        gs = slab.decompose(mm, 10, world)
        rows = []
        for g in gs:
            rows += list(range(g.own0, g.own1 + 1))
            assert g.win0 >= 1 and g.win1 <= mm + 1
            assert g.ghost_s == (slab.GHOST if g.rank > 0 else 0)
            assert g.ghost_n == (slab.GHOST if g.rank < world - 1 else 0)
            assert g.global_rows() == list(range(g.win0, g.win1 + 1))
        assert rows == list(range(1, mm + 2))
    # ring (frame periodic in y): rows 1..mm dealt out, every band has ghosts on both sides, ghosts wrap
    for mm, world in ((2048, 8), (47, 3), (60, 1)):
        gs = slab.decompose(mm, 10, world, yper=True)
        rows = []
        for g in gs:
            rows += list(range(g.own0, g.own1 + 1))
            assert g.ghost_s == slab.GHOST and g.ghost_n == slab.GHOST and g.ring
            gr = g.global_rows()
            assert all(1 <= r <= mm for r in gr) and len(gr) == g.rows
            assert g.south == (g.rank - 1) % world and g.north == (g.rank + 1) % world
        assert rows == list(range(1, mm + 1))
        assert gs[0].global_rows()[:slab.GHOST] == list(range(mm - slab.GHOST + 1, mm + 1))
        assert gs[-1].global_rows()[-slab.GHOST:] == list(range(1, slab.GHOST + 1))
        assert sum(b - a + 1 for a, b in gs[0].pieces()) == gs[0].rows


def test_library_window_agrees_with_python_geometry():
    """beom_multi_window (what beom_multi_create_local expects) == beom_amd.slab.decompose."""
    from beom_amd import capi, inputs as I, slab
    for recipe, worlds in ((I.recipe_headline(30, 200, 2), (1, 2, 3, 8)), (I.recipe_unstable_jet(lm=21, mm=147, nlay=2), (1, 2, 5))):
        p = recipe.p
        yper = float(p.yper) > 0.5
        for world in worlds:
            for g in slab.decompose(p.mm, p.lm, world, yper=yper):
                w = capi.multi_window(p, world, g.rank, yper)
                assert (w["own0"], w["own1"], w["ghost_s"], w["ghost_n"]) == (g.own0, g.own1, g.ghost_s, g.ghost_n)


def test_window_builds_need_a_dense_frame():
    from beom_amd import inputs as I, slab
    from beom_amd.grid import read_input_data
    from helpers import Golden
    f = Golden("island_3l_forced").fields()                   # land inside the frame
    with pytest.raises(ValueError):
        slab.slice_fields(f, slab.decompose(f.p.mm, f.p.lm, 1)[0])
