"""The rank-local band path of the library — beom_multi_create_local with nb = 2 and 3, what every rank of
`bench.py --gpus N` runs — as 2 and 3 PROCESSES on the one GPU of the test box.

RCCL refuses two ranks on one device, so the ghost rows travel through the library's shared-memory transport
(BEOM_XCHG_SHM, beom_multi.hip): the same multi_one_step, the same neighbour numbering (south_of / north_of /
has_s / has_n), the same streams and events and split steps as the RCCL branch; only the two calls that move a
packed buffer differ.  Every rank builds its window from the recipe alone (slab.build_band — no array of global
size), steps 13 steps in uneven calls, and its owned rows must equal the single handle's bit for bit; for the
y-periodic jet the bands form a ring and rank 0 also carries the orphan row mm+1 (companion frame).

The processes are started before anything in this process touches the GPU (spawn)."""
import os
import sys
import uuid

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _recipe(I, case):
    if case == "closed":
        return I.recipe_headline(150, 131, 3)
    if case == "sill_nudged":          # N/S sponges (nudging on both end bands), outcropping layers, Leith viscosity
        return I.recipe_sill_exchange3d(lm=60, mm=203, nlay=4, dt_s=30.0, npts=15, sill_halfwidth=20.0)
    if case == "jet_ring":             # periodic in x and y: a ring of bands + the companion frame on rank 0
        return I.recipe_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5)
    if case in ("closed_obc", "jet_ring_obc"):
        # nudged OPEN boundaries with mcbc = 0 (no_gradient_obc, private_mod.f95:2613-2679): every rank finds the segments of its
        # own rows (western and eastern sponges cross every band; the closed frame's northern one belongs to the last band;
        # the jet stays periodic in y: a ring of bands, the companion frame's segments put together from rank 0's)
        import numpy as np
        yper = case == "jet_ring_obc"
        base = I.recipe_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5) if yper else I.recipe_headline(150, 131, 3)
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
    raise ValueError(case)


def _worker(rank, world, case, shm_name, overlap, calls):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("BEOM_SHM_TIMEOUT_S", "90")
    from beom_amd import capi, inputs as I, slab
    from beom_amd.grid import read_input_data
    from helpers import same, same_bits
    recipe = _recipe(I, case)
    p = recipe.p
    yper = float(p.yper) > 0.5
    f, g, orphan = slab.build_band(recipe, world, rank)            # this rank's rows only
    band = capi.BandEngine(f, p, world, rank, device=0, shm_name=shm_name, orphan=orphan)
    d = band.describe()
    assert d["bands_total"] == world and d["bands_local"] == 1 and d["ring"] == int(yper), d
    assert "shared memory" in d["transport"], d
    band.set_option("overlap", int(overlap))
    t = 1
    for n in calls:
        band.step(t, n)
        t += n
    nsteps = t - 1
    st, so = band.download(orphan=True) if orphan is not None else (band.download(), None)
    stats = band.stats()
    # every exchange is an appointment with the neighbours: all ranks finish stepping before anyone compares
    whole = capi.Engine(read_input_data(p, files=recipe.rows(0, p.mm + 1)))
    whole.step(1, nsteps)
    ref = whole.download()
    L = p.lm + 1
    a, b = 1 + (g.own0 - 1) * L, 1 + g.own1 * L
    la, lb = g.local_rows(g.own0, g.own1)
    for k in ("hlay", "u", "v", "h_u", "h_v"):
        assert same_bits(st[k][:, la:lb], ref[k][:, a:b]), (case, rank, k)
        if so is not None:
            assert same_bits(so[k][:, 1:], ref[k][:, -L:]), (case, rank, "orphan row", k)
    for k in ("rs_h", "dmdx", "dmdy"):
        assert same_bits(st[k][:, la:lb, :], ref[k][:, a:b, :]), (case, rank, k)
        if so is not None:
            assert same_bits(so[k][:, 1:, :], ref[k][:, -L:, :]), (case, rank, "orphan row", k)
    # the ghost rows hold the neighbours' owned rows after the last exchange
    rows = g.global_rows()
    for jl, jg in enumerate(rows):
        if g.own0 <= g.win0 + jl <= g.own1:
            continue
        assert same_bits(st["u"][:, 1 + jl * L: 1 + (jl + 1) * L], ref["u"][:, 1 + (jg - 1) * L: 1 + jg * L]), (case, rank, "ghost row", jg)
    if case.endswith("_obc"):
        assert stats["split"] == 0 and stats["plain"] == nsteps, stats      # a step with an open-boundary pass runs whole
    elif overlap:
        assert stats["split"] >= nsteps - 5, stats           # steps 1-3 and the first step of a call after an upload are whole
    else:
        assert stats["split"] == 0, stats
    band.close(); whole.close()


def _run(world, case, overlap, calls=(7, 6)):
    import torch.multiprocessing as mp
    name = "/beom_test_%d_%s" % (os.getpid(), uuid.uuid4().hex[:12])
    try:
        mp.spawn(_worker, args=(world, case, name, overlap, calls), nprocs=world, join=True)
    finally:
        try:
            os.unlink("/dev/shm" + name)          # (band 0 removes it once all have attached; only a failed start leaves it)
        except OSError:
            pass


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["closed", "sill_nudged", "jet_ring"])
def test_bands_in_separate_processes_match_single_handle(world, case):
    _run(world, case, overlap=True)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["closed_obc", "jet_ring_obc"])
def test_open_boundaries_on_rank_local_windows(world, case):
    """beom_multi_set_open_boundaries_local: the segments of no_gradient_obc found by every rank from its own rows."""
    _run(world, case, overlap=True)


def test_two_processes_in_the_64x8_tile_geometry():
    """Bands of the headline frame cut 2 or 4 ways are tall enough for the 64 x 8 tiles: the cut steps in that geometry."""
    old = os.environ.get("BEOM_TILE4")
    os.environ["BEOM_TILE4"] = "0"                 # (inherited by the ranks; read when a handle is created)
    try:
        _run(2, "closed", overlap=True)
        _run(2, "jet_ring", overlap=True)
    finally:
        if old is None:
            os.environ.pop("BEOM_TILE4")
        else:
            os.environ["BEOM_TILE4"] = old


def test_two_processes_plain_exchange_matches_too():
    _run(2, "closed", overlap=False, calls=(5, 1, 7))


def test_missing_neighbour_is_an_error_not_a_hang():
    """A rank whose neighbour never shows up gets an error from beom_multi_create_local_ex after twice the timeout of the
    shared-memory transport (BEOM_SHM_TIMEOUT_S) — every wait of that transport is bounded."""
    import time
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from beom_amd import capi, inputs as I, slab
    recipe = _recipe(I, "closed")
    f, g, orphan = slab.build_band(recipe, 2, 0)
    name = "/beom_test_alone_%d_%s" % (os.getpid(), uuid.uuid4().hex[:12])
    old = os.environ.get("BEOM_SHM_TIMEOUT_S")
    os.environ["BEOM_SHM_TIMEOUT_S"] = "1.5"
    t0 = time.time()
    try:
        with pytest.raises(capi.BeomError, match="waited"):
            capi.BandEngine(f, recipe.p, 2, 0, device=0, shm_name=name, orphan=orphan)
    finally:
        if old is None:
            os.environ.pop("BEOM_SHM_TIMEOUT_S")
        else:
            os.environ["BEOM_SHM_TIMEOUT_S"] = old
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
    assert time.time() - t0 < 30.0
