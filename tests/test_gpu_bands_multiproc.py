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
    if overlap:
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


def test_two_processes_plain_exchange_matches_too():
    _run(2, "closed", overlap=False, calls=(5, 1, 7))
