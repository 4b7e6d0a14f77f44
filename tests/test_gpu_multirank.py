"""Two ranks, one GPU each in production — here both on the one GPU of the test box, with the
gloo backend moving the (CUDA) ghost buffers: the whole overlapped multi-rank path of
beom_amd.slab.SlabRunner (second stream, events, beom_step_phase 1/2, one-launch packing) with a
genuinely asynchronous exchange.  Owned rows must equal the single-domain GPU run bit for bit."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, overlap, nsteps):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import torch.distributed as dist
    from beom_amd import capi, inputs as I, slab
    from beom_amd.grid import read_input_data
    from helpers import same
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        p, files = I.case_headline(150, 131, 3)
        f = read_input_data(p, files=files)
        run = slab.SlabRunner.from_global_case(p, files, rank, world, device=0, overlap=overlap)
        assert run.overlap == overlap and run.engine.is_dense
        run.step(1, nsteps)
        run.sync()
        whole = capi.Engine(f)
        whole.step(1, nsteps)
        ref, st, g = whole.download(), run.engine.download(), run.g
        a, b = 1 + (g.own0 - 1) * g.L, 1 + g.own1 * g.L
        la, lb = g.local_rows(g.own0, g.own1)
        for k in ("hlay", "u", "v", "h_u", "h_v"):
            assert same(st[k][:, la:lb], ref[k][:, a:b]), (rank, k)
        for k in ("rs_h", "dmdx", "dmdy"):
            assert same(st[k][:, la:lb, :], ref[k][:, a:b, :]), (rank, k)
        # ghost rows hold the neighbour's values after the final exchange
        if g.ghost_n:
            ga, gb = g.local_rows(g.own1 + 1, g.win1)
            assert same(st["u"][:, ga:gb], ref["u"][:, 1 + g.own1 * g.L: 1 + g.win1 * g.L]), (rank, "ghost")
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_on_one_gpu_match_single_domain(overlap):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), overlap, 14), nprocs=2, join=True)


@pytest.mark.parametrize("case", ["jet_ring_of_one", "closed_single_band"])
def test_band_from_recipe_rows_matches_single_handle(case):
    """beom_multi_create_local — what every rank of bench.py's N-GPU run does: the band's rows come from the
    recipe alone, the library generates connectivity and masks, the exchange runs over RCCL.  A one-GPU box
    can hold one rank: for the y-periodic jet its ring closes on itself (self send/recv over a one-rank RCCL
    communicator, companion frame for the orphan row); results must equal the single handle's bit for bit."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    from beom_amd import capi, inputs as I, slab
    from beom_amd.grid import read_input_data
    from helpers import same, same_bits
    recipe = I.recipe_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5) if case == "jet_ring_of_one" else I.recipe_headline(150, 131, 3)
    p = recipe.p
    yper = float(p.yper) > 0.5
    f, g, orphan = slab.build_band(recipe, 1, 0)
    assert (orphan is not None) == yper and g.rows == (p.mm + 2 * slab.GHOST if yper else p.mm + 1)
    band = capi.BandEngine(f, p, 1, 0, device=0, rccl_id=capi.rccl_unique_id() if yper else None, orphan=orphan)
    assert band.describe()["ring"] == int(yper)
    whole = capi.Engine(read_input_data(p, files=recipe.rows(0, p.mm + 1)))
    band.step(1, 7); band.step(8, 6)
    whole.step(1, 13)
    ref = whole.download()
    st, so = band.download(orphan=True)
    L = p.lm + 1
    a, b = 1 + (g.own0 - 1) * L, 1 + g.own1 * L
    la, lb = g.local_rows(g.own0, g.own1)
    for k in ("hlay", "u", "v", "h_u", "h_v"):
        assert same(st[k][:, la:lb], ref[k][:, a:b]), k
        if yper:
            assert same_bits(so[k][:, 1:], ref[k][:, -L:]), ("orphan row", k)
    for k in ("rs_h", "dmdx", "dmdy"):
        assert same(st[k][:, la:lb, :], ref[k][:, a:b, :]), k
        if yper:
            assert same_bits(so[k][:, 1:, :], ref[k][:, -L:, :]), ("orphan row", k)
    if yper:
        assert band.stats()["split"] >= 9
    band.close(); whole.close()
