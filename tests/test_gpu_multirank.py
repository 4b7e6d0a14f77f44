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
