"""Multi-rank path on CPU: world_size 2 and 3 over gloo.  Each rank runs beom_amd.slab's
SlabRunner (the same decomposition / pack / exchange / unpack code the GPU path uses)
around a CPU adapter of the oracle, and checks that its OWNED rows equal the
single-domain oracle run bit for bit after 12 steps (steps 1-3 plain FB, both U/V
orders, G = 4 ghost rows, one exchange per step)."""
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
    from beom_amd import inputs as I
    if name == "closed_3l":
        return I.case_headline(30, 44, 3)
    if name == "sill_ocrp_nudg":
        return I.case_sill_exchange3d(lm=15, mm=47, nlay=2, dt_s=0.01, npts=5, sill_halfwidth=6.0)
    if name == "soliton_xper":
        return I.case_soliton(lm=31, mm=39, dt_s=5.0)
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
    from helpers import same
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, files = _case(case)
        f = read_input_data(p, files=files)
        geom = slab.decompose(p.mm, p.lm, world)[rank]
        lf = slab.slice_fields(f, geom)
        run = slab.SlabRunner(OracleEngine(lf), geom, p.nlay, dist=dist)
        run.step(1, nsteps)
        ref = oracle_lib.Oracle(f)
        ref.step(1, nsteps)
        a, b = 1 + (geom.own0 - 1) * geom.L, 1 + geom.own1 * geom.L
        for k in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "v_cc", "v_ll"):
            loc = run.engine.o.a[k]
            if loc.ndim == 3:
                la, lb = geom.local_rows(geom.own0, geom.own1)
                ok = same(loc[:, la:lb, :], ref.a[k][:, a:b, :])
            else:
                ok = same(run.owned(loc), ref.a[k][..., a:b])
            assert ok, (case, rank, k)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,case", [(2, "closed_3l"), (3, "closed_3l"), (2, "sill_ocrp_nudg"),
                                        (2, "soliton_xper")])
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
        assert rows == list(range(1, mm + 2))


def test_y_periodic_multi_rank_is_refused():
    from beom_amd import inputs as I, slab
    from beom_amd.grid import read_input_data
    p, files = I.case_unstable_jet(lm=21, mm=27, nlay=1, dt_s=1.0)
    f = read_input_data(p, files=files)
    with pytest.raises(NotImplementedError):
        slab.slice_fields(f, slab.decompose(p.mm, p.lm, 2)[0])
