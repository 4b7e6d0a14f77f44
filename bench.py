#!/usr/bin/env python3
"""bench.py — BEOM time-step throughput on MI355X (contract: see README / DESIGN.md §6).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one model time step (update_h → update_mont_rvor_pvor_dive_kine →
update_viscosity → update_u/update_v, private_mod.f95:2259-2290) of the headline
workload of SURVEY.md §8(d): closed flat basin, 4096 x 4096 cells x 4 layers, FP64,
g_fb=1, uadv=1, Leith viscosity every step.  State is resident in HBM when the timed
region starts.  For N > 1 the SAME global grid is cut into N j-slabs (strong scaling)
with a ghost-row exchange per step over RCCL (beom_amd/slab.py).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_STEP = 416.0            # algorithmic bytes per cell-layer update (52 FP64 words)
B_ALG_KERNEL = {              # per sweep, bytes per cell-layer (SURVEY §8d word counts x 8)
    "update_h": 7 * 8, "update_mont": 9 * 8, "update_viscosity": 4 * 8, "update_u": 16 * 8, "update_v": 16 * 8,
    # fused launches do the work of two reference sweeps: their algorithmic bytes are the sum
    "update_mont+update_viscosity": (9 + 4) * 8, "update_u+update_v": (16 + 16) * 8,
}
KERNEL_ORDER = ("update_h", "update_mont", "update_viscosity", "update_u", "update_v",
                "update_mont+update_viscosity", "update_u+update_v")
NCLS = len(KERNEL_ORDER)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (≈6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--lm", type=int, default=4096)
    ap.add_argument("--mm", type=int, default=4096)
    ap.add_argument("--nlay", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1024x1024x4")
    return ap.parse_args()


def cpu_baseline(sample: str):
    """Times the reference Fortran (oracle/_ref/baseline, built by __graft_entry__.build()
    where /root/reference exists) on this host's cores over a bounded sample of the
    same workload; falls back to the C restatement (kind 'port') if the binary is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import cpu_baseline as cb
        return cb.run(sample)
    except Exception as exc:  # the baseline is a reported extra, never fatal
        return {"value": None, "unit": "cell-layer updates/s", "cores": None, "kind": "unavailable",
                "sample": "%s (%s)" % (sample, str(exc)[:200])}


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("BEOM_DIST_BACKEND", "nccl")     # "gloo": rehearsal of N ranks on one GPU
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local_rank, ndev))
    local_rank = local_rank % ndev
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = world
    if a.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)

    from beom_amd import capi, inputs
    from beom_amd.grid import read_input_data

    t0 = time.time()
    p, files = inputs.case_headline(a.lm, a.mm, a.nlay)
    if world == 1:
        f = read_input_data(p, files=files)
        del files
        eng = capi.Engine(f, device=local_rank)
        runner = eng
        dense = eng.is_dense
        exchange_desc = None
    else:
        from beom_amd import slab
        runner = slab.SlabRunner.from_global_case(p, files, rank, world, device=local_rank,
                                                  overlap=os.environ.get("BEOM_NO_OVERLAP") is None)
        del files
        dense = runner.engine.is_dense
        # self-check on THIS machine: the overlapped exchange must give the owned rows of the plain
        # (step, exchange, step, ...) form bit for bit; if not, time the plain form
        verified = None
        if runner.overlap:
            def owned_copy():
                a, b = runner.g.local_rows(runner.g.own0, runner.g.own1)
                return [t[:, a:b].clone() for t in runner.engine.field_tensors().values()]
            runner.overlap = False
            runner.step(1, 9); runner.sync()
            ref_rows = owned_copy()
            runner.reset_state()
            runner.overlap = True
            runner.step(1, 9); runner.sync()
            same = all(torch.equal(x, y) for x, y in zip(ref_rows, owned_copy()))
            flag = torch.tensor([1 if same else 0], device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified = bool(flag.item())
            runner.overlap = verified
            runner.reset_state()
            del ref_rows
        exchange_desc = runner.describe()
        exchange_desc["overlap_verified_bitwise_vs_plain_exchange"] = verified
    t_setup = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: steps 1..W (the first three are plain forward-backward, private_mod.f95:1859-1877)
    W = max(a.warmup, 3)
    runner.step(1, W)
    barrier()
    K = a.steps
    t1 = time.perf_counter()
    ms, nl = runner.profile_steps(W + 1, K)       # launches K steps with HIP events around each kernel, then syncs
    barrier()
    t2 = time.perf_counter()
    elapsed = t2 - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    units_per_step = float(p.ndeg) * p.nlay              # cell-layer updates in one step, whole job
    value = units_per_step * K / elapsed
    # dominant kernel = the longest-running sweep of this run
    # a split step launches each sweep twice (interior + edge rows): account per STEP
    per_launch_ms = [ms[i] / K if nl[i] else 0.0 for i in range(NCLS)]
    dom = max(range(NCLS), key=lambda i: ms[i])
    units_per_launch = units_per_step / world            # one launch covers this rank's slab, all layers
    ach = B_ALG_KERNEL[KERNEL_ORDER[dom]] * units_per_launch / (per_launch_ms[dom] * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": KERNEL_ORDER[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "avg_launch_ms": per_launch_ms[dom],
            "alg_bytes_per_launch": B_ALG_KERNEL[KERNEL_ORDER[dom]] * units_per_launch,
            "per_kernel": {KERNEL_ORDER[i]: {"avg_ms": per_launch_ms[i], "launches": nl[i],
                                             "alg_GBs": (B_ALG_KERNEL[KERNEL_ORDER[i]] * units_per_launch
                                                         / (per_launch_ms[i] * 1e-3) / 1e9) if nl[i] else None}
                           for i in range(NCLS) if nl[i]},
            "step_alg_GBs": B_ALG_STEP * value / 1e9, "step_frac": B_ALG_STEP * value / 1e9 / HBM_PEAK_GBS}
    traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(traffic_file) and world == 1 and (a.lm, a.mm, a.nlay) == (4096, 4096, 4):   # measured for that launch only
        try:
            roof["traffic"] = json.load(open(traffic_file)).get(KERNEL_ORDER[dom])
        except Exception:
            pass

    out = {
        "metric": "cell-layer updates/s", "value": value, "unit": "cell-layer updates/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "closed flat basin %dx%d cells x %d layers (SURVEY §8d headline; dense frame ndeg=%d), "
                               "g_fb=1 uadv=1 dvis=0.2, FP64" % (a.lm, a.mm, a.nlay, p.ndeg),
                   "global_cells": p.ndeg, "layers": p.nlay, "parallelism": "j-slab x%d" % world,
                   "dense_fast_path": bool(dense), "setup_s": round(t_setup, 1)},
        "roofline": roof,
    }
    if exchange_desc:
        out["config"]["halo"] = exchange_desc
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
