#!/usr/bin/env python3
"""bench.py — BEOM time-step throughput on MI355X (contract: see README / DESIGN.md §6).

  python bench.py --gpus N --steps K --warmup W [--case headline|sill4|beach8|jet]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one model time step (update_h → update_mont_rvor_pvor_dive_kine →
update_viscosity → update_u/update_v, private_mod.f95:2259-2290) of the headline
workload of SURVEY.md §8(d): closed flat basin, 4096 x 4096 cells x 4 layers, FP64,
g_fb=1, uadv=1, Leith viscosity every step.  State is resident in HBM when the timed
region starts.

N > 1: the SAME global grid is cut into N bands of rows (strong scaling), one process per GPU,
every rank builds only its own rows from the recipe, and the K steps of the timed region run
inside the library (beom_multi_step: a step is cut boundary first — the edge strips of the momentum
sweep, the packing and the ghost-row exchange over RCCL run inside the sweep's interior rows).  Without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the
environment) this script starts its own N ranks before anything touches a GPU; a WORLD_SIZE that
disagrees with --gpus is an error.  torch.distributed (gloo) carries only the control plane:
the RCCL unique id, barriers, the max over ranks.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import datetime
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_STEP = 416.0            # algorithmic bytes per cell-layer update (52 FP64 words)
B_ALG_KERNEL = {              # per sweep, bytes per cell-layer (SURVEY §8d word counts x 8)
    "update_h": 7 * 8, "update_mont": 9 * 8, "update_viscosity": 4 * 8, "update_u": 16 * 8, "update_v": 16 * 8,
    # fused launches do the work of two reference sweeps: their algorithmic bytes are the sum
    "update_mont+update_viscosity": (9 + 4) * 8, "update_u+update_v": (16 + 16) * 8,
    "(unused)": 0,
}
KERNEL_ORDER = ("update_h", "update_mont", "update_viscosity", "update_u", "update_v",
                "update_mont+update_viscosity", "update_u+update_v", "(unused)")
NCLS = len(KERNEL_ORDER)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (≈6.3 TB/s achievable)

CASES = {
    # name: (recipe factory, description) — BASELINE.json configs; sizes overridable with --lm/--mm/--nlay
    "headline": (lambda I, a: I.recipe_headline(a.lm or 4096, a.mm or 4096, a.nlay or 4),
                 "closed flat basin (SURVEY §8d headline), g_fb=1 uadv=1 dvis=0.2"),
    "sill4": (lambda I, a: I.recipe_sill_exchange3d(lm=a.lm or 4096, mm=a.mm or 512, nlay=a.nlay or 4, dt_s=30.0, npts=15,
                                                     sill_halfwidth=50.0),
              "sill_exchange3D recipe (BASELINE config 4): Gaussian sill, ocrp=1, N/S sponge, dvis=0.9"),
    "beach8": (lambda I, a: I.recipe_carrier_beach(lm=a.lm or 8192, mm=a.mm or 8192, nlay=a.nlay or 8, dt_s=0.08),
               "carrier_beach recipe replicated in y (BASELINE config 5): wetting/drying ocrp=1, W sponge, no Leith refresh"),
    "jet": (lambda I, a: I.recipe_unstable_jet(lm=a.lm or 2048, mm=a.mm or 2048, nlay=a.nlay or 2, dt_s=50.0),
            "unstable_jet recipe (BASELINE config 3): doubly periodic, dvis=0.2"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--case", default="headline", choices=sorted(CASES))
    ap.add_argument("--lm", type=int, default=0)
    ap.add_argument("--mm", type=int, default=0)
    ap.add_argument("--nlay", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1024x1024x4")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="seconds of untimed stepping before the W warm-up steps (clocks settle); the initial state is "
                         "uploaded again afterwards")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: ONE process drives all N devices (beom_multi_create from global arrays)")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "peer", "shm"),
                    help="rccl (default) | peer: --single-process only | shm: one rank per band as usual, the ghost rows staged "
                         "through shared memory — a rehearsal of the N-rank run on a box with fewer GPUs than ranks (the ranks "
                         "then share devices: LOCAL_RANK modulo the visible count); a result measured that way says so")
    ap.add_argument("--force-bands", action="store_true",
                    help="rehearsal on one GPU: take the N > 1 path (rows from the recipe, beom_multi_create_local, RCCL) "
                         "with a single band; with --case jet the ring then closes on itself over RCCL")
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


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) BEFORE
    anything in this process touches a GPU, give them the torchrun environment, pass rank 0's line on."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BEOM_BENCH_SPAWNED="1",
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies early (fewer GPUs than ranks, a failed build) must not leave the others waiting in a rendezvous:
    # watch them all, and stop exactly the processes started here as soon as one has failed
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad:
            failed = bad[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return abs(failed) if failed else max(abs(rc) for rc in rcs)


def gpu_sensors(pci: str | None) -> dict:
    """Shader clock (MHz) and power (W) of the card with this PCI address, from sysfs — cheap enough to read
    right at the edges of the timed region; None where the box does not expose them."""
    out = {"sclk_mhz": None, "power_w": None}
    try:
        cards = [c for c in glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk")
                 if pci and os.path.realpath(os.path.dirname(c)).lower().endswith(pci)]
        if cards:
            path = cards[0]
            for line in open(path).read().splitlines():
                if line.strip().endswith("*"):
                    out["sclk_mhz"] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            hw = glob.glob(os.path.join(os.path.dirname(path), "hwmon", "hwmon*", "power1_average")) + \
                glob.glob(os.path.join(os.path.dirname(path), "hwmon", "hwmon*", "power1_input"))
            if hw:
                out["power_w"] = round(int(open(hw[0]).read().strip()) / 1e6, 1)
    except Exception:
        pass
    return out


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1 and not a.single_process:
        sys.exit(spawn_ranks(a))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.single_process:
        if world != 1:
            raise SystemExit("bench.py: --single-process drives all devices from one process; do not use a launcher")
    elif a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world))

    # the CPU leg first: no process of this job has touched a GPU yet when the reference binary is started
    # (N > 1: the other ranks wait in the rendezvous meanwhile — before anything is timed)
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_sample)

    # Libraries underneath (RCCL prints a version banner at communicator creation) write to fd 1: the ONE line this
    # script owes its caller goes to the real stdout, everything else to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    ndev = max(torch.cuda.device_count(), 1)
    shm = a.transport == "shm" and not a.single_process
    if shm:
        local_rank = local_rank % ndev
    if local_rank >= ndev:
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local_rank, ndev))
    if a.single_process and a.gpus > ndev and a.transport == "rccl":
        raise SystemExit("bench.py: --single-process --gpus %d but only %d GPUs visible" % (a.gpus, ndev))
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane only; the ghost rows travel over RCCL inside the library
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))

    from beom_amd import capi, inputs, slab
    from beom_amd.grid import read_input_data

    t0 = time.time()
    recipe = CASES[a.case][0](inputs, a)
    p = recipe.p
    n_gpus = a.gpus
    halo = None
    rccl_failed = None
    banded = n_gpus > 1 or a.force_bands
    if not banded:
        f = read_input_data(p, files=recipe.rows(0, p.mm + 1))
        eng = capi.Engine(f, device=local_rank)
        dense = eng.is_dense
        reset = lambda: eng.upload(**{k: getattr(f, k) for k in capi.STATE_NAMES})
    elif a.single_process:
        f = read_input_data(p, files=recipe.rows(0, p.mm + 1))
        eng = capi.MultiEngine(f, devices=[d % ndev for d in range(n_gpus)],
                               transport=capi.XCHG_RCCL if a.transport == "rccl" else capi.XCHG_PEER)
        dense = True
        reset = lambda: eng.upload(**{k: getattr(f, k) for k in capi.STATE_NAMES})
    else:
        why = None
        eng = None
        try:
            if shm:
                uid = ["/beom_bench_%d_%s" % (os.getpid(), os.urandom(6).hex()) if rank == 0 else None]
            else:
                uid = [capi.rccl_unique_id() if rank == 0 else None]
        except capi.BeomError as exc:                                    # librccl could not be bound on this machine
            uid, why = [b""], str(exc)
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        if uid[0]:
            try:
                f, geom, orphan = slab.build_band(recipe, world, rank)   # this rank's rows only
                eng = capi.BandEngine(f, p, world, rank, device=local_rank, orphan=orphan,
                                      **(dict(shm_name=uid[0]) if shm else dict(rccl_id=uid[0])))
            except capi.BeomError as exc:
                why = str(exc)
        ok = eng is not None
        if world > 1:
            t = torch.tensor([1 if ok else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item())
        if not ok:
            # No RCCL communicator on this machine.  Rather than no measurement at all: ONE process (rank 0) drives the N
            # devices with peer copies (the other transport of the same library loop); config.halo says so, with the reason.
            reasons = [why]
            if world > 1:
                reasons = [None] * world
                dist.all_gather_object(reasons, why)
            if eng is not None:
                eng.close()
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            if rank != 0:
                os.dup2(real_stdout, 1)
                sys.exit(0)
            world = 1
            rccl_failed = "; ".join(sorted({str(r)[:300] for r in reasons if r})) or "unknown"
            print("bench.py: RCCL transport unavailable (%s): one process, peer copies" % rccl_failed, file=sys.stderr)
            a.single_process, a.transport = True, "peer"
            f = read_input_data(p, files=recipe.rows(0, p.mm + 1))
            eng = capi.MultiEngine(f, devices=[d % ndev for d in range(n_gpus)], transport=capi.XCHG_PEER)
            reset = lambda: eng.upload(**{k: getattr(f, k) for k in capi.STATE_NAMES})
        else:
            reset = lambda: eng.upload()
        dense = True
    if banded:
        d = eng.describe()
        L = p.lm + 1
        seen = [{"rank": rank, "device": local_rank, "host": socket.gethostname(), "pid": os.getpid()}]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, seen[0])
            seen = gathered
        halo = {"backend": d["transport"] + (" (librccl %d)" % d["rccl_version"] if d["rccl_version"] else ""),
                "control_plane": "torch.distributed gloo" if world > 1 else "none (one process)",
                "ranks_seen": len(seen) if world > 1 else 1, "devices": [s["device"] for s in seen] if world > 1 else
                [x % ndev for x in range(n_gpus)], "ranks": seen,
                "bands": d["bands_total"], "ring_in_y": bool(d["ring"]), "ghost_rows": slab.GHOST, "exchanges_per_step": 1,
                "fields": list(slab.EXCHANGED), "bytes_per_direction_per_step": len(slab.EXCHANGED) * p.nlay * slab.GHOST * L * 8,
                "step_loop": "inside the library (beom_multi_step)", "state_build": "global arrays, cut by the library"
                if a.single_process else "each rank builds its own rows from the recipe"}
        if shm:
            halo["rehearsal"] = ("ghost rows staged through shared memory, %d ranks on %d device(s): the N-rank code path, NOT an "
                                 "N-GPU measurement" % (world, len({x["device"] for x in seen})))
        halo["per_kernel_note"] = ("a cut step launches the momentum sweep twice (the strips next to the ghost zones on the second stream, "
                                   "the rows in between on the main one, side by side): roofline.per_kernel sums both, so concurrent time "
                                   "is counted twice there; `value` is wall time")
        if rccl_failed:
            halo["fallback"] = "RCCL transport unavailable (%s): ONE process drives the %d devices over peer copies" % (rccl_failed, n_gpus)
    t_setup = time.time() - t0

    def all_min(flag: bool) -> bool:
        if world == 1:
            return flag
        t = torch.tensor([1 if flag else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: on THIS machine the overlapped form (split steps around the exchange in flight) must reproduce the
    # plain form (every step waits for its ghosts) bit for bit on every band
    if banded:
        names = ("hlay", "u", "v", "h_u", "h_v")
        eng.set_option("overlap", 0)
        eng.step(1, 9)
        plain = eng.download(names)
        reset()
        eng.set_option("overlap", 1)
        eng.step(1, 9)
        over = eng.download(names)
        same = all(np.array_equal(plain[k], over[k]) for k in names)
        halo["overlap_verified_bitwise_vs_plain_exchange"] = all_min(same)
        halo["band_steps_split_vs_plain"] = eng.stats()
        del plain, over
        reset()
        # which form is faster on THIS machine: the step cut boundary first (edge strips, packing and the exchange inside the
        # interior rows of the momentum sweep) or the step in one piece with the exchange behind it?  Over a wire the cut form
        # hides the transfer; over a loop-back (one rank) there is nothing to hide.  20 + 60 steps each, untimed region.
        form_ms = {}
        for name, flag in (("cut", 1), ("whole", 0)):
            eng.set_option("overlap", flag)
            eng.step(1, 20)
            barrier()
            tq = time.perf_counter()
            eng.step(21, 60)
            barrier()
            dtq = time.perf_counter() - tq
            if world > 1:
                tt = torch.tensor([dtq], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtq = float(tt.item())
            form_ms[name] = dtq / 60 * 1e3
            reset()
        use_cut = halo["overlap_verified_bitwise_vs_plain_exchange"] and form_ms["cut"] <= form_ms["whole"]
        if os.environ.get("BEOM_BENCH_FORM") in ("cut", "whole"):
            use_cut = os.environ["BEOM_BENCH_FORM"] == "cut" and halo["overlap_verified_bitwise_vs_plain_exchange"]
        eng.set_option("overlap", 1 if use_cut else 0)
        halo["step_form"] = {"chosen": "cut boundary first (exchange inside the interior rows of the momentum sweep)" if use_cut
                             else "whole step, exchange behind it", "ms_per_step_probe": form_ms}

    # pre-warm: untimed stepping so that the timed region does not start on idle clocks; then the initial state again
    prewarm_steps = 0
    if a.prewarm_s > 0:
        tp = time.perf_counter()
        eng.step(1, 10)
        per = max((time.perf_counter() - tp) / 10, 1e-5)
        if world > 1:                 # every rank must take the SAME number of steps (each step is an exchange)
            tt = torch.tensor([per], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            per = float(tt.item())
        more = max(0, min(int(a.prewarm_s / per) - 10, 20000))
        if more:
            eng.step(11, more)
        prewarm_steps = 10 + more
        reset()

    # warm-up: steps 1..W (the first three are plain forward-backward, private_mod.f95:1859-1877)
    W = max(a.warmup, 3)
    eng.step(1, W)
    barrier()
    K = a.steps
    pci = capi.device_pci_bus_id(local_rank)
    s0 = gpu_sensors(pci)
    # HIP events around every launch cost a few microseconds of pipeline bubble each — 1-2 % of a 3.8 ms step, ~10 % of a
    # 0.5 ms band-step at N = 8: every 4th step of the timed region is bracketed (at least five sampled steps)
    # ... and a sampled step brackets only ONE kind of sweep, by turns, so that no bracketed launch starts in the bubble of the
    # bracket before it (round 2: the per-kernel times summed to more than the wall time per step)
    stride = 2 if K >= 12 else 1
    rotate = K >= 12
    if os.environ.get("BEOM_BENCH_STRIDE") in ("1", "2", "4"):       # (A/B of the sampling itself)
        stride = int(os.environ["BEOM_BENCH_STRIDE"])
    if os.environ.get("BEOM_BENCH_ROTATE") in ("0", "1"):
        rotate = os.environ["BEOM_BENCH_ROTATE"] == "1"
    eng.set_option("profile_stride", stride)
    eng.set_option("profile_rotate", int(rotate))
    KIND = {0: 0, 1: 1, 2: 1, 5: 1, 3: 2, 4: 2, 6: 2, 7: 2}          # sweep class -> what a rotating step brackets (StepTimer::kind)
    sampled_steps = [t for t in range(W + 1, W + K + 1) if t % stride == 0]
    sampled = len(sampled_steps)
    steps_of_kind = [sum(1 for t in sampled_steps if (not rotate) or (t // stride) % 3 == k) for k in range(3)]
    t1 = time.perf_counter()
    ms, nl = eng.profile_steps(W + 1, K)       # K steps in ONE library call, HIP events around the kernels of the sampled steps, then a sync
    barrier()
    t2 = time.perf_counter()
    s1 = gpu_sensors(pci)
    elapsed = t2 - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    units_per_step = float(p.ndeg) * p.nlay              # cell-layer updates in one step, whole job
    value = units_per_step * K / elapsed
    # dominant kernel = the longest-running sweep of this run
    # a split step launches each sweep twice (interior + edge rows): account per STEP
    per_launch_ms = [ms[i] / max(steps_of_kind[KIND[i]], 1) if nl[i] else 0.0 for i in range(NCLS)]
    dom = max(range(NCLS), key=lambda i: ms[i])
    units_per_launch = units_per_step / n_gpus           # one launch covers one band, all layers
    ach = B_ALG_KERNEL[KERNEL_ORDER[dom]] * units_per_launch / (per_launch_ms[dom] * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": KERNEL_ORDER[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": None, "measured_GBs": None, "measured_frac": None,
            "avg_launch_ms": per_launch_ms[dom], "events": "HIP events inside the timed region: every %s step brackets %s "
            "(%d of %d steps sampled; %s)" % ({1: "", 2: "2nd", 4: "4th"}[stride], "ONE kind of sweep, by turns" if rotate else "its sweeps",
                                              sampled, K, "steps per kind h / mont+visc / u+v: %s" % steps_of_kind),
            "sum_of_kernel_ms_per_step": sum(per_launch_ms),
            "alg_bytes_per_launch": B_ALG_KERNEL[KERNEL_ORDER[dom]] * units_per_launch,
            "per_kernel": {KERNEL_ORDER[i]: {"avg_ms": per_launch_ms[i], "launches": nl[i],
                                             "alg_GBs": (B_ALG_KERNEL[KERNEL_ORDER[i]] * units_per_launch
                                                         / (per_launch_ms[i] * 1e-3) / 1e9) if nl[i] else None}
                           for i in range(NCLS) if nl[i]},
            "step_alg_GBs": B_ALG_STEP * value / 1e9, "step_frac": B_ALG_STEP * value / 1e9 / HBM_PEAK_GBS}
    # HBM bytes per launch come from rocprofv3 PMC passes of this same command (tools/gpu_profile.sh): PMC counters
    # cannot be read from inside the run, so the stored figure of the latest profiled build is replayed and labelled
    traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(traffic_file) and n_gpus == 1 and a.case == "headline" and (p.lm, p.mm, p.nlay) == (4096, 4096, 4):
        try:
            tj = json.load(open(traffic_file))
            roof["traffic"] = tj.get(KERNEL_ORDER[dom])
            roof["traffic_source"] = ("profiles/traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                                      "`bench.py --steps 20 --warmup 5`, 2xFETCH_SIZE + WRITE_SIZE; replayed, not measured in this run")
            if roof["traffic"]:
                roof["measured_GBs"] = roof["traffic"] / (per_launch_ms[dom] * 1e-3) / 1e9
                roof["measured_frac"] = roof["measured_GBs"] / HBM_PEAK_GBS
            step_bytes = sum(tj.get(KERNEL_ORDER[i], 0.0) for i in range(NCLS) if nl[i])
            if step_bytes:
                roof["step_measured_bytes_per_update"] = step_bytes / units_per_step
                roof["step_measured_frac"] = step_bytes / (elapsed / K) / 1e9 / HBM_PEAK_GBS
        except Exception:
            pass

    # the state after the timed region: finite, and a checksum of hlay, u, v (sum of |x| per field, over the owned rows of all ranks)
    after = eng.download(("hlay", "u", "v"))
    if banded and not a.single_process:
        la, lb = geom.local_rows(geom.own0, geom.own1)
        after = {k: v[:, la:lb] for k, v in after.items()}
    finite = all(bool(np.isfinite(v).all()) for v in after.values())
    sums = [float(np.abs(after[k]).sum()) for k in ("hlay", "u", "v")]
    if world > 1:
        tt = torch.tensor(sums + [1.0 if finite else 0.0], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        sums, finite = [float(x) for x in tt[:3]], float(tt[3]) == float(world)
    state_check = {"finite": finite, "sum_abs": dict(zip(("hlay", "u", "v"), sums)), "after_steps": W + K}
    del after

    out = {
        "metric": "cell-layer updates/s", "value": value, "unit": "cell-layer updates/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %dx%d cells x %d layers (dense frame ndeg=%d), FP64; %s"
                               % (a.case, p.lm, p.mm, p.nlay, p.ndeg, CASES[a.case][1]),
                   "case": a.case, "global_cells": p.ndeg, "layers": p.nlay, "parallelism": "j-slab x%d" % n_gpus,
                   "dense_fast_path": bool(dense), "setup_s": round(t_setup, 1),
                   "prewarm": {"seconds": a.prewarm_s, "steps": prewarm_steps,
                               "note": "untimed; the initial state is uploaded again before the W warm-up steps"},
                   "gpu_at_start": s0, "gpu_at_end": s1, "state_after_timed_region": state_check},
        "roofline": roof,
    }
    if halo:
        out["config"]["halo"] = halo
    out["cpu_baseline"] = cpu
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
