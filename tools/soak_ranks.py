#!/usr/bin/env python3
"""Soak of the rank-local band path: 3 processes on one GPU (shared-memory transport), 400 steps in uneven calls, owned
rows, histories and ghost rows against the single handle bit for bit — closed frame, sill with sponges, the y-periodic jet
(ring + companion frame) and the jet with open boundaries (mcbc = 0).  python tools/soak_ranks.py [steps=400] [world=3]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_bands_multiproc as T

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    calls = tuple([97] * (N // 97) + ([N % 97] if N % 97 else []))
    for case in ("closed", "sill_nudged", "jet_ring", "jet_ring_obc"):
        t = time.time()
        T._run(world, case, overlap=True, calls=calls)          # raises if any rank's comparison fails
        print(json.dumps({"case": case, "processes": world, "steps": N, "calls": list(calls), "bitwise_equal": True,
                          "seconds": round(time.time() - t, 1)}), flush=True)
