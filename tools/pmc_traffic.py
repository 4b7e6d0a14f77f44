#!/usr/bin/env python3
"""Turns rocprofv3 PMC passes into per-launch HBM traffic per kernel class.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -- python3 bench.py ...
  python tools/pmc_traffic.py DIR_F DIR_W > profiles/traffic_latest.json

Units and corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it is
doubled.  The factor was calibrated for this code's 8-byte-per-lane loads on
update_viscosity, whose compulsory read set is exactly rvor + dive (2 words per
cell-layer): 2 x FETCH_SIZE reproduces it within 7 %.  WRITE_SIZE is exact.
The first half of each kernel's launches (warm-up, steps 1-3) is skipped.
"""
import collections
import csv
import glob
import json
import sys

CLASS = {"k_update_h": "update_h", "k_update_mont": "update_mont",  # first match wins: order matters "k_update_visc": "update_viscosity",
         "k_update_uv<CellDenseT<false>, true>": "update_u", "k_update_uv<CellDenseT<false>, false>": "update_v",
         "k_update_uv<CellGather, true>": "update_u", "k_update_uv<CellGather, false>": "update_v",
         "k_mont_visc": "update_mont+update_viscosity", "k_uv_fused": "update_u+update_v"}


def classify(name):
    for k, v in CLASS.items():
        if k in name:
            return v
    return None


def load(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        c = classify(r["Kernel_Name"])
        if c:
            agg[c].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in agg.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        out[k] = rd + wr
        out[k + "_detail"] = {"read_bytes_2xFETCH": rd, "write_bytes": wr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
