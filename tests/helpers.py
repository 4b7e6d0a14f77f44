"""Shared test helpers: golden fixtures (tests/golden/*.npz, generated from the real
reference by tests/golden/make_golden.py) and comparisons."""
from __future__ import annotations

import glob
import json
import os

import copy

import numpy as np

from beom_amd.grid import read_input_data, restart_from_files
from beom_amd.params import Params

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_STEPS = (1, 2, 3, 4, 5, 10)
STATE = ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "v_cc", "v_ll", "tt3d", "tb3d", "tu3d")
SCRATCH = ("mont", "rvor", "pvor", "dive", "d2hx", "d2hy")


def _fname(key):
    return key.replace("_bin", ".bin").replace("_txt", ".txt")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


_FIELDS = {}


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.p = Params.from_json(json.loads(str(self.z["params_json"])))
        self.variant = 1 if str(self.z["engine"]) == "private_mod3d.f95" else 0
        self.files = {k[3:]: self.z[k] for k in self.z.files if k.startswith("in_")}
        # restart fixtures (rsta = 1): the output files of the run that is continued, as the reference found them in odir
        self.pre = {_fname(k[9:]): (str(self.z[k]) if k.endswith("_txt") else self.z[k].tobytes())
                    for k in self.z.files if k.startswith("pre_file_")}

    def fields(self):
        """The init mirror's module state for this fixture.  Built once per session and name (the outcropping rest state is an
        iteration in numpy: seconds); callers get their own shallow copy — they set attributes (invf, p), never array elements."""
        if self.name not in _FIELDS:
            f = read_input_data(self.p, files=self.files)
            if float(self.p.rsta) > 0.5:
                restart_from_files(f, self.pre)
            _FIELDS[self.name] = f
        return copy.copy(_FIELDS[self.name])

    def static(self, key):
        return self.z["static_" + key]

    def step(self, t, key):
        return self.z["step%d_%s" % (t, key)]

    def uses_cos(self):
        return "tide" in self.files


def same(a, b):
    """Numerically identical (signed zeros compare equal, NaNs in the same places)."""
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def same_bits(a, b):
    """Bit-identical, the sign of zero included."""
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))


def maxrel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / scale
