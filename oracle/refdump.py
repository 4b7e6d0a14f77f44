"""TEST INFRASTRUCTURE — readers for the FP64 dumps written by the P4 hooks of
oracle/ref_build.py (``oracle_static.bin``, ``oracle_step_NNNNNN.bin``) and for the
compact golden fixtures derived from them (tests/golden/*.npz).  Layouts follow the
``write`` statements in ref_build.DUMP_CODE; arrays come back in the C-ordered shapes
documented in beom_amd/grid.py."""
from __future__ import annotations

import numpy as np


class _Rd:
    def __init__(self, path):
        self.buf = np.fromfile(path, dtype=np.uint8)
        self.pos = 0

    def take(self, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = self.buf[self.pos:self.pos + n].view(dtype).reshape(shape)
        self.pos += n
        return a.copy()

    def done(self):
        assert self.pos == self.buf.size, (self.pos, self.buf.size)


def read_static(path):
    r = _Rd(path)
    lm, mm, nlay, ndeg = (int(x) for x in r.take("<i4", (4,)))
    n1 = ndeg + 1
    d = dict(lm=lm, mm=mm, nlay=nlay, ndeg=ndeg)
    d["neig"] = r.take("<i4", (n1, 8))
    d["subc"] = r.take("<i4", (2, n1))
    for k in ("mk_u", "mk_v", "mk_n", "mkpe", "mkpi", "fcor", "h_th", "h_to"):
        d[k] = r.take("<f8", (n1,))
    d["nudg"] = r.take("<f8", (3, n1))
    d["fnud"] = r.take("<f8", (3, nlay, n1))
    d["hdot"] = r.take("<f8", (nlay, n1))
    d["tide"] = r.take("<f8", (3, n1, 1, 2))
    d["w_ti"] = r.take("<f8", (1,))
    d["bodf"] = r.take("<f8", (2, nlay))
    d["taus"] = r.take("<f8", (2, n1))
    d["invf"], d["dt"] = (float(x) for x in r.take("<f8", (2,)))
    for k in ("hlay", "u", "v"):
        d[k] = r.take("<f8", (nlay, n1))
    if r.pos < r.buf.size:                    # rgld = 1: lid pressure and its operators
        for k in ("pi_s", "Ow", "Os", "Osum_"):
            d[k] = r.take("<f8", (n1,))
    r.done()
    return d


STEP_FIELDS = ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "v_cc", "v_ll",
               "mont", "rvor", "pvor", "dive", "d2hx", "d2hy", "tt3d", "tb3d", "tu3d")


def read_step(path, nlay, ndeg):
    r = _Rd(path)
    n1 = ndeg + 1
    d = {}
    for k in ("hlay", "u", "v", "h_u", "h_v"):
        d[k] = r.take("<f8", (nlay, n1))
    d["rs_h"] = r.take("<f8", (nlay, n1, 2))
    d["dmdx"] = r.take("<f8", (nlay, n1, 3))
    d["dmdy"] = r.take("<f8", (nlay, n1, 3))
    d["v_cc"] = r.take("<f8", (nlay, n1))
    d["v_ll"] = r.take("<f8", (nlay, n1))
    for k in ("mont", "rvor", "pvor", "dive", "d2hx", "d2hy"):
        d[k] = r.take("<f8", (n1,))          # 2-D scratch: values of the LAST layer processed
    for k in ("tt3d", "tb3d", "tu3d"):
        d[k] = r.take("<f8", (nlay, 2, n1))
    d["ctim"], d["ramp"], d["gene"] = (float(x) for x in r.take("<f8", (3,)))
    left = (r.buf.size - r.pos) // 8
    if left >= 4 * nlay * n1:                 # svis > 0: biharmonic work arrays
        for k in ("delu", "delv", "uu4", "vv4"):
            d[k] = r.take("<f8", (nlay, n1))
    if r.pos < r.buf.size:                    # rgld = 1: the lid pressure
        d["pi_s"] = r.take("<f8", (n1,))
    r.done()
    return d
