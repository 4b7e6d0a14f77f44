"""BASELINE.json's configurations at their FULL sizes on the GPU.

configs[1] soliton 2048x256x1, configs[2] unstable_jet 2048x2048x2, configs[3]
sill_exchange3D 4096x512x4 and the headline 4096x4096x4: the HIP engine against the oracle on the
same inputs, bit for bit (the OpenMP oracle steps even the headline frame in well under a second
per step on the host cores).  On the headline frame size-independent properties of the scheme are
checked as well:
  * volume conservation: the flux form of update_h (:1612-1622) telescopes over a closed
    basin, so the sum of every layer's thickness is constant to rounding (doc Test-case 3:
    'mean layer thickness stays within 1e-10 m');
  * mirror symmetry: the initial mound is centred, f-plane rotation breaks mirror symmetry
    but the point symmetry (i,j) -> (lm+1-i, mm+1-j) of h is kept exactly by the stencils up
    to rounding;
  * dense path == gather path on a band of rows would need two 16 GB states, so instead the
    fused and the unfused sweeps are compared bitwise after the same 6 steps.
configs[4] (8192x8192x8, 8 GPUs): ONE GPU's share of it — carrier beach 8192x1024x8, ocrp = 1, western
sponge, 8 layers — against the oracle bit for bit, and the same frame as 8 row bands (beom_multi_*)
against the single handle; the recipe is also covered at 120x3 by the golden fixture and at 2048x64x2
here, and tools/config5_slab_size.py runs the full frame on one GPU (profiles/r01_config5_full_frame.txt)."""
import numpy as np
import pytest

import oracle_lib
from beom_amd import capi, inputs as I
from beom_amd.grid import read_input_data
from helpers import same

pytestmark = pytest.mark.gpu
PROG = ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy")


def _vs_oracle(p, files, nsteps):
    f = read_input_data(p, files=files)
    e = capi.Engine(f)
    assert e.is_dense
    o = oracle_lib.Oracle(f, per_layer_scratch=False)
    e.step(1, nsteps)
    o.step(1, nsteps)
    st = e.download(PROG)
    for k in PROG:
        assert same(st[k], o.state()[k]), k
    assert np.isfinite(st["hlay"]).all()
    e.close()


def test_config1_soliton_2048x256_vs_oracle():
    p, files = I.case_soliton(lm=2048, mm=256, dt_s=60.0)
    _vs_oracle(p, files, 40)


def test_config2_unstable_jet_2048x2048x2_vs_oracle():
    p, files = I.case_unstable_jet(lm=2048, mm=2048, nlay=2, dt_s=50.0)
    _vs_oracle(p, files, 8)


def test_config3_sill_exchange3d_4096x512x4_vs_oracle():
    p, files = I.case_sill_exchange3d(lm=4096, mm=512, nlay=4, dt_s=30.0, npts=15, sill_halfwidth=50.0)
    _vs_oracle(p, files, 6)


def test_config4_recipe_carrier_beach_2048x64x2_vs_oracle():
    p, files = I.case_carrier_beach(lm=2048, mm=64, nlay=2, dt_s=0.08)
    _vs_oracle(p, files, 12)


def test_config5_share_carrier_beach_8192x1024x8_vs_oracle_and_bands():
    """BASELINE configs[4] per-GPU share: 8 layers (k_mont_visc<8,*>, the 8-deep register column),
    outcropping (ocrp = 1, Salmon term), W sponge, no Leith refresh after step 3."""
    p, files = I.case_carrier_beach(lm=8192, mm=1024, nlay=8, dt_s=0.08)
    f = read_input_data(p, files=files)
    del files
    e = capi.Engine(f)
    assert e.is_dense
    o = oracle_lib.Oracle(f, per_layer_scratch=False)
    e.step(1, 7)
    o.step(1, 7)
    st = e.download(PROG)
    for k in PROG:
        assert same(st[k], o.state()[k]), k
    assert np.isfinite(st["hlay"]).all() and float(np.max(np.abs(st["u"]))) > 0.0
    del o
    many = capi.MultiEngine(f, devices=[0] * 8)
    many.step(1, 7)
    sb = many.download(PROG)
    for k in PROG:
        assert same(st[k], sb[k]), ("bands", k)
    assert many.stats()["split"] >= 3 * 8
    many.close(); e.close()


def test_headline_4096x4096x4_vs_oracle_and_properties():
    p, files = I.case_headline(4096, 4096, 4)
    f = read_input_data(p, files=files)
    del files
    e = capi.Engine(f)
    assert e.is_dense
    lm, mm, nlay = p.lm, p.mm, p.nlay
    vol0 = [float(np.sum(f.hlay[k], dtype=np.longdouble)) for k in range(nlay)]
    e.step(1, 6)
    st = e.download(PROG)
    # (0) the oracle on the same inputs, steps 1-6 (both u/v orders, gene 0 -> g_fb), bit for bit
    o = oracle_lib.Oracle(f, per_layer_scratch=False)
    o.step(1, 6)
    for k in PROG:
        assert same(st[k], o.state()[k]), k
    del o
    # (a) volume of every layer
    for k in range(nlay):
        vol = float(np.sum(st["hlay"][k], dtype=np.longdouble))
        assert abs(vol - vol0[k]) <= 1e-12 * abs(vol0[k]), (k, vol, vol0[k])
    # (b) point symmetry of h about the basin centre (interior cells i=1..lm, j=1..mm)
    h = st["hlay"][:, 1:].reshape(nlay, mm + 1, lm + 1)[:, :mm, :lm]
    asym = np.max(np.abs(h - h[:, ::-1, ::-1]))
    assert asym <= 1e-9, asym                       # metres, on 1000-m-thick layers
    # (c) fused sweeps == unfused sweeps, bitwise, from the same start
    g = capi.Engine(f)
    g.set_option("fuse", 0)
    g.step(1, 6)
    sg = g.download(("hlay", "u", "v"))
    for k in ("hlay", "u", "v"):
        assert same(st[k], sg[k]), k
    assert float(np.max(np.abs(st["u"]))) > 0.0
    g.close()
    # (d) the frame as 2 and as 4 row bands — what `bench.py --gpus 2 | 4` gives each GPU: bands tall enough for the 64 x 8
    #     tile geometry, every step cut boundary first — bit for bit, the sign of zero included
    from helpers import same_bits
    for nb in (2, 4):
        many = capi.MultiEngine(f, devices=[0] * nb)
        assert many.info("tile_rows") == 8
        many.step(1, 6)
        sb = many.download(PROG)
        for k in PROG:
            assert same_bits(st[k], sb[k]), (nb, "bands", k)
        assert many.stats() == {"split": 6 * nb, "plain": 0}
        many.close()
    e.close()
