"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the
C-ABI of include/beom_hip.h, against (a) the golden vectors of the REAL reference and
(b) the oracle on the same inputs.  Bar: bit-exact FP64 (numeric equality; the sign of
zero is not compared) for every configuration; configurations with a tidal term call
cos() whose device implementation is not glibc's, tolerance 1e-12 relative there."""
import numpy as np
import pytest

import oracle_lib
from beom_amd import capi
from helpers import GOLDEN_STEPS, SCRATCH, STATE, Golden, golden_names, maxrel, same, same_bits

pytestmark = pytest.mark.gpu
NAMES = golden_names()
COS_TOL = 1e-12


def _fields(g):
    f = g.fields()
    f.invf = float(g.static("invf"))
    return f


def _check(a, b, exact, what):
    if exact:
        assert same(a, b), (what, maxrel(a, b))
    else:
        assert maxrel(a, b) <= COS_TOL, (what, maxrel(a, b))


MODES = {          # name: (dense_hint, fuse_mont_visc, fuse_uv, keep_diag)
    "gather": (0, 1, 1, 0),
    "dense_unfused": (1, 0, 0, 0),
    "dense_fused_keepdiag": (1, 1, 1, 1),
    "dense_fuse_mv_only": (1, 1, 0, 0),
    "dense_fuse_uv_only": (1, 0, 1, 0),
    "dense_fused": (1, 1, 1, 0),       # the production default: mont+visc and u+v as two fused sweeps
}
PROGNOSTIC = ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy", "tt3d", "tb3d", "tu3d")


def _live(e, keys):
    """`keys` without the stress arrays when the engine forms distribute_stress inside its momentum sweep (option
    "fold_stress": constant layer fractions, a refresh on every step): tt3d, tb3d, tu3d are then neither written nor read —
    the next step recomputes what they would hold — and keep their values of step 1."""
    return [k for k in keys if not (k in ("tt3d", "tb3d", "tu3d") and e.info("stress_folded"))]


def _engine(f, variant=0, mode="dense_fused", **kw):
    dh, fmv, fuv, keep = MODES[mode]
    e = capi.Engine(f, variant=variant, dense_hint=dh, **kw)
    e.set_option("fuse_mont_visc", fmv)
    e.set_option("fuse_uv", fuv)
    e.set_option("keep_diag", keep)
    return e


def _fuses(p):
    """Montgomery(+Leith) runs as the fused sweep for this configuration (beom_engine.hip can_fuse): the
    viscosity is refreshed every step, or never after step 3."""
    if float(p.svis) > 0.0 or p.nlay > 8 or float(p.rgld) > 0.5:      # (a rigid-lid handle keeps the separate sweeps)
        return False
    return True


def _fusion_active(g, e):
    return e.is_dense and _fuses(g.p)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", NAMES)
def test_step_matches_reference_golden(name, mode):
    """beom_step over steps 1..10 == FP64 module state of the reference run.  In the default
    fused mode v_cc, v_ll, rvor, dive are not kept (only their products are formed), so the
    check covers everything that determines the future: prognostic fields and histories."""
    g = Golden(name)
    e = _engine(_fields(g), variant=g.variant, mode=mode)
    exact = not g.uses_cos()
    lossy = mode in ("dense_fused", "dense_fuse_mv_only") and _fusion_active(g, e)
    t = 0
    for tgt in GOLDEN_STEPS:
        e.step(t + 1, tgt - t)
        t = tgt
        st = e.download()
        for k in _live(e, PROGNOSTIC if lossy else STATE):
            _check(st[k], g.step(tgt, k), exact, (name, tgt, k))
        if exact:                                  # prognostic fields and the histories the next steps read: even the sign of zero
            for k in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy"):
                assert same_bits(st[k], g.step(tgt, k)), (name, tgt, k, "sign of zero")
        sc = e.download_scratch()
        keys = SCRATCH if not lossy else (("mont", "pvor") if mode == "dense_fused" else ("mont", "pvor", "d2hx", "d2hy"))
        for k in keys:                                                        # reference scratch = last layer
            _check(sc[k][g.p.nlay - 1], g.step(tgt, k), exact, (name, tgt, k))
        if float(g.p.rgld) > 0.5:                  # rigid lid: the pressure the Gauss-Seidel sweeps converged to
            assert same_bits(e.download_pressure(), g.step(tgt, "pi_s")), (name, tgt, "pi_s")
    e.close()


@pytest.mark.parametrize("name", NAMES)
def test_dense_detection(name):
    g = Golden(name)
    f = _fields(g)
    e = capi.Engine(f, variant=g.variant, dense_hint=1)
    full = g.p.ndeg == (g.p.lm + 1) * (g.p.mm + 1)
    assert (e.is_dense and not e.is_embedded) == full
    # frames with land run on the same rectangle ("embedded") unless they use what only the table path has
    # (biharmonic viscosity), the rectangle is mostly land, or a coast cell sits on a
    # periodic seam (the reference wraps row by row there, :614-640: offsets cannot express that)
    obc = bool(f.flag_nudging) and float(g.p.mcbc) < 0.5
    periodic = float(g.p.xper) > 0.5 or float(g.p.yper) > 0.5
    if not full and not periodic and float(g.p.svis) == 0.0 and g.p.ndeg * 10 >= (g.p.lm + 1) * (g.p.mm + 1) * 3:
        assert e.is_embedded, name
    e.close()


@pytest.mark.parametrize("name", ["jet_2l_xyper", "sill_4l_ocrp", "island_3l_forced"])
def test_per_sweep_entry_points_match_oracle(name):
    """update_h / update_mont / update_viscosity / update_u / update_v one layer at a
    time, in the reference's order (private_mod.f95:2259-2290), vs the oracle sweeps."""
    g = Golden(name)
    f = _fields(g)
    p = g.p
    e = capi.Engine(f, variant=g.variant)
    o = oracle_lib.Oracle(f, variant=g.variant)
    for tstp in range(1, 7):
        ctim = float(p.dtd8) * tstp
        first3 = tstp <= 3
        c = float(p.dtd8) * (1 if first3 else tstp)
        ramp = c / float(p.dt_r) if (float(p.rsta) < 0.5 and c < float(p.dt_r)) else 1.0
        gene = 0.0 if first3 else float(p.g_fb)
        upst = tstp == 1 or (not first3 and tstp % p.n_3d == 0)
        for x in (e, o):
            if upst:
                x.distribute_stress()
            if first3:
                x.rebuild_fluxes()
            x.update_h(gene, ramp, ctim)
        for il in range(1, p.nlay + 1):
            for x in (e, o):
                x.update_mont(il)
                if first3 or (float(p.dvis) > 1e-3 and upst):
                    x.update_viscosity(il)
            sc = e.download_scratch()
            for k in SCRATCH:
                assert same(sc[k][il - 1], o.a[k]), (tstp, il, k)
            for w in (("u", "v") if tstp % 2 == 0 else ("v", "u")):
                for x in (e, o):
                    getattr(x, "update_" + w)(il, gene, ramp, ctim)
        e.sync()
        st = e.download()
        for k in STATE:
            assert same(st[k], o.state()[k]), (tstp, k)
    e.close()


def test_long_run_matches_oracle():
    """200 steps of the island case (wind, drag, ramp, dt3d cadence): still bit-exact."""
    g = Golden("island_3l_forced")
    f = _fields(g)
    e = capi.Engine(f)
    o = oracle_lib.Oracle(f)
    e.step(1, 200)
    o.step(1, 200)
    st = e.download()
    for k in _live(e, STATE):
        assert same(st[k], o.state()[k]), k
    assert np.isfinite(st["hlay"]).all()
    e.close()


@pytest.mark.parametrize("case", ["island_leith", "island_wind_drag", "bay_ocrp_nudged", "sponge_obc_island"])
def test_land_frames_on_the_rectangle_match_oracle_and_table_path(case):
    """Frames WITH land wide enough to have regular tiles away from the coast: the embedded form (packed cells in the
    slots of their (i, j), land slots holding the sentinel's values, masks from the caller's arrays in the tiles that
    touch land, the fused mask-free paths elsewhere) against the oracle and against the table path, bit for bit."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    if case == "island_leith":
        p, files = I.case_headline(400, 130, 3)
    elif case == "island_wind_drag":
        p, files = I.case_stommel(lm=400, mm=130, dl=50.0e3, dt_s=0.2)
        files = dict(files, h_bo=np.full((402, 132), 200.0))
    elif case == "sponge_obc_island":                # nudged open boundaries with mcbc = 0 (no_gradient_obc, :2613-2679) around an island
        p, files = I.case_wave_sponge(lx=1400.0e3, ly=1100.0e3)
        p = p.replace(mcbc="0.")
        depth = np.zeros((p.lm + 2, p.mm + 2)); depth[1:-1, 1:-1] = float(p.cext) ** 2 / float(p.grav)     # the default flat depth (:121), as a file
        files = dict(files, h_bo=depth)
    else:
        p, files = I.case_sill_exchange3d(lm=400, mm=130, nlay=3, dt_s=0.01, npts=5, sill_halfwidth=20.0)
    files = {k: np.array(v, dtype=np.float64) for k, v in files.items()}
    h = files["h_bo"]
    x = np.arange(p.lm + 2)[:, None]; y = np.arange(p.mm + 2)[None, :]
    land = ((x - 0.3 * p.lm) ** 2 + (y - 0.55 * p.mm) ** 2) < (0.12 * p.lm) ** 2          # an island
    land |= (x > 0.8 * p.lm) & (y < 0.3 * p.mm) & ((x + y) % 7 != 0) if case == "bay_ocrp_nudged" else False   # a ragged corner
    h[land] = 0.0
    if "init" in files:
        files["init"][land] = 0.0
    p = p.replace(ndeg=I.get_nbr_deg_freedom(h))
    f = read_input_data(p, files=files)
    emb, tab, o = capi.Engine(f), capi.Engine(f, dense_hint=0), oracle_lib.Oracle(f)
    assert emb.is_embedded and not tab.is_dense
    for x_ in (emb, tab, o):
        x_.step(1, 13)
    a, b = emb.download(), tab.download()
    for k in _live(emb, PROGNOSTIC):
        assert same(a[k], o.state()[k]), (case, k, "embedded vs oracle", maxrel(a[k], o.state()[k]))
        assert same(a[k], b[k]), (case, k, "embedded vs table path")
    for k in ("hlay", "u", "v", "h_u", "h_v"):
        assert same_bits(a[k], o.state()[k]), (case, k, "sign of zero")
    eta, u4, v4, mm_, thin = emb.download_outputs(np.ascontiguousarray(f.h_0[:, 1:], dtype=np.float32))
    eta2, u42, v42, mm2, thin2 = tab.download_outputs(np.ascontiguousarray(f.h_0[:, 1:], dtype=np.float32))
    assert np.array_equal(eta.view(np.uint32), eta2.view(np.uint32)) and np.array_equal(mm_, mm2) and thin == thin2
    emb.close(); tab.close()


@pytest.mark.parametrize("case", ["sill_3l", "basin_coast_wind_2l", "sill_xper_2l"])
def test_rigid_lid_larger_frames_match_oracle(case):
    """rgld = 1 (private_mod.f95:1648-1700, 1705-1838, 2207-2221, 2237-2257, 2292-2314) on frames wider than one tile,
    with land (the rectangle form and the table path; a straight coast — the reference's operators are 1/0 at any coast
    cell whose four faces are dry, e.g. around an island) and with an x-periodic seam (where the reference's serial scatter
    of the Poisson right-hand side meets its eastern neighbour BEFORE its own term; the reference's lid is not made for
    periodic frames and the run diverges, so only the 7 steps that stay finite): states, transports and the lid pressure
    the Gauss-Seidel wavefronts converge to, bit for bit against the oracle's serial sweeps, through a restart."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    if case == "sill_3l":
        p, files = I.case_sill_exchange3d(lm=133, mm=41, nlay=3, dt_s=0.01, npts=5, sill_halfwidth=6.0)
        p = p.replace(rgld="1.")
    elif case == "sill_xper_2l":
        p, files = I.case_sill_exchange3d(lm=40, mm=25, nlay=2, dt_s=0.01, npts=5, sill_halfwidth=5.0)
        p = p.replace(rgld="1.", xper="1.")
    else:
        p, files = I.case_headline(150, 60, 2)
        p = p.replace(rgld="1.", ocrp="1.", bdrg="2.e-4", tauw=["0.05", "0.02"], g_fb="0.")
        files = {k: np.array(v, dtype=np.float64) for k, v in files.items()}
        h = files["h_bo"]
        h[:, int(0.8 * p.mm) + 1:] = 0.0             # land north of row 0.8 mm
        p = p.replace(ndeg=I.get_nbr_deg_freedom(h))
    f = read_input_data(p, files=files)
    assert float(f.p.rgld) > 0.5 and np.abs(f.Osum_).max() > 0 and np.isfinite(f.Osum_).all()
    fast, tab, o = capi.Engine(f), capi.Engine(f, dense_hint=0), oracle_lib.Oracle(f)
    assert fast.is_dense and not tab.is_dense and fast.is_embedded == (case == "basin_coast_wind_2l")
    t = 1
    for n in ((3, 4) if case == "sill_xper_2l" else (3, 4, 6)):
        for x_ in (fast, tab, o):
            x_.step(t, n)
        t += n
        if n == 4:                                   # scatter state and pressure again
            fast.upload(**fast.download()); fast.upload_pressure(fast.download_pressure())
        for e in (fast, tab):
            st = e.download()
            for k in PROGNOSTIC:
                assert same(st[k], o.state()[k]), (case, k, t, maxrel(st[k], o.state()[k]))
            for k in ("hlay", "u", "v", "h_u", "h_v"):
                assert same_bits(st[k], o.state()[k]), (case, k, "sign of zero")
            assert same_bits(e.download_pressure(), o.rgld["pi_s"]), (case, t, "pi_s")
    eta, _, _, _, _ = fast.download_outputs(np.ascontiguousarray(f.h_0[:, 1:], dtype=np.float32))
    assert np.array_equal(eta[0].view(np.uint32), o.rgld["pi_s"][1:].astype(np.float32).view(np.uint32))     # :2864-2872
    fast.close(); tab.close()


def test_rigid_lid_pipeline_of_sweeps_at_size():
    """surf_pressure's Gauss-Seidel iteration as a pipeline of wavefronts (several sweeps in flight, one copy of the pressure
    per sweep; beom_engine.hip lid_solve) on a frame of half a million cells, with a wind strong enough for dozens of sweeps
    per step: lid pressure and state bit for bit against the oracle's serial sweeps, the count of sweeps kept included."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    p, files = I.case_headline(1024, 512, 2)
    p = p.replace(rgld="1.", ocrp="1.", g_fb="0.", bdrg="2.e-4", tauw=["0.5", "0.2"])
    f = read_input_data(p, files=files)
    e, o = capi.Engine(f), oracle_lib.Oracle(f)
    assert e.info("lid_sweep_distance") == 2          # a plain frame: sweep s + 1 follows two levels behind sweep s
    for t, n in ((1, 3), (4, 5)):
        e.step(t, n); o.step(t, n)
        assert same_bits(e.download_pressure(), o.rgld["pi_s"]), t
        st = e.download()
        for k in ("hlay", "u", "v", "h_u", "h_v"):
            assert same_bits(st[k], o.state()[k]), (t, k)
    assert e.info("lid_solves") == 8 and e.info("lid_sweeps") > 8
    e.close()


def test_restart_split_equals_single_run():
    """download → new handle → upload → continue == uninterrupted run (state incl. histories)."""
    g = Golden("sill_2l_ocrp")
    f = _fields(g)
    a = capi.Engine(f)
    a.step(1, 9)
    b = capi.Engine(f)
    b.step(1, 5)
    mid = b.download()
    c = capi.Engine(f, upload=False)
    c.upload(**mid)
    c.step(6, 4)
    sa, sc = a.download(), c.download()
    for k in STATE:
        assert same(sa[k], sc[k]), k
    for x in (a, b, c):
        x.close()


@pytest.mark.parametrize("case", ["closed", "sill_ocrp", "beach_zero_visc", "soliton_zero_visc"])
def test_lean_thickness_curvature_matches_oracle(case):
    """Production pair of fused sweeps on a frame with DEEP tiles (>= 3 tiles away from every
    edge): k_mont_visc stores d2hx/d2hy only around non-interior tiles, k_uv_fused re-derives
    them from the staged hlay.  Prognostic state must equal the oracle and the run with
    lean_d2h = 0 bit for bit; the curvature arrays really are left stale in deep tiles."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    if case == "closed":
        p, files = I.case_headline(330, 75, 2)
    elif case == "beach_zero_visc":        # dvis = bvis = 0: the fused pair also drops the (+-0) viscous products
        p, files = I.case_carrier_beach(lm=330, mm=75, nlay=2, dt_s=0.08)
    elif case == "soliton_zero_visc":      # ... at rest far from the soliton: many exactly-zero right-hand sides
        p, files = I.case_soliton(lm=331, mm=75, dt_s=5.0)
    else:
        p, files = I.case_sill_exchange3d(lm=330, mm=75, nlay=3, dt_s=0.01, npts=5, sill_halfwidth=20.0)
    f = read_input_data(p, files=files)
    lean, full = capi.Engine(f), capi.Engine(f)
    full.set_option("lean_d2h", 0)
    full.set_option("lean_visc", 0)
    assert lean.is_dense
    o = oracle_lib.Oracle(f)
    n = 14
    for x in (lean, full, o):
        x.step(1, n)
    sl, sf = lean.download(), full.download()
    for k in PROGNOSTIC:
        assert same(sl[k], o.state()[k]), (case, k, maxrel(sl[k], o.state()[k]))
        assert same(sf[k], sl[k]), (case, k)
    for k in ("hlay", "u", "v", "h_u", "h_v"):            # the sign of zero too
        assert same_bits(sl[k], o.state()[k]), (case, k, "sign of zero")
    cl, cf = lean.download_scratch(), full.download_scratch()
    L = p.lm + 1
    ip = 200 + (40 - 1) * L                      # cell (200, 40): tile x0 = 193, y0 = 33 is deep
    if case == "closed":                         # (the sill frame is flat there: both are zero)
        assert not np.array_equal(cl["d2hx"][:, ip - 3:ip + 3], cf["d2hx"][:, ip - 3:ip + 3])
    assert same(cl["mont"], cf["mont"]) and same(cl["pvor"], cf["pvor"])
    lean.close(); full.close()


def test_unsupported_options_fail_loudly():
    g = Golden("stommel_24x16")
    f = _fields(g)
    assert float(f.p.ocrp) < 0.5
    f.p = f.p.replace(rgld="1.")        # rigid lid without outcropping: the reference never initialises the operators (:505)
    with pytest.raises(capi.BeomError):
        capi.Engine(f)


def test_rigid_lid_needs_its_operators():
    """A rgld = 1 handle refuses to step until beom_set_rigid_lid has been called, and a free-surface handle
    refuses the lid calls."""
    g = Golden("rigid_lid_sill_2l")
    f = _fields(g)
    lib = capi.load()
    e = capi.Engine.__new__(capi.Engine)
    import ctypes as C
    e.lib, e.f, e.p, e.device = lib, f, f.p, 0
    e.prm = capi.make_params_struct(f.p, f, 0, 1, 0, 0)
    e._err = C.create_string_buffer(capi.ERRLEN + 1)
    e.h = C.c_void_p()
    opt = lambda k: capi._dp(getattr(f, k)) if f.has.get(k, True) else None
    rc = lib.beom_create(C.byref(e.prm), 0, capi._ip(f.neig), capi._ip(f.subc), capi._dp(f.mk_u), capi._dp(f.mk_v),
                         capi._dp(f.mk_n), capi._dp(f.mkpe), capi._dp(f.mkpi), capi._dp(f.fcor), capi._dp(f.h_th),
                         capi._dp(f.h_to), capi._dp(f.nudg), capi._dp(f.fnud), opt("hdot"), opt("tide"), opt("bodf"),
                         capi._dp(f.taus), C.byref(e.h), e._err, capi.ERRLEN)
    assert rc == 0, e._err.value
    e.upload(**{k: getattr(f, k) for k in capi.STATE_NAMES})
    with pytest.raises(capi.BeomError, match="beom_set_rigid_lid"):
        e.step(1, 1)
    e.close()
    g2 = Golden("stommel_24x16")
    e2 = capi.Engine(_fields(g2))
    with pytest.raises(capi.BeomError):
        e2.download_pressure()
    e2.close()


def _big_cases():
    from beom_amd import inputs as I
    return {
        "closed_3l": lambda: I.case_headline(150, 37, 3),
        "closed_3l_biharm": lambda: (lambda pf: (pf[0].replace(svis="1.e9"), pf[1]))(I.case_headline(150, 37, 3)),
        "soliton_xper": lambda: I.case_soliton(lm=141, mm=23, dt_s=5.0),
        "jet_xyper_2l": lambda: I.case_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5),
        "sill_ocrp_nudg_4l": lambda: I.case_sill_exchange3d(lm=133, mm=41, nlay=4, dt_s=0.01, npts=5,
                                                            sill_halfwidth=6.0),
        "stommel_wind_drag": lambda: I.case_stommel(lm=200, mm=30, dl=50.0e3, dt_s=0.2),
        # 8 layers (BASELINE config 5's layer count): outcropping beach, and a closed basin with Leith viscosity
        "beach_ocrp_8l": lambda: I.case_carrier_beach(lm=150, mm=37, nlay=8, dt_s=0.08),
        "closed_8l": lambda: I.case_headline(150, 37, 8),
    }


@pytest.mark.parametrize("case", ["closed_3l", "closed_3l_biharm", "soliton_xper", "jet_xyper_2l",
                                  "sill_ocrp_nudg_4l", "stommel_wind_drag", "beach_ocrp_8l", "closed_8l"])
def test_dense_interior_waves_match_oracle_and_gather(case):
    """Grids wide enough (L >= 130) that most waves take the INTERIOR specialisation of
    CellDenseT; the dense path, the gather path and the oracle must agree bitwise."""
    from beom_amd.grid import read_input_data
    p, files = _big_cases()[case]()
    f = read_input_data(p, files=files)
    engines = {m: _engine(f, mode=m) for m in MODES}
    assert engines["dense_fused"].is_dense and not engines["gather"].is_dense
    o = oracle_lib.Oracle(f)
    for x in list(engines.values()) + [o]:
        x.step(1, 12)
    ref_sc = engines["gather"].download_scratch()
    for m, e in engines.items():
        lossy = m in ("dense_fused", "dense_fuse_mv_only") and _fuses(p)
        st = e.download()
        for k in _live(e, PROGNOSTIC if lossy else STATE):
            assert same(st[k], o.state()[k]), (case, m, k, maxrel(st[k], o.state()[k]))
        sc = e.download_scratch()
        keys = SCRATCH if not lossy else (("mont", "pvor") if m == "dense_fused" else ("mont", "pvor", "d2hx", "d2hy"))
        for k in keys:
            assert same(sc[k], ref_sc[k]), (case, m, k)
        e.close()


def _forced_cases():
    """Dense frames with constant layer fractions (ocrp = 0) and a stress refresh on every step (dt3d = 0): the engine forms
    distribute_stress inside its fused momentum sweep from step 4 on."""
    from beom_amd import inputs as I

    def basin(nlay, **lits):
        p, files = I.case_headline(150, 131, nlay)
        lm, mm = p.lm, p.mm
        x = np.arange(lm + 2)[:, None] / lm; y = np.arange(mm + 2)[None, :] / mm
        taus = np.zeros((lm + 2, mm + 2, 2))
        taus[:, :, 0] = 0.1 * np.cos(np.pi * y) * np.ones_like(x)
        taus[:, :, 1] = -0.03 * np.sin(2 * np.pi * x) * np.ones_like(y)
        init = np.array(files["init"], dtype=np.float64)
        init[:, :, 0, 1] += 0.2 * np.sin(3 * np.pi * y)                 # velocities for the drag to act on, top ...
        init[:, :, nlay - 1, 2] += -0.1 * np.cos(2 * np.pi * x)         # ... and bottom layer
        return p.replace(**lits), dict(files, taus=taus, init=init)

    return {
        "wind_only_2l": lambda: basin(2),
        "wind_linear_drag_1l": lambda: basin(1, bdrg="2.e-4"),
        "wind_quadratic_bottom_top_drag_3l": lambda: basin(3, bdrg="3.e-3", tdrg="2.e-3", qdrg="1.", dt_r="0.002"),
        "bottom_drag_only_3l_bodf": lambda: (lambda pf: (pf[0], dict({k: v for k, v in pf[1].items() if k != "taus"},
                                                                  bodf=np.array([[1e-7, 0.0], [0.0, 0.0], [0.0, -2e-7]]))))(basin(3, bdrg="1.e-3", qdrg="0.5")),
        "stommel": lambda: I.case_stommel(lm=150, mm=131, dl=50.0e3, dt_s=0.2),
        "mixed_open_bc_wind_sponges": lambda: I.case_mixed_open_bc(lm=150, mm=131, npts=15),
    }


@pytest.mark.parametrize("tile_rows", [4, 8])
@pytest.mark.parametrize("case", ["wind_only_2l", "wind_linear_drag_1l", "wind_quadratic_bottom_top_drag_3l",
                                  "bottom_drag_only_3l_bodf", "stommel", "mixed_open_bc_wind_sponges"])
def test_stress_folded_into_momentum_sweep(case, tile_rows):
    """distribute_stress (private_mod.f95:1921-2149) formed inside the fused u+v sweep (k_uv_fused_sf, both tile geometries):
    against the oracle, against the same engine with the fold off (its own launch + the three arrays) and on three bands, bit
    for bit with the sign of zero."""
    import os
    from beom_amd.grid import read_input_data
    p, files = _forced_cases()[case]()
    f = read_input_data(p, files=files)
    obc = bool(f.flag_nudging) and float(p.mcbc) < 0.5
    old = os.environ.get("BEOM_TILE4")
    os.environ["BEOM_TILE4"] = "1" if tile_rows == 4 else "0"          # (read when a handle is created)
    try:
        fold, plain, o = capi.Engine(f), capi.Engine(f), oracle_lib.Oracle(f)
        bands = None if obc else capi.MultiEngine(f, devices=[0, 0, 0])
    finally:
        if old is None: os.environ.pop("BEOM_TILE4")
        else: os.environ["BEOM_TILE4"] = old
    assert fold.info("tile_rows") == tile_rows
    plain.set_option("fold_stress", 0)
    n = 15
    for x in [fold, plain, o] + ([bands] if bands else []):
        x.step(1, 3); x.step(4, n - 3)
    assert fold.info("stress_folded") == 1 and plain.info("stress_folded") == 0
    a, b = fold.download(), plain.download()
    for k in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy"):
        assert same_bits(a[k], o.state()[k]), (case, k, "folded vs oracle", maxrel(a[k], o.state()[k]))
        assert same_bits(b[k], o.state()[k]), (case, k, "own launch vs oracle")
    for k in ("tt3d", "tb3d", "tu3d"):
        assert same(b[k], o.state()[k]), (case, k)               # kept current only without the fold
    if bands:
        assert bands.info("stress_folded") == 1 and bands.stats()["split"] == 3 * n
        c = bands.download()
        for k in ("hlay", "u", "v", "h_u", "h_v", "rs_h", "dmdx", "dmdy"):
            assert same_bits(c[k], o.state()[k]), (case, k, "three bands, folded")
        bands.close()
    fold.set_option("keep_diag", 1)                               # keeps the three arrays current again
    fold.step(n + 1, 2); o.step(n + 1, 2)
    a = fold.download()
    assert fold.info("stress_folded") == 0
    for k in STATE:
        assert same(a[k], o.state()[k]), (case, k, "keep_diag")
    fold.close(); plain.close()


@pytest.mark.parametrize("case,world", [("closed_3l", 2), ("closed_3l", 3), ("sill_ocrp_nudg_4l", 2),
                                        ("soliton_xper", 2), ("closed_3l_biharm", 3)])
def test_slabs_on_one_gpu_match_single_domain(case, world):
    """j-slab windows (beom_params.slab_row0/slab_mm: masks from global coordinates, dense
    fast path on every slab) stepped side by side on ONE GPU, ghost rows moved with the
    pack/unpack code of beom_amd.slab and a device copy in place of RCCL: owned rows must
    equal the single-domain GPU run bit for bit."""
    import torch
    from beom_amd import slab
    from beom_amd.grid import read_input_data
    p, files = _big_cases()[case]()
    f = read_input_data(p, files=files)
    whole = capi.Engine(f)
    geoms = slab.decompose(p.mm, p.lm, world)
    runs = []
    for g in geoms:
        lf = slab.slice_fields(f, g)
        e = capi.Engine(lf, slab_row0=g.row0, slab_mm=p.mm)
        assert e.is_dense
        runs.append(slab.SlabRunner(e, g, p.nlay, dist=None))
        e.set_stream(torch.cuda.current_stream().cuda_stream)     # this test drives everything on ONE stream
    # the tensors are views of the live device state, not copies
    t0 = runs[0].t["hlay"]
    keep = t0[0, 5].item()
    t0[0, 5] = 12345.0
    torch.cuda.synchronize()
    assert runs[0].engine.download(("hlay",))["hlay"][0, 5] == 12345.0
    t0[0, 5] = keep
    nsteps = 12
    for t in range(1, nsteps + 1):
        for r in runs:
            r.engine.step(t, 1, sync=False)
        for r in runs:
            r.pack_all()
        for k in range(world - 1):
            runs[k + 1].recv_s.copy_(runs[k].send_n)
            runs[k].recv_n.copy_(runs[k + 1].send_s)
        for r in runs:
            r.unpack_all()
    torch.cuda.synchronize()
    whole.step(1, nsteps)
    ref = whole.download()
    for r in runs:
        g = r.g
        a, b = 1 + (g.own0 - 1) * g.L, 1 + g.own1 * g.L
        la, lb = g.local_rows(g.own0, g.own1)
        st = r.engine.download()
        for k in STATE:
            if st[k].ndim == 3 and st[k].shape[-1] in (2, 3) and k in ("rs_h", "dmdx", "dmdy"):
                assert same(st[k][:, la:lb, :], ref[k][:, a:b, :]), (case, g.rank, k)
            else:
                assert same(st[k][..., la:lb], ref[k][..., a:b]), (case, g.rank, k)
        r.engine.close()
    whole.close()


def test_profile_start_stop_counts_launches():
    g = Golden("jet_2l_xyper")
    e = capi.Engine(_fields(g))
    e.step(1, 4)
    e.profile_start()
    e.step(5, 6, sync=False)
    ms, nl = e.profile_stop()
    assert nl == [6, 0, 0, 0, 0, 6, 6, 0]    # fused pairs: H, mont+visc, u+v
    assert all(m > 0 for i, m in enumerate(ms) if nl[i])
    e.set_option("fuse", 0)
    e.profile_start()
    e.step(17, 4, sync=False)
    ms, nl = e.profile_stop()
    assert nl == [4, 4, 4, 4, 4, 0, 0, 0] and all(m > 0 for m in ms[:5])
    e.set_option("fuse", 1)
    e.set_option("profile_stride", 4)        # only the steps with tstp % 4 == 0 are bracketed: 24 and 28 of 21..30
    e.profile_start()
    e.step(21, 10, sync=False)
    ms, nl = e.profile_stop()
    assert nl == [2, 0, 0, 0, 0, 2, 2, 0]
    e.close()


@pytest.mark.parametrize("tile_rows", [4, 8])
@pytest.mark.parametrize("case,world", [("closed_tall", 2), ("closed_tall", 3), ("sill_tall", 2), ("beach_tall_noleith", 2),
                                        ("soliton_tall_noleith", 3)])
def test_cut_step_matches_single_domain(case, world, tile_rows):
    """beom_step_phase: a band's step in three parts — everything up to the momentum sweeps on all rows, the momentum
    sweep on the strips next to the ghost zones, the same sweep on the rows in between (+ pointer rotations).  Here on one
    stream in program order (the row ranges and the rotations are what is tested; real stream concurrency: next test).
    Owned rows must equal the single-domain run bit for bit."""
    import torch
    from beom_amd import inputs as I, slab
    from beom_amd.grid import read_input_data
    if case == "closed_tall":
        p, files = I.case_headline(150, 131, 3)
    elif case == "beach_tall_noleith":      # dvis = 0: plain Montgomery sweep + fused u+v in the cut step
        p, files = I.case_carrier_beach(lm=140, mm=150, nlay=2, dt_s=0.08)
    elif case == "soliton_tall_noleith":
        p, files = I.case_soliton(lm=141, mm=151, dt_s=5.0)
    else:
        p, files = I.case_sill_exchange3d(lm=133, mm=141, nlay=2, dt_s=0.01, npts=5, sill_halfwidth=6.0)
    f = read_input_data(p, files=files)
    import os
    old = os.environ.get("BEOM_TILE4")
    os.environ["BEOM_TILE4"] = "1" if tile_rows == 4 else "0"          # (read when a handle is created; bands of the headline
    try:                                                               #  frame cut 2 or 4 ways run the 64 x 8 geometry)
        whole = capi.Engine(f)
        runs = []
        for g in slab.decompose(p.mm, p.lm, world):
            e = capi.Engine(slab.slice_fields(f, g), slab_row0=g.row0, slab_mm=p.mm)
            assert e.info("tile_rows") == tile_rows
            runs.append(slab.SlabRunner(e, g, p.nlay, dist=None))
            e.set_stream(torch.cuda.current_stream().cuda_stream)     # one stream: ordering by program order
    finally:
        if old is None: os.environ.pop("BEOM_TILE4")
        else: os.environ["BEOM_TILE4"] = old

    def move():
        for k in range(world - 1):
            runs[k + 1].recv_s.copy_(runs[k].send_n)
            runs[k].recv_n.copy_(runs[k + 1].send_s)

    nsteps, cut = 14, 0
    for t in range(1, nsteps + 1):
        for r in runs:
            if r.engine.step_phase(t, 1):
                cut += 1
                assert r.engine.step_phase(t, 2) and r.engine.step_phase(t, 3)
            else:
                r.engine.step(t, 1, sync=False)
        for r in runs: r.pack_all()
        move()
        for r in runs: r.unpack_all()
    assert cut == world * nsteps                     # every step of such a configuration can be cut, the first three too
    torch.cuda.synchronize()
    whole.step(1, nsteps)
    ref = whole.download()
    for r in runs:
        g = r.g
        a, b = 1 + (g.own0 - 1) * g.L, 1 + g.own1 * g.L
        la, lb = g.local_rows(g.own0, g.own1)
        st = r.engine.download()
        for k in PROGNOSTIC:
            if k in ("rs_h", "dmdx", "dmdy"):
                assert same_bits(st[k][:, la:lb, :], ref[k][:, a:b, :]), (case, g.rank, k)
            else:
                assert same_bits(st[k][..., la:lb], ref[k][..., a:b]), (case, g.rank, k)
        r.engine.close()
    whole.close()


def test_overlapped_streams_two_slabs_one_process():
    """The overlapped SlabRunner protocol with REAL stream concurrency on one GPU: each slab has its main and its second
    stream; the edge strips of the momentum sweep, the packing, the transfer (an asynchronous device copy) and the unpack
    run on the second stream while the interior rows of the same sweep run on the main one."""
    import torch
    from beom_amd import inputs as I, slab
    from beom_amd.grid import read_input_data
    p, files = I.case_headline(300, 259, 4)
    f = read_input_data(p, files=files)
    whole = capi.Engine(f)
    geoms = slab.decompose(p.mm, p.lm, 2)
    runs = []
    for g in geoms:
        e = capi.Engine(slab.slice_fields(f, g), slab_row0=g.row0, slab_mm=p.mm)
        r = slab.SlabRunner(e, g, p.nlay, dist=None, overlap=False)
        # switch the overlap machinery on by hand (no process group in this test)
        r.overlap = True
        r.main, r.comm = torch.cuda.Stream(), torch.cuda.Stream()
        r.main.wait_stream(torch.cuda.current_stream())
        e.set_stream(r.main.cuda_stream)
        runs.append(r)
    a, b = runs

    def xfer(dst, src_runner, src):
        def go():
            torch.cuda.current_stream().wait_event(src_runner._packed)
            dst.copy_(src, non_blocking=True)
        return go

    nsteps, cut = 40, 0
    for t in range(1, nsteps + 1):
        landed = [r._pending for r in runs]
        for r, other in ((a, landed[1]), (b, landed[0])):
            if other is not None:                       # the neighbour copies from my send buffers until ITS ghosts have landed
                r.comm.wait_event(other); r.main.wait_event(other)
            cut += int(r._advance(t))
        a._begin_transfer(xfer(a.recv_n, b, b.send_s))
        b._begin_transfer(xfer(b.recv_s, a, a.send_n))
    for r in runs:
        r.finish()
    assert cut == 2 * nsteps
    whole.step(1, nsteps)
    ref = whole.download()
    for r in runs:
        g = r.g
        ga, gb = 1 + (g.own0 - 1) * g.L, 1 + g.own1 * g.L
        la, lb = g.local_rows(g.own0, g.own1)
        st = r.engine.download()
        for k in ("hlay", "u", "v", "h_u", "h_v"):
            assert same_bits(st[k][:, la:lb], ref[k][:, ga:gb]), (g.rank, k)
        r.engine.close()
    whole.close()


@pytest.mark.parametrize("name", ["sill_4l_ocrp", "soliton_31x15_xper", "island_3l_forced"])
def test_device_side_output_records(name):
    """beom_download_outputs == the host-side write_array arithmetic (:2848-2883) and scans
    (:2772-2808) applied to the downloaded FP64 state."""
    g = Golden(name)
    f = _fields(g)
    e = capi.Engine(f, variant=g.variant)
    e.step(1, 7)
    st = e.download(("hlay", "u", "v"))
    h0r4 = np.ascontiguousarray(f.h_0[:, 1:].astype(np.float32))
    eta, u4, v4, mm, thin = e.download_outputs(h0r4)
    nl = g.p.nlay
    ref = np.zeros_like(eta)
    for k in range(nl - 1, -1, -1):
        d = st["hlay"][k, 1:] - h0r4[k].astype(np.float64)
        ref[k] = d.astype(np.float32) if k == nl - 1 else (d + ref[k + 1].astype(np.float64)).astype(np.float32)
    assert np.array_equal(eta.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(u4.view(np.uint32), st["u"][:, 1:].astype(np.float32).view(np.uint32))
    assert np.array_equal(v4.view(np.uint32), st["v"][:, 1:].astype(np.float32).view(np.uint32))
    wet = f.mk_n > 0.5
    for k in range(nl):
        assert mm[k, 0] == st["hlay"][k][wet].min() and mm[k, 1] == st["hlay"][k][wet].max()
        mu = f.mk_u > 0.5 if (f.mk_u > 0.5).any() else np.ones_like(wet)
        mv = f.mk_v > 0.5 if (f.mk_v > 0.5).any() else np.ones_like(wet)
        assert mm[k, 2] == st["u"][k][mu].min() and mm[k, 3] == st["u"][k][mu].max()
        assert mm[k, 4] == st["v"][k][mv].min() and mm[k, 5] == st["v"][k][mv].max()
    assert thin == 0
    # second call without h_0 (cached on the device) gives the same records
    eta2, _, _, _, _ = e.download_outputs(None)
    assert np.array_equal(eta2.view(np.uint32), eta.view(np.uint32))
    e.close()


@pytest.mark.parametrize("case,nband", [("closed_tall", 3), ("sill_tall", 2), ("stommel_tall", 2), ("soliton_xper", 2),
                                        ("beach_tall_noleith", 3), ("closed_tall_dt3d", 3),
                                        ("jet_xyper_tall", 2), ("jet_xyper_tall", 3), ("jet_yper_wind_tall", 2),
                                        ("jet_xyper_tall", 1), ("jet_xyper_tall_rccl", 1), ("closed_tall_rccl", 1),
                                        ("sponge_obc_tall", 2), ("sponge_obc_tall", 3),
                                        ("jet_yper_obc_tall", 1), ("jet_yper_obc_tall", 2), ("jet_yper_obc_tall", 3)])
def test_one_process_several_bands_match_single_handle(case, nband):
    """beom_multi_* (the single-process multi-GPU form the Fortran host uses): bands of rows, ghost
    exchange by peer copy on second streams, overlapped split steps — here with every band on
    the one GPU of the box.  The gathered state must equal the single handle's bit for bit,
    through an upload/download round trip in the middle.
    Frames periodic in y (private_mod.f95:642-668): the bands form a ring and the orphan row mm+1 is
    carried by the companion frame; with ONE band the ring closes on itself — over peer copies, and
    (*_rccl) over RCCL with a one-rank communicator, the only RCCL form a one-GPU box can run.
    sponge_obc_tall: nudged open boundaries with mcbc = 0 — the segments of no_gradient_obc (:2613-2679) are dealt to the bands
    (western and eastern boundaries cross every band; the southern and northern ones belong to the first and last).
    jet_yper_obc_tall: the same on a frame periodic in y — a ring of bands whose companion frame gets the segments' ends in
    rows 1..6, mm-3..mm and in the orphan row mm+1."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    rccl = case.endswith("_rccl")
    case = case[:-5] if rccl else case

    def jet_wind():
        p, files = I.case_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5)
        return p.replace(xper="0.", bdrg="2.e-4", tauw=["0.05", "0.02"]), files
    def jet_obc():
        p, files = I.case_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5)
        nudg = np.zeros((p.lm + 2, p.mm + 2, 3))
        for i in range(0, 9):                                   # western sponge (eta, u), dry margin included
            nudg[i, :, 0:2] = 0.3 * (9 - i) / 9.0
        for i in range(p.lm + 1, p.lm - 7, -1):                 # eastern sponge
            w = 0.25 * (i - (p.lm - 7)) / 9.0
            nudg[i, :, 0] = np.maximum(nudg[i, :, 0], w); nudg[i, :, 1] = np.maximum(nudg[i, :, 1], w)
        return p.replace(xper="0.", mcbc="0."), dict(files, nudg=nudg)
    p, files = {
        "jet_yper_obc_tall": jet_obc,
        "jet_xyper_tall": lambda: I.case_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5),
        "jet_yper_wind_tall": jet_wind,
        "closed_tall": lambda: I.case_headline(150, 260, 3),
        "sill_tall": lambda: I.case_sill_exchange3d(lm=133, mm=199, nlay=4, dt_s=0.01, npts=5, sill_halfwidth=20.0),
        "stommel_tall": lambda: I.case_stommel(lm=140, mm=150, dl=50.0e3, dt_s=0.2),
        "soliton_xper": lambda: I.case_soliton(lm=141, mm=63, dt_s=5.0),
        "sponge_obc_tall": lambda: (lambda pf: (pf[0].replace(mcbc="0."), pf[1]))(I.case_wave_sponge(lx=1400.0e3, ly=1100.0e3)),
        "beach_tall_noleith": lambda: I.case_carrier_beach(lm=140, mm=260, nlay=2, dt_s=0.08),
        # Leith viscosity refreshed every 3rd step only: the bands keep v_cc, v_ll standing in between
        "closed_tall_dt3d": lambda: (lambda pf: (pf[0].replace(dt3d="%.9f" % (3.2 * float(pf[0].dt) / 86400.0)), pf[1]))(
            I.case_headline(150, 260, 3)),
    }[case]()
    f = read_input_data(p, files=files)
    yper = float(p.yper) > 0.5
    one = capi.Engine(f)
    many = capi.MultiEngine(f, devices=[0] * nband, ring1=True,
                            transport=capi.XCHG_RCCL if rccl else capi.XCHG_PEER)
    assert many.count == nband
    d = many.describe()
    assert d["ring"] == int(yper) and d["bands_total"] == nband
    if rccl and (nband > 1 or yper):
        assert d["rccl_version"] > 0 and "RCCL" in d["transport"]
    b0 = many.band(0)
    assert b0["own0"] == 1
    if nband > 1:
        b1 = many.band(1)
        assert b1["own0"] == b0["own1"] + 1 and b1["win0"] == b1["own0"] - 4
    one.step(1, 9); many.step(1, 9)
    a, b = one.download(), many.download()
    lossy = _fuses(p)            # the fused sweeps do not keep v_cc, v_ll (single handle and bands alike)
    for k in (PROGNOSTIC if lossy else STATE):
        assert same(a[k], b[k]), (case, k, "after 9 steps")
    if yper:                     # the orphan row mm+1, the sign of zero included
        r0 = 1 + p.mm * (p.lm + 1)
        for k in ("hlay", "u", "v", "h_u", "h_v"):
            assert same_bits(a[k][:, r0:], b[k][:, r0:]), (case, k, "orphan row")
        for k in ("rs_h", "dmdx", "dmdy"):
            assert same_bits(a[k][:, r0:, :], b[k][:, r0:, :]), (case, k, "orphan row")
    many.upload(**b)                                   # round trip: scatter the gathered state again
    one.step(10, 14); many.step(10, 7); many.step(17, 7)
    a, b = one.download(), many.download()
    for k in PROGNOSTIC:
        assert same(a[k], b[k]), (case, k, "after 23 steps")
    st = many.stats()
    if nband > 1 or yper:
        assert st["split"] + st["plain"] == 23 * nband
    if case in ("closed_tall", "beach_tall_noleith", "jet_xyper_tall") and (nband > 1 or yper):  # every step after the 3rd of a call sequence is split
        assert st["split"] >= 15 * nband, st
    one.close(); many.close()


@pytest.mark.parametrize("case,nband", [("closed", 3), ("jet_ring", 2), ("sill_nudged", 2)])
def test_output_and_diag_records_of_bands_match_single_handle(case, nband):
    """write_array's records (eta_, u___, v___ and the diag trio pvor, mont, v_cc; private_mod.f95:2848-2974), min/max
    and the thin-layer scan formed on the devices of a frame cut into bands == those of the single handle, bit for bit;
    the single handle's diag records == a numpy restatement of the reference's expressions."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    p, files = {"closed": lambda: I.case_headline(150, 260, 3),
                "jet_ring": lambda: I.case_unstable_jet(lm=131, mm=151, nlay=2, dt_s=1.5),
                "sill_nudged": lambda: I.case_sill_exchange3d(lm=133, mm=199, nlay=4, dt_s=0.01, npts=5, sill_halfwidth=20.0)}[case]()
    f = read_input_data(p, files=files)
    one, many = capi.Engine(f), capi.MultiEngine(f, devices=[0] * nband)
    one.step(1, 9); many.step(1, 9)
    h0 = np.ascontiguousarray(f.h_0[:, 1:], dtype=np.float32)
    a, b = one.download_outputs(h0), many.download_outputs(h0)
    for x, y, nm in zip(a[:3], b[:3], ("eta", "u", "v")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (case, nm)
    assert np.array_equal(a[3], b[3]) and a[4] == b[4], (case, "minmax / thin layer")
    da, db = one.download_diag(), many.download_diag()
    for x, y, nm in zip(da, db, ("pvor", "mont", "v_cc")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (case, nm)
    # the reference's expressions in numpy (float64 state -> real*4 records)
    st = one.download(("hlay", "u", "v"))
    nb = f.neig.astype(np.int64)
    c1, c2, c3, c5, c6, c7 = (nb[:, k] for k in (0, 1, 2, 4, 5, 6))
    dl = float(p.dl)
    for k in range(p.nlay):
        u, v, h = st["u"][k], st["v"][k], st["hlay"][k]
        w1 = ((v - v[c5]) / dl - (u - u[c7]) / dl) * f.mkpe
        w2 = (u[c1] - u) / dl + (v[c3] - v) / dl
        q = ((w1[c1] - w1) ** 2 + (w1[c2] - w1[c3]) ** 2 + (w1[c3] - w1) ** 2 + (w1[c2] - w1[c1]) ** 2
             + (w2[c1] - w2) ** 2 + (w2 - w2[c5]) ** 2 + (w2[c3] - w2) ** 2 + (w2 - w2[c7]) ** 2)
        vcc = (float(p.bvis) + float(p.dvis) * (dl * dl) * np.sqrt(q)).astype(np.float32)
        assert np.array_equal(vcc[1:].view(np.uint32), da[2][k].view(np.uint32)), (case, "v_cc", k)
        with np.errstate(divide="ignore", invalid="ignore"):
            pv = ((f.fcor + w1 * float(p.uadv)) * f.mkpi * (f.mk_n + f.mk_n[c5] + f.mk_n[c7] + f.mk_n[c6])
                  / (h + h[c5] + h[c6] + h[c7])).astype(np.float32)
        assert np.array_equal(pv[1:].view(np.uint32), da[0][k].view(np.uint32)), (case, "pvor", k)
    one.close(); many.close()


@pytest.mark.parametrize("case,nband", [("island_leith", 2), ("island_leith", 3), ("bay_ocrp_nudged", 3), ("island_wind_drag", 2)])
def test_bands_of_frames_with_land_match_single_handle(case, nband):
    """Frames WITH land cut into bands: a band is the packed range of its rows (rows differ in length), dealt by packed-cell
    count, with the caller's own tables re-indexed to the window; every band runs the rectangle ("embedded") form and the
    ghost rows move as rows of that rectangle.  State after 9 and 23 steps (round trip in between), the output and diag
    records formed band by band: all equal to the single handle's bit for bit."""
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    if case == "island_leith":
        p, files = I.case_headline(200, 260, 3)
    elif case == "island_wind_drag":
        p, files = I.case_stommel(lm=200, mm=190, dl=50.0e3, dt_s=0.2)
        files = dict(files, h_bo=np.full((202, 192), 200.0))
    else:
        p, files = I.case_sill_exchange3d(lm=200, mm=230, nlay=3, dt_s=0.01, npts=5, sill_halfwidth=20.0)
    files = {k: np.array(v, dtype=np.float64) for k, v in files.items()}
    h = files["h_bo"]
    x = np.arange(p.lm + 2)[:, None]; y = np.arange(p.mm + 2)[None, :]
    land = ((x - 0.3 * p.lm) ** 2 + (y - 0.55 * p.mm) ** 2) < (0.22 * p.lm) ** 2          # an island across the band seams
    land |= (x > 0.8 * p.lm) & (y < 0.3 * p.mm) & ((x + y) % 7 != 0) if case == "bay_ocrp_nudged" else False   # a ragged corner
    h[land] = 0.0
    if "init" in files:
        files["init"][land] = 0.0
    p = p.replace(ndeg=I.get_nbr_deg_freedom(h))
    f = read_input_data(p, files=files)
    one, many = capi.Engine(f), capi.MultiEngine(f, devices=[0] * nband)
    assert one.is_embedded and many.count == nband
    bands = [many.band(k) for k in range(nband)]
    assert bands[0]["own0"] == 1 and bands[-1]["own1"] == p.mm + 1
    assert all(bands[k + 1]["own0"] == bands[k]["own1"] + 1 for k in range(nband - 1))
    j = f.subc[1, 1:]                                           # dealt by packed-cell count, not by rows
    cells = [int(((j >= b["own0"]) & (j <= b["own1"])).sum()) for b in bands]
    assert max(cells) - min(cells) <= 2 * (p.lm + 1), cells
    one.step(1, 9); many.step(1, 9)
    a, b = one.download(), many.download()
    for k in PROGNOSTIC:
        assert same(a[k], b[k]), (case, k, "after 9 steps")
    for k in ("hlay", "u", "v", "h_u", "h_v"):
        assert same_bits(a[k], b[k]), (case, k, "sign of zero")
    many.upload(**b)
    one.step(10, 14); many.step(10, 7); many.step(17, 7)
    a, b = one.download(), many.download()
    for k in PROGNOSTIC:
        assert same(a[k], b[k]), (case, k, "after 23 steps")
    st = many.stats()
    assert st["split"] + st["plain"] == 23 * nband
    if case != "island_wind_drag":                               # (a stress update every step: such steps are not split)
        assert st["split"] >= 15 * nband, st                     # the steps run in two phases around the exchange in flight
    h0 = np.ascontiguousarray(f.h_0[:, 1:], dtype=np.float32)
    ra, rb = one.download_outputs(h0), many.download_outputs(h0)
    for x_, y_, nm in zip(ra[:3], rb[:3], ("eta", "u", "v")):
        assert np.array_equal(x_.view(np.uint32), y_.view(np.uint32)), (case, nm)
    assert np.array_equal(ra[3], rb[3]) and ra[4] == rb[4]
    for x_, y_, nm in zip(one.download_diag(), many.download_diag(), ("pvor", "mont", "v_cc")):
        assert np.array_equal(x_.view(np.uint32), y_.view(np.uint32)), (case, nm)
    one.close(); many.close()


def test_multi_refuses_what_it_cannot_split():
    from beom_amd import inputs as I
    from beom_amd.grid import read_input_data
    p, files = I.case_unstable_jet(lm=40, mm=60, nlay=2, dt_s=1.5)       # periodic in y
    f = read_input_data(p, files=files)
    with pytest.raises(capi.BeomError):                                   # too few rows per band
        capi.MultiEngine(f, devices=[0] * 11)
    with pytest.raises(capi.BeomError):                                   # RCCL wants one device per band
        capi.MultiEngine(f, devices=[0, 0], transport=capi.XCHG_RCCL)
    m = capi.MultiEngine(f, devices=[0])                                  # one band = the frame itself
    m.step(1, 5)
    one = capi.Engine(f); one.step(1, 5)
    a, b = one.download(), m.download()
    for k in PROGNOSTIC:
        assert same(a[k], b[k]), k
    m.close(); one.close()
    files = {k: np.array(v, dtype=np.float64) for k, v in files.items()}  # land on a frame periodic in y: no ring of packed bands
    files["h_bo"][15:22, 25:33] = 0.0
    if "init" in files:
        files["init"][15:22, 25:33] = 0.0
    fl = read_input_data(p.replace(ndeg=I.get_nbr_deg_freedom(files["h_bo"])), files=files)
    with pytest.raises(capi.BeomError):
        capi.MultiEngine(fl, devices=[0, 0])
