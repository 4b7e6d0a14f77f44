// beom_kernels.h — the five sweeps of the BEOM time step as HIP kernels (gfx950).
//
// One thread = one packed cell of one layer (layer = blockIdx.y + 1 unless the sweep
// couples layers).  Every kernel is a template over the cell context of beom_dev.h
// (CellGather: any coastline, neighbours and masks from the caller's tables; CellDense:
// index arithmetic + mask predicates, XCD-aware tile order, no table traffic).
// Arithmetic follows the reference statement by statement and in its operation order
// (compiled with -ffp-contract=off) so results are bit-identical to the Fortran.
// HBM-bound FP64 stencils: no MFMA (nothing here is a contraction).
// (no include guard: beom_engine.hip includes this file once per tile geometry, each time inside a namespace,
//  after beom_dev.h; MV_Q and UV_WAVES select the geometry)

#define LL(a, ip, il) (a)[(long long)(ip) + d.n1 * (long long)((il) - 1)]
#define FNUD_(ip, il, iv) d.fnud[(long long)(ip) + d.n1 * ((long long)((il) - 1) + (long long)d.nlay * ((iv) - 1))]
#define NUDG_(ip, iv) d.nudg[(long long)(ip) + d.n1 * ((iv) - 1)]
#define TIDE_(m, ip, iv) d.tide[((m) - 1) + 2 * ((long long)(ip) + d.n1 * ((iv) - 1))]
#define T3_(a, ip, id, il) (a)[(long long)(ip) + d.n1 * ((long long)((id) - 1) + 2 * (long long)((il) - 1))]

// REAL**INTEGER with a constant exponent as flang emits it: the left-to-right product
// ((x*x)*x)*... (nsal is a parameter of shared_mod.f95; probed with flang 22 for n = 4..8)
__device__ __forceinline__ double powi_dev(double x, int n) {
    double r = x;
    for (int q = 1; q < n; ++q) r = r * x;
    return r;
}

// ---- first_three_timesteps prologue, private_mod.f95:2166-2177 ---------------------
template <class C>
__device__ __forceinline__ void body_rebuild_fluxes(const C &c, const DevView &d) {
    const int ipnt = c.ipnt, ilay = blockIdx.y + 1;
    const int c5 = c.template nb<5>(), c7 = c.template nb<7>();
    const double h0 = LL(d.hlay, ipnt, ilay);
    LL(d.h_u, ipnt, ilay) = LL(d.u, ipnt, ilay) * (h0 + LL(d.hlay, c5, ilay)) / (1.0 + c.mk_u());
    LL(d.h_v, ipnt, ilay) = LL(d.v, ipnt, ilay) * (h0 + LL(d.hlay, c7, ilay)) / (1.0 + c.mk_v());
}
template <class CTX>
__global__ __launch_bounds__(BEOM_BLOCK) void k_rebuild_fluxes(DevView d) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_rebuild_fluxes(c.as_interior(), d);
    else body_rebuild_fluxes(c, d);
}

// ---- update_h, private_mod.f95:1593-1646; variant 1 = private_mod3d.f95:1635-1683 ---
// Layers nlay..1 are walked inside the thread (variant 1 reads hlay(ipnt,3) across layers).
// FORCED=false compiles the nudging / tide / private_mod3d epilogue out (keeps cos() and its
// registers away from the common unforced launch); the host picks by d.has_nudg.
// FORCE: 0 = no nudging, 1 = nudging, 2 = nudging + tidal constituent (the cos path costs ~50 VGPR)
template <int FORCE, class C>
__device__ __forceinline__ void body_update_h(const C &c, const DevView &d, double gene, double ramp,
                                                         double ctim, int copy_hist) {
    constexpr bool FORCED = FORCE > 0;
    const int ipnt = c.ipnt;
    const int c1 = c.template nb<1>(), c3 = c.template nb<3>();
    const double i_dl = d.i_dl;
    const double mkn = c.mk_n();
    const double ng = FORCED ? nudg_rate<1>(c, d) : 0.0;
    const int ilay_hi = gridDim.y == 1 ? d.nlay : (int)blockIdx.y + 1;
    const int ilay_lo = gridDim.y == 1 ? 1 : (int)blockIdx.y + 1;
    for (int ilay = ilay_hi; ilay >= ilay_lo; --ilay) {
        double hold = LL(d.hlay, ipnt, ilay);
        // Sponges cover a few rows or columns of a frame: where the cell's relaxation rate is zero the target
        // thickness is not fetched at all (FORCE = 1, variant 0; see the epilogue) — a word per cell-layer saved
        const bool lazy_hfor = FORCE == 1 && d.variant == 0;
        const double hfor0 = (FORCED && !(lazy_hfor && ng == 0.0)) ? FNUD_(ipnt, ilay, 1) : 0.0;        // fetched with the rest, used at the end
        const double hu0 = LL(d.h_u, ipnt, ilay);
        double huE;
        if (C::kLanesAreRowNeighbours) {        // flux divergence in x: east value by wavefront shuffle
            huE = __shfl_down(hu0, 1, 64);
            if ((threadIdx.x & 63) == 63) huE = LL(d.h_u, c1, ilay);
        } else {
            huE = LL(d.h_u, c1, ilay);
        }
        double rs_3 = (hu0 - huE) * i_dl
                    + (LL(d.h_v, ipnt, ilay) - LL(d.h_v, c3, ilay)) * i_dl
                    + (d.has_hdot ? LL(d.hdot, ipnt, ilay) : 0.0);
        rs_3 = rs_3 * mkn;
        const double r2 = LL(d.rs[1], ipnt, ilay);
        double rhsi;
        if (gene != 0.0) {
            const double r1 = LL(d.rs[0], ipnt, ilay);
            rhsi = ((1.5 + d.beta) * rs_3 - (0.5 + 2.0 * d.beta) * r2 + d.beta * r1) * d.dt * gene
                 + rs_3 * d.dt * (1.0 - gene);
        } else {
            rhsi = rs_3 * d.dt;               // (..)*dt*0 + rs_3*dt*(1-0)
        }
        hold = hold + rhsi;
        // unforced: hfor*0 + (1-0)*hold with hfor = fnud_n >= +0  ==  (+0) + hold (turns -0 into +0)
        double hnew = 0.0 + hold;
        if (FORCED) {
            double hfor = hfor0;
            if (FORCE > 1) {
                const double vecl = (ilay == 1) ? 1.0 : 0.0;
                hfor = hfor + ramp * TIDE_(1, ipnt, 1) * vecl * cos(TIDE_(2, ipnt, 1) - d.w_ti * ctim);
            }
            if (lazy_hfor && ng == 0.0) {
                // hfor*0 + (1-0)*hold = (+-0) + hold: hold itself unless hold is an exact zero — only those lanes
                // (dry cells) fetch hfor for the sign of the result
                hnew = hold;
                if (hold == 0.0) hnew = FNUD_(ipnt, ilay, 1) * ng + (1.0 - ng) * hold;
            } else if (d.variant == 0) {
                hnew = hfor * ng + (1.0 - ng) * hold;
            } else {
                const double hfor1 = 0.0, hfor2 = 800.0, hfor3 = 0.0;
                const int isub = c.isub();
                const int half = d.lm / 2;
                const double h3v = (ilay == 3) ? hold : LL(d.hlay, ipnt, 3);
                double hl = hold;
                if (h3v > 20.0 * d.hsal && isub > half) {
                    if (ilay == 1)      hl = hl + 0.0 * ng + fmax(hfor1 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 2) hl = hl + 0.0 * ng + fmax(hfor2 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 3) hl = hl - 0.0 * ng + fmin(hfor3 * ng + (-ng) * hl, 0.0);
                } else if (h3v < 20.0 * d.hsal && isub > half) {
                    if (ilay == 1)      hl = hl + 0.0 * ng + 1.0 * fmax(hfor2 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 2) hl = hl - 0.0 * ng + 1.0 * fmin(hfor1 * ng + (-ng) * hl, 0.0);
                }
                if (isub < half) hl = hfor * ng + (1.0 - ng) * hold;
                hnew = hl;
            }
        }
        LL(d.hlay, ipnt, ilay) = hnew;
        if (copy_hist) {
            LL(d.rs[0], ipnt, ilay) = r2;
            LL(d.rs[1], ipnt, ilay) = rs_3;
        } else {
            LL(d.rs[0], ipnt, ilay) = rs_3;   // host swaps rs[0] <-> rs[1] afterwards
        }
    }
}
template <class CTX, int FORCE>
__global__ __launch_bounds__(BEOM_BLOCK) void k_update_h(DevView d, double gene, double ramp,
                                                         double ctim, int copy_hist) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_update_h<FORCE>(c.as_interior(), d, gene, ramp, ctim, copy_hist);
    else body_update_h<FORCE>(c, d, gene, ramp, ctim, copy_hist);
}

// ---- update_mont_rvor_pvor_dive_kine, private_mod.f95:2318-2439 ---------------------
template <class C>
__device__ __forceinline__ void body_update_mont(const C &c, const DevView &d, int ilay_only) {
    const int ipnt = c.ipnt;
    const int ilay = ilay_only ? ilay_only : (int)blockIdx.y + 1;
    const int c1 = c.template nb<1>(), c3 = c.template nb<3>(), c5 = c.template nb<5>(),
              c6 = c.template nb<6>(), c7 = c.template nb<7>();
    const double i_dl = d.i_dl, i_gr = d.i_gr, i_ns = d.i_ns;
    const double hs_8 = d.hsal;
    const double u_le = LL(d.u, ipnt, ilay), u_ri = LL(d.u, c1, ilay);
    const double v_bo = LL(d.v, ipnt, ilay), v_to = LL(d.v, c3, ilay);
    const double mkn = c.mk_n();
    const double h0 = LL(d.hlay, ipnt, ilay);
    double mpot = -0.0;          // ocrp = 0: (finite) * (-(0*i_ns*hsal*mk_n)) is -0 exactly
    if (d.ocrp != 0.0) {
        mpot = h0 + d.hmin * (1.0 - mkn);
        mpot = powi_dev(d.hsal / mpot, d.nsal - 1);
        mpot = mpot * (-d.ocrp * i_ns * d.hsal * mkn);
    }
    mpot = mpot - (d.has_hto ? d.h_to[ipnt] : 0.0);
    const double i_rn = d.i_rn[ilay - 1];
    for (int i = 1; i <= ilay - 1; ++i)
        mpot = mpot - (d.rhon[ilay - 1] - d.rhon[i - 1]) * i_rn * LL(d.hlay, ipnt, i);
    if (d.rgld < 0.5) {
        double hcol = 0.0;
        for (int i = 1; i <= d.nlay; ++i) hcol = hcol + LL(d.hlay, ipnt, i);
        mpot = hcol - d.h_th[ipnt] + mpot;
    }
    LL(d.mont, ipnt, ilay) = mpot + 0.25 * d.uadv * i_gr
                                  * (u_ri * u_ri + u_le * u_le + v_to * v_to + v_bo * v_bo);
    const double rv = (v_bo - LL(d.v, c5, ilay) - u_le + LL(d.u, c7, ilay)) * i_dl * c.mkpe();
    LL(d.rvor, ipnt, ilay) = rv;
    const double hE = LL(d.hlay, c1, ilay), hW = LL(d.hlay, c5, ilay),
                 hN = LL(d.hlay, c3, ilay), hS = LL(d.hlay, c7, ilay);
    const double mk1 = c.template mk_n_nb<1>(c1), mk3 = c.template mk_n_nb<3>(c3), mk5 = c.template mk_n_nb<5>(c5),
                 mk6 = c.template mk_n_nb<6>(c6), mk7 = c.template mk_n_nb<7>(c7);
    double d2x = (hE + hW - h0 * 2.0) * mk1 * mk5 * mkn;
    double d2y = (hN + hS - h0 * 2.0) * mk3 * mk7 * mkn;
    if (d.ocrp > 0.5) {
        if (hE < 2.0 * hs_8 || hW < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2x = 0.0;
        if (hN < 2.0 * hs_8 || hS < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2y = 0.0;
    }
    LL(d.d2hx, ipnt, ilay) = d2x;
    LL(d.d2hy, ipnt, ilay) = d2y;
    const double have = h0 + hW + LL(d.hlay, c6, ilay) + hS;
    LL(d.pvor, ipnt, ilay) = (d.fcor[ipnt] + rv * d.uadv) * c.mkpi() * (mkn + mk5 + mk6 + mk7) / have;
    LL(d.dive, ipnt, ilay) = (u_ri - u_le + v_to - v_bo) * i_dl;
}
template <class CTX>
__global__ __launch_bounds__(BEOM_BLOCK) void k_update_mont(DevView d, int ilay_only) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_update_mont(c.as_interior(), d, ilay_only);
    else body_update_mont(c, d, ilay_only);
}

// Same sweep, all NL layers of a cell in one thread: the column hlay(ipnt,1:nlay), the
// per-cell statics (fcor, h_th, masks) and the neighbour indices are fetched once instead
// of once per layer (the per-layer form moves ~2.4x the algorithmic bytes at nlay = 4).
// Operation order per layer is unchanged, so results are bit-identical to k_update_mont.
template <int NL, class C>
__device__ __forceinline__ void body_update_mont_all(const C &c, const DevView &d) {
    const int ipnt = c.ipnt;
    const int c1 = c.template nb<1>(), c3 = c.template nb<3>(), c5 = c.template nb<5>(),
              c6 = c.template nb<6>(), c7 = c.template nb<7>();
    const double i_dl = d.i_dl, i_gr = d.i_gr, i_ns = d.i_ns;
    const double hs_8 = d.hsal;
    const double mkn = c.mk_n(), mkpe = c.mkpe(), mkpi = c.mkpi();
    const double mk1 = c.template mk_n_nb<1>(c1), mk3 = c.template mk_n_nb<3>(c3), mk5 = c.template mk_n_nb<5>(c5),
                 mk6 = c.template mk_n_nb<6>(c6), mk7 = c.template mk_n_nb<7>(c7);
    const double fcor = d.fcor[ipnt], h_th = d.h_th[ipnt];
    const double h_to = d.has_hto ? d.h_to[ipnt] : 0.0;
    double h[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) h[l] = LL(d.hlay, ipnt, l + 1);
    double hcol = 0.0;
#pragma unroll
    for (int l = 0; l < NL; ++l) hcol = hcol + h[l];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int ilay = l + 1;
        const double u_le = LL(d.u, ipnt, ilay), u_ri = LL(d.u, c1, ilay);
        const double v_bo = LL(d.v, ipnt, ilay), v_to = LL(d.v, c3, ilay);
        const double h0 = h[l];
        double mpot = -0.0;
        if (d.ocrp != 0.0) {
            mpot = h0 + d.hmin * (1.0 - mkn);
            mpot = powi_dev(d.hsal / mpot, d.nsal - 1);
            mpot = mpot * (-d.ocrp * i_ns * d.hsal * mkn);
        }
        mpot = mpot - h_to;
        const double i_rn = d.i_rn[l];
#pragma unroll
        for (int i = 0; i < l; ++i) mpot = mpot - (d.rhon[l] - d.rhon[i]) * i_rn * h[i];
        if (d.rgld < 0.5) mpot = hcol - h_th + mpot;
        LL(d.mont, ipnt, ilay) = mpot + 0.25 * d.uadv * i_gr
                                      * (u_ri * u_ri + u_le * u_le + v_to * v_to + v_bo * v_bo);
        const double rv = (v_bo - LL(d.v, c5, ilay) - u_le + LL(d.u, c7, ilay)) * i_dl * mkpe;
        LL(d.rvor, ipnt, ilay) = rv;
        const double hE = LL(d.hlay, c1, ilay), hW = LL(d.hlay, c5, ilay),
                     hN = LL(d.hlay, c3, ilay), hS = LL(d.hlay, c7, ilay);
        double d2x = (hE + hW - h0 * 2.0) * mk1 * mk5 * mkn;
        double d2y = (hN + hS - h0 * 2.0) * mk3 * mk7 * mkn;
        if (d.ocrp > 0.5) {
            if (hE < 2.0 * hs_8 || hW < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2x = 0.0;
            if (hN < 2.0 * hs_8 || hS < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2y = 0.0;
        }
        LL(d.d2hx, ipnt, ilay) = d2x;
        LL(d.d2hy, ipnt, ilay) = d2y;
        const double have = h0 + hW + LL(d.hlay, c6, ilay) + hS;
        LL(d.pvor, ipnt, ilay) = (fcor + rv * d.uadv) * mkpi * (mkn + mk5 + mk6 + mk7) / have;
        LL(d.dive, ipnt, ilay) = (u_ri - u_le + v_to - v_bo) * i_dl;
    }
}
template <class CTX, int NL>
__global__ __launch_bounds__(BEOM_BLOCK) void k_update_mont_all(DevView d) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_update_mont_all<NL>(c.as_interior(), d);
    else body_update_mont_all<NL>(c, d);
}

// ---- fused sweep: update_mont_rvor_pvor_dive_kine (:2318-2439) + update_viscosity
//      (:2441-2502) for dense frames.  rvor and dive of a 64 x MV_TY tile plus a one-cell ring
//      are staged in LDS (double-buffered over layers, one barrier per layer); the Leith
//      stencil reads them from LDS, and the sweep hands update_u/update_v the products
//          pcd = v_cc*dive   and   qlr = v_ll*rvor
//      which is all they use of the four arrays (:1476-1479, :1561-1564); each product is
//      rounded before the differences are taken, exactly as in the reference.  Saves the
//      Leith launch, its 4 words, and 2 words in each momentum sweep.  With keep_diag the
//      four arrays are stored as well (parity tests; needed when viscosity is not refreshed
//      every step — then the unfused path runs instead).
#define MV_TX 64
#ifndef MV_Q
#define MV_Q 2                         // rows (cells) per thread
#endif
#define MV_TY (4 * MV_Q)
#ifndef MV_UNROLL
#define MV_UNROLL 4                    // measured: not unrolling the layer loop is 20 % slower
#endif
// tile of the fused u+v sweep (k_uv_fused, further down): k_mont_visc needs its geometry for lean_d2h
#define UV_TX 64
#ifndef UV_Q
#define UV_Q 2                         // rows (cells) per thread
#endif
#ifndef UV_WAVES
#define UV_WAVES 4                     // waves (rows of 64 cells) per workgroup
#endif
#define UV_BLOCK (64 * UV_WAVES)
#define UV_TY (UV_WAVES * UV_Q)
static constexpr int kUvBlock = UV_BLOCK;      // (for the launch code: the macros hold the LAST geometry included)
#define MV_LDX (MV_TX + 2 + 1)          // +1 pad column
#define MV_LDY (MV_TY + 2)

// u, v around a cell: loads and arithmetic kept apart so that a thread can issue the loads of all
// its cells (and of its ring cell) before the first use — one memory round trip per layer
template <bool INT>
__device__ __forceinline__ void uv6_load(const DevView &d, const CellDenseT<INT> &c, int ilay, double (&w)[6]) {
    const int ipnt = c.ipnt;
    const int c1 = c.template nb<1>(), c3 = c.template nb<3>(), c5 = c.template nb<5>(), c7 = c.template nb<7>();
    w[0] = LL(d.u, ipnt, ilay); w[1] = LL(d.u, c1, ilay);
    w[2] = LL(d.v, ipnt, ilay); w[3] = LL(d.v, c3, ilay);
    w[4] = LL(d.v, c5, ilay);   w[5] = LL(d.u, c7, ilay);
}
template <bool INT>
__device__ __forceinline__ void rv_dv_calc(const DevView &d, const CellDenseT<INT> &c, const double (&w)[6],
                                           double &rv, double &dv) {
    rv = (w[2] - w[4] - w[0] + w[5]) * d.i_dl * c.mkpe();        // (v_bo - v(W) - u_le + u(S))
    dv = (w[1] - w[0] + w[3] - w[2]) * d.i_dl;                   // (u_ri - u_le + v_to - v_bo)
}
// local target (a, b) as a NEIGHBOUR lookup resolves it: wraps applied, false = out of the frame (index 0)
template <bool INT>
__device__ __forceinline__ bool halo_target(const DevView &d, int &a, int &b) {
    if (!INT) {
        if (d.xper) { if (a == 0) a = d.L - 1; else if (a == d.L) a = 1; }
        if (d.yper && !d.slab) { if (b == 0) b = d.M - 1; else if (b == d.M) b = 1; }
        if (a < 1 || a > d.L || b < 1 || b > d.M) return false;
    }
    return true;
}

// LEITH = false: configurations whose viscosity is never refreshed after the first three steps
// (dvis <= 1e-3, svis = 0): v_cc, v_ll keep whatever update_viscosity left there (:2188) and only
// the products with this step's dive, rvor are formed — no ring of rvor/dive is needed.
template <int NL, bool INT, bool LEITH>
__device__ __forceinline__ void body_mont_visc(const DevView &d, int x0, int y0, bool wr_d2h, bool wr_prod,
                                               double (*s_rv)[MV_LDY][MV_LDX], double (*s_dv)[MV_LDY][MV_LDX],
                                               double (*s_hh)[MV_LDY][MV_LDX]) {
    const int tid = threadIdx.x;
    const int lx = tid & 63, wy = tid >> 6;              // column in tile, wave = row pair
    const int i = x0 + lx;
    const double i_gr = d.i_gr, i_ns = d.i_ns, hs_8 = d.hsal;
    // own cells: rows y0+wy (+4 per extra cell)
    CellDenseT<INT> c[MV_Q];
    bool ok[MV_Q], wr[MV_Q];
    double hcol[MV_Q], fcor[MV_Q], h_th[MV_Q], h_to[MV_Q], hown[NL][MV_Q];
    int n1[MV_Q], n3[MV_Q], n5[MV_Q], n6[MV_Q], n7[MV_Q];
    // The staged entry at a geometric position is what a NEIGHBOUR lookup of that target returns.
    // Under periodic wraps the orphan column i = L / row j = M are wrap targets
    // (private_mod.f95:619-620,647-648): they must hold the wrapped cell's values (widx).
    int widx[MV_Q];                                      // -1: the cell's own values are staged
#pragma unroll
    for (int q = 0; q < MV_Q; ++q) {
        const int j = y0 + wy + 4 * q;
        ok[q] = (i <= d.L) && (j <= d.M);
        if (!INT && ok[q] && !slot_is_cell(d, i + (j - 1) * d.P)) ok[q] = false;     // embedded: a land slot is the sentinel
        wr[q] = ok[q] && row_selected(d, j);           // cells outside the strips are staged, not stored
        c[q].set_cell(d, ok[q] ? i : 1, ok[q] ? j : 1);
        n1[q] = c[q].template nb<1>(); n3[q] = c[q].template nb<3>(); n5[q] = c[q].template nb<5>();
        n6[q] = c[q].template nb<6>(); n7[q] = c[q].template nb<7>();
        fcor[q] = d.fcor[c[q].ipnt]; h_th[q] = d.h_th[c[q].ipnt];
        h_to[q] = d.has_hto ? d.h_to[c[q].ipnt] : 0.0;
        hcol[q] = 0.0;
#pragma unroll
        for (int l = 0; l < NL; ++l) { hown[l][q] = LL(d.hlay, c[q].ipnt, l + 1); hcol[q] = hcol[q] + hown[l][q]; }
        widx[q] = -1;
        if (!INT) {
            if (!ok[q]) widx[q] = 0;                   // beyond the frame: lookups give the sentinel
            else if ((d.xper && i == d.L) || (d.yper && !d.slab && j == d.M)) {
                int a = i, b = j;
                widx[q] = (halo_target<INT>(d, a, b) && slot_is_cell(d, a + (b - 1) * d.P)) ? a + (b - 1) * d.P : 0;
            }
        }
    }
    // halo cell of this thread: ring of the (MV_TX+2) x (MV_TY+2) region
    int hr = -1, hc = -1;
    if (tid < MV_TX + 2) { hr = 0; hc = tid; }
    else if (tid < 2 * (MV_TX + 2)) { hr = MV_TY + 1; hc = tid - (MV_TX + 2); }
    else if (tid < 2 * (MV_TX + 2) + MV_TY) { hr = 1 + tid - 2 * (MV_TX + 2); hc = 0; }
    else if (tid < 2 * (MV_TX + 2) + 2 * MV_TY) { hr = 1 + tid - 2 * (MV_TX + 2) - MV_TY; hc = MV_TX + 1; }
    CellDenseT<INT> hcell;
    int hidx = 0;                                        // packed index of the ring target, 0 = sentinel
    hcell.set_cell(d, 1, 1);
    if (hr >= 0) {
        int a = x0 - 1 + hc, b = y0 - 1 + hr;
        if (halo_target<INT>(d, a, b) && (INT || slot_is_cell(d, a + (b - 1) * d.P))) { hcell.set_cell(d, a, b); hidx = hcell.ipnt; }
    }

#pragma unroll
    for (int l = 0; l < NL; ++l) {      // must stay unrolled: d.rhon[l] may not become a dynamic index
        const int ilay = l + 1, buf = l & 1;
        // every load of this layer first ...
        double w[MV_Q][6], wh[6], hring = 0.0, vcc0[MV_Q], vll0[MV_Q];
#pragma unroll
        for (int q = 0; q < MV_Q; ++q) {
            if (INT || ok[q]) uv6_load<INT>(d, c[q], ilay, w[q]);
            else { w[q][0] = w[q][1] = w[q][2] = w[q][3] = w[q][4] = w[q][5] = 0.0; }
            if (!LEITH) {     // zero_visc: v_cc = v_ll = +0 everywhere (dvis = bvis = 0), verified by the engine
                vcc0[q] = d.zero_visc ? 0.0 : LL(d.v_cc, c[q].ipnt, ilay);
                vll0[q] = d.zero_visc ? 0.0 : LL(d.v_ll, c[q].ipnt, ilay);
            }
        }
        if (hr >= 0) {
            hring = LL(d.hlay, hidx, ilay);
            if (LEITH && hidx != 0) uv6_load<INT>(d, hcell, ilay, wh);
        }
        // ... then the arithmetic and the stage
        double rv[MV_Q], dv[MV_Q];
#pragma unroll
        for (int q = 0; q < MV_Q; ++q) {
            rv[q] = 0.0; dv[q] = 0.0;
            if (INT || ok[q]) rv_dv_calc<INT>(d, c[q], w[q], rv[q], dv[q]);
            double srv = rv[q], sdv = dv[q], shh = hown[l][q];
            if (!INT && widx[q] >= 0) {                  // rare: orphan column/row, cells beyond the frame
                srv = 0.0; sdv = 0.0;                    // sentinel: rvor(0) = dive(0) = 0 (:273,275)
                shh = LL(d.hlay, widx[q], ilay);
                if (LEITH && widx[q] > 0) {
                    int a = i, b = y0 + wy + 4 * q;
                    halo_target<INT>(d, a, b);
                    CellDenseT<INT> t; t.set_cell(d, a, b);
                    double wt[6];
                    uv6_load<INT>(d, t, ilay, wt);
                    rv_dv_calc<INT>(d, t, wt, srv, sdv);
                }
            }
            if (LEITH) {
                s_rv[buf][1 + wy + 4 * q][1 + lx] = srv;
                s_dv[buf][1 + wy + 4 * q][1 + lx] = sdv;
            }
            s_hh[buf][1 + wy + 4 * q][1 + lx] = shh;
        }
        if (hr >= 0) {
            if (LEITH) {
                double a = 0.0, b = 0.0;
                if (hidx != 0) rv_dv_calc<INT>(d, hcell, wh, a, b);
                s_rv[buf][hr][hc] = a;
                s_dv[buf][hr][hc] = b;
            }
            s_hh[buf][hr][hc] = hring;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MV_Q; ++q) {
            if (!wr[q]) continue;
            const CellDenseT<INT> &cc = c[q];
            const int ipnt = cc.ipnt;
            const int r = 1 + wy + 4 * q, cx = 1 + lx;
            const double mkn = cc.mk_n(), mkpi = cc.mkpi();
            const double mk1 = cc.template mk_n_nb<1>(n1[q]), mk3 = cc.template mk_n_nb<3>(n3[q]),
                         mk5 = cc.template mk_n_nb<5>(n5[q]), mk6 = cc.template mk_n_nb<6>(n6[q]),
                         mk7 = cc.template mk_n_nb<7>(n7[q]);
            const double u_le = w[q][0], u_ri = w[q][1], v_bo = w[q][2], v_to = w[q][3];
            const double h0 = hown[l][q];
            double mpot = -0.0;
            if (d.ocrp != 0.0) {
                mpot = h0 + d.hmin * (1.0 - mkn);
                mpot = powi_dev(d.hsal / mpot, d.nsal - 1);
                mpot = mpot * (-d.ocrp * i_ns * d.hsal * mkn);
            }
            mpot = mpot - h_to[q];
            const double i_rn = d.i_rn[l];
#pragma unroll
            for (int m = 0; m < l; ++m) mpot = mpot - (d.rhon[l] - d.rhon[m]) * i_rn * hown[m][q];
            if (d.rgld < 0.5) mpot = hcol[q] - h_th[q] + mpot;
            LL(d.mont, ipnt, ilay) = mpot + 0.25 * d.uadv * i_gr
                                          * (u_ri * u_ri + u_le * u_le + v_to * v_to + v_bo * v_bo);
            // thickness of the neighbours from the stage (what the lookups n1, n5, n3, n7, n6 return)
            const double hE = s_hh[buf][r][cx + 1], hW = s_hh[buf][r][cx - 1], hN = s_hh[buf][r + 1][cx],
                         hS = s_hh[buf][r - 1][cx], h6 = s_hh[buf][r - 1][cx - 1];
            if (wr_d2h) {
                double d2x = (hE + hW - h0 * 2.0) * mk1 * mk5 * mkn;
                double d2y = (hN + hS - h0 * 2.0) * mk3 * mk7 * mkn;
                if (d.ocrp > 0.5) {
                    if (hE < 2.0 * hs_8 || hW < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2x = 0.0;
                    if (hN < 2.0 * hs_8 || hS < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2y = 0.0;
                }
                LL(d.d2hx, ipnt, ilay) = d2x;
                LL(d.d2hy, ipnt, ilay) = d2y;
            }
            const double have = h0 + hW + h6 + hS;
            LL(d.pvor, ipnt, ilay) = (fcor[q] + rv[q] * d.uadv) * mkpi * (mkn + mk5 + mk6 + mk7) / have;
            const double r_bl = rv[q], d_cc = dv[q];
            double vcc, vll;
            if (LEITH) {       // Leith viscosity from the staged ring (same names as :2458-2470)
                const double r_br = s_rv[buf][r][cx + 1], r_tr = s_rv[buf][r + 1][cx + 1],
                             r_tl = s_rv[buf][r + 1][cx], rbll = s_rv[buf][r][cx - 1], rbbl = s_rv[buf][r - 1][cx];
                const double d_ri = s_dv[buf][r][cx + 1], d_to = s_dv[buf][r + 1][cx],
                             d_le = s_dv[buf][r][cx - 1], d_bl = s_dv[buf][r - 1][cx - 1], d_bo = s_dv[buf][r - 1][cx];
                double a = (r_br - r_bl) * (r_br - r_bl) + (r_bl - rbll) * (r_bl - rbll)
                         + (r_tl - r_bl) * (r_tl - r_bl) + (r_bl - rbbl) * (r_bl - rbbl)
                         + (d_cc - d_le) * (d_cc - d_le) + (d_bo - d_bl) * (d_bo - d_bl)
                         + (d_cc - d_bo) * (d_cc - d_bo) + (d_le - d_bl) * (d_le - d_bl);
                vll = sqrt(a) * d.dvis * d.dl * d.dl + d.bvis;
                double b = (r_br - r_bl) * (r_br - r_bl) + (r_tr - r_tl) * (r_tr - r_tl)
                         + (r_tl - r_bl) * (r_tl - r_bl) + (r_tr - r_br) * (r_tr - r_br)
                         + (d_ri - d_cc) * (d_ri - d_cc) + (d_cc - d_le) * (d_cc - d_le)
                         + (d_to - d_cc) * (d_to - d_cc) + (d_cc - d_bo) * (d_cc - d_bo);
                vcc = sqrt(b) * d.dvis * d.dl * d.dl + d.bvis;
            } else {
                vcc = vcc0[q]; vll = vll0[q];
            }
            if (LEITH || wr_prod) {       // (+0)*dive, (+0)*rvor: only the boundary workgroups of k_uv_fused read them
                LL(d.pcd, ipnt, ilay) = vcc * d_cc;
                LL(d.qlr, ipnt, ilay) = vll * r_bl;
            }
            if (d.keep_diag) {
                LL(d.rvor, ipnt, ilay) = r_bl; LL(d.dive, ipnt, ilay) = d_cc;
                if (LEITH) { LL(d.v_cc, ipnt, ilay) = vcc; LL(d.v_ll, ipnt, ilay) = vll; }
            } else if (LEITH && d.keep_visc) {       // refreshed every n_3d > 1 steps: must stand until the next refresh
                LL(d.v_cc, ipnt, ilay) = vcc; LL(d.v_ll, ipnt, ilay) = vll;
            }
        }
    }
}

// Register budget of the fused Montgomery + Leith sweep: capped for THREE waves per SIMD (168 VGPRs; the 4-layer Leith form
// would take 172 and run two).  Measured on one box, alternating builds (tools/ab_variants.sh): mont+visc 1000 -> 905 us at
// 4096^2 x 4, 180 -> 157 us on the sill frame (Leith + outcropping), 822 -> 706 us at 8192 x 1024 x 8 despite 84-328 B of
// scratch per lane there; a cap for four waves (128 VGPRs) spills more than it hides.
#ifndef MV_WAVES_PER_EU
#define MV_WAVES_PER_EU 3
#endif
#if MV_WAVES_PER_EU > 0
#define MV_OCC_ATTR __attribute__((amdgpu_waves_per_eu(MV_WAVES_PER_EU)))
#else
#define MV_OCC_ATTR
#endif
template <int NL, bool LEITH = true>
__global__ __launch_bounds__(BEOM_BLOCK) MV_OCC_ATTR void k_mont_visc(DevView d) {
    __shared__ double s_rv[LEITH ? 2 : 1][LEITH ? MV_LDY : 1][LEITH ? MV_LDX : 1];
    __shared__ double s_dv[LEITH ? 2 : 1][LEITH ? MV_LDY : 1][LEITH ? MV_LDX : 1];
    __shared__ double s_hh[2][MV_LDY][MV_LDX];               // hlay of tile + ring
    const TileMap tm(d, MV_TX, MV_TY);
    int ty, ch;
    if (!tm.locate(blockIdx.x, ty, ch)) return;               // whole block: no barrier is skipped by part of it
    const int x0 = ch * MV_TX + 1, y0 = ty * MV_TY + 1;
    // block-uniform: tile and its ring lie in 2..L-2 x 2..M-2 (global rows too) -> no wraps, masks = 1
    const bool interior = x0 - 1 >= 2 && x0 + MV_TX <= d.L - 2 && y0 - 1 >= 2 && y0 + MV_TY <= d.M - 2
                          && y0 - 1 + d.joff >= 2 && y0 + MV_TY + d.joff <= d.Mg - 2 && tile_regular(d, x0, y0, MV_TY);
    // d.lean_d2h (the fused u+v sweep follows): its interior workgroups re-derive d2hx, d2hy from
    // hlay; only tiles that touch a non-interior tile of that sweep (same tiling) still store them
    const int uy0 = ((y0 - 1) / UV_TY) * UV_TY + 1;          // first row of the k_uv_fused tile this tile lies in
    bool deep = x0 - 1 - UV_TX >= 2 && x0 + 2 * UV_TX <= d.L - 2 && uy0 - 1 - UV_TY >= 2 && uy0 + 2 * UV_TY <= d.M - 2
                && uy0 - 1 - UV_TY + d.joff >= 2 && uy0 + 2 * UV_TY + d.joff <= d.Mg - 2;
    if (deep && d.embedded)           // all nine k_uv_fused tiles around this one take the staged interior path
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) deep = deep && tile_regular(d, x0 + dx * UV_TX, uy0 + dy * UV_TY, UV_TY);
    const bool wr_d2h = !(d.lean_d2h && deep);
    const bool wr_prod = !(d.zero_visc && deep);   // zero viscosity: interior workgroups of k_uv_fused skip the term
    if (interior) body_mont_visc<NL, true, LEITH>(d, x0, y0, wr_d2h, wr_prod, (double (*)[MV_LDY][MV_LDX])s_rv, (double (*)[MV_LDY][MV_LDX])s_dv, s_hh);
    else body_mont_visc<NL, false, LEITH>(d, x0, y0, wr_d2h, wr_prod, (double (*)[MV_LDY][MV_LDX])s_rv, (double (*)[MV_LDY][MV_LDX])s_dv, s_hh);
}
static inline dim3 mont_visc_grid(const DevView &d) { return dim3(TileMap(d, MV_TX, MV_TY).blocks(), 1, 1); }

// ---- update_viscosity (Leith part), private_mod.f95:2441-2502 -----------------------
template <class C>
__device__ __forceinline__ void body_update_visc(const C &c, const DevView &d, int ilay_only) {
    const int ipnt = c.ipnt;
    const int ilay = ilay_only ? ilay_only : (int)blockIdx.y + 1;
    const int c1 = c.template nb<1>(), c2 = c.template nb<2>(), c3 = c.template nb<3>(),
              c5 = c.template nb<5>(), c6 = c.template nb<6>(), c7 = c.template nb<7>();
    const double r_bl = LL(d.rvor, ipnt, ilay), r_br = LL(d.rvor, c1, ilay), r_tr = LL(d.rvor, c2, ilay),
                 r_tl = LL(d.rvor, c3, ilay), rbll = LL(d.rvor, c5, ilay), rbbl = LL(d.rvor, c7, ilay);
    const double d_cc = LL(d.dive, ipnt, ilay), d_ri = LL(d.dive, c1, ilay), d_to = LL(d.dive, c3, ilay),
                 d_le = LL(d.dive, c5, ilay), d_bl = LL(d.dive, c6, ilay), d_bo = LL(d.dive, c7, ilay);
    double a = (r_br - r_bl) * (r_br - r_bl) + (r_bl - rbll) * (r_bl - rbll)
             + (r_tl - r_bl) * (r_tl - r_bl) + (r_bl - rbbl) * (r_bl - rbbl)
             + (d_cc - d_le) * (d_cc - d_le) + (d_bo - d_bl) * (d_bo - d_bl)
             + (d_cc - d_bo) * (d_cc - d_bo) + (d_le - d_bl) * (d_le - d_bl);
    LL(d.v_ll, ipnt, ilay) = sqrt(a) * d.dvis * d.dl * d.dl + d.bvis;
    double b = (r_br - r_bl) * (r_br - r_bl) + (r_tr - r_tl) * (r_tr - r_tl)
             + (r_tl - r_bl) * (r_tl - r_bl) + (r_tr - r_br) * (r_tr - r_br)
             + (d_ri - d_cc) * (d_ri - d_cc) + (d_cc - d_le) * (d_cc - d_le)
             + (d_to - d_cc) * (d_to - d_cc) + (d_cc - d_bo) * (d_cc - d_bo);
    LL(d.v_cc, ipnt, ilay) = sqrt(b) * d.dvis * d.dl * d.dl + d.bvis;
}
template <class CTX>
__global__ __launch_bounds__(BEOM_BLOCK) void k_update_visc(DevView d, int ilay_only) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_update_visc(c.as_interior(), d, ilay_only);
    else body_update_visc(c, d, ilay_only);
}

// ---- update_u (XDIR=true, private_mod.f95:1422-1503) and update_v (XDIR=false,
//      :1505-1591).  The two routines are mirror images: W<->S, N<->E, NW<->SE. -------
// Where one momentum update reads and writes its prognostic data.  The unfused sweeps update
// in place (vel_out = vel_in, dm_out = oldest history level); the fused U+V sweep writes the
// FIRST component out of place because other workgroups re-evaluate it on their halo cells.
struct UVio {
    const double *vel_in; double *vel_out;      // u or v
    double *hp_out;                             // h_u or h_v (transport of THIS component)
    const double *dm0, *dm1, *dm2;              // history, oldest .. newest
    double *dm_out;                             // receives dmd4
};

// One cell of update_u (XDIR) / update_v.  q0,qb,qa,qd = transport of the OTHER component at
// self, cb, ca, cd (u: W,N,NW; v: S,E,SE) — from global memory, or from the LDS tile in the
// fused sweep.  STORE=false evaluates without any global store (halo cells) and returns the
// new transport of this component.
// Where a momentum update finds the five fields BOTH updates read (hlay, mont, pvor and the
// viscous products pcd, qlr) at the cell, at its "b" neighbour (u: W, v: S) and at its "a"
// neighbour (u: N, v: E): global memory, or the LDS image staged by the fused u+v sweep.
#define UV_LDX (UV_TX + 1 + 1)
#define UV_SROWS (UV_TY + 2)
#define UV_SLDX (UV_TX + 2)
typedef double UVstage[UV_SROWS][UV_SLDX];           // rows y0-1 .. y0+TY, cols x0-1 .. x0+TX
// hlay is staged one cell wider: the thickness curvatures d2hx, d2hy (:2393-2404) at a cell and
// at its W / S neighbour are re-evaluated from it, so interior workgroups never read (and the
// Montgomery sweep never writes) those two arrays
#define UV_HROWS (UV_TY + 4)
#define UV_HLDX (UV_TX + 4)
typedef double UVhstage[UV_HROWS][UV_HLDX];          // rows y0-2 .. y0+TY+1, cols x0-2 .. x0+TX+1

struct ShGlobal {
    const DevView &d; int ipnt, cb, ca, ilay;
    __device__ __forceinline__ double hlay_s() const { return LL(d.hlay, ipnt, ilay); }
    __device__ __forceinline__ double hlay_b() const { return LL(d.hlay, cb, ilay); }
    __device__ __forceinline__ double mont_s() const { return LL(d.mont, ipnt, ilay); }
    __device__ __forceinline__ double mont_b() const { return LL(d.mont, cb, ilay); }
    __device__ __forceinline__ double pvor_s() const { return LL(d.pvor, ipnt, ilay); }
    __device__ __forceinline__ double pvor_a() const { return LL(d.pvor, ca, ilay); }
    __device__ __forceinline__ double pcd_s() const { return LL(d.pcd, ipnt, ilay); }
    __device__ __forceinline__ double pcd_b() const { return LL(d.pcd, cb, ilay); }
    __device__ __forceinline__ double qlr_s() const { return LL(d.qlr, ipnt, ilay); }
    __device__ __forceinline__ double qlr_a() const { return LL(d.qlr, ca, ilay); }
    template <bool XDIR> __device__ __forceinline__ double d2h_s() const { return LL(XDIR ? d.d2hx : d.d2hy, ipnt, ilay); }
    template <bool XDIR> __device__ __forceinline__ double d2h_b() const { return LL(XDIR ? d.d2hx : d.d2hy, cb, ilay); }
};
template <bool XDIR>
struct ShLds {                                       // field order in the stage: 0 mont 1 pvor 2 pcd 3 qlr
    const UVstage *s; const double (*h)[UV_HLDX];    // h = the wider hlay stage
    int r, c;                                        // staged position of the cell (UVstage coordinates)
    double ocrp, hs2;                                // d.ocrp, 2*hsal (outcropping guard of d2h)
    static constexpr int RB = XDIR ? 0 : -1, CB = XDIR ? -1 : 0;   // b neighbour: W | S
    static constexpr int RA = XDIR ? 1 : 0, CA = XDIR ? 0 : 1;     // a neighbour: N | E
    __device__ __forceinline__ double hlay_s() const { return h[r + 1][c + 1]; }
    __device__ __forceinline__ double hlay_b() const { return h[r + 1 + RB][c + 1 + CB]; }
    __device__ __forceinline__ double mont_s() const { return s[0][r][c]; }
    __device__ __forceinline__ double mont_b() const { return s[0][r + RB][c + CB]; }
    __device__ __forceinline__ double pvor_s() const { return s[1][r][c]; }
    __device__ __forceinline__ double pvor_a() const { return s[1][r + RA][c + CA]; }
    __device__ __forceinline__ double pcd_s() const { return s[2][r][c]; }
    __device__ __forceinline__ double pcd_b() const { return s[2][r + RB][c + CB]; }
    __device__ __forceinline__ double qlr_s() const { return s[3][r][c]; }
    __device__ __forceinline__ double qlr_a() const { return s[3][r + RA][c + CA]; }
    // d2hx | d2hy of the wet interior (all three masks are 1 there), as body_mont_visc writes it
    __device__ __forceinline__ double d2h_at(int rr, int cc) const {
        const double h0 = h[rr][cc];
        const double hp = XDIR ? h[rr][cc + 1] : h[rr + 1][cc];    // E | N
        const double hm = XDIR ? h[rr][cc - 1] : h[rr - 1][cc];    // W | S
        double v = (hp + hm - h0 * 2.0) * 1.0 * 1.0 * 1.0;
        if (ocrp > 0.5) { if (hp < hs2 || hm < hs2 || h0 < hs2) v = 0.0; }
        return v;
    }
    template <bool X> __device__ __forceinline__ double d2h_s() const { return d2h_at(r + 1, c + 1); }
    template <bool X> __device__ __forceinline__ double d2h_b() const { return d2h_at(r + 1 + RB, c + 1 + CB); }
};

// distribute_stress folded into the momentum sweep (DevView::stress_fold: ocrp = 0, so the drag layer is the bottom resp.
// the top layer itself and its fraction is 1 at the cell and at its W / S neighbour): tb3d | tu3d (ipnt, direction, ilay)
// as :2015-2071 | :2075-2146 form it, from the velocities BEFORE this sweep (d.u, d.v: the fused sweep writes out of place).
template <bool XDIR, bool TOP, class C>
__device__ __forceinline__ double fold_drag(const C &c, const DevView &d, int ilay, double vel_self) {
    const int ipnt = c.ipnt;
    double other;                                  // the other component averaged to this point (:2019-2026, 2034-2041)
    if (XDIR) other = 0.25 * LL(d.v, ipnt, ilay) + 0.25 * LL(d.v, c.template nb<3>(), ilay) + 0.25 * LL(d.v, c.template nb<4>(), ilay)
                      + 0.25 * LL(d.v, c.template nb<5>(), ilay);
    else      other = 0.25 * LL(d.u, ipnt, ilay) + 0.25 * LL(d.u, c.template nb<1>(), ilay) + 0.25 * LL(d.u, c.template nb<7>(), ilay)
                      + 0.25 * LL(d.u, c.template nb<8>(), ilay);
    const double drg = TOP ? d.tdrg : d.bdrg;
    const double tau = vel_self * drg * (TOP ? d.rho_top : d.rho_bot) * (d.qdrg * sqrt(vel_self * vel_self + other * other) + 1.0 - d.qdrg);   // (rhon of layer 1 | nlay: no indexing into the kernel argument)
    return tau * 0.5 * (1.0 + 1.0);
}

// SF (stress fold): distribute_stress formed here (a template switch: the unforced sweeps must not carry its registers)
template <bool XDIR, bool PROD, bool STORE, bool SF, class C, class SH>
__device__ __forceinline__ double uv_core(const C &c, const DevView &d, int ilay, double gene, double ramp,
                                          double ctim, int copy_hist, const UVio &io,
                                          double q0, double qb, double qa, double qd, const SH &sh,
                                          bool do_store = true, const double *pre = nullptr, bool zv = false) {
    const int ipnt = c.ipnt;
    // u: cb = W(5), ca = N(3);   v: cb = S(7), ca = E(1)
    const int cb = XDIR ? c.template nb<5>() : c.template nb<7>();
    const int ca = XDIR ? c.template nb<3>() : c.template nb<1>();
    constexpr int IV = XDIR ? 2 : 3;               // ix_u / ix_v
    constexpr int ID = XDIR ? 1 : 2;               // stress component of this direction
    constexpr int IO = XDIR ? 2 : 1;               // the other one (Ekman term of ufor/vfor)
    const double i_dl = d.i_dl, i_r0 = d.i_r0, i_r1 = d.i_r1;
    const double mask = XDIR ? c.mk_u() : c.mk_v();
    const double h_self = sh.hlay_s(), m_self = sh.mont_s(), pv0 = sh.pvor_s();
    const double hcen = XDIR ? (sh.hlay_b() + h_self) / (1.0 + mask)
                             : (h_self + sh.hlay_b()) / (1.0 + mask);
    // pre (staged fused sweep): [0] this cell's velocity, [5..7] its history levels, loaded ahead of use
    double vold = pre ? pre[0] : LL(io.vel_in, ipnt, ilay);
    // nudging terms (wave-uniform switch): fetched up front, not next to their use further down
    // nudging rate of this point; the target velocity is fetched only where the rate is not zero (sponges cover a
    // few rows or columns of a frame) or where the sign of an exact zero is at stake
    double f_ng = 0.0;
    if (d.has_nudg) f_ng = NUDG_(ipnt, IV);       // (the tile table of update_h costs this sweep more registers than it saves bytes)
    const double dmd4 = (sh.mont_b() - m_self) * i_dl * d.grav * mask;
    const double pva = sh.pvor_a();
    double rhsi = dmd4 * (1.0 - gene);
    if (XDIR) rhsi = rhsi + 0.25 * pv0 * (q0 + qb) + 0.25 * pva * (qa + qd);
    else      rhsi = rhsi - 0.25 * pv0 * (q0 + qb) - 0.25 * pva * (qa + qd);
    if (SF) {
        // distribute_stress (:1921-2149) inside the sweep — ocrp = 0, a refresh on every step: the fractions are 1 in the top
        // layer (wind :1953, top drag :1999) resp. the bottom layer (:1977) and 0 elsewhere, so tt3d = taus there, tb3d / tu3d =
        // the drag at this point from the (old) velocities of that layer, by the reference's own operations.  In every other
        // layer all three terms are +-0 and the sum that follows (+ bodf, or + (+0)) hides their signs: nothing to do.
        const bool lay_t = ilay == 1, lay_b = ilay == d.nlay;
        if ((lay_t && (d.has_wind || d.has_top)) || (lay_b && d.has_bot)) {
            const double i__h = 1.0 / (hcen + 1.0 - mask);
            if (lay_t && d.has_wind) {
                const double tauw = 0.5 * (d.taus_cells[cb + d.n1 * (ID - 1)] + d.taus_cells[ipnt + d.n1 * (ID - 1)]) * ramp;
                rhsi = rhsi + tauw * i_r0 * i__h;
            } else rhsi = rhsi + 0.0;
            if (lay_b && d.has_bot) rhsi = rhsi - fold_drag<XDIR, false>(c, d, ilay, vold) * i_r0 * i__h;
            if (lay_t && d.has_top) rhsi = rhsi - fold_drag<XDIR, true>(c, d, ilay, vold) * i_r0 * i__h;
        }
    } else if (d.has_stress) {
        const double i__h = 1.0 / (hcen + 1.0 - mask);
        const double tauw = 0.5 * (T3_(d.tt3d, cb, ID, ilay) + T3_(d.tt3d, ipnt, ID, ilay)) * ramp;
        rhsi = rhsi + tauw * i_r0 * i__h;
        rhsi = rhsi - T3_(d.tb3d, ipnt, ID, ilay) * i_r0 * i__h;
        rhsi = rhsi - T3_(d.tu3d, ipnt, ID, ilay) * i_r0 * i__h;
    }
    rhsi = rhsi + (d.has_bodf ? d.bodf[(ilay - 1) + d.nlay * (ID - 1)] : 0.0);   // + (+0) is not a no-op for -0
    // gene = 0 (steps 1-3, g_fb = 0): the term is (finite)*0 = +-0 and only matters for the sign of
    // an exactly-zero rhsi; fetch the history on those (rare) lanes only.
    if (gene != 0.0 || rhsi == 0.0) {
        const bool p = pre && gene != 0.0;
        rhsi = rhsi + (d.del1 * dmd4 + d.del2 * (p ? pre[7] : LL(io.dm2, ipnt, ilay)) + d.gamm * (p ? pre[6] : LL(io.dm1, ipnt, ilay))
                       + d.epsi * (p ? pre[5] : LL(io.dm0, ipnt, ilay))) * gene;
    }
    if (!PROD && d.svis > 0.0) {   // (the engine never stages products when svis > 0)  biharmonic form (:1471-1473, :1555-1557); the u form rounds to default real
        const double i__h = 1.0 / (hcen + 1.0 - mask);
        if (XDIR) rhsi = rhsi - d.svis * i_dl * (double)(float)(LL(d.uu4, ipnt, ilay) - LL(d.uu4, cb, ilay)
                                                               + LL(d.vv4, ca, ilay) - LL(d.vv4, ipnt, ilay)) * i__h;
        else      rhsi = rhsi - d.svis * i_dl * (LL(d.vv4, ca, ilay) - LL(d.vv4, ipnt, ilay)
                                                - LL(d.uu4, ipnt, ilay) + LL(d.uu4, cb, ilay)) * i__h;
    } else if (PROD && zv) {
        // v_cc = v_ll = +0 everywhere: both differences are +-0, so rhsi +- them is rhsi itself unless
        // rhsi is an exact zero — and (+0) + (+-0) -+ (+-0) is +0 whatever the signs, so only rhsi = -0
        // can change (to +0).  Those lanes rebuild dive, rvor at the three cells from the (old) u, v with
        // the interior form of update_mont (:2388-2389, :2435-2436; masks 1) — all others skip the term.
        if (rhsi == 0.0 && __builtin_signbit(rhsi)) {
            const int oE = c.template nb<1>() - ipnt, oN = c.template nb<3>() - ipnt,
                      oW = c.template nb<5>() - ipnt, oS = c.template nb<7>() - ipnt;
            auto dive_at = [&](int x) {
                return (LL(d.u, x + oE, ilay) - LL(d.u, x, ilay) + LL(d.v, x + oN, ilay) - LL(d.v, x, ilay)) * d.i_dl;
            };
            auto rvor_at = [&](int x) {
                return (LL(d.v, x, ilay) - LL(d.v, x + oW, ilay) - LL(d.u, x, ilay) + LL(d.u, x + oS, ilay)) * d.i_dl * 1.0;
            };
            const double p0 = 0.0 * dive_at(ipnt), pb = 0.0 * dive_at(cb);
            const double l0 = 0.0 * rvor_at(ipnt), la = 0.0 * rvor_at(ca);
            if (XDIR) rhsi = rhsi + (p0 - pb) * i_dl - (la - l0) * i_dl;
            else      rhsi = rhsi + (p0 - pb) * i_dl + (la - l0) * i_dl;
        }
    } else if (PROD) {       // products staged by k_mont_visc: pcd = v_cc*dive, qlr = v_ll*rvor
        const double p0 = sh.pcd_s(), pb = sh.pcd_b();
        const double l0 = sh.qlr_s(), la = sh.qlr_a();
        if (XDIR) rhsi = rhsi + (p0 - pb) * i_dl - (la - l0) * i_dl;
        else      rhsi = rhsi + (p0 - pb) * i_dl + (la - l0) * i_dl;
    } else {
        const double vc0 = LL(d.v_cc, ipnt, ilay), vcb = LL(d.v_cc, cb, ilay);
        const double vl0 = LL(d.v_ll, ipnt, ilay), vla = LL(d.v_ll, ca, ilay);
        const double dv0 = LL(d.dive, ipnt, ilay), dvb = LL(d.dive, cb, ilay);
        const double rv0 = LL(d.rvor, ipnt, ilay), rva = LL(d.rvor, ca, ilay);
        if (XDIR) rhsi = rhsi + (vc0 * dv0 - vcb * dvb) * i_dl - (vla * rva - vl0 * rv0) * i_dl;
        else      rhsi = rhsi + (vc0 * dv0 - vcb * dvb) * i_dl + (vla * rva - vl0 * rv0) * i_dl;
    }
    vold = vold + rhsi * mask * d.dt;
    // Rate zero (or no nudging at all): the reference still evaluates vfor*0 + vold*(1-0) = (+-0) + vold, i.e. vold
    // unless vold is an exact zero, so only those (rare) lanes go through the full expression.
    if (f_ng != 0.0 || vold == 0.0) {
        const double i__hh = 1.0 / (hcen + 1.0 - mask);
        double vfor = FNUD_(ipnt, ilay, IV);
        if (SF) {                              // tt3d = taus * fraction (1 in the top layer, 0 below), formed here (see above)
            const double lt = ilay == 1 ? 1.0 : 0.0;
            const double t0 = d.has_wind ? d.taus_cells[ipnt + d.n1 * (IO - 1)] * lt : 0.0, tb_ = d.has_wind ? d.taus_cells[cb + d.n1 * (IO - 1)] * lt : 0.0;
            const double ek = 0.5 * (t0 + tb_) * i_r1 * d.invf * i__hh * ramp;
            vfor = XDIR ? vfor + ek : vfor - ek;
        } else if (d.has_stress) {
            const double ek = 0.5 * (T3_(d.tt3d, ipnt, IO, ilay) + T3_(d.tt3d, cb, IO, ilay))
                              * i_r1 * d.invf * i__hh * ramp;
            vfor = XDIR ? vfor + ek : vfor - ek;
        } else {
            vfor = XDIR ? vfor + 0.0 : vfor - 0.0;                      // the Ekman term is +0 then
        }
        if (d.has_tide) vfor = vfor + ramp * TIDE_(1, ipnt, IV) * cos(TIDE_(2, ipnt, IV) - d.w_ti * ctim);
        else vfor = vfor + 0.0;                                         // ramp*0*cos(0)
        vold = vfor * f_ng + vold * (1.0 - f_ng);
    }
    const double hnew = 0.5 * (vold + fabs(vold)) * (hcen - 0.16667 * sh.template d2h_b<XDIR>())
                      + 0.5 * (vold - fabs(vold)) * (hcen - 0.16667 * sh.template d2h_s<XDIR>());   // rgld = 0 (:1491,1577)
    if (STORE && do_store) {
        LL(io.vel_out, ipnt, ilay) = vold;
        if (d.rgld < 0.5) LL(io.hp_out, ipnt, ilay) = hnew;          // (:1491, :1577: with a lid the transports are rebuilt after the sweeps)
        if (copy_hist) {                       // single-layer entry points: shift like the reference
            const double m2 = LL(io.dm1, ipnt, ilay), m3 = LL(io.dm2, ipnt, ilay);
            LL(const_cast<double *>(io.dm0), ipnt, ilay) = m2;
            LL(const_cast<double *>(io.dm1), ipnt, ilay) = m3;
            LL(const_cast<double *>(io.dm2), ipnt, ilay) = dmd4;
        } else {
            LL(io.dm_out, ipnt, ilay) = dmd4;  // the host rotates the history pointers afterwards
        }
    }
    return hnew;
}

template <bool XDIR, bool PROD, class C>
__device__ __forceinline__ void body_update_uv(const C &c, const DevView &d, int ilay_only, double gene,
                                               double ramp, double ctim, int copy_hist) {
    const int ipnt = c.ipnt;
    const int ilay = ilay_only ? ilay_only : (int)blockIdx.y + 1;
    const int cb = XDIR ? c.template nb<5>() : c.template nb<7>();
    const int ca = XDIR ? c.template nb<3>() : c.template nb<1>();
    const int cd = XDIR ? c.template nb<4>() : c.template nb<8>();
    const double *hq = XDIR ? d.h_v : d.h_u;       // the transport of the OTHER component
    double *const *dm = XDIR ? d.dmx : d.dmy;
    const UVio io{XDIR ? d.u : d.v, XDIR ? d.u : d.v, XDIR ? d.h_u : d.h_v, dm[0], dm[1], dm[2], dm[0]};
    const ShGlobal sh{d, ipnt, cb, ca, ilay};
    uv_core<XDIR, PROD, true, false>(c, d, ilay, gene, ramp, ctim, copy_hist, io,
                              LL(hq, ipnt, ilay), LL(hq, cb, ilay), LL(hq, ca, ilay), LL(hq, cd, ilay), sh);
}
template <class CTX, bool XDIR, bool PROD = false>
__global__ __launch_bounds__(BEOM_BLOCK) void k_update_uv(DevView d, int ilay_only, double gene,
                                                          double ramp, double ctim, int copy_hist) {
    CTX c;
    if (!c.init(d)) return;
    if (c.wave_is_interior()) body_update_uv<XDIR, PROD>(c.as_interior(), d, ilay_only, gene, ramp, ctim, copy_hist);
    else body_update_uv<XDIR, PROD>(c, d, ilay_only, gene, ramp, ctim, copy_hist);
}

// ---- fused momentum sweep for dense frames: update_u and update_v (:1422-1591) of one
//      time step in ONE launch, in the order the step's parity asks for (:2276-2282).
//      The second update needs the first one's NEW transport at three neighbours
//      (v: h_u at E,S,SE :1531-1534; u: h_v at N,NW,W :1446-1449).  A workgroup evaluates the
//      first update on its 64 x 8 tile plus the one-cell row/column the second needs (73 extra
//      cells, 14 %), keeps the new transport in LDS, and runs the second update from it; its
//      own re-reads of hlay, mont, pvor, pcd, qlr hit L1/L2.  Because halo cells are evaluated
//      by a workgroup that does not own them, everything the FIRST update reads of a cell must
//      stay untouched during the launch: its velocity and newest history level go to spare
//      buffers (pointer swap afterwards), and the SECOND update's transport is written out of
//      place as well (the first update of other workgroups still reads the old one).
//      Algorithmic traffic: 22 words per cell-layer instead of 14 + 14.
// first update at one cell; SH = where its shared fields come from
template <bool FIRST_X, bool PROD, bool STORE, bool INT, bool SF, class SH>
__device__ __forceinline__ double uv_first_eval(const DevView &d, const CellDenseT<INT> &c, int ilay, double gene,
                                                double ramp, double ctim, const SH &sh, bool do_store = true,
                                                const double *pre = nullptr, bool zv = false) {
    const int ipnt = c.ipnt;
    const int cb = FIRST_X ? c.template nb<5>() : c.template nb<7>();
    const int ca = FIRST_X ? c.template nb<3>() : c.template nb<1>();
    const int cd = FIRST_X ? c.template nb<4>() : c.template nb<8>();
    const double *hq = FIRST_X ? d.h_v : d.h_u;
    double *const *dm = FIRST_X ? d.dmx : d.dmy;
    const UVio io{FIRST_X ? d.u : d.v, FIRST_X ? d.u_alt : d.v_alt, FIRST_X ? d.h_u : d.h_v,
                  dm[0], dm[1], dm[2], dm[3]};
    if (pre) return uv_core<FIRST_X, PROD, STORE, SF>(c, d, ilay, gene, ramp, ctim, 0, io, pre[1], pre[2], pre[3], pre[4],
                                                  sh, do_store, pre, zv);
    return uv_core<FIRST_X, PROD, STORE, SF>(c, d, ilay, gene, ramp, ctim, 0, io, LL(hq, ipnt, ilay),
                                         LL(hq, cb, ilay), LL(hq, ca, ilay), LL(hq, cd, ilay), sh, do_store);
}

// boundary workgroups: new first-component transport seen by a NEIGHBOUR lookup of the local
// target (a, b) — wraps / sentinel applied, everything from global memory
template <bool FIRST_X, bool PROD, bool SF>
__device__ __forceinline__ double uv_first_halo(const DevView &d, int a, int b, int ilay, double gene,
                                                double ramp, double ctim) {
    if (d.xper) { if (a == 0) a = d.L - 1; else if (a == d.L) a = 1; }
    if (d.yper && !d.slab) { if (b == 0) b = d.M - 1; else if (b == d.M) b = 1; }
    if (a < 1 || a > d.L || b < 1 || b > d.M) return 0.0;          // sentinel: h_u(0) = h_v(0) = 0
    if (!slot_is_cell(d, a + (b - 1) * d.P)) return 0.0;           // embedded: a land slot reads as the sentinel
    CellDenseT<false> h;
    h.set_cell(d, a, b);
    const int cb = FIRST_X ? h.template nb<5>() : h.template nb<7>();
    const int ca = FIRST_X ? h.template nb<3>() : h.template nb<1>();
    const ShGlobal sh{d, h.ipnt, cb, ca, ilay};
    return uv_first_eval<FIRST_X, PROD, false, false, SF>(d, h, ilay, gene, ramp, ctim, sh);
}

static_assert(MV_TX == UV_TX && UV_TY % MV_TY == 0, "lean_d2h: every tile of k_mont_visc lies inside one tile of k_uv_fused");
// Interior workgroups of the production pair (PROD, tile and ring strictly inside the wet interior):
// the five shared fields of tile + ring are staged in LDS once and both updates, ring cells
// included, read them there.  All loads of a phase are issued before the first use — the
// compiler keeps a load next to its use and would otherwise chain ~25 memory round trips
// per workgroup (stage loop, then every cell's velocity / transports / history one after the other).
template <bool FIRST_X>
__device__ __forceinline__ void uv_pre_load(const DevView &d, const CellDenseT<true> &c, int ilay, double gene,
                                            bool with_q, double (&pre)[8]) {
    const int ipnt = c.ipnt;
    double *const *dm = FIRST_X ? d.dmx : d.dmy;
    pre[0] = LL(FIRST_X ? d.u : d.v, ipnt, ilay);
    if (with_q) {       // old transport of the OTHER component at self, b, a, d (u: W,N,NW; v: S,E,SE)
        const double *hq = FIRST_X ? d.h_v : d.h_u;
        const int cb = FIRST_X ? c.template nb<5>() : c.template nb<7>();
        const int ca = FIRST_X ? c.template nb<3>() : c.template nb<1>();
        const int cd = FIRST_X ? c.template nb<4>() : c.template nb<8>();
        pre[1] = LL(hq, ipnt, ilay); pre[2] = LL(hq, cb, ilay); pre[3] = LL(hq, ca, ilay); pre[4] = LL(hq, cd, ilay);
    }
    if (gene != 0.0) { pre[5] = LL(dm[0], ipnt, ilay); pre[6] = LL(dm[1], ipnt, ilay); pre[7] = LL(dm[2], ipnt, ilay); }
}

template <bool FIRST_X, bool ZV, bool SF>
__device__ __forceinline__ void body_uv_fused_staged(const DevView &d, int x0, int y0, int ilay, double gene,
                                                     double ramp, double ctim, double (*s_h)[UV_LDX], UVstage *s_f,
                                                     double (*s_hl)[UV_HLDX]) {
    const int tid = threadIdx.x;
    const int lx = tid & 63, wy = tid >> 6;
    const int i = x0 + lx;
    constexpr int ROFF = FIRST_X ? 1 : 0, COFF = FIRST_X ? 0 : 1;      // s_h coordinates as in body_uv_fused
    constexpr int NST = UV_SROWS * (UV_TX + 2), NIT = (NST + UV_BLOCK - 1) / UV_BLOCK;
    const long long lay = d.n1 * (long long)(ilay - 1);
    // ---- phase A loads: stage elements, outer hlay ring, first update of own cells and of the ring cell
    // ZV (zero viscosity): pcd, qlr are neither staged nor read
    constexpr int NF = ZV ? 2 : 4;
    const double *src[5] = {d.mont, d.pvor, ZV ? d.hlay : d.pcd, ZV ? d.hlay : d.qlr, d.hlay};
    double fv[NIT][5];
    int frr[NIT], fcc[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * UV_BLOCK;
        const int idc = idx < NST ? idx : tid;           // clamped: the load is harmless, the store is skipped
        frr[k] = idc / (UV_TX + 2); fcc[k] = idc - frr[k] * (UV_TX + 2);
        const long long ip = (long long)(x0 - 1 + fcc[k]) + (long long)(y0 - 2 + frr[k]) * d.P + lay;
#pragma unroll
        for (int f = 0; f < 5; ++f) if (f >= NF && f < 4) fv[k][f] = 0.0; else fv[k][f] = src[f][ip];
    }
    // outer ring of the hlay stage without its corners: d2hy needs the rows y0-2 and y0+TY+1,
    // d2hx the columns x0-2 and x0+TX+1
    int hrr = -1, hcc = -1;
    if (tid < UV_TX + 2) { hrr = 0; hcc = 1 + tid; }
    else if (tid < 2 * (UV_TX + 2)) { hrr = UV_HROWS - 1; hcc = 1 + tid - (UV_TX + 2); }
    else if (tid < 2 * (UV_TX + 2) + UV_SROWS) { hrr = 1 + tid - 2 * (UV_TX + 2); hcc = 0; }
    else if (tid < 2 * (UV_TX + 2) + 2 * UV_SROWS) { hrr = 1 + tid - 2 * (UV_TX + 2) - UV_SROWS; hcc = UV_HLDX - 1; }
    const double hring = d.hlay[(long long)(x0 - 2 + (hrr >= 0 ? hcc : 2)) + (long long)(y0 - 3 + (hrr >= 0 ? hrr : 2)) * d.P + lay];
    CellDenseT<true> c[UV_Q];
    bool wr[UV_Q];
    double pre[UV_Q][8];
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        const int j = y0 + wy + UV_WAVES * q;
        wr[q] = row_selected(d, j);                  // cells outside the strips are evaluated, not stored
        c[q].set_cell(d, i, j);
        uv_pre_load<FIRST_X>(d, c[q], ilay, gene, true, pre[q]);
    }
    // ring cells of the first update: one row (65) + one column (UV_TY); other threads load their own cell again
    int rr1 = -1, cc1 = -1;
    if (tid <= UV_TX) { rr1 = FIRST_X ? 0 : UV_TY; cc1 = tid; }
    else if (tid <= UV_TX + UV_TY) { rr1 = (tid - UV_TX - 1) + ROFF; cc1 = FIRST_X ? UV_TX : 0; }
    const int ra = rr1 >= 0 ? (FIRST_X ? x0 : x0 - 1) + cc1 : i;
    const int rb = rr1 >= 0 ? (FIRST_X ? y0 - 1 : y0) + rr1 : y0 + wy;
    CellDenseT<true> hc;
    hc.set_cell(d, ra, rb);
    double preR[8];
    uv_pre_load<FIRST_X>(d, hc, ilay, gene, true, preR);
    // ---- stage
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        if (tid + k * UV_BLOCK < NST) {
#pragma unroll
            for (int f = 0; f < NF; ++f) s_f[f][frr[k]][fcc[k]] = fv[k][f];
            s_hl[frr[k] + 1][fcc[k] + 1] = fv[k][4];
        }
    }
    if (hrr >= 0) s_hl[hrr][hcc] = hring;
    __syncthreads();
    // ---- phase B loads (second update): in flight while the first update is evaluated
    double pre2[UV_Q][8];
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) uv_pre_load<!FIRST_X>(d, c[q], ilay, gene, false, pre2[q]);
    // ---- first update: own cells, then the ring cell
    const double hs2 = 2.0 * d.hsal;
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        const int r = wy + UV_WAVES * q;
        const ShLds<FIRST_X> sh{s_f, s_hl, r + 1, lx + 1, d.ocrp, hs2};
        s_h[r + ROFF][lx + COFF] = uv_first_eval<FIRST_X, true, true, true, SF>(d, c[q], ilay, gene, ramp, ctim, sh, wr[q], pre[q], ZV);
    }
    if (rr1 >= 0) {
        const ShLds<FIRST_X> sh{s_f, s_hl, rb - (y0 - 1), ra - (x0 - 1), d.ocrp, hs2};
        s_h[rr1][cc1] = uv_first_eval<FIRST_X, true, false, true, SF>(d, hc, ilay, gene, ramp, ctim, sh, true, preR, ZV);
    }
    __syncthreads();
    // ---- second component, transport of the first from LDS
    double *const *dm = FIRST_X ? d.dmy : d.dmx;
    const UVio io{FIRST_X ? d.v : d.u, FIRST_X ? d.v_alt : d.u_alt, FIRST_X ? d.hv_alt : d.hu_alt,
                  dm[0], dm[1], dm[2], dm[0]};
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        if (!wr[q]) continue;
        const int r = wy + UV_WAVES * q;
        double q0, qb, qa, qd;
        if (FIRST_X) {   // v: self, S, E, SE of h_u
            q0 = s_h[r + 1][lx]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r][lx + 1];
        } else {         // u: self, W, N, NW of h_v
            q0 = s_h[r][lx + 1]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r + 1][lx];
        }
        const ShLds<!FIRST_X> sh{s_f, s_hl, r + 1, lx + 1, d.ocrp, hs2};
        uv_core<!FIRST_X, true, true, SF>(c[q], d, ilay, gene, ramp, ctim, 0, io, q0, qb, qa, qd, sh, true, pre2[q], ZV);
    }
}

// Every other workgroup — boundary tiles (wraps, sentinel, masks), and all tiles of the v_cc/v_ll
// form (PROD = false) — reads global memory as the unfused sweeps do.
template <bool FIRST_X, bool PROD, bool INT, bool SF>
__device__ __forceinline__ void body_uv_fused(const DevView &d, int x0, int y0, int ilay, double gene,
                                              double ramp, double ctim, double (*s_h)[UV_LDX]) {
    const int tid = threadIdx.x;
    const int lx = tid & 63, wy = tid >> 6;
    const int i = x0 + lx;
    // s_h coordinates: FIRST_X  -> rows y0-1 .. y0+TY-1, cols x0 .. x0+TX   (own cell at [r+1][lx])
    //                  !FIRST_X -> rows y0 .. y0+TY,     cols x0-1 .. x0+TX-1 (own cell at [r][lx+1])
    constexpr int ROFF = FIRST_X ? 1 : 0, COFF = FIRST_X ? 0 : 1;
    CellDenseT<INT> c[UV_Q];
    bool ok[UV_Q], wr[UV_Q];
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        const int r = wy + UV_WAVES * q, j = y0 + r;
        ok[q] = (i <= d.L) && (j <= d.M);
        if (!INT && ok[q] && !slot_is_cell(d, i + (j - 1) * d.P)) ok[q] = false;     // embedded: land slots are never evaluated
        wr[q] = ok[q] && row_selected(d, j);       // cells outside the strips are evaluated, not stored
        c[q].set_cell(d, ok[q] ? i : 1, ok[q] ? j : 1);
        double hnew = 0.0;
        if (ok[q]) {
            const int cb = FIRST_X ? c[q].template nb<5>() : c[q].template nb<7>();
            const int ca = FIRST_X ? c[q].template nb<3>() : c[q].template nb<1>();
            const ShGlobal sh{d, c[q].ipnt, cb, ca, ilay};
            hnew = uv_first_eval<FIRST_X, PROD, true, INT, SF>(d, c[q], ilay, gene, ramp, ctim, sh, wr[q]);
            if (!INT) {     // orphan column/row are wrap TARGETS: stage what a neighbour lookup returns
                if ((d.xper && i == d.L) || (d.yper && !d.slab && j == d.M))
                    hnew = uv_first_halo<FIRST_X, PROD, SF>(d, i, j, ilay, gene, ramp, ctim);
            }
        }
        s_h[r + ROFF][lx + COFF] = hnew;
    }
    // ring cells: one row (65) + one column (UV_TY)
    {
        int rr = -1, cc = -1;
        if (tid <= UV_TX) {                       // the extra row
            rr = FIRST_X ? 0 : UV_TY; cc = tid;
        } else if (tid <= UV_TX + UV_TY) {        // the extra column
            rr = (tid - UV_TX - 1) + ROFF; cc = FIRST_X ? UV_TX : 0;
        }
        if (rr >= 0) {
            const int a = (FIRST_X ? x0 : x0 - 1) + cc;
            const int b = (FIRST_X ? y0 - 1 : y0) + rr;
            s_h[rr][cc] = uv_first_halo<FIRST_X, PROD, SF>(d, a, b, ilay, gene, ramp, ctim);
        }
    }
    __syncthreads();
    // second component, transport of the first from LDS
    double *const *dm = FIRST_X ? d.dmy : d.dmx;
    // the second component's velocity also goes to its partner buffer: an edge pass launched
    // after this one must still find the OLD u and v of every row (rvor/dive of its ring)
    const UVio io{FIRST_X ? d.v : d.u, FIRST_X ? d.v_alt : d.u_alt, FIRST_X ? d.hv_alt : d.hu_alt,
                  dm[0], dm[1], dm[2], dm[0]};
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        if (!wr[q]) continue;
        const int r = wy + UV_WAVES * q;
        double q0, qb, qa, qd;
        if (FIRST_X) {   // v: self, S, E, SE of h_u
            q0 = s_h[r + 1][lx]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r][lx + 1];
        } else {         // u: self, W, N, NW of h_v
            q0 = s_h[r][lx + 1]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r + 1][lx];
        }
        const int cb = !FIRST_X ? c[q].template nb<5>() : c[q].template nb<7>();
        const int ca = !FIRST_X ? c[q].template nb<3>() : c[q].template nb<1>();
        const ShGlobal sh{d, c[q].ipnt, cb, ca, ilay};
        uv_core<!FIRST_X, PROD, true, SF>(c[q], d, ilay, gene, ramp, ctim, 0, io, q0, qb, qa, qd, sh);
    }
}

// Edge workgroups of the production pair (wraps, sentinel, masks): the five shared fields of tile + ring are
// staged in LDS BY LOOKUP — the entry at a geometric position is what a NEIGHBOUR lookup of that position
// returns (periodic wraps applied, index 0 outside the frame) — so both updates, ring cells included, read
// them there like the interior workgroups do; masks stay the predicates of CellDenseT<false>, and the thickness
// curvatures come from the arrays k_mont_visc stores for these tiles (they carry the coast masks, :2393-2404).
// (Frames a few tiles tall or wide — the soliton channel, a band of a multi-GPU run — are mostly edge tiles.)
template <bool XDIR>
struct ShLdsEdge {                                   // field order in the stage: 0 mont 1 pvor 2 pcd 3 qlr
    const UVstage *s; const double (*h)[UV_SLDX];    // h = the hlay image
    int r, c;                                        // staged position of the cell (UVstage coordinates)
    const DevView &d; int ipnt, cb, ilay;            // for the curvature arrays
    static constexpr int RB = XDIR ? 0 : -1, CB = XDIR ? -1 : 0;   // b neighbour: W | S
    static constexpr int RA = XDIR ? 1 : 0, CA = XDIR ? 0 : 1;     // a neighbour: N | E
    __device__ __forceinline__ double hlay_s() const { return h[r][c]; }
    __device__ __forceinline__ double hlay_b() const { return h[r + RB][c + CB]; }
    __device__ __forceinline__ double mont_s() const { return s[0][r][c]; }
    __device__ __forceinline__ double mont_b() const { return s[0][r + RB][c + CB]; }
    __device__ __forceinline__ double pvor_s() const { return s[1][r][c]; }
    __device__ __forceinline__ double pvor_a() const { return s[1][r + RA][c + CA]; }
    __device__ __forceinline__ double pcd_s() const { return s[2][r][c]; }
    __device__ __forceinline__ double pcd_b() const { return s[2][r + RB][c + CB]; }
    __device__ __forceinline__ double qlr_s() const { return s[3][r][c]; }
    __device__ __forceinline__ double qlr_a() const { return s[3][r + RA][c + CA]; }
    template <bool X> __device__ __forceinline__ double d2h_s() const { return LL(X ? d.d2hx : d.d2hy, ipnt, ilay); }
    template <bool X> __device__ __forceinline__ double d2h_b() const { return LL(X ? d.d2hx : d.d2hy, cb, ilay); }
};

template <bool FIRST_X, bool SF>
__device__ __forceinline__ void body_uv_fused_edge(const DevView &d, int x0, int y0, int ilay, double gene, double ramp,
                                                   double ctim, double (*s_h)[UV_LDX], UVstage *s_f, double (*s_hl)[UV_HLDX]) {
    static_assert(sizeof(double) * UV_HROWS * UV_HLDX >= sizeof(UVstage), "the hlay image fits the widened hlay stage");
    double (*s_hs)[UV_SLDX] = (double (*)[UV_SLDX])s_hl;
    const int tid = threadIdx.x;
    const int lx = tid & 63, wy = tid >> 6;
    const int i = x0 + lx;
    constexpr int ROFF = FIRST_X ? 1 : 0, COFF = FIRST_X ? 0 : 1;      // s_h coordinates as in body_uv_fused
    constexpr int NST = UV_SROWS * UV_SLDX, NIT = (NST + UV_BLOCK - 1) / UV_BLOCK;
    // ---- stage by lookup
    {
        const double *src[5] = {d.mont, d.pvor, d.pcd, d.qlr, d.hlay};
        double fv[NIT][5];
        int frr[NIT], fcc[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + k * UV_BLOCK;
            const int idc = idx < NST ? idx : tid;
            frr[k] = idc / UV_SLDX; fcc[k] = idc - frr[k] * UV_SLDX;
            int a = x0 - 1 + fcc[k], b = y0 - 1 + frr[k];
            const int t = halo_target<false>(d, a, b) ? a + (b - 1) * d.P : 0;     // 0: the sentinel (zero in every one of these arrays)
#pragma unroll
            for (int f = 0; f < 5; ++f) fv[k][f] = LL(src[f], t, ilay);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (tid + k * UV_BLOCK < NST) {
#pragma unroll
                for (int f = 0; f < 4; ++f) s_f[f][frr[k]][fcc[k]] = fv[k][f];
                s_hs[frr[k]][fcc[k]] = fv[k][4];
            }
        }
    }
    CellDenseT<false> c[UV_Q];
    bool ok[UV_Q], wr[UV_Q];
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        const int j = y0 + wy + UV_WAVES * q;
        ok[q] = (i <= d.L) && (j <= d.M);
        if (ok[q] && !slot_is_cell(d, i + (j - 1) * d.P)) ok[q] = false;     // embedded: land slots are never evaluated
        wr[q] = ok[q] && row_selected(d, j);       // cells outside the strips are evaluated, not stored
        c[q].set_cell(d, ok[q] ? i : 1, ok[q] ? j : 1);
    }
    __syncthreads();
    // ---- first update: own cells, then the ring cell
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        const int r = wy + UV_WAVES * q, j = y0 + r;
        double hnew = 0.0;
        if (ok[q]) {
            const int cb = FIRST_X ? c[q].template nb<5>() : c[q].template nb<7>();
            if ((d.xper && i == d.L) || (d.yper && !d.slab && j == d.M)) {
                // orphan column / row: the image at its position holds the WRAPPED cell's values (it is a wrap target),
                // so the cell's own update reads global memory; and what it stages is what a lookup of it returns
                const int ca = FIRST_X ? c[q].template nb<3>() : c[q].template nb<1>();
                const ShGlobal sh{d, c[q].ipnt, cb, ca, ilay};
                (void)uv_first_eval<FIRST_X, true, true, false, SF>(d, c[q], ilay, gene, ramp, ctim, sh, wr[q]);
                hnew = uv_first_halo<FIRST_X, true, SF>(d, i, j, ilay, gene, ramp, ctim);
            } else {
                const ShLdsEdge<FIRST_X> sh{s_f, s_hs, r + 1, lx + 1, d, c[q].ipnt, cb, ilay};
                hnew = uv_first_eval<FIRST_X, true, true, false, SF>(d, c[q], ilay, gene, ramp, ctim, sh, wr[q]);
            }
        }
        s_h[r + ROFF][lx + COFF] = hnew;
    }
    {
        int rr = -1, cc = -1;
        if (tid <= UV_TX) { rr = FIRST_X ? 0 : UV_TY; cc = tid; }                              // the extra row
        else if (tid <= UV_TX + UV_TY) { rr = (tid - UV_TX - 1) + ROFF; cc = FIRST_X ? UV_TX : 0; }   // the extra column
        if (rr >= 0) {
            const int ga = (FIRST_X ? x0 : x0 - 1) + cc, gb = (FIRST_X ? y0 - 1 : y0) + rr;    // geometric position
            int a = ga, b = gb;
            double val = 0.0;                                     // outside the frame: h_u(0) = h_v(0) = 0
            if (halo_target<false>(d, a, b) && slot_is_cell(d, a + (b - 1) * d.P)) {
                if (a != ga || b != gb) {
                    // a ring position beyond the periodic seam: the image around it is not the neighbourhood of the
                    // cell it stands for (the wraps act on every lookup anew) — the rare global path
                    val = uv_first_halo<FIRST_X, true, SF>(d, ga, gb, ilay, gene, ramp, ctim);
                } else {
                    CellDenseT<false> h;
                    h.set_cell(d, a, b);
                    const int cb = FIRST_X ? h.template nb<5>() : h.template nb<7>();
                    const ShLdsEdge<FIRST_X> sh{s_f, s_hs, gb - (y0 - 1), ga - (x0 - 1), d, h.ipnt, cb, ilay};
                    val = uv_first_eval<FIRST_X, true, false, false, SF>(d, h, ilay, gene, ramp, ctim, sh);
                }
            }
            s_h[rr][cc] = val;
        }
    }
    __syncthreads();
    // ---- second component, transport of the first from LDS
    double *const *dm = FIRST_X ? d.dmy : d.dmx;
    const UVio io{FIRST_X ? d.v : d.u, FIRST_X ? d.v_alt : d.u_alt, FIRST_X ? d.hv_alt : d.hu_alt,
                  dm[0], dm[1], dm[2], dm[0]};
#pragma unroll
    for (int q = 0; q < UV_Q; ++q) {
        if (!wr[q]) continue;
        const int r = wy + UV_WAVES * q;
        double q0, qb, qa, qd;
        if (FIRST_X) { q0 = s_h[r + 1][lx]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r][lx + 1]; }
        else         { q0 = s_h[r][lx + 1]; qb = s_h[r][lx]; qa = s_h[r + 1][lx + 1]; qd = s_h[r + 1][lx]; }
        const int cb = !FIRST_X ? c[q].template nb<5>() : c[q].template nb<7>();
        if ((d.xper && i == d.L) || (d.yper && !d.slab && y0 + r == d.M)) {      // orphan column / row: see above
            const int ca = !FIRST_X ? c[q].template nb<3>() : c[q].template nb<1>();
            const ShGlobal sh{d, c[q].ipnt, cb, ca, ilay};
            uv_core<!FIRST_X, true, true, SF>(c[q], d, ilay, gene, ramp, ctim, 0, io, q0, qb, qa, qd, sh);
        } else {
            const ShLdsEdge<!FIRST_X> sh{s_f, s_hs, r + 1, lx + 1, d, c[q].ipnt, cb, ilay};
            uv_core<!FIRST_X, true, true, SF>(c[q], d, ilay, gene, ramp, ctim, 0, io, q0, qb, qa, qd, sh);
        }
    }
}

// ZV (with PROD): v_cc = v_ll = +0 everywhere — interior workgroups drop the viscous products
#if defined(UV_WAVES_PER_EU) && UV_WAVES_PER_EU > 0
#define UV_OCC_ATTR __attribute__((amdgpu_waves_per_eu(UV_WAVES_PER_EU)))
#else
#define UV_OCC_ATTR
#endif
template <bool FIRST_X, bool PROD, bool ZV, bool SF>
__device__ __forceinline__ void uv_fused_workgroup(const DevView &d, double gene, double ramp, double ctim) {
    __shared__ double s_h[UV_TY + 1][UV_LDX];
    __shared__ UVstage s_f[PROD ? 4 : 1];                    // (ZV: the interior workgroups use two of them, the edge ones all four)
    __shared__ double s_hl[PROD ? UV_HROWS : 1][UV_HLDX];
    const TileMap tm(d, UV_TX, UV_TY);
    int ty, ch;
    if (!tm.locate(blockIdx.x, ty, ch)) return;
    const int x0 = ch * UV_TX + 1, y0 = ty * UV_TY + 1;
    const int ilay = blockIdx.y + 1;
    const bool interior = x0 - 1 >= 2 && x0 + UV_TX <= d.L - 2 && y0 - 1 >= 2 && y0 + UV_TY <= d.M - 2
                          && y0 - 1 + d.joff >= 2 && y0 + UV_TY + d.joff <= d.Mg - 2 && tile_regular(d, x0, y0, UV_TY);
    if (interior && PROD) body_uv_fused_staged<FIRST_X, ZV, SF>(d, x0, y0, ilay, gene, ramp, ctim, s_h, s_f, s_hl);
    else if (interior) body_uv_fused<FIRST_X, PROD, true, SF>(d, x0, y0, ilay, gene, ramp, ctim, s_h);
    else if (PROD && !d.edge_global) body_uv_fused_edge<FIRST_X, SF>(d, x0, y0, ilay, gene, ramp, ctim, s_h, s_f, s_hl);
    else body_uv_fused<FIRST_X, PROD, false, SF>(d, x0, y0, ilay, gene, ramp, ctim, s_h);
}
template <bool FIRST_X, bool PROD, bool ZV = false>
__global__ __launch_bounds__(UV_BLOCK) UV_OCC_ATTR void k_uv_fused(DevView d, double gene, double ramp, double ctim) {
    uv_fused_workgroup<FIRST_X, PROD, ZV, false>(d, gene, ramp, ctim);
}
// ... with distribute_stress formed inside (SF): its own kernels, so that the unforced ones do not carry its registers.  Their
// zero-viscosity form takes 129 VGPRs; capped for four waves per SIMD it is slower than left at three (wind-driven
// 4096x2048x2, same box: u+v 730 vs 714 us)
template <bool FIRST_X, bool PROD, bool ZV = false>
#ifndef UV_SF_WAVES_PER_EU
#define UV_SF_WAVES_PER_EU 3
#endif
__global__ __launch_bounds__(UV_BLOCK) __attribute__((amdgpu_waves_per_eu(UV_SF_WAVES_PER_EU))) void k_uv_fused_sf(DevView d, double gene, double ramp, double ctim) {
    uv_fused_workgroup<FIRST_X, PROD, ZV, true>(d, gene, ramp, ctim);
}
static inline dim3 uv_fused_grid(const DevView &d) {
    return dim3(TileMap(d, UV_TX, UV_TY).blocks(), (unsigned)d.nlay, 1);
}

// ---- distribute_stress, private_mod.f95:1921-2149, as ONE launch --------------------------------------------------
// The reference forms (a) the layer fractions layt / layb / layu of every cell (:1945-2009), (b) the bottom and top stress at
// the u / v points (:2015-2049, 2075-2109) and (c) their distribution over the layers, which reads the fractions of the W and
// S neighbours (:2056-2071, 2116-2146).  The fractions are a function of one cell's thickness column, so a thread forms them
// for its own cell and for its two neighbours itself (the sentinel and land slots hold hlay = 0 and give the sentinel's
// fractions) and nothing has to wait for a neighbour's thread: one kernel instead of three or four, no work arrays.
// Every value is computed by the reference's own sequence of operations.
template <bool FROM_TOP>
__device__ __forceinline__ void stress_fractions(const DevView &d, int q, double (*lay)[BEOM_BLOCK]) {
    const int nlay = d.nlay, t = threadIdx.x;
    if (FROM_TOP) {                                  // wind (layt) and top drag (layu): the boundary layer of depth hsbl from above
        if (d.ocrp > 0.5) {
            for (int ilay = 1; ilay <= nlay; ++ilay) {
                double hcum = 0.0, sofar = 0.0;
                for (int k = 1; k < ilay; ++k) sofar = sofar + lay[k - 1][t];
                sofar = sofar + 0.0;                              // lay(ipnt,ilay) = 0 first (:1949,1995)
                for (int k = 1; k <= ilay; ++k) hcum = hcum + fmax(0.0, LL(d.hlay, q, k) - 1.5 * d.hsal);
                const double x = fmin(hcum, d.hsbl) / d.hsbl - sofar;
                lay[ilay - 1][t] = fmax(x, 0.0);
            }
        } else {
            for (int ilay = 1; ilay <= nlay; ++ilay) lay[ilay - 1][t] = (ilay == 1) ? 1.0 : 0.0;
        }
    } else {                                         // bottom drag (layb): depth hbbl from below
        if (d.ocrp > 0.5) {
            for (int ilay = nlay; ilay >= 1; --ilay) {
                double sofar = 0.0, hcum = 0.0;
                sofar = sofar + 0.0;                              // layb(ipnt,ilay) = 0 first (:1973)
                for (int k = ilay + 1; k <= nlay; ++k) sofar = sofar + lay[k - 1][t];
                for (int k = ilay; k <= nlay; ++k) hcum = hcum + LL(d.hlay, q, k);
                const double x = fmin(hcum, d.hbbl) / d.hbbl - sofar;
                lay[ilay - 1][t] = fmax(x, 0.0);
            }
        } else {
            for (int ilay = 1; ilay <= nlay; ++ilay) lay[ilay - 1][t] = (ilay == nlay) ? 1.0 : 0.0;
        }
    }
}
// bottom (TOP = false, :2015-2049) / top (:2075-2109) stress at the u and v point of cell ipnt
template <bool TOP>
__device__ __forceinline__ void stress_tau(const DevView &d, int ipnt, double &tau_u, double &tau_v) {
    const int nlay = d.nlay;
    int ilay = TOP ? 1 : nlay;
    if (d.ocrp > 0.5) {
        if (!TOP) { for (int k = nlay; k >= 1; --k) if (LL(d.hlay, ipnt, k) > 2.0 * d.hsal) { ilay = k; break; } }
        else      { for (int k = 1; k <= nlay; ++k) if (LL(d.hlay, ipnt, k) > 2.0 * d.hsal) { ilay = k; break; } }
    }
    const int32_t *row = d.neig + 8ll * ipnt;
    const int c1 = row[0], c3 = row[2], c4 = row[3], c5 = row[4], c7 = row[6], c8 = row[7];
    const double uu = LL(d.u, ipnt, ilay), vv = LL(d.v, ipnt, ilay);
    const double vatu = 0.25 * vv + 0.25 * LL(d.v, c3, ilay) + 0.25 * LL(d.v, c4, ilay) + 0.25 * LL(d.v, c5, ilay);
    const double uatv = 0.25 * uu + 0.25 * LL(d.u, c1, ilay) + 0.25 * LL(d.u, c7, ilay) + 0.25 * LL(d.u, c8, ilay);
    const double drg = TOP ? d.tdrg : d.bdrg;
    const double rh = d.rhon[ilay - 1];
    tau_u = uu * drg * rh * (d.qdrg * sqrt(uu * uu + vatu * vatu) + 1.0 - d.qdrg);
    tau_v = vv * drg * rh * (d.qdrg * sqrt(vv * vv + uatv * uatv) + 1.0 - d.qdrg);
}
__global__ __launch_bounds__(BEOM_BLOCK) void k_stress(DevView d, int wind, int bot, int top) {
    __shared__ double s_own[BEOM_MAX_LAYERS][BEOM_BLOCK], s_w[BEOM_MAX_LAYERS][BEOM_BLOCK], s_s[BEOM_MAX_LAYERS][BEOM_BLOCK];
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1, t = threadIdx.x;
    if (!cell_slot(d, ipnt)) return;                 // (no barrier below: every thread owns its columns of the three arrays)
    const int nlay = d.nlay;
    const int c5 = d.neig[8ll * ipnt + 4], c7 = d.neig[8ll * ipnt + 6];
    if (wind || top) {
        stress_fractions<true>(d, ipnt, s_own);
        if (wind)
            for (int ilay = 1; ilay <= nlay; ++ilay) {
                T3_(d.tt3d, ipnt, 1, ilay) = d.taus[ipnt] * s_own[ilay - 1][t];
                T3_(d.tt3d, ipnt, 2, ilay) = d.taus[ipnt + d.n1] * s_own[ilay - 1][t];
            }
        if (top) {
            double tu, tv;
            stress_tau<true>(d, ipnt, tu, tv);
            stress_fractions<true>(d, c5, s_w);
            stress_fractions<true>(d, c7, s_s);
            for (int ilay = 1; ilay <= nlay; ++ilay) {
                T3_(d.tu3d, ipnt, 1, ilay) = tu * 0.5 * (s_own[ilay - 1][t] + s_w[ilay - 1][t]);
                T3_(d.tu3d, ipnt, 2, ilay) = tv * 0.5 * (s_own[ilay - 1][t] + s_s[ilay - 1][t]);
            }
        }
    }
    if (bot) {
        double tu, tv;
        stress_tau<false>(d, ipnt, tu, tv);
        stress_fractions<false>(d, ipnt, s_own);
        stress_fractions<false>(d, c5, s_w);
        stress_fractions<false>(d, c7, s_s);
        for (int ilay = 1; ilay <= nlay; ++ilay) {
            T3_(d.tb3d, ipnt, 1, ilay) = tu * 0.5 * (s_own[ilay - 1][t] + s_w[ilay - 1][t]);
            T3_(d.tb3d, ipnt, 2, ilay) = tv * 0.5 * (s_own[ilay - 1][t] + s_s[ilay - 1][t]);
        }
    }
}

// ---- packed (the caller's Fortran storage) <-> padded device layout, one slice [0:ndeg] at a time ----
// src/dst element = `inner` consecutive T; AoS histories: K interleaved levels, level m selected.
// REMAP: the values themselves are cell indices (neig): translate them too.
// slot_of (embedded frames): packed index -> slot of the rectangle; null: the dense closed form
template <class T, bool TO_DEVICE, bool REMAP>
__global__ __launch_bounds__(BEOM_BLOCK) void k_repack(T *dev, T *host_img, long long ndeg, int L, int P, int inner, int K, int m,
                                                       const int32_t *slot_of) {
    const long long t = (long long)blockIdx.x * BEOM_BLOCK + threadIdx.x;
    const long long n = (ndeg + 1) * inner;
    if (t >= n) return;
    const long long ip = t / inner;
    const int e = (int)(t - ip * inner);
    long long dp = ip;
    if (slot_of) dp = slot_of[ip];
    else if (ip > 0) { const long long r = (ip - 1) / L; dp = 1 + r * P + ((ip - 1) - r * L); }
    if (TO_DEVICE) {
        T v = host_img[(ip * inner + e) * K + m];
        if (REMAP) {
            const long long q = (long long)v;
            if (q > 0) { if (slot_of) v = (T)slot_of[q]; else { const long long r = (q - 1) / L; v = (T)(1 + r * P + ((q - 1) - r * L)); } }
        }
        dev[dp * inner + e] = v;
    } else {
        host_img[(ip * inner + e) * K + m] = dev[dp * inner + e];
    }
}

// embedded frames: every slot of a device slice <- the sentinel's element(s) of the caller's slice (land slots keep them)
template <class T>
__global__ __launch_bounds__(BEOM_BLOCK) void k_fill_sentinel(T *dev, const T *host_img, long long nslots, int inner, int K, int m) {
    const long long t = (long long)blockIdx.x * BEOM_BLOCK + threadIdx.x;
    if (t >= nslots * inner) return;
    const int e = (int)(t % inner);
    dev[t] = host_img[(long long)e * K + m];
}

// ---- ghost-row exchange support: rows [jlo, jlo+nrows) of hlay,u,v,h_u,h_v <-> one buffer ----
// buffer layout: [field 0..4][layer][row][column], contiguous.  blockIdx.y = 1: a second group of rows <-> a second buffer
// in the same launch (a band's south and north side).
template <bool PACK>
__global__ __launch_bounds__(BEOM_BLOCK) void k_rows_copy(DevView d, int jlo, int nrows, double *buf, int jlo2, double *buf2) {
    if (blockIdx.y == 1) { jlo = jlo2; buf = buf2; }
    const long long per_lay = (long long)nrows * d.L;
    const long long total = 5ll * d.nlay * per_lay;
    const long long t = (long long)blockIdx.x * BEOM_BLOCK + threadIdx.x;
    if (t >= total) return;
    const int f = (int)(t / (d.nlay * per_lay));
    const long long r = t - (long long)f * d.nlay * per_lay;
    const int lay = (int)(r / per_lay);
    const long long c = r - (long long)lay * per_lay;
    double *fld = f == 0 ? d.hlay : f == 1 ? d.u : f == 2 ? d.v : f == 3 ? d.h_u : d.h_v;
    const long long row = c / d.L;
    const long long ip = 1 + (long long)(jlo - 1 + row) * (d.P ? d.P : d.L) + (c - row * d.L) + d.n1 * (long long)lay;
    if (PACK) buf[t] = fld[ip];
    else fld[ip] = buf[t];
}

// ---- device-side output preparation: write_array 'eta_','u___','v___'
//      (:2848-2883) and the scans of write_outputs (:2772-2808) ---------------------------
// eta: interface elevation accumulated bottom-up in real*4 exactly as the reference does
// (each partial sum is rounded to real*4 before the next layer is added, :2855-2858).
__global__ __launch_bounds__(BEOM_BLOCK) void k_out_convert(DevView d, const float *h0r4, float *eta, float *u4,
                                                            float *v4) {
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    if (!cell_slot(d, ipnt)) return;
    const long long nd = d.ndeg, pk = packed_index0(d, ipnt);     // the records are packed (ndeg, nlay) whatever the device pitch
    float acc = 0.f;
    for (int k = d.nlay; k >= 1; --k) {
        const long long o = pk + nd * (k - 1);
        const double h = LL(d.hlay, ipnt, k) - (double)h0r4[o];
        acc = (k == d.nlay) ? (float)h : (float)(h + (double)acc);
        if (d.rgld > 0.5 && k == 1) acc = (float)d.pi_s[ipnt];       // :2864-2872: the lid pressure in the top record
        if (eta) eta[o] = acc;
        if (u4) u4[o] = (float)LL(d.u, ipnt, k);
        if (v4) v4[o] = (float)LL(d.v, ipnt, k);
    }
}

// ---- the three `diag` records of write_array (private_mod.f95:2884-2974), real*4 (ndeg, nlay) packed: the Leith
//      viscosity diagnosed from u, v ('v_cc'), the Montgomery potential without its kinetic part ('mont') and the
//      potential vorticity ('pvor').  They are NOT the step's own scratch fields (other operation order, real*4
//      accumulation), so they are formed here from the state, with the caller's neig and mask arrays.
__global__ __launch_bounds__(BEOM_BLOCK) void k_diag_w12(DevView d, double *w1, double *w2) {      // :2892-2904
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const int32_t *nb = d.neig + 8ll * ipnt;
    const int c1 = nb[0], c3 = nb[2], c5 = nb[4], c7 = nb[6];
    LL(w1, ipnt, ilay) = ((LL(d.v, ipnt, ilay) - LL(d.v, c5, ilay)) / d.dl - (LL(d.u, ipnt, ilay) - LL(d.u, c7, ilay)) / d.dl) * d.mkpe[ipnt];
    LL(w2, ipnt, ilay) = (LL(d.u, c1, ilay) - LL(d.u, ipnt, ilay)) / d.dl + (LL(d.v, c3, ilay) - LL(d.v, ipnt, ilay)) / d.dl;
}
__global__ __launch_bounds__(BEOM_BLOCK) void k_diag_records(DevView d, const double *w1, const double *w2, float *pvor4, float *mont4,
                                                              float *vcc4) {
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const long long o = packed_index0(d, ipnt) + (long long)d.ndeg * (ilay - 1);
    const int32_t *nb = d.neig + 8ll * ipnt;
    const int c1 = nb[0], c2 = nb[1], c3 = nb[2], c5 = nb[4], c6 = nb[5], c7 = nb[6];
    if (vcc4) {                                                                                      // :2906-2927
        const double a0 = LL(w1, ipnt, ilay), a1 = LL(w1, c1, ilay), a2 = LL(w1, c2, ilay), a3 = LL(w1, c3, ilay);
        const double b0 = LL(w2, ipnt, ilay), b1 = LL(w2, c1, ilay), b3 = LL(w2, c3, ilay), b5 = LL(w2, c5, ilay), b7 = LL(w2, c7, ilay);
        const double q = (a1 - a0) * (a1 - a0) + (a2 - a3) * (a2 - a3) + (a3 - a0) * (a3 - a0) + (a2 - a1) * (a2 - a1)
                       + (b1 - b0) * (b1 - b0) + (b0 - b5) * (b0 - b5) + (b3 - b0) * (b3 - b0) + (b0 - b7) * (b0 - b7);
        vcc4[o] = (float)(d.bvis + d.dvis * (d.dl * d.dl) * sqrt(q));
    }
    if (mont4) {                                                                                     // :2930-2950
        const double mkn = d.mk_n[ipnt];
        float a = (float)(-(d.ocrp / (double)(d.nsal - 1) * d.hsal * mkn
                            * powi_dev(d.hsal / (d.hmin * (1.0 - mkn) + LL(d.hlay, ipnt, ilay)), d.nsal - 1)));
        for (int m = 1; m <= ilay - 1; ++m)
            a = a - (float)((d.rhon[ilay - 1] - d.rhon[m - 1]) * LL(d.hlay, ipnt, m) / d.rhon[ilay - 1]);
        double hsum = 0.0;                                                 // sum( hlay(ipnt, :) ): left to right
        for (int m = 1; m <= d.nlay; ++m) hsum = m == 1 ? LL(d.hlay, ipnt, 1) : hsum + LL(d.hlay, ipnt, m);
        mont4[o] = a + (float)(hsum - d.h_th[ipnt]);
    }
    if (pvor4) {                                                                                     // :2951-2974
        const double zeta = ((LL(d.v, ipnt, ilay) - LL(d.v, c5, ilay)) / d.dl - (LL(d.u, ipnt, ilay) - LL(d.u, c7, ilay)) / d.dl) * d.mkpe[ipnt];
        pvor4[o] = (float)((d.fcor[ipnt] + zeta * d.uadv) * d.mkpi[ipnt] * (d.mk_n[ipnt] + d.mk_n[c5] + d.mk_n[c7] + d.mk_n[c6])
                           / (LL(d.hlay, ipnt, ilay) + LL(d.hlay, c5, ilay) + LL(d.hlay, c6, ilay) + LL(d.hlay, c7, ilay)));
    }
}

// per-workgroup partial min/max of h (wet cells), u, v (their points, or all cells 0..ndeg when a
// mask is empty, :2776-2793) and the thin-layer flag (:2799-2800); the host reduces the partials.
// out: [block][layer][7] = hmin,hmax,umin,umax,vmin,vmax,thin
__global__ __launch_bounds__(BEOM_BLOCK) void k_out_scan(DevView d, int any_u, int any_v, double *out) {
    __shared__ double red[BEOM_BLOCK / 64][7];
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x;          // 0..ndeg
    const bool in = cell_slot(d, ipnt);
    const double big = 1.7976931348623157e308;
    for (int k = 1; k <= d.nlay; ++k) {
        double v[7] = {big, -big, big, -big, big, -big, 0.0};
        if (in) {
            const double h = LL(d.hlay, ipnt, k), uu = LL(d.u, ipnt, k), vv = LL(d.v, ipnt, k);
            if (d.mk_n[ipnt] > 0.5) { v[0] = h; v[1] = h; if (ipnt >= 1 && h < 0.5 * d.hmin) v[6] = 1.0; }
            if (!any_u || d.mk_u[ipnt] > 0.5) { v[2] = uu; v[3] = uu; }
            if (!any_v || d.mk_v[ipnt] > 0.5) { v[4] = vv; v[5] = vv; }
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            double x = v[q];
            for (int off = 32; off > 0; off >>= 1) {
                const double y = __shfl_down(x, off, 64);
                x = (q == 0 || q == 2 || q == 4) ? fmin(x, y) : fmax(x, y);
            }
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = x;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int q = threadIdx.x;
            double x = red[0][q];
            for (int w = 1; w < BEOM_BLOCK / 64; ++w)
                x = (q == 0 || q == 2 || q == 4) ? fmin(x, red[w][q]) : fmax(x, red[w][q]);
            out[((long long)blockIdx.x * d.nlay + (k - 1)) * 7 + q] = x;
        }
        __syncthreads();
    }
}

// ---- no_gradient_obc, private_mod.f95:2613-2679: vanishing normal derivative of the velocity
//      at nudged open boundaries.  pass 0 = tangential component at the wet boundary cell
//      (:2624-2651), pass 1 = normal component at the boundary point (:2657-2678).  One thread
//      per (segment, layer); beom_set_open_boundaries checked that no thread of a pass reads a
//      value another thread of the same pass writes (the reference loops are serial). ----------
__global__ __launch_bounds__(BEOM_BLOCK) void k_no_gradient_obc(DevView d, int pass) {
    const int is = blockIdx.x * BEOM_BLOCK + threadIdx.x;
    const int ilay = blockIdx.y + 1;
    if (is >= d.nseg) return;
#define SEG(col) d.segm[(long long)is + (long long)d.nseg * ((col) - 1)]
    const bool ns = SEG(5) == 1, ew = SEG(4) == 1;       // northern/southern | eastern/western boundary
    const int ipnt = pass == 0 ? SEG(10) : SEG(1);
    const int in = pass == 0 ? SEG(16) : SEG(13);
    if (ipnt < 1) return;                                // (this pass of the segment is another band's)
    // which component: pass 0 -> u on N/S boundaries, v on E/W; pass 1 -> v on N/S, u on E/W
    const bool do_u = pass == 0 ? ns : (!ns && ew);
    const bool do_v = pass == 0 ? (!ns && ew) : ns;
    if (do_u) {
        const double mk = d.mk_u[ipnt];
        if (pass == 1 || mk > 0.5) {
            const double un = LL(d.u, in, ilay) - FNUD_(in, ilay, 2) + FNUD_(ipnt, ilay, 2);
            LL(d.u, ipnt, ilay) = un;
            LL(d.h_u, ipnt, ilay) = un * (LL(d.hlay, ipnt, ilay) + LL(d.hlay, d.neig[8ll * ipnt + 4], ilay)) / (1.0 + mk);
        }
    } else if (do_v) {
        const double mk = d.mk_v[ipnt];
        if (pass == 1 || mk > 0.5) {
            const double vn = LL(d.v, in, ilay) - FNUD_(in, ilay, 3) + FNUD_(ipnt, ilay, 3);
            LL(d.v, ipnt, ilay) = vn;
            LL(d.h_v, ipnt, ilay) = vn * (LL(d.hlay, ipnt, ilay) + LL(d.hlay, d.neig[8ll * ipnt + 6], ilay)) / (1.0 + mk);
        }
    }
#undef SEG
}

// ---- rigid lid (rgld = 1), the fork's addition: private_mod.f95:1648-1700, 1705-1838, 2237-2257, 2292-2314 ----------
// (a fork-specific option outside every BASELINE configuration: table-driven kernels, one launch per piece)
__global__ __launch_bounds__(BEOM_BLOCK) void k_rgld_h_epilogue(DevView d) {                 // :1648-1700
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    if (!cell_slot(d, ipnt)) return;
    const int l2 = d.nlay < 2 ? d.nlay : 2;
    for (int pass = 1; pass <= 2; ++pass) {
        double s = LL(d.hlay, ipnt, 1);
        for (int i = 2; i <= d.nlay; ++i) s = s + LL(d.hlay, ipnt, i);
        const float corr = 0.5f * (float)(s - d.h_th[ipnt]);       // `0.5*real(...)`: DEFAULT real, rounded before the subtraction
        const int il = pass == 1 ? 1 : l2;
        LL(d.hlay, ipnt, il) = LL(d.hlay, ipnt, il) - (double)corr;
    }
}
__global__ __launch_bounds__(BEOM_BLOCK) void k_rgld_upstream_fluxes(DevView d) {            // :2237-2257, 2292-2314
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const int c5 = d.neig[8ll * ipnt + 4], c7 = d.neig[8ll * ipnt + 6];
    const int nl = d.nlay;                     // the reference's 2-D d2hx, d2hy: as the LAST Montgomery sweep (layer nlay) left them
    double hcen = (LL(d.hlay, c5, ilay) + LL(d.hlay, ipnt, ilay)) / (1.0 + d.mk_u[ipnt]);
    const double uu = LL(d.u, ipnt, ilay);
    LL(d.h_u, ipnt, ilay) = 0.5 * (uu + fabs(uu)) * (hcen - 0.16667 * LL(d.d2hx, c5, nl))
                          + 0.5 * (uu - fabs(uu)) * (hcen - 0.16667 * LL(d.d2hx, ipnt, nl));
    hcen = (LL(d.hlay, c7, ilay) + LL(d.hlay, ipnt, ilay)) / (1.0 + d.mk_v[ipnt]);
    const double vv = LL(d.v, ipnt, ilay);
    LL(d.h_v, ipnt, ilay) = 0.5 * (vv + fabs(vv)) * (hcen - 0.16667 * LL(d.d2hy, c7, nl))
                          + 0.5 * (vv - fabs(vv)) * (hcen - 0.16667 * LL(d.d2hy, ipnt, nl));
}
// right-hand side of the Poisson equation (:1723-1755).  The reference scatters (pi_rhs(ipnt) -= f, pi_rhs(c5) += f) in
// packed order, layer by layer: per cell that is a fixed sequence of terms per layer — its own at "time" ipnt, and one
// from every cell whose neig(5) (neig(7)) it is at that cell's index (its eastern / northern neighbour; under
// periodicity also cells of the orphan column / row).  beom_set_rigid_lid lists them per cell in exactly that order
// (code 0: -h_u, 1: +h_u, 2: -h_v, 3: +h_v of the source cell); this kernel adds them up.
__global__ __launch_bounds__(BEOM_BLOCK) void k_rgld_rhs(DevView d) {
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    if (!cell_slot(d, ipnt)) return;
    const double den = d.dl * d.dt;
    const int e0 = d.lid_rhs_start[ipnt], e1 = d.lid_rhs_start[ipnt + 1];
    double r = 0.0;
    for (int ilay = d.nlay; ilay >= 1; --ilay)
        for (int e = e0; e < e1; ++e) {
            const int ent = d.lid_rhs_ent[e], src = ent >> 2, code = ent & 3;
            const double f = (code < 2 ? LL(d.h_u, src, ilay) : LL(d.h_v, src, ilay)) / den;
            r = (code & 1) ? r + f : r - f;
        }
    d.pi_rhs[ipnt] = r;
}
// Gauss-Seidel sweeps in packed order (rp = 1, :1757-1802), the serial arithmetic kept: a cell reads the NEW pressure of the
// neighbours before it in packed order and the OLD one of those after it.  That is a dependency graph; its levels within one
// sweep (beom_set_rigid_lid; on a plain frame the anti-diagonals i + j) say which cells may be updated side by side, and a
// cell of sweep s + 1 only needs the cells of sweep s one level further on — so SEVERAL SWEEPS ARE IN FLIGHT AT ONCE, a
// pipeline of wavefronts: cell p of sweep s runs at time level(p) + (s - s0) * D (D = 1 + the largest level difference along
// an edge: 2 on a plain frame).  Every sweep writes its own copy of the pressure (a ring of B + 1 copies; sweep s reads copy
// s for the neighbours before p and copy s - 1 for those after p and for p itself), so the sweeps that run on past the one
// the convergence test picks leave that one intact.  One launch per time step: blockIdx.y = the sweep of the batch, the
// blocks of a sweep cover its level at that time.  maxd[b] = max |change| of sweep s0 + b (bit pattern of a non-negative
// double: ordered like the unsigned integer).
__global__ __launch_bounds__(BEOM_BLOCK) void k_rgld_gs_front(DevView d, double *ring, int nring, int s0, int t, int dstep,
                                                               unsigned long long *maxd) {
    const int b = blockIdx.y, lev = t - b * dstep;
    if (lev < 0 || lev >= d.sor_ndiag) return;
    const int k = d.sor_dstart[lev] + blockIdx.x * BEOM_BLOCK + threadIdx.x;
    double diff = 0.0;
    if (k < d.sor_dstart[lev + 1]) {
        const double rp = 1.000;
        const int s = s0 + b;
        double *cur = ring + (long long)(s % nring) * d.n1;
        const double *prv = ring + (long long)((s - 1) % nring) * d.n1;
        const int ipnt = d.sor_order[k];
        const int i = d.subc[ipnt], j = d.subc[ipnt + d.n1];
        const int32_t *nb = d.neig + 8ll * ipnt;
        const double old = prv[ipnt], os_ = d.Osum_[ipnt];
        double x = (1 - rp) * old - rp * os_ * d.pi_rhs[ipnt];
        if (i < d.lm) { const int c1 = nb[0]; x = x + rp * os_ * d.Ow[c1] * (c1 < ipnt ? cur : prv)[c1]; }
        if (j < d.mm) { const int c3 = nb[2]; x = x + rp * os_ * d.Os[c3] * (c3 < ipnt ? cur : prv)[c3]; }
        if (i > 1) { const int c5 = nb[4]; x = x + rp * os_ * d.Ow[ipnt] * (c5 < ipnt ? cur : prv)[c5]; }
        if (j > 1) { const int c7 = nb[6]; x = x + rp * os_ * d.Os[ipnt] * (c7 < ipnt ? cur : prv)[c7]; }
        cur[ipnt] = x;
        diff = fabs(x - old);
    }
    for (int off = 32; off > 0; off >>= 1) diff = fmax(diff, __shfl_down(diff, off, 64));
    if ((threadIdx.x & 63) == 0 && diff > 0.0) atomicMax(&maxd[b], (unsigned long long)__double_as_longlong(diff));
}
__global__ __launch_bounds__(BEOM_BLOCK) void k_rgld_correct(DevView d) {                    // :1806-1833
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const int i = d.subc[ipnt], j = d.subc[ipnt + d.n1];
    if (i > 1 && i < d.lm + 1) {
        const int c5 = d.neig[8ll * ipnt + 4];
        double uu = LL(d.u, ipnt, ilay) - d.dt / d.dl * d.pi_s[ipnt];
        LL(d.u, ipnt, ilay) = uu + d.dt / d.dl * d.pi_s[c5];
    }
    if (j > 1 && j < d.mm_glob + 1) {
        const int c7 = d.neig[8ll * ipnt + 6];
        double vv = LL(d.v, ipnt, ilay) - d.dt / d.dl * d.pi_s[ipnt];
        LL(d.v, ipnt, ilay) = vv + d.dt / d.dl * d.pi_s[c7];
    }
}

// ---- biharmonic viscosity, update_viscosity's svis > 0 part (private_mod.f95:2508-2599).
//      A fork-specific option outside every BASELINE configuration: always via the neig/mask
//      tables, five separate sweeps.  The reference rounds two expressions to DEFAULT real
//      (`real(x)` without a kind, :2565 and :1472); so do we. ------------------------------------
__global__ __launch_bounds__(BEOM_BLOCK) void k_biharm_lap(DevView d, int ilay0) {             // :2508-2550
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = ilay0 ? ilay0 : (int)blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const int32_t *nb = d.neig + 8ll * ipnt;
    const int c1 = nb[0], c3 = nb[2], c5 = nb[4], c7 = nb[6];
    const double dl = d.dl;
    double du = 0.0, dv = 0.0;
    if (d.mk_u[ipnt] > 0.5) {
        du = du + 1.0 / (dl * dl) * (d.mk_u[c1] * LL(d.u, c1, ilay) + d.mk_u[c3] * LL(d.u, c3, ilay)
                                     + d.mk_u[c5] * LL(d.u, c5, ilay) + d.mk_u[c7] * LL(d.u, c7, ilay));
        du = du - 1.0 / (dl * dl) * (d.mk_u[c1] + d.mk_u[c3] + d.mk_u[c5] + d.mk_u[c7]) * LL(d.u, ipnt, ilay);
    }
    if (d.mk_v[ipnt] > 0.5) {
        dv = dv + 1.0 / (dl * dl) * (d.mk_v[c1] * LL(d.v, c1, ilay) + d.mk_v[c3] * LL(d.v, c3, ilay)
                                     + d.mk_v[c5] * LL(d.v, c5, ilay) + d.mk_v[c7] * LL(d.v, c7, ilay));
        dv = dv - 1.0 / (dl * dl) * (d.mk_v[c1] + d.mk_v[c3] + d.mk_v[c5] + d.mk_v[c7]) * LL(d.v, ipnt, ilay);
    }
    LL(d.delu, ipnt, ilay) = du;
    LL(d.delv, ipnt, ilay) = dv;
}
__global__ __launch_bounds__(BEOM_BLOCK) void k_biharm_flux(DevView d, int ilay0) {            // :2557-2598
    const int ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
    const int ilay = ilay0 ? ilay0 : (int)blockIdx.y + 1;
    if (!cell_slot(d, ipnt)) return;
    const int32_t *nb = d.neig + 8ll * ipnt;
    const int c1 = nb[0], c3 = nb[2], c5 = nb[4], c6 = nb[5], c7 = nb[6];
    const double dl = d.dl;
    const double h = LL(d.hlay, ipnt, ilay);
    const double hh_q = (double)(float)(h + d.mk_n[c5] * LL(d.hlay, c5, ilay) + d.mk_n[c6] * LL(d.hlay, c6, ilay)
                                        + d.mk_n[c7] * LL(d.hlay, c7, ilay))
                        / (1.0 + d.mk_n[c5] + d.mk_n[c6] + d.mk_n[c7]);
    double uu = 0.0, vv = 0.0;
    uu = uu - 1.0 / dl * h * LL(d.delu, ipnt, ilay) + 1.0 / dl * h * LL(d.delv, ipnt, ilay);
    vv = vv + 1.0 / dl * hh_q * LL(d.delu, ipnt, ilay) + 1.0 / dl * hh_q * LL(d.delv, ipnt, ilay);
    const int si = d.subc[ipnt], sj = d.subc[ipnt + d.n1];
    if (si <= d.lm - 1) uu = uu + 1.0 / dl * h * LL(d.delu, c1, ilay);
    if (sj <= d.mm_glob - 1) uu = uu - 1.0 / dl * h * LL(d.delv, c3, ilay);
    if (si > 1) vv = vv - 1.0 / dl * hh_q * LL(d.delv, c5, ilay);
    if (sj > 1) vv = vv - 1.0 / dl * hh_q * LL(d.delu, c7, ilay);
    if (d.mk_u[ipnt] * d.mk_v[ipnt] < 0.5) vv = 0.0;
    LL(d.uu4, ipnt, ilay) = uu;
    LL(d.vv4, ipnt, ilay) = vv;
}
