// beom_dev.h — device-side view of the engine state and neighbour lookup.
//
// Device layout = the reference's packed layout (private_mod.f95:27-93):
// X(0:ndeg, nlay) at x[ipnt + n1*(ilay-1)], n1 = ndeg+1, index 0 = land sentinel that
// is never written.  Two deliberate differences, both invisible through the C-ABI:
//   * history arrays are SoA and rotated by pointer instead of copied
//     (rs_h(2,..) -> rs[2], dmdx(3,..) -> dmx[3], dmdy(3,..) -> dmy[3]);
//   * the six diagnostics of update_mont_rvor_pvor_dive_kine get a layer dimension so
//     that all layers of a sweep run in one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/beom_hip.h"

struct DevView {
    // sizes
    int ndeg, nlay, lm, mm, nsal, variant;
    int mm_glob;                  // mm of the whole frame (= mm unless the handle is a j-slab)
    long long n1;                 // ndeg + 1
    int L, M, xper, yper;         // dense closed form: L = lm+1, M = mm+1 (local rows)
    // Dense handles keep every (0:ndeg) array with a PADDED row pitch on the device (invisible through the
    // C-ABI: uploads scatter, downloads gather): cell (i, j) lives at i + (j-1)*P, P = L rounded up to 16
    // doubles, and the arrays are placed so that index 1 is 128-byte aligned — every 64-cell tile row is then
    // four whole cache lines (a tile row of the packed pitch lm+1 straddles five, and two workgroups write each
    // shared line in part: tools/micro/march2.hip measures 3.9 vs 5.4 TB/s for the u+v sweep's shape).
    // P = 0: packed layout (handles on the table path).  ncell = last cell index (P*M, or ndeg).
    int P;
    long long ncell;
    // Frames WITH LAND on the same rectangle ("embedded", round 2): every (i, j) of the (lm+1) x (mm+1) frame has a slot;
    // slots that are not packed cells of the caller's vector hold the sentinel's values and are never written, so a
    // neighbour taken by offset reads there exactly what the reference reads at index 0.  Masks then come from the
    // caller's arrays, and only tiles whose whole stencil is wet interior (reg4) take the mask-free paths.
    int embedded;
    const int32_t *pk_of;         // slot -> packed index (ipnt of the caller), 0 = not a packed cell; null unless embedded
    const unsigned char *reg4;    // per 64 x 4 tile: 1 = every cell within 3 cells of it is wet interior with unit masks
    int reg_nx;                   // tiles per row of reg4
    // nudged frames on the rectangle: per 64 x 4 tile, bit iv-1 = some cell of the tile has a non-zero relaxation rate
    // nudg(:, iv) (iv = 1 eta, 2 u, 3 v).  Sponges are a few rows or columns: elsewhere a wave of update_h learns from ONE
    // byte that its 64 rates are zero instead of loading them; null = no table (load always).  (The momentum sweep does not
    // use it: the look-up cost k_uv_fused 6-9 VGPRs in every instantiation — three waves per SIMD for the zero-viscosity
    // forms — for 2.6 % on the one frame it helped.)
    const unsigned char *ngt;
    int ngt_nx;
    int joff, Mg, slab;           // j-slab: local row j is global row j + joff of Mg = mm_global+1 rows
    // rows a launch may WRITE: up to two strips [jlo, jhi] (local, inclusive); default one strip 1..M.
    // Used to split a step into an interior pass and an edge pass around the ghost-row exchange.
    int nstrip, jlo0, jhi0, jlo1, jhi1;   // (scalars, never indexed: indexing a kernel argument sends it to scratch)
    // static
    const int32_t *neig, *subc;
    const double *mk_u, *mk_v, *mk_n, *mkpe, *mkpi, *fcor, *h_th, *h_to;
    const double *nudg, *fnud, *hdot, *tide, *bodf, *taus;
    // prognostic
    double *hlay, *u, *v, *h_u, *h_v;
    double *rs[2];                // rs[0] = rs_h(1,..) older, rs[1] = rs_h(2,..) newer
    double *dmx[4], *dmy[4];      // [0] oldest .. [2] newest; [3] spare (fused U+V sweep writes there)
    double *u_alt, *v_alt, *hu_alt, *hv_alt;   // ping-pong partners of u, v, h_u, h_v (fused U+V sweep)
    double *v_cc, *v_ll, *tt3d, *tb3d, *tu3d;
    // per-layer diagnostics (0:ndeg, nlay)
    double *mont, *rvor, *pvor, *dive, *d2hx, *d2hy;
    // fused Montgomery+Leith sweep: products v_cc*dive and v_ll*rvor (0:ndeg, nlay)
    double *pcd, *qlr;
    int keep_diag;                // fused sweep also stores rvor, dive, v_cc, v_ll
    int keep_visc;                // fused sweep (Leith) also stores v_cc, v_ll: they stand for n_3d - 1 more steps
    int zero_visc;                // v_cc = v_ll = +0 everywhere and never refreshed: the viscous products are +-0
    int lean_d2h;                 // fused sweep stores d2hx, d2hy only where the fused u+v sweep reads them
    int edge_global;              // k_uv_fused: edge workgroups read global memory throughout (A/B switch; default: staged by lookup)
    // biharmonic viscosity (svis > 0, :2508-2599): Laplacians and thickness-weighted fluxes
    double *delu, *delv, *uu4, *vv4; double svis;
    // packed layout: per run of 64 cells (dN, dS) if the run is a uniform wet interior, else (0, 0); may be null
    const int32_t *woff;
    // nudged open-boundary segments, Fortran segm(nseg, 18) (no_gradient_obc, :2613-2679)
    const int32_t *segm; int nseg;
    // rigid lid (rgld = 1, private_mod.f95:505-563, 1705-1838): lid pressure, Poisson operators, right-hand side, previous
    // iterate; the packed cells in Gauss-Seidel wavefront order (levels of the serial sweep's dependency graph)
    double *pi_s, *pi_rhs, *pi_prev;
    const double *Ow, *Os, *Osum_;
    const int32_t *sor_order, *sor_dstart;
    const int32_t *lid_rhs_start, *lid_rhs_ent;  // per cell: its terms of the Poisson right-hand side in the serial loops' order (4 * source + code)
    int sor_ndiag;
    // constants by value
    double dl, dt, grav, rho0, beta, epsi, gamm, del1, del2, hmin, hsal, bvis, dvis, bdrg, tdrg,
        qdrg, hsbl, hbbl, uadv, ocrp, rgld, invf, w_ti;
    double rhon[BEOM_MAX_LAYERS];
    // uniform reciprocals, divided once on the host (IEEE division: same bits as on the device)
    double i_dl, i_gr, i_ns, i_r0, i_r1, i_rn[BEOM_MAX_LAYERS];
    // which optional terms are live (wave-uniform branches)
    int has_hdot, has_tide, has_bodf, has_nudg, has_stress, has_wind, has_hto;
    // distribute_stress inside the fused momentum sweep (uv_core; set per launch by the engine: ocrp = 0, a stress refresh on
    // every step, steps > 3): the wind stress at real cells only (0 in the sentinel and in every slot that is no cell, as
    // tt3d is), bottom / top drag switched on, the densities of the top and the bottom layer
    int stress_fold, has_bot, has_top;
    const double *taus_cells;
    double rho_top, rho_bot;
};

// ---- tiles of the selected strips --------------------------------------------------------
// A launch covers the tile rows (TY rows each) that intersect the strips.  Consecutive workgroups go round-robin to the 8
// XCDs (MI355X_MICROARCH.md), so workgroup b serves XCD b & 7, and each XCD gets one contiguous band of the tile rows (its
// L2 keeps the neighbours), swept x fastest: HBM wants the long contiguous row streams (strips a few tiles wide, which
// would keep the halo rows in L2, made every sweep slower: profiles/r02_ab_experiments.txt).
//  * Whole tile rows per XCD leave the last XCD short: 65 tile rows are 9 + ... + 9 + 2 and the launch lasts as long as the
//    XCDs with 9 (+11 %).  Where that costs more than 3 % on a launch of several rounds, the bands are cut to the tile
//    instead (an eighth of the tiles in row-major order each): 4096 x 512 x 4 -3 %, 4096 x 1024 x 4 -2 %.  (Frames whose
//    rows divide well keep whole rows: cutting mid-row cost 4096^2 x 4 +0.9 %.)
//  * The tiles on the frame's rim take the general paths (masks, wraps) and live longer than interior ones, so on launches
//    of at most 4096 tiles — which are mostly tail — they are handed out FIRST: the rim columns at the head of every
//    tile row, the top rim row at the head of the last XCD's band (soliton 2048 x 256: -3 %).
struct TileMap {
    int tr0a, ntra, tr0b, ntrb, total, gx, rpx, tpx, nt, xlast, nlast, rim_first, by_tiles;
    // forceinline: an out-of-line call would take the address of the kernel argument and push all
    // of DevView (~1 KB per lane) into scratch memory
    __host__ __device__ __forceinline__ TileMap(const DevView &d, int TX, int TY) {
        gx = (d.L + TX - 1) / TX;
        tr0a = 0; ntra = 0; tr0b = 0; ntrb = 0;
        if (d.jhi0 >= d.jlo0) { tr0a = (d.jlo0 - 1) / TY; ntra = (d.jhi0 - 1) / TY - tr0a + 1; }
        if (d.nstrip > 1 && d.jhi1 >= d.jlo1) { tr0b = (d.jlo1 - 1) / TY; ntrb = (d.jhi1 - 1) / TY - tr0b + 1; }
        total = ntra + ntrb;
        nt = total * gx;                                    // tiles of the launch
        rpx = (total + 7) / 8;                              // tile rows per XCD
        rim_first = nt <= 4096;
        by_tiles = !rim_first && (rpx * 8 - total) * 100 > 3 * total;
        tpx = by_tiles ? (nt + 7) / 8 : rpx * gx;           // workgroups per XCD
        xlast = total > 0 ? (total - 1) / rpx : 0;          // (whole rows) the last XCD that has tile rows, and how many
        nlast = total - xlast * rpx;
    }
    __host__ unsigned blocks() const { return (unsigned)(8 * tpx); }
    // blockIdx.x -> tile row (absolute) and column chunk; false if this workgroup has no tile
    __device__ __forceinline__ bool locate(int b, int &ty, int &ch) const {
        const int xcd = b & 7, k = b >> 3;
        int vt;
        if (by_tiles) {
            const int v = xcd * tpx + k;
            if (v >= nt) return false;
            vt = v / gx; ch = v - vt * gx;
        } else {
            int rib = k / gx;
            const int c = k - rib * gx;
            ch = c;
            if (rim_first) {
                ch = c == 0 ? 0 : (c == 1 ? gx - 1 : c - 1);              // rim columns first
                if (xcd == xlast && rib < nlast) rib = nlast - 1 - rib;    // the top band top-down
            }
            vt = xcd * rpx + rib;
            if (rib >= rpx || vt >= total) return false;
        }
        ty = vt < ntra ? tr0a + vt : tr0b + (vt - ntra);
        return true;
    }
};
__host__ __device__ __forceinline__ bool row_selected(const DevView &d, int j) {
    return (j >= d.jlo0 && j <= d.jhi0) || (d.nstrip > 1 && j >= d.jlo1 && j <= d.jhi1);
}
// embedded frames: is the 64 x ny tile at (x0, y0) (1-based, ny = 4 or 8, aligned) regular — no land, coast or mask
// within 3 cells of it?  (frames without land: always)
__device__ __forceinline__ bool tile_regular(const DevView &d, int x0, int y0, int ny) {
    if (!d.embedded) return true;
    const int tx = (x0 - 1) >> 6, t4 = (y0 - 1) >> 2;
    if (tx < 0 || tx >= d.reg_nx || t4 < 0 || (t4 + (ny >> 2) - 1) * 4 >= d.M) return false;
    bool r = d.reg4[(long long)t4 * d.reg_nx + tx] != 0;
    if (ny > 4) r = r && d.reg4[(long long)(t4 + 1) * d.reg_nx + tx] != 0;
    return r;
}
__device__ __forceinline__ bool slot_is_cell(const DevView &d, int ip) { return !d.embedded || d.pk_of[ip] != 0; }

// launches over ALL cell slots 0..d.ncell (stress, output scans, ...): false for the padding slots of the dense layout
__device__ __forceinline__ bool cell_slot(const DevView &d, long long ip) {
    if (ip > d.ncell) return false;
    if (d.P == 0 || ip == 0) return true;
    if (d.embedded) return d.pk_of[ip] != 0;
    return (int)((ip - 1) % d.P) < d.L;
}
// 0-based packed index (the reference's ipnt - 1) of a real cell slot
__device__ __forceinline__ long long packed_index0(const DevView &d, long long ip) {
    if (d.P == 0) return ip - 1;
    if (d.embedded) return (long long)d.pk_of[ip] - 1;
    const long long r = (ip - 1) / d.P;
    return r * d.L + ((ip - 1) - r * d.P);
}

// ---- cell contexts: where a thread is, who its neighbours are, what its masks are ----
// Slots follow private_mod.f95:28-30: 1=E 2=NE 3=N 4=NW 5=W 6=SW 7=S 8=SE.
#define BEOM_BLOCK 256

template <int K> struct NbOff {
    static constexpr int di = (K == 1 || K == 2 || K == 8) ? 1 : ((K == 4 || K == 5 || K == 6) ? -1 : 0);
    static constexpr int dj = (K == 2 || K == 3 || K == 4) ? 1 : ((K == 6 || K == 7 || K == 8) ? -1 : 0);
};

// Any coastline: neighbours and masks come from the caller's tables.
// Packed (table) layout, a wave whose 64 cells are all wet interior cells with wet neighbours at the
// same index offsets (E/W = +-1, N = +dN, S = -dS for the whole wave; beom_create builds the per-wave
// table woff): neighbours by arithmetic, every mask 1 — no table traffic, as in the dense interior.
struct CellPackedInt {
    static constexpr bool kLanesAreRowNeighbours = true;
    static constexpr bool kHasIJ = false;
    int ipnt, dN, dS;
    const DevView *dv;
    template <int K> __device__ __forceinline__ int nb() const {
        constexpr int di = NbOff<K>::di, dj = NbOff<K>::dj;
        return ipnt + di + (dj > 0 ? dN : (dj < 0 ? -dS : 0));
    }
    __device__ __forceinline__ double mk_n() const { return 1.0; }
    __device__ __forceinline__ double mk_u() const { return 1.0; }
    __device__ __forceinline__ double mk_v() const { return 1.0; }
    __device__ __forceinline__ double mkpe() const { return 1.0; }
    __device__ __forceinline__ double mkpi() const { return 1.0; }
    template <int K> __device__ __forceinline__ double mk_n_nb(int) const { return 1.0; }
    __device__ __forceinline__ int isub() const { return dv->subc[ipnt]; }
    __device__ __forceinline__ bool wave_is_interior() const { return true; }
    __device__ __forceinline__ const CellPackedInt &as_interior() const { return *this; }
};

struct CellGather {
    static constexpr bool kLanesAreRowNeighbours = false;
    static constexpr bool kHasIJ = false;
    int ipnt;
    const int32_t *row;
    const DevView *dv;
    static dim3 grid(const DevView &d, int nz) {
        return dim3((unsigned)((d.ndeg + BEOM_BLOCK - 1) / BEOM_BLOCK), (unsigned)nz, 1);
    }
    __device__ __forceinline__ bool init(const DevView &d) {
        ipnt = blockIdx.x * BEOM_BLOCK + threadIdx.x + 1;
        row = d.neig + 8ll * ipnt;
        dv = &d;
        return ipnt <= d.ndeg;
    }
    template <int K> __device__ __forceinline__ int nb() const { return row[K - 1]; }
    __device__ __forceinline__ double mk_n() const { return dv->mk_n[ipnt]; }
    __device__ __forceinline__ double mk_u() const { return dv->mk_u[ipnt]; }
    __device__ __forceinline__ double mk_v() const { return dv->mk_v[ipnt]; }
    __device__ __forceinline__ double mkpe() const { return dv->mkpe[ipnt]; }
    __device__ __forceinline__ double mkpi() const { return dv->mkpi[ipnt]; }
    template <int K> __device__ __forceinline__ double mk_n_nb(int c) const { return dv->mk_n[c]; }
    __device__ __forceinline__ int isub() const { return dv->subc[ipnt]; }
    // wave-uniform (a 256-thread workgroup = four aligned runs of 64 packed cells)
    __device__ __forceinline__ bool wave_is_interior() const {
        if (!dv->woff) return false;
        const int w = __builtin_amdgcn_readfirstlane((ipnt - 1) >> 6);
        return dv->woff[2 * w] != 0;
    }
    __device__ __forceinline__ CellPackedInt as_interior() const {
        const int w = __builtin_amdgcn_readfirstlane((ipnt - 1) >> 6);
        CellPackedInt r; r.ipnt = ipnt; r.dN = dv->woff[2 * w]; r.dS = dv->woff[2 * w + 1]; r.dv = dv;
        return r;
    }
};

// Dense frame (interior entirely wet; every BASELINE config).  Closed form of SURVEY
// App. A: neighbour (di,dj) of (i,j) is indc0(wx(i+di), wy(j+dj)), the periodic wraps of
// index_grid_points (private_mod.f95:614-685) acting on the TARGET coordinate; masks are
// the predicates of :701-714 (+ :621,627,649,655,676 when periodic).  Both are verified
// cell by cell against the caller's tables in beom_create — no table traffic at run time.
//
// Launch shape: a 256-thread workgroup covers 64 columns x 4 consecutive rows (one wave
// per row, so the N/S rows a wave re-reads were just touched by its sibling waves on the
// same CU).  Block -> tile map is XCD-aware: consecutive workgroups go round-robin to the
// 8 XCDs (MI355X_MICROARCH.md), so XCD x sweeps its own band of tile rows and the
// neighbour rows it re-reads stay in ITS L2.
// INTERIOR = true is the specialisation for waves whose 64 cells all have
// 2 <= i <= L-2 and 2 <= j <= M-2: every neighbour is a plain offset and every mask is 1
// (x*1.0 folds exactly), chosen per wave by a scalar test in the kernels.
#ifndef BEOM_TILE_X
#define BEOM_TILE_X 64
#endif
#define BEOM_TILE_Y (BEOM_BLOCK / BEOM_TILE_X)
#define BEOM_TILE_WX (BEOM_TILE_X / 64)

template <bool INTERIOR>
struct CellDenseT {
    // lane l and lane l+1 of an INTERIOR wave hold cells (i, j) and (i+1, j): an east/west
    // neighbour value is one wavefront shuffle away instead of one more load
    static constexpr bool kLanesAreRowNeighbours = INTERIOR;
    static constexpr bool kHasIJ = true;
    int i, j, ipnt, L, M, P, xper, yper;
    int jg, Mg, ywrap;            // global row / row count (masks); y wrap only when not a slab
    const DevView *dv;            // embedded frames: masks from the caller's arrays
    static dim3 grid(const DevView &d, int nz) {
        return dim3(TileMap(d, BEOM_TILE_X, BEOM_TILE_Y).blocks(), (unsigned)nz, 1);
    }
    __device__ __forceinline__ bool init(const DevView &d) {
        L = d.L; M = d.M; P = d.P; xper = d.xper; yper = d.yper;
        const TileMap tm(d, BEOM_TILE_X, BEOM_TILE_Y);
        int ty, ch;
        if (!tm.locate(blockIdx.x, ty, ch)) return false;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        j = ty * BEOM_TILE_Y + wave / BEOM_TILE_WX + 1;
        i = ch * BEOM_TILE_X + (wave % BEOM_TILE_WX) * 64 + ((int)threadIdx.x & 63) + 1;
        ipnt = i + (j - 1) * P;
        jg = j + d.joff; Mg = d.Mg; ywrap = d.yper && !d.slab;
        dv = &d;
        if (!(j <= M && i <= L && row_selected(d, j))) return false;
        return slot_is_cell(d, ipnt);             // (embedded: land slots keep the sentinel's values)
    }
    // context of an arbitrary LOCAL cell (a, b) of the same frame (used for halo cells)
    __device__ __forceinline__ void set_cell(const DevView &d, int a, int b) {
        L = d.L; M = d.M; P = d.P; xper = d.xper; yper = d.yper;
        i = a; j = b; ipnt = a + (b - 1) * P;
        jg = b + d.joff; Mg = d.Mg; ywrap = d.yper && !d.slab;
        dv = &d;
    }
    // wave-uniform: do all 64 cells of this wave satisfy 2 <= i <= L-2, 2 <= j <= M-2 ?
    __device__ __forceinline__ bool wave_is_interior() const {
        const int i0 = __builtin_amdgcn_readfirstlane(i - ((int)threadIdx.x & 63));   // first column of the wave
        if (!(i0 >= 2 && i0 + 63 <= L - 2 && j >= 2 && j <= M - 2 && jg >= 2 && jg <= Mg - 2)) return false;
        if (!dv->embedded) return true;
        // (a wave = 64 cells of one row starting at a multiple of 64 + 1: inside one 64 x 4 tile of reg4)
        return dv->reg4[(long long)((j - 1) >> 2) * dv->reg_nx + ((i0 - 1) >> 6)] != 0 && ((i0 - 1) & 63) == 0;
    }
    __device__ __forceinline__ CellDenseT<true> as_interior() const {
        CellDenseT<true> r; r.i = i; r.j = j; r.ipnt = ipnt; r.L = L; r.M = M; r.P = P; r.xper = xper; r.yper = yper;
        r.jg = jg; r.Mg = Mg; r.ywrap = ywrap; r.dv = dv;
        return r;
    }
    __device__ __forceinline__ int at(int a, int b) const {     // a, b: LOCAL target coordinates
        if (xper) { if (a == 0) a = L - 1; else if (a == L) a = 1; }
        if (ywrap) { if (b == 0) b = M - 1; else if (b == M) b = 1; }
        return (a >= 1 && a <= L && b >= 1 && b <= M) ? (a + (b - 1) * P) : 0;
    }
    template <int K> __device__ __forceinline__ int nb() const {
        constexpr int di = NbOff<K>::di, dj = NbOff<K>::dj;
        if (INTERIOR) return ipnt + di + dj * P;
        return at(i + di, j + dj);
    }
    __device__ __forceinline__ static double f(bool c) { return c ? 1.0 : 0.0; }
    // masks are functions of the GLOBAL coordinates (i, jg) on the Mg-row frame
    __device__ __forceinline__ double mk_n_ij(int a, int bg) const { return f(a >= 1 && a <= L - 1 && bg >= 1 && bg <= Mg - 1); }
    __device__ __forceinline__ double mk_n() const { return INTERIOR ? 1.0 : (dv->embedded ? dv->mk_n[ipnt] : mk_n_ij(i, jg)); }
    __device__ __forceinline__ double mk_u() const {
        return INTERIOR ? 1.0 : (dv->embedded ? dv->mk_u[ipnt] : f(jg <= Mg - 1 && i <= L - 1 && (i >= 2 || xper)));
    }
    __device__ __forceinline__ double mk_v() const {
        return INTERIOR ? 1.0 : (dv->embedded ? dv->mk_v[ipnt] : f(i <= L - 1 && jg <= Mg - 1 && (jg >= 2 || yper)));
    }
    __device__ __forceinline__ double mkpe() const {
        return INTERIOR ? 1.0 : (dv->embedded ? dv->mkpe[ipnt] : f(i <= L - 1 && jg <= Mg - 1 && (i >= 2 || xper) && (jg >= 2 || yper)));
    }
    __device__ __forceinline__ double mkpi() const { return 1.0; }          // (only packed cells are evaluated)
    template <int K> __device__ __forceinline__ double mk_n_nb(int c) const {
        if (INTERIOR) return 1.0;
        if (dv->embedded) return dv->mk_n[c];                    // (0 at the sentinel and at every land slot)
        if (c == 0) return 0.0;                                  // sentinel (also: outside a slab's window)
        int a = i + NbOff<K>::di, bg = jg + NbOff<K>::dj;
        if (xper) { if (a == 0) a = L - 1; else if (a == L) a = 1; }
        if (ywrap) { if (bg == 0) bg = Mg - 1; else if (bg == Mg) bg = 1; }
        return mk_n_ij(a, bg);
    }
    __device__ __forceinline__ int isub() const { return i; }
};
using CellDense = CellDenseT<false>;

// relaxation rate nudg(ipnt, IV) of a cell (IV = 1 eta, 2 u, 3 v): contexts that know the cell's (i, j) ask the tile table
// first (DevView::ngt) — away from the sponges a zero costs one byte per wave instead of a double per lane
template <int IV, class C>
__device__ __forceinline__ double nudg_rate(const C &c, const DevView &d) {
    if constexpr (C::kHasIJ) {
        if (d.ngt && !((d.ngt[(long long)((c.j - 1) >> 2) * d.ngt_nx + ((c.i - 1) >> 6)] >> (IV - 1)) & 1)) return 0.0;
    }
    return d.nudg[(long long)c.ipnt + d.n1 * (IV - 1)];
}
