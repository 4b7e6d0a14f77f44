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
    long long n1;                 // ndeg + 1
    int L, M, xper, yper;         // dense closed form: L = lm+1, M = mm+1
    // static
    const int32_t *neig, *subc;
    const double *mk_u, *mk_v, *mk_n, *mkpe, *mkpi, *fcor, *h_th, *h_to;
    const double *nudg, *fnud, *hdot, *tide, *bodf, *taus;
    // prognostic
    double *hlay, *u, *v, *h_u, *h_v;
    double *rs[2];                // rs[0] = rs_h(1,..) older, rs[1] = rs_h(2,..) newer
    double *dmx[3], *dmy[3];      // [0] oldest .. [2] newest
    double *v_cc, *v_ll, *tt3d, *tb3d, *tu3d;
    // per-layer diagnostics (0:ndeg, nlay)
    double *mont, *rvor, *pvor, *dive, *d2hx, *d2hy;
    // stress work arrays
    double *layt, *layb, *layu, *taub, *taum;
    // constants by value
    double dl, dt, grav, rho0, beta, epsi, gamm, del1, del2, hmin, hsal, bvis, dvis, bdrg, tdrg,
        qdrg, hsbl, hbbl, uadv, ocrp, rgld, invf, w_ti;
    double rhon[BEOM_MAX_LAYERS];
    // which optional terms are live (wave-uniform branches)
    int has_hdot, has_tide, has_bodf, has_nudg, has_stress, has_wind, has_hto;
};

// ---- neighbour lookup -------------------------------------------------------------
// Slots follow private_mod.f95:28-30: 1=E 2=NE 3=N 4=NW 5=W 6=SW 7=S 8=SE.
struct NbGather {
    const int32_t *row;
    __device__ __forceinline__ NbGather(const DevView &d, int ipnt) : row(d.neig + 8ll * ipnt) {}
    template <int K> __device__ __forceinline__ int get() const { return row[K - 1]; }
};

// Closed form for a frame whose interior is entirely wet (every BASELINE config):
// neighbour (di,dj) of (i,j) is indc0(wx(i+di), wy(j+dj)), the periodic wraps of
// index_grid_points (private_mod.f95:614-685) acting on the TARGET coordinate.
// Verified cell by cell against the caller's neig table in beom_create.
struct NbDense {
    int i, j, ipnt, L, M, xper, yper;
    __device__ __forceinline__ NbDense(const DevView &d, int ip)
        : ipnt(ip), L(d.L), M(d.M), xper(d.xper), yper(d.yper) {
        unsigned q = (unsigned)(ip - 1) / (unsigned)d.L;
        j = (int)q + 1;
        i = ip - (int)q * d.L;            // 1..L
    }
    __device__ __forceinline__ int at(int a, int b) const {
        // a in 0..L+1, b in 0..M+1
        if (xper) { if (a == 0) a = L - 1; else if (a == L) a = 1; }
        if (yper) { if (b == 0) b = M - 1; else if (b == M) b = 1; }
        return (a >= 1 && a <= L && b >= 1 && b <= M) ? (a + (b - 1) * L) : 0;
    }
    template <int K> __device__ __forceinline__ int get() const {
        constexpr int di = (K == 1 || K == 2 || K == 8) ? 1 : ((K == 4 || K == 5 || K == 6) ? -1 : 0);
        constexpr int dj = (K == 2 || K == 3 || K == 4) ? 1 : ((K == 6 || K == 7 || K == 8) ? -1 : 0);
        return at(i + di, j + dj);
    }
};
