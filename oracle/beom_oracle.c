/*
 * beom_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's time-step hot path (zhazorken/beom,
 * private_mod.f95), statement by statement and in the reference's operation order, so
 * that it is bit-identical to the flang-compiled reference at -ffp-contract=off.
 * Pinned against FP64 dumps of the real reference (oracle/ref_build.py, tests/golden).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this;
 * nothing under beom_amd/ does.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * All arrays use the Fortran storage documented in include/beom_hip.h.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "../include/beom_hip.h"

typedef struct oracle_state {
    /* static */
    const int32_t *neig, *subc;
    const double *mk_u, *mk_v, *mk_n, *mkpe, *mkpi, *fcor, *h_th, *h_to;
    const double *nudg, *fnud, *hdot, *tide, *bodf, *taus;
    /* prognostic + history */
    double *hlay, *u, *v, *h_u, *h_v, *rs_h, *dmdx, *dmdy, *v_cc, *v_ll, *tt3d, *tb3d, *tu3d;
    /* 2-D scratch of the reference, (0:ndeg) each, reused layer after layer */
    double *mont, *rvor, *pvor, *dive, *d2hx, *d2hy;
    /* optional per-layer copies of the scratch (0:ndeg, nlay) for per-layer checks; may be NULL */
    double *mont_l, *rvor_l, *pvor_l, *dive_l, *d2hx_l, *d2hy_l;
    /* nudged open-boundary segments, Fortran segm(nseg, 18) (private_mod.f95:1060-1240); may be NULL */
    const int32_t *segm;
    int64_t nseg;
    /* biharmonic viscosity work arrays (0:ndeg, nlay), needed when svis > 0 (private_mod.f95:40-43) */
    double *delu, *delv, *uu4, *vv4;
    /* rigid lid (rgld = 1): surface pressure pi_s(0:ndeg) and the operators of its Poisson equation
     * (private_mod.f95:64-67, 91, 505-563); needed when rgld > 0.5 */
    double *pi_s;
    const double *Ow, *Os, *Osum_;
} oracle_state;

#define N1 ((size_t)P->ndeg + 1)
#define L2(a, ip, il) (a)[(size_t)(ip) + N1 * (size_t)((il) - 1)]           /* X(ipnt,ilay)       */
#define NEIG(k, ip) S->neig[((k) - 1) + 8 * (size_t)(ip)]                    /* neig(k,ipnt)       */
#define H2(a, m, ip, il) (a)[((m) - 1) + 2 * ((size_t)(ip) + N1 * (size_t)((il) - 1))]
#define H3(a, m, ip, il) (a)[((m) - 1) + 3 * ((size_t)(ip) + N1 * (size_t)((il) - 1))]
#define FNUD(ip, il, iv) S->fnud[(size_t)(ip) + N1 * ((size_t)((il) - 1) + (size_t)P->nlay * ((iv) - 1))]
#define NUDG(ip, iv) S->nudg[(size_t)(ip) + N1 * ((iv) - 1)]
#define TIDE(m, ip, iv) S->tide[((m) - 1) + 2 * ((size_t)(ip) + N1 * ((iv) - 1))]
#define T3(a, ip, id, il) (a)[(size_t)(ip) + N1 * ((size_t)((id) - 1) + 2 * (size_t)((il) - 1))]
#define BODF(il, id) S->bodf[((il) - 1) + (size_t)P->nlay * ((id) - 1)]
#define TAUS(ip, id) S->taus[(size_t)(ip) + N1 * ((id) - 1)]
#define IX_N 1
#define IX_U 2
#define IX_V 3

/* x**n with n an integer CONSTANT (nsal is a parameter, shared_mod.f95:105): flang emits the
 * left-to-right product ((x*x)*x)*... — probed with flang 22 for n = 4..8; a run-time exponent
 * would go through repeated squaring instead, which differs in the last bit from n = 4 on. */
static inline double powi(double x, int n) {
    double r = x;
    for (int k = 1; k < n; ++k) r = r * x;
    return r;
}

/* ---- update_h, private_mod.f95:1593-1646 (variant 0) and
 *      private_mod3d.f95:1593-1689 (variant 1) ----------------------------------- */
void oracle_update_h(const beom_params *P, oracle_state *S, double gene, double ramp, double ctim) {
    const double i_dl = 1.0 / P->dl;                                  /* :1599 */
    const int ndeg = P->ndeg, nlay = P->nlay;
    for (int ilay = nlay; ilay >= 1; --ilay) {                        /* :1604 */
        const double vecl = (ilay == 1) ? 1.0 : 0.0;                  /* :1600-1601 */
#pragma omp parallel for schedule(static)
        for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
            const int c1 = NEIG(1, ipnt), c3 = NEIG(3, ipnt);
            double hold = L2(S->hlay, ipnt, ilay);                    /* :1610 */
            double rs_3 = (L2(S->h_u, ipnt, ilay) - L2(S->h_u, c1, ilay)) * i_dl
                        + (L2(S->h_v, ipnt, ilay) - L2(S->h_v, c3, ilay)) * i_dl
                        + (S->hdot ? L2(S->hdot, ipnt, ilay) : 0.0);  /* :1612-1620 */
            rs_3 = rs_3 * S->mk_n[ipnt];                              /* :1622 */
            double rhsi = ((1.5 + P->beta) * rs_3
                           - (0.5 + 2.0 * P->beta) * H2(S->rs_h, 2, ipnt, ilay)
                           + P->beta * H2(S->rs_h, 1, ipnt, ilay)) * P->dt * gene
                        + rs_3 * P->dt * (1.0 - gene);                /* :1624-1628 */
            hold = hold + rhsi;                                       /* :1630 */
            double hfor = FNUD(ipnt, ilay, IX_N);
            if (S->tide)
                hfor = hfor + ramp * TIDE(1, ipnt, IX_N) * vecl
                            * cos(TIDE(2, ipnt, IX_N) - P->w_ti * ctim); /* :1632-1634 */
            else
                hfor = hfor + 0.0;          /* tide == 0: the term is +0, which turns a -0 into +0 */
            const double ng = NUDG(ipnt, IX_N);
            if (P->variant == 0) {
                L2(S->hlay, ipnt, ilay) = hfor * ng + (1.0 - ng) * hold;   /* :1637-1638 */
            } else {                                                  /* private_mod3d.f95:1636-1683 */
                const double hfor1 = 0.0, hfor2 = 800.0, hfor3 = 0.0;
                const int isub = S->subc[ipnt];                       /* subc(ipnt,1) */
                const int half = P->lm / 2;
                double hl = hold;
                const double h3 = L2(S->hlay, ipnt, 3);               /* current value of layer 3 */
                const double h3v = (ilay == 3) ? hold : h3;
                if (h3v > 20.0 * P->hsal && isub > half) {
                    if (ilay == 1)
                        hl = hl + 0.0 * ng + fmax(hfor1 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 2)
                        hl = hl + 0.0 * ng + fmax(hfor2 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 3)
                        hl = hl - 0.0 * ng + fmin(hfor3 * ng + (-ng) * hl, 0.0);
                } else if (h3v < 20.0 * P->hsal && isub > half) {
                    if (ilay == 1)
                        hl = hl + 0.0 * ng + 1.0 * fmax(hfor2 * ng + (-ng) * hl, 0.0);
                    else if (ilay == 2)
                        hl = hl - 0.0 * ng + 1.0 * fmin(hfor1 * ng + (-ng) * hl, 0.0);
                }
                if (isub < half)
                    hl = hfor * ng + (1.0 - ng) * hold;
                L2(S->hlay, ipnt, ilay) = hl;
            }
            H2(S->rs_h, 1, ipnt, ilay) = H2(S->rs_h, 2, ipnt, ilay);  /* :1642 */
            H2(S->rs_h, 2, ipnt, ilay) = rs_3;                        /* :1643 */
        }
    }
}

/* ---- update_mont_rvor_pvor_dive_kine, private_mod.f95:2318-2439 ----------------- */
void oracle_update_mont(const beom_params *P, oracle_state *S, int ilay) {
    const double i_dl = 1.0 / P->dl, i_gr = 1.0 / P->grav;            /* :2321-2322 */
    const double i_ns = 1.0 / (double)(P->nsal - 1);                  /* :2328 */
    const double hs_8 = P->hsal;
    const int ndeg = P->ndeg, nlay = P->nlay, nsal = P->nsal;
    double i_rn[BEOM_MAX_LAYERS];
    for (int i = 0; i < nlay; ++i) i_rn[i] = 1.0 / P->rhon[i];        /* :2329 */
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
        const int c1 = NEIG(1, ipnt), c3 = NEIG(3, ipnt), c5 = NEIG(5, ipnt),
                  c6 = NEIG(6, ipnt), c7 = NEIG(7, ipnt);
        const double u_le = L2(S->u, ipnt, ilay), u_ri = L2(S->u, c1, ilay);
        const double v_bo = L2(S->v, ipnt, ilay), v_to = L2(S->v, c3, ilay);
        const double mkn = S->mk_n[ipnt];
        double mpot = L2(S->hlay, ipnt, ilay) + P->hmin * (1.0 - mkn);       /* :2351 */
        mpot = powi(P->hsal / mpot, nsal - 1);                               /* :2352 */
        mpot = mpot * (-P->ocrp * i_ns * P->hsal * mkn);                     /* :2353 */
        mpot = mpot - (S->h_to ? S->h_to[ipnt] : 0.0);                       /* :2356 */
        for (int i = 1; i <= ilay - 1; ++i)                                  /* :2357-2361 */
            mpot = mpot - (P->rhon[ilay - 1] - P->rhon[i - 1]) * i_rn[ilay - 1] * L2(S->hlay, ipnt, i);
        if (P->rgld < 0.5) {                                                 /* :2365-2375 */
            double hcol = 0.0;
            for (int i = 1; i <= nlay; ++i) hcol = hcol + L2(S->hlay, ipnt, i);
            mpot = hcol - S->h_th[ipnt] + mpot;
        }
        S->mont[ipnt] = mpot + 0.25 * P->uadv * i_gr
                             * (u_ri * u_ri + u_le * u_le + v_to * v_to + v_bo * v_bo); /* :2380-2383 */
        S->rvor[ipnt] = (v_bo - L2(S->v, c5, ilay) - u_le + L2(S->u, c7, ilay))
                        * i_dl * S->mkpe[ipnt];                              /* :2388-2389 */
        const double h0 = L2(S->hlay, ipnt, ilay), hE = L2(S->hlay, c1, ilay),
                     hW = L2(S->hlay, c5, ilay), hN = L2(S->hlay, c3, ilay),
                     hS = L2(S->hlay, c7, ilay);
        double d2x = (hE + hW - h0 * 2.0) * S->mk_n[c1] * S->mk_n[c5] * mkn; /* :2394-2397 */
        double d2y = (hN + hS - h0 * 2.0) * S->mk_n[c3] * S->mk_n[c7] * mkn; /* :2399-2402 */
        if (P->ocrp > 0.5) {                                                 /* :2404-2416 */
            if (hE < 2.0 * hs_8 || hW < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2x = 0.0;
            if (hN < 2.0 * hs_8 || hS < 2.0 * hs_8 || h0 < 2.0 * hs_8) d2y = 0.0;
        }
        S->d2hx[ipnt] = d2x;
        S->d2hy[ipnt] = d2y;
        const double have = h0 + hW + L2(S->hlay, c6, ilay) + hS;            /* :2421-2424 */
        S->pvor[ipnt] = (S->fcor[ipnt] + S->rvor[ipnt] * P->uadv) * S->mkpi[ipnt]
                        * (mkn + S->mk_n[c5] + S->mk_n[c6] + S->mk_n[c7]) / have; /* :2426-2433 */
        S->dive[ipnt] = (u_ri - u_le + v_to - v_bo) * i_dl;                  /* :2435-2436 */
    }
    if (S->mont_l) {
        const size_t off = N1 * (size_t)(ilay - 1);
        memcpy(S->mont_l + off, S->mont, N1 * sizeof(double));
        memcpy(S->rvor_l + off, S->rvor, N1 * sizeof(double));
        memcpy(S->pvor_l + off, S->pvor, N1 * sizeof(double));
        memcpy(S->dive_l + off, S->dive, N1 * sizeof(double));
        memcpy(S->d2hx_l + off, S->d2hx, N1 * sizeof(double));
        memcpy(S->d2hy_l + off, S->d2hy, N1 * sizeof(double));
    }
}

/* ---- update_viscosity (Leith), private_mod.f95:2441-2502.  svis>0 (biharmonic,
 *      :2508-2599) is out of scope and rejected by the drivers. ------- */
void oracle_update_viscosity(const beom_params *P, oracle_state *S, int ilay) {
    const int ndeg = P->ndeg;
    const double dl = P->dl;
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
        const int c1 = NEIG(1, ipnt), c2 = NEIG(2, ipnt), c3 = NEIG(3, ipnt),
                  c5 = NEIG(5, ipnt), c6 = NEIG(6, ipnt), c7 = NEIG(7, ipnt);
        const double r_bl = S->rvor[ipnt], r_br = S->rvor[c1], r_tr = S->rvor[c2],
                     r_tl = S->rvor[c3], rbll = S->rvor[c5], rbbl = S->rvor[c7];
        const double d_cc = S->dive[ipnt], d_ri = S->dive[c1], d_to = S->dive[c3],
                     d_le = S->dive[c5], d_bl = S->dive[c6], d_bo = S->dive[c7];
        double a = (r_br - r_bl) * (r_br - r_bl)
                 + (r_bl - rbll) * (r_bl - rbll)
                 + (r_tl - r_bl) * (r_tl - r_bl)
                 + (r_bl - rbbl) * (r_bl - rbbl)
                 + (d_cc - d_le) * (d_cc - d_le)
                 + (d_bo - d_bl) * (d_bo - d_bl)
                 + (d_cc - d_bo) * (d_cc - d_bo)
                 + (d_le - d_bl) * (d_le - d_bl);                            /* :2477-2487 */
        L2(S->v_ll, ipnt, ilay) = sqrt(a) * P->dvis * dl * dl + P->bvis;     /* :2488-2489 */
        double b = (r_br - r_bl) * (r_br - r_bl)
                 + (r_tr - r_tl) * (r_tr - r_tl)
                 + (r_tl - r_bl) * (r_tl - r_bl)
                 + (r_tr - r_br) * (r_tr - r_br)
                 + (d_ri - d_cc) * (d_ri - d_cc)
                 + (d_cc - d_le) * (d_cc - d_le)
                 + (d_to - d_cc) * (d_to - d_cc)
                 + (d_cc - d_bo) * (d_cc - d_bo);                            /* :2492-2500 */
        L2(S->v_cc, ipnt, ilay) = sqrt(b) * P->dvis * dl * dl + P->bvis;     /* :2501-2502 */
    }
    if (!(P->svis > 0.0)) return;
    /* biharmonic viscosity, thickness-weighted grad^4(u,v) (:2508-2599) */
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {                               /* :2508-2550 */
        const int c1 = NEIG(1, ipnt), c3 = NEIG(3, ipnt), c5 = NEIG(5, ipnt), c7 = NEIG(7, ipnt);
        double du = 0.0, dv = 0.0;
        if (S->mk_u[ipnt] > 0.5) {
            du = du + 1.0 / (dl * dl) * (S->mk_u[c1] * L2(S->u, c1, ilay) + S->mk_u[c3] * L2(S->u, c3, ilay)
                                         + S->mk_u[c5] * L2(S->u, c5, ilay) + S->mk_u[c7] * L2(S->u, c7, ilay));
            du = du - 1.0 / (dl * dl) * (S->mk_u[c1] + S->mk_u[c3] + S->mk_u[c5] + S->mk_u[c7]) * L2(S->u, ipnt, ilay);
        }
        if (S->mk_v[ipnt] > 0.5) {
            dv = dv + 1.0 / (dl * dl) * (S->mk_v[c1] * L2(S->v, c1, ilay) + S->mk_v[c3] * L2(S->v, c3, ilay)
                                         + S->mk_v[c5] * L2(S->v, c5, ilay) + S->mk_v[c7] * L2(S->v, c7, ilay));
            dv = dv - 1.0 / (dl * dl) * (S->mk_v[c1] + S->mk_v[c3] + S->mk_v[c5] + S->mk_v[c7]) * L2(S->v, ipnt, ilay);
        }
        L2(S->delu, ipnt, ilay) = du;
        L2(S->delv, ipnt, ilay) = dv;
    }
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {                               /* :2557-2598 */
        const int c1 = NEIG(1, ipnt), c3 = NEIG(3, ipnt), c5 = NEIG(5, ipnt), c6 = NEIG(6, ipnt), c7 = NEIG(7, ipnt);
        const double h = L2(S->hlay, ipnt, ilay);
        /* :2565 `real( ... )` has no kind argument: the sum is rounded to DEFAULT (single) real */
        const double hh_q = (double)(float)(h + S->mk_n[c5] * L2(S->hlay, c5, ilay) + S->mk_n[c6] * L2(S->hlay, c6, ilay)
                                            + S->mk_n[c7] * L2(S->hlay, c7, ilay))
                            / (1.0 + S->mk_n[c5] + S->mk_n[c6] + S->mk_n[c7]);
        double uu = 0.0, vv = 0.0;
        uu = uu - 1.0 / dl * h * L2(S->delu, ipnt, ilay) + 1.0 / dl * h * L2(S->delv, ipnt, ilay);
        vv = vv + 1.0 / dl * hh_q * L2(S->delu, ipnt, ilay) + 1.0 / dl * hh_q * L2(S->delv, ipnt, ilay);
        if (S->subc[ipnt] <= P->lm - 1) uu = uu + 1.0 / dl * h * L2(S->delu, c1, ilay);
        if (S->subc[ipnt + N1] <= P->mm - 1) uu = uu - 1.0 / dl * h * L2(S->delv, c3, ilay);
        if (S->subc[ipnt] > 1) vv = vv - 1.0 / dl * hh_q * L2(S->delv, c5, ilay);
        if (S->subc[ipnt + N1] > 1) vv = vv - 1.0 / dl * hh_q * L2(S->delu, c7, ilay);
        if (S->mk_u[ipnt] * S->mk_v[ipnt] < 0.5) vv = 0.0;
        L2(S->uu4, ipnt, ilay) = uu;
        L2(S->vv4, ipnt, ilay) = vv;
    }
}

/* ---- update_u, private_mod.f95:1422-1503 ---------------------------------------- */
void oracle_update_u(const beom_params *P, oracle_state *S, int ilay,
                     double gene, double ramp, double ctim) {
    const double i_dl = 1.0 / P->dl, i_r0 = 1.0 / P->rho0, i_r1 = 1.0 / P->rhon[0];
    const int ndeg = P->ndeg;
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
        const int c3 = NEIG(3, ipnt), c4 = NEIG(4, ipnt), c5 = NEIG(5, ipnt);
        const double mask = S->mk_u[ipnt];
        const double hcen = (L2(S->hlay, c5, ilay) + L2(S->hlay, ipnt, ilay)) / (1.0 + mask); /* :1438 */
        const double i__h = 1.0 / (hcen + 1.0 - mask);                       /* :1439 */
        double uold = L2(S->u, ipnt, ilay);
        const double dmd4 = (S->mont[c5] - S->mont[ipnt]) * i_dl * P->grav * mask; /* :1442 */
        const double tauw = 0.5 * (T3(S->tt3d, c5, 1, ilay) + T3(S->tt3d, ipnt, 1, ilay)) * ramp; /* :1444 */
        double ufor = FNUD(ipnt, ilay, IX_U)
                    + 0.5 * (T3(S->tt3d, ipnt, 2, ilay) + T3(S->tt3d, c5, 2, ilay))
                      * i_r1 * P->invf * i__h * ramp;                        /* :1450-1452 */
        if (S->tide)
            ufor = ufor + ramp * TIDE(1, ipnt, IX_U) * cos(TIDE(2, ipnt, IX_U) - P->w_ti * ctim); /* :1453 */
        else
            ufor = ufor + 0.0;
        double rhsi = dmd4 * (1.0 - gene)
                    + 0.25 * S->pvor[ipnt] * (L2(S->h_v, ipnt, ilay) + L2(S->h_v, c5, ilay))
                    + 0.25 * S->pvor[c3] * (L2(S->h_v, c3, ilay) + L2(S->h_v, c4, ilay))
                    + tauw * i_r0 * i__h
                    - T3(S->tb3d, ipnt, 1, ilay) * i_r0 * i__h
                    - T3(S->tu3d, ipnt, 1, ilay) * i_r0 * i__h
                    + (S->bodf ? BODF(ilay, 1) : 0.0)
                    + (P->del1 * dmd4
                       + P->del2 * H3(S->dmdx, 3, ipnt, ilay)
                       + P->gamm * H3(S->dmdx, 2, ipnt, ilay)
                       + P->epsi * H3(S->dmdx, 1, ipnt, ilay)) * gene;       /* :1456-1469 */
        if (P->svis > 0.0)                                                   /* :1471-1473 */
            /* `real( ... )` without kind at :1472: single-precision rounding of the difference (u only) */
            rhsi = rhsi - P->svis * i_dl * (double)(float)(L2(S->uu4, ipnt, ilay) - L2(S->uu4, c5, ilay)
                                                           + L2(S->vv4, c3, ilay) - L2(S->vv4, ipnt, ilay)) * i__h;
        else
        rhsi = rhsi + (L2(S->v_cc, ipnt, ilay) * S->dive[ipnt]
                       - L2(S->v_cc, c5, ilay) * S->dive[c5]) * i_dl
                    - (L2(S->v_ll, c3, ilay) * S->rvor[c3]
                       - L2(S->v_ll, ipnt, ilay) * S->rvor[ipnt]) * i_dl;    /* :1476-1479 */
        uold = uold + rhsi * mask * P->dt;                                   /* :1481 */
        uold = ufor * NUDG(ipnt, IX_U) + uold * (1.0 - NUDG(ipnt, IX_U));    /* :1483-1484 */
        L2(S->u, ipnt, ilay) = uold;
        if (P->rgld < 0.5)                                                   /* :1491-1496 */
            L2(S->h_u, ipnt, ilay) = 0.5 * (uold + fabs(uold)) * (hcen - 0.16667 * S->d2hx[c5])
                                   + 0.5 * (uold - fabs(uold)) * (hcen - 0.16667 * S->d2hx[ipnt]);
        H3(S->dmdx, 1, ipnt, ilay) = H3(S->dmdx, 2, ipnt, ilay);             /* :1498-1500 */
        H3(S->dmdx, 2, ipnt, ilay) = H3(S->dmdx, 3, ipnt, ilay);
        H3(S->dmdx, 3, ipnt, ilay) = dmd4;
    }
}

/* ---- update_v, private_mod.f95:1505-1591 ---------------------------------------- */
void oracle_update_v(const beom_params *P, oracle_state *S, int ilay,
                     double gene, double ramp, double ctim) {
    const double i_dl = 1.0 / P->dl, i_r0 = 1.0 / P->rho0, i_r1 = 1.0 / P->rhon[0];
    const int ndeg = P->ndeg;
#pragma omp parallel for schedule(static)
    for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
        const int c1 = NEIG(1, ipnt), c7 = NEIG(7, ipnt), c8 = NEIG(8, ipnt);
        const double mask = S->mk_v[ipnt];
        const double hcen = (L2(S->hlay, ipnt, ilay) + L2(S->hlay, c7, ilay)) / (1.0 + mask); /* :1521 */
        const double i__h = 1.0 / (hcen + 1.0 - mask);                       /* :1524 */
        double vold = L2(S->v, ipnt, ilay);
        const double dmd4 = (S->mont[c7] - S->mont[ipnt]) * i_dl * P->grav * mask; /* :1527 */
        const double tauw = 0.5 * (T3(S->tt3d, c7, 2, ilay) + T3(S->tt3d, ipnt, 2, ilay)) * ramp; /* :1529 */
        double vfor = FNUD(ipnt, ilay, IX_V)
                    - 0.5 * (T3(S->tt3d, ipnt, 1, ilay) + T3(S->tt3d, c7, 1, ilay))
                      * i_r1 * P->invf * i__h * ramp;                        /* :1535-1537 */
        if (S->tide)
            vfor = vfor + ramp * TIDE(1, ipnt, IX_V) * cos(TIDE(2, ipnt, IX_V) - P->w_ti * ctim); /* :1538 */
        else
            vfor = vfor + 0.0;
        double rhsi = dmd4 * (1.0 - gene)
                    - 0.25 * S->pvor[ipnt] * (L2(S->h_u, ipnt, ilay) + L2(S->h_u, c7, ilay))
                    - 0.25 * S->pvor[c1] * (L2(S->h_u, c1, ilay) + L2(S->h_u, c8, ilay))
                    + tauw * i_r0 * i__h
                    - T3(S->tb3d, ipnt, 2, ilay) * i_r0 * i__h
                    - T3(S->tu3d, ipnt, 2, ilay) * i_r0 * i__h
                    + (S->bodf ? BODF(ilay, 2) : 0.0)
                    + (P->del1 * dmd4
                       + P->del2 * H3(S->dmdy, 3, ipnt, ilay)
                       + P->gamm * H3(S->dmdy, 2, ipnt, ilay)
                       + P->epsi * H3(S->dmdy, 1, ipnt, ilay)) * gene;       /* :1541-1554 */
        if (P->svis > 0.0)                                                   /* :1555-1557 */
            rhsi = rhsi - P->svis * i_dl * (L2(S->vv4, c1, ilay) - L2(S->vv4, ipnt, ilay)
                                            - L2(S->uu4, ipnt, ilay) + L2(S->uu4, c7, ilay)) * i__h;
        else
        rhsi = rhsi + (L2(S->v_cc, ipnt, ilay) * S->dive[ipnt]
                       - L2(S->v_cc, c7, ilay) * S->dive[c7]) * i_dl
                    + (L2(S->v_ll, c1, ilay) * S->rvor[c1]
                       - L2(S->v_ll, ipnt, ilay) * S->rvor[ipnt]) * i_dl;    /* :1561-1564 */
        vold = vold + rhsi * mask * P->dt;                                   /* :1567 */
        vold = vfor * NUDG(ipnt, IX_V) + vold * (1.0 - NUDG(ipnt, IX_V));    /* :1569-1570 */
        L2(S->v, ipnt, ilay) = vold;
        if (P->rgld < 0.5)                                                   /* :1577-1582 */
            L2(S->h_v, ipnt, ilay) = 0.5 * (vold + fabs(vold)) * (hcen - 0.16667 * S->d2hy[c7])
                                   + 0.5 * (vold - fabs(vold)) * (hcen - 0.16667 * S->d2hy[ipnt]);
        H3(S->dmdy, 1, ipnt, ilay) = H3(S->dmdy, 2, ipnt, ilay);             /* :1584-1586 */
        H3(S->dmdy, 2, ipnt, ilay) = H3(S->dmdy, 3, ipnt, ilay);
        H3(S->dmdy, 3, ipnt, ilay) = dmd4;
    }
}

/* ---- first_three_timesteps prologue, private_mod.f95:2166-2177 ------------------ */
void oracle_rebuild_fluxes(const beom_params *P, oracle_state *S) {
    const int ndeg = P->ndeg, nlay = P->nlay;
    for (int ilay = 1; ilay <= nlay; ++ilay) {
#pragma omp parallel for schedule(static)
        for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
            const int c5 = NEIG(5, ipnt), c7 = NEIG(7, ipnt);
            L2(S->h_u, ipnt, ilay) = L2(S->u, ipnt, ilay)
                * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, c5, ilay)) / (1.0 + S->mk_u[ipnt]);
            L2(S->h_v, ipnt, ilay) = L2(S->v, ipnt, ilay)
                * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, c7, ilay)) / (1.0 + S->mk_v[ipnt]);
        }
    }
}

/* ---- distribute_stress, private_mod.f95:1921-2149 ------------------------------- */
static int any_taus(const beom_params *P, const oracle_state *S) {
    if (!S->taus) return 0;
    for (size_t i = 0; i < 2 * N1; ++i)
        if (fabs(S->taus[i]) > 1.e-7) return 1;                              /* :1945 */
    return 0;
}

void oracle_distribute_stress(const beom_params *P, oracle_state *S) {
    const int ndeg = P->ndeg, nlay = P->nlay;
    const double hs_8 = P->hsal;
    const int wind = any_taus(P, S);
    const int bot = P->bdrg > 1.e-7, top = P->tdrg > 1.e-7;
    if (!wind && !bot && !top) return;
    double *layt = (double *)calloc(N1 * nlay, sizeof(double));
    double *layb = (double *)calloc(N1 * nlay, sizeof(double));
    double *layu = (double *)calloc(N1 * nlay, sizeof(double));
    double *taub = (double *)calloc(N1 * 2, sizeof(double));
    double *taum = (double *)calloc(N1 * 2, sizeof(double));
#define LY(a, ip, il) (a)[(size_t)(ip) + N1 * (size_t)((il) - 1)]
    if (wind && P->ocrp > 0.5) {                                             /* :1945-1959 */
        for (int ilay = 1; ilay <= nlay; ++ilay)
            for (int ipnt = 0; ipnt <= ndeg; ++ipnt) {
                LY(layt, ipnt, ilay) = 0.0;
                double hcum = 0.0, sofar = 0.0;
                for (int k = 1; k <= ilay; ++k) sofar = sofar + LY(layt, ipnt, k);
                for (int k = 1; k <= ilay; ++k)
                    hcum = hcum + fmax(0.0, L2(S->hlay, ipnt, k) - 1.5 * P->hsal);
                double t = fmin(hcum, P->hsbl) / P->hsbl - sofar;
                LY(layt, ipnt, ilay) = fmax(t, 0.0);
            }
    } else if (wind) {                                                       /* :1960-1967 */
        for (int ipnt = 0; ipnt <= ndeg; ++ipnt) LY(layt, ipnt, 1) = 1.0;
    }
    if (bot && P->ocrp > 0.5) {                                              /* :1969-1980 */
        for (int ilay = nlay; ilay >= 1; --ilay)
            for (int ipnt = 0; ipnt <= ndeg; ++ipnt) {
                LY(layb, ipnt, ilay) = 0.0;
                double sofar = 0.0, hcum = 0.0;
                for (int k = ilay; k <= nlay; ++k) sofar = sofar + LY(layb, ipnt, k);
                for (int k = ilay; k <= nlay; ++k) hcum = hcum + L2(S->hlay, ipnt, k);
                double t = fmin(hcum, P->hbbl) / P->hbbl - sofar;
                LY(layb, ipnt, ilay) = fmax(t, 0.0);
            }
    } else if (bot) {                                                        /* :1981-1988 */
        for (int ipnt = 0; ipnt <= ndeg; ++ipnt) LY(layb, ipnt, nlay) = 1.0;
    }
    if (top && P->ocrp > 0.5) {                                              /* :1991-2005 */
        for (int ilay = 1; ilay <= nlay; ++ilay)
            for (int ipnt = 0; ipnt <= ndeg; ++ipnt) {
                LY(layu, ipnt, ilay) = 0.0;
                double hcum = 0.0, sofar = 0.0;
                for (int k = 1; k <= ilay; ++k) sofar = sofar + LY(layu, ipnt, k);
                for (int k = 1; k <= ilay; ++k)
                    hcum = hcum + fmax(0.0, L2(S->hlay, ipnt, k) - 1.5 * P->hsal);
                double t = fmin(hcum, P->hsbl) / P->hsbl - sofar;
                LY(layu, ipnt, ilay) = fmax(t, 0.0);
            }
    } else if (top) {                                                        /* :2006-2013 */
        for (int ipnt = 0; ipnt <= ndeg; ++ipnt) LY(layu, ipnt, 1) = 1.0;
    }
    for (int pass = 0; pass < 2; ++pass) {       /* 0: bottom (:2015-2072), 1: top (:2075-2134) */
        if (pass == 0 && !bot) continue;
        if (pass == 1 && !top) continue;
        const double drg = pass == 0 ? P->bdrg : P->tdrg;
        double *tau = pass == 0 ? taub : taum;
        const double *lay = pass == 0 ? layb : layu;
        double *t3 = pass == 0 ? S->tb3d : S->tu3d;
        for (int ipnt = 0; ipnt <= ndeg; ++ipnt) {
            int ilay = pass == 0 ? nlay : 1;
            if (P->ocrp > 0.5) {
                if (pass == 0) { for (int k = nlay; k >= 1; --k) if (L2(S->hlay, ipnt, k) > 2.0 * hs_8) { ilay = k; break; } }
                else           { for (int k = 1; k <= nlay; ++k) if (L2(S->hlay, ipnt, k) > 2.0 * hs_8) { ilay = k; break; } }
            }
            const int c1 = NEIG(1, ipnt), c3 = NEIG(3, ipnt), c4 = NEIG(4, ipnt),
                      c5 = NEIG(5, ipnt), c7 = NEIG(7, ipnt), c8 = NEIG(8, ipnt);
            const double vatu = 0.25 * L2(S->v, ipnt, ilay) + 0.25 * L2(S->v, c3, ilay)
                              + 0.25 * L2(S->v, c4, ilay) + 0.25 * L2(S->v, c5, ilay);
            const double uatv = 0.25 * L2(S->u, ipnt, ilay) + 0.25 * L2(S->u, c1, ilay)
                              + 0.25 * L2(S->u, c7, ilay) + 0.25 * L2(S->u, c8, ilay);
            const double uu = L2(S->u, ipnt, ilay), vv = L2(S->v, ipnt, ilay);
            const double rh = P->rhon[ilay - 1];
            tau[ipnt]      = uu * drg * rh * (P->qdrg * sqrt(uu * uu + vatu * vatu) + 1.0 - P->qdrg);
            tau[ipnt + N1] = vv * drg * rh * (P->qdrg * sqrt(vv * vv + uatv * uatv) + 1.0 - P->qdrg);
        }
        for (int ilay = 1; ilay <= nlay; ++ilay)
            for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
                const int c5 = NEIG(5, ipnt), c7 = NEIG(7, ipnt);
                T3(t3, ipnt, 1, ilay) = tau[ipnt] * 0.5 * (LY(lay, ipnt, ilay) + LY(lay, c5, ilay));
                T3(t3, ipnt, 2, ilay) = tau[ipnt + N1] * 0.5 * (LY(lay, ipnt, ilay) + LY(lay, c7, ilay));
            }
    }
    if (wind)                                                                /* :2136-2146 */
        for (int ilay = 1; ilay <= nlay; ++ilay)
            for (int ipnt = 1; ipnt <= ndeg; ++ipnt) {
                T3(S->tt3d, ipnt, 1, ilay) = TAUS(ipnt, 1) * LY(layt, ipnt, ilay);
                T3(S->tt3d, ipnt, 2, ilay) = TAUS(ipnt, 2) * LY(layt, ipnt, ilay);
            }
    free(layt); free(layb); free(layu); free(taub); free(taum);
#undef LY
}

/* ---- no_gradient_obc, private_mod.f95:2613-2679 (only when flag_nudging and mcbc < 0.5) ---- */
#define SEGM(is, col) S->segm[(size_t)(is) + (size_t)S->nseg * ((col) - 1)]
void oracle_no_gradient_obc(const beom_params *P, oracle_state *S, int ilay) {
    for (int64_t is = 0; is < S->nseg; ++is) {                               /* :2624-2651 */
        const int ipnt = SEGM(is, 10), in = SEGM(is, 16);
        if (SEGM(is, 5) == 1) {
            if (S->mk_u[ipnt] > 0.5) {
                L2(S->u, ipnt, ilay) = L2(S->u, in, ilay) - FNUD(in, ilay, IX_U) + FNUD(ipnt, ilay, IX_U);
                L2(S->h_u, ipnt, ilay) = L2(S->u, ipnt, ilay)
                    * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, NEIG(5, ipnt), ilay)) / (1.0 + S->mk_u[ipnt]);
            }
        } else if (SEGM(is, 4) == 1) {
            if (S->mk_v[ipnt] > 0.5) {
                L2(S->v, ipnt, ilay) = L2(S->v, in, ilay) - FNUD(in, ilay, IX_V) + FNUD(ipnt, ilay, IX_V);
                L2(S->h_v, ipnt, ilay) = L2(S->v, ipnt, ilay)
                    * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, NEIG(7, ipnt), ilay)) / (1.0 + S->mk_v[ipnt]);
            }
        }
    }
    for (int64_t is = 0; is < S->nseg; ++is) {                               /* :2657-2678 */
        const int ipnt = SEGM(is, 1), in = SEGM(is, 13);
        if (SEGM(is, 5) == 1) {
            L2(S->v, ipnt, ilay) = L2(S->v, in, ilay) - FNUD(in, ilay, IX_V) + FNUD(ipnt, ilay, IX_V);
            L2(S->h_v, ipnt, ilay) = L2(S->v, ipnt, ilay)
                * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, NEIG(7, ipnt), ilay)) / (1.0 + S->mk_v[ipnt]);
        } else if (SEGM(is, 4) == 1) {
            L2(S->u, ipnt, ilay) = L2(S->u, in, ilay) - FNUD(in, ilay, IX_U) + FNUD(ipnt, ilay, IX_U);
            L2(S->h_u, ipnt, ilay) = L2(S->u, ipnt, ilay)
                * (L2(S->hlay, ipnt, ilay) + L2(S->hlay, NEIG(5, ipnt), ilay)) / (1.0 + S->mk_u[ipnt]);
        }
    }
}

/* ---- rigid lid (rgld = 1), the fork's addition ------------------------------------------------ */
/* epilogue of update_h, private_mod.f95:1648-1700: the two upper layers absorb the column's misfit.
 * `real(x)` without a kind is DEFAULT real and so is the literal 0.5: the correction is rounded to
 * real*4 before it is subtracted; the sum is taken again after layer 1 has changed. */
void oracle_rgld_h_epilogue(const beom_params *P, oracle_state *S) {
    const int l2 = P->nlay < 2 ? P->nlay : 2;            /* (hlay(ipnt, 2); oracle patch P2 for nlay = 1) */
    for (int ipnt = 1; ipnt <= P->ndeg; ++ipnt) {
        for (int pass = 1; pass <= 2; ++pass) {
            double s = L2(S->hlay, ipnt, 1);
            for (int i = 2; i <= P->nlay; ++i) s = s + L2(S->hlay, ipnt, i);
            const float corr = 0.5f * (float)(s - S->h_th[ipnt]);
            const int il = pass == 1 ? 1 : l2;
            L2(S->hlay, ipnt, il) = L2(S->hlay, ipnt, il) - (double)corr;
        }
    }
}

/* h_u, h_v from the 3rd-order upstream-biased form with the 2-D d2hx, d2hy as the LAST Montgomery
 * sweep left them (private_mod.f95:2237-2257, 2292-2314: all layers use the last layer's curvature) */
void oracle_rgld_upstream_fluxes(const beom_params *P, oracle_state *S) {
    for (int ilay = 1; ilay <= P->nlay; ++ilay)
        for (int ipnt = 1; ipnt <= P->ndeg; ++ipnt) {
            const int c5 = NEIG(5, ipnt), c7 = NEIG(7, ipnt);
            double mask = S->mk_u[ipnt];
            double hcen = (L2(S->hlay, c5, ilay) + L2(S->hlay, ipnt, ilay)) / (1.0 + mask);
            const double uu = L2(S->u, ipnt, ilay);
            L2(S->h_u, ipnt, ilay) = 0.5 * (uu + fabs(uu)) * (hcen - 0.16667 * S->d2hx[c5])
                                   + 0.5 * (uu - fabs(uu)) * (hcen - 0.16667 * S->d2hx[ipnt]);
            mask = S->mk_v[ipnt];
            hcen = (L2(S->hlay, c7, ilay) + L2(S->hlay, ipnt, ilay)) / (1.0 + mask);
            const double vv = L2(S->v, ipnt, ilay);
            L2(S->h_v, ipnt, ilay) = 0.5 * (vv + fabs(vv)) * (hcen - 0.16667 * S->d2hy[c7])
                                   + 0.5 * (vv - fabs(vv)) * (hcen - 0.16667 * S->d2hy[ipnt]);
        }
}

/* surf_pressure, private_mod.f95:1705-1838: Poisson equation for the lid pressure by Gauss-Seidel
 * sweeps in packed order (rp = 1), then the velocity correction */
void oracle_surf_pressure(const beom_params *P, oracle_state *S) {
    const int nd = P->ndeg, lm = P->lm, mm = P->mm;
    const size_t n1 = N1;
    double *rhs = (double *)calloc(n1, sizeof(double)), *prev = (double *)calloc(n1, sizeof(double));
    const double rp = 1.000, pi_tol = 1.e-5, dl = P->dl, dt = P->dt;
    const int maxiters = 1000;
    for (int ilay = P->nlay; ilay >= 1; --ilay) {
        for (int ipnt = 1; ipnt <= nd; ++ipnt)
            if (S->subc[ipnt] > 1) {                                                     /* :1727-1736 */
                const int c5 = NEIG(5, ipnt);
                rhs[ipnt] = rhs[ipnt] - L2(S->h_u, ipnt, ilay) / (dl * dt);
                rhs[c5] = rhs[c5] + L2(S->h_u, ipnt, ilay) / (dl * dt);
            }
        for (int ipnt = 1; ipnt <= nd; ++ipnt)
            if (S->subc[ipnt + n1] > 1) {                                                /* :1740-1752 */
                const int c7 = NEIG(7, ipnt);
                rhs[ipnt] = rhs[ipnt] - L2(S->h_v, ipnt, ilay) / (dl * dt);
                rhs[c7] = rhs[c7] + L2(S->h_v, ipnt, ilay) / (dl * dt);
            }
    }
    double maxdiff = pi_tol + 1;
    int iters = 0;
    while (maxdiff > pi_tol && iters < maxiters) {                                       /* :1759-1802 */
        maxdiff = 0;
        for (int ipnt = 1; ipnt <= nd; ++ipnt) {
            const int i = S->subc[ipnt], j = S->subc[ipnt + n1];
            prev[ipnt] = S->pi_s[ipnt];
            double x = (1 - rp) * S->pi_s[ipnt] - rp * S->Osum_[ipnt] * rhs[ipnt];
            if (i < lm) { const int c1 = NEIG(1, ipnt); x = x + rp * S->Osum_[ipnt] * S->Ow[c1] * S->pi_s[c1]; }
            if (j < mm) { const int c3 = NEIG(3, ipnt); x = x + rp * S->Osum_[ipnt] * S->Os[c3] * S->pi_s[c3]; }
            if (i > 1) { const int c5 = NEIG(5, ipnt); x = x + rp * S->Osum_[ipnt] * S->Ow[ipnt] * S->pi_s[c5]; }
            if (j > 1) { const int c7 = NEIG(7, ipnt); x = x + rp * S->Osum_[ipnt] * S->Os[ipnt] * S->pi_s[c7]; }
            S->pi_s[ipnt] = x;
        }
        for (int ipnt = 1; ipnt <= nd; ++ipnt) {
            const double diff = fabs(S->pi_s[ipnt] - prev[ipnt]);
            if (diff > maxdiff) maxdiff = diff;
        }
        iters = iters + 1;
    }
    for (int ilay = 1; ilay <= P->nlay; ++ilay)                                          /* :1806-1819 */
        for (int ipnt = 1; ipnt <= nd; ++ipnt)
            if (S->subc[ipnt] > 1 && S->subc[ipnt] < lm + 1) {
                const int c5 = NEIG(5, ipnt);
                L2(S->u, ipnt, ilay) = L2(S->u, ipnt, ilay) - dt / dl * S->pi_s[ipnt];
                L2(S->u, ipnt, ilay) = L2(S->u, ipnt, ilay) + dt / dl * S->pi_s[c5];
            }
    for (int ilay = 1; ilay <= P->nlay; ++ilay)                                          /* :1823-1833 */
        for (int ipnt = 1; ipnt <= nd; ++ipnt)
            if (S->subc[ipnt + n1] > 1 && S->subc[ipnt + n1] < mm + 1) {
                const int c7 = NEIG(7, ipnt);
                L2(S->v, ipnt, ilay) = L2(S->v, ipnt, ilay) - dt / dl * S->pi_s[ipnt];
                L2(S->v, ipnt, ilay) = L2(S->v, ipnt, ilay) + dt / dl * S->pi_s[c7];
            }
    free(rhs); free(prev);
}

/* ---- one time step: first_three_timesteps (:2151-2223) / gener_forward_backward
 *      (:2225-2316) ------------------------------------------------------------- */
static void oracle_one_step(const beom_params *P, oracle_state *S, int tstp, int first3,
                            int upst, double gene, double ramp, double ctim) {
    const int rgld = P->rgld > 0.5;
    if (first3) oracle_rebuild_fluxes(P, S);
    else if (rgld) oracle_rgld_upstream_fluxes(P, S);                        /* :2237-2257 */
    oracle_update_h(P, S, gene, ramp, ctim);
    if (rgld) oracle_rgld_h_epilogue(P, S);                                  /* :1648-1700 */
    for (int ilay = 1; ilay <= P->nlay; ++ilay) {
        oracle_update_mont(P, S, ilay);
        if (first3 || (P->dvis > 1.e-3 && upst) || P->svis > 0)              /* :2188,2268 */
            oracle_update_viscosity(P, S, ilay);
        if (tstp % 2 == 0) {                                                 /* :2193,2276 */
            oracle_update_u(P, S, ilay, gene, ramp, ctim);
            oracle_update_v(P, S, ilay, gene, ramp, ctim);
        } else {
            oracle_update_v(P, S, ilay, gene, ramp, ctim);
            oracle_update_u(P, S, ilay, gene, ramp, ctim);
        }
        if (P->flag_nudging && P->mcbc < 0.5 && S->segm)                    /* :2201-2204,2285-2288 */
            oracle_no_gradient_obc(P, S, ilay);
    }
    if (rgld) {                                                              /* :2207-2221, 2292-2314 */
        if (first3) oracle_rebuild_fluxes(P, S);
        else oracle_rgld_upstream_fluxes(P, S);
        oracle_surf_pressure(P, S);
    }
}

/* ---- integrate_time, private_mod.f95:1853-1912, for steps tstp_first.. ---------- */
int oracle_step(const beom_params *P, oracle_state *S, int tstp_first, int nsteps,
                double tres, double dtd8, double dt_r, double rsta, int n_3d) {
    if (P->rgld > 0.5 && !(S->pi_s && S->Ow && S->Os && S->Osum_)) return -1;
    if (P->svis > 0 && !S->uu4) return -1;
    if (P->flag_nudging && P->mcbc < 0.5 && !S->segm) return -2;
    for (int tstp = tstp_first; tstp < tstp_first + nsteps; ++tstp) {
        const double ctim = tres + dtd8 * (double)tstp;                      /* :1862,1887 */
        double ramp = 1.0;
        const int first3 = tstp <= 3;
        int upst = 0;
        if (tstp == 1) {
            oracle_distribute_stress(P, S);                                  /* :1863 */
            upst = 1;
        } else if (!first3) {
            upst = (tstp % n_3d) == 0;                                       /* :1889-1892 */
            if (upst) oracle_distribute_stress(P, S);                        /* :1894-1896 */
        }
        if (tstp == 1 || !first3) {
            if (rsta < 0.5 && ctim < dt_r) ramp = ctim / dt_r;               /* :1864-1866,1898-1901 */
        } else {
            /* tstp 2,3: ramp keeps the value set at tstp 1 (:1858-1875) */
            const double c1 = tres + dtd8 * 1.0;
            if (rsta < 0.5 && c1 < dt_r) ramp = c1 / dt_r;
        }
        const double gene = (first3 || (P->g_fb > 0.5 && P->rgld > 0.5)) ? 0.0 : P->g_fb;       /* :1859,1877; :1880-1884: no multistep with a lid */
        oracle_one_step(P, S, tstp, first3, upst, gene, ramp, ctim);
    }
    return 0;
}

int oracle_sizeof_params(void) { return (int)sizeof(beom_params); }
int oracle_sizeof_state(void) { return (int)sizeof(oracle_state); }
