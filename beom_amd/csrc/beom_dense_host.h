// beom_dense_host.h — host-side closed forms of the dense frame: the
// connectivity table and the five masks index_grid_points builds (private_mod.f95:567-764) for a
// frame whose interior is entirely wet, generalised to a window of rows of a taller frame
// (a j-slab).  Used to VERIFY a caller's tables (beom_create) and to GENERATE the tables of the
// row bands the multi-GPU driver cuts (beom_multi.hip).
#pragma once
#include <cstdint>
#include <vector>

namespace beom_dense {

struct HostNb {   // host twin of CellDenseT::at (beom_dev.h): wraps act on the TARGET coordinate
    int L, M, xper, ywrap;
    int at(int a, int b) const {
        if (xper) { if (a == 0) a = L - 1; else if (a == L) a = 1; }
        if (ywrap) { if (b == 0) b = M - 1; else if (b == M) b = 1; }
        return (a >= 1 && a <= L && b >= 1 && b <= M) ? (a + (b - 1) * L) : 0;
    }
};

// slots 1..8 = E NE N NW W SW S SE (private_mod.f95:28-30)
static const int kDi[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int kDj[8] = {0, 1, 1, 1, 0, -1, -1, -1};

// mask predicates of private_mod.f95:701-714 (+ :621,627,649,655,676 when periodic) at column i
// of global row jg in a frame of Mg rows
struct Masks { double n, u, v, pe, pi; };
inline Masks masks_at(int i, int jg, int L, int Mg, int xper, int yper) {
    const bool in = i <= L - 1 && jg <= Mg - 1;
    Masks m;
    m.n = in ? 1.0 : 0.0;
    m.u = (in && (i >= 2 || xper)) ? 1.0 : 0.0;
    m.v = (in && (jg >= 2 || yper)) ? 1.0 : 0.0;
    m.pe = (in && (i >= 2 || xper) && (jg >= 2 || yper)) ? 1.0 : 0.0;
    m.pi = 1.0;
    return m;
}

// Local rows 1..M are global rows joff+1..joff+M of an Mg-row frame.  Neighbours are local
// indices (0 outside the window); masks and subc(:,2) are those of the global frame.  A slab
// never wraps in y by itself (a y-periodic frame cut into bands gets its wrap from the exchange).
inline bool verify(int L, int M, int joff, int Mg, int slab, int xper, int yper, long long ndeg,
                   const int32_t *neig, const int32_t *subc, const double *mk_u, const double *mk_v,
                   const double *mk_n, const double *mkpe, const double *mkpi) {
    if (ndeg != (long long)L * M) return false;
    const HostNb nb{L, M, xper, (yper && !slab) ? 1 : 0};
    const long long n1 = ndeg + 1;
    for (int j = 1; j <= M; ++j) {
        const int jg = j + joff;
        for (int i = 1; i <= L; ++i) {
            const long long ip = i + (long long)(j - 1) * L;
            if (subc[ip] != i || subc[ip + n1] != jg) return false;
            for (int k = 0; k < 8; ++k)
                if (neig[k + 8 * ip] != nb.at(i + kDi[k], j + kDj[k])) return false;
            const Masks m = masks_at(i, jg, L, Mg, xper, yper);
            if (mk_n[ip] != m.n || mk_u[ip] != m.u || mk_v[ip] != m.v || mkpe[ip] != m.pe || mkpi[ip] != m.pi) return false;
        }
    }
    return true;
}

struct Tables {
    std::vector<int32_t> neig, subc;
    std::vector<double> mk_u, mk_v, mk_n, mkpe, mkpi;
};
// the tables verify() accepts, sentinel entries (index 0) zero
inline Tables generate(int L, int M, int joff, int Mg, int slab, int xper, int yper) {
    Tables t;
    const size_t n1 = (size_t)L * M + 1;
    t.neig.assign(8 * n1, 0); t.subc.assign(2 * n1, 0);
    t.mk_u.assign(n1, 0.0); t.mk_v.assign(n1, 0.0); t.mk_n.assign(n1, 0.0); t.mkpe.assign(n1, 0.0); t.mkpi.assign(n1, 0.0);
    const HostNb nb{L, M, xper, (yper && !slab) ? 1 : 0};
    for (int j = 1; j <= M; ++j) {
        const int jg = j + joff;
        for (int i = 1; i <= L; ++i) {
            const size_t ip = (size_t)i + (size_t)(j - 1) * L;
            t.subc[ip] = i; t.subc[ip + n1] = jg;
            for (int k = 0; k < 8; ++k) t.neig[k + 8 * ip] = nb.at(i + kDi[k], j + kDj[k]);
            const Masks m = masks_at(i, jg, L, Mg, xper, yper);
            t.mk_n[ip] = m.n; t.mk_u[ip] = m.u; t.mk_v[ip] = m.v; t.mkpe[ip] = m.pe; t.mkpi[ip] = m.pi;
        }
    }
    return t;
}

}  // namespace beom_dense
