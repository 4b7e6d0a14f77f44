// beom_multi.hip — one process, several GPUs: the j-slab decomposition of SURVEY.md §8(e) behind
// the same C-ABI shape as the single-device handle, for hosts that stay single-process (the
// Fortran host under main.f95; SURVEY §8(b) "Threading").  The reference has no counterpart
// (OpenMP only).  Built only from the public entry points of include/beom_hip.h:
//   * the frame (dense: ndeg = (lm+1)(mm+1)) is cut into bands of rows; every band becomes an
//     ordinary slab handle (beom_params.slab_row0/slab_mm) on its own device, with G = 4 ghost
//     rows per neighbour — the scheme of beom_amd/slab.py, which drives one process per GPU;
//   * one exchange per time step of hlay,u,v,h_u,h_v: beom_pack_rows on the owner, a peer copy
//     over xGMI (hipMemcpyPeerAsync) on the receiver's second stream, beom_unpack_rows there;
//   * the exchange of step n overlaps phase 1 of step n+1 (beom_step_phase), as in slab.py.
// No CPU fallback: every call needs its HIP devices.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/beom_hip.h"

namespace {

constexpr int kGhost = 4;          // rows per neighbour; see beom_amd/slab.py for why 4 is enough
constexpr int kFields = 5;         // hlay, u, v, h_u, h_v

void m_err(char *errm, int len, const char *fmt, ...) {
    if (!errm || len <= 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errm, (size_t)len, fmt, ap);
    va_end(ap);
}

#define M_HIP(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            m_err(errm, errm_len, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                       \
            return -100 - (int)e_;                                                           \
        }                                                                                    \
    } while (0)
#define M_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

struct Band {
    int own0, own1, win0, win1;    // global rows (1-based, inclusive): owned band and local window
    int L;                         // columns = lm + 1
    long long a, b;                // global packed range [a, b) of the window
    int rows() const { return win1 - win0 + 1; }
    long long n_loc() const { return b - a; }
    int loc(int grow) const { return grow - win0 + 1; }          // global row -> local row
};

// [outer][0:n1g][inner]  ->  [outer][0:n_loc][inner]: sentinel first, then the window's cells
template <class T>
std::vector<T> cut(const T *x, size_t outer, size_t inner, size_t n1g, const Band &s) {
    std::vector<T> z;
    if (!x) return z;
    const size_t n1l = (size_t)s.n_loc() + 1;
    z.resize(outer * n1l * inner);
    for (size_t o = 0; o < outer; ++o) {
        std::memcpy(&z[o * n1l * inner], &x[o * n1g * inner], inner * sizeof(T));
        std::memcpy(&z[(o * n1l + 1) * inner], &x[(o * n1g + (size_t)s.a) * inner], (size_t)s.n_loc() * inner * sizeof(T));
    }
    return z;
}
// owned rows of a local [outer][0:n_loc][inner] array back into the global one (+ sentinel from band 0)
template <class T>
void paste(T *glob, const std::vector<T> &loc, size_t outer, size_t inner, size_t n1g, const Band &s, bool sentinel) {
    if (!glob) return;
    const size_t n1l = (size_t)s.n_loc() + 1;
    const size_t la = 1 + (size_t)(s.own0 - s.win0) * s.L, lb = 1 + (size_t)(s.own1 - s.win0 + 1) * s.L;
    const size_t ga = 1 + (size_t)(s.own0 - 1) * s.L;
    for (size_t o = 0; o < outer; ++o) {
        if (sentinel) std::memcpy(&glob[o * n1g * inner], &loc[o * n1l * inner], inner * sizeof(T));
        std::memcpy(&glob[(o * n1g + ga) * inner], &loc[(o * n1l + la) * inner], (lb - la) * inner * sizeof(T));
    }
}
template <class T> const T *ptr(const std::vector<T> &v) { return v.empty() ? nullptr : v.data(); }
template <class T> T *ptr(std::vector<T> &v) { return v.empty() ? nullptr : v.data(); }

}  // namespace

struct beom_multi {
    beom_params P{};               // global frame
    int n = 0;
    size_t n1g = 0;
    std::vector<int> dev;
    std::vector<Band> band;
    std::vector<beom_handle> eng;
    std::vector<hipStream_t> main_s, comm_s;
    std::vector<hipEvent_t> packed, landed;
    std::vector<char> pending;     // an exchange into this band is in flight
    std::vector<double *> send_s, recv_s, send_n, recv_n;   // device buffers on the band's device
    size_t xbytes = 0;
    long long n_split = 0, n_plain = 0;   // band-steps taken in two phases / in one piece
};

namespace {

void destroy_all(beom_multi *M) {
    if (!M) return;
    for (int k = 0; k < M->n; ++k) {
        (void)hipSetDevice(M->dev[k]);
        if (k < (int)M->main_s.size() && M->main_s[k]) (void)hipStreamSynchronize(M->main_s[k]);
        if (k < (int)M->comm_s.size() && M->comm_s[k]) (void)hipStreamSynchronize(M->comm_s[k]);
    }
    for (int k = 0; k < M->n; ++k) {
        (void)hipSetDevice(M->dev[k]);
        if (k < (int)M->eng.size() && M->eng[k]) (void)beom_destroy(M->eng[k]);
        for (auto *v : {&M->send_s, &M->recv_s, &M->send_n, &M->recv_n})
            if (k < (int)v->size() && (*v)[k]) (void)hipFree((*v)[k]);
        if (k < (int)M->packed.size() && M->packed[k]) (void)hipEventDestroy(M->packed[k]);
        if (k < (int)M->landed.size() && M->landed[k]) (void)hipEventDestroy(M->landed[k]);
        if (k < (int)M->comm_s.size() && M->comm_s[k]) (void)hipStreamDestroy(M->comm_s[k]);
        if (k < (int)M->main_s.size() && M->main_s[k]) (void)hipStreamDestroy(M->main_s[k]);
    }
    delete M;
}

}  // namespace

extern "C" {

int beom_multi_create(const beom_params *prm, int ndev, const int *devices,
                      const int32_t *neig, const int32_t *subc,
                      const double *mk_u, const double *mk_v, const double *mk_n,
                      const double *mkpe, const double *mkpi,
                      const double *fcor, const double *h_th, const double *h_to,
                      const double *nudg, const double *fnud, const double *hdot,
                      const double *tide, const double *bodf, const double *taus,
                      beom_multi_handle *out, char *errm, int errm_len) {
    if (!prm || !out || !devices || !neig || !subc) { m_err(errm, errm_len, "beom_multi_create: null argument"); return -1; }
    *out = nullptr;
    if (prm->abi_version != BEOM_ABI_VERSION) { m_err(errm, errm_len, "beom_multi_create: ABI version mismatch"); return -2; }
    const int L = prm->lm + 1, Mg = prm->mm + 1, nl = prm->nlay;
    if (ndev < 1 || ndev > 64) { m_err(errm, errm_len, "beom_multi_create: bad device count %d", ndev); return -3; }
    if ((long long)prm->ndeg != (long long)L * Mg || prm->slab_mm != 0) {
        m_err(errm, errm_len, "beom_multi_create: the row decomposition needs a whole dense frame (ndeg = (lm+1)(mm+1))");
        return -3;
    }
    if (Mg < ndev * (kGhost + 1)) { m_err(errm, errm_len, "beom_multi_create: %d rows are too few for %d bands", Mg, ndev); return -3; }
    if (prm->flag_nudging && prm->mcbc < 0.5) {
        m_err(errm, errm_len, "beom_multi_create: mcbc = 0 (no_gradient_obc) runs on a single-device handle only");
        return -4;
    }
    if (ndev > 1) {   // a frame periodic in y would need the exchange to close the ring (not offered)
        const int32_t *nb1 = neig + 8ll * 1;               // cell (1,1): S neighbour is slot 7
        if (nb1[6] != 0) { m_err(errm, errm_len, "beom_multi_create: frames periodic in y run on one device only"); return -4; }
    }
    beom_multi *M = new beom_multi();
    M->P = *prm; M->n = ndev; M->n1g = (size_t)prm->ndeg + 1;
    M->dev.assign(devices, devices + ndev);
    if (getenv("BEOM_MULTI_WRAP_DEVICES")) {     // rehearsals: more bands than GPUs, ids taken modulo the visible count
        int nvis = 0;
        if (hipGetDeviceCount(&nvis) == hipSuccess && nvis > 0)
            for (int &dv : M->dev) dv %= nvis;
    }
    M->eng.assign(ndev, nullptr);
    M->main_s.assign(ndev, nullptr); M->comm_s.assign(ndev, nullptr);
    M->packed.assign(ndev, nullptr); M->landed.assign(ndev, nullptr);
    M->pending.assign(ndev, 0);
    M->send_s.assign(ndev, nullptr); M->recv_s.assign(ndev, nullptr);
    M->send_n.assign(ndev, nullptr); M->recv_n.assign(ndev, nullptr);
    M->xbytes = (size_t)kFields * nl * kGhost * L * sizeof(double);
    // equal row counts, remainders to the first bands (dense frames: equal work)
    const int base = Mg / ndev, rem = Mg % ndev;
    int j = 1;
    for (int k = 0; k < ndev; ++k) {
        Band s{};
        const int cnt = base + (k < rem ? 1 : 0);
        s.own0 = j; s.own1 = j + cnt - 1; j += cnt;
        s.win0 = k > 0 ? s.own0 - kGhost : s.own0;
        s.win1 = k < ndev - 1 ? s.own1 + kGhost : s.own1;
        s.L = L;
        s.a = 1 + (long long)(s.win0 - 1) * L; s.b = 1 + (long long)s.win1 * L;
        M->band.push_back(s);
    }
    const size_t n1g = M->n1g;
    for (int k = 0; k < ndev; ++k) {
        const Band &s = M->band[k];
        beom_params lp = *prm;
        lp.mm = s.rows() - 1; lp.ndeg = (int32_t)s.n_loc();
        lp.slab_row0 = ndev > 1 ? s.win0 - 1 : 0; lp.slab_mm = ndev > 1 ? prm->mm : 0;
        lp.dense_hint = 1;
        std::vector<int32_t> nb = cut(neig, 1, 8, n1g, s);
        for (int32_t &g : nb) g = (g >= s.a && g < s.b) ? (int32_t)(g - s.a + 1) : 0;   // local index, 0 outside the window
        auto sc = cut(subc, 2, 1, n1g, s);
        auto a_mk_u = cut(mk_u, 1, 1, n1g, s), a_mk_v = cut(mk_v, 1, 1, n1g, s), a_mk_n = cut(mk_n, 1, 1, n1g, s);
        auto a_mkpe = cut(mkpe, 1, 1, n1g, s), a_mkpi = cut(mkpi, 1, 1, n1g, s);
        auto a_fcor = cut(fcor, 1, 1, n1g, s), a_h_th = cut(h_th, 1, 1, n1g, s), a_h_to = cut(h_to, 1, 1, n1g, s);
        auto a_nudg = cut(nudg, 3, 1, n1g, s), a_fnud = cut(fnud, (size_t)3 * nl, 1, n1g, s), a_hdot = cut(hdot, nl, 1, n1g, s);
        auto a_tide = cut(tide, 3, 2, n1g, s), a_taus = cut(taus, 2, 1, n1g, s);
        hipError_t he = hipSetDevice(M->dev[k]);
        if (he != hipSuccess) { m_err(errm, errm_len, "hipSetDevice(%d): %s", M->dev[k], hipGetErrorString(he)); destroy_all(M); return -100 - (int)he; }
        int rc = beom_create(&lp, M->dev[k], nb.data(), sc.data(), ptr(a_mk_u), ptr(a_mk_v), ptr(a_mk_n), ptr(a_mkpe),
                             ptr(a_mkpi), ptr(a_fcor), ptr(a_h_th), ptr(a_h_to), ptr(a_nudg), ptr(a_fnud), ptr(a_hdot),
                             ptr(a_tide), bodf, ptr(a_taus), &M->eng[k], errm, errm_len);
        if (rc) { destroy_all(M); return rc; }
        if (ndev > 1 && !beom_is_dense(M->eng[k])) {
            m_err(errm, errm_len, "beom_multi_create: band %d did not qualify for the dense path", k);
            destroy_all(M); return -4;
        }
#define M_TRY_D(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { m_err(errm, errm_len, "%s failed: %s", #expr, hipGetErrorString(e_)); destroy_all(M); return -100 - (int)e_; } } while (0)
        M_TRY_D(hipStreamCreateWithFlags(&M->main_s[k], hipStreamNonBlocking));
        M_TRY_D(hipStreamCreateWithFlags(&M->comm_s[k], hipStreamNonBlocking));
        M_TRY_D(hipEventCreateWithFlags(&M->packed[k], hipEventDisableTiming));
        M_TRY_D(hipEventCreateWithFlags(&M->landed[k], hipEventDisableTiming));
        if (k > 0) { M_TRY_D(hipMalloc((void **)&M->send_s[k], M->xbytes)); M_TRY_D(hipMalloc((void **)&M->recv_s[k], M->xbytes)); }
        if (k < ndev - 1) { M_TRY_D(hipMalloc((void **)&M->send_n[k], M->xbytes)); M_TRY_D(hipMalloc((void **)&M->recv_n[k], M->xbytes)); }
        (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0);
        // direct peer copies over xGMI where the devices allow it (already-enabled is fine)
        for (int nbk : {k - 1, k + 1}) {
            if (nbk < 0 || nbk >= ndev || M->dev[nbk] == M->dev[k]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, M->dev[k], M->dev[nbk]) == hipSuccess && can) {
                hipError_t pe = hipDeviceEnablePeerAccess(M->dev[nbk], 0);
                if (pe != hipSuccess) (void)hipGetLastError();
            }
        }
#undef M_TRY_D
    }
    *out = M;
    return 0;
}

int beom_multi_destroy(beom_multi_handle M) { destroy_all(M); return 0; }

int beom_multi_count(beom_multi_handle M) { return M ? M->n : -1; }

int beom_multi_band(beom_multi_handle M, int k, int *own0, int *own1, int *win0, int *win1, int *device) {
    if (!M || k < 0 || k >= M->n) return -3;
    if (own0) *own0 = M->band[k].own0;
    if (own1) *own1 = M->band[k].own1;
    if (win0) *win0 = M->band[k].win0;
    if (win1) *win1 = M->band[k].win1;
    if (device) *device = M->dev[k];
    return 0;
}

int beom_multi_stats(beom_multi_handle M, long long *split_band_steps, long long *plain_band_steps) {
    if (!M) return -1;
    if (split_band_steps) *split_band_steps = M->n_split;
    if (plain_band_steps) *plain_band_steps = M->n_plain;
    return 0;
}

int beom_multi_sync(beom_multi_handle M, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    for (int k = 0; k < M->n; ++k) {
        M_HIP(hipSetDevice(M->dev[k]));
        M_HIP(hipStreamSynchronize(M->main_s[k]));
        M_HIP(hipStreamSynchronize(M->comm_s[k]));
        M->pending[k] = 0;          // whatever was in flight has landed
    }
    return 0;
}

int beom_multi_upload_state(beom_multi_handle M, const double *hlay, const double *u, const double *v,
                            const double *h_u, const double *h_v, const double *rs_h, const double *dmdx,
                            const double *dmdy, const double *v_cc, const double *v_ll, const double *tt3d,
                            const double *tb3d, const double *tu3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    const size_t nl = (size_t)M->P.nlay, n1g = M->n1g;
    for (int k = 0; k < M->n; ++k) {
        const Band &s = M->band[k];
        auto a0 = cut(hlay, nl, 1, n1g, s), a1 = cut(u, nl, 1, n1g, s), a2 = cut(v, nl, 1, n1g, s);
        auto a3 = cut(h_u, nl, 1, n1g, s), a4 = cut(h_v, nl, 1, n1g, s);
        auto a5 = cut(rs_h, nl, 2, n1g, s), a6 = cut(dmdx, nl, 3, n1g, s), a7 = cut(dmdy, nl, 3, n1g, s);
        auto a8 = cut(v_cc, nl, 1, n1g, s), a9 = cut(v_ll, nl, 1, n1g, s);
        auto b0 = cut(tt3d, 2 * nl, 1, n1g, s), b1 = cut(tb3d, 2 * nl, 1, n1g, s), b2 = cut(tu3d, 2 * nl, 1, n1g, s);
        M_RC(beom_upload_state(M->eng[k], ptr(a0), ptr(a1), ptr(a2), ptr(a3), ptr(a4), ptr(a5), ptr(a6), ptr(a7),
                               ptr(a8), ptr(a9), ptr(b0), ptr(b1), ptr(b2), errm, errm_len));
    }
    return 0;
}

int beom_multi_download_state(beom_multi_handle M, double *hlay, double *u, double *v, double *h_u, double *h_v,
                              double *rs_h, double *dmdx, double *dmdy, double *v_cc, double *v_ll,
                              double *tt3d, double *tb3d, double *tu3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    const size_t nl = (size_t)M->P.nlay, n1g = M->n1g;
    for (int k = 0; k < M->n; ++k) {
        const Band &s = M->band[k];
        const size_t n1l = (size_t)s.n_loc() + 1;
        auto buf = [&](double *want, size_t per) { return std::vector<double>(want ? per * n1l : 0); };
        auto a0 = buf(hlay, nl), a1 = buf(u, nl), a2 = buf(v, nl), a3 = buf(h_u, nl), a4 = buf(h_v, nl);
        auto a5 = buf(rs_h, 2 * nl), a6 = buf(dmdx, 3 * nl), a7 = buf(dmdy, 3 * nl), a8 = buf(v_cc, nl), a9 = buf(v_ll, nl);
        auto b0 = buf(tt3d, 2 * nl), b1 = buf(tb3d, 2 * nl), b2 = buf(tu3d, 2 * nl);
        M_RC(beom_download_state(M->eng[k], ptr(a0), ptr(a1), ptr(a2), ptr(a3), ptr(a4), ptr(a5), ptr(a6), ptr(a7),
                                 ptr(a8), ptr(a9), ptr(b0), ptr(b1), ptr(b2), errm, errm_len));
        const bool s0 = k == 0;
        paste(hlay, a0, nl, 1, n1g, s, s0); paste(u, a1, nl, 1, n1g, s, s0); paste(v, a2, nl, 1, n1g, s, s0);
        paste(h_u, a3, nl, 1, n1g, s, s0); paste(h_v, a4, nl, 1, n1g, s, s0);
        paste(rs_h, a5, nl, 2, n1g, s, s0); paste(dmdx, a6, nl, 3, n1g, s, s0); paste(dmdy, a7, nl, 3, n1g, s, s0);
        paste(v_cc, a8, nl, 1, n1g, s, s0); paste(v_ll, a9, nl, 1, n1g, s, s0);
        paste(tt3d, b0, 2 * nl, 1, n1g, s, s0); paste(tb3d, b1, 2 * nl, 1, n1g, s, s0); paste(tu3d, b2, 2 * nl, 1, n1g, s, s0);
    }
    return 0;
}

int beom_multi_step(beom_multi_handle M, int tstp_first, int nsteps, double tres, double dtd8, double dt_r,
                    double rsta, int n_3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (tstp_first < 1 || nsteps < 0 || n_3d < 1) { m_err(errm, errm_len, "beom_multi_step: bad arguments"); return -3; }
    const int n = M->n;
    if (n == 1) return beom_step(M->eng[0], tstp_first, nsteps, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len);
    std::vector<char> split(n);
    for (int t = tstp_first; t < tstp_first + nsteps; ++t) {
        // phase 1: the rows that cannot depend on the ghosts still in flight
        for (int k = 0; k < n; ++k) {
            split[k] = 0;
            if (!M->pending[k]) continue;
            const int rc = beom_step_phase(M->eng[k], t, tres, dtd8, dt_r, rsta, n_3d, 1, errm, errm_len);
            if (rc == 0) split[k] = 1;
            else if (rc != -20) return rc;
        }
        // ghosts of the previous step have landed (mine: before I read them; my neighbours':
        // before I overwrite the send buffers they copy from) -> the rest of the step, then pack
        for (int k = 0; k < n; ++k) {
            M_HIP(hipSetDevice(M->dev[k]));
            for (int q : {k - 1, k, k + 1})
                if (q >= 0 && q < n && M->pending[q]) M_HIP(hipStreamWaitEvent(M->main_s[k], M->landed[q], 0));
        }
        for (int k = 0; k < n; ++k) M->pending[k] = 0;
        for (int k = 0; k < n; ++k) {
            const Band &s = M->band[k];
            if (split[k]) { M_RC(beom_step_phase(M->eng[k], t, tres, dtd8, dt_r, rsta, n_3d, 2, errm, errm_len)); ++M->n_split; }
            else { M_RC(beom_step(M->eng[k], t, 1, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len)); ++M->n_plain; }
            // what I send = my outermost OWNED rows
            if (k > 0 && beom_pack_rows(M->eng[k], s.loc(s.own0), kGhost, M->send_s[k])) { m_err(errm, errm_len, "beom_pack_rows failed"); return -3; }
            if (k < n - 1 && beom_pack_rows(M->eng[k], s.loc(s.own1 - kGhost + 1), kGhost, M->send_n[k])) { m_err(errm, errm_len, "beom_pack_rows failed"); return -3; }
            M_HIP(hipSetDevice(M->dev[k]));
            M_HIP(hipEventRecord(M->packed[k], M->main_s[k]));
        }
        // exchange on the receivers' second streams: peer copy + unpack into the ghost rows
        for (int k = 0; k < n; ++k) {
            const Band &s = M->band[k];
            M_HIP(hipSetDevice(M->dev[k]));
            M_HIP(hipStreamWaitEvent(M->comm_s[k], M->packed[k], 0));      // my own step has read its ghosts
            (void)beom_set_stream(M->eng[k], (void *)M->comm_s[k], 0);
            if (k > 0) {
                M_HIP(hipStreamWaitEvent(M->comm_s[k], M->packed[k - 1], 0));
                M_HIP(hipMemcpyPeerAsync(M->recv_s[k], M->dev[k], M->send_n[k - 1], M->dev[k - 1], M->xbytes, M->comm_s[k]));
                if (beom_unpack_rows(M->eng[k], s.loc(s.win0), kGhost, M->recv_s[k])) { m_err(errm, errm_len, "beom_unpack_rows failed"); return -3; }
            }
            if (k < n - 1) {
                M_HIP(hipStreamWaitEvent(M->comm_s[k], M->packed[k + 1], 0));
                M_HIP(hipMemcpyPeerAsync(M->recv_n[k], M->dev[k], M->send_s[k + 1], M->dev[k + 1], M->xbytes, M->comm_s[k]));
                if (beom_unpack_rows(M->eng[k], s.loc(s.own1 + 1), kGhost, M->recv_n[k])) { m_err(errm, errm_len, "beom_unpack_rows failed"); return -3; }
            }
            (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0);
            M_HIP(hipEventRecord(M->landed[k], M->comm_s[k]));
            M->pending[k] = 1;
        }
    }
    return 0;
}

}  // extern "C"
