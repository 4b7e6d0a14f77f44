// beom_engine.hip — C-ABI of include/beom_hip.h: device state, uploads/downloads, the
// step driver (integrate_time / first_three_timesteps / gener_forward_backward of the
// reference, private_mod.f95:1840-1919, 2151-2316) and kernel launches.
// No CPU fallback exists: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "beom_dev.h"
// the kernels, once per tile geometry (beom_launch_tiled.h): t8 also serves every kernel that is not tiled
namespace t8 {
#include "beom_kernels.h"
}
#undef MV_Q
#undef UV_Q
#define MV_Q 1
#define UV_Q 1
namespace t4 {
#include "beom_kernels.h"
}
using namespace t8;
#include "beom_dense_host.h"

namespace {

void set_err(char *errm, int len, const char *fmt, ...) {
    if (!errm || len <= 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errm, (size_t)len, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err(errm, errm_len, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                    \
            return -100 - (int)e_;                                                          \
        }                                                                                   \
    } while (0)

}  // namespace

constexpr int kLidBatch = 256;  // rigid lid: at most this many Gauss-Seidel sweeps in flight at once (lid_solve)

struct StepTimer {        // optional HIP-event bracket around each kernel class
    std::vector<hipEvent_t> ev; std::vector<int> cls;
    hipStream_t st;
    int stride = 1;       // only the steps with tstp % stride == 0 are bracketed (an event pair costs a few us of pipeline bubble)
    // rotate: a sampled step brackets ONE kind of sweep only — update_h | Montgomery, viscosity | momentum — by turns
    // ((tstp / stride) % 3), so no bracketed launch has another bracket's bubble in front of it
    bool rotate = false;
    int slot = -1;        // the kind this step brackets (-1: all)
    bool open = false;
    static int kind(int c) { return c == 0 ? 0 : (c == 1 || c == 2 || c == 5) ? 1 : 2; }
    void step(int tstp) { slot = rotate ? (tstp / stride) % 3 : -1; }
    void begin(int c) {
        open = slot < 0 || kind(c) == slot;
        if (!open) return;
        hipEvent_t a; (void)hipEventCreate(&a); (void)hipEventRecord(a, st); ev.push_back(a); cls.push_back(c);
    }
    void end() { if (!open) return; hipEvent_t b; (void)hipEventCreate(&b); (void)hipEventRecord(b, st); ev.push_back(b); open = false; }
};

struct beom_engine {
    beom_params P;
    int device = 0;
    hipStream_t stream = nullptr;      // stream in use (own_stream unless beom_set_stream)
    hipStream_t own_stream = nullptr;
    StepTimer *timer = nullptr;        // non-null between beom_profile_start/stop
    DevView d{};
    bool dense = false;
    std::vector<void *> allocs;
    int32_t *slot_of_dev = nullptr;    // embedded frames: packed index -> slot (device copy; null otherwise)
    bool embedded = false;
    void *stage = nullptr;             // device image of ONE caller-layout slice [0:ndeg] (<= 32 B per cell) for uploads / downloads
    size_t stage_bytes = 0;
    // geometry of the launches
    dim3 grid_cells0, grid_cells_layers_flat;
    bool wind = false, bot = false, top = false;
    // distribute_stress inside the fused momentum sweep (option "fold_stress", default on): possible when the fractions are
    // constants (ocrp = 0) and nothing the sweep would have to read was uploaded into an array the engine does not refresh
    bool fold_stress = true, fold_static_ok = false;
    bool up_tt = false, up_tb = false, up_tu = false;   // a non-zero tt3d / tb3d / tu3d has been uploaded
    bool last_folded = false;          // the last step formed its stress inside the momentum sweep (beom_info "stress_folded")
    float *h0r4_dev = nullptr, *out4[3] = {nullptr, nullptr, nullptr};   // device-side output staging
    float *diag4[3] = {nullptr, nullptr, nullptr};
    double *scan_dev = nullptr;
    int any_u = 0, any_v = 0;
    long long uniform_waves = 0, total_waves = 0;   // table path: runs of 64 cells handled by offset arithmetic
    bool obc = false;                  // no_gradient_obc active (flag_nudging, mcbc < 0.5, segments set)
    bool obc_set = false;              // beom_set_open_boundaries has been called (a band may hold no segment at all)
    bool fuse = true;                  // dense frames: Montgomery+Leith in one sweep (k_mont_visc)
    bool fuse_uv = true;               // dense frames: update_u + update_v in one sweep (k_uv_fused)
    bool lean_visc = true;             // zero viscosity (dvis = bvis = 0, v_cc = v_ll = +0): fused pair drops the viscous products
    bool visc_all_zero = true;         // no non-(+0) v_cc / v_ll has been uploaded
    bool lean_d2h = true;              // fused pair: d2hx, d2hy re-derived from hlay in k_uv_fused, not stored by k_mont_visc
    // rigid lid (rgld = 1): the caller's subc and the packed -> device index map, kept for beom_set_rigid_lid
    std::vector<int32_t> subc_host, neig_host, dev_index;
    bool lid = false, lid_ready = false;
    // the lid's Gauss-Seidel pipeline (k_rgld_gs_front): time between two sweeps, ring of pressure copies, per-sweep max |change|
    int lid_dstep = 2, lid_nring = 0, lid_maxwidth = 1;
    double *lid_ring = nullptr;
    unsigned long long *lid_maxd = nullptr;
    long long lid_sweeps = 0, lid_solves = 0, lid_launches = 0;    // statistics (beom_info)
    int lid_last = 0;                  // sweeps the last solve kept
    int profile_stride = 1;            // option "profile_stride"
    bool profile_rotate = false;       // option "profile_rotate"
    bool split_prod = false;           // split steps: part 1's Montgomery sweep left the viscous products for parts 2 and 3
    bool tile4 = false;                // the tiled sweeps run the 64 x 4 geometry (frames of one or two rounds of workgroups)
    char last_err[512] = {0};
};

namespace {

// Device arrays of doubles start 15 elements into their allocation so that cell index 1 is 128-byte
// aligned (hipMalloc aligns to 256 B); see DevView::P.
template <class T>
int dev_alloc(beom_engine *E, T **p, size_t n, char *errm, int errm_len, bool zero = true) {
    void *q = nullptr;
    const size_t lead = sizeof(T) == 8 ? 15 : 0;
    HIP_TRY(hipMalloc(&q, (n + lead + 1) * sizeof(T)));
    E->allocs.push_back(q);
    if (zero) HIP_TRY(hipMemsetAsync(q, 0, (n + lead + 1) * sizeof(T), E->stream));
    *p = (T *)q + lead;
    return 0;
}

// one slice [0:ndeg][inner] (x K interleaved levels, level m) of a caller array <-> its device array
template <class T, bool REMAP = false>
int slice_to_device(beom_engine *E, T *dev, const T *host, int inner, int K, int m, char *errm, int errm_len) {
    const DevView &d = E->d;
    const size_t n = ((size_t)d.ndeg + 1) * inner * K;
    if (n * sizeof(T) > E->stage_bytes) { set_err(errm, errm_len, "internal: staging buffer too small"); return -11; }
    if (host) HIP_TRY(hipMemcpyAsync(E->stage, host, n * sizeof(T), hipMemcpyHostToDevice, E->stream));   // nullptr: the image is there already
    const long long work = ((long long)d.ndeg + 1) * inner;
    if (E->embedded) {                 // land slots: the sentinel's value of this slice
        const long long all = (d.ncell + 1) * inner;
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3((unsigned)((all + BEOM_BLOCK - 1) / BEOM_BLOCK)), dim3(BEOM_BLOCK), 0, E->stream,
                           dev, (const T *)E->stage, d.ncell + 1, inner, K, m);
    }
    hipLaunchKernelGGL((k_repack<T, true, REMAP>), dim3((unsigned)((work + BEOM_BLOCK - 1) / BEOM_BLOCK)), dim3(BEOM_BLOCK), 0, E->stream,
                       dev, (T *)E->stage, (long long)d.ndeg, d.L, d.P ? d.P : d.L, inner, K, m, (const int32_t *)E->slot_of_dev);
    return 0;
}
template <class T>
int slice_to_host(beom_engine *E, const T *dev, T *host, int inner, int K, char *errm, int errm_len) {   // all K levels: dev[m]
    (void)dev;
    const DevView &d = E->d;
    const size_t n = ((size_t)d.ndeg + 1) * inner * K;
    HIP_TRY(hipMemcpyAsync(host, E->stage, n * sizeof(T), hipMemcpyDeviceToHost, E->stream));
    HIP_TRY(hipStreamSynchronize(E->stream));
    return 0;
}
template <class T>
void slice_gather(beom_engine *E, const T *dev, int inner, int K, int m) {        // device array -> staging image (level m of K)
    const DevView &d = E->d;
    const long long work = ((long long)d.ndeg + 1) * inner;
    hipLaunchKernelGGL((k_repack<T, false, false>), dim3((unsigned)((work + BEOM_BLOCK - 1) / BEOM_BLOCK)), dim3(BEOM_BLOCK), 0, E->stream,
                       const_cast<T *>(dev), (T *)E->stage, (long long)d.ndeg, d.L, d.P ? d.P : d.L, inner, K, m,
                       (const int32_t *)E->slot_of_dev);
}

// a static array [outer][0:ndeg][inner] of the caller -> a new device array; src == nullptr: zeros
template <class T, bool REMAP = false>
int dev_upload(beom_engine *E, const T **dst, const T *src, size_t outer, int inner, char *errm, int errm_len) {
    const DevView &d = E->d;
    T *q = nullptr;
    int rc = dev_alloc(E, &q, outer * (size_t)d.n1 * inner, errm, errm_len, true);
    if (rc) return rc;
    const size_t n1h = (size_t)d.ndeg + 1;
    if (src)
        for (size_t o = 0; o < outer; ++o)
            if ((rc = slice_to_device<T, REMAP>(E, q + o * (size_t)d.n1 * inner, src + o * n1h * inner, inner, 1, 0, errm, errm_len))) return rc;
    *dst = q;
    return 0;
}

bool any_nonzero(const double *a, size_t n) {
    if (!a) return false;
    for (size_t i = 0; i < n; ++i)
        if (a[i] != 0.0) return true;
    return false;
}

}  // namespace

extern "C" {

int beom_abi_version(void) { return BEOM_ABI_VERSION; }

int beom_device_count(char *errm, int errm_len) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    return n;
}

int beom_device_pci_bus_id(int device, char *out, int out_len) {
    if (!out || out_len < 16) return -1;
    return hipDeviceGetPCIBusId(out, out_len, device) == hipSuccess ? 0 : -9;
}

int beom_create(const beom_params *prm, int device, const int32_t *neig, const int32_t *subc,
                const double *mk_u, const double *mk_v, const double *mk_n, const double *mkpe,
                const double *mkpi, const double *fcor, const double *h_th, const double *h_to,
                const double *nudg, const double *fnud, const double *hdot, const double *tide,
                const double *bodf, const double *taus, beom_handle *out, char *errm, int errm_len) {
    if (!prm || !out) { set_err(errm, errm_len, "beom_create: null argument"); return -1; }
    if (prm->abi_version != BEOM_ABI_VERSION) { set_err(errm, errm_len, "beom_create: ABI version mismatch (%d vs %d)", prm->abi_version, BEOM_ABI_VERSION); return -2; }
    if (prm->nlay < 1 || prm->nlay > BEOM_MAX_LAYERS || prm->ndeg < 1 || prm->lm < 1 || prm->mm < 1) { set_err(errm, errm_len, "beom_create: bad sizes"); return -3; }
    if (prm->rgld > 0.5 && (prm->variant == 1 || prm->slab_mm > 0 || prm->ocrp < 0.5)) {
        // (private_mod3d.f95 has no lid; the Poisson operators are only initialised with ocrp = 1, :505-563; the pressure sweep
        // couples the whole frame, so no bands)
        set_err(errm, errm_len, "beom_create: rgld = 1 (rigid lid, private_mod.f95:1705-1838) needs variant 0, ocrp = 1 and a whole frame (no bands)");
        return -5;
    }
    if (prm->variant == 1 && prm->nlay < 3) { set_err(errm, errm_len, "beom_create: variant 1 (private_mod3d.f95) needs nlay >= 3"); return -7; }
    if (!neig || !subc || !mk_u || !mk_v || !mk_n || !mkpe || !mkpi || !fcor || !h_th || !nudg || !fnud) { set_err(errm, errm_len, "beom_create: null static array"); return -1; }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device < 0 || device >= ndev) { set_err(errm, errm_len, "beom_create: no usable HIP device (%d visible, asked for %d); there is no CPU fallback", ndev, device); return -8; }
    HIP_TRY(hipSetDevice(device));
    beom_engine *E = new beom_engine();
    E->P = *prm;
    E->device = device;
    // from here on every failure releases the handle (its stream and device arrays) on the way out
#define HIP_TRY_E(expr)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err(errm, errm_len, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                    \
            beom_destroy(E);                                                                \
            return -100 - (int)e_;                                                          \
        }                                                                                   \
    } while (0)
    HIP_TRY_E(hipStreamCreateWithFlags(&E->own_stream, hipStreamNonBlocking));
    E->stream = E->own_stream;
    DevView &d = E->d;
    const size_t n1h = (size_t)prm->ndeg + 1, nl = (size_t)prm->nlay;      // n1h: cells per layer in the caller's arrays
    d.ndeg = prm->ndeg; d.nlay = prm->nlay; d.lm = prm->lm; d.mm = prm->mm; d.nsal = prm->nsal;
    d.variant = prm->variant; d.n1 = (long long)n1h;
    d.L = prm->lm + 1; d.M = prm->mm + 1;
    d.dl = prm->dl; d.dt = prm->dt; d.grav = prm->grav; d.rho0 = prm->rho0; d.beta = prm->beta;
    d.epsi = prm->epsi; d.gamm = prm->gamm; d.del1 = prm->del1; d.del2 = prm->del2; d.hmin = prm->hmin;
    d.hsal = prm->hsal; d.bvis = prm->bvis; d.dvis = prm->dvis; d.bdrg = prm->bdrg; d.tdrg = prm->tdrg;
    d.qdrg = prm->qdrg; d.hsbl = prm->hsbl; d.hbbl = prm->hbbl; d.uadv = prm->uadv; d.ocrp = prm->ocrp;
    d.rgld = prm->rgld; d.invf = prm->invf; d.w_ti = prm->w_ti; d.svis = prm->svis;
    d.mm_glob = prm->slab_mm > 0 ? prm->slab_mm : prm->mm;
    for (int i = 0; i < BEOM_MAX_LAYERS; ++i) d.rhon[i] = prm->rhon[i];
    d.i_dl = 1.0 / prm->dl;                      // private_mod.f95:1428,1511,1599,2321
    d.i_gr = 1.0 / prm->grav;                    // :2322
    d.i_ns = 1.0 / (double)(prm->nsal - 1);      // :2328
    d.i_r0 = 1.0 / prm->rho0;                    // :1429
    d.i_r1 = 1.0 / prm->rhon[0];                 // :1430
    for (int i = 0; i < prm->nlay; ++i) d.i_rn[i] = 1.0 / prm->rhon[i];   // :2329
    // periodicity is encoded only in neig (private_mod.f95:614-685); recover it for the dense form
    d.xper = 0; d.yper = 0;
    d.nstrip = 1; d.jlo0 = 1; d.jhi0 = d.M; d.jlo1 = 1; d.jhi1 = 0;
    d.slab = prm->slab_mm > 0 ? 1 : 0;
    d.joff = d.slab ? prm->slab_row0 : 0;
    d.Mg = d.slab ? prm->slab_mm + 1 : d.M;
    if (d.slab && (d.joff < 0 || d.joff + d.M > d.Mg)) { set_err(errm, errm_len, "beom_create: slab rows outside the global frame"); beom_destroy(E); return -3; }
    E->dense = false;
    if (prm->dense_hint && (long long)prm->ndeg == (long long)d.L * d.M) {
        for (int xp = 0; xp < 2 && !E->dense; ++xp)
            for (int yp = 0; yp < 2 && !E->dense; ++yp) {
                if (d.slab && yp) continue;      // a slab of a y-periodic frame gets its wrap from the exchange, not from neig
                if (beom_dense::verify(d.L, d.M, d.joff, d.Mg, d.slab, xp, yp, prm->ndeg, neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi)) {
                    E->dense = true; d.xper = xp; d.yper = yp;
                }
            }
    }
    // Frames with land: the same rectangle, every packed cell in the slot of its (i, j) (subc), if the
    // caller's connectivity is what offsets on the rectangle give — wraps as for a dense frame, and wherever the table says
    // 0 the offset lands outside the rectangle or on a slot that is no packed cell (it then holds the sentinel's values).
    std::vector<int32_t> slot_of, pk_of;
    std::vector<unsigned char> reg4;
    if (!E->dense && prm->dense_hint && prm->svis == 0.0 && getenv("BEOM_NO_EMBED") == nullptr &&      // (a band of a frame with land too: d.slab)
        (long long)d.L * d.M < 2000000000ll && (long long)prm->ndeg * 10 >= (long long)d.L * d.M * 3) {     // (at least 30 % of the rectangle in use)
        const int L = d.L, M = d.M, P = (L + 15) / 16 * 16;
        slot_of.assign(n1h, 0); pk_of.assign((size_t)P * M + 1, 0);
        bool ok = true;
        for (size_t p = 1; p < n1h && ok; ++p) {
            const int i = subc[p], j = subc[p + n1h];
            if (i < 1 || i > L || j < 1 || j > M) { ok = false; break; }
            const int32_t sl = (int32_t)(i + (long long)(j - 1) * P);
            if (pk_of[sl]) ok = false;
            slot_of[p] = sl; pk_of[sl] = (int32_t)p;
        }
        for (int xp = 0; xp < 2 && ok && !E->embedded; ++xp)
            for (int yp = 0; yp < 2 && !E->embedded; ++yp) {
                if (d.slab && yp) continue;
                const beom_dense::HostNb nb{L, M, xp, yp};
                bool match = true;
                for (size_t p = 1; p < n1h && match; ++p) {
                    const int i = subc[p], j = subc[p + n1h];
                    for (int k = 0; k < 8 && match; ++k) {
                        const int q = nb.at(i + beom_dense::kDi[k], j + beom_dense::kDj[k]);      // packed-pitch index on the L x M rectangle, or 0
                        const int32_t want = neig[k + 8 * p];
                        const int32_t got = q ? pk_of[(size_t)((q - 1) % L + 1) + (size_t)((q - 1) / L) * P] : 0;
                        match = want == got;
                    }
                }
                if (match) { E->embedded = true; d.xper = xp; d.yper = yp; }
            }
        if (E->embedded) {
            // 64 x 4 tiles whose every cell, and every cell within 3 of it, is wet interior with unit masks
            const int ntx = (L + 63) / 64, nty = (M + 3) / 4;
            std::vector<unsigned char> good((size_t)(L + 2) * (M + 2), 0);       // (i, j) in 0..L+1 x 0..M+1
            for (size_t p = 1; p < n1h; ++p)
                if (mk_n[p] == 1.0 && mk_u[p] == 1.0 && mk_v[p] == 1.0 && mkpe[p] == 1.0 && mkpi[p] == 1.0)
                    good[(size_t)subc[p] + (size_t)subc[p + n1h] * (L + 2)] = 1;
            // 2-D prefix sums of "not good" -> any bad cell in a window
            std::vector<int32_t> bad((size_t)(L + 3) * (M + 3), 0);
            for (int j = 0; j <= M + 1; ++j)
                for (int i = 0; i <= L + 1; ++i)
                    bad[(size_t)(i + 1) + (size_t)(j + 1) * (L + 3)] = (good[(size_t)i + (size_t)j * (L + 2)] ? 0 : 1)
                        + bad[(size_t)i + (size_t)(j + 1) * (L + 3)] + bad[(size_t)(i + 1) + (size_t)j * (L + 3)] - bad[(size_t)i + (size_t)j * (L + 3)];
            auto any_bad = [&](int i0, int i1, int j0, int j1) {          // inclusive window, clipped to 0..L+1 x 0..M+1 (the margin is bad)
                if (i0 < 0 || j0 < 0 || i1 > L + 1 || j1 > M + 1) return true;
                return bad[(size_t)(i1 + 1) + (size_t)(j1 + 1) * (L + 3)] - bad[(size_t)i0 + (size_t)(j1 + 1) * (L + 3)]
                       - bad[(size_t)(i1 + 1) + (size_t)j0 * (L + 3)] + bad[(size_t)i0 + (size_t)j0 * (L + 3)] != 0;
            };
            reg4.assign((size_t)ntx * nty, 0);
            for (int ty = 0; ty < nty; ++ty)
                for (int tx = 0; tx < ntx; ++tx) {
                    const int x0 = tx * 64 + 1, y0 = ty * 4 + 1;
                    reg4[(size_t)ty * ntx + tx] = any_bad(x0 - 3, x0 + 63 + 3, y0 - 3, y0 + 3 + 3) ? 0 : 1;
                }
            d.reg_nx = ntx;
            E->dense = true;              // the dense kernels, with masks from arrays where a tile is not regular
        }
    }
    d.embedded = E->embedded ? 1 : 0;
    E->lid = prm->rgld > 0.5;
    if (E->lid) {
        E->subc_host.assign(subc, subc + 2 * n1h);
        E->neig_host.assign(neig, neig + 8 * n1h);
    }
    if (E->lid || (E->embedded && prm->flag_nudging && prm->mcbc < 0.5)) {      // packed index -> device index, for tables uploaded later
        E->dev_index.assign(n1h, 0);
        const int P = (d.L + 15) / 16 * 16;
        for (size_t p = 1; p < n1h; ++p)
            E->dev_index[p] = E->embedded ? slot_of[p] : E->dense ? (int32_t)((p - 1) % d.L + 1 + ((p - 1) / d.L) * (size_t)P) : (int32_t)p;
    }
    // device layout: padded row pitch for dense frames (DevView::P), the caller's packed layout otherwise
    d.P = 0; d.ncell = prm->ndeg;
    if (E->dense) {
        d.P = (getenv("BEOM_NO_PITCH") && !E->embedded) ? d.L : (d.L + 15) / 16 * 16;      // (BEOM_NO_PITCH: the packed pitch, for A/B measurements)
        d.ncell = (long long)d.P * d.M;
        d.n1 = (d.ncell + 1 + 15) / 16 * 16;
    }
    const size_t n1 = (size_t)d.n1;                   // cells per layer on the device
    int rc = 0;
    E->stage_bytes = n1h * 32;                        // the widest slice: neig (8 x int32), a history (3 doubles)
    HIP_TRY_E(hipMalloc(&E->stage, E->stage_bytes));
    if (E->embedded) {
        int32_t *q = nullptr;
        if ((rc = dev_alloc(E, &q, slot_of.size(), errm, errm_len, false))) { beom_destroy(E); return rc; }
        HIP_TRY_E(hipMemcpyAsync(q, slot_of.data(), slot_of.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
        E->slot_of_dev = q;
        if ((rc = dev_alloc(E, &q, (size_t)d.n1, errm, errm_len, true))) { beom_destroy(E); return rc; }
        HIP_TRY_E(hipMemcpyAsync(q, pk_of.data(), pk_of.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
        d.pk_of = q;
        unsigned char *r = nullptr;
        if ((rc = dev_alloc(E, &r, reg4.size(), errm, errm_len, false))) { beom_destroy(E); return rc; }
        HIP_TRY_E(hipMemcpyAsync(r, reg4.data(), reg4.size(), hipMemcpyHostToDevice, E->stream));
        d.reg4 = r;
        HIP_TRY_E(hipStreamSynchronize(E->stream));
    }
#define UP(name, src, outer, inner) if ((rc = dev_upload(E, &d.name, src, (size_t)(outer), (inner), errm, errm_len))) { beom_destroy(E); return rc; }
    if ((rc = dev_upload<int32_t, true>(E, &d.neig, neig, 1, 8, errm, errm_len))) { beom_destroy(E); return rc; }
    UP(subc, subc, 2, 1)
    UP(mk_u, mk_u, 1, 1) UP(mk_v, mk_v, 1, 1) UP(mk_n, mk_n, 1, 1) UP(mkpe, mkpe, 1, 1) UP(mkpi, mkpi, 1, 1)
    UP(fcor, fcor, 1, 1) UP(h_th, h_th, 1, 1) UP(h_to, h_to, 1, 1)
    UP(nudg, nudg, 3, 1) UP(fnud, fnud, 3 * nl, 1) UP(hdot, hdot, nl, 1)
    UP(tide, tide, 3, 2) UP(taus, taus, 2, 1)
    {   // the wind stress as tt3d holds it: at real cells only (distribute_stress writes cells 1..ndeg, :1945-1966; index 0 and
        // every device slot that is no cell stay +0)
        std::vector<double> tc(2 * n1h, 0.0);
        if (taus) { std::memcpy(tc.data(), taus, 2 * n1h * sizeof(double)); tc[0] = 0.0; tc[n1h] = 0.0; }
        UP(taus_cells, tc.data(), 2, 1)
    }
#undef UP
    {   // bodf(nlay, 2): no cell dimension
        double *q = nullptr;
        if ((rc = dev_alloc(E, &q, 2 * nl, errm, errm_len, true))) { beom_destroy(E); return rc; }
        if (bodf) HIP_TRY_E(hipMemcpyAsync(q, bodf, 2 * nl * sizeof(double), hipMemcpyHostToDevice, E->stream));
        d.bodf = q;
    }
    for (size_t i = 0; i < n1h; ++i) { if (mk_u[i] > 0.5) E->any_u = 1; if (mk_v[i] > 0.5) E->any_v = 1; }
    d.woff = nullptr;
    if (!E->dense && getenv("BEOM_NO_WAVE_TABLE") == nullptr) {      // table path: which runs of 64 cells are uniform interior?
        const long long nw = ((long long)prm->ndeg + 63) / 64;
        std::vector<int32_t> woff((size_t)(2 * nw), 0);
        long long nuni = 0;
        for (long long w = 0; w < nw; ++w) {
            const long long p0 = 64 * w + 1;
            if (p0 + 63 > prm->ndeg) break;
            const int32_t *r0 = neig + 8 * p0;
            const int dN = r0[2] - (int)p0, dS = (int)p0 - r0[6];
            bool ok = dN > 0 && dS > 0 && r0[2] != 0 && r0[6] != 0;
            for (long long p = p0; ok && p < p0 + 64; ++p) {
                const int32_t *r = neig + 8 * p;
                ok = r[0] == p + 1 && r[4] == p - 1 && r[2] == p + dN && r[6] == p - dS &&
                     r[1] == p + dN + 1 && r[3] == p + dN - 1 && r[5] == p - dS - 1 && r[7] == p - dS + 1 &&
                     r[3] >= 1 && r[5] >= 1 && r[1] <= prm->ndeg &&
                     mk_u[p] == 1.0 && mk_v[p] == 1.0 && mk_n[p] == 1.0 && mkpe[p] == 1.0 && mkpi[p] == 1.0;
                for (int q = 0; ok && q < 8; ++q) ok = mk_n[r[q]] == 1.0;
            }
            if (ok) { woff[2 * w] = dN; woff[2 * w + 1] = dS; ++nuni; }
        }
        E->uniform_waves = nuni; E->total_waves = nw;
        if (nuni > 0) {      // (table path: packed layout, no repacking)
            int32_t *q = nullptr;
            if ((rc = dev_alloc(E, &q, (size_t)(2 * nw), errm, errm_len, false))) { beom_destroy(E); return rc; }
            HIP_TRY_E(hipMemcpyAsync(q, woff.data(), (size_t)(2 * nw) * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
            HIP_TRY_E(hipStreamSynchronize(E->stream));
            d.woff = q;
        }
    }
    d.has_hdot = any_nonzero(hdot, nl * n1h);
    d.has_tide = any_nonzero(tide, 6 * n1h);
    d.has_bodf = any_nonzero(bodf, 2 * nl);
    d.has_nudg = any_nonzero(nudg, 3 * n1h);
    d.ngt = nullptr; d.ngt_nx = 0;
    if (E->dense && d.has_nudg && getenv("BEOM_NO_NUDG_TILES") == nullptr) {
        // which 64 x 4 tiles of the rectangle hold a non-zero relaxation rate at all (sponges are a few rows or columns)
        const int ntx = (d.L + 63) / 64, nty = (d.M + 3) / 4;
        std::vector<unsigned char> ngt((size_t)ntx * nty, 0);
        for (size_t pk = 1; pk < n1h; ++pk) {
            // (i, j) of the packed cell on the rectangle — local rows: subc(:, 2) of a band holds the global row
            int ci, cj;
            if (E->embedded) { const long long sl = slot_of[pk]; ci = (int)((sl - 1) % d.P) + 1; cj = (int)((sl - 1) / d.P) + 1; }
            else { ci = (int)((pk - 1) % (size_t)d.L) + 1; cj = (int)((pk - 1) / (size_t)d.L) + 1; }
            if (ci < 1 || ci > d.L || cj < 1 || cj > d.M) continue;
            unsigned char &t = ngt[(size_t)((cj - 1) >> 2) * ntx + ((ci - 1) >> 6)];
            for (int iv = 0; iv < 3; ++iv) if (nudg[pk + (size_t)iv * n1h] != 0.0) t |= (unsigned char)(1u << iv);
        }
        size_t flagged = 0;
        for (unsigned char t : ngt) flagged += t != 0;
        // (the look-up is one more dependent load in front of the rate: it pays where most tiles are free of nudging — carrier
        //  beach 8192x1024x8 -2 % per step; a frame nudged over a third of its tiles goes without, wind case +1.5 % with it)
        if (3 * flagged <= ngt.size()) {
            unsigned char *r = nullptr;
            if ((rc = dev_alloc(E, &r, ngt.size(), errm, errm_len, false))) { beom_destroy(E); return rc; }
            HIP_TRY_E(hipMemcpyAsync(r, ngt.data(), ngt.size(), hipMemcpyHostToDevice, E->stream));
            HIP_TRY_E(hipStreamSynchronize(E->stream));
            d.ngt = r; d.ngt_nx = ntx;
        }
    }
    d.has_hto = any_nonzero(h_to, n1h);
    d.keep_diag = 0; d.lean_d2h = 0; E->lean_d2h = true;
    E->fuse = getenv("BEOM_NO_FUSE") == nullptr;
    E->fuse_uv = getenv("BEOM_NO_FUSE") == nullptr && getenv("BEOM_NO_FUSE_UV") == nullptr;
    // frames of few rounds of workgroups: the 64 x 4 tile geometry, one row per thread (a workgroup's lifetime is what the
    // step time is made of there).  Same box, us per step, 64 x 8 -> 64 x 4: stommel 128^2 31.2 -> 24.0, soliton 2048x256
    // 53.5 -> 43.6, 1024x128x4 74.8 -> 56.3, sill 4096x512x4 614 -> 596; jet 2048^2 x 2 535 -> 557, 4096^2 x 4 and larger: slower
    E->tile4 = E->dense && (long long)((d.L + 63) / 64) * ((d.M + 7) / 8) <= 5000;
    if (getenv("BEOM_TILE4")) E->tile4 = E->dense && atoi(getenv("BEOM_TILE4")) != 0;      // (A/B switch)
    if (E->lid) E->fuse = E->fuse_uv = false;      // the lid's flux rebuild reads the stored d2hx, d2hy of the last layer
    d.edge_global = getenv("BEOM_EDGE_GLOBAL") != nullptr;
    E->wind = false;
    if (taus) for (size_t i = 0; i < 2 * n1h; ++i) if (std::fabs(taus[i]) > 1.e-7) { E->wind = true; break; }   // :1945
    E->bot = prm->bdrg > 1.e-7;                                                                                  // :1969
    E->top = prm->tdrg > 1.e-7;                                                                                  // :1991
    d.has_wind = E->wind; d.has_bot = E->bot; d.has_top = E->top;
    d.has_stress = E->wind || E->bot || E->top;
    d.stress_fold = 0;
    d.rho_top = prm->rhon[0]; d.rho_bot = prm->rhon[nl - 1];
    {
        bool neg0 = false;                         // a body force of exactly -0 would make the sign of a skipped +-0 visible
        if (bodf) for (size_t i = 0; i < 2 * nl; ++i) if (bodf[i] == 0.0 && std::signbit(bodf[i])) neg0 = true;
        E->fold_static_ok = E->dense && prm->ocrp < 0.5 && !E->lid && d.has_stress && !neg0 && getenv("BEOM_NO_FOLD_STRESS") == nullptr;
    }
#define AL(name, n) if ((rc = dev_alloc(E, &d.name, (size_t)(n), errm, errm_len))) { beom_destroy(E); return rc; }
    AL(hlay, nl * n1) AL(u, nl * n1) AL(v, nl * n1) AL(h_u, nl * n1) AL(h_v, nl * n1)
    AL(rs[0], nl * n1) AL(rs[1], nl * n1)
    AL(dmx[0], nl * n1) AL(dmx[1], nl * n1) AL(dmx[2], nl * n1)
    AL(dmy[0], nl * n1) AL(dmy[1], nl * n1) AL(dmy[2], nl * n1)
    if (E->dense) {            // partners for the fused U+V sweep
        AL(dmx[3], nl * n1) AL(dmy[3], nl * n1)
        AL(u_alt, nl * n1) AL(v_alt, nl * n1) AL(hu_alt, nl * n1) AL(hv_alt, nl * n1)
    }
    AL(v_cc, nl * n1) AL(v_ll, nl * n1)
    AL(tt3d, 2 * nl * n1) AL(tb3d, 2 * nl * n1) AL(tu3d, 2 * nl * n1)
    AL(pcd, nl * n1) AL(qlr, nl * n1)
    AL(mont, nl * n1) AL(rvor, nl * n1) AL(pvor, nl * n1) AL(dive, nl * n1) AL(d2hx, nl * n1) AL(d2hy, nl * n1)
    if (prm->svis > 0.0) { AL(delu, nl * n1) AL(delv, nl * n1) AL(uu4, nl * n1) AL(vv4, nl * n1) }
    if (E->lid) { AL(pi_s, n1) AL(pi_rhs, n1) AL(pi_prev, n1) }
#undef AL
    // initialize_variables: v_cc = v_ll = bvis everywhere, sentinel included (:276-277)
    if (prm->bvis != 0.0) {
        std::vector<double> b(nl * n1, prm->bvis);      // (padding slots too: never read)
        HIP_TRY_E(hipMemcpyAsync(d.v_cc, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, E->stream));
        HIP_TRY_E(hipMemcpyAsync(d.v_ll, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, E->stream));
        HIP_TRY_E(hipStreamSynchronize(E->stream));
    }
    const unsigned gx = (unsigned)((d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK);          // launches over all cell slots
    const unsigned gx0 = (unsigned)((d.ncell + 1 + BEOM_BLOCK - 1) / BEOM_BLOCK);
    E->grid_cells_layers_flat = dim3(gx, (unsigned)prm->nlay, 1);
    E->grid_cells0 = dim3(gx0, 1, 1);
    HIP_TRY_E(hipStreamSynchronize(E->stream));
#undef HIP_TRY_E
    *out = E;
    return 0;
}

int beom_destroy(beom_handle E) {
    if (!E) return 0;
    (void)hipSetDevice(E->device);
    if (E->stream) (void)hipStreamSynchronize(E->stream);
    for (void *p : E->allocs) (void)hipFree(p);
    if (E->stage) (void)hipFree(E->stage);
    if (E->timer) { for (hipEvent_t ev : E->timer->ev) (void)hipEventDestroy(ev); delete E->timer; }
    if (E->own_stream) (void)hipStreamDestroy(E->own_stream);
    delete E;
    return 0;
}

// ---- host <-> device copies with the Fortran layouts ------------------------------
// [outer][0:ndeg] arrays of the caller (outer = nlay, or 2*nlay for the stresses), slice by slice
static int copy_in(beom_engine *E, double *dst, const double *src, size_t outer, char *errm, int errm_len) {
    if (!src) return 0;
    const size_t n1h = (size_t)E->d.ndeg + 1;
    for (size_t o = 0; o < outer; ++o) {
        const int rc = slice_to_device<double>(E, dst + o * (size_t)E->d.n1, src + o * n1h, 1, 1, 0, errm, errm_len);
        if (rc) return rc;
    }
    return 0;
}
static int copy_out(beom_engine *E, double *dst, const double *src, size_t outer, char *errm, int errm_len) {
    if (!dst) return 0;
    const size_t n1h = (size_t)E->d.ndeg + 1;
    for (size_t o = 0; o < outer; ++o) {
        slice_gather<double>(E, src + o * (size_t)E->d.n1, 1, 1, 0);
        const int rc = slice_to_host<double>(E, src, dst + o * n1h, 1, 1, errm, errm_len);
        if (rc) return rc;
    }
    return 0;
}

// AoS history (m, 0:ndeg, nlay) <-> K separate (0:ndeg, nlay) device arrays, layer by layer
static int hist_in(beom_engine *E, double *const *dev, int K, const double *src, char *errm, int errm_len) {
    if (!src) return 0;
    const size_t n1h = (size_t)E->d.ndeg + 1;
    for (int l = 0; l < E->d.nlay; ++l)
        for (int m = 0; m < K; ++m) {       // the layer's image is uploaded once, then one scatter per level
            const int rc = slice_to_device<double>(E, dev[m] + (size_t)l * E->d.n1, m == 0 ? src + (size_t)l * n1h * K : nullptr, 1, K, m,
                                                   errm, errm_len);
            if (rc) return rc;
        }
    return 0;
}
static int hist_out(beom_engine *E, double *const *dev, int K, double *dst, char *errm, int errm_len) {
    if (!dst) return 0;
    const size_t n1h = (size_t)E->d.ndeg + 1;
    for (int l = 0; l < E->d.nlay; ++l) {
        for (int m = 0; m < K; ++m) slice_gather<double>(E, dev[m] + (size_t)l * E->d.n1, 1, K, m);
        const int rc = slice_to_host<double>(E, dev[0], dst + (size_t)l * n1h * K, 1, K, errm, errm_len);
        if (rc) return rc;
    }
    return 0;
}

int beom_upload_state(beom_handle E, const double *hlay, const double *u, const double *v,
                      const double *h_u, const double *h_v, const double *rs_h, const double *dmdx,
                      const double *dmdy, const double *v_cc, const double *v_ll, const double *tt3d,
                      const double *tb3d, const double *tu3d, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    const size_t nl = (size_t)d.nlay, n = ((size_t)d.ndeg + 1) * nl;
    int rc;
    if ((rc = copy_in(E, d.hlay, hlay, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.u, u, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.v, v, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.h_u, h_u, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.h_v, h_v, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.v_cc, v_cc, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.v_ll, v_ll, nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.tt3d, tt3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.tb3d, tb3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = copy_in(E, d.tu3d, tu3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = hist_in(E, d.rs, 2, rs_h, errm, errm_len))) return rc;
    if ((rc = hist_in(E, d.dmx, 3, dmdx, errm, errm_len))) return rc;
    if ((rc = hist_in(E, d.dmy, 3, dmdy, errm, errm_len))) return rc;
    // a caller may upload stresses computed elsewhere: keep those terms live
    if (any_nonzero(tt3d, 2 * n)) E->up_tt = true;
    if (any_nonzero(tb3d, 2 * n)) E->up_tb = true;
    if (any_nonzero(tu3d, 2 * n)) E->up_tu = true;
    if (E->up_tt || E->up_tb || E->up_tu) d.has_stress = 1;
    // zero-viscosity shortcut: only while v_cc, v_ll are +0 bit for bit (-0 would flip the sign of the products)
    for (const double *a : {v_cc, v_ll})
        if (a) for (size_t i = 0; i < n && E->visc_all_zero; ++i) { uint64_t b; memcpy(&b, &a[i], 8); if (b != 0) E->visc_all_zero = false; }
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_download_state(beom_handle E, double *hlay, double *u, double *v, double *h_u, double *h_v,
                        double *rs_h, double *dmdx, double *dmdy, double *v_cc, double *v_ll,
                        double *tt3d, double *tb3d, double *tu3d, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    const size_t nl = (size_t)d.nlay;
    int rc;
    if ((rc = copy_out(E, hlay, d.hlay, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, u, d.u, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, v, d.v, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, h_u, d.h_u, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, h_v, d.h_v, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, v_cc, d.v_cc, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, v_ll, d.v_ll, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, tt3d, d.tt3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, tb3d, d.tb3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, tu3d, d.tu3d, 2 * nl, errm, errm_len))) return rc;
    if ((rc = hist_out(E, d.rs, 2, rs_h, errm, errm_len))) return rc;
    if ((rc = hist_out(E, d.dmx, 3, dmdx, errm, errm_len))) return rc;
    if ((rc = hist_out(E, d.dmy, 3, dmdy, errm, errm_len))) return rc;
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_download_scratch(beom_handle E, double *mont, double *rvor, double *pvor, double *dive,
                          double *d2hx, double *d2hy, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    const size_t nl = (size_t)d.nlay;
    int rc;
    if ((rc = copy_out(E, mont, d.mont, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, rvor, d.rvor, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, pvor, d.pvor, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, dive, d.dive, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, d2hx, d.d2hx, nl, errm, errm_len))) return rc;
    if ((rc = copy_out(E, d2hy, d.d2hy, nl, errm, errm_len))) return rc;
    HIP_TRY(hipStreamSynchronize(E->stream));
    return 0;
}

int beom_set_rigid_lid(beom_handle E, const double *Ow, const double *Os, const double *Osum_, const double *pi_s,
                       char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    if (!E->lid) { set_err(errm, errm_len, "beom_set_rigid_lid: the handle was created with rgld = 0"); return -5; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    int rc;
    if (!E->lid_ready) {
        if (!Ow || !Os || !Osum_) { set_err(errm, errm_len, "beom_set_rigid_lid: the operators Ow, Os, Osum_ are needed on the first call"); return -1; }
        if ((rc = dev_upload(E, &d.Ow, Ow, 1, 1, errm, errm_len))) return rc;
        if ((rc = dev_upload(E, &d.Os, Os, 1, 1, errm, errm_len))) return rc;
        if ((rc = dev_upload(E, &d.Osum_, Osum_, 1, 1, errm, errm_len))) return rc;
        // Levels of the serial sweep's dependency graph: a cell reads the NEW pressure of the neighbours before it in packed
        // order (it comes after them) and the OLD pressure of those after it (they come after it).  On a plain frame the
        // levels are the anti-diagonals i + j; the wrapped neighbours of an orphan column / row cell bend them.
        const size_t n1h = (size_t)d.ndeg + 1;
        std::vector<int32_t> level(n1h, 0), after(n1h, 0);
        int nlevel = 1;
        for (size_t p = 1; p < n1h; ++p) {
            const int i = E->subc_host[p], j = E->subc_host[p + n1h];
            const int32_t *nb = &E->neig_host[8 * p];
            const int32_t reads[4] = {i < d.lm ? nb[0] : 0, j < d.mm_glob ? nb[2] : 0, i > 1 ? nb[4] : 0, j > 1 ? nb[6] : 0};
            int32_t lv = after[p];
            for (int32_t qn : reads)
                if (qn > 0 && (size_t)qn < p) lv = std::max(lv, level[(size_t)qn] + 1);
            level[p] = lv;
            for (int32_t qn : reads)
                if (qn > 0 && (size_t)qn > p) after[(size_t)qn] = std::max(after[(size_t)qn], lv + 1);
            nlevel = std::max(nlevel, lv + 1);
        }
        // time between two sweeps of the pipeline: sweep s + 1 may touch a cell once every neighbour AFTER it in packed order has
        // been updated by sweep s — 1 + the largest level difference along such an edge (2 on a plain frame; about lm where
        // a periodic seam makes a cell read the far end of its row)
        int dstep = 2;
        for (size_t p = 1; p < n1h; ++p) {
            const int i = E->subc_host[p], j = E->subc_host[p + n1h];
            const int32_t *nb = &E->neig_host[8 * p];
            const int32_t reads[4] = {i < d.lm ? nb[0] : 0, j < d.mm_glob ? nb[2] : 0, i > 1 ? nb[4] : 0, j > 1 ? nb[6] : 0};
            for (int32_t qn : reads)
                if (qn > 0 && (size_t)qn > p) dstep = std::max(dstep, level[(size_t)qn] - level[p] + 1);
        }
        E->lid_dstep = dstep;
        const int ndiag = nlevel;
        std::vector<int32_t> start((size_t)ndiag + 1, 0), order(n1h > 1 ? n1h - 1 : 1, 0);
        for (size_t p = 1; p < n1h; ++p) ++start[(size_t)level[p] + 1];
        for (int k = 0; k < ndiag; ++k) start[(size_t)k + 1] += start[k];
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (size_t p = 1; p < n1h; ++p) order[(size_t)fill[(size_t)level[p]]++] = E->dev_index[p];
        int32_t *q = nullptr;
        if ((rc = dev_alloc(E, &q, order.size(), errm, errm_len, false))) return rc;
        HIP_TRY(hipMemcpyAsync(q, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
        d.sor_order = q;
        if ((rc = dev_alloc(E, &q, start.size(), errm, errm_len, false))) return rc;
        HIP_TRY(hipMemcpyAsync(q, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
        d.sor_dstart = q;
        d.sor_ndiag = ndiag;
        E->lid_maxwidth = 1;
        for (int k = 0; k < ndiag; ++k) E->lid_maxwidth = std::max(E->lid_maxwidth, (int)(start[(size_t)k + 1] - start[k]));
        // one copy of the pressure per sweep in flight: up to kLidBatch of them within ~6 GB
        E->lid_nring = (int)std::max<long long>(17, std::min<long long>(kLidBatch, (6ll << 30) / ((long long)d.n1 * 8))) + 1;
        if ((rc = dev_alloc(E, &E->lid_ring, (size_t)E->lid_nring * d.n1, errm, errm_len, true))) return rc;      // zeroed: index 0 and every slot that is no cell stay 0
        if ((rc = dev_alloc(E, &E->lid_maxd, (size_t)kLidBatch, errm, errm_len, true))) return rc;
        // the terms of every cell's right-hand side in the order of the serial scatter loops (:1727-1752): the x loop over
        // the packed cells, then the y loop; a cell with i > 1 (j > 1) subtracts its transport from itself and adds it to neig(5)
        // (neig(7)); what goes to the sentinel is dropped
        if ((long long)d.n1 >= (1ll << 29)) { set_err(errm, errm_len, "beom_set_rigid_lid: frame too large for the lid's tables"); return -3; }
        std::vector<int32_t> cnt((size_t)d.n1 + 2, 0);
        for (int pass = 0; pass < 2; ++pass) {                       // pass 0: count, pass 1: fill
            std::vector<int32_t> at;
            std::vector<int32_t> ent;
            if (pass) {
                for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];          // cnt[dev] = first entry of cell dev
                at.assign(cnt.begin(), cnt.end());
                ent.assign((size_t)cnt.back() + 1, 0);
            }
            for (int dir = 0; dir < 2; ++dir)
                for (size_t qk = 1; qk < n1h; ++qk) {
                    if (E->subc_host[qk + dir * n1h] <= 1) continue;
                    const int32_t src = E->dev_index[qk], tgt = E->neig_host[8 * qk + (dir ? 6 : 4)];
                    if (!pass) { ++cnt[(size_t)src + 1]; if (tgt > 0) ++cnt[(size_t)E->dev_index[(size_t)tgt] + 1]; }
                    else {
                        ent[(size_t)at[src]++] = 4 * src + 2 * dir;
                        if (tgt > 0) ent[(size_t)at[E->dev_index[(size_t)tgt]]++] = 4 * src + 2 * dir + 1;
                    }
                }
            if (pass) {
                if ((rc = dev_alloc(E, &q, cnt.size(), errm, errm_len, false))) return rc;
                HIP_TRY(hipMemcpyAsync(q, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
                d.lid_rhs_start = q;
                if ((rc = dev_alloc(E, &q, ent.size(), errm, errm_len, false))) return rc;
                HIP_TRY(hipMemcpyAsync(q, ent.data(), ent.size() * sizeof(int32_t), hipMemcpyHostToDevice, E->stream));
                d.lid_rhs_ent = q;
                HIP_TRY(hipStreamSynchronize(E->stream));
            }
        }
        HIP_TRY(hipStreamSynchronize(E->stream));       // (order, start are host temporaries)
        E->lid_ready = true;
    }
    if (pi_s && (rc = slice_to_device<double>(E, d.pi_s, pi_s, 1, 1, 0, errm, errm_len))) return rc;
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_download_pressure(beom_handle E, double *pi_s, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    if (!E->lid) { set_err(errm, errm_len, "beom_download_pressure: the handle was created with rgld = 0"); return -5; }
    HIP_TRY(hipSetDevice(E->device));
    const int rc = copy_out(E, pi_s, E->d.pi_s, 1, errm, errm_len);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_sync(beom_handle E, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- launches -------------------------------------------------------------------------
static inline void rot2(double *(&a)[2]) { double *t = a[0]; a[0] = a[1]; a[1] = t; }
static inline void rot3(double *(&a)[4]) { double *t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = t; }
static inline void rot4(double *(&a)[4]) { double *t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t; }
static inline void swp(double *&a, double *&b) { double *t = a; a = b; b = t; }

// LAUNCH(kernel-template-name, extra template args..., nz, args...) picks the cell context.
#define LAUNCH_CTX(KERNEL_G, KERNEL_D, nz, ...)                                                            \
    do {                                                                                                  \
        if (E->dense) hipLaunchKernelGGL(KERNEL_D, CellDense::grid(E->d, (nz)), dim3(BEOM_BLOCK), 0, E->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL_G, CellGather::grid(E->d, (nz)), dim3(BEOM_BLOCK), 0, E->stream, __VA_ARGS__);         \
    } while (0)

static void launch_rebuild(beom_engine *E) {
    LAUNCH_CTX(k_rebuild_fluxes<CellGather>, k_rebuild_fluxes<CellDense>, E->d.nlay, E->d);
}
static void launch_h(beom_engine *E, double gene, double ramp, double ctim, bool rotate = true) {
    // variant 1 couples layers inside a cell -> one thread walks nlay..1; variant 0: layer = blockIdx.y
    int nz = (E->d.variant == 1) ? 1 : E->d.nlay;
    // nudged frames: one thread walks the layers of its cell, so the cell's relaxation rate is read once instead of once per
    // layer (carrier beach 8192x1024x8: 802 -> 747 us per launch, sill 4096x512x4: 89 -> 83; same-box A/B, round 3)
    if (E->d.has_nudg) nz = 1;
    if (E->d.has_nudg && E->d.has_tide) LAUNCH_CTX((k_update_h<CellGather, 2>), (k_update_h<CellDense, 2>), nz, E->d, gene, ramp, ctim, 0);
    else if (E->d.has_nudg) LAUNCH_CTX((k_update_h<CellGather, 1>), (k_update_h<CellDense, 1>), nz, E->d, gene, ramp, ctim, 0);
    else LAUNCH_CTX((k_update_h<CellGather, 0>), (k_update_h<CellDense, 0>), nz, E->d, gene, ramp, ctim, 0);
    if (rotate) rot2(E->d.rs);
}
template <class CTX>
static bool launch_mont_all(beom_engine *E) {
    const dim3 g = CTX::grid(E->d, 1), b(BEOM_BLOCK);
    switch (E->d.nlay) {
#define CASE_NL(n) case n: hipLaunchKernelGGL((k_update_mont_all<CTX, n>), g, b, 0, E->stream, E->d); return true;
        CASE_NL(1) CASE_NL(2) CASE_NL(3) CASE_NL(4) CASE_NL(5) CASE_NL(6) CASE_NL(7) CASE_NL(8)
#undef CASE_NL
        default: return false;
    }
}
static void launch_mont(beom_engine *E, int ilay) {
    if (ilay == 0) {
        if (E->dense ? launch_mont_all<CellDense>(E) : launch_mont_all<CellGather>(E)) return;
    }
    const int nz = ilay ? 1 : E->d.nlay;
    LAUNCH_CTX(k_update_mont<CellGather>, k_update_mont<CellDense>, nz, E->d, ilay);
}
static void launch_visc(beom_engine *E, int ilay) {
    const int nz = ilay ? 1 : E->d.nlay;
    LAUNCH_CTX(k_update_visc<CellGather>, k_update_visc<CellDense>, nz, E->d, ilay);
    if (E->d.svis > 0.0) {                         // biharmonic part of update_viscosity (:2508-2599)
        const dim3 g = ilay ? E->grid_cells0 : E->grid_cells_layers_flat;
        hipLaunchKernelGGL(k_biharm_lap, g, dim3(BEOM_BLOCK), 0, E->stream, E->d, ilay);
        hipLaunchKernelGGL(k_biharm_flux, g, dim3(BEOM_BLOCK), 0, E->stream, E->d, ilay);
    }
}
template <bool XDIR>
static void launch_uv(beom_engine *E, int ilay, double gene, double ramp, double ctim, bool prod = false) {
    const int nz = ilay ? 1 : E->d.nlay;
    const int copy_hist = ilay ? 1 : 0;       // a single-layer call cannot rotate shared pointers
    if (prod) hipLaunchKernelGGL((k_update_uv<CellDense, XDIR, true>), CellDense::grid(E->d, nz), dim3(BEOM_BLOCK), 0, E->stream, E->d, ilay, gene, ramp, ctim, copy_hist);
    else LAUNCH_CTX((k_update_uv<CellGather, XDIR>), (k_update_uv<CellDense, XDIR>), nz, E->d, ilay, gene, ramp, ctim, copy_hist);
    if (!copy_hist) { if (XDIR) rot3(E->d.dmx); else rot3(E->d.dmy); }
}
#define TNS t8
#define TSUF t8
#include "beom_launch_tiled.h"
#undef TNS
#undef TSUF
#define TNS t4
#define TSUF t4
#include "beom_launch_tiled.h"
#undef TNS
#undef TSUF
// fused Montgomery + Leith sweep (dense frames); false if no instantiation for this nlay
// leith: this step refreshes the Leith viscosity (:2188, :2268); else the sweep forms the products of the
// standing v_cc, v_ll.  keep_visc: a refreshed viscosity has to stand for later steps (n_3d > 1).
static bool launch_mont_visc(beom_engine *E, bool uv_fused_follows, bool leith, bool keep_visc) {
    E->d.lean_d2h = uv_fused_follows && E->lean_d2h && !E->d.keep_diag;
    E->d.keep_visc = keep_visc;
    E->d.zero_visc = !leith && uv_fused_follows && E->lean_visc && E->visc_all_zero && E->P.dvis == 0.0 && E->P.bvis == 0.0 &&
                     !E->d.keep_diag;
    return E->tile4 ? raw_mont_visc_t4(E, leith) : raw_mont_visc_t8(E, leith);
}
// fused U+V sweep (dense frames): first_x = update_u first (even tstp)
static void uv_fused_swap(beom_engine *E, bool first_x) {
    DevView &d = E->d;
    swp(d.u, d.u_alt); swp(d.v, d.v_alt);                 // both velocities are written out of place
    if (first_x) { rot4(d.dmx); swp(d.h_v, d.hv_alt); rot3(d.dmy); }
    else         { rot4(d.dmy); swp(d.h_u, d.hu_alt); rot3(d.dmx); }
}
static void launch_uv_fused(beom_engine *E, bool first_x, bool prod, double gene, double ramp, double ctim,
                            bool swap = true) {
    const bool zv = prod && E->d.zero_visc;       // set by launch_mont_visc of this step
    if (E->tile4) raw_uv_fused_t4(E, first_x, prod, zv, gene, ramp, ctim);
    else raw_uv_fused_t8(E, first_x, prod, zv, gene, ramp, ctim);
    if (swap) uv_fused_swap(E, first_x);
}
static bool can_fuse(const beom_engine *E, int n_3d, bool first3) {
    // either every step refreshes the viscosity (dvis > 1e-3 and n_3d = 1, :2268) — Montgomery + Leith
    // in one sweep — or no step after the third ever does (dvis <= 1e-3, svis = 0): v_cc, v_ll stand
    // and the sweep forms their products with this step's dive, rvor
    const int nl = E->d.nlay;
    if (!(E->dense && E->fuse && nl <= 8) || E->P.svis > 0.0) return false;
    (void)n_3d;
    if (E->P.dvis > 1.e-3) return true;           // refresh steps: Leith in the sweep; others: standing v_cc, v_ll
    return !first3;                               // steps 1-3 call update_viscosity unconditionally (:2188)
}
// rigid lid: rebuild the transports from the new h and velocities (steps 1-3: centred, :2166-2177; later: upstream with the
// curvatures of the last layer, :2237-2257), then the pressure solve and the velocity correction (surf_pressure, :1705-1838)
static void launch_lid_fluxes(beom_engine *E, bool first3) {
    if (first3) { launch_rebuild(E); return; }
    const dim3 g((unsigned)((E->d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK), (unsigned)E->d.nlay, 1);
    hipLaunchKernelGGL(k_rgld_upstream_fluxes, g, dim3(BEOM_BLOCK), 0, E->stream, E->d);
}
static void launch_lid_h_epilogue(beom_engine *E) {
    hipLaunchKernelGGL(k_rgld_h_epilogue, dim3((unsigned)((E->d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK)), dim3(BEOM_BLOCK), 0, E->stream, E->d);
}
// surf_pressure's iteration (:1757-1802): Gauss-Seidel sweeps until max |change| <= 1e-5 or 1000 sweeps, as a pipeline of
// wavefronts (k_rgld_gs_front): batches of up to kLidBatch sweeps in flight, one launch per time step of the pipeline, the
// host reads the batch's per-sweep maxima and picks the sweep the serial loop would have stopped after (one small
// device-to-host copy per batch: a lid step is not asynchronous).
static int lid_solve(beom_engine *E) {
    DevView &d = E->d;
    const int maxiters = 1000;
    const double pi_tol = 1.e-5;
    const int R = E->lid_nring, D = E->lid_dstep, nlev = d.sor_ndiag;
    const size_t bytes = (size_t)d.n1 * sizeof(double);
    if (hipMemcpyAsync(E->lid_ring, d.pi_s, bytes, hipMemcpyDeviceToDevice, E->stream) != hipSuccess) return -1;      // copy 0 = the pressure before the first sweep
    // A batch costs (levels + 2 * sweeps) dependent launches whatever it finds, and a launch costs the more the more sweeps are
    // in flight: the first batch is sized by what the step before needed (+ 25 %), later ones take all the ring holds.
    int s0 = 1, batch = std::min(R - 1, std::max(16, E->lid_last * 5 / 4 + 8)), chosen = 0;
    unsigned long long maxd[kLidBatch];
    while (!chosen) {
        const int nb = std::min(batch, maxiters - s0 + 1);
        if (hipMemsetAsync(E->lid_maxd, 0, nb * sizeof(unsigned long long), E->stream) != hipSuccess) return -1;
        const dim3 g((unsigned)((E->lid_maxwidth + BEOM_BLOCK - 1) / BEOM_BLOCK), (unsigned)nb, 1);
        const int tend = nlev - 1 + (nb - 1) * D;
        for (int t = 0; t <= tend; ++t)
            hipLaunchKernelGGL(k_rgld_gs_front, g, dim3(BEOM_BLOCK), 0, E->stream, d, E->lid_ring, R, s0, t, D, E->lid_maxd);
        E->lid_launches += tend + 1;
        if (hipMemcpyAsync(maxd, E->lid_maxd, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, E->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(E->stream) != hipSuccess) return -1;
        for (int b = 0; b < nb && !chosen; ++b) {
            double m; memcpy(&m, &maxd[b], sizeof(double));
            if (!(m > pi_tol) || s0 + b == maxiters) chosen = s0 + b;           // `do while (maxdiff > pi_tol .and. iters < maxiters)`
        }
        s0 += nb;
        batch = R - 1;
    }
    E->lid_sweeps += chosen; ++E->lid_solves; E->lid_last = chosen;
    if (hipMemcpyAsync(d.pi_s, E->lid_ring + (size_t)(chosen % R) * d.n1, bytes, hipMemcpyDeviceToDevice, E->stream) != hipSuccess) return -1;
    return 0;
}
static void launch_lid_pressure(beom_engine *E) {
    const dim3 g1((unsigned)((E->d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK)), g((unsigned)((E->d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK), (unsigned)E->d.nlay, 1);
    hipLaunchKernelGGL(k_rgld_rhs, g1, dim3(BEOM_BLOCK), 0, E->stream, E->d);
    if (lid_solve(E)) snprintf(E->last_err, sizeof(E->last_err), "the lid's pressure iteration failed: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(k_rgld_correct, g, dim3(BEOM_BLOCK), 0, E->stream, E->d);
}
static void launch_stress(beom_engine *E) {
    if (!(E->wind || E->bot || E->top)) return;
    const dim3 g((unsigned)((E->d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK));
    hipLaunchKernelGGL(k_stress, g, dim3(BEOM_BLOCK), 0, E->stream, E->d, (int)E->wind, (int)E->bot, (int)E->top);
}

extern "C" {

#define NEED(E) do { if (!(E)) return -1; if (hipSetDevice((E)->device) != hipSuccess) return -9; } while (0)

#define LAUNCHED() (hipGetLastError() == hipSuccess ? 0 : -10)
int beom_update_h(beom_handle E, double gene, double ramp, double ctim) { NEED(E); launch_h(E, gene, ramp, ctim); return LAUNCHED(); }
int beom_update_mont_rvor_pvor_dive_kine(beom_handle E, int ilay) { NEED(E); if (ilay < 0 || ilay > E->d.nlay) return -3; launch_mont(E, ilay); return LAUNCHED(); }
int beom_update_viscosity(beom_handle E, int ilay) { NEED(E); if (ilay < 0 || ilay > E->d.nlay) return -3; launch_visc(E, ilay); return LAUNCHED(); }
int beom_update_u(beom_handle E, int ilay, double gene, double ramp, double ctim) { NEED(E); if (ilay < 0 || ilay > E->d.nlay) return -3; launch_uv<true>(E, ilay, gene, ramp, ctim); return LAUNCHED(); }
int beom_update_v(beom_handle E, int ilay, double gene, double ramp, double ctim) { NEED(E); if (ilay < 0 || ilay > E->d.nlay) return -3; launch_uv<false>(E, ilay, gene, ramp, ctim); return LAUNCHED(); }
int beom_rebuild_fluxes(beom_handle E) { NEED(E); launch_rebuild(E); return LAUNCHED(); }
int beom_distribute_stress(beom_handle E) { NEED(E); launch_stress(E); return LAUNCHED(); }
#undef LAUNCHED

}  // extern "C"

// Per-step scalars of integrate_time (private_mod.f95:1858-1901).
struct StepScalars { double ctim, ramp, gene; bool first3, upst, stress, fused, fused_uv; int n_3d; };
static StepScalars step_scalars(const beom_engine *E, int tstp, double tres, double dtd8, double dt_r,
                                double rsta, int n_3d) {
    StepScalars s;
    s.ctim = tres + dtd8 * (double)tstp;                           // :1862,1887
    s.first3 = tstp <= 3;
    s.ramp = 1.0;
    if (s.first3) {
        const double c1 = tres + dtd8 * 1.0;                       // ramp is set once, at tstp = 1 (:1864-1866)
        if (rsta < 0.5 && c1 < dt_r) s.ramp = c1 / dt_r;
        s.upst = true;                                             // update_viscosity is unconditional (:2188)
        s.stress = tstp == 1;                                      // :1863
    } else {
        if (rsta < 0.5 && s.ctim < dt_r) s.ramp = s.ctim / dt_r;   // :1898-1901
        s.upst = (tstp % n_3d) == 0;                               // :1889-1892
        s.stress = s.upst;                                         // :1894-1896
    }
    s.gene = (s.first3 || (E->lid && E->P.g_fb > 0.5)) ? 0.0 : E->P.g_fb;      // :1859,1877; :1880-1884: no multistep with a lid
    s.n_3d = n_3d;
    s.fused = can_fuse(E, n_3d, s.first3);
    s.fused_uv = E->dense && E->fuse_uv && !(E->P.svis > 0.0);
    return s;
}

// distribute_stress of this step inside its fused momentum sweep (uv_core): the fractions are constants (ocrp = 0), every
// step refreshes the stress (so tt3d, tb3d, tu3d are never read back; steps 2 and 3 reuse step 1's, :1863), and an array
// the engine does not refresh holds nothing but zeros.  The three arrays then keep their values of step 3 ("keep_diag" = 1
// keeps them current).
static bool stress_folds(const beom_engine *E, const StepScalars &s) {
    return E->fold_stress && E->fold_static_ok && s.fused_uv && s.stress && !s.first3 && s.n_3d == 1 && !E->d.keep_diag &&
           (E->wind || !E->up_tt) && (E->bot || !E->up_tb) && (E->top || !E->up_tu);
}

static void one_step(beom_engine *E, int tstp, const StepScalars &s) {
    StepTimer *T = (E->timer && tstp % E->timer->stride == 0) ? E->timer : nullptr;
    if (T) { T->st = E->stream; T->step(tstp); }
    E->d.stress_fold = stress_folds(E, s) ? 1 : 0;
    E->last_folded = E->d.stress_fold != 0;
    if (s.stress && !E->d.stress_fold) launch_stress(E);
    if (E->lid) launch_lid_fluxes(E, s.first3);
    else if (s.first3) launch_rebuild(E);                          // :2166-2177
    if (T) T->begin(0);
    launch_h(E, s.gene, s.ramp, s.ctim);                           // :2181,2259
    if (E->lid) launch_lid_h_epilogue(E);                          // :1648-1700
    const bool leith = E->P.dvis > 1.e-3 && s.upst;
    const bool u_first = tstp % 2 == 0;                            // :2193-2199,2276-2282
    if (T) { T->end(); T->begin(s.fused ? 5 : 1); }
    const bool prod = s.fused && launch_mont_visc(E, s.fused_uv, leith, leith && s.n_3d > 1);              // :2187-2188, 2266-2269 in one sweep
    if (!prod) launch_mont(E, 0);
    if (T) T->end();
    if (!prod && (s.first3 || (E->P.dvis > 1.e-3 && s.upst) || E->P.svis > 0.0)) {     // :2188,2268
        if (T) T->begin(2);
        launch_visc(E, 0);
        if (T) T->end();
    }
    if (s.fused_uv) {
        if (T) T->begin(6);
        launch_uv_fused(E, u_first, prod, s.gene, s.ramp, s.ctim);
        if (T) T->end();
    } else if (u_first) {
        if (T) T->begin(3);
        launch_uv<true>(E, 0, s.gene, s.ramp, s.ctim, prod);
        if (T) { T->end(); T->begin(4); }
        launch_uv<false>(E, 0, s.gene, s.ramp, s.ctim, prod);
        if (T) T->end();
    } else {
        if (T) T->begin(4);
        launch_uv<false>(E, 0, s.gene, s.ramp, s.ctim, prod);
        if (T) { T->end(); T->begin(3); }
        launch_uv<true>(E, 0, s.gene, s.ramp, s.ctim, prod);
        if (T) T->end();
    }
    if (E->obc) {                                                  // :2201-2204, 2285-2288
        const dim3 g((unsigned)((E->d.nseg + BEOM_BLOCK - 1) / BEOM_BLOCK), (unsigned)E->d.nlay, 1);
        hipLaunchKernelGGL(k_no_gradient_obc, g, dim3(BEOM_BLOCK), 0, E->stream, E->d, 0);
        hipLaunchKernelGGL(k_no_gradient_obc, g, dim3(BEOM_BLOCK), 0, E->stream, E->d, 1);
    }
    if (E->lid) { launch_lid_fluxes(E, s.first3); launch_lid_pressure(E); }       // :2206-2222, 2290-2316
    E->d.stress_fold = 0;
}

// rows [jlo, jlo+nrows) of hlay,u,v,h_u,h_v  ->  dbuf (device memory, 5*nlay*nrows*(lm+1) doubles); the *2 forms move a second
// group of as many rows to / from a second buffer in the same launch
template <bool PACK>
static int rows_copy(beom_handle E, int jlo, int nrows, void *dbuf, int jlo2, void *dbuf2) {
    if (!E || !dbuf || jlo < 1 || nrows < 1 || jlo + nrows - 1 > E->d.M) return -3;
    if (dbuf2 && (jlo2 < 1 || jlo2 + nrows - 1 > E->d.M)) return -3;
    if (hipSetDevice(E->device) != hipSuccess) return -9;
    const long long total = 5ll * E->d.nlay * nrows * E->d.L;
    hipLaunchKernelGGL((k_rows_copy<PACK>), dim3((unsigned)((total + BEOM_BLOCK - 1) / BEOM_BLOCK), dbuf2 ? 2u : 1u), dim3(BEOM_BLOCK),
                       0, E->stream, E->d, jlo, nrows, (double *)dbuf, jlo2, (double *)dbuf2);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}
extern "C" {

int beom_step(beom_handle E, int tstp_first, int nsteps, double tres, double dtd8, double dt_r,
              double rsta, int n_3d, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    if (tstp_first < 1 || nsteps < 0 || n_3d < 1) { set_err(errm, errm_len, "beom_step: bad arguments"); return -3; }
    if (E->P.flag_nudging && E->P.mcbc < 0.5 && !E->obc && !E->obc_set) { set_err(errm, errm_len, "beom_step: mcbc = 0 with nudging needs beom_set_open_boundaries (no_gradient_obc, private_mod.f95:2613-2679)"); return -6; }
    if (E->lid && !E->lid_ready) { set_err(errm, errm_len, "beom_step: rgld = 1 needs beom_set_rigid_lid (the Poisson operators Ow, Os, Osum_ and the lid pressure, private_mod.f95:505-563)"); return -6; }
    HIP_TRY(hipSetDevice(E->device));
    for (int tstp = tstp_first; tstp < tstp_first + nsteps; ++tstp)
        one_step(E, tstp, step_scalars(E, tstp, tres, dtd8, dt_r, rsta, n_3d));
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_profile_start(beom_handle E) {
    if (!E) return -1;
    if (E->timer) { for (hipEvent_t ev : E->timer->ev) (void)hipEventDestroy(ev); delete E->timer; }
    E->timer = new StepTimer();
    E->timer->st = E->stream;
    E->timer->stride = E->profile_stride;
    E->timer->rotate = E->profile_rotate;
    return 0;
}

int beom_profile_stop(beom_handle E, double *ms, int *launches, char *errm, int errm_len) {
    if (!E || !ms || !launches) { set_err(errm, errm_len, "null argument"); return -1; }
    if (!E->timer) { set_err(errm, errm_len, "beom_profile_stop without beom_profile_start"); return -3; }
    HIP_TRY(hipSetDevice(E->device));
    HIP_TRY(hipStreamSynchronize(E->stream));
    StepTimer &T = *E->timer;
    for (int c = 0; c < 8; ++c) { ms[c] = 0.0; launches[c] = 0; }
    for (size_t k = 0; k < T.cls.size(); ++k) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, T.ev[2 * k], T.ev[2 * k + 1]);
        ms[T.cls[k]] += (double)t;
        launches[T.cls[k]] += 1;
    }
    for (hipEvent_t ev : T.ev) (void)hipEventDestroy(ev);
    delete E->timer;
    E->timer = nullptr;
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_profile_steps(beom_handle E, int tstp_first, int nsteps, double tres, double dtd8, double dt_r,
                       double rsta, int n_3d, double *ms, int *launches, char *errm, int errm_len) {
    int rc = beom_profile_start(E);
    if (rc) { set_err(errm, errm_len, "null handle"); return rc; }
    rc = beom_step(E, tstp_first, nsteps, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len);
    if (rc) return rc;
    return beom_profile_stop(E, ms, launches, errm, errm_len);
}

// ---- split step for the ghost-row exchange overlap ------------------------
// A band's step in three parts, so that the rows the neighbours are waiting for are finished, packed and on their way
// while the bulk of the momentum sweep still runs — "boundary first" within ONE step; every part sees ghost rows that
// have landed, nothing in flight is read:
//   part 1: everything up to the momentum sweeps on all rows (stress, rebuild of steps 1-3, update_h, Montgomery + Leith);
//   part 2: the fused u+v sweep on the strips next to the ghost zones (rows 1..8 and M-7..M: the ghost rows and the
//           outermost owned rows, which are what beom_pack_rows sends) — the caller enqueues it on ANOTHER stream,
//           behind part 1, followed by pack -> transport -> unpack;
//   part 3: the same sweep on the rows in between, on the handle's usual stream, then the pointer rotations.
// The fused sweep writes out of place and every workgroup evaluates its own halo, so parts 2 and 3 may run side by side;
// the unpack writes rows 1..4 / M-3..M, part 3 reads no row below 6 / above M-5.  G = 4 ghost rows per neighbour
// (beom_amd/slab.py).  Needs the fused u+v sweep and no open-boundary pass; returns -20 when the caller must use
// beom_step instead (call part 1 first: it decides).
static void set_rows(DevView &d, int n, int lo0, int hi0, int lo1 = 1, int hi1 = 0) {
    d.nstrip = n; d.jlo0 = lo0; d.jhi0 = hi0; d.jlo1 = lo1; d.jhi1 = hi1;
}
constexpr int kEdgeRows = 8;           // ghost rows + the owned rows a neighbour receives

int beom_step_phase(beom_handle E, int tstp, double tres, double dtd8, double dt_r, double rsta, int n_3d,
                    int phase, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    if (E->P.flag_nudging && E->P.mcbc < 0.5 && !E->obc && !E->obc_set) { set_err(errm, errm_len, "beom_step_phase: mcbc = 0 with nudging needs beom_set_open_boundaries (no_gradient_obc, private_mod.f95:2613-2679)"); return -6; }
    const StepScalars s = step_scalars(E, tstp, tres, dtd8, dt_r, rsta, n_3d);
    const bool south = d.slab && d.joff > 0, north = d.slab && d.joff + d.M < d.Mg;
    if (!s.fused_uv || !(south || north) || d.M < 4 * kEdgeRows || phase < 1 || phase > 3 || E->obc || E->lid) {
        set_err(errm, errm_len, "beom_step_phase: split step not available for this step/configuration");
        return -20;
    }
    const int M = d.M;
    const bool u_first = tstp % 2 == 0;
    StepTimer *T = (E->timer && tstp % E->timer->stride == 0) ? E->timer : nullptr;
    if (T) { T->st = E->stream; T->step(tstp); }
    E->d.stress_fold = stress_folds(E, s) ? 1 : 0;
    E->last_folded = E->d.stress_fold != 0;
    if (phase == 1) {                                                  // one_step up to the momentum sweeps
        if (s.stress && !E->d.stress_fold) launch_stress(E);
        if (s.first3) launch_rebuild(E);                               // :2166-2177
        if (T) T->begin(0);
        launch_h(E, s.gene, s.ramp, s.ctim);                           // :2181,2259
        const bool leith = E->P.dvis > 1.e-3 && s.upst;
        if (T) { T->end(); T->begin(s.fused ? 5 : 1); }
        E->split_prod = s.fused && launch_mont_visc(E, true, leith, leith && s.n_3d > 1);
        if (!E->split_prod) launch_mont(E, 0);
        if (T) T->end();
        if (!E->split_prod && (s.first3 || (E->P.dvis > 1.e-3 && s.upst) || E->P.svis > 0.0)) {     // :2188,2268
            if (T) T->begin(2);
            launch_visc(E, 0);
            if (T) T->end();
        }
    } else if (phase == 2) {
        // a side without a neighbour has no strip (its rows belong to part 3)
        if (south && north) set_rows(d, 2, 1, kEdgeRows, M - kEdgeRows + 1, M);
        else if (south) set_rows(d, 1, 1, kEdgeRows);
        else set_rows(d, 1, M - kEdgeRows + 1, M);
        if (T) T->begin(6);
        launch_uv_fused(E, u_first, E->split_prod, s.gene, s.ramp, s.ctim, false);
        if (T) T->end();
    } else {
        set_rows(d, 1, south ? kEdgeRows + 1 : 1, north ? M - kEdgeRows : M);
        if (T) T->begin(6);
        launch_uv_fused(E, u_first, E->split_prod, s.gene, s.ramp, s.ctim, true);
        if (T) T->end();
    }
    set_rows(d, 1, 1, M);
    d.stress_fold = 0;
    HIP_TRY(hipGetLastError());
    return 0;
}

int beom_pack_rows(beom_handle E, int jlo, int nrows, void *dbuf) { return rows_copy<true>(E, jlo, nrows, dbuf, 0, nullptr); }
int beom_unpack_rows(beom_handle E, int jlo, int nrows, const void *dbuf) { return rows_copy<false>(E, jlo, nrows, const_cast<void *>(dbuf), 0, nullptr); }
int beom_pack_rows2(beom_handle E, int nrows, int jlo_a, void *dbuf_a, int jlo_b, void *dbuf_b) {
    return rows_copy<true>(E, jlo_a, nrows, dbuf_a, jlo_b, dbuf_b);
}
int beom_unpack_rows2(beom_handle E, int nrows, int jlo_a, const void *dbuf_a, int jlo_b, const void *dbuf_b) {
    return rows_copy<false>(E, jlo_a, nrows, const_cast<void *>(dbuf_a), jlo_b, const_cast<void *>(dbuf_b));
}

// Output preparation on the device: replaces the array work of write_array for
// 'eta_', 'u___', 'v___' (private_mod.f95:2848-2883) and the min/max + thin-layer scans of
// write_outputs (:2772-2808).  Only real*4 records cross PCIe.
int beom_download_outputs(beom_handle E, const float *h0r4, float *eta, float *u4, float *v4,
                          double *minmax, int *thin_layer, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    const size_t n = (size_t)d.ndeg * d.nlay;
    if (!E->h0r4_dev) {
        if (!h0r4) { set_err(errm, errm_len, "beom_download_outputs: h_0 (real*4) needed on the first call"); return -3; }
        HIP_TRY(hipMalloc((void **)&E->h0r4_dev, n * sizeof(float)));
        E->allocs.push_back(E->h0r4_dev);
        for (int q = 0; q < 3; ++q) { HIP_TRY(hipMalloc((void **)&E->out4[q], n * sizeof(float))); E->allocs.push_back(E->out4[q]); }
        const size_t nb = (size_t)E->grid_cells0.x;
        HIP_TRY(hipMalloc((void **)&E->scan_dev, nb * d.nlay * 7 * sizeof(double)));
        E->allocs.push_back(E->scan_dev);
    }
    if (h0r4) HIP_TRY(hipMemcpyAsync(E->h0r4_dev, h0r4, n * sizeof(float), hipMemcpyHostToDevice, E->stream));
    const unsigned gx = (unsigned)((d.ncell + BEOM_BLOCK - 1) / BEOM_BLOCK);
    if (eta || u4 || v4) {
        hipLaunchKernelGGL(k_out_convert, dim3(gx), dim3(BEOM_BLOCK), 0, E->stream, d, E->h0r4_dev,
                           eta ? E->out4[0] : nullptr, u4 ? E->out4[1] : nullptr, v4 ? E->out4[2] : nullptr);
        float *host[3] = {eta, u4, v4};
        for (int q = 0; q < 3; ++q)
            if (host[q]) HIP_TRY(hipMemcpyAsync(host[q], E->out4[q], n * sizeof(float), hipMemcpyDeviceToHost, E->stream));
    }
    if (minmax || thin_layer) {
        const size_t nb = (size_t)E->grid_cells0.x;
        hipLaunchKernelGGL(k_out_scan, E->grid_cells0, dim3(BEOM_BLOCK), 0, E->stream, d, E->any_u, E->any_v, E->scan_dev);
        std::vector<double> part(nb * d.nlay * 7);
        HIP_TRY(hipMemcpyAsync(part.data(), E->scan_dev, part.size() * sizeof(double), hipMemcpyDeviceToHost, E->stream));
        HIP_TRY(hipStreamSynchronize(E->stream));
        if (thin_layer) *thin_layer = 0;
        for (int k = 0; k < d.nlay; ++k) {
            double r[7] = {1.7976931348623157e308, -1.7976931348623157e308, 1.7976931348623157e308,
                           -1.7976931348623157e308, 1.7976931348623157e308, -1.7976931348623157e308, 0.0};
            for (size_t b = 0; b < nb; ++b) {
                const double *p = &part[(b * d.nlay + k) * 7];
                for (int q = 0; q < 7; ++q) r[q] = (q == 0 || q == 2 || q == 4) ? std::fmin(r[q], p[q]) : std::fmax(r[q], p[q]);
            }
            if (minmax) for (int q = 0; q < 6; ++q) minmax[k * 6 + q] = r[q];
            if (thin_layer && r[6] > 0.5 && *thin_layer == 0) *thin_layer = k + 1;
        }
    }
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// The three `diag` records of write_array (private_mod.f95:2884-2974) formed on the device: only real*4 crosses PCIe.
// Between time steps only (the work arrays are the step's d2hx / d2hy scratch).
int beom_download_diag(beom_handle E, float *pvor4, float *mont4, float *vcc4, char *errm, int errm_len) {
    if (!E) { set_err(errm, errm_len, "null handle"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    DevView &d = E->d;
    const size_t n = (size_t)d.ndeg * d.nlay;
    float *dst[3] = {pvor4, mont4, vcc4};
    if (!E->diag4[0])
        for (int q = 0; q < 3; ++q) { HIP_TRY(hipMalloc((void **)&E->diag4[q], n * sizeof(float))); E->allocs.push_back(E->diag4[q]); }
    const dim3 g = E->grid_cells_layers_flat, b(BEOM_BLOCK);
    // work arrays: scratch that is dead between steps — the curvatures, or with a lid (whose next step reads the stored
    // curvatures of the last layer, :2237-2257) the product arrays of the fused sweeps, which a lid handle never runs
    double *w1 = E->lid ? d.pcd : d.d2hx, *w2 = E->lid ? d.qlr : d.d2hy;
    if (vcc4) hipLaunchKernelGGL(k_diag_w12, g, b, 0, E->stream, d, w1, w2);
    hipLaunchKernelGGL(k_diag_records, g, b, 0, E->stream, d, (const double *)w1, (const double *)w2,
                       pvor4 ? E->diag4[0] : nullptr, mont4 ? E->diag4[1] : nullptr, vcc4 ? E->diag4[2] : nullptr);
    for (int q = 0; q < 3; ++q)
        if (dst[q]) HIP_TRY(hipMemcpyAsync(dst[q], E->diag4[q], n * sizeof(float), hipMemcpyDeviceToHost, E->stream));
    HIP_TRY(hipStreamSynchronize(E->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// Replaces index_boundary_points' product (private_mod.f95:1060-1240): the table segm(nseg, 18)
// of nudged open-boundary segments, Fortran storage.  Activates no_gradient_obc after the
// momentum sweeps of every step when flag_nudging and mcbc < 0.5 (:2201-2204, 2285-2288).
int beom_set_open_boundaries(beom_handle E, int nseg, const int32_t *segm, char *errm, int errm_len) {
    if (!E || nseg < 0 || (nseg > 0 && !segm)) { set_err(errm, errm_len, "beom_set_open_boundaries: bad arguments"); return -1; }
    HIP_TRY(hipSetDevice(E->device));
    if (nseg == 0) {                    // (a band of a frame whose segments all lie in other bands)
        E->d.segm = nullptr; E->d.nseg = 0; E->obc = false; E->obc_set = true;
        return 0;
    }
    auto S = [&](int is, int col) { return segm[(size_t)is + (size_t)nseg * (col - 1)]; };
    // The reference loops are serial.  A parallel pass is equivalent iff nothing it writes is read
    // or written by another segment of the same pass: check (component, cell) sets.
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<long long> wr, rd;
        for (int is = 0; is < nseg; ++is) {
            const bool ns = S(is, 5) == 1, ew = S(is, 4) == 1;
            const int comp = pass == 0 ? (ns ? 0 : (ew ? 1 : -1)) : (ns ? 1 : (ew ? 0 : -1));
            if (comp < 0) continue;
            const int ip = pass == 0 ? S(is, 10) : S(is, 1), in = pass == 0 ? S(is, 16) : S(is, 13);
            if (ip == -1) continue;      // (this pass of the segment belongs to another band: beom_multi_set_open_boundaries)
            if (ip < 1 || ip > E->d.ndeg || in < 0 || in > E->d.ndeg) { set_err(errm, errm_len, "beom_set_open_boundaries: index out of range"); return -3; }
            wr.push_back(2ll * ip + comp);
            rd.push_back(2ll * in + comp);
        }
        std::vector<long long> w2 = wr;
        std::sort(w2.begin(), w2.end());
        if (std::adjacent_find(w2.begin(), w2.end()) != w2.end()) { set_err(errm, errm_len, "beom_set_open_boundaries: two segments update the same point (serial order would matter)"); return -7; }
        for (long long r : rd)
            if (std::binary_search(w2.begin(), w2.end(), r)) { set_err(errm, errm_len, "beom_set_open_boundaries: a segment reads a point another segment updates in the same pass (serial order would matter)"); return -7; }
    }
    // columns 1, 7, 10, 13, 16 are cell indices: to the device pitch
    std::vector<int32_t> sg(segm, segm + (size_t)nseg * 18);
    if (E->embedded) {                   // frames with land on the rectangle: packed index -> slot
        for (int col : {1, 7, 10, 13, 16})
            for (int is = 0; is < nseg; ++is) {
                int32_t &q = sg[(size_t)is + (size_t)nseg * (col - 1)];
                if (q > 0) { if ((size_t)q >= E->dev_index.size()) { set_err(errm, errm_len, "beom_set_open_boundaries: index out of range"); return -3; } q = E->dev_index[(size_t)q]; }
            }
    } else if (E->d.P)
        for (int col : {1, 7, 10, 13, 16})
            for (int is = 0; is < nseg; ++is) {
                int32_t &q = sg[(size_t)is + (size_t)nseg * (col - 1)];
                if (q > 0) q = (int32_t)(1 + (long long)((q - 1) / E->d.L) * E->d.P + (q - 1) % E->d.L);
            }
    int32_t *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, (size_t)nseg * 18 * sizeof(int32_t)));
    E->allocs.push_back(dev);
    HIP_TRY(hipMemcpy(dev, sg.data(), (size_t)nseg * 18 * sizeof(int32_t), hipMemcpyHostToDevice));
    E->d.segm = dev;
    E->d.nseg = nseg;
    E->obc = E->P.flag_nudging && E->P.mcbc < 0.5;
    E->obc_set = true;
    return 0;
}

int beom_info(beom_handle E, const char *what) {
    if (!E || !what) return -1;
    if (!strcmp(what, "stress_folded")) return E->last_folded ? 1 : 0;
    if (!strcmp(what, "tile_rows")) return E->dense ? (E->tile4 ? 4 : 8) : 0;
    if (!strcmp(what, "lid_sweeps")) return (int)std::min<long long>(E->lid_sweeps, 2000000000ll);        // Gauss-Seidel sweeps kept, all steps so far
    if (!strcmp(what, "lid_solves")) return (int)std::min<long long>(E->lid_solves, 2000000000ll);
    if (!strcmp(what, "lid_launches")) return (int)std::min<long long>(E->lid_launches, 2000000000ll);
    if (!strcmp(what, "lid_sweep_distance")) return E->lid_dstep;
    return -3;
}

int beom_set_option(beom_handle E, const char *name, int value) {
    if (!E || !name) return -1;
    if (!strcmp(name, "fuse")) { E->fuse = value != 0; E->fuse_uv = value != 0; }
    else if (!strcmp(name, "fuse_mont_visc")) E->fuse = value != 0 && !E->lid;      // (a lid handle keeps the separate sweeps)
    else if (!strcmp(name, "fuse_uv")) E->fuse_uv = value != 0 && !E->lid;
    else if (!strcmp(name, "keep_diag")) E->d.keep_diag = value != 0;
    else if (!strcmp(name, "fold_stress")) E->fold_stress = value != 0;
    else if (!strcmp(name, "profile_stride")) E->profile_stride = value > 0 ? value : 1;
    else if (!strcmp(name, "profile_rotate")) E->profile_rotate = value != 0;
    else if (!strcmp(name, "lean_d2h")) E->lean_d2h = value != 0;
    else if (!strcmp(name, "lean_visc")) E->lean_visc = value != 0;
    else return -3;
    return 0;
}

int beom_set_stream(beom_handle E, void *hip_stream, int use_own) {
    if (!E) return -1;
    // hip_stream == NULL is a valid stream (the legacy default stream torch uses by default)
    E->stream = use_own ? E->own_stream : (hipStream_t)hip_stream;
    return 0;
}

int beom_is_dense(beom_handle E) { return (E && E->dense) ? (E->embedded ? 2 : 1) : 0; }

int beom_device_field(beom_handle E, const char *name, void **dptr, int64_t *stride_layer,
                      int64_t *stride_row, int64_t *row0_offset) {
    if (!E || !name || !dptr) return -1;
    const DevView &d = E->d;
    double *p = nullptr;
    if (!strcmp(name, "hlay")) p = d.hlay;
    else if (!strcmp(name, "u")) p = d.u;
    else if (!strcmp(name, "v")) p = d.v;
    else if (!strcmp(name, "h_u")) p = d.h_u;
    else if (!strcmp(name, "h_v")) p = d.h_v;
    else return -3;
    *dptr = p;
    if (stride_layer) *stride_layer = d.n1;
    if (stride_row) *stride_row = d.P ? d.P : (E->dense ? d.L : 0);
    if (row0_offset) *row0_offset = 1;
    return 0;
}

}  // extern "C"
