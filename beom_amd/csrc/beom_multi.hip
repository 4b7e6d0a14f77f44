// beom_multi.hip — the j-slab decomposition of SURVEY.md §8(e) behind the C-ABI: a dense frame cut
// into bands of rows, one band per GPU, ghost rows exchanged once per time step, the exchange of
// step n overlapped with the interior rows of step n+1.  The reference has no counterpart (OpenMP
// only).  Built only from the public entry points of include/beom_hip.h.
//
//   * every band is an ordinary slab handle (beom_params.slab_row0/slab_mm) with G = 4 ghost rows per
//     neighbour; per step ONE exchange of hlay,u,v,h_u,h_v (beom_pack_rows -> transport ->
//     beom_unpack_rows on the band's second stream), beom_step_phase 1/2 around it;
//   * transports: peer copies between the bands of ONE process (hipMemcpyPeerAsync over xGMI), or RCCL
//     (grouped ncclSend/ncclRecv; librccl is loaded at run time) — either all bands in one process
//     (ncclCommInitAll) or ONE band per process (ncclCommInitRank: bench.py under torchrun);
//   * two ways in: GLOBAL arrays that the library cuts (beom_multi_create: the Fortran host), or this
//     band's WINDOW only (beom_multi_create_local: nothing of global size exists on any rank);
//   * frames periodic in y: the bands form a ring over rows 1..mm (a band of a ring is a slab deep inside
//     a taller fake frame: every mask is 1, the wrap comes from the exchange).  The orphan row mm+1
//     keeps its slot in every array and output record (private_mod.f95:642-668: its E/W neighbours are
//     cells of row 1, its S neighbours cells of row mm, nothing points to it) and is carried by a
//     companion frame on band 0's device: rows 1..6, mm-3..mm and mm+1 as ONE small y-periodic frame,
//     its first ten rows refreshed from band 0 before every step.
// The K steps of one call run inside the library: no per-step host language in the loop.
// No CPU fallback: every call needs its HIP devices.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <string>
#include <vector>

#include "../../include/beom_hip.h"
#include "beom_dense_host.h"

namespace {

constexpr int kGhost = 4;          // rows per neighbour; see DESIGN.md §5 for why 4 is enough
constexpr int kFields = 5;         // hlay, u, v, h_u, h_v
constexpr int kMiniLo = 6;         // rows 1..6 of a y-periodic frame that the companion frame carries
constexpr int kFakePad = 8;        // a ring band is presented as rows 9.. of a frame 16 rows taller

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

// ---- RCCL, bound at run time (one copy per process: PyTorch-ROCm ships librccl.so.1 too) --------
typedef void *nccl_comm;
struct nccl_uid { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    int (*CommInitAll)(nccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    bool load(char *errm, int errm_len) {
        if (lib) return true;
        if (getenv("BEOM_RCCL_DISABLE")) { m_err(errm, errm_len, "RCCL switched off by BEOM_RCCL_DISABLE (rehearsal of a machine without it)"); return false; }
        const char *names[] = {getenv("BEOM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { m_err(errm, errm_len, "RCCL transport: librccl.so.1 not found (%s)", dlerror()); return false; }
#define SYM(field, name) do { *(void **)(&field) = dlsym(lib, name); if (!field) { m_err(errm, errm_len, "RCCL transport: %s missing in librccl", name); lib = nullptr; return false; } } while (0)
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
        SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString"); SYM(GetVersion, "ncclGetVersion");
#undef SYM
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclChar = 0;       // ncclInt8 / ncclChar (rccl.h): the rows travel as bytes

#define M_NCCL(expr)                                                                         \
    do {                                                                                     \
        int e_ = (expr);                                                                     \
        if (e_ != 0) {                                                                       \
            m_err(errm, errm_len, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(e_), \
                  __FILE__, __LINE__);                                                       \
            return -300 - e_;                                                                \
        }                                                                                    \
    } while (0)

// ---- ghost rows staged through a POSIX shared-memory segment (BEOM_XCHG_SHM) ---------------------------
// One band per process, the processes on ONE node — several of them may share a device, which RCCL refuses
// ("duplicate GPU"): the -m gpu tests run two and three ranks of beom_multi_create_local on the one GPU of
// the box this way, through the same multi_one_step as the RCCL branch.  In stream order on a band's second
// stream, where RCCL has its grouped send/recv:
//     copy both send buffers into this band's slots of the segment -> host function: publish "exchange s is there"
//     host function: wait for the north neighbour's s -> copy its south-going slot into recv_n; the same for the south
// Slots are double-buffered by the parity of s; a band reaches exchange s + 2 only after it has seen its neighbours'
// s + 1, and a neighbour publishes s + 1 only after its step has consumed this band's s — so no slot is overwritten
// before it has been read.  A wait that lasts longer than BEOM_SHM_TIMEOUT_S (default 60) marks the handle failed
// instead of hanging the stream.
struct ShmCtl { volatile uint64_t ready; volatile uint64_t seq; char pad[112]; };      // one per band, 128 B apart
struct ShmHdr { uint64_t magic; uint64_t nb; uint64_t xbytes; uint64_t slot_stride; };
constexpr uint64_t kShmMagic = 0x42454f4d58434847ull;            // "BEOMXCHG"
constexpr size_t kShmCtl0 = 4096;

struct ShmXchg {
    std::string name;
    char *base = nullptr;
    size_t size = 0, slot_stride = 0, data0 = 0;
    int nb = 0;
    bool registered = false;
    uint64_t seq = 0;                       // exchanges issued by this band so far
    volatile int failed = 0;                // a wait timed out (set from a host function)
    double timeout_s = 60.0;
    ShmCtl *ctl(int band) const { return (ShmCtl *)(base + kShmCtl0 + (size_t)band * sizeof(ShmCtl)); }
    // what band `band` sends towards dir (0 = south, 1 = north) in exchange s
    char *slot(int band, int dir, uint64_t s) const { return base + data0 + (((size_t)band * 2 + dir) * 2 + (size_t)(s & 1)) * slot_stride; }
};

static double mono_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void nap_us(long us) { timespec ts{0, us * 1000}; nanosleep(&ts, nullptr); }

struct ShmOp { ShmXchg *x; int band; uint64_t seq; };
// host functions (hipLaunchHostFunc): no HIP call inside
static void shm_publish(void *p) {
    ShmOp *op = (ShmOp *)p;
    __atomic_store_n(&op->x->ctl(op->band)->seq, op->seq, __ATOMIC_RELEASE);
    delete op;
}
static void shm_wait(void *p) {
    ShmOp *op = (ShmOp *)p;
    const double t0 = mono_s();
    int spins = 0;
    while (__atomic_load_n(&op->x->ctl(op->band)->seq, __ATOMIC_ACQUIRE) < op->seq) {
        if (op->x->failed) break;
        if (++spins > 2000) nap_us(20);
        if ((spins & 1023) == 0 && mono_s() - t0 > op->x->timeout_s) { op->x->failed = 1; break; }
    }
    delete op;
}

static void shm_close(ShmXchg *x) {
    if (!x) return;
    if (x->base) {
        if (x->registered) (void)hipHostUnregister(x->base);
        munmap(x->base, x->size);
    }
    delete x;
}

// every band maps the segment `name` (created by whoever comes first), announces itself and waits for the others;
// band 0 then removes the name, so nothing is left in /dev/shm once the last process has gone
static int shm_open_all(ShmXchg **out, const char *name, int nb, int band, size_t xbytes, bool loopback, char *errm, int errm_len) {
    if (!name || name[0] != '/' || strlen(name) > 200) { m_err(errm, errm_len, "beom_multi: the shared-memory transport needs a segment name \"/...\" common to all ranks"); return -3; }
    ShmXchg *x = new ShmXchg();
    x->name = name; x->nb = nb;
    x->slot_stride = (xbytes + 4095) / 4096 * 4096;
    x->data0 = (kShmCtl0 + (size_t)nb * sizeof(ShmCtl) + 4095) / 4096 * 4096;
    x->size = x->data0 + (size_t)nb * 4 * x->slot_stride;
    if (const char *t = getenv("BEOM_SHM_TIMEOUT_S")) { const double v = atof(t); if (v > 0.0) x->timeout_s = v; }
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { m_err(errm, errm_len, "beom_multi: shm_open(%s) failed: %s", name, strerror(errno)); delete x; return -33; }
    if (ftruncate(fd, (off_t)x->size) != 0) { m_err(errm, errm_len, "beom_multi: ftruncate(%s, %zu) failed: %s", name, x->size, strerror(errno)); close(fd); delete x; return -33; }
    void *m = mmap(nullptr, x->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { m_err(errm, errm_len, "beom_multi: mmap(%s) failed: %s", name, strerror(errno)); delete x; return -33; }
    x->base = (char *)m;
    ShmHdr *h = (ShmHdr *)x->base;
    if (band == 0 || loopback) { h->nb = (uint64_t)nb; h->xbytes = xbytes; h->slot_stride = x->slot_stride; __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE); }
    __atomic_store_n(&x->ctl(band)->ready, (uint64_t)1, __ATOMIC_RELEASE);
    const double t0 = mono_s();
    for (;;) {
        bool all = __atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) == kShmMagic;
        for (int b = 0; b < nb && all && !loopback; ++b) all = __atomic_load_n(&x->ctl(b)->ready, __ATOMIC_ACQUIRE) != 0;
        if (all) break;
        if (mono_s() - t0 > 2.0 * x->timeout_s) {
            m_err(errm, errm_len, "beom_multi: band %d waited %.0f s for the other %d bands at segment %s", band, 2.0 * x->timeout_s, nb - 1, name);
            shm_unlink(name); shm_close(x); return -34;
        }
        nap_us(200);
    }
    if (h->nb != (uint64_t)nb || h->xbytes != xbytes) {
        m_err(errm, errm_len, "beom_multi: segment %s belongs to another frame (bands %llu, bytes %llu)", name, (unsigned long long)h->nb, (unsigned long long)h->xbytes);
        shm_close(x); return -34;
    }
    if (band == 0 || loopback) shm_unlink(name);
    // pinned: the copies to and from the segment are then asynchronous like any other (without it they still work, staged)
    if (hipHostRegister(x->base, x->size, hipHostRegisterDefault) == hipSuccess) x->registered = true;
    else (void)hipGetLastError();
    *out = x;
    return 0;
}

// ---- geometry ---------------------------------------------------------------------------------------
struct Band {
    int index = 0;                 // position in the chain / ring of nb bands
    int own0 = 0, own1 = 0;        // owned global rows (1-based, inclusive)
    int gs = 0, gn = 0;            // ghost rows on the south / north side
    int L = 0;                     // columns = lm + 1
    int Mr = 0;                    // rows of the ring (frames periodic in y), else 0
    std::vector<long long> lst;    // frames with land: first local packed cell of local row j = 1..rows()+1 (rows differ in length)
    int nown() const { return own1 - own0 + 1; }
    int rows() const { return gs + nown() + gn; }
    long long n_loc() const { return lst.empty() ? (long long)rows() * L : lst[(size_t)rows() + 1] - 1; }
    int grow(int j) const {        // global row of local row j (ghosts of a ring wrap)
        int g = own0 - gs + (j - 1);
        if (Mr) { while (g < 1) g += Mr; while (g > Mr) g -= Mr; }
        return g;
    }
    std::vector<int> row_list() const { std::vector<int> r; for (int j = 1; j <= rows(); ++j) r.push_back(grow(j)); return r; }
};

// shapes of the caller's arrays around the packed index: x[outer][0:n][inner]
struct Shape { int outer_nl, outer_c, inner; size_t outer(int nl) const { return (size_t)outer_nl * nl + outer_c; } };
//                         fcor     h_th     h_to     nudg     fnud     hdot     tide     taus
const Shape kStatic[8] = {{0,1,1}, {0,1,1}, {0,1,1}, {0,3,1}, {3,0,1}, {1,0,1}, {0,3,2}, {0,2,1}};
//                        hlay     u        v        h_u      h_v      rs_h     dmdx     dmdy     v_cc     v_ll     tt3d     tb3d     tu3d
const Shape kState[13] = {{1,0,1}, {1,0,1}, {1,0,1}, {1,0,1}, {1,0,1}, {1,0,2}, {1,0,3}, {1,0,3}, {1,0,1}, {1,0,1}, {2,0,1}, {2,0,1}, {2,0,1}};

// [outer][0:n1src][inner] -> [outer][0:rows*L][inner]: sentinel first, then the listed rows of the source
template <class T>
std::vector<T> cut(const T *x, size_t outer, size_t inner, size_t n1src, const std::vector<int> &rows, int L) {
    std::vector<T> z;
    if (!x) return z;
    const size_t n1l = rows.size() * (size_t)L + 1;
    z.resize(outer * n1l * inner);
    for (size_t o = 0; o < outer; ++o) {
        std::memcpy(&z[o * n1l * inner], &x[o * n1src * inner], inner * sizeof(T));
        for (size_t r = 0; r < rows.size(); ++r)
            std::memcpy(&z[(o * n1l + 1 + r * L) * inner], &x[(o * n1src + 1 + (size_t)(rows[r] - 1) * L) * inner],
                        (size_t)L * inner * sizeof(T));
    }
    return z;
}
// local rows [j0, j0+n) of a local [outer][0:n1l][inner] array -> rows g(j) of a [outer][0:n1dst][inner] array
template <class T>
void paste(T *dst, const std::vector<T> &loc, size_t outer, size_t inner, size_t n1dst, size_t n1l, int L,
           int j0, int n, const std::vector<int> &dst_rows, bool sentinel) {
    if (!dst || loc.empty()) return;
    for (size_t o = 0; o < outer; ++o) {
        if (sentinel) std::memcpy(&dst[o * n1dst * inner], &loc[o * n1l * inner], inner * sizeof(T));
        for (int r = 0; r < n; ++r)
            std::memcpy(&dst[(o * n1dst + 1 + (size_t)(dst_rows[r] - 1) * L) * inner],
                        &loc[(o * n1l + 1 + (size_t)(j0 - 1 + r) * L) * inner], (size_t)L * inner * sizeof(T));
    }
}
template <class T> const T *ptr(const std::vector<T> &v) { return v.empty() ? nullptr : v.data(); }
template <class T> T *ptr(std::vector<T> &v) { return v.empty() ? nullptr : v.data(); }

// ---- the same for frames WITH land: a row is the packed range [gst[j], gst[j+1]) of the caller's arrays (packing is
//      j-major), rows differ in length; lst = the starts of a band's rows in its own packed order ----
template <class T>
std::vector<T> cut_v(const T *x, size_t outer, size_t inner, size_t n1src, const std::vector<int> &rows, const std::vector<long long> &gst) {
    std::vector<T> z;
    if (!x) return z;
    size_t n1l = 1;
    for (int g : rows) n1l += (size_t)(gst[(size_t)g + 1] - gst[(size_t)g]);
    z.resize(outer * n1l * inner);
    for (size_t o = 0; o < outer; ++o) {
        std::memcpy(&z[o * n1l * inner], &x[o * n1src * inner], inner * sizeof(T));
        size_t at = 1;
        for (int g : rows) {
            const size_t len = (size_t)(gst[(size_t)g + 1] - gst[(size_t)g]);
            if (len) std::memcpy(&z[(o * n1l + at) * inner], &x[(o * n1src + (size_t)gst[(size_t)g]) * inner], len * inner * sizeof(T));
            at += len;
        }
    }
    return z;
}
template <class T>
void paste_v(T *dst, const std::vector<T> &loc, size_t outer, size_t inner, size_t n1dst, size_t n1l, const std::vector<long long> &lst,
             const std::vector<long long> &gst, int j0, int n, const std::vector<int> &dst_rows, bool sentinel) {
    if (!dst || loc.empty()) return;
    for (size_t o = 0; o < outer; ++o) {
        if (sentinel) std::memcpy(&dst[o * n1dst * inner], &loc[o * n1l * inner], inner * sizeof(T));
        for (int r = 0; r < n; ++r) {
            const size_t g = (size_t)dst_rows[(size_t)r], len = (size_t)(gst[g + 1] - gst[g]);
            if (len) std::memcpy(&dst[(o * n1dst + (size_t)gst[g]) * inner], &loc[(o * n1l + (size_t)lst[(size_t)(j0 + r)]) * inner], len * inner * sizeof(T));
        }
    }
}

struct StaticsV {                  // one band's (or the companion frame's) static arrays in window layout
    std::vector<double> a[8];
    const double *bodf = nullptr;
};
struct StateV { std::vector<double> a[13]; };

const double *const *statics_ptrs(const beom_statics *s, const double *(&p)[8]) {
    p[0] = s->fcor; p[1] = s->h_th; p[2] = s->h_to; p[3] = s->nudg; p[4] = s->fnud; p[5] = s->hdot; p[6] = s->tide; p[7] = s->taus;
    return p;
}
void state_ptrs(const beom_state *s, double *(&p)[13]) {
    p[0] = s->hlay; p[1] = s->u; p[2] = s->v; p[3] = s->h_u; p[4] = s->h_v; p[5] = s->rs_h; p[6] = s->dmdx; p[7] = s->dmdy;
    p[8] = s->v_cc; p[9] = s->v_ll; p[10] = s->tt3d; p[11] = s->tb3d; p[12] = s->tu3d;
}

}  // namespace

struct beom_multi {
    beom_params P{};               // global frame
    int nb = 0;                    // bands in the chain / ring, over all processes
    int n = 0;                     // bands of THIS process
    bool ring = false;             // frame periodic in y
    int xper = 0;
    int transport = BEOM_XCHG_PEER;
    bool local_mode = false;       // created from this band's window (beom_multi_create_local)
    bool failed = false;           // a step failed half way: the state is undefined, only destroy is allowed
    bool overlap = true;           // steps cut boundary first, the exchange inside the interior sweep (beom_multi_set_option "overlap")
    size_t n1g = 0;
    bool land = false;             // a frame with land: bands are packed row ranges of unequal length, on the rectangle ("embedded") form
    std::vector<long long> gst;    // land: first packed cell of every global row j = 1..mm+2
    std::vector<int> dev;
    std::vector<Band> band;
    std::vector<beom_handle> eng;
    std::vector<hipStream_t> main_s, comm_s;          // a band's sweeps | the edge strips of a cut step and its exchange
    std::vector<hipEvent_t> packed, landed, p1done;
    std::vector<char> pending;     // an exchange into this band is in flight
    std::vector<double *> send_s, recv_s, send_n, recv_n;   // device buffers on the band's device
    std::vector<nccl_comm> comm;
    ShmXchg *shm = nullptr;        // BEOM_XCHG_SHM: the segment shared by the bands' processes
    bool loopback = false;         // BEOM_XCHG_LOOPBACK: this band receives what it sends (timing rehearsals)
    size_t xbytes = 0;
    long long n_split = 0, n_plain = 0;   // band-steps taken in two phases / in one piece
    // companion frame of a y-periodic ring (lives with band 0): rows 1..kMiniLo, Mr-3..Mr, Mr+1
    beom_handle mini = nullptr;
    int mini_k = -1;               // local index of band 0, or -1 if band 0 is not here
    hipStream_t mini_s = nullptr;
    double *mini_lo = nullptr, *mini_hi = nullptr;
    hipEvent_t ev_hi = nullptr, ev_free = nullptr;
    bool free_recorded = false;
    std::vector<int> mini_rows;

    int local_of(int gidx) const { for (int k = 0; k < n; ++k) if (band[k].index == gidx) return k; return -1; }
    bool has_s(int k) const { return ring || band[k].index > 0; }
    bool has_n(int k) const { return ring || band[k].index < nb - 1; }
    int south_of(int k) const { return loopback ? band[k].index : (band[k].index - 1 + nb) % nb; }
    int north_of(int k) const { return loopback ? band[k].index : (band[k].index + 1) % nb; }
    int rank_of(int gidx) const { return loopback ? 0 : gidx; }        // RCCL rank of a band (a looped-back band is alone in its communicator)
};

namespace {

void destroy_all(beom_multi *M) {
    if (!M) return;
    for (int k = 0; k < M->n; ++k) {
        (void)hipSetDevice(M->dev[k]);
        if (k < (int)M->main_s.size() && M->main_s[k]) (void)hipStreamSynchronize(M->main_s[k]);
        if (k < (int)M->comm_s.size() && M->comm_s[k]) (void)hipStreamSynchronize(M->comm_s[k]);
    }
    if (M->mini_k >= 0) {
        (void)hipSetDevice(M->dev[M->mini_k]);
        if (M->mini_s) (void)hipStreamSynchronize(M->mini_s);
        if (M->mini) (void)beom_destroy(M->mini);
        if (M->mini_lo) (void)hipFree(M->mini_lo);
        if (M->mini_hi) (void)hipFree(M->mini_hi);
        for (hipEvent_t e : {M->ev_hi, M->ev_free}) if (e) (void)hipEventDestroy(e);
        if (M->mini_s) (void)hipStreamDestroy(M->mini_s);
    }
    if (M->shm) { if (M->n > 0) (void)hipSetDevice(M->dev[0]); shm_close(M->shm); M->shm = nullptr; }
    for (int k = 0; k < M->n; ++k) {
        (void)hipSetDevice(M->dev[k]);
        if (k < (int)M->comm.size() && M->comm[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(M->comm[k]);
        if (k < (int)M->eng.size() && M->eng[k]) (void)beom_destroy(M->eng[k]);
        for (auto *v : {&M->send_s, &M->recv_s, &M->send_n, &M->recv_n})
            if (k < (int)v->size() && (*v)[k]) (void)hipFree((*v)[k]);
        if (k < (int)M->packed.size() && M->packed[k]) (void)hipEventDestroy(M->packed[k]);
        if (k < (int)M->landed.size() && M->landed[k]) (void)hipEventDestroy(M->landed[k]);
        if (k < (int)M->p1done.size() && M->p1done[k]) (void)hipEventDestroy(M->p1done[k]);
        if (k < (int)M->comm_s.size() && M->comm_s[k]) (void)hipStreamDestroy(M->comm_s[k]);
        if (k < (int)M->main_s.size() && M->main_s[k]) (void)hipStreamDestroy(M->main_s[k]);
    }
    delete M;
}

// rows of the chain / ring dealt to nb bands: equal counts, remainders to the first bands
void deal_rows(int nrows_total, int nb, int idx, int *own0, int *own1) {
    const int base = nrows_total / nb, rem = nrows_total % nb;
    int j = 1;
    for (int k = 0; k < nb; ++k) {
        const int cnt = base + (k < rem ? 1 : 0);
        if (k == idx) { *own0 = j; *own1 = j + cnt - 1; }
        j += cnt;
    }
}

int check_frame(const beom_params *prm, int nb, int yper, bool global_arrays, bool land, char *errm, int errm_len) {
    const int L = prm->lm + 1, Mg = prm->mm + 1;
    if (prm->abi_version != BEOM_ABI_VERSION) { m_err(errm, errm_len, "beom_multi: ABI version mismatch"); return -2; }
    if (nb < 1 || nb > 64) { m_err(errm, errm_len, "beom_multi: bad band count %d", nb); return -3; }
    if ((!land && (long long)prm->ndeg != (long long)L * Mg) || prm->slab_mm != 0) {
        m_err(errm, errm_len, "beom_multi: the row decomposition needs a whole frame; one with land (ndeg < (lm+1)(mm+1)) only from the global arrays");
        return -3;
    }
    if (land && (yper || prm->svis > 0.0 || (prm->flag_nudging && prm->mcbc < 0.5) || prm->rgld > 0.5)) {
        m_err(errm, errm_len, "beom_multi: bands of a frame with land: not periodic in y, no biharmonic viscosity, no mcbc = 0, no rigid lid");
        return -4;
    }
    const int ring_rows = yper ? prm->mm : Mg;
    // every band sends its outermost kGhost owned rows; band 0 of a ring also lends rows 1..kMiniLo to the companion frame
    if (ring_rows < nb * (yper ? (kGhost > kMiniLo ? kGhost : kMiniLo) : kGhost + 1)) { m_err(errm, errm_len, "beom_multi: %d rows are too few for %d bands", ring_rows, nb); return -3; }
    (void)global_arrays;      // (mcbc = 0: a handle from global arrays gets the global segment table, beom_multi_set_open_boundaries; a
                              //  rank that holds only its window brings the segments of its own rows, ..._local; beom_step refuses until then)
    if (yper && prm->svis > 0.0) { m_err(errm, errm_len, "beom_multi: biharmonic viscosity on a frame periodic in y runs on a single-device handle only"); return -4; }
    return 0;
}

// geometry of band idx; a single band of a non-periodic frame is the whole frame (no slab at all)
Band make_band(const beom_params *prm, int nb, int idx, bool ring) {
    Band s;
    s.index = idx; s.L = prm->lm + 1; s.Mr = ring ? prm->mm : 0;
    deal_rows(ring ? prm->mm : prm->mm + 1, nb, idx, &s.own0, &s.own1);
    s.gs = (ring || idx > 0) ? kGhost : 0;
    s.gn = (ring || idx < nb - 1) ? kGhost : 0;
    return s;
}

int finish_band(beom_multi *M, int k, char *errm, int errm_len);

// one band's engine from its window statics (tables come from the closed form)
int create_band(beom_multi *M, int k, const StaticsV &st, char *errm, int errm_len) {
    const Band &s = M->band[k];
    beom_params lp = M->P;
    lp.mm = s.rows() - 1; lp.ndeg = (int32_t)s.n_loc();
    lp.dense_hint = 1;
    int joff = 0, Mg = s.rows(), slab = 0;
    if (M->ring) { joff = kFakePad; Mg = s.rows() + 2 * kFakePad; slab = 1; }                 // deep inside a taller frame
    else if (M->nb > 1) { joff = s.own0 - s.gs - 1; Mg = M->P.mm + 1; slab = 1; }
    lp.slab_row0 = slab ? joff : 0; lp.slab_mm = slab ? Mg - 1 : 0;
    const beom_dense::Tables t = beom_dense::generate(s.L, s.rows(), joff, Mg, slab, M->xper, 0);
    int rc = beom_create(&lp, M->dev[k], t.neig.data(), t.subc.data(), t.mk_u.data(), t.mk_v.data(), t.mk_n.data(),
                         t.mkpe.data(), t.mkpi.data(), ptr(st.a[0]), ptr(st.a[1]), ptr(st.a[2]), ptr(st.a[3]), ptr(st.a[4]),
                         ptr(st.a[5]), ptr(st.a[6]), st.bodf, ptr(st.a[7]), &M->eng[k], errm, errm_len);
    if (rc) return rc;
    if (!beom_is_dense(M->eng[k])) { m_err(errm, errm_len, "beom_multi: band %d did not qualify for the dense path", s.index); return -4; }
    return finish_band(M, k, errm, errm_len);
}

// Band k of a frame WITH land: the rows' packed cells with the caller's own tables, re-indexed to the window (links that
// leave the window become the sentinel: they start in the outermost ghost row, whose values nobody uses).  The engine
// lays such a band out on its rectangle ("embedded", beom_engine.hip) — that is what the ghost-row copies rely on.
int create_band_land(beom_multi *M, int k, const int32_t *neig, const int32_t *subc, const double *const *masks, const StaticsV &st,
                     char *errm, int errm_len) {
    const Band &s = M->band[k];
    const size_t n1g = M->n1g, nloc = (size_t)s.n_loc(), n1l = nloc + 1;
    const int row0 = s.own0 - s.gs;                               // global row of local row 1
    const std::vector<int> rows = s.row_list();
    auto to_local = [&](int32_t p) -> int32_t {
        if (p <= 0) return 0;
        const int jl = subc[(size_t)p + n1g] - row0 + 1;
        if (jl < 1 || jl > s.rows()) return 0;
        return (int32_t)(s.lst[(size_t)jl] + ((long long)p - M->gst[(size_t)(row0 + jl - 1)]));
    };
    std::vector<int32_t> ln(8 * n1l, 0), lsub(2 * n1l, 0);
    size_t q = 1;
    for (int g : rows)
        for (long long p = M->gst[(size_t)g]; p < M->gst[(size_t)g + 1]; ++p, ++q) {
            for (int c = 0; c < 8; ++c) ln[8 * q + c] = to_local(neig[8 * (size_t)p + c]);
            lsub[q] = subc[(size_t)p];
            lsub[q + n1l] = subc[(size_t)p + n1g] - row0 + 1;
        }
    std::vector<double> lm[5];
    for (int f = 0; f < 5; ++f) lm[f] = cut_v(masks[f], 1, 1, n1g, rows, M->gst);
    beom_params lp = M->P;
    lp.mm = s.rows() - 1; lp.ndeg = (int32_t)nloc; lp.dense_hint = 1;
    lp.slab_row0 = row0 - 1; lp.slab_mm = M->P.mm;              // a window of the global frame: the engine splits its steps around the exchange
    int rc = beom_create(&lp, M->dev[k], ln.data(), lsub.data(), lm[0].data(), lm[1].data(), lm[2].data(), lm[3].data(), lm[4].data(),
                         ptr(st.a[0]), ptr(st.a[1]), ptr(st.a[2]), ptr(st.a[3]), ptr(st.a[4]), ptr(st.a[5]), ptr(st.a[6]), st.bodf,
                         ptr(st.a[7]), &M->eng[k], errm, errm_len);
    if (rc) return rc;
    if (beom_is_dense(M->eng[k]) < 1) {           // (2: the rectangle form with land; 1: no land at all in this window)
        m_err(errm, errm_len, "beom_multi: band %d (rows %d..%d) does not fit the rectangle form (fewer than 30 %% of its cells wet, or a coast on a periodic seam)",
              s.index, s.own0, s.own1);
        return -4;
    }
    return finish_band(M, k, errm, errm_len);
}

int finish_band(beom_multi *M, int k, char *errm, int errm_len) {
    M_HIP(hipSetDevice(M->dev[k]));
    M_HIP(hipStreamCreateWithFlags(&M->main_s[k], hipStreamNonBlocking));
    {   // the second stream at the highest priority the device offers: its kernels (the edge strips of the momentum sweep, the
        // packing, RCCL's send/recv) are few workgroups that must get through while the interior sweep fills the chip
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            M_HIP(hipStreamCreateWithPriority(&M->comm_s[k], hipStreamNonBlocking, greatest));
        else { (void)hipGetLastError(); M_HIP(hipStreamCreateWithFlags(&M->comm_s[k], hipStreamNonBlocking)); }
    }
    M_HIP(hipEventCreateWithFlags(&M->packed[k], hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&M->landed[k], hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&M->p1done[k], hipEventDisableTiming));
    if (M->has_s(k)) { M_HIP(hipMalloc((void **)&M->send_s[k], M->xbytes)); M_HIP(hipMalloc((void **)&M->recv_s[k], M->xbytes)); }
    if (M->has_n(k)) { M_HIP(hipMalloc((void **)&M->send_n[k], M->xbytes)); M_HIP(hipMalloc((void **)&M->recv_n[k], M->xbytes)); }
    (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0);
    return 0;
}

// the companion frame of a ring: a y-periodic frame of kMiniLo + kGhost + 1 rows on band 0's device
int create_mini(beom_multi *M, const StaticsV &st, char *errm, int errm_len) {
    const int k = M->mini_k, L = M->P.lm + 1, Mm = kMiniLo + kGhost + 1;
    beom_params lp = M->P;
    lp.mm = Mm - 1; lp.ndeg = Mm * L; lp.dense_hint = 1; lp.slab_row0 = 0; lp.slab_mm = 0;
    const beom_dense::Tables t = beom_dense::generate(L, Mm, 0, Mm, 0, M->xper, 1);
    M_RC(beom_create(&lp, M->dev[k], t.neig.data(), t.subc.data(), t.mk_u.data(), t.mk_v.data(), t.mk_n.data(),
                     t.mkpe.data(), t.mkpi.data(), ptr(st.a[0]), ptr(st.a[1]), ptr(st.a[2]), ptr(st.a[3]), ptr(st.a[4]),
                     ptr(st.a[5]), ptr(st.a[6]), st.bodf, ptr(st.a[7]), &M->mini, errm, errm_len));
    if (!beom_is_dense(M->mini)) { m_err(errm, errm_len, "beom_multi: the companion frame did not qualify for the dense path"); return -4; }
    M_HIP(hipSetDevice(M->dev[k]));
    M_HIP(hipStreamCreateWithFlags(&M->mini_s, hipStreamNonBlocking));
    for (hipEvent_t *e : {&M->ev_hi, &M->ev_free}) M_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    const size_t row = (size_t)kFields * M->P.nlay * L * sizeof(double);
    M_HIP(hipMalloc((void **)&M->mini_lo, row * kMiniLo));
    M_HIP(hipMalloc((void **)&M->mini_hi, row * kGhost));
    (void)beom_set_stream(M->mini, (void *)M->mini_s, 0);
    return 0;
}

int init_transport(beom_multi *M, const void *rccl_id, char *errm, int errm_len) {
    const bool all_local = M->n == M->nb;
    if (M->transport == BEOM_XCHG_PEER) {
        if (!all_local) { m_err(errm, errm_len, "beom_multi: peer copies need all bands in one process; use BEOM_XCHG_RCCL"); return -3; }
        for (int k = 0; k < M->n; ++k) {           // direct peer copies over xGMI where the devices allow it (already-enabled is fine)
            M_HIP(hipSetDevice(M->dev[k]));
            for (int q : {M->south_of(k), M->north_of(k)}) {
                const int ql = M->local_of(q);
                if (ql < 0 || M->dev[ql] == M->dev[k]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, M->dev[k], M->dev[ql]) == hipSuccess && can) {
                    hipError_t pe = hipDeviceEnablePeerAccess(M->dev[ql], 0);
                    if (pe != hipSuccess) (void)hipGetLastError();
                }
            }
        }
        return 0;
    }
    if (M->transport == BEOM_XCHG_SHM) {
        if (M->n != 1) { m_err(errm, errm_len, "beom_multi: the shared-memory transport carries one band per process"); return -3; }
        M_HIP(hipSetDevice(M->dev[0]));
        return shm_open_all(&M->shm, (const char *)rccl_id, M->nb, M->band[0].index, M->xbytes, M->loopback, errm, errm_len);
    }
    if (M->transport != BEOM_XCHG_RCCL) { m_err(errm, errm_len, "beom_multi: unknown transport %d", M->transport); return -3; }
    if (!(all_local || M->n == 1)) { m_err(errm, errm_len, "beom_multi: RCCL transport: all bands in one process, or one band per process"); return -3; }
    if (!g_rccl.load(errm, errm_len)) return -31;
    M->comm.assign(M->n, nullptr);
    if (all_local && !rccl_id) {
        for (int k = 0; k < M->n; ++k)
            for (int q = 0; q < k; ++q)
                if (M->dev[k] == M->dev[q]) { m_err(errm, errm_len, "beom_multi: RCCL needs one distinct device per band (device %d named twice)", M->dev[k]); return -3; }
        M_NCCL(g_rccl.CommInitAll(M->comm.data(), M->n, M->dev.data()));
    } else {
        if (!rccl_id) { m_err(errm, errm_len, "beom_multi: RCCL transport across processes needs the unique id of beom_rccl_unique_id"); return -3; }
        nccl_uid id;
        std::memcpy(&id, rccl_id, sizeof(id));
        M_HIP(hipSetDevice(M->dev[0]));
        M_NCCL(g_rccl.CommInitRank(&M->comm[0], M->loopback ? 1 : M->nb, id, M->rank_of(M->band[0].index)));
    }
    return 0;
}

void size_vectors(beom_multi *M) {
    const int n = M->n;
    M->eng.assign(n, nullptr);
    M->main_s.assign(n, nullptr); M->comm_s.assign(n, nullptr);
    M->packed.assign(n, nullptr); M->landed.assign(n, nullptr); M->p1done.assign(n, nullptr);
    M->pending.assign(n, 0);
    M->send_s.assign(n, nullptr); M->recv_s.assign(n, nullptr);
    M->send_n.assign(n, nullptr); M->recv_n.assign(n, nullptr);
    M->xbytes = (size_t)kFields * M->P.nlay * kGhost * (M->P.lm + 1) * sizeof(double);
}

std::vector<int> mini_row_list(int Mr) {
    std::vector<int> r;
    for (int j = 1; j <= kMiniLo; ++j) r.push_back(j);
    for (int j = Mr - kGhost + 1; j <= Mr; ++j) r.push_back(j);
    r.push_back(Mr + 1);
    return r;
}

// window layout of band 0 of a ring + the orphan row -> the companion frame's layout
// win: [outer][0:(rows*L)][inner] with local rows 1..gs = ring rows Mr-gs+1..Mr, then rows 1..; orph: [outer][0:L][inner]
std::vector<double> mini_from_window(const double *win, const double *orph, size_t outer, size_t inner, const Band &b0) {
    std::vector<double> z;
    if (!win) return z;
    const int L = b0.L, Mm = kMiniLo + kGhost + 1;
    const size_t n1w = (size_t)b0.n_loc() + 1, n1m = (size_t)Mm * L + 1, n1o = (size_t)L + 1;
    z.assign(outer * n1m * inner, 0.0);
    for (size_t o = 0; o < outer; ++o) {
        std::memcpy(&z[o * n1m * inner], &win[o * n1w * inner], inner * sizeof(double));
        // rows 1..kMiniLo = band 0's first owned rows (local rows gs+1..)
        std::memcpy(&z[(o * n1m + 1) * inner], &win[(o * n1w + 1 + (size_t)b0.gs * L) * inner], (size_t)kMiniLo * L * inner * sizeof(double));
        // rows Mr-3..Mr = band 0's south ghosts (local rows 1..gs)
        std::memcpy(&z[(o * n1m + 1 + (size_t)kMiniLo * L) * inner], &win[(o * n1w + 1) * inner], (size_t)kGhost * L * inner * sizeof(double));
        if (orph) std::memcpy(&z[(o * n1m + 1 + (size_t)(kMiniLo + kGhost) * L) * inner], &orph[(o * n1o + 1) * inner], (size_t)L * inner * sizeof(double));
    }
    return z;
}

}  // namespace

extern "C" {

int beom_rccl_unique_id(void *id128, char *errm, int errm_len) {
    if (!id128) { m_err(errm, errm_len, "beom_rccl_unique_id: null argument"); return -1; }
    if (!g_rccl.load(errm, errm_len)) return -31;
    nccl_uid id;
    M_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int beom_rccl_version(char *errm, int errm_len) {
    if (!g_rccl.load(errm, errm_len)) return -31;
    int v = 0;
    M_NCCL(g_rccl.GetVersion(&v));
    return v;
}

int beom_multi_create_ex(const beom_params *prm, int ndev, const int *devices, int transport_and_flags,
                         const int32_t *neig, const int32_t *subc,
                         const double *mk_u, const double *mk_v, const double *mk_n,
                         const double *mkpe, const double *mkpi,
                         const double *fcor, const double *h_th, const double *h_to,
                         const double *nudg, const double *fnud, const double *hdot,
                         const double *tide, const double *bodf, const double *taus,
                         beom_multi_handle *out, char *errm, int errm_len) {
    if (!prm || !out || !devices || !neig || !subc || !mk_u || !mk_v || !mk_n || !mkpe || !mkpi || !fcor || !h_th || !nudg || !fnud) {
        m_err(errm, errm_len, "beom_multi_create: null argument"); return -1;
    }
    *out = nullptr;
    const int L = prm->lm + 1, Mg = prm->mm + 1, nl = prm->nlay;
    // periodicity is encoded only in neig (private_mod.f95:614-685): W of cell (1,1) / S of cell (1,1)
    const int xper = neig[8 * 1 + 4] != 0, yper = neig[8 * 1 + 6] != 0;
    const int transport = transport_and_flags & 0xff;
    const bool whole = ndev == 1 && !(yper && (transport_and_flags & BEOM_XCHG_RING1));      // one band = the frame itself
    bool land = false;
    if (!whole) {
        land = (long long)prm->ndeg != (long long)L * Mg;
        M_RC(check_frame(prm, ndev, yper, true, land, errm, errm_len));
        if (!land && !beom_dense::verify(L, Mg, 0, Mg, 0, xper, yper, prm->ndeg, neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi)) {
            m_err(errm, errm_len, "beom_multi_create: the row decomposition needs a dense frame (interior entirely wet)");
            return -4;
        }
    } else if (prm->abi_version != BEOM_ABI_VERSION) { m_err(errm, errm_len, "beom_multi_create: ABI version mismatch"); return -2; }
    beom_multi *M = new beom_multi();
    M->P = *prm; M->nb = ndev; M->n = ndev; M->n1g = (size_t)prm->ndeg + 1;
    M->ring = !whole && yper; M->xper = xper; M->transport = transport;
    M->dev.assign(devices, devices + ndev);
    if (getenv("BEOM_MULTI_WRAP_DEVICES")) {     // rehearsals: more bands than GPUs, ids taken modulo the visible count
        int nvis = 0;
        if (hipGetDeviceCount(&nvis) == hipSuccess && nvis > 0)
            for (int &dv : M->dev) dv %= nvis;
    }
    size_vectors(M);
    const size_t n1g = M->n1g;
    int rc = 0;
    if (whole) {                                  // the caller's own tables, any coastline
        Band s; s.index = 0; s.own0 = 1; s.own1 = Mg; s.L = L;
        M->band.push_back(s);
        rc = beom_create(prm, M->dev[0], neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor, h_th, h_to, nudg, fnud, hdot, tide, bodf, taus,
                         &M->eng[0], errm, errm_len);
        if (!rc && hipStreamCreateWithFlags(&M->main_s[0], hipStreamNonBlocking) != hipSuccess) { m_err(errm, errm_len, "hipStreamCreate failed"); rc = -100; }
        if (!rc) (void)beom_set_stream(M->eng[0], (void *)M->main_s[0], 0);
        if (rc) { destroy_all(M); return rc; }
        *out = M;
        return 0;
    }
    const double *src[8] = {fcor, h_th, h_to, nudg, fnud, hdot, tide, taus};
    if (land) {
        // rows as packed ranges (packing is j-major: subc's row number never decreases along the packed index)
        M->land = true;
        M->gst.assign((size_t)Mg + 3, 0);
        int jprev = 1;
        M->gst[1] = 1;
        for (size_t p = 1; p < n1g && !rc; ++p) {
            const int j = subc[p + n1g];
            if (j < jprev || j > Mg) { m_err(errm, errm_len, "beom_multi_create: the packed cells are not ordered by row (cell %d)", (int)p); rc = -4; break; }
            for (; jprev < j; ++jprev) M->gst[(size_t)jprev + 1] = (long long)p;
        }
        for (; jprev <= Mg + 1; ++jprev) M->gst[(size_t)jprev + 1] = (long long)n1g;
        // bands of equal packed-cell count (balance on the cells of a row range), at least kGhost + 1 rows each
        const int minrows = kGhost + 1;
        int j = 1;
        for (int k = 0; k < ndev && !rc; ++k) {
            Band b; b.index = k; b.L = L;
            b.own0 = j;
            const long long target = (long long)prm->ndeg * (k + 1) / ndev;
            const int last_allowed = Mg - (ndev - 1 - k) * minrows;
            int e = j + minrows - 1;
            while (e < last_allowed && M->gst[(size_t)e + 1] - 1 < target) ++e;
            if (k == ndev - 1) e = Mg;
            if (e > last_allowed || e < j) { m_err(errm, errm_len, "beom_multi_create: %d rows are too few for %d bands", Mg, ndev); rc = -3; break; }
            b.own1 = e; j = e + 1;
            b.gs = k > 0 ? kGhost : 0; b.gn = k < ndev - 1 ? kGhost : 0;
            b.lst.assign((size_t)b.rows() + 2, 0);
            b.lst[1] = 1;
            for (int jl = 1; jl <= b.rows(); ++jl) {
                const size_t g = (size_t)(b.own0 - b.gs + jl - 1);
                b.lst[(size_t)jl + 1] = b.lst[(size_t)jl] + (M->gst[g + 1] - M->gst[g]);
            }
            M->band.push_back(b);
        }
        const double *masks[5] = {mk_u, mk_v, mk_n, mkpe, mkpi};
        for (int k = 0; k < ndev && !rc; ++k) {
            const std::vector<int> rows = M->band[k].row_list();
            StaticsV st;
            for (int f = 0; f < 8; ++f) st.a[f] = cut_v(src[f], kStatic[f].outer(nl), kStatic[f].inner, n1g, rows, M->gst);
            st.bodf = bodf;
            rc = create_band_land(M, k, neig, subc, masks, st, errm, errm_len);
        }
        if (!rc) rc = init_transport(M, nullptr, errm, errm_len);
        if (rc) { destroy_all(M); return rc; }
        *out = M;
        return 0;
    }
    for (int k = 0; k < ndev && !rc; ++k) {
        M->band.push_back(make_band(prm, ndev, k, M->ring));
        const std::vector<int> rows = M->band[k].row_list();
        StaticsV st;
        for (int f = 0; f < 8; ++f) st.a[f] = cut(src[f], kStatic[f].outer(nl), kStatic[f].inner, n1g, rows, L);
        st.bodf = bodf;
        rc = create_band(M, k, st, errm, errm_len);
    }
    if (!rc && M->ring) {
        M->mini_k = 0;
        M->mini_rows = mini_row_list(prm->mm);
        StaticsV st;
        for (int f = 0; f < 8; ++f) st.a[f] = cut(src[f], kStatic[f].outer(nl), kStatic[f].inner, n1g, M->mini_rows, L);
        st.bodf = bodf;
        rc = create_mini(M, st, errm, errm_len);
    }
    if (!rc) rc = init_transport(M, nullptr, errm, errm_len);
    if (rc) { destroy_all(M); return rc; }
    *out = M;
    return 0;
}

int beom_multi_create(const beom_params *prm, int ndev, const int *devices,
                      const int32_t *neig, const int32_t *subc,
                      const double *mk_u, const double *mk_v, const double *mk_n,
                      const double *mkpe, const double *mkpi,
                      const double *fcor, const double *h_th, const double *h_to,
                      const double *nudg, const double *fnud, const double *hdot,
                      const double *tide, const double *bodf, const double *taus,
                      beom_multi_handle *out, char *errm, int errm_len) {
    const char *t = getenv("BEOM_XCHG");          // "rccl": the Fortran host picks the transport by environment
    const int transport = (t && (!strcmp(t, "rccl") || !strcmp(t, "RCCL"))) ? BEOM_XCHG_RCCL : BEOM_XCHG_PEER;
    return beom_multi_create_ex(prm, ndev, devices, transport, neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor, h_th, h_to,
                                nudg, fnud, hdot, tide, bodf, taus, out, errm, errm_len);
}

int beom_multi_window(const beom_params *prm, int nb, int band, int yper, int *own0, int *own1, int *ghost_s, int *ghost_n) {
    if (!prm || nb < 1 || band < 0 || band >= nb) return -3;
    const Band s = make_band(prm, nb, band, yper != 0);
    if (own0) *own0 = s.own0;
    if (own1) *own1 = s.own1;
    if (ghost_s) *ghost_s = s.gs;
    if (ghost_n) *ghost_n = s.gn;
    return 0;
}

int beom_multi_create_local(const beom_params *prm, int nb, int band, int device, int xper, int yper,
                            const void *rccl_id, const beom_statics *win, const beom_statics *orphan,
                            beom_multi_handle *out, char *errm, int errm_len) {
    return beom_multi_create_local_ex(prm, nb, band, device, xper, yper, BEOM_XCHG_RCCL, rccl_id, win, orphan, out, errm, errm_len);
}

int beom_multi_create_local_ex(const beom_params *prm, int nb, int band, int device, int xper, int yper,
                               int transport_and_flags, const void *xchg_id, const beom_statics *win, const beom_statics *orphan,
                               beom_multi_handle *out, char *errm, int errm_len) {
    if (!prm || !out || !win || !win->fcor || !win->h_th || !win->nudg || !win->fnud) { m_err(errm, errm_len, "beom_multi_create_local: null argument"); return -1; }
    *out = nullptr;
    const int transport = transport_and_flags & 0xff;
    const void *rccl_id = xchg_id;
    if (transport != BEOM_XCHG_RCCL && transport != BEOM_XCHG_SHM) { m_err(errm, errm_len, "beom_multi_create_local: one band per process exchanges over RCCL or shared memory"); return -3; }
    if (band < 0 || band >= nb) { m_err(errm, errm_len, "beom_multi_create_local: band %d of %d", band, nb); return -3; }
    M_RC(check_frame(prm, nb, yper, false, false, errm, errm_len));
    const bool ring = yper != 0;
    if (ring && band == 0 && (!orphan || !orphan->fcor || !orphan->h_th || !orphan->nudg || !orphan->fnud)) {
        m_err(errm, errm_len, "beom_multi_create_local: band 0 of a frame periodic in y also carries row mm+1 (orphan statics needed)");
        return -1;
    }
    beom_multi *M = new beom_multi();
    M->P = *prm; M->nb = nb; M->n = 1; M->n1g = (size_t)prm->ndeg + 1;
    M->ring = ring; M->xper = xper != 0; M->transport = transport; M->local_mode = true;
    M->loopback = (transport_and_flags & BEOM_XCHG_LOOPBACK) != 0;
    M->dev.assign(1, device);
    size_vectors(M);
    M->band.push_back(make_band(prm, nb, band, ring));
    const Band &s = M->band[0];
    const int nl = prm->nlay;
    const size_t n1w = (size_t)s.n_loc() + 1;
    const double *wp[8], *op[8];
    statics_ptrs(win, wp);
    StaticsV st;
    for (int f = 0; f < 8; ++f)
        if (wp[f]) st.a[f].assign(wp[f], wp[f] + kStatic[f].outer(nl) * n1w * kStatic[f].inner);
    st.bodf = win->bodf;
    int rc = create_band(M, 0, st, errm, errm_len);
    if (!rc && ring && band == 0) {
        M->mini_k = 0;
        M->mini_rows = mini_row_list(prm->mm);
        statics_ptrs(orphan, op);
        StaticsV sm;
        for (int f = 0; f < 8; ++f) sm.a[f] = mini_from_window(wp[f], op[f], kStatic[f].outer(nl), kStatic[f].inner, s);
        sm.bodf = win->bodf;
        rc = create_mini(M, sm, errm, errm_len);
    }
    if (!rc && (nb > 1 || ring)) rc = init_transport(M, rccl_id, errm, errm_len);
    if (rc) { destroy_all(M); return rc; }
    *out = M;
    return 0;
}

int beom_multi_destroy(beom_multi_handle M) { destroy_all(M); return 0; }

int beom_multi_count(beom_multi_handle M) { return M ? M->n : -1; }

int beom_multi_band(beom_multi_handle M, int k, int *own0, int *own1, int *win0, int *win1, int *device) {
    if (!M || k < 0 || k >= M->n) return -3;
    if (own0) *own0 = M->band[k].own0;
    if (own1) *own1 = M->band[k].own1;
    if (win0) *win0 = M->band[k].own0 - M->band[k].gs;        // <= 0 / > mm: ghost rows of a ring wrap
    if (win1) *win1 = M->band[k].own1 + M->band[k].gn;
    if (device) *device = M->dev[k];
    return 0;
}

int beom_multi_engine(beom_multi_handle M, int k, beom_handle *out) {
    if (!M || !out || k < -1 || k >= M->n) return -3;
    *out = k < 0 ? M->mini : M->eng[k];
    return 0;
}

int beom_multi_stats(beom_multi_handle M, long long *split_band_steps, long long *plain_band_steps) {
    if (!M) return -1;
    if (split_band_steps) *split_band_steps = M->n_split;
    if (plain_band_steps) *plain_band_steps = M->n_plain;
    return 0;
}

// "overlap" (default 1): steps run in two phases around the ghost exchange still in flight; 0 = every step waits
// for its ghosts first (same results; bench.py checks one form against the other on the machine at hand).
// Any other name is forwarded to every local band (beom_set_option).
int beom_multi_set_option(beom_multi_handle M, const char *name, int value) {
    if (!M || !name) return -1;
    if (!strcmp(name, "overlap")) { M->overlap = value != 0; return 0; }
    if (!strcmp(name, "edge_stream")) return 0;       // (an option of the earlier split form; accepted, without effect)
    int rc = 0;
    for (int k = 0; k < M->n && !rc; ++k) rc = beom_set_option(M->eng[k], name, value);
    if (!rc && M->mini) rc = beom_set_option(M->mini, name, value);
    return rc;
}

int beom_multi_describe(beom_multi_handle M, int *bands_total, int *bands_local, int *transport, int *ring, int *rccl_version) {
    if (!M) return -1;
    if (bands_total) *bands_total = M->nb;
    if (bands_local) *bands_local = M->n;
    if (transport) *transport = M->transport;
    if (ring) *ring = M->ring ? 1 : 0;
    if (rccl_version) { *rccl_version = 0; if (g_rccl.lib) (void)g_rccl.GetVersion(rccl_version); }
    return 0;
}

int beom_multi_sync(beom_multi_handle M, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    for (int k = 0; k < M->n; ++k) {
        M_HIP(hipSetDevice(M->dev[k]));
        M_HIP(hipStreamSynchronize(M->main_s[k]));
        if (M->comm_s[k]) M_HIP(hipStreamSynchronize(M->comm_s[k]));
        M->pending[k] = 0;          // whatever was in flight has landed
    }
    if (M->mini) { M_HIP(hipSetDevice(M->dev[M->mini_k])); M_HIP(hipStreamSynchronize(M->mini_s)); }
    M_HIP(hipGetLastError());
    if (M->shm && M->shm->failed) { M->failed = true; m_err(errm, errm_len, "beom_multi: a neighbour's ghost rows did not arrive within %.0f s (shared-memory transport)", M->shm->timeout_s); return -35; }
    return 0;
}

int beom_multi_upload_state(beom_multi_handle M, const double *hlay, const double *u, const double *v,
                            const double *h_u, const double *h_v, const double *rs_h, const double *dmdx,
                            const double *dmdy, const double *v_cc, const double *v_ll, const double *tt3d,
                            const double *tb3d, const double *tu3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (M->local_mode) { m_err(errm, errm_len, "beom_multi_upload_state: this handle holds a window (use beom_multi_upload_local)"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    const int nl = M->P.nlay, L = M->P.lm + 1;
    const size_t n1g = M->n1g;
    const double *src[13] = {hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc, v_ll, tt3d, tb3d, tu3d};
    if (M->nb == 1 && !M->ring)
        return beom_upload_state(M->eng[0], hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc, v_ll, tt3d, tb3d, tu3d, errm, errm_len);
    for (int k = -1; k < M->n; ++k) {
        if (k < 0 && !M->mini) continue;
        const std::vector<int> rows = k < 0 ? M->mini_rows : M->band[k].row_list();
        StateV a;
        for (int f = 0; f < 13; ++f)
            a.a[f] = M->land ? cut_v(src[f], kState[f].outer(nl), kState[f].inner, n1g, rows, M->gst)
                             : cut(src[f], kState[f].outer(nl), kState[f].inner, n1g, rows, L);
        M_RC(beom_upload_state(k < 0 ? M->mini : M->eng[k], ptr(a.a[0]), ptr(a.a[1]), ptr(a.a[2]), ptr(a.a[3]), ptr(a.a[4]), ptr(a.a[5]),
                               ptr(a.a[6]), ptr(a.a[7]), ptr(a.a[8]), ptr(a.a[9]), ptr(a.a[10]), ptr(a.a[11]), ptr(a.a[12]), errm, errm_len));
    }
    return 0;
}

int beom_multi_download_state(beom_multi_handle M, double *hlay, double *u, double *v, double *h_u, double *h_v,
                              double *rs_h, double *dmdx, double *dmdy, double *v_cc, double *v_ll,
                              double *tt3d, double *tb3d, double *tu3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (M->local_mode) { m_err(errm, errm_len, "beom_multi_download_state: this handle holds a window (use beom_multi_download_local)"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    const int nl = M->P.nlay, L = M->P.lm + 1;
    const size_t n1g = M->n1g;
    double *dst[13] = {hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc, v_ll, tt3d, tb3d, tu3d};
    if (M->nb == 1 && !M->ring)
        return beom_download_state(M->eng[0], hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc, v_ll, tt3d, tb3d, tu3d, errm, errm_len);
    for (int k = -1; k < M->n; ++k) {
        if (k < 0 && !M->mini) continue;
        const size_t n1l = k < 0 ? (size_t)M->mini_rows.size() * L + 1 : (size_t)M->band[k].n_loc() + 1;
        StateV a;
        for (int f = 0; f < 13; ++f) if (dst[f]) a.a[f].assign(kState[f].outer(nl) * n1l * kState[f].inner, 0.0);
        M_RC(beom_download_state(k < 0 ? M->mini : M->eng[k], ptr(a.a[0]), ptr(a.a[1]), ptr(a.a[2]), ptr(a.a[3]), ptr(a.a[4]), ptr(a.a[5]),
                                 ptr(a.a[6]), ptr(a.a[7]), ptr(a.a[8]), ptr(a.a[9]), ptr(a.a[10]), ptr(a.a[11]), ptr(a.a[12]), errm, errm_len));
        int j0, n;
        std::vector<int> rows;
        if (k < 0) { j0 = (int)M->mini_rows.size(); n = 1; rows.push_back(M->P.mm + 1); }      // the orphan row
        else { const Band &s = M->band[k]; j0 = s.gs + 1; n = s.nown(); for (int j = 0; j < n; ++j) rows.push_back(s.own0 + j); }
        for (int f = 0; f < 13; ++f) {
            if (M->land) paste_v(dst[f], a.a[f], kState[f].outer(nl), kState[f].inner, n1g, n1l, M->band[k].lst, M->gst, j0, n, rows, k == 0);
            else paste(dst[f], a.a[f], kState[f].outer(nl), kState[f].inner, n1g, n1l, L, j0, n, rows, k == 0);
        }
    }
    return 0;
}

// ---- output records of all bands (SURVEY §8f N2 for the multi-device handle): as beom_download_outputs /
//      beom_download_diag with GLOBAL (ndeg, nlay) real*4 records; every band forms its rows on its device.
namespace {
// (ndeg, nlay) records have no sentinel: cell (row r, column i) of layer k at (i-1) + (r-1)*L + ndeg*(k-1)
static std::vector<float> cut_rec(const float *x, int nl, size_t ndeg_g, const std::vector<int> &rows, int L) {
    std::vector<float> z;
    if (!x) return z;
    const size_t nloc = rows.size() * (size_t)L;
    z.resize(nloc * nl);
    for (int k = 0; k < nl; ++k)
        for (size_t r = 0; r < rows.size(); ++r)
            std::memcpy(&z[k * nloc + r * L], &x[k * ndeg_g + (size_t)(rows[r] - 1) * L], (size_t)L * sizeof(float));
    return z;
}
// (frames with land: rows as packed ranges; records have no sentinel, so cell p sits at p - 1)
static std::vector<float> cut_rec_v(const float *x, int nl, size_t ndeg_g, const std::vector<int> &rows, const std::vector<long long> &gst) {
    std::vector<float> z;
    if (!x) return z;
    size_t nloc = 0;
    for (int g : rows) nloc += (size_t)(gst[(size_t)g + 1] - gst[(size_t)g]);
    z.resize(nloc * nl);
    for (int k = 0; k < nl; ++k) {
        size_t at = 0;
        for (int g : rows) {
            const size_t len = (size_t)(gst[(size_t)g + 1] - gst[(size_t)g]);
            if (len) std::memcpy(&z[k * nloc + at], &x[k * ndeg_g + (size_t)gst[(size_t)g] - 1], len * sizeof(float));
            at += len;
        }
    }
    return z;
}
void paste_rec_v(float *dst, const std::vector<float> &loc, int nl, size_t ndeg_g, size_t nloc, const std::vector<long long> &lst,
                 const std::vector<long long> &gst, int j0, int n, const std::vector<int> &dst_rows) {
    if (!dst || loc.empty()) return;
    for (int k = 0; k < nl; ++k)
        for (int r = 0; r < n; ++r) {
            const size_t g = (size_t)dst_rows[(size_t)r], len = (size_t)(gst[g + 1] - gst[g]);
            if (len) std::memcpy(&dst[k * ndeg_g + (size_t)gst[g] - 1], &loc[k * nloc + (size_t)lst[(size_t)(j0 + r)] - 1], len * sizeof(float));
        }
}
void paste_rec(float *dst, const std::vector<float> &loc, int nl, size_t ndeg_g, size_t nloc, int L, int j0, int n,
               const std::vector<int> &dst_rows) {
    if (!dst || loc.empty()) return;
    for (int k = 0; k < nl; ++k)
        for (int r = 0; r < n; ++r)
            std::memcpy(&dst[k * ndeg_g + (size_t)(dst_rows[r] - 1) * L], &loc[k * nloc + (size_t)(j0 - 1 + r) * L], (size_t)L * sizeof(float));
}
}  // namespace

int beom_multi_download_outputs(beom_multi_handle M, const float *h0r4, float *eta, float *u4, float *v4,
                                double *minmax, int *thin_layer, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (M->local_mode) { m_err(errm, errm_len, "beom_multi_download_outputs: this handle holds a window"); return -3; }
    if (!h0r4) { m_err(errm, errm_len, "beom_multi_download_outputs: h_0 (real*4) is needed"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    if (M->nb == 1 && !M->ring) return beom_download_outputs(M->eng[0], h0r4, eta, u4, v4, minmax, thin_layer, errm, errm_len);
    const int nl = M->P.nlay, L = M->P.lm + 1;
    const size_t ndeg_g = (size_t)M->P.ndeg;
    if (thin_layer) *thin_layer = 0;
    for (int k = -1; k < M->n; ++k) {
        if (k < 0 && !M->mini) continue;
        const std::vector<int> rows = k < 0 ? M->mini_rows : M->band[k].row_list();
        const size_t nloc = M->land ? (size_t)M->band[k].n_loc() : rows.size() * (size_t)L;
        std::vector<float> h0 = M->land ? cut_rec_v(h0r4, nl, ndeg_g, rows, M->gst) : cut_rec(h0r4, nl, ndeg_g, rows, L);
        std::vector<float> e(eta ? nloc * nl : 0), a(u4 ? nloc * nl : 0), b(v4 ? nloc * nl : 0);
        std::vector<double> mm((size_t)nl * 6);
        int thin = 0;
        M_RC(beom_download_outputs(k < 0 ? M->mini : M->eng[k], h0.data(), ptr(e), ptr(a), ptr(b), mm.data(), &thin, errm, errm_len));
        int j0, n;
        std::vector<int> dst_rows;
        if (k < 0) { j0 = (int)rows.size(); n = 1; dst_rows.push_back(M->P.mm + 1); }       // the orphan row; its scans are not merged
        else { const Band &s = M->band[k]; j0 = s.gs + 1; n = s.nown(); for (int j = 0; j < n; ++j) dst_rows.push_back(s.own0 + j); }
        if (M->land) {
            paste_rec_v(eta, e, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
            paste_rec_v(u4, a, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
            paste_rec_v(v4, b, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
        } else {
            paste_rec(eta, e, nl, ndeg_g, nloc, L, j0, n, dst_rows);
            paste_rec(u4, a, nl, ndeg_g, nloc, L, j0, n, dst_rows);
            paste_rec(v4, b, nl, ndeg_g, nloc, L, j0, n, dst_rows);
        }
        if (k < 0) continue;
        // a band's scans cover its ghost rows too: they hold the neighbours' owned values (the exchange has landed)
        if (minmax)
            for (int q = 0; q < nl * 6; ++q)
                minmax[q] = (k == 0) ? mm[q] : ((q % 2 == 0) ? std::fmin(minmax[q], mm[q]) : std::fmax(minmax[q], mm[q]));
        if (thin_layer && thin > 0 && (*thin_layer == 0 || thin < *thin_layer)) *thin_layer = thin;
    }
    return 0;
}

// no_gradient_obc (private_mod.f95:2613-2679) on a frame cut into bands: the table segm(nseg, 18) of beom_set_open_boundaries
// with GLOBAL cell indices; every band gets the passes of the segments whose updated cell and source cell lie in its rows
// (owned or ghost), re-indexed to its window.  A band's step applies them after its momentum sweeps, as the single handle does
// (such steps are not split: the exchange follows the whole step).
int beom_multi_set_open_boundaries(beom_multi_handle M, int nseg, const int32_t *segm, char *errm, int errm_len) {
    if (!M || nseg < 1 || !segm) { m_err(errm, errm_len, "beom_multi_set_open_boundaries: bad arguments"); return -1; }
    if (M->local_mode) { m_err(errm, errm_len, "beom_multi_set_open_boundaries: needs a handle created from the global arrays"); return -3; }
    if (M->land) { m_err(errm, errm_len, "beom_multi_set_open_boundaries: not for bands of a frame with land"); return -3; }
    if (M->nb == 1 && !M->ring) return beom_set_open_boundaries(M->eng[0], nseg, segm, errm, errm_len);
    const int L = M->P.lm + 1;
    auto S = [&](int is, int col) { return segm[(size_t)is + (size_t)nseg * (col - 1)]; };
    // k = -1: the companion frame of a ring (rows 1..6, mm-3..mm, mm+1 — it holds the orphan row's end of every segment)
    for (int k = M->mini ? -1 : 0; k < M->n; ++k) {
        const std::vector<int> wrows = k < 0 ? M->mini_rows : M->band[k].row_list();   // global row of every local row (a ring's ghosts wrap)
        // A pass of a segment (its updated cell: column 10 for the first pass, 1 for the second; its source cell: 16 / 13) goes
        // to EVERY local row that holds the updated cell's global row — in a ring a row can be there twice, owned and as a
        // wrapped ghost — with the source cell taken from the local row next to it (or itself) that holds the source's row.
        std::vector<std::array<int32_t, 18>> rowsv;
        auto row_of = [&](int32_t q) { return (q - 1) / L + 1; };
        auto col_of = [&](int32_t q) { return (q - 1) % L + 1; };
        for (int is = 0; is < nseg; ++is)
            for (int pass = 0; pass < 2; ++pass) {
                const int cu = pass == 0 ? 10 : 1, cs = pass == 0 ? 16 : 13;
                const int32_t qu = S(is, cu), qs = S(is, cs);
                if (qu < 1) continue;
                for (size_t jl = 0; jl < wrows.size(); ++jl) {
                    if (wrows[jl] != row_of(qu)) continue;
                    int32_t src = qs == 0 ? 0 : -1;
                    if (qs > 0)
                        for (long long js = (long long)jl - 1; js <= (long long)jl + 1; ++js)
                            if (js >= 0 && js < (long long)wrows.size() && wrows[(size_t)js] == row_of(qs)) { src = (int32_t)(col_of(qs) + js * L); break; }
                    if (src < 0) continue;                                   // the source's row is not next to this copy of the row: another band's
                    std::array<int32_t, 18> r;
                    for (int c = 1; c <= 18; ++c) r[(size_t)c - 1] = S(is, c);
                    r[0] = r[9] = -1; r[12] = r[15] = r[6] = 0;              // both passes off, then this one on
                    r[(size_t)cu - 1] = (int32_t)(col_of(qu) + (long long)jl * L);
                    r[(size_t)cs - 1] = src;
                    rowsv.push_back(r);
                }
            }
        const int nloc = (int)rowsv.size();
        std::vector<int32_t> tab((size_t)nloc * 18);
        for (int is = 0; is < nloc; ++is)
            for (int c = 0; c < 18; ++c) tab[(size_t)is + (size_t)nloc * c] = rowsv[(size_t)is][(size_t)c];
        M_RC(beom_set_open_boundaries(k < 0 ? M->mini : M->eng[k], nloc, nloc ? tab.data() : nullptr, errm, errm_len));
    }
    return 0;
}

// The same for a handle that holds ONE band's window (beom_multi_create_local): the caller found the segments of its own
// rows — the finder (index_boundary_points, private_mod.f95:1060-1240) looks at a cell and its four neighbours only, so a
// window's rows (ghost rows included, the row below and above them known) give exactly the global table's entries of those
// rows — with cell indices of the window.  Band 0 of a ring also passes the segments of the orphan row mm+1 (indices of a
// one-row frame); the companion frame's table is put together here from both.
int beom_multi_set_open_boundaries_local(beom_multi_handle M, int nseg, const int32_t *segm, int nseg_orphan, const int32_t *segm_orphan,
                                         char *errm, int errm_len) {
    if (!M || nseg < 0 || (nseg > 0 && !segm) || nseg_orphan < 0 || (nseg_orphan > 0 && !segm_orphan)) { m_err(errm, errm_len, "beom_multi_set_open_boundaries_local: bad arguments"); return -1; }
    if (!M->local_mode) { m_err(errm, errm_len, "beom_multi_set_open_boundaries_local: this handle was created from global arrays"); return -3; }
    M_RC(beom_set_open_boundaries(M->eng[0], nseg, nseg ? segm : nullptr, errm, errm_len));
    if (!M->mini) return 0;
    const int L = M->P.lm + 1, gs = M->band[0].gs;
    std::vector<std::array<int32_t, 18>> rowsv;
    // window row jl (1-based) -> row of the companion frame, 0 = not there
    auto from_window = [&](int jl) { return (jl >= 1 && jl <= gs) ? kMiniLo + jl : (jl > gs && jl <= gs + kMiniLo) ? jl - gs : 0; };
    auto from_orphan = [&](int jl) { return jl == 1 ? kMiniLo + kGhost + 1 : 0; };
    for (int src = 0; src < 2; ++src) {
        const int n = src ? nseg_orphan : nseg;
        const int32_t *tab = src ? segm_orphan : segm;
        for (int is = 0; is < n; ++is)
            for (int pass = 0; pass < 2; ++pass) {
                const int cu = pass == 0 ? 10 : 1, cs = pass == 0 ? 16 : 13;
                const int32_t qu = tab[(size_t)is + (size_t)n * (cu - 1)], qs = tab[(size_t)is + (size_t)n * (cs - 1)];
                if (qu < 1) continue;
                const int mu = src ? from_orphan((qu - 1) / L + 1) : from_window((qu - 1) / L + 1);
                const int ms = qs > 0 ? (src ? from_orphan((qs - 1) / L + 1) : from_window((qs - 1) / L + 1)) : 0;
                if (!mu || (qs > 0 && !ms)) continue;
                std::array<int32_t, 18> r;
                for (int c = 1; c <= 18; ++c) r[(size_t)c - 1] = tab[(size_t)is + (size_t)n * (c - 1)];
                r[0] = r[9] = -1; r[12] = r[15] = r[6] = 0;
                r[(size_t)cu - 1] = (int32_t)((qu - 1) % L + 1 + (long long)(mu - 1) * L);
                r[(size_t)cs - 1] = qs > 0 ? (int32_t)((qs - 1) % L + 1 + (long long)(ms - 1) * L) : 0;
                rowsv.push_back(r);
            }
    }
    const int nloc = (int)rowsv.size();
    std::vector<int32_t> t2((size_t)nloc * 18);
    for (int is = 0; is < nloc; ++is)
        for (int c = 0; c < 18; ++c) t2[(size_t)is + (size_t)nloc * c] = rowsv[(size_t)is][(size_t)c];
    return beom_set_open_boundaries(M->mini, nloc, nloc ? t2.data() : nullptr, errm, errm_len);
}

int beom_multi_download_diag(beom_multi_handle M, float *pvor4, float *mont4, float *vcc4, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (M->local_mode) { m_err(errm, errm_len, "beom_multi_download_diag: this handle holds a window"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    if (M->nb == 1 && !M->ring) return beom_download_diag(M->eng[0], pvor4, mont4, vcc4, errm, errm_len);
    const int nl = M->P.nlay, L = M->P.lm + 1;
    const size_t ndeg_g = (size_t)M->P.ndeg;
    for (int k = -1; k < M->n; ++k) {
        if (k < 0 && !M->mini) continue;
        const std::vector<int> rows = k < 0 ? M->mini_rows : M->band[k].row_list();
        const size_t nloc = M->land ? (size_t)M->band[k].n_loc() : rows.size() * (size_t)L;
        std::vector<float> a(pvor4 ? nloc * nl : 0), b(mont4 ? nloc * nl : 0), c(vcc4 ? nloc * nl : 0);
        M_RC(beom_download_diag(k < 0 ? M->mini : M->eng[k], ptr(a), ptr(b), ptr(c), errm, errm_len));
        int j0, n;
        std::vector<int> dst_rows;
        if (k < 0) { j0 = (int)rows.size(); n = 1; dst_rows.push_back(M->P.mm + 1); }
        else { const Band &s = M->band[k]; j0 = s.gs + 1; n = s.nown(); for (int j = 0; j < n; ++j) dst_rows.push_back(s.own0 + j); }
        if (M->land) {
            paste_rec_v(pvor4, a, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
            paste_rec_v(mont4, b, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
            paste_rec_v(vcc4, c, nl, ndeg_g, nloc, M->band[k].lst, M->gst, j0, n, dst_rows);
        } else {
            paste_rec(pvor4, a, nl, ndeg_g, nloc, L, j0, n, dst_rows);
            paste_rec(mont4, b, nl, ndeg_g, nloc, L, j0, n, dst_rows);
            paste_rec(vcc4, c, nl, ndeg_g, nloc, L, j0, n, dst_rows);
        }
    }
    return 0;
}

int beom_multi_upload_local(beom_multi_handle M, const beom_state *win, const beom_state *orphan, char *errm, int errm_len) {
    if (!M || !win) { m_err(errm, errm_len, "null argument"); return -1; }
    if (!M->local_mode) { m_err(errm, errm_len, "beom_multi_upload_local: this handle was created from global arrays"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    M_RC(beom_upload_state(M->eng[0], win->hlay, win->u, win->v, win->h_u, win->h_v, win->rs_h, win->dmdx, win->dmdy,
                           win->v_cc, win->v_ll, win->tt3d, win->tb3d, win->tu3d, errm, errm_len));
    if (M->mini) {
        if (!orphan) { m_err(errm, errm_len, "beom_multi_upload_local: band 0 of a ring needs the orphan row's state"); return -1; }
        double *wp[13], *op[13];
        state_ptrs(win, wp); state_ptrs(orphan, op);
        StateV a;
        for (int f = 0; f < 13; ++f) a.a[f] = mini_from_window(wp[f], op[f], kState[f].outer(M->P.nlay), kState[f].inner, M->band[0]);
        M_RC(beom_upload_state(M->mini, ptr(a.a[0]), ptr(a.a[1]), ptr(a.a[2]), ptr(a.a[3]), ptr(a.a[4]), ptr(a.a[5]), ptr(a.a[6]),
                               ptr(a.a[7]), ptr(a.a[8]), ptr(a.a[9]), ptr(a.a[10]), ptr(a.a[11]), ptr(a.a[12]), errm, errm_len));
    }
    return 0;
}

int beom_multi_download_local(beom_multi_handle M, beom_state *win, beom_state *orphan, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null argument"); return -1; }
    if (!M->local_mode) { m_err(errm, errm_len, "beom_multi_download_local: this handle was created from global arrays"); return -3; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    if (win)
        M_RC(beom_download_state(M->eng[0], win->hlay, win->u, win->v, win->h_u, win->h_v, win->rs_h, win->dmdx, win->dmdy,
                                 win->v_cc, win->v_ll, win->tt3d, win->tb3d, win->tu3d, errm, errm_len));
    if (orphan && M->mini) {
        const int nl = M->P.nlay, L = M->P.lm + 1;
        const size_t n1m = (size_t)M->mini_rows.size() * L + 1, n1o = (size_t)L + 1;
        double *op[13];
        state_ptrs(orphan, op);
        StateV a;
        for (int f = 0; f < 13; ++f) if (op[f]) a.a[f].assign(kState[f].outer(nl) * n1m * kState[f].inner, 0.0);
        M_RC(beom_download_state(M->mini, ptr(a.a[0]), ptr(a.a[1]), ptr(a.a[2]), ptr(a.a[3]), ptr(a.a[4]), ptr(a.a[5]), ptr(a.a[6]),
                                 ptr(a.a[7]), ptr(a.a[8]), ptr(a.a[9]), ptr(a.a[10]), ptr(a.a[11]), ptr(a.a[12]), errm, errm_len));
        const std::vector<int> one(1, 1);
        for (int f = 0; f < 13; ++f)
            paste(op[f], a.a[f], kState[f].outer(nl), kState[f].inner, n1o, n1m, L, (int)M->mini_rows.size(), 1, one, true);
    }
    return 0;
}

int beom_multi_profile_start(beom_multi_handle M) {
    if (!M) return -1;
    for (int k = 0; k < M->n; ++k) (void)beom_profile_start(M->eng[k]);
    return 0;
}
// per sweep class: the slowest local band (ms summed over its launches) and that band's launch count
int beom_multi_profile_stop(beom_multi_handle M, double *ms, int *launches, char *errm, int errm_len) {
    if (!M || !ms || !launches) { m_err(errm, errm_len, "null argument"); return -1; }
    M_RC(beom_multi_sync(M, errm, errm_len));
    for (int c = 0; c < 8; ++c) { ms[c] = 0.0; launches[c] = 0; }
    for (int k = 0; k < M->n; ++k) {
        double m[8] = {0}; int l[8] = {0};
        M_RC(beom_profile_stop(M->eng[k], m, l, errm, errm_len));
        for (int c = 0; c < 8; ++c) if (m[c] > ms[c]) { ms[c] = m[c]; launches[c] = l[c]; }
    }
    return 0;
}

}  // extern "C"

// ---- one time step of all local bands -------------------------------------------------------------
// Boundary first (beom_step_phase): a band's main stream runs the step up to the momentum sweeps on all rows, then the
// momentum sweep on the rows in between; its second stream, behind the first part, runs the momentum sweep on the strips
// next to the ghost zones, packs the outermost owned rows, moves them, unpacks what the neighbours sent.  The main stream
// looks at the second one exactly once per step — "have my ghost rows landed?" before the next step starts — and by then
// the exchange has had the whole interior sweep to finish.  Steps that cannot be cut that way (open-boundary passes, the
// separate u and v sweeps, option "overlap" = 0) run whole, the exchange after them.
static int multi_one_step(beom_multi *M, int t, double tres, double dtd8, double dt_r, double rsta, int n_3d,
                          char *errm, int errm_len) {
    const int n = M->n;
    const int mk = M->mini_k;
    if (M->shm && M->shm->failed) { m_err(errm, errm_len, "beom_multi: a neighbour's ghost rows did not arrive within %.0f s (shared-memory transport)", M->shm->timeout_s); return -35; }
    // the ghost rows of the previous step have landed: before this step reads them
    for (int k = 0; k < n; ++k) {
        if (!M->pending[k]) continue;
        M_HIP(hipSetDevice(M->dev[k]));
        M_HIP(hipStreamWaitEvent(M->main_s[k], M->landed[k], 0));
    }
    // companion frame of a ring: rows 1..6 of band 0 and its south ghosts (= rows mm-3..mm) as they stand before this
    // step, then the companion's own step on its own stream
    if (M->mini) {
        M_HIP(hipSetDevice(M->dev[mk]));
        if (M->free_recorded) M_HIP(hipStreamWaitEvent(M->main_s[mk], M->ev_free, 0));      // (the companion has taken the copies of the step before)
        if (beom_pack_rows(M->eng[mk], M->band[mk].gs + 1, kMiniLo, M->mini_lo) || beom_pack_rows(M->eng[mk], 1, kGhost, M->mini_hi)) {
            m_err(errm, errm_len, "beom_pack_rows failed"); return -3;
        }
        M_HIP(hipEventRecord(M->ev_hi, M->main_s[mk]));
        M_HIP(hipStreamWaitEvent(M->mini_s, M->ev_hi, 0));
        if (beom_unpack_rows(M->mini, 1, kMiniLo, M->mini_lo) || beom_unpack_rows(M->mini, kMiniLo + 1, kGhost, M->mini_hi)) {
            m_err(errm, errm_len, "beom_unpack_rows failed"); return -3;
        }
        M_HIP(hipEventRecord(M->ev_free, M->mini_s));
        M->free_recorded = true;
        M_RC(beom_step(M->mini, t, 1, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len));
    }
    // the step; a band's packing stream X is its second stream when the step is cut, else its main stream
    std::vector<hipStream_t> X(n, nullptr);
    for (int k = 0; k < n; ++k) {
        const Band &s = M->band[k];
        M_HIP(hipSetDevice(M->dev[k]));
        struct Back { beom_multi *M; int k; ~Back() { (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0); } } back{M, k};
        int rc = M->overlap ? beom_step_phase(M->eng[k], t, tres, dtd8, dt_r, rsta, n_3d, 1, errm, errm_len) : -20;
        if (rc == 0) {
            M_HIP(hipEventRecord(M->p1done[k], M->main_s[k]));
            M_HIP(hipStreamWaitEvent(M->comm_s[k], M->p1done[k], 0));
            (void)beom_set_stream(M->eng[k], (void *)M->comm_s[k], 0);
            M_RC(beom_step_phase(M->eng[k], t, tres, dtd8, dt_r, rsta, n_3d, 2, errm, errm_len));
            (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0);
            M_RC(beom_step_phase(M->eng[k], t, tres, dtd8, dt_r, rsta, n_3d, 3, errm, errm_len));
            X[k] = M->comm_s[k];
            ++M->n_split;
        } else if (rc == -20) {
            M_RC(beom_step(M->eng[k], t, 1, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len));
            // peer copies read a neighbour's send buffer from ITS stream: keep the exchange of a whole step on the second
            // stream there; RCCL and the shared-memory transport order everything themselves: no stream hop at all
            X[k] = M->transport == BEOM_XCHG_PEER ? M->comm_s[k] : M->main_s[k];
            if (X[k] != M->main_s[k]) {
                M_HIP(hipEventRecord(M->p1done[k], M->main_s[k]));
                M_HIP(hipStreamWaitEvent(X[k], M->p1done[k], 0));
            }
            ++M->n_plain;
        } else return rc;
        // what the neighbours need = my outermost OWNED rows.  With peer copies my neighbours still read the send buffers
        // of the step before until THEIR ghosts have landed.
        if (M->transport == BEOM_XCHG_PEER)
            for (int q : {M->south_of(k), M->north_of(k)}) {
                const int ql = M->local_of(q);
                if (ql >= 0 && ql != k && M->pending[ql]) M_HIP(hipStreamWaitEvent(X[k], M->landed[ql], 0));
            }
        (void)beom_set_stream(M->eng[k], (void *)X[k], 0);
        if (M->has_s(k) && M->has_n(k)) {
            if (beom_pack_rows2(M->eng[k], kGhost, s.gs + 1, M->send_s[k], s.gs + s.nown() - kGhost + 1, M->send_n[k])) { m_err(errm, errm_len, "beom_pack_rows2 failed"); return -3; }
        } else {
            if (M->has_s(k) && beom_pack_rows(M->eng[k], s.gs + 1, kGhost, M->send_s[k])) { m_err(errm, errm_len, "beom_pack_rows failed"); return -3; }
            if (M->has_n(k) && beom_pack_rows(M->eng[k], s.gs + s.nown() - kGhost + 1, kGhost, M->send_n[k])) { m_err(errm, errm_len, "beom_pack_rows failed"); return -3; }
        }
        M_HIP(hipEventRecord(M->packed[k], X[k]));
    }
    for (int k = 0; k < n; ++k) M->pending[k] = 0;
    // the exchange, in the packing stream's order
    if (M->transport == BEOM_XCHG_RCCL) {
        // sends in the order (south, north), receives in the order (north, south): with two bands in a ring, or
        // one, both neighbours are the same peer and RCCL matches a pair's messages in issue order
        M_NCCL(g_rccl.GroupStart());
        int e = 0;                                  // (a failed call must not leave the process-wide group open)
        for (int k = 0; k < n && !e; ++k) {
            if (hipSetDevice(M->dev[k]) != hipSuccess) { e = 1; break; }
            const int S = M->rank_of(M->south_of(k)), N = M->rank_of(M->north_of(k));
            if (!e && M->has_s(k)) e = g_rccl.Send(M->send_s[k], M->xbytes, kNcclChar, S, M->comm[k], X[k]);
            if (!e && M->has_n(k)) e = g_rccl.Send(M->send_n[k], M->xbytes, kNcclChar, N, M->comm[k], X[k]);
            if (!e && M->has_n(k)) e = g_rccl.Recv(M->recv_n[k], M->xbytes, kNcclChar, N, M->comm[k], X[k]);
            if (!e && M->has_s(k)) e = g_rccl.Recv(M->recv_s[k], M->xbytes, kNcclChar, S, M->comm[k], X[k]);
        }
        const int ge = g_rccl.GroupEnd();
        if (e || ge) { m_err(errm, errm_len, "RCCL ghost exchange failed: %s", g_rccl.GetErrorString(e ? e : ge)); return -300 - (e ? e : ge); }
    } else if (M->transport == BEOM_XCHG_SHM) {
        // the same place in the same stream order as the grouped send/recv above (see ShmXchg)
        ShmXchg *x = M->shm;
        const int k = 0;
        const uint64_t sq = ++x->seq;
        const int me = M->band[k].index;
        M_HIP(hipSetDevice(M->dev[k]));
        if (M->has_s(k)) M_HIP(hipMemcpyAsync(x->slot(me, 0, sq), M->send_s[k], M->xbytes, hipMemcpyDeviceToHost, X[k]));
        if (M->has_n(k)) M_HIP(hipMemcpyAsync(x->slot(me, 1, sq), M->send_n[k], M->xbytes, hipMemcpyDeviceToHost, X[k]));
        M_HIP(hipLaunchHostFunc(X[k], shm_publish, new ShmOp{x, me, sq}));
        if (M->has_n(k)) {          // what my north neighbour sent south (looped back: what I sent myself)
            const int q = M->north_of(k), dir = M->loopback ? (M->has_s(k) ? 0 : 1) : 0;
            M_HIP(hipLaunchHostFunc(X[k], shm_wait, new ShmOp{x, q, sq}));
            M_HIP(hipMemcpyAsync(M->recv_n[k], x->slot(q, dir, sq), M->xbytes, hipMemcpyHostToDevice, X[k]));
        }
        if (M->has_s(k)) {
            const int q = M->south_of(k), dir = M->loopback ? (M->has_n(k) ? 1 : 0) : 1;
            M_HIP(hipLaunchHostFunc(X[k], shm_wait, new ShmOp{x, q, sq}));
            M_HIP(hipMemcpyAsync(M->recv_s[k], x->slot(q, dir, sq), M->xbytes, hipMemcpyHostToDevice, X[k]));
        }
    }
    for (int k = 0; k < n; ++k) {
        const Band &s = M->band[k];
        M_HIP(hipSetDevice(M->dev[k]));
        struct Restore { beom_multi *M; int k; ~Restore() { (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0); } } restore{M, k};
        (void)beom_set_stream(M->eng[k], (void *)X[k], 0);
        if (M->has_s(k)) {
            if (M->transport == BEOM_XCHG_PEER) {
                const int q = M->local_of(M->south_of(k));
                if (q != k) M_HIP(hipStreamWaitEvent(X[k], M->packed[q], 0));
                M_HIP(hipMemcpyPeerAsync(M->recv_s[k], M->dev[k], M->send_n[q], M->dev[q], M->xbytes, X[k]));
            }
            if (!M->has_n(k) && beom_unpack_rows(M->eng[k], 1, kGhost, M->recv_s[k])) { m_err(errm, errm_len, "beom_unpack_rows failed"); return -3; }
        }
        if (M->has_n(k)) {
            if (M->transport == BEOM_XCHG_PEER) {
                const int q = M->local_of(M->north_of(k));
                if (q != k) M_HIP(hipStreamWaitEvent(X[k], M->packed[q], 0));
                M_HIP(hipMemcpyPeerAsync(M->recv_n[k], M->dev[k], M->send_s[q], M->dev[q], M->xbytes, X[k]));
            }
            if (!M->has_s(k) && beom_unpack_rows(M->eng[k], s.gs + s.nown() + 1, kGhost, M->recv_n[k])) { m_err(errm, errm_len, "beom_unpack_rows failed"); return -3; }
        }
        if (M->has_s(k) && M->has_n(k) && beom_unpack_rows2(M->eng[k], kGhost, 1, M->recv_s[k], s.gs + s.nown() + 1, M->recv_n[k])) {
            m_err(errm, errm_len, "beom_unpack_rows2 failed"); return -3;       // both sides in one launch, behind both transfers
        }
        if (X[k] != M->main_s[k] || M->transport == BEOM_XCHG_PEER) {
            M_HIP(hipEventRecord(M->landed[k], X[k]));
            M->pending[k] = 1;
        }
    }
    return 0;
}

extern "C" {

int beom_multi_step(beom_multi_handle M, int tstp_first, int nsteps, double tres, double dtd8, double dt_r,
                    double rsta, int n_3d, char *errm, int errm_len) {
    if (!M) { m_err(errm, errm_len, "null handle"); return -1; }
    if (M->failed) { m_err(errm, errm_len, "beom_multi_step: an earlier step failed half way; destroy the handle"); return -30; }
    if (tstp_first < 1 || nsteps < 0 || n_3d < 1) { m_err(errm, errm_len, "beom_multi_step: bad arguments"); return -3; }
    if (M->nb == 1 && !M->ring) return beom_step(M->eng[0], tstp_first, nsteps, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len);
    for (int t = tstp_first; t < tstp_first + nsteps; ++t) {
        const int rc = multi_one_step(M, t, tres, dtd8, dt_r, rsta, n_3d, errm, errm_len);
        if (rc) {       // streams and events are in an unknown order: refuse further steps, keep destroy safe
            M->failed = true;
            for (int k = 0; k < M->n; ++k) { (void)beom_set_stream(M->eng[k], (void *)M->main_s[k], 0); M->pending[k] = 0; }
            return rc;
        }
    }
    return 0;
}

}  // extern "C"
