/*
 * beom_hip.h — C-ABI of the MI355X-native BEOM time-step engine (libbeom_hip.so).
 *
 * The reference (zhazorken/beom) has no plugin/FFI interface: `program main` calls
 * `run()` (main.f95:34) and the five hot routines communicate through module-global
 * arrays (private_mod.f95:27-93).  The cut is therefore made one level above the hot
 * routines, at the time step.  Each entry point below names the
 * reference code it replaces.  Every array argument is a plain host pointer with the
 * exact Fortran storage of the corresponding module array, index 0 (land sentinel)
 * included:
 *
 *   X(0:ndeg, nlay)        -> x[ipnt + (ndeg+1)*(ilay-1)]
 *   neig(8, 0:ndeg)        -> neig[(k-1) + 8*ipnt]           default integer (4 B)
 *   rs_h(2, 0:ndeg, nlay)  -> rs_h[(m-1) + 2*(ipnt + (ndeg+1)*(ilay-1))]
 *   dmdx(3, 0:ndeg, nlay)  -> dmdx[(m-1) + 3*(ipnt + (ndeg+1)*(ilay-1))]
 *   fnud(0:ndeg, nlay, 3)  -> fnud[ipnt + (ndeg+1)*((ilay-1) + nlay*(ivar-1))]
 *   nudg(0:ndeg, 3)        -> nudg[ipnt + (ndeg+1)*(ivar-1)]
 *   tide(2, 1, 0:ndeg, 3)  -> tide[(m-1) + 2*(ipnt + (ndeg+1)*(ivar-1))]
 *   tt3d(0:ndeg, 2, nlay)  -> tt3d[ipnt + (ndeg+1)*((idir-1) + 2*(ilay-1))]
 *   bodf(nlay, 2)          -> bodf[(ilay-1) + nlay*(idir-1)]
 *   taus(0:ndeg, 2)        -> taus[ipnt + (ndeg+1)*(idir-1)]
 *
 * Ownership: the caller owns every host array; the library copies at create/upload and
 * never keeps a host pointer.  Device memory belongs to the handle.
 * Errors: every function returns 0 on success and a negative code on failure, and
 * writes a NUL-terminated message into errm (capacity errm_len; may be NULL) — the
 * errc/errm convention of shared_mod.f95:113-157.  There is NO CPU fallback: without a
 * usable HIP device beom_create fails.
 * Threading: one host thread per handle, never from inside an OpenMP region
 * (the reference calls the hot routines from the master thread, private_mod.f95:1867-1906).
 */
#ifndef BEOM_HIP_H
#define BEOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEOM_MAX_LAYERS 16
#define BEOM_ABI_VERSION 2

/* Constants of shared_mod.f95:41-99, passed BY VALUE from the host so that the
 * single-precision literals widened to double (grav = 9.8, beta = 0.281105, ...) keep
 * the host compiler's bits.  Fortran side: type, bind(C) :: beom_params. */
typedef struct beom_params {
    int32_t abi_version;        /* BEOM_ABI_VERSION                                  */
    int32_t lm, mm, nlay, ndeg; /* shared_mod.f95:41-45                              */
    int32_t nsal;               /* shared_mod.f95:105 Salmon exponent (4)            */
    int32_t variant;            /* 0: update_h of private_mod.f95:1593-1702;
                                   1: epilogue of private_mod3d.f95:1635-1683        */
    int32_t flag_nudging;       /* private_mod.f95:93,868-871                        */
    int32_t dense_hint;         /* 1: let the library verify neig against the dense
                                   closed form and use the fast path */
    int32_t slab_row0;          /* j-slab of a larger frame (multi-GPU): the
                                   handle's row 1 is global row slab_row0 + 1 ...        */
    int32_t slab_mm;            /* ... of a frame with this many rows (global mm);
                                   0 = the handle holds the whole frame                  */
    double dl, dt;              /* shared_mod.f95:47,84                              */
    double grav, rho0;          /* :89-90                                            */
    double beta, epsi, gamm, del1, del2; /* :91-95                                   */
    double hmin, hsal;          /* :59,85                                            */
    double bvis, dvis, svis;    /* :56-57 (+fork's svis)                             */
    double bdrg, tdrg, qdrg;    /* :58,65 (+fork's tdrg)                             */
    double hsbl, hbbl;          /* :60-61                                            */
    double g_fb, uadv, ocrp, rgld, mcbc; /* :63-72                                   */
    double invf;                /* private_mod.f95:223-229                           */
    double w_ti;                /* private_mod.f95:85,953  tidal frequency (rad/day) */
    double rhon[BEOM_MAX_LAYERS]; /* shared_mod.f95:50                               */
} beom_params;

typedef struct beom_engine *beom_handle;

/* Version/ABI probe (no GPU needed). */
int beom_abi_version(void);
/* sha1 prefix of the sources the library was built from (Makefile); bindings compare it with the tree. */
const char *beom_source_hash(void);
/* Number of visible HIP devices, or a negative error code. */
int beom_device_count(char *errm, int errm_len);
/* PCI address "dddd:bb:dd.f" of a HIP device (to find its sysfs node for clock / power read-outs). */
int beom_device_pci_bus_id(int device, char *out, int out_len);

/* Replaces the static part of the module state built by read_input_data
 * (private_mod.f95:105-250): connectivity (index_grid_points :567-764), masks,
 * Coriolis, depth, and the optional forcings of read_input_file (:766-967).
 * hdot, tide, bodf, taus, h_to may be NULL (= all zero, i.e. file absent). */
int beom_create(const beom_params *prm, int device,
                const int32_t *neig, const int32_t *subc,
                const double *mk_u, const double *mk_v, const double *mk_n,
                const double *mkpe, const double *mkpi,
                const double *fcor, const double *h_th, const double *h_to,
                const double *nudg, const double *fnud, const double *hdot,
                const double *tide, const double *bodf, const double *taus,
                beom_handle *out, char *errm, int errm_len);

int beom_destroy(beom_handle h);

/* Replaces the product of index_boundary_points (private_mod.f95:1060-1240): segm(nseg, 18),
 * Fortran storage segm[(iseg-1) + nseg*(col-1)], default integers.  With flag_nudging and
 * mcbc < 0.5 the engine then applies no_gradient_obc (:2613-2679) after the momentum sweeps of
 * every step (:2201-2204, 2285-2288); beom_step refuses such a configuration until this call
 * has been made.  nseg = 0 (segm NULL) declares that the handle holds no segment; a pass of a segment whose updated cell
 * (column 10 for the first pass, 1 for the second) is -1 is skipped — both are what beom_multi_set_open_boundaries gives
 * the bands of a frame. */
int beom_set_open_boundaries(beom_handle h, int nseg, const int32_t *segm, char *errm, int errm_len);

/* Prognostic + history state, host -> device.  Any pointer may be NULL (left as is;
 * a fresh handle holds the values of initialize_variables, private_mod.f95:252-307). */
int beom_upload_state(beom_handle h,
                      const double *hlay, const double *u, const double *v,
                      const double *h_u, const double *h_v,
                      const double *rs_h, const double *dmdx, const double *dmdy,
                      const double *v_cc, const double *v_ll,
                      const double *tt3d, const double *tb3d, const double *tu3d,
                      char *errm, int errm_len);

/* Device -> host; any pointer may be NULL.  Needed before write_outputs
 * (private_mod.f95:1908-1910) and for restart/parity checks. */
int beom_download_state(beom_handle h,
                        double *hlay, double *u, double *v, double *h_u, double *h_v,
                        double *rs_h, double *dmdx, double *dmdy,
                        double *v_cc, double *v_ll,
                        double *tt3d, double *tb3d, double *tu3d,
                        char *errm, int errm_len);

/* Output preparation on the device: replaces the array work of write_array for 'eta_', 'u___',
 * 'v___' (private_mod.f95:2848-2883; eta = interface elevation accumulated bottom-up in real*4)
 * and the scans of write_outputs (:2772-2808).  h0r4 = the (ndeg, nlay) real*4 content of
 * h_0.bin (needed on the first call, may be NULL afterwards); eta/u4/v4 = (ndeg, nlay) real*4
 * host records (any may be NULL); minmax[nlay][6] = min/max of h over wet cells, of u and of v
 * over their points (over all cells 0..ndeg if a mask is empty, :2776-2793); *thin_layer = first
 * layer with a wet cell thinner than 0.5*hmin, or 0 (:2798-2808). */
int beom_download_outputs(beom_handle h, const float *h0r4, float *eta, float *u4, float *v4,
                          double *minmax, int *thin_layer, char *errm, int errm_len);

/* The `diag` records of write_array (private_mod.f95:2884-2974), formed on the device from the state: 'pvor', 'mont'
 * (without the kinetic part, real*4 accumulation as in the reference) and 'v_cc' (Leith viscosity diagnosed from u, v) —
 * each (ndeg, nlay) real*4, any pointer may be NULL.  Call between time steps (uses the step's scratch arrays). */
int beom_download_diag(beom_handle h, float *pvor, float *mont, float *v_cc, char *errm, int errm_len);

/* Rigid lid (rgld = 1; the fork's addition, private_mod.f95:64-67, 91, 505-563, 1648-1700, 1705-1838, 2207-2221,
 * 2237-2257, 2292-2314).  A handle created with prm->rgld = 1 (needs variant 0, ocrp = 1 — the reference only
 * initialises the operators then — and a whole frame) steps as the reference does: transports rebuilt before update_h
 * and after the momentum sweeps, the column-misfit epilogue of update_h, then surf_pressure — the Poisson right-hand
 * side, Gauss-Seidel sweeps in packed order until max|change| <= 1e-5 or 1000 sweeps (on the device as wavefronts over
 * the anti-diagonals, same arithmetic), and the velocity correction.  beom_set_rigid_lid uploads the module arrays
 * Ow, Os, Osum_ (0:ndeg) (first call: required) and the lid pressure pi_s(0:ndeg) (NULL: keep; zero after create);
 * beom_step refuses (-6) until it has been called.  beom_download_pressure returns pi_s; beom_download_outputs puts
 * real(pi_s) into the top 'eta_' record as write_array does (:2864-2872). */
int beom_set_rigid_lid(beom_handle h, const double *Ow, const double *Os, const double *Osum_, const double *pi_s,
                       char *errm, int errm_len);
int beom_download_pressure(beom_handle h, double *pi_s, char *errm, int errm_len);

/* The six per-layer diagnostics of update_mont_rvor_pvor_dive_kine, which the library
 * keeps per layer: each is (0:ndeg, nlay).  (The reference keeps (0:ndeg) and reuses it
 * layer after layer, private_mod.f95:48-61.) */
int beom_download_scratch(beom_handle h,
                          double *mont, double *rvor, double *pvor, double *dive,
                          double *d2hx, double *d2hy, char *errm, int errm_len);

/* Replaces the body of integrate_time (private_mod.f95:1853-1912) for time steps
 * tstp_first .. tstp_first+nsteps-1: ctim (:1862,1887), ramp (:1864-1866,1898-1901),
 * gene (:1859,1877), distribute_stress cadence (:1863,1889-1896), and per step
 * first_three_timesteps (:2151-2223) for tstp <= 3 or gener_forward_backward
 * (:2225-2316) afterwards.  Asynchronous on the handle's stream; beom_sync to wait. */
int beom_step(beom_handle h, int tstp_first, int nsteps,
              double tres, double dtd8, double dt_r, double rsta, int n_3d,
              char *errm, int errm_len);

int beom_sync(beom_handle h, char *errm, int errm_len);

/* Multi-GPU j-slabs (no reference counterpart — the reference is OpenMP only).
 * One time step of a band (a handle with slab_mm != 0) in three parts, "boundary first", so that the rows its neighbours
 * wait for are finished, packed and on their way while the bulk of the momentum sweep still runs:
 *   phase 1 = the step up to the momentum sweeps on all rows (stress, the transports of steps 1-3, update_h,
 *             update_mont..., update_viscosity);
 *   phase 2 = update_u/update_v (the fused sweep) on the strips next to the ghost zones: rows 1..8 and M-7..M — call it
 *             with ANOTHER stream set (beom_set_stream) that waits for phase 1, and follow it there with
 *             beom_pack_rows -> transport -> beom_unpack_rows;
 *   phase 3 = the same sweep on the rows in between, on the usual stream, then the pointer rotations (call it last).
 * All three see ghost rows that have landed; the next step may start once this step's unpack has run.
 * -20 (from phase 1) = this step cannot be cut that way (separate u and v sweeps, open-boundary passes, rigid lid, fewer
 * than 32 rows): use beom_step. */
int beom_step_phase(beom_handle h, int tstp, double tres, double dtd8, double dt_r, double rsta,
                    int n_3d, int phase, char *errm, int errm_len);
/* Rows [jlo, jlo+nrows) (local, 1-based) of hlay,u,v,h_u,h_v <-> one contiguous DEVICE buffer
 * of 5*nlay*nrows*(lm+1) doubles ([field][layer][row][column]); one launch each, on the
 * handle's stream. */
int beom_pack_rows(beom_handle h, int jlo, int nrows, void *device_buffer);
int beom_unpack_rows(beom_handle h, int jlo, int nrows, const void *device_buffer);
/* two groups of nrows rows (a band's south and north side) <-> two buffers in ONE launch */
int beom_pack_rows2(beom_handle h, int nrows, int jlo_a, void *buffer_a, int jlo_b, void *buffer_b);
int beom_unpack_rows2(beom_handle h, int nrows, int jlo_a, const void *buffer_a, int jlo_b, const void *buffer_b);

/* Options (dense frames only; results are bit-identical either way):
 *  "fuse_mont_visc" (default 1): with the Leith viscosity refreshed every step (dvis > 1e-3,
 *      n_3d = 1) update_mont... and update_viscosity run as ONE sweep that hands update_u/v
 *      the products v_cc*dive and v_ll*rvor; v_cc, v_ll, rvor, dive are then kept up to date
 *      only with "keep_diag" = 1 (default 0).  With dvis <= 1e-3 (and svis = 0) the same sweep,
 *      from step 4 on, forms the products of the standing v_cc, v_ll instead (rvor, dive again
 *      only with "keep_diag"); with n_3d > 1 the refresh steps also store v_cc, v_ll and the
 *      steps in between use that standing-viscosity form.
 *  "lean_d2h", "lean_visc" (default 1): the fused pair re-derives d2hx, d2hy in the momentum sweep,
 *      and drops the viscous products altogether when v_cc = v_ll = +0 and never refreshed.
 *  "fuse_uv" (default 1): update_u and update_v of a step run as ONE sweep.
 *  "fuse": sets both.  0 = always five separate sweeps.
 *  "profile_stride" (default 1): beom_profile_start brackets only the steps with tstp % stride == 0 with HIP events
 *      (an event pair between two launches costs a few microseconds of pipeline bubble: sampled, the timed region is
 *      hardly disturbed; the launch counts beom_profile_stop returns are those of the sampled steps).
 *  "profile_rotate" (default 0): a sampled step brackets only ONE kind of sweep — update_h | update_mont, update_viscosity |
 *      update_u, update_v — by turns ((tstp / stride) % 3 = 0 | 1 | 2), so that no bracketed launch starts behind another
 *      bracket's bubble; the launch counts are then those of the steps that bracketed that kind.
 *  "fold_stress" (default 1): with constant layer fractions (ocrp = 0) and a stress refresh on every step (n_3d = 1), steps
 *      after the third form distribute_stress (private_mod.f95:1921-2149) inside the fused update_u/update_v sweep: no
 *      launch of its own, tt3d/tb3d/tu3d neither written nor read (they keep their values of step 1 unless "keep_diag" = 1).
 * Returns -3 for an unknown name. */
int beom_set_option(beom_handle h, const char *name, int value);
/* Introspection (>= 0, or -3 for an unknown name): "stress_folded" = the last step formed its stress inside the momentum
 * sweep (1) or through distribute_stress' own launch and the three arrays (0); "tile_rows" = rows of a tile of the tiled
 * sweeps (8 | 4; 0 on the table path). */
int beom_info(beom_handle h, const char *what);

/* Run all launches of this handle on the caller's HIP stream (e.g. the stream a
 * ghost-row exchange is enqueued on).  hip_stream may be NULL = the default stream;
 * use_own != 0 restores the handle's own stream. */
int beom_set_stream(beom_handle h, void *hip_stream, int use_own);

/* Per-sweep entry points with the reference routines' meaning; used by parity tests.
 * ilay is 1-based as in Fortran; ilay = 0 means "all layers" (one batched launch).
 * The per-step scalars gene/ramp/ctim are module variables in the reference
 * (private_mod.f95:70-73) and are passed explicitly here. */
int beom_update_h(beom_handle h, double gene, double ramp, double ctim);          /* :1593 */
int beom_update_mont_rvor_pvor_dive_kine(beom_handle h, int ilay);               /* :2318 */
int beom_update_viscosity(beom_handle h, int ilay);                              /* :2441 */
int beom_update_u(beom_handle h, int ilay, double gene, double ramp, double ctim); /* :1422 */
int beom_update_v(beom_handle h, int ilay, double gene, double ramp, double ctim); /* :1505 */
int beom_rebuild_fluxes(beom_handle h);                                          /* :2166-2177 */
int beom_distribute_stress(beom_handle h);                                       /* :1921 */

/* Introspection for measurement and zero-copy interop. */
/* name in {hlay,u,v,h_u,h_v}: device pointer + layout of the library's internal copy.
 * stride_layer/stride_row in elements; for the packed (gather) layout stride_row = 0. */
int beom_device_field(beom_handle h, const char *name, void **dptr,
                      int64_t *stride_layer, int64_t *stride_row, int64_t *row0_offset);
/* 1 if the dense fast path is active for this handle, else 0. */
int beom_is_dense(beom_handle h);
/* Per-kernel device time, measured with HIP events on the handle's stream around every
 * sweep launched by beom_step between start and stop (no host synchronisation in
 * between): ms[0..7] = update_h, update_mont, update_viscosity, update_u, update_v,
 * fused mont+viscosity, fused u+v, (unused) (sums over launches), launches[0..7] = number of
 * launches in each class.  Both arrays need 8 entries. */
int beom_profile_start(beom_handle h);
int beom_profile_stop(beom_handle h, double *ms, int *launches, char *errm, int errm_len);
/* = beom_profile_start; beom_step(...); beom_profile_stop. */
int beom_profile_steps(beom_handle h, int tstp_first, int nsteps,
                       double tres, double dtd8, double dt_r, double rsta, int n_3d,
                       double *ms, int *launches, char *errm, int errm_len);

/* ---- Several GPUs: the frame cut into bands of rows (SURVEY §8b "Threading", §8e) -------------
 * No reference counterpart (the reference is OpenMP only).  The whole DENSE frame (ndeg = (lm+1)(mm+1))
 * is cut into bands of rows, one band per HIP device, each an ordinary slab handle with 4 ghost rows
 * per neighbour; per time step ONE exchange of hlay,u,v,h_u,h_v (beom_pack_rows -> transport ->
 * beom_unpack_rows on the band's second stream) inside the interior rows of the step's own momentum sweep
 * (beom_step_phase); the steps of one beom_multi_step call run inside the library.  Results are bit-identical to the single handle.
 * Frames periodic in y: the bands form a ring over rows 1..mm and row mm+1 (which nothing points to but
 * every record contains, private_mod.f95:642-668) is carried by a small companion frame next to band 0.
 *
 * Transports of the exchange: */
#define BEOM_XCHG_PEER 0     /* hipMemcpyPeerAsync between the bands of ONE process (xGMI peer copies)      */
#define BEOM_XCHG_RCCL 1     /* grouped ncclSend/ncclRecv (librccl.so.1, bound at run time): all bands in one
                                process (ncclCommInitAll, distinct devices) or one band per process          */
#define BEOM_XCHG_SHM  2     /* one band per process, the processes on ONE node: the packed ghost rows are staged through a
                                POSIX shared-memory segment, in the same stream order as the RCCL send/recv (device -> segment,
                                publish; wait for the neighbour, segment -> device).  Unlike RCCL it lets several ranks share
                                a device: the -m gpu tests run 2 and 3 ranks of beom_multi_create_local on one GPU this way  */
#define BEOM_XCHG_LOOPBACK 0x200 /* flag for beom_multi_create_local_ex: the band receives what it sends itself (its own edge
                                rows arrive as its ghost rows).  ONE band of a frame cut nb ways then runs alone with the
                                whole exchange machinery — a timing rehearsal; the values are not the frame's              */
#define BEOM_XCHG_RING1 0x100 /* flag for beom_multi_create_ex: cut a frame periodic in y as a ring even when
                                there is ONE band (it then exchanges with itself; exercises the ring form)   */

typedef struct beom_multi *beom_multi_handle;

/* no_gradient_obc (mcbc = 0) on a frame cut into bands: beom_set_open_boundaries' table with GLOBAL cell indices; the
 * library deals the segments to the bands (a pass of a segment goes to every band whose rows hold both its updated and its
 * source cell).  Handles created from the global arrays (beom_multi_create[_ex]) of a frame not periodic in y; steps of
 * such a handle are not split (the exchange follows the whole step). */
int beom_multi_set_open_boundaries(beom_multi_handle m, int nseg, const int32_t *segm, char *errm, int errm_len);

/* (a) From GLOBAL arrays, all bands in this process — arguments as beom_create / beom_upload_state /
 * beom_download_state / beom_step.  Band k runs on HIP device devices[k] (with peer copies the same device may
 * be named more than once).  This is what the Fortran host under main.f95 uses (BEOM_NGPU); beom_multi_create
 * takes the transport from the environment (BEOM_XCHG=rccl), default peer copies. */
int beom_multi_create(const beom_params *prm, int ndev, const int *devices,
                      const int32_t *neig, const int32_t *subc,
                      const double *mk_u, const double *mk_v, const double *mk_n,
                      const double *mkpe, const double *mkpi,
                      const double *fcor, const double *h_th, const double *h_to,
                      const double *nudg, const double *fnud, const double *hdot,
                      const double *tide, const double *bodf, const double *taus,
                      beom_multi_handle *out, char *errm, int errm_len);
int beom_multi_create_ex(const beom_params *prm, int ndev, const int *devices, int transport_and_flags,
                         const int32_t *neig, const int32_t *subc,
                         const double *mk_u, const double *mk_v, const double *mk_n,
                         const double *mkpe, const double *mkpi,
                         const double *fcor, const double *h_th, const double *h_to,
                         const double *nudg, const double *fnud, const double *hdot,
                         const double *tide, const double *bodf, const double *taus,
                         beom_multi_handle *out, char *errm, int errm_len);
int beom_multi_destroy(beom_multi_handle h);
int beom_multi_count(beom_multi_handle h);       /* bands of this process */
/* band k: owned global rows own0..own1; local window = rows win0..win1 (1-based, inclusive; in a ring
 * win0 <= 0 and win1 > mm denote wrapped ghost rows); device */
int beom_multi_band(beom_multi_handle h, int k, int *own0, int *own1, int *win0, int *win1, int *device);
int beom_multi_upload_state(beom_multi_handle h,
                            const double *hlay, const double *u, const double *v,
                            const double *h_u, const double *h_v,
                            const double *rs_h, const double *dmdx, const double *dmdy,
                            const double *v_cc, const double *v_ll,
                            const double *tt3d, const double *tb3d, const double *tu3d,
                            char *errm, int errm_len);
int beom_multi_download_state(beom_multi_handle h,
                              double *hlay, double *u, double *v, double *h_u, double *h_v,
                              double *rs_h, double *dmdx, double *dmdy,
                              double *v_cc, double *v_ll,
                              double *tt3d, double *tb3d, double *tu3d,
                              char *errm, int errm_len);
/* as beom_download_outputs / beom_download_diag: GLOBAL (ndeg, nlay) real*4 records, every band forming its own rows on
 * its device (h_0 is needed on every call here); global-array handles only */
int beom_multi_download_outputs(beom_multi_handle h, const float *h0r4, float *eta, float *u4, float *v4,
                                double *minmax, int *thin_layer, char *errm, int errm_len);
int beom_multi_download_diag(beom_multi_handle h, float *pvor, float *mont, float *v_cc, char *errm, int errm_len);
int beom_multi_step(beom_multi_handle h, int tstp_first, int nsteps,
                    double tres, double dtd8, double dt_r, double rsta, int n_3d,
                    char *errm, int errm_len);
int beom_multi_sync(beom_multi_handle h, char *errm, int errm_len);
/* how many band-steps were cut boundary first (exchange inside the momentum sweep) and how many ran in one piece */
int beom_multi_stats(beom_multi_handle h, long long *split_band_steps, long long *plain_band_steps);
/* "overlap" (default 1; 0 = every step runs in one piece, the exchange after it); other names go to every band */
int beom_multi_set_option(beom_multi_handle h, const char *name, int value);
int beom_multi_describe(beom_multi_handle h, int *bands_total, int *bands_local, int *transport, int *ring,
                        int *rccl_version);
/* the slab handle of local band k (k = -1: the companion frame of a ring, NULL if it is not here);
 * owned by the multi handle — for introspection (beom_device_field, beom_set_option), not for stepping */
int beom_multi_engine(beom_multi_handle h, int k, beom_handle *out);
/* as beom_profile_start/stop: per sweep class the slowest local band */
int beom_multi_profile_start(beom_multi_handle h);
int beom_multi_profile_stop(beom_multi_handle h, double *ms, int *launches, char *errm, int errm_len);

/* (b) ONE band per process from that band's WINDOW only (bench.py under torchrun: one process per GPU; nothing
 * of global size exists on any rank).  prm describes the GLOBAL frame; xper/yper say whether it is periodic
 * (with global arrays the library reads that off neig).  The window's rows, south to north:
 *     ghost_s ghost rows | the owned rows own0..own1 | ghost_n ghost rows        (beom_multi_window)
 * (in a ring the ghosts of the first and the last band wrap); every array has the storage of the
 * corresponding beom_create / beom_upload_state argument for a frame of that many rows — index 0 = sentinel,
 * then rows*(lm+1) cells.  No connectivity or mask tables: bands are dense frames, the library generates them.
 * Band 0 of a frame periodic in y also passes row mm+1 ("orphan": index 0 + (lm+1) cells per array).
 * Exchange over RCCL: rccl_id = the 128 bytes beom_rccl_unique_id returned on ONE rank, distributed by the caller
 * (bench.py: through torch.distributed's store); may be NULL when nb = 1. */
typedef struct beom_statics {
    const double *fcor, *h_th, *h_to, *nudg, *fnud, *hdot, *tide, *bodf, *taus;   /* h_to, hdot, tide, bodf, taus may be NULL */
} beom_statics;
typedef struct beom_state {                                                        /* any pointer may be NULL */
    double *hlay, *u, *v, *h_u, *h_v, *rs_h, *dmdx, *dmdy, *v_cc, *v_ll, *tt3d, *tb3d, *tu3d;
} beom_state;
int beom_rccl_unique_id(void *id128, char *errm, int errm_len);
int beom_rccl_version(char *errm, int errm_len);        /* e.g. 22204, or a negative error code */
int beom_multi_window(const beom_params *prm, int nb, int band, int yper,
                      int *own0, int *own1, int *ghost_s, int *ghost_n);
int beom_multi_create_local(const beom_params *prm, int nb, int band, int device, int xper, int yper,
                            const void *rccl_id, const beom_statics *window, const beom_statics *orphan,
                            beom_multi_handle *out, char *errm, int errm_len);
/* The same with the transport named: BEOM_XCHG_RCCL (xchg_id = the 128-byte unique id, as above) or BEOM_XCHG_SHM
 * (xchg_id = NUL-terminated name "/..." of a shared-memory segment, the same on all nb ranks and not in use by any other
 * job; created by whoever comes first, removed again once all nb ranks have attached), optionally | BEOM_XCHG_LOOPBACK. */
int beom_multi_create_local_ex(const beom_params *prm, int nb, int band, int device, int xper, int yper,
                               int transport_and_flags, const void *xchg_id,
                               const beom_statics *window, const beom_statics *orphan,
                               beom_multi_handle *out, char *errm, int errm_len);
/* no_gradient_obc (mcbc = 0) for a handle that holds one band's window: the segments of the window's own rows, as
 * beom_set_open_boundaries takes them, with cell indices of the window (the finder of private_mod.f95:1060-1240 looks at a
 * cell and its four neighbours only, so a rank finds them from its rows alone); band 0 of a frame periodic in y adds those of
 * the orphan row mm+1 (indices of a one-row frame; else 0 / NULL).  nseg = 0: this band holds no segment. */
int beom_multi_set_open_boundaries_local(beom_multi_handle h, int nseg, const int32_t *segm, int nseg_orphan, const int32_t *segm_orphan,
                                         char *errm, int errm_len);
int beom_multi_upload_local(beom_multi_handle h, const beom_state *window, const beom_state *orphan,
                            char *errm, int errm_len);
int beom_multi_download_local(beom_multi_handle h, beom_state *window, beom_state *orphan,
                              char *errm, int errm_len);

#ifdef __cplusplus
}
#endif
#endif /* BEOM_HIP_H */
