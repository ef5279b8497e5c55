/*
 * physicl_hip.h -- C ABI of libphysicl_hip.so: the MI355X (gfx950) implementation of PhysiCL's
 * per-particle time-step hot path.
 *
 * This is the drop-in boundary.  The reference crosses it in ONE place: ``CLProgram.run``
 * (physicl/__init__.py:602-664) and the hand-rolled launch in
 * ``ScatterDeleteStepReference.__run_cl`` (physicl/light.py:164-205) gather per-object Python
 * attributes into numpy arrays, copy them to an OpenCL device, launch one kernel and copy the
 * result back.  A PhysiCL maintainer binds the functions below with ctypes instead of PyOpenCL
 * (see INTEGRATION.md).  Two levels are offered:
 *
 *   Level 1  "reference-ABI kernels"  (pcl_k_*): same argument lists as the three OpenCL kernels
 *            the reference builds at run time, on raw device pointers.  One call == one
 *            ``prog.<kernel>(queue, (N,), None, *args)``.
 *   Level 2  "particle store" (pcl_store_*, pcl_step_*): particles stay resident in HBM (one tiled
 *            slab, see pcl_store_alloc) for the whole simulation; a Step, a whole loop body
 *            (pcl_step_fused, pcl_step_fused_delete) or K loop bodies (pcl_step_fused_multi,
 *            pcl_step_fused_delete_multi) are one pass over it; replaces gather + H2D + launch + D2H +
 *            Python write-back.
 *
 * Conventions: plain C types only; every function returns 0 (PCL_OK) or a negative PCL_ERR_* code
 * and never throws; pcl_last_error() gives the calling thread's last message.  Unless a parameter
 * says "host", pointers are DEVICE pointers.  Every entry point binds its context's device to the
 * calling thread first (the reference creates its context on one thread and launches from the
 * Simulation thread, physicl/__init__.py:427-429, 501), so calls may come from any thread; at most
 * one call per context may be in flight at a time.  All work is enqueued on the context's stream;
 * functions that return values to the host synchronise that stream, the others do not.
 */
#ifndef PHYSICL_HIP_H
#define PHYSICL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCL_ABI_VERSION 1

#define PCL_OK          0
#define PCL_ERR_HIP    (-1) /* a HIP runtime call or kernel launch failed */
#define PCL_ERR_ARG    (-2) /* bad argument (null pointer, negative size, unknown enum) */
#define PCL_ERR_STATE  (-3) /* call not valid in the context's current state */
#define PCL_ERR_RTC    (-4) /* hipRTC compile/load failure; log is in pcl_last_error() */
#define PCL_ERR_EXPR   (-5) /* variable_n_fn expression rejected by the validator */
#define PCL_ERR_NOMEM  (-6) /* device or host allocation failed */

/* Per-particle state = the fields of physicl.Object / PhotonObject that a Step reads or writes
 * (physicl/__init__.py:390-394, physicl/light.py:26-35), one row per component, of the store's dtype. */
enum pcl_field {
    PCL_R0 = 0, PCL_R1, PCL_R2,      /* Object.r   position                    */
    PCL_V0, PCL_V1, PCL_V2,          /* Object.v   velocity                    */
    PCL_DR0, PCL_DR1, PCL_DR2,       /* Object.dr  last displacement           */
    PCL_DV0, PCL_DV1, PCL_DV2,       /* Object.dv  last velocity change        */
    PCL_E,                           /* PhotonObject.E                         */
    PCL_NFIELDS
};

/* element type of a particle store.  The reference is fp64-only (every input is marshalled as
 * np.double, physicl/__init__.py:613): PCL_DTYPE_F64 is the parity path; PCL_DTYPE_F32 exists for the
 * precision sweep of BASELINE.json configs[4]. */
#define PCL_DTYPE_F64 0
#define PCL_DTYPE_F32 1

/* pcl_step_scatter_isotropic / pcl_k_light_scatter_step_sphere ``flags`` */
#define PCL_SCATTER_WAVELENGTH 1 /* wavelength_dep_scattering=True  (light.py:275, 300-301) */
#define PCL_SCATTER_VARIABLE_N 2 /* variable_n=True                  (light.py:276, 299)     */
#define PCL_FUSED_LAZY         4 /* pcl_step_fused only: leave dr and dv implicit (see there)  */
#define PCL_SCATTER_PY_DV      8 /* pcl_step_scatter_isotropic only: a hit leaves dv = v_old, the write-back of the
                                    reference's CPU path ScatterIsotropicStep.__run_py (light.py:346-348)       */

/* where a step's three random numbers per photon come from */
#define PCL_RNG_INPUT  0 /* arrays uploaded with pcl_store_upload_rand: the reference's contract
                            ("randoms are kernel inputs", light.py:285, __init__.py:606-619)       */
#define PCL_RNG_PHILOX 1 /* generated in-kernel: Philox4x32-10 keyed by (seed, step, particle id):
                            decision block (id, step >> 1, 0) shared by two steps, direction block
                            (id, step, 1) on a hit (DESIGN.md "Device RNG")                         */

/* kind[] values (pcl_store_upload_kind) */
#define PCL_KIND_OBJECT 0 /* plain physicl.Object: moved by Newton, skipped by the light steps
                             (``if type(obj) != PhotonObject: continue``, light.py:233, 283)      */
#define PCL_KIND_PHOTON 1

/* index layout of pcl_step_counters() output */
#define PCL_CNT_N      0 /* particles in the store                                  */
#define PCL_CNT_XP     1 /* v_x > 0   (ScatterSignMeasureStep, light.py:424)         */
#define PCL_CNT_YP     2
#define PCL_CNT_ZP     3
#define PCL_CNT_PLANE0 4 /* first plane-crossing count (ScatterMeasureStep, light.py:385-399) */
#define PCL_MAX_PLANES 12

typedef struct pcl_ctx pcl_ctx; /* opaque */

/* ---------------------------------------------------------------- library / context ---------- */
int         pcl_abi_version(void);
const char *pcl_last_error(void);                 /* thread-local, never NULL */
int         pcl_device_count(int *n_out);         /* host pointer */
/* A/B switches ("knobs").  Every PCL_* environment variable the library reads at call time (the delete path's: PCL_ALIVE,
 * PCL_ALIVE_RATIO, PCL_ALIVE_MIN_SLOTS, PCL_ALIVE_POLL, PCL_ALIVE_FLUSH_KERNEL, PCL_COMPACT_SPARSE, PCL_AHEAD, PCL_AHEAD_K,
 * PCL_AHEAD_MAX_SLOTS, PCL_AHEAD_K_BIG, PCL_AHEAD_LIVE, PCL_MULTI_AHEAD; the K-step passes': PCL_MULTI_NQ2, PCL_MULTI_NQ3, PCL_MULTI_SAT,
 * PCL_MIXED_NE3, PCL_MIXED_INPLACE, PCL_MIXED_COMPACT_BELOW) can also be set from the program:
 * value = its text, NULL = back to the environment.  Process-wide, takes effect at the next call; results never depend
 * on a knob (that is what the tests that flip them check), only which formulation runs.                            */
int         pcl_set_knob(const char *name, const char *value);
/* Device blocks of >= 64 MB (stores, scratch, pcl_dev_alloc buffers) are not handed back to the driver when freed: the
 * process keeps up to PCL_POOL_GB (environment; default a third of the device's memory, 0 = off; without PCL_POOL_GB a freed
 * block is only kept while at least a quarter of the device stays free) of them for its next store of about that size --
 * a hipMalloc of tens of GB right after a hipFree of that size was measured to stall for seconds on this runtime.
 * pcl_pool_trim() releases everything the pool holds (bytes released in *released_out, may be NULL);
 * pcl_pool_bytes() tells how much it holds.                                                                          */
int         pcl_pool_trim(int64_t *released_out);
int         pcl_pool_bytes(int64_t *idle_out);
/* The same in detail (host pointers, any may be NULL): bytes of idle blocks; bytes of physical handles kept of ranges that were
 * unmapped (both count against PCL_POOL_GB); bytes of ADDRESS SPACE parked behind freed ranges -- on this runtime a new mapping at
 * addresses that were mapped before loses writes, so a freed range's addresses are reserved again at once and never mapped
 * (bounded at 32 TB of the 128 TB address space, then big blocks come from hipMalloc) --; how many times a fresh reservation was
 * found to overlap a formerly mapped range and was set aside; 1 while big blocks are still built with the virtual-memory API. */
int         pcl_pool_info(int64_t *idle_blocks_out, int64_t *idle_handles_out, int64_t *parked_va_out, int64_t *remaps_avoided_out,
                          int *vmm_on_out);

/* ``stream``: a hipStream_t to adopt (e.g. torch.cuda.current_stream().cuda_stream) or NULL to let
 * the context create its own non-blocking stream.  Replaces cl.create_some_context() +
 * cl.CommandQueue() (physicl/__init__.py:428-429). */
int pcl_ctx_create(int device, void *stream, pcl_ctx **ctx_out);
int pcl_ctx_destroy(pcl_ctx *ctx);
int pcl_ctx_sync(pcl_ctx *ctx);
int pcl_ctx_stream(pcl_ctx *ctx, void **stream_out);
/* name: host buffer of name_len bytes.  Replaces Simulation.get_device_info (__init__.py:470-499). */
int pcl_ctx_device_info(pcl_ctx *ctx, char *name, int name_len, int64_t *hbm_bytes, int *n_cu,
                        int *wavefront);
/* Free and total bytes of the context's device as the driver reports them (hipMemGetInfo; blocks idle in the library's
 * pool count as used -- add pcl_pool_bytes() for what the process could still get).  Either pointer may be NULL.        */
int pcl_ctx_mem_info(pcl_ctx *ctx, int64_t *free_out, int64_t *total_out);
/* PCI bus id of the context's device ("0000:05:00.0"; host buffer of >= 16 bytes): lets N ranks show that they
 * drive N different GPUs (bench.py "collective"). */
int pcl_ctx_device_pci(pcl_ctx *ctx, char *pci, int pci_len);
/* variable_n_fn texts of the three built-in shapes (the reference's examples) can start on the
 * ahead-of-time kernels at once while hipRTC compiles their specialisation on another thread (~2 s), which is taken over
 * at the first step call after it is ready -- same bits either way, and about the same speed (the ahead-of-time kernels
 * of the hot passes are compiled per shape and axis).
 * Off by default for a bare context (a measurement must not time the stand-in); physicl_amd.Simulation switches it on.
 * pcl_ctx_rtc_wait blocks until every pending specialisation of the context is in place (*pending_out: how many were).   */
int pcl_ctx_set_rtc_background(pcl_ctx *ctx, int on);
int pcl_ctx_rtc_wait(pcl_ctx *ctx, int *pending_out /* may be NULL */);

/* raw device memory for Level 1 callers: replaces cl_array.to_device / cl_array.empty / .get()
 * (physicl/__init__.py:614, 653, 662).  Copies are ordered on the context stream and return when
 * the host buffer may be reused (pcl_h2d) or holds the data (pcl_d2h). */
int pcl_dev_alloc(pcl_ctx *ctx, int64_t bytes, void **dev_out);
int pcl_dev_free(pcl_ctx *ctx, void *dev);
int pcl_h2d(pcl_ctx *ctx, void *dev, const void *host, int64_t bytes);
int pcl_d2h(pcl_ctx *ctx, void *host, const void *dev, int64_t bytes);
int pcl_dev_memset(pcl_ctx *ctx, void *dev, int value, int64_t bytes);

/* stream timing with HIP events (bench.py): *ms_out = device time between the two records. */
int pcl_timer_start(pcl_ctx *ctx);
int pcl_timer_stop(pcl_ctx *ctx, double *ms_out); /* synchronises */

/* Per-kernel device timing for bench.py: while enabled, every Level-2 step records a HIP event pair
 * on the context stream immediately around its kernel launch (no host synchronisation).
 * pcl_prof_enable(ctx, 1) clears earlier samples; pcl_prof_read synchronises and sums the samples of
 * one kernel.  Host pointers, any may be NULL. */
#define PCL_PROF_NEWTON      0 /* k_newton                         */
#define PCL_PROF_SCATTER     1 /* k_scatter / hipRTC specialisation */
#define PCL_PROF_DELETE_MASK 2 /* k_delete_mask                    */
#define PCL_PROF_COMPACT     3 /* k_compact                        */
#define PCL_PROF_COUNTERS    4 /* k_counters                       */
#define PCL_PROF_FUSED       5 /* k_fused / hipRTC specialisation  */
#define PCL_PROF_MULTI       6 /* k_multi / k_mixed / hipRTC specialisations */
#define PCL_PROF_ONEPASS     7 /* k_delete_onepass                 */
#define PCL_PROF_DELETE_AHEAD 8 /* k_delete_ahead / k_delete_ahead_live: delete loop bodies worked out a launch at a time */
int pcl_prof_enable(pcl_ctx *ctx, int on);
int pcl_prof_read(pcl_ctx *ctx, int kernel_id, int64_t *launches_out, double *total_ms_out,
                  double *min_ms_out, double *max_ms_out);

/* ---------------------------------------------------------------- Level 1: reference-ABI kernels
 * Argument ORDER and meaning follow the OpenCL kernels exactly; N is the global work size.        */

/* kernel light_scatter_step_del(dx, dy, dz, rand, n, A, result)          physicl/light.py:146-158 */
int pcl_k_light_scatter_step_del(pcl_ctx *ctx, const double *dx, const double *dy, const double *dz,
                                 const double *rand, double n, double A, int32_t *result, int64_t N);

/* kernel test(d0, d1, d2, rand, A, n, res) -- ScatterDeleteStep's CLProgram
 * body physicl/light.py:239-249, signature generated by physicl/__init__.py:583-597             */
int pcl_k_scatter_delete_test(pcl_ctx *ctx, const double *d0, const double *d1, const double *d2,
                              const double *rand, double A, double n, int32_t *res, int64_t N);

/* kernel light_scatter_step_sphere(d0,d1,d2, rtheta,rphi,rand, A,n, [E], [r0,r1,r2], res0,res1,res2)
 * physicl/light.py:299-315.  ``E`` is read iff flags & PCL_SCATTER_WAVELENGTH; ``r0..r2`` iff
 * flags & PCL_SCATTER_VARIABLE_N, in which case ``n_expr`` is the OpenCL-C expression the reference
 * splices into the source (light.py:299) and kernel argument ``n`` is unused, as in the reference.
 * ``c`` and ``h`` are the values of the literals str(c), str(h).upper() pasted into the source
 * (light.py:301, 309-311).  Miss: res0[i] = NaN, res1[i]/res2[i] NOT written (light.py:313).      */
int pcl_k_light_scatter_step_sphere(pcl_ctx *ctx, const double *d0, const double *d1, const double *d2,
                                    const double *rtheta, const double *rphi, const double *rand,
                                    double A, double n, const double *E, const double *r0,
                                    const double *r1, const double *r2, double *res0, double *res1,
                                    double *res2, int64_t N, int flags, double c, double h,
                                    const char *n_expr /* host string, may be NULL */);

/* Stable compaction indices of a flag array: idx_out[0..*n_keep) = ascending i with flags[i] == 0
 * -- the survivors of ``for idx, x in enumerate(out["res"]): if x == 1: sim.remove_obj(...)``
 * (physicl/light.py:258-260).  idx_out needs room for N entries; n_keep_out is a host pointer.    */
int pcl_k_compact_indices(pcl_ctx *ctx, const int32_t *flags, int64_t N, int64_t *idx_out,
                          int64_t *n_keep_out);

/* Validate a variable_n_fn expression without compiling it (host strings).  Accepted grammar:
 * numbers, + - * / ( ) , the functions exp sqrt pow log log2 log10 exp2 sin cos tanh fabs fmin fmax,
 * and the array reads r0[gid] r1[gid] r2[gid] d0[gid] d1[gid] d2[gid] E[gid].                     */
int pcl_expr_validate(const char *n_expr);
/* (The hipRTC code objects of an expression are cached in $PCL_RTC_CACHE -- a directory, "off" disables -- default
 * ~/.cache/physicl_amd/rtc, checksummed; a second process loads them instead of compiling.) */

/* User kernels -- the role of CLProgram.build_kernel()/run() (physicl/__init__.py:583-597, 648-656): the
 * caller supplies the parameter list it generated from its CLInput/CLOutput metadata (e.g.
 * "double *d0, double A, int *res") and the kernel BODY in the OpenCL-C dialect the reference's kernels use
 * (get_global_id(0), __global, NAN, pow/sqrt/sin/cos/exp ...).  The body is compiled with hipRTC (unfused
 * arithmetic) into a kernel that runs once per work-item 0..n-1.  argbuf = the arguments packed as a C struct
 * in declaration order (pointers and doubles 8-byte aligned, ints 4).  Unlike variable_n_fn expressions a body
 * is arbitrary code and is NOT validated: like any OpenCL kernel it can read out of bounds. */
int pcl_user_kernel_build(pcl_ctx *ctx, const char *name, const char *params, const char *body, void **kernel_out);
int pcl_user_kernel_launch(pcl_ctx *ctx, void *kernel, int64_t n, const void *argbuf, int64_t argbuf_bytes);
int pcl_user_kernel_free(pcl_ctx *ctx, void *kernel);

/* ---------------------------------------------------------------- Level 2: resident particle store */

/* Allocate storage for up to ``capacity`` particles; count is set to 0.  pcl_store_alloc = fp64.
 * The 13 fields (plus 4 internal rows: the velocity double buffer and the wavelength-term cache) live in ONE
 * allocation, tiled:  element i of a field is at  row0 + (i / T) * tile_stride + (i % T)  elements, T =
 * tile_len = 2048 particles, tile_stride = 17 * T -- so the ~13 streams a step reads are interleaved at 16 KiB
 * (fp64) granularity inside one slab and every HBM channel sees the same mix, wherever the driver placed the
 * slab.  upload/download translate to and from dense host arrays; pcl_store_layout reports T and the stride
 * for callers that read the rows directly.  ids, kinds, the compaction double buffer and the random-input
 * arrays are dense and allocated on first use.  Host buffers passed to upload/download/upload_rand hold
 * elements of the store's dtype (double or float). */
int pcl_store_alloc(pcl_ctx *ctx, int64_t capacity);
int pcl_store_alloc_dtype(pcl_ctx *ctx, int64_t capacity, int dtype);
int pcl_store_dtype(pcl_ctx *ctx, int *dtype_out);
int pcl_store_free(pcl_ctx *ctx);
int pcl_store_capacity(pcl_ctx *ctx, int64_t *capacity_out);
int pcl_store_count(pcl_ctx *ctx, int64_t *count_out);         /* host mirror, no sync */
/* The one-launch-per-body delete path (pcl_step_fused_delete with PCL_FUSED_LAZY, all photons, PCL_RNG_PHILOX) keeps
 * removed photons' slots until fewer than half of the slots are alive: a bit per slot says who is there, a loop body is
 * one kernel that moves nothing, and the stable compaction the reference's removal loop amounts to
 * (physicl/light.py:258-260) is run for several bodies at once.  *slots_out = slots the store currently spans (== the
 * count when it is dense); *pending_moves_out (may be NULL) = Newton moves r has not been given yet.  Every other entry
 * point sees the dense store: it is compacted first, survivors in order, r up to date.                               */
int pcl_store_slots(pcl_ctx *ctx, int64_t *slots_out, int *pending_moves_out);
/* Delete loop bodies ahead of their calls.  When a pcl_step_fused_delete call repeats the previous one with ``step``
 * advanced by one -- a run's loop, physicl/__init__.py:512-516 --, or is the first delete body of a population (taken for
 * the start of such a loop), the library works out that body AND the next ones in one
 * launch that leaves the store untouched (PCL_AHEAD_K = 24 bodies for stores of up to PCL_AHEAD_MAX_SLOTS = 2^22 slots,
 * PCL_AHEAD_K_BIG = 16, 12 beyond 2^25 slots, above: one sweep of the extent serves them all), and answers the following calls, if they are the
 * predicted ones, from those rows without a launch (a loop body of a small store is a 20 us round trip to the host, not
 * bytes; a big store's body is a sweep of its extent).  Any other call first makes the state after the bodies handed out so
 * far real (one kernel; a big store whose alive photons have fallen below the compaction threshold is compacted from those
 * masks), so nothing but timing ever shows; a caller whose loop keeps looking at the store between bodies makes the library
 * pause (exponentially) before it tries again.  Statistics since the context was created (host pointers, any may be NULL):
 * launches of the K-body kernel, bodies answered (the launching one included), launches whose rows were not all used.  */
int pcl_store_ahead_stats(pcl_ctx *ctx, int64_t *launches_out, int64_t *served_out, int64_t *missed_out);
/* What k_delete_ahead_live did in its launches since the context was created, as the kernel itself tallied it (host
 * pointers, any may be NULL): groups of 128 slots loaded (groups without an alive photon are skipped; the photons' first
 * Philox block is decided where they are loaded) whose first pass decided two bodies / one body, rounds of 64 listed
 * photons that decided two bodies, rounds that decided one.  bench.py prices them with the kernel's instruction counts
 * (profiles/isa_counts.json, "k_delete_ahead_live<double>") for the VALU roofline of the delete legs. */
int pcl_store_ahead_work(pcl_ctx *ctx, int64_t *groups_two_out, int64_t *groups_one_out, int64_t *rounds_two_out, int64_t *rounds_one_out);
/* The clock (GHz) the chip held under the k_delete_ahead_live launches of this context (as pcl_store_last_multi_clock,
 * over all of them); 0 before any launch. */
int pcl_store_ahead_clock(pcl_ctx *ctx, double *ghz_out);
/* Allocate now what the first compaction of the store would allocate on demand (the second slab -- chosen among a few
 * candidates like the first, tens of ms for a big store --, the id arrays, the mask scratch), so that a run whose step
 * list holds a delete step pays for it at set-up and not inside its third loop body.  Optional.                        */
int pcl_store_reserve_compaction(pcl_ctx *ctx);

/* Set the particle count (<= capacity) and declare a new population: ids = id_base + index (no id array is read) and
 * every particle a photon (a kind array of an earlier upload is dropped) until pcl_store_upload_ids / _kind. */
int pcl_store_set_count(pcl_ctx *ctx, int64_t count, int64_t id_base);

int pcl_store_upload(pcl_ctx *ctx, int field, const void *host, int64_t offset, int64_t n);
int pcl_store_download(pcl_ctx *ctx, int field, void *host, int64_t offset, int64_t n);
int pcl_store_upload_ids(pcl_ctx *ctx, const int64_t *host, int64_t offset, int64_t n);
int pcl_store_download_ids(pcl_ctx *ctx, int64_t *host, int64_t offset, int64_t n);
int pcl_store_upload_kind(pcl_ctx *ctx, const uint8_t *host, int64_t offset, int64_t n);
int pcl_store_download_kind(pcl_ctx *ctx, uint8_t *host, int64_t offset, int64_t n);
/* device pointer of element 0 of a field's CURRENT row (changes after a compaction and after a lazy fused
 * step); element i is at row0[(i / tile_len) * tile_stride + i % tile_len], see pcl_store_layout */
int pcl_store_field_ptr(pcl_ctx *ctx, int field, void **dev_out);
int pcl_store_layout(pcl_ctx *ctx, int64_t *tile_len_out, int64_t *tile_stride_out);
/* How the store's slab was chosen (stores of >= 512 MB; INTEGRATION.md, "Device memory"): the sweep rates in GB/s of the
 * candidate blocks in the order they were tried (up to ``cap`` <= 8 values), their number (0: no selection took place) and
 * the rate of the block that was kept.  Any pointer may be NULL.                                                        */
int pcl_store_alloc_info(pcl_ctx *ctx, int *n_candidates_out, double *rates_gbps_out, int cap, double *chosen_gbps_out);

/* Random inputs for PCL_RNG_INPUT: which = 0 rtheta, 1 rphi, 2 rand; n values for particles
 * [0, n) in store order (entries of non-photon particles are ignored). */
int pcl_store_upload_rand(pcl_ctx *ctx, int which, const void *host, int64_t n);
/* The three inputs of an isotropic scatter step at once, from the uniforms as the reference draws them: per photon three
 * np.random.random() in the order rtheta, rphi, rand (physicl/__init__.py:606-619), i.e. the (n, 3) row-major array of
 * np.random.random((n, 3)).  u3_host holds the rows of particles [offset, offset + n), n <= 2**20 per call; calls follow
 * one another in particle order starting at offset 0 (the same stream as one big draw).  The scaling of light.py:285,
 * rtheta = (u * 2) * pi and rphi = u * pi, is done on the device with those very operations.  The call returns when the
 * chunk has been staged (two pinned slots): the caller draws the next chunk while this one is copied.               */
int pcl_store_upload_rand3(pcl_ctx *ctx, const double *u3_host, int64_t offset, int64_t n);

/* Bulk creation on the device of ``n`` photons at r = 0 with v = (c, 0, 0), dr = dv = 0 and
 * E = e_min + (e_max - e_min) * U^(1/3) -- the SoA equivalent of light.generate_photons with its
 * default sampler np.random.power(3) (physicl/light.py:112-128).  U is Philox-keyed by
 * (seed, id_base + index).  Sets count = n, ids = id_base + index, every kind = photon. */
int pcl_store_fill_photons(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, double e_min,
                           double e_max, uint64_t seed);

/* The same for a tabulated energy distribution: cdf_host[nbins] (non-decreasing, last = 1) and grid_host[nbins];
 * photon energy = grid[x] for the first x with cdf[x] >= U -- the binned Planck sampler
 * planck_phot_distribution (physicl/light.py:73-104) for all n photons at once (host pointers). */
int pcl_store_fill_photons_table(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, const double *cdf_host,
                                 const double *grid_host, int nbins, uint64_t seed);

/* NewtonianKinematicsStep.run (physicl/newton.py:10-16): dr = v*dt (rounded, stored); r = r + dr.
 * Applies to every particle of every kind. */
int pcl_step_newton(pcl_ctx *ctx, double dt);

/* ScatterIsotropicStep.__run_cl (physicl/light.py:281-331) fused: hit test, new direction AND the
 * host write-back (hit: v = v', dv = v' - v_old; miss: dv = 0).  ``A``/``n`` are the KERNEL
 * constants, i.e. after the reference's swap (light.py:287); the Python layer applies the swap.
 * hits_out (host, may be NULL): number of photons scattered in this call; when non-NULL the call
 * synchronises. */
int pcl_step_scatter_isotropic(pcl_ctx *ctx, double A, double n, int flags, double c, double h,
                               const char *n_expr, int rng_mode, uint64_t seed, uint32_t step,
                               int64_t *hits_out);

/* The reference's CPU paths of the light steps (``cl_on=False``: ScatterIsotropicStep.__run_py light.py:335-350,
 * ScatterDeleteStepReference.__run_py light.py:216-223) consume np.random in a DATA-DEPENDENT order -- a photon that is
 * hit draws two more numbers, a removal makes the list iteration skip the next object -- so the host has to see every
 * photon's collision probability before it can hand the device its randoms / removal flags.  pcl_step_scatter_pcoll
 * writes pcoll = A * n * |dr| [* pow((h*c)/E, -4)] of every particle (store dtype, constant n only: the CPU path has
 * no variable_n) to a host array; pcl_step_scatter_isotropic(PCL_RNG_INPUT, flags | PCL_SCATTER_PY_DV) then applies
 * the scatter with the host's draws, and pcl_step_delete_flags removes the particles whose flag is 1 (stable). */
int pcl_step_scatter_pcoll(pcl_ctx *ctx, double A, double n, int flags, double c, double h, void *pcoll_out_host);
int pcl_step_delete_flags(pcl_ctx *ctx, const int32_t *flags_host, int64_t *n_alive_out, int64_t *n_removed_out);

/* The whole Simulation loop body in ONE kernel: NewtonianKinematicsStep, then (do_scatter != 0)
 * ScatterIsotropicStep, then (n_planes >= 0) the counters of ScatterSignMeasureStep /
 * ScatterMeasureStep on the post-step state -- the step order every example and test uses
 * (test/test_light.py:32-36, physicl/__init__.py:512-516).  Per-particle arithmetic and order are
 * those of pcl_step_newton + pcl_step_scatter_isotropic + pcl_step_counters, so results are
 * bit-identical to calling the three; r, v, E are read once and dr is consumed from registers.
 * Scatter parameters as pcl_step_scatter_isotropic.  n_planes = -1 switches the counters off.
 * flags | PCL_FUSED_LAZY: dr and dv are NOT written by the step.  Both are pure functions of data the
 * store keeps anyway -- dr = v_before * dt (newton.py:15) and dv = v_after - v_before, which is the
 * reference's v' - v_old on a hit and exactly +0 on a miss (light.py:329-331) -- so the step writes
 * the new velocities into the other half of a v double buffer (whole lines, no read-modify-write)
 * and every later call that touches the store (download, field_ptr, any other step) first runs one
 * materialise pass that produces bit-identical dr/dv arrays.  A chain of lazy steps never pays for
 * the intermediate dr/dv that nothing reads: 104 B per particle-step instead of 128 + 24h.
 * out_host (may be NULL = no synchronisation): int64[5 + n_planes] =
 *   { N, xp, yp, zp, plane counts..., hits }. */
int pcl_step_fused(pcl_ctx *ctx, double dt, int do_scatter, double A, double n, int flags, double c,
                   double h, const char *n_expr, int rng_mode, uint64_t seed, uint32_t step,
                   const double *planes_host, int n_planes, int64_t *out_host);

/* 1 if every particle is a photon and ids are implicit (id_base + index: nothing has been compacted or uploaded
 * with explicit ids/kinds) -- the precondition of pcl_step_fused_multi.  (Any other store takes pcl_step_mixed_multi;
 * the fast single-step kernel has a variant that reads the explicit ids and the kind bytes.) */
int pcl_store_is_uniform(pcl_ctx *ctx, int *uniform_out);

/* k_steps consecutive fused steps (Newton + ScatterIsotropic + sign counters) in ONE pass over the store.  Photons
 * do not interact, so each one is loaded once, advanced k_steps times in registers -- the very operations of
 * k_steps calls of pcl_step_fused(dt, 1, ..., PCL_FUSED_LAZY, PCL_RNG_PHILOX, seed, step0 + k) in the same order --
 * and stored once: results and per-step counters are bit-identical to the step-by-step sequence while the HBM
 * traffic per particle-step falls from 104 B to 128 / k_steps B (fp64).  It is the device side of a run whose
 * passes are [UpdateTimeStep][NewtonStep][ScatterIsotropicStep][counting measures] with nothing on the host
 * looking at the photons in between (physicl/__init__.py:441-448 runs the same steps every pass).
 * Requirements: all-photon store with implicit ids (no compaction yet), device RNG; dr/dv are left implicit
 * exactly as after the last single lazy step.  flags: PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N.
 * planes_host / n_planes (0..12) as in pcl_step_fused.  out_host (may be NULL): int64[k_steps][5 + n_planes] =
 * { N, xp, yp, zp, plane counts..., hits } per step; 1 <= k_steps <= 64. */
int pcl_step_fused_multi(pcl_ctx *ctx, double dt, int k_steps, double A, double n, int flags, double c, double h,
                         const char *n_expr, uint64_t seed, uint32_t step0, const double *planes_host, int n_planes,
                         int64_t *out_host);
/* The work the last pcl_step_fused_multi launch did, as its kernel tallied it (host pointers, any may be NULL): the
 * K-step pass is bound by VALU issue, not by HBM, and its instruction count per wave is
 *   (instructions of a step's decision part) x wave-steps + (instructions of a dense pass) x dense passes,
 * the two static counts being properties of the code object (profiles/isa_counts.json).  *dense_passes_out = passes of
 * the waves' hit queues (ceil(hits of the wave in that step / 64) summed over waves and steps), *wave_steps_out = waves x
 * K, *photons_per_wave_out = 128 or 256 (fp64; the form the launch took), *saturated_wave_steps_out = wave-steps whose
 * variable_n_fn values came from exp's saturation shortcut instead of its polynomial (-1: the launch ran the variant
 * without the probe; the library picks per launch, PCL_MULTI_SAT = 1 / 0 forces).                                    */
int pcl_store_last_multi_work(pcl_ctx *ctx, int64_t *dense_passes_out, int64_t *wave_steps_out, int *photons_per_wave_out,
                              int64_t *saturated_wave_steps_out);
/* The clock (GHz) the chip held under the last pcl_step_fused_multi launch: shader cycles (s_memtime) over 100 MHz ticks
 * (s_memrealtime) between the start and the end of every workgroup, summed over the launch.  The ceiling of a kernel bound
 * by VALU issue is 1024 SIMDs x THIS clock, in SIMD-cycles per second (bench.py's roofline record); 0 before any launch.
 * No counterpart in the reference (instrumentation of this library).                                                    */
int pcl_store_last_multi_clock(pcl_ctx *ctx, double *ghz_out);
/* Rows of 64 particles a wave kept per trip in the last pcl_step_mixed_multi launch: 2 (pcl_mixed_body, every variable-n
 * form) or 3 (pcl_mixed_body_lds: constant n while a photon's hit probability A n c dt is below 0.33; PCL_MIXED_NE3 = 1 / 0
 * forces); 0 before any.  No counterpart in the reference (instrumentation of this library).                            */
int pcl_store_last_mixed_rows(pcl_ctx *ctx, int *rows_out);
/* Debug builds of the K-step kernels only (environment PCL_RTC_EXTRA=PCL_HIT_HIST, knob PCL_MULTI_HIST=1): the last
 * launch's histogram of hits queued per wave and step (per round in the 256-photon form), bins 0 .. 127 and ">= 128"
 * (host pointer, 129 elements).  tools/hit_hist.py; PCL_ERR_STATE otherwise.                                          */
int pcl_store_last_multi_hist(pcl_ctx *ctx, int64_t *hist129_out);

/* Counters of the OLDEST not-yet-read pcl_step_fused that was called with out_host == NULL and counters on
 * (same layout, same n_planes).  Up to two such steps may be outstanding: enqueue step k+1, then read step k --
 * the call waits for step k only (an event, not the stream), so the GPU runs step k+1 while the host handles
 * step k's counters (exit condition, all-reduce, measure rows). */
int pcl_step_fused_read(pcl_ctx *ctx, int n_planes, int64_t *out_host);

/* Hit count of the most recent pcl_step_scatter_isotropic / pcl_step_fused (host pointer).  Free of extra
 * synchronisation when a pcl_step_counters call has completed since that step. */
int pcl_store_last_scatter_hits(pcl_ctx *ctx, int64_t *hits_out);

/* ScatterDeleteStep.run (physicl/light.py:231-260) fused: flag kernel + stable compaction of every
 * state array.  Non-photon particles are never removed.  Outputs are host pointers (may be NULL);
 * the call synchronises (the new count is needed by exit conditions such as
 * ``len(sim.objects) == 0``, physicl/__init__.py:414). */
int pcl_step_scatter_delete(pcl_ctx *ctx, double A, double n, int rng_mode, uint64_t seed,
                            uint32_t step, int64_t *n_alive_out, int64_t *n_removed_out);
/* The loop body of a delete simulation as one pipeline: NewtonianKinematicsStep, then ScatterDeleteStep,
 * then (n_planes >= 0) the measure counters on the survivors -- the step order of test/test_light.py:52-59.
 * Pass 1 moves the particles and computes the delete flags in one sweep over r and v; pass 3 (the stable
 * compaction) counts while the survivors go through its registers.  Bit-identical to pcl_step_newton +
 * pcl_step_scatter_delete + pcl_step_counters.  flags: 0 or PCL_FUSED_LAZY (dr is neither written nor moved:
 * it stays v*dt until something reads it).  out_host (may be NULL): int64[5 + n_planes] =
 * { N alive, xp, yp, zp, plane counts..., removed }.  Synchronises (the new count is needed on the host). */
int pcl_step_fused_delete(pcl_ctx *ctx, double dt, double A, double n, int flags, int rng_mode, uint64_t seed,
                          uint32_t step, const double *planes_host, int n_planes, int64_t *out_host);

/* k_steps consecutive delete loop bodies (Newton + ScatterDelete + the counters of the measure steps) in one pass
 * and ONE compaction: a photon of a delete run never changes its velocity, so per step only r += v*dt, the
 * decision draw and the compare remain until it is removed.  State and per-step rows are identical to k_steps
 * calls of pcl_step_fused_delete(dt, A, n, PCL_FUSED_LAZY, PCL_RNG_PHILOX, seed, step0 + k, ...); dr is left
 * implicit.  Device RNG only.  out_host (may be NULL): int64[k_steps][5 + n_planes] =
 * { N alive after the step, xp, yp, zp, plane counts..., removed in the step }; n_planes = -1: alive/removed only. */
int pcl_step_fused_delete_multi(pcl_ctx *ctx, double dt, int k_steps, double A, double n, uint64_t seed, uint32_t step0,
                                const double *planes_host, int n_planes, int64_t *out_host);

/* k_passes whole passes of a loop whose body holds one isotropic-scatter phase and/or one delete phase, each phase =
 * NewtonianKinematicsStep + the light step + the counters of the measure steps that follow it:
 *   {ISOTROPIC}          [Newton, ScatterIsotropic]                      on ANY store (explicit ids, plain Objects)
 *   {DELETE}             [Newton, ScatterDelete]
 *   {ISOTROPIC, DELETE}  [Newton, ScatterIsotropic, Newton, ScatterDelete]   (BASELINE.json configs[4]; or the reverse)
 * in ONE pass over the store and, with a delete phase, ONE stable compaction afterwards.  Phase j of pass p uses launch
 * index step0 + p * n_phases + j, so state, survivor order and rows are identical to the same passes run one launch at
 * a time (pcl_step_fused / pcl_step_fused_delete with PCL_FUSED_LAZY, PCL_RNG_PHILOX).  A photon removed by a delete
 * phase takes no further part (physicl/__init__.py:455-459).  dr and dv are left implicit.  A, n, flags, c, h, n_expr:
 * the isotropic phase, as pcl_step_fused_multi; A_del, n_del: the delete phase.  Device RNG only.
 * out_host (may be NULL): int64[k_passes * n_phases][5 + n_planes] = { N alive after the phase, xp, yp, zp, plane
 * counts..., hits (isotropic phase) | removed (delete phase) }; k_passes * n_phases <= 64. */
#define PCL_PHASE_ISOTROPIC 0
#define PCL_PHASE_DELETE    1
int pcl_step_mixed_multi(pcl_ctx *ctx, double dt, int k_passes, int n_phases, const int *phase_kinds_host, double A, double n,
                         int flags, double c, double h, const char *n_expr, double A_del, double n_del, uint64_t seed,
                         uint32_t step0, const double *planes_host, int n_planes, int64_t *out_host);

/* TracePathMeasureStep.run for a tracked subset (physicl/light.py:447-458), worked out AHEAD of the K-pass launch that
 * will move the store: the positions the particles with the ids ``ids_host`` (strictly ascending, n_ids <= PCL_TRACE_MAX)
 * will have when the trace step runs in each of the next k_passes passes of a loop whose body holds the phases
 * ``phase_kinds_host`` -- the same description of the loop, with the same constants, seed and step0, that is about to be
 * handed to pcl_step_fused_multi (n_phases = 1, ISOTROPIC), pcl_step_fused_delete_multi (n_phases = 1, DELETE) or
 * pcl_step_mixed_multi.  ``record_phase``: the trace step sits behind that phase's light step (and its counting measures)
 * in the pass.  A photon's history is a pure function of its own state and of (seed, launch index, id) -- photons do not
 * interact, the device random stream is keyed by the id -- so one thread per TRACKED photon runs the per-photon
 * operations of the K-step kernels on the store as it stands and writes one row per pass; the store is NOT changed, and the
 * K-step kernels (bound by VALU issue) carry nothing for the trace.  Bit-identical to downloading r after every pass.
 * Works on any store (explicit ids after compactions, plain Objects, behind an alive mask, either dtype).
 * out_host: double[k_passes][n_ids][4] = { r0, r1, r2, moved } -- moved = 1 if the particle's dv is not the zero vector
 * at that point (trace_dv, light.py:456) else 0; four NaNs where the particle is not in the store at that point (removed
 * by a delete phase, or never there: the reference traces 'nan;nan;nan', light.py:435).  Device RNG only.  Synchronises --
 * unless out_host is NULL: the kernel is then only enqueued (it writes its rows into pinned host memory of the context) and the
 * K-pass launch can follow at once on the same in-order stream; pcl_store_trace_read(ctx, out_host, k_passes * n_ids * 4) hands
 * the rows out afterwards -- behind a launch that has returned its counter rows it does not wait at all.  The context holds
 * ONE set of rows: a second pcl_store_trace_ahead before the read waits for nothing and replaces them (several tracked sets
 * per launch: pass out_host, as physicl_amd/core.py does for more than one TracePathMeasureStep).                            */
#define PCL_TRACE_MAX 65536
int pcl_store_trace_ahead(pcl_ctx *ctx, const int64_t *ids_host, int n_ids, double dt, int k_passes, int n_phases,
                          const int *phase_kinds_host, int record_phase, double A, double n, int flags, double c, double h,
                          const char *n_expr, double A_del, double n_del, uint64_t seed, uint32_t step0, double *out_host);
int pcl_store_trace_read(pcl_ctx *ctx, double *out_host, int64_t n_doubles);

/* The int32 flag array of the most recent pcl_step_scatter_delete / pcl_step_fused_delete, in PRE-compaction order
 * (what the reference's kernel returns in ``res``).  flags_host needs room for the pre-delete count. */
int pcl_store_last_delete_flags(pcl_ctx *ctx, int32_t *flags_host, int64_t n);

/* ScatterMeasureStep(measure_E=True) (physicl/light.py:383-399): the energies of the photons whose last move
 * crossed the plane (one 3-vector, NaN in the coordinates that do not define it), in particle order.  *n_out =
 * number of crossing photons; the first min(n, cap) energies (store dtype) are copied to E_out_host (host pointers). */
int pcl_step_plane_energies(pcl_ctx *ctx, const double *plane_host, void *E_out_host, int64_t cap, int64_t *n_out);

/* ScatterSignMeasureStep.run + the counting part of ScatterMeasureStep.run
 * (physicl/light.py:414-431, 374-400).  planes_host: n_planes x 3 doubles, NaN = coordinate not
 * defining the plane.  out_host: int64[PCL_CNT_PLANE0 + n_planes].  Synchronises. */
int pcl_step_counters(pcl_ctx *ctx, const double *planes_host, int n_planes, int64_t *out_host);

/* ---- Device groups: several GPUs from ONE process ---------------------------------------------------------------
 * The reference is a single process with one simulation thread (physicl/__init__.py:400-432, 501-524); this is how
 * a host written against this ABI uses a node's GPUs the same way, without one process per GPU.  A group owns one
 * context (device, stream, store) and one worker thread per entry of device_ids (an id may repeat: two contexts on one
 * GPU).  Particles are sharded by GLOBAL index in contiguous blocks -- shard g of G owns [g*N/G, (g+1)*N/G) -- and ids
 * are global, so the id-keyed device RNG gives exactly the rows and the per-photon histories of a one-device run.  A
 * group step hands the call to every shard's worker, waits for all of them, and returns the SUM of the int64 counter
 * rows (same layout as the per-context call): the rows are in host memory when a launch returns, so this sum is the
 * group's collective (the counters are the only global quantities of the path).  Windows of the
 * particle order are read shard after shard (contiguous blocks + stable compaction = global particle order).
 * pcl_group_ctx gives the i-th context for anything per shard (uploads, pcl_store_*).  One group call at a time.
 * A tracked subset (pcl_store_trace_ahead) over a group: hand every shard's context the whole id list before the group's
 * K-pass launch -- a shard answers NaN rows for the ids it does not hold, a particle lives in exactly one shard, so the row
 * that is not NaN is the particle's (what physicl_amd.multidev.MultiDevice.trace_ahead does).
 * Errors: the first failing shard's code; pcl_last_error() names the shard.                                      */
typedef struct pcl_group pcl_group;
int pcl_group_create(int n_dev, const int *device_ids, pcl_group **group_out);
int pcl_group_destroy(pcl_group *group);
int pcl_group_size(pcl_group *group, int *n_out);
int pcl_group_ctx(pcl_group *group, int i, pcl_ctx **ctx_out);
int pcl_group_shard(pcl_group *group, int64_t n_global, int i, int64_t *lo_out, int64_t *hi_out);
int pcl_group_store_alloc(pcl_group *group, int64_t capacity_global, int dtype);
int pcl_group_store_dtype(pcl_group *group, int *dtype_out);      /* PCL_DTYPE_F64 / PCL_DTYPE_F32 of the shards' stores */
int pcl_group_fill_photons(pcl_group *group, int64_t n_global, int64_t id_base, double c, double e_min, double e_max,
                           uint64_t seed);
int pcl_group_count(pcl_group *group, int64_t *count_out);
int pcl_group_sync(pcl_group *group);
int pcl_group_reserve_compaction(pcl_group *group);
/* the steps: arguments and out_host layout of pcl_step_fused (synchronous form: out_host and n_planes >= 0 required),
 * pcl_step_fused_delete, pcl_step_fused_multi, pcl_step_fused_delete_multi, pcl_step_mixed_multi; counters summed     */
int pcl_group_step_fused(pcl_group *group, double dt, int do_scatter, double A, double n, int flags, double c, double h,
                         const char *n_expr, int rng_mode, uint64_t seed, uint32_t step, const double *planes_host,
                         int n_planes, int64_t *out_host);
int pcl_group_step_fused_delete(pcl_group *group, double dt, double A, double n, int flags, int rng_mode, uint64_t seed,
                                uint32_t step, const double *planes_host, int n_planes, int64_t *out_host);
int pcl_group_step_fused_multi(pcl_group *group, double dt, int k_steps, double A, double n, int flags, double c, double h,
                               const char *n_expr, uint64_t seed, uint32_t step0, const double *planes_host, int n_planes,
                               int64_t *out_host);
int pcl_group_step_fused_delete_multi(pcl_group *group, double dt, int k_steps, double A, double n, uint64_t seed,
                                      uint32_t step0, const double *planes_host, int n_planes, int64_t *out_host);
int pcl_group_step_mixed_multi(pcl_group *group, double dt, int k_passes, int n_phases, const int *phase_kinds_host, double A,
                               double n, int flags, double c, double h, const char *n_expr, double A_del, double n_del,
                               uint64_t seed, uint32_t step0, const double *planes_host, int n_planes, int64_t *out_host);
/* elements [offset, offset + n) of a field / of the ids in GLOBAL particle order (host pointers, store dtype / int64) */
int pcl_group_download(pcl_group *group, int field, void *host, int64_t offset, int64_t n);
int pcl_group_download_ids(pcl_group *group, int64_t *host, int64_t offset, int64_t n);

/* ---------------------------------------------------------------- the counters' collective (one process per GPU)
 * The path shards by global index with no data-path exchange; the only global quantities are the int64
 * counter rows a step returns (alive, hits / removed, sign counts, plane crossings) -- they are in host memory when the
 * call returns.  A host that runs ONE PROCESS PER GPU sums them over its ranks with RCCL (over xGMI inside a node):
 *   rank 0:      pcl_comm_unique_id(id)      and ships the PCL_COMM_ID_BYTES bytes to the other ranks by whatever it has
 *   every rank:  pcl_comm_create(ctx, id, rank, world, &comm)   -- collective: returns when all ranks have arrived and a
 *                                                                   one-element all-reduce has seen every one of them
 *   per launch:  pcl_comm_allreduce_sum_i64(comm, rows, n)       -- in place, host pointer, on the context's stream
 * librccl is loaded at run time (PCL_RCCL_LIB overrides the search).  There is no fallback: a missing library or a
 * failed bring-up is PCL_ERR_STATE / PCL_ERR_HIP, and the caller must treat it as fatal (physicl_amd/dist.py does).
 * One process on several GPUs needs none of this: pcl_group_* sums on the host.                                       */
#define PCL_COMM_ID_BYTES 128
typedef struct pcl_comm pcl_comm;
int pcl_comm_unique_id(void *id_out_host);                                                      /* PCL_COMM_ID_BYTES bytes */
int pcl_comm_create(pcl_ctx *ctx, const void *id_host, int rank, int world, pcl_comm **comm_out);
int pcl_comm_allreduce_sum_i64(pcl_comm *comm, int64_t *inout_host, int n);                     /* n <= 2048 */
int pcl_comm_info(pcl_comm *comm, int *rank_out, int *world_out, int *rccl_version_out, int64_t *reduces_out);
int pcl_comm_destroy(pcl_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* PHYSICL_HIP_H */
