// physicl_hip.hip -- libphysicl_hip.so: hand-written gfx950 (CDNA4, wave64) kernels + the C ABI of
// include/physicl_hip.h.  Build: see __graft_entry__.build() (hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off).  No PyOpenCL, no CUDA compatibility layer, no Triton.
//
// Kernel inventory (HBM-bandwidth-bound streaming kernels, except the K-step passes, which trade HBM bytes for
// arithmetic and are VALU-bound; roofline and bytes in DESIGN.md).  The resident store is one tiled slab
// ([tile][17 rows][2048 particles], pcl_tix / pcl_tq).
// <T> = double (the reference's precision) or float (precision sweep, BASELINE.json configs[4]):
//   k_newton<T>           NewtonianKinematicsStep.run                 physicl/newton.py:10-16
//   k_delete_flags        light_scatter_step_del / "test"             physicl/light.py:146-158, 239-249
//   k_sphere<E>           light_scatter_step_sphere                   physicl/light.py:303-315
//   k_scatter<T,E>        ScatterIsotropicStep.__run_cl as one step   physicl/light.py:281-331
//   k_fused<T,E>          Newton + scatter + counters in one pass     (pcl_device.h)
//   k_fast<T,E>           the same, fast path                         (pcl_device.h)
//   k_multi<T,E>          K of those loop bodies per pass over the store (pcl_device.h)
//   k_lam4<T>             cache of pow((h*c)/E, -4)                   physicl/light.py:301
//   k_materialize<T>      dr, dv after lazy fused steps               physicl/newton.py:15, light.py:329-331
//   k_delete_mask<T>      delete-flag kernel -> wave64 ballot masks   physicl/light.py:239-249
//   k_newton_mask<T>      Newton + delete flag -> masks (fused delete pipeline, pass 1; any store, any RNG mode)
//   k_flag_mask2<T>       the same flag for all-photon stores with the device RNG: two photons per lane, 16-B loads
//   k_newton_mask_multi<T> K delete loop bodies per pass -> one mask + per-step measure rows (K <= 2)
//   k_newton_mask_multi_q<T,RINGS> the same with the removed photons taken out of the lanes (LDS rings; K > 2)
//   k_mixed<T,E>          K passes of [Newton, ScatterIsotropic] and/or [Newton, ScatterDelete] on any store (pcl_device.h)
//   k_compact_count<T,W>  compaction with the measure counters folded in (pass 3, 8 B per surviving lane)
//   k_compact_lds<T,W>    the same through LDS: aligned 16-B groups (chosen on the device when > 35 % survive)
//   (opt-in experiments, measured and not adopted: k_delete_onepass, k_newton_mask_multi_p)
//   k_tile_scan           exclusive scan of per-tile survivor counts
//   k_compact<W,NF>       stable compaction of the SoA state          physicl/light.py:258-260,
//                                                                      physicl/__init__.py:455-459
//   k_counters<T>         ScatterSignMeasureStep / ScatterMeasureStep physicl/light.py:374-431
//   k_fill_photons<T>     generate_photons (bulk, on device)          physicl/light.py:112-128
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <thread>
#include <memory>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/physicl_hip.h"
#include "pcl_device.h"
#include "pcl_rtc_source.inc" // generated: static const char pcl_rtc_source[] = <text of pcl_device.h>

namespace {

// =================================================================================================
// errors
// =================================================================================================
thread_local std::string g_err = "";

int fail(int code, const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define PCL_HIP(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return fail(e__ == hipErrorOutOfMemory ? PCL_ERR_NOMEM : PCL_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                                \
    } while (0)

#define PCL_TRY(expr)                    \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != PCL_OK) return rc__; \
    } while (0)

// =================================================================================================
// knobs: the A/B switches of the library.  A knob's value is the override set with pcl_set_knob(name, value) if there
// is one, else the environment variable of that name, else unset.  The delete path's switches are read at every call
// (``knob`` objects below: a cached value re-read when any override changes), so that one process can run the same
// program under several settings (tests/conftest.py ``pcl_knobs``); the remaining getenv() switches are read once.
// =================================================================================================
std::mutex g_knob_mu;
std::map<std::string, std::string> g_knob_over; // name -> value ("" erases)
std::atomic<int> g_knob_gen{1};

struct knob {
    const char *name;
    std::atomic<int> gen{0};
    std::atomic<bool> has{false};
    std::atomic<double> num{0.0};
    std::atomic<bool> zero{false}; // the text starts with '0' (the "=0 switches off" convention)
    explicit knob(const char *n) : name(n) {}
    void refresh() {
        const int g = g_knob_gen.load(std::memory_order_acquire);
        if (gen.load(std::memory_order_acquire) == g) return;
        std::lock_guard<std::mutex> lk(g_knob_mu);
        const char *v = nullptr;
        auto it = g_knob_over.find(name);
        if (it != g_knob_over.end())
            v = it->second.c_str();
        else
            v = getenv(name);
        has.store(v != nullptr);
        num.store(v ? atof(v) : 0.0);
        zero.store(v && v[0] == '0');
        gen.store(g, std::memory_order_release);
    }
    bool set() { refresh(); return has.load(); }
    bool off() { refresh(); return has.load() && zero.load(); }       // NAME=0
    double value(double dflt) { refresh(); return has.load() ? num.load() : dflt; }
};

// =================================================================================================
// geometry
// =================================================================================================
constexpr int kBlock = 256;           // 4 wave64 per workgroup
constexpr int kTileRows = 32;         // compaction tile = 32 rows of 64 particles
constexpr int kTile = kTileRows * 64; // 2048 particles: one workgroup, 8 rows per wave
constexpr int kCounterSlots = 32;
constexpr int kAccSlots = 32;          // accumulators of k_delete_alive, behind the counter slots in d_cnt
constexpr int kMultiSlots = PCL_MULTI_MAX * (5 + PCL_MAX_PLANES); // per-step counter rows of a K-step pass
static_assert(PCL_MAXPL == PCL_MAX_PLANES, "device header and C ABI disagree on the number of measure planes");
static_assert(PCL_MULTI_MAX <= 64, "per-step tallies are kept one per lane of a wave64");
constexpr int kRows = PCL_NFIELDS + 4;       // rows per tile of the store slab: 13 fields + vprev0..2 + lam4
constexpr int kRowVprev = PCL_NFIELDS, kRowLam4 = PCL_NFIELDS + 3;
constexpr int64_t kTileT = PCL_T;            // particles per tile (rows of 2048 elements)

__host__ __device__ inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// =================================================================================================
// kernels
// =================================================================================================

// ---- NewtonianKinematicsStep.run: dr = v*dt (rounded, STORED), r = r + dr   newton.py:15-16 -----
// 96 B per fp64 particle-step (6 loads + 6 stores).  Each lane moves 16 B per access (VEC consecutive
// particles); unfused mul/add so results are bit-exact.  Arrays are padded, so the last group is whole.
template <typename T>
struct newton_args {
    const T *v[3];
    T *r[3];
    T *dr[3];
    T dt;
    int64_t N;
    int64_t ts; // tile stride (elements)
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_newton(newton_args<T> a) {
    typedef pcl_rt<T> R;
    constexpr int VEC = R::VEC;
    typedef pcl_vec<T, VEC> VV;
    const int64_t nq = (a.N + VEC - 1) / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        T v[3][VEC], r[3][VEC], d[3][VEC];
        const int64_t qt = pcl_tq<VEC>(q, a.ts);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            VV::ld(a.v[k], qt, v[k]);
            VV::ld(a.r[k], qt, r[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                d[k][e] = R::mul(v[k][e], a.dt);
                r[k][e] = R::add(r[k][e], d[k][e]);
            }
            VV::st(a.dr[k], qt, d[k]);
            VV::st(a.r[k], qt, r[k]);
        }
    }
}

// ---- materialise what lazy steps left implicit: dr = v_move*dt, dv = v - v_prev ---------------------
// v_move = the velocity the LAST Newton move used (the vprev rows if the last step was a lazy Newton + scatter pass,
// else the current v rows); v - v_prev is +0 for a photon that was not scattered: exactly the reference's dv = 0.
template <typename T>
struct materialize_args {
    const T *vmove[3], *vprev[3], *v[3];
    T *dr[3], *dv[3];
    const unsigned char *kind;
    T dt;
    int do_dr, do_dv;
    int64_t N;
    int64_t ts;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_materialize(materialize_args<T> a) {
    typedef pcl_rt<T> R;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += stride) {
        const bool photon = a.kind ? (a.kind[i] != 0) : true;
        const int64_t ti = pcl_tix(i, a.ts);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.do_dr) a.dr[k][ti] = R::mul(a.vmove[k][ti], a.dt);
            if (a.do_dv && photon) a.dv[k][ti] = R::sub(a.v[k][ti], a.vprev[k][ti]);
        }
    }
}

// ---- delete-flag kernel (Level 1): result = (A*n*norm >= rand) ? 1 : 0   light.py:146-158 -------
__global__ void __launch_bounds__(kBlock) k_delete_flags(const double *__restrict__ d0, const double *__restrict__ d1,
                                                         const double *__restrict__ d2,
                                                         const double *__restrict__ rand, double A, double n,
                                                         int32_t *__restrict__ res, int64_t N) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double An = __dmul_rn(A, n);
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < N; gid += stride) {
        const double pcoll = __dmul_rn(An, pcl_step_norm<double>(d0[gid], d1[gid], d2[gid]));
        res[gid] = (pcoll >= rand[gid]) ? 1 : 0;
    }
}

// ---- ahead-of-time variants of the kernels whose bodies live in pcl_device.h: constant n (VAR_N = false) and the three
// parametrised variable-n shapes of the examples (VAR_N = true, pcl_nprof) for machines without hipRTC ---------------
template <bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_sphere(pcl_sphere_args a) {
    pcl_sphere_body<USE_E, VAR_N>(a);
}
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_scatter(pcl_scatter_args<T> a) {
    pcl_scatter_body<T, USE_E, VAR_N>(a);
}
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_fused(pcl_fused_args<T> a) {
    pcl_fused_body<T, USE_E, VAR_N>(a);
}
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_fast(pcl_fast_args<T> a) {
    pcl_fast_body<T, USE_E, VAR_N, pcl_rt<T>::VEC>(a);
}
template <typename T> constexpr int kMultiNQ = sizeof(T) == 8 ? 2 : 1; // VEC-wide groups per lane of the K-step pass
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_multi(pcl_multi_args<T> a) {
    pcl_multi_body_lds<T, USE_E, VAR_N, pcl_rt<T>::VEC, kMultiNQ<T>>(a); // 256 photons per wave in either precision
}
// constant n, fp64, three photons per lane (192 per wave, 8-byte accesses): where a wave of 256 photons queues 64 to 85 hits a step --
// a full dense pass and a nearly empty one -- 192 queue 48 to 64, one pass (pcl_rtc_multi3_*: the same form of the hipRTC
// specialisations).  A constant-n loop knows its hit probability A n c dt before the first launch (step_multi_t picks).
// 31 KB of LDS: five workgroups per CU; five waves per SIMD (91 VGPRs) without the wavelength term, four with it (its lam4 values
// are the registers that would spill under the bound of five).
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) k_multi3_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, 0, 1, 3>(a);
}
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) k_multi3_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, 0, 1, 3>(a);
}
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_fastg(pcl_fast_args<T> a) { // explicit ids and/or plain Objects
    pcl_fast_body<T, USE_E, VAR_N, pcl_rt<T>::VEC, true>(a);
}
// (constant n, fp64: 133 VGPRs left three waves per SIMD, and the K-pass loop is arithmetic -- with four, what 128 registers
// allow without a spill, configs[4]'s 100 iterations at 1e8 photons take 0.089 instead of 0.095 s; the wavelength-term and
// variable-n forms would spill to scratch under that bound and keep theirs)
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock, (VAR_N == 0 && !USE_E && sizeof(T) == 8) ? 4 : 1) k_mixed(pcl_mixed_args<T> a) {
    pcl_mixed_body<T, USE_E, VAR_N>(a);
}

// constant n, three rows of 64 particles per trip, velocities in LDS (pcl_mixed_body_lds): fuller dense passes below a hit
// fraction of ~0.33 (step_mixed_t picks)
template <typename T, bool USE_E>
__global__ void __launch_bounds__(kBlock, 4) k_mixed3(pcl_mixed_args<T> a) {
    pcl_mixed_body_lds<T, USE_E, 0, 3>(a);
}

// the tracked subset's positions over the passes of the next K-pass launch (pcl_trace_body): one thread per tracked photon
template <typename T, bool USE_E, int VAR_N>
__global__ void __launch_bounds__(kBlock) k_trace(pcl_trace_args<T> a) {
    pcl_trace_body<T, USE_E, VAR_N>(a);
}

// where are the tracked ids in a store whose ids are explicit (it has been compacted, or was uploaded with ids)?  One sweep
// over the id row: a slot whose id is in ``want`` (ascending) writes itself into slot_out (preset to -1).  8 B per slot.
// ``alive``: the store's alive bits, if it is behind a mask -- a dead slot's id is whatever was left there (the photon may live
// on in another slot of its segment: pcl_mixed_put) and must not be taken for the photon.
__global__ void __launch_bounds__(kBlock) k_trace_slots(const int64_t *__restrict__ ids, const uint64_t *__restrict__ alive, int64_t N,
                                                        const int64_t *__restrict__ want, int n_want, int64_t *__restrict__ slot_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t lo_id = want[0], hi_id = want[n_want - 1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        if (alive && !((alive[i >> 6] >> (i & 63)) & 1ull)) continue;
        const int64_t id = ids[i];
        if (id < lo_id || id > hi_id) continue;
        int lo = 0, hi = n_want - 1;
        while (lo < hi) { // first element >= id
            const int mid = (lo + hi) >> 1;
            if (want[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        if (want[lo] == id) slot_out[lo] = i;
    }
}

// The variable-n shape is an argument of the ahead-of-time kernels, so a body compiled once carries all three expressions and
// the axis select through its loops: the K-step pass ran 11 % slower than its hipRTC specialisation for that alone (1.60e11
// against 1.78e11; with one literal shape: 1.78e11).  The arithmetic-heavy kernels therefore exist once more for each of the
// two one-component shapes on each axis, what the reference's examples use (VAR_N = 2 .. 7: pcl_device.h); the host picks.
#define PCL_AOT_LAUNCH_SHAPED_E(K, T, E, vn, grid, args)                                                            \
    do {                                                                                                          \
        switch (vn) {                                                                                             \
        case 0: hipLaunchKernelGGL((K<T, E, 0>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 2: hipLaunchKernelGGL((K<T, E, 2>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 3: hipLaunchKernelGGL((K<T, E, 3>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 4: hipLaunchKernelGGL((K<T, E, 4>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 5: hipLaunchKernelGGL((K<T, E, 5>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 6: hipLaunchKernelGGL((K<T, E, 6>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 7: hipLaunchKernelGGL((K<T, E, 7>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        case 8: hipLaunchKernelGGL((K<T, E, 8>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;           \
        default: hipLaunchKernelGGL((K<T, E, 1>), dim3(grid), dim3(kBlock), 0, ctx->stream, args); break;          \
        }                                                                                                         \
    } while (0)
#define PCL_AOT_LAUNCH_SHAPED(K, T, use_e, var_n, grid, args)                                                       \
    do {                                                                                                          \
        const int sh_ = (args).np.shape, ax_ = (args).np.axis;                                                     \
        const int vn_ = !(var_n) ? 0                                                                              \
                        : ((sh_ == PCL_NPROF_EXP_OFFSET || sh_ == PCL_NPROF_EXP_SCALE) && ax_ >= 0 && ax_ <= 2)    \
                            ? PCL_VARN_SHAPED(sh_ == PCL_NPROF_EXP_SCALE ? 1 : 0, ax_)                             \
                            : (sh_ == PCL_NPROF_EXP_RADIAL ? 8 : 1);                                               \
        if (use_e) PCL_AOT_LAUNCH_SHAPED_E(K, T, true, vn_, grid, args);                                           \
        else PCL_AOT_LAUNCH_SHAPED_E(K, T, false, vn_, grid, args);                                                \
    } while (0)

// launch K<[T,] USE_E, VAR_N> for run-time use_e / var_n
#define PCL_AOT_LAUNCH(K, T, use_e, var_n, grid, args)                                                              \
    do {                                                                                                          \
        if (use_e) {                                                                                              \
            if (var_n) hipLaunchKernelGGL((K<T, true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, args);     \
            else hipLaunchKernelGGL((K<T, true, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, args);          \
        } else {                                                                                                  \
            if (var_n) hipLaunchKernelGGL((K<T, false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, args);    \
            else hipLaunchKernelGGL((K<T, false, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, args);         \
        }                                                                                                         \
    } while (0)

// is any element of three rows anything but +0.0 (bit pattern 0)?  flag |= 1
template <typename W>
__global__ void __launch_bounds__(kBlock) k_any_nonzero(const W *__restrict__ a0, const W *__restrict__ a1, const W *__restrict__ a2,
                                                        int64_t N, int64_t ts, int *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const int64_t ti = pcl_tix(i, ts);
        any = any || a0[ti] != 0 || a1[ti] != 0 || a2[ti] != 0;
    }
    if (pcl_ballot(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// per-photon wavelength factor pow((h*c)/E, -4) (light.py:301): E never changes during a run, so the
// store caches it; the fast fused path multiplies by the cached value (bit-identical: same device pow)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_lam4(const T *__restrict__ E, T *__restrict__ lam4, T h, T c, int64_t N,
                                                 int64_t ts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const int64_t ti = pcl_tix(i, ts);
        lam4[ti] = pcl_wavelength_term<T>(h, c, E[ti]);
    }
}

// collision probability of every particle as the scatter kernels compute it (light.py:299-306, constant n), dense output:
// what the host needs to replay the data-dependent RNG order of the reference's CPU paths (light.py:216-223, 335-350)
template <typename T, bool USE_E>
__global__ void __launch_bounds__(kBlock) k_pcoll(const T *__restrict__ d0, const T *__restrict__ d1, const T *__restrict__ d2,
                                                  const T *__restrict__ E, T A, T n, T h, T c, T *__restrict__ out, int64_t N,
                                                  int64_t ts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const int64_t ti = pcl_tix(i, ts);
        const pcl_nprof<T> none = {0, 0, (T)0, (T)0, (T)0};
        out[i] = pcl_pcoll<T, USE_E, false>(none, A, n, h, c, d0[ti], d1[ti], d2[ti], (T)0, (T)0, (T)0, USE_E ? E[ti] : (T)0);
    }
}

// ---- delete step, pass 1: flag -> wave64 ballot keep-mask + per-tile survivor count --------------
// One workgroup per 2048-particle tile; wave w owns rows 8w..8w+7 (64 consecutive particles per
// row, lane == particle so the ballot bit order IS the particle order -> stable compaction).
template <typename T>
struct delmask_args {
    const T *d0, *d1, *d2;
    const T *rand;             // PCL_RNG_INPUT
    const int64_t *ids;        // PCL_RNG_PHILOX with materialised ids, else NULL
    const unsigned char *kind; // NULL = all photons
    const int32_t *flags_in;   // when non-NULL: take flags from memory instead (pcl_k_compact_indices)
    uint64_t *masks;           // [n_tiles * 32] bit l of row mask = particle survives
    int32_t *tile_keep;        // [n_tiles]
    int64_t id_base, N;
    int64_t ts;
    T An;                      // A * n, rounded once like the kernel's left-to-right product
    uint64_t seed;
    uint32_t step;
    int rng_mode;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_delete_mask(delmask_args<T> a) {
    typedef pcl_rt<T> R;
    __shared__ int s_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    int kept = 0;
#pragma unroll 4
    for (int rr = 0; rr < kTileRows / 4; ++rr) {
        const int row = wave * (kTileRows / 4) + rr;
        const int64_t i = tile * kTile + (int64_t)row * 64 + lane;
        bool keep = false;
        if (i < a.N) {
            if (a.flags_in) {
                keep = (a.flags_in[i] == 0);
            } else {
                const bool photon = a.kind ? (a.kind[i] != 0) : true;
                const int64_t ti = pcl_tix(i, a.ts);
                const T pcoll = R::mul(a.An, pcl_step_norm<T>(a.d0[ti], a.d1[ti], a.d2[ti]));
                T rand;
                if (a.rng_mode == PCL_RNG_PHX) {
                    const uint64_t id = (uint64_t)(a.ids ? a.ids[i] : a.id_base + i);
                    rand = pcl_draw_rand<T>(id, a.step, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
                } else {
                    rand = a.rand[i];
                }
                keep = !(photon && (pcoll >= rand));
            }
        }
        const uint64_t m = pcl_ballot(keep);
        if (lane == 0) a.masks[tile * kTileRows + row] = m;
        kept += __popcll(m);
    }
    if (lane == 0) s_cnt[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---- delete step, pass 2: exclusive scan of the per-tile counts (one workgroup) ------------------
// Also decides which pass-3 kernel runs (``*choice``): both are enqueued behind the scan and the one that was not chosen
// returns at once -- the survivor fraction is only known here, on the device, and the host must not wait for it.
//   1  k_compact_lds   (16-byte traffic, whole tile through LDS): wins when most particles survive and the store is large
//   0  k_compact_count (8 bytes per surviving lane, nothing read for the dead): wins when few survive or the store is small
__global__ void __launch_bounds__(1024) k_tile_scan(const int32_t *__restrict__ tile_keep, int64_t n_tiles,
                                                    int64_t *__restrict__ tile_off, int64_t *__restrict__ total,
                                                    int64_t n_particles, int *__restrict__ choice,
                                                    int64_t *__restrict__ total_host /* pinned, device-visible */, int lds_min_pct) {
    __shared__ int64_t s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t per = div_up(n_tiles, 1024);
    const int64_t lo = (int64_t)t * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    int64_t sum = 0;
#pragma unroll 8
    for (int64_t k = lo; k < hi; ++k) sum += tile_keep[k]; // (independent loads: unrolled, they are in flight together)
    // inclusive scan of the 1024 partial sums: wave shuffle scan, then scan of the 16 wave totals
    int64_t inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int64_t w = lane < 16 ? s_wave[lane] : 0;
        for (int off = 1; off < 16; off <<= 1) {
            const int64_t up = __shfl_up(w, off, 64);
            if (lane >= off) w += up;
        }
        if (lane < 16) s_wave[lane] = w; // inclusive wave totals
    }
    __syncthreads();
    int64_t run = (wave ? s_wave[wave - 1] : 0) + inc - sum; // exclusive prefix of this thread's chunk
#pragma unroll 8
    for (int64_t k = lo; k < hi; ++k) {
        tile_off[k] = run;
        run += tile_keep[k];
    }
    if (t == 1023) {
        *total = s_wave[15];
        *total_host = s_wave[15]; // straight into the host's pinned mirror: no copy to enqueue, the event below publishes it
        *choice = (n_particles >= (int64_t)1 << 22 && s_wave[15] * 100 > n_particles * lds_min_pct) ? 1 : 0; // > 35 % survive
    }
}

// ---- ScatterMeasureStep(measure_E=True): which photons crossed a plane in this step's move ----------------
//   (r - dr <= L <= r) or (r - dr >= L >= r) on the plane's axis, physicl/light.py:385-399; mask bit = crossed.
//   The energies of those photons are then gathered in particle order with the compaction kernels.
template <typename T>
struct crossmask_args {
    const T *x, *dx;           // the plane's axis of r and of dr
    const unsigned char *kind;
    uint64_t *masks;
    int32_t *tile_keep;
    int64_t N, ts;
    T L;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_cross_mask(crossmask_args<T> a) {
    typedef pcl_rt<T> R;
    __shared__ int s_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    int kept = 0;
#pragma unroll 4
    for (int rr = 0; rr < kTileRows / 4; ++rr) {
        const int row = wave * (kTileRows / 4) + rr;
        const int64_t i = tile * kTile + (int64_t)row * 64 + lane;
        bool cross = false;
        if (i < a.N && (a.kind ? (a.kind[i] != 0) : true)) {
            const int64_t ti = pcl_tix(i, a.ts);
            const T x = a.x[ti], prev = R::sub(x, a.dx[ti]);
            cross = (prev <= a.L && a.L <= x) || (prev >= a.L && a.L >= x);
        }
        const uint64_t m = pcl_ballot(cross);
        if (lane == 0) a.masks[tile * kTileRows + row] = m;
        kept += __popcll(m);
    }
    if (lane == 0) s_cnt[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---- delete step, pass 3: stable compaction of NF arrays of W-sized words (+ ids, kind bytes) -----
// Reads: 1 bit of mask per particle + the survivors' state; writes the survivors densely, in order.
constexpr int kMaxCompactFields = 16;
struct compact_args {
    const void *src[kMaxCompactFields];
    void *dst[kMaxCompactFields];
    const unsigned char *ksrc;
    unsigned char *kdst;
    int64_t *ids_dst;       // non-NULL with ids_src == NULL: write id_base + i (ids were implicit)
    const int64_t *ids_src;
    int64_t *idx_dst;       // non-NULL: write the source index i (pcl_k_compact_indices)
    const uint64_t *masks;
    const int64_t *tile_off;
    int64_t id_base, N;
    int64_t ts; // tile stride of BOTH slabs (elements)
    int dense_dst; // 1: dst[] are plain dense arrays (pcl_step_plane_energies), not rows of a slab
    const int *choice; // non-NULL: which of the two pass-3 kernels the scan picked (k_tile_scan)
    uint32_t sparse_max; // k_compact_count: a wave with at most this many survivors (of 512 slots) takes them survivor by survivor
};

template <typename W, int NF>
__global__ void __launch_bounds__(kBlock) k_compact(compact_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const uint64_t *tm = a.masks + tile * kTileRows;
    int64_t dest = a.tile_off[tile];
    for (int r = 0; r < wave * (kTileRows / 4); ++r) dest += __popcll(tm[r]); // rows owned by earlier waves
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll 2
    for (int rr = 0; rr < kTileRows / 4; ++rr) {
        const int row = wave * (kTileRows / 4) + rr;
        const uint64_t m = tm[row];
        const int64_t i = tile * kTile + (int64_t)row * 64 + lane;
        if ((m >> lane) & 1ull) {
            const int64_t o = dest + __popcll(m & below);
            W val[NF > 0 ? NF : 1];
            const int64_t ti = pcl_tix(i, a.ts), to = a.dense_dst ? o : pcl_tix(o, a.ts);
#pragma unroll
            for (int f = 0; f < NF; ++f) val[f] = static_cast<const W *>(a.src[f])[ti];
#pragma unroll
            for (int f = 0; f < NF; ++f) static_cast<W *>(a.dst[f])[to] = val[f];
            if (a.ids_dst) a.ids_dst[o] = a.ids_src ? a.ids_src[i] : a.id_base + i;
            if (a.kdst) a.kdst[o] = a.ksrc[i];
            if (a.idx_dst) a.idx_dst[o] = i;
        }
        dest += __popcll(m);
    }
}

// ---- fused loop body for delete simulations, pass 1: Newton + delete flag -> ballot masks ---------------
//   physicl/newton.py:15-16 then physicl/light.py:239-249, one pass over r and v (72 B fp64 per particle when
//   dr stays implicit, 96 B when it is written).  Tile geometry of k_delete_mask (lane == particle).
template <typename T>
struct newtonmask_args {
    const T *v[3];
    T *r[3];
    T *dr[3];                  // written unless lazy
    const T *rand;             // PCL_RNG_INPUT
    const int64_t *ids;
    const unsigned char *kind;
    uint64_t *masks;
    int32_t *tile_keep;
    int64_t id_base, N;
    int64_t ts;
    T dt, An;
    uint64_t seed;
    uint32_t step;
    int rng_mode;
    int lazy;
    int flag_only; // 1: do not touch r at all -- the flag needs |v * dt| only; pass 3 moves the survivors (lazy mode)
    uint64_t *zero_cnt; // counter slots pass 3 will add into: cleared here (one stream operation fewer than a memset)
    int n_zero;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_newton_mask(newtonmask_args<T> a) {
    typedef pcl_rt<T> R;
    __shared__ int s_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    if (tile == 0 && (int)threadIdx.x < a.n_zero) a.zero_cnt[threadIdx.x] = 0;
    int kept = 0;
#pragma unroll 2
    for (int rr = 0; rr < kTileRows / 4; ++rr) {
        const int row = wave * (kTileRows / 4) + rr;
        const int64_t i = tile * kTile + (int64_t)row * 64 + lane;
        bool keep = false;
        if (i < a.N) {
            T d[3];
            const int64_t ti = pcl_tix(i, a.ts);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = R::mul(a.v[k][ti], a.dt);
                if (!a.flag_only) {
                    a.r[k][ti] = R::add(a.r[k][ti], d[k]);
                    if (!a.lazy) a.dr[k][ti] = d[k];
                }
            }
            const bool photon = a.kind ? (a.kind[i] != 0) : true;
            const T pcoll = R::mul(a.An, pcl_step_norm<T>(d[0], d[1], d[2]));
            T rand;
            if (a.rng_mode == PCL_RNG_PHX) {
                const uint64_t id = (uint64_t)(a.ids ? a.ids[i] : a.id_base + i);
                rand = pcl_draw_rand<T>(id, a.step, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
            } else {
                rand = a.rand[i];
            }
            keep = !(photon && (pcoll >= rand));
        }
        const uint64_t m = pcl_ballot(keep);
        if (lane == 0) a.masks[tile * kTileRows + row] = m;
        kept += __popcll(m);
    }
    if (lane == 0) s_cnt[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__device__ __forceinline__ uint64_t spread_bits(uint32_t x) { // bit i -> bit 2i
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// ---- pass 1, flag-only form for all-photon stores with the device RNG (the lazy delete pipeline's usual case) --------
//   Same flag, masks and tile counts as k_newton_mask with flag_only = 1; a lane takes TWO neighbouring particles so
//   that the three v rows (and the id row after a compaction) come in with 16-B loads: half the load instructions per
//   byte, twice the bytes in flight per wave.  A wave covers a pair of mask rows per trip; the two ballots (even /
//   odd particles) are interleaved into the two row words.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_flag_mask2(newtonmask_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    __shared__ int s_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    if (tile == 0 && (int)threadIdx.x < a.n_zero) a.zero_cnt[threadIdx.x] = 0;
    int kept = 0;
#pragma unroll 2
    for (int pp = 0; pp < kTileRows / 8; ++pp) {
        const int row = (wave * (kTileRows / 8) + pp) * 2;
        const int64_t i = tile * kTile + (int64_t)row * 64 + 2 * lane;
        bool keep0 = false, keep1 = false;
        if (i < a.N) {
            const bool two = i + 1 < a.N;
            const int64_t ti = pcl_tix(i, a.ts);
            T vx[2], vy[2], vz[2];
            uint64_t id[2];
            if (two) {
                const T2 x = *reinterpret_cast<const T2 *>(a.v[0] + ti), y = *reinterpret_cast<const T2 *>(a.v[1] + ti),
                         z = *reinterpret_cast<const T2 *>(a.v[2] + ti);
                vx[0] = x.x, vx[1] = x.y, vy[0] = y.x, vy[1] = y.y, vz[0] = z.x, vz[1] = z.y;
                if (a.ids) {
                    const longlong2 q = *reinterpret_cast<const longlong2 *>(a.ids + i);
                    id[0] = (uint64_t)q.x, id[1] = (uint64_t)q.y;
                } else {
                    id[0] = (uint64_t)(a.id_base + i), id[1] = id[0] + 1;
                }
            } else {
                vx[0] = vx[1] = a.v[0][ti], vy[0] = vy[1] = a.v[1][ti], vz[0] = vz[1] = a.v[2][ti];
                id[0] = id[1] = (uint64_t)(a.ids ? a.ids[i] : a.id_base + i);
            }
            bool kp[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const T pcoll = R::mul(a.An, pcl_step_norm<T>(R::mul(vx[e], a.dt), R::mul(vy[e], a.dt), R::mul(vz[e], a.dt)));
                const T rand = pcl_draw_rand<T>(id[e], a.step, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
                kp[e] = !(pcoll >= rand);
            }
            keep0 = kp[0];
            keep1 = two && kp[1];
        }
        const uint64_t b0 = pcl_ballot(keep0), b1 = pcl_ballot(keep1);
        if (lane == 0) {
            a.masks[tile * kTileRows + row] = spread_bits((uint32_t)b0) | (spread_bits((uint32_t)b1) << 1);
            a.masks[tile * kTileRows + row + 1] = spread_bits((uint32_t)(b0 >> 32)) | (spread_bits((uint32_t)(b1 >> 32)) << 1);
        }
        kept += __popcll(b0) + __popcll(b1);
    }
    if (lane == 0) s_cnt[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---- the delete loop body behind an ALIVE MASK: one kernel, nothing moves ----------------------------------------------
//   physicl/newton.py:15-16 + physicl/light.py:239-249 + the removal loop light.py:258-260 (which only needs the
//   survivors to keep their order) + the counting measure steps behind it (light.py:385-399, 414-431).
//   The store keeps every slot it had at its last compaction; a bit per slot says whether the photon is still there
//   (the ballot masks of the delete passes ARE that mask: bit set = alive).  A loop body then is ONE kernel: per
//   64-slot row pair read the alive words, for the alive photons read v (16-B loads, a lane takes two neighbouring
//   slots as in k_flag_mask2), draw the flag, write the new alive words and the tile's alive count; the alive count of
//   the whole store and the measure counters of the survivors are added up over the grid and the LAST workgroup to
//   finish writes them straight into the host's pinned counter block -- no scan, no second kernel, no copy to enqueue.
//   Dead slots cost their share of a 64-B sector and nothing else; the host compacts (k_tile_scan + k_compact_*, fed
//   with these very masks and tile counts) only when fewer than half of the slots are alive (pcl_step_fused_delete).
//   r is not touched either: a photon of a delete run keeps its velocity, so the moves of the bodies since r was last
//   written (at most kPendMax, their dt in pend_dt[]) are applied in registers, in order, with the operations of
//   newton.py:15-16 -- to whoever needs r: the plane counters here (NEED_R), the compaction, k_apply_pending.
constexpr int kPendMax = PCL_PEND_MAX;
template <typename T>
struct alive_args {
    const T *v[3];
    T *r[3];                    // NEED_R only; written back when write_r
    const int64_t *ids;         // NULL: id = id_base + slot
    uint64_t *masks;            // alive bits, read (unless fresh) and written in place
    uint64_t *masks_prev;       // the alive bits as they were before this body (pcl_store_last_delete_flags)
    int32_t *tile_keep;         // alive photons per tile after this body (input of k_tile_scan)
    unsigned long long *acc;    // device accumulators: [0] ticket, [1] alive, [2..4] sign counts, [5..] plane crossings
    uint64_t *host;             // pinned host block: [kCounterSlots - 1] alive, [1..3] sign, [4..] planes
    int64_t id_base, slots;     // slots: extent of the store (alive + dead)
    int64_t ts;
    T dt, An;
    T pend_dt[kPendMax];        // moves r has not seen yet, oldest first, run-length: pend_rep[q] moves of pend_dt[q]
    int pend_rep[kPendMax];
    int n_pend;
    uint64_t seed;
    uint32_t step;
    uint64_t *zero_cnt;         // count == 0: counter slots the compaction behind this launch adds into, cleared here
    int n_zero;
    uint64_t seq;               // count == 1: written to host[kCounterSlots - 6] after the totals -- the host polls for it
    int r_axes;                 // NEED_R: bit k = component k of r is read (a plane lies on that axis, or write_r: all three)
    int write_r;                // NEED_R: store r with every move up to and including this body's applied (the list of
                                // pending moves was full): the caller starts a new, empty list
    int fresh;                  // 1: the store is dense -- every slot below ``slots`` is alive, masks are not read
    int count;                  // 1: measure counters + totals to the host (a body without compaction); 0: flags only
    int n_planes;
    int plane_ax[PCL_MAX_PLANES];
    T plane_L[PCL_MAX_PLANES];
};

template <typename T, bool NEED_R>
__global__ void __launch_bounds__(kBlock) k_delete_alive(alive_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    __shared__ uint32_t s_cnt[4 + PCL_MAX_PLANES]; // [0] alive, [1..3] sign, [4..] planes: sums over the workgroup's tiles
    __shared__ uint32_t s_keep[2][kBlock / 64];    // alive per wave of the current tile (two buffers: one barrier per tile)
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4 + PCL_MAX_PLANES) s_cnt[threadIdx.x] = 0;
    if (!a.count && blockIdx.x == 0 && (int)threadIdx.x < a.n_zero) a.zero_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t kept_all = 0, w_s[3] = {0, 0, 0};
    const bool hi = lane >= 32;
    const int bit = 2 * (lane & 31);
    const int64_t n_tiles = (a.slots + kTile - 1) / kTile;
    int par = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
        uint32_t kept = 0;
        // two row pairs (256 slots) per trip: their loads are issued together, unconditionally -- whole tiles exist in
        // the slab whatever the extent, and a dead lane's share of a sector is fetched anyway -- so that a wave has
        // 2 x (3 [+ 3] + 1) 16-byte loads in flight instead of one row pair's
#pragma unroll 1
        for (int pp = 0; pp < kTileRows / 8; pp += 2) {
            uint64_t m_lo[2], m_hi[2];
            T vv[2][3][2], xx[2][3][2];
            uint64_t id[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (wave * (kTileRows / 8) + pp + u) * 2;
                const int64_t i = tile * kTile + (int64_t)row * 64 + 2 * lane; // the lane's two slots: i, i + 1
                if (a.fresh) {
                    const int64_t left = a.slots - (tile * kTile + (int64_t)row * 64); // slots from the row pair's start
                    m_lo[u] = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
                    m_hi[u] = left >= 128 ? ~0ull : (left > 64 ? (1ull << (left - 64)) - 1ull : 0ull);
                } else {
                    m_lo[u] = a.masks[tile * kTileRows + row];
                    m_hi[u] = a.masks[tile * kTileRows + row + 1];
                }
                const int64_t ti = pcl_tix(i, a.ts);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T2 q = *reinterpret_cast<const T2 *>(a.v[k] + ti);
                    vv[u][k][0] = q.x, vv[u][k][1] = q.y;
                }
                if constexpr (NEED_R) { // only the components somebody looks at: the planes' axes, or all three for write_r
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        xx[u][k][0] = xx[u][k][1] = (T)0;
                        if ((a.r_axes >> k) & 1) { // (uniform)
                            const T2 q = *reinterpret_cast<const T2 *>(a.r[k] + ti);
                            xx[u][k][0] = q.x, xx[u][k][1] = q.y;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) xx[u][k][0] = xx[u][k][1] = (T)0;
                }
                if (a.ids) { // dense array of ``capacity`` ids padded to whole 64-element groups: pairs beyond the extent read pair 0
                    const longlong2 q = *reinterpret_cast<const longlong2 *>(a.ids + (i < a.slots ? i : 0));
                    id[u][0] = (uint64_t)q.x, id[u][1] = (uint64_t)q.y;
                } else {
                    id[u][0] = (uint64_t)(a.id_base + i), id[u][1] = id[u][0] + 1;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (wave * (kTileRows / 8) + pp + u) * 2;
                if (lane == 0) {
                    a.masks_prev[tile * kTileRows + row] = m_lo[u];
                    a.masks_prev[tile * kTileRows + row + 1] = m_hi[u];
                }
                if ((m_lo[u] | m_hi[u]) == 0ull) { // nobody left in these 128 slots (wave-uniform; their r may stay stale)
                    if (lane == 0 && a.fresh) a.masks[tile * kTileRows + row] = 0ull, a.masks[tile * kTileRows + row + 1] = 0ull;
                    continue;
                }
                const uint64_t mm = hi ? m_hi[u] : m_lo[u];
                const bool al0 = (mm >> bit) & 1ull, al1 = (mm >> (bit + 1)) & 1ull;
                bool kp[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const T d0 = R::mul(vv[u][0][e], a.dt), d1 = R::mul(vv[u][1][e], a.dt), d2 = R::mul(vv[u][2][e], a.dt); // newton.py:15
                    const T pcoll = R::mul(a.An, pcl_step_norm<T>(d0, d1, d2));                                                // light.py:241-247
                    const T rand = pcl_draw_rand<T>(id[u][e], a.step, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
                    kp[e] = !(pcoll >= rand);
                    if constexpr (NEED_R) { // r as the reference holds it after this body's move: earlier bodies' moves first
                        for (int q = 0; q < a.n_pend; ++q) {
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                if ((a.r_axes >> k) & 1) {
                                    const T dq = R::mul(vv[u][k][e], a.pend_dt[q]); // (the same product every time: formed once)
                                    for (int t = 0; t < a.pend_rep[q]; ++t) xx[u][k][e] = R::add(xx[u][k][e], dq);
                                }
                        }
                        xx[u][0][e] = R::add(xx[u][0][e], d0);                                                                // newton.py:16
                        xx[u][1][e] = R::add(xx[u][1][e], d1);
                        xx[u][2][e] = R::add(xx[u][2][e], d2);
                    }
                }
                if constexpr (NEED_R) {
                    if (a.write_r) { // (wave-uniform; dead slots get a value too -- nobody reads them again)
                        const int64_t ti = pcl_tix(tile * kTile + (int64_t)row * 64 + 2 * lane, a.ts);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            T2 q;
                            q.x = xx[u][k][0], q.y = xx[u][k][1];
                            *reinterpret_cast<T2 *>(a.r[k] + ti) = q;
                        }
                    }
                }
                const bool keep0 = al0 && kp[0], keep1 = al1 && kp[1];
                const uint64_t b0 = pcl_ballot(keep0), b1 = pcl_ballot(keep1);
                if (lane == 0) {
                    a.masks[tile * kTileRows + row] = spread_bits((uint32_t)b0) | (spread_bits((uint32_t)b1) << 1);
                    a.masks[tile * kTileRows + row + 1] = spread_bits((uint32_t)(b0 >> 32)) | (spread_bits((uint32_t)(b1 >> 32)) << 1);
                }
                kept += (uint32_t)(__popcll(b0) + __popcll(b1));
                if (a.count && a.n_planes >= 0) { // wave-uniform
#pragma unroll
                    for (int k = 0; k < 3; ++k)                                                                              // light.py:424-426
                        w_s[k] += (uint32_t)__popcll(pcl_ballot(keep0 && vv[u][k][0] > (T)0)) + (uint32_t)__popcll(pcl_ballot(keep1 && vv[u][k][1] > (T)0));
                    if constexpr (NEED_R) {
                        for (int p = 0; p < a.n_planes; ++p) {                                                               // light.py:385-399
                            const int ax = a.plane_ax[p];
                            const T L = a.plane_L[p];
                            uint32_t np = 0;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const T x = pcl_pick<T>(ax, xx[u][0][e], xx[u][1][e], xx[u][2][e]);
                                const T prev = R::sub(x, R::mul(pcl_pick<T>(ax, vv[u][0][e], vv[u][1][e], vv[u][2][e]), a.dt));
                                np += (uint32_t)__popcll(pcl_ballot((e ? keep1 : keep0) && ((prev <= L && L <= x) || (prev >= L && L >= x))));
                            }
                            if (lane == 0 && np) atomicAdd(&s_cnt[4 + p], np);
                        }
                    }
                }
            }
        }
        kept_all += kept;
        if (lane == 0) s_keep[par][wave] = kept;
        __syncthreads();
        if (threadIdx.x == 0) a.tile_keep[tile] = (int32_t)(s_keep[par][0] + s_keep[par][1] + s_keep[par][2] + s_keep[par][3]);
    }
    if (!a.count) return;
    if (lane == 0) {
        atomicAdd(&s_cnt[0], kept_all);
        if (a.n_planes >= 0)
            for (int k = 0; k < 3; ++k) atomicAdd(&s_cnt[1 + k], w_s[k]);
    }
    __syncthreads();
    // grid totals: every workgroup adds its sums, the last one to arrive hands them to the host and leaves the
    // accumulators zero for the next launch.  Agent-scope atomics only (they are coherent across the XCDs' L2s by
    // themselves); NO fence: an agent-scope release would write the L2's dirty lines back once per workgroup.  The sums
    // are in place before the ticket is drawn because the adds return (their data comes back before the barrier).
    // What this relies on (gfx942 / gfx950 ISA, not the HIP memory model -- ADVICE r3): a RETURNING global atomic is
    // performed at the memory side (agent scope: in the device-coherent L2 / MALL path shared by the XCDs) before its value
    // travels back, and s_waitcnt vmcnt(0) -- which the consumed return value forces ahead of the s_barrier -- waits for that
    // value; the ticket's own atomic is issued after the barrier, so whoever draws the last ticket finds every sum applied.
    // The host double-checks the one number that matters (wait_alive: the alive count can only fall; densify: the
    // compaction must find exactly ``count`` photons), and a launch that dies midway has its accumulators cleared
    // (reset_alive_acc) so that the next launch's totals start from zero.
    const int nslots = 4 + (a.n_planes > 0 ? a.n_planes : 0);
    unsigned long long seen = 0;
    if ((int)threadIdx.x < nslots && s_cnt[threadIdx.x])
        seen = __hip_atomic_fetch_add(&a.acc[1 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(seen) : "memory"); // the returned value is consumed: the add has been performed
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&a.acc[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!s_last) return;
    if ((int)threadIdx.x < nslots) {
        const unsigned long long v = __hip_atomic_exchange(&a.acc[1 + threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        volatile uint64_t *h = a.host;
        if (threadIdx.x == 0) h[kCounterSlots - 1] = v;
        else h[threadIdx.x] = v;
        __threadfence_system(); // (once per launch, by the last workgroup's first wave: the totals are in host memory ...
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_exchange(&a.acc[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ... before the launch's sequence number, which the host is polling for: it learns the count a few microseconds
        // after the last workgroup is done instead of waiting for the kernel's completion signal to travel)
        __hip_atomic_store(&a.host[kCounterSlots - 6], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the pending moves of an alive-mask store made real: r = (...((r + v*dt_1) + v*dt_2)...) for every slot of the extent
// (dead slots included -- nobody reads them again), when the list is full and no compaction is due
template <typename T>
__global__ void __launch_bounds__(kBlock) k_apply_pending(alive_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; 2 * p < a.slots; p += (int64_t)gridDim.x * kBlock) {
        const int64_t ti = pcl_tix(2 * p, a.ts);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T2 v = *reinterpret_cast<const T2 *>(a.v[k] + ti);
            T2 x = *reinterpret_cast<const T2 *>(a.r[k] + ti);
            for (int q = 0; q < a.n_pend; ++q) {
                const T dx = R::mul(v.x, a.pend_dt[q]), dy = R::mul(v.y, a.pend_dt[q]);
                for (int t = 0; t < a.pend_rep[q]; ++t) {
                    x.x = R::add(x.x, dx);
                    x.y = R::add(x.y, dy);
                }
            }
            *reinterpret_cast<T2 *>(a.r[k] + ti) = x;
        }
    }
}

// ---- delete loop bodies AHEAD of their calls (small stores) -----------------------------------------------------------
//   A loop body of a small store is latency: a 5-10 us kernel and a 15-20 us round trip to the host, which has to learn
//   the alive count before it can ask ``exit``.  But the bodies of a delete run are a deterministic function of the
//   store and of the call's arguments, and a run repeats the call with the launch number advanced by one.  When the
//   library has seen that pattern and the extent is small (pcl_step_fused_delete, "ahead"), ONE launch of this kernel
//   works out the next K bodies WITHOUT touching the store: per slot the body it is removed in (``death``: 0 = was dead,
//   k = removed by the k-th body from here, 255 = survives all K) and one counter row per body in the host's pinned
//   block.  The following K - 1 calls -- if they are the predicted ones -- are answered from those rows without a launch;
//   whenever anything else looks at the store, k_ahead_commit first makes the state after the bodies handed out so far
//   real (masks, masks_prev, tile counts, r).  Nothing is guessed: a call that does not match simply commits and runs
//   the ordinary way.  Same operations per photon as k_delete_alive (tests/test_gpu_alive_mask.py runs both).
constexpr int kAheadMax = 64;                       // bodies per launch, at most
constexpr int kAheadRow = 4 + PCL_MAX_PLANES;       // counters per body: alive, sign x 3, planes
constexpr int kAheadWork = 6;                       // k_delete_ahead_live's own work tally: groups loaded (first pass of two bodies, of one), rounds of two bodies, of one;
                                                    // [4], [5]: shader cycles and 100 MHz ticks of the workgroups' lifetimes (the clock under the launch)
constexpr int kAheadAcc = 1 + kAheadMax * kAheadRow + kAheadWork; // device accumulators: ticket, rows, work; pinned block: rows, sequence word, work
template <typename T>
struct ahead_args {
    const T *v[3];
    T *r[3];
    const int64_t *ids;
    const uint64_t *masks;      // alive bits before the first body (unless fresh)
    uint8_t *death;
    unsigned long long *acc;    // device accumulators: [0] ticket, [1 + k * kAheadRow + c], [1 + kAheadMax * kAheadRow + i] work tally
    uint64_t *host;             // pinned: [k * kAheadRow + c]; [kAheadMax * kAheadRow] = the launch's sequence number, then the work tally
    int64_t id_base, slots, ts;
    T dt, An;
    T pend_dt[kPendMax];
    int pend_rep[kPendMax];
    int n_pend;
    uint64_t seed;
    uint32_t step0;
    int K;
    int fresh;
    int r_axes;
    int n_planes;               // -1: no measure step (alive counts only)
    int plane_ax[PCL_MAX_PLANES];
    T plane_L[PCL_MAX_PLANES];
    uint64_t seq;
    int j;                      // k_ahead_commit: bodies handed out
    uint64_t *masks_out, *masks_prev;
    int32_t *tile_keep;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_delete_ahead(ahead_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    __shared__ uint32_t s_cnt[kAheadMax * kAheadRow];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < kAheadMax * kAheadRow; q += kBlock) s_cnt[q] = 0;
    __syncthreads();
    const bool hi = lane >= 32;
    const int bit = 2 * (lane & 31);
    const int64_t n_tiles = (a.slots + kTile - 1) / kTile;
    uint32_t t_kept = 0, t_s[3] = {0, 0, 0}, t_p0 = 0; // lane b: this wave's sums of body b (alive, sign counts, first plane)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll 1
        for (int pp = 0; pp < kTileRows / 8; ++pp) {
            const int row = (wave * (kTileRows / 8) + pp) * 2;
            const int64_t i = tile * kTile + (int64_t)row * 64 + 2 * lane; // the lane's two slots: i, i + 1
            uint64_t m_lo, m_hi;
            if (a.fresh) {
                const int64_t left = a.slots - (tile * kTile + (int64_t)row * 64);
                m_lo = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
                m_hi = left >= 128 ? ~0ull : (left > 64 ? (1ull << (left - 64)) - 1ull : 0ull);
            } else {
                m_lo = a.masks[tile * kTileRows + row];
                m_hi = a.masks[tile * kTileRows + row + 1];
            }
            uchar2 dth;
            dth.x = dth.y = 0;
            if ((m_lo | m_hi) != 0ull) { // (wave-uniform)
                const int64_t ti = pcl_tix(i, a.ts);
                T vv[3][2], xx[3][2], dd[3][2], pcoll[2];
                uint64_t id[2];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T2 q = *reinterpret_cast<const T2 *>(a.v[k] + ti);
                    vv[k][0] = q.x, vv[k][1] = q.y;
                    xx[k][0] = xx[k][1] = (T)0;
                    if ((a.r_axes >> k) & 1) {
                        const T2 x = *reinterpret_cast<const T2 *>(a.r[k] + ti);
                        xx[k][0] = x.x, xx[k][1] = x.y;
                    }
                }
                if (a.ids) {
                    const longlong2 q = *reinterpret_cast<const longlong2 *>(a.ids + (i < a.slots ? i : 0));
                    id[0] = (uint64_t)q.x, id[1] = (uint64_t)q.y;
                } else {
                    id[0] = (uint64_t)(a.id_base + i), id[1] = id[0] + 1;
                }
                const uint64_t mm = hi ? m_hi : m_lo;
                bool al[2] = {(bool)((mm >> bit) & 1ull), (bool)((mm >> (bit + 1)) & 1ull)};
                uint32_t dth_e[2] = {al[0] ? 255u : 0u, al[1] ? 255u : 0u};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) dd[k][e] = R::mul(vv[k][e], a.dt);                                           // newton.py:15
                    pcoll[e] = R::mul(a.An, pcl_step_norm<T>(dd[0][e], dd[1][e], dd[2][e]));                                 // light.py:241-247
                    for (int q = 0; q < a.n_pend; ++q) {
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if ((a.r_axes >> k) & 1) {
                                const T dq = R::mul(vv[k][e], a.pend_dt[q]);
                                for (int t = 0; t < a.pend_rep[q]; ++t) xx[k][e] = R::add(xx[k][e], dq);
                            }
                    }
                }
                // Phase 1, lane by lane: the body each slot is removed in.  One Philox block decides the steps 2m and 2m + 1
                // (pcl_draw_rand): the bodies are taken in such pairs -- the block once, no selects between its halves --
                // after a leading body when the first step is odd.  Nothing here crosses lanes but the "all decided" test.
                auto block = [&](uint32_t st, T (&lo)[2], T (&hi2)[2]) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id[e], (pcl_u32)(id[e] >> 32), st >> 1, 0u, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
                        lo[e] = R::uniform(w.x, w.y);
                        hi2[e] = R::uniform(w.z, w.w);
                    }
                };
                {
                    int b = 0;
                    T r_lo[2], r_hi[2];
                    if (a.step0 & 1u) {
                        block(a.step0, r_lo, r_hi);
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            if (dth_e[e] == 255u && pcoll[e] >= r_hi[e]) dth_e[e] = 1u;
                        b = 1;
                    }
                    for (; b < a.K; b += 2) {
                        if (pcl_ballot(dth_e[0] == 255u || dth_e[1] == 255u) == 0ull) break; // every slot of the wave has its body
                        block(a.step0 + (uint32_t)b, r_lo, r_hi);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if (dth_e[e] == 255u && pcoll[e] >= r_lo[e]) dth_e[e] = (uint32_t)(b + 1);
                            if (b + 1 < a.K && dth_e[e] == 255u && pcoll[e] >= r_hi[e]) dth_e[e] = (uint32_t)(b + 2);
                        }
                    }
                }
                // Phase 2, body by body: the counter rows of the survivors (light.py:385-399, 414-431).  Alive after body b =
                // removed later or never; the velocity signs are ballots taken once (a delete run never changes a velocity), so
                // the sign counts are scalar work; the first plane's coordinate lives in two registers per slot (its axis is
                // picked once), further planes take the general way; the sums go to lane b's accumulators.
                uint64_t sg[3][2] = {{0ull, 0ull}, {0ull, 0ull}, {0ull, 0ull}};
                if (a.n_planes >= 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) sg[k][0] = pcl_ballot(vv[k][0] > (T)0), sg[k][1] = pcl_ballot(vv[k][1] > (T)0);
                }
                T xp[2] = {(T)0, (T)0}, dp[2] = {(T)0, (T)0};
                const T L0 = a.n_planes > 0 ? a.plane_L[0] : (T)0;
                if (a.n_planes > 0) {
                    const int ax = a.plane_ax[0];
#pragma unroll
                    for (int e = 0; e < 2; ++e) xp[e] = pcl_pick<T>(ax, xx[0][e], xx[1][e], xx[2][e]), dp[e] = pcl_pick<T>(ax, dd[0][e], dd[1][e], dd[2][e]);
                }
                uint64_t cur0 = pcl_ballot(al[0]), cur1 = pcl_ballot(al[1]);
                for (int b = 0; b < a.K; ++b) {
                    if ((cur0 | cur1) == 0ull) break; // nobody of these 128 slots was left before this body (wave-uniform)
                    const uint64_t a0 = pcl_ballot(dth_e[0] > (uint32_t)(b + 1)), a1 = pcl_ballot(dth_e[1] > (uint32_t)(b + 1));
                    const bool mine = lane == b;
                    t_kept += mine ? (uint32_t)(__popcll(a0) + __popcll(a1)) : 0u;
                    if (a.n_planes >= 0) { // (uniform)
#pragma unroll
                        for (int k = 0; k < 3; ++k)                                                                          // light.py:424-426
                            t_s[k] += mine ? (uint32_t)(__popcll(a0 & sg[k][0]) + __popcll(a1 & sg[k][1])) : 0u;
                        if (a.n_planes > 0) {                                                                                // light.py:385-399
                            uint64_t c[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                xp[e] = R::add(xp[e], dp[e]);                                                                // newton.py:16
                                const T prev = R::sub(xp[e], dp[e]);
                                c[e] = pcl_ballot((prev <= L0 && L0 <= xp[e]) || (prev >= L0 && L0 >= xp[e]));
                            }
                            t_p0 += mine ? (uint32_t)(__popcll(a0 & c[0]) + __popcll(a1 & c[1])) : 0u;
                        }
                        if (a.n_planes > 1) { // the other planes: every needed component of r moves, the plane picks its own
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
#pragma unroll
                                for (int k = 0; k < 3; ++k)
                                    if ((a.r_axes >> k) & 1) xx[k][e] = R::add(xx[k][e], dd[k][e]);
                            }
                            for (int p = 1; p < a.n_planes; ++p) {
                                const int ax = a.plane_ax[p];
                                const T L = a.plane_L[p];
                                uint32_t np = 0;
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const T x = pcl_pick<T>(ax, xx[0][e], xx[1][e], xx[2][e]);
                                    const T prev = R::sub(x, pcl_pick<T>(ax, dd[0][e], dd[1][e], dd[2][e]));
                                    np += (uint32_t)__popcll((e ? a1 : a0) & pcl_ballot((prev <= L && L <= x) || (prev >= L && L >= x)));
                                }
                                if (lane == 0 && np) atomicAdd(&s_cnt[b * kAheadRow + 4 + p], np);
                            }
                        }
                    }
                    cur0 = a0, cur1 = a1;
                }
                dth.x = (unsigned char)dth_e[0], dth.y = (unsigned char)dth_e[1];
            }
            *reinterpret_cast<uchar2 *>(a.death + i) = dth; // (whole tiles exist in the buffer)
        }
    }
    if (lane < a.K) { // the wave's sums join the workgroup's
        if (t_kept) atomicAdd(&s_cnt[lane * kAheadRow + 0], t_kept);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (t_s[k]) atomicAdd(&s_cnt[lane * kAheadRow + 1 + k], t_s[k]);
        if (t_p0) atomicAdd(&s_cnt[lane * kAheadRow + 4], t_p0);
    }
    __syncthreads();
    // grid totals as in k_delete_alive: returning agent-scope atomics, the last workgroup reports
    const int nrow = 4 + (a.n_planes > 0 ? a.n_planes : 0);
    unsigned long long seen = 0;
    for (int q = threadIdx.x; q < a.K * kAheadRow; q += kBlock)
        if ((q % kAheadRow) < nrow && s_cnt[q])
            seen += __hip_atomic_fetch_add(&a.acc[1 + q], (unsigned long long)s_cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(seen) : "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&a.acc[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!s_last) return;
    volatile uint64_t *h = a.host;
    for (int q = threadIdx.x; q < a.K * kAheadRow; q += kBlock)
        h[q] = (q % kAheadRow) < nrow ? __hip_atomic_exchange(&a.acc[1 + q], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_exchange(&a.acc[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.host[kAheadMax * kAheadRow], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// (``pcoll >= uniform`` as an integer compare: pcl_thr in pcl_device.h)
template <typename T> using ahead_draw = pcl_thr<T>;

// ---- the same, for at most one plane: a body costs what the photons still ALIVE cost -----------------------------------
//   k_delete_ahead decides every slot's bodies with the slot's own lane: after four bodies three quarters of the lanes
//   work for photons that are gone, and the counter rows are ballots over all slots once more.  Here a wave takes 256
//   slots at a time and lists the alive photons in LDS (id, collision probability, coordinate and move along the plane's
//   axis, place and velocity signs); each pass over the list decides one Philox block's two bodies for 64 listed photons
//   per round -- all lanes busy --, counts the round's survivors into those bodies' rows right there and writes them back
//   to the front of the list (in place: a round writes no further than it has read).  Same operations per photon, same
//   rows, same death bytes (tests run both kernels: PCL_AHEAD_LIVE).
// IDS: the store's ids are an array (a listed photon's id travels in LDS); false: implicit, id = id_base + slot -- no id
// array in LDS, and a fifth workgroup per CU (a population's first launches, the ones that sweep the whole extent)
template <typename T, bool IDS>
__global__ void __launch_bounds__(kBlock) k_delete_ahead_live(ahead_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    typedef ahead_draw<T> D;
    typedef typename D::thr_t thr_t;
    constexpr int kGroups = 2; // groups of 128 slots (two per lane) a wave takes together
    constexpr int kBatch = kGroups * 128;
    __shared__ uint32_t s_cnt[kAheadMax * kAheadRow];
    __shared__ int s_last;
    __shared__ uint32_t s_work[kAheadWork];
    __shared__ uint64_t s_id[kBlock / 64][IDS ? kBatch : 1];
    __shared__ thr_t s_pc[kBlock / 64][kBatch]; // the collision probability as the largest draw that removes (ahead_draw)
    __shared__ T s_xp[kBlock / 64][kBatch], s_dp[kBlock / 64][kBatch];
    __shared__ uint16_t s_ix[kBlock / 64][kBatch]; // place in the batch (8 bits), velocity signs (bits 8-10)
    __shared__ uint8_t s_death[kBlock / 64][kBatch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < kAheadMax * kAheadRow; q += kBlock) s_cnt[q] = 0;
    if (threadIdx.x < kAheadWork) s_work[threadIdx.x] = 0;
    __shared__ pcl_u64 s_clk[2];
    pcl_clock_begin(s_clk);
    __syncthreads();
    const bool hi = lane >= 32;
    const int bit = 2 * (lane & 31);
    const uint64_t below = (1ull << lane) - 1ull;
    const int64_t n_tiles = (a.slots + kTile - 1) / kTile;
    const bool has_plane = a.n_planes > 0, has_signs = a.n_planes >= 0;
    const int ax0 = has_plane ? a.plane_ax[0] : 0;
    const T L0 = has_plane ? a.plane_L[0] : (T)0;
    uint32_t t_kept = 0, t_s[3] = {0, 0, 0}, t_p0 = 0; // lane b: this wave's sums of body b (alive, sign counts, the plane)
    // what this wave did, for the VALU roofline of the launch (wave-uniform: scalar adds): groups of 128 slots it loaded (their
    // photons' first Philox block decided on the spot: two bodies, or one), rounds of 64 listed photons deciding two bodies,
    // rounds deciding one (pcl_store_ahead_work)
    uint32_t w_groups2 = 0, w_groups1 = 0, w_rounds2 = 0, w_rounds1 = 0;
    // one body's row from the photons of a round that it leaves alive (``bs``: their wave mask): into lane b's accumulators.
    // The votes are taken on bare compares and combined as masks in the scalar unit (a vote on a compound predicate costs a
    // v_cndmask and a second v_cmp -- 8 SIMD-cycles, five votes per body: pcl_ballot)
    auto tally = [&](int b, uint64_t bs, uint32_t ix, T prev, T xp) {
        const bool mine = lane == b;
        t_kept += mine ? (uint32_t)__popcll(bs) : 0u;
        if (has_signs) {                                                                                                 // light.py:424-426
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint64_t bk = pcl_ballot((ix & (0x100u << k)) != 0u) & bs; // (every lane votes: not inside the select)
                t_s[k] += mine ? (uint32_t)__popcll(bk) : 0u;
            }
            if (has_plane) {                                                                                             // light.py:385-399
                const uint64_t bp = ((pcl_ballot(prev <= L0) & pcl_ballot(L0 <= xp)) | (pcl_ballot(prev >= L0) & pcl_ballot(L0 >= xp))) & bs;
                t_p0 += mine ? (uint32_t)__popcll(bp) : 0u;
            }
        }
    };
    // One Philox block of one photon: the body (or the two bodies) it decides, starting with body ``b`` -- ``single``: only the
    // block's second half is a body of this launch; ``two``: both halves are.  Counts the photon into the rows of the bodies
    // that leave it alive, advances its coordinate along the plane's axis, returns whether it is still alive afterwards and,
    // if not, the body that removed it (counted from 1).  Called by all lanes together (ballots inside).
    auto decide = [&](int b, bool single, bool two, uint32_t step, bool on, uint64_t id, thr_t pc, T &xp, T dp, uint32_t ix, uint32_t &d) -> bool {
        const pcl_u32x4 wd = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), step >> 1, 0u, (pcl_u32)a.seed, (pcl_u32)(a.seed >> 32));
        const thr_t m_first = single ? D::draw(wd.z, wd.w) : D::draw(wd.x, wd.y);
        const uint64_t on_m = pcl_ballot(on);
        const bool s0 = on && !(m_first <= pc); // alive after body b                                                              light.py:243
        const uint64_t b0 = ~pcl_ballot(m_first <= pc) & on_m;
        xp = R::add(xp, dp);                                                                                                 // newton.py:16
        T prev = R::sub(xp, dp);
        tally(b, b0, ix, prev, xp);
        bool left = s0;
        d = s0 ? 255u : (uint32_t)(b + 1);
        if (two) { // (uniform)
            const thr_t m_second = D::draw(wd.z, wd.w);
            left = s0 && !(m_second <= pc);
            const uint64_t b1 = ~pcl_ballot(m_second <= pc) & b0;
            if (s0 && !left) d = (uint32_t)(b + 2);
            xp = R::add(xp, dp);
            prev = R::sub(xp, dp);
            tally(b + 1, b1, ix, prev, xp);
        }
        return left;
    };
    // the launch's first pass is the same for every photon: decided where the photon is loaded, before it is listed
    const bool single0 = (a.step0 & 1u) != 0u, two0 = !single0 && a.K > 1;
    const int b_first = two0 ? 2 : 1;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll 1
        for (int bt = 0; bt < kTileRows / (8 * kGroups); ++bt) {
            // ---- the batch's alive photons join the list ----------------------------------------------------------------------
            uint32_t n_list = 0;
            const int row0 = (wave * (kTileRows / 8) + bt * kGroups) * 2;
            const int64_t i0 = tile * kTile + (int64_t)row0 * 64; // the batch's first slot
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                const int row = row0 + 2 * g;
                const int64_t i = i0 + g * 128 + 2 * lane; // the lane's two slots: i, i + 1
                uint64_t m_lo, m_hi;
                if (a.fresh) {
                    const int64_t left = a.slots - (i0 + g * 128);
                    m_lo = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
                    m_hi = left >= 128 ? ~0ull : (left > 64 ? (1ull << (left - 64)) - 1ull : 0ull);
                } else {
                    m_lo = a.masks[tile * kTileRows + row];
                    m_hi = a.masks[tile * kTileRows + row + 1];
                }
                uchar2 d0;
                d0.x = d0.y = 0;
                if ((m_lo | m_hi) != 0ull) { // (wave-uniform)
                    if (two0) ++w_groups2; else ++w_groups1;
                    const uint64_t mm = hi ? m_hi : m_lo;
                    const bool al[2] = {(bool)((mm >> bit) & 1ull), (bool)((mm >> (bit + 1)) & 1ull)};
                    const int64_t ti = pcl_tix(i, a.ts);
                    T vv[3][2], x0[2] = {(T)0, (T)0};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const T2 q = *reinterpret_cast<const T2 *>(a.v[k] + ti);
                        vv[k][0] = q.x, vv[k][1] = q.y;
                    }
                    if (has_plane) {
                        const T2 x = *reinterpret_cast<const T2 *>(a.r[ax0] + ti);
                        x0[0] = x.x, x0[1] = x.y;
                    }
                    uint64_t id[2];
                    if (a.ids) {
                        const longlong2 q = *reinterpret_cast<const longlong2 *>(a.ids + (i < a.slots ? i : 0));
                        id[0] = (uint64_t)q.x, id[1] = (uint64_t)q.y;
                    } else {
                        id[0] = (uint64_t)(a.id_base + i), id[1] = id[0] + 1;
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        T dd[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) dd[k] = R::mul(vv[k][e], a.dt);                                          // newton.py:15
                        const T pcoll = R::mul(a.An, pcl_step_norm<T>(dd[0], dd[1], dd[2]));                                 // light.py:241-247
                        const T va = pcl_pick<T>(ax0, vv[0][e], vv[1][e], vv[2][e]);
                        T x = x0[e];
                        if (has_plane)
                            for (int q = 0; q < a.n_pend; ++q) { // the moves the store still owes its r
                                const T dq = R::mul(va, a.pend_dt[q]);
                                for (int t = 0; t < a.pend_rep[q]; ++t) x = R::add(x, dq);
                            }
                        const uint32_t ix = (uint32_t)(g * 128 + 2 * lane + e) | (vv[0][e] > (T)0 ? 0x100u : 0u) | (vv[1][e] > (T)0 ? 0x200u : 0u) |
                                            (vv[2][e] > (T)0 ? 0x400u : 0u);
                        const thr_t thr = D::threshold(pcoll);
                        const T dpe = pcl_pick<T>(ax0, dd[0], dd[1], dd[2]);
                        uint32_t d;
                        const bool left = decide(0, single0, two0, a.step0, al[e], id[e], thr, x, dpe, ix, d);
                        if (e == 0) d0.x = (unsigned char)(al[e] ? d : 0u); else d0.y = (unsigned char)(al[e] ? d : 0u);
                        const uint64_t bal = pcl_ballot(left);
                        if (left) { // still alive after the first pass: listed for the next ones
                            const uint32_t pos = n_list + (uint32_t)__popcll(bal & below);
                            if constexpr (IDS) s_id[wave][pos] = id[e];
                            s_pc[wave][pos] = thr;
                            s_xp[wave][pos] = x;
                            s_dp[wave][pos] = dpe;
                            s_ix[wave][pos] = (uint16_t)ix;
                        }
                        n_list += (uint32_t)__popcll(bal);
                    }
                }
                *reinterpret_cast<uchar2 *>(&s_death[wave][g * 128 + 2 * lane]) = d0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // ---- the bodies, one Philox block at a time: it decides the steps 2m and 2m + 1 (pcl_draw_rand), so the bodies are
            // taken in such pairs, after a single one when the first step is odd ------------------------------------------------
            for (int b = b_first; b < a.K && n_list > 0;) {
                const uint32_t st = a.step0 + (uint32_t)b;
                const bool single = (st & 1u) != 0u;          // only the block's second half is a body of this launch
                const bool two = !single && b + 1 < a.K;
                uint32_t w = 0;
                for (uint32_t r0 = 0; r0 < n_list; r0 += 64) {
                    if (two) ++w_rounds2; else ++w_rounds1;
                    const uint32_t j = r0 + (uint32_t)lane;   // (< 256: beyond n_list a stale entry is read -- ``on`` gates every effect)
                    const bool on = j < n_list;
                    const thr_t pc = s_pc[wave][j];
                    T xp = s_xp[wave][j];
                    const T dp = s_dp[wave][j];
                    const uint32_t ix = s_ix[wave][j];
                    uint64_t id;
                    if constexpr (IDS)
                        id = s_id[wave][j];
                    else
                        id = (uint64_t)(a.id_base + i0 + (int64_t)(ix & 0xFFu));
                    uint32_t d;
                    const bool left = decide(b, single, two, st, on, id, pc, xp, dp, ix, d);
                    if (on && d != 255u) s_death[wave][ix & 0xFFu] = (uint8_t)d;
                    const uint64_t bal = pcl_ballot(left);
                    if (left) {
                        const uint32_t pos = w + (uint32_t)__popcll(bal & below);
                        if constexpr (IDS) s_id[wave][pos] = id;
                        s_pc[wave][pos] = pc;
                        s_xp[wave][pos] = xp;
                        s_dp[wave][pos] = dp;
                        s_ix[wave][pos] = (uint16_t)ix;
                    }
                    w += (uint32_t)__popcll(bal);
                }
                n_list = w;
                b += two ? 2 : 1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // the batch's death bytes: four consecutive slots per lane (whole tiles exist in the buffer)
            *reinterpret_cast<uint32_t *>(a.death + i0 + 4 * lane) = *reinterpret_cast<const uint32_t *>(&s_death[wave][4 * lane]);
        }
    }
    if (lane == 0) {
        atomicAdd(&s_work[0], w_groups2);
        atomicAdd(&s_work[1], w_groups1);
        atomicAdd(&s_work[2], w_rounds2);
        atomicAdd(&s_work[3], w_rounds1);
    }
    if (threadIdx.x == 0) { // (this workgroup's lifetime so far: its rows are done)
        s_work[4] = (uint32_t)(__builtin_amdgcn_s_memtime() - s_clk[0]);
        s_work[5] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - s_clk[1]);
    }
    if (lane < a.K) { // the wave's sums join the workgroup's
        if (t_kept) atomicAdd(&s_cnt[lane * kAheadRow + 0], t_kept);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (t_s[k]) atomicAdd(&s_cnt[lane * kAheadRow + 1 + k], t_s[k]);
        if (t_p0) atomicAdd(&s_cnt[lane * kAheadRow + 4], t_p0);
    }
    __syncthreads();
    // grid totals as in k_delete_alive: returning agent-scope atomics, the last workgroup reports
    const int nrow = 4 + (a.n_planes > 0 ? a.n_planes : 0);
    unsigned long long seen = 0;
    for (int q = threadIdx.x; q < a.K * kAheadRow; q += kBlock)
        if ((q % kAheadRow) < nrow && s_cnt[q])
            seen += __hip_atomic_fetch_add(&a.acc[1 + q], (unsigned long long)s_cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < kAheadWork && s_work[threadIdx.x])
        seen += __hip_atomic_fetch_add(&a.acc[1 + kAheadMax * kAheadRow + threadIdx.x], (unsigned long long)s_work[threadIdx.x], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(seen) : "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&a.acc[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!s_last) return;
    volatile uint64_t *h = a.host;
    for (int q = threadIdx.x; q < a.K * kAheadRow; q += kBlock)
        h[q] = (q % kAheadRow) < nrow ? __hip_atomic_exchange(&a.acc[1 + q], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    if (threadIdx.x < kAheadWork)
        h[kAheadMax * kAheadRow + 1 + threadIdx.x] = __hip_atomic_exchange(&a.acc[1 + kAheadMax * kAheadRow + threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_exchange(&a.acc[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.host[kAheadMax * kAheadRow], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// the state after ``j`` of the bodies k_delete_ahead worked out, made real: alive bits (masks_out), the alive bits before
// the j-th body (masks_prev: pcl_store_last_delete_flags), alive counts per tile, and -- WRITE_R, small stores -- r with the
// pending moves and the j moves of those bodies applied (newton.py:15-16, one rounded multiply and one rounded add per
// move, in order).  Big stores leave r where it is: the j moves join the list of pending ones (a sweep of r and v is the
// price of a whole body there), and the compaction that is usually due next applies them on the way.
template <typename T, bool WRITE_R>
__global__ void __launch_bounds__(kBlock) k_ahead_commit(ahead_args<T> a) {
    typedef pcl_rt<T> R;
    typedef typename std::conditional<sizeof(T) == 8, double2, float2>::type T2;
    __shared__ uint32_t s_keep[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = (a.slots + kTile - 1) / kTile;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t kept = 0;
#pragma unroll 1
        for (int pp = 0; pp < kTileRows / 8; ++pp) {
            const int row = (wave * (kTileRows / 8) + pp) * 2;
            const int64_t i = tile * kTile + (int64_t)row * 64 + 2 * lane;
            const uchar2 d = *reinterpret_cast<const uchar2 *>(a.death + i);
            const uint32_t j = (uint32_t)a.j;
            const uint64_t b0 = pcl_ballot(d.x > j), b1 = pcl_ballot(d.y > j);
            const uint64_t p0 = pcl_ballot(d.x >= j && d.x != 0), p1 = pcl_ballot(d.y >= j && d.y != 0);
            if (lane == 0) {
                a.masks_out[tile * kTileRows + row] = spread_bits((uint32_t)b0) | (spread_bits((uint32_t)b1) << 1);
                a.masks_out[tile * kTileRows + row + 1] = spread_bits((uint32_t)(b0 >> 32)) | (spread_bits((uint32_t)(b1 >> 32)) << 1);
                a.masks_prev[tile * kTileRows + row] = spread_bits((uint32_t)p0) | (spread_bits((uint32_t)p1) << 1);
                a.masks_prev[tile * kTileRows + row + 1] = spread_bits((uint32_t)(p0 >> 32)) | (spread_bits((uint32_t)(p1 >> 32)) << 1);
            }
            kept += (uint32_t)(__popcll(b0) + __popcll(b1));
            if constexpr (!WRITE_R) continue; // big stores: r stays behind, the host adds the j moves to its list of pending ones
            if ((p0 | p1) == 0ull) continue; // nobody was alive before that body: r of these slots is never read again
            const int64_t ti = pcl_tix(i, a.ts);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T2 v = *reinterpret_cast<const T2 *>(a.v[k] + ti);
                T2 x = *reinterpret_cast<const T2 *>(a.r[k] + ti);
                for (int q = 0; q < a.n_pend; ++q) {
                    const T px = R::mul(v.x, a.pend_dt[q]), py = R::mul(v.y, a.pend_dt[q]);
                    for (int t = 0; t < a.pend_rep[q]; ++t) {
                        x.x = R::add(x.x, px);
                        x.y = R::add(x.y, py);
                    }
                }
                const T dx = R::mul(v.x, a.dt), dy = R::mul(v.y, a.dt);
                for (int q = 0; q < a.j; ++q) {
                    x.x = R::add(x.x, dx);
                    x.y = R::add(x.y, dy);
                }
                *reinterpret_cast<T2 *>(a.r[k] + ti) = x;
            }
        }
        if (lane == 0) s_keep[wave] = kept;
        __syncthreads();
        if (threadIdx.x == 0) a.tile_keep[tile] = (int32_t)(s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3]);
        __syncthreads();
    }
}

// k_ahead_commit for big stores (r stays behind): only the alive bits, the alive bits before the last body and the tile
// counts come out of the death bytes.  A lane takes 16 consecutive slots (one 16-byte load; 1 KiB per wave and load
// instruction -- the two-byte loads of the general kernel made this sweep pure latency: 210 us for 1e8 slots), four lanes
// put their 16-bit pieces together into a mask word, a wave (half a tile) adds its alive count to the tile's.
__global__ void __launch_bounds__(kBlock) k_ahead_masks(const uint8_t *__restrict__ death, int j, uint64_t *__restrict__ masks_out,
                                                        uint64_t *__restrict__ masks_prev, int32_t *__restrict__ tile_keep, int64_t n_groups) {
    // (a workgroup's 256 lanes x 16 slots are two whole tiles -- n_groups is a multiple of a tile's 128 --: each tile's count is
    // WRITTEN by the one workgroup that sees all of it, waves 0-1 the first tile, waves 2-3 the second; nothing to zero first)
    __shared__ uint32_t s_c[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t g0 = (int64_t)blockIdx.x * kBlock; g0 < n_groups; g0 += (int64_t)gridDim.x * kBlock) {
        const int64_t g = g0 + threadIdx.x;
        const bool in = g0 + (threadIdx.x & ~63) < n_groups; // (wave-uniform)
        uint32_t cnt = 0;
        if (in) {
            const uint4 q = *reinterpret_cast<const uint4 *>(death + 16 * g);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t alive = 0, prev = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t d = (w[t >> 2] >> (8 * (t & 3))) & 0xFFu;
                alive |= (d > (uint32_t)j ? 1u : 0u) << t;
                prev |= ((d >= (uint32_t)j && d != 0u) ? 1u : 0u) << t;
            }
            // lanes 4m .. 4m + 3 hold the four 16-bit quarters of mask word m of the wave's 16 words
            const uint64_t a1 = (uint64_t)__shfl_down(alive, 1), a2 = (uint64_t)__shfl_down(alive, 2), a3 = (uint64_t)__shfl_down(alive, 3);
            const uint64_t p1 = (uint64_t)__shfl_down(prev, 1), p2 = (uint64_t)__shfl_down(prev, 2), p3 = (uint64_t)__shfl_down(prev, 3);
            if ((lane & 3) == 0) {
                const int64_t word = g >> 2;
                masks_out[word] = (uint64_t)alive | (a1 << 16) | (a2 << 32) | (a3 << 48);
                masks_prev[word] = (uint64_t)prev | (p1 << 16) | (p2 << 32) | (p3 << 48);
            }
            cnt = (uint32_t)__popc(alive);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt += (uint32_t)__shfl_down(cnt, off);
        }
        if (lane == 0) s_c[wave] = cnt;
        __syncthreads();
        if (threadIdx.x < 2 && g0 + 128 * (int64_t)threadIdx.x < n_groups)
            tile_keep[g0 / (kTile / 16) + threadIdx.x] = (int32_t)(s_c[2 * threadIdx.x] + s_c[2 * threadIdx.x + 1]);
        __syncthreads();
    }
}

// ---- K delete loop bodies in one pass: Newton + delete test K times per photon, ONE mask at the end -------
//   A photon of a delete simulation never changes its velocity, so dr = v*dt, |dr| and pcoll = A*n*|dr| are constants
//   of the photon: per step only r += dr (3 adds), the decision draw and the compare remain, until the photon is
//   removed.  The per-step rows of the measure steps (alive count, sign counts, plane crossings of the survivors,
//   physicl/light.py:385-399, 414-431) are tallied as the steps go; survivors of all K steps get their final r and a
//   mask bit, and the usual scan + compaction runs once.  Identical to K rounds of k_newton_mask / scan /
//   k_compact_count (tests/test_gpu_multi.py).  dr is left implicit (= v*dt, PCL_FUSED_LAZY).
template <typename T>
struct newtonmask_multi_args {
    const T *v[3];
    T *r[3];
    const int64_t *ids;
    const unsigned char *kind;
    uint64_t *masks;
    int32_t *tile_keep;
    uint64_t *cnt; // [K][4 + n_planes]: alive, sign x/y/z, plane crossings -- of the survivors of each step
    int64_t id_base, N;
    int64_t ts;
    T dt, An;
    uint64_t seed;
    uint32_t step;
    int K, n_planes;
    T plane_L[PCL_MAX_PLANES];
    int plane_ax[PCL_MAX_PLANES];
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_newton_mask_multi(newtonmask_multi_args<T> a) {
    typedef pcl_rt<T> R;
    __shared__ uint32_t s_cnt[PCL_MULTI_MAX * (4 + PCL_MAX_PLANES)];
    __shared__ int s_keep[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = a.n_planes > 0 ? a.n_planes : 0, nslots = 4 + np;
    for (int j = threadIdx.x; j < a.K * nslots; j += blockDim.x) s_cnt[j] = 0;
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int64_t tile = blockIdx.x;
    int kept = 0;
    // per-step tallies of this wave, lane k holding step k's (K <= 64): one add per step instead of an LDS atomic
    uint32_t t_alive = 0, t_sx = 0, t_sy = 0, t_sz = 0;
    for (int rr = 0; rr < kTileRows / 4; ++rr) {
        const int row = wave * (kTileRows / 4) + rr;
        const int64_t i = tile * kTile + (int64_t)row * 64 + lane;
        const bool in = i < a.N;
        const int64_t ti = pcl_tix(in ? i : 0, a.ts);
        T rv[3], d[3];
        uint64_t sgn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T vk = a.v[k][ti];
            rv[k] = a.r[k][ti];
            d[k] = R::mul(vk, a.dt);                                  // newton.py:15
            sgn[k] = pcl_ballot(in && vk > (T)0);
        }
        const bool photon = in && (a.kind ? (a.kind[in ? i : 0] != 0) : true);
        const T pcoll = R::mul(a.An, pcl_step_norm<T>(d[0], d[1], d[2])); // light.py:241-244
        const pcl_u64 id = (pcl_u64)(a.ids ? a.ids[in ? i : 0] : a.id_base + i);
        bool alive = in;
        pcl_u32 wodd0 = 0, wodd1 = 0;
        for (int k = 0; k < a.K; ++k) {
            if (!pcl_ballot(alive)) break; // nobody of this row is left: the remaining rows of the counters stay 0
            const pcl_u32 st = a.step + (pcl_u32)k;
            rv[0] = R::add(rv[0], d[0]);                              // newton.py:16
            rv[1] = R::add(rv[1], d[1]);
            rv[2] = R::add(rv[2], d[2]);
            T rand;
            if ((st & 1u) == 0u || k == 0) { // decision block of the step pair (pcl_draw_rand)
                const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), st >> 1, 0u, k0, k1);
                rand = (st & 1u) ? R::uniform(w.z, w.w) : R::uniform(w.x, w.y);
                wodd0 = w.z;
                wodd1 = w.w;
            } else {
                rand = R::uniform(wodd0, wodd1);
            }
            alive = alive && !(photon && (pcoll >= rand));
            const uint64_t m = pcl_ballot(alive);
            uint32_t *c = &s_cnt[k * nslots];
            const bool mine = lane == k;
            t_alive += mine ? (uint32_t)__popcll(m) : 0u;
            t_sx += mine ? (uint32_t)__popcll(m & sgn[0]) : 0u;
            t_sy += mine ? (uint32_t)__popcll(m & sgn[1]) : 0u;
            t_sz += mine ? (uint32_t)__popcll(m & sgn[2]) : 0u;
            for (int p = 0; p < np; ++p) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                const T x = pcl_pick<T>(ax, rv[0], rv[1], rv[2]);
                const T prev = R::sub(x, pcl_pick<T>(ax, d[0], d[1], d[2]));
                const uint32_t nc = (uint32_t)__popcll(pcl_ballot(alive && ((prev <= L && L <= x) || (prev >= L && L >= x))));
                if (lane == 0 && nc) atomicAdd(&c[4 + p], nc);
            }
        }
        if (alive) {
            a.r[0][ti] = rv[0];
            a.r[1][ti] = rv[1];
            a.r[2][ti] = rv[2];
        }
        const uint64_t m = pcl_ballot(alive);
        if (lane == 0) a.masks[tile * kTileRows + row] = m;
        kept += __popcll(m);
    }
    if (lane < a.K) {
        if (t_alive) atomicAdd(&s_cnt[lane * nslots + 0], t_alive);
        if (a.n_planes >= 0) {
            if (t_sx) atomicAdd(&s_cnt[lane * nslots + 1], t_sx);
            if (t_sy) atomicAdd(&s_cnt[lane * nslots + 2], t_sy);
            if (t_sz) atomicAdd(&s_cnt[lane * nslots + 3], t_sz);
        }
    }
    if (lane == 0) s_keep[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    for (int j = threadIdx.x; j < a.K * nslots; j += blockDim.x)
        if (s_cnt[j]) atomicAdd(reinterpret_cast<unsigned long long *>(&a.cnt[j]), (unsigned long long)s_cnt[j]);
}

// ---- EXPERIMENT, opt-in (PCL_MULTI_FORM=p): K delete loop bodies with persistent lanes -------------------------------
//   The lane, not the photon, is the unit of work: a wave owns 512 photons and 64 lanes; whenever a lane's photon has
//   been removed (or has finished its K steps) the lane takes the next unassigned photon of the wave, at the start of
//   the next step PAIR -- every lane then needs exactly one Philox decision block per round, whatever step its photon
//   is at, so rounds stay convergent while every round starts with all lanes busy.  With 30 % removed per step a wave
//   runs ~17 rounds for K = 8 and ~20 for K = 16 where lane == photon costs 32 and 64 (and the LDS ring of
//   k_newton_mask_multi_q 20 and 35).
//   Lanes are at different steps, so the per-step rows cannot be tallied with ballots.  They do not have to be: a photon
//   removed in its step D was among the survivors of steps 0..D-1 and of no later one, so the workgroup keeps a
//   histogram over D (D = K: never removed) of the photon count and the three sign counts (v never changes here) and
//   the rows are its suffix sums; plane crossings are per-step events and go to an LDS counter of their step.  Sums of
//   the same integers: identical rows, masks, tile counts and r to k_newton_mask_multi (tests/test_gpu_multi.py).
template <typename T>
__global__ void __launch_bounds__(kBlock) k_newton_mask_multi_p(newtonmask_multi_args<T> a) {
    typedef pcl_rt<T> R;
    constexpr int kWaves = kBlock / 64, kRowsPerWave = kTileRows / kWaves, kPool = kRowsPerWave * 64;
    __shared__ uint32_t s_hist[(PCL_MULTI_MAX + 1) * 2]; // [D]: count | (#v_x>0) << 16,  (#v_y>0) | (#v_z>0) << 16   (<= 2048 each)
    __shared__ uint32_t s_plane[PCL_MULTI_MAX * PCL_MAX_PLANES];
    __shared__ uint32_t s_mask[kWaves][2 * kRowsPerWave];
    __shared__ int s_keep[kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = a.n_planes > 0 ? a.n_planes : 0, nslots = 4 + np;
    for (int j = threadIdx.x; j < (a.K + 1) * 2; j += blockDim.x) s_hist[j] = 0;
    for (int j = threadIdx.x; j < a.K * np; j += blockDim.x) s_plane[j] = 0;
    if (lane < 2 * kRowsPerWave) s_mask[wave][lane] = 0;
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int64_t tile = blockIdx.x;
    const int64_t wave_base = tile * kTile + (int64_t)wave * kPool;   // first particle of this wave's pool
    const int64_t ti_base = tile * a.ts + (int64_t)wave * kPool;
    const int n_pool = a.N - wave_base >= kPool ? kPool : (a.N > wave_base ? (int)(a.N - wave_base) : 0);
    int next = 0, kept = 0;                                           // wave-uniform
    bool has = false;
    int k = 0, slot = 0;
    uint32_t sgn = 0; // bit 0..2: v_x, v_y, v_z > 0
    T rv[3] = {(T)0, (T)0, (T)0}, d[3] = {(T)0, (T)0, (T)0}, pcoll = (T)-1;
    pcl_u64 id = 0;

    // one loop body for the lane's photon, deciding with ``rand``            newton.py:16, light.py:241-249
    auto body = [&](T rand) {
        rv[0] = R::add(rv[0], d[0]);
        rv[1] = R::add(rv[1], d[1]);
        rv[2] = R::add(rv[2], d[2]);
        const bool dies = pcoll >= rand; // plain Objects carry pcoll = -1
        if (!dies) {
            for (int p = 0; p < np; ++p) { // the measure step sees the survivors of the step (light.py:385-399)
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                const T x = pcl_pick<T>(ax, rv[0], rv[1], rv[2]);
                const T prev = R::sub(x, pcl_pick<T>(ax, d[0], d[1], d[2]));
                if ((prev <= L && L <= x) || (prev >= L && L >= x)) atomicAdd(&s_plane[k * np + p], 1u);
            }
        }
        ++k;
        if (dies || k == a.K) {
            const int D = dies ? k - 1 : a.K;
            atomicAdd(&s_hist[2 * D], 1u | ((sgn & 1u) << 16));
            atomicAdd(&s_hist[2 * D + 1], ((sgn >> 1) & 1u) | ((sgn >> 2) << 16));
            if (!dies) { // through all K steps: r to its slot, its bit into the row's mask word
                const int64_t ti = ti_base + slot;
                a.r[0][ti] = rv[0];
                a.r[1][ti] = rv[1];
                a.r[2][ti] = rv[2];
                atomicOr(&s_mask[wave][slot >> 5], 1u << (slot & 31));
            }
            has = false;
        }
        return !dies && k == a.K;
    };

    while (true) {
        // ---- idle lanes take the next photons of the pool, in order
        const uint64_t fm = pcl_ballot(!has);
        const int want = __popcll(fm), left = n_pool - next;
        const int take = want < left ? want : left;
        if (take > 0) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            if (!has && rank < take) {
                slot = next + rank;
                const int64_t i = wave_base + slot, ti = ti_base + slot;
                sgn = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const T vc = a.v[c][ti];
                    rv[c] = a.r[c][ti];
                    d[c] = R::mul(vc, a.dt);                          // newton.py:15
                    sgn |= vc > (T)0 ? 1u << c : 0u;
                }
                const bool photon = a.kind ? (a.kind[i] != 0) : true;
                pcoll = photon ? R::mul(a.An, pcl_step_norm<T>(d[0], d[1], d[2])) : (T)-1; // light.py:241-244
                id = (pcl_u64)(a.ids ? a.ids[i] : a.id_base + i);
                k = 0;
                has = true;
            }
            next += take;
        }
        if (!pcl_ballot(has)) break;
        // ---- one round: the decision block of the lane's current step pair, then its one or two steps
        const pcl_u32 st = a.step + (pcl_u32)k;
        const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), st >> 1, 0u, k0, k1);
        bool through = false;
        if (has) through = body((st & 1u) ? R::uniform(w.z, w.w) : R::uniform(w.x, w.y));
        if (has && (st & 1u) == 0u) through = body(R::uniform(w.z, w.w)) || through;
        kept += __popcll(pcl_ballot(through));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < kRowsPerWave)
        a.masks[tile * kTileRows + wave * kRowsPerWave + lane] = (uint64_t)s_mask[wave][2 * lane] | ((uint64_t)s_mask[wave][2 * lane + 1] << 32);
    if (lane == 0) s_keep[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    // rows: survivors of step j = everything removed later or never
    if ((int)threadIdx.x < a.K) {
        const int j = threadIdx.x;
        uint32_t n = 0, sx = 0, sy = 0, sz = 0;
        for (int D = j + 1; D <= a.K; ++D) {
            const uint32_t h0 = s_hist[2 * D], h1 = s_hist[2 * D + 1];
            n += h0 & 0xFFFFu;
            sx += h0 >> 16;
            sy += h1 & 0xFFFFu;
            sz += h1 >> 16;
        }
        unsigned long long *c = reinterpret_cast<unsigned long long *>(a.cnt) + (int64_t)j * nslots;
        if (n) atomicAdd(&c[0], (unsigned long long)n);
        if (a.n_planes >= 0) {
            if (sx) atomicAdd(&c[1], (unsigned long long)sx);
            if (sy) atomicAdd(&c[2], (unsigned long long)sy);
            if (sz) atomicAdd(&c[3], (unsigned long long)sz);
        }
        for (int p = 0; p < np; ++p)
            if (s_plane[j * np + p]) atomicAdd(&c[4 + p], (unsigned long long)s_plane[j * np + p]);
    }
}

// ---- the same K loop bodies with the dead photons taken out of the lanes -------------------------------------------
//   A removed photon still occupies its lane in k_newton_mask_multi: with 30 % removed per step two thirds of the Philox
//   blocks of an 8-step launch are computed for photons that are gone.  Here a wave runs only the first step pair (one
//   Philox block: the pair shares it) with lane == photon; survivors go into a wave-private ring in LDS (position, id,
//   home slot, the three sign bits and the kind: 36 B -- d = v*dt and pcoll are formed again from the v rows, which are
//   still in L2), and whenever 64 are waiting they are taken out and run on together, all lanes busy at the start:
//   to the end of the launch (RINGS = 1), or for one more step pair, after which the survivors are parked in a second
//   ring and finished from there (RINGS = 2: pays from K = 12 on, same-box A/B in profiles/r02_experiments).  Per-step
//   tallies are sums, so regrouping cannot change them; a survivor's r goes back to its home slot and its bit into the
//   row's mask word in LDS.  Same masks, tile counts, r and counter rows as k_newton_mask_multi, bit for bit
//   (tests/test_gpu_multi.py run both).  A ring never holds more than 63 + 64 entries, so plain Objects (never
//   removed) need no special case.
template <typename T, int RINGS>
__global__ void __launch_bounds__(kBlock) k_newton_mask_multi_q(newtonmask_multi_args<T> a) {
    typedef pcl_rt<T> R;
    constexpr int kWaves = kBlock / 64, kCap = 128, kRowsPerWave = kTileRows / kWaves;
    // two rings per wave: [0] survivors of the first step pair, [1] survivors of the second.  An entry is r, id and a
    // word of bookkeeping (36 B); d = v * dt and pcoll are formed again from the v rows (L2) when the entry is taken out
    __shared__ T q_f[kWaves][RINGS][3][kCap];
    __shared__ uint64_t q_id[kWaves][RINGS][kCap];
    __shared__ uint32_t q_meta[kWaves][RINGS][kCap]; // home slot within the wave's rows (9 bits) | sign bits << 9 | photon << 12
    __shared__ uint32_t s_mask[kWaves][2 * kRowsPerWave];
    __shared__ int s_keep[kWaves];
    extern __shared__ uint32_t s_cnt[];          // [K][4 + n_planes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = a.n_planes > 0 ? a.n_planes : 0, nslots = 4 + np;
    for (int j = threadIdx.x; j < a.K * nslots; j += blockDim.x) s_cnt[j] = 0;
    if (lane < 2 * kRowsPerWave) s_mask[wave][lane] = 0;
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int64_t tile = blockIdx.x;
    const int64_t wave_base = tile * kTile + (int64_t)wave * kRowsPerWave * 64; // first particle of this wave's rows
    const int first = ((a.step & 1u) ? 1 : 2) < a.K ? ((a.step & 1u) ? 1 : 2) : a.K; // steps run lane == photon: up to the pair boundary
    uint32_t t_alive = 0, t_sx = 0, t_sy = 0, t_sz = 0; // per-step tallies, lane k holding step k's
    int qhead[2] = {0, 0}, qcount[2] = {0, 0};           // wave-uniform
    int kept = 0;
    // ring 0's entries run steps [first, mid), ring 1's [mid, K); short launches do not repay the second parking (measured)
    const int mid = (RINGS == 2 && first + 2 < a.K) ? first + 2 : a.K;

    // steps [k_from, K) of one photon per lane; tallies as in k_newton_mask_multi
    auto run_steps = [&](int k_from, bool &alive, T (&rv)[3], const T (&d)[3], T pcoll, pcl_u64 id, bool photon, uint64_t sx,
                         uint64_t sy, uint64_t sz, int k_to) {
        pcl_u32 wodd0 = 0, wodd1 = 0;
        for (int k = k_from; k < k_to; ++k) {
            if (!pcl_ballot(alive)) break;
            const pcl_u32 st = a.step + (pcl_u32)k;
            rv[0] = R::add(rv[0], d[0]);                              // newton.py:16
            rv[1] = R::add(rv[1], d[1]);
            rv[2] = R::add(rv[2], d[2]);
            T rand;
            if ((st & 1u) == 0u || k == k_from) { // decision block of the step pair (pcl_draw_rand)
                const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), st >> 1, 0u, k0, k1);
                rand = (st & 1u) ? R::uniform(w.z, w.w) : R::uniform(w.x, w.y);
                wodd0 = w.z;
                wodd1 = w.w;
            } else {
                rand = R::uniform(wodd0, wodd1);
            }
            alive = alive && !(photon && (pcoll >= rand));
            const uint64_t m = pcl_ballot(alive);
            const bool mine = lane == k;
            t_alive += mine ? (uint32_t)__popcll(m) : 0u;
            t_sx += mine ? (uint32_t)__popcll(m & sx) : 0u;
            t_sy += mine ? (uint32_t)__popcll(m & sy) : 0u;
            t_sz += mine ? (uint32_t)__popcll(m & sz) : 0u;
            for (int p = 0; p < np; ++p) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                const T x = pcl_pick<T>(ax, rv[0], rv[1], rv[2]);
                const T prev = R::sub(x, pcl_pick<T>(ax, d[0], d[1], d[2]));
                const uint32_t nc = (uint32_t)__popcll(pcl_ballot(alive && ((prev <= L && L <= x) || (prev >= L && L >= x))));
                if (lane == 0 && nc) atomicAdd(&s_cnt[k * nslots + 4 + p], nc);
            }
        }
    };
    // a survivor of all K steps: r to its home slot, its bit into the row's mask word
    auto settle = [&](bool alive, int slot, const T (&rv)[3]) {
        if (alive) {
            const int64_t ti = tile * a.ts + (int64_t)wave * kRowsPerWave * 64 + slot;
            a.r[0][ti] = rv[0];
            a.r[1][ti] = rv[1];
            a.r[2][ti] = rv[2];
            atomicOr(&s_mask[wave][slot >> 5], 1u << (slot & 31));
        }
        kept += __popcll(pcl_ballot(alive));
    };
    // park the lanes' surviving photons in ring ``lv``
    auto park = [&](int lv, bool alive, const T (&rv)[3], pcl_u64 id, uint32_t meta) {
        const uint64_t m = pcl_ballot(alive);
        if (alive) {
            const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const int pos = (qhead[lv] + qcount[lv] + pre) & (kCap - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) q_f[wave][lv][k][pos] = rv[k];
            q_id[wave][lv][pos] = id;
            q_meta[wave][lv][pos] = meta;
        }
        qcount[lv] += __popcll(m);
    };
    // take n waiting photons (n <= 64) out of ring ``lv`` and run their next steps: ring 0's one more step pair (the
    // survivors move on to ring 1), ring 1's all that are left
    auto finish = [&](int lv, int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // wave-private rings: ordering only, no barrier
        __builtin_amdgcn_wave_barrier();
        const bool have = lane < n;
        const int pos = (qhead[lv] + lane) & (kCap - 1);
        const uint32_t meta = have ? q_meta[wave][lv][pos] : 0u;
        const int64_t hti = tile * a.ts + (int64_t)wave * kRowsPerWave * 64 + (int64_t)(meta & 511u);
        T rv[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = R::mul(a.v[k][hti], a.dt);                         // newton.py:15, as in the first pass over the row
            rv[k] = q_f[wave][lv][k][pos];
        }
        // a plain Object: pcoll >= rand never holds (rand >= 0)
        const T pcoll = (meta >> 12 & 1u) ? R::mul(a.An, pcl_step_norm<T>(d[0], d[1], d[2])) : (T)-1; // light.py:241-244
        const pcl_u64 id = q_id[wave][lv][pos];
        bool alive = have;
        const int k_from = lv == 0 ? first : mid, k_to = lv == 0 ? mid : a.K;
        run_steps(k_from, alive, rv, d, pcoll, id, true, pcl_ballot(have && (meta >> 9 & 1u)), pcl_ballot(have && (meta >> 10 & 1u)),
                  pcl_ballot(have && (meta >> 11 & 1u)), k_to);
        qhead[lv] = (qhead[lv] + n) & (kCap - 1);
        qcount[lv] -= n;
        if (RINGS == 1 || k_to == a.K)
            settle(alive, (int)(meta & 511u), rv);
        else
            park(RINGS - 1, alive, rv, id, meta);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };

    // the next row's loads are issued before the current row's steps (a row's steps are a long dependent chain)
    T nv[3], nr[3];
    pcl_u64 nid = 0;
    bool nphoton = false;
    auto fetch = [&](int rr) {
        const int64_t i = wave_base + (int64_t)rr * 64 + lane;
        const bool in = i < a.N;
        const int64_t ti = pcl_tix(in ? i : 0, a.ts);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            nv[k] = a.v[k][ti];
            nr[k] = a.r[k][ti];
        }
        nphoton = in && (a.kind ? (a.kind[in ? i : 0] != 0) : true);
        nid = (pcl_u64)(a.ids ? a.ids[in ? i : 0] : a.id_base + i);
    };
    fetch(0);
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int64_t i = wave_base + (int64_t)rr * 64 + lane;
        const bool in = i < a.N;
        T rv[3], d[3];
        bool sg[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rv[k] = nr[k];
            d[k] = R::mul(nv[k], a.dt);                               // newton.py:15
            sg[k] = in && nv[k] > (T)0;
        }
        const bool photon = nphoton;
        const pcl_u64 id = nid;
        if (rr + 1 < kRowsPerWave) fetch(rr + 1);
        const T pcoll = R::mul(a.An, pcl_step_norm<T>(d[0], d[1], d[2])); // light.py:241-244
        bool alive = in;
        run_steps(0, alive, rv, d, pcoll, id, photon, pcl_ballot(sg[0]), pcl_ballot(sg[1]), pcl_ballot(sg[2]), first);
        const int slot = rr * 64 + lane;
        if (first == a.K) { // (K <= 2: nothing left to run densely)
            settle(alive, slot, rv);
            continue;
        }
        park(0, alive, rv, id,
             (uint32_t)slot | (sg[0] ? 1u << 9 : 0u) | (sg[1] ? 1u << 10 : 0u) | (sg[2] ? 1u << 11 : 0u) | (photon ? 1u << 12 : 0u));
        while (qcount[0] >= 64) {
            finish(0, 64);
            if (RINGS == 2)
                while (qcount[RINGS - 1] >= 64) finish(RINGS - 1, 64);
        }
    }
    if (qcount[0] > 0) finish(0, qcount[0]);
    if (RINGS == 2)
        while (qcount[RINGS - 1] > 0) finish(RINGS - 1, qcount[RINGS - 1] < 64 ? qcount[RINGS - 1] : 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < kRowsPerWave)
        a.masks[tile * kTileRows + wave * kRowsPerWave + lane] = (uint64_t)s_mask[wave][2 * lane] | ((uint64_t)s_mask[wave][2 * lane + 1] << 32);
    if (lane < a.K) {
        if (t_alive) atomicAdd(&s_cnt[lane * nslots + 0], t_alive);
        if (a.n_planes >= 0) {
            if (t_sx) atomicAdd(&s_cnt[lane * nslots + 1], t_sx);
            if (t_sy) atomicAdd(&s_cnt[lane * nslots + 2], t_sy);
            if (t_sz) atomicAdd(&s_cnt[lane * nslots + 3], t_sz);
        }
    }
    if (lane == 0) s_keep[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_keep[tile] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    for (int j = threadIdx.x; j < a.K * nslots; j += blockDim.x)
        if (s_cnt[j]) atomicAdd(reinterpret_cast<unsigned long long *>(&a.cnt[j]), (unsigned long long)s_cnt[j]);
}

// ---- pass 3 with the measure counters folded in: the survivors' r, v (and dr) pass through registers anyway ----
//   field order in compact_args: r0 r1 r2 v0 v1 v2 [dr0 dr1 dr2] [dv0 dv1 dv2 | vprev0..2] E   (compact_fields)
template <typename T>
struct compact_counter_args {
    uint64_t *cnt; // [1..3] sign counts, [4..] plane crossings (slot 0 is the scatter-hit counter)
    T plane_L[PCL_MAX_PLANES];
    int plane_ax[PCL_MAX_PLANES];
    int n_planes;  // -1: no counters
    T dt;          // dr = v*dt when it is implicit
    int move;      // 1: pass 1 only flagged; the survivors' Newton move r = r + v*dt (newton.py:15-16) happens here
    int n_pend;    // alive-mask stores: moves of earlier loop bodies that r has not seen yet, applied first, in order
    T pend_dt[kPendMax];       // (run-length: pend_rep[q] moves of pend_dt[q])
    int pend_rep[kPendMax];
};

template <typename T, typename W> __device__ __forceinline__ T word_as(W w);
template <> __device__ __forceinline__ double word_as<double, uint64_t>(uint64_t w) { return __longlong_as_double((long long)w); }
template <> __device__ __forceinline__ float word_as<float, uint32_t>(uint32_t w) { return __uint_as_float(w); }

template <typename T, typename W, int NF> // NF = 13 (dr travels), 10 (dr implicit) or 7 (dr implicit, dv known to be zero)
__global__ void __launch_bounds__(kBlock) k_compact_count(compact_args a, compact_counter_args<T> c) {
    typedef pcl_rt<T> R;
    constexpr bool HAS_DR = NF == 13;
    if (a.choice && *a.choice != 0) return; // the scan chose k_compact_lds for this launch
    __shared__ uint32_t s_cnt[4 + PCL_MAX_PLANES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4 + PCL_MAX_PLANES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    const uint64_t *tm = a.masks + tile * kTileRows;
    int64_t dest = a.tile_off[tile];
    for (int r = 0; r < wave * (kTileRows / 4); ++r) dest += __popcll(tm[r]);
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t w_s[3] = {0, 0, 0};
    // How many of the wave's 512 slots survive?  With few of them (a store compacted after several bodies: 6 - 12 %) a
    // row-by-row sweep keeps a handful of lanes busy per row -- eight rows, eight rounds of loads with next to nothing in
    // flight.  Then the survivors' slots are first listed in order (two bytes each, the wave's own part of an LDS array: no
    // barrier), and lane j takes the j-th survivor: every lane loads, the stores are consecutive.  Same order, same values.
    __shared__ uint16_t s_list[kBlock / 64][64 * (kTileRows / 4)];
    uint32_t total = 0;
    for (int rr = 0; rr < kTileRows / 4; ++rr) total += (uint32_t)__popcll(tm[wave * (kTileRows / 4) + rr]);
    const bool sparse = total <= a.sparse_max; // (wave-uniform) PCL_COMPACT_SPARSE, default: at most half of the slots
    if (sparse) {
        uint32_t at = 0;
        for (int rr = 0; rr < kTileRows / 4; ++rr) {
            const uint64_t m = tm[wave * (kTileRows / 4) + rr];
            if ((m >> lane) & 1ull) s_list[wave][at + (uint32_t)__popcll(m & below)] = (uint16_t)(rr * 64 + lane);
            at += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    auto item = [&](const bool keep, const int64_t i, const int64_t o) {
        W val[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) val[f] = 0;
        if (keep) {
            const int64_t ti = pcl_tix(i, a.ts), to = pcl_tix(o, a.ts);
#pragma unroll
            for (int f = 0; f < NF; ++f) val[f] = static_cast<const W *>(a.src[f])[ti];
            if (c.move || c.n_pend) { // the survivors' Newton move(s), with the operations of k_newton_mask
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    T rn = word_as<T, W>(val[k]);
                    const T vk = word_as<T, W>(val[3 + k]);
                    for (int q = 0; q < c.n_pend; ++q) {
                        const T dq = R::mul(vk, c.pend_dt[q]);
                        for (int t = 0; t < c.pend_rep[q]; ++t) rn = R::add(rn, dq);
                    }
                    if (c.move) rn = R::add(rn, R::mul(vk, c.dt));
                    __builtin_memcpy(&val[k], &rn, sizeof(W));
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) static_cast<W *>(a.dst[f])[to] = val[f];
            if (a.ids_dst) a.ids_dst[o] = a.ids_src ? a.ids_src[i] : a.id_base + i;
            if (a.kdst) a.kdst[o] = a.ksrc[i];
        }
        if (c.n_planes >= 0) { // wave-uniform
            T rv[3], vv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rv[k] = word_as<T, W>(val[k]);
                vv[k] = word_as<T, W>(val[3 + k]);
                w_s[k] += (uint32_t)__popcll(pcl_ballot(keep && vv[k] > (T)0));
            }
            for (int p = 0; p < c.n_planes; ++p) {
                const int ax = c.plane_ax[p];
                const T L = c.plane_L[p];
                const T x = pcl_pick<T>(ax, rv[0], rv[1], rv[2]);
                T dx;
                if constexpr (HAS_DR)
                    dx = pcl_pick<T>(ax, word_as<T, W>(val[6]), word_as<T, W>(val[7]), word_as<T, W>(val[8]));
                else
                    dx = R::mul(pcl_pick<T>(ax, vv[0], vv[1], vv[2]), c.dt);
                const T prev = R::sub(x, dx);
                const uint32_t np = (uint32_t)__popcll(pcl_ballot(keep && ((prev <= L && L <= x) || (prev >= L && L >= x))));
                if (lane == 0 && np) atomicAdd(&s_cnt[4 + p], np);
            }
        }
    };
    if (sparse) {
        for (uint32_t j = (uint32_t)lane; j < ((total + 63u) & ~63u); j += 64u) {
            const bool keep = j < total;
            const uint32_t loc = keep ? (uint32_t)s_list[wave][j] : 0u;
            item(keep, tile * kTile + (int64_t)wave * (64 * (kTileRows / 4)) + (int64_t)loc, dest + (int64_t)j);
        }
    } else {
#pragma unroll 2
        for (int rr = 0; rr < kTileRows / 4; ++rr) {
            const int row = wave * (kTileRows / 4) + rr;
            const uint64_t m = tm[row];
            item((m >> lane) & 1ull, tile * kTile + (int64_t)row * 64 + lane, dest + __popcll(m & below));
            dest += __popcll(m);
        }
    }
    if (c.n_planes >= 0) {
        if (lane == 0)
            for (int k = 0; k < 3; ++k) atomicAdd(&s_cnt[1 + k], w_s[k]);
        __syncthreads();
        const int nslots = 4 + (c.n_planes > 0 ? c.n_planes : 0);
        if ((int)threadIdx.x >= 1 && (int)threadIdx.x < nslots && s_cnt[threadIdx.x])
            atomicAdd(&c.cnt[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    }
}

// ---- pass 3, LDS-staged: the same stable compaction (+ measure counters) with 16-byte traffic on both sides ----------
//   k_compact_count moves 8 bytes per lane: a wave reads one 64-particle row and writes its survivors, a segment of
//   arbitrary length and alignment, straight to the destination.  Here a lane owns TWO consecutive particles (16-byte
//   loads, a wave covers a 128-particle double row), the workgroup gathers one field of its whole 2048-particle tile into
//   LDS in survivor order, and writes it out as aligned 16-byte groups of the destination row (the first / last group of
//   a tile's output range, which it shares with its neighbours, goes element by element).  Fields go one after the
//   other through two LDS buffers: one barrier per field.  The v rows go first and stay in registers until the r rows
//   arrive: with ``move`` set pass 1 has only flagged (it needs |v * dt|, not r) and the survivors' Newton move happens
//   here, r = r + v * dt, the very operations of k_newton_mask; the sign / plane counters see the same values as in
//   k_compact_count (dr = v * dt is recomputed: it is what pass 1 would have written).
template <typename W> struct w2_of;
template <> struct w2_of<uint64_t> { typedef ulonglong2 type; };
template <> struct w2_of<uint32_t> { typedef uint2 type; };

template <typename T, typename W, int NF>
__global__ void __launch_bounds__(kBlock) k_compact_lds(compact_args a, compact_counter_args<T> c) {
    typedef pcl_rt<T> R;
    typedef typename w2_of<W>::type W2;
    constexpr int VW = 16 / (int)sizeof(W);      // elements per 16-byte store
    constexpr int TRIPS = kTile / 128 / (kBlock / 64); // double rows per wave: 4
    __shared__ uint64_t s_raw[2][kTile];         // two staging buffers (fp32 stores use half of each)
    __shared__ int s_rowoff[kTileRows + 1];
    __shared__ uint32_t s_cnt[4 + PCL_MAX_PLANES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const uint64_t *tm = a.masks + tile * kTileRows;
    if (a.choice && *a.choice != 1) return; // the scan chose k_compact_count for this launch
    if (threadIdx.x < 4 + PCL_MAX_PLANES) s_cnt[threadIdx.x] = 0;
    if (wave == 0) { // survivors before each 64-particle row of the tile
        int inc = lane < kTileRows ? (int)__popcll(tm[lane]) : 0;
        for (int off = 1; off < kTileRows; off <<= 1) {
            const int up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        if (lane < kTileRows) s_rowoff[lane + 1] = inc;
        if (lane == 0) s_rowoff[0] = 0;
    }
    __syncthreads();
    const int kept = s_rowoff[kTileRows];
    const int64_t o0 = a.tile_off[tile];
    // the lane's particle pairs: LDS slots and keep bits
    int pos[TRIPS];
    bool k0[TRIPS], k1[TRIPS];
    int64_t src_pair[TRIPS]; // index of the pair in a tiled row, in 2-element units
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        const int dr = wave * TRIPS + t; // double row of 128 particles
        const uint64_t m_lo = tm[2 * dr], m_hi = tm[2 * dr + 1];
        const bool hi = lane >= 32;
        const uint64_t m = hi ? m_hi : m_lo;
        const int bit = 2 * (lane & 31);
        k0[t] = (m >> bit) & 1ull;
        k1[t] = (m >> (bit + 1)) & 1ull;
        pos[t] = s_rowoff[2 * dr] + (hi ? (int)__popcll(m_lo) : 0) + (int)__popcll(m & ((1ull << bit) - 1ull));
        const int64_t i0 = tile * kTile + (int64_t)dr * 128 + 2 * lane;
        src_pair[t] = pcl_tix(i0, a.ts) >> 1;
    }
    const bool counters = c.n_planes >= 0;
    T vkeep[3][TRIPS][2]; // the v rows go first and wait here for the r rows (Newton move of the survivors, plane crossings)
    uint32_t w_s[3] = {0, 0, 0};
    auto write_out = [&](void *dst_row, const W *buf, bool dense) {
        // aligned groups of VW destination elements; group g covers outputs [VW*g, VW*g + VW)
        const int64_t g0 = o0 / VW, g1 = (o0 + kept + VW - 1) / VW;
        for (int64_t g = g0 + threadIdx.x; g < g1; g += kBlock) {
            const int64_t first = g * VW;
            const int64_t d_el = dense ? first : pcl_tix(first, a.ts);
            W *d = static_cast<W *>(dst_row) + d_el;
            const int j = (int)(first - o0);
            if (j >= 0 && j + VW <= kept) {
                if constexpr (VW == 2) {
                    *reinterpret_cast<ulonglong2 *>(d) = make_ulonglong2(buf[j], buf[j + 1]);
                } else {
                    *reinterpret_cast<uint4 *>(d) = make_uint4(buf[j], buf[j + 1], buf[j + 2], buf[j + 3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    if (j + e >= 0 && j + e < kept) d[e] = buf[j + e];
            }
        }
    };
    // the loads of field j+1 are issued before field j goes through LDS, and the barrier between the fields waits for the
    // LDS traffic only (a __syncthreads would drain the loads in flight as well: loads and stores share vmcnt on gfx9)
    auto field_of = [](int j) { return j < 3 ? j + 3 : (j < 6 ? j - 3 : j); }; // processing order: v0 v1 v2, r0 r1 r2, then the rest
    auto fetch = [&](int j, W2 (&o)[TRIPS]) {
        const W2 *src = static_cast<const W2 *>(a.src[field_of(j)]);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) o[t] = src[src_pair[t]]; // unconditional: a 64-B sector almost always holds a survivor
                                                                  // anyway, and straight-line loads let the waits be counted
    };
    W2 cur[TRIPS], nxt[TRIPS];
    fetch(0, cur);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = field_of(j);
        W *buf = reinterpret_cast<W *>(s_raw[j & 1]);
        if (j + 1 < NF) fetch(j + 1, nxt);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            W2 x = cur[t];
            if (j < 3) { // v rows: sign counts                                        light.py:424-426
                const T v0 = word_as<T, W>(x.x), v1 = word_as<T, W>(x.y);
                vkeep[j][t][0] = v0;
                vkeep[j][t][1] = v1;
                if (counters)
                    w_s[j] += (uint32_t)__popcll(pcl_ballot(k0[t] && v0 > (T)0)) + (uint32_t)__popcll(pcl_ballot(k1[t] && v1 > (T)0));
            } else if (j < 6) { // r rows: the survivors' move when pass 1 only flagged, then the planes on this axis
                const int ax = j - 3;
                T x0 = word_as<T, W>(x.x), x1 = word_as<T, W>(x.y);
                const T d0 = R::mul(vkeep[ax][t][0], c.dt), d1 = R::mul(vkeep[ax][t][1], c.dt); // newton.py:15
                for (int q = 0; q < c.n_pend; ++q) { // moves of earlier loop bodies r has not seen yet (alive-mask stores)
                    const T p0 = R::mul(vkeep[ax][t][0], c.pend_dt[q]), p1 = R::mul(vkeep[ax][t][1], c.pend_dt[q]);
                    for (int w = 0; w < c.pend_rep[q]; ++w) {
                        x0 = R::add(x0, p0);
                        x1 = R::add(x1, p1);
                    }
                }
                if (c.move) {
                    x0 = R::add(x0, d0);                                                              // newton.py:16
                    x1 = R::add(x1, d1);
                }
                if (c.move || c.n_pend) {
                    __builtin_memcpy(&x.x, &x0, sizeof(W));
                    __builtin_memcpy(&x.y, &x1, sizeof(W));
                }
                if (counters)
                    for (int p = 0; p < c.n_planes; ++p) {                                           // light.py:385-399
                        if (c.plane_ax[p] != ax) continue;
                        const T L = c.plane_L[p];
                        const T p0 = R::sub(x0, d0), p1 = R::sub(x1, d1);
                        const uint32_t np = (uint32_t)__popcll(pcl_ballot(k0[t] && ((p0 <= L && L <= x0) || (p0 >= L && L >= x0)))) +
                                            (uint32_t)__popcll(pcl_ballot(k1[t] && ((p1 <= L && L <= x1) || (p1 >= L && L >= x1))));
                        if (lane == 0 && np) atomicAdd(&s_cnt[4 + p], np);
                    }
            }
            if (k0[t]) buf[pos[t]] = x.x;
            if (k1[t]) buf[pos[t] + (k0[t] ? 1 : 0)] = x.y;
        }
        // this field is complete in its buffer (and the write-out of the previous one is behind every thread)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        write_out(a.dst[f], buf, false);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) cur[t] = nxt[t];
    }
    if (a.ids_dst) { // ids: one more "field", always 8 bytes, dense destination
        uint64_t *buf = s_raw[NF & 1];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int64_t i0 = tile * kTile + (int64_t)(wave * TRIPS + t) * 128 + 2 * lane;
            if (k0[t]) buf[pos[t]] = (uint64_t)(a.ids_src ? a.ids_src[i0] : a.id_base + i0);
            if (k1[t]) buf[pos[t] + (k0[t] ? 1 : 0)] = (uint64_t)(a.ids_src ? a.ids_src[i0 + 1] : a.id_base + i0 + 1);
        }
        __syncthreads();
        const int64_t g0 = o0 / 2, g1 = (o0 + kept + 1) / 2;
        for (int64_t g = g0 + threadIdx.x; g < g1; g += kBlock) {
            const int j = (int)(g * 2 - o0);
            uint64_t *d = reinterpret_cast<uint64_t *>(a.ids_dst) + g * 2;
            if (j >= 0 && j + 2 <= kept) {
                *reinterpret_cast<ulonglong2 *>(d) = make_ulonglong2(buf[j], buf[j + 1]);
            } else {
                if (j >= 0 && j < kept) d[0] = buf[j];
                if (j + 1 >= 0 && j + 1 < kept) d[1] = buf[j + 1];
            }
        }
    }
    if (counters) {
        if (lane == 0)
            for (int k = 0; k < 3; ++k) atomicAdd(&s_cnt[1 + k], w_s[k]);
        __syncthreads();
        const int nslots = 4 + (c.n_planes > 0 ? c.n_planes : 0);
        if ((int)threadIdx.x >= 1 && (int)threadIdx.x < nslots && s_cnt[threadIdx.x])
            atomicAdd(&c.cnt[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    }
}

// ---- the delete loop body in ONE pass: Newton + delete flag + stable compaction with a decoupled look-back ----------
//   The three-kernel pipeline reads r and v twice (pass 1 moves and flags, pass 3 moves the survivors) and writes r twice.
//   Here a workgroup takes a UNIT of 1024 consecutive particles (ticket order = particle order), moves them and draws
//   their flags with r and v in registers, learns where its survivors go from the units before it -- every unit
//   publishes its survivor count (AGGREGATE), adds up the counts of the predecessors that have not yet published a
//   running total, and publishes its own (PREFIX): Merrill & Garland's decoupled look-back, one 8-byte status word
//   per unit carrying flag and count together, relaxed agent-scope atomics (the word IS the payload) -- and writes them,
//   field by field through LDS in aligned 16-byte groups (as k_compact_lds).  Per particle: 80 B read + 88 B per
//   survivor written, instead of 72 + 80 + 88 per survivor.  Same arithmetic, same Philox draws, same stable order as
//   k_newton_mask -> k_tile_scan -> k_compact_*: bit-identical state, masks and counters.
//   Progress: a unit only ever waits for units with smaller tickets, which are resident or finished.  Every wait is
//   bounded all the same: a unit that gives up raises ``err`` (all others then stop waiting too) and the host, which has
//   not lost anything -- the source slab is never written -- runs the three-kernel pipeline instead.
constexpr int kUnit = 1024; // particles per look-back unit: 8 double rows of 128, two per wave
constexpr unsigned long long kLbAgg = 1ull << 62, kLbPrefix = 2ull << 62, kLbCount = (1ull << 62) - 1ull;
constexpr int kLbSpinLimit = 400000;

template <typename T, typename W>
struct onepass_args {
    const W *src[10]; // rows of the current slab: r0 r1 r2 v0 v1 v2 x0 x1 x2 (dv, or vprev while dv is implicit) E
    W *dst[10];       // the same rows of the other slab
    const T *rand;    // PCL_RNG_INPUT
    const int64_t *ids_src; // NULL: id = id_base + index
    int64_t *ids_dst;
    uint64_t *masks;             // keep-masks, k_newton_mask layout (pcl_store_last_delete_flags)
    unsigned long long *status;  // [units], zeroed before the launch
    unsigned int *ticket;        // zeroed before the launch
    int *err;                    // zeroed before the launch
    int64_t *total;              // survivors of the whole store
    uint64_t *cnt;               // [1..3] sign counts, [4..] plane crossings of the survivors
    int64_t id_base, N, ts;
    T dt, An;
    uint64_t seed;
    uint32_t step;
    int rng_mode;
    int n_planes; // -1: no counters
    T plane_L[PCL_MAX_PLANES];
    int plane_ax[PCL_MAX_PLANES];
};


template <typename T, typename W>
__global__ void __launch_bounds__(kBlock) k_delete_onepass(onepass_args<T, W> a) {
    typedef pcl_rt<T> R;
    typedef typename w2_of<W>::type W2;
    constexpr int VW = 16 / (int)sizeof(W);
    constexpr int TRIPS = kUnit / 128 / (kBlock / 64); // double rows per wave: 2
    __shared__ uint64_t s_raw[2][kUnit];
    __shared__ int s_drc[kUnit / 128 + 1];
    __shared__ long long s_excl;
    __shared__ unsigned int s_unit;
    __shared__ uint32_t s_cnt[4 + PCL_MAX_PLANES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_unit = atomicAdd(a.ticket, 1u);
    if (threadIdx.x < 4 + PCL_MAX_PLANES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t u = s_unit;
    const int64_t n_units = (a.N + kUnit - 1) / kUnit;
    const pcl_u32 k0s = (pcl_u32)a.seed, k1s = (pcl_u32)(a.seed >> 32);
    const uint64_t below = (1ull << lane) - 1ull;
    T Rn[3][TRIPS][2], V[3][TRIPS][2];
    W2 X[4][TRIPS];
    int64_t ids[TRIPS][2];
    bool keep[TRIPS][2];
    int pre[TRIPS];
    uint32_t w_s[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        const int drow = wave * TRIPS + t;
        const int64_t i0 = u * kUnit + (int64_t)drow * 128 + 2 * lane; // the lane's pair; the slab holds whole tiles, so it exists
        const int64_t pair = pcl_tix(i0, a.ts) >> 1;
        W2 rr[3], vv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rr[k] = reinterpret_cast<const W2 *>(a.src[k])[pair];
            vv[k] = reinterpret_cast<const W2 *>(a.src[3 + k])[pair];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) X[j][t] = reinterpret_cast<const W2 *>(a.src[6 + j])[pair];
        T d[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            V[k][t][0] = word_as<T, W>(vv[k].x);
            V[k][t][1] = word_as<T, W>(vv[k].y);
            d[k][0] = R::mul(V[k][t][0], a.dt);                                  // newton.py:15
            d[k][1] = R::mul(V[k][t][1], a.dt);
            Rn[k][t][0] = R::add(word_as<T, W>(rr[k].x), d[k][0]);              // newton.py:16
            Rn[k][t][1] = R::add(word_as<T, W>(rr[k].y), d[k][1]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t i = i0 + e;
            const bool in = i < a.N;
            ids[t][e] = in ? (a.ids_src ? a.ids_src[i] : a.id_base + i) : 0;
            const T pcoll = R::mul(a.An, pcl_step_norm<T>(d[0][e], d[1][e], d[2][e]));   // light.py:241-244
            T rand;
            if (a.rng_mode == PCL_RNG_PHX)
                rand = pcl_draw_rand<T>((pcl_u64)ids[t][e], a.step, k0s, k1s);
            else
                rand = in ? a.rand[i] : (T)0;
            keep[t][e] = in && !(pcoll >= rand);
        }
        const uint64_t b0 = pcl_ballot(keep[t][0]), b1 = pcl_ballot(keep[t][1]);
        pre[t] = (int)__popcll(b0 & below) + (int)__popcll(b1 & below);
        if (lane == 0) {
            s_drc[drow] = (int)__popcll(b0) + (int)__popcll(b1);
            const int64_t row = (u * kUnit + (int64_t)drow * 128) >> 6; // two 64-particle rows of the mask array
            a.masks[row] = spread_bits((uint32_t)b0) | (spread_bits((uint32_t)b1) << 1);
            a.masks[row + 1] = spread_bits((uint32_t)(b0 >> 32)) | (spread_bits((uint32_t)(b1 >> 32)) << 1);
        }
        if (a.n_planes >= 0) { // the counters of the measure steps, on the survivors      light.py:385-399, 414-431
#pragma unroll
            for (int k = 0; k < 3; ++k)
                w_s[k] += (uint32_t)__popcll(pcl_ballot(keep[t][0] && V[k][t][0] > (T)0)) +
                          (uint32_t)__popcll(pcl_ballot(keep[t][1] && V[k][t][1] > (T)0));
            for (int p = 0; p < a.n_planes; ++p) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                uint32_t np = 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const T x = pcl_pick<T>(ax, Rn[0][t][e], Rn[1][t][e], Rn[2][t][e]);
                    const T prev = R::sub(x, pcl_pick<T>(ax, d[0][e], d[1][e], d[2][e]));
                    np += (uint32_t)__popcll(pcl_ballot(keep[t][e] && ((prev <= L && L <= x) || (prev >= L && L >= x))));
                }
                if (lane == 0 && np) atomicAdd(&s_cnt[4 + p], np);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { // exclusive offsets of the unit's double rows
        int run = 0;
        for (int k = 0; k < kUnit / 128; ++k) {
            const int c = s_drc[k];
            s_drc[k] = run;
            run += c;
        }
        s_drc[kUnit / 128] = run;
    }
    __syncthreads();
    const int kept = s_drc[kUnit / 128];
    if (wave == 0) { // decoupled look-back: where do this unit's survivors go?
        long long excl = 0;
        if (u == 0) {
            if (lane == 0) __hip_atomic_store(&a.status[0], kLbPrefix | (unsigned long long)kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&a.status[u], kLbAgg | (unsigned long long)kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long base = u - 1; // lane i looks at unit base - i
            int spins = 0;
            bool done = false;
            while (!done) {
                const long long idx = base - lane;
                const unsigned long long val = idx >= 0 ? __hip_atomic_load(&a.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                        : kLbPrefix; // before unit 0: running total 0
                const unsigned flag = (unsigned)(val >> 62);
                const uint64_t has_prefix = pcl_ballot(flag == 2u), not_ready = pcl_ballot(flag == 0u);
                const int first = has_prefix ? (int)__ffsll((long long)has_prefix) - 1 : 63;
                const uint64_t window = first >= 63 ? ~0ull : ((2ull << first) - 1ull); // lanes 0..first: the units that count
                if (not_ready & window) {
                    if (++spins > kLbSpinLimit || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        if (lane == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break; // give up: the host discards this launch
                    }
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
                long long part = ((window >> lane) & 1ull) ? (long long)(val & kLbCount) : 0ll;
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                excl += part;
                if (has_prefix)
                    done = true;
                else
                    base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&a.status[u], kLbPrefix | (unsigned long long)(excl + kept), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            if (u == n_units - 1) *a.total = excl + kept;
        }
    }
    __syncthreads();
    const int64_t o0 = s_excl;
    int pos[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) pos[t] = s_drc[wave * TRIPS + t] + pre[t];
    auto write_out = [&](W *dst_row, const W *buf) {
        const int64_t g0 = o0 / VW, g1 = (o0 + kept + VW - 1) / VW;
        for (int64_t g = g0 + threadIdx.x; g < g1; g += kBlock) {
            const int64_t first = g * VW;
            W *d = dst_row + pcl_tix(first, a.ts);
            const int j = (int)(first - o0);
            if (j >= 0 && j + VW <= kept) {
                if constexpr (VW == 2)
                    *reinterpret_cast<ulonglong2 *>(d) = make_ulonglong2(buf[j], buf[j + 1]);
                else
                    *reinterpret_cast<uint4 *>(d) = make_uint4(buf[j], buf[j + 1], buf[j + 2], buf[j + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    if (j + e >= 0 && j + e < kept) d[e] = buf[j + e];
            }
        }
    };
    auto as_word = [](T x) -> W {
        W w;
        __builtin_memcpy(&w, &x, sizeof(W));
        return w;
    };
#pragma unroll
    for (int f = 0; f < 10; ++f) {
        W *buf = reinterpret_cast<W *>(s_raw[f & 1]);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            W x0, x1;
            if (f < 3) {
                x0 = as_word(Rn[f][t][0]);
                x1 = as_word(Rn[f][t][1]);
            } else if (f < 6) {
                x0 = as_word(V[f - 3][t][0]);
                x1 = as_word(V[f - 3][t][1]);
            } else {
                x0 = X[f - 6][t].x;
                x1 = X[f - 6][t].y;
            }
            if (keep[t][0]) buf[pos[t]] = x0;
            if (keep[t][1]) buf[pos[t] + (keep[t][0] ? 1 : 0)] = x1;
        }
        __syncthreads();
        write_out(a.dst[f], buf);
    }
    { // ids: always 8 bytes, dense destination
        uint64_t *buf = s_raw[0]; // field 9 used buffer 1; field 8's write-out (buffer 0) is behind the last barrier
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            if (keep[t][0]) buf[pos[t]] = (uint64_t)ids[t][0];
            if (keep[t][1]) buf[pos[t] + (keep[t][0] ? 1 : 0)] = (uint64_t)ids[t][1];
        }
        __syncthreads();
        const int64_t g0 = o0 / 2, g1 = (o0 + kept + 1) / 2;
        for (int64_t g = g0 + threadIdx.x; g < g1; g += kBlock) {
            const int j = (int)(g * 2 - o0);
            uint64_t *d = reinterpret_cast<uint64_t *>(a.ids_dst) + g * 2;
            if (j >= 0 && j + 2 <= kept) {
                *reinterpret_cast<ulonglong2 *>(d) = make_ulonglong2(buf[j], buf[j + 1]);
            } else {
                if (j >= 0 && j < kept) d[0] = buf[j];
                if (j + 1 >= 0 && j + 1 < kept) d[1] = buf[j + 1];
            }
        }
    }
    if (a.n_planes >= 0) {
        if (lane == 0)
            for (int k = 0; k < 3; ++k) atomicAdd(&s_cnt[1 + k], w_s[k]);
        __syncthreads();
        const int nslots = 4 + (a.n_planes > 0 ? a.n_planes : 0);
        if ((int)threadIdx.x >= 1 && (int)threadIdx.x < nslots && s_cnt[threadIdx.x])
            atomicAdd(&a.cnt[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    }
}

// expand the keep-masks of the last delete back into the reference's int32 ``res`` array
__global__ void __launch_bounds__(kBlock) k_masks_to_flags(const uint64_t *__restrict__ masks,
                                                           int32_t *__restrict__ flags, int64_t N) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
        flags[i] = ((masks[i >> 6] >> (i & 63)) & 1ull) ? 0 : 1;
}

// ---- counters: sign counts + plane crossings, LDS-staged, one atomic per workgroup per counter ---
template <typename T>
struct counter_args {
    const T *v[3], *r[3], *dr[3];
    uint64_t *out; // [3 + n_planes]
    T plane_L[PCL_MAX_PLANES];
    int plane_ax[PCL_MAX_PLANES];
    int n_planes;
    int64_t N;
    int64_t ts;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_counters(counter_args<T> a) {
    typedef pcl_rt<T> R;
    __shared__ uint32_t s_part[kBlock / 64][3 + PCL_MAX_PLANES];
    uint32_t cnt[3 + PCL_MAX_PLANES];
#pragma unroll
    for (int k = 0; k < 3 + PCL_MAX_PLANES; ++k) cnt[k] = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += stride) {
        const int64_t ti = pcl_tix(i, a.ts);
        cnt[0] += a.v[0][ti] > (T)0; // strictly positive: "Do we count 0 as positive? No" light.py:415
        cnt[1] += a.v[1][ti] > (T)0;
        cnt[2] += a.v[2][ti] > (T)0;
#pragma unroll
        for (int p = 0; p < PCL_MAX_PLANES; ++p) {
            if (p < a.n_planes) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p], x = a.r[ax][ti], prev = R::sub(x, a.dr[ax][ti]);
                cnt[3 + p] += ((prev <= L && L <= x) || (prev >= L && L >= x)); // light.py:386
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3 + PCL_MAX_PLANES; ++k) {
        uint32_t x = cnt[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) s_part[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3 + a.n_planes) {
        const int k = threadIdx.x;
        const uint64_t tot = (uint64_t)s_part[0][k] + s_part[1][k] + s_part[2][k] + s_part[3][k];
        if (tot) atomicAdd(&a.out[k], (unsigned long long)tot);
    }
}

// ---- host-drawn randoms of a ScatterIsotropicStep, as the reference draws them: per photon rtheta-, rphi-, rand-uniform
//   (physicl/__init__.py:606-619: three np.random.random() per photon, in that order; light.py:285 scales the first two).
//   The host hands over the raw uniforms U[n][3] exactly as np.random.random((n, 3)) returns them; the split into the three
//   input arrays and the scaling -- (u * 2) * pi and u * pi, the reference's own operations, rounded to the store's
//   precision afterwards -- happen here instead of in three strided numpy passes.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_rand3_split(const double *__restrict__ u3, T *__restrict__ rtheta, T *__restrict__ rphi,
                                                        T *__restrict__ rand, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double u0 = u3[3 * i], u1 = u3[3 * i + 1], u2 = u3[3 * i + 2];
        rtheta[i] = (T)__dmul_rn(__dmul_rn(u0, 2.0), PCL_PI);
        rphi[i] = (T)__dmul_rn(u1, PCL_PI);
        rand[i] = (T)u2;
    }
}

// ---- bulk photon creation --------------------------------------------------------------------------
template <typename T>
struct fill_args {
    T *f[PCL_NFIELDS];
    int64_t n, id_base;
    int64_t ts;
    double c, e_min, e_max;
    uint64_t seed;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_fill_photons(fill_args<T> a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint64_t id = (uint64_t)(a.id_base + i);
        const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), 0xFFFFFFFFu, 2u, (pcl_u32)a.seed,
                                              (pcl_u32)(a.seed >> 32));
        // the energy is always sampled in fp64 and rounded to the store's precision: an fp32 store then
        // holds exactly float(E_fp64), so both precisions describe the same photons
        const double u = pcl_u53(w.x, w.y);
        const double E = __dadd_rn(a.e_min, __dmul_rn(__dsub_rn(a.e_max, a.e_min), pow(u, 1.0 / 3.0)));
#pragma unroll
        for (int f = 0; f < PCL_NFIELDS; ++f) {
            double val = 0.0;
            if (f == PCL_V0) val = a.c;
            if (f == PCL_E) val = E;
            a.f[f][pcl_tix(i, a.ts)] = (T)val;
        }
    }
}

// ---- bulk photon creation from a tabulated energy distribution (Planck sampler) ----------------------
//   physicl/light.py:73-104: draw u, find the bin x with cdf[x-1] <= u <= cdf[x], return grid[x]
//   (u below cdf[0] -- where the reference returns None -- yields grid[0], the lower edge of bin 0)
template <typename T>
struct fill_table_args {
    T *f[PCL_NFIELDS];
    const double *cdf, *grid;
    int nbins;
    int64_t n, id_base;
    int64_t ts;
    double c;
    uint64_t seed;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) k_fill_table(fill_table_args<T> a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint64_t id = (uint64_t)(a.id_base + i);
        const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), 0xFFFFFFFFu, 3u, (pcl_u32)a.seed,
                                              (pcl_u32)(a.seed >> 32));
        const double u = pcl_u53(w.x, w.y);
        int lo = 0, hi = a.nbins - 1; // smallest x with cdf[x] >= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.cdf[mid] >= u) hi = mid; else lo = mid + 1;
        }
        const double E = a.grid[lo];
#pragma unroll
        for (int f = 0; f < PCL_NFIELDS; ++f) {
            double val = 0.0;
            if (f == PCL_V0) val = a.c;
            if (f == PCL_E) val = E;
            a.f[f][pcl_tix(i, a.ts)] = (T)val;
        }
    }
}

// =================================================================================================
// expression validator (variable_n_fn): keeps arbitrary text out of the hipRTC compile and keeps
// every array read at index [gid] so a user expression cannot fault the GPU.  Optionally produces the
// fp32 spelling of the expression (an f suffix on every floating literal).
// =================================================================================================
const char *const kFuncs[] = {"exp", "sqrt", "pow", "log", "log2", "log10", "exp2", "sin",
                              "cos", "tanh", "fabs", "fmin", "fmax", nullptr};
const char *const kArrays[] = {"r0", "r1", "r2", "d0", "d1", "d2", "E", nullptr};

bool in_list(const std::string &s, const char *const *list) {
    for (; *list; ++list)
        if (s == *list) return true;
    return false;
}

int validate_expr(const char *e, std::string *f32_out = nullptr) {
    if (!e) return fail(PCL_ERR_EXPR, "variable_n_fn is NULL");
    const size_t len = strlen(e);
    if (len == 0 || len > 2000) return fail(PCL_ERR_EXPR, "variable_n_fn is empty or longer than 2000 characters");
    size_t i = 0;
    int depth = 0;
    std::string f32;
    auto skip_ws = [&]() {
        while (i < len && (e[i] == ' ' || e[i] == '\t' || e[i] == '\n')) {
            f32 += ' ';
            ++i;
        }
    };
    while (true) {
        skip_ws();
        if (i >= len) break;
        const char ch = e[i];
        if ((ch >= '0' && ch <= '9') || ch == '.') {
            // number: digits [. digits] [e|E [+-] digits]; suffixes are not allowed (the reference is fp64-only)
            size_t j = i;
            bool floating = false;
            while (j < len && ((e[j] >= '0' && e[j] <= '9') || e[j] == '.')) {
                floating = floating || e[j] == '.';
                ++j;
            }
            if (j < len && (e[j] == 'e' || e[j] == 'E')) {
                size_t k = j + 1;
                if (k < len && (e[k] == '+' || e[k] == '-')) ++k;
                if (k >= len || e[k] < '0' || e[k] > '9')
                    return fail(PCL_ERR_EXPR, "variable_n_fn: malformed exponent at offset %zu", j);
                while (k < len && e[k] >= '0' && e[k] <= '9') ++k;
                j = k;
                floating = true;
            }
            if (j < len && ((e[j] >= 'a' && e[j] <= 'z') || (e[j] >= 'A' && e[j] <= 'Z') || e[j] == '_'))
                return fail(PCL_ERR_EXPR, "variable_n_fn: number followed by '%c' at offset %zu", e[j], j);
            f32.append(e + i, j - i);
            if (floating) f32 += 'f';
            i = j;
        } else if ((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_') {
            size_t j = i;
            while (j < len && ((e[j] >= 'a' && e[j] <= 'z') || (e[j] >= 'A' && e[j] <= 'Z') ||
                               (e[j] >= '0' && e[j] <= '9') || e[j] == '_'))
                ++j;
            const std::string id(e + i, j - i);
            i = j;
            f32 += id;
            if (in_list(id, kFuncs)) {
                skip_ws();
                if (i >= len || e[i] != '(')
                    return fail(PCL_ERR_EXPR, "variable_n_fn: function '%s' must be called", id.c_str());
            } else if (in_list(id, kArrays)) {
                // must be exactly  name [ gid ]
                skip_ws();
                if (i >= len || e[i] != '[')
                    return fail(PCL_ERR_EXPR, "variable_n_fn: '%s' must be indexed as %s[gid]", id.c_str(), id.c_str());
                ++i;
                skip_ws();
                if (len - i < 3 || strncmp(e + i, "gid", 3) != 0)
                    return fail(PCL_ERR_EXPR, "variable_n_fn: only the index [gid] is allowed on '%s'", id.c_str());
                i += 3;
                skip_ws();
                if (i >= len || e[i] != ']')
                    return fail(PCL_ERR_EXPR, "variable_n_fn: only the index [gid] is allowed on '%s'", id.c_str());
                ++i;
                f32 += "[gid]";
            } else {
                return fail(PCL_ERR_EXPR, "variable_n_fn: identifier '%s' is not allowed", id.c_str());
            }
        } else if (ch == '(') {
            ++depth;
            f32 += ch;
            ++i;
        } else if (ch == ')') {
            if (--depth < 0) return fail(PCL_ERR_EXPR, "variable_n_fn: unbalanced ')' at offset %zu", i);
            f32 += ch;
            ++i;
        } else if (ch == '+' || ch == '-' || ch == '*' || ch == '/' || ch == ',') {
            f32 += ch;
            ++i;
        } else {
            return fail(PCL_ERR_EXPR, "variable_n_fn: character '%c' at offset %zu is not allowed", ch, i);
        }
    }
    if (depth != 0) return fail(PCL_ERR_EXPR, "variable_n_fn: unbalanced '('");
    if (f32_out) *f32_out = f32;
    return PCL_OK;
}

// =================================================================================================
// context
// =================================================================================================
// A hipRTC compile running beside the simulation: the expression is one of the built-in shapes, so the ahead-of-time
// kernels (same bits, compiled per shape and axis: about as fast) carry the first ~2 s of steps.
struct rtc_job {
    std::thread th;
    std::atomic<int> state{0}; // 0 compiling, 1 code ready, 2 failed
    std::vector<char> code;
    std::string err;
    ~rtc_job() {
        if (th.joinable()) th.join();
    }
};

struct rtc_entry {
    // module == nullptr: hipRTC was not available; the expression matched one of the built-in shapes and runs on the
    // ahead-of-time VAR_N kernels with these parameters (double and float spellings of the user's literals)
    pcl_nprof<double> np64 = {0, 0, 0.0, 0.0, 0.0};
    pcl_nprof<float> np32 = {0, 0, 0.f, 0.f, 0.f};
    hipModule_t module = nullptr;
    hipFunction_t sphere[2] = {nullptr, nullptr};     // [USE_E]           (fp64 only: the reference's ABI)
    hipFunction_t scatter[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // [dtype][USE_E]
    hipFunction_t fused[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t fast[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t multi[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t fastg[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t mixed[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t mixed3[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // three rows per wave and trip, velocities in LDS
    hipFunction_t trace[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // the tracked subset's positions, ahead of a K-pass launch
    hipFunction_t multi2[2] = {nullptr, nullptr}; // two groups per lane and trip (256 photons per wave), fp64: low hit fractions
    hipFunction_t multis[2] = {nullptr, nullptr}; // the 128-photon instantiation (PCL_MULTI_NQ2=0) with the saturation probe (pcl_n_expr_sat), fp64
    hipFunction_t multi2s[2] = {nullptr, nullptr}; // 256 photons per wave with the probe
    hipFunction_t multi3[2] = {nullptr, nullptr}, multi3s[2] = {nullptr, nullptr}; // 192 photons per wave (three per lane), without / with the probe
    std::shared_ptr<struct rtc_job> job;          // a specialisation still compiling in the background (get_rtc)
};

} // namespace

struct pcl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipDeviceProp_t prop;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_count = nullptr; // recorded behind the D2H copy of a compaction's survivor count (scan_tiles)
    int grid_cap = 16384; // workgroups for grid-stride kernels: 64 per CU

    // particle store.  Every array holds elements of the store's dtype (8 or 4 bytes).
    int dtype = PCL_DTYPE_F64;
    size_t esz = 8;
    int64_t capacity = 0, count = 0, id_base = 0;
    bool ids_iota = true;
    // Tiled slab [tile][row][kTileT]: rows 0..12 = the 13 fields, 13..15 = the second half of the v double buffer,
    // 16 = the lam4 cache.  row[] says which physical row currently plays which role (the v buffers swap roles
    // after every lazy fused step); field[f] / vprev[k] / lam4 are the addresses of those rows in tile 0.
    void *slab = nullptr, *slab_alt = nullptr; // slab_alt: compaction destination (lazy)
    int row[kRows];
    int64_t tiles = 0;
    void *field[PCL_NFIELDS] = {};
    void *field_alt[PCL_NFIELDS] = {};
    int64_t *ids = nullptr, *ids_alt = nullptr;
    unsigned char *kind = nullptr, *kind_alt = nullptr;
    // lazy fused steps: second half of the v double buffer + what is still implicit
    void *vprev[3] = {nullptr, nullptr, nullptr};
    void *vprev_alt[3] = {nullptr, nullptr, nullptr}; // the same rows of the compaction slab
    bool lazy_dr = false; // dr not materialised: dr = (lazy_dr_vprev ? vprev : v) * lazy_dt
    bool lazy_dr_vprev = false; // the last Newton move used the velocities now in the vprev rows (a scatter followed it)
    bool lazy_dv = false; // dv not materialised: dv = v - vprev (photons)
    double lazy_dt = 0.0;
    // cache of pow((h*c)/E, -4) per photon for the fast fused path
    void *lam4 = nullptr;
    bool lam4_valid = false;
    double lam4_h = 0.0, lam4_c = 0.0;
    void *rnd[3] = {nullptr, nullptr, nullptr};
    int64_t rnd_n[3] = {0, 0, 0};
    // pcl_store_upload_rand3: two pinned host slots + two device slots of kRand3Chunk photons, used alternately so that
    // the copy of one chunk runs while the caller draws the next
    double *r3_host[2] = {nullptr, nullptr};
    double *r3_dev[2] = {nullptr, nullptr};
    hipEvent_t r3_ev[2] = {nullptr, nullptr};
    int r3_slot = 0;

    // compaction scratch (sized for scratch_cap particles)
    int64_t scratch_cap = 0;
    uint64_t *masks = nullptr;
    int32_t *tile_keep = nullptr;
    int64_t *tile_off = nullptr;
    int64_t last_delete_n = -1;
    // alive-mask state of the one-launch-per-body delete path (k_delete_alive): while ``holes`` the store's extent is
    // ``slots`` (alive + dead), ``count`` is the alive count, ``masks`` holds the alive bits and ``tile_keep`` the alive
    // count per tile; r lags pend_n moves behind.  Every entry point but pcl_step_fused_delete makes the store dense
    // again first (densify, through need_store / ensure_scratch).
    int64_t multi_work[4] = {0, 0, 0, 0}; // last pcl_step_fused_multi launch: dense passes, wave-steps, photons per wave, wave-steps that took
                                          // the saturation shortcut (-1: the launch did not probe) (pcl_store_last_multi_work)
    int multi_hist_at = -1;          // where the last launch's hit histogram lies in h_multi (debug builds; -1: none)
    double multi_clock_ghz = 0.0;    // the clock the chip held under the last pcl_step_fused_multi launch (pcl_store_last_multi_clock)
    double ahead_clock_cycles = 0.0, ahead_clock_ticks = 0.0; // k_delete_ahead_live launches of this context: shader cycles / 100 MHz ticks
    int64_t multi_launches = 0;      // pcl_step_fused_multi launches on this population
    bool multi_sat_on = false, multi_sat_used = false; // the probing variant paid on its last launch / the current launch uses it
    int multi_sat_next = 0;          // launches until the probing variant is tried again
    int mixed_rows = 0;              // rows of 64 particles per wave and trip in the last pcl_step_mixed_multi launch (2: k_mixed, 3: k_mixed3)
    double mixed_last_h = -1.0;      // hit fraction of the last scatter phase of the previous pcl_step_mixed_multi launch (-1: unknown)
    double multi_last_h = -1.0;      // hit fraction of the last step of the previous pcl_step_fused_multi launch (-1: unknown)
    bool holes = false;
    bool seg_prefix = false;         // holes, and the alive bits of every 512-slot segment are a prefix of it (an inplace launch of the mixed
                                     // K-pass kernels left the store so: their next launch takes it as it is, pcl_step_mixed_multi)
    uint64_t alive_seq = 0;          // launches of k_delete_alive that reported to the host (h_cnt[kCounterSlots - 6])
    int64_t slots = 0;
    int pend_n = 0;                  // runs of equal moves r has not seen yet: pend_rep[q] moves of pend_dt[q], oldest first
    double pend_dt[kPendMax] = {0, 0, 0, 0, 0, 0, 0, 0};
    int pend_rep[kPendMax] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t *masks_prev = nullptr;  // alive bits before the last body of the alive path (last_delete_flags)
    bool last_delete_masked = false; // the last delete was such a body: flags come from masks_prev / masks
    int64_t last_delete_slots = 0;
    // delete loop bodies worked out ahead of their calls (k_delete_ahead): while ``ahead.active`` the arrays and the
    // holes / slots / pend_* fields above describe the store BEFORE the first of those bodies, ``count`` follows the
    // bodies handed out; ahead_commit makes that state real.  ``ahead_last``: the previous call (pattern detection).
    struct ahead_state {
        bool active = false;
        int K = 0, used = 0, n_planes = -1;
        double dt = 0, A = 0, n = 0, planes[3 * PCL_MAX_PLANES] = {};
        uint64_t seed = 0;
        uint32_t step0 = 0;
        int64_t before0 = 0, slots = 0;
        bool fresh = false;
        bool big = false;  // extent above PCL_AHEAD_MAX_SLOTS: few bodies per launch, r is not rewritten at the commit
    } ahead, ahead_last;
    bool ahead_last_valid = false;
    uint8_t *ahead_death = nullptr;          // one byte per slot
    int64_t ahead_cap = 0;
    unsigned long long *ahead_acc = nullptr; // device accumulators of k_delete_ahead
    uint64_t *ahead_host = nullptr;          // pinned rows + sequence word
    uint64_t ahead_seq = 0;
    int64_t ahead_launches = 0, ahead_served = 0, ahead_missed = 0; // statistics (pcl_store_ahead_stats)
    int64_t ahead_work[kAheadWork] = {0, 0, 0, 0, 0, 0}; // k_delete_ahead_live's own tally, summed over its launches (pcl_store_ahead_work)
    int ahead_wait = 0, ahead_backoff = 0;   // bodies to run the plain way before the next attempt / the last such pause (doubles per
                                             // launch that was cut short: a loop that looks at the store between its bodies)
    unsigned long long *lb_status = nullptr; // decoupled look-back words of k_delete_onepass, one per 1024-particle unit
    int64_t lb_units = 0;
    // Rows that are known to hold nothing but +0.0 need not travel through a compaction: in a run that never scatters
    // (delete-until-empty, BASELINE configs[1](ii)) that is dv, 24 of the 88 bytes a survivor costs.
    int dv_zero = 0;              // 0 unknown (checked on the device when it matters), 1 every live dv element is +0.0, 2 not
    int64_t dv_zero_n = 0;        // leading elements of THIS slab's dv rows known to be +0.0 (>= count while dv_zero == 1; a
                                  // high-water mark: whatever was zero beyond the live range stays zero until somebody writes dv)
    int64_t alt_dv_zero_n = 0;    // leading elements of the OTHER slab's dv rows known to be +0.0
    int64_t compact_known_alive = -1; // >= 0: the caller of the compaction knows the survivor count (densify): the host picks the pass-3 kernel
    int compact_dv_mode = 0;      // how the last compaction launch treated dv (kDvMove / kDvVprev / kDvSkip), for adopt_compacted

    // counters: device slots + pinned host mirror
    uint64_t *d_cnt = nullptr;
    uint64_t *h_cnt = nullptr;
    bool hits_on_host = true; // h_cnt[0] holds the hit count of the most recent scatter step
    // two counter banks for pcl_step_fused calls that do not synchronise: step k+1 is enqueued before the host
    // reads step k's counters, so the GPU never waits for Python (pcl_step_fused_read drains them in order)
    uint64_t *d_multi = nullptr, *h_multi = nullptr; // kMultiSlots counters of a K-step pass
    void *e_out = nullptr;                           // dense gather buffer of pcl_step_plane_energies
    int64_t e_out_cap = 0;
    // pcl_store_trace_ahead: the tracked ids (host copy + device), their slots, the rows
    std::vector<int64_t> trace_ids;
    int64_t *trace_want = nullptr, *trace_slot = nullptr;
    double *trace_out = nullptr;                     // PINNED HOST memory: the kernel writes its rows there, nothing is copied back
    int64_t trace_want_cap = 0, trace_out_cap = 0, trace_out_n = 0; // (trace_out_n: doubles of the rows a launch has left to be read)
    uint64_t *d_bank[2] = {nullptr, nullptr};
    uint64_t *h_bank[2] = {nullptr, nullptr};
    hipEvent_t bank_ev[2] = {nullptr, nullptr};
    int64_t bank_count[2] = {0, 0};
    int bank_np[2] = {0, 0};
    int bank_head = 0, bank_pending = 0; // FIFO of un-read async steps: banks head, head^1
    int last_async_bank = -1;            // bank of the most recent step if it was asynchronous, else -1
    uint64_t *cnt_target = nullptr;      // where the running step's kernels accumulate

    std::map<std::string, rtc_entry> rtc;
    bool rtc_background = false; // built-in expression shapes start on the ahead-of-time kernels while hipRTC compiles
    double slab_rates[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // alloc_slab: sweep rates (GB/s) of the candidates of the store's slab
    int slab_tries = 0;
    double slab_chosen = 0.0;

    // per-kernel HIP-event timing (pcl_prof_*): pairs recorded immediately around each launch
    struct prof_slot {
        hipEvent_t a, b;
        int kid;
    };
    std::vector<prof_slot> prof;
    size_t prof_used = 0;
    bool prof_on = false;
};

namespace {

template <typename T> T *F(pcl_ctx *c, int f) { return static_cast<T *>(c->field[f]); }

// what is known about the live dv rows (pcl_ctx::dv_zero).  1 = "[0, n) was just written with / found to hold +0.0";
// 0 (unknown) and 2 (not zero) forget the slab's high-water mark: somebody wrote dv.
void set_dv_zero(pcl_ctx *c, int state, int64_t n = 0) {
    c->dv_zero = state;
    c->dv_zero_n = state == 1 ? (c->dv_zero_n > n ? c->dv_zero_n : n) : 0;
}

// bytes between the rows of a tile beyond the row's own kTileT elements (EXPERIMENT hook PCL_ROW_PAD, multiple of 16)
inline size_t row_pad_bytes() {
    static const size_t pad = [] {
        const char *e = getenv("PCL_ROW_PAD");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)(v / 16 * 16) : (size_t)0;
    }();
    return pad;
}
inline size_t row_pitch_bytes(const pcl_ctx *c) { return (size_t)kTileT * c->esz + row_pad_bytes(); }
inline size_t slab_bytes(const pcl_ctx *c) { return (size_t)c->tiles * kRows * row_pitch_bytes(c); }
inline int64_t tile_stride(const pcl_ctx *c) { return (int64_t)(kRows * row_pitch_bytes(c) / c->esz); } // elements from tile to tile

// recompute the tile-0 row addresses after the slab pointers or the row roles changed
void refresh_rows(pcl_ctx *c) {
    const size_t rowb = row_pitch_bytes(c);
    for (int f = 0; f < PCL_NFIELDS; ++f) {
        c->field[f] = c->slab ? static_cast<char *>(c->slab) + rowb * c->row[f] : nullptr;
        c->field_alt[f] = c->slab_alt ? static_cast<char *>(c->slab_alt) + rowb * c->row[f] : nullptr;
    }
    for (int k = 0; k < 3; ++k) {
        c->vprev[k] = c->slab ? static_cast<char *>(c->slab) + rowb * c->row[kRowVprev + k] : nullptr;
        c->vprev_alt[k] = c->slab_alt ? static_cast<char *>(c->slab_alt) + rowb * c->row[kRowVprev + k] : nullptr;
    }
    c->lam4 = c->slab ? static_cast<char *>(c->slab) + rowb * c->row[kRowLam4] : nullptr;
}

// element i of a row lives at (i / T) * tile_stride + i % T
inline int64_t tix_host(const pcl_ctx *c, int64_t i) { return (i / kTileT) * tile_stride(c) + (i % kTileT); }

int64_t rows_in_handle(const void *d, size_t pitch, size_t rowbytes, int64_t k); // below, with the allocator

// copy n elements [offset, offset+n) of one row between a dense host array and the tiled slab
int copy_row(pcl_ctx *ctx, void *row0, void *host, int64_t offset, int64_t n, bool to_device) {
    char *h = static_cast<char *>(host);
    const size_t esz = ctx->esz;
    int64_t i = offset;
    while (n > 0) {
        const int64_t lo = i % kTileT;
        char *d = static_cast<char *>(row0) + (size_t)tix_host(ctx, i) * esz;
        const size_t tile_pitch = (size_t)tile_stride(ctx) * esz;
        const int64_t k = (lo == 0 && n >= kTileT) ? rows_in_handle(d, tile_pitch, (size_t)kTileT * esz, n / kTileT) : 0;
        if (k > 0) { // whole tiles: one strided copy (per physical handle of the slab)
            PCL_HIP(hipMemcpy2DAsync(to_device ? (void *)d : (void *)h, to_device ? tile_pitch : (size_t)kTileT * esz,
                                     to_device ? (const void *)h : (const void *)d, to_device ? (size_t)kTileT * esz : tile_pitch,
                                     (size_t)kTileT * esz, (size_t)k, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                                     ctx->stream));
            i += k * kTileT;
            h += (size_t)k * kTileT * esz;
            n -= k * kTileT;
        } else {
            const int64_t len = (kTileT - lo) < n ? (kTileT - lo) : n;
            PCL_HIP(hipMemcpyAsync(to_device ? (void *)d : (void *)h, to_device ? (const void *)h : (const void *)d, (size_t)len * esz,
                                   to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
            i += len;
            h += (size_t)len * esz;
            n -= len;
        }
    }
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

// run fn<double> or fn<float> according to the store's dtype
#define PCL_DISPATCH(ctx, call_f64, call_f32) ((ctx)->dtype == PCL_DTYPE_F64 ? (call_f64) : (call_f32))

int bind(pcl_ctx *ctx) {
    if (!ctx) return fail(PCL_ERR_ARG, "ctx is NULL");
    PCL_HIP(hipSetDevice(ctx->device));
    return PCL_OK;
}

int launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PCL_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PCL_OK;
}

// Workgroups of a grid-stride kernel.  Above the cap (64 per CU) every workgroup takes the SAME number of trips
// (k = ceil(blocks / cap), grid = ceil(blocks / k)): a grid of exactly ``cap`` leaves a second, nearly empty round when the
// work is 1.0-1.3 caps (1e7 photons: 19532 blocks of work on 16384 workgroups -- the one-step kernel ran at 0.49 of peak
// there against 0.74 at 1e8).  PCL_GRID_UNBALANCED=1: the plain cap (A/B).
int grid_for(const pcl_ctx *ctx, int64_t items, int per_block) {
    static const bool plain = getenv("PCL_GRID_UNBALANCED") != nullptr;
    int64_t g = div_up(items, per_block);
    if (g > ctx->grid_cap) g = plain ? (int64_t)ctx->grid_cap : div_up(g, div_up(g, ctx->grid_cap));
    if (g < 1) g = 1;
    return (int)g;
}

// Every particle array is padded to a multiple of 64 elements: the 16-byte-per-lane kernels load and
// store whole groups, so the elements after N in the last group must exist (their contents are never used).
// ---- big device blocks are kept for the next store of this process ----------------------------------------------
//   A simulation script creates and drops stores of the same size over and over (one Simulation per parameter set, the
//   reference's notebooks).  On this runtime a hipMalloc of tens of GB that follows the hipFree of a block of that size
//   now and then takes 2-4 s (measured: 13.6 GB slab, about one run in four; the first allocation of a process takes
//   0.3 ms) -- the driver is still taking the freed pages apart.  Blocks of >= 64 MB therefore go to a small per-process
//   pool instead of back to the driver and the next request of about that size takes one from there.  PCL_POOL_GB
//   (default: a third of the device's memory; 0 = off) bounds what the pool may hold; an out-of-memory hipMalloc empties it and tries again.
struct pool_block {
    void *p;
    size_t bytes;
    int device;
};
struct big_block {
    size_t bytes;
    int device; // the device the block was allocated on (the caller's current device may differ when it is freed)
};
std::mutex g_pool_mu;
std::vector<pool_block> g_pool;               // idle blocks, oldest first
std::unordered_map<void *, big_block> g_big;  // live blocks that may go to the pool when freed
size_t g_pool_bytes = 0;
// blocks of at least this size are built with the virtual-memory API and pooled (64 MB; PCL_BIG_MIN_MB lowers it so that
// tests reach those paths with small stores)
size_t big_min_bytes() {
    static const size_t b = [] {
        const char *e = getenv("PCL_BIG_MIN_MB");
        const double mb = e ? atof(e) : 64.0;
        return (size_t)((mb > 0 ? mb : 64.0) * (double)((size_t)1 << 20));
    }();
    return b;
}
#define kPoolMinBlock big_min_bytes()

// ---- where big blocks come from: the virtual-memory API, not hipMalloc ----------------------------------------------
//   The store's passes run 13 to 20 streams side by side inside every 272 KB tile.  How fast that goes depends on the
//   physical memory behind the slab, not on the kernel: the same kernel at the same virtual address takes 1.87 or
//   2.07 ms per step (0.695 or 0.63 of the HBM peak) from one hipMalloc to the next, fixed for the life of the
//   allocation, about half of the allocations each way (tools/alloc_variance.py, tools/alloc_candidates.py);
//   hipExtMallocWithFlags(hipDeviceMallocContiguous) always gives the slow kind; row padding does not change it
//   (tools/pad_sweep.py), nor does the order the tiles are visited in (tools/attic/spread_probe.hip); ranges mapped from
//   ~1 GB hipMemCreate handles came out at least as fast as the best hipMalloc blocks (tools/attic/vmm_probe.hip) and are
//   what alloc_slab's candidates are made of.  Blocks of >= 64 MB are built that way; PCL_VMM=0 goes back to hipMalloc
//   and so does any failure of the virtual-memory calls.  hipMemcpy2DAsync refuses rows that reach from one handle into
//   the next (tools/attic/vmm_copy_test2.hip): handles hold whole tiles and copy_row cuts its strided copies at their ends.
//   A virtual address range is mapped ONCE (round 5).  Measured on this runtime (tools/attic/debug_shard.py,
//   profiles/r05_vmm_remap.log): after hipMemUnmap + hipMemAddressFree of a 109 GB range, hipMemAddressReserve hands the
//   same addresses out again, and the first kernels that go through the new hipMemMap'ing there still hit translations of
//   the old one -- of the 1e8 photons a fill kernel wrote, 10 % to 76 % were not in the new range afterwards (zeros where
//   the candidate sweep or the driver had left zeros), with fresh handles and with re-used ones alike, whether or not the
//   process slept in between; the store's second fill was intact, and so was everything with hipMalloc'ed blocks
//   (PCL_VMM=0).  With the freed range's addresses out of circulation -- so that a new range never lies where an old one
//   was -- the fill is intact (PCL_VMM_KEEP_VA=0 brings the fault back).  The runtime only lets go of a range's physical
//   memory at hipMemAddressFree (keeping the reservation across the unmap kept 54 GB "in use" after every handle had been
//   released, tools/attic/debug_trim.py), so raw_free frees the addresses and reserves the very same ones again at once,
//   unmapped for good.  Address space is plentiful (a 13.6 GB store per range, 2^47 bytes to spend) and most stores
//   never get here: they come out of the pool still mapped.
//   The physical handles of an unmapped range wait in g_vmm_idle (per device, by length) and the next range is mapped
//   from them before new ones are created; pcl_pool_trim, PCL_POOL_GB=0 or an allocation that runs out of memory hand
//   them back to the driver.
struct vmm_block {
    size_t bytes;
    size_t chunk; // bytes per physical handle (the last one may be shorter)
    int device;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
struct vmm_idle_handle {
    hipMemGenericAllocationHandle_t h;
    size_t len;
    int device;
};
std::unordered_map<void *, vmm_block> g_vmm; // guarded by g_vmm_mu
std::vector<vmm_idle_handle> g_vmm_idle;     // physical handles of freed ranges, kept for the next ones; guarded by g_vmm_mu
size_t g_vmm_idle_bytes = 0;
std::mutex g_vmm_mu;
// Address ranges that WERE mapped once (guarded by g_vmm_mu).  ``parked``: freed and reserved again at once, never to be mapped --
// the runtime cannot hand those addresses out a second time.  ``tainted``: the second reservation failed or came back
// somewhere else, so the addresses are back in circulation: every new range is checked against these (vmm_malloc) and a
// reservation that overlaps one is parked too.  The address space is 2^47 bytes; the parked total is bounded at a quarter of
// it (kParkedVaMax: ~300 stores of 8e8 photons), beyond which big blocks come from hipMalloc for the rest of the process.
struct va_range {
    uintptr_t base;
    size_t bytes;
};
std::vector<va_range> g_vmm_tainted;
size_t g_vmm_parked_bytes = 0, g_vmm_remaps_avoided = 0;
std::atomic<bool> g_vmm_off{false}; // the virtual-memory path is switched off for the rest of the process (with the reason on stderr)
constexpr size_t kParkedVaMax = (size_t)32 << 40;
std::unordered_map<void *, double> g_rate; // whole-block sweep rate (GB/s) of live and idle blocks (alloc_slab); guarded by g_vmm_mu

bool vmm_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCL_VMM");
        return !(e && e[0] == '0');
    }();
    return on && !g_vmm_off.load(std::memory_order_relaxed);
}

void vmm_switch_off(const char *why) {
    if (!g_vmm_off.exchange(true)) fprintf(stderr, "libphysicl_hip: big blocks come from hipMalloc from here on: %s\n", why);
}

// A range that was mapped (in whole or in part) is unmapped for good: its addresses are freed -- the runtime lets go of the
// physical memory only there -- and reserved again at once, never to be mapped (profiles/r05_vmm_remap.log: a new mapping at
// addresses that were mapped before loses writes; a device-wide synchronise before the unmap and seconds of sleep did not
// change that, so it is not work in flight but translations of the old mapping that the runtime leaves behind).  If the
// second reservation cannot be had, the range is remembered as tainted and every later range is checked against it.
void vmm_park(void *p, size_t bytes) {
    (void)hipMemAddressFree(p, bytes);
    static const bool park = [] { const char *e = getenv("PCL_VMM_KEEP_VA"); return !(e && e[0] == '0'); }();
    static const bool test_fail = getenv("PCL_VMM_TEST_PARK_FAIL") != nullptr; // tests: behave as if the second reservation had failed
    bool parked = false;
    if (park && !test_fail) {
        void *again = nullptr;
        if (hipMemAddressReserve(&again, bytes, (size_t)2 << 20, p, 0) != hipSuccess) {
            (void)hipGetLastError();
        } else if (again != p) { // (somebody else's addresses: not ours to hold)
            (void)hipMemAddressFree(again, bytes);
        } else {
            parked = true;
        }
    }
    if (!park) return; // (PCL_VMM_KEEP_VA=0: the round-5 fault on purpose, for A/B)
    bool too_much = false;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        if (parked)
            g_vmm_parked_bytes += bytes;
        else
            g_vmm_tainted.push_back({reinterpret_cast<uintptr_t>(p), bytes});
        too_much = g_vmm_parked_bytes > kParkedVaMax;
    }
    if (too_much) vmm_switch_off("a quarter of the address space is parked behind freed ranges");
}

// the store's own access pattern, as a probe: every workgroup writes the 13 field rows of a [17][2048] fp64 tile
__global__ void __launch_bounds__(256) k_slab_sweep(double2 *slab, int64_t tiles, int64_t tile_pitch16, int64_t row_pitch16) {
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        double2 *tile = slab + t * tile_pitch16;
#pragma unroll
        for (int row = 0; row < PCL_NFIELDS; ++row)
            for (int q = threadIdx.x; q < (int)(kTileT * 8 / 16); q += 256) tile[row * row_pitch16 + q] = make_double2(0.0, 0.0);
    }
}

// GB/s of that sweep over ``tiles`` tiles at p, on ``stream`` (the context's own: the NULL stream would order the sweep
// against every blocking stream of the host application); synchronous; 0 on failure
double sweep_rate_raw(void *p, int64_t tiles, hipStream_t stream) {
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0.0;
    const int grid = (int)(tiles < 2048 ? tiles : 2048);
    const int64_t tile_pitch16 = (int64_t)kRows * kTileT * 8 / 16, row_pitch16 = kTileT * 8 / 16;
    double rate = 0.0;
    hipLaunchKernelGGL(k_slab_sweep, dim3(grid), dim3(256), 0, stream, static_cast<double2 *>(p), tiles, tile_pitch16, row_pitch16); // first touch
    (void)hipEventRecord(a, stream);
    for (int k = 0; k < 2; ++k)
        hipLaunchKernelGGL(k_slab_sweep, dim3(grid), dim3(256), 0, stream, static_cast<double2 *>(p), tiles, tile_pitch16, row_pitch16);
    (void)hipEventRecord(b, stream);
    float ms = 0.f;
    if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0.f)
        rate = 2.0 * (double)tiles * PCL_NFIELDS * (double)(kTileT * 8) / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipGetLastError();
    return rate;
}

// physical handles of 3855 fp64 tiles (1.07 GB; PCL_VMM_CHUNK_TILES): whole tiles, so no row of a slab ever straddles two
constexpr int64_t kChunkTilesDefault = 3855;
size_t vmm_chunk_bytes() {
    static const size_t c = [] {
        const char *e = getenv("PCL_VMM_CHUNK_TILES");
        const long t = e ? atol(e) : kChunkTilesDefault;
        return (size_t)(t > 0 ? t : kChunkTilesDefault) * kRows * kTileT * 8;
    }();
    return c;
}

// an idle handle of exactly ``len`` bytes on ``device``, if there is one (the caller maps it)
bool vmm_take_idle(size_t len, int device, hipMemGenericAllocationHandle_t *h) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    for (size_t k = g_vmm_idle.size(); k-- > 0;)
        if (g_vmm_idle[k].len == len && g_vmm_idle[k].device == device) {
            *h = g_vmm_idle[k].h;
            g_vmm_idle_bytes -= len;
            g_vmm_idle.erase(g_vmm_idle.begin() + (long)k);
            return true;
        }
    return false;
}

void vmm_keep_idle(const std::vector<hipMemGenericAllocationHandle_t> &handles, size_t total, size_t chunk, int device) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    size_t off = 0;
    for (auto h : handles) {
        const size_t len = total - off < chunk ? total - off : chunk;
        g_vmm_idle.push_back({h, len, device});
        g_vmm_idle_bytes += len;
        off += len;
    }
}

// every idle handle back to the driver
size_t vmm_release_idle() {
    std::vector<vmm_idle_handle> drop;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        drop.swap(g_vmm_idle);
        bytes = g_vmm_idle_bytes;
        g_vmm_idle_bytes = 0;
    }
    for (const vmm_idle_handle &d : drop) (void)hipMemRelease(d.h);
    return bytes;
}

hipError_t vmm_malloc(void **p, size_t bytes, int device) {
    const size_t chunk = vmm_chunk_bytes();
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    const size_t need = (bytes + chunk - 1) / chunk;
    // the last handle is cut to what is needed (a multiple of 2 MB)
    const size_t tail = bytes - (need - 1) * chunk, tail_len = (tail + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    const size_t total = (need - 1) * chunk + (tail_len < chunk ? tail_len : chunk);
    void *va = nullptr;
    hipError_t e = hipSuccess;
    for (int attempt = 0;; ++attempt) { // a range must never lie where a mapped range of this process lay before (vmm_park)
        e = hipMemAddressReserve(&va, total, (size_t)2 << 20, nullptr, 0);
        if (e != hipSuccess) return e;
        bool clash = false;
        {
            std::lock_guard<std::mutex> lk(g_vmm_mu);
            const uintptr_t a = reinterpret_cast<uintptr_t>(va);
            for (const va_range &t : g_vmm_tainted)
                if (a < t.base + t.bytes && t.base < a + total) clash = true;
            if (clash) { // the reservation is kept, unmapped: these addresses are out of circulation now
                g_vmm_parked_bytes += total;
                ++g_vmm_remaps_avoided;
            }
        }
        if (!clash) break;
        if (attempt == 7) {
            vmm_switch_off("the runtime keeps handing out addresses of freed ranges");
            return hipErrorNotSupported;
        }
    }
    vmm_block blk;
    blk.bytes = total;
    blk.chunk = chunk;
    blk.device = device;
    size_t mapped = 0;
    while (e == hipSuccess && mapped < total) {
        const size_t len = total - mapped < chunk ? total - mapped : chunk;
        hipMemGenericAllocationHandle_t h;
        if (!vmm_take_idle(len, device, &h)) {
            e = hipMemCreate(&h, len, &prop, 0);
            if (e == hipErrorOutOfMemory && vmm_release_idle() > 0) { // idle handles of other lengths held the memory
                (void)hipGetLastError();
                e = hipMemCreate(&h, len, &prop, 0);
            }
            if (e != hipSuccess) break;
        }
        e = hipMemMap(static_cast<char *>(va) + mapped, len, 0, h, 0);
        if (e != hipSuccess) {
            vmm_keep_idle({h}, len, len, device);
            break;
        }
        blk.handles.push_back(h);
        mapped += len;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, total, &acc, 1);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (mapped) (void)hipMemUnmap(va, mapped);
        vmm_keep_idle(blk.handles, mapped, chunk, device); // (the handles stay with the process: the caller's retry maps them again)
        if (!mapped) (void)hipMemAddressFree(va, total);
        else vmm_park(va, total);                          // (a range that was mapped, even in part, is never mapped again)
        return e;
    }
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_vmm[va] = std::move(blk);
    }
    *p = va;
    return hipSuccess;
}

hipError_t raw_malloc(void **p, size_t bytes, int device) {
    static const bool contig = getenv("PCL_SLAB_CONTIG") != nullptr; // EXPERIMENT (tools/pad_sweep.py): physically contiguous blocks
    if (contig && bytes >= kPoolMinBlock && hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
    if (bytes >= kPoolMinBlock && vmm_enabled()) {
        const hipError_t e = vmm_malloc(p, bytes, device);
        if (e == hipSuccess || e == hipErrorOutOfMemory) return e;
        (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
}

size_t pool_limit();

// ``release``: the caller wants the memory back at the driver (out of memory, the device is short, the pool is being emptied):
// no physical handle is kept.  Otherwise handles are kept for the next range only within the pool's own bounds -- they are
// memory the process holds, like the pool's idle blocks: pool_limit() together with those, and pool_may_keep().
bool pool_may_keep(size_t bytes);
std::atomic<size_t> g_pool_bytes_seen{0}; // g_pool_bytes as last written (raw_free runs with and without g_pool_mu held)

void raw_free(void *p, bool release = false) {
    if (!p) return;
    vmm_block blk;
    bool mine = false;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_rate.erase(p);
        auto it = g_vmm.find(p);
        if (it != g_vmm.end()) {
            blk = std::move(it->second);
            g_vmm.erase(it);
            mine = true;
        }
    }
    if (!mine) {
        (void)hipFree(p);
        return;
    }
    (void)hipDeviceSynchronize(); // hipFree would have waited for the work that still uses the block
    (void)hipMemUnmap(p, blk.bytes);
    vmm_park(p, blk.bytes);
    // which of the handles stay with the process?  As many as fit under the pool's limit beside what idles already
    size_t budget = 0;
    if (!release && pool_limit() > 0 && pool_may_keep(blk.bytes)) {
        size_t idle;
        {
            std::lock_guard<std::mutex> lk(g_vmm_mu);
            idle = g_vmm_idle_bytes;
        }
        const size_t held = g_pool_bytes_seen.load() + idle;
        budget = pool_limit() > held ? pool_limit() - held : 0;
    }
    std::vector<hipMemGenericAllocationHandle_t> keep;
    size_t off = 0, kept = 0;
    for (auto h : blk.handles) {
        const size_t len = blk.bytes - off < blk.chunk ? blk.bytes - off : blk.chunk;
        off += len;
        if (len == blk.chunk && kept + len <= budget) { // (whole chunks only: an odd-sized tail would never be asked for again)
            keep.push_back(h);
            kept += len;
        } else {
            (void)hipMemRelease(h);
        }
    }
    if (!keep.empty()) vmm_keep_idle(keep, kept, blk.chunk, blk.device);
}

// hipMemcpy2DAsync refuses a copy that reaches from one physical handle of a mapped range into the next (measured:
// "invalid argument" as soon as the rows span a handle boundary; 1-D copies and kernels do not care).  Largest number
// of rows, ``pitch`` bytes apart and ``rowbytes`` long, starting at device address d, that stay inside d's handle
// (<= k; 0: the first row itself straddles a boundary -- copy it 1-D).
int64_t rows_in_handle(const void *d, size_t pitch, size_t rowbytes, int64_t k) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    for (const auto &kv : g_vmm) {
        const char *base = static_cast<const char *>(kv.first);
        const char *p = static_cast<const char *>(d);
        if (p < base || p >= base + kv.second.bytes) continue;
        const size_t chunk = kv.second.chunk;
        if (chunk >= kv.second.bytes) return k;
        const size_t off = (size_t)(p - base), room = (off / chunk + 1) * chunk - off;
        if (room < rowbytes) return 0;
        const int64_t fit = (int64_t)((room - rowbytes) / pitch) + 1;
        return fit < k ? fit : k;
    }
    return k;
}

size_t pool_limit() {
    static const size_t lim = [] {
        const char *e = getenv("PCL_POOL_GB");
        if (e) {
            const double gb = atof(e);
            return gb > 0 ? (size_t)(gb * (double)((size_t)1 << 30)) : (size_t)0;
        }
        size_t free_b = 0, total_b = 0; // default: a third of the device's memory (96 GB on an MI355X)
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            total_b = (size_t)192 << 30;
        }
        return total_b / 3;
    }();
    return lim;
}

void pool_flush_locked() { // (everything back to the driver: no handle is kept)
    g_pool_bytes = 0;
    g_pool_bytes_seen = 0;
    for (const pool_block &b : g_pool) {
        (void)hipSetDevice(b.device);
        raw_free(b.p, true);
    }
    g_pool.clear();
}

hipError_t big_malloc(void **p, size_t bytes) {
    const bool eligible = bytes >= kPoolMinBlock && pool_limit() > 0;
    int device = 0;
    (void)hipGetDevice(&device);
    if (eligible) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int k = 0; k < (int)g_pool.size(); ++k) // smallest idle block that holds the request without wasting half of itself
            if (g_pool[k].device == device && g_pool[k].bytes >= bytes && g_pool[k].bytes <= bytes + bytes / 2 &&
                (best < 0 || g_pool[k].bytes < g_pool[best].bytes))
                best = k;
        if (best >= 0) {
            *p = g_pool[best].p;
            g_big[*p] = {g_pool[best].bytes, device};
            g_pool_bytes -= g_pool[best].bytes;
            g_pool_bytes_seen = g_pool_bytes;
            g_pool.erase(g_pool.begin() + best);
            return hipSuccess;
        }
    }
    hipError_t e = raw_malloc(p, bytes, device);
    if (e == hipErrorOutOfMemory) {
        // whichever allocator ran out (hipMemCreate of a mapped range, or hipMalloc: small blocks, the fallback): everything this
        // process holds idle goes back to the driver first -- the pool's blocks and the physical handles of unmapped ranges --,
        // then the device is synchronised (what was released is the device's again once the work that used it has ended) and the
        // request is made again, once
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            pool_flush_locked();
        }
        (void)vmm_release_idle();
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        e = raw_malloc(p, bytes, device);
        if (e == hipErrorOutOfMemory) (void)hipGetLastError();
    }
    if (e == hipSuccess && eligible) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_big[*p] = {bytes, device};
    }
    return e;
}

// A freed block is kept for the next store unless the device is short of memory: other allocators of the process (torch,
// RCCL) and other processes on the device cannot take memory out of this pool, so it only holds what leaves at least a
// quarter of the device free (PCL_POOL_GB, when set, is the only bound).
bool pool_may_keep(size_t bytes) {
    static const bool fixed = getenv("PCL_POOL_GB") != nullptr;
    if (fixed) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return free_b >= total_b / 4;
}

void big_free(void *p) {
    if (!p) return;
    big_block blk{0, -1};
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_big.find(p);
        if (it != g_big.end()) {
            blk = it->second;
            g_big.erase(it);
        }
    }
    if (blk.device < 0 || blk.bytes > pool_limit()) { // (not pooled: raw_free keeps what handles fit under the pool's limit)
        raw_free(p);
        return;
    }
    int device = 0;
    (void)hipGetDevice(&device);
    (void)hipSetDevice(blk.device);
    (void)hipDeviceSynchronize(); // hipFree would have waited for the work that still uses the block (no lock held here)
    const bool keep = pool_may_keep(blk.bytes);
    {   // physical handles kept of unmapped ranges are idle memory like the pool's blocks, and the cheaper kind to give up
        size_t idle;
        {
            std::lock_guard<std::mutex> lk(g_vmm_mu);
            idle = g_vmm_idle_bytes;
        }
        if (idle && (!keep || g_pool_bytes_seen.load() + idle + blk.bytes > pool_limit())) (void)vmm_release_idle();
    }
    std::vector<pool_block> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        while ((!keep || g_pool_bytes + blk.bytes > pool_limit()) && !g_pool.empty()) { // make room: the oldest idle block goes
            drop.push_back(g_pool.front());
            g_pool_bytes -= g_pool.front().bytes;
            g_pool.erase(g_pool.begin());
            if (!keep) break; // short of memory: this block and one old one go back to the driver
        }
        if (keep) {
            g_pool.push_back({p, blk.bytes, blk.device});
            g_pool_bytes += blk.bytes;
        }
        g_pool_bytes_seen = g_pool_bytes;
    }
    for (const pool_block &d : drop) { // (evicted to make room, or because the device is short of memory: back to the driver)
        (void)hipSetDevice(d.device);
        raw_free(d.p, true);
    }
    if (!keep) {
        (void)hipSetDevice(blk.device);
        raw_free(p, true);
    }
    (void)hipSetDevice(device);
}

int dev_alloc_bytes(void **p, int64_t n_elems, size_t esz) {
    void *q = nullptr;
    const int64_t n = ((n_elems > 0 ? n_elems : 1) + 63) & ~int64_t(63);
    PCL_HIP(big_malloc(&q, (size_t)n * esz));
    *p = q;
    return PCL_OK;
}
template <typename T>
int dev_alloc(T **p, int64_t n) {
    void *q = nullptr;
    PCL_TRY(dev_alloc_bytes(&q, n, sizeof(T)));
    *p = static_cast<T *>(q);
    return PCL_OK;
}

template <typename T>
void dev_free(T *&p) {
    if (p) big_free(p);
    p = nullptr;
}

int densify(pcl_ctx *ctx);

// scratch of the delete / gather pipelines.  The masks of a store in the alive-mask state are not scratch but state: any
// other user makes the store dense first (``for_alive``: the alive path itself asking)
int ensure_scratch(pcl_ctx *ctx, int64_t n, bool for_alive = false) {
    if (ctx->holes && !for_alive) PCL_TRY(densify(ctx));
    if (n <= ctx->scratch_cap) return PCL_OK;
    dev_free(ctx->masks);
    dev_free(ctx->masks_prev);
    dev_free(ctx->tile_keep);
    dev_free(ctx->tile_off);
    ctx->scratch_cap = 0;
    const int64_t tiles = div_up(n, kTile);
    PCL_TRY(dev_alloc(&ctx->masks, tiles * kTileRows));
    PCL_TRY(dev_alloc(&ctx->masks_prev, tiles * kTileRows));
    PCL_TRY(dev_alloc(&ctx->tile_keep, tiles));
    PCL_TRY(dev_alloc(&ctx->tile_off, tiles));
    ctx->scratch_cap = tiles * kTile;
    return PCL_OK;
}

// ---- choosing the slab among a few candidates ----------------------------------------------------------------------
//   (see "where big blocks come from": how fast the many-stream passes run is a property of the memory behind a slab,
//   fixed when it is allocated, 10-15 % apart from one allocation to the next.)  A store of >= 512 MB takes the fastest
//   of up to PCL_ALLOC_TRIES (default 6; 1 = take the first) candidate blocks -- idle pool blocks first, then fresh
//   ones -- each measured with the 13-row write sweep over the WHOLE block (a block's speed is not the sum of its
//   handles' speeds measured one at a time: tools/attic/vmm_chunk_probe.hip); the others go (back) to the pool with their
//   rate remembered, where the compaction's second slab finds them.  ~10 ms per candidate at 1e8 photons, once per store.
//   Six since round 5 (four before): eight fresh processes with eight candidates each (tools/attic/alloc_tries.py,
//   profiles/r05_alloc_tries.log) -- a candidate sweeps at 5.6-5.9 TB/s or at 6.4-7.0, about half of them each way, the
//   one-step kernel runs at 0.73-0.74 of the HBM peak on any block that sweeps above ~6.7 and at 0.68 on one at 6.4; the best
//   of the first four was below 6.6 in one process of eight (and in the driver's round-4 run: 5.75 5.55 5.98 6.39), the best
//   of the first six at least 6.76 in all eight.
int tries_wanted() {
    static const int n = [] {
        const char *e = getenv("PCL_ALLOC_TRIES");
        const int v = e ? atoi(e) : 6;
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    return n;
}

double block_rate(void *blk, int64_t tiles64, hipStream_t stream) {
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_rate.find(blk);
        if (it != g_rate.end()) return it->second;
    }
    const double r = sweep_rate_raw(blk, tiles64, stream);
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    g_rate[blk] = r;
    return r;
}

int alloc_slab(pcl_ctx *ctx, void **out) {
    const size_t bytes = slab_bytes(ctx);
    // (candidates that lose must fit the pool: handing tens of GB back to the driver is what stalls the next allocation)
    // slab, second slab and every candidate that lost end up in the pool when the store goes: they must all fit
    const int fit = (int)(pool_limit() / bytes) - 1;
    int tries = (bytes >= ((size_t)512 << 20) && row_pad_bytes() == 0 && fit > 1) ? (tries_wanted() < fit ? tries_wanted() : fit) : 1;
    {   // a store above 1/8 of the device's memory compares two candidates at most: the transient footprint of the
        // selection (candidates + the slab in use) stays below half of the device whatever PCL_ALLOC_TRIES says
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
        else if (bytes > total_b / 8 && tries > 2) tries = 2;
    }
    if (tries == 1) {
        if (out == &ctx->slab) ctx->slab_tries = 0, ctx->slab_chosen = 0.0;
        PCL_HIP(big_malloc(out, bytes));
        return PCL_OK;
    }
    const int64_t tiles64 = (int64_t)(bytes / ((size_t)kRows * kTileT * 8)); // the sweep's tiles are fp64-sized whatever the dtype
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    void *best = nullptr;
    double best_rate = -1.0;
    std::vector<void *> losers;
    std::vector<double> seen;
    for (int k = 0; k < tries; ++k) {
        if (k > 0) { // another candidate only while twice its size is still free
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bytes) break;
        }
        void *cand = nullptr;
        if (big_malloc(&cand, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        const double r = block_rate(cand, tiles64, ctx->stream);
        seen.push_back(r);
        if (r > best_rate) {
            if (best) losers.push_back(best);
            best = cand;
            best_rate = r;
        } else {
            losers.push_back(cand);
        }
    }
    if (out == &ctx->slab) { // what pcl_store_alloc_info reports
        ctx->slab_tries = (int)seen.size();
        for (int k = 0; k < 8; ++k) ctx->slab_rates[k] = k < (int)seen.size() ? seen[k] : 0.0;
        ctx->slab_chosen = best_rate;
    }
    static const bool debug = getenv("PCL_ALLOC_DEBUG") != nullptr;
    if (debug) {
        fprintf(stderr, "physicl_hip: slab of %.2f GB: chose %.0f GB/s among", bytes / 1e9, best_rate);
        for (double r : seen) fprintf(stderr, " %.0f", r);
        fprintf(stderr, "\n");
    }
    for (void *l : losers) big_free(l); // only now: a freed candidate must not come back as the next one
    if (!best) PCL_HIP(big_malloc(&best, bytes)); // reports the allocation error
    *out = best;
    return PCL_OK;
}

// zero three rows (dv0..2) of every tile of a slab
__global__ void __launch_bounds__(256) k_zero_rows3(double2 *r0, double2 *r1, double2 *r2, int64_t tiles, int64_t tile_pitch16, int row_len16) {
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x)
        for (int q = threadIdx.x; q < row_len16; q += 256) {
            r0[t * tile_pitch16 + q] = make_double2(0.0, 0.0);
            r1[t * tile_pitch16 + q] = make_double2(0.0, 0.0);
            r2[t * tile_pitch16 + q] = make_double2(0.0, 0.0);
        }
}

int ensure_alt(pcl_ctx *ctx) {
    if (!ctx->slab_alt) {
        PCL_TRY(alloc_slab(ctx, &ctx->slab_alt));
        refresh_rows(ctx);
        // The new slab's dv rows are zeroed once, here: a delete run never scatters, its dv is all +0.0, and a
        // compaction whose destination is known to hold zeros does not move the dv rows at all (launch_compact_count:
        // 24 of the 88 bytes a survivor costs, and 24 bytes per SLOT of reads) -- from the FIRST compaction on.
        const int64_t tile_pitch16 = (int64_t)(kRows * row_pitch_bytes(ctx) / 16);
        const int row_len16 = (int)(kTileT * ctx->esz / 16);
        const int grid = (int)(ctx->tiles < 4096 ? ctx->tiles : 4096);
        hipLaunchKernelGGL(k_zero_rows3, dim3(grid), dim3(256), 0, ctx->stream, static_cast<double2 *>(ctx->field_alt[PCL_DV0]),
                           static_cast<double2 *>(ctx->field_alt[PCL_DV1]), static_cast<double2 *>(ctx->field_alt[PCL_DV2]), ctx->tiles,
                           tile_pitch16, row_len16);
        PCL_TRY(launch_check("k_zero_rows3"));
        ctx->alt_dv_zero_n = ctx->tiles * kTileT;
    }
    if (!ctx->ids) PCL_TRY(dev_alloc(&ctx->ids, ctx->capacity));
    if (!ctx->ids_alt) PCL_TRY(dev_alloc(&ctx->ids_alt, ctx->capacity));
    if (ctx->kind && !ctx->kind_alt) PCL_TRY(dev_alloc(&ctx->kind_alt, ctx->capacity));
    return PCL_OK;
}

// Make dr / dv real arrays again after lazy fused steps (no-op otherwise).
template <typename T>
int materialize_t(pcl_ctx *ctx) {
    const int64_t N = ctx->count;
    if (N > 0) {
        materialize_args<T> a{};
        for (int k = 0; k < 3; ++k) {
            a.vmove[k] = static_cast<const T *>(ctx->lazy_dr_vprev ? ctx->vprev[k] : ctx->field[PCL_V0 + k]);
            a.vprev[k] = static_cast<const T *>(ctx->vprev[k]);
            a.v[k] = F<T>(ctx, PCL_V0 + k);
            a.dr[k] = F<T>(ctx, PCL_DR0 + k);
            a.dv[k] = F<T>(ctx, PCL_DV0 + k);
        }
        a.kind = ctx->kind;
        a.dt = (T)ctx->lazy_dt;
        a.do_dr = ctx->lazy_dr ? 1 : 0;
        a.do_dv = ctx->lazy_dv ? 1 : 0;
        a.N = N;
        a.ts = tile_stride(ctx);
        hipLaunchKernelGGL(k_materialize<T>, dim3(grid_for(ctx, N, kBlock)), dim3(kBlock), 0, ctx->stream, a);
        PCL_TRY(launch_check("k_materialize"));
    }
    return PCL_OK;
}

int materialize(pcl_ctx *ctx) {
    if (!ctx->lazy_dr && !ctx->lazy_dv) return PCL_OK;
    PCL_TRY(PCL_DISPATCH(ctx, materialize_t<double>(ctx), materialize_t<float>(ctx)));
    if (ctx->lazy_dv) set_dv_zero(ctx, 2); // real dv values were just written
    ctx->lazy_dr = ctx->lazy_dv = ctx->lazy_dr_vprev = false;
    return PCL_OK;
}

int ahead_commit(pcl_ctx *ctx);

// ``keep_ahead``: pcl_step_fused_delete itself asking -- it may be the call the bodies worked out ahead are waiting for
int need_store_raw(pcl_ctx *ctx, bool keep_ahead = false) {
    PCL_TRY(bind(ctx));
    if (ctx->capacity <= 0) return fail(PCL_ERR_STATE, "no particle store: call pcl_store_alloc first");
    if (ctx->ahead.active && !keep_ahead) PCL_TRY(ahead_commit(ctx));
    return PCL_OK;
}

// every entry point that reads or writes store arrays goes through here: implicit dr/dv become real first
int need_store(pcl_ctx *ctx) {
    PCL_TRY(need_store_raw(ctx));
    PCL_TRY(densify(ctx)); // a store behind an alive mask becomes dense (stable) first
    return materialize(ctx);
}

int check_range(pcl_ctx *ctx, int64_t offset, int64_t n, const void *host) {
    if (n < 0 || offset < 0 || offset + n > ctx->capacity)
        return fail(PCL_ERR_ARG, "range [%lld, %lld) outside store capacity %lld", (long long)offset,
                    (long long)(offset + n), (long long)ctx->capacity);
    if (n > 0 && !host) return fail(PCL_ERR_ARG, "host pointer is NULL");
    return PCL_OK;
}

// ---- hipRTC, loaded at run time -------------------------------------------------------------------------------------
// libphysicl_hip.so does not link libhiprtc: a machine without it still loads the library and runs everything except
// user-written kernels and variable_n_fn expressions outside the three built-in shapes (match_nprof below).
struct hiprtc_api {
    void *handle = nullptr;
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
    decltype(&hiprtcVersion) Version = nullptr;
    std::string why; // why it is unavailable
    bool ok = false;
};

hiprtc_api load_hiprtc() {
    hiprtc_api a;
    const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"};
    for (const char *nm : names) {
        a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
    }
    if (!a.handle) {
        const char *e = dlerror();
        a.why = std::string("libhiprtc.so could not be loaded") + (e ? std::string(": ") + e : std::string());
        return a;
    }
#define PCL_RTC_SYM(field, sym)                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, #sym));        \
    if (!a.field) {                                                              \
        a.why = "libhiprtc.so lacks " #sym;                                      \
        return a;                                                                \
    }
    PCL_RTC_SYM(CreateProgram, hiprtcCreateProgram)
    PCL_RTC_SYM(CompileProgram, hiprtcCompileProgram)
    PCL_RTC_SYM(GetProgramLogSize, hiprtcGetProgramLogSize)
    PCL_RTC_SYM(GetProgramLog, hiprtcGetProgramLog)
    PCL_RTC_SYM(GetCodeSize, hiprtcGetCodeSize)
    PCL_RTC_SYM(GetCode, hiprtcGetCode)
    PCL_RTC_SYM(DestroyProgram, hiprtcDestroyProgram)
    PCL_RTC_SYM(GetErrorString, hiprtcGetErrorString)
    PCL_RTC_SYM(Version, hiprtcVersion)
#undef PCL_RTC_SYM
    a.ok = true;
    return a;
}

// PCL_NO_RTC (any value): behave as if libhiprtc were absent -- how the tests exercise the ahead-of-time fallback
const hiprtc_api &rtc_api() {
    static const hiprtc_api real = load_hiprtc();
    static const hiprtc_api off = [] {
        hiprtc_api a;
        a.why = "hipRTC switched off by PCL_NO_RTC";
        return a;
    }();
    return getenv("PCL_NO_RTC") ? off : real;
}

// ---- the three built-in variable_n_fn shapes (pcl_device.h, pcl_nprof) ---------------------------------------------------
// Token-level match of the user's text (already validated) against
//   1  NUM * exp ( rA [ gid ] - NUM )
//   2  NUM * exp ( - 1 * ( sqrt ( pow ( r0 [ gid ] , 2 ) + pow ( r1 [ gid ] , 2 ) + pow ( r2 [ gid ] , 2 ) ) - NUM ) / ( NUM ) )
//   3  NUM * exp ( rA [ gid ] / NUM )
// NUM = an unsigned literal of the text; its double value comes from strtod, its float value from strtof of the very
// characters (what the compiler would have done with the f-suffixed spelling under hipRTC).
bool match_nprof(const char *expr, rtc_entry *ent) {
    std::vector<std::string> tok;
    for (const char *q = expr; *q;) {
        if (*q == ' ' || *q == '\t' || *q == '\n') {
            ++q;
        } else if ((*q >= '0' && *q <= '9') || *q == '.') {
            const char *b = q;
            while ((*q >= '0' && *q <= '9') || *q == '.') ++q;
            if (*q == 'e' || *q == 'E') {
                ++q;
                if (*q == '+' || *q == '-') ++q;
                while (*q >= '0' && *q <= '9') ++q;
            }
            tok.push_back(std::string(b, q));
        } else if ((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || *q == '_') {
            const char *b = q;
            while ((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || (*q >= '0' && *q <= '9') || *q == '_') ++q;
            tok.push_back(std::string(b, q));
        } else {
            tok.push_back(std::string(1, *q++));
        }
    }
    static const char *const shapes[3] = {
        "NUM * exp ( AX [ gid ] - NUM )",
        "NUM * exp ( - 1 * ( sqrt ( pow ( r0 [ gid ] , 2 ) + pow ( r1 [ gid ] , 2 ) + pow ( r2 [ gid ] , 2 ) ) - NUM ) / ( NUM ) )",
        "NUM * exp ( AX [ gid ] / NUM )"};
    for (int sh = 0; sh < 3; ++sh) {
        std::vector<std::string> pat;
        for (const char *q = shapes[sh]; *q;) {
            const char *b = q;
            while (*q && *q != ' ') ++q;
            pat.push_back(std::string(b, q));
            while (*q == ' ') ++q;
        }
        if (pat.size() != tok.size()) continue;
        double pd[3] = {0, 0, 0};
        float pf[3] = {0, 0, 0};
        int np = 0, axis = 0;
        bool ok = true;
        for (size_t i = 0; ok && i < pat.size(); ++i) {
            const std::string &t = tok[i];
            if (pat[i] == "NUM") {
                ok = !t.empty() && ((t[0] >= '0' && t[0] <= '9') || t[0] == '.') && np < 3;
                if (ok) {
                    const bool floating = t.find_first_of(".eE") != std::string::npos;
                    pd[np] = strtod(t.c_str(), nullptr);
                    pf[np] = floating ? strtof(t.c_str(), nullptr) : (float)strtol(t.c_str(), nullptr, 10); // int literal -> float
                    ++np;
                }
            } else if (pat[i] == "AX") {
                ok = t == "r0" || t == "r1" || t == "r2";
                if (ok) axis = t[1] - '0';
            } else {
                ok = pat[i] == t;
            }
        }
        if (!ok) continue;
        ent->np64 = {sh + 1, axis, pd[0], pd[1], pd[2]};
        ent->np32 = {sh + 1, axis, pf[0], pf[1], pf[2]};
        return true;
    }
    return false;
}

void set_np(pcl_nprof<double> &np, const rtc_entry *ent) { np = ent ? ent->np64 : pcl_nprof<double>{0, 0, 0.0, 0.0, 0.0}; }
void set_np(pcl_nprof<float> &np, const rtc_entry *ent) { np = ent ? ent->np32 : pcl_nprof<float>{0, 0, 0.f, 0.f, 0.f}; }

// ---- hipRTC specialisation cache ---------------------------------------------------------------------
// Two levels: compiled code objects are kept per process (keyed by arch + defines + expression), loaded
// modules per context.  A second Simulation with the same variable_n_fn only pays hipModuleLoadData.
std::mutex g_code_mutex;
std::map<std::string, std::vector<char>> g_code_cache;

int load_rtc_module(pcl_ctx *ctx, const char *expr, const std::vector<char> &code, rtc_entry **out);

// ---- on-disk cache of hipRTC code objects: $PCL_RTC_CACHE (a directory; "off" disables), default ~/.cache/physicl_amd/rtc
uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

std::string rtc_cache_path(const std::string &src, const std::string &arch, const std::string &extra) {
    const char *dir_env = getenv("PCL_RTC_CACHE");
    std::string dir;
    if (dir_env && *dir_env) {
        if (!strcmp(dir_env, "off") || !strcmp(dir_env, "0")) return "";
        dir = dir_env;
    } else {
        const char *home = getenv("HOME");
        if (!home || !*home) return "";
        dir = std::string(home) + "/.cache/physicl_amd/rtc";
    }
    int major = 0, minor = 0;
    if (rtc_api().ok) (void)rtc_api().Version(&major, &minor);
    char tag[64];
    snprintf(tag, sizeof tag, "|%d.%d|abi%d", major, minor, PCL_ABI_VERSION);
    const uint64_t h1 = fnv1a(src + "|" + arch + "|" + extra + tag);
    const uint64_t h2 = fnv1a(arch + tag + "|" + extra + "|" + src, 0x9E3779B97F4A7C15ull);
    char name[64];
    snprintf(name, sizeof name, "/%016llx%016llx.hsaco", (unsigned long long)h1, (unsigned long long)h2);
    // mkdir -p, private to the user: a code object found here is loaded into the GPU context, so the directory must not be
    // writable by anyone else -- a directory that is not ours, or is group/world-writable, is not used at all
    std::string partial;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || (dir[i] == '/' && i > 0)) {
            partial = dir.substr(0, i);
            if (!partial.empty()) (void)mkdir(partial.c_str(), 0700);
        }
    }
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return "";
    return dir + name;
}

bool read_file(const std::string &path, std::vector<char> *out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<char> buf;
    char chunk[65536];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    const bool ok = !ferror(f);
    fclose(f);
    if (ok) out->swap(buf);
    return ok;
}

void write_file_atomic(const std::string &path, const std::vector<char> &data) {
    // (process AND thread: the contexts of a device group compile the same text side by side on their own threads)
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp%ld.%zx", (long)getpid(), std::hash<std::thread::id>()(std::this_thread::get_id()));
    const std::string t = path + tmp;
    FILE *f = fopen(t.c_str(), "wb");
    if (!f) return; // read-only home, full disk ...: the cache is an optimisation only
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}

// cache file = "PCLRTC01" | payload size (u64) | FNV-1a of the payload (u64) | payload.  A file that fails any of the
// three checks is treated as absent: the module loader must never see a truncated or foreign code object.
bool read_cached_code(const std::string &path, std::vector<char> *code) {
    std::vector<char> raw;
    if (!read_file(path, &raw) || raw.size() < 24 || memcmp(raw.data(), "PCLRTC01", 8) != 0) return false;
    uint64_t size = 0, sum = 0;
    memcpy(&size, raw.data() + 8, 8);
    memcpy(&sum, raw.data() + 16, 8);
    if (size == 0 || size != raw.size() - 24) return false;
    if (fnv1a(std::string(raw.data() + 24, (size_t)size)) != sum) return false;
    code->assign(raw.begin() + 24, raw.end());
    return true;
}

void write_cached_code(const std::string &path, const std::vector<char> &code) {
    std::vector<char> raw(24 + code.size());
    const uint64_t size = code.size(), sum = fnv1a(std::string(code.data(), code.size()));
    memcpy(raw.data(), "PCLRTC01", 8);
    memcpy(raw.data() + 8, &size, 8);
    memcpy(raw.data() + 16, &sum, 8);
    memcpy(raw.data() + 24, code.data(), code.size());
    write_file_atomic(path, raw);
}

int compile_rtc(pcl_ctx *ctx, const char *expr, const std::string &expr_f32, int dt, bool use_e, const std::string &key, rtc_entry **out);

// The implementation of a variable_n_fn expression for this context: its hipRTC specialisation, or -- when hipRTC is
// not available on this machine (or the compile fails) and the text is one of the three built-in shapes -- the
// ahead-of-time VAR_N kernels with the text's literals as parameters (entry with module == nullptr).
int load_rtc_into(const std::vector<char> &code, rtc_entry &ent); // below

// a background compile that has finished is taken over: from here on the entry runs the specialised kernels
void poll_job(rtc_entry &ent, bool wait) {
    if (!ent.job) return;
    if (!wait && ent.job->state.load() == 0) return;
    if (ent.job->th.joinable()) ent.job->th.join();
    if (ent.job->state.load() == 1) {
        rtc_entry up;
        if (load_rtc_into(ent.job->code, up) == PCL_OK) {
            up.np64 = ent.np64;
            up.np32 = ent.np32;
            ent = up; // (job dropped with the old value)
            return;
        }
        if (up.module) (void)hipModuleUnload(up.module);
        (void)hipGetLastError();
    }
    ent.job.reset(); // compile or load failed: the ahead-of-time kernels stay
}

// One entry per (text, element type, wavelength term): a store has one element type and a step one setting of the
// wavelength term, so only that quarter of the kernels is compiled (1.7 s instead of 2.5 s for the first use of a text).
int get_rtc(pcl_ctx *ctx, const char *expr, int dt, bool use_e, rtc_entry **out) {
    std::string expr_f32;
    PCL_TRY(validate_expr(expr, &expr_f32));
    const std::string key = std::string(expr) + (dt ? "\x01" "f32" : "\x01" "f64") + (use_e ? "e1" : "e0");
    auto it = ctx->rtc.find(key);
    if (it != ctx->rtc.end()) {
        poll_job(it->second, false);
        *out = &it->second;
        return PCL_OK;
    }
    std::string why = rtc_api().why;
    if (rtc_api().ok) {
        if (compile_rtc(ctx, expr, expr_f32, dt, use_e, key, out) == PCL_OK) return PCL_OK;
        why = g_err;
        ctx->rtc.erase(key);
    }
    rtc_entry ent;
    if (!match_nprof(expr, &ent))
        return fail(PCL_ERR_RTC, "variable_n_fn \"%s\" needs hipRTC (%s) -- without it only the built-in shapes "
                    "\"K * exp(rA[gid] - X)\", \"K * exp(rA[gid] / X)\" and the radial exponential of the examples run", expr,
                    why.c_str());
    auto ins = ctx->rtc.emplace(key, ent);
    *out = &ins.first->second;
    return PCL_OK;
}

int compile_rtc(pcl_ctx *ctx, const char *expr, const std::string &expr_f32, int dt, bool use_e, const std::string &key, rtc_entry **out) {
    const char *extra_env = getenv("PCL_RTC_EXTRA"), *define_env = getenv("PCL_RTC_DEFINE");
    const std::string code_key = std::string(ctx->prop.gcnArchName) + "|" + (extra_env ? extra_env : "") + "|" +
                                 (define_env ? define_env : "") + "|" + key;
    {
        std::lock_guard<std::mutex> lock(g_code_mutex);
        auto ci = g_code_cache.find(code_key);
        if (ci != g_code_cache.end()) return load_rtc_module(ctx, key.c_str(), ci->second, out);
    }
    std::string src = "#define PCL_RTC 1\n";
    src += dt ? "#define PCL_RTC_DT 1\n" : "#define PCL_RTC_DT 0\n";
    src += use_e ? "#define PCL_RTC_E 1\n" : "#define PCL_RTC_E 0\n";
    // perf-experiment hook (never set in production): PCL_RTC_EXTRA="NAME1,NAME2" -> "#define NAME 1" lines
    if (const char *extra = getenv("PCL_RTC_EXTRA")) {
        std::string tok;
        for (const char *q = extra;; ++q) {
            if (*q == ',' || *q == '\0') {
                if (!tok.empty()) src += "#define " + tok + " 1\n";
                tok.clear();
                if (!*q) break;
            } else if ((*q >= 'A' && *q <= 'Z') || (*q >= '0' && *q <= '9') || *q == '_') {
                tok += *q;
            }
        }
    }
    src += "#define PCL_N_EXPR (";
    src += expr;
    src += ")\n#define PCL_N_EXPR_F (";
    src += expr_f32;
    src += ")\n";
    src += pcl_rtc_source;
    std::string arch = std::string("--offload-arch=") + ctx->prop.gcnArchName;
    const char *extra = getenv("PCL_RTC_DEFINE"); // timing experiments only: e.g. -DPCL_ABLATE_TRIG
    // code objects are kept on disk, keyed by a hash of everything that determines them (the whole specialised source
    // text, target, options, hipRTC version): a second process starts without the ~2 s compile
    // Every option of the compile in one blank-separated string (it is part of the cache key).  -disable-machine-licm: the
    // pre-RA pass hoists every scalar constant of the K loop (Philox keys, exp / sincos coefficients) out of it, the scalar
    // register file overflows, and the overflow comes back into the loop as v_readlane -- VALU work in kernels bound by
    // VALU issue.  Without the pass the constants are rebuilt in the loop by the otherwise idle scalar unit: K-step pass
    // +1 .. 3 %, same bits (profiles/r05_ab_machine_licm.md; PCL_RTC_LICM=1 compiles with the pass, for A/B runs).
    std::string all_opts = "-O3 -ffp-contract=off -std=c++17";
    if (!getenv("PCL_RTC_LICM")) all_opts += " -mllvm -disable-machine-licm";
    if (extra && *extra) all_opts += std::string(" ") + extra;
    const std::string disk = rtc_cache_path(src, arch, all_opts);
    if (!disk.empty()) {
        std::vector<char> cached;
        if (read_cached_code(disk, &cached)) {
            rtc_entry *e = nullptr;
            if (load_rtc_module(ctx, key.c_str(), cached, &e) == PCL_OK) {
                std::lock_guard<std::mutex> lock(g_code_mutex);
                g_code_cache[code_key] = cached;
                *out = e;
                return PCL_OK;
            }
            ctx->rtc.erase(key); // unreadable / stale file: compile again and overwrite it
        }
    }
    // the compile itself: hipRTC only, no device call -- it may run on another thread
    auto compile = [](const std::string src, const std::string arch, const std::string extra, const std::string expr,
                      const std::string code_key, const std::string disk, std::vector<char> *code, std::string *err) {
        hiprtcProgram prog;
        hiprtcResult r = rtc_api().CreateProgram(&prog, src.c_str(), "pcl_variable_n.hip", 0, nullptr, nullptr);
        if (r != HIPRTC_SUCCESS) {
            *err = std::string("hiprtcCreateProgram: ") + rtc_api().GetErrorString(r);
            return false;
        }
        std::vector<std::string> more; // the options, one per blank-separated word
        for (size_t i = 0; i < extra.size();) {
            const size_t j = extra.find(' ', i);
            if (j != i) more.push_back(extra.substr(i, j == std::string::npos ? j : j - i));
            if (j == std::string::npos) break;
            i = j + 1;
        }
        std::vector<const char *> opts = {arch.c_str()};
        for (const std::string &m : more) opts.push_back(m.c_str());
        r = rtc_api().CompileProgram(prog, (int)opts.size(), opts.data());
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0;
            rtc_api().GetProgramLogSize(prog, &n);
            std::string log(n, '\0');
            if (n) rtc_api().GetProgramLog(prog, &log[0]);
            rtc_api().DestroyProgram(&prog);
            *err = "hipRTC could not compile variable_n_fn \"" + expr + "\": " + rtc_api().GetErrorString(r) + "\n" + log;
            return false;
        }
        size_t code_n = 0;
        rtc_api().GetCodeSize(prog, &code_n);
        code->resize(code_n);
        rtc_api().GetCode(prog, code->data());
        rtc_api().DestroyProgram(&prog);
        {
            std::lock_guard<std::mutex> lock(g_code_mutex);
            g_code_cache[code_key] = *code;
        }
        if (!disk.empty()) write_cached_code(disk, *code);
        return true;
    };
    const std::string extra_s = all_opts;
    // One of the built-in shapes, nothing cached: the ahead-of-time kernels start at once and the specialisation is
    // compiled beside them (~2 s); get_rtc takes it over when it is ready, pcl_ctx_rtc_wait waits for it.  Same bits
    // either way (tests/test_gpu_aot_fallback.py).  Asked for per context (pcl_ctx_set_rtc_background: Simulation does,
    // a bare context compiles first, so that a measurement never times the stand-in kernels); PCL_RTC_SYNC overrides.
    static const bool sync_env = getenv("PCL_RTC_SYNC") != nullptr;
    rtc_entry aot;
    if (ctx->rtc_background && !sync_env && match_nprof(expr, &aot)) {
        aot.job = std::make_shared<rtc_job>();
        rtc_job *job = aot.job.get();
        {   // a process that ends while a compile is running waits for it (contexts that were never destroyed)
            static std::mutex mu;
            static std::vector<std::weak_ptr<rtc_job>> *all = nullptr;
            std::lock_guard<std::mutex> lk(mu);
            if (!all) {
                all = new std::vector<std::weak_ptr<rtc_job>>();
                std::atexit([] {
                    std::lock_guard<std::mutex> lk2(mu);
                    for (auto &w : *all)
                        if (auto j = w.lock())
                            if (j->th.joinable()) j->th.join();
                });
            }
            all->push_back(aot.job);
        }
        const std::string expr_s = expr;
        job->th = std::thread([=]() {
            const bool ok = compile(src, arch, extra_s, expr_s, code_key, disk, &job->code, &job->err);
            job->state.store(ok ? 1 : 2);
        });
        auto ins = ctx->rtc.emplace(key, aot);
        *out = &ins.first->second;
        return PCL_OK;
    }
    std::vector<char> code;
    std::string err;
    if (!compile(src, arch, extra_s, expr, code_key, disk, &code, &err)) return fail(PCL_ERR_RTC, "%s", err.c_str());
    return load_rtc_module(ctx, key.c_str(), code, out);
}

// module + kernel handles of a compiled specialisation into ``ent``
int load_rtc_into(const std::vector<char> &code, rtc_entry &ent) {
    PCL_HIP(hipModuleLoadData(&ent.module, code.data()));
    // a module holds the kernels of ONE element type and wavelength setting (PCL_RTC_DT / PCL_RTC_E): the rest stay NULL
    auto get = [&](hipFunction_t *f, const char *name) {
        if (hipModuleGetFunction(f, ent.module, name) != hipSuccess) *f = nullptr;
    };
    get(&ent.sphere[0], "pcl_rtc_sphere_e0");
    get(&ent.sphere[1], "pcl_rtc_sphere_e1");
    const char *dt_tag[2] = {"", "f_"};
    for (int d = 0; d < 2; ++d)
        for (int e = 0; e < 2; ++e) {
            char nm[64];
            snprintf(nm, sizeof nm, "pcl_rtc_scatter_%se%d", dt_tag[d], e);
            get(&ent.scatter[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_fused_%se%d", dt_tag[d], e);
            get(&ent.fused[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_fast_%se%d", dt_tag[d], e);
            get(&ent.fast[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_multi_%se%d", dt_tag[d], e);
            get(&ent.multi[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_fastg_%se%d", dt_tag[d], e);
            get(&ent.fastg[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_mixed_%se%d", dt_tag[d], e);
            get(&ent.mixed[d][e], nm);
            snprintf(nm, sizeof nm, "pcl_rtc_mixed3_%se%d", dt_tag[d], e);
            if (hipModuleGetFunction(&ent.mixed3[d][e], ent.module, nm) != hipSuccess) ent.mixed3[d][e] = nullptr;
            snprintf(nm, sizeof nm, "pcl_rtc_trace_%se%d", dt_tag[d], e);
            get(&ent.trace[d][e], nm);
        }
    for (int e = 0; e < 2; ++e) { // fp64 modules only
        char nm[64];
        snprintf(nm, sizeof nm, "pcl_rtc_multi2_e%d", e);
        if (hipModuleGetFunction(&ent.multi2[e], ent.module, nm) != hipSuccess) ent.multi2[e] = nullptr;
        snprintf(nm, sizeof nm, "pcl_rtc_multis_e%d", e);
        if (hipModuleGetFunction(&ent.multis[e], ent.module, nm) != hipSuccess) ent.multis[e] = nullptr;
        snprintf(nm, sizeof nm, "pcl_rtc_multi2s_e%d", e);
        if (hipModuleGetFunction(&ent.multi2s[e], ent.module, nm) != hipSuccess) ent.multi2s[e] = nullptr;
        snprintf(nm, sizeof nm, "pcl_rtc_multi3_e%d", e);
        if (hipModuleGetFunction(&ent.multi3[e], ent.module, nm) != hipSuccess) ent.multi3[e] = nullptr;
        snprintf(nm, sizeof nm, "pcl_rtc_multi3s_e%d", e);
        if (hipModuleGetFunction(&ent.multi3s[e], ent.module, nm) != hipSuccess) ent.multi3s[e] = nullptr;
    }
    (void)hipGetLastError();
    return PCL_OK;
}

int load_rtc_module(pcl_ctx *ctx, const char *expr, const std::vector<char> &code, rtc_entry **out) {
    rtc_entry ent;
    PCL_TRY(load_rtc_into(code, ent));
    auto ins = ctx->rtc.emplace(std::string(expr), ent);
    *out = &ins.first->second;
    return PCL_OK;
}

constexpr size_t kProfMax = 16384;

// record the start event of a timed launch; returns the slot or -1 when profiling is off / full
int prof_begin(pcl_ctx *ctx, int kid) {
    if (!ctx->prof_on || ctx->prof_used >= kProfMax) return -1;
    if (ctx->prof_used == ctx->prof.size()) {
        pcl_ctx::prof_slot s{nullptr, nullptr, kid};
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
        ctx->prof.push_back(s);
    }
    const int i = (int)ctx->prof_used++;
    ctx->prof[i].kid = kid;
    (void)hipEventRecord(ctx->prof[i].a, ctx->stream);
    return i;
}

void prof_end(pcl_ctx *ctx, int slot) {
    if (slot >= 0) (void)hipEventRecord(ctx->prof[slot].b, ctx->stream);
}

template <typename Args>
int launch_module(pcl_ctx *ctx, hipFunction_t fn, int grid, Args &args, const char *what) {
    size_t sz = sizeof(Args);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    PCL_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, kBlock, 1, 1, 0, ctx->stream, nullptr, config));
    return launch_check(what);
}

// run pass 2 of the compaction pipeline on tile_keep already produced for n particles
int compact_lds_min_pct() { // perf-experiment hook: survivor percentage above which pass 3 is k_compact_lds
    static const int pct = [] {
        const char *e = getenv("PCL_COMPACT_LDS_MIN_PCT");
        return e ? atoi(e) : 15; // (round 2: 35; re-measured with the alive path, whose compactions start at 34 % survivors)
    }();
    return pct;
}

int scan_tiles(pcl_ctx *ctx, int64_t n) {
    const int64_t tiles = div_up(n, kTile);
    const int lds_min_pct = compact_lds_min_pct();
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->tile_keep, tiles, ctx->tile_off,
                       reinterpret_cast<int64_t *>(ctx->d_cnt + kCounterSlots - 1), n,
                       reinterpret_cast<int *>(ctx->d_cnt + kCounterSlots - 2),
                       reinterpret_cast<int64_t *>(ctx->h_cnt + kCounterSlots - 1), lds_min_pct);
    PCL_TRY(launch_check("k_tile_scan"));
    // the host needs the survivor count (next launch geometry, exit tests), not the end of the compaction that follows:
    // it waits for this event and prepares the next step while pass 3 is still moving the survivors
    PCL_HIP(hipEventRecord(ctx->ev_count, ctx->stream));
    return PCL_OK;
}

// survivor count of the scan most recently enqueued by scan_tiles
// Waiting for the stream (or an event on it) where the wait is usually SHORT -- a K-pass launch on a 1e7-photon store is a
// few hundred microseconds, and a host thread that blocks pays the wake-up on top, with a jitter as large as the kernel:
// poll for up to 1.5 ms, then block (PCL_SPIN_US changes the limit, 0 = always block).
int64_t spin_limit_us() {
    static knob k("PCL_SPIN_US");
    const double v = k.value(1500.0);
    return (int64_t)(v > 0 ? v : 0);
}
template <typename Query, typename Block>
int spin_then_block(Query query, Block block, const char *what) {
    const int64_t limit = spin_limit_us();
    if (limit > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1;; ++spins) {
            const hipError_t e = query();
            if (e == hipSuccess) return PCL_OK;
            if (e != hipErrorNotReady) return fail(PCL_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
            if ((spins & 15u) == 0 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > limit)
                break;
        }
    }
    const hipError_t e = block();
    return e == hipSuccess ? PCL_OK : fail(PCL_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
}
int stream_wait(pcl_ctx *ctx) {
    return spin_then_block([&] { return hipStreamQuery(ctx->stream); }, [&] { return hipStreamSynchronize(ctx->stream); }, "waiting for the stream");
}

int wait_count(pcl_ctx *ctx, int64_t before, int64_t *alive_out) {
    PCL_TRY(spin_then_block([&] { return hipEventQuery(ctx->ev_count); }, [&] { return hipEventSynchronize(ctx->ev_count); },
                            "waiting for the survivor count"));
    const int64_t alive = (int64_t)ctx->h_cnt[kCounterSlots - 1];
    if (alive < 0 || alive > before)
        return fail(PCL_ERR_HIP, "compaction produced an impossible count %lld of %lld", (long long)alive, (long long)before);
    *alive_out = alive;
    return PCL_OK;
}

// the store after a compaction into the other slab
void adopt_compacted(pcl_ctx *ctx, int64_t alive, int64_t last_delete_n) {
    // dv bookkeeping.  The slab left behind keeps zero dv rows over the old count if the live dv rows were all zero.  The
    // new slab's dv rows hold what was moved (zero iff it was zero), zeros that were there (skip), or -- vprev travelled in
    // their place -- whatever the other slab held, in which case "all zero" is no longer known.
    // Both slabs carry a high-water mark "elements [0, z) of my dv rows are +0.0": rows nobody writes keep what they hold, so
    // a store that is filled again (the second run of a script, a bench repetition) finds the other slab's dv rows still
    // zero over everything and skips them in its FIRST compaction too.
    const int64_t left_behind = ctx->dv_zero == 1 ? (ctx->dv_zero_n > ctx->count ? ctx->dv_zero_n : ctx->count) : 0;
    int64_t dest = ctx->alt_dv_zero_n;                                   // kDvSkip, kDvVprev: the destination's dv rows were not touched
    if (ctx->compact_dv_mode == 0 /* kDvMove */)
        dest = ctx->dv_zero == 1 ? (dest > alive ? dest : alive) : 0;    // zeros (or not) were moved into [0, alive)
    if (ctx->compact_dv_mode == 1 /* kDvVprev */ && ctx->dv_zero == 1 && dest < alive) ctx->dv_zero = 0;
    ctx->alt_dv_zero_n = left_behind;
    ctx->dv_zero_n = ctx->dv_zero == 1 ? dest : 0;
    ctx->compact_dv_mode = 0;
    std::swap(ctx->slab, ctx->slab_alt);
    refresh_rows(ctx);
    std::swap(ctx->ids, ctx->ids_alt);
    if (ctx->kind) std::swap(ctx->kind, ctx->kind_alt);
    ctx->ids_iota = false;
    ctx->count = alive;
    ctx->last_delete_n = last_delete_n;
    ctx->last_delete_masked = false; // (the alive path sets it again behind this call)
    ctx->lam4_valid = false;
}

// ---- typed implementations of the Level-2 steps ------------------------------------------------------
template <typename T>
int step_newton_t(pcl_ctx *ctx, double dt) {
    const int64_t N = ctx->count;
    newton_args<T> a{};
    for (int k = 0; k < 3; ++k) {
        a.v[k] = F<T>(ctx, PCL_V0 + k);
        a.r[k] = F<T>(ctx, PCL_R0 + k);
        a.dr[k] = F<T>(ctx, PCL_DR0 + k);
    }
    a.dt = (T)dt;
    a.N = N;
    a.ts = tile_stride(ctx);
    const int ps = prof_begin(ctx, PCL_PROF_NEWTON);
    hipLaunchKernelGGL(k_newton<T>, dim3(grid_for(ctx, div_up(N, pcl_rt<T>::VEC), kBlock)), dim3(kBlock), 0, ctx->stream, a);
    prof_end(ctx, ps);
    return launch_check("k_newton");
}

template <typename T>
int fill_photons_t(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, double e_min, double e_max, uint64_t seed) {
    fill_args<T> a{};
    for (int f = 0; f < PCL_NFIELDS; ++f) a.f[f] = F<T>(ctx, f);
    a.n = n;
    a.id_base = id_base;
    a.ts = tile_stride(ctx);
    a.c = c;
    a.e_min = e_min;
    a.e_max = e_max;
    a.seed = seed;
    hipLaunchKernelGGL(k_fill_photons<T>, dim3(grid_for(ctx, n, kBlock)), dim3(kBlock), 0, ctx->stream, a);
    return launch_check("k_fill_photons");
}

template <typename T>
int fill_table_t(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, const double *cdf_dev, const double *grid_dev,
                 int nbins, uint64_t seed) {
    fill_table_args<T> a{};
    for (int f = 0; f < PCL_NFIELDS; ++f) a.f[f] = F<T>(ctx, f);
    a.cdf = cdf_dev;
    a.grid = grid_dev;
    a.nbins = nbins;
    a.n = n;
    a.id_base = id_base;
    a.ts = tile_stride(ctx);
    a.c = c;
    a.seed = seed;
    hipLaunchKernelGGL(k_fill_table<T>, dim3(grid_for(ctx, n, kBlock)), dim3(kBlock), 0, ctx->stream, a);
    return launch_check("k_fill_table");
}

template <typename T>
int step_scatter_t(pcl_ctx *ctx, double A, double n, bool use_e, bool var_n, rtc_entry *ent, double c, double h,
                   int rng_mode, uint64_t seed, uint32_t step, bool py_dv) {
    const int64_t N = ctx->count;
    pcl_scatter_args<T> a{};
    a.d0 = F<T>(ctx, PCL_DR0); a.d1 = F<T>(ctx, PCL_DR1); a.d2 = F<T>(ctx, PCL_DR2);
    a.E = F<T>(ctx, PCL_E);
    a.r0 = F<T>(ctx, PCL_R0); a.r1 = F<T>(ctx, PCL_R1); a.r2 = F<T>(ctx, PCL_R2);
    a.v0 = F<T>(ctx, PCL_V0); a.v1 = F<T>(ctx, PCL_V1); a.v2 = F<T>(ctx, PCL_V2);
    a.dv0 = F<T>(ctx, PCL_DV0); a.dv1 = F<T>(ctx, PCL_DV1); a.dv2 = F<T>(ctx, PCL_DV2);
    a.rtheta = static_cast<const T *>(ctx->rnd[0]);
    a.rphi = static_cast<const T *>(ctx->rnd[1]);
    a.rand = static_cast<const T *>(ctx->rnd[2]);
    a.ids = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->ids);
    a.kind = ctx->kind;
    a.hits = reinterpret_cast<pcl_u64 *>(ctx->d_cnt);
    a.id_base = ctx->id_base;
    a.N = N;
    a.ts = tile_stride(ctx);
    a.A = (T)A; a.n = (T)n; a.c = (T)c; a.h = (T)h;
    a.seed = seed;
    a.step = step;
    a.rng_mode = rng_mode;
    a.py_dv = py_dv ? 1 : 0;
    const int grid = grid_for(ctx, N, kBlock * PCL_SCATTER_ROWS);
    const int d = sizeof(T) == 8 ? 0 : 1;
    const int ps = prof_begin(ctx, PCL_PROF_SCATTER);
    set_np(a.np, ent);
    if (var_n && ent->module) {
        PCL_TRY(launch_module(ctx, ent->scatter[d][use_e ? 1 : 0], grid, a, "scatter_isotropic (hipRTC)"));
    } else {
        PCL_AOT_LAUNCH(k_scatter, T, use_e, var_n, grid, a);
        PCL_TRY(launch_check("k_scatter"));
    }
    prof_end(ctx, ps);
    return PCL_OK;
}

template <typename T>
int ensure_lam4_t(pcl_ctx *ctx, double h, double c) {
    if (ctx->lam4_valid && ctx->lam4_h == h && ctx->lam4_c == c) return PCL_OK;
    const int64_t n = ctx->holes ? ctx->slots : ctx->count; // (behind an alive mask: every slot of the extent)
    hipLaunchKernelGGL(k_lam4<T>, dim3(grid_for(ctx, n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const T *)F<T>(ctx, PCL_E), static_cast<T *>(ctx->lam4), (T)h, (T)c, n, tile_stride(ctx));
    PCL_TRY(launch_check("k_lam4"));
    ctx->lam4_valid = true;
    ctx->lam4_h = h;
    ctx->lam4_c = c;
    return PCL_OK;
}

template <typename T>
int step_fast_t(pcl_ctx *ctx, double dt, double A, double n, bool use_e, bool var_n, rtc_entry *ent, double c, double h,
                uint64_t seed, uint32_t step) {
    const int64_t N = ctx->holes ? ctx->slots : ctx->count; // behind an alive mask the kernel sweeps the extent
    if (use_e) PCL_TRY(ensure_lam4_t<T>(ctx, h, c));
    pcl_fast_args<T> f{};
    f.r0 = F<T>(ctx, PCL_R0); f.r1 = F<T>(ctx, PCL_R1); f.r2 = F<T>(ctx, PCL_R2);
    f.vi0 = F<T>(ctx, PCL_V0); f.vi1 = F<T>(ctx, PCL_V1); f.vi2 = F<T>(ctx, PCL_V2);
    f.vo0 = static_cast<T *>(ctx->vprev[0]); f.vo1 = static_cast<T *>(ctx->vprev[1]); f.vo2 = static_cast<T *>(ctx->vprev[2]);
    f.lam4 = static_cast<const T *>(ctx->lam4);
    f.E = F<T>(ctx, PCL_E);
    f.cnt = reinterpret_cast<pcl_u64 *>(ctx->cnt_target);
    f.id_base = ctx->id_base;
    f.N = N;
    f.ts = tile_stride(ctx);
    f.dt = (T)dt; f.A = (T)A; f.n = (T)n; f.c = (T)c;
    f.seed = seed;
    f.step = step;
    // a store that has been compacted (explicit ids) or holds plain Objects takes the GEN variant: same arithmetic,
    // two more streams (8 B of id, 1 B of kind per particle)
    f.ids = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->ids);
    f.kind = ctx->kind;
    if (ctx->holes) { // the store keeps removed photons' slots: the kernel reads the alive bits, and r catches up first
        f.alive = reinterpret_cast<const pcl_u64 *>(ctx->masks);
        f.n_pend = ctx->pend_n;
        for (int q = 0; q < ctx->pend_n; ++q) f.pend_dt[q] = (T)ctx->pend_dt[q], f.pend_rep[q] = ctx->pend_rep[q];
    }
    const bool gen = f.ids || f.kind || f.alive;
    const int grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC), kBlock);
    const int d = sizeof(T) == 8 ? 0 : 1;
    const int ps = prof_begin(ctx, PCL_PROF_FUSED);
    set_np(f.np, ent);
    if (var_n && ent->module) {
        PCL_TRY(launch_module(ctx, (gen ? ent->fastg : ent->fast)[d][use_e ? 1 : 0], grid, f, "step_fused fast path (hipRTC)"));
    } else {
        if (gen)
            PCL_AOT_LAUNCH_SHAPED(k_fastg, T, use_e, var_n, grid, f);
        else
            PCL_AOT_LAUNCH_SHAPED(k_fast, T, use_e, var_n, grid, f);
        PCL_TRY(launch_check("k_fast"));
    }
    prof_end(ctx, ps);
    if (ctx->holes) ctx->pend_n = 0; // the kernel wrote r with every pending move (and this step's) applied
    return PCL_OK;
}

template <typename T>
int step_multi_t(pcl_ctx *ctx, double dt, int k_steps, double A, double n, bool use_e, bool var_n, rtc_entry *ent, double c,
                 double h, uint64_t seed, uint32_t step, const double *planes_host, int n_planes) {
    const int64_t N = ctx->count;
    if (use_e) PCL_TRY(ensure_lam4_t<T>(ctx, h, c));
    pcl_multi_args<T> f{};
    f.r0 = F<T>(ctx, PCL_R0); f.r1 = F<T>(ctx, PCL_R1); f.r2 = F<T>(ctx, PCL_R2);
    f.v0 = F<T>(ctx, PCL_V0); f.v1 = F<T>(ctx, PCL_V1); f.v2 = F<T>(ctx, PCL_V2);
    f.vp0 = static_cast<T *>(ctx->vprev[0]); f.vp1 = static_cast<T *>(ctx->vprev[1]); f.vp2 = static_cast<T *>(ctx->vprev[2]);
    f.lam4 = static_cast<const T *>(ctx->lam4);
    f.E = F<T>(ctx, PCL_E);
    f.cnt = reinterpret_cast<pcl_u64 *>(ctx->d_multi);
    f.id_base = ctx->id_base;
    f.N = N;
    f.ts = tile_stride(ctx);
    f.dt = (T)dt; f.A = (T)A; f.n = (T)n; f.c = (T)c;
    f.seed = seed;
    f.step = step;
    f.K = k_steps;
    f.n_planes = n_planes > 0 ? n_planes : 0;
    for (int p = 0; p < f.n_planes; ++p) {
        const double *loc = planes_host + 3 * p;
        const int ax = !std::isnan(loc[0]) ? 0 : (!std::isnan(loc[1]) ? 1 : 2); // light.py:385-396
        f.plane_ax[p] = ax;
        f.plane_L[p] = (T)loc[ax];
    }
    int grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC), kBlock);
    const int d = sizeof(T) == 8 ? 0 : 1;
    const int ps = prof_begin(ctx, PCL_PROF_MULTI);
    // 256 photons per wave (NQ = 2 in fp64): faster than 128 at every hit fraction since the velocities live in LDS
    // (pcl_multi_body_lds).  PCL_MULTI_NQ2=0 takes the 128-photon instantiation (hipRTC, fp64), for A/B runs and the tests.
    static knob k_nq2("PCL_MULTI_NQ2");
    const bool nq2 = !(k_nq2.set() && k_nq2.off());
    set_np(f.np, ent);
    ctx->multi_work[2] = 64 * pcl_rt<T>::VEC;
    // The saturation probe (pcl_n_expr_sat; 128-photon form, fp64): worth its dozen instructions per wave-step only where
    // exp's arguments are out of range for (nearly) whole waves.  The kernel says how many wave-steps took the shortcut;
    // the host tries the probing variant on a store's second launch, keeps it while more than half of them did, and
    // tries again every 16th launch otherwise.  PCL_MULTI_SAT=1 always, =0 never.
    static knob k_sat("PCL_MULTI_SAT");
    const int sat_mode = !k_sat.set() ? -1 : (k_sat.off() ? 0 : 1);
    bool sat = false;
    if (var_n && ent->module && d == 0 && ent->multis[use_e ? 1 : 0] && ent->multi2s[use_e ? 1 : 0]) {
        const bool due = ctx->multi_sat_on || ctx->multi_sat_next <= 0;
        sat = sat_mode == 1 || (sat_mode == -1 && ctx->multi_launches >= 1 && due);
    }
    ctx->multi_sat_used = sat;
    ++ctx->multi_launches;
    // 192 photons per wave (three per lane) for a launch that STARTS at a hit fraction between 0.28 and 0.355 -- about 0.25 to
    // 0.32 over its steps: a wave of 256 photons then queues 64 to 82 hits per step, a full dense pass and a nearly empty one,
    // where 192 photons queue 48 to 62, one pass (expected dense passes per photon, binomial: h = 0.27: 1.03 / 192 against
    // 1.74 / 256; at h <= 0.234 and h >= 0.34 the 256-photon form is as good or better).  The previous launch's last step is
    // (nearly) this launch's first.  PCL_MULTI_NQ3=1 always, =0 never.
    static knob k_nq3("PCL_MULTI_NQ3");
    const int nq3_mode = !k_nq3.set() ? -1 : (k_nq3.off() ? 0 : 1);
    const bool have3 = var_n && ent->module && d == 0 && ent->multi3[use_e ? 1 : 0] && ent->multi3s[use_e ? 1 : 0];
    const bool nq3 = have3 && nq2 && (nq3_mode == 1 || (nq3_mode == -1 && ctx->multi_last_h >= 0.28 && ctx->multi_last_h < 0.355));
    if (nq3) {
        ctx->multi_work[2] = 192;
        grid = grid_for(ctx, div_up(N, (int64_t)3), kBlock);
        PCL_TRY(launch_module(ctx, (sat ? ent->multi3s : ent->multi3)[use_e ? 1 : 0], grid, f, "step_fused_multi, 192 photons per wave (hipRTC)"));
    } else if (sat && nq2) {
        ctx->multi_work[2] = 128 * pcl_rt<T>::VEC;
        grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC * 2), kBlock);
        PCL_TRY(launch_module(ctx, ent->multi2s[use_e ? 1 : 0], grid, f, "step_fused_multi NQ=2 with the saturation probe (hipRTC)"));
    } else if (sat) {
        PCL_TRY(launch_module(ctx, ent->multis[use_e ? 1 : 0], grid, f, "step_fused_multi with the saturation probe (hipRTC)"));
    } else if (var_n && ent->module && nq2 && d == 0 && ent->multi2[use_e ? 1 : 0]) {
        ctx->multi_work[2] = 128 * pcl_rt<T>::VEC;
        grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC * 2), kBlock);
        PCL_TRY(launch_module(ctx, ent->multi2[use_e ? 1 : 0], grid, f, "step_fused_multi NQ=2 (hipRTC)"));
    } else if (var_n && ent->module) {
        PCL_TRY(launch_module(ctx, ent->multi[d][use_e ? 1 : 0], grid, f, "step_fused_multi (hipRTC)"));
    } else if (!var_n && d == 0 && nq2 && nq3_mode != 0 &&
               (nq3_mode == 1 || (!use_e ? (A * n * c * dt >= 0.25 && A * n * c * dt < 0.333)           // constant n: known before the first launch
                                          : (ctx->multi_last_h >= 0.28 && ctx->multi_last_h < 0.355)))) { // wavelength term: by the launch before
        ctx->multi_work[2] = 192;
        grid = grid_for(ctx, div_up(N, (int64_t)3), kBlock);
        pcl_multi_args<double> &f64 = reinterpret_cast<pcl_multi_args<double> &>(f); // (d == 0: T is double)
        if (use_e) hipLaunchKernelGGL(k_multi3_e1, dim3(grid), dim3(kBlock), 0, ctx->stream, f64);
        else hipLaunchKernelGGL(k_multi3_e0, dim3(grid), dim3(kBlock), 0, ctx->stream, f64);
        PCL_TRY(launch_check("k_multi3"));
    } else {
        ctx->multi_work[2] = 64 * pcl_rt<T>::VEC * kMultiNQ<T>;
        grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC * kMultiNQ<T>), kBlock);
        PCL_AOT_LAUNCH_SHAPED(k_multi, T, use_e, var_n, grid, f);
        PCL_TRY(launch_check("k_multi"));
    }
    prof_end(ctx, ps);
    return PCL_OK;
}

template <typename T>
int step_fused_t(pcl_ctx *ctx, double dt, bool do_scatter, double A, double n, bool use_e, bool var_n, rtc_entry *ent,
                 double c, double h, int rng_mode, uint64_t seed, uint32_t step, const double *planes_host, int n_planes,
                 bool lazy) {
    const int64_t N = ctx->count;
    const int np = n_planes > 0 ? n_planes : 0;
    pcl_fused_args<T> a{};
    a.r0 = F<T>(ctx, PCL_R0); a.r1 = F<T>(ctx, PCL_R1); a.r2 = F<T>(ctx, PCL_R2);
    a.vi0 = F<T>(ctx, PCL_V0); a.vi1 = F<T>(ctx, PCL_V1); a.vi2 = F<T>(ctx, PCL_V2);
    a.vo0 = F<T>(ctx, PCL_V0); a.vo1 = F<T>(ctx, PCL_V1); a.vo2 = F<T>(ctx, PCL_V2);
    if (lazy && do_scatter) {
        a.vo0 = static_cast<T *>(ctx->vprev[0]); a.vo1 = static_cast<T *>(ctx->vprev[1]); a.vo2 = static_cast<T *>(ctx->vprev[2]);
    }
    a.lazy = lazy ? 1 : 0;
    a.dr0 = F<T>(ctx, PCL_DR0); a.dr1 = F<T>(ctx, PCL_DR1); a.dr2 = F<T>(ctx, PCL_DR2);
    a.dv0 = F<T>(ctx, PCL_DV0); a.dv1 = F<T>(ctx, PCL_DV1); a.dv2 = F<T>(ctx, PCL_DV2);
    a.E = F<T>(ctx, PCL_E);
    a.rtheta = static_cast<const T *>(ctx->rnd[0]);
    a.rphi = static_cast<const T *>(ctx->rnd[1]);
    a.rand = static_cast<const T *>(ctx->rnd[2]);
    a.ids = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->ids);
    a.kind = ctx->kind;
    a.cnt = reinterpret_cast<pcl_u64 *>(ctx->cnt_target);
    a.id_base = ctx->id_base;
    a.N = N;
    a.ts = tile_stride(ctx);
    a.dt = (T)dt; a.A = (T)A; a.n = (T)n; a.c = (T)c; a.h = (T)h;
    a.seed = seed;
    a.step = step;
    a.rng_mode = rng_mode;
    a.do_scatter = do_scatter ? 1 : 0;
    a.n_planes = n_planes;
    for (int p = 0; p < np; ++p) {
        const double *loc = planes_host + 3 * p;
        const int ax = !std::isnan(loc[0]) ? 0 : (!std::isnan(loc[1]) ? 1 : 2); // light.py:385-396
        a.plane_ax[p] = ax;
        a.plane_L[p] = (T)loc[ax];
    }
    const int grid = grid_for(ctx, div_up(N, pcl_rt<T>::VEC), kBlock);
    const int d = sizeof(T) == 8 ? 0 : 1;
    const int ps = prof_begin(ctx, PCL_PROF_FUSED);
    set_np(a.np, ent);
    if (var_n && ent->module) {
        PCL_TRY(launch_module(ctx, ent->fused[d][use_e ? 1 : 0], grid, a, "step_fused (hipRTC)"));
    } else {
        PCL_AOT_LAUNCH(k_fused, T, use_e, var_n, grid, a);
        PCL_TRY(launch_check("k_fused"));
    }
    prof_end(ctx, ps);
    return PCL_OK;
}

template <typename T>
int delete_mask_t(pcl_ctx *ctx, double A, double n, int rng_mode, uint64_t seed, uint32_t step) {
    const int64_t N = ctx->count;
    const int tiles = (int)div_up(N, kTile);
    delmask_args<T> m{};
    m.d0 = F<T>(ctx, PCL_DR0); m.d1 = F<T>(ctx, PCL_DR1); m.d2 = F<T>(ctx, PCL_DR2);
    m.rand = static_cast<const T *>(ctx->rnd[2]);
    m.ids = ctx->ids_iota ? nullptr : ctx->ids;
    m.kind = ctx->kind;
    m.masks = ctx->masks;
    m.tile_keep = ctx->tile_keep;
    m.id_base = ctx->id_base;
    m.N = N;
    m.ts = tile_stride(ctx);
    m.An = (T)A * (T)n; // one IEEE multiply in the store's precision == the kernel's (A * n)
    m.seed = seed;
    m.step = step;
    m.rng_mode = rng_mode;
    const int ps = prof_begin(ctx, PCL_PROF_DELETE_MASK);
    hipLaunchKernelGGL(k_delete_mask<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    prof_end(ctx, ps);
    return launch_check("k_delete_mask");
}

template <typename T>
int counters_t(pcl_ctx *ctx, const double *planes_host, int n_planes) {
    const int64_t N = ctx->count;
    counter_args<T> a{};
    for (int k = 0; k < 3; ++k) {
        a.v[k] = F<T>(ctx, PCL_V0 + k);
        a.r[k] = F<T>(ctx, PCL_R0 + k);
        a.dr[k] = F<T>(ctx, PCL_DR0 + k);
    }
    for (int p = 0; p < n_planes; ++p) {
        // first non-NaN of x, y decides the axis, else z                            light.py:385-396
        const double *loc = planes_host + 3 * p;
        const int ax = !std::isnan(loc[0]) ? 0 : (!std::isnan(loc[1]) ? 1 : 2);
        a.plane_ax[p] = ax;
        a.plane_L[p] = (T)loc[ax];
    }
    a.n_planes = n_planes;
    a.N = N;
    a.ts = tile_stride(ctx);
    a.out = ctx->d_cnt + 1;
    const int ps = prof_begin(ctx, PCL_PROF_COUNTERS);
    hipLaunchKernelGGL(k_counters<T>, dim3(grid_for(ctx, N, kBlock)), dim3(kBlock), 0, ctx->stream, a);
    prof_end(ctx, ps);
    return launch_check("k_counters");
}

// Fields a compaction has to move.  r, v and E always; dr unless it is implicit after the call (dr = v*dt); dv -- or,
// while dv is implicit (dv = v - vprev, all-photon stores), the vprev rows in its place -- or nothing, when every dv
// element is known to be +0.0 and the destination's dv rows are too.  Order: r v [dr] [dv | vprev] E
// (the compaction kernels read r, v and dr at fixed positions for the measure counters).
enum { kDvMove = 0, kDvVprev = 1, kDvSkip = 2 };
int compact_fields(pcl_ctx *ctx, compact_args &ca, bool move_dr, int dv_mode) {
    const bool dv_implicit = dv_mode == kDvVprev;
    int nf = 0;
    for (int f = PCL_R0; f <= PCL_V2; ++f) {
        ca.src[nf] = ctx->field[f];
        ca.dst[nf++] = ctx->field_alt[f];
    }
    if (move_dr)
        for (int f = PCL_DR0; f <= PCL_DR2; ++f) {
            ca.src[nf] = ctx->field[f];
            ca.dst[nf++] = ctx->field_alt[f];
        }
    for (int k = 0; k < 3 && dv_mode != kDvSkip; ++k) {
        ca.src[nf] = dv_implicit ? ctx->vprev[k] : ctx->field[PCL_DV0 + k];
        ca.dst[nf++] = dv_implicit ? ctx->vprev_alt[k] : ctx->field_alt[PCL_DV0 + k];
    }
    ca.src[nf] = ctx->field[PCL_E];
    ca.dst[nf++] = ctx->field_alt[PCL_E];
    ca.ids_src = ctx->ids_iota ? nullptr : ctx->ids;
    ca.ids_dst = ctx->ids_alt;
    ca.ksrc = ctx->kind;
    ca.kdst = ctx->kind ? ctx->kind_alt : nullptr;
    ca.masks = ctx->masks;
    ca.tile_off = ctx->tile_off;
    ca.id_base = ctx->id_base;
    ca.N = ctx->count;
    ca.ts = tile_stride(ctx);
    return nf;
}

template <typename T>
void plane_table(const double *planes_host, int n_planes, int *ax_out, T *L_out) {
    for (int p = 0; p < (n_planes > 0 ? n_planes : 0); ++p) {
        const double *loc = planes_host + 3 * p;
        const int ax = !std::isnan(loc[0]) ? 0 : (!std::isnan(loc[1]) ? 1 : 2); // light.py:385-396
        ax_out[p] = ax;
        L_out[p] = (T)loc[ax];
    }
}

// pass 3 of a delete pipeline: stable compaction of the store into the other slab (+ the measure counters when
// cc.n_planes >= 0); has_dr says whether the dr rows travel (13 fields) or stay implicit (10)
// PCL_COMPACT_SPARSE: waves of k_compact_count with at most this many survivors among their 512 slots take them survivor
// by survivor (default 256; 0 = always row by row, the round-3 form)
uint32_t compact_sparse_max() {
    static knob k("PCL_COMPACT_SPARSE");
    const double v = k.value(256.0);
    return v < 0 ? 0u : (v > 512.0 ? 512u : (uint32_t)v);
}

template <typename T>
int launch_compact_count(pcl_ctx *ctx, bool has_dr, int dv_mode, compact_counter_args<T> &cc) {
    const int tiles = (int)div_up(ctx->count, kTile);
    if (has_dr && dv_mode == kDvSkip) dv_mode = kDvMove; // (13-field form: nothing is skipped)
    // Not writing the survivors' dv needs a destination whose dv rows are zero already.  The first compaction of a run
    // therefore still moves them (zeros); from then on the slabs swap roles and the one left behind holds zeros over
    // [0, the larger, earlier count) -- adopt_compacted keeps the book.
    if (dv_mode == kDvSkip && ctx->alt_dv_zero_n < ctx->count) dv_mode = kDvMove;
    ctx->compact_dv_mode = dv_mode;
    compact_args ca{};
    compact_fields(ctx, ca, has_dr, dv_mode);
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type W;
    const int pc = prof_begin(ctx, PCL_PROF_COMPACT);
    // Two formulations, both enqueued; the scan's verdict (d_cnt[kCounterSlots - 2]) lets exactly one of them work.
    // Stores with kind bytes (plain Objects) always take the direct kernel -- the staged one does not move them.
    static const bool direct_only = getenv("PCL_COMPACT_DIRECT") != nullptr; // perf-experiment hook
    bool both = !ctx->kind && !direct_only && ctx->count >= ((int64_t)1 << 22); // (the scan's rule needs >= 4M particles)
    // ... unless the host knows the survivor count already (the compaction behind bodies worked out ahead): the scan's rule,
    // applied here, and ONE launch
    bool direct = true, staged = both;
    if (both && ctx->compact_known_alive >= 0) {
        staged = ctx->compact_known_alive * 100 > ctx->count * (int64_t)compact_lds_min_pct();
        direct = !staged;
        both = false;
    }
    ca.choice = both ? reinterpret_cast<const int *>(ctx->d_cnt + kCounterSlots - 2) : nullptr;
    ca.sparse_max = compact_sparse_max();
    if (direct) {
        if (has_dr)
            hipLaunchKernelGGL((k_compact_count<T, W, 13>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
        else if (dv_mode == kDvSkip)
            hipLaunchKernelGGL((k_compact_count<T, W, 7>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
        else
            hipLaunchKernelGGL((k_compact_count<T, W, 10>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
        PCL_TRY(launch_check("k_compact_count"));
    }
    if (staged) {
        if (has_dr)
            hipLaunchKernelGGL((k_compact_lds<T, W, 13>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
        else if (dv_mode == kDvSkip)
            hipLaunchKernelGGL((k_compact_lds<T, W, 7>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
        else
            hipLaunchKernelGGL((k_compact_lds<T, W, 10>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca, cc);
    }
    prof_end(ctx, pc);
    return launch_check("k_compact");
}

// How the dv rows go through the compaction of a lazy delete step.  A still-implicit dv (lazy scatter step: dv = v - vprev)
// stays implicit on all-photon stores -- the vprev rows travel in place of the dv rows -- and is made real otherwise; real
// dv rows that hold nothing but +0.0 (a run that never scatters; checked on the device once, then remembered) do not
// travel at all.
int decide_dv_mode(pcl_ctx *ctx, bool lazy, int *mode_out) {
    *mode_out = kDvMove;
    if (ctx->lazy_dv) {
        if (lazy && !ctx->kind)
            *mode_out = kDvVprev;
        else
            PCL_TRY(materialize(ctx));
        return PCL_OK;
    }
    static const bool no_skip = getenv("PCL_NO_DV_SKIP") != nullptr; // perf-experiment hook
    if (!lazy || ctx->kind || no_skip || ctx->count == 0) return PCL_OK;
    if (ctx->dv_zero == 0) {
        int *flag = reinterpret_cast<int *>(ctx->d_cnt + kCounterSlots - 5);
        PCL_HIP(hipMemsetAsync(flag, 0, sizeof(uint64_t), ctx->stream));
        const int grid = grid_for(ctx, ctx->count, kBlock);
        if (ctx->dtype == PCL_DTYPE_F64)
            hipLaunchKernelGGL(k_any_nonzero<uint64_t>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const uint64_t *)ctx->field[PCL_DV0],
                               (const uint64_t *)ctx->field[PCL_DV1], (const uint64_t *)ctx->field[PCL_DV2], ctx->count, tile_stride(ctx), flag);
        else
            hipLaunchKernelGGL(k_any_nonzero<uint32_t>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)ctx->field[PCL_DV0],
                               (const uint32_t *)ctx->field[PCL_DV1], (const uint32_t *)ctx->field[PCL_DV2], ctx->count, tile_stride(ctx), flag);
        PCL_TRY(launch_check("k_any_nonzero"));
        PCL_HIP(hipMemcpyAsync(ctx->h_cnt + kCounterSlots - 5, flag, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        set_dv_zero(ctx, ctx->h_cnt[kCounterSlots - 5] ? 2 : 1, ctx->count);
    }
    if (ctx->dv_zero == 1) *mode_out = kDvSkip;
    return PCL_OK;
}

// ---- the alive-mask delete path (k_delete_alive) ---------------------------------------------------------------------
// fewer than 1/kAliveRatio of the slots alive -> the next body compacts; stores below kAliveMinSlots never do (their
// kernels are latency, not bytes).  PCL_ALIVE=0 switches the path off (every body compacts, as in round 2).
bool alive_enabled() {
    static knob k("PCL_ALIVE");
    return !k.off();
}
double alive_ratio() {
    static knob k("PCL_ALIVE_RATIO");
    const double v = k.value(0.5);
    return v > 0.0 && v <= 1.0 ? v : 0.5;
}
int64_t alive_min_slots() {
    static knob k("PCL_ALIVE_MIN_SLOTS");
    const double v = k.value(65536.0);
    return (int64_t)(v >= 0 ? v : 65536);
}

// The list of moves r has not seen yet is run-length coded: a run's loop repeats one dt, so however many bodies r lags
// behind the list holds one entry; only kPendMax CHANGES of dt fill it (then a body's kernel writes r back).
bool pend_has_room(const pcl_ctx *ctx, double dt) {
    return ctx->pend_n < kPendMax || ctx->pend_dt[ctx->pend_n - 1] == dt;
}
void pend_push(pcl_ctx *ctx, double dt, int count) {
    if (count <= 0) return;
    if (ctx->pend_n > 0 && ctx->pend_dt[ctx->pend_n - 1] == dt) {
        ctx->pend_rep[ctx->pend_n - 1] += count;
        return;
    }
    ctx->pend_dt[ctx->pend_n] = dt;
    ctx->pend_rep[ctx->pend_n++] = count;
}
int pend_moves(const pcl_ctx *ctx) {
    int n = 0;
    for (int q = 0; q < ctx->pend_n; ++q) n += ctx->pend_rep[q];
    return n;
}

template <typename T>
void fill_alive_args(pcl_ctx *ctx, alive_args<T> &a, int64_t slots) {
    for (int k = 0; k < 3; ++k) {
        a.v[k] = F<T>(ctx, PCL_V0 + k);
        a.r[k] = F<T>(ctx, PCL_R0 + k);
    }
    a.ids = ctx->ids_iota ? nullptr : ctx->ids;
    a.masks = ctx->masks;
    a.masks_prev = ctx->masks_prev;
    a.tile_keep = ctx->tile_keep;
    a.acc = reinterpret_cast<unsigned long long *>(ctx->d_cnt + kCounterSlots);
    a.host = ctx->h_cnt;
    a.id_base = ctx->id_base;
    a.slots = slots;
    a.ts = tile_stride(ctx);
    a.n_pend = ctx->pend_n;
    for (int q = 0; q < ctx->pend_n; ++q) a.pend_dt[q] = (T)ctx->pend_dt[q], a.pend_rep[q] = ctx->pend_rep[q];
}

// r catches up with the moves it has not seen (the list is full and no compaction is due)
template <typename T>
int apply_pending_t(pcl_ctx *ctx) {
    alive_args<T> a{};
    fill_alive_args<T>(ctx, a, ctx->slots);
    hipLaunchKernelGGL(k_apply_pending<T>, dim3(grid_for(ctx, div_up(ctx->slots, 2), kBlock)), dim3(kBlock), 0, ctx->stream, a);
    PCL_TRY(launch_check("k_apply_pending"));
    ctx->pend_n = 0;
    return PCL_OK;
}

// one loop body on the alive mask.  count: totals and measure counters go to the host (a body without compaction);
// otherwise only the new masks and tile counts are produced (the compaction behind it counts)
template <typename T>
int delete_alive_t(pcl_ctx *ctx, int64_t slots, bool fresh, bool count, bool write_r, double dt, double A, double n, uint64_t seed,
                   uint32_t step, const double *planes_host, int n_planes) {
    alive_args<T> a{};
    fill_alive_args<T>(ctx, a, slots);
    a.dt = (T)dt;
    a.An = (T)A * (T)n;
    a.seed = seed;
    a.step = step;
    a.fresh = fresh ? 1 : 0;
    a.write_r = write_r ? 1 : 0;
    a.count = count ? 1 : 0;
    a.n_planes = n_planes;
    a.seq = count ? ++ctx->alive_seq : 0;
    a.zero_cnt = ctx->d_cnt + 1;
    a.n_zero = 3 + (n_planes > 0 ? n_planes : 0);
    plane_table<T>(planes_host, n_planes, a.plane_ax, a.plane_L);
    a.r_axes = write_r ? 7 : 0; // a plane counter needs r along its own axis only (8 instead of 24 B per slot for one plane)
    for (int p = 0; count && p < n_planes; ++p) a.r_axes |= 1 << a.plane_ax[p];
    // A workgroup's sums reach the grid totals with one atomic per counter, so the grid is capped (64 workgroups per
    // CU, PCL_ALIVE_WG_PER_CU) and workgroups walk the tiles with the grid's stride -- every one of them the SAME number
    // of tiles (k = ceil(tiles / cap), grid = ceil(tiles / k)): a grid just below the tile count leaves a second, nearly
    // empty round (measured at 16761 tiles: 500 us with 16384 workgroups, 390 us with 4096).
    static const int wg_per_cu = [] {
        const char *e = getenv("PCL_ALIVE_WG_PER_CU");
        const int v = e ? atoi(e) : 64;
        return v > 0 ? v : 64;
    }();
    const int64_t n_tiles = div_up(slots, kTile), cap = (int64_t)ctx->prop.multiProcessorCount * wg_per_cu;
    const int64_t grid = div_up(n_tiles, div_up(n_tiles, cap));
    const int ps = prof_begin(ctx, PCL_PROF_DELETE_MASK);
    if ((count && n_planes > 0) || write_r)
        hipLaunchKernelGGL((k_delete_alive<T, true>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((k_delete_alive<T, false>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    prof_end(ctx, ps);
    return launch_check("k_delete_alive");
}

// scan + stable compaction of a store in the alive-mask state into the other slab: the survivors' r catches up with the
// pending moves (and, ``move``, with this body's) on the way.  The caller adopts the result.
template <typename T>
int compact_alive_t(pcl_ctx *ctx, int dv_mode, bool move, double dt, const double *planes_host, int n_planes) {
    PCL_TRY(scan_tiles(ctx, ctx->count));
    compact_counter_args<T> cc{};
    cc.cnt = ctx->d_cnt;
    cc.n_planes = n_planes;
    cc.dt = (T)dt;
    cc.move = move ? 1 : 0;
    cc.n_pend = ctx->pend_n;
    for (int q = 0; q < ctx->pend_n; ++q) cc.pend_dt[q] = (T)ctx->pend_dt[q], cc.pend_rep[q] = ctx->pend_rep[q];
    plane_table<T>(planes_host, n_planes, cc.plane_ax, cc.plane_L);
    return launch_compact_count<T>(ctx, false, dv_mode, cc);
}

// The host side of k_delete_alive's report: poll the pinned block for the launch's sequence number (written after the
// totals, system-scope release) instead of waiting for the stream -- a loop body of a small store is a 5 us kernel, and
// the completion signal's way to a blocked host thread takes longer than that.  The stream is asked now and then, so a
// failed launch ends the wait with its error (PCL_ALIVE_POLL=0: hipStreamSynchronize).
// after a failed launch: ticket and sums of k_delete_alive / k_delete_ahead back to zero (best effort, on the stream)
void reset_alive_acc(pcl_ctx *ctx) {
    (void)hipMemsetAsync(ctx->d_cnt + kCounterSlots, 0, (size_t)kAccSlots * sizeof(uint64_t), ctx->stream);
    if (ctx->ahead_acc) (void)hipMemsetAsync(ctx->ahead_acc, 0, (size_t)kAheadAcc * sizeof(unsigned long long), ctx->stream);
}

int wait_alive_inner(pcl_ctx *ctx, int64_t before, int64_t *alive_out);
int wait_alive(pcl_ctx *ctx, int64_t before, int64_t *alive_out) {
    const int rc = wait_alive_inner(ctx, before, alive_out);
    if (rc != PCL_OK) reset_alive_acc(ctx);
    return rc;
}

int wait_alive_inner(pcl_ctx *ctx, int64_t before, int64_t *alive_out) {
    static knob k_poll("PCL_ALIVE_POLL");
    const bool poll = !k_poll.off();
    volatile uint64_t *seq = ctx->h_cnt + kCounterSlots - 6;
    if (!poll) {
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        if (*seq != ctx->alive_seq) return fail(PCL_ERR_HIP, "k_delete_alive did not report");
    } else {
        for (uint64_t spins = 1; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != ctx->alive_seq; ++spins) {
            if ((spins & 0x3FFF) == 0) {
                const hipError_t e = hipStreamQuery(ctx->stream);
                if (e == hipSuccess) { // the kernel is over: its report must be there
                    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != ctx->alive_seq) return fail(PCL_ERR_HIP, "k_delete_alive did not report");
                    break;
                }
                if (e != hipErrorNotReady) return fail(PCL_ERR_HIP, "k_delete_alive failed: %s", hipGetErrorString(e));
            }
        }
    }
    const int64_t alive = (int64_t)ctx->h_cnt[kCounterSlots - 1];
    if (alive < 0 || alive > before)
        return fail(PCL_ERR_HIP, "the alive mask holds an impossible count %lld of %lld", (long long)alive, (long long)before);
    *alive_out = alive;
    return PCL_OK;
}

// ---- delete loop bodies ahead of their calls (k_delete_ahead) ---------------------------------------------------------
// PCL_AHEAD=0 switches it off; PCL_AHEAD_K bodies per launch (default 64 = kAheadMax); extents up to
// PCL_AHEAD_MAX_SLOTS (default 2^22) take the small stores' form (r written at the commit); above it PCL_AHEAD_K_BIG bodies at
// most (default 64; 0 or 1 = none).
// Round 5: with k_delete_ahead_live a body costs what the photons still alive cost, and a delete loop runs until its store is
// empty, so ONE launch for (nearly) the whole run beats several with a commit, a compaction and a host round trip each between
// them.  Delete-until-empty at a survival rate of 0.70 per body, one call per body, ms per run (same box, tools/ab_knob.sh,
// profiles/r05_ab_ahead_k.log): bodies per launch 16 / 32 / 48 / 64 -- 1e8 photons 2.21 / 2.14 / 2.09 / 1.90, 1e7 photons
// 0.54 / 0.49 / 0.50 / 0.425; small stores 24 / 48 / 64 -- 3e6 photons 0.435 / 0.39 / 0.30, 1e6 0.304 / 0.218 / 0.215, 2e5
// 0.26 / 0.185 / 0.185.  (Rounds 3-4, whose kernel gave every slot a lane through every body, had measured 12 to 24 as best.)
int ahead_k() {
    static knob k_on("PCL_AHEAD"), k_k("PCL_AHEAD_K");
    if (k_on.off()) return 0;
    const int k = (int)k_k.value((double)kAheadMax);
    return k < 2 ? 0 : (k > kAheadMax ? kAheadMax : k);
}
int64_t ahead_max_slots() {
    static knob k("PCL_AHEAD_MAX_SLOTS");
    const double v = k.value(4194304.0);
    return (int64_t)(v > 0 ? v : 0);
}
int ahead_k_big(int64_t slots) {
    static knob k("PCL_AHEAD_K_BIG");
    (void)slots;
    const int v = (int)k.value((double)kAheadMax);
    return v < 0 ? 0 : (v > kAheadMax ? kAheadMax : v);
}

bool ahead_same_call(const pcl_ctx::ahead_state &s, double dt, double A, double n, uint64_t seed, const double *planes_host, int n_planes) {
    if (s.dt != dt || s.A != A || s.n != n || s.seed != seed || s.n_planes != n_planes) return false;
    return n_planes <= 0 || memcmp(s.planes, planes_host, (size_t)n_planes * 3 * sizeof(double)) == 0; // (bitwise: NaN marks the free axes)
}

template <typename T>
void fill_ahead_args(pcl_ctx *ctx, ahead_args<T> &a) {
    const pcl_ctx::ahead_state &s = ctx->ahead;
    for (int k = 0; k < 3; ++k) {
        a.v[k] = F<T>(ctx, PCL_V0 + k);
        a.r[k] = F<T>(ctx, PCL_R0 + k);
    }
    a.ids = ctx->ids_iota ? nullptr : ctx->ids;
    a.masks = ctx->masks;
    a.death = ctx->ahead_death;
    a.acc = ctx->ahead_acc;
    a.host = ctx->ahead_host;
    a.id_base = ctx->id_base;
    a.slots = s.slots;
    a.ts = tile_stride(ctx);
    a.dt = (T)s.dt;
    a.An = (T)s.A * (T)s.n;
    a.n_pend = ctx->pend_n;
    for (int q = 0; q < ctx->pend_n; ++q) a.pend_dt[q] = (T)ctx->pend_dt[q], a.pend_rep[q] = ctx->pend_rep[q];
    a.seed = s.seed;
    a.step0 = s.step0;
    a.K = s.K;
    a.fresh = s.fresh ? 1 : 0;
    a.n_planes = s.n_planes;
    plane_table<T>(s.planes, s.n_planes, a.plane_ax, a.plane_L);
    a.r_axes = 0;
    for (int p = 0; p < s.n_planes; ++p) a.r_axes |= 1 << a.plane_ax[p];
    a.masks_out = ctx->masks;
    a.masks_prev = ctx->masks_prev;
    a.tile_keep = ctx->tile_keep;
}

template <typename T>
int ahead_launch_t(pcl_ctx *ctx) {
    ahead_args<T> a{};
    fill_ahead_args<T>(ctx, a);
    a.seq = ++ctx->ahead_seq;
    // (workgroups walk the tiles with the grid's stride, every one of them the same number of tiles, as in delete_alive_t:
    // a workgroup's sums reach the grid totals with K x (4 + planes) atomics)
    const int64_t n_tiles = div_up(ctx->ahead.slots, kTile), cap = (int64_t)ctx->prop.multiProcessorCount * 64;
    const int64_t grid = div_up(n_tiles, div_up(n_tiles, cap));
    const int ps = prof_begin(ctx, PCL_PROF_DELETE_AHEAD);
    static knob k_live("PCL_AHEAD_LIVE"); // 0: always the slot-per-lane kernel
    for (int i = 0; i < kAheadWork; ++i) ctx->ahead_host[kAheadMax * kAheadRow + 1 + i] = 0; // (only the live kernel tallies its work)
    if (a.n_planes <= 1 && !k_live.off() && a.ids)
        hipLaunchKernelGGL((k_delete_ahead_live<T, true>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    else if (a.n_planes <= 1 && !k_live.off())
        hipLaunchKernelGGL((k_delete_ahead_live<T, false>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_delete_ahead<T>, dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    prof_end(ctx, ps);
    return launch_check("k_delete_ahead");
}

template <typename T>
int ahead_commit_t(pcl_ctx *ctx, bool write_r) {
    ahead_args<T> a{};
    fill_ahead_args<T>(ctx, a);
    a.j = ctx->ahead.used;
    const int64_t n_tiles = div_up(ctx->ahead.slots, kTile), cap = (int64_t)ctx->prop.multiProcessorCount * 64;
    const int64_t grid = div_up(n_tiles, div_up(n_tiles, cap));
    if (write_r) {
        hipLaunchKernelGGL((k_ahead_commit<T, true>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, a);
    } else { // big stores: masks and tile counts only, 16 slots per lane
        const int64_t n_groups = n_tiles * (kTile / 16);
        const int64_t blocks = div_up(n_groups, kBlock);
        const int64_t g2 = blocks < cap ? blocks : cap;
        hipLaunchKernelGGL(k_ahead_masks, dim3((unsigned)g2), dim3(kBlock), 0, ctx->stream, ctx->ahead_death, a.j, ctx->masks, ctx->masks_prev,
                           ctx->tile_keep, n_groups);
    }
    return launch_check("k_ahead_commit");
}

// make the state after the bodies handed out so far real (asynchronous: one small kernel on the stream)
int ahead_commit(pcl_ctx *ctx) {
    pcl_ctx::ahead_state &s = ctx->ahead;
    if (!s.active) return PCL_OK;
    if (s.used < s.K) ++ctx->ahead_missed;
    if (2 * s.used < s.K && ctx->count > 0) { // most of the launch's work was for nothing: pause, longer every time in a row
        ctx->ahead_backoff = ctx->ahead_backoff ? (ctx->ahead_backoff < 256 ? 2 * ctx->ahead_backoff : 256) : 4;
        ctx->ahead_wait = ctx->ahead_backoff;
    } else {
        ctx->ahead_backoff = 0;
    }
    const bool write_r = !s.big;
    PCL_TRY(PCL_DISPATCH(ctx, ahead_commit_t<double>(ctx, write_r), ahead_commit_t<float>(ctx, write_r)));
    s.active = false;
    ctx->holes = true;
    ctx->seg_prefix = false;
    ctx->slots = s.slots;
    if (write_r) {
        ctx->pend_n = 0; // the kernel wrote r with every move applied
    } else {
        pend_push(ctx, s.dt, s.used); // (room for the run was checked when the bodies were launched)
    }
    ctx->last_delete_masked = true;
    ctx->last_delete_slots = s.slots;
    // A big store whose alive photons have fallen below the compaction threshold is compacted NOW, from the masks just
    // written: the compacting body's own flag pass (a sweep of v over the whole extent) has already been made by
    // k_delete_ahead.  The survivors' r catches up with every pending move on the way (densify).
    if (s.big && ctx->count > 0 && s.slots > alive_min_slots() && (double)ctx->count < alive_ratio() * (double)s.slots) {
        const int64_t last_n = ctx->last_delete_n;
        PCL_TRY(densify(ctx));
        ctx->last_delete_n = last_n;
    }
    return PCL_OK;
}

int ahead_wait_inner(pcl_ctx *ctx);
int ahead_wait(pcl_ctx *ctx) {
    const int rc = ahead_wait_inner(ctx);
    if (rc != PCL_OK) reset_alive_acc(ctx);
    return rc;
}

int ahead_wait_inner(pcl_ctx *ctx) {
    volatile uint64_t *seq = ctx->ahead_host + kAheadMax * kAheadRow;
    for (uint64_t spins = 1; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != ctx->ahead_seq; ++spins) {
        if ((spins & 0x3FFF) == 0) {
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != ctx->ahead_seq) return fail(PCL_ERR_HIP, "k_delete_ahead did not report");
                break;
            }
            if (e != hipErrorNotReady) return fail(PCL_ERR_HIP, "k_delete_ahead failed: %s", hipGetErrorString(e));
        }
    }
    return PCL_OK;
}

// hand out the next of the bodies worked out ahead: count, counters (into h_cnt[1..], where pcl_step_fused_delete reads them)
int ahead_serve(pcl_ctx *ctx, int64_t *alive_out) {
    pcl_ctx::ahead_state &s = ctx->ahead;
    const uint64_t *row = ctx->ahead_host + (size_t)s.used * kAheadRow;
    const int64_t before = ctx->count, alive = (int64_t)row[0];
    if (alive < 0 || alive > before)
        return fail(PCL_ERR_HIP, "a body worked out ahead holds an impossible count %lld of %lld", (long long)alive, (long long)before);
    for (int k = 0; k < 3 + (s.n_planes > 0 ? s.n_planes : 0); ++k) ctx->h_cnt[1 + k] = row[1 + k];
    ++s.used;
    ++ctx->ahead_served;
    ctx->count = alive;
    ctx->last_delete_n = before;
    *alive_out = alive;
    if (s.used == s.K) PCL_TRY(ahead_commit(ctx));
    return PCL_OK;
}

int ahead_resources(pcl_ctx *ctx, int64_t slots) {
    if (!ctx->ahead_acc) {
        const size_t n = (size_t)kAheadAcc;
        PCL_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->ahead_acc), n * sizeof(unsigned long long)));
        PCL_HIP(hipMemsetAsync(ctx->ahead_acc, 0, n * sizeof(unsigned long long), ctx->stream));
        PCL_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->ahead_host), n * sizeof(uint64_t)));
        memset(ctx->ahead_host, 0, n * sizeof(uint64_t));
    }
    const int64_t need = div_up(slots, kTile) * kTile;
    if (need > ctx->ahead_cap) {
        if (ctx->ahead_death) {
            PCL_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->ahead_death);
            ctx->ahead_death = nullptr;
            ctx->ahead_cap = 0;
        }
        PCL_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->ahead_death), (size_t)need));
        ctx->ahead_cap = need;
    }
    return PCL_OK;
}

void drop_holes(pcl_ctx *ctx) {
    ctx->multi_last_h = ctx->mixed_last_h = -1.0; // (called whenever the population is replaced)
    ctx->multi_launches = 0;
    ctx->multi_sat_on = false;
    ctx->multi_sat_next = 0;
    ctx->ahead.active = false;
    ctx->ahead_last_valid = false;
    ctx->ahead_wait = ctx->ahead_backoff = 0;
    ctx->holes = false;
    ctx->seg_prefix = false;
    ctx->slots = 0;
    ctx->pend_n = 0;
    ctx->last_delete_masked = false;
}

// Make a store in the alive-mask state dense again (stable: the survivors keep their order), r up to date.  What every
// entry point other than the alive path sees.  dr stays implicit (= v * lazy_dt) as the last body left it.
int densify(pcl_ctx *ctx) {
    if (!ctx->holes) return PCL_OK;
    const int64_t alive = ctx->count, slots = ctx->slots;
    const int64_t last_n = ctx->last_delete_n;
    const bool last_masked = ctx->last_delete_masked;
    ctx->holes = false; // (ensure_alt and the helpers below must not come back here)
    ctx->count = slots; // the pipeline's launch geometry is the extent
    int rc = ensure_alt(ctx);
    int dv_mode = kDvMove;
    if (rc == PCL_OK) rc = decide_dv_mode(ctx, true, &dv_mode);
    if (rc == PCL_OK) {
        ctx->compact_known_alive = alive;
        rc = PCL_DISPATCH(ctx, compact_alive_t<double>(ctx, dv_mode, false, 0.0, nullptr, -1),
                          compact_alive_t<float>(ctx, dv_mode, false, 0.0, nullptr, -1));
        ctx->compact_known_alive = -1;
    }
    int64_t got = 0;
    if (rc == PCL_OK) rc = wait_count(ctx, slots, &got);
    if (rc == PCL_OK && got != alive)
        rc = fail(PCL_ERR_HIP, "compaction of the alive mask found %lld photons, the store holds %lld", (long long)got, (long long)alive);
    if (rc != PCL_OK) {
        ctx->holes = true;
        ctx->count = alive;
        return rc;
    }
    adopt_compacted(ctx, alive, last_n);
    ctx->last_delete_masked = last_masked; // the two mask arrays still describe the last delete
    ctx->slots = 0;
    ctx->pend_n = 0;
    ctx->seg_prefix = false;
    return PCL_OK;
}

// pcl_step_fused_delete on the alive mask (lazy, all photons, device RNG).  Returns the alive count in *alive_out; the
// measure counters are in h_cnt[1..] when n_planes >= 0.
// ``known``: how many bodies of this run the caller is certain to ask for, this one included (pcl_step_fused_delete_multi
// knows; 0 = a single call, the library goes by the pattern of the calls).
int fused_delete_alive(pcl_ctx *ctx, double dt, double A, double n, uint64_t seed, uint32_t step, const double *planes_host,
                       int n_planes, int64_t *alive_out, int known = 0) {
    const int64_t before = ctx->count;
    pcl_ctx::ahead_state &sp = ctx->ahead;
    if (sp.active) { // bodies were worked out ahead: is this the call they are waiting for?
        if (ahead_same_call(sp, dt, A, n, seed, planes_host, n_planes) && step == sp.step0 + (uint32_t)sp.used) {
            ctx->ahead_last.step0 = step;
            return ahead_serve(ctx, alive_out);
        }
        PCL_TRY(ahead_commit(ctx));
    }
    const bool fresh = !ctx->holes;
    const int64_t slots = fresh ? ctx->count : ctx->slots;
    const int np = n_planes > 0 ? n_planes : 0;
    PCL_TRY(ensure_scratch(ctx, slots, true));
    const bool compact_now = slots > alive_min_slots() && (double)before < alive_ratio() * (double)slots;
    // the same call as last time with the launch number advanced by one: a run's loop.  The FIRST delete body of a
    // population (no call to compare with: drop_holes forgets the last one whenever the particles are replaced) is taken for
    // the start of such a loop -- delete bodies come in loops (physicl/__init__.py:512-516) --; if the next call is not its
    // continuation the library has lost part of one sweep, pauses (ahead_commit), and goes by evidence from then on.
    const bool pattern = !ctx->ahead_last_valid ||
                         (ahead_same_call(ctx->ahead_last, dt, A, n, seed, planes_host, n_planes) && step == ctx->ahead_last.step0 + 1u);
    const bool repeat = pattern || known > 1;
    {
        pcl_ctx::ahead_state &l = ctx->ahead_last;
        l.dt = dt, l.A = A, l.n = n, l.seed = seed, l.n_planes = n_planes, l.step0 = step;
        if (n_planes > 0) memcpy(l.planes, planes_host, (size_t)n_planes * 3 * sizeof(double));
        ctx->ahead_last_valid = true;
    }
    int64_t alive = 0;
    if (ctx->ahead_wait > 0) --ctx->ahead_wait;
    // How many bodies ahead?  Small stores (extent <= PCL_AHEAD_MAX_SLOTS): PCL_AHEAD_K, their sweeps cost next to nothing.
    // Big stores: PCL_AHEAD_K_BIG (default 8; measured 3 .. 16 at 1e7 and 1e8 photons), if the list of
    // pending moves can take their run (r is not rewritten at a big store's commit).  One sweep of the extent then serves all
    // of them, and the compaction that has become due meanwhile runs from the committed masks, without a flag sweep of its own.
    int k_ahead = 0;
    bool big = false;
    if (!compact_now && repeat && ahead_k() > 0 && (ctx->ahead_wait == 0 || known > 1)) {
        // (a caller that has said how many bodies it wants gets those -- and the usual number when the pattern of the calls
        // promises more --, never fewer than two, never more than a launch holds or a big store's sweep is worth)
        const int by_pattern = pattern && ctx->ahead_wait == 0;
        if (slots <= ahead_max_slots()) {
            k_ahead = by_pattern ? ahead_k() : 0;
            if (known > 1 && known > k_ahead) k_ahead = known < kAheadMax ? known : kAheadMax;
        } else if (ahead_k_big(slots) > 1 && pend_has_room(ctx, dt)) {
            k_ahead = by_pattern || known >= ahead_k_big(slots) ? ahead_k_big(slots) : (known > 1 ? known : 0);
            big = k_ahead > 0;
        }
    }
    if (k_ahead > 0) {
        // predictable caller: this body and the next K - 1 in one launch that leaves the store as it is
        PCL_TRY(ahead_resources(ctx, slots));
        sp = ctx->ahead_last;
        sp.K = k_ahead;
        sp.big = big;
        sp.used = 0;
        sp.before0 = before;
        sp.slots = slots;
        sp.fresh = fresh;
        PCL_TRY(PCL_DISPATCH(ctx, ahead_launch_t<double>(ctx), ahead_launch_t<float>(ctx)));
        PCL_TRY(ahead_wait(ctx));
        for (int i = 0; i < kAheadWork; ++i) ctx->ahead_work[i] += (int64_t)ctx->ahead_host[kAheadMax * kAheadRow + 1 + i];
        ctx->ahead_clock_cycles += (double)ctx->ahead_host[kAheadMax * kAheadRow + 1 + 4];
        ctx->ahead_clock_ticks += (double)ctx->ahead_host[kAheadMax * kAheadRow + 1 + 5];
        sp.active = true;
        ++ctx->ahead_launches;
        return ahead_serve(ctx, alive_out);
    }
    if (!compact_now) {
        // the list of moves r has not seen is full: this body's kernel writes r back, this body's move included
        // (PCL_ALIVE_FLUSH_KERNEL: a separate k_apply_pending launch first, for A/B)
        static knob k_flush("PCL_ALIVE_FLUSH_KERNEL");
        const bool flush_kernel = k_flush.set();
        bool write_r = !pend_has_room(ctx, dt);
        if (write_r && flush_kernel) {
            ctx->slots = slots;
            PCL_TRY(PCL_DISPATCH(ctx, apply_pending_t<double>(ctx), apply_pending_t<float>(ctx)));
            write_r = false;
        }
        PCL_TRY(PCL_DISPATCH(ctx, delete_alive_t<double>(ctx, slots, fresh, true, write_r, dt, A, n, seed, step, planes_host, n_planes),
                             delete_alive_t<float>(ctx, slots, fresh, true, write_r, dt, A, n, seed, step, planes_host, n_planes)));
        PCL_TRY(wait_alive(ctx, before, &alive)); // the last workgroup wrote the totals into the pinned block
        ctx->holes = true;
        ctx->seg_prefix = false;
        ctx->slots = slots;
        if (write_r)
            ctx->pend_n = 0;
        else
            pend_push(ctx, dt, 1);
        ctx->count = alive;
        ctx->last_delete_n = before;
    } else {
        // flags on the mask, then the usual scan + compaction of the survivors (whose r catches up on the way)
        PCL_TRY(ensure_alt(ctx));
        ctx->holes = false;
        ctx->count = slots; // launch geometry of the pipeline = the extent
        int dv_mode = kDvMove;
        int rc = decide_dv_mode(ctx, true, &dv_mode);
        if (rc == PCL_OK)
            rc = PCL_DISPATCH(ctx, delete_alive_t<double>(ctx, slots, fresh, false, false, dt, A, n, seed, step, planes_host, n_planes),
                              delete_alive_t<float>(ctx, slots, fresh, false, false, dt, A, n, seed, step, planes_host, n_planes));
        if (rc == PCL_OK)
            rc = PCL_DISPATCH(ctx, compact_alive_t<double>(ctx, dv_mode, true, dt, planes_host, n_planes),
                              compact_alive_t<float>(ctx, dv_mode, true, dt, planes_host, n_planes));
        if (rc == PCL_OK && n_planes >= 0) {
            rc = hipMemcpyAsync(ctx->h_cnt + 1, ctx->d_cnt + 1, (size_t)(3 + np) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                         hipStreamSynchronize(ctx->stream) == hipSuccess
                     ? PCL_OK : fail(PCL_ERR_HIP, "reading the counters failed");
        }
        if (rc == PCL_OK) rc = wait_count(ctx, before, &alive);
        if (rc != PCL_OK) {
            ctx->holes = !fresh;
            ctx->count = before;
            return rc;
        }
        adopt_compacted(ctx, alive, before);
        ctx->slots = 0;
        ctx->pend_n = 0;
    }
    ctx->last_delete_masked = true;
    ctx->last_delete_slots = slots;
    *alive_out = alive;
    return PCL_OK;
}

template <typename T>
int fused_delete_t(pcl_ctx *ctx, double dt, double A, double n, bool lazy, int dv_mode, int rng_mode, uint64_t seed,
                   uint32_t step, const double *planes_host, int n_planes) {
    const int64_t N = ctx->count;
    const int tiles = (int)div_up(N, kTile);
    newtonmask_args<T> m{};
    for (int k = 0; k < 3; ++k) {
        m.v[k] = F<T>(ctx, PCL_V0 + k);
        m.r[k] = F<T>(ctx, PCL_R0 + k);
        m.dr[k] = F<T>(ctx, PCL_DR0 + k);
    }
    m.rand = static_cast<const T *>(ctx->rnd[2]);
    m.ids = ctx->ids_iota ? nullptr : ctx->ids;
    m.kind = ctx->kind;
    m.masks = ctx->masks;
    m.tile_keep = ctx->tile_keep;
    m.id_base = ctx->id_base;
    m.N = N;
    m.ts = tile_stride(ctx);
    m.dt = (T)dt;
    m.An = (T)A * (T)n;
    m.seed = seed;
    m.step = step;
    m.rng_mode = rng_mode;
    m.lazy = lazy ? 1 : 0;
    // dr implicit: the flag needs |v * dt| only, so pass 1 reads v (24 B) and leaves r alone; pass 3 moves the survivors
    // while they go through its registers.  Eager mode writes dr and r for everybody here, as before.
    static const bool move_in_pass1 = getenv("PCL_MOVE_IN_PASS1") != nullptr; // perf-experiment hook: round 1's split
    m.flag_only = (lazy && !move_in_pass1) ? 1 : 0;
    m.zero_cnt = ctx->d_cnt + 1;
    m.n_zero = 3 + (n_planes > 0 ? n_planes : 0);
    const int ps = prof_begin(ctx, PCL_PROF_DELETE_MASK);
    static const bool no_pair = getenv("PCL_MASK_SINGLE") != nullptr; // perf-experiment hook: the lane == particle kernel
    if (m.flag_only && !m.kind && rng_mode == PCL_RNG_PHILOX && !no_pair)
        hipLaunchKernelGGL(k_flag_mask2<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    else
        hipLaunchKernelGGL(k_newton_mask<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    prof_end(ctx, ps);
    PCL_TRY(launch_check("k_newton_mask"));
    PCL_TRY(scan_tiles(ctx, N));
    compact_counter_args<T> cc{};
    cc.cnt = ctx->d_cnt;
    cc.n_planes = n_planes;
    cc.dt = (T)dt;
    cc.move = m.flag_only;
    plane_table<T>(planes_host, n_planes, cc.plane_ax, cc.plane_L);
    return launch_compact_count<T>(ctx, !lazy, dv_mode, cc);
}

// The delete loop body as ONE kernel (k_delete_onepass).  Returns PCL_OK with *gave_up = true when a look-back timed out:
// nothing of the store has been modified then and the caller runs the three-kernel pipeline.
template <typename T>
int fused_delete_onepass_t(pcl_ctx *ctx, double dt, double A, double n, int dv_mode, int rng_mode, uint64_t seed,
                           uint32_t step, const double *planes_host, int n_planes) {
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type W;
    const int64_t N = ctx->count;
    const int64_t units = div_up(N, kUnit);
    if (ctx->lb_units < units) {
        dev_free(ctx->lb_status);
        ctx->lb_units = 0;
        PCL_TRY(dev_alloc(&ctx->lb_status, div_up(ctx->capacity, kUnit)));
        ctx->lb_units = div_up(ctx->capacity, kUnit);
    }
    PCL_HIP(hipMemsetAsync(ctx->lb_status, 0, (size_t)units * sizeof(unsigned long long), ctx->stream));
    PCL_HIP(hipMemsetAsync(ctx->d_cnt + kCounterSlots - 4, 0, 2 * sizeof(uint64_t), ctx->stream)); // ticket, err
    compact_args ca{};
    if (dv_mode == kDvSkip) dv_mode = kDvMove; // (this kernel always moves ten fields)
    ctx->compact_dv_mode = dv_mode;
    compact_fields(ctx, ca, false, dv_mode); // r v (dv | vprev) E, and the id arrays
    onepass_args<T, W> a{};
    for (int f = 0; f < 10; ++f) {
        a.src[f] = static_cast<const W *>(ca.src[f]);
        a.dst[f] = static_cast<W *>(ca.dst[f]);
    }
    a.rand = static_cast<const T *>(ctx->rnd[2]);
    a.ids_src = ca.ids_src;
    a.ids_dst = ca.ids_dst;
    a.masks = ctx->masks;
    a.status = ctx->lb_status;
    a.ticket = reinterpret_cast<unsigned int *>(ctx->d_cnt + kCounterSlots - 4);
    a.err = reinterpret_cast<int *>(ctx->d_cnt + kCounterSlots - 3);
    a.total = reinterpret_cast<int64_t *>(ctx->d_cnt + kCounterSlots - 1);
    a.cnt = ctx->d_cnt;
    a.id_base = ctx->id_base;
    a.N = N;
    a.ts = tile_stride(ctx);
    a.dt = (T)dt;
    a.An = (T)A * (T)n;
    a.seed = seed;
    a.step = step;
    a.rng_mode = rng_mode;
    a.n_planes = n_planes;
    plane_table<T>(planes_host, n_planes, a.plane_ax, a.plane_L);
    const int ps = prof_begin(ctx, PCL_PROF_ONEPASS);
    hipLaunchKernelGGL((k_delete_onepass<T, W>), dim3((unsigned)units), dim3(kBlock), 0, ctx->stream, a);
    prof_end(ctx, ps);
    PCL_TRY(launch_check("k_delete_onepass"));
    PCL_HIP(hipMemcpyAsync(ctx->h_cnt + kCounterSlots - 4, ctx->d_cnt + kCounterSlots - 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    return PCL_OK;
}

template <typename T>
int fused_delete_multi_t(pcl_ctx *ctx, double dt, int k_steps, double A, double n, int dv_mode, uint64_t seed,
                         uint32_t step, const double *planes_host, int n_planes) {
    const int64_t N = ctx->count;
    const int tiles = (int)div_up(N, kTile);
    newtonmask_multi_args<T> m{};
    for (int k = 0; k < 3; ++k) {
        m.v[k] = F<T>(ctx, PCL_V0 + k);
        m.r[k] = F<T>(ctx, PCL_R0 + k);
    }
    m.ids = ctx->ids_iota ? nullptr : ctx->ids;
    m.kind = ctx->kind;
    m.masks = ctx->masks;
    m.tile_keep = ctx->tile_keep;
    m.cnt = ctx->d_multi;
    m.id_base = ctx->id_base;
    m.N = N;
    m.ts = tile_stride(ctx);
    m.dt = (T)dt;
    m.An = (T)A * (T)n;
    m.seed = seed;
    m.step = step;
    m.K = k_steps;
    m.n_planes = n_planes;
    plane_table<T>(planes_host, n_planes, m.plane_ax, m.plane_L);
    const int ps = prof_begin(ctx, PCL_PROF_DELETE_MASK);
    static const bool no_queue = getenv("PCL_MULTI_NOQUEUE") != nullptr; // perf-experiment hook: lane == photon throughout
    // EXPERIMENT (PCL_MULTI_FORM=p): persistent lanes (k_newton_mask_multi_p).  Bit-identical, 20 % fewer VALU instructions
    // than the ring at K = 16, but its per-round dependent gathers and LDS atomics leave the VALU 45 % busy: slower at
    // K = 8 (2.97 vs 2.12 ms at 1e8 photons), equal from K = 16 on -- not the default.
    static const char *form = getenv("PCL_MULTI_FORM");
    if (k_steps > 2 && !no_queue && form && !strcmp(form, "p"))
        hipLaunchKernelGGL(k_newton_mask_multi_p<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    else if (k_steps > 2 && !no_queue) {
        const size_t lds = (size_t)k_steps * (4 + (n_planes > 0 ? n_planes : 0)) * sizeof(uint32_t); // the per-step rows
        const bool two = form && !strncmp(form, "ring", 4) && form[4] ? form[4] == '2' : k_steps >= 12;
        if (two)
            hipLaunchKernelGGL((k_newton_mask_multi_q<T, 2>), dim3(tiles), dim3(kBlock), lds, ctx->stream, m);
        else
            hipLaunchKernelGGL((k_newton_mask_multi_q<T, 1>), dim3(tiles), dim3(kBlock), lds, ctx->stream, m);
    }
    else
        hipLaunchKernelGGL(k_newton_mask_multi<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    prof_end(ctx, ps);
    PCL_TRY(launch_check("k_newton_mask_multi"));
    PCL_TRY(scan_tiles(ctx, N));
    compact_counter_args<T> cc{};
    cc.cnt = ctx->d_cnt;
    cc.n_planes = -1; // the counters were taken step by step in pass 1
    cc.dt = (T)dt;
    return launch_compact_count<T>(ctx, false, dv_mode, cc);
}

// the tracked subset's positions over the next K passes (pcl_trace_body), from the store as it stands
template <typename T>
int trace_ahead_t(pcl_ctx *ctx, int n_ids, double dt, int k_passes, int n_phases, const int *phase_del, int record_phase, double A,
                  double n, bool use_e, bool var_n, rtc_entry *ent, double c, double h, double A_del, double n_del, uint64_t seed,
                  uint32_t step0) {
    if (use_e) PCL_TRY(ensure_lam4_t<T>(ctx, h, c));
    pcl_trace_args<T> f{};
    f.r0 = F<T>(ctx, PCL_R0); f.r1 = F<T>(ctx, PCL_R1); f.r2 = F<T>(ctx, PCL_R2);
    f.v0 = F<T>(ctx, PCL_V0); f.v1 = F<T>(ctx, PCL_V1); f.v2 = F<T>(ctx, PCL_V2);
    f.p0 = static_cast<const T *>(ctx->vprev[0]); f.p1 = static_cast<const T *>(ctx->vprev[1]); f.p2 = static_cast<const T *>(ctx->vprev[2]);
    f.q0 = F<T>(ctx, PCL_DV0); f.q1 = F<T>(ctx, PCL_DV1); f.q2 = F<T>(ctx, PCL_DV2);
    f.dv_mode = ctx->lazy_dv ? 1 : (ctx->dv_zero == 1 ? 0 : 2);
    f.lam4 = static_cast<const T *>(ctx->lam4);
    f.E = F<T>(ctx, PCL_E);
    f.ids = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->ids);
    f.kind = ctx->kind;
    f.want = reinterpret_cast<const pcl_i64 *>(ctx->trace_want);
    f.slot = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->trace_slot);
    f.out = ctx->trace_out;
    f.id_base = ctx->id_base;
    f.N = ctx->holes ? ctx->slots : ctx->count;
    f.ts = tile_stride(ctx);
    f.n_want = n_ids;
    f.dt = (T)dt; f.A = (T)A; f.n = (T)n; f.c = (T)c;
    f.An_del = (T)A_del * (T)n_del; // one IEEE multiply in the store's precision == the kernels' (A * n)
    f.seed = seed;
    f.step = step0;
    f.K = k_passes;
    f.P = n_phases;
    for (int j = 0; j < PCL_MIXED_MAXPH; ++j) f.phase_del[j] = j < n_phases ? phase_del[j] : 0;
    f.record_phase = record_phase;
    if (ctx->holes) { // the store keeps removed photons' slots: the alive bits, and the moves r has not seen yet
        f.alive = reinterpret_cast<const pcl_u64 *>(ctx->masks);
        f.n_pend = ctx->pend_n;
        for (int q = 0; q < ctx->pend_n; ++q) f.pend_dt[q] = (T)ctx->pend_dt[q], f.pend_rep[q] = ctx->pend_rep[q];
    }
    if (f.slot) { // explicit ids: one sweep over the id row finds the tracked photons' slots
        PCL_HIP(hipMemsetAsync(ctx->trace_slot, 0xFF, (size_t)n_ids * sizeof(int64_t), ctx->stream));
        hipLaunchKernelGGL(k_trace_slots, dim3(grid_for(ctx, f.N, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->ids,
                           ctx->holes ? ctx->masks : nullptr, f.N, ctx->trace_want,
                           n_ids, ctx->trace_slot);
        PCL_TRY(launch_check("k_trace_slots"));
    }
    const int grid = (int)div_up(n_ids, kBlock);
    const int d = sizeof(T) == 8 ? 0 : 1;
    set_np(f.np, ent);
    if (var_n && ent->module && ent->trace[d][use_e ? 1 : 0]) {
        PCL_TRY(launch_module(ctx, ent->trace[d][use_e ? 1 : 0], grid, f, "trace_ahead (hipRTC)"));
    } else {
        PCL_AOT_LAUNCH(k_trace, T, use_e, var_n, grid, f);
        PCL_TRY(launch_check("k_trace"));
    }
    return PCL_OK;
}

// K passes of a loop with an isotropic-scatter phase and/or a delete phase (pcl_mixed_body): the pass itself.
template <typename T>
int step_mixed_t(pcl_ctx *ctx, double dt, int k_passes, int n_phases, const int *phase_del, double A, double n, bool use_e,
                 bool var_n, rtc_entry *ent, double c, double h, double A_del, double n_del, uint64_t seed, uint32_t step,
                 const double *planes_host, int n_planes, bool has_delete, int last_iso, bool inplace) {
    const int64_t N = ctx->holes ? ctx->slots : ctx->count; // (behind a segment-prefix mask: the extent)
    const int tiles = (int)div_up(N, kTile);
    if (use_e) PCL_TRY(ensure_lam4_t<T>(ctx, h, c));
    pcl_mixed_args<T> f{};
    f.r0 = F<T>(ctx, PCL_R0); f.r1 = F<T>(ctx, PCL_R1); f.r2 = F<T>(ctx, PCL_R2);
    f.v0 = F<T>(ctx, PCL_V0); f.v1 = F<T>(ctx, PCL_V1); f.v2 = F<T>(ctx, PCL_V2);
    f.vp0 = static_cast<T *>(ctx->vprev[0]); f.vp1 = static_cast<T *>(ctx->vprev[1]); f.vp2 = static_cast<T *>(ctx->vprev[2]);
    f.lam4 = static_cast<const T *>(ctx->lam4);
    f.E = F<T>(ctx, PCL_E);
    f.ids = ctx->ids_iota ? nullptr : reinterpret_cast<const pcl_i64 *>(ctx->ids);
    f.kind = ctx->kind;
    f.masks = reinterpret_cast<pcl_u64 *>(ctx->masks);
    f.tile_keep = ctx->tile_keep;
    f.cnt = reinterpret_cast<pcl_u64 *>(ctx->d_multi);
    f.id_base = ctx->id_base;
    f.N = N;
    f.ts = tile_stride(ctx);
    f.alive_in = ctx->holes ? reinterpret_cast<const pcl_u64 *>(ctx->masks) : nullptr; // (a wave reads its eight words before it writes them)
    f.inplace = inplace ? 1 : 0;
    f.E_w = F<T>(ctx, PCL_E);
    f.lam4_w = (use_e && ctx->lam4_valid) ? static_cast<T *>(ctx->lam4) : nullptr;
    f.ids_out = reinterpret_cast<pcl_i64 *>(ctx->ids);
    f.move_vp = (last_iso >= 0 || ctx->lazy_dv) ? 1 : 0;
    f.dt = (T)dt; f.A = (T)A; f.n = (T)n; f.c = (T)c;
    f.An_del = (T)A_del * (T)n_del;
    f.seed = seed;
    f.step = step;
    f.K = k_passes;
    f.P = n_phases;
    for (int j = 0; j < PCL_MIXED_MAXPH; ++j) f.phase_del[j] = j < n_phases ? phase_del[j] : 0;
    f.has_delete = has_delete ? 1 : 0;
    f.last_iso = last_iso;
    f.n_planes = n_planes > 0 ? n_planes : 0;
    plane_table<T>(planes_host, n_planes, f.plane_ax, f.plane_L);
    const int d = sizeof(T) == 8 ? 0 : 1;
    const int ps = prof_begin(ctx, PCL_PROF_MULTI);
    set_np(f.np, ent);
    ctx->mixed_rows = 2;
    // Three rows of 64 particles per wave and trip (velocities in LDS: pcl_mixed_body_lds) fill the dense passes better than two
    // while fewer than a third of the photons scatter per step -- expected passes per 64 particles, binomial: 0.33 / 0.38 / 0.48
    // at h = 0.25 / 0.30 / 0.33 against 0.50 --, above that two rows do.  PCL_MIXED_NE3=1 always, =0 never.
    static knob k_ne3("PCL_MIXED_NE3");
    const int ne3_mode = !k_ne3.set() ? -1 : (k_ne3.off() ? 0 : 1);
    if (var_n && ent->module) {
        // a variable_n_fn: the hit fraction is whatever the last scatter phase of the launch before showed (unknown: two rows)
        hipFunction_t fn3 = ent->mixed3[d][use_e ? 1 : 0];
        const bool ne3 = fn3 && (ne3_mode == 1 || (ne3_mode == -1 && ctx->mixed_last_h >= 0.0 && ctx->mixed_last_h < 0.33));
        ctx->mixed_rows = ne3 ? 3 : 2;
        PCL_TRY(launch_module(ctx, ne3 ? fn3 : ent->mixed[d][use_e ? 1 : 0], tiles, f, "step_mixed_multi (hipRTC)"));
    } else {
        // constant n: a photon at speed c scatters with probability A n c dt a step -- known before the first launch.  With the
        // wavelength term the probability differs from photon to photon: by the launch before, as above.
        const double h_est = use_e ? ctx->mixed_last_h : A * n * c * dt;
        const bool ne3 = !var_n && (ne3_mode == 1 || (ne3_mode == -1 && h_est >= 0.0 && h_est < 0.33));
        ctx->mixed_rows = ne3 ? 3 : 2;
        if (ne3 && use_e) {
            hipLaunchKernelGGL((k_mixed3<T, true>), dim3(tiles), dim3(kBlock), 0, ctx->stream, f);
        } else if (ne3) {
            hipLaunchKernelGGL((k_mixed3<T, false>), dim3(tiles), dim3(kBlock), 0, ctx->stream, f);
        } else {
            PCL_AOT_LAUNCH_SHAPED(k_mixed, T, use_e, var_n, tiles, f);
        }
        PCL_TRY(launch_check("k_mixed"));
    }
    prof_end(ctx, ps);
    return PCL_OK;
}

// scan + stable compaction after a pass that left keep-masks (counters off: the pass tallied them)
template <typename T>
int compact_after_pass_t(pcl_ctx *ctx, double dt, bool has_dr, int dv_mode) {
    PCL_TRY(scan_tiles(ctx, ctx->count));
    compact_counter_args<T> cc{};
    cc.cnt = ctx->d_cnt;
    cc.n_planes = -1;
    cc.dt = (T)dt;
    return launch_compact_count<T>(ctx, has_dr, dv_mode, cc);
}

template <typename T>
int scatter_pcoll_t(pcl_ctx *ctx, double A, double n, bool use_e, double c, double h, void *out_dev) {
    const int64_t N = ctx->count;
    const int grid = grid_for(ctx, N, kBlock);
    if (use_e)
        hipLaunchKernelGGL((k_pcoll<T, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const T *)F<T>(ctx, PCL_DR0),
                           (const T *)F<T>(ctx, PCL_DR1), (const T *)F<T>(ctx, PCL_DR2), (const T *)F<T>(ctx, PCL_E), (T)A, (T)n, (T)h,
                           (T)c, static_cast<T *>(out_dev), N, tile_stride(ctx));
    else
        hipLaunchKernelGGL((k_pcoll<T, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const T *)F<T>(ctx, PCL_DR0),
                           (const T *)F<T>(ctx, PCL_DR1), (const T *)F<T>(ctx, PCL_DR2), (const T *)F<T>(ctx, PCL_E), (T)A, (T)n, (T)h,
                           (T)c, static_cast<T *>(out_dev), N, tile_stride(ctx));
    return launch_check("k_pcoll");
}

template <typename T>
int plane_energies_t(pcl_ctx *ctx, int ax, double L) {
    const int64_t N = ctx->count;
    const int tiles = (int)div_up(N, kTile);
    crossmask_args<T> m{};
    m.x = F<T>(ctx, PCL_R0 + ax);
    m.dx = F<T>(ctx, PCL_DR0 + ax);
    m.kind = ctx->kind;
    m.masks = ctx->masks;
    m.tile_keep = ctx->tile_keep;
    m.N = N;
    m.ts = tile_stride(ctx);
    m.L = (T)L;
    hipLaunchKernelGGL(k_cross_mask<T>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    PCL_TRY(launch_check("k_cross_mask"));
    PCL_TRY(scan_tiles(ctx, N));
    compact_args ca{};
    ca.src[0] = ctx->field[PCL_E];
    ca.dst[0] = ctx->e_out;
    ca.masks = ctx->masks;
    ca.tile_off = ctx->tile_off;
    ca.N = N;
    ca.ts = tile_stride(ctx);
    ca.dense_dst = 1;
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type W;
    hipLaunchKernelGGL((k_compact<W, 1>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca);
    return launch_check("k_compact");
}

} // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int pcl_abi_version(void) { return PCL_ABI_VERSION; }
const char *pcl_last_error(void) { return g_err.c_str(); }

int pcl_set_knob(const char *name, const char *value) {
    if (!name || strncmp(name, "PCL_", 4) != 0) return fail(PCL_ERR_ARG, "a knob's name starts with PCL_");
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (value)
        g_knob_over[name] = value;
    else
        g_knob_over.erase(name);
    g_knob_gen.fetch_add(1, std::memory_order_acq_rel);
    return PCL_OK;
}

int pcl_store_last_multi_work(pcl_ctx *ctx, int64_t *dense_passes_out, int64_t *wave_steps_out, int *photons_per_wave_out,
                              int64_t *saturated_wave_steps_out) {
    if (!ctx) return fail(PCL_ERR_ARG, "NULL argument");
    if (dense_passes_out) *dense_passes_out = ctx->multi_work[0];
    if (wave_steps_out) *wave_steps_out = ctx->multi_work[1];
    if (photons_per_wave_out) *photons_per_wave_out = (int)ctx->multi_work[2];
    if (saturated_wave_steps_out) *saturated_wave_steps_out = ctx->multi_work[3];
    return PCL_OK;
}

int pcl_store_last_multi_hist(pcl_ctx *ctx, int64_t *hist129_out) {
    if (!ctx || !hist129_out) return fail(PCL_ERR_ARG, "NULL argument");
    if (ctx->multi_hist_at < 0) return fail(PCL_ERR_STATE, "no histogram: it needs a debug build of the K-step kernels (PCL_RTC_EXTRA=PCL_HIT_HIST) and PCL_MULTI_HIST=1");
    for (int b = 0; b < 129; ++b) hist129_out[b] = (int64_t)ctx->h_multi[ctx->multi_hist_at + b];
    return PCL_OK;
}

int pcl_store_ahead_stats(pcl_ctx *ctx, int64_t *launches_out, int64_t *served_out, int64_t *missed_out) {
    if (!ctx) return fail(PCL_ERR_ARG, "NULL argument");
    if (launches_out) *launches_out = ctx->ahead_launches;
    if (served_out) *served_out = ctx->ahead_served;
    if (missed_out) *missed_out = ctx->ahead_missed;
    return PCL_OK;
}

int pcl_store_last_multi_clock(pcl_ctx *ctx, double *ghz_out) {
    if (!ctx || !ghz_out) return fail(PCL_ERR_ARG, "NULL argument");
    *ghz_out = ctx->multi_clock_ghz;
    return PCL_OK;
}

int pcl_store_last_mixed_rows(pcl_ctx *ctx, int *rows_out) {
    if (!ctx || !rows_out) return fail(PCL_ERR_ARG, "NULL argument");
    *rows_out = ctx->mixed_rows;
    return PCL_OK;
}

int pcl_store_ahead_clock(pcl_ctx *ctx, double *ghz_out) {
    if (!ctx || !ghz_out) return fail(PCL_ERR_ARG, "NULL argument");
    *ghz_out = ctx->ahead_clock_ticks > 0.0 ? ctx->ahead_clock_cycles / ctx->ahead_clock_ticks * 0.1 : 0.0;
    return PCL_OK;
}

int pcl_store_ahead_work(pcl_ctx *ctx, int64_t *groups_two_out, int64_t *groups_one_out, int64_t *rounds_two_out, int64_t *rounds_one_out) {
    if (!ctx) return fail(PCL_ERR_ARG, "NULL argument");
    if (groups_two_out) *groups_two_out = ctx->ahead_work[0];
    if (groups_one_out) *groups_one_out = ctx->ahead_work[1];
    if (rounds_two_out) *rounds_two_out = ctx->ahead_work[2];
    if (rounds_one_out) *rounds_one_out = ctx->ahead_work[3];
    return PCL_OK;
}

int pcl_ctx_set_rtc_background(pcl_ctx *ctx, int on) {
    if (!ctx) return fail(PCL_ERR_ARG, "ctx is NULL");
    ctx->rtc_background = on != 0;
    return PCL_OK;
}

int pcl_ctx_rtc_wait(pcl_ctx *ctx, int *pending_out) {
    PCL_TRY(bind(ctx));
    int n = 0;
    for (auto &kv : ctx->rtc)
        if (kv.second.job) {
            ++n;
            poll_job(kv.second, true);
        }
    if (pending_out) *pending_out = n;
    return PCL_OK;
}

int pcl_pool_trim(int64_t *released_out) {
    size_t held = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        held = g_pool_bytes;
        int device = 0;
        const bool had = hipGetDevice(&device) == hipSuccess;
        pool_flush_locked(); // (the idle blocks, physical handles and all, go back to the driver ...)
        if (had) (void)hipSetDevice(device);
    }
    const size_t idle = vmm_release_idle(); // (... and so do the handles that were kept of ranges unmapped earlier)
    if (released_out) *released_out = (int64_t)(held + idle);
    return PCL_OK;
}

int pcl_pool_info(int64_t *idle_blocks_out, int64_t *idle_handles_out, int64_t *parked_va_out, int64_t *remaps_avoided_out, int *vmm_on_out) {
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        if (idle_handles_out) *idle_handles_out = (int64_t)g_vmm_idle_bytes;
        if (parked_va_out) *parked_va_out = (int64_t)g_vmm_parked_bytes;
        if (remaps_avoided_out) *remaps_avoided_out = (int64_t)g_vmm_remaps_avoided;
    }
    if (vmm_on_out) *vmm_on_out = vmm_enabled() ? 1 : 0;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (idle_blocks_out) *idle_blocks_out = (int64_t)g_pool_bytes;
    return PCL_OK;
}

int pcl_pool_bytes(int64_t *idle_out) {
    if (!idle_out) return fail(PCL_ERR_ARG, "idle_out is NULL");
    size_t handles;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        handles = g_vmm_idle_bytes;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    *idle_out = (int64_t)(g_pool_bytes + handles); // idle blocks + idle physical handles of ranges that were unmapped
    return PCL_OK;
}

int pcl_device_count(int *n_out) {
    if (!n_out) return fail(PCL_ERR_ARG, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        return fail(PCL_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n_out = n;
    return PCL_OK;
}

int pcl_ctx_create(int device, void *stream, pcl_ctx **ctx_out) {
    if (!ctx_out) return fail(PCL_ERR_ARG, "ctx_out is NULL");
    *ctx_out = nullptr;
    int n = 0;
    PCL_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(PCL_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    PCL_HIP(hipSetDevice(device));
    pcl_ctx *c = new (std::nothrow) pcl_ctx();
    if (!c) return fail(PCL_ERR_NOMEM, "out of host memory");
    c->device = device;
    hipError_t e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) {
        delete c;
        return fail(PCL_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(PCL_ERR_HIP, "libphysicl_hip is built for gfx950 (MI355X) only; device %d is %s", device,
                    arch.c_str());
    }
    // 64 per CU (round 2, measured with chosen slabs: K-step pass 1.62e11 at 8 per CU, 1.71e11 at 16, 1.75e11 at 32, 1.77e11 at
    // 64; one-step kernel 0.707 / 0.723 / 0.725 / 0.733 of peak; one chunk per workgroup loses the prefetch across trips: 0.48)
    c->grid_cap = c->prop.multiProcessorCount * 64;
    if (const char *g = getenv("PCL_GRID_PER_CU")) // perf-experiment hook
        if (atoi(g) > 0) c->grid_cap = c->prop.multiProcessorCount * atoi(g);
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(PCL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_count, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_cnt), (kCounterSlots + kAccSlots) * sizeof(uint64_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_cnt), kCounterSlots * sizeof(uint64_t)) != hipSuccess) {
        pcl_ctx_destroy(c);
        return fail(PCL_ERR_HIP, "context resource allocation failed");
    }
    memset(c->h_cnt, 0, kCounterSlots * sizeof(uint64_t));
    (void)hipMemset(c->d_cnt, 0, (kCounterSlots + kAccSlots) * sizeof(uint64_t));
    for (int b = 0; b < 2; ++b) {
        if (hipMalloc(reinterpret_cast<void **>(&c->d_bank[b]), kCounterSlots * sizeof(uint64_t)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&c->h_bank[b]), kCounterSlots * sizeof(uint64_t)) != hipSuccess ||
            hipEventCreateWithFlags(&c->bank_ev[b], hipEventDisableTiming) != hipSuccess) {
            pcl_ctx_destroy(c);
            return fail(PCL_ERR_HIP, "context resource allocation failed");
        }
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_multi), kMultiSlots * sizeof(uint64_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_multi), kMultiSlots * sizeof(uint64_t)) != hipSuccess) {
        pcl_ctx_destroy(c);
        return fail(PCL_ERR_HIP, "context resource allocation failed");
    }
    c->cnt_target = c->d_cnt;
    *ctx_out = c;
    return PCL_OK;
}

int pcl_ctx_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    pcl_store_free(ctx);
    for (auto &kv : ctx->rtc)
        if (kv.second.module) (void)hipModuleUnload(kv.second.module);
    if (ctx->d_cnt) (void)hipFree(ctx->d_cnt);
    if (ctx->h_cnt) (void)hipHostFree(ctx->h_cnt);
    if (ctx->d_multi) (void)hipFree(ctx->d_multi);
    if (ctx->h_multi) (void)hipHostFree(ctx->h_multi);
    if (ctx->ahead_acc) (void)hipFree(ctx->ahead_acc);
    if (ctx->ahead_host) (void)hipHostFree(ctx->ahead_host);
    if (ctx->ahead_death) (void)hipFree(ctx->ahead_death);
    for (int b = 0; b < 2; ++b) {
        if (ctx->d_bank[b]) (void)hipFree(ctx->d_bank[b]);
        if (ctx->h_bank[b]) (void)hipHostFree(ctx->h_bank[b]);
        if (ctx->bank_ev[b]) (void)hipEventDestroy(ctx->bank_ev[b]);
    }
    for (auto &p : ctx->prof) {
        if (p.a) (void)hipEventDestroy(p.a);
        if (p.b) (void)hipEventDestroy(p.b);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_count) (void)hipEventDestroy(ctx->ev_count);
    for (int k = 0; k < 2; ++k) {
        if (ctx->r3_host[k]) (void)hipHostFree(ctx->r3_host[k]);
        if (ctx->r3_dev[k]) (void)hipFree(ctx->r3_dev[k]);
        if (ctx->r3_ev[k]) (void)hipEventDestroy(ctx->r3_ev[k]);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PCL_OK;
}

int pcl_ctx_sync(pcl_ctx *ctx) {
    PCL_TRY(bind(ctx));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_ctx_stream(pcl_ctx *ctx, void **stream_out) {
    if (!ctx || !stream_out) return fail(PCL_ERR_ARG, "NULL argument");
    *stream_out = ctx->stream;
    return PCL_OK;
}

int pcl_ctx_mem_info(pcl_ctx *ctx, int64_t *free_out, int64_t *total_out) {
    PCL_TRY(bind(ctx));
    size_t free_b = 0, total_b = 0;
    PCL_HIP(hipMemGetInfo(&free_b, &total_b));
    if (free_out) *free_out = (int64_t)free_b;
    if (total_out) *total_out = (int64_t)total_b;
    return PCL_OK;
}

int pcl_ctx_device_info(pcl_ctx *ctx, char *name, int name_len, int64_t *hbm_bytes, int *n_cu, int *wavefront) {
    if (!ctx) return fail(PCL_ERR_ARG, "ctx is NULL");
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (wavefront) *wavefront = ctx->prop.warpSize;
    return PCL_OK;
}

int pcl_ctx_device_pci(pcl_ctx *ctx, char *pci, int pci_len) {
    if (!ctx || !pci || pci_len < 16) return fail(PCL_ERR_ARG, "pci buffer must hold at least 16 bytes");
    PCL_HIP(hipDeviceGetPCIBusId(pci, pci_len, ctx->device));
    return PCL_OK;
}

int pcl_dev_alloc(pcl_ctx *ctx, int64_t bytes, void **dev_out) {
    PCL_TRY(bind(ctx));
    if (!dev_out || bytes < 0) return fail(PCL_ERR_ARG, "bad argument");
    PCL_HIP(big_malloc(dev_out, (size_t)(bytes > 0 ? bytes : 1)));
    return PCL_OK;
}

int pcl_dev_free(pcl_ctx *ctx, void *dev) {
    PCL_TRY(bind(ctx));
    if (dev) {
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        big_free(dev);
    }
    return PCL_OK;
}

int pcl_h2d(pcl_ctx *ctx, void *dev, const void *host, int64_t bytes) {
    PCL_TRY(bind(ctx));
    if (bytes < 0 || (bytes > 0 && (!dev || !host))) return fail(PCL_ERR_ARG, "bad argument");
    if (bytes == 0) return PCL_OK;
    PCL_HIP(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_d2h(pcl_ctx *ctx, void *host, const void *dev, int64_t bytes) {
    PCL_TRY(bind(ctx));
    if (bytes < 0 || (bytes > 0 && (!dev || !host))) return fail(PCL_ERR_ARG, "bad argument");
    if (bytes == 0) return PCL_OK;
    PCL_HIP(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_dev_memset(pcl_ctx *ctx, void *dev, int value, int64_t bytes) {
    PCL_TRY(bind(ctx));
    if (bytes < 0 || (bytes > 0 && !dev)) return fail(PCL_ERR_ARG, "bad argument");
    if (bytes) PCL_HIP(hipMemsetAsync(dev, value, (size_t)bytes, ctx->stream));
    return PCL_OK;
}

int pcl_timer_start(pcl_ctx *ctx) {
    PCL_TRY(bind(ctx));
    PCL_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return PCL_OK;
}

int pcl_timer_stop(pcl_ctx *ctx, double *ms_out) {
    PCL_TRY(bind(ctx));
    if (!ms_out) return fail(PCL_ERR_ARG, "ms_out is NULL");
    PCL_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    PCL_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PCL_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = ms;
    return PCL_OK;
}

int pcl_prof_enable(pcl_ctx *ctx, int on) {
    PCL_TRY(bind(ctx));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    ctx->prof_on = on != 0;
    ctx->prof_used = 0;
    return PCL_OK;
}

int pcl_prof_read(pcl_ctx *ctx, int kernel_id, int64_t *launches_out, double *total_ms_out, double *min_ms_out,
                  double *max_ms_out) {
    PCL_TRY(bind(ctx));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    int64_t n = 0;
    double tot = 0.0, mn = 0.0, mx = 0.0;
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        if (ctx->prof[i].kid != kernel_id) continue;
        float ms = 0.f;
        PCL_HIP(hipEventElapsedTime(&ms, ctx->prof[i].a, ctx->prof[i].b));
        tot += ms;
        mn = (n == 0 || ms < mn) ? ms : mn;
        mx = (n == 0 || ms > mx) ? ms : mx;
        ++n;
    }
    if (launches_out) *launches_out = n;
    if (total_ms_out) *total_ms_out = tot;
    if (min_ms_out) *min_ms_out = mn;
    if (max_ms_out) *max_ms_out = mx;
    return PCL_OK;
}

// ------------------------------------------------------------------------------------ Level 1 ----
int pcl_k_light_scatter_step_del(pcl_ctx *ctx, const double *dx, const double *dy, const double *dz,
                                 const double *rand, double n, double A, int32_t *result, int64_t N) {
    PCL_TRY(bind(ctx));
    if (N < 0) return fail(PCL_ERR_ARG, "N < 0");
    if (N == 0) return PCL_OK;
    if (!dx || !dy || !dz || !rand || !result) return fail(PCL_ERR_ARG, "NULL array argument");
    // the kernel text multiplies ``A * n * norm`` (light.py:150): (A*n) first, whatever the argument order
    hipLaunchKernelGGL(k_delete_flags, dim3(grid_for(ctx, N, kBlock)), dim3(kBlock), 0, ctx->stream, dx, dy, dz, rand,
                       A, n, result, N);
    return launch_check("light_scatter_step_del");
}

int pcl_k_scatter_delete_test(pcl_ctx *ctx, const double *d0, const double *d1, const double *d2, const double *rand,
                              double A, double n, int32_t *res, int64_t N) {
    return pcl_k_light_scatter_step_del(ctx, d0, d1, d2, rand, n, A, res, N);
}

int pcl_k_light_scatter_step_sphere(pcl_ctx *ctx, const double *d0, const double *d1, const double *d2,
                                    const double *rtheta, const double *rphi, const double *rand, double A, double n,
                                    const double *E, const double *r0, const double *r1, const double *r2,
                                    double *res0, double *res1, double *res2, int64_t N, int flags, double c, double h,
                                    const char *n_expr) {
    PCL_TRY(bind(ctx));
    if (N < 0) return fail(PCL_ERR_ARG, "N < 0");
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N)) return fail(PCL_ERR_ARG, "unknown flag bits");
    const bool use_e = flags & PCL_SCATTER_WAVELENGTH, var_n = flags & PCL_SCATTER_VARIABLE_N;
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, 0, use_e, &ent)); // (fp64: the reference's ABI) compile even for N == 0 so a bad expression fails early
    if (N == 0) return PCL_OK;
    if (!d0 || !d1 || !d2 || !rtheta || !rphi || !rand || !res0 || !res1 || !res2)
        return fail(PCL_ERR_ARG, "NULL array argument");
    if (use_e && !E) return fail(PCL_ERR_ARG, "E is NULL but PCL_SCATTER_WAVELENGTH is set");
    if (var_n && (!r0 || !r1 || !r2)) return fail(PCL_ERR_ARG, "r0..r2 NULL but PCL_SCATTER_VARIABLE_N is set");
    pcl_sphere_args a{d0, d1, d2, rtheta, rphi, rand, A, n, E, r0, r1, r2, res0, res1, res2, N, c, h, {0, 0, 0.0, 0.0, 0.0}};
    const int grid = grid_for(ctx, N, kBlock);
    set_np(a.np, ent);
    if (var_n && ent->module) return launch_module(ctx, ent->sphere[use_e ? 1 : 0], grid, a, "light_scatter_step_sphere (hipRTC)");
    if (use_e) {
        if (var_n) hipLaunchKernelGGL((k_sphere<true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_sphere<true, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    } else {
        if (var_n) hipLaunchKernelGGL((k_sphere<false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_sphere<false, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    }
    return launch_check("light_scatter_step_sphere");
}

int pcl_k_compact_indices(pcl_ctx *ctx, const int32_t *flags, int64_t N, int64_t *idx_out, int64_t *n_keep_out) {
    PCL_TRY(bind(ctx));
    if (N < 0 || !n_keep_out) return fail(PCL_ERR_ARG, "bad argument");
    *n_keep_out = 0;
    if (N == 0) return PCL_OK;
    if (!flags || !idx_out) return fail(PCL_ERR_ARG, "NULL array argument");
    PCL_TRY(ensure_scratch(ctx, N));
    ctx->last_delete_n = -1; // scratch masks no longer describe the store
    const int tiles = (int)div_up(N, kTile);
    delmask_args<double> m{};
    m.flags_in = flags;
    m.masks = ctx->masks;
    m.tile_keep = ctx->tile_keep;
    m.N = N;
    hipLaunchKernelGGL(k_delete_mask<double>, dim3(tiles), dim3(kBlock), 0, ctx->stream, m);
    PCL_TRY(launch_check("k_delete_mask"));
    PCL_TRY(scan_tiles(ctx, N));
    compact_args ca{};
    ca.idx_dst = idx_out;
    ca.masks = ctx->masks;
    ca.tile_off = ctx->tile_off;
    ca.N = N;
    ca.ts = tile_stride(ctx);
    hipLaunchKernelGGL((k_compact<uint64_t, 0>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca);
    PCL_TRY(launch_check("k_compact"));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    *n_keep_out = (int64_t)ctx->h_cnt[kCounterSlots - 1];
    return PCL_OK;
}

int pcl_expr_validate(const char *n_expr) { return validate_expr(n_expr); }

// ---- user kernels: what CLProgram.build_kernel / run did with OpenCL (physicl/__init__.py:583-597, 656) -------
struct pcl_user_kernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
};

int pcl_user_kernel_build(pcl_ctx *ctx, const char *name, const char *params, const char *body, void **kernel_out) {
    PCL_TRY(bind(ctx));
    if (!name || !params || !body || !kernel_out) return fail(PCL_ERR_ARG, "NULL argument");
    *kernel_out = nullptr;
    if (!rtc_api().ok) return fail(PCL_ERR_RTC, "user kernels need hipRTC: %s", rtc_api().why.c_str());
    for (const char *q = name; *q; ++q)
        if (!((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || (*q >= '0' && *q <= '9') || *q == '_'))
            return fail(PCL_ERR_ARG, "kernel name '%s' is not an identifier", name);
    // the OpenCL-C dialect the reference's kernels are written in, mapped onto HIP; one work-item per element,
    // work-items beyond the global size return before the body runs
    std::string src =
        "#define __kernel\n#define __global\n#define __constant const\n#define __private\n"
        "#define get_global_id(d) ((int)(blockIdx.x * blockDim.x + threadIdx.x))\n"
        "#define get_global_size(d) ((int)pcl_n__)\n"
        "#ifndef NAN\n#define NAN __builtin_nan(\"\")\n#endif\n"
        "#ifndef INFINITY\n#define INFINITY __builtin_inf()\n#endif\n"
        "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n"
        "extern \"C\" __global__ void __launch_bounds__(256) ";
    src += name;
    src += "(";
    src += params;
    src += (*params ? ", " : "");
    src += "long long pcl_n__) {\n  if ((long long)blockIdx.x * blockDim.x + threadIdx.x >= pcl_n__) return;\n";
    src += body;
    src += "\n}\n";
    hiprtcProgram prog;
    hiprtcResult r = rtc_api().CreateProgram(&prog, src.c_str(), "pcl_user_kernel.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return fail(PCL_ERR_RTC, "hiprtcCreateProgram: %s", rtc_api().GetErrorString(r));
    std::string arch = std::string("--offload-arch=") + ctx->prop.gcnArchName;
    const char *opts[] = {arch.c_str(), "-O3", "-ffp-contract=off", "-std=c++17"};
    r = rtc_api().CompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        rtc_api().GetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) rtc_api().GetProgramLog(prog, &log[0]);
        rtc_api().DestroyProgram(&prog);
        return fail(PCL_ERR_RTC, "hipRTC could not compile kernel '%s': %s\n%s", name, rtc_api().GetErrorString(r), log.c_str());
    }
    size_t code_n = 0;
    rtc_api().GetCodeSize(prog, &code_n);
    std::vector<char> code(code_n);
    rtc_api().GetCode(prog, code.data());
    rtc_api().DestroyProgram(&prog);
    pcl_user_kernel *k = new (std::nothrow) pcl_user_kernel();
    if (!k) return fail(PCL_ERR_NOMEM, "out of host memory");
    if (hipModuleLoadData(&k->module, code.data()) != hipSuccess || hipModuleGetFunction(&k->fn, k->module, name) != hipSuccess) {
        if (k->module) (void)hipModuleUnload(k->module);
        delete k;
        return fail(PCL_ERR_RTC, "could not load kernel '%s'", name);
    }
    *kernel_out = k;
    return PCL_OK;
}

int pcl_user_kernel_launch(pcl_ctx *ctx, void *kernel, int64_t n, const void *argbuf, int64_t argbuf_bytes) {
    PCL_TRY(bind(ctx));
    if (!kernel || n < 0 || argbuf_bytes < 0 || (argbuf_bytes > 0 && !argbuf)) return fail(PCL_ERR_ARG, "bad argument");
    if (n == 0) return PCL_OK;
    pcl_user_kernel *k = static_cast<pcl_user_kernel *>(kernel);
    // kernarg = the caller's packed arguments + the trailing global size (8-byte aligned)
    std::vector<char> buf(((size_t)argbuf_bytes + 7) / 8 * 8 + 8, 0);
    if (argbuf_bytes) memcpy(buf.data(), argbuf, (size_t)argbuf_bytes);
    const long long nn = n;
    memcpy(buf.data() + buf.size() - 8, &nn, 8);
    size_t sz = buf.size();
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    const int64_t blocks = div_up(n, kBlock);
    if (blocks > 0x7FFFFFFF) return fail(PCL_ERR_ARG, "global size too large");
    PCL_HIP(hipModuleLaunchKernel(k->fn, (unsigned)blocks, 1, 1, kBlock, 1, 1, 0, ctx->stream, nullptr, config));
    return launch_check("user kernel");
}

int pcl_user_kernel_free(pcl_ctx *ctx, void *kernel) {
    PCL_TRY(bind(ctx));
    if (!kernel) return PCL_OK;
    pcl_user_kernel *k = static_cast<pcl_user_kernel *>(kernel);
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    return PCL_OK;
}

// ------------------------------------------------------------------------------------ Level 2 ----
int pcl_store_alloc_dtype(pcl_ctx *ctx, int64_t capacity, int dtype) {
    PCL_TRY(bind(ctx));
    if (capacity <= 0) return fail(PCL_ERR_ARG, "capacity must be positive");
    if (dtype != PCL_DTYPE_F64 && dtype != PCL_DTYPE_F32) return fail(PCL_ERR_ARG, "unknown dtype %d", dtype);
    pcl_store_free(ctx);
    ctx->dtype = dtype;
    ctx->esz = dtype == PCL_DTYPE_F64 ? 8 : 4;
    ctx->tiles = div_up(capacity, kTileT);
    for (int k = 0; k < kRows; ++k) ctx->row[k] = k;
    PCL_TRY(alloc_slab(ctx, &ctx->slab));
    refresh_rows(ctx);
    ctx->capacity = capacity;
    ctx->count = 0;
    ctx->id_base = 0;
    ctx->ids_iota = true;
    set_dv_zero(ctx, 0);
    ctx->alt_dv_zero_n = 0;
    return PCL_OK;
}

int pcl_store_alloc(pcl_ctx *ctx, int64_t capacity) { return pcl_store_alloc_dtype(ctx, capacity, PCL_DTYPE_F64); }

int pcl_store_dtype(pcl_ctx *ctx, int *dtype_out) {
    if (!ctx || !dtype_out) return fail(PCL_ERR_ARG, "NULL argument");
    *dtype_out = ctx->dtype;
    return PCL_OK;
}

int pcl_store_free(pcl_ctx *ctx) {
    if (!ctx) return PCL_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx->slab);
    dev_free(ctx->slab_alt);
    dev_free(ctx->e_out);
    ctx->e_out_cap = 0;
    dev_free(ctx->trace_want);
    dev_free(ctx->trace_slot);
    if (ctx->trace_out) (void)hipHostFree(ctx->trace_out);
    ctx->trace_out = nullptr;
    ctx->trace_want_cap = ctx->trace_out_cap = ctx->trace_out_n = 0;
    ctx->trace_ids.clear();
    ctx->tiles = 0;
    refresh_rows(ctx); // all row addresses -> NULL
    ctx->lazy_dr = ctx->lazy_dv = false;
    ctx->lam4_valid = false;
    dev_free(ctx->ids);
    dev_free(ctx->ids_alt);
    dev_free(ctx->kind);
    dev_free(ctx->kind_alt);
    for (int k = 0; k < 3; ++k) {
        dev_free(ctx->rnd[k]);
        ctx->rnd_n[k] = 0;
    }
    dev_free(ctx->masks);
    dev_free(ctx->masks_prev);
    dev_free(ctx->tile_keep);
    dev_free(ctx->tile_off);
    dev_free(ctx->lb_status);
    drop_holes(ctx);
    ctx->lb_units = 0;
    set_dv_zero(ctx, 0);
    ctx->alt_dv_zero_n = 0;
    ctx->scratch_cap = 0;
    ctx->capacity = ctx->count = 0;
    ctx->last_delete_n = -1;
    ctx->ids_iota = true;
    return PCL_OK;
}

int pcl_store_capacity(pcl_ctx *ctx, int64_t *capacity_out) {
    if (!ctx || !capacity_out) return fail(PCL_ERR_ARG, "NULL argument");
    *capacity_out = ctx->capacity;
    return PCL_OK;
}

int pcl_store_count(pcl_ctx *ctx, int64_t *count_out) {
    if (!ctx || !count_out) return fail(PCL_ERR_ARG, "NULL argument");
    *count_out = ctx->count;
    return PCL_OK;
}

int pcl_store_reserve_compaction(pcl_ctx *ctx) {
    PCL_TRY(need_store_raw(ctx));
    if (ctx->holes) return PCL_OK; // (only a store that has already deleted is behind a mask: everything exists)
    PCL_TRY(ensure_scratch(ctx, ctx->capacity));
    // (... and what the bodies worked out ahead need: a byte per slot, the accumulators, the pinned rows)
    if (!ctx->kind && alive_enabled() && ahead_k() > 0) PCL_TRY(ahead_resources(ctx, ctx->capacity));
    return ensure_alt(ctx);
}

int pcl_store_slots(pcl_ctx *ctx, int64_t *slots_out, int *pending_moves_out) {
    if (!ctx || !slots_out) return fail(PCL_ERR_ARG, "NULL argument");
    if (ctx->ahead.active) { // (bodies worked out ahead: the extent they swept; r is owed the moves of those handed out)
        *slots_out = ctx->ahead.slots;
        if (pending_moves_out) *pending_moves_out = pend_moves(ctx) + ctx->ahead.used;
        return PCL_OK;
    }
    *slots_out = ctx->holes ? ctx->slots : ctx->count;
    if (pending_moves_out) *pending_moves_out = ctx->holes ? pend_moves(ctx) : 0;
    return PCL_OK;
}

int pcl_store_set_count(pcl_ctx *ctx, int64_t count, int64_t id_base) {
    PCL_TRY(need_store(ctx));
    if (count < 0 || count > ctx->capacity) return fail(PCL_ERR_ARG, "count %lld outside [0, capacity]", (long long)count);
    ctx->count = count;
    ctx->id_base = id_base;
    ctx->ids_iota = true;
    ctx->last_delete_n = -1;
    ctx->lam4_valid = false;
    ctx->ahead_last_valid = false; // (a new population: its first delete body is taken for the start of a loop again)
    ctx->ahead_wait = ctx->ahead_backoff = 0;
    set_dv_zero(ctx, 0);
    ctx->alt_dv_zero_n = 0;
    // a new population: every particle is a photon again until pcl_store_upload_kind says otherwise (a kind array left
    // over from an earlier, mixed upload would silently switch the light steps off for whoever sits at those indices)
    if (ctx->kind || ctx->kind_alt) PCL_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->kind);
    dev_free(ctx->kind_alt);
    return PCL_OK;
}

int pcl_store_upload(pcl_ctx *ctx, int field, const void *host, int64_t offset, int64_t n) {
    PCL_TRY(need_store(ctx));
    if (field < 0 || field >= PCL_NFIELDS) return fail(PCL_ERR_ARG, "unknown field %d", field);
    PCL_TRY(check_range(ctx, offset, n, host));
    if (field == PCL_E) ctx->lam4_valid = false;
    if (field >= PCL_DV0 && field <= PCL_DV2) set_dv_zero(ctx, 0); // unknown again: looked at on the device when it matters
    return copy_row(ctx, ctx->field[field], const_cast<void *>(host), offset, n, true);
}

int pcl_store_download(pcl_ctx *ctx, int field, void *host, int64_t offset, int64_t n) {
    if (field < 0 || field >= PCL_NFIELDS) return fail(PCL_ERR_ARG, "unknown field %d", field);
    // Reading r, v or E leaves an implicit dr / dv implicit: only the dr / dv rows themselves have to be made real to be read.
    // (A run sampled between its launches -- a measure step that walks r -- would otherwise pay for six more rows in every
    // later compaction: bench.py's mixed leg with state samples, 0.096 s against 0.078 s.)
    PCL_TRY(need_store_raw(ctx));
    PCL_TRY(densify(ctx)); // a store behind an alive mask becomes dense (stable) first
    if (field >= PCL_DR0 && field <= PCL_DV2) PCL_TRY(materialize(ctx));
    PCL_TRY(check_range(ctx, offset, n, host));
    return copy_row(ctx, ctx->field[field], host, offset, n, false);
}

int pcl_store_upload_ids(pcl_ctx *ctx, const int64_t *host, int64_t offset, int64_t n) {
    PCL_TRY(need_store(ctx));
    PCL_TRY(check_range(ctx, offset, n, host));
    if (!ctx->ids) PCL_TRY(dev_alloc(&ctx->ids, ctx->capacity));
    ctx->ids_iota = false;
    return pcl_h2d(ctx, ctx->ids + offset, host, n * (int64_t)sizeof(int64_t));
}

int pcl_store_download_ids(pcl_ctx *ctx, int64_t *host, int64_t offset, int64_t n) {
    PCL_TRY(need_store_raw(ctx));
    PCL_TRY(densify(ctx)); // (ids have nothing to do with an implicit dr / dv: those stay as they are)
    PCL_TRY(check_range(ctx, offset, n, host));
    if (ctx->ids_iota) {
        for (int64_t i = 0; i < n; ++i) host[i] = ctx->id_base + offset + i;
        return PCL_OK;
    }
    return pcl_d2h(ctx, host, ctx->ids + offset, n * (int64_t)sizeof(int64_t));
}

int pcl_store_upload_kind(pcl_ctx *ctx, const uint8_t *host, int64_t offset, int64_t n) {
    PCL_TRY(need_store(ctx));
    PCL_TRY(check_range(ctx, offset, n, host));
    if (!ctx->kind) {
        PCL_TRY(dev_alloc(&ctx->kind, ctx->capacity));
        PCL_HIP(hipMemsetAsync(ctx->kind, PCL_KIND_PHOTON, (size_t)ctx->capacity, ctx->stream));
    }
    return pcl_h2d(ctx, ctx->kind + offset, host, n);
}

int pcl_store_download_kind(pcl_ctx *ctx, uint8_t *host, int64_t offset, int64_t n) {
    PCL_TRY(need_store(ctx));
    PCL_TRY(check_range(ctx, offset, n, host));
    if (!ctx->kind) {
        memset(host, PCL_KIND_PHOTON, (size_t)n);
        return PCL_OK;
    }
    return pcl_d2h(ctx, host, ctx->kind + offset, n);
}

int pcl_store_field_ptr(pcl_ctx *ctx, int field, void **dev_out) {
    PCL_TRY(need_store(ctx));
    if (field < 0 || field >= PCL_NFIELDS || !dev_out) return fail(PCL_ERR_ARG, "bad argument");
    if (field == PCL_E) ctx->lam4_valid = false; // the caller may write through the pointer
    if (field >= PCL_DV0 && field <= PCL_DV2) set_dv_zero(ctx, 0);
    *dev_out = ctx->field[field];
    return PCL_OK;
}

int pcl_store_alloc_info(pcl_ctx *ctx, int *n_candidates_out, double *rates_gbps_out, int cap, double *chosen_gbps_out) {
    PCL_TRY(need_store_raw(ctx));
    if (n_candidates_out) *n_candidates_out = ctx->slab_tries;
    if (chosen_gbps_out) *chosen_gbps_out = ctx->slab_chosen;
    if (rates_gbps_out)
        for (int k = 0; k < cap && k < 8; ++k) rates_gbps_out[k] = ctx->slab_rates[k];
    return PCL_OK;
}

int pcl_store_layout(pcl_ctx *ctx, int64_t *tile_len_out, int64_t *tile_stride_out) {
    PCL_TRY(need_store_raw(ctx));
    if (tile_len_out) *tile_len_out = kTileT;
    if (tile_stride_out) *tile_stride_out = tile_stride(ctx);
    return PCL_OK;
}

int pcl_store_upload_rand(pcl_ctx *ctx, int which, const void *host, int64_t n) {
    PCL_TRY(need_store(ctx));
    if (which < 0 || which > 2) return fail(PCL_ERR_ARG, "which must be 0 (rtheta), 1 (rphi) or 2 (rand)");
    PCL_TRY(check_range(ctx, 0, n, host));
    if (!ctx->rnd[which]) PCL_TRY(dev_alloc_bytes(&ctx->rnd[which], ctx->capacity, ctx->esz));
    ctx->rnd_n[which] = n;
    return pcl_h2d(ctx, ctx->rnd[which], host, n * (int64_t)ctx->esz);
}

constexpr int64_t kRand3Chunk = (int64_t)1 << 20; // photons per pcl_store_upload_rand3 call (24 MB of uniforms)

int pcl_store_upload_rand3(pcl_ctx *ctx, const double *u3_host, int64_t offset, int64_t n) {
    PCL_TRY(need_store(ctx));
    if (n < 0 || n > kRand3Chunk) return fail(PCL_ERR_ARG, "n outside [0, %lld] per call", (long long)kRand3Chunk);
    PCL_TRY(check_range(ctx, offset, n, u3_host));
    for (int w = 0; w < 3; ++w) {
        if (!ctx->rnd[w]) PCL_TRY(dev_alloc_bytes(&ctx->rnd[w], ctx->capacity, ctx->esz));
        if (offset != 0 && ctx->rnd_n[w] != offset)
            return fail(PCL_ERR_ARG, "chunks must follow one another: %lld particles are in, this chunk starts at %lld",
                        (long long)ctx->rnd_n[w], (long long)offset);
    }
    if (n == 0) {
        if (offset == 0) ctx->rnd_n[0] = ctx->rnd_n[1] = ctx->rnd_n[2] = 0;
        return PCL_OK;
    }
    const int slot = ctx->r3_slot;
    ctx->r3_slot ^= 1;
    if (!ctx->r3_host[slot]) {
        PCL_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->r3_host[slot]), (size_t)kRand3Chunk * 3 * sizeof(double)));
        PCL_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->r3_dev[slot]), (size_t)kRand3Chunk * 3 * sizeof(double)));
        PCL_HIP(hipEventCreateWithFlags(&ctx->r3_ev[slot], hipEventDisableTiming));
    } else {
        PCL_HIP(hipEventSynchronize(ctx->r3_ev[slot])); // the slot's previous chunk has left the pinned buffer
    }
    memcpy(ctx->r3_host[slot], u3_host, (size_t)n * 3 * sizeof(double));
    PCL_HIP(hipMemcpyAsync(ctx->r3_dev[slot], ctx->r3_host[slot], (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCL_HIP(hipEventRecord(ctx->r3_ev[slot], ctx->stream));
    const int grid = grid_for(ctx, n, kBlock);
    if (ctx->dtype == PCL_DTYPE_F64)
        hipLaunchKernelGGL(k_rand3_split<double>, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->r3_dev[slot],
                           static_cast<double *>(ctx->rnd[0]) + offset, static_cast<double *>(ctx->rnd[1]) + offset,
                           static_cast<double *>(ctx->rnd[2]) + offset, n);
    else
        hipLaunchKernelGGL(k_rand3_split<float>, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->r3_dev[slot],
                           static_cast<float *>(ctx->rnd[0]) + offset, static_cast<float *>(ctx->rnd[1]) + offset,
                           static_cast<float *>(ctx->rnd[2]) + offset, n);
    PCL_TRY(launch_check("k_rand3_split"));
    // (the device slot is reused two calls later: by then this kernel has run -- same stream, in order)
    ctx->rnd_n[0] = ctx->rnd_n[1] = ctx->rnd_n[2] = offset + n;
    return PCL_OK;
}

int pcl_store_fill_photons(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, double e_min, double e_max,
                           uint64_t seed) {
    PCL_TRY(need_store_raw(ctx));
    if (n < 0 || n > ctx->capacity) return fail(PCL_ERR_ARG, "n outside [0, capacity]");
    drop_holes(ctx);
    ctx->lazy_dr = ctx->lazy_dv = false; // every array is overwritten
    ctx->lam4_valid = false;
    if (n > 0) {
        PCL_TRY(PCL_DISPATCH(ctx, fill_photons_t<double>(ctx, n, id_base, c, e_min, e_max, seed),
                             fill_photons_t<float>(ctx, n, id_base, c, e_min, e_max, seed)));
        if (ctx->kind) PCL_HIP(hipMemsetAsync(ctx->kind, PCL_KIND_PHOTON, (size_t)n, ctx->stream));
    }
    ctx->count = n;
    ctx->id_base = id_base;
    ctx->ids_iota = true;
    ctx->last_delete_n = -1;
    set_dv_zero(ctx, 1, n); // the fill wrote +0.0 into every dv element
    return PCL_OK;
}

int pcl_store_fill_photons_table(pcl_ctx *ctx, int64_t n, int64_t id_base, double c, const double *cdf_host,
                                 const double *grid_host, int nbins, uint64_t seed) {
    PCL_TRY(need_store_raw(ctx));
    if (n < 0 || n > ctx->capacity) return fail(PCL_ERR_ARG, "n outside [0, capacity]");
    if (nbins < 1 || !cdf_host || !grid_host) return fail(PCL_ERR_ARG, "bad table");
    drop_holes(ctx);
    ctx->lazy_dr = ctx->lazy_dv = false;
    ctx->lam4_valid = false;
    if (n > 0) {
        double *tab = nullptr;
        PCL_TRY(dev_alloc(&tab, 2 * (int64_t)nbins));
        int rc = pcl_h2d(ctx, tab, cdf_host, nbins * (int64_t)sizeof(double));
        if (rc == PCL_OK) rc = pcl_h2d(ctx, tab + nbins, grid_host, nbins * (int64_t)sizeof(double));
        if (rc == PCL_OK)
            rc = PCL_DISPATCH(ctx, fill_table_t<double>(ctx, n, id_base, c, tab, tab + nbins, nbins, seed),
                              fill_table_t<float>(ctx, n, id_base, c, tab, tab + nbins, nbins, seed));
        if (rc == PCL_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(PCL_ERR_HIP, "sync failed");
        dev_free(tab);
        PCL_TRY(rc);
        if (ctx->kind) PCL_HIP(hipMemsetAsync(ctx->kind, PCL_KIND_PHOTON, (size_t)n, ctx->stream));
    }
    ctx->count = n;
    ctx->id_base = id_base;
    ctx->ids_iota = true;
    ctx->last_delete_n = -1;
    set_dv_zero(ctx, 1, n);
    return PCL_OK;
}

int pcl_step_newton(pcl_ctx *ctx, double dt) {
    PCL_TRY(need_store(ctx));
    if (ctx->count == 0) return PCL_OK;
    return PCL_DISPATCH(ctx, step_newton_t<double>(ctx, dt), step_newton_t<float>(ctx, dt));
}

int pcl_step_scatter_isotropic(pcl_ctx *ctx, double A, double n, int flags, double c, double h, const char *n_expr,
                               int rng_mode, uint64_t seed, uint32_t step, int64_t *hits_out) {
    PCL_TRY(need_store(ctx));
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N | PCL_SCATTER_PY_DV)) return fail(PCL_ERR_ARG, "unknown flag bits");
    if (rng_mode != PCL_RNG_INPUT && rng_mode != PCL_RNG_PHILOX) return fail(PCL_ERR_ARG, "unknown rng_mode %d", rng_mode);
    const bool use_e = flags & PCL_SCATTER_WAVELENGTH, var_n = flags & PCL_SCATTER_VARIABLE_N, py_dv = flags & PCL_SCATTER_PY_DV;
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, ctx->dtype == PCL_DTYPE_F32 ? 1 : 0, use_e, &ent));
    const int64_t N = ctx->count;
    if (hits_out) *hits_out = 0;
    if (N == 0) return PCL_OK;
    if (rng_mode == PCL_RNG_INPUT)
        for (int k = 0; k < 3; ++k)
            if (!ctx->rnd[k] || ctx->rnd_n[k] < N)
                return fail(PCL_ERR_STATE, "PCL_RNG_INPUT needs pcl_store_upload_rand(which=%d) for all %lld particles", k,
                            (long long)N);
    PCL_HIP(hipMemsetAsync(ctx->d_cnt, 0, sizeof(uint64_t), ctx->stream));
    ctx->hits_on_host = false;
    ctx->last_async_bank = -1;
    set_dv_zero(ctx, 2); // the step writes dv
    PCL_TRY(PCL_DISPATCH(ctx, step_scatter_t<double>(ctx, A, n, use_e, var_n, ent, c, h, rng_mode, seed, step, py_dv),
                         step_scatter_t<float>(ctx, A, n, use_e, var_n, ent, c, h, rng_mode, seed, step, py_dv)));
    if (hits_out) {
        PCL_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        ctx->hits_on_host = true;
        *hits_out = (int64_t)ctx->h_cnt[0];
    }
    return PCL_OK;
}

int pcl_step_scatter_pcoll(pcl_ctx *ctx, double A, double n, int flags, double c, double h, void *pcoll_out_host) {
    PCL_TRY(need_store(ctx));
    if (flags & ~PCL_SCATTER_WAVELENGTH) return fail(PCL_ERR_ARG, "only PCL_SCATTER_WAVELENGTH is meaningful here");
    const int64_t N = ctx->count;
    if (N == 0) return PCL_OK;
    if (!pcoll_out_host) return fail(PCL_ERR_ARG, "pcoll_out_host is NULL");
    void *tmp = nullptr;
    PCL_TRY(dev_alloc_bytes(&tmp, N, ctx->esz));
    int rc = PCL_DISPATCH(ctx, scatter_pcoll_t<double>(ctx, A, n, flags & PCL_SCATTER_WAVELENGTH, c, h, tmp),
                          scatter_pcoll_t<float>(ctx, A, n, flags & PCL_SCATTER_WAVELENGTH, c, h, tmp));
    if (rc == PCL_OK) rc = pcl_d2h(ctx, pcoll_out_host, tmp, N * (int64_t)ctx->esz);
    dev_free(tmp);
    return rc;
}

int pcl_step_delete_flags(pcl_ctx *ctx, const int32_t *flags_host, int64_t *n_alive_out, int64_t *n_removed_out) {
    PCL_TRY(need_store(ctx));
    const int64_t N = ctx->count;
    if (n_alive_out) *n_alive_out = N;
    if (n_removed_out) *n_removed_out = 0;
    if (N == 0) return PCL_OK;
    if (!flags_host) return fail(PCL_ERR_ARG, "flags_host is NULL");
    PCL_TRY(ensure_scratch(ctx, N));
    PCL_TRY(ensure_alt(ctx));
    int32_t *d_flags = nullptr;
    PCL_TRY(dev_alloc(&d_flags, N));
    int rc = pcl_h2d(ctx, d_flags, flags_host, N * (int64_t)sizeof(int32_t));
    if (rc == PCL_OK) {
        delmask_args<double> m{};
        m.flags_in = d_flags;
        m.masks = ctx->masks;
        m.tile_keep = ctx->tile_keep;
        m.N = N;
        hipLaunchKernelGGL(k_delete_mask<double>, dim3((int)div_up(N, kTile)), dim3(kBlock), 0, ctx->stream, m);
        rc = launch_check("k_delete_mask");
    }
    if (rc == PCL_OK) rc = scan_tiles(ctx, N);
    if (rc == PCL_OK) {
        compact_args ca{};
        const int nf = compact_fields(ctx, ca, true, false); // every field is real here (need_store materialised)
        (void)nf;
        if (ctx->dtype == PCL_DTYPE_F64)
            hipLaunchKernelGGL((k_compact<uint64_t, PCL_NFIELDS>), dim3((int)div_up(N, kTile)), dim3(kBlock), 0, ctx->stream, ca);
        else
            hipLaunchKernelGGL((k_compact<uint32_t, PCL_NFIELDS>), dim3((int)div_up(N, kTile)), dim3(kBlock), 0, ctx->stream, ca);
        rc = launch_check("k_compact");
    }
    if (rc == PCL_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(PCL_ERR_HIP, "sync failed");
    dev_free(d_flags);
    PCL_TRY(rc);
    int64_t alive = 0;
    PCL_TRY(wait_count(ctx, N, &alive));
    adopt_compacted(ctx, alive, N);
    if (n_alive_out) *n_alive_out = alive;
    if (n_removed_out) *n_removed_out = N - alive;
    return PCL_OK;
}

int pcl_step_fused(pcl_ctx *ctx, double dt, int do_scatter, double A, double n, int flags, double c, double h,
                   const char *n_expr, int rng_mode, uint64_t seed, uint32_t step, const double *planes_host,
                   int n_planes, int64_t *out_host) {
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N | PCL_FUSED_LAZY))
        return fail(PCL_ERR_ARG, "unknown flag bits");
    const bool lazy = flags & PCL_FUSED_LAZY;
    static const bool no_fast = getenv("PCL_NO_FAST") != nullptr; // perf-experiment hook
    const bool fast = lazy && do_scatter && rng_mode == PCL_RNG_PHILOX && n_planes <= 0 && !no_fast;
    if (lazy) {
        // a Newton-only pass keeps the dv of a still-implicit scatter step implicit (dv = v - vprev): only the source of
        // dr changes, to the current v rows
        PCL_TRY(need_store_raw(ctx));
        // a store behind an alive mask: the fast kernels read the mask (all photons: a mask only exists on such stores);
        // every other formulation sees the dense store (dr / dv stay implicit)
        if (!(fast && !ctx->kind)) PCL_TRY(densify(ctx));
    } else {
        PCL_TRY(need_store(ctx));
    }
    if (rng_mode != PCL_RNG_INPUT && rng_mode != PCL_RNG_PHILOX) return fail(PCL_ERR_ARG, "unknown rng_mode %d", rng_mode);
    if (n_planes < -1 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [-1, %d]", PCL_MAX_PLANES);
    if (n_planes > 0 && !planes_host) return fail(PCL_ERR_ARG, "planes_host is NULL");
    if (out_host && n_planes < 0) return fail(PCL_ERR_ARG, "out_host given but counters are off (n_planes = -1)");
    const bool use_e = do_scatter && (flags & PCL_SCATTER_WAVELENGTH), var_n = do_scatter && (flags & PCL_SCATTER_VARIABLE_N);
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, ctx->dtype == PCL_DTYPE_F32 ? 1 : 0, use_e, &ent));
    const int64_t N = ctx->count;
    const int np = n_planes > 0 ? n_planes : 0;
    if (out_host) {
        out_host[0] = N;
        for (int k = 1; k < 5 + np; ++k) out_host[k] = 0;
    }
    if (N == 0) return PCL_OK;
    if (do_scatter && rng_mode == PCL_RNG_INPUT)
        for (int k = 0; k < 3; ++k)
            if (!ctx->rnd[k] || ctx->rnd_n[k] < N)
                return fail(PCL_ERR_STATE, "PCL_RNG_INPUT needs pcl_store_upload_rand(which=%d) for all %lld particles", k,
                            (long long)N);
    // synchronous call: the context's counter slots; asynchronous call with counters: the next free bank
    const bool use_bank = !out_host && n_planes >= 0;
    int bank = -1;
    if (use_bank) {
        if (ctx->bank_pending >= 2)
            return fail(PCL_ERR_STATE, "two un-read asynchronous fused steps are outstanding: call pcl_step_fused_read");
        bank = (ctx->bank_head + ctx->bank_pending) & 1;
        ctx->cnt_target = ctx->d_bank[bank];
    } else {
        ctx->cnt_target = ctx->d_cnt;
    }
    PCL_HIP(hipMemsetAsync(ctx->cnt_target, 0, (size_t)(4 + np) * sizeof(uint64_t), ctx->stream));
    ctx->hits_on_host = false;
    ctx->last_async_bank = bank;
    if (fast) {
        // device RNG, implicit dr/dv, sign counters only (any store: explicit ids / plain Objects take k_fastg)
        PCL_TRY(PCL_DISPATCH(ctx, step_fast_t<double>(ctx, dt, A, n, use_e, var_n, ent, c, h, seed, step),
                             step_fast_t<float>(ctx, dt, A, n, use_e, var_n, ent, c, h, seed, step)));
    } else {
        PCL_TRY(PCL_DISPATCH(
            ctx, step_fused_t<double>(ctx, dt, do_scatter, A, n, use_e, var_n, ent, c, h, rng_mode, seed, step, planes_host, n_planes, lazy),
            step_fused_t<float>(ctx, dt, do_scatter, A, n, use_e, var_n, ent, c, h, rng_mode, seed, step, planes_host, n_planes, lazy)));
    }
    if (!lazy && do_scatter) set_dv_zero(ctx, 2); // the eager step wrote dv
    if (lazy) {
        if (do_scatter) {
            for (int k = 0; k < 3; ++k) std::swap(ctx->row[PCL_V0 + k], ctx->row[kRowVprev + k]); // V rows = new v
            refresh_rows(ctx);
            ctx->lazy_dv = true;
        }
        ctx->lazy_dr = true;
        ctx->lazy_dr_vprev = do_scatter != 0; // the move used the velocities that are now in the vprev rows
        ctx->lazy_dt = dt;
    }
    ctx->cnt_target = ctx->d_cnt;
    if (use_bank) {
        PCL_HIP(hipMemcpyAsync(ctx->h_bank[bank], ctx->d_bank[bank], (size_t)(4 + np) * sizeof(uint64_t),
                               hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipEventRecord(ctx->bank_ev[bank], ctx->stream));
        ctx->bank_count[bank] = N;
        ctx->bank_np[bank] = np;
        ++ctx->bank_pending;
    }
    if (out_host) {
        PCL_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, (size_t)(4 + np) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        ctx->hits_on_host = true;
        for (int k = 0; k < 3 + np; ++k) out_host[1 + k] = (int64_t)ctx->h_cnt[1 + k];
        out_host[4 + np] = (int64_t)ctx->h_cnt[0];
    }
    return PCL_OK;
}

int pcl_store_is_uniform(pcl_ctx *ctx, int *uniform_out) {
    PCL_TRY(need_store_raw(ctx));
    if (!uniform_out) return fail(PCL_ERR_ARG, "uniform_out is NULL");
    *uniform_out = (!ctx->kind && ctx->ids_iota && !ctx->holes) ? 1 : 0; // (behind an alive mask the ids are about to become explicit)
    return PCL_OK;
}

int pcl_step_fused_multi(pcl_ctx *ctx, double dt, int k_steps, double A, double n, int flags, double c, double h,
                         const char *n_expr, uint64_t seed, uint32_t step0, const double *planes_host, int n_planes,
                         int64_t *out_host) {
    PCL_TRY(need_store_raw(ctx));
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N)) return fail(PCL_ERR_ARG, "unknown flag bits");
    if (k_steps < 1 || k_steps > PCL_MULTI_MAX) return fail(PCL_ERR_ARG, "k_steps outside [1, %d]", PCL_MULTI_MAX);
    if (n_planes < 0 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [0, %d]", PCL_MAX_PLANES);
    if (n_planes > 0 && !planes_host) return fail(PCL_ERR_ARG, "planes_host is NULL");
    const int np = n_planes, nslots = 4 + np;
    if (ctx->kind || !ctx->ids_iota || ctx->holes)
        return fail(PCL_ERR_STATE, "pcl_step_fused_multi needs an all-photon store with implicit ids (use pcl_step_fused)");
    if (ctx->bank_pending) return fail(PCL_ERR_STATE, "un-read asynchronous fused steps are outstanding");
    const bool use_e = flags & PCL_SCATTER_WAVELENGTH, var_n = flags & PCL_SCATTER_VARIABLE_N;
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, ctx->dtype == PCL_DTYPE_F32 ? 1 : 0, use_e, &ent));
    const int64_t N = ctx->count;
    if (out_host)
        for (int k = 0; k < k_steps; ++k) {
            out_host[(5 + np) * k] = N;
            for (int j = 1; j < 5 + np; ++j) out_host[(5 + np) * k + j] = 0;
        }
    if (N == 0) return PCL_OK;
    // the v rows must hold the current velocity; a pending implicit dv of an earlier step is simply superseded
    // rows + the two work tallies (+ the 129 bins of a PCL_HIT_HIST debug build's histogram, knob PCL_MULTI_HIST)
    static knob k_hist("PCL_MULTI_HIST");
    const int n_extra = 4 + ((k_hist.set() && !k_hist.off() && nslots * k_steps + 133 <= kMultiSlots) ? 129 : 0);
    PCL_HIP(hipMemsetAsync(ctx->d_multi, 0, (size_t)(nslots * k_steps + n_extra) * sizeof(uint64_t), ctx->stream));
    ctx->hits_on_host = false;
    ctx->last_async_bank = -1;
    PCL_TRY(PCL_DISPATCH(ctx, step_multi_t<double>(ctx, dt, k_steps, A, n, use_e, var_n, ent, c, h, seed, step0, planes_host, np),
                         step_multi_t<float>(ctx, dt, k_steps, A, n, use_e, var_n, ent, c, h, seed, step0, planes_host, np)));
    ctx->lazy_dv = true; // vprev rows = v before the last step
    ctx->lazy_dr = true;
    ctx->lazy_dr_vprev = true;
    ctx->lazy_dt = dt;
    // pcl_store_last_scatter_hits() reports the last of the K steps
    PCL_HIP(hipMemcpyAsync(ctx->d_cnt, ctx->d_multi + nslots * (k_steps - 1), sizeof(uint64_t), hipMemcpyDeviceToDevice,
                           ctx->stream));
    if (out_host) {
        PCL_HIP(hipMemcpyAsync(ctx->h_multi, ctx->d_multi, (size_t)(nslots * k_steps + n_extra) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < k_steps; ++k) {
            for (int j = 0; j < 3 + np; ++j) out_host[(5 + np) * k + 1 + j] = (int64_t)ctx->h_multi[nslots * k + 1 + j];
            out_host[(5 + np) * k + 4 + np] = (int64_t)ctx->h_multi[nslots * k];
        }
        ctx->multi_hist_at = n_extra > 4 ? nslots * k_steps + 4 : -1;
        {   // the clock the chip held under this launch: shader cycles / 100 MHz ticks, summed over the workgroups (pcl_clock_end)
            const uint64_t cyc = ctx->h_multi[nslots * k_steps + 2], ticks = ctx->h_multi[nslots * k_steps + 3];
            ctx->multi_clock_ghz = ticks ? (double)cyc / (double)ticks * 0.1 : 0.0;
        }
        ctx->multi_work[0] = (int64_t)ctx->h_multi[nslots * k_steps];                                  // dense passes
        ctx->multi_work[1] = div_up(N, (int64_t)ctx->multi_work[2]) * (int64_t)k_steps;                // wave-steps
        ctx->multi_work[3] = ctx->multi_sat_used ? (int64_t)ctx->h_multi[nslots * k_steps + 1] : -1;
        if (ctx->multi_sat_used) { // did the probe pay?  (more than half of the wave-steps took the shortcut)
            ctx->multi_sat_on = 2 * ctx->multi_work[3] > ctx->multi_work[1];
            ctx->multi_sat_next = ctx->multi_sat_on ? 0 : 16;
        } else if (ctx->multi_sat_next > 0) {
            --ctx->multi_sat_next;
        }
        ctx->multi_last_h = (double)ctx->h_multi[nslots * (k_steps - 1)] / (double)N; // what the next launch's form goes by
    } else {
        ctx->multi_last_h = -1.0;
        ctx->multi_sat_on = false; // (the tallies of a launch whose rows nobody read are not known)
    }
    return PCL_OK;
}

int pcl_step_fused_read(pcl_ctx *ctx, int n_planes, int64_t *out_host) {
    PCL_TRY(need_store_raw(ctx));
    if (n_planes < 0 || n_planes > PCL_MAX_PLANES || !out_host) return fail(PCL_ERR_ARG, "bad argument");
    if (ctx->bank_pending == 0) return fail(PCL_ERR_STATE, "no asynchronous fused step is waiting to be read");
    const int b = ctx->bank_head;
    if (ctx->bank_np[b] != n_planes) return fail(PCL_ERR_ARG, "n_planes differs from the step being read (%d)", ctx->bank_np[b]);
    PCL_HIP(hipEventSynchronize(ctx->bank_ev[b])); // waits for THAT step only; later steps keep running
    out_host[0] = ctx->bank_count[b];
    for (int k = 0; k < 3 + n_planes; ++k) out_host[1 + k] = (int64_t)ctx->h_bank[b][1 + k];
    out_host[4 + n_planes] = (int64_t)ctx->h_bank[b][0];
    ctx->bank_head ^= 1;
    --ctx->bank_pending;
    return PCL_OK;
}

int pcl_store_last_scatter_hits(pcl_ctx *ctx, int64_t *hits_out) {
    PCL_TRY(need_store_raw(ctx));
    if (!hits_out) return fail(PCL_ERR_ARG, "hits_out is NULL");
    if (ctx->last_async_bank >= 0) { // the most recent step was an asynchronous fused one: its bank holds the hits
        PCL_HIP(hipEventSynchronize(ctx->bank_ev[ctx->last_async_bank]));
        *hits_out = (int64_t)ctx->h_bank[ctx->last_async_bank][0];
        return PCL_OK;
    }
    if (!ctx->hits_on_host) {
        PCL_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        ctx->hits_on_host = true;
    }
    *hits_out = (int64_t)ctx->h_cnt[0];
    return PCL_OK;
}

int pcl_step_scatter_delete(pcl_ctx *ctx, double A, double n, int rng_mode, uint64_t seed, uint32_t step,
                            int64_t *n_alive_out, int64_t *n_removed_out) {
    PCL_TRY(need_store(ctx));
    if (rng_mode != PCL_RNG_INPUT && rng_mode != PCL_RNG_PHILOX) return fail(PCL_ERR_ARG, "unknown rng_mode %d", rng_mode);
    const int64_t N = ctx->count;
    if (n_alive_out) *n_alive_out = N;
    if (n_removed_out) *n_removed_out = 0;
    if (N == 0) return PCL_OK;
    if (rng_mode == PCL_RNG_INPUT && (!ctx->rnd[2] || ctx->rnd_n[2] < N))
        return fail(PCL_ERR_STATE, "PCL_RNG_INPUT needs pcl_store_upload_rand(which=2) for all %lld particles", (long long)N);
    PCL_TRY(ensure_scratch(ctx, N));
    PCL_TRY(ensure_alt(ctx));
    const int tiles = (int)div_up(N, kTile);
    PCL_TRY(PCL_DISPATCH(ctx, delete_mask_t<double>(ctx, A, n, rng_mode, seed, step),
                         delete_mask_t<float>(ctx, A, n, rng_mode, seed, step)));
    PCL_TRY(scan_tiles(ctx, N));
    compact_args ca{};
    for (int f = 0; f < PCL_NFIELDS; ++f) {
        ca.src[f] = ctx->field[f];
        ca.dst[f] = ctx->field_alt[f];
    }
    ca.ids_src = ctx->ids_iota ? nullptr : ctx->ids;
    ca.ids_dst = ctx->ids_alt;
    ca.ksrc = ctx->kind;
    ca.kdst = ctx->kind ? ctx->kind_alt : nullptr;
    ca.masks = ctx->masks;
    ca.tile_off = ctx->tile_off;
    ca.id_base = ctx->id_base;
    ca.N = N;
    ca.ts = tile_stride(ctx);
    const int ps_cmp = prof_begin(ctx, PCL_PROF_COMPACT);
    if (ctx->dtype == PCL_DTYPE_F64)
        hipLaunchKernelGGL((k_compact<uint64_t, PCL_NFIELDS>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca);
    else
        hipLaunchKernelGGL((k_compact<uint32_t, PCL_NFIELDS>), dim3(tiles), dim3(kBlock), 0, ctx->stream, ca);
    prof_end(ctx, ps_cmp);
    PCL_TRY(launch_check("k_compact"));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t alive = (int64_t)ctx->h_cnt[kCounterSlots - 1];
    if (alive < 0 || alive > N)
        return fail(PCL_ERR_HIP, "compaction produced an impossible count %lld of %lld", (long long)alive, (long long)N);
    ctx->compact_dv_mode = kDvMove;
    adopt_compacted(ctx, alive, N); // every field row was written into the other slab
    if (n_alive_out) *n_alive_out = alive;
    if (n_removed_out) *n_removed_out = N - alive;
    return PCL_OK;
}

int pcl_step_fused_delete(pcl_ctx *ctx, double dt, double A, double n, int flags, int rng_mode, uint64_t seed,
                          uint32_t step, const double *planes_host, int n_planes, int64_t *out_host) {
    if (flags & ~PCL_FUSED_LAZY) return fail(PCL_ERR_ARG, "unknown flag bits");
    const bool lazy = flags & PCL_FUSED_LAZY;
    PCL_TRY(need_store_raw(ctx, true)); // (bodies worked out ahead stay: fused_delete_alive decides whether this call is theirs)
    // every check comes before the first change of the store's state: a refused call leaves it as it was
    if (rng_mode != PCL_RNG_INPUT && rng_mode != PCL_RNG_PHILOX) return fail(PCL_ERR_ARG, "unknown rng_mode %d", rng_mode);
    if (n_planes < -1 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [-1, %d]", PCL_MAX_PLANES);
    if (n_planes > 0 && !planes_host) return fail(PCL_ERR_ARG, "planes_host is NULL");
    const int64_t N = ctx->count;
    const int np = n_planes > 0 ? n_planes : 0;
    if (out_host)
        for (int k = 0; k < 5 + np; ++k) out_host[k] = 0;
    if (N == 0) return PCL_OK;
    if (rng_mode == PCL_RNG_INPUT && (!ctx->rnd[2] || ctx->rnd_n[2] < N))
        return fail(PCL_ERR_STATE, "PCL_RNG_INPUT needs pcl_store_upload_rand(which=2) for all %lld particles", (long long)N);
    static const bool onepass = getenv("PCL_ONEPASS") != nullptr;
    if (lazy && !ctx->kind && rng_mode == PCL_RNG_PHILOX && !onepass && alive_enabled()) {
        // the usual case of a delete run: the body runs on the alive mask -- one kernel, nothing moves -- and the
        // store is compacted only when fewer than half of its slots are alive (k_delete_alive)
        ctx->lazy_dr = ctx->lazy_dr_vprev = false; // an implicit dr is superseded by this body's move
        int64_t alive = 0;
        PCL_TRY(fused_delete_alive(ctx, dt, A, n, seed, step, planes_host, n_planes, &alive));
        ctx->lazy_dr = true; // dr = v*dt with the (unchanged) velocities of the survivors
        ctx->lazy_dt = dt;
        if (out_host) {
            out_host[0] = alive;
            if (n_planes >= 0)
                for (int k = 0; k < 3 + np; ++k) out_host[1 + k] = (int64_t)ctx->h_cnt[1 + k];
            out_host[4 + np] = N - alive;
        }
        return PCL_OK;
    }
    PCL_TRY(ahead_commit(ctx));
    PCL_TRY(densify(ctx));
    PCL_TRY(ensure_scratch(ctx, N));
    PCL_TRY(ensure_alt(ctx));
    // A still-implicit dv (lazy scatter step): an all-photon store keeps it implicit -- the vprev rows travel through the
    // compaction in place of the dv rows -- anything else makes it real before the state is moved.  An implicit dr is
    // simply superseded by this step's Newton move.
    int dv_mode = kDvMove;
    PCL_TRY(decide_dv_mode(ctx, lazy, &dv_mode));
    ctx->lazy_dr = ctx->lazy_dr_vprev = false;
    int64_t alive = 0;
    // EXPERIMENT (PCL_ONEPASS=1): one kernel for the whole loop body (k_delete_onepass, decoupled look-back).  Bit-identical,
    // but measured slower than the pipeline below on this chip (0.20 vs 0.5 of peak at 1e8 photons: with ~1000 units in
    // flight a unit's look-back walks ~16 windows of 64 predecessors at ~2 us of cross-XCD latency each, as long as the
    // unit's own memory traffic takes), so it is not the default.  Falls back to the pipeline should a look-back give up.
    bool done = false;
    if (lazy && !ctx->kind && onepass) {
        PCL_HIP(hipMemsetAsync(ctx->d_cnt + 1, 0, (size_t)(3 + np) * sizeof(uint64_t), ctx->stream));
        PCL_TRY(PCL_DISPATCH(ctx, fused_delete_onepass_t<double>(ctx, dt, A, n, dv_mode, rng_mode, seed, step, planes_host, n_planes),
                             fused_delete_onepass_t<float>(ctx, dt, A, n, dv_mode, rng_mode, seed, step, planes_host, n_planes)));
        if (n_planes >= 0)
            PCL_HIP(hipMemcpyAsync(ctx->h_cnt + 1, ctx->d_cnt + 1, (size_t)(3 + np) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->h_cnt[kCounterSlots - 3] == 0) { // no look-back gave up
            alive = (int64_t)ctx->h_cnt[kCounterSlots - 1];
            if (alive < 0 || alive > N)
                return fail(PCL_ERR_HIP, "one-pass compaction produced an impossible count %lld of %lld", (long long)alive, (long long)N);
            done = true;
        }
    }
    if (!done)
        PCL_TRY(PCL_DISPATCH(ctx, fused_delete_t<double>(ctx, dt, A, n, lazy, dv_mode, rng_mode, seed, step, planes_host, n_planes),
                             fused_delete_t<float>(ctx, dt, A, n, lazy, dv_mode, rng_mode, seed, step, planes_host, n_planes)));
    if (!done && n_planes >= 0 && out_host) {
        // the measure counters come out of pass 3 itself: wait for it
        PCL_HIP(hipMemcpyAsync(ctx->h_cnt + 1, ctx->d_cnt + 1, (size_t)(3 + np) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (!done) PCL_TRY(wait_count(ctx, N, &alive)); // the count alone is ready before pass 3 has finished
    adopt_compacted(ctx, alive, N);      // (lazy: the dr rows of the new slab are stale -- dr is implicit anyway)
    if (lazy) {
        ctx->lazy_dr = true; // dr = v*dt with the (unchanged) velocities of the survivors
        ctx->lazy_dt = dt;
    }
    if (out_host) {
        out_host[0] = alive;
        if (n_planes >= 0)
            for (int k = 0; k < 3 + np; ++k) out_host[1 + k] = (int64_t)ctx->h_cnt[1 + k];
        out_host[4 + np] = N - alive;
    }
    return PCL_OK;
}

int pcl_step_fused_delete_multi(pcl_ctx *ctx, double dt, int k_steps, double A, double n, uint64_t seed, uint32_t step0,
                                const double *planes_host, int n_planes, int64_t *out_host) {
    PCL_TRY(need_store_raw(ctx, true)); // (bodies worked out ahead stay: they may be this call's)
    if (k_steps < 1 || k_steps > PCL_MULTI_MAX) return fail(PCL_ERR_ARG, "k_steps outside [1, %d]", PCL_MULTI_MAX);
    if (n_planes < -1 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [-1, %d]", PCL_MAX_PLANES);
    if (n_planes > 0 && !planes_host) return fail(PCL_ERR_ARG, "planes_host is NULL");
    if (ctx->bank_pending) return fail(PCL_ERR_STATE, "un-read asynchronous fused steps are outstanding");
    // An all-photon store takes the path of the single calls, body by body, with the number of bodies still to come as a
    // promise: one k_delete_ahead(_live) launch works out as many as a launch holds and the rest of them are answered from
    // its rows -- the same sweeps a loop of pcl_step_fused_delete calls gets, without the calls.  (PCL_MULTI_AHEAD=0: the
    // K-step flag kernel + compaction below, which stores with kind bytes always take.)
    static knob k_via("PCL_MULTI_AHEAD");
    static const bool onepass = getenv("PCL_ONEPASS") != nullptr;
    if (!ctx->kind && alive_enabled() && ahead_k() > 0 && !k_via.off() && !onepass) {
        const int np = n_planes > 0 ? n_planes : 0;
        if (out_host)
            for (int k = 0; k < k_steps * (5 + np); ++k) out_host[k] = 0;
        for (int k = 0; k < k_steps && ctx->count > 0; ++k) {
            const int64_t before = ctx->count;
            int64_t alive = 0;
            ctx->lazy_dr = ctx->lazy_dr_vprev = false; // an implicit dr is superseded by this body's move
            PCL_TRY(fused_delete_alive(ctx, dt, A, n, seed, step0 + (uint32_t)k, planes_host, n_planes, &alive, k_steps - k));
            ctx->lazy_dr = true; // dr = v*dt with the (unchanged) velocities of the survivors
            ctx->lazy_dt = dt;
            if (out_host) {
                int64_t *o = out_host + (int64_t)k * (5 + np);
                o[0] = alive;
                if (n_planes >= 0)
                    for (int j = 0; j < 3 + np; ++j) o[1 + j] = (int64_t)ctx->h_cnt[1 + j];
                o[4 + np] = before - alive;
            }
        }
        return PCL_OK;
    }
    PCL_TRY(ahead_commit(ctx));
    PCL_TRY(densify(ctx));
    const int64_t N = ctx->count;
    const int np = n_planes > 0 ? n_planes : 0, nslots = 4 + np;
    if (out_host)
        for (int k = 0; k < k_steps * (5 + np); ++k) out_host[k] = 0;
    if (N == 0) return PCL_OK;
    PCL_TRY(ensure_scratch(ctx, N));
    PCL_TRY(ensure_alt(ctx));
    int dv_mode = kDvMove;                                   // see pcl_step_fused_delete
    PCL_TRY(decide_dv_mode(ctx, true, &dv_mode));
    ctx->lazy_dr = ctx->lazy_dr_vprev = false;               // an implicit dr is superseded by these steps' own moves
    PCL_HIP(hipMemsetAsync(ctx->d_multi, 0, (size_t)k_steps * nslots * sizeof(uint64_t), ctx->stream));
    PCL_TRY(PCL_DISPATCH(ctx, fused_delete_multi_t<double>(ctx, dt, k_steps, A, n, dv_mode, seed, step0, planes_host, n_planes),
                         fused_delete_multi_t<float>(ctx, dt, k_steps, A, n, dv_mode, seed, step0, planes_host, n_planes)));
    // the per-step rows were complete when pass 1 ended: their copy was enqueued behind the compaction only to keep
    // one wait; the count event fires before pass 3
    PCL_HIP(hipMemcpyAsync(ctx->h_multi, ctx->d_multi, (size_t)k_steps * nslots * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    PCL_TRY(stream_wait(ctx));
    int64_t alive = 0;
    PCL_TRY(wait_count(ctx, N, &alive));
    if (alive != (int64_t)ctx->h_multi[(k_steps - 1) * nslots])
        return fail(PCL_ERR_HIP, "compaction kept %lld particles, the last step counted %lld", (long long)alive,
                    (long long)ctx->h_multi[(k_steps - 1) * nslots]);
    adopt_compacted(ctx, alive, -1); // the masks describe K steps at once: no per-step flag array to hand out
    ctx->lazy_dr = true;             // dr = v*dt with the (unchanged) velocities of the survivors
    ctx->lazy_dt = dt;
    if (out_host) {
        int64_t before = N;
        for (int k = 0; k < k_steps; ++k) {
            int64_t *o = out_host + (int64_t)k * (5 + np);
            const uint64_t *c = ctx->h_multi + (int64_t)k * nslots;
            o[0] = (int64_t)c[0];
            for (int j = 0; j < 3 + np; ++j) o[1 + j] = (int64_t)c[1 + j];
            o[4 + np] = before - (int64_t)c[0];
            before = (int64_t)c[0];
        }
    }
    return PCL_OK;
}

int pcl_step_mixed_multi(pcl_ctx *ctx, double dt, int k_passes, int n_phases, const int *phase_kinds_host, double A, double n,
                         int flags, double c, double h, const char *n_expr, double A_del, double n_del, uint64_t seed,
                         uint32_t step0, const double *planes_host, int n_planes, int64_t *out_host) {
    PCL_TRY(need_store_raw(ctx));
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N)) return fail(PCL_ERR_ARG, "unknown flag bits");
    if (n_phases < 1 || n_phases > PCL_MIXED_MAXPH || !phase_kinds_host)
        return fail(PCL_ERR_ARG, "n_phases outside [1, %d]", PCL_MIXED_MAXPH);
    if (k_passes < 1 || k_passes * n_phases > PCL_MULTI_MAX)
        return fail(PCL_ERR_ARG, "k_passes * n_phases outside [1, %d]", PCL_MULTI_MAX);
    if (n_planes < 0 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [0, %d]", PCL_MAX_PLANES);
    if (n_planes > 0 && !planes_host) return fail(PCL_ERR_ARG, "planes_host is NULL");
    if (ctx->bank_pending) return fail(PCL_ERR_STATE, "un-read asynchronous fused steps are outstanding");
    int phase_del[PCL_MIXED_MAXPH] = {0, 0};
    int n_iso = 0, n_delete = 0, last_iso_in_pass = -1;
    for (int j = 0; j < n_phases; ++j) {
        if (phase_kinds_host[j] != PCL_PHASE_ISOTROPIC && phase_kinds_host[j] != PCL_PHASE_DELETE)
            return fail(PCL_ERR_ARG, "unknown phase kind %d", phase_kinds_host[j]);
        phase_del[j] = phase_kinds_host[j] == PCL_PHASE_DELETE;
        if (phase_del[j]) {
            ++n_delete;
        } else {
            ++n_iso;
            last_iso_in_pass = j;
        }
    }
    if (n_iso > 1 || n_delete > 1) return fail(PCL_ERR_ARG, "at most one isotropic and one delete phase per pass");
    const bool has_iso = n_iso > 0, has_delete = n_delete > 0;
    // Loops with a delete phase on all-photon stores: the kernel writes a wave's survivors back to the front of the wave's own
    // 512-slot segment of the tile (stable, every field they own), so the launch needs NO compaction pass behind it -- the store
    // stays behind an alive mask whose set bits are a prefix of every segment, the next launch of this entry point takes it as
    // it is (its waves skip the empty rows), and the global compaction (scan + k_compact_*) only runs once fewer than half of the
    // slots are alive.  Round 5 compacted after every launch: 30 % of configs[4]'s GPU time to drop 6 % of the slots
    // (profiles/r05_driver_cmd_pmc.md).  PCL_MIXED_INPLACE=0: that form (A/B, and what stores with plain Objects take).
    static knob k_inpl("PCL_MIXED_INPLACE");
    const bool inplace = has_delete && !ctx->kind && !k_inpl.off() && (has_iso || ctx->lazy_dv || ctx->dv_zero == 1);
    if (!(inplace && ctx->holes && ctx->seg_prefix && ctx->pend_n == 0)) PCL_TRY(densify(ctx));
    const bool use_e = has_iso && (flags & PCL_SCATTER_WAVELENGTH), var_n = has_iso && (flags & PCL_SCATTER_VARIABLE_N);
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, ctx->dtype == PCL_DTYPE_F32 ? 1 : 0, use_e, &ent));
    const int np = n_planes, n_rows = k_passes * n_phases, nslots = 5 + np;
    const int64_t N = ctx->count, extent = ctx->holes ? ctx->slots : ctx->count;
    if (out_host)
        for (int k = 0; k < n_rows * (5 + np); ++k) out_host[k] = 0;
    if (N == 0) return PCL_OK;
    // a pending implicit dv: superseded by the pass's own scatter phase; a pass without one (delete only) carries it
    // through the compaction (all-photon stores) or makes it real first
    if (has_delete) {
        PCL_TRY(ensure_scratch(ctx, extent, true)); // (a store behind its segment-prefix mask stays as it is)
        PCL_TRY(ensure_alt(ctx));
    }
    if (!has_iso && ctx->lazy_dv && ctx->kind) PCL_TRY(materialize(ctx));
    ctx->lazy_dr = ctx->lazy_dr_vprev = false; // superseded by the pass's own moves
    PCL_HIP(hipMemsetAsync(ctx->d_multi, 0, (size_t)n_rows * nslots * sizeof(uint64_t), ctx->stream));
    ctx->hits_on_host = false;
    ctx->last_async_bank = -1;
    const int last_iso = has_iso ? (k_passes - 1) * n_phases + last_iso_in_pass : -1;
    PCL_TRY(PCL_DISPATCH(ctx,
                         step_mixed_t<double>(ctx, dt, k_passes, n_phases, phase_del, A, n, use_e, var_n, ent, c, h, A_del, n_del,
                                              seed, step0, planes_host, np, has_delete, last_iso, inplace),
                         step_mixed_t<float>(ctx, dt, k_passes, n_phases, phase_del, A, n, use_e, var_n, ent, c, h, A_del, n_del,
                                             seed, step0, planes_host, np, has_delete, last_iso, inplace)));
    PCL_HIP(hipMemcpyAsync(ctx->h_multi, ctx->d_multi, (size_t)n_rows * nslots * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    // what is implicit now: dr = (velocity of the last move) * dt; dv = v - vprev once a scatter phase has run
    if (has_iso) ctx->lazy_dv = true;
    ctx->lazy_dr = true;
    ctx->lazy_dr_vprev = has_iso && !phase_del[n_phases - 1]; // last phase scattered: its move used what is now vprev
    ctx->lazy_dt = dt;
    if (has_delete && !inplace) {
        bool has_dr = false;
        int dv_mode = (ctx->lazy_dv && !ctx->kind) ? kDvVprev : kDvMove;
        if (ctx->lazy_dv && ctx->kind) { // plain Objects keep real dv rows: make the photons' real too, move everything
            PCL_TRY(materialize(ctx));
            has_dr = true;
        }
        PCL_TRY(PCL_DISPATCH(ctx, compact_after_pass_t<double>(ctx, dt, has_dr, dv_mode),
                             compact_after_pass_t<float>(ctx, dt, has_dr, dv_mode)));
    }
    PCL_TRY(stream_wait(ctx));
    if (has_delete && !inplace) {
        int64_t alive = 0;
        PCL_TRY(wait_count(ctx, N, &alive));
        if (alive != (int64_t)ctx->h_multi[(n_rows - 1) * nslots])
            return fail(PCL_ERR_HIP, "compaction kept %lld particles, the last phase counted %lld", (long long)alive,
                        (long long)ctx->h_multi[(n_rows - 1) * nslots]);
        adopt_compacted(ctx, alive, -1);
    }
    if (has_delete && inplace) { // the store is behind its alive mask now, a prefix of set bits per 512-slot segment, ids explicit
        const int64_t alive = (int64_t)ctx->h_multi[(n_rows - 1) * nslots];
        if (alive < 0 || alive > N) return fail(PCL_ERR_HIP, "the last phase counted an impossible %lld of %lld particles", (long long)alive, (long long)N);
        ctx->holes = alive > 0;
        ctx->seg_prefix = alive > 0;
        ctx->slots = alive > 0 ? extent : 0;
        ctx->count = alive;
        ctx->pend_n = 0;
        ctx->ids_iota = false;
        ctx->last_delete_n = -1;
        ctx->last_delete_masked = false;
        if (!(use_e && ctx->lam4_valid)) ctx->lam4_valid = false; // (the cache travelled with the photons only when the launch used it)
        static knob k_ratio("PCL_MIXED_COMPACT_BELOW");               // compact once fewer than this share of the slots is alive (default 0.5)
        if (alive > 0 && (double)alive < k_ratio.value(0.5) * (double)extent) PCL_TRY(densify(ctx));
    }
    if (has_iso) // pcl_store_last_scatter_hits() reports the last scatter phase
        ctx->h_cnt[0] = ctx->h_multi[(int64_t)((k_passes - 1) * n_phases + last_iso_in_pass) * nslots + 1], ctx->hits_on_host = true;
    if (has_iso) { // what the next launch's form goes by (step_mixed_t)
        const uint64_t *row = ctx->h_multi + (int64_t)((k_passes - 1) * n_phases + last_iso_in_pass) * nslots;
        ctx->mixed_last_h = row[0] ? (double)row[1] / (double)row[0] : -1.0;
    }
    if (out_host)
        for (int k = 0; k < n_rows; ++k) {
            int64_t *o = out_host + (int64_t)k * (5 + np);
            const uint64_t *cr = ctx->h_multi + (int64_t)k * nslots;
            o[0] = (int64_t)cr[0];
            for (int j = 0; j < 3 + np; ++j) o[1 + j] = (int64_t)cr[2 + j];
            o[4 + np] = (int64_t)cr[1];
        }
    return PCL_OK;
}

int pcl_store_trace_ahead(pcl_ctx *ctx, const int64_t *ids_host, int n_ids, double dt, int k_passes, int n_phases,
                          const int *phase_kinds_host, int record_phase, double A, double n, int flags, double c, double h,
                          const char *n_expr, double A_del, double n_del, uint64_t seed, uint32_t step0, double *out_host) {
    PCL_TRY(need_store_raw(ctx)); // (bodies worked out ahead of their calls are committed: the trace starts from the state they left)
    if (flags & ~(PCL_SCATTER_WAVELENGTH | PCL_SCATTER_VARIABLE_N)) return fail(PCL_ERR_ARG, "unknown flag bits");
    if (n_phases < 1 || n_phases > PCL_MIXED_MAXPH || !phase_kinds_host)
        return fail(PCL_ERR_ARG, "n_phases outside [1, %d]", PCL_MIXED_MAXPH);
    if (k_passes < 1 || k_passes * n_phases > PCL_MULTI_MAX)
        return fail(PCL_ERR_ARG, "k_passes * n_phases outside [1, %d]", PCL_MULTI_MAX);
    if (record_phase < 0 || record_phase >= n_phases) return fail(PCL_ERR_ARG, "record_phase outside [0, n_phases)");
    if (n_ids < 0 || n_ids > PCL_TRACE_MAX) return fail(PCL_ERR_ARG, "n_ids outside [0, %d]", PCL_TRACE_MAX);
    if (n_ids > 0 && !ids_host) return fail(PCL_ERR_ARG, "ids_host is NULL");
    int phase_del[PCL_MIXED_MAXPH] = {0, 0};
    bool has_iso = false;
    for (int j = 0; j < n_phases; ++j) {
        if (phase_kinds_host[j] != PCL_PHASE_ISOTROPIC && phase_kinds_host[j] != PCL_PHASE_DELETE)
            return fail(PCL_ERR_ARG, "unknown phase kind %d", phase_kinds_host[j]);
        phase_del[j] = phase_kinds_host[j] == PCL_PHASE_DELETE;
        has_iso = has_iso || !phase_del[j];
    }
    for (int j = 1; j < n_ids; ++j)
        if (ids_host[j] <= ids_host[j - 1]) return fail(PCL_ERR_ARG, "ids_host must be strictly ascending");
    const bool use_e = has_iso && (flags & PCL_SCATTER_WAVELENGTH), var_n = has_iso && (flags & PCL_SCATTER_VARIABLE_N);
    rtc_entry *ent = nullptr;
    if (var_n) PCL_TRY(get_rtc(ctx, n_expr, ctx->dtype == PCL_DTYPE_F32 ? 1 : 0, use_e, &ent));
    if (n_ids == 0) return PCL_OK;
    // the tracked ids go over once per set, not once per launch
    if (ctx->trace_want_cap < n_ids) {
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        dev_free(ctx->trace_want);
        dev_free(ctx->trace_slot);
        ctx->trace_want_cap = 0;
        ctx->trace_ids.clear();
        PCL_TRY(dev_alloc(&ctx->trace_want, n_ids));
        PCL_TRY(dev_alloc(&ctx->trace_slot, n_ids));
        ctx->trace_want_cap = n_ids;
    }
    if ((int64_t)ctx->trace_ids.size() != n_ids || memcmp(ctx->trace_ids.data(), ids_host, (size_t)n_ids * sizeof(int64_t)) != 0) {
        PCL_HIP(hipStreamSynchronize(ctx->stream)); // (an earlier trace may still be reading the old set)
        ctx->trace_ids.assign(ids_host, ids_host + n_ids);
        PCL_HIP(hipMemcpyAsync(ctx->trace_want, ctx->trace_ids.data(), (size_t)n_ids * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    const int64_t n_out = (int64_t)k_passes * n_ids * 4;
    if (ctx->trace_out_cap < n_out) {
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->trace_out) (void)hipHostFree(ctx->trace_out);
        ctx->trace_out = nullptr;
        ctx->trace_out_cap = 0;
        PCL_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->trace_out), (size_t)n_out * sizeof(double)));
        ctx->trace_out_cap = n_out;
    }
    PCL_TRY(PCL_DISPATCH(ctx,
                         trace_ahead_t<double>(ctx, n_ids, dt, k_passes, n_phases, phase_del, record_phase, A, n, use_e, var_n, ent, c, h,
                                               A_del, n_del, seed, step0),
                         trace_ahead_t<float>(ctx, n_ids, dt, k_passes, n_phases, phase_del, record_phase, A, n, use_e, var_n, ent, c, h,
                                              A_del, n_del, seed, step0)));
    ctx->trace_out_n = n_out;
    if (!out_host) return PCL_OK; // enqueued only: the rows are read (pcl_store_trace_read) behind the launch they belong to
    return pcl_store_trace_read(ctx, out_host, n_out);
}

int pcl_store_trace_read(pcl_ctx *ctx, double *out_host, int64_t n_doubles) {
    PCL_TRY(bind(ctx));
    if (!out_host || n_doubles != ctx->trace_out_n || n_doubles <= 0)
        return fail(PCL_ERR_ARG, "pcl_store_trace_read: %lld doubles asked for, the last pcl_store_trace_ahead left %lld", (long long)n_doubles,
                    (long long)ctx->trace_out_n);
    PCL_TRY(stream_wait(ctx)); // (no wait at all behind a K-pass launch that has returned its rows: the stream is in order)
    memcpy(out_host, ctx->trace_out, (size_t)n_doubles * sizeof(double));
    ctx->trace_out_n = 0;
    return PCL_OK;
}

int pcl_store_last_delete_flags(pcl_ctx *ctx, int32_t *flags_host, int64_t n) {
    PCL_TRY(need_store_raw(ctx));
    if (ctx->last_delete_n < 0) return fail(PCL_ERR_STATE, "no delete step has run since the store last changed");
    if (n != ctx->last_delete_n || !flags_host)
        return fail(PCL_ERR_ARG, "flags_host must hold exactly the pre-delete count %lld", (long long)ctx->last_delete_n);
    if (ctx->last_delete_masked) {
        // the last delete ran on the alive mask: the photons it saw are the set bits of masks_prev, in slot order, and it
        // removed those whose bit in masks is clear.  Two small arrays (a bit per slot) expanded on the host; the store
        // is not touched (no compaction is forced by looking).
        const int64_t words = div_up(ctx->last_delete_slots, kTile) * kTileRows;
        std::vector<uint64_t> prev((size_t)words), cur((size_t)words);
        PCL_TRY(pcl_d2h(ctx, prev.data(), ctx->masks_prev, words * (int64_t)sizeof(uint64_t)));
        PCL_TRY(pcl_d2h(ctx, cur.data(), ctx->masks, words * (int64_t)sizeof(uint64_t)));
        int64_t k = 0;
        for (int64_t w = 0; w < words; ++w) {
            uint64_t m = prev[(size_t)w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                if (k < n) flags_host[k] = ((cur[(size_t)w] >> b) & 1ull) ? 0 : 1;
                ++k;
            }
        }
        if (k != n) return fail(PCL_ERR_HIP, "the alive masks describe %lld photons, expected %lld", (long long)k, (long long)n);
        return PCL_OK;
    }
    PCL_TRY(need_store(ctx));
    int32_t *d = nullptr;
    PCL_TRY(dev_alloc(&d, n));
    hipLaunchKernelGGL(k_masks_to_flags, dim3(grid_for(ctx, n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->masks, d, n);
    int rc = launch_check("k_masks_to_flags");
    if (rc == PCL_OK) rc = pcl_d2h(ctx, flags_host, d, n * (int64_t)sizeof(int32_t));
    dev_free(d);
    return rc;
}

int pcl_step_plane_energies(pcl_ctx *ctx, const double *plane_host, void *E_out_host, int64_t cap, int64_t *n_out) {
    PCL_TRY(need_store(ctx)); // reads the real dr: a lazy step's implicit one is materialised first
    if (!plane_host || !n_out || cap < 0 || (cap > 0 && !E_out_host)) return fail(PCL_ERR_ARG, "bad argument");
    *n_out = 0;
    const int64_t N = ctx->count;
    if (N == 0) return PCL_OK;
    const int ax = !std::isnan(plane_host[0]) ? 0 : (!std::isnan(plane_host[1]) ? 1 : 2); // light.py:385-396
    PCL_TRY(ensure_scratch(ctx, N));
    ctx->last_delete_n = -1; // scratch masks no longer describe a delete
    if (ctx->e_out_cap < N) {
        dev_free(ctx->e_out);
        ctx->e_out_cap = 0;
        PCL_TRY(dev_alloc_bytes(&ctx->e_out, ctx->capacity, ctx->esz));
        ctx->e_out_cap = ctx->capacity;
    }
    PCL_TRY(PCL_DISPATCH(ctx, plane_energies_t<double>(ctx, ax, plane_host[ax]), plane_energies_t<float>(ctx, ax, plane_host[ax])));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n = (int64_t)ctx->h_cnt[kCounterSlots - 1];
    if (n < 0 || n > N) return fail(PCL_ERR_HIP, "gather produced an impossible count %lld of %lld", (long long)n, (long long)N);
    *n_out = n;
    const int64_t take = n < cap ? n : cap;
    if (take > 0) PCL_TRY(pcl_d2h(ctx, E_out_host, ctx->e_out, take * (int64_t)ctx->esz));
    return PCL_OK;
}

int pcl_step_counters(pcl_ctx *ctx, const double *planes_host, int n_planes, int64_t *out_host) {
    PCL_TRY(need_store(ctx));
    if (n_planes < 0 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "n_planes outside [0, %d]", PCL_MAX_PLANES);
    if (!out_host || (n_planes > 0 && !planes_host)) return fail(PCL_ERR_ARG, "NULL argument");
    const int64_t N = ctx->count;
    const int nc = 3 + n_planes;
    out_host[PCL_CNT_N] = N;
    for (int k = 0; k < nc; ++k) out_host[1 + k] = 0;
    if (N == 0) return PCL_OK;
    PCL_HIP(hipMemsetAsync(ctx->d_cnt + 1, 0, (size_t)nc * sizeof(uint64_t), ctx->stream));
    PCL_TRY(PCL_DISPATCH(ctx, counters_t<double>(ctx, planes_host, n_planes), counters_t<float>(ctx, planes_host, n_planes)));
    // slot 0 (hits of the last scatter step) rides along, so reading it later costs no extra sync
    PCL_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, (size_t)(1 + nc) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    ctx->hits_on_host = true;
    for (int k = 0; k < nc; ++k) out_host[1 + k] = (int64_t)ctx->h_cnt[1 + k];
    return PCL_OK;
}

} // extern "C"

// =================================================================================================
// Device groups: several GPUs from ONE process (SURVEY.md 8(b) ``pcl_init(n_dev, dev_ids)``, 8(e))
// =================================================================================================
// The reference is one process with one simulation thread (physicl/__init__.py:400-432).  A group owns one context
// (HIP device, stream, store) and one worker thread per entry of its device list; particles are sharded by global index
// in contiguous blocks (shard g of G owns [g*N/G, (g+1)*N/G), ids global, so the id-keyed RNG gives the rows of a
// one-device run); a group call hands the same step to every worker, waits for all, and SUMS the int64 counter rows
// the shards return -- they are in host memory when a launch returns, so that sum is the group's collective.
struct pcl_group {
    int dtype = PCL_DTYPE_F64; // element type of the shards' stores (pcl_group_store_alloc)
    std::vector<pcl_ctx *> ctx;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<int(int, pcl_ctx *)> job; // guarded by mu
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> rc;
    std::vector<std::string> err;
};

namespace {

void group_worker(pcl_group *g, int i) {
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int, pcl_ctx *)> job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_go.wait(lk, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
            job = g->job;
        }
        const int rc = job(i, g->ctx[i]);
        const std::string err = rc != PCL_OK ? g_err : std::string(); // (thread-local: this worker's last error)
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rc[i] = rc;
            g->err[i] = err;
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

// job(i, ctx) on every shard, concurrently; the first failure (lowest shard) is the call's result and error text
int group_run(pcl_group *g, std::function<int(int, pcl_ctx *)> job) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    std::unique_lock<std::mutex> lk(g->mu);
    g->job = std::move(job);
    g->pending = (int)g->ctx.size();
    ++g->generation;
    g->cv_go.notify_all();
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->rc[i] != PCL_OK) return fail(g->rc[i], "shard %d: %s", (int)i, g->err[i].c_str());
    return PCL_OK;
}

inline void shard_of(int64_t n, int i, int G, int64_t *lo, int64_t *hi) {
    *lo = n * i / G;
    *hi = n * (i + 1) / G;
}

// sum of the shards' int64 rows
int group_sum(pcl_group *g, int64_t *out_host, size_t n, const std::vector<std::vector<int64_t>> &part) {
    if (!out_host) return PCL_OK;
    for (size_t k = 0; k < n; ++k) {
        int64_t t = 0;
        for (const auto &p : part) t += p[k];
        out_host[k] = t;
    }
    return PCL_OK;
}

} // namespace

extern "C" {

int pcl_group_create(int n_dev, const int *device_ids, pcl_group **group_out) {
    if (n_dev < 1 || n_dev > 64 || !device_ids || !group_out) return fail(PCL_ERR_ARG, "n_dev outside [1, 64] or NULL argument");
    std::unique_ptr<pcl_group> g(new (std::nothrow) pcl_group);
    if (!g) return fail(PCL_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_dev; ++i) {
        pcl_ctx *c = nullptr;
        const int rc = pcl_ctx_create(device_ids[i], nullptr, &c);
        if (rc != PCL_OK) {
            for (pcl_ctx *d : g->ctx) pcl_ctx_destroy(d);
            return rc;
        }
        g->ctx.push_back(c);
    }
    g->rc.assign(n_dev, PCL_OK);
    g->err.assign(n_dev, std::string());
    for (int i = 0; i < n_dev; ++i) g->workers.emplace_back(group_worker, g.get(), i);
    *group_out = g.release();
    return PCL_OK;
}

int pcl_group_destroy(pcl_group *g) {
    if (!g) return PCL_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_go.notify_all();
    for (auto &w : g->workers) w.join();
    for (pcl_ctx *c : g->ctx) pcl_ctx_destroy(c);
    delete g;
    return PCL_OK;
}

int pcl_group_size(pcl_group *g, int *n_out) {
    if (!g || !n_out) return fail(PCL_ERR_ARG, "NULL argument");
    *n_out = (int)g->ctx.size();
    return PCL_OK;
}

int pcl_group_ctx(pcl_group *g, int i, pcl_ctx **ctx_out) {
    if (!g || !ctx_out || i < 0 || i >= (int)g->ctx.size()) return fail(PCL_ERR_ARG, "shard index outside the group");
    *ctx_out = g->ctx[i];
    return PCL_OK;
}

int pcl_group_shard(pcl_group *g, int64_t n_global, int i, int64_t *lo_out, int64_t *hi_out) {
    if (!g || !lo_out || !hi_out || i < 0 || i >= (int)g->ctx.size() || n_global < 0) return fail(PCL_ERR_ARG, "bad argument");
    shard_of(n_global, i, (int)g->ctx.size(), lo_out, hi_out);
    return PCL_OK;
}

int pcl_group_store_alloc(pcl_group *g, int64_t capacity_global, int dtype) {
    if (!g || capacity_global <= 0) return fail(PCL_ERR_ARG, "capacity must be positive");
    const int G = (int)g->ctx.size();
    // ceil(capacity / G) slots per shard: the sizes of shard_of(n, i, G) are not monotone in n, so a later fill of fewer
    // photons must find room for the largest shard any n <= capacity can produce
    const int64_t per_shard = (capacity_global + G - 1) / G;
    g->dtype = dtype;
    return group_run(g, [=](int i, pcl_ctx *c) { return pcl_store_alloc_dtype(c, per_shard, dtype); });
}

int pcl_group_store_dtype(pcl_group *g, int *dtype_out) {
    if (!g || !dtype_out) return fail(PCL_ERR_ARG, "NULL argument");
    *dtype_out = g->dtype;
    return PCL_OK;
}

int pcl_group_fill_photons(pcl_group *g, int64_t n_global, int64_t id_base, double c_light, double e_min, double e_max, uint64_t seed) {
    if (!g || n_global < 0) return fail(PCL_ERR_ARG, "bad argument");
    const int G = (int)g->ctx.size();
    return group_run(g, [=](int i, pcl_ctx *c) {
        int64_t lo, hi;
        shard_of(n_global, i, G, &lo, &hi);
        return pcl_store_fill_photons(c, hi - lo, id_base + lo, c_light, e_min, e_max, seed);
    });
}

int pcl_group_count(pcl_group *g, int64_t *count_out) {
    if (!g || !count_out) return fail(PCL_ERR_ARG, "NULL argument");
    int64_t t = 0;
    for (pcl_ctx *c : g->ctx) t += c->count;
    *count_out = t;
    return PCL_OK;
}

int pcl_group_sync(pcl_group *g) {
    return group_run(g, [](int, pcl_ctx *c) { return pcl_ctx_sync(c); });
}

int pcl_group_reserve_compaction(pcl_group *g) {
    return group_run(g, [](int, pcl_ctx *c) { return pcl_store_reserve_compaction(c); });
}

int pcl_group_step_fused(pcl_group *g, double dt, int do_scatter, double A, double n, int flags, double c, double h, const char *n_expr,
                         int rng_mode, uint64_t seed, uint32_t step, const double *planes_host, int n_planes, int64_t *out_host) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    if (!out_host || n_planes < 0) return fail(PCL_ERR_ARG, "the group form is synchronous: out_host and n_planes >= 0 are required");
    const size_t w = 5 + (size_t)n_planes;
    std::vector<std::vector<int64_t>> part(g->ctx.size(), std::vector<int64_t>(w, 0));
    PCL_TRY(group_run(g, [&](int i, pcl_ctx *cx) {
        return pcl_step_fused(cx, dt, do_scatter, A, n, flags, c, h, n_expr, rng_mode, seed, step, planes_host, n_planes, part[i].data());
    }));
    return group_sum(g, out_host, w, part);
}

int pcl_group_step_fused_delete(pcl_group *g, double dt, double A, double n, int flags, int rng_mode, uint64_t seed, uint32_t step,
                                const double *planes_host, int n_planes, int64_t *out_host) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    const size_t w = 5 + (size_t)(n_planes > 0 ? n_planes : 0);
    std::vector<std::vector<int64_t>> part(g->ctx.size(), std::vector<int64_t>(w, 0));
    PCL_TRY(group_run(g, [&](int i, pcl_ctx *cx) {
        return pcl_step_fused_delete(cx, dt, A, n, flags, rng_mode, seed, step, planes_host, n_planes, part[i].data());
    }));
    return group_sum(g, out_host, w, part);
}

int pcl_group_step_fused_multi(pcl_group *g, double dt, int k_steps, double A, double n, int flags, double c, double h, const char *n_expr,
                               uint64_t seed, uint32_t step0, const double *planes_host, int n_planes, int64_t *out_host) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    if (k_steps < 1 || k_steps > PCL_MULTI_MAX || n_planes < 0 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "k_steps or n_planes out of range");
    const size_t w = (size_t)k_steps * (5 + (size_t)n_planes);
    std::vector<std::vector<int64_t>> part(g->ctx.size(), std::vector<int64_t>(w, 0));
    PCL_TRY(group_run(g, [&](int i, pcl_ctx *cx) {
        return pcl_step_fused_multi(cx, dt, k_steps, A, n, flags, c, h, n_expr, seed, step0, planes_host, n_planes, part[i].data());
    }));
    return group_sum(g, out_host, w, part);
}

int pcl_group_step_fused_delete_multi(pcl_group *g, double dt, int k_steps, double A, double n, uint64_t seed, uint32_t step0,
                                      const double *planes_host, int n_planes, int64_t *out_host) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    if (k_steps < 1 || k_steps > PCL_MULTI_MAX || n_planes < -1 || n_planes > PCL_MAX_PLANES) return fail(PCL_ERR_ARG, "k_steps or n_planes out of range");
    const size_t w = (size_t)k_steps * (5 + (size_t)(n_planes > 0 ? n_planes : 0));
    std::vector<std::vector<int64_t>> part(g->ctx.size(), std::vector<int64_t>(w, 0));
    PCL_TRY(group_run(g, [&](int i, pcl_ctx *cx) {
        return pcl_step_fused_delete_multi(cx, dt, k_steps, A, n, seed, step0, planes_host, n_planes, part[i].data());
    }));
    return group_sum(g, out_host, w, part);
}

int pcl_group_step_mixed_multi(pcl_group *g, double dt, int k_passes, int n_phases, const int *phase_kinds_host, double A, double n, int flags,
                               double c, double h, const char *n_expr, double A_del, double n_del, uint64_t seed, uint32_t step0,
                               const double *planes_host, int n_planes, int64_t *out_host) {
    if (!g) return fail(PCL_ERR_ARG, "group is NULL");
    if (k_passes < 1 || n_phases < 1 || k_passes * n_phases > PCL_MULTI_MAX || n_planes < 0 || n_planes > PCL_MAX_PLANES)
        return fail(PCL_ERR_ARG, "k_passes * n_phases or n_planes out of range");
    const size_t w = (size_t)k_passes * n_phases * (5 + (size_t)n_planes);
    std::vector<std::vector<int64_t>> part(g->ctx.size(), std::vector<int64_t>(w, 0));
    PCL_TRY(group_run(g, [&](int i, pcl_ctx *cx) {
        return pcl_step_mixed_multi(cx, dt, k_passes, n_phases, phase_kinds_host, A, n, flags, c, h, n_expr, A_del, n_del, seed, step0,
                                    planes_host, n_planes, part[i].data());
    }));
    return group_sum(g, out_host, w, part);
}

// a window [offset, offset + n) of the GLOBAL particle order, shard after shard (contiguous blocks + stable compaction)
static int group_window(pcl_group *g, int64_t offset, int64_t n, size_t esz, char *host,
                        const std::function<int(pcl_ctx *, void *, int64_t, int64_t)> &get) {
    if (!g || offset < 0 || n < 0 || (n > 0 && !host)) return fail(PCL_ERR_ARG, "bad argument");
    int64_t at = 0, total = 0;
    for (pcl_ctx *c : g->ctx) total += c->count;
    if (offset + n > total) return fail(PCL_ERR_ARG, "window [%lld, %lld) outside the %lld particles of the group", (long long)offset,
                                        (long long)(offset + n), (long long)total);
    for (pcl_ctx *c : g->ctx) {
        const int64_t cnt = c->count;
        const int64_t lo = offset > at ? offset : at, hi = offset + n < at + cnt ? offset + n : at + cnt;
        if (hi > lo) PCL_TRY(get(c, host + (size_t)(lo - offset) * esz, lo - at, hi - lo));
        at += cnt;
    }
    return PCL_OK;
}

int pcl_group_download(pcl_group *g, int field, void *host, int64_t offset, int64_t n) {
    if (!g || g->ctx.empty()) return fail(PCL_ERR_ARG, "group is NULL");
    return group_window(g, offset, n, g->ctx[0]->esz, static_cast<char *>(host),
                        [field](pcl_ctx *c, void *h, int64_t off, int64_t cnt) { return pcl_store_download(c, field, h, off, cnt); });
}

int pcl_group_download_ids(pcl_group *g, int64_t *host, int64_t offset, int64_t n) {
    if (!g || g->ctx.empty()) return fail(PCL_ERR_ARG, "group is NULL");
    return group_window(g, offset, n, sizeof(int64_t), reinterpret_cast<char *>(host),
                        [](pcl_ctx *c, void *h, int64_t off, int64_t cnt) { return pcl_store_download_ids(c, static_cast<int64_t *>(h), off, cnt); });
}

} // extern "C"

// =================================================================================================
// pcl_comm_*: the counters' collective for hosts that run ONE PROCESS PER GPU and are not Python
// ("RCCL all-reduce over xGMI only for the global scatter / escape counters").  librccl is loaded at run
// time (dlopen: the library does not link it, a single-GPU host never needs it), the communicator is created from a
// unique id the caller ships to its ranks by whatever it has (a file, a socket, MPI, an environment variable), and the
// one collective there is sums an int64 vector on the context's stream.  Nothing falls back: a missing librccl, a failed
// bring-up or a failed all-reduce is an error (a throughput number must not hide a collective that did not run) --
// physicl_amd/dist.py behaves the same way on the Python side.
// =================================================================================================
namespace {

struct rccl_id { char internal[128]; }; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
static_assert(sizeof(rccl_id) == PCL_COMM_ID_BYTES, "pcl_comm_unique_id hands out an ncclUniqueId");

struct rccl_api {
    void *lib = nullptr;
    std::string path, why;
    int (*GetVersion)(int *) = nullptr;
    int (*GetUniqueId)(rccl_id *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

rccl_api &rccl() {
    static rccl_api api = [] {
        rccl_api a;
        std::vector<std::string> names;
        const char *forced = getenv("PCL_RCCL_LIB"); // (says which file to load: no other copy is looked for)
        if (forced) names.push_back(forced);
        for (const char *n : {"librccl.so.1", "librccl.so"}) { // a copy the process already holds (torch ships one) comes first
            if (forced) break;
            if (void *h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) {
                a.lib = h;
                a.path = n;
                break;
            }
        }
        if (!a.lib) {
            if (!forced) {
                names.insert(names.end(), {"librccl.so.1", "librccl.so"});
                if (const char *r = getenv("ROCM_PATH")) names.push_back(std::string(r) + "/lib/librccl.so.1");
                names.push_back("/opt/rocm/lib/librccl.so.1");
            }
            for (const std::string &n : names) {
                if (void *h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL)) {
                    a.lib = h;
                    a.path = n;
                    break;
                }
                a.why = dlerror() ? dlerror() : "";
            }
        }
        if (!a.lib) return a;
        auto sym = [&](const char *n) {
            void *p = dlsym(a.lib, n);
            if (!p) a.why = std::string("symbol ") + n + " is missing from " + a.path;
            return p;
        };
        a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(sym("ncclGetVersion"));
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        if (!a.GetVersion || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
            dlclose(a.lib);
            a.lib = nullptr;
        }
        return a;
    }();
    return api;
}

int need_rccl() {
    if (!rccl().lib) return fail(PCL_ERR_STATE, "librccl could not be loaded (%s): no collective, and no fallback", rccl().why.c_str());
    return PCL_OK;
}

#define PCL_RCCL(expr)                                                                                          \
    do {                                                                                                        \
        int r__ = (expr);                                                                                       \
        if (r__ != 0) return fail(PCL_ERR_HIP, "%s failed: %s (RCCL result %d)", #expr, rccl().GetErrorString(r__), r__); \
    } while (0)

} // namespace

struct pcl_comm {
    pcl_ctx *ctx = nullptr;
    void *nccl = nullptr;
    int rank = 0, world = 1;
    int64_t *d_buf = nullptr, *h_buf = nullptr; // the vector on the device / pinned on the host
    int cap = 0;
    int64_t reduces = 0;
};

extern "C" {

int pcl_comm_unique_id(void *id_out_host) {
    if (!id_out_host) return fail(PCL_ERR_ARG, "NULL argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        return fail(PCL_ERR_STATE, "the RCCL collective needs a GPU: no HIP device is visible to this process (librccl is not asked)");
    PCL_TRY(need_rccl());
    rccl_id id;
    memset(&id, 0, sizeof id);
    PCL_RCCL(rccl().GetUniqueId(&id));
    memcpy(id_out_host, &id, sizeof id);
    return PCL_OK;
}

int pcl_comm_create(pcl_ctx *ctx, const void *id_host, int rank, int world, pcl_comm **comm_out) {
    if (!ctx || !id_host || !comm_out) return fail(PCL_ERR_ARG, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(PCL_ERR_ARG, "rank %d outside a world of %d", rank, world);
    PCL_TRY(need_rccl());
    PCL_TRY(bind(ctx));
    rccl_id id;
    memcpy(&id, id_host, sizeof id);
    std::unique_ptr<pcl_comm> c(new (std::nothrow) pcl_comm);
    if (!c) return fail(PCL_ERR_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    PCL_RCCL(rccl().CommInitRank(&c->nccl, world, id, rank)); // (blocks until every rank of the id has arrived)
    c->cap = kCounterSlots * PCL_MULTI_MAX;                     // a K-pass launch's rows and more
    if (hipMalloc(reinterpret_cast<void **>(&c->d_buf), (size_t)c->cap * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_buf), (size_t)c->cap * sizeof(int64_t)) != hipSuccess) {
        if (c->d_buf) (void)hipFree(c->d_buf);
        (void)rccl().CommDestroy(c->nccl);
        return fail(PCL_ERR_NOMEM, "buffers of the communicator");
    }
    // the proof that the collective is up and sees every rank: sum of ones == world (what dist.py's start-up probe does)
    c->h_buf[0] = 1;
    auto probe = [&]() -> int {
        PCL_HIP(hipMemcpyAsync(c->d_buf, c->h_buf, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        PCL_RCCL(rccl().AllReduce(c->d_buf, c->d_buf, 1, 4 /* ncclInt64 */, 0 /* ncclSum */, c->nccl, ctx->stream));
        PCL_HIP(hipMemcpyAsync(c->h_buf, c->d_buf, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        PCL_HIP(hipStreamSynchronize(ctx->stream));
        if (c->h_buf[0] != world) return fail(PCL_ERR_HIP, "the start-up all-reduce saw %lld of %d ranks", (long long)c->h_buf[0], world);
        return PCL_OK;
    };
    const int rc = probe();
    if (rc != PCL_OK) { // (the message of the failing call stays in pcl_last_error)
        (void)rccl().CommDestroy(c->nccl);
        (void)hipFree(c->d_buf);
        (void)hipHostFree(c->h_buf);
        return rc;
    }
    *comm_out = c.release();
    return PCL_OK;
}

int pcl_comm_allreduce_sum_i64(pcl_comm *comm, int64_t *inout_host, int n) {
    if (!comm || (n > 0 && !inout_host)) return fail(PCL_ERR_ARG, "NULL argument");
    if (n < 0 || n > comm->cap) return fail(PCL_ERR_ARG, "n outside [0, %d]", comm->cap);
    if (n == 0) return PCL_OK;
    pcl_ctx *ctx = comm->ctx;
    PCL_TRY(bind(ctx));
    memcpy(comm->h_buf, inout_host, (size_t)n * sizeof(int64_t));
    PCL_HIP(hipMemcpyAsync(comm->d_buf, comm->h_buf, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PCL_RCCL(rccl().AllReduce(comm->d_buf, comm->d_buf, (size_t)n, 4 /* ncclInt64 */, 0 /* ncclSum */, comm->nccl, ctx->stream));
    PCL_HIP(hipMemcpyAsync(comm->h_buf, comm->d_buf, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PCL_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(inout_host, comm->h_buf, (size_t)n * sizeof(int64_t));
    ++comm->reduces;
    return PCL_OK;
}

int pcl_comm_info(pcl_comm *comm, int *rank_out, int *world_out, int *rccl_version_out, int64_t *reduces_out) {
    if (!comm) return fail(PCL_ERR_ARG, "NULL argument");
    if (rank_out) *rank_out = comm->rank;
    if (world_out) *world_out = comm->world;
    if (rccl_version_out) {
        *rccl_version_out = 0;
        (void)rccl().GetVersion(rccl_version_out);
    }
    if (reduces_out) *reduces_out = comm->reduces;
    return PCL_OK;
}

int pcl_comm_destroy(pcl_comm *comm) {
    if (!comm) return PCL_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->nccl) (void)rccl().CommDestroy(comm->nccl);
    if (comm->d_buf) (void)hipFree(comm->d_buf);
    if (comm->h_buf) (void)hipHostFree(comm->h_buf);
    delete comm;
    return PCL_OK;
}

} // extern "C"
