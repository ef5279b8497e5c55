// pcl_device.h -- device-side code shared by the ahead-of-time build (physicl_hip.hip) and the
// hipRTC specialisations for variable_n_fn expressions.  Device code only, no #includes, so the
// very same text compiles under hiprtc.  Written for gfx950 (wave64) only.
//
// Arithmetic contract (see DESIGN.md "Numerics"): everything the reference kernels compute with
// + - * / sqrt is evaluated left to right in IEEE fp64 with NO fma contraction (the library is
// built with -ffp-contract=off and hiprtc gets the same flag), so those results are bit-identical
// to the CPU oracle.  sin/cos/exp/pow come from ROCm's OCML.
#ifndef PCL_DEVICE_H
#define PCL_DEVICE_H

typedef long long pcl_i64;
typedef unsigned long long pcl_u64;
typedef unsigned int pcl_u32;

#define PCL_PI 3.141592653589793 /* == numpy.pi */

#define PCL_F_WAVELENGTH 1
#define PCL_F_VARIABLE_N 2
#define PCL_RNG_IN 0
#define PCL_RNG_PHX 1

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter (c0..c3), key (k0,k1).
// ------------------------------------------------------------------------------------------------
struct pcl_u32x4 {
    pcl_u32 x, y, z, w;
};

__device__ __forceinline__ pcl_u32x4 pcl_philox4x32_10(pcl_u32 c0, pcl_u32 c1, pcl_u32 c2, pcl_u32 c3,
                                                        pcl_u32 k0, pcl_u32 k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const pcl_u32 hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const pcl_u32 hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const pcl_u32 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    pcl_u32x4 r;
    r.x = c0;
    r.y = c1;
    r.z = c2;
    r.w = c3;
    return r;
}

// two 32-bit words -> double in [0,1) carrying 53 random bits (the MT19937 "res53" recipe numpy's
// random() uses): ((a>>5) * 2^26 + (b>>6)) / 2^53.  Exact in fp64.
__device__ __forceinline__ double pcl_u53(pcl_u32 a, pcl_u32 b) {
    const pcl_u64 m = ((pcl_u64)(a >> 5) << 26) | (pcl_u64)(b >> 6);
    return (double)m * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------------
// reference kernel maths
// ------------------------------------------------------------------------------------------------
// sqrt(pow(d0,2) + pow(d1,2) + pow(d2,2))            physicl/light.py:149, 241, 305
__device__ __forceinline__ double pcl_step_norm(double d0, double d1, double d2) {
    return __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2)));
}

// pow((h * c) / E[gid], -4)                           physicl/light.py:301
__device__ __forceinline__ double pcl_wavelength_term(double h, double c, double E) {
    return pow(__ddiv_rn(__dmul_rn(h, c), E), -4.0);
}

// res0 = c * sin(rtheta) * cos(rphi); res1 = c * sin(rtheta) * sin(rphi); res2 = c * cos(rtheta)
//                                                     physicl/light.py:309-311
__device__ __forceinline__ void pcl_new_velocity(double c, double rtheta, double rphi, double &o0, double &o1,
                                                 double &o2) {
    double st, ct, sp, cp;
    sincos(rtheta, &st, &ct);
    sincos(rphi, &sp, &cp);
    const double cs = __dmul_rn(c, st);
    o0 = __dmul_rn(cs, cp);
    o1 = __dmul_rn(cs, sp);
    o2 = __dmul_rn(c, ct);
}

// The number-density factor of pcoll.  Under hipRTC, PCL_N_EXPR is the user's OpenCL-C expression
// (variable_n_fn, physicl/light.py:299), which names the kernel arrays r0,r1,r2,d0,d1,d2,E and the
// work-item index gid; those names are bound here.  hipcc drops the loads of arrays the
// expression does not mention.
#ifdef PCL_N_EXPR
__device__ __forceinline__ double pcl_n_expr(pcl_i64 gid, const double *__restrict__ r0, const double *__restrict__ r1,
                                             const double *__restrict__ r2, const double *__restrict__ d0,
                                             const double *__restrict__ d1, const double *__restrict__ d2,
                                             const double *__restrict__ E) {
    (void)gid; (void)r0; (void)r1; (void)r2; (void)d0; (void)d1; (void)d2; (void)E;
    return (double)(PCL_N_EXPR);
}
#endif

// ------------------------------------------------------------------------------------------------
// Level 1: kernel light_scatter_step_sphere                        physicl/light.py:303-315
// ------------------------------------------------------------------------------------------------
struct pcl_sphere_args {
    const double *d0, *d1, *d2, *rtheta, *rphi, *rand;
    double A, n;
    const double *E, *r0, *r1, *r2;
    double *res0, *res1, *res2;
    pcl_i64 N;
    double c, h;
};

template <bool USE_E, bool VAR_N>
__device__ __forceinline__ void pcl_sphere_body(const pcl_sphere_args &a) {
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
    for (pcl_i64 gid = (pcl_i64)blockIdx.x * blockDim.x + threadIdx.x; gid < a.N; gid += stride) {
        const double norm = pcl_step_norm(a.d0[gid], a.d1[gid], a.d2[gid]);
        double pcoll;
        if constexpr (VAR_N) {
#ifdef PCL_N_EXPR
            pcoll = __dmul_rn(__dmul_rn(a.A, pcl_n_expr(gid, a.r0, a.r1, a.r2, a.d0, a.d1, a.d2, a.E)), norm);
#else
            pcoll = 0.0;
#endif
        } else {
            pcoll = __dmul_rn(__dmul_rn(a.A, a.n), norm);
        }
        if constexpr (USE_E) pcoll = __dmul_rn(pcoll, pcl_wavelength_term(a.h, a.c, a.E[gid]));
        if (pcoll >= a.rand[gid]) {
            double o0, o1, o2;
            pcl_new_velocity(a.c, a.rtheta[gid], a.rphi[gid], o0, o1, o2);
            a.res0[gid] = o0;
            a.res1[gid] = o1;
            a.res2[gid] = o2;
        } else {
            a.res0[gid] = __builtin_nan(""); // "Mark it as unaffected"; res1/res2 untouched
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Level 2: fused ScatterIsotropicStep on the resident store
//   kernel (light.py:303-315) + host write-back (light.py:325-331) + hit counter
// ------------------------------------------------------------------------------------------------
struct pcl_scatter_args {
    const double *d0, *d1, *d2; // Object.dr
    const double *E;            // PhotonObject.E           (USE_E)
    const double *r0, *r1, *r2; // Object.r                 (VAR_N)
    double *v0, *v1, *v2;       // Object.v   (read on hit, overwritten on hit)
    double *dv0, *dv1, *dv2;    // Object.dv  (always written)
    const double *rtheta, *rphi, *rand; // PCL_RNG_IN
    const pcl_i64 *ids;         // NULL: id = id_base + index
    const unsigned char *kind;  // NULL: every particle is a photon
    pcl_u64 *hits;              // one counter, += photons scattered
    pcl_i64 id_base;
    pcl_i64 N;
    double A, n, c, h;
    pcl_u64 seed;
    pcl_u32 step;
    int rng_mode;
};

#define PCL_SCATTER_ROWS 4 /* particles per thread per grid-stride trip (memory-level parallelism) */

template <bool USE_E, bool VAR_N>
__device__ __forceinline__ void pcl_scatter_body(const pcl_scatter_args &a) {
    const pcl_i64 tile = (pcl_i64)blockDim.x * PCL_SCATTER_ROWS;
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    pcl_u32 my_hits = 0;
    for (pcl_i64 base = (pcl_i64)blockIdx.x * tile; base < a.N; base += (pcl_i64)gridDim.x * tile) {
        double pcoll[PCL_SCATTER_ROWS];
        bool photon[PCL_SCATTER_ROWS];
#pragma unroll
        for (int j = 0; j < PCL_SCATTER_ROWS; ++j) {
            const pcl_i64 i = base + (pcl_i64)j * blockDim.x + threadIdx.x;
            photon[j] = false;
            pcoll[j] = 0.0;
            if (i < a.N) {
                photon[j] = a.kind ? (a.kind[i] != 0) : true;
                const double norm = pcl_step_norm(a.d0[i], a.d1[i], a.d2[i]);
                double p;
                if constexpr (VAR_N) {
#ifdef PCL_N_EXPR
                    p = __dmul_rn(__dmul_rn(a.A, pcl_n_expr(i, a.r0, a.r1, a.r2, a.d0, a.d1, a.d2, a.E)), norm);
#else
                    p = 0.0;
#endif
                } else {
                    p = __dmul_rn(__dmul_rn(a.A, a.n), norm);
                }
                if constexpr (USE_E) p = __dmul_rn(p, pcl_wavelength_term(a.h, a.c, a.E[i]));
                pcoll[j] = p;
            }
        }
#pragma unroll
        for (int j = 0; j < PCL_SCATTER_ROWS; ++j) {
            const pcl_i64 i = base + (pcl_i64)j * blockDim.x + threadIdx.x;
            if (i >= a.N || !photon[j]) continue;
            double rand, rtheta = 0.0, rphi = 0.0;
            pcl_u32 c0 = 0, c1 = 0;
            if (a.rng_mode == PCL_RNG_PHX) {
                const pcl_u64 id = (pcl_u64)(a.ids ? a.ids[i] : a.id_base + i);
                c0 = (pcl_u32)id;
                c1 = (pcl_u32)(id >> 32);
                const pcl_u32x4 w = pcl_philox4x32_10(c0, c1, a.step, 0u, k0, k1);
                rand = pcl_u53(w.x, w.y);
                rtheta = __dmul_rn(__dmul_rn(pcl_u53(w.z, w.w), 2.0), PCL_PI);
            } else {
                rand = a.rand[i];
            }
            // NaN pcoll compares false, +inf compares true: same as the reference's ``pcoll >= rand``
            if (pcoll[j] >= rand) {
                if (a.rng_mode == PCL_RNG_PHX) {
                    const pcl_u32x4 w = pcl_philox4x32_10(c0, c1, a.step, 1u, k0, k1);
                    rphi = __dmul_rn(pcl_u53(w.x, w.y), PCL_PI);
                } else {
                    rtheta = a.rtheta[i];
                    rphi = a.rphi[i];
                }
                double n0, n1, n2;
                pcl_new_velocity(a.c, rtheta, rphi, n0, n1, n2);
                const double o0 = a.v0[i], o1 = a.v1[i], o2 = a.v2[i];
                a.v0[i] = n0;
                a.v1[i] = n1;
                a.v2[i] = n2;
                a.dv0[i] = __dsub_rn(n0, o0);
                a.dv1[i] = __dsub_rn(n1, o1);
                a.dv2[i] = __dsub_rn(n2, o2);
                ++my_hits;
            } else {
                a.dv0[i] = 0.0;
                a.dv1[i] = 0.0;
                a.dv2[i] = 0.0;
            }
        }
    }
    // wave reduction by DPP-free shuffle, then one atomic per wave that saw a hit
    for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_down(my_hits, off, 64);
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(a.hits, (pcl_u64)my_hits);
}

#ifdef PCL_RTC
// hipRTC translation unit: one expression, both wavelength variants of both kernels.
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_sphere_e0(pcl_sphere_args a) { pcl_sphere_body<false, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_sphere_e1(pcl_sphere_args a) { pcl_sphere_body<true, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_scatter_e0(pcl_scatter_args a) { pcl_scatter_body<false, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_scatter_e1(pcl_scatter_args a) { pcl_scatter_body<true, true>(a); }
#endif
#endif // PCL_DEVICE_H
