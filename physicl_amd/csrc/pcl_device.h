// pcl_device.h -- device-side code shared by the ahead-of-time build (physicl_hip.hip) and the
// hipRTC specialisations for variable_n_fn expressions.  Device code only, no #includes, so the
// very same text compiles under hiprtc.  Written for gfx950 (wave64) only.
//
// Arithmetic contract (see DESIGN.md "Numerics"): everything the reference kernels compute with
// + - * / sqrt is evaluated left to right in IEEE arithmetic with NO fma contraction (the library is
// built with -ffp-contract=off and hiprtc gets the same flag), so fp64 results are bit-identical
// to the CPU oracle.  exp/pow come from ROCm's OCML; fp64 sin/cos of the scatter angles from pcl_sincos.h
// (< 1 ulp, OCML outside [0, 6.5]); fp32 sin/cos from OCML.  The reference is fp64-only
// (physicl/__init__.py:613); the fp32 instantiations (T = float) exist for the precision sweep of
// BASELINE.json configs[4] and are bit-exact against the oracle's float32 restatement.
#ifndef PCL_DEVICE_H
#define PCL_DEVICE_H

typedef long long pcl_i64;
typedef unsigned long long pcl_u64;
typedef unsigned int pcl_u32;

#define PCL_SC_FN __device__ __forceinline__
#include "pcl_sincos.h" /* build.py splices the file in here for the hipRTC copy of this text */

// The wave's vote as the hardware gives it: the compare's own lane mask (ANDed with exec).  HIP's __ballot(int) first turns
// the predicate into an integer and compares that with zero again -- a v_cndmask and a second v_cmp, 8 SIMD-cycles per
// vote (tools/valu_issue_probe.hip) in kernels that vote four times per photon and step.
__device__ __forceinline__ pcl_u64 pcl_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

#define PCL_PI 3.141592653589793 /* == numpy.pi */
#define PCL_RNG_IN 0
#define PCL_RNG_PHX 1
#define PCL_MAXPL 12
#define PCL_PEND_MAX 8 /* Newton moves a store behind an alive mask may owe its r rows (kPendMax of physicl_hip.hip) */

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter (c0..c3), key (k0,k1).
// ------------------------------------------------------------------------------------------------
struct pcl_u32x4 {
    pcl_u32 x, y, z, w;
};

__device__ __forceinline__ pcl_u32x4 pcl_philox4x32_10(pcl_u32 c0, pcl_u32 c1, pcl_u32 c2, pcl_u32 c3,
                                                        pcl_u32 k0, pcl_u32 k1) {
#ifdef PCL_ABLATE_PHILOX /* timing experiment only */
    const int kRounds = 1;
#else
    const int kRounds = 10;
#endif
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: integer
        // multiplies are quarter rate on CDNA and Philox is most of the integer work of a step
        const pcl_u64 p0 = (pcl_u64)0xD2511F53u * (pcl_u64)c0, p1 = (pcl_u64)0xCD9E8D57u * (pcl_u64)c2;
        const pcl_u32 hi0 = (pcl_u32)(p0 >> 32), lo0 = (pcl_u32)p0;
        const pcl_u32 hi1 = (pcl_u32)(p1 >> 32), lo1 = (pcl_u32)p1;
        // three-input XOR in one instruction (gfx950 v_bitop3_b32, truth table 0x96 = a ^ b ^ c)
        const pcl_u32 n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
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
// real-number traits: the rounding-explicit operations of each precision
// ------------------------------------------------------------------------------------------------
template <typename T> struct pcl_rt;

template <> struct pcl_rt<double> {
    static constexpr int VEC = 2; // elements per 16-byte access
    static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
    static __device__ __forceinline__ double div(double a, double b) { return __ddiv_rn(a, b); }
    static __device__ __forceinline__ double sqrt_(double a) { return __dsqrt_rn(a); }
    static __device__ __forceinline__ double pow_m4(double x) {
#ifdef PCL_ABLATE_POW /* timing experiment only */
        const double x2 = x * x;
        return 1.0 / (x2 * x2);
#else
        return pow(x, -4.0);
#endif
    }
    static __device__ __forceinline__ void sincos_(double x, double &s, double &c) {
#ifdef PCL_ABLATE_TRIG /* timing experiment only */
        s = x * 0.1;
        c = 1.0 - s;
#else
        // the step's own angles lie in [0, 2*pi] (light.py:285): short reduction, < 1 ulp (pcl_sincos.h); anything
        // else a caller of the reference-ABI kernels may pass -- negative, huge, NaN -- goes to the library
        if (x >= 0.0 && x <= PCL_SINCOS_XMAX)
            pcl_sincos_2pi(x, &s, &c);
        else
            sincos(x, &s, &c);
#endif
    }
    // an angle the kernel itself built from a uniform draw (u*2*pi or u*pi): always inside [0, 2*pi]
    static __device__ __forceinline__ void sincos_angle_(double x, double &s, double &c) {
#ifdef PCL_ABLATE_TRIG /* timing experiment only */
        s = x * 0.1;
        c = 1.0 - s;
#else
        pcl_sincos_2pi(x, &s, &c);
#endif
    }
    // uniform in [0,1) from two Philox words: all 53 bits
    static __device__ __forceinline__ double uniform(pcl_u32 a, pcl_u32 b) { return pcl_u53(a, b); }
    static __device__ __forceinline__ double pi() { return PCL_PI; }
    static __device__ __forceinline__ double nan_() { return __builtin_nan(""); }
};

template <> struct pcl_rt<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
    static __device__ __forceinline__ float div(float a, float b) { return __fdiv_rn(a, b); }
    static __device__ __forceinline__ float sqrt_(float a) { return __fsqrt_rn(a); }
    static __device__ __forceinline__ float pow_m4(float x) { return powf(x, -4.0f); }
    static __device__ __forceinline__ void sincos_(float x, float &s, float &c) { sincosf(x, &s, &c); }
    static __device__ __forceinline__ void sincos_angle_(float x, float &s, float &c) { sincosf(x, &s, &c); }
    // the top 24 bits of the SAME words the fp64 path uses: u32 <= u64 < u32 + 2^-24, so both
    // precisions follow the same random stream (hit decisions differ only within that sliver)
    static __device__ __forceinline__ float uniform(pcl_u32 a, pcl_u32 b) {
        (void)b;
        return (float)(a >> 8) * (1.0f / 16777216.0f);
    }
    static __device__ __forceinline__ float pi() { return 3.14159274101257324f; } // float32(numpy.pi)
    static __device__ __forceinline__ float nan_() { return __builtin_nanf(""); }
};

// VEC consecutive elements per lane, as one 16-byte (or narrower) access
template <typename T, int VEC> struct pcl_vec;
template <typename T> struct pcl_vec<T, 1> {
    static __device__ __forceinline__ void ld(const T *b, pcl_i64 q, T (&o)[1]) { o[0] = b[q]; }
    static __device__ __forceinline__ void st(T *b, pcl_i64 q, const T (&o)[1]) { b[q] = o[0]; }
};
template <> struct pcl_vec<double, 2> {
    static __device__ __forceinline__ void ld(const double *b, pcl_i64 q, double (&o)[2]) {
        const double2 t = reinterpret_cast<const double2 *>(b)[q];
        o[0] = t.x;
        o[1] = t.y;
    }
    static __device__ __forceinline__ void st(double *b, pcl_i64 q, const double (&o)[2]) {
        reinterpret_cast<double2 *>(b)[q] = make_double2(o[0], o[1]);
    }
};
template <> struct pcl_vec<float, 4> {
    static __device__ __forceinline__ void ld(const float *b, pcl_i64 q, float (&o)[4]) {
        const float4 t = reinterpret_cast<const float4 *>(b)[q];
        o[0] = t.x;
        o[1] = t.y;
        o[2] = t.z;
        o[3] = t.w;
    }
    static __device__ __forceinline__ void st(float *b, pcl_i64 q, const float (&o)[4]) {
        reinterpret_cast<float4 *>(b)[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
};

// ------------------------------------------------------------------------------------------------
// Tiled store layout ("AoSoA"): a tile holds PCL_T consecutive particles; inside a tile every field is one
// contiguous row of PCL_T elements and the rows lie back to back: [tile][field][PCL_T].  A Level-2 kernel
// gets, per field, the address of that field's row in tile 0 plus ``ts`` = elements from one tile to the
// next.  Lanes still read consecutive elements (full coalescing) but the 13+ streams of a pass now fall into
// one contiguous ~100-270 KiB region per tile, so how the driver scatters large allocations over HBM
// channels no longer decides the speed (tools/attic/bw_probe2.hip).
// ------------------------------------------------------------------------------------------------
#define PCL_TLOG 11
#define PCL_T (1 << PCL_TLOG)
__device__ __forceinline__ pcl_i64 pcl_tix(pcl_i64 i, pcl_i64 ts) { return (i >> PCL_TLOG) * ts + (i & (PCL_T - 1)); }
template <int VEC>
__device__ __forceinline__ pcl_i64 pcl_tq(pcl_i64 q, pcl_i64 ts) { // q counts VEC-wide groups of elements
    constexpr int LG = PCL_TLOG - (VEC == 4 ? 2 : (VEC == 2 ? 1 : 0));
    return (q >> LG) * (ts / VEC) + (q & ((1 << LG) - 1));
}

// ------------------------------------------------------------------------------------------------
// reference kernel maths
// ------------------------------------------------------------------------------------------------
// sqrt(pow(d0,2) + pow(d1,2) + pow(d2,2))            physicl/light.py:149, 241, 305
template <typename T>
__device__ __forceinline__ T pcl_step_norm(T d0, T d1, T d2) {
    typedef pcl_rt<T> R;
    return R::sqrt_(R::add(R::add(R::mul(d0, d0), R::mul(d1, d1)), R::mul(d2, d2)));
}

// pow((h * c) / E[gid], -4)                           physicl/light.py:301
template <typename T>
__device__ __forceinline__ T pcl_wavelength_term(T h, T c, T E) {
    typedef pcl_rt<T> R;
    return R::pow_m4(R::div(R::mul(h, c), E));
}

// res0 = c * sin(rtheta) * cos(rphi); res1 = c * sin(rtheta) * sin(rphi); res2 = c * cos(rtheta)
//                                                     physicl/light.py:309-311
// OWN_ANGLES: rtheta/rphi were drawn by the kernel itself (device RNG), so they are inside [0, 2*pi] and the
// range test + library fallback of sincos_ is not compiled in at all; same values either way.
template <typename T, bool OWN_ANGLES = false>
__device__ __forceinline__ void pcl_new_velocity(T c, T rtheta, T rphi, T &o0, T &o1, T &o2) {
    typedef pcl_rt<T> R;
    T st, ct, sp, cp;
    if constexpr (OWN_ANGLES) {
        R::sincos_angle_(rtheta, st, ct);
        R::sincos_angle_(rphi, sp, cp);
    } else {
        R::sincos_(rtheta, st, ct);
        R::sincos_(rphi, sp, cp);
    }
    const T cs = R::mul(c, st);
    o0 = R::mul(cs, cp);
    o1 = R::mul(cs, sp);
    o2 = R::mul(c, ct);
}

// rtheta = u * 2 * pi, rphi = u * pi                  physicl/light.py:285 (left to right)
template <typename T>
__device__ __forceinline__ T pcl_rtheta(pcl_u32 a, pcl_u32 b) {
    typedef pcl_rt<T> R;
    return R::mul(R::mul(R::uniform(a, b), (T)2), R::pi());
}
template <typename T>
__device__ __forceinline__ T pcl_rphi(pcl_u32 a, pcl_u32 b) {
    typedef pcl_rt<T> R;
    return R::mul(R::uniform(a, b), R::pi());
}

// Device random stream of a light step (DESIGN.md "Device RNG"), keyed by (seed, launch number ``step``, photon id):
//   decision block  Philox(id_lo, id_hi, step >> 1, 0):  rand of an even step = u(w0, w1), of an odd step = u(w2, w3)
//                   -- one block serves the hit decisions of two consecutive steps (a K-step pass computes it once);
//   direction block Philox(id_lo, id_hi, step, 1), hit only:  rtheta = u(w0, w1) * 2 * pi, rphi = u(w2, w3) * pi.
template <typename T>
__device__ __forceinline__ T pcl_draw_rand(pcl_u64 id, pcl_u32 step, pcl_u32 k0, pcl_u32 k1) {
    typedef pcl_rt<T> R;
    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), step >> 1, 0u, k0, k1);
    return (step & 1u) ? R::uniform(w.z, w.w) : R::uniform(w.x, w.y);
}
template <typename T>
__device__ __forceinline__ void pcl_draw_angles(pcl_u64 id, pcl_u32 step, pcl_u32 k0, pcl_u32 k1, T &rtheta, T &rphi) {
    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), step, 1u, k0, k1);
    rtheta = pcl_rtheta<T>(w.x, w.y);
    rphi = pcl_rphi<T>(w.z, w.w);
}

// The number-density factor of pcoll (variable_n_fn, physicl/light.py:299).
// Under hipRTC, PCL_N_EXPR is the user's OpenCL-C expression, which names the kernel arrays r0,r1,r2,d0,d1,d2,E and the
// work-item index gid; those names are bound here to this particle's values (one-element arrays, gid = 0).  hipcc drops
// the loads of arrays the expression does not mention.  PCL_N_EXPR_F is the same text with an f suffix on every floating
// literal (fp32 stores).
// In the ahead-of-time library there is no expression text: its VAR_N kernels evaluate one of three parametrised
// SHAPES -- the forms the reference's examples use -- with the literals of the user's text as kernel arguments
// (pcl_nprof), written token for token like the examples so that the compiler sees the same operations as it does
// under hipRTC.  They serve when hipRTC is not available at run time (DESIGN.md "variable_n_fn").
//   1  p0 * exp(rA[gid] - p1)                    examples/variable_n_scattering.ipynb:30
//   2  p0 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - p1)/(p2))
//                                                examples/presentation_example.ipynb:31, presentation_example_2.ipynb:40
//   3  p0 * exp(rA[gid] / p1)                    examples/presentation_example_2.ipynb:41
#define PCL_NPROF_EXP_OFFSET 1
#define PCL_NPROF_EXP_RADIAL 2
#define PCL_NPROF_EXP_SCALE 3
template <typename T>
struct pcl_nprof {
    int shape; // 0: none (hipRTC, or constant n)
    int axis;  // A of rA[gid] (shapes 1 and 3)
    T p0, p1, p2;
};

// exp() inside a variable_n_fn expression.  EXPERIMENT, off unless PCL_EXP_WAVE is defined (PCL_RTC_EXTRA=PCL_EXP_WAVE):
// where the argument lies beyond the range in which exp saturates the result is exactly +inf or +0 in any IEEE libm
// (fp64: x > 709.79 / x < -745.14; fp32: 88.73 / -103.98), so when EVERY active lane of the wave is that far out -- the
// state of examples/variable_n_scattering.ipynb after its first step -- the ~40-instruction polynomial can be skipped for
// the whole wave at the price of a compare, a ballot and a branch.  Same value either way.  Measured on the K-step kernel
// (DESIGN.md section 4): +9 % on the example's regime (with the kernel pinned to 4 waves / SIMD), -6 % on a regime where
// exp never saturates -- the branch splits the two photons' dependency chains and costs registers -- so it is not on.
__device__ __forceinline__ double pcl_exp_wave(double x) {
#ifdef PCL_EXP_WAVE
    if (pcl_ballot(!(__builtin_fabs(x) > 750.0)) == 0ull) return x > 0.0 ? __builtin_inf() : 0.0;
#endif
    return exp(x);
}
__device__ __forceinline__ float pcl_exp_wave(float x) {
#ifdef PCL_EXP_WAVE
    if (pcl_ballot(!(__builtin_fabsf(x) > 105.0f)) == 0ull) return x > 0.0f ? __builtin_inff() : 0.0f;
#endif
    return expf(x);
}
__device__ __forceinline__ double pcl_exp_wave(int x) { return pcl_exp_wave((double)x); } // exp(2) in user text: C promotes

#define exp(x) pcl_exp_wave(x) /* for the body of pcl_n_expr_val only; undefined again right after it */
// VAR_N (template parameter of every kernel body): 0 = constant n, 1 = variable n -- the spliced text under hipRTC, or
// whatever pcl_nprof says in the ahead-of-time kernels --, 2 .. 7 = ahead-of-time only: the two one-component shapes
// (the reference's examples) on a literal axis, so that their loops carry one expression instead of three and a select:
// 2 + 3 * (0: K * exp(rA - X), 1: K * exp(rA / X)) + axis; 8 = the radial exponential
#define PCL_VARN_SHAPED(scale, axis) (2 + 3 * (scale) + (axis))
template <typename T, int VAR_N = 1>
__device__ __forceinline__ T pcl_n_expr_val(const pcl_nprof<T> &np, T r0v, T r1v, T r2v, T d0v, T d1v, T d2v, T Ev) {
    const T r0[1] = {r0v}, r1[1] = {r1v}, r2[1] = {r2v}, d0[1] = {d0v}, d1[1] = {d1v}, d2[1] = {d2v}, E[1] = {Ev};
    const int gid = 0;
    (void)r0; (void)r1; (void)r2; (void)d0; (void)d1; (void)d2; (void)E; (void)gid; (void)np;
#ifdef PCL_N_EXPR
    if constexpr (sizeof(T) == 8) {
        return (T)(PCL_N_EXPR);
    } else {
        return (T)(PCL_N_EXPR_F);
    }
#else
    if constexpr (VAR_N == 8) // the radial exponential, literal shape
        return (T)(np.p0 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - np.p1)/(np.p2)));
    if constexpr (VAR_N >= 2) {
        constexpr int ax = (VAR_N - 2) % 3;
        const T rL[1] = {ax == 0 ? r0v : (ax == 1 ? r1v : r2v)}; // (constant-folded)
        if constexpr ((VAR_N - 2) / 3 == 0) return (T)(np.p0 * exp(rL[gid] - np.p1));
        return (T)(np.p0 * exp(rL[gid] / np.p1));
    }
    const T rA[1] = {np.axis == 0 ? r0v : (np.axis == 1 ? r1v : r2v)};
    if (np.shape == PCL_NPROF_EXP_OFFSET) return (T)(np.p0 * exp(rA[gid] - np.p1));
    if (np.shape == PCL_NPROF_EXP_RADIAL)
        return (T)(np.p0 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - np.p1)/(np.p2)));
    return (T)(np.p0 * exp(rA[gid] / np.p1));
#endif
}
#undef exp

// The K-step pass's "saturation probe" (hipRTC builds; pcl_multi_body_lds<..., SATP = true>).  Where its argument lies beyond
// the range in which exp saturates, the result is exactly +inf or +0 in any IEEE libm (fp64: x > 709.79 / x < -745.14;
// fp32: 88.73 / -103.98).  The expression is first evaluated with exp() replaced by that shortcut for BOTH photons of a
// lane, noting whether any argument was NOT that far out; when no lane of the wave noted one -- the state of
// examples/variable_n_scattering.ipynb after its first step: every photon is 1.5e6 m from the origin -- those values ARE the
// expression's values and the ~25-instruction polynomial is never run; otherwise the wave evaluates the expression the
// ordinary way.  One wave-uniform branch per step around both photons' polynomials (round 2's per-call branch,
// PCL_EXP_WAVE above, split the two photons' dependency chains and cost 40 VGPRs).  Same values either way.
#ifdef PCL_N_EXPR
__device__ __forceinline__ double pcl_exp_sat(double x, bool &unsat) {
    unsat = unsat || !(__builtin_fabs(x) > 750.0);
    return x > 0.0 ? __builtin_inf() : 0.0;
}
__device__ __forceinline__ float pcl_exp_sat(float x, bool &unsat) {
    unsat = unsat || !(__builtin_fabsf(x) > 105.0f);
    return x > 0.0f ? __builtin_inff() : 0.0f;
}
__device__ __forceinline__ double pcl_exp_sat(int x, bool &unsat) { return pcl_exp_sat((double)x, unsat); }
#define exp(x) pcl_exp_sat(x, pcl_unsat)
template <typename T>
__device__ __forceinline__ T pcl_n_expr_sat(T r0v, T r1v, T r2v, T d0v, T d1v, T d2v, T Ev, bool &pcl_unsat) {
    const T r0[1] = {r0v}, r1[1] = {r1v}, r2[1] = {r2v}, d0[1] = {d0v}, d1[1] = {d1v}, d2[1] = {d2v}, E[1] = {Ev};
    const int gid = 0;
    (void)r0; (void)r1; (void)r2; (void)d0; (void)d1; (void)d2; (void)E; (void)gid;
    if constexpr (sizeof(T) == 8) {
        return (T)(PCL_N_EXPR);
    } else {
        return (T)(PCL_N_EXPR_F);
    }
}
#undef exp
#endif

// pcoll exactly as the generated kernel text multiplies it            physicl/light.py:299-306
//   A * n * norm  |  A * (<expr>) * norm  [ * pow((h*c)/E, -4) ]   -- left to right
// (the norm is passed in: a K-step pass keeps it in a register between the photon's hits)
template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ T pcl_pcoll_norm(const pcl_nprof<T> &np, T A, T n, T h, T c, T norm, T d0, T d1, T d2, T r0, T r1, T r2,
                                            T E) {
    typedef pcl_rt<T> R;
    T p;
    if constexpr (VAR_N) {
        p = R::mul(R::mul(A, pcl_n_expr_val<T, VAR_N>(np, r0, r1, r2, d0, d1, d2, E)), norm);
    } else {
        p = R::mul(R::mul(A, n), norm);
    }
    if constexpr (USE_E) p = R::mul(p, pcl_wavelength_term<T>(h, c, E));
    return p;
}
template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ T pcl_pcoll(const pcl_nprof<T> &np, T A, T n, T h, T c, T d0, T d1, T d2, T r0, T r1, T r2, T E) {
    return pcl_pcoll_norm<T, USE_E, VAR_N>(np, A, n, h, c, pcl_step_norm<T>(d0, d1, d2), d0, d1, d2, r0, r1, r2, E);
}

// ------------------------------------------------------------------------------------------------
// Level 1: kernel light_scatter_step_sphere (fp64, the reference's ABI)   physicl/light.py:303-315
// ------------------------------------------------------------------------------------------------
struct pcl_sphere_args {
    const double *d0, *d1, *d2, *rtheta, *rphi, *rand;
    double A, n;
    const double *E, *r0, *r1, *r2; // NULL when the kernel does not take them
    double *res0, *res1, *res2;
    pcl_i64 N;
    double c, h;
    pcl_nprof<double> np; // ahead-of-time VAR_N kernels only
};

template <bool USE_E, int VAR_N>
__device__ __forceinline__ void pcl_sphere_body(const pcl_sphere_args &a) {
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
    for (pcl_i64 gid = (pcl_i64)blockIdx.x * blockDim.x + threadIdx.x; gid < a.N; gid += stride) {
        const double pcoll = pcl_pcoll<double, USE_E, VAR_N>(a.np, a.A, a.n, a.h, a.c, a.d0[gid], a.d1[gid], a.d2[gid], a.r0 ? a.r0[gid] : 0.0, a.r1 ? a.r1[gid] : 0.0,
            a.r2 ? a.r2[gid] : 0.0, a.E ? a.E[gid] : 0.0);
        if (pcoll >= a.rand[gid]) {
            double o0, o1, o2;
            pcl_new_velocity<double>(a.c, a.rtheta[gid], a.rphi[gid], o0, o1, o2);
            a.res0[gid] = o0;
            a.res1[gid] = o1;
            a.res2[gid] = o2;
        } else {
            a.res0[gid] = __builtin_nan(""); // "Mark it as unaffected"; res1/res2 untouched
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Level 2: ScatterIsotropicStep as its own step on the resident store
//   kernel (light.py:303-315) + host write-back (light.py:325-331) + hit counter
// ------------------------------------------------------------------------------------------------
template <typename T>
struct pcl_scatter_args {
    const T *d0, *d1, *d2;     // Object.dr
    const T *E;                // PhotonObject.E
    const T *r0, *r1, *r2;     // Object.r
    T *v0, *v1, *v2;           // Object.v   (read on hit, overwritten on hit)
    T *dv0, *dv1, *dv2;        // Object.dv  (always written)
    const T *rtheta, *rphi, *rand; // PCL_RNG_IN
    const pcl_i64 *ids;        // NULL: id = id_base + index
    const unsigned char *kind; // NULL: every particle is a photon
    pcl_u64 *hits;             // one counter, += photons scattered
    pcl_i64 id_base;
    pcl_i64 N;
    pcl_i64 ts;                // tile stride of the store (elements)
    T A, n, c, h;
    pcl_u64 seed;
    pcl_u32 step;
    int rng_mode;
    int py_dv;                 // 1: a hit leaves dv = v_old (the reference's CPU path, light.py:346-348) instead of v' - v_old
    pcl_nprof<T> np;           // ahead-of-time VAR_N kernels only
};

#define PCL_SCATTER_ROWS 4 /* particles per thread per grid-stride trip (memory-level parallelism) */

template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ void pcl_scatter_body(const pcl_scatter_args<T> &a) {
    typedef pcl_rt<T> R;
    const pcl_i64 tile = (pcl_i64)blockDim.x * PCL_SCATTER_ROWS;
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    pcl_u32 my_hits = 0;
    for (pcl_i64 base = (pcl_i64)blockIdx.x * tile; base < a.N; base += (pcl_i64)gridDim.x * tile) {
        T pcoll[PCL_SCATTER_ROWS];
        bool photon[PCL_SCATTER_ROWS];
#pragma unroll
        for (int j = 0; j < PCL_SCATTER_ROWS; ++j) {
            const pcl_i64 i = base + (pcl_i64)j * blockDim.x + threadIdx.x;
            photon[j] = false;
            pcoll[j] = (T)0;
            if (i < a.N) {
                photon[j] = a.kind ? (a.kind[i] != 0) : true;
                const pcl_i64 ti = pcl_tix(i, a.ts);
                pcoll[j] = pcl_pcoll<T, USE_E, VAR_N>(a.np, a.A, a.n, a.h, a.c, a.d0[ti], a.d1[ti], a.d2[ti], a.r0[ti],
                                                      a.r1[ti], a.r2[ti], a.E[ti]);
            }
        }
#pragma unroll
        for (int j = 0; j < PCL_SCATTER_ROWS; ++j) {
            const pcl_i64 i = base + (pcl_i64)j * blockDim.x + threadIdx.x;
            if (i >= a.N || !photon[j]) continue;
            const pcl_i64 ti = pcl_tix(i, a.ts);
            T rand, rtheta = (T)0, rphi = (T)0;
            pcl_u64 id = 0;
            if (a.rng_mode == PCL_RNG_PHX) {
                id = (pcl_u64)(a.ids ? a.ids[i] : a.id_base + i);
                rand = pcl_draw_rand<T>(id, a.step, k0, k1);
            } else {
                rand = a.rand[i];
            }
            // NaN pcoll compares false, +inf compares true: same as the reference's ``pcoll >= rand``
            if (pcoll[j] >= rand) {
                if (a.rng_mode == PCL_RNG_PHX) {
                    pcl_draw_angles<T>(id, a.step, k0, k1, rtheta, rphi);
                } else {
                    rtheta = a.rtheta[i];
                    rphi = a.rphi[i];
                }
                T n0, n1, n2;
                pcl_new_velocity<T>(a.c, rtheta, rphi, n0, n1, n2);
                const T o0 = a.v0[ti], o1 = a.v1[ti], o2 = a.v2[ti];
                a.v0[ti] = n0;
                a.v1[ti] = n1;
                a.v2[ti] = n2;
                a.dv0[ti] = a.py_dv ? o0 : R::sub(n0, o0);
                a.dv1[ti] = a.py_dv ? o1 : R::sub(n1, o1);
                a.dv2[ti] = a.py_dv ? o2 : R::sub(n2, o2);
                ++my_hits;
            } else {
                a.dv0[ti] = (T)0;
                a.dv1[ti] = (T)0;
                a.dv2[ti] = (T)0;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_down(my_hits, off, 64);
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(a.hits, (pcl_u64)my_hits);
}

// ------------------------------------------------------------------------------------------------
// Level 2: ONE pass for the whole loop body  Newton -> ScatterIsotropic -> measure counters
//   physicl/newton.py:15-16, physicl/light.py:303-315 + 325-331, physicl/light.py:385-399 + 424-426
// Same per-particle arithmetic, in the same order, as the three separate kernels -- so results are
// bit-identical to running them one after the other -- but r, v, E are read once and dr never
// travels back from HBM: 128 + 24h B per fp64 particle-step instead of 96 + (64 + 48h) + 24.
// Each lane owns VEC consecutive particles (16-byte loads/stores); counters are wave-ballot
// popcounts kept in scalar registers, LDS-staged per workgroup, one atomic per workgroup per counter.
// This is the general version (plain Objects mixed in, explicit ids, host randoms, plane counters,
// eager or lazy dr/dv); pcl_fast_body below is the specialisation the benchmark runs.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct pcl_fused_args {
    T *r0, *r1, *r2;             // Object.r   (read, written)
    const T *vi0, *vi1, *vi2;    // Object.v as the step finds it
    T *vo0, *vo1, *vo2;          // Object.v as the step leaves it: eager = same arrays, written on a hit;
                                 // lazy = the other half of the v double buffer, always written
    T *dr0, *dr1, *dr2;          // Object.dr  (written; not in lazy mode)
    T *dv0, *dv1, *dv2;          // Object.dv  (written for photons; not in lazy mode)
    const T *E;
    const T *rtheta, *rphi, *rand; // PCL_RNG_IN
    const pcl_i64 *ids;
    const unsigned char *kind;
    pcl_u64 *cnt;                // [0] hits, [1..3] sign counts, [4..] plane crossings
    pcl_i64 id_base;
    pcl_i64 N;
    pcl_i64 ts;                  // tile stride of the store (elements)
    T dt, A, n, c, h;
    pcl_u64 seed;
    pcl_u32 step;
    int rng_mode;
    int lazy;                    // 1: dr/dv stay implicit (dr = v_in*dt, dv = v_out - v_in), see pcl_step_fused
    int do_scatter;              // 0: Newton (+ counters) only
    int n_planes;                // -1: no counters at all
    T plane_L[PCL_MAXPL];
    int plane_ax[PCL_MAXPL];
    pcl_nprof<T> np;             // ahead-of-time VAR_N kernels only
};

template <typename T>
__device__ __forceinline__ T pcl_pick(int ax, T a0, T a1, T a2) { // by value: never pins arrays in scratch
    return ax == 0 ? a0 : (ax == 1 ? a1 : a2);
}

template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ void pcl_fused_body(const pcl_fused_args<T> &a) {
    typedef pcl_rt<T> R;
    constexpr int VEC = R::VEC;
    typedef pcl_vec<T, VEC> VV;
    __shared__ pcl_u32 s_cnt[4 + PCL_MAXPL];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 4 + PCL_MAXPL) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    pcl_u32 w_hits = 0, w_sx = 0, w_sy = 0, w_sz = 0; // wave-uniform tallies (live in SGPRs)
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const bool counters = a.n_planes >= 0;
    const pcl_i64 nq = (a.N + VEC - 1) / VEC;
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
    // every lane of a workgroup makes the same number of trips, so ballots always see whole waves
    for (pcl_i64 base = (pcl_i64)blockIdx.x * blockDim.x; base < nq; base += stride) {
        const pcl_i64 q = base + threadIdx.x;
        const bool live_q = q < nq;
        const pcl_i64 qq = pcl_tq<VEC>(live_q ? q : 0, a.ts); // idle lanes re-read group 0 and store nothing
        const pcl_i64 qs = pcl_tq<VEC>(q, a.ts);               // where this lane's group lives
        T Rr[3][VEC], V[3][VEC], D[3][VEC], DV[3][VEC], Ev[VEC];
        VV::ld(a.r0, qq, Rr[0]);
        VV::ld(a.r1, qq, Rr[1]);
        VV::ld(a.r2, qq, Rr[2]);
        VV::ld(a.vi0, qq, V[0]);
        VV::ld(a.vi1, qq, V[1]);
        VV::ld(a.vi2, qq, V[2]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) Ev[e] = (T)1;
        if (a.do_scatter) VV::ld(a.E, qq, Ev);
        // ---- NewtonianKinematicsStep: dr = v*dt (rounded), r = r + dr              newton.py:15-16
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                D[k][e] = R::mul(V[k][e], a.dt);
                Rr[k][e] = R::add(Rr[k][e], D[k][e]);
            }
        if (live_q) {
            if (!a.lazy) {
                VV::st(a.dr0, qs, D[0]);
                VV::st(a.dr1, qs, D[1]);
                VV::st(a.dr2, qs, D[2]);
            }
            VV::st(a.r0, qs, Rr[0]);
            VV::st(a.r1, qs, Rr[1]);
            VV::st(a.r2, qs, Rr[2]);
        }
        bool live[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) live[e] = live_q && q * VEC + e < a.N;
        // ---- ScatterIsotropicStep on each of the lane's particles                  light.py:303-331
        if (a.do_scatter) {
            bool hit[VEC], photon[VEC];
            bool all_photon = true, all_hit = true;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const pcl_i64 i = q * VEC + e;
                photon[e] = live[e] && (a.kind ? (a.kind[i] != 0) : true);
                const T pc = pcl_pcoll<T, USE_E, VAR_N>(a.np, a.A, a.n, a.h, a.c, D[0][e], D[1][e], D[2][e], Rr[0][e],
                                                        Rr[1][e], Rr[2][e], Ev[e]);
                T rand = (T)0, rtheta = (T)0, rphi = (T)0;
                pcl_u64 id = 0;
                if (photon[e]) {
                    if (a.rng_mode == PCL_RNG_PHX) {
                        id = (pcl_u64)(a.ids ? a.ids[i] : a.id_base + i);
                        rand = pcl_draw_rand<T>(id, a.step, k0, k1);
                    } else {
                        rand = a.rand[i];
                    }
                }
                hit[e] = photon[e] && (pc >= rand);
                T n0 = (T)0, n1 = (T)0, n2 = (T)0;
                if (hit[e]) {
                    if (a.rng_mode == PCL_RNG_PHX) {
                        pcl_draw_angles<T>(id, a.step, k0, k1, rtheta, rphi);
                    } else {
                        rtheta = a.rtheta[i];
                        rphi = a.rphi[i];
                    }
                    pcl_new_velocity<T>(a.c, rtheta, rphi, n0, n1, n2);
                }
                // hit: dv = v' - v_old, v = v' ; miss: dv = 0                       light.py:327-331
                DV[0][e] = hit[e] ? R::sub(n0, V[0][e]) : (T)0;
                DV[1][e] = hit[e] ? R::sub(n1, V[1][e]) : (T)0;
                DV[2][e] = hit[e] ? R::sub(n2, V[2][e]) : (T)0;
                if (hit[e]) {
                    V[0][e] = n0;
                    V[1][e] = n1;
                    V[2][e] = n2;
                }
                all_photon = all_photon && photon[e];
                all_hit = all_hit && hit[e];
                w_hits += (pcl_u32)__popcll(pcl_ballot(hit[e]));
            }
            if (a.lazy) {
                // v double buffer: every particle's (possibly new) velocity goes to the other buffer, whole
                // 16-byte stores; dr and dv are not written -- they stay derivable from (v_in, v_out, dt)
                if (live_q) {
                    VV::st(a.vo0, qs, V[0]);
                    VV::st(a.vo1, qs, V[1]);
                    VV::st(a.vo2, qs, V[2]);
                }
            } else {
                // photons always get dv written; plain Objects keep theirs (light.py:283 skips them)
                if (all_photon) {
                    VV::st(a.dv0, qs, DV[0]);
                    VV::st(a.dv1, qs, DV[1]);
                    VV::st(a.dv2, qs, DV[2]);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (photon[e]) {
                            a.dv0[qs * VEC + e] = DV[0][e];
                            a.dv1[qs * VEC + e] = DV[1][e];
                            a.dv2[qs * VEC + e] = DV[2][e];
                        }
                }
                if (all_hit) {
                    VV::st(a.vo0, qs, V[0]);
                    VV::st(a.vo1, qs, V[1]);
                    VV::st(a.vo2, qs, V[2]);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (hit[e]) {
                            a.vo0[qs * VEC + e] = V[0][e];
                            a.vo1[qs * VEC + e] = V[1][e];
                            a.vo2[qs * VEC + e] = V[2][e];
                        }
                }
            }
        }
        // ---- measure counters on the post-step state                    light.py:424-426, 385-399
        if (counters) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                w_sx += (pcl_u32)__popcll(pcl_ballot(live[e] && V[0][e] > (T)0));
                w_sy += (pcl_u32)__popcll(pcl_ballot(live[e] && V[1][e] > (T)0));
                w_sz += (pcl_u32)__popcll(pcl_ballot(live[e] && V[2][e] > (T)0));
            }
            for (int p = 0; p < a.n_planes; ++p) { // rolled: planes are rare, keep their state out of registers
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                pcl_u32 np = 0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const T x = pcl_pick<T>(ax, Rr[0][e], Rr[1][e], Rr[2][e]);
                    const T prev = R::sub(x, pcl_pick<T>(ax, D[0][e], D[1][e], D[2][e]));
                    const bool cross = live[e] && ((prev <= L && L <= x) || (prev >= L && L >= x));
                    np += (pcl_u32)__popcll(pcl_ballot(cross));
                }
                if (lane == 0 && np) atomicAdd(&s_cnt[4 + p], np);
            }
        }
    }
    if (lane == 0) {
        if (w_hits) atomicAdd(&s_cnt[0], w_hits);
        if (counters) {
            atomicAdd(&s_cnt[1], w_sx);
            atomicAdd(&s_cnt[2], w_sy);
            atomicAdd(&s_cnt[3], w_sz);
        }
    }
    __syncthreads();
    const int nslots = 4 + (a.n_planes > 0 ? a.n_planes : 0);
    if ((int)threadIdx.x < nslots && s_cnt[threadIdx.x]) atomicAdd(&a.cnt[threadIdx.x], (pcl_u64)s_cnt[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Level 2, fast path of the fused loop body: all-photon store, implicit ids (no compaction yet),
// device RNG, dr/dv implicit (PCL_FUSED_LAZY), sign counters only.  Same arithmetic as
// pcl_fused_body -- results are bit-identical -- with 11 instead of 22 live pointers, no
// per-plane state, and the wavelength factor pow((h*c)/E, -4) read from the store's cache
// (lam4[i], computed once per photon by k_lam4 with the very same device pow) instead of being
// re-evaluated every step.  104 B per fp64 particle-step (52 B in fp32).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct pcl_fast_args {
    T *r0, *r1, *r2;          // read + written
    const T *vi0, *vi1, *vi2; // v before the step
    T *vo0, *vo1, *vo2;       // v after the step (other half of the double buffer)
    const T *lam4;            // pow((h*c)/E, -4) per photon            (USE_E)
    const T *E;               // only dereferenced if the expression names E[gid]
    pcl_u64 *cnt;             // [0] hits, [1..3] sign counts
    pcl_i64 id_base, N;
    pcl_i64 ts;               // tile stride of the store (elements)
    T dt, A, n, c;
    pcl_u64 seed;
    pcl_u32 step;
    // GEN variants only (stores that have been compacted or hold plain Objects): explicit photon ids and the kind
    // bytes, dense arrays padded to whole 64-element groups; NULL = implicit ids / every particle is a photon
    const pcl_i64 *ids;
    const unsigned char *kind;
    // GEN variants on a store behind an alive mask (the delete path keeps removed photons' slots, pcl_step_fused_delete):
    // one bit per slot, set = a photon is there; N is then the store's extent.  Dead slots are moved like everybody else
    // (nobody reads them again) but neither scatter nor count.  r may lag behind by n_pend Newton moves (their dt in
    // pend_dt[]): they are applied first, with the velocity the photon had all along -- this step may change it.
    const pcl_u64 *alive;
    int n_pend;
    T pend_dt[PCL_PEND_MAX];   // run-length: pend_rep[p] moves of pend_dt[p]
    int pend_rep[PCL_PEND_MAX];
    pcl_nprof<T> np;          // ahead-of-time VAR_N kernels only
};

template <typename T, int VEC>
struct pcl_fast_tile {
    T R[3][VEC], V[3][VEC], L4[VEC];
    pcl_i64 ID[VEC];          // GEN only
    unsigned char KD[VEC];    // GEN only
    pcl_u64 AL;               // GEN only: the alive word of the lane's group (VEC divides 64: one word)
};

// q = the lane's group in the tiled rows, qd = the same group in the dense id / kind arrays (GEN)
template <typename T, bool USE_E, int VEC, bool GEN>
__device__ __forceinline__ void pcl_fast_load(const pcl_fast_args<T> &a, pcl_i64 q, pcl_i64 qd, pcl_fast_tile<T, VEC> &t) {
    typedef pcl_vec<T, VEC> VV;
    VV::ld(a.r0, q, t.R[0]);
    VV::ld(a.r1, q, t.R[1]);
    VV::ld(a.r2, q, t.R[2]);
    VV::ld(a.vi0, q, t.V[0]);
    VV::ld(a.vi1, q, t.V[1]);
    VV::ld(a.vi2, q, t.V[2]);
    if constexpr (USE_E) VV::ld(a.lam4, q, t.L4);
    if constexpr (GEN) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { // consecutive elements of whole (padded) groups: the compiler merges the loads
            t.ID[e] = a.ids ? a.ids[qd * VEC + e] : (pcl_i64)0;
            t.KD[e] = a.kind ? a.kind[qd * VEC + e] : (unsigned char)1;
        }
        t.AL = a.alive ? a.alive[(qd * VEC) >> 6] : ~(pcl_u64)0;
    }
}

template <typename T, bool USE_E, int VAR_N, int VEC, bool GEN = false>
__device__ __forceinline__ void pcl_fast_body(const pcl_fast_args<T> &a) {
    typedef pcl_rt<T> R;
    typedef pcl_vec<T, VEC> VV;
    __shared__ pcl_u32 s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    pcl_u32 w_hits = 0, w_sx = 0, w_sy = 0, w_sz = 0; // wave-uniform tallies (SGPRs)
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const pcl_i64 nq = (a.N + VEC - 1) / VEC; // VEC-wide groups
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
#if defined(PCL_XCD_MAP) // EXPERIMENT: blocks of one XCD (b % 8) walk neighbouring chunks
    const unsigned xb = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    pcl_i64 base = (pcl_i64)xb * blockDim.x;
#elif defined(PCL_TILE_MAP) // EXPERIMENT: the four blocks of a tile are b, b+8, b+16, b+24: one XCD per tile
    const unsigned g32 = blockIdx.x / 32u, r32 = blockIdx.x % 32u;
    const unsigned xb = (gridDim.x % 32u == 0u) ? g32 * 32u + (r32 % 8u) * 4u + r32 / 8u : blockIdx.x;
    pcl_i64 base = (pcl_i64)xb * blockDim.x;
#else
    pcl_i64 base = (pcl_i64)blockIdx.x * blockDim.x;
#endif
    // register double buffer: the next trip's loads are in flight while this trip computes
    pcl_fast_tile<T, VEC> cur;
    if (base < nq) {
        const pcl_i64 q0 = base + threadIdx.x < nq ? base + threadIdx.x : 0;
        pcl_fast_load<T, USE_E, VEC, GEN>(a, pcl_tq<VEC>(q0, a.ts), q0, cur);
    }
    for (; base < nq; base += stride) {
        const pcl_i64 q = base + threadIdx.x;
        const bool live_q = q < nq;
        pcl_fast_tile<T, VEC> nxt;
        const pcl_i64 nb = base + stride;
        if (nb < nq) {
            const pcl_i64 q1 = nb + threadIdx.x < nq ? nb + threadIdx.x : 0;
            pcl_fast_load<T, USE_E, VEC, GEN>(a, pcl_tq<VEC>(q1, a.ts), q1, nxt);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const pcl_i64 i = q * VEC + e;
            bool live = live_q && i < a.N;
            if constexpr (GEN) {
                live = live && ((cur.AL >> (i & 63)) & 1ull);
                for (int p = 0; p < a.n_pend; ++p) { // moves of earlier delete bodies that r has not seen yet
                    const T p0 = R::mul(cur.V[0][e], a.pend_dt[p]), p1 = R::mul(cur.V[1][e], a.pend_dt[p]), p2 = R::mul(cur.V[2][e], a.pend_dt[p]);
                    for (int w = 0; w < a.pend_rep[p]; ++w) {
                        cur.R[0][e] = R::add(cur.R[0][e], p0);
                        cur.R[1][e] = R::add(cur.R[1][e], p1);
                        cur.R[2][e] = R::add(cur.R[2][e], p2);
                    }
                }
            }
            // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
            const T d0 = R::mul(cur.V[0][e], a.dt), d1 = R::mul(cur.V[1][e], a.dt), d2 = R::mul(cur.V[2][e], a.dt);
            cur.R[0][e] = R::add(cur.R[0][e], d0);
            cur.R[1][e] = R::add(cur.R[1][e], d1);
            cur.R[2][e] = R::add(cur.R[2][e], d2);
            // scatter                                                                light.py:303-315
            T pc = pcl_pcoll<T, false, VAR_N>(a.np, a.A, a.n, (T)0, a.c, d0, d1, d2, cur.R[0][e], cur.R[1][e], cur.R[2][e],
                                              a.E[pcl_tix(live ? i : 0, a.ts)]);
            if constexpr (USE_E) pc = R::mul(pc, cur.L4[e]);
            pcl_u64 id = (pcl_u64)(a.id_base + i);
            bool photon = true;
            if constexpr (GEN) { // explicit ids after a compaction; plain Objects are moved but never scattered (light.py:283)
                if (a.ids) id = (pcl_u64)cur.ID[e];
                photon = cur.KD[e] != 0;
            }
            const T rand = pcl_draw_rand<T>(id, a.step, k0, k1);
            const bool hit = live && photon && (pc >= rand);
            if (hit) {
                T rtheta, rphi;
                pcl_draw_angles<T>(id, a.step, k0, k1, rtheta, rphi);
                pcl_new_velocity<T, true>(a.c, rtheta, rphi, cur.V[0][e], cur.V[1][e], cur.V[2][e]);
            }
            w_hits += (pcl_u32)__popcll(pcl_ballot(hit));
            w_sx += (pcl_u32)__popcll(pcl_ballot(live && cur.V[0][e] > (T)0));
            w_sy += (pcl_u32)__popcll(pcl_ballot(live && cur.V[1][e] > (T)0));
            w_sz += (pcl_u32)__popcll(pcl_ballot(live && cur.V[2][e] > (T)0));
        }
        if (live_q) {
            const pcl_i64 qs = pcl_tq<VEC>(q, a.ts);
            VV::st(a.r0, qs, cur.R[0]);
            VV::st(a.r1, qs, cur.R[1]);
            VV::st(a.r2, qs, cur.R[2]);
            VV::st(a.vo0, qs, cur.V[0]);
            VV::st(a.vo1, qs, cur.V[1]);
            VV::st(a.vo2, qs, cur.V[2]);
        }
        cur = nxt;
    }
    if ((threadIdx.x & 63) == 0) {
        if (w_hits) atomicAdd(&s_cnt[0], w_hits);
        atomicAdd(&s_cnt[1], w_sx);
        atomicAdd(&s_cnt[2], w_sy);
        atomicAdd(&s_cnt[3], w_sz);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&a.cnt[threadIdx.x], (pcl_u64)s_cnt[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Level 2, K consecutive fused steps in ONE pass over the store.  Particles do not interact, so a photon
// can be loaded once, stepped K times in registers (Newton, scatter with launch index step + k, sign
// counters of step k) and stored once: the result is exactly that of K launches of pcl_fast_body -- same
// operations in the same order per photon -- while the HBM traffic per particle-step drops from 104 B to
// 128/K B (fp64).  The velocity is updated in place and the velocity before the LAST step goes to the vp
// rows, so dr = vp*dt and dv = v - vp stay implicit exactly as after a single lazy step.
// cnt[(4 + n_planes)*k + {0: hits, 1..3: sign counts, 4..: plane crossings}] for k = 0..K-1.
// cnt[(4 + n_planes)*K] = dense passes of the hit queues over the whole launch (the work tally of the VALU roofline record);
// cnt[(4 + n_planes)*K + 1] = wave-steps that took the saturation shortcut (SATP variants).
// cnt[(4 + n_planes)*K + 2], [.. + 3] = shader cycles and 100 MHz ticks between start and end, summed over the workgroups:
// their quotient x 100 MHz is the clock the chip held under THIS launch (the VALU roofline's ceiling is in cycles).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// What clock did the chip hold under this launch?  The kernels that are bound by VALU issue have their ceiling in
// SIMD-cycles, and the chip lowers its clock under exactly such kernels (2.1 - 2.3 GHz instead of the 2.4 GHz of the data
// sheet).  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6):
// every workgroup notes both when it starts and adds the two differences to a pair of launch-wide sums when it ends --
// two scalar reads at either end and two atomics per workgroup.  The start values wait in LDS, not in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pcl_clock_begin(pcl_u64 *s_clk) {
    if (threadIdx.x == 0) {
        s_clk[1] = __builtin_amdgcn_s_memrealtime();
        s_clk[0] = __builtin_amdgcn_s_memtime();
    }
}
// (call after the workgroup's last barrier; sums[0] += shader cycles, sums[1] += 100 MHz ticks)
__device__ __forceinline__ void pcl_clock_end(const pcl_u64 *s_clk, pcl_u64 *sums) {
    if (threadIdx.x == 0) {
        const pcl_u64 t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&sums[0], t1 - s_clk[0]);
        atomicAdd(&sums[1], w1 - s_clk[1]);
    }
}

#define PCL_MULTI_MAX 64
template <typename T>
struct pcl_multi_args {
    T *r0, *r1, *r2;    // read + written
    T *v0, *v1, *v2;    // read + written
    T *vp0, *vp1, *vp2; // written: v before the last of the K steps
    const T *lam4;      // pow((h*c)/E, -4) per photon            (USE_E)
    const T *E;         // only dereferenced if the expression names E[gid]
    pcl_u64 *cnt;       // [K][4 + n_planes]: hits, sign x/y/z, plane crossings
    pcl_i64 id_base, N;
    pcl_i64 ts;
    T dt, A, n, c;
    pcl_u64 seed;
    pcl_u32 step; // launch index of the first of the K steps
    int K;
    int n_planes;             // 0..PCL_MAXPL measure planes (physicl/light.py:385-399)
    int plane_ax[PCL_MAXPL];
    T plane_L[PCL_MAXPL];
    pcl_nprof<T> np;          // ahead-of-time VAR_N kernels only
};

// ------------------------------------------------------------------------------------------------
// The K-step pass.  The scatter branch (second Philox block, two sincos, the new velocity: ~750 SIMD-cycles) is more than
// half of a step's arithmetic but only the hit photons need it; executed in place it costs every wave the full branch
// however few of its lanes hit.  So each wave queues the hits of its photons (slots from the ballots' prefix counts) and
// processes them DENSELY -- item j by lane j, ceil(hits / 64) passes -- with the same operations on the same operands per
// photon, so nothing changes in the results; only which lane executes them.  (A workgroup-wide queue packs better still,
// but its two barriers per step cost more than that: measured 7 % slower.)  What a dense pass costs does not depend on
// how many of its 64 lanes carry a hit, so a wave should own as many photons as it can: 256 (NQ = 2 in fp64, VEC = 4 in
// fp32) fill a pass with 54 hits at a 21 % hit fraction where 128 fill it with 27.
// Rounds 2-4 kept a lane's photons in registers (pcl_multi_body: 128 per wave; pcl_multi_body_nq: 256 per wave at 72 VGPRs
// of state, 168 in all, 20 of them spilled to scratch, one wave per SIMD fewer) and handed the new velocities back through
// eight selects per photon and step.  Here r (and lam4, and the odd step's random words) stay in registers; v and |v dt| of
// every photon live in LDS ("home", 32 B per fp64 photon): a step READS them from there, the dense pass WRITES the new ones
// straight into the owner's home, and nothing is handed back.  40 VGPRs of state per lane at four photons: 124 in all, four
// waves per SIMD, nothing spilled (measured at 1e8 photons: 0.38 instead of 0.50 ms per step at a 21 % hit fraction, faster
// than the 128-photon form at every hit fraction from 0.52 down).
//   * per photon and step: 4 ds_read_b64 instead of 8 v_cndmask + 4 conditional ds_reads; the hit queue carries only the
//     owners (2 B per entry, room for every photon of the wave: never more than one round);
//   * the sign counts of step k are those of the velocities step k + 1 reads (Newton does not change v): they are tallied
//     there, the last step's when the velocities are read for the final store -- no second look at v after the dense pass;
//   * the velocity before the last step (vp rows) is copied from the homes when the last step begins.
// Same operations on the same operands per photon, in the same order, as K launches of pcl_fast_body: bit-identical state and rows
// (tests/test_gpu_multi.py runs every form against the single steps and the oracle).
// ------------------------------------------------------------------------------------------------
// ``pcoll >= uniform`` without building the uniform: R::uniform is an integer m of 53 (fp32: 24) random bits times 2^-53 (2^-24),
// both exact, so  pcoll >= m * 2^-53  <=>  m <= floor(pcoll * 2^53)  (the scaling is exact too; a pcoll of 1 or more passes whatever
// m, a NaN or a negative one never does: threshold -1) -- the reference's ``pcoll >= rand`` (physicl/light.py:243, 306) bit for bit,
// NaN and inf included (tests/test_draw_threshold_cpu.py).  Where pcoll only changes when the photon scatters (constant n, no
// wavelength term: A * n * |v dt|) the threshold is worked out once per velocity and a step's decision is shifts and one integer compare.
template <typename T> struct pcl_thr;
template <> struct pcl_thr<double> {
    typedef pcl_i64 thr_t;
    static __device__ __forceinline__ thr_t threshold(double pc) {
        if (!(pc >= 0.0)) return -1;
        const double y = pc * 9007199254740992.0;
        return y >= 9007199254740992.0 ? (thr_t)9007199254740992ll : (thr_t)y; // (truncation = floor: y >= 0)
    }
    static __device__ __forceinline__ thr_t draw(pcl_u32 a, pcl_u32 b) { return (thr_t)(((pcl_u64)(a >> 5) << 26) | (pcl_u64)(b >> 6)); } // pcl_u53
    static __device__ __forceinline__ double as_real(thr_t t) { return __longlong_as_double(t); } // (the bits, kept where a T is kept)
    static __device__ __forceinline__ thr_t of_real(double x) { return __double_as_longlong(x); }
};
template <> struct pcl_thr<float> {
    typedef int thr_t;
    static __device__ __forceinline__ thr_t threshold(float pc) {
        if (!(pc >= 0.0f)) return -1;
        const float y = pc * 16777216.0f;
        return y >= 16777216.0f ? (thr_t)16777216 : (thr_t)y;
    }
    static __device__ __forceinline__ thr_t draw(pcl_u32 a, pcl_u32 b) { (void)b; return (thr_t)(a >> 8); } // pcl_rt<float>::uniform
    static __device__ __forceinline__ float as_real(thr_t t) { return __int_as_float(t); }
    static __device__ __forceinline__ thr_t of_real(float x) { return __float_as_int(x); }
};

template <typename T, int NP>
struct pcl_multi_home {
    T c[4][NP][256];                   // v0, v1, v2, |v dt| of photon p of thread t (component-major: lanes read consecutive words)
    unsigned short owner[4][NP * 64 + 64]; // per wave: the step's hits, p * 64 + lane; behind them a slot per lane for the writes of the
                                           // lanes that did not hit (the queue is written without a branch)
};

template <typename T, bool USE_E, int VAR_N, int VEC, int NQ, bool SATP = false>
__device__ __forceinline__ void pcl_multi_body_lds(const pcl_multi_args<T> &a) {
    typedef pcl_rt<T> R;
    typedef pcl_vec<T, VEC> VV;
    typedef pcl_thr<T> D;
    // constant n, no wavelength term: pcoll = A * n * |v dt| only changes when the photon scatters -- its home keeps the decision's
    // integer threshold (pcl_thr) in place of |v dt|, a step decides with shifts and one integer compare
    constexpr bool THR = VAR_N == 0 && !USE_E;
    constexpr int NP = VEC * NQ; // photons per lane
    __shared__ pcl_u32 s_cnt[(4 + PCL_MAXPL) * PCL_MULTI_MAX];
    const int nslots = 4 + a.n_planes;
    __shared__ pcl_multi_home<T, NP> s_h;
    __shared__ pcl_u32 s_pass, s_sat;
    __shared__ pcl_u64 s_clk[2];
    for (int k = threadIdx.x; k < nslots * a.K; k += blockDim.x) s_cnt[k] = 0;
    if (threadIdx.x == 0) s_pass = 0, s_sat = 0;
    pcl_clock_begin(s_clk);
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lane0 = lane == 0;
    const pcl_i64 nq = (a.N + VEC - 1) / VEC;
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x * NQ;
    pcl_u32 w_passes = 0; // dense passes this wave made (wave-uniform)
    pcl_u32 w_sat = 0;    // SATP: wave-steps whose expression values came from the saturation shortcut
    for (pcl_i64 base = (pcl_i64)blockIdx.x * blockDim.x * NQ; base < nq; base += stride) {
        T Rr[3][NP], L4[NP], Ev[NP];
        pcl_u32 wodd0[NP], wodd1[NP]; // the decision block's second half, waiting for the odd step
        bool live[NP];
        // the wave's votes are taken on bare compares and masked with the photons' live masks in the scalar unit: a vote on a
        // compound predicate makes the compiler turn it into an integer and compare again (v_cndmask + v_cmp, 8 cycles a vote)
        pcl_u64 lm[NP];
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            const pcl_i64 q = base + (pcl_i64)g * blockDim.x + tid;
            const bool live_q = q < nq;
            const pcl_i64 qs = pcl_tq<VEC>(live_q ? q : 0, a.ts);
            T t0[VEC], t1[VEC], t2[VEC], t3[VEC], t4[VEC], t5[VEC], t6[VEC];
            VV::ld(a.r0, qs, t0);
            VV::ld(a.r1, qs, t1);
            VV::ld(a.r2, qs, t2);
            VV::ld(a.v0, qs, t3);
            VV::ld(a.v1, qs, t4);
            VV::ld(a.v2, qs, t5);
            if constexpr (USE_E) VV::ld(a.lam4, qs, t6);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int p = g * VEC + e;
                live[p] = live_q && q * VEC + e < a.N;
                lm[p] = pcl_ballot(live[p]);
                Rr[0][p] = t0[e];
                Rr[1][p] = t1[e];
                Rr[2][p] = t2[e];
                L4[p] = (T)1;
                if constexpr (USE_E) L4[p] = t6[e];
                Ev[p] = a.E[pcl_tix(live[p] ? q * VEC + e : 0, a.ts)];
                s_h.c[0][p][tid] = t3[e];
                s_h.c[1][p][tid] = t4[e];
                s_h.c[2][p][tid] = t5[e];
                // |dr| = |v * dt| only changes when the photon scatters: kept beside v, recomputed with the new velocity
                const T nm0 = pcl_step_norm<T>(R::mul(t3[e], a.dt), R::mul(t4[e], a.dt), R::mul(t5[e], a.dt));
                if constexpr (THR) s_h.c[3][p][tid] = D::as_real(D::threshold(R::mul(R::mul(a.A, a.n), nm0)));
                else s_h.c[3][p][tid] = nm0;
            }
        }
        pcl_u32 hits_prev = 0; // the hits of the step before (wave-uniform)
        for (int k = 0; k < a.K; ++k) {
            const pcl_u32 st = a.step + (pcl_u32)k;
            // the 20 Philox round keys are loop-invariant; hoisted out of the k loop they cost 20 SGPRs and push other scalars
            // into VGPR-lane spills (v_readlane in the loop).  Opaque copies make the compiler rebuild them per iteration on the
            // otherwise idle scalar unit instead.
            pcl_u32 kk0 = k0, kk1 = k1;
            asm volatile("" : "+s"(kk0), "+s"(kk1));
            const bool new_block = (st & 1u) == 0u || k == 0; // wave-uniform
            if (k + 1 == a.K) { // the velocity before the LAST step is what dv = v - v_prev needs: copied from the homes
                int tid_here = tid; // (opaque: the addresses below are worked out HERE, once per trip, not hoisted out of the K loop
                asm volatile("" : "+v"(tid_here)); // into registers that would have to live -- or spill -- through every step)
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    const pcl_i64 q = base + (pcl_i64)g * blockDim.x + tid_here;
                    if (q < nq) {
                        T t3[VEC], t4[VEC], t5[VEC];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            t3[e] = s_h.c[0][g * VEC + e][tid];
                            t4[e] = s_h.c[1][g * VEC + e][tid];
                            t5[e] = s_h.c[2][g * VEC + e][tid];
                        }
                        const pcl_i64 qs = pcl_tq<VEC>(q, a.ts);
                        VV::st(a.vp0, qs, t3);
                        VV::st(a.vp1, qs, t4);
                        VV::st(a.vp2, qs, t5);
                    }
                }
            }
            pcl_u32 w_hits = 0, w_sx = 0, w_sy = 0, w_sz = 0;
            T pcn[NP], nmv[NP]; // per photon: the collision probability, or (SATP) the expression's value and |v dt|
            bool unsat = false;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const T v0 = s_h.c[0][p][tid], v1 = s_h.c[1][p][tid], v2 = s_h.c[2][p][tid], nm = s_h.c[3][p][tid];
                // sign counts of the state step k - 1 left behind (Newton does not touch v)                light.py:424-426
                w_sx += (pcl_u32)__popcll(pcl_ballot(v0 > (T)0) & lm[p]);
                w_sy += (pcl_u32)__popcll(pcl_ballot(v1 > (T)0) & lm[p]);
                w_sz += (pcl_u32)__popcll(pcl_ballot(v2 > (T)0) & lm[p]);
                // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
                const T d0 = R::mul(v0, a.dt), d1 = R::mul(v1, a.dt), d2 = R::mul(v2, a.dt);
                Rr[0][p] = R::add(Rr[0][p], d0);
                Rr[1][p] = R::add(Rr[1][p], d1);
                Rr[2][p] = R::add(Rr[2][p], d2);
                // (the plane crossings of this move are counted behind this loop: a loop over the planes HERE would cut the NP photons'
                // moves and decisions into NP basic blocks, each waiting for its own LDS reads)
                // scatter decision                                                       light.py:303-308
#ifdef PCL_N_EXPR
                if constexpr (SATP && VAR_N != 0) { // by exp's saturation shortcut first (pcl_n_expr_sat): settled wave-wide below
                    pcn[p] = pcl_n_expr_sat<T>(Rr[0][p], Rr[1][p], Rr[2][p], d0, d1, d2, Ev[p], unsat);
                    nmv[p] = nm;
                } else
#endif
                if constexpr (THR) {
                    pcn[p] = nm; // (the threshold's bits)
                } else {
                    T pc = pcl_pcoll_norm<T, false, VAR_N>(a.np, a.A, a.n, (T)0, a.c, nm, d0, d1, d2, Rr[0][p], Rr[1][p], Rr[2][p], Ev[p]);
                    if constexpr (USE_E) pc = R::mul(pc, L4[p]);
                    pcn[p] = pc;
                }
            }
            // plane crossings of this step's move (r - dr, r)                            light.py:385-399
            if (a.n_planes > 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const T d0 = R::mul(s_h.c[0][p][tid], a.dt), d1 = R::mul(s_h.c[1][p][tid], a.dt), d2 = R::mul(s_h.c[2][p][tid], a.dt); // (as above)
                    for (int pl = 0; pl < a.n_planes; ++pl) {
                        const int ax = a.plane_ax[pl];
                        const T L = a.plane_L[pl];
                        const T x = pcl_pick<T>(ax, Rr[0][p], Rr[1][p], Rr[2][p]);
                        const T prev = R::sub(x, pcl_pick<T>(ax, d0, d1, d2));
                        const pcl_u32 nc = (pcl_u32)__popcll(((pcl_ballot(prev <= L) & pcl_ballot(L <= x)) | (pcl_ballot(prev >= L) & pcl_ballot(L >= x))) & lm[p]);
                        if (lane0 && nc) atomicAdd(&s_cnt[nslots * k + 4 + pl], nc);
                    }
                }
            }
            // row k - 1: its hits and the sign counts of the state it left, ONE LDS atomic of four lanes (one per column) instead of
            // four single-lane ones, each in a basic block of its own
            if (k > 0 && lane < 4) {
                const pcl_u32 val = lane == 0 ? hits_prev : (lane == 1 ? w_sx : (lane == 2 ? w_sy : w_sz));
                if (val) atomicAdd(&s_cnt[nslots * (k - 1) + lane], val);
            }
#ifdef PCL_N_EXPR
            if constexpr (SATP && VAR_N != 0) {
                if (pcl_ballot(unsat) != 0ull) { // (wave-uniform) some argument is in exp's working range: the ordinary way
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const T d0 = R::mul(s_h.c[0][p][tid], a.dt), d1 = R::mul(s_h.c[1][p][tid], a.dt), d2 = R::mul(s_h.c[2][p][tid], a.dt);
                        pcn[p] = pcl_n_expr_val<T, VAR_N>(a.np, Rr[0][p], Rr[1][p], Rr[2][p], d0, d1, d2, Ev[p]);
                    }
                } else {
                    ++w_sat;
                    asm volatile("" : "+v"(w_sat));
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) { // pcl_pcoll_norm with the expression's value in hand
                    T pc = R::mul(R::mul(a.A, pcn[p]), nmv[p]);
                    if constexpr (USE_E) pc = R::mul(pc, L4[p]);
                    pcn[p] = pc;
                }
            }
#endif
            // The draws of all NP photons first, in one straight-line block: a Philox chain is ten dependent rounds, and the
            // queue writes below are control flow -- with a photon's draw between two of them the compiler kept the NP chains in
            // NP basic blocks, one after the other (5.1 cycles an instruction at four waves per SIMD,
            // profiles/r05_valu_issue_probe.txt); side by side they fill each other's latencies.
            T rnd[NP];
            typename D::thr_t mdr[NP]; // THR: the draw as the integer it is
#ifndef PCL_PHILOX_SIDE_BY_SIDE
#define PCL_PHILOX_SIDE_BY_SIDE 2 /* chains the scheduler may interleave at a time */
#endif
            if (new_block && (st & 1u) == 0u) { // decision block of steps (st, st | 1): computed once for the pair
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (p > 0 && p % PCL_PHILOX_SIDE_BY_SIDE == 0) __builtin_amdgcn_sched_barrier(0);
                    const pcl_u64 id = (pcl_u64)(a.id_base + (base + (pcl_i64)(p / VEC) * blockDim.x + tid) * VEC + (p % VEC));
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), st >> 1, 0u, kk0, kk1);
                    if constexpr (THR) mdr[p] = D::draw(w.x, w.y);
                    else rnd[p] = R::uniform(w.x, w.y);
                    wodd0[p] = w.z;
                    wodd1[p] = w.w;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (new_block) { // a launch that starts on an odd step: the second half of the block of (st - 1, st)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (p > 0 && p % PCL_PHILOX_SIDE_BY_SIDE == 0) __builtin_amdgcn_sched_barrier(0);
                    const pcl_u64 id = (pcl_u64)(a.id_base + (base + (pcl_i64)(p / VEC) * blockDim.x + tid) * VEC + (p % VEC));
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), st >> 1, 0u, kk0, kk1);
                    if constexpr (THR) mdr[p] = D::draw(w.z, w.w);
                    else rnd[p] = R::uniform(w.z, w.w);
                    wodd0[p] = w.z;
                    wodd1[p] = w.w;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (THR) mdr[p] = D::draw(wodd0[p], wodd1[p]);
                    else rnd[p] = R::uniform(wodd0[p], wodd1[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                // (NaN compares false, +inf true: the reference's ``pcoll >= rand``; THR: the same decision as an integer compare, pcl_thr)
                pcl_u64 bal;
                if constexpr (THR) bal = pcl_ballot(mdr[p] <= D::of_real(pcn[p])) & lm[p];
                else bal = pcl_ballot(pcn[p] >= rnd[p]) & lm[p];
                const bool hit = (bal >> lane) & 1ull;
                // (unconditional: a lane that did not hit writes its own slot behind the queue -- an ``if (hit)`` is a basic block per
                // photon, and the NP compares, votes and writes schedule better as one)
                // (three photons per lane: the branches are kept -- the straight-line form costs the two registers that decide between
                // five waves per SIMD and four for that form)
                const pcl_u32 pos = w_hits + __builtin_amdgcn_mbcnt_hi((pcl_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((pcl_u32)bal, 0u));
                if constexpr (NP >= 4) {
                    s_h.owner[wave][hit ? pos : (pcl_u32)(NP * 64 + lane)] = (unsigned short)(p * 64 + lane);
                } else {
                    if (hit) s_h.owner[wave][pos] = (unsigned short)(p * 64 + lane);
                }
                w_hits += (pcl_u32)__popcll(bal);
            }
            // the queue and the homes of a wave's photons are private to the wave: LDS executes a wave's accesses in order, so
            // a compiler-level fence is all the hand-over needs -- no workgroup barrier anywhere in the K loop
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            w_passes += (w_hits + 63u) >> 6;
            asm volatile("" : "+v"(w_passes)); // lives in a VGPR: no lane spill for a tally
#ifdef PCL_HIT_HIST /* debug build (PCL_RTC_EXTRA=PCL_HIT_HIST): how many hits did this wave queue in this step?  129 bins */
            if (lane0) atomicAdd(&a.cnt[nslots * a.K + 4 + (w_hits < 128u ? w_hits : 128u)], (pcl_u64)1);
#endif
            // the scatter itself, densely: item j by lane j, the new velocity straight into its owner's home  light.py:309-311
            for (pcl_u32 j = (pcl_u32)lane; j < w_hits; j += 64) {
                const pcl_u32 o = s_h.owner[wave][j];
                const pcl_u32 op = o >> 6, ot = (pcl_u32)wave * 64u + (o & 63u);
                const pcl_u64 id = (pcl_u64)(a.id_base + (base + (pcl_i64)(op / VEC) * blockDim.x + (pcl_i64)ot) * VEC + (pcl_i64)(op % VEC));
                T rtheta, rphi;
                pcl_draw_angles<T>(id, st, kk0, kk1, rtheta, rphi);
                T o0, o1, o2;
                pcl_new_velocity<T, true>(a.c, rtheta, rphi, o0, o1, o2);
                s_h.c[0][op][ot] = o0;
                s_h.c[1][op][ot] = o1;
                s_h.c[2][op][ot] = o2;
                const T nm1 = pcl_step_norm<T>(R::mul(o0, a.dt), R::mul(o1, a.dt), R::mul(o2, a.dt));
                if constexpr (THR) s_h.c[3][op][ot] = D::as_real(D::threshold(R::mul(R::mul(a.A, a.n), nm1)));
                else s_h.c[3][op][ot] = nm1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            hits_prev = w_hits; // (joins row k's sign counts in one atomic, above or behind the loop)
        }
        // the last step's sign counts, and the store
        pcl_u32 w_sx = 0, w_sy = 0, w_sz = 0;
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            const pcl_i64 q = base + (pcl_i64)g * blockDim.x + tid;
            T t0[VEC], t1[VEC], t2[VEC], t3[VEC], t4[VEC], t5[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int p = g * VEC + e;
                t0[e] = Rr[0][p];
                t1[e] = Rr[1][p];
                t2[e] = Rr[2][p];
                t3[e] = s_h.c[0][p][tid];
                t4[e] = s_h.c[1][p][tid];
                t5[e] = s_h.c[2][p][tid];
                w_sx += (pcl_u32)__popcll(pcl_ballot(t3[e] > (T)0) & lm[p]);
                w_sy += (pcl_u32)__popcll(pcl_ballot(t4[e] > (T)0) & lm[p]);
                w_sz += (pcl_u32)__popcll(pcl_ballot(t5[e] > (T)0) & lm[p]);
            }
            if (q < nq) {
                const pcl_i64 qs = pcl_tq<VEC>(q, a.ts);
                VV::st(a.r0, qs, t0);
                VV::st(a.r1, qs, t1);
                VV::st(a.r2, qs, t2);
                VV::st(a.v0, qs, t3);
                VV::st(a.v1, qs, t4);
                VV::st(a.v2, qs, t5);
            }
        }
        if (lane < 4) {
            const pcl_u32 val = lane == 0 ? hits_prev : (lane == 1 ? w_sx : (lane == 2 ? w_sy : w_sz));
            if (val) atomicAdd(&s_cnt[nslots * (a.K - 1) + lane], val);
        }
    }
    if (lane0 && w_passes) atomicAdd(&s_pass, w_passes);
    if (lane0 && w_sat) atomicAdd(&s_sat, w_sat);
    __syncthreads();
    for (int k = threadIdx.x; k < nslots * a.K; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&a.cnt[k], (pcl_u64)s_cnt[k]);
    if (threadIdx.x == 0 && s_pass) atomicAdd(&a.cnt[nslots * a.K], (pcl_u64)s_pass);
    if (threadIdx.x == 0 && s_sat) atomicAdd(&a.cnt[nslots * a.K + 1], (pcl_u64)s_sat);
    pcl_clock_end(s_clk, &a.cnt[nslots * a.K + 2]);
}

// ------------------------------------------------------------------------------------------------
// Level 2, K whole passes of a loop whose body holds an isotropic-scatter phase and/or a delete phase --
//   [Newton, ScatterIsotropic] | [Newton, ScatterDelete] | [Newton, ScatterIsotropic, Newton, ScatterDelete] (either order)
// -- in ONE pass over the store and (with a delete phase) ONE compaction afterwards.  It is the general form of
// pcl_multi_body_lds: any store (explicit ids after earlier compactions, plain Objects mixed in), photons that are removed
// stop where the reference's list.remove takes them out (physicl/__init__.py:455-459), and every phase tallies its
// own measure row on the particles alive after it.  Per particle the operations are those of the single steps
// (pcl_fast_body / k_newton_mask) in the same order with launch index step + phase number, so state, masks and rows
// are bit-identical to running the passes one launch at a time (tests/test_gpu_mixed.py).
// Geometry: one workgroup per 2048-particle tile, wave w owns rows 8w..8w+7 of 64 particles, two rows per trip
// (lane == particle within a row, so a ballot IS the row's keep-mask in particle order).
// cnt[(5 + n_planes) * phase + {0: alive after the phase, 1: hits | removed, 2..4: sign counts, 5..: plane crossings}].
// ------------------------------------------------------------------------------------------------
#define PCL_MIXED_MAXPH 2
template <typename T>
struct pcl_mixed_args {
    T *r0, *r1, *r2;    // read + written
    T *v0, *v1, *v2;    // read + written
    T *vp0, *vp1, *vp2; // written: v before the LAST isotropic-scatter phase (dv = v - vp stays implicit)
    const T *lam4;      // pow((h*c)/E, -4) per photon            (USE_E)
    const T *E;         // only dereferenced if the expression names E[gid]
    const pcl_i64 *ids;        // NULL: id = id_base + index
    const unsigned char *kind; // NULL: every particle is a photon
    pcl_u64 *masks;     // [tiles * 32] keep-masks, written when the loop has a delete phase
    int *tile_keep;     // [tiles]
    pcl_u64 *cnt;
    pcl_i64 id_base, N;
    pcl_i64 ts;
    // ``inplace`` (all-photon stores, loops with a delete phase): the survivors of a wave's 512-slot segment (its eight rows of
    // the tile) are written back to the FRONT of that segment, in order, with every field they own -- r, v, the vprev rows, E,
    // lam4, the id -- so that the store needs no compaction pass behind the launch: ``masks`` then holds a prefix of set bits per
    // segment, which is what the next launch takes as ``alive_in`` (N = the extent in slots).  The global compaction only runs
    // once fewer than half of the slots are alive (pcl_step_mixed_multi).  Stable: segments in order, each in order.
    const pcl_u64 *alive_in; // NULL: slots [0, N) all hold a particle; else one bit per slot, a prefix per 512-slot segment
    T *E_w, *lam4_w;         // inplace: the E and lam4 rows, writable (lam4_w NULL: the cache is not in use)
    pcl_i64 *ids_out;        // inplace: the store's id array (may be the array ``ids`` points to)
    int inplace;
    int move_vp;             // inplace: the vprev rows hold something (a scatter phase of this launch, or an earlier implicit dv)
    T dt, A, n, c;      // isotropic phase: kernel constants after the reference's swap (light.py:287)
    T An_del;           // delete phase: A * n rounded once (light.py:243)
    pcl_u64 seed;
    pcl_u32 step;       // launch index of the first phase
    int K;              // passes
    int P;              // phases per pass (1..PCL_MIXED_MAXPH)
    int phase_del[PCL_MIXED_MAXPH]; // 1: ScatterDeleteStep phase, 0: ScatterIsotropicStep phase
    int has_delete;
    int last_iso;       // index (0 .. K*P-1) of the last isotropic phase, -1 if the loop has none
    int n_planes;
    int plane_ax[PCL_MAXPL];
    T plane_L[PCL_MAXPL];
    pcl_nprof<T> np;    // ahead-of-time VAR_N kernels only
};

template <typename T>
struct pcl_mixed_queue {
    pcl_u64 id[256 * 2]; // photon id of the queued hit
    T out[4][256 * 2];   // the new velocity and its step length |v' * dt|
};

// particles a wave's segment (rows 8w .. 8w+7 of its tile: 512 slots) holds on entry: the set bits of its eight alive words --
// a prefix of the segment by construction (inplace launches leave it so) -- or, for a dense store, what lies below N
template <typename T>
__device__ __forceinline__ int pcl_mixed_seg_count(const pcl_mixed_args<T> &a, pcl_i64 tile, int wave) {
    if (a.alive_in) {
        int n = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) n += (int)__popcll(a.alive_in[tile * 32 + wave * 8 + j]);
        return n;
    }
    const pcl_i64 left = a.N - (tile * PCL_T + (pcl_i64)wave * 512);
    return left <= 0 ? 0 : (left >= 512 ? 512 : (int)left);
}

// the end of a trip of an inplace launch, one row: the row's survivors (wave mask ``m``) go to positions kept .. kept + popc(m) - 1
// of the wave's segment with everything they own.  ``pos_in``: the row's first position in the segment.  The vprev rows were
// written at the particle's OLD position when the last scatter phase began (or hold an earlier launch's): read back from
// there.  A row's destinations lie below its own first position and below every later row's: nothing unread is overwritten.
template <typename T, bool USE_E>
__device__ __forceinline__ void pcl_mixed_put(const pcl_mixed_args<T> &a, pcl_i64 seg0, int pos_in, pcl_u32 kept, pcl_u64 m, int lane, T r0, T r1,
                                              T r2, T v0, T v1, T v2, T L4, pcl_u64 id) {
    if (!((m >> lane) & 1ull)) return;
    // (the addresses are worked out HERE, from an opaque copy of the lane: hoisted out of the phase loop they would have to live --
    // or spill -- through every phase; what a particle owns goes out first, the vprev rows are read back last: few values live at a time)
    asm volatile("" : "+v"(lane));
    const pcl_u32 rank = __builtin_amdgcn_mbcnt_hi((pcl_u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((pcl_u32)m, 0u));
    const pcl_i64 i_out = seg0 + (pcl_i64)(kept + rank);
    const pcl_i64 tout = pcl_tix(i_out, a.ts);
    const pcl_i64 tin = pcl_tix(seg0 + pos_in + lane, a.ts);
    a.ids_out[i_out] = (pcl_i64)id;
    a.E_w[tout] = a.E_w[tin]; // (read back from where the particle was: E need not live in a register through the phases)
    if constexpr (USE_E) {
        if (a.lam4_w) a.lam4_w[tout] = L4;
    }
    a.r0[tout] = r0;
    a.r1[tout] = r1;
    a.r2[tout] = r2;
    a.v0[tout] = v0;
    a.v1[tout] = v1;
    a.v2[tout] = v2;
    if (a.move_vp) {
        const T p0 = a.vp0[tin], p1 = a.vp1[tin], p2 = a.vp2[tin];
        a.vp0[tout] = p0;
        a.vp1[tout] = p1;
        a.vp2[tout] = p2;
    }
}

// the eight alive words of a wave's segment after an inplace launch: a prefix of ``kept`` set bits
template <typename T>
__device__ __forceinline__ void pcl_mixed_put_masks(const pcl_mixed_args<T> &a, pcl_i64 tile, int wave, int lane, int kept) {
    if (lane < 8) {
        const int n = kept - 64 * lane;
        a.masks[tile * 32 + wave * 8 + lane] = n >= 64 ? ~(pcl_u64)0 : (n <= 0 ? (pcl_u64)0 : (((pcl_u64)1 << n) - 1ull));
    }
}

template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ void pcl_mixed_body(const pcl_mixed_args<T> &a) {
    typedef pcl_rt<T> R;
    constexpr int NE = 2; // rows (particles per lane) per trip
    __shared__ pcl_u32 s_cnt[(5 + PCL_MAXPL) * PCL_MULTI_MAX];
    __shared__ pcl_mixed_queue<T> s_q;
    __shared__ int s_keep[4];
    const int nslots = 5 + a.n_planes;
    const int n_ph = a.K * a.P;
    for (int k = threadIdx.x; k < nslots * n_ph; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool lane0 = lane == 0;
    const pcl_i64 tile = blockIdx.x;
    const pcl_u32 qbase = (pcl_u32)wave * 64u * NE; // this wave's part of the hit queue
    int kept = 0;
    const int cnt_in = pcl_mixed_seg_count(a, tile, wave); // particles of this wave's segment: they sit at its front
    const pcl_i64 seg0 = tile * PCL_T + (pcl_i64)wave * 512;
    for (int trip = 0; trip < 4; ++trip) {
        if (a.inplace && trip * NE * 64 >= cnt_in) break; // (wave-uniform) the rest of the segment is empty
        const int row0 = wave * 8 + trip * NE;
        T Rr[3][NE], V[3][NE], L4[NE], Ev[NE], NM[NE];
        pcl_u32 wodd0[NE], wodd1[NE];
        pcl_u64 id[NE];
        bool in[NE], photon[NE], alive[NE];
        // the same predicates as wave masks (scalar registers): the votes below are taken on bare compares and combined with
        // these in the scalar unit -- a vote on a compound predicate costs a v_cndmask and a second v_cmp (pcl_ballot)
        pcl_u64 am[NE], pm[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const pcl_i64 i = tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane;
            in[e] = (trip * NE + e) * 64 + lane < cnt_in;
            const pcl_i64 is = in[e] ? i : 0;
            const pcl_i64 ti = pcl_tix(is, a.ts);
            Rr[0][e] = a.r0[ti];
            Rr[1][e] = a.r1[ti];
            Rr[2][e] = a.r2[ti];
            V[0][e] = a.v0[ti];
            V[1][e] = a.v1[ti];
            V[2][e] = a.v2[ti];
            L4[e] = (T)1;
            if constexpr (USE_E) L4[e] = a.lam4[ti];
            Ev[e] = a.E[ti];
            id[e] = (pcl_u64)(a.ids ? a.ids[is] : a.id_base + i);
            photon[e] = in[e] && (a.kind ? (a.kind[is] != 0) : true);
            alive[e] = in[e];
            am[e] = pcl_ballot(in[e]);
            pm[e] = a.kind ? (pcl_ballot(a.kind[is] != 0) & am[e]) : am[e];
            wodd0[e] = wodd1[e] = 0u;
            // |dr| = |v * dt| only changes when the photon scatters: kept here, recomputed with the new velocity
            NM[e] = pcl_step_norm<T>(R::mul(V[0][e], a.dt), R::mul(V[1][e], a.dt), R::mul(V[2][e], a.dt));
        }
        for (int ph = 0; ph < n_ph; ++ph) {
            if (!(am[0] | am[1])) break; // nobody of these rows is left: their rows stay 0
            const pcl_u32 st = a.step + (pcl_u32)ph;
            const bool is_del = a.phase_del[ph % a.P] != 0; // wave-uniform
            pcl_u32 kk0 = k0, kk1 = k1;                     // see pcl_multi_body_lds: keeps the round keys off the VGPR spills
            asm volatile("" : "+s"(kk0), "+s"(kk1));
            const bool new_block = (st & 1u) == 0u || ph == 0;
            T d[3][NE], rand[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
                d[0][e] = R::mul(V[0][e], a.dt);
                d[1][e] = R::mul(V[1][e], a.dt);
                d[2][e] = R::mul(V[2][e], a.dt);
                Rr[0][e] = R::add(Rr[0][e], d[0][e]);
                Rr[1][e] = R::add(Rr[1][e], d[1][e]);
                Rr[2][e] = R::add(Rr[2][e], d[2][e]);
            }
            // the rows' draws side by side in ONE basic block (a Philox chain is ten dependent rounds; the uniform select of the
            // block's half stays outside the chains: pcl_multi_body_lds)
            if (new_block && (st & 1u) == 0u) { // decision block of the launch pair (st, st | 1): computed once for both
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id[e], (pcl_u32)(id[e] >> 32), st >> 1, 0u, kk0, kk1);
                    rand[e] = R::uniform(w.x, w.y);
                    wodd0[e] = w.z;
                    wodd1[e] = w.w;
                }
            } else if (new_block) { // a launch that starts on an odd index: the second half of the block of (st - 1, st)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id[e], (pcl_u32)(id[e] >> 32), st >> 1, 0u, kk0, kk1);
                    rand[e] = R::uniform(w.z, w.w);
                    wodd0[e] = w.z;
                    wodd1[e] = w.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) rand[e] = R::uniform(wodd0[e], wodd1[e]);
            }
            pcl_u32 w_evt = 0; // hits (isotropic phase) or removals (delete phase) of this wave
            if (is_del) {
                // ScatterDeleteStep: flag = (A*n*norm >= rand), flagged photons leave the list      light.py:239-260
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const T pc = R::mul(a.An_del, NM[e]);
                    const bool gone = alive[e] && photon[e] && (pc >= rand[e]);
                    const pcl_u64 gm = pcl_ballot(pc >= rand[e]) & am[e] & pm[e];
                    w_evt += (pcl_u32)__popcll(gm);
                    alive[e] = alive[e] && !gone;
                    am[e] &= ~gm;
                }
            } else {
                // ScatterIsotropicStep: decision in place, the hits densely through the wave's queue   light.py:303-331
                bool hit[NE];
                pcl_u32 slot[NE];
                pcl_u32 wbase = qbase;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    T pc = pcl_pcoll_norm<T, false, VAR_N>(a.np, a.A, a.n, (T)0, a.c, NM[e], d[0][e], d[1][e], d[2][e], Rr[0][e],
                                                           Rr[1][e], Rr[2][e], Ev[e]);
                    if constexpr (USE_E) pc = R::mul(pc, L4[e]);
                    hit[e] = alive[e] && photon[e] && (pc >= rand[e]);
                    const pcl_u64 b = pcl_ballot(pc >= rand[e]) & am[e] & pm[e];
                    slot[e] = wbase + __builtin_amdgcn_mbcnt_hi((pcl_u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((pcl_u32)b, 0u));
                    wbase += (pcl_u32)__popcll(b);
                    if (hit[e]) s_q.id[slot[e]] = id[e];
                }
                w_evt = wbase - qbase;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // wave-private queue: ordering only, no barrier
                __builtin_amdgcn_wave_barrier();
                for (pcl_u32 j = qbase + (pcl_u32)lane; j < wbase; j += 64) {
                    T rtheta, rphi, o0, o1, o2;
                    pcl_draw_angles<T>(s_q.id[j], st, kk0, kk1, rtheta, rphi);
                    pcl_new_velocity<T, true>(a.c, rtheta, rphi, o0, o1, o2);
                    s_q.out[0][j] = o0;
                    s_q.out[1][j] = o1;
                    s_q.out[2][j] = o2;
                    s_q.out[3][j] = pcl_step_norm<T>(R::mul(o0, a.dt), R::mul(o1, a.dt), R::mul(o2, a.dt));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (ph == a.last_iso) { // the velocity before the LAST scatter phase is what dv = v - v_prev needs
                    int lane_here = lane; // (opaque: the slab indices are worked out HERE, not carried in registers through every phase)
                    asm volatile("" : "+v"(lane_here));
#pragma unroll
                    for (int e = 0; e < NE; ++e)
                        if (in[e]) {
                            const pcl_i64 ti = pcl_tix(tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane_here, a.ts);
                            a.vp0[ti] = V[0][e];
                            a.vp1[ti] = V[1][e];
                            a.vp2[ti] = V[2][e];
                        }
                }
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (hit[e]) {
                        V[0][e] = s_q.out[0][slot[e]];
                        V[1][e] = s_q.out[1][slot[e]];
                        V[2][e] = s_q.out[2][slot[e]];
                        NM[e] = s_q.out[3][slot[e]];
                    }
            }
            // this phase's measure row, on the particles alive after it            light.py:385-399, 414-431
            pcl_u32 w_n = 0, w_sx = 0, w_sy = 0, w_sz = 0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                w_n += (pcl_u32)__popcll(am[e]);
                w_sx += (pcl_u32)__popcll(pcl_ballot(V[0][e] > (T)0) & am[e]);
                w_sy += (pcl_u32)__popcll(pcl_ballot(V[1][e] > (T)0) & am[e]);
                w_sz += (pcl_u32)__popcll(pcl_ballot(V[2][e] > (T)0) & am[e]);
            }
            pcl_u32 *c = &s_cnt[nslots * ph];
            for (int p = 0; p < a.n_planes; ++p) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                pcl_u32 nc = 0;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const T x = pcl_pick<T>(ax, Rr[0][e], Rr[1][e], Rr[2][e]);
                    const T prev = R::sub(x, pcl_pick<T>(ax, d[0][e], d[1][e], d[2][e]));
                    nc += (pcl_u32)__popcll(((pcl_ballot(prev <= L) & pcl_ballot(L <= x)) | (pcl_ballot(prev >= L) & pcl_ballot(L >= x))) & am[e]);
                }
                if (lane0 && nc) atomicAdd(&c[5 + p], nc);
            }
            if (lane < 5) { // the row's five columns: ONE LDS atomic of five lanes, not five single-lane ones in five basic blocks
                const pcl_u32 val = lane == 0 ? w_n : (lane == 1 ? w_evt : (lane == 2 ? w_sx : (lane == 3 ? w_sy : w_sz)));
                if (val) atomicAdd(&c[lane], val);
            }
        }
        if (a.inplace) { // the trip's survivors to the front of the wave's segment, row after row, with everything they own
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                pcl_mixed_put<T, USE_E>(a, seg0, (trip * NE + e) * 64, (pcl_u32)kept, am[e], lane, Rr[0][e], Rr[1][e], Rr[2][e], V[0][e], V[1][e],
                                        V[2][e], L4[e], id[e]);
                kept += (int)__popcll(am[e]);
            }
            continue;
        }
        int lane_end = lane;
        asm volatile("" : "+v"(lane_end));
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (alive[e]) {
                const pcl_i64 ti = pcl_tix(tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane_end, a.ts);
                a.r0[ti] = Rr[0][e];
                a.r1[ti] = Rr[1][e];
                a.r2[ti] = Rr[2][e];
                a.v0[ti] = V[0][e];
                a.v1[ti] = V[1][e];
                a.v2[ti] = V[2][e];
            }
            if (a.has_delete) {
                const pcl_u64 m = am[e];
                if (lane0) a.masks[tile * 32 + row0 + e] = m;
                kept += (int)__popcll(m);
            }
        }
    }
    if (a.inplace) pcl_mixed_put_masks(a, tile, wave, lane, kept);
    if (a.has_delete && lane0) s_keep[wave] = kept;
    __syncthreads();
    if (a.has_delete && threadIdx.x == 0) a.tile_keep[tile] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    for (int k = threadIdx.x; k < nslots * n_ph; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&a.cnt[k], (pcl_u64)s_cnt[k]);
}

// ------------------------------------------------------------------------------------------------
// pcl_mixed_body with the velocities (and |v dt|) of a wave's particles in LDS "homes", as pcl_multi_body_lds keeps them:
// NE rows of 64 particles per trip instead of two, the dense pass writes the new velocity straight into its owner's home.
// What it buys is fuller dense passes: a wave's pass costs the same for 38 hits as for 64, and at a hit fraction of 0.3
// (BASELINE configs[4]: A = n = 1e-3, dt = 1e-3) 128 particles queue 38 hits a step -- 0.50 passes per 64 particles --
// where 192 queue 58 -- 0.38; from a hit fraction of ~0.33 up two rows per trip are the better form again (the host
// picks: step_mixed_t).  Same operations per particle in the same order, same launch indices: bit-identical to
// pcl_mixed_body (tests/test_gpu_mixed.py runs both).  Rows past the wave's eight (NE = 3: the ninth) are empty.
// ------------------------------------------------------------------------------------------------
template <int N> struct pcl_ic { static constexpr int value = N; }; // (an integer as a type: hipRTC has no <type_traits>)

template <typename T, int NE>
struct pcl_mixed_home {
    T c[4][NE][256];                 // v0, v1, v2, |v dt| of row e of thread t
    pcl_u64 qid[256 * NE + 256];           // the queued hits: photon id ...
    unsigned short owner[256 * NE + 256];  // ... and home, e * 64 + lane (a wave's part of the queue is its own); behind the queues a
                                           // slot per thread for the writes of the lanes that did not hit (no branch around the writes)
};

template <typename T, bool USE_E, int VAR_N, int NE>
__device__ __forceinline__ void pcl_mixed_body_lds(const pcl_mixed_args<T> &a) {
    typedef pcl_rt<T> R;
    constexpr int ROWS = 8; // rows of 64 particles a wave owns in its tile
    constexpr int TRIPS = (ROWS + NE - 1) / NE;
    __shared__ pcl_u32 s_cnt[(5 + PCL_MAXPL) * PCL_MULTI_MAX];
    __shared__ pcl_mixed_home<T, NE> s_h;
    __shared__ int s_keep[4];
    const int nslots = 5 + a.n_planes;
    const int n_ph = a.K * a.P;
    for (int k = threadIdx.x; k < nslots * n_ph; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lane0 = lane == 0;
    const pcl_i64 tile = blockIdx.x;
    const pcl_u32 qbase = (pcl_u32)wave * 64u * NE; // this wave's part of the hit queue
    int kept = 0;
    const int cnt_in = pcl_mixed_seg_count(a, tile, wave); // particles of this wave's segment: they sit at its front
    const pcl_i64 seg0 = tile * PCL_T + (pcl_i64)wave * 512;
    // A trip takes up to NE rows of the wave's segment; its code exists once per NUMBER of rows (NA = 1 .. NE): a wave's eight rows
    // are trips of 3 + 3 + 2, and a segment that has lost photons ends in a trip of one or two rows -- with NE-row code those trips
    // issued every instruction for rows nobody is in (a ninth of a full tile's work, a fifth of a half-empty one's).
    auto do_trip = [&](auto na_tag, const int trip) {
        constexpr int NA = decltype(na_tag)::value;
        const int row0 = wave * ROWS + trip * NE;
        T Rr[3][NA], L4[NA], Ev[NA];
        pcl_u32 wodd0[NA], wodd1[NA];
        pcl_u64 id[NA];
        bool in[NA];
        pcl_u64 am[NA], pm[NA]; // alive / photon, as wave masks: the votes are taken on bare compares (pcl_ballot)
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const pcl_i64 i = tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane;
            in[e] = trip * NE + e < ROWS && (trip * NE + e) * 64 + lane < cnt_in;
            const pcl_i64 is = in[e] ? i : 0;
            const pcl_i64 ti = pcl_tix(is, a.ts);
            Rr[0][e] = a.r0[ti];
            Rr[1][e] = a.r1[ti];
            Rr[2][e] = a.r2[ti];
            const T v0 = a.v0[ti], v1 = a.v1[ti], v2 = a.v2[ti];
            L4[e] = (T)1;
            if constexpr (USE_E) L4[e] = a.lam4[ti];
            Ev[e] = a.E[ti];
            id[e] = (pcl_u64)(a.ids ? a.ids[is] : a.id_base + i);
            am[e] = pcl_ballot(in[e]);
            pm[e] = a.kind ? (pcl_ballot(a.kind[is] != 0) & am[e]) : am[e];
            wodd0[e] = wodd1[e] = 0u;
            s_h.c[0][e][tid] = v0;
            s_h.c[1][e][tid] = v1;
            s_h.c[2][e][tid] = v2;
            // |dr| = |v * dt| only changes when the photon scatters: kept beside v, recomputed with the new velocity
            s_h.c[3][e][tid] = pcl_step_norm<T>(R::mul(v0, a.dt), R::mul(v1, a.dt), R::mul(v2, a.dt));
        }
        for (int ph = 0; ph < n_ph; ++ph) {
            pcl_u64 any = 0;
#pragma unroll
            for (int e = 0; e < NA; ++e) any |= am[e];
            if (!any) break; // nobody of these rows is left: their rows stay 0
            const pcl_u32 st = a.step + (pcl_u32)ph;
            const bool is_del = a.phase_del[ph % a.P] != 0; // wave-uniform
            pcl_u32 kk0 = k0, kk1 = k1;                     // see pcl_multi_body_lds: keeps the round keys off the VGPR spills
            asm volatile("" : "+s"(kk0), "+s"(kk1));
            const bool new_block = (st & 1u) == 0u || ph == 0;
            T d[3][NA], rand[NA], nm[NA];
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
                d[0][e] = R::mul(s_h.c[0][e][tid], a.dt);
                d[1][e] = R::mul(s_h.c[1][e][tid], a.dt);
                d[2][e] = R::mul(s_h.c[2][e][tid], a.dt);
                nm[e] = s_h.c[3][e][tid];
                Rr[0][e] = R::add(Rr[0][e], d[0][e]);
                Rr[1][e] = R::add(Rr[1][e], d[1][e]);
                Rr[2][e] = R::add(Rr[2][e], d[2][e]);
            }
            // the rows' draws side by side in ONE basic block (a Philox chain is ten dependent rounds; the uniform select of the
            // block's half stays outside the chains: pcl_multi_body_lds)
            if (new_block && (st & 1u) == 0u) { // decision block of the launch pair (st, st | 1): computed once for both
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id[e], (pcl_u32)(id[e] >> 32), st >> 1, 0u, kk0, kk1);
                    rand[e] = R::uniform(w.x, w.y);
                    wodd0[e] = w.z;
                    wodd1[e] = w.w;
                }
            } else if (new_block) { // a launch that starts on an odd index: the second half of the block of (st - 1, st)
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id[e], (pcl_u32)(id[e] >> 32), st >> 1, 0u, kk0, kk1);
                    rand[e] = R::uniform(w.z, w.w);
                    wodd0[e] = w.z;
                    wodd1[e] = w.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NA; ++e) rand[e] = R::uniform(wodd0[e], wodd1[e]);
            }
            pcl_u32 w_evt = 0; // hits (isotropic phase) or removals (delete phase) of this wave
            pcl_u32 wbase = qbase;
            if (is_del) {
                // ScatterDeleteStep: flag = (A*n*norm >= rand), flagged photons leave the list      light.py:239-260
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    const T pc = R::mul(a.An_del, nm[e]);
                    const pcl_u64 gm = pcl_ballot(pc >= rand[e]) & am[e] & pm[e];
                    w_evt += (pcl_u32)__popcll(gm);
                    am[e] &= ~gm;
                }
            } else {
                // ScatterIsotropicStep: decision in place, the hits densely through the wave's queue   light.py:303-331
                wbase = qbase;
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    T pc = pcl_pcoll_norm<T, false, VAR_N>(a.np, a.A, a.n, (T)0, a.c, nm[e], d[0][e], d[1][e], d[2][e], Rr[0][e],
                                                           Rr[1][e], Rr[2][e], Ev[e]);
                    if constexpr (USE_E) pc = R::mul(pc, L4[e]);
                    const pcl_u64 b = pcl_ballot(pc >= rand[e]) & am[e] & pm[e];
                    const pcl_u32 slot = ((b >> lane) & 1ull) ? wbase + __builtin_amdgcn_mbcnt_hi((pcl_u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((pcl_u32)b, 0u))
                                                              : (pcl_u32)(256 * NE + tid);
                    s_h.qid[slot] = id[e];
                    s_h.owner[slot] = (unsigned short)(e * 64 + lane);
                    wbase += (pcl_u32)__popcll(b);
                }
                w_evt = wbase - qbase;
                if (ph == a.last_iso) { // the velocity before the LAST scatter phase is what dv = v - v_prev needs
                    int lane_here = lane; // (opaque: the slab indices are worked out HERE, not carried in registers through every phase)
                    asm volatile("" : "+v"(lane_here));
#pragma unroll
                    for (int e = 0; e < NA; ++e)
                        if (in[e]) {
                            const pcl_i64 ti = pcl_tix(tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane_here, a.ts);
                            a.vp0[ti] = s_h.c[0][e][tid];
                            a.vp1[ti] = s_h.c[1][e][tid];
                            a.vp2[ti] = s_h.c[2][e][tid];
                        }
                }
            }
            // plane crossings of this phase's move, on the particles alive after the phase (an isotropic phase removes nobody): counted
            // HERE, so that dr = v * dt need not live through the dense pass, where the registers are needed                light.py:385-399
            pcl_u32 *c = &s_cnt[nslots * ph];
            for (int p = 0; p < a.n_planes; ++p) {
                const int ax = a.plane_ax[p];
                const T L = a.plane_L[p];
                pcl_u32 nc = 0;
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    const T x = pcl_pick<T>(ax, Rr[0][e], Rr[1][e], Rr[2][e]);
                    const T prev = R::sub(x, pcl_pick<T>(ax, d[0][e], d[1][e], d[2][e]));
                    nc += (pcl_u32)__popcll(((pcl_ballot(prev <= L) & pcl_ballot(L <= x)) | (pcl_ballot(prev >= L) & pcl_ballot(L >= x))) & am[e]);
                }
                if (lane0 && nc) atomicAdd(&c[5 + p], nc);
            }
            if (!is_del) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // wave-private queue and homes: ordering only, no barrier
                __builtin_amdgcn_wave_barrier();
                for (pcl_u32 j = qbase + (pcl_u32)lane; j < wbase; j += 64) {
                    const pcl_u32 o = s_h.owner[j];
                    const pcl_u32 oe = o >> 6, ot = (pcl_u32)wave * 64u + (o & 63u);
                    T rtheta, rphi, o0, o1, o2;
                    pcl_draw_angles<T>(s_h.qid[j], st, kk0, kk1, rtheta, rphi);
                    pcl_new_velocity<T, true>(a.c, rtheta, rphi, o0, o1, o2);
                    s_h.c[0][oe][ot] = o0;
                    s_h.c[1][oe][ot] = o1;
                    s_h.c[2][oe][ot] = o2;
                    s_h.c[3][oe][ot] = pcl_step_norm<T>(R::mul(o0, a.dt), R::mul(o1, a.dt), R::mul(o2, a.dt));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            // this phase's measure row, on the particles alive after it            light.py:414-431
            pcl_u32 w_n = 0, w_sx = 0, w_sy = 0, w_sz = 0;
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                w_n += (pcl_u32)__popcll(am[e]);
                w_sx += (pcl_u32)__popcll(pcl_ballot(s_h.c[0][e][tid] > (T)0) & am[e]);
                w_sy += (pcl_u32)__popcll(pcl_ballot(s_h.c[1][e][tid] > (T)0) & am[e]);
                w_sz += (pcl_u32)__popcll(pcl_ballot(s_h.c[2][e][tid] > (T)0) & am[e]);
            }
            if (lane < 5) { // the row's five columns: ONE LDS atomic of five lanes, not five single-lane ones in five basic blocks
                const pcl_u32 val = lane == 0 ? w_n : (lane == 1 ? w_evt : (lane == 2 ? w_sx : (lane == 3 ? w_sy : w_sz)));
                if (val) atomicAdd(&c[lane], val);
            }
        }
        int lane_end = lane;
        asm volatile("" : "+v"(lane_end));
        pcl_i64 tile_end = tile; // (opaque as well: the address of the rows' mask words is not carried -- or spilled -- through the phases)
        asm volatile("" : "+s"(tile_end));
        if (a.inplace) { // the trip's survivors to the front of the wave's segment, row after row, with everything they own
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                if (trip * NE + e >= ROWS) continue;
                pcl_mixed_put<T, USE_E>(a, seg0, (trip * NE + e) * 64, (pcl_u32)kept, am[e], lane_end, Rr[0][e], Rr[1][e], Rr[2][e], s_h.c[0][e][tid],
                                        s_h.c[1][e][tid], s_h.c[2][e][tid], L4[e], id[e]);
                kept += (int)__popcll(am[e]);
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            if ((am[e] >> lane) & 1ull) { // (alive lanes are in range)
                const pcl_i64 ti = pcl_tix(tile * PCL_T + (pcl_i64)(row0 + e) * 64 + lane_end, a.ts);
                a.r0[ti] = Rr[0][e];
                a.r1[ti] = Rr[1][e];
                a.r2[ti] = Rr[2][e];
                a.v0[ti] = s_h.c[0][e][tid];
                a.v1[ti] = s_h.c[1][e][tid];
                a.v2[ti] = s_h.c[2][e][tid];
            }
            if (a.has_delete && trip * NE + e < ROWS) {
                const pcl_u64 m = am[e];
                if (lane0) a.masks[tile_end * 32 + row0 + e] = m;
                kept += (int)__popcll(m);
            }
        }
    };
    const int rows_seg = (cnt_in + 63) >> 6; // rows of the segment that hold anybody (they sit at its front)
    for (int trip = 0; trip < TRIPS; ++trip) {
        int rows = rows_seg - trip * NE; // (wave-uniform)
        rows = rows > NE ? NE : rows;
        if (rows <= 0 && a.inplace) break; // the rest of the segment is empty
        // (constant-n kernels only: the variable-n forms sit at their register bound -- 125-127 of 128 VGPRs -- and a second copy of the
        //  trip pushed them into scratch; one shorter form, NE - 1 rows: a third copy for single rows cost registers, not time)
        constexpr bool SPLIT = VAR_N == 0 && NE > 1;
        if (rows >= NE || (!SPLIT && rows > 0)) {
            do_trip(pcl_ic<NE>{}, trip);
        } else if (rows > 0) {
            if constexpr (SPLIT) do_trip(pcl_ic<(NE > 1 ? NE - 1 : 1)>{}, trip);
        }
        if (!a.inplace && a.has_delete && lane0) // rows of the trip nobody is in (a dense store's last tile): their keep-masks are empty
            for (int e = (rows >= NE || (!SPLIT && rows > 0)) ? NE : (rows > 0 ? NE - 1 : 0); e < NE; ++e)
                if (trip * NE + e < ROWS) a.masks[tile * 32 + wave * ROWS + trip * NE + e] = 0;
    }
    if (a.inplace) pcl_mixed_put_masks(a, tile, wave, lane, kept);
    if (a.has_delete && lane0) s_keep[wave] = kept;
    __syncthreads();
    if (a.has_delete && threadIdx.x == 0) a.tile_keep[tile] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    for (int k = threadIdx.x; k < nslots * n_ph; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&a.cnt[k], (pcl_u64)s_cnt[k]);
}

// ------------------------------------------------------------------------------------------------
// TracePathMeasureStep for a tracked subset (physicl/light.py:447-458), worked out AHEAD of the K-pass launch that will
// move the store.  A photon's history over the next K passes is a pure function of its own state and of (seed, launch
// index, id): photons do not interact and the random stream is keyed by the id.  So the positions the tracked photons
// will have after every pass need not be written by the K-step kernels (which are bound by VALU issue and keep no
// register to spare): one thread per TRACKED photon reads its state from the store as it stands, runs the same per-photon
// operations the K-step kernels will run -- Newton (newton.py:15-16), the isotropic decision and re-direction
// (light.py:303-315) and/or the delete decision (light.py:239-249), launch index step + pass * P + phase -- and writes
// one row per pass; the store is not touched.  Same operations on the same operands in the same order as
// pcl_fast_body / pcl_multi_body_lds / pcl_mixed_body: the rows are bit-identical to what a download after each pass
// would show (tests/test_gpu_trace.py checks exactly that, and the last row against the store after the launch).
//   out[(pass * n_want + j) * 4 + {0, 1, 2}] = r of tracked photon j when the trace step of that pass runs (behind phase
//   ``record_phase``), NaN when it is not in the store then (removed, or never there);  [.. + 3] = 1 if its dv is not
//   the zero vector at that point (TracePathMeasureStep(trace_dv=True), light.py:456), else 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct pcl_trace_args {
    const T *r0, *r1, *r2;
    const T *v0, *v1, *v2;
    const T *p0, *p1, *p2;     // the vprev rows: dv_mode 1 = the photons' dv is implicit, v - vprev (plain Objects keep real dv rows)
    const T *q0, *q1, *q2;     // the dv rows: dv_mode 2 = everybody's dv is there; dv_mode 0 = every dv is known to be the zero vector
    const T *lam4;             // pow((h*c)/E, -4) per photon            (USE_E)
    const T *E;
    const pcl_i64 *ids;        // explicit ids of the store's slots, NULL: id = id_base + slot
    const unsigned char *kind; // NULL: every particle is a photon
    const pcl_u64 *alive;      // a store behind an alive mask: one bit per slot; NULL: every slot holds a particle
    const pcl_i64 *want;       // [n_want] the tracked ids, ascending
    const pcl_i64 *slot;       // [n_want] where each of them is in the store (-1: nowhere); NULL: slot = id - id_base
    double *out;               // [K][n_want][4]
    pcl_i64 id_base, N, ts;    // N = the store's extent in slots
    int n_want;
    T dt, A, n, c;             // isotropic phase: kernel constants after the reference's swap (light.py:287)
    T An_del;                  // delete phase: A * n rounded once (light.py:243)
    pcl_u64 seed;
    pcl_u32 step;              // launch index of the first phase
    int K, P;                  // passes, phases per pass
    int phase_del[PCL_MIXED_MAXPH];
    int record_phase;          // the trace step runs behind this phase of every pass
    int dv_mode;
    int n_pend;                // Newton moves r has not seen yet (stores behind an alive mask, pcl_fast_args)
    T pend_dt[PCL_PEND_MAX];
    int pend_rep[PCL_PEND_MAX];
    pcl_nprof<T> np;           // ahead-of-time VAR_N kernels only
};

template <typename T, bool USE_E, int VAR_N>
__device__ __forceinline__ void pcl_trace_body(const pcl_trace_args<T> &a) {
    typedef pcl_rt<T> R;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= a.n_want) return;
    const double nan = __builtin_nan("");
    const pcl_u64 id = (pcl_u64)a.want[j];
    const pcl_i64 s = a.slot ? a.slot[j] : (pcl_i64)id - a.id_base;
    bool here = s >= 0 && s < a.N;
    if (here && a.alive) here = (a.alive[s >> 6] >> (s & 63)) & 1ull;
    int k = 0;
    if (here) {
        const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
        const pcl_i64 ti = pcl_tix(s, a.ts);
        T r[3] = {a.r0[ti], a.r1[ti], a.r2[ti]}, v[3] = {a.v0[ti], a.v1[ti], a.v2[ti]};
        const T Ev = a.E[ti];
        T L4 = (T)1;
        if constexpr (USE_E) L4 = a.lam4[ti];
        const bool photon = a.kind ? (a.kind[s] != 0) : true;
        bool moved = false; // is dv anything but the zero vector?
        if (a.dv_mode == 1 && photon) moved = R::sub(v[0], a.p0[ti]) != (T)0 || R::sub(v[1], a.p1[ti]) != (T)0 || R::sub(v[2], a.p2[ti]) != (T)0;
        else if (a.dv_mode != 0) moved = a.q0[ti] != (T)0 || a.q1[ti] != (T)0 || a.q2[ti] != (T)0;
        for (int p = 0; p < a.n_pend; ++p) { // moves of earlier delete bodies that r has not seen yet (pcl_fast_body)
            const T m0 = R::mul(v[0], a.pend_dt[p]), m1 = R::mul(v[1], a.pend_dt[p]), m2 = R::mul(v[2], a.pend_dt[p]);
            for (int w = 0; w < a.pend_rep[p]; ++w) {
                r[0] = R::add(r[0], m0);
                r[1] = R::add(r[1], m1);
                r[2] = R::add(r[2], m2);
            }
        }
        for (; k < a.K && here; ++k) {
            for (int ph = 0; ph < a.P; ++ph) {
                const pcl_u32 st = a.step + (pcl_u32)(k * a.P + ph);
                // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
                const T d0 = R::mul(v[0], a.dt), d1 = R::mul(v[1], a.dt), d2 = R::mul(v[2], a.dt);
                r[0] = R::add(r[0], d0);
                r[1] = R::add(r[1], d1);
                r[2] = R::add(r[2], d2);
                if (photon && here) {
                    const T rand = pcl_draw_rand<T>(id, st, k0, k1);
                    if (a.phase_del[ph]) {
                        // ScatterDeleteStep: flag = (A*n*norm >= rand), flagged photons leave the list      light.py:239-260
                        if (R::mul(a.An_del, pcl_step_norm<T>(d0, d1, d2)) >= rand) here = false;
                    } else {
                        // ScatterIsotropicStep                                                              light.py:303-331
                        T pc = pcl_pcoll<T, false, VAR_N>(a.np, a.A, a.n, (T)0, a.c, d0, d1, d2, r[0], r[1], r[2], Ev);
                        if constexpr (USE_E) pc = R::mul(pc, L4);
                        moved = false; // a miss leaves dv = 0                                               light.py:330-331
                        if (pc >= rand) {
                            T rtheta, rphi, n0, n1, n2;
                            pcl_draw_angles<T>(id, st, k0, k1, rtheta, rphi);
                            pcl_new_velocity<T, true>(a.c, rtheta, rphi, n0, n1, n2);
                            moved = R::sub(n0, v[0]) != (T)0 || R::sub(n1, v[1]) != (T)0 || R::sub(n2, v[2]) != (T)0;
                            v[0] = n0;
                            v[1] = n1;
                            v[2] = n2;
                        }
                    }
                }
                if (ph == a.record_phase) {
                    double *o = a.out + ((pcl_i64)k * a.n_want + j) * 4;
                    o[0] = here ? (double)r[0] : nan;
                    o[1] = here ? (double)r[1] : nan;
                    o[2] = here ? (double)r[2] : nan;
                    o[3] = here ? (moved ? 1.0 : 0.0) : nan;
                }
            }
        }
    }
    for (; k < a.K; ++k) { // not in the store (any more): "nan;nan;nan"                                      light.py:435
        double *o = a.out + ((pcl_i64)k * a.n_want + j) * 4;
        o[0] = o[1] = o[2] = o[3] = nan;
    }
}

#ifdef PCL_RTC
// hipRTC translation unit: one expression; both wavelength variants of every kernel, fp64 and fp32.
#define PCL_RTC_KERNEL(name, argtype, call) \
    extern "C" __global__ void __launch_bounds__(256) name(argtype a) { call(a); }
// PCL_RTC_DT (0 fp64, 1 fp32) and PCL_RTC_E (wavelength term) select the quarter of the kernels a (store, step) pair can
// launch; without them every kernel is emitted
#ifndef PCL_RTC_DT
#define PCL_RTC_ALL 1
#define PCL_RTC_DT 0
#define PCL_RTC_E 0
#else
#define PCL_RTC_ALL 0
#endif
#define PCL_RTC_WANT(d, e) (PCL_RTC_ALL || (PCL_RTC_DT == (d) && PCL_RTC_E == (e)))
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_sphere_e0, pcl_sphere_args, (pcl_sphere_body<false, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_sphere_e1, pcl_sphere_args, (pcl_sphere_body<true, true>))
#endif
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_scatter_e0, pcl_scatter_args<double>, (pcl_scatter_body<double, false, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_scatter_e1, pcl_scatter_args<double>, (pcl_scatter_body<double, true, true>))
#endif
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_fused_e0, pcl_fused_args<double>, (pcl_fused_body<double, false, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_fused_e1, pcl_fused_args<double>, (pcl_fused_body<double, true, true>))
#endif
#ifndef PCL_FAST_ATTR /* timing experiments: e.g. -DPCL_FAST_ATTR=__attribute__((amdgpu_waves_per_eu(5,5))) */
#define PCL_FAST_ATTR
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_FAST_ATTR pcl_rtc_fast_e0(pcl_fast_args<double> a) { pcl_fast_body<double, false, true, 2>(a); }
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_FAST_ATTR pcl_rtc_fast_e1(pcl_fast_args<double> a) { pcl_fast_body<double, true, true, 2>(a); }
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_scatter_f_e0, pcl_scatter_args<float>, (pcl_scatter_body<float, false, true>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_scatter_f_e1, pcl_scatter_args<float>, (pcl_scatter_body<float, true, true>))
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_fused_f_e0, pcl_fused_args<float>, (pcl_fused_body<float, false, true>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_fused_f_e1, pcl_fused_args<float>, (pcl_fused_body<float, true, true>))
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_fast_f_e0, pcl_fast_args<float>, (pcl_fast_body<float, false, true, 4>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_fast_f_e1, pcl_fast_args<float>, (pcl_fast_body<float, true, true, 4>))
#endif
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_fastg_e0, pcl_fast_args<double>, (pcl_fast_body<double, false, true, 2, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_fastg_e1, pcl_fast_args<double>, (pcl_fast_body<double, true, true, 2, true>))
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_fastg_f_e0, pcl_fast_args<float>, (pcl_fast_body<float, false, true, 4, true>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_fastg_f_e1, pcl_fast_args<float>, (pcl_fast_body<float, true, true, 4, true>))
#endif
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_mixed_e0, pcl_mixed_args<double>, (pcl_mixed_body<double, false, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_mixed_e1, pcl_mixed_args<double>, (pcl_mixed_body<double, true, true>))
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_mixed_f_e0, pcl_mixed_args<float>, (pcl_mixed_body<float, false, true>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_mixed_f_e1, pcl_mixed_args<float>, (pcl_mixed_body<float, true, true>))
#endif
// the tracked subset's positions over the passes of the next K-pass launch (pcl_trace_body)
#if PCL_RTC_WANT(0, 0)
PCL_RTC_KERNEL(pcl_rtc_trace_e0, pcl_trace_args<double>, (pcl_trace_body<double, false, true>))
#endif
#if PCL_RTC_WANT(0, 1)
PCL_RTC_KERNEL(pcl_rtc_trace_e1, pcl_trace_args<double>, (pcl_trace_body<double, true, true>))
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_trace_f_e0, pcl_trace_args<float>, (pcl_trace_body<float, false, true>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_trace_f_e1, pcl_trace_args<float>, (pcl_trace_body<float, true, true>))
#endif
// three rows of 64 particles per wave and trip, velocities in LDS (pcl_mixed_body_lds): what a loop with a variable_n_fn takes once
// a launch has shown its hit fraction to be below 0.33 (step_mixed_t)
#ifndef PCL_MIXED3_ATTR
#define PCL_MIXED3_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MIXED3_ATTR pcl_rtc_mixed3_e0(pcl_mixed_args<double> a) { pcl_mixed_body_lds<double, false, true, 3>(a); }
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MIXED3_ATTR pcl_rtc_mixed3_e1(pcl_mixed_args<double> a) { pcl_mixed_body_lds<double, true, true, 3>(a); }
#endif
#if PCL_RTC_WANT(1, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MIXED3_ATTR pcl_rtc_mixed3_f_e0(pcl_mixed_args<float> a) { pcl_mixed_body_lds<float, false, true, 3>(a); }
#endif
#if PCL_RTC_WANT(1, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MIXED3_ATTR pcl_rtc_mixed3_f_e1(pcl_mixed_args<float> a) { pcl_mixed_body_lds<float, true, true, 3>(a); }
#endif
#ifndef PCL_MULTI_ATTR /* timing experiments: e.g. -DPCL_MULTI_ATTR=__attribute__((amdgpu_waves_per_eu(5,5))) */
#define PCL_MULTI_ATTR
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI_ATTR pcl_rtc_multi_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 2, 1>(a);
}
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI_ATTR pcl_rtc_multi_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 2, 1>(a);
}
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI_ATTR pcl_rtc_multis_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 2, 1, true>(a); // with the saturation probe (pcl_n_expr_sat)
}
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI_ATTR pcl_rtc_multis_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 2, 1, true>(a);
}
#endif
#if PCL_RTC_WANT(1, 0)
PCL_RTC_KERNEL(pcl_rtc_multi_f_e0, pcl_multi_args<float>, (pcl_multi_body_lds<float, false, true, 4, 1>))
#endif
#if PCL_RTC_WANT(1, 1)
PCL_RTC_KERNEL(pcl_rtc_multi_f_e1, pcl_multi_args<float>, (pcl_multi_body_lds<float, true, true, 4, 1>))
#endif
// two VEC groups per lane and trip (256 photons per wave): the dense hit pass costs the same for 20 hits as for 64, so at
// LOW hit fractions a wave that owns twice the photons fills its passes better (+5 .. +13 % below ~25 % hits) -- at the
// price of the fourth wave per SIMD (-8 % above 27 %).  pcl_step_fused_multi picks per launch, by the hit fraction of the
// launch before ("K steps per pass").
#ifndef PCL_MULTI2_ATTR
#define PCL_MULTI2_ATTR __attribute__((amdgpu_waves_per_eu(4, 4))) /* four waves per SIMD: 128 VGPRs (the variant without the probe would take 159) */
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi2_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 2, 2>(a);
}
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi2_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 2, 2>(a);
}
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi2s_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 2, 2, true>(a); // 256 photons per wave with the saturation probe
}
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi2s_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 2, 2, true>(a);
}
#endif

// three photons per lane (8-byte accesses), 192 per wave: between hit fractions of ~0.25 and ~0.32 a wave of 256 photons
// queues 64 to 82 hits -- a full dense pass and a nearly empty one -- where 192 photons queue 48 to 62: one pass.
// pcl_step_fused_multi takes this form for a launch that starts in that band.
// 28 KB of LDS per workgroup and 89-90 VGPRs: FIVE waves per SIMD (the 256-photon forms' 37 KB allow four) -- same-box
// -3.3 % on the form's block of the driver's command, six no better (profiles/r05_ab_occupancy.log).
// (the variant with the saturation probe: its exp polynomial is off the hot path; the one without keeps 96-116 VGPRs: four waves)
#ifndef PCL_MULTI3_ATTR
#define PCL_MULTI3_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
#if PCL_RTC_WANT(0, 0)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi3_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 1, 3>(a);
}
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI3_ATTR pcl_rtc_multi3s_e0(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, false, true, 1, 3, true>(a);
}
#endif
#if PCL_RTC_WANT(0, 1)
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI2_ATTR pcl_rtc_multi3_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 1, 3>(a);
}
extern "C" __global__ void __launch_bounds__(256) PCL_MULTI3_ATTR pcl_rtc_multi3s_e1(pcl_multi_args<double> a) {
    pcl_multi_body_lds<double, true, true, 1, 3, true>(a);
}
#endif

#endif
#endif // PCL_DEVICE_H
