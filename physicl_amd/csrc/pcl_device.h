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
#ifdef PCL_ABLATE_PHILOX /* timing experiment only */
    const int kRounds = 1;
#else
    const int kRounds = 10;
#endif
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
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
#ifdef PCL_ABLATE_POW /* timing experiment only */
    const double x = __ddiv_rn(__dmul_rn(h, c), E), x2 = x * x;
    return 1.0 / (x2 * x2);
#else
    return pow(__ddiv_rn(__dmul_rn(h, c), E), -4.0);
#endif
}

// res0 = c * sin(rtheta) * cos(rphi); res1 = c * sin(rtheta) * sin(rphi); res2 = c * cos(rtheta)
//                                                     physicl/light.py:309-311
__device__ __forceinline__ void pcl_new_velocity(double c, double rtheta, double rphi, double &o0, double &o1,
                                                 double &o2) {
    double st, ct, sp, cp;
#ifdef PCL_ABLATE_TRIG /* timing experiment only */
    st = rtheta * 0.1; ct = 1.0 - st; sp = rphi * 0.2; cp = 1.0 - sp;
#else
    sincos(rtheta, &st, &ct);
    sincos(rphi, &sp, &cp);
#endif
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

// ------------------------------------------------------------------------------------------------
// Level 2: ONE pass for the whole loop body  Newton -> ScatterIsotropic -> measure counters
//   physicl/newton.py:15-16, physicl/light.py:303-315 + 325-331, physicl/light.py:385-399 + 424-426
// Same per-particle arithmetic, in the same order, as the three separate kernels -- so results are
// bit-identical to running them one after the other -- but r, v, E are read once and dr never
// travels back from HBM: 128 + 24h B per particle-step instead of 96 + (64 + 48h) + 24.
// Each lane owns two consecutive particles (16-byte loads/stores); counters are wave-ballot
// popcounts kept in scalar registers, LDS-staged per workgroup, one atomic per workgroup per counter.
// ------------------------------------------------------------------------------------------------
#define PCL_MAXPL 12

// Streaming accesses of the fused pass: every byte is touched once per step, so nothing is worth
// keeping in L2.  PCL_NT_LOADS / PCL_NT_STORES switch the nontemporal forms on.
typedef double pcl_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 pcl_ld2(const double *base, pcl_i64 p) {
#ifdef PCL_NT_LOADS
    const pcl_d2 t = __builtin_nontemporal_load(reinterpret_cast<const pcl_d2 *>(base) + p);
    return make_double2(t.x, t.y);
#else
    return reinterpret_cast<const double2 *>(base)[p];
#endif
}
__device__ __forceinline__ void pcl_st2(double *base, pcl_i64 p, const double2 &v) {
#ifdef PCL_NT_STORES
    pcl_d2 t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<pcl_d2 *>(base) + p);
#else
    reinterpret_cast<double2 *>(base)[p] = v;
#endif
}
struct pcl_fused_args {
    double *r0, *r1, *r2;       // Object.r   (read, written)
    const double *vi0, *vi1, *vi2; // Object.v as the step finds it
    double *vo0, *vo1, *vo2;    // Object.v as the step leaves it: eager = same arrays, written on a hit;
                                // lazy = the other half of the v double buffer, always written
    double *dr0, *dr1, *dr2;    // Object.dr  (written; not in lazy mode)
    double *dv0, *dv1, *dv2;    // Object.dv  (written for photons; not in lazy mode)
    const double *E;
    const double *rtheta, *rphi, *rand; // PCL_RNG_IN
    const pcl_i64 *ids;
    const unsigned char *kind;
    pcl_u64 *cnt;               // [0] hits, [1..3] sign counts, [4..] plane crossings
    pcl_i64 id_base;
    pcl_i64 N;
    double dt, A, n, c, h;
    pcl_u64 seed;
    pcl_u32 step;
    int rng_mode;
    int lazy;                   // 1: dr/dv stay implicit (dr = v_in*dt, dv = v_out - v_in), see pcl_step_fused
    int do_scatter;             // 0: Newton (+ counters) only
    int n_planes;               // -1: no counters at all
    double plane_L[PCL_MAXPL];
    int plane_ax[PCL_MAXPL];
};

#ifdef PCL_N_EXPR
// the expression with its array names bound to this particle's in-register values
__device__ __forceinline__ double pcl_n_expr_val(double r0v, double r1v, double r2v, double d0v, double d1v,
                                                 double d2v, double Ev) {
    const double r0[1] = {r0v}, r1[1] = {r1v}, r2[1] = {r2v}, d0[1] = {d0v}, d1[1] = {d1v}, d2[1] = {d2v},
                 E[1] = {Ev};
    const int gid = 0;
    (void)r0; (void)r1; (void)r2; (void)d0; (void)d1; (void)d2; (void)E; (void)gid;
    return (double)(PCL_N_EXPR);
}
#endif

__device__ __forceinline__ double pcl_pick(int ax, double a0, double a1, double a2) {
    return ax == 0 ? a0 : (ax == 1 ? a1 : a2);
}
__device__ __forceinline__ double pcl_sel(const double2 &q, int e) { return e ? q.y : q.x; }
__device__ __forceinline__ void pcl_put(double2 &q, int e, double x) {
    if (e) q.y = x; else q.x = x;
}

template <bool USE_E, bool VAR_N>
__device__ __forceinline__ void pcl_fused_body(const pcl_fused_args &a) {
    __shared__ pcl_u32 s_cnt[4 + PCL_MAXPL];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 4 + PCL_MAXPL) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // wave-uniform tallies (live in SGPRs): hits, sign x/y/z, plane crossings
    pcl_u32 w_hits = 0, w_sx = 0, w_sy = 0, w_sz = 0;
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const bool counters = a.n_planes >= 0;
    const pcl_i64 npair = (a.N + 1) >> 1;
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
    // every lane of a workgroup makes the same number of trips, so ballots always see whole waves
    for (pcl_i64 base = (pcl_i64)blockIdx.x * blockDim.x; base < npair; base += stride) {
        const pcl_i64 p = base + threadIdx.x;
        const bool live_pair = p < npair;
        const pcl_i64 pp = live_pair ? p : 0; // idle lanes re-read pair 0 and store nothing
        const bool live[2] = {live_pair && 2 * p < a.N, live_pair && 2 * p + 1 < a.N};
        double2 R[3], V[3], D[3], DV[3];
        R[0] = pcl_ld2(a.r0, pp);
        R[1] = pcl_ld2(a.r1, pp);
        R[2] = pcl_ld2(a.r2, pp);
        V[0] = pcl_ld2(a.vi0, pp);
        V[1] = pcl_ld2(a.vi1, pp);
        V[2] = pcl_ld2(a.vi2, pp);
        double2 Ev = make_double2(1.0, 1.0);
        if (a.do_scatter) Ev = pcl_ld2(a.E, pp);
        // ---- NewtonianKinematicsStep: dr = v*dt (rounded), r = r + dr              newton.py:15-16
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            D[k].x = __dmul_rn(V[k].x, a.dt);
            D[k].y = __dmul_rn(V[k].y, a.dt);
            R[k].x = __dadd_rn(R[k].x, D[k].x);
            R[k].y = __dadd_rn(R[k].y, D[k].y);
        }
        if (live_pair) {
            if (!a.lazy) {
                pcl_st2(a.dr0, p, D[0]);
                pcl_st2(a.dr1, p, D[1]);
                pcl_st2(a.dr2, p, D[2]);
            }
            pcl_st2(a.r0, p, R[0]);
            pcl_st2(a.r1, p, R[1]);
            pcl_st2(a.r2, p, R[2]);
        }
        // ---- ScatterIsotropicStep on each of the lane's two particles              light.py:303-331
        if (a.do_scatter) {
            bool hit[2], photon[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const pcl_i64 i = 2 * p + e;
                const double d0 = pcl_sel(D[0], e), d1 = pcl_sel(D[1], e), d2 = pcl_sel(D[2], e);
                photon[e] = live[e] && (a.kind ? (a.kind[i] != 0) : true);
                const double norm = pcl_step_norm(d0, d1, d2);
                double pc;
                if constexpr (VAR_N) {
#ifdef PCL_N_EXPR
                    pc = __dmul_rn(__dmul_rn(a.A, pcl_n_expr_val(pcl_sel(R[0], e), pcl_sel(R[1], e), pcl_sel(R[2], e),
                                                                  d0, d1, d2, pcl_sel(Ev, e))), norm);
#else
                    pc = 0.0;
#endif
                } else {
                    pc = __dmul_rn(__dmul_rn(a.A, a.n), norm);
                }
                if constexpr (USE_E) pc = __dmul_rn(pc, pcl_wavelength_term(a.h, a.c, pcl_sel(Ev, e)));
                double rand = 0.0, rtheta = 0.0, rphi = 0.0;
                pcl_u32 c0 = 0, c1 = 0;
                if (photon[e]) {
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
                }
                hit[e] = photon[e] && (pc >= rand);
                double n0 = 0.0, n1 = 0.0, n2 = 0.0;
                if (hit[e]) {
                    if (a.rng_mode == PCL_RNG_PHX) {
                        const pcl_u32x4 w = pcl_philox4x32_10(c0, c1, a.step, 1u, k0, k1);
                        rphi = __dmul_rn(pcl_u53(w.x, w.y), PCL_PI);
                    } else {
                        rtheta = a.rtheta[i];
                        rphi = a.rphi[i];
                    }
                    pcl_new_velocity(a.c, rtheta, rphi, n0, n1, n2);
                }
                // hit: dv = v' - v_old, v = v' ; miss: dv = 0                       light.py:327-331
                pcl_put(DV[0], e, hit[e] ? __dsub_rn(n0, pcl_sel(V[0], e)) : 0.0);
                pcl_put(DV[1], e, hit[e] ? __dsub_rn(n1, pcl_sel(V[1], e)) : 0.0);
                pcl_put(DV[2], e, hit[e] ? __dsub_rn(n2, pcl_sel(V[2], e)) : 0.0);
                if (hit[e]) {
                    pcl_put(V[0], e, n0);
                    pcl_put(V[1], e, n1);
                    pcl_put(V[2], e, n2);
                }
            }
            if (a.lazy) {
                // v double buffer: every particle's (possibly new) velocity goes to the other buffer, whole
                // 16-byte stores; dr and dv are not written -- they stay derivable from (v_in, v_out, dt)
                if (live_pair) {
                    pcl_st2(a.vo0, p, V[0]);
                    pcl_st2(a.vo1, p, V[1]);
                    pcl_st2(a.vo2, p, V[2]);
                }
            } else {
                // photons always get dv written; plain Objects keep theirs (light.py:283 skips them)
                if (photon[0] && photon[1]) {
                    pcl_st2(a.dv0, p, DV[0]);
                    pcl_st2(a.dv1, p, DV[1]);
                    pcl_st2(a.dv2, p, DV[2]);
                } else {
                    if (photon[0]) { a.dv0[2 * p] = DV[0].x; a.dv1[2 * p] = DV[1].x; a.dv2[2 * p] = DV[2].x; }
                    if (photon[1]) { a.dv0[2 * p + 1] = DV[0].y; a.dv1[2 * p + 1] = DV[1].y; a.dv2[2 * p + 1] = DV[2].y; }
                }
                if (hit[0] && hit[1]) {
                    pcl_st2(a.vo0, p, V[0]);
                    pcl_st2(a.vo1, p, V[1]);
                    pcl_st2(a.vo2, p, V[2]);
                } else {
                    if (hit[0]) { a.vo0[2 * p] = V[0].x; a.vo1[2 * p] = V[1].x; a.vo2[2 * p] = V[2].x; }
                    if (hit[1]) { a.vo0[2 * p + 1] = V[0].y; a.vo1[2 * p + 1] = V[1].y; a.vo2[2 * p + 1] = V[2].y; }
                }
            }
            w_hits += (pcl_u32)__popcll(__ballot(hit[0])) + (pcl_u32)__popcll(__ballot(hit[1]));
        }
        // ---- measure counters on the post-step state                    light.py:424-426, 385-399
        if (counters) {
            w_sx += (pcl_u32)__popcll(__ballot(live[0] && V[0].x > 0.0)) + (pcl_u32)__popcll(__ballot(live[1] && V[0].y > 0.0));
            w_sy += (pcl_u32)__popcll(__ballot(live[0] && V[1].x > 0.0)) + (pcl_u32)__popcll(__ballot(live[1] && V[1].y > 0.0));
            w_sz += (pcl_u32)__popcll(__ballot(live[0] && V[2].x > 0.0)) + (pcl_u32)__popcll(__ballot(live[1] && V[2].y > 0.0));
            for (int q = 0; q < a.n_planes; ++q) { // rolled: planes are rare, keep their state out of registers
                const int ax = a.plane_ax[q];
                const double L = a.plane_L[q];
                // by-value picks: selecting between the double2 lvalues would pin R/D in scratch
                const double Xx = pcl_pick(ax, R[0].x, R[1].x, R[2].x), Xy = pcl_pick(ax, R[0].y, R[1].y, R[2].y);
                const double px = __dsub_rn(Xx, pcl_pick(ax, D[0].x, D[1].x, D[2].x));
                const double py = __dsub_rn(Xy, pcl_pick(ax, D[0].y, D[1].y, D[2].y));
                const bool cx = live[0] && ((px <= L && L <= Xx) || (px >= L && L >= Xx));
                const bool cy = live[1] && ((py <= L && L <= Xy) || (py >= L && L >= Xy));
                const pcl_u32 nq = (pcl_u32)__popcll(__ballot(cx)) + (pcl_u32)__popcll(__ballot(cy));
                if (lane == 0 && nq) atomicAdd(&s_cnt[4 + q], nq);
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
// re-evaluated every step.  104 B per particle-step.
// ------------------------------------------------------------------------------------------------
struct pcl_fast_args {
    double *r0, *r1, *r2;          // read + written
    const double *vi0, *vi1, *vi2; // v before the step
    double *vo0, *vo1, *vo2;       // v after the step (other half of the double buffer)
    const double *lam4;            // pow((h*c)/E, -4) per photon            (USE_E)
    const double *E;               // only dereferenced if the expression names E[gid]
    pcl_u64 *cnt;                  // [0] hits, [1..3] sign counts
    pcl_i64 id_base, N;
    double dt, A, n, c;
    pcl_u64 seed;
    pcl_u32 step;
};

template <int VEC> struct pcl_vec;
template <> struct pcl_vec<1> {
    __device__ static __forceinline__ void ld(const double *b, pcl_i64 q, double (&o)[1]) { o[0] = b[q]; }
    __device__ static __forceinline__ void st(double *b, pcl_i64 q, const double (&o)[1]) { b[q] = o[0]; }
};
template <> struct pcl_vec<2> {
    __device__ static __forceinline__ void ld(const double *b, pcl_i64 q, double (&o)[2]) {
        const double2 t = reinterpret_cast<const double2 *>(b)[q];
        o[0] = t.x;
        o[1] = t.y;
    }
    __device__ static __forceinline__ void st(double *b, pcl_i64 q, const double (&o)[2]) {
        reinterpret_cast<double2 *>(b)[q] = make_double2(o[0], o[1]);
    }
};

template <int VEC>
struct pcl_fast_tile {
    double R[3][VEC], V[3][VEC], L4[VEC];
};

template <bool USE_E, int VEC>
__device__ __forceinline__ void pcl_fast_load(const pcl_fast_args &a, pcl_i64 q, pcl_fast_tile<VEC> &t) {
    pcl_vec<VEC>::ld(a.r0, q, t.R[0]);
    pcl_vec<VEC>::ld(a.r1, q, t.R[1]);
    pcl_vec<VEC>::ld(a.r2, q, t.R[2]);
    pcl_vec<VEC>::ld(a.vi0, q, t.V[0]);
    pcl_vec<VEC>::ld(a.vi1, q, t.V[1]);
    pcl_vec<VEC>::ld(a.vi2, q, t.V[2]);
    if constexpr (USE_E) pcl_vec<VEC>::ld(a.lam4, q, t.L4);
}

template <bool USE_E, bool VAR_N, int VEC>
__device__ __forceinline__ void pcl_fast_body(const pcl_fast_args &a) {
    __shared__ pcl_u32 s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    pcl_u32 w_hits = 0, w_sx = 0, w_sy = 0, w_sz = 0; // wave-uniform tallies (SGPRs)
    const pcl_u32 k0 = (pcl_u32)a.seed, k1 = (pcl_u32)(a.seed >> 32);
    const pcl_i64 nq = (a.N + VEC - 1) / VEC; // VEC-wide groups
    const pcl_i64 stride = (pcl_i64)gridDim.x * blockDim.x;
    pcl_i64 base = (pcl_i64)blockIdx.x * blockDim.x;
    // register double buffer: the next trip's loads are in flight while this trip computes
    pcl_fast_tile<VEC> cur;
    if (base < nq) pcl_fast_load<USE_E, VEC>(a, base + threadIdx.x < nq ? base + threadIdx.x : 0, cur);
    for (; base < nq; base += stride) {
        const pcl_i64 q = base + threadIdx.x;
        const bool live_q = q < nq;
#ifndef PCL_FAST_NOPIPE
        pcl_fast_tile<VEC> nxt;
        const pcl_i64 nb = base + stride;
        if (nb < nq) pcl_fast_load<USE_E, VEC>(a, nb + threadIdx.x < nq ? nb + threadIdx.x : 0, nxt);
#endif
        bool hit[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const pcl_i64 i = q * VEC + e;
            const bool live = live_q && i < a.N;
            // Newton: dr = v*dt (rounded), r = r + dr                                  newton.py:15-16
            const double d0 = __dmul_rn(cur.V[0][e], a.dt), d1 = __dmul_rn(cur.V[1][e], a.dt),
                         d2 = __dmul_rn(cur.V[2][e], a.dt);
            cur.R[0][e] = __dadd_rn(cur.R[0][e], d0);
            cur.R[1][e] = __dadd_rn(cur.R[1][e], d1);
            cur.R[2][e] = __dadd_rn(cur.R[2][e], d2);
            // scatter                                                                light.py:303-315
            const double norm = pcl_step_norm(d0, d1, d2);
            double pc;
            if constexpr (VAR_N) {
#ifdef PCL_N_EXPR
                pc = __dmul_rn(__dmul_rn(a.A, pcl_n_expr_val(cur.R[0][e], cur.R[1][e], cur.R[2][e], d0, d1, d2,
                                                              a.E[live ? i : 0])), norm);
#else
                pc = 0.0;
#endif
            } else {
                pc = __dmul_rn(__dmul_rn(a.A, a.n), norm);
            }
            if constexpr (USE_E) pc = __dmul_rn(pc, cur.L4[e]);
            const pcl_u64 id = (pcl_u64)(a.id_base + i);
            const pcl_u32x4 w = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), a.step, 0u, k0, k1);
            const double rand = pcl_u53(w.x, w.y);
            hit[e] = live && (pc >= rand);
            if (hit[e]) {
                const double rtheta = __dmul_rn(__dmul_rn(pcl_u53(w.z, w.w), 2.0), PCL_PI);
                const pcl_u32x4 w2 = pcl_philox4x32_10((pcl_u32)id, (pcl_u32)(id >> 32), a.step, 1u, k0, k1);
                const double rphi = __dmul_rn(pcl_u53(w2.x, w2.y), PCL_PI);
                pcl_new_velocity(a.c, rtheta, rphi, cur.V[0][e], cur.V[1][e], cur.V[2][e]);
            }
            w_hits += (pcl_u32)__popcll(__ballot(hit[e]));
            w_sx += (pcl_u32)__popcll(__ballot(live && cur.V[0][e] > 0.0));
            w_sy += (pcl_u32)__popcll(__ballot(live && cur.V[1][e] > 0.0));
            w_sz += (pcl_u32)__popcll(__ballot(live && cur.V[2][e] > 0.0));
        }
        if (live_q) {
            pcl_vec<VEC>::st(a.r0, q, cur.R[0]);
            pcl_vec<VEC>::st(a.r1, q, cur.R[1]);
            pcl_vec<VEC>::st(a.r2, q, cur.R[2]);
            pcl_vec<VEC>::st(a.vo0, q, cur.V[0]);
            pcl_vec<VEC>::st(a.vo1, q, cur.V[1]);
            pcl_vec<VEC>::st(a.vo2, q, cur.V[2]);
        }
#ifndef PCL_FAST_NOPIPE
        cur = nxt;
#else
        if (base + stride < nq)
            pcl_fast_load<USE_E, VEC>(a, base + stride + threadIdx.x < nq ? base + stride + threadIdx.x : 0, cur);
#endif
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

#ifndef PCL_FAST_VEC
#define PCL_FAST_VEC 2
#endif

#ifdef PCL_RTC
// hipRTC translation unit: one expression, both wavelength variants of both kernels.
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_sphere_e0(pcl_sphere_args a) { pcl_sphere_body<false, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_sphere_e1(pcl_sphere_args a) { pcl_sphere_body<true, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_scatter_e0(pcl_scatter_args a) { pcl_scatter_body<false, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_scatter_e1(pcl_scatter_args a) { pcl_scatter_body<true, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_fused_e0(pcl_fused_args a) { pcl_fused_body<false, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_fused_e1(pcl_fused_args a) { pcl_fused_body<true, true>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_fast_e0(pcl_fast_args a) { pcl_fast_body<false, true, PCL_FAST_VEC>(a); }
extern "C" __global__ void __launch_bounds__(256) pcl_rtc_fast_e1(pcl_fast_args a) { pcl_fast_body<true, true, PCL_FAST_VEC>(a); }
#endif
#endif // PCL_DEVICE_H
