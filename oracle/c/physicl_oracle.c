/* physicl_oracle.c -- plain-C (OpenMP) restatement of the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same algorithm as oracle/physicl_oracle.py (which is pinned against the reference's golden
 * vectors); tests/test_oracle_c.py pins this file against that one.  Used (a) as a second,
 * independent checker and (b) as the multi-core "cpu_baseline" of bench.py (kind "port").
 * Never linked into or called from physicl_amd.  Citations are into /root/reference.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <omp.h>

#define PI 3.141592653589793

int orc_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { omp_set_num_threads(n); }

/* newton.py:15-16 : dr = v*dt (rounded, stored); r = r + dr */
void orc_newton(double *r0, double *r1, double *r2, const double *v0, const double *v1, const double *v2,
                double *dr0, double *dr1, double *dr2, double dt, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double a = v0[i] * dt, b = v1[i] * dt, c = v2[i] * dt;
        dr0[i] = a; dr1[i] = b; dr2[i] = c;
        r0[i] = r0[i] + a; r1[i] = r1[i] + b; r2[i] = r2[i] + c;
    }
}

/* Philox4x32-10, Salmon et al. SC'11 */
static inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u53(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

void orc_philox_words(const int64_t *ids, int64_t n, uint32_t step, uint32_t block, uint64_t seed, uint32_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        philox((uint32_t)ids[i], (uint32_t)((uint64_t)ids[i] >> 32), step, block, (uint32_t)seed,
               (uint32_t)(seed >> 32), out + 4 * i);
}

static inline double step_norm(double a, double b, double c) { return sqrt((a * a + b * b) + c * c); }

/* light.py:146-158 / 239-249 : flag = (A*n*norm >= rand) */
void orc_delete_flags(const double *d0, const double *d1, const double *d2, const double *rand, double A, double n,
                      int32_t *flags, int64_t N) {
    const double An = A * n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) flags[i] = (An * step_norm(d0[i], d1[i], d2[i]) >= rand[i]) ? 1 : 0;
}

/* A delete run body after body (test/test_light.py:52-59: [UpdateTimeStep, NewtonianKinematicsStep, ScatterDeleteStep]): a photon of
 * such a run never changes its velocity, so body k moves it by dr = v*dt (newton.py:15), flags it if (A*n)*|dr| >= rand(id, step0 + k)
 * (light.py:239-249; the device draw of oracle/physicl_oracle.py:philox_draws: decision block (id, step >> 1, 0), even step
 * u53(w0, w1), odd step u53(w2, w3)) and the flagged photon leaves the list (light.py:258-260).
 * death[i] = the body that removes photon i (0 .. K-1), K if none of the K bodies does.  ids == NULL: id = id_base + i. */
void orc_delete_chain(const double *v0, const double *v1, const double *v2, const int64_t *ids, int64_t id_base, int64_t N,
                      double dt, double A, double n, uint64_t seed, uint32_t step0, int K, int32_t *death) {
    const double An = A * n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const double p = An * step_norm(v0[i] * dt, v1[i] * dt, v2[i] * dt);
        const uint64_t id = (uint64_t)(ids ? ids[i] : id_base + i);
        int k = 0;
        uint32_t w[4] = {0, 0, 0, 0};
        for (; k < K; ++k) {
            const uint32_t step = step0 + (uint32_t)k;
            if (k == 0 || (step & 1u) == 0u)
                philox((uint32_t)id, (uint32_t)(id >> 32), step >> 1, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
            const double rand = (step & 1u) ? u53(w[2], w[3]) : u53(w[0], w[1]);
            if (p >= rand) break;
        }
        death[i] = k;
    }
}

/* light.py:258-260 + __init__.py:455-459 : stable removal; returns the survivor count */
int64_t orc_compact_indices(const int32_t *flags, int64_t N, int64_t *idx) {
    int64_t k = 0;
    for (int64_t i = 0; i < N; ++i)
        if (flags[i] == 0) idx[k++] = i;
    return k;
}

/* Fused ScatterIsotropicStep (kernel light.py:303-315 + write-back light.py:325-331) with the
 * device-RNG draw order of oracle/physicl_oracle.py:philox_draws.
 * profile 0: pcoll = A*n*norm ; profile 1: pcoll = A*(k*exp(r0 - off))*norm  (the expression
 * "k * exp(r0[gid] - off)" of examples/variable_n_scattering.ipynb:30); use_E: * pow((h*c)/E, -4).
 * ids == NULL: id = id_base + i.  rt/rp/ra non-NULL: randoms are inputs instead of Philox.
 * Returns the number of hits. */
int64_t orc_scatter_isotropic(const double *d0, const double *d1, const double *d2, const double *E, const double *r0,
                              double *v0, double *v1, double *v2, double *dv0, double *dv1, double *dv2,
                              const int64_t *ids, int64_t id_base, int64_t N, double A, double n, double c, double h,
                              int use_E, int profile, double prof_k, double prof_off, uint64_t seed, uint32_t step,
                              const double *rt, const double *rp, const double *ra) {
    int64_t hits = 0;
#pragma omp parallel for schedule(static) reduction(+ : hits)
    for (int64_t i = 0; i < N; ++i) {
        const double norm = step_norm(d0[i], d1[i], d2[i]);
        double p = profile == 1 ? (A * (prof_k * exp(r0[i] - prof_off))) * norm : (A * n) * norm;
        if (use_E) p = p * pow((h * c) / E[i], -4.0);
        double rand, rtheta, rphi;
        uint32_t w[4];
        const uint64_t id = (uint64_t)(ids ? ids[i] : id_base + i);
        if (ra) {
            rand = ra[i];
        } else {
            philox((uint32_t)id, (uint32_t)(id >> 32), step >> 1, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
            rand = (step & 1u) ? u53(w[2], w[3]) : u53(w[0], w[1]);
        }
        if (p >= rand) {
            if (ra) {
                rtheta = rt[i]; rphi = rp[i];
            } else {
                philox((uint32_t)id, (uint32_t)(id >> 32), step, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
                rtheta = u53(w[0], w[1]) * 2 * PI;
                rphi = u53(w[2], w[3]) * PI;
            }
            const double st = sin(rtheta), ct = cos(rtheta), sp = sin(rphi), cp = cos(rphi);
            const double n0 = (c * st) * cp, n1 = (c * st) * sp, n2 = c * ct;
            dv0[i] = n0 - v0[i]; dv1[i] = n1 - v1[i]; dv2[i] = n2 - v2[i];
            v0[i] = n0; v1[i] = n1; v2[i] = n2;
            ++hits;
        } else {
            dv0[i] = 0.0; dv1[i] = 0.0; dv2[i] = 0.0;
        }
    }
    return hits;
}

/* light.py:424-426 and 385-399 */
void orc_counters(const double *v0, const double *v1, const double *v2, const double *x, const double *dx, double L,
                  int64_t N, int64_t out[4]) {
    int64_t a = 0, b = 0, c = 0, d = 0;
#pragma omp parallel for schedule(static) reduction(+ : a, b, c, d)
    for (int64_t i = 0; i < N; ++i) {
        a += v0[i] > 0.0; b += v1[i] > 0.0; c += v2[i] > 0.0;
        if (x) {
            const double p = x[i] - dx[i];
            d += ((p <= L && L <= x[i]) || (p >= L && L >= x[i]));
        }
    }
    out[0] = a; out[1] = b; out[2] = c; out[3] = d;
}
