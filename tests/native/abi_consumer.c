/* A host program in plain C that binds libphysicl_hip.so the way a non-Python host would:
 * nothing but include/physicl_hip.h, pointers and sizes.  tests/test_gpu_c_consumer.py compiles it with gcc, runs it
 * and compares what it prints with the CPU oracle on the same inputs (generated here and there by the same 64-bit LCG).
 *
 *   abi_consumer N      prints:  flags <N chars 0/1>     kernel `test` of ScatterDeleteStep      physicl/light.py:239-249
 *                                del <N chars 0/1>       kernel light_scatter_step_del           physicl/light.py:146-158
 *                                keep <n_keep> <sum of surviving indices>                        physicl/light.py:258-260
 *                                r <3N hex doubles>      r after one pcl_step_newton             physicl/newton.py:15-16
 *                                dr <3N hex doubles>
 */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "physicl_hip.h"

#define CK(call)                                                                                   \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != PCL_OK) {                                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pcl_last_error());                     \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double lcg_uniform(void) { /* 53 random bits in [0, 1) */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}

static void print_hex(const char *tag, const double *a, int64_t n) {
    printf("%s", tag);
    for (int64_t i = 0; i < n; ++i) {
        uint64_t u;
        memcpy(&u, &a[i], 8);
        printf(" %016" PRIx64, u);
    }
    printf("\n");
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 1000;
    const double A = 1e-3, n = 1e-3, dt = 1e-3, c = 299792458.0;
    if (pcl_abi_version() != PCL_ABI_VERSION) {
        fprintf(stderr, "header says ABI %d, library %d\n", PCL_ABI_VERSION, pcl_abi_version());
        return 1;
    }
    double *h[4]; /* d0 d1 d2 rand */
    for (int k = 0; k < 4; ++k) h[k] = (double *)malloc((size_t)N * 8);
    for (int64_t i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) h[k][i] = (2.0 * lcg_uniform() - 1.0) * c * dt;
        h[3][i] = lcg_uniform();
    }
    pcl_ctx *ctx = NULL;
    CK(pcl_ctx_create(0, NULL, &ctx));
    void *d[4], *res = NULL, *idx = NULL;
    for (int k = 0; k < 4; ++k) {
        CK(pcl_dev_alloc(ctx, N * 8, &d[k]));
        CK(pcl_h2d(ctx, d[k], h[k], N * 8));
    }
    CK(pcl_dev_alloc(ctx, N * 4, &res));
    CK(pcl_dev_alloc(ctx, N * 8, &idx));
    int32_t *flags = (int32_t *)malloc((size_t)N * 4);
    /* ScatterDeleteStep: kernel test(d0, d1, d2, rand, A, n, res) */
    CK(pcl_k_scatter_delete_test(ctx, (const double *)d[0], (const double *)d[1], (const double *)d[2], (const double *)d[3], A, n,
                                 (int32_t *)res, N));
    CK(pcl_d2h(ctx, flags, res, N * 4));
    printf("flags ");
    for (int64_t i = 0; i < N; ++i) putchar(flags[i] ? '1' : '0');
    printf("\n");
    /* the survivors, as the reference's removal loop leaves them */
    int64_t n_keep = 0;
    CK(pcl_k_compact_indices(ctx, (const int32_t *)res, N, (int64_t *)idx, &n_keep));
    int64_t *keep = (int64_t *)malloc((size_t)(n_keep > 0 ? n_keep : 1) * 8);
    CK(pcl_d2h(ctx, keep, idx, n_keep * 8));
    int64_t sum = 0;
    for (int64_t i = 0; i < n_keep; ++i) {
        if (i && keep[i] <= keep[i - 1]) {
            fprintf(stderr, "indices not ascending at %" PRId64 "\n", i);
            return 1;
        }
        sum += keep[i];
    }
    /* ScatterDeleteStepReference: kernel light_scatter_step_del(dx, dy, dz, rand, n, A, result) */
    CK(pcl_k_light_scatter_step_del(ctx, (const double *)d[0], (const double *)d[1], (const double *)d[2], (const double *)d[3], n, A,
                                    (int32_t *)res, N));
    CK(pcl_d2h(ctx, flags, res, N * 4));
    printf("del ");
    for (int64_t i = 0; i < N; ++i) putchar(flags[i] ? '1' : '0');
    printf("\n");
    printf("keep %" PRId64 " %" PRId64 "\n", n_keep, sum);
    /* the resident store: upload r (= d0..d2 scaled) and v, one Newton step, read r and dr back */
    CK(pcl_store_alloc(ctx, N));
    CK(pcl_store_set_count(ctx, N, 0));
    double *v = (double *)malloc((size_t)N * 8), *z = (double *)calloc((size_t)N, 8);
    for (int k = 0; k < 3; ++k) {
        for (int64_t i = 0; i < N; ++i) v[i] = h[k][i] / dt; /* |v| <= c */
        CK(pcl_store_upload(ctx, PCL_R0 + k, h[(k + 1) % 3], 0, N));
        CK(pcl_store_upload(ctx, PCL_V0 + k, v, 0, N));
        CK(pcl_store_upload(ctx, PCL_DR0 + k, z, 0, N));
        CK(pcl_store_upload(ctx, PCL_DV0 + k, z, 0, N));
    }
    CK(pcl_store_upload(ctx, PCL_E, h[3], 0, N));
    CK(pcl_step_newton(ctx, dt));
    double *out = (double *)malloc((size_t)N * 3 * 8);
    for (int k = 0; k < 3; ++k) CK(pcl_store_download(ctx, PCL_R0 + k, out + k * N, 0, N));
    print_hex("r", out, 3 * N);
    for (int k = 0; k < 3; ++k) CK(pcl_store_download(ctx, PCL_DR0 + k, out + k * N, 0, N));
    print_hex("dr", out, 3 * N);
    int64_t cnt[PCL_CNT_PLANE0];
    CK(pcl_step_counters(ctx, NULL, 0, cnt));
    printf("counters %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 "\n", cnt[PCL_CNT_N], cnt[PCL_CNT_XP], cnt[PCL_CNT_YP], cnt[PCL_CNT_ZP]);
    /* TracePathMeasureStep for a tracked subset (physicl/light.py:447-458): where photons 0 and N-1 will be after each of the
     * next three passes of [Newton, ScatterIsotropic(A = n = 1e-3)], worked out BEFORE the launch that runs those passes -- and
     * the launch then leaves them exactly where the last row says */
    {
        const int64_t ids[2] = {0, N - 1};
        const int n_ids = N > 1 ? 2 : 1, kinds[1] = {PCL_PHASE_ISOTROPIC};
        double rows[3 * 2 * 4];
        int64_t mrows[3 * 5];
        CK(pcl_store_trace_ahead(ctx, ids, n_ids, dt, 3, 1, kinds, 0, A, n, 0, 299792458.0, 6.62607015e-34, NULL, 0.0, 0.0, 9, 5, rows));
        CK(pcl_step_fused_multi(ctx, dt, 3, A, n, 0, 299792458.0, 6.62607015e-34, NULL, 9, 5, NULL, 0, mrows));
        print_hex("trace", rows, (int64_t)3 * n_ids * 4);
        for (int j = 0; j < n_ids; ++j)
            for (int k = 0; k < 3; ++k) {
                double got;
                CK(pcl_store_download(ctx, PCL_R0 + k, &got, ids[j], 1));
                if (memcmp(&got, &rows[(2 * n_ids + j) * 4 + k], 8) != 0) {
                    fprintf(stderr, "photon %" PRId64 ": the launch left r%d elsewhere than the trace said\n", ids[j], k);
                    return 1;
                }
            }
        printf("traced %d\n", n_ids);
    }
    CK(pcl_store_free(ctx));
    for (int k = 0; k < 4; ++k) CK(pcl_dev_free(ctx, d[k]));
    CK(pcl_dev_free(ctx, res));
    CK(pcl_dev_free(ctx, idx));
    CK(pcl_ctx_destroy(ctx));
    return 0;
}
