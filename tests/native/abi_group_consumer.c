/* A host program in plain C that drives several library contexts from ONE process through the device-group entry
 * points of include/physicl_hip.h (pcl_group_*): the C-level form of the reference's single-process shape
 * (physicl/__init__.py:400-432).  tests/test_gpu_c_consumer.py compiles it with gcc -std=c99 -pedantic, runs it with
 * one, two and three contexts on device 0 and requires identical output: rows of K scatter steps in one launch, rows of
 * a delete-until-empty run one call per loop body, and a checksum over the surviving ids.
 *
 *   abi_group_consumer N n_ctx
 */
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "physicl_hip.h"

#define CK(call)                                                                                   \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != PCL_OK) {                                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pcl_last_error());                     \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int G = argc > 2 ? atoi(argv[2]) : 2;
    const double c = 299792458.0, h = 6.62607015e-34;
    int dev[64];
    for (int i = 0; i < G && i < 64; ++i) dev[i] = 0;
    pcl_group *g = NULL;
    CK(pcl_group_create(G, dev, &g));
    int n = 0;
    CK(pcl_group_size(g, &n));
    if (n != G) return 2;
    CK(pcl_group_store_alloc(g, N, PCL_DTYPE_F64));
    CK(pcl_group_fill_photons(g, N, 1000, c, 2.8e-19, 9.9e-19, 77));
    /* K = 6 loop bodies [Newton, ScatterIsotropic, sign rows] in one launch per shard: rows = sums over the shards */
    enum { K = 6 };
    int64_t rows[K][5];
    CK(pcl_group_step_fused_multi(g, 1e-3, K, 1e-3, 1e-3, 0, c, h, NULL, 77, 1, NULL, 0, &rows[0][0]));
    for (int k = 0; k < K; ++k)
        printf("iso %d %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 "\n", k, rows[k][0], rows[k][1], rows[k][2], rows[k][3], rows[k][4]);
    /* delete until empty, one call per loop body, with a plane counter; then the survivors' ids half way */
    const double plane[3] = {2.0e6, NAN, NAN};
    int64_t alive = N, body = 0, idsum = 0;
    while (alive > 0 && body < 4096) {
        int64_t o[6];
        CK(pcl_group_step_fused_delete(g, 1e-3, 1e-3, 1e-3, PCL_FUSED_LAZY, PCL_RNG_PHILOX, 77, (uint32_t)(100 + body), plane, 1, o));
        alive = o[0];
        if (body < 12) printf("del %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 " %" PRId64 "\n", body, o[0], o[1], o[2], o[3], o[4], o[5]);
        if (body == 5) { /* global particle order: ids ascending across the shards */
            int64_t cnt = 0, prev = -1;
            CK(pcl_group_count(g, &cnt));
            int64_t *ids = (int64_t *)malloc((size_t)(cnt > 0 ? cnt : 1) * 8);
            CK(pcl_group_download_ids(g, ids, 0, cnt));
            for (int64_t i = 0; i < cnt; ++i) {
                if (ids[i] <= prev) return 3;
                prev = ids[i];
                idsum += ids[i] % 1000003;
            }
            free(ids);
            printf("ids %" PRId64 " %" PRId64 "\n", cnt, idsum);
        }
        ++body;
    }
    printf("bodies %" PRId64 "\n", body);
    CK(pcl_group_destroy(g));
    return 0;
}
