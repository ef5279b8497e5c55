// Does spreading the concurrently written tiles over the whole slab remove the slow kind?  13-row tile sweep in tile
// order (window of ~0.5 GB active at a time) vs in a scattered order (tile = i * P mod tiles), on a physically
// contiguous block (always slow in order) and on hipMalloc blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) sweep(double2 *slab, long tiles, long P) {
    for (long i = blockIdx.x; i < tiles; i += gridDim.x) {
        const long t = P ? (i * P) % tiles : i;
        double2 *tile = slab + t * (17 * 1024);
#pragma unroll
        for (int row = 0; row < 13; ++row)
            for (int q = threadIdx.x; q < 1024; q += 256) tile[row * 1024 + q] = make_double2(1.0 + row, 2.0);
    }
}
static double rate(void *p, long tiles, long P) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    sweep<<<2048, 256>>>((double2 *)p, tiles, P); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int k = 0; k < 3; ++k) sweep<<<2048, 256>>>((double2 *)p, tiles, P);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return 3.0 * tiles * 13 * 16384 / (ms * 1e-3) / 1e12;
}
int main() {
    const long tiles = 48829;
    const size_t bytes = (size_t)tiles * 17 * 16384;
    const long Ps[] = {0, 7, 97, 1021, 7919, 24499};
    for (int rep = 0; rep < 2; ++rep) {
        void *c = nullptr;
        CK(hipExtMallocWithFlags(&c, bytes, hipDeviceMallocContiguous));
        printf("contiguous:");
        for (long P : Ps) printf("  P=%ld %.2f", P, rate(c, tiles, P));
        printf(" TB/s\n");
        CK(hipFree(c));
        void *m[3];
        for (int k = 0; k < 3; ++k) CK(hipMalloc(&m[k], bytes));
        for (int k = 0; k < 3; ++k) {
            printf("hipMalloc %d:", k);
            for (long P : Ps) printf("  P=%ld %.2f", P, rate(m[k], tiles, P));
            printf(" TB/s\n");
        }
        for (int k = 0; k < 3; ++k) CK(hipFree(m[k]));
        fflush(stdout);
    }
    return 0;
}
