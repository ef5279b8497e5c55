// Is the slow/fast kind a property of the whole block or of its 1 GB physical handles?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) sweep(double2 *slab, long tiles) {
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        double2 *tile = slab + t * (17 * 1024);
#pragma unroll
        for (int row = 0; row < 13; ++row)
            for (int q = threadIdx.x; q < 1024; q += 256) tile[row * 1024 + q] = make_double2(1.0 + row, 2.0);
    }
}
static double rate(void *p, long tiles, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    sweep<<<2048, 256>>>((double2 *)p, tiles); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int k = 0; k < reps; ++k) sweep<<<2048, 256>>>((double2 *)p, tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)reps * tiles * 13 * 16384 / (ms * 1e-3) / 1e12;
}
int main() {
    const size_t tile = 17 * 16384, chunk_tiles = 3855, chunk = chunk_tiles * tile; // ~1 GB, whole tiles
    const int nchunk = 13, nblk = 5;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int b = 0; b < nblk; ++b) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, chunk * nchunk, 2 << 20, nullptr, 0));
        for (int k = 0; k < nchunk; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char *)va + k * chunk, chunk, 0, h, 0)); }
        CK(hipMemSetAccess(va, chunk * nchunk, &acc, 1));
        printf("block %d whole: %.2f TB/s; per 1 GB handle:", b, rate(va, chunk_tiles * nchunk, 3));
        for (int k = 0; k < nchunk; ++k) printf(" %.2f", rate((char *)va + k * chunk, chunk_tiles, 20));
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
