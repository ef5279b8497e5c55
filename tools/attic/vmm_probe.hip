// Which kind of device allocation gives the 13-stream tile sweep its fast mode?  ("Placement")
//   hipcc --offload-arch=gfx950 -O3 tools/attic/vmm_probe.hip -o gpurun_out/vmm_probe && gpurun_out/vmm_probe
// For each allocation method: a kernel that writes 13 of the 17 rows of every [17][2048] fp64 tile (the store's fill
// pattern), timed with HIP events over 4 sweeps; three allocations of each kind.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
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

static int measure(const char *name, void *p, long tiles) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    sweep<<<2048, 256>>>((double2 *)p, tiles);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int k = 0; k < 4; ++k) sweep<<<2048, 256>>>((double2 *)p, tiles);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("  %-34s %.2f TB/s\n", name, 4.0 * tiles * 13 * 16384 / (ms * 1e-3) / 1e12);
    fflush(stdout);
    return 0;
}

struct vmm_block { void *va; size_t bytes; std::vector<hipMemGenericAllocationHandle_t> h; };

static int vmm_alloc(vmm_block &blk, size_t bytes, size_t chunk, bool shuffle) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran;
    if (chunk > bytes) chunk = (bytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    blk.bytes = n * chunk;
    CK(hipMemAddressReserve(&blk.va, blk.bytes, 0, nullptr, 0));
    blk.h.resize(n);
    for (size_t k = 0; k < n; ++k) CK(hipMemCreate(&blk.h[k], chunk, &prop, 0));
    std::vector<size_t> order(n);
    for (size_t k = 0; k < n; ++k) order[k] = k;
    if (shuffle) { std::mt19937_64 g(12345); std::shuffle(order.begin(), order.end(), g); }
    for (size_t k = 0; k < n; ++k) CK(hipMemMap((char *)blk.va + k * chunk, chunk, 0, blk.h[order[k]], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(blk.va, blk.bytes, &acc, 1));
    return 0;
}

static void vmm_free(vmm_block &blk) {
    (void)hipMemUnmap(blk.va, blk.bytes);
    for (auto h : blk.h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(blk.va, blk.bytes);
}

int main(int argc, char **argv) {
    const long tiles = argc > 1 ? atol(argv[1]) : 48829; // 1e8 photons
    const size_t bytes = (size_t)tiles * 17 * 16384;
    size_t gran = 0;
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t gmin = 0;
        CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        printf("%ld tiles, %.2f GB; VMM granularity recommended %zu, minimum %zu\n", tiles, bytes / 1e9, gran, gmin);
    }
    for (int rep = 0; rep < 3; ++rep) {
        printf("rep %d\n", rep);
        void *p[3];
        for (int k = 0; k < 3; ++k) CK(hipMalloc(&p[k], bytes));
        for (int k = 0; k < 3; ++k) if (measure("hipMalloc", p[k], tiles)) return 1;
        for (int k = 0; k < 3; ++k) CK(hipFree(p[k]));
        void *c = nullptr;
        if (hipExtMallocWithFlags(&c, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            if (measure("hipExtMalloc contiguous", c, tiles)) return 1;
            CK(hipFree(c));
        } else { (void)hipGetLastError(); printf("  contiguous refused\n"); }
        {
            void *a = nullptr;
            hipStream_t st; CK(hipStreamCreate(&st));
            if (hipMallocAsync(&a, bytes, st) == hipSuccess) {
                CK(hipStreamSynchronize(st));
                if (measure("hipMallocAsync", a, tiles)) return 1;
                CK(hipFreeAsync(a, st)); CK(hipStreamSynchronize(st));
            } else { (void)hipGetLastError(); printf("  hipMallocAsync refused\n"); }
            void *u = nullptr;
            if (hipExtMallocWithFlags(&u, bytes, hipDeviceMallocUncached) == hipSuccess) {
                if (measure("hipExtMalloc uncached", u, tiles)) return 1;
                CK(hipFree(u));
            } else { (void)hipGetLastError(); printf("  uncached refused\n"); }
        }
        struct { const char *name; size_t chunk; bool shuffle; } v[] = {
            {"VMM 2 MB chunks, in order", (size_t)2 << 20, false}, {"VMM 2 MB chunks, shuffled", (size_t)2 << 20, true},
            {"VMM 64 MB chunks, shuffled", (size_t)64 << 20, true}, {"VMM 64 MB chunks, in order", (size_t)64 << 20, false},
            {"VMM 256 MB chunks, in order", (size_t)256 << 20, false}, {"VMM 1 GB chunks, in order", (size_t)1 << 30, false},
            {"VMM one 13.6 GB handle", (size_t)1 << 40, false}};
        for (auto &x : v) {
            vmm_block blk;
            if (vmm_alloc(blk, bytes, x.chunk, x.shuffle)) return 1;
            if (measure(x.name, blk.va, tiles)) return 1;
            vmm_free(blk);
        }
    }
    return 0;
}
