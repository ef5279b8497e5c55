// hipMemcpy2DAsync with many rows over a range made of several physical handles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
int main(int argc, char **argv) {
    const size_t chunk = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    const size_t pitch = 278528, width = 16384;
    const size_t heights[] = {1000, 3800, 3900, 8000, 16000, 48828};
    const size_t total = ((48828 * pitch + chunk - 1) / chunk) * chunk;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 2 << 20, nullptr, 0));
    for (size_t off = 0; off < total; off += chunk) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char *)va + off, chunk, 0, h, 0)); }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    CK(hipMemset(va, 5, total));
    std::vector<char> host(48828 * width, 1);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t h : heights) {
        hipError_t e = hipMemcpy2DAsync(host.data(), width, va, pitch, width, h, hipMemcpyDeviceToHost, s);
        hipError_t e2 = hipStreamSynchronize(s);
        printf("chunk %zu MB, 2-D D2H height %zu (span %.2f GB): %s / %s; first %d last %d\n", chunk >> 20, h, h * pitch / 1e9, hipGetErrorString(e),
               hipGetErrorString(e2), host[0], host[h * width - 1]);
        (void)hipGetLastError();
        e = hipMemcpy2DAsync(va, pitch, host.data(), width, width, h, hipMemcpyHostToDevice, s);
        e2 = hipStreamSynchronize(s);
        printf("   H2D: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
        (void)hipGetLastError();
    }
    return 0;
}
