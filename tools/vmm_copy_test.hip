// Do copies and memsets work across the physical handles of one hipMemMap'ed range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } else printf("%s ok\n", #x); } while (0)
int main() {
    const size_t chunk = (size_t)64 << 20, n = 3;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, chunk * n, 2 << 20, nullptr, 0));
    for (size_t k = 0; k < n; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char *)va + k * chunk, chunk, 0, h, 0)); }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, chunk * n, &acc, 1));
    std::vector<char> host(chunk * n, 1);
    CK(hipMemset(va, 7, chunk * n));
    CK(hipMemcpy(host.data(), va, chunk * n, hipMemcpyDeviceToHost));                       // 1-D across all handles
    printf("host[0]=%d host[last]=%d\n", host[0], host[chunk * n - 1]);
    CK(hipMemcpy((char *)va + chunk - 4096, host.data(), 8192, hipMemcpyHostToDevice));    // small 1-D straddling a boundary
    CK(hipMemcpy2D(host.data(), 16384, (char *)va + chunk - 20 * 278528, 278528, 16384, 40, hipMemcpyDeviceToHost)); // 2-D straddling
    CK(hipMemcpy2D(host.data(), 16384, va, 278528, 16384, 40, hipMemcpyDeviceToHost));     // 2-D inside one handle
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(hipMemcpy2DAsync(host.data(), 16384, (char *)va + chunk - 20 * 278528, 278528, 16384, 40, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpyAsync(host.data(), (char *)va + chunk - 4096, 8192, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return 0;
}
