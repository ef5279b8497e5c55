// bw_probe.hip -- micro-benchmark: HBM throughput of the fused pass's access pattern under different
// data layouts (plain SoA vs tiled AoSoA), to decide the store layout with measurements.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/attic/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int NRD = 7, NWR = 12, NF = 13; // fused pass at h = 1: read r,v,E ; write r,dr,dv,v

// SoA: field f at base + f*stride_f, element i at [i]
template <int UNROLL>
__global__ void __launch_bounds__(256) k_soa(double *base, long stride_f, long npair) {
    const long stride = (long)gridDim.x * blockDim.x * UNROLL;
    for (long p0 = ((long)blockIdx.x * blockDim.x) * UNROLL + threadIdx.x; p0 < npair; p0 += stride) {
        double2 x[UNROLL][NRD];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long p = p0 + u * blockDim.x;
            if (p < npair)
#pragma unroll
                for (int f = 0; f < NRD; ++f) x[u][f] = reinterpret_cast<const double2 *>(base + f * stride_f)[p];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long p = p0 + u * blockDim.x;
            if (p < npair)
#pragma unroll
                for (int f = 0; f < NWR; ++f) {
                    double2 y = x[u][f % NRD];
                    y.x += 1.0; y.y *= 1.5;
                    reinterpret_cast<double2 *>(base + (f < 6 ? f : f + 1) * stride_f)[p] = y; // fields 0..5,7..12
                }
        }
    }
}

// AoSoA: tile of T particles holds NF fields back to back: element i of field f at
//   base[((i / T) * NF + f) * T + i % T]
template <int LOGT>
__global__ void __launch_bounds__(256) k_aosoa(double *base, long npair) {
    constexpr long T = 1L << LOGT, TP = T / 2; // pairs per tile
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
        const long tile = p >> (LOGT - 1), lp = p & (TP - 1);
        double2 *tb = reinterpret_cast<double2 *>(base + tile * NF * T);
        double2 x[NRD];
#pragma unroll
        for (int f = 0; f < NRD; ++f) x[f] = tb[f * TP + lp];
#pragma unroll
        for (int f = 0; f < NWR; ++f) {
            double2 y = x[f % NRD];
            y.x += 1.0; y.y *= 1.5;
            tb[(f < 6 ? f : f + 1) * TP + lp] = y;
        }
    }
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 100000000L;
    const long npair = N / 2;
    const long stride_f = ((N + 511) / 512) * 512 + (argc > 2 ? atol(argv[2]) : 0); // optional skew in elements
    double *buf;
    CK(hipMalloc(&buf, sizeof(double) * stride_f * NF + 4096));
    CK(hipMemset(buf, 0, sizeof(double) * stride_f * NF));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = (double)N * 8 * (NRD + NWR);
    auto time_it = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a));
        const int R = 10;
        for (int i = 0; i < R; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-40s %7.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
    };
    for (int g : {2048, 3072, 4096}) {
        char nm[64];
        snprintf(nm, 64, "SoA   unroll1 grid%d", g); time_it(nm, [&] { k_soa<1><<<g, 256>>>(buf, stride_f, npair); });
        snprintf(nm, 64, "SoA   unroll2 grid%d", g); time_it(nm, [&] { k_soa<2><<<g, 256>>>(buf, stride_f, npair); });
        snprintf(nm, 64, "AoSoA T=128   grid%d", g); time_it(nm, [&] { k_aosoa<7><<<g, 256>>>(buf, npair); });
        snprintf(nm, 64, "AoSoA T=512   grid%d", g); time_it(nm, [&] { k_aosoa<9><<<g, 256>>>(buf, npair); });
        snprintf(nm, 64, "AoSoA T=2048  grid%d", g); time_it(nm, [&] { k_aosoa<11><<<g, 256>>>(buf, npair); });
        snprintf(nm, 64, "AoSoA T=8192  grid%d", g); time_it(nm, [&] { k_aosoa<13><<<g, 256>>>(buf, npair); });
    }
    CK(hipGetLastError());
    return 0;
}
