// bw_probe2.hip -- HBM throughput of the FAST fused pass's stream mix (reads r0-2, v0-2, lam4; writes r0-2, vnew0-2:
// 104 B per fp64 particle) under three placements: 17 separate hipMallocs (what the library does), one slab,
// and tiled AoSoA.   hipcc --offload-arch=gfx950 -O3 tools/attic/bw_probe2.hip -o /tmp/bw_probe2 && /tmp/bw_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct ptrs { const double *rd[7]; double *wr[6]; };

__global__ void __launch_bounds__(256) k_soa(ptrs a, long npair) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
        double2 x[7];
#pragma unroll
        for (int f = 0; f < 7; ++f) x[f] = reinterpret_cast<const double2 *>(a.rd[f])[p];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            double2 y = x[f];
            y.x += x[6].x; y.y *= 1.5;
            reinterpret_cast<double2 *>(a.wr[f])[p] = y;
        }
    }
}

// AoSoA tile of T particles: [r0 r1 r2 | vA0 vA1 vA2 | vB0 vB1 vB2 | lam4] x T, v double buffer inside the tile
template <int LOGT>
__global__ void __launch_bounds__(256) k_aosoa(double *base, long npair, int flip) {
    constexpr long T = 1L << LOGT, TP = T / 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
        const long tile = p >> (LOGT - 1), lp = p & (TP - 1);
        double2 *tb = reinterpret_cast<double2 *>(base + tile * 10 * T);
        const int vi = flip ? 6 : 3, vo = flip ? 3 : 6;
        double2 x[7];
#pragma unroll
        for (int f = 0; f < 3; ++f) x[f] = tb[f * TP + lp];
#pragma unroll
        for (int f = 0; f < 3; ++f) x[3 + f] = tb[(vi + f) * TP + lp];
        x[6] = tb[9 * TP + lp];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            double2 y = x[f];
            y.x += x[6].x; y.y *= 1.5;
            tb[((f < 3 ? f : vo + f - 3)) * TP + lp] = y;
        }
    }
}

int main() {
    const long N = 100000000L, npair = N / 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = (double)N * 104;
    auto time_it = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipEventRecord(a));
        const int R = 20;
        for (int i = 0; i < R; ++i) launch(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
    };
    {   // separate allocations: r(3) vA(3) vB(3) lam4 + 7 dummies in between like the real store (dr, dv, E)
        double *f[17];
        for (int i = 0; i < 17; ++i) { CK(hipMalloc(&f[i], N * 8 + 512)); CK(hipMemset(f[i], 0, N * 8)); }
        ptrs pa{{f[0], f[1], f[2], f[3], f[4], f[5], f[16]}, {f[0], f[1], f[2], f[13], f[14], f[15]}};
        ptrs pb{{f[0], f[1], f[2], f[13], f[14], f[15], f[16]}, {f[0], f[1], f[2], f[3], f[4], f[5]}};
        for (int g : {2048, 3072}) {
            char nm[64]; snprintf(nm, 64, "SoA separate hipMallocs   grid%d", g);
            time_it(nm, [&](int i) { k_soa<<<g, 256>>>((i & 1) ? pb : pa, npair); });
        }
        for (int i = 0; i < 17; ++i) CK(hipFree(f[i]));
    }
    {   // one slab
        double *s; const long stride = N + 544;
        CK(hipMalloc(&s, stride * 8 * 17)); CK(hipMemset(s, 0, stride * 8 * 17));
        auto F = [&](int i) { return s + i * stride; };
        ptrs pa{{F(0), F(1), F(2), F(3), F(4), F(5), F(16)}, {F(0), F(1), F(2), F(13), F(14), F(15)}};
        ptrs pb{{F(0), F(1), F(2), F(13), F(14), F(15), F(16)}, {F(0), F(1), F(2), F(3), F(4), F(5)}};
        time_it("SoA one slab              grid2048", [&](int i) { k_soa<<<2048, 256>>>((i & 1) ? pb : pa, npair); });
        CK(hipFree(s));
    }
    {   // AoSoA
        double *s; CK(hipMalloc(&s, (N + 65536) * 8 * 10)); CK(hipMemset(s, 0, (N + 65536) * 8 * 10));
        for (int g : {2048, 3072}) {
            char nm[64];
            snprintf(nm, 64, "AoSoA T=256               grid%d", g); time_it(nm, [&](int i) { k_aosoa<8><<<g, 256>>>(s, npair, i & 1); });
            snprintf(nm, 64, "AoSoA T=512               grid%d", g); time_it(nm, [&](int i) { k_aosoa<9><<<g, 256>>>(s, npair, i & 1); });
            snprintf(nm, 64, "AoSoA T=2048              grid%d", g); time_it(nm, [&](int i) { k_aosoa<11><<<g, 256>>>(s, npair, i & 1); });
            snprintf(nm, 64, "AoSoA T=8192              grid%d", g); time_it(nm, [&](int i) { k_aosoa<13><<<g, 256>>>(s, npair, i & 1); });
        }
        CK(hipFree(s));
    }
    CK(hipGetLastError());
    return 0;
}
