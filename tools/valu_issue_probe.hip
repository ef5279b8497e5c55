// valu_issue_probe.hip -- what one wave64 vector instruction costs a SIMD on gfx950, class by class.
//
// The K-step kernels of libphysicl_hip are bound by VALU issue, not by HBM.  Their roofline record therefore needs a
// ceiling in SIMD-cycles and a price per instruction class; "1024 SIMDs x 2.4 GHz / 4 cycles per instruction" (rounds 2-4)
// is neither (a 32-bit VALU instruction issues in 2 cycles on a busy SIMD, a quarter-rate integer multiply in 8 or more,
// and the chip does not hold 2.4 GHz under these kernels).  This probe measures both:
//
//   * per class: a stream of independent instructions of ONE opcode (8 accumulator chains, unrolled 8 x: 64 per loop trip),
//     every SIMD of the chip holding W waves (W = 1, 2, 4, 8: grid = CUs x W workgroups of 4 waves; where each wave ran
//     is read from HW_ID / XCC_ID and the table says how even the placement was);
//     cycles per wave-instruction per SIMD = (last s_memtime - first s_memtime over the waves of a SIMD) / (W x instructions),
//     median over the SIMDs (s_memtime ticks in shader cycles, MI355X_MICROARCH.md "s_memtime tick");
//   * the clock the chip held: delta s_memtime / delta s_memrealtime x 100 MHz (same guide, "DVFS give-back" item 6),
//     and, for comparison, cycles x waves / HIP-event time.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_issue_probe.hip -o /tmp/valu_issue_probe && /tmp/valu_issue_probe [--json out.json]
//
// Run under  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE  to check the clock against the counter (tools/prof_valu_probe.sh).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                                          \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess) {                                                                           \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
            exit(1);                                                                                      \
        }                                                                                                 \
    } while (0)

struct stamp {
    unsigned long long t0, t1;   // s_memtime (shader cycles)
    unsigned long long w0, w1;   // s_memrealtime (100 MHz)
    unsigned hw_id, xcc_id;
};

#define HW_REG_HW_ID 4
#define HW_REG_XCC_ID 20
#define GETREG_ALL(id) ((31 << 11) | (0 << 6) | (id))

// 8 independent chains x 8 = 64 instructions of one opcode per asm block
#define REP8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define BLOCK8(T) REP8(T) REP8(T) REP8(T) REP8(T) REP8(T) REP8(T) REP8(T) REP8(T)

#define PROLOGUE()                                                                                      \
    stamp s;                                                                                            \
    s.hw_id = __builtin_amdgcn_s_getreg(GETREG_ALL(HW_REG_HW_ID));                                      \
    s.xcc_id = __builtin_amdgcn_s_getreg(GETREG_ALL(HW_REG_XCC_ID));                                    \
    __syncthreads();                                                                                    \
    s.w0 = __builtin_amdgcn_s_memrealtime();                                                            \
    s.t0 = __builtin_amdgcn_s_memtime();
#define EPILOGUE(sink)                                                                                  \
    s.t1 = __builtin_amdgcn_s_memtime();                                                                \
    s.w1 = __builtin_amdgcn_s_memrealtime();                                                            \
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s; \
    if ((sink) == 0x7fffffffu) out[0].hw_id = 1; /* keeps the chains alive */

// ---- kernels: 64-bit accumulators -------------------------------------------------------------------------------------
#define KERNEL64(NAME, TEXT)                                                                            \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, double seed) {                   \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7;                                                                             \
        double b = seed * 0.999, c = seed * 1e-3;                                                       \
        PROLOGUE()                                                                                      \
        for (int it = 0; it < iters; ++it)                                                              \
            asm volatile(BLOCK8(TEXT)                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc", "s4", "s5");                                                          \
        const double sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                       \
        EPILOGUE((unsigned)__double2hiint(sum))                                                         \
    }

// ---- kernels: 32-bit accumulators -------------------------------------------------------------------------------------
#define KERNEL32(NAME, CT, TEXT)                                                                        \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, double seed) {                   \
        CT a0 = (CT)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
           a7 = a0 + 7;                                                                                 \
        CT b = (CT)(seed * 3), c = (CT)(seed * 7 + 1);                                                  \
        PROLOGUE()                                                                                      \
        for (int it = 0; it < iters; ++it)                                                              \
            asm volatile(BLOCK8(TEXT)                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc", "s4", "s5");                                                          \
        const CT sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                           \
        unsigned u;                                                                                     \
        __builtin_memcpy(&u, &sum, 4);                                                                  \
        EPILOGUE(u)                                                                                     \
    }

// ---- kernels: 64-bit accumulator fed by 32-bit operands (v_mad_u64_u32), 32-bit result from 64-bit sources (conversions) ----
#define KERNEL_MAD64(NAME, TEXT)                                                                        \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, double seed) {                   \
        unsigned long long a0 = (unsigned long long)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, \
                           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                       \
        unsigned b = (unsigned)(seed * 3) | 0x80000001u, c = (unsigned)(seed * 7) | 0x40000001u;        \
        PROLOGUE()                                                                                      \
        for (int it = 0; it < iters; ++it)                                                              \
            asm volatile(BLOCK8(TEXT)                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc");                                                                      \
        const unsigned long long sum = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                           \
        EPILOGUE((unsigned)(sum >> 7))                                                                  \
    }

// ---- kernels: scalar destination (v_readlane_b32, v_cmp into an SGPR pair) --------------------------------------------------
#define KERNEL_SDST(NAME, TEXT)                                                                         \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, double seed) {                   \
        unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                         \
        unsigned b = (unsigned)(seed * 3) + threadIdx.x;                                                \
        double c = seed + threadIdx.x;                                                                  \
        PROLOGUE()                                                                                      \
        for (int it = 0; it < iters; ++it)                                                              \
            asm volatile(BLOCK8(TEXT)                                                                   \
                         : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc");                                                                      \
        EPILOGUE(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)                                                 \
    }

// one instruction per chain; %8, %9 = the two loop-invariant operands
#define T_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define T_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define T_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define T_RCP_F64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define T_SQRT_F64(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define T_RSQ_F64(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define T_LDEXP_F64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define T_FRACT_F64(i) "v_fract_f64 %" #i ", %" #i "\n"
#define T_RNDNE_F64(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define T_DIVFIX_F64(i) "v_div_fixup_f64 %" #i ", %" #i ", %8, %9\n"
#define T_MINMAX_F64(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define T_LSHL_B64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define T_CMP_F64(i) "v_cmp_ge_f64 vcc, %" #i ", %8\n"
#define T_CMPX_CND64(i) "v_cmp_ge_f64 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %" #i ", vcc\n" /* unused */
#define T_FMAC_F64(i) "v_fmac_f64 %" #i ", %8, %9\n"
#define T_MOV_B64(i) "v_mov_b64 %" #i ", %8\n"
#define T_CMP_CLASS_F64(i) "v_cmp_class_f64 vcc, %" #i ", 60\n"
#define T_CMP_F64_E64(i) "v_cmp_ge_f64_e64 s[4:5], %" #i ", %8\n"
#define T_CNDMASK_PAIR(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
KERNEL64(k_fmac_f64, T_FMAC_F64)
KERNEL64(k_mov_b64, T_MOV_B64)
KERNEL64(k_cmp_class_f64, T_CMP_CLASS_F64)
KERNEL64(k_cmp_ge_f64_e64, T_CMP_F64_E64)
KERNEL64(k_fma_f64, T_FMA_F64)
KERNEL64(k_mul_f64, T_MUL_F64)
KERNEL64(k_add_f64, T_ADD_F64)
KERNEL64(k_rcp_f64, T_RCP_F64)
KERNEL64(k_sqrt_f64, T_SQRT_F64)
KERNEL64(k_rsq_f64, T_RSQ_F64)
KERNEL64(k_ldexp_f64, T_LDEXP_F64)
KERNEL64(k_fract_f64, T_FRACT_F64)
KERNEL64(k_rndne_f64, T_RNDNE_F64)
KERNEL64(k_div_fixup_f64, T_DIVFIX_F64)
KERNEL64(k_max_f64, T_MINMAX_F64)
KERNEL64(k_lshlrev_b64, T_LSHL_B64)
KERNEL64(k_cmp_ge_f64, T_CMP_F64)

#define T_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define T_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define T_EXP_F32(i) "v_exp_f32 %" #i ", %" #i "\n"
#define T_RCP_F32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define T_SQRT_F32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define T_SIN_F32(i) "v_sin_f32 %" #i ", %" #i "\n"
KERNEL32(k_fma_f32, float, T_FMA_F32)
KERNEL32(k_add_f32, float, T_ADD_F32)
KERNEL32(k_mul_f32, float, T_MUL_F32)
KERNEL32(k_exp_f32, float, T_EXP_F32)
KERNEL32(k_rcp_f32, float, T_RCP_F32)
KERNEL32(k_sqrt_f32, float, T_SQRT_F32)
KERNEL32(k_sin_f32, float, T_SIN_F32)

#define T_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_ADDCO_U32(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define T_ADDC_U32(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define T_XOR_B32(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_AND_B32(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define T_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define T_LSHR_B32(i) "v_lshrrev_b32 %" #i ", 5, %" #i "\n"
#define T_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define T_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define T_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_MOV_B32(i) "v_mov_b32 %" #i ", %8\n"
#define T_MUL_LO_U32(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MUL_HI_U32(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define T_MUL_U32_U24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T_MAD_U32_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define T_MBCNT_LO(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define T_MBCNT_HI(i) "v_mbcnt_hi_u32_b32 %" #i ", %8, %" #i "\n"
#define T_CMP_U32(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define T_CVT_F64_U32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define T_ADD3_U32(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_PERM_B32(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_CNDMASK_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[4:5]\n"
#define T_CNDMASK_SRC(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define T_CMP_CNDMASK(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T_OR_B32(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define T_LSHL_B32(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
KERNEL32(k_cndmask_e64, unsigned, T_CNDMASK_E64)
KERNEL32(k_cndmask_src, unsigned, T_CNDMASK_SRC)
KERNEL32(k_cmp_cndmask, unsigned, T_CMP_CNDMASK)
KERNEL32(k_or_b32, unsigned, T_OR_B32)
KERNEL32(k_lshlrev_b32, unsigned, T_LSHL_B32)
KERNEL32(k_add_u32, unsigned, T_ADD_U32)
KERNEL32(k_add_co_u32, unsigned, T_ADDCO_U32)
KERNEL32(k_addc_co_u32, unsigned, T_ADDC_U32)
KERNEL32(k_xor_b32, unsigned, T_XOR_B32)
KERNEL32(k_and_b32, unsigned, T_AND_B32)
KERNEL32(k_bitop3_b32, unsigned, T_BITOP3)
KERNEL32(k_lshrrev_b32, unsigned, T_LSHR_B32)
KERNEL32(k_lshl_or_b32, unsigned, T_LSHL_OR)
KERNEL32(k_alignbit_b32, unsigned, T_ALIGNBIT)
KERNEL32(k_cndmask_b32, unsigned, T_CNDMASK)
KERNEL32(k_mov_b32, unsigned, T_MOV_B32)
KERNEL32(k_mul_lo_u32, unsigned, T_MUL_LO_U32)
KERNEL32(k_mul_hi_u32, unsigned, T_MUL_HI_U32)
KERNEL32(k_mul_u32_u24, unsigned, T_MUL_U32_U24)
KERNEL32(k_mad_u32_u24, unsigned, T_MAD_U32_U24)
KERNEL32(k_mbcnt_lo, unsigned, T_MBCNT_LO)
KERNEL32(k_mbcnt_hi, unsigned, T_MBCNT_HI)
KERNEL32(k_cmp_lt_u32, unsigned, T_CMP_U32)
KERNEL32(k_cvt_f32_u32, unsigned, T_CVT_F64_U32)
KERNEL32(k_add3_u32, unsigned, T_ADD3_U32)
KERNEL32(k_perm_b32, unsigned, T_PERM_B32)

#define T_MAD_U64_U32(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define T_LSHL_ADD_U64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %" #i "\n"
#define T_CVT_F64_U32B(i) "v_cvt_f64_u32 %" #i ", %8\n"
KERNEL_MAD64(k_mad_u64_u32, T_MAD_U64_U32)
KERNEL_MAD64(k_lshl_add_u64, T_LSHL_ADD_U64)
KERNEL_MAD64(k_cvt_f64_u32, T_CVT_F64_U32B)

#define T_READLANE(i) "v_readlane_b32 %" #i ", %8, 5\n"
#define T_READFIRSTLANE(i) "v_readfirstlane_b32 %" #i ", %8\n"
#define T_CMP_SGPR(i) "v_cmp_ge_f64 vcc, %9, %9\n"
KERNEL_SDST(k_readlane_b32, T_READLANE)
KERNEL_SDST(k_readfirstlane_b32, T_READFIRSTLANE)

// v_writelane: vector destination, scalar sources
__global__ __launch_bounds__(256) void k_writelane_b32(stamp *out, int iters, double seed) {
    unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    (void)seed;
    PROLOGUE()
#define T_WRITELANE(i) "v_writelane_b32 %" #i ", s4, 5\n" /* (any SGPR: the value written does not matter) */
    for (int it = 0; it < iters; ++it)
        asm volatile(BLOCK8(T_WRITELANE) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    EPILOGUE(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
}

// the mix of one Philox4x32 round as libphysicl_hip compiles it: 2 v_mad_u64_u32 + 2 v_bitop3_b32 (key adds are scalar)
__global__ __launch_bounds__(256) void k_philox_round_mix(stamp *out, int iters, double seed) {
    unsigned c0 = threadIdx.x, c1 = 1, c2 = (unsigned)seed, c3 = 3;
    unsigned long long p0 = 0, p1 = 0;
    unsigned k0 = (unsigned)seed * 3, k1 = 77;
    const unsigned m0 = 0xD2511F53u, m1 = 0xCD9E8D57u;
    PROLOGUE()
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { // 16 rounds x 4 = 64 vector instructions per trip
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %6, 0\n"
                         "v_mad_u64_u32 %1, vcc, %5, %7, 0\n"
                         : "=&v"(p0), "=&v"(p1), "+v"(c1), "+v"(c3)
                         : "v"(m0), "v"(m1), "v"(c0), "v"(c2)
                         : "vcc");
            const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
            unsigned n0, n2;
            asm volatile("v_bitop3_b32 %0, %2, %3, %4 bitop3:0x96\n"
                         "v_bitop3_b32 %1, %5, %6, %7 bitop3:0x96\n"
                         : "=&v"(n0), "=&v"(n2)
                         : "v"(hi1), "v"(c1), "s"(k0), "v"(hi0), "v"(c3), "s"(k1));
            c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
            k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
        }
    }
    EPILOGUE(c0 ^ c1 ^ c2 ^ c3)
}

typedef void (*kern_t)(stamp *, int, double);
struct probe {
    const char *name;
    const char *klass;
    kern_t fn;
    int per_trip; // vector instructions per loop trip
};

static const probe kProbes[] = {
    {"v_fma_f64", "fp64 arithmetic", k_fma_f64, 64},
    {"v_mul_f64", "fp64 arithmetic", k_mul_f64, 64},
    {"v_add_f64", "fp64 arithmetic", k_add_f64, 64},
    {"v_max_f64", "fp64 arithmetic", k_max_f64, 64},
    {"v_ldexp_f64", "fp64 arithmetic", k_ldexp_f64, 64},
    {"v_fract_f64", "fp64 arithmetic", k_fract_f64, 64},
    {"v_rndne_f64", "fp64 arithmetic", k_rndne_f64, 64},
    {"v_div_fixup_f64", "fp64 arithmetic", k_div_fixup_f64, 64},
    {"v_cmp_ge_f64", "fp64 compare", k_cmp_ge_f64, 64},
    {"v_cmp_ge_f64_e64", "fp64 compare", k_cmp_ge_f64_e64, 64},
    {"v_cmp_class_f64", "fp64 compare", k_cmp_class_f64, 64},
    {"v_fmac_f64", "fp64 arithmetic", k_fmac_f64, 64},
    {"v_mov_b64", "64-bit integer", k_mov_b64, 64},
    {"v_rcp_f64", "fp64 transcendental", k_rcp_f64, 64},
    {"v_sqrt_f64", "fp64 transcendental", k_sqrt_f64, 64},
    {"v_rsq_f64", "fp64 transcendental", k_rsq_f64, 64},
    {"v_cvt_f64_u32", "fp64 conversion", k_cvt_f64_u32, 64},
    {"v_lshlrev_b64", "64-bit integer", k_lshlrev_b64, 64},
    {"v_lshl_add_u64", "64-bit integer", k_lshl_add_u64, 64},
    {"v_mad_u64_u32", "integer multiply", k_mad_u64_u32, 64},
    {"v_mul_lo_u32", "integer multiply", k_mul_lo_u32, 64},
    {"v_mul_hi_u32", "integer multiply", k_mul_hi_u32, 64},
    {"v_mul_u32_u24", "32-bit", k_mul_u32_u24, 64},
    {"v_mad_u32_u24", "32-bit", k_mad_u32_u24, 64},
    {"v_fma_f32", "32-bit", k_fma_f32, 64},
    {"v_add_f32", "32-bit", k_add_f32, 64},
    {"v_mul_f32", "32-bit", k_mul_f32, 64},
    {"v_add_u32", "32-bit", k_add_u32, 64},
    {"v_add3_u32", "32-bit", k_add3_u32, 64},
    {"v_add_co_u32", "32-bit", k_add_co_u32, 64},
    {"v_addc_co_u32", "32-bit", k_addc_co_u32, 64},
    {"v_xor_b32", "32-bit", k_xor_b32, 64},
    {"v_and_b32", "32-bit", k_and_b32, 64},
    {"v_bitop3_b32", "32-bit", k_bitop3_b32, 64},
    {"v_lshrrev_b32", "32-bit", k_lshrrev_b32, 64},
    {"v_lshl_or_b32", "32-bit", k_lshl_or_b32, 64},
    {"v_alignbit_b32", "32-bit", k_alignbit_b32, 64},
    {"v_perm_b32", "32-bit", k_perm_b32, 64},
    {"v_cndmask_b32", "select", k_cndmask_b32, 64},
    {"v_cndmask_b32_e64", "select", k_cndmask_e64, 64},
    {"v_cndmask_b32 (dst != src)", "select", k_cndmask_src, 64},
    {"v_cmp_lt_u32 + v_cndmask", "select", k_cmp_cndmask, 128},
    {"v_or_b32", "32-bit", k_or_b32, 64},
    {"v_lshlrev_b32", "32-bit", k_lshlrev_b32, 64},
    {"v_mov_b32", "32-bit", k_mov_b32, 64},
    {"v_cmp_lt_u32", "32-bit", k_cmp_lt_u32, 64},
    {"v_mbcnt_lo_u32_b32", "32-bit", k_mbcnt_lo, 64},
    {"v_mbcnt_hi_u32_b32", "32-bit", k_mbcnt_hi, 64},
    {"v_cvt_f32_u32", "32-bit", k_cvt_f32_u32, 64},
    {"v_exp_f32", "fp32 transcendental", k_exp_f32, 64},
    {"v_rcp_f32", "fp32 transcendental", k_rcp_f32, 64},
    {"v_sqrt_f32", "fp32 transcendental", k_sqrt_f32, 64},
    {"v_sin_f32", "fp32 transcendental", k_sin_f32, 64},
    {"v_readlane_b32", "cross-lane", k_readlane_b32, 64},
    {"v_readfirstlane_b32", "cross-lane", k_readfirstlane_b32, 64},
    {"v_writelane_b32", "cross-lane", k_writelane_b32, 64},
    {"philox_round_mix", "mix: 2 v_mad_u64_u32 + 2 v_bitop3_b32 per round", k_philox_round_mix, 64},
};

static double median(std::vector<double> v) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    const char *json_path = nullptr;
    const char *only = nullptr;
    int block = 256; // --block 64: one wave per workgroup (which SIMDs of a CU are busy is then up to the dispatcher)
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json") && i + 1 < argc) json_path = argv[++i];
        if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
        if (!strcmp(argv[i], "--block") && i + 1 < argc) block = atoi(argv[++i]);
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate attribute %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const int waves_per_simd[] = {1, 2, 4, 8};
    const int max_blocks = cus * 8;
    stamp *d_out;
    CHECK(hipMalloc(&d_out, sizeof(stamp) * max_blocks * 4));
    std::vector<stamp> h(max_blocks * 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    FILE *jf = json_path ? fopen(json_path, "w") : nullptr;
    if (jf) fprintf(jf, "{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.gcnArchName, cus);
    bool first_row = true;
    printf("%-22s %2s %9s %9s %9s %8s %8s %7s %s\n", "instruction", "W", "cyc/instr", "p10", "p90", "GHz(mem)", "GHz(evt)", "ms", "waves per SIMD seen (min..max), SIMDs");
    for (const probe &p : kProbes) {
        if (only && !strstr(p.name, only)) continue;
        for (int W : waves_per_simd) {
            const int blocks = cus * W;
            // aim at ~2 ms per launch at 4 cycles per instruction and W waves: iters x 64 x 4 x W cycles
            int iters = (int)(2.0e-3 * 2.0e9 / (64.0 * 4.0 * W));
            if (iters < 200) iters = 200;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) { // the third launch is the one read (clock and caches settled)
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(block), 0, 0, d_out, iters, 1.25 + rep);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const int waves = blocks * (block / 64);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(stamp) * waves, hipMemcpyDeviceToHost));
            // group the waves by the SIMD they ran on
            struct acc {
                unsigned long long t0 = ~0ull, t1 = 0;
                int waves = 0;
            };
            std::map<unsigned long long, acc> simd;
            std::vector<double> ghz;
            for (int w = 0; w < waves; ++w) {
                const stamp &s = h[w];
                const unsigned long long key = ((unsigned long long)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xff30u);
                acc &a = simd[key];
                a.t0 = std::min(a.t0, s.t0);
                a.t1 = std::max(a.t1, s.t1);
                ++a.waves;
                if (s.w1 > s.w0) ghz.push_back((double)(s.t1 - s.t0) / (double)(s.w1 - s.w0) * 0.1);
            }
            std::vector<double> cpi;
            int wmin = 1 << 30, wmax = 0;
            const double n_instr = (double)iters * p.per_trip;
            for (auto &kv : simd) {
                cpi.push_back((double)(kv.second.t1 - kv.second.t0) / (n_instr * kv.second.waves));
                wmin = std::min(wmin, kv.second.waves);
                wmax = std::max(wmax, kv.second.waves);
            }
            std::sort(cpi.begin(), cpi.end());
            const double med = cpi[cpi.size() / 2], p10 = cpi[cpi.size() / 10], p90 = cpi[(cpi.size() * 9) / 10];
            const double clock = median(ghz);
            // event-time clock: the SIMD-cycles the median SIMD spent / wall time
            const double evt_ghz = med * n_instr * W / (ms * 1e-3) / 1e9;
            printf("%-22s %2d %9.3f %9.3f %9.3f %8.3f %8.3f %7.3f %d..%d, %zu\n", p.name, W, med, p10, p90, clock, evt_ghz, ms, wmin, wmax, simd.size());
            if (jf) {
                fprintf(jf, "%s{\"instruction\": \"%s\", \"class\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instruction\": %.4f, "
                            "\"p10\": %.4f, \"p90\": %.4f, \"clock_GHz_memtime\": %.4f, \"clock_GHz_event\": %.4f, \"launch_ms\": %.4f, "
                            "\"waves_per_simd_seen\": [%d, %d], \"simds_seen\": %zu, \"instructions_per_wave\": %.0f}",
                        first_row ? "" : ",\n", p.name, p.klass, W, med, p10, p90, clock, evt_ghz, ms, wmin, wmax, simd.size(), n_instr);
                first_row = false;
            }
        }
    }
    if (jf) {
        fprintf(jf, "\n]}\n");
        fclose(jf);
    }
    CHECK(hipFree(d_out));
    return 0;
}
