// valu_ops.hip - issue cost of single vector opcodes on gfx950, one opcode per kernel (inline asm, 16 independent
// registers per lane so nothing waits on its own result), at the join kernel's residency (4 waves per SIMD) and at 8.
// Prints cycles per wave64 instruction per SIMD at a nominal 2.4 GHz.  tools/isa_cost.py prices a kernel's ISA with it.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM, ...)                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(uint32_t* out, int iters, uint32_t seed)                               \
    {                                                                                                                   \
        uint32_t a[16];                                                                                                 \
        _Pragma("unroll") for (int x = 0; x < 16; ++x) a[x] = seed * (x + 1) + threadIdx.x;                             \
        uint32_t s = seed | 1u;                                                                                         \
        uint64_t m = 0x5555555555555555ull ^ seed;                                                                      \
        asm volatile("" : "+s"(s), "+s"(m));                                                                            \
        for (int it = 0; it < iters; ++it) {                                                                            \
            _Pragma("unroll") for (int x = 0; x < 16; ++x) {                                                            \
                uint32_t& d = a[x];                                                                                     \
                const uint32_t b = a[(x + 5) & 15], c = a[(x + 9) & 15];                                                \
                (void)b; (void)c;                                                                                       \
                asm volatile(ASM : "+v"(d) : "v"(b), "v"(c), "s"(s), "s"(m) : __VA_ARGS__);                             \
            }                                                                                                           \
        }                                                                                                               \
        uint32_t r = 0;                                                                                                 \
        _Pragma("unroll") for (int x = 0; x < 16; ++x) r ^= a[x];                                                       \
        if (r == 0x12345678u) out[blockIdx.x] = r;                                                                      \
    }

// %0 = destination / first source (VGPR), %1 %2 = other VGPRs, %3 = SGPR, %4 = SGPR pair
KERNEL(k_mov,        "v_mov_b32 %0, %1", "memory")
KERNEL(k_and,        "v_and_b32 %0, %0, %1", "memory")
KERNEL(k_and_e64,    "v_and_b32_e64 %0, %0, %1", "memory")
KERNEL(k_and_lit,    "v_and_b32 %0, 0x0f0f0f0f, %0", "memory")
KERNEL(k_and_sgpr,   "v_and_b32 %0, %3, %0", "memory")
KERNEL(k_or,         "v_or_b32 %0, %0, %1", "memory")
KERNEL(k_xor,        "v_xor_b32 %0, %0, %1", "memory")
KERNEL(k_add,        "v_add_u32 %0, %0, %1", "memory")
KERNEL(k_sub,        "v_sub_u32 %0, %0, %1", "memory")
KERNEL(k_addco,      "v_add_co_u32 %0, vcc, %0, %1", "memory", "vcc")
KERNEL(k_min,        "v_min_u32 %0, %0, %1", "memory")
KERNEL(k_lshl,       "v_lshlrev_b32 %0, 3, %0", "memory")
KERNEL(k_lshr,       "v_lshrrev_b32 %0, 3, %0", "memory")
KERNEL(k_lshr_v,     "v_lshrrev_b32 %0, %1, %0", "memory")
KERNEL(k_ashr,       "v_ashrrev_i32 %0, 3, %0", "memory")
KERNEL(k_mul24,      "v_mul_u32_u24 %0, %0, %1", "memory")
KERNEL(k_mullo,      "v_mul_lo_u32 %0, %0, %1", "memory")
KERNEL(k_mad24,      "v_mad_u32_u24 %0, %0, %1, %2", "memory")
KERNEL(k_cndmask,    "v_cndmask_b32 %0, %0, %1, vcc", "memory")
KERNEL(k_cndmask64,  "v_cndmask_b32_e64 %0, %0, %1, %4", "memory")
KERNEL(k_cmp,        "v_cmp_eq_u32 vcc, %0, %1", "memory", "vcc")
KERNEL(k_cmp64,      "v_cmp_eq_u32_e64 s[20:21], %0, %1", "memory", "s20", "s21")
KERNEL(k_cmp_lt_lit, "v_cmp_gt_u32 vcc, 0x12345, %0", "memory", "vcc")
KERNEL(k_bfe,        "v_bfe_u32 %0, %0, 3, 7", "memory")
KERNEL(k_bfi,        "v_bfi_b32 %0, %3, %0, %1", "memory")
KERNEL(k_lshl_or,    "v_lshl_or_b32 %0, %0, 3, %1", "memory")
KERNEL(k_lshl_add,   "v_lshl_add_u32 %0, %0, 3, %1", "memory")
KERNEL(k_add_lshl,   "v_add_lshl_u32 %0, %0, %1, 3", "memory")
KERNEL(k_and_or,     "v_and_or_b32 %0, %0, %1, %2", "memory")
KERNEL(k_or3,        "v_or3_b32 %0, %0, %1, %2", "memory")
KERNEL(k_add3,       "v_add3_u32 %0, %0, %1, %2", "memory")
KERNEL(k_xad,        "v_xad_u32 %0, %0, %1, %2", "memory")
KERNEL(k_bitop3,     "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "memory")
KERNEL(k_alignbit,   "v_alignbit_b32 %0, %0, %1, 7", "memory")
KERNEL(k_alignbit_v, "v_alignbit_b32 %0, %0, %1, %2", "memory")
KERNEL(k_perm,       "v_perm_b32 %0, %0, %1, %3", "memory")
KERNEL(k_bcnt,       "v_bcnt_u32_b32 %0, %0, %1", "memory")
KERNEL(k_mbcnt_lo,   "v_mbcnt_lo_u32_b32 %0, %3, %0", "memory")
KERNEL(k_mbcnt_hi,   "v_mbcnt_hi_u32_b32 %0, %3, %0", "memory")
KERNEL(k_ffbl,       "v_ffbl_b32 %0, %0", "memory")
KERNEL(k_ffbh,       "v_ffbh_u32 %0, %0", "memory")
KERNEL(k_bfrev,      "v_bfrev_b32 %0, %0", "memory")
KERNEL(k_readlane,   "v_readlane_b32 s20, %0, 5", "memory", "s20")
KERNEL(k_readfirst,  "v_readfirstlane_b32 s20, %0", "memory", "s20")
KERNEL(k_dpp_shr,    "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "memory")
KERNEL(k_add_dpp,    "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "memory")
KERNEL(k_add_sdwa,   "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1", "memory")
KERNEL(k_pk_add16,   "v_pk_add_u16 %0, %0, %1", "memory")
KERNEL(k_swap_nop,   "s_nop 0", "memory")

typedef void (*kern_t)(uint32_t*, int, uint32_t);

static void run(const char* name, kern_t k, int per_cu, uint32_t* d, int cus)
{
    const int iters = 8000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus * per_cu), dim3(1024), 0, 0, d, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double waves_per_simd = 4.0 * per_cu;
    const double winstr = (double)iters * 16 * waves_per_simd;
    printf("%-22s %d waves/SIMD  %7.3f ms  %.2f cycles\n", name, (int)waves_per_simd, best, best * 1e-3 * 2.4e9 / winstr);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    uint32_t* d;
    hipMalloc(&d, 8192 * 4);
#define R(K) run(#K, K, per_cu, d, cus)
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        R(k_mov); R(k_and); R(k_and_e64); R(k_and_lit); R(k_and_sgpr); R(k_or); R(k_xor); R(k_add); R(k_sub); R(k_addco); R(k_min);
        R(k_lshl); R(k_lshr); R(k_lshr_v); R(k_ashr); R(k_mul24); R(k_mullo); R(k_mad24);
        R(k_cndmask); R(k_cndmask64); R(k_cmp); R(k_cmp64); R(k_cmp_lt_lit);
        R(k_bfe); R(k_bfi); R(k_lshl_or); R(k_lshl_add); R(k_add_lshl); R(k_and_or); R(k_or3); R(k_add3); R(k_xad); R(k_bitop3);
        R(k_alignbit); R(k_alignbit_v); R(k_perm); R(k_bcnt); R(k_mbcnt_lo); R(k_mbcnt_hi); R(k_ffbl); R(k_ffbh); R(k_bfrev);
        R(k_readlane); R(k_readfirst); R(k_dpp_shr); R(k_add_dpp); R(k_add_sdwa); R(k_pk_add16); R(k_swap_nop);
    }
    hipFree(d);
    return 0;
}
