// valu_peak.hip - what the vector unit sustains for the integer / bit instructions the join kernel is made of, at the
// residency the join has (one 1024-thread workgroup per CU holding the whole LDS = 4 waves per SIMD) and at 1, 2 and 8
// waves per SIMD.  Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(1024) loop_kernel(uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    uint32_t a[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) a[x] = seed * (x + 1) + threadIdx.x;
    uint32_t s = seed | 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            if (KIND == 0) a[x] = __builtin_amdgcn_alignbit(a[x], a[(x + 1) & 15], 7) ;          // v_alignbit_b32
            else if (KIND == 1) a[x] = (a[x] & 0x0F0F0F0Fu) + s;                                  // v_and_b32 + v_add_u32
            else if (KIND == 2) a[x] = (a[x] >> 3) ^ a[(x + 5) & 15];                             // v_lshrrev + v_xor
            else if (KIND == 3) a[x] = a[x] * 0x9E3779B1u + s;                                    // v_mul_lo_u32 (+ add)
            else if (KIND == 4) a[x] = __umul24(a[x], 0xC2B2AFu) + s;                             // v_mul_u32_u24 (+ add / mad)
            else if (KIND == 5) a[x] = __popc(a[x]) + a[(x + 1) & 15];                            // v_bcnt_u32_b32 (accumulating)
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int x = 0; x < 16; ++x) r ^= a[x];
    if (r == 0x12345678u) out[blockIdx.x] = r + lds[threadIdx.x & 7];
}

template <int KIND>
static void run(const char* name, int ops_per_elem, int threads, size_t lds_bytes, int blocks_per_cu)
{
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    uint32_t* d;
    hipMalloc(&d, 4096 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop_kernel<KIND>, dim3(cus * blocks_per_cu), dim3(threads), lds_bytes, 0, d, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = (double)threads / 64.0 * blocks_per_cu / 4.0;
    const double winstr_per_simd = (double)iters * 16 * ops_per_elem * waves_per_simd;     // wave-instructions one SIMD issued
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s %4d thr x %d/CU (%.0f waves/SIMD): %7.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
           name, threads, blocks_per_cu, waves_per_simd, ms, cycles / winstr_per_simd);
    hipFree(d);
}

int main()
{
    const size_t big = 159 * 1024;
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int threads = cfg == 0 ? 256 : cfg == 1 ? 512 : 1024;
        const int per_cu = cfg == 3 ? 2 : 1;
        const size_t lds = cfg == 3 ? 1024 : big;
        run<0>("v_alignbit_b32", 1, threads, lds, per_cu);
        run<1>("v_and_b32 + v_add_u32", 2, threads, lds, per_cu);
        run<2>("v_lshrrev_b32 + v_xor_b32", 2, threads, lds, per_cu);
        run<3>("v_mul_lo_u32 + v_add_u32", 2, threads, lds, per_cu);
        run<4>("v_mul_u32_u24 + add (v_mad_u32_u24)", 1, threads, lds, per_cu);
        run<5>("v_bcnt_u32_b32", 1, threads, lds, per_cu);
    }
    return 0;
}
