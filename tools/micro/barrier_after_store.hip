// What does __syncthreads() wait for after global stores on gfx950?   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only
// Answer (ROCm 7.2 compiler): `s_waitcnt lgkmcnt(0); s_barrier` - the LDS traffic, not the stores (no vmcnt(0)): in the
// non-tgsplit mode the waves of a workgroup share their CU's L1, which keeps their vector memory operations in order, so a
// load that another wave issues after the barrier sees the store.  clean_kernel's remap_for_target relies on exactly the
// language-level guarantee (records stored by one wave, read back by another after __syncthreads()); this probe only shows
// that the barrier does not pay a store acknowledgement for it.
#include <hip/hip_runtime.h>
// does a workgroup barrier after global stores wait for the stores (s_waitcnt vmcnt(0)) on gfx950?
__global__ void k(unsigned long long* dst, const unsigned long long* src, int n, unsigned* out)
{
    __shared__ unsigned c;
    const int t = threadIdx.x;
    if (t == 0) c = 0;
    __syncthreads();
    if (t < n) { dst[t] = src[t] + 1; atomicAdd(&c, 1u); }
    __syncthreads();
    if (t == 0) out[blockIdx.x] = c;
    __syncthreads();
    out[64 + t] = (unsigned)dst[(t + 1) & 255];
}
