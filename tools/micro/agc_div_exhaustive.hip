// Is a short reciprocal-and-correct sequence the correctly rounded quotient desired / x for EVERY x the post chain's AGC can
// see?  (w_t = desired / (peak_t + 1e-10f), desired = 0.2f: src/utils/audioprocessing.cpp:5-16, 55-66; x >= 1e-10f.)
// Every float bit pattern of x in [1e-10f, hi] against __fdiv_rn - the reference's IEEE division - for three candidates:
//   A  r = rcp(x); q = n r; q += r (n - x q)
//   B  A with r refined once first:  r += r (1 - x r)
//   C  B with a second residual step
// hipcc --offload-arch=gfx950 -O3 agc_div_exhaustive.hip -o agc_div_exhaustive ; ./agc_div_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float cand_a(float n, float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float q = __fmul_rn(n, r);
    return __fmaf_rn(__fmaf_rn(-x, q, n), r, q);
}
__device__ __forceinline__ float cand_b(float n, float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = __fmaf_rn(__fmaf_rn(-x, r, 1.0f), r, r);
    const float q = __fmul_rn(n, r);
    return __fmaf_rn(__fmaf_rn(-x, q, n), r, q);
}
__device__ __forceinline__ float cand_c(float n, float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = __fmaf_rn(__fmaf_rn(-x, r, 1.0f), r, r);
    float q = __fmul_rn(n, r);
    q = __fmaf_rn(__fmaf_rn(-x, q, n), r, q);
    return __fmaf_rn(__fmaf_rn(-x, q, n), r, q);
}
__global__ void k_check(float n, unsigned lo, unsigned hi, unsigned long long *bad, unsigned *first) {
    unsigned long long ba = 0, bb = 0, bc = 0;
    for (unsigned long long b = (unsigned long long)lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const unsigned want = __float_as_uint(__fdiv_rn(n, x));
        if (__float_as_uint(cand_a(n, x)) != want) { if (!ba) atomicMin(first + 0, (unsigned)b); ba++; }
        if (__float_as_uint(cand_b(n, x)) != want) { if (!bb) atomicMin(first + 1, (unsigned)b); bb++; }
        if (__float_as_uint(cand_c(n, x)) != want) { if (!bc) atomicMin(first + 2, (unsigned)b); bc++; }
    }
    if (ba) atomicAdd(bad + 0, ba);
    if (bb) atomicAdd(bad + 1, bb);
    if (bc) atomicAdd(bad + 2, bc);
}
int main() {
    unsigned long long *bad; unsigned *first;
    hipMalloc(&bad, 24); hipMalloc(&first, 12);
    const float lo_f = 1e-10f;
    for (float hi_f : {7.9e28f, 1e36f, 3.4028235e38f}) {
        unsigned lo, hi; memcpy(&lo, &lo_f, 4); memcpy(&hi, &hi_f, 4);
        hipMemset(bad, 0, 24); hipMemset(first, 0xff, 12);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 0.2f, lo, hi, bad, first);
        unsigned long long h[3]; unsigned f[3];
        hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); hipMemcpy(f, first, 12, hipMemcpyDeviceToHost);
        printf("{\"numerator\": 0.2, \"x_from\": %.9g, \"x_to\": %.9g, \"values\": %u, \"mismatches_A\": %llu, \"mismatches_B\": %llu, \"mismatches_C\": %llu, \"first_A\": \"0x%08x\", \"first_B\": \"0x%08x\", \"first_C\": \"0x%08x\"}\n",
               lo_f, hi_f, hi - lo + 1, h[0], h[1], h[2], f[0], f[1], f[2]);
    }
    return 0;
}
