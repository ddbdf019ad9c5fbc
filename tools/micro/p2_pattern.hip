// micro-benchmark: what pass 2's MEMORY PATTERN alone costs (no FFT, no LDS).  Persistent
// 512-thread work-groups move 128 KiB tiles: reads as pass 2 reads Y (2 KiB chunks, one per
// pass-1 block, 128 KiB apart) or contiguous; writes as pass 2 writes the spectrum (128-byte lines
// 8 KiB apart, 8 lines per wave instruction) or contiguous.  The next tile's 16 loads are in
// flight while the current tile's 16 stores are issued, as in the pass.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p2_pattern tools/micro/p2_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NLD = 16, M1 = 1024, T = 16;
constexpr size_t FRAME = (size_t)1 << 20;  // complex elements (8 B) per frame

// element (8-byte) offsets inside a frame
template <int RD>
__device__ __forceinline__ size_t rd_off(int tl, int i, int tid) {
    const int idx = 2 * (i * NT + tid);
    if (RD == 0) return (size_t)(idx >> 8) * (M1 * T) + (size_t)tl * 256 + (idx & 255);  // block j, chunk tl
    return (size_t)tl * (M1 * T) + idx;                                                     // contiguous
}
template <int WR>
__device__ __forceinline__ size_t wr_off(int tl, int s, int tid) {
    const int p = tid & 7, i0 = tid >> 3;
    if (WR == 0) return ((size_t)(i0 + 64 * s) << 10) + tl * T + 2 * p;  // line of 16 bins, 8 KiB stride
    if (WR == 2) {  // 256-byte runs (two tiles' worth of columns side by side): 4 runs per wave
        const int q = tid & 15, j0 = tid >> 4;
        return ((size_t)(j0 + 32 * s) << 10) + (tl >> 1) * 2 * T + 2 * q + ((size_t)(tl & 1) << 19);
    }
    return (size_t)tl * (M1 * T) + 2 * (s * NT + tid);                   // contiguous
}

template <int RD, int WR>
__global__ __launch_bounds__(NT) void k_move(const float2 *src, float2 *dst, unsigned total, unsigned *ticket) {
    const int tid = threadIdx.x;
    __shared__ unsigned s_next;
    f4 r[NLD], q[NLD];
    unsigned s = blockIdx.x;
    auto issue = [&](unsigned slot) {
        const unsigned f = slot >> 6, tl = slot & 63;
#pragma unroll
        for (int i = 0; i < NLD; i++) r[i] = *reinterpret_cast<const f4 *>(src + f * FRAME + rd_off<RD>(tl, i, tid));
    };
    if (s < total) issue(s);
    while (s < total) {
        if (tid == 0) s_next = atomicAdd(ticket, 1u) + gridDim.x;
#pragma unroll
        for (int i = 0; i < NLD; i++) q[i] = r[i];
        __syncthreads();
        const unsigned sn = s_next;
        __syncthreads();
        if (sn < total) issue(sn);
        const unsigned f = s >> 6, tl = s & 63;
#pragma unroll
        for (int i = 0; i < NLD; i++) *reinterpret_cast<f4 *>(dst + f * FRAME + wr_off<WR>(tl, i, tid)) = q[i];
        s = sn;
    }
}

int main() {
    const int F = 64;
    const size_t bytes = F * FRAME * 8;
    float2 *src, *dst; unsigned *tk;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&tk, 4));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *rdn[2] = {"chunks", "contig"}, *wrn[3] = {"lines128", "contig", "runs256"};
    for (int rd = 0; rd < 2; rd++) for (int wr = 0; wr < 3; wr++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemsetAsync(tk, 0, 4));
            CK(hipEventRecord(a));
#define L(R, W) if (rd == R && wr == W) hipLaunchKernelGGL((k_move<R, W>), dim3(256), dim3(NT), 0, 0, src, dst, 64u * F, tk);
            L(0, 0) L(0, 1) L(0, 2) L(1, 0) L(1, 1) L(1, 2)
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        printf("read %-7s write %-8s %7.1f us  %5.2f TB/s (read+write)  %5.2f us/tile/WG\n", rdn[rd], wrn[wr], best * 1e3,
               2.0 * bytes / best / 1e9, best * 1e3 / (64.0 * F / 256));
    }
    return 0;
}
