// micro-benchmark: ceiling of streaming WRITES (and read+write copies) on this chip for the
// shapes the FFT passes use: persistent work-groups, 128 KiB contiguous blocks, 16 B per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: plain store, 1: nontemporal store, 2: copy (load + store), 3: read only
template <int MODE>
__global__ void k_stream(f4* dst, const f4* src, size_t blk_elems, int nblk, float* sink) {
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    f4 acc = {0, 0, 0, 0};
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        f4* d = dst + (size_t)b * blk_elems;
        const f4* s = src + (size_t)b * blk_elems;
#pragma unroll 4
        for (size_t i = threadIdx.x; i < blk_elems; i += blockDim.x) {
            if (MODE == 0) d[i] = v;
            else if (MODE == 1) __builtin_nontemporal_store(v, &d[i]);
            else if (MODE == 2) d[i] = s[i];
            else acc += s[i];
        }
    }
    if (MODE == 3 && acc.x == 1.2345f) sink[0] = acc.y;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    f4 *d, *s; float* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&s, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 0, bytes)); CK(hipMemset(s, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[4] = {"store", "nt-store", "copy", "read"};
    struct G { int grid, threads; size_t blk; };
    G gs[] = {{256, 512, 128 << 10}, {256, 1024, 128 << 10}, {512, 512, 128 << 10}, {1024, 256, 128 << 10},
              {2048, 256, 64 << 10}, {8192, 256, 128 << 10}, {256, 512, 1 << 20}};
    for (size_t total : {(size_t)128 << 20, (size_t)1 << 30})
    for (auto g : gs) for (int mode = 0; mode < 4; mode++) {
        const int nblk = (int)(total / g.blk);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            const size_t be = g.blk / 16;
            if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(g.grid), dim3(g.threads), 0, 0, d, s, be, nblk, sink);
            if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(g.grid), dim3(g.threads), 0, 0, d, s, be, nblk, sink);
            if (mode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(g.grid), dim3(g.threads), 0, 0, d, s, be, nblk, sink);
            if (mode == 3) hipLaunchKernelGGL(k_stream<3>, dim3(g.grid), dim3(g.threads), 0, 0, d, s, be, nblk, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        const double moved = (mode == 2 ? 2.0 : 1.0) * total;
        printf("total %4zu MiB grid %5d x %4d blk %4zu KiB %-8s %7.1f us  %6.2f TB/s\n", total >> 20, g.grid, g.threads,
               g.blk >> 10, names[mode], best * 1e3, moved / best / 1e9);
    }
    return 0;
}
