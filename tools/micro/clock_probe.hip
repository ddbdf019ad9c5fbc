// Shader clock under different loads: s_memtime ticks (shader cycles) per wall-clock nanosecond for (a) a pure
// VALU kernel, (b) a streaming copy, on the whole chip.  hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_valu(float *out, unsigned long long *t, int iters) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) a = fmaf(a, b, c);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = c1 - c0;
        t[blockIdx.x * 2 + 1] = w1 - w0;
    }
}
__global__ void k_copy(const float4 *in, float4 *out, size_t n, unsigned long long *t) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = c1 - c0;
        t[blockIdx.x * 2 + 1] = w1 - w0;
    }
}
int main(int argc, char **argv) {
    const int secs = argc > 2 ? atoi(argv[2]) : 0;  // clock_probe valu|copy <seconds>: keep one load running (for rocm-smi --showpower)
    const char *mode = argc > 1 ? argv[1] : "";
    float *out;
    unsigned long long *t;
    const int G = 1024;
    hipMalloc(&out, G * 256 * sizeof(float));
    hipMalloc(&t, G * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(G * 2);
    auto report = [&](const char *name) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        double c = 0, w = 0;
        for (int i = 0; i < G; i++) c += h[2 * i], w += h[2 * i + 1];
        printf("%s: %.3f shader ticks per ns (wall clock 100 MHz), mean block time %.1f us\n", name, c / (w * 10.0), w / G / 100.0);
    };
    for (int rep = 0; rep < (mode[0] == 'v' ? secs * 170 : mode[0] ? 0 : 3); rep++) {
        hipLaunchKernelGGL(k_valu, dim3(G), dim3(256), 0, 0, out, t, 20000);
        if (!mode[0] || rep % 170 == 0) report("pure VALU (fma chain)");
    }
    const size_t n = (size_t)1 << 27;  // 2 GiB in, 2 GiB out
    float4 *a, *b;
    hipMalloc(&a, n * 16);
    hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    for (int rep = 0; rep < (mode[0] == 'c' ? secs * 1400 : mode[0] ? 0 : 3); rep++) {
        hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, a, b, n, t);
        if (!mode[0] || rep % 1400 == 0) report("streaming copy");
    }
    return 0;
}
