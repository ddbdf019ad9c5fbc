// micro-benchmark: how fast can 256 persistent work-groups read column tiles
// (SEG bytes wide, ROWS rows, row stride STRIDE bytes) out of a big matrix?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// each work-group (512 threads) reads, per tile, ROWS row-segments of SEG bytes; lanes cover a
// segment with VEC-byte loads; tiles are laid out like pass 1: tile tl of frame f at byte
// offset f*FRAME + tl*SEG, rows STRIDE bytes apart.
template <int VEC>
__global__ __launch_bounds__(512) void k_read(const unsigned char* base, size_t frame_bytes, int stride, int seg,
                                              int rows, int tiles_per_frame, int total, unsigned* sink) {
    const int lanes_per_seg = seg / VEC;
    const int rows_per_pass = 512 / lanes_per_seg;
    const int lane = threadIdx.x % lanes_per_seg, r0 = threadIdx.x / lanes_per_seg;
    unsigned acc = 0;
    for (int s = blockIdx.x; s < total; s += gridDim.x) {
        const int f = s / tiles_per_frame, tl = s % tiles_per_frame;
        const unsigned char* p = base + (size_t)f * frame_bytes + (size_t)tl * seg + lane * VEC;
        for (int r = r0; r < rows; r += rows_per_pass) {
            const unsigned char* q = p + (size_t)r * stride;
            if (VEC == 4) acc += *reinterpret_cast<const unsigned*>(q);
            else if (VEC == 8) { uint2 v = *reinterpret_cast<const uint2*>(q); acc += v.x ^ v.y; }
            else { uint4 v = *reinterpret_cast<const uint4*>(q); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)600 << 20;
    unsigned char* d; unsigned* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Cfg { int stride, seg, rows, vec; const char* name; };
    std::vector<Cfg> cfgs = {
        {4096, 64, 1024, 4, "pass1 cs16: 64B seg, 4KB stride, dword"},
        {4096, 128, 1024, 8, "128B seg, 4KB stride, dwordx2"},
        {4096, 256, 1024, 16, "256B seg, 4KB stride, dwordx4"},
        {4096, 512, 1024, 16, "512B seg, 4KB stride"},
        {4096, 4096, 64, 16, "full 4KB rows (contiguous 256KB tiles)"},
        {4096 + 64, 64, 1024, 4, "64B seg, 4KB+64 stride"},
        {8192, 128, 1024, 8, "128B seg, 8KB stride (f32 input)"},
        {8192, 128, 1024, 16, "128B seg, 8KB stride, dwordx4"},
        {2048, 64, 1024, 4, "64B seg, 2KB stride"},
    };
    for (auto& c : cfgs) {
        const int tiles_per_frame = 4096 / c.seg > 0 ? (c.stride >= 4096 ? (c.stride / c.seg > 64 ? 64 : c.stride / c.seg) : c.stride / c.seg) : 1;
        const size_t tile_bytes = (size_t)c.seg * c.rows;
        const size_t frame_bytes = (size_t)c.stride * c.rows;      // one "frame" = rows x stride
        const int nframes = (int)(((size_t)512 << 20) / frame_bytes);
        const int total = nframes * tiles_per_frame;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a));
            if (c.vec == 4) hipLaunchKernelGGL(k_read<4>, dim3(256), dim3(512), 0, 0, d, frame_bytes, c.stride, c.seg, c.rows, tiles_per_frame, total, sink);
            else if (c.vec == 8) hipLaunchKernelGGL(k_read<8>, dim3(256), dim3(512), 0, 0, d, frame_bytes, c.stride, c.seg, c.rows, tiles_per_frame, total, sink);
            else hipLaunchKernelGGL(k_read<16>, dim3(256), dim3(512), 0, 0, d, frame_bytes, c.stride, c.seg, c.rows, tiles_per_frame, total, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("%-48s tiles=%5d bytes=%6.1f MB  %7.1f us  %6.2f TB/s\n", c.name, total, total * tile_bytes / 1e6, ms * 1e3, total * tile_bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
