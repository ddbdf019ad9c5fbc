// fft_pass.h — the two LDS-tiled Stockham passes of the large forward FFT (gfx950).
//
// Replaces, for the HIP back-end, what the reference does with
//   window_real/window_complex + cufftExecC2C/R2C        (src/fft_cuda.cu:62-99,132-138)
//   dsp_multiply_* + fftwf_execute                       (src/fft_impl.cpp:119-145)
//   convert<T,T_signed>                                  (src/samplereader.cpp:29-40)
//   power_and_quantize / half_and_quantize (IQ, fused)   (src/fft_impl.cpp:24-61,144-172)
//
// Four-step decomposition of the M-point complex transform, M = M1*M2:
//   n = M2*n1 + n2, output bin c = c1 + M1*c2
//   pass 1 (columns): Y[c1][n2] = tw(n2,c1) * sum_n1 v[M2*n1+n2] W_M1^{n1*k1}
//   pass 2 (rows)   : X[c1 + M1*c2] = sum_n2 Y[c1][n2] W_M2^{n2*c2}
// For IQ input the output is produced directly in CLIENT order c (reference bin
// k = (c + N/2 + 1) mod N, src/fft_impl.cpp:149-160): pass 1 stores bin k1 at row
// c1 = (k1-1) mod M1 with twiddle (-1)^{n2} W_M^{n2*(c1+1)}; no input modulation, so
// the arithmetic is that of an ordinary FFT.
//
// Work-group = one tile of T sequences x L points (L*T = 16K complex, 128 KiB LDS),
// (L/16)*T threads; every thread owns the 16 points {i0 + e*L/16} of one sequence at
// every stage and performs 16/R radix-R butterflies per stage (R in {16,8,4,2}); lanes
// run along the T (contiguous-in-HBM) dimension, so stage traffic in LDS is
// conflict-free and HBM segments are T*8 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "butterfly.h"
#include "quantize.h"

#ifndef PSDR_ABL
#define PSDR_ABL 0  // ablation bitmask, tuning builds only (tools/ablate.sh)
#endif

namespace psdr {

template <int L>
struct Plan;
template <>
struct Plan<64> {
    static constexpr int NS = 2, R0 = 16, R1 = 4, R2 = 1;
};
template <>
struct Plan<128> {
    static constexpr int NS = 2, R0 = 16, R1 = 8, R2 = 1;
};
template <>
struct Plan<256> {
    static constexpr int NS = 2, R0 = 16, R1 = 16, R2 = 1;
};
template <>
struct Plan<512> {
    static constexpr int NS = 3, R0 = 16, R1 = 8, R2 = 4;
};
template <>
struct Plan<1024> {
    static constexpr int NS = 3, R0 = 16, R1 = 16, R2 = 4;
};
template <>
struct Plan<2048> {
    static constexpr int NS = 3, R0 = 16, R1 = 16, R2 = 8;
};

template <int R>
__device__ __forceinline__ void dftR(cf (&x)[R]);
template <>
__device__ __forceinline__ void dftR<2>(cf (&x)[2]) {
    dft2(x[0], x[1]);
}
template <>
__device__ __forceinline__ void dftR<4>(cf (&x)[4]) {
    dft4(x[0], x[1], x[2], x[3]);
}
template <>
__device__ __forceinline__ void dftR<8>(cf (&x)[8]) {
    dft8(x);
}
template <>
__device__ __forceinline__ void dftR<16>(cf (&x)[16]) {
    dft16(x);
}

// One Stockham stage on the 16 points a thread owns.  P = product of earlier radices.
//   butterfly i = i0 + b*L/16, k = i mod P, j = (i-k)*R + k
//   x_q = u[b + q*16/R] * W_L^{q*k*L/(P*R)};  out[j + s*P] = DFT_R(x)[s]
template <int L, int R, int P, typename Emit>
__device__ __forceinline__ void stage_compute(cf (&u)[16], int i0, const cf *Wl, Emit emit) {
    constexpr int NB = 16 / R;
    constexpr int L16 = L / 16;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int i = i0 + b * L16;
        const int k = i & (P - 1);
        const int j = (i - k) * R + k;
        cf x[R];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = u[b + q * NB];
        if (P > 1) {
            const int step = k * (L / (P * R));
#pragma unroll
            for (int q = 1; q < R; q++) x[q] = cmul(x[q], Wl[q * step]);
        }
        dftR<R>(x);
#pragma unroll
        for (int s = 0; s < R; s++) emit(j + s * P, x[s]);
    }
}

// LDS element index of (row = position in the sequence, col = sequence in the tile).
// SWZ spreads the transposing store of pass 2 over the banks.
template <int T, bool SWZ>
__device__ __forceinline__ int lds_idx(int row, int col) {
    return row * T + (SWZ ? (col ^ (row & (T - 1))) : col);
}

template <int L, int T, bool SWZ>
__device__ __forceinline__ void tile_read(cf (&u)[16], const cf *tile, int i0, int t) {
#pragma unroll
    for (int e = 0; e < 16; e++) u[e] = tile[lds_idx<T, SWZ>(i0 + e * (L / 16), t)];
}

// Runs stages 0..NS-1 on data already in `u` (stage 0 input), exchanging through the
// LDS tile between stages; the last stage's outputs go to emit_last(pos, value).
template <int L, int T, bool SWZ, typename EmitLast>
__device__ __forceinline__ void run_stages(cf (&u)[16], cf *tile, const cf *Wl, int i0, int t,
                                           EmitLast emit_last) {
    using P = Plan<L>;
    auto to_lds = [&](int pos, cf v) { tile[lds_idx<T, SWZ>(pos, t)] = v; };
    // stage 0
    stage_compute<L, P::R0, 1>(u, i0, Wl, to_lds);
    __syncthreads();
    tile_read<L, T, SWZ>(u, tile, i0, t);
    if constexpr (P::NS == 2) {
        __syncthreads();  // everyone has read before the tile is reused by the caller
        stage_compute<L, P::R1, P::R0>(u, i0, Wl, emit_last);
    } else {
        __syncthreads();
        stage_compute<L, P::R1, P::R0>(u, i0, Wl, to_lds);
        __syncthreads();
        tile_read<L, T, SWZ>(u, tile, i0, t);
        __syncthreads();
        stage_compute<L, P::R2, P::R0 * P::R1>(u, i0, Wl, emit_last);
    }
}

// XCD-aware slot mapping: work-group b runs on XCD b%8 (observed; used for speed only).
// Groups of 8 adjacent tiles (one 128-byte line of int8 output, 1 KiB of spectrum) are
// given to one XCD back to back so their partial lines merge in that XCD's L2.
__device__ __forceinline__ unsigned xcd_slot(unsigned bid, unsigned total) {
    if (total & 63u) return bid;
    const unsigned x = bid & 7u, y = bid >> 3;
    return ((x + 8u * (y >> 3)) << 3) + (y & 7u);
}

// raw sample pair -> float2 (src/samplereader.cpp:29-40): unsigned formats flip the MSB,
// integers are divided by 2^(bits-1) (exact, so multiply by the reciprocal).
__device__ __forceinline__ cf load_raw_pair(const void *raw, size_t idx, int fmt) {
    switch (fmt) {
    case 0: {  // u8
        const uchar2 v = reinterpret_cast<const uchar2 *>(raw)[idx];
        return make_float2((float)(int8_t)(v.x ^ 0x80) * (1.0f / 128.0f),
                           (float)(int8_t)(v.y ^ 0x80) * (1.0f / 128.0f));
    }
    case 1: {  // s8
        const char2 v = reinterpret_cast<const char2 *>(raw)[idx];
        return make_float2((float)v.x * (1.0f / 128.0f), (float)v.y * (1.0f / 128.0f));
    }
    case 2: {  // u16
        const ushort2 v = reinterpret_cast<const ushort2 *>(raw)[idx];
        return make_float2((float)(int16_t)(v.x ^ 0x8000) * (1.0f / 32768.0f),
                           (float)(int16_t)(v.y ^ 0x8000) * (1.0f / 32768.0f));
    }
    case 3: {  // s16
        const short2 v = reinterpret_cast<const short2 *>(raw)[idx];
        return make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
    case 4: {  // f32
        return reinterpret_cast<const float2 *>(raw)[idx];
    }
    default: {  // f64
        const double2 v = reinterpret_cast<const double2 *>(raw)[idx];
        return make_float2((float)v.x, (float)v.y);
    }
    }
}

struct Pass1Args {
    const void *raw;      // nframes+1 raw half-frames, contiguous
    const float *window;  // N floats (Hann, src/utils/dsp.cpp:6-11)
    cf *Y;                // [nframes][M1][M2]
    const cf *Wl;         // W_L^j, j < L (L = M1)
    const cf *TA;         // W_M^{h*B}
    const cf *TB;         // W_M^{l}, l < B
    int log2B;
    int M2;
    int log2M2;
    int fmt;
    int is_real;  // window pairs (w[2n], w[2n+1]) instead of (w[n], w[n])
    int rot;      // IQ: produce client order
    unsigned tiles_per_frame;
    unsigned total_slots;
};

// pass 1: convert + window + column FFT (length L = M1) + inter-pass twiddle
template <int L, int T>
__global__ __launch_bounds__((L / 16) * T) void k_fft_pass1(Pass1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf *tile = reinterpret_cast<cf *>(smem);
    cf *Wl = tile + L * T;
    constexpr int NT = (L / 16) * T;
    const int tid = threadIdx.x;
    const unsigned slot = xcd_slot(blockIdx.x, a.total_slots);
    const unsigned f = slot / a.tiles_per_frame;
    const unsigned tl = slot - f * a.tiles_per_frame;
    const int t = tid % T, i0 = tid / T;
    const int M2 = a.M2;
    const size_t M = (size_t)L << a.log2M2;
    const int n2 = tl * T + t;

    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];

    // frame f = M complex samples starting at complex index f*M/2 of the raw stream
    const size_t base = (size_t)f * (M / 2);
    cf u[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const size_t n = (size_t)(i0 + e * (L / 16)) * M2 + n2;
        cf v = (PSDR_ABL & 8) ? make_float2(1.f + n, 2.f) : load_raw_pair(a.raw, base + n, a.fmt);
        if (PSDR_ABL & 2) {
        } else if (a.is_real) {
            const cf w = reinterpret_cast<const cf *>(a.window)[n];
            v.x *= w.x;
            v.y *= w.y;
        } else {
            const float w = a.window[n];
            v.x *= w;
            v.y *= w;
        }
        u[e] = v;
    }
    cf *Yf = a.Y + (size_t)f * M;
    const unsigned Bm = (1u << a.log2B) - 1u;
    if (PSDR_ABL & 4) {
#pragma unroll
        for (int e = 0; e < 16; e++) Yf[(size_t)(i0 + e * (L / 16)) * M2 + n2] = u[e];
        return;
    }
    run_stages<L, T, false>(u, tile, Wl, i0, t, [&](int k1, cf v) {
        const int c1 = a.rot ? ((k1 - 1) & (L - 1)) : k1;
        const unsigned ex = (unsigned)n2 * (unsigned)(a.rot ? c1 + 1 : k1);
        const cf w = (PSDR_ABL & 1) ? make_float2(1.f, 0.f) : cmul(a.TA[ex >> a.log2B], a.TB[ex & Bm]);
        if (a.rot && (n2 & 1)) {
            v.x = -v.x;
            v.y = -v.y;
        }
        Yf[(size_t)c1 * M2 + n2] = cmul(v, w);
    });
}

struct Pass2Args {
    const cf *Y;   // [nframes][M1][M2]
    cf *X;         // [nframes][spec_stride]: bin c1 + M1*c2
    size_t spec_stride;
    const cf *Wl;  // W_L^j, L = M2
    int M1;
    int log2M1;
    // fused IQ epilogue
    float inv_n;
    int size_log2;
    int nlevels;
    int8_t *Q;  // [nframes][q_stride]
    size_t q_stride;
    float *Pscr;  // [nframes][R >> LT]
    size_t p_stride;
    unsigned tiles_per_frame;
    unsigned total_slots;
};

// pass 2: row FFT (length L = M2); FUSED adds /N, |X|^2, int8 level 0..LT of the pyramid
template <int L, int T, bool FUSED>
__global__ __launch_bounds__((L / 16) * T) void k_fft_pass2(Pass2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf *tile = reinterpret_cast<cf *>(smem);
    cf *Wl = tile + L * T;
    constexpr int NT = (L / 16) * T;
    const int tid = threadIdx.x;
    const unsigned slot = xcd_slot(blockIdx.x, a.total_slots);
    const unsigned f = slot / a.tiles_per_frame;
    const unsigned tl = slot - f * a.tiles_per_frame;
    const int t = tid % T, i0 = tid / T;
    const int M1 = a.M1;
    const size_t M = (size_t)L << a.log2M1;
    const int c1base = tl * T;

    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];

    // transposing load: rows c1base..+T of Y (each L contiguous points) -> tile[n2][c1]
    const cf *Yf = a.Y + (size_t)f * M + (size_t)c1base * L;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int idx = e * NT + tid;
        const int r = idx / L, n2 = idx % L;
        tile[lds_idx<T, true>(n2, r)] = Yf[idx];
    }
    __syncthreads();
    cf u[16];
    tile_read<L, T, true>(u, tile, i0, t);
    __syncthreads();

    cf *Xf = a.X + (size_t)f * a.spec_stride;
    if (PSDR_ABL & 32) {
#pragma unroll
        for (int e = 0; e < 16; e++) Xf[(size_t)(i0 + e * (L / 16)) * M1 + c1base + t] = u[e];
        return;
    }
    float *Pst = reinterpret_cast<float *>(smem);  // reuses the (dead) tile after the last read
    run_stages<L, T, true>(u, tile, Wl, i0, t, [&](int c2, cf v) {
        if (FUSED) {
            v.x *= a.inv_n;
            v.y *= a.inv_n;
            Pst[c2 * T + t] = fmaf(v.x, v.x, v.y * v.y);  // src/fft_impl.cpp:36-38
        }
        if (!(PSDR_ABL & 64) || v.x == 1.2345e30f) Xf[(size_t)c2 * M1 + c1base + t] = v;
    });

    if (FUSED && !(PSDR_ABL & 16)) {
        __syncthreads();
        constexpr int CH = T < 16 ? T : 16;  // values per chunk (one aligned group)
        constexpr int NCH = 16 / CH;
        constexpr int LT = CH == 16 ? 4 : (CH == 8 ? 3 : 2);
        const size_t R = M;
        int8_t *Qf = a.Q + (size_t)f * a.q_stride;
        float *Pf = a.Pscr + (size_t)f * a.p_stride;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) {
            const int g = tid * NCH + cc;  // chunk id; chunks tile Pst linearly
            const int row = (g * CH) / T, sub = (g * CH) % T;
            const size_t c = (size_t)row * M1 + c1base + sub;  // client-order bin of value 0
            float p[CH];
#pragma unroll
            for (int v4 = 0; v4 < CH / 4; v4++) {
                const float4 q4 = reinterpret_cast<const float4 *>(Pst)[(g * CH) / 4 + v4];
                p[4 * v4] = q4.x;
                p[4 * v4 + 1] = q4.y;
                p[4 * v4 + 2] = q4.z;
                p[4 * v4 + 3] = q4.w;
            }
            pyr_levels<CH, 0>(p, Qf, 0, R, c, a.nlevels, a.size_log2);
            Pf[c >> LT] = p[0];
        }
    }
}

}  // namespace psdr
