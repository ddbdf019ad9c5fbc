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
// Execution shape (what the MI355X wants, from measurement — profiles/r01_*):
//  * a tile of T sequences x L points (L*T = 16K complex = 128 KiB) lives in LDS, so one
//    work-group owns a CU; a CU can only pull ~10 B/clk from HBM, so memory time and
//    butterfly time must OVERLAP inside that one work-group: kernels are persistent (one
//    work-group per CU walks the tile slots) and the global loads of the NEXT tile are
//    issued into registers before the current tile's butterflies start.
//  * that needs > 128 VGPRs, so a work-group is 512 threads (2 waves/SIMD, 256 VGPRs)
//    and each thread plays V = 2 "virtual threads" of the 1024-point-wide schedule.
//  * every virtual thread owns the 16 points {i0 + e*L/16} of one sequence at every
//    stage and performs 16/R radix-R butterflies per stage (R in {16,8,4,2}); lanes run
//    along the T (contiguous-in-HBM) dimension: stage traffic in LDS is conflict-free.
//  * HBM segments shorter than 128 B run at a fraction of the bandwidth (64-B raw
//    segments measured 1.4 TB/s), so for <= 16-bit sample formats pass 1 loads TWO
//    adjacent columns per virtual thread (8-byte loads, 128-B segments for cs16) and
//    runs the two column sets through the LDS tile one after the other.
//  * the Hann window is evaluated on the fly from the twiddle tables
//    (w = 0.5 - 0.5*Re(W_M1^{n1} * W_M^{n2})), the inter-pass twiddles by short power
//    recurrences from three table look-ups per column: no per-point table traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "butterfly.h"
#include "quantize.h"

#ifndef PSDR_ABL
#define PSDR_ABL 0  // ablation bitmask, tuning builds only (tools/ablate.sh)
#endif

#define PSDR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// phase timestamps of work-group 0 (debug builds of the tuning tools only)
#ifdef PSDR_TRACE_ON
#define PSDR_TRACE(buf, it, k)                                                      \
    do {                                                                            \
        if ((buf) && threadIdx.x == 0 && blockIdx.x == 0 && (it) < 8)               \
            (buf)[(it) * 16 + (k)] = __builtin_readcyclecounter();                  \
    } while (0)
#else
#define PSDR_TRACE(buf, it, k) \
    do {                       \
    } while (0)
#endif

namespace psdr {

// compile-time loops/dispatch: register arrays must only ever be indexed by constants
// (a runtime-looking index puts the whole array in scratch memory)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_switch(int k, F &&f) {
    if constexpr (I < N) {
        if (k == I)
            f(std::integral_constant<int, I>{});
        else
            static_switch<I + 1, N>(k, f);
    }
}

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
template <int L>
struct LastStage {
    using P = Plan<L>;
    static constexpr int R = P::NS == 2 ? P::R1 : P::R2;
    static constexpr int Pp = L / R;  // product of the earlier radices
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

// One Stockham stage on the 16 points a virtual thread owns.  P = product of earlier
// radices.  butterfly i = i0 + b*L/16, k = i mod P, j = (i-k)*R + k
//   x_q = u[b + q*16/R] * W_L^{q*k*L/(P*R)};  out[j + s*P] = DFT_R(x)[s]
// emit(b, s, pos, value)
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
        for (int s = 0; s < R; s++) emit(b, s, j + s * P, x[s]);
        if (R >= 8) PSDR_SCHED_FENCE();
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

// All stages for the V virtual threads of one real thread.
//   load0(v, u)   produces the stage-0 input of virtual thread v (called right before its
//                 first butterfly so the V inputs never coexist in registers)
//   pre_last(v)   runs right before v's last-stage butterflies (twiddle set-up)
//   emit_last(v, b, s, pos, value)  receives the last stage's outputs
// When the last stage starts, every thread has passed the barrier that follows the last
// LDS read (the tile is dead and may be reused by emit_last).
//   tick(k)       k = stage*V + v, called after each virtual thread's butterflies: the
//                 caller issues a slice of the next tile's global loads there, so they
//                 trickle through the compute phases instead of blocking in one burst
template <int L, int T, int V, bool SWZ, typename Load0, typename PreLast, typename EmitLast, typename Tick,
          typename Mark>
__device__ __forceinline__ void run_stages(cf *tile, const cf *Wl, const int (&i0)[V], const int (&t)[V],
                                           Load0 load0, PreLast pre_last, EmitLast emit_last, Tick tick,
                                           Mark mark) {
    using P = Plan<L>;
    cf u[V][16];
#pragma unroll
    for (int v = 0; v < V; v++) {
        load0(v, u[v]);
        stage_compute<L, P::R0, 1>(u[v], i0[v], Wl,
                                   [&](int, int, int pos, cf x) { tile[lds_idx<T, SWZ>(pos, t[v])] = x; });
        PSDR_SCHED_FENCE();  // keep the virtual threads' butterflies apart (register pressure)
        tick(v);
        PSDR_SCHED_FENCE();
    }
    mark(4);
    __syncthreads();
    mark(5);
#pragma unroll
    for (int v = 0; v < V; v++) tile_read<L, T, SWZ>(u[v], tile, i0[v], t[v]);
    __syncthreads();
    mark(6);
    if constexpr (P::NS == 3) {
#pragma unroll
        for (int v = 0; v < V; v++) {
            stage_compute<L, P::R1, P::R0>(u[v], i0[v], Wl, [&](int, int, int pos, cf x) {
                tile[lds_idx<T, SWZ>(pos, t[v])] = x;
            });
            PSDR_SCHED_FENCE();
            tick(V + v);
            PSDR_SCHED_FENCE();
        }
        mark(7);
        __syncthreads();
        mark(8);
#pragma unroll
        for (int v = 0; v < V; v++) tile_read<L, T, SWZ>(u[v], tile, i0[v], t[v]);
        __syncthreads();
        mark(9);
    }
#pragma unroll
    for (int v = 0; v < V; v++) {
        pre_last(v);
        stage_compute<L, LastStage<L>::R, LastStage<L>::Pp>(
            u[v], i0[v], Wl, [&](int b, int s, int pos, cf x) { emit_last(v, b, s, pos, x); });
        PSDR_SCHED_FENCE();
        tick((P::NS - 1) * V + v);
        PSDR_SCHED_FENCE();
    }
}

// Global stores are ISSUE-bound on this chip (a VMEM store costs ~100+ cycles of a CU's
// memory pipe whatever its width), so every store carries 16 bytes per lane: lanes t (even)
// and t+1 hold adjacent columns of the same rows; they swap one value through DPP and the
// even lane stores row A (both columns), the odd lane row B.  rowA/rowB point at the even
// column of the respective row (16-byte aligned).
__device__ __forceinline__ float dpp_swap1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ void pair_store(bool odd, cf xa, cf xb, cf *rowA, cf *rowB) {
    const cf send = odd ? xa : xb;
    const cf recv = make_float2(dpp_swap1(send.x), dpp_swap1(send.y));
    const float4 o = odd ? make_float4(recv.x, recv.y, xb.x, xb.y) : make_float4(xa.x, xa.y, recv.x, recv.y);
    *reinterpret_cast<float4 *>(odd ? rowB : rowA) = o;
}

// XCD-aware slot mapping: work-group b runs on XCD b%8 (observed; used for speed only).
// Groups of 8 adjacent tiles (one 128-byte line of int8 output, 1 KiB of spectrum) are
// given to one XCD in the same persistent iteration so their partial lines merge in that
// XCD's L2.
__device__ __forceinline__ unsigned xcd_slot(unsigned bid, unsigned total) {
    if (total & 63u) return bid;
    const unsigned x = bid & 7u, y = bid >> 3;
    return ((x + 8u * (y >> 3)) << 3) + (y & 7u);
}

// two-level twiddle: W_M^e = TA[e >> log2B] * TB[e & (B-1)]
struct Tw2 {
    const cf *TA, *TB;
    int log2B;
    __device__ __forceinline__ cf operator()(unsigned e) const {
        return cmul(TA[e >> log2B], TB[e & ((1u << log2B) - 1u)]);
    }
    // the two factors, not yet multiplied (lets the loads stay in flight)
    __device__ __forceinline__ void raw(unsigned e, cf &fa, cf &fb) const {
        fa = TA[e >> log2B];
        fb = TB[e & ((1u << log2B) - 1u)];
    }
};

// ---- raw sample access (src/samplereader.cpp:29-40): unsigned formats flip the MSB,
// integers are divided by 2^(bits-1) (exact, so multiply by the reciprocal) ----
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
// two adjacent complex samples of a <=16-bit format, kept packed (1 or 2 VGPRs)
__device__ __forceinline__ uint2 load_raw_packed2(const void *raw, size_t pair_idx, int fmt) {
    if (fmt <= 1) return make_uint2(reinterpret_cast<const unsigned *>(raw)[pair_idx], 0u);
    return reinterpret_cast<const uint2 *>(raw)[pair_idx];
}
__device__ __forceinline__ cf unpack_raw(uint2 p, int col, int fmt) {
    if (fmt <= 1) {
        unsigned h = col ? (p.x >> 16) : (p.x & 0xFFFFu);
        if (fmt == 0) h ^= 0x8080u;
        return make_float2((float)(int8_t)(h & 0xFFu) * (1.0f / 128.0f),
                           (float)(int8_t)(h >> 8) * (1.0f / 128.0f));
    }
    unsigned w = col ? p.y : p.x;
    if (fmt == 2) w ^= 0x80008000u;
    return make_float2((float)(int16_t)(w & 0xFFFFu) * (1.0f / 32768.0f),
                       (float)(int16_t)(w >> 16) * (1.0f / 32768.0f));
}

struct Pass1Args {
    const void *raw;  // nframes+1 raw half-frames, contiguous
    cf *Y;            // blocked: [nframes][tile][M1][T]
    const cf *Wl;     // W_L^j, j < L (L = M1)
    const cf *TB;     // W_M^l, l < M2
    cf wdelta;        // W_N^1 (real input: window angle of the odd sample)
    int M2;
    int log2M2;
    int fmt;
    int is_real;  // window pairs (w[2n], w[2n+1]) instead of (w[n], w[n])
    int rot;      // IQ: produce client order
    size_t yblk;  // elements between consecutive blocks of Y (>= L*T)
    size_t yframe;  // elements between frames of Y
    unsigned tiles_per_frame;
    unsigned total_slots;
    unsigned long long *trace;
};

// one complex sample of the LDS raw image -> float2 (src/samplereader.cpp:29-40): unsigned
// formats flip the MSB, integers are divided by 2^(bits-1) (exact: multiply by the reciprocal)
template <int SB>
__device__ __forceinline__ cf image_to_cf(const unsigned char *img, int elem, int fmt) {
    if constexpr (SB == 2) {
        unsigned h = reinterpret_cast<const unsigned short *>(img)[elem];
        if (fmt == 0) h ^= 0x8080u;
        return make_float2((float)(int8_t)(h & 0xFFu) * (1.0f / 128.0f),
                           (float)(int8_t)(h >> 8) * (1.0f / 128.0f));
    } else if constexpr (SB == 4) {
        unsigned w = reinterpret_cast<const unsigned *>(img)[elem];
        if (fmt == 2) w ^= 0x80008000u;
        return make_float2((float)(int16_t)(w & 0xFFFFu) * (1.0f / 32768.0f),
                           (float)(int16_t)(w >> 16) * (1.0f / 32768.0f));
    } else {
        return reinterpret_cast<const cf *>(img)[elem];
    }
}

// pass 1: convert + window + column FFT (length L = M1) + inter-pass twiddle.
//   T columns per tile, V virtual threads per thread, SB bytes per complex sample of the
//   raw image (u8/s8: 2, u16/s16: 4, f32 and f64-narrowed-to-f32: 8).
// HBM throughput on this chip is proportional to the bytes a load instruction carries
// (measured: dword 1.9 TB/s, dwordx2 3.4, dwordx4 5.6 for the same 128-byte segments), so the
// raw tile is fetched with 16-byte loads in row order, parked in LDS as an image, and the
// virtual threads pick their strided points out of LDS.
template <int L, int T, int V, int SB>
__global__ __launch_bounds__((L / 16) * T / V) void k_fft_pass1(Pass1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf *tile = reinterpret_cast<cf *>(smem);
    cf *Wl = tile + L * T;
    constexpr int NTV = (L / 16) * T;  // virtual threads
    constexpr int NT = NTV / V;
    constexpr int L16 = L / 16;
    constexpr int RL = LastStage<L>::R, PL = LastStage<L>::Pp, NBL = 16 / RL;
    constexpr int ROWB = T * SB;                            // bytes of one image row
    constexpr int LPR = ROWB / 16 > 0 ? ROWB / 16 : 1;      // 16-byte chunks (lanes) per row
    constexpr int NCHK = (L * T * SB) / (16 * NT);          // chunks per thread
    static_assert(ROWB >= 16 && (L * T * SB) % (16 * NT) == 0 && NT % LPR == 0, "tile shape");
    constexpr int NTICK = (Plan<L>::NS - 1) * V;            // ticks that carry loads
    constexpr int EARLY = NCHK < 4 ? NCHK : NCHK / 4;       // loads issued right after the image write
    constexpr int LPT = (NCHK - EARLY + NTICK - 1) / NTICK;
    const int tid = threadIdx.x;
    int t_[V], i0_[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        t_[v] = (tid + v * NT) % T;
        i0_[v] = (tid + v * NT) / T;
    }
    const int M2 = a.M2;
    const size_t M = (size_t)L << a.log2M2;
    const unsigned total = a.total_slots;
    const int fmt = a.fmt;

    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];
    // two-level W_M table with B = M2: W_M^e = W_M1^{e >> log2M2} * W_M^{e & (M2-1)}.  The
    // first factor IS the stage table Wl (already in LDS); the second (M2 entries) is staged
    // next to it, so a per-tile twiddle look-up costs two LDS reads, no L2 round trip.
    cf *ldsTB = Wl + L;
    for (int i = tid; i < M2; i += NT) ldsTB[i] = a.TB[i];
    auto tw = [&](unsigned e) -> cf { return cmul(Wl[e >> a.log2M2], ldsTB[e & (unsigned)(M2 - 1)]); };

    // chunk i*NT + tid of the image = row (i*NT + tid)/LPR, byte (tid % LPR)*16 of that row.
    // global: frame f starts at complex sample f*M/2; row r is M2 samples further.
    const unsigned gsb = fmt == 5 ? 16u : (unsigned)SB;  // bytes per complex sample in HBM
    const size_t g_row = (size_t)M2 * gsb;
    const size_t g_step = (size_t)(NT / LPR) * g_row;  // between a thread's consecutive chunks
    const size_t g_lane = (size_t)(tid / LPR) * g_row + (size_t)(tid % LPR) * (16 / SB) * gsb;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // plain vector: SROA keeps it in VGPRs
    u32x4 rq[NCHK];
    const unsigned char *nxt = nullptr;
    auto point_at = [&](unsigned sidx) {
        const unsigned slot = xcd_slot(sidx, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        nxt = reinterpret_cast<const unsigned char *>(a.raw) + ((size_t)f * (M / 2) + (size_t)tl * T) * gsb + g_lane;
    };
    auto issue = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const unsigned char *p = nxt + (size_t)i * g_step;
        if (PSDR_ABL & 8) {
            rq[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
        } else if (SB == 8 && fmt == 5) {  // f64: two samples = 32 bytes, narrowed to f32 here
            const double2 s0 = reinterpret_cast<const double2 *>(p)[0];
            const double2 s1 = reinterpret_cast<const double2 *>(p)[1];
            rq[i] = u32x4{__float_as_uint((float)s0.x), __float_as_uint((float)s0.y),
                          __float_as_uint((float)s1.x), __float_as_uint((float)s1.y)};
        } else {
            rq[i] = *reinterpret_cast<const u32x4 *>(p);
        }
    };
    unsigned s = blockIdx.x;
    if (s < total) {
        point_at(s);
        static_for<0, NCHK>(issue);
    }
    __syncthreads();  // Wl and the twiddle table are visible

    int it = 0;
    for (; s < total; s += gridDim.x) {
        PSDR_TRACE(a.trace, it, 0);
        const unsigned slot = xcd_slot(s, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        cf *Yb = a.Y + (size_t)f * a.yframe + (size_t)tl * a.yblk;  // this tile's block
        const bool more = s + gridDim.x < total;
        if (more) point_at(s + gridDim.x);
        // opaque per-iteration copies: stop LICM from hoisting the ~100 loop-invariant LDS
        // addresses of all stages out of the persistent loop (that costs >100 VGPRs)
        int i0[V], t[V], tidx = tid;
#pragma unroll
        for (int v = 0; v < V; v++) {
            i0[v] = i0_[v];
            t[v] = t_[v];
            asm volatile("" : "+v"(i0[v]), "+v"(t[v]));
        }
        asm volatile("" : "+v"(tidx));

        // ---- raw image into LDS (linear, 16 bytes per lane)
        static_for<0, NCHK>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            reinterpret_cast<u32x4 *>(smem)[i * NT + tidx] = rq[i];
        });
        PSDR_SCHED_FENCE();
        if (more) static_for<0, EARLY>(issue);
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, it, 1);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 2);

        // ---- stage-0 input: convert (src/samplereader.cpp:29-40) + Hann window; the image
        // shares the tile's LDS, so every virtual thread reads before anyone writes
        int n2[V];
        cf pre[V][16];
#pragma unroll
        for (int v = 0; v < V; v++) {
            n2[v] = (int)tl * T + t[v];
            const cf wb = tw((unsigned)n2[v]);  // W_M^{n2}: window angle of the column
#pragma unroll
            for (int e = 0; e < 16; e++)
                pre[v][e] = image_to_cf<SB>(smem, (i0[v] + e * L16) * T + t[v], fmt);
            if (!(PSDR_ABL & 2)) {
                // periodic Hann (src/utils/dsp.cpp:6-11) from the twiddle tables:
                // exp(-i*2*pi*n/M) = W_M1^{n1} * W_M^{n2}
                if (a.is_real) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const cf z = cmul(Wl[i0[v] + e * L16], wb);
                        pre[v][e].x *= fmaf(-0.5f, z.x, 0.5f);
                        pre[v][e].y *= fmaf(-0.5f, cmul(z, a.wdelta).x, 0.5f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const cf wl = Wl[i0[v] + e * L16];
                        const float w = fmaf(-0.5f, fmaf(wl.x, wb.x, -wl.y * wb.y), 0.5f);
                        pre[v][e].x *= w;
                        pre[v][e].y *= w;
                    }
                }
            }
            PSDR_SCHED_FENCE();
        }
        __syncthreads();
        PSDR_TRACE(a.trace, it, 3);

        if (PSDR_ABL & 4) {
#pragma unroll
            for (int v = 0; v < V; v++)
#pragma unroll
                for (int e = 0; e < 16; e++) Yb[(size_t)(i0[v] + e * L16) * T + t[v]] = pre[v][e];
            if (more) static_for<EARLY, NCHK>(issue);
            it++;
            continue;
        }
        cf tb[NBL], ts[RL], w00;  // inter-pass twiddles of the virtual thread in flight
        cf held = make_float2(0.f, 0.f);
        cf *held_row = nullptr;
        run_stages<L, T, V, false>(
            tile, Wl, i0, t,
            [&](int v, cf(&u)[16]) {
#pragma unroll
                for (int e = 0; e < 16; e++) u[e] = pre[v][e];
            },
            // ---- inter-pass twiddle W_M^{n2*kappa}, kappa = i0 + b*L/16 + s*P, as
            // base * stepB^b * stepS^s (three table look-ups per column)
            [&](int v) {
                const unsigned n2u = (unsigned)n2[v];
                cf base = tw(n2u * (unsigned)i0[v]);
                if (PSDR_ABL & 1) base = make_float2(1.f, 0.f);
                if (a.rot && (n2u & 1u)) {
                    base.x = -base.x;
                    base.y = -base.y;
                }
                cf sb = tw(n2u * (unsigned)L16);
                cf ss = tw(n2u * (unsigned)PL);
                if (PSDR_ABL & 1) sb = ss = make_float2(1.f, 0.f);
                tb[0] = base;
#pragma unroll
                for (int b = 1; b < NBL; b++) tb[b] = cmul(tb[b - 1], sb);
                ts[0] = make_float2(1.f, 0.f);
#pragma unroll
                for (int q = 1; q < RL; q++) ts[q] = q == 1 ? ss : cmul(ts[q - 1], ss);
                // output (b=0,s=0) of the virtual thread with i0 = 0 is bin k1 = 0: in
                // client order it goes to row M1-1 with W_M^{n2*M1}
                w00 = base;
                if (a.rot && i0[v] == 0 && !(PSDR_ABL & 1)) {
                    w00 = tw(n2u * (unsigned)L);
                    if (n2u & 1u) {
                        w00.x = -w00.x;
                        w00.y = -w00.y;
                    }
                }
            },
            [&](int v, int b, int sidx, int k1, cf x) {
                const cf w = (sidx == 0) ? (b == 0 ? w00 : tb[b]) : cmul(tb[b], ts[sidx]);
                const int c1 = a.rot ? ((k1 - 1) & (L - 1)) : k1;
                const cf y = cmul(x, w);
                // outputs arrive as (s even, s odd) pairs: one 16-byte store per pair
                cf *row = Yb + (size_t)c1 * T + (t[v] & ~1);
                if (PSDR_ABL & 256) {
                    Yb[(size_t)c1 * T + t[v]] = y;
                } else if ((sidx & 1) == 0) {
                    held = y;
                    held_row = row;
                } else {
                    pair_store(t[v] & 1, held, y, held_row, row);
                }
            },
            // ---- trickle the rest of the next tile's loads through the stages
            [&](int k) {
                if (more)
                    static_switch<0, NTICK>(k, [&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        constexpr int lo = EARLY + K * LPT < NCHK ? EARLY + K * LPT : NCHK;
                        constexpr int hi = lo + LPT < NCHK ? lo + LPT : NCHK;
                        static_for<lo, hi>(issue);
                    });
            },
            [&](int k) { PSDR_TRACE(a.trace, it, k); });
        PSDR_TRACE(a.trace, it, 10);
        it++;
    }
}

struct Pass2Args {
    const cf *Y;   // blocked: [nframes][tiles1][M1][TW]
    cf *X;         // [nframes][spec_stride]: bin c1 + M1*c2
    size_t spec_stride;
    const cf *Wl;  // W_L^j, L = M2
    int M1;
    int log2M1;
    int TW;       // pass-1 tile width (columns per block of Y)
    int log2TW;
    size_t yblk, yframe;  // block / frame strides of Y in elements
    // fused IQ epilogue
    float inv_n;
    int size_log2;
    int nlevels;
    int8_t *Qt;  // tiled pyramid records, [nframes][qt_stride] (quantize.h)
    size_t qt_stride;
    float *Pscr;  // [nframes][R >> LT]
    size_t p_stride;
    unsigned tiles_per_frame;
    unsigned total_slots;
    unsigned long long *trace;
};

// pass 2: row FFT (length L = M2); FUSED adds /N, |X|^2, int8 level 0..LT of the pyramid.
template <int L, int T, bool FUSED, int V>
__global__ __launch_bounds__((L / 16) * T / V) void k_fft_pass2(Pass2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf *tile = reinterpret_cast<cf *>(smem);
    cf *Wl = tile + L * T;
    constexpr int NTV = (L / 16) * T;
    constexpr int NT = NTV / V;
    constexpr int NLD = 8 * V;                       // 16-byte loads per thread and tile
    constexpr int NTICK = (Plan<L>::NS - 1) * V;     // ticks that carry loads (not the last stage)
    constexpr int EARLY = 4 < NLD ? 4 : NLD;         // loads issued right after the registers die
    constexpr int LPT = (NLD - EARLY + NTICK - 1) / NTICK;
    const int tid = threadIdx.x;
    int t_[V], i0_[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        t_[v] = (tid + v * NT) % T;
        i0_[v] = (tid + v * NT) / T;
    }
    const int M1 = a.M1;
    const size_t M = (size_t)L << a.log2M1;
    const unsigned total = a.total_slots;
    const int TW = a.TW;
    const int chunk = T * TW;  // contiguous elements of one pass-1 block that belong to this tile
    const size_t blk = a.yblk;

    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];

    // Element idx of the tile is (block j, row rr, column cc) with idx = j*chunk + rr*TW + cc:
    // Y block j, row c1base + rr, column cc, and n2 = j*TW + cc.  A thread loads 16 bytes =
    // elements idx, idx+1 with idx = 2*(i*NTV + vt), i < 8 (vt = virtual thread id): the
    // address is a uniform per-i part plus ONE per-lane offset per virtual thread.
    const int lc = a.log2TW + (31 - __clz(T));  // log2(chunk)
    float4 r[V][8];
    const cf *nxt = nullptr;
    unsigned lane_off[V];  // j0*blk + w  (elements)
    auto point_at = [&](unsigned sidx) {
        const unsigned slot = xcd_slot(sidx, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        nxt = a.Y + (size_t)f * a.yframe + (size_t)(tl * T) * TW;
    };
    auto issue = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int v = q / 8, i = q % 8;
        // uniform part of idx = 2*i*NTV: block (2*i*NTV)>>lc, offset (2*i*NTV)&(chunk-1)
        const cf *p = nxt + (size_t)((2 * i * NTV) >> lc) * blk + ((2 * i * NTV) & (chunk - 1)) + lane_off[v];
        r[v][i] = *reinterpret_cast<const float4 *>(p);
    };
#pragma unroll
    for (int v = 0; v < V; v++) {
        const int vt2 = 2 * (v * NT + tid);
        lane_off[v] = (unsigned)((size_t)(vt2 >> lc) * blk + (vt2 & (chunk - 1)));
    }
    unsigned s = blockIdx.x;
    if (s < total) {
        point_at(s);
        static_for<0, NLD>(issue);
    }

    int it = 0;
    for (; s < total; s += gridDim.x, it++) {
        PSDR_TRACE(a.trace, it, 0);
        const unsigned slot = xcd_slot(s, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        const int c1base = tl * T;
        const bool more = s + gridDim.x < total;
        if (more) point_at(s + gridDim.x);
        int i0[V], t[V], tidx = tid;  // opaque copies (see pass 1)
#pragma unroll
        for (int v = 0; v < V; v++) {
            i0[v] = i0_[v];
            t[v] = t_[v];
            asm volatile("" : "+v"(i0[v]), "+v"(t[v]));
        }
        asm volatile("" : "+v"(tidx));
        // transposing store: tile[n2][c1]
#pragma unroll
        for (int v = 0; v < V; v++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int vt2 = 2 * (v * NT + tidx);
                const int w = ((2 * i * NTV) & (chunk - 1)) + (vt2 & (chunk - 1));
                const int rr = w >> a.log2TW, cc = w & (TW - 1);
                const int n2 = (((2 * i * NTV) >> lc) + (vt2 >> lc)) * TW + cc;
                tile[lds_idx<T, true>(n2, rr)] = make_float2(r[v][i].x, r[v][i].y);
                tile[lds_idx<T, true>(n2 + 1, rr)] = make_float2(r[v][i].z, r[v][i].w);
            }
        PSDR_SCHED_FENCE();
        if (more) static_for<0, EARLY>(issue);
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, it, 1);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 2);
        // stage-0 input comes from the tile itself: all V reads, then a barrier, before
        // any in-place write
        cf pre[V][16];
#pragma unroll
        for (int v = 0; v < V; v++) tile_read<L, T, true>(pre[v], tile, i0[v], t[v]);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 3);

        cf *Xf = a.X + (size_t)f * a.spec_stride;
        if (PSDR_ABL & 32) {
#pragma unroll
            for (int v = 0; v < V; v++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    Xf[(size_t)(i0[v] + e * (L / 16)) * M1 + c1base + t[v]] = pre[v][e];
            if (more) static_for<EARLY, NLD>(issue);
            continue;
        }
        float *Pst = reinterpret_cast<float *>(smem);  // reuses the (dead) tile after the last read
        cf held = make_float2(0.f, 0.f);
        cf *held_row = nullptr;
        run_stages<L, T, V, true>(
            tile, Wl, i0, t,
            [&](int v, cf(&u)[16]) {
#pragma unroll
                for (int e = 0; e < 16; e++) u[e] = pre[v][e];
            },
            [&](int) {},
            [&](int v, int, int sidx, int c2, cf x) {
                if (FUSED) {
                    x.x *= a.inv_n;
                    x.y *= a.inv_n;
                    Pst[c2 * T + t[v]] = fmaf(x.x, x.x, x.y * x.y);  // src/fft_impl.cpp:36-38
                }
                if (PSDR_ABL & 64) return;
                cf *row = Xf + (size_t)c2 * M1 + c1base + (t[v] & ~1);
                if ((sidx & 1) == 0) {
                    held = x;
                    held_row = row;
                } else {
                    pair_store(t[v] & 1, held, x, held_row, row);
                }
            },
            [&](int k) {  // trickle the rest of the next tile's loads through the stages
                if (more)
                    static_switch<0, NTICK>(k, [&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        constexpr int lo = EARLY + K * LPT < NLD ? EARLY + K * LPT : NLD;
                        constexpr int hi = lo + LPT < NLD ? lo + LPT : NLD;
                        static_for<lo, hi>(issue);
                    });
            },
            [&](int k) { PSDR_TRACE(a.trace, it, k); });
        PSDR_TRACE(a.trace, it, 10);

        if (FUSED && !(PSDR_ABL & 16)) {
            __syncthreads();
            PSDR_TRACE(a.trace, it, 11);
            constexpr int CH = T < 16 ? T : 16;  // values per chunk (one aligned group)
            constexpr int NCH = 16 / CH;
            constexpr int LT = CH == 16 ? 4 : 3;
            static_assert(CH == 16 || CH == 8, "tile width");
            int8_t *Qf = a.Qt + (size_t)f * a.qt_stride;
            float *Pf = a.Pscr + (size_t)f * a.p_stride;
#pragma unroll
            for (int v = 0; v < V; v++) {
#pragma unroll
                for (int cc = 0; cc < NCH; cc++) {
                    const int g = (tidx + v * NT) * NCH + cc;  // chunk id; chunks tile Pst linearly
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
                    // levels 0..LT of this aligned group -> one contiguous record
                    uint4 *rec = reinterpret_cast<uint4 *>(Qf + (c / CH) * (2 * CH));
                    if constexpr (CH == 16) {
                        uint4 lo, hi;
                        pyr_record16(p, a.size_log2, lo, hi);
                        rec[0] = lo;
                        rec[1] = hi;
                    } else {
                        uint4 r8;
                        pyr_record8(p, a.size_log2, r8);
                        rec[0] = r8;
                    }
                    Pf[c >> LT] = p[0];
                    PSDR_SCHED_FENCE();
                }
            }
        }
        PSDR_TRACE(a.trace, it, 12);
        __syncthreads();  // the tile is free again
        PSDR_TRACE(a.trace, it, 13);
    }
}

}  // namespace psdr
