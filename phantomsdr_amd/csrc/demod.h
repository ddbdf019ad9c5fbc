// demod.h — batched per-client downconversion: frequency-domain slice -> small inverse
// DFT -> USB/LSB/AM/FM, with the 50 % overlap-add carried across frames.
//
// Replaces AudioClient::send_audio (src/signal.cpp:102-275) for all clients and all
// frames of a batch at once:
//   k_demod_idft  one work-group per (client, frame): average_power (:117-119), the
//                 mode-specific bin copy (:125-153, :175-198), the n-point backward
//                 transform (fftwf c2r / c2c, :138,154,214), LSB reversal (:155) and the
//                 odd-frame sign flip (:160-168, :223-234).
//   k_demod_ola   overlap-add with the previous frame's second half (:171-172,
//                 :235-237), AM envelope (dsp_am_demod, src/utils/dsp.cpp:116-126),
//                 FM polar discriminator (src/utils/dsp.cpp:27-35), NaN guard (:266-271)
//                 and the state carried to the next batch (:200-203, :273-275).
// The AM "carrier" transform (:205-222,230-241) feeds only the liquid-dsp PLL branch
// (:242-252), which is not part of the parity target; without liquid it has no
// observable effect and is not computed.
//
// n = audio_fft_size is any multiple of 4 (248, 360, 720, 10068 ...): the transform is
// a generic-radix Stockham, each radix-R butterfly a direct R-point DFT whose
// inter-stage twiddle is folded into a single table lookup:
//   y[j + s*p] = sum_q x[i + q*n/R] * W_n^{ q*(k + s*p)*n/(p*R) },  k = i mod p,
//   j = (i-k)*R + k,  W_n = exp(+2*pi*i/n)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "butterfly.h"
#include "quantize.h"
#include "types.h"

namespace psdr {



struct DemodArgs {
    const cf *spec;  // [nframes][spec_stride]; IQ: client order, real: k order, through `lay`
    size_t spec_stride;
    SpecLayout lay;  // where bin c of a frame sits (quantize.h); natural order unless the fused real path
    int is_real;
    int n;  // audio_fft_size
    int nframes;
    int max_batch;
    unsigned long long first_frame_num;
    const ClientParams *clients;  // compact list of active clients
    const cf *Wn;                 // exp(+2 pi i j / n), j < n
    int nstages;
    int radix[PSDR_MAX_STAGES];
    const int4 *stage_tab;  // [nstages][n]: {first input i, output index j+s*p, twiddle step e1, 0}
    cf *ypost;  // [slots][max_batch][n]: transform output after reversal/flip
    float *pwr;  // [slots][max_batch]
    // workspace for transforms too large for LDS: [gridDim.x*gridDim.y][2][n]
    cf *gscratch;
    int lds_mode;  // 0: bufA, bufB, Wn in LDS; 1: bufA, bufB in LDS; 2: all global
    // ola
    float *audio;  // [slots][max_batch][n/2]
    int *nan_flags;  // [slots][max_batch]
    float *real_prev;  // [2][slots][n/2]
    cf *bb_tail;       // [2][slots][n/2]
    cf *bb_last;       // [2][slots]
    int slots;
    // The NaN guard of USB / LSB (src/signal.cpp:266-275) is a recurrence over a client's frames: frame f is dropped iff
    // y_f[0..n/2) + prev has a NaN, and prev moves on only if it was not.  The batched kernels evaluate it per chain of frames
    // with "the frame before was dropped iff its transform carries NaN" - exact as long as every value is finite or a
    // transform is NaN throughout (a NaN input sample).  A client whose batch shows ANY non-finite value (an Inf sample, a
    // tail that went bad in an earlier batch) is marked here and walked again, strictly in frame order, by one wave
    // (`replay`: k_demod_chain_fixed with the whole batch as one chain / k_demod_ola_seq) - the reference's rule on the
    // values at hand, whatever they are.  Integer input formats never get there.
    unsigned *ssb_mark;   // [slots]: == mark_epoch: the slot's batch needs the sequential walk
    unsigned mark_epoch;  // of this batch (never 0)
    int replay;           // 1: the sequential walk itself (marked slots only)
};
__device__ __forceinline__ bool not_finite(float v) { return !(fabsf(v) <= 3.402823466e38f); }  // NaN or +-Inf

__device__ __forceinline__ bool flip_frame(unsigned long long frame_num, int m_idx, int is_real) {
    // src/signal.cpp:160-162 with C++ remainder semantics for negative m_idx
    return (frame_num % 2 == 1) && ((m_idx % 2 == 0 && !is_real) || (m_idx % 2 == 1 && is_real));
}

// one Stockham stage; RC > 0: radix known at compile time (all operand loads of a butterfly
// are issued before the MAC chain), RC == 0: run-time radix (large prime factors)
template <int RC>
__device__ __forceinline__ void idft_stage(const cf *src, cf *dst, const cf *Wn, const int4 *tab, int n,
                                           int R, int tid, int NT) {
    const int tlen = n / R;
    for (int o = tid; o < n; o += NT) {
        const int4 tb = tab[o];
        const int e1 = tb.z;
        const cf *xp = src + tb.x;
        float ar = 0.f, ai = 0.f;
        if constexpr (RC > 0) {
            // operands in groups of 4: enough loads in flight, few registers (the kernel shares
            // the SIMDs' register file with the FFT passes: <= 48 VGPRs doubles its occupancy)
            int e = 0;
#pragma unroll
            for (int q0 = 0; q0 < RC; q0 += 4) {
                constexpr int G = 4;
                cf x[G], w[G];
#pragma unroll
                for (int g = 0; g < G; g++)
                    if (q0 + g < RC) {
                        x[g] = xp[(q0 + g) * tlen];
                        w[g] = Wn[e];
                        e += e1;
                        if (e >= n) e -= n;
                    }
#pragma unroll
                for (int g = 0; g < G; g++)
                    if (q0 + g < RC) {
                        ar = fmaf(x[g].x, w[g].x, fmaf(-x[g].y, w[g].y, ar));
                        ai = fmaf(x[g].x, w[g].y, fmaf(x[g].y, w[g].x, ai));
                    }
            }
        } else {
            int e = 0;
#pragma unroll 4
            for (int q = 0; q < R; q++) {
                const cf x = xp[q * tlen];
                const cf w = Wn[e];
                ar = fmaf(x.x, w.x, fmaf(-x.y, w.y, ar));
                ai = fmaf(x.x, w.y, fmaf(x.y, w.x, ai));
                e += e1;
                if (e >= n) e -= n;
            }
        }
        dst[tb.y] = make_float2(ar, ai);
    }
}

// blockDim.x = 128 (n <= 512) or 256
__global__ __launch_bounds__(256) void k_demod_idft(DemodArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = a.n, tid = threadIdx.x, NT = blockDim.x;
    const ClientParams cp = a.clients[blockIdx.x];
    const int f = blockIdx.y;
    const unsigned long long frame_num = a.first_frame_num + (unsigned long long)f;

    cf *bufA, *bufB;
    const cf *Wn;
    if (a.lds_mode == 0) {
        bufA = reinterpret_cast<cf *>(smem);
        bufB = bufA + n;
        cf *w = bufB + n;
        for (int i = tid; i < n; i += NT) w[i] = a.Wn[i];
        Wn = w;
    } else if (a.lds_mode == 1) {
        bufA = reinterpret_cast<cf *>(smem);
        bufB = bufA + n;
        Wn = a.Wn;
    } else {
        bufA = a.gscratch + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2) * n;
        bufB = bufA + n;
        Wn = a.Wn;
    }
    __shared__ float red[4];

    const int len = cp.r - cp.l;
    const int m = cp.m_floor - cp.l;  // audio_m
    const cf *S = a.spec + (size_t)f * a.spec_stride;  // slice bin t at lay.pos(cp.l + t)

    for (int i = tid; i < n; i += NT) bufA[i] = make_float2(0.f, 0.f);
    __syncthreads();

    float pw = 0.f;
    for (int t = tid; t < len; t += NT) {
        const cf v = S[a.lay.pos(cp.l + t)];
        pw += fmaf(v.x, v.x, v.y * v.y);
        if (cp.mode == 0) {  // USB :125-137
            if (t >= m && t < m + n) bufA[t - m] = v;
        } else if (cp.mode == 1) {  // LSB :139-153
            if (t >= m - n + 1 && t < m + 1) bufA[m - t] = v;
        } else {  // AM/FM :175-198
            if (t >= m && t < m + n / 2) bufA[t - m] = v;
            if (t >= m - n / 2 + 1 && t < m) bufA[n - m + t] = v;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pw += __shfl_xor(pw, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = pw;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < NT / 64; w++) tot += red[w];
        a.pwr[(size_t)cp.slot * a.max_batch + f] = tot;
    }

    if (cp.mode < 2) {
        // c2r semantics: bins 0..n/2 only, Im of bin 0 and bin n/2 ignored; extend to the
        // Hermitian-symmetric full spectrum so the complex transform returns the real signal
        for (int k = tid + 1; k < n / 2; k += NT) {
            const cf v = bufA[k];
            bufA[n - k] = make_float2(v.x, -v.y);
        }
        if (tid == 0) {
            bufA[0].y = 0.f;
            bufA[n / 2].y = 0.f;
        }
        __syncthreads();
    }

    // generic-radix Stockham, backward; the index arithmetic of every (stage, output) pair is
    // precomputed on the host (stage_tab), the butterfly is R complex MACs
    cf *src = bufA, *dst = bufB;
    for (int st = 0; st < a.nstages; st++) {
        const int R = a.radix[st];
        const int4 *tab = a.stage_tab + (size_t)st * n;
        switch (R) {
#define PSDR_RCASE(r) case r: idft_stage<r>(src, dst, Wn, tab, n, r, tid, NT); break;
            PSDR_RCASE(2) PSDR_RCASE(3) PSDR_RCASE(4) PSDR_RCASE(5) PSDR_RCASE(6) PSDR_RCASE(7)
            PSDR_RCASE(8) PSDR_RCASE(9) PSDR_RCASE(10) PSDR_RCASE(12) PSDR_RCASE(14)
            PSDR_RCASE(15) PSDR_RCASE(16)
#undef PSDR_RCASE
            default: idft_stage<0>(src, dst, Wn, tab, n, R, tid, NT); break;
        }
        cf *tmp = src;
        src = dst;
        dst = tmp;
        __syncthreads();
    }

    const bool flip = flip_frame(frame_num, cp.m_floor, a.is_real);
    const float sg = flip ? -1.f : 1.f;
    cf *yp = a.ypost + ((size_t)cp.slot * a.max_batch + f) * n;
    for (int jx = tid; jx < n; jx += NT) {
        cf v;
        if (cp.mode == 0)
            v = make_float2(src[jx].x * sg, 0.f);
        else if (cp.mode == 1)
            v = make_float2(src[n - 1 - jx].x * sg, 0.f);  // std::reverse :155
        else
            v = make_float2(src[jx].x * sg, src[jx].y * sg);
        yp[jx] = v;
    }
}

// The same transform with one WAVE per (client, frame) and no work-group barrier: the 64
// lanes of a wave execute their LDS operations in order, so a stage boundary is just
// "s_waitcnt lgkmcnt(0)".  With hundreds of clients the one-work-group-per-item kernel above
// is bound by its ~10 barriers per item while it shares the CUs with the persistent FFT
// passes; here a 128-thread work-group carries two independent items in < 16 KiB of LDS (the
// space an FFT pass leaves free on a CU).  n <= 512 (audio_fft_size 248, 360, ...).
//   grid = ceil(nact * nframes / WAVES); dynamic LDS = (2 * WAVES + 1) * n * 8 bytes
#define PSDR_IDFT_WAVES 2
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(64 * PSDR_IDFT_WAVES) void k_demod_idft_wave(DemodArgs a, int nact) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = a.n, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    cf *Wn = reinterpret_cast<cf *>(smem);
    for (int i = threadIdx.x; i < n; i += blockDim.x) Wn[i] = a.Wn[i];
    __syncthreads();  // the only work-group barrier: the shared twiddle table
    const int item = blockIdx.x * PSDR_IDFT_WAVES + wv;
    if (item >= nact * a.nframes) return;
    // consecutive items = consecutive frames of one client: the slice addresses of a
    // work-group's waves are F*8N bytes apart, their table look-ups identical
    const int ci = item / a.nframes, f = item - ci * a.nframes;
    const ClientParams cp = a.clients[ci];
    const unsigned long long frame_num = a.first_frame_num + (unsigned long long)f;
    cf *bufA = Wn + n + (size_t)wv * 2 * n, *bufB = bufA + n;

    const int len = cp.r - cp.l;
    const int m = cp.m_floor - cp.l;  // audio_m
    const cf *S = a.spec + (size_t)f * a.spec_stride;  // slice bin t at lay.pos(cp.l + t)
    for (int i = lane; i < n; i += 64) bufA[i] = make_float2(0.f, 0.f);
    wave_lds_sync();
    float pw = 0.f;
    for (int t = lane; t < len; t += 64) {
        const cf v = S[a.lay.pos(cp.l + t)];
        pw += fmaf(v.x, v.x, v.y * v.y);
        if (cp.mode == 0) {  // USB :125-137
            if (t >= m && t < m + n) bufA[t - m] = v;
        } else if (cp.mode == 1) {  // LSB :139-153
            if (t >= m - n + 1 && t < m + 1) bufA[m - t] = v;
        } else {  // AM/FM :175-198
            if (t >= m && t < m + n / 2) bufA[t - m] = v;
            if (t >= m - n / 2 + 1 && t < m) bufA[n - m + t] = v;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pw += __shfl_xor(pw, d, 64);
    if (lane == 0) a.pwr[(size_t)cp.slot * a.max_batch + f] = pw;
    wave_lds_sync();
    if (cp.mode < 2) {  // c2r semantics, see k_demod_idft
        for (int k = lane + 1; k < n / 2; k += 64) {
            const cf v = bufA[k];
            bufA[n - k] = make_float2(v.x, -v.y);
        }
        if (lane == 0) {
            bufA[0].y = 0.f;
            bufA[n / 2].y = 0.f;
        }
        wave_lds_sync();
    }
    cf *src = bufA, *dst = bufB;
    for (int st = 0; st < a.nstages; st++) {
        const int R = a.radix[st];
        const int4 *tab = a.stage_tab + (size_t)st * n;
        switch (R) {
#define PSDR_RCASE(r) case r: idft_stage<r>(src, dst, Wn, tab, n, r, lane, 64); break;
            PSDR_RCASE(2) PSDR_RCASE(3) PSDR_RCASE(4) PSDR_RCASE(5) PSDR_RCASE(6) PSDR_RCASE(7)
            PSDR_RCASE(8) PSDR_RCASE(9) PSDR_RCASE(10) PSDR_RCASE(12) PSDR_RCASE(14)
            PSDR_RCASE(15) PSDR_RCASE(16)
#undef PSDR_RCASE
            default: idft_stage<0>(src, dst, Wn, tab, n, R, lane, 64); break;
        }
        cf *tmp = src;
        src = dst;
        dst = tmp;
        wave_lds_sync();
    }
    const bool flip = flip_frame(frame_num, cp.m_floor, a.is_real);
    const float sg = flip ? -1.f : 1.f;
    cf *yp = a.ypost + ((size_t)cp.slot * a.max_batch + f) * n;
    for (int jx = lane; jx < n; jx += 64) {
        cf v;
        if (cp.mode == 0)
            v = make_float2(src[jx].x * sg, 0.f);
        else if (cp.mode == 1)
            v = make_float2(src[n - 1 - jx].x * sg, 0.f);  // std::reverse :155
        else
            v = make_float2(src[jx].x * sg, src[jx].y * sg);
        yp[jx] = v;
    }
}

// ---- compile-time plans for the audio sizes the BASELINE configurations use (n = 360, 720)
// The generic kernels above spend most of their ~2000 wave instructions per 360-point item on
// index arithmetic, table look-ups and predication, and with hundreds of clients the
// demodulation is VALU-issue bound (65 536 items: 227 us on an idle chip, 1.5-2x that beside
// the FFT passes).  Here n and the radices are template parameters: a lane owns whole radix-R
// butterflies (R inputs read once, outputs from registers with the R-th roots, conjugate pairs
// (s, R-s) sharing their sums), every LDS access is base + immediate, the twiddle exponent
// q*(i%p)*n/(p*R) never wraps, and each stage runs IN PLACE (all reads of a stage precede its
// first write; LDS operations of one wave execute in order) so four items share 15 KiB.
// ---- in-register INVERSE (sign +) butterflies of the compile-time plans: x[s] <- sum_q x[q] exp(+2 pi i q s / R) ----------
// Round 5's stage computed every butterfly as a direct R-point DFT (conjugate pairs sharing their sums: ~R^2 scalar FMAs -
// 124 / 160 / 48 instructions for R = 8 / 9 / 5), and with hundreds of clients the demodulation is VALU-bound beside the
// passes (1024 clients: a third of the step).  Here R = 8 = 2.2.2, 9 = 3.3, 5 (the symmetric form) and 10 = 2.5 with literal
// roots on packed (re, im) pairs: 30 / 44 / 18 packed instructions.  butterfly.h's add_mi(a, d) = a - i d, sub_mi(a, d) = a + i d.
__device__ __forceinline__ cf id_scale(cf a, float k) { return make_float2(a.x * k, a.y * k); }
__device__ __forceinline__ cf id_fma(cf a, float k, cf c) { return make_float2(fmaf(a.x, k, c.x), fmaf(a.y, k, c.y)); }  // a k + c
__device__ __forceinline__ void idft3(cf &x0, cf &x1, cf &x2) {
    constexpr float S3 = 0.86602540378443864676f;  // sin(2 pi / 3)
    const cf t = cadd(x1, x2), d = csub(x1, x2);
    const cf u = id_fma(t, -0.5f, x0), e = id_scale(d, S3);
    x0 = cadd(x0, t);
    x1 = sub_mi(u, e);  // u + i e
    x2 = add_mi(u, e);  // u - i e
}
__device__ __forceinline__ void idft4(cf &x0, cf &x1, cf &x2, cf &x3) {
    const cf s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = sub_mi(d02, d13);  // d02 + i d13
    x3 = add_mi(d02, d13);
}
__device__ __forceinline__ void idft5(cf &x0, cf &x1, cf &x2, cf &x3, cf &x4) {
    constexpr float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;    // sin(2 pi / 5), sin(4 pi / 5)
    const cf t1 = cadd(x1, x4), t2 = cadd(x2, x3), d1 = csub(x1, x4), d2 = csub(x2, x3);
    const cf a1 = id_fma(t2, C2, id_fma(t1, C1, x0)), a2 = id_fma(t2, C1, id_fma(t1, C2, x0));
    const cf b1 = id_fma(d2, S2, id_scale(d1, S1)), b2 = id_fma(d2, -S1, id_scale(d1, S2));
    x0 = cadd(x0, cadd(t1, t2));
    x1 = sub_mi(a1, b1);
    x4 = add_mi(a1, b1);
    x2 = sub_mi(a2, b2);
    x3 = add_mi(a2, b2);
}
template <int R>
__device__ __forceinline__ void idft_bfly(cf (&x)[R]) {
    if constexpr (R == 5) {
        idft5(x[0], x[1], x[2], x[3], x[4]);
    } else if constexpr (R == 8) {
        // q = 2 q1 + q2, s = s1 + 4 s2: four-point transforms of the even and the odd inputs, odd branch times W8^{s1}
        idft4(x[0], x[2], x[4], x[6]);
        idft4(x[1], x[3], x[5], x[7]);
        const cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o2 = x[5];
        const cf o1 = id_scale(sub_mi(x[3], x[3]), PSDR_SQRT1_2);  // (1 + i) / sqrt 2 * o1
        const cf t3 = id_scale(sub_mi(x[7], x[7]), PSDR_SQRT1_2);  // W8^3 o3 = i * this
        x[0] = cadd(e0, o0);
        x[4] = csub(e0, o0);
        x[1] = cadd(e1, o1);
        x[5] = csub(e1, o1);
        x[2] = sub_mi(e2, o2);  // + i o2
        x[6] = add_mi(e2, o2);
        x[3] = sub_mi(e3, t3);
        x[7] = add_mi(e3, t3);
    } else if constexpr (R == 9) {
        // q = 3 q1 + q2, s = s1 + 3 s2: three-point transforms over q1, y[s1][q2] *= W9^{q2 s1}, three-point transforms over q2
        idft3(x[0], x[3], x[6]);
        idft3(x[1], x[4], x[7]);
        idft3(x[2], x[5], x[8]);
        const cf w1 = make_float2(0.76604444311897803520f, 0.64278760968653932632f);   // exp(+2 pi i 1/9)
        const cf w2 = make_float2(0.17364817766693034885f, 0.98480775301220805937f);   // exp(+2 pi i 2/9)
        const cf w4 = make_float2(-0.93969262078590838405f, 0.34202014332566873304f);  // exp(+2 pi i 4/9)
        // after the first step x[3 s1 + q2] holds y[s1][q2]
        cmul_pair(x[4], x[4], w1, x[5], x[5], w2);
        cmul_pair(x[7], x[7], w2, x[8], x[8], w4);
        idft3(x[0], x[1], x[2]);  // s1 = 0: X[0], X[3], X[6]
        idft3(x[3], x[4], x[5]);  // s1 = 1: X[1], X[4], X[7]
        idft3(x[6], x[7], x[8]);  // s1 = 2: X[2], X[5], X[8]
        // x[3 s1 + s2] = X[s1 + 3 s2]: transpose to natural order
        cf t;
        t = x[1], x[1] = x[3], x[3] = t;
        t = x[2], x[2] = x[6], x[6] = t;
        t = x[5], x[5] = x[7], x[7] = t;
    } else {
        static_assert(R == 10, "butterflies with literal roots: 5, 8, 9, 10");
        // q = 2 q1 + q2, s = s1 + 5 s2
        idft5(x[0], x[2], x[4], x[6], x[8]);
        idft5(x[1], x[3], x[5], x[7], x[9]);
        const cf w1 = make_float2(0.80901699437494742410f, 0.58778525229247312917f);   // exp(+2 pi i 1/10)
        const cf w2 = make_float2(0.30901699437494742410f, 0.95105651629515357212f);
        const cf w3 = make_float2(-0.30901699437494742410f, 0.95105651629515357212f);
        const cf w4 = make_float2(-0.80901699437494742410f, 0.58778525229247312917f);
        cf o1, o2, o3, o4;
        cmul_pair(o1, x[3], w1, o2, x[5], w2);
        cmul_pair(o3, x[7], w3, o4, x[9], w4);
        const cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], e4 = x[8], o0 = x[1];
        x[0] = cadd(e0, o0), x[5] = csub(e0, o0);
        x[1] = cadd(e1, o1), x[6] = csub(e1, o1);
        x[2] = cadd(e2, o2), x[7] = csub(e2, o2);
        x[3] = cadd(e3, o3), x[8] = csub(e3, o3);
        x[4] = cadd(e4, o4), x[9] = csub(e4, o4);
    }
}

template <int N, int R, int PP>
__device__ __forceinline__ void idft_stage_fixed(cf *buf, const cf *Wn, int lane) {
    constexpr int TLEN = N / R, ROUNDS = (TLEN + 63) / 64, STEP = N / (PP * R);
    constexpr bool RAGGED = (TLEN % 64) != 0;
    cf x[ROUNDS][R];
    int jo[ROUNDS];
#pragma unroll
    for (int rr = 0; rr < ROUNDS; rr++) {
        const int i = lane + 64 * rr;
        if (!RAGGED || rr + 1 < ROUNDS || i < TLEN) {
            const int k = PP == 1 ? 0 : i % PP;
            jo[rr] = (i - k) * R + k;
            x[rr][0] = buf[i];
#pragma unroll
            for (int q = 1; q < R; q++) {
                const cf v = buf[i + q * TLEN];
                if (PP > 1)
                    x[rr][q] = cmul(v, Wn[q * k * STEP]);
                else
                    x[rr][q] = v;
            }
        }
    }
    // Compiler barrier, no hardware wait: the LDS executes a wave's operations in order, but the
    // compiler reasons per lane - with compile-time indices it can prove that a lane's own
    // loads and stores never overlap and sink the later rounds' loads below the first stores,
    // which other LANES' stores then clobber (seen as 0.4 % errors in 40 outputs of n = 360).
    asm volatile("" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < ROUNDS; rr++) {
        const int i = lane + 64 * rr;
        if (!RAGGED || rr + 1 < ROUNDS || i < TLEN) {
            cf *o = buf + jo[rr];
            idft_bfly<R>(x[rr]);
#pragma unroll
            for (int s = 0; s < R; s++) o[s * PP] = x[rr][s];
        }
    }
}

// the transform of ONE (client, frame) item by one wave: slice load, mode-specific bin copy, c2r symmetry, the
// stages.  Leaves the n outputs in buf (before reversal / sign flip) and returns the slice power (all lanes).
// the slice of one (client, frame) item, bin t = lane + 64 u in sv[u] (at most n bins, src/signal.cpp:309-311)
template <int N>
__device__ __forceinline__ void idft_load_slice(const DemodArgs &a, const ClientParams &cp, int f, int lane, cf (&sv)[(N + 63) / 64]) {
    const int len = cp.r - cp.l;
    const cf *S = a.spec + (size_t)f * a.spec_stride;  // slice bin t at lay.pos(cp.l + t)
#pragma unroll
    for (int u = 0; u < (N + 63) / 64; u++) {
        const int t = lane + 64 * u;
        sv[u] = t < len ? S[a.lay.pos(cp.l + t)] : make_float2(0.f, 0.f);
    }
}
// the same in two steps for a wave that walks SEVERAL frames of one client: where bin t of the slice sits inside a frame
// (SpecLayout::pos: ~25 instructions per bin in the tile-major layouts) does not depend on the frame - computed once per
// chain, a frame's load is base + offset (k_demod_chain_fixed: a fifth of its instructions per frame were these).  The
// first HO rounds of 64 bins only (registers: the kernel lives in the ~80 a pass leaves per SIMD lane): slices of up to
// 64 HO bins - every SSB / AM / FM window of the usual widths - never compute a position inside the frame loop
template <int N, int HO>
__device__ __forceinline__ void idft_slice_offsets(const DemodArgs &a, const ClientParams &cp, int lane, unsigned (&so)[HO]) {
    const int len = cp.r - cp.l;
#pragma unroll
    for (int u = 0; u < HO; u++) {
        const int t = lane + 64 * u;
        so[u] = t < len ? (unsigned)a.lay.pos(cp.l + t) : 0u;  // (element index inside a frame / band region: < 2^32)
    }
}
template <int N, int HO>
__device__ __forceinline__ void idft_load_slice_at(const DemodArgs &a, const ClientParams &cp, int f, const unsigned (&so)[HO], int lane, cf (&sv)[(N + 63) / 64]) {
    const int len = cp.r - cp.l;
    const cf *S = a.spec + (size_t)f * a.spec_stride;
#pragma unroll
    for (int u = 0; u < (N + 63) / 64; u++) {
        const int t = lane + 64 * u;
        if (u < HO)
            sv[u] = t < len ? S[so[u]] : make_float2(0.f, 0.f);
        else
            sv[u] = t < len ? S[a.lay.pos(cp.l + t)] : make_float2(0.f, 0.f);
    }
}
template <int N, int R0, int R1, int R2>
__device__ __forceinline__ float idft_slice_fixed(const ClientParams &cp, const cf (&sv)[(N + 63) / 64], cf *buf, const cf *Wn, int lane);
template <int N, int R0, int R1, int R2>
__device__ __forceinline__ float idft_item_fixed(const DemodArgs &a, const ClientParams &cp, int f, cf *buf, const cf *Wn, int lane) {
    cf sv[(N + 63) / 64];  // loads first, LDS after
    idft_load_slice<N>(a, cp, f, lane, sv);
    return idft_slice_fixed<N, R0, R1, R2>(cp, sv, buf, Wn, lane);
}
template <int N, int R0, int R1, int R2>
__device__ __forceinline__ float idft_slice_fixed(const ClientParams &cp, const cf (&sv)[(N + 63) / 64], cf *buf, const cf *Wn, int lane) {
    const int len = cp.r - cp.l;
    const int m = cp.m_floor - cp.l;  // audio_m
    constexpr int NR = (N + 63) / 64;
#pragma unroll
    for (int u = 0; u < NR; u++) {
        const int i = lane + 64 * u;
        if (i < N) buf[i] = make_float2(0.f, 0.f);
    }
    asm volatile("" ::: "memory");  // zero-fill before any lane's scatter (compiler order, see above)
    float pw = 0.f;
#pragma unroll
    for (int u = 0; u < NR; u++) {
        const int t = lane + 64 * u;
        if (t < len) {
            const cf v = sv[u];
            pw += fmaf(v.x, v.x, v.y * v.y);
            // USB / LSB are c2r transforms (fftwf_plan_dft_c2r_1d, src/signal.cpp:138, 154): only bins 0..N/2 of the input
            // array are read, Im of bins 0 and N/2 is ignored, the rest is the Hermitian mirror.  The scatter writes a bin
            // AND its mirror (round 5: a second pass over the buffer behind an LDS round trip)
            if (cp.mode < 2) {
                const int idx = cp.mode == 0 ? t - m : m - t;  // USB :125-137 / LSB :139-153
                if (idx == 0 || idx == N / 2) {
                    buf[idx] = make_float2(v.x, 0.f);
                } else if (idx > 0 && idx < N / 2) {
                    buf[idx] = v;
                    buf[N - idx] = make_float2(v.x, -v.y);
                }
            } else {  // AM/FM :175-198
                if (t >= m && t < m + N / 2) buf[t - m] = v;
                if (t >= m - N / 2 + 1 && t < m) buf[N - m + t] = v;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pw += __shfl_xor(pw, d, 64);
    wave_lds_sync();
    idft_stage_fixed<N, R0, 1>(buf, Wn, lane);
    wave_lds_sync();
    idft_stage_fixed<N, R1, R0>(buf, Wn, lane);
    wave_lds_sync();
    if constexpr (R2 > 1) {
        idft_stage_fixed<N, R2, R0 * R1>(buf, Wn, lane);
        wave_lds_sync();
    }
    return pw;
}

//   grid = ceil(nact * nframes / W), W = blockDim.x / 64 items per work-group;
//   dynamic LDS = (1 + W) * N * 8 bytes
template <int N, int R0, int R1, int R2>
#ifndef PSDR_IDFT_WPE
#define PSDR_IDFT_WPE 5
#endif
__global__ __launch_bounds__(256, PSDR_IDFT_WPE) void k_demod_idft_fixed(DemodArgs a, int nact) {
    static_assert(R0 * R1 * R2 == N, "plan");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
    cf *Wn = reinterpret_cast<cf *>(smem);
    for (int i = threadIdx.x; i < N; i += blockDim.x) Wn[i] = a.Wn[i];
    __syncthreads();  // the only work-group barrier: the shared twiddle table
    // everything about the item is wave-uniform: keep it in scalar registers (the mode
    // branches become scalar branches, the range tests compare against scalars)
    const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * W + wv);
    if (item >= nact * a.nframes) return;
    const int ci = item / a.nframes, f = item - ci * a.nframes;
    ClientParams cp = a.clients[ci];
    cp.l = __builtin_amdgcn_readfirstlane(cp.l);
    cp.r = __builtin_amdgcn_readfirstlane(cp.r);
    cp.m_floor = __builtin_amdgcn_readfirstlane(cp.m_floor);
    cp.mode = __builtin_amdgcn_readfirstlane(cp.mode);
    cp.slot = __builtin_amdgcn_readfirstlane(cp.slot);
    const unsigned long long frame_num = a.first_frame_num + (unsigned long long)f;
    cf *buf = Wn + N + (size_t)wv * N;

    const float pw = idft_item_fixed<N, R0, R1, R2>(a, cp, f, buf, Wn, lane);
    if (lane == 0) a.pwr[(size_t)cp.slot * a.max_batch + f] = pw;
    constexpr int NR = (N + 63) / 64;
    const bool flip = flip_frame(frame_num, cp.m_floor, a.is_real);
    const float sg = flip ? -1.f : 1.f;
    cf *yp = a.ypost + ((size_t)cp.slot * a.max_batch + f) * N;
    // (mode is a scalar: one plain loop per mode.  The single unrolled loop with the three-way
    // per-lane select inside was miscompiled by this toolchain - the real part of the last,
    // partial round was read through a stale address register in the complex modes.)
    if (cp.mode == 0) {
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int jx = lane + 64 * u;
            if (jx < N) yp[jx] = make_float2(buf[jx].x * sg, 0.f);
        }
    } else if (cp.mode == 1) {
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int jx = lane + 64 * u;
            if (jx < N) yp[jx] = make_float2(buf[N - 1 - jx].x * sg, 0.f);  // std::reverse :155
        }
    } else {
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int jx = lane + 64 * u;
            if (jx < N) yp[jx] = make_float2(buf[jx].x * sg, buf[jx].y * sg);
        }
    }
}

// one WAVE per (client, group of PSDR_OLA_FG consecutive frames): no shared memory, no barrier (NaN flag by
// wave vote).  A wave's life is three dependent round trips to memory (client parameters, then the two
// halves it adds, then the stores) whatever it does in between - with one frame per wave 65 536 waves of
// ~10 us each passed through the few wave slots the FFT passes leave free (256 clients x 256 frames:
// 700-800 us); a group of frames shares the parameter load and has all its loads in flight together.
// grid = ceil(nact * ceil(nframes / FG) / 4) work-groups of 256 threads
#ifndef PSDR_OLA_FG
#define PSDR_OLA_FG 8
#endif
__global__ __launch_bounds__(256) void k_demod_ola(DemodArgs a, int nact) {
    constexpr int FG = PSDR_OLA_FG;
    const int n = a.n, h = n / 2, tid = threadIdx.x & 63, NT = 64;
    const int F = a.nframes, ngrp = (F + FG - 1) / FG;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nact * ngrp) return;
    const int ci = item / ngrp;
    const ClientParams cp = a.clients[ci];
    const int f0 = (item - ci * ngrp) * FG;
    const size_t srow = (size_t)cp.slot;
    const cf *yp = a.ypost + (srow * a.max_batch) * n;  // this client's frames
    const int cur = cp.state_cur, nxt = cur ^ 1;
    const float *rp_old = a.real_prev + ((size_t)cur * a.slots + srow) * h;
    float *rp_new = a.real_prev + ((size_t)nxt * a.slots + srow) * h;
    const cf *bt_old = a.bb_tail + ((size_t)cur * a.slots + srow) * h;
    cf *bt_new = a.bb_tail + ((size_t)nxt * a.slots + srow) * h;
#pragma unroll
    for (int g = 0; g < FG; g++) {
        const int f = f0 + g;
        if (f >= F) break;
        const cf *y = yp + (size_t)f * n;
        float *out = a.audio + (srow * a.max_batch + f) * h;
        int s_nan = 0;  // per lane; combined by a wave vote at the end
        const bool last = (f == F - 1);
        if (cp.mode < 2) {
            // the half to add is the second half of the latest EARLIER frame that survived the NaN guard: a dropped
            // frame throws at src/signal.cpp:266-271, before audio_real_prev is replaced (:273-275).  With finite values
            // nothing is ever dropped, and a NaN input sample makes a transform NaN throughout: "frame g was dropped" is
            // then read off the half that is loaded anyway.  Anything else non-finite: the slot is marked and
            // k_demod_ola_seq walks its batch in frame order afterwards (DemodArgs::ssb_mark).
            {
                int bad = 0;
                for (int j = tid; j < n; j += NT) bad |= not_finite(y[j].x) ? 1 : 0;
                if (f == 0)
                    for (int j = tid; j < h; j += NT) bad |= not_finite(rp_old[j]) ? 1 : 0;
                if (__any(bad) && tid == 0) a.ssb_mark[srow] = a.mark_epoch;
            }
            int g = f - 1;
            while (g >= 0) {
                int d = 0;
                for (int j = tid; j < h; j += NT) d |= isnan(yp[(size_t)g * n + h + j].x) ? 1 : 0;
                if (!__any(d)) break;
                g--;
            }
            for (int j = tid; j < h; j += NT) {
                const float prev = (g < 0) ? rp_old[j] : yp[(size_t)g * n + h + j].x;
                const float v = y[j].x + prev;  // dsp_add_float :171
                out[j] = v;
                if (isnan(v)) s_nan = 1;
            }
            if (last) {
                const int dropped = __any(s_nan);
                for (int j = tid; j < h; j += NT) {
                    // :273-275, unless this frame was dropped: then the tail it would have added stays
                    rp_new[j] = dropped ? ((g < 0) ? rp_old[j] : yp[(size_t)g * n + h + j].x) : y[h + j].x;
                    bt_new[j] = bt_old[j];
                }
                if (tid == 0) a.bb_last[(size_t)nxt * a.slots + srow] = a.bb_last[(size_t)cur * a.slots + srow];
            }
        } else {
            for (int j = tid; j < h; j += NT) {
                const cf pv = (f == 0) ? bt_old[j] : yp[(size_t)(f - 1) * n + h + j];
                const cf b = make_float2(y[j].x + pv.x, y[j].y + pv.y);  // dsp_add_complex :235
                float v;
                if (cp.mode == 2) {
                    v = sqrtf(fmaf(b.x, b.x, b.y * b.y));  // dsp_am_demod
                } else {
                    cf pr;
                    if (j > 0) {
                        const cf pv1 = (f == 0) ? bt_old[j - 1] : yp[(size_t)(f - 1) * n + h + j - 1];
                        pr = make_float2(y[j - 1].x + pv1.x, y[j - 1].y + pv1.y);
                    } else if (f == 0) {
                        pr = a.bb_last[(size_t)cur * a.slots + srow];
                    } else {
                        // B'_{f-1}[h-1] = y_{f-1}[h-1] + (tail of frame f-2, or the carried tail)
                        const cf y1 = yp[(size_t)(f - 1) * n + h - 1];
                        const cf t1 = (f == 1) ? bt_old[h - 1] : yp[(size_t)(f - 2) * n + n - 1];
                        pr = make_float2(y1.x + t1.x, y1.y + t1.y);
                    }
                    // arg(b * conj(pr)), src/utils/dsp.cpp:32
                    const float re = fmaf(b.x, pr.x, b.y * pr.y);
                    const float im = fmaf(b.x, -pr.y, b.y * pr.x);
                    v = atan2f(im, re);
                }
                out[j] = v;
                if (isnan(v)) s_nan = 1;
                if (last) {
                    bt_new[j] = y[h + j];  // :200-203 (second half kept for the next frame)
                    rp_new[j] = rp_old[j];
                    if (j == h - 1) a.bb_last[(size_t)nxt * a.slots + srow] = b;  // `prev` of :200
                }
            }
        }
        const int any_nan = __any(s_nan);
        if (tid == 0) a.nan_flags[srow * a.max_batch + f] = any_nan ? 1 : 0;
    }
}

// The NaN guard of USB / LSB as the recurrence it is (src/signal.cpp:266-275), for the slots k_demod_ola marked: one wave
// per client walks the batch in frame order - v = y_f[0..h) + prev; dropped iff v has a NaN; prev = y_f[h..n) unless dropped -
// and writes audio, flags and the carried tail again.  grid = ceil(nact / 4) work-groups of 256 threads.
__global__ __launch_bounds__(256) void k_demod_ola_seq(DemodArgs a, int nact) {
    const int n = a.n, h = n / 2, tid = threadIdx.x & 63, NT = 64;
    const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= nact) return;
    const ClientParams cp = a.clients[ci];
    const size_t srow = (size_t)cp.slot;
    if (cp.mode >= 2 || a.ssb_mark[srow] != a.mark_epoch) return;
    const cf *yp = a.ypost + (srow * a.max_batch) * n;
    const int cur = cp.state_cur, nxt = cur ^ 1;
    const float *rp_old = a.real_prev + ((size_t)cur * a.slots + srow) * h;
    float *rp_new = a.real_prev + ((size_t)nxt * a.slots + srow) * h;
    int g = -1;  // the latest frame that survived (-1: the carried tail)
    for (int f = 0; f < a.nframes; f++) {
        const cf *y = yp + (size_t)f * n;
        float *out = a.audio + (srow * a.max_batch + f) * h;
        int s_nan = 0;
        for (int j = tid; j < h; j += NT) {
            const float prev = (g < 0) ? rp_old[j] : yp[(size_t)g * n + h + j].x;
            const float v = y[j].x + prev;  // dsp_add_float :171
            out[j] = v;
            if (isnan(v)) s_nan = 1;
        }
        const int dropped = __any(s_nan);
        if (tid == 0) a.nan_flags[srow * a.max_batch + f] = dropped ? 1 : 0;
        if (!dropped) g = f;
    }
    for (int j = tid; j < h; j += NT) rp_new[j] = (g < 0) ? rp_old[j] : yp[(size_t)g * n + h + j].x;  // :273-275
}

// ---- transform + overlap-add + demodulation in ONE kernel (compile-time plans) -------------------------------
// One wave walks a CHAIN of K consecutive frames of one client: the second half of frame f-1's transform stays in
// registers until frame f adds it (src/signal.cpp:171-172, 235-237), so the n complex values per item that
// k_demod_idft_fixed writes for k_demod_ola to read back (2.9 KB at n = 360: 190 MB per step at 256 clients x 256
// frames) never leave the CU, and k_demod_ola's three dependent round trips per wave disappear with it.  A chain
// that does not start the batch first repeats the transform of the frame before it (FM: of the two frames before
// it - its first sample needs B'_{f0-1}[h-1] = y_{f0-1}[h-1] + y_{f0-2}[n-1]) as warm-up: 1 or 2 transforms more per K.
// Same operations in the same order as the two-kernel path (explicit __fmul_rn / __fadd_rn where the fused form
// would otherwise let the compiler contract what used to be split across two kernels): bit-identical outputs.
//   grid = ceil(nact * ceil(nframes / K) / W), W = blockDim.x / 64; dynamic LDS = (1 + W) * N * 8 bytes
template <int N, int R0, int R1, int R2>
__global__ __launch_bounds__(256, N <= 512 ? PSDR_IDFT_WPE : PSDR_IDFT_WPE - 1) void k_demod_chain_fixed(DemodArgs a, int nact, int K) {
    static_assert(R0 * R1 * R2 == N, "plan");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int h = N / 2, NH = (h + 63) / 64;
    const int lane_ = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
    cf *Wn = reinterpret_cast<cf *>(smem);
    for (int i = threadIdx.x; i < N; i += blockDim.x) Wn[i] = a.Wn[i];
    __syncthreads();
    const int F = a.nframes, nch = (F + K - 1) / K;
    const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * W + wv);
    if (item >= nact * nch) return;
    const int ci = item / nch, f0 = (item - ci * nch) * K;
    const int f1 = f0 + K < F ? f0 + K : F;
    ClientParams cp = a.clients[ci];
    // replay (DemodArgs::ssb_mark): the whole batch as ONE chain (K >= F: no warm-up frame, every decision in frame order),
    // for the USB / LSB slots the first launch marked
    if (a.replay && (cp.mode >= 2 || a.ssb_mark[cp.slot] != a.mark_epoch)) return;
    cp.l = __builtin_amdgcn_readfirstlane(cp.l);
    cp.r = __builtin_amdgcn_readfirstlane(cp.r);
    cp.m_floor = __builtin_amdgcn_readfirstlane(cp.m_floor);
    cp.mode = __builtin_amdgcn_readfirstlane(cp.mode);
    cp.slot = __builtin_amdgcn_readfirstlane(cp.slot);
    cp.state_cur = __builtin_amdgcn_readfirstlane(cp.state_cur);
    cf *buf = Wn + N + (size_t)wv * N;
    const size_t srow = (size_t)cp.slot;
    const int cur = cp.state_cur, nxt = cur ^ 1;
    const float *rp_old = a.real_prev + ((size_t)cur * a.slots + srow) * h;
    float *rp_new = a.real_prev + ((size_t)nxt * a.slots + srow) * h;
    const cf *bt_old = a.bb_tail + ((size_t)cur * a.slots + srow) * h;
    cf *bt_new = a.bb_tail + ((size_t)nxt * a.slots + srow) * h;
    const bool ssb = cp.mode < 2;
    auto sign_of = [&](int f) { return flip_frame(a.first_frame_num + (unsigned long long)f, cp.m_floor, a.is_real) ? -1.f : 1.f; };
    cf tail[NH];                         // y_{f-1}[h + j], j = lane + 64 u
    cf blast = make_float2(0.f, 0.f);    // FM: B'_{f-1}[h-1] (wave-uniform)
    // warm-up frames (transformed, nothing written): one before the chain, two for FM; ONE loop body for both kinds
    // of frame (three inlined copies of the transform cost 100 VGPRs more)
    const int fs = f0 == 0 ? 0 : (f0 - (cp.mode == 3 ? 2 : 1) > 0 ? f0 - (cp.mode == 3 ? 2 : 1) : 0);
#pragma unroll
    for (int u = 0; u < NH; u++) {
        const int j = lane_ + 64 * u;
        tail[u] = make_float2(0.f, 0.f);
        // the batch's first frame: the carried state (src/signal.h:86-101)
        if (fs == 0 && j < h) tail[u] = ssb ? make_float2(rp_old[j], 0.f) : bt_old[j];
    }
    if (fs == 0 && cp.mode == 3) blast = a.bb_last[(size_t)cur * a.slots + srow];
    int bad = 0;  // USB / LSB: a non-finite value seen (per lane)
    if (ssb) {
#pragma unroll
        for (int u = 0; u < NH; u++) bad |= not_finite(tail[u].x) ? 1 : 0;
    }
    int f = fs;
    // PSDR_DEMOD_PREFETCH=1: the slice of the NEXT frame is fetched while this frame is transformed (a wave that walks its
    // chain frame by frame has one frame's loads in flight at a time).  Measured on the round-4 build, same box, three
    // interleaved repetitions: 256 clients on cfg2's stream 94.75-94.79 GS/s with it, 94.65-95.06 without; cfg3 and the
    // cfg5 share inside their spread; n = 720 with it (=2: 128 VGPRs + 48 bytes of scratch) -1 %.  The chain kernel's time
    // beside the passes is not its own latency (docs/history.md 5.4): off.
#ifndef PSDR_DEMOD_PREFETCH
#define PSDR_DEMOD_PREFETCH 0
#endif
    constexpr int NR = (N + 63) / 64;
    cf svn[NR];
    int fpre = -1;  // frame whose slice svn holds
    constexpr int HO = NR < 3 ? NR : (N <= 512 ? 3 : 6);
    unsigned so[HO];  // where the slice's first 64 HO bins sit inside a frame: the same for every frame of the chain
    idft_slice_offsets<N, HO>(a, cp, lane_, so);
    while (f < f1) {
        const bool emit = f >= f0;
        // (an opaque copy per iteration: with the loop-invariant lane the compiler keeps every address of every stage
        // in registers across the loop - 100 VGPRs more than the transform itself needs, or 300-700 bytes of scratch)
        int ln = lane_;
        asm volatile("" : "+v"(ln));
        const int lane = ln;
        float pw;
        if (PSDR_DEMOD_PREFETCH && (N <= 512 || PSDR_DEMOD_PREFETCH > 1)) {  // (n = 720: 128 VGPRs + 48 bytes of scratch with it)
            cf sv[NR];
            if (f == fpre) {
#pragma unroll
                for (int u = 0; u < NR; u++) sv[u] = svn[u];
            } else {
                idft_load_slice_at<N, HO>(a, cp, f, so, lane, sv);  // the chain's first frame; a warm-up frame looked for further back
            }
            if (f + 1 < f1) {
                idft_load_slice_at<N, HO>(a, cp, f + 1, so, lane, svn);
                fpre = f + 1;
            }
            pw = idft_slice_fixed<N, R0, R1, R2>(cp, sv, buf, Wn, lane);
        } else {
            cf sv[NR];  // loads first, LDS after
            idft_load_slice_at<N, HO>(a, cp, f, so, lane, sv);
            pw = idft_slice_fixed<N, R0, R1, R2>(cp, sv, buf, Wn, lane);
        }
        if (emit && lane == 0) a.pwr[srow * a.max_batch + f] = pw;
        const float sg = sign_of(f);
        float *out = a.audio + (srow * a.max_batch + f) * h;
        const bool last = (f == F - 1);
        int s_nan = 0;
        cf carry = blast;  // FM: b of sample j-1 for the lane that holds j = 64 u (lane 0)
        // y_f[j] and y_f[h + j] of the two-kernel path: the transform output after reversal (LSB) and sign flip.
        // (mode is a scalar: one plain loop per mode - the three-way select inside one unrolled loop is miscompiled
        // by this toolchain, see k_demod_idft_fixed)
        cf yv[NH], yn[NH];
        if (cp.mode == 0) {
#pragma unroll
            for (int u = 0; u < NH; u++) {
                const int j = lane + 64 * u;
                yv[u] = yn[u] = make_float2(0.f, 0.f);
                if (j < h) {
                    yv[u].x = __fmul_rn(buf[j].x, sg);
                    yn[u].x = __fmul_rn(buf[h + j].x, sg);
                }
            }
        } else if (cp.mode == 1) {
#pragma unroll
            for (int u = 0; u < NH; u++) {
                const int j = lane + 64 * u;
                yv[u] = yn[u] = make_float2(0.f, 0.f);
                if (j < h) {
                    yv[u].x = __fmul_rn(buf[N - 1 - j].x, sg);  // std::reverse :155
                    yn[u].x = __fmul_rn(buf[N - 1 - h - j].x, sg);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NH; u++) {
                const int j = lane + 64 * u;
                yv[u] = yn[u] = make_float2(0.f, 0.f);
                if (j < h) {
                    const cf v0 = buf[j], v1 = buf[h + j];
                    yv[u] = make_float2(__fmul_rn(v0.x, sg), __fmul_rn(v0.y, sg));
                    yn[u] = make_float2(__fmul_rn(v1.x, sg), __fmul_rn(v1.y, sg));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NH; u++) {
            const int j = lane + 64 * u;
            const bool ok = j < h;
            const cf y = yv[u], ynext = yn[u];
            float v = 0.f;
            cf b = make_float2(0.f, 0.f);
            if (ssb) {
                v = __fadd_rn(y.x, tail[u].x);  // dsp_add_float :171
            } else {
                b = make_float2(__fadd_rn(y.x, tail[u].x), __fadd_rn(y.y, tail[u].y));  // dsp_add_complex :235
                if (cp.mode == 2) {
                    v = sqrtf(fmaf(b.x, b.x, b.y * b.y));  // dsp_am_demod
                } else {
                    // previous sample: lane-1 of this round; lane 0 takes the last sample of the round before
                    cf pr = make_float2(__shfl_up(b.x, 1, 64), __shfl_up(b.y, 1, 64));
                    if (lane == 0) pr = carry;
                    // arg(b * conj(pr)), src/utils/dsp.cpp:32
                    const float re = fmaf(b.x, pr.x, b.y * pr.y);
                    const float im = fmaf(b.x, -pr.y, b.y * pr.x);
                    v = atan2f(im, re);
                    carry = make_float2(__shfl(b.x, 63, 64), __shfl(b.y, 63, 64));
                    if (u == (h - 1) / 64) blast = make_float2(__shfl(b.x, (h - 1) & 63, 64), __shfl(b.y, (h - 1) & 63, 64));
                }
            }
            if (ok && isnan(v)) s_nan = 1;
            if (ok && emit) {
                out[j] = v;
                if (last && !ssb) {  // the state the next batch starts from (:200-203); the other mode family's is kept
                    bt_new[j] = ynext;
                    rp_new[j] = rp_old[j];
                    if (j == h - 1) a.bb_last[(size_t)nxt * a.slots + srow] = b;
                }
            }
            if (!ssb) tail[u] = ynext;  // (:200-203 precede the NaN guard: the complex modes' state always moves)
        }
        const int any_nan = __any(s_nan);
        if (ssb) {
#pragma unroll
            for (int u = 0; u < NH; u++) bad |= (not_finite(yv[u].x) || not_finite(yn[u].x)) ? 1 : 0;
            // A dropped USB / LSB frame throws at src/signal.cpp:266-271, BEFORE audio_real_prev is replaced (:273-275):
            // the next frame adds the tail of the latest frame that survived.  (With finite values nothing is dropped; a
            // NaN input sample makes a transform NaN throughout, so for a warm-up frame - whose own tail is unknown here -
            // "y + 0 has a NaN" says the same; any other non-finite value marks the slot for the sequential replay.)
            if (!emit) {
                // the warm-up frame of a chain that does not start the batch: if it was dropped, look further back
                if (any_nan && f > 0) {
                    f--;
                    wave_lds_sync();
                    continue;
                }
#pragma unroll
                for (int u = 0; u < NH; u++) {
                    const int j = lane + 64 * u;
                    tail[u] = yn[u];
                    if (any_nan && j < h) tail[u] = make_float2(rp_old[j], 0.f);  // nothing survived before the chain: the carried tail
                }
                f = f0;
                wave_lds_sync();
                continue;
            }
#pragma unroll
            for (int u = 0; u < NH; u++) {
                const int j = lane + 64 * u;
                if (!any_nan) tail[u] = yn[u];
                if (last && j < h) {  // :273-275
                    rp_new[j] = tail[u].x;
                    bt_new[j] = bt_old[j];
                }
            }
            if (last && lane == 0) a.bb_last[(size_t)nxt * a.slots + srow] = a.bb_last[(size_t)cur * a.slots + srow];
        }
        if (emit && lane == 0) a.nan_flags[srow * a.max_batch + f] = any_nan ? 1 : 0;
        wave_lds_sync();  // buf is read out: the next frame's transform may overwrite it
        f++;
    }
    if (ssb && !a.replay && __any(bad) && lane_ == 0) a.ssb_mark[srow] = a.mark_epoch;
}

}  // namespace psdr
