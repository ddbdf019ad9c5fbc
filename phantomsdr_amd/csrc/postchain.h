// postchain.h — the post-demodulation chain of AudioClient::send_audio, batched for all
// clients:
//   dc.removeDC      src/signal.cpp:278, DCBlocker / MovingAverage src/utils.h:76-99,139-169
//   agc.process      src/signal.cpp:281, src/utils/audioprocessing.cpp:5-68
//   dsp_float_to_int16 (mult 65536/4)  src/signal.cpp:283-284, src/utils/dsp.cpp:152-165
// Frames dropped by the NaN guard (src/signal.cpp:266-271) never reach the chain: each
// client's stream is the concatenation of its surviving frames.
//
// Every stage is a float recurrence along time (running sums, one-pole gain), sequential per
// client and bit-exact only in the reference's order.  The parallelism is ACROSS clients:
// the stream is transposed to time-major [t][slot], a lane owns a client, a wave walks time
// and every step is one coalesced row access.
//   k_pc_gather   audio[slot][frame][j] -> v0[t][slot], frames with the NaN flag skipped
//   k_pc_dc       two cascaded moving averages (f32 running sums, rings in LDS) -> v1
//   k_pc_scan     AGC look-ahead peak: the sliding maximum of |x| over L samples (the
//                 reference's monotonic deque) as van Herk prefix / suffix maxima of blocks of L
//   k_pc_gain     attack / release gain recurrence, delayed sample * gain, int16 conversion
//   k_pc_history  keeps the last L-1 samples of v1 for the next batch
//   k_pc_scatter  pcm[t][slot] -> pcm[slot][frame][j]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod.h"

namespace psdr {

struct PostArgs {
    const ClientParams *clients;  // active clients (compact), .slot = column
    int nact, nframes, max_batch, h;  // h = n/2 samples per frame
    int slots;                        // row pitch of the time-major arrays
    int D, L;                         // DC delay, AGC look-ahead (samples)
    float desired, attack, release;   // AGC
    const float *audio;               // [slots][max_batch][h]
    const int *nan_flags;             // [slots][max_batch]
    int *fstart;                      // [slots][max_batch] stream offset of a frame, -1 = dropped
    int *len;                         // [slots] samples of this batch's stream
    float *v0;                        // [max_batch*h][slots]
    float *v1;                        // [L-1 + max_batch*h][slots], rows < L-1: history
    float *P, *S;                     // prefix / suffix maxima, like v1
    int *pcm_t;                       // [max_batch*h][slots]
    int32_t *pcm;                     // [slots][max_batch][h]
    // carried state
    float *dc_s1, *dc_s2, *dc_rx, *dc_rm;  // [slots], [slots], [D][slots], [D][slots]
    int *dc_head;                          // [slots]
    float *agc_gain;
    int *agc_n0;  // samples pushed since the last reset, saturating at L
};

// tile of 64 clients x 32 samples of one frame through LDS (both accesses coalesced)
__global__ __launch_bounds__(256) void k_pc_gather(PostArgs a) {
    __shared__ float tile[32][65];
    const int c0 = blockIdx.x * 64, f = blockIdx.y, j0 = blockIdx.z * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // stream offset of frame f for client c0 + lane (count of surviving frames before f)
    int pos = -1, slot = 0;
    if (c0 + tx < a.nact) {
        slot = a.clients[c0 + tx].slot;
        const int *nf = a.nan_flags + (size_t)slot * a.max_batch;
        int cnt = 0;
        for (int g = 0; g < f; g++) cnt += nf[g] ? 0 : 1;
        pos = nf[f] ? -1 : cnt * a.h;
        if (blockIdx.z == 0 && ty == 0) {
            a.fstart[(size_t)slot * a.max_batch + f] = pos;
            if (f == a.nframes - 1) a.len[slot] = (cnt + (nf[f] ? 0 : 1)) * a.h;
        }
    }
    // load: lanes along j
    for (int r = ty; r < 64; r += 4) {
        const int ci = c0 + r;
        const int j = j0 + (tx & 31);
        if ((tx < 32) && ci < a.nact && j < a.h) {
            const int sl = a.clients[ci].slot;
            tile[tx & 31][r] = a.audio[((size_t)sl * a.max_batch + f) * a.h + j];
        }
    }
    __syncthreads();
    if (pos >= 0)
        for (int jj = ty; jj < 32; jj += 4)
            if (j0 + jj < a.h) a.v0[(size_t)(pos + j0 + jj) * a.slots + slot] = tile[jj][tx];
}

// lane = client; rings [D][64] in dynamic LDS (2 * D * 64 floats)
__global__ __launch_bounds__(64) void k_pc_dc(PostArgs a) {
    extern __shared__ float rings[];
    float *rx = rings, *rm = rings + (size_t)a.D * 64;
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    const bool on = ci < a.nact;
    const int slot = on ? a.clients[ci].slot : 0;
    const bool fresh = on && a.clients[ci].agc_reset == 2;  // a new client in this slot: zero state
    const int D = a.D;
    float s1 = 0.f, s2 = 0.f;
    int head = 0, T = 0;
    if (on) {
        T = a.len[slot];
        if (!fresh) {
            s1 = a.dc_s1[slot];
            s2 = a.dc_s2[slot];
            head = a.dc_head[slot];
        }
        for (int i = 0; i < D; i++) {
            rx[i * 64 + lane] = fresh ? 0.f : a.dc_rx[(size_t)i * a.slots + slot];
            rm[i * 64 + lane] = fresh ? 0.f : a.dc_rm[(size_t)i * a.slots + slot];
        }
    }
    const float fD = (float)D;
    const bool pow2 = (D & (D - 1)) == 0;  // x / 2^k == x * 2^-k exactly
    const float rD = 1.0f / fD;
    const size_t hist = (size_t)(a.L - 1);
    // Blocks of KB steps: the inputs and the ring entries the block will evict are fetched up
    // front (they are all older than the block: KB < D), so the only latency left inside the
    // block is the f32 recurrence itself.
    constexpr int KB = 16;
    const float *__restrict__ v0 = a.v0;
    float *__restrict__ v1 = a.v1;
    auto fetch = [&](float (&x)[KB], int t0) {
#pragma unroll
        for (int i = 0; i < KB; i++) x[i] = v0[(size_t)(t0 + i) * a.slots + slot];
    };
    auto block = [&](const float (&x)[KB], int t0) {
        float ox[KB + 1], om[KB], out[KB];
        int idx = head;  // entry evicted by step i: head - 1 - i (mod D)
#pragma unroll
        for (int i = 0; i <= KB; i++) {
            idx = idx == 0 ? D - 1 : idx - 1;
            ox[i] = rx[idx * 64 + lane];
            if (i < KB) om[i] = rm[idx * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < KB; i++) {
            s1 = __fadd_rn(s1, -ox[i]);
            s2 = __fadd_rn(s2, -om[i]);
            head = head == 0 ? D - 1 : head - 1;
            rx[head * 64 + lane] = x[i];
            s1 = __fadd_rn(s1, x[i]);
            const float m1 = pow2 ? __fmul_rn(s1, rD) : __fdiv_rn(s1, fD);
            rm[head * 64 + lane] = m1;
            s2 = __fadd_rn(s2, m1);
            const float m2 = pow2 ? __fmul_rn(s2, rD) : __fdiv_rn(s2, fD);
            out[i] = __fsub_rn(ox[i + 1], m2);  // getLatest(delay-1) = the next step's evictee
        }
#pragma unroll
        for (int i = 0; i < KB; i++) v1[(hist + t0 + i) * a.slots + slot] = out[i];
    };
    // whole blocks, the loads of block b+1 in flight while block b runs (two register sets)
    const int nblk = (D > KB) ? T / KB : 0;
    int t0 = 0;
    if (nblk > 0) {
        float xa[KB], xb[KB];
        fetch(xa, 0);
        int b = 0;
        for (; b + 1 < nblk; b += 2) {
            fetch(xb, (b + 1) * KB);
            block(xa, b * KB);
            if (b + 2 < nblk) fetch(xa, (b + 2) * KB);
            block(xb, (b + 1) * KB);
        }
        if (b < nblk) block(xa, b * KB);
        t0 = nblk * KB;
    }
    for (int t = t0; t < T; t++) {  // remainder (and D <= KB): the plain form
        const float x = v0[(size_t)t * a.slots + slot];
        // MovingAverage::insert (src/utils.h:84-93): sum -= oldest; push_front; sum += val
        int oldest = head + D - 1;
        if (oldest >= D) oldest -= D;
        s1 = __fadd_rn(s1, -rx[oldest * 64 + lane]);
        s2 = __fadd_rn(s2, -rm[oldest * 64 + lane]);
        head = oldest;  // (head + D - 1) % D
        rx[head * 64 + lane] = x;
        s1 = __fadd_rn(s1, x);
        const float m1 = __fdiv_rn(s1, fD);
        rm[head * 64 + lane] = m1;
        s2 = __fadd_rn(s2, m1);
        const float m2 = __fdiv_rn(s2, fD);
        int back = head + D - 1;  // getLatest(delay - 1): the oldest after the insert
        if (back >= D) back -= D;
        v1[(hist + t) * a.slots + slot] = __fsub_rn(rx[back * 64 + lane], m2);
    }
    if (on) {
        a.dc_s1[slot] = s1;
        a.dc_s2[slot] = s2;
        a.dc_head[slot] = head;
        for (int i = 0; i < D; i++) {
            a.dc_rx[(size_t)i * a.slots + slot] = rx[i * 64 + lane];
            a.dc_rm[(size_t)i * a.slots + slot] = rm[i * 64 + lane];
        }
    }
}

// blockIdx.y = block k of L rows, blockIdx.z = 0: prefix maxima, 1: suffix maxima
__global__ __launch_bounds__(64) void k_pc_scan(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    const int rows = a.L - 1 + a.len[slot];
    const int r0 = blockIdx.y * a.L, r1 = min(r0 + a.L, rows);
    if (r0 >= rows) return;
    float m = 0.f;
    constexpr int KB = 16;
    const float *__restrict__ v1 = a.v1;
    if (blockIdx.z == 0) {
        float *__restrict__ P = a.P;
        int r = r0;
        for (; r + KB <= r1; r += KB) {
            float x[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) x[i] = v1[(size_t)(r + i) * a.slots + slot];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                m = fmaxf(m, fabsf(x[i]));
                P[(size_t)(r + i) * a.slots + slot] = m;
            }
        }
        for (; r < r1; r++) {
            m = fmaxf(m, fabsf(v1[(size_t)r * a.slots + slot]));
            P[(size_t)r * a.slots + slot] = m;
        }
    } else {
        float *__restrict__ S = a.S;
        int r = r1 - 1;
        for (; r - KB + 1 >= r0; r -= KB) {
            float x[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) x[i] = v1[(size_t)(r - i) * a.slots + slot];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                m = fmaxf(m, fabsf(x[i]));
                S[(size_t)(r - i) * a.slots + slot] = m;
            }
        }
        for (; r >= r0; r--) {
            m = fmaxf(m, fabsf(v1[(size_t)r * a.slots + slot]));
            S[(size_t)r * a.slots + slot] = m;
        }
    }
}

__global__ __launch_bounds__(64) void k_pc_gain(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const ClientParams cp = a.clients[ci];
    const int slot = cp.slot, L = a.L;
    const int T = a.len[slot];
    float gain = a.agc_gain[slot];
    int n0 = a.agc_n0[slot];
    if (cp.agc_reset) {  // AGC::reset() on a demodulation change (src/signal.cpp:322-326)
        gain = 0.f;
        n0 = 0;
    }
    constexpr int KB = 16;
    const float *__restrict__ v1 = a.v1;
    const float *__restrict__ Sx = a.S;
    const float *__restrict__ Px = a.P;
    int *__restrict__ pcm_t = a.pcm_t;
    auto to_i16 = [](float y) {  // dsp_float_to_int16, src/utils/dsp.cpp:152-165
        int v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
        return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
    };
    auto fetch = [&](float (&cur)[KB], float (&s)[KB], float (&p)[KB], int t0) {
#pragma unroll
        for (int i = 0; i < KB; i++) {
            cur[i] = v1[(size_t)(t0 + i) * a.slots + slot];
            s[i] = Sx[(size_t)(t0 + i) * a.slots + slot];
            p[i] = Px[(size_t)(t0 + i + L - 1) * a.slots + slot];
        }
    };
    auto block = [&](const float (&cur)[KB], const float (&s)[KB], const float (&p)[KB], int t0) {
        float want[KB];
#pragma unroll
        for (int i = 0; i < KB; i++)  // everything that does not depend on the gain, up front
            want[i] = __fdiv_rn(a.desired, __fadd_rn(fmaxf(s[i], p[i]), 1e-10f));
#pragma unroll
        for (int i = 0; i < KB; i++) {
            float y = 0.f;
            if (n0 + t0 + i + 1 >= L) {  // the look-ahead buffer is full
                if (want[i] < gain)
                    gain = __fmaf_rn(-a.attack, __fsub_rn(gain, want[i]), gain);
                else
                    gain = __fmaf_rn(a.release, __fsub_rn(want[i], gain), gain);
                y = __fmul_rn(cur[i], gain);
            }
            pcm_t[(size_t)(t0 + i) * a.slots + slot] = to_i16(y);
        }
    };
    const int nblk = T / KB;
    int t = 0;
    if (nblk > 0) {
        float ca[KB], sa[KB], pa[KB], cb[KB], sb[KB], pb[KB];
        fetch(ca, sa, pa, 0);
        int b = 0;
        for (; b + 1 < nblk; b += 2) {
            fetch(cb, sb, pb, (b + 1) * KB);
            block(ca, sa, pa, b * KB);
            if (b + 2 < nblk) fetch(ca, sa, pa, (b + 2) * KB);
            block(cb, sb, pb, (b + 1) * KB);
        }
        if (b < nblk) block(ca, sa, pa, b * KB);
        t = nblk * KB;
    }
    for (; t < T; t++) {
        float y = 0.f;
        if (n0 + t + 1 >= L) {
            const float cur = v1[(size_t)t * a.slots + slot];
            const float peak = fmaxf(Sx[(size_t)t * a.slots + slot], Px[(size_t)(t + L - 1) * a.slots + slot]);
            const float want = __fdiv_rn(a.desired, __fadd_rn(peak, 1e-10f));
            if (want < gain)
                gain = __fmaf_rn(-a.attack, __fsub_rn(gain, want), gain);
            else
                gain = __fmaf_rn(a.release, __fsub_rn(want, gain), gain);
            y = __fmul_rn(cur, gain);
        }
        pcm_t[(size_t)t * a.slots + slot] = to_i16(y);
    }
    a.agc_gain[slot] = gain;
    a.agc_n0[slot] = min(n0 + T, L);
}

// rows [T, T+L-1) of v1 become the history rows [0, L-1) of the next batch (in place:
// ascending order reads ahead of the writes)
__global__ __launch_bounds__(64) void k_pc_history(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    const int T = a.len[slot];
    if (T == 0) return;
    constexpr int KB = 16;
    int r = 0;
    for (; r + KB <= a.L - 1; r += KB) {  // the KB reads of a block happen before its writes
        float x[KB];
#pragma unroll
        for (int i = 0; i < KB; i++) x[i] = a.v1[(size_t)(r + i + T) * a.slots + slot];
#pragma unroll
        for (int i = 0; i < KB; i++) a.v1[(size_t)(r + i) * a.slots + slot] = x[i];
    }
    for (; r < a.L - 1; r++) a.v1[(size_t)r * a.slots + slot] = a.v1[(size_t)(r + T) * a.slots + slot];
}

__global__ __launch_bounds__(256) void k_pc_scatter(PostArgs a) {
    __shared__ int tile[32][65];
    const int c0 = blockIdx.x * 64, f = blockIdx.y, j0 = blockIdx.z * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (c0 + tx < a.nact) {
        const int slot = a.clients[c0 + tx].slot;
        const int pos = a.fstart[(size_t)slot * a.max_batch + f];
        for (int jj = ty; jj < 32; jj += 4)
            tile[jj][tx] = (pos >= 0 && j0 + jj < a.h) ? a.pcm_t[(size_t)(pos + j0 + jj) * a.slots + slot] : 0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int ci = c0 + r;
        const int j = j0 + (tx & 31);
        if ((tx < 32) && ci < a.nact && j < a.h) {
            const int sl = a.clients[ci].slot;
            a.pcm[((size_t)sl * a.max_batch + f) * a.h + j] = tile[tx & 31][r];
        }
    }
}

}  // namespace psdr
