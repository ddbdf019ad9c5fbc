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
    int Tmax = T;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, d, 64));
    const float fD = (float)D;
    const size_t hist = (size_t)(a.L - 1);
    for (int t = 0; t < Tmax; t++) {
        if (t < T) {
            const float x = a.v0[(size_t)t * a.slots + slot];
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
            a.v1[(hist + t) * a.slots + slot] = __fsub_rn(rx[back * 64 + lane], m2);
        }
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
    if (blockIdx.z == 0) {
        for (int r = r0; r < r1; r++) {
            m = fmaxf(m, fabsf(a.v1[(size_t)r * a.slots + slot]));
            a.P[(size_t)r * a.slots + slot] = m;
        }
    } else {
        for (int r = r1 - 1; r >= r0; r--) {
            m = fmaxf(m, fabsf(a.v1[(size_t)r * a.slots + slot]));
            a.S[(size_t)r * a.slots + slot] = m;
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
    for (int t = 0; t < T; t++) {
        float y = 0.f;
        if (n0 + t + 1 >= L) {  // the look-ahead buffer is full: sample t-L+1.. of the stream
            const float cur = a.v1[(size_t)t * a.slots + slot];
            const float peak = fmaxf(a.S[(size_t)t * a.slots + slot], a.P[(size_t)(t + L - 1) * a.slots + slot]);
            const float want = __fdiv_rn(a.desired, __fadd_rn(peak, 1e-10f));
            if (want < gain)
                gain = __fmaf_rn(-a.attack, __fsub_rn(gain, want), gain);
            else
                gain = __fmaf_rn(a.release, __fsub_rn(want, gain), gain);
            y = __fmul_rn(cur, gain);
        }
        // dsp_float_to_int16, src/utils/dsp.cpp:152-165
        int v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
        v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
        a.pcm_t[(size_t)t * a.slots + slot] = v;
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
    for (int r = 0; r < a.L - 1; r++) a.v1[(size_t)r * a.slots + slot] = a.v1[(size_t)(r + T) * a.slots + slot];
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
