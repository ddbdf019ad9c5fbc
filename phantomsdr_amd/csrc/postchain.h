// postchain.h — the post-demodulation chain of AudioClient::send_audio, batched for all
// clients:
//   dc.removeDC      src/signal.cpp:278, DCBlocker / MovingAverage src/utils.h:76-99,139-169
//   agc.process      src/signal.cpp:281, src/utils/audioprocessing.cpp:5-68
//   dsp_float_to_int16 (mult 65536/4)  src/signal.cpp:283-284, src/utils/dsp.cpp:152-165
// Frames dropped by the NaN guard (src/signal.cpp:266-271) never reach the chain: each
// client's stream is the concatenation of its surviving frames.
//
// Three f32 recurrences run along time and are bit-exact only in the reference's order:
//   s1_t = (s1_{t-1} - x_{t-D}) + x_t            first moving average  (m1 = s1 / D)
//   s2_t = (s2_{t-1} - m1_{t-D}) + m1_t          second moving average (out = x_{t-D+1} - s2 / D)
//   g_t  = fma(-a, g - w_t, g) or fma(r, w_t - g, g)        AGC attack / release
// Everything else is a pure function of the streams and runs fully parallel.  The streams are
// time-major [t][slot] with the history they need kept IN FRONT of the new samples (D rows for
// the averages, L-1 rows for the AGC look-ahead), so there are no rings: a lane owns a client,
// a wave walks time, every step is a few coalesced row accesses fetched a block ahead, and the
// loop bodies are just the recurrences.
//   k_pc_gather   audio[slot][frame][j] -> X[D + t][slot], frames with the NaN flag skipped
//   k_pc_ma<0|1>  the two running sums                                     (sequential)
//   k_pc_scan     AGC look-ahead peak: sliding maximum of |x| over L samples (the reference's
//                 monotonic deque) as van Herk prefix / suffix maxima of blocks of L (sequential,
//                 but over (client, block) pairs)
//   k_pc_want     w_t = desired / (peak_t + 1e-10)                          (parallel)
//   k_pc_gain     the gain recurrence                                       (sequential)
//   k_pc_out      delayed sample * gain, int16 conversion                   (parallel)
//   k_pc_history  the last D / L-1 rows become the next batch's history
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
    float *X;                         // [D + max_batch*h][slots]: demodulated audio, rows < D history
    float *M1;                        // [D + max_batch*h][slots]: first moving average, rows < D history
    float *V1;                        // [L-1 + max_batch*h][slots]: DC-blocked stream, rows < L-1 history
    float *P, *S;                     // like V1: prefix / suffix maxima; then S = w_t, P = g_t
    int *pcm_t;                       // [max_batch*h][slots]
    int32_t *pcm;                     // [slots][max_batch][h]
    // carried state
    float *dc_s1, *dc_s2;             // [slots] running sums
    float *agc_gain;
    int *agc_n0;  // samples pushed since the last reset, saturating at L
    int ma_fused;  // k_pc_ma2 keeps M1's history itself (k_pc_history leaves M1 alone)
};

__device__ __forceinline__ unsigned pc_at(const PostArgs &a, int row, int slot) {
    return (unsigned)row * (unsigned)a.slots + (unsigned)slot;
}

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
            // a new client in this slot starts from zero history (the sums are reset in k_pc_ma)
            if (f == 0 && a.clients[c0 + tx].agc_reset == 2)
                for (int r = 0; r < a.D; r++) {
                    a.X[pc_at(a, r, slot)] = 0.f;
                    a.M1[pc_at(a, r, slot)] = 0.f;
                }
        }
    }
    for (int r = ty; r < 64; r += 4) {  // load: lanes along j
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
            if (j0 + jj < a.h) a.X[pc_at(a, a.D + pos + j0 + jj, slot)] = tile[jj][tx];
}

// One moving average (MovingAverage::insert, src/utils.h:84-93: sum -= oldest; push; sum += val).
//   SECOND = false: in = X,  sum = s1, writes M1[D + t] = s1 / D
//   SECOND = true : in = M1, sum = s2, writes V1[L-1 + t] = X[t + 1] - s2 / D
//                   (getLatest(delay - 1) = x_{t-D+1}, src/utils.h:160-166)
// lane = client; the evicted value of step t is row t, the inserted one row D + t.
// POW2: D is a power of two (x / 2^k == x * 2^-k exactly: no division in the loop)
template <bool SECOND, bool POW2>
__global__ __launch_bounds__(64) void k_pc_ma(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const ClientParams cp = a.clients[ci];
    const int slot = cp.slot, D = a.D;
    const bool fresh = cp.agc_reset == 2;  // a new client in this slot: zero sums (k_pc_gather zeroed the history)
    const int T = a.len[slot];
    const float *__restrict__ in = SECOND ? a.M1 : a.X;
    const float *__restrict__ X = a.X;
    float *__restrict__ out = SECOND ? a.V1 : a.M1;
    const int orow = SECOND ? a.L - 1 : D;
    float s = fresh ? 0.f : (SECOND ? a.dc_s2 : a.dc_s1)[slot];
    const float fD = (float)D, rD = 1.0f / fD;
    constexpr int KB = 16;
    auto fetch = [&](float (&ev)[KB], float (&nw)[KB], float (&xd)[KB], int t0) {
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const int t = t0 + i;
            ev[i] = in[pc_at(a, t, slot)];
            nw[i] = in[pc_at(a, D + t, slot)];
            if (SECOND) xd[i] = X[pc_at(a, t + 1, slot)];
        }
    };
    auto block = [&](const float (&ev)[KB], const float (&nw)[KB], const float (&xd)[KB], int t0) {
        float o[KB];
#pragma unroll
        for (int i = 0; i < KB; i++) {
            s = __fadd_rn(s, -ev[i]);
            s = __fadd_rn(s, nw[i]);
            const float m = POW2 ? __fmul_rn(s, rD) : __fdiv_rn(s, fD);
            o[i] = SECOND ? __fsub_rn(xd[i], m) : m;
        }
#pragma unroll
        for (int i = 0; i < KB; i++) out[pc_at(a, orow + t0 + i, slot)] = o[i];
    };
    const int nblk = T / KB;
    if (nblk > 0) {  // the loads of block b+1 in flight while block b runs (two register sets)
        float ea[KB], na[KB], xa[KB], eb[KB], nb[KB], xb[KB];
        fetch(ea, na, xa, 0);
        int b = 0;
        for (; b + 1 < nblk; b += 2) {
            fetch(eb, nb, xb, (b + 1) * KB);
            block(ea, na, xa, b * KB);
            if (b + 2 < nblk) fetch(ea, na, xa, (b + 2) * KB);
            block(eb, nb, xb, (b + 1) * KB);
        }
        if (b < nblk) block(ea, na, xa, b * KB);
    }
    for (int t = nblk * KB; t < T; t++) {
        s = __fadd_rn(s, -in[pc_at(a, t, slot)]);
        s = __fadd_rn(s, in[pc_at(a, D + t, slot)]);
        const float m = POW2 ? __fmul_rn(s, rD) : __fdiv_rn(s, fD);
        if (SECOND) {
            out[pc_at(a, orow + t, slot)] = __fsub_rn(X[pc_at(a, t + 1, slot)], m);
        } else {
            out[pc_at(a, orow + t, slot)] = m;
        }
    }
    (SECOND ? a.dc_s2 : a.dc_s1)[slot] = s;
}

// Both moving averages in one loop for D = 32: the m1 values the second average evicts are
// the ones this lane produced two 16-step blocks ago - they stay in registers (two alternating
// sets), M1 only holds the 32 carried values (time order, oldest first).  Two independent
// recurrences per step instead of one: the same time as a single average.  A stream that is
// not a whole number of blocks ends with a partial block whose surplus steps are predicated off.
__global__ __launch_bounds__(64) void k_pc_ma2(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const ClientParams cp = a.clients[ci];
    const int slot = cp.slot;
    constexpr int KB = 16, D = 32;
    const bool fresh = cp.agc_reset == 2;  // zero sums (k_pc_gather zeroed the history rows)
    const int T = a.len[slot];
    if (T == 0) return;
    const int nfull = T / KB, rem = T - nfull * KB;
    const float *__restrict__ X = a.X;
    float *__restrict__ M1 = a.M1;
    float *__restrict__ V1 = a.V1;
    float s1 = fresh ? 0.f : a.dc_s1[slot], s2 = fresh ? 0.f : a.dc_s2[slot];
    const float rD = 1.0f / 32.0f;
    const int orow = a.L - 1;
    float ma[KB], mb[KB];  // m1 of the blocks two and one back (alternating roles)
#pragma unroll
    for (int i = 0; i < KB; i++) {
        ma[i] = M1[pc_at(a, i, slot)];
        mb[i] = M1[pc_at(a, KB + i, slot)];
    }
    auto fetch = [&](float (&ev)[KB], float (&nw)[KB], float &xlast, int t0) {
#pragma unroll
        for (int i = 0; i < KB; i++) {  // (rows past the stream exist: the arrays are padded)
            ev[i] = X[pc_at(a, t0 + i, slot)];
            nw[i] = X[pc_at(a, D + t0 + i, slot)];
        }
        xlast = X[pc_at(a, t0 + KB, slot)];  // x_{t-D+1} of the block's last step
    };
    // m2old: m1 of block b-2 (evicted, then overwritten with block b's m1); valid < KB: only
    // the first `valid` steps exist
    auto block = [&](const float (&ev)[KB], const float (&nw)[KB], float xlast, float (&m2old)[KB], int t0,
                     int valid) {
        float o[KB];
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const bool ok = i < valid;
            const float s1n = __fadd_rn(__fadd_rn(s1, -ev[i]), nw[i]);
            const float m1 = __fmul_rn(s1n, rD);
            const float s2n = __fadd_rn(__fadd_rn(s2, -m2old[i]), m1);
            s1 = ok ? s1n : s1;
            s2 = ok ? s2n : s2;
            m2old[i] = ok ? m1 : m2old[i];
            // getLatest(delay - 1) = x_{t-D+1} = the next step's evictee
            o[i] = __fsub_rn(i + 1 < KB ? ev[i + 1] : xlast, __fmul_rn(s2n, rD));
        }
#pragma unroll
        for (int i = 0; i < KB; i++)
            if (i < valid) V1[pc_at(a, orow + t0 + i, slot)] = o[i];
    };
    const int nblk = nfull + (rem ? 1 : 0);
    float ea[KB], na[KB], eb[KB], nb[KB], xa, xb;
    fetch(ea, na, xa, 0);
    int b = 0;
    for (; b + 1 < nblk; b += 2) {
        fetch(eb, nb, xb, (b + 1) * KB);
        block(ea, na, xa, ma, b * KB, KB);
        if (b + 2 < nblk) fetch(ea, na, xa, (b + 2) * KB);
        block(eb, nb, xb, mb, (b + 1) * KB, (b + 1 == nblk - 1 && rem) ? rem : KB);
    }
    const bool odd = b < nblk;
    if (odd) block(ea, na, xa, ma, b * KB, rem ? rem : KB);
    // the last 32 m1 values in time order.  After the loop the set of the LAST block is `ma` if
    // the block count is odd, else `mb`; with a partial last block (rem steps) that set holds
    // [new m1 x rem | m1 of three blocks ago x (KB-rem)], the other set the block before.
    const int r = rem ? rem : KB;  // valid entries of the last block's set
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const float lastv = odd ? ma[i] : mb[i], prevv = odd ? mb[i] : ma[i];
        // time order of the 32 + (KB - r) candidates: [last set's stale tail (i >= r), prev set, last set's head]
        // rows: stale tail entry i -> row i - r - (KB - r) ... dropped unless it is among the newest 32
        const int row_prev = KB - r + i;        // prev set entry i
        const int row_last = 2 * KB - r + i;    // last set entry i (valid for i < r)
        const int row_stale = i - r;            // last set entry i >= r: m1 of the block before prev
        if (i < r) M1[pc_at(a, row_last, slot)] = lastv;
        M1[pc_at(a, row_prev, slot)] = prevv;
        if (i >= r) M1[pc_at(a, row_stale, slot)] = lastv;
    }
    a.dc_s1[slot] = s1;
    a.dc_s2[slot] = s2;
}

// blockIdx.y = block k of L rows of V1, blockIdx.z = 0: prefix maxima, 1: suffix maxima
__global__ __launch_bounds__(64) void k_pc_scan(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    const int rows = a.L - 1 + a.len[slot];
    const int r0 = blockIdx.y * a.L, r1 = min(r0 + a.L, rows);
    if (r0 >= rows) return;
    float m = 0.f;
    constexpr int KB = 16;
    const float *__restrict__ v1 = a.V1;
    if (blockIdx.z == 0) {
        float *__restrict__ P = a.P;
        int r = r0;
        for (; r + KB <= r1; r += KB) {
            float x[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) x[i] = v1[pc_at(a, r + i, slot)];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                m = fmaxf(m, fabsf(x[i]));
                P[pc_at(a, r + i, slot)] = m;
            }
        }
        for (; r < r1; r++) {
            m = fmaxf(m, fabsf(v1[pc_at(a, r, slot)]));
            P[pc_at(a, r, slot)] = m;
        }
    } else {
        float *__restrict__ S = a.S;
        int r = r1 - 1;
        for (; r - KB + 1 >= r0; r -= KB) {
            float x[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) x[i] = v1[pc_at(a, r - i, slot)];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                m = fmaxf(m, fabsf(x[i]));
                S[pc_at(a, r - i, slot)] = m;
            }
        }
        for (; r >= r0; r--) {
            m = fmaxf(m, fabsf(v1[pc_at(a, r, slot)]));
            S[pc_at(a, r, slot)] = m;
        }
    }
}

// w_t = desired / (peak_t + 1e-10), peak_t = max |V1| over rows [t, t+L-1] = max(S[t], P[t+L-1]);
// in place into S[t] (only this thread reads S[t]).  256 threads = 4 rows x 64 clients.
__global__ __launch_bounds__(256) void k_pc_want(PostArgs a) {
    const int ci = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    if (t >= a.len[slot]) return;
    const float peak = fmaxf(a.S[pc_at(a, t, slot)], a.P[pc_at(a, t + a.L - 1, slot)]);
    a.S[pc_at(a, t, slot)] = __fdiv_rn(a.desired, __fadd_rn(peak, 1e-10f));
}

// the gain recurrence (src/utils/audioprocessing.cpp:55-66); g_t -> P[t] (0 while the
// look-ahead buffer is still filling: the reference outputs 0 there and leaves the gain alone;
// an active gain is never 0: w_t > 0)
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
    const float *__restrict__ W = a.S;
    float *__restrict__ G = a.P;
    const float natt = -a.attack, rel = a.release;
    auto fetch = [&](float (&w)[KB], int t0) {
#pragma unroll
        for (int i = 0; i < KB; i++) w[i] = W[pc_at(a, t0 + i, slot)];
    };
    auto step = [&](float w, int t) -> float {
        if (n0 + t + 1 < L) return 0.f;  // the look-ahead buffer is not full yet
        const bool att = w < gain;
        gain = __fmaf_rn(att ? natt : rel, att ? __fsub_rn(gain, w) : __fsub_rn(w, gain), gain);
        return gain;
    };
    auto block = [&](const float (&w)[KB], int t0) {
        float g[KB];
#pragma unroll
        for (int i = 0; i < KB; i++) g[i] = step(w[i], t0 + i);
#pragma unroll
        for (int i = 0; i < KB; i++) G[pc_at(a, t0 + i, slot)] = g[i];
    };
    const int nblk = T / KB;
    if (nblk > 0) {
        float wa[KB], wb[KB];
        fetch(wa, 0);
        int b = 0;
        for (; b + 1 < nblk; b += 2) {
            fetch(wb, (b + 1) * KB);
            block(wa, b * KB);
            if (b + 2 < nblk) fetch(wa, (b + 2) * KB);
            block(wb, (b + 1) * KB);
        }
        if (b < nblk) block(wa, b * KB);
    }
    for (int t = nblk * KB; t < T; t++) G[pc_at(a, t, slot)] = step(W[pc_at(a, t, slot)], t);
    a.agc_gain[slot] = gain;
    a.agc_n0[slot] = min(n0 + T, L);
}

// current_sample * gain (row t of V1 is the oldest sample of the look-ahead window; gain 0 =
// buffer still filling -> 0) and dsp_float_to_int16 (src/utils/dsp.cpp:152-165)
__global__ __launch_bounds__(256) void k_pc_out(PostArgs a) {
    const int ci = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    if (t >= a.len[slot]) return;
    const float g = a.P[pc_at(a, t, slot)];
    const float y = g == 0.f ? 0.f : __fmul_rn(a.V1[pc_at(a, t, slot)], g);
    int v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
    v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
    a.pcm_t[pc_at(a, t, slot)] = v;
}

// the last D rows of X / M1 and the last L-1 rows of V1 become the history rows of the next
// batch (in place, ascending: the reads stay ahead of the writes); blockIdx.y: 0 = X and M1, 1 = V1
__global__ __launch_bounds__(64) void k_pc_history(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const int slot = a.clients[ci].slot;
    const int T = a.len[slot];
    if (T == 0) return;
    constexpr int KB = 16;
    if (blockIdx.y == 0) {
        for (int r = 0; r < a.D; r++) {  // D is small
            a.X[pc_at(a, r, slot)] = a.X[pc_at(a, r + T, slot)];
            if (!a.ma_fused) a.M1[pc_at(a, r, slot)] = a.M1[pc_at(a, r + T, slot)];
        }
        return;
    }
    int r = 0;
    for (; r + KB <= a.L - 1; r += KB) {  // the KB reads of a block happen before its writes
        float x[KB];
#pragma unroll
        for (int i = 0; i < KB; i++) x[i] = a.V1[pc_at(a, r + i + T, slot)];
#pragma unroll
        for (int i = 0; i < KB; i++) a.V1[pc_at(a, r + i, slot)] = x[i];
    }
    for (; r < a.L - 1; r++) a.V1[pc_at(a, r, slot)] = a.V1[pc_at(a, r + T, slot)];
}

__global__ __launch_bounds__(256) void k_pc_scatter(PostArgs a) {
    __shared__ int tile[32][65];
    const int c0 = blockIdx.x * 64, f = blockIdx.y, j0 = blockIdx.z * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (c0 + tx < a.nact) {
        const int slot = a.clients[c0 + tx].slot;
        const int pos = a.fstart[(size_t)slot * a.max_batch + f];
        for (int jj = ty; jj < 32; jj += 4)
            tile[jj][tx] = (pos >= 0 && j0 + jj < a.h) ? a.pcm_t[pc_at(a, pos + j0 + jj, slot)] : 0;
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
