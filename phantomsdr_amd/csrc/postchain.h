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
//   g_t  = g + (w_t < g ? attack : release) * (w_t - g)           AGC attack / release
// Everything else is a pure function of the streams and runs fully parallel.
//
// Layout (round 2): every stream is CLIENT-MAJOR, [slot][row], with the history it needs kept IN
// FRONT of the new samples (D rows for the averages, L-1 rows for the AGC look-ahead) - no rings.
// A lane of a sequential kernel owns a client and walks its stream with 16-byte loads and stores
// (four samples per memory instruction, a whole 16-step block = four loads, prefetched three
// blocks ahead); the parallel kernels put their lanes along time.  Round 1 kept the streams
// time-major with one 4-byte access per sample and one block of prefetch: with a single wave per 64
// clients the two recurrence loops were bound by memory latency (43 and 32 ns per sample, 2.0 and
// 1.5 ms per 256-frame batch) rather than by their 7 and 3 arithmetic operations per sample.
//   k_pc_index    stream offset of every frame of every client (NaN-flagged frames dropped)
//   k_pc_gather   audio[slot][frame][j] -> X[slot][D + t]
//   k_pc_ma2      both running sums in one loop (D = 32)          (sequential)
//   k_pc_ma<0|1>  the two running sums, any D                      (sequential, fallback)
//   k_pc_scan     AGC look-ahead peak: sliding maximum of |x| over L samples (the reference's
//                 monotonic deque) as van Herk prefix / suffix maxima of blocks of L (sequential,
//                 but over (client, block) pairs)
//   k_pc_want     w_t = desired / (peak_t + 1e-10)                 (parallel)
//   k_pc_gain     the gain recurrence                              (sequential)
//   k_pc_out      delayed sample * gain, int16 conversion, straight into pcm[slot][frame][j]
//   k_pc_history  the last D / L-1 rows become the next batch's history
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "butterfly.h"
#include "types.h"

namespace psdr {


typedef float pc_f4 __attribute__((ext_vector_type(4)));
#ifndef PSDR_PC_SETPRIO
#define PSDR_PC_SETPRIO 3
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void pc_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pc_static_for<I + 1, N>(f);
    }
}

// one wave per client: where each surviving frame starts in the client's stream (a ballot per 64
// frames: the count of surviving frames below a lane is a popcount)
__global__ __launch_bounds__(64) void k_pc_index(PostArgs a) {
    const int lane = threadIdx.x;
    const ClientParams cp = a.clients[blockIdx.x];
    const int slot = cp.slot;
    const int *nf = a.nan_flags + (size_t)slot * a.max_batch;
    int *fs = a.fstart + (size_t)slot * a.max_batch;
    int cnt = 0;
    for (int f0 = 0; f0 < a.nframes; f0 += 64) {
        const int f = f0 + lane;
        const bool alive = f < a.nframes && nf[f] == 0 && !cp.paused;  // a paused client: an empty stream, state untouched
        const unsigned long long m = __ballot(alive);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (f < a.nframes) fs[f] = alive ? (cnt + before) * a.h : -1;
        cnt += __popcll(m);
    }
    if (lane == 0) a.len[slot] = cnt * a.h;
    if (cp.agc_reset == 2) {  // a new client in this slot starts from zero history (sums: see k_pc_ma*)
        float *x = a.X + (size_t)slot * a.px, *m1 = a.M1 + (size_t)slot * a.px;
        for (int r = lane; r < a.D; r += 64) x[r] = m1[r] = 0.f;
    }
}

// grid (client, frame): one frame of audio to its place in the stream (contiguous both sides)
__global__ __launch_bounds__(256) void k_pc_gather(PostArgs a) {
    const int slot = a.clients[blockIdx.x].slot, f = blockIdx.y;
    const int pos = a.fstart[(size_t)slot * a.max_batch + f];
    if (pos < 0) return;
    const float *src = a.audio + ((size_t)slot * a.max_batch + f) * a.h;
    float *dst = a.X + (size_t)slot * a.px + a.D + pos;
    for (int j = threadIdx.x; j < a.h; j += blockDim.x) dst[j] = src[j];
}

// One moving average (MovingAverage::insert, src/utils.h:84-93: sum -= oldest; push; sum += val),
// any delay D (the fused kernel below covers the reference's D = 32).
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
    const bool fresh = cp.agc_reset == 2;  // a new client in this slot: zero sums (k_pc_index zeroed the history)
    const int T = a.len[slot];
    const float *__restrict__ in = (SECOND ? a.M1 : a.X) + (size_t)slot * a.px;
    const float *__restrict__ X = a.X + (size_t)slot * a.px;
    float *__restrict__ out = SECOND ? a.V1 + (size_t)slot * a.pv + a.vo + (a.L - 1) : a.M1 + (size_t)slot * a.px + D;
    float s = fresh ? 0.f : (SECOND ? a.dc_s2 : a.dc_s1)[slot];
    const float fD = (float)D, rD = 1.0f / fD;
    for (int t = 0; t < T; t++) {
        s = __fadd_rn(s, -in[t]);
        s = __fadd_rn(s, in[D + t]);
        const float m = POW2 ? __fmul_rn(s, rD) : __fdiv_rn(s, fD);
        out[t] = SECOND ? __fsub_rn(X[t + 1], m) : m;
    }
    (SECOND ? a.dc_s2 : a.dc_s1)[slot] = s;
}

// Both moving averages in one loop for D = 32.  The x values the first average evicts were
// inserted two 16-step blocks earlier and the m1 values the second average evicts were produced two
// blocks earlier: both stay in two alternating register sets, so a 16-step block costs four 16-byte
// loads (the new x), four 16-byte stores and 16 x 7 arithmetic operations.  M1 only holds the 32
// carried values between batches (time order, oldest first).  A stream that is not a whole number
// of blocks ends with a short scalar loop on the in-memory history.
__global__ __launch_bounds__(64) void k_pc_ma2(PostArgs a) {
    const int lane = threadIdx.x, ci = blockIdx.x * 64 + lane;
    if (ci >= a.nact) return;
    const ClientParams cp = a.clients[ci];
    const int slot = cp.slot;
    // RING register sets of one 16-step block each: blocks b-2, b-1 (evicted values), b, and AHEAD = RING - 3 blocks of
    // loads in flight.  Round 3: 13 blocks ahead instead of 3 - vmcnt is ONE in-order counter for loads and stores,
    // so waiting for a load also waits for every store issued before it, and beside the FFT passes a store is
    // acknowledged thousands of cycles after issue: the distance has to cover THAT, not the load latency.  A single-wave
    // kernel has the registers (RING must be even: the two m1 sets alternate).
    constexpr int KB = 16, D = 32, RING = PSDR_PC_RING, AHEAD = RING - 3;
    static_assert(RING % 2 == 0 && RING >= 6, "ring of x blocks");
    // one wave next to the FFT passes' eight issue-bound waves: let it issue first
    __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
    const bool fresh = cp.agc_reset == 2;  // zero sums (k_pc_index zeroed the history rows)
    const int T = a.len[slot];
    if (T == 0) return;
    const int nfull = T / KB;
    float *__restrict__ X = a.X + (size_t)slot * a.px;
    float *__restrict__ M1 = a.M1 + (size_t)slot * a.px;
    float *__restrict__ V1 = a.V1 + (size_t)slot * a.pv + a.vo + (a.L - 1);  // 16-byte aligned (vo)
    float s1 = fresh ? 0.f : a.dc_s1[slot], s2 = fresh ? 0.f : a.dc_s2[slot];
    const float rD = 1.0f / 32.0f;
    // x lives in a ring of SIX register sets of one 16-step block each: at block b the sets hold the blocks
    // b-2 and b-1 (the values the first sum evicts; the first of b-1 is x_{t-D+1} of the block's last step,
    // getLatest(delay - 1), src/utils.h:160-166), b (inserted now) and b+1..b+3 (loads in flight).  Nothing is
    // ever copied from set to set - the ring is indexed at compile time, six blocks per trip.  ms: the first
    // running SUM of the steps whose average the second sum evicts, two alternating sets (s1 = 32 * m1 exactly,
    // so the eviction and the insertion are one fma each).
    pc_f4 xr[RING][4];
    float ms[2][KB];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        xr[RING - 2][q] = reinterpret_cast<const pc_f4 *>(X)[q];          // block -2
        xr[RING - 1][q] = reinterpret_cast<const pc_f4 *>(X + KB)[q];     // block -1
    }
#pragma unroll
    for (int i = 0; i < KB; i++) {
        ms[0][i] = __fmul_rn(M1[i], 32.0f);
        ms[1][i] = __fmul_rn(M1[KB + i], 32.0f);
    }
    auto fetch = [&](auto kc, int blk) {
        constexpr int k = decltype(kc)::value;
        // UNCONDITIONAL (round 3): a load under `if (blk < nfull)` is a load under an exec mask, and behind those the
        // compiler waits with s_waitcnt vmcnt(0) at the top of every block.  Blocks up to nfull + AHEAD are read:
        // inside the row's PC_PAD floats of padding (psdr_set_post_chain: px, pv), values never used.
        const pc_f4 *src = reinterpret_cast<const pc_f4 *>(X + D + blk * KB);  // D, KB, px: multiples of 4
#pragma unroll
        for (int q = 0; q < 4; q++) xr[k][q] = src[q];
    };
    // m1 = s1 / 32 is exact (a power of two), so rounding (s2 - m1_old) + m1 and x - s2 / 32 after the exact
    // products is the reference's arithmetic with three fused operations instead of five:
    //   t = s2 - s1_old / 32      s2 = t + s1 / 32      out = x_{t-D+1} - s2 / 32
    // (the loop is bound by its own instruction stream: 7 -> 5 operations per sample, and no moves)
    const float nrD = -rD;
    auto block = [&](auto jc, int t0) {
        constexpr int J = decltype(jc)::value, E = (J + RING - 2) % RING, N1 = (J + RING - 1) % RING, P = J & 1;
        pc_f4 o[4];
#pragma unroll
        for (int i = 0; i < KB; i++) {
            s1 = __fadd_rn(__fsub_rn(s1, xr[E][i >> 2][i & 3]), xr[J][i >> 2][i & 3]);
            const float t2 = __fmaf_rn(ms[P][i], nrD, s2);
            s2 = __fmaf_rn(s1, rD, t2);
            ms[P][i] = s1;
            const float xd = i + 1 < KB ? xr[E][(i + 1) >> 2][(i + 1) & 3] : xr[N1][0][0];
            o[i >> 2][i & 3] = __fmaf_rn(s2, nrD, xd);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) reinterpret_cast<pc_f4 *>(V1 + t0)[q] = o[q];
    };
    pc_static_for<0, AHEAD>([&](auto kc) { fetch(kc, decltype(kc)::value); });
    int b = 0;
    for (; b + RING <= nfull; b += RING)
        pc_static_for<0, RING>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b + J + AHEAD);
            block(jc, (b + J) * KB);
        });
    // up to RING - 1 more whole blocks (b is a multiple of RING: block b + j lives in set j)
    {
        const int b0 = b;
        pc_static_for<0, RING - 1>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if (b0 + J < nfull) {
                fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b0 + J + AHEAD);
                block(jc, (b0 + J) * KB);
                b = b0 + J + 1;
            }
        });
    }
    // now (b even) ms[0] is the older set of m1 values, else ms[1]: the window of m1 values in time
    // order goes to rows t1.. of M1 for the remaining T - nfull*KB (< KB) steps, one by one
    const int t1 = nfull * KB;
#pragma unroll
    for (int i = 0; i < KB; i++) {
        M1[t1 + i] = __fmul_rn((b & 1) ? ms[1][i] : ms[0][i], rD);
        M1[t1 + KB + i] = __fmul_rn((b & 1) ? ms[0][i] : ms[1][i], rD);
    }
    for (int t = t1; t < T; t++) {
        s1 = __fadd_rn(__fadd_rn(s1, -X[t]), X[D + t]);
        const float m1 = __fmul_rn(s1, rD);
        s2 = __fadd_rn(__fadd_rn(s2, -M1[t]), m1);
        M1[D + t] = m1;
        V1[t] = __fsub_rn(X[t + 1], __fmul_rn(s2, rD));
    }
    // the last 32 m1 values in time order become rows 0..31 (X's own history is moved by k_pc_history)
    float keep[D];
#pragma unroll
    for (int i = 0; i < D; i++) keep[i] = M1[T + i];
#pragma unroll
    for (int i = 0; i < D; i++) M1[i] = keep[i];
    a.dc_s1[slot] = s1;
    a.dc_s2[slot] = s2;
}

// grid (client, block k of L rows of V1, 2): z = 0: prefix maxima of |v| inside the block, 1: suffix
// maxima.  max is associative and exact, so this one IS parallel: the block's rows go through LDS
// (coalesced both ways), each lane scans a contiguous piece, the piece totals are combined by a wave
// scan.  The block is walked in chunks of PC_SCAN_CHUNK rows with the running maximum carried from chunk
// to chunk, so the LDS need does not grow with the look-ahead L (200 ms: 2400 rows at 12 kHz, 38400 at
// the 192 kHz of the reference's shipped config.toml - 150 KB if the block had to fit at once).
constexpr int PC_SCAN_CHUNK = 4096;
__global__ __launch_bounds__(64) void k_pc_scan(PostArgs a) {
    __shared__ float pc_blk[PC_SCAN_CHUNK];
    const int lane = threadIdx.x;
    const int slot = a.clients[blockIdx.x].slot;
    const int rows = a.L - 1 + a.len[slot];
    const int r0 = blockIdx.y * a.L, r1 = min(r0 + a.L, rows);
    if (r0 >= rows) return;
    const int n = r1 - r0;
    const float *__restrict__ v1 = a.V1 + (size_t)slot * a.pv + a.vo + r0;
    const bool suffix = blockIdx.z != 0;
    float *__restrict__ out = (suffix ? a.S : a.P) + (size_t)slot * a.pv + r0;
    float carry = 0.f;  // maximum of everything before this chunk (scan order)
    // scan position q = distance from the scan's start (suffix scans run backwards): row q, or n - 1 - q
    for (int q0 = 0; q0 < n; q0 += PC_SCAN_CHUNK) {
        const int m = min(PC_SCAN_CHUNK, n - q0);
        for (int i = lane; i < m; i += 64) pc_blk[i] = fabsf(v1[suffix ? n - 1 - (q0 + i) : q0 + i]);
        __syncthreads();
        const int C = (m + 63) / 64, i0 = min(lane * C, m), i1 = min(i0 + C, m);
        float mx = 0.f;
        for (int i = i0; i < i1; i++) mx = fmaxf(mx, pc_blk[i]);
        float incl = mx;  // inclusive wave scan of the piece maxima
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(incl, d, 64);
            if (lane >= d) incl = fmaxf(incl, o);
        }
        float run = __shfl_up(incl, 1, 64);  // maximum of all earlier pieces of this chunk ...
        if (lane == 0) run = 0.f;
        run = fmaxf(run, carry);             // ... and of all earlier chunks
        for (int i = i0; i < i1; i++) {
            run = fmaxf(run, pc_blk[i]);
            pc_blk[i] = run;
        }
        carry = fmaxf(carry, __shfl(incl, 63, 64));
        __syncthreads();
        for (int i = lane; i < m; i += 64) out[suffix ? n - 1 - (q0 + i) : q0 + i] = pc_blk[i];
        __syncthreads();
    }
}

// w_t = desired / (peak_t + 1e-10), peak_t = max |V1| over rows [t, t+L-1] = max(S[t], P[t+L-1]);
// in place into S[t] (only this thread reads S[t]).  grid (client, blocks of 256 samples)
__global__ __launch_bounds__(256) void k_pc_want(PostArgs a) {
    const int slot = a.clients[blockIdx.x].slot;
    const int t = blockIdx.y * 256 + threadIdx.x;
    if (t >= a.len[slot]) return;
    float *S = a.S + (size_t)slot * a.pv;
    const float *P = a.P + (size_t)slot * a.pv;
    const float peak = fmaxf(S[t], P[t + a.L - 1]);
    S[t] = __fdiv_rn(a.desired, __fadd_rn(peak, 1e-10f));
}

// the gain recurrence (src/utils/audioprocessing.cpp:55-66); g_t -> P[t] (0 while the
// look-ahead buffer is still filling: the reference outputs 0 there and leaves the gain alone;
// an active gain is never 0: w_t > 0)
template <bool ATT_FASTER>
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
    constexpr int KB = 16, GR = PSDR_PC_RING, AHEAD = GR - 1;  // blocks of loads in flight: see k_pc_ma2
    __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
    const float *__restrict__ W = a.S + (size_t)slot * a.pv;
    float *__restrict__ G = a.P + (size_t)slot * a.pv;
    const float att = a.attack, rel = a.release;
    // gain <- gain + (w < gain ? attack : release) * (w - gain): three operations per sample
    // (fma(-a, g - w, g) and fma(a, w - g, g) are the same value: negating both factors is exact)
    // The coefficient is picked by the sign BIT of d (arithmetic shift + bit-field insert): a compare would
    // go through VCC, and VCC -> v_cndmask costs two wait states on this part in a loop that is nothing but
    // its own instruction stream.  (d = -0 picks the attack coefficient where the reference picks release:
    // the product is a zero either way and the sum is the same.)
    // Round 3: the loop is ONE wave's dependent chain (~8 cycles from an instruction to the next that needs its
    // result: 36 cycles per sample measured), so what counts is the DEPTH per sample.  Both candidates are computed
    // side by side and the choice is a min / max: with attack > release (the reference's 50 ms against 300 ms)
    // d < 0 makes attack * d the smaller product and d > 0 the larger, and fma rounds monotonically - min(A, R) IS
    // the reference's pick, bit for bit (d = 0: both are the gain).  sub -> fma, fma -> min: depth 3 instead of 4.
    // (ATT_FASTER = attack >= release, a template parameter: as a run-time flag the compiler computed min AND max and
    // selected - six instructions and depth 4 again)
    auto step = [&](float w) -> float {
        const float d = __fsub_rn(w, gain);
        const float ga = __fmaf_rn(att, d, gain), gr = __fmaf_rn(rel, d, gain);
        gain = ATT_FASTER ? fminf(ga, gr) : fmaxf(ga, gr);
        return gain;
    };
    // while the look-ahead buffer is filling (only right after a reset / for a new client) the
    // reference outputs 0 and leaves the gain alone: those steps, then (once 16-byte aligned) whole
    // blocks with the loads three blocks ahead, then the rest
    int t = 0;
    for (; t < T && n0 + t + 1 < L; t++) G[t] = 0.f;
    for (; t < T && (t & 3); t++) G[t] = step(W[t]);
    const int nblk = (T - t) / KB;
    pc_f4 w[GR][4];
    auto fetch = [&](auto kc, int blk) {
        constexpr int k = decltype(kc)::value;
        // (unconditional, like k_pc_ma2's: blocks up to nblk + AHEAD stay inside the row's padding)
        const pc_f4 *src = reinterpret_cast<const pc_f4 *>(W + t + blk * KB);
#pragma unroll
        for (int q = 0; q < 4; q++) w[k][q] = src[q];
    };
    auto block = [&](const pc_f4 (&wv)[4], int blk) {
        pc_f4 g[4];
#pragma unroll
        for (int i = 0; i < KB; i++) g[i >> 2][i & 3] = step(wv[i >> 2][i & 3]);
#pragma unroll
        for (int q = 0; q < 4; q++) reinterpret_cast<pc_f4 *>(G + t + blk * KB)[q] = g[q];
    };
    pc_static_for<0, AHEAD>([&](auto kc) { fetch(kc, decltype(kc)::value); });
    int b = 0;
    for (; b + GR <= nblk; b += GR)
        pc_static_for<0, GR>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            fetch(std::integral_constant<int, (J + AHEAD) % GR>{}, b + J + AHEAD);
            block(w[J], b + J);
        });
    {
        const int b0 = b;
        pc_static_for<0, GR - 1>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if (b0 + J < nblk) {
                fetch(std::integral_constant<int, (J + AHEAD) % GR>{}, b0 + J + AHEAD);
                block(w[J], b0 + J);
            }
        });
    }
    for (t += nblk * KB; t < T; t++) G[t] = step(W[t]);
    a.agc_gain[slot] = gain;
    a.agc_n0[slot] = min(n0 + T, L);
}

// current_sample * gain (row t of V1 is the oldest sample of the look-ahead window; gain 0 =
// buffer still filling -> 0) and dsp_float_to_int16 (src/utils/dsp.cpp:152-165), written to the
// frame's place in pcm[slot][frame][j]; rows of dropped frames are zero.  grid (client, frame)
__global__ __launch_bounds__(256) void k_pc_out(PostArgs a) {
    const int slot = a.clients[blockIdx.x].slot, f = blockIdx.y;
    const int pos = a.fstart[(size_t)slot * a.max_batch + f];
    int32_t *dst = a.pcm + ((size_t)slot * a.max_batch + f) * a.h;
    const float *G = a.P + (size_t)slot * a.pv + pos, *V = a.V1 + (size_t)slot * a.pv + a.vo + pos;
    for (int j = threadIdx.x; j < a.h; j += blockDim.x) {
        int v = 0;
        if (pos >= 0) {
            const float g = G[j];
            const float y = g == 0.f ? 0.f : __fmul_rn(V[j], g);
            v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
            v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
        }
        dst[j] = v;
    }
}

// the last D rows of X / M1 and the last L-1 rows of V1 become the history rows of the next
// batch.  grid (client); hist_sel = 0: X and M1, 1: V1 -> V1n.  One work-group per stream: all reads, a
// barrier, all writes (source and destination overlap when the batch is shorter than the history).
// dynamic LDS: max(D, L - 1) floats
__global__ __launch_bounds__(256) void k_pc_history(PostArgs a) {
    extern __shared__ float pc_hist[];
    const int slot = a.clients[blockIdx.x].slot;
    const int T = a.len[slot];
    if (T == 0 && a.hist_sel == 0) return;
    if (a.hist_sel == 0) {
        float *x = a.X + (size_t)slot * a.px, *m = a.M1 + (size_t)slot * a.px;
        for (int r = threadIdx.x; r < a.D; r += blockDim.x) pc_hist[r] = x[r + T];
        __syncthreads();
        for (int r = threadIdx.x; r < a.D; r += blockDim.x) x[r] = pc_hist[r];
        if (!a.ma_fused) {
            __syncthreads();
            for (int r = threadIdx.x; r < a.D; r += blockDim.x) pc_hist[r] = m[r + T];
            __syncthreads();
            for (int r = threadIdx.x; r < a.D; r += blockDim.x) m[r] = pc_hist[r];
        }
        return;
    }
    // V1 and V1n are the two buffers of the double-buffered stream: source and destination never overlap, a plain
    // copy (no LDS: at 192 kHz the L - 1 rows would be 150 KB of it)
    const float *v = a.V1 + (size_t)slot * a.pv + a.vo;
    float *vn = a.V1n + (size_t)slot * a.pv + a.vo;
    for (int r = threadIdx.x; r < a.L - 1; r += blockDim.x) vn[r] = v[r + T];
}

}  // namespace psdr
