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
// Nothing else in the chain is sequential, but the recurrences decide the shape of everything: they cannot be split
// along time, so their parallelism is the CLIENTS - one lane per client, one wave per 64 clients, ~25 cycles per sample
// (the latency of three dependent f32 operations), 2 ms per 512-frame batch whatever the client count.
//
// Layout (round 5): every stream is LANE-INTERLEAVED per group of 64 slots - sample t of slot s at float
//   ((s >> 6) * pitch + (t >> 2) * 4) * 64 + (s & 63) * 4 + (t & 3)
// i.e. [group][t / 4][lane][4].  Lane l of a wave owns slot 64 g + l: its four next samples are ONE 16-byte access,
// and the wave's 64 of them one contiguous KiB.  (Rounds 2-4 kept the streams client-major, [slot][t]: a wave's load
// touched 64 different lines, and with all 64 lanes in use - 256 clients - the two recurrence kernels were bound by the
// texture addresser, 4.5 ms instead of 2 ms; the parallel kernels ran lanes-along-time through LDS and starved beside
// the FFT passes, which own all but 32 KiB of every CU's LDS.)  The history a kernel needs sits IN FRONT of the new
// samples (D rows for the averages, L-1 rows for the AGC look-ahead); the sets rotate per batch (three of them: a
// batch's chain may still be running when the next two start), the tails are copied into the next set's history rows.
// EVERY kernel is lane = client now; the ones that are not recurrences split time into independent pieces:
//   k_pc_index    stream offset of every frame of every client (NaN-flagged frames dropped)
//   k_pc_gather   audio[slot][frame][j] -> X[slot][D + t]
//   k_pc_ma2      both running sums in one loop (D = 32)          (sequential)
//   k_pc_mad      the same for any power-of-two D (48 kHz: 128, 192 kHz: 512): the sums of the last D steps in an LDS ring
//   k_pc_ma<0|1>  the two running sums, any D                      (sequential, one wave, fallback)
//   k_pc_history  the last D / L-1 rows become the next set's history
//   k_pc_submax / k_pc_prefix / k_pc_want
//                 AGC look-ahead peak: the sliding maximum of |x| over L samples (the reference's monotonic deque) as
//                 van Herk prefix / suffix maxima of blocks of L rows, a wave per (group, sub-block of a block), then
//                 w_t = desired / (peak_t + 1e-10)
//   k_pc_gain     the gain recurrence                              (sequential)
//   k_pc_out      delayed sample * gain, int16 conversion, straight into pcm[slot][frame][j]
// Round 6 (PSDR_OPT_POST_CHAIN_AGC = 1, the default; what the chain costs the step is its memory traffic - see the end of
// this file): k_pc_submax / _prefix / _want / _gain / _out become
//   k_pc_cm / k_pc_cscan   maxima of |V1| per chunk of 16 floats and their block scans (history chunks only when the moving
//                 averages leave the maxima of the new samples on their way: k_pc_ma2 CMW)
//   k_pc_agc      look-ahead peak, w_t, the gain recurrence and the int16 output in ONE four-wave kernel
//   k_pc_zero     zero rows for dropped frames
// and k_pc_ma2 reads the demodulator's rows itself where a work-group's streams are whole (DIRECT: no k_pc_gather4 for it).
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
// A recurrence wave that owns its SIMD: naming the last vector and accumulator register makes the kernel's allocation the
// whole register file of a SIMD lane (512), so no other wave shares the SIMD - and the wave only fits where the passes
// left a CU free (postchain.hip).
// (256 clients: 3.46 -> 3.17 ms per step with two CUs per XCD free; level with fewer clients - profiles/r05_post_chain_reserve_own.jsonl)
#define PC_OWN_SIMD() asm volatile("; the wave owns its SIMD" ::: "v255", "a255")

// the lane-interleaved stream layout (see above): a slot's lane base, and the offset of its row t from there
__device__ __forceinline__ size_t pc_base(int slot, size_t pitch) { return (size_t)(slot >> 6) * pitch * 64 + (size_t)(slot & 63) * 4; }
__device__ __forceinline__ size_t pc_el(int t) { return ((size_t)(t >> 2) << 8) + (size_t)(t & 3); }
// the 16-byte row group q (rows 4q .. 4q+3) of a lane
__device__ __forceinline__ pc_f4 *pc_row4(float *lane_base, int q) { return reinterpret_cast<pc_f4 *>(lane_base) + (size_t)q * 64; }
__device__ __forceinline__ const pc_f4 *pc_row4(const float *lane_base, int q) { return reinterpret_cast<const pc_f4 *>(lane_base) + (size_t)q * 64; }

__device__ __forceinline__ int pc_mul24(int x, int y) {
    int p;
    asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
    return p;
}

// may a slot's new samples be read from the demodulator's rows as they lie (k_pc_ma2 DIRECT)?  Its stream is the whole batch
// (nothing dropped, not paused) and at least as long as the DC delay (k_pc_history takes the next history from the same rows);
// a slot without a stream does not care.  k_pc_ma2, k_pc_gather4 and k_pc_history decide from this one predicate.
__device__ __forceinline__ bool pc_direct_ok(bool listed, int T, int Tfull, int D) { return !listed || T == 0 || (T == Tfull && T >= D); }

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
        // (the inverse, for the kernel that walks the STREAM - k_pc_agc: the k-th surviving frame, lane-interleaved like the streams)
        if (alive && a.falive) a.falive[((size_t)(slot >> 6) * a.max_batch + (cnt + before)) * 64 + (slot & 63)] = f;
        cnt += __popcll(m);
    }
    if (lane == 0) a.len[slot] = cnt * a.h;
    // (a new client in this slot - agc_reset 2 - starts from zero history: the moving-average kernels zero their own
    // history rows, k_pc_ma*)
}

// grid (client, frame): one frame of audio to its place in the stream (contiguous both sides)
__global__ __launch_bounds__(256) void k_pc_gather(PostArgs a) {
    const int slot = a.clients[blockIdx.x].slot, f = blockIdx.y;
    const int pos = a.fstart[(size_t)slot * a.max_batch + f];
    if (pos < 0) return;
    const float *src = a.audio + ((size_t)slot * a.max_batch + f) * a.h;
    float *dst = a.X + pc_base(slot, a.px);
    for (int j = threadIdx.x; j < a.h; j += blockDim.x) dst[pc_el(a.D + pos + j)] = src[j];
}

// the same with lane = slot (h and D multiples of 4: every frame starts on a row group): grid (groups of 64 slots, frame),
// thread (lane, j / 4) - 16 bytes from the client's audio row (64 rows per wave, each line used up over eight turns),
// one contiguous KiB into X per wave
typedef int pc_i4 __attribute__((ext_vector_type(4)));
typedef int pc_i2 __attribute__((ext_vector_type(2)));
// four clamped samples as int16 (PSDR_OPT_POST_CHAIN_PCM16: half the bytes on their way to the host)
__device__ __forceinline__ pc_i2 pc_pack16(pc_i4 o) {
    return pc_i2{(int)(((unsigned)o[0] & 0xffffu) | ((unsigned)o[1] << 16)), (int)(((unsigned)o[2] & 0xffffu) | ((unsigned)o[3] << 16))};
}
__global__ __launch_bounds__(256) void k_pc_gather4(PostArgs a) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63), f = blockIdx.y;
    const bool listed = slot < a.slots && a.slot_ci[slot] >= 0;
    if (a.direct) {
        // the slots of a k_pc_ma2 work-group (a.lanes consecutive ones) whose streams are all the demodulator's rows as they
        // lie: the moving averages read a.audio themselves (k_pc_ma2 DIRECT) - nothing to gather
        const unsigned long long ok = __ballot(pc_direct_ok(listed, listed ? a.len[slot] : 0, a.nframes * a.h, a.D));
        const int first = (int)(threadIdx.x & 63) & ~(a.lanes - 1);
        const unsigned long long grp = a.lanes == 64 ? ~0ull : (((1ull << a.lanes) - 1ull) << first);
        if ((ok & grp) == grp) return;
    }
    if (!listed) return;
    const int pos = a.fstart[(size_t)slot * a.max_batch + f];
    if (pos < 0) return;
    const pc_f4 *src = reinterpret_cast<const pc_f4 *>(a.audio + ((size_t)slot * a.max_batch + f) * a.h);
    float *X = a.X + pc_base(slot, a.px);
    const int q0 = (a.D + pos) >> 2;
    for (int j4 = threadIdx.x >> 6; j4 < (a.h >> 2); j4 += 4) *pc_row4(X, q0 + j4) = src[j4];
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
    const int slot = blockIdx.x * 64 + threadIdx.x;  // lane = slot & 63: the wave's accesses are contiguous
    if (slot >= a.slots) return;
    const int ci = a.slot_ci[slot];
    if (ci < 0) return;
    const ClientParams cp = a.clients[ci];
    const int D = a.D;
    const bool fresh = cp.agc_reset == 2;  // a new client in this slot: zero sums and zero history
    const int T = a.len[slot];
    float *inw = (SECOND ? a.M1 : a.X) + pc_base(slot, a.px);
    if (fresh)
        for (int r = 0; r < D; r++) inw[pc_el(r)] = 0.f;
    const float *in = inw;
    const float *X = a.X + pc_base(slot, a.px);
    float *out = SECOND ? a.V1 + pc_base(slot, a.pv) : a.M1 + pc_base(slot, a.px);
    const int o0 = SECOND ? a.vo + a.L - 1 : D;
    float s = fresh ? 0.f : (SECOND ? a.dc_s2 : a.dc_s1)[slot];
    const float fD = (float)D, rD = 1.0f / fD;
    for (int t = 0; t < T; t++) {
        s = __fadd_rn(s, -in[pc_el(t)]);
        s = __fadd_rn(s, in[pc_el(D + t)]);
        const float m = POW2 ? __fmul_rn(s, rD) : __fdiv_rn(s, fD);
        out[pc_el(o0 + t)] = SECOND ? __fsub_rn(X[pc_el(t + 1)], m) : m;
    }
    (SECOND ? a.dc_s2 : a.dc_s1)[slot] = s;
}

// Both moving averages for D = 32, as TWO waves of one work-group (round 5).  One wave doing both sums issues six
// instructions per sample - 25 cycles, 1.9 - 2.8 ms per 512 frames at the capped clock beside the FFT passes: longer than
// the step it is meant to hide behind, and no pipeline helps a stage that is sequential across batches.  Split:
//   wave 0  loads x (a ring of register sets, the loads nine blocks ahead), runs s1 and leaves each 16-step block of x
//           and s1 in LDS;  no global stores - its s_waitcnt vmcnt never waits for a store's acknowledgement
//   wave 1  takes the block from LDS one barrier later, runs s2 and the output, stores V1;  no global loads in its loop -
//           its stores are fire and forget
// two instructions on the critical path of either (sub, add / fma, fma): ~17 cycles per sample with the barrier.
// The x values the first average evicts were inserted two blocks earlier and the m1 values the second average evicts were
// produced two blocks earlier: both stay in alternating register sets.  M1 only holds the 32 carried values between
// batches (time order, oldest first).  A stream that is not a whole number of blocks ends with a short scalar loop on
// the in-memory history (wave 1).  Lanes of a group may have streams of different lengths (dropped frames, paused
// clients): the trip count is the group's maximum, a lane past its own end keeps its state.
constexpr int PC_MA_RING = 12;  // wave 0's register sets of x: blocks b-2 .. b (in use), b+1 .. b+9 in flight
// CMW (round 6, with the one-kernel AGC): a THIRD wave takes the blocks of output from wave 1 through LDS, stores them and
// leaves the maximum of |V1| over each 16-sample block as CM[L/16 + b] - the chunk maxima k_pc_agc's look-ahead peak is made
// of, without another pass over V1 (k_pc_cm then covers the L/16 history chunks only).  Wave 1 swaps four global stores per
// block for four LDS writes: the same number of instructions in its loop.
template <bool OWN, bool CMW = false>
__global__ __launch_bounds__(CMW ? 192 : 128) void k_pc_ma2(PostArgs a) {
    __shared__ pc_f4 hand[2][8][64];  // [buffer][0-3: x of the block, 4-7: s1 of the block][lane]: 16 KiB
    __shared__ float fin[64];         // wave 0's s1 after its last block
    __shared__ pc_f4 ohand[CMW ? 2 : 1][4][CMW ? 64 : 1];  // CMW: [buffer][row group][lane] a block of output on its way to wave 2
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // a work-group owns a.lanes (16, 32 or 64) consecutive slots of a group of 64: lane = slot & (a.lanes - 1), the other
    // lanes leave below.  (A recurrence costs the same for 1 lane or 64, but its memory operations do not: 1 KiB loads and
    // stores per wave are acknowledged later than 256-byte ones beside the passes - 2.8 against 2.0 ms per 512 frames.)
    const int wpg = 64 / a.lanes;
    const int slot = ((int)blockIdx.x / wpg) * 64 + ((int)blockIdx.x % wpg) * a.lanes + lane;  // (the streams of a group are allocated whole)
    const int ci = (lane < a.lanes && slot < a.slots) ? a.slot_ci[slot] : -1;
    const bool listed = ci >= 0;
    const bool fresh = listed && a.clients[ci].agc_reset == 2;  // a new client in this slot: zero sums, zero history rows
    constexpr int KB = 16, D = 32;
    __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
    if constexpr (OWN) PC_OWN_SIMD();  // a few waves next to the FFT passes' issue-bound ones: let them issue first
    const int T = listed ? a.len[slot] : 0;
    const int nfull = T / KB;
    int nmax = nfull;
#pragma unroll
    for (int d = 32; d; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
    // DIRECT (round 6, a.direct): when every stream of the work-group is the demodulator's rows as they lie - no frame dropped,
    // nobody paused in mid-batch - the new samples are read from a.audio itself and k_pc_gather4 leaves the work-group's X
    // rows alone (it takes the same decision from the same numbers): a pass of 4 bytes per sample written and one read less,
    // of the fifteen the chain cost the step by (profiles/r06_post_chain_ablation.json).  A lane's 16-byte pieces then lie
    // max_batch * h floats apart - a load touches a.lanes lines instead of a.lanes / 8 - but each line serves eight of them
    // in a row, and the loader wave has the time.
    const bool direct = a.direct && __all(pc_direct_ok(listed, T, a.nframes * a.h, D));
    if (lane >= a.lanes) return;  // (the wave goes on, barriers included, with the lanes that own a slot)
    float *__restrict__ X = a.X + pc_base(slot, a.px);
    float *__restrict__ M1 = a.M1 + pc_base(slot, a.px);
    float *__restrict__ M1n = a.M1n + pc_base(slot, a.px);
    float *__restrict__ V1 = a.V1 + pc_base(slot, a.pv);
    const float *__restrict__ aud = a.audio + (size_t)min(slot, a.slots - 1) * a.max_batch * a.h;
    auto xrow = [&](int r) -> float { return (direct && r >= D) ? aud[r - D] : X[pc_el(r)]; };  // row r of the stream: D history rows, then the samples
    const float rD = 1.0f / 32.0f, nrD = -rD;
    if (wid == 0) {
        // ---- wave 0: s1_t = (s1 - x_{t-32}) + x_t
        constexpr int RING = PC_MA_RING, AHEAD = RING - 3;
        if (fresh)
            for (int r = 0; r < D; r++) X[pc_el(r)] = 0.f;
        float s1 = (fresh || !listed) ? 0.f : a.dc_s1[slot];
        pc_f4 xr[RING][4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xr[RING - 2][q] = *pc_row4(X, q);      // block -2
            xr[RING - 1][q] = *pc_row4(X, 4 + q);  // block -1
        }
        // block blk of the new samples: base + blk * bstep floats, its four row groups qstep floats apart (X: lane-interleaved;
        // direct: the slot's audio rows, blocks past the slot's own rows clamped - X has its padding for that)
        const float *fbase = direct ? aud : X + (size_t)(D >> 2) * 256;
        const int bstep = direct ? KB : 4 * 256, qstep = direct ? 4 : 256;
        const int blast = direct ? (a.max_batch * a.h) / KB - 1 : 0x7fffffff;
        auto fetch = [&](auto kc, int blk) {
            constexpr int k = decltype(kc)::value;
            // UNCONDITIONAL: a load under an exec mask makes the compiler wait with vmcnt(0) at the top of every block.
            // Blocks up to nmax + AHEAD are read: inside the pitch's PC_PAD floats of padding, values never used.
            const float *src = fbase + (size_t)min(blk, blast) * bstep;
#pragma unroll
            for (int q = 0; q < 4; q++) xr[k][q] = *reinterpret_cast<const pc_f4 *>(src + q * qstep);
        };
        pc_f4 sv[4] = {};
        auto block = [&](auto jc, int b) {
            constexpr int J = decltype(jc)::value, E = (J + RING - 2) % RING;
            if (b < nfull) {
#pragma unroll
                for (int i = 0; i < KB; i++) {
                    s1 = __fadd_rn(__fsub_rn(s1, xr[E][i >> 2][i & 3]), xr[J][i >> 2][i & 3]);
                    sv[i >> 2][i & 3] = s1;
                }
            }
            // the barrier comes BEFORE this block's LDS writes: it publishes the previous block, whose writes have had a
            // whole block's time to land (waiting for one's own writes in front of every barrier cost a third of the loop)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                hand[b & 1][q][lane] = xr[J][q];
                hand[b & 1][4 + q][lane] = sv[q];
            }
            if (b + 1 == nmax) fin[lane] = s1;
        };
        pc_static_for<0, AHEAD>([&](auto kc) { fetch(kc, decltype(kc)::value); });
        int b = 0;
        for (; b + RING <= nmax; b += RING)
            pc_static_for<0, RING>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b + J + AHEAD);
                block(jc, b + J);
            });
        {
            const int b0 = b;
            pc_static_for<0, RING - 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                if (b0 + J < nmax) {
                    fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b0 + J + AHEAD);
                    block(jc, b0 + J);
                }
            });
        }
        if (nmax == 0) fin[lane] = s1;
        __syncthreads();  // (wave 1 runs three blocks behind)
        __syncthreads();
        __syncthreads();
        return;
    }
    if constexpr (CMW) {
        if (wid == 2) {
            // ---- wave 2: block b of output from LDS (complete behind the barrier that ends wave 1's block b; wave 1 writes that
            // buffer again in block b + 2, two barriers on) -> V1, and its maximum -> CM[L/16 + b]
            const int vq = (a.vo + a.L - 1) >> 2;
            float *__restrict__ cm = a.CM + ((size_t)(slot >> 6) * a.nch + (size_t)(a.L >> 4)) * 64 + (size_t)(slot & 63);
            __syncthreads();
            __syncthreads();
            __syncthreads();
            for (int b = 0; b < nmax; b++) {
                __syncthreads();
                pc_f4 o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) o[q] = ohand[b & 1][q][lane];
                if (b < nfull) {
                    float m = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        *pc_row4(V1, vq + b * (KB / 4) + q) = o[q];
#pragma unroll
                        for (int i = 0; i < 4; i++) m = fmaxf(m, fabsf(o[q][i]));
                    }
                    cm[(size_t)b * 64] = m;
                }
            }
            return;
        }
    }
    // ---- wave 1: s2_t = (s2 - m1_{t-32}) + m1_t, out_t = x_{t-31} - s2_t / 32, one block behind wave 0
    // m1 = s1 / 32 is exact (a power of two), so rounding (s2 - m1_old) + m1 and x - s2 / 32 after the exact products is the
    // reference's arithmetic with three fused operations instead of five:
    //   t = s2 - s1_old / 32      s2 = t + s1 / 32      out = x_{t-D+1} - s2 / 32
    if (fresh)
        for (int r = 0; r < D; r++) M1[pc_el(r)] = 0.f;
    float s2 = (fresh || !listed) ? 0.f : a.dc_s2[slot];
    const int vq = (a.vo + a.L - 1) >> 2;  // sample 0's row group (vo makes row L-1 a multiple of 4)
    // x in four register sets (blocks b-2, b-1, b and the block being read: x_{t-31} of the block's last step is the first
    // of b-1, getLatest(delay - 1), src/utils.h:160-166), the first running SUM of the steps whose average the second sum
    // evicts in two (s1 = 32 m1 exactly), the incoming s1 in two: all indexed at compile time, four blocks per trip.  The
    // LDS reads of block b+1 are issued BEFORE block b is computed (and waited for at the barrier): three blocks behind wave 0.
    pc_f4 xb[4][4], svs[2][4];
    float ms[2][KB];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        xb[2][q] = *pc_row4(X, q);      // block -2 (b mod 4)
        xb[3][q] = *pc_row4(X, 4 + q);  // block -1
        if (fresh) xb[2][q] = xb[3][q] = pc_f4{0.f, 0.f, 0.f, 0.f};  // (wave 0 is zeroing those rows)
    }
#pragma unroll
    for (int i = 0; i < KB; i++) {
        ms[0][i] = __fmul_rn(M1[pc_el(i)], 32.0f);
        ms[1][i] = __fmul_rn(M1[pc_el(KB + i)], 32.0f);
    }
    auto take = [&](auto jc, int r) {  // block r (r mod 4 == J) from LDS into its register sets
        constexpr int J = decltype(jc)::value;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xb[J][q] = hand[r & 1][q][lane];
            svs[J & 1][q] = hand[r & 1][4 + q][lane];
        }
    };
    auto block = [&](auto jc, int b) {
        constexpr int J = decltype(jc)::value, E = (J + 2) % 4, N1 = (J + 3) % 4, P = J & 1;
        if (b + 1 < nmax) take(std::integral_constant<int, (J + 1) % 4>{}, b + 1);
        if (b < nfull) {
            pc_f4 o[4];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                const float s1 = svs[P][i >> 2][i & 3];
                const float t2 = __fmaf_rn(ms[P][i], nrD, s2);
                s2 = __fmaf_rn(s1, rD, t2);
                ms[P][i] = s1;
                const float xd = i + 1 < KB ? xb[E][(i + 1) >> 2][(i + 1) & 3] : xb[N1][0][0];
                o[i >> 2][i & 3] = __fmaf_rn(s2, nrD, xd);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (CMW)
                    ohand[b & 1][q][lane] = o[q];
                else
                    *pc_row4(V1, vq + b * (KB / 4) + q) = o[q];
            }
        }
        __syncthreads();
    };
    __syncthreads();
    __syncthreads();  // block 0 is in LDS
    if (nmax > 0) take(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    int b = 0;
    for (; b + 4 <= nmax; b += 4) pc_static_for<0, 4>([&](auto jc) { block(jc, b + decltype(jc)::value); });
    {
        const int b0 = b;
        pc_static_for<0, 3>([&](auto jc) {
            if (b0 + decltype(jc)::value < nmax) block(jc, b0 + decltype(jc)::value);
        });
    }
    if (!listed) return;
    float s1 = fin[lane];  // (a lane whose stream is shorter than the group's longest kept its state from its own last block on)
    // now (nfull even) ms[0] is the older set of m1 values, else ms[1]: the window of m1 values in time order goes to rows
    // t1.. of M1 for the remaining T - nfull * KB (< KB) steps, one by one
    const int t1 = nfull * KB;
#pragma unroll
    for (int i = 0; i < KB; i++) {
        M1[pc_el(t1 + i)] = __fmul_rn((nfull & 1) ? ms[1][i] : ms[0][i], rD);
        M1[pc_el(t1 + KB + i)] = __fmul_rn((nfull & 1) ? ms[0][i] : ms[1][i], rD);
    }
    const int v0 = a.vo + a.L - 1;
    for (int t = t1; t < T; t++) {
        s1 = __fadd_rn(__fadd_rn(s1, -xrow(t)), xrow(D + t));
        const float m1 = __fmul_rn(s1, rD);
        s2 = __fadd_rn(__fadd_rn(s2, -M1[pc_el(t)]), m1);
        M1[pc_el(D + t)] = m1;
        V1[pc_el(v0 + t)] = __fsub_rn(xrow(t + 1), __fmul_rn(s2, rD));
    }
    // the last 32 m1 values in time order become rows 0..31 of the NEXT set (X's own history is moved by k_pc_history)
#pragma unroll
    for (int i = 0; i < D; i++) M1n[pc_el(i)] = M1[pc_el(T + i)];
    a.dc_s1[slot] = s1;
    a.dc_s2[slot] = s2;
}

// The same two waves for any delay D that is a power of two and a multiple of 16 (48 kHz: 128, 192 kHz: 512 - the audio
// rates of the reference's shipped configs; the generic k_pc_ma pair takes 19 / 81 ms per 92 160 / 368 640 samples there,
// this one the 2 ms per 92 160 of k_pc_ma2).  What k_pc_ma2 keeps in registers for D = 32 does not fit for D = 512:
//   wave 0  loads the stream TWICE - the block it inserts (rows D + 16 b ..) and the block it evicts (rows 16 b .., read D
//           samples earlier as the new one: L2) - and hands the EVICTED block on (x_{t-D+1} is its neighbour)
//   wave 1  keeps the first running sums of the last D steps in an LDS ring it alone reads and writes (D x lanes x 4 bytes:
//           64 KiB for D = 512 and 32 lanes - the kernel lives on a CU the passes leave free, postchain.hip)
// The arithmetic is k_pc_ma2's (1 / D exact).  dynamic LDS: D * a.lanes floats.
constexpr int PC_MAD_RING = 6;  // wave 0: register sets of either stream (5 blocks in flight)
template <bool OWN>
__global__ __launch_bounds__(128) void k_pc_mad(PostArgs a) {
    __shared__ pc_f4 hand[2][8][64];  // [buffer][0-3: the evicted x of the block, 4-7: s1 of the block][lane]: 16 KiB
    __shared__ float fin[64];         // wave 0's s1 after its last block
    extern __shared__ pc_f4 sring[];  // [D / 16][4][a.lanes]: wave 1's ring of s1 blocks
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wpg = 64 / a.lanes;  // (a.lanes slots per work-group: see k_pc_ma2)
    const int slot = ((int)blockIdx.x / wpg) * 64 + ((int)blockIdx.x % wpg) * a.lanes + lane;
    const int ci = (lane < a.lanes && slot < a.slots) ? a.slot_ci[slot] : -1;
    const bool listed = ci >= 0;
    const bool fresh = listed && a.clients[ci].agc_reset == 2;
    constexpr int KB = 16;
    const int D = a.D, NBD = D / KB;
    __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
    if constexpr (OWN) PC_OWN_SIMD();
    const int T = listed ? a.len[slot] : 0;
    const int nfull = T / KB;
    int nmax = nfull;
#pragma unroll
    for (int d = 32; d; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
    if (lane >= a.lanes) return;
    float *__restrict__ X = a.X + pc_base(slot, a.px);
    float *__restrict__ M1 = a.M1 + pc_base(slot, a.px);
    float *__restrict__ M1n = a.M1n + pc_base(slot, a.px);
    float *__restrict__ V1 = a.V1 + pc_base(slot, a.pv);
    const float fD = (float)D, rD = 1.0f / fD, nrD = -rD;
    if (wid == 0) {
        // ---- wave 0: s1_t = (s1 - x_{t-D}) + x_t
        constexpr int RING = PC_MAD_RING, AHEAD = RING - 1;
        if (fresh)
            for (int r = 0; r < D; r++) X[pc_el(r)] = 0.f;
        float s1 = (fresh || !listed) ? 0.f : a.dc_s1[slot];
        pc_f4 xn[RING][4], xo[RING][4];
        const int dq = D >> 2;
        auto fetch = [&](auto kc, int blk) {  // (unconditional, inside the padding: see k_pc_ma2)
            constexpr int k = decltype(kc)::value;
            const pc_f4 *src = pc_row4(X, (blk * KB) >> 2);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                xo[k][q] = src[q * 64];
                xn[k][q] = src[(size_t)(dq + q) * 64];
            }
        };
        pc_f4 sv[4] = {};
        auto block = [&](auto jc, int b) {
            constexpr int J = decltype(jc)::value;
            if (b < nfull) {
#pragma unroll
                for (int i = 0; i < KB; i++) {
                    s1 = __fadd_rn(__fsub_rn(s1, xo[J][i >> 2][i & 3]), xn[J][i >> 2][i & 3]);
                    sv[i >> 2][i & 3] = s1;
                }
            }
            __syncthreads();  // (publishes the previous block: see k_pc_ma2)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                hand[b & 1][q][lane] = xo[J][q];
                hand[b & 1][4 + q][lane] = sv[q];
            }
            if (b + 1 == nmax) fin[lane] = s1;
        };
        pc_static_for<0, AHEAD>([&](auto kc) { fetch(kc, decltype(kc)::value); });
        int b = 0;
        for (; b + RING <= nmax; b += RING)
            pc_static_for<0, RING>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b + J + AHEAD);
                block(jc, b + J);
            });
        {
            const int b0 = b;
            pc_static_for<0, RING - 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                if (b0 + J < nmax) {
                    fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b0 + J + AHEAD);
                    block(jc, b0 + J);
                }
            });
        }
        // one more evicted block for wave 1: x_{t-D+1} of the last step of block nmax - 1 is the first of block nmax
        {
            const pc_f4 *src = pc_row4(X, (nmax * KB) >> 2);
            pc_f4 t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = src[q * 64];
            if (nmax == 0) fin[lane] = s1;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) hand[nmax & 1][q][lane] = t[q];
        }
        __syncthreads();  // (nmax + 3 barriers in either wave)
        __syncthreads();
        return;
    }
    // ---- wave 1: s2_t = (s2 - m1_{t-D}) + m1_t, out_t = x_{t-D+1} - s2_t / D, three blocks behind wave 0
    if (fresh)
        for (int r = 0; r < D; r++) M1[pc_el(r)] = 0.f;
    float s2 = (fresh || !listed) ? 0.f : a.dc_s2[slot];
    const int vq = (a.vo + a.L - 1) >> 2;
    auto ring = [&](int blk, int q) -> pc_f4 & { return sring[((size_t)(blk & (NBD - 1)) * 4 + q) * a.lanes + lane]; };  // (D: a power of two)
    for (int k = 0; k < NBD; k++)  // the carried averages of the last D steps as sums (s1 = D m1 exactly)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pc_f4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = __fmul_rn(M1[pc_el(k * KB + 4 * q + i)], fD);
            ring(k, q) = v;
        }
    // the evicted x in three register sets (blocks b, b + 1 and the one being read), the incoming s1 in two
    pc_f4 xb[3][4], svs[2][4];
    auto take = [&](auto jc, int r, bool sums) {  // block r (r mod 6 == J) from LDS into its register sets
        constexpr int J = decltype(jc)::value;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xb[J % 3][q] = hand[r & 1][q][lane];
            if (sums) svs[J & 1][q] = hand[r & 1][4 + q][lane];
        }
    };
    auto block = [&](auto jc, int b) {
        constexpr int J = decltype(jc)::value, C = J % 3, N1 = (J + 1) % 3, P = J & 1;
        take(std::integral_constant<int, (J + 1) % 6>{}, b + 1, b + 1 < nmax);  // (block nmax: the evicted x only)
        if (b < nfull) {
            pc_f4 o[4], so[4];
#pragma unroll
            for (int q = 0; q < 4; q++) so[q] = ring(b, q);
#pragma unroll
            for (int i = 0; i < KB; i++) {
                const float s1 = svs[P][i >> 2][i & 3];
                const float t2 = __fmaf_rn(so[i >> 2][i & 3], nrD, s2);
                s2 = __fmaf_rn(s1, rD, t2);
                const float xd = i + 1 < KB ? xb[C][(i + 1) >> 2][(i + 1) & 3] : xb[N1][0][0];
                o[i >> 2][i & 3] = __fmaf_rn(s2, nrD, xd);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ring(b, q) = svs[P][q];
                *pc_row4(V1, vq + b * (KB / 4) + q) = o[q];
            }
        }
        __syncthreads();
    };
    __syncthreads();
    __syncthreads();  // block 0 is in LDS
    take(std::integral_constant<int, 0>{}, 0, nmax > 0);
    __syncthreads();
    int b = 0;
    for (; b + 6 <= nmax; b += 6) pc_static_for<0, 6>([&](auto jc) { block(jc, b + decltype(jc)::value); });
    {
        const int b0 = b;
        pc_static_for<0, 5>([&](auto jc) {
            if (b0 + decltype(jc)::value < nmax) block(jc, b0 + decltype(jc)::value);
        });
    }
    if (!listed) return;
    float s1 = fin[lane];
    // the ring in time order (oldest: the slot block nfull would take) goes to rows t1.. of M1 for the remaining
    // T - nfull * KB (< KB) steps, one by one, as in the two-kernel form
    const int t1 = nfull * KB;
    for (int k = 0; k < NBD; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const pc_f4 v = ring(nfull + k, q);
#pragma unroll
            for (int i = 0; i < 4; i++) M1[pc_el(t1 + k * KB + 4 * q + i)] = __fmul_rn(v[i], rD);
        }
    const int v0 = a.vo + a.L - 1;
    for (int t = t1; t < T; t++) {
        s1 = __fadd_rn(__fadd_rn(s1, -X[pc_el(t)]), X[pc_el(D + t)]);
        const float m1 = __fmul_rn(s1, rD);
        s2 = __fadd_rn(__fadd_rn(s2, -M1[pc_el(t)]), m1);
        M1[pc_el(D + t)] = m1;
        V1[pc_el(v0 + t)] = __fsub_rn(X[pc_el(t + 1)], __fmul_rn(s2, rD));
    }
    for (int i = 0; i < D; i++) M1n[pc_el(i)] = M1[pc_el(T + i)];
    a.dc_s1[slot] = s1;
    a.dc_s2[slot] = s2;
}

// ---- the AGC look-ahead peak: peak_t = max |V1| over rows [t, t + L - 1] (the reference's monotonic deque) ----------
// van Herk: with the rows cut into blocks of L, a window is a suffix of one block and a prefix of the next, so
//   peak_t = max(S[t], P[t + L - 1]),  P[r] = max over [block start, r],  S[r] = max over [r, block end].
// max is associative and exact: this IS parallel along time.  Lane = slot like everything else here; a wave owns one
// SUB-BLOCK (a.sb rows) of one block of 64 slots and walks it row by row - forwards for P, backwards for S, the loads
// 16 rows ahead - and starts from the maximum of the block's other sub-blocks on its side (k_pc_submax: only when the
// look-ahead is longer than a sub-block - 38400 rows at the 192 kHz of the reference's shipped config.toml).
// grid (groups of 64 slots, blocks x sub-blocks); sub-block j of block k: rows [kL + j sb, min(kL + (j + 1) sb, (k + 1) L, rows))
struct PcPiece {
    int slot, k, j, r0, r1;  // r1 <= r0: nothing to do
    bool listed;
};
__device__ __forceinline__ PcPiece pc_piece(const PostArgs &a) {
    PcPiece p;
    p.slot = blockIdx.x * 64 + threadIdx.x;
    p.listed = p.slot < a.slots && a.slot_ci[p.slot] >= 0;
    p.k = blockIdx.y / a.nsub;
    p.j = blockIdx.y - p.k * a.nsub;
    const int rows = p.listed ? a.L - 1 + a.len[p.slot] : 0;
    p.r0 = p.k * a.L + p.j * a.sb;
    p.r1 = min(min(p.r0 + a.sb, (p.k + 1) * a.L), rows);
    return p;
}
constexpr int PC_PEAK_U = 16;  // rows of loads in flight

// SM[group][block * nsub + j][lane] = max |V1| over the sub-block (0 where it is empty)
__global__ __launch_bounds__(64) void k_pc_submax(PostArgs a) {
    const PcPiece p = pc_piece(a);
    float mx = 0.f;
    if (p.listed) {
        const float *v1 = a.V1 + pc_base(p.slot, a.pv);
        for (int r = p.r0; r < p.r1; r += PC_PEAK_U) {
            float v[PC_PEAK_U];
#pragma unroll
            for (int i = 0; i < PC_PEAK_U; i++) v[i] = v1[pc_el(a.vo + min(r + i, p.r1 - 1))];
#pragma unroll
            for (int i = 0; i < PC_PEAK_U; i++) mx = fmaxf(mx, fabsf(v[i]));
        }
    }
    a.SM[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 + threadIdx.x] = mx;
}

// P[r] = max |V1| over [block start, r]
__global__ __launch_bounds__(64) void k_pc_prefix(PostArgs a) {
    const PcPiece p = pc_piece(a);
    if (!p.listed || p.r1 <= p.r0) return;
    const float *v1 = a.V1 + pc_base(p.slot, a.pv);
    float *P = a.P + pc_base(p.slot, a.pv);
    float run = 0.f;
    for (int j = 0; j < p.j; j++) run = fmaxf(run, a.SM[((size_t)blockIdx.x * gridDim.y + (blockIdx.y - p.j + j)) * 64 + threadIdx.x]);
    for (int r = p.r0; r < p.r1; r += PC_PEAK_U) {
        float v[PC_PEAK_U];
#pragma unroll
        for (int i = 0; i < PC_PEAK_U; i++) v[i] = v1[pc_el(a.vo + min(r + i, p.r1 - 1))];
#pragma unroll
        for (int i = 0; i < PC_PEAK_U; i++)
            if (r + i < p.r1) {
                run = fmaxf(run, fabsf(v[i]));
                P[pc_el(a.vo + r + i)] = run;
            }
    }
}

// w_t = desired / (peak_t + 1e-10) for the samples t = r of the sub-block (r < T), peak_t = max(S[t], P[t + L - 1]); w_t goes
// to sample t's row (vo + L - 1 + t, where V1 keeps sample t and k_pc_gain will put g_t) of S.  (t + L - 1 is the
// last row of the same block when t is the block's first row: S[t] is the whole block then, and so is that P.)
__global__ __launch_bounds__(64) void k_pc_want(PostArgs a) {
    const PcPiece p = pc_piece(a);
    if (!p.listed || p.r1 <= p.r0) return;
    const int T = a.len[p.slot], v0 = a.vo + a.L - 1;
    const float *v1 = a.V1 + pc_base(p.slot, a.pv);
    const float *P = a.P + pc_base(p.slot, a.pv);
    float *W = a.S + pc_base(p.slot, a.pv);
    float run = 0.f;
    for (int j = p.j + 1; j < a.nsub; j++) run = fmaxf(run, a.SM[((size_t)blockIdx.x * gridDim.y + (blockIdx.y - p.j + j)) * 64 + threadIdx.x]);
    for (int r = p.r1 - 1; r >= p.r0; r -= PC_PEAK_U) {
        float v[PC_PEAK_U], q[PC_PEAK_U];
#pragma unroll
        for (int i = 0; i < PC_PEAK_U; i++) {
            const int rr = max(r - i, p.r0);
            v[i] = v1[pc_el(a.vo + rr)];
            q[i] = P[pc_el(v0 + min(rr, T - 1 + (T == 0)))];  // (rows r >= T have no sample: clamped, unused)
        }
#pragma unroll
        for (int i = 0; i < PC_PEAK_U; i++)
            if (r - i >= p.r0) {
                run = fmaxf(run, fabsf(v[i]));
                if (r - i < T) W[pc_el(v0 + r - i)] = __fdiv_rn(a.desired, __fadd_rn(fmaxf(run, q[i]), 1e-10f));
            }
    }
}

// the gain recurrence (src/utils/audioprocessing.cpp:55-66); g_t -> sample t's row of P (0 while the
// look-ahead buffer is still filling: the reference outputs 0 there and leaves the gain alone;
// an active gain is never 0: w_t > 0)
// Two waves like k_pc_ma2 (round 5): wave 0 loads w (a ring of register sets, the loads far ahead) and leaves each
// 16-step block in LDS, wave 1 takes it a barrier later and runs the recurrence - no global load in its loop, so its
// s_waitcnt never stands behind the acknowledgement of its own stores (with all 64 lanes in use, 1 KiB per store, the
// one-wave form took 2.2 - 3.0 ms per 512 frames against 1.5 ms with 16 lanes).
constexpr int PC_GAIN_RING = 12;
static_assert(16 * (PC_MA_RING + 4) <= PSDR_PC_PAD && 16 * (PC_GAIN_RING + 4) <= PSDR_PC_PAD, "the loader waves read ahead inside the padding");
template <bool ATT_FASTER, bool OWN>
__global__ __launch_bounds__(128) void k_pc_gain(PostArgs a) {
    __shared__ pc_f4 hand[2][4][64];  // [buffer][row group of the block][lane]: 8 KiB
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wpg = 64 / a.lanes;  // (a.lanes slots per work-group: see k_pc_ma2)
    const int slot = ((int)blockIdx.x / wpg) * 64 + ((int)blockIdx.x % wpg) * a.lanes + lane;
    const int ci = (lane < a.lanes && slot < a.slots) ? a.slot_ci[slot] : -1;
    const bool listed = ci >= 0;
    const int L = a.L;
    const int T = listed ? a.len[slot] : 0;
    float gain = listed ? a.agc_gain[slot] : 0.f;
    int n0 = listed ? a.agc_n0[slot] : 0;
    if (listed && a.clients[ci].agc_reset) {  // AGC::reset() on a demodulation change (src/signal.cpp:322-326)
        gain = 0.f;
        n0 = 0;
    }
    constexpr int KB = 16;
    __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
    if constexpr (OWN) PC_OWN_SIMD();
    // sample t's row in S (w_t) and P (g_t): vo + L - 1 + t, a multiple of 4 at t = 0
    const float *__restrict__ W = a.S + pc_base(slot, a.pv) + (size_t)((a.vo + L - 1) >> 2) * 256;
    float *__restrict__ G = a.P + pc_base(slot, a.pv) + (size_t)((a.vo + L - 1) >> 2) * 256;
    // While the look-ahead buffer is filling (only right after a reset / for a new client) the reference outputs 0 and leaves
    // the gain alone: those steps (t < tfill), then single steps up to a row group (t < t0), then whole blocks, then the rest
    const int tfill = min(T, max(0, L - 1 - n0));
    const int t0 = min(T, (tfill + 3) & ~3);
    const int nblk = (T - t0) / KB;
    int nmax = nblk;
#pragma unroll
    for (int d = 32; d; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
    if (lane >= a.lanes) return;
    if (wid == 0) {
        // ---- wave 0: w blocks from memory to LDS (unconditional loads: up to RING blocks past the stream's end, inside the padding)
        constexpr int RING = PC_GAIN_RING, AHEAD = RING - 1;
        pc_f4 w[RING][4];
        // (a lane that starts late - a look-ahead buffer still filling - is walked to the GROUP's block count: clamped to
        // its own rows, what it reads there is never used)
        const int qlim = (((int)a.pv - (a.vo + L - 1)) >> 2) - KB / 4;
        auto fetch = [&](auto kc, int blk) {
            constexpr int k = decltype(kc)::value;
            const pc_f4 *src = pc_row4(W, min((t0 + blk * KB) >> 2, qlim));
#pragma unroll
            for (int q = 0; q < 4; q++) w[k][q] = src[q * 64];
        };
        auto block = [&](auto jc, int b) {
            constexpr int J = decltype(jc)::value;
            __syncthreads();  // (publishes the previous block: see k_pc_ma2)
#pragma unroll
            for (int q = 0; q < 4; q++) hand[b & 1][q][lane] = w[J][q];
        };
        pc_static_for<0, AHEAD>([&](auto kc) { fetch(kc, decltype(kc)::value); });
        int b = 0;
        for (; b + RING <= nmax; b += RING)
            pc_static_for<0, RING>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b + J + AHEAD);
                block(jc, b + J);
            });
        {
            const int b0 = b;
            pc_static_for<0, RING - 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                if (b0 + J < nmax) {
                    fetch(std::integral_constant<int, (J + AHEAD) % RING>{}, b0 + J + AHEAD);
                    block(jc, b0 + J);
                }
            });
        }
        __syncthreads();
        __syncthreads();
        __syncthreads();
        return;
    }
    // ---- wave 1: the recurrence
    const float att = a.attack, rel = a.release;
    // gain <- gain + (w < gain ? attack : release) * (w - gain): three operations per sample
    // (fma(-a, g - w, g) and fma(a, w - g, g) are the same value: negating both factors is exact)
    // The loop is ONE wave's dependent chain (~8 cycles from an instruction to the next that needs its result), so what
    // counts is the DEPTH per sample.  Both candidates are computed side by side and the choice is a min / max: with
    // attack > release (the reference's 50 ms against 300 ms) d < 0 makes attack * d the smaller product and d > 0 the
    // larger, and fma rounds monotonically - min(A, R) IS the reference's pick, bit for bit (d = 0: both are the gain).
    // sub -> fma, fma -> min: depth 3 instead of 4.  (ATT_FASTER = attack >= release, a template parameter: as a run-time
    // flag the compiler computed min AND max and selected - six instructions and depth 4 again)
    auto step = [&](float w) -> float {
        const float d = __fsub_rn(w, gain);
        const float ga = __fmaf_rn(att, d, gain), gr = __fmaf_rn(rel, d, gain);
        gain = ATT_FASTER ? fminf(ga, gr) : fmaxf(ga, gr);
        return gain;
    };
    for (int t = 0; t < tfill; t++) G[pc_el(t)] = 0.f;
    for (int t = tfill; t < t0; t++) G[pc_el(t)] = step(W[pc_el(t)]);
    pc_f4 wv[2][4];
    auto take = [&](auto pc, int r) {
        constexpr int P = decltype(pc)::value;
#pragma unroll
        for (int q = 0; q < 4; q++) wv[P][q] = hand[r & 1][q][lane];
    };
    auto block = [&](auto pc, int b) {
        constexpr int P = decltype(pc)::value;
        if (b + 1 < nmax) take(std::integral_constant<int, P ^ 1>{}, b + 1);
        if (b < nblk) {
            pc_f4 g[4];
#pragma unroll
            for (int i = 0; i < KB; i++) g[i >> 2][i & 3] = step(wv[P][i >> 2][i & 3]);
#pragma unroll
            for (int q = 0; q < 4; q++) *pc_row4(G, ((t0 + b * KB) >> 2) + q) = g[q];
        }
        __syncthreads();
    };
    __syncthreads();
    __syncthreads();  // block 0 is in LDS
    if (nmax > 0) take(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    int b = 0;
    for (; b + 2 <= nmax; b += 2) {
        block(std::integral_constant<int, 0>{}, b);
        block(std::integral_constant<int, 1>{}, b + 1);
    }
    if (b < nmax) block(std::integral_constant<int, 0>{}, b);
    if (!listed) return;
    for (int t = t0 + nblk * KB; t < T; t++) G[pc_el(t)] = step(W[pc_el(t)]);
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
    const float *G = a.P + pc_base(slot, a.pv), *V = a.V1 + pc_base(slot, a.pv);
    const int g0 = a.vo + a.L - 1 + pos, v0 = a.vo + pos;  // sample pos + j: g in its own row, the delayed sample L - 1 rows earlier
    for (int j = threadIdx.x; j < a.h; j += blockDim.x) {
        int v = 0;
        if (pos >= 0) {
            const float g = G[pc_el(g0 + j)];
            const float y = g == 0.f ? 0.f : __fmul_rn(V[pc_el(v0 + j)], g);
            v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
            v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
        }
        if (a.pcm16)
            reinterpret_cast<int16_t *>(a.pcm)[((size_t)slot * a.max_batch + f) * a.h + j] = (int16_t)v;
        else
            dst[j] = v;
    }
}

// the same with lane = slot (h a multiple of 4): grid (groups of 64 slots, frame), thread (lane, j / 4)
__global__ __launch_bounds__(256) void k_pc_out4(PostArgs a) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63), f = blockIdx.y;
    if (slot >= a.slots || a.slot_ci[slot] < 0) return;
    const int pos = a.fstart[(size_t)slot * a.max_batch + f];
    pc_i4 *dst = reinterpret_cast<pc_i4 *>(a.pcm + ((size_t)slot * a.max_batch + f) * a.h);
    const float *G = a.P + pc_base(slot, a.pv), *V = a.V1 + pc_base(slot, a.pv);
    const int gq = (a.vo + a.L - 1 + max(pos, 0)) >> 2, v0 = a.vo + max(pos, 0);
    for (int j4 = threadIdx.x >> 6; j4 < (a.h >> 2); j4 += 4) {
        pc_i4 o = {0, 0, 0, 0};
        if (pos >= 0) {
            const pc_f4 g = *pc_row4(G, gq + j4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float y = g[i] == 0.f ? 0.f : __fmul_rn(V[pc_el(v0 + 4 * j4 + i)], g[i]);
                int v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;
                o[i] = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
            }
        }
        if (a.pcm16)
            reinterpret_cast<pc_i2 *>(reinterpret_cast<int16_t *>(a.pcm) + ((size_t)slot * a.max_batch + f) * a.h)[j4] = pc_pack16(o);
        else
            dst[j4] = o;
    }
}

// the last D rows of X / M1 and the last L-1 rows of V1 become the history rows of the NEXT set (the sets rotate per
// batch: source and destination never overlap, however short the batch).  grid (listed client); an empty stream (T = 0)
// copies its history as it is.  M1: only on the two-kernel path (k_pc_ma2 moves its own 32 values).
__global__ __launch_bounds__(256) void k_pc_history(PostArgs a) {
    const int slot = a.clients[blockIdx.x].slot;
    const int T = a.len[slot];
    const float *x = a.X + pc_base(slot, a.px), *m = a.M1 + pc_base(slot, a.px), *v = a.V1 + pc_base(slot, a.pv);
    float *xn = a.Xn + pc_base(slot, a.px), *mn = a.M1n + pc_base(slot, a.px), *vn = a.V1n + pc_base(slot, a.pv);
    // (k_pc_ma2 DIRECT: a stream that is the demodulator's rows as they lie may never have been gathered into X - its last D
    // samples are the last D of those rows; T >= D then, pc_direct_ok)
    const bool from_audio = a.direct && T > 0 && pc_direct_ok(true, T, a.nframes * a.h, a.D);
    const float *aud = a.audio + (size_t)slot * a.max_batch * a.h;
    for (int r = threadIdx.x; r < a.D; r += blockDim.x) {
        xn[pc_el(r)] = from_audio ? aud[T - a.D + r] : x[pc_el(r + T)];
        if (!a.ma_fused) mn[pc_el(r)] = m[pc_el(r + T)];
    }
    for (int r = threadIdx.x; r < a.L - 1; r += blockDim.x) vn[pc_el(a.vo + r)] = v[pc_el(a.vo + r + T)];
}

// ==== the AGC in ONE kernel behind chunk maxima (round 6) ==============================================================
// What the chain costs the step is its TRAFFIC (profiles/r06_post_chain_ablation.json: 0.6 - 0.8 % of the step per pass of
// 4 bytes per sample and client over a stream, beside passes that are bound by the memory system): the five kernels above
// - sub-block maxima, prefix maxima, w_t, gain, int16 - read or write a stream eleven times.  This form reads V1 twice
// and writes the PCM once:
//   k_pc_cm      CM[c] = max |V1| over CHUNK c = the 16 floats [16 c, 16 c + 16) of a slot's V1 (history pad included)
//   k_pc_cscan   van Herk one level up: CP / CS = prefix / suffix maxima of CM inside blocks of W = L/16 - 1 chunks
//   k_pc_agc     per chunk b of 16 samples (t = 16 b + i; the window of output step t is V1[t + 1 .. t + L], vo = 1):
//                    peak_t = max( |V1[t+1 .. 16 b + 15]|,  max CM[b+1 .. b+W],  |V1[16 (b + L/16) .. t + L]| )
//                             (a suffix of chunk b, W whole chunks = max(CS[b+1], CP[b+W]), a prefix of chunk b + L/16)
//                    w_t = desired / (peak_t + 1e-10), the gain recurrence, delayed sample V1[t + 1] * gain -> int16
// max is exact and associative: the peak is the reference's monotonic deque's bit for bit, however it is grouped.
// Needs L % 16 == 0 (vo = 1: sample 0's row is chunk L/16's first float), h % 4 == 0, h >= 16 - every audio rate that is a
// multiple of 80 Hz with the usual audio sizes; anything else keeps the kernels above.
//
// k_pc_agc is a pipeline of FOUR waves, one per SIMD of a CU the passes leave free (each names v255 / a255, PC_OWN_SIMD):
//   waves 1-3 (producers)  chunk by chunk: the loads (two rounds ahead, a ring of three register sets), suffix / prefix maxima,
//                          the division, w -> LDS; two rounds later the gains of the same chunk come back through LDS:
//                          delayed sample * gain, int16 conversion, the store to pcm[slot][frame][j] (the frame of a stream
//                          position through FA, the list of a slot's surviving frames)
//   wave 0   (recurrence)  takes a round's w from LDS, runs the gain recurrence (three dependent operations per sample: the
//                          one thing here that cannot be spread), leaves the gains in LDS; no memory operation in its loop
// A work-group owns a.lanes slots; with 32 (16) of them a producer wave works on two (four) chunks at once - lane = (chunk of
// the wave's set, slot) - so the recurrence, not the division, bounds the kernel.  One barrier per ROUND of 3 * 64 / a.lanes
// chunks.  w and g are double-buffered by round parity: producers write w of round r while the recurrence reads round r - 1
// and they read g of round r - 2.
constexpr int PC_AGC_NP = 3;     // producer waves
constexpr int PC_AGC_AHEAD = 2;  // rounds between a producer's loads and their use (~4600 cycles: k_pc_gain's twelve blocks)
static_assert(16 * 3 <= PSDR_PC_PAD, "producers read up to two chunks past the longest stream: inside the padding");

// CM[group][chunk][lane]: grid (groups of 64 slots, ceil(nchunks / 16)), one wave, 16 chunks each
__global__ __launch_bounds__(64) void k_pc_cm(PostArgs a, int nchunks) {
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= a.slots || a.slot_ci[slot] < 0) return;
    const float *v1 = a.V1 + pc_base(slot, a.pv);
    float *cm = a.CM + ((size_t)blockIdx.x * a.nch) * 64 + threadIdx.x;
    const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, nchunks);
    for (int c = c0; c < c1; c += 4) {
        pc_f4 v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) v[k][q] = *pc_row4(v1, 4 * min(c + k, c1 - 1) + q);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 4; i++) m = fmaxf(m, fabsf(v[k][q][i]));
            if (c + k < c1) cm[(size_t)(c + k) * 64] = m;
        }
    }
}

// CP[c] = max CM[block start .. c], CS[c] = max CM[c .. block end], blocks of W chunks: grid (groups, blocks), one wave
__global__ __launch_bounds__(64) void k_pc_cscan(PostArgs a, int nchunks, int W) {
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= a.slots || a.slot_ci[slot] < 0) return;
    const size_t base = ((size_t)blockIdx.x * a.nch) * 64 + threadIdx.x;
    const float *cm = a.CM + base;
    float *cp = a.CP + base, *cs = a.CS + base;
    const int c0 = blockIdx.y * W, c1 = min(c0 + W, nchunks);
    float run = 0.f;
    for (int c = c0; c < c1; c += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = cm[(size_t)min(c + i, c1 - 1) * 64];
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (c + i < c1) {
                run = fmaxf(run, v[i]);
                cp[(size_t)(c + i) * 64] = run;
            }
    }
    run = 0.f;
    for (int c = c1 - 1; c >= c0; c -= 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = cm[(size_t)max(c - i, c0) * 64];
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (c - i >= c0) {
                run = fmaxf(run, v[i]);
                cs[(size_t)(c - i) * 64] = run;
            }
    }
}

// rows of dropped frames (and every row of a paused client) are zero in the PCM: grid (groups of 64 slots, frame) like
// k_pc_out4 - which wrote them on its way; the fused kernel only walks the surviving frames
__global__ __launch_bounds__(256) void k_pc_zero(PostArgs a) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63), f = blockIdx.y;
    if (slot >= a.slots || a.slot_ci[slot] < 0) return;
    if (a.fstart[(size_t)slot * a.max_batch + f] >= 0) return;
    if (a.pcm16) {
        pc_i2 *dst = reinterpret_cast<pc_i2 *>(reinterpret_cast<int16_t *>(a.pcm) + ((size_t)slot * a.max_batch + f) * a.h);
        for (int j4 = threadIdx.x >> 6; j4 < (a.h >> 2); j4 += 4) dst[j4] = pc_i2{0, 0};
        return;
    }
    pc_i4 *dst = reinterpret_cast<pc_i4 *>(a.pcm + ((size_t)slot * a.max_batch + f) * a.h);
    for (int j4 = threadIdx.x >> 6; j4 < (a.h >> 2); j4 += 4) dst[j4] = pc_i4{0, 0, 0, 0};
}

template <bool ATT_FASTER, bool PCM16 = false>
__global__ __launch_bounds__(64 * (1 + PC_AGC_NP)) void k_pc_agc(PostArgs a) {
    constexpr int NP = PC_AGC_NP, AH = PC_AGC_AHEAD;
    // [buffer][chunk of the round (up to NP * 4)][row group][slot lane] - 16 floats per chunk and slot: NP * 64 * 16 floats per
    // round whatever a.lanes is (12 KiB), w and g, two buffers each: 48 KiB
    __shared__ pc_f4 wbuf[2][NP * 64 * 4], gbuf[2][NP * 64 * 4];
    __shared__ pc_f4 dbuf[2][NP * 64 * 4];  // the delayed samples of a round's chunks, from produce(r) to emit(r) (a lane's own: registers are what the ring needs)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lanes = a.lanes, sub = 64 / lanes;  // chunks a producer wave works on at once
    const int cpr = NP * sub;                      // chunks per round
    const int wpg = 64 / lanes;
    const int g64 = (int)blockIdx.x / wpg;                    // group of 64 slots
    const int sl0 = ((int)blockIdx.x % wpg) * lanes;          // first slot lane of the work-group inside its group
    PC_OWN_SIMD();
    const int L = a.L, LC = L >> 4, W = LC - 1;
    // the group's longest stream decides the trip count (lanes of a group may have dropped frames, be paused or not listed)
    int Tm = 0;
    {
        const int s = g64 * 64 + sl0 + (lane & (lanes - 1));
        const int ci = s < a.slots ? a.slot_ci[s] : -1;
        Tm = ci >= 0 ? a.len[s] : 0;
#pragma unroll
        for (int d = 32; d; d >>= 1) Tm = max(Tm, __shfl_xor(Tm, d, 64));
    }
    const int nchk = (Tm + 15) >> 4;               // chunks with a sample
    // (at least two rounds: the producers' timeline below has two rounds in front of and two behind its steady state; a round
    // without samples runs on values nobody uses - the gain stays, the stores go to the dump line)
    const int nrounds = max((nchk + cpr - 1) / cpr, 2);
    auto lds_at = [&](int chunk_in_round, int q, int sl) { return (chunk_in_round * 4 + q) * lanes + sl; };
    if (wid == 0) {
        // ---- the recurrence wave: lane = slot lane
        __builtin_amdgcn_s_setprio(PSDR_PC_SETPRIO);
        const int sl = lane;
        const int slot = g64 * 64 + sl0 + sl;
        const int ci = (sl < lanes && slot < a.slots) ? a.slot_ci[slot] : -1;
        const bool listed = ci >= 0;
        const int T = listed ? a.len[slot] : 0;
        float gain = listed ? a.agc_gain[slot] : 0.f;
        int n0 = listed ? a.agc_n0[slot] : 0;
        if (listed && a.clients[ci].agc_reset) {  // AGC::reset() on a demodulation change (src/signal.cpp:322-326)
            gain = 0.f;
            n0 = 0;
        }
        const float g_init = gain;
        // while the look-ahead buffer is filling (t < tfill: right after a reset / for a new client) the reference outputs 0
        // and leaves the gain alone (src/utils/audioprocessing.cpp:40-54)
        const int tfill = min(T, max(0, L - 1 - n0));
        const float att = a.attack, rel = a.release;
        auto step = [&](float w) -> float {  // (see k_pc_gain: both candidates side by side, the pick is a min / max)
            const float d = __fsub_rn(w, gain);
            const float ga = __fmaf_rn(att, d, gain), gr = __fmaf_rn(rel, d, gain);
            gain = ATT_FASTER ? fminf(ga, gr) : fmaxf(ga, gr);
            return gain;
        };
        const bool mine = sl < lanes;  // (the other lanes idle along, barriers included)
        // round r of the recurrence runs one barrier behind the producers' round r
        __syncthreads();  // producers: w of round 0 written
        for (int r = 0; r < nrounds; r++) {
            const pc_f4 *wb = wbuf[r & 1];
            pc_f4 *gb = gbuf[r & 1];
            pc_f4 wv[4];
            if (mine) {
#pragma unroll
                for (int q = 0; q < 4; q++) wv[q] = wb[lds_at(0, q, sl)];
            }
            for (int k = 0; k < cpr; k++) {
                const int c = r * cpr + k;
                pc_f4 cur[4];
#pragma unroll
                for (int q = 0; q < 4; q++) cur[q] = wv[q];
                if (mine && k + 1 < cpr) {
#pragma unroll
                    for (int q = 0; q < 4; q++) wv[q] = wb[lds_at(k + 1, q, sl)];
                }
                // a chunk that is not wholly inside [tfill, T) for some lane with a stream: the careful form
                const bool whole = 16 * c >= tfill && 16 * c + 16 <= T;
                const bool careful = __any(mine && T > 0 && !whole);
                pc_f4 g[4];
                if (!careful) {
#pragma unroll
                    for (int i = 0; i < 16; i++) g[i >> 2][i & 3] = step(cur[i >> 2][i & 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int t = 16 * c + i;
                        const float before = gain;
                        const float gk = step(cur[i >> 2][i & 3]);
                        if (!(t >= tfill && t < T)) gain = before;  // buffer still filling / past the lane's stream: the gain stays
                        g[i >> 2][i & 3] = t < tfill ? 0.f : gk;
                    }
                }
                if (mine) {
#pragma unroll
                    for (int q = 0; q < 4; q++) gb[lds_at(k, q, sl)] = g[q];
                }
            }
            __syncthreads();
        }
        __syncthreads();  // (the producers' two rounds of output behind the last recurrence round)
        __syncthreads();
        if (listed) {
            a.agc_gain[slot] = T > 0 ? gain : g_init;
            a.agc_n0[slot] = min(n0 + T, L);
        }
        return;
    }
    // ---- producer wave p: chunk (r * cpr + p * sub + j) of round r for lane = (j, slot lane)
    const int p = wid - 1;
    const int sl = lane & (lanes - 1), j = lane / lanes;
    const int slot = g64 * 64 + sl0 + sl;
    const int ci = slot < a.slots ? a.slot_ci[slot] : -1;
    const bool listed = ci >= 0;
    const int T = listed ? a.len[slot] : 0;
    const float *__restrict__ V = a.V1 + pc_base(min(slot, a.slots - 1), a.pv);
    const size_t cbase = ((size_t)g64 * a.nch) * 64 + (size_t)(sl0 + sl);
    const float *__restrict__ CS = a.CS + cbase;
    const float *__restrict__ CP = a.CP + cbase;
    const int *__restrict__ FA = a.falive + ((size_t)g64 * a.max_batch) * 64 + (size_t)(sl0 + sl);
    // (PCM16: the rows as int16 - the same buffer, half of it; a row group is 8 bytes then)
    using pcm_t = std::conditional_t<PCM16, int16_t, int32_t>;
    pcm_t *__restrict__ pcm = reinterpret_cast<pcm_t *>(a.pcm) + (size_t)min(slot, a.slots - 1) * a.max_batch * a.h;
    pcm_t *dump = reinterpret_cast<pcm_t *>(a.pcm_dump + 4 * (int)threadIdx.x);
    const int kin = p * sub + j;  // the lane's chunk inside a round
    const int h = a.h;
    const unsigned hmagic = a.h_magic;  // ceil(2^32 / h): t / h = umulhi(t, hmagic) for t * h < 2^32
    struct Set {
        pc_f4 nr[4], fr[4];
        float nx;  // the first float of the next chunk (the delayed sample of the chunk's last step)
        float cs, cp;
        int fa0, fa1;
    };
    Set ring[AH + 1];
    int dst[2][4];     // by round parity: the places of a chunk's row groups in the PCM (int32 index from the slot's base; < 0: no sample)
    auto fetch = [&](auto kc, int r) {  // UNCONDITIONAL (see k_pc_ma2): past the stream's end inside the padding, values never used
        constexpr int k = decltype(kc)::value;
        const int c = min(r * cpr + kin, nchk);  // (the rounds past the last chunk read its neighbour again)
        const pc_f4 *n = pc_row4(V, 4 * c), *f = pc_row4(V, 4 * (c + LC));
#pragma unroll
        for (int q = 0; q < 4; q++) ring[k].nr[q] = n[(size_t)q * 64];
        ring[k].nx = *reinterpret_cast<const float *>(n + (size_t)4 * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) ring[k].fr[q] = f[(size_t)q * 64];
        ring[k].cs = CS[(size_t)(c + 1) * 64];
        ring[k].cp = CP[(size_t)(c + W) * 64];
        const int k0 = min((int)__umulhi((unsigned)(16 * c), hmagic), a.max_batch - 1), k1 = min(k0 + 1, a.max_batch - 1);
        ring[k].fa0 = FA[(size_t)k0 * 64];
        ring[k].fa1 = FA[(size_t)k1 * 64];
    };
    auto produce = [&](auto kc, auto pc, int r) {
        constexpr int k = decltype(kc)::value, P = decltype(pc)::value;
        const Set &s = ring[k];
        const int c = r * cpr + kin;
        // suffix maxima of |chunk c| from position i + 1, prefix maxima of |chunk c + L/16| up to position i
        float sf[17], pf[16];
        sf[16] = 0.f;
#pragma unroll
        for (int i = 15; i >= 1; i--) sf[i] = fmaxf(sf[i + 1], fabsf(s.nr[i >> 2][i & 3]));
        pf[0] = fabsf(s.fr[0][0]);
#pragma unroll
        for (int i = 1; i < 16; i++) pf[i] = fmaxf(pf[i - 1], fabsf(s.fr[i >> 2][i & 3]));
        const float mid = fmaxf(s.cs, s.cp);
        pc_f4 w[4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float peak = fmaxf(fmaxf(sf[i + 1], mid), pf[i]);
            w[i >> 2][i & 3] = __fdiv_rn(a.desired, __fadd_rn(peak, 1e-10f));
        }
        // (the chunk's first float is in no window of its own steps; without a use the compiler recycles its register right
        // behind the load - and waits for the load to land first, a round trip to memory in every round)
        asm volatile("" ::"v"(s.nr[0][0]));
        pc_f4 *wb = wbuf[r & 1];
#pragma unroll
        for (int q = 0; q < 4; q++) wb[lds_at(kin, q, sl)] = w[q];
        // the delayed samples V1[t + 1] and where they go: frame ordinal k = t / h of row group q (a chunk touches at most
        // two frames: h >= 16), the surviving frame FA[k], offset t - k h
        const int t0 = 16 * c;
        const int k0 = (int)__umulhi((unsigned)t0, hmagic);
        pc_f4 *db = dbuf[P] + ((wid - 1) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pc_f4 dl;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = 4 * q + i + 1;
                dl[i] = e < 16 ? s.nr[(e >> 2) & 3][e & 3] : s.nx;
            }
            db[q * 64] = dl;
            const int t = t0 + 4 * q;
            const int second = t >= __mul24(k0 + 1, h) ? 1 : 0;
            const int fr = second ? s.fa1 : s.fa0;
            // (an opaque 24-bit multiplication: the compiler's own is a 64-bit multiply-add whose don't-care upper addend it
            // leaves in whatever register - one a load of this round is still writing: a wait for that load)
            const int off = pc_mul24(fr - (k0 + second), h) + t;
            dst[P][q] = t < T ? off : -1;  // (T = 0 for a slot that is not listed)
        }
    };
    auto emit = [&](auto pc, int r) {  // the gains of round r are in gbuf[r & 1]
        constexpr int P = decltype(pc)::value;
        const pc_f4 *gb = gbuf[r & 1];
        const pc_f4 *db = dbuf[P] + ((wid - 1) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const pc_f4 g = gb[lds_at(kin, q, sl)], dl = db[q * 64];
            pc_i4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float y = g[i] == 0.f ? 0.f : __fmul_rn(dl[i], g[i]);
                int v = (int)__fmaf_rn(y, 16384.f, 32768.5f) - 32768;  // dsp_float_to_int16, src/utils/dsp.cpp:152-165
                o[i] = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
            }
            // (an UNCONDITIONAL store: a row group without a sample goes to a dump line of its own - see the timeline below)
            pcm_t *where = dst[P][q] >= 0 ? pcm + dst[P][q] : dump;
            if constexpr (PCM16)
                *reinterpret_cast<pc_i2 *>(where) = pc_pack16(o);
            else
                *reinterpret_cast<pc_i4 *>(where) = o;
        }
    };
    // Producer timeline: round rr sits between barriers #rr and #rr+1: fetch(rr + AH), emit(rr - 2) [its gains were written by the
    // recurrence during round rr - 1], produce(rr) [read by the recurrence one barrier later].  The delayed samples (dbuf) and dst of round rr live until
    // emit(rr) two rounds later: two parities suffice because emit(rr - 2) comes BEFORE produce(rr).  The steady state
    // (rounds 2 .. nrounds - 1) is branch-free - every fetch, store and LDS operation unconditional - so that the compiler
    // can COUNT the memory operations between a load and its use (a conditional one turns every wait into vmcnt(0): a drain
    // of two rounds of prefetch and of the stores behind them); the first two and the last two rounds stand outside.
    static_assert(AH == 2, "ring position = round mod 3, parity = round mod 2: six rounds per trip");
    pc_static_for<0, AH>([&](auto kc) { fetch(kc, decltype(kc)::value); });
    {   // rounds 0 and 1: nothing to emit yet
        fetch(std::integral_constant<int, 2>{}, AH);
        produce(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        __syncthreads();
        fetch(std::integral_constant<int, 0>{}, AH + 1);
        produce(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 1);
        __syncthreads();
    }
    auto round = [&](auto jc, auto pc, int rr) {
        constexpr int J = decltype(jc)::value;  // rr mod 3
        // (scheduling fences: the scheduler otherwise hoists the first uses of a LATER round's loads - the |x| of its maxima - to
        // the top of the six-round block, and the wait for them with it: the ring's two rounds of distance gone)
        fetch(std::integral_constant<int, (J + AH) % (AH + 1)>{}, rr + AH);
        __builtin_amdgcn_sched_barrier(0);
        emit(pc, rr - 2);
        __builtin_amdgcn_sched_barrier(0);
        produce(jc, pc, rr);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    int r = 2;  // (2 mod 3 = 2, 2 mod 2 = 0)
    for (; r + 6 <= nrounds; r += 6) {
        round(I2{}, I0{}, r);
        round(I0{}, I1{}, r + 1);
        round(I1{}, I0{}, r + 2);
        round(I2{}, I1{}, r + 3);
        round(I0{}, I0{}, r + 4);
        round(I1{}, I1{}, r + 5);
    }
    if (r < nrounds) round(I2{}, I0{}, r), r++;
    if (r < nrounds) round(I0{}, I1{}, r), r++;
    if (r < nrounds) round(I1{}, I0{}, r), r++;
    if (r < nrounds) round(I2{}, I1{}, r), r++;
    if (r < nrounds) round(I0{}, I0{}, r), r++;
    // the last two rounds' gains (rounds nrounds and nrounds + 1 of the timeline: emit only)
    for (int e = 0; e < 2; e++) {
        const int rr = nrounds + e;
        if (rr & 1)
            emit(std::integral_constant<int, 1>{}, rr - 2);
        else
            emit(std::integral_constant<int, 0>{}, rr - 2);
        __syncthreads();
    }
    __syncthreads();  // (nrounds + 3 in either kind of wave)
}

}  // namespace psdr
