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
//    work-group owns a CU.  Memory time and butterfly time must overlap inside that one
//    work-group: kernels are persistent (one work-group per CU walks the tile slots) and
//    the global loads of the NEXT tile are issued into registers, a few at a time, between
//    the current tile's stages.
//  * that needs > 128 VGPRs, so a work-group is 512 threads (2 waves/SIMD, 256 VGPRs).
//  * the passes are VALU-ISSUE bound (ISA count x 4 cycles x 2 waves/SIMD = the measured
//    tile time; de-phasing the CUs or running two 64 KiB work-groups per CU changes
//    nothing), so the schedule minimises instructions, not flops: every thread owns the
//    16 points {i0 + e*L/16} of TWO adjacent sequences (a "column couple", type c2) at
//    every stage and performs 16/R radix-R butterflies per stage (R in {16,8,4,2}) on
//    both at once.  The couple shares every LDS address, every stage twiddle and every LDS
//    instruction (16-byte ds_read/ds_write_b128 carry both), and its two outputs are
//    adjacent in HBM: every global store is a natural 16-byte store.
//  * lanes run along the T (contiguous-in-HBM) dimension: 8 lanes cover a 128-byte row of
//    the tile, stage traffic in LDS is conflict-free (pass 2 XOR-swizzles the couple index
//    so that its transposing fill is conflict-free as well).
//  * HBM throughput is proportional to the bytes a load instruction carries (dword 1.9,
//    dwordx2 3.4, dwordx4 5.6 TB/s on 128-byte segments): raw samples are fetched with
//    16-byte loads in row order, parked in LDS as an image, and the threads pick their
//    strided points out of LDS.
//  * the Hann window is evaluated on the fly from the twiddle tables
//    (w = 0.5 - 0.5*Re(W_M1^{n1} * W_M^{n2})), the inter-pass twiddles by short power
//    recurrences from three table look-ups per column: no per-point table traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "butterfly.h"
#include "quantize.h"

#define PSDR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// phase timestamps of work-group 0 (debug builds of the tuning tools only)
#ifdef PSDR_TRACE_ON
#define PSDR_TRACE(buf, it, k)                                                      \
    do {                                                                            \
        if ((buf) && threadIdx.x == 0 && blockIdx.x == 0 && (it) < 8) {             \
            (buf)[(it) * 16 + (k)] = __builtin_readcyclecounter();                  \
            if ((k) == 0) (buf)[(it) * 16 + 14] = wall_clock64(); /* 100 MHz */     \
            if ((k) == 10) (buf)[(it) * 16 + 15] = wall_clock64();                  \
        }                                                                           \
    } while (0)
// wall-clock (100 MHz) timeline of EVERY work-group: [wg][8] after the 256 phase stamps
#define PSDR_WGTRACE(buf, slot)                                                           \
    do {                                                                                  \
        if ((buf) && threadIdx.x == 0 && blockIdx.x < 256 && (slot) < 8)                  \
            (buf)[256 + blockIdx.x * 8 + (slot)] = wall_clock64();                        \
    } while (0)
#else
#define PSDR_TRACE(buf, it, k) \
    do {                       \
    } while (0)
#define PSDR_WGTRACE(buf, slot) \
    do {                        \
    } while (0)
#endif

// The passes fetch the NEXT tile while they work on the current one, under `if (there is a next tile)`.  The fused real
// second pass issues those loads unconditionally (counted waits behind them: see there).  For the plain first pass and
// the IQ second pass the same change measured -1.2 ... -1.4 % on cfg2 (same box, three interleaved repetitions: the IQ
// second pass goes from 212 to 247 VGPRs and +1.4 %): conditional there.

namespace psdr {

// Device-clock stamps of a launch (psdr_set_profiling mode 2): k[0] = earliest work-group entry, k[1] = latest
// work-group exit (after its stores were acknowledged) on the constant 100 MHz clock.  Two fire-and-forget
// atomics per work-group and launch: the duration of a pass is measured inside the very loop that is being
// timed, with no marker packets between the kernels (hipEvent brackets lengthen the passes by ~9 %).
__device__ __forceinline__ void kclk_begin(unsigned long long *k) {
    if (k && threadIdx.x == 0) __hip_atomic_fetch_min(k, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void kclk_end(unsigned long long *k) {
    if (k) {  // uniform
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_max(k + 1, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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

// One Stockham stage on the 16 points (of both columns) a thread owns.  P = product of
// earlier radices.  butterfly i = i0 + b*L/16, k = i mod P, j = (i-k)*R + k
//   x_q = u[b + q*16/R] * W_L^{q*k*L/(P*R)};  out[j + s*P] = DFT_R(x)[s]
// emit(b, s, pos, value)
// The stage twiddles of a thread do not depend on the tile (k is a function of i0 and b only): StageTw keeps
// them in registers for the whole persistent loop instead of reading the LDS table once per tile (27 of the 43
// table reads per thread and tile of a 1024-point pass; under the package power cap of DESIGN.md 5.2 an LDS byte
// is time).  Pass 1 keeps the second stage's (TWM); in pass 2 the same hoist measures nothing.
template <int L>
struct StageTw {
    using P = Plan<L>;
    static constexpr int NB1 = 16 / P::R1, NB2 = P::NS == 3 ? 16 / P::R2 : 1;
    cf s1[NB1][P::R1 - 1];                          // stage 1: W_L^{q k L/(R0 R1)}, k = i & (R0 - 1)
    cf s2[NB2][(P::NS == 3 ? P::R2 : 2) - 1];       // stage 2: W_L^{q k L/(R0 R1 R2)}, k = i & (R0 R1 - 1)
    __device__ __forceinline__ void load(const cf *W, int i0) {  // W: the L-entry table (global or LDS)
        constexpr int L16 = L / 16;
#pragma unroll
        for (int b = 0; b < NB1; b++) {
            const int k = (i0 + b * L16) & (P::R0 - 1);
#pragma unroll
            for (int q = 1; q < P::R1; q++) s1[b][q - 1] = W[q * k * (L / (P::R0 * P::R1))];
        }
        if constexpr (P::NS == 3) {
#pragma unroll
            for (int b = 0; b < NB2; b++) {
                const int k = (i0 + b * L16) & (P::R0 * P::R1 - 1);
#pragma unroll
                for (int q = 1; q < P::R2; q++) s2[b][q - 1] = W[q * k * (L / (P::R0 * P::R1 * P::R2))];
            }
        }
    }
};

// twr: nullptr (twiddles from the table Wl) or this stage's rows of a StageTw, [NB][R - 1]
template <int L, int R, int P, typename Emit>
__device__ __forceinline__ void stage_compute(c2 (&u)[16], int i0, const cf *Wl, Emit emit, const cf (*twr)[R - 1] = nullptr) {
    constexpr int NB = 16 / R;
    constexpr int L16 = L / 16;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int i = i0 + b * L16;
        const int k = i & (P - 1);
        const int j = (i - k) * R + k;
        c2 x[R];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = u[b + q * NB];
        if (P > 1) {
            const int step = k * (L / (P * R));
            if (twr) {
#pragma unroll
                for (int q = 1; q < R; q++) x[q] = cmul(x[q], twr[b][q - 1]);
            } else {
#pragma unroll
                for (int q = 1; q < R; q++) x[q] = cmul(x[q], Wl[q * step]);
            }
        }
        dftR<R>(x);
#pragma unroll
        for (int s = 0; s < R; s++) emit(b, s, j + s * P, x[s]);
        if (NB > 1) PSDR_SCHED_FENCE();
    }
}

// LDS slot (16 bytes: one point of a column couple) of (row = position in the sequence,
// p = couple).  SWZ spreads the transposing fill of pass 2 over the banks; it only uses
// row bits 1.. so that rows n2, n2+1 (one 16-byte global load) share it.
template <int H, bool SWZ>
__device__ __forceinline__ int lds_slot(int row, int p) {
    return row * H + (SWZ ? (p ^ ((row >> 1) & (H - 1))) : p);
}
__device__ __forceinline__ float4 pack_c2(c2 v) { return make_float4(v.a.x, v.a.y, v.b.x, v.b.y); }
__device__ __forceinline__ c2 unpack_c2(float4 v) { return c2{make_float2(v.x, v.y), make_float2(v.z, v.w)}; }

template <int L, int H, bool SWZ>
__device__ __forceinline__ void tile_read(c2 (&u)[16], const float4 *tile, int i0, int p) {
#pragma unroll
    for (int e = 0; e < 16; e++) u[e] = unpack_c2(tile[lds_slot<H, SWZ>(i0 + e * (L / 16), p)]);
}

// All stages of one thread; u holds the stage-0 input.
//   pre_last()    runs right before the last-stage butterflies (twiddle set-up)
//   emit_last(b, s, pos, value)  receives the last stage's outputs
// When the last stage starts, every thread has passed the barrier that follows the last
// LDS read (the tile is dead and may be reused by emit_last).
//   tick(k)       k < 2*(NS-1): called after each stage's LDS writes (even k) and after the
//                 read-back (odd k): the caller issues a slice of the next tile's global
//                 loads there, so they trickle through the compute phases instead of
//                 blocking the memory pipe in one burst
// every stage but the last (the last stage's outputs go to the caller's emit)
template <int L, int T, bool SWZ, bool kX2 = true, typename Tick, typename Mark>
__device__ __forceinline__ void run_front_stages(float4 *tile, const cf *Wl, int i0, int p, c2 (&u)[16], Tick tick,
                                                 Mark mark, const StageTw<L> *stw = nullptr) {
    using P = Plan<L>;
    constexpr int H = T / 2;
    constexpr bool kX1 = true;  // (kX1 / kX2 = false: timing-only ablations - a stage without its LDS exchange; docs/history.md 5.2)
#define PSDR_P1_BARRIER() __syncthreads()
    if constexpr (kX1) {
        stage_compute<L, P::R0, 1>(u, i0, Wl,
                                   [&](int, int, int pos, c2 x) { tile[lds_slot<H, SWZ>(pos, p)] = pack_c2(x); });
    } else {
        c2 v[16];
        stage_compute<L, P::R0, 1>(u, i0, Wl, [&](int b, int s, int, c2 x) { v[b + s * (16 / P::R0)] = x; });
#pragma unroll
        for (int e = 0; e < 16; e++) u[e] = v[e];
    }
    PSDR_SCHED_FENCE();
    tick(0);
    PSDR_SCHED_FENCE();
    mark(4);
    if constexpr (kX1) {
        PSDR_P1_BARRIER();
        mark(5);
        tile_read<L, H, SWZ>(u, tile, i0, p);
        PSDR_P1_BARRIER();
    }
    mark(6);
    PSDR_SCHED_FENCE();
    tick(1);
    PSDR_SCHED_FENCE();
    if constexpr (P::NS == 3) {
        if constexpr (kX2) {
            stage_compute<L, P::R1, P::R0>(
                u, i0, Wl, [&](int, int, int pos, c2 x) { tile[lds_slot<H, SWZ>(pos, p)] = pack_c2(x); },
                stw ? stw->s1 : nullptr);
        } else {
            c2 v[16];
            stage_compute<L, P::R1, P::R0>(u, i0, Wl, [&](int b, int s, int, c2 x) { v[b + s * (16 / P::R1)] = x; });
#pragma unroll
            for (int e = 0; e < 16; e++) u[e] = v[e];
        }
        PSDR_SCHED_FENCE();
        tick(2);
        PSDR_SCHED_FENCE();
        mark(7);
        if constexpr (kX2) {
            PSDR_P1_BARRIER();
            mark(8);
            tile_read<L, H, SWZ>(u, tile, i0, p);
            PSDR_P1_BARRIER();
        }
        mark(9);
        PSDR_SCHED_FENCE();
        tick(3);
        PSDR_SCHED_FENCE();
    }
#undef PSDR_P1_BARRIER
}
template <int L, typename EmitLast>
__device__ __forceinline__ void run_last_stage(const cf *Wl, int i0, c2 (&u)[16], EmitLast emit_last,
                                               const StageTw<L> *stw = nullptr) {
    if constexpr (Plan<L>::NS == 3) {
        stage_compute<L, LastStage<L>::R, LastStage<L>::Pp>(
            u, i0, Wl, [&](int b, int s, int pos, c2 x) { emit_last(b, s, pos, x); }, stw ? stw->s2 : nullptr);
    } else {
        stage_compute<L, LastStage<L>::R, LastStage<L>::Pp>(
            u, i0, Wl, [&](int b, int s, int pos, c2 x) { emit_last(b, s, pos, x); }, stw ? stw->s1 : nullptr);
    }
}
template <int L, int T, bool SWZ, bool kX2 = true, typename PreLast, typename EmitLast, typename Tick, typename Mark>
__device__ __forceinline__ void run_stages(float4 *tile, const cf *Wl, int i0, int p, c2 (&u)[16],
                                           PreLast pre_last, EmitLast emit_last, Tick tick, Mark mark,
                                           const StageTw<L> *stw_front = nullptr, const StageTw<L> *stw_last = nullptr) {
    run_front_stages<L, T, SWZ, kX2>(tile, Wl, i0, p, u, tick, mark, stw_front);
    pre_last();
    run_last_stage<L>(Wl, i0, u, emit_last, stw_last);
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

// Dynamic tile hand-out for the persistent passes.  Work-group b takes sequence indices b
// and b + gridDim.x first; every further index is drawn from a ticket counter, one counter
// per XCD (index s belongs to XCD s%8, as in a static round-robin) or, where XCD affinity buys
// nothing, from one chip-wide counter.  Work-groups of one launch do not start together (they share the
// CUs with the previous batch's consumer kernels and with the other pass): with tickets a
// late starter simply takes fewer tiles.  A ticket is drawn a whole tile ahead of its use
// (the atomic goes to memory through the same queues as the passes' HBM streams).
//   tickets[8] must be zero at launch; gridDim.x is a multiple of 8 or >= `total`.
struct TileQueue {
    unsigned *tickets;
    unsigned total, base;  // indices below 2*gridDim.x are the static first tiles
    unsigned first_dyn;    // index of ticket 0 of the chip-wide counter (2*gridDim.x)
    unsigned pending;      // owner thread: ticket drawn, not yet examined
    unsigned ptx;          // ... and the XCD whose counter it came from
    unsigned owner;        // the thread that draws (0 unless the kernel has a loader wave)
    unsigned level;        // 0: drawing from the own XCD's queue, k: from the queue of XCD x ^ k
    unsigned vb, vgrid;    // this work-group's index among the work-groups that share the queue, and their number
    bool dynamic, global, own_done;
    // global_: ONE counter for the whole chip (perfect balance, no XCD affinity)
    __device__ __forceinline__ void init(unsigned *t, unsigned total_, bool global_, unsigned owner_, unsigned vb_, unsigned vgrid_) {
        global = global_;
        owner = owner_;
        tickets = t;
        total = total_;
        vb = vb_;
        vgrid = vgrid_;
        base = vgrid >> 2;  // 2*vgrid / 8
        first_dyn = 2u * vgrid;
        pending = 0;
        ptx = vb & 7u;
        own_done = false;
        level = 0;
        dynamic = t != nullptr && 2u * vgrid < total_;
    }
    __device__ __forceinline__ void init(unsigned *t, unsigned total_, bool global_ = false, unsigned owner_ = 0) {
        init(t, total_, global_, owner_, blockIdx.x, gridDim.x);
    }
    // thread 0: start drawing (no wait)
    __device__ __forceinline__ void draw_begin() {
        if (threadIdx.x == owner && dynamic) {
            // The address goes through a vector register the compiler cannot see through: with a
            // uniform address the atomic optimiser rewrites this into "one lane adds, v_readfirstlane
            // broadcasts", and the broadcast needs the result AT ONCE - an s_waitcnt vmcnt(0) at the top of
            // every tile, which (vmcnt counts stores too, in order) also waits for the acknowledgement of
            // every store of the previous tile.  Written this way the result is first touched one tile
            // later, behind a counted wait (vmcnt(22) in the IQ pass 2).  Measured: nothing for the IQ
            // passes (pass 2 moves 4.9 GB per 256 frames at 5.6 TB/s - it waits for memory either way),
            // -3 % on the fused real-input pass 2 together with the scalar twiddle load there.
            typedef __attribute__((address_space(1))) unsigned gu32;  // global, not flat: flat returns out of order
            ptx = (vb & 7u) ^ level;
            gu32 *p = (gu32 *)(tickets + (global ? 0u : ptx));
            asm volatile("" : "+v"(p));
            pending = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the first draw, before the tile loop: waits for it (and the first tile's loads, needed at once
    // anyway), so that the loop is entered with nothing pending and the wait the compiler places in
    // draw_end() counts only what one iteration issues after the atomic (loads of the next tile, stores
    // of this one) instead of falling back to vmcnt(0)
    __device__ __forceinline__ void draw_first() {
        draw_begin();
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
    }
    // thread 0: finish the draw begun one tile ago and publish the index (or 0xFFFFFFFF) to
    // *slot; `prev` is the index two positions earlier in this work-group's sequence
    // (returns the published index - meaningful in the owner thread only)
    __device__ __forceinline__ unsigned draw_end(unsigned *slot, unsigned prev2) {
        unsigned s = 0xFFFFFFFFu;
        if (threadIdx.x == owner) {
            (void)prev2;
            if (dynamic) {
                s = global ? pending + first_dyn : (pending + base) * 8u + ptx;
                constexpr unsigned kStealLevels = 1;  // queues of other XCDs a work-group goes on with after its own: x ^ 1 (three or all
                                                      // seven measured -3 ... -4 %: docs/history.md section 5)
                while (!global && s >= total && level < kStealLevels) {
                    // The own queue is empty: go on with the queue of the neighbouring XCD (x ^ 1).  Pass 1's
                    // even XCDs are consistently ~2 % slower than the odd ones (tools/trace_phases.py, every
                    // box seen), which left the chip half idle for the last 12-17 us of every launch.  One
                    // synchronous draw per work-group (its result is needed now), asynchronous ones after.
                    // (Probing all seven other counters was tried in round 1: +27 us per launch.)
                    level++;
                    own_done = true;
                    ptx = (vb & 7u) ^ level;
                    typedef __attribute__((address_space(1))) unsigned gu32;
                    gu32 *p = (gu32 *)(tickets + ptx);
                    asm volatile("" : "+v"(p));
                    const unsigned t2 = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s = (t2 + base) * 8u + ptx;
                }
                if (s >= total) s = 0xFFFFFFFFu;
            }
            *slot = s;
        }
        return s;
    }
};

struct Pass1Args {
    const void *raw;  // nframes+1 raw half-frames, contiguous
    cf *Y;            // !PAIR: [nframes][pass-1 tile][M1][T]; PAIR: [nframes][pass-2 tile][pass-1 tile][row in tile][T]
    const cf *Wl;     // W_L^j, j < L (L = M1)
    const cf *TB;     // W_M^l, l < M2
    cf wdelta;        // W_N^1 (real input: window angle of the odd sample)
    int M2;
    int log2M2;
    int fmt;
    int is_real;  // window pairs (w[2n], w[2n+1]) instead of (w[n], w[n])
    int rot;      // IQ: produce client order
    size_t yblk;    // !PAIR: elements of one pass-1 tile's block (M1 * T)
    int l2t2;       // PAIR: log2 of the rows per pass-2 tile
    size_t ytile;   // PAIR: elements between the chunks this pass-1 tile contributes to consecutive pass-2 tiles
                    //       (tile-major: M2 * 16, the pass-2 tile's block; blocked: 16 * T, inside its own block)
    size_t ytl;     // PAIR: elements between the first chunks of consecutive pass-1 tiles (tile-major: 16 * T; blocked: M1 * T)
    size_t yframe;  // elements between frames of Y
    unsigned tiles_per_frame;
    unsigned total_slots;
    unsigned *tickets;  // TileQueue counters of this launch (8, zeroed)
    unsigned long long *trace;
    unsigned long long *kclk;  // device-clock stamps of this launch (kclk_begin / kclk_end) or nullptr
    unsigned ymask;            // frame index mask of Y (~0u; a timing-only experiment aliases frames: PSDR_Y_ALIAS)
    float yscale;              // a power of two carried by the window weights, so that Y - and with it the second pass's
                               // outputs - arrive scaled: 1/N for IQ input, 0.5/N for the fused real path (the untangle's
                               // 1/2 with it), 1 for the three-pass real path.  Scaling by a power of two commutes with every
                               // rounding of the transform (src/fft_impl.cpp:156-160 divides last): no bit of X changes.
};

// the raw words of two adjacent complex samples (columns 2p, 2p+1 of one row) -> c2
// (src/samplereader.cpp:29-40): unsigned formats flip the MSB, integers are divided by
// 2^(bits-1) (exact: multiply by the reciprocal).
// One instruction per component: v_cvt_f32_i32 with an SDWA source selector picks the (sign-extended) byte or half-word
// itself - the plain C form is a bit-field extract plus a conversion per component, and the conversion of a tile's raw
// words was a sixth of pass 1's vector instructions.
template <int HALF>
__device__ __forceinline__ float cvt_s16(unsigned x) {
    float f;
    if constexpr (HALF == 0)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(x));
    else
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(x));
    return f;
}
template <int B>
__device__ __forceinline__ float cvt_s8(unsigned x) {
    float f;
    if constexpr (B == 0)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f) : "v"(x));
    else if constexpr (B == 1)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f) : "v"(x));
    else if constexpr (B == 2)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f) : "v"(x));
    else
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f) : "v"(x));
    return f;
}
// the raw words of a couple -> c2, UNSCALED integers as floats (the caller folds 2^-(bits-1) into its window weights).
// FLIP: the unsigned formats' MSB flip (src/samplereader.cpp:29-40) - a template parameter so that the signed formats
// (the BASELINE configurations' s16) pay nothing for it
template <int SB, bool FLIP>
__device__ __forceinline__ c2 words_to_c2_fast(const unsigned (&w)[SB / 2]) {
    if constexpr (SB == 2) {
        const unsigned v = FLIP ? (w[0] ^ 0x80808080u) : w[0];
        return c2{make_float2(cvt_s8<0>(v), cvt_s8<1>(v)), make_float2(cvt_s8<2>(v), cvt_s8<3>(v))};
    } else if constexpr (SB == 4) {
        const unsigned x = FLIP ? (w[0] ^ 0x80008000u) : w[0], y = FLIP ? (w[1] ^ 0x80008000u) : w[1];
        return c2{make_float2(cvt_s16<0>(x), cvt_s16<1>(x)), make_float2(cvt_s16<0>(y), cvt_s16<1>(y))};
    } else {
        return c2{make_float2(__uint_as_float(w[0]), __uint_as_float(w[1])),
                  make_float2(__uint_as_float(w[2]), __uint_as_float(w[3]))};
    }
}
template <int SB, bool SCALED>
__device__ __forceinline__ c2 words_to_c2(const unsigned (&w)[SB / 2], int fmt) {
    if constexpr (SB == 2) {
        const unsigned v = w[0] ^ (fmt == 0 ? 0x80808080u : 0u);
        const float k = SCALED ? 1.0f / 128.0f : 1.0f;
        return c2{make_float2((float)(int8_t)(v & 0xFFu) * k, (float)(int8_t)((v >> 8) & 0xFFu) * k),
                  make_float2((float)(int8_t)((v >> 16) & 0xFFu) * k, (float)(int8_t)(v >> 24) * k)};
    } else if constexpr (SB == 4) {
        const unsigned flip = fmt == 2 ? 0x80008000u : 0u;
        const unsigned x = w[0] ^ flip, y = w[1] ^ flip;
        const float k = SCALED ? 1.0f / 32768.0f : 1.0f;
        return c2{make_float2((float)(int16_t)(x & 0xFFFFu) * k, (float)(int16_t)(x >> 16) * k),
                  make_float2((float)(int16_t)(y & 0xFFFFu) * k, (float)(int16_t)(y >> 16) * k)};
    } else {
        return c2{make_float2(__uint_as_float(w[0]), __uint_as_float(w[1])),
                  make_float2(__uint_as_float(w[2]), __uint_as_float(w[3]))};
    }
}
// the integer formats' 2^-(bits-1) (a power of two: folding it into the window weight
// instead of the sample changes no bit of the product)
enum { PSDR_FMT_U8_ = 0, PSDR_FMT_U16_ = 2 };  // psdr_format (include/psdr.h)
template <int SB>
__device__ __forceinline__ constexpr float image_scale() {
    return SB == 2 ? 1.0f / 128.0f : (SB == 4 ? 1.0f / 32768.0f : 1.0f);
}

// pass 1: convert + window + column FFT (length L = M1) + inter-pass twiddle.
//   T columns per tile (T/2 couples), SB bytes per complex sample of the raw image
//   (u8/s8: 2, u16/s16: 4, f32 and f64-narrowed-to-f32: 8).  L*T/32 threads.
// Y layout.  Plain (IQ, small real transforms): one linear block per PASS-1 tile, [c1][T]; pass 2
// gathers 2 KiB pieces (16 rows x T) from the 64 blocks of a frame - efficient because the 64 pass-2
// work-groups of a frame read neighbouring pieces at about the same time.  PAIR (fused real input):
// PASS-2-TILE-MAJOR - a pass-1 tile contributes one (rows per pass-2 tile) x T chunk to every pass-2
// tile's block, so that pass 2 reads its tile as ONE contiguous 128 KiB block: its work-groups walk
// chains of tiles of different frames and have no neighbours in time (measured on cfg3: gathered
// reads 1092 us, linear reads 960 us per 256 frames; the scattered 2 KiB writes cost pass 1 28 us,
// which is why the IQ path, whose pass 2 gains only 5 us, keeps the linear pass-1 blocks).
// PAIR: real input feeding the fused pass 2 (k_fft_pass2_real).  The packed N/2-point transform Z is
// untangled into the real signal's spectrum from the pairs (Z[k], conj Z[M-k]); bin k = c1 + M1*c2
// pairs with row M1-c1, column M2-1-c2.  Two changes make that pairing thread-local in pass 2:
//   * rows are regrouped: pass-2 tile g holds rows 8g..8g+7 (slots 0..7) and their mirrors M1-8g-p
//     (slots 8+p; row M1/2 at slot 8 of tile 0, beside row 0): a pass-2 couple is a (row, mirror row)
//     pair, and eight consecutive rows stay 1 KiB contiguous for pass 1's stores;
//   * mirror rows (c1 > M1/2) are stored as conj(Y[c1][n2]) * W_M2^{n2}: the plain forward row
//     transform of that sequence is G[c2] = conj(Z[c1][M2-1-c2]), exactly the partner of the
//     couple's other half at the same output index c2.
// CP (PAIR): (row, mirror row) couples per pass-2 tile - 8 for 1024-point rows (tiles of 16 rows), 4 for 2048-point rows
// (tiles of 8 rows: 2^22-point real frames split 1024 x 2048)
template <int L, int T, int SB, bool PAIR = false, int CP = 8>
__device__ __forceinline__ void pass1_body(const Pass1Args &a, unsigned vb, unsigned vgrid) {
    static_assert(CP == 8 || CP == 4, "couples per pass-2 tile");
    constexpr int L2CP = CP == 8 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    cf *Wl = reinterpret_cast<cf *>(smem) + L * T;
    constexpr int H = T / 2;        // couples per row
    constexpr int NT = (L / 16) * H;
    constexpr int L16 = L / 16;
    constexpr int RL = LastStage<L>::R, PL = LastStage<L>::Pp, NBL = 16 / RL;
    // The raw samples go straight from HBM into the registers of the thread that converts them:
    // one load per row of the thread (16 rows x one couple = 2*SB bytes each: 4 / 8 / 16), eight
    // lanes = one 64-byte (s16) row segment, eight rows per wave instruction.  (Round 1 staged a
    // linear image of the tile through LDS with 16-byte loads: 8 more ds_write_b128, 16 more
    // ds_read_b64 and two more barriers per tile for the same lines touched.)
    constexpr int WPL = SB / 2;                             // 32-bit words per load
    constexpr int NCHK = 16;                                // loads per thread and tile
    constexpr int NTICK = 2 * (Plan<L>::NS - 1);            // ticks that carry loads
    constexpr int EARLY = NCHK / 4;                         // loads issued right after the image write
    constexpr int LPT = (NCHK - EARLY + NTICK - 1) / NTICK;
    const int tid = threadIdx.x;
    PSDR_WGTRACE(a.trace, 0);
    kclk_begin(a.kclk);
    const int p_ = tid % H, i0_ = tid / H;
    const int M2 = a.M2;
    const size_t M = (size_t)L << a.log2M2;
    const unsigned total = a.total_slots;
    const int fmt = a.fmt;

    // two-level W_M table with B = M2: W_M^e = W_M1^{e >> log2M2} * W_M^{e & (M2-1)}.  The
    // first factor IS the stage table Wl (in LDS); the second (M2 entries) is staged next
    // to it, so a per-tile twiddle look-up costs two LDS reads, no L2 round trip.
    cf *ldsTB = Wl + L;
    auto tw = [&](unsigned e) -> cf { return cmul(Wl[e >> a.log2M2], ldsTB[e & (unsigned)(M2 - 1)]); };
    auto tw2 = [&](unsigned eA, unsigned eB, cf &rA, cf &rB) {  // two look-ups, products interleaved
        cmul_pair(rA, Wl[eA >> a.log2M2], ldsTB[eA & (unsigned)(M2 - 1)], rB, Wl[eB >> a.log2M2],
                  ldsTB[eB & (unsigned)(M2 - 1)]);
    };

    // chunk i*NT + tid of the image = row (i*NT + tid)/LPR, byte (tid % LPR)*16 of that row.
    // global: frame f starts at complex sample f*M/2; row r is M2 samples further.
    const unsigned gsb = fmt == 5 ? 16u : (unsigned)SB;  // bytes per complex sample in HBM
    const size_t g_row = (size_t)M2 * gsb;
    const size_t g_step = (size_t)L16 * g_row;  // between a thread's consecutive rows
    const size_t g_lane = (size_t)i0_ * g_row + (size_t)(2 * p_) * gsb;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // plain vector: SROA keeps it in VGPRs
    unsigned rq[NCHK][WPL];
    const unsigned char *nxt = nullptr;
    auto point_at = [&](unsigned sidx) {
        const unsigned slot = xcd_slot(sidx, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        nxt = reinterpret_cast<const unsigned char *>(a.raw) + ((size_t)f * (M / 2) + (size_t)tl * T) * gsb + g_lane;
    };
    auto issue = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const unsigned char *q = nxt + (size_t)i * g_step;
        if constexpr (SB == 8) {
            if (fmt == 5) {  // f64: two samples = 32 bytes, narrowed to f32 here
                const double2 s0 = reinterpret_cast<const double2 *>(q)[0];
                const double2 s1 = reinterpret_cast<const double2 *>(q)[1];
                rq[i][0] = __float_as_uint((float)s0.x);
                rq[i][1] = __float_as_uint((float)s0.y);
                rq[i][2] = __float_as_uint((float)s1.x);
                rq[i][3] = __float_as_uint((float)s1.y);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(q);
                rq[i][0] = v.x, rq[i][1] = v.y, rq[i][2] = v.z, rq[i][3] = v.w;
            }
        } else if constexpr (SB == 4) {
            const uint2 v = *reinterpret_cast<const uint2 *>(q);
            rq[i][0] = v.x, rq[i][1] = v.y;
        } else {
            rq[i][0] = *reinterpret_cast<const unsigned *>(q);
        }
    };
    __shared__ unsigned s_next[4];  // [0..1]: TileQueue
    TileQueue tq;
    tq.init(a.tickets, total, false, 0, vb, vgrid);
    unsigned s = vb, snext = vb + vgrid;
    if (s < total) {
        point_at(s);
        static_for<0, NCHK>(issue);
    }
    // table staging after the first tile's loads are in flight (one latency, not two)
    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];
    for (int i = tid; i < M2; i += NT) ldsTB[i] = a.TB[i];
    tq.draw_first();
    __syncthreads();  // Wl and the twiddle table are visible
    PSDR_WGTRACE(a.trace, 1);
    // This thread's twiddles of the SECOND stage stay in registers for every tile of the persistent loop (StageTw):
    // same box, interleaved, cfg2, per 256 frames - pass 1 577 -> 552 us and the whole step -1.5 ... -2 %.  Keeping the
    // last stage's twelve as well (PSDR_TW_P1_MASK=3) makes pass 1 itself faster still (525 us) but the step no faster:
    // at 238 VGPRs the previous batch's consumer kernels no longer fit beside a pass-1 work-group and run later instead.
    constexpr int TWM = 1;  // bit 0: second stage, bit 1: last stage
    StageTw<L> stw;
    if (TWM) stw.load(Wl, i0_);
    const StageTw<L> *stw_front = (Plan<L>::NS == 3 && (TWM & 1)) ? &stw : nullptr, *stw_last = (TWM & 2) ? &stw : nullptr;

    // W_L^{i0}: the thread's part of every row's window angle, times the output scale (Pass1Args::yscale)
    const cf wl0 = make_float2(Wl[i0_].x * a.yscale, Wl[i0_].y * a.yscale);

    int it = 0;
    for (; s < total; it++) {
        PSDR_TRACE(a.trace, it, 0);
        const unsigned slot = xcd_slot(s, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        // this tile's block (plain) / its chunk of pass-2 tile 0 (PAIR: pass-2 tiles have 16 rows)
        cf *Yb = a.Y + (size_t)(f & a.ymask) * a.yframe + (PAIR ? (size_t)tl * a.ytl : (size_t)tl * a.yblk);
        // PAIR: where the lane's part of a row index puts it (see the store below)
        cf *Ylo = Yb + (size_t)(i0_ >> L2CP) * a.ytile + (i0_ & (CP - 1)) * T + 2 * p_;
        cf *Yhi = Yb - (size_t)((i0_ + CP - 1) >> L2CP) * a.ytile + (CP + ((-i0_) & (CP - 1))) * T + 2 * p_;
        (void)Ylo;
        (void)Yhi;
        const bool has_next = snext < total;
        if (has_next) point_at(snext);
        const bool more = has_next;  // (the loads below; nxt stays on this tile without a next one)
        // the index after `snext`: drawn one tile ago, published now, read after the barrier
        tq.draw_end(&s_next[it & 1], s);
        tq.draw_begin();
        // opaque per-iteration copies: stop LICM from hoisting the loop-invariant LDS
        // addresses of all stages out of the persistent loop (that costs >100 VGPRs)
        int i0 = i0_, p = p_, tidx = tid;
        asm volatile("" : "+v"(i0), "+v"(p), "+v"(tidx));

        (void)tidx;
        PSDR_TRACE(a.trace, it, 1);
        PSDR_TRACE(a.trace, it, 2);

        // ---- stage-0 input: convert (src/samplereader.cpp:29-40) + Hann window, from the
        // registers the prefetch left the raw words in.
        // periodic Hann (src/utils/dsp.cpp:6-11) from the twiddle tables:
        // exp(-i*2*pi*n/M) = W_M1^{n1} * W_M^{n2}
        const unsigned nA = tl * T + 2u * (unsigned)p, nB = nA + 1u;  // n2 of the two columns
        c2 u[16];
        // The thread's rows are i0 + e L/16: exp(-i 2 pi row / L) = W_L^{i0} W_16^e - a per-thread constant times a
        // sixteenth root of unity known at compile time.  With z0 = W_L^{i0} W_M^{n2} (one complex product per column
        // and tile) the Hann weight of row e is 0.5 - 0.5 Re(z0 W_16^e) = 0.5 - 0.5 (z0.x cos t_e + z0.y sin t_e),
        // t_e = 2 pi e / 16: two packed FMAs with literal coefficients per row and column couple, no table read per
        // row (round 3: a stage-table read, two packed products and two FMAs per row for IQ; four complex products
        // per row for real input, whose odd samples need a second angle)
        cf wbA, wbB;  // W_M^{n2}: window angle of the columns
        tw2(nA, nB, wbA, wbB);
        cf z0A, z0B;
        cmul_pair(z0A, wl0, wbA, z0B, wl0, wbB);
        constexpr float C16[16] = {1.f, PSDR_C1_16, PSDR_SQRT1_2, PSDR_S1_16, 0.f, -PSDR_S1_16, -PSDR_SQRT1_2, -PSDR_C1_16,
                                   -1.f, -PSDR_C1_16, -PSDR_SQRT1_2, -PSDR_S1_16, 0.f, PSDR_S1_16, PSDR_SQRT1_2, PSDR_C1_16};
        constexpr float S16[16] = {0.f, PSDR_S1_16, PSDR_SQRT1_2, PSDR_C1_16, 1.f, PSDR_C1_16, PSDR_SQRT1_2, PSDR_S1_16,
                                   0.f, -PSDR_S1_16, -PSDR_SQRT1_2, -PSDR_C1_16, -1.f, -PSDR_C1_16, -PSDR_SQRT1_2, -PSDR_S1_16};
        // the integer formats' 2^-(bits-1) rides in the weights (a power of two: no bit of the product changes); so does the
        // output scale: z0 carries it through wl0, the constant term through hs
        constexpr float hk = 0.5f * image_scale<SB>();
        const float hs = hk * a.yscale;
        const bool flip = SB <= 4 && (fmt == PSDR_FMT_U8_ || fmt == PSDR_FMT_U16_);  // uniform: one branch per tile
        auto fill = [&](auto flipc) {
            constexpr bool FLIP = decltype(flipc)::value;
            if (a.is_real) {
                // even samples: angle of z0; odd samples: one sample further, z0 W_N^1
                cf z1A, z1B;
                cmul_pair(z1A, z0A, a.wdelta, z1B, z0B, a.wdelta);
                const v2f zxA = {z0A.x, z1A.x}, zyA = {z0A.y, z1A.y}, zxB = {z0B.x, z1B.x}, zyB = {z0B.y, z1B.y};
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const c2 x = words_to_c2_fast<SB, FLIP>(rq[e]);
                    const v2f kc = {-hk * C16[e], -hk * C16[e]}, ks = {-hk * S16[e], -hk * S16[e]}, hf = {hs, hs};
                    const v2f wA = __builtin_elementwise_fma(zxA, kc, __builtin_elementwise_fma(zyA, ks, hf));
                    const v2f wB = __builtin_elementwise_fma(zxB, kc, __builtin_elementwise_fma(zyB, ks, hf));
                    u[e].a = from_v2f(to_v2f(x.a) * wA);
                    u[e].b = from_v2f(to_v2f(x.b) * wB);
                }
            } else {
                const v2f zx = {z0A.x, z0B.x}, zy = {z0A.y, z0B.y};
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const c2 x = words_to_c2_fast<SB, FLIP>(rq[e]);
                    const v2f kc = {-hk * C16[e], -hk * C16[e]}, ks = {-hk * S16[e], -hk * S16[e]}, hf = {hs, hs};
                    const v2f w2 = __builtin_elementwise_fma(zx, kc, __builtin_elementwise_fma(zy, ks, hf));
                    u[e].a = scale_lo(x.a, w2);
                    u[e].b = scale_hi(x.b, w2);
                }
            }
        };
        if (flip)
            fill(std::true_type{});
        else
            fill(std::false_type{});
        PSDR_SCHED_FENCE();
        if (more) static_for<0, EARLY>(issue);  // the raw registers are free again
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, it, 3);

        cf tbA[NBL], tbB[NBL], tsA[RL], tsB[RL], w00A, w00B;  // inter-pass twiddles
        cf cWA = make_float2(1.f, 0.f), cWB = cWA;             // PAIR: W_M2^{n2} of the two columns
        cf tmA[RL], tmB[RL];                                   // PAIR: conj(ts[q]) * W_M2^{n2}, q >= RL/2 (the mirror rows)
#ifdef PSDR_ABL_P1_NOX2  // (timing-only: the second stage without its LDS exchange - wrong results)
        constexpr bool kP1X2 = false;
#else
        constexpr bool kP1X2 = true;
#endif
        run_stages<L, T, false, kP1X2>(
            tile, Wl, i0, p, u,
            // ---- inter-pass twiddle W_M^{n2*kappa}, kappa = i0 + b*L/16 + s*P, as
            // base * stepB^b * stepS^s (three table look-ups per column); client order
            // carries (-1)^{n2}: nothing for the even column, a sign for the odd one
            [&]() {
                tw2(nA * (unsigned)i0, nB * (unsigned)i0, tbA[0], tbB[0]);
                if (a.rot) tbB[0] = make_float2(-tbB[0].x, -tbB[0].y);
                cf sbA, sbB, ssA, ssB;
                tw2(nA * (unsigned)L16, nB * (unsigned)L16, sbA, sbB);
                tw2(nA * (unsigned)PL, nB * (unsigned)PL, ssA, ssB);
#pragma unroll
                for (int b = 1; b < NBL; b++) {
                    cmul_pair(tbA[b], tbA[b - 1], sbA, tbB[b], tbB[b - 1], sbB);
                }
                tsA[0] = tsB[0] = make_float2(1.f, 0.f);
#pragma unroll
                for (int q = 1; q < RL; q++) {
                    if (q == 1) {
                        tsA[q] = ssA;
                        tsB[q] = ssB;
                    } else {
                        cmul_pair(tsA[q], tsA[q - 1], ssA, tsB[q], tsB[q - 1], ssB);
                    }
                }
                // output (b=0,s=0) of the thread with i0 = 0 is bin k1 = 0: in client order
                // it goes to row M1-1 with W_M^{n2*M1}
                w00A = tbA[0];
                w00B = tbB[0];
                if (a.rot && i0 == 0) {
                    tw2(nA * (unsigned)L, nB * (unsigned)L, w00A, w00B);
                    w00B = make_float2(-w00B.x, -w00B.y);
                }
                if constexpr (PAIR) {
                    tw2(nA * (unsigned)L, nB * (unsigned)L, cWA, cWB);  // W_M^{n2*M1}
#pragma unroll
                    for (int q = RL / 2; q < RL; q++) cmul_cja_pair(tmA[q], tsA[q], cWA, tmB[q], tsB[q], cWB);
                }
            },
            [&](int b, int sidx, int k1, c2 x) {
                cf wA, wB, yA, yB;
                const int c1 = a.rot ? ((k1 - 1) & (L - 1)) : k1;
                (void)c1;
                const bool mirror = PAIR && sidx >= RL / 2;                    // k1 >= L/2 (compile time): mirror form ...
                const bool maybe_natural = mirror && sidx == RL / 2 && b == 0;  // ... except row L/2 itself (i0 == 0)
                if (!mirror || maybe_natural) {
                    if (sidx == 0) {
                        wA = b == 0 ? w00A : tbA[b];
                        wB = b == 0 ? w00B : tbB[b];
                    } else {
                        cmul_pair(wA, tbA[b], tsA[sidx], wB, tbB[b], tsB[sidx]);
                    }
                    cmul_pair(yA, x.a, wA, yB, x.b, wB);
                }
                if constexpr (PAIR) {
                    if (mirror) {
                        // conj(x w) W_M2^{n2} = conj(x) (conj(tb) (conj(ts) W_M2^{n2})): the last factor once per tile (tm), two
                        // products per output instead of three (round 3 multiplied x w out and conjugated it)
                        cf wmA, wmB, mA, mB;
                        cmul_cja_pair(wmA, tbA[b], tmA[sidx], wmB, tbB[b], tmB[sidx]);
                        cmul_cja_pair(mA, x.a, wmA, mB, x.b, wmB);
                        const bool natural = maybe_natural && i0 == 0;  // k1 == L/2
                        yA = natural ? yA : mA;
                        yB = natural ? yB : mB;
                    }
                }
                cf *dst;
                if constexpr (PAIR) {
                    // row k1 < L/2 is row k1 & (CP-1) of pass-2 tile k1 / CP, its mirror L-k1 row CP + (k1 & (CP-1)) of the
                    // same tile.  k1 = i0 + (a multiple of L/16 known at compile time): the lane part lives
                    // in Ylo / Yhi, the rest is a compile-time multiple of the (uniform) tile stride
                    if (sidx < RL / 2) {
                        dst = Ylo + (size_t)((b * L16 + sidx * PL) >> L2CP) * a.ytile;
                    } else {
                        dst = Yhi + (size_t)((L - sidx * PL - b * L16) >> L2CP) * a.ytile;
                        if (sidx == RL / 2 && b == 0) dst = i0 == 0 ? Yb + CP * T + 2 * p : dst;  // row L/2: beside row 0
                    }
                } else {
                    dst = Yb + (size_t)c1 * T + 2 * p;
                }
                *reinterpret_cast<float4 *>(dst) = make_float4(yA.x, yA.y, yB.x, yB.y);
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
            [&](int k) { PSDR_TRACE(a.trace, it, k); }, stw_front, stw_last);
        PSDR_TRACE(a.trace, it, 10);
        PSDR_WGTRACE(a.trace, 2 + it);
        // (published by thread 0 before the stages' barriers)
        const unsigned s2 = s_next[it & 1];
        s = snext;
        snext = s2;
    }
    PSDR_WGTRACE(a.trace, 7);
    kclk_end(a.kclk);
}
template <int L, int T, int SB, bool PAIR = false, int CP = 8>
__global__ __launch_bounds__(L *T / 32) void k_fft_pass1(Pass1Args a) {
    pass1_body<L, T, SB, PAIR, CP>(a, blockIdx.x, gridDim.x);
}

struct Pass2Args {
    const cf *Y;   // k_fft_pass2: [nframes][tiles1][M1][TW]; k_fft_pass2_real: [nframes][pass-2 tile][tiles1][rows][TW]
    cf *X;         // [nframes][spec_stride]: bin c1 + M1*c2
    size_t spec_stride;
    const cf *Wl;  // W_L^j, L = M2
    int M1;
    int log2M1;
    int TW;       // pass-1 tile width (columns per block of Y)
    int log2TW;
    size_t yblk, ytile, yframe;  // pass-1 block (k_fft_pass2) / pass-2 tile block (k_fft_pass2_real) / frame strides, elements
    size_t yjs;                  // k_fft_pass2_real: elements between the chunks of consecutive pass-1 tiles
                                 // (tile-major: the chunk itself, 16 * TW: one linear block; blocked: M1 * TW)
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
    unsigned *tickets;  // TileQueue counters of this launch (8, zeroed)
    unsigned long long *trace;
    unsigned long long *kclk;  // device-clock stamps of this launch (kclk_begin / kclk_end) or nullptr
    unsigned ymask;            // as Pass1Args::ymask
    // BAND kernels: the spectrum goes out in band regions (SpecLayout mode 3, quantize.h): column c2 of tile tl is the
    // line ((c2 >> l2Lb) * band_stride) + (tl * Lw + (c2 & lbmask)) * 16 of the frame's slot in its band's region
    int l2Lb, lbmask, Lw;
    size_t band_stride;
    // fused real-input epilogue (k_fft_pass2_real)
    const cf *UA, *UB;  // W_N^{h << log2UB}, W_N^{l}: untangle twiddles
    const cf *UG;       // W_N^{8g}, g < M1/16: the tile's factor of the untangle twiddle
    int log2UB;
    // chain segments of the fused real pass (forward.hip, build_seg_table): entry sg = { frame, first (highest) tile | tiles
    // << 16, the segment ABOVE it in the frame (whose last tile carries into this one's first; the top segment's: the
    // frame's bottom segment, whose tile 0 holds row M1/2), flags }
    const uint4 *segtab;
    float *seamP;       // [seam segments][L][8]: partial octets of the first tile of a segment without a carry-in
    float *seamC;       // [segments][L]: carry-out of every segment's last tile
    unsigned *segflag;  // [segments]: == epoch once the segment's carry-out row is in memory (hand-off plans; else nullptr)
    unsigned *segmark;  // [segments]: == epoch if the segment's first tile did NOT find its carry-in in time and left its
                        // partial octets in seamP like a segment without one (k_real_seam completes them)
    unsigned epoch;     // of this launch (never 0)
};
enum { PSDR_SEG_CARRY_MEM = 1 };  // segtab flags: the first tile's carry-in comes from seamC[above] behind segflag[above]

// pass 2: row FFT (length L = M2) of T adjacent rows c1 (T/2 couples); FUSED adds /N,
// |X|^2, int8 level 0..LT of the pyramid.  L*T/32 threads.
// TWC: pass-1 tile width when known at compile time (all fill addresses fold), 0: a.TW
// BAND: banded spectrum layout (band sharding without a pack pass: every band's lines of a whole batch form one
// contiguous region, which is what is sent over the link)
template <int L, int T, bool FUSED, int TWC, bool BAND = false>
__device__ __forceinline__ void pass2_body(const Pass2Args &a, unsigned vb, unsigned vgrid) {
    static_assert(!BAND || (FUSED && L == 1024 && T == 16), "banded layout: the tile-major IQ spectrum only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    cf *tile_cf = reinterpret_cast<cf *>(smem);
    cf *Wl = tile_cf + L * T;
    constexpr int H = T / 2;
    constexpr int NT = (L / 16) * H;
    constexpr int NLD = 16;                          // 16-byte loads per thread and tile
    constexpr int NTICK = 2 * (Plan<L>::NS - 1);     // ticks that carry loads (not the last stage)
    constexpr int EARLY = 4;                         // loads issued right after the registers die
    // ... and LATE of them only in the epilogue (half before the last stage, half before the record loop):
    // with all sixteen issued during the front stages the read side of the memory system idles while the
    // tile's 150 KB of stores drain.  Same box, interleaved, 256 frames: 857 -> 833 us (LATE 8 or 10, EARLY
    // 4 or 2 alike); the fused real-input pass 1014 -> 970 us.  (Pass 1 gains nothing from it - its stores
    // are the whole last stage - and loses 35 us with eight loads that late.)
    constexpr int LATE = FUSED ? 8 : 0;
    constexpr int NFRONT = NLD - LATE;
    constexpr int LPT = (NFRONT - EARLY + NTICK - 1) / NTICK;
    const int tid = threadIdx.x;
    PSDR_WGTRACE(a.trace, 0);
    kclk_begin(a.kclk);
    const int p_ = tid % H, i0_ = tid / H;
    const int M1 = a.M1;
    const unsigned total = a.total_slots;
    const int TW = TWC ? TWC : a.TW;
    const int log2TW = TWC ? 31 - __builtin_clz((unsigned)(TWC ? TWC : 1)) : a.log2TW;
    const int chunk = T * TW;  // elements one pass-1 tile contributes to this tile

    // Element idx of the tile is (block j, row rr, column cc) with idx = j*chunk + rr*TW + cc:
    // Y block j, row c1base + rr, column cc, and n2 = j*TW + cc.  A thread loads 16 bytes =
    // elements idx, idx+1 with idx = 2*(i*NT + tid), i < 16: the address is a uniform per-i
    // part plus ONE per-lane offset.
    const int lc = log2TW + (31 - __builtin_clz((unsigned)T));  // log2(chunk)
    const size_t blk = a.yblk;
    float4 r[NLD];
    const cf *nxt = nullptr;
    const unsigned lane_off = (unsigned)((size_t)((2 * tid) >> lc) * blk + ((2 * tid) & (chunk - 1)));
    auto point_at = [&](unsigned sidx) {
        const unsigned slot = xcd_slot(sidx, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        nxt = a.Y + (size_t)(f & a.ymask) * a.yframe + (size_t)(tl * T) * TW;
    };
    // SPLIT (fused kernels): register i holds the tile's load number i ^ 8, so that the loads issued FIRST (i < 8) are
    // the upper half of the LDS tile (n2 >= L/2).  The powers the record loop reads (Pst) live in the lower half only,
    // so a wave that has finished its records fills the upper half of the NEXT tile at once; the "tile is free again"
    // barrier moves between the two half-fills, where the late loads (i >= 8: the lower half) have had longer to land.
    constexpr bool SPLIT = FUSED;
    auto issue = [&](auto qc) {
        constexpr int i = decltype(qc)::value;
        constexpr int ip = SPLIT ? (i ^ (NLD / 2)) : i;  // which sixteenth of the tile
        // uniform part of idx = 2*ip*NT: block (2*ip*NT)>>lc, offset (2*ip*NT)&(chunk-1)
        const cf *q = nxt + (size_t)((2 * ip * NT) >> lc) * blk + ((2 * ip * NT) & (chunk - 1)) + lane_off;
        r[i] = *reinterpret_cast<const float4 *>(q);
    };
    __shared__ unsigned s_next[4];  // [0..1]: TileQueue
    TileQueue tq;
    // pass 2 has no use for XCD affinity (full-line stores, tile-major records) and the XCDs
    // differ by ~10 % in speed: one chip-wide counter (pass 1 keeps the per-XCD queues: adjacent
    // tiles share the 128-byte lines of the raw rows)
    tq.init(a.tickets, total, true, 0, vb, vgrid);
    unsigned s = vb, snext = vb + vgrid;
    if (s < total) {
        point_at(s);
        static_for<0, NLD>(issue);
    }
    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];  // visible after the loop's first barrier
    // (register-resident stage twiddles, as in pass 1, measure nothing here - 866 vs 865 us with the second stage's fifteen
    // at 240 VGPRs, spills with the last stage's twelve on top: the stage table stays in LDS)
    const StageTw<L> *stw_front = nullptr, *stw_last = nullptr;
    tq.draw_first();
    PSDR_WGTRACE(a.trace, 1);

    int it = 0;
    for (; s < total; it++) {
        PSDR_TRACE(a.trace, it, 0);
        const unsigned slot = xcd_slot(s, total);
        const unsigned f = slot / a.tiles_per_frame;
        const unsigned tl = slot - f * a.tiles_per_frame;
        const int c1base = tl * T;
        const bool has_next = snext < total;
        if (has_next) point_at(snext);
        const bool more = has_next;  // (the loads below; nxt stays on this tile without a next one)
        // the index after `snext`: drawn one tile ago, published now, read after the barrier
        tq.draw_end(&s_next[it & 1], s);
        tq.draw_begin();
        int i0 = i0_, p = p_, tidx = tid;  // opaque copies (see pass 1)
        asm volatile("" : "+v"(i0), "+v"(p), "+v"(tidx));
        // transposing fill: point (n2, row rr) goes to half rr&1 of couple rr>>1
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int ip = SPLIT ? (i ^ (NLD / 2)) : i;
            const int w = ((2 * ip * NT) & (chunk - 1)) + ((2 * tidx) & (chunk - 1));
            const int rr = w >> log2TW, cc = w & (TW - 1);
            const int n2 = (((2 * ip * NT) >> lc) + ((2 * tidx) >> lc)) * TW + cc;  // even
            const int slot0 = lds_slot<H, true>(n2, rr >> 1);  // rows n2, n2+1 share the swizzle
            tile_cf[2 * slot0 + (rr & 1)] = make_float2(r[i].x, r[i].y);
            tile_cf[2 * (slot0 + H) + (rr & 1)] = make_float2(r[i].z, r[i].w);
            if (SPLIT && i == NLD / 2 - 1) {
                PSDR_SCHED_FENCE();
                __syncthreads();  // every wave has finished the previous tile's record loop: the lower half is free
                PSDR_SCHED_FENCE();
            }
        }
        PSDR_SCHED_FENCE();
        if (more) static_for<0, EARLY>(issue);
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, it, 1);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 2);
        // stage-0 input comes from the tile itself: all reads, then a barrier, before any
        // in-place write
        c2 u[16];
        tile_read<L, H, true>(u, tile, i0, p);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 3);
        const unsigned s2 = s_next[it & 1];

        cf *Xf = a.X + (size_t)f * a.spec_stride;
        cf *Xt = Xf + (size_t)tl * (L * T);  // the tile's block of a tile-major frame
        (void)Xt;
        float *Pst = reinterpret_cast<float *>(smem);  // reuses the (dead) tile after the last read
        run_stages<L, T, true>(
            tile, Wl, i0, p, u,
            [&]() {
                if (LATE > 0 && more) static_for<NFRONT, NFRONT + LATE / 2>(issue);
            },
            [&](int bb, int ss, int c2i, c2 x) {
                (void)bb, (void)ss;
                if (FUSED) {
                    // (/N, src/fft_impl.cpp:29-31, came in with the first pass's window weights: Pass1Args::yscale)
                    // src/fft_impl.cpp:36-38
                    *reinterpret_cast<float2 *>(Pst + c2i * T + 2 * p) =
                        make_float2(fmaf(x.a.x, x.a.x, x.a.y * x.a.y), fmaf(x.b.x, x.b.x, x.b.y * x.b.y));
                }
                // FUSED (IQ) tiles of 1024-point rows: tile-major lines (SpecLayout mode 1, quantize.h)
                if constexpr (BAND) {
                    // c2i = i0 + tcol with i0 < 64 <= band width: band and column inside the band follow from tcol
                    // alone (uniform, known after unrolling)
                    const int tcol = (L / 16) * (bb + (16 / LastStage<L>::R) * ss);
                    cf *Xb = Xf + (size_t)(tcol >> a.l2Lb) * a.band_stride + ((size_t)tl * a.Lw + (tcol & a.lbmask)) * T;
                    *reinterpret_cast<float4 *>(Xb + i0 * T + 2 * p) = pack_c2(x);
                } else if constexpr (FUSED && L == 1024 && T == 16)
                    *reinterpret_cast<float4 *>(Xt + c2i * T + 2 * p) = pack_c2(x);
                else
                    *reinterpret_cast<float4 *>(Xf + ((size_t)c2i << a.log2M1) + c1base + 2 * p) = pack_c2(x);
            },
            [&](int k) {  // trickle the rest of the next tile's loads through the stages
                if (more)
                    static_switch<0, NTICK>(k, [&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        constexpr int lo = EARLY + K * LPT < NFRONT ? EARLY + K * LPT : NFRONT;
                        constexpr int hi = lo + LPT < NFRONT ? lo + LPT : NFRONT;
                        static_for<lo, hi>(issue);
                    });
            },
            [&](int k) { PSDR_TRACE(a.trace, it, k); }, stw_front, stw_last);
        PSDR_TRACE(a.trace, it, 10);

        if (FUSED) {
            __syncthreads();
            if (LATE > 0 && more) static_for<NFRONT + LATE / 2, NLD>(issue);
            PSDR_TRACE(a.trace, it, 11);
            constexpr int CH = T < 16 ? T : 16;  // values per chunk (one aligned group)
            constexpr int NG = (L * T / CH) / NT;  // chunks per thread; chunks tile Pst linearly
            static_assert(CH == 16 || CH == 8, "tile width");
            int8_t *Qf = a.Qt + (size_t)f * a.qt_stride;
            float *Pf = a.Pscr + (size_t)f * a.p_stride;
            constexpr int RGRP = 1;  // chunks whose LDS reads are issued together (2 measured nothing here; the real pass's octet loop gains from it)
#pragma unroll
            for (int k0 = 0; k0 < NG; k0 += RGRP) {
                float4 q4[RGRP][CH / 4];
#pragma unroll
                for (int j = 0; j < RGRP; j++)
#pragma unroll
                    for (int v4 = 0; v4 < CH / 4; v4++)
                        q4[j][v4] = reinterpret_cast<const float4 *>(Pst)[(((k0 + j) * NT + tidx) * CH) / 4 + v4];
#pragma unroll
                for (int j = 0; j < RGRP; j++) {
                    const int g = (k0 + j) * NT + tidx;  // chunk id
                    float pw[CH];
#pragma unroll
                    for (int v4 = 0; v4 < CH / 4; v4++) {
                        pw[4 * v4] = q4[j][v4].x;
                        pw[4 * v4 + 1] = q4[j][v4].y;
                        pw[4 * v4 + 2] = q4[j][v4].z;
                        pw[4 * v4 + 3] = q4[j][v4].w;
                    }
                    // levels 0..LT of this aligned group -> one record; tile-major record order
                    // (RecMap, quantize.h): chunk g of tile tl is record tl*(L*T/CH) + g
                    const size_t rp = (size_t)tl * (L * T / CH) + g;
                    uint4 *rec = reinterpret_cast<uint4 *>(Qf + rp * (2 * CH));
                    if constexpr (CH == 16) {
                        uint4 lo, hi;
                        pyr_record16(pw, a.size_log2, lo, hi);
                        rec[0] = lo;
                        rec[1] = hi;
                    } else {
                        uint4 r8;
                        pyr_record8(pw, a.size_log2, r8);
                        rec[0] = r8;
                    }
                    Pf[rp] = pw[0];
                }
                PSDR_SCHED_FENCE();
            }
        }
        PSDR_TRACE(a.trace, it, 12);
        if (!SPLIT) __syncthreads();  // the tile is free again (SPLIT: between the next tile's two half-fills)
        PSDR_TRACE(a.trace, it, 13);
        PSDR_WGTRACE(a.trace, 2 + it);
        s = snext;
        snext = s2;
    }
    PSDR_WGTRACE(a.trace, 7);
    kclk_end(a.kclk);
}

template <int L, int T, bool FUSED, int TWC, bool BAND = false>
__global__ __launch_bounds__(L *T / 32) void k_fft_pass2(Pass2Args a) {
    pass2_body<L, T, FUSED, TWC, BAND>(a, blockIdx.x, gridDim.x);
}

// ---- pass 2 for REAL input, fused with the Hermitian untangle, /N, |X|^2 and pyramid levels 0..3 ----
// (replaces cufftExecR2C / fftwf r2c + power_and_quantize + the first half_and_quantize calls,
//  src/fft_impl.cpp:104-117,144-172; round 1 ran a third full pass over the spectrum for this)
//
// Pass 1 (PAIR) left the rows of Y as (row c1, mirror row M1-c1) couples, mirror rows pre-conjugated,
// so after the row transform a thread holds, for each of its 16 output indices c2,
//     a = Z[c1 + M1*c2]            b = conj(Z[M - (c1 + M1*c2)])
// and both real-signal bins of the pair come out of its own registers:
//     X[k]   = (a+b)/2 + W_N^k * (-i)(a-b)/2          X[M-k] = conj((a+b)/2 - W_N^k * (-i)(a-b)/2)
// Tile g = couples (8g+p, M1-8g-p), p = 0..7.  The self-paired rows 0 and M1/2 share couple 0 of
// tile 0 (their partners sit in OTHER threads' registers: one LDS exchange, that tile only).
//
// Output of tile g at column c2: rows 8g..8g+7 (from the couples' first halves) and, at column
// L-1-c2, rows M1-8g-7..M1-8g (second halves).  In the reference's k order these are two 64-byte
// pieces 8 KiB away from the next column's - and the neighbouring rows belong to tiles another
// work-group writes at an unrelated time: stored like that, the spectrum stores cost more than the
// whole rest of the tile (measured 1035 of 1940 us per 256 frames, with 8- or 16-byte stores alike).
// So the frame is laid out in HBM as 128-byte lines [ low octet of column c | mirror octet of tile g
// of column L-1-c ], tile-major (SpecLayout, quantize.h): the lane pair (p, p^1) swaps one bin per
// output (DPP), even lanes store the low-side pairs, odd lanes the mirror-side pairs, ONE
// 16-byte-per-lane store instruction writes eight adjacent whole lines, and a tile's output is one
// contiguous 128 KiB block.  Consumers index through SpecLayout::pos (demodulation slices) or ask
// for k order (psdr_read_spectrum).
//
// Pyramid: octets in TRUE k order.  The low octet [8g, 8g+8) is complete in the tile.  The high
// octet [M1-8g-8, M1-8g) has seven rows here and its first row (M1-8g-8) in tile g+1 (couple 0), so
// a work-group walks a SEGMENT of consecutive tiles of one frame (Pass2Args::segtab) in DEcreasing g and
// carries that one row of powers (4 KiB of LDS) to the next tile.  The first tile of a segment has no
// carry-in in LDS: either it fetches the row the segment above left in seamC (hand-off plans, see the
// kernel) or it leaves the seven partial rows in seamP; the last tile leaves its carry-out in seamC;
// k_real_seam (epilogue.h) completes the octets left open (tile 0's carry-out is row M1/2, which closes
// the ring at octet [M1/2, M1/2+8) of the LAST tile).  Records: [tile g][c2][low, high] x 16 bytes
// (RecMap mode 2), level-3 sums in the same order for the tail kernel.
// W_32^t = exp(-2 pi i t/32), t < 16: a thread's 16 outputs are c2 = i0 + (L/16)*t, and
// W_N^{M1*c2} = W_{2L}^{c2} = W_{2L}^{i0} * W_32^t
__device__ __forceinline__ cf w32(int t) {
    constexpr float C[16] = {1.f,
                             0.98078528040323044913f,
                             0.92387953251128675613f,
                             0.83146961230254523708f,
                             0.70710678118654752440f,
                             0.55557023301960222474f,
                             0.38268343236508977173f,
                             0.19509032201612826785f,
                             0.f,
                             -0.19509032201612826785f,
                             -0.38268343236508977173f,
                             -0.55557023301960222474f,
                             -0.70710678118654752440f,
                             -0.83146961230254523708f,
                             -0.92387953251128675613f,
                             -0.98078528040323044913f};
    // sin(2 pi t/32) = C[(t + 24) & 31] for the full circle; t < 16: sin >= 0 = C[|8 - t|]
    const int u = t <= 8 ? 8 - t : t - 8;
    return make_float2(C[t], -C[u]);
}
// the two bins of a pair from a = Z[k], b = conj(Z[M-k]); w = W_N^k.  The 0.5/N of the reference's untangle and
// normalisation came in with the first pass's window weights (Pass1Args::yscale: a power of two, no bit changes).
// xm is X[M-k].
__device__ __forceinline__ void untangle_pair(cf a, cf b, cf w, cf &xk, cf &xm) {
    const cf s = cadd(a, b), d = csub(a, b);
    const cf wo = cmul(w, make_float2(d.y, -d.x));  // W_N^k * (-i)(a-b)
    xk = make_float2(s.x + wo.x, s.y + wo.y);
    xm = make_float2(s.x - wo.x, wo.y - s.y);
}
// |x|^2 as the reference rounds it (src/fft_impl.cpp:36-38: fma(re, re, im*im)).  Two plain VALU
// instructions: left to itself the compiler packs two of these into v_pk_mul/v_pk_fma and pays four
// v_mov to gather the operands (a packed f32 op costs two plain ones: nothing gained, moves lost).
__device__ __forceinline__ float bin_power(cf x) {
    float t, p;
    asm("v_mul_f32 %0, %1, %1" : "=v"(t) : "v"(x.y));
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(p) : "v"(x.x), "v"(t));
    return p;
}

// Octet staging Pst[c2][16] floats (one 64-byte row per column: low octet, then elements 1..7 of the high octet and the
// tile's carry-out).  Side-major records: the octet loop reads ONE 16-byte quarter per lane with lanes = adjacent
// rows, 64 bytes apart - a four-way bank conflict in the plain order - so quarter j of row c sits at j ^ ((c >> 2) & 3).
// (CP = 4 couples per tile - 2048-point rows: a row is 8 floats, [low quartet | elements 1..3 of the high quartet, carry-out],
// one 16-byte quarter per side; lanes = adjacent rows read the SAME quarter 32 bytes apart: two-way, left as it is)
template <int CP = 8>
__device__ __forceinline__ int pst_at(int row, int e) {
    if constexpr (CP == 4) return row * 8 + e;
    return row * 16 + ((((e >> 2) ^ (row >> 2)) & 3) << 2) + (e & 3);
}

template <int L, int T, int TWC>
__global__ __launch_bounds__(L *T / 32) void k_fft_pass2_real(Pass2Args a) {
    static_assert((T == 16 && L == 1024) || (T == 8 && L == 2048), "eight (row, mirror) couples per tile of 1024-point rows, four of 2048-point rows");
    constexpr int CP = T / 2, L2CP = CP == 8 ? 3 : 2, LINE = 2 * CP;  // couples per tile; bins per spectrum line = floats per staging row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    cf *tile_cf = reinterpret_cast<cf *>(smem);
    cf *Wl = tile_cf + L * T;
    // the carried row: double-buffered where LDS has room; 2048-point rows (tile 128 KiB + table 16 KiB) keep ONE buffer - the
    // thread that reads entry c in the group loop is the thread that then writes entry c, and the row that arrives through
    // memory is written before the last stage, behind the barriers that follow the previous tile's group loop
    constexpr int NCARRY = L == 2048 ? 1 : 2;
    float *carry = reinterpret_cast<float *>(Wl + L);  // [NCARRY][L]
    constexpr int H = T / 2;
    constexpr int NT = (L / 16) * H;
    constexpr int L16 = L / 16;
    constexpr int NLD = 16;
    constexpr int NTICK = 2 * (Plan<L>::NS - 1);
    constexpr int EARLY = 4;
    PSDR_WGTRACE(a.trace, 0);
    kclk_begin(a.kclk);
    constexpr int LATE = 8;  // loads of the next tile issued in the epilogue (half before the last stage, half before the octet loop; see pass2_body)
    constexpr int NFRONT = NLD - LATE;
    constexpr int LPT = (NFRONT - EARLY + NTICK - 1) / NTICK;
    constexpr int NBL = 16 / LastStage<L>::R;
    const int tid = threadIdx.x;
    const int p_ = tid % H, i0_ = tid / H;
    const int M1 = a.M1;
    const unsigned total = a.total_slots;  // segments
    const int TW = TWC;
    constexpr int log2TW = TWC == 16 ? 4 : 3;
    static_assert(TWC == 16 || TWC == 8, "pass-1 tile width");
    const int chunk = T * TW;
    const int lc = log2TW + (T == 16 ? 4 : 3);  // log2(chunk)
    float4 r[NLD];
    const cf *nxt = nullptr;
    // element idx of the tile = (pass-1 tile j, row rr, column cc), idx = j*chunk + rr*TW + cc: chunk j of
    // pass-2 tile g starts at g*ytile + j*yjs (tile-major Y: yjs = chunk, the tile is one linear block)
    const unsigned lane_off = (unsigned)((size_t)((2 * tid) >> lc) * a.yjs + ((2 * tid) & (chunk - 1)));
    // tile j of segment sg: frame and first tile from the segment table, g = first - j (the segment index is
    // wave-uniform: the entry comes through the scalar cache, the tile's coordinates and addresses stay in scalar registers)
    typedef unsigned seg_u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) seg_u32x4 seg_entry_t;
    auto seg_entry = [&](unsigned sg) -> seg_u32x4 { return ((seg_entry_t *)a.segtab)[__builtin_amdgcn_readfirstlane(sg)]; };
    auto point_at = [&](seg_u32x4 e, int j) {
        const int g = (int)(e.y & 0xFFFFu) - j;
        nxt = a.Y + (size_t)(e.x & a.ymask) * a.yframe + (size_t)g * a.ytile + lane_off;
    };
    // SPLIT (as in pass2_body): register i holds the tile's load number i ^ 8, so that the loads issued FIRST (i < 8) are
    // the upper half of the LDS tile (n2 >= L/2).  The octet loop reads the staging rows (the lower half) and the carried
    // row only: a wave that has finished its octets fills the upper half of the NEXT tile at once, and the "tile is free
    // again" barrier sits between the two half-fills instead of at the end of the loop, where the waves of a work-group
    // arrived up to 2200 cycles apart (tools/trace_real.py, round 4: 9 % of a tile).
    // Same box, interleaved, F = 512: cfg3 (TWC = 16) 180.3 -> 183.4 GS/s; cfg5's share (TWC = 8: 1 KiB chunks of Y) 167.7 ->
    // 165.5 - there the end barrier costs less than the permuted load order, so it keeps the plain order.
    constexpr bool SPLIT = TWC == 16;
    auto issue = [&](auto qc) {
        constexpr int i = decltype(qc)::value;
        constexpr int ip = SPLIT ? (i ^ (NLD / 2)) : i;  // which sixteenth of the tile
        const cf *q = nxt + (size_t)((2 * ip * NT) >> lc) * a.yjs + ((2 * ip * NT) & (chunk - 1));
        r[i] = *reinterpret_cast<const float4 *>(q);
    };
    __shared__ unsigned s_next[4];  // [0..1]: the index after snext (TileQueue); [2]: the carry-in of this segment is in memory
    TileQueue tq;
    tq.init(a.tickets, total, true);
    unsigned s = blockIdx.x, snext = blockIdx.x + gridDim.x;
    // The table entries of the current and of the next segment live in scalar registers; the entry of a segment is
    // fetched when its index becomes known, a whole segment before it is used (a first touch misses the scalar cache
    // AND the L2: fetched where they are needed, two entries cost a work-group ~5 k cycles per segment)
    seg_u32x4 se = {0u, 0u, 0u, 0u}, sen = se;
    if (s < total) {
        se = seg_entry(s);
        if (snext < total) sen = seg_entry(snext);
        point_at(se, 0);
        static_for<0, NLD>(issue);
    }
    for (int i = tid; i < L; i += NT) Wl[i] = a.Wl[i];  // visible after the loop's first barrier
    tq.draw_first();
    // Hand-off of the carried row between chain segments through memory (Pass2Args::segflag != nullptr).
    // Producer: the LAST tile of a segment stores its carry-out row write-through (sc1: the reader sits on another XCD),
    // every wave waits for its own stores with a COUNTED wait two stages into the next tile (tick 2: by then they are
    // ~10 k cycles old and only the next tile's seven youngest loads are allowed to be outstanding), the barrier of that
    // stage follows, thread 0 publishes the launch's epoch in segflag[segment] (tick 3).
    // Consumer: thread 0 reads the flag of the segment above a tile EARLY (at the top of the previous segment's last tile,
    // or before the loop) and says at tick 0 of the segment's first tile - through s_next[2] - whether it was up.  If so,
    // every thread fetches its piece of the row with sc1 loads at tick 1 (behind the observed flag) and drops it into the
    // carry buffer before the last stage; the tile is then a tile like any other.  If not, NOBODY WAITS: the tile leaves
    // its partial octets in seamP as a segment without a carry-in does and marks itself in segmark, k_real_seam completes
    // them after the launch.  No spinning means no assumption about which work-groups are resident (two contexts on one
    // device, static first tiles) and no cost when the timing of a launch's first segments is off; with the level-major
    // ticket order of build_seg_table() the predecessor finished a round of segments earlier and the fallback is rare.
    // (cdna_hip_programming.md: in-launch hand-off, sc1 form - relaxed agent-scope accesses, one counted drain per
    // publication, never a fence per tile.)
    constexpr unsigned NOSEG = 0xFFFFFFFFu;
    unsigned post_seg = NOSEG;  // segment whose carry-out row is stored but not published yet
    // The flag and the row are CONDITIONAL loads whose results are used in other conditional blocks: at the joins the
    // compiler has to assume them pending, and wherever it then re-used their registers it put a conservative wait on the
    // path EVERY tile takes (vmcnt(4) before the last stage: a drain of the next tile's loads per tile).  They live in
    // registers of their own for the whole loop instead (never re-initialised, a dummy use at the loop's end, where
    // thirty younger operations make any such wait a formality).
    unsigned fl = 0;                 // thread 0: flag of the segment above the NEXT segment to start
    constexpr int NCIN = L / (2 * NT);   // 8-byte pieces of the carried row per thread (1; 2048-point rows: 2)
    static_assert(NCIN == 1 || NCIN == 2, "carried row: L floats over NT threads");
    unsigned long long cin_mem = 0, cin_mem2 = 0;  // carry-in row, floats 2 tid and 2 tid + 1 (and 2 (tid + NT), + 1)
    if (a.segflag && tid == 0 && s < total && (se.w & PSDR_SEG_CARRY_MEM))
        fl = __hip_atomic_load(a.segflag + se.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const float unscale = 2.0f / a.inv_n;  // 1 / Pass1Args::yscale: bin N/2 is never normalised by the reference
    const unsigned ubm = (1u << a.log2UB) - 1u;
    // Untangle twiddle of the thread's first output: W_N^{c1 + M1*i0}, c1 = 8g + p.  The (p, i0) part
    // is a per-thread constant; the tile's part W_N^{8g} is wave-uniform and comes through the
    // SCALAR cache: a vector load here would sit behind the previous tile's ~40 stores in the
    // in-order vmcnt queue and waiting for it drains them all, once per tile.
    // Lane roles of the pair exchange below: even lanes keep the low-side bin X[k] ("mine") and pass
    // the mirror-side bin X[M-k] on ("send"), odd lanes the other way round.  With sigma = +1 / -1:
    //   mine = (s + sigma*wo) * (1, sigma)      send = (s - sigma*wo) * (1, -sigma)
    // (s = a+b, wo = W_N^k * (-i)(a-b); the signed second component is the conjugation of X[M-k]),
    // so the roles cost no select: sigma rides in the thread's twiddle constant and in two per-thread sign
    // masks for the imaginary parts (one v_xor each; rounds 2-4 multiplied by (h, +-h) pairs here - the
    // 0.5/N now comes in with the first pass's window weights, Pass1Args::yscale).
    const bool ev_ = (p_ & 1) == 0;
    cf wc;
    {
        const unsigned e = (unsigned)p_ + (unsigned)M1 * (unsigned)i0_;
        wc = cmul(a.UA[e >> a.log2UB], a.UB[e & ubm]);
        if (!ev_) wc = make_float2(-wc.x, -wc.y);
    }
    const unsigned sgn_mine = ev_ ? 0u : 0x80000000u, sgn_send = ev_ ? 0x80000000u : 0u;
    const int off_ = ev_ ? (p_ & ~1) : (LINE - 2) - (p_ & ~1);  // bin of the lane's pair inside the line of LINE bins
    const int xflip_ = ev_ ? 0 : L - 1;                 // mirror-side values of column c belong to column L-1-c
    int j = 0, segit = 0;
    for (int it = 0; s < total; it++) {
        const unsigned f = se.x;
        const int g = (int)(se.y & 0xFFFFu) - j;
        PSDR_TRACE(a.trace, it, 0);
        const bool seg_first = j == 0, seg_last = j == (int)(se.y >> 16) - 1;
        const bool carry_mem = seg_first && (se.w & PSDR_SEG_CARRY_MEM) != 0;  // carry-in through memory, if it is there (uniform)
        // The next tile's loads are issued UNCONDITIONALLY: a work-group's very last tile fetches its own block once more
        // (nxt stays where it is; 128 KiB per work-group and launch).  With the loads under `if (more)` every wait the
        // compiler places behind them inside the same tile - the flag and the carried row of the hand-off below - has a
        // path with nothing younger in flight and degenerates to vmcnt(0).
        if (!seg_last)
            point_at(se, j + 1);
        else if (snext < total)
            point_at(sen, 0);
        if (seg_first) {
            tq.draw_end(&s_next[segit & 1], s);
            tq.draw_begin();
        }
        int i0 = i0_, p = p_, tidx = tid;  // opaque copies (see pass 1)
        asm volatile("" : "+v"(i0), "+v"(p), "+v"(tidx));
        // (constant address space: the only way to get s_load here - as a plain global pointer the
        // compiler picks a vector load, and since nothing is issued behind it when there is no next
        // tile to prefetch, its wait degenerates to vmcnt(0) = "all of the next tile's loads are in")
        typedef const __attribute__((address_space(4))) v2f ccf;
        const v2f wgv = ((ccf *)a.UG)[__builtin_amdgcn_readfirstlane(g)];
        const cf wg = make_float2(wgv.x, wgv.y);
        // transposing fill (as pass2_body)
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int ip = SPLIT ? (i ^ (NLD / 2)) : i;
            const int w = ((2 * ip * NT) & (chunk - 1)) + ((2 * tidx) & (chunk - 1));
            const int rr = w >> log2TW, cc = w & (TW - 1);
            const int n2 = (((2 * ip * NT) >> lc) + ((2 * tidx) >> lc)) * TW + cc;  // even
            const int slot0 = lds_slot<H, true>(n2, rr & (CP - 1));  // couple p = (row p, row CP + p of the tile)
            tile_cf[2 * slot0 + (rr >> L2CP)] = make_float2(r[i].x, r[i].y);
            tile_cf[2 * (slot0 + H) + (rr >> L2CP)] = make_float2(r[i].z, r[i].w);
            if (SPLIT && i == NLD / 2 - 1) {
                PSDR_SCHED_FENCE();
                __syncthreads();  // every wave has finished the previous tile's octet loop: the lower half is free
                PSDR_SCHED_FENCE();
            }
        }
        PSDR_SCHED_FENCE();
        asm volatile("" ::"v"(fl), "v"(cin_mem), "v"(cin_mem2));  // (see their declaration; behind the fill's own waits)
        static_for<0, EARLY>(issue);
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, it, 1);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 2);
        c2 u[16];
        tile_read<L, H, true>(u, tile, i0, p);
        __syncthreads();
        PSDR_TRACE(a.trace, it, 3);

        cf *Xf = a.X + (size_t)f * a.spec_stride;
        float *Pst = reinterpret_cast<float *>(smem);             // [L][LINE]: low octet, high octet
        cf *exA = tile_cf + (size_t)L * CP, *exB = exA + L;       // tile 0: rows 0 and M1/2 (beyond Pst)
        float *carry_w = carry + (NCARRY == 2 ? (it & 1) * L : 0), *carry_r = carry + (NCARRY == 2 ? ((it & 1) ^ 1) * L : 0);
        float *seamC = a.seamC + (size_t)s * L;
        bool cin_ok = false;  // the carried row is in memory and was fetched at tick 1 (uniform: thread 0 looked at the flag)
        // The counted drain at tick 2 (hand-off of the carried row) is correct ONLY IF every wave has issued exactly
        // EARLY + 3 * LPT vector-memory loads since its carry-out stores and nothing else in between: `issue` must stay
        // UNCONDITIONAL in this kernel (a load under a condition, or one the compiler can drop, makes the wait too weak and
        // the flag could be published before the row is acknowledged), the ticks must carry LPT loads each, and loads and
        // stores retire in order (one vmcnt counter).  PSDR_HANDOFF_FULL_DRAIN=1 (tuning builds) waits with vmcnt(0)
        // instead: the A/B that the bit-identity tests run against.
        static_assert(NTICK == 4 && LPT == 1, "the counted drain below knows what the ticks issue");
        static_assert(NFRONT - EARLY == NTICK * LPT && EARLY + 3 * LPT == EARLY + 3, "tick k issues load EARLY + k: three of them by tick 2");
        static_assert(EARLY + 3 * LPT <= 63, "vmcnt is a 6-bit field");
        run_front_stages<L, T, true>(
            tile, Wl, i0, p, u,
            [&](int k) {
                static_switch<0, NTICK>(k, [&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    constexpr int lo = EARLY + K * LPT < NFRONT ? EARLY + K * LPT : NFRONT;
                    constexpr int hi = lo + LPT < NFRONT ? lo + LPT : NFRONT;
                    static_for<lo, hi>(issue);
                });
                if (k == 2 && post_seg != NOSEG) {
                    // this wave's carry-out stores of the previous tile: older than the EARLY + 3 loads of the next tile
                    // issued since (and than the flag load and wave 0's ticket, if any: then the wait is only stricter)
#ifdef PSDR_HANDOFF_FULL_DRAIN
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EARLY + 3 * LPT) : "memory");
#endif
                }
                if (k == 0 && tid == 0) {
                    if (carry_mem) {
                        // (a tile old: issued at tick 0 of the previous segment's last tile, or before the loop)
                        asm volatile("" : "+v"(fl));  // (not to be speculated onto the path of the other tiles, wait included)
                        bool up = fl == a.epoch;
                        if (!up) {
                            // a tile ago it was not: look again, synchronously (the short segments at the end of a frame
                            // follow their predecessors by a tile or two; ~3 k cycles of wave 0 here against a seam)
                            const unsigned f2 = __hip_atomic_load(a.segflag + se.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            up = f2 == a.epoch;
                        }
                        s_next[2] = up;
                    }
                    // ... and the flag the NEXT segment will ask about
                    if (seg_last && a.segflag && snext < total && (sen.w & PSDR_SEG_CARRY_MEM))
                        fl = __hip_atomic_load(a.segflag + sen.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (k == 1 && carry_mem) {  // behind stage 0's barriers
                    cin_ok = s_next[2] != 0;
                    if (cin_ok) {  // sc1: from memory, not from this XCD's L2; a stage and a half before it is needed
                        cin_mem = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.seamC + (size_t)se.z * L) + tid,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if constexpr (NCIN == 2)
                            cin_mem2 = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.seamC + (size_t)se.z * L) + NT + tid,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (k == 3 && post_seg != NOSEG) {  // behind stage 1's barriers: every wave has drained
                    if (tid == 0) __hip_atomic_store(a.segflag + post_seg, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    post_seg = NOSEG;
                }
            },
            [&](int k) { PSDR_TRACE(a.trace, it, k); });
        if (LATE > 0) static_for<NFRONT, NFRONT + LATE / 2>(issue);
        // the carried row from memory goes into the carry buffer BEFORE the last stage's stores: behind them the compiler's
        // wait for it would be a wait for their acknowledgements (the buffer was last read in the previous tile's octet
        // loop, and every wave is past that: the barriers of this tile)
        if (cin_ok) {
            reinterpret_cast<float2 *>(carry_r)[tid] =
                make_float2(__uint_as_float((unsigned)cin_mem), __uint_as_float((unsigned)(cin_mem >> 32)));
            if constexpr (NCIN == 2)
                reinterpret_cast<float2 *>(carry_r)[NT + tid] =
                    make_float2(__uint_as_float((unsigned)cin_mem2), __uint_as_float((unsigned)(cin_mem2 >> 32)));
        }
        // no carry-in (the table says so, or the segment above was not published in time): k_real_seam completes the octets
        const bool seam_in = seg_first && !cin_ok;
        if (seam_in && carry_mem && tid == 0) a.segmark[s] = a.epoch;
        const cf w0 = cmul(wc, wg);
        cf *Xt = Xf + (size_t)g * (LINE * L);  // line (g, c) of the frame starts at Xt + LINE * c
        // Octet staging Pst[c2][16]: [0..8) the low octet of column c2; [8..15) elements 1..7 of the
        // high octet of column c2 (its element 0 is the carried row).
        // Tile 0 only (couple 0 is special there): 8 bytes per lane.
        auto emit_pair = [&](int t, int c2i, c2 x) {
            cf xk, xm;
            const cf w0p = (p & 1) ? make_float2(-w0.x, -w0.y) : w0;  // (w0 carries the lane's sigma)
            untangle_pair(x.a, x.b, cmul(w0p, w32(t)), xk, xm);
            const int cm = L - 1 - c2i;
            Xt[LINE * c2i + p] = xk;
            Xt[LINE * c2i + LINE - 1 - p] = xm;  // mirror row M1-p: element CP-1-p of the mirror octet
            Pst[pst_at<CP>(c2i, p)] = bin_power(xk);
            Pst[pst_at<CP>(cm, LINE - 1 - p)] = bin_power(xm);  // (p >= 1 here: element CP-p at [CP-1+CP-p])
        };
        // Every other tile: two outputs (A, B) per call.  Lane pair (2q, 2q+1): the even lane ends up
        // with rows 8g+2q, 8g+2q+1 of both outputs, the odd lane with the mirror rows M1-8g-2q-1,
        // M1-8g-2q of both (one DPP swap per value), so that lanes 0..7 of an i0 write one whole line.
        // (The carried row - the odd lane's partner value at q = 0 - lands in slot 15 of the staging
        // row and is picked up from there by the octet loop.)
        cf mineA = make_float2(0.f, 0.f), sendA = mineA;
        // Addresses: output t of the thread is column c = i0 + (L/16)*t.  Global: the tile's block
        // (scalar) + t * 8 KiB (scalar) + one per-thread byte offset.  Staging: the even lane writes
        // row c, the odd lane row L-1-c = (L/16-1-i0) + (L/16)*(15-t): per-thread base and +-4 KiB
        // stride.  (Opaque copies: see the loop-invariant-address note in pass 1.)
        unsigned gofs = (unsigned)((i0_ * LINE + off_) * (int)sizeof(cf));
        // (the row moves by +-L/16 = 64 per output: the quarter swizzle of pst_at is the same for all sixteen)
        int lbase = pst_at<CP>(ev_ ? i0_ : (L16 - 1 - i0_) + 15 * L16, off_) * (int)sizeof(float);
        int lstride = (ev_ ? L16 : -L16) * LINE * (int)sizeof(float);
        asm volatile("" : "+v"(gofs), "+v"(lbase), "+v"(lstride));
        char *Xtb = reinterpret_cast<char *>(Xt);
        char *Pstb = reinterpret_cast<char *>(Pst);
        auto swap1 = [](float v) {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
        };
        auto emit_two = [&](int tA, int tB, cf mineB, cf sendB) {
            const cf recvA = make_float2(swap1(sendA.x), swap1(sendA.y)), recvB = make_float2(swap1(sendB.x), swap1(sendB.y));
            // scalar base + 32-bit lane offset: the store's own address mode, no vector address math
            typedef __attribute__((address_space(1))) char gchar;
            typedef float gf4 __attribute__((ext_vector_type(4)));
            gchar *bA = (gchar *)(Xtb + (size_t)tA * (L16 * LINE * sizeof(cf))), *bB = (gchar *)(Xtb + (size_t)tB * (L16 * LINE * sizeof(cf)));
            asm volatile("" : "+s"(bA), "+s"(bB));
            *(__attribute__((address_space(1))) gf4 *)(bA + gofs) = gf4{mineA.x, mineA.y, recvA.x, recvA.y};
            *(__attribute__((address_space(1))) gf4 *)(bB + gofs) = gf4{mineB.x, mineB.y, recvB.x, recvB.y};
            *reinterpret_cast<float2 *>(Pstb + lbase + tA * lstride) = make_float2(bin_power(mineA), bin_power(recvA));
            *reinterpret_cast<float2 *>(Pstb + lbase + tB * lstride) = make_float2(bin_power(mineB), bin_power(recvB));
        };
        int tA = 0;
        if (g != 0) {
            constexpr int RL = LastStage<L>::R;
            static_assert((RL == 4 || RL == 8) && NBL * RL == 16, "outputs t and t + 8 of a thread are RL / 2 last-stage outputs apart");
            cf wsv[RL / 2];  // sigma * W_N^k of outputs t = b + NBL * sidx, sidx < RL / 2
            run_last_stage<L>(Wl, i0, u, [&](int b, int sidx, int c2i, c2 x) {
                v2f mine, send;
#ifdef PSDR_ABL_R2_NOUNT  // (timing-only: no untangle arithmetic - wrong results)
                mine = to_v2f(x.a), send = to_v2f(x.b);
                (void)wsv;
#else
                const cf sm = cadd(x.a, x.b), d = csub(x.a, x.b);
                const cf e = make_float2(d.y, -d.x);  // (-i)(a-b)
                if (sidx < RL / 2) {
                    wsv[sidx] = cmul(w0, w32(b + NBL * sidx));
                    const cf wo = cmul(wsv[sidx], e);  // sigma * W_N^k * (-i)(a-b)
                    mine = to_v2f(sm) + to_v2f(wo), send = to_v2f(sm) - to_v2f(wo);
                } else {
                    // W_32^{t+8} = -i W_32^t exactly (the same table entries; cmul(w0, -i c) = -i cmul(w0, c) bit for bit), and
                    // a product with -i is a swap and a sign: the twiddle of output t + 8 is not multiplied out, the -i
                    // rides in the addition's operand selectors.  (The product with (-i)(a-b) rounds its two partial
                    // products in the other order than before: an ulp of the bin, inside every bound the tests state.)
                    const cf wb = cmul(wsv[sidx - RL / 2], e);
                    mine = to_v2f(add_mi(sm, wb)), send = to_v2f(sub_mi(sm, wb));
                }
                mine.y = __uint_as_float(__float_as_uint(mine.y) ^ sgn_mine);
                send.y = __uint_as_float(__float_as_uint(send.y) ^ sgn_send);
#endif
                (void)c2i;  // = i0 + (L/16) * (b + NBL * sidx)
                if ((sidx & 1) == 0) {
                    mineA = from_v2f(mine);
                    sendA = from_v2f(send);
                    tA = b + NBL * sidx;
                } else {
                    emit_two(tA, b + NBL * sidx, from_v2f(mine), from_v2f(send));
                }
            });
        } else {
            run_last_stage<L>(Wl, i0, u, [&](int b, int sidx, int c2i, c2 x) {
                if (p == 0) {  // rows 0 and M1/2: the partners are in other threads
                    exA[c2i] = x.a;
                    exB[c2i] = x.b;
                } else {
                    emit_pair(b + NBL * sidx, c2i, x);
                }
            });
            __syncthreads();
            if (p == 0) {
                // row 0: k = M1*c2 pairs with M1*(L-c2) (same row, column (L-c2) mod L);
                // row M1/2: k = M1/2 + M1*c2 pairs with column L-1-c2 of the same row
                const cf w0z = cmul(a.UA[((unsigned)M1 * (unsigned)i0) >> a.log2UB], a.UB[((unsigned)M1 * (unsigned)i0) & ubm]);
                const unsigned eh = (unsigned)(M1 >> 1) + (unsigned)M1 * (unsigned)i0;
                const cf w0h = cmul(a.UA[eh >> a.log2UB], a.UB[eh & ubm]);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const int c2i = i0 + L16 * t;
                    cf xk, xm;
                    const cf pz = exA[(L - c2i) & (L - 1)];
                    untangle_pair(exA[c2i], make_float2(pz.x, -pz.y), cmul(w0z, w32(t)), xk, xm);
                    Xf[LINE * c2i] = xk;  // line (0, c2), bin 0
                    Pst[pst_at<CP>(c2i, 0)] = bin_power(xk);
                    // bin N/2 is never normalised by the reference (src/fft_impl.cpp:156-160 visits
                    // k < N/2 only): X[N/2] = Re Z[0] - Im Z[0]
                    if (c2i == 0) Xf[(size_t)L << a.log2M1] = make_float2((exA[0].x - exA[0].y) * unscale, 0.f);  // after the M bins
                    const cf ph = exB[L - 1 - c2i];
                    untangle_pair(exB[c2i], make_float2(ph.x, -ph.y), cmul(w0h, w32(t)), xk, xm);
                    Xf[LINE * (L - 1 - c2i) + LINE - 1] = xk;  // row M1/2 closes tile 0's mirror octet, line (0, L-1-c2)
                    seamC[c2i] = bin_power(xk);  // element 0 of the octet [M1/2, M1/2+8) at column c2
                }
            }
        }
        PSDR_TRACE(a.trace, it, 10);
        __syncthreads();  // octet staging (and the carry) complete
        PSDR_TRACE(a.trace, it, 11);
        if (LATE > 0) static_for<NFRONT + LATE / 2, NLD>(issue);
        {
            int8_t *Qf = a.Qt + (size_t)f * a.qt_stride;
            float *Pf = a.Pscr + (size_t)f * a.p_stride;
            float *seamP = a.seamP + (size_t)s * L * CP;  // (seam_in only: those segments come first in the table)
            constexpr int NG = 2 * L / NT;  // octets per thread: chunk q = 2*c2 + side
            // GRP octets at a time: their LDS reads (staging + carried row) are issued together, then the records
            constexpr int GRP = 2;  // same box, interleaved, cfg3: step 6.57 -> 6.52 us per frame, pass 2 1045 -> 1025 us (4: the same)
            static_assert(NG % GRP == 0, "octet groups");
            if constexpr (CP == 4) {
                // Quartets (2048-point rows, four couples per tile): group kk of the thread is side kk & 1 of column
                // (kk >> 1) * NT + tidx; a staging row is [low quartet | elements 1..3 of the high quartet, carry-out]: one
                // 16-byte quarter per side.  Records of 8 bytes [q0 x4 | q1 x2 | q2 | pad], levels 0..2, the two of a column
                // side by side; the level-2 sums go to Pf in the same order for the column tail.
                const float4 *P4 = reinterpret_cast<const float4 *>(Pst);
#pragma unroll
                for (int k0 = 0; k0 < NG; k0 += GRP) {
                    float4 v[GRP];
                    float cin[GRP];
                    uint2 rec_prev = make_uint2(0u, 0u);
                    float pf_prev = 0.f;
                    static_assert(GRP == 2, "a group = the two sides of one column");
#pragma unroll
                    for (int j = 0; j < GRP; j++) {
                        const int kk = k0 + j, sd = kk & 1, c2i = (kk >> 1) * NT + tidx;
                        v[j] = P4[2 * c2i + sd];
                        cin[j] = sd ? carry_r[c2i] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < GRP; j++) {
                        const int kk = k0 + j, sd = kk & 1, c2i = (kk >> 1) * NT + tidx;
                        float pw[4];
                        if (sd) {  // high quartet: element 0 is the carried row, 1..3 sit at [0..3), slot 3 holds this tile's carry-out
                            carry_w[c2i] = v[j].w;
                            if (seg_last && g != 0)
                                __hip_atomic_store(reinterpret_cast<unsigned *>(seamC) + c2i, __float_as_uint(v[j].w), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                            pw[0] = cin[j], pw[1] = v[j].x, pw[2] = v[j].y, pw[3] = v[j].z;
                            if (seam_in) reinterpret_cast<float4 *>(seamP)[c2i] = v[j];  // no carry-in: k_real_seam completes the quartet
                        } else {
                            pw[0] = v[j].x, pw[1] = v[j].y, pw[2] = v[j].z, pw[3] = v[j].w;
                        }
                        uint2 rec;
                        pyr_record4(pw, a.size_log2, rec);
                        // the LOW quartet's record waits for the HIGH one of the same column (the next group of this
                        // iteration): ONE 16-byte store for the two records, one 8-byte store for their level-2 sums
                        // (RecMap::pair: [tile][column][side])
                        if (sd) {
                            const size_t rq = ((size_t)g * L + c2i) * 2;
                            *reinterpret_cast<uint4 *>(Qf + rq * 8) = make_uint4(rec_prev.x, rec_prev.y, rec.x, rec.y);
                            *reinterpret_cast<float2 *>(Pf + rq) = make_float2(pf_prev, pw[0]);
                        } else {
                            rec_prev = rec;
                            pf_prev = pw[0];
                        }
                    }
                    PSDR_SCHED_FENCE();
                }
            } else {
            // Octet kk of the thread: side = kk & 1 (known after unrolling: no per-lane selects), column c2 = (kk >> 1) * NT +
            // tidx - a wave's 64 records of one side are 1 KiB of adjacent records (RecMap mode 2: [tile][side][column]).
            static_assert(L16 % 4 == 0 && NT % 16 == 0, "quarter swizzle: constant per thread");
            const float4 *P4 = reinterpret_cast<const float4 *>(Pst);
            int pbase = 4 * tidx + ((tidx >> 2) & 3);  // quarter j of the thread's row: (pbase ^ j), + 4 * NT per column group
            asm volatile("" : "+v"(pbase));
#pragma unroll
            for (int k0 = 0; k0 < NG; k0 += GRP) {
                float4 v0[GRP], v1[GRP];
                float cin[GRP];
#pragma unroll
                for (int j = 0; j < GRP; j++) {
                    const int kk = k0 + j, sd = kk & 1, cg = kk >> 1;
                    v0[j] = P4[(pbase ^ (2 * sd)) + 4 * NT * cg];
                    v1[j] = P4[(pbase ^ (2 * sd + 1)) + 4 * NT * cg];
                    cin[j] = sd ? carry_r[cg * NT + tidx] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < GRP; j++) {
                    const int kk = k0 + j, sd = kk & 1;
                    const int c2i = (kk >> 1) * NT + tidx;
                    float pw[8];
                    if (sd) {  // high octet: element 0 is the carried row, 1..7 sit at [0..7), and slot 7
                               // holds this tile's own carry-out (row M1-8g of column c2i)
                        carry_w[c2i] = v1[j].w;
                        // (write-through: in hand-off mode the reader is a work-group on another XCD, inside this launch)
                        if (seg_last && g != 0)
                            __hip_atomic_store(reinterpret_cast<unsigned *>(seamC) + c2i, __float_as_uint(v1[j].w), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        pw[0] = cin[j];
                        pw[1] = v0[j].x, pw[2] = v0[j].y, pw[3] = v0[j].z, pw[4] = v0[j].w, pw[5] = v1[j].x, pw[6] = v1[j].y, pw[7] = v1[j].z;
                        if (seam_in) {  // no carry-in: the octet is completed by k_real_seam
                            reinterpret_cast<float4 *>(seamP)[2 * c2i] = v0[j];
                            reinterpret_cast<float4 *>(seamP)[2 * c2i + 1] = v1[j];
                        }
                    } else {
                        pw[0] = v0[j].x, pw[1] = v0[j].y, pw[2] = v0[j].z, pw[3] = v0[j].w, pw[4] = v1[j].x, pw[5] = v1[j].y, pw[6] = v1[j].z, pw[7] = v1[j].w;
                    }
                    // The record is written by EVERY lane (a segment's first tile: with a stale carried row in the
                    // high octets - k_real_seam, which runs before any consumer, writes those records again): stores the
                    // compiler can count, see the note in the lane-alternating form below.
                    {
                        const size_t rp = (size_t)g * (2 * L) + (size_t)sd * L + c2i;  // RecMap mode 2
                        uint4 rec;
                        pyr_record8(pw, a.size_log2, rec);
                        *reinterpret_cast<uint4 *>(Qf + rp * 16) = rec;
                        Pf[rp] = pw[0];
                    }
                }
                PSDR_SCHED_FENCE();
            }
            }  // (CP == 8)
        }
        PSDR_TRACE(a.trace, it, 12);
        if (!SPLIT) __syncthreads();  // the tile is free again (SPLIT: between the next tile's two half-fills)
        PSDR_TRACE(a.trace, it, 13);
        if (seg_last) {
            if (a.segflag && g != 0) post_seg = s;  // (tile 0's carry-out, row M1/2, is k_real_seam's: a later kernel)
            const unsigned s2 = s_next[segit & 1];
            s = snext;
            snext = s2;
            se = sen;
            if (s2 < total) sen = seg_entry(s2);
            j = 0;
            PSDR_WGTRACE(a.trace, 1 + segit);  // (tuning builds) end of the work-group's segit-th segment
            segit++;
        } else {
            j++;
        }
    }
    if (post_seg != NOSEG) {  // the work-group's last segment
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.segflag + post_seg, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PSDR_WGTRACE(a.trace, 7);
    kclk_end(a.kclk);
}

}  // namespace psdr
