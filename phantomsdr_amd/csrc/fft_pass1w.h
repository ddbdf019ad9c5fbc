// fft_pass1w.h — pass 1 of the 2^20-point IQ transform with WAVE-OWNED column couples (gfx950).
//
// Same arithmetic as k_fft_pass1<1024, 16, SB> (fft_pass.h): convert (src/samplereader.cpp:29-40) +
// periodic Hann (src/utils/dsp.cpp:6-11) + 1024-point column FFT as 16 x 16 x 4 + inter-pass twiddle, for
// what cufftExecC2C / fftwf_execute do in the reference (src/fft_cuda.cu:132-138, src/fft_impl.cpp:145).
// What differs is who owns what:
//
//  * k_fft_pass1 spreads a column couple over all eight waves of the work-group (8 lanes = one 64-byte
//    piece of a raw row), so both stage exchanges cross waves and cost four work-group barriers per
//    tile.  Measured on MI355X (round 3, timing-only ablation): the same kernel with the barriers replaced
//    by plain LDS waits runs 515 instead of 606 us per 256 frames - and keeping a single one of the four
//    gives nothing back (624 us): any barrier re-aligns the eight waves, and aligned waves want the VALU,
//    the LDS and the store path all at the same time.
//  * Here compute wave w owns couple w of the tile outright: lane i0 holds rows i0 + 64 e of both columns,
//    both exchanges stay inside the wave (LDS operations of one wave execute in order: no barrier, no flag),
//    and the sixteen outputs of a lane pair with 64 consecutive rows c1 in its neighbours' - Y is stored
//    COUPLE-MAJOR, [frame][pass-1 tile][couple][c1][2], 1 KiB contiguous per store instruction
//    (pass 2 reads it with k_fft_pass2<.., YCM = true>: 256-byte pieces, measured no slower than the
//    2 KiB pieces of the row-major blocks).
//  * The one thing the eight waves still share is the raw tile (a row piece of 64 bytes holds all eight
//    couples).  It is fetched the way k_fft_pass1 fetches it - coalesced, 8 lanes per row piece, a whole tile
//    ahead, into registers P of whichever thread the lane mapping says - and handed over through ONE image
//    in LDS (1024 rows x 64 bytes = 64 KiB for cs16: there is no room for two): at checkpoint alpha of its
//    tile a wave writes its eighth of the NEXT tile's rows into the image, at checkpoint beta (half a tile
//    later) it reads its own column of that image into registers rq, and converts it at the top of the
//    next tile.  Two monotonic LDS counters stand in for the barrier: `written` (+1 per wave and image) and
//    `freed` (+1 per wave that has read its column).  Nothing slow sits between the two sides of a
//    hand-over - the HBM latency is spent in P, a tile earlier - and each side has about half a tile of
//    slack, so the waves drift apart by up to that much.  (Two earlier forms measured slower than the
//    barrier kernel: LDS-DMA by the compute waves - vmcnt is ONE in-order counter for loads and stores and
//    pass 1 is bound by its stores, so waiting for one's own DMA meant waiting 3.5 k cycles per tile for
//    one's own stores - and LDS-DMA by a ninth loader wave: beside 128 KiB of stores per tile the 64 KiB
//    fetch takes 10-18 k cycles to issue and land, and with a single image the loader cannot start before
//    the slowest wave has read the previous one: period = fetch + skew.)
//    The image is XOR-swizzled (position of a couple in its row) so that both the row-wise writes and the
//    column-wise reads are conflict-free.
//  * LDS: a wave's exchange region holds ONE column (8 KiB + padding) and is used twice per exchange
//    (column a, then column b: the registers of b wait their turn), which is what makes room for the
//    image: 8 x 8.5 KiB regions + 16 KiB of twiddle tables + 64 KiB image (cs16) = 148 KiB.
#pragma once
#include "fft_pass.h"

#ifdef PSDR_TRACE_ON
#define PSDR_LTRACE(slot_)                                                                 \
    do {                                                                                   \
        if (a.trace && blockIdx.x == 0 && i0_ == 0 && lw == 0 && k < 8) a.trace[k * 16 + (slot_)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PSDR_LTRACE(slot_) \
    do {                   \
    } while (0)
#endif

namespace psdr {

namespace p1w {
constexpr int L = 1024, T = 16, NW = 8;
constexpr int XSLOTS = 1088;  // 1024 points + one pad slot per 16: slot(q) = q + (q >> 4)
typedef __attribute__((address_space(3))) void lds_void;
// explicit LDS pointers (a laundered generic pointer compiles to flat_* accesses); plain vector element types:
// HIP's float2 / uint2 classes have no address-space-qualified copy operations
typedef __attribute__((address_space(3))) v2f lds_cf;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ unsigned lds_load_relaxed(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until *p >= target (monotonic counter in LDS).  Everything this guards is LDS traffic of the
// same CU, which the LDS executes in issue order: no fence beyond the compiler's is needed.
__device__ __forceinline__ void spin_ge(const unsigned *p, unsigned target) {
    while ((int)(__builtin_amdgcn_readfirstlane(lds_load_relaxed(p)) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void signal_inc(unsigned *p) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
}  // namespace p1w

// LDS footprint of k_fft_pass1_w<SB>
template <int SB>
constexpr size_t pass1w_lds_bytes() {
    return (size_t)p1w::NW * p1w::XSLOTS * sizeof(cf) + 2 * (size_t)p1w::L * sizeof(cf) + (size_t)p1w::L * 16 * SB + 64;
}

// SB: bytes per complex sample of the raw input (2: u8/s8, 4: u16/s16).  IQ input only (a.is_real == 0),
// a.M2 == 1024.  512 threads.
constexpr int kPass1wThreads = 512;
template <int SB>
__global__ __launch_bounds__(kPass1wThreads) void k_fft_pass1_w(Pass1Args a) {
    using namespace p1w;
    static_assert(SB == 2 || SB == 4, "8- and 16-bit integer samples");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWB = 16 * SB;  // bytes of one image row (16 columns)
    constexpr int IMGB = L * ROWB; // the image
    constexpr int CB = 2 * SB;     // bytes of one couple
    constexpr int WPL = SB / 2;    // 32-bit words of one couple
    constexpr int L16 = L / 16;
    constexpr int RL = 4, PL = 256, NBL = 4;  // last stage: radix 4, earlier radices' product 256
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    cf *xall = reinterpret_cast<cf *>(smem);
    cf *Wl = xall + NW * XSLOTS;
    cf *ldsTB = Wl + L;
    unsigned char *img = reinterpret_cast<unsigned char *>(ldsTB + L);
    unsigned *flags = reinterpret_cast<unsigned *>(img + IMGB);
    unsigned *written = flags, *freed = flags + 1, *seqn = flags + 2, *seq = flags + 8;  // seq[8]: entry j at [j & 7]

    const int tid = threadIdx.x;
    kclk_begin(a.kclk);
    const int i0_ = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = couple of the tile (compute role)
    const unsigned total = a.total_slots;
    const int fmt = a.fmt;
    const size_t M = (size_t)L << a.log2M2;
    const int M2 = a.M2;
    const size_t g_row = (size_t)M2 * SB;  // bytes between raw rows

    auto tw2 = [&](unsigned eA, unsigned eB, cf &rA, cf &rB) {  // W_M^eA, W_M^eB from the two-level table
        cmul_pair(rA, Wl[eA >> a.log2M2], ldsTB[eA & (unsigned)(M2 - 1)], rB, Wl[eB >> a.log2M2], ldsTB[eB & (unsigned)(M2 - 1)]);
    };
    auto tile_coords = [&](unsigned sidx, unsigned &f, unsigned &tl) {
        const unsigned slot = xcd_slot(sidx, total);
        f = slot / a.tiles_per_frame;
        tl = slot - f * a.tiles_per_frame;
    };

    // ---- per-lane constants
    // fetch role (the lane mapping of k_fft_pass1: 8 lanes = one row piece): couple pc of rows i0c + 64 e
    const int pc_ = tid & 7, i0c_ = tid >> 3;
    const unsigned g_lane = (unsigned)i0c_ * (unsigned)g_row + (unsigned)pc_ * CB;  // + e * 64 rows
    // image: row r keeps couple c at position c ^ ((r >> 2) & 7): a row piece is written by eight lanes at once
    // (any permutation of its eight positions is conflict-free), a column is read by lanes with consecutive r
    // (32 consecutive rows hit 32 different 8-byte bank pairs)
    const unsigned iw_lane = (unsigned)i0c_ * ROWB + (((unsigned)pc_ ^ (((unsigned)i0c_ >> 2) & 7u)) * CB);
    const unsigned ir_lane = (unsigned)i0_ * ROWB + (((unsigned)w ^ (((unsigned)i0_ >> 2) & 7u)) * CB);
    // exchange region of this wave: slot(q) = q + (q >> 4)
    lds_cf *xreg = (lds_cf *)xall + w * XSLOTS;
    const int xw0_ = 17 * i0_;                       // stage-0 outputs: q = 16 i0 + s      -> + s
    const int xr_ = i0_ + (i0_ >> 4);                // inputs of the next stage: q = i0 + 64 e -> + 68 e
    const int xw1_ = 272 * (i0_ >> 4) + (i0_ & 15);  // stage-1 outputs: q = 256 a + 16 s + k -> + 17 s

    unsigned P[16][WPL];   // fetched share of the image after next
    unsigned rq[16][WPL];  // this wave's column of the next image
    auto fetch = [&](unsigned sidx) {  // tile sidx's rows i0c + 64 e, couple pc -> P (asynchronous)
        unsigned f, tl;
        tile_coords(sidx, f, tl);
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.raw) + ((size_t)f * (M / 2) + (size_t)tl * T) * SB + g_lane;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const unsigned char *q = src + (size_t)e * (L16 * g_row);
            if constexpr (SB == 4) {
                const uint2 v = *reinterpret_cast<const uint2 *>(q);
                P[e][0] = v.x, P[e][1] = v.y;
            } else {
                P[e][0] = *reinterpret_cast<const unsigned *>(q);
            }
        }
    };
    auto write_image = [&](unsigned iw) {
        lds_u8 *ib = (lds_u8 *)img + iw;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if constexpr (SB == 4)
                *(lds_u32x2 *)(ib + e * (L16 * ROWB)) = u32x2{P[e][0], P[e][1]};
            else
                *(lds_u32 *)(ib + e * (L16 * ROWB)) = P[e][0];
        }
        signal_inc(written);
    };
    auto read_image = [&](unsigned ir) {
        const lds_u8 *ib = (const lds_u8 *)img + ir;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if constexpr (SB == 4) {
                const u32x2 v = *(const lds_u32x2 *)(ib + e * (L16 * ROWB));
                rq[e][0] = v.x, rq[e][1] = v.y;
            } else {
                rq[e][0] = *(const lds_u32 *)(ib + e * (L16 * ROWB));
            }
        }
        signal_inc(freed);  // (LDS operations of one wave execute in issue order: the counter moves after the reads)
    };

    // ---- the work-group's tile sequence: entries 0 and 1 are static, entry j >= 2 comes from the ticket queue and is
    // published by thread 0 two tiles before anyone fetches it (seq[j & 7] valid once *seqn >= j - 1)
    TileQueue tq;
    tq.init(a.tickets, total);
    unsigned s = blockIdx.x, s1 = blockIdx.x + gridDim.x;
    if (tid < 16) flags[tid] = 0;
    for (int i = tid; i < L; i += kPass1wThreads) Wl[i] = a.Wl[i];
    for (int i = tid; i < M2; i += kPass1wThreads) ldsTB[i] = a.TB[i];
    if (s < total) fetch(s);
    tq.draw_first();  // (waits for the ticket - and the first fetch, needed at once anyway)
    __syncthreads();  // the only work-group barrier before the end of the kernel: tables and zeroed flags are visible
    auto publish = [&](unsigned j, unsigned prev2) {  // thread 0: entry j from the pending draw, then the next draw
        unsigned e = 0xFFFFFFFFu;
        tq.draw_end(&e, prev2);
        tq.draw_begin();
        if (tid == 0) {
            *(lds_u32 *)(seq + (j & 7u)) = e;
            asm volatile("" ::: "memory");
            __hip_atomic_store(seqn, j - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto entry = [&](unsigned j) {  // blocking: entry j of the sequence
        spin_ge(seqn, j - 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane(lds_load_relaxed(&seq[j & 7u]));
    };
    publish(2u, s);
    if (s < total) {
        write_image(iw_lane);                 // image 0
        if (s1 < total) fetch(s1);            // image 1 -> P
        spin_ge(written, 8u);
        read_image(ir_lane);
    }
    // The tile loop is ROTATED: an iteration is [tile k after its checkpoint alpha | checkpoint alpha of tile k+1], so that
    // the fetch is the last vector-memory operation on both edges into the loop header.  Otherwise the compiler - which
    // merges the pending-operation state of the entry edge (fetch, nothing after it) and of the back edge (fetch, then
    // sixteen stores) conservatively - waits for the fetched registers with vmcnt(0): for the tile's own stores, 3-7 k cycles.
    unsigned s2 = 0xFFFFFFFFu;
    // ---- checkpoint alpha, at the top of tile k: the next tile's rows (fetched a tile ago) go into the image once every wave
    // has read its column of this tile's image (late in the previous tile); then the fetch of the tile after next
    auto alpha = [&](unsigned k) {
        unsigned iw = iw_lane;
        asm volatile("" : "+v"(iw));
        PSDR_TRACE(a.trace, k, 0);
        publish(k + 3u, s1);  // thread 0: entry k + 3 (needed by the fetch at tile k + 1's checkpoint alpha)
        s2 = 0xFFFFFFFFu;
        if (s1 < total) {
            spin_ge(freed, 8u * (k + 1u));
            PSDR_TRACE(a.trace, k, 8);
            write_image(iw);
            s2 = entry(k + 2u);
            PSDR_SCHED_FENCE();
            if (s2 < total) fetch(s2);
            PSDR_SCHED_FENCE();
        }
        PSDR_TRACE(a.trace, k, 9);
    };
    if (s < total) alpha(0u);

    for (unsigned k = 0; s < total;) {
        unsigned f, tl;
        tile_coords(s, f, tl);
        const unsigned nA = tl * T + 2u * (unsigned)w, nB = nA + 1u;  // n2 of the two columns (wave-uniform)
        const bool more = s1 < total;
        // opaque per-iteration copies: keep the loop-invariant LDS addresses out of long-lived registers
        int i0 = i0_;
        int xw0i = xw0_, xri = xr_, xw1i = xw1_;
        unsigned ir = ir_lane;
        asm volatile("" : "+v"(i0), "+v"(xw0i), "+v"(xri), "+v"(xw1i), "+v"(ir));
        lds_cf *xw0 = xreg + xw0i, *xr = xreg + xri, *xw1 = xreg + xw1i;

        // ---- convert (src/samplereader.cpp:29-40) and window (src/utils/dsp.cpp:6-11; exp(-i 2 pi n/M) =
        // W_M1^{n1} W_M^{n2}) this wave's column couple of the tile's image, read into rq late in the previous tile
        c2 u[16];
        {
            cf wbA, wbB;
            tw2(nA, nB, wbA, wbB);
            const v2f wx = {wbA.x, wbB.x}, wy = {wbA.y, wbB.y};
            const float hk = 0.5f * image_scale<SB>() * a.yscale;  // (Pass1Args::yscale: the output scale rides in the weights)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const c2 x = words_to_c2<SB, false>(rq[e], fmt);
                const cf wl = Wl[i0 + e * L16];
                v2f t, t2;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(to_v2f(wl)), "v"(wy));
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                    : "=v"(t2)
                    : "v"(to_v2f(wl)), "v"(wx), "v"(t));
                const v2f w2 = {fmaf(-hk, t2.x, hk), fmaf(-hk, t2.y, hk)};
                u[e].a = scale_lo(x.a, w2);
                u[e].b = scale_hi(x.b, w2);
            }
        }
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, k, 1);
        // ---- stage 0: radix 16 over e; outputs s of lane i0 are points q = 16 i0 + s
        c2 v[16];
        stage_compute<L, 16, 1>(u, i0, Wl, [&](int, int sidx, int, c2 x) { v[sidx] = x; });
        PSDR_TRACE(a.trace, k, 2);
        // exchange 1, one column at a time through the wave's own 8 KiB (in-order LDS: no barrier)
#pragma unroll
        for (int j = 0; j < 16; j++) xw0[j] = to_v2f(v[j].a);
        PSDR_SCHED_FENCE();
#pragma unroll
        for (int e = 0; e < 16; e++) u[e].a = from_v2f(xr[68 * e]);
        PSDR_SCHED_FENCE();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; j++) xw0[j] = to_v2f(v[j].b);
        PSDR_SCHED_FENCE();
#pragma unroll
        for (int e = 0; e < 16; e++) u[e].b = from_v2f(xr[68 * e]);
        PSDR_SCHED_FENCE();
        asm volatile("" ::: "memory");
        PSDR_TRACE(a.trace, k, 3);

        // ---- stage 1: radix 16, twiddles W_1024^{4 q k}, k = i0 & 15; outputs s are points 256 a + 16 s + k
        stage_compute<L, 16, 16>(u, i0, Wl, [&](int, int sidx, int, c2 x) { v[sidx] = x; });
        PSDR_TRACE(a.trace, k, 4);
#pragma unroll
        for (int j = 0; j < 16; j++) xw1[17 * j] = to_v2f(v[j].a);
        PSDR_SCHED_FENCE();
#pragma unroll
        for (int e = 0; e < 16; e++) u[e].a = from_v2f(xr[68 * e]);
        PSDR_SCHED_FENCE();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; j++) xw1[17 * j] = to_v2f(v[j].b);
        PSDR_SCHED_FENCE();
#pragma unroll
        for (int e = 0; e < 16; e++) u[e].b = from_v2f(xr[68 * e]);
        PSDR_SCHED_FENCE();
        PSDR_TRACE(a.trace, k, 5);

        // ---- checkpoint beta: this wave's column of the next tile's image into registers, once all eight waves have
        // written their rows (about half a tile ago)
        if (more) {
            spin_ge(written, 8u * (k + 2u));
            PSDR_TRACE(a.trace, k, 6);
            read_image(ir);
        }
        PSDR_TRACE(a.trace, k, 7);

        // ---- inter-pass twiddle W_M^{n2 kappa}, kappa = i0 + 64 b + 256 s, as base * stepB^b * stepS^s; client
        // order (a.rot): row c1 = (k1 - 1) mod M1 with W_M^{n2 (c1 + 1)}, and (-1)^{n2} on the odd column
        cf tbA[NBL], tbB[NBL], tsA[RL], tsB[RL], w00A, w00B;
        {
            tw2(nA * (unsigned)i0, nB * (unsigned)i0, tbA[0], tbB[0]);
            if (a.rot) tbB[0] = make_float2(-tbB[0].x, -tbB[0].y);
            cf sbA, sbB, ssA, ssB;
            tw2(nA * (unsigned)L16, nB * (unsigned)L16, sbA, sbB);
            tw2(nA * (unsigned)PL, nB * (unsigned)PL, ssA, ssB);
#pragma unroll
            for (int b = 1; b < NBL; b++) cmul_pair(tbA[b], tbA[b - 1], sbA, tbB[b], tbB[b - 1], sbB);
            tsA[0] = tsB[0] = make_float2(1.f, 0.f);
            tsA[1] = ssA, tsB[1] = ssB;
#pragma unroll
            for (int q = 2; q < RL; q++) cmul_pair(tsA[q], tsA[q - 1], ssA, tsB[q], tsB[q - 1], ssB);
            w00A = tbA[0];
            w00B = tbB[0];
            if (a.rot && i0 == 0) {  // bin k1 = 0 goes to row M1 - 1 with W_M^{n2 M1}
                tw2(nA * (unsigned)L, nB * (unsigned)L, w00A, w00B);
                w00B = make_float2(-w00B.x, -w00B.y);
            }
        }
        // couple-major block of this tile: [couple][c1][2]
        cf *Yb = a.Y + (size_t)(f & a.ymask) * a.yframe + (size_t)tl * a.yblk + (size_t)w * (2 * L);
        cf *Yl = Yb + 2 * (i0 - (a.rot ? 1 : 0));                   // row k1 - rot of this lane's first output
        cf *Y00 = (a.rot && i0 == 0) ? Yb + 2 * (L - 1) : Yl;      // ... which wraps for bin 0
        stage_compute<L, RL, PL>(u, i0, Wl, [&](int b, int sidx, int, c2 x) {
            cf wA, wB, yA, yB;
            if (sidx == 0) {
                wA = b == 0 ? w00A : tbA[b];
                wB = b == 0 ? w00B : tbB[b];
            } else {
                cmul_pair(wA, tbA[b], tsA[sidx], wB, tbB[b], tsB[sidx]);
            }
            cmul_pair(yA, x.a, wA, yB, x.b, wB);
            cf *dst = (b == 0 && sidx == 0) ? Y00 : Yl + 2 * (b * L16 + sidx * PL);
            *reinterpret_cast<float4 *>(dst) = make_float4(yA.x, yA.y, yB.x, yB.y);
        });
        PSDR_TRACE(a.trace, k, 10);
        // ---- checkpoint alpha of the next tile
        s = s1;
        s1 = s2;
        k++;
        if (s < total) alpha(k);
    }
    kclk_end(a.kclk);
}

}  // namespace psdr
