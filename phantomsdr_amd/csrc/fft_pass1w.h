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
//    couples).  A NINTH wave, the loader, brings it in by LDS-DMA (global_load_lds_dwordx4: 16 rows x 64 bytes
//    per instruction, no VGPRs) into ONE image (1024 rows x 64 bytes = 64 KiB for cs16: there is no room for
//    two).  Two monotonic LDS counters stand in for the barrier: `freed` (+1 when a compute wave has read its
//    column of an image) and `ready` (= k + 1 when image k has landed).  The registers are the second buffer:
//    a compute wave reads its column of image k+1 into 32 VGPRs late in tile k and converts it at the top of
//    tile k+1, so the loader has a whole tile to fetch image k+2, and every hand-over has about half a tile
//    of slack.  Why a loader wave: vmcnt is one in-order counter for loads AND stores, and pass 1 is bound by
//    its stores (a store is acknowledged 6-7 k cycles after issue): a compute wave that waited for its own DMA
//    to land also waited for every store it had issued before - 3.5 k cycles per tile in the first version.
//    The loader issues no stores, the compute waves never wait on vmcnt.  It also draws the tile tickets.
//    The image is XOR-swizzled through the per-lane SOURCE address (the DMA's LDS side is lane-linear) so
//    that the column reads are (nearly) conflict-free.
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
// a.M2 == 1024.  576 threads: eight compute waves + the loader wave.
#ifndef PSDR_P1W_LOADERS
#define PSDR_P1W_LOADERS 2
#endif
constexpr int kPass1wLoaders = PSDR_P1W_LOADERS;  // loader waves: one alone issues 64 KiB of LDS-DMA in 4-12 k cycles beside the stores
constexpr int kPass1wThreads = (p1w::NW + kPass1wLoaders) * 64;
template <int SB>
__global__ __launch_bounds__(kPass1wThreads) void k_fft_pass1_w(Pass1Args a) {
    using namespace p1w;
    static_assert(SB == 2 || SB == 4, "8- and 16-bit integer samples");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWB = 16 * SB;        // bytes of one image row (16 columns)
    constexpr int IMGB = L * ROWB;       // one image
    constexpr int CPR = ROWB / 16;       // 16-byte chunks per row
    constexpr int NDMA = IMGB / 1024;    // DMA instructions per tile (1 KiB each)
    constexpr int L16 = L / 16;
    constexpr int RL = 4, PL = 256, NBL = 4;  // last stage: radix 4, earlier radices' product 256
    constexpr unsigned END = 0xFFFFFFFFu;
    cf *xall = reinterpret_cast<cf *>(smem);
    cf *Wl = xall + NW * XSLOTS;
    cf *ldsTB = Wl + L;
    unsigned char *img = reinterpret_cast<unsigned char *>(ldsTB + L);
    unsigned *flags = reinterpret_cast<unsigned *>(img + IMGB);
    unsigned *ready = flags, *freed = flags + 1, *seqn = flags + 2, *seq = flags + 4;  // seq[4]: tile index of tile number k at [k & 3]
    constexpr int NLD = kPass1wLoaders;

    const int tid = threadIdx.x;
    kclk_begin(a.kclk);
    const int i0_ = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // compute wave = couple of the tile; 8 = the loader
    const unsigned total = a.total_slots;
    const int fmt = a.fmt;
    const size_t M = (size_t)L << a.log2M2;
    const int M2 = a.M2;
    const size_t g_row = (size_t)M2 * SB;  // bytes between raw rows

    auto tile_coords = [&](unsigned sidx, unsigned &f, unsigned &tl) {
        const unsigned slot = xcd_slot(sidx, total);
        f = slot / a.tiles_per_frame;
        tl = slot - f * a.tiles_per_frame;
    };

    if (tid < 16) flags[tid] = 0;
    for (int i = tid; i < L; i += kPass1wThreads) Wl[i] = a.Wl[i];
    for (int i = tid; i < M2; i += kPass1wThreads) ldsTB[i] = a.TB[i];
    __syncthreads();  // the only work-group barrier of the tile loop's lifetime: tables and zeroed flags are visible

    if (w >= NW) {
        // ================= the loader waves =================
        // Tile number k of this work-group's sequence: index e_k (static for k < 2, then tickets, drawn by loader 0),
        // image fetched as soon as all eight compute waves have read image k-1 (`freed` >= 8 k), each loader its
        // 1/NLD of the rows, `ready` += 1 per loader once its part has landed.
        const int lw = w - NW;
#ifndef PSDR_P1W_NO_PRIO
        __builtin_amdgcn_s_setprio(3);  // the fetches must not queue behind the eight waves' stores for an issue slot
#endif
        TileQueue tq;
        tq.init(a.tickets, total, false, NW * 64);
        tq.draw_first();
        // instruction m covers 16-byte units P = m*64 + lane of the image: row P / CPR, position P % CPR, which holds
        // chunk (position ^ swizzle(row)) of that row (so that the column reads below spread over the banks)
        const unsigned lds_img = (unsigned)(size_t)(lds_void *)img;
        unsigned e0 = blockIdx.x, e1 = blockIdx.x + gridDim.x;
        for (unsigned k = 0;; k++) {
            unsigned tk;
            if (lw == 0) {
                tk = e0 < total ? e0 : END;
                if (i0_ == 0) {
                    asm volatile("" ::: "memory");
                    *(__attribute__((address_space(3))) unsigned *)(seq + (k & 3u)) = tk;
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(seqn, k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                spin_ge(seqn, k + 1u);
                tk = __builtin_amdgcn_readfirstlane(lds_load_relaxed(&seq[k & 3u]));
            }
            if (tk == END) {
                // (`ready` counts signals, not tiles: a loader that ran ahead to the end of the sequence must not make up
                // for another loader's missing share of the tile before)
                spin_ge(freed, 8u * k);
                signal_inc(ready);
                break;
            }
            unsigned f, tl;
            tile_coords(tk, f, tl);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(a.raw) + ((size_t)f * (M / 2) + (size_t)tl * T) * SB;
            PSDR_LTRACE(8);
            spin_ge(freed, 8u * k);
            PSDR_LTRACE(11);
            // Inline asm, not __builtin_amdgcn_global_load_lds: the compiler treats the builtin as a store to LDS that any
            // later LDS read may alias and puts an s_waitcnt vmcnt(0) in front of the next ds_read of the wave.
            // M0 = LDS byte address of the 1 KiB piece (wave-uniform), restored after the instruction.
#pragma unroll 8
            for (int mm = 0; mm < NDMA / NLD; mm++) {
                const int m = lw * (NDMA / NLD) + mm;
                const unsigned P = (unsigned)(m * 64 + i0_);
                const unsigned row = P / CPR, j = P % CPR, c = j ^ ((row >> 2) & (CPR - 1));
                const unsigned char *gsrc = src + (size_t)row * g_row + c * 16u;
                const unsigned ldst = __builtin_amdgcn_readfirstlane(lds_img + (unsigned)m * 1024u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(gsrc), "s"(ldst)
                             : "memory");
            }
            PSDR_LTRACE(12);
            // the next index while the image is in flight: the ticket drawn one tile ago has long arrived
            unsigned e2 = END;
            if (lw == 0) {
                tq.draw_end(&e2, tk);  // (owner lane only; static mode: tk + 2 * gridDim.x)
                tq.draw_begin();
                e2 = __builtin_amdgcn_readfirstlane(e2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // landed (this wave issues no stores: nothing else to wait for)
            PSDR_LTRACE(13);
            signal_inc(ready);
            e0 = e1;
            e1 = e2;
        }
    } else {
        // ================= the eight compute waves =================
        auto tw2 = [&](unsigned eA, unsigned eB, cf &rA, cf &rB) {  // W_M^eA, W_M^eB from the two-level table
            cmul_pair(rA, Wl[eA >> a.log2M2], ldsTB[eA & (unsigned)(M2 - 1)], rB, Wl[eB >> a.log2M2], ldsTB[eB & (unsigned)(M2 - 1)]);
        };
        // image read: row i0 + 64 e, the couple's bytes inside its (swizzled) chunk
        constexpr int CB = 2 * SB;  // bytes of one couple
        const unsigned my_chunk = (unsigned)(w * CB) / 16u, in_chunk = (unsigned)(w * CB) % 16u;
        const unsigned img_lane = (unsigned)i0_ * ROWB + ((my_chunk ^ (((unsigned)i0_ >> 2) & (CPR - 1))) * 16u) + in_chunk;
        // exchange region of this wave: slot(q) = q + (q >> 4)
        lds_cf *xreg = (lds_cf *)xall + w * XSLOTS;
        const int xw0_ = 17 * i0_;                       // stage-0 outputs: q = 16 i0 + s      -> + s
        const int xr_ = i0_ + (i0_ >> 4);                // inputs of the next stage: q = i0 + 64 e -> + 68 e
        const int xw1_ = 272 * (i0_ >> 4) + (i0_ & 15);  // stage-1 outputs: q = 256 a + 16 s + k -> + 17 s

        unsigned rq[16][SB / 2];
        auto read_image = [&](unsigned il) {
            const __attribute__((address_space(3))) unsigned char *ib = (const __attribute__((address_space(3))) unsigned char *)img + il;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if constexpr (SB == 4) {
                    const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(ib + e * (L16 * ROWB));
                    rq[e][0] = v.x, rq[e][1] = v.y;
                } else {
                    rq[e][0] = *(const __attribute__((address_space(3))) unsigned *)(ib + e * (L16 * ROWB));
                }
            }
            // (LDS reads of one wave execute in issue order: the counter moves after the image was read)
            signal_inc(freed);
        };
        auto seq_entry = [&](unsigned k) {
            return (unsigned)__builtin_amdgcn_readfirstlane(lds_load_relaxed(&seq[k & 3u]));
        };
        spin_ge(ready, (unsigned)NLD);
        unsigned s = seq_entry(0);
        if (s != END) read_image(img_lane);

        for (unsigned it = 0; s != END; it++) {
            unsigned f, tl;
            tile_coords(s, f, tl);
            // opaque per-iteration copies: keep the loop-invariant LDS addresses out of long-lived registers
            int i0 = i0_;
            int xw0i = xw0_, xri = xr_, xw1i = xw1_;
            unsigned il = img_lane;
            asm volatile("" : "+v"(i0), "+v"(xw0i), "+v"(xri), "+v"(xw1i), "+v"(il));
            lds_cf *xw0 = xreg + xw0i, *xr = xreg + xri, *xw1 = xreg + xw1i;
            PSDR_TRACE(a.trace, it, 0);

            // ---- convert (src/samplereader.cpp:29-40) and window (src/utils/dsp.cpp:6-11; exp(-i 2 pi n/M) =
            // W_M1^{n1} W_M^{n2}) this wave's column couple of the tile's image, read into rq late in the previous tile
            const unsigned nA = tl * T + 2u * (unsigned)w, nB = nA + 1u;  // n2 of the two columns (wave-uniform)
            c2 u[16];
            {
                cf wbA, wbB;
                tw2(nA, nB, wbA, wbB);
                const v2f wx = {wbA.x, wbB.x}, wy = {wbA.y, wbB.y};
                constexpr float hk = 0.5f * image_scale<SB>();
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
            PSDR_TRACE(a.trace, it, 1);

            // ---- stage 0: radix 16 over e; outputs s of lane i0 are points q = 16 i0 + s
            c2 v[16];
            stage_compute<L, 16, 1>(u, i0, Wl, [&](int, int sidx, int, c2 x) { v[sidx] = x; });
            PSDR_TRACE(a.trace, it, 2);
            // exchange 1, one column at a time through the wave's own 8 KiB (in-order LDS: no barrier)
#pragma unroll
            for (int k = 0; k < 16; k++) xw0[k] = to_v2f(v[k].a);
            PSDR_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 16; e++) u[e].a = from_v2f(xr[68 * e]);
            PSDR_SCHED_FENCE();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; k++) xw0[k] = to_v2f(v[k].b);
            PSDR_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 16; e++) u[e].b = from_v2f(xr[68 * e]);
            PSDR_SCHED_FENCE();
            asm volatile("" ::: "memory");
            PSDR_TRACE(a.trace, it, 3);

            // ---- stage 1: radix 16, twiddles W_1024^{4 q k}, k = i0 & 15; outputs s are points 256 a + 16 s + k
            stage_compute<L, 16, 16>(u, i0, Wl, [&](int, int sidx, int, c2 x) { v[sidx] = x; });
            PSDR_TRACE(a.trace, it, 4);
#pragma unroll
            for (int k = 0; k < 16; k++) xw1[17 * k] = to_v2f(v[k].a);
            PSDR_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 16; e++) u[e].a = from_v2f(xr[68 * e]);
            PSDR_SCHED_FENCE();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; k++) xw1[17 * k] = to_v2f(v[k].b);
            PSDR_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 16; e++) u[e].b = from_v2f(xr[68 * e]);
            PSDR_SCHED_FENCE();
            PSDR_TRACE(a.trace, it, 5);

            // ---- the NEXT tile (number it + 1): its index and, if there is one, this wave's column of its image into
            // registers.  The loader published it about half a tile ago.
            spin_ge(ready, (unsigned)NLD * (it + 2u));
            PSDR_TRACE(a.trace, it, 6);
            const unsigned snext = seq_entry(it + 1u);
            if (snext != END) read_image(il);
            PSDR_TRACE(a.trace, it, 7);

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
            cf *Yb = a.Y + (size_t)f * a.yframe + (size_t)tl * a.yblk + (size_t)w * (2 * L);
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
            PSDR_TRACE(a.trace, it, 10);
            s = snext;
        }
    }
    kclk_end(a.kclk);
}

}  // namespace psdr
