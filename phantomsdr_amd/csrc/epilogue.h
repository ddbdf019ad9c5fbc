// epilogue.h — kernels after the FFT passes: real-input untangle + power/quantise,
// upper levels of the waterfall pyramid, waterfall slice gather.
//
//   k_untangle_real  : R2C as an N/2-point C2C + Hermitian untangle, fused with the
//                      reference's power_and_quantize (src/fft_impl.cpp:24-44, called
//                      with base_idx = 0 for real input, :149-160) and pyramid levels 1..7
//   k_pyramid_tail   : half_and_quantize (src/fft_impl.cpp:45-61,162-172) from the
//                      partial level left in scratch by the fused pass-2 / untangle kernel
//   k_waterfall_gather: waterfall_loop + send_waterfall byte ranges
//                      (src/websocket.cpp:207-236, src/waterfall.cpp:44-51)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "butterfly.h"
#include "quantize.h"
#include "types.h"

namespace psdr {

// Wave-level continuation of the pyramid: lane holds `s` = P_lvl[idx]; adjacent lanes
// hold adjacent indices.  Emits levels lvl+1 .. lvl+6 (pairs across lanes via xor
// shuffles: the tree (a+b)+(c+d) is exactly the reference's repeated pair sum).
// Returns the level lvl+6 sum (valid in every lane; lane 0 of the wave owns it).
__device__ __forceinline__ float wave_pyramid(float s, size_t idx, int lvl, int nlevels,
                                              int size_log2, int8_t *Qf, size_t R, bool valid) {
    const int lane = threadIdx.x & 63;
    size_t qoff = 0;
    for (int i = 0; i <= lvl; i++) qoff += R >> i;  // byte offset of level lvl+1
#pragma unroll
    for (int d = 0; d < 6; d++) {
        const float o = __shfl_xor(s, 1 << d, 64);
        s = __fadd_rn(s, o);
        const int lv = lvl + 1 + d;
        idx >>= 1;
        if (valid && lv < nlevels && (lane & ((2 << d) - 1)) == 0 && idx < (R >> lv))
            Qf[qoff + idx] = (int8_t)quantize_u8(s, size_log2 - lv);
        qoff += R >> lv;
    }
    return s;
}

struct TailArgs {
    const float *Pin;  // [nframes][in_stride] level lvl_in powers, len_in valid
    size_t in_stride;
    size_t len_in;
    int lvl_in;
    int nlevels;
    int size_log2;
    int8_t *Q;
    size_t q_stride;
    size_t R;
    float *Pout;  // [nframes][out_stride] level lvl_in+7
    size_t out_stride;
    RecMap map;  // order of Pin (the fused pass-2 epilogue leaves it tile-major)
};

// each thread sums one pair (level lvl_in+1), then the wave continues 6 more levels
__global__ __launch_bounds__(256) void k_pyramid_tail(TailArgs a) {
    const unsigned f = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t npairs = a.len_in >> 1;
    const bool valid = j < npairs;
    const float *Pf = a.Pin + (size_t)f * a.in_stride;
    int8_t *Qf = a.Q + (size_t)f * a.q_stride;
    float s = 0.f;
    if (valid) {
        s = __fadd_rn(Pf[a.map.pos(2 * j)], Pf[a.map.pos(2 * j + 1)]);
        const int lv = a.lvl_in + 1;
        if (lv < a.nlevels) {
            size_t qoff = 0;
            for (int i = 0; i < lv; i++) qoff += a.R >> i;
            Qf[qoff + j] = (int8_t)quantize_u8(s, a.size_log2 - lv);
        }
    }
    const float top = wave_pyramid(s, j, a.lvl_in + 1, a.nlevels, a.size_log2, Qf, a.R, valid);
    if (valid && (threadIdx.x & 63) == 0 && a.Pout) a.Pout[(size_t)f * a.out_stride + (j >> 6)] = top;
}

// ---- column tail: the levels a fused pass 2 leaves open, one THREAD per output row ---------------
// The fused epilogues leave the level-LT sums in tile-major order (RecMap): the groups of one output
// row c2 (M1 adjacent bins = NG groups) sit L floats (or octet pairs) apart, so the generic tail above
// reads every 4-byte value from a different line (it took 220 us per 256 frames of cfg2 and 460 us for
// the fused real path, on the side stream, in the way of the next batch's passes).  Here lane = row
// c2: every load is contiguous across the wave, the pair-sum tree of a row runs in registers
// (half_and_quantize, src/fft_impl.cpp:45-61), and levels LT+1 .. LT+log2(NG) go straight to the
// level-major int8 buffer.  Pout gets the row sums (level LT+log2 NG, natural order) for
// k_pyramid_tail to finish the few levels above.
struct ColTailArgs {
    const float *Pin;  // [nframes][in_stride]
    size_t in_stride;
    int mode;  // RecMap::mapped: 1 = IQ tiles of 16 rows (group i of row c at i*L + c),
               // 2 = fused real: low octet of tile g of column c at (2g)*L + c, its mirror octet at (2g+1)*L + c
    int L, l2L;
    int lvl_in, nlevels, size_log2;
    int8_t *Q;
    size_t q_stride;
    size_t R;
    float *Pout;
    size_t out_stride;
};
// levels +1..+4 of 16 consecutive groups (chunk `ch` of row c); returns their sum.  The quantised bytes go to the
// wave's LDS image sq: level lvl_in + d of the wave's 64 rows is the contiguous piece [64][NG >> d] at soff[d].
template <int NG>
__device__ __forceinline__ float col_chunk16(float (&v)[16], int ch, int cl, const ColTailArgs &a, int8_t *sq, const int (&soff)[12]) {
#pragma unroll
    for (int d = 1; d <= 4; d++) {
        const int cnt = 16 >> d;
#pragma unroll
        for (int i = 0; i < cnt; i++) v[i] = __fadd_rn(v[2 * i], v[2 * i + 1]);
        const int lv = a.lvl_in + d;
        if (lv < a.nlevels) {
            int8_t *dst = sq + soff[d] + cl * (NG >> d) + ch * cnt;
            if (d == 1)
                store_q<8>(dst, v, a.size_log2 - lv);
            else if (d == 2)
                store_q<4>(dst, v, a.size_log2 - lv);
            else if (d == 3)
                store_q<2>(dst, v, a.size_log2 - lv);
            else
                store_q<1>(dst, v, a.size_log2 - lv);
        }
    }
    return v[0];
}
// One wave = 64 adjacent rows.  Stores: a row's bytes of level lvl_in + d are NG >> d <= 128 bytes, so a lane that
// stored its own row's pieces straight to HBM wrote 1 .. 8 bytes at a time, 32 .. 128 bytes apart (measured: 129 MB
// written per 256 frames of cfg3 for 33 MB of output, 457 MB for 67 MB at 2^22 points).  The wave's 64 rows of one
// level are ONE contiguous piece of the level-major buffer: they are collected in LDS and go out as whole lines.
// One wave per image.  Beside a pass's work-group a CU has ~16 KiB of LDS left - two images of 8 KiB (NG = 128), ONE of
// 16 KiB (NG = 256) - so the kernel takes 0.9 - 1.0 ms per 512 frames of 2^21 points on the side stream.  Round 6 built the
// form in which FOUR waves share an image (chunks dealt round robin, the chunk sums meeting in LDS): 0.45 ms
// (profiles/r06_side_stream_timeline.txt) - and 60 % more instructions per image (every wave's offset tables, the exchange),
// concentrated beside the first pass: bench.py's cfg3 2 % SLOWER (first pass +11 %), level with 1024 clients, where the side
// stream is the critical path (profiles/r06_col_tail_waves.json).  What a consumer costs the step is its instruction
// count, not its duration: taken out again.
#ifndef PSDR_CT_WPE
#define PSDR_CT_WPE 1
#endif
template <int NG, bool PAIRED = false>
__global__ __launch_bounds__(64, PSDR_CT_WPE) void k_col_tail(ColTailArgs a) {
    static_assert(NG % 32 == 0 && NG <= 256, "groups per row");
    constexpr int NC = NG / 16, LOGNG = NG == 64 ? 6 : (NG == 128 ? 7 : 8);
    __shared__ __attribute__((aligned(16))) int8_t sq[64 * NG];
    const int cl = threadIdx.x, c0 = blockIdx.x * 64, c = c0 + cl, f = blockIdx.y;
    const float *Pf = a.Pin + (size_t)f * a.in_stride;
    int8_t *Qf = a.Q + (size_t)f * a.q_stride;
    size_t qoff[12];  // byte offset of level lvl_in + d
    int soff[12];     // ... and of its 64-row piece inside sq
    {
        size_t o = 0;
        for (int i = 0; i <= a.lvl_in; i++) o += a.R >> i;
        int so = 0;
#pragma unroll
        for (int d = 1; d < 12; d++) {
            qoff[d] = o;
            o += a.R >> (a.lvl_in + d);
            soff[d] = so;
            so += d <= LOGNG ? 64 * (NG >> d) : 0;
        }
        qoff[0] = 0;
        soff[0] = 0;
    }
    float cs[NC];
    if (a.mode == 1) {
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = Pf[((size_t)(16 * ch + i) << a.l2L) + c];
            cs[ch] = col_chunk16<NG>(v, ch, cl, a, sq, soff);
        }
    } else {
        // tile g's two rows of sums: the low octets (group g) and the mirror octets (group NG-1-g) of the 64 columns
#pragma unroll
        for (int ch = 0; ch < NC / 2; ch++) {
            float lo[16], hi[16];
            if constexpr (PAIRED) {  // (RecMap::pair) the two sums of a (tile, column) side by side: one 8-byte load
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float2 lh = *reinterpret_cast<const float2 *>(Pf + ((((size_t)(16 * ch + i) << a.l2L) + c) << 1));
                    lo[i] = lh.x;
                    hi[15 - i] = lh.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    lo[i] = Pf[((size_t)(2 * (16 * ch + i)) << a.l2L) + c];
                    hi[15 - i] = Pf[((size_t)(2 * (16 * ch + i) + 1) << a.l2L) + c];
                }
            }
            cs[ch] = col_chunk16<NG>(lo, ch, cl, a, sq, soff);
            cs[NC - 1 - ch] = col_chunk16<NG>(hi, NC - 1 - ch, cl, a, sq, soff);
        }
    }
    // levels +5 .. +log2(NG) over the chunk sums
#pragma unroll
    for (int d = 5; d <= LOGNG; d++) {
        const int cnt = NG >> d;
#pragma unroll
        for (int i = 0; i < NC / 2; i++)
            if (i < cnt) cs[i] = __fadd_rn(cs[2 * i], cs[2 * i + 1]);
        const int lv = a.lvl_in + d;
        if (lv < a.nlevels) {
            int8_t *dst = sq + soff[d] + cl * cnt;
            if (cnt >= 8)
                store_q<8>(dst, cs, a.size_log2 - lv);
            else if (cnt == 4)
                store_q<4>(dst, cs, a.size_log2 - lv);
            else if (cnt == 2)
                store_q<2>(dst, cs, a.size_log2 - lv);
            else
                store_q<1>(dst, cs, a.size_log2 - lv);
        }
    }
    if (a.Pout) a.Pout[(size_t)f * a.out_stride + c] = cs[0];
    __syncthreads();  // (one wave: orders the LDS bytes of all lanes before the flush)
#pragma unroll
    for (int d = 1; d <= LOGNG; d++) {
        if (a.lvl_in + d >= a.nlevels) break;
        const int bytes = 64 * (NG >> d);  // >= 64: whole 16-byte pieces
        int8_t *dst = Qf + qoff[d] + (size_t)c0 * (NG >> d);
        for (int o = cl * 16; o < bytes; o += 64 * 16)
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(sq + soff[d] + o);
    }
}

struct UntangleArgs {
    const cf *Z;  // [nframes][M] unnormalised N/2-point transform of the packed input
    cf *X;        // [nframes][spec_stride] (spec_stride >= M+1): k order
    size_t spec_stride;
    size_t M;      // N/2
    const cf *TA;  // W_N^{h*B}
    const cf *TB;  // W_N^{l}
    int log2B;
    float inv_n;
    int size_log2;
    int nlevels;
    int8_t *Q;
    size_t q_stride;
    float *Pscr;  // level 8 sums
    size_t p_stride;
};

// X[k] = E + W_N^k O, E = (Z[k]+conj Z[M-k])/2, O = -i (Z[k]-conj Z[M-k])/2
__device__ __forceinline__ cf untangle_one(cf a, cf b, cf w) {
    const cf E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    const cf D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
    const cf O = make_float2(D.y, -D.x);
    const cf wo = cmul(w, O);
    return make_float2(E.x + wo.x, E.y + wo.y);
}

// X[M-k] from the same pair of inputs: W_N^{M-k} = -conj(W_N^k)
__device__ __forceinline__ cf untangle_mirror(cf zk, cf zmk, cf w) {
    return untangle_one(zmk, zk, make_float2(-w.x, w.y));
}

// Thread j (< M/8) owns the forward group k0..k0+3 (k0 = 4j) AND the mirrored aligned group
// M-k0-4..M-k0-1: both come from the same ten inputs Z[k0..k0+4], Z[M-k0-4..M-k0], so Z is
// read once; every store and all but two loads are 16 bytes per lane.  Lanes ascend through
// the forward groups and descend through the mirrored ones; the pair-sum tree is the same.
__global__ __launch_bounds__(256) void k_untangle_real(UntangleArgs a) {
    const unsigned f = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = a.M;
    const bool valid = j < M / 8;
    const cf *Zf = a.Z + (size_t)f * M;
    cf *Xf = a.X + (size_t)f * a.spec_stride;
    int8_t *Qf = a.Q + (size_t)f * a.q_stride;
    float sF = 0.f, sM = 0.f;
    const size_t gF = j, gM = M / 4 - 1 - j;  // level-2 indices of the two groups
    if (valid) {
        const size_t k0 = 4 * j;
        const float4 f0 = reinterpret_cast<const float4 *>(Zf)[2 * j];      // Z[k0], Z[k0+1]
        const float4 f1 = reinterpret_cast<const float4 *>(Zf)[2 * j + 1];  // Z[k0+2], Z[k0+3]
        const size_t mb = (M - k0 - 4) / 2;
        const float4 g0 = reinterpret_cast<const float4 *>(Zf)[mb];      // Z[M-k0-4], Z[M-k0-3]
        const float4 g1 = reinterpret_cast<const float4 *>(Zf)[mb + 1];  // Z[M-k0-2], Z[M-k0-1]
        // Z[k0+4] is the next lane's Z[k0] and Z[M-k0] the previous lane's Z[M-k0'-4]: two lane
        // shifts instead of two more 8-byte loads per lane (only the edge lanes load)
        const int lane = threadIdx.x & 63;
        cf zx = make_float2(__shfl_down(f0.x, 1, 64), __shfl_down(f0.y, 1, 64));
        cf m0 = make_float2(__shfl_up(g0.x, 1, 64), __shfl_up(g0.y, 1, 64));
        if (lane == 63 || j + 1 >= M / 8) zx = Zf[k0 + 4];  // k0+4 <= M/2 (the next lane may be past the end)
        if (lane == 0) m0 = Zf[(M - k0) & (M - 1)];
        // fwd[i] = Z[k0+i], mir[i] = Z[M-k0-i], i = 0..4
        const cf fwd[5] = {make_float2(f0.x, f0.y), make_float2(f0.z, f0.w), make_float2(f1.x, f1.y),
                           make_float2(f1.z, f1.w), zx};
        const cf mir[5] = {m0, make_float2(g1.z, g1.w), make_float2(g1.x, g1.y), make_float2(g0.z, g0.w),
                           make_float2(g0.x, g0.y)};
        const unsigned Bm = (1u << a.log2B) - 1u;
        cf xf[4], xm[4];  // xm[i] = X[M-k0-4+i]
        float pf[4], pm[4];
        // W_N^{k0} from the two-level table, W_N^{k0+i} by four multiplications with W_N^1
        cf w = cmul(a.TA[k0 >> a.log2B], a.TB[k0 & Bm]);
        const cf w1 = a.TB[1];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i > 0) w = cmul(w, w1);
            if (i < 4) {
                cf x = untangle_one(fwd[i], mir[i], w);
                x.x *= a.inv_n;
                x.y *= a.inv_n;
                xf[i] = x;
                pf[i] = fmaf(x.x, x.x, x.y * x.y);
            }
            if (i > 0) {  // bin M-k0-i lands at slot 4-i of the mirrored group
                cf x = untangle_mirror(fwd[i], mir[i], w);
                x.x *= a.inv_n;
                x.y *= a.inv_n;
                xm[4 - i] = x;
                pm[4 - i] = fmaf(x.x, x.x, x.y * x.y);
            }
        }
        if (j == 0) {
            // bin N/2 is never normalised by the reference (src/fft_impl.cpp:156-160
            // visits k < N/2 only); X[N/2] = Re Z[0] - Im Z[0]
            Xf[M] = make_float2(fwd[0].x - fwd[0].y, 0.f);
        }
        reinterpret_cast<float4 *>(Xf)[2 * j] = make_float4(xf[0].x, xf[0].y, xf[1].x, xf[1].y);
        reinterpret_cast<float4 *>(Xf)[2 * j + 1] = make_float4(xf[2].x, xf[2].y, xf[3].x, xf[3].y);
        reinterpret_cast<float4 *>(Xf)[mb] = make_float4(xm[0].x, xm[0].y, xm[1].x, xm[1].y);
        reinterpret_cast<float4 *>(Xf)[mb + 1] = make_float4(xm[2].x, xm[2].y, xm[3].x, xm[3].y);
        if (0 < a.nlevels) {
            reinterpret_cast<unsigned *>(Qf)[gF] = pack4(pf[0], pf[1], pf[2], pf[3], a.size_log2);
            reinterpret_cast<unsigned *>(Qf)[gM] = pack4(pm[0], pm[1], pm[2], pm[3], a.size_log2);
        }
        const float f01 = __fadd_rn(pf[0], pf[1]), f23 = __fadd_rn(pf[2], pf[3]);
        const float m01 = __fadd_rn(pm[0], pm[1]), m23 = __fadd_rn(pm[2], pm[3]);
        if (1 < a.nlevels) {
            unsigned short *q1 = reinterpret_cast<unsigned short *>(Qf + M);
            q1[gF] = (unsigned short)(quantize_u8(f01, a.size_log2 - 1) | (quantize_u8(f23, a.size_log2 - 1) << 8));
            q1[gM] = (unsigned short)(quantize_u8(m01, a.size_log2 - 1) | (quantize_u8(m23, a.size_log2 - 1) << 8));
        }
        sF = __fadd_rn(f01, f23);
        sM = __fadd_rn(m01, m23);
        if (2 < a.nlevels) {
            Qf[M + M / 2 + gF] = (int8_t)quantize_u8(sF, a.size_log2 - 2);
            Qf[M + M / 2 + gM] = (int8_t)quantize_u8(sM, a.size_log2 - 2);
        }
    }
    // lanes hold level-2 values of adjacent groups: levels 3..8 across the wave
    const float topF = wave_pyramid(sF, gF, 2, a.nlevels, a.size_log2, Qf, M, valid);
    const float topM = wave_pyramid(sM, gM, 2, a.nlevels, a.size_log2, Qf, M, valid);
    if (valid && (threadIdx.x & 63) == 0) {
        a.Pscr[(size_t)f * a.p_stride + (gF >> 6)] = topF;
        a.Pscr[(size_t)f * a.p_stride + (gM >> 6)] = topM;
    }
}

// ---- fused real-input path (k_fft_pass2_real, fft_pass.h) -------------------------------------
// Completes the high octets a segment's first tile could not finish: element 0 of the octet comes
// from the carry-out of the segment above (the LAST segment's from tile 0: row M1/2), elements
// 1..7 from the tile's own partial rows.  grid = the segments; those whose carried row arrived inside the launch
// (hand-off plans: fft_pass.h) have nothing left to do.
struct SeamArgs {
    const float *seamP;  // [seam segments][L][8]: elements 1..7 of the octet at [0..7)
    const float *seamC;  // [segments][L]
    const uint4 *segtab; // Pass2Args::segtab
    const unsigned *segmark;  // hand-off plans: == epoch for a segment whose carry-in was not in memory in time (else nullptr)
    unsigned epoch;
    int L;               // row length (M2)
    int cp;              // couples per tile: 8 (octets, records of 16 bytes) or 4 (quartets, 8 bytes)
    int size_log2;
    int8_t *Qt;
    size_t qt_stride;
    float *Pscr;
    size_t p_stride;
};
__global__ __launch_bounds__(256) void k_real_seam(SeamArgs a) {
    const uint4 se = a.segtab[blockIdx.x];
    if ((se.w & 1u) && a.segmark[blockIdx.x] != a.epoch) return;  // (PSDR_SEG_CARRY_MEM) the carried row arrived inside the launch
    const int f = (int)se.x;
    const int g = (int)(se.y & 0xFFFFu);  // the segment's first tile
    const float *P = a.seamP + (size_t)blockIdx.x * a.L * a.cp;
    const float *Cc = a.seamC + (size_t)se.z * a.L;  // carry-out of the segment above (the top segment: row M1/2 from tile 0)
    int8_t *Qf = a.Qt + (size_t)f * a.qt_stride;
    float *Pf = a.Pscr + (size_t)f * a.p_stride;
    if (a.cp == 4) {  // quartets (2048-point rows)
        for (int c = threadIdx.x; c < a.L; c += blockDim.x) {
            const float4 v = reinterpret_cast<const float4 *>(P)[c];
            float pw[4] = {Cc[c], v.x, v.y, v.z};  // elements 1..3 at [0..3)
            const size_t rp = ((size_t)g * a.L + c) * 2 + 1;  // RecMap::pair: the HIGH quartet of column c of tile g
            uint2 rec;
            pyr_record4(pw, a.size_log2, rec);
            *reinterpret_cast<uint2 *>(Qf + rp * 8) = rec;
            Pf[rp] = pw[0];
        }
        return;
    }
    for (int c = threadIdx.x; c < a.L; c += blockDim.x) {
        const float4 v0 = reinterpret_cast<const float4 *>(P)[2 * c], v1 = reinterpret_cast<const float4 *>(P)[2 * c + 1];
        float pw[8] = {Cc[c], v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z};  // elements 1..7 at [0..7)
        const size_t rp = ((size_t)g * 2 + 1) * a.L + c;
        uint4 rec;
        pyr_record8(pw, a.size_log2, rec);
        *reinterpret_cast<uint4 *>(Qf + rp * 16) = rec;
        Pf[rp] = pw[0];
    }
}

// device layout (SpecLayout) -> the reference's k order, one frame: `nbins` bins (+ the un-normalised
// bin N/2 of real input, kept after the M laid-out bins).  IQ: bin k is client-order bin
// c = (k - N/2 - 1) mod N (src/fft_impl.cpp:149-160).
__global__ __launch_bounds__(256) void k_spec_k_order(const cf *X, cf *out, size_t M, int is_real, SpecLayout lay) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (is_real) {
        if (k > M) return;
        out[k] = k == M ? X[M] : X[lay.pos((int)k)];
    } else {
        if (k >= M) return;
        out[k] = X[lay.pos((int)((k + M - (M / 2 + 1)) & (M - 1)))];
    }
}

// band sharding (SURVEY 8e variant ii): bins [first, first + count) of every frame, in the order the
// clients index them (IQ: client order, real: k order), out of the device layout into a linear
// buffer - what one rank's clients need of the spectrum.  Indices wrap at R (the last band's halo).
__global__ __launch_bounds__(256) void k_band_pack(const cf *X, size_t spec_stride, SpecLayout lay, int R, int first,
                                                   int count, cf *out, size_t out_stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    int k = first + j;
    if (k >= R) k -= R;
    out[(size_t)blockIdx.y * out_stride + j] = X[(size_t)blockIdx.y * spec_stride + lay.pos(k)];
}

// banded spectrum (SpecLayout mode 3): lines Lb .. Lb+H-1 of every tile of band b = lines 0 .. H-1 of the same tile of
// band b+1 (the last band gets the spectrum's first columns: never read, windows do not wrap).  One thread per 16 bytes.
__global__ __launch_bounds__(256) void k_band_halo(cf *X, size_t spec_stride, SpecLayout lay, int nbands, int nframes,
                                                   int tiles, int H) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int piece = (int)(q & 7);
    size_t r = q >> 3;
    const int h = (int)(r % (size_t)H);
    r /= (size_t)H;
    const int tl = (int)(r % (size_t)tiles);
    r /= (size_t)tiles;
    const int f = (int)(r % (size_t)nframes);
    const size_t b = r / (size_t)nframes;
    if (b >= (size_t)nbands) return;
    const int Lb = 1 << lay.l2Lb;
    const size_t bn = b + 1 == (size_t)nbands ? 0 : b + 1;
    const float4 *src = reinterpret_cast<const float4 *>(X + bn * lay.band_stride + (size_t)f * spec_stride + ((size_t)tl * lay.Lw + h) * 16);
    float4 *dst = reinterpret_cast<float4 *>(X + b * lay.band_stride + (size_t)f * spec_stride + ((size_t)tl * lay.Lw + Lb + h) * 16);
    dst[piece] = src[piece];
}


// one work-group row per (client, sent frame): copies q_level[l..r).  Levels <= tiled_lt live
// in the tiled records (quantize.h), the upper levels in the level-major buffer.
__global__ __launch_bounds__(256) void k_waterfall_gather(const int8_t *Q, size_t q_stride,
                                                          const int8_t *Qt, size_t qt_stride, int tiled_lt,
                                                          int ch, RecMap map, const WfClient *cl,
                                                          const int *sent_frames, int nsent, int8_t *out) {
    const WfClient c = cl[blockIdx.x];
    if (!c.active) return;
    const int si = blockIdx.y;
    if (si >= nsent) return;
    const int f = sent_frames[si];
    const int len = c.r - c.l;
    int8_t *dst = out + c.out_off + (size_t)si * len;
    if (c.level <= tiled_lt) {
        const int8_t *src = Qt + (size_t)f * qt_stride;
        const int per = ch >> c.level;  // values of this level per record
        const int loff = tiled_level_offset(ch, c.level);
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const size_t j = (size_t)c.l + i;
            dst[i] = src[map.pos(j / per) * (2 * ch) + loff + (j % per)];
        }
    } else {
        const int8_t *src = Q + (size_t)f * q_stride + c.qoff + c.l;
        for (int i = threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
    }
}

// tiled records -> the reference's level-major layout (levels 0..tiled_lt of one frame)
__global__ __launch_bounds__(256) void k_untile_q(const int8_t *Qt, int8_t *Q, size_t R, int ch, int tiled_lt,
                                                  int nlevels, RecMap map) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // record
    if (g >= R / ch) return;
    const int8_t *rec = Qt + map.pos(g) * (2 * ch);
    size_t qoff = 0;
    for (int lv = 0; lv <= tiled_lt && lv < nlevels; lv++) {
        const int per = ch >> lv;
        const int loff = tiled_level_offset(ch, lv);
        for (int i = 0; i < per; i++) Q[qoff + g * per + i] = rec[loff + i];
        qoff += R >> lv;
    }
}

}  // namespace psdr
