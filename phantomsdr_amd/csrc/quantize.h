// quantize.h — the waterfall int8 log-power quantiser, bit-for-bit the reference's CPU
// path: vec_log2 (src/fft_impl.cpp:14-23) and
//   q = (int8) max(-128.f, vec_log2(P, off) * 0.3010299956639812f * 20.f + 127.f)
// (src/fft_impl.cpp:40-42, 57-59), with the FMA placement GCC emits for the reference's
// own flags (-O3 -march=native, meson.build:5,14): t = fma(c2,m,c1); poly = fma(t,m,c0);
// q = fma(log*0.30103f, 20, 127).  All fusions are explicit so -ffp-contract cannot add
// or remove one.  Values above +127 are undefined in the reference (float->int8
// overflow); this build saturates at +127.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psdr {

__device__ __forceinline__ float vec_log2(float val, int power_offset) {
    unsigned bits = __float_as_uint(val);
    float log_val = __fadd_rn((float)((int)((bits >> 23) & 0xFFu) - 128), (float)power_offset);
    bits &= ~(255u << 23);
    bits += 127u << 23;
    const float m = __uint_as_float(bits);
    const float t = __fmaf_rn(-0.34484843f, m, 2.02466578f);
    const float poly = __fmaf_rn(t, m, -0.67487759f);
    return __fadd_rn(log_val, poly);
}

// returns the int8 result as an unsigned byte (two's complement), ready for packing
__device__ __forceinline__ unsigned quantize_u8(float power, int power_offset) {
    const float v = __fmul_rn(vec_log2(power, power_offset), 0.3010299956639812f);
    const float q = __fmaf_rn(v, 20.f, 127.f);
    const float c = (-128.f < q) ? q : -128.f;  // std::max(-128.f, q); NaN -> -128
    const int i = (c >= 127.f) ? 127 : (int)c;  // truncation toward zero
    return (unsigned)i & 0xFFu;
}

// ---- the same quantiser on packed pairs (the fused pass-2 epilogue is VALU-issue bound:
// ~16 instructions per value as plain C, 8 here).  Same operations in the same order and
// the same fusions as quantize_u8, two values per v_pk_*_f32; constants ride in three
// register pairs and are broadcast by op_sel.
typedef float q_v2f __attribute__((ext_vector_type(2)));
struct QConst {
    q_v2f c21, c0k, s20;  // (-0.3448.., 2.0246..), (-0.6748.., 0.30103), (20, 127)
    unsigned mmask, mone; // 0x807FFFFF (keeps the sign bit like the reference's &= ~(255<<23)), 127<<23
    unsigned magic;       // 0x4B000000 >> 9: the high word of v_alignbit that leaves 2^23's bit pattern above bits >> 23
    float lo, hi;         // -128, 127
};
__device__ __forceinline__ QConst qconst() {
    QConst k;
    k.c21 = q_v2f{-0.34484843f, 2.02466578f};
    k.c0k = q_v2f{-0.67487759f, 0.3010299956639812f};
    k.s20 = q_v2f{20.f, 127.f};
    k.mmask = 0x807FFFFFu;
    k.mone = 127u << 23;
    k.magic = 0x4B000000u >> 9;
    k.lo = -128.f;
    k.hi = 127.f;
    return k;
}
// q (still float, clamped to [-128, 127]) of two powers; koff = (float)(power_offset - 128)
__device__ __forceinline__ q_v2f quantize2(float a, float b, float koff, const QConst &k) {
    const unsigned ba = __float_as_uint(a), bb = __float_as_uint(b);
    // exponent: (float)((bits >> 23) & 255) + (power_offset - 128), exact in either order
    q_v2f lf = {(float)((ba >> 23) & 0xFFu), (float)((bb >> 23) & 0xFFu)};
    const q_v2f kk = {koff, koff};
    unsigned ma, mb;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ma) : "v"(ba), "v"(k.mmask), "v"(k.mone));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(mb) : "v"(bb), "v"(k.mmask), "v"(k.mone));
    const q_v2f m = {__uint_as_float(ma), __uint_as_float(mb)};
    q_v2f t, poly, lg, v, q;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lf) : "v"(lf), "v"(kk));
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(m), "v"(k.c21));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(poly) : "v"(t), "v"(m), "v"(k.c0k));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lg) : "v"(lf), "v"(poly));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(v) : "v"(lg), "v"(k.c0k));
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(q) : "v"(v), "v"(k.s20));
    // max(-128, q) and the saturation at +127 in one instruction (NaN -> -128 like std::max)
    float qa, qb;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(qa) : "v"(q.x), "v"(k.lo), "v"(k.hi));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(qb) : "v"(q.y), "v"(k.lo), "v"(k.hi));
    return q_v2f{qa, qb};
}
// truncation toward zero + insertion of the low byte into byte B of acc
template <int B>
__device__ __forceinline__ void q_insert(unsigned &acc, float c) {
    if constexpr (B == 0)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(c));
    else if constexpr (B == 1)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(c));
    else if constexpr (B == 2)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(c));
    else
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(c));
}

__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d, int off) {
    const QConst k = qconst();
    const float koff = (float)(off - 128);
    const q_v2f q01 = quantize2(a, b, koff, k), q23 = quantize2(c, d, koff, k);
    unsigned w = 0;
    q_insert<0>(w, q01.x);
    q_insert<1>(w, q01.y);
    q_insert<2>(w, q23.x);
    q_insert<3>(w, q23.y);
    return w;
}
// two values -> low 16 bits
__device__ __forceinline__ unsigned pack2(float a, float b, int off) {
    const QConst k = qconst();
    const q_v2f q01 = quantize2(a, b, (float)(off - 128), k);
    unsigned w = 0;
    q_insert<0>(w, q01.x);
    q_insert<1>(w, q01.y);
    return w;
}

// stores CNT (16/8/4/2/1) consecutive quantised values with one store
template <int CNT>
__device__ __forceinline__ void store_q(int8_t *dst, const float *p, int off) {
    if constexpr (CNT == 16) {
        uint4 w;
        w.x = pack4(p[0], p[1], p[2], p[3], off);
        w.y = pack4(p[4], p[5], p[6], p[7], off);
        w.z = pack4(p[8], p[9], p[10], p[11], off);
        w.w = pack4(p[12], p[13], p[14], p[15], off);
        *reinterpret_cast<uint4 *>(dst) = w;
    } else if constexpr (CNT == 8) {
        uint2 w;
        w.x = pack4(p[0], p[1], p[2], p[3], off);
        w.y = pack4(p[4], p[5], p[6], p[7], off);
        *reinterpret_cast<uint2 *>(dst) = w;
    } else if constexpr (CNT == 4) {
        *reinterpret_cast<unsigned *>(dst) = pack4(p[0], p[1], p[2], p[3], off);
    } else if constexpr (CNT == 2) {
        *reinterpret_cast<unsigned short *>(dst) =
            (unsigned short)pack2(p[0], p[1], off);
    } else {
        *dst = (int8_t)quantize_u8(p[0], off);
    }
}

// levels LV.. of the pair-sum pyramid on one aligned group of CH powers held in registers:
//   P_i[j] = P_{i-1}[2j] + P_{i-1}[2j+1]; q_i[j] = Q(P_i[j], size_log2 - i)
// (half_and_quantize, src/fft_impl.cpp:45-61; level i lives at byte offset
// sum_{t<i} R>>t of the frame's int8 buffer, src/fft_impl.cpp:162-172).
template <int CH, int LV>
__device__ __forceinline__ void pyr_levels(float (&p)[CH], int8_t *Qf, size_t qoff, size_t len,
                                           size_t cidx, int nlevels, int size_log2) {
    constexpr int CNT = CH >> LV;
    if constexpr (LV > 0) {
#pragma unroll
        for (int i = 0; i < CNT; i++) p[i] = __fadd_rn(p[2 * i], p[2 * i + 1]);
    }
    if (LV < nlevels) store_q<CNT>(Qf + qoff + cidx, p, size_log2 - LV);
    if constexpr (CNT > 1)
        pyr_levels<CH, LV + 1>(p, Qf, qoff + len, len >> 1, cidx >> 1, nlevels, size_log2);
}

// ---- tiled pyramid records (levels 0..LT of one aligned group of CH bins, contiguous) ----
// The fused pass-2 epilogue owns CH = 16 (or 8) consecutive client-order bins per LDS row;
// writing every level to its own array costs LT+1 scattered partial-line stores per row.
// Instead the device keeps levels 0..LT of group g = c / CH in ONE record of 2*CH bytes:
//   CH = 16: [q0 x16 | q1 x8 | q2 x4 | q3 x2 | q4 | pad]     (offsets 0,16,24,28,30)
//   CH =  8: [q0 x8 | q1 x4 | q2 x2 | q3 | pad]              (offsets 0, 8,12,14)
//   CH =  4: [q0 x4 | q1 x2 | q2 | pad]                      (offsets 0, 4, 6; the fused real pass with 2048-point rows)
// Consumers on the GPU (waterfall gather) index the records directly; the reference's
// level-major layout (src/fft_impl.cpp:162-172) is produced on demand by k_untile_q.
// Record ORDER: tile-major.  The pass-2 work-group that owns rows c1base..c1base+T-1 writes
// the records of all its L output rows back to back (one contiguous stream of full lines
// instead of one 32-byte piece per 2 KiB: the piecewise order cost 16 % of pass 2).  Group
// g = c / CH of client-order bin c sits in record
//   pos(g) = tl * (rows * gpt) + row * gpt + k,  row = g / tpr, tl = (g % tpr) / gpt, k = g % gpt
// (tpr = M1/CH groups per output row, gpt = T/CH groups per tile and row, rows = M2).
struct RecMap {
    int l2tpr, l2gpt, l2rows;
    int pair;    // mode 2, quartets (2048-point rows): a column's LOW and HIGH quartet records are adjacent,
                 // pos = ((g * rows) + c2) * 2 + side - one 16-byte store writes both (round 5: [tile][side][column] like the
                 // octets, 8-byte record stores and 4-byte sum stores: +48 % store instructions against the IQ pass)
    int mapped;  // 0: identity (level-major producers: the three-pass real-input path)
                 // 1: IQ tile-major (above)
                 // 2: fused real-input pass 2 (k_fft_pass2_real): octet (or quartet: CH = the couples of a tile) o = k / CH
                 //    of bin k lives in row c2 = o / tpr (tpr = M1/CH groups per row); the lower half of a row's octets
                 //    belongs to tile g = o % tpr as its LOW octet, the upper half to tile
                 //    g = tpr - 1 - o % tpr as its HIGH octet: pos = (g * 2 + side) * rows + c2 (a tile's low octets,
                 //    then its high octets: a wave of the octet loop stores 64 adjacent records of ONE side)
    __host__ __device__ __forceinline__ size_t pos(size_t g) const {
        if (!mapped) return g;
        const size_t row = g >> l2tpr, gc = g & (((size_t)1 << l2tpr) - 1);
        if (mapped == 2) {
            const size_t tpr = (size_t)1 << l2tpr;
            const size_t side = gc >= (tpr >> 1) ? 1 : 0;
            const size_t tl = side ? tpr - 1 - gc : gc;
            if (pair) return (((tl << l2rows) + row) << 1) + side;
            return (((tl << 1) + side) << l2rows) + row;
        }
        const size_t tl = gc >> l2gpt, k = gc & (((size_t)1 << l2gpt) - 1);
        return (((tl << l2rows) + row) << l2gpt) + k;
    }
};

// Device layout of the spectrum of one frame: where bin c (IQ: client order, real: k order) sits.
// The passes whose row transform has 1024 points keep the spectrum in 128-byte LINES of 16 bins,
// TILE-MAJOR: a pass-2 work-group owns 16 rows c1 of the (c1, c2) output grid (bin = c1 + M1*c2) and
// writes the 1024 lines of its tile as one contiguous 128 KiB block (a wave's store instruction
// covers 8 adjacent lines).  In the natural order the same lines are 8 KiB apart and their
// neighbours belong to tiles written by other work-groups at other times; measured on the fused
// real path that cost twice the whole rest of the tile.
//   mode 0  natural order
//   mode 1  IQ, tiles of 16 adjacent rows: line (tl, c2) = tl * L + c2 = rows 16tl..16tl+15 of column c2
//   mode 2  fused real input (k_fft_pass2_real, fft_pass.h): line (g, c) = g * L + c =
//           [ rows 8g..8g+7 of column c | the mirror octet of tile g of column L-1-c ]; the mirror
//           octet of tile g is rows M1-8g-7..M1-8g ascending (g = 0: its last element is row M1/2
//           instead of "row M1").  The halves of a line are the two real-signal bins a
//           (row, mirror row) couple produces together.  With 2048-point rows (2^22-point real frames
//           split 1024 x 2048) a tile holds CP = 4 couples: lines of 8 bins, [ rows 4g..4g+3 | mirror
//           quartet ] (l2cp = 2; 3 for the octets above).
//   mode 3  IQ, BANDED (band sharding, psdr_set_band_layout): the columns are split into 2^(l2L - l2Lb) bands of
//           Lb = 2^l2Lb columns; band b owns one region of the buffer (band_stride bins apart) that holds ALL frames of
//           the batch, each frame as tiles of Lw = Lb + halo lines: line (tl, c2) of mode 1 is line tl * Lw + (c2 & (Lb-1))
//           of the frame's slot in region c2 >> l2Lb; lines Lb .. Lw-1 of a tile repeat the first columns of the NEXT
//           band (k_band_halo), so that a window that starts in a band can be read from that band's region alone.
//           One region is what one peer receives: no pack pass.
//   mode 4  one band region as received (psdr_demod_batch_from_band_region): columns c2_0 .. c2_0 + Lw - 1
// Consumers index through pos() (demodulation slices) or ask for k order (psdr_read_spectrum).
struct SpecLayout {
    int mode, m1, l2m1, L, l2L;
    int k0;  // mode 0 only: the buffer starts at bin k0 (a band of the spectrum, psdr_demod_batch_from_band)
    int l2Lb, Lw, c2_0;  // modes 3, 4
    size_t band_stride;  // mode 3
    int l2cp;            // mode 2: log2 of the couples per tile (3: lines of 16 bins; 2: lines of 8)
    __host__ __device__ __forceinline__ size_t pos(int k) const {
        if (!mode) return (size_t)(k - k0);
        const int c1 = k & (m1 - 1), c2 = k >> l2m1;
        if (mode == 1) return ((((size_t)(c1 >> 4) << l2L) + c2) << 4) + (c1 & 15);
        if (mode == 3)
            return (size_t)(c2 >> l2Lb) * band_stride + ((((size_t)(c1 >> 4) * Lw) + (c2 & ((1 << l2Lb) - 1))) << 4) + (c1 & 15);
        if (mode == 4) {
            int cl = c2 - c2_0;
            if (cl < 0) cl += L;  // the last band's halo is the spectrum's first column
            return ((((size_t)(c1 >> 4) * Lw) + cl) << 4) + (c1 & 15);
        }
        const int cp = 1 << l2cp;
        if (c1 < (m1 >> 1)) return ((((size_t)(c1 >> l2cp) << l2L) + c2) << (l2cp + 1)) + (c1 & (cp - 1));
        const int hp = c1 == (m1 >> 1) ? m1 - 1 : c1 - 1;  // rows above M1/2 shift down, M1/2 goes last
        const int g = (m1 - 1 - hp) >> l2cp;
        return ((((size_t)g << l2L) + (L - 1 - c2)) << (l2cp + 1)) + cp + (hp & (cp - 1));
    }
};

__host__ __device__ __forceinline__ int tiled_level_offset(int ch, int lv) {
    // ch * (2 - 2^(1-lv)) for lv >= 1, 0 for lv = 0
    return lv == 0 ? 0 : 2 * ch - (2 * ch >> lv);
}

__device__ __forceinline__ void pyr_record16(float (&p)[16], int size_log2, uint4 &lo, uint4 &hi) {
    lo.x = pack4(p[0], p[1], p[2], p[3], size_log2);
    lo.y = pack4(p[4], p[5], p[6], p[7], size_log2);
    lo.z = pack4(p[8], p[9], p[10], p[11], size_log2);
    lo.w = pack4(p[12], p[13], p[14], p[15], size_log2);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = __fadd_rn(p[2 * i], p[2 * i + 1]);
    hi.x = pack4(p[0], p[1], p[2], p[3], size_log2 - 1);
    hi.y = pack4(p[4], p[5], p[6], p[7], size_log2 - 1);
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = __fadd_rn(p[2 * i], p[2 * i + 1]);
    hi.z = pack4(p[0], p[1], p[2], p[3], size_log2 - 2);
    p[0] = __fadd_rn(p[0], p[1]);
    p[1] = __fadd_rn(p[2], p[3]);
    hi.w = pack2(p[0], p[1], size_log2 - 3);
    p[0] = __fadd_rn(p[0], p[1]);
    hi.w |= quantize_u8(p[0], size_log2 - 4) << 16;
}
__device__ __forceinline__ void pyr_record8(float (&p)[8], int size_log2, uint4 &rec) {
    rec.x = pack4(p[0], p[1], p[2], p[3], size_log2);
    rec.y = pack4(p[4], p[5], p[6], p[7], size_log2);
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = __fadd_rn(p[2 * i], p[2 * i + 1]);
    rec.z = pack4(p[0], p[1], p[2], p[3], size_log2 - 1);
    p[0] = __fadd_rn(p[0], p[1]);
    p[1] = __fadd_rn(p[2], p[3]);
    rec.w = pack2(p[0], p[1], size_log2 - 2);
    p[0] = __fadd_rn(p[0], p[1]);
    rec.w |= quantize_u8(p[0], size_log2 - 3) << 16;
}

__device__ __forceinline__ void pyr_record4(float (&p)[4], int size_log2, uint2 &rec) {
    rec.x = pack4(p[0], p[1], p[2], p[3], size_log2);
    p[0] = __fadd_rn(p[0], p[1]);
    p[1] = __fadd_rn(p[2], p[3]);
    rec.y = pack2(p[0], p[1], size_log2 - 1);
    p[0] = __fadd_rn(p[0], p[1]);
    rec.y |= quantize_u8(p[0], size_log2 - 2) << 16;
}

}  // namespace psdr
