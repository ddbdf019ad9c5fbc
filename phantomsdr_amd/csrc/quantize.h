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

__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d, int off) {
    return quantize_u8(a, off) | (quantize_u8(b, off) << 8) | (quantize_u8(c, off) << 16) |
           (quantize_u8(d, off) << 24);
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
            (unsigned short)(quantize_u8(p[0], off) | (quantize_u8(p[1], off) << 8));
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
// Consumers on the GPU (waterfall gather) index the records directly; the reference's
// level-major layout (src/fft_impl.cpp:162-172) is produced on demand by k_untile_q.
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
    hi.w = quantize_u8(p[0], size_log2 - 3) | (quantize_u8(p[1], size_log2 - 3) << 8);
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
    rec.w = quantize_u8(p[0], size_log2 - 2) | (quantize_u8(p[1], size_log2 - 2) << 8);
    p[0] = __fadd_rn(p[0], p[1]);
    rec.w |= quantize_u8(p[0], size_log2 - 3) << 16;
}

}  // namespace psdr
