// butterfly.h — in-register forward (sign -1) DFT butterflies of radix 2/4/8/16 for the
// LDS-tiled Stockham passes.  Natural-order in, natural-order out:
//   x[s] <- sum_q x[q] * exp(-2*pi*i*q*s/R)
// gfx950 only; plain f32 VALU (the forward FFT is HBM-bound, no MFMA).
#pragma once
#include <hip/hip_runtime.h>

namespace psdr {

typedef float2 cf;

__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (-i)
__device__ __forceinline__ cf mul_mi(cf a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void dft2(cf &a, cf &b) {
    cf t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
// 4-point forward DFT, natural order
__device__ __forceinline__ void dft4(cf &x0, cf &x1, cf &x2, cf &x3) {
    cf s02 = cadd(x0, x2), d02 = csub(x0, x2);
    cf s13 = cadd(x1, x3), d13 = mul_mi(csub(x1, x3));
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = cadd(d02, d13);
    x3 = csub(d02, d13);
}

#define PSDR_SQRT1_2 0.70710678118654752440f
#define PSDR_C1_16 0.92387953251128675613f /* cos(pi/8) */
#define PSDR_S1_16 0.38268343236508977173f /* sin(pi/8) */

// W8^k = exp(-2 pi i k/8), k = 1,2,3
__device__ __forceinline__ cf mul_w8_1(cf a) {  // (1 - i)/sqrt2
    return make_float2((a.x + a.y) * PSDR_SQRT1_2, (a.y - a.x) * PSDR_SQRT1_2);
}
__device__ __forceinline__ cf mul_w8_3(cf a) {  // (-1 - i)/sqrt2
    return make_float2((a.y - a.x) * PSDR_SQRT1_2, -(a.x + a.y) * PSDR_SQRT1_2);
}

// 8-point forward DFT: q = 2*q1 + q2 (q1<4, q2<2), s = s1 + 4*s2
__device__ __forceinline__ void dft8(cf (&x)[8]) {
    // 4-point DFTs over q1 for q2 = 0 (even) and q2 = 1 (odd)
    dft4(x[0], x[2], x[4], x[6]);  // y[s1][0] in x[0],x[2],x[4],x[6]
    dft4(x[1], x[3], x[5], x[7]);  // y[s1][1] in x[1],x[3],x[5],x[7]
    // twiddle W8^{s1} on the odd branch
    x[3] = mul_w8_1(x[3]);
    x[5] = mul_mi(x[5]);
    x[7] = mul_w8_3(x[7]);
    // 2-point DFTs over q2: X[s1] = e+o, X[s1+4] = e-o
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    cf o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    x[0] = cadd(e0, o0);
    x[4] = csub(e0, o0);
    x[1] = cadd(e1, o1);
    x[5] = csub(e1, o1);
    x[2] = cadd(e2, o2);
    x[6] = csub(e2, o2);
    x[3] = cadd(e3, o3);
    x[7] = csub(e3, o3);
}

// 16-point forward DFT: q = 4*q1 + q2, s = s1 + 4*s2
__device__ __forceinline__ void dft16(cf (&x)[16]) {
    // step 1: for each q2, DFT4 over q1 of x[4*q1+q2]; result y[s1][q2] stored at x[4*s1+q2]
    dft4(x[0], x[4], x[8], x[12]);
    dft4(x[1], x[5], x[9], x[13]);
    dft4(x[2], x[6], x[10], x[14]);
    dft4(x[3], x[7], x[11], x[15]);
    // step 2: twiddle y[s1][q2] *= W16^{q2*s1}
    const cf w1 = make_float2(PSDR_C1_16, -PSDR_S1_16);
    const cf w2 = make_float2(PSDR_SQRT1_2, -PSDR_SQRT1_2);
    const cf w3 = make_float2(PSDR_S1_16, -PSDR_C1_16);
    const cf w6 = make_float2(-PSDR_SQRT1_2, -PSDR_SQRT1_2);
    const cf w9 = make_float2(-PSDR_C1_16, PSDR_S1_16);
    // s1 = 1: q2 = 1,2,3 -> W^1, W^2, W^3
    x[5] = cmul(x[5], w1);
    x[6] = mul_w8_1(x[6]);
    x[7] = cmul(x[7], w3);
    // s1 = 2: W^2, W^4 = -i, W^6
    x[9] = mul_w8_1(x[9]);
    x[10] = mul_mi(x[10]);
    x[11] = mul_w8_3(x[11]);
    // s1 = 3: W^3, W^6, W^9
    x[13] = cmul(x[13], w3);
    x[14] = mul_w8_3(x[14]);
    x[15] = cmul(x[15], w9);
    (void)w2;
    (void)w6;
    // step 3: for each s1, DFT4 over q2 of x[4*s1+q2] -> X[s1 + 4*s2] at x[4*s1+s2]
    dft4(x[0], x[1], x[2], x[3]);
    dft4(x[4], x[5], x[6], x[7]);
    dft4(x[8], x[9], x[10], x[11]);
    dft4(x[12], x[13], x[14], x[15]);
    // now x[4*s1 + s2] = X[s1 + 4*s2]: transpose the 4x4 index to natural order
    cf t;
#define PSDR_SWAP(a, b) \
    t = x[a];           \
    x[a] = x[b];        \
    x[b] = t;
    PSDR_SWAP(1, 4)
    PSDR_SWAP(2, 8)
    PSDR_SWAP(3, 12)
    PSDR_SWAP(6, 9)
    PSDR_SWAP(7, 13)
    PSDR_SWAP(11, 14)
#undef PSDR_SWAP
}

}  // namespace psdr
