// butterfly.h — in-register forward (sign -1) DFT butterflies of radix 2/4/8/16 for the
// LDS-tiled Stockham passes.  Natural-order in, natural-order out:
//   x[s] <- sum_q x[q] * exp(-2*pi*i*q*s/R)
// gfx950 only; f32 VALU (the compiler maps the (re, im) pairs onto v_pk_*_f32).
//
// Every butterfly is a template over the point type C:
//   cf   one complex value
//   c2   the same point of TWO adjacent sequences (columns 2p and 2p+1 of a tile).  The
//        passes run on c2: the two sequences share every address, every stage twiddle and
//        every LDS instruction (one 16-byte access carries both), which is what bounds the
//        passes (they are VALU-issue-bound, not flop-bound).
#pragma once
#include <hip/hip_runtime.h>

namespace psdr {

typedef float2 cf;
typedef float v2f __attribute__((ext_vector_type(2)));
struct c2 {
    cf a, b;
};

// The complex primitives below are single VOP3P instructions with operand swizzles
// (op_sel / op_sel_hi pick the half of each 64-bit source that feeds the low / high result
// lane, neg_lo / neg_hi negate it).  Written as asm because the compiler materialises the
// swizzles with extra v_mov / v_pk_mov (about one per complex multiply and per +-i
// rotation), and VALU issue is what bounds the FFT passes.
__device__ __forceinline__ cf from_v2f(v2f v) {
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ v2f to_v2f(cf a) { return v2f{a.x, a.y}; }
// a * b:  t = (-a.y*b.y, a.y*b.x);  r = (a.x*b.x + t.x, a.x*b.y + t.y)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(to_v2f(a)), "v"(to_v2f(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r)
        : "v"(to_v2f(a)), "v"(to_v2f(b)), "v"(t));
    return from_v2f(r);
}
// two independent products, interleaved: mul1, mul2, fma1, fma2.  A v_pk_fma_f32 right behind
// the v_pk_mul_f32 it depends on costs a wait state (the compiler puts an s_nop between two asm
// statements; ~8 % of pass 2's instructions were such nops); with the other product's
// instruction in between there is none.  (early clobbers: outputs are written while inputs are
// still to be read)
__device__ __forceinline__ void cmul_pair(cf &r1, cf a1, cf b1, cf &r2, cf a2, cf b2) {
    v2f t1, t2, o1, o2;
    asm("v_pk_mul_f32 %0, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=&v"(t1), "=&v"(t2), "=&v"(o1), "=&v"(o2)
        : "v"(to_v2f(a1)), "v"(to_v2f(b1)), "v"(to_v2f(a2)), "v"(to_v2f(b2)));
    r1 = from_v2f(o1);
    r2 = from_v2f(o2);
}
// conj(a1) * b1 and conj(a2) * b2: the same two instructions per product, the conjugation is the other sign selector
// (t = (a.y*b.y, -a.y*b.x);  r = (a.x*b.x + t.x, a.x*b.y + t.y))
__device__ __forceinline__ void cmul_cja_pair(cf &r1, cf a1, cf b1, cf &r2, cf a2, cf b2) {
    v2f t1, t2, o1, o2;
    asm("v_pk_mul_f32 %0, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=&v"(t1), "=&v"(t2), "=&v"(o1), "=&v"(o2)
        : "v"(to_v2f(a1)), "v"(to_v2f(b1)), "v"(to_v2f(a2)), "v"(to_v2f(b2)));
    r1 = from_v2f(o1);
    r2 = from_v2f(o2);
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// a + (-i)*d = (a.x + d.y, a.y - d.x)
__device__ __forceinline__ cf add_mi(cf a, cf d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(to_v2f(a)), "v"(to_v2f(d)));
    return from_v2f(r);
}
// a - (-i)*d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ cf sub_mi(cf a, cf d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(to_v2f(a)), "v"(to_v2f(d)));
    return from_v2f(r);
}
// (x * w.x, y * w.x) and (x * w.y, y * w.y): scale by one half of a packed pair
__device__ __forceinline__ cf scale_lo(cf a, v2f w) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(to_v2f(a)), "v"(w));
    return from_v2f(r);
}
__device__ __forceinline__ cf scale_hi(cf a, v2f w) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(to_v2f(a)), "v"(w));
    return from_v2f(r);
}

#define PSDR_SQRT1_2 0.70710678118654752440f
#define PSDR_C1_16 0.92387953251128675613f /* cos(pi/8) */
#define PSDR_S1_16 0.38268343236508977173f /* sin(pi/8) */

// W8^k = exp(-2 pi i k/8), k = 1, 3
__device__ __forceinline__ cf mul_w8_1(cf a) {  // (1 - i)/sqrt2 * a
    const cf t = add_mi(a, a);                  // (1 - i) a = (a.x + a.y, a.y - a.x)
    return make_float2(t.x * PSDR_SQRT1_2, t.y * PSDR_SQRT1_2);
}
__device__ __forceinline__ cf mul_w8_3(cf a) {  // (-1 - i)/sqrt2 * a = (-i)(1 - i)/sqrt2 * a
    const cf t = add_mi(a, a);
    const v2f c = {PSDR_SQRT1_2, PSDR_SQRT1_2};
    v2f r;  // (t.y*c, -t.x*c)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(to_v2f(t)), "v"(c));
    return from_v2f(r);
}

// the same operations on a column couple; w is shared by the two columns
__device__ __forceinline__ c2 cmul(c2 v, cf w) {
    c2 r;
    cmul_pair(r.a, v.a, w, r.b, v.b, w);
    return r;
}
__device__ __forceinline__ c2 cadd(c2 u, c2 v) { return c2{cadd(u.a, v.a), cadd(u.b, v.b)}; }
__device__ __forceinline__ c2 csub(c2 u, c2 v) { return c2{csub(u.a, v.a), csub(u.b, v.b)}; }
__device__ __forceinline__ c2 add_mi(c2 u, c2 v) { return c2{add_mi(u.a, v.a), add_mi(u.b, v.b)}; }
__device__ __forceinline__ c2 sub_mi(c2 u, c2 v) { return c2{sub_mi(u.a, v.a), sub_mi(u.b, v.b)}; }
__device__ __forceinline__ c2 mul_w8_1(c2 v) { return c2{mul_w8_1(v.a), mul_w8_1(v.b)}; }
__device__ __forceinline__ c2 mul_w8_3(c2 v) { return c2{mul_w8_3(v.a), mul_w8_3(v.b)}; }

template <class C>
__device__ __forceinline__ void dft2(C &a, C &b) {
    C t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
// 4-point forward DFT, natural order
template <class C>
__device__ __forceinline__ void dft4(C &x0, C &x1, C &x2, C &x3) {
    C s02 = cadd(x0, x2), d02 = csub(x0, x2);
    C s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = add_mi(d02, d13);
    x3 = sub_mi(d02, d13);
}
// the same with a pending factor (-i) on x2 (saves materialising the rotation)
template <class C>
__device__ __forceinline__ void dft4_r2(C &x0, C &x1, C &x2, C &x3) {
    C s02 = add_mi(x0, x2), d02 = sub_mi(x0, x2);
    C s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = add_mi(d02, d13);
    x3 = sub_mi(d02, d13);
}

// 8-point forward DFT: q = 2*q1 + q2 (q1<4, q2<2), s = s1 + 4*s2
template <class C>
__device__ __forceinline__ void dft8(C (&x)[8]) {
    // 4-point DFTs over q1 for q2 = 0 (even) and q2 = 1 (odd)
    dft4(x[0], x[2], x[4], x[6]);
    dft4(x[1], x[3], x[5], x[7]);
    // twiddle W8^{s1} on the odd branch
    x[3] = mul_w8_1(x[3]);
    x[7] = mul_w8_3(x[7]);  // x[5] carries a pending (-i)
    // 2-point DFTs over q2: X[s1] = e+o, X[s1+4] = e-o
    C e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    C o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    x[0] = cadd(e0, o0);
    x[4] = csub(e0, o0);
    x[1] = cadd(e1, o1);
    x[5] = csub(e1, o1);
    x[2] = add_mi(e2, o2);
    x[6] = sub_mi(e2, o2);
    x[3] = cadd(e3, o3);
    x[7] = csub(e3, o3);
}

// 16-point forward DFT: q = 4*q1 + q2, s = s1 + 4*s2
template <class C>
__device__ __forceinline__ void dft16(C (&x)[16]) {
    // step 1: for each q2, DFT4 over q1 of x[4*q1+q2]; result y[s1][q2] stored at x[4*s1+q2]
    dft4(x[0], x[4], x[8], x[12]);
    dft4(x[1], x[5], x[9], x[13]);
    dft4(x[2], x[6], x[10], x[14]);
    dft4(x[3], x[7], x[11], x[15]);
    // step 2: twiddle y[s1][q2] *= W16^{q2*s1}
    const cf w1 = make_float2(PSDR_C1_16, -PSDR_S1_16);
    const cf w3 = make_float2(PSDR_S1_16, -PSDR_C1_16);
    const cf w9 = make_float2(-PSDR_C1_16, PSDR_S1_16);
    // s1 = 1: q2 = 1,2,3 -> W^1, W^2, W^3
    x[5] = cmul(x[5], w1);
    x[6] = mul_w8_1(x[6]);
    x[7] = cmul(x[7], w3);
    // s1 = 2: W^2, W^4 = -i, W^6
    x[9] = mul_w8_1(x[9]);
    x[11] = mul_w8_3(x[11]);  // x[10] carries a pending (-i)
    // s1 = 3: W^3, W^6, W^9
    x[13] = cmul(x[13], w3);
    x[14] = mul_w8_3(x[14]);
    x[15] = cmul(x[15], w9);
    // step 3: for each s1, DFT4 over q2 of x[4*s1+q2] -> X[s1 + 4*s2] at x[4*s1+s2]
    dft4(x[0], x[1], x[2], x[3]);
    dft4(x[4], x[5], x[6], x[7]);
    dft4_r2(x[8], x[9], x[10], x[11]);
    dft4(x[12], x[13], x[14], x[15]);
    // now x[4*s1 + s2] = X[s1 + 4*s2]: transpose the 4x4 index to natural order
    C t;
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

template <int R, class C>
__device__ __forceinline__ void dftR(C (&x)[R]) {
    if constexpr (R == 2)
        dft2(x[0], x[1]);
    else if constexpr (R == 4)
        dft4(x[0], x[1], x[2], x[3]);
    else if constexpr (R == 8)
        dft8(x);
    else
        dft16(x);
}

}  // namespace psdr
