// pass1.hip - launchers of the first FFT pass (fft_pass.h: convert + Hann + column transform + inter-pass twiddle)
#include "ctx.h"
#include "fft_pass.h"

namespace psdr {

template <int L, int T, int SB, bool PAIR = false, int CP = 8>
static int launch_pass1_t(psdr_ctx *c, const Pass1Args &a, unsigned blocks) {
    // tile + W_L (= first twiddle factor) + second twiddle factor (M2 entries)
    const size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf) + (size_t)a.M2 * sizeof(cf);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass1<L, T, SB, PAIR, CP>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass1<L, T, SB, PAIR, CP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    ProfScope ps(c, K_PASS1, c->p1);
    // persistent: as many work-groups per CU as their LDS admits (a 128 KiB tile: one)
    const unsigned grid = persistent_grid(c, blocks, lds);
    hipLaunchKernelGGL((k_fft_pass1<L, T, SB, PAIR, CP>), dim3(grid), dim3(L * T / 32), lds, c->p1, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}

#define P1CASE(L_, T_)                                                   \
    if (L == L_ && T == T_) {                                            \
        if (sb == 2) return launch_pass1_t<L_, T_, 2>(c, a, blocks);     \
        if (sb == 4) return launch_pass1_t<L_, T_, 4>(c, a, blocks);     \
        return launch_pass1_t<L_, T_, 8>(c, a, blocks);                  \
    }
#define P1PAIR(L_, T_)                                                         \
    if (L == L_ && T == T_) {                                                  \
        if (sb == 2) return launch_pass1_t<L_, T_, 2, true>(c, a, blocks);     \
        if (sb == 4) return launch_pass1_t<L_, T_, 4, true>(c, a, blocks);     \
        return launch_pass1_t<L_, T_, 8, true>(c, a, blocks);                  \
    }
// sb: bytes per complex sample slot of the raw image (2, 4, 8); pair: the real-input form feeding k_fft_pass2_real
int launch_pass1(psdr_ctx *c, int L, int T, int sb, const Pass1Args &a, unsigned blocks, bool pair) {
    if (pair) {
        if (L == 1024 && T == 16 && a.l2t2 == 3) {  // pass-2 tiles of 8 rows = four (row, mirror row) couples: 2048-point rows (1024 x 2048)
            if (sb == 2) return launch_pass1_t<1024, 16, 2, true, 4>(c, a, blocks);
            if (sb == 4) return launch_pass1_t<1024, 16, 4, true, 4>(c, a, blocks);
            return launch_pass1_t<1024, 16, 8, true, 4>(c, a, blocks);
        }
        P1PAIR(1024, 16)
        P1PAIR(2048, 8)
        return fail(PSDR_ERR_UNSUPPORTED, "no paired pass-1 kernel for L=%d T=%d", L, T);
    }
    P1CASE(64, 64)
    P1CASE(128, 64)
    P1CASE(128, 128)
    P1CASE(256, 64)
    P1CASE(512, 32)
    P1CASE(1024, 16)
    P1CASE(1024, 8)
    P1CASE(2048, 8)
    return fail(PSDR_ERR_UNSUPPORTED, "no pass-1 kernel for L=%d T=%d", L, T);
}

}  // namespace psdr
