// pass2.hip - launchers of the second FFT pass (fft_pass.h: row transform + /N + |X|^2 + int8 pyramid records; real input:
// Hermitian untangle fused)
#include "ctx.h"
#include "fft_pass.h"

namespace psdr {

template <int L, int T, bool FUSED, int TWC, bool BAND = false>
static int launch_pass2_t(psdr_ctx *c, const Pass2Args &a, unsigned blocks) {
    constexpr size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass2<L, T, FUSED, TWC, BAND>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass2<L, T, FUSED, TWC, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, K_PASS2);
    const unsigned grid = persistent_grid(c, blocks, lds);
    hipLaunchKernelGGL((k_fft_pass2<L, T, FUSED, TWC, BAND>), dim3(grid), dim3(L * T / 32), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
#define P2CASE(L_, T_)                                                         \
    if (L == L_ && T == T_)                                                    \
        return fused ? launch_pass2_t<L_, T_, true, 0>(c, a, blocks)           \
                     : launch_pass2_t<L_, T_, false, 0>(c, a, blocks);
int launch_pass2(psdr_ctx *c, int L, int T, bool fused, const Pass2Args &a, unsigned blocks) {
    P2CASE(64, 64)
    P2CASE(64, 128)
    P2CASE(128, 128)
    P2CASE(256, 64)
    P2CASE(512, 32)
    if (L == 1024 && T == 16 && a.TW == 16)  // the 2^20-point transform: fill addresses fold
        return fused ? launch_pass2_t<1024, 16, true, 16>(c, a, blocks)
                     : launch_pass2_t<1024, 16, false, 16>(c, a, blocks);
    if (L == 1024 && T == 16 && a.TW == 8)  // 2^21 points (2^22 real): 2048 x 1024
        return fused ? launch_pass2_t<1024, 16, true, 8>(c, a, blocks)
                     : launch_pass2_t<1024, 16, false, 8>(c, a, blocks);
    P2CASE(1024, 16)
    P2CASE(2048, 8)
    return fail(PSDR_ERR_UNSUPPORTED, "no pass-2 kernel for L=%d T=%d", L, T);
}
// banded spectrum layout (psdr_set_band_layout): 2^20- and 2^21-point IQ frames
int launch_pass2_band(psdr_ctx *c, const Pass2Args &a, unsigned blocks) {
    return a.TW == 16 ? launch_pass2_t<1024, 16, true, 16, true>(c, a, blocks)
                      : launch_pass2_t<1024, 16, true, 8, true>(c, a, blocks);
}

// fused real-input pass 2 (TWC = pass-1 tile width: 16 for 1024 x 1024, 8 for 2048 x 1024; rows of 2048 points: 1024 x 2048,
// tiles of four couples, ONE carried-row buffer)
template <int L, int T, int TWC>
static int launch_pass2_real_t(psdr_ctx *c, const Pass2Args &a) {
    constexpr size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf) + (L == 2048 ? 1 : 2) * (size_t)L * sizeof(float);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass2_real<L, T, TWC>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass2_real<L, T, TWC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, K_PASS2);
    const unsigned grid = persistent_grid(c, a.total_slots, lds);
    hipLaunchKernelGGL((k_fft_pass2_real<L, T, TWC>), dim3(grid), dim3(L * T / 32), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
int launch_pass2_real(psdr_ctx *c, const Pass2Args &a) {
    if (c->M2 == 2048) return launch_pass2_real_t<2048, 8, 16>(c, a);
    return a.TW == 16 ? launch_pass2_real_t<1024, 16, 16>(c, a) : launch_pass2_real_t<1024, 16, 8>(c, a);
}

}  // namespace psdr
