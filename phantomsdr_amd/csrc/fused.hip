// fused.hip - launcher of k_fft_fused (fft_pass.h): both passes of an IQ transform in ONE persistent launch whose
// work-groups split into a pass-1 and a pass-2 role, Y a ring of a few frames that stays in the Infinity Cache.
#include "ctx.h"
#include "fft_pass.h"

namespace psdr {

template <int L1, int T1, int SB, int L2, int T2, int TWC>
static int launch_fused_t(psdr_ctx *c, const Pass1Args &a1, const Pass2Args &a2) {
    // the larger of the two roles' needs: pass 1 = tile + W_L + the second twiddle factor (M2 entries), pass 2 = tile + W_L
    const size_t lds1 = (size_t)L1 * T1 * sizeof(cf) + (size_t)L1 * sizeof(cf) + (size_t)a1.M2 * sizeof(cf);
    const size_t lds2 = (size_t)L2 * T2 * sizeof(cf) + (size_t)L2 * sizeof(cf);
    const size_t lds = std::max(lds1, lds2);
    const void *fn = (const void *)k_fft_fused<L1, T1, SB, L2, T2, TWC>;
    if (c->lds_attr_done.insert(fn).second) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    ProfScope ps(c, K_FUSED);
    const unsigned grid = ((unsigned)c->num_cus) & ~7u;  // one work-group per CU (the tile fills its LDS), whole XCD octets
    hipLaunchKernelGGL((k_fft_fused<L1, T1, SB, L2, T2, TWC>), dim3(grid), dim3(512), lds, c->stream, a1, a2);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
bool fused_supported(const psdr_ctx *c, int sb) {
    (void)sb;
    return !c->is_real && c->M2 == 1024 && c->T2 == 16 && ((c->M1 == 1024 && c->T1 == 16) || (c->M1 == 2048 && c->T1 == 8)) && c->num_cus >= 64;
}
#define FUSEDCASE(L1_, T1_, TWC_)                                                            \
    if (c->M1 == L1_ && c->T1 == T1_) {                                                      \
        if (sb == 2) return launch_fused_t<L1_, T1_, 2, 1024, 16, TWC_>(c, a1, a2);          \
        if (sb == 4) return launch_fused_t<L1_, T1_, 4, 1024, 16, TWC_>(c, a1, a2);          \
        return launch_fused_t<L1_, T1_, 8, 1024, 16, TWC_>(c, a1, a2);                       \
    }
int launch_fused(psdr_ctx *c, int sb, const Pass1Args &a1, const Pass2Args &a2) {
    FUSEDCASE(1024, 16, 16)
    FUSEDCASE(2048, 8, 8)
    return fail(PSDR_ERR_UNSUPPORTED, "no one-launch kernel for M1=%d T1=%d", c->M1, c->T1);
}

}  // namespace psdr
