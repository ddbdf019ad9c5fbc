"""What the built library's gfx950 code objects say about its kernels (metadata of libpsdr_hip.so itself, no GPU needed):
assumptions of the design that a compiler release could silently break."""
import os

import pytest

import codeobj

pytestmark = pytest.mark.skipif(not (os.path.exists(codeobj.SO) and os.path.exists(codeobj.READELF)),
                                reason="needs the built library and llvm-readelf")


@pytest.fixture(scope="module")
def meta():
    return codeobj.kernel_metadata()


def _find(meta, prefix):
    hits = {k: v for k, v in meta.items() if k.startswith(prefix)}
    assert hits, f"no kernel {prefix}* in the library"
    return hits


def test_recurrence_waves_of_the_post_chain_own_their_simd(meta):
    """PC_OWN_SIMD() (postchain.h) names v255 / a255 so that the kernel's allocation is a SIMD lane's whole register file:
    512 registers, nothing else fits on the SIMD.  If a compiler stops honouring the clobber the
    kernels still run - beside other waves, slower - and nothing else would notice."""
    for name in ("psdr::k_pc_ma2<true, false>", "psdr::k_pc_ma2<true, true>", "psdr::k_pc_mad<true>", "psdr::k_pc_gain<false, true>", "psdr::k_pc_gain<true, true>",
                 "psdr::k_pc_agc<true, false>", "psdr::k_pc_agc<false, false>", "psdr::k_pc_agc<true, true>", "psdr::k_pc_agc<false, true>"):
        for k, v in _find(meta, name).items():
            assert v["vgpr"] == 512 and v["agpr"] == 256, (k, v)
    # the one-kernel AGC: four such waves = a whole CU per work-group, 72 KiB of LDS, nothing spilled (its producers' ring of
    # register sets is what the two rounds of prefetch live in)
    for k, v in _find(meta, "psdr::k_pc_agc<").items():
        assert v["wg"] == 256 and v["scratch"] == 0 and v["lds"] <= 80 * 1024, (k, v)
    # the moving averages with their third wave (block maxima for that kernel): three SIMDs of a CU
    for k, v in _find(meta, "psdr::k_pc_ma2<true, true>").items():
        assert v["wg"] == 192 and v["scratch"] == 0, (k, v)
    # ... and the plain forms do NOT (they are the ones that share a CU with a pass when no CU is left free)
    for name in ("psdr::k_pc_ma2<false, false>", "psdr::k_pc_gain<false, false>", "psdr::k_pc_gain<true, false>"):
        for k, v in _find(meta, name).items():
            assert v["vgpr"] <= 256 and v["agpr"] == 0, (k, v)


def test_fft_passes_of_the_baseline_shapes_fit_two_waves_per_simd_without_scratch(meta):
    """The passes are 512-thread work-groups, one per CU: two waves per SIMD = at most 256 registers each, no accumulator
    registers, and no scratch (a spill is HBM traffic inside the hot loop)."""
    for name in ("psdr::k_fft_pass1<1024, 16, 4, false, 8>", "psdr::k_fft_pass1<1024, 16, 4, true, 8>", "psdr::k_fft_pass1<1024, 16, 4, true, 4>",
                 "psdr::k_fft_pass2<1024, 16, true, 16, false>", "psdr::k_fft_pass2_real<1024, 16, 16>", "psdr::k_fft_pass2_real<2048, 8, 16>"):
        for k, v in _find(meta, name).items():
            assert v["vgpr"] <= 256 and v["agpr"] == 0 and v["scratch"] == 0 and v["wg"] == 512, (k, v)


def test_column_tail_fits_beside_a_pass(meta):
    """k_col_tail runs beside the next batch's first pass: at most the 96 registers a 208-register pass work-group leaves per
    SIMD lane (a run-time `pair` branch instead of the template parameter cost 16 registers in round 6 - and the co-residency)."""
    for k, v in _find(meta, "psdr::k_col_tail<").items():
        assert v["vgpr"] <= 96 and v["scratch"] == 0, (k, v)


def test_demodulation_chain_kernels_fit_beside_a_pass(meta):
    """k_demod_chain_fixed runs in the wave slots a pass's work-group leaves on its CU (2 x 256 registers of 512 per SIMD lane
    are taken): at most 128 registers, no scratch."""
    for k, v in _find(meta, "psdr::k_demod_chain_fixed<").items():
        assert v["vgpr"] <= 128 and v["scratch"] == 0, (k, v)
    # n = 360 (the 12 kHz audio of every BASELINE shape): the 80 registers per SIMD lane that cfg2's second pass and the PAIR first
    # passes leave - allocated in eights, so 81 would be 88 and the kernel would wait for a CU without a pass
    for k, v in _find(meta, "psdr::k_demod_chain_fixed<360").items():
        assert v["vgpr"] <= 80, (k, v)
