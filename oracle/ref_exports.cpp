// oracle/ref_exports.cpp — extern "C" doorways into the REFERENCE's own compiled code.
// TEST INFRASTRUCTURE ONLY.  This file contains no DSP: it forwards to functions that
// are compiled, unmodified and in place, from /root/reference/src/utils/dsp.cpp and
// /root/reference/src/utils/audioprocessing.cpp (see oracle/Makefile, target _ref).
// Those two files are the only parts of the hot path that build from their own sources
// with g++ alone; everything else needs fftw3.h / boost / websocketpp (absent here).
#include <complex>
#include <cstddef>
#include <cstdint>

#include "utils/audioprocessing.h"
#include "utils/dsp.h"

extern "C" {
void ref_build_hann_window(float *arr, int num) { build_hann_window(arr, num); }
void ref_polar_discriminator_fm(float *buf, float prev_re, float prev_im, float *output,
                                size_t len) {
    polar_discriminator_fm((std::complex<float> *)buf, std::complex<float>(prev_re, prev_im),
                           output, len);
}
void ref_dsp_negate_float(float *arr, size_t len) { dsp_negate_float(arr, len); }
void ref_dsp_negate_complex(float *arr, size_t len) {
    dsp_negate_complex((std::complex<float> *)arr, len);
}
void ref_dsp_add_float(float *a, float *b, size_t len) { dsp_add_float(a, b, len); }
void ref_dsp_add_complex(float *a, float *b, size_t len) {
    dsp_add_complex((std::complex<float> *)a, (std::complex<float> *)b, len);
}
void ref_dsp_am_demod(float *arr, float *output, size_t len) {
    dsp_am_demod((std::complex<float> *)arr, output, len);
}
void ref_dsp_float_to_int16(float *arr, int32_t *output, float mult, size_t len) {
    dsp_float_to_int16(arr, output, mult, len);
}
void *ref_agc_create(float desired, float attack_ms, float release_ms, float lookahead_ms,
                     float sr) {
    return new AGC(desired, attack_ms, release_ms, lookahead_ms, sr);
}
void ref_agc_destroy(void *a) { delete (AGC *)a; }
void ref_agc_process(void *a, float *arr, size_t len) { ((AGC *)a)->process(arr, len); }
void ref_agc_reset(void *a) { ((AGC *)a)->reset(); }
}
