// oracle/ref_core_exports.cpp - extern "C" doorways into MORE of the reference's own compiled code, for an image that
// carries what these translation units need.  TEST INFRASTRUCTURE ONLY; contains no DSP.
//
// Built by `make -C oracle ref_core` ONLY when the compiler finds a genuine <fftw3.h>, <boost/circular_buffer.hpp> and a
// linkable libfftw3f (+ libfftw3f_threads): then /root/reference/src/fft_impl.cpp (class FFTW: window multiply, forward
// transform, /N, power_and_quantize, half_and_quantize - src/fft_impl.cpp:14-61, 80-183) and samplereader.cpp
// (convert<T> - src/samplereader.cpp:29-70) are compiled unmodified and in place into oracle/_ref/libref_core.so, and
// tests/test_oracle_ref_core.py pins the oracle's quantiser / pyramid / index rotation / sample conversion to them bit for
// bit.  No stand-in header is ever written for this: without the real dependencies the target prints why it did nothing
// and those parts of the oracle stay "parity unpinned" (psdr_oracle.h).  (In the round-5 image: no fftw3.h, no boost.)
#include <cstdint>
#include <cstring>
#include <memory>

#include "fft.h"
#include "samplereader.h"

namespace {
// a SampleReader over a memory block (the reference reads stdin / a file: src/samplereader.cpp:9-16)
class MemReader : public SampleReader {
  public:
    MemReader(const uint8_t *p, size_t n) : p{p}, n{n} {}
    int read(void *arr, int num) override {
        const size_t k = (size_t)num < n ? (size_t)num : n;
        memcpy(arr, p, k);
        p += k;
        n -= k;
        return (int)k;
    }

  private:
    const uint8_t *p;
    size_t n;
};
template <typename T> void convert_with(const void *raw, float *out, int num) {
    SampleConverter<T> c(std::make_unique<MemReader>((const uint8_t *)raw, sizeof(T) * (size_t)num));
    c.read(out, num);
}
}  // namespace

extern "C" {
// class FFTW exactly as spectrumserver.cpp sets it up (src/spectrumserver.cpp:192-213: set_output_additional_size, then
// plan_c2c(FORWARD) / plan_r2c with FFTW_MEASURE | FFTW_DESTROY_INPUT; ESTIMATE here: the plan changes the summation order
// of the transform, not the code under test)
void *refc_fft_create(size_t size, int is_real, int nthreads, int downsample_levels, int brightness_offset, int additional_size) {
    FFTW *f = new FFTW(size, nthreads, downsample_levels, brightness_offset);
    f->set_output_additional_size((size_t)additional_size);
    if (is_real)
        f->plan_r2c(FFTW_ESTIMATE);
    else
        f->plan_c2c(FFT::FORWARD, FFTW_ESTIMATE);
    return f;
}
void refc_fft_destroy(void *h) { delete (FFTW *)h; }
// a1, a2: the two half-frames (src/fft.cpp:69-78); runs load_*_input + execute
int refc_fft_execute(void *h, int is_real, float *a1, float *a2) {
    FFTW *f = (FFTW *)h;
    const int rc = is_real ? f->load_real_input(a1, a2) : f->load_complex_input(a1, a2);
    return rc ? rc : f->execute();
}
float *refc_fft_output(void *h) { return ((FFTW *)h)->get_output_buffer(); }
int8_t *refc_fft_quantized(void *h) { return ((FFTW *)h)->get_quantized_buffer(); }

// SampleConverter<T>::read over `num` values of format fmt (psdr_format order: u8 s8 u16 s16 f32 f64);
// out must hold `num` floats (the converter uses the tail of `out` as its scratch: src/samplereader.cpp:42-50)
int refc_convert(int fmt, const void *raw, float *out, int num) {
    switch (fmt) {
    case 0: convert_with<uint8_t>(raw, out, num); return 0;
    case 1: convert_with<int8_t>(raw, out, num); return 0;
    case 2: convert_with<uint16_t>(raw, out, num); return 0;
    case 3: convert_with<int16_t>(raw, out, num); return 0;
    case 4: convert_with<float>(raw, out, num); return 0;
    case 5: convert_with<double>(raw, out, num); return 0;
    }
    return -1;
}
}
