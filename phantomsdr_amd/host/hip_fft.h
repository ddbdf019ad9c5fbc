// hip_fft.h — `class hipFFT : public FFT`: the MI355X back-end of PhantomSDR's FFT plug-in
// interface (src/fft.h:33-63), forwarding to the C-ABI of libpsdr_hip.so (include/psdr.h).
//
// Drop this file next to the reference's src/fft.h and add the factory arm shown in
// INTEGRATION.md; nothing else in the server changes (Level 1 of the drop-in).  It follows
// the sibling back-ends' contract (FFTW src/fft_impl.cpp:80-183, cuFFT src/fft_cuda.cu):
// the constructor throws std::runtime_error when no device is present
// (src/fft_cuda.cu:8-13), plan_*/load_*/execute return 0, execute() is synchronous and
// get_output_buffer()/get_quantized_buffer() are host-readable until the next execute().
#ifndef PSDR_HIP_FFT_H
#define PSDR_HIP_FFT_H

#include <stdexcept>
#include <string>

#include "fft.h"   // the reference's class FFT (src/fft.h)
#include "psdr.h"  // include/psdr.h of this repository

class hipFFT : public FFT {
  public:
    hipFFT(size_t size, int nthreads, int downsample_levels, int brightness_offset)
        : FFT(size, nthreads, downsample_levels, brightness_offset), ctx{nullptr},
          brightness{brightness_offset} {
        // the base class built the host Hann table (src/fft_impl.cpp:63-70); the device path
        // evaluates the window itself, so the table is simply left to the base destructor
    }
    // FFT::malloc / FFT::free: pinned, host-writable (cuFFT twin: cudaHostAlloc,
    // src/fft_cuda.cu:22-28).  The reference allocates the three half-frame buffers BEFORE it
    // plans (src/fft.cpp:17-29), so allocation does not need the context.
    float *malloc(size_t nfloats) override {
        float *p = nullptr;
        if (psdr_host_alloc(ctx, nfloats, &p) != PSDR_OK) throw std::runtime_error(psdr_last_error());
        return p;
    }
    void free(float *buf) override { psdr_host_free(ctx, buf); }
    int plan_c2c(direction d, int) override {
        if (d != FORWARD) throw std::runtime_error("hipFFT plans forward transforms only");
        create(false);
        return 0;
    }
    int plan_r2c(int) override {
        create(true);
        return 0;
    }
    int load_real_input(float *a1, float *a2) override { return chk(psdr_load_real_input(ctx, a1, a2)); }
    int load_complex_input(float *a1, float *a2) override { return chk(psdr_load_complex_input(ctx, a1, a2)); }
    int execute() override {
        chk(psdr_execute(ctx));
        // refresh the base-class pointers the server reads (src/fft.cpp:30,40-41)
        chk(psdr_get_output_buffer(ctx, &outbuf));
        chk(psdr_get_quantized_buffer(ctx, &quantizedbuf));
        return 0;
    }
    float *get_output_buffer() override {
        chk(psdr_get_output_buffer(ctx, &outbuf));
        return outbuf;
    }
    int8_t *get_quantized_buffer() override {
        chk(psdr_get_quantized_buffer(ctx, &quantizedbuf));
        return quantizedbuf;
    }
    ~hipFFT() override { psdr_destroy(ctx); }

  protected:
    psdr_ctx *ctx;
    int brightness;

    // the context needs is_real, which the reference only reveals at plan time
    void create(bool is_real) {
        if (ctx) throw std::runtime_error("hipFFT: already planned");  // assert(!p), src/fft_impl.cpp:90
        psdr_config cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.fft_size = (uint32_t)size;
        cfg.is_real = is_real ? 1 : 0;
        cfg.downsample_levels = downsample_levels;
        cfg.brightness_offset = brightness;
        cfg.additional_size = additional_size;
        cfg.input_format = PSDR_FMT_F32;
        cfg.max_batch = 1;
        cfg.max_clients = 1;
        cfg.max_waterfall_clients = 1;
        cfg.skip_num = 1;
        cfg.waterfall_size = 0;  // Level 1 serves no waterfall clients itself
        if (psdr_create(&cfg, &ctx) != PSDR_OK) throw std::runtime_error(psdr_last_error());
    }
    static int chk(int rc) {
        if (rc != PSDR_OK) throw std::runtime_error(psdr_last_error());
        return 0;
    }
};

#endif
