/*
 * psdr_oracle.c — CPU ORACLE for the PhantomSDR hot path.  TEST INFRASTRUCTURE ONLY.
 * See psdr_oracle.h for the usage rule and the pinning status ("parity unpinned" for
 * the parts of fft_impl.cpp / signal.cpp / samplereader.cpp / utils.h that cannot be
 * compiled in this image).
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * behaviour it restates.  Nothing here is copied: it is written from the behaviour.
 *
 * Floating-point evaluation order: the reference is built with
 * `-O3 -march=native -std=c++23` (meson.build:5,14), i.e. GCC's default
 * -ffp-contract=fast on an FMA-capable x86 host.  GCC 11 contracts the quantiser
 * expressions as written below with explicit fmaf() (checked with
 * `g++ -O3 -march=haswell -S`); this file must therefore be compiled with
 * -ffp-contract=off so that only the explicit fmaf() calls fuse.
 */
#include "psdr_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    float re, im;
} cf32;
typedef struct {
    double re, im;
} cf64;

static int g_threads = 1;
void orc_set_threads(int n) {
    g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}
int orc_get_threads(void) { return g_threads; }

static void *xaligned(size_t bytes) {
    void *p = NULL;
    if (bytes == 0) bytes = 64;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) return NULL;
    memset(p, 0, bytes);
    return p;
}

/* ------------------------------------------------------------------------------------
 * Sample conversion — src/samplereader.cpp:29-40 (convert<T,T_signed>) and :59-65
 * (scale = (float)max(T_signed)+1 for integers, 1 for floats).  Unsigned types flip
 * the MSB and are then read as the signed type; value = (float)signed / scale.
 * ---------------------------------------------------------------------------------- */
void orc_convert(const void *raw, int fmt, size_t num, float *out) {
    switch (fmt) {
    case ORC_FMT_U8: {
        const uint8_t *p = (const uint8_t *)raw;
        for (size_t i = 0; i < num; i++) out[i] = (float)(int8_t)(p[i] ^ 0x80u) / 128.0f;
        break;
    }
    case ORC_FMT_S8: {
        const int8_t *p = (const int8_t *)raw;
        for (size_t i = 0; i < num; i++) out[i] = (float)p[i] / 128.0f;
        break;
    }
    case ORC_FMT_U16: {
        const uint16_t *p = (const uint16_t *)raw;
        for (size_t i = 0; i < num; i++) out[i] = (float)(int16_t)(p[i] ^ 0x8000u) / 32768.0f;
        break;
    }
    case ORC_FMT_S16: {
        const int16_t *p = (const int16_t *)raw;
        for (size_t i = 0; i < num; i++) out[i] = (float)p[i] / 32768.0f;
        break;
    }
    case ORC_FMT_F32: {
        const float *p = (const float *)raw;
        for (size_t i = 0; i < num; i++) out[i] = p[i] / 1.0f;
        break;
    }
    case ORC_FMT_F64: {
        const double *p = (const double *)raw;
        for (size_t i = 0; i < num; i++) out[i] = (float)p[i] / 1.0f;
        break;
    }
    default:
        break;
    }
}

/* ------------------------------------------------------------------------------------
 * Periodic Hann window — src/utils/dsp.cpp:6-11:
 *   arr[i] = 0.5 * (1 - cosf(2 * M_PI * i / num));
 * the argument is formed in double, narrowed to float for cosf; 1-cosf() is float;
 * the 0.5* is a double multiply narrowed back to float (exact).
 * ---------------------------------------------------------------------------------- */
void orc_build_hann_window(float *arr, int num) {
    for (int i = 0; i < num; i++) {
        float a = (float)(2 * M_PI * i / num);
        arr[i] = (float)(0.5 * (double)(1 - cosf(a)));
    }
}

/* ------------------------------------------------------------------------------------
 * vec_log2 — src/fft_impl.cpp:14-23; quantiser expression — :40-42 / :57-59.
 * FMA placement as GCC emits it for the reference's own flags (header comment).
 * Conversion float->int8: C truncation toward zero; the reference's behaviour above
 * +127 is undefined, the build saturates at +127.
 * ---------------------------------------------------------------------------------- */
float orc_vec_log2(float val, int power_offset) {
    uint32_t bits;
    memcpy(&bits, &val, 4);
    float log_val = (float)((int)((bits >> 23) & 0xFF) - 128) + (float)power_offset;
    bits &= ~(255u << 23);
    bits += 127u << 23;
    float m;
    memcpy(&m, &bits, 4);
    float t = fmaf(-0.34484843f, m, 2.02466578f);
    float poly = fmaf(t, m, -0.67487759f);
    log_val += poly;
    return log_val;
}
int8_t orc_quantize(float power, int power_offset) {
    float v = orc_vec_log2(power, power_offset) * 0.3010299956639812f;
    float q = fmaf(v, 20.f, 127.f);
    /* std::max(-128.f, q): returns q only if -128 < q (NaN -> -128) */
    float c = (-128.f < q) ? q : -128.f;
    if (c >= 127.f) return 127;
    return (int8_t)c;
}

/* ------------------------------------------------------------------------------------
 * DFT kernels.  FFTW 3.3.10 (subprojects/fftw3.wrap:2-5; call sites
 * src/fft_impl.cpp:98-101,113-115,145 and src/signal.cpp:65-77,138,154,214,221) is
 * not in /root/reference and not installed here.  Its contract is the exact,
 * unnormalised DFT  Y[k] = sum_j X[j] exp(sign*2*pi*i*j*k/n), natural order; this
 * restates that contract with a Stockham autosort mixed-radix transform.
 * ---------------------------------------------------------------------------------- */

/* power-of-two, single precision, radix-4 (+ one radix-2) Stockham.  W holds
 * exp(-2*pi*i*k/n), k<n, generated in double.  in is not modified. */
static void fft_pow2_f32(const cf32 *in, cf32 *out, cf32 *scratch, size_t n, int sign,
                         const cf32 *W) {
    int L = 0;
    while (((size_t)1 << L) < n) L++;
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    int n4 = L / 2, n2 = L & 1, nst = n4 + n2;
    /* choose ping-pong so that the last stage lands in out */
    const cf32 *src = in;
    cf32 *dst = (nst & 1) ? out : scratch;
    size_t p = 1;
    for (int st = 0; st < n4; st++) {
        size_t t = n / 4, tw = n / (4 * p);
        long long nb = (long long)t;
        long long i;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (n >= 16384)
        for (i = 0; i < nb; i++) {
            size_t k = (size_t)i & (p - 1);
            size_t j = (((size_t)i - k) << 2) + k;
            cf32 u0 = src[i], u1 = src[i + t], u2 = src[i + 2 * t], u3 = src[i + 3 * t];
            if (k) {
                cf32 w1 = W[k * tw], w2 = W[2 * k * tw], w3 = W[3 * k * tw];
                if (sign > 0) {
                    w1.im = -w1.im;
                    w2.im = -w2.im;
                    w3.im = -w3.im;
                }
                cf32 a;
                a.re = u1.re * w1.re - u1.im * w1.im;
                a.im = u1.re * w1.im + u1.im * w1.re;
                u1 = a;
                a.re = u2.re * w2.re - u2.im * w2.im;
                a.im = u2.re * w2.im + u2.im * w2.re;
                u2 = a;
                a.re = u3.re * w3.re - u3.im * w3.im;
                a.im = u3.re * w3.im + u3.im * w3.re;
                u3 = a;
            }
            cf32 s02 = {u0.re + u2.re, u0.im + u2.im}, d02 = {u0.re - u2.re, u0.im - u2.im};
            cf32 s13 = {u1.re + u3.re, u1.im + u3.im}, d13 = {u1.re - u3.re, u1.im - u3.im};
            cf32 v0 = {s02.re + s13.re, s02.im + s13.im};
            cf32 v2 = {s02.re - s13.re, s02.im - s13.im};
            /* forward: v1 = d02 - i*d13, v3 = d02 + i*d13 ; backward swapped */
            cf32 m = {d13.im, -d13.re}; /* -i * d13 */
            cf32 v1, v3;
            if (sign < 0) {
                v1.re = d02.re + m.re;
                v1.im = d02.im + m.im;
                v3.re = d02.re - m.re;
                v3.im = d02.im - m.im;
            } else {
                v1.re = d02.re - m.re;
                v1.im = d02.im - m.im;
                v3.re = d02.re + m.re;
                v3.im = d02.im + m.im;
            }
            dst[j] = v0;
            dst[j + p] = v1;
            dst[j + 2 * p] = v2;
            dst[j + 3 * p] = v3;
        }
        p *= 4;
        src = dst;
        dst = (dst == out) ? scratch : out;
    }
    if (n2) {
        size_t t = n / 2, tw = n / (2 * p);
        long long nb = (long long)t;
        long long i;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (n >= 16384)
        for (i = 0; i < nb; i++) {
            size_t k = (size_t)i & (p - 1);
            size_t j = (((size_t)i - k) << 1) + k;
            cf32 u0 = src[i], u1 = src[i + t];
            if (k) {
                cf32 w1 = W[k * tw];
                if (sign > 0) w1.im = -w1.im;
                cf32 a;
                a.re = u1.re * w1.re - u1.im * w1.im;
                a.im = u1.re * w1.im + u1.im * w1.re;
                u1 = a;
            }
            dst[j].re = u0.re + u1.re;
            dst[j].im = u0.im + u1.im;
            dst[j + p].re = u0.re - u1.re;
            dst[j + p].im = u0.im - u1.im;
        }
    }
}

static cf32 *make_twiddles_f32(size_t n) {
    cf32 *W = (cf32 *)xaligned(sizeof(cf32) * (n ? n : 1));
    for (size_t k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        W[k].re = (float)cos(a);
        W[k].im = (float)sin(a);
    }
    return W;
}

/* any n, double precision internally, generic-radix Stockham (each radix-R butterfly is
 * a direct R-point DFT with the inter-stage twiddle folded into one table lookup):
 *   y[j + s*p] = sum_q x[i + q*n/R] * W_n^{ q*(k + s*p)*(n/(p*R)) },  k = i mod p,
 *   j = (i-k)*R + k. */
/* plan cache (per thread): radices + twiddle table of the last few (n, sign) */
typedef struct {
    size_t n;
    int sign, ns;
    size_t radices[64];
    cf64 *W;
} gen_plan;
static _Thread_local gen_plan g_plans[8];
static _Thread_local int g_plan_next = 0;
static const gen_plan *get_plan(size_t n, int sign) {
    for (int i = 0; i < 8; i++)
        if (g_plans[i].n == n && g_plans[i].sign == sign) return &g_plans[i];
    gen_plan *p = &g_plans[g_plan_next];
    g_plan_next = (g_plan_next + 1) % 8;
    free(p->W);
    p->n = n;
    p->sign = sign;
    p->ns = 0;
    size_t m = n;
    while (m % 4 == 0) {
        p->radices[p->ns++] = 4;
        m /= 4;
    }
    while (m % 2 == 0) {
        p->radices[p->ns++] = 2;
        m /= 2;
    }
    for (size_t f = 3; f * f <= m; f += 2)
        while (m % f == 0) {
            p->radices[p->ns++] = f;
            m /= f;
        }
    if (m > 1) p->radices[p->ns++] = m;
    p->W = (cf64 *)malloc(sizeof(cf64) * n);
    for (size_t k = 0; k < n; k++) {
        double a = (double)sign * 2.0 * M_PI * (double)k / (double)n;
        p->W[k].re = cos(a);
        p->W[k].im = sin(a);
    }
    return p;
}

static void dft_generic_f64(const cf64 *in, cf64 *out, size_t n, int sign) {
    if (n == 0) return;
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    const gen_plan *pl = get_plan(n, sign);
    const size_t *radices = pl->radices;
    const int ns = pl->ns;
    const cf64 *W = pl->W;
    cf64 *bufA = (cf64 *)malloc(sizeof(cf64) * n);
    cf64 *bufB = (cf64 *)malloc(sizeof(cf64) * n);
    memcpy(bufA, in, sizeof(cf64) * n);
    cf64 *src = bufA, *dst = bufB;
    size_t p = 1;
    for (int st = 0; st < ns; st++) {
        size_t R = radices[st], t = n / R, step = n / (p * R);
        for (size_t i = 0; i < t; i++) {
            size_t k = i % p;
            size_t j = (i - k) * R + k;
            for (size_t s = 0; s < R; s++) {
                size_t e1 = ((k + s * p) * step) % n; /* exponent per unit q */
                double ar = 0, ai = 0;
                size_t e = 0;
                for (size_t q = 0; q < R; q++) {
                    cf64 x = src[i + q * t];
                    cf64 w = W[e];
                    ar += x.re * w.re - x.im * w.im;
                    ai += x.re * w.im + x.im * w.re;
                    e += e1;
                    if (e >= n) e -= n;
                }
                dst[j + s * p].re = ar;
                dst[j + s * p].im = ai;
            }
        }
        p *= R;
        cf64 *tmp = src;
        src = dst;
        dst = tmp;
    }
    memcpy(out, src, sizeof(cf64) * n);
    free(bufA);
    free(bufB);
}

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

void orc_dft_c2c(const float *in, float *out, size_t n, int sign) {
    if (is_pow2(n) && n >= 2) {
        cf32 *W = make_twiddles_f32(n);
        cf32 *scratch = (cf32 *)xaligned(sizeof(cf32) * n);
        cf32 *tmp = (cf32 *)xaligned(sizeof(cf32) * n);
        fft_pow2_f32((const cf32 *)in, tmp, scratch, n, sign, W);
        memcpy(out, tmp, sizeof(cf32) * n);
        free(W);
        free(scratch);
        free(tmp);
        return;
    }
    cf64 *a = (cf64 *)malloc(sizeof(cf64) * (n ? n : 1));
    cf64 *b = (cf64 *)malloc(sizeof(cf64) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        a[i].re = in[2 * i];
        a[i].im = in[2 * i + 1];
    }
    dft_generic_f64(a, b, n, sign);
    for (size_t i = 0; i < n; i++) {
        out[2 * i] = (float)b[i].re;
        out[2 * i + 1] = (float)b[i].im;
    }
    free(a);
    free(b);
}

/* r2c of even power-of-two length n via an n/2-point complex transform + untangle.
 * X[k] = E[k] + W_n^k O[k], E = (Z[k]+conj Z[M-k])/2, O = -i (Z[k]-conj Z[M-k])/2. */
static void r2c_pow2_f32(const float *in, cf32 *out, size_t n, const cf32 *Whalf,
                         const cf32 *Wn, cf32 *scr1, cf32 *scr2) {
    size_t M = n / 2;
    fft_pow2_f32((const cf32 *)in, scr1, scr2, M, -1, Whalf);
    const cf32 *Z = scr1;
    long long k;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (n >= 16384)
    for (k = 0; k <= (long long)M; k++) {
        cf32 a = Z[(size_t)k % M];
        cf32 b = Z[(M - (size_t)k) % M];
        b.im = -b.im;
        cf32 E = {0.5f * (a.re + b.re), 0.5f * (a.im + b.im)};
        cf32 D = {0.5f * (a.re - b.re), 0.5f * (a.im - b.im)};
        cf32 O = {D.im, -D.re}; /* -i * D */
        cf32 w = Wn[(size_t)k];
        if ((size_t)k == M) {
            w.re = -1.f;
            w.im = 0.f;
        }
        out[k].re = E.re + (w.re * O.re - w.im * O.im);
        out[k].im = E.im + (w.re * O.im + w.im * O.re);
    }
}

void orc_dft_r2c(const float *in, float *out, size_t n) {
    size_t M = n / 2;
    cf32 *Wh = make_twiddles_f32(M);
    cf32 *Wn = make_twiddles_f32(n);
    cf32 *s1 = (cf32 *)xaligned(sizeof(cf32) * (M + 1));
    cf32 *s2 = (cf32 *)xaligned(sizeof(cf32) * (M + 1));
    r2c_pow2_f32(in, (cf32 *)out, n, Wh, Wn, s1, s2);
    free(Wh);
    free(Wn);
    free(s1);
    free(s2);
}

/* c2r, any even n: FFTW's fftwf_plan_dft_c2r_1d reads bins 0..n/2 only and treats bin
 * 0 and bin n/2 as purely real (the imaginary parts are ignored):
 *   y[j] = Re A[0] + (-1)^j Re A[n/2] + 2 * sum_{k=1}^{n/2-1} Re(A[k] e^{+2 pi i jk/n}) */
void orc_dft_c2r(const float *in, float *out, size_t n) {
    cf64 *a = (cf64 *)calloc(n ? n : 1, sizeof(cf64));
    cf64 *b = (cf64 *)calloc(n ? n : 1, sizeof(cf64));
    size_t h = n / 2;
    a[0].re = in[0];
    a[0].im = 0;
    for (size_t k = 1; k < h; k++) {
        a[k].re = in[2 * k];
        a[k].im = in[2 * k + 1];
        a[n - k].re = in[2 * k];
        a[n - k].im = -(double)in[2 * k + 1];
    }
    if (h >= 1 && h < n) {
        a[h].re = in[2 * h];
        a[h].im = 0;
    }
    dft_generic_f64(a, b, n, +1);
    for (size_t j = 0; j < n; j++) out[j] = (float)b[j].re;
    free(a);
    free(b);
}

/* ------------------------------------------------------------------------------------
 * class FFTW — src/fft_impl.cpp:63-174.
 * ---------------------------------------------------------------------------------- */
struct orc_fft {
    size_t size;
    int is_real;
    int size_log2;
    int downsample_levels;
    int additional_size;
    size_t outbuf_len;
    float *windowbuf;
    float *inbuf;
    float *outbuf;
    float *powerbuf;
    int8_t *quantizedbuf;
    size_t quantized_len;
    cf32 *W, *Whalf, *scr1, *scr2;
    void *lib_plan; /* FFTW3-API plan bound to inbuf/outbuf (orc_fft_use_library), or NULL */
};

/* ------------------------------------------------------------------------------------
 * Optional FFT library for the big forward transform.  The reference's FFTW back-end calls
 * fftwf_plan_dft_1d / fftwf_plan_dft_r2c_1d + fftwf_execute (src/fft_impl.cpp:89-117,145).
 * FFTW itself is not in this image; any library that exports the FFTW3 single-precision API
 * (libfftw3f.so.3, or MKL's wrappers in libmkl_rt.so) can be dlopen()ed here so that the CPU
 * baseline of bench.py times the kind of FFT the reference would run, and so that the
 * built-in transform can be cross-checked against an independent production FFT.  Parity
 * tests use the built-in transform (deterministic, same bits everywhere).
 * ---------------------------------------------------------------------------------- */
typedef void *(*fn_plan_c2c)(int, void *, void *, int, unsigned);
typedef void *(*fn_plan_r2c)(int, float *, void *, unsigned);
typedef void (*fn_execute)(void *);
typedef void (*fn_destroy)(void *);
static struct {
    void *handle;
    fn_plan_c2c plan_c2c;
    fn_plan_r2c plan_r2c;
    fn_execute execute;
    fn_destroy destroy;
    char name[256];
} g_fftlib;
static pthread_mutex_t g_plan_mtx = PTHREAD_MUTEX_INITIALIZER; /* FFTW planners are not thread-safe */

int orc_fft_use_library(const char *path) {
    if (!path || !*path) { /* back to the built-in transform (existing plans keep theirs) */
        memset(&g_fftlib, 0, sizeof g_fftlib);
        return 0;
    }
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    fn_plan_c2c pc = (fn_plan_c2c)dlsym(h, "fftwf_plan_dft_1d");
    fn_plan_r2c pr = (fn_plan_r2c)dlsym(h, "fftwf_plan_dft_r2c_1d");
    fn_execute ex = (fn_execute)dlsym(h, "fftwf_execute");
    fn_destroy de = (fn_destroy)dlsym(h, "fftwf_destroy_plan");
    if (!pc || !pr || !ex || !de) {
        dlclose(h);
        return -2;
    }
    g_fftlib.handle = h;
    g_fftlib.plan_c2c = pc;
    g_fftlib.plan_r2c = pr;
    g_fftlib.execute = ex;
    g_fftlib.destroy = de;
    strncpy(g_fftlib.name, path, sizeof g_fftlib.name - 1);
    return 0;
}
const char *orc_fft_library(void) { return g_fftlib.handle ? g_fftlib.name : ""; }
/* Plans created from now on use n threads inside the library, as the reference's FFTW back-end does with
 * fftwf_init_threads() + fftwf_plan_with_nthreads(nthreads) (src/fft_impl.cpp:82-88).  Returns 0, or -1 when the loaded
 * library does not export the threads API (MKL's wrappers do; they additionally follow MKL_NUM_THREADS). */
int orc_fft_library_threads(int n) {
    if (!g_fftlib.handle) return -1;
    int (*init_threads)(void) = (int (*)(void))dlsym(g_fftlib.handle, "fftwf_init_threads");
    void (*with_nthreads)(int) = (void (*)(int))dlsym(g_fftlib.handle, "fftwf_plan_with_nthreads");
    if (!init_threads || !with_nthreads) return -1;
    pthread_mutex_lock(&g_plan_mtx);
    init_threads();
    with_nthreads(n < 1 ? 1 : n);
    pthread_mutex_unlock(&g_plan_mtx);
    return 0;
}

/* FFT::FFT src/fft_impl.cpp:63-70 + FFTW::plan_c2c :89-103 / plan_r2c :104-117 */
orc_fft *orc_fft_create(size_t size, int is_real, int downsample_levels,
                        int brightness_offset, int additional_size) {
    orc_fft *f = (orc_fft *)calloc(1, sizeof(orc_fft));
    f->size = size;
    f->is_real = is_real;
    f->downsample_levels = downsample_levels;
    f->additional_size = additional_size;
    f->size_log2 = (int)round(log2((double)size)) + brightness_offset;
    f->windowbuf = (float *)xaligned(sizeof(float) * size);
    orc_build_hann_window(f->windowbuf, (int)size);
    if (!is_real) {
        f->inbuf = (float *)xaligned(sizeof(float) * size * 2);
        f->outbuf = (float *)xaligned(sizeof(float) * (size * 2 + (size_t)additional_size * 2));
        f->outbuf_len = size;
        f->powerbuf = (float *)xaligned(sizeof(float) * size * 2);
        f->quantizedbuf = (int8_t *)xaligned(size * 2);
        f->quantized_len = size * 2;
        f->W = make_twiddles_f32(size);
        f->scr1 = (cf32 *)xaligned(sizeof(cf32) * size);
    } else {
        f->inbuf = (float *)xaligned(sizeof(float) * size);
        f->outbuf = (float *)xaligned(sizeof(float) * (size + 2));
        f->outbuf_len = size / 2;
        f->powerbuf = (float *)xaligned(sizeof(float) * size);
        f->quantizedbuf = (int8_t *)xaligned(size);
        f->quantized_len = size;
        f->W = make_twiddles_f32(size);
        f->Whalf = make_twiddles_f32(size / 2);
        f->scr1 = (cf32 *)xaligned(sizeof(cf32) * (size / 2 + 1));
        f->scr2 = (cf32 *)xaligned(sizeof(cf32) * (size / 2 + 1));
    }
    if (g_fftlib.handle) { /* FFTW_FORWARD = -1, FFTW_ESTIMATE = 1 << 6 (src/fft_impl.cpp:89-117) */
        pthread_mutex_lock(&g_plan_mtx);
        f->lib_plan = is_real ? g_fftlib.plan_r2c((int)size, f->inbuf, f->outbuf, 1u << 6)
                              : g_fftlib.plan_c2c((int)size, f->inbuf, f->outbuf, -1, 1u << 6);
        pthread_mutex_unlock(&g_plan_mtx);
    }
    return f;
}
void orc_fft_destroy(orc_fft *f) {
    if (!f) return;
    if (f->lib_plan && g_fftlib.destroy) {
        pthread_mutex_lock(&g_plan_mtx);
        g_fftlib.destroy(f->lib_plan);
        pthread_mutex_unlock(&g_plan_mtx);
    }
    free(f->windowbuf);
    free(f->inbuf);
    free(f->outbuf);
    free(f->powerbuf);
    free(f->quantizedbuf);
    free(f->W);
    free(f->Whalf);
    free(f->scr1);
    free(f->scr2);
    free(f);
}
float *orc_fft_output(orc_fft *f) { return f->outbuf; }
int8_t *orc_fft_quantized(orc_fft *f) { return f->quantizedbuf; }
float *orc_fft_power(orc_fft *f) { return f->powerbuf; }
size_t orc_fft_outbuf_len(orc_fft *f) { return f->outbuf_len; }
size_t orc_fft_quantized_len(orc_fft *f) { return f->quantized_len; }

/* FFTW::load_real_input src/fft_impl.cpp:131-135 (dsp_multiply_float :119-123) */
void orc_fft_load_real_input(orc_fft *f, const float *a1, const float *a2) {
    size_t h = f->size / 2;
    for (size_t i = 0; i < h; i++) f->inbuf[i] = a1[i] * f->windowbuf[i];
    for (size_t i = 0; i < h; i++) f->inbuf[h + i] = a2[i] * f->windowbuf[h + i];
}
/* FFTW::load_complex_input src/fft_impl.cpp:136-143 (dsp_multiply_complex :124-129:
 * complex<float> * float scales both parts) */
void orc_fft_load_complex_input(orc_fft *f, const float *a1, const float *a2) {
    size_t h = f->size / 2;
    for (size_t i = 0; i < h; i++) {
        f->inbuf[2 * i] = a1[2 * i] * f->windowbuf[i];
        f->inbuf[2 * i + 1] = a1[2 * i + 1] * f->windowbuf[i];
    }
    for (size_t i = 0; i < h; i++) {
        f->inbuf[f->size + 2 * i] = a2[2 * i] * f->windowbuf[h + i];
        f->inbuf[f->size + 2 * i + 1] = a2[2 * i + 1] * f->windowbuf[h + i];
    }
}

/* power_and_quantize src/fft_impl.cpp:24-44: in-place /= normalize (exact: N is a power
 * of two), power = re*re + im*im contracted by GCC to fma(re, re, im*im). */
static void power_and_quantize(float *complexbuf, float *powerbuf, int8_t *quantizedbuf,
                               float normalize, size_t len, int power_offset) {
    long long i;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (len >= 16384)
    for (i = 0; i < (long long)len; i++) {
        complexbuf[i * 2] /= normalize;
        complexbuf[i * 2 + 1] /= normalize;
        float re = complexbuf[i * 2];
        float im = complexbuf[i * 2 + 1];
        float power = fmaf(re, re, im * im);
        powerbuf[i] = power;
        quantizedbuf[i] = orc_quantize(power, power_offset);
    }
}
/* half_and_quantize src/fft_impl.cpp:45-61 */
static void half_and_quantize(const float *powerbuf, float *halfbuf, int8_t *quantizedbuf,
                              size_t len, int power_offset) {
    long long i;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (len >= 16384)
    for (i = 0; i < (long long)len; i++) {
        float power = powerbuf[i * 2] + powerbuf[i * 2 + 1];
        halfbuf[i] = power;
        quantizedbuf[i] = orc_quantize(power, power_offset);
    }
}

/* FFTW::execute src/fft_impl.cpp:144-174, followed by the caller's IQ wrap copy
 * src/fft.cpp:91-98 (memcpy(&X[R], &X[0], A bins)). */
void orc_fft_execute(orc_fft *f) {
    size_t N = f->size;
    if (f->lib_plan) {
        g_fftlib.execute(f->lib_plan); /* fftwf_execute(p), src/fft_impl.cpp:145 */
    } else if (!f->is_real) {
        /* out-of-place so inbuf survives (FFTW_DESTROY_INPUT makes that unobservable) */
        fft_pow2_f32((const cf32 *)f->inbuf, (cf32 *)f->outbuf, f->scr1, N, -1, f->W);
    } else {
        r2c_pow2_f32(f->inbuf, (cf32 *)f->outbuf, N, f->Whalf, f->W, f->scr1, f->scr2);
    }
    size_t base_idx = f->is_real ? 0 : N / 2 + 1;
    size_t L = f->outbuf_len;
    power_and_quantize(&f->outbuf[base_idx * 2], f->powerbuf, f->quantizedbuf, (float)N,
                       L - base_idx, f->size_log2);
    power_and_quantize(f->outbuf, &f->powerbuf[L - base_idx], &f->quantizedbuf[L - base_idx],
                       (float)N, base_idx, f->size_log2);
    size_t out_len = L;
    int8_t *q = f->quantizedbuf;
    float *pw = f->powerbuf;
    for (int i = 0; i < f->downsample_levels - 1; i++) {
        half_and_quantize(pw, pw + out_len, q + out_len, out_len / 2, f->size_log2 - i - 1);
        pw += out_len;
        q += out_len;
        out_len /= 2;
    }
    if (!f->is_real && f->additional_size > 0)
        memcpy(&f->outbuf[2 * N], &f->outbuf[0], sizeof(float) * 2 * (size_t)f->additional_size);
}

/* The quantiser + pyramid of FFTW::execute (src/fft_impl.cpp:149-172) applied to an
 * ALREADY NORMALISED spectrum in the reference's k order (what get_output_buffer()
 * holds after execute()).  Lets a test check another implementation's int8 pyramid
 * bit-exactly against its own spectrum.  power (may be NULL) receives the f32 pyramid. */
void orc_pyramid_from_spectrum(const float *spec, size_t size, int is_real, int downsample_levels,
                               int size_log2, int8_t *q, float *power) {
    size_t L = is_real ? size / 2 : size;
    size_t base_idx = is_real ? 0 : size / 2 + 1;
    size_t total = 0;
    for (int i = 0; i < downsample_levels; i++) total += L >> i;
    float *pw = power ? power : (float *)xaligned(sizeof(float) * (total + L));
    long long i;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (L >= 16384)
    for (i = 0; i < (long long)L; i++) {
        size_t k = is_real ? (size_t)i : ((size_t)i + base_idx) % size;
        float re = spec[2 * k], im = spec[2 * k + 1];
        float p = fmaf(re, re, im * im);
        pw[i] = p;
        q[i] = orc_quantize(p, size_log2);
    }
    size_t out_len = L;
    float *pp = pw;
    int8_t *qq = q;
    for (int lv = 0; lv < downsample_levels - 1; lv++) {
        half_and_quantize(pp, pp + out_len, qq + out_len, out_len / 2, size_log2 - lv - 1);
        pp += out_len;
        qq += out_len;
        out_len /= 2;
    }
    if (!power) free(pw);
}

/* ------------------------------------------------------------------------------------
 * Post-demodulation chain pieces.
 * ---------------------------------------------------------------------------------- */

/* MovingAverage<float> src/utils.h:76-99: boost::circular_buffer(length, 0) (index 0 =
 * newest after push_front), running sum held in a Neumaier<float> whose operator T()
 * returns `sum` only (src/utils.h:24) => a plain f32 running sum. */
typedef struct {
    int length;
    float *q; /* ring, head = newest */
    int head;
    float sum;
} orc_ma;
static void ma_init(orc_ma *m, int length) {
    m->length = length;
    m->q = (float *)calloc((size_t)length, sizeof(float));
    m->head = 0;
    m->sum = 0;
}
static float ma_at(const orc_ma *m, int idx) { /* idx 0 = newest */
    return m->q[(m->head + idx) % m->length];
}
static float ma_insert(orc_ma *m, float val) {
    float oldest = ma_at(m, m->length - 1);
    m->sum = m->sum + (-oldest); /* Neumaier::operator-= => += -value */
    m->head = (m->head + m->length - 1) % m->length;
    m->q[m->head] = val; /* push_front evicts the back */
    m->sum = m->sum + val;
    return m->sum / (float)m->length;
}
/* DCBlocker<float> src/utils.h:139-169 */
struct orc_dcblocker {
    int delay;
    orc_ma ma1, ma2;
};
orc_dcblocker *orc_dc_create(int delay) {
    orc_dcblocker *d = (orc_dcblocker *)calloc(1, sizeof(*d));
    d->delay = delay;
    ma_init(&d->ma1, delay);
    ma_init(&d->ma2, delay);
    return d;
}
void orc_dc_destroy(orc_dcblocker *d) {
    if (!d) return;
    free(d->ma1.q);
    free(d->ma2.q);
    free(d);
}
void orc_dc_remove(orc_dcblocker *d, float *arr, int length) {
    for (int i = 0; i < length; i++) {
        float ma1 = ma_insert(&d->ma1, arr[i]);
        float ma2 = ma_insert(&d->ma2, ma1);
        arr[i] = ma_at(&d->ma1, d->delay - 1) - ma2;
    }
}

/* AGC src/utils/audioprocessing.cpp:5-74 */
struct orc_agc {
    float desired_level, attack_coeff, release_coeff, gain, sample_rate;
    size_t look_ahead_samples;
    /* deques as growable rings */
    float *buf;
    size_t bcap, bhead, bsize;
    float *mx;
    size_t mcap, mhead, msize;
};
orc_agc *orc_agc_create(float desired, float attack_ms, float release_ms, float lookahead_ms,
                        float sr) {
    orc_agc *a = (orc_agc *)calloc(1, sizeof(*a));
    a->desired_level = desired;
    a->gain = 0;
    a->sample_rate = sr;
    a->look_ahead_samples = (size_t)(lookahead_ms * sr / 1000.0f);
    /* `1 - exp(-1.0f / (ms * 0.001f * sr))` (audioprocessing.cpp:13-14): the unqualified
     * exp() on a float argument is C's exp(double); the double result narrows into the
     * float member.  Pinned against oracle/_ref in tests/test_oracle_ref.py. */
    a->attack_coeff = (float)(1 - exp((double)(-1.0f / (attack_ms * 0.001f * sr))));
    a->release_coeff = (float)(1 - exp((double)(-1.0f / (release_ms * 0.001f * sr))));
    a->bcap = a->look_ahead_samples + 4;
    a->buf = (float *)calloc(a->bcap, sizeof(float));
    a->mcap = a->look_ahead_samples + 4;
    a->mx = (float *)calloc(a->mcap, sizeof(float));
    return a;
}
void orc_agc_destroy(orc_agc *a) {
    if (!a) return;
    free(a->buf);
    free(a->mx);
    free(a);
}
void orc_agc_reset(orc_agc *a) {
    a->gain = 0;
    a->bhead = a->bsize = 0;
    a->mhead = a->msize = 0;
}
static void agc_pop(orc_agc *a) {
    float sample = a->buf[a->bhead];
    a->bhead = (a->bhead + 1) % a->bcap;
    a->bsize--;
    if (sample == a->mx[a->mhead]) {
        a->mhead = (a->mhead + 1) % a->mcap;
        a->msize--;
    }
}
static void agc_push(orc_agc *a, float sample) {
    a->buf[(a->bhead + a->bsize) % a->bcap] = sample;
    a->bsize++;
    while (a->msize && fabsf(a->mx[(a->mhead + a->msize - 1) % a->mcap]) < fabsf(sample))
        a->msize--;
    a->mx[(a->mhead + a->msize) % a->mcap] = sample;
    a->msize++;
    if (a->bsize > a->look_ahead_samples) agc_pop(a);
}
void orc_agc_process(orc_agc *a, float *arr, size_t len) {
    for (size_t i = 0; i < len; i++) {
        agc_push(a, arr[i]);
        if (a->bsize == a->look_ahead_samples) {
            float current_sample = a->buf[a->bhead];
            float peak_sample = fabsf(a->mx[a->mhead]);
            float desired_gain = a->desired_level / (peak_sample + 1e-10f);
            if (desired_gain < a->gain)
                a->gain = fmaf(-a->attack_coeff, a->gain - desired_gain, a->gain);
            else
                a->gain = fmaf(a->release_coeff, desired_gain - a->gain, a->gain);
            arr[i] = current_sample * a->gain;
        } else {
            arr[i] = 0.0f;
        }
    }
}

/* dsp_float_to_int16 src/utils/dsp.cpp:152-165 */
void orc_float_to_int16(const float *arr, int32_t *out, float mult, size_t len) {
    for (size_t i = 0; i < len; i++) {
        int32_t v = (int32_t)fmaf(arr[i], mult, 32768.5f) - 32768;
        if (v > 32767) v = 32767;
        if (v < -32768) v = -32768;
        out[i] = v;
    }
}
/* dsp_am_demod src/utils/dsp.cpp:116-126 */
void orc_am_demod(const float *c, float *out, size_t len) {
    for (size_t i = 0; i < len; i++) {
        float re = c[2 * i], im = c[2 * i + 1];
        out[i] = sqrtf(fmaf(re, re, im * im)); /* contraction as in power_and_quantize */
    }
}
/* polar_discriminator_fm src/utils/dsp.cpp:27-35: arg(buf[i] * conj(prev)) */
void orc_polar_discriminator_fm(const float *c, float pre, float pim, float *out, size_t len) {
    for (size_t i = 0; i < len; i++) {
        float a = c[2 * i], b = c[2 * i + 1];
        /* (a+bi)(pre - pim i) = (a*pre + b*pim) + (b*pre - a*pim) i */
        /* GCC's contraction of the complex product for the reference's flags (pinned
         * bit-exact against oracle/_ref): re = fma(a,c,-(b*d)), im = fma(a,d,b*c), d=-pim */
        float re = fmaf(a, pre, b * pim);
        float im = fmaf(a, -pim, b * pre);
        out[i] = atan2f(im, re);
        pre = a;
        pim = b;
    }
}

/* ------------------------------------------------------------------------------------
 * class AudioClient — src/signal.cpp:8-298, state src/signal.h:72-101.
 * The liquid-dsp PLL branch (signal.cpp:242-252) is not restated: liquid's source is
 * not in the reference; the AM target is the envelope branch :253-257.  The carrier
 * transform (p_complex_carrier, :205-222,230-233,238-241) feeds only the liquid branch
 * and has no observable effect otherwise; it is omitted.
 * ---------------------------------------------------------------------------------- */
struct orc_client {
    int is_real, n, fft_result_size, audio_rate;
    int l, r;
    double audio_mid;
    int mode;
    float *fft_input;  /* n complex */
    float *baseband;   /* n complex: audio_complex_baseband */
    float *baseband_prev;
    float *audio_real; /* n */
    float *audio_real_prev;
    orc_dcblocker *dc;
    orc_agc *agc;
};
orc_client *orc_client_create(int is_real, int n, int audio_rate, int fft_result_size) {
    orc_client *c = (orc_client *)calloc(1, sizeof(*c));
    c->is_real = is_real;
    c->n = n;
    c->audio_rate = audio_rate;
    c->fft_result_size = fft_result_size;
    c->fft_input = (float *)xaligned(sizeof(float) * 2 * (size_t)n);
    c->baseband = (float *)xaligned(sizeof(float) * 2 * (size_t)n);
    c->baseband_prev = (float *)xaligned(sizeof(float) * 2 * (size_t)n);
    c->audio_real = (float *)xaligned(sizeof(float) * (size_t)n);
    c->audio_real_prev = (float *)xaligned(sizeof(float) * (size_t)n);
    c->dc = orc_dc_create(audio_rate / 750 * 2);               /* signal.cpp:54 */
    c->agc = orc_agc_create(0.2f, 50.0f, 300.0f, 200.0f, (float)audio_rate); /* :55 */
    c->mode = ORC_USB;
    return c;
}
void orc_client_destroy(orc_client *c) {
    if (!c) return;
    free(c->fft_input);
    free(c->baseband);
    free(c->baseband_prev);
    free(c->audio_real);
    free(c->audio_real_prev);
    orc_dc_destroy(c->dc);
    orc_agc_destroy(c->agc);
    free(c);
}
/* src/signal.cpp:81-94 */
void orc_client_set_audio_range(orc_client *c, int l, double m, int r) {
    c->audio_mid = m;
    c->l = l;
    c->r = r;
}
/* src/signal.cpp:95-97 and :316-328 (on_demodulation_message also resets the AGC) */
void orc_client_set_audio_demodulation(orc_client *c, int mode) {
    c->mode = mode;
    orc_agc_reset(c->agc);
}
/* src/signal.cpp:300-314 */
int orc_client_on_window_message(orc_client *c, int new_l, double m, int new_r) {
    if (new_l < 0 || new_l >= c->fft_result_size || new_r < 0 || new_r >= c->fft_result_size ||
        new_l > new_r)
        return 0;
    if (new_r - new_l > c->n) return 0;
    orc_client_set_audio_range(c, new_l, m, new_r);
    return 1;
}
const float *orc_client_real_prev(orc_client *c) { return c->audio_real_prev; }
const float *orc_client_baseband(orc_client *c) { return c->baseband; }

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

int orc_client_send_audio(orc_client *c, const float *buf, size_t frame_num, float *audio_pre,
                          float *pwr, int32_t *pcm) {
    const int n = c->n;
    const int audio_l = 0;
    const int audio_r = c->r - c->l;
    const int audio_m = (int)floor(c->audio_mid) - c->l;
    const int audio_m_idx = (int)floor(c->audio_mid);
    int len = audio_r - audio_l;

    /* :117-119 sequential f32 accumulate of std::norm (re*re+im*im, GCC contraction
     * fma(re,re,im*im) as in power_and_quantize) */
    float average_power = 0.0f;
    for (int i = 0; i < len; i++) {
        float re = buf[2 * i], im = buf[2 * i + 1];
        average_power = average_power + fmaf(re, re, im * im);
    }

    /* C++ % semantics for negative audio_m_idx, as in the source */
    int flip = (frame_num % 2 == 1) && ((audio_m_idx % 2 == 0 && !c->is_real) ||
                                        (audio_m_idx % 2 == 1 && c->is_real));

    if (c->mode == ORC_USB || c->mode == ORC_LSB) {
        memset(c->fft_input, 0, sizeof(float) * 2 * (size_t)n);
        if (c->mode == ORC_USB) { /* :125-138 */
            int copy_l = imax(audio_l, audio_m);
            int copy_r = imin(audio_r, audio_m + n);
            if (copy_r >= copy_l)
                for (int t = copy_l; t < copy_r; t++) {
                    c->fft_input[2 * (t - audio_m)] = buf[2 * (t - audio_l)];
                    c->fft_input[2 * (t - audio_m) + 1] = buf[2 * (t - audio_l) + 1];
                }
            orc_dft_c2r(c->fft_input, c->audio_real, (size_t)n);
        } else { /* :139-156 */
            int copy_l = imax(audio_l, audio_m - n + 1);
            int copy_r = imin(audio_r, audio_m + 1);
            if (copy_r >= copy_l) {
                /* reverse_copy(buf+copy_l, buf+copy_r, fft_input + audio_m - copy_r + 1) */
                for (int t = copy_l; t < copy_r; t++) {
                    int dst = (audio_m - copy_r + 1) + (copy_r - 1 - t);
                    c->fft_input[2 * dst] = buf[2 * (t - audio_l)];
                    c->fft_input[2 * dst + 1] = buf[2 * (t - audio_l) + 1];
                }
            }
            orc_dft_c2r(c->fft_input, c->audio_real, (size_t)n);
            for (int i = 0; i < n / 2; i++) { /* std::reverse :155 */
                float t = c->audio_real[i];
                c->audio_real[i] = c->audio_real[n - 1 - i];
                c->audio_real[n - 1 - i] = t;
            }
        }
        if (flip) /* :160-168 */
            for (int i = 0; i < n; i++) c->audio_real[i] = -c->audio_real[i];
        for (int i = 0; i < n / 2; i++) c->audio_real[i] += c->audio_real_prev[i]; /* :171 */
    } else { /* AM / FM :173-263 */
        memset(c->fft_input, 0, sizeof(float) * 2 * (size_t)n);
        int pos_copy_l = imax(audio_l, audio_m);
        int pos_copy_r = imin(audio_r, audio_m + n / 2);
        if (pos_copy_r >= pos_copy_l)
            for (int t = pos_copy_l; t < pos_copy_r; t++) {
                c->fft_input[2 * (t - audio_m)] = buf[2 * (t - audio_l)];
                c->fft_input[2 * (t - audio_m) + 1] = buf[2 * (t - audio_l) + 1];
            }
        int neg_copy_l = imax(audio_l, audio_m - n / 2 + 1);
        int neg_copy_r = imin(audio_r, audio_m);
        if (neg_copy_r >= neg_copy_l)
            for (int t = neg_copy_l; t < neg_copy_r; t++) {
                int dst = n - (audio_m - neg_copy_l) + (t - neg_copy_l);
                c->fft_input[2 * dst] = buf[2 * (t - audio_l)];
                c->fft_input[2 * dst + 1] = buf[2 * (t - audio_l) + 1];
            }
        float prev_re = c->baseband[2 * (n / 2 - 1)], prev_im = c->baseband[2 * (n / 2 - 1) + 1];
        memcpy(c->baseband_prev, c->baseband + n, sizeof(float) * (size_t)n); /* second half */
        orc_dft_c2c(c->fft_input, c->baseband, (size_t)n, +1);                /* :214 */
        if (flip)                                                              /* :223-234 */
            for (int i = 0; i < 2 * n; i++) c->baseband[i] = -c->baseband[i];
        for (int i = 0; i < n; i++) c->baseband[i] += c->baseband_prev[i]; /* n/2 complex :235 */
        if (c->mode == ORC_AM) orc_am_demod(c->baseband, c->audio_real, (size_t)(n / 2));
        if (c->mode == ORC_FM)
            orc_polar_discriminator_fm(c->baseband, prev_re, prev_im, c->audio_real,
                                       (size_t)(n / 2));
    }

    /* :266-271 NaN guard: throws, caught at :295 => nothing sent, prev not updated */
    for (int i = 0; i < n / 2; i++)
        if (isnan(c->audio_real[i])) return 1;

    /* :273-275 */
    memcpy(c->audio_real_prev, c->audio_real + n / 2, sizeof(float) * (size_t)(n / 2));

    if (audio_pre) memcpy(audio_pre, c->audio_real, sizeof(float) * (size_t)(n / 2));
    if (pwr) *pwr = average_power;

    if (pcm) {
        orc_dc_remove(c->dc, c->audio_real, n / 2);          /* :278 */
        orc_agc_process(c->agc, c->audio_real, (size_t)(n / 2)); /* :281 */
        orc_float_to_int16(c->audio_real, pcm, 65536 / 4, (size_t)(n / 2)); /* :283-284 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * WaterfallClient::on_window_message level choice — src/waterfall.cpp:53-94.
 * returns -1 if rejected by the sanitiser (:56-58).
 * ---------------------------------------------------------------------------------- */
int orc_waterfall_pick_level(int downsample_levels, int min_waterfall_fft, int *l, int *r) {
    int new_l = *l, new_r = *r;
    if (new_l < 0 || new_r < 0 || new_l >= new_r) return -1;
    float new_l_f = (float)new_l;
    float new_r_f = (float)new_r;
    int new_level = downsample_levels - 1;
    float best_difference = (float)(min_waterfall_fft * 2);
    for (int i = 0; i < downsample_levels; i++) {
        /* `abs((new_r_f - new_l_f) - min_waterfall_fft)` with <cmath>: float overload */
        float send_size = fabsf((new_r_f - new_l_f) - (float)min_waterfall_fft);
        if (send_size < best_difference) {
            best_difference = send_size;
            new_level = i;
            new_l = (int)roundf(new_l_f);
            new_r = (int)roundf(new_r_f);
        }
        new_l_f /= 2;
        new_r_f /= 2;
    }
    *l = new_l;
    *r = new_r;
    return new_level;
}
