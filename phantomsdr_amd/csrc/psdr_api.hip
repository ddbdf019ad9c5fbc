// psdr_api.hip — host side of libpsdr_hip.so: context, tables, launch logic and the
// C-ABI declared in include/psdr.h.  gfx950 only; no CPU fallback lives here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/psdr.h"
#include "demod.h"
#include "epilogue.h"
#include "fft_pass.h"
#include "fft_pass1w.h"
#include "postchain.h"
#include "wire.h"

using namespace psdr;

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(PSDR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

static int drain(psdr_ctx *c);
extern "C" const char *psdr_last_error(void) { return g_err.c_str(); }
extern "C" const char *psdr_version(void) { return "phantomsdr_amd 0.1 (gfx950)"; }

namespace {

enum KernelId { K_PASS1, K_PASS2, K_UNTANGLE, K_TAIL, K_IDFT, K_OLA, K_WFALL, K_POST, K_SEAM, K_BAND, K_COUNT };
const char *kKernelNames[K_COUNT] = {"fft_pass1",  "fft_pass2", "untangle_real", "pyramid_tail",
                                     "demod_idft", "demod_ola", "waterfall_gather", "post_chain",
                                     "real_seam",  "band_pack"};

struct PendingEvent {
    hipEvent_t a, b;
    int kid;
};

struct AudioSlot {
    bool active = false;
    int l = 0, r = 0;
    double mid = 0;
    int mode = PSDR_USB;
    int state_cur = 0;
    int agc_reset = 2;  // post chain: 1 = AGC::reset pending (set_audio_demodulation), 2 = fresh client
    bool paused = false;  // psdr_client_set_paused: sits out the demodulation batches, all state frozen
    uint64_t last_seq = 0;  // the demodulation batch (ctx->demod_seq) that last included this slot; 0: none yet
    int b_l = 0, b_r = 0;   // the window that batch was demodulated with ...
    double b_mid = 0;
    int f_l = 0, f_r = 0;   // ... and the one of the batch psdr_fetch_batch copied
    double f_mid = 0;
};
struct WfSlot {
    bool active = false;
    int level = 0, l = 0, r = 0;
    // the last psdr_waterfall_batch: what was gathered, and with which window (set_range may run
    // on another thread between the batch and psdr_read_waterfall)
    size_t out_off = 0;
    int nsent = 0;
    int b_level = 0, b_l = 0, b_r = 0;
};

// Small host->device parameter blocks (client lists) are double-buffered K deep so a new
// batch can be enqueued without waiting for the previous one to drain.
struct ParamRing {
    static constexpr int K = 8;
    unsigned char *h = nullptr, *d = nullptr;
    size_t slot_bytes = 0;
    hipEvent_t ev[K] = {};
    bool used[K] = {};
    int idx = 0;
    int init(size_t bytes) {
        slot_bytes = (bytes + 255) & ~(size_t)255;
        if (hipHostMalloc((void **)&h, slot_bytes * K, hipHostMallocDefault) != hipSuccess) return -1;
        if (hipMalloc((void **)&d, slot_bytes * K) != hipSuccess) return -1;
        for (int i = 0; i < K; i++)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return -1;
        return 0;
    }
    void destroy() {
        if (h) hipHostFree(h);
        if (d) hipFree(d);
        for (int i = 0; i < K; i++)
            if (ev[i]) hipEventDestroy(ev[i]);
        h = d = nullptr;
    }
    // returns the slot to fill (waits only if the ring wrapped onto a still-busy slot); -1: HIP error
    int acquire() {
        idx = (idx + 1) % K;
        if (used[idx] && hipEventSynchronize(ev[idx]) != hipSuccess) return -1;
        return idx;
    }
    void *host(int i) { return h + slot_bytes * i; }
    void *dev(int i) { return d + slot_bytes * i; }
    hipError_t release(int i, hipStream_t s) {
        const hipError_t e = hipEventRecord(ev[i], s);
        used[i] = e == hipSuccess;
        return e;
    }
};

int ilog2(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) l++;
    return l;
}

}  // namespace

struct psdr_ctx {
    psdr_config cfg;
    int device = 0;
    int num_cus = 256;
    size_t N = 0, M = 0, R = 0;
    int M1 = 0, M2 = 0, log2M1 = 0, log2M2 = 0;
    int T1 = 0, T2 = 0;
    bool is_real = false;
    // real input, N/2 = 1024*1024 or 2048*1024 points: pass 2 untangles, normalises, takes the power
    // and builds pyramid levels 0..3 itself (k_fft_pass2_real); smaller real transforms keep the
    // three-pass form (pass 1, pass 2, k_untangle_real)
    bool real_fused = false;
    // 2^20-point IQ transforms of 8/16-bit samples: pass 1 with wave-owned column couples (fft_pass1w.h), Y
    // couple-major.  Experimental, off unless PSDR_P1_WAVE=1: correct (parity tests run it), not yet faster
    bool p1_wave = false;
    SpecLayout lay{};                // device layout of the spectrum (natural unless real_fused)
    int nbands = 0, band_H = 0;      // psdr_set_band_layout: band regions (SpecLayout mode 3), halo columns per band
    bool y_blocked = false;          // PSDR_REAL_YBLOCKED (tuning)
    int seg_len_env = 0;             // PSDR_SEG_LEN (tuning): tiles per chain segment
    float *d_seamP = nullptr, *d_seamC = nullptr;  // of the current result set
    float *seam_pool[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    size_t seam_cap = 0;             // segments the seam buffers hold
    int size_log2 = 0;
    int levels = 0;
    size_t spec_stride = 0;  // complex elements per frame
    size_t q_len = 0, q_stride = 0;
    int LT = 0;  // pyramid levels finished inside the fused kernel
    size_t p_stride = 0;
    int max_batch = 1;
    int min_waterfall_fft = 0;  // input.waterfall_size (src/spectrumserver.cpp:56)
    std::set<const void *> lds_attr_done;  // kernels whose dynamic-LDS limit was raised on this device
    // Two streams: the FFT passes run on `stream`; everything that only consumes a finished
    // batch (pyramid tail, demodulation, waterfall gather) runs on `side`, so it overlaps
    // the next batch's pass 1 (its work-groups fit next to the persistent FFT work-groups).
    hipStream_t stream = nullptr, side = nullptr;
    hipStream_t own_stream = nullptr, own_side = nullptr, own_p1 = nullptr;
    // pass 1 runs on its own stream so that pass 1 of batch i+1 fills the CUs that pass 2 of
    // batch i leaves one by one (persistent work-groups: launch ramp, prologue and tail of one
    // kernel overlap with the other kernel's steady state); Y is double-buffered for that
    hipStream_t p1 = nullptr;
    cf *y_pool[2] = {nullptr, nullptr};
    int cur_y = 0;
    bool y_pending[2] = {false, false};
    // TileQueue counters: a ring of TICKET_SLOTS launches x 8 counters per pass; half the ring is
    // re-zeroed (in stream order) whenever the other half starts being used
    unsigned *d_tickets[2] = {nullptr, nullptr};
    unsigned ticket_pos[2] = {0, 0};
    bool no_col_tail = false;  // tuning (PSDR_NO_COL_TAIL=1)
    bool static_tiles = false, no_p1_stream = true;  // tuning knobs (PSDR_STATIC_TILES, PSDR_P1_STREAM)
    unsigned p1_grid = 0, p2_grid = 0;  // PSDR_P1_GRID / PSDR_P2_GRID: work-groups of each pass (0: all CUs)
    bool input_on_main = false;  // level-1 H2D staging was enqueued on the main stream
    hipEvent_t ev_in = nullptr, ev_p1[2] = {nullptr, nullptr}, ev_p2[2] = {nullptr, nullptr};
    hipEvent_t ev_fft_done = nullptr, ev_side_done = nullptr;
    bool side_pending = false;
    // Result buffers (spectrum, pyramid, level powers) exist twice: batch b+1 is produced
    // into the other set while the side stream still consumes batch b, so the FFT stream only
    // ever waits for the consumers of batch b-1.  d_spec/d_q/d_qt/d_pscr point at the set of
    // the LAST processed batch.
    int cur_set = 0;
    cf *spec_pool[2] = {nullptr, nullptr};
    int8_t *q_pool[2] = {nullptr, nullptr}, *qt_pool[2] = {nullptr, nullptr};
    float *pscr_pool[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipEvent_t ev_set_done[2] = {nullptr, nullptr};
    bool set_pending[2] = {false, false};

    cf *d_Wl1 = nullptr, *d_Wl2 = nullptr, *d_TA = nullptr, *d_TB = nullptr;
    cf *d_UA = nullptr, *d_UB = nullptr, *d_UG = nullptr;
    cf wdelta = {1.f, 0.f};  // W_N^1
    unsigned long long *d_trace = nullptr;  // PSDR_TRACE tuning builds: per pass [8][16] phase stamps + [256][8] work-group timeline
    int log2B = 0, log2UB = 0;
    cf *d_Z = nullptr, *d_spec = nullptr;
    int8_t *d_q = nullptr;   // level-major pyramid (the reference's layout)
    int8_t *d_qt = nullptr;  // tiled records of levels 0..LT (IQ fused epilogue), quantize.h
    size_t qt_stride = 0;
    int tiled_lt = -1, tile_ch = 16;
    RecMap recmap{};  // order of the tiled records and of the level-LT scratch
    std::vector<char> q_untiled;  // per frame: level-major copy of the tiled levels is current
    float *d_pscr[2] = {nullptr, nullptr};

    // level 1
    float *d_stage = nullptr;
    float *h_out = nullptr;
    int8_t *h_q = nullptr;
    bool loaded = false, executed = false, out_valid = false, q_valid = false;
    int last_nframes = 0;

    // audio clients
    std::mutex mtx;
    std::vector<AudioSlot> aslots;
    int n = 0;  // audio_fft_size
    int nstages = 0;
    int radix[PSDR_MAX_STAGES];
    int lds_mode = 0;
    bool demod_chain = true;  // PSDR_DEMOD_CHAIN=0: the two-kernel path (k_demod_idft_fixed + k_demod_ola) for n = 360 / 720 too
    int demod_chain_k = 0;    // PSDR_DEMOD_K: frames per chain (0: 8, 4 when there are few clients)
    size_t idft_lds = 0;
    int4 *d_stage_tab = nullptr;
    int idft_threads = 256;
    bool idft_generic = false;  // tuning (PSDR_IDFT_GENERIC=1): never the compile-time plans
    bool idft_block = false;  // tuning (PSDR_IDFT_BLOCK=1): force the one-work-group-per-item kernel
    cf *d_Wn = nullptr, *d_ypost = nullptr, *d_gscratch = nullptr, *d_bb_tail = nullptr,
       *d_bb_last = nullptr;
    // post-demodulation chain (postchain.h), allocated by psdr_set_post_chain
    bool post_on = false;
    PostArgs post{};
    // The chain is a two-stage pipeline across batches: stage 1 (index, gather, moving averages) of
    // batch b+1 runs on `side` while stage 2 (look-ahead peak, gain, int16) of batch b runs on
    // `side2`; what the stages share is double-buffered (V1, frame offsets, stream lengths)
    hipStream_t side2 = nullptr;
    // ... and stage 1 has a stream of its own too (round 3): on `side` its sequential kernel (k_pc_ma2, ~0.9 ms beside
    // the passes) sat between this batch's demodulation and the NEXT batch's tails and demodulation - the side
    // stream, not the GPU, set the step (1.88 ms of serial work per 1.4 ms of passes)
    hipStream_t side3 = nullptr;
    hipEvent_t ev_want[2] = {nullptr, nullptr};  // w_t of this parity is ready (the gain recurrence may start)
    hipEvent_t ev_demod = nullptr, ev_gather = nullptr;  // demodulation done (stage 1 may read); audio rows read (the next demodulation may write)
    bool gather_pending = false;
    float *post_v1[2] = {nullptr, nullptr};
    float *post_p[2] = {nullptr, nullptr}, *post_s[2] = {nullptr, nullptr};  // prefix / suffix maxima, then w_t / g_t
    int *post_fstart[2] = {nullptr, nullptr}, *post_len[2] = {nullptr, nullptr};
    hipEvent_t ev_s1[2] = {nullptr, nullptr}, ev_s2[2] = {nullptr, nullptr};
    uint64_t chain_seq = 0;
    bool side2_pending = false;
    std::vector<void *> post_allocs;
    float *d_pwr = nullptr, *d_audio = nullptr, *d_real_prev = nullptr;
    int *d_nan = nullptr;
    ParamRing client_ring;
    int last_demod_frames = 0;
    uint64_t demod_seq = 0;  // number of demodulation batches so far (AudioSlot::last_seq)
    // psdr_fetch_batch: pinned host mirror of the last batch's results, [slot][frame][...]
    float *h_audio = nullptr, *h_pwr = nullptr;
    int32_t *h_nan = nullptr, *h_pcm = nullptr;
    int fetched_frames = 0;
    uint64_t fetched_seq = 0;
    bool fetched_pcm = false;

    // waterfall clients
    std::vector<WfSlot> wslots;
    ParamRing wf_ring;  // [WfClient x W][int x F]
    size_t wf_sent_off = 0;
    int8_t *d_wfout = nullptr;
    size_t wfout_cap = 0;

    // streaming ingest ring (psdr_ring_*)
    struct IngestRing {
        static constexpr int NEV = 16;
        unsigned char *d = nullptr;  // nhalves + 1 slots (the last mirrors slot 0: a frame window may end there)
        int nhalves = 0;
        size_t hb = 0;
        hipStream_t copy = nullptr;
        std::vector<hipEvent_t> ev_written;   // per slot: its last H2D copy
        std::vector<char> ever_written;
        std::vector<uint64_t> reader_seq;     // per slot: the last psdr_process_ring call that read it
        hipEvent_t ev_read[NEV] = {};         // pass 1 of process call seq % NEV has consumed its halves
        uint64_t seq = 0;
    } ring;

    // instrumentation
    bool profiling = false;   // psdr_set_profiling mode 1: hipEvent brackets around every launch
    bool kclock = false;      // mode 2: device-clock stamps inside the two FFT passes (fft_pass.h kclk_*)
    static constexpr unsigned KCLK_SLOTS = 8192;  // launches per pass that can be stamped between two resets
    unsigned long long *d_kclk = nullptr;         // [2 passes][KCLK_SLOTS][begin, end]
    unsigned kclk_pos[2] = {0, 0}, kclk_done[2] = {0, 0};
    double wall_clock_khz = 100000.0;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> pool;
    double k_ms[K_COUNT] = {0};
    int64_t k_n[K_COUNT] = {0};
    std::vector<float> k_samples[K_COUNT];  // per-launch durations in us since the last reset (bounded)
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace {

struct ProfScope {
    psdr_ctx *c;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    ProfScope(psdr_ctx *c_, int kid_, hipStream_t st_ = nullptr) : c(c_), kid(kid_), st(st_ ? st_ : c_->stream) {
        if (!c->profiling) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->pool.empty()) {
                e = c->pool.back();
                c->pool.pop_back();
            } else if (hipEventCreate(&e) != hipSuccess) {
                e = nullptr;
            }
            return e;
        };
        a = get();
        b = get();
        if (!a || !b || hipEventRecord(a, st) != hipSuccess) {  // no timing for this launch
            if (a) c->pool.push_back(a);
            if (b) c->pool.push_back(b);
            a = b = nullptr;
        }
    }
    ~ProfScope() {
        if (!c->profiling || !a) return;
        if (hipEventRecord(b, st) == hipSuccess) {
            c->pending.push_back({a, b, kid});
        } else {
            c->pool.push_back(a);
            c->pool.push_back(b);
        }
    }
};

void resolve_pending(psdr_ctx *c) {
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->p1);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->side);
    if (c->side2) hipStreamSynchronize(c->side2);
    if (c->side3) hipStreamSynchronize(c->side3);
    for (auto &p : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->k_ms[p.kid] += ms;
            c->k_n[p.kid] += 1;
            if (c->k_samples[p.kid].size() < 65536) c->k_samples[p.kid].push_back(ms * 1e3f);
        }
        c->pool.push_back(p.a);
        c->pool.push_back(p.b);
    }
    c->pending.clear();
}

// mode 2: the stamps of the launches since the last call -> k_ms / k_n / k_samples of the two passes
void resolve_kclock(psdr_ctx *c) {
    if (!c->d_kclk) return;
    bool any = false;
    for (int w = 0; w < 2; w++) any = any || c->kclk_done[w] < std::min(c->kclk_pos[w], psdr_ctx::KCLK_SLOTS);
    if (!any) return;
    hipStreamSynchronize(c->p1);
    hipStreamSynchronize(c->stream);
    std::vector<unsigned long long> h((size_t)2 * psdr_ctx::KCLK_SLOTS * 2);
    if (hipMemcpy(h.data(), c->d_kclk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int w = 0; w < 2; w++) {
        const int kid = w == 0 ? K_PASS1 : K_PASS2;
        const unsigned end = std::min(c->kclk_pos[w], psdr_ctx::KCLK_SLOTS);
        for (unsigned i = c->kclk_done[w]; i < end; i++) {
            const unsigned long long b = h[((size_t)w * psdr_ctx::KCLK_SLOTS + i) * 2], e = h[((size_t)w * psdr_ctx::KCLK_SLOTS + i) * 2 + 1];
            if (e <= b) continue;  // (never launched / no work-group ran)
            const double us = (double)(e - b) * 1e3 / c->wall_clock_khz;
            c->k_ms[kid] += us * 1e-3;
            c->k_n[kid] += 1;
            if (c->k_samples[kid].size() < 65536) c->k_samples[kid].push_back((float)us);
        }
        c->kclk_done[w] = end;
    }
}
// stamp slot of the next launch of pass `which` (nullptr: mode 2 off or the ring is full)
unsigned long long *next_kclk(psdr_ctx *c, int which) {
    if (!c->kclock || !c->d_kclk || c->kclk_pos[which] >= psdr_ctx::KCLK_SLOTS) return nullptr;
    return c->d_kclk + ((size_t)which * psdr_ctx::KCLK_SLOTS + c->kclk_pos[which]++) * 2;
}
// re-arm the whole ring: begin = ~0, end = 0 (streams drained by the caller)
int reset_kclock(psdr_ctx *c) {
    if (!c->d_kclk) return PSDR_OK;
    std::vector<unsigned long long> h((size_t)2 * psdr_ctx::KCLK_SLOTS * 2);
    for (size_t i = 0; i < h.size(); i += 2) h[i] = ~0ull, h[i + 1] = 0ull;
    HIPCHK(hipMemcpy(c->d_kclk, h.data(), h.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    c->kclk_pos[0] = c->kclk_pos[1] = c->kclk_done[0] = c->kclk_done[1] = 0;
    return PSDR_OK;
}

std::vector<cf> make_twiddles(size_t count, size_t mult, size_t period, int sign) {
    // exp(sign * 2 pi i * (j*mult) / period), j < count, generated in double
    std::vector<cf> w(count);
    for (size_t j = 0; j < count; j++) {
        const double a = (double)sign * 2.0 * M_PI * (double)((j * mult) % period) / (double)period;
        w[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    return w;
}

template <typename T>
int upload(T **dst, const std::vector<T> &v) {
    HIPCHK(hipMalloc((void **)dst, v.size() * sizeof(T)));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PSDR_OK;
}

constexpr unsigned TICKET_SLOTS = 64;
// counters for the next launch of pass `which` on stream st
int next_tickets(psdr_ctx *c, int which, hipStream_t st, unsigned **out) {
    if (c->static_tiles) {
        *out = nullptr;
        return PSDR_OK;
    }
    unsigned &pos = c->ticket_pos[which];
    const unsigned slot = pos % TICKET_SLOTS;
    if (slot % (TICKET_SLOTS / 2) == 0 && pos >= TICKET_SLOTS / 2) {
        // entering a half of the ring: its counters were last used TICKET_SLOTS/2 launches ago on
        // this same stream, so clearing them here is ordered after those launches
        HIPCHK(hipMemsetAsync(c->d_tickets[which] + (size_t)slot * 8, 0, (TICKET_SLOTS / 2) * 8 * sizeof(unsigned), st));
    }
    *out = c->d_tickets[which] + (size_t)slot * 8;
    pos++;
    return PSDR_OK;
}

// tile widths: T = min(16384/L, other dimension)
int pick_T(int L, int other) { return std::min(16384 / L, other); }

// persistent launch: as many work-groups as the CUs hold (LDS-limited), a multiple of 8 (XCD
// round-robin of the TileQueue), or one per tile when there are fewer tiles than that
unsigned persistent_grid(psdr_ctx *c, unsigned blocks, size_t lds) {
    const unsigned cap = ((unsigned)c->num_cus * (unsigned)std::max<size_t>(1, 160 * 1024 / lds)) & ~7u;
    return blocks <= cap ? blocks : std::max(cap, 8u);
}

template <int L, int T, int SB, bool PAIR = false>
int launch_pass1_t(psdr_ctx *c, const Pass1Args &a, unsigned blocks) {
    // tile + W_L (= first twiddle factor) + second twiddle factor (M2 entries)
    const size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf) + (size_t)a.M2 * sizeof(cf);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass1<L, T, SB, PAIR>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass1<L, T, SB, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    ProfScope ps(c, K_PASS1, c->p1);
    // persistent: as many work-groups per CU as their LDS admits (a 128 KiB tile: one)
    unsigned grid = persistent_grid(c, blocks, lds);
    if (c->p1_grid && c->p1_grid < grid) grid = c->p1_grid;
    hipLaunchKernelGGL((k_fft_pass1<L, T, SB, PAIR>), dim3(grid), dim3(L * T / 32), lds, c->p1, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
template <int L, int T, bool FUSED, int TWC, bool YCM = false, bool BAND = false>
int launch_pass2_t(psdr_ctx *c, const Pass2Args &a, unsigned blocks) {
    constexpr size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass2<L, T, FUSED, TWC, YCM, BAND>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass2<L, T, FUSED, TWC, YCM, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, K_PASS2);
    unsigned grid = persistent_grid(c, blocks, lds);
    if (c->p2_grid && c->p2_grid < grid) grid = c->p2_grid;
    hipLaunchKernelGGL((k_fft_pass2<L, T, FUSED, TWC, YCM, BAND>), dim3(grid), dim3(L * T / 32), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
// pass 1 with wave-owned column couples (fft_pass1w.h): 2^20-point IQ frames of 8/16-bit samples
template <int SB>
int launch_pass1_w(psdr_ctx *c, const Pass1Args &a, unsigned blocks) {
    constexpr size_t lds = pass1w_lds_bytes<SB>();
    if (c->lds_attr_done.insert((const void *)k_fft_pass1_w<SB>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass1_w<SB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, K_PASS1, c->p1);
    unsigned grid = persistent_grid(c, blocks, lds);
    if (c->p1_grid && c->p1_grid < grid) grid = c->p1_grid;
    hipLaunchKernelGGL((k_fft_pass1_w<SB>), dim3(grid), dim3(kPass1wThreads), lds, c->p1, a);
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
int launch_pass1(psdr_ctx *c, int L, int T, int sb, const Pass1Args &a, unsigned blocks, bool pair = false) {
    if (pair) {
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

// fused real-input pass 2 (TWC = pass-1 tile width: 16 for 1024 x 1024, 8 for 2048 x 1024)
template <int TWC>
int launch_pass2_real_t(psdr_ctx *c, const Pass2Args &a) {
    constexpr int L = 1024, T = 16;
    constexpr size_t lds = (size_t)L * T * sizeof(cf) + (size_t)L * sizeof(cf) + 2 * (size_t)L * sizeof(float);
    // (per context = per device: the attribute is a property of the function ON a device)
    if (c->lds_attr_done.insert((const void *)k_fft_pass2_real<L, T, TWC>).second)
        HIPCHK(hipFuncSetAttribute((const void *)k_fft_pass2_real<L, T, TWC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, K_PASS2);
    unsigned grid = persistent_grid(c, a.total_slots, lds);
    if (c->p2_grid && c->p2_grid < grid) grid = c->p2_grid;
    hipLaunchKernelGGL((k_fft_pass2_real<L, T, TWC>), dim3(grid), dim3(L * T / 32), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
// Tiles of one frame a work-group walks in a chain: long chains carry the mirror-side octets in LDS
// (nothing extra in HBM), short chains give the persistent grid enough independent segments.  Aim
// for about two segments per work-group: the pass is bound by memory, not by balance (256 frames of
// 2^21 points, same box: 1012-1023 us with 16 segments per work-group, 990-1030 with 8, 4 or 2), and
// every segment costs one seam (k_real_seam: 177 / 89 / 52 / 31 us at 4 / 8 / 16 / 32 tiles per segment,
// run beside the next batch's pass 1, which it slows).
int real_seg_len(const psdr_ctx *c, int nframes) {
    const int G = c->M1 / 16;
    if (c->seg_len_env > 0) {
        int sl = 1;
        while (sl * 2 <= c->seg_len_env && sl * 2 <= G) sl *= 2;
        return sl;
    }
    const long long want = (long long)G * nframes / (2LL * std::max(c->num_cus, 1));
    int sl = 1;
    while (sl * 2 <= want && sl * 2 <= G) sl *= 2;
    return sl;
}

size_t fmt_bytes(int fmt) {
    switch (fmt) {
    case PSDR_FMT_U8:
    case PSDR_FMT_S8:
        return 1;
    case PSDR_FMT_U16:
    case PSDR_FMT_S16:
        return 2;
    case PSDR_FMT_F32:
        return 4;
    default:
        return 8;
    }
}

// forward FFT + power + int8 pyramid for nframes frames (src/fft.cpp:61-98 per frame)
void select_set(psdr_ctx *c, int set) {
    c->cur_set = set;
    c->d_spec = c->spec_pool[set];
    c->d_q = c->q_pool[set];
    c->d_qt = c->qt_pool[set];
    c->d_pscr[0] = c->pscr_pool[set][0];
    c->d_pscr[1] = c->pscr_pool[set][1];
    c->d_seamP = c->seam_pool[set][0];
    c->d_seamC = c->seam_pool[set][1];
}

int process_frames(psdr_ctx *c, const void *d_halves, int nframes, int fmt, hipEvent_t ev_raw_consumed = nullptr) {
    // alternate the result set when the consumers run on their own stream
    // (banded spectrum: also on a caller's stream - the regions of batch b are read by the peers, asynchronously,
    // while batch b+1 is transformed)
    if (c->side != c->stream || c->nbands) select_set(c, c->cur_set ^ 1);
    const int cols = 1;
    const int sb = fmt <= PSDR_FMT_S8 ? 2 : (fmt <= PSDR_FMT_S16 ? 4 : 8);  // image bytes per sample
    const unsigned tiles1 = (unsigned)(c->M2 / (c->T1 * cols)), tiles2 = (unsigned)(c->M1 / c->T2);
    const bool piped = c->p1 != c->stream;
    if (piped) {
        c->cur_y ^= 1;
        if (c->input_on_main) {  // the staged input was copied on the main stream
            HIPCHK(hipEventRecord(c->ev_in, c->stream));
            HIPCHK(hipStreamWaitEvent(c->p1, c->ev_in, 0));
        }
        // this Y buffer's previous reader (pass 2, two batches ago) must be done
        if (c->y_pending[c->cur_y]) HIPCHK(hipStreamWaitEvent(c->p1, c->ev_p2[c->cur_y], 0));
    }
    c->input_on_main = false;
    cf *Y = c->y_pool[c->cur_y];
    Pass1Args a1{};
    a1.raw = d_halves;
    a1.Y = Y;
    a1.Wl = c->d_Wl1;
    a1.TB = c->d_TB;
    a1.yblk = (size_t)c->M1 * (c->T1 * cols);  // plain: one linear block per pass-1 tile
    a1.l2t2 = ilog2((size_t)c->T2);
    // fused real: pass-2-tile-major by default (fft_pass.h, "Y layout"); PSDR_REAL_YBLOCKED=1: rows regrouped
    // inside the pass-1 tile's own linear block
    a1.ytile = c->y_blocked ? (size_t)16 * (c->T1 * cols) : (size_t)c->M2 * c->T2;
    a1.ytl = c->y_blocked ? a1.yblk : (size_t)16 * (c->T1 * cols);
    a1.yframe = c->M;
    a1.wdelta = c->wdelta;
    a1.M2 = c->M2;
    a1.log2M2 = c->log2M2;
    a1.fmt = fmt;
    a1.is_real = c->is_real ? 1 : 0;
    a1.rot = c->is_real ? 0 : 1;
    a1.trace = c->d_trace;
    a1.kclk = next_kclk(c, 0);
    a1.ymask = ~0u;
#ifdef PSDR_TUNING_BUILD  // never in the shipped library: a timing-only experiment with WRONG results (frames share Y)
    if (const char *e = getenv("PSDR_Y_ALIAS")) a1.ymask = (unsigned)atoi(e) - 1u;
#endif
    {
        int rc = next_tickets(c, 0, c->p1, &a1.tickets);
        if (rc) return rc;
    }
    a1.tiles_per_frame = tiles1;
    a1.total_slots = tiles1 * (unsigned)nframes;
    const bool wave1 = c->p1_wave && sb <= 4;  // (f32 / f64 samples: the image does not fit, classic kernel)
    int rc = wave1 ? (sb == 2 ? launch_pass1_w<2>(c, a1, a1.total_slots) : launch_pass1_w<4>(c, a1, a1.total_slots))
                   : launch_pass1(c, c->M1, c->T1, sb, a1, a1.total_slots, c->real_fused);
    if (rc) return rc;
    if (ev_raw_consumed) HIPCHK(hipEventRecord(ev_raw_consumed, c->p1));  // pass 1 is the only reader of the raw halves
    if (piped) {
        HIPCHK(hipEventRecord(c->ev_p1[c->cur_y], c->p1));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_p1[c->cur_y], 0));
    }

    Pass2Args a2{};
    a2.Y = Y;
    a2.Wl = c->d_Wl2;
    a2.M1 = c->M1;
    a2.log2M1 = c->log2M1;
    a2.TW = c->T1 * cols;
    a2.yblk = a1.yblk;
    a2.ytile = c->y_blocked ? (size_t)16 * (c->T1 * cols) : a1.ytile;
    a2.yjs = c->y_blocked ? a1.yblk : (size_t)16 * (c->T1 * cols);
    a2.yframe = a1.yframe;
    a2.log2TW = ilog2((size_t)(c->T1 * cols));
    a2.inv_n = 1.0f / (float)c->N;
    a2.size_log2 = c->size_log2;
    a2.nlevels = c->levels;
    a2.Qt = c->d_qt;
    a2.qt_stride = c->qt_stride;
    a2.Pscr = c->d_pscr[0];
    a2.p_stride = c->p_stride;
    a2.trace = c->d_trace ? c->d_trace + 128 + 2304 : nullptr;
    a2.kclk = next_kclk(c, 1);
    a2.ymask = a1.ymask;
    {
        int rc2 = next_tickets(c, 1, c->stream, &a2.tickets);
        if (rc2) return rc2;
    }
    a2.tiles_per_frame = tiles2;
    a2.total_slots = tiles2 * (unsigned)nframes;
    // pass 2 overwrites this result set: its previous consumers (two batches ago) must be done
    if (c->set_pending[c->cur_set] && c->side != c->stream)
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_set_done[c->cur_set], 0));
    auto run_pass2 = [&](bool fused) -> int {
        if (wave1) return launch_pass2_t<1024, 16, true, 16, true>(c, a2, a2.total_slots);  // couple-major Y
        return launch_pass2(c, c->M2, c->T2, fused, a2, a2.total_slots);
    };
    int seam_S = 0, seam_SL = 0;  // fused real path: the seam kernel runs with the consumers
    if (!c->is_real) {
        a2.X = c->d_spec;
        a2.spec_stride = c->spec_stride;
        if (c->nbands) {
            a2.l2Lb = c->lay.l2Lb;
            a2.lbmask = (1 << c->lay.l2Lb) - 1;
            a2.Lw = c->lay.Lw;
            a2.band_stride = c->lay.band_stride;
            rc = a2.TW == 16 ? launch_pass2_t<1024, 16, true, 16, false, true>(c, a2, a2.total_slots)
                             : launch_pass2_t<1024, 16, true, 8, false, true>(c, a2, a2.total_slots);
            if (rc) return rc;
            if (c->band_H > 0) {  // the first columns of band b+1 once more, behind band b's own
                const size_t n16 = (size_t)c->nbands * nframes * (c->M1 / 16) * c->band_H * 8;  // 16-byte pieces
                hipLaunchKernelGGL(k_band_halo, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, c->d_spec,
                                   c->spec_stride, c->lay, c->nbands, nframes, c->M1 / 16, c->band_H);
                HIPCHK(hipGetLastError());
            }
        } else {
            rc = run_pass2(true);
            if (rc) return rc;
        }
    } else if (c->real_fused) {
        a2.X = c->d_spec;
        a2.spec_stride = c->spec_stride;
        a2.UA = c->d_UA;
        a2.UB = c->d_UB;
        a2.UG = c->d_UG;
        a2.log2UB = c->log2UB;
        a2.seg_len = real_seg_len(c, nframes);
        const unsigned S = tiles2 / (unsigned)a2.seg_len;
        if ((size_t)nframes * S > c->seam_cap)
            return fail(PSDR_ERR_STATE, "seam buffers too small (%zu segments, %zu allocated)", (size_t)nframes * S, c->seam_cap);
        a2.seamP = c->d_seamP;
        a2.seamC = c->d_seamC;
        a2.total_slots = S * (unsigned)nframes;
        rc = c->M1 == 1024 ? launch_pass2_real_t<16>(c, a2) : launch_pass2_real_t<8>(c, a2);
        if (rc) return rc;
        seam_S = (int)S;
        seam_SL = a2.seg_len;
    } else {
        a2.X = c->d_Z;
        a2.spec_stride = c->M;
        rc = run_pass2(false);
        if (rc) return rc;
        UntangleArgs u{};
        u.Z = c->d_Z;
        u.X = c->d_spec;
        u.spec_stride = c->spec_stride;
        u.M = c->M;
        u.TA = c->d_UA;
        u.TB = c->d_UB;
        u.log2B = c->log2UB;
        u.inv_n = 1.0f / (float)c->N;
        u.size_log2 = c->size_log2;
        u.nlevels = c->levels;
        u.Q = c->d_q;
        u.q_stride = c->q_stride;
        u.Pscr = c->d_pscr[0];
        u.p_stride = c->p_stride;
        ProfScope ps(c, K_UNTANGLE);
        const unsigned nb = (unsigned)((c->M / 8 + 255) / 256);
        hipLaunchKernelGGL(k_untangle_real, dim3(nb, nframes), dim3(256), 0, c->stream, u);
        HIPCHK(hipGetLastError());
    }
    if (piped) {
        HIPCHK(hipEventRecord(c->ev_p2[c->cur_y], c->stream));
        c->y_pending[c->cur_y] = true;
    }
    // consumers of the finished batch go to the side stream
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_fft_done, c->stream));
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_fft_done, 0));
    }
    if (seam_S) {  // fused real input: the mirror octets of every chain segment's first tile (epilogue.h)
        SeamArgs sa{};
        sa.seamP = c->d_seamP;
        sa.seamC = c->d_seamC;
        sa.S = seam_S;
        sa.SL = seam_SL;
        sa.L = c->M2;
        sa.size_log2 = c->size_log2;
        sa.Qt = c->d_qt;
        sa.qt_stride = c->qt_stride;
        sa.Pscr = c->d_pscr[0];
        sa.p_stride = c->p_stride;
        ProfScope ps(c, K_SEAM, c->side);
        hipLaunchKernelGGL(k_real_seam, dim3(seam_S, nframes), dim3(256), 0, c->side, sa);
        HIPCHK(hipGetLastError());
    }
    // remaining pyramid levels from the partial level in scratch
    int lvl = c->LT;
    size_t len = c->R >> lvl;
    int cur = 0;
    // tile-major sums of a fused pass 2 (rows of 1024 outputs): one thread per output row takes the
    // levels inside a row (k_col_tail), the generic kernel the few above
    const int ng = (int)(len >> c->log2M2);  // groups per output row
    const bool col_tail = c->recmap.mapped && c->M2 == 1024 && c->recmap.l2gpt == 0 && (ng == 64 || ng == 128 || ng == 256) &&
                          !c->no_col_tail;
    if (col_tail && lvl + 1 < c->levels) {
        ColTailArgs t{};
        t.Pin = c->d_pscr[0];
        t.in_stride = c->p_stride;
        t.mode = c->recmap.mapped;
        t.L = c->M2;
        t.l2L = c->log2M2;
        t.lvl_in = lvl;
        t.nlevels = c->levels;
        t.size_log2 = c->size_log2;
        t.Q = c->d_q;
        t.q_stride = c->q_stride;
        t.R = c->R;
        t.Pout = c->d_pscr[1];
        t.out_stride = c->p_stride;
        ProfScope ps(c, K_TAIL, c->side);
        const dim3 grid((unsigned)(c->M2 / 64), (unsigned)nframes);
        if (ng == 64)
            hipLaunchKernelGGL(k_col_tail<64>, grid, dim3(64), 0, c->side, t);
        else if (ng == 128)
            hipLaunchKernelGGL(k_col_tail<128>, grid, dim3(64), 0, c->side, t);
        else
            hipLaunchKernelGGL(k_col_tail<256>, grid, dim3(64), 0, c->side, t);
        HIPCHK(hipGetLastError());
        lvl += ilog2((size_t)ng);
        len = (size_t)c->M2;
        cur = 1;
    }
    const int lvl_mapped = col_tail ? -1 : c->LT;  // the level whose sums are still in RecMap order
    while (lvl + 1 < c->levels && len >= 2) {
        TailArgs t{};
        t.Pin = c->d_pscr[cur];
        t.in_stride = c->p_stride;
        t.len_in = len;
        t.lvl_in = lvl;
        t.nlevels = c->levels;
        t.size_log2 = c->size_log2;
        t.Q = c->d_q;
        t.q_stride = c->q_stride;
        t.R = c->R;
        t.Pout = c->d_pscr[cur ^ 1];
        t.out_stride = c->p_stride;
        t.map = c->recmap;
        if (lvl != lvl_mapped) t.map.mapped = 0;  // only pass 2's own output is tile-major
        ProfScope ps(c, K_TAIL, c->side);
        const unsigned nb = (unsigned)((len / 2 + 255) / 256);
        hipLaunchKernelGGL(k_pyramid_tail, dim3(nb, nframes), dim3(256), 0, c->side, t);
        HIPCHK(hipGetLastError());
        lvl += 7;
        len >>= 7;
        cur ^= 1;
    }
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_side_done, c->side));
        c->side_pending = true;
        HIPCHK(hipEventRecord(c->ev_set_done[c->cur_set], c->side));
        c->set_pending[c->cur_set] = true;
    }
    c->last_nframes = nframes;
    c->out_valid = c->q_valid = false;
    std::fill(c->q_untiled.begin(), c->q_untiled.end(), 0);
    return PSDR_OK;
}

void free_all(psdr_ctx *c) {
    auto F = [](void *p) {
        if (p) hipFree(p);
    };
    F(c->d_Wl1);
    if (c->d_Wl2 != c->d_Wl1) F(c->d_Wl2);
    F(c->d_TA);
    F(c->d_trace);
    F(c->d_kclk);
    F(c->d_TB);
    F(c->d_UA);
    F(c->d_UB);
    F(c->d_UG);
    F(c->ring.d);
    for (auto e : c->ring.ev_written)
        if (e) hipEventDestroy(e);
    for (auto e : c->ring.ev_read)
        if (e) hipEventDestroy(e);
    if (c->ring.copy) hipStreamDestroy(c->ring.copy);
    for (int st = 0; st < 2; st++) {
        F(c->seam_pool[st][0]);
        F(c->seam_pool[st][1]);
    }
    F(c->d_tickets[0]);
    F(c->d_tickets[1]);
    F(c->y_pool[0]);
    F(c->y_pool[1]);
    F(c->d_Z);
    for (int s = 0; s < 2; s++) {
        F(c->spec_pool[s]);
        F(c->q_pool[s]);
        F(c->qt_pool[s]);
        F(c->pscr_pool[s][0]);
        F(c->pscr_pool[s][1]);
        if (c->ev_set_done[s]) hipEventDestroy(c->ev_set_done[s]);
    }
    F(c->d_stage);
    F(c->d_Wn);
    F(c->d_stage_tab);
    F(c->d_ypost);
    F(c->d_gscratch);
    F(c->d_bb_tail);
    F(c->d_bb_last);
    for (void *q : c->post_allocs) hipFree(q);
    F(c->d_pwr);
    F(c->d_audio);
    F(c->d_real_prev);
    F(c->d_nan);
    c->client_ring.destroy();
    c->wf_ring.destroy();
    F(c->d_wfout);
    auto H = [](void *p) {
        if (p) hipHostFree(p);
    };
    H(c->h_out);
    H(c->h_q);
    H(c->h_audio);
    H(c->h_pwr);
    H(c->h_nan);
    H(c->h_pcm);
    for (auto &p : c->pending) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    for (auto e : c->pool) hipEventDestroy(e);
    if (c->t0) hipEventDestroy(c->t0);
    if (c->t1) hipEventDestroy(c->t1);
    if (c->ev_fft_done) hipEventDestroy(c->ev_fft_done);
    if (c->ev_side_done) hipEventDestroy(c->ev_side_done);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    if (c->own_p1) hipStreamDestroy(c->own_p1);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    for (int i = 0; i < 2; i++) {
        if (c->ev_p1[i]) hipEventDestroy(c->ev_p1[i]);
        if (c->ev_p2[i]) hipEventDestroy(c->ev_p2[i]);
    }
    if (c->own_side) hipStreamDestroy(c->own_side);
    if (c->side2) hipStreamDestroy(c->side2);
    if (c->side3) hipStreamDestroy(c->side3);
    if (c->ev_demod) hipEventDestroy(c->ev_demod);
    if (c->ev_gather) hipEventDestroy(c->ev_gather);
    for (int i = 0; i < 2; i++)
        if (c->ev_want[i]) hipEventDestroy(c->ev_want[i]);
    for (int i = 0; i < 2; i++) {
        if (c->ev_s1[i]) hipEventDestroy(c->ev_s1[i]);
        if (c->ev_s2[i]) hipEventDestroy(c->ev_s2[i]);
    }
}

int build(psdr_ctx *c) {
    const psdr_config &g = c->cfg;
    HIPCHK(hipSetDevice(c->device));
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c->device));
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    {
        // the consumers are short kernels that must squeeze in next to the persistent FFT
        // work-groups: give their stream the highest priority
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&c->own_side, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipStreamCreateWithFlags(&c->own_p1, hipStreamNonBlocking));
    c->stream = c->own_stream;
    c->side = c->own_side;
    c->static_tiles = getenv("PSDR_STATIC_TILES") != nullptr;
    c->no_col_tail = getenv("PSDR_NO_COL_TAIL") != nullptr;
    // pass 1 on its own stream overlaps the two passes of consecutive batches; it pays only when
    // both batches' intermediates fit the 256 MiB MALL together (measured: F=16 2^20-point frames
    // lose 12 %, F>=32 gain nothing), so it is opt-in
    c->no_p1_stream = getenv("PSDR_P1_STREAM") == nullptr;
    if (const char *e = getenv("PSDR_P1_GRID")) c->p1_grid = (unsigned)atoi(e) & ~7u;
    if (const char *e = getenv("PSDR_P2_GRID")) c->p2_grid = (unsigned)atoi(e) & ~7u;
    c->p1 = c->no_p1_stream ? c->own_stream : c->own_p1;
    HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_p1[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_p2[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fft_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->t0));
    HIPCHK(hipEventCreate(&c->t1));

    // The Hann window (build_hann_window, src/utils/dsp.cpp:6-11) is evaluated inside pass 1
    // from the twiddle tables; only W_N^1 (odd real samples) is needed on top of them.
    {
        const double ang = -2.0 * M_PI / (double)c->N;
        c->wdelta = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    // ---- twiddles
    {
        int rc = upload(&c->d_Wl1, make_twiddles((size_t)c->M1, 1, (size_t)c->M1, -1));
        if (rc) return rc;
        if (c->M2 == c->M1) {
            c->d_Wl2 = c->d_Wl1;
        } else {
            rc = upload(&c->d_Wl2, make_twiddles((size_t)c->M2, 1, (size_t)c->M2, -1));
            if (rc) return rc;
        }
        // inter-pass twiddle W_M^e = W_M1^{e >> log2M2} * W_M^{e & (M2-1)}: the first factor
        // is the pass-1 stage table, the second has M2 entries
        rc = upload(&c->d_TB, make_twiddles((size_t)c->M2, 1, c->M, -1));
        if (rc) return rc;
        if (c->is_real) {
            c->log2UB = std::min(10, ilog2(c->N));
            const size_t UB = (size_t)1 << c->log2UB;
            rc = upload(&c->d_UA, make_twiddles(c->N / UB + 1, UB, c->N, -1));
            if (rc) return rc;
            rc = upload(&c->d_UB, make_twiddles(UB, 1, c->N, -1));
            if (rc) return rc;
            if (c->real_fused) {
                rc = upload(&c->d_UG, make_twiddles((size_t)(c->M1 / 16), 8, c->N, -1));
                if (rc) return rc;
            }
        }
    }
#ifdef PSDR_TRACE_ON
    HIPCHK(hipMalloc((void **)&c->d_trace, 4864 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(c->d_trace, 0, 4864 * sizeof(unsigned long long)));
#endif
    // ---- work buffers
    const size_t F = (size_t)c->max_batch;
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc((void **)&c->d_tickets[i], TICKET_SLOTS * 8 * sizeof(unsigned)));
        HIPCHK(hipMemset(c->d_tickets[i], 0, TICKET_SLOTS * 8 * sizeof(unsigned)));
    }
    // the second Y buffer only exists when pass 1 runs on its own stream (PSDR_P1_STREAM)
    for (int i = 0; i < (c->no_p1_stream ? 1 : 2); i++)
        HIPCHK(hipMalloc((void **)&c->y_pool[i], F * c->M * sizeof(cf)));
    if (c->is_real && !c->real_fused) HIPCHK(hipMalloc((void **)&c->d_Z, F * c->M * sizeof(cf)));
    if (!c->is_real && c->lay.mode) HIPCHK(hipMalloc((void **)&c->d_Z, (c->M + 2) * sizeof(cf)));  // k-order staging
    if (c->real_fused) {
        // one frame of k-order staging for psdr_read_spectrum / psdr_get_output_buffer
        HIPCHK(hipMalloc((void **)&c->d_Z, (c->M + 2) * sizeof(cf)));
        if (const char *e = getenv("PSDR_SEG_LEN")) c->seg_len_env = atoi(e);
        c->y_blocked = getenv("PSDR_REAL_YBLOCKED") != nullptr;
        size_t cap = 0;
        for (int nf = 1; nf <= c->max_batch; nf++)
            cap = std::max(cap, (size_t)nf * (size_t)((c->M1 / 16) / real_seg_len(c, nf)));
        c->seam_cap = cap;
        for (int st = 0; st < 2; st++) {  // part of the double-buffered result sets: k_real_seam is a consumer
            HIPCHK(hipMalloc((void **)&c->seam_pool[st][0], cap * (size_t)c->M2 * 8 * sizeof(float)));
            HIPCHK(hipMalloc((void **)&c->seam_pool[st][1], cap * (size_t)c->M2 * sizeof(float)));
        }
    }
    for (int s = 0; s < 2; s++) {
        HIPCHK(hipMalloc((void **)&c->spec_pool[s], F * c->spec_stride * sizeof(cf)));
        HIPCHK(hipMemset(c->spec_pool[s], 0, F * c->spec_stride * sizeof(cf)));
        HIPCHK(hipMalloc((void **)&c->q_pool[s], F * c->q_stride));
        HIPCHK(hipMemset(c->q_pool[s], 0, F * c->q_stride));
        if (c->tiled_lt >= 0) {
            HIPCHK(hipMalloc((void **)&c->qt_pool[s], F * c->qt_stride));
            HIPCHK(hipMemset(c->qt_pool[s], 0, F * c->qt_stride));
        }
        HIPCHK(hipMalloc((void **)&c->pscr_pool[s][0], F * c->p_stride * sizeof(float)));
        HIPCHK(hipMalloc((void **)&c->pscr_pool[s][1], F * c->p_stride * sizeof(float)));
        HIPCHK(hipEventCreateWithFlags(&c->ev_set_done[s], hipEventDisableTiming));
    }
    select_set(c, 0);
    c->q_untiled.assign(F, 0);
    // ---- level-1 staging
    HIPCHK(hipMalloc((void **)&c->d_stage, (c->is_real ? c->N : 2 * c->N) * sizeof(float)));
    {
        const size_t nb = c->is_real ? (c->N / 2 + 1) : (c->N + (size_t)g.additional_size);
        HIPCHK(hipHostMalloc((void **)&c->h_out, nb * sizeof(cf), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&c->h_q, std::max<size_t>(c->q_len, 16), hipHostMallocDefault));
    }
    // ---- audio clients
    c->n = g.audio_fft_size;
    if (c->n > 0) {
        const int n = c->n;
        // factorise n, grouping prime factors into radices <= 16 (fewer barriers)
        std::vector<int> primes;
        int m = n;
        for (int p = 2; (long long)p * p <= m; p++)
            while (m % p == 0) {
                primes.push_back(p);
                m /= p;
            }
        if (m > 1) primes.push_back(m);
        std::vector<int> rad;
        int cur = 1;
        for (int p : primes) {
            if (cur * p <= 16)
                cur *= p;
            else {
                if (cur > 1) rad.push_back(cur);
                cur = p;
            }
        }
        if (cur > 1) rad.push_back(cur);
        if ((int)rad.size() > PSDR_MAX_STAGES)
            return fail(PSDR_ERR_UNSUPPORTED, "audio_fft_size %d has too many factors", n);
        c->nstages = (int)rad.size();
        for (int i = 0; i < c->nstages; i++) c->radix[i] = rad[i];
        const size_t cap = 144 * 1024;
        if ((size_t)n * 24 <= cap) {
            c->lds_mode = 0;
            c->idft_lds = (size_t)n * 24;
        } else if ((size_t)n * 16 <= cap) {
            c->lds_mode = 1;
            c->idft_lds = (size_t)n * 16;
        } else {
            c->lds_mode = 2;
            c->idft_lds = 0;
        }
        if (c->idft_lds > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)k_demod_idft,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->idft_lds));
        int rc = upload(&c->d_Wn, make_twiddles((size_t)n, 1, (size_t)n, +1));
        if (rc) return rc;
        {
            // stage tables of the generic-radix Stockham: output o = s*(n/R) + i of a stage with
            // radix R and p = product of the earlier radices reads x[i + q*n/R] and writes
            // y[j + s*p], j = (i - i%p)*R + i%p, with twiddle exponent q*e1, e1 = (i%p + s*p)*n/(p*R)
            std::vector<int4> tab((size_t)c->nstages * n);
            int pp = 1;
            for (int st = 0; st < c->nstages; st++) {
                const int R = c->radix[st], tlen = n / R, step = n / (pp * R);
                for (int o = 0; o < n; o++) {
                    const int s = o / tlen, i = o - s * tlen, k = i % pp, j = (i - k) * R + k;
                    const long long e1 = ((long long)(k + s * pp) * step) % n;
                    tab[(size_t)st * n + o] = make_int4(i, j + s * pp, (int)e1, 0);
                }
                pp *= R;
            }
            rc = upload(&c->d_stage_tab, tab);
            if (rc) return rc;
            c->idft_threads = n <= 512 ? 128 : 256;
            c->idft_block = getenv("PSDR_IDFT_BLOCK") != nullptr;
            c->idft_generic = getenv("PSDR_IDFT_GENERIC") != nullptr;
            if (const char *e = getenv("PSDR_DEMOD_CHAIN")) c->demod_chain = atoi(e) != 0;
            if (const char *e = getenv("PSDR_DEMOD_K")) c->demod_chain_k = std::max(1, atoi(e));
        }
        const size_t S = (size_t)std::max(1, g.max_clients);
        c->aslots.resize(S);
        HIPCHK(hipMalloc((void **)&c->d_ypost, S * F * n * sizeof(cf)));
        HIPCHK(hipMalloc((void **)&c->d_pwr, S * F * sizeof(float)));
        HIPCHK(hipMalloc((void **)&c->d_audio, S * F * (n / 2) * sizeof(float)));
        HIPCHK(hipMalloc((void **)&c->d_nan, S * F * sizeof(int)));
        HIPCHK(hipMalloc((void **)&c->d_real_prev, 2 * S * (n / 2) * sizeof(float)));
        HIPCHK(hipMalloc((void **)&c->d_bb_tail, 2 * S * (n / 2) * sizeof(cf)));
        HIPCHK(hipMalloc((void **)&c->d_bb_last, 2 * S * sizeof(cf)));
        HIPCHK(hipMemset(c->d_real_prev, 0, 2 * S * (n / 2) * sizeof(float)));
        HIPCHK(hipMemset(c->d_bb_tail, 0, 2 * S * (n / 2) * sizeof(cf)));
        HIPCHK(hipMemset(c->d_bb_last, 0, 2 * S * sizeof(cf)));
        HIPCHK(hipMemset(c->d_audio, 0, S * F * (n / 2) * sizeof(float)));
        HIPCHK(hipMemset(c->d_pwr, 0, S * F * sizeof(float)));
        HIPCHK(hipMemset(c->d_nan, 0, S * F * sizeof(int)));
        if (c->lds_mode == 2) HIPCHK(hipMalloc((void **)&c->d_gscratch, S * F * 2 * n * sizeof(cf)));
        if (c->client_ring.init(S * sizeof(ClientParams)))
            return fail(PSDR_ERR_HIP, "client parameter ring allocation failed");
    }
    // ---- waterfall clients
    {
        const size_t W = (size_t)std::max(1, g.max_waterfall_clients);
        c->wslots.resize(W);
        c->wf_sent_off = (W * sizeof(WfClient) + 63) & ~(size_t)63;
        if (c->wf_ring.init(c->wf_sent_off + F * sizeof(int)))
            return fail(PSDR_ERR_HIP, "waterfall parameter ring allocation failed");
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipDeviceSynchronize());
    return PSDR_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------
extern "C" int psdr_create(const psdr_config *cfg, psdr_ctx **out) {
    if (!cfg || !out) return fail(PSDR_ERR_INVALID, "null argument");
    if (cfg->struct_size != sizeof(psdr_config))
        return fail(PSDR_ERR_INVALID, "psdr_config.struct_size mismatch (%u vs %zu)",
                    cfg->struct_size, sizeof(psdr_config));
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(PSDR_ERR_NO_DEVICE, "No HIP devices found");
    if (cfg->device < 0 || cfg->device >= count)
        return fail(PSDR_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, count);
    const size_t N = cfg->fft_size;
    if (N == 0 || (N & (N - 1))) return fail(PSDR_ERR_INVALID, "fft_size must be a power of two");
    const bool is_real = cfg->is_real != 0;
    const size_t M = is_real ? N / 2 : N;
    const int m = ilog2(M);
    if (m < 12 || m > 22)
        return fail(PSDR_ERR_UNSUPPORTED,
                    "fft_size %zu unsupported: complex transform length must be 2^12..2^22", N);
    if (cfg->downsample_levels < 1 || ((M >> (cfg->downsample_levels - 1)) < 1))
        return fail(PSDR_ERR_INVALID, "downsample_levels %d invalid", cfg->downsample_levels);
    if (cfg->audio_fft_size < 0 || (cfg->audio_fft_size % 4) != 0)
        return fail(PSDR_ERR_INVALID, "audio_fft_size must be a non-negative multiple of 4");
    if (cfg->input_format < PSDR_FMT_U8 || cfg->input_format > PSDR_FMT_F64)
        return fail(PSDR_ERR_INVALID, "unknown input_format %d", cfg->input_format);
    if (cfg->max_batch < 1) return fail(PSDR_ERR_INVALID, "max_batch must be >= 1");

    psdr_ctx *c = new (std::nothrow) psdr_ctx();
    if (!c) return fail(PSDR_ERR_NOMEM, "out of memory");
    c->cfg = *cfg;
    c->device = cfg->device;
    c->N = N;
    c->M = M;
    c->is_real = is_real;
    c->R = M;  // fft_result_size: N (IQ) or N/2 (real), src/spectrumserver.cpp:99-105
    c->log2M2 = m / 2;
    if (const char *e = getenv("PSDR_LOG2M2")) c->log2M2 = atoi(e);  // tuning: split M = M1 * M2
    c->log2M1 = m - c->log2M2;
    c->M1 = 1 << c->log2M1;
    c->M2 = 1 << c->log2M2;
    c->T1 = pick_T(c->M1, c->M2);
    if (const char *e = getenv("PSDR_T1")) c->T1 = std::min(c->T1, std::max(8, atoi(e)));  // tuning: narrower pass-1 tiles, several work-groups per CU
    c->T2 = pick_T(c->M2, c->M1);
    c->size_log2 = (int)std::lround(std::log2((double)N)) + cfg->brightness_offset;
    c->levels = cfg->downsample_levels;
    c->max_batch = cfg->max_batch;
    c->spec_stride = is_real ? (M + 2) : N;
    c->q_len = 0;
    for (int i = 0; i < c->levels; i++) c->q_len += c->R >> i;
    c->q_stride = (c->q_len + 127) & ~(size_t)127;
    c->real_fused = is_real && c->M2 == 1024 && c->T2 == 16 && (c->M1 == 1024 || c->M1 == 2048) &&
                    getenv("PSDR_REAL_3PASS") == nullptr;
    if (c->real_fused) {
        c->tile_ch = 8;  // octet records, levels 0..3 (quantize.h, RecMap mode 2)
        c->LT = 3;
        c->tiled_lt = 3;
        c->recmap.l2tpr = ilog2((size_t)(c->M1 / 8));
        c->recmap.l2gpt = 0;
        c->recmap.l2rows = c->log2M2;
        c->recmap.mapped = 2;
        c->qt_stride = 2 * c->R;
        c->lay.mode = 2;
        c->lay.m1 = c->M1;
        c->lay.l2m1 = c->log2M1;
        c->lay.L = c->M2;
        c->lay.l2L = c->log2M2;
    } else if (is_real) {
        c->LT = 8;  // the untangle kernel finishes levels 0..8 (4 bins per lane, 64 lanes)
        c->tiled_lt = -1;
    } else {
        c->tile_ch = (c->T2 >= 16) ? 16 : 8;
        c->LT = (c->T2 >= 16) ? 4 : 3;
        c->tiled_lt = c->LT;
        c->recmap.l2tpr = ilog2((size_t)(c->M1 / c->tile_ch));
        c->recmap.l2gpt = ilog2((size_t)(c->T2 / c->tile_ch));
        c->recmap.l2rows = c->log2M2;
        c->recmap.mapped = 1;
        c->qt_stride = 2 * c->R;  // R/CH records of 2*CH bytes
        if (c->M2 == 1024 && c->T2 == 16) {  // k_fft_pass2<1024, 16, true, *> writes tile-major lines
            c->lay.mode = 1;
            c->lay.m1 = c->M1;
            c->lay.l2m1 = c->log2M1;
            c->lay.L = c->M2;
            c->lay.l2L = c->log2M2;
        }
    }
    // opt-in (PSDR_P1_WAVE=1): measured 640-670 us against the barrier kernel's 600-615 us per 256 frames
    c->p1_wave = !is_real && c->M1 == 1024 && c->M2 == 1024 && c->T1 == 16 && c->T2 == 16 && getenv("PSDR_P1_WAVE") != nullptr;
    c->p_stride = std::max<size_t>(c->R >> c->LT, 64);
    if (cfg->skip_num < 1) c->cfg.skip_num = 1;
    if (cfg->waterfall_size < 0) {
        delete c;
        return fail(PSDR_ERR_INVALID, "waterfall_size must be >= 0");
    }
    c->min_waterfall_fft = cfg->waterfall_size > 0 ? cfg->waterfall_size : (int)(c->R >> (c->levels - 1));

    int rc = build(c);
    if (rc) {
        free_all(c);
        delete c;
        return rc;
    }
    *out = c;
    return PSDR_OK;
}

extern "C" void psdr_destroy(psdr_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->side) hipStreamSynchronize(c->side);
    free_all(c);
    delete c;
}

// ---- level 1 ---------------------------------------------------------------------------
extern "C" int psdr_host_alloc(psdr_ctx *c, size_t nfloats, float **out) {
    if (!out) return fail(PSDR_ERR_INVALID, "null argument");
    // ctx may be NULL: the reference allocates its half-frame buffers before planning
    // (src/fft.cpp:17-29), i.e. before the back-end knows whether the input is real
    if (c) HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipHostMalloc((void **)out, std::max<size_t>(nfloats, 1) * sizeof(float),
                         hipHostMallocDefault));
    return PSDR_OK;
}
extern "C" int psdr_host_free(psdr_ctx *, float *buf) {
    if (buf) HIPCHK(hipHostFree(buf));
    return PSDR_OK;
}
static int load_input(psdr_ctx *c, const float *a1, const float *a2) {
    if (!c || !a1 || !a2) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const size_t half_floats = c->is_real ? c->N / 2 : c->N;
    HIPCHK(hipMemcpyAsync(c->d_stage, a1, half_floats * sizeof(float), hipMemcpyHostToDevice,
                          c->stream));
    HIPCHK(hipMemcpyAsync(c->d_stage + half_floats, a2, half_floats * sizeof(float),
                          hipMemcpyHostToDevice, c->stream));
    c->loaded = true;
    c->input_on_main = true;
    return PSDR_OK;
}
extern "C" int psdr_load_real_input(psdr_ctx *c, const float *a1, const float *a2) {
    if (c && !c->is_real) return fail(PSDR_ERR_STATE, "context was planned for complex input");
    return load_input(c, a1, a2);
}
extern "C" int psdr_load_complex_input(psdr_ctx *c, const float *a1, const float *a2) {
    if (c && c->is_real) return fail(PSDR_ERR_STATE, "context was planned for real input");
    return load_input(c, a1, a2);
}
extern "C" int psdr_execute(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (!c->loaded) return fail(PSDR_ERR_STATE, "execute() before load_*_input()");
    HIPCHK(hipSetDevice(c->device));
    int rc = process_frames(c, c->d_stage, 1, PSDR_FMT_F32);
    if (rc) return rc;
    rc = drain(c);
    if (rc) return rc;
    c->executed = true;
    return PSDR_OK;
}

// make the level-major copy of frame `frame`'s pyramid current (levels 0..LT live in tiled
// records on the device; the reference's layout is produced on demand)
static int ensure_level_major(psdr_ctx *c, int frame) {
    if (c->tiled_lt < 0 || c->q_untiled[frame]) return PSDR_OK;
    const size_t nrec = c->R / (size_t)c->tile_ch;
    hipLaunchKernelGGL(k_untile_q, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->side,
                       c->d_qt + (size_t)frame * c->qt_stride, c->d_q + (size_t)frame * c->q_stride, c->R,
                       c->tile_ch, c->tiled_lt, c->levels, c->recmap);
    HIPCHK(hipGetLastError());
    c->q_untiled[frame] = 1;
    return PSDR_OK;
}

// spectrum of `frame` to host in the reference's k order
static int copy_spectrum_k_order(psdr_ctx *c, int frame, cf *dst) {
    const cf *src = c->d_spec + (size_t)frame * c->spec_stride;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    if (c->lay.mode) {
        // the device keeps the frame in tile-major lines (SpecLayout): k order through one frame of staging
        hipLaunchKernelGGL(k_spec_k_order, dim3((unsigned)((c->M + 1 + 255) / 256)), dim3(256), 0, c->stream, src, c->d_Z,
                           c->M, c->is_real ? 1 : 0, c->lay);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dst, c->d_Z, (c->is_real ? c->N / 2 + 1 : c->N) * sizeof(cf), hipMemcpyDeviceToHost, c->stream));
    } else if (c->is_real) {
        HIPCHK(hipMemcpyAsync(dst, src, (c->N / 2 + 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
    } else {
        // client order c -> bin k = (c + N/2 + 1) mod N: two contiguous runs
        const size_t N = c->N, h = N / 2;
        HIPCHK(hipMemcpyAsync(dst, src + (h - 1), (h + 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
        HIPCHK(hipMemcpyAsync(dst + h + 1, src, (h - 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
extern "C" int psdr_get_output_buffer(psdr_ctx *c, float **out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (c->last_nframes > 0 && !c->out_valid) {
        int rc = copy_spectrum_k_order(c, 0, (cf *)c->h_out);
        if (rc) return rc;
        if (!c->is_real && c->cfg.additional_size > 0)  // wrap copy, src/fft.cpp:91-98
            memcpy((cf *)c->h_out + c->N, c->h_out, sizeof(cf) * (size_t)c->cfg.additional_size);
        c->out_valid = true;
    }
    *out = c->h_out;
    return PSDR_OK;
}
extern "C" int psdr_get_quantized_buffer(psdr_ctx *c, int8_t **out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (c->last_nframes > 0 && !c->q_valid) {
        {
            int rc = drain(c);
            if (rc) return rc;
            rc = ensure_level_major(c, 0);
            if (rc) return rc;
            rc = drain(c);
            if (rc) return rc;
        }
        HIPCHK(hipMemcpyAsync(c->h_q, c->d_q, c->q_len, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->q_valid = true;
    }
    *out = c->h_q;
    return PSDR_OK;
}

// ---- device helpers ----------------------------------------------------------------------
extern "C" int psdr_dev_alloc(psdr_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(out, std::max<size_t>(bytes, 16)));
    return PSDR_OK;
}
extern "C" int psdr_dev_free(psdr_ctx *c, void *p) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (p) HIPCHK(hipFree(p));
    return PSDR_OK;
}
extern "C" int psdr_memcpy_h2d(psdr_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || !dst || !src) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
extern "C" int psdr_memcpy_d2h(psdr_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || !dst || !src) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
extern "C" int psdr_synchronize(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    return drain(c);
}
extern "C" size_t psdr_half_frame_bytes(const psdr_ctx *c) {
    if (!c) return 0;
    return (c->N / 2) * (c->is_real ? 1 : 2) * fmt_bytes(c->cfg.input_format);
}

// ---- level 2 -----------------------------------------------------------------------------
extern "C" int psdr_process_batch(psdr_ctx *c, const void *d_halves, int nframes) {
    if (!c || !d_halves) return fail(PSDR_ERR_INVALID, "null argument");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    HIPCHK(hipSetDevice(c->device));
    return process_frames(c, d_halves, nframes, c->cfg.input_format);
}

// ---- streaming ingest (src/fft.cpp:56-67, src/samplereader.cpp:42-70 on the device) -----------------
extern "C" int psdr_ring_create(psdr_ctx *c, int nhalves) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (nhalves < 2) return fail(PSDR_ERR_INVALID, "a ring needs at least 2 half-frames");
    if (c->ring.d) return fail(PSDR_ERR_STATE, "the context already has an ingest ring");
    HIPCHK(hipSetDevice(c->device));
    auto &r = c->ring;
    r.hb = psdr_half_frame_bytes(c);
    r.nhalves = nhalves;
    HIPCHK(hipMalloc((void **)&r.d, (size_t)(nhalves + 1) * r.hb));
    HIPCHK(hipMemset(r.d, 0, (size_t)(nhalves + 1) * r.hb));
    HIPCHK(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
    r.ev_written.assign(nhalves, nullptr);
    r.ever_written.assign(nhalves, 0);
    r.reader_seq.assign(nhalves, 0);
    for (auto &e : r.ev_written) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : r.ev_read) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PSDR_OK;
}
extern "C" int psdr_ring_write_async(psdr_ctx *c, uint64_t half_index, const void *host_half) {
    if (!c || !host_half) return fail(PSDR_ERR_INVALID, "null argument");
    auto &r = c->ring;
    if (!r.d) return fail(PSDR_ERR_STATE, "psdr_ring_create() first");
    HIPCHK(hipSetDevice(c->device));
    const int slot = (int)(half_index % (uint64_t)r.nhalves);
    // the slot's previous content may still be waiting for its reader (a process call issued at
    // most NEV calls ago: older ones were waited for when their event was reused)
    const uint64_t rs = r.reader_seq[slot];
    if (rs && r.seq - rs < (uint64_t)psdr_ctx::IngestRing::NEV)
        HIPCHK(hipStreamWaitEvent(r.copy, r.ev_read[rs % psdr_ctx::IngestRing::NEV], 0));
    if (slot == 0 && r.reader_seq[r.nhalves - 1]) {  // the mirror of slot 0 is read with the LAST slot's frame
        const uint64_t rl = r.reader_seq[r.nhalves - 1];
        if (r.seq - rl < (uint64_t)psdr_ctx::IngestRing::NEV)
            HIPCHK(hipStreamWaitEvent(r.copy, r.ev_read[rl % psdr_ctx::IngestRing::NEV], 0));
    }
    HIPCHK(hipMemcpyAsync(r.d + (size_t)slot * r.hb, host_half, r.hb, hipMemcpyHostToDevice, r.copy));
    if (slot == 0)
        HIPCHK(hipMemcpyAsync(r.d + (size_t)r.nhalves * r.hb, host_half, r.hb, hipMemcpyHostToDevice, r.copy));
    HIPCHK(hipEventRecord(r.ev_written[slot], r.copy));
    r.ever_written[slot] = 1;
    return PSDR_OK;
}
extern "C" int psdr_ring_wait(psdr_ctx *c, uint64_t half_index) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    auto &r = c->ring;
    if (!r.d) return fail(PSDR_ERR_STATE, "psdr_ring_create() first");
    const int slot = (int)(half_index % (uint64_t)r.nhalves);
    if (r.ever_written[slot]) HIPCHK(hipEventSynchronize(r.ev_written[slot]));
    return PSDR_OK;
}
extern "C" int psdr_process_ring(psdr_ctx *c, uint64_t first_half, int nframes) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    auto &r = c->ring;
    if (!r.d) return fail(PSDR_ERR_STATE, "psdr_ring_create() first");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    const int s0 = (int)(first_half % (uint64_t)r.nhalves);
    if (s0 + nframes > r.nhalves)
        return fail(PSDR_ERR_INVALID, "frames %d..%d cross the end of the %d-half ring (one guard half-frame): split the batch",
                    s0, s0 + nframes - 1, r.nhalves);
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i <= nframes; i++) {  // halves s0 .. s0+nframes (the last may be the mirror of slot 0)
        const int slot = (s0 + i) % r.nhalves;
        if (!r.ever_written[slot]) return fail(PSDR_ERR_STATE, "half-frame slot %d was never written", slot);
        HIPCHK(hipStreamWaitEvent(c->p1, r.ev_written[slot], 0));
    }
    r.seq++;
    hipEvent_t ev = r.ev_read[r.seq % psdr_ctx::IngestRing::NEV];
    HIPCHK(hipEventSynchronize(ev));  // the call that used it NEV calls ago (no-op if never recorded)
    int rc = process_frames(c, r.d + (size_t)s0 * r.hb, nframes, c->cfg.input_format, ev);
    if (rc) return rc;
    for (int i = 0; i <= nframes; i++) r.reader_seq[(s0 + i) % r.nhalves] = r.seq;
    return PSDR_OK;
}

static int drain(psdr_ctx *c) {
    if (c->ring.copy) HIPCHK(hipStreamSynchronize(c->ring.copy));
    if (c->p1 != c->stream) HIPCHK(hipStreamSynchronize(c->p1));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->side != c->stream) HIPCHK(hipStreamSynchronize(c->side));
    if (c->side2) HIPCHK(hipStreamSynchronize(c->side2));
    if (c->side3) HIPCHK(hipStreamSynchronize(c->side3));
    return PSDR_OK;
}
static int check_slot(psdr_ctx *c, int id) {
    if (id < 0 || id >= (int)c->aslots.size() || !c->aslots[id].active)
        return fail(PSDR_ERR_INVALID, "no audio client with id %d", id);
    return PSDR_OK;
}
extern "C" int psdr_client_add(psdr_ctx *c, int *id_out) {
    if (!c || !id_out) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    std::lock_guard<std::mutex> lk(c->mtx);
    HIPCHK(hipSetDevice(c->device));
    for (size_t i = 0; i < c->aslots.size(); i++)
        if (!c->aslots[i].active) {
            AudioSlot &s = c->aslots[i];
            s = AudioSlot();
            s.active = true;
            // a fresh AudioClient starts from zeroed buffers (src/signal.h:42-51)
            const size_t S = c->aslots.size(), h = (size_t)c->n / 2;
            for (int b = 0; b < 2; b++) {
                HIPCHK(hipMemsetAsync(c->d_real_prev + ((size_t)b * S + i) * h, 0, h * sizeof(float),
                                      c->side));
                HIPCHK(hipMemsetAsync(c->d_bb_tail + ((size_t)b * S + i) * h, 0, h * sizeof(cf),
                                      c->side));
                HIPCHK(hipMemsetAsync(c->d_bb_last + ((size_t)b * S + i), 0, sizeof(cf), c->side));
            }
            *id_out = (int)i;
            return PSDR_OK;
        }
    return fail(PSDR_ERR_NOMEM, "all %zu audio client slots are in use", c->aslots.size());
}
extern "C" int psdr_client_remove(psdr_ctx *c, int id) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    c->aslots[id].active = false;
    return PSDR_OK;
}
extern "C" int psdr_client_set_audio_range(psdr_ctx *c, int id, int l, double mid, int r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    // the reference does not validate here (src/signal.cpp:81-94); a range outside the
    // spectrum would read out of bounds there, so it is refused here
    if (l < 0 || r < l || (size_t)r > c->R || r - l > c->n)
        return fail(PSDR_ERR_INVALID, "range [%d,%d) outside the spectrum or wider than %d", l, r, c->n);
    AudioSlot &s = c->aslots[id];
    s.l = l;
    s.r = r;
    s.mid = mid;
    return PSDR_OK;
}
extern "C" int psdr_client_on_window_message(psdr_ctx *c, int id, int l, double mid, int r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
    }
    const int R = (int)c->R;  // src/signal.cpp:305-311
    if (l < 0 || l >= R || r < 0 || r >= R || l > r)
        return fail(PSDR_ERR_INVALID, "window [%d,%d] rejected", l, r);
    if (r - l > c->n) return fail(PSDR_ERR_INVALID, "window wider than audio_fft_size");
    return psdr_client_set_audio_range(c, id, l, mid, r);
}
// signal_loop's slow-client rule (src/websocket.cpp:170-176): a client with more than 50 kB queued on its socket gets no
// send_audio call for the frame - nothing of its state moves (src/signal.cpp:200-203, 273-284).  A paused client sits
// out every demodulation batch until it is resumed; its results read as PSDR_ERR_NO_DATA meanwhile.
extern "C" int psdr_client_set_paused(psdr_ctx *c, int id, int paused) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    c->aslots[id].paused = paused != 0;
    return PSDR_OK;
}
extern "C" int psdr_client_set_audio_demodulation(psdr_ctx *c, int id, int mode) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    if (mode < PSDR_USB || mode > PSDR_FM) return fail(PSDR_ERR_INVALID, "unknown mode %d", mode);
    c->aslots[id].mode = mode;
    if (c->aslots[id].agc_reset == 0) c->aslots[id].agc_reset = 1;  // src/signal.cpp:316-328: resets the AGC
    return PSDR_OK;
}

// band != nullptr: `spec` is a window of bins [band[0], band[0] + band[1]) per frame - linear, or (band_tiled) one
// band region of a banded spectrum (SpecLayout mode 4)
static int demod_impl(psdr_ctx *c, const cf *spec, size_t spec_stride, int nframes, uint64_t first_frame_num,
                      const uint32_t *band = nullptr, bool band_tiled = false) {
    if (c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    HIPCHK(hipSetDevice(c->device));
    int nact = 0, npaused = 0;
    const int ring = c->client_ring.acquire();
    if (ring < 0) return fail(PSDR_ERR_HIP, "client parameter ring: event wait failed");
    ClientParams *h_clients = (ClientParams *)c->client_ring.host(ring);
    ClientParams *d_clients = (ClientParams *)c->client_ring.dev(ring);
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        if (band) {  // checked under the same lock that fixes the windows this batch is demodulated with
            for (size_t i = 0; i < c->aslots.size(); i++) {
                const AudioSlot &s = c->aslots[i];
                // an empty window (a client between psdr_client_add and its first set_audio_range) reads no bin
                if (s.active && !s.paused && s.r > s.l && ((uint32_t)s.l < band[0] || (uint64_t)s.r > (uint64_t)band[0] + band[1])) {
                    c->client_ring.idx = (c->client_ring.idx + ParamRing::K - 1) % ParamRing::K;  // hand the slot back
                    return fail(PSDR_ERR_INVALID, "client %zu: window [%d, %d) outside the band [%u, %u)", i, s.l, s.r,
                                band[0], band[0] + band[1]);
                }
            }
        }
        c->demod_seq++;
        for (size_t i = 0; i < c->aslots.size(); i++) {
            AudioSlot &s = c->aslots[i];
            if (!s.active || s.paused) continue;
            s.last_seq = c->demod_seq;
            s.b_l = s.l, s.b_r = s.r, s.b_mid = s.mid;
            ClientParams &p = h_clients[nact++];
            p.l = s.l;
            p.r = s.r;
            p.m_floor = (int)std::floor(s.mid);
            p.mode = s.mode;
            p.slot = (int)i;
            p.state_cur = s.state_cur;
            s.state_cur ^= 1;
            p.agc_reset = c->post_on ? s.agc_reset : 0;
            p.paused = 0;
            if (c->post_on) s.agc_reset = 0;
        }
        // Paused clients (psdr_client_set_paused) are not demodulated: signal_loop never calls send_audio for a client
        // whose socket is backed up (src/websocket.cpp:170-176), so its overlap-add tails, FM sample, DC blocker and
        // AGC stand still (src/signal.cpp:273-284).  The post chain lists them BEHIND the active ones with an empty
        // stream: its double-buffered histories alternate per batch for every listed client, state unchanged.  A
        // pending AGC reset stays pending until the client's next batch (it only takes effect there anyway).
        if (c->post_on && nact > 0)
            for (size_t i = 0; i < c->aslots.size(); i++) {
                const AudioSlot &s = c->aslots[i];
                if (!s.active || !s.paused || s.agc_reset == 2) continue;  // (a client that never ran has no history)
                ClientParams &p = h_clients[nact + npaused++];
                p = ClientParams{};
                p.slot = (int)i;
                p.state_cur = s.state_cur;
                p.paused = 1;
            }
    }
    c->last_demod_frames = nframes;
    if (nact == 0) return PSDR_OK;
    HIPCHK(hipMemcpyAsync(d_clients, h_clients, (size_t)(nact + npaused) * sizeof(ClientParams),
                          hipMemcpyHostToDevice, c->side));
    DemodArgs a{};
    a.spec = spec;
    a.spec_stride = spec_stride;
    a.is_real = c->is_real ? 1 : 0;
    a.lay = c->lay;
    if (band) {
        a.lay = SpecLayout{};
        a.lay.k0 = (int)band[0];
        if (band_tiled) {
            a.lay.mode = 4;
            a.lay.m1 = c->M1;
            a.lay.l2m1 = c->log2M1;
            a.lay.L = c->M2;
            a.lay.l2L = c->log2M2;
            a.lay.Lw = (int)(band[1] >> c->log2M1);
            a.lay.c2_0 = (int)(band[0] >> c->log2M1);
        }
    }
    a.n = c->n;
    a.nframes = nframes;
    a.max_batch = c->max_batch;
    a.first_frame_num = first_frame_num;
    a.clients = d_clients;
    a.Wn = c->d_Wn;
    a.nstages = c->nstages;
    for (int i = 0; i < c->nstages; i++) a.radix[i] = c->radix[i];
    a.stage_tab = c->d_stage_tab;
    a.ypost = c->d_ypost;
    a.pwr = c->d_pwr;
    a.gscratch = c->d_gscratch;
    a.lds_mode = c->lds_mode;
    a.audio = c->d_audio;
    a.nan_flags = c->d_nan;
    a.real_prev = c->d_real_prev;
    a.bb_tail = c->d_bb_tail;
    a.bb_last = c->d_bb_last;
    a.slots = (int)c->aslots.size();
    if (c->gather_pending && c->side3) {  // post chain: the previous batch's audio rows are still being gathered
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_gather, 0));
        c->gather_pending = false;
    }
    bool ola_done = false;
    {
        ProfScope ps(c, K_IDFT, c->side);
        const bool fixed_plan = (c->n == 360 || c->n == 720) && !c->idft_block && !c->idft_generic;
        if (fixed_plan && c->demod_chain) {
            // transform + overlap-add + demodulation in one kernel, one wave per chain of K consecutive frames of a
            // client (demod.h): long chains repeat fewer transforms (1 or 2 per chain), short ones give few clients
            // enough waves
            // (256 clients x 256 frames, same box: K = 4 / 8 / 16 / 32 -> 5.81 / 5.77 / 5.93 / 6.04 us per frame, the
            // two-kernel path 5.99)
            int K = c->demod_chain_k > 0 ? c->demod_chain_k : 8;
            if (c->demod_chain_k <= 0)
                while (K > 4 && (unsigned)nact * (unsigned)((nframes + K - 1) / K) < 1024u) K >>= 1;
            const unsigned items = (unsigned)nact * (unsigned)((nframes + K - 1) / K);
            const unsigned W = c->n == 360 ? 4u : 1u;
            const size_t lds = (size_t)(1 + W) * c->n * sizeof(cf);
            if (c->n == 360)
                hipLaunchKernelGGL((k_demod_chain_fixed<360, 8, 9, 5>), dim3((items + W - 1) / W), dim3(64 * W), lds,
                                   c->side, a, nact, K);
            else
                hipLaunchKernelGGL((k_demod_chain_fixed<720, 8, 9, 10>), dim3((items + W - 1) / W), dim3(64 * W), lds,
                                   c->side, a, nact, K);
            ola_done = true;
        } else if (fixed_plan) {
            // compile-time plans (demod.h): 360 = 8*9*5, 720 = 8*9*10; W items per work-group in
            // the 15 KiB of LDS an FFT pass leaves free on a CU
            const unsigned items = (unsigned)nact * (unsigned)nframes;
            const unsigned W = c->n == 360 ? 4u : 1u;
            const size_t lds = (size_t)(1 + W) * c->n * sizeof(cf);
            if (c->n == 360)
                hipLaunchKernelGGL((k_demod_idft_fixed<360, 8, 9, 5>), dim3((items + W - 1) / W), dim3(64 * W), lds,
                                   c->side, a, nact);
            else
                hipLaunchKernelGGL((k_demod_idft_fixed<720, 8, 9, 10>), dim3((items + W - 1) / W), dim3(64 * W), lds,
                                   c->side, a, nact);
        } else if (c->n <= 512 && !c->idft_block) {
            // one wave per (client, frame), no work-group barriers (demod.h)
            const unsigned items = (unsigned)nact * (unsigned)nframes;
            const size_t lds = (size_t)(2 * PSDR_IDFT_WAVES + 1) * c->n * sizeof(cf);
            hipLaunchKernelGGL(k_demod_idft_wave, dim3((items + PSDR_IDFT_WAVES - 1) / PSDR_IDFT_WAVES),
                               dim3(64 * PSDR_IDFT_WAVES), lds, c->side, a, nact);
        } else {
            hipLaunchKernelGGL(k_demod_idft, dim3(nact, nframes), dim3(c->idft_threads), c->idft_lds, c->side,
                               a);
        }
        HIPCHK(hipGetLastError());
    }
    if (!ola_done) {
        ProfScope ps(c, K_OLA, c->side);
        const unsigned items = (unsigned)nact * (unsigned)((nframes + PSDR_OLA_FG - 1) / PSDR_OLA_FG);
        hipLaunchKernelGGL(k_demod_ola, dim3((items + 3) / 4), dim3(256), 0, c->side, a, nact);
        HIPCHK(hipGetLastError());
    }
    hipStream_t last_user = c->side;
    if (c->post_on && nact > 0) {
        const int par = (int)(c->chain_seq & 1);
        const bool piped = c->side != c->stream && c->side2 != nullptr;
        hipStream_t s2 = piped ? c->side2 : c->side, s1 = piped ? c->side3 : c->side;
        PostArgs pa = c->post;
        pa.clients = d_clients;
        pa.nact = nact + npaused;
        pa.nframes = nframes;
        pa.V1 = c->post_v1[par];
        pa.V1n = c->post_v1[par ^ 1];
        pa.fstart = c->post_fstart[par];
        pa.len = c->post_len[par];
        pa.P = c->post_p[par];
        pa.S = c->post_s[par];
        const int nall = nact + npaused;
        const unsigned cb = (unsigned)((nall + 63) / 64);
        const size_t Tb = (size_t)nframes * pa.h;  // longest possible stream of this batch
        const unsigned nblk = (unsigned)((pa.L - 1 + Tb + pa.L - 1) / pa.L);
        {  // ---- stage 1 (its own stream when the consumers have theirs)
            if (piped) {
                HIPCHK(hipEventRecord(c->ev_demod, c->side));
                HIPCHK(hipStreamWaitEvent(s1, c->ev_demod, 0));
                if (c->chain_seq >= 2) HIPCHK(hipStreamWaitEvent(s1, c->ev_s2[par], 0));  // stage 2 of batch b-2 read this set
            }
            ProfScope ps(c, K_POST, s1);
            hipLaunchKernelGGL(k_pc_index, dim3(nall), dim3(64), 0, s1, pa);
            hipLaunchKernelGGL(k_pc_gather, dim3(nall, nframes), dim3(256), 0, s1, pa);
            if (piped) {  // the audio rows and NaN flags are read: the next batch's demodulation may overwrite them
                HIPCHK(hipEventRecord(c->ev_gather, s1));
                c->gather_pending = true;
            }
            pa.ma_fused = pa.D == 32 ? 1 : 0;  // both averages in one loop (postchain.h)
            if (pa.ma_fused) {
                hipLaunchKernelGGL(k_pc_ma2, dim3(cb), dim3(64), 0, s1, pa);
            } else if ((pa.D & (pa.D - 1)) == 0) {
                hipLaunchKernelGGL((k_pc_ma<false, true>), dim3(cb), dim3(64), 0, s1, pa);
                hipLaunchKernelGGL((k_pc_ma<true, true>), dim3(cb), dim3(64), 0, s1, pa);
            } else {
                hipLaunchKernelGGL((k_pc_ma<false, false>), dim3(cb), dim3(64), 0, s1, pa);
                hipLaunchKernelGGL((k_pc_ma<true, false>), dim3(cb), dim3(64), 0, s1, pa);
            }
            pa.hist_sel = 0;
            hipLaunchKernelGGL(k_pc_history, dim3(nall), dim3(256), (size_t)pa.D * sizeof(float), s1, pa);
            // the look-ahead maxima and w_t are parallel work: they ride in this stage (P and S exist per parity), so
            // that stage 2 is nothing but the gain recurrence and the output - the two sequential kernels (k_pc_ma2
            // here, k_pc_gain there: ~1.1 ms each beside the FFT passes) sit in different stages
            hipLaunchKernelGGL(k_pc_scan, dim3(nall, nblk, 2), dim3(64), 0, s1, pa);
            hipLaunchKernelGGL(k_pc_want, dim3(nall, (unsigned)((Tb + 255) / 256)), dim3(256), 0, s1, pa);
            // w_t is all the gain recurrence needs: it must not wait for the history copy below, which in turn waits
            // for the previous batch's output kernel (gain -> out -> history -> gain would be one serial chain per batch)
            if (piped) HIPCHK(hipEventRecord(c->ev_want[par], s1));
            // V1's tail -> the other parity's history rows (a plain copy, no LDS).  The other parity's stage 2 (the
            // previous batch: k_pc_out reads those rows) must be done with them
            if (piped && c->chain_seq >= 1) HIPCHK(hipStreamWaitEvent(s1, c->ev_s2[par ^ 1], 0));
            pa.hist_sel = 1;
            hipLaunchKernelGGL(k_pc_history, dim3(nall), dim3(256), 0, s1, pa);
            HIPCHK(hipGetLastError());
        }
        if (piped) {
            HIPCHK(hipEventRecord(c->ev_s1[par], s1));
            HIPCHK(hipStreamWaitEvent(s2, c->ev_want[par], 0));
        }
        {  // ---- stage 2
            ProfScope ps(c, K_POST, s2);
            if (pa.attack >= pa.release)
                hipLaunchKernelGGL(k_pc_gain<true>, dim3(cb), dim3(64), 0, s2, pa);
            else
                hipLaunchKernelGGL(k_pc_gain<false>, dim3(cb), dim3(64), 0, s2, pa);
            if (piped) HIPCHK(hipStreamWaitEvent(s2, c->ev_s1[par], 0));  // k_pc_out reads V1's history rows
            hipLaunchKernelGGL(k_pc_out, dim3(nall, nframes), dim3(256), 0, s2, pa);
            HIPCHK(hipGetLastError());
        }
        if (piped) {
            HIPCHK(hipEventRecord(c->ev_s2[par], s2));
            c->side2_pending = true;
        }
        c->chain_seq++;
        last_user = s2;
    }
    HIPCHK(c->client_ring.release(ring, last_user));
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_side_done, c->side));
        c->side_pending = true;
        HIPCHK(hipEventRecord(c->ev_set_done[c->cur_set], c->side));
        c->set_pending[c->cur_set] = true;
    }
    return PSDR_OK;
}

// A slot whose client attached AFTER the last demodulation batch holds the previous occupant's results (or
// nothing): the reference's per-client task would not exist for that frame either (src/websocket.cpp:156-185 walks
// signal_slices at the time of the frame).  PSDR_ERR_NO_DATA, nothing is copied.
static int slot_in_last_batch(psdr_ctx *c, int id) {
    std::lock_guard<std::mutex> lk(c->mtx);
    if (c->demod_seq == 0 || c->aslots[id].last_seq != c->demod_seq)
        return fail(PSDR_ERR_NO_DATA, "client %d was not part of the last demodulation batch", id);
    return PSDR_OK;
}

// ---- batched read-back: ONE synchronisation and at most four copies per batch for ALL clients ----------------
// (src/websocket.cpp:156-185 makes one pass over signal_slices per frame; per-client psdr_read_audio would pay a
// synchronisation and three copies per client and frame)
extern "C" int psdr_fetch_batch(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    const size_t F = (size_t)c->last_demod_frames, h = (size_t)c->n / 2, mb = (size_t)c->max_batch, S = c->aslots.size();
    if (F == 0 || c->demod_seq == 0) return fail(PSDR_ERR_STATE, "no demodulated batch to fetch");
    HIPCHK(hipSetDevice(c->device));
    // (each block on its own: a failed allocation leaves nothing half-initialised behind for the next call)
    if (!c->h_audio) HIPCHK(hipHostMalloc((void **)&c->h_audio, S * mb * h * sizeof(float), hipHostMallocDefault));
    if (!c->h_pwr) HIPCHK(hipHostMalloc((void **)&c->h_pwr, S * mb * sizeof(float), hipHostMallocDefault));
    if (!c->h_nan) HIPCHK(hipHostMalloc((void **)&c->h_nan, S * mb * sizeof(int32_t), hipHostMallocDefault));
    if (c->post_on && !c->h_pcm) HIPCHK(hipHostMalloc((void **)&c->h_pcm, S * mb * h * sizeof(int32_t), hipHostMallocDefault));
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    // rows [slot][0..F) of the device arrays [slot][max_batch][...]: one strided copy each
    HIPCHK(hipMemcpy2DAsync(c->h_audio, mb * h * sizeof(float), c->d_audio, mb * h * sizeof(float), F * h * sizeof(float), S,
                            hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpy2DAsync(c->h_pwr, mb * sizeof(float), c->d_pwr, mb * sizeof(float), F * sizeof(float), S,
                            hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpy2DAsync(c->h_nan, mb * sizeof(int32_t), c->d_nan, mb * sizeof(int), F * sizeof(int32_t), S,
                            hipMemcpyDeviceToHost, c->stream));
    if (c->post_on)
        HIPCHK(hipMemcpy2DAsync(c->h_pcm, mb * h * sizeof(int32_t), c->post.pcm, mb * h * sizeof(int32_t), F * h * sizeof(int32_t), S,
                                hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        for (auto &s : c->aslots)
            if (s.last_seq == c->demod_seq) s.f_l = s.b_l, s.f_r = s.b_r, s.f_mid = s.b_mid;
        c->fetched_frames = (int)F;
        c->fetched_seq = c->demod_seq;
        c->fetched_pcm = c->post_on;
    }
    return PSDR_OK;
}
extern "C" int psdr_fetched_window(psdr_ctx *c, int id, int *l, double *audio_mid, int *r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    if (c->fetched_seq == 0) return fail(PSDR_ERR_STATE, "psdr_fetch_batch() first");
    const AudioSlot &s = c->aslots[id];
    if (s.last_seq != c->fetched_seq) return fail(PSDR_ERR_NO_DATA, "client %d was not part of the fetched batch", id);
    if (l) *l = s.f_l;
    if (audio_mid) *audio_mid = s.f_mid;
    if (r) *r = s.f_r;
    return PSDR_OK;
}
extern "C" int psdr_fetched_audio(psdr_ctx *c, int id, int frame, const float **audio, float *pwr, int32_t *nan_flag,
                                  const int32_t **pcm) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
        if (c->fetched_seq == 0) return fail(PSDR_ERR_STATE, "psdr_fetch_batch() first");
        if (c->aslots[id].last_seq != c->fetched_seq)
            return fail(PSDR_ERR_NO_DATA, "client %d was not part of the fetched batch", id);
    }
    if (frame < 0 || frame >= c->fetched_frames) return fail(PSDR_ERR_INVALID, "frame %d not in the fetched batch of %d", frame, c->fetched_frames);
    const size_t h = (size_t)c->n / 2, mb = (size_t)c->max_batch, row = (size_t)id * mb + (size_t)frame;
    if (audio) *audio = c->h_audio + row * h;
    if (pwr) *pwr = c->h_pwr[row];
    if (nan_flag) *nan_flag = c->h_nan[row];
    if (pcm) *pcm = c->fetched_pcm ? c->h_pcm + row * h : nullptr;
    return PSDR_OK;
}

// ---- post-demodulation chain ---------------------------------------------------
extern "C" int psdr_set_post_chain(psdr_ctx *c, int enable) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    if (!enable) {
        c->post_on = false;
        return PSDR_OK;
    }
    if (c->post_allocs.empty()) {
        const int rate = c->cfg.audio_rate;
        if (rate < 750) return fail(PSDR_ERR_INVALID, "audio_rate %d too small for the DC blocker", rate);
        PostArgs &a = c->post;
        const size_t S = c->aslots.size(), h = (size_t)c->n / 2, Tm = (size_t)c->max_batch * h;
        a.max_batch = c->max_batch;
        a.h = (int)h;
        a.slots = (int)S;
        a.D = rate / 750 * 2;  // DCBlocker(audio_max_sps / 750 * 2), src/signal.cpp:54
        // AGC(0.2f, 50.0f, 300.0f, 200.0f, audio_max_sps), src/signal.cpp:55 and
        // src/utils/audioprocessing.cpp:5-16 (exp() on a float argument is C's double exp)
        const float sr = (float)rate;
        a.L = (int)(size_t)(200.0f * sr / 1000.0f);
        a.desired = 0.2f;
        a.attack = (float)(1 - std::exp((double)(-1.0f / (50.0f * 0.001f * sr))));
        a.release = (float)(1 - std::exp((double)(-1.0f / (300.0f * 0.001f * sr))));
        // the DC blocker's history (D floats) is moved in place through LDS (k_pc_history): 12288 floats = 48 KiB, i.e.
        // audio rates up to 4.6 MHz; the AGC look-ahead L has no such limit (k_pc_scan walks it in chunks)
        if (a.D < 1 || a.L < 2 || a.D > 12288)
            return fail(PSDR_ERR_UNSUPPORTED, "audio_rate %d: DC delay %d / look-ahead %d unsupported", rate, a.D, a.L);
        auto alloc = [&](void **ptr, size_t bytes) -> int {
            HIPCHK(hipMalloc(ptr, std::max<size_t>(bytes, 16)));
            HIPCHK(hipMemset(*ptr, 0, std::max<size_t>(bytes, 16)));
            c->post_allocs.push_back(*ptr);
            return PSDR_OK;
        };
        // client-major streams (postchain.h): pitches are multiples of 4 floats, + padding for the
        // blocked kernels' look-ahead
        a.px = ((size_t)a.D + Tm + PSDR_PC_PAD + 3) & ~(size_t)3;
        a.vo = (4 - ((a.L - 1) & 3)) & 3;
        a.pv = ((size_t)a.vo + (size_t)a.L - 1 + Tm + PSDR_PC_PAD + 3) & ~(size_t)3;
        int rc = 0;
        for (int i = 0; i < 2; i++) {
            rc |= alloc((void **)&c->post_fstart[i], S * c->max_batch * sizeof(int));
            rc |= alloc((void **)&c->post_len[i], S * sizeof(int));
            rc |= alloc((void **)&c->post_v1[i], a.pv * S * sizeof(float));
            rc |= alloc((void **)&c->post_p[i], a.pv * S * sizeof(float));
            rc |= alloc((void **)&c->post_s[i], a.pv * S * sizeof(float));
            if (!c->ev_s1[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_s1[i], hipEventDisableTiming));
            if (!c->ev_s2[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_s2[i], hipEventDisableTiming));
        }
        if (!c->side2) {
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&c->side2, hipStreamNonBlocking, hi));
            HIPCHK(hipStreamCreateWithPriority(&c->side3, hipStreamNonBlocking, hi));
            HIPCHK(hipEventCreateWithFlags(&c->ev_demod, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
            for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_want[i], hipEventDisableTiming));
        }
        rc |= alloc((void **)&a.X, a.px * S * sizeof(float));
        rc |= alloc((void **)&a.M1, a.px * S * sizeof(float));
        rc |= alloc((void **)&a.pcm, S * Tm * sizeof(int32_t));
        rc |= alloc((void **)&a.dc_s1, S * sizeof(float));
        rc |= alloc((void **)&a.dc_s2, S * sizeof(float));
        rc |= alloc((void **)&a.agc_gain, S * sizeof(float));
        rc |= alloc((void **)&a.agc_n0, S * sizeof(int));
        if (rc) return PSDR_ERR_NOMEM;
        a.audio = c->d_audio;
        a.nan_flags = c->d_nan;
    }
    c->post_on = true;
    return PSDR_OK;
}
extern "C" int psdr_read_pcm(psdr_ctx *c, int id, int nframes, int32_t *pcm, int *nframes_out) {
    if (!c || !pcm) return fail(PSDR_ERR_INVALID, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
    }
    if (!c->post_on) return fail(PSDR_ERR_STATE, "post chain not enabled (psdr_set_post_chain)");
    HIPCHK(hipSetDevice(c->device));
    const size_t F = (size_t)c->last_demod_frames, h = (size_t)c->n / 2, mb = (size_t)c->max_batch;
    if (F == 0) return fail(PSDR_ERR_STATE, "no demodulated batch to read");
    if (nframes < (int)F) return fail(PSDR_ERR_INVALID, "buffer holds %d frames, the last batch has %zu", nframes, F);
    {
        int rc = slot_in_last_batch(c, id);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpy(pcm, c->post.pcm + (size_t)id * mb * h, F * h * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nframes_out) *nframes_out = (int)F;
    return PSDR_OK;
}

extern "C" int psdr_demod_batch(psdr_ctx *c, uint64_t first_frame_num) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->last_nframes < 1) return fail(PSDR_ERR_STATE, "demod_batch before process_batch/execute");
    return demod_impl(c, c->d_spec, c->spec_stride, c->last_nframes, first_frame_num);
}
extern "C" int psdr_demod_batch_from(psdr_ctx *c, const float *d_spec, size_t frame_stride_bins,
                                     int nframes, uint64_t first_frame_num) {
    if (!c || !d_spec) return fail(PSDR_ERR_INVALID, "null argument");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    if (frame_stride_bins < (c->is_real ? c->N / 2 + 1 : c->N))
        return fail(PSDR_ERR_INVALID, "frame stride smaller than one spectrum");
    return demod_impl(c, (const cf *)d_spec, frame_stride_bins, nframes, first_frame_num);
}

extern "C" int psdr_pack_band(psdr_ctx *c, int nframes, uint32_t first_bin, uint32_t nbins, float *d_out,
                              size_t out_stride_bins) {
    if (!c || !d_out) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->last_nframes < 1) return fail(PSDR_ERR_STATE, "pack_band before process_batch/execute");
    if (nframes < 1 || nframes > c->last_nframes)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, %d] (the last batch)", nframes, c->last_nframes);
    const uint32_t R = (uint32_t)(c->is_real ? c->N / 2 : c->N);
    if (first_bin >= R || nbins < 1 || nbins > R) return fail(PSDR_ERR_INVALID, "band [%u, +%u) outside [0, %u)", first_bin, nbins, R);
    if (out_stride_bins < nbins) return fail(PSDR_ERR_INVALID, "output stride smaller than the band");
    HIPCHK(hipSetDevice(c->device));
    ProfScope ps(c, K_BAND, c->stream);
    hipLaunchKernelGGL(k_band_pack, dim3((nbins + 255) / 256, nframes), dim3(256), 0, c->stream, c->d_spec, c->spec_stride,
                       c->lay, (int)R, (int)first_bin, (int)nbins, (cf *)d_out, out_stride_bins);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
extern "C" int psdr_demod_batch_from_band(psdr_ctx *c, const float *d_band, size_t frame_stride_bins, uint32_t first_bin,
                                          uint32_t nbins, int nframes, uint64_t first_frame_num) {
    if (!c || !d_band) return fail(PSDR_ERR_INVALID, "null argument");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    if (nbins < 1 || frame_stride_bins < nbins) return fail(PSDR_ERR_INVALID, "frame stride smaller than the band");
    const uint32_t band[2] = {first_bin, nbins};
    return demod_impl(c, (const cf *)d_band, frame_stride_bins, nframes, first_frame_num, band);
}

// ---- band sharding without the pack: pass 2 writes band regions ------------------------------------------------
extern "C" int psdr_set_band_layout(psdr_ctx *c, int nbands, uint32_t halo_bins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->is_real || c->lay.mode == 0 || c->lay.mode == 2 || (c->M1 != 1024 && c->M1 != 2048) || c->M2 != 1024 || c->p1_wave)
        return fail(PSDR_ERR_UNSUPPORTED, "banded spectrum: 2^20- and 2^21-point IQ frames only (use psdr_pack_band)");
    if (nbands < 1 || nbands > 16 || (nbands & (nbands - 1))) return fail(PSDR_ERR_INVALID, "nbands %d: a power of two <= 16", nbands);
    if (halo_bins > (uint32_t)c->M) return fail(PSDR_ERR_INVALID, "halo of %u bins", halo_bins);
    const int H = (int)((halo_bins + (uint32_t)c->M1 - 1) >> c->log2M1), Lb = c->M2 / nbands, Lw = Lb + H;
    // k_band_halo repeats the first H columns of band b+1 behind band b: they must be band b+1's OWN columns (a halo
    // longer than a band would reach into band b+2's, which band b+1's region only holds as its own halo - written
    // by the same launch)
    if (H > Lb)
        return fail(PSDR_ERR_INVALID, "halo of %u bins = %d columns of %d bins exceeds a band of %d columns (%d bands)", halo_bins, H,
                    c->M1, Lb, nbands);
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipDeviceSynchronize());
    const size_t F = (size_t)c->max_batch, fs = (size_t)c->M1 * Lw;
    {  // the new buffers first: a failed allocation leaves the context as it was
        cf *fresh[2] = {nullptr, nullptr};
        for (int s = 0; s < 2; s++) {
            if (hipMalloc((void **)&fresh[s], (size_t)nbands * F * fs * sizeof(cf)) != hipSuccess ||
                hipMemset(fresh[s], 0, (size_t)nbands * F * fs * sizeof(cf)) != hipSuccess) {
                (void)hipGetLastError();
                for (int t = 0; t <= s; t++)
                    if (fresh[t]) (void)hipFree(fresh[t]);
                return fail(PSDR_ERR_NOMEM, "banded spectrum: %zu bytes per result set", (size_t)nbands * F * fs * sizeof(cf));
            }
        }
        for (int s = 0; s < 2; s++) {
            if (c->spec_pool[s]) (void)hipFree(c->spec_pool[s]);
            c->spec_pool[s] = fresh[s];
        }
    }
    c->nbands = nbands;
    c->band_H = H;
    c->spec_stride = fs;  // frame to frame INSIDE a region
    c->lay.mode = 3;
    c->lay.l2Lb = ilog2((size_t)Lb);
    c->lay.Lw = Lw;
    c->lay.band_stride = F * fs;
    c->lay.c2_0 = 0;
    c->set_pending[0] = c->set_pending[1] = false;
    select_set(c, c->cur_set);
    c->last_nframes = 0;
    c->out_valid = false;
    return PSDR_OK;
}
extern "C" int psdr_band_region(psdr_ctx *c, int band, const float **d_region, size_t *frame_stride_bins, uint32_t *first_bin,
                                uint32_t *nbins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (!c->nbands) return fail(PSDR_ERR_STATE, "psdr_set_band_layout() first");
    if (band < 0 || band >= c->nbands) return fail(PSDR_ERR_INVALID, "band %d outside [0, %d)", band, c->nbands);
    if (d_region) *d_region = (const float *)(c->d_spec + (size_t)band * c->lay.band_stride);
    if (frame_stride_bins) *frame_stride_bins = c->spec_stride;
    if (first_bin) *first_bin = (uint32_t)((size_t)band << (c->lay.l2Lb + c->log2M1));
    if (nbins) *nbins = (uint32_t)c->spec_stride;
    return PSDR_OK;
}
extern "C" int psdr_demod_batch_from_band_region(psdr_ctx *c, const float *d_region, size_t frame_stride_bins, uint32_t first_bin,
                                                 uint32_t nbins, int nframes, uint64_t first_frame_num) {
    if (!c || !d_region) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->is_real || c->M2 != 1024) return fail(PSDR_ERR_UNSUPPORTED, "band regions: IQ frames with 1024-point rows only");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    const uint32_t m1 = (uint32_t)c->M1;
    if (nbins < m1 || (nbins & (m1 - 1)) || (first_bin & (m1 - 1)) || first_bin >= (uint32_t)c->M)
        return fail(PSDR_ERR_INVALID, "band region [%u, +%u): whole columns of %u bins", first_bin, nbins, m1);
    if (frame_stride_bins < nbins) return fail(PSDR_ERR_INVALID, "frame stride smaller than the band");
    const uint32_t band[2] = {first_bin, nbins};
    return demod_impl(c, (const cf *)d_region, frame_stride_bins, nframes, first_frame_num, band, true);
}

extern "C" int psdr_read_audio(psdr_ctx *c, int id, int nframes, float *audio, float *pwr, int32_t *nan_flags,
                               int *nframes_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
    }
    HIPCHK(hipSetDevice(c->device));
    const size_t F = (size_t)c->last_demod_frames, h = (size_t)c->n / 2, mb = (size_t)c->max_batch;
    if (F == 0) return fail(PSDR_ERR_STATE, "no demodulated batch to read");
    if (nframes < (int)F) return fail(PSDR_ERR_INVALID, "buffers hold %d frames, the last batch has %zu", nframes, F);
    if (nframes_out) *nframes_out = (int)F;
    {
        int rc = slot_in_last_batch(c, id);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    if (audio)
        HIPCHK(hipMemcpyAsync(audio, c->d_audio + (size_t)id * mb * h, F * h * sizeof(float),
                              hipMemcpyDeviceToHost, c->stream));
    if (pwr)
        HIPCHK(hipMemcpyAsync(pwr, c->d_pwr + (size_t)id * mb, F * sizeof(float), hipMemcpyDeviceToHost,
                              c->stream));
    if (nan_flags)
        HIPCHK(hipMemcpyAsync(nan_flags, c->d_nan + (size_t)id * mb, F * sizeof(int),
                              hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
extern "C" int psdr_audio_device_ptr(psdr_ctx *c, int id, const float **d_audio, const float **d_pwr) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (id < 0 || id >= (int)c->aslots.size()) return fail(PSDR_ERR_INVALID, "bad id %d", id);
    const size_t h = (size_t)c->n / 2, mb = (size_t)c->max_batch;
    if (d_audio) *d_audio = c->d_audio + (size_t)id * mb * h;
    if (d_pwr) *d_pwr = c->d_pwr + (size_t)id * mb;
    return PSDR_OK;
}

// ---- waterfall ---------------------------------------------------------------------------
static int check_wslot(psdr_ctx *c, int id) {
    if (id < 0 || id >= (int)c->wslots.size() || !c->wslots[id].active)
        return fail(PSDR_ERR_INVALID, "no waterfall client with id %d", id);
    return PSDR_OK;
}
extern "C" int psdr_waterfall_add(psdr_ctx *c, int *id_out) {
    if (!c || !id_out) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    for (size_t i = 0; i < c->wslots.size(); i++)
        if (!c->wslots[i].active) {
            WfSlot &s = c->wslots[i];
            s = WfSlot();
            s.active = true;
            // default = whole spectrum at the coarsest level (src/websocket.cpp:198)
            s.level = c->levels - 1;
            s.l = 0;
            s.r = std::min(c->min_waterfall_fft, (int)(c->R >> s.level));  // set_waterfall_range(levels-1, 0, min_waterfall_fft)
            *id_out = (int)i;
            return PSDR_OK;
        }
    return fail(PSDR_ERR_NOMEM, "all %zu waterfall client slots are in use", c->wslots.size());
}
extern "C" int psdr_waterfall_remove(psdr_ctx *c, int id) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    c->wslots[id].active = false;
    return PSDR_OK;
}
extern "C" int psdr_waterfall_set_range(psdr_ctx *c, int id, int level, int l, int r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    if (level < 0 || level >= c->levels) return fail(PSDR_ERR_INVALID, "level %d out of range", level);
    const int len = (int)(c->R >> level);
    if (l < 0) l = 0;
    if (r > len) r = len;  // the reference forgets this bound (src/waterfall.cpp:55-58)
    if (l > r) return fail(PSDR_ERR_INVALID, "empty waterfall range");
    WfSlot &s = c->wslots[id];
    s.level = level;
    s.l = l;
    s.r = r;
    return PSDR_OK;
}
extern "C" int psdr_waterfall_on_window_message(psdr_ctx *c, int id, int new_l, int new_r,
                                                int *level_out, int *l_out, int *r_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    // src/waterfall.cpp:53-79
    if (new_l < 0 || new_r < 0 || new_l >= new_r) return fail(PSDR_ERR_INVALID, "window rejected");
    const int mwf = c->min_waterfall_fft;
    float new_l_f = (float)new_l, new_r_f = (float)new_r;
    int new_level = c->levels - 1;
    float best = (float)(mwf * 2);
    for (int i = 0; i < c->levels; i++) {
        const float send_size = std::fabs((new_r_f - new_l_f) - (float)mwf);
        if (send_size < best) {
            best = send_size;
            new_level = i;
            new_l = (int)std::round(new_l_f);
            new_r = (int)std::round(new_r_f);
        }
        new_l_f /= 2;
        new_r_f /= 2;
    }
    int rc = psdr_waterfall_set_range(c, id, new_level, new_l, new_r);
    if (rc) return rc;
    if (level_out) *level_out = new_level;
    if (l_out) *l_out = c->wslots[id].l;
    if (r_out) *r_out = c->wslots[id].r;
    return PSDR_OK;
}

extern "C" int psdr_waterfall_batch(psdr_ctx *c, uint64_t first_frame_num) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->last_nframes < 1) return fail(PSDR_ERR_STATE, "waterfall_batch before process_batch/execute");
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    const int ring = c->wf_ring.acquire();
    if (ring < 0) return fail(PSDR_ERR_HIP, "waterfall parameter ring: event wait failed");
    WfClient *h_wf = (WfClient *)c->wf_ring.host(ring);
    WfClient *d_wf = (WfClient *)c->wf_ring.dev(ring);
    int *h_sent = (int *)((unsigned char *)c->wf_ring.host(ring) + c->wf_sent_off);
    int *d_sent = (int *)((unsigned char *)c->wf_ring.dev(ring) + c->wf_sent_off);
    int nsent = 0;
    for (int f = 0; f < c->last_nframes; f++)
        if ((first_frame_num + (uint64_t)f) % (uint64_t)c->cfg.skip_num == 0) h_sent[nsent++] = f;
    size_t total = 0;
    int maxid = -1;
    for (size_t i = 0; i < c->wslots.size(); i++) {
        WfSlot &s = c->wslots[i];
        WfClient &w = h_wf[i];
        w.active = s.active ? 1 : 0;
        w.level = s.level;
        w.l = s.l;
        w.r = s.r;
        w.qoff = 0;
        for (int t = 0; t < s.level; t++) w.qoff += c->R >> t;
        w.out_off = total;
        s.out_off = total;
        s.nsent = s.active ? nsent : 0;
        s.b_level = s.level;
        s.b_l = s.l;
        s.b_r = s.r;
        if (s.active) {
            total += (size_t)nsent * (size_t)(s.r - s.l);
            total = (total + 15) & ~(size_t)15;
            maxid = (int)i;
        }
    }
    if (maxid < 0 || nsent == 0 || total == 0) return PSDR_OK;
    if (total > c->wfout_cap) {
        if (c->d_wfout) HIPCHK(hipFree(c->d_wfout));
        c->d_wfout = nullptr;
        HIPCHK(hipMalloc((void **)&c->d_wfout, total));
        c->wfout_cap = total;
    }
    HIPCHK(hipMemcpyAsync(d_wf, h_wf, (size_t)(maxid + 1) * sizeof(WfClient), hipMemcpyHostToDevice,
                          c->side));
    HIPCHK(hipMemcpyAsync(d_sent, h_sent, (size_t)nsent * sizeof(int), hipMemcpyHostToDevice,
                          c->side));
    {
        ProfScope ps(c, K_WFALL, c->side);
        hipLaunchKernelGGL(k_waterfall_gather, dim3(maxid + 1, nsent), dim3(256), 0, c->side, c->d_q,
                           c->q_stride, c->d_qt, c->qt_stride, c->tiled_lt, c->tile_ch, c->recmap, d_wf, d_sent, nsent,
                           c->d_wfout);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(c->wf_ring.release(ring, c->side));
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_side_done, c->side));
        c->side_pending = true;
        HIPCHK(hipEventRecord(c->ev_set_done[c->cur_set], c->side));
        c->set_pending[c->cur_set] = true;
    }
    return PSDR_OK;
}
extern "C" int psdr_read_waterfall(psdr_ctx *c, int id, int8_t *out, size_t out_cap, int *nsent_out, int *level_out,
                                   int *l_out, int *r_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    const WfSlot &s = c->wslots[id];
    // the window of the BATCH, not the live one (it may have changed since)
    const size_t bytes = (size_t)s.nsent * (size_t)(s.b_r - s.b_l);
    if (nsent_out) *nsent_out = s.nsent;
    if (level_out) *level_out = s.b_level;
    if (l_out) *l_out = s.b_l;
    if (r_out) *r_out = s.b_r;
    if (bytes == 0 || !out) return PSDR_OK;
    if (bytes > out_cap) return fail(PSDR_ERR_INVALID, "output buffer too small (%zu > %zu)", bytes, out_cap);
    {
        int rc2 = drain(c);
        if (rc2) return rc2;
    }
    HIPCHK(hipMemcpyAsync(out, c->d_wfout + s.out_off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}

// ---- raw results ---------------------------------------------------------------------------
extern "C" int psdr_spectrum_device_ptr(psdr_ctx *c, int frame, const float **d_spec, size_t *nbins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->max_batch) return fail(PSDR_ERR_INVALID, "frame %d out of range", frame);
    if (c->nbands) return fail(PSDR_ERR_UNSUPPORTED, "banded spectrum: a frame is not one piece (psdr_band_region)");
    if (d_spec) *d_spec = (const float *)(c->d_spec + (size_t)frame * c->spec_stride);
    if (nbins) *nbins = c->is_real ? c->N / 2 + 1 : c->N;
    return PSDR_OK;
}
extern "C" int psdr_quantized_device_ptr(psdr_ctx *c, int frame, const int8_t **d_q, size_t *nbytes) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->max_batch) return fail(PSDR_ERR_INVALID, "frame %d out of range", frame);
    {
        int rc = drain(c);
        if (rc) return rc;
        rc = ensure_level_major(c, frame);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    if (d_q) *d_q = c->d_q + (size_t)frame * c->q_stride;
    if (nbytes) *nbytes = c->q_len;
    return PSDR_OK;
}
extern "C" int psdr_read_spectrum(psdr_ctx *c, int frame, float *out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->last_nframes) return fail(PSDR_ERR_INVALID, "frame %d not in the last batch", frame);
    HIPCHK(hipSetDevice(c->device));
    return copy_spectrum_k_order(c, frame, (cf *)out);
}
extern "C" int psdr_read_quantized(psdr_ctx *c, int frame, int8_t *out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->last_nframes) return fail(PSDR_ERR_INVALID, "frame %d not in the last batch", frame);
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
        rc = ensure_level_major(c, frame);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(out, c->d_q + (size_t)frame * c->q_stride, c->q_len, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}

// ---- instrumentation -------------------------------------------------------------------------
extern "C" int psdr_set_profiling(psdr_ctx *c, int mode) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(PSDR_ERR_INVALID, "profiling mode %d (0 off, 1 hipEvents, 2 device clocks)", mode);
    HIPCHK(hipSetDevice(c->device));
    resolve_pending(c);
    resolve_kclock(c);
    c->profiling = mode == 1;
    if (mode == 2 && !c->d_kclk) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_clock_khz = khz;
        HIPCHK(hipMalloc((void **)&c->d_kclk, (size_t)2 * psdr_ctx::KCLK_SLOTS * 2 * sizeof(unsigned long long)));
        int rc = drain(c);
        if (rc) return rc;
        rc = reset_kclock(c);
        if (rc) return rc;
    }
    c->kclock = mode == 2;
    return PSDR_OK;
}
extern "C" int psdr_get_kernel_samples(psdr_ctx *c, const char *name, double *us_out, int cap, int *n_out) {
    if (!c || !name || !n_out) return fail(PSDR_ERR_INVALID, "null argument");
    resolve_pending(c);
    resolve_kclock(c);
    for (int k = 0; k < K_COUNT; k++) {
        if (strcmp(name, kKernelNames[k]) != 0) continue;
        const int n = (int)c->k_samples[k].size();
        for (int i = 0; i < n && i < cap && us_out; i++) us_out[i] = c->k_samples[k][i];
        *n_out = n;
        return PSDR_OK;
    }
    return fail(PSDR_ERR_INVALID, "no kernel named '%s'", name);
}
extern "C" int psdr_get_kernel_stats(psdr_ctx *c, int max_entries, const char **names, double *total_ms,
                                     int64_t *launches, int *n_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    resolve_pending(c);
    resolve_kclock(c);
    int n = 0;
    for (int k = 0; k < K_COUNT && n < max_entries; k++) {
        if (c->k_n[k] == 0) continue;
        if (names) names[n] = kKernelNames[k];
        if (total_ms) total_ms[n] = c->k_ms[k];
        if (launches) launches[n] = c->k_n[k];
        n++;
    }
    if (n_out) *n_out = n;
    return PSDR_OK;
}
extern "C" int psdr_reset_kernel_stats(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    resolve_pending(c);
    for (int k = 0; k < K_COUNT; k++) {
        c->k_ms[k] = 0;
        c->k_n[k] = 0;
        c->k_samples[k].clear();
    }
    if (c->d_kclk) {  // re-arm the stamp ring
        HIPCHK(hipSetDevice(c->device));
        int rc = drain(c);
        if (rc) return rc;
        return reset_kclock(c);
    }
    return PSDR_OK;
}
extern "C" int psdr_timer_start(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipEventRecord(c->t0, c->stream));
    return PSDR_OK;
}
extern "C" int psdr_timer_stop_ms(psdr_ctx *c, double *ms_out) {
    if (!c || !ms_out) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->side_pending && c->side != c->stream) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_side_done, 0));
    if (c->side2_pending && c->chain_seq > 0) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_s2[(c->chain_seq - 1) & 1], 0));
    HIPCHK(hipEventRecord(c->t1, c->stream));
    HIPCHK(hipEventSynchronize(c->t1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_out = ms;
    return PSDR_OK;
}
extern "C" void *psdr_stream(psdr_ctx *c) { return c ? (void *)c->stream : nullptr; }
// tuning builds (-DPSDR_TRACE_ON): 4864 values; pass 1 at [0], pass 2 at [128 + 2304]: 128 phase
// stamps of work-group 0 ([iteration 0..7][16]) ... [256..]: wall clock [work-group][8]
extern "C" int psdr_debug_trace(psdr_ctx *c, unsigned long long *out256) {
    if (!c || !out256) return fail(PSDR_ERR_INVALID, "null argument");
    if (!c->d_trace) return fail(PSDR_ERR_UNSUPPORTED, "library built without PSDR_TRACE_ON");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out256, c->d_trace, 4864 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PSDR_OK;
}
extern "C" int psdr_set_stream(psdr_ctx *c, void *hip_stream) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    resolve_pending(c);
    c->side_pending = false;
    c->set_pending[0] = c->set_pending[1] = false;
    select_set(c, 0);
    if (hip_stream) {  // everything in order on the caller's stream
        c->stream = (hipStream_t)hip_stream;
        c->side = c->stream;
        c->p1 = c->stream;
    } else {
        c->stream = c->own_stream;
        c->side = c->own_side;
        c->p1 = c->no_p1_stream ? c->own_stream : c->own_p1;
    }
    c->y_pending[0] = c->y_pending[1] = false;
    return PSDR_OK;
}
