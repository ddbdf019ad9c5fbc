// ctx.h - internal: the context behind the C-ABI (include/psdr.h) and what the translation units of libpsdr_hip.so
// share.  context.hip (lifetime, Level 1, ingest ring, instrumentation), pass1.hip / pass2.hip (the FFT pass
// launchers), forward.hip (the frame loop: passes + pyramid tails, spectrum / pyramid read-back, band layout,
// waterfall), demod.hip (audio clients + demodulation), postchain.hip (DC blocker / AGC / int16), wire.hip (packets).
#pragma once
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
#include "butterfly.h"
#include "quantize.h"
#include "types.h"

// sets psdr_last_error() of the calling thread and returns `code`
int psdr_fail(int code, const char *fmt, ...);
#define fail psdr_fail
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(PSDR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

// Run-time knobs of the PRODUCT library (documented in DESIGN.md): PSDR_SEG_LEN (tiles per chain segment of the fused
// real-input pass), PSDR_DEMOD_CHAIN / PSDR_DEMOD_K (one-kernel demodulation and its frames per chain), PSDR_REAL_3PASS
// (the three-pass real path).  Every other A/B switch of the tuning rounds is read only by a -DPSDR_TUNING_BUILD
// library (tools/build_variants.py tuning=PSDR_TUNING_BUILD), never by the one that ships.
inline const char *psdr_tuning_env(const char *name) {
#ifdef PSDR_TUNING_BUILD
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace psdr {

enum KernelId { K_PASS1, K_PASS2, K_UNTANGLE, K_TAIL, K_IDFT, K_OLA, K_WFALL, K_POST, K_SEAM, K_BAND, K_COUNT };
extern const char *kKernelNames[K_COUNT];

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
    int b_l = 0, b_r = 0;   // the window that batch was demodulated with (psdr_fetch_begin copies it into its FetchSet)
    double b_mid = 0;
    uint64_t born = 0;      // psdr_client_add's serial number: a fetched set answers only for the occupant it was filled with
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

inline int ilog2(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) l++;
    return l;
}

}  // namespace psdr
using namespace psdr;

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
    SpecLayout lay{};                // device layout of the spectrum (natural unless real_fused)
    int nbands = 0, band_H = 0;      // psdr_set_band_layout: band regions (SpecLayout mode 3), halo columns per band
    int seg_len_env = 0;             // PSDR_SEG_LEN: tiles per chain segment (uniform segments, every one with a seam)
    float *d_seamP = nullptr, *d_seamC = nullptr;  // of the current result set
    float *seam_pool[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    size_t seam_cap = 0, seg_cap = 0;  // segments without a carry-in the seamP buffers hold; segments (seamC, flags)
    // chain segments of the fused real second pass for a batch of `nframes` frames (forward.hip: seg_plan)
    struct SegPlan {
        int nframes = 0;
        unsigned nsegs = 0, nseam = 0;  // the nseam segments without a carry-in come first
        bool handoff = false;           // the others get their carried row through memory inside the launch
        uint4 *d_tab = nullptr;
    };
    std::vector<SegPlan> seg_plans;  // one per batch size seen
    unsigned *d_segflag = nullptr;   // [seg_cap] epoch of the launch that last published the segment's carry-out; behind it
                                     // [2][seg_cap] the fallback marks of the two result sets (fft_pass.h: segmark)
    unsigned seg_epoch = 0;
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
    hipStream_t own_stream = nullptr, own_side = nullptr;
    // pass 1 runs on its own stream so that pass 1 of batch i+1 fills the CUs that pass 2 of
    // batch i leaves one by one (persistent work-groups: launch ramp, prologue and tail of one
    // kernel overlap with the other kernel's steady state); Y is double-buffered for that
    hipStream_t p1 = nullptr;
    cf *d_Y = nullptr;  // the inter-pass array, max_batch frames of M complex values
    // TileQueue counters: a ring of TICKET_SLOTS launches x 8 counters per pass; half the ring is
    // re-zeroed (in stream order) whenever the other half starts being used
    unsigned *d_tickets[2] = {nullptr, nullptr};
    unsigned ticket_pos[2] = {0, 0};
    bool input_on_main = false;  // level-1 H2D staging was enqueued on the main stream
    hipEvent_t ev_in = nullptr;
    hipEvent_t ev_fft_done = nullptr, ev_side_done = nullptr;
    bool side_pending = false;
    // Result buffers (spectrum, pyramid, level powers) exist twice: batch b+1 is produced
    // into the other set while the side stream still consumes batch b, so the FFT stream only
    // ever waits for the consumers of batch b-1.  d_spec/d_q/d_qt/d_pscr point at the set of
    // the LAST processed batch.
    // (forward.hip: enqueue_tails - what the tail kernels of the batch just transformed need)
    bool tails_pending = false;
    const struct SegPlan *tails_plan = nullptr;
    int tails_nframes = 0;
    int cur_set = 0;
    bool alt_sets = false;  // alternate the sets also on a caller's stream (a group's root: the peers read batch b's spectrum while b + 1 is transformed)
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
    cf *d_Wn = nullptr, *d_ypost = nullptr, *d_gscratch = nullptr, *d_bb_tail = nullptr,
       *d_bb_last = nullptr;
    // post-demodulation chain (postchain.h), allocated by psdr_set_post_chain
    bool post_on = false;
    bool post_ready = false;   // the chain's buffers, events and streams exist (psdr_set_post_chain's one-time set-up went through)
    int opt_pc_streams = 0;    // PSDR_OPT_POST_CHAIN_STREAMS: 0 = creation order (deterministic), 1 = chosen by measurement
    int opt_pc_pcm16 = 0;      // PSDR_OPT_POST_CHAIN_PCM16: the chain's PCM as int16 rows (half the bytes to the host)
    bool pcm_is16 = false;     // ... of the last chain batch (what a fetch / psdr_read_pcm finds in the PCM buffer)
    int opt_pc_agc = 1;        // PSDR_OPT_POST_CHAIN_AGC: 1 = chunk maxima + k_pc_agc where it applies, 0 = the five-kernel form
    PostArgs post{};
    // The chain is a pipeline across batches (round 5), in the order of a batch's data:
    //   side     index, gather (behind the demodulation)
    //   pc_s[0]  moving averages (sequential), history
    //   pc_s[2]  look-ahead peak, w_t, gain recurrence (sequential), int16 output
    // (pc_s[1] is created and left idle: postchain.hip says why.)  The two sequential kernels (1.5 - 2 ms per 512 frames each, whatever the client count)
    // are what a stream must not share: rounds 3-4 had the moving averages AND the six short kernels of their stage on
    // one stream - longer than the step it hid behind (3.2 ms against 2.6).  What the stages hand on rotates over PC_SETS
    // sets, so a batch's chain may take up to two steps longer than a step.
    static constexpr int PC_SETS = 3;
    hipStream_t pc_s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_pc[4][PC_SETS] = {};  // [stage][set]: the stage's kernels of the batch that used the set are done
    float *post_x[PC_SETS] = {}, *post_m1[PC_SETS] = {}, *post_v1[PC_SETS] = {};
    float *post_p[PC_SETS] = {}, *post_s[PC_SETS] = {}, *post_sm[PC_SETS] = {};  // prefix maxima then g_t / w_t / sub-block maxima
    int *post_fstart[PC_SETS] = {}, *post_len[PC_SETS] = {};
    // the AGC in one kernel behind chunk maxima (postchain.h k_pc_cm / k_pc_cscan / k_pc_agc): per set like P / S
    float *post_cm[PC_SETS] = {}, *post_cp[PC_SETS] = {}, *post_cs[PC_SETS] = {};
    int *post_falive[PC_SETS] = {};
    bool post_direct = false;  // the last chain batch's moving averages read d_audio themselves (k_pc_ma2 DIRECT)
    bool post_agc_ok = false;  // the rate / audio size allow it (L % 16 == 0, h % 4 == 0, h >= 16, D % 4 == 0)
    uint64_t chain_seq = 0;
    bool chain_pending = false;
    int post_reserve = 8;  // CUs the FFT passes leave free while the chain is on (a multiple of 8: one per XCD); 0: none
    int post_lanes = 32;   // slots per work-group of the chain's two recurrence kernels
    bool post_own = true;  // their waves allocate a whole SIMD's registers
    std::vector<void *> post_allocs;
    // Results of the LAST demodulation batch: d_audio / d_pwr / d_nan point into one of TWO sets that alternate from batch to
    // batch, so that the copies of batch b to the host (psdr_fetch_begin) run beside the demodulation of batch b + 1 instead
    // of holding it up (256 clients: 96 MB per step, 1.7 ms on the link - longer than the demodulation it follows)
    float *d_pwr = nullptr, *d_audio = nullptr, *d_real_prev = nullptr;
    int *d_nan = nullptr;
    float *audio_pool[2] = {nullptr, nullptr}, *pwr_pool[2] = {nullptr, nullptr};
    int *nan_pool[2] = {nullptr, nullptr};
    int32_t *pcm_pool[2] = {nullptr, nullptr};  // post chain: post.pcm points at the last batch's
    int out_set = 0, pcm_set = 0;
    unsigned *d_ssb_mark = nullptr;  // [slots] DemodArgs::ssb_mark (demod.h): USB / LSB batches that need the frame-ordered NaN guard
    ParamRing client_ring;
    int last_demod_frames = 0;
    uint64_t demod_seq = 0;  // number of demodulation batches so far (AudioSlot::last_seq)
    // psdr_fetch_begin / _end / psdr_fetch_batch: the served end of the path (src/signal.cpp:283-291, src/audio.cpp:26-44,
    // src/waterfall.cpp:44-51 hand HOST buffers to the encoders).  A ring of pinned host sets, [slot][frame][...] each; a fetch is
    // enqueued on `fetch_stream` behind the kernels that produce the batch's results and runs beside the next batch's passes;
    // the device buffers it reads exist once, so the next batch's WRITERS (demodulation, waterfall gather, the chain's
    // output kernel) wait for `done` of the newest fetch in stream order (fetch_guard).
    struct FetchSet {
        float *audio = nullptr, *pwr = nullptr;
        int32_t *nan = nullptr, *pcm = nullptr;
        int8_t *wf = nullptr;
        size_t wf_cap = 0;
        hipEvent_t done = nullptr;       // every copy of the fetch on the first copy stream has landed
        hipEvent_t ev_pcm = nullptr;     // ... and the PCM (its own copy stream: it waits for the post chain, up to two steps late)
        bool has_pcm = false;
        bool pcm16 = false;              // its PCM rows are int16 (PSDR_OPT_POST_CHAIN_PCM16 at that batch)
        hipEvent_t ev_wf = nullptr;      // ... the waterfall rows (first in the copy stream: d_wfout exists once)
        hipEvent_t ev_audio = nullptr;   // ... pwr, NaN flags, float audio (what the demodulation of batch b + 2 overwrites)
        bool inflight = false;
        unsigned what = 0;     // PSDR_FETCH_* bits the copies covered
        int frames = 0;        // frames of the demodulation batch (0: none was fetched)
        uint64_t seq = 0;      // its demod_seq
        struct Win {
            uint64_t last_seq = 0, born = 0;
            int l = 0, r = 0;
            double mid = 0;
        };
        std::vector<Win> win;      // per audio slot: the window the batch was demodulated with
        std::vector<WfSlot> wfm;   // per waterfall slot: what psdr_waterfall_batch gathered (out_off, nsent, b_*)
    } fset[PSDR_FETCH_SETS];
    uint64_t slot_births = 0;      // psdr_client_add calls so far (AudioSlot::born)
    int fetch_fill = 0;            // the set the next psdr_fetch_begin fills (a ring: the oldest in flight is fetch_fill - fetch_inflight)
    int fetch_inflight = 0;        // fetches begun and not yet ended
    int fetch_cur = -1;            // the set psdr_fetched_* read: completed by the last psdr_fetch_end
    hipStream_t fetch_stream = nullptr, fetch_stream_pcm = nullptr;
    hipEvent_t ev_fetch_src = nullptr;
    // what the next WRITER of a device-side result buffer waits for (stream-ordered): the newest fetch that read it
    hipEvent_t guard_wf = nullptr, guard_audio[2] = {nullptr, nullptr}, guard_pcm[2] = {nullptr, nullptr};

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

namespace psdr {

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

constexpr unsigned TICKET_SLOTS = 64;
// tile widths: T = min(16384/L, other dimension)
inline int pick_T(int L, int other) { return std::min(16384 / L, other); }
// persistent launch: as many work-groups as the CUs hold (LDS-limited), a multiple of 8 (XCD
// round-robin of the TileQueue), or one per tile when there are fewer tiles than that
inline unsigned persistent_grid(psdr_ctx *c, unsigned blocks, size_t lds) {
    unsigned cap = ((unsigned)c->num_cus * (unsigned)std::max<size_t>(1, 160 * 1024 / lds)) & ~7u;
    // post chain on: one to three CUs per XCD stay free of the passes' work-groups - the home of the chain's recurrence waves
    // (postchain.hip: they allocate a whole SIMD's registers each, or ask for more LDS than a pass leaves, so they land
    // THERE and nowhere else; beside a pass's eight waves and the other consumers a recurrence runs 1.9 - 2.8 ms per 512
    // frames, longer than the step).  0.5 % of the plain step per CU and XCD.
    unsigned reserve = c->post_on ? (unsigned)c->post_reserve : 0u;
    if (const char *e = psdr_tuning_env("PSDR_GRID_RESERVE")) reserve = (unsigned)atoi(e) & ~7u;  // (tuning build)
    if (lds * 2 > 160 * 1024 && cap >= reserve + 8u) cap -= reserve;
    return blocks <= cap ? blocks : std::max(cap, 8u);
}

// context.hip
int drain(psdr_ctx *c);
// demod.hip: the next writer of a device-side result buffer on stream `st` waits for the fetch that read it (ev may be null)
int fetch_guard_wait(psdr_ctx *c, hipStream_t st, hipEvent_t ev);
void resolve_pending(psdr_ctx *c);
void resolve_kclock(psdr_ctx *c);
int reset_kclock(psdr_ctx *c);
// forward.hip
void select_set(psdr_ctx *c, int set);
int real_seg_len(const psdr_ctx *c, int nframes);
void seg_plan_counts(const psdr_ctx *c, int nframes, unsigned *nsegs, unsigned *nseam, bool *handoff);
int seg_plan(psdr_ctx *c, int nframes, const psdr_ctx::SegPlan **out);  // (built and uploaded on first use of a batch size)
int process_frames(psdr_ctx *c, const void *d_halves, int nframes, int fmt, hipEvent_t ev_raw_consumed = nullptr);
// pass1.hip / pass2.hip (Pass1Args / Pass2Args: fft_pass.h)
struct Pass1Args;
struct Pass2Args;
int launch_pass1(psdr_ctx *c, int L, int T, int sb, const Pass1Args &a, unsigned blocks, bool pair);
int launch_pass2(psdr_ctx *c, int L, int T, bool fused, const Pass2Args &a, unsigned blocks);
int launch_pass2_band(psdr_ctx *c, const Pass2Args &a, unsigned blocks);
int launch_pass2_real(psdr_ctx *c, const Pass2Args &a);
// postchain.hip: the chain's kernels for the batch demod_impl has just enqueued; *last_user = the last stream that reads
// the client parameter block
int post_chain_enqueue(psdr_ctx *c, const ClientParams *d_clients, const int *d_slot_ci, int nact, int npaused, int nframes,
                       hipStream_t *last_user);

}  // namespace psdr
