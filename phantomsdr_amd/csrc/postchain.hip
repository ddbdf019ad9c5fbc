// postchain.hip - the post-demodulation chain of AudioClient::send_audio (src/signal.cpp:277-284) for all clients of a
// batch: DC blocker, AGC, int16 conversion (postchain.h) as a four-stage pipeline across batches over three rotating buffer sets.
#include <chrono>

#include "ctx.h"
#include "postchain.h"

// ---- post-demodulation chain ---------------------------------------------------
// Which hardware queue a stream gets is the runtime's business (round robin over the process's queues: it depends on every
// stream the process ever made), and it matters: a busy queue on the SAME pipe of the command processor as the main
// stream's makes every launch of the passes start 50 - 60 us late instead of 6 - 9 (4 % of the step; in a second context of
// the same process - bench.py's sub-workloads - the chain's stream landed exactly there: +35 % instead of +16 % with 256
// clients; on the side stream's pipe it is worse: its many short kernels wait, the passes wait for them, +52 %).  HIP
// does not tell, so the streams are CHOSEN BY MEASUREMENT when the chain is first enabled: of six candidates, the two
// beside which sixteen empty kernels on the main stream AND on the side stream take the least time while the candidate
// runs a 2 ms kernel (2 us against 6 per launch) - and which run side by side with each other and with the side stream
// (two streams can share a queue).  ~60 ms, once per context.
namespace {
__global__ void k_pc_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void k_pc_nop() {}

int pc_launch_gap_us(psdr_ctx *c, hipStream_t cand, hipStream_t on, double *us) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    const unsigned long long ticks = (unsigned long long)(c->wall_clock_khz * 2.0);  // 2 ms
    hipLaunchKernelGGL(k_pc_spin, dim3(1), dim3(64), 0, cand, ticks);
    HIPCHK(hipEventRecord(a, on));
    for (int i = 0; i < 16; i++) hipLaunchKernelGGL(k_pc_nop, dim3(1), dim3(64), 0, on);
    HIPCHK(hipEventRecord(b, on));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    HIPCHK(hipStreamSynchronize(cand));
    hipEventDestroy(a);
    hipEventDestroy(b);
    *us = ms * 1e3 / 16.0;
    return PSDR_OK;
}
// do two streams run side by side?  (2 ms on either: ~2 ms together, ~4 ms in the same queue)
int pc_side_by_side(psdr_ctx *c, hipStream_t s1, hipStream_t s2, bool *yes) {
    const unsigned long long ticks = (unsigned long long)(c->wall_clock_khz * 2.0);
    HIPCHK(hipStreamSynchronize(s1));
    HIPCHK(hipStreamSynchronize(s2));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_pc_spin, dim3(1), dim3(64), 0, s1, ticks);
    hipLaunchKernelGGL(k_pc_spin, dim3(1), dim3(64), 0, s2, ticks);
    HIPCHK(hipStreamSynchronize(s1));
    HIPCHK(hipStreamSynchronize(s2));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *yes = ms < 3.2;
    return PSDR_OK;
}
int pc_pick_streams(psdr_ctx *c) {
    constexpr int NC = 6;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t cand[NC] = {};
    double gap[NC] = {};
    struct Guard {  // an error path leaves no candidate stream behind
        hipStream_t *c;
        bool keep = false;
        ~Guard() {
            if (!keep)
                for (int i = 0; i < NC; i++)
                    if (c[i]) hipStreamDestroy(c[i]);
        }
    } guard{cand};
    for (hipStream_t &st : cand) HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    hipLaunchKernelGGL(k_pc_nop, dim3(1), dim3(64), 0, c->stream);  // (code objects loaded, queues created)
    if (c->side != c->stream) hipLaunchKernelGGL(k_pc_nop, dim3(1), dim3(64), 0, c->side);
    for (hipStream_t st : cand) hipLaunchKernelGGL(k_pc_spin, dim3(1), dim3(64), 0, st, 1ull);
    HIPCHK(hipDeviceSynchronize());
    int order[NC];
    double gside[NC] = {}, score[NC] = {};
    for (int i = 0; i < NC; i++) {  // (the better of two readings each: beside the main stream, beside the side stream)
        double g1 = 0, g2 = 0, g3 = 0, g4 = 0;
        int rc = pc_launch_gap_us(c, cand[i], c->stream, &g1);
        if (!rc) rc = pc_launch_gap_us(c, cand[i], c->stream, &g2);
        if (!rc && c->side != c->stream) rc = pc_launch_gap_us(c, cand[i], c->side, &g3);
        if (!rc && c->side != c->stream) rc = pc_launch_gap_us(c, cand[i], c->side, &g4);
        if (rc) return rc;
        gap[i] = std::min(g1, g2);
        gside[i] = std::min(g3, g4);
        order[i] = i;
    }
    {
        const double m0 = std::max(*std::min_element(gap, gap + NC), 0.1), m1 = std::max(*std::min_element(gside, gside + NC), 0.1);
        for (int i = 0; i < NC; i++) score[i] = std::max(gap[i] / m0, c->side != c->stream ? gside[i] / m1 : 0.0);
    }
    std::stable_sort(order, order + NC, [&](int x, int y) { return score[x] < score[y]; });
    // The moving averages' stream: the quietest candidate (beside both).  The gain's stream: one that is quiet beside the
    // MAIN stream but shares the side stream's pipe, if there is one - two chain streams on the same quiet pipe measured
    // +46 % with 256 clients, quiet + side's pipe +14-16 % (five runs; the first context of a process gets that by itself) -
    // else the next quietest.  Either must run side by side with the other and with the side stream.
    auto beside = [&](int i, int j, bool *ok) -> int {  // j < 0: the side stream
        *ok = true;
        if (j < 0 && c->side == c->stream) return PSDR_OK;
        return pc_side_by_side(c, cand[i], j < 0 ? c->side : cand[j], ok);
    };
    int first = -1, second = -1;
    for (int k = 0; k < NC && first < 0; k++) {
        bool ok = true;
        int rc = beside(order[k], -1, &ok);
        if (rc) return rc;
        if (ok) first = order[k];
    }
    if (first < 0) first = order[0];
    const double m0 = std::max(*std::min_element(gap, gap + NC), 0.1), m1 = std::max(*std::min_element(gside, gside + NC), 0.1);
    for (int pass = 0; pass < 2 && second < 0; pass++)
        for (int k = 0; k < NC && second < 0; k++) {
            const int i = order[k];
            if (i == first) continue;
            const bool quiet_main = gap[i] < 1.6 * m0, on_side_pipe = c->side != c->stream && gside[i] > 1.6 * m1;
            if (pass == 0 && !(quiet_main && on_side_pipe)) continue;
            bool ok = true, ok2 = true;
            int rc = beside(i, first, &ok);
            if (!rc) rc = beside(i, -1, &ok2);
            if (rc) return rc;
            if (ok && ok2) second = i;
        }
    if (second < 0) second = order[0] == first ? order[1] : order[0];
    int third = -1;
    for (int k = 0; k < NC; k++)
        if (order[k] != first && order[k] != second) {
            third = order[k];
            break;
        }
    if (psdr_tuning_env("PSDR_PC_VERBOSE")) {
        fprintf(stderr, "psdr post chain: launch gap of the main / side stream beside each candidate stream [us]:");
        for (int i = 0; i < NC; i++) fprintf(stderr, " %.1f/%.1f", gap[i], gside[i]);
        fprintf(stderr, " -> streams %d and %d\n", first, second);
    }
    c->pc_s[0] = cand[first];
    c->pc_s[2] = cand[second];
    c->pc_s[1] = cand[third];
    guard.keep = true;
    for (int i = 0; i < NC; i++)
        if (i != first && i != second && i != third) hipStreamDestroy(cand[i]);
    return PSDR_OK;
}
}  // namespace

extern "C" int psdr_set_option(psdr_ctx *c, int option, int value) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    switch (option) {
    case PSDR_OPT_POST_CHAIN_STREAMS:
        if (value != 0 && value != 1) return fail(PSDR_ERR_INVALID, "PSDR_OPT_POST_CHAIN_STREAMS: 0 (creation order) or 1 (measured), not %d", value);
        if (c->post_ready) return fail(PSDR_ERR_STATE, "PSDR_OPT_POST_CHAIN_STREAMS after the post chain was set up");
        c->opt_pc_streams = value;
        return PSDR_OK;
    case PSDR_OPT_POST_CHAIN_PCM16: {
        if (value != 0 && value != 1) return fail(PSDR_ERR_INVALID, "PSDR_OPT_POST_CHAIN_PCM16: 0 (int32 rows, the reference's buffer) or 1 (int16 rows), not %d", value);
        HIPCHK(hipSetDevice(c->device));
        const int rc = drain(c);
        if (rc) return rc;
        c->opt_pc_pcm16 = value;
        return PSDR_OK;
    }
    case PSDR_OPT_POST_CHAIN_AGC: {
        if (value != 0 && value != 1) return fail(PSDR_ERR_INVALID, "PSDR_OPT_POST_CHAIN_AGC: 0 (five kernels) or 1 (one kernel behind chunk maxima), not %d", value);
        HIPCHK(hipSetDevice(c->device));
        const int rc = drain(c);
        if (rc) return rc;
        c->opt_pc_agc = value;
        return PSDR_OK;
    }
    default:
        return fail(PSDR_ERR_INVALID, "unknown option %d", option);
    }
}
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
    if (!c->post_ready) {
        // (one-time set-up; `post_ready` is raised only when EVERY allocation and the stream choice went through - a failure
        // half-way gives everything back, so a retry starts from scratch instead of running the chain on null pointers)
        auto undo = [&]() {
            for (void *q : c->post_allocs) hipFree(q);
            c->post_allocs.clear();
            c->post = PostArgs{};
            c->pcm_pool[0] = c->pcm_pool[1] = nullptr;
            for (int i = 0; i < psdr_ctx::PC_SETS; i++)
                c->post_fstart[i] = c->post_len[i] = c->post_falive[i] = nullptr, c->post_x[i] = c->post_m1[i] = c->post_v1[i] = c->post_p[i] = c->post_s[i] = c->post_sm[i] = c->post_cm[i] = c->post_cp[i] = c->post_cs[i] = nullptr;
            c->post_agc_ok = false;
        };
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
        // (D up to 12288 - audio rates up to 4.6 MHz - as in rounds 2-4; the AGC look-ahead L has no such limit: k_pc_submax /
        // k_pc_prefix / k_pc_want walk it in 256-row pieces)
        if (a.D < 1 || a.L < 2 || a.D > 12288)
            return fail(PSDR_ERR_UNSUPPORTED, "audio_rate %d: DC delay %d / look-ahead %d unsupported", rate, a.D, a.L);
        auto alloc = [&](void **ptr, size_t bytes) -> int {
            HIPCHK(hipMalloc(ptr, std::max<size_t>(bytes, 16)));
            c->post_allocs.push_back(*ptr);
            HIPCHK(hipMemset(*ptr, 0, std::max<size_t>(bytes, 16)));
            return PSDR_OK;
        };
        // lane-interleaved streams (postchain.h): pitches are multiples of 4 floats per slot, + padding for the
        // blocked kernels' look-ahead; a block of 64 slots is allocated whole
        const size_t S64 = (S + 63) / 64 * 64;
        a.px = ((size_t)a.D + Tm + PSDR_PC_PAD + 3) & ~(size_t)3;
        a.vo = (4 - ((a.L - 1) & 3)) & 3;
        a.pv = ((size_t)a.vo + (size_t)a.L - 1 + Tm + PSDR_PC_PAD + 3) & ~(size_t)3;
        // look-ahead peak: sub-blocks of at most 256 rows of a block of L rows (postchain.h: a wave per sub-block and 64 slots,
        // 16 rows per round trip to memory - beside the FFT passes a round trip is microseconds)
        a.nsub = (a.L + 255) / 256;
        a.sb = (a.L + a.nsub - 1) / a.nsub;
        const size_t nblk = (((size_t)a.L - 1 + Tm) + a.L - 1) / a.L;
        // the AGC in one kernel (postchain.h k_pc_agc): whole chunks of 16 floats must line up with sample 0's row and with the
        // row groups of a frame; stream position / h by one 32-bit multiplication
        c->post_agc_ok = (a.L % 16) == 0 && a.L >= 32 && a.vo == 1 && (h % 4) == 0 && h >= 16 && (a.D % 4) == 0 && (Tm + 4096) * h < ((size_t)1 << 32);
        a.nch = (int)((size_t)a.L / 16 + (Tm + 15) / 16 + 8);
        a.h_magic = (unsigned)((((uint64_t)1 << 32) + h - 1) / h);
        int rc = 0;
        for (int i = 0; i < psdr_ctx::PC_SETS && !rc; i++) {
            rc = alloc((void **)&c->post_fstart[i], S * c->max_batch * sizeof(int));
            if (!rc) rc = alloc((void **)&c->post_len[i], S * sizeof(int));
            if (!rc) rc = alloc((void **)&c->post_x[i], a.px * S64 * sizeof(float));
            if (!rc) rc = alloc((void **)&c->post_m1[i], a.px * S64 * sizeof(float));
            if (!rc) rc = alloc((void **)&c->post_v1[i], a.pv * S64 * sizeof(float));
            if (!rc) rc = alloc((void **)&c->post_p[i], a.pv * S64 * sizeof(float));
            if (!rc) rc = alloc((void **)&c->post_s[i], a.pv * S64 * sizeof(float));
            if (!rc) rc = alloc((void **)&c->post_sm[i], S64 * nblk * a.nsub * sizeof(float));
            if (c->post_agc_ok) {
                if (!rc) rc = alloc((void **)&c->post_cm[i], S64 * a.nch * sizeof(float));
                if (!rc) rc = alloc((void **)&c->post_cp[i], S64 * a.nch * sizeof(float));
                if (!rc) rc = alloc((void **)&c->post_cs[i], S64 * a.nch * sizeof(float));
                if (!rc) rc = alloc((void **)&c->post_falive[i], S64 * c->max_batch * sizeof(int));
            }
            for (auto &stage : c->ev_pc)
                if (!rc && !stage[i] && hipEventCreateWithFlags(&stage[i], hipEventDisableTiming) != hipSuccess)
                    rc = fail(PSDR_ERR_HIP, "post chain: event creation failed");
        }
        for (int k = 0; k < 2 && !rc; k++) rc = alloc((void **)&c->pcm_pool[k], S * Tm * sizeof(int32_t));
        a.pcm = c->pcm_pool[0];
        if (!rc) rc = alloc((void **)&a.pcm_dump, 64 * (1 + PC_AGC_NP) * 16);
        if (!rc) rc = alloc((void **)&a.dc_s1, S * sizeof(float));
        if (!rc) rc = alloc((void **)&a.dc_s2, S * sizeof(float));
        if (!rc) rc = alloc((void **)&a.agc_gain, S * sizeof(float));
        if (!rc) rc = alloc((void **)&a.agc_n0, S * sizeof(int));
        if (rc) {
            const std::string msg = psdr_last_error();
            undo();
            return fail(rc == PSDR_ERR_HIP ? PSDR_ERR_NOMEM : rc, "post chain set-up: %s", msg.c_str());
        }
        // The chain's streams (only a context that owns its side stream pipelines the chain: with a caller's stream - group
        // members - everything rides on that one stream and no chain stream is made).  PSDR_OPT_POST_CHAIN_STREAMS:
        //   0 (default)  three streams in creation order, the first and the third used: deterministic; the FIRST context of a
        //                process gets the quiet queues by itself
        //   1 (opt-in)   chosen by measurement (pc_pick_streams: ~60 ms, wall-clock thresholds): what a process that creates
        //                several contexts (bench.py's sub-workloads) needs to see +3 % instead of +15 %; a measurement that
        //                fails falls back to creation order instead of failing the call
        if (!c->pc_s[0] && c->side != c->stream) {
            int pick = c->opt_pc_streams;
            if (const char *e = psdr_tuning_env("PSDR_PC_PICK")) pick = atoi(e) != 0;  // (tuning build)
            if (pick && pc_pick_streams(c) != PSDR_OK) {
                for (hipStream_t &st : c->pc_s) {
                    if (st) hipStreamDestroy(st);
                    st = nullptr;
                }
            }
            if (!c->pc_s[0]) {
                int lo = 0, hi = 0;
                hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
                for (hipStream_t &st : c->pc_s)
                    if (e == hipSuccess) e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
                if (e != hipSuccess) {
                    for (hipStream_t &st : c->pc_s) {
                        if (st) hipStreamDestroy(st);
                        st = nullptr;
                    }
                    undo();
                    return fail(PSDR_ERR_HIP, "post chain: stream creation failed: %s", hipGetErrorString(e));
                }
            }
        }
        c->post_ready = true;
    }
    {
        // The two recurrence kernels (k_pc_ma2, k_pc_gain: postchain.h): 32 slots per work-group (half a wave in use, 512-byte
        // memory operations: 256 clients 3.80-3.90 -> 3.55-3.67 ms per step, level at 16 - profiles/r05_post_chain_lanes.jsonl),
        // whole waves beyond 512 slots.  Their waves own a SIMD each (512 registers allocated) as long as the CUs the passes
        // leave free hold them all: 2 kernels x 2 waves per work-group = one CU per work-group of either; one, two or three
        // CUs per XCD stay free (8: +0.5 % on the plain step, 16: +1 %, 24: +2.5 %).
        const unsigned groups = (unsigned)((c->aslots.size() + 63) / 64);
        c->post_lanes = groups <= 8 ? 32 : 64;
        if (const char *e = psdr_tuning_env("PSDR_PC_LANES")) c->post_lanes = atoi(e) == 16 ? 16 : atoi(e) == 32 ? 32 : 64;  // (tuning build)
        const unsigned rgroups = groups * (unsigned)(64 / c->post_lanes);
        c->post_reserve = (int)std::min(24u, 8u * (1u + rgroups / 8u));
        if (const char *e = psdr_tuning_env("PSDR_PC_RESERVE")) c->post_reserve = atoi(e) & ~7;  // (tuning build)
        c->post_own = (int)rgroups <= c->post_reserve;
        if (const char *e = psdr_tuning_env("PSDR_PC_OWN")) c->post_own = atoi(e) != 0;  // (tuning build)
    }
    c->post_on = true;
    return PSDR_OK;
}

// the chain's kernels for the batch whose demodulation has just been enqueued on c->side (d_clients: its parameter
// block - the nact active clients first, then npaused paused ones with an empty stream - and d_slot_ci: the list index of
// every slot's client, -1 = not listed)
int psdr::post_chain_enqueue(psdr_ctx *c, const ClientParams *d_clients, const int *d_slot_ci, int nact, int npaused, int nframes,
                             hipStream_t *last_user) {
    constexpr int NS = psdr_ctx::PC_SETS;
    const int set = (int)(c->chain_seq % NS), nxt = (set + 1) % NS;
    const uint64_t seq = c->chain_seq;
    const bool piped = c->side != c->stream && c->pc_s[0] != nullptr;
    // Streams: stage 0 rides behind the demodulation on `side` (two short kernels), stage 2 in front of stage 3 on ITS stream
    // (they are a chain anyway), the moving averages on the other.  Hardware queues are what is scarce: with four chain
    // streams the fourth shared a queue with the third (the gain recurrence in front of the next batch's peak), and with
    // three - five busy queues with the main and the side stream - every second launch of the PASSES started 50 - 60 us
    // late (6 - 9 us with four queues, as without the chain): 4 % of the step.
    // WHICH queues matters as much (tools/runs/r05_w.sh, r05_y.sh; rocprofv3 Queue_Id): the chain on queues 4 and 6 leaves the
    // passes' launches alone, on 4 and 5 it delays them as three chain queues do - queue 5 shares its pipe of the command
    // processor with queue 1, the main stream's.  pc_pick_streams() chose pc_s[0] and pc_s[2] by that measure.
    hipStream_t sg = c->side, sm = piped ? c->pc_s[0] : c->side, sc = piped ? c->pc_s[2] : c->side, sp = sc;
    if (const char *e = psdr_tuning_env("PSDR_PC_STREAMS")) {  // (tuning build)
        if (atoi(e) == 3 && piped) sp = c->pc_s[1];           // the peak kernels on a stream of their own
        if (atoi(e) == 2 && piped) sp = sc = c->pc_s[1];       // the two chain streams on neighbouring queues
        if (atoi(e) == 1 && piped) sp = sm;                    // the peak kernels behind the moving averages
    }
    // the prefix maxima ride behind the moving averages, w_t in front of the gain recurrence (with 256 clients the gain's
    // stream is the longer one: 0.8 ms of peak kernels + 1.9 + 0.3 against 2.3)
    bool split_peak = piped && sp == sc;
    if (const char *e = psdr_tuning_env("PSDR_PC_SPLIT_PEAK")) split_peak = split_peak && atoi(e) != 0;  // (tuning build)
    // the recurrence kernels go to the CUs the passes leave free (ctx.h persistent_grid): with this much LDS they do not fit
    // beside a pass's work-group (128 KiB of 160)
    size_t home_lds = c->post_reserve > 0 ? 34 * 1024 : 0;
    const bool rows4 = (c->post.h & 3) == 0 && (c->post.D & 3) == 0;  // every frame starts on a row group: the lane = slot gather / output
    // The AGC as chunk maxima + ONE four-wave kernel (postchain.h k_pc_agc: V1 read twice, the PCM written once - the five
    // kernels of the other form pass over a stream eleven times) whenever the rate allows it and its work-groups - a whole
    // CU each: four waves that own their SIMD - have the CUs the passes leave free
    bool agc_fused = c->post_agc_ok && c->post_own && c->opt_pc_agc != 0 && c->post_lanes <= 32;
    if (const char *e = psdr_tuning_env("PSDR_PC_FUSED")) agc_fused = agc_fused && atoi(e) != 0;  // (tuning build)
    // ... and the chunk maxima of the new samples from a third wave of the moving averages (k_pc_ma2 CMW) instead of a pass over
    // V1, while the free CUs hold a three-wave work-group of those beside every four-wave one of the AGC (a CU each)
    bool ma_cmw = false;  // (set below, once the work-group count is known)
    // this batch's PCM goes to the other of two buffers (the copy of the last batch's to the host may still read its own)
    c->pcm_set ^= 1;
    c->post.pcm = c->pcm_pool[c->pcm_set];
    PostArgs pa = c->post;
    pa.pcm16 = c->opt_pc_pcm16;
    c->pcm_is16 = pa.pcm16 != 0;
    pa.audio = c->d_audio;  // (of THIS demodulation batch: the result sets alternate)
    pa.nan_flags = c->d_nan;
    pa.clients = d_clients;
    pa.slot_ci = d_slot_ci;
    pa.nact = nact + npaused;
    pa.nframes = nframes;
    pa.X = c->post_x[set], pa.Xn = c->post_x[nxt];
    pa.M1 = c->post_m1[set], pa.M1n = c->post_m1[nxt];
    pa.V1 = c->post_v1[set], pa.V1n = c->post_v1[nxt];
    pa.P = c->post_p[set];
    pa.S = c->post_s[set];
    pa.SM = c->post_sm[set];
    pa.fstart = c->post_fstart[set];
    pa.len = c->post_len[set];
    pa.CM = c->post_cm[set], pa.CP = c->post_cp[set], pa.CS = c->post_cs[set];
    pa.falive = c->post_falive[set];
    pa.ma_fused = pa.D == 32 ? 1 : 0;  // both averages in one loop (postchain.h)
    // ... which may read the demodulator's rows themselves instead of a gathered copy (k_pc_ma2 DIRECT; part of the round-6
    // form of the chain, PSDR_OPT_POST_CHAIN_AGC = 1; the demodulation two batches on waits for this batch's stage 1: demod.hip)
    pa.direct = (pa.ma_fused && rows4 && c->opt_pc_agc != 0) ? 1 : 0;
    if (const char *e = psdr_tuning_env("PSDR_PC_DIRECT")) pa.direct = pa.direct && atoi(e) != 0;  // (tuning build)
    c->post_direct = pa.direct != 0;
    const int nall = nact + npaused;
    const unsigned groups = (unsigned)((pa.slots + 63) / 64);
    pa.lanes = c->post_lanes;
    const unsigned rgroups = groups * (unsigned)(64 / pa.lanes);  // work-groups of each of the two recurrence kernels
    ma_cmw = agc_fused && pa.D == 32 && 2 * (int)rgroups <= c->post_reserve;
    if (const char *e = psdr_tuning_env("PSDR_PC_CMW")) ma_cmw = ma_cmw && atoi(e) != 0;  // (tuning build)
    if (rgroups > 16 || c->post_own) home_lds = 0;  // (waves that own a SIMD fit nowhere else anyway)
    const size_t Tb = (size_t)nframes * pa.h;  // longest possible stream of this batch
    const unsigned nblk = (unsigned)((pa.L - 1 + Tb + pa.L - 1) / pa.L);
    // Who touched what last (sets rotate: batch b uses set b mod 3 and writes the history rows of set b + 1):
    //   X[set] rows >= D, fstart / len[set]   gather(b)      <- moving averages, history, output of batch b - 3
    //   V1[set] rows >= L-1                   averages(b)    <- peak / output of batch b - 3
    //   X / M1 / V1[nxt] history rows         history(b)     <- averages(b - 2) (same stream), peak / output of batch b - 2
    //   P / S / SM[set]                       peak(b)        <- gain / output of batch b - 3
    //   CM / CP / CS[set] (one-kernel AGC)    averages(b) [CMW], chunk maxima / scans(b) on the averages' stream  <- k_pc_agc of batch b - 3
    //   falive[set]                           index(b)       <- k_pc_agc of batch b - 3
    //   d_audio[b mod 2] (DIRECT)             demodulation(b) <- averages / history of batch b - 2 (the wait is in demod.hip)
    // i.e. a stage waits for its predecessor of THIS batch and for the output stage of the batch that had the set.
    auto wait = [&](hipStream_t st, int stage, int which) -> int {
        if (piped) HIPCHK(hipStreamWaitEvent(st, c->ev_pc[stage][which], 0));
        return PSDR_OK;
    };
    auto done = [&](hipStream_t st, int stage) -> int {
        if (piped) HIPCHK(hipEventRecord(c->ev_pc[stage][set], st));
        return PSDR_OK;
    };
    int rc = 0;
    // (tuning build: PSDR_PC_SKIP = bit mask of chain kernels NOT launched - wrong results, a timing bound of what each costs the
    // step: 1 gather, 2 moving averages, 4 history, 8 sub-block maxima, 16 prefix maxima, 32 w_t, 64 gain, 128 int16 output)
    int skip = 0;
    if (const char *e = psdr_tuning_env("PSDR_PC_SKIP")) skip = (int)strtol(e, nullptr, 0);
    {  // ---- stage 0: frame offsets, audio rows -> X
        if (seq >= NS && ((rc = wait(sg, 1, set)) || (rc = wait(sg, 3, set)))) return rc;
        ProfScope ps(c, K_POST, sg);
        hipLaunchKernelGGL(k_pc_index, dim3(nall), dim3(64), 0, sg, pa);
        if (skip & 1)
            ;
        else if (rows4)
            hipLaunchKernelGGL(k_pc_gather4, dim3(groups, nframes), dim3(256), 0, sg, pa);
        else
            hipLaunchKernelGGL(k_pc_gather, dim3(nall, nframes), dim3(256), 0, sg, pa);
        HIPCHK(hipGetLastError());
        if ((rc = done(sg, 0))) return rc;
    }
    {  // ---- stage 1: the DC blocker's two moving averages (sequential), tails -> the next set's history rows
        if ((rc = wait(sm, 0, set))) return rc;
        if (seq >= NS && (rc = wait(sm, 3, set))) return rc;
        if (seq >= NS - 1 && (rc = wait(sm, 3, nxt))) return rc;
        ProfScope ps(c, K_POST, sm);
        if (skip & 2) {
        } else if (pa.ma_fused && ma_cmw) {
            hipLaunchKernelGGL((k_pc_ma2<true, true>), dim3(rgroups), dim3(192), 0, sm, pa);
        } else if (pa.ma_fused) {
            if (c->post_own)
                hipLaunchKernelGGL(k_pc_ma2<true>, dim3(rgroups), dim3(128), 0, sm, pa);
            else
                hipLaunchKernelGGL(k_pc_ma2<false>, dim3(rgroups), dim3(128), home_lds ? home_lds - 17 * 1024 : 0, sm, pa);
        } else if ((pa.D & (pa.D - 1)) == 0 && pa.D >= 16 && (size_t)pa.D * pa.lanes * sizeof(float) <= 128 * 1024) {
            // any other power-of-two delay (48 kHz: 128, 192 kHz: 512): the two-wave form with its ring of sums in LDS
            const size_t ring_lds = (size_t)pa.D * pa.lanes * sizeof(float);
            const void *fn = c->post_own ? (const void *)k_pc_mad<true> : (const void *)k_pc_mad<false>;
            if (c->lds_attr_done.insert(fn).second) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            if (c->post_own)
                hipLaunchKernelGGL(k_pc_mad<true>, dim3(rgroups), dim3(128), ring_lds, sm, pa);
            else
                hipLaunchKernelGGL(k_pc_mad<false>, dim3(rgroups), dim3(128), ring_lds, sm, pa);
        } else if ((pa.D & (pa.D - 1)) == 0) {
            hipLaunchKernelGGL((k_pc_ma<false, true>), dim3(groups), dim3(64), 0, sm, pa);
            hipLaunchKernelGGL((k_pc_ma<true, true>), dim3(groups), dim3(64), 0, sm, pa);
        } else {
            hipLaunchKernelGGL((k_pc_ma<false, false>), dim3(groups), dim3(64), 0, sm, pa);
            hipLaunchKernelGGL((k_pc_ma<true, false>), dim3(groups), dim3(64), 0, sm, pa);
        }
        if (!(skip & 4)) hipLaunchKernelGGL(k_pc_history, dim3(nall), dim3(256), 0, sm, pa);
        HIPCHK(hipGetLastError());
        if ((rc = done(sm, 1))) return rc;
    }
    if (agc_fused) {
        // ---- stage 2: chunk maxima of |V1| and their block scans, behind the moving averages on THEIR stream (P / S / SM are not
        // used; CM / CP / CS of this set were last read by stage 3 of batch b - 3: waited for in stage 1)
        const int nchunks = (int)((pa.vo + pa.L - 1 + Tb + 15) / 16);
        const int W = pa.L / 16 - 1;
        {
            ProfScope ps(c, K_POST, sm);
            const int ncm = ma_cmw ? pa.L / 16 : nchunks;  // (k_pc_ma2 CMW left the new samples' chunk maxima: the history chunks only)
            if (!(skip & 8)) hipLaunchKernelGGL(k_pc_cm, dim3(groups, (ncm + 15) / 16), dim3(64), 0, sm, pa, ncm);
            if (!(skip & 16)) hipLaunchKernelGGL(k_pc_cscan, dim3(groups, (nchunks + W - 1) / W), dim3(64), 0, sm, pa, nchunks, W);
            HIPCHK(hipGetLastError());
        }
        if (sm != sc && ((rc = done(sm, 2)) || (rc = wait(sc, 2, set)))) return rc;
        // ---- stage 3: w_t, the gain recurrence, int16 - one kernel
        ProfScope ps(c, K_POST, sc);
        if ((rc = fetch_guard_wait(c, sc, c->guard_pcm[c->pcm_set]))) return rc;  // what read this PCM buffer two batches ago has landed
        c->guard_pcm[c->pcm_set] = nullptr;
        if (!(skip & 128)) hipLaunchKernelGGL(k_pc_zero, dim3(groups, nframes), dim3(256), 0, sc, pa);
        if (skip & 64) {
        } else if (pa.attack >= pa.release) {
            if (pa.pcm16)
                hipLaunchKernelGGL((k_pc_agc<true, true>), dim3(rgroups), dim3(64 * (1 + PC_AGC_NP)), 0, sc, pa);
            else
                hipLaunchKernelGGL((k_pc_agc<true, false>), dim3(rgroups), dim3(64 * (1 + PC_AGC_NP)), 0, sc, pa);
        } else {
            if (pa.pcm16)
                hipLaunchKernelGGL((k_pc_agc<false, true>), dim3(rgroups), dim3(64 * (1 + PC_AGC_NP)), 0, sc, pa);
            else
                hipLaunchKernelGGL((k_pc_agc<false, false>), dim3(rgroups), dim3(64 * (1 + PC_AGC_NP)), 0, sc, pa);
        }
        HIPCHK(hipGetLastError());
        if ((rc = done(sc, 3))) return rc;
    } else {
        {  // ---- stage 2: look-ahead peak and w_t (parallel along time)
            hipStream_t sp1 = split_peak ? sm : sp;  // (P[set]'s last readers - gain / output of batch b - 3 - were waited for in stage 1)
            if (sp1 != sm && (rc = wait(sp1, 1, set))) return rc;
            if (seq >= NS && sp1 != sm && (rc = wait(sp1, 3, set))) return rc;
            {
                ProfScope ps(c, K_POST, sp1);
                if (pa.nsub > 1 && !(skip & 8)) hipLaunchKernelGGL(k_pc_submax, dim3(groups, nblk * pa.nsub), dim3(64), 0, sp1, pa);
                if (!(skip & 16)) hipLaunchKernelGGL(k_pc_prefix, dim3(groups, nblk * pa.nsub), dim3(64), 0, sp1, pa);
            }
            if (split_peak) {
                if ((rc = done(sp1, 2)) || (rc = wait(sp, 2, set))) return rc;
            }
            {
                ProfScope ps(c, K_POST, sp);
                if (!(skip & 32)) hipLaunchKernelGGL(k_pc_want, dim3(groups, nblk * pa.nsub), dim3(64), 0, sp, pa);
            }
            HIPCHK(hipGetLastError());
            if (!split_peak && (rc = done(sp, 2))) return rc;
        }
        {  // ---- stage 3: the gain recurrence (sequential), int16 output
            if (sp != sc && (rc = wait(sc, 2, set))) return rc;
            ProfScope ps(c, K_POST, sc);
            const size_t glds = home_lds ? home_lds - 8 * 1024 : 0;
            if (skip & 64) {
            } else if (pa.attack >= pa.release) {
                if (c->post_own)
                    hipLaunchKernelGGL((k_pc_gain<true, true>), dim3(rgroups), dim3(128), 0, sc, pa);
                else
                    hipLaunchKernelGGL((k_pc_gain<true, false>), dim3(rgroups), dim3(128), glds, sc, pa);
            } else {
                if (c->post_own)
                    hipLaunchKernelGGL((k_pc_gain<false, true>), dim3(rgroups), dim3(128), 0, sc, pa);
                else
                    hipLaunchKernelGGL((k_pc_gain<false, false>), dim3(rgroups), dim3(128), glds, sc, pa);
            }
            if ((rc = fetch_guard_wait(c, sc, c->guard_pcm[c->pcm_set]))) return rc;  // what read this PCM buffer two batches ago has landed
            c->guard_pcm[c->pcm_set] = nullptr;
            if (skip & 128)
                ;
            else if (rows4)
                hipLaunchKernelGGL(k_pc_out4, dim3(groups, nframes), dim3(256), 0, sc, pa);
            else
                hipLaunchKernelGGL(k_pc_out, dim3(nall, nframes), dim3(256), 0, sc, pa);
            HIPCHK(hipGetLastError());
            if ((rc = done(sc, 3))) return rc;
        }
    }
    if (piped) c->chain_pending = true;
    c->chain_seq++;
    *last_user = sc;
    return PSDR_OK;
}
