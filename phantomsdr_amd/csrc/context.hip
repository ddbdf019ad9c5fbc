// context.hip - lifetime of a context (tables, buffers, streams), Level 1 of the C-ABI (the FFT plug-in: host buffers,
// load_*_input, execute), device-memory helpers, the streaming ingest ring and the instrumentation.  No kernels here.
#include "ctx.h"

static thread_local std::string g_err;
int psdr_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
extern "C" const char *psdr_last_error(void) { return g_err.c_str(); }
extern "C" int psdr_abi_version(void) { return PSDR_ABI_VERSION; }
extern "C" const char *psdr_version(void) {
#ifdef PSDR_TUNING_BUILD
    return "phantomsdr_amd 0.3 (gfx950, tuning build)";  // reads the A/B knobs of psdr_tuning_env(); not the library that ships
#else
    return "phantomsdr_amd 0.3 (gfx950)";
#endif
}

namespace psdr {
const char *kKernelNames[K_COUNT] = {"fft_pass1",  "fft_pass2", "untangle_real", "pyramid_tail",
                                     "demod_idft", "demod_ola", "waterfall_gather", "post_chain",
                                     "real_seam",  "band_pack"};

void resolve_pending(psdr_ctx *c) {
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->p1);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->side);
    for (hipStream_t st : c->pc_s)
        if (st) hipStreamSynchronize(st);
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
}  // namespace psdr

namespace {

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
    F(c->d_segflag);
    for (auto &sp : c->seg_plans) F(sp.d_tab);
    F(c->d_tickets[0]);
    F(c->d_tickets[1]);
    F(c->d_Y);
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
    for (int k = 0; k < 2; k++) {
        F(c->pwr_pool[k]);
        F(c->audio_pool[k]);
        F(c->nan_pool[k]);
    }
    F(c->d_real_prev);
    F(c->d_ssb_mark);
    c->client_ring.destroy();
    c->wf_ring.destroy();
    F(c->d_wfout);
    auto H = [](void *p) {
        if (p) hipHostFree(p);
    };
    H(c->h_out);
    H(c->h_q);
    for (auto &fs : c->fset) {
        H(fs.audio);
        H(fs.pwr);
        H(fs.nan);
        H(fs.pcm);
        H(fs.wf);
        if (fs.done) hipEventDestroy(fs.done);
        if (fs.ev_wf) hipEventDestroy(fs.ev_wf);
        if (fs.ev_pcm) hipEventDestroy(fs.ev_pcm);
        if (fs.ev_audio) hipEventDestroy(fs.ev_audio);
    }
    if (c->ev_fetch_src) hipEventDestroy(c->ev_fetch_src);
    if (c->fetch_stream) hipStreamDestroy(c->fetch_stream);
    if (c->fetch_stream_pcm) hipStreamDestroy(c->fetch_stream_pcm);
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
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->own_side) hipStreamDestroy(c->own_side);
    for (hipStream_t st : c->pc_s)
        if (st) hipStreamDestroy(st);
    for (auto &stage : c->ev_pc)
        for (hipEvent_t e : stage)
            if (e) hipEventDestroy(e);
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
        // work-groups: give their stream the highest priority (the other way round, and the passes' stream at the
        // highest, measured within +-1 %: docs/history.md section 5)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&c->own_side, hipStreamNonBlocking, hi));
    }
    c->stream = c->own_stream;
    c->side = c->own_side;
    // (both passes run on `stream`; the first pass of batch b+1 on a stream of its own beside the second pass of batch b
    // was measured in rounds 1-3 - -12 % at F = 16, nothing from F = 32 on - and taken out in round 5)
    c->p1 = c->own_stream;
    HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
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
                rc = upload(&c->d_UG, make_twiddles((size_t)(c->M1 / c->T2), (size_t)(c->T2 / 2), c->N, -1));  // W_N^{CP g}: the tile's factor
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
    HIPCHK(hipMalloc((void **)&c->d_Y, F * c->M * sizeof(cf)));
    if (c->is_real && !c->real_fused) HIPCHK(hipMalloc((void **)&c->d_Z, F * c->M * sizeof(cf)));
    if (!c->is_real && c->lay.mode) HIPCHK(hipMalloc((void **)&c->d_Z, (c->M + 2) * sizeof(cf)));  // k-order staging
    if (c->real_fused) {
        // one frame of k-order staging for psdr_read_spectrum / psdr_get_output_buffer
        HIPCHK(hipMalloc((void **)&c->d_Z, (c->M + 2) * sizeof(cf)));
        if (const char *e = getenv("PSDR_SEG_LEN")) c->seg_len_env = atoi(e);
        size_t cap = 0, capc = 0;
        for (int nf = 1; nf <= c->max_batch; nf++) {
            unsigned ns, nm;
            bool ho;
            seg_plan_counts(c, nf, &ns, &nm, &ho);
            cap = std::max(cap, (size_t)nm);
            capc = std::max(capc, (size_t)ns);
        }
        c->seam_cap = cap;
        c->seg_cap = capc;
        for (int st = 0; st < 2; st++) {  // part of the double-buffered result sets: k_real_seam is a consumer
            HIPCHK(hipMalloc((void **)&c->seam_pool[st][0], cap * (size_t)c->M2 * (size_t)(c->T2 / 2) * sizeof(float)));  // [segment][M2][couples per tile]
            HIPCHK(hipMalloc((void **)&c->seam_pool[st][1], capc * (size_t)c->M2 * sizeof(float)));
        }
        // flags (inside a launch only), then the fallback marks - ONE ARRAY PER RESULT SET: k_real_seam is a consumer, it may
        // run after the next batch's second pass has started writing its own marks
        HIPCHK(hipMalloc((void **)&c->d_segflag, 3 * capc * sizeof(unsigned)));
        HIPCHK(hipMemset(c->d_segflag, 0, 3 * capc * sizeof(unsigned)));
        {  // the plans of the two batch sizes every caller uses, now rather than in the first batch (a synchronous upload)
            const psdr_ctx::SegPlan *sp;
            int rc = seg_plan(c, c->max_batch, &sp);
            if (!rc && c->max_batch > 1) rc = seg_plan(c, 1, &sp);
            if (rc) return rc;
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
            if (const char *e = getenv("PSDR_DEMOD_CHAIN")) c->demod_chain = atoi(e) != 0;
            if (const char *e = getenv("PSDR_DEMOD_K")) c->demod_chain_k = std::max(1, atoi(e));
        }
        const size_t S = (size_t)std::max(1, g.max_clients);
        c->aslots.resize(S);
        HIPCHK(hipMalloc((void **)&c->d_ypost, S * F * n * sizeof(cf)));
        for (int k = 0; k < 2; k++) {
            HIPCHK(hipMalloc((void **)&c->pwr_pool[k], S * F * sizeof(float)));
            HIPCHK(hipMalloc((void **)&c->audio_pool[k], S * F * (n / 2) * sizeof(float)));
            HIPCHK(hipMalloc((void **)&c->nan_pool[k], S * F * sizeof(int)));
            HIPCHK(hipMemset(c->audio_pool[k], 0, S * F * (n / 2) * sizeof(float)));
            HIPCHK(hipMemset(c->pwr_pool[k], 0, S * F * sizeof(float)));
            HIPCHK(hipMemset(c->nan_pool[k], 0, S * F * sizeof(int)));
        }
        c->d_pwr = c->pwr_pool[0], c->d_audio = c->audio_pool[0], c->d_nan = c->nan_pool[0];
        HIPCHK(hipMalloc((void **)&c->d_real_prev, 2 * S * (n / 2) * sizeof(float)));
        HIPCHK(hipMalloc((void **)&c->d_bb_tail, 2 * S * (n / 2) * sizeof(cf)));
        HIPCHK(hipMalloc((void **)&c->d_bb_last, 2 * S * sizeof(cf)));
        HIPCHK(hipMemset(c->d_real_prev, 0, 2 * S * (n / 2) * sizeof(float)));
        HIPCHK(hipMemset(c->d_bb_tail, 0, 2 * S * (n / 2) * sizeof(cf)));
        HIPCHK(hipMemset(c->d_bb_last, 0, 2 * S * sizeof(cf)));
        HIPCHK(hipMalloc((void **)&c->d_ssb_mark, S * sizeof(unsigned)));
        HIPCHK(hipMemset(c->d_ssb_mark, 0, S * sizeof(unsigned)));
        if (c->lds_mode == 2) HIPCHK(hipMalloc((void **)&c->d_gscratch, S * F * 2 * n * sizeof(cf)));
        if (c->client_ring.init(S * (sizeof(ClientParams) + sizeof(int))))  // the batch's client list + the slot -> list index table
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
    // 2^22-point real frames (a 2^21-point packed transform): 1024 x 2048 - the first pass is then the 16-column kernel
    // of the 2^20 / 2^21-point shapes (128-byte raw rows and Y rows: 5.3 TB/s against the 8-column 2048-point kernel's
    // 4.1), the second pass walks tiles of FOUR (row, mirror row) couples of 2048-point rows (k_fft_pass2_real<2048, 8, 16>).
    // PSDR_REAL_SPLIT=2048x1024 selects round 4's split.
    if (is_real && m == 21) {
        const char *e = getenv("PSDR_REAL_SPLIT");
        if (!(e && strcmp(e, "2048x1024") == 0)) c->log2M2 = 11;
    }
    if (const char *e = psdr_tuning_env("PSDR_LOG2M2")) c->log2M2 = atoi(e);  // tuning: split M = M1 * M2
    c->log2M1 = m - c->log2M2;
    c->M1 = 1 << c->log2M1;
    c->M2 = 1 << c->log2M2;
    c->T1 = pick_T(c->M1, c->M2);
    c->T2 = pick_T(c->M2, c->M1);
    c->size_log2 = (int)std::lround(std::log2((double)N)) + cfg->brightness_offset;
    c->levels = cfg->downsample_levels;
    c->max_batch = cfg->max_batch;
    c->spec_stride = is_real ? (M + 2) : N;
    c->q_len = 0;
    for (int i = 0; i < c->levels; i++) c->q_len += c->R >> i;
    c->q_stride = (c->q_len + 127) & ~(size_t)127;
    c->real_fused = is_real && ((c->M2 == 1024 && c->T2 == 16 && (c->M1 == 1024 || c->M1 == 2048)) || (c->M2 == 2048 && c->T2 == 8 && c->M1 == 1024 && c->T1 == 16)) &&
                    getenv("PSDR_REAL_3PASS") == nullptr;
    if (c->real_fused) {
        const int cp = c->T2 / 2;  // (row, mirror row) couples per tile: 8 (octet records, levels 0..3) or 4 (quartets, levels 0..2)
        c->tile_ch = cp;           // quantize.h, RecMap mode 2
        c->LT = ilog2((size_t)cp);
        c->tiled_lt = c->LT;
        c->recmap.l2tpr = ilog2((size_t)(c->M1 / cp));
        c->recmap.l2gpt = 0;
        c->recmap.l2rows = c->log2M2;
        c->recmap.mapped = 2;
        c->recmap.pair = cp == 4 ? 1 : 0;  // quartets: a column's two records side by side (quantize.h)
        c->qt_stride = 2 * c->R;
        c->lay.mode = 2;
        c->lay.m1 = c->M1;
        c->lay.l2m1 = c->log2M1;
        c->lay.L = c->M2;
        c->lay.l2L = c->log2M2;
        c->lay.l2cp = c->LT;
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

int psdr::drain(psdr_ctx *c) {
    if (c->ring.copy) HIPCHK(hipStreamSynchronize(c->ring.copy));
    if (c->p1 != c->stream) HIPCHK(hipStreamSynchronize(c->p1));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->side != c->stream) HIPCHK(hipStreamSynchronize(c->side));
    for (hipStream_t st : c->pc_s)
        if (st) HIPCHK(hipStreamSynchronize(st));
    if (c->fetch_stream) HIPCHK(hipStreamSynchronize(c->fetch_stream));
    if (c->fetch_stream_pcm) HIPCHK(hipStreamSynchronize(c->fetch_stream_pcm));
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
    if (c->chain_pending && c->chain_seq > 0) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_pc[3][(c->chain_seq - 1) % psdr_ctx::PC_SETS], 0));
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
// (not in psdr.h: tools/seg_fallbacks.py and the hand-off test) segments of the LAST fused real-input launch, and how many
// of them did not find their carried row in memory in time and fell back to a seam (fft_pass.h: hand-off); 0 / 0 when the
// launch used uniform segments
extern "C" int psdr_debug_seg_fallbacks(psdr_ctx *c, unsigned *nsegs, unsigned *fallbacks, unsigned char *which, unsigned cap) {
    if (!c || !nsegs || !fallbacks) return fail(PSDR_ERR_INVALID, "null argument");
    *nsegs = *fallbacks = 0;
    if (!c->real_fused || !c->d_segflag || c->last_nframes <= 0) return PSDR_OK;
    const psdr_ctx::SegPlan *plan = nullptr;
    for (const auto &sp : c->seg_plans)
        if (sp.nframes == c->last_nframes) plan = &sp;
    if (!plan || !plan->handoff) return PSDR_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<unsigned> m(plan->nsegs);
    HIPCHK(hipMemcpy(m.data(), c->d_segflag + (size_t)(1 + c->cur_set) * c->seg_cap, m.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    *nsegs = plan->nsegs;
    for (unsigned i = 0; i < plan->nsegs; i++) {
        const bool fb = m[i] == c->seg_epoch;
        *fallbacks += fb;
        if (which && i < cap) which[i] = fb;  // (level-major: segment i is level i / nframes of frame i % nframes)
    }
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
        c->p1 = c->own_stream;
    }
    return PSDR_OK;
}
