// demod.hip - the audio clients (AudioClient, src/signal.h:53-123) and their batched demodulation: signal_loop +
// AudioClient::send_audio up to the NaN guard (src/websocket.cpp:156-185, src/signal.cpp:102-275) for every client and
// every frame of a batch, and the read-back of its results.
#include "ctx.h"
#include "demod.h"

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
            s.born = ++c->slot_births;
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
    // behind the list, for the post chain: the list index of every slot's client (its kernels walk the SLOTS, lane = slot & 63)
    const size_t S = c->aslots.size();
    int *h_slot_ci = (int *)(h_clients + S), *d_slot_ci = (int *)(d_clients + S);
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
        if (c->post_on) {
            for (size_t i = 0; i < S; i++) h_slot_ci[i] = -1;
            for (int i = 0; i < nact; i++) h_slot_ci[h_clients[i].slot] = i;
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
                h_slot_ci[i] = nact + npaused - 1;
            }
    }
    c->last_demod_frames = nframes;
    if (nact == 0) return PSDR_OK;
    {  // this batch's results go to the OTHER set (the copies of the last batch to the host may still be reading theirs); what
       // read this set two batches ago must have landed
        c->out_set ^= 1;
        c->d_audio = c->audio_pool[c->out_set], c->d_pwr = c->pwr_pool[c->out_set], c->d_nan = c->nan_pool[c->out_set];
        int rc = fetch_guard_wait(c, c->side, c->guard_audio[c->out_set]);
        if (rc) return rc;
        c->guard_audio[c->out_set] = nullptr;
        // ... and the post chain's moving averages of that batch, which read its audio rows themselves (postchain.h k_pc_ma2
        // DIRECT) on a stream of their own, up to two steps behind the passes (psdr_set_post_chain drains: the chain has
        // been on for every batch since chain_seq started to count)
        if (c->post_on && c->post_direct && c->chain_seq >= 2 && c->pc_s[0] && c->side != c->stream)
            HIPCHK(hipStreamWaitEvent(c->side, c->ev_pc[1][(c->chain_seq - 2) % psdr_ctx::PC_SETS], 0));
    }
    HIPCHK(hipMemcpyAsync(d_clients, h_clients, c->post_on ? S * (sizeof(ClientParams) + sizeof(int)) : (size_t)nact * sizeof(ClientParams),
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
    a.ssb_mark = c->d_ssb_mark;
    a.mark_epoch = (unsigned)(c->demod_seq % 0xFFFFFFFFull) + 1u;  // never 0
    a.replay = 0;
    // the frame-ordered second walk of marked USB / LSB slots (demod.h: DemodArgs::ssb_mark) can only find work where
    // non-finite values can arise: float input formats, or a spectrum that comes from the caller
    const bool can_be_nonfinite = c->cfg.input_format >= PSDR_FMT_F32 || spec != c->d_spec;
    bool ola_done = false;
    {
        ProfScope ps(c, K_IDFT, c->side);
        const bool fixed_plan = c->n == 360 || c->n == 720;
        if (fixed_plan && c->demod_chain) {
            // transform + overlap-add + demodulation in one kernel, one wave per chain of K consecutive frames of a
            // client (demod.h): long chains repeat fewer transforms (1 or 2 per chain), short ones give few clients
            // enough waves
            // (256 clients x 256 frames, same box: K = 4 / 8 / 16 / 32 -> 5.81 / 5.77 / 5.93 / 6.04 us per frame, the
            // two-kernel path 5.99)
            // (round 4, 512-frame launches: with 256 clients and more, chains of 16 still leave 8192 waves and repeat half as
            // many warm-up transforms: 93.4 -> 94.6 GS/s on the 256-client shape, same box, interleaved twice)
            int K = c->demod_chain_k > 0 ? c->demod_chain_k : (nact >= 256 && (unsigned)nact * (unsigned)((nframes + 15) / 16) >= 8192u ? 16 : 8);
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
            if (can_be_nonfinite) {
                DemodArgs ar = a;
                ar.replay = 1;
                const int KF = nframes;  // one chain = the whole batch
                if (c->n == 360)
                    hipLaunchKernelGGL((k_demod_chain_fixed<360, 8, 9, 5>), dim3(((unsigned)nact + W - 1) / W), dim3(64 * W), lds,
                                       c->side, ar, nact, KF);
                else
                    hipLaunchKernelGGL((k_demod_chain_fixed<720, 8, 9, 10>), dim3(((unsigned)nact + W - 1) / W), dim3(64 * W), lds,
                                       c->side, ar, nact, KF);
            }
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
        } else if (c->n <= 512) {
            // one wave per (client, frame), no work-group barriers (demod.h)
            const unsigned items = (unsigned)nact * (unsigned)nframes;
            const size_t lds = (size_t)(2 * PSDR_IDFT_WAVES + 1) * c->n * sizeof(cf);
            hipLaunchKernelGGL(k_demod_idft_wave, dim3((items + PSDR_IDFT_WAVES - 1) / PSDR_IDFT_WAVES),
                               dim3(64 * PSDR_IDFT_WAVES), lds, c->side, a, nact);
        } else {
            if (c->idft_lds > 64 * 1024 && c->lds_attr_done.insert((const void *)k_demod_idft).second)
                HIPCHK(hipFuncSetAttribute((const void *)k_demod_idft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->idft_lds));
            hipLaunchKernelGGL(k_demod_idft, dim3(nact, nframes), dim3(c->idft_threads), c->idft_lds, c->side,
                               a);
        }
        HIPCHK(hipGetLastError());
    }
    if (!ola_done) {
        ProfScope ps(c, K_OLA, c->side);
        const unsigned items = (unsigned)nact * (unsigned)((nframes + PSDR_OLA_FG - 1) / PSDR_OLA_FG);
        hipLaunchKernelGGL(k_demod_ola, dim3((items + 3) / 4), dim3(256), 0, c->side, a, nact);
        if (can_be_nonfinite) hipLaunchKernelGGL(k_demod_ola_seq, dim3(((unsigned)nact + 3) / 4), dim3(256), 0, c->side, a, nact);
        HIPCHK(hipGetLastError());
    }
    hipStream_t last_user = c->side;
    if (c->post_on && nact > 0) {
        int rc = post_chain_enqueue(c, d_clients, d_slot_ci, nact, npaused, nframes, &last_user);
        if (rc) return rc;
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

// ---- batched read-back: the served end of the path --------------------------------------------------------------
// (src/websocket.cpp:156-185 makes one pass over signal_slices per frame and every send_audio ends in host memory:
// src/signal.cpp:283-291 -> src/audio.cpp:26-44; send_waterfall: src/waterfall.cpp:44-51.  Per-client psdr_read_audio
// would pay a synchronisation and three copies per client and frame.)
// psdr_fetch_begin ENQUEUES the copies of the last demodulation batch (and of the last waterfall batch) into one of a ring
// of PSDR_FETCH_SETS pinned host sets on a copy stream, behind the kernels that produce them, and returns; psdr_fetch_end waits for the
// oldest fetch in flight and makes its set the one psdr_fetched_* read.  Between the two the caller enqueues the NEXT batch:
// the copies run beside its FFT passes.  The device-side result buffers exist once: the next batch's demodulation,
// waterfall gather and PCM output wait (in stream order, no host wait) for the newest fetch's copies.
int psdr::fetch_guard_wait(psdr_ctx *c, hipStream_t st, hipEvent_t ev) {
    (void)c;
    if (ev) HIPCHK(hipStreamWaitEvent(st, ev, 0));
    return PSDR_OK;
}
extern "C" int psdr_fetch_begin(psdr_ctx *c, unsigned what) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (what == 0 || (what & ~(PSDR_FETCH_AUDIO | PSDR_FETCH_PCM | PSDR_FETCH_WATERFALL))) return fail(PSDR_ERR_INVALID, "PSDR_FETCH_* bits 0x%x", what);
    const bool want_audio = (what & (PSDR_FETCH_AUDIO | PSDR_FETCH_PCM)) != 0;
    if (want_audio && c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    const size_t F = (size_t)c->last_demod_frames, h = (size_t)c->n / 2, mb = (size_t)c->max_batch, S = c->aslots.size();
    if (want_audio && (F == 0 || c->demod_seq == 0)) return fail(PSDR_ERR_STATE, "no demodulated batch to fetch");
    if ((what & PSDR_FETCH_PCM) && !c->post_on) return fail(PSDR_ERR_STATE, "PSDR_FETCH_PCM: post chain not enabled (psdr_set_post_chain)");
    HIPCHK(hipSetDevice(c->device));
    if (!c->fetch_stream) HIPCHK(hipStreamCreateWithFlags(&c->fetch_stream, hipStreamNonBlocking));
    if ((what & PSDR_FETCH_PCM) && !c->fetch_stream_pcm) HIPCHK(hipStreamCreateWithFlags(&c->fetch_stream_pcm, hipStreamNonBlocking));
    if (!c->ev_fetch_src) HIPCHK(hipEventCreateWithFlags(&c->ev_fetch_src, hipEventDisableTiming));
    psdr_ctx::FetchSet &fs = c->fset[c->fetch_fill];
    if (fs.inflight) {  // every set in flight: the oldest one has to land first (its results are given up: psdr_fetch_end was not called)
        HIPCHK(hipEventSynchronize(fs.done));
        if (fs.has_pcm) HIPCHK(hipEventSynchronize(fs.ev_pcm));
        fs.inflight = false;
        c->fetch_inflight--;
    }
    if (c->fetch_cur == c->fetch_fill) c->fetch_cur = -1;  // its pointers die now
    // (each block on its own: a failed allocation leaves nothing half-initialised behind for the next call)
    if (!fs.done) HIPCHK(hipEventCreateWithFlags(&fs.done, hipEventDisableTiming));
    if (!fs.ev_wf) HIPCHK(hipEventCreateWithFlags(&fs.ev_wf, hipEventDisableTiming));
    if (!fs.ev_audio) HIPCHK(hipEventCreateWithFlags(&fs.ev_audio, hipEventDisableTiming));
    if (!fs.ev_pcm) HIPCHK(hipEventCreateWithFlags(&fs.ev_pcm, hipEventDisableTiming));
    if (want_audio && !fs.pwr) HIPCHK(hipHostMalloc((void **)&fs.pwr, S * mb * sizeof(float), hipHostMallocDefault));
    if (want_audio && !fs.nan) HIPCHK(hipHostMalloc((void **)&fs.nan, S * mb * sizeof(int32_t), hipHostMallocDefault));
    if ((what & PSDR_FETCH_AUDIO) && !fs.audio) HIPCHK(hipHostMalloc((void **)&fs.audio, S * mb * h * sizeof(float), hipHostMallocDefault));
    if ((what & PSDR_FETCH_PCM) && !fs.pcm) HIPCHK(hipHostMalloc((void **)&fs.pcm, S * mb * h * sizeof(int32_t), hipHostMallocDefault));
    size_t wf_bytes = 0;
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        fs.win.resize(S);
        for (size_t i = 0; i < S; i++) {
            const AudioSlot &sl = c->aslots[i];
            fs.win[i].last_seq = sl.active ? sl.last_seq : 0;
            fs.win[i].born = sl.born;
            fs.win[i].l = sl.b_l, fs.win[i].r = sl.b_r, fs.win[i].mid = sl.b_mid;
        }
        fs.wfm.assign(c->wslots.begin(), c->wslots.end());
        if (what & PSDR_FETCH_WATERFALL)
            for (const WfSlot &w : fs.wfm)
                if (w.active && w.nsent > 0) wf_bytes = std::max(wf_bytes, (w.out_off + (size_t)w.nsent * (size_t)(w.b_r - w.b_l) + 15) & ~(size_t)15);
    }
    if (wf_bytes > fs.wf_cap) {
        if (fs.wf) HIPCHK(hipHostFree(fs.wf));
        fs.wf = nullptr, fs.wf_cap = 0;
        HIPCHK(hipHostMalloc((void **)&fs.wf, wf_bytes, hipHostMallocDefault));
        fs.wf_cap = wf_bytes;
    }
    hipStream_t fst = c->fetch_stream;
    // rows [slot][0..F) of a device array [slot][max_batch][row_bytes]: ONE plain copy when the batch fills max_batch (the
    // DMA engines' case; a pitched copy may be done by a copy kernel on the CUs the passes are using), else one strided copy
    auto rows_d2h = [&](void *dst, const void *src, size_t row_bytes) -> hipError_t {
        if (F == mb) return hipMemcpyAsync(dst, src, S * mb * row_bytes, hipMemcpyDeviceToHost, fst);
        return hipMemcpy2DAsync(dst, mb * row_bytes, src, mb * row_bytes, F * row_bytes, S, hipMemcpyDeviceToHost, fst);
    };
    // behind the demodulation and the waterfall gather of the last batch (both on `side`) ...
    HIPCHK(hipEventRecord(c->ev_fetch_src, c->side));
    HIPCHK(hipStreamWaitEvent(fst, c->ev_fetch_src, 0));
    // the waterfall rows first (small; their device buffer exists once: the next gather waits for ev_wf alone)
    if (wf_bytes) {
        HIPCHK(hipMemcpyAsync(fs.wf, c->d_wfout, wf_bytes, hipMemcpyDeviceToHost, fst));
        HIPCHK(hipEventRecord(fs.ev_wf, fst));
        c->guard_wf = fs.ev_wf;
    }
    // rows [slot][0..F) of the device arrays [slot][max_batch][...]: one strided copy each
    if (want_audio) {
        HIPCHK(rows_d2h(fs.pwr, c->d_pwr, sizeof(float)));
        HIPCHK(rows_d2h(fs.nan, c->d_nan, sizeof(int32_t)));
        if (what & PSDR_FETCH_AUDIO) HIPCHK(rows_d2h(fs.audio, c->d_audio, h * sizeof(float)));
        HIPCHK(hipEventRecord(fs.ev_audio, fst));
        c->guard_audio[c->out_set] = fs.ev_audio;
    }
    HIPCHK(hipEventRecord(fs.done, fst));
    fs.has_pcm = false;
    if (what & PSDR_FETCH_PCM) {
        // ... the PCM behind the chain's output kernel of that batch (up to two steps after the passes) on a copy stream of
        // ITS OWN: on the one stream the next batch's waterfall rows and audio would queue behind a copy that waits for
        // the chain - and the gather / demodulation that wait for THOSE copies with them (256 clients: +33 % on the step)
        hipStream_t fsp = c->fetch_stream_pcm;
        HIPCHK(hipStreamWaitEvent(fsp, c->ev_fetch_src, 0));
        if (c->chain_seq > 0 && c->pc_s[0] && c->side != c->stream)
            HIPCHK(hipStreamWaitEvent(fsp, c->ev_pc[3][(c->chain_seq - 1) % psdr_ctx::PC_SETS], 0));
        // (PSDR_OPT_POST_CHAIN_PCM16: the rows are int16 - the same buffers, half the bytes)
        const size_t sb = c->pcm_is16 ? sizeof(int16_t) : sizeof(int32_t);
        fs.pcm16 = c->pcm_is16;
        if (F == mb)
            HIPCHK(hipMemcpyAsync(fs.pcm, c->post.pcm, S * mb * h * sb, hipMemcpyDeviceToHost, fsp));
        else
            HIPCHK(hipMemcpy2DAsync(fs.pcm, mb * h * sb, c->post.pcm, mb * h * sb, F * h * sb, S, hipMemcpyDeviceToHost, fsp));
        HIPCHK(hipEventRecord(fs.ev_pcm, fsp));
        c->guard_pcm[c->pcm_set] = fs.ev_pcm;
        fs.has_pcm = true;
    }
    fs.inflight = true;
    fs.what = what;
    fs.frames = want_audio ? (int)F : 0;
    fs.seq = want_audio ? c->demod_seq : 0;
    c->fetch_fill = (c->fetch_fill + 1) % PSDR_FETCH_SETS;
    c->fetch_inflight++;
    return PSDR_OK;
}
extern "C" int psdr_fetch_end(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    // the oldest fetch in flight: the set that would be filled next if it is in flight, else the other one
    if (c->fetch_inflight <= 0) return fail(PSDR_ERR_STATE, "psdr_fetch_end without a psdr_fetch_begin in flight");
    const int k = (c->fetch_fill - c->fetch_inflight + 2 * PSDR_FETCH_SETS) % PSDR_FETCH_SETS;
    psdr_ctx::FetchSet &fs = c->fset[k];
    if (!fs.inflight) return fail(PSDR_ERR_STATE, "psdr_fetch_end: the fetch ring is out of step");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(fs.done));
    if (fs.has_pcm) HIPCHK(hipEventSynchronize(fs.ev_pcm));
    fs.inflight = false;
    c->fetch_inflight--;
    // (everything of this fetch has landed: nothing left for a writer to wait for)
    if (c->guard_wf == fs.ev_wf) c->guard_wf = nullptr;
    for (int i = 0; i < 2; i++) {
        if (c->guard_audio[i] == fs.ev_audio) c->guard_audio[i] = nullptr;
        if (c->guard_pcm[i] == fs.ev_pcm) c->guard_pcm[i] = nullptr;
    }
    c->fetch_cur = k;
    // one-launch transforms: a flow-control timeout of the batches since the last synchronisation is reported by drain();
    // a fetch does not drain (that is its point) - psdr_synchronize still does
    return PSDR_OK;
}
extern "C" int psdr_fetch_batch(psdr_ctx *c) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->n <= 0) return fail(PSDR_ERR_STATE, "context created with audio_fft_size 0");
    if (c->last_demod_frames == 0 || c->demod_seq == 0) return fail(PSDR_ERR_STATE, "no demodulated batch to fetch");
    // the synchronous form: everything there is, and the device drained (errors of the batch surface here)
    while (c->fetch_inflight > 0) {
        int rc = psdr_fetch_end(c);
        if (rc) return rc;
    }
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    int rc = psdr_fetch_begin(c, PSDR_FETCH_AUDIO | (c->post_on ? PSDR_FETCH_PCM : 0u) | PSDR_FETCH_WATERFALL);
    if (rc) return rc;
    return psdr_fetch_end(c);
}
static int fetched_set(psdr_ctx *c, const psdr_ctx::FetchSet **out) {
    if (c->fetch_cur < 0) return fail(PSDR_ERR_STATE, "psdr_fetch_batch() / psdr_fetch_end() first");
    *out = &c->fset[c->fetch_cur];
    return PSDR_OK;
}
extern "C" int psdr_fetched_window(psdr_ctx *c, int id, int *l, double *audio_mid, int *r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_slot(c, id);
    if (rc) return rc;
    const psdr_ctx::FetchSet *fs = nullptr;
    if ((rc = fetched_set(c, &fs))) return rc;
    if (fs->seq == 0) return fail(PSDR_ERR_STATE, "the fetched batch carries no audio (PSDR_FETCH_AUDIO / _PCM)");
    if ((size_t)id >= fs->win.size() || fs->win[id].last_seq != fs->seq || fs->win[id].born != c->aslots[id].born)
        return fail(PSDR_ERR_NO_DATA, "client %d was not part of the fetched batch", id);
    if (l) *l = fs->win[id].l;
    if (audio_mid) *audio_mid = fs->win[id].mid;
    if (r) *r = fs->win[id].r;
    return PSDR_OK;
}
extern "C" int psdr_fetched_audio(psdr_ctx *c, int id, int frame, const float **audio, float *pwr, int32_t *nan_flag,
                                  const int32_t **pcm) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    const psdr_ctx::FetchSet *fs = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
        if ((rc = fetched_set(c, &fs))) return rc;
        if (fs->seq == 0) return fail(PSDR_ERR_STATE, "the fetched batch carries no audio (PSDR_FETCH_AUDIO / _PCM)");
        // (a slot handed to a new client since the batch was demodulated holds the previous occupant's rows: not this client's)
        if ((size_t)id >= fs->win.size() || fs->win[id].last_seq != fs->seq || fs->win[id].born != c->aslots[id].born)
            return fail(PSDR_ERR_NO_DATA, "client %d was not part of the fetched batch", id);
    }
    if (frame < 0 || frame >= fs->frames) return fail(PSDR_ERR_INVALID, "frame %d not in the fetched batch of %d", frame, fs->frames);
    const size_t h = (size_t)c->n / 2, mb = (size_t)c->max_batch, row = (size_t)id * mb + (size_t)frame;
    if (audio) *audio = (fs->what & PSDR_FETCH_AUDIO) ? fs->audio + row * h : nullptr;
    if (pwr) *pwr = fs->pwr[row];
    if (nan_flag) *nan_flag = fs->nan[row];
    if (pcm) *pcm = ((fs->what & PSDR_FETCH_PCM) && !fs->pcm16) ? fs->pcm + row * h : nullptr;  // (int16 rows: psdr_fetched_pcm16)
    return PSDR_OK;
}
extern "C" int psdr_fetched_pcm16(psdr_ctx *c, int id, int frame, const int16_t **pcm) {
    if (!c || !pcm) return fail(PSDR_ERR_INVALID, "null argument");
    const psdr_ctx::FetchSet *fs = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int rc = check_slot(c, id);
        if (rc) return rc;
        if ((rc = fetched_set(c, &fs))) return rc;
        if (fs->seq == 0 || !(fs->what & PSDR_FETCH_PCM)) return fail(PSDR_ERR_STATE, "the fetched batch carries no PCM (PSDR_FETCH_PCM)");
        if (!fs->pcm16) return fail(PSDR_ERR_STATE, "the fetched PCM rows are int32 (PSDR_OPT_POST_CHAIN_PCM16 was 0 for that batch): psdr_fetched_audio");
        if ((size_t)id >= fs->win.size() || fs->win[id].last_seq != fs->seq || fs->win[id].born != c->aslots[id].born)
            return fail(PSDR_ERR_NO_DATA, "client %d was not part of the fetched batch", id);
    }
    if (frame < 0 || frame >= fs->frames) return fail(PSDR_ERR_INVALID, "frame %d not in the fetched batch of %d", frame, fs->frames);
    const size_t h = (size_t)c->n / 2, mb = (size_t)c->max_batch, row = (size_t)id * mb + (size_t)frame;
    *pcm = reinterpret_cast<const int16_t *>(fs->pcm) + row * h;
    return PSDR_OK;
}
extern "C" int psdr_fetched_waterfall(psdr_ctx *c, int id, const int8_t **rows, int *nsent_out, int *level_out, int *l_out, int *r_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    const psdr_ctx::FetchSet *fs = nullptr;
    int rc = fetched_set(c, &fs);
    if (rc) return rc;
    if (!(fs->what & PSDR_FETCH_WATERFALL)) return fail(PSDR_ERR_STATE, "the fetched batch carries no waterfall rows (PSDR_FETCH_WATERFALL)");
    if (id < 0 || (size_t)id >= fs->wfm.size() || !fs->wfm[id].active) return fail(PSDR_ERR_NO_DATA, "waterfall client %d was not part of the fetched batch", id);
    const WfSlot &w = fs->wfm[id];
    if (rows) *rows = w.nsent > 0 ? fs->wf + w.out_off : nullptr;
    if (nsent_out) *nsent_out = w.nsent;
    if (level_out) *level_out = w.b_level;
    if (l_out) *l_out = w.b_l;
    if (r_out) *r_out = w.b_r;
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
    if (c->pcm_is16) {  // PSDR_OPT_POST_CHAIN_PCM16: the rows are int16 on the device; this call still delivers the reference's int32 buffer
        std::vector<int16_t> tmp(F * h);
        HIPCHK(hipMemcpy(tmp.data(), reinterpret_cast<const int16_t *>(c->post.pcm) + (size_t)id * mb * h, F * h * sizeof(int16_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < F * h; i++) pcm[i] = tmp[i];
    } else {
        HIPCHK(hipMemcpy(pcm, c->post.pcm + (size_t)id * mb * h, F * h * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
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
extern "C" int psdr_demod_batch_from_band(psdr_ctx *c, const float *d_band, size_t frame_stride_bins, uint32_t first_bin,
                                          uint32_t nbins, int nframes, uint64_t first_frame_num) {
    if (!c || !d_band) return fail(PSDR_ERR_INVALID, "null argument");
    if (nframes < 1 || nframes > c->max_batch)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c->max_batch);
    if (nbins < 1 || frame_stride_bins < nbins) return fail(PSDR_ERR_INVALID, "frame stride smaller than the band");
    const uint32_t band[2] = {first_bin, nbins};
    return demod_impl(c, (const cf *)d_band, frame_stride_bins, nframes, first_frame_num, band);
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
