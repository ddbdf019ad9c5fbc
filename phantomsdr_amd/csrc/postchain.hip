// postchain.hip - the post-demodulation chain of AudioClient::send_audio (src/signal.cpp:277-284) for all clients of a
// batch: DC blocker, AGC, int16 conversion (postchain.h) as a two-stage pipeline across batches.
#include "ctx.h"
#include "postchain.h"

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
            if (const char *e = psdr_tuning_env("PSDR_PC_ABL"))
                if (atoi(e) & 16) hi = 0;  // (tuning build: the chain's two streams at normal priority)
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

// the chain's kernels for the batch whose demodulation has just been enqueued on c->side (d_clients: its parameter
// block, the nact active clients first, then npaused paused ones with an empty stream)
int psdr::post_chain_enqueue(psdr_ctx *c, const ClientParams *d_clients, int nact, int npaused, int nframes, hipStream_t *last_user) {
    const int par = (int)(c->chain_seq & 1);
    int abl = 0;  // (tuning build only: which part of the chain costs the step what)
    if (const char *e = psdr_tuning_env("PSDR_PC_ABL")) abl = atoi(e);
    const bool piped = c->side != c->stream && c->side2 != nullptr && !(abl & 8);
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
        if (abl & 1) {
        } else if (pa.ma_fused) {
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
        if (!(abl & 4)) {
            hipLaunchKernelGGL(k_pc_scan, dim3(nall, nblk, 2), dim3(64), 0, s1, pa);
            hipLaunchKernelGGL(k_pc_want, dim3(nall, (unsigned)((Tb + 255) / 256)), dim3(256), 0, s1, pa);
        }
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
        if (abl & 2) {
        } else if (pa.attack >= pa.release)
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
    *last_user = s2;
    return PSDR_OK;
}
