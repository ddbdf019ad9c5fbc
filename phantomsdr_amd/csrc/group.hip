// group.hip - SURVEY 8e from C: ONE process drives n GPUs of a node.  GPU 0 ("root") owns the raw ring, the forward
// FFT and the waterfall clients; the audio clients are spread over all n contexts; one exchange per batch over xGMI,
// issued DIRECTLY through RCCL (librccl.so is dlopen()ed when a group of more than one device is created - a single-GPU
// user of libpsdr_hip.so never loads it):
//   PSDR_SHARD_CLIENTS  ncclBroadcast of the spectrum, F x 8 (N + A) bytes per batch (BASELINE.json configs[3]: the
//                       north star's shape; link-bound at ~9.5 GS/s by construction)
//   PSDR_SHARD_RAW      ncclBroadcast of the raw half-frames (4 x fewer bytes for cs16), every GPU runs the forward FFT
//   PSDR_SHARD_BAND     GPU b gets only band b of the spectrum (+ a halo of one maximal window): ncclSend / ncclRecv of
//                       R/n + halo bins per frame.  2^20- and 2^21-point IQ contexts: the root's second pass writes the
//                       band regions itself (psdr_set_band_layout) and the regions ARE the send buffers; otherwise
//                       psdr_pack_band fills them.
// Everything of one rank - its kernels and its side of the collective - is enqueued in order on ONE stream per device
// (psdr_set_stream), so a step needs no host synchronisation; psdr_group_synchronize() drains all devices.
// The Python twin for one PROCESS per GPU (torch.distributed, what bench.py --gpus N runs) is phantomsdr_amd/distributed.py.
#include <dlfcn.h>

#include "ctx.h"

namespace {

// the slice of RCCL's C API used here (rccl.h: ncclCommInitAll :236, ncclBroadcast :591, ncclSend :700, ncclRecv :722)
typedef void *ncclComm_t;
enum { ncclSuccess = 0 };
enum { ncclChar = 0, ncclFloat = 7 };
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int load() {
        if (h) return PSDR_OK;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return fail(PSDR_ERR_UNSUPPORTED, "librccl.so could not be loaded (%s): a multi-GPU group needs RCCL", dlerror());
#define SYM(f)                                                                       \
    *(void **)(&f) = dlsym(h, "nccl" #f);                                            \
    if (!f) return fail(PSDR_ERR_UNSUPPORTED, "librccl.so does not export nccl" #f)
        SYM(CommInitAll);
        SYM(CommDestroy);
        SYM(GroupStart);
        SYM(GroupEnd);
        SYM(Broadcast);
        SYM(Send);
        SYM(Recv);
        SYM(GetErrorString);
#undef SYM
        return PSDR_OK;
    }
};
Rccl g_rccl;

#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        int e_ = (expr);                                                                                       \
        if (e_ != ncclSuccess) return fail(PSDR_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(e_));  \
    } while (0)

}  // namespace

struct psdr_group {
    int n = 0, shard = 0;
    bool comm_on = false;           // collectives are issued (n > 1, or PSDR_SHARD_FORCE_COMM)
    std::vector<int> dev;
    std::vector<psdr_ctx *> ctx;
    std::vector<hipStream_t> st;    // the one stream per device everything of that rank is ordered on
    std::vector<ncclComm_t> comm;
    std::vector<void *> rbuf;       // per rank: receive buffer (raw halves / band region), nullptr where unused
    size_t rbuf_bytes = 0;
    std::vector<void *> sbuf;       // band sharding with the pack pass: the root's send buffers, one per band
    // band sharding
    bool banded = false;            // the root writes band regions itself (no pack)
    uint32_t band_first[16] = {0}, band_bins[16] = {0};
    size_t band_stride = 0;         // bins between frames inside a band buffer
    int next_rr = 0;                // round-robin cursor of psdr_group_client_add
    // link accounting
    double link_bytes = 0;          // bytes that crossed ONE link (root -> one peer) in the last step
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint64_t steps = 0;
};

static int gid_split(psdr_group *g, int gid, int *rank, int *id) {
    *rank = gid >> 16;
    *id = gid & 0xFFFF;
    if (gid < 0 || *rank >= g->n) return fail(PSDR_ERR_INVALID, "no client %d in this group", gid);
    return PSDR_OK;
}
static int band_of(const psdr_group *g, int l) {
    const size_t R = g->ctx[0]->R, per = R / (size_t)g->n;
    int b = (int)((size_t)std::max(l, 0) / per);
    return std::min(b, g->n - 1);
}

extern "C" void psdr_group_destroy(psdr_group *g) {
    if (!g) return;
    for (int r = 0; r < (int)g->ctx.size(); r++) {
        if (r < (int)g->dev.size()) hipSetDevice(g->dev[r]);
        if (g->ctx[r]) {
            psdr_synchronize(g->ctx[r]);
            psdr_set_stream(g->ctx[r], nullptr);
        }
        if (r < (int)g->comm.size() && g->comm[r]) g_rccl.CommDestroy(g->comm[r]);
        if (r < (int)g->rbuf.size() && g->rbuf[r]) hipFree(g->rbuf[r]);
        if (r == 0)
            for (void *p : g->sbuf)
                if (p) hipFree(p);
        if (r == 0 && g->ev0) hipEventDestroy(g->ev0), hipEventDestroy(g->ev1);
        if (g->ctx[r]) psdr_destroy(g->ctx[r]);
        if (r < (int)g->st.size() && g->st[r]) hipStreamDestroy(g->st[r]);
    }
    delete g;
}

extern "C" int psdr_group_create(const psdr_config *cfg, const int *devices, int ndevices, int shard, psdr_group **out) {
    if (!cfg || !devices || !out) return fail(PSDR_ERR_INVALID, "null argument");
    const bool force_comm = (shard & PSDR_SHARD_FORCE_COMM) != 0;
    shard &= ~PSDR_SHARD_FORCE_COMM;
    if (ndevices < 1 || ndevices > 16) return fail(PSDR_ERR_INVALID, "a group has 1..16 devices, not %d", ndevices);
    if (shard < PSDR_SHARD_CLIENTS || shard > PSDR_SHARD_BAND) return fail(PSDR_ERR_INVALID, "unknown sharding %d", shard);
    for (int i = 0; i < ndevices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return fail(PSDR_ERR_INVALID, "device %d listed twice (RCCL wants one rank per device)", devices[i]);
    if (shard == PSDR_SHARD_BAND && (ndevices & (ndevices - 1))) return fail(PSDR_ERR_INVALID, "band sharding: a power-of-two number of devices, not %d", ndevices);
    psdr_group *g = new (std::nothrow) psdr_group();
    if (!g) return fail(PSDR_ERR_NOMEM, "out of memory");
    g->n = ndevices;
    g->shard = shard;
    g->comm_on = ndevices > 1 || force_comm;
    g->dev.assign(devices, devices + ndevices);
    g->ctx.assign(ndevices, nullptr);
    g->st.assign(ndevices, nullptr);
    g->comm.assign(ndevices, nullptr);
    g->rbuf.assign(ndevices, nullptr);
    auto bail = [&](int rc) {
        const std::string msg = psdr_last_error();  // (the clean-up below must not overwrite it)
        psdr_group_destroy(g);
        return fail(rc, "%s", msg.c_str());
    };
    for (int r = 0; r < ndevices; r++) {
        psdr_config c = *cfg;
        c.device = devices[r];
        int rc = psdr_create(&c, &g->ctx[r]);
        if (rc) return bail(rc);
        if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&g->st[r], hipStreamNonBlocking) != hipSuccess) {
            fail(PSDR_ERR_HIP, "stream creation on device %d failed", devices[r]);
            return bail(PSDR_ERR_HIP);
        }
        rc = psdr_set_stream(g->ctx[r], g->st[r]);
        if (rc) return bail(rc);
    }
    psdr_ctx *c0 = g->ctx[0];
    const size_t F = (size_t)c0->max_batch;
    if (g->comm_on) {
        int rc = g_rccl.load();
        if (rc) return bail(rc);
        const int e = g_rccl.CommInitAll(g->comm.data(), ndevices, devices);
        if (e != ncclSuccess) {
            fail(PSDR_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", ndevices, g_rccl.GetErrorString(e));
            return bail(PSDR_ERR_HIP);
        }
    }
    // ---- per-sharding buffers
    if (shard == PSDR_SHARD_RAW) {
        g->rbuf_bytes = (F + 1) * psdr_half_frame_bytes(c0);
        for (int r = 1; r < ndevices; r++) {
            if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&g->rbuf[r], g->rbuf_bytes) != hipSuccess) {
                fail(PSDR_ERR_NOMEM, "raw receive buffer of %zu bytes on device %d", g->rbuf_bytes, devices[r]);
                return bail(PSDR_ERR_NOMEM);
            }
        }
    } else if (shard == PSDR_SHARD_BAND) {
        const uint32_t halo = (uint32_t)std::max(cfg->audio_fft_size, 0);  // a window is at most audio_fft_size bins wide (src/signal.cpp:309-311)
        const size_t R = c0->R;
        g->banded = !c0->is_real && c0->lay.mode == 1 && ndevices > 1;
        if (g->banded) {
            int rc = psdr_set_band_layout(c0, ndevices, halo);
            if (rc) return bail(rc);
            for (int b = 0; b < ndevices; b++) {
                const float *p;
                rc = psdr_band_region(c0, b, &p, &g->band_stride, &g->band_first[b], &g->band_bins[b]);
                if (rc) return bail(rc);
            }
        } else {
            // linear band buffers filled by psdr_pack_band: [b R/n, (b+1) R/n + 1 + halo), the same count on every rank
            const size_t per = (R + (size_t)ndevices - 1) / (size_t)ndevices;
            const uint32_t cnt = (uint32_t)std::min(per + 1 + halo, R);
            g->band_stride = cnt;
            for (int b = 0; b < ndevices; b++) g->band_first[b] = (uint32_t)((size_t)b * (R / (size_t)ndevices)), g->band_bins[b] = cnt;
            g->sbuf.assign(ndevices, nullptr);
            for (int b = 1; b < ndevices; b++) {
                if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&g->sbuf[b], F * g->band_stride * sizeof(cf)) != hipSuccess) {
                    fail(PSDR_ERR_NOMEM, "band send buffer of %zu bytes", F * g->band_stride * sizeof(cf));
                    return bail(PSDR_ERR_NOMEM);
                }
            }
        }
        g->rbuf_bytes = F * g->band_stride * sizeof(cf);
        for (int r = 1; r < ndevices; r++) {
            if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&g->rbuf[r], g->rbuf_bytes) != hipSuccess) {
                fail(PSDR_ERR_NOMEM, "band receive buffer of %zu bytes on device %d", g->rbuf_bytes, devices[r]);
                return bail(PSDR_ERR_NOMEM);
            }
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) {
        fail(PSDR_ERR_HIP, "event creation failed");
        return bail(PSDR_ERR_HIP);
    }
    *out = g;
    return PSDR_OK;
}

extern "C" int psdr_group_size(const psdr_group *g) { return g ? g->n : 0; }
extern "C" psdr_ctx *psdr_group_ctx(psdr_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

// ---- audio clients: the group picks the GPU ---------------------------------------------------------------------
extern "C" int psdr_group_client_add(psdr_group *g, int l, double audio_mid, int r, int mode, int *gid_out) {
    if (!g || !gid_out) return fail(PSDR_ERR_INVALID, "null argument");
    // clients / raw: round robin (client i on GPU i mod n, like assign_clients of distributed.py); band: the band the
    // window STARTS in (it may end in the halo)
    const int rank = g->shard == PSDR_SHARD_BAND ? band_of(g, l) : g->next_rr % g->n;
    int id = -1;
    int rc = psdr_client_add(g->ctx[rank], &id);
    if (rc) return rc;
    rc = psdr_client_set_audio_demodulation(g->ctx[rank], id, mode);
    if (!rc) rc = psdr_client_set_audio_range(g->ctx[rank], id, l, audio_mid, r);
    if (rc) {
        const std::string msg = psdr_last_error();
        psdr_client_remove(g->ctx[rank], id);
        return fail(rc, "%s", msg.c_str());
    }
    if (g->shard != PSDR_SHARD_BAND) g->next_rr++;
    *gid_out = (rank << 16) | id;
    return PSDR_OK;
}
extern "C" int psdr_group_client_remove(psdr_group *g, int gid) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_client_remove(g->ctx[rank], id);
}
extern "C" int psdr_group_client_set_audio_demodulation(psdr_group *g, int gid, int mode) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_client_set_audio_demodulation(g->ctx[rank], id, mode);
}
extern "C" int psdr_group_client_set_paused(psdr_group *g, int gid, int paused) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_client_set_paused(g->ctx[rank], id, paused);
}
// A retune inside the client's GPU is AudioClient::set_audio_range.  Band sharding only: a window that now starts in
// another band moves the client to that band's GPU - a new slot there, *gid changes, and the overlap-add tail does not
// travel (one frame of audio starts from silence; the reference keeps the tail across a retune, src/signal.cpp:81-94).
extern "C" int psdr_group_client_set_audio_range(psdr_group *g, int *gid, int l, double audio_mid, int r) {
    if (!g || !gid) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, *gid, &rank, &id);
    if (rc) return rc;
    const int want = g->shard == PSDR_SHARD_BAND ? band_of(g, l) : rank;
    if (want == rank) return psdr_client_set_audio_range(g->ctx[rank], id, l, audio_mid, r);
    int mode, nid = -1;
    {
        std::lock_guard<std::mutex> lk(g->ctx[rank]->mtx);
        if (id >= (int)g->ctx[rank]->aslots.size() || !g->ctx[rank]->aslots[id].active) return fail(PSDR_ERR_INVALID, "no client %d in this group", *gid);
        mode = g->ctx[rank]->aslots[id].mode;
    }
    rc = psdr_client_add(g->ctx[want], &nid);
    if (rc) return rc;
    rc = psdr_client_set_audio_demodulation(g->ctx[want], nid, mode);
    if (!rc) rc = psdr_client_set_audio_range(g->ctx[want], nid, l, audio_mid, r);
    if (rc) {
        const std::string msg = psdr_last_error();
        psdr_client_remove(g->ctx[want], nid);
        return fail(rc, "%s", msg.c_str());
    }
    psdr_client_remove(g->ctx[rank], id);
    *gid = (want << 16) | nid;
    return PSDR_OK;
}

// ---- one batch ------------------------------------------------------------------------------------------------
// raw_root: nframes + 1 raw half-frames on the root device (nullptr: the root's ingest ring from first_half on)
static int group_step(psdr_group *g, const void *raw_root, uint64_t first_half, int nframes, uint64_t first_frame_num) {
    psdr_ctx *c0 = g->ctx[0];
    if (nframes < 1 || nframes > c0->max_batch) return fail(PSDR_ERR_INVALID, "nframes %d outside [1, max_batch=%d]", nframes, c0->max_batch);
    const size_t F = (size_t)nframes;
    const size_t hb = psdr_half_frame_bytes(c0);
    int rc;
    auto root_transform = [&]() -> int { return raw_root ? psdr_process_batch(c0, raw_root, nframes) : psdr_process_ring(c0, first_half, nframes); };
    g->timed = false;
    g->link_bytes = 0;
    if (g->shard == PSDR_SHARD_RAW) {
        // the raw half-frames cross the links, every GPU transforms them itself
        const unsigned char *src = (const unsigned char *)raw_root;
        if (!src) {
            if (!c0->ring.d) return fail(PSDR_ERR_STATE, "psdr_ring_create() on the root context first");
            const int s0 = (int)(first_half % (uint64_t)c0->ring.nhalves);
            if (s0 + nframes > c0->ring.nhalves) return fail(PSDR_ERR_INVALID, "frames cross the end of the ring: split the batch");
            src = c0->ring.d + (size_t)s0 * hb;
            for (int i = 0; i <= nframes; i++) {  // the copies of exactly the halves that are sent
                const int slot = (s0 + i) % c0->ring.nhalves;
                if (!c0->ring.ever_written[slot]) return fail(PSDR_ERR_STATE, "half-frame slot %d was never written", slot);
                HIPCHK(hipStreamWaitEvent(g->st[0], c0->ring.ev_written[slot], 0));
            }
        }
        const size_t bytes = (F + 1) * hb;
        if (g->comm_on) {
            HIPCHK(hipSetDevice(g->dev[0]));
            HIPCHK(hipEventRecord(g->ev0, g->st[0]));
            NCCLCHK(g_rccl.GroupStart());
            for (int r = 0; r < g->n; r++) {
                HIPCHK(hipSetDevice(g->dev[r]));
                // (root: in place - its own transform reads the caller's buffer / the ring)
                NCCLCHK(g_rccl.Broadcast(src, r == 0 ? (void *)src : g->rbuf[r], bytes, ncclChar, 0, g->comm[r], g->st[r]));
            }
            NCCLCHK(g_rccl.GroupEnd());
            HIPCHK(hipSetDevice(g->dev[0]));
            HIPCHK(hipEventRecord(g->ev1, g->st[0]));
            g->timed = true;
            g->link_bytes = (double)bytes;
        }
        rc = root_transform();
        if (rc) return rc;
        for (int r = 1; r < g->n; r++) {
            rc = psdr_process_batch(g->ctx[r], g->rbuf[r], nframes);
            if (rc) return rc;
        }
        for (int r = 0; r < g->n; r++) {
            rc = psdr_demod_batch(g->ctx[r], first_frame_num);
            if (rc) return rc;
        }
    } else {
        rc = root_transform();
        if (rc) return rc;
        if (g->shard == PSDR_SHARD_CLIENTS) {
            const size_t count = F * c0->spec_stride * 2;  // floats: the device layout, frames spec_stride bins apart
            if (g->comm_on) {
                HIPCHK(hipSetDevice(g->dev[0]));
                HIPCHK(hipEventRecord(g->ev0, g->st[0]));
                NCCLCHK(g_rccl.GroupStart());
                for (int r = 0; r < g->n; r++) {
                    HIPCHK(hipSetDevice(g->dev[r]));
                    // straight out of the root's spectrum buffer into every rank's own
                    NCCLCHK(g_rccl.Broadcast(c0->d_spec, r == 0 ? (void *)c0->d_spec : (void *)g->ctx[r]->d_spec, count, ncclFloat, 0, g->comm[r], g->st[r]));
                }
                NCCLCHK(g_rccl.GroupEnd());
                HIPCHK(hipSetDevice(g->dev[0]));
                HIPCHK(hipEventRecord(g->ev1, g->st[0]));
                g->timed = true;
                g->link_bytes = (double)count * sizeof(float);
            }
            rc = psdr_demod_batch(c0, first_frame_num);
            if (rc) return rc;
            for (int r = 1; r < g->n; r++) {
                rc = psdr_demod_batch_from(g->ctx[r], (const float *)g->ctx[r]->d_spec, g->ctx[r]->spec_stride, nframes, first_frame_num);
                if (rc) return rc;
            }
        } else {  // PSDR_SHARD_BAND
            const size_t count = F * g->band_stride * 2;  // floats per band
            std::vector<const float *> send(g->n, nullptr);
            for (int b = 1; b < g->n; b++) {
                if (g->banded) {
                    rc = psdr_band_region(c0, b, &send[b], nullptr, nullptr, nullptr);  // the region of THIS batch (the sets alternate)
                } else {
                    rc = psdr_pack_band(c0, nframes, g->band_first[b], g->band_bins[b], (float *)g->sbuf[b], g->band_stride);
                    send[b] = (const float *)g->sbuf[b];
                }
                if (rc) return rc;
            }
            if (g->comm_on && g->n > 1) {
                HIPCHK(hipSetDevice(g->dev[0]));
                HIPCHK(hipEventRecord(g->ev0, g->st[0]));
                NCCLCHK(g_rccl.GroupStart());
                for (int b = 1; b < g->n; b++) {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    NCCLCHK(g_rccl.Send(send[b], count, ncclFloat, b, g->comm[0], g->st[0]));
                    HIPCHK(hipSetDevice(g->dev[b]));
                    NCCLCHK(g_rccl.Recv(g->rbuf[b], count, ncclFloat, 0, g->comm[b], g->st[b]));
                }
                NCCLCHK(g_rccl.GroupEnd());
                HIPCHK(hipSetDevice(g->dev[0]));
                HIPCHK(hipEventRecord(g->ev1, g->st[0]));
                g->timed = true;
                g->link_bytes = (double)count * sizeof(float);
            }
            rc = psdr_demod_batch(c0, first_frame_num);  // the root's own clients read its spectrum through SpecLayout::pos
            if (rc) return rc;
            for (int b = 1; b < g->n; b++) {
                rc = g->banded ? psdr_demod_batch_from_band_region(g->ctx[b], (const float *)g->rbuf[b], g->band_stride, g->band_first[b], g->band_bins[b], nframes, first_frame_num)
                               : psdr_demod_batch_from_band(g->ctx[b], (const float *)g->rbuf[b], g->band_stride, g->band_first[b], g->band_bins[b], nframes, first_frame_num);
                if (rc) return rc;
            }
        }
    }
    rc = psdr_waterfall_batch(c0, first_frame_num);  // waterfall clients stay on the root (they read only its pyramid)
    if (rc) return rc;
    g->steps++;
    return PSDR_OK;
}
extern "C" int psdr_group_step(psdr_group *g, const void *d_halves_root, int nframes, uint64_t first_frame_num) {
    if (!g || !d_halves_root) return fail(PSDR_ERR_INVALID, "null argument");
    return group_step(g, d_halves_root, 0, nframes, first_frame_num);
}
extern "C" int psdr_group_step_ring(psdr_group *g, uint64_t first_half, int nframes, uint64_t first_frame_num) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    return group_step(g, nullptr, first_half, nframes, first_frame_num);
}
extern "C" int psdr_group_synchronize(psdr_group *g) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    for (int r = 0; r < g->n; r++) {
        int rc = psdr_synchronize(g->ctx[r]);
        if (rc) return rc;
    }
    return PSDR_OK;
}
// bytes that crossed ONE root -> peer link in the last step and how long the exchange took on the root's stream (0 when
// nothing was exchanged: one device, or time not yet available).  Synchronises the root.
extern "C" int psdr_group_link_stats(psdr_group *g, double *bytes_per_link, double *exchange_ms) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    if (bytes_per_link) *bytes_per_link = g->link_bytes;
    if (exchange_ms) {
        *exchange_ms = 0;
        if (g->timed) {
            HIPCHK(hipSetDevice(g->dev[0]));
            HIPCHK(hipEventSynchronize(g->ev1));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
            *exchange_ms = ms;
        }
    }
    return PSDR_OK;
}

// ---- results: one pinned copy per GPU and batch ----------------------------------------------------------------
extern "C" int psdr_group_fetch(psdr_group *g) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    bool any = false;
    for (int r = 0; r < g->n; r++) {
        const int rc = psdr_fetch_batch(g->ctx[r]);
        if (rc == PSDR_OK)
            any = true;
        else if (rc != PSDR_ERR_STATE)  // (PSDR_ERR_STATE: no client has been demodulated on that GPU yet)
            return rc;
    }
    return any ? PSDR_OK : fail(PSDR_ERR_STATE, "no demodulated batch to fetch on any device");
}
extern "C" int psdr_group_fetched_audio(psdr_group *g, int gid, int frame, const float **audio, float *pwr, int32_t *nan_flag, const int32_t **pcm) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_fetched_audio(g->ctx[rank], id, frame, audio, pwr, nan_flag, pcm);
}
extern "C" int psdr_group_fetched_window(psdr_group *g, int gid, int *l, double *audio_mid, int *r) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_fetched_window(g->ctx[rank], id, l, audio_mid, r);
}
