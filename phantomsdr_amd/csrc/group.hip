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
// | PSDR_SHARD_PEER_COPY: no RCCL - the peers pull their share from the root with hipMemcpyPeerAsync on their own
//                       streams (one copy per root->peer link, ordered by events).  A device may then be listed more than
//                       once (n ranks on fewer GPUs): the whole multi-rank logic - placement, band regions and halos,
//                       migration, fetch - runs on a one-GPU box, only the transport differs.
// Everything of one rank - its kernels and its side of the collective - is enqueued in order on ONE stream per device
// (psdr_set_stream), so a step needs no host synchronisation; psdr_group_synchronize() drains all devices.
// Threads: client calls arrive on the server's websocket threads while the frame loop steps the group.  A client is named
// by a STABLE gid (an index into the group's table of (rank, slot)); every psdr_group_* call that touches the table, a
// stream or a client takes the group mutex, so a band migration can never interleave with a step's enqueue.
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

struct GroupClient {
    int rank = -1, id = -1;  // device rank and the slot of that rank's context; rank < 0: free entry
};
struct psdr_group {
    int n = 0, shard = 0;
    bool comm_on = false;           // RCCL collectives are issued (n > 1, or PSDR_SHARD_FORCE_COMM)
    bool peer_copy = false;         // PSDR_SHARD_PEER_COPY: peers pull with hipMemcpyPeerAsync instead
    std::mutex mtx;                 // the client table, next_rr, and every enqueue on the ranks' streams
    std::vector<GroupClient> clients;  // gid -> (rank, slot): the gid a caller holds never changes
    std::vector<hipEvent_t> ev_rank;   // per rank, on its device: migration hand-over / peer copy done
    std::vector<hipEvent_t> ev_t0, ev_t1;  // peer copy: the copy's own duration on the peer's stream
    hipEvent_t ev_x = nullptr;         // the root's data of this step is ready (peer copy; overlapped exchange)
    bool copies_pending = false;       // peer copy: ev_rank[r] of the last step not yet waited for by the root
    // The exchange of batch b beside the root's transform of batch b + 1 (PSDR_SHARD_CLIENTS, banded PSDR_SHARD_BAND): the
    // root alternates its two result sets (psdr_ctx::alt_sets) and issues ITS side of the collective on a stream of its
    // own (`xs`), behind ev_x; what it waits for - before it overwrites a set two steps later - is that set's exchange
    // (ev_xdone[set]; peer copies: ev_pull[set][r], recorded by the peers).  PSDR_SHARD_SERIAL switches it off (A/B, tests).
    bool overlap = false;
    hipStream_t xs = nullptr;
    hipEvent_t ev_xdone[2] = {nullptr, nullptr};
    bool xpending[2] = {false, false};
    std::vector<hipEvent_t> ev_pull[2];
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

// (callers hold g->mtx)
static int gid_split(psdr_group *g, int gid, int *rank, int *id) {
    if (gid < 0 || gid >= (int)g->clients.size() || g->clients[gid].rank < 0) return fail(PSDR_ERR_INVALID, "no client %d in this group", gid);
    *rank = g->clients[gid].rank;
    *id = g->clients[gid].id;
    return PSDR_OK;
}
static int gid_new(psdr_group *g, int rank, int id) {
    for (size_t i = 0; i < g->clients.size(); i++)
        if (g->clients[i].rank < 0) {
            g->clients[i] = GroupClient{rank, id};
            return (int)i;
        }
    g->clients.push_back(GroupClient{rank, id});
    return (int)g->clients.size() - 1;
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
        if (r == 0 && g->ev_x) hipEventDestroy(g->ev_x);
        if (r == 0)
            for (hipEvent_t e : g->ev_xdone)
                if (e) hipEventDestroy(e);
        for (auto &v : g->ev_pull)
            if (r < (int)v.size() && v[r]) hipEventDestroy(v[r]);
        if (r == 0 && g->xs) hipStreamDestroy(g->xs);
        if (r < (int)g->ev_rank.size() && g->ev_rank[r]) hipEventDestroy(g->ev_rank[r]);
        if (r < (int)g->ev_t0.size() && g->ev_t0[r]) hipEventDestroy(g->ev_t0[r]), hipEventDestroy(g->ev_t1[r]);
        if (g->ctx[r]) psdr_destroy(g->ctx[r]);
        if (r < (int)g->st.size() && g->st[r]) hipStreamDestroy(g->st[r]);
    }
    delete g;
}

extern "C" int psdr_group_create(const psdr_config *cfg, const int *devices, int ndevices, int shard, psdr_group **out) {
    if (!cfg || !devices || !out) return fail(PSDR_ERR_INVALID, "null argument");
    const bool force_comm = (shard & PSDR_SHARD_FORCE_COMM) != 0, peer_copy = (shard & PSDR_SHARD_PEER_COPY) != 0;
    const bool serial = (shard & PSDR_SHARD_SERIAL) != 0;
    shard &= ~(PSDR_SHARD_FORCE_COMM | PSDR_SHARD_PEER_COPY | PSDR_SHARD_SERIAL);
    if (ndevices < 1 || ndevices > 16) return fail(PSDR_ERR_INVALID, "a group has 1..16 devices, not %d", ndevices);
    if (shard < PSDR_SHARD_CLIENTS || shard > PSDR_SHARD_BAND) return fail(PSDR_ERR_INVALID, "unknown sharding %d", shard);
    if (force_comm && peer_copy) return fail(PSDR_ERR_INVALID, "PSDR_SHARD_FORCE_COMM (RCCL) and PSDR_SHARD_PEER_COPY (no RCCL) exclude each other");
    for (int i = 0; i < ndevices && !peer_copy; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j])
                return fail(PSDR_ERR_INVALID, "device %d listed twice (RCCL wants one rank per device; PSDR_SHARD_PEER_COPY allows it)", devices[i]);
    if (shard == PSDR_SHARD_BAND && (ndevices & (ndevices - 1))) return fail(PSDR_ERR_INVALID, "band sharding: a power-of-two number of devices, not %d", ndevices);
    psdr_group *g = new (std::nothrow) psdr_group();
    if (!g) return fail(PSDR_ERR_NOMEM, "out of memory");
    g->n = ndevices;
    g->shard = shard;
    g->peer_copy = peer_copy && ndevices > 1;
    g->comm_on = !peer_copy && (ndevices > 1 || force_comm);
    g->dev.assign(devices, devices + ndevices);
    g->ctx.assign(ndevices, nullptr);
    g->st.assign(ndevices, nullptr);
    g->comm.assign(ndevices, nullptr);
    g->rbuf.assign(ndevices, nullptr);
    g->ev_rank.assign(ndevices, nullptr);
    g->ev_t0.assign(ndevices, nullptr);
    g->ev_t1.assign(ndevices, nullptr);
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
        // (events belong to the device that is current when they are created)
        if (hipEventCreateWithFlags(&g->ev_rank[r], hipEventDisableTiming) != hipSuccess || hipEventCreate(&g->ev_t0[r]) != hipSuccess ||
            hipEventCreate(&g->ev_t1[r]) != hipSuccess) {
            fail(PSDR_ERR_HIP, "event creation on device %d failed", devices[r]);
            return bail(PSDR_ERR_HIP);
        }
        if (g->peer_copy && r > 0 && devices[r] != devices[0]) {
            (void)hipDeviceEnablePeerAccess(devices[0], 0);  // (direct xGMI reads; without it the copy is staged)
            (void)hipGetLastError();
        }
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
            // (one device with the collectives forced: band 0 - the whole spectrum - is packed, sent to and received from
            // oneself and demodulated from the received buffer, so that the pack + ncclSend / ncclRecv + band demodulation
            // path runs on a single-GPU box)
            for (int b = (ndevices == 1 && g->comm_on) ? 0 : 1; b < ndevices; b++) {
                if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&g->sbuf[b], F * g->band_stride * sizeof(cf)) != hipSuccess) {
                    fail(PSDR_ERR_NOMEM, "band send buffer of %zu bytes", F * g->band_stride * sizeof(cf));
                    return bail(PSDR_ERR_NOMEM);
                }
            }
        }
        g->rbuf_bytes = F * g->band_stride * sizeof(cf);
        for (int r = (ndevices == 1 && g->comm_on) ? 0 : 1; r < ndevices; r++) {
            if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&g->rbuf[r], g->rbuf_bytes) != hipSuccess) {
                fail(PSDR_ERR_NOMEM, "band receive buffer of %zu bytes on device %d", g->rbuf_bytes, devices[r]);
                return bail(PSDR_ERR_NOMEM);
            }
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_x, hipEventDisableTiming) != hipSuccess) {
        fail(PSDR_ERR_HIP, "event creation failed");
        return bail(PSDR_ERR_HIP);
    }
    // the exchange beside the next transform: spectrum broadcast, and band regions the root's second pass writes itself
    g->overlap = !serial && (g->comm_on || g->peer_copy) && (shard == PSDR_SHARD_CLIENTS || (shard == PSDR_SHARD_BAND && g->banded));
    if (g->overlap) {
        c0->alt_sets = true;
        bool ok = hipStreamCreateWithFlags(&g->xs, hipStreamNonBlocking) == hipSuccess;
        for (hipEvent_t &e : g->ev_xdone) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto &v : g->ev_pull) {
            v.assign(ndevices, nullptr);
            for (int r = 1; r < ndevices && ok; r++) ok = hipSetDevice(devices[r]) == hipSuccess && hipEventCreateWithFlags(&v[r], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {
            fail(PSDR_ERR_HIP, "exchange stream / events could not be created");
            return bail(PSDR_ERR_HIP);
        }
    }
    *out = g;
    return PSDR_OK;
}

extern "C" int psdr_group_size(const psdr_group *g) { return g ? g->n : 0; }
extern "C" psdr_ctx *psdr_group_ctx(psdr_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

// ---- audio clients: the group picks the GPU ---------------------------------------------------------------------
extern "C" int psdr_group_client_add(psdr_group *g, int l, double audio_mid, int r, int mode, int *gid_out) {
    if (!g || !gid_out) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
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
    *gid_out = gid_new(g, rank, id);
    return PSDR_OK;
}
extern "C" int psdr_group_client_remove(psdr_group *g, int gid) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    if (rc) return rc;
    g->clients[gid].rank = -1;
    return psdr_client_remove(g->ctx[rank], id);
}
extern "C" int psdr_group_client_set_audio_demodulation(psdr_group *g, int gid, int mode) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_client_set_audio_demodulation(g->ctx[rank], id, mode);
}
extern "C" int psdr_group_client_set_paused(psdr_group *g, int gid, int paused) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_client_set_paused(g->ctx[rank], id, paused);
}
extern "C" int psdr_group_client_rank(psdr_group *g, int gid) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id;
    return gid_split(g, gid, &rank, &id) ? -1 : rank;
}
// A retune inside the client's GPU is AudioClient::set_audio_range.  Band sharding only: a window that now starts in
// another band moves the client to that band's GPU BEHIND its gid (the caller's handle does not change).  What travels:
// the demodulation state the reference keeps across a retune (src/signal.cpp:81-94 touches none of it) - the SSB
// overlap-add tail, the AM / FM baseband tail and FM's last sample (one peer copy of the current state rows, ordered
// after the old device's last batch and before the new device's next one), the mode and the paused flag.  What does NOT
// travel: the post chain's history on the GPU (DC-blocker sums, AGC gain and look-ahead: the client starts there like a
// fresh one - an AGC transient of 0.2 s, where the reference has none), and the results of the batch demodulated before
// the move (psdr_group_fetched_audio answers PSDR_ERR_NO_DATA until the new device has demodulated a batch: one frame
// at F = 1).
extern "C" int psdr_group_client_set_audio_range(psdr_group *g, int gid, int l, double audio_mid, int r) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    if (rc) return rc;
    const int want = g->shard == PSDR_SHARD_BAND ? band_of(g, l) : rank;
    if (want == rank) return psdr_client_set_audio_range(g->ctx[rank], id, l, audio_mid, r);
    psdr_ctx *src = g->ctx[rank], *dst = g->ctx[want];
    int mode, nid = -1, src_par;
    bool paused;
    {
        std::lock_guard<std::mutex> lk2(src->mtx);
        if (id >= (int)src->aslots.size() || !src->aslots[id].active) return fail(PSDR_ERR_INVALID, "no client %d in this group", gid);
        mode = src->aslots[id].mode;
        paused = src->aslots[id].paused;
        src_par = src->aslots[id].state_cur;  // the parity the NEXT batch would read: what the last one wrote
    }
    rc = psdr_client_add(dst, &nid);
    if (rc) return rc;
    rc = psdr_client_set_audio_demodulation(dst, nid, mode);
    if (!rc) rc = psdr_client_set_audio_range(dst, nid, l, audio_mid, r);
    if (!rc && paused) rc = psdr_client_set_paused(dst, nid, 1);
    if (rc) {
        const std::string msg = psdr_last_error();
        psdr_client_remove(dst, nid);
        return fail(rc, "%s", msg.c_str());
    }
    // the state rows: [parity][slot][n/2] (ctx.h); the fresh slot reads parity 0 first.  Ordered on the two ranks' streams:
    // after everything the old device has enqueued, before anything the new one enqueues from here on; the old slot may be
    // handed out again only after the copy has read it.
    auto move_state = [&]() -> int {
        const size_t h = (size_t)src->n / 2, Ss = src->aslots.size(), Sd = dst->aslots.size();
        HIPCHK(hipSetDevice(g->dev[rank]));
        HIPCHK(hipEventRecord(g->ev_rank[rank], g->st[rank]));
        HIPCHK(hipSetDevice(g->dev[want]));
        HIPCHK(hipStreamWaitEvent(g->st[want], g->ev_rank[rank], 0));
        auto copy = [&](void *d, const void *sp, size_t bytes) -> hipError_t {
            return g->dev[rank] == g->dev[want] ? hipMemcpyAsync(d, sp, bytes, hipMemcpyDeviceToDevice, g->st[want])
                                                : hipMemcpyPeerAsync(d, g->dev[want], sp, g->dev[rank], bytes, g->st[want]);
        };
        HIPCHK(copy(dst->d_real_prev + ((size_t)0 * Sd + nid) * h, src->d_real_prev + ((size_t)src_par * Ss + id) * h, h * sizeof(float)));
        HIPCHK(copy(dst->d_bb_tail + ((size_t)0 * Sd + nid) * h, src->d_bb_tail + ((size_t)src_par * Ss + id) * h, h * sizeof(cf)));
        HIPCHK(copy(dst->d_bb_last + ((size_t)0 * Sd + nid), src->d_bb_last + ((size_t)src_par * Ss + id), sizeof(cf)));
        HIPCHK(hipEventRecord(g->ev_rank[want], g->st[want]));
        HIPCHK(hipSetDevice(g->dev[rank]));
        HIPCHK(hipStreamWaitEvent(g->st[rank], g->ev_rank[want], 0));
        return PSDR_OK;
    };
    rc = move_state();
    if (rc) {  // the client stays where it was (gid -> src), the half-made slot on the new device is given back
        const std::string msg = psdr_last_error();
        psdr_client_remove(dst, nid);
        return fail(rc, "%s", msg.c_str());
    }
    psdr_client_remove(src, id);
    g->clients[gid] = GroupClient{want, nid};
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
    // PSDR_SHARD_PEER_COPY: what the root is about to overwrite (its spectrum / band regions / send buffers) was the source
    // of the peers' copies of the previous step
    auto root_waits_for_copies = [&]() -> int {
        if (!g->copies_pending) return PSDR_OK;
        HIPCHK(hipSetDevice(g->dev[0]));
        for (int r = 1; r < g->n; r++) HIPCHK(hipStreamWaitEvent(g->st[0], g->ev_rank[r], 0));
        g->copies_pending = false;
        return PSDR_OK;
    };
    // ... and the pull itself: peer r copies `bytes` from the root's src[r] into dst[r] on ITS OWN stream, behind the
    // root's "data ready" event - n - 1 independent copies, one per root -> peer link
    auto peers_pull = [&](const std::vector<const void *> &src, const std::vector<void *> &dst, size_t bytes) -> int {
        HIPCHK(hipSetDevice(g->dev[0]));
        HIPCHK(hipEventRecord(g->ev_x, g->st[0]));
        for (int r = 1; r < g->n; r++) {
            HIPCHK(hipSetDevice(g->dev[r]));
            HIPCHK(hipStreamWaitEvent(g->st[r], g->ev_x, 0));
            HIPCHK(hipEventRecord(g->ev_t0[r], g->st[r]));
            if (g->dev[r] == g->dev[0])
                HIPCHK(hipMemcpyAsync(dst[r], src[r], bytes, hipMemcpyDeviceToDevice, g->st[r]));
            else
                HIPCHK(hipMemcpyPeerAsync(dst[r], g->dev[r], src[r], g->dev[0], bytes, g->st[r]));
            HIPCHK(hipEventRecord(g->ev_t1[r], g->st[r]));
            HIPCHK(hipEventRecord(g->ev_rank[r], g->st[r]));
        }
        g->copies_pending = true;
        g->timed = true;
        g->link_bytes = (double)bytes;
        return PSDR_OK;
    };
    // overlapped exchange: the result set the root's transform is about to overwrite (the OTHER one: process_frames toggles)
    // was the source of the exchange two steps ago - that, and only that, is waited for
    auto root_waits_for_set = [&](int set) -> int {
        HIPCHK(hipSetDevice(g->dev[0]));
        if (g->xpending[set]) {
            if (g->comm_on) HIPCHK(hipStreamWaitEvent(g->st[0], g->ev_xdone[set], 0));
            if (g->peer_copy)
                for (int r = 1; r < g->n; r++) HIPCHK(hipStreamWaitEvent(g->st[0], g->ev_pull[set][r], 0));
            g->xpending[set] = false;
        }
        return PSDR_OK;
    };
    // ... its side of the collective goes to the exchange stream, behind the transform; the peers' sides stay on their streams
    auto exchange_begin = [&]() -> int {
        HIPCHK(hipSetDevice(g->dev[0]));
        HIPCHK(hipEventRecord(g->ev_x, g->st[0]));
        HIPCHK(hipStreamWaitEvent(g->xs, g->ev_x, 0));
        HIPCHK(hipEventRecord(g->ev0, g->xs));
        return PSDR_OK;
    };
    auto exchange_end = [&](int set, double bytes) -> int {
        HIPCHK(hipSetDevice(g->dev[0]));
        HIPCHK(hipEventRecord(g->ev1, g->xs));
        HIPCHK(hipEventRecord(g->ev_xdone[set], g->xs));
        g->xpending[set] = true;
        g->timed = true;
        g->link_bytes = bytes;
        return PSDR_OK;
    };
    auto peers_pull_set = [&](int set, const std::vector<const void *> &src, const std::vector<void *> &dst, size_t bytes) -> int {
        HIPCHK(hipSetDevice(g->dev[0]));
        HIPCHK(hipEventRecord(g->ev_x, g->st[0]));
        for (int r = 1; r < g->n; r++) {
            HIPCHK(hipSetDevice(g->dev[r]));
            HIPCHK(hipStreamWaitEvent(g->st[r], g->ev_x, 0));
            HIPCHK(hipEventRecord(g->ev_t0[r], g->st[r]));
            if (g->dev[r] == g->dev[0])
                HIPCHK(hipMemcpyAsync(dst[r], src[r], bytes, hipMemcpyDeviceToDevice, g->st[r]));
            else
                HIPCHK(hipMemcpyPeerAsync(dst[r], g->dev[r], src[r], g->dev[0], bytes, g->st[r]));
            HIPCHK(hipEventRecord(g->ev_t1[r], g->st[r]));
            HIPCHK(hipEventRecord(g->ev_pull[set][r], g->st[r]));
        }
        g->xpending[set] = true;
        g->timed = true;
        g->link_bytes = (double)bytes;
        return PSDR_OK;
    };
    rc = g->overlap ? root_waits_for_set(c0->cur_set ^ 1) : root_waits_for_copies();
    if (rc) return rc;
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
        } else if (g->peer_copy) {
            rc = peers_pull(std::vector<const void *>(g->n, src), g->rbuf, bytes);
            if (rc) return rc;
            // the raw halves are the caller's (or the ingest ring's, whose slots are released by the root's first pass):
            // the root's transform - and with it everything the caller orders behind it - comes after the pulls
            rc = root_waits_for_copies();
            if (rc) return rc;
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
                // overlap: the root's side on the exchange stream (its own stream goes on with the demodulation and the next
                // batch's transform into the other result set)
                hipStream_t s0 = g->overlap ? g->xs : g->st[0];
                if (g->overlap) {
                    rc = exchange_begin();
                    if (rc) return rc;
                } else {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    HIPCHK(hipEventRecord(g->ev0, g->st[0]));
                }
                NCCLCHK(g_rccl.GroupStart());
                for (int r = 0; r < g->n; r++) {
                    HIPCHK(hipSetDevice(g->dev[r]));
                    // straight out of the root's spectrum buffer into every rank's own
                    NCCLCHK(g_rccl.Broadcast(c0->d_spec, r == 0 ? (void *)c0->d_spec : (void *)g->ctx[r]->d_spec, count, ncclFloat, 0, g->comm[r], r == 0 ? s0 : g->st[r]));
                }
                NCCLCHK(g_rccl.GroupEnd());
                if (g->overlap) {
                    rc = exchange_end(c0->cur_set, (double)count * sizeof(float));
                    if (rc) return rc;
                } else {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    HIPCHK(hipEventRecord(g->ev1, g->st[0]));
                    g->timed = true;
                    g->link_bytes = (double)count * sizeof(float);
                }
            } else if (g->peer_copy) {
                std::vector<void *> dst(g->n, nullptr);
                for (int r = 1; r < g->n; r++) dst[r] = g->ctx[r]->d_spec;
                rc = g->overlap ? peers_pull_set(c0->cur_set, std::vector<const void *>(g->n, c0->d_spec), dst, count * sizeof(float))
                                : peers_pull(std::vector<const void *>(g->n, c0->d_spec), dst, count * sizeof(float));
                if (rc) return rc;
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
            const bool self_loop = g->comm_on && g->n == 1;  // forced on one device: band 0 goes through RCCL to oneself
            if (self_loop) {
                rc = psdr_pack_band(c0, nframes, g->band_first[0], g->band_bins[0], (float *)g->sbuf[0], g->band_stride);
                if (rc) return rc;
                send[0] = (const float *)g->sbuf[0];
            }
            if (g->comm_on) {
                hipStream_t s0 = g->overlap ? g->xs : g->st[0];  // (overlap: banded regions only - they alternate with the result sets)
                if (g->overlap) {
                    rc = exchange_begin();
                    if (rc) return rc;
                } else {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    HIPCHK(hipEventRecord(g->ev0, g->st[0]));
                }
                NCCLCHK(g_rccl.GroupStart());
                for (int b = self_loop ? 0 : 1; b < g->n; b++) {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    NCCLCHK(g_rccl.Send(send[b], count, ncclFloat, b, g->comm[0], s0));
                    HIPCHK(hipSetDevice(g->dev[b]));
                    NCCLCHK(g_rccl.Recv(g->rbuf[b], count, ncclFloat, 0, g->comm[b], g->st[b]));
                }
                NCCLCHK(g_rccl.GroupEnd());
                if (g->overlap) {
                    rc = exchange_end(c0->cur_set, (double)count * sizeof(float));
                    if (rc) return rc;
                } else {
                    HIPCHK(hipSetDevice(g->dev[0]));
                    HIPCHK(hipEventRecord(g->ev1, g->st[0]));
                    g->timed = true;
                    g->link_bytes = (double)count * sizeof(float);
                }
            } else if (g->peer_copy) {
                std::vector<const void *> src(g->n, nullptr);
                for (int b = 1; b < g->n; b++) src[b] = send[b];
                rc = g->overlap ? peers_pull_set(c0->cur_set, src, g->rbuf, count * sizeof(float)) : peers_pull(src, g->rbuf, count * sizeof(float));
                if (rc) return rc;
            }
            // the root's own clients read its spectrum through SpecLayout::pos (self_loop: the band buffer that came back)
            rc = self_loop ? psdr_demod_batch_from_band(c0, (const float *)g->rbuf[0], g->band_stride, g->band_first[0], g->band_bins[0], nframes, first_frame_num)
                           : psdr_demod_batch(c0, first_frame_num);
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
    std::lock_guard<std::mutex> lk(g->mtx);  // (enqueue only: a step never waits for the device)
    return group_step(g, d_halves_root, 0, nframes, first_frame_num);
}
extern "C" int psdr_group_step_ring(psdr_group *g, uint64_t first_half, int nframes, uint64_t first_frame_num) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    return group_step(g, nullptr, first_half, nframes, first_frame_num);
}
extern "C" int psdr_group_synchronize(psdr_group *g) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    for (int r = 0; r < g->n; r++) {
        int rc = psdr_synchronize(g->ctx[r]);
        if (rc) return rc;
    }
    if (g->xs) {
        HIPCHK(hipSetDevice(g->dev[0]));
        HIPCHK(hipStreamSynchronize(g->xs));
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
        if (g->timed && g->peer_copy) {  // the slowest peer's own copy
            for (int r = 1; r < g->n; r++) {
                HIPCHK(hipSetDevice(g->dev[r]));
                HIPCHK(hipEventSynchronize(g->ev_t1[r]));
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, g->ev_t0[r], g->ev_t1[r]));
                *exchange_ms = std::max(*exchange_ms, (double)ms);
            }
        } else if (g->timed) {
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
    for (int r = 0; r < g->n; r++) {  // (no group lock: this waits for the devices; a context's own mutex guards its slots)
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
    // (the lock is held across the lookup AND the answer: a migration or a remove + add between the two would make the call act
    // on a stale (rank, id); psdr_fetched_* never wait for a device)
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_fetched_audio(g->ctx[rank], id, frame, audio, pwr, nan_flag, pcm);
}
extern "C" int psdr_group_fetched_window(psdr_group *g, int gid, int *l, double *audio_mid, int *r) {
    if (!g) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mtx);
    int rank, id, rc = gid_split(g, gid, &rank, &id);
    return rc ? rc : psdr_fetched_window(g->ctx[rank], id, l, audio_mid, r);
}
