// hip_fanout.h — Level 2 of the drop-in: the per-frame fan-out of broadcast_server on the GPU.
//
// The reference posts one asio task per client and frame:
//   signal_loop()     src/websocket.cpp:156-185  -> AudioClient::send_audio      src/signal.cpp:102-298
//   waterfall_loop()  src/websocket.cpp:207-236  -> WaterfallClient::send_waterfall src/waterfall.cpp:44-51
// With this class the slice, the small inverse transform, the demodulation (and optionally the DC
// blocker / AGC / int16 conversion) of ALL audio clients and the byte gather of ALL waterfall clients
// run as a few kernel launches per frame; what stays per client on the asio pool is the hand-over to the
// encoder and the socket, exactly the tail of send_audio (src/signal.cpp:277-296) and send_waterfall.
// The class only forwards to the C-ABI (include/psdr.h); integration/level2.patch shows every line of
// the server that changes.
#ifndef PSDR_HIP_FANOUT_H
#define PSDR_HIP_FANOUT_H

#include <cstdint>
#include <stdexcept>
#include <vector>

#include "psdr.h"

class HipFanout {
  public:
    // the derived parameters broadcast_server computes in its constructor (src/spectrumserver.cpp:99-105,
    // :151, :186-190) and fft_task (src/fft.cpp:33)
    struct Params {
        uint32_t fft_size;
        bool is_real;
        int downsample_levels, brightness_offset, audio_max_fft_size, audio_max_sps, skip_num, min_waterfall_fft;
        psdr_format input_format;  // input.driver.format, src/spectrumserver.cpp:349-364
        int max_audio_clients, max_waterfall_clients;
        bool post_chain;           // DC blocker + AGC + int16 on the GPU too
        int ring_halves;           // half-frames of raw samples kept in HBM (>= 3)
        // more than one GPU of the node: devices[0] owns the ring, the FFT and the waterfall clients, the
        // audio clients are spread over all of them and every frame's spectrum crosses xGMI once (psdr_group_*,
        // RCCL called from the library).  Empty = device 0 alone.
        std::vector<int> devices;
        int shard = PSDR_SHARD_CLIENTS;  // PSDR_SHARD_CLIENTS | _RAW | _BAND
        bool force_group = false;        // the group path (and RCCL) even with one device: testing
    };
    // (the constructor and the frame loop's own calls throw std::runtime_error on failure: they run on the server's
    // main / fft_task threads, where the reference's own set-up throws too)
    explicit HipFanout(const Params &p) : ctx{nullptr}, prm{p}, next_half{0} {
        psdr_config cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.fft_size = p.fft_size;
        cfg.is_real = p.is_real ? 1 : 0;
        cfg.downsample_levels = p.downsample_levels;
        cfg.brightness_offset = p.brightness_offset;
        cfg.additional_size = p.audio_max_fft_size;
        cfg.audio_fft_size = p.audio_max_fft_size;
        cfg.audio_rate = p.audio_max_sps;
        cfg.input_format = p.input_format;
        cfg.max_batch = 1;  // a live receiver transforms every frame as it arrives
        cfg.max_clients = p.max_audio_clients;
        cfg.max_waterfall_clients = p.max_waterfall_clients;
        cfg.skip_num = p.skip_num;
        cfg.waterfall_size = p.min_waterfall_fft;
        if (p.devices.size() > 1 || p.force_group) {
            const std::vector<int> devs = p.devices.empty() ? std::vector<int>{0} : p.devices;
            chk(psdr_group_create(&cfg, devs.data(), (int)devs.size(), p.shard | (p.force_group ? PSDR_SHARD_FORCE_COMM : 0), &grp));
            ctx = psdr_group_ctx(grp, 0);
        } else {
            cfg.device = p.devices.empty() ? 0 : p.devices[0];
            chk(psdr_create(&cfg, &ctx));
        }
        chk(psdr_ring_create(ctx, p.ring_halves));
        if (p.post_chain)
            for (int r = 0; r < ranks(); r++) chk(psdr_set_post_chain(rank_ctx(r), 1));
    }
    ~HipFanout() {
        if (grp)
            psdr_group_destroy(grp);
        else
            psdr_destroy(ctx);
    }
    HipFanout(const HipFanout &) = delete;
    HipFanout &operator=(const HipFanout &) = delete;

    // ---- fft_task (src/fft.cpp:47-105) -------------------------------------------------------------
    // pinned buffer for reader->read() to fill with RAW samples (no CPU conversion: src/samplereader.cpp:29-40
    // runs inside the first FFT pass)
    void *alloc_half() {
        float *p = nullptr;
        chk(psdr_host_alloc(ctx, (psdr_half_frame_bytes(ctx) + 3) / 4, &p));
        return p;
    }
    // (the reference frees its input_buffers when fft_task ends, src/fft.cpp:116-118; every copy out of the buffer has
    // left the host first)
    void free_half(void *buf) {
        psdr_synchronize(ctx);
        psdr_host_free(ctx, (float *)buf);
    }
    // a new half-frame has been read: its copy to HBM starts at once and overlaps the GPU work on the
    // previous frame (the reference overlaps the read of half k+2 with the FFT of (k, k+1), src/fft.cpp:56-67)
    void push_half(const void *raw_half) { chk(psdr_ring_write_async(ctx, next_half++, raw_half)); }
    // the frame made of the two newest half-frames: FFT + pyramid, then every client (src/fft.cpp:61-105), and ONE
    // copy of all clients' results into pinned host memory (psdr_fetch_batch) - the per-client tasks the server posts
    // afterwards only read that block.  frame_num is the server's counter before its increment.  Returns false when
    // there is nothing to send yet (fewer than two half-frames).
    // ONE frame per call: a window / mode / pause change made between two calls lands on exactly the frame the reference's
    // next send_audio would have used it for (INTEGRATION.md "Command timing"; with larger batches - psdr_process_batch
    // with F > 1 - a change lands on the next batch boundary instead).
    bool process_frame(uint64_t frame_num) {
        if (next_half < 2) return false;
        const uint64_t first = next_half - 2;
        // a frame window must not cross the ring end more than by the guard half-frame
        if (grp) {  // the root transforms, the spectrum crosses xGMI, every GPU demodulates its clients
            chk(psdr_group_step_ring(grp, first, 1, frame_num));
            have_audio = psdr_group_fetch(grp) == PSDR_OK;
            return true;
        }
        chk(psdr_process_ring(ctx, first, 1));
        chk(psdr_demod_batch(ctx, frame_num));                                         // signal_loop()
        if (frame_num % (uint64_t)prm.skip_num == 0) chk(psdr_waterfall_batch(ctx, frame_num));  // waterfall_loop()
        have_audio = psdr_fetch_batch(ctx) == PSDR_OK;  // (PSDR_ERR_STATE: no audio client yet)
        return true;
    }

    // ---- AudioClient (src/signal.cpp:8-97, 300-336) --------------------------------------------------
    // The hooks below run on websocket / asio handler threads inside AudioClient::set_audio_range and
    // ::set_audio_demodulation, which never throw in the reference: they report failure through their return value
    // (the previous window / mode stays in force on the GPU) instead of an exception.
    // (with more than one GPU the id is the group's: it names the device as well; band sharding needs the window to pick
    // the device, so a client is created at its first set_audio_range there)
    int add_audio_client() {
        int id = -1;
        if (grp) return psdr_group_client_add(grp, 0, 0.0, 0, PSDR_USB, &id) == PSDR_OK ? id : -1;
        return psdr_client_add(ctx, &id) == PSDR_OK ? id : -1;
    }
    void remove_audio_client(int id) {
        if (id < 0) return;
        if (grp)
            psdr_group_client_remove(grp, id);
        else
            psdr_client_remove(ctx, id);
    }
    // (group: band sharding may move the client to another GPU BEHIND its id - the id a client holds never changes, so
    // the websocket thread that retunes, the frame loop that reads client->psdr_id under signal_slice_mtx and the asio
    // send tasks that read it without a lock all see one constant; the group's own mutex orders the move against a step)
    bool set_audio_range(int id, int l, double m, int r) {
        if (id < 0) return false;
        return (grp ? psdr_group_client_set_audio_range(grp, id, l, m, r) : psdr_client_set_audio_range(ctx, id, l, m, r)) == PSDR_OK;
    }
    bool on_audio_window_message(int id, int l, double m, int r) {
        if (id < 0) return false;
        if (!grp) return psdr_client_on_window_message(ctx, id, l, m, r) == PSDR_OK;  // false: the reference returns silently
        const int R = (int)(prm.is_real ? prm.fft_size / 2 : prm.fft_size);           // src/signal.cpp:305-311
        if (l < 0 || l >= R || r < 0 || r >= R || l > r || r - l > prm.audio_max_fft_size) return false;
        return psdr_group_client_set_audio_range(grp, id, l, m, r) == PSDR_OK;
    }
    bool set_audio_demodulation(int id, psdr_mode mode) {
        if (id < 0) return false;
        return (grp ? psdr_group_client_set_audio_demodulation(grp, id, mode) : psdr_client_set_audio_demodulation(ctx, id, mode)) == PSDR_OK;
    }
    // signal_loop's slow-client rule (src/websocket.cpp:170-176): no send_audio call at all for a client whose socket is
    // backed up - its overlap-add tails, FM sample, DC blocker and AGC stand still.  Called by the frame loop BEFORE
    // process_frame() for every audio client (hip_level2.h).
    bool set_audio_paused(int id, bool paused) {
        if (id < 0) return false;
        return (grp ? psdr_group_client_set_paused(grp, id, paused ? 1 : 0) : psdr_client_set_paused(ctx, id, paused ? 1 : 0)) == PSDR_OK;
    }
    // the tail of send_audio for one client (asio pool): pointers into the block process_frame() fetched; no device
    // call.  Returns false when there is nothing to send: the reference would have dropped the frame (NaN guard,
    // src/signal.cpp:266-271), the client was paused for this frame, or it attached after the frame was demodulated.
    // audio: audio_max_fft_size/2 floats; pcm: as many int32, nullptr unless the post chain runs on the GPU; l, m, r: the
    // window this frame was DEMODULATED with (the labels of src/signal.cpp:104-105, 287 must describe the samples they
    // travel with: the CPU-side l / audio_mid / r may have moved since, or hold a window the GPU refused).
    struct AudioFrame {
        const float *audio = nullptr;
        const int32_t *pcm = nullptr;
        float average_power = 0;
        int l = 0, r = 0;
        double m = 0;
    };
    bool fetch_audio(int id, AudioFrame *out) {
        int32_t nan = 0;
        if (!have_audio || id < 0) return false;
        if (grp) {
            if (psdr_group_fetched_audio(grp, id, 0, &out->audio, &out->average_power, &nan, &out->pcm) != PSDR_OK) return false;
            if (psdr_group_fetched_window(grp, id, &out->l, &out->m, &out->r) != PSDR_OK) return false;
            return nan == 0;
        }
        if (psdr_fetched_audio(ctx, id, 0, &out->audio, &out->average_power, &nan, &out->pcm) != PSDR_OK) return false;
        if (psdr_fetched_window(ctx, id, &out->l, &out->m, &out->r) != PSDR_OK) return false;
        return nan == 0;
    }

    // ---- WaterfallClient (src/waterfall.cpp:6-99) -----------------------------------------------------
    int add_waterfall_client() {
        int id = -1;
        return psdr_waterfall_add(ctx, &id) == PSDR_OK ? id : -1;
    }
    void remove_waterfall_client(int id) {
        if (id >= 0) psdr_waterfall_remove(ctx, id);
    }
    bool on_waterfall_window_message(int id, int l, int r, int *level, int *nl, int *nr) {
        return psdr_waterfall_on_window_message(ctx, id, l, r, level, nl, nr) == PSDR_OK;
    }
    // the bytes send_waterfall hands to the encoder, with the labels (l << level, r << level) of the
    // batch they were gathered in; empty when this frame was not a waterfall frame
    bool fetch_waterfall(int id, std::vector<int8_t> &row, int *l_label, int *r_label) {
        int ns = 0, lv = 0, l = 0, r = 0;
        if (id < 0 || psdr_read_waterfall(ctx, id, nullptr, 0, &ns, &lv, &l, &r) != PSDR_OK || ns == 0) return false;
        row.resize((size_t)ns * (size_t)(r - l));
        if (psdr_read_waterfall(ctx, id, row.data(), row.size(), &ns, &lv, &l, &r) != PSDR_OK) return false;
        *l_label = l << lv;
        *r_label = r << lv;
        return true;
    }
    psdr_ctx *context() { return ctx; }  // (the root's: ring, FFT, waterfall clients)
    bool post_chain() const { return prm.post_chain; }
    int ranks() const { return grp ? psdr_group_size(grp) : 1; }

  private:
    psdr_ctx *rank_ctx(int r) { return grp ? psdr_group_ctx(grp, r) : ctx; }
    psdr_ctx *ctx;
    psdr_group *grp = nullptr;
    Params prm;
    uint64_t next_half;
    bool have_audio = false;
    static void chk(int rc) {
        if (rc != PSDR_OK) throw std::runtime_error(psdr_last_error());
    }
};

#endif
