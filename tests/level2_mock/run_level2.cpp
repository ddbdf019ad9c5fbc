// Runs the Level-2 templates (phantomsdr_amd/host/hip_level2.h) against the mock of the reference's classes and a
// scripted fan-out: what reaches the encoders must be what AudioClient::send_audio / WaterfallClient::send_waterfall
// would have sent (src/signal.cpp:277-296, src/waterfall.cpp:44-51, src/fft.cpp:47-105).
#include "mock_reference.h"

#include <cassert>
#include <cmath>
#include <cstring>

struct psdr_ctx;
static size_t psdr_half_frame_bytes(const psdr_ctx *) { return 64; }
static std::vector<uint64_t> g_ring_waits;
static int psdr_ring_wait(psdr_ctx *, uint64_t half) {
    g_ring_waits.push_back(half);
    return 0;
}

std::vector<Call> g_calls;
std::mutex g_calls_mtx;
static void record(Call c) {
    std::scoped_lock lk(g_calls_mtx);
    g_calls.push_back(std::move(c));
}
void dsp_float_to_int16(float *arr, int32_t *output, float mult, size_t len) {
    for (size_t i = 0; i < len; i++) output[i] = (int32_t)std::lround(arr[i] * mult);
}
template <typename T> void DCBlocker<T>::removeDC(T *buf, size_t len) {
    for (size_t i = 0; i < len; i++) buf[i] += 1000;  // visible in the output: the CPU chain ran
}
void AGC::process(float *arr, size_t len) {
    for (size_t i = 0; i < len; i++) arr[i] *= 2;
}
void AudioEncoder::set_data(uint64_t frame_num, int l, double m, int r, double pwr) {
    pending = Call{"audio", frame_num, l, r, m, pwr, {}};
}
struct RecordingAudioEncoder : AudioEncoder {
    int process(int32_t *data, size_t size) override {
        Call c = pending;
        c.data.assign(data, data + size);
        record(c);
        return 0;
    }
};
struct RecordingWaterfallEncoder : WaterfallEncoder {
    int send(const void *buffer, size_t bytes, uint64_t frame_num, int l, int r) override {
        Call c{"waterfall", frame_num, l, r, 0, 0, {}};
        for (size_t i = 0; i < bytes; i++) c.data.push_back(((const int8_t *)buffer)[i]);
        record(c);
        return 0;
    }
};

// the scripted stand-in for HipFanout: same member functions, same signatures (phantomsdr_amd/host/hip_fanout.h)
class MockFanout {
  public:
    bool post_chain_on = false;
    int halves_pushed = 0, frames = 0;
    std::vector<uint64_t> frame_nums;
    std::vector<float> audio = std::vector<float>(8);
    std::vector<int32_t> pcm = std::vector<int32_t>(8);
    uint64_t cur = 0;
    psdr_ctx *context() { return nullptr; }
    void *alloc_half() {
        pool.emplace_back(64);
        return pool.back().data();
    }
    void push_half(const void *) { halves_pushed++; }
    bool process_frame(uint64_t frame_num) {
        frames++;
        frame_nums.push_back(frame_num);
        cur = frame_num;
        return true;
    }
    // what the frame loop decided for (client id, frame) BEFORE the frame was processed (src/websocket.cpp:170-176)
    std::vector<std::pair<int, bool>> pause_calls;   // (id, paused), in call order
    std::vector<int> pause_calls_at_frame;           // frames processed when the call was made
    std::vector<bool> paused_now = std::vector<bool>(8, false);
    bool set_audio_paused(int id, bool paused) {
        pause_calls.push_back({id, paused});
        pause_calls_at_frame.push_back(frames);
        if (id >= 0) paused_now[id] = paused;
        return id >= 0;
    }
    int freed = 0;
    void free_half(void *) { freed++; }
    struct AudioFrame {
        const float *audio = nullptr;
        const int32_t *pcm = nullptr;
        float average_power = 0;
        int l = 0, r = 0;
        double m = 0;
    };
    // client 0: always data; client 1: NaN-dropped on frame 1; client 2 (id -1): never attached; client 3: paused on the
    // frames its socket is backed up.  The window reported is the one the GPU demodulated with - for client 0 NOT the
    // one its CPU-side members hold on frame 5 (a set_audio_range the GPU refused)
    bool fetch_audio(int id, AudioFrame *fr) {
        if (id < 0) return false;
        if (id == 1 && cur == 1) return false;
        if (paused_now[id]) return false;
        for (int i = 0; i < 8; i++) audio[i] = (float)(i + 10 * id), pcm[i] = 100 * id + i + (int)cur;
        fr->audio = audio.data();
        fr->pcm = post_chain_on ? pcm.data() : nullptr;
        fr->average_power = 0.5f + (float)id;
        static const int L[4] = {20000, 41000, 0, 300}, R[4] = {20090, 41160, 0, 420};
        static const double M[4] = {20000.5, 41080.0, 0, 360.25};
        fr->l = L[id], fr->r = R[id], fr->m = M[id];
        return true;
    }
    bool fetch_waterfall(int id, std::vector<int8_t> &row, int *l, int *r) {
        if (id < 0) return false;
        row.assign(4, (int8_t)(cur + 1));
        *l = 40, *r = 48;  // l << level, r << level
        return true;
    }

  private:
    std::deque<std::vector<char>> pool;
};

#include "hip_level2.h"

void AudioClient::send_audio_hip(MockFanout *fo, size_t frame_num) { psdr_level2::Access::send_audio(*this, *fo, frame_num); }
void WaterfallClient::send_waterfall_hip(MockFanout *fo, size_t frame_num) { psdr_level2::Access::send_waterfall(*this, *fo, frame_num); }

struct RawScript {
    int left;
    int read(void *arr, int num) {
        if (left <= 0) return 0;
        left--;
        memset(arr, 7, (size_t)num);
        return num;
    }
};
void broadcast_server::fft_task_hip() {}

struct TestSetup {
    static int run(bool post_chain) {
        g_calls.clear();
        g_ring_waits.clear();
        broadcast_server srv;
        srv.fft_size = 1 << 16, srv.sps = 2048000, srv.downsample_levels = 2, srv.running = true, srv.frame_num = 0;  // skip_num = 6
        srv.waterfall_slices.resize(2);
        srv.waterfall_slice_mtx.resize(2);
        srv.fanout = std::make_unique<MockFanout>();
        srv.fanout->post_chain_on = post_chain;
        std::vector<std::shared_ptr<void>> cons;
        auto mk_audio = [&](int id, int l, double mid, int r) {
            auto c = std::make_shared<AudioClient>();
            cons.push_back(std::make_shared<int>(id));
            c->hdl = cons.back();
            c->l = l, c->r = r, c->audio_mid = mid, c->psdr_id = id, c->audio_fft_size = 16;
            c->audio_real.resize(16), c->audio_real_int16.resize(16);
            c->encoder = std::make_unique<RecordingAudioEncoder>();
            srv.signal_slices.insert({{l, r}, c});
            return c;
        };
        mk_audio(0, 20000, 20000.5, 20090)->r = 20111;  // (a window the GPU refused: the labels must still be the demodulated one's)
        mk_audio(1, 41000, 41080.0, 41160);
        mk_audio(-1, 100, 150.0, 200);  // every GPU slot taken: psdr_attach got -1
        auto slow_audio = mk_audio(3, 300, 360.25, 420);  // its socket is backed up on frames 2, 3 and 6
        void *slow_audio_hdl = cons.back().get();
        auto w = std::make_shared<WaterfallClient>();
        cons.push_back(std::make_shared<int>(9));
        w->hdl = cons.back();
        w->psdr_id = 0, w->level = 1, w->l = 20, w->r = 24;
        w->waterfall_encoder = std::make_unique<RecordingWaterfallEncoder>();
        srv.waterfall_slices[1].insert({{20, 24}, w});
        auto slow = std::make_shared<WaterfallClient>();  // a client whose socket is backed up: skipped (src/websocket.cpp:222-225)
        cons.push_back(std::make_shared<int>(10));
        slow->hdl = cons.back();
        slow->psdr_id = 1, slow->level = 0;
        slow->waterfall_encoder = std::make_unique<RecordingWaterfallEncoder>();
        srv.waterfall_slices[0].insert({{0, 8}, slow});
        RawScript raw{9};  // 9 half-frames -> 8 frames, then end of input
        void *slow_hdl = cons.back().get();
        psdr_level2::Access::fft_task(
            srv, raw, [](auto fn) { return std::async(std::launch::async, fn); },
            [&](connection_hdl h) -> size_t {
                if (h.lock().get() == slow_audio_hdl) return (srv.frame_num == 2 || srv.frame_num == 3 || srv.frame_num == 6) ? 50001 : 50000;
                return h.lock().get() == slow_hdl ? 60000 : 0;
            });
        auto &fo = *srv.fanout;
        assert(fo.halves_pushed == 9 && fo.frames == 8 && srv.frame_num == 8);
        for (int f = 0; f < 8; f++) assert(fo.frame_nums[f] == (uint64_t)f);
        assert(g_ring_waits.size() == 7 && g_ring_waits[0] == 0 && g_ring_waits[6] == 6);  // a buffer is reused from the 4th read on, once its copy has left the host
        // every audio client's pause decision was made BEFORE its frame was processed, once per client and frame
        assert(fo.pause_calls.size() == 4 * 8 && fo.freed == 3);
        for (size_t i = 0; i < fo.pause_calls.size(); i++) {
            const int f = (int)i / 4;
            assert(fo.pause_calls_at_frame[i] == f);
            const bool want = fo.pause_calls[i].first == 3 && (f == 2 || f == 3 || f == 6);
            assert(fo.pause_calls[i].second == want);
        }
        int na0 = 0, na1 = 0, na3 = 0, nw = 0;
        for (auto &c : g_calls) {
            if (c.what == "audio" && c.m == 360.25) {
                assert(c.frame_num != 2 && c.frame_num != 3 && c.frame_num != 6 && c.l == 0 && c.r == 120 && c.pwr == 3.5);
                na3++;
                continue;
            }
            if (c.what == "audio") {
                const bool c0 = c.m == 20000.5;
                assert(c0 || c.m == 41080.0);  // the client without a slot never sends
                // labels: l = audio_l = l - l = 0, r = audio_r = r - l, m = audio_mid (src/signal.cpp:104-105, 287)
                assert(c.l == 0 && c.r == (c0 ? 90 : 160));
                assert(c.pwr == (c0 ? 0.5 : 1.5) && c.data.size() == 8);
                if (!c0) assert(c.frame_num != 1);  // NaN guard dropped it
                const int id = c0 ? 0 : 1;
                for (int i = 0; i < 8; i++) {
                    if (post_chain)
                        assert(c.data[i] == 100 * id + i + (int)c.frame_num);  // the GPU's PCM, untouched
                    else
                        assert(c.data[i] == (int32_t)std::lround(((float)(i + 10 * id) + 1000) * 2 * 16384.0f));  // DC, AGC, int16 on the CPU
                }
                (c0 ? na0 : na1)++;
            } else {
                assert(c.frame_num % 6 == 0 && c.l == 40 && c.r == 48 && c.data.size() == 4 && c.data[0] == (int32_t)c.frame_num + 1);
                nw++;
            }
        }
        assert(na0 == 8 && na1 == 7 && na3 == 5 && nw == 2);  // waterfall frames 0 and 6; the backed-up client got nothing
        return 0;
    }
    static int run_without_users() {  // src/fft.cpp:70-80: the input is still read (and shipped), nothing is transformed
        broadcast_server srv;
        srv.fft_size = 1 << 16, srv.sps = 2048000, srv.downsample_levels = 2, srv.running = true, srv.frame_num = 0;
        srv.waterfall_slices.resize(2);
        srv.waterfall_slice_mtx.resize(2);
        srv.fanout = std::make_unique<MockFanout>();
        RawScript raw{5};
        psdr_level2::Access::fft_task(
            srv, raw, [](auto fn) { return std::async(std::launch::async, fn); }, [](connection_hdl) -> size_t { return 0; });
        assert(srv.fanout->halves_pushed == 5 && srv.fanout->frames == 0 && srv.frame_num == 0);
        return 0;
    }
};

int main() {
    if (TestSetup::run(false) || TestSetup::run(true) || TestSetup::run_without_users()) return 1;
    puts("level2 ok");
    return 0;
}
