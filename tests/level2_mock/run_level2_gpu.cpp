// Level 2 of the drop-in END TO END on the GPU: the templates of phantomsdr_amd/host/hip_level2.h (the bodies of
// fft_task_hip / send_audio_hip / send_waterfall_hip that integration/src/fft_hip.cpp instantiates) run with the REAL
// HipFanout (phantomsdr_amd/host/hip_fanout.h) on the real libpsdr_hip.so, against the mock of the reference's server
// classes (mock_reference.h: exactly the members the reference declares).  A script drives what a live server would
// see - clients with windows and modes, sockets that back up on given frames (src/websocket.cpp:170-176), a mode or
// window change mid-stream - and everything that reaches the encoders is dumped for tests/test_gpu_level2.py, which
// drives the oracle the same way (a skipped frame = no send_audio call) and compares.
//
// The CPU post chain of the mock server (dc.removeDC / agc.process / dsp_float_to_int16, src/signal.cpp:277-284) is the
// ORACLE's restatement of those classes (test infrastructure may link it): with the GPU chain off, the PCM that reaches
// the encoder is GPU float audio -> the reference's CPU chain, as in a patched server.
//
//   run_level2_gpu <script.txt> <raw.bin> <out.bin>
#define MockFanout HipFanout
#include "hip_fanout.h"
#include "mock_reference.h"

#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

extern "C" {
typedef struct orc_dcblocker orc_dcblocker;
typedef struct orc_agc orc_agc;
orc_dcblocker *orc_dc_create(int delay);
void orc_dc_remove(orc_dcblocker *d, float *arr, int length);
orc_agc *orc_agc_create(float desired, float attack_ms, float release_ms, float look_ahead_ms, float sr);
void orc_agc_process(orc_agc *a, float *arr, size_t len);
void orc_agc_reset(orc_agc *a);
void orc_float_to_int16(const float *arr, int32_t *out, float mult, size_t len);
}

std::vector<Call> g_calls;
std::mutex g_calls_mtx;
static int g_audio_rate = 12000;
// the mock's DCBlocker / AGC are stateless shells (mock_reference.h): their state lives here, keyed by the member's address
static std::map<const void *, orc_dcblocker *> g_dc;
static std::map<const void *, orc_agc *> g_agc;
static std::mutex g_state_mtx;
static std::atomic<int> g_tasks_in_flight{0};  // posted send_audio / send_waterfall tasks that have not finished
void dsp_float_to_int16(float *arr, int32_t *output, float mult, size_t len) { orc_float_to_int16(arr, output, mult, len); }
template <typename T> void DCBlocker<T>::removeDC(T *buf, size_t len) {
    orc_dcblocker *d;
    {
        std::scoped_lock lk(g_state_mtx);
        auto &p = g_dc[this];
        if (!p) p = orc_dc_create(g_audio_rate / 750 * 2);  // src/signal.cpp:54
        d = p;
    }
    orc_dc_remove(d, buf, (int)len);
}
static orc_agc *agc_of(const void *key) {
    std::scoped_lock lk(g_state_mtx);
    auto &p = g_agc[key];
    if (!p) p = orc_agc_create(0.2f, 50.0f, 300.0f, 200.0f, (float)g_audio_rate);  // src/signal.cpp:55
    return p;
}
void AGC::process(float *arr, size_t len) { orc_agc_process(agc_of(this), arr, len); }
void AudioEncoder::set_data(uint64_t frame_num, int l, double m, int r, double pwr) { pending = Call{"audio", frame_num, l, r, m, pwr, {}}; }

struct Rec {
    int kind, client;  // 0 audio, 1 waterfall
    Call c;
};
static std::vector<Rec> g_recs;
struct DumpAudioEncoder : AudioEncoder {
    int id;
    explicit DumpAudioEncoder(int i) : id(i) {}
    int process(int32_t *data, size_t size) override {
        Call c = pending;
        c.data.assign(data, data + size);
        std::scoped_lock lk(g_calls_mtx);
        g_recs.push_back({0, id, c});
        return 0;
    }
};
struct DumpWaterfallEncoder : WaterfallEncoder {
    int id;
    explicit DumpWaterfallEncoder(int i) : id(i) {}
    int send(const void *buffer, size_t bytes, uint64_t frame_num, int l, int r) override {
        Call c{"waterfall", frame_num, l, r, 0, 0, {}};
        for (size_t i = 0; i < bytes; i++) c.data.push_back(((const int8_t *)buffer)[i]);
        std::scoped_lock lk(g_calls_mtx);
        g_recs.push_back({1, id, c});
        return 0;
    }
};

#include "hip_level2.h"

void AudioClient::send_audio_hip(HipFanout *fo, size_t frame_num) { psdr_level2::Access::send_audio(*this, *fo, frame_num); }
void WaterfallClient::send_waterfall_hip(HipFanout *fo, size_t frame_num) { psdr_level2::Access::send_waterfall(*this, *fo, frame_num); }
void broadcast_server::fft_task_hip() {}

struct Event {
    std::string what;
    int client, frame, mode, l, r;
    double m;
};

struct TestSetup {
    static int run(const char *script, const char *rawfile, const char *outfile) {
        // ---- the script
        int log2n = 16, is_real = 0, sps = 2048000, post_chain = 0, brightness = 0, audio_sps = 12000, wf_size = 1024, group = 0, shard = 0;
        std::string fmt = "s16";
        struct CSpec {
            int mode, l, r;
            double m;
        };
        std::vector<CSpec> cspec;
        std::vector<std::pair<int, int>> wspec;
        std::vector<Event> events;
        {
            std::ifstream in(script);
            std::string line;
            while (std::getline(in, line)) {
                std::istringstream ss(line);
                std::string w;
                if (!(ss >> w) || w[0] == '#') continue;
                if (w == "config") ss >> log2n >> is_real >> sps >> fmt >> post_chain >> brightness >> audio_sps >> wf_size >> group >> shard;
                else if (w == "client") {
                    CSpec c{};
                    ss >> c.mode >> c.l >> c.m >> c.r;
                    cspec.push_back(c);
                } else if (w == "wf") {
                    int l, r;
                    ss >> l >> r;
                    wspec.push_back({l, r});
                } else {
                    Event e{w, 0, 0, 0, 0, 0, 0};
                    ss >> e.client >> e.frame;
                    if (w == "mode") ss >> e.mode;
                    if (w == "window") ss >> e.l >> e.m >> e.r;
                    events.push_back(e);
                }
            }
        }
        g_audio_rate = audio_sps;
        const int N = 1 << log2n, R = is_real ? N / 2 : N;
        const int n = (int)std::ceil((double)audio_sps * N / sps / 4.) * 4;  // src/spectrumserver.cpp:151
        int levels = 0;
        for (int cur = R; cur >= wf_size; cur /= 2) levels++;               // :186-190
        broadcast_server srv;
        srv.fft_size = N, srv.fft_result_size = R, srv.sps = sps, srv.is_real = is_real, srv.downsample_levels = levels;
        srv.running = true, srv.frame_num = 0;
        srv.waterfall_slices.resize(levels);
        srv.waterfall_slice_mtx.resize(levels);
        HipFanout::Params hp{};
        hp.fft_size = (uint32_t)N;
        hp.is_real = is_real;
        hp.downsample_levels = levels;
        hp.brightness_offset = brightness;
        hp.audio_max_fft_size = n;
        hp.audio_max_sps = audio_sps;
        hp.skip_num = std::max(1, (int)std::floor(((float)sps / N) / 10.) * 2);
        hp.min_waterfall_fft = wf_size;
        hp.input_format = fmt == "u8" ? PSDR_FMT_U8 : fmt == "s16" ? PSDR_FMT_S16 : PSDR_FMT_F32;
        hp.max_audio_clients = 16, hp.max_waterfall_clients = 8;
        hp.post_chain = post_chain != 0;
        hp.ring_halves = 8;
        hp.force_group = group != 0;  // the multi-GPU path (psdr_group_*, RCCL from the library) on the one device of the box
        hp.shard = shard;
        srv.fanout = std::make_unique<HipFanout>(hp);
        HipFanout &fo = *srv.fanout;
        // ---- clients, attached the way the patched server does it (integration/level2.patch: psdr_attach + the hooks
        // in set_audio_range / set_audio_demodulation)
        std::vector<std::shared_ptr<void>> cons;
        std::vector<std::shared_ptr<AudioClient>> acl;
        std::vector<std::shared_ptr<WaterfallClient>> wcl;
        for (size_t i = 0; i < cspec.size(); i++) {
            auto c = std::make_shared<AudioClient>();
            cons.push_back(std::make_shared<int>((int)i));
            c->hdl = cons.back();
            c->audio_fft_size = n;
            c->audio_real.resize(n), c->audio_real_int16.resize(n);
            c->encoder = std::make_unique<DumpAudioEncoder>((int)i);
            c->psdr_fo = &fo;
            c->psdr_id = fo.add_audio_client();
            c->demodulation = (demodulation_mode)cspec[i].mode;
            fo.set_audio_demodulation(c->psdr_id, (psdr_mode)cspec[i].mode);
            c->l = cspec[i].l, c->r = cspec[i].r, c->audio_mid = cspec[i].m;
            if (!fo.set_audio_range(c->psdr_id, c->l, c->audio_mid, c->r)) return 3;
            srv.signal_slices.insert({{c->l, c->r}, c});
            acl.push_back(c);
        }
        for (size_t i = 0; i < wspec.size(); i++) {
            auto w = std::make_shared<WaterfallClient>();
            cons.push_back(std::make_shared<int>(100 + (int)i));
            w->hdl = cons.back();
            w->waterfall_encoder = std::make_unique<DumpWaterfallEncoder>((int)i);
            w->psdr_fo = &fo;
            w->psdr_id = fo.add_waterfall_client();
            int lv = 0, nl = 0, nr = 0;
            if (!fo.on_waterfall_window_message(w->psdr_id, wspec[i].first, wspec[i].second, &lv, &nl, &nr)) return 4;
            w->level = lv, w->l = nl, w->r = nr;
            srv.waterfall_slices[lv].insert({{nl, nr}, w});
            wcl.push_back(w);
        }
        // ---- the raw sample stream; read() number k >= 1 precedes frame k - 1: the script's events for that frame are
        // applied there, on the frame loop's thread, like a websocket message that arrived between two frames
        struct Raw {
            FILE *f;
            int calls = 0;
            std::vector<Event> *ev;
            std::vector<std::shared_ptr<AudioClient>> *acl;
            HipFanout *fo;
            int read(void *arr, int num) {
                const int frame = calls - 1;
                calls++;
                // The send tasks of the frame before run on pool threads and fft_task waits for them only AFTER this
                // read (src/fft.cpp:82-88): a mode change applied here could reset the AGC under a task that is still
                // converting that frame (seen once in ~100 runs: client 1's last LSB packet before its change to AM).
                // The reference has the same window between its websocket and pool threads; the oracle this test
                // compares with has not, so the script's events wait for the tasks.
                while (g_tasks_in_flight.load() != 0) std::this_thread::yield();
                for (auto &e : *ev) {
                    if (e.frame != frame) continue;
                    auto &c = (*acl)[e.client];
                    if (e.what == "mode") {  // AudioClient::on_demodulation_message, src/signal.cpp:316-328 + its hook
                        TestSetup::set_mode(*c, *fo, e.mode);
                    } else if (e.what == "window") {  // AudioClient::set_audio_range, src/signal.cpp:81-94 + its hook
                        TestSetup::set_window(*c, *fo, e.l, e.m, e.r);
                    }
                }
                return (int)fread(arr, 1, (size_t)num, f);
            }
        } raw{fopen(rawfile, "rb"), 0, &events, &acl, &fo};
        if (!raw.f) return 5;
        auto slow = [&](int id, bool wf) {
            for (auto &e : events)
                if (e.what == (wf ? "slowwf" : "slow") && e.client == id && e.frame == srv.frame_num) return true;
            return false;
        };
        psdr_level2::Access::fft_task(
            srv, raw,
            [](auto fn) {
                g_tasks_in_flight++;
                return std::async(std::launch::async, [fn] {
                    fn();
                    g_tasks_in_flight--;
                });
            },
            [&](connection_hdl h) -> size_t {
                const int tag = *std::static_pointer_cast<int>(h.lock());
                return (tag >= 100 ? slow(tag - 100, true) : slow(tag, false)) ? 50001 : 50000;  // > 50000 is slow (src/websocket.cpp:174)
            });
        fclose(raw.f);
        // ---- dump
        FILE *o = fopen(outfile, "wb");
        if (!o) return 6;
        const int32_t hdr[4] = {n, levels, (int32_t)srv.frame_num, hp.skip_num};
        fwrite(hdr, 4, 4, o);
        for (auto &r : g_recs) {
            const int32_t h2[6] = {r.kind, r.client, (int32_t)r.c.frame_num, r.c.l, r.c.r, (int32_t)r.c.data.size()};
            fwrite(h2, 4, 6, o);
            const double d2[2] = {r.c.m, r.c.pwr};
            fwrite(d2, 8, 2, o);
            fwrite(r.c.data.data(), 4, r.c.data.size(), o);
        }
        fclose(o);
        printf("level2 gpu: %d frames, %zu records\n", srv.frame_num, g_recs.size());
        return 0;
    }
    static void set_mode(AudioClient &c, HipFanout &fo, int mode) {
        c.demodulation = (demodulation_mode)mode;
        orc_agc_reset(agc_of(&c.agc));                        // this->agc.reset(), src/signal.cpp:327
        fo.set_audio_demodulation(c.psdr_id, (psdr_mode)mode);  // the hook (also resets the GPU AGC)
    }
    static void set_window(AudioClient &c, HipFanout &fo, int l, double m, int r) {
        c.audio_mid = m, c.l = l, c.r = r;  // the CPU side stores it unconditionally (src/signal.cpp:81-94) ...
        (void)fo.set_audio_range(c.psdr_id, l, m, r);  // ... the GPU may refuse it: the old window stays in force
    }
};

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s script raw out\n", argv[0]);
        return 2;
    }
    try {
        return TestSetup::run(argv[1], argv[2], argv[3]);
    } catch (const std::exception &e) {
        fprintf(stderr, "level2 gpu: %s\n", e.what());
        return 7;
    }
}
