// Mock of the parts of the reference's server that phantomsdr_amd/host/hip_level2.h touches.  Every class has
// exactly the member names, types and access the reference declares (cited per member); nothing the reference
// does not have may be added here - the point of the test is that hip_level2.h compiles against THESE names.
// Written from the reference's declarations, not copied: bodies are recording stubs.
#pragma once
#include <cstdint>
#include <cstdio>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using connection_hdl = std::weak_ptr<void>;  // websocketpp::connection_hdl (src/client.h:12)
enum conn_type { SIGNAL, WATERFALL, AUDIO, EVENTS, WATERFALL_RAW, SIGNAL_RAW, UNKNOWN };  // src/client.h:14-22
enum demodulation_mode { USB, LSB, AM, FM };                                              // src/client.h:43

class WaterfallClient;
class AudioClient;
typedef std::vector<std::multimap<std::pair<int, int>, std::shared_ptr<WaterfallClient>>> waterfall_slices_t;  // src/client.h:51-53
typedef std::deque<std::mutex> waterfall_mutexes_t;                                                            // :54
typedef std::multimap<std::pair<int, int>, std::shared_ptr<AudioClient>> signal_slices_t;                      // :55-56

struct Call {
    std::string what;
    uint64_t frame_num;
    int l, r;
    double m, pwr;
    std::vector<int32_t> data;
};
extern std::vector<Call> g_calls;
extern std::mutex g_calls_mtx;

// src/utils/dsp.h:20
void dsp_float_to_int16(float *arr, int32_t *output, float mult, size_t len);

template <typename T> struct DCBlocker {  // src/utils.h:139-169
    void removeDC(T *buf, size_t len);
};
struct AGC {  // src/utils/audioprocessing.h
    void process(float *arr, size_t len);
};

class AudioEncoder {  // src/audio.h:23-38
  public:
    void set_data(uint64_t frame_num, int l, double m, int r, double pwr);
    virtual int process(int32_t *data, size_t size) = 0;
    virtual ~AudioEncoder() {}

  protected:
    Call pending;
};
class WaterfallEncoder {  // src/waterfallcompression.h:18-32
  public:
    virtual int send(const void *buffer, size_t bytes, uint64_t frame_num, int l, int r) = 0;
    virtual ~WaterfallEncoder() {}
};

namespace psdr_level2 {
struct Access;
}
class MockFanout;

class Client {  // src/client.h:83-118
  public:
    conn_type type;
    std::string user_id;
    std::string unique_id;
    connection_hdl hdl;
    double audio_mid;
    int frame_num;
    bool mute;
    int l;
    int r;
};

class AudioClient : public Client {  // src/signal.h:53-123 + the members integration/level2.patch adds
  public:
    void send_audio_hip(MockFanout *fo, size_t frame_num);
    MockFanout *psdr_fo = nullptr;
    int psdr_id = -1;
    friend struct psdr_level2::Access;
    friend struct TestSetup;

  protected:
    demodulation_mode demodulation;
    bool is_real;
    int audio_fft_size;
    int fft_result_size;
    int audio_rate;
    std::vector<float> audio_real;          // (AlignedAllocator in the reference)
    std::vector<float> audio_real_prev;
    std::vector<int32_t> audio_real_int16;
    DCBlocker<float> dc;
    AGC agc;
    std::unique_ptr<AudioEncoder> encoder;
};

class WaterfallClient : public Client {  // src/waterfall.h:7-33 + the members the patch adds
  public:
    void send_waterfall_hip(MockFanout *fo, size_t frame_num);
    MockFanout *psdr_fo = nullptr;
    int psdr_id = -1;
    friend struct psdr_level2::Access;
    friend struct TestSetup;

  protected:
    int min_waterfall_fft;
    int level;
    std::unique_ptr<WaterfallEncoder> waterfall_encoder;
};

class broadcast_server {  // src/spectrumserver.h:88-175 (the members fft_task touches) + the patch's
  public:
    void fft_task_hip();
    friend struct psdr_level2::Access;
    friend struct TestSetup;

  private:
    int fft_size;
    int fft_result_size;
    int sps;
    bool is_real;
    int downsample_levels;
    bool running;
    int frame_num;
    signal_slices_t signal_slices;
    std::mutex signal_slice_mtx;
    waterfall_slices_t waterfall_slices;
    std::deque<std::mutex> waterfall_slice_mtx;
    std::unique_ptr<MockFanout> fanout;
};
