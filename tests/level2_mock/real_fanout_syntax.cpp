// -fsyntax-only: the same templates instantiated with the REAL HipFanout (phantomsdr_amd/host/hip_fanout.h) in place
// of the scripted one - every call hip_level2.h makes on a fan-out exists there with a matching signature.
#define MockFanout HipFanout
#include "hip_fanout.h"
#include "mock_reference.h"
#include "hip_level2.h"
void AudioClient::send_audio_hip(HipFanout *fo, size_t frame_num) { psdr_level2::Access::send_audio(*this, *fo, frame_num); }
void WaterfallClient::send_waterfall_hip(HipFanout *fo, size_t frame_num) { psdr_level2::Access::send_waterfall(*this, *fo, frame_num); }
struct Raw {
    int read(void *, int);
};
struct TestSetup {
    static void use(broadcast_server &srv, Raw &raw) {
        psdr_level2::Access::fft_task(
            srv, raw, [](auto fn) { return std::async(std::launch::async, fn); }, [](connection_hdl) -> size_t { return 0; });
    }
};
