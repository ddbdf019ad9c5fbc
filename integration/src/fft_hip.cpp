// fft_hip.cpp — Level 2 of the MI355X drop-in, to be added to the reference's src/ (integration/level2.patch
// adds it to meson.build and the hooks that call it).  Everything in this file is new code of this
// repository; it uses the reference's classes through the members they declare.
//
// The logic lives in phantomsdr_amd/host/hip_level2.h as templates over the server's own types.  This file only
// instantiates them with the reference's classes and supplies the two things that need boost / websocketpp (how a
// task is posted to the server's pool, how a client's socket backlog is read).  The reference itself cannot be
// built in this repository's image (boost, websocketpp, toml++, glaze are not installed), so this translation unit
// is not compiled here; tests/test_abi_host.py compiles AND runs the same templates against mock classes that
// expose exactly the members the reference's classes declare.
//
// What changes against the CPU path (citations: reference tree):
//   * fft_task (src/fft.cpp:47-105): the raw half-frame read from stdin goes to HBM as it is
//     (src/samplereader.cpp:29-70 runs inside the first FFT pass), asynchronously while the GPU works on
//     the previous frame; one call transforms the frame, serves every client and copies all results to the host;
//   * signal_loop / waterfall_loop (src/websocket.cpp:156-185,207-236) still post one task per client, but
//     the task only hands the client's result to the encoder - the tail of AudioClient::send_audio
//     (src/signal.cpp:277-296) and WaterfallClient::send_waterfall (src/waterfall.cpp:44-51).
#ifdef PSDR_HIP
#include "spectrumserver.h"
#include "utils/dsp.h"  // dsp_float_to_int16: must be declared before the templates that call it

#include "hip_fanout.h"
#include "hip_level2.h"

#include <boost/asio/use_future.hpp>

void broadcast_server::fft_task_hip() {
    FileSampleReader raw(stdin);  // src/spectrumserver.cpp:346; the converter in `reader` is bypassed: the GPU converts
    auto &io_service = m_server.get_io_service();
    psdr_level2::Access::fft_task(
        *this, raw, [&](auto fn) { return io_service.post(boost::asio::use_future(fn)); },  // src/websocket.cpp:179-181
        [&](connection_hdl hdl) { return m_server.get_con_from_hdl(hdl)->get_buffered_amount(); });  // :174
}

void AudioClient::send_audio_hip(HipFanout *fo, size_t frame_num) { psdr_level2::Access::send_audio(*this, *fo, frame_num); }

void WaterfallClient::send_waterfall_hip(HipFanout *fo, size_t frame_num) {
    psdr_level2::Access::send_waterfall(*this, *fo, frame_num);
}
#endif
