// fft_hip.cpp — Level 2 of the MI355X drop-in, to be added to the reference's src/ (integration/level2.patch
// adds it to meson.build and the hooks that call it).  Everything in this file is new code of this
// repository; it uses the reference's classes through their public/protected interface only.
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: the reference needs boost, websocketpp, toml++ and glaze, none of
// which is installed here; hip_fanout.h and hip_fft.h, which carry the logic, are compile-checked by
// tests/test_abi_host.py.
//
// What changes against the CPU path (citations: reference tree):
//   * fft_task (src/fft.cpp:47-105): the raw half-frame read from stdin goes to HBM as it is
//     (src/samplereader.cpp:29-70 runs inside the first FFT pass), asynchronously while the GPU works on
//     the previous frame; one call transforms the frame and serves every client;
//   * signal_loop / waterfall_loop (src/websocket.cpp:156-185,207-236) still post one task per client, but
//     the task only fetches the client's result and hands it to the encoder - the tail of
//     AudioClient::send_audio (src/signal.cpp:277-296) and WaterfallClient::send_waterfall
//     (src/waterfall.cpp:44-51).
#ifdef PSDR_HIP
#include "hip_fanout.h"
#include "spectrumserver.h"

#include <boost/asio/use_future.hpp>

void broadcast_server::fft_task_hip() {
    HipFanout &fo = *fanout;
    const size_t half_bytes = psdr_half_frame_bytes(fo.context());
    // three pinned buffers, like input_buffers[3] of src/fft.cpp:17-22: one being read, two in flight
    void *raw[3] = {fo.alloc_half(), fo.alloc_half(), fo.alloc_half()};
    FileSampleReader in(stdin);  // the converter in `reader` is bypassed: the GPU converts
    const int skip_num = std::max(1, (int)floor(((float)sps / fft_size) / 10.) * 2);  // src/fft.cpp:33
    std::vector<std::future<void>> signal_futures, waterfall_futures;
    auto &io_service = m_server.get_io_service();
    uint64_t half = 0;
    in.read(raw[0], (int)half_bytes);
    fo.push_half(raw[0]);
    half++;
    while (running) {
        void *buf = raw[half % 3];
        if (half >= 3) psdr_ring_wait(fo.context(), half - 3);  // the copy that last used this buffer
        in.read(buf, (int)half_bytes);                          // blocks at the receiver's sample rate
        fo.push_half(buf);                                      // H2D on the copy stream
        half++;
        for (auto &f : signal_futures) f.wait();                // src/fft.cpp:82-88
        for (auto &f : waterfall_futures) f.wait();
        signal_futures.clear();
        waterfall_futures.clear();
        fo.process_frame(frame_num);                            // FFT + pyramid + all clients, asynchronous
        {
            std::scoped_lock lg(signal_slice_mtx);
            for (auto &[slice, client] : signal_slices) {
                if (m_server.get_con_from_hdl(client->hdl)->get_buffered_amount() > 50000) continue;
                signal_futures.emplace_back(io_service.post(
                    boost::asio::use_future(std::bind(&AudioClient::send_audio_hip, client, fanout.get(), frame_num))));
            }
        }
        if (frame_num % skip_num == 0) {
            for (int i = 0; i < downsample_levels; i++) {
                std::scoped_lock lg(waterfall_slice_mtx[i]);
                for (auto &[slice, client] : waterfall_slices[i]) {
                    if (m_server.get_con_from_hdl(client->hdl)->get_buffered_amount() > 50000) continue;
                    waterfall_futures.emplace_back(io_service.post(boost::asio::use_future(
                        std::bind(&WaterfallClient::send_waterfall_hip, client, fanout.get(), frame_num))));
                }
            }
        }
        frame_num++;
    }
}

// the tail of send_audio (src/signal.cpp:277-296) on results that are already demodulated
void AudioClient::send_audio_hip(HipFanout *fo, size_t frame_num) {
    try {
        float pwr = 0;
        const bool chain_on_gpu = fo->post_chain();
        if (!fo->fetch_audio(psdr_id, audio_real.data(), chain_on_gpu ? audio_real_int16.data() : nullptr, &pwr))
            return;  // the NaN guard dropped this frame (src/signal.cpp:266-271)
        if (!chain_on_gpu) {
            dc.removeDC(audio_real.data(), audio_fft_size / 2);
            agc.process(audio_real.data(), audio_fft_size / 2);
            dsp_float_to_int16(audio_real.data(), audio_real_int16.data(), 65536 / 4, audio_fft_size / 2);
        }
        encoder->set_data(frame_num, audio_l, audio_mid, audio_r, pwr);
        encoder->process(audio_real_int16.data(), audio_fft_size / 2);
    } catch (const std::exception &) {
    }
}

void WaterfallClient::send_waterfall_hip(HipFanout *fo, size_t frame_num) {
    try {
        std::vector<int8_t> row;
        int ll = 0, rl = 0;
        if (fo->fetch_waterfall(psdr_id, row, &ll, &rl)) waterfall_encoder->send(row.data(), row.size(), frame_num, ll, rl);
    } catch (...) {
    }
}
#endif
