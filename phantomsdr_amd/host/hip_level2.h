// hip_level2.h — the bodies of the three functions integration/level2.patch adds to the reference's server,
// as templates over the server's OWN types:
//
//   broadcast_server::fft_task_hip()          replaces fft_task's loop            src/fft.cpp:47-105
//                                             + signal_loop / waterfall_loop      src/websocket.cpp:156-185, 207-236
//   AudioClient::send_audio_hip()             the tail of send_audio              src/signal.cpp:277-296
//   WaterfallClient::send_waterfall_hip()     send_waterfall                      src/waterfall.cpp:44-51
//
// integration/src/fft_hip.cpp instantiates them with the reference's classes (it is the only file that needs
// boost); tests/test_abi_host.py instantiates the very same templates with mock classes that expose exactly the
// members the reference's classes declare (src/signal.h:53-123, src/waterfall.h:7-33, src/client.h:83-118,
// src/spectrumserver.h:88-175, src/audio.h:23-38, src/waterfallcompression.h:18-32) and RUNS them against a
// scripted HipFanout stand-in, so an identifier the reference does not have cannot ship again (round 2 shipped
// `audio_l` / `audio_r`, locals of AudioClient::send_audio, as if they were members).
//
// Access: the functions read protected members of the client classes, so the patch befriends
// `psdr_level2::Access` in AudioClient, WaterfallClient and broadcast_server.
#ifndef PSDR_HIP_LEVEL2_H
#define PSDR_HIP_LEVEL2_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <future>
#include <mutex>
#include <vector>

namespace psdr_level2 {

struct Access {
    // ---- AudioClient::send_audio_hip(fo, frame_num) -------------------------------------------------------------
    // Everything before src/signal.cpp:277 has already happened on the GPU for all clients at once.  What is left is
    // the per-client tail: (DC blocker, AGC, int16 conversion unless the GPU ran them too,) the packet labels and the
    // encoder.  Labels exactly as the reference sends them (src/signal.cpp:104-105, 287): l = audio_l = l - l = 0,
    // m = audio_mid (absolute), r = audio_r = r - l.
    template <class AudioClientT, class Fanout>
    static void send_audio(AudioClientT &c, Fanout &fo, size_t frame_num) {
        try {
            typename Fanout::AudioFrame fr;
            // false: the NaN guard dropped this frame (src/signal.cpp:266-271), the client was paused for it (slow
            // socket, src/websocket.cpp:170-176), or it attached after the frame was demodulated - nothing is sent
            if (!fo.fetch_audio(c.psdr_id, &fr)) return;
            const int half = c.audio_fft_size / 2;
            if (fr.pcm) {
                std::copy(fr.pcm, fr.pcm + half, c.audio_real_int16.begin());
            } else {
                std::copy(fr.audio, fr.audio + half, c.audio_real.begin());
                c.dc.removeDC(c.audio_real.data(), half);                                        // :278
                c.agc.process(c.audio_real.data(), half);                                        // :281
                dsp_float_to_int16(c.audio_real.data(), c.audio_real_int16.data(), 65536 / 4, half);  // :283-284
            }
            // :104-105 with the window the samples were demodulated with (the client's own l / audio_mid / r may have
            // moved since the frame was transformed, or hold a window the GPU refused)
            const int audio_l = fr.l - fr.l, audio_r = fr.r - fr.l;
            c.encoder->set_data(frame_num, audio_l, fr.m, audio_r, fr.average_power);  // :287
            c.encoder->process(c.audio_real_int16.data(), half);                       // :291
        } catch (const std::exception &) {  // :295
        }
    }

    // ---- WaterfallClient::send_waterfall_hip(fo, frame_num) -------------------------------------------------------
    template <class WaterfallClientT, class Fanout>
    static void send_waterfall(WaterfallClientT &c, Fanout &fo, size_t frame_num) {
        try {
            std::vector<int8_t> row;
            int l_label = 0, r_label = 0;  // l << level, r << level of the window the row was gathered with (src/waterfall.cpp:47)
            if (fo.fetch_waterfall(c.psdr_id, row, &l_label, &r_label))
                c.waterfall_encoder->send(row.data(), row.size(), frame_num, l_label, r_label);
        } catch (...) {  // src/waterfall.cpp:48
        }
    }

    // ---- broadcast_server::fft_task_hip() -------------------------------------------------------------------------
    // raw: anything with `int read(void *, int)` (the reference's FileSampleReader on stdin: the per-format converter
    //      of src/samplereader.cpp:29-70 is bypassed, the GPU converts);
    // post(fn) -> std::future<void>: how the server runs a task on its pool (the reference:
    //      io_service.post(boost::asio::use_future(fn)), src/websocket.cpp:179-181);
    // backlog(hdl) -> size_t: bytes queued on the client's socket (get_con_from_hdl(hdl)->get_buffered_amount(),
    //      src/websocket.cpp:174-177).
    template <class ServerT, class RawReader, class Post, class Backlog>
    static void fft_task(ServerT &srv, RawReader &raw, Post post, Backlog backlog) {
        auto &fo = *srv.fanout;
        const int half_bytes = (int)psdr_half_frame_bytes(fo.context());
        // three pinned buffers, like input_buffers[3] of src/fft.cpp:17-22: one being read, two in flight
        void *bufs[3] = {fo.alloc_half(), fo.alloc_half(), fo.alloc_half()};
        const int skip_num = std::max(1, (int)std::floor(((float)srv.sps / srv.fft_size) / 10.) * 2);  // src/fft.cpp:33
        std::vector<std::future<void>> signal_futures, waterfall_futures;
        uint64_t half = 0;
        if (raw.read(bufs[0], half_bytes) != half_bytes) {
            for (void *b : bufs) fo.free_half(b);
            return;
        }
        fo.push_half(bufs[0]);
        half++;
        while (srv.running) {
            void *buf = bufs[half % 3];
            if (half >= 3) psdr_ring_wait(fo.context(), half - 3);  // the copy that last used this buffer
            if (raw.read(buf, half_bytes) != half_bytes) break;     // blocks at the receiver's sample rate
            fo.push_half(buf);                                      // H2D on the copy stream
            half++;
            {  // no users: skip the frame, frame_num stands still (src/fft.cpp:70-80)
                size_t users = srv.signal_slices.size();
                for (auto &lvl : srv.waterfall_slices) users += lvl.size();
                if (users == 0) continue;
            }
            for (auto &f : signal_futures) f.wait();                // src/fft.cpp:82-88
            for (auto &f : waterfall_futures) f.wait();
            signal_futures.clear();
            waterfall_futures.clear();
            const size_t frame_num = (size_t)srv.frame_num;
            // signal_loop decides per client and frame whether send_audio is called AT ALL (src/websocket.cpp:170-176):
            // a client with more than 50 kB queued is passed over, and since send_audio is what advances its overlap-add
            // tails, FM sample, DC blocker and AGC (src/signal.cpp:200-203, 273-284), all of that stands still.  On the
            // GPU the frame is demodulated for everybody at once, so the decision comes FIRST: slow clients are paused
            // for this frame, the others are the ones whose task is posted below.
            std::vector<decltype(srv.signal_slices.begin()->second)> to_send;
            {
                std::scoped_lock lg(srv.signal_slice_mtx);          // src/websocket.cpp:161
                to_send.reserve(srv.signal_slices.size());
                for (auto &[slice, client] : srv.signal_slices) {
                    const bool slow = backlog(client->hdl) > 50000;  // :174-177
                    fo.set_audio_paused(client->psdr_id, slow);
                    if (!slow) to_send.push_back(client);            // (keeps the client alive inside the task)
                }
            }
            fo.process_frame(frame_num);  // FFT + pyramid + all clients + one copy of their results to the host
            signal_futures.reserve(to_send.size());
            for (auto &cl : to_send)
                signal_futures.emplace_back(post([cl, &fo, frame_num] { cl->send_audio_hip(&fo, frame_num); }));
            if (frame_num % (size_t)skip_num == 0) {                // src/fft.cpp:101-103
                for (int i = 0; i < srv.downsample_levels; i++) {
                    std::scoped_lock lg(srv.waterfall_slice_mtx[i]);  // src/websocket.cpp:217
                    for (auto &[slice, client] : srv.waterfall_slices[i]) {
                        if (backlog(client->hdl) > 50000) continue;
                        auto cl = client;
                        waterfall_futures.emplace_back(post([cl, &fo, frame_num] { cl->send_waterfall_hip(&fo, frame_num); }));
                    }
                }
            }
            srv.frame_num++;                                         // src/fft.cpp:104
        }
        for (auto &f : signal_futures) f.wait();
        for (auto &f : waterfall_futures) f.wait();
        for (void *b : bufs) fo.free_half(b);  // src/fft.cpp:116-118
    }
};

}  // namespace psdr_level2

#endif
