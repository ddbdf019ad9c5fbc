/* stream_demo.c — the whole path without the web server, from plain C over include/psdr.h:
 *
 *   stdin (raw samples, input.driver.format)  ->  pinned staging  ->  psdr_ring_write_async (copy stream)
 *     ->  psdr_process_ring (window + FFT + pyramid)  ->  psdr_demod_batch / psdr_waterfall_batch
 *     ->  the reference's packets on stdout (psdr_wire_*: hello JSON, audio CBOR, waterfall CBOR in a
 *         per-client zstd stream)
 *
 * i.e. broadcast_server::fft_task (src/fft.cpp:10-119) with the sample reader's double buffering
 * (src/samplereader.cpp:42-70), signal_loop / waterfall_loop (src/websocket.cpp:156-236) and the packet
 * framing of src/audio.cpp:17-36 / src/waterfallcompression.cpp:13-37, minus sockets and codecs: the
 * audio payload is the demodulated float32 frame itself (n/2 samples, what the reference hands to the DC
 * blocker at src/signal.cpp:278).  Clients are configured with the reference's own JSON command frames
 * (src/client.cpp:19-117).  SURVEY 8f-3 / 8f-4.
 *
 *   stream_demo <log2 N> <real 0|1> <u8|s8|u16|s16|f32|f64> <sps> <audio_sps> <frames per batch> \
 *       [--audio '{"cmd":"window","l":..,"r":..,"m":..}' '{"cmd":"demodulation","demodulation":"USB"}']... \
 *       [--waterfall '{"cmd":"window","l":..,"r":..}']...   < samples > records
 *
 * Output records: 1 byte kind ('H' hello JSON, 'A' audio CBOR, 'W' waterfall CBOR, 'Z' waterfall CBOR inside
 * the client's zstd stream), u32 client index, u32 length (little endian), payload.
 *   gcc -O2 -Iinclude examples/stream_demo.c -Lphantomsdr_amd -lpsdr_hip -lm -Wl,-rpath,$PWD/phantomsdr_amd */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psdr.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != PSDR_OK) {                                                        \
            fprintf(stderr, "%s: %d %s\n", #call, rc_, psdr_last_error());          \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

enum { MAXC = 64 };

static void put_record(char kind, uint32_t client, const void *p, uint32_t len) {
    unsigned char h[9] = {(unsigned char)kind,
                          (unsigned char)client, (unsigned char)(client >> 8), (unsigned char)(client >> 16), (unsigned char)(client >> 24),
                          (unsigned char)len,    (unsigned char)(len >> 8),    (unsigned char)(len >> 16),    (unsigned char)(len >> 24)};
    if (fwrite(h, 1, 9, stdout) != 9 || (len && fwrite(p, 1, len, stdout) != len)) {
        fprintf(stderr, "short write\n");
        exit(3);
    }
}

static int mode_of(const char *s) {
    return !strcmp(s, "USB") ? PSDR_USB : !strcmp(s, "LSB") ? PSDR_LSB : !strcmp(s, "AM") ? PSDR_AM : !strcmp(s, "FM") ? PSDR_FM : -1;
}

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s log2N real fmt sps audio_sps batch [--audio WINDOW_JSON DEMOD_JSON]... [--waterfall WINDOW_JSON]...\n", argv[0]);
        return 1;
    }
    const int log2n = atoi(argv[1]), is_real = atoi(argv[2]);
    static const char *fmts[] = {"u8", "s8", "u16", "s16", "f32", "f64"};
    int fmt = -1;
    for (int i = 0; i < 6; i++)
        if (!strcmp(argv[3], fmts[i])) fmt = i;
    const double sps = atof(argv[4]), audio_sps = atof(argv[5]);
    const int batch = atoi(argv[6]);
    if (fmt < 0 || log2n < 12 || log2n > 22 || batch < 1) return 1;

    /* derived parameters: src/spectrumserver.cpp:99-105,151,186-190, src/fft.cpp:33 */
    const uint32_t N = 1u << log2n, R = is_real ? N / 2 : N;
    const int waterfall_size = 1024;
    const int n = (int)(ceil(audio_sps * (double)N / sps / 4.0) * 4.0);
    int levels = 0;
    for (uint32_t cur = R; cur >= (uint32_t)waterfall_size; cur /= 2) levels++;
    int skip = (int)floorf(((float)sps / (float)N) / 10.0f) * 2;
    if (skip < 1) skip = 1;

    psdr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.fft_size = N;
    cfg.is_real = is_real;
    cfg.downsample_levels = levels;
    cfg.additional_size = n;
    cfg.audio_fft_size = n;
    cfg.audio_rate = (int)audio_sps;
    cfg.input_format = fmt;
    cfg.max_batch = batch;
    cfg.max_clients = MAXC;
    cfg.max_waterfall_clients = MAXC;
    cfg.skip_num = skip;
    cfg.waterfall_size = waterfall_size;
    psdr_ctx *ctx = NULL;
    CHECK(psdr_create(&cfg, &ctx));

    /* clients, configured through the reference's command frames */
    int aid[MAXC], wid[MAXC], na = 0, nw = 0;
    int al[MAXC], ar[MAXC];
    double am[MAXC];
    psdr_zstd *wz[MAXC];
    for (int i = 7; i < argc; i++) {
        psdr_command c;
        if (!strcmp(argv[i], "--audio") && i + 2 < argc && na < MAXC) {
            CHECK(psdr_client_add(ctx, &aid[na]));
            CHECK(psdr_wire_parse_command(argv[i + 2], strlen(argv[i + 2]), &c));
            if (c.cmd != PSDR_CMD_DEMODULATION || mode_of(c.text) < 0) return 1;
            CHECK(psdr_client_set_audio_demodulation(ctx, aid[na], mode_of(c.text)));
            CHECK(psdr_wire_parse_command(argv[i + 1], strlen(argv[i + 1]), &c));
            if (c.cmd != PSDR_CMD_WINDOW || !c.has_m) return 1;
            CHECK(psdr_client_on_window_message(ctx, aid[na], c.l, c.m, c.r));
            al[na] = c.l, am[na] = c.m, ar[na] = c.r;
            na++;
            i += 2;
        } else if (!strcmp(argv[i], "--waterfall") && i + 1 < argc && nw < MAXC) {
            CHECK(psdr_waterfall_add(ctx, &wid[nw]));
            CHECK(psdr_wire_parse_command(argv[i + 1], strlen(argv[i + 1]), &c));
            if (c.cmd != PSDR_CMD_WINDOW) return 1;
            CHECK(psdr_waterfall_on_window_message(ctx, wid[nw], c.l, c.r, NULL, NULL, NULL));
            wz[nw] = NULL;
            psdr_wire_zstd_create(&wz[nw]); /* no libzstd on the host: plain 'W' records */
            nw++;
            i += 1;
        } else {
            fprintf(stderr, "bad argument %s\n", argv[i]);
            return 1;
        }
    }

    /* the text frame every client receives first (src/websocket.cpp:42-66) */
    {
        psdr_hello h;
        memset(&h, 0, sizeof h);
        h.sps = sps, h.audio_max_sps = audio_sps, h.audio_max_fft = n, h.fft_size = N, h.fft_result_size = R;
        h.waterfall_size = waterfall_size, h.basefreq = 0, h.total_bandwidth = is_real ? sps / 2 : sps;
        h.default_frequency = 0, h.default_l = na ? al[0] : 0, h.default_m = na ? am[0] : 0, h.default_r = na ? ar[0] : 0;
        h.default_modulation = "USB", h.waterfall_compression = "zstd", h.audio_compression = "none";
        char js[1024];
        size_t len = 0;
        CHECK(psdr_wire_hello_json(&h, js, sizeof js, &len));
        put_record('H', 0, js, (uint32_t)len);
    }

    /* ingest ring in HBM + pinned staging of the same shape: half k sits in slot k % nh of both.  Three
     * batches long: batches start at multiples of `batch`, so none crosses the end of the ring, and the copies
     * of the next batch never have to wait for the transform of the current one */
    const size_t hb = psdr_half_frame_bytes(ctx);
    const int nh = 3 * batch;
    CHECK(psdr_ring_create(ctx, nh));
    float *staging = NULL;
    CHECK(psdr_host_alloc(ctx, (size_t)nh * hb / sizeof(float) + 1, &staging));
    const int h2 = n / 2;
    int8_t *rows = malloc((size_t)batch * R);
    const size_t pcap = psdr_wire_packet_bound((size_t)(R > (uint32_t)h2 * 4 ? R : (uint32_t)h2 * 4));
    uint8_t *pkt = malloc(pcap), *zbuf = malloc(psdr_wire_zstd_bound(pcap) + 64);

    uint64_t have = 0;      /* half-frames read so far */
    uint64_t frame = 0;     /* next frame to transform = halves frame, frame + 1 */
    int eof = 0;
    while (!eof) {
        /* read up to `batch` new halves (batch + 1 the first time): the copies of these halves run on
         * the copy stream while the previous batch is still being transformed and demodulated */
        const uint64_t want = frame + (uint64_t)batch + 1;
        while (have < want) {
            char *dst = (char *)staging + (size_t)(have % (uint64_t)nh) * hb;
            if (have >= (uint64_t)nh) CHECK(psdr_ring_wait(ctx, have - (uint64_t)nh)); /* the slot's previous copy has left the host */
            if (fread(dst, 1, hb, stdin) != hb) {
                eof = 1;
                break;
            }
            CHECK(psdr_ring_write_async(ctx, have, dst));
            have++;
        }
        const int nf = have > frame + 1 ? (int)(have - 1 - frame) : 0;
        if (nf == 0) break;
        CHECK(psdr_process_ring(ctx, frame, nf));
        if (na) CHECK(psdr_demod_batch(ctx, frame));
        if (nw) CHECK(psdr_waterfall_batch(ctx, frame));
        /* ONE copy of every client's results of the batch to the host (psdr_fetch_batch), then pointers into it */
        if (na) CHECK(psdr_fetch_batch(ctx));
        for (int c = 0; c < na; c++) {
            for (int f = 0; f < nf; f++) {
                const float *a = NULL;
                float p = 0;
                int32_t nanflag = 0;
                CHECK(psdr_fetched_audio(ctx, aid[c], f, &a, &p, &nanflag, NULL));
                if (nanflag) continue; /* the reference drops the frame (src/signal.cpp:266-271) */
                size_t len = 0;
                /* labels as AudioClient::send_audio sends them (src/signal.cpp:104-105, 287): l = audio_l = l - l = 0,
                 * m = audio_mid, r = audio_r = r - l */
                CHECK(psdr_wire_audio_packet(frame + (uint64_t)f, 0, am[c], ar[c] - al[c], p, a, (size_t)h2 * sizeof(float), pkt, pcap,
                                             &len));
                put_record('A', (uint32_t)c, pkt, (uint32_t)len);
            }
        }
        for (int c = 0; c < nw; c++) {
            int nsent = 0, level = 0, l = 0, r = 0;
            CHECK(psdr_read_waterfall(ctx, wid[c], rows, (size_t)batch * R, &nsent, &level, &l, &r));
            int si = 0;
            for (int f = 0; f < nf && si < nsent; f++) {
                if ((frame + (uint64_t)f) % (uint64_t)skip) continue;
                size_t len = 0, zl = 0;
                /* l, r labelled in level-0 bins (src/waterfall.cpp:47) */
                CHECK(psdr_wire_waterfall_packet(frame + (uint64_t)f, l << level, r << level, rows + (size_t)si * (size_t)(r - l),
                                                 (size_t)(r - l), pkt, pcap, &len));
                if (wz[c]) {
                    CHECK(psdr_wire_zstd_flush(wz[c], pkt, len, zbuf, psdr_wire_zstd_bound(pcap) + 64, &zl));
                    put_record('Z', (uint32_t)c, zbuf, (uint32_t)zl);
                } else {
                    put_record('W', (uint32_t)c, pkt, (uint32_t)len);
                }
                si++;
            }
        }
        frame += (uint64_t)nf;
    }
    fflush(stdout);
    fprintf(stderr, "stream_demo: %llu frames of %u points, %d audio + %d waterfall clients, n = %d, levels = %d, skip = %d\n",
            (unsigned long long)frame, N, na, nw, n, levels, skip);
    for (int c = 0; c < nw; c++) psdr_wire_zstd_destroy(wz[c]);
    psdr_host_free(ctx, staging);
    psdr_destroy(ctx);
    free(rows), free(pkt), free(zbuf);
    return 0;
}
