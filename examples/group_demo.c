/* group_demo.c - SURVEY 8e from plain C: ONE process drives every GPU of the node through psdr_group_* (include/psdr.h).
 * Device 0 owns the raw ring and the forward FFT; the audio clients are spread over all devices; every batch crosses
 * xGMI once through RCCL, which the library calls itself (librccl.so is dlopen()ed by psdr_group_create - this file links
 * against libpsdr_hip.so only).
 *
 *   gcc -O2 -Iinclude examples/group_demo.c -Lphantomsdr_amd -lpsdr_hip -lm -Wl,-rpath,$PWD/phantomsdr_amd -o group_demo
 *   ./group_demo <ndevices> <shard: 0 clients | 1 raw | 2 band> [outdir]
 *
 * A synthetic cs16 stream (complex tones on known bins) goes through 3 batches of 4 frames of 2^16 points; one USB client
 * per tone.  Prints every client's device and slice power; with outdir, dumps the raw stream and every client's audio
 * (client<i>.bin: [frames][n/2] float) for tests/test_gpu_group.py to compare with the oracle.  With one device the
 * communicator and the collectives are forced (PSDR_SHARD_FORCE_COMM): the same code path as on a node. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psdr.h"

#define CHK(call)                                                                \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != PSDR_OK) {                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, psdr_last_error()); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

enum { LOG2N = 16, N = 1 << LOG2N, F = 4, NB = 3, NCL = 8, NAUDIO = 248 };

int main(int argc, char **argv) {
    const int ndev = argc > 1 ? atoi(argv[1]) : 1;
    const int shard = argc > 2 ? atoi(argv[2]) : PSDR_SHARD_CLIENTS;
    const char *outdir = argc > 3 ? argv[3] : NULL;
    if (ndev < 1 || ndev > 16) return 1;
    int devices[16];
    for (int i = 0; i < ndev; i++) devices[i] = i;

    psdr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.fft_size = N;
    cfg.is_real = 0;
    cfg.downsample_levels = 7;
    cfg.additional_size = NAUDIO;
    cfg.audio_fft_size = NAUDIO;
    cfg.audio_rate = 12000;
    cfg.input_format = PSDR_FMT_S16;
    cfg.max_batch = F;
    cfg.max_clients = NCL;
    cfg.max_waterfall_clients = 1;
    cfg.skip_num = 1;
    psdr_group *g = NULL;
    CHK(psdr_group_create(&cfg, devices, ndev, shard | (ndev == 1 ? PSDR_SHARD_FORCE_COMM : 0), &g));

    /* the stream: NB*F + 1 half-frames of N/2 complex samples, one tone per client at bin k_i + 20.25 */
    const size_t nhalves = NB * F + 1, nsamp = nhalves * (N / 2);
    short *raw = (short *)malloc(nsamp * 2 * sizeof(short));
    int kbin[NCL];
    for (int i = 0; i < NCL; i++) kbin[i] = (int)((i + 0.5) * N / NCL) - N / 2;  /* signed FFT bin of the window's left edge */
    for (size_t s = 0; s < nsamp; s++) {
        double re = 0, im = 0;
        for (int i = 0; i < NCL; i++) {
            const double ph = 2.0 * M_PI * (kbin[i] + 20.25) * (double)s / N;
            re += 0.05 * cos(ph);
            im += 0.05 * sin(ph);
        }
        raw[2 * s] = (short)lrint(re * 32768.0);
        raw[2 * s + 1] = (short)lrint(im * 32768.0);
    }
    psdr_ctx *root = psdr_group_ctx(g, 0);
    void *d_raw = NULL;
    CHK(psdr_dev_alloc(root, nsamp * 2 * sizeof(short), &d_raw));
    CHK(psdr_memcpy_h2d(root, d_raw, raw, nsamp * 2 * sizeof(short)));

    /* clients in CLIENT coordinates: FFT bin k is c = (k - (N/2 + 1)) mod N (src/websocket.cpp:157-160) */
    int gid[NCL];
    for (int i = 0; i < NCL; i++) {
        const int c = ((kbin[i] - (N / 2 + 1)) % N + N) % N;
        CHK(psdr_group_client_add(g, c, (double)c, c + 60, PSDR_USB, &gid[i]));
    }
    FILE *fa[NCL];
    memset(fa, 0, sizeof fa);
    if (outdir) {
        char path[512];
        snprintf(path, sizeof path, "%s/raw.bin", outdir);
        FILE *fr = fopen(path, "wb");
        if (!fr) return 3;
        fwrite(raw, sizeof(short), nsamp * 2, fr);
        fclose(fr);
        for (int i = 0; i < NCL; i++) {
            snprintf(path, sizeof path, "%s/client%d.bin", outdir, i);
            fa[i] = fopen(path, "wb");
            if (!fa[i]) return 3;
        }
    }
    const size_t hb = psdr_half_frame_bytes(root);
    double pw_last[NCL];
    for (int b = 0; b < NB; b++) {
        CHK(psdr_group_step(g, (const char *)d_raw + (size_t)b * F * hb, F, (uint64_t)b * F));
        CHK(psdr_group_fetch(g));
        for (int i = 0; i < NCL; i++)
            for (int f = 0; f < F; f++) {
                const float *audio = NULL;
                float pwr = 0;
                int32_t nan = 0;
                CHK(psdr_group_fetched_audio(g, gid[i], f, &audio, &pwr, &nan, NULL));
                if (nan) return 4;
                pw_last[i] = pwr;
                if (fa[i]) fwrite(audio, sizeof(float), NAUDIO / 2, fa[i]);
            }
    }
    double link_bytes = 0, ex_ms = 0;
    CHK(psdr_group_link_stats(g, &link_bytes, &ex_ms));
    CHK(psdr_group_synchronize(g));
    for (int i = 0; i < NCL; i++) {
        /* a tone of amplitude A on a Hann-windowed frame: slice power ~ A^2 * 1.5 / 4 ... the exact figure is the oracle's
         * business; here: every client sees its tone (power far above the empty slices' zero) */
        printf("client %d on rank %d: slice power %.6g\n", i, psdr_group_client_rank(g, gid[i]), pw_last[i]);
        if (!(pw_last[i] > 1e-4)) return 5;
        if (fa[i]) fclose(fa[i]);
    }
    printf("group of %d device(s), sharding %d: %.0f bytes per link and step, exchange %.3f ms\n", psdr_group_size(g), shard, link_bytes, ex_ms);
    CHK(psdr_dev_free(root, d_raw));
    psdr_group_destroy(g);
    free(raw);
    puts("group demo ok");
    return 0;
}
