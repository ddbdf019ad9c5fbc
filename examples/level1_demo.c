/* level1_demo.c — drives libpsdr_hip.so exactly like broadcast_server::fft_task drives the
 * FFT plug-in (src/fft.cpp:17-30,61-98), from plain C.  Also serves as the compile/link check
 * of include/psdr.h (tests/test_abi_host.py builds it without a GPU).
 *   gcc -Iinclude examples/level1_demo.c -Lphantomsdr_amd -lpsdr_hip -Wl,-rpath,$PWD/phantomsdr_amd */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psdr.h"

/* optional argv[1] = directory: the input half-frames and, per frame, the spectrum and the int8
 * pyramid are dumped there so a test can compare them with the CPU oracle */
static void dump(const char *dir, const char *name, int f, const void *p, size_t bytes) {
    char path[1024];
    if (!dir) return;
    snprintf(path, sizeof path, "%s/%s%d.bin", dir, name, f);
    FILE *fp = fopen(path, "wb");
    if (!fp || fwrite(p, 1, bytes, fp) != bytes) {
        fprintf(stderr, "cannot write %s\n", path);
        exit(5);
    }
    fclose(fp);
}

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : NULL;
    psdr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.fft_size = 1u << 16;
    cfg.is_real = 0;
    cfg.downsample_levels = 7;
    cfg.additional_size = 248;
    cfg.input_format = PSDR_FMT_F32;
    cfg.max_batch = 1;
    cfg.max_clients = 1;
    cfg.max_waterfall_clients = 1;
    cfg.skip_num = 1;
    psdr_ctx *ctx = NULL;
    if (psdr_create(&cfg, &ctx) != PSDR_OK) {
        fprintf(stderr, "psdr_create: %s\n", psdr_last_error());
        return 2;
    }
    const size_t N = cfg.fft_size;
    float *buf[3];
    for (int i = 0; i < 3; i++)
        if (psdr_host_alloc(ctx, N, &buf[i]) != PSDR_OK) return 3;
    /* a tone 1000.25 bins above DC */
    for (int h = 0; h < 3; h++)
        for (size_t i = 0; i < N / 2; i++) {
            double ph = 2 * M_PI * 1000.25 * (double)(h * (N / 2) + i) / (double)N;
            buf[h][2 * i] = (float)(0.01 * cos(ph));
            buf[h][2 * i + 1] = (float)(0.01 * sin(ph));
        }
    for (int h = 0; h < 3; h++) dump(dir, "half", h, buf[h], N * sizeof(float));
    for (int f = 0; f < 2; f++) {
        psdr_load_complex_input(ctx, buf[f], buf[f + 1]);
        if (psdr_execute(ctx) != PSDR_OK) {
            fprintf(stderr, "execute: %s\n", psdr_last_error());
            return 4;
        }
        float *X;
        int8_t *q;
        psdr_get_output_buffer(ctx, &X);
        psdr_get_quantized_buffer(ctx, &q);
        dump(dir, "spec", f, X, (N + (size_t)cfg.additional_size) * 2 * sizeof(float));
        {
            size_t qlen = 0;
            for (int i = 0; i < cfg.downsample_levels; i++) qlen += N >> i;
            dump(dir, "q", f, q, qlen);
        }
        size_t kmax = 0;
        for (size_t k = 1; k < N; k++)
            if (hypotf(X[2 * k], X[2 * k + 1]) > hypotf(X[2 * kmax], X[2 * kmax + 1])) kmax = k;
        printf("frame %d: peak bin %zu, |X| = %g, waterfall byte at DC+1000 = %d\n", f, kmax,
               hypotf(X[2 * kmax], X[2 * kmax + 1]), (int)q[N / 2 - 1 + 1000]);
    }
    for (int i = 0; i < 3; i++) psdr_host_free(ctx, buf[i]);
    psdr_destroy(ctx);
    return 0;
}
