// types.h - plain parameter structs shared by the host translation units and the kernels (no device code: every .hip
// of the library includes this; the kernel headers are included by exactly one translation unit each)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace psdr {

#define PSDR_MAX_STAGES 12

struct ClientParams {
    int l, r;       // [l, r) in client bin coordinates
    int m_floor;    // floor(audio_mid)
    int mode;       // psdr_mode
    int slot;       // persistent slot (state + output rows)
    int state_cur;  // which half of the double-buffered state is current
    int agc_reset;  // post chain: 1 = the demodulation changed since the last batch (AGC::reset),
                    // 2 = a new client took this slot (all chain state starts from zero)
    int paused;     // psdr_client_set_paused: listed BEHIND the batch's active clients, for the post chain only (its
                    // double-buffered streams must carry the client's history across the batch it sits out)
};

#ifndef PSDR_PC_RING
#define PSDR_PC_RING 16  // register sets of 16 samples in the two recurrence kernels (even, >= 6)
#endif
// floats of padding behind every client's stream rows: the recurrence kernels read whole blocks ahead of the stream's end
#define PSDR_PC_PAD (16 * (PSDR_PC_RING + 4))

struct PostArgs {
    const ClientParams *clients;  // active clients (compact), .slot = row block
    int nact, nframes, max_batch, h;  // h = n/2 samples per frame
    int slots;
    int D, L;                         // DC delay, AGC look-ahead (samples)
    float desired, attack, release;   // AGC
    const float *audio;               // [slots][max_batch][h]
    const int *nan_flags;             // [slots][max_batch]
    int *fstart;                      // [slots][max_batch] stream offset of a frame, -1 = dropped
    int *len;                         // [slots] samples of this batch's stream
    size_t px, pv;                    // row pitches (floats per client) of X/M1 and of V1/P/S, multiples of 4
    int vo;                           // V1 only: leading pad so that its NEW rows (from row L-1) are 16-byte aligned
    float *X;                         // [slots][px]: demodulated audio, rows < D history          (D + T + pad)
    float *M1;                        // [slots][px]: first moving average, rows < D history
    float *V1;                        // [slots][pv]: DC-blocked stream, rows < L-1 history         (L-1 + T + pad)
    float *V1n;                       // the NEXT batch's V1 (double-buffered: k_pc_history moves the tail there)
    int hist_sel;                     // k_pc_history: 0 = X and M1, 1 = V1 -> V1n
    float *P, *S;                     // like V1: prefix / suffix maxima; then S = w_t, P = g_t
    int32_t *pcm;                     // [slots][max_batch][h]
    // carried state
    float *dc_s1, *dc_s2;             // [slots] running sums
    float *agc_gain;
    int *agc_n0;  // samples pushed since the last reset, saturating at L
    int ma_fused;  // k_pc_ma2 keeps M1's history itself (k_pc_history leaves M1 alone)
};

struct WfClient {
    int level, l, r;
    int active;
    size_t qoff;     // byte offset of `level` inside a frame's level-major int8 buffer
    size_t out_off;  // byte offset of this client's output block
};

}  // namespace psdr
