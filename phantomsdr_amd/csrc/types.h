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

// floats of padding behind every slot's stream rows: the recurrence kernels' loader waves read whole blocks of 16 ahead of
// the stream's end (rings of 12 register sets: postchain.h PC_MA_RING, PC_GAIN_RING)
#define PSDR_PC_PAD (16 * 20)

struct PostArgs {
    const ClientParams *clients;  // this batch's list: the active clients, then the paused ones (empty streams)
    const int *slot_ci;           // [slots] index into `clients` of the slot's client, -1 = not listed in this batch
    int nact, nframes, max_batch, h;  // h = n/2 samples per frame
    int slots;
    int D, L;                         // DC delay, AGC look-ahead (samples)
    float desired, attack, release;   // AGC
    const float *audio;               // [slots][max_batch][h]
    const int *nan_flags;             // [slots][max_batch]
    int *fstart;                      // [slots][max_batch] stream offset of a frame, -1 = dropped
    int *len;                         // [slots] samples of this batch's stream
    // The streams are LANE-INTERLEAVED per group of 64 slots (round 5; postchain.h pc_at): sample t of slot s sits at
    // float ((s >> 6) * pitch + (t >> 2) * 4) * 64 + (s & 63) * 4 + (t & 3) - a wave whose lane l owns slot 64 g + l
    // reads four samples of each of its clients with ONE contiguous 1 KiB access.
    size_t px, pv;                    // pitches (floats per slot, multiples of 4) of X / M1 and of V1 / P / S
    int vo;                           // V1 / P / S: leading pad so that sample 0's row (row L-1) is a multiple of 4
    float *X, *Xn;                    // demodulated audio, rows < D history (D + T + pad); Xn: the NEXT batch's set
    float *M1, *M1n;                  // first moving average, rows < D history
    float *V1, *V1n;                  // DC-blocked stream, rows < L-1 history (L-1 + T + pad)
    float *P, *S;                     // like V1: P = prefix maxima of |V1| per row, then g_t; S = w_t (both at sample t's row vo + L-1 + t)
    float *SM;                        // [groups][sub-blocks][64] maxima of whole sub-blocks (look-ahead longer than one sub-block)
    int sb, nsub;                     // sub-block length (rows), sub-blocks per look-ahead block of L rows
    int32_t *pcm;                     // [slots][max_batch][h] (pcm16: the same rows as int16, in the front half of the buffer)
    int pcm16;                        // PSDR_OPT_POST_CHAIN_PCM16: the output kernels store int16 instead of int32
    // carried state
    float *dc_s1, *dc_s2;             // [slots] running sums
    float *agc_gain;
    int *agc_n0;  // samples pushed since the last reset, saturating at L
    int ma_fused;  // k_pc_ma2 keeps M1's history itself (k_pc_history leaves M1 alone)
    int lanes;     // k_pc_ma2 / k_pc_gain / k_pc_agc: slots per work-group (16, 32 or 64 lanes of its waves in use)
    // the AGC behind chunk maxima (k_pc_cm / k_pc_cscan / k_pc_agc, postchain.h): [groups of 64 slots][nch chunks of 16 floats][64]
    float *CM, *CP, *CS;  // maxima of |V1| per chunk; prefix / suffix maxima of CM inside blocks of L/16 - 1 chunks
    int nch;
    int *falive;          // [groups][max_batch][64] the k-th surviving frame of a slot's stream (k_pc_index)
    unsigned h_magic;     // ceil(2^32 / h): stream position / h by one multiplication
    int direct;           // k_pc_ma2 may read a work-group's new samples from `audio` itself (k_pc_gather4 then skips it)
    int32_t *pcm_dump;    // 16 bytes per thread of a k_pc_agc work-group: where its unconditional stores of row groups without a sample go
};

struct WfClient {
    int level, l, r;
    int active;
    size_t qoff;     // byte offset of `level` inside a frame's level-major int8 buffer
    size_t out_off;  // byte offset of this client's output block
};

}  // namespace psdr
