// forward.hip - the frame loop of broadcast_server::fft_task (src/fft.cpp:47-105) for a batch of frames: the two FFT
// passes, the pyramid levels above the tile (seam, column tail, pyramid tail), then what reads a finished batch without
// demodulating it: spectrum / pyramid read-back in the reference's order, the band layout of multi-GPU sharding and
// the waterfall clients (src/waterfall.cpp, src/websocket.cpp:207-236).
#include "ctx.h"
#include "epilogue.h"
#include "fft_pass.h"

namespace psdr {

// stamp slot of the next launch of pass `which` (nullptr: mode 2 off or the ring is full)
static unsigned long long *next_kclk(psdr_ctx *c, int which) {
    if (!c->kclock || !c->d_kclk || c->kclk_pos[which] >= psdr_ctx::KCLK_SLOTS) return nullptr;
    return c->d_kclk + ((size_t)which * psdr_ctx::KCLK_SLOTS + c->kclk_pos[which]++) * 2;
}
// counters for the next launch of pass `which` on stream st
static int next_tickets(psdr_ctx *c, int which, hipStream_t st, unsigned **out) {
    unsigned &pos = c->ticket_pos[which];
    const unsigned slot = pos % TICKET_SLOTS;
    if (slot % (TICKET_SLOTS / 2) == 0 && pos >= TICKET_SLOTS / 2) {
        // entering a half of the ring: its counters were last used TICKET_SLOTS/2 launches ago on
        // this same stream, so clearing them here is ordered after those launches
        HIPCHK(hipMemsetAsync(c->d_tickets[which] + (size_t)slot * 8, 0, (TICKET_SLOTS / 2) * 8 * sizeof(unsigned), st));
    }
    *out = c->d_tickets[which] + (size_t)slot * 8;
    pos++;
    return PSDR_OK;
}
// Tiles of one frame a work-group walks in a chain: long chains carry the mirror-side octets in LDS
// (nothing extra in HBM), short chains give the persistent grid enough independent segments.  Aim
// for about two segments per work-group: the pass is bound by memory, not by balance (256 frames of
// 2^21 points, same box: 1012-1023 us with 16 segments per work-group, 990-1030 with 8, 4 or 2), and
// every segment costs one seam (k_real_seam: 177 / 89 / 52 / 31 us at 4 / 8 / 16 / 32 tiles per segment,
// run beside the next batch's pass 1, which it slows).
int real_seg_len(const psdr_ctx *c, int nframes) {
    const int G = c->M1 / c->T2;  // tiles of a frame
    if (c->seg_len_env > 0) {
        int sl = 1;
        while (sl * 2 <= c->seg_len_env && sl * 2 <= G) sl *= 2;
        return sl;
    }
    const int W = std::max(c->num_cus, 1);
    const long long tiles = (long long)G * nframes;  // of the batch
    auto pow2_floor = [&](long long v) {
        int sl = 1;
        while (sl * 2 <= v && sl * 2 <= G) sl *= 2;
        return sl;
    };
    const long long want = tiles / (2LL * W);
    // What the measurements say (cfg3's stream, tools/ab_small_batches.sh, profiles/r04b_uniform_segments_small_batches.jsonl):
    // up to 16 tiles per work-group ONE static segment each, rounded up to a power of two (32 frames: 256 segments of 8
    // tiles 151 GS/s, 512 of 4 - the old choice - 140; 48 frames: 192 of 16 tiles 155, 768 of 4 141); two static segments per
    // work-group when that comes out even (128, 256 frames of 2^21 points: nothing to balance, nothing drawn); else five
    // to ten segments per work-group, so that the ticket counter has something to balance with - a work-group draws two
    // tickets ahead, and with three or four segments each half the chip ends up with one more than the other half (192
    // frames: 16-tile segments 158 GS/s, 8-tile segments 168; 384 frames: 162 / 182).
    const long long per_wg = (tiles + W - 1) / W;
    if (per_wg <= 16) {
        int sl = 1;
        while (sl < per_wg && sl * 2 <= G) sl *= 2;
        return sl;
    }
    if (2LL * W * want == tiles && (want & (want - 1)) == 0 && want <= G) return (int)want;
    return pow2_floor(tiles / (5LL * W));
}
// Chain segments of a batch (k_fft_pass2_real, fft_pass.h).  Two forms:
//  * uniform segments of real_seg_len() tiles, frame-major, every segment's first tile WITHOUT a carry-in (it leaves its
//    partial octets in seamP, k_real_seam completes them): small batches, PSDR_SEG_LEN, static tile hand-out;
//  * hand-off (batches of more than one frame per work-group - seg_plan_counts): a frame is cut into segments of G/4, G/4, G/4, G/8, ... 2,
//    1, 1 tiles from the top, handed out LEVEL-major - all frames' top segments first, then all second segments, ... - by
//    the ticket counter alone.  Only the top segment of a frame has no carry-in (the ring closes through tile 0's row
//    M1/2: one seam per frame, as with whole-frame segments); every other segment reads the carried row its predecessor -
//    handed out nframes tickets earlier, i.e. finished about a round of segments ago - left in seamC behind its flag.
//    What it buys: the work-groups of a launch differ by +-3.5 % in speed (the even XCDs are ~3 % slower: tools/trace_real.py),
//    and with two whole-frame segments each nothing balanced that - the launch ended with its slowest work-group, 3.4 %
//    after the median one.  With tickets and segments that shrink to single tiles the spread at the end is one tile, at
//    a cost of 8 KiB of traffic per hand-off (a seam: 92 KiB and a record written twice).
// tiles per segment from the top of a frame: G/4, G/4, G/4, G/8, ... 2, 1, 1
static std::vector<int> seg_plan_lens(const psdr_ctx *c) {
    const int G = c->M1 / c->T2;  // tiles of a frame
    std::vector<int> lens = {G / 4, G / 4, G / 4};
    for (int l = G / 8; l >= 1; l >>= 1) lens.push_back(l);
    lens.push_back(1);
    return lens;
}
void seg_plan_counts(const psdr_ctx *c, int nframes, unsigned *nsegs, unsigned *nseam, bool *handoff) {
    const int G = c->M1 / c->T2;  // tiles of a frame
    // (PSDR_SEG_LEN=n is the way back to uniform segments)
    // Which batches take the hand-off plan: those of more than one frame per work-group (cfg3's stream at 258 / 288 / 320 /
    // 384 / 448 / 512 frames: +11 / +29 / +25 / +12 / +6 / +1.5 % over the uniform segments of round 3; against the uniform
    // segments real_seg_len() picks now it is level at 320 and 384 frames and 1 - 5 % ahead at 512).  Up to one frame per
    // work-group a segment's predecessor is less than a segment ahead, most hand-offs end as seams, and well-chosen uniform
    // segments are 3 - 9 % faster (profiles/r04b_handoff_vs_uniform_by_batch.jsonl, r04b_uniform_segments_small_batches.jsonl).
    const int W = std::max(c->num_cus, 1);
    const bool want = nframes >= W + 1;
    const bool ho = c->seg_len_env <= 0 && G >= 16 && want;
    if (ho) {
        const int levels = (int)seg_plan_lens(c).size();
        *nsegs = (unsigned)(levels * nframes);
        *nseam = *nsegs;  // (any segment may fall back to a seam: seamP has room for all of them)
    } else {
        *nsegs = *nseam = (unsigned)(nframes * (G / real_seg_len(c, nframes)));
    }
    *handoff = ho;
}
int seg_plan(psdr_ctx *c, int nframes, const psdr_ctx::SegPlan **out) {
    for (const auto &sp : c->seg_plans)
        if (sp.nframes == nframes) {
            *out = &sp;
            return PSDR_OK;
        }
    psdr_ctx::SegPlan sp;
    sp.nframes = nframes;
    seg_plan_counts(c, nframes, &sp.nsegs, &sp.nseam, &sp.handoff);
    const int G = c->M1 / c->T2;  // tiles of a frame
    std::vector<uint4> tab(sp.nsegs);
    if (sp.handoff) {
        const std::vector<int> lens = seg_plan_lens(c);
        const int levels = (int)lens.size();
        // Level-major, frames in order: a segment's predecessor - the same frame one level up - is handed out nframes >=
        // 2 * grid indices earlier, i.e. about two segments of every work-group earlier.
        int g = G - 1;
        for (int lv = 0; lv < levels; lv++) {
            for (int f = 0; f < nframes; f++) {
                const unsigned above = (unsigned)((lv ? lv - 1 : levels - 1) * nframes + f);  // (the top segment's: the bottom one)
                tab[(size_t)lv * nframes + f] = make_uint4((unsigned)f, (unsigned)g | ((unsigned)lens[lv] << 16), above,
                                                           lv ? (unsigned)PSDR_SEG_CARRY_MEM : 0u);
            }
            g -= lens[lv];
        }
    } else {
        const int SL = real_seg_len(c, nframes), S = G / SL;
        for (int f = 0; f < nframes; f++)
            for (int si = 0; si < S; si++)
                tab[(size_t)f * S + si] = make_uint4((unsigned)f, (unsigned)((si + 1) * SL - 1) | ((unsigned)SL << 16),
                                                     (unsigned)(f * S + (si + 1) % S), 0u);
    }
    if (sp.nseam > c->seam_cap || sp.nsegs > c->seg_cap)
        return fail(PSDR_ERR_STATE, "seam buffers too small (%u + %u segments, %zu + %zu allocated)", sp.nseam, sp.nsegs, c->seam_cap, c->seg_cap);
    HIPCHK(hipMalloc((void **)&sp.d_tab, tab.size() * sizeof(uint4)));
    HIPCHK(hipMemcpy(sp.d_tab, tab.data(), tab.size() * sizeof(uint4), hipMemcpyHostToDevice));  // (once per batch size)
    c->seg_plans.push_back(sp);
    *out = &c->seg_plans.back();
    return PSDR_OK;
}
void select_set(psdr_ctx *c, int set) {
    c->cur_set = set;
    c->d_spec = c->spec_pool[set];
    c->d_q = c->q_pool[set];
    c->d_qt = c->qt_pool[set];
    c->d_pscr[0] = c->pscr_pool[set][0];
    c->d_pscr[1] = c->pscr_pool[set][1];
    c->d_seamP = c->seam_pool[set][0];
    c->d_seamC = c->seam_pool[set][1];
}

// the pyramid levels above the tile for the batch process_frames has just transformed, on the side stream
static int enqueue_tails(psdr_ctx *c) {
    c->tails_pending = false;
    const psdr_ctx::SegPlan *plan = c->tails_plan;
    const int nframes = c->tails_nframes;
    if (plan) {  // fused real input: the mirror octets of the first tile of every chain segment without a carry-in (epilogue.h)
        SeamArgs sa{};
        sa.seamP = c->d_seamP;
        sa.seamC = c->d_seamC;
        sa.segtab = plan->d_tab;
        sa.segmark = c->d_segflag + (size_t)(1 + c->cur_set) * c->seg_cap;
        sa.epoch = c->seg_epoch;
        sa.L = c->M2;
        sa.cp = c->T2 / 2;
        sa.size_log2 = c->size_log2;
        sa.Qt = c->d_qt;
        sa.qt_stride = c->qt_stride;
        sa.Pscr = c->d_pscr[0];
        sa.p_stride = c->p_stride;
        ProfScope ps(c, K_SEAM, c->side);
        hipLaunchKernelGGL(k_real_seam, dim3(plan->handoff ? plan->nsegs : plan->nseam), dim3(256), 0, c->side, sa);
        HIPCHK(hipGetLastError());
    }
    // remaining pyramid levels from the partial level in scratch
    int lvl = c->LT;
    size_t len = c->R >> lvl;
    int cur = 0;
    // tile-major sums of a fused pass 2 (rows of 1024 outputs): one thread per output row takes the
    // levels inside a row (k_col_tail), the generic kernel the few above
    const int ng = (int)(len >> c->log2M2);  // groups per output row
    const bool col_tail = c->recmap.mapped && (c->M2 == 1024 || (c->M2 == 2048 && c->real_fused)) && c->recmap.l2gpt == 0 &&
                          (ng == 64 || ng == 128 || ng == 256);
    if (col_tail && lvl + 1 < c->levels) {
        ColTailArgs t{};
        t.Pin = c->d_pscr[0];
        t.in_stride = c->p_stride;
        t.mode = c->recmap.mapped;
        t.L = c->M2;
        t.l2L = c->log2M2;
        t.lvl_in = lvl;
        t.nlevels = c->levels;
        t.size_log2 = c->size_log2;
        t.Q = c->d_q;
        t.q_stride = c->q_stride;
        t.R = c->R;
        t.Pout = c->d_pscr[1];
        t.out_stride = c->p_stride;
        ProfScope ps(c, K_TAIL, c->side);
        const dim3 grid((unsigned)(c->M2 / 64), (unsigned)nframes);
        if (ng == 64)
            hipLaunchKernelGGL(k_col_tail<64>, grid, dim3(64), 0, c->side, t);
        else if (ng == 128)
            hipLaunchKernelGGL(k_col_tail<128>, grid, dim3(64), 0, c->side, t);
        else if (c->recmap.pair)  // quartet records side by side (2048-point rows): one 8-byte load per (tile, column)
            hipLaunchKernelGGL((k_col_tail<256, true>), grid, dim3(64), 0, c->side, t);
        else
            hipLaunchKernelGGL(k_col_tail<256>, grid, dim3(64), 0, c->side, t);
        HIPCHK(hipGetLastError());
        lvl += ilog2((size_t)ng);
        len = (size_t)c->M2;
        cur = 1;
    }
    const int lvl_mapped = col_tail ? -1 : c->LT;  // the level whose sums are still in RecMap order
    while (lvl + 1 < c->levels && len >= 2) {
        TailArgs t{};
        t.Pin = c->d_pscr[cur];
        t.in_stride = c->p_stride;
        t.len_in = len;
        t.lvl_in = lvl;
        t.nlevels = c->levels;
        t.size_log2 = c->size_log2;
        t.Q = c->d_q;
        t.q_stride = c->q_stride;
        t.R = c->R;
        t.Pout = c->d_pscr[cur ^ 1];
        t.out_stride = c->p_stride;
        t.map = c->recmap;
        if (lvl != lvl_mapped) t.map.mapped = 0;  // only pass 2's own output is tile-major
        ProfScope ps(c, K_TAIL, c->side);
        const unsigned nb = (unsigned)((len / 2 + 255) / 256);
        hipLaunchKernelGGL(k_pyramid_tail, dim3(nb, nframes), dim3(256), 0, c->side, t);
        HIPCHK(hipGetLastError());
        lvl += 7;
        len >>= 7;
        cur ^= 1;
    }
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_side_done, c->side));
        c->side_pending = true;
        HIPCHK(hipEventRecord(c->ev_set_done[c->cur_set], c->side));
        c->set_pending[c->cur_set] = true;
    }
    return PSDR_OK;
}

// forward FFT + power + int8 pyramid for nframes frames (src/fft.cpp:61-98 per frame)
int process_frames(psdr_ctx *c, const void *d_halves, int nframes, int fmt, hipEvent_t ev_raw_consumed) {
    // alternate the result set when the consumers run on their own stream
    // (banded spectrum: also on a caller's stream - the regions of batch b are read by the peers, asynchronously,
    // while batch b+1 is transformed)
    if (c->side != c->stream || c->nbands || c->alt_sets) select_set(c, c->cur_set ^ 1);
    const int cols = 1;
    const int sb = fmt <= PSDR_FMT_S8 ? 2 : (fmt <= PSDR_FMT_S16 ? 4 : 8);  // image bytes per sample
    const unsigned tiles1 = (unsigned)(c->M2 / (c->T1 * cols)), tiles2 = (unsigned)(c->M1 / c->T2);
    c->input_on_main = false;
    cf *Y = c->d_Y;
    Pass1Args a1{};
    a1.raw = d_halves;
    a1.Y = Y;
    a1.Wl = c->d_Wl1;
    a1.TB = c->d_TB;
    a1.yblk = (size_t)c->M1 * (c->T1 * cols);  // plain: one linear block per pass-1 tile
    a1.l2t2 = ilog2((size_t)c->T2);
    // fused real: pass-2-tile-major (fft_pass.h, "Y layout")
    a1.ytile = (size_t)c->M2 * c->T2;
    a1.ytl = (size_t)c->T2 * (c->T1 * cols);
    a1.yframe = c->M;
    a1.wdelta = c->wdelta;
    a1.M2 = c->M2;
    a1.log2M2 = c->log2M2;
    a1.fmt = fmt;
    a1.is_real = c->is_real ? 1 : 0;
    a1.rot = c->is_real ? 0 : 1;
    a1.trace = c->d_trace;
    a1.kclk = next_kclk(c, 0);
    a1.ymask = ~0u;
    // /N (src/fft_impl.cpp:29-31) - and the real untangle's 1/2 - as a power of two in the window weights: the fused
    // second passes store what their last stage leaves; the three-pass real path scales in k_untangle_real as before
    a1.yscale = !c->is_real ? 1.0f / (float)c->N : (c->real_fused ? 0.5f / (float)c->N : 1.0f);
    // (tuning builds only: a timing-only experiment with WRONG results - all frames of a launch share a few frames of Y)
    if (const char *e = psdr_tuning_env("PSDR_Y_ALIAS")) a1.ymask = (unsigned)atoi(e) - 1u;
    {
        int rc = next_tickets(c, 0, c->p1, &a1.tickets);
        if (rc) return rc;
    }
    a1.tiles_per_frame = tiles1;
    a1.total_slots = tiles1 * (unsigned)nframes;
    int rc = launch_pass1(c, c->M1, c->T1, sb, a1, a1.total_slots, c->real_fused);
    if (rc) return rc;
    if (ev_raw_consumed) HIPCHK(hipEventRecord(ev_raw_consumed, c->p1));  // pass 1 is the only reader of the raw halves

    Pass2Args a2{};
    a2.Y = Y;
    a2.Wl = c->d_Wl2;
    a2.M1 = c->M1;
    a2.log2M1 = c->log2M1;
    a2.TW = c->T1 * cols;
    a2.yblk = a1.yblk;
    a2.ytile = a1.ytile;
    a2.yjs = (size_t)c->T2 * (c->T1 * cols);
    a2.yframe = a1.yframe;
    a2.log2TW = ilog2((size_t)(c->T1 * cols));
    a2.inv_n = 1.0f / (float)c->N;
    a2.size_log2 = c->size_log2;
    a2.nlevels = c->levels;
    a2.Qt = c->d_qt;
    a2.qt_stride = c->qt_stride;
    a2.Pscr = c->d_pscr[0];
    a2.p_stride = c->p_stride;
    a2.trace = c->d_trace ? c->d_trace + 128 + 2304 : nullptr;
    a2.kclk = next_kclk(c, 1);
    a2.ymask = a1.ymask;
    {
        int rc2 = next_tickets(c, 1, c->stream, &a2.tickets);
        if (rc2) return rc2;
    }
    a2.tiles_per_frame = tiles2;
    a2.total_slots = tiles2 * (unsigned)nframes;
    // pass 2 overwrites this result set: its previous consumers (two batches ago) must be done
    if (c->set_pending[c->cur_set] && c->side != c->stream)
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_set_done[c->cur_set], 0));
    auto run_pass2 = [&](bool fused) -> int { return launch_pass2(c, c->M2, c->T2, fused, a2, a2.total_slots); };
    const psdr_ctx::SegPlan *plan = nullptr;  // fused real path: the seam kernel runs with the consumers
    if (!c->is_real) {
        a2.X = c->d_spec;
        a2.spec_stride = c->spec_stride;
        if (c->nbands) {
            a2.l2Lb = c->lay.l2Lb;
            a2.lbmask = (1 << c->lay.l2Lb) - 1;
            a2.Lw = c->lay.Lw;
            a2.band_stride = c->lay.band_stride;
            rc = launch_pass2_band(c, a2, a2.total_slots);
            if (rc) return rc;
            if (c->band_H > 0) {  // the first columns of band b+1 once more, behind band b's own
                const size_t n16 = (size_t)c->nbands * nframes * (c->M1 / 16) * c->band_H * 8;  // 16-byte pieces
                hipLaunchKernelGGL(k_band_halo, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, c->d_spec,
                                   c->spec_stride, c->lay, c->nbands, nframes, c->M1 / 16, c->band_H);
                HIPCHK(hipGetLastError());
            }
        } else {
            rc = run_pass2(true);
            if (rc) return rc;
        }
    } else if (c->real_fused) {
        a2.X = c->d_spec;
        a2.spec_stride = c->spec_stride;
        a2.UA = c->d_UA;
        a2.UB = c->d_UB;
        a2.UG = c->d_UG;
        a2.log2UB = c->log2UB;
        rc = seg_plan(c, nframes, &plan);
        if (rc) return rc;
        a2.segtab = plan->d_tab;
        a2.seamP = c->d_seamP;
        a2.seamC = c->d_seamC;
        a2.segflag = plan->handoff ? c->d_segflag : nullptr;
        a2.segmark = c->d_segflag + (size_t)(1 + c->cur_set) * c->seg_cap;  // (of this result set, like seamP / seamC)
        if (++c->seg_epoch == 0) {  // 2^32 launches: no flag or mark of the previous cycle may look current
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->side != c->stream) HIPCHK(hipStreamSynchronize(c->side));
            HIPCHK(hipMemset(c->d_segflag, 0, 3 * c->seg_cap * sizeof(unsigned)));
            c->seg_epoch = 1;
        }
        a2.epoch = c->seg_epoch;
        a2.total_slots = plan->nsegs;
        rc = launch_pass2_real(c, a2);
        if (rc) return rc;
    } else {
        a2.X = c->d_Z;
        a2.spec_stride = c->M;
        rc = run_pass2(false);
        if (rc) return rc;
        UntangleArgs u{};
        u.Z = c->d_Z;
        u.X = c->d_spec;
        u.spec_stride = c->spec_stride;
        u.M = c->M;
        u.TA = c->d_UA;
        u.TB = c->d_UB;
        u.log2B = c->log2UB;
        u.inv_n = 1.0f / (float)c->N;
        u.size_log2 = c->size_log2;
        u.nlevels = c->levels;
        u.Q = c->d_q;
        u.q_stride = c->q_stride;
        u.Pscr = c->d_pscr[0];
        u.p_stride = c->p_stride;
        ProfScope ps(c, K_UNTANGLE);
        const unsigned nb = (unsigned)((c->M / 8 + 255) / 256);
        hipLaunchKernelGGL(k_untangle_real, dim3(nb, nframes), dim3(256), 0, c->stream, u);
        HIPCHK(hipGetLastError());
    }
    // consumers of the finished batch go to the side stream
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_fft_done, c->stream));
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_fft_done, 0));
    }
    // The pyramid levels above the tile (seam, column tail, generic tail) go FIRST on the side stream, the demodulation behind
    // them.  Round 6 measured the other order (tails enqueued behind the batch's demodulation, which then starts beside the
    // next batch's first pass) and a delayed start of all consumers (beside the next second pass): bench.py's cfg2 level,
    // cfg3 level or 1 % worse, cfg5's share 1.2 % worse - where a batch's consumers run is a zero-sum choice between the
    // two passes, what they cost is their instruction count (profiles/r06_consumer_placement.json).
    c->tails_plan = plan;
    c->tails_nframes = nframes;
    c->tails_pending = true;
    rc = enqueue_tails(c);
    if (rc) return rc;
    c->last_nframes = nframes;
    c->out_valid = c->q_valid = false;
    std::fill(c->q_untiled.begin(), c->q_untiled.end(), 0);
    return PSDR_OK;
}

}  // namespace psdr

static int ensure_level_major(psdr_ctx *c, int frame) {
    if (c->tiled_lt < 0 || c->q_untiled[frame]) return PSDR_OK;
    const size_t nrec = c->R / (size_t)c->tile_ch;
    hipLaunchKernelGGL(k_untile_q, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->side,
                       c->d_qt + (size_t)frame * c->qt_stride, c->d_q + (size_t)frame * c->q_stride, c->R,
                       c->tile_ch, c->tiled_lt, c->levels, c->recmap);
    HIPCHK(hipGetLastError());
    c->q_untiled[frame] = 1;
    return PSDR_OK;
}

// spectrum of `frame` to host in the reference's k order
static int copy_spectrum_k_order(psdr_ctx *c, int frame, cf *dst) {
    const cf *src = c->d_spec + (size_t)frame * c->spec_stride;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    if (c->lay.mode) {
        // the device keeps the frame in tile-major lines (SpecLayout): k order through one frame of staging
        hipLaunchKernelGGL(k_spec_k_order, dim3((unsigned)((c->M + 1 + 255) / 256)), dim3(256), 0, c->stream, src, c->d_Z,
                           c->M, c->is_real ? 1 : 0, c->lay);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dst, c->d_Z, (c->is_real ? c->N / 2 + 1 : c->N) * sizeof(cf), hipMemcpyDeviceToHost, c->stream));
    } else if (c->is_real) {
        HIPCHK(hipMemcpyAsync(dst, src, (c->N / 2 + 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
    } else {
        // client order c -> bin k = (c + N/2 + 1) mod N: two contiguous runs
        const size_t N = c->N, h = N / 2;
        HIPCHK(hipMemcpyAsync(dst, src + (h - 1), (h + 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
        HIPCHK(hipMemcpyAsync(dst + h + 1, src, (h - 1) * sizeof(cf), hipMemcpyDeviceToHost,
                              c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
extern "C" int psdr_get_output_buffer(psdr_ctx *c, float **out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (c->last_nframes > 0 && !c->out_valid) {
        int rc = copy_spectrum_k_order(c, 0, (cf *)c->h_out);
        if (rc) return rc;
        if (!c->is_real && c->cfg.additional_size > 0)  // wrap copy, src/fft.cpp:91-98
            memcpy((cf *)c->h_out + c->N, c->h_out, sizeof(cf) * (size_t)c->cfg.additional_size);
        c->out_valid = true;
    }
    *out = c->h_out;
    return PSDR_OK;
}
extern "C" int psdr_get_quantized_buffer(psdr_ctx *c, int8_t **out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (c->last_nframes > 0 && !c->q_valid) {
        {
            int rc = drain(c);
            if (rc) return rc;
            rc = ensure_level_major(c, 0);
            if (rc) return rc;
            rc = drain(c);
            if (rc) return rc;
        }
        HIPCHK(hipMemcpyAsync(c->h_q, c->d_q, c->q_len, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->q_valid = true;
    }
    *out = c->h_q;
    return PSDR_OK;
}
extern "C" int psdr_pack_band(psdr_ctx *c, int nframes, uint32_t first_bin, uint32_t nbins, float *d_out,
                              size_t out_stride_bins) {
    if (!c || !d_out) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->last_nframes < 1) return fail(PSDR_ERR_STATE, "pack_band before process_batch/execute");
    if (nframes < 1 || nframes > c->last_nframes)
        return fail(PSDR_ERR_INVALID, "nframes %d outside [1, %d] (the last batch)", nframes, c->last_nframes);
    const uint32_t R = (uint32_t)(c->is_real ? c->N / 2 : c->N);
    if (first_bin >= R || nbins < 1 || nbins > R) return fail(PSDR_ERR_INVALID, "band [%u, +%u) outside [0, %u)", first_bin, nbins, R);
    if (out_stride_bins < nbins) return fail(PSDR_ERR_INVALID, "output stride smaller than the band");
    HIPCHK(hipSetDevice(c->device));
    ProfScope ps(c, K_BAND, c->stream);
    hipLaunchKernelGGL(k_band_pack, dim3((nbins + 255) / 256, nframes), dim3(256), 0, c->stream, c->d_spec, c->spec_stride,
                       c->lay, (int)R, (int)first_bin, (int)nbins, (cf *)d_out, out_stride_bins);
    HIPCHK(hipGetLastError());
    return PSDR_OK;
}
// ---- band sharding without the pack: pass 2 writes band regions ------------------------------------------------
extern "C" int psdr_set_band_layout(psdr_ctx *c, int nbands, uint32_t halo_bins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->is_real || c->lay.mode == 0 || c->lay.mode == 2 || (c->M1 != 1024 && c->M1 != 2048) || c->M2 != 1024)
        return fail(PSDR_ERR_UNSUPPORTED, "banded spectrum: 2^20- and 2^21-point IQ frames only (use psdr_pack_band)");
    if (nbands < 1 || nbands > 16 || (nbands & (nbands - 1))) return fail(PSDR_ERR_INVALID, "nbands %d: a power of two <= 16", nbands);
    if (halo_bins > (uint32_t)c->M) return fail(PSDR_ERR_INVALID, "halo of %u bins", halo_bins);
    const int H = (int)((halo_bins + (uint32_t)c->M1 - 1) >> c->log2M1), Lb = c->M2 / nbands, Lw = Lb + H;
    // k_band_halo repeats the first H columns of band b+1 behind band b: they must be band b+1's OWN columns (a halo
    // longer than a band would reach into band b+2's, which band b+1's region only holds as its own halo - written
    // by the same launch)
    if (H > Lb)
        return fail(PSDR_ERR_INVALID, "halo of %u bins = %d columns of %d bins exceeds a band of %d columns (%d bands)", halo_bins, H,
                    c->M1, Lb, nbands);
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipDeviceSynchronize());
    const size_t F = (size_t)c->max_batch, fs = (size_t)c->M1 * Lw;
    {  // the new buffers first: a failed allocation leaves the context as it was
        cf *fresh[2] = {nullptr, nullptr};
        for (int s = 0; s < 2; s++) {
            if (hipMalloc((void **)&fresh[s], (size_t)nbands * F * fs * sizeof(cf)) != hipSuccess ||
                hipMemset(fresh[s], 0, (size_t)nbands * F * fs * sizeof(cf)) != hipSuccess) {
                (void)hipGetLastError();
                for (int t = 0; t <= s; t++)
                    if (fresh[t]) (void)hipFree(fresh[t]);
                return fail(PSDR_ERR_NOMEM, "banded spectrum: %zu bytes per result set", (size_t)nbands * F * fs * sizeof(cf));
            }
        }
        for (int s = 0; s < 2; s++) {
            if (c->spec_pool[s]) (void)hipFree(c->spec_pool[s]);
            c->spec_pool[s] = fresh[s];
        }
    }
    c->nbands = nbands;
    c->band_H = H;
    c->spec_stride = fs;  // frame to frame INSIDE a region
    c->lay.mode = 3;
    c->lay.l2Lb = ilog2((size_t)Lb);
    c->lay.Lw = Lw;
    c->lay.band_stride = F * fs;
    c->lay.c2_0 = 0;
    c->set_pending[0] = c->set_pending[1] = false;
    select_set(c, c->cur_set);
    c->last_nframes = 0;
    c->out_valid = false;
    return PSDR_OK;
}
extern "C" int psdr_band_region(psdr_ctx *c, int band, const float **d_region, size_t *frame_stride_bins, uint32_t *first_bin,
                                uint32_t *nbins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (!c->nbands) return fail(PSDR_ERR_STATE, "psdr_set_band_layout() first");
    if (band < 0 || band >= c->nbands) return fail(PSDR_ERR_INVALID, "band %d outside [0, %d)", band, c->nbands);
    if (d_region) *d_region = (const float *)(c->d_spec + (size_t)band * c->lay.band_stride);
    if (frame_stride_bins) *frame_stride_bins = c->spec_stride;
    if (first_bin) *first_bin = (uint32_t)((size_t)band << (c->lay.l2Lb + c->log2M1));
    if (nbins) *nbins = (uint32_t)c->spec_stride;
    return PSDR_OK;
}
// ---- waterfall ---------------------------------------------------------------------------
static int check_wslot(psdr_ctx *c, int id) {
    if (id < 0 || id >= (int)c->wslots.size() || !c->wslots[id].active)
        return fail(PSDR_ERR_INVALID, "no waterfall client with id %d", id);
    return PSDR_OK;
}
extern "C" int psdr_waterfall_add(psdr_ctx *c, int *id_out) {
    if (!c || !id_out) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    for (size_t i = 0; i < c->wslots.size(); i++)
        if (!c->wslots[i].active) {
            WfSlot &s = c->wslots[i];
            s = WfSlot();
            s.active = true;
            // default = whole spectrum at the coarsest level (src/websocket.cpp:198)
            s.level = c->levels - 1;
            s.l = 0;
            s.r = std::min(c->min_waterfall_fft, (int)(c->R >> s.level));  // set_waterfall_range(levels-1, 0, min_waterfall_fft)
            *id_out = (int)i;
            return PSDR_OK;
        }
    return fail(PSDR_ERR_NOMEM, "all %zu waterfall client slots are in use", c->wslots.size());
}
extern "C" int psdr_waterfall_remove(psdr_ctx *c, int id) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    c->wslots[id].active = false;
    return PSDR_OK;
}
extern "C" int psdr_waterfall_set_range(psdr_ctx *c, int id, int level, int l, int r) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    if (level < 0 || level >= c->levels) return fail(PSDR_ERR_INVALID, "level %d out of range", level);
    const int len = (int)(c->R >> level);
    if (l < 0) l = 0;
    if (r > len) r = len;  // the reference forgets this bound (src/waterfall.cpp:55-58)
    if (l > r) return fail(PSDR_ERR_INVALID, "empty waterfall range");
    WfSlot &s = c->wslots[id];
    s.level = level;
    s.l = l;
    s.r = r;
    return PSDR_OK;
}
extern "C" int psdr_waterfall_on_window_message(psdr_ctx *c, int id, int new_l, int new_r,
                                                int *level_out, int *l_out, int *r_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    // src/waterfall.cpp:53-79
    if (new_l < 0 || new_r < 0 || new_l >= new_r) return fail(PSDR_ERR_INVALID, "window rejected");
    const int mwf = c->min_waterfall_fft;
    float new_l_f = (float)new_l, new_r_f = (float)new_r;
    int new_level = c->levels - 1;
    float best = (float)(mwf * 2);
    for (int i = 0; i < c->levels; i++) {
        const float send_size = std::fabs((new_r_f - new_l_f) - (float)mwf);
        if (send_size < best) {
            best = send_size;
            new_level = i;
            new_l = (int)std::round(new_l_f);
            new_r = (int)std::round(new_r_f);
        }
        new_l_f /= 2;
        new_r_f /= 2;
    }
    int rc = psdr_waterfall_set_range(c, id, new_level, new_l, new_r);
    if (rc) return rc;
    if (level_out) *level_out = new_level;
    if (l_out) *l_out = c->wslots[id].l;
    if (r_out) *r_out = c->wslots[id].r;
    return PSDR_OK;
}

extern "C" int psdr_waterfall_batch(psdr_ctx *c, uint64_t first_frame_num) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (c->last_nframes < 1) return fail(PSDR_ERR_STATE, "waterfall_batch before process_batch/execute");
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    const int ring = c->wf_ring.acquire();
    if (ring < 0) return fail(PSDR_ERR_HIP, "waterfall parameter ring: event wait failed");
    WfClient *h_wf = (WfClient *)c->wf_ring.host(ring);
    WfClient *d_wf = (WfClient *)c->wf_ring.dev(ring);
    int *h_sent = (int *)((unsigned char *)c->wf_ring.host(ring) + c->wf_sent_off);
    int *d_sent = (int *)((unsigned char *)c->wf_ring.dev(ring) + c->wf_sent_off);
    int nsent = 0;
    for (int f = 0; f < c->last_nframes; f++)
        if ((first_frame_num + (uint64_t)f) % (uint64_t)c->cfg.skip_num == 0) h_sent[nsent++] = f;
    size_t total = 0;
    int maxid = -1;
    for (size_t i = 0; i < c->wslots.size(); i++) {
        WfSlot &s = c->wslots[i];
        WfClient &w = h_wf[i];
        w.active = s.active ? 1 : 0;
        w.level = s.level;
        w.l = s.l;
        w.r = s.r;
        w.qoff = 0;
        for (int t = 0; t < s.level; t++) w.qoff += c->R >> t;
        w.out_off = total;
        s.out_off = total;
        s.nsent = s.active ? nsent : 0;
        s.b_level = s.level;
        s.b_l = s.l;
        s.b_r = s.r;
        if (s.active) {
            total += (size_t)nsent * (size_t)(s.r - s.l);
            total = (total + 15) & ~(size_t)15;
            maxid = (int)i;
        }
    }
    if (maxid < 0 || nsent == 0 || total == 0) return PSDR_OK;
    if (total > c->wfout_cap) {
        if (c->d_wfout) HIPCHK(hipFree(c->d_wfout));
        c->d_wfout = nullptr;
        HIPCHK(hipMalloc((void **)&c->d_wfout, total));
        c->wfout_cap = total;
    }
    {  // d_wfout exists once: a result fetch in flight (psdr_fetch_begin) reads it first
        int rc = fetch_guard_wait(c, c->side, c->guard_wf);
        if (rc) return rc;
        c->guard_wf = nullptr;
    }
    HIPCHK(hipMemcpyAsync(d_wf, h_wf, (size_t)(maxid + 1) * sizeof(WfClient), hipMemcpyHostToDevice,
                          c->side));
    HIPCHK(hipMemcpyAsync(d_sent, h_sent, (size_t)nsent * sizeof(int), hipMemcpyHostToDevice,
                          c->side));
    {
        ProfScope ps(c, K_WFALL, c->side);
        hipLaunchKernelGGL(k_waterfall_gather, dim3(maxid + 1, nsent), dim3(256), 0, c->side, c->d_q,
                           c->q_stride, c->d_qt, c->qt_stride, c->tiled_lt, c->tile_ch, c->recmap, d_wf, d_sent, nsent,
                           c->d_wfout);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(c->wf_ring.release(ring, c->side));
    if (c->side != c->stream) {
        HIPCHK(hipEventRecord(c->ev_side_done, c->side));
        c->side_pending = true;
        HIPCHK(hipEventRecord(c->ev_set_done[c->cur_set], c->side));
        c->set_pending[c->cur_set] = true;
    }
    return PSDR_OK;
}
extern "C" int psdr_read_waterfall(psdr_ctx *c, int id, int8_t *out, size_t out_cap, int *nsent_out, int *level_out,
                                   int *l_out, int *r_out) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = check_wslot(c, id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    const WfSlot &s = c->wslots[id];
    // the window of the BATCH, not the live one (it may have changed since)
    const size_t bytes = (size_t)s.nsent * (size_t)(s.b_r - s.b_l);
    if (nsent_out) *nsent_out = s.nsent;
    if (level_out) *level_out = s.b_level;
    if (l_out) *l_out = s.b_l;
    if (r_out) *r_out = s.b_r;
    if (bytes == 0 || !out) return PSDR_OK;
    if (bytes > out_cap) return fail(PSDR_ERR_INVALID, "output buffer too small (%zu > %zu)", bytes, out_cap);
    {
        int rc2 = drain(c);
        if (rc2) return rc2;
    }
    HIPCHK(hipMemcpyAsync(out, c->d_wfout + s.out_off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}

// ---- raw results ---------------------------------------------------------------------------
extern "C" int psdr_spectrum_device_ptr(psdr_ctx *c, int frame, const float **d_spec, size_t *nbins) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->max_batch) return fail(PSDR_ERR_INVALID, "frame %d out of range", frame);
    if (c->nbands) return fail(PSDR_ERR_UNSUPPORTED, "banded spectrum: a frame is not one piece (psdr_band_region)");
    if (d_spec) *d_spec = (const float *)(c->d_spec + (size_t)frame * c->spec_stride);
    if (nbins) *nbins = c->is_real ? c->N / 2 + 1 : c->N;
    return PSDR_OK;
}
extern "C" int psdr_quantized_device_ptr(psdr_ctx *c, int frame, const int8_t **d_q, size_t *nbytes) {
    if (!c) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->max_batch) return fail(PSDR_ERR_INVALID, "frame %d out of range", frame);
    {
        int rc = drain(c);
        if (rc) return rc;
        rc = ensure_level_major(c, frame);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    if (d_q) *d_q = c->d_q + (size_t)frame * c->q_stride;
    if (nbytes) *nbytes = c->q_len;
    return PSDR_OK;
}
extern "C" int psdr_read_spectrum(psdr_ctx *c, int frame, float *out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->last_nframes) return fail(PSDR_ERR_INVALID, "frame %d not in the last batch", frame);
    HIPCHK(hipSetDevice(c->device));
    return copy_spectrum_k_order(c, frame, (cf *)out);
}
extern "C" int psdr_read_quantized(psdr_ctx *c, int frame, int8_t *out) {
    if (!c || !out) return fail(PSDR_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->last_nframes) return fail(PSDR_ERR_INVALID, "frame %d not in the last batch", frame);
    HIPCHK(hipSetDevice(c->device));
    {
        int rc = drain(c);
        if (rc) return rc;
        rc = ensure_level_major(c, frame);
        if (rc) return rc;
        rc = drain(c);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(out, c->d_q + (size_t)frame * c->q_stride, c->q_len, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PSDR_OK;
}
