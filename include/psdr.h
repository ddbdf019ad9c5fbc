/*
 * psdr.h — C-ABI of the MI355X-native spectrum-distributor DSP core (libpsdr_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of PhantomSDR (citations are
 * file:line under the reference tree):
 *
 *   Level 1  replaces the `class FFT` plug-in (src/fft.h:33-63; siblings FFTW
 *            src/fft_impl.cpp:80-183, cuFFT src/fft_cuda.cu).  A ~60-line
 *            `class hipFFT : public FFT` (phantomsdr_amd/host/hip_fft.h) forwards to it.
 *   Level 2  replaces the per-frame fan-out + per-client DSP:
 *            broadcast_server::signal_loop / waterfall_loop (src/websocket.cpp:156-185,
 *            207-236), AudioClient::send_audio up to the NaN guard
 *            (src/signal.cpp:102-275) and WaterfallClient::send_waterfall
 *            (src/waterfall.cpp:44-51), batched over frames and clients on the GPU.
 *
 * Conventions: plain pointers and sizes only; every function returns PSDR_OK (0) or a
 * negative psdr_status and never throws; psdr_last_error() gives the text for the
 * calling thread.  A context is single-producer (load/execute/process/demod from one
 * thread, like the reference's fft_thread, src/fft.cpp:14); client add/set/remove may
 * come from any thread (internal mutex = the reference's signal_slice_mtx,
 * src/signal.cpp:88).  The HIP library is the only implementation: there is no CPU
 * fallback behind this ABI.
 */
#ifndef PSDR_H
#define PSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psdr_ctx psdr_ctx;

typedef enum psdr_status {
    PSDR_OK = 0,
    PSDR_ERR_INVALID = -1,     /* bad argument / rejected range */
    PSDR_ERR_NO_DEVICE = -2,   /* no HIP device (cuFFT ctor throws, src/fft_cuda.cu:8-13) */
    PSDR_ERR_HIP = -3,         /* a HIP runtime call failed */
    PSDR_ERR_STATE = -4,       /* call order violated (e.g. execute before load) */
    PSDR_ERR_NOMEM = -5,
    PSDR_ERR_UNSUPPORTED = -6, /* size outside what the kernels are built for */
    PSDR_ERR_NO_DATA = -7      /* the client slot was not part of the batch whose results are asked for */
} psdr_status;

/* input.driver.format, src/spectrumserver.cpp:349-364 / src/samplereader.cpp:72-81 */
typedef enum psdr_format {
    PSDR_FMT_U8 = 0,
    PSDR_FMT_S8 = 1,
    PSDR_FMT_U16 = 2,
    PSDR_FMT_S16 = 3,
    PSDR_FMT_F32 = 4,
    PSDR_FMT_F64 = 5
} psdr_format;

/* demodulation_mode, src/client.h:43 */
typedef enum psdr_mode { PSDR_USB = 0, PSDR_LSB = 1, PSDR_AM = 2, PSDR_FM = 3 } psdr_mode;

typedef struct psdr_config {
    uint32_t struct_size;        /* = sizeof(psdr_config) */
    uint32_t fft_size;           /* N, power of two (FFT::FFT size, src/fft_impl.cpp:63) */
    int32_t is_real;             /* plan_r2c vs plan_c2c (src/fft.cpp:25-29) */
    int32_t downsample_levels;   /* src/spectrumserver.cpp:186-190 */
    int32_t brightness_offset;   /* src/fft_impl.cpp:69 */
    int32_t additional_size;     /* set_output_additional_size(), src/spectrumserver.cpp:214 */
    int32_t audio_fft_size;      /* n = ceil(audio_sps*N/sps/4)*4, src/websocket.cpp:133 */
    int32_t audio_rate;          /* audio_max_sps (AM carrier cutoff, AGC/DC) */
    int32_t input_format;        /* psdr_format of the raw ring used by psdr_process_batch */
    int32_t device;              /* HIP device ordinal */
    int32_t max_batch;           /* frames per psdr_process_batch call (>=1) */
    int32_t max_clients;         /* audio client slots */
    int32_t max_waterfall_clients;
    int32_t skip_num;            /* waterfall sent when frame_num % skip_num == 0 (src/fft.cpp:33,102) */
    int32_t waterfall_size;      /* min_waterfall_fft = input.waterfall_size (src/spectrumserver.cpp:56): default
                                    width of a new waterfall client (src/websocket.cpp:198) and target of the
                                    level search (src/waterfall.cpp:62-79).  0 = R >> (downsample_levels-1),
                                    which equals it whenever waterfall_size is a power of two */
} psdr_config;

const char *psdr_last_error(void);
const char *psdr_version(void);
/* ABI number of this header: bumped whenever a signature or a struct changes incompatibly.  A caller built against another
 * header should compare psdr_abi_version() (the library's) with the PSDR_ABI_VERSION it was compiled with.
 *   2 (library 0.2, round 4): psdr_group_client_set_audio_range(g, int *gid, ...), gid = (rank << 16) | slot
 *   3 (library 0.3, round 5/6): psdr_group_client_set_audio_range takes the gid BY VALUE and gids are stable handles;
 *     psdr_fetch_begin / _end / psdr_fetched_waterfall added (additions alone do not bump the number) */
#define PSDR_ABI_VERSION 3
int psdr_abi_version(void);

/* ---- lifetime -------------------------------------------------------------------- */
int psdr_create(const psdr_config *cfg, psdr_ctx **out);
void psdr_destroy(psdr_ctx *ctx);

/* ---- Level 1: the FFT plug-in (src/fft.h:33-63) ----------------------------------- */
/* FFT::malloc / FFT::free (src/fft.h:36-37; cuFFT twin src/fft_cuda.cu:22-28): pinned,
 * host-writable buffer of nfloats floats.  ctx may be NULL (the reference allocates its
 * half-frame buffers before planning, src/fft.cpp:17-29). */
int psdr_host_alloc(psdr_ctx *ctx, size_t nfloats, float **out);
int psdr_host_free(psdr_ctx *ctx, float *buf);
/* FFT::load_real_input / load_complex_input (src/fft.h:45-46, src/fft_impl.cpp:131-143):
 * a1 = older half-frame, a2 = newer, each N/2 samples (IQ: N floats each). */
int psdr_load_real_input(psdr_ctx *ctx, const float *a1, const float *a2);
int psdr_load_complex_input(psdr_ctx *ctx, const float *a1, const float *a2);
/* FFT::execute (src/fft.h:47, src/fft_impl.cpp:144-174): window, FFT, /N, power, int8
 * pyramid for the loaded frame.  Synchronous like the reference (src/fft_cuda.cu:175). */
int psdr_execute(psdr_ctx *ctx);
/* FFT::get_output_buffer (src/fft.h:43): host pointer, natural k order, N + additional
 * complex bins (IQ, the wrap copy of src/fft.cpp:91-98 already applied) or N/2+1 (real);
 * valid until the next execute/process call. */
int psdr_get_output_buffer(psdr_ctx *ctx, float **out);
/* FFT::get_quantized_buffer (src/fft.h:44): host pointer to the int8 pyramid, levels
 * back to back (level i at offset sum_{t<i} R>>t, src/websocket.cpp:233). */
int psdr_get_quantized_buffer(psdr_ctx *ctx, int8_t **out);

/* ---- device memory helpers (raw sample ring lives in HBM) ------------------------- */
int psdr_dev_alloc(psdr_ctx *ctx, size_t bytes, void **out);
int psdr_dev_free(psdr_ctx *ctx, void *p);
int psdr_memcpy_h2d(psdr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int psdr_memcpy_d2h(psdr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int psdr_synchronize(psdr_ctx *ctx);
/* bytes of one raw half-frame in cfg.input_format (N/2 samples, x2 components for IQ) */
size_t psdr_half_frame_bytes(const psdr_ctx *ctx);

/* ---- streaming ingest: the sample reader's double buffering (src/fft.cpp:56-67 reads half k+2
 * while frame (k, k+1) is transformed; src/samplereader.cpp:42-70) on the device ------------------
 * The context owns a ring of `nhalves` raw half-frames in HBM.  psdr_ring_write_async() copies one
 * half-frame from (pinned) host memory into slot `half_index % nhalves` on a dedicated COPY stream and
 * returns at once; psdr_process_ring() transforms frames first_half .. first_half+nframes-1 (frame f
 * = halves f, f+1; the frames of one call must not cross the end of the ring - first_half % nhalves +
 * nframes <= nhalves, the last frame's second half may be slot 0 again: size the ring as a multiple of the
 * batch) after waiting, on the device, for the copies
 * of exactly the halves it reads - so the PCIe transfer of later halves overlaps the transform of
 * earlier ones.  A slot may be rewritten as soon as the psdr_process_ring() call that read it has
 * been issued (the copy waits for that batch on the device).  The source buffer must stay valid
 * until psdr_ring_wait(ctx, half_index) (or any synchronising call) returns. */
int psdr_ring_create(psdr_ctx *ctx, int nhalves);
int psdr_ring_write_async(psdr_ctx *ctx, uint64_t half_index, const void *host_half);
int psdr_ring_wait(psdr_ctx *ctx, uint64_t half_index);
int psdr_process_ring(psdr_ctx *ctx, uint64_t first_half, int nframes);

/* ---- Level 2: batched frames ------------------------------------------------------ */
/* Frame loop body, src/fft.cpp:47-105, for nframes consecutive frames at once.
 * d_halves: device pointer to nframes+1 consecutive raw half-frames (cfg.input_format);
 * frame f = [half f ; half f+1] (50 % overlap, src/fft.cpp:49-53).  Device-side sample
 * conversion = src/samplereader.cpp:29-40.  Asynchronous on the context's stream. */
int psdr_process_batch(psdr_ctx *ctx, const void *d_halves, int nframes);

/* audio clients: AudioClient (src/signal.h:53-123) */
int psdr_client_add(psdr_ctx *ctx, int *id_out);
int psdr_client_remove(psdr_ctx *ctx, int id);
/* AudioClient::set_audio_range (src/signal.cpp:81-94): unchecked, like the reference */
int psdr_client_set_audio_range(psdr_ctx *ctx, int id, int l, double audio_mid, int r);
/* AudioClient::on_window_message (src/signal.cpp:300-314): validated; PSDR_ERR_INVALID
 * (and no change) when the reference would silently return */
int psdr_client_on_window_message(psdr_ctx *ctx, int id, int l, double audio_mid, int r);
/* AudioClient::set_audio_demodulation / on_demodulation_message (src/signal.cpp:95-97,316-328) */
int psdr_client_set_audio_demodulation(psdr_ctx *ctx, int id, int mode);
/* signal_loop's slow-client rule (src/websocket.cpp:170-176): the reference does not call send_audio at all for a
 * client with more than 50 kB queued on its socket, so NOTHING of that client moves for the frame - overlap-add tails and
 * FM's last sample (src/signal.cpp:200-203, 273-275), DC blocker and AGC (:277-284).  A paused client sits out every
 * psdr_demod_batch* until it is resumed (paused = 0): its state is frozen bit for bit, and its results read as
 * PSDR_ERR_NO_DATA for those batches.  A Level-2 caller evaluates the backlog BEFORE the frame is demodulated
 * (phantomsdr_amd/host/hip_level2.h). */
int psdr_client_set_paused(psdr_ctx *ctx, int id, int paused);
/* signal_loop + send_audio (src/websocket.cpp:156-185, src/signal.cpp:102-275) for every
 * active client over the frames of the last psdr_process_batch.  first_frame_num is the
 * server's frame counter of the first frame (flip parity, src/signal.cpp:160-168,223). */
int psdr_demod_batch(psdr_ctx *ctx, uint64_t first_frame_num);
/* same, but reading nframes spectra from a caller-supplied device buffer in the device layout of
 * psdr_spectrum_device_ptr() of an identically configured context (frame_stride_bins complex bins
 * between frames): used when the spectrum was produced on another GPU and received over xGMI
 * (SURVEY 8e). */
int psdr_demod_batch_from(psdr_ctx *ctx, const float *d_spec, size_t frame_stride_bins,
                          int nframes, uint64_t first_frame_num);
/* Band sharding (SURVEY 8e variant ii): a GPU that serves only the clients of one frequency band needs
 * only that band of the spectrum.  psdr_pack_band copies bins [first_bin, first_bin + nbins) - indexed
 * like the clients' l and r (IQ: client order, real: k), wrapping at the spectrum's end - of the first
 * nframes frames of the last batch out of the device layout into a linear device buffer (stream
 * ordered, no synchronisation): that is what is sent.  psdr_demod_batch_from_band is psdr_demod_batch_from
 * on such a buffer; every active client's [l, r) must lie inside the band (PSDR_ERR_INVALID names the
 * first that does not, and nothing is demodulated). */
int psdr_pack_band(psdr_ctx *ctx, int nframes, uint32_t first_bin, uint32_t nbins, float *d_out,
                   size_t out_stride_bins);
int psdr_demod_batch_from_band(psdr_ctx *ctx, const float *d_band, size_t frame_stride_bins, uint32_t first_bin,
                               uint32_t nbins, int nframes, uint64_t first_frame_num);
/* Band sharding WITHOUT the pack pass (2^20- and 2^21-point IQ contexts; PSDR_ERR_UNSUPPORTED otherwise - use
 * psdr_pack_band).
 * After psdr_set_band_layout(ctx, nbands, halo_bins) the second FFT pass writes the spectrum as nbands (a power of two,
 * <= 16) band REGIONS: region b holds bins [b*R/nbands, (b+1)*R/nbands + halo) - the halo, rounded up to whole
 * columns of M1 = N / 1024 bins, repeats the start of band b+1 - of ALL frames of the batch in one contiguous piece (frames
 * frame_stride_bins apart), in device order.  psdr_band_region gives the piece of the LAST processed batch: that is
 * what is sent to peer b, as it is.  Two result sets alternate from batch to batch (also on a caller's stream), so a
 * region stays valid until the batch after the next one is processed.  Everything else (demodulation, pyramid,
 * psdr_read_spectrum, psdr_pack_band) works unchanged on the root; psdr_spectrum_device_ptr does not (a frame is no
 * longer one piece).  Call before the first batch; reallocates the spectrum buffers.
 * psdr_demod_batch_from_band_region: psdr_demod_batch_from_band on a received region (an IQ context of the same size). */
int psdr_set_band_layout(psdr_ctx *ctx, int nbands, uint32_t halo_bins);
int psdr_band_region(psdr_ctx *ctx, int band, const float **d_region, size_t *frame_stride_bins, uint32_t *first_bin,
                     uint32_t *nbins);
int psdr_demod_batch_from_band_region(psdr_ctx *ctx, const float *d_region, size_t frame_stride_bins, uint32_t first_bin,
                                      uint32_t nbins, int nframes, uint64_t first_frame_num);
/* results of the last demod batch for one client: audio [nframes][n/2] floats (the
 * demodulated, overlap-added samples handed to the DC blocker at src/signal.cpp:278),
 * pwr [nframes] (average_power, src/signal.cpp:117-119), nan_flags [nframes] (1 = the
 * reference would have dropped the frame, src/signal.cpp:266-271).  Any may be NULL. */
/* nframes = rows the caller's buffers hold; it must be >= the frames of the last demod batch
 * (PSDR_ERR_INVALID otherwise, nothing is written); *nframes_out (may be NULL) = rows written. */
int psdr_read_audio(psdr_ctx *ctx, int id, int nframes, float *audio, float *pwr, int32_t *nan_flags,
                    int *nframes_out);
/* A client added after the last psdr_demod_batch has no results in it (the reference's frame loop would not
 * have posted a task for it either, src/websocket.cpp:156-185): psdr_read_audio / psdr_read_pcm / psdr_fetched_audio
 * return PSDR_ERR_NO_DATA for such a slot instead of the previous occupant's samples.
 *
 * Batched read-back - what a per-frame fan-out should use: psdr_fetch_batch copies the last demod batch's audio,
 * pwr, NaN flags (and PCM with the post chain on) of ALL client slots into pinned host memory owned by the
 * context with ONE synchronisation and at most four strided copies; psdr_fetched_audio then hands out pointers
 * into that block (valid until the next psdr_fetch_batch) without touching the device.  frame: index inside the
 * batch.  audio / pcm: audio_fft_size/2 values.  Any output pointer may be NULL. */
int psdr_fetch_batch(psdr_ctx *ctx);
int psdr_fetched_audio(psdr_ctx *ctx, int id, int frame, const float **audio, float *pwr, int32_t *nan_flag,
                       const int32_t **pcm);
/* The same read-back WITHOUT a stall of the frame loop - the served end of the path: every send_audio / send_waterfall of
 * the reference ends in host memory (src/signal.cpp:283-291 -> src/audio.cpp:26-44; src/waterfall.cpp:44-51).
 * psdr_fetch_begin enqueues the device-to-host copies of the last psdr_demod_batch* (pwr and NaN flags always; float audio
 * with PSDR_FETCH_AUDIO; the post chain's PCM with PSDR_FETCH_PCM) and of the last psdr_waterfall_batch
 * (PSDR_FETCH_WATERFALL) on a copy stream of the context, behind the kernels that produce them, into one of
 * PSDR_FETCH_SETS pinned host sets (a ring; a set's buffers are allocated when it is first used), and returns at once.  The caller then enqueues the next batch (psdr_process_* / psdr_demod_batch /
 * psdr_waterfall_batch): the copies run beside its FFT passes; the kernels that overwrite the device-side results wait for
 * the copies in stream order (no host wait).  psdr_fetch_end waits for the OLDEST fetch in flight; from then on
 * psdr_fetched_audio / _window / _waterfall answer from that set, until the next psdr_fetch_end - its pointers stay valid
 * until PSDR_FETCH_SETS - 1 further psdr_fetch_begin calls have been made.  How far the host stays behind is the caller's
 * choice: one batch for the float audio; the post chain's PCM is ready up to two steps after its passes (three with
 * hundreds of clients), so a caller that fetches it keeps two or three fetches in flight.  A psdr_fetch_begin with every
 * set in flight waits for the oldest copy and gives its results up.  Frame-loop thread only.  psdr_fetch_batch = every outstanding
 * psdr_fetch_end + a full drain + begin(all) + end.  bench.py's `with_fetch` times this pattern. */
#define PSDR_FETCH_SETS 4
#define PSDR_FETCH_AUDIO 1u
#define PSDR_FETCH_PCM 2u
#define PSDR_FETCH_WATERFALL 4u
int psdr_fetch_begin(psdr_ctx *ctx, unsigned what);
int psdr_fetch_end(psdr_ctx *ctx);
/* one frame's PCM row (audio_fft_size/2 int16) of client `id` in the fetched set when the batch was produced with
 * PSDR_OPT_POST_CHAIN_PCM16 = 1 (PSDR_ERR_STATE otherwise); pointer into pinned host memory, valid like psdr_fetched_audio's */
int psdr_fetched_pcm16(psdr_ctx *ctx, int id, int frame, const int16_t **pcm);
/* rows [nsent][r - l] of waterfall client `id` in the fetched set (pointer into pinned host memory, valid like
 * psdr_fetched_audio's), with the level and window they were GATHERED with.  PSDR_ERR_NO_DATA: the client was not
 * active in that batch.  Any output pointer may be NULL. */
int psdr_fetched_waterfall(psdr_ctx *ctx, int id, const int8_t **rows, int *nsent_out, int *level_out, int *l_out, int *r_out);
/* the window [l, r) and audio_mid client `id` was DEMODULATED with in the fetched batch (set_audio_range may have run on
 * another thread since, or have been refused): what the packet labels of src/signal.cpp:104-105, 287 must be computed
 * from.  Same error behaviour as psdr_fetched_audio. */
int psdr_fetched_window(psdr_ctx *ctx, int id, int *l, double *audio_mid, int *r);
/* device-resident results (no copy): audio of client slot `id` */
int psdr_audio_device_ptr(psdr_ctx *ctx, int id, const float **d_audio, const float **d_pwr);

/* Post-demodulation chain of AudioClient::send_audio (src/signal.cpp:277-284), batched for all
 * clients: DCBlocker (src/utils.h:139-169, delay audio_rate/750*2), AGC(0.2, 50 ms, 300 ms,
 * 200 ms look-ahead) (src/utils/audioprocessing.cpp:5-68; reset by
 * psdr_client_set_audio_demodulation like src/signal.cpp:316-328), dsp_float_to_int16 with
 * mult 65536/4 (src/utils/dsp.cpp:152-165).  Off by default; when on, every demod_batch also
 * produces the int16 PCM (held in int32, like the reference's int32_t buffer) that the reference
 * hands to its audio encoder.  Frames whose NaN flag is set are skipped by the chain (the
 * reference drops them before it, src/signal.cpp:266-271); their PCM rows are zero. */
int psdr_set_post_chain(psdr_ctx *ctx, int enable);
/* Knobs that are not part of psdr_config (whose layout is frozen per PSDR_ABI_VERSION).
 * PSDR_OPT_POST_CHAIN_STREAMS (before the first psdr_set_post_chain(ctx, 1); PSDR_ERR_STATE after it): which HIP streams the
 *   post chain's two sequential stages run on.  0 (default): streams created in a fixed order - deterministic, and in the
 *   FIRST context of a process these are the hardware queues that leave the FFT passes' launches alone.  1: chosen by a
 *   ~60 ms measurement of launch gaps (the chain's kernels beside empty launches on the main and the side stream) - for a
 *   process that creates several contexts, where creation order lands on a busy pipe of the command processor (+15 %
 *   instead of +3 % on the step); the outcome depends on wall-clock thresholds, and a failed
 *   measurement falls back to 0.
 * PSDR_OPT_POST_CHAIN_AGC (any time; drains the context): the form of the chain.  1 (default): the AGC as maxima of 16-sample
 *   chunks + ONE four-wave kernel for look-ahead peak, gain recurrence and int16 conversion - whenever the audio rate is a
 *   multiple of 80 Hz, the audio size a multiple of 8 and the CUs the chain reserves hold its work-groups - and the DC
 *   blocker's moving averages reading the demodulated rows themselves (no gathered copy while no frame is dropped and no
 *   client paused) and leaving the chunk maxima on their way: a third of the memory traffic of the other form, which is
 *   what the chain costs the FFT passes.  0: round 5's form everywhere (gather + five AGC kernels).
 *   Same bits either way, and the chain's carried state is the same: the forms may alternate between batches.
 * PSDR_OPT_POST_CHAIN_PCM16 (any time; drains the context): 1 = the chain writes its PCM as int16 rows instead of the int32
 *   buffer the reference hands its encoder (dsp_float_to_int16's output, src/utils/dsp.cpp:152-165, holds 16-bit values):
 *   half the bytes for psdr_fetch_begin(PSDR_FETCH_PCM) to move - with hundreds of clients the copy to the host is what
 *   bounds the served path.  psdr_fetched_pcm16 hands the rows out; psdr_fetched_audio's pcm is NULL for
 *   such a batch; psdr_read_pcm still delivers int32 (widened on the host).  0 (default): int32 rows. */
enum { PSDR_OPT_POST_CHAIN_STREAMS = 1, PSDR_OPT_POST_CHAIN_AGC = 2, PSDR_OPT_POST_CHAIN_PCM16 = 3 };
int psdr_set_option(psdr_ctx *ctx, int option, int value);
/* pcm: [frames of the last demod_batch][audio_fft_size/2]; nframes = rows pcm holds (as psdr_read_audio) */
int psdr_read_pcm(psdr_ctx *ctx, int id, int nframes, int32_t *pcm, int *nframes_out);

/* waterfall clients: WaterfallClient (src/waterfall.h) */
int psdr_waterfall_add(psdr_ctx *ctx, int *id_out);
int psdr_waterfall_remove(psdr_ctx *ctx, int id);
/* WaterfallClient::set_waterfall_range (src/waterfall.cpp:25-42); r is clamped to
 * R>>level (the reference forgets the upper bound, src/waterfall.cpp:55-58) */
int psdr_waterfall_set_range(psdr_ctx *ctx, int id, int level, int l, int r);
/* WaterfallClient::on_window_message (src/waterfall.cpp:53-94): picks the level */
int psdr_waterfall_on_window_message(psdr_ctx *ctx, int id, int l, int r, int *level_out,
                                     int *l_out, int *r_out);
/* waterfall_loop + send_waterfall (src/websocket.cpp:207-236, src/waterfall.cpp:44-51):
 * gathers q_level[l..r) of every waterfall client for every frame f of the last batch
 * with (first_frame_num+f) % skip_num == 0. */
int psdr_waterfall_batch(psdr_ctx *ctx, uint64_t first_frame_num);
/* bytes [nsent][r-l] for one client, with the range the rows were GATHERED with: the window may
 * have been changed by another thread since psdr_waterfall_batch, so level/l/r of that batch are
 * returned (any of the out pointers may be NULL).  *nsent_out = number of sent frames in the batch;
 * out == NULL only queries. */
int psdr_read_waterfall(psdr_ctx *ctx, int id, int8_t *out, size_t out_cap, int *nsent_out, int *level_out,
                        int *l_out, int *r_out);

/* last batch, raw device-side results (for consumers that stay on the GPU, and tests) */
/* spectrum of frame f, normalised by 1/N exactly like src/fft_impl.cpp:34-35.  IQ: N complex bins
 * indexed by the CLIENT coordinate c (bin k = (c+N/2+1) mod N); real: N/2+1 bins indexed by k.
 * The DEVICE LAYOUT is opaque: transforms whose row pass has 1024 points (2^20/2^21-point IQ,
 * 2^21/2^22-point real) keep a frame in tile-major 128-byte lines (phantomsdr_amd/csrc/quantize.h,
 * SpecLayout); psdr_demod_batch_from() on a context created with the same configuration understands
 * it (that is what the spectrum broadcast between GPUs ships), psdr_read_spectrum() delivers the
 * reference's k order.  nbins complex values per frame; frames are spec_stride apart:
 * N (IQ) or N/2+2 (real) bins. */
int psdr_spectrum_device_ptr(psdr_ctx *ctx, int frame, const float **d_spec, size_t *nbins);
/* (level-major layout as in the reference; materialised on demand from the device's tiled
 * records, so this call synchronises) */
int psdr_quantized_device_ptr(psdr_ctx *ctx, int frame, const int8_t **d_q, size_t *nbytes);
/* copies of the same to host; spectrum is delivered in the reference's k order */
int psdr_read_spectrum(psdr_ctx *ctx, int frame, float *out_k_order);
int psdr_read_quantized(psdr_ctx *ctx, int frame, int8_t *out);

/* ---- multi-GPU from C: ONE process, n devices of a node ---------------------------------------------
 * Device devices[0] is the root: it owns the raw ring, the forward FFT and the waterfall clients (use
 * psdr_group_ctx(g, 0) with the psdr_ring_* and psdr_waterfall_* calls above); the audio clients are spread over all n
 * contexts and every batch is exchanged ONCE over xGMI through RCCL, called directly (librccl.so is dlopen()ed when a
 * group of more than one device is created; a single-device group never loads it and issues no collective):
 *   PSDR_SHARD_CLIENTS  ncclBroadcast of the spectrum (BASELINE.json configs[3]); client i lives on device i mod n
 *   PSDR_SHARD_RAW      ncclBroadcast of the raw half-frames, every device runs the forward FFT itself
 *   PSDR_SHARD_BAND     device b receives band b of the spectrum + a halo of one maximal window (ncclSend / ncclRecv of
 *                       1/n of the bytes; n a power of two); a client lives on the device of the band its window starts in
 * | PSDR_SHARD_FORCE_COMM: create the communicator and issue the collectives even for ONE device (testing the RCCL
 * plumbing on a single-GPU box): CLIENTS / RAW broadcast to the one rank; BAND packs band 0 (the whole spectrum), sends it to
 * and receives it from itself (ncclSend / ncclRecv in one group) and demodulates from the received buffer.
 * | PSDR_SHARD_PEER_COPY: no RCCL at all - after the root's transform every peer pulls its share with hipMemcpyPeerAsync
 * on its own stream (n - 1 independent copies, one per root-to-peer xGMI link, ordered by events).  With it a device may
 * be listed more than once: n ranks on fewer GPUs, which is how a one-GPU box runs the multi-rank logic (placement, band
 * regions and halos, migration, fetch) for real.
 * The exchange of batch b runs BESIDE the root's transform of batch b + 1 (spectrum broadcast, and band regions written by
 * the root's second pass): the root alternates its two result sets and issues its side of the collective on an exchange
 * stream of its own; a set is overwritten only after its exchange has completed (stream-ordered, no host wait).
 * | PSDR_SHARD_SERIAL: exchange and transform strictly one after the other on the root's stream (round 5's schedule; A/B
 * and the bit-identity test of the overlapped schedule).  Raw sharding and packed band buffers are always serial.
 * Everything else of a rank is ordered on one stream per device; psdr_group_step returns without synchronising.  The calls
 * below may come from different threads (the server's websocket threads and its frame loop): the group serialises the
 * client calls (add / remove / set_* / fetched_*) against each other and against a step's enqueue.  psdr_group_step*,
 * psdr_group_fetch, psdr_group_synchronize and psdr_group_link_stats belong to the FRAME-LOOP thread alone (they wait for
 * devices and read the step's timing events: no lock is held while they do).
 * The process-per-GPU twin of this (torch.distributed over RCCL) is phantomsdr_amd/distributed.py.
 * Time sharding (batch g on device g mod n, no collective) needs no group: n independent contexts.
 * STATUS: no group of more than one physical device has run on hardware yet (every round's GPU box had one MI355X): the
 * RCCL calls have executed with one rank only, the multi-rank logic through PSDR_SHARD_PEER_COPY on one device.  Treat
 * ndevices > 1 - and PSDR_SHARD_BAND over RCCL in particular - as EXPERIMENTAL until a node run exists. */
typedef struct psdr_group psdr_group;
enum { PSDR_SHARD_CLIENTS = 0, PSDR_SHARD_RAW = 1, PSDR_SHARD_BAND = 2, PSDR_SHARD_FORCE_COMM = 0x100, PSDR_SHARD_PEER_COPY = 0x200,
       PSDR_SHARD_SERIAL = 0x400 };
int psdr_group_create(const psdr_config *cfg, const int *devices, int ndevices, int shard, psdr_group **out);
void psdr_group_destroy(psdr_group *g);
int psdr_group_size(const psdr_group *g);
psdr_ctx *psdr_group_ctx(psdr_group *g, int rank);
/* audio clients: the group picks the device; *gid_out names the client in every psdr_group_client_* / _fetched_* call and
 * NEVER changes (an index into the group's own table of (device, slot)).
 * psdr_group_client_set_audio_range: band sharding moves a client whose window now starts in another band to that
 * band's device behind its gid.  The demodulation state the reference keeps across a retune (src/signal.cpp:81-94) -
 * overlap-add tails, FM's last sample - the mode and the paused flag travel with it; the post chain's history on the GPU
 * (DC sums, AGC gain and look-ahead) does not: the client starts there like a fresh one (an AGC transient the reference
 * does not have); and psdr_group_fetched_audio answers PSDR_ERR_NO_DATA until the new device has demodulated a batch.
 * psdr_group_client_rank: the rank (index into `devices`) a client lives on now, -1 for an unknown gid. */
int psdr_group_client_add(psdr_group *g, int l, double audio_mid, int r, int mode, int *gid_out);
int psdr_group_client_remove(psdr_group *g, int gid);
int psdr_group_client_set_audio_range(psdr_group *g, int gid, int l, double audio_mid, int r);
int psdr_group_client_rank(psdr_group *g, int gid);
int psdr_group_client_set_audio_demodulation(psdr_group *g, int gid, int mode);
int psdr_group_client_set_paused(psdr_group *g, int gid, int paused);
/* one batch: the root transforms nframes frames (d_halves_root: nframes + 1 raw half-frames on the ROOT device; _ring:
 * the root context's ingest ring from first_half on), the exchange, every device demodulates its clients, the root
 * gathers the waterfall rows (psdr_demod_batch + psdr_waterfall_batch of a single context, for the whole group) */
int psdr_group_step(psdr_group *g, const void *d_halves_root, int nframes, uint64_t first_frame_num);
int psdr_group_step_ring(psdr_group *g, uint64_t first_half, int nframes, uint64_t first_frame_num);
int psdr_group_synchronize(psdr_group *g);
/* bytes that crossed ONE root-to-peer link in the last step, and the exchange's duration (RCCL: on the root's stream;
 * PSDR_SHARD_PEER_COPY: the slowest peer's own copy) */
int psdr_group_link_stats(psdr_group *g, double *bytes_per_link, double *exchange_ms);
/* psdr_fetch_batch on every device, then psdr_fetched_audio / psdr_fetched_window by gid */
int psdr_group_fetch(psdr_group *g);
int psdr_group_fetched_audio(psdr_group *g, int gid, int frame, const float **audio, float *pwr, int32_t *nan_flag,
                             const int32_t **pcm);
int psdr_group_fetched_window(psdr_group *g, int gid, int *l, double *audio_mid, int *r);

/* ---- wire formats of the reference's packets (host side) ----------------------- */
/* The CBOR map nlohmann::json::to_cbor produces in AudioEncoder::send (src/audio.cpp:17-36):
 * {"data": payload, "frame_num", "l", "m", "pwr", "r"} (keys in std::map order, shortest integer
 * heads, binary32 floats when exact).  payload = the encoded audio frame (FLAC/Opus bytes: the codecs
 * stay outside).  out must hold psdr_wire_packet_bound(bytes); *len = bytes written. */
size_t psdr_wire_packet_bound(size_t payload_bytes);
int psdr_wire_audio_packet(uint64_t frame_num, int l, double m, int r, double pwr, const void *payload,
                           size_t bytes, uint8_t *out, size_t cap, size_t *len);
/* WaterfallEncoder::set_data + the CBOR of ZstdEncoder::send (src/waterfallcompression.cpp:13-31):
 * {"data": int8 row, "frame_num", "l", "r"}; l, r are the client's range << level (src/waterfall.cpp:47) */
int psdr_wire_waterfall_packet(uint64_t frame_num, int l, int r, const void *payload, size_t bytes,
                               uint8_t *out, size_t cap, size_t *len);
/* ... and its zstd stream: one ZSTD_CStream per waterfall client, every packet flushed with
 * ZSTD_compressStream2(..., ZSTD_e_flush) (src/waterfallcompression.cpp:32-35).  libzstd is looked up
 * at run time; PSDR_ERR_UNSUPPORTED if the host has none. */
typedef struct psdr_zstd psdr_zstd;
int psdr_wire_zstd_create(psdr_zstd **out);
void psdr_wire_zstd_destroy(psdr_zstd *zs);
size_t psdr_wire_zstd_bound(size_t nbytes);
int psdr_wire_zstd_flush(psdr_zstd *zs, const void *in, size_t nbytes, uint8_t *out, size_t cap, size_t *len);

/* The text frame a client receives first (broadcast_server::send_basic_info, src/websocket.cpp:42-66):
 * a glaze (v2.4.4, subprojects/glaze.wrap) json_t object, i.e. a std::map - keys in lexicographic
 * order at both levels - whose numbers are all doubles, written in their shortest round-trip form
 * (integers without a fraction).  Strings are written as they are (the reference's are plain ASCII
 * mode and codec names; '"' and '\\' are escaped).  *len excludes the terminating NUL that is also
 * written. */
typedef struct psdr_hello {
    double sps, audio_max_sps, audio_max_fft, fft_size, fft_result_size, waterfall_size, basefreq;
    double total_bandwidth;           /* is_real ? sps / 2 : sps */
    double default_frequency, default_l, default_m, default_r;
    const char *default_modulation;   /* "USB" | "LSB" | "AM" | "FM" */
    const char *waterfall_compression, *audio_compression;
} psdr_hello;
int psdr_wire_hello_json(const psdr_hello *h, char *out, size_t cap, size_t *len);
/* A client's command frame (Client::on_message, src/client.cpp:19-117): a JSON object tagged by
 * "cmd" = "window" {l, r, m?, level?} | "demodulation" {demodulation} | "userid" {userid} | "mute"
 * {mute}.  Like glz::read_json with default options: an unknown key, a value of the wrong type or
 * malformed JSON rejects the message (PSDR_ERR_INVALID; the reference then ignores it, `if (ec)
 * return`), a missing key leaves its field at 0 / absent, null is "absent" for the optional m and
 * level.  text = the demodulation name or the user id cut to 32 characters (src/client.cpp:121). */
enum { PSDR_CMD_WINDOW = 0, PSDR_CMD_DEMODULATION = 1, PSDR_CMD_USERID = 2, PSDR_CMD_MUTE = 3 };
typedef struct psdr_command {
    int32_t cmd;
    int32_t l, r;
    int32_t has_m, has_level;
    double m;
    int32_t level;
    int32_t mute;
    char text[36];
} psdr_command;
int psdr_wire_parse_command(const char *msg, size_t len, psdr_command *out);

/* ---- instrumentation --------------------------------------------------------------- */
/* mode 0: off.  1: every kernel launch is bracketed by hipEvents on the stream it is launched on (the
 * marker packets between the kernels lengthen the two FFT passes by several per cent: use it for a replay,
 * not for a timed region).  2: no events; the two FFT passes stamp the device's constant 100 MHz clock at
 * their first work-group's entry and their last work-group's exit (two fire-and-forget atomics per
 * work-group) - cheap enough to stay on inside a timed region; up to 8192 launches per pass between resets. */
int psdr_set_profiling(psdr_ctx *ctx, int mode);
/* per-launch durations (microseconds) of kernel `name` ("fft_pass1", "fft_pass2", ...) since the last
 * reset, oldest first; *n_out = how many exist (may exceed cap) */
int psdr_get_kernel_samples(psdr_ctx *ctx, const char *name, double *us_out, int cap, int *n_out);
/* accumulated since the last reset: name[i] (static strings), total ms, launch count */
int psdr_get_kernel_stats(psdr_ctx *ctx, int max_entries, const char **names, double *total_ms,
                          int64_t *launches, int *n_out);
int psdr_reset_kernel_stats(psdr_ctx *ctx);
/* hipEvent-timed wall time of a region on the context's stream */
int psdr_timer_start(psdr_ctx *ctx);
int psdr_timer_stop_ms(psdr_ctx *ctx, double *ms_out);
/* the HIP stream (hipStream_t) the context launches on, for interop */
void *psdr_stream(psdr_ctx *ctx);
/* enqueue on the caller's stream instead (e.g. the stream RCCL collectives are ordered
 * against); NULL restores the context's own stream */
int psdr_set_stream(psdr_ctx *ctx, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PSDR_H */
