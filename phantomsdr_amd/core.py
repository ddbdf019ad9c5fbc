"""Host-side mirror of the reference's plug-in/operator interface for the hot path, on top
of the C-ABI (include/psdr.h).  Names and argument meaning follow the reference:

  HipFFT            <-> class FFT / FFTW / cuFFT      (src/fft.h:33-63, src/fft_cuda.cu)
  AudioClient       <-> class AudioClient             (src/signal.h:53-123, src/signal.cpp)
  WaterfallClient   <-> class WaterfallClient         (src/waterfall.h, src/waterfall.cpp)
  SpectrumEngine    <-> the frame loop + fan-out      (src/fft.cpp:47-105,
                                                       src/websocket.cpp:156-185,207-236)

Everything numeric happens in libpsdr_hip.so (hand-written HIP, gfx950); this module only
marshals pointers.  numpy is used for host arrays; torch is not required here.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import PsdrError, check, psdr_config

FORMATS = {"u8": 0, "s8": 1, "u16": 2, "s16": 3, "f32": 4, "f64": 5}
FORMAT_DTYPES = {"u8": np.uint8, "s8": np.int8, "u16": np.uint16, "s16": np.int16,
                 "f32": np.float32, "f64": np.float64}
USB, LSB, AM, FM = 0, 1, 2, 3
MODES = {"USB": USB, "LSB": LSB, "AM": AM, "FM": FM}

FFTW_MEASURE, FFTW_DESTROY_INPUT, FFTW_ESTIMATE = 0, 1, 1 << 6  # accepted and ignored


def derived_params(sps, fft_size, is_real, audio_sps=12000, waterfall_size=1024):
    """src/spectrumserver.cpp:99-105 (R), :151 (audio_max_fft_size), :186-190
    (downsample_levels) and src/fft.cpp:33 (skip_num)."""
    R = fft_size // 2 if is_real else fft_size
    n = int(math.ceil(float(audio_sps) * fft_size / sps / 4.0) * 4)
    levels, cur = 0, R
    while cur >= waterfall_size:
        levels += 1
        cur //= 2
    skip = max(1, int(math.floor((np.float32(sps) / np.float32(fft_size)) / 10.0)) * 2)
    return dict(fft_result_size=R, audio_fft_size=n, downsample_levels=levels, skip_num=skip)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns one psdr_ctx."""

    @staticmethod
    def _config(fft_size, is_real, downsample_levels, brightness_offset=0,
                additional_size=0, audio_fft_size=0, audio_rate=12000, input_format="f32",
                device=0, max_batch=1, max_clients=1, max_waterfall_clients=1, skip_num=1, waterfall_size=0):
        cfg = psdr_config()
        cfg.struct_size = C.sizeof(psdr_config)
        cfg.fft_size = fft_size
        cfg.is_real = int(bool(is_real))
        cfg.downsample_levels = downsample_levels
        cfg.brightness_offset = brightness_offset
        cfg.additional_size = additional_size
        cfg.audio_fft_size = audio_fft_size
        cfg.audio_rate = audio_rate
        cfg.input_format = FORMATS[input_format] if isinstance(input_format, str) else input_format
        cfg.device = device
        cfg.max_batch = max_batch
        cfg.max_clients = max_clients
        cfg.max_waterfall_clients = max_waterfall_clients
        cfg.skip_num = skip_num
        cfg.waterfall_size = waterfall_size
        return cfg

    def __init__(self, fft_size, is_real, downsample_levels, brightness_offset=0,
                 additional_size=0, audio_fft_size=0, audio_rate=12000, input_format="f32",
                 device=0, max_batch=1, max_clients=1, max_waterfall_clients=1, skip_num=1, waterfall_size=0):
        self.lib = _lib.load()
        cfg = Context._config(fft_size, is_real, downsample_levels, brightness_offset, additional_size, audio_fft_size,
                              audio_rate, input_format, device, max_batch, max_clients, max_waterfall_clients, skip_num,
                              waterfall_size)
        self.h = C.c_void_p()
        check(self.lib.psdr_create(C.byref(cfg), C.byref(self.h)))
        self._owned = True
        self._describe(cfg)

    @classmethod
    def _view(cls, lib, handle, cfg):
        """a Context object over a psdr_ctx somebody else owns (a psdr_group's member): close() leaves it alone"""
        self = cls.__new__(cls)
        self.lib = lib
        self.h = C.c_void_p(handle)
        self._owned = False
        self._describe(cfg)
        return self

    def _describe(self, cfg):
        self.cfg = cfg
        fft_size, is_real = cfg.fft_size, bool(cfg.is_real)
        self.N = fft_size
        self.is_real = is_real
        self.R = fft_size // 2 if is_real else fft_size
        self.levels = cfg.downsample_levels
        self.n = cfg.audio_fft_size
        self.max_batch = cfg.max_batch
        self.q_len = sum(self.R >> i for i in range(cfg.downsample_levels))
        self.nbins = fft_size // 2 + 1 if is_real else fft_size
        self.last_nframes = 0
        self.last_demod_frames = 0
        self.input_format = cfg.input_format

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            for p in getattr(self, "_pinned", []):
                self.lib.psdr_host_free(self.h, p)
            self._pinned = []
            if getattr(self, "_owned", True):
                self.lib.psdr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- device memory -------------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.psdr_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        check(self.lib.psdr_dev_free(self.h, p))

    def h2d(self, dptr, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        check(self.lib.psdr_memcpy_h2d(self.h, C.c_void_p(dptr.value + offset), _ptr(arr), arr.nbytes))

    def d2h(self, arr, dptr, offset=0):
        check(self.lib.psdr_memcpy_d2h(self.h, _ptr(arr), C.c_void_p(dptr.value + offset), arr.nbytes))

    def synchronize(self):
        check(self.lib.psdr_synchronize(self.h))

    def half_frame_bytes(self):
        return self.lib.psdr_half_frame_bytes(self.h)

    # --- batched path ---------------------------------------------------------------
    def process_batch(self, d_halves, nframes, offset_bytes=0):
        """d_halves: device pointer (ctypes.c_void_p or int) to nframes+1 raw half-frames."""
        base = d_halves.value if isinstance(d_halves, C.c_void_p) else int(d_halves)
        check(self.lib.psdr_process_batch(self.h, C.c_void_p(base + offset_bytes), nframes))
        self.last_nframes = nframes

    OPT_POST_CHAIN_STREAMS = 1
    OPT_POST_CHAIN_PCM16 = 3  # 1: the chain's PCM as int16 rows (half the bytes to the host; fetched_pcm16), 0 (default): int32 rows
    OPT_POST_CHAIN_AGC = 2  # 1 (default): chunk maxima + one kernel for the AGC where the rate allows it; 0: the five-kernel form

    def set_option(self, option, value):
        check(self.lib.psdr_set_option(self.h, int(option), int(value)))

    def set_post_chain(self, enable=True, measured_streams=None):
        """batched DC blocker + AGC + int16 conversion after every demod_batch.  measured_streams (first enable only): choose
        the chain's streams by measurement (psdr.h: PSDR_OPT_POST_CHAIN_STREAMS = 1) instead of creation order"""
        if measured_streams is not None and not getattr(self, "post_chain_on", False) and not getattr(self, "_pc_set_up", False):
            self.set_option(self.OPT_POST_CHAIN_STREAMS, 1 if measured_streams else 0)
        if enable:
            self._pc_set_up = True
        check(self.lib.psdr_set_post_chain(self.h, 1 if enable else 0))
        self.post_chain_on = bool(enable)

    def demod_batch(self, first_frame_num):
        check(self.lib.psdr_demod_batch(self.h, first_frame_num))
        self.last_demod_frames = self.last_nframes

    # --- the served end: results to pinned host memory (psdr_fetch_*) ------------------
    FETCH_AUDIO, FETCH_PCM, FETCH_WATERFALL = 1, 2, 4

    def fetch_batch(self):
        check(self.lib.psdr_fetch_batch(self.h))

    def fetch_begin(self, what=FETCH_AUDIO | FETCH_WATERFALL):
        """enqueue the device-to-host copies of the last demodulation / waterfall batch (returns at once)"""
        check(self.lib.psdr_fetch_begin(self.h, int(what)))

    def fetch_end(self):
        """wait for the oldest fetch in flight; fetched_audio / fetched_waterfall answer from it afterwards"""
        check(self.lib.psdr_fetch_end(self.h))

    def fetched_audio(self, cid, frame, pcm=False):
        """(audio[n/2] copy or None, pwr, nan flag[, pcm[n/2] copy or None]) of one frame of the fetched batch"""
        a, pc = C.POINTER(C.c_float)(), C.POINTER(C.c_int32)()
        pw, nan = C.c_float(0), C.c_int32(0)
        check(self.lib.psdr_fetched_audio(self.h, int(cid), int(frame), C.byref(a), C.byref(pw), C.byref(nan), C.byref(pc)))
        h = self.cfg.audio_fft_size // 2
        au = np.ctypeslib.as_array(a, shape=(h,)).copy() if a else None
        if not pcm:
            return au, pw.value, nan.value
        return au, pw.value, nan.value, (np.ctypeslib.as_array(pc, shape=(h,)).copy() if pc else None)

    def fetched_pcm16(self, cid, frame):
        """the int16 PCM row [n/2] (copy) of one frame of the fetched batch (OPT_POST_CHAIN_PCM16 = 1)"""
        pc = C.POINTER(C.c_int16)()
        check(self.lib.psdr_fetched_pcm16(self.h, int(cid), int(frame), C.byref(pc)))
        return np.ctypeslib.as_array(pc, shape=(self.cfg.audio_fft_size // 2,)).copy()

    def fetched_waterfall(self, wid):
        """(rows [nsent][r - l] copy, level, l, r) of a waterfall client in the fetched batch"""
        rows = C.POINTER(C.c_int8)()
        ns, lv, l, r = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.lib.psdr_fetched_waterfall(self.h, int(wid), C.byref(rows), C.byref(ns), C.byref(lv), C.byref(l), C.byref(r)))
        ln = r.value - l.value
        if ns.value == 0 or ln == 0:
            return np.zeros((0, ln), np.int8), lv.value, l.value, r.value
        return np.ctypeslib.as_array(rows, shape=(ns.value, ln)).copy(), lv.value, l.value, r.value

    # --- streaming ingest (psdr_ring_*): pinned host half-frames -> HBM ring on a copy stream -----
    def ring_create(self, nhalves):
        check(self.lib.psdr_ring_create(self.h, int(nhalves)))

    def ring_write_async(self, half_index, host_half):
        """host_half: numpy array of one raw half-frame (ideally from pinned_array()); it must stay
        alive and unchanged until ring_wait(half_index) or a synchronising call returns."""
        assert host_half.nbytes == self.half_frame_bytes()
        check(self.lib.psdr_ring_write_async(self.h, int(half_index), _ptr(host_half)))

    def ring_wait(self, half_index):
        check(self.lib.psdr_ring_wait(self.h, int(half_index)))

    def process_ring(self, first_half, nframes):
        check(self.lib.psdr_process_ring(self.h, int(first_half), int(nframes)))
        self.last_nframes = nframes

    def pinned_array(self, nbytes, dtype=np.uint8):
        """numpy view of pinned host memory from psdr_host_alloc (freed with the context)"""
        p = C.c_void_p()
        nfl = (nbytes + 3) // 4
        check(self.lib.psdr_host_alloc(self.h, nfl, C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value)).view(dtype)

    def waterfall_batch(self, first_frame_num):
        check(self.lib.psdr_waterfall_batch(self.h, first_frame_num))

    def read_spectrum(self, frame):
        """complex64[nbins] in the reference's k order."""
        out = np.empty(self.nbins, np.complex64)
        check(self.lib.psdr_read_spectrum(self.h, frame, _ptr(out)))
        return out

    def read_quantized(self, frame):
        out = np.empty(self.q_len, np.int8)
        check(self.lib.psdr_read_quantized(self.h, frame, _ptr(out)))
        return out

    def quantized_level(self, q, i):
        off = sum(self.R >> t for t in range(i))
        return q[off:off + (self.R >> i)]

    # --- instrumentation ---------------------------------------------------------------
    def set_profiling(self, mode):
        """0 / False: off; 1 / True: hipEvent brackets around every launch; 2: device-clock stamps inside the
        two FFT passes (cheap enough for a timed region)"""
        check(self.lib.psdr_set_profiling(self.h, int(mode)))

    def kernel_samples(self, name):
        """per-launch durations (microseconds) of kernel `name` since the last reset"""
        n = C.c_int(0)
        check(self.lib.psdr_get_kernel_samples(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), np.float64)
        check(self.lib.psdr_get_kernel_samples(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n.value,
                                               C.byref(n)))
        return out[: n.value]

    def reset_kernel_stats(self):
        check(self.lib.psdr_reset_kernel_stats(self.h))

    def kernel_stats(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        cnt = (C.c_int64 * 16)()
        n = C.c_int(0)
        check(self.lib.psdr_get_kernel_stats(self.h, 16, names, ms, cnt, C.byref(n)))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}

    def timer_start(self):
        check(self.lib.psdr_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_double(0)
        check(self.lib.psdr_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value


class HipFFT:
    """The reference's FFT plug-in interface (src/fft.h:33-63) on the HIP back-end.

    Same call sequence as broadcast_server (src/spectrumserver.cpp:192-214, src/fft.cpp:17-30,
    61-98): ctor(size, nthreads, downsample_levels, brightness_offset),
    set_output_additional_size(A), plan_c2c()/plan_r2c(), malloc() x3,
    load_*_input(a1, a2), execute(), get_output_buffer(), get_quantized_buffer().
    """
    FORWARD, BACKWARD = 0, 1

    def __init__(self, size, nthreads=1, downsample_levels=1, brightness_offset=0, **ctx_kwargs):
        self.size = int(size)
        self.nthreads = nthreads  # CPU thread count of the FFTW sibling; unused on the GPU
        self.downsample_levels = downsample_levels
        self.brightness_offset = brightness_offset
        self.additional_size = 0
        self.ctx = None
        self._ctx_kwargs = ctx_kwargs
        self._bufs = {}

    def set_output_additional_size(self, size):
        self.additional_size = int(size)

    def _plan(self, is_real):
        assert self.ctx is None, "already planned"  # assert(!p), src/fft_impl.cpp:90,105
        self.ctx = Context(self.size, is_real, self.downsample_levels, self.brightness_offset,
                           self.additional_size, **self._ctx_kwargs)
        return 0

    def plan_c2c(self, direction=0, options=0):
        if direction != self.FORWARD:
            raise PsdrError(-6, "only FORWARD transforms are planned (src/fft.cpp:28)")
        return self._plan(False)

    def plan_r2c(self, options=0):
        return self._plan(True)

    def malloc(self, nfloats):
        """pinned host buffer as a float32 numpy view (FFT::malloc)."""
        assert self.ctx is not None, "plan first"
        p = C.c_void_p()
        check(self.ctx.lib.psdr_host_alloc(self.ctx.h, nfloats, C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_float * nfloats).from_address(p.value))
        self._bufs[arr.ctypes.data] = p
        return arr

    def free(self, arr):
        p = self._bufs.pop(arr.ctypes.data)
        check(self.ctx.lib.psdr_host_free(self.ctx.h, p))

    def load_real_input(self, a1, a2):
        a1 = np.ascontiguousarray(a1, np.float32)
        a2 = np.ascontiguousarray(a2, np.float32)
        assert a1.size == self.size // 2 and a2.size == self.size // 2
        return check(self.ctx.lib.psdr_load_real_input(self.ctx.h, _ptr(a1), _ptr(a2)))

    def load_complex_input(self, a1, a2):
        a1 = np.ascontiguousarray(a1).view(np.float32)
        a2 = np.ascontiguousarray(a2).view(np.float32)
        assert a1.size == self.size and a2.size == self.size
        return check(self.ctx.lib.psdr_load_complex_input(self.ctx.h, _ptr(a1), _ptr(a2)))

    def execute(self):
        rc = check(self.ctx.lib.psdr_execute(self.ctx.h))
        self.ctx.last_nframes = 1
        return rc

    def get_output_buffer(self):
        """complex64 view: N+A bins (IQ) / N/2+1 bins (real), natural k order."""
        p = C.c_void_p()
        check(self.ctx.lib.psdr_get_output_buffer(self.ctx.h, C.byref(p)))
        nb = self.size // 2 + 1 if self.ctx.is_real else self.size + self.additional_size
        return np.ctypeslib.as_array((C.c_float * (2 * nb)).from_address(p.value)).view(np.complex64)

    def get_quantized_buffer(self):
        p = C.c_void_p()
        check(self.ctx.lib.psdr_get_quantized_buffer(self.ctx.h, C.byref(p)))
        return np.ctypeslib.as_array((C.c_int8 * self.ctx.q_len).from_address(p.value))

    def close(self):
        if self.ctx is not None:
            for p in list(self._bufs.values()):
                self.ctx.lib.psdr_host_free(self.ctx.h, p)
            self._bufs.clear()
            self.ctx.close()
            self.ctx = None


class AudioClient:
    """AudioClient (src/signal.h:53-123): one tuned demodulator channel on a Context."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        cid = C.c_int(-1)
        check(ctx.lib.psdr_client_add(ctx.h, C.byref(cid)))
        self.id = cid.value
        self.l = self.r = 0
        self.audio_mid = 0.0
        self.demodulation = USB

    def set_audio_range(self, l, audio_mid, r):
        check(self.ctx.lib.psdr_client_set_audio_range(self.ctx.h, self.id, int(l), float(audio_mid), int(r)))
        self.l, self.audio_mid, self.r = int(l), float(audio_mid), int(r)

    def set_audio_demodulation(self, demodulation):
        mode = MODES[demodulation] if isinstance(demodulation, str) else int(demodulation)
        check(self.ctx.lib.psdr_client_set_audio_demodulation(self.ctx.h, self.id, mode))
        self.demodulation = mode

    def set_paused(self, paused):
        """signal_loop's slow-client rule (src/websocket.cpp:170-176): a paused client gets no send_audio call -
        it sits out the demodulation batches with all of its state frozen."""
        check(self.ctx.lib.psdr_client_set_paused(self.ctx.h, self.id, 1 if paused else 0))

    def on_window_message(self, l, m, r):
        """returns False where the reference silently returns (src/signal.cpp:302-311)."""
        if m is None:
            return False
        rc = self.ctx.lib.psdr_client_on_window_message(self.ctx.h, self.id, int(l), float(m), int(r))
        if rc == -1:
            return False
        check(rc)
        self.l, self.audio_mid, self.r = int(l), float(m), int(r)
        return True

    def on_demodulation_message(self, demodulation: str):
        if demodulation in MODES:  # unknown strings leave the mode alone (src/signal.cpp:318-326)
            self.set_audio_demodulation(demodulation)

    def read_audio(self, nframes=None):
        """(audio[F][n/2], pwr[F], nan[F]) of the last demod batch (F = its frame count; `nframes`,
        if given, is the number of rows to allocate and must not be smaller)."""
        F = nframes or self.ctx.last_demod_frames or self.ctx.last_nframes
        h = self.ctx.n // 2
        audio = np.empty((F, h), np.float32)
        pwr = np.empty(F, np.float32)
        nan = np.empty(F, np.int32)
        got = C.c_int(0)
        check(self.ctx.lib.psdr_read_audio(self.ctx.h, self.id, F, _ptr(audio), _ptr(pwr), _ptr(nan), C.byref(got)))
        return audio[:got.value], pwr[:got.value], nan[:got.value]

    def read_pcm(self, nframes=None):
        """int16 PCM (in int32, like the reference's buffer) of the last demod batch after the
        DC blocker / AGC / int16 conversion (src/signal.cpp:277-284); needs
        Context.set_post_chain(True)."""
        F = nframes or self.ctx.last_demod_frames or self.ctx.last_nframes
        pcm = np.empty((F, self.ctx.n // 2), np.int32)
        got = C.c_int(0)
        check(self.ctx.lib.psdr_read_pcm(self.ctx.h, self.id, F, _ptr(pcm), C.byref(got)))
        return pcm[:got.value]

    def on_close(self):
        if self.id >= 0 and self.ctx.h:
            check(self.ctx.lib.psdr_client_remove(self.ctx.h, self.id))
            self.id = -1


class WaterfallClient:
    """WaterfallClient (src/waterfall.h): a [level, l, r) window on the int8 pyramid."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        wid = C.c_int(-1)
        check(ctx.lib.psdr_waterfall_add(ctx.h, C.byref(wid)))
        self.id = wid.value
        self.level = ctx.levels - 1
        mwf = ctx.cfg.waterfall_size or (ctx.R >> self.level)
        self.l, self.r = 0, min(mwf, ctx.R >> self.level)

    def set_waterfall_range(self, level, l, r):
        check(self.ctx.lib.psdr_waterfall_set_range(self.ctx.h, self.id, int(level), int(l), int(r)))
        self.level, self.l = int(level), max(0, int(l))
        self.r = min(int(r), self.ctx.R >> self.level)

    def on_window_message(self, l, r):
        lv, nl, nr = C.c_int(), C.c_int(), C.c_int()
        rc = self.ctx.lib.psdr_waterfall_on_window_message(self.ctx.h, self.id, int(l), int(r),
                                                           C.byref(lv), C.byref(nl), C.byref(nr))
        if rc == -1:
            return False
        check(rc)
        self.level, self.l, self.r = lv.value, nl.value, nr.value
        return True

    def read_waterfall(self):
        """int8[nsent][r-l] for the frames of the last waterfall batch that were sent, plus
        the (l << level, r << level) labels of send_waterfall (src/waterfall.cpp:47).  Row length
        and labels are those of the BATCH (the window may have moved since)."""
        ns, lv, l, r = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib, h = self.ctx.lib, self.ctx.h
        check(lib.psdr_read_waterfall(h, self.id, None, 0, C.byref(ns), C.byref(lv), C.byref(l), C.byref(r)))
        ln = r.value - l.value
        cap = max(1, ns.value * ln)
        out = np.empty(cap, np.int8)
        check(lib.psdr_read_waterfall(h, self.id, _ptr(out), cap, C.byref(ns), C.byref(lv), C.byref(l), C.byref(r)))
        ln = r.value - l.value
        return out[: ns.value * ln].reshape(ns.value, ln), (l.value << lv.value, r.value << lv.value)

    def on_close(self):
        if self.id >= 0 and self.ctx.h:
            check(self.ctx.lib.psdr_waterfall_remove(self.ctx.h, self.id))
            self.id = -1


class Group:
    """psdr_group (include/psdr.h): one process, n GPUs, the batch exchanged over xGMI through RCCL called from C.
    shard: "clients" | "raw" | "band"; force_comm: issue the collectives even with one device (testing); peer_copy: no
    RCCL, the peers pull with hipMemcpyPeerAsync (a device may then be listed more than once)."""
    SHARDS = {"clients": 0, "raw": 1, "band": 2}

    def __init__(self, devices, shard, fft_size, is_real, downsample_levels, force_comm=False, peer_copy=False, serial=False, **ctx_kwargs):
        # a throw-away Context object only to build the psdr_config the same way Context does
        self.lib = _lib.load()
        cfg = Context._config(fft_size, is_real, downsample_levels, **ctx_kwargs)
        devs = (C.c_int * len(devices))(*devices)
        self.h = C.c_void_p()
        # serial: PSDR_SHARD_SERIAL - exchange and transform one after the other (default: the exchange of batch b beside the transform of b + 1)
        flag = self.SHARDS[shard] | (0x100 if force_comm else 0) | (0x200 if peer_copy else 0) | (0x400 if serial else 0)
        check(self.lib.psdr_group_create(C.byref(cfg), devs, len(devices), flag, C.byref(self.h)))
        self.n = len(devices)
        self.cfg = cfg
        self.h_audio = cfg.audio_fft_size // 2
        self.root = Context._view(self.lib, self.lib.psdr_group_ctx(self.h, 0), cfg)

    def client_add(self, l, mid, r, mode):
        gid = C.c_int(-1)
        mode = MODES[mode] if isinstance(mode, str) else int(mode)
        check(self.lib.psdr_group_client_add(self.h, int(l), float(mid), int(r), mode, C.byref(gid)))
        return gid.value

    def client_set_audio_range(self, gid, l, mid, r):
        """(band sharding may move the client to another device behind its gid, which stays what it is)"""
        check(self.lib.psdr_group_client_set_audio_range(self.h, int(gid), int(l), float(mid), int(r)))
        return gid

    def client_rank(self, gid):
        return int(self.lib.psdr_group_client_rank(self.h, int(gid)))

    def ctx_view(self, rank):
        """the context of `rank` (psdr_group_ctx), owned by the group"""
        return Context._view(self.lib, self.lib.psdr_group_ctx(self.h, int(rank)), self.cfg)

    def client_set_paused(self, gid, paused):
        check(self.lib.psdr_group_client_set_paused(self.h, gid, 1 if paused else 0))

    def step(self, d_halves, nframes, first_frame_num, offset_bytes=0):
        base = d_halves.value if isinstance(d_halves, C.c_void_p) else int(d_halves)
        check(self.lib.psdr_group_step(self.h, C.c_void_p(base + offset_bytes), nframes, first_frame_num))
        self.root.last_nframes = self.root.last_demod_frames = nframes

    def step_ring(self, first_half, nframes, first_frame_num):
        check(self.lib.psdr_group_step_ring(self.h, first_half, nframes, first_frame_num))
        self.root.last_nframes = self.root.last_demod_frames = nframes

    def synchronize(self):
        check(self.lib.psdr_group_synchronize(self.h))

    def fetch(self):
        check(self.lib.psdr_group_fetch(self.h))

    def fetched_audio(self, gid, frame):
        """(audio[n/2] copy, pwr, nan flag) of one frame of the fetched batch"""
        a = C.POINTER(C.c_float)()
        pw, nan = C.c_float(0), C.c_int32(0)
        check(self.lib.psdr_group_fetched_audio(self.h, gid, frame, C.byref(a), C.byref(pw), C.byref(nan), None))
        return np.ctypeslib.as_array(a, shape=(self.h_audio,)).copy(), pw.value, nan.value

    def link_stats(self):
        b, ms = C.c_double(0), C.c_double(0)
        check(self.lib.psdr_group_link_stats(self.h, C.byref(b), C.byref(ms)))
        return b.value, ms.value

    def close(self):
        if self.h:
            self.lib.psdr_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SpectrumEngine:
    """The frame loop of broadcast_server::fft_task (src/fft.cpp:47-105) over a raw sample
    ring that lives in HBM, batched F frames per launch, with the signal_loop /
    waterfall_loop fan-out (src/websocket.cpp:156-185,207-236) executed on the GPU."""

    def __init__(self, sps, fft_size, is_real, input_format="s16", audio_sps=12000,
                 waterfall_size=1024, brightness_offset=0, max_batch=16, max_clients=256,
                 max_waterfall_clients=64, device=0):
        p = derived_params(sps, fft_size, is_real, audio_sps, waterfall_size)
        self.params = p
        self.sps, self.fft_size, self.is_real = sps, fft_size, bool(is_real)
        self.input_format = input_format
        self.ctx = Context(fft_size, is_real, p["downsample_levels"], brightness_offset,
                           additional_size=p["audio_fft_size"], audio_fft_size=p["audio_fft_size"],
                           audio_rate=audio_sps, input_format=input_format, device=device,
                           max_batch=max_batch, max_clients=max_clients,
                           max_waterfall_clients=max_waterfall_clients, skip_num=p["skip_num"],
                           waterfall_size=waterfall_size)
        self.frame_num = 0
        self.audio_clients = []
        self.waterfall_clients = []
        self.ring = None
        self.ring_halves = 0

    def add_audio_client(self, l, m, r, mode="USB"):
        c = AudioClient(self.ctx)
        c.set_audio_demodulation(mode)
        c.set_audio_range(l, m, r)
        self.audio_clients.append(c)
        return c

    def add_waterfall_client(self, level=None, l=None, r=None):
        w = WaterfallClient(self.ctx)
        if level is not None:
            w.set_waterfall_range(level, l, r)
        self.waterfall_clients.append(w)
        return w

    def upload_ring(self, raw):
        """raw: numpy array of whole half-frames in input_format; becomes the device ring."""
        raw = np.ascontiguousarray(raw, FORMAT_DTYPES[self.input_format])
        hb = self.ctx.half_frame_bytes()
        assert raw.nbytes % hb == 0, "ring must hold whole half-frames"
        if self.ring is not None:
            self.ctx.dev_free(self.ring)
        self.ring = self.ctx.dev_alloc(raw.nbytes)
        self.ctx.h2d(self.ring, raw)
        self.ring_halves = raw.nbytes // hb

    def step(self, first_half, nframes, demod=True, waterfall=True):
        """frames [first_half, first_half+nframes) of the ring: FFT + pyramid, then the
        per-client fan-out.  Asynchronous; results are read through the client objects."""
        assert first_half + nframes + 1 <= self.ring_halves
        self.ctx.process_batch(self.ring, nframes, first_half * self.ctx.half_frame_bytes())
        if demod and self.audio_clients:
            self.ctx.demod_batch(self.frame_num)
        if waterfall and self.waterfall_clients:
            self.ctx.waterfall_batch(self.frame_num)
        self.frame_num += nframes

    def close(self):
        if self.ring is not None:
            self.ctx.dev_free(self.ring)
            self.ring = None
        self.ctx.close()
