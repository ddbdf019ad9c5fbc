"""ctypes front-end to the CPU oracle (oracle/liboracle.so) and to the reference's own
compiled helpers (oracle/_ref/libref_dsp.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  phantomsdr_amd/ never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FMT = {"u8": 0, "s8": 1, "u16": 2, "s16": 3, "f32": 4, "f64": 5}
FMT_DTYPE = {"u8": np.uint8, "s8": np.int8, "u16": np.uint16, "s16": np.int16,
             "f32": np.float32, "f64": np.float64}
USB, LSB, AM, FM = 0, 1, 2, 3
MODES = {"USB": USB, "LSB": LSB, "AM": AM, "FM": FM}


def build(force=False):
    """(Re)build liboracle.so and, if /root/reference exists, _ref/libref_dsp.so."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "psdr_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    if need:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_dsp.so")
    if (force or not os.path.exists(ref_so)) and os.path.isdir("/root/reference/src/utils"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.orc_set_threads.argtypes = [i32]
        L.orc_convert.argtypes = [vp, i32, sz, vp]
        L.orc_build_hann_window.argtypes = [vp, i32]
        L.orc_vec_log2.argtypes = [f32, i32]
        L.orc_vec_log2.restype = f32
        L.orc_quantize.argtypes = [f32, i32]
        L.orc_quantize.restype = C.c_int8
        L.orc_dft_c2c.argtypes = [vp, vp, sz, i32]
        L.orc_dft_r2c.argtypes = [vp, vp, sz]
        L.orc_dft_c2r.argtypes = [vp, vp, sz]
        L.orc_fft_use_library.argtypes = [C.c_char_p]
        L.orc_fft_use_library.restype = i32
        L.orc_fft_library.restype = C.c_char_p
        L.orc_fft_library_threads.argtypes = [i32]
        L.orc_fft_library_threads.restype = i32
        L.orc_fft_create.argtypes = [sz, i32, i32, i32, i32]
        L.orc_fft_create.restype = vp
        L.orc_fft_destroy.argtypes = [vp]
        L.orc_fft_load_real_input.argtypes = [vp, vp, vp]
        L.orc_fft_load_complex_input.argtypes = [vp, vp, vp]
        L.orc_fft_execute.argtypes = [vp]
        for n in ("orc_fft_output", "orc_fft_quantized", "orc_fft_power"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = vp
        L.orc_fft_outbuf_len.argtypes = [vp]
        L.orc_fft_outbuf_len.restype = sz
        L.orc_fft_quantized_len.argtypes = [vp]
        L.orc_fft_quantized_len.restype = sz
        L.orc_pyramid_from_spectrum.argtypes = [vp, sz, i32, i32, i32, vp, vp]
        L.orc_client_create.argtypes = [i32, i32, i32, i32]
        L.orc_client_create.restype = vp
        L.orc_client_destroy.argtypes = [vp]
        L.orc_client_set_audio_range.argtypes = [vp, i32, C.c_double, i32]
        L.orc_client_set_audio_demodulation.argtypes = [vp, i32]
        L.orc_client_on_window_message.argtypes = [vp, i32, C.c_double, i32]
        L.orc_client_on_window_message.restype = i32
        L.orc_client_send_audio.argtypes = [vp, vp, sz, vp, vp, vp]
        L.orc_client_send_audio.restype = i32
        L.orc_client_real_prev.argtypes = [vp]
        L.orc_client_real_prev.restype = vp
        L.orc_client_baseband.argtypes = [vp]
        L.orc_client_baseband.restype = vp
        L.orc_dc_create.argtypes = [i32]
        L.orc_dc_create.restype = vp
        L.orc_dc_destroy.argtypes = [vp]
        L.orc_dc_remove.argtypes = [vp, vp, i32]
        L.orc_agc_create.argtypes = [f32] * 5
        L.orc_agc_create.restype = vp
        L.orc_agc_destroy.argtypes = [vp]
        L.orc_agc_process.argtypes = [vp, vp, sz]
        L.orc_agc_reset.argtypes = [vp]
        L.orc_float_to_int16.argtypes = [vp, vp, f32, sz]
        L.orc_am_demod.argtypes = [vp, vp, sz]
        L.orc_polar_discriminator_fm.argtypes = [vp, f32, f32, vp, sz]
        L.orc_waterfall_pick_level.argtypes = [i32, i32, vp, vp]
        L.orc_waterfall_pick_level.restype = i32
        _lib = L
    return _lib


def ref():
    """The reference's own dsp.cpp/audioprocessing.cpp (None if _ref was never built)."""
    global _ref
    if _ref is None:
        build()
        p = os.path.join(_HERE, "_ref", "libref_dsp.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        R.ref_build_hann_window.argtypes = [vp, i32]
        R.ref_polar_discriminator_fm.argtypes = [vp, f32, f32, vp, sz]
        R.ref_dsp_negate_float.argtypes = [vp, sz]
        R.ref_dsp_negate_complex.argtypes = [vp, sz]
        R.ref_dsp_add_float.argtypes = [vp, vp, sz]
        R.ref_dsp_add_complex.argtypes = [vp, vp, sz]
        R.ref_dsp_am_demod.argtypes = [vp, vp, sz]
        R.ref_dsp_float_to_int16.argtypes = [vp, vp, f32, sz]
        R.ref_agc_create.argtypes = [f32] * 5
        R.ref_agc_create.restype = vp
        R.ref_agc_destroy.argtypes = [vp]
        R.ref_agc_process.argtypes = [vp, vp, sz]
        R.ref_agc_reset.argtypes = [vp]
        _ref = R
    return _ref


_ref_core = None


def ref_core():
    """More of the reference's own code - class FFTW (fft_impl.cpp), convert<T> (samplereader.cpp), DCBlocker (utils.h) -
    when `make -C oracle ref_core` could build it (an image with a genuine fftw3.h + libfftw3f + boost); None otherwise."""
    global _ref_core
    if _ref_core is None:
        p = os.path.join(_HERE, "_ref", "libref_core.so")
        if not os.path.exists(p) and os.path.isdir("/root/reference/src/utils"):
            subprocess.call(["make", "-C", _HERE, "ref_core"], stdout=subprocess.DEVNULL)  # (prints why when it cannot)
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
        R.refc_fft_create.argtypes = [sz, i32, i32, i32, i32, i32]
        R.refc_fft_create.restype = vp
        R.refc_fft_destroy.argtypes = [vp]
        R.refc_fft_execute.argtypes = [vp, i32, vp, vp]
        R.refc_fft_output.argtypes = [vp]
        R.refc_fft_output.restype = vp
        R.refc_fft_quantized.argtypes = [vp]
        R.refc_fft_quantized.restype = vp
        R.refc_convert.argtypes = [i32, vp, vp, i32]
        R.refc_dc_create.argtypes = [i32]
        R.refc_dc_create.restype = vp
        R.refc_dc_destroy.argtypes = [vp]
        R.refc_dc_process.argtypes = [vp, vp, i32]
        R.refc_dc_reset.argtypes = [vp]
        _ref_core = R
    return _ref_core


def aligned(n, dtype, align=64):
    """numpy array whose data pointer is `align`-byte aligned (the reference's dsp.cpp
    uses std::assume_aligned<64>)."""
    dtype = np.dtype(dtype)
    raw = np.zeros(n * dtype.itemsize + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * dtype.itemsize].view(dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    lib().orc_set_threads(int(n))


FFT_LIBRARY_CANDIDATES = ("libfftw3f.so.3", "libfftw3f.so", "libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so",
                          "/opt/conda/lib/libmkl_rt.so")


def use_fft_library(path=None):
    """Make FFT objects created from now on run their big forward transform through a library with
    the FFTW3 API (what the reference's FFTW back-end calls, src/fft_impl.cpp:89-117,145).
    path=None probes FFT_LIBRARY_CANDIDATES in order; "" switches back to the built-in transform.
    Returns the name in use ("" = built-in).  MKL is run single-threaded (its threads are the
    caller's business: bench.py runs one pipeline per core)."""
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    L = lib()
    if path == "":
        L.orc_fft_use_library(b"")
        return ""
    for cand in ([path] if path else FFT_LIBRARY_CANDIDATES):
        if L.orc_fft_use_library(cand.encode()) == 0:
            return cand
    return ""


def fft_library():
    return lib().orc_fft_library().decode()


def fft_library_threads(n):
    """plans created from now on use n threads inside the FFT library (fftwf_plan_with_nthreads, src/fft_impl.cpp:82-88);
    False when the loaded library has no threads API"""
    return lib().orc_fft_library_threads(int(n)) == 0


def convert(raw, fmt):
    raw = np.ascontiguousarray(raw, dtype=FMT_DTYPE[fmt])
    out = np.empty(raw.size, np.float32)
    lib().orc_convert(_p(raw), FMT[fmt], raw.size, _p(out))
    return out


def hann(n):
    w = np.empty(n, np.float32)
    lib().orc_build_hann_window(_p(w), n)
    return w


def quantize(power, offset):
    power = np.ascontiguousarray(power, np.float32)
    L = lib()
    return np.array([L.orc_quantize(float(p), int(offset)) for p in power.ravel()],
                    np.int8).reshape(power.shape)


def dft_c2c(x, sign):
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty_like(x)
    lib().orc_dft_c2c(_p(x), _p(out), x.size, sign)
    return out


def dft_r2c(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size // 2 + 1, np.complex64)
    lib().orc_dft_r2c(_p(x), _p(out), x.size)
    return out


def dft_c2r(a, n):
    a = np.ascontiguousarray(a, np.complex64)
    assert a.size >= n // 2 + 1
    out = np.empty(n, np.float32)
    lib().orc_dft_c2r(_p(a), _p(out), n)
    return out


def pyramid_from_spectrum(spec_k_order, size, is_real, levels, brightness_offset=0):
    """int8 pyramid the reference computes from a given (normalised, k-order) spectrum."""
    import math
    spec = np.ascontiguousarray(spec_k_order, np.complex64)
    R = size // 2 if is_real else size
    q = np.zeros(sum(R >> i for i in range(levels)) + R, np.int8)
    size_log2 = int(round(math.log2(size))) + brightness_offset
    lib().orc_pyramid_from_spectrum(_p(spec), size, int(is_real), levels, size_log2, _p(q), None)
    return q[: sum(R >> i for i in range(levels))]


class FFT:
    """Mirror of the reference's `class FFTW` (src/fft_impl.cpp:80-183)."""

    def __init__(self, size, is_real, downsample_levels, brightness_offset=0,
                 additional_size=0):
        self.size, self.is_real = size, bool(is_real)
        self.levels = downsample_levels
        self.additional = additional_size
        self.h = lib().orc_fft_create(size, int(is_real), downsample_levels,
                                      brightness_offset, additional_size)
        self.R = size // 2 if is_real else size

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fft_destroy(self.h)
            self.h = None

    def load_real_input(self, a1, a2):
        a1 = np.ascontiguousarray(a1, np.float32)
        a2 = np.ascontiguousarray(a2, np.float32)
        assert a1.size == self.size // 2 and a2.size == self.size // 2
        lib().orc_fft_load_real_input(self.h, _p(a1), _p(a2))

    def load_complex_input(self, a1, a2):
        a1 = np.ascontiguousarray(a1, np.complex64)
        a2 = np.ascontiguousarray(a2, np.complex64)
        assert a1.size == self.size // 2 and a2.size == self.size // 2
        lib().orc_fft_load_complex_input(self.h, _p(a1), _p(a2))

    def load(self, a1, a2):
        (self.load_real_input if self.is_real else self.load_complex_input)(a1, a2)

    def execute(self):
        lib().orc_fft_execute(self.h)

    def output(self):
        """complex64 view of get_output_buffer(): N+A bins (IQ) or N/2+1 (real)."""
        nb = self.size // 2 + 1 if self.is_real else self.size + self.additional
        addr = lib().orc_fft_output(self.h)
        return np.ctypeslib.as_array((C.c_float * (2 * nb)).from_address(addr)).view(np.complex64)

    def quantized(self):
        n = sum(self.R >> i for i in range(self.levels))
        addr = lib().orc_fft_quantized(self.h)
        return np.ctypeslib.as_array((C.c_int8 * n).from_address(addr))

    def quantized_level(self, i):
        off = sum(self.R >> t for t in range(i))
        return self.quantized()[off:off + (self.R >> i)]

    def power(self):
        n = sum(self.R >> i for i in range(self.levels))
        addr = lib().orc_fft_power(self.h)
        return np.ctypeslib.as_array((C.c_float * n).from_address(addr))

    def slice_ptr_index(self, l):
        """signal_loop base addressing, src/websocket.cpp:157-160,182."""
        base = 0 if self.is_real else self.size // 2 + 1
        return (l + base) % self.R


class AudioClient:
    """Mirror of the reference's AudioClient DSP (src/signal.cpp:8-298)."""

    def __init__(self, is_real, audio_fft_size, audio_rate, fft_result_size):
        self.n = audio_fft_size
        self.is_real = bool(is_real)
        self.R = fft_result_size
        self.h = lib().orc_client_create(int(is_real), audio_fft_size, audio_rate,
                                         fft_result_size)
        self.l = self.r = 0
        self.m = 0.0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_client_destroy(self.h)
            self.h = None

    def set_audio_range(self, l, m, r):
        self.l, self.m, self.r = int(l), float(m), int(r)
        lib().orc_client_set_audio_range(self.h, int(l), float(m), int(r))

    def set_audio_demodulation(self, mode):
        mode = MODES[mode] if isinstance(mode, str) else int(mode)
        self.mode = mode
        lib().orc_client_set_audio_demodulation(self.h, mode)

    def on_window_message(self, l, m, r):
        ok = lib().orc_client_on_window_message(self.h, int(l), float(m), int(r))
        if ok:
            self.l, self.m, self.r = int(l), float(m), int(r)
        return bool(ok)

    def send_audio(self, spectrum, frame_num, fft: "FFT" = None, post=False, stats=True):
        """spectrum: the reference's output buffer (k order, with wrap copy).  Returns
        (audio_pre[n/2], pwr, pcm[n/2] or None, dropped).  stats=False skips the bookkeeping the tests' conditioned
        bounds use (bb_prev, fwd_scale: a pass over the whole spectrum) - the timed CPU baseline is the C call alone."""
        spectrum = np.ascontiguousarray(spectrum, np.complex64)
        if fft is not None:
            start = fft.slice_ptr_index(self.l)
        else:
            base = 0 if self.is_real else (self.R // 2 + 1)
            start = (self.l + base) % self.R
        assert start + (self.r - self.l) <= spectrum.size
        buf = spectrum[start:]
        audio = np.zeros(self.n // 2, np.float32)
        pwr = C.c_float(0)
        pcm = np.zeros(self.n // 2, np.int32) if post else None
        # FM's first sample pairs with the previous frame's last baseband sample (src/signal.cpp:258-262):
        # kept for the tests' conditioned FM bound (tests/helpers.py fm_tolerance)
        if stats:
            self._stats(spectrum)
        rc = lib().orc_client_send_audio(self.h, _p(buf), int(frame_num), _p(audio),
                                         C.byref(pwr), _p(pcm) if post else None)
        return audio, pwr.value, pcm, bool(rc)

    def _stats(self, spectrum):
        self.bb_prev = complex(self.baseband()[self.n // 2 - 1])
        # ... and the scale of what the FORWARD transform's rounding contributes to this frame's baseband: rms of the
        # whole spectrum (f32 butterfly errors are proportional to what flows through them, the strong carriers
        # included) times sqrt(bins summed); the previous frame's is kept beside it (FM pairs across the frame edge)
        self.fwd_scale_prev = getattr(self, "fwd_scale", 0.0)
        self.fwd_scale = float(np.sqrt(np.mean(np.abs(spectrum[: self.R].astype(np.complex128)) ** 2)) * np.sqrt(max(self.r - self.l, 1)))

    def real_prev(self):
        addr = lib().orc_client_real_prev(self.h)
        return np.ctypeslib.as_array((C.c_float * self.n).from_address(addr))[: self.n // 2].copy()

    def baseband(self):
        addr = lib().orc_client_baseband(self.h)
        return np.ctypeslib.as_array((C.c_float * (2 * self.n)).from_address(addr)).view(
            np.complex64).copy()


class PostChain:
    """DCBlocker + AGC + dsp_float_to_int16 exactly as AudioClient applies them after a frame
    survived the NaN guard (src/signal.cpp:277-284; state created at src/signal.cpp:54-55)."""

    def __init__(self, audio_rate):
        L = lib()
        self._L = L
        self.dc = L.orc_dc_create(audio_rate // 750 * 2)
        self.agc = L.orc_agc_create(0.2, 50.0, 300.0, 200.0, float(audio_rate))

    def __del__(self):
        if getattr(self, "dc", None):
            self._L.orc_dc_destroy(self.dc)
            self._L.orc_agc_destroy(self.agc)
            self.dc = None

    def reset_agc(self):
        self._L.orc_agc_reset(self.agc)

    def process(self, audio):
        """audio: float32 [h] of one frame -> int32 [h] (int16 range)"""
        a = np.ascontiguousarray(audio, np.float32).copy()
        self._L.orc_dc_remove(self.dc, _p(a), a.size)
        self._L.orc_agc_process(self.agc, _p(a), a.size)
        pcm = np.zeros(a.size, np.int32)
        self._L.orc_float_to_int16(_p(a), _p(pcm), 16384.0, a.size)
        return pcm


def waterfall_pick_level(levels, min_waterfall_fft, l, r):
    cl, cr = C.c_int(l), C.c_int(r)
    lv = lib().orc_waterfall_pick_level(levels, min_waterfall_fft, C.byref(cl), C.byref(cr))
    return lv, cl.value, cr.value


# --- derived parameters (src/spectrumserver.cpp:99-105,151,186-190; src/fft.cpp:33) ---
def derived_params(sps, fft_size, is_real, audio_sps=12000, waterfall_size=1024):
    import math
    R = fft_size // 2 if is_real else fft_size
    n = int(math.ceil(float(audio_sps) * fft_size / sps / 4.0) * 4)
    levels = 0
    cur = R
    while cur >= waterfall_size:
        levels += 1
        cur //= 2
    skip = max(1, int(math.floor((np.float32(sps) / np.float32(fft_size)) / 10.0)) * 2)
    return dict(R=R, audio_fft_size=n, downsample_levels=levels, skip_num=skip)
