"""ctypes binding of the C-ABI in include/psdr.h (libpsdr_hip.so, hand-written HIP for
gfx950).  There is no CPU fallback: if the library is missing the import fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSDR_LIB selects another build of the same library (kernel tuning variants)
_SO = os.environ.get("PSDR_LIB") or os.path.join(_HERE, "libpsdr_hip.so")


class PsdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"psdr error {code}: {msg}")
        self.code = code


class psdr_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("fft_size", C.c_uint32), ("is_real", C.c_int32),
        ("downsample_levels", C.c_int32), ("brightness_offset", C.c_int32),
        ("additional_size", C.c_int32), ("audio_fft_size", C.c_int32),
        ("audio_rate", C.c_int32), ("input_format", C.c_int32), ("device", C.c_int32),
        ("max_batch", C.c_int32), ("max_clients", C.c_int32),
        ("max_waterfall_clients", C.c_int32), ("skip_num", C.c_int32), ("waterfall_size", C.c_int32),
    ]


# every symbol include/psdr.h declares: (name, restype, argtypes)
_vp, _i, _sz, _u64 = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64
_pp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("psdr_last_error", C.c_char_p, []),
    ("psdr_version", C.c_char_p, []),
    ("psdr_create", _i, [C.POINTER(psdr_config), _pp]),
    ("psdr_destroy", None, [_vp]),
    ("psdr_host_alloc", _i, [_vp, _sz, _pp]),
    ("psdr_host_free", _i, [_vp, _vp]),
    ("psdr_load_real_input", _i, [_vp, _vp, _vp]),
    ("psdr_load_complex_input", _i, [_vp, _vp, _vp]),
    ("psdr_execute", _i, [_vp]),
    ("psdr_get_output_buffer", _i, [_vp, _pp]),
    ("psdr_get_quantized_buffer", _i, [_vp, _pp]),
    ("psdr_dev_alloc", _i, [_vp, _sz, _pp]),
    ("psdr_dev_free", _i, [_vp, _vp]),
    ("psdr_memcpy_h2d", _i, [_vp, _vp, _vp, _sz]),
    ("psdr_memcpy_d2h", _i, [_vp, _vp, _vp, _sz]),
    ("psdr_synchronize", _i, [_vp]),
    ("psdr_half_frame_bytes", _sz, [_vp]),
    ("psdr_wire_packet_bound", _sz, [_sz]),
    ("psdr_wire_audio_packet", _i, [_u64, _i, C.c_double, _i, C.c_double, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("psdr_wire_waterfall_packet", _i, [_u64, _i, _i, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("psdr_wire_zstd_create", _i, [_pp]),
    ("psdr_wire_zstd_destroy", None, [_vp]),
    ("psdr_wire_zstd_bound", _sz, [_sz]),
    ("psdr_wire_zstd_flush", _i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("psdr_wire_hello_json", _i, [_vp, C.c_char_p, _sz, C.POINTER(_sz)]),
    ("psdr_wire_parse_command", _i, [C.c_char_p, _sz, _vp]),
    ("psdr_ring_create", _i, [_vp, _i]),
    ("psdr_ring_write_async", _i, [_vp, _u64, _vp]),
    ("psdr_ring_wait", _i, [_vp, _u64]),
    ("psdr_process_ring", _i, [_vp, _u64, _i]),
    ("psdr_process_batch", _i, [_vp, _vp, _i]),
    ("psdr_client_add", _i, [_vp, C.POINTER(_i)]),
    ("psdr_client_remove", _i, [_vp, _i]),
    ("psdr_client_set_audio_range", _i, [_vp, _i, _i, C.c_double, _i]),
    ("psdr_client_on_window_message", _i, [_vp, _i, _i, C.c_double, _i]),
    ("psdr_client_set_audio_demodulation", _i, [_vp, _i, _i]),
    ("psdr_client_set_paused", _i, [_vp, _i, _i]),
    ("psdr_demod_batch", _i, [_vp, _u64]),
    ("psdr_demod_batch_from", _i, [_vp, _vp, _sz, _i, _u64]),
    ("psdr_pack_band", _i, [_vp, _i, C.c_uint32, C.c_uint32, _vp, _sz]),
    ("psdr_demod_batch_from_band", _i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _i, _u64]),
    ("psdr_set_band_layout", _i, [_vp, _i, C.c_uint32]),
    ("psdr_band_region", _i, [_vp, _i, _pp, C.POINTER(_sz), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("psdr_demod_batch_from_band_region", _i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _i, _u64]),
    ("psdr_read_audio", _i, [_vp, _i, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    ("psdr_audio_device_ptr", _i, [_vp, _i, _pp, _pp]),
    ("psdr_set_post_chain", _i, [_vp, _i]),
    ("psdr_set_option", _i, [_vp, _i, _i]),
    ("psdr_read_pcm", _i, [_vp, _i, _i, _vp, C.POINTER(_i)]),
    ("psdr_waterfall_add", _i, [_vp, C.POINTER(_i)]),
    ("psdr_waterfall_remove", _i, [_vp, _i]),
    ("psdr_waterfall_set_range", _i, [_vp, _i, _i, _i, _i]),
    ("psdr_waterfall_on_window_message", _i,
     [_vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("psdr_waterfall_batch", _i, [_vp, _u64]),
    ("psdr_read_waterfall", _i, [_vp, _i, _vp, _sz, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("psdr_spectrum_device_ptr", _i, [_vp, _i, _pp, C.POINTER(_sz)]),
    ("psdr_quantized_device_ptr", _i, [_vp, _i, _pp, C.POINTER(_sz)]),
    ("psdr_read_spectrum", _i, [_vp, _i, _vp]),
    ("psdr_read_quantized", _i, [_vp, _i, _vp]),
    ("psdr_set_profiling", _i, [_vp, _i]),
    ("psdr_get_kernel_stats", _i,
     [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(_i)]),
    ("psdr_fetch_batch", _i, [_vp]),
    ("psdr_fetched_audio", _i, [_vp, _i, _i, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                C.POINTER(C.POINTER(C.c_int32))]),
    ("psdr_fetched_window", _i, [_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("psdr_abi_version", _i, []),
    ("psdr_fetch_begin", _i, [_vp, C.c_uint]),
    ("psdr_fetched_pcm16", _i, [_vp, _i, _i, C.POINTER(C.POINTER(C.c_int16))]),
    ("psdr_fetch_end", _i, [_vp]),
    ("psdr_fetched_waterfall", _i, [_vp, _i, C.POINTER(C.POINTER(C.c_int8)), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("psdr_get_kernel_samples", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), _i, C.POINTER(_i)]),
    ("psdr_reset_kernel_stats", _i, [_vp]),
    ("psdr_timer_start", _i, [_vp]),
    ("psdr_timer_stop_ms", _i, [_vp, C.POINTER(C.c_double)]),
    ("psdr_stream", _vp, [_vp]),
    ("psdr_set_stream", _i, [_vp, _vp]),
    ("psdr_group_create", _i, [C.POINTER(psdr_config), C.POINTER(_i), _i, _i, _pp]),
    ("psdr_group_destroy", None, [_vp]),
    ("psdr_group_size", _i, [_vp]),
    ("psdr_group_ctx", _vp, [_vp, _i]),
    ("psdr_group_client_add", _i, [_vp, _i, C.c_double, _i, _i, C.POINTER(_i)]),
    ("psdr_group_client_remove", _i, [_vp, _i]),
    ("psdr_group_client_set_audio_range", _i, [_vp, _i, _i, C.c_double, _i]),
    ("psdr_group_client_rank", _i, [_vp, _i]),
    ("psdr_group_client_set_audio_demodulation", _i, [_vp, _i, _i]),
    ("psdr_group_client_set_paused", _i, [_vp, _i, _i]),
    ("psdr_group_step", _i, [_vp, _vp, _i, _u64]),
    ("psdr_group_step_ring", _i, [_vp, _u64, _i, _u64]),
    ("psdr_group_synchronize", _i, [_vp]),
    ("psdr_group_link_stats", _i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("psdr_group_fetch", _i, [_vp]),
    ("psdr_group_fetched_audio", _i, [_vp, _i, _i, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                      C.POINTER(C.POINTER(C.c_int32))]),
    ("psdr_group_fetched_window", _i, [_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
]

_lib = None


def load():
    """dlopen libpsdr_hip.so and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "phantomsdr_amd has no CPU fallback.")
    L = C.CDLL(_SO)
    # PSDR_LIB_LENIENT=1 (tools only: A/B against a library of an EARLIER round through PSDR_LIB): entry points that library
    # does not have yet are skipped instead of refusing to load
    lenient = os.environ.get("PSDR_LIB_LENIENT") == "1"
    for name, res, args in SYMBOLS:
        if lenient and not hasattr(L, name):
            continue
        fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PsdrError(rc, load().psdr_last_error().decode())
    return rc


class Hello(C.Structure):
    """struct psdr_hello (include/psdr.h)"""
    _fields_ = [(n, C.c_double) for n in ("sps", "audio_max_sps", "audio_max_fft", "fft_size", "fft_result_size",
                                          "waterfall_size", "basefreq", "total_bandwidth", "default_frequency",
                                          "default_l", "default_m", "default_r")] + [
        ("default_modulation", C.c_char_p), ("waterfall_compression", C.c_char_p), ("audio_compression", C.c_char_p)]


class Command(C.Structure):
    """struct psdr_command (include/psdr.h)"""
    _fields_ = [("cmd", C.c_int32), ("l", C.c_int32), ("r", C.c_int32), ("has_m", C.c_int32), ("has_level", C.c_int32),
                ("m", C.c_double), ("level", C.c_int32), ("mute", C.c_int32), ("text", C.c_char * 36)]
