"""Wire formats of the reference's packets, host-side C behind the ABI (no GPU):
the CBOR maps of AudioEncoder::send (src/audio.cpp:17-36) and ZstdEncoder::send
(src/waterfallcompression.cpp:13-37) as nlohmann::json 3.11.2's to_cbor emits them, and the
per-client zstd stream.  Checked against hand-assembled RFC 8949 byte strings and an independent
minimal CBOR decoder written here; the zstd stream is decoded with libzstd's streaming API."""
import ctypes as C
import struct

import numpy as np
import pytest

from phantomsdr_amd import _lib


def _audio(frame_num, l, m, r, pwr, payload):
    L = _lib.load()
    cap = L.psdr_wire_packet_bound(len(payload))
    out = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    buf = (C.c_uint8 * max(len(payload), 1)).from_buffer_copy(payload or b"\0")
    rc = L.psdr_wire_audio_packet(frame_num, l, m, r, pwr, buf, len(payload), out, cap, C.byref(n))
    assert rc == 0
    return bytes(out[: n.value])


def _wf(frame_num, l, r, payload):
    L = _lib.load()
    cap = L.psdr_wire_packet_bound(len(payload))
    out = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    buf = (C.c_uint8 * max(len(payload), 1)).from_buffer_copy(payload or b"\0")
    assert L.psdr_wire_waterfall_packet(frame_num, l, r, buf, len(payload), out, cap, C.byref(n)) == 0
    return bytes(out[: n.value])


def cbor_decode(b):
    """minimal RFC 8949 decoder (what the browser's CBOR library does with the packet)"""
    def item(i):
        ib = b[i]
        major, info = ib >> 5, ib & 31
        i += 1
        if major == 7:
            if info == 25:
                return np.frombuffer(b[i:i + 2][::-1], np.float16)[0].item(), i + 2
            if info == 26:
                return struct.unpack(">f", b[i:i + 4])[0], i + 4
            if info == 27:
                return struct.unpack(">d", b[i:i + 8])[0], i + 8
            raise ValueError(info)
        if info < 24:
            v = info
        else:
            nb = {24: 1, 25: 2, 26: 4, 27: 8}[info]
            v = int.from_bytes(b[i:i + nb], "big")
            i += nb
        if major == 0:
            return v, i
        if major == 1:
            return -1 - v, i
        if major == 2:
            return bytes(b[i:i + v]), i + v
        if major == 3:
            return b[i:i + v].decode(), i + v
        if major == 5:
            d = {}
            for _ in range(v):
                k, i = item(i)
                d[k], i = item(i)
            return d, i
        raise ValueError(major)
    v, i = item(0)
    assert i == len(b), "trailing bytes"
    return v


def test_audio_packet_known_answer():
    # {"data": h'010203', "frame_num": 5, "l": 100, "m": 100.5, "pwr": 0.25, "r": 189}
    want = bytes([0xA6,
                  0x64]) + b"data" + bytes([0x43, 1, 2, 3,
                  0x69]) + b"frame_num" + bytes([0x05,
                  0x61]) + b"l" + bytes([0x18, 100,
                  0x61]) + b"m" + bytes([0xFA]) + struct.pack(">f", 100.5) + bytes([
                  0x63]) + b"pwr" + bytes([0xFA]) + struct.pack(">f", 0.25) + bytes([
                  0x61]) + b"r" + bytes([0x18, 189])
    assert _audio(5, 100, 100.5, 189, 0.25, bytes([1, 2, 3])) == want


def test_audio_packet_number_forms():
    """shortest integer heads, negative ints, binary64 when the double does not survive binary32,
    binary16 NaN/inf (nlohmann write_cbor / write_compact_float)"""
    p = _audio(1 << 40, -3, 0.1, 70000, float("nan"), b"")
    d = cbor_decode(p)
    assert list(d) == ["data", "frame_num", "l", "m", "pwr", "r"]          # std::map order
    assert d["data"] == b"" and d["frame_num"] == 1 << 40 and d["l"] == -3 and d["r"] == 70000
    assert d["m"] == 0.1 and np.isnan(d["pwr"])
    assert bytes([0x61]) + b"m" + bytes([0xFB]) + struct.pack(">d", 0.1) in p     # 0.1 needs binary64
    assert bytes([0x63]) + b"pwr" + bytes([0xF9, 0x7E, 0x00]) in p
    assert bytes([0x69]) + b"frame_num" + bytes([0x1B]) + (1 << 40).to_bytes(8, "big") in p
    assert bytes([0x61]) + b"l" + bytes([0x22]) in p                             # -3 -> 0x20 + 2
    assert bytes([0x61]) + b"r" + bytes([0x1A]) + (70000).to_bytes(4, "big") in p
    p = _audio(23, 24, -np.inf, 255, 1e39, b"x" * 300)
    d = cbor_decode(p)
    assert d["frame_num"] == 23 and d["l"] == 24 and d["r"] == 255 and d["m"] == -np.inf and d["pwr"] == 1e39
    assert p[:1] == b"\xA6" and p[6:9] == bytes([0x59, 0x01, 0x2C])               # 300-byte string: 2-byte length
    assert bytes([0xF9, 0xFC, 0x00]) in p and bytes([0xFB]) + struct.pack(">d", 1e39) in p
    # a float audio_mid that is exactly representable in binary32 is sent as binary32
    assert bytes([0xFA]) + struct.pack(">f", 524288.5) in _audio(0, 0, 524288.5, 0, 0.0, b"")


def test_waterfall_packet_and_zstd_stream():
    rng = np.random.default_rng(3)
    rows = [rng.integers(-128, 127, 1024, dtype=np.int8).tobytes() for _ in range(3)]
    pk = [_wf(6 * i, 4096, 4096 + 1024 * 4, rows[i]) for i in range(3)]
    want0 = bytes([0xA4, 0x64]) + b"data" + bytes([0x59, 0x04, 0x00]) + rows[0] + bytes([0x69]) + b"frame_num" + \
        bytes([0x00, 0x61]) + b"l" + bytes([0x19, 0x10, 0x00, 0x61]) + b"r" + bytes([0x19, 0x20, 0x00])
    assert pk[0] == want0
    for i in range(3):
        d = cbor_decode(pk[i])
        assert list(d) == ["data", "frame_num", "l", "r"] and d["data"] == rows[i] and d["frame_num"] == 6 * i
    L = _lib.load()
    zs = C.c_void_p()
    rc = L.psdr_wire_zstd_create(C.byref(zs))
    if rc == -6:
        pytest.skip("no libzstd on this host")
    assert rc == 0
    try:
        z = C.CDLL("libzstd.so.1")
        z.ZSTD_createDStream.restype = C.c_void_p
        z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        z.ZSTD_decompressStream.restype = C.c_size_t
        z.ZSTD_freeDStream.argtypes = [C.c_void_p]
        ds = z.ZSTD_createDStream()

        class Buf(C.Structure):
            _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
        sizes = []
        for i in range(3):
            cap = L.psdr_wire_zstd_bound(len(pk[i]))
            out = (C.c_uint8 * cap)()
            n = C.c_size_t(0)
            src = (C.c_uint8 * len(pk[i])).from_buffer_copy(pk[i])
            assert L.psdr_wire_zstd_flush(zs, src, len(pk[i]), out, cap, C.byref(n)) == 0
            sizes.append(n.value)
            # ONE decompression stream across the packets, like the browser: every flushed packet
            # decodes completely on arrival
            dst = (C.c_uint8 * (len(pk[i]) + 64))()
            ib, ob = Buf(C.cast(out, C.c_void_p), n.value, 0), Buf(C.cast(dst, C.c_void_p), len(dst), 0)
            z.ZSTD_decompressStream(ds, C.byref(ob), C.byref(ib))
            assert ib.pos == n.value and bytes(dst[: ob.pos]) == pk[i]
        assert sizes[1] < sizes[0]      # the stream carries its header (and history) across packets
        z.ZSTD_freeDStream(ds)
    finally:
        L.psdr_wire_zstd_destroy(zs)


# ---- text frames: hello (src/websocket.cpp:42-66) and commands (src/client.cpp:19-117) -------------
def _hello(**kw):
    import json
    L = _lib.load()
    h = _lib.Hello()
    for k, v in kw.items():
        setattr(h, k, v.encode() if isinstance(v, str) else float(v))
    out = C.create_string_buffer(1024)
    n = C.c_size_t(0)
    assert L.psdr_wire_hello_json(C.byref(h), out, 1024, C.byref(n)) == 0
    assert len(out.value) == n.value
    return out.value.decode(), json.loads(out.value)


def test_hello_json_is_the_sorted_map_of_doubles():
    txt, js = _hello(sps=2000000, audio_max_sps=12000, audio_max_fft=360, fft_size=1 << 20, fft_result_size=1 << 20,
                     waterfall_size=1024, basefreq=14000000, total_bandwidth=2000000, default_frequency=14074000,
                     default_l=38797, default_m=38797.5, default_r=40370, default_modulation="USB",
                     waterfall_compression="zstd", audio_compression="flac")
    # std::map order at both levels, integers without a fraction, the one fractional number in its shortest form
    assert txt == ('{"audio_compression":"flac","audio_max_fft":360,"audio_max_sps":12000,"basefreq":14000000,'
                   '"defaults":{"frequency":14074000,"l":38797,"m":38797.5,"modulation":"USB","r":40370},'
                   '"fft_result_size":1048576,"fft_size":1048576,"sps":2000000,"total_bandwidth":2000000,'
                   '"waterfall_compression":"zstd","waterfall_size":1024}')
    assert list(js) == sorted(js) and list(js["defaults"]) == sorted(js["defaults"])
    # shortest round-trip form of a value that is not exact in binary
    _, js = _hello(default_m=0.1, default_modulation="AM", waterfall_compression="zstd", audio_compression="opus")
    assert js["defaults"]["m"] == 0.1 and '"m":0.1,' in _hello(default_m=0.1, default_modulation="AM",
                                                            waterfall_compression="z", audio_compression="o")[0]
    # too small a buffer is an error, the needed length is still reported
    L = _lib.load()
    h = _lib.Hello()
    h.default_modulation = h.waterfall_compression = h.audio_compression = b"x"
    small = C.create_string_buffer(8)
    n = C.c_size_t(0)
    assert L.psdr_wire_hello_json(C.byref(h), small, 8, C.byref(n)) != 0 and n.value > 8


def _cmd(msg):
    L = _lib.load()
    c = _lib.Command()
    b = msg.encode()
    rc = L.psdr_wire_parse_command(b, len(b), C.byref(c))
    return rc, c


def test_command_frames():
    rc, c = _cmd('{"cmd":"window","l":100,"r":460,"m":280.5,"level":3}')
    assert rc == 0 and (c.cmd, c.l, c.r, c.has_m, c.m, c.has_level, c.level) == (0, 100, 460, 1, 280.5, 1, 3)
    rc, c = _cmd(' { "l" : -5 , "cmd" : "window" , "r" : 7 } ')        # tag anywhere, optionals absent
    assert rc == 0 and (c.cmd, c.l, c.r, c.has_m, c.has_level) == (0, -5, 7, 0, 0)
    rc, c = _cmd('{"cmd":"window","l":1,"r":2,"m":null,"level":null}')  # null = std::nullopt
    assert rc == 0 and (c.has_m, c.has_level) == (0, 0)
    rc, c = _cmd('{"cmd":"window","r":2}')                              # a missing key keeps its default
    assert rc == 0 and (c.l, c.r) == (0, 2)
    rc, c = _cmd('{"cmd":"demodulation","demodulation":"LSB"}')
    assert rc == 0 and c.cmd == 1 and c.text == b"LSB"
    rc, c = _cmd('{"cmd":"userid","userid":"%s"}' % ("u" * 40))
    assert rc == 0 and c.cmd == 2 and c.text == b"u" * 32              # substr(0, 32), src/client.cpp:121
    rc, c = _cmd('{"cmd":"userid","userid":"a\\"b\\u00e9"}')
    assert rc == 0 and c.text == 'a"b\u00e9'.encode("utf-8")
    rc, c = _cmd('{"cmd":"mute","mute":true}')
    assert rc == 0 and c.cmd == 3 and c.mute == 1
    rc, c = _cmd('{"cmd":"mute","mute":false}')
    assert rc == 0 and c.mute == 0
    # the alternative is decided by the FIRST deciding key: the tag, or a key only one alternative has (every key of
    # these four structs is such a key); a tag that comes later is skipped
    rc, c = _cmd('{"l":1,"r":2}')
    assert rc == 0 and (c.cmd, c.l, c.r) == (0, 1, 2)
    rc, c = _cmd('{"mute":true}')
    assert rc == 0 and (c.cmd, c.mute) == (3, 1)
    rc, c = _cmd('{"demodulation":"AM","cmd":"window"}')
    assert rc == 0 and c.cmd == 1 and c.text == b"AM"
    rc, c = _cmd('{"cmd":"window","l":1e2,"r":-0,"m":1.25e1}')          # JSON numbers: exponents, -0
    assert rc == 0 and (c.l, c.r, c.m) == (100, 0, 12.5)
    # rejected like `if (ec) return` (src/client.cpp:96-99)
    for bad in ['', 'window', '{}', '{"cmd":"window","l":1,"r":2', '{"cmd":"zoom","l":1,"r":2}', '{"l":1,"mute":true}',
                '{"cmd":"window","l":1.5,"r":2}', '{"cmd":"window","l":"1","r":2}', '{"cmd":"window","l":1,"r":2,"x":0}',
                '{"cmd":"mute","mute":1}', '{"cmd":"mute","mute":true,"l":1}', '{"cmd":"demodulation","demodulation":7}',
                '{"cmd":"window","l":1,"r":2} trailing', '{"cmd":"window","l":3000000000,"r":2}', '{"l":1,"cmd":7}',
                '{"cmd":"window","l":01,"r":2}', '{"cmd":"window","l":1.,"r":2}', '{"cmd":"window","l":+1,"r":2}',
                '{"cmd":"window","l":1e,"r":2}', '{"cmd":"window","l":.5,"r":2}', '{"cmd":"window","l":-,"r":2}']:
        assert _cmd(bad)[0] != 0, bad
