// wire.h — the reference's wire formats for audio and waterfall packets, host side.
//   AudioEncoder::set_data/send        src/audio.cpp:17-36
//   WaterfallEncoder::set_data, ZstdEncoder::send   src/waterfallcompression.cpp:13-37
// A packet is a CBOR map produced by nlohmann::json::to_cbor (v3.11.2, subprojects/nlohmann_json.wrap)
// from a json object, i.e. a std::map: keys in lexicographic order,
//   audio      {"data": bytes, "frame_num": uint, "l": int, "m": float, "pwr": float, "r": int}
//   waterfall  {"data": bytes, "frame_num": uint, "l": int, "r": int}     then zstd-streamed
// with to_cbor's encodings: the shortest integer head (RFC 8949 3.1), floats as binary32 when the
// double survives the round trip through float, else binary64 (NaN/inf as binary16), byte strings
// without a tag.  The payload ("data") is opaque here: FLAC/Opus frames or int8 waterfall rows.
// zstd: one ZSTD_CStream per client, ZSTD_compressStream2(..., ZSTD_e_flush) per packet
// (src/waterfallcompression.cpp:33); libzstd is dlopen()ed (it is a system library, not part of this
// repository), absent -> PSDR_ERR_UNSUPPORTED.
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <limits>
#include <mutex>

#include "../../include/psdr.h"

namespace psdr_wire {

struct Out {
    uint8_t *p;
    size_t cap, n;
    bool ok;
    void put(uint8_t b) {
        if (n < cap)
            p[n] = b;
        else
            ok = false;
        n++;
    }
    void be(uint64_t v, int bytes) {
        for (int i = bytes - 1; i >= 0; i--) put((uint8_t)(v >> (8 * i)));
    }
    // major type + argument, shortest form
    void head(unsigned major, uint64_t v) {
        const uint8_t m = (uint8_t)(major << 5);
        if (v <= 0x17) {
            put(m | (uint8_t)v);
        } else if (v <= 0xFF) {
            put(m | 0x18);
            be(v, 1);
        } else if (v <= 0xFFFF) {
            put(m | 0x19);
            be(v, 2);
        } else if (v <= 0xFFFFFFFFull) {
            put(m | 0x1A);
            be(v, 4);
        } else {
            put(m | 0x1B);
            be(v, 8);
        }
    }
    void key(const char *s) {
        const size_t len = strlen(s);
        head(3, len);
        for (size_t i = 0; i < len; i++) put((uint8_t)s[i]);
    }
    void integer(int64_t v) {  // number_integer: >= 0 like an unsigned, < 0 as -1 - n
        if (v >= 0)
            head(0, (uint64_t)v);
        else
            head(1, (uint64_t)(-1 - v));
    }
    void number(double v) {  // number_float: write_compact_float
        if (std::isnan(v)) {
            put(0xF9), put(0x7E), put(0x00);
        } else if (std::isinf(v)) {
            put(0xF9), put(v < 0 ? 0xFC : 0x7C), put(0x00);
        } else if (v >= (double)std::numeric_limits<float>::lowest() && v <= (double)std::numeric_limits<float>::max() &&
                   (double)(float)v == v) {
            const float f = (float)v;
            uint32_t u;
            memcpy(&u, &f, 4);
            put(0xFA);
            be(u, 4);
        } else {
            uint64_t u;
            memcpy(&u, &v, 8);
            put(0xFB);
            be(u, 8);
        }
    }
    void bytes(const void *d, size_t len) {
        head(2, len);
        if (n + len <= cap && ok)
            memcpy(p + n, d, len);
        else
            ok = false;
        n += len;
    }
};

struct ZstdApi {
    void *h = nullptr;
    void *(*createCStream)() = nullptr;
    size_t (*freeCStream)(void *) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    size_t (*compressStream2)(void *, void *, void *, int) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool load() {
        static std::once_flag once;
        std::call_once(once, [this]() {
            for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
            if (!h) return;
            createCStream = (void *(*)())dlsym(h, "ZSTD_createCStream");
            freeCStream = (size_t(*)(void *))dlsym(h, "ZSTD_freeCStream");
            compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
            compressStream2 = (size_t(*)(void *, void *, void *, int))dlsym(h, "ZSTD_compressStream2");
            isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        });
        return h && createCStream && freeCStream && compressBound && compressStream2 && isError;
    }
};
inline ZstdApi &zstd() {
    static ZstdApi z;
    return z;
}

}  // namespace psdr_wire

struct psdr_zstd {
    void *stream;
};

extern "C" size_t psdr_wire_packet_bound(size_t payload_bytes) { return payload_bytes + 96; }

extern "C" int psdr_wire_audio_packet(uint64_t frame_num, int l, double m, int r, double pwr, const void *payload,
                                      size_t bytes, uint8_t *out, size_t cap, size_t *len) {
    if ((!payload && bytes) || !out || !len) return PSDR_ERR_INVALID;
    psdr_wire::Out o{out, cap, 0, true};
    o.head(5, 6);
    o.key("data");
    o.bytes(payload, bytes);
    o.key("frame_num");
    o.head(0, frame_num);
    o.key("l");
    o.integer(l);
    o.key("m");
    o.number(m);
    o.key("pwr");
    o.number(pwr);
    o.key("r");
    o.integer(r);
    *len = o.n;
    return o.ok ? PSDR_OK : PSDR_ERR_INVALID;
}

extern "C" int psdr_wire_waterfall_packet(uint64_t frame_num, int l, int r, const void *payload, size_t bytes,
                                          uint8_t *out, size_t cap, size_t *len) {
    if ((!payload && bytes) || !out || !len) return PSDR_ERR_INVALID;
    psdr_wire::Out o{out, cap, 0, true};
    o.head(5, 4);
    o.key("data");
    o.bytes(payload, bytes);
    o.key("frame_num");
    o.head(0, frame_num);
    o.key("l");
    o.integer(l);
    o.key("r");
    o.integer(r);
    *len = o.n;
    return o.ok ? PSDR_OK : PSDR_ERR_INVALID;
}

extern "C" int psdr_wire_zstd_create(psdr_zstd **out) {
    if (!out) return PSDR_ERR_INVALID;
    auto &z = psdr_wire::zstd();
    if (!z.load()) return PSDR_ERR_UNSUPPORTED;
    void *s = z.createCStream();
    if (!s) return PSDR_ERR_NOMEM;
    *out = new psdr_zstd{s};
    return PSDR_OK;
}
extern "C" void psdr_wire_zstd_destroy(psdr_zstd *zs) {
    if (!zs) return;
    psdr_wire::zstd().freeCStream(zs->stream);
    delete zs;
}
extern "C" size_t psdr_wire_zstd_bound(size_t n) {
    auto &z = psdr_wire::zstd();
    return z.load() ? z.compressBound(n) : 0;
}
extern "C" int psdr_wire_zstd_flush(psdr_zstd *zs, const void *in, size_t n, uint8_t *out, size_t cap, size_t *len) {
    if (!zs || (!in && n) || !out || !len) return PSDR_ERR_INVALID;
    auto &z = psdr_wire::zstd();
    struct {
        const void *src;
        size_t size, pos;
    } ib = {in, n, 0};
    struct {
        void *dst;
        size_t size, pos;
    } ob = {out, cap, 0};
    const size_t rc = z.compressStream2(zs->stream, &ob, &ib, 1 /* ZSTD_e_flush */);
    if (z.isError(rc) || rc != 0 || ib.pos != n) return PSDR_ERR_INVALID;  // rc > 0: the output buffer was too small
    *len = ob.pos;
    return PSDR_OK;
}
