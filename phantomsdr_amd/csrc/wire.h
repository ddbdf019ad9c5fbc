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
// The two text formats of the same section: the hello frame (send_basic_info, src/websocket.cpp:42-66)
// and the command frame (Client::on_message, src/client.cpp:19-117), both glaze v2.4.4 JSON.
// zstd: one ZSTD_CStream per client, ZSTD_compressStream2(..., ZSTD_e_flush) per packet
// (src/waterfallcompression.cpp:33); libzstd is dlopen()ed (it is a system library, not part of this
// repository), absent -> PSDR_ERR_UNSUPPORTED.
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <string>

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

// ---- hello / command text frames --------------------------------------------------------------
namespace psdr_wire {

// shortest decimal form that reads back to the same double; integers without a fraction
inline std::string json_number(double v) {
    if (!std::isfinite(v)) return "null";  // glaze writes non-finite numbers as null
    char buf[40];
    if (v == std::floor(v) && std::fabs(v) < 9007199254740992.0) {
        snprintf(buf, sizeof buf, "%.0f", v);
        return buf[0] == '-' && buf[1] == '0' && !buf[2] ? "0" : buf;
    }
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (strtod(buf, nullptr) == v) break;
    }
    return buf;
}
inline std::string json_string(const char *s) {
    std::string o = "\"";
    for (; s && *s; s++) {
        if (*s == '"' || *s == '\\') o += '\\';
        o += *s;
    }
    return o + "\"";
}

// the few productions of JSON the command frames use: one flat object of strings, numbers, booleans, null
struct Cursor {
    const char *p, *e;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool eat(char c) {
        ws();
        if (p < e && *p == c) {
            p++;
            return true;
        }
        return false;
    }
    bool lit(const char *w) {
        const size_t n = strlen(w);
        if ((size_t)(e - p) >= n && !memcmp(p, w, n)) {
            p += n;
            return true;
        }
        return false;
    }
    bool str(std::string &o) {
        ws();
        if (p >= e || *p != '"') return false;
        p++;
        o.clear();
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return false;
                switch (*p) {
                case '"': o += '"'; break;
                case '\\': o += '\\'; break;
                case '/': o += '/'; break;
                case 'b': o += '\b'; break;
                case 'f': o += '\f'; break;
                case 'n': o += '\n'; break;
                case 'r': o += '\r'; break;
                case 't': o += '\t'; break;
                case 'u': {  // BMP code point -> UTF-8 (surrogate pairs are not combined)
                    if (e - p < 5) return false;
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; i++) {
                        const char c = p[i];
                        const int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
                        if (d < 0) return false;
                        cp = cp * 16 + (unsigned)d;
                    }
                    p += 4;
                    if (cp < 0x80) {
                        o += (char)cp;
                    } else if (cp < 0x800) {
                        o += (char)(0xC0 | (cp >> 6));
                        o += (char)(0x80 | (cp & 0x3F));
                    } else {
                        o += (char)(0xE0 | (cp >> 12));
                        o += (char)(0x80 | ((cp >> 6) & 0x3F));
                        o += (char)(0x80 | (cp & 0x3F));
                    }
                    break;
                }
                default: return false;
                }
                p++;
            } else {
                o += *p++;
            }
        }
        if (p >= e) return false;
        p++;
        return true;
    }
    // a JSON number, strictly: -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?  (strtod alone would take "01", "1." or "1e")
    bool num(double &v) {
        ws();
        const char *q = p;
        auto digit = [&](const char *x) { return x < e && *x >= '0' && *x <= '9'; };
        if (q < e && *q == '-') q++;
        if (!digit(q)) return false;
        if (*q == '0') {
            q++;
            if (digit(q)) return false;  // leading zero
        } else {
            while (digit(q)) q++;
        }
        if (q < e && *q == '.') {
            q++;
            if (!digit(q)) return false;
            while (digit(q)) q++;
        }
        if (q < e && (*q == 'e' || *q == 'E')) {
            q++;
            if (q < e && (*q == '+' || *q == '-')) q++;
            if (!digit(q)) return false;
            while (digit(q)) q++;
        }
        const std::string t(p, q);
        char *end = nullptr;
        v = strtod(t.c_str(), &end);
        if (end != t.c_str() + t.size()) return false;
        p = q;
        return true;
    }
};

}  // namespace psdr_wire

extern "C" int psdr_wire_hello_json(const psdr_hello *h, char *out, size_t cap, size_t *len) {
    if (!h || !out || !len) return PSDR_ERR_INVALID;
    using psdr_wire::json_number;
    using psdr_wire::json_string;
    std::string s = "{";
    s += "\"audio_compression\":" + json_string(h->audio_compression);
    s += ",\"audio_max_fft\":" + json_number(h->audio_max_fft);
    s += ",\"audio_max_sps\":" + json_number(h->audio_max_sps);
    s += ",\"basefreq\":" + json_number(h->basefreq);
    s += ",\"defaults\":{\"frequency\":" + json_number(h->default_frequency);
    s += ",\"l\":" + json_number(h->default_l);
    s += ",\"m\":" + json_number(h->default_m);
    s += ",\"modulation\":" + json_string(h->default_modulation);
    s += ",\"r\":" + json_number(h->default_r) + "}";
    s += ",\"fft_result_size\":" + json_number(h->fft_result_size);
    s += ",\"fft_size\":" + json_number(h->fft_size);
    s += ",\"sps\":" + json_number(h->sps);
    s += ",\"total_bandwidth\":" + json_number(h->total_bandwidth);
    s += ",\"waterfall_compression\":" + json_string(h->waterfall_compression);
    s += ",\"waterfall_size\":" + json_number(h->waterfall_size) + "}";
    *len = s.size();
    if (s.size() + 1 > cap) return PSDR_ERR_INVALID;
    memcpy(out, s.c_str(), s.size() + 1);
    return PSDR_OK;
}

extern "C" int psdr_wire_parse_command(const char *msg, size_t len, psdr_command *out) {
    if (!msg || !out) return PSDR_ERR_INVALID;
    psdr_wire::Cursor c{msg, msg + len};
    psdr_command r;
    memset(&r, 0, sizeof r);
    r.cmd = -1;
    // Client::on_message reads a std::variant of four structs tagged by "cmd" (src/client.cpp:19-89).  glaze picks the
    // alternative from the FIRST key that decides it: the tag, or a key that only one alternative has - and every key
    // of these four structs belongs to exactly one of them.  The object is then read as that struct: its own keys in
    // any order, missing ones keep their defaults, the tag key is skipped wherever it stands (whatever it says), any
    // other key is an error; an object without a deciding key matches no alternative.  (glaze 2.4.4 is not in the
    // image: restated from its documented variant handling, pinned only by tests/test_wire_formats.py.)
    int sel = -1;  // PSDR_CMD_*
    bool has_m = false, has_lv = false;
    double l = 0, rr = 0, m = 0, lv = 0;
    std::string dem, uid;
    auto owner = [](const std::string &k) {
        if (k == "l" || k == "r" || k == "m" || k == "level") return (int)PSDR_CMD_WINDOW;
        if (k == "demodulation") return (int)PSDR_CMD_DEMODULATION;
        if (k == "userid") return (int)PSDR_CMD_USERID;
        if (k == "mute") return (int)PSDR_CMD_MUTE;
        return -1;
    };
    if (!c.eat('{')) return PSDR_ERR_INVALID;
    if (!c.eat('}')) {
        for (;;) {
            std::string key;
            if (!c.str(key) || !c.eat(':')) return PSDR_ERR_INVALID;
            c.ws();
            if (key == "cmd") {
                std::string cmd;
                if (!c.str(cmd)) return PSDR_ERR_INVALID;
                if (sel < 0) {
                    sel = cmd == "window" ? PSDR_CMD_WINDOW : cmd == "demodulation" ? PSDR_CMD_DEMODULATION
                        : cmd == "userid" ? PSDR_CMD_USERID : cmd == "mute" ? PSDR_CMD_MUTE : -1;
                    if (sel < 0) return PSDR_ERR_INVALID;
                }
            } else {
                const int own = owner(key);
                if (own < 0) return PSDR_ERR_INVALID;  // unknown key
                if (sel < 0) sel = own;
                if (own != sel) return PSDR_ERR_INVALID;  // a key of another alternative
                if (key == "l") {
                    if (!c.num(l)) return PSDR_ERR_INVALID;
                } else if (key == "r") {
                    if (!c.num(rr)) return PSDR_ERR_INVALID;
                } else if (key == "m") {
                    if (c.lit("null"))
                        has_m = false;
                    else if (c.num(m))
                        has_m = true;
                    else
                        return PSDR_ERR_INVALID;
                } else if (key == "level") {
                    if (c.lit("null"))
                        has_lv = false;
                    else if (c.num(lv))
                        has_lv = true;
                    else
                        return PSDR_ERR_INVALID;
                } else if (key == "demodulation") {
                    if (!c.str(dem)) return PSDR_ERR_INVALID;
                } else if (key == "userid") {
                    if (!c.str(uid)) return PSDR_ERR_INVALID;
                } else {  // mute
                    if (c.lit("true"))
                        r.mute = 1;
                    else if (c.lit("false"))
                        r.mute = 0;
                    else
                        return PSDR_ERR_INVALID;
                }
            }
            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return PSDR_ERR_INVALID;
        }
    }
    c.ws();
    if (c.p != c.e) return PSDR_ERR_INVALID;
    if (sel < 0) return PSDR_ERR_INVALID;  // {}: no alternative
    auto as_int = [](double v, int32_t &o) {  // an int field: a fraction or an out-of-range value is a parse error
        if (v != std::floor(v) || v < -2147483648.0 || v > 2147483647.0) return false;
        o = (int32_t)v;
        return true;
    };
    r.cmd = sel;
    if (sel == PSDR_CMD_WINDOW) {
        if (!as_int(l, r.l) || !as_int(rr, r.r)) return PSDR_ERR_INVALID;
        r.has_m = has_m;
        r.m = m;
        r.has_level = has_lv;
        if (has_lv && !as_int(lv, r.level)) return PSDR_ERR_INVALID;
    } else if (sel == PSDR_CMD_DEMODULATION) {
        snprintf(r.text, sizeof r.text, "%.32s", dem.c_str());
    } else if (sel == PSDR_CMD_USERID) {
        snprintf(r.text, sizeof r.text, "%.32s", uid.c_str());
    }
    *out = r;
    return PSDR_OK;
}
