// container.hip — host-side header walking of aukit.wav / aukit.aiff / aukit.au (aukit.lua:1456-1651) and of their stream.* twins
// (:2927-3113): bytes in, a codec descriptor + the payload's byte range out.  No kernel here — the file is compiled by hipcc with the
// rest of the library so that every host language binds ONE shared object (the LuaJIT shim and the Python mirror both call
// aukit_parse_container and then aukit_decode / aukit_stream_decode on the payload).
//
// What is reproduced of the Lua, because callers can observe it:
//   * `string.unpack` raising on a truncated chunk header ("data string too short") — aukit.wav loops `while pos <= #data` and reads 8
//     bytes: a file with 1-7 stray bytes behind its last chunk raises, it does not end quietly (:1466-1468);
//   * aukit.wav keeps walking after a `data` chunk (a later `data` chunk replaces the audio; `LIST`/`INFO` is walked entry by entry and a
//     malformed one raises, :1553-1563), stream.wav returns at the first `data` chunk (:2980);
//   * WAVE_FORMAT_EXTENSIBLE (0xFFFE): bit depth from byte 18 of the chunk, sub-format GUID at bytes 24-39 (:1493-1503);
//   * AIFF: COMM is read with a fixed 18-byte layout and `pos` continues behind THOSE bytes, not behind the chunk (:1601); the 80-bit
//     rate as the Lua computes it (:1603-1605); `sowt` is little-endian for aukit.aiff (:1613) and big-endian for stream.aiff (:3065);
//     aukit.aiff does not mind a short SSND chunk (:1610), stream.aiff raises (:3046);
//   * AU: the header's 0-based `offset` is used as a 1-based `string.sub` index, so one header byte becomes a sample (:1643, :3094).
// Metadata (LIST / NAME / AUTH ...) is walked for its effect on `pos` and errors only; its strings are not returned (out of scope).
#include <cmath>
#include "common.h"

namespace aukit {
namespace {

struct Rd {  // a Lua string with 1-based positions
    const uint8_t *p;
    uint64_t n;
    bool ok = true;
    bool have(uint64_t pos, uint64_t k) const { return pos >= 1 && pos - 1 <= n && k <= n - (pos - 1); }  // string.unpack's "data string too short"
    uint32_t u32le(uint64_t pos) { if (!have(pos, 4)) { ok = false; return 0; } const uint8_t *q = p + pos - 1; return q[0] | q[1] << 8 | q[2] << 16 | (uint32_t)q[3] << 24; }
    uint32_t u32be(uint64_t pos) { if (!have(pos, 4)) { ok = false; return 0; } const uint8_t *q = p + pos - 1; return (uint32_t)q[0] << 24 | q[1] << 16 | q[2] << 8 | q[3]; }
    uint16_t u16le(uint64_t pos) { if (!have(pos, 2)) { ok = false; return 0; } const uint8_t *q = p + pos - 1; return (uint16_t)(q[0] | q[1] << 8); }
    uint16_t u16be(uint64_t pos) { if (!have(pos, 2)) { ok = false; return 0; } const uint8_t *q = p + pos - 1; return (uint16_t)(q[0] << 8 | q[1]); }
    bool tag(uint64_t pos, const char *t) { if (!have(pos, 4)) { ok = false; return false; } return memcmp(p + pos - 1, t, 4) == 0; }
};
const char *kShort = "bad argument #2 to 'unpack' (data string too short)";

struct Guid { uint8_t b[16]; int dt; };
// wavExtensible (:131-139)
const Guid kGuids[] = {
    {{0x01, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_SIGNED /* pcm: signed, or unsigned at 8 bits */},
    {{0x02, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_MSADPCM},
    {{0x03, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_FLOAT},
    {{0x06, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_ALAW},
    {{0x07, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_ULAW},
    {{0x11, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71}, AUKIT_WAVDT_ADPCM},
    {{0x3a, 0xc1, 0xfa, 0x38, 0x81, 0x1d, 0x43, 0x61, 0xa4, 0x0d, 0xce, 0x53, 0xca, 0x60, 0x7c, 0xd1}, AUKIT_WAVDT_DFPWM},
};

void desc_from_wav(aukit_container *o, int dt, int channels, double rate, int bits, int block_align) {
    aukit_codec_desc &d = o->desc;
    d.channels = channels; d.sample_rate = rate; d.interleaved = 1; d.big_endian = 0;
    switch (dt) {
    case AUKIT_WAVDT_ADPCM: d.codec = AUKIT_CODEC_ADPCM_WAV; d.block_align = block_align; break;
    case AUKIT_WAVDT_MSADPCM: d.codec = AUKIT_CODEC_MSADPCM; d.block_align = block_align; break;
    case AUKIT_WAVDT_ALAW: d.codec = AUKIT_CODEC_G711; d.ulaw = 0; break;
    case AUKIT_WAVDT_ULAW: d.codec = AUKIT_CODEC_G711; d.ulaw = 1; break;
    case AUKIT_WAVDT_DFPWM: d.codec = AUKIT_CODEC_DFPWM; break;
    default:
        d.codec = AUKIT_CODEC_PCM; d.bit_depth = bits;
        d.data_type = dt == AUKIT_WAVDT_FLOAT ? AUKIT_FLOAT : (dt == AUKIT_WAVDT_UNSIGNED ? AUKIT_UNSIGNED : AUKIT_SIGNED);
    }
    o->wav_data_type = dt;
    o->bit_depth = bits;
}

// str_sub(data, i, j) with 1-based inclusive bounds → 0-based [off, off + len)
void lua_sub(uint64_t n, double i, double j, uint64_t *off, uint64_t *len) {
    if (i < 1) i = 1;
    if (j > (double)n) j = (double)n;
    if (i > j) { *off = (uint64_t)std::min<double>(i - 1, (double)n); *len = 0; return; }
    *off = (uint64_t)i - 1;
    *len = (uint64_t)j - (uint64_t)i + 1;
}

// stream: 0 = aukit.wav, 1 = aukit.stream.wav on a string, 2 = on the first piece of a reader function (`fn` set: the size checks of
// :2977 / :3046 do not apply, the payload is what the piece holds)
int parse_wav(const uint8_t *bytes, uint64_t n, int stream, aukit_container *o) {
    Rd r{bytes, n};
    if (!r.have(1, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
    if (memcmp(bytes, "RIFF", 4) != 0) return fail(AUKIT_E_ARG, "bad argument #1 (not a WAV file)");
    if (!r.have(9, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
    if (memcmp(bytes + 8, "WAVE", 4) != 0) return fail(AUKIT_E_ARG, "bad argument #1 (not a WAV file)");
    uint64_t pos = 13;
    bool have_fmt = false, have_obj = false;
    int dt = 0, channels = 0, bits = 0, block_align = 0;
    double rate = 0;
    while (pos <= n) {
        if (!r.have(pos, 8)) return fail(AUKIT_E_LUA, "%s", kShort);  // str_unpack("<c4I", data, pos)
        const uint8_t *tg = bytes + pos - 1;
        const uint64_t size = r.u32le(pos + 4);
        pos += 8;
        if (memcmp(tg, "fmt ", 4) == 0) {
            uint64_t coff, clen;
            lua_sub(n, (double)pos, (double)pos + (double)size - 1, &coff, &clen);
            Rd c{bytes + coff, clen};
            pos += size;
            if (!c.have(1, 16)) return fail(AUKIT_E_LUA, "%s", kShort);  // "<HHIxxxxHH"
            const int format = c.u16le(1);
            channels = c.u16le(3);
            rate = (double)c.u32le(5);
            block_align = c.u16le(13);
            bits = c.u16le(15);
            o->desc.ncoef = 0;
            if (format == 1) dt = bits == 8 ? AUKIT_WAVDT_UNSIGNED : AUKIT_WAVDT_SIGNED;
            else if (format == 2) {
                dt = AUKIT_WAVDT_MSADPCM;
                if (!c.have(21, 2)) return fail(AUKIT_E_LUA, "%s", kShort);
                const int nco = c.u16le(21);
                if (nco > 32) return fail(AUKIT_E_UNSUPPORTED, "MS-ADPCM with more than 32 coefficient pairs");
                for (int i = 1; i <= nco; i++) {  // str_unpack("<hh", chunk, i * 4 + 19)
                    if (!c.have((uint64_t)i * 4 + 19, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
                    o->desc.coef1[i - 1] = (int16_t)c.u16le((uint64_t)i * 4 + 19);
                    o->desc.coef2[i - 1] = (int16_t)c.u16le((uint64_t)i * 4 + 21);
                }
                o->desc.ncoef = nco;
            } else if (format == 3) dt = AUKIT_WAVDT_FLOAT;
            else if (format == 6) dt = AUKIT_WAVDT_ALAW;
            else if (format == 7) dt = AUKIT_WAVDT_ULAW;
            else if (format == 0x11) dt = AUKIT_WAVDT_ADPCM;
            else if (format == 0xFFFE) {
                if (!c.have(19, 2)) return fail(AUKIT_E_LUA, "%s", kShort);
                bits = c.u16le(19);
                dt = -1;
                if (clen >= 40)  // str_sub(chunk, 25, 40) of a shorter chunk equals no GUID
                    for (const Guid &g : kGuids)
                        if (memcmp(c.p + 24, g.b, 16) == 0) dt = g.dt;
                if (dt < 0) return fail(AUKIT_E_LUA, "unsupported WAV file");
                if (dt == AUKIT_WAVDT_SIGNED && bits == 8) dt = AUKIT_WAVDT_UNSIGNED;
            } else return fail(AUKIT_E_LUA, "unsupported WAV file");
            have_fmt = true;
        } else if (memcmp(tg, "data", 4) == 0) {
            uint64_t off, len;
            lua_sub(n, (double)pos, (double)pos + (double)size - 1, &off, &len);
            if (len < size && stream != 2) return fail(AUKIT_E_LUA, "invalid WAV file");
            if (!have_fmt) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");  // aukit.pcm(data, nil, ...): the first check that fails
            desc_from_wav(o, dt, channels, rate, bits, block_align);
            o->payload_off = off; o->payload_len = len;
            o->length_seconds = (dt == AUKIT_WAVDT_DFPWM || o->desc.codec == AUKIT_CODEC_PCM) ? (double)size / channels / ((double)bits / 8) / rate : std::nan("");  // :2994, :2996
            have_obj = true;
            if (stream) return AUKIT_OK;
            pos += size;
        } else if (!stream && memcmp(tg, "LIST", 4) == 0) {
            if (!r.have(pos, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
            if (memcmp(bytes + pos - 1, "INFO", 4) == 0) {
                const uint64_t e = pos + size;
                pos += 4;
                while (pos < e) {  // str_unpack("!2<c4s4Xh", data, pos): tag, 4-byte length + bytes, align to 2
                    if (!r.have(pos, 8)) return fail(AUKIT_E_LUA, "%s", kShort);
                    const uint64_t sl = r.u32le(pos + 4);
                    if (!r.have(pos + 8, sl)) return fail(AUKIT_E_LUA, "%s", kShort);
                    pos += 8 + sl;
                    if ((pos - 1) & 1) {  // Xh: padding to a multiple of 2 (0-based offset)
                        if (!r.have(pos, 1)) return fail(AUKIT_E_LUA, "%s", kShort);
                        pos += 1;
                    }
                }
            } else pos += size;
        } else pos += size;
    }
    if (have_obj) return AUKIT_OK;
    return fail(AUKIT_E_LUA, "invalid WAV file");
}

int parse_aiff(const uint8_t *bytes, uint64_t n, int stream, aukit_container *o) {
    Rd r{bytes, n};
    if (!r.have(1, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
    if (memcmp(bytes, "FORM", 4) != 0) return fail(AUKIT_E_ARG, "bad argument #1 (not an AIFF file)");
    if (!r.have(9, 4)) return fail(AUKIT_E_LUA, "%s", kShort);
    bool aifc = false;
    if (memcmp(bytes + 8, "AIFC", 4) == 0) aifc = true;
    else if (memcmp(bytes + 8, "AIFF", 4) != 0) return fail(AUKIT_E_ARG, "bad argument #1 (not an AIFF file)");
    uint64_t pos = 13;
    bool have_comm = false;
    int channels = 0, bits = 0;
    double rate = 0, length = 0;
    char comp[5] = {0, 0, 0, 0, 0};
    bool have_comp = false;
    while (pos <= n) {
        if (!r.have(pos, 8)) return fail(AUKIT_E_LUA, "%s", kShort);
        const uint8_t *tg = bytes + pos - 1;
        const uint64_t size = r.u32be(pos + 4);
        pos += 8;
        if (memcmp(tg, "COMM", 4) == 0) {
            if (!r.have(pos, 18)) return fail(AUKIT_E_LUA, "%s", kShort);  // ">hIhHI7x"
            channels = (int16_t)r.u16be(pos);
            const double frames = (double)r.u32be(pos + 2);
            bits = (int16_t)r.u16be(pos + 6);
            unsigned e = r.u16be(pos + 8);
            uint64_t m = 0;
            for (int i = 0; i < 7; i++) m = m << 8 | bytes[pos - 1 + 10 + i];
            pos += 18;
            if (aifc) {
                if (!r.have(pos, 5)) return fail(AUKIT_E_LUA, "%s", kShort);  // ">c4s1"
                memcpy(comp, bytes + pos - 1, 4);
                have_comp = true;
                const unsigned sl = bytes[pos - 1 + 4];
                if (!r.have(pos + 5, sl)) return fail(AUKIT_E_LUA, "%s", kShort);
                pos += 5 + sl;
                if (sl % 2 == 0) pos += 1;
            }
            length = frames * channels * std::floor((double)bits / 8);
            const bool neg = (e & 0x8000) != 0;
            const int ex = (int)((((int)(e & 0x7FFF) - 0x3FFE) % 0x800 + 0x800) % 0x800);  // Lua's % is floored
            rate = std::ldexp((double)m * (neg ? -1 : 1) / 72057594037927936.0, ex);        // m / 0x100000000000000  :1605
            have_comm = true;
        } else if (memcmp(tg, "SSND", 4) == 0) {
            if (!r.have(pos, 8)) return fail(AUKIT_E_LUA, "%s", kShort);  // ">II"
            const double offset = (double)r.u32be(pos);
            pos += 8;
            if (!have_comm) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (local 'length')");
            uint64_t off, len;
            lua_sub(n, (double)pos + offset, (double)pos + offset + length - 1, &off, &len);
            if (stream == 1 && (double)len < length) return fail(AUKIT_E_LUA, "invalid AIFF file");
            aukit_codec_desc &d = o->desc;
            d.channels = channels; d.sample_rate = rate; d.interleaved = 1;
            o->bit_depth = bits;
            o->payload_off = off; o->payload_len = len;
            const bool none = !have_comp || memcmp(comp, "NONE", 4) == 0, sowt = have_comp && memcmp(comp, "sowt", 4) == 0;
            if (none || sowt) {
                d.codec = AUKIT_CODEC_PCM; d.bit_depth = bits; d.data_type = AUKIT_SIGNED;
                d.big_endian = (none || stream) ? 1 : 0;  // stream.aiff hands `true` to stream.pcm for sowt as well (:3065)
                o->length_seconds = length / channels / ((double)bits / 8) / rate;
            } else if (memcmp(comp, "fl32", 4) == 0 || memcmp(comp, "FL32", 4) == 0) {
                d.codec = AUKIT_CODEC_PCM; d.bit_depth = 32; d.data_type = AUKIT_FLOAT; d.big_endian = 1;
                o->length_seconds = length / channels / 4 / rate;
            } else if (memcmp(comp, "alaw", 4) == 0 || memcmp(comp, "ALAW", 4) == 0 || memcmp(comp, "ulaw", 4) == 0 || memcmp(comp, "ULAW", 4) == 0) {
                d.codec = AUKIT_CODEC_G711; d.ulaw = (comp[0] == 'u' || comp[0] == 'U') ? 1 : 0;
                o->length_seconds = length / channels / rate;
            } else {
                char printable[5] = {comp[0], comp[1], comp[2], comp[3], 0};
                return fail(AUKIT_E_LUA, "Unsupported compression scheme %s", printable);
            }
            return AUKIT_OK;
        } else pos += size;  // NAME / AUTH / (c) / ANNO and everything else: skipped
    }
    return fail(AUKIT_E_LUA, "invalid AIFF file");
}

int parse_au(const uint8_t *bytes, uint64_t n, int, aukit_container *o) {
    Rd r{bytes, n};
    if (!r.have(1, 24)) return fail(AUKIT_E_LUA, "%s", kShort);  // ">c4IIIII"
    if (memcmp(bytes, ".snd", 4) != 0) return fail(AUKIT_E_LUA, "invalid AU file");
    const double offset = (double)r.u32be(5), size = (double)r.u32be(9);
    const unsigned enc = r.u32be(13);
    const double rate = (double)r.u32be(17);
    const int channels = (int)r.u32be(21);
    uint64_t off, len;
    lua_sub(n, offset, size != 4294967295.0 ? offset + size - 1 : (double)n, &off, &len);  // str_sub(data, offset, ...): the 0-based offset taken as a 1-based index
    aukit_codec_desc &d = o->desc;
    d.channels = channels; d.sample_rate = rate; d.interleaved = 1; d.big_endian = 1;
    o->payload_off = off; o->payload_len = len;
    if (enc == 1 || enc == 27) { d.codec = AUKIT_CODEC_G711; d.ulaw = enc == 1; o->bit_depth = 8; o->length_seconds = size / channels / rate; }
    else if (enc >= 2 && enc <= 5) { d.codec = AUKIT_CODEC_PCM; d.bit_depth = 8 * (int)(enc - 1); d.data_type = AUKIT_SIGNED; o->bit_depth = d.bit_depth; o->length_seconds = size / channels / (double)(enc - 1) / rate; }
    else if (enc == 6) { d.codec = AUKIT_CODEC_PCM; d.bit_depth = 32; d.data_type = AUKIT_FLOAT; o->bit_depth = 32; o->length_seconds = size / channels / 4 / rate; }
    else return fail(AUKIT_E_LUA, "unsupported encoding type %u", enc);
    return AUKIT_OK;
}

}  // namespace
}  // namespace aukit

using namespace aukit;

extern "C" int aukit_parse_container(const uint8_t *bytes, uint64_t n, int kind, int stream, aukit_container *out) {
    if (!out || (!bytes && n)) return fail(AUKIT_E_ARG, "null argument");
    memset(out, 0, sizeof *out);
    out->length_seconds = std::nan("");
    switch (kind) {
    case AUKIT_CONTAINER_WAV: return parse_wav(bytes, n, stream, out);
    case AUKIT_CONTAINER_AIFF: return parse_aiff(bytes, n, stream, out);
    case AUKIT_CONTAINER_AU: return parse_au(bytes, n, stream, out);
    }
    return fail(AUKIT_E_ARG, "unknown container kind %d", kind);
}
