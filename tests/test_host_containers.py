"""Host-side byte parsing of the mirror (no GPU): aukit.detect (aukit.lua:2156-2195) and the container header readers
behind aukit.wav/aiff/au and stream.wav/aiff/au (aukit.lua:1456-1651, :2927-3113).  Expected values are worked out by
hand from the Lua, not from the oracle."""
import struct

import numpy as np
import pytest

import aukit_amd.aukit as aukit


# ---------------------------------------------------------------- aukit.detect
@pytest.mark.parametrize("data, kind", [
    (b"RIFF\x24\x00\x00\x00WAVEfmt ", "wav"),
    (b"RIFF\n\n\n\nWAVE", "wav"),  # Lua '.' matches newlines too
    (b"FORM\x00\x00\x00\x10AIFFCOMM", "aiff"),
    (b"FORM\x00\x00\x00\x10AIFCFVER", "aiff"),
    (b".snd\x00\x00\x00\x18", "au"),
    (b"fLaC\x00\x00\x00\x22", "flac"),
    (b"MDFPWM\x03" + bytes(40), "mdfpwm"),
    (b"qoaf\x00\x00\x10\x00", "qoa"),
])
def test_detect_container_magic(data, kind):
    assert aukit.detect(data) == (kind, None, None)


def test_detect_magic_must_be_anchored():
    # "^RIFF....WAVE": not at the start → not a wav; 12 idle bytes make it dfpwm instead
    assert aukit.detect(b"x" + b"RIFF\x24\x00\x00\x00WAVE" + b"\xaa" * 12) == ("dfpwm", None, None)
    loud = b"\x5b" * 100  # out of range for every row of the table
    assert aukit.detect(b"FORM\x00\x00\x00\x10AIFX" + loud) == (None, None, None)
    assert aukit.detect(b"MDFPWM\x04" + loud) == (None, None, None)
    # 0xFF bytes at the end are -1 as signed 8-bit: "near silence", so the tail heuristic fires (:2183)
    assert aukit.detect(b"MDFPWM\x04" + b"\xff" * 40) == ("pcm", 8, "signed")


def test_detect_pcm_heuristic_order_and_gaps():
    loud = b"\x5b" * 100  # out of range for every row of the table
    # 8-bit signed: |v| <= 8, not all zero, at the start
    assert aukit.detect(bytes([0, 1, 0xFF, 8, 0xF8, 0, 0, 0]) + loud) == ("pcm", 8, "signed")
    # 9 is outside the 8-bit signed gap; around 128 it is 8-bit unsigned
    assert aukit.detect(bytes([128, 127, 129, 136, 120, 128, 128, 128]) + loud) == ("pcm", 8, "unsigned")
    # 16-bit signed: |v| <= 2048 but the bytes themselves fail both 8-bit tests
    s16 = struct.pack("<8h", 300, -300, 2048, -2048, 17, 0, 0, 0)
    assert aukit.detect(s16 + loud) == ("pcm", 16, "signed")
    # 32-bit signed comes before float in the table: small ints with a big low half
    s32 = struct.pack("<8i", 70000, -70000, 8 << 24, -(8 << 24), 40000, 0, 0, 0)
    assert aukit.detect(s32 + loud) == ("pcm", 32, "signed")
    # floats within 0.001 whose bit patterns are huge as int32
    f32 = struct.pack("<8f", 0.0005, -0.0009, 0.0009, 1e-4, 0, 0, 0, 0)
    assert aukit.detect(f32 + loud) == ("pcm", 32, "float")
    # float32(0.001) is a hair above the double 0.001 the reference compares with: already out of range
    f32 = struct.pack("<8f", 0.0005, -0.0009, 0.001, 1e-4, 0, 0, 0, 0)
    assert aukit.detect(f32 + loud) == (None, None, None)


def test_detect_pcm_24bit_and_unsigned_wide():
    loud = b"\x5b" * 100  # out of range for every row of the table

    def i3(v, signed):
        return int(v).to_bytes(3, "little", signed=signed)
    # 24-bit signed within ±8·2^16 that no earlier format accepts
    vals = [400000, -400000, 524288, -524288, 300000, -300000, 70000, -70000]
    data = b"".join(i3(v, True) for v in vals) + loud
    assert aukit.detect(data) == ("pcm", 24, "signed")
    # 32-bit unsigned around 2^31
    u32 = struct.pack("<8I", 2 ** 31, 2 ** 31 + (8 << 24), 2 ** 31 - (8 << 24), 2 ** 31 + 5, 2 ** 31, 2 ** 31, 2 ** 31, 2 ** 31)
    assert aukit.detect(u32 + loud) == ("pcm", 32, "unsigned")
    # 16-bit unsigned around 32768 is the last entry
    u16 = struct.pack("<8H", 32768, 32768 + 2048, 32768 - 2048, 32700, 32768, 32768, 32768, 32768)
    assert aukit.detect(u16 + loud) == ("pcm", 16, "unsigned")


def test_detect_all_zero_is_not_pcm_and_tail_window():
    assert aukit.detect(bytes(64)) == (None, None, None)
    # the tail read starts at `#data - bitDepth` (1-based), i.e. it ends one byte before the end of the string
    loud = b"\x5b" * 100  # out of range for every row of the table
    tail = bytes([0, 1, 0, 0, 0, 0, 0, 2])
    assert aukit.detect(loud + tail + b"\x7f") == ("pcm", 8, "signed")
    assert aukit.detect(loud + tail) != ("pcm", 8, "signed")  # shifted by one: the window starts with a byte of `loud`
    # exactly bitDepth bytes long: `#data - bitDepth` = 0 is out of the string, the head read decides alone
    assert aukit.detect(bytes([200] * 8)) == (None, None, None)
    # shorter than any format: every unpack raises, nothing matches
    assert aukit.detect(b"\x01\x02\x03") == (None, None, None)
    assert aukit.detect(b"") == (None, None, None)


def test_detect_dfpwm_idle_pattern_anywhere():
    noise = bytes((i * 73 + 41) % 251 + 3 for i in range(200))
    assert aukit.detect(noise) == (None, None, None)
    assert aukit.detect(noise + b"\x55" * 12 + noise) == ("dfpwm", None, None)
    assert aukit.detect(noise + b"\xaa" * 12) == ("dfpwm", None, None)
    assert aukit.detect(noise + b"\xaa" * 11 + b"\x55" * 11) == (None, None, None)


def test_detect_nan_floats_pass_the_range_test():
    # NaN compares false both ways (:2177), so a block of NaNs is "ok and not allzero" once the float row is reached;
    # 0x7FC00000 words fail the integer rows before it
    nan = struct.pack("<I", 0x7FC00000) * 8
    assert aukit.detect(nan + b"\x5b" * 100) == ("pcm", 32, "float")


def test_detect_argument_check():
    with pytest.raises(aukit.LuaError):
        aukit.detect(12)


# ---------------------------------------------------------------- WAV header (aukit.lua:1459-1507)
def _wav(fmt_chunk, payload, extra=b""):
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + extra + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_parse_wav_pcm_and_skipped_chunks():
    fmt = struct.pack("<HHIIHH", 1, 2, 44100, 176400, 4, 16)
    f = aukit._parse_wav(_wav(fmt, b"\x01\x02\x03\x04", extra=b"LIST" + struct.pack("<I", 4) + b"abcd"))
    assert (f["dataType"], f["channels"], f["sampleRate"], f["bitDepth"], f["blockAlign"]) == ("signed", 2, 44100, 16, 4)
    assert f["payload"] == b"\x01\x02\x03\x04"
    fmt = struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8)
    assert aukit._parse_wav(_wav(fmt, b"\x80"))["dataType"] == "unsigned"


@pytest.mark.parametrize("tag, kind", [(3, "float"), (6, "alaw"), (7, "ulaw"), (0x11, "adpcm")])
def test_parse_wav_format_tags(tag, kind):
    fmt = struct.pack("<HHIIHH", tag, 1, 22050, 22050, 256, 4)
    assert aukit._parse_wav(_wav(fmt, bytes(8)))["dataType"] == kind


def test_parse_wav_msadpcm_coefficients():
    co = [(256, 0), (512, -256), (0, 0), (192, 64)]
    fmt = struct.pack("<HHIIHH", 2, 1, 22050, 11100, 256, 4) + struct.pack("<HHH", 32, 500, len(co)) + b"".join(struct.pack("<hh", a, b) for a, b in co)
    f = aukit._parse_wav(_wav(fmt, bytes(256)))
    assert f["dataType"] == "msadpcm" and f["blockAlign"] == 256
    assert f["coefficients"] == [[256, 512, 0, 192], [0, -256, 0, 64]]


def test_parse_wav_extensible_guids_and_errors():
    tail = bytes.fromhex("000000001000800000aa00389b71")
    for code, kind in ((1, "signed"), (3, "float"), (6, "alaw"), (7, "ulaw"), (0x11, "adpcm")):
        fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 48000, 0, 8, 32) + struct.pack("<HHI", 22, 24, 3) + struct.pack("<H", code) + tail
        f = aukit._parse_wav(_wav(fmt, bytes(16)))
        assert f["dataType"] == kind and f["bitDepth"] == 24  # valid bits replace the container size (:1483)
    fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 48000, 0, 8, 32) + struct.pack("<HHI", 22, 24, 3) + bytes(16)
    with pytest.raises(aukit.LuaError, match="unsupported WAV file"):
        aukit._parse_wav(_wav(fmt, bytes(16)))
    with pytest.raises(aukit.LuaError, match="unsupported WAV file"):
        aukit._parse_wav(_wav(struct.pack("<HHIIHH", 0x55, 2, 48000, 0, 8, 32), bytes(16)))
    with pytest.raises(aukit.LuaError, match="not a WAV file"):
        aukit._parse_wav(b"RIFX" + bytes(40))
    with pytest.raises(aukit.LuaError, match="invalid WAV file"):
        aukit._parse_wav(b"RIFF" + struct.pack("<I", 4) + b"WAVE")
    with pytest.raises(aukit.LuaError, match="invalid WAV file"):  # data before fmt
        aukit._parse_wav(b"RIFF" + struct.pack("<I", 16) + b"WAVE" + b"data" + struct.pack("<I", 4) + bytes(4))


# ---------------------------------------------------------------- AIFF / AIFC (aukit.lua:1580-1633)
def _ext80(rate):
    m, e = np.frexp(float(rate))  # rate = m·2^e, 0.5 <= m < 1
    return struct.pack(">HQ", int(e) + 0x3FFE, int(m * 2.0 ** 64))


@pytest.mark.parametrize("rate", [8000, 11025, 22050, 44100, 48000, 96000, 44100.5])
def test_aiff_rate_decoding(rate):
    e, m = struct.unpack(">HQ", _ext80(rate))
    assert aukit._aiff_rate(e, m >> 8) == pytest.approx(rate, rel=2 ** -50)


def test_parse_aiff_and_aifc():
    comm = struct.pack(">hIh", 2, 3, 16) + _ext80(44100)
    ssnd = struct.pack(">II", 0, 0) + bytes(range(12))
    body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    f = aukit._parse_aiff(b"FORM" + struct.pack(">I", len(body)) + body)
    assert (f["channels"], f["bitDepth"], f["sampleRate"], f["compression"]) == (2, 16, 44100, None)
    assert f["payload"] == bytes(range(12))  # frames · channels · bytes
    # AIFC: compression id + pascal string (padded to even), SSND offset skips leading bytes
    name = b"\x0enot compressed\x00"
    comm = struct.pack(">hIh", 1, 4, 8) + _ext80(8000) + b"NONE" + name
    ssnd = struct.pack(">II", 2, 0) + b"\xee\xee" + bytes([1, 2, 3, 4])
    body = b"AIFC" + b"FVER" + struct.pack(">I", 4) + bytes(4) + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    f = aukit._parse_aiff(b"FORM" + struct.pack(">I", len(body)) + body)
    assert (f["channels"], f["bitDepth"], f["sampleRate"], f["compression"]) == (1, 8, 8000, b"NONE")
    assert f["payload"] == bytes([1, 2, 3, 4])
    with pytest.raises(aukit.LuaError, match="not an AIFF file"):
        aukit._parse_aiff(b"FORM" + bytes(4) + b"WAVE")
    with pytest.raises(aukit.LuaError, match="invalid AIFF file"):
        aukit._parse_aiff(b"FORM" + struct.pack(">I", 4) + b"AIFF")


# ---------------------------------------------------------------- AU (aukit.lua:1639-1651)
def test_parse_au():
    hdr = struct.pack(">4sIIIII", b".snd", 24, 6, 3, 22050, 2)
    f = aukit._parse_au(hdr + bytes([9, 8, 7, 6, 5, 4, 3]))
    assert (f["encoding"], f["sampleRate"], f["channels"]) == (3, 22050, 2)
    # str_sub(data, offset, offset + size - 1) is 1-based: the window starts one byte before the AU data offset
    assert f["payload"] == hdr[-1:] + bytes([9, 8, 7, 6, 5])
    f = aukit._parse_au(struct.pack(">4sIIIII", b".snd", 24, 0xFFFFFFFF, 1, 8000, 1) + b"abc")
    assert f["payload"] == b"\x01abc"
    with pytest.raises(aukit.LuaError, match="invalid AU file"):
        aukit._parse_au(struct.pack(">4sIIIII", b".sne", 24, 0, 1, 8000, 1))
