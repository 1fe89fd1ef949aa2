"""Host-side byte parsing of the mirror (no GPU): aukit.detect (aukit.lua:2156-2195) and the container header readers
behind aukit.wav/aiff/au and stream.wav/aiff/au (aukit.lua:1456-1651, :2927-3113) — the latter through the C ABI (aukit_parse_container:
host code of libaukit_hip.so, no GPU needed).  Expected values are worked out by
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


def _p(data, kind, stream=False):
    """aukit_parse_container through the C ABI (no GPU needed: the walk is host code of libaukit_hip.so) → a plain dict"""
    from aukit_amd import _native as N
    c, payload = aukit._parse(data, kind, stream)
    d = c.desc
    return dict(codec=d.codec, channels=d.channels, sampleRate=d.sample_rate, bitDepth=c.bit_depth, blockAlign=d.block_align, dataType=N.WAVDT[c.wav_data_type],
                pcmType=d.data_type, bigEndian=d.big_endian, ulaw=d.ulaw, coefficients=[list(d.coef1[:d.ncoef]), list(d.coef2[:d.ncoef])], payload=payload,
                length=c.length_seconds)


WAV, AIFF, AU = 0, 1, 2


def test_parse_wav_pcm_and_skipped_chunks():
    fmt = struct.pack("<HHIIHH", 1, 2, 44100, 176400, 4, 16)
    f = _p(_wav(fmt, b"\x01\x02\x03\x04", extra=b"LIST" + struct.pack("<I", 4) + b"abcd"), WAV)
    assert (f["dataType"], f["channels"], f["sampleRate"], f["bitDepth"]) == ("signed", 2, 44100, 16)
    assert f["payload"] == b"\x01\x02\x03\x04"
    fmt = struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8)
    assert _p(_wav(fmt, b"\x80"), WAV)["dataType"] == "unsigned"
    # stream.wav computes the length itself for PCM (:2996): size / channels / (bitDepth / 8) / sampleRate
    assert _p(_wav(fmt, bytes(4000)), WAV, stream=True)["length"] == 0.5


@pytest.mark.parametrize("tag, kind", [(3, "float"), (6, "alaw"), (7, "ulaw"), (0x11, "adpcm")])
def test_parse_wav_format_tags(tag, kind):
    fmt = struct.pack("<HHIIHH", tag, 1, 22050, 22050, 256, 4)
    f = _p(_wav(fmt, bytes(8)), WAV)
    assert f["dataType"] == kind
    if kind == "adpcm":
        assert f["blockAlign"] == 256


def test_parse_wav_msadpcm_coefficients():
    co = [(256, 0), (512, -256), (0, 0), (192, 64)]
    fmt = struct.pack("<HHIIHH", 2, 1, 22050, 11100, 256, 4) + struct.pack("<HHH", 32, 500, len(co)) + b"".join(struct.pack("<hh", a, b) for a, b in co)
    f = _p(_wav(fmt, bytes(256)), WAV)
    assert f["dataType"] == "msadpcm" and f["blockAlign"] == 256
    assert f["coefficients"] == [[256, 512, 0, 192], [0, -256, 0, 64]]


def test_parse_wav_extensible_guids_and_errors():
    tail = bytes.fromhex("000000001000800000aa00389b71")
    for code, kind in ((1, "signed"), (3, "float"), (6, "alaw"), (7, "ulaw"), (0x11, "adpcm")):
        fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 48000, 0, 8, 32) + struct.pack("<HHI", 22, 24, 3) + struct.pack("<H", code) + tail
        f = _p(_wav(fmt, bytes(16)), WAV)
        assert f["dataType"] == kind and f["bitDepth"] == 24  # valid bits replace the container size (:1483)
    fmt = struct.pack("<HHIIHH", 0xFFFE, 1, 48000, 0, 8, 32) + struct.pack("<HHI", 22, 1, 3) + bytes.fromhex("3ac1fa38811d4361a40dce53ca607cd1")
    assert _p(_wav(fmt, bytes(16)), WAV)["dataType"] == "dfpwm"
    fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 48000, 0, 8, 32) + struct.pack("<HHI", 22, 24, 3) + bytes(16)
    with pytest.raises(aukit.LuaError, match="unsupported WAV file"):
        _p(_wav(fmt, bytes(16)), WAV)
    with pytest.raises(aukit.LuaError, match="unsupported WAV file"):
        _p(_wav(struct.pack("<HHIIHH", 0x55, 2, 48000, 0, 8, 32), bytes(16)), WAV)
    with pytest.raises(aukit.LuaError, match="not a WAV file"):
        _p(b"RIFX" + bytes(40), WAV)
    with pytest.raises(aukit.LuaError, match="invalid WAV file"):
        _p(b"RIFF" + struct.pack("<I", 4) + b"WAVE", WAV)
    with pytest.raises(aukit.LuaError, match="expected number, got nil"):  # data before fmt: aukit.pcm(data, nil, ...) fails its first check
        _p(b"RIFF" + struct.pack("<I", 16) + b"WAVE" + b"data" + struct.pack("<I", 4) + bytes(4), WAV)
    with pytest.raises(aukit.LuaError, match="invalid WAV file"):  # the data chunk announces more bytes than the file holds (:1507)
        _p(_wav(struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8), bytes(8))[:-3], WAV)


def test_wav_walk_goes_on_behind_the_data_chunk():
    """aukit.wav loops `while pos <= #data` and unpacks 8 bytes per chunk header (:1466-1468): 1-7 stray bytes behind the last chunk
    RAISE, a second data chunk replaces the first, a LIST/INFO chunk is walked entry by entry.  stream.wav returns at the first data
    chunk (:2980) and never sees any of it."""
    fmt = struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8)
    good = _wav(fmt, b"\x10\x20\x30\x40")
    for stray in (1, 3, 7):
        with pytest.raises(aukit.LuaError, match="data string too short"):
            _p(good + bytes(stray), WAV)
        assert _p(good + bytes(stray), WAV, stream=True)["payload"] == b"\x10\x20\x30\x40"
    second = good + b"data" + struct.pack("<I", 2) + b"\x77\x88"
    assert _p(second, WAV)["payload"] == b"\x77\x88" and _p(second, WAV, stream=True)["payload"] == b"\x10\x20\x30\x40"
    info = b"INFO" + b"INAM" + struct.pack("<I", 3) + b"abc" + b"\x00" + b"IART" + struct.pack("<I", 2) + b"xy"
    assert _p(good + b"LIST" + struct.pack("<I", len(info)) + info, WAV)["payload"] == b"\x10\x20\x30\x40"
    bad = b"INFO" + b"INAM" + struct.pack("<I", 300) + b"abc"  # the entry's string runs past the file
    with pytest.raises(aukit.LuaError, match="data string too short"):
        _p(good + b"LIST" + struct.pack("<I", len(bad)) + bad, WAV)
    with pytest.raises(aukit.LuaError, match="data string too short"):  # fmt chunk shorter than the 16 bytes "<HHIxxxxHH" reads
        _p(b"RIFF" + struct.pack("<I", 20) + b"WAVE" + b"fmt " + struct.pack("<I", 8) + bytes(8), WAV)


# ---------------------------------------------------------------- AIFF / AIFC (aukit.lua:1580-1633)
def _ext80(rate):
    m, e = np.frexp(float(rate))  # rate = m·2^e, 0.5 <= m < 1
    return struct.pack(">HQ", int(e) + 0x3FFE, int(m * 2.0 ** 64))


def _aiff(comm, ssnd, kind=b"AIFF", pre=b""):
    body = kind + pre + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    return b"FORM" + struct.pack(">I", len(body)) + body


@pytest.mark.parametrize("rate", [8000, 11025, 22050, 44100, 48000, 96000, 44100.5])
def test_aiff_rate_decoding(rate):
    comm = struct.pack(">hIh", 1, 2, 16) + _ext80(rate)
    assert _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(4)), AIFF)["sampleRate"] == pytest.approx(rate, rel=2 ** -50)


def test_parse_aiff_and_aifc():
    from aukit_amd import _native as N
    comm = struct.pack(">hIh", 2, 3, 16) + _ext80(44100)
    f = _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(range(12))), AIFF)
    assert (f["codec"], f["channels"], f["bitDepth"], f["sampleRate"], f["bigEndian"]) == (N.CODEC_PCM, 2, 16, 44100, 1)
    assert f["payload"] == bytes(range(12))  # frames · channels · bytes
    # AIFC: compression id + pascal string (padded to even), SSND offset skips leading bytes
    name = b"\x0enot compressed\x00"
    comm = struct.pack(">hIh", 1, 4, 8) + _ext80(8000) + b"NONE" + name
    ssnd = struct.pack(">II", 2, 0) + b"\xee\xee" + bytes([1, 2, 3, 4])
    f = _p(_aiff(comm, ssnd, b"AIFC", b"FVER" + struct.pack(">I", 4) + bytes(4)), AIFF)
    assert (f["channels"], f["bitDepth"], f["sampleRate"], f["bigEndian"]) == (1, 8, 8000, 1)
    assert f["payload"] == bytes([1, 2, 3, 4])
    # sowt: little-endian for aukit.aiff (:1613), big-endian for stream.aiff (:3065) — the reference's own inconsistency
    comm = struct.pack(">hIh", 1, 2, 16) + _ext80(8000) + b"sowt" + b"\x00\x00"
    ssnd = struct.pack(">II", 0, 0) + bytes(4)
    assert _p(_aiff(comm, ssnd, b"AIFC"), AIFF)["bigEndian"] == 0 and _p(_aiff(comm, ssnd, b"AIFC"), AIFF, stream=True)["bigEndian"] == 1
    comm = struct.pack(">hIh", 1, 4, 8) + _ext80(8000) + b"ulaw" + b"\x00\x00"
    f = _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(4), b"AIFC"), AIFF, stream=True)
    assert (f["codec"], f["ulaw"], f["length"]) == (N.CODEC_G711, 1, 4 / 8000)
    comm = struct.pack(">hIh", 1, 4, 8) + _ext80(8000) + b"MAC3" + b"\x00\x00"
    with pytest.raises(aukit.LuaError, match="Unsupported compression scheme MAC3"):
        _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(4), b"AIFC"), AIFF)
    # a short SSND chunk: aukit.aiff takes what is there (:1610 is commented out), stream.aiff raises (:3046)
    comm = struct.pack(">hIh", 1, 8, 8) + _ext80(8000)
    assert _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(5)), AIFF)["payload"] == bytes(5)
    with pytest.raises(aukit.LuaError, match="invalid AIFF file"):
        _p(_aiff(comm, struct.pack(">II", 0, 0) + bytes(5)), AIFF, stream=True)
    with pytest.raises(aukit.LuaError, match="not an AIFF file"):
        _p(b"FORM" + bytes(4) + b"WAVE", AIFF)
    with pytest.raises(aukit.LuaError, match="invalid AIFF file"):
        _p(b"FORM" + struct.pack(">I", 4) + b"AIFF", AIFF)
    with pytest.raises(aukit.LuaError, match="data string too short"):
        _p(b"FORM" + struct.pack(">I", 4) + b"AIFF" + b"COM", AIFF)


# ---------------------------------------------------------------- AU (aukit.lua:1639-1651)
def test_parse_au():
    from aukit_amd import _native as N
    hdr = struct.pack(">4sIIIII", b".snd", 24, 6, 3, 22050, 2)
    f = _p(hdr + bytes([9, 8, 7, 6, 5, 4, 3]), AU)
    assert (f["codec"], f["bitDepth"], f["sampleRate"], f["channels"], f["bigEndian"]) == (N.CODEC_PCM, 16, 22050, 2, 1)
    # str_sub(data, offset, offset + size - 1) is 1-based: the window starts one byte before the AU data offset
    assert f["payload"] == hdr[-1:] + bytes([9, 8, 7, 6, 5])
    f = _p(struct.pack(">4sIIIII", b".snd", 24, 0xFFFFFFFF, 1, 8000, 1) + b"abc", AU, stream=True)
    assert f["payload"] == b"\x01abc" and f["codec"] == N.CODEC_G711 and f["ulaw"] == 1 and f["length"] == 0xFFFFFFFF / 8000  # :3107
    assert _p(struct.pack(">4sIIIII", b".snd", 24, 2, 27, 8000, 1) + b"abc", AU)["ulaw"] == 0
    with pytest.raises(aukit.LuaError, match="invalid AU file"):
        _p(struct.pack(">4sIIIII", b".sne", 24, 0, 1, 8000, 1), AU)
    with pytest.raises(aukit.LuaError, match="unsupported encoding type 23"):
        _p(struct.pack(">4sIIIII", b".snd", 24, 0, 23, 8000, 1), AU)
    with pytest.raises(aukit.LuaError, match="data string too short"):
        _p(b".snd" + bytes(10), AU)
