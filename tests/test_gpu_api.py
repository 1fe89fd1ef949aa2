"""GPU: the host mirror of the reference's Lua API (aukit_amd.aukit) — call sequences of auplay.lua / austream.lua."""
import struct

import numpy as np
import pytest

from tests.util import pcm16

pytestmark = pytest.mark.gpu


def _wav(fmt, ch, rate, ba, bits, payload, extra=b""):
    fmtc = struct.pack("<HHIIHH", fmt, ch, rate, rate * ba, ba, bits) + extra
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmtc) + 8 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmtc)) + fmtc + \
        b"LIST" + struct.pack("<I", 4) + b"INFO" + b"data" + struct.pack("<I", len(payload)) + payload


def test_auplay_sequence_on_a_wav(ctx, oracle):
    """auplay.lua:11-31: aukit.wav → resample(48000) → mono() → effects.normalize(0.8) → effects.lowpass(sr/2) → stream(48000)."""
    import aukit_amd.aukit as aukit
    st = np.stack([pcm16(12000, 44100, 5, 0), pcm16(12000, 44100, 5, 1)], 1)
    w = _wav(1, 2, 44100, 4, 16, st.tobytes())
    audio = aukit.wav(w)
    assert audio.channels() == 2 and audio.sampleRate == 44100 and abs(audio.len() - 12000 / 44100) < 1e-12
    mono = audio.resample(48000).mono()
    assert aukit.effects.normalize(mono, 0.8) is mono  # effects mutate and return the same object
    aukit.effects.lowpass(mono, audio.sampleRate / 2)
    ref = oracle.fx_lowpass(oracle.fx_normalize(oracle.mono(oracle.resample(oracle.pcm(st.tobytes(), 16, oracle.SIGNED, 2, 44100), 48000, oracle.LINEAR)), 0.8), 22050.0)
    assert np.max(np.abs(mono.data[0] - ref.data[0])) <= 1e-12
    it, length = mono.stream(48000)
    chunks = list(it)
    assert abs(length - len(ref.data[0]) / 48000) < 1e-12 and len(chunks) == 1
    exp = np.where(ref.data[0] < 0, ref.data[0] * 128, ref.data[0] * 127)
    assert np.max(np.abs(chunks[0][0][0] - exp)) <= 1e-10
    assert audio.channels() == 2  # the source object is never mutated by Audio methods


def test_austream_dispatch(ctx, oracle):
    """austream.lua:85-92: aukit.stream.<type>(data, mono) picked by container; aukit.defaultInterpolation is honoured."""
    import aukit_amd.aukit as aukit
    x = pcm16(44100 + 5000, 44100, 1, 0)
    w = _wav(1, 1, 44100, 2, 16, x.tobytes())
    for interp in ("linear", "cubic"):
        aukit.defaultInterpolation = interp
        it, length = aukit.stream.wav(w, True)
        got = list(it)
        ref = oracle.stream_pcm(x.tobytes(), 16, oracle.SIGNED, 1, 44100, False, False, oracle.INTERP[interp])
        assert len(got) == ref.nchunks and abs(length - len(x) / 44100) < 1e-12
        off = 0
        for (chunk, pos), n, p in zip(got, ref.chunk_len[:, 0], ref.chunk_pos):
            assert pos == p and np.max(np.abs(chunk[0] - ref.data[0][off:off + n])) <= 1e-13
            off += n
    aukit.defaultInterpolation = "cubic"
    ima = oracle.gen_ima(pcm16(1016 * 30, 22050, 3, 0), 1, 512, 88)
    it, _ = aukit.stream.wav(_wav(0x11, 1, 22050, 512, 4, ima, struct.pack("<HH", 2, 1017)))
    got = np.concatenate([c[0] for c, _ in it])
    assert np.array_equal(got, oracle.stream_adpcm(ima, 512, 1, 22050, False, oracle.CUBIC).data[0])
    g = oracle.gen_g711(pcm16(8000, 8000, 2, 0), True)
    au = b".snd" + struct.pack(">IIIII", 25, len(g), 1, 8000, 1) + g  # the reference uses the offset field as a 1-based str_sub index
    it, _ = aukit.stream.au(au)
    first, _ = next(it)
    assert np.array_equal(first[0], oracle.stream_g711(g, True, 1, 8000, False, oracle.CUBIC).data[0])
    aukit.defaultInterpolation = "linear"


def test_wav_variants_and_errors(ctx, oracle):
    import aukit_amd.aukit as aukit
    ima = oracle.gen_ima(pcm16(1016 * 4, 22050, 3, 1), 1, 512, 15)
    a = aukit.wav(_wav(0x11, 1, 22050, 512, 4, ima, struct.pack("<HH", 2, 1017)))
    assert np.array_equal(a.data[0], oracle.wav_adpcm(ima, 512, 1, 22050).data[0])
    u8 = bytes(range(256))
    b = aukit.wav(_wav(1, 1, 8000, 1, 8, u8))
    assert np.array_equal(b.data[0], oracle.pcm(u8, 8, oracle.UNSIGNED, 1, 8000).data[0])
    ext = struct.pack("<HHI", 22, 16, 3) + bytes.fromhex("0100000000001000800000aa00389b71")
    c = aukit.wav(_wav(0xFFFE, 1, 8000, 2, 16, pcm16(100, 8000, 1, 0).tobytes(), ext))
    assert len(c.data[0]) == 100
    for bad, msg in ((b"RIFX" + b"\0" * 40, "not a WAV file"), (_wav(0x55, 1, 8000, 1, 8, b"\0" * 8), "unsupported WAV file")):
        with pytest.raises(aukit.LuaError) as e:
            aukit.wav(bad)
        assert msg in str(e.value)
    with pytest.raises(aukit.LuaError) as e:
        aukit.pcm(b"\0" * 4, 12)
    assert "invalid bit depth" in str(e.value)
    with pytest.raises(aukit.LuaError) as e:
        aukit.pcm(b"\0" * 4, 16).resample(48000, "spline")
    assert "invalid interpolation type" in str(e.value)
    with pytest.raises(aukit.LuaError) as e:
        aukit.effects.amplify(aukit.pcm(b"\0" * 4, 16), "loud")
    assert "bad argument #2 (expected number, got string)" in str(e.value)


def test_mix_dfpwm_and_pcm_methods(ctx, oracle):
    import aukit_amd.aukit as aukit
    aukit.defaultInterpolation = "linear"
    a = aukit.pcm(pcm16(3000, 48000, 1, 0).tobytes(), 16, "signed", 1, 48000)
    b = aukit.pcm(pcm16(2000, 24000, 1, 1).tobytes(), 16, "signed", 1, 24000)
    m = a.mix(0.5, b)  # b is resampled to 48 kHz with the default (linear) interpolation first
    oa = oracle.pcm(pcm16(3000, 48000, 1, 0).tobytes(), 16, oracle.SIGNED, 1, 48000)
    ob = oracle.resample(oracle.pcm(pcm16(2000, 24000, 1, 1).tobytes(), 16, oracle.SIGNED, 1, 24000), 48000, oracle.LINEAR)
    assert np.array_equal(m.data[0], oracle.mix([oa, ob], 0.5).data[0])
    assert a.dfpwm() == oracle.audio_dfpwm(oa, True)
    assert np.array_equal(a.pcm(16, "signed"), oracle.encode_pcm(oa, 16, oracle.SIGNED, True))
    d = oracle.dfpwm_encode(np.round(np.sin(np.arange(48000) / 20) * 90))
    assert np.array_equal(aukit.dfpwm(d, 1, 48000).data[0], oracle.dfpwm(d, 1, 48000).data[0])


def _ext80(rate):
    """IEEE 754 80-bit extended big-endian, as AIFF stores the sample rate"""
    import math
    m, e = math.frexp(rate)  # rate = m * 2^e, 0.5 <= m < 1
    return struct.pack(">HQ", 16382 + e, int(m * (1 << 64)))


def _aiff(ch, frames, bits, rate, payload, comp=None, ssnd_offset=0, extra_chunks=b""):
    comm = struct.pack(">hIh", ch, frames, bits) + _ext80(rate)
    if comp is not None:
        name = b"not compressed"  # pascal string, even length → one pad byte
        comm += comp + bytes([len(name)]) + name + b"\0"
    body = (b"AIFC" if comp is not None else b"AIFF") + extra_chunks + b"COMM" + struct.pack(">I", len(comm)) + comm + \
        b"SSND" + struct.pack(">III", 8 + ssnd_offset + len(payload), ssnd_offset, 0) + b"\x55" * ssnd_offset + payload
    return b"FORM" + struct.pack(">I", len(body)) + body


def test_aiff_aifc_and_au_containers(ctx, oracle):
    """aukit.aiff (:1580-1633: COMM with the 80-bit rate, AIFC compression ids, SSND offset) and aukit.au (:1639-1651: 1-based data
    offset, encodings 1-6 and 27) hand the right bytes and descriptors to the codecs"""
    import aukit_amd.aukit as aukit
    rng = np.random.Generator(np.random.PCG64(5))
    n, ch = 500, 2
    for rate in (22050, 44100, 8000, 48000, 11025):
        x = rng.integers(-32768, 32768, n * ch, dtype=np.int64).astype(">i2").tobytes()
        a = aukit.aiff(_aiff(ch, n, 16, rate, x, ssnd_offset=rate % 7, extra_chunks=b"NAME" + struct.pack(">I", 4) + b"abcd"))
        assert a.sampleRate == rate and a.channels() == ch
        ref = oracle.pcm(x, 16, oracle.SIGNED, ch, rate, True, True)
        for c in range(ch):
            assert np.array_equal(a.data[c], ref.data[c])
    x24 = bytes(rng.integers(0, 256, n * 3, dtype=np.uint8))
    assert np.array_equal(aukit.aiff(_aiff(1, n, 24, 32000, x24)).data[0], oracle.pcm(x24, 24, oracle.SIGNED, 1, 32000, True, True).data[0])
    le = rng.integers(-32768, 32768, n, dtype=np.int64).astype("<i2").tobytes()
    assert np.array_equal(aukit.aiff(_aiff(1, n, 16, 44100, le, b"sowt")).data[0], oracle.pcm(le, 16, oracle.SIGNED, 1, 44100, True, False).data[0])
    assert np.array_equal(aukit.aiff(_aiff(1, n, 16, 44100, x[:2 * n], b"NONE")).data[0], oracle.pcm(x[:2 * n], 16, oracle.SIGNED, 1, 44100, True, True).data[0])
    fl = rng.uniform(-1, 1, n).astype(">f4").tobytes()
    assert np.array_equal(aukit.aiff(_aiff(1, n, 32, 48000, fl, b"fl32")).data[0], oracle.pcm(fl, 32, oracle.FLOAT, 1, 48000, True, True).data[0])
    g = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    assert np.array_equal(aukit.aiff(_aiff(1, n, 8, 8000, g, b"ulaw")).data[0], oracle.g711(g, True, 1, 8000).data[0])
    assert np.array_equal(aukit.aiff(_aiff(1, n, 8, 8000, g, b"ALAW")).data[0], oracle.g711(g, False, 1, 8000).data[0])
    with pytest.raises(aukit.LuaError, match="Unsupported compression scheme"):
        aukit.aiff(_aiff(1, n, 8, 8000, g, b"ima4"))
    with pytest.raises(aukit.LuaError, match="not an AIFF file"):
        aukit.aiff(b"RIFF" + b"\0" * 40)
    # AU: header 24 bytes (+ annotation); the offset field is used as a 1-based index (a file written with offset = 24 loses no byte
    # only if the writer meant 25), size 0xFFFFFFFF = to the end
    for enc, bits, dt in ((2, 8, oracle.SIGNED), (3, 16, oracle.SIGNED), (4, 24, oracle.SIGNED), (5, 32, oracle.SIGNED), (6, 32, oracle.FLOAT)):
        raw = rng.uniform(-1, 1, n * 2).astype(">f4").tobytes() if enc == 6 else bytes(rng.integers(0, 256, n * 2 * (bits // 8), dtype=np.uint8))
        for size in (len(raw), 0xFFFFFFFF):
            au = b".snd" + struct.pack(">IIIII", 29, size, enc, 16000, 2) + b"anno" + raw
            a = aukit.au(au)
            ref = oracle.pcm(raw, bits, dt, 2, 16000, True, True)
            for c in range(2):
                assert np.array_equal(a.data[c], ref.data[c]), (enc, size)
    for enc, ul in ((1, True), (27, False)):
        a = aukit.au(b".snd" + struct.pack(">IIIII", 25, len(g), enc, 8000, 1) + g)
        assert np.array_equal(a.data[0], oracle.g711(g, ul, 1, 8000).data[0])
    with pytest.raises(aukit.LuaError, match="unsupported encoding type 23"):
        aukit.au(b".snd" + struct.pack(">IIIII", 25, 4, 23, 8000, 1) + b"\0" * 4)
    # stream.aiff / stream.au dispatch to the same stream kernels
    aukit.defaultInterpolation = "linear"
    it, length = aukit.stream.aiff(_aiff(1, n, 16, 22050, x[:2 * n]))
    got = np.concatenate([c[0] for c, _ in it])
    ref = oracle.stream_pcm(x[:2 * n], 16, oracle.SIGNED, 1, 22050, True, False, oracle.LINEAR)
    assert np.max(np.abs(got - ref.data[0])) <= 1e-13


def test_wav_formats_through_loader_and_stream(ctx, oracle):
    """every `fmt ` the reference understands (:1473-1503, :2941-2975) through aukit.wav and aukit.stream.wav: MS-ADPCM with its
    coefficient table, IEEE float, A-law / µ-law, unsigned 8-bit, and WAVE_FORMAT_EXTENSIBLE with the PCM and the DFPWM GUID"""
    import aukit_amd.aukit as aukit
    rng = np.random.Generator(np.random.PCG64(11))
    aukit.defaultInterpolation = "linear"

    def stream_all(w, mono=None):
        import itertools
        it, _ = aukit.stream.wav(w, mono)
        chunks = []
        for c in itertools.islice(it, 50):  # stream.g711 on a string never returns nil (Q13): stop at the first empty chunk
            if len(c[0][0]) == 0:
                break
            chunks.append(c)
        return [np.concatenate([c[0][k] for c in chunks]) for k in range(len(chunks[0][0]))]

    # MS-ADPCM, stereo, the standard 7 coefficient pairs in the header
    ba, ch = 256, 2
    spb = (ba - 14) + 2
    ms = oracle.gen_msadpcm(np.stack([pcm16(spb * 20, 22050, 6, c) for c in range(ch)], 1).ravel(), ch, ba)
    co = [(256, 0), (512, -256), (0, 0), (192, 64), (240, 0), (460, -208), (392, -232)]
    extra = struct.pack("<HHH", 32, spb, 7) + b"".join(struct.pack("<hh", a, b) for a, b in co)
    w = _wav(2, ch, 22050, ba, 4, ms, extra)
    a = aukit.wav(w)
    ref = oracle.msadpcm(ms, ba, ch, 22050, [[c[0] for c in co], [c[1] for c in co]])
    for c in range(ch):
        assert np.array_equal(a.data[c], ref.data[c])
    rs = oracle.stream_msadpcm(ms, ba, ch, 22050, False, [[c[0] for c in co], [c[1] for c in co]], oracle.LINEAR)
    got = stream_all(w)
    for c in range(ch):
        assert np.array_equal(got[c], rs.data[c])
    # IEEE float 32, mono
    fl = rng.uniform(-1, 1, 3000).astype("<f4").tobytes()
    w = _wav(3, 1, 32000, 4, 32, fl)
    assert np.array_equal(aukit.wav(w).data[0], oracle.pcm(fl, 32, oracle.FLOAT, 1, 32000).data[0])
    assert np.max(np.abs(stream_all(w)[0] - oracle.stream_pcm(fl, 32, oracle.FLOAT, 1, 32000, False, False, oracle.LINEAR).data[0])) <= 1e-13
    # A-law (6) and µ-law (7), stereo, with the mono mix in the stream
    g = bytes(rng.integers(0, 256, 8000 * 2 + 10, dtype=np.uint8))
    for fmt, ul in ((6, False), (7, True)):
        w = _wav(fmt, 2, 8000, 2, 8, g)
        a = aukit.wav(w)
        r = oracle.g711(g, ul, 2, 8000)
        assert np.array_equal(a.data[0], r.data[0]) and np.array_equal(a.data[1], r.data[1])
        sg = stream_all(w, True)[0]
        rg = oracle.stream_g711(g, ul, 2, 8000, True, oracle.LINEAR).data[0]
        assert len(sg) == len(rg) and np.array_equal(sg, rg)
    # extensible: PCM 16 and DFPWM
    x = pcm16(2000, 44100, 1, 3).tobytes()
    ext_pcm = struct.pack("<HHI", 22, 16, 4) + bytes.fromhex("0100000000001000800000aa00389b71")
    assert np.array_equal(aukit.wav(_wav(0xFFFE, 1, 44100, 2, 16, x, ext_pcm)).data[0], oracle.pcm(x, 16, oracle.SIGNED, 1, 44100).data[0])
    d = oracle.dfpwm_encode(np.round(np.sin(np.arange(48000) / 17) * 100))
    ext_df = struct.pack("<HHI", 22, 1, 4) + bytes.fromhex("3ac1fa38811d4361a40dce53ca607cd1")
    w = _wav(0xFFFE, 1, 48000, 1, 1, d, ext_df)
    assert np.array_equal(aukit.wav(w).data[0], oracle.dfpwm(d, 1, 48000).data[0])
    assert np.max(np.abs(stream_all(w)[0] - oracle.stream_dfpwm(d, 48000, 1, False, oracle.LINEAR).data[0])) <= 1e-13
    with pytest.raises(aukit.LuaError, match="unsupported WAV file"):
        aukit.wav(_wav(0xFFFE, 1, 48000, 1, 1, d, struct.pack("<HHI", 22, 1, 4) + b"\x55" * 16))


def test_table_inputs_of_stream_pcm_and_adpcm(ctx, oracle):
    """VERDICT r02 'missing' 3 / 4: aukit.stream.pcm on a TABLE of numbers (aukit.lua:2255-2290: `read()` hands out data[pos], normalised like the
    string's samples, `len = #data / channels`) and aukit.adpcm on a TABLE of nibbles (:1232-1238).  Both mirrors route them through
    aukit_stream_decode_table / aukit_decode_nibbles; here against the oracle's string versions of the same numbers."""
    import aukit_amd.aukit as aukit
    rng = np.random.Generator(np.random.PCG64(77))
    aukit.defaultInterpolation = "cubic"
    try:
        # signed 16-bit integers, mono and stereo (+ mono mix): the table IS the string's sample sequence
        for ch, mono in ((1, False), (2, False), (2, True)):
            x = rng.integers(-32768, 32768, 50001 * ch if ch == 1 else 50000 * ch).astype(np.int16)
            it, length = aukit.stream.pcm([int(v) for v in x], 16, "signed", ch, 44100, False, mono)
            ref = oracle.stream_pcm(x.astype("<i2").tobytes(), 16, oracle.SIGNED, ch, 44100, False, mono, oracle.CUBIC)
            got = list(it)
            assert len(got) == ref.nchunks and length == len(x) / ch / 44100
            off = 0
            for k, (chunk, pos) in enumerate(got):
                assert pos == ref.chunk_pos[k]
                for c in range(ref.channels):
                    assert np.max(np.abs(chunk[c] - ref.data[c][off:off + len(chunk[c])]), initial=0) <= 1e-13, (ch, mono, k, c)
                off += len(chunk[0])
        # unsigned 8-bit (Q4) and 32-bit floats
        u = rng.integers(0, 256, 30000).astype(np.uint8)
        it, _ = aukit.stream.pcm([int(v) for v in u], 8, "unsigned", 1, 22050)
        ref = oracle.stream_pcm(u.tobytes(), 8, oracle.UNSIGNED, 1, 22050, False, False, oracle.CUBIC)
        assert np.max(np.abs(np.concatenate([c[0] for c, _ in it]) - ref.data[0])) <= 1e-13
        f = rng.uniform(-1, 1, 20000).astype(np.float32)
        it, _ = aukit.stream.pcm([float(v) for v in f], 32, "float", 1, 48000)
        ref = oracle.stream_pcm(f.astype("<f4").tobytes(), 32, oracle.FLOAT, 1, 48000, False, False, oracle.CUBIC)
        assert np.max(np.abs(np.concatenate([c[0] for c, _ in it]) - ref.data[0])) <= 1e-13
        # numbers no string could hold (fractions, beyond the bit depth): `s / (s < 0 and maxValue or maxValue-1)` as they are (:2264) —
        # the same chunks as a float table of the quotients
        v = rng.uniform(-40000, 40000, 12000)
        a, _ = aukit.stream.pcm(list(v), 16, "signed", 1, 32000)
        b, _ = aukit.stream.pcm(list(v / np.where(v < 0, 32768.0, 32767.0)), 32, "float", 1, 32000)
        ga, gb = np.concatenate([c[0] for c, _ in a]), np.concatenate([c[0] for c, _ in b])
        # (the integer readers raise on their first read past the end, :2264 on nil, where a float reader hands out nil and the interpolators
        # fall back on their neighbours: the float table runs a few outputs further — the common part is the same numbers)
        assert 0 < len(gb) - len(ga) <= 4 and np.array_equal(ga, gb[:len(ga)])
    finally:
        aukit.defaultInterpolation = "linear"
    # ---- aukit.adpcm on nibbles
    for ch, inter, count in ((1, True, 4001), (1, True, 4000), (2, True, 6001), (2, False, 6000), (3, False, 6002)):
        nib = rng.integers(0, 16, count).astype(np.uint8)
        a = aukit.adpcm([int(v) for v in nib], ch, 22050, None, inter, [5] * ch if ch > 1 else 5, [3] * ch if ch > 1 else 3)
        used = (count // ch) * ch
        if inter:
            pad = np.concatenate([nib[:used], np.zeros(used % 2, dtype=np.uint8)])
            packed = bytes(int(pad[i]) << 4 | int(pad[i + 1]) for i in range(0, len(pad), 2))
            ref = oracle.adpcm(packed, ch, 22050, True, True, [5] * ch, [3] * ch)
            for c in range(ch):
                assert np.array_equal(a.data[c], ref.data[c][: used // ch]), (ch, inter, count, c)
        else:  # channels laid end to end, each continuing where the one before stopped reading (its own predictor / index)
            per = count // ch
            for c in range(ch):
                seg = nib[c * per:(c + 1) * per]
                pad = np.concatenate([seg, np.zeros(per % 2, dtype=np.uint8)])
                packed = bytes(int(pad[i]) << 4 | int(pad[i + 1]) for i in range(0, len(pad), 2))
                ref = oracle.adpcm(packed, 1, 22050, True, True, [5], [3])
                assert np.array_equal(a.data[c], ref.data[0][:per]), (ch, inter, count, c)
    with pytest.raises(aukit.LuaError):
        aukit.adpcm([1, 2, 16])
