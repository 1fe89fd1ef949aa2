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
