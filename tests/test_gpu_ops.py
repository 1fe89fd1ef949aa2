"""GPU parity for the callers either side of the hot path against oracle/oracle_ops.py:
structural Audio methods, generators, packing.  Bit-exact (pure data movement / integer conversion) except sine tones."""
import numpy as np
import pytest

from tests.util import signal

pytestmark = pytest.mark.gpu


def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


@pytest.fixture()
def OPS():
    from oracle import oracle_ops
    return oracle_ops


def _streams(lens, channels, seed=0):
    return [[signal(n, 48000, 7, 10 * s + c + seed)[:n] for c in range(channels)] for s, n in enumerate(lens)]


def _same(got, ref):
    assert len(got) == len(ref[0])
    for g, r in zip(got, ref[0]):
        assert np.array_equal(g, r)


@pytest.mark.parametrize("dtype", ["F64", "F32"])
def test_structural_ops_match_reference(ctx, OPS, dtype):
    B, N = _B(), _N()
    dt = getattr(N, dtype)
    cast = (lambda a: np.asarray(a, np.float32).astype(np.float64)) if dtype == "F32" else (lambda a: np.asarray(a, np.float64))
    lens = [1000, 1, 4801, 0, 77]
    x = [[cast(c) for c in s] for s in _streams(lens, 2)]
    y = [[cast(c) for c in s] for s in _streams([300, 5, 0, 10, 9000], 1, seed=50)]
    z = [[cast(c) for c in s] for s in _streams([10, 10, 10, 10, 10], 3, seed=90)]
    ax, ay, az = (B.AudioBatch.upload(ctx, v, 4800, dtype=dt) for v in (x, y, z))
    for got, i in zip(B.concat(ctx, [ax, ay, az]).download(), range(5)):
        _same(got, OPS.concat([(x[i], 4800), (y[i], 4800), (z[i], 4800)]))
    for got, i in zip(B.combine(ctx, [ax, ay, az]).download(), range(5)):
        _same(got, OPS.combine([(x[i], 4800), (y[i], 4800), (z[i], 4800)]))
    for got, i in zip(B.split(ctx, az, [3, 1]).download(), range(5)):
        _same(got, OPS.split((z[i], 4800), [3, 1])[0])
    for cnt in (3, 2.5, 1, 0.5, 0):
        for got, i in zip(B.rep(ctx, ax, cnt).download(), range(5)):
            _same(got, OPS.rep((x[i], 4800), cnt))
    for got, i in zip(B.reverse(ctx, ax).download(), range(5)):
        _same(got, OPS.reverse((x[i], 4800)))
    with pytest.raises(N.AukitError, match="out of range"):
        B.split(ctx, az, [4])
    with pytest.raises(N.AukitError, match="empty table"):
        B.split(ctx, az, [])


def test_sub_matches_reference(ctx, OPS):
    B, N = _B(), _N()
    for rate, n in ((4800, 4800 * 3), (4800, 4800 * 2 + 1234), (2.5, 10), (44100, 100000)):
        x = _streams([n], 2)
        a = B.AudioBatch.upload(ctx, x, rate, dtype=N.F64)
        length = n / rate
        for start, last in ((0, 0), (1, 2), (-1, 0), (0, -2), (1.7, 2.9), (2, 1), (0, length), (-2, -1)):
            try:
                ref = OPS.sub((x[0], rate), start, last)
            except OPS.LuaError:
                with pytest.raises(N.AukitError, match="outside of range"):
                    B.sub(ctx, a, start, last)
                continue
            _same(B.sub(ctx, a, start, last).download()[0], ref)
    with pytest.raises(N.AukitError, match="outside of range"):
        B.sub(ctx, B.AudioBatch.upload(ctx, _streams([100], 1), 100, dtype=N.F64), 2, 0)


def test_generators_match_reference(ctx, OPS):
    B, N = _B(), _N()
    z = B.tone(ctx, 3, 0, 0.25, wave="none", channels=2, sample_rate=22050, dtype=N.F64).download()
    assert len(z) == 3 and all(np.array_equal(c, np.zeros(5512)) for s in z for c in s)
    for wave, tol in (("triangle", 0), ("sawtooth", 0), ("square", 0), ("sine", 4e-16)):
        for freq, dur, amp, duty, rate in ((440, 0.5, 0.8, 0.5, 48000), (1000.5, 0.01, 1.0, 0.2, 44100), (3, 2.0, 0.25, 0.9, 8000)):
            got = B.tone(ctx, 2, freq, dur, amp, wave, duty, 2, rate, dtype=N.F64).download()
            ref = OPS.tone(freq, dur, amp, wave, duty, 2, rate)
            for s in got:
                assert len(s) == 2
                for g, r in zip(s, ref[0]):
                    assert len(g) == len(r) and np.max(np.abs(g - r), initial=0) <= tol, (wave, freq)
    with pytest.raises(N.AukitError, match="invalid wave type"):
        B.tone(ctx, 1, 440, 1, 1, 9)
    with pytest.raises(N.AukitError, match="outside of range"):
        B.tone(ctx, 1, 440, 1, 1.5, "sine")


@pytest.mark.parametrize("bits,dtype_name", [(8, "unsigned"), (8, "signed"), (16, "signed"), (24, "signed"), (32, "signed"), (16, "unsigned"), (32, "float")])
def test_pack_pcm_matches_reference(ctx, OPS, bits, dtype_name):
    B, N = _B(), _N()
    x = _streams([999, 0, 48000], 2)
    x[0][0][:4] = [1.0, -1.0, 0.0, -0.0]
    a = B.AudioBatch.upload(ctx, x, 48000, dtype=N.F64)
    for big in (False, True):
        for inter in (True, False):
            for mode in (N.PACK_TRUNC, N.PACK_FLOOR):
                got = B.pack_pcm(ctx, a, bits, dtype_name, big, inter, mode).download()
                for s in range(3):
                    ref = OPS.pack(OPS.encode_pcm((x[s], 48000), bits, dtype_name, inter), bits, dtype_name, big, mode)
                    assert got[s] == ref, (big, inter, mode, s)
    if dtype_name != "float":
        with pytest.raises(N.AukitError, match="no integer representation"):
            B.pack_pcm(ctx, a, bits, dtype_name, False, True, N.PACK_STRICT)
        whole = B.AudioBatch.upload(ctx, [[np.array([1.0, -1.0, 0.0])]], 48000, dtype=N.F64)
        assert B.pack_pcm(ctx, whole, bits, dtype_name, False, True, N.PACK_STRICT).download()[0] == OPS.pack(
            OPS.encode_pcm(([np.array([1.0, -1.0, 0.0])], 48000), bits, dtype_name), bits, dtype_name, False, OPS.STRICT)


def test_wav_sample_bytes_round_trip(ctx, oracle):
    """Audio:wav's sample bytes (aukit.lua:966-971) decoded again by aukit.pcm give the audio back to within one quantisation step."""
    B, N = _B(), _N()
    x = _streams([5000], 2)
    a = B.AudioBatch.upload(ctx, x, 48000, dtype=N.F64)
    data = B.pack_pcm(ctx, a, 16, "signed", False, True, N.PACK_TRUNC)
    back = B.decode(ctx, data, B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"), dtype=N.F64).download()[0]
    for c in range(2):
        assert np.max(np.abs(back[c] - x[0][c])) <= 1.0 / 32767


def test_mirror_structural_methods_and_wav(oracle, OPS):
    """The Lua-shaped mirror (aukit_amd.aukit): Audio:concat/sub/combine/split/rep/reverse, aukit.new/tone/pack, Audio:wav → aukit.wav."""
    import aukit_amd.aukit as aukit
    x = [signal(4800, 4800, 7, 1), signal(4800, 4800, 7, 2)]
    y = [signal(1000, 4800, 7, 3)]
    a, b = aukit.Audio.from_arrays(x, 4800), aukit.Audio.from_arrays(y, 4800)
    _same(a.concat(b).data, OPS.concat([(x, 4800), (y, 4800)]))
    _same(a.combine(b).data, OPS.combine([(x, 4800), (y, 4800)]))
    _same(a.sub(0, -0.5 - 0.5).data, OPS.sub((x, 4800), 0, -1))
    l, r = a.split([1], [2, 1])
    _same(l.data, OPS.split((x, 4800), [1])[0]) and _same(r.data, OPS.split((x, 4800), [2, 1])[0])
    _same(a.rep(2).reverse().data, OPS.reverse(OPS.rep((x, 4800), 2)))
    with pytest.raises(aukit.LuaError, match="out of range"):
        a.split([3])
    # a different sample rate is resampled first (aukit.lua:702), with aukit.defaultInterpolation
    c = aukit.Audio.from_arrays([signal(2400, 2400, 7, 4)], 2400)
    ref_c = oracle.resample(oracle.Audio([signal(2400, 2400, 7, 4)], 2400), 4800, oracle.INTERP[aukit.defaultInterpolation])
    got = a.concat(c).data
    assert len(got[0]) == 4800 + len(ref_c.data[0]) and np.max(np.abs(got[0][4800:] - ref_c.data[0])) <= 1e-15 and np.all(got[1][4800:] == 0)
    t = aukit.tone(440, 0.1, 0.5, "square", 0.3, 2, 8000)
    _same(t.data, OPS.tone(440, 0.1, 0.5, "square", 0.3, 2, 8000))
    assert aukit.new(0.01, 2, 8000).data[1].tolist() == [0.0] * 80
    assert aukit.pack([1, -2, 300.7], 16, "signed", True) == OPS.pack([1, -2, 300.7], 16, "signed", True, OPS.TRUNC)
    # Audio:wav → aukit.wav: header fields the loader reads back, samples within one quantisation step
    for bits in (8, 16, 24, 32):
        w = a.wav(bits)
        assert w[:4] == b"RIFF" and int.from_bytes(w[4:8], "little") == len(w) - 8
        back = aukit.wav(w)
        assert back.sampleRate == 4800 and back.channels() == 2
        for ch in range(2):
            assert np.max(np.abs(back.data[ch] - x[ch])) <= 2.0 / (2 ** (bits - 1) - 1)
    wd = a.wav(1)  # DFPWM in WAVE_FORMAT_EXTENSIBLE
    assert wd[8:16] == b"WAVEfmt " and int.from_bytes(wd[20:22], "little") == 0xFFFE and wd[-len(a.dfpwm(True)):] == a.dfpwm(True)
    bd = aukit.wav(wd)
    assert bd.channels() == 2 and abs(len(bd.data[0]) - 4800) <= 16


def test_pcm_table_input(ctx, oracle):
    """aukit.pcm on a TABLE of numbers (aukit.lua:1077-1096, :1161-1171): values as they are — fractions, out-of-range numbers, every data
    type and layout — bit for bit the oracle's doubles; the uneven-table error; several tables per call."""
    import aukit_amd.batch as B
    import aukit_amd._native as N
    rng = np.random.default_rng(7)
    cases = [(8, "signed", 1, True), (16, "signed", 2, True), (16, "signed", 2, False), (24, "unsigned", 3, True), (8, "unsigned", 1, True), (32, "float", 2, False),
             (32, "signed", 4, True)]
    for depth, dt, ch, inter in cases:
        tabs = []
        for n in (0, ch * 1, ch * 777, ch * 4097):
            v = rng.integers(-40000, 40000, n).astype(np.float64)
            v[::5] += 0.25  # Lua numbers need not be integers
            tabs.append(v)
        d = B.make_desc(N.CODEC_PCM, ch, 44100, depth, dt, False, inter)
        for dtype, tol in ((N.F64, 0.0), (N.F32, 1e-6)):
            c2 = B.Context(0, dtype=dtype)
            got = B.decode_table(c2, tabs, d).download()
            for v, g in zip(tabs, got):
                ref = oracle.pcm_table(v, depth, {"signed": oracle.SIGNED, "unsigned": oracle.UNSIGNED, "float": oracle.FLOAT}[dt], ch, 44100, inter)
                assert len(g) == ch
                for c in range(ch):
                    assert len(g[c]) == len(ref.data[c])
                    if tol == 0.0:
                        assert np.array_equal(g[c], ref.data[c]), (depth, dt, ch, inter, c)
                    elif len(g[c]):
                        assert np.max(np.abs(g[c] - ref.data[c]) / np.maximum(1.0, np.abs(ref.data[c]))) <= tol
            c2.close()
    with pytest.raises(N.AukitError, match="uneven amount of data per channel"):
        B.decode_table(ctx, [np.arange(5.0)], B.make_desc(N.CODEC_PCM, 2, 48000, 8, "signed"))
    # the mirror: a Python list is a Lua table
    from aukit_amd import aukit as A
    a = A.pcm([0, 127, -128, 64.5], 8, "signed", 2, 48000)
    assert a.channels() == 2 and a.len() == 2 / 48000
    assert np.array_equal(a.data[0], [0.0, -1.0]) and np.array_equal(a.data[1], [1.0, 64.5 / 127])


def test_mirror_wav_writes_the_list_info_chunk(oracle):
    """Audio:wav with metadata (aukit.lua:946-956, :980-996): "LIST" .. s4("INFO" .. (tag .. s4(tostring(value)) .. pad to even) ...) between the
    format (and fact) chunk and "data"; the RIFF size field does not count it (as the reference writes it); aukit.wav walks it back."""
    import struct
    import aukit_amd.aukit as aukit
    x = [signal(480, 4800, 7, 5)]
    a = aukit.Audio.from_arrays(x, 4800)
    plain = a.wav(16)
    a.metadata = {"title": "Song", "artist": "Me!", "trackNumber": 7, "unknown key": "dropped"}
    lst = (b"INFO" + b"INAM" + struct.pack("<I", 4) + b"Song" + b"IART" + struct.pack("<I", 3) + b"Me!" + b"\0" + b"IPRT" + struct.pack("<I", 1) + b"7" + b"\0")
    want = plain[:36] + b"LIST" + struct.pack("<I", len(lst)) + lst + plain[36:]
    got = a.wav(16)
    assert got == want and got[4:8] == plain[4:8]
    back = aukit.wav(got)
    assert np.array_equal(back.data[0], aukit.wav(plain).data[0])
    wd, pd = a.wav(1), aukit.Audio.from_arrays(x, 4800).wav(1)
    assert wd == pd[:72] + b"LIST" + struct.pack("<I", len(lst)) + lst + pd[72:]   # behind "fact": 12 + (8 + 40) + (8 + 4) = 72
    assert len(aukit.wav(wd).data[0]) == len(aukit.wav(pd).data[0])


def test_noise_generator(ctx):
    """aukit.noise (aukit.lua:1840-1853) on the device: (random() * 2 - 1) * amplitude from a Philox stream keyed by a seed — not the reference's
    math.random samples (nobody can reproduce those), so what is checked is what the reference's loop guarantees: the length, the range, and that it
    is noise (mean, variance of a uniform, nothing shared between channels, streams or seeds); plus the seed's promise."""
    import aukit_amd.aukit as aukit
    B, N = _B(), _N()
    a = B.noise(ctx, 3, 0.5, 0.8, 4, 48000, seed=1234, dtype=N.F64).download()
    b = B.noise(ctx, 3, 0.5, 0.8, 4, 48000, seed=1234, dtype=N.F64).download()
    c = B.noise(ctx, 3, 0.5, 0.8, 4, 48000, seed=1235, dtype=N.F64).download()
    rows = [r for s in a for r in s]
    assert len(rows) == 12 and all(len(r) == 24000 for r in rows)
    for r in rows:
        assert np.max(np.abs(r)) <= 0.8 and np.max(np.abs(r)) > 0.79
        assert abs(np.mean(r)) < 0.02 and abs(np.var(r) - 0.8 ** 2 / 3) < 0.01
        assert abs(np.corrcoef(r[:-1], r[1:])[0, 1]) < 0.03          # white
    for i in range(12):
        for j in range(i + 1, 12):
            assert abs(np.corrcoef(rows[i], rows[j])[0, 1]) < 0.03   # every channel of every stream its own draw
    assert all(np.array_equal(x, y) for s, t in zip(a, b) for x, y in zip(s, t))
    assert all(abs(np.corrcoef(x, y)[0, 1]) < 0.03 for s, t in zip(a, c) for x, y in zip(s, t))
    f = B.noise(ctx, 1, 0.01, 1.0, 1, 48000, seed=9, dtype=N.F32).download()[0][0]
    assert len(f) == 480 and np.max(np.abs(f)) <= 1.0 and np.array_equal(f, f.astype(np.float32))   # (download hands doubles: f32 storage shows in the values)
    # the mirror: argument checks as the reference's, a fresh audio per call without a seed
    m1, m2 = aukit.noise(0.05, 0.5, 2, 8000), aukit.noise(0.05, 0.5, 2, 8000)
    assert m1.channels() == 2 and len(m1.data[0]) == 400 and not np.array_equal(m1.data[0], m2.data[0])
    assert np.array_equal(aukit.noise(0.05, 0.5, 1, 8000, seed=3).data[0], aukit.noise(0.05, 0.5, 1, 8000, seed=3).data[0])
    with pytest.raises(aukit.LuaError):
        aukit.noise(0.05, 1.5)
    with pytest.raises(aukit.LuaError):
        aukit.noise(0.05, 0.5, 0)
