"""GPU parity: Audio:mono / :mix / :pcm and aukit.effects.* vs the CPU oracle (through the C ABI).

Maps are evaluated in the reference's fp64 order → bit-identical with AUKIT_F64 storage.  Scans (lowpass,
highpass) and tree reductions (center) re-associate → ≤ 1e-12; everything is ≤ 1e-6 RMS with AUKIT_F32.
"""
import numpy as np
import pytest

from tests.util import rms, signal

pytestmark = pytest.mark.gpu


def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


def _audios(rate=22050, lens=(30000, 777, 4097), ch=2, cfg=7):
    return [[signal(n, rate, cfg, 10 * i + c) for c in range(ch)] for i, n in enumerate(lens)]


CASES = [
    ("amplify", (1.5,), lambda O, a: O.fx_amplify(a, 1.5), 0.0),
    ("amplify", (1.0,), lambda O, a: O.fx_amplify(a, 1.0), 0.0),
    ("invert", (), lambda O, a: O.fx_invert(a), 0.0),
    ("normalize", (0.8,), lambda O, a: O.fx_normalize(a, 0.8), 0.0),
    ("normalize", (1.0, 1.0), lambda O, a: O.fx_normalize(a, 1.0, True), 0.0),
    ("center", (), lambda O, a: O.fx_center(a), 1e-13),
    ("delay", (0.01, 0.5), lambda O, a: O.fx_delay(a, 0.01, 0.5), 0.0),
    ("delay", (0.0, 0.25), lambda O, a: O.fx_delay(a, 0.0, 0.25), 0.0),
    ("echo", (0.01, 0.5), lambda O, a: O.fx_echo(a, 0.01, 0.5), 0.0),
    ("echo", (0.0005, 0.9), lambda O, a: O.fx_echo(a, 0.0005, 0.9), 0.0),
    ("lowpass", (11025.0,), lambda O, a: O.fx_lowpass(a, 11025.0), 1e-12),
    ("lowpass", (200.0,), lambda O, a: O.fx_lowpass(a, 200.0), 1e-12),
    ("highpass", (20.0,), lambda O, a: O.fx_highpass(a, 20.0), 1e-11),
    ("highpass", (3000.0,), lambda O, a: O.fx_highpass(a, 3000.0), 1e-12),
    ("fade", (0.1, 1.0, 0.2, 0.0), lambda O, a: O.fx_fade(a, 0.1, 1.0, 0.2, 0.0), 0.0),
]


@pytest.mark.parametrize("name,args,ref_fn,tol", CASES)
def test_effect_f64(ctx, oracle, name, args, ref_fn, tol):
    B, N = _B(), _N()
    lens = (30000, 4410, 5000) if name == "fade" else (30000, 777, 4097)
    a = _audios(lens=lens)
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    B.effect(ctx, ab, name, *args)
    got = ab.download()
    for s in range(len(a)):
        ref = ref_fn(oracle, oracle.Audio(a[s], 22050))
        for c in range(2):
            err = np.max(np.abs(got[s][c] - ref.data[c]))
            assert err <= tol, (name, s, c, err)


def test_effects_f32_tolerance(ctx, oracle):
    B, N = _B(), _N()
    a = _audios(lens=(48000, 5000))
    for name, args, ref_fn, _ in CASES:
        if name == "fade":
            continue
        ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F32)
        B.effect(ctx, ab, name, *args)
        got = ab.download()
        for s in range(len(a)):
            ref = ref_fn(oracle, oracle.Audio([x.astype(np.float32).astype(np.float64) for x in a[s]], 22050))
            for c in range(2):
                assert rms(got[s][c], ref.data[c]) <= 1e-6, name


def test_reverb(ctx, oracle):
    B, N = _B(), _N()
    a = _audios(rate=22050, lens=(30000, 9000), ch=2)
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    B.effect(ctx, ab, "reverb", 100.0, 0.3, 1.0, 0.0)
    got = ab.download()
    for s in range(2):
        ref = oracle.fx_reverb(oracle.Audio(a[s], 22050), 100.0, 0.3, 1.0, 0.0)
        for c in range(2):
            assert np.max(np.abs(got[s][c] - ref.data[c])) <= 1e-13
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    B.effect(ctx, ab, "reverb", 60.0, 0.4, 0.7, 0.3)
    ref = oracle.fx_reverb(oracle.Audio(a[0], 22050), 60.0, 0.4, 0.7, 0.3)
    assert np.max(np.abs(ab.download()[0][1] - ref.data[1])) <= 1e-13


@pytest.mark.parametrize("rate,args,fused", [(48000, (100.0, 0.3, 1.0, 0.0), True), (44100, (100.0, 0.3, 0.8, 0.2), True), (22050, (100.0, 0.3, 1.0, 0.0), True),
                                             (48000, (60.0, 0.45, 0.7, 0.3), True), (48000, (130.0, 0.3, 1.0, 0.0), True), (48000, (250.0, 0.3, 1.0, 0.0), False), (96000, (100.0, 0.3, 1.0, 0.0), False),
                                             (8000, (100.0, 0.3, 1.0, 0.0), False), (48000, (700.0, 0.3, 1.0, 0.0), False)])
def test_reverb_f32_in_one_pass(ctx, oracle, rate, args, fused):
    """effects.reverb on an F32 audio: k_reverb_f32 (four comb rings + the all-pass ring in LDS, one pass over the row) within 1e-6 RMS of
    the oracle; rates / delays whose state does not fit (or whose lags are shorter than a block) keep the multi-launch path."""
    B, N = _B(), _N()
    S = int(np.floor(0.08927 * rate))
    lens = (rate * 2 + 17, S + 1, S + 2, S + 5000, 3 * S)
    a = [[signal(n, rate, 2, 2 * i + c).astype(np.float32).astype(np.float64) for c in range(2)] for i, n in enumerate(lens)]
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F32)
    B.effect(ctx, ab, "reverb", *args)
    assert (ctx.last_kernel()[0] == "k_reverb_f32") == fused, ctx.last_kernel()
    got = ab.download()
    for s in range(len(a)):
        ref = oracle.fx_reverb(oracle.Audio(a[s], rate), *args)
        for c in range(2):
            assert rms(got[s][c], ref.data[c]) <= 1e-6, (s, c)
            assert np.max(np.abs(got[s][c] - ref.data[c])) <= 4e-6, (s, c)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("rate", [224.5, 230, 235, 236, 300])
def test_reverb_at_low_sample_rates(ctx, oracle, rate):
    """floor(0.08927 * rate) == 20 for rates of about 224.1 … 235.2 Hz: the second all-pass tap of aukit.lua:3575 is then the element
    itself (sum[i + 20 - samples] = sum[i]); the block-parallel all-pass must neither hang (its block width was samples - 20 = 0) nor
    differ from the oracle.  236 / 300 Hz: the first rates with one / six elements per block."""
    B, N = _B(), _N()
    a = _audios(rate=22050, lens=(4000, 700), ch=2)
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F64)
    B.effect(ctx, ab, "reverb", 100.0, 0.3, 0.8, 0.2)
    got = ab.download()
    for s in range(2):
        ref = oracle.fx_reverb(oracle.Audio(a[s], rate), 100.0, 0.3, 0.8, 0.2)
        for c in range(2):
            assert np.max(np.abs(got[s][c] - ref.data[c])) <= 1e-13


def test_speed_and_trim(ctx, oracle):
    B, N = _B(), _N()
    a = _audios(lens=(20000, 3000))
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    B.effect(ctx, ab, "speed", 1.25, 1)  # default interpolation = linear
    got = ab.download()
    for s in range(2):
        ref = oracle.fx_speed(oracle.Audio(a[s], 22050), 1.25, oracle.LINEAR)
        assert len(got[s][0]) == len(ref.data[0])
        assert np.max(np.abs(got[s][1] - ref.data[1])) <= 1e-15
    assert ab.info()["sample_rate"] == 22050
    with pytest.raises(N.AukitError) as e:  # the reference always raises here (Q17)
        B.effect(ctx, ab, "trim", 1 / 65536)
    assert "string expected, got table" in str(e.value)


def test_fade_error_cases_match_reference(ctx):
    B, N = _B(), _N()
    ab = B.AudioBatch.upload(ctx, _audios(lens=(1000,)), 22050, dtype=N.F64)
    with pytest.raises(N.AukitError):  # startTime = 0 indexes ch[0] (nil)
        B.effect(ctx, ab, "fade", 0.0, 1.0, 0.01, 0.0)
    with pytest.raises(N.AukitError):  # runs past the end
        B.effect(ctx, ab, "fade", 0.01, 1.0, 1.0, 0.0)
    B.effect(ctx, ab, "fade", 0.01, 1.0, 0.02, 1.0)  # both amplitudes 1: no-op, no error


@pytest.mark.parametrize("count", [9, 33])
def test_mix_any_number_of_audios(ctx, oracle, count):
    """Audio:mix sums whatever `...` holds (aukit.lua:804-835): beyond eight audios the sources' table travels in device memory (k_mix_many);
    the sum runs in the argument list's order, term by term, like the reference's — bit-exact in F64"""
    B, N = _B(), _N()
    lens = [(3000 + 37 * k, 200 - k) for k in range(count)]
    chans = [1 + (k % 3) for k in range(count)]
    data = [[[signal(n, 22050, 8, 11 * k + 3 * i + c) * (0.2 + 0.01 * k) for c in range(chans[k])] for i, n in enumerate(lens[k])] for k in range(count)]
    abs_ = [B.AudioBatch.upload(ctx, d, 22050, dtype=N.F64) for d in data]
    mixed = B.mix(ctx, abs_, 0.31).download()
    for s in range(2):
        ref = oracle.mix([oracle.Audio(data[k][s], 22050) for k in range(count)], 0.31)
        assert len(mixed[s]) == max(chans)
        for c in range(max(chans)):
            assert np.array_equal(mixed[s][c], ref.data[c])


def test_mono_mix_encode_pcm(ctx, oracle):
    B, N = _B(), _N()
    a = _audios(lens=(5000, 123), ch=3)
    b = [[signal(n, 22050, 8, 5 * i + c) for c in range(2)] for i, n in enumerate((7000, 50))]
    ab, bb = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64), B.AudioBatch.upload(ctx, b, 22050, dtype=N.F64)
    mono = B.mono(ctx, ab).download()
    mixed = B.mix(ctx, [ab, bb], 0.7).download()
    for s in range(2):
        oa, ob = oracle.Audio(a[s], 22050), oracle.Audio(b[s], 22050)
        assert np.array_equal(mono[s][0], oracle.mono(oa).data[0])
        ref = oracle.mix([oa, ob], 0.7)
        assert len(mixed[s]) == 3
        for c in range(3):
            assert np.array_equal(mixed[s][c], ref.data[c])
    for bits, dt, inter in ((8, "signed", True), (16, "unsigned", False), (32, "float", True)):
        got = B.encode_pcm(ctx, ab, bits, dt, inter).download()
        for s in range(2):
            ref = oracle.encode_pcm(oracle.Audio(a[s], 22050), bits, oracle.DTYPE[dt], inter)
            assert np.array_equal(got[s][0], ref)


def test_auplay_pipeline(ctx, oracle):
    """auplay.lua:20-31: resample(48000) → mono → normalize(0.8) → lowpass(sr/2), default (linear) interpolation."""
    B, N = _B(), _N()
    from tests.util import pcm16
    st = np.stack([pcm16(20000, 44100, 5, 0), pcm16(20000, 44100, 5, 1)], 1).tobytes()
    bt = B.Batch.upload(ctx, [st])
    desc = B.make_desc(N.CODEC_PCM, 2, 44100, 16, "signed")
    au = B.decode(ctx, bt, desc, dtype=N.F64)
    rs = B.resample(ctx, au, 48000, "linear")
    mo = B.mono(ctx, rs)
    B.effect(ctx, mo, "normalize", 0.8)
    B.effect(ctx, mo, "lowpass", 44100 / 2)
    got = mo.download()[0][0]
    ref = oracle.fx_lowpass(oracle.fx_normalize(oracle.mono(oracle.resample(oracle.pcm(st, 16, oracle.SIGNED, 2, 44100), 48000, oracle.LINEAR)), 0.8), 22050.0)
    assert np.max(np.abs(got - ref.data[0])) <= 1e-12


@pytest.mark.parametrize("dt", ["F64", "F32"])
@pytest.mark.parametrize("independent", [False, True])
@pytest.mark.parametrize("first", ["highpass", "lowpass", "none"])
def test_deferred_normalize_and_reused_row_maxima(ctx, oracle, monkeypatch, first, independent, dt):
    """config 5's tail (aukit.lua:3604-3618, :3439-3456, :682-687).  effects.normalize takes its peak from the per-row maxima the
    filter pass left behind and defers its scaling to the next reader: Audio:mono applies it while reading, anything else (a download,
    another effect, a second mono) materialises it.  Every observable result is bit-identical to the five-pass sequence
    (AUKIT_NO_TAIL_FUSION=1) and within tolerance of the oracle."""
    B, N = _B(), _N()
    dtype = getattr(N, dt)
    a = _audios(rate=22050, lens=(30000, 1, 2, 4097, 700), ch=2)
    src = [[x.astype(np.float32).astype(np.float64) for x in s] for s in a] if dt == "F32" else a

    def run():
        ab = B.AudioBatch.upload(ctx, src, 22050, dtype=dtype)
        if first != "none":
            B.effect(ctx, ab, first, 300.0)
        B.effect(ctx, ab, "normalize", 0.8, 1.0 if independent else 0.0)
        m1 = B.mono(ctx, ab).download()          # fused: the map is applied while mono reads
        st = ab.download()                       # materialises the deferred map
        m2 = B.mono(ctx, ab).download()          # plain mono of the materialised rows
        B.effect(ctx, ab, "normalize", 0.5)      # a second normalize: peak search from scratch, deferred again ...
        B.effect(ctx, ab, "invert")              # ... and flushed by the next effect
        return m1, st, m2, ab.download()

    got = run()
    monkeypatch.setenv("AUKIT_NO_TAIL_FUSION", "1")
    plain = run()
    monkeypatch.delenv("AUKIT_NO_TAIL_FUSION")
    for g, p in zip(got, plain):
        for s in range(len(a)):
            for c in range(len(g[s])):
                assert np.array_equal(g[s][c], p[s][c]), (s, c)
    tol = 1e-12 if dt == "F64" else 1e-6
    for s in range(len(a)):
        ref = oracle.Audio([x.copy() for x in src[s]], 22050)
        if first == "highpass":
            ref = oracle.fx_highpass(ref, 300.0)
        elif first == "lowpass":
            ref = oracle.fx_lowpass(ref, 300.0)
        ref = oracle.fx_normalize(ref, 0.8, independent)
        assert np.array_equal(got[0][s][0], got[2][s][0])  # fused mono == mono of the materialised rows
        assert rms(got[0][s][0], oracle.mono(ref).data[0]) <= tol
        for c in range(2):
            assert rms(got[1][s][c], ref.data[c]) <= tol
