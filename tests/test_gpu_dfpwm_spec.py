"""The chunk-speculative DFPWM transcoder (aukit_amd/csrc/dfpwm_spec.hip): aukit.dfpwm(d, 2, rate):mono():dfpwm()
(aukit.lua:1392-1414, :677-689, :1005-1018) with a lane per (stream, time chunk).  Whatever the speculation does — right guesses,
wrong classes, warm-ups too short to converge, chunk boundaries anywhere — the bytes are the oracle's and the serial kernel's."""
import numpy as np
import pytest

from util import signal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _speculate_on_small_batches(monkeypatch):
    """by default a few short streams (<= 16 of <= 40 stream-seconds together) stay with the exact parallel encoder k_dfe_* (dfx_run): the
    schedule under test here takes them all the same"""
    monkeypatch.setenv("AUKIT_DFX_FEW", "0")


def _B():
    from aukit_amd import batch
    return batch


def _N():
    from aukit_amd import _native
    return _native


def _enc_stereo(oracle, l, r):
    return oracle.dfpwm_encode(np.stack([np.round(l), np.round(r)], 1).ravel())


def _ref(oracle, s):
    return oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(s, 2, 48000)), True) if s else b""


ENVS = ({}, {"AUKIT_DFX_WE": "64", "AUKIT_DFX_G": "1"}, {"AUKIT_DFX_WE": "64", "AUKIT_DFX_ROUNDS": "1", "AUKIT_DFX_G": "2"}, {"AUKIT_DFX_CHUNKS": "1000", "AUKIT_DFX_MIN_BPC": "1"}, {"AUKIT_DFX_ROUNDS": "1"},
        {"AUKIT_DFX_WE": "128", "AUKIT_DFX_CHUNKS": "7", "AUKIT_DFX_ROUNDS": "2", "AUKIT_DFX_NOPROBE": "1"}, {"AUKIT_DFX_WPS": "1", "AUKIT_DFX_WE": "1008"})


def _all_schedules(ctx, monkeypatch, bt, want):
    B = _B()
    for env in ENVS:
        env = dict(env, AUKIT_DFX_NOPROBE="1")   # (the probe would decline the batches with silence and noise in them: here the schedule itself is under test)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        name = ctx.last_kernel()[0]
        for k in env:
            monkeypatch.delenv(k)
        assert name == "k_dfx_chunks", (env, name)
        assert got == want, env


def test_spec_transcode_config4_signal(ctx, oracle, monkeypatch):
    """the config-4 signal, streams of six lengths (a Q10 slice boundary inside, an exact multiple of the block, short ones)"""
    B = _B()
    streams = []
    for i, n in enumerate((120000, 24000, 6000, 6002, 45056, 30001)):
        streams.append(_enc_stereo(oracle, signal(n * 4, 48000, 4, 2 * i) * 100, signal(n * 4, 48000, 4, 2 * i + 1) * 90))
    bt = B.Batch.upload(ctx, streams)
    want = [_ref(oracle, s) for s in streams]
    assert len(want[0]) == 60010
    monkeypatch.setenv("AUKIT_DFPWM_SERIAL", "1")
    assert B.dfpwm_transcode_mono(ctx, bt, 2).download() == want
    monkeypatch.delenv("AUKIT_DFPWM_SERIAL")
    _all_schedules(ctx, monkeypatch, bt, want)


def test_spec_transcode_ragged_batch(ctx, oracle, monkeypatch):
    """150 streams of six lengths (an empty one, one of a single slice, odd byte counts), unaligned starts: lanes of one wave at different
    places of their streams"""
    B = _B()
    base = []
    for i, n in enumerate((30000, 0, 6000, 18016, 6001, 12345 * 2 + 1)):
        base.append(_enc_stereo(oracle, signal(n * 4, 48000, 4, 60 + 2 * i) * 100, signal(n * 4, 48000, 4, 61 + 2 * i) * 90) if n else b"")
    order = [(i * 5 + i // 7) % 6 for i in range(150)]
    bt = B.Batch.upload(ctx, [base[c] for c in order])
    refs = [_ref(oracle, b) for b in base]
    want = [refs[c] for c in order]
    _all_schedules(ctx, monkeypatch, bt, want)


def test_spec_transcode_class_changes_and_noise(ctx, oracle, monkeypatch):
    """inputs on which the single guess is wrong: silence in front of and inside the signal (the true encoder sits at its strength floor and
    leaves it in another class: the verify pass re-speculates the rest), random bytes (noise: the class changes all the time; the verify
    lanes run most blocks again), saturating patterns"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(7))
    n = 60000 * 4
    sig = [signal(n, 48000, 4, 90 + k) * 100 for k in range(4)]
    gate = np.ones(n)
    gate[: n // 5] = 0
    gate[n // 2: n // 2 + n // 8] = 0
    streams = [_enc_stereo(oracle, sig[0] * gate, sig[1] * gate), _enc_stereo(oracle, sig[2], sig[3] * gate),
               bytes(rng.integers(0, 256, 60000, dtype=np.uint8)), b"\xaa" * 30000 + bytes(rng.integers(0, 256, 30000, dtype=np.uint8)),
               b"\xff" * 20000 + b"\x00" * 20000 + b"\x0f" * 20000]
    bt = B.Batch.upload(ctx, streams)
    want = [_ref(oracle, s) for s in streams]
    _all_schedules(ctx, monkeypatch, bt, want)
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    monkeypatch.setenv("AUKIT_DFX_NOPROBE", "1")
    B.dfpwm_transcode_mono(ctx, bt, 2)
    monkeypatch.delenv("AUKIT_DFX_NOPROBE")
    ctx.sync()
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    assert ctx.counter(N.COUNTER_DFPWM_CHUNKS) > 0
    assert ctx.counter(N.COUNTER_DFPWM_RESPECULATED) >= 1   # the gated streams left the floor in a class the prologue could not know
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    monkeypatch.setenv("AUKIT_DFX_NOPROBE", "1")
    monkeypatch.setenv("AUKIT_DFX_ROUNDS", "1")   # no re-speculation: whatever the one round leaves is given up on and goes to the lane-per-stream encoder
    assert B.dfpwm_transcode_mono(ctx, bt, 2).download() == want
    monkeypatch.delenv("AUKIT_DFX_NOPROBE")
    monkeypatch.delenv("AUKIT_DFX_ROUNDS")
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    assert ctx.counter(N.COUNTER_DFPWM_HARD) >= 1
    # by default one guess per stream is tried first (the probe): whatever it decides, the same bytes; cut into few chunks per stream (what a large
    # batch gets: a failed speculation would cost it a whole step) streams that start in silence count against the batch, and this one is declined
    assert B.dfpwm_transcode_mono(ctx, bt, 2).download() == want
    monkeypatch.setenv("AUKIT_DFX_CHUNKS", "7")
    got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    monkeypatch.delenv("AUKIT_DFX_CHUNKS")
    assert ctx.last_kernel()[0] != "k_dfx_chunks" and got == want


def test_spec_transcode_leading_silence_takes_one_round(ctx, oracle, monkeypatch):
    """streams that START in digital silence (0.1 - 0.8 s of it, from the first or the second sample: 0x55 or 0xAA bytes): k_dfx_onset finds where it
    ends, the prologue walks there and leaves a second reference behind the onset — the lanes in the signal model their guess on that one and the
    batch is through in ONE round, also cut the way a large batch is (few chunks per stream: no second round to be had).  Silence BEHIND the
    leading one is found by the strength scan: more than eight such streams in a batch cut that way and it is declined."""
    B, N = _B(), _N()
    n = 60000 * 4
    streams = []
    for k, lead in enumerate((4800, 9601, 20000, 38400, 12000, 7000, 30001, 16000, 5000, 26000)):
        l, r = signal(n, 48000, 4, 120 + 2 * k) * 100, signal(n, 48000, 4, 121 + 2 * k) * 90
        l[k % 2: lead] = 0
        r[k % 2: lead] = 0
        streams.append(_enc_stereo(oracle, l, r))
    assert all(b"\x55" * 64 in s[:600] or b"\xaa" * 64 in s[:600] for s in streams)
    bt = B.Batch.upload(ctx, streams)
    want = [_ref(oracle, s) for s in streams]
    for env in ({}, {"AUKIT_DFX_CHUNKS": "7"}, {"AUKIT_DFX_PROBE_ASIDE": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx.set_option(N.OPT_COLLECT_STATS, 1)
        got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        name, respec, hard = ctx.last_kernel()[0], ctx.counter(N.COUNTER_DFPWM_RESPECULATED), ctx.counter(N.COUNTER_DFPWM_HARD)
        ctx.set_option(N.OPT_COLLECT_STATS, 0)
        for k in env:
            monkeypatch.delenv(k)
        assert got == want, env
        assert name == "k_dfx_chunks" and respec == 0 and hard == 0, (env, name, respec, hard)
    # Audio:dfpwm on samples that start with zeros (a PCM source's digital silence: one sample value repeated): the same second reference
    t = np.arange(n) / 48000
    pcm = []
    for k, lead in enumerate((4801, 12000, 26003, 9000, 40000, 7001, 15000, 20000, 5003, 33000)):
        x = 0.6 * np.sin(2 * np.pi * (220 + 35 * k) * t) + 0.15 * np.sin(2 * np.pi * (1900 + 111 * k) * t)
        x[:lead] = 0
        pcm.append(x)
    ab = B.AudioBatch.upload(ctx, [[x] for x in pcm], 48000, dtype=N.F64)
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    enc = B.dfpwm_encode(ctx, ab, True).download()
    name, respec = ctx.last_kernel()[0], ctx.counter(N.COUNTER_DFPWM_RESPECULATED)
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    assert enc == [oracle.audio_dfpwm(oracle.Audio([x], 48000), True) for x in pcm]
    assert name == "k_dfpwm_quantize+k_dfx_chunks<rows>" and respec == 0, (name, respec)
    # a second passage of silence inside every stream: rounds where they are to be had, declined where they are not
    inner = []
    for k in range(10):
        l, r = signal(n, 48000, 4, 150 + 2 * k) * 100, signal(n, 48000, 4, 151 + 2 * k) * 90
        l[: 9000] = 0; r[: 9000] = 0
        l[n // 2: n // 2 + 12000] = 0; r[n // 2: n // 2 + 12000] = 0
        inner.append(_enc_stereo(oracle, l, r))
    bt2 = B.Batch.upload(ctx, inner)
    want2 = [_ref(oracle, s) for s in inner]
    assert B.dfpwm_transcode_mono(ctx, bt2, 2).download() == want2 and ctx.last_kernel()[0] == "k_dfx_chunks"
    monkeypatch.setenv("AUKIT_DFX_CHUNKS", "7")
    got2 = B.dfpwm_transcode_mono(ctx, bt2, 2).download()
    monkeypatch.delenv("AUKIT_DFX_CHUNKS")
    assert got2 == want2 and ctx.last_kernel()[0] != "k_dfx_chunks"


def test_spec_transcode_mid_batch_speculation_holds(ctx, oracle):
    """2048 streams (the shard one GPU of eight gets of BASELINE config 4), 2 s each, 8 distinct signals: the oracle's bytes, and the single
    guess is right for nearly every chunk (what the speed rests on)"""
    B, N = _B(), _N()
    K = 8
    base = [_enc_stereo(oracle, signal(96000, 48000, 4, 2 * i) * 100, signal(96000, 48000, 4, 2 * i + 1) * 90) for i in range(K)]
    bt = B.Batch.upload(ctx, [base[i % K] for i in range(2048)])
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    assert ctx.last_kernel()[0] == "k_dfx_chunks"
    chunks, redone, respec = (ctx.counter(c) for c in (N.COUNTER_DFPWM_CHUNKS, N.COUNTER_DFPWM_CHUNKS_REDONE, N.COUNTER_DFPWM_RESPECULATED))
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    for c in range(K):
        assert all(g == got[c] for g in got[c::K])
        assert got[c] == _ref(oracle, base[c])
    assert respec == 0 and redone <= chunks // 50, (chunks, redone, respec)


def test_spec_transcode_many_short_and_few_long_streams(ctx, oracle):
    """the two ends of the planner: 40 000 one-second streams (more streams than the chip has lanes for two chunks each: the cut degenerates to
    two chunks per stream, one round) with a fifth of a second of leading silence on every other class, and three two-minute streams (thousands of chunks each, the
    verify lane's long walk) — the oracle's bytes for every class"""
    B, N = _B(), _N()
    K = 8
    short = []
    for i in range(K):
        l, r = signal(48000, 48000, 4, 40 + 2 * i) * 100, signal(48000, 48000, 4, 41 + 2 * i) * 90
        if i % 2:
            l[:9600] = 0; r[:9600] = 0
        short.append(_enc_stereo(oracle, l, r))
    bt = B.Batch.upload(ctx, [short[i % K] for i in range(40000)])
    got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    assert ctx.last_kernel()[0] == "k_dfx_chunks"
    for c in range(K):
        assert got[c] == _ref(oracle, short[c])
        assert all(g == got[c] for g in got[c::K])
    del bt, got
    n = 48000 * 120
    long_ = []
    for i in range(3):
        l, r = signal(n, 48000, 4, 60 + 2 * i) * 100, signal(n, 48000, 4, 61 + 2 * i) * 90
        if i == 1:
            l[:96000] = 0; r[:96000] = 0
        long_.append(_enc_stereo(oracle, l, r))
    got = B.dfpwm_transcode_mono(ctx, B.Batch.upload(ctx, long_), 2).download()
    assert ctx.last_kernel()[0] == "k_dfx_chunks"
    assert got == [_ref(oracle, s) for s in long_]


ENC_ENVS = ({}, {"AUKIT_DFX_WE": "64", "AUKIT_DFX_G": "1"}, {"AUKIT_DFX_CHUNKS": "1000", "AUKIT_DFX_MIN_BPC": "1"}, {"AUKIT_DFX_ROUNDS": "1"},
            {"AUKIT_DFX_WE": "128", "AUKIT_DFX_CHUNKS": "7", "AUKIT_DFX_ROUNDS": "2"}, {"AUKIT_DFX_WPS": "1", "AUKIT_DFX_WE": "1008"})


def test_spec_encoder_audio_dfpwm(ctx, oracle, monkeypatch):
    """Audio:dfpwm (aukit.lua:1005-1018) through the same engine on the quantized samples (k_dfx_chunks<rows>): signal, silence in front and inside,
    noise, rails, lengths that are no multiple of 4 or 8 (the padded last byte), stereo interleaved and channel after channel — the oracle's bytes
    under every schedule, and with the probe deciding"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(23))
    n = 150001
    t = np.arange(n) / 48000
    gate = np.ones(n); gate[:30000] = 0; gate[80000:95000] = 0
    sigs = [0.6 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.1, 0.1, n), (0.5 * np.sin(2 * np.pi * 330 * t) + rng.uniform(-0.2, 0.2, n)) * gate,
            rng.uniform(-1, 1, n), np.where((np.arange(n) // 300) % 2 == 0, 1.0, -1.0), np.zeros(n), signal(n, 48000, 4, 3)[:n - 3], signal(70006, 48000, 4, 4), signal(37, 48000, 4, 5)]
    ab = B.AudioBatch.upload(ctx, [[x] for x in sigs], 48000, dtype=N.F64)
    want = [oracle.audio_dfpwm(oracle.Audio([x], 48000), True) for x in sigs]
    monkeypatch.setenv("AUKIT_DFPWM_SERIAL", "1")
    assert B.dfpwm_encode(ctx, ab, True).download() == want
    monkeypatch.delenv("AUKIT_DFPWM_SERIAL")
    for env in ENC_ENVS:
        env = dict(env, AUKIT_DFX_NOPROBE="1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = B.dfpwm_encode(ctx, ab, True).download()
        name = ctx.last_kernel()[0]
        for k in env:
            monkeypatch.delenv(k)
        assert name == "k_dfpwm_quantize+k_dfx_chunks<rows>", (env, name)
        assert got == want, env
    assert B.dfpwm_encode(ctx, ab, True).download() == want   # the probe's decision, whatever it is
    st = [[sigs[0][:100003], sigs[1][:100003]], [sigs[5][:60000], sigs[0][:60000]]]
    ab2 = B.AudioBatch.upload(ctx, st, 48000, dtype=N.F64)
    for inter in (True, False):
        monkeypatch.setenv("AUKIT_DFX_NOPROBE", "1")
        got = B.dfpwm_encode(ctx, ab2, inter).download()
        monkeypatch.delenv("AUKIT_DFX_NOPROBE")
        assert ctx.last_kernel()[0] == "k_dfpwm_quantize+k_dfx_chunks<rows>"
        for s in range(2):
            assert got[s] == oracle.audio_dfpwm(oracle.Audio(st[s], 48000), inter), (inter, s)


def test_spec_encoder_mid_batch(ctx, oracle):
    """the regime between "a few streams" and "a group of 64 per CU" (Audio:dfpwm on a mid-size batch): 512 mono streams of 2 s, 8 distinct
    signals — the oracle's bytes, class identity, and nearly every guess right"""
    B, N = _B(), _N()
    K = 8
    base = [np.round(signal(96000, 48000, 4, 20 + i) * 100) / 127 for i in range(K)]
    ab = B.AudioBatch.upload(ctx, [[base[i % K]] for i in range(512)], 48000, dtype=N.F64)
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    got = B.dfpwm_encode(ctx, ab, True).download()
    assert ctx.last_kernel()[0] == "k_dfpwm_quantize+k_dfx_chunks<rows>"
    chunks, redone, hard = (ctx.counter(c) for c in (N.COUNTER_DFPWM_CHUNKS, N.COUNTER_DFPWM_CHUNKS_REDONE, N.COUNTER_DFPWM_HARD))
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    for c in range(K):
        assert all(g == got[c] for g in got[c::K])
        assert got[c] == oracle.audio_dfpwm(oracle.Audio([base[c]], 48000), True)
    assert hard == 0 and redone <= chunks // 50, (chunks, redone, hard)


def _class_pcm(kind, n, seed, rng):
    """n stereo frames (int8-valued floats) of one input class of tools/r06_dfx_grid.py: signal / lead (silence in front) / gated (silence in front
    and inside) / None for noise (random DFPWM bytes)"""
    l, r = signal(n, 48000, 4, seed) * 100, signal(n, 48000, 4, seed + 1) * 90
    if kind == "lead":
        l[: n // 20] = 0; r[: n // 20] = 0
    if kind == "gated":
        l[: n // 5] = 0; r[: n // 5] = 0
        l[n // 2: n // 2 + n // 10] = 0; r[n // 2: n // 2 + n // 10] = 0
    return l, r


@pytest.mark.parametrize("n", [1, 5, 8, 17, 64])
@pytest.mark.parametrize("kind", ["signal", "lead", "gated", "noise"])
def test_grid_cells_bytes(ctx, oracle, monkeypatch, n, kind):
    """every cell of tools/r06_dfx_grid.py at reduced length: batch size x input class x entry point — the default engine's
    bytes are the older schedule's (AUKIT_DFPWM_NOSPEC=1) and, on the first streams, the oracle's.  (The grid's times: profiles/r06_dfx_grid.txt.)"""
    B, N = _B(), _N()
    monkeypatch.delenv("AUKIT_DFX_FEW")   # the DEFAULT routing is under test here: few short streams stay with the exact parallel encoder
    rng = np.random.default_rng(1000 * n + len(kind))
    nb = 36000   # stereo DFPWM bytes per stream: 144 000 frames, three seconds
    if kind == "noise":
        streams = [rng.integers(0, 256, nb, dtype=np.uint8).tobytes() for _ in range(n)]
    else:
        distinct = [_enc_stereo(oracle, *_class_pcm(kind, nb * 4, 900 + 2 * i, rng)) for i in range(min(n, 3))]
        streams = [distinct[i % len(distinct)] for i in range(n)]
    bt = B.Batch.upload(ctx, streams)
    d = B.make_desc(N.CODEC_DFPWM, 2, 48000)

    def run():
        t = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        mono = B.mono(ctx, B.decode(ctx, bt, d, dtype=N.F32))
        e = B.dfpwm_encode(ctx, mono, True).download()
        a = [x[0].copy() for x in B.decode(ctx, bt, d, dtype=N.F32).download()]
        return t, e, a
    t1, e1, a1 = run()
    monkeypatch.setenv("AUKIT_DFPWM_NOSPEC", "1")
    t0, e0, a0 = run()
    monkeypatch.delenv("AUKIT_DFPWM_NOSPEC")
    assert t1 == t0 and e1 == e0
    assert all(np.array_equal(x, y) for x, y in zip(a1, a0))
    for s in range(min(n, 2)):
        assert t1[s] == _ref(oracle, streams[s])   # (Audio:dfpwm runs on the F32 rows of the mono mix here: its bytes are held to the older schedule's above)
