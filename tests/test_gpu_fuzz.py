"""GPU: seeded random sweeps over the fast paths against the CPU oracle — odd sample rates, ragged and tiny streams, full-scale
random samples, plateaus.  Every case goes through the C ABI twice: the f32 tolerance path
(≤ 1e-6 RMS, the bar of SURVEY §8d) and the reference-order path (exact / 1e-13), and the chunk bookkeeping of the stream paths
must equal the oracle's.  Seeds are fixed: the sweep is the same on every run (AUKIT_FUZZ_SEED_OFFSET=k shifts every seed range
by k for soak runs: `tools/fuzz_soak.sh`).
"""
import os

import numpy as np
import pytest

from tests.util import rms

pytestmark = pytest.mark.gpu

RATES = [4000, 6000, 8000, 11025, 12000, 16000, 22050, 24000, 32000, 37800, 44100, 47999, 8001, 44056, 30000]


def _seeds(n):
    off = int(os.environ.get("AUKIT_FUZZ_SEED_OFFSET", "0"))
    return range(off, off + n)


def _maxdiff(got, ref):
    """max |got - ref| where the oracle is a number; NaNs (0 · inf in normalize of an all-zero row, inf - inf) must sit at the same places"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan), int(nan.sum())
    return np.max(np.abs(got[~nan] - ref[~nan]), initial=0)


def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


def _lens(rng, rate, k):
    """k stream lengths (frames): around chunk and tile boundaries, tiny, and a long one"""
    picks = [1, 2, 3, 5, 17, 1023, 1024, 1025, rate - 1, rate, rate + 1, int(rate * 1.5) + int(rng.integers(0, 50)), int(rate * 2.2)]
    return [int(picks[i]) for i in rng.choice(len(picks), k, replace=False)]


@pytest.mark.parametrize("seed", _seeds(24))
def test_fuzz_pcm16_audio_and_stream(ctx, oracle, seed):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    rate = int(RATES[rng.integers(0, len(RATES))])
    ch = int(rng.integers(1, 3))
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    lens = _lens(rng, rate, 6)
    # full-scale random samples (worst case for the interpolators), a few constant runs
    streams = []
    for n in lens:
        x = rng.integers(-32768, 32768, n * ch, dtype=np.int64).astype(np.int16)
        if n > 40:
            x[10 * ch:30 * ch] = x[10 * ch]
        streams.append(x.tobytes())
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, 16, "signed")
    oi = oracle.INTERP[interp]
    # Audio path
    g32 = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32).download()
    g64 = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    for i, s in enumerate(streams):
        if len(s) % (2 * ch):
            continue
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, ch, rate), 48000, oi)
        for c in range(ch):
            assert len(g32[i][c]) == len(ref.data[c]) == len(g64[i][c]), (rate, ch, interp, i)
            assert np.max(np.abs(g64[i][c] - ref.data[c]), initial=0) <= 1e-15, (rate, ch, interp, i, c)
            assert rms(g32[i][c], ref.data[c]) <= 1e-6, (rate, ch, interp, i, c)
    # stream path (whole frames only)
    ok = [s for s in streams if len(s) % (2 * ch) == 0 and len(s)]
    bts = B.Batch.upload(ctx, ok)
    for mono in ([False, True] if ch == 2 else [False]):
        o32, ck32 = B.stream_decode(ctx, bts, desc, interp, mono=mono, dtype=N.F32)
        o64, ck64 = B.stream_decode(ctx, bts, desc, interp, mono=mono, dtype=N.F64)
        a32, a64 = o32.download(), o64.download()
        for i, s in enumerate(ok):
            ref = oracle.stream_pcm(s, 16, oracle.SIGNED, ch, rate, False, mono, oi)
            for ck in (ck32, ck64):
                assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (rate, ch, interp, mono, i)
                assert ck.status[i] == ref.final_status
            for c in range(ref.channels):
                assert np.max(np.abs(a64[i][c] - ref.data[c]), initial=0) <= 1e-13, (rate, ch, interp, mono, i, c)
                assert rms(a32[i][c] / 128, ref.data[c] / 128) <= 1e-6, (rate, ch, interp, mono, i, c)


@pytest.mark.parametrize("seed", _seeds(24))
def test_fuzz_g711_audio_and_stream(ctx, oracle, seed):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    rate = int(RATES[rng.integers(0, len(RATES))])
    ulaw = bool(rng.integers(0, 2))
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    lens = _lens(rng, rate, 6)
    streams = []
    for n in lens:
        x = rng.integers(0, 256, n, dtype=np.uint8)
        if n > 64:
            x[20:50] = x[20]  # a plateau: interpolated values that are exact multiples of 1/64 (and, in the stream, exact integers)
        streams.append(x.tobytes())
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, 1, rate, ulaw=ulaw)
    oi = oracle.INTERP[interp]
    g32 = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32).download()
    g64 = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=N.I8)
    st = out.download()
    for i, s in enumerate(streams):
        ref = oracle.resample(oracle.g711(s, ulaw, 1, rate), 48000, oi)
        assert len(g32[i][0]) == len(ref.data[0])
        assert np.max(np.abs(g64[i][0] - ref.data[0]), initial=0) <= 1e-15, (rate, ulaw, interp, i)
        assert rms(g32[i][0], ref.data[0]) <= 1e-6, (rate, ulaw, interp, i)
        rs = oracle.stream_g711(s, ulaw, 1, rate, False, oi)
        assert ck.nchunks[i] == rs.nchunks and list(ck.lens[i][:rs.nchunks]) == list(rs.chunk_len[:, 0]), (rate, ulaw, interp, i)
        assert np.array_equal(st[i][0], rs.data[0]), (rate, ulaw, interp, i)  # floored outputs: bit-exact


@pytest.mark.parametrize("seed", _seeds(8))
def test_fuzz_dfpwm_paths(ctx, oracle, seed):
    """random DFPWM byte streams of odd lengths through the loader (1-3 channels where the sample count divides), stream.dfpwm and
    the fused stereo → mono → DFPWM transcode, bit-exact"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    lens = [int(v) for v in rng.choice([1, 2, 7, 599, 6000, 6001, 6002, 11999, 12000, 12001, 18003, 30000, 48001], 6, replace=False)]
    streams = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in lens]
    bt = B.Batch.upload(ctx, streams)
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 1, 48000), dtype=N.F64).download()
    for s, g in zip(streams, got):
        assert np.array_equal(g[0], oracle.dfpwm(s, 1, 48000).data[0]), len(s)
    fused = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    for s, f in zip(streams, fused):
        assert f == oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(s, 2, 48000)), True), len(s)
    rate = [48000, 24000, 32000, 44100][seed % 4]
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 1, rate), "linear", dtype=N.F64)
    a = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_dfpwm(s, rate, 1, False, oracle.LINEAR)
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (rate, len(s))
        assert np.max(np.abs(a[i][0] - ref.data[0]), initial=0) <= 1e-13, (rate, len(s))


@pytest.mark.parametrize("seed", _seeds(24))
def test_fuzz_ima_stream(ctx, oracle, seed):
    """stream.adpcm over random block sizes / channel counts / rates, full-scale random PCM (saturating predictors), ragged tails:
    chunk bookkeeping and every floored output equal to the oracle's"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(4000 + seed))
    ch = int(rng.integers(1, 3))
    ba = int(rng.choice([36, 68, 260, 512, 1024, 2048] if ch == 1 else [72, 264, 512, 1024, 2048]))  # 4 ch + whole words per channel
    rate = int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100]))
    interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
    mono = bool(ch == 2 and rng.integers(0, 2))
    spb = (ba - 4 * ch) * 2 // ch
    streams = []
    for nb in (int(rng.integers(1, 4)), int(rng.integers(20, 60)), 1):
        amp = [32767, 3000, 300][int(rng.integers(0, 3))]
        x = rng.integers(-amp, amp + 1, spb * nb * ch, dtype=np.int64).astype(np.int16)
        streams.append(oracle.gen_ima(x, ch, ba, 88))
    streams.append(streams[1][: ba * 5 + int(rng.integers(4 * ch + 1, ba))])  # short final block
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, ch, rate, block_align=ba), interp, mono=mono, dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_adpcm(s, ba, ch, rate, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, (ba, ch, rate, interp, mono, i)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (ba, ch, rate, interp, mono, i)
        for c in range(ref.channels):
            assert np.array_equal(got[i][c], ref.data[c]), (ba, ch, rate, interp, mono, i, c)


@pytest.mark.parametrize("seed", _seeds(24))
@pytest.mark.parametrize("decoder", ["auto", "stream"])
def test_fuzz_flac(ctx, oracle, seed, decoder, monkeypatch):
    """FLAC files of random depth / channels / block size / length (the oracle's encoder picks predictor orders and Rice parameters
    per block) through the loader (lossless), the resampled f32 pipeline and stream.flac"""
    if decoder != "auto":   # (round 6: k_flac_pq takes small batches by default — the one-wave kernel sees the same cases)
        monkeypatch.setenv("AUKIT_FLAC_DECODER", decoder)
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    depth = int(rng.choice([8, 16, 24]))
    ch = int(rng.integers(1, 3))
    bs = int(rng.choice([192, 576, 1000, 1152, 4096, 4608]))
    rate = int(rng.choice([22050, 32000, 44100, 48000]))
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    streams, pcms = [], []
    for n in (int(rng.integers(1, 50)), bs, bs + 1, int(rng.integers(3 * bs, 6 * bs)), rate + 17):
        lim = 1 << (depth - 1)
        t = np.arange(n)[:, None] / rate
        x = 0.6 * lim * np.sin(2 * np.pi * np.array([440.0, 557.0][:ch]) * t) + rng.integers(-lim // 8, lim // 8 + 1, (n, ch))
        x = np.clip(np.round(x), -lim, lim - 1).astype(np.int64)
        pcms.append(x)
        streams.append(oracle.gen_flac(x.ravel(), ch, depth, rate, bs))
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_FLAC)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for x, g in zip(pcms, got):
        for c in range(ch):
            assert np.array_equal(np.round(g[c] * (1 << depth)).astype(np.int64), x[:, c]), (depth, ch, bs, len(x))
    if rate != 48000:
        g32 = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32).download()
        for s, g in zip(streams, g32):
            ref = oracle.resample(oracle.flac(s), 48000, oracle.INTERP[interp])
            for c in range(ch):
                assert rms(g[c], ref.data[c]) <= 1e-6, (depth, ch, bs, rate, interp)
    out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=N.F64)
    a = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_flac(s, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (depth, ch, bs, rate, i)
        for c in range(ref.channels):
            assert np.max(np.abs(a[i][c] - ref.data[c]), initial=0) <= 1e-12, (depth, ch, bs, rate, i, c)


@pytest.mark.parametrize("seed", _seeds(20))
def test_fuzz_msadpcm(ctx, oracle, seed):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(6000 + seed))
    ch = int(rng.integers(1, 3))
    ba = int(rng.choice([64, 256, 1024, 2048])) * ch
    rate = int(rng.choice([8000, 11025, 22050, 44100]))
    interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
    mono = bool(ch == 2 and rng.integers(0, 2))
    spb = (ba - 14) + 2 if ch == 2 else (ba - 7) * 2 + 2
    streams = []
    for nb in (1, int(rng.integers(2, 9)), int(rng.integers(10, 50))):
        amp = [32767, 4000, 200][int(rng.integers(0, 3))]
        x = rng.integers(-amp, amp + 1, spb * nb * ch, dtype=np.int64).astype(np.int16)
        streams.append(oracle.gen_msadpcm(x, ch, ba))
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_MSADPCM, ch, rate, block_align=ba)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.msadpcm(s, ba, ch, rate)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c], equal_nan=True), (ba, ch, rate)  # full-scale noise can drive `delta` to inf / nan, as in Lua
    out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.I8)
    g = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_msadpcm(s, ba, ch, rate, mono, None, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (ba, ch, rate, interp, mono, i)
        for c in range(ref.channels):
            ok = ~np.isnan(ref.data[c])  # where `delta` overflowed the Lua number is nan, which an int8 chunk cannot hold
            assert np.array_equal(g[i][c][ok], ref.data[c][ok]), (ba, ch, rate, interp, mono, i, c)


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_qoa(ctx, oracle, seed):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    ch = int(rng.integers(1, 3))
    rate = int(rng.choice([8000, 22050, 44100, 48000]))
    interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
    mono = bool(ch == 2 and rng.integers(0, 2))
    streams = []
    for n in (int(rng.integers(1, 40)), 20, 5120, 5121, int(rng.integers(5120 * 2, 5120 * 5)), rate + int(rng.integers(0, 300))):
        amp = [32767, 5000][int(rng.integers(0, 2))]
        x = rng.integers(-amp, amp + 1, n * ch, dtype=np.int64).astype(np.int16)
        streams.append(oracle.gen_qoa(x, ch, rate) + b"\0" * (8 if rng.integers(0, 2) else 0))  # with / without the trailing bytes of Q18
    bt = B.Batch.upload(ctx, streams)
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_QOA), dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.qoa(s)
        assert len(g) == ref.channels
        for c in range(ref.channels):
            assert np.array_equal(g[c], ref.data[c]), (ch, rate, len(s))
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA), interp, mono=mono, dtype=N.F64)
    a = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_qoa(s, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (ch, rate, interp, mono, i)
        assert ck.status[i] == ref.final_status
        for c in range(ref.channels):
            assert np.max(np.abs(a[i][c] - ref.data[c]), initial=0) <= 1e-12, (ch, rate, interp, mono, i, c)


@pytest.mark.parametrize("seed", _seeds(12))
def test_fuzz_pcm_formats(ctx, oracle, seed):
    """the generic PCM source: 8/16/24/32-bit signed / unsigned / float, either endianness, 1-3 channels, interleaved or planar,
    through aukit.pcm (exact), :resample (fp64 order) and stream.pcm (rates <= 48 kHz)"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(8000 + seed))
    bits = int(rng.choice([8, 16, 24, 32]))
    dt = ["signed", "unsigned", "float"][int(rng.integers(0, 3 if bits == 32 else 2))]
    be = bool(rng.integers(0, 2))
    ch = int(rng.integers(1, 4))
    rate = int(rng.choice([8000, 12000, 22050, 44100, 48000] + RATES))
    interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
    odt = {"signed": oracle.SIGNED, "unsigned": oracle.UNSIGNED, "float": oracle.FLOAT}[dt]
    streams = []
    for n in (1, 3, int(rng.integers(100, 3000)), rate + int(rng.integers(1, 99))):
        if dt == "float":
            raw = rng.uniform(-1, 1, n * ch).astype(">f4" if be else "<f4").tobytes()
        else:
            raw = bytes(rng.integers(0, 256, n * ch * (bits // 8), dtype=np.uint8))
        streams.append(raw)
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dt, big_endian=be)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    res = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    for s, g, r in zip(streams, got, res):
        ref = oracle.pcm(s, bits, odt, ch, rate, True, be)
        rr = oracle.resample(ref, 48000, oracle.INTERP[interp])
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c]), (bits, dt, be, ch)
            assert len(r[c]) == len(rr.data[c]) and np.max(np.abs(r[c] - rr.data[c]), initial=0) <= 1e-15, (bits, dt, be, ch, rate, interp)
    if interp != "none":
        mono = bool(ch > 1 and rng.integers(0, 2))
        out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F64)
        a = out.download()
        for i, s in enumerate(streams):
            ref = oracle.stream_pcm(s, bits, odt, ch, rate, be, mono, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (bits, dt, be, ch, rate, interp, mono, i)
            for c in range(ref.channels):
                assert np.max(np.abs(a[i][c] - ref.data[c]), initial=0) <= 1e-13, (bits, dt, be, ch, rate, interp, mono, i, c)


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_pcm_formats_f32_one_launch(ctx, oracle, seed):
    """F32 storage on a random format (8 / 16 / 24 / 32-bit, signed / unsigned / float, either byte order, 1-3 channels), rate and interpolation:
    `aukit.pcm(...):resample(48000)` in f32 arithmetic and in fp64 arithmetic (AUKIT_OPT_EXACT_MATH 1) and `aukit.stream.pcm` (with and without
    `mono`) — k_fast_wave_fmt and the paths around it — within 1e-6 RMS of the oracle; float strings now and then hold samples beyond ±1 (the
    flagged reference-order redo)."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(91000 + seed))
    bits = int(rng.choice([8, 16, 24, 32]))
    dt = ["signed", "unsigned", "float"][int(rng.integers(0, 3 if bits == 32 else 2))]
    be = bool(rng.integers(0, 2))
    ch = int(rng.integers(1, 4))
    rate = int(rng.choice([8000, 11025, 22050, 32000, 44100, 48000]))
    new_rate = int(rng.choice([48000, 48000, 44100, 22050]))
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    odt = {"signed": oracle.SIGNED, "unsigned": oracle.UNSIGNED, "float": oracle.FLOAT}[dt]
    wild = dt == "float" and bool(rng.integers(0, 3) == 0)
    streams = []
    for n in (1, 2, int(rng.integers(100, 3000)), rate + int(rng.integers(1, 99)), int(rng.integers(20000, 60000))):
        if dt == "float":
            x = rng.uniform(-1, 1, n * ch)
            if wild and n > 50:
                x[rng.integers(0, n * ch, 5)] *= 2.5
            raw = x.astype(">f4" if be else "<f4").tobytes()
        else:
            raw = bytes(rng.integers(0, 256, n * ch * (bits // 8), dtype=np.uint8))
        streams.append(raw)
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dt, big_endian=be)
    refs = [oracle.pcm(s, bits, odt, ch, rate, True, be) for s in streams]
    rrs = [oracle.resample(r, new_rate, oracle.INTERP[interp]) for r in refs]
    for em in (0, 1):
        ctx.set_option(N.OPT_EXACT_MATH, em)
        try:
            res = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
            name = ctx.last_kernel()[0]
        finally:
            ctx.set_option(N.OPT_EXACT_MATH, 0)
        for r, rr in zip(res, rrs):
            for c in range(ch):
                assert len(r[c]) == len(rr.data[c]), (bits, dt, be, ch, rate, new_rate, interp, em, name)
                if len(r[c]):
                    scale = max(1.0, float(np.max(np.abs(rr.data[c]))))
                    assert np.max(np.abs(r[c] - rr.data[c])) <= 4e-6 * scale, (bits, dt, be, ch, rate, new_rate, interp, em, name, wild)
    mono = bool(ch > 1 and rng.integers(0, 2))
    out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F32)
    name = ctx.last_kernel()[0]
    a = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, bits, odt, ch, rate, be, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (bits, dt, be, ch, rate, interp, mono, i, name)
        for c in range(ref.channels):
            assert len(a[i][c]) == len(ref.data[c])
            if len(ref.data[c]):
                assert np.max(np.abs(a[i][c] - ref.data[c])) <= 1e-3, (bits, dt, be, ch, rate, interp, mono, i, c, name)   # the [-128, 127] scale: 8e-6 of it


@pytest.mark.parametrize("seed", _seeds(8))
def test_fuzz_reverb_f32(ctx, oracle, seed):
    """effects.reverb on F32 audios with random rate / delay / decay / wet / dry and ragged lengths: the one-pass kernel where its state fits a CU,
    the launches of round 2 elsewhere — 1e-6 RMS from the oracle either way."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(92000 + seed))
    rate = int(rng.choice([16000, 22050, 32000, 44100, 48000, 48000, 64000]))
    S = int(np.floor(0.08927 * rate))
    delay = float(rng.choice([40.0, 60.0, 100.0, 100.0, 133.7, 180.0]))
    decay = float(rng.uniform(0.32, 0.6))
    wet, dry = float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.0, 0.5))
    lens = [S + 1 + int(rng.integers(0, 3)), S + int(rng.integers(100, 9000)), rate + int(rng.integers(0, 5000))]
    ch = int(rng.integers(1, 3))
    a = [[rng.uniform(-1, 1, n).astype(np.float32).astype(np.float64) * float(rng.uniform(0.1, 0.6)) for _ in range(ch)] for n in lens]
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F32)
    B.effect(ctx, ab, "reverb", delay, decay, wet, dry)
    name = ctx.last_kernel()[0]
    got = ab.download()
    for s in range(len(a)):
        ref = oracle.fx_reverb(oracle.Audio(a[s], rate), delay, decay, wet, dry)
        for c in range(ch):
            assert rms(got[s][c], ref.data[c]) <= 1e-6, (rate, delay, decay, wet, dry, s, c, name)
            assert np.max(np.abs(got[s][c] - ref.data[c])) <= 4e-6, (rate, delay, decay, wet, dry, s, c, name)


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_effects_and_audio_methods(ctx, oracle, seed):
    """random audio (1-3 channels, ragged lengths, values over the full [-1, 1] range incl. exact ±1 and 0) through a random effect
    with random parameters, then :mono / :mix / :resample — F64 storage against the oracle (maps exact, scans 1e-11)"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    ch = int(rng.integers(1, 4))
    rate = int(rng.choice([8000, 22050, 44100, 48000]))
    lens = [int(rng.integers(1, 40)), int(rng.integers(900, 1200)), int(rng.integers(rate // 2, rate * 2))]
    a = []
    for n in lens:
        chans = []
        for _ in range(ch):
            x = rng.uniform(-1, 1, n)
            x[rng.integers(0, n, max(1, n // 50))] = rng.choice([-1.0, 0.0, 1.0], max(1, n // 50))
            chans.append(x)
        a.append(chans)
    choices = [
        ("amplify", (float(rng.uniform(0, 3)),), lambda O, au, p: O.fx_amplify(au, *p), 0.0),
        ("invert", (), lambda O, au, p: O.fx_invert(au), 0.0),
        ("normalize", (float(rng.uniform(0.1, 1)), float(rng.integers(0, 2))), lambda O, au, p: O.fx_normalize(au, p[0], bool(p[1])), 0.0),
        ("center", (), lambda O, au, p: O.fx_center(au), 1e-12),
        ("delay", (float(rng.uniform(0, 0.02)), float(rng.uniform(0, 1))), lambda O, au, p: O.fx_delay(au, *p), 0.0),
        ("echo", (float(rng.uniform(0.0003, 0.02)), float(rng.uniform(0, 0.95))), lambda O, au, p: O.fx_echo(au, *p), 0.0),
        ("lowpass", (float(rng.uniform(50, rate / 2)),), lambda O, au, p: O.fx_lowpass(au, *p), 1e-11),
        ("highpass", (float(rng.uniform(10, rate / 4)),), lambda O, au, p: O.fx_highpass(au, *p), 1e-11),
    ]
    name, params, ref_fn, tol = choices[int(rng.integers(0, len(choices)))]
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F64)
    B.effect(ctx, ab, name, *params)
    got = ab.download()
    refs = [ref_fn(oracle, oracle.Audio(x, rate), params) for x in a]
    for s in range(len(a)):
        for c in range(ch):
            assert _maxdiff(got[s][c], refs[s].data[c]) <= tol, (name, params, ch, rate, s, c)
    m = B.mono(ctx, ab).download()
    for s in range(len(a)):
        assert _maxdiff(m[s][0], oracle.mono(refs[s]).data[0]) <= max(tol, 1e-15), (name, s)
    if rate != 48000:
        interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
        r = B.resample(ctx, ab, 48000, interp).download()
        for s in range(len(a)):
            rr = oracle.resample(refs[s], 48000, oracle.INTERP[interp])
            for c in range(ch):
                assert len(r[s][c]) == len(rr.data[c]) and _maxdiff(r[s][c], rr.data[c]) <= max(4 * tol, 1e-15), (name, interp, s, c)


@pytest.mark.parametrize("seed", _seeds(6))
def test_fuzz_mdfpwm(ctx, oracle, seed):
    """MDFPWM files with random payloads (1-9 L/R block pairs), random metadata lengths and a length field at, below and above the
    payload size: aukit.mdfpwm (trim at length * 8 samples, :1444) and stream.mdfpwm (Q12) against the oracle"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9500 + seed))
    files = []
    for pairs in (1, int(rng.integers(2, 5)), int(rng.integers(5, 10))):
        l = bytes(rng.integers(0, 256, 6000 * pairs, dtype=np.uint8))
        r = bytes(rng.integers(0, 256, 6000 * pairs, dtype=np.uint8))
        md = bytearray(oracle.gen_mdfpwm(l, r, bytes(rng.integers(65, 91, int(rng.integers(0, 40)), dtype=np.uint8)), b"t" * int(rng.integers(0, 20)), b""))
        if rng.integers(0, 2):  # shrink the length field: the loader trims, the stream stops early (Q12)
            newlen = int(rng.integers(1, 12000 * pairs))
            md[7:11] = newlen.to_bytes(4, "little")
        files.append(bytes(md))
    bt = B.Batch.upload(ctx, files)
    try:
        got = B.decode(ctx, bt, B.make_desc(N.CODEC_MDFPWM), dtype=N.F64).download()
        loader_err = None
    except N.AukitError as e:
        loader_err = str(e)
    for i, f in enumerate(files):
        try:
            ref = oracle.mdfpwm(f)
        except Exception as e:  # an odd trimmed length is an error in the reference ("uneven amount of data per channel")
            assert loader_err is not None, (i, e)
            continue
        if loader_err is None:
            for c in range(2):
                assert np.array_equal(got[i][c], ref.data[c]), (i, c)
    for mono in (False, True):
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_MDFPWM), "linear", mono=mono, dtype=N.I8)
        g = out.download()
        for i, f in enumerate(files):
            o = oracle.stream_mdfpwm(f, mono)
            assert ck.nchunks[i] == o.nchunks and ck.status[i] == o.final_status, (i, mono)
            for c in range(o.channels):
                assert np.array_equal(g[i][c], o.data[c]), (i, mono, c)


@pytest.mark.parametrize("seed", _seeds(10))
def test_fuzz_multichannel_streams(ctx, oracle, seed):
    """stream.g711 and stream.dfpwm with 2-3 channels, with and without `mono` (Q11, Q13: the reference-order kernels)"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9700 + seed))
    ch = int(rng.integers(2, 4))
    mono = bool(rng.integers(0, 2))
    interp = ["none", "linear", "cubic"][int(rng.integers(0, 3))]
    rate = int(rng.choice([8000, 11025, 16000, 24000, 48000]))
    ulaw = bool(rng.integers(0, 2))
    streams = [bytes(rng.integers(0, 256, n * ch, dtype=np.uint8)) for n in (1, 7, rate // 2 + 3, rate + 11, 2 * rate + 1)]
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_G711, ch, rate, ulaw=ulaw), interp, mono=mono, dtype=N.I8)
    g = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_g711(s, ulaw, ch, rate, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (ch, mono, interp, rate, i)
        for c in range(ref.channels):
            assert np.array_equal(g[i][c], ref.data[c]), (ch, mono, interp, rate, i, c)
    dstreams = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (3 * ch, 6000 * ch, 6000 * ch + ch, 13000 * ch)]
    bd = B.Batch.upload(ctx, dstreams)
    drate = int(rng.choice([48000, 24000, 44100]))
    out, ck = B.stream_decode(ctx, bd, B.make_desc(N.CODEC_DFPWM, ch, drate), "linear" if interp == "none" else interp, mono=mono, dtype=N.F64)
    a = out.download()
    for i, s in enumerate(dstreams):
        ref = oracle.stream_dfpwm(s, drate, ch, mono, oracle.INTERP["linear" if interp == "none" else interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), (ch, mono, drate, i)
        for c in range(ref.channels):
            assert np.max(np.abs(a[i][c] - ref.data[c]), initial=0) <= 1e-13, (ch, mono, drate, i, c)


@pytest.mark.parametrize("seed", _seeds(12))
def test_fuzz_mix_encode_dfpwm(ctx, oracle, seed):
    """Audio:mix over 2-4 audios of different lengths and channel counts (zero padding, clamp of sum * amplifier), Audio:pcm in every
    depth / type / layout and Audio:dfpwm (interleaved or channel after channel) on the result — exact"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9900 + seed))
    k = int(rng.integers(2, 5))
    nstreams = 3
    auds, refs = [], [[] for _ in range(nstreams)]
    for _ in range(k):
        ch = int(rng.integers(1, 4))
        per_stream = []
        for s in range(nstreams):
            n = int(rng.integers(1, 3000))
            per_stream.append([rng.uniform(-1, 1, n) for _ in range(ch)])
            refs[s].append(oracle.Audio(per_stream[-1], 48000))
        auds.append(B.AudioBatch.upload(ctx, per_stream, 48000, dtype=N.F64))
    amp = float(rng.uniform(0.1, 1.5))
    m = B.mix(ctx, auds, amp)
    got = m.download()
    mixed = [oracle.mix(refs[s], amp) for s in range(nstreams)]
    for s in range(nstreams):
        assert len(got[s]) == mixed[s].channels
        for c in range(mixed[s].channels):
            assert np.array_equal(got[s][c], mixed[s].data[c]), (k, amp, s, c)
    bits = int(rng.choice([8, 16, 24, 32]))
    dt = ["signed", "unsigned"][int(rng.integers(0, 2))]
    inter = bool(rng.integers(0, 2))
    enc = B.encode_pcm(ctx, m, bits, dt, inter).download()
    for s in range(nstreams):
        ref = oracle.encode_pcm(mixed[s], bits, {"signed": oracle.SIGNED, "unsigned": oracle.UNSIGNED}[dt], inter)
        assert np.array_equal(np.asarray(enc[s][0]), ref), (bits, dt, inter, s)
    df = B.dfpwm_encode(ctx, m, inter).download()
    for s in range(nstreams):
        assert df[s] == oracle.audio_dfpwm(mixed[s], inter), (inter, s)


@pytest.mark.parametrize("seed", _seeds(48))
@pytest.mark.parametrize("decoder", ["auto", "stream"])
def test_fuzz_flac_corrupted(ctx, oracle, seed, decoder, monkeypatch):
    """FLAC files with a few random bytes overwritten or cut short after the metadata: whatever decodeFLAC does with them — decode
    garbage, lose sync, raise — the stream (errors swallowed, it just ends) must deliver the same chunks as the oracle, and the
    loader must raise exactly when the oracle does, with the same samples otherwise"""
    if decoder != "auto":   # (round 6: k_flac_pq takes small batches by default — the one-wave kernel sees the same cases)
        monkeypatch.setenv("AUKIT_FLAC_DECODER", decoder)
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9950 + seed))
    ch = int(rng.integers(1, 3))
    bs = int(rng.choice([192, 576, 1152, 4096]))
    n = int(rng.integers(2 * bs, 6 * bs))
    t = np.arange(n)[:, None] / 44100
    x = np.clip(np.round(9000 * np.sin(2 * np.pi * np.array([440.0, 557.0][:ch]) * t) + rng.integers(-900, 900, (n, ch))), -32768, 32767).astype(np.int64)
    good = oracle.gen_flac(x.ravel(), ch, 16, 44100, bs)
    files = []
    for _ in range(4):
        b = bytearray(good)
        kind = int(rng.integers(0, 3))
        if kind == 0:    # overwrite 1-3 bytes somewhere in the frames
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(60, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:  # cut the file short
            b = b[: int(rng.integers(60, len(b)))]
        else:            # flip one bit
            p = int(rng.integers(60, len(b)))
            b[p] ^= 1 << int(rng.integers(0, 8))
        files.append(bytes(b))
    desc = B.make_desc(N.CODEC_FLAC)
    for f in files:  # one file per call: a loader error concerns the whole batch
        bt = B.Batch.upload(ctx, [f])
        try:
            ref = oracle.flac(f)
        except Exception:
            ref = None
        try:
            got = B.decode(ctx, bt, desc, dtype=N.F64).download()[0]
        except N.AukitError:
            got = None
        assert (ref is None) == (got is None), (ch, bs, len(f))
        if ref is not None:
            for c in range(ref.channels):
                assert np.array_equal(got[c], ref.data[c], equal_nan=True), (ch, bs, c)
        rs = oracle.stream_flac(f, oracle.LINEAR)
        out, ck = B.stream_decode(ctx, bt, desc, "linear", dtype=N.F64)
        a = out.download()[0]
        assert ck.nchunks[0] == rs.nchunks and list(ck.lens[0][:rs.nchunks]) == list(rs.chunk_len[:, 0]), (ch, bs, len(f))
        for c in range(rs.channels):
            # garbage residuals can run the doubles to ±inf; inf - inf in the 2-tap low-pass is NaN, and clamp passes NaN through
            nan = np.isnan(rs.data[c])
            assert np.array_equal(np.isnan(a[c]), nan), (ch, bs, c, int(nan.sum()))
            assert np.max(np.abs(a[c][~nan] - rs.data[c][~nan]), initial=0) <= 1e-12, (ch, bs, c)


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_ima_random_bytes(ctx, oracle, seed):
    """stream.adpcm / aukit.wav's IMA path on RANDOM bytes: header step indices above 88 (the stream uses them unmasked and dies with a
    Lua error in the middle of a call, aukit.wav masks mono ones with 0x0F), saturating predictors, partial blocks"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9990 + seed))
    ch = int(rng.integers(1, 3))
    ba = int(rng.choice([36, 260, 512] if ch == 1 else [72, 264, 512]))
    rate = int(rng.choice([11025, 22050, 44100]))
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    mono = bool(ch == 2 and rng.integers(0, 2))
    streams = []
    for nb in (1, 3, int(rng.integers(5, 40))):
        raw = bytearray(rng.integers(0, 256, ba * nb + int(rng.integers(0, ba)), dtype=np.uint8).tobytes())
        if rng.integers(0, 2):  # keep most header indices legal so that some streams survive for a while
            for b in range(0, len(raw) - 4 * ch, ba):
                for c in range(ch):
                    raw[b + 4 * c + 2] %= 89
        streams.append(bytes(raw))
    desc = B.make_desc(N.CODEC_ADPCM_WAV, ch, rate, block_align=ba)
    for s in streams:  # one stream per call: statuses are per stream, loader errors per batch
        bt = B.Batch.upload(ctx, [s])
        ref = oracle.stream_adpcm(s, ba, ch, rate, mono, oracle.INTERP[interp])
        out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.I8)
        g = out.download()[0]
        assert ck.nchunks[0] == ref.nchunks and ck.status[0] == ref.final_status, (ba, ch, rate, len(s))
        assert list(ck.lens[0][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        for c in range(ref.channels):
            assert np.array_equal(g[c], ref.data[c]), (ba, ch, rate, c)
        try:
            ra = oracle.wav_adpcm(s, ba, ch, rate)
        except Exception:
            ra = None
        try:
            ga = B.decode(ctx, bt, desc, dtype=N.F64).download()[0]
        except N.AukitError:
            ga = None
        assert (ra is None) == (ga is None), (ba, ch, len(s))
        if ra is not None:
            for c in range(ch):
                assert np.array_equal(ga[c], ra.data[c]), (ba, ch, c)


@pytest.mark.parametrize("seed", _seeds(24))
def test_fuzz_qoa_corrupted(ctx, oracle, seed):
    """QOA files with overwritten bytes (slices, LMS state, now and then a frame header) or cut short: aukit.qoa raises exactly when
    the oracle does and decodes the same samples otherwise; stream.qoa delivers the same chunks and ends with the same status"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(9960 + seed))
    ch = int(rng.integers(1, 3))
    n = int(rng.integers(5120, 5120 * 4))
    x = rng.integers(-20000, 20000, n * ch).astype(np.int16)
    good = oracle.gen_qoa(x, ch, 44100) + b"\0" * 8
    files = []
    for _ in range(4):
        b = bytearray(good)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(16, len(b)))] = int(rng.integers(0, 256))  # payload of the first frame onwards
        elif kind == 1:
            b = b[: int(rng.integers(9, len(b)))]
        elif kind == 2:
            p = int(rng.integers(8, len(b)))
            b[p] ^= 1 << int(rng.integers(0, 8))
        else:
            b[int(rng.integers(8, 16))] = int(rng.integers(0, 256))  # the first frame header: channels / rate / samples / size
        files.append(bytes(b))
    desc = B.make_desc(N.CODEC_QOA)
    for f in files:
        bt = B.Batch.upload(ctx, [f])
        try:
            ref = oracle.qoa(f)
        except Exception:
            ref = None
        try:
            got = B.decode(ctx, bt, desc, dtype=N.F64).download()[0]
        except N.AukitError:
            got = None
        assert (ref is None) == (got is None), (ch, len(f))
        if ref is not None:
            assert len(got) == ref.channels
            for c in range(ref.channels):
                assert np.array_equal(got[c], ref.data[c]), (ch, c)
        try:
            rs = oracle.stream_qoa(f, False, oracle.LINEAR)
        except Exception:
            rs = None
        try:
            out, ck = B.stream_decode(ctx, bt, desc, "linear", dtype=N.F64)
            a = out.download()[0]
        except N.AukitError:
            a = None
        assert (rs is None) == (a is None), (ch, len(f))
        if rs is not None:
            assert ck.nchunks[0] == rs.nchunks and ck.status[0] == rs.final_status and list(ck.lens[0][:rs.nchunks]) == list(rs.chunk_len[:, 0]), (ch, len(f))
            for c in range(rs.channels):
                assert np.max(np.abs(a[c] - rs.data[c]), initial=0) <= 1e-12, (ch, c)


def test_degenerate_batches_do_not_crash(ctx, oracle):
    """zero-length streams, one-byte streams and batches of nothing but those through every loader and stream factory: an
    error from the library (the reference raises on most of them) or an empty result, never a crash or a hang"""
    B, N = _B(), _N()
    descs = [B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), B.make_desc(N.CODEC_PCM, 2, 8000, 8, "unsigned"), B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True),
             B.make_desc(N.CODEC_DFPWM, 1, 48000), B.make_desc(N.CODEC_DFPWM, 2, 48000), B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512),
             B.make_desc(N.CODEC_MSADPCM, 1, 44100, block_align=256), B.make_desc(N.CODEC_QOA), B.make_desc(N.CODEC_FLAC), B.make_desc(N.CODEC_MDFPWM)]
    batches = [[b""], [b"", b""], [b"\\x00"], [b"", b"\\x01\\x02", b""], [b"\\x00" * 3]]
    for d in descs:
        for streams in batches:
            bt = B.Batch.upload(ctx, streams)
            for fn in (lambda: B.decode(ctx, bt, d, dtype=N.F64).download(),
                       lambda: B.decode_resample(ctx, bt, d, 48000, "cubic", dtype=N.F32).download(),
                       lambda: B.stream_decode(ctx, bt, d, "linear", dtype=(N.I8 if d.codec in (N.CODEC_G711, N.CODEC_ADPCM_WAV, N.CODEC_MSADPCM, N.CODEC_MDFPWM) else N.F64))[0].download(),
                       lambda: B.dfpwm_transcode_mono(ctx, bt, 2).download() if d.codec == N.CODEC_DFPWM else None):
                try:
                    fn()
                except N.AukitError:
                    pass
    # and the library is still healthy afterwards
    s = np.arange(-500, 500, dtype=np.int16).tobytes()
    got = B.decode(ctx, B.Batch.upload(ctx, [s]), descs[0], dtype=N.F64).download()[0][0]
    assert np.array_equal(got, oracle.pcm(s, 16, oracle.SIGNED, 1, 44100).data[0])


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_adpcm_loaders(ctx, oracle, seed):
    """aukit.adpcm on random bytes with random channel count / nibble order / planar layout / start predictor and step index
    (aukit.lua:1183-1274), and aukit.wav's IMA blocks (aukit.lua:1509-1548) at random block sizes incl. a partial last block"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(12000 + seed))
    ch = int(rng.integers(1, 4))
    top_first, interleaved = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    pred = [int(rng.integers(-32768, 32768)) for _ in range(ch)]
    idx = [int(rng.integers(0, 89)) for _ in range(ch)]
    streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (1, 2, ch, ch * int(rng.integers(2, 700)), 3 * ch * int(rng.integers(100, 900)))]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM, ch, 22050, interleaved=interleaved, top_first=top_first, predictor=pred, step_index=idx)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.adpcm(s, ch, 22050, top_first, interleaved, pred, idx)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c]), (ch, top_first, interleaved, len(s), c)
    wch = int(rng.integers(1, 3))
    ba = 4 * wch * int(rng.integers(2, 260))
    spb = (ba - 4 * wch) * 2 // wch
    files = []
    for nb in (1, int(rng.integers(2, 9))):
        x = rng.integers(-20000, 20000, (spb * nb, wch)).astype(np.int16)
        f = oracle.gen_ima(x.ravel(), wch, ba, int(rng.integers(0, 16)))
        files.append(f)
    if wch == 1 and ba > 9:
        files.append(files[-1][: len(files[-1]) - int(rng.integers(1, ba - 8))])
    bt = B.Batch.upload(ctx, files)
    desc = B.make_desc(N.CODEC_ADPCM_WAV, wch, 22050, block_align=ba)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    interp = ["linear", "cubic"][int(rng.integers(0, 2))]
    res = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    for f, g, r in zip(files, got, res):
        ref = oracle.wav_adpcm(f, ba, wch, 22050)
        rr = oracle.resample(ref, 48000, oracle.INTERP[interp])
        for c in range(wch):
            assert np.array_equal(g[c], ref.data[c]), (wch, ba, len(f), c)
            assert len(r[c]) == len(rr.data[c]) and np.max(np.abs(r[c] - rr.data[c]), initial=0) <= 1e-15, (wch, ba, interp, c)


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_encode_pcm_sinc_speed(ctx, oracle, seed):
    """Audio:pcm (aukit.lua:901) at every bit depth / type / layout on samples over the whole [-1, 1] range incl. exact ±1 and 0,
    sinc resampling (aukit.lua:267-281) between random rates, effects.speed (aukit.lua:3376) with a random multiplier"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(13000 + seed))
    ch = int(rng.integers(1, 4))
    rate = int(rng.choice(RATES))
    a = []
    for n in (1, 2, int(rng.integers(3, 40)), int(rng.integers(500, 4000))):
        rows = [rng.uniform(-1, 1, n) for _ in range(ch)]
        for r in rows:
            r[rng.integers(0, n)] = [1.0, -1.0, 0.0][int(rng.integers(0, 3))]
        a.append(rows)
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F64)
    bits = int(rng.choice([8, 16, 24, 32]))
    dt = ["signed", "unsigned", "float"][int(rng.integers(0, 3 if bits == 32 else 2))]
    inter = bool(rng.integers(0, 2))
    got = B.encode_pcm(ctx, ab, bits, dt, inter).download()
    for s in range(len(a)):
        ref = oracle.encode_pcm(oracle.Audio(a[s], rate), bits, oracle.DTYPE[dt], inter)
        assert np.array_equal(got[s][0], ref), (bits, dt, inter, ch, s)
    new_rate = int(rng.choice([48000, 16000, 44100, 8000, 96000]))
    if new_rate != rate:
        r = B.resample(ctx, ab, new_rate, "sinc").download()
        for s in range(len(a)):
            rr = oracle.resample(oracle.Audio(a[s], rate), new_rate, oracle.SINC)
            for c in range(ch):
                assert len(r[s][c]) == len(rr.data[c]) and _maxdiff(r[s][c], rr.data[c]) <= 1e-11, (rate, new_rate, s, c)
    mult = float(rng.choice([0.5, 0.75, 1.25, 2.0, float(rng.uniform(0.3, 3.0))]))
    if mult != 1.0:
        di = int(rng.integers(1, 3))  # aukit.defaultInterpolation: linear / cubic
        B.effect(ctx, ab, "speed", mult, di)
        sp = ab.download()
        for s in range(len(a)):
            ref = oracle.fx_speed(oracle.Audio(a[s], rate), mult, [None, oracle.LINEAR, oracle.CUBIC][di])
            for c in range(ch):
                assert len(sp[s][c]) == len(ref.data[c]) and _maxdiff(sp[s][c], ref.data[c]) <= 1e-15, (mult, di, s, c)


@pytest.mark.parametrize("seed", _seeds(10))
def test_fuzz_dfpwm_parallel_encoder(ctx, oracle, seed, monkeypatch):
    """Audio:dfpwm (aukit.lua:1005) on small batches of long streams — the chunk-parallel exact encoder — over random signal
    characters (tones with noise, bursts, rails, silence with clicks, slow ramps, full-scale noise), lengths just above the
    65 536-sample threshold up to a few hundred thousand, 1-3 channels, random chunk counts: bytes equal to the oracle's."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(14000 + seed))
    monkeypatch.setenv("AUKIT_DFPWM_ENC_CHUNKS", str(int(rng.choice([64, 3, 17, 128, 251]))))
    ch = int(rng.integers(1, 4))
    nstreams = int(rng.integers(1, 4))
    a = []
    for _ in range(nstreams):
        n = int(rng.choice([65536, 65537, 70001, int(rng.integers(66000, 300000))]))
        rows = []
        for _ in range(ch):
            t = np.arange(n) / 48000
            kind = int(rng.integers(0, 6))
            if kind == 0:
                x = rng.uniform(0.05, 0.9) * np.sin(2 * np.pi * rng.uniform(30, 9000) * t) + rng.uniform(-1, 1, n) * rng.uniform(0, 0.1)
            elif kind == 1:
                x = np.where(rng.uniform(0, 1, n) < 0.001, rng.uniform(-1, 1, n), 0.0)  # silence with clicks
            elif kind == 2:
                x = np.where((np.arange(n) // int(rng.integers(2, 5000))) % 2 == 0, 1.0, -1.0) * rng.choice([1.0, 0.5])
            elif kind == 3:
                x = np.linspace(-1, 1, n) * rng.choice([1.0, -1.0])
            elif kind == 4:
                env = (np.sin(2 * np.pi * rng.uniform(0.5, 5) * t) > 0.3).astype(np.float64)
                x = env * rng.uniform(-1, 1, n) * rng.uniform(0.1, 1.0)  # noise bursts between silences
            else:
                x = rng.uniform(-1, 1, n) * rng.uniform(0.3, 1.0)
            rows.append(np.clip(x, -1, 1))
        a.append(rows)
    ab = B.AudioBatch.upload(ctx, a, 48000, dtype=N.F64)
    inter = bool(rng.integers(0, 2))
    got = B.dfpwm_encode(ctx, ab, inter).download()
    name = ctx.last_kernel()[0]
    # noise-like chunks can leave more candidate states than the tables hold: those batches take the one-lane-per-stream encoder
    assert name in ("k_dfpwm_quantize+k_dfx_chunks<rows>", "k_dfpwm_quantize+k_dfe_*", "k_dfpwm_quantize+k_dfpwm_encode_i8"), name
    print("encoder:", name)
    for s in range(nstreams):
        assert got[s] == oracle.audio_dfpwm(oracle.Audio(a[s], 48000), inter), (ch, nstreams, s, inter)


def _fuzz_signal(rng, n):
    """one channel of one of the characters the DFPWM encoder behaves differently on: tone + noise, silence with clicks, rail-to-rail squares, slow ramps,
    noise bursts between digital silence, full-scale noise, signal behind / around stretches of digital silence"""
    t = np.arange(n) / 48000
    kind = int(rng.integers(0, 8))
    if kind == 0:
        x = rng.uniform(0.05, 0.9) * np.sin(2 * np.pi * rng.uniform(30, 9000) * t) + rng.uniform(-1, 1, n) * rng.uniform(0, 0.1)
    elif kind == 1:
        x = np.where(rng.uniform(0, 1, n) < 0.001, rng.uniform(-1, 1, n), 0.0)
    elif kind == 2:
        x = np.where((np.arange(n) // int(rng.integers(2, 5000))) % 2 == 0, 1.0, -1.0) * rng.choice([1.0, 0.5])
    elif kind == 3:
        x = np.linspace(-1, 1, n) * rng.choice([1.0, -1.0])
    elif kind == 4:
        x = (np.sin(2 * np.pi * rng.uniform(0.5, 5) * t) > 0.3).astype(np.float64) * rng.uniform(-1, 1, n) * rng.uniform(0.1, 1.0)
    elif kind == 5:
        x = rng.uniform(-1, 1, n) * rng.uniform(0.3, 1.0)
    else:
        x = 0.5 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.25, 0.25, n)   # the config signal ...
        for _ in range(int(rng.integers(1, 4))):                               # ... with stretches of digital silence
            a = int(rng.integers(0, n)); x[a:a + int(rng.integers(100, n // 3 + 101))] = 0
        if kind == 7:
            x[: int(rng.integers(1, n // 2))] = 0
    return np.clip(x, -1, 1)


@pytest.mark.parametrize("seed", _seeds(10))
def test_fuzz_dfpwm_speculation(ctx, oracle, seed, monkeypatch):
    """the chunk-speculative engine (dfpwm_spec.hip) with the probe and the few-short-streams rule switched off — every batch is speculated on, whatever
    it holds — under random
    schedules (warm-up length, chunks per stream, checkpoint spacing, rounds): Audio:dfpwm on 1 - 2 channels and the stereo -> mono -> DFPWM transcode of
    what it made, bytes equal to the oracle's"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(15000 + seed))
    env = {"AUKIT_DFX_FEW": "0", "AUKIT_DFX_NOPROBE": "1", "AUKIT_DFX_WE": str(int(rng.choice([64, 128, 384, 640, 960]))), "AUKIT_DFX_G": str(int(rng.choice([1, 2, 4]))),
           "AUKIT_DFX_ROUNDS": str(int(rng.integers(1, 7))), "AUKIT_DFX_MIN_BPC": str(int(rng.integers(1, 5)))}
    if rng.integers(0, 2):
        env["AUKIT_DFX_CHUNKS"] = str(int(rng.choice([2, 5, 13, 40, 200, 1000])))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nstreams = int(rng.integers(1, 6))
    if seed % 3 == 2:   # (a third of the cases with the probe: beside the first pass or in front of it; whatever it decides, the oracle's bytes)
        monkeypatch.delenv("AUKIT_DFX_NOPROBE")
        monkeypatch.setenv("AUKIT_DFX_PROBE_ASIDE", str(seed // 3 % 2))
    ch = int(rng.integers(1, 3))
    a = [[_fuzz_signal(rng, n) for _ in range(ch)] for n in (int(rng.choice([12000, 48001, 70003, int(rng.integers(20000, 200000))])) for _ in range(nstreams))]
    inter = bool(rng.integers(0, 2))
    enc = B.dfpwm_encode(ctx, B.AudioBatch.upload(ctx, a, 48000, dtype=N.F64), inter).download()
    for s in range(nstreams):
        assert enc[s] == oracle.audio_dfpwm(oracle.Audio(a[s], 48000), inter), ("encode", env, ch, nstreams, s, inter)
    # the transcode of stereo DFPWM made from two such channels (byte counts whose samples divide by two: always)
    st = [oracle.audio_dfpwm(oracle.Audio([_fuzz_signal(rng, n), _fuzz_signal(rng, n)], 48000), True) for n in (int(rng.integers(6000, 90000)) for _ in range(nstreams))]
    got = B.dfpwm_transcode_mono(ctx, B.Batch.upload(ctx, st), 2).download()
    for s in range(nstreams):
        assert got[s] == oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(st[s], 2, 48000)), True), ("transcode", env, s, len(st[s]))


@pytest.mark.parametrize("seed", _seeds(16))
def test_fuzz_deferred_resample_into_the_filter(ctx, oracle, seed, monkeypatch):
    """`loader(...):resample(48000, interp)` then `effects.lowpass / highpass` [then `Audio:mono`] with F32 storage: ONE pass over the decoder's integer rows
    (aukit.lua:648-680, :3586-3618, :682-687) — k_rsp (rs_periodic.hip) where the shape is its (int16 rows, cubic, 44.1 / 22.05 kHz), k_rs_onepole
    (flac_tail.hip) else and under AUKIT_RS_GENERIC=1.  Random loader, lengths around the tile sizes, cut-off, forced runs of tiles: the oracle within 1e-6
    RMS and 1e-6 at the worst sample, the two kernels within a few f32 ulps of each other."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(15000 + seed))
    kind = ["ima", "msadpcm", "qoa", "flac", "flac"][int(rng.integers(0, 5))]
    rate = int(rng.choice([44100, 22050])) if kind != "msadpcm" else 44100
    ch = 1 if kind == "ima" else int(rng.integers(1, 3))
    lens = [int(rng.choice([1, 7, 319, 320, 321, 639, 640, 1281, 2048, int(rng.integers(3000, 30000)), int(rng.integers(30000, 140000))])) for _ in range(4)]

    def sig(n, k):
        t = np.arange(n) / rate
        x = rng.uniform(0.1, 0.9) * np.sin(2 * np.pi * rng.uniform(20, 9000) * t + k) + rng.uniform(-1, 1, n) * rng.uniform(0, 0.1)
        return np.clip(x * 32767, -32768, 32767).astype(np.int16)

    streams = []
    for i, n in enumerate(lens):
        x = np.stack([sig(n, i + c) for c in range(ch)], 1)
        if kind == "ima":
            nb = max(-(-n // 1016), 1)
            streams.append(oracle.gen_ima(sig(1016 * nb, i), 1, 512, int(rng.integers(0, 89))))
        elif kind == "msadpcm":
            spb = (1024 - 7 * ch) * 2 // ch + 2
            nb = max(-(-n // spb), 1)
            streams.append(oracle.gen_msadpcm(np.stack([sig(spb * nb, i + c) for c in range(ch)], 1).ravel(), ch, 1024))
        elif kind == "qoa":
            streams.append(oracle.gen_qoa(x.ravel(), ch, rate) + b"\0" * 8)
        else:
            streams.append(oracle.gen_flac(x.astype(np.int64).ravel(), ch, 16, rate, int(rng.choice([4096, 1152, 2304, 700]))))
    if kind == "ima":
        desc, load = B.make_desc(N.CODEC_ADPCM_WAV, 1, rate, block_align=512), (lambda d: oracle.wav_adpcm(d, 512, 1, rate))
    elif kind == "msadpcm":
        desc, load = B.make_desc(N.CODEC_MSADPCM, ch, rate, block_align=1024), (lambda d: oracle.msadpcm(d, 1024, ch, rate))
    elif kind == "qoa":
        desc, load = B.make_desc(N.CODEC_QOA, ch, rate), oracle.qoa
    else:
        desc, load = B.make_desc(N.CODEC_FLAC), oracle.flac
    interp = "cubic" if rng.integers(0, 4) else "linear"
    which = ["lowpass", "highpass"][int(rng.integers(0, 2))]
    freq = float(rng.choice([20.0, 300.0, 3000.0, 11025.0, float(rng.uniform(10, 20000))]))
    mono = ch == 2 and bool(rng.integers(0, 2))
    segs = int(rng.choice([0, 0, 2, 5]))
    bt = B.Batch.upload(ctx, streams)
    got = {}
    for kern in ("default", "generic"):
        if kern == "generic":
            monkeypatch.setenv("AUKIT_RS_GENERIC", "1")
        if segs:
            monkeypatch.setenv("AUKIT_RS_SEGS", str(segs))
        a = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32)
        B.effect(ctx, a, which, freq)
        got[kern] = (B.mono(ctx, a) if mono else a).download()
    monkeypatch.delenv("AUKIT_RS_GENERIC")
    for i, s in enumerate(streams):
        ref = oracle.resample(load(s), 48000, oracle.INTERP[interp])
        ref = (oracle.fx_lowpass if which == "lowpass" else oracle.fx_highpass)(ref, freq)
        if mono:
            ref = oracle.mono(ref)
        for c in range(len(ref.data)):
            g, h = got["default"][i][c], got["generic"][i][c]
            assert len(g) == len(ref.data[c]) == len(h), (kind, i, c)
            if len(g):
                # (a forced run of tiles warms up over the tiles before it: 1e-12 of full scale; the recurrence in f32 where the slope allows it: 6e-7)
                assert rms(g, ref.data[c]) <= 1e-6 and np.max(np.abs(g.astype(np.float64) - ref.data[c])) <= 1e-6, (kind, rate, ch, interp, which, freq, mono, segs, i, c, len(g))
                assert np.max(np.abs(g - h)) <= 4e-7, (kind, rate, ch, interp, which, freq, mono, segs, i, c)
