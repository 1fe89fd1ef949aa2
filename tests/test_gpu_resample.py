"""GPU parity: fused decode → resample kernels vs the CPU oracle (same seeded inputs, through the C ABI).

Bars: integer-valued stream outputs bit-exact; float stages ≤ 1e-6 RMS on the [-1,1]
scale (un-floored stream outputs are divided by 128 first).  With AUKIT_F64 storage the kernels use the
reference's fp64 operation order, so the tests additionally demand ≤ 1 ulp-level agreement there.
"""
import numpy as np
import pytest

from tests.util import pcm16, rms, signal

pytestmark = pytest.mark.gpu

RATES = [8000, 11025, 22050, 44100, 48000]


def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
@pytest.mark.parametrize("rate", [8000, 22050, 44100])
def test_pcm16_mono_resample_f64(ctx, oracle, rate, interp):
    B, N = _B(), _N()
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate([rate, rate // 3 + 7, 5, 2000])]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed")
    out = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    for s, got in zip(streams, out):
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, rate), 48000, oracle.INTERP[interp])
        assert len(got[0]) == len(ref.data[0])
        assert np.max(np.abs(got[0] - ref.data[0]), initial=0) <= 1e-15  # fp64 op order is reproduced; pow(fx,3) may differ by 1 ulp


def test_pcm16_mono_resample_f32_tolerance(ctx, oracle):
    B, N = _B(), _N()
    s = pcm16(44100, 44100, 1, 0).tobytes()
    bt = B.Batch.upload(ctx, [s])
    out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), 48000, "cubic", dtype=N.F32).download()
    ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.CUBIC)
    assert rms(out[0][0], ref.data[0]) <= 1e-6


def test_decode_is_exact(ctx, oracle):
    B, N = _B(), _N()
    s = pcm16(10000, 44100, 1, 3).tobytes()
    bt = B.Batch.upload(ctx, [s, s[:200]])
    out = B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), dtype=N.F64).download()
    ref = oracle.pcm(s, 16, oracle.SIGNED, 1, 44100)
    assert np.array_equal(out[0][0], ref.data[0])
    assert np.array_equal(out[1][0], ref.data[0][:100])


@pytest.mark.parametrize("bits,dtype,be,ch,interleaved", [
    (8, "signed", False, 1, True), (8, "unsigned", False, 2, True), (16, "unsigned", True, 1, True), (16, "signed", True, 2, True),
    (24, "signed", False, 2, True), (24, "unsigned", True, 1, True), (32, "signed", False, 1, True), (32, "float", False, 2, True),
    (32, "float", True, 1, True), (16, "signed", False, 2, False), (16, "signed", False, 3, True),
])
def test_pcm_formats(ctx, oracle, bits, dtype, be, ch, interleaved):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(77 + bits + ch))
    frames = 3001
    if dtype == "float":
        raw = rng.uniform(-1, 1, frames * ch).astype(">f4" if be else "<f4").tobytes()
    else:
        raw = rng.integers(0, 256, frames * ch * (bits // 8), dtype=np.uint8).tobytes()
    bt = B.Batch.upload(ctx, [raw])
    desc = B.make_desc(N.CODEC_PCM, ch, 22050, bits, dtype, big_endian=be, interleaved=interleaved)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()[0]
    ref = oracle.pcm(raw, bits, oracle.DTYPE[dtype], ch, 22050, interleaved, be)
    for c in range(ch):
        assert np.array_equal(got[c], ref.data[c]), f"channel {c}"
    got = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F64).download()[0]
    ref = oracle.resample(ref, 48000, oracle.CUBIC)
    for c in range(ch):
        assert np.max(np.abs(got[c] - ref.data[c])) <= 1e-15


@pytest.mark.parametrize("bits,dtype,be,ch,interleaved", [
    (8, "unsigned", False, 2, True), (16, "unsigned", True, 1, True), (16, "signed", True, 2, True), (24, "signed", False, 2, True), (24, "unsigned", True, 1, True),
    (32, "signed", False, 1, True), (32, "float", False, 2, True), (32, "float", True, 1, True), (16, "signed", False, 2, False), (16, "signed", False, 3, True),
])
@pytest.mark.parametrize("rate,new_rate", [(22050, 48000), (48000, 44100)])
def test_pcm_formats_resample_f32_through_rows(ctx, oracle, bits, dtype, be, ch, interleaved, rate, new_rate):
    """aukit.pcm(any format):resample(...) with F32 storage: unpacked to f32 rows, then the f32 wave kernel — ≤ 1e-6 RMS from the oracle."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(177 + bits + ch))
    streams = []
    for frames in (3001, 1, 40000):
        if dtype == "float":
            streams.append(rng.uniform(-1, 1, frames * ch).astype(">f4" if be else "<f4").tobytes())
        else:
            streams.append(rng.integers(0, 256, frames * ch * (bits // 8), dtype=np.uint8).tobytes())
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dtype, big_endian=be, interleaved=interleaved)
    got = B.decode_resample(ctx, bt, desc, new_rate, "cubic", dtype=N.F32).download()
    in_range = dtype == "signed" or (dtype == "unsigned" and bits == 8)  # samples within [-1, 1]: the only formats the rows path takes (see api_resample.hip)
    one_launch = interleaved and ch <= 2 and (in_range or dtype == "float")   # fast_fmt.hip
    assert ctx.last_kernel()[0].startswith("k_fast_wave_fmt<" if one_launch else ("k_fast_wave<audio_f32", "k_fast_resample<audio_f32") if in_range else "k_resample<"), ctx.last_kernel()
    for s, g in zip(streams, got):
        ref = oracle.resample(oracle.pcm(s, bits, oracle.DTYPE[dtype], ch, rate, interleaved, be), new_rate, oracle.CUBIC)
        for c in range(ch):
            assert len(g[c]) == len(ref.data[c])
            if len(g[c]):
                assert rms(g[c], ref.data[c]) <= 1e-6, c


@pytest.mark.parametrize("bits,dtype,be", [(8, "signed", False), (8, "unsigned", False), (16, "signed", True), (24, "signed", False), (24, "signed", True),
                                           (32, "signed", False), (32, "signed", True), (32, "float", False), (32, "float", True)])
@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("rate,new_rate,interp", [(44100, 48000, "cubic"), (8000, 48000, "linear"), (48000, 44100, "linear"), (96000, 48000, "cubic"), (48000, 48000, "cubic")])
def test_pcm_formats_decode_and_resample_in_one_launch(ctx, oracle, bits, dtype, be, ch, rate, new_rate, interp):
    """k_fast_wave_fmt (fast_fmt.hip): every interleaved format of one or two channels, decode + resample fused, F32 storage: ≤ 1e-6 RMS from
    the oracle.  Ragged streams (odd byte offsets for the odd frame sizes), one frame, none."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(977 + bits + ch))
    streams = []
    for frames in (3001, 1, 20001, 0, 777):
        if dtype == "float":
            streams.append(rng.uniform(-1, 1, frames * ch).astype(">f4" if be else "<f4").tobytes())
        else:
            streams.append(rng.integers(0, 256, frames * ch * (bits // 8), dtype=np.uint8).tobytes())
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dtype, big_endian=be)
    got = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
    name = ctx.last_kernel()[0]
    special = (bits == 16 and not be and dtype == "signed") or (bits == 8 and ch == 1)   # the specialised kernels keep their formats
    served = not (rate >= 2 * new_rate and (bits // 8) * ch >= 4)   # the tile's raw window + tables fit 64 KiB of LDS per workgroup (stronger down-sampling of wide frames: the older paths)
    if served: assert name.startswith("k_fast_wave") and (special or name.startswith("k_fast_wave_fmt<")), name
    for s, g in zip(streams, got):
        ref = oracle.resample(oracle.pcm(s, bits, oracle.DTYPE[dtype], ch, rate, True, be), new_rate, oracle.INTERP[interp])
        for c in range(ch):
            assert len(g[c]) == len(ref.data[c])
            if len(g[c]):
                assert rms(g[c], ref.data[c]) <= 1e-6, (c, name)
                assert np.max(np.abs(g[c] - ref.data[c])) <= 2e-6, (c, name)


@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (22050, 48000), (48000, 48000)])
def test_float_strings_beyond_one_fall_back_to_the_reference_order(ctx, oracle, ch, rate, new_rate):
    """A float string may hold samples beyond ±1: where the reference's rounded position misses an integer it clamps what an exact position
    copies.  The fused kernel flags such input and the reference-order kernel queued behind it redoes the call: the result is the oracle's."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(31 + ch))
    wild = (rng.uniform(-1, 1, 30000 * ch) * 3).astype("<f4")
    tame = rng.uniform(-1, 1, 30000 * ch).astype("<f4")
    one = tame.copy(); one[12345] = 1.5
    for raw, flagged in ((wild, True), (tame, False), (one, True)):
        bt = B.Batch.upload(ctx, [tame.tobytes(), raw.tobytes()])
        desc = B.make_desc(N.CODEC_PCM, ch, rate, 32, "float")
        got = B.decode_resample(ctx, bt, desc, new_rate, "cubic", dtype=N.F32).download()
        assert ctx.last_kernel()[0].startswith("k_fast_wave_fmt<float32"), ctx.last_kernel()
        for s, g in zip((tame, raw), got):
            ref = oracle.resample(oracle.pcm(s.tobytes(), 32, oracle.DTYPE["float"], ch, rate, True, False), new_rate, oracle.CUBIC)
            for c in range(ch):
                assert np.max(np.abs(g[c] - ref.data[c])) <= (1e-6 if flagged else 2e-6), (c, flagged)


@pytest.mark.parametrize("ulaw", [True, False])
@pytest.mark.parametrize("rate,new_rate,interp", [(8000, 48000, "cubic"), (8000, 44100, "linear"), (16000, 8000, "cubic")])
def test_g711_stereo_decode_and_resample_in_one_launch(ctx, oracle, ulaw, rate, new_rate, interp):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(6))
    streams = [rng.integers(0, 256, n * 2, dtype=np.uint8).tobytes() for n in (8000, 8000 * 2 + 1235, 17, 1, 0)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, 2, rate, ulaw=ulaw)
    res = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
    assert ctx.last_kernel()[0].startswith("k_fast_wave_fmt<" + ("ulaw" if ulaw else "alaw")), ctx.last_kernel()
    for s, r in zip(streams, res):
        refr = oracle.resample(oracle.g711(s, ulaw, 2, rate), new_rate, oracle.INTERP[interp])
        for c in range(2):
            assert len(r[c]) == len(refr.data[c])
            if len(r[c]):
                assert np.max(np.abs(r[c] - refr.data[c])) <= 1e-6


@pytest.mark.parametrize("rate,new_rate", [(48000, 48000), (44100, 44100), (96000, 48000), (48000, 24000)])
def test_equal_rates_and_integer_decimation_every_kernel_family(oracle, rate, new_rate):
    """Every position is an integer (the fast kernels run b == 1 as 2a / 2): F64 storage bit for bit the oracle, F32 storage ≤ 1e-6 in the f32
    kernels, in fp64 arithmetic (exact_math 1) and in the reference order (exact_math 2); ragged streams incl. one sample and none."""
    B, N = _B(), _N()
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate((50000, 1, 1025, 0))]
    for dtype, em in ((N.F64, 0), (N.F32, 0), (N.F32, 1), (N.F32, 2)):
        c2 = B.Context(0, dtype=dtype)
        c2.set_option(N.OPT_EXACT_MATH, em)
        bt = B.Batch.upload(c2, streams)
        for interp in ("linear", "cubic"):
            got = B.decode_resample(c2, bt, B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed"), new_rate, interp).download()
            for s, g in zip(streams, got):
                ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, rate), new_rate, oracle.INTERP[interp]).data[0]
                assert len(g[0]) == len(ref)
                if len(ref):
                    assert np.max(np.abs(g[0] - ref)) <= (0.0 if dtype == N.F64 else 1e-6), (dtype, em, interp, c2.last_kernel()[0])
        c2.close()


def test_pcm_uneven_data_is_an_error(ctx):
    B, N = _B(), _N()
    bt = B.Batch.upload(ctx, [b"\0" * 7])
    with pytest.raises(N.AukitError) as e:
        B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 2, 44100, 16, "signed"))
    assert "uneven amount of data per channel" in str(e.value)


@pytest.mark.parametrize("ulaw", [True, False])
@pytest.mark.parametrize("ch", [1, 2])
def test_g711_audio_path(ctx, oracle, ulaw, ch):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(5))
    streams = [rng.integers(0, 256, n * ch, dtype=np.uint8).tobytes() for n in (8000, 8000 * 2 + 1234, 17)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, ch, 8000, ulaw=ulaw)
    dec = B.decode(ctx, bt, desc, dtype=N.F64).download()
    res = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F64).download()
    for s, d, r in zip(streams, dec, res):
        ref = oracle.g711(s, ulaw, ch, 8000)
        refr = oracle.resample(ref, 48000, oracle.CUBIC)
        for c in range(ch):
            assert np.array_equal(d[c], ref.data[c])
            assert np.max(np.abs(r[c] - refr.data[c])) <= 1e-15


def test_g711_all_bytes_match_itu_tables(ctx):
    """Independent KAT: every µ-law / A-law code point against the ITU-T G.711 expansion."""
    B, N = _B(), _N()
    codes = bytes(range(256))
    bt = B.Batch.upload(ctx, [codes])
    for ulaw in (True, False):
        got = B.decode(ctx, bt, B.make_desc(N.CODEC_G711, 1, 8000, ulaw=ulaw), dtype=N.F64).download()[0][0]
        exp = np.array([_itu_ulaw(b) / 32768.0 if ulaw else _itu_alaw(b) / 32768.0 for b in range(256)])
        assert np.array_equal(got, exp)


def _itu_ulaw(u):
    u = ~u & 0xFF
    t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4)
    return (0x84 - t) if (u & 0x80) else (t - 0x84)


def _itu_alaw(a):
    a ^= 0x55
    t = (a & 0x0F) << 4
    seg = (a & 0x70) >> 4
    if seg == 0:
        t += 8
    elif seg == 1:
        t += 0x108
    else:
        t = (t + 0x108) << (seg - 1)
    return t if (a & 0x80) else -t


def test_audio_resample_matches_fused(ctx, oracle):
    B, N = _B(), _N()
    a = [[signal(5000, 22050, 2, 0), signal(5000, 22050, 2, 1)], [signal(1234, 22050, 2, 2), signal(1234, 22050, 2, 3)]]
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    for interp in ("none", "linear", "cubic"):
        got = B.resample(ctx, ab, 48000, interp).download()
        for s in range(2):
            ref = oracle.resample(oracle.Audio(a[s], 22050), 48000, oracle.INTERP[interp])
            for c in range(2):
                assert np.max(np.abs(got[s][c] - ref.data[c])) <= 1e-15
    # downsampling too (ratio < 1)
    got = B.resample(ctx, ab, 8000, "cubic").download()
    ref = oracle.resample(oracle.Audio(a[0], 22050), 8000, oracle.CUBIC)
    assert np.max(np.abs(got[0][0] - ref.data[0])) <= 1e-15


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
@pytest.mark.parametrize("rate", RATES)
def test_stream_pcm_mono16(ctx, oracle, rate, interp):
    B, N = _B(), _N()
    nsamp = [int(rate * 2.5), rate, rate + 3, 10, int(rate * 1.0001) + 2]
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(nsamp)]
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed"), interp, dtype=N.F64)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 1, rate, False, False, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, (i, ck.nchunks[i], ref.nchunks)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert np.allclose(ck.pos[i][:ref.nchunks], ref.chunk_pos, rtol=0, atol=0)
        assert ck.status[i] == ref.final_status
        assert abs(ck.length_seconds[i] - ref.length_seconds) == 0
        assert np.max(np.abs(got[i][0] - ref.data[0]), initial=0) <= 1e-13  # values up to 128: 1 ulp = 2.8e-14


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate", [44100, 22050, 8000, 32000])
def test_stream_pcm_mono16_f32_wave_kernel(ctx, oracle, rate, interp):
    """F32 storage takes the f32 wave kernel with the stream.pcm epilogue: same chunking, ≤ 1e-6 RMS on the [-1,1] scale, and ≤ 1e-6
    RMS from the fp64 reference-order kernel on the same batch (ragged streams: chunk boundaries, tile boundaries, short tails)."""
    B, N = _B(), _N()
    nsamp = [int(rate * 2.5), rate, rate + 3, 10, int(rate * 1.0001) + 2, 1100, 2]
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(nsamp)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed")
    out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=N.F32)
    assert ctx.last_kernel()[0].startswith("k_fast_wave_stream<pcm_s16le_mono") and ctx.last_kernel()[0].endswith("stream_pcm>") and "k_fast_wave_stream" in ctx.last_kernel()[0]
    got = out.download()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        out2, ck2 = B.stream_decode(ctx, bt, desc, interp, dtype=N.F32)
        assert ctx.last_kernel()[0].startswith(("k_resample<", "k_exact_wave<", "k_wave_f64<"))  # fp64 arithmetic (round 3: wave_f64.hip's stream.pcm epilogue; reference-order kernels where it does not apply)
        got2 = out2.download()
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 1, rate, False, False, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert len(got[i][0]) == len(ref.data[0])
        if len(ref.data[0]):
            assert rms(got[i][0] / 128, ref.data[0] / 128) <= 1e-6, i
            assert np.max(np.abs(got[i][0] - ref.data[0])) <= 2e-4, i  # absolute, on the [-128, 127] scale
            assert rms(got[i][0] / 128, got2[i][0] / 128) <= 1e-6, i


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("dt", ["unsigned", "signed"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (22050, 48000), (8000, 48000), (48000, 44100), (48000, 48000)])
def test_fast_f32_pcm8_mono(ctx, oracle, rate, new_rate, dt, interp):
    """aukit.pcm(d, 8, dt, 1, rate):resample(new_rate) with F32 storage: the wave kernel reads the bytes themselves — ≤ 1e-6 RMS from the
    oracle, every byte value present, ragged lengths incl. one sample and an empty string."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(rate + new_rate))
    streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (int(rate * 1.3), 5000, 1024 * 3 + 5, 1, 0, 77)]
    streams[1] = bytes(range(256)) * 20
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 1, rate, 8, dt)
    got = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
    assert ctx.last_kernel()[0].startswith("k_fast_wave<pcm8_mono"), ctx.last_kernel()
    for s, g in zip(streams, got):
        ref = oracle.resample(oracle.pcm(s, 8, oracle.DTYPE[dt], 1, rate), new_rate, oracle.INTERP[interp])
        assert len(g[0]) == len(ref.data[0])
        if len(g[0]):
            assert rms(g[0], ref.data[0]) <= 1e-6


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("bits,dt,be,ch,mono,rate,kernel", [
    (8, "unsigned", False, 1, False, 48000, "k_fast_wave_stream<pcm8_mono"),   # the classic pre-converted speaker file: bytes read directly
    (8, "signed", False, 1, False, 44100, "k_fast_wave_stream<pcm8_mono"),
    (8, "unsigned", False, 2, False, 48000, "k_fast_wave_fmt<unsigned8,2ch"),   # every other interleaved format of one or two channels: one launch (fast_fmt.hip)
    (8, "unsigned", False, 2, True, 22050, "k_fast_wave_fmt<unsigned8,2ch"),
    (24, "signed", False, 2, False, 44100, "k_fast_wave_fmt<signed24,2ch"),
    (24, "signed", True, 2, True, 32000, "k_fast_wave_fmt<signed24be,2ch"),
    (24, "signed", True, 3, True, 32000, "k_fast_wave_stream<audio_f32"),     # more channels: unpacked to f32 rows first
    (16, "unsigned", True, 1, False, 48000, "k_fast_wave_fmt<unsigned16be,1ch"),
    (24, "unsigned", False, 2, True, 44100, "k_fast_wave_fmt<unsigned24,2ch"),
    (32, "unsigned", False, 1, False, 22050, "k_fast_wave_fmt<unsigned32,1ch"),
    (32, "signed", False, 1, False, 8000, "k_fast_wave_fmt<signed32,1ch"),
    (32, "float", False, 2, False, 48000, "k_fast_wave_fmt<float32,2ch"),
    (32, "float", True, 2, True, 44100, "k_fast_wave_fmt<float32be,2ch"),
    (16, "signed", True, 2, False, 44100, "k_fast_wave_fmt<signed16be,2ch"),
    (16, "signed", False, 1, False, 48000, "k_fast_wave_stream<pcm_s16le_mono"),  # equal rates take the 16-bit kernels too
    (16, "signed", False, 2, False, 48000, "k_fast_wave_stream_s16x2<"),
])
def test_stream_pcm_other_formats_f32_wave_kernels(ctx, oracle, bits, dt, be, ch, mono, rate, kernel, interp):
    """stream.pcm with F32 storage on PCM formats other than 16-bit 44.1 kHz: 8-bit mono read directly, every other depth / type / byte
    order / channel count through f32 rows (the mono mix made while unpacking), equal rates (48 kHz sources) included — the oracle's
    chunking, ≤ 1e-6 RMS on the [-1,1] scale, ≤ 1e-6 RMS from the reference-order kernel; ragged streams."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(1000 + bits * 7 + ch))
    nfr = [int(rate * 2.5), rate, rate + 3, 10, int(rate * 1.0001) + 2, 1100, 2]
    streams = []
    for n in nfr:
        if dt == "float":
            streams.append(rng.uniform(-1, 1, n * ch).astype(">f4" if be else "<f4").tobytes())
        else:
            streams.append(rng.integers(0, 256, n * ch * (bits // 8), dtype=np.uint8).tobytes())
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dt, big_endian=be)
    out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F32)
    assert ctx.last_kernel()[0].startswith(kernel), ctx.last_kernel()
    got = out.download()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        out2, _ = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F32)
        assert ctx.last_kernel()[0].startswith(("k_resample<", "k_exact_wave<", "k_wave_f64<")), ctx.last_kernel()
        got2 = out2.download()
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, bits, oracle.DTYPE[dt], ch, rate, be, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        assert ck.status[i] == ref.final_status
        assert len(got[i]) == ref.channels == (1 if (mono and ch > 1) else ch)
        for c in range(ref.channels):
            assert len(got[i][c]) == len(ref.data[c])
            if len(ref.data[c]):
                assert rms(got[i][c] / 128, ref.data[c] / 128) <= 1e-6, (i, c)
                assert rms(got[i][c] / 128, got2[i][c] / 128) <= 1e-6, (i, c)


@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate", [44100, 32000, 22050, 8000])
def test_stream_pcm_stereo16_f32_wave_kernel(ctx, oracle, rate, interp, mono):
    """Interleaved 16-bit stereo (the WAV layout) with F32 storage takes the stereo wave kernel with the stream.pcm epilogue — both
    channels, or the channels averaged as they are read (`mono`): same chunking as the oracle, ≤ 1e-6 RMS on the [-1,1] scale, and
    ≤ 1e-6 RMS from the fp64 reference-order kernel on the same batch (ragged streams, a batch offset that is not 16-byte aligned)."""
    B, N = _B(), _N()
    nfr = [int(rate * 2.5), rate, rate + 3, 10, int(rate * 1.0001) + 2, 1100, 2]
    streams = [np.stack([pcm16(n, rate, 1, 2 * i), pcm16(n, rate, 1, 2 * i + 1)], 1).tobytes() for i, n in enumerate(nfr)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 2, rate, 16, "signed")
    out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F32)
    assert ctx.last_kernel()[0].startswith("k_fast_wave_stream_s16x2<") and ctx.last_kernel()[0].endswith(("mono>" if mono else "stereo>")), ctx.last_kernel()
    got = out.download()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        out2, ck2 = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.F32)
        assert ctx.last_kernel()[0].startswith("k_resample<")
        got2 = out2.download()
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 2, rate, False, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert len(got[i]) == ref.channels == (1 if mono else 2)
        for c in range(ref.channels):
            assert len(got[i][c]) == len(ref.data[c])
            if len(ref.data[c]):
                assert rms(got[i][c] / 128, ref.data[c] / 128) <= 1e-6, (i, c)
                assert np.max(np.abs(got[i][c] - ref.data[c])) <= 2e-4, (i, c)  # absolute, on the [-128, 127] scale
                assert rms(got[i][c] / 128, got2[i][c] / 128) <= 1e-6, (i, c)


def test_stream_pcm_stereo_and_mono_mix(ctx, oracle):
    B, N = _B(), _N()
    st = np.stack([pcm16(30000, 22050, 1, 0), pcm16(30000, 22050, 1, 1)], 1).tobytes()
    bt = B.Batch.upload(ctx, [st])
    desc = B.make_desc(N.CODEC_PCM, 2, 22050, 16, "signed")
    for mono in (False, True):
        out, ck = B.stream_decode(ctx, bt, desc, "cubic", mono=mono, dtype=N.F64)
        got = out.download()[0]
        ref = oracle.stream_pcm(st, 16, oracle.SIGNED, 2, 22050, False, mono, oracle.CUBIC)
        assert len(got) == ref.channels
        assert ck.nchunks[0] == ref.nchunks
        for c in range(ref.channels):
            assert np.max(np.abs(got[c] - ref.data[c])) <= 1e-13


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("ch,extra", [(2, 1), (3, 1), (3, 2)])
def test_stream_pcm_data_ends_inside_a_frame(ctx, oracle, interp, ch, extra):
    """stream.pcm on whole samples but not whole frames (aukit.lua:2367-2407; refused until round 6): with the mono mix-down every index is read for all
    channels, so the partial frame counts for nothing — chunk for chunk the oracle's (which reads sample by sample like the Lua); without the
    mix-down the reference's last chunk is longer in its first channels, which the ABI cannot say: refused by name"""
    B, N = _B(), _N()
    frames = 22050 * 2 + 777
    x = np.stack([pcm16(frames, 22050, 1, 40 + c) for c in range(ch)], 1).astype("<i2").tobytes()
    ragged = x + pcm16(extra, 22050, 1, 99).astype("<i2").tobytes()
    bt = B.Batch.upload(ctx, [ragged, x])
    desc = B.make_desc(N.CODEC_PCM, ch, 22050, 16, "signed")
    out, ck = B.stream_decode(ctx, bt, desc, interp, mono=True, dtype=N.F64)
    got = out.download()
    for i, s in enumerate((ragged, x)):
        ref = oracle.stream_pcm(s, 16, oracle.SIGNED, ch, 22050, False, True, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        assert (ck.status[i] != 0) == (ref.final_status != 0), i
        assert np.max(np.abs(got[i][0] - ref.data[0]), initial=0) <= 1e-13
    with pytest.raises(N.AukitError, match="ends inside a frame"):
        B.stream_decode(ctx, bt, desc, interp, mono=False, dtype=N.F64)
    with pytest.raises(N.AukitError, match="ends inside a sample"):
        B.stream_decode(ctx, B.Batch.upload(ctx, [x + b"\x01"]), desc, interp, mono=True, dtype=N.F64)


def test_stream_pcm_f32_tolerance_and_float_input(ctx, oracle):
    B, N = _B(), _N()
    s = pcm16(44100 * 2, 44100, 1, 9).tobytes()
    bt = B.Batch.upload(ctx, [s])
    out, _ = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), "cubic", dtype=N.F32)
    ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 1, 44100, False, False, oracle.CUBIC)
    assert rms(out.download()[0][0] / 128, ref.data[0] / 128) <= 1e-6
    f = signal(30000, 32000, 1, 4).astype("<f4").tobytes()  # float input ends with nil reads, not an error
    bt = B.Batch.upload(ctx, [f])
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 32000, 32, "float"), "cubic", dtype=N.F64)
    ref = oracle.stream_pcm(f, 32, oracle.FLOAT, 1, 32000, False, False, oracle.CUBIC)
    assert ck.nchunks[0] == ref.nchunks and list(ck.lens[0][:ref.nchunks]) == list(ref.chunk_len[:, 0])
    assert np.max(np.abs(out.download()[0][0] - ref.data[0])) <= 1e-13


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("ch,mono", [(2, False), (2, True), (3, False), (3, True)])
def test_stream_pcm_float_end_of_data(ctx, oracle, interp, ch, mono):
    """32-bit float strings: read() returns nil past the end (aukit.lua:2291-2311).  Per channel the interpolators fall back on
    the neighbouring sample and the chunk runs on to the next whole index; with a mono mix-down the lazy __index adds that nil
    (:2368) and the chunk ends at the first read past the end, like the integer formats (found by the seed soak)."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(77 + ch))
    streams = [rng.uniform(-1, 1, n * ch).astype("<f4").tobytes() for n in (3, 4, 1716, 12000 + 7)]
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_PCM, ch, 12000, 32, "float"), interp, mono=mono, dtype=N.F64)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, 32, oracle.FLOAT, ch, 12000, False, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        for c in range(ref.channels):
            assert np.max(np.abs(got[i][c] - ref.data[c]), initial=0) <= 1e-13, (i, c)


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
@pytest.mark.parametrize("ch,mono", [(1, False), (2, False), (2, True)])
def test_stream_g711_bit_exact(ctx, oracle, interp, ch, mono):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(11))
    streams = [oracle.gen_g711(pcm16(n * ch, 8000, 2, i), True) for i, n in enumerate([8000 * 3, 8000 * 2 + 4000, 100])]
    streams.append(rng.integers(0, 256, 8000 * ch, dtype=np.uint8).tobytes())
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_G711, ch, 8000, ulaw=True), interp, mono=mono, dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_g711(s, True, ch, 8000, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        for c in range(ref.channels):
            assert np.array_equal(got[i][c], ref.data[c]), (i, c)


@pytest.mark.parametrize("ch", [2, 3])
@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_stream_g711_ragged_byte_count(ctx, oracle, ch, interp):
    """a byte count that is not a multiple of the channel count (aukit.lua:2878-2911; refused until round 6): the calls before the last deliver
    their chunks, the last one — whose shorter channels read nil at the end — raises: chunk for chunk the oracle's, and its final status"""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(77 + ch))
    streams = [rng.integers(0, 256, 8000 * ch * 2 + 1, dtype=np.uint8).tobytes(), rng.integers(0, 256, 8000 * ch + ch - 1, dtype=np.uint8).tobytes(),
               rng.integers(0, 256, ch + 1, dtype=np.uint8).tobytes(), rng.integers(0, 256, 8000 * ch * 2, dtype=np.uint8).tobytes()]
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_G711, ch, 8000, ulaw=True), interp, dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_g711(s, True, ch, 8000, False, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, i
        assert (ck.status[i] != 0) == (ref.final_status != 0), i
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        for c in range(ch):
            assert np.array_equal(got[i][c], ref.data[c]), (i, c)
    assert [int(x != 0) for x in ck.status] == [1, 1, 1, 0]


@pytest.mark.parametrize("ch", [2, 3])
@pytest.mark.parametrize("rate,interp,alaw", [(8000, "cubic", False), (8000, "linear", True), (44100, "cubic", False), (22050, "linear", False), (11025, "cubic", True)])
def test_stream_g711_several_channels_on_the_floor_kernel(ctx, oracle, monkeypatch, ch, rate, interp, alaw):
    """stream.g711 with interleaved channels (aukit.lua:2878-2911; round 4): planar byte rows (k_deinterleave_bytes) + the mono three-tier floor
    kernel on n * channels rows — bit for bit the oracle's chunks and the reference-order kernel's, ragged lengths, random bytes and plateaus."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(rate + ch))
    flat = np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), 25 * ch)
    streams = [oracle.gen_g711(pcm16(int(rate * 2.3) * ch, rate, 2, 5), not alaw), rng.integers(0, 256, (rate + 17) * ch, dtype=np.uint8).tobytes(), flat.tobytes()[:len(flat) // ch * ch],
               (b"\x00\xff\x80" * 700)[:2100 // ch * ch], b"\x7f" * ch]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, ch, rate, ulaw=not alaw)
    for dt in (N.I8, N.F64):
        out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
        assert ctx.last_kernel()[0].startswith("k_floor_wave_g711"), ctx.last_kernel()
        got = out.download()
        monkeypatch.setenv("AUKIT_G711_NO_PLANAR", "1")
        out2, _ = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
        monkeypatch.delenv("AUKIT_G711_NO_PLANAR")
        assert ctx.last_kernel()[0].startswith("k_resample<"), ctx.last_kernel()
        got2 = out2.download()
        for i, s in enumerate(streams):
            ref = oracle.stream_g711(s, not alaw, ch, rate, False, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            for c in range(ch):
                assert np.array_equal(got[i][c], ref.data[c]), (i, c, dt)
                assert np.array_equal(got2[i][c], ref.data[c]), (i, c, dt)


@pytest.mark.parametrize("alaw", [False, True])
@pytest.mark.parametrize("rate", [8000, 11025, 16000, 22050, 44100, 6000, 9600, 7200])   # the last three: one and five phases per lane in registers (8000: three)
@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_stream_g711_guarded_wave_kernel(ctx, oracle, interp, rate, alaw):
    """Mono stream.g711 takes the guarded short-cut kernel (floor_wave.hip): every output equal to the oracle's and to the
    reference-order kernel's, for integer (8 k, 16 k → 48 k) and non-integer ratios, both laws, I8 and F64 outputs.  The byte
    patterns include runs of equal samples and full-scale steps: interpolated values that are exact integers (guard → fallback)."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(rate + (1 if alaw else 0)))
    flat = np.repeat(rng.integers(0, 256, 400, dtype=np.uint8), 25)  # plateaus: value == integer between equal samples
    streams = [oracle.gen_g711(pcm16(int(rate * 2.3), rate, 2, 5), not alaw), rng.integers(0, 256, rate + 17, dtype=np.uint8).tobytes(), flat.tobytes(), b"\x00\xff" * 700, b"\x7f"]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, 1, rate, ulaw=not alaw)
    for dt in (N.I8, N.F64):
        out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
        assert ctx.last_kernel()[0].startswith("k_floor_wave_g711"), ctx.last_kernel()
        got = out.download()
        ctx.set_option(N.OPT_EXACT_MATH, 1)
        try:
            out2, _ = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
            assert ctx.last_kernel()[0].startswith("k_resample<")
            got2 = out2.download()
        finally:
            ctx.set_option(N.OPT_EXACT_MATH, 0)
        for i, s in enumerate(streams):
            ref = oracle.stream_g711(s, not alaw, 1, rate, False, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            assert np.array_equal(got[i][0], ref.data[0]), (i, dt)
            assert np.array_equal(got[i][0], got2[i][0]), (i, dt)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (22050, 48000), (48000, 44100), (32000, 48000)])
def test_fast_f32_pcm16_stereo(ctx, oracle, rate, new_rate, interp):
    """Interleaved 16-bit stereo (the WAV layout) with F32 storage: wave kernel with two LDS tables, both channels ≤ 1e-6 RMS from
    the oracle and from the fp64 reference-order kernel; ragged lengths incl. one frame and an empty stream."""
    B, N = _B(), _N()
    nfr = [int(rate * 1.7), 5000, 1024 * 3 + 5, 1, 0, 77]
    streams = [np.stack([pcm16(n, rate, 1, 2 * i), pcm16(n, rate, 1, 2 * i + 1)], 1).tobytes() for i, n in enumerate(nfr)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 2, rate, 16, "signed")
    got = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
    name = ctx.last_kernel()[0]
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        got2 = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32).download()
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)
    if new_rate >= rate:  # frames start on dword boundaries in this batch and the window fits 4 vectors per lane → the stereo wave kernel ran
        assert name.startswith("k_fast_wave_s16x2<"), name
    else:                 # down-sampling needs a longer window: eight vectors per lane (nv8) of the same kernel
        assert name.startswith("k_fast_wave_s16x2<") and "nv8" in name, name
    for s, g, g2 in zip(streams, got, got2):
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 2, rate), new_rate, oracle.INTERP[interp])
        for c in range(2):
            assert len(g[c]) == len(ref.data[c])
            if len(ref.data[c]):
                assert rms(g[c], ref.data[c]) <= 1e-6 and np.max(np.abs(g[c] - ref.data[c])) <= 2e-6, c
                assert rms(g[c], g2[c]) <= 1e-6


def test_empty_and_ragged_batches(ctx, oracle):
    B, N = _B(), _N()
    streams = [b"", pcm16(3, 44100, 1, 0).tobytes(), b"", pcm16(50000, 44100, 1, 1).tobytes()]
    bt = B.Batch.upload(ctx, streams)
    out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), 48000, "cubic", dtype=N.F64).download()
    for s, got in zip(streams, out):
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.CUBIC)
        assert len(got[0]) == len(ref.data[0])
        assert np.max(np.abs(got[0] - ref.data[0]), initial=0) <= 1e-15


# ---------------------------------------------------------------- F32 fast path (fast.hip): tolerance parity
@pytest.mark.parametrize("x4", [1, 0])
@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (8000, 48000), (22050, 48000), (48000, 44100), (32000, 48000), (11025, 48000)])
def test_fast_f32_pcm16(ctx, oracle, rate, new_rate, interp, x4):
    B, N = _B(), _N()
    ctx.set_option(N.OPT_STORE_X4, x4)
    try:
        lens = [rate * 2 + 11, 9000, 4097, 1, 2, 3, 5, 700]
        streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(lens)]
        bt = B.Batch.upload(ctx, streams)
        out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed"), new_rate, interp, dtype=N.F32)
        name, _, _ = ctx.last_kernel()
        assert name.startswith("k_fast_"), name
        got = out.download()
        for s, g in zip(streams, got):
            ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, rate), new_rate, oracle.INTERP[interp])
            assert len(g[0]) == len(ref.data[0])
            if len(g[0]):
                assert rms(g[0], ref.data[0]) <= 1e-6
                assert np.max(np.abs(g[0] - ref.data[0])) <= 4e-6
    finally:
        ctx.set_option(N.OPT_STORE_X4, 1)


def test_fast_f32_g711_and_audio(ctx, oracle):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(21))
    streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (80000, 8000, 33, 1)]
    bt = B.Batch.upload(ctx, streams)
    out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True), 48000, "cubic", dtype=N.F32)
    assert ctx.last_kernel()[0].startswith("k_fast_")
    for s, g in zip(streams, out.download()):
        ref = oracle.resample(oracle.g711(s, True, 1, 8000), 48000, oracle.CUBIC)
        assert len(g[0]) == len(ref.data[0]) and rms(g[0], ref.data[0]) <= 1e-6
    a = [[signal(30000, 22050, 3, 0), signal(30000, 22050, 3, 1)], [signal(777, 22050, 3, 2), signal(777, 22050, 3, 3)]]
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F32)
    got = B.resample(ctx, ab, 48000, "cubic").download()
    assert ctx.last_kernel()[0].startswith("k_fast_")
    for s in range(2):
        ref = oracle.resample(oracle.Audio([x.astype(np.float32).astype(np.float64) for x in a[s]], 22050), 48000, oracle.CUBIC)
        for c in range(2):
            assert rms(got[s][c], ref.data[c]) <= 1e-6


def test_exact_math_option_uses_fp64_kernel(ctx, oracle):
    B, N = _B(), _N()
    s = pcm16(20000, 44100, 1, 0).tobytes()
    bt = B.Batch.upload(ctx, [s])
    ctx.set_option(N.OPT_EXACT_MATH, 2)
    try:
        out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), 48000, "cubic", dtype=N.F32)
        assert ctx.last_kernel()[0].startswith("k_exact_wave<")  # reference-order fp64, wave tiles (exact_wave.hip); level 1: tests/test_gpu_wave_f64.py
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.CUBIC)
        assert np.array_equal(out.download()[0][0], ref.data[0].astype(np.float32).astype(np.float64)) or rms(out.download()[0][0], ref.data[0]) <= 5e-8
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (8000, 48000), (22050, 48000), (48000, 44100), (32000, 8000), (47999, 48000), (11025, 22050)])
def test_exact_wave_kernel_is_bit_identical_to_the_tiled_one(ctx, oracle, monkeypatch, rate, new_rate, interp):
    """AUKIT_F64 storage: the fused s16 decode + resample and Audio:resample on f64 rows run the reference-order code on wave tiles
    (exact_wave.hip) — every output bit-identical to the workgroup-tiled k_resample (AUKIT_EXACT_TILED=1) and within 1e-15 of the
    oracle, on ragged batches (tiny streams, tile boundaries, 3 channels), up- and down-sampling."""
    B, N = _B(), _N()
    lens = [int(rate * 1.3), 1, 2, 1023, 1024, 1025, 5000, rate + 7]
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(lens)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed")
    out = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F64)
    wave = new_rate >= rate and rate != 47999  # down-sampling may need a longer window than a wave stages, 47999 / 48000 does not reduce: k_resample runs
    if wave:
        assert ctx.last_kernel()[0].startswith("k_exact_wave<pcm_s16le_mono"), ctx.last_kernel()
    got = out.download()
    monkeypatch.setenv("AUKIT_EXACT_TILED", "1")
    out2 = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F64)
    assert ctx.last_kernel()[0].startswith("k_resample<")
    got2 = out2.download()
    monkeypatch.delenv("AUKIT_EXACT_TILED")
    for i, s in enumerate(streams):
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, rate), new_rate, oracle.INTERP[interp])
        assert len(got[i][0]) == len(ref.data[0])
        assert np.array_equal(got[i][0], got2[i][0]), i
        assert np.max(np.abs(got[i][0] - ref.data[0]), initial=0) <= 1e-15, i
    # Audio:resample on f64 rows, 3 channels
    rng = np.random.Generator(np.random.PCG64(rate + new_rate))
    a = [[rng.uniform(-1, 1, n) for _ in range(3)] for n in (4000, 1, 1025, rate // 2 + 3)]
    ab = B.AudioBatch.upload(ctx, a, rate, dtype=N.F64)
    r = B.resample(ctx, ab, new_rate, interp)
    if wave:
        assert ctx.last_kernel()[0].startswith("k_exact_wave<audio_f64"), ctx.last_kernel()
    g = r.download()
    monkeypatch.setenv("AUKIT_EXACT_TILED", "1")
    g2 = B.resample(ctx, ab, new_rate, interp).download()
    monkeypatch.delenv("AUKIT_EXACT_TILED")
    for i in range(len(a)):
        ref = oracle.resample(oracle.Audio(a[i], rate), new_rate, oracle.INTERP[interp])
        for c in range(3):
            assert len(g[i][c]) == len(ref.data[c])
            assert np.array_equal(g[i][c], g2[i][c]), (i, c)
            assert np.max(np.abs(g[i][c] - ref.data[c]), initial=0) <= 1e-15, (i, c)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate", [44100, 8000, 22050, 32000, 48000 - 1])
def test_exact_wave_stream_pcm_is_bit_identical_to_the_tiled_kernel(ctx, oracle, monkeypatch, rate, interp):
    """stream.pcm on s16 mono with AUKIT_F64 storage: the reference-order code + stream epilogue on wave tiles, bit-identical to
    k_resample's (AUKIT_EXACT_TILED=1) and within 1e-13 of the oracle; ragged streams, chunk and tile boundaries"""
    B, N = _B(), _N()
    nsamp = [int(rate * 2.5), rate, rate + 3, 10, int(rate * 1.0001) + 2, 1100, 2, 1025]
    streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(nsamp)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed")
    out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=N.F64)
    if rate != 47999:
        assert ctx.last_kernel()[0].startswith("k_exact_wave<pcm_s16le_mono") and ctx.last_kernel()[0].endswith("stream_pcm>"), ctx.last_kernel()
    got = out.download()
    monkeypatch.setenv("AUKIT_EXACT_TILED", "1")
    out2, ck2 = B.stream_decode(ctx, bt, desc, interp, dtype=N.F64)
    assert ctx.last_kernel()[0].startswith("k_resample<")
    got2 = out2.download()
    monkeypatch.delenv("AUKIT_EXACT_TILED")
    for i, s in enumerate(streams):
        ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 1, rate, False, False, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert np.array_equal(got[i][0], got2[i][0]), i
        assert np.max(np.abs(got[i][0] - ref.data[0]), initial=0) <= 1e-13, i
