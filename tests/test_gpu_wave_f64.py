"""GPU parity of the graded configuration: `aukit.pcm(d, 16, "signed", 1, rate):resample(new_rate, interp)`
computed in fp64 and stored as f32 (AUKIT_OPT_EXACT_MATH = 1, k_wave_f64) against the fp64 oracle (aukit.lua:653-673, :257-266).

Bar: 1e-6 RMS on the [-1, 1] scale.  What the kernel delivers is much tighter and is asserted here: every stored f32 is within ONE
f32 ulp of the oracle's double rounded to f32, and all but a small fraction are that value exactly.  The fraction is not zero
because the two disagree below 1e-10: the kernel's position is the exact rational (i-1)·a/b, the reference's is the rounded
double x = (i-1)/ratio + 1 (relative error 1e-16, i.e. up to 5e-11 in x - floor(x) ten seconds into a 44.1 kHz stream), and
where the interpolated double lies that close to an f32 rounding boundary the two round to neighbouring floats."""
import numpy as np
import pytest

from tests.util import pcm16, rms

pytestmark = pytest.mark.gpu


def _mods():
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    return B, N


def _check(got, ref):
    assert len(got) == len(ref)
    if not len(ref):
        return 0, 0
    r32 = ref.astype(np.float32)
    g32 = got.astype(np.float32)
    assert rms(got, ref) <= 1e-6
    # one f32 ulp at most (spacing(1) = 1.19e-7 bounds every ulp in [-1, 1])
    assert np.max(np.abs(got - r32.astype(np.float64))) <= 1.2e-7
    return int(np.count_nonzero(g32 != r32)), len(ref)


@pytest.mark.parametrize("tile", ["512", "1024", "auto"])
@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (8000, 48000), (22050, 48000), (48000, 44100), (32000, 48000), (11025, 48000), (47999, 48000), (24000, 48000), (16000, 48000), (4800, 48000)])
def test_wave_f64_rounds_the_oracles_double(ctx, oracle, monkeypatch, rate, new_rate, interp, tile):
    B, N = _mods()
    if tile != "auto":
        monkeypatch.setenv("AUKIT_F64_TILE", tile)
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        lens = [rate * 2 + 11, 9000, 4097, 1, 2, 3, 5, 700, 941, 942, 1024, 1025]
        streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(lens)]
        streams.append(np.random.Generator(np.random.PCG64(5)).integers(-32768, 32768, 30000).astype(np.int16).tobytes())  # full-scale noise: the clamp works
        bt = B.Batch.upload(ctx, streams)
        out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed"), new_rate, interp, dtype=N.F32)
        name = ctx.last_kernel()[0]
        if rate == 47999:  # 47999 / 48000 does not reduce: the (q, rem) arithmetic of a tile would overflow 32 bits, the reference-order kernels run
            assert name.startswith(("k_resample<", "k_exact_wave<")), name
        elif rate == 48000 and tile == "1024":  # down-sampling: the window of a 1024-output tile + its raw samples do not fit 64 KiB of LDS next to three others
            assert name.startswith(("k_exact_wave<", "k_fast_wave_fmt<signed16,1ch")), name   # (round 3: the generic format kernel with fp64 tables and 512-output tiles fits)
        elif tile == "auto":   # no tile asked for: the phases-in-registers kernel wherever a lane meets at most five phases (up-sampling, b = 2^i, 3 * 2^i, 5 * 2^i)
            assert name.startswith("k_wave_f64<pcm_s16le_mono," + interp + ",tile"), name
            assert ("phase_regs" in name) == (rate not in (48000, 11025)), name
        else:
            assert name.startswith("k_wave_f64<pcm_s16le_mono," + interp + ",tile" + tile), name
            assert ("horner" in name) == (rate == 11025), name  # 640 phases do not fit the table (b <= 512)
        got = out.download()
        diff = total = 0
        for s, g in zip(streams, got):
            ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, rate), new_rate, oracle.INTERP[interp])
            d, t = _check(g[0], ref.data[0])
            diff += d
            total += t
        assert diff <= max(2, total // 500), (diff, total)  # neighbouring floats: measured 1e-4 of the outputs two seconds into a stream
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


def test_wave_f64_horner_form_agrees(ctx, oracle, monkeypatch):
    B, N = _mods()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        s = pcm16(100000, 44100, 1, 7).tobytes()
        bt = B.Batch.upload(ctx, [s, s[:5000]])
        desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
        a = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32).download()
        assert "phase_regs" in ctx.last_kernel()[0]
        monkeypatch.setenv("AUKIT_F64_REGS", "0")
        t = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32).download()
        assert "phase_table" in ctx.last_kernel()[0]
        for x, y in zip(a, t):
            assert np.array_equal(x[0], y[0])   # same weights, same order of operations: the two are bit-identical
        monkeypatch.setenv("AUKIT_F64_HORNER", "1")
        b = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32).download()
        assert "horner" in ctx.last_kernel()[0]
        ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.CUBIC)
        for g in (a, b):
            d, t = _check(g[0][0], ref.data[0])
            assert d <= t // 500
        assert np.count_nonzero(a[0][0] != b[0][0]) <= len(a[0][0]) // 5000  # both forms on the same exact positions: they differ by ulps of fp64 only (measured 4 of 108 843)
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


def test_exact_math_levels_pick_their_kernels(ctx, oracle):
    B, N = _mods()
    s = pcm16(20000, 44100, 1, 0).tobytes()
    bt = B.Batch.upload(ctx, [s])
    desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
    ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.CUBIC).data[0]
    try:
        for level, prefix in ((0, "k_fast_wave<"), (1, "k_wave_f64<"), (2, "k_exact_wave<")):
            ctx.set_option(N.OPT_EXACT_MATH, level)
            out = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32).download()[0][0]
            assert ctx.last_kernel()[0].startswith(prefix), (level, ctx.last_kernel())
            assert rms(out, ref) <= 1e-6
            if level == 2:
                assert np.count_nonzero(out.astype(np.float32) != ref.astype(np.float32)) <= 1  # the reference's own operation order
            if level == 1:
                assert np.count_nonzero(out.astype(np.float32) != ref.astype(np.float32)) <= len(ref) // 500
        # F64 storage is the reference's operation order whatever the option says
        ctx.set_option(N.OPT_EXACT_MATH, 1)
        out = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F64).download()[0][0]
        assert ctx.last_kernel()[0].startswith("k_exact_wave<")
        assert np.max(np.abs(out - ref)) <= 1e-15
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate", [44100, 8000, 22050, 32000, 48000, 11025])
def test_wave_f64_stream_pcm_epilogue(ctx, oracle, rate, interp):
    """aukit.stream.pcm on 16-bit mono strings with AUKIT_OPT_EXACT_MATH = 1 and f32 storage: k_wave_f64 with the stream.pcm epilogue — the raw
    interpolated sample and the one before it in fp64, `ns = ls + lp_alpha (s - ls)`, the scale by 128 / 127 and the clamp in fp64, one rounding to
    f32 (aukit.lua:2397-2403).  Every stored f32 is the oracle's double rounded to f32 or its neighbour (chunk samples reach 128: one f32 ulp is
    7.6e-6 there, 6e-8 of the [-128, 127] scale), and the chunk plan is the exact path's."""
    B, N = _mods()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        lens = [rate * 3 + 11, rate + 1, 4097, 5, 700, 48001]
        streams = [pcm16(n, rate, 1, i).tobytes() for i, n in enumerate(lens)]
        streams.append(np.random.Generator(np.random.PCG64(6)).integers(-32768, 32768, 60000).astype(np.int16).tobytes())
        bt = B.Batch.upload(ctx, streams)
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, rate, 16, "signed"), interp, dtype=N.F32)
        name = ctx.last_kernel()[0]
        assert name.startswith("k_wave_f64<pcm_s16le_mono," + interp) and name.endswith("stream_pcm>"), name
        got = out.download()
        diff = total = 0
        for i, s in enumerate(streams):
            ref = oracle.stream_pcm(s, 16, oracle.SIGNED, 1, rate, False, False, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            g, r = got[i][0], ref.data[0]
            assert len(g) == len(r)
            if len(r):
                assert rms(g / 128, r / 128) <= 1e-6
                r32 = r.astype(np.float32)
                assert np.max(np.abs(g - r32.astype(np.float64))) <= 7.7e-6   # one f32 ulp below 128
                diff += int(np.count_nonzero(g.astype(np.float32) != r32))
                total += len(r)
        assert diff <= max(4, total // 300), (diff, total)
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("ulaw", [True, False])
@pytest.mark.parametrize("rate,new_rate", [(8000, 48000), (8000, 44100), (6000, 48000), (11025, 48000)])
def test_wave_coef_f64_g711_rounds_the_oracles_double(ctx, oracle, rate, new_rate, ulaw, interp):
    """BASELINE config 2a in the reference's arithmetic type: aukit.g711(d, ulaw, 1, rate):resample(new_rate, interp) with AUKIT_OPT_EXACT_MATH = 1 and
    f32 storage runs k_wave_coef_f64 (per-source-sample fp64 coefficients, fp64 Horner form, one rounding to f32) where the up-sampling factor
    exceeds ≈ 4.6; every stored f32 is the oracle's double rounded to f32 or its neighbour.  Other ratios keep the reference-order kernels."""
    B, N = _mods()
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        rng = np.random.Generator(np.random.PCG64(11))
        streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (rate * 2 + 11, 9000, 4097, 1, 2, 3, 5, 700, 171, 172, 1024)]
        bt = B.Batch.upload(ctx, streams)
        out = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_G711, 1, rate, ulaw=ulaw), new_rate, interp, dtype=N.F32)
        name = ctx.last_kernel()[0]
        if new_rate / rate > 4.7:
            assert name == "k_wave_coef_f64<g711_mono," + interp + ">", name
        else:
            assert name.startswith(("k_fast_wave_fmt<", "k_wave_coef_f64<")) and (name.endswith(",f64>") or name.startswith("k_wave_coef")), name   # the generic format kernel with doubles in its tables (fast_fmt.hip)
        got = out.download()
        diff = total = 0
        for s, g in zip(streams, got):
            ref = oracle.resample(oracle.g711(s, ulaw, 1, rate), new_rate, oracle.INTERP[interp])
            d, t = _check(g[0], ref.data[0])
            diff += d
            total += t
        assert diff <= max(2, total // (500 if name.startswith("k_wave_coef") else 100)), (diff, total)   # random bytes = full-scale noise: see test_every_other_format_in_fp64_arithmetic
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("kind", ["s16x2", "s16be", "s8", "u8", "u8x2", "s24x2", "s24be", "s32", "f32", "f32x2", "ulaw2", "alaw2"])
@pytest.mark.parametrize("rate,new_rate", [(44100, 48000), (22050, 48000), (48000, 44100)])
def test_every_other_format_in_fp64_arithmetic(ctx, oracle, kind, rate, new_rate, interp):
    """AUKIT_OPT_EXACT_MATH = 1 with f32 storage beyond 16-bit mono: k_fast_wave_fmt<..., f64> keeps doubles in its LDS tables (samples as correctly
    rounded double quotients, fp64 Horner form, one rounding to f32): every stored f32 is the oracle's double rounded to f32 or its neighbour."""
    B, N = _mods()
    spec = {"s16x2": (16, "signed", False, 2), "s16be": (16, "signed", True, 1), "s8": (8, "signed", False, 1), "u8": (8, "unsigned", False, 1), "u8x2": (8, "unsigned", False, 2),
            "s24x2": (24, "signed", False, 2), "s24be": (24, "signed", True, 1), "s32": (32, "signed", False, 1), "f32": (32, "float", False, 1), "f32x2": (32, "float", False, 2)}
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    try:
        rng = np.random.Generator(np.random.PCG64(19))
        if kind in ("ulaw2", "alaw2"):
            ch = 2
            streams = [rng.integers(0, 256, n * ch, dtype=np.uint8).tobytes() for n in (rate + 11, 4097, 1, 3, 700)]
            desc = B.make_desc(N.CODEC_G711, 2, rate, ulaw=kind == "ulaw2")
            dec = lambda s: oracle.g711(s, kind == "ulaw2", 2, rate)
        else:
            bits, dt, be, ch = spec[kind]
            if dt == "float":
                streams = [rng.uniform(-1, 1, n * ch).astype("<f4").tobytes() for n in (rate + 11, 4097, 1, 3, 700)]
            else:
                streams = [rng.integers(0, 256, n * ch * (bits // 8), dtype=np.uint8).tobytes() for n in (rate + 11, 4097, 1, 3, 700)]
            desc = B.make_desc(N.CODEC_PCM, ch, rate, bits, dt, big_endian=be)
            dec = lambda s: oracle.pcm(s, bits, oracle.DTYPE[dt], ch, rate, True, be)
        bt = B.Batch.upload(ctx, streams)
        out = B.decode_resample(ctx, bt, desc, new_rate, interp, dtype=N.F32)
        name = ctx.last_kernel()[0]
        if not (kind == "f32x2" and rate > new_rate):   # 8-byte frames, down-sampling, fp64 tables: beyond 64 KiB of LDS — the reference-order kernel
            assert name.startswith("k_fast_wave_fmt<") and name.endswith(",f64>"), name
        got = out.download()
        diff = total = 0
        for s, g in zip(streams, got):
            ref = oracle.resample(dec(s), new_rate, oracle.INTERP[interp])
            for c in range(ch):
                d, t = _check(g[c], ref.data[c])
                diff += d
                total += t
        # Neighbouring floats: the kernel's position is the exact rational, the reference's a rounded double (see the module docstring); on
        # full-scale NOISE (these inputs) the interpolated doubles differ by up to 1.4e-11 and 0.5 % of them straddle an f32 rounding boundary —
        # a numpy evaluation of the exact-position formula against the oracle shows the same 0.53 % (253 of 48 011 outputs, 24-bit, linear).
        assert diff <= max(2, total // 50), (diff, total)   # (uniform float noise, linear: 1.4 %)
    finally:
        ctx.set_option(N.OPT_EXACT_MATH, 0)
