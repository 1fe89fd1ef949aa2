"""The guard band of the bit-exact floor()ed stream paths (aukit.lua:2909 stream.g711, :2827 stream.adpcm, :2727 stream.msadpcm).

Those kernels answer most outputs from an f32 evaluation (tier 1) and take it only when it lies further from an integer than a guard (5e-4,
MS-ADPCM 6e-4) — twice the error bound derived in their headers (2.3e-4 G.711, 2.5e-4 IMA, 3e-4 MS-ADPCM).  A derivation is not a measurement:
  * with AUKIT_OPT_COLLECT_STATS the calls run an audited instantiation that compares EVERY output's tier-1 value with tier 2's fp64 value and
    reports the largest difference (AUKIT_COUNTER_TIER1_ERR_NANO): asserted below the header's bound here, on noise, full-scale square waves,
    encoder-made and random-byte inputs, at several rates;
  * directed inputs drive the exact interpolant ONTO the band — within 1e-3 of an integer, on both sides, at the steepest slope the samples allow
    (exact rational arithmetic picks them) — and every output must still equal the oracle's."""
from fractions import Fraction

import numpy as np
import pytest

from tests.util import pcm16

pytestmark = pytest.mark.gpu

BOUND = {"g711": 2.3e-4, "ima": 2.5e-4, "msadpcm": 3.0e-4}


def _mods():
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    return N, B


def _audited(ctx, fn):
    N, _ = _mods()
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    try:
        out = fn()
        return out, ctx.counter(N.COUNTER_TIER1_ERR_NANO) * 1e-9, ctx.counter(N.COUNTER_TIER1_OUTPUTS)
    finally:
        ctx.set_option(N.OPT_COLLECT_STATS, 0)


def _ulaw_values():
    """the 256 µ-law code points as stream.g711 sees them (multiples of 1/64, aukit.lua:2880-2891): value -> byte"""
    out = {}
    for b in range(256):
        x = b ^ 0xFF
        m, e = x & 15, (x >> 4) & 7
        v = ((2 * m + 33) << e) - 33
        v = -v if (x & 0x80) else v
        out.setdefault(Fraction(v, 64), b)
    return out


@pytest.mark.parametrize("rate,interp", [(8000, "cubic"), (8000, "linear"), (44100, "cubic"), (22050, "cubic"), (11025, "linear"), (32000, "cubic")])
def test_g711_tier1_error_measured(ctx, oracle, rate, interp):
    N, B = _mods()
    rng = np.random.Generator(np.random.PCG64(rate + len(interp)))
    n = rate * 3
    streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes(),                                   # noise: every code point
               bytes(np.where((np.arange(n) // 3) % 2 == 0, 0x00, 0x80).astype(np.uint8)),         # full-scale square wave: the steepest slopes
               oracle.gen_g711(pcm16(n, rate, 2, 7), True)]                                         # an encoded signal
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_G711, 1, rate, ulaw=True)
    (out, ck), err, cnt = _audited(ctx, lambda: B.stream_decode(ctx, bt, desc, interp, dtype=N.I8))
    assert ctx.last_kernel()[0].startswith("k_floor_wave_g711"), ctx.last_kernel()
    assert cnt >= sum(len(s) for s in streams) * (48000 // rate) // 2 and 0 < err <= BOUND["g711"], (err, cnt)
    calls = -(-n // rate)
    for s, g in zip(streams, out.download()):
        ref = oracle.stream_g711(s, True, 1, rate, False, oracle.INTERP[interp], max_calls=calls)
        assert np.array_equal(g[0], ref.data[0])
    print(f"stream.g711 {rate} Hz {interp}: largest |tier 1 - tier 2| = {err:.3e} over {cnt} outputs (bound {BOUND['g711']:.1e}, guard 5e-4)")


@pytest.mark.parametrize("interp", ["cubic", "linear"])
def test_ima_and_msadpcm_tier1_error_measured(ctx, oracle, interp):
    N, B = _mods()
    rng = np.random.Generator(np.random.PCG64(91))
    # stream.adpcm: encoder-made blocks and random bytes (saturated predictors: the largest samples and slopes)
    ima = [oracle.gen_ima(pcm16(1016 * 40, 22050, 3, i), 1, 512, 88) for i in range(3)]
    rnd = rng.integers(0, 256, (40, 512), dtype=np.uint8)
    rnd[:, 2] = rng.integers(0, 89, 40)   # a valid header step index (:2799)
    ima.append(rnd.tobytes())
    for rate in (22050, 44100):
        bt = B.Batch.upload(ctx, ima)
        desc = B.make_desc(N.CODEC_ADPCM_WAV, 1, rate, block_align=512)
        (out, ck), err, cnt = _audited(ctx, lambda: B.stream_decode(ctx, bt, desc, interp, dtype=N.I8))
        assert ctx.last_kernel()[0] == "k_ima_stream_f32"
        assert cnt > 100000 and 0 < err <= BOUND["ima"], (err, cnt)
        for s, g in zip(ima, out.download()):
            assert np.array_equal(g[0], oracle.stream_adpcm(s, 512, 1, rate, False, oracle.INTERP[interp]).data[0])
        print(f"stream.adpcm {rate} Hz {interp}: largest |tier 1 - tier 2| = {err:.3e} over {cnt} outputs (bound {BOUND['ima']:.1e}, guard 5e-4)")
    ms = [oracle.gen_msadpcm(pcm16(2036 * 30, 44100, 3, 10 + i), 1, 1024) for i in range(3)]
    for mono_mix, ch in ((False, 1),):
        bt = B.Batch.upload(ctx, ms)
        desc = B.make_desc(N.CODEC_MSADPCM, ch, 44100, block_align=1024)
        (out, ck), err, cnt = _audited(ctx, lambda: B.stream_decode(ctx, bt, desc, interp, dtype=N.I8))
        assert ctx.last_kernel()[0] == "k_ms_wave"
        assert cnt > 100000 and err <= BOUND["msadpcm"], (err, cnt)
        for s, g in zip(ms, out.download()):
            assert np.array_equal(g[0], oracle.stream_msadpcm(s, 1024, ch, 44100, False, None, oracle.INTERP[interp]).data[0])
        print(f"stream.msadpcm 44100 Hz {interp}: largest |tier 1 - tier 2| = {err:.3e} over {cnt} outputs (bound {BOUND['msadpcm']:.1e}, guard 6e-4)")


def _cubic(p0, p1, p2, p3, fx):   # interpolate.cubic (:261-266) in exact rational arithmetic
    return ((-p0 / 2 + 3 * p1 / 2 - 3 * p2 / 2 + p3 / 2) * fx ** 3 + (p0 - 5 * p1 / 2 + 2 * p2 - p3 / 2) * fx ** 2 + (-p0 / 2 + p2 / 2) * fx + p1)


def test_g711_inputs_on_the_guard_band(ctx, oracle):
    """44.1 kHz µ-law (b = 160 phases: interpolants on a grid of 1 / 1 310 720 under the cubic): four-sample windows at the extremes of the code
    (slopes of up to 252 per sample) whose EXACT interpolant lies within 1e-3 of an integer — below, above and exactly on it — laid end to end."""
    N, B = _mods()
    vals = _ulaw_values()
    big = sorted(vals)[:6] + sorted(vals)[-6:]          # the twelve largest magnitudes
    rng = np.random.Generator(np.random.PCG64(3))
    keep, below, above, on = [], 0, 0, 0
    phases = [Fraction(r, 160) for r in range(1, 160)]
    tries = 0
    while len(keep) < 400 and tries < 200000:
        tries += 1
        w = [big[int(i)] for i in rng.integers(0, len(big), 4)]
        if abs(w[2] - w[1]) < 200:      # the steep ones only
            continue
        for fx in phases[::7]:
            v = _cubic(*w, fx)
            d = v - round(v)
            if abs(d) < Fraction(1, 1000):
                keep.append(w)
                below += d < 0; above += d > 0; on += d == 0
                break
    assert len(keep) >= 100 and below and above, (len(keep), below, above, on)
    samples = [x for w in keep for x in (w + w)]        # every window twice: its taps meet every phase somewhere
    stream = bytes(vals[x] for x in samples)
    bt = B.Batch.upload(ctx, [stream])
    for interp in ("cubic", "linear"):
        (out, ck), err, cnt = _audited(ctx, lambda: B.stream_decode(ctx, bt, B.make_desc(N.CODEC_G711, 1, 44100, ulaw=True), interp, dtype=N.I8))
        assert err <= BOUND["g711"], err
        ref = oracle.stream_g711(stream, True, 1, 44100, False, oracle.INTERP[interp], max_calls=1)
        assert np.array_equal(out.download()[0][0], ref.data[0]), interp
        # and with the plain (unaudited) instantiation, the one that ships
        out2, _ = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_G711, 1, 44100, ulaw=True), interp, dtype=N.I8)
        assert np.array_equal(out2.download()[0][0], ref.data[0]), interp


def test_adpcm_blocks_on_the_guard_band(ctx, oracle):
    """stream.adpcm / stream.msadpcm cannot be handed samples, only blocks: random blocks are searched (fp64 on the oracle's decoded samples) for the
    ones whose interpolants come closest to integers at steep slopes; those blocks, side by side, must decode bit for bit."""
    N, B = _mods()
    rng = np.random.Generator(np.random.PCG64(17))
    ratio = 48000 / 22050
    picked = []
    for _ in range(600):
        blk = rng.integers(0, 256, 512, dtype=np.uint8)
        blk[2] = rng.integers(0, 89)
        d = oracle.wav_adpcm(blk.tobytes(), 512, 1, 22050).data[0] * 32768.0     # the predictors (:1255 scale undone: close enough to rank blocks)
        d = np.where(d < 0, d / 128.0, d / 127.0)                                 # stream scale (:2812)
        x = np.arange(1, int(len(d) * ratio) - 4) / ratio
        k = np.floor(x).astype(int)
        ok = (k >= 1) & (k + 2 < len(d))
        k, fx = k[ok], (x - np.floor(x))[ok]
        p0, p1, p2, p3 = d[k - 1], d[k], d[k + 1], d[k + 2]
        v = (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1
        near = (np.abs(v - np.round(v)) < 1e-3) & (np.abs(p2 - p1) > 50) & (fx > 0)
        if near.sum() >= 2:
            picked.append(blk.tobytes())
    assert len(picked) >= 20, len(picked)
    stream = b"".join(picked)
    bt = B.Batch.upload(ctx, [stream])
    for interp in ("cubic", "linear"):
        (out, ck), err, cnt = _audited(ctx, lambda: B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), interp, dtype=N.I8))
        assert err <= BOUND["ima"], err
        ref = oracle.stream_adpcm(stream, 512, 1, 22050, False, oracle.INTERP[interp])
        assert np.array_equal(out.download()[0][0], ref.data[0]), interp
        out2, _ = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), interp, dtype=N.I8)
        assert np.array_equal(out2.download()[0][0], ref.data[0]), interp
