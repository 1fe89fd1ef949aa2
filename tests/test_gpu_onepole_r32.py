"""k_rs_onepole<..., R32> (aukit_amd/csrc/flac_tail.hip): effects.lowpass on rows whose resample is owed runs its recurrence
d[i] = l + a * (d[i] - l) (aukit.lua:3586-3598) and the scan that carries it across a wave in f32 when the slope 1 - a is at most 1/2, in
fp64 otherwise.  Directed inputs for the f32 scan: full-scale steps (the state swings over the whole range, the scan's products are as large
as they get), the smallest alternation, the slope on either side of 1/2, a row long enough for a carried error to pile up if it could.
Asserted: which arithmetic ran (AUKIT_COUNTER_RECURRENCE_F32), the RMS bar of SURVEY 8d AND the largest single error."""
import numpy as np
import pytest

from util import pcm16, rms, tail_kernel

pytestmark = pytest.mark.gpu


def _mods():
    from aukit_amd import _native as N, batch as B
    return N, B


def _rows(n):
    t = np.arange(n)
    sq = np.where((t // 3000) % 2 == 0, 32767, -32768)                    # full-scale DC steps: the encoder slews to the rails and sits there
    alt = np.where(t % 2 == 0, 1, -1) * 1                                   # +-1 LSB alternation
    burst = np.where((t // 500) % 7 == 0, sq, (3000 * np.sin(t * 0.3)).astype(np.int64))  # rail-to-rail edges between quiet passages
    chirp = (32000 * np.sin(2 * np.pi * (50 + t * 0.2) * t / 22050)).astype(np.int64)      # every frequency up to the band edge at full scale
    return [np.asarray(x, dtype=np.int16) for x in (sq, alt, burst, chirp, pcm16(n, 22050, 3, 7))]


EDGE = 48000 * np.log(2) / (2 * np.pi)   # the cut-off at which the slope exp(-2 pi f / rate) is exactly 1/2: 5295.25 Hz


@pytest.mark.parametrize("freq,f32", [(EDGE + 0.5, True), (EDGE - 0.5, False), (11025.0, True), (20000.0, True), (23999.0, True), (3000.0, False)])
def test_r32_recurrence_directed(ctx, oracle, rs_kernel, freq, f32):
    N, B = _mods()
    streams = [oracle.gen_ima(r, 1, 512, 88) for r in _rows(1016 * 40)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
    for interp in ("cubic", "linear"):
        a = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32)
        assert ctx.last_kernel()[0] == "(resample deferred)"
        B.effect(ctx, a, "lowpass", float(freq))
        assert ctx.last_kernel()[0] == tail_kernel("lowpass", rs_kernel, interp == "cubic")
        assert ctx.counter(N.COUNTER_RECURRENCE_F32) == (1 if f32 else 0), freq
        got = a.download()
        for i, s in enumerate(streams):
            ref = oracle.fx_lowpass(oracle.resample(oracle.wav_adpcm(s, 512, 1, 22050), 48000, oracle.INTERP[interp]), float(freq)).data[0]
            err = got[i][0].astype(np.float64) - ref
            assert len(err) == len(ref)
            # f32 recurrence: a step rounds to 2^-24 of the state and is worth at most 1 / (1 - slope) <= 2 of itself in the end; with the f32 store
            # and the f32 interpolation that is a few 1e-7 at full scale.  The fp64 recurrence leaves the interpolation's and the store's rounding.
            assert rms(got[i][0], ref) <= 1e-6 and np.max(np.abs(err)) <= 6e-7, (freq, interp, i, rms(got[i][0], ref), np.max(np.abs(err)))


def test_r32_recurrence_thirty_minute_row(ctx, oracle, rs_kernel):
    """one row of 30 minutes (86.4 M outputs, 84 375 tiles chained through the carried state): the error at the end of the row is what it is at its
    start — the recurrence forgets (slope <= 1/2), nothing piles up along the chain"""
    N, B = _mods()
    n = 1016 * 39065   # 30 min at 22 050 Hz in whole IMA blocks
    rng = np.random.Generator(np.random.PCG64(31))
    t = np.arange(n)
    x = (20000 * np.sin(2 * np.pi * 440 * t / 22050) + rng.integers(-8000, 8001, n) + 4000 * np.sign(np.sin(t / 50000.0))).astype(np.int16)
    s = oracle.gen_ima(x, 1, 512, 88)
    a = B.decode_resample(ctx, B.Batch.upload(ctx, [s]), B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), 48000, "cubic", dtype=N.F32)
    B.effect(ctx, a, "lowpass", 11025.0)
    assert ctx.last_kernel()[0] == tail_kernel("lowpass", rs_kernel, True) and ctx.counter(N.COUNTER_RECURRENCE_F32) == 1
    got = a.download()[0][0]
    ref = oracle.fx_lowpass(oracle.resample(oracle.wav_adpcm(s, 512, 1, 22050), 48000, oracle.CUBIC), 11025.0).data[0]
    assert len(got) == len(ref) == int(n * 48000 // 22050)
    err = got.astype(np.float64) - ref
    q = len(err) // 4
    parts = [float(np.sqrt(np.mean(err[k * q:(k + 1) * q] ** 2))) for k in range(4)]
    assert max(parts) <= 1e-6 and np.max(np.abs(err)) <= 6e-7, (parts, np.max(np.abs(err)))
    assert parts[3] <= 1.25 * parts[0] + 1e-9, parts   # no growth along the row
