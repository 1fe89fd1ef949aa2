"""k_rsp (aukit_amd/csrc/rs_periodic.hip): the tile chain of `resample owed -> effects.lowpass / highpass [-> Audio:mono]` for int16 rows at 44.1 / 22.05 kHz
-> 48 kHz, weights and tap offsets in registers (aukit.lua:648-680 interpolate.cubic / Audio:resample, :3586-3618 the filters, :682-687 Audio:mono).
What the wider modules do not pin down: every row length around the tile sizes (320 outputs a sub-tile, 640 / 1280 a tile), runs of tiles with a warm-up
forced on short rows, frame-by-frame FLAC rows with short frames, and the two kernels against each other (same interpolation bit for bit, another scan
decomposition: a few f32 ulps)."""
import numpy as np
import pytest

from util import pcm16, rms, tail_kernel

pytestmark = pytest.mark.gpu


def _mods():
    from aukit_amd import _native as N, batch as B
    return N, B


def _ima(oracle, n, seed):
    nb = max(-(-n // 1016), 1)
    return oracle.gen_ima(pcm16(1016 * nb, 22050, 3, seed), 1, 512, 40)


@pytest.mark.parametrize("which,freq", [("lowpass", 11025.0), ("lowpass", 2500.0), ("highpass", 20.0)])
def test_row_lengths_around_the_tiles_22050(ctx, oracle, monkeypatch, which, freq):
    """IMA blocks of 1016 samples -> 2211 outputs a block: 1 .. 9 blocks cross the 1280-output tile and its 320-output sub-tiles at every phase"""
    N, B = _mods()
    streams = [_ima(oracle, 1016 * k, k) for k in range(1, 10)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
    got = {}
    for kern in ("default", "generic"):
        if kern == "generic":
            monkeypatch.setenv("AUKIT_RS_GENERIC", "1")
        a = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32)
        B.effect(ctx, a, which, freq)
        assert ctx.last_kernel()[0] == tail_kernel(which, kern, True)
        got[kern] = a.download()
    monkeypatch.delenv("AUKIT_RS_GENERIC")
    for i, s in enumerate(streams):
        ref = oracle.resample(oracle.wav_adpcm(s, 512, 1, 22050), 48000, oracle.CUBIC)
        ref = (oracle.fx_lowpass if which == "lowpass" else oracle.fx_highpass)(ref, freq).data[0]
        g = got["default"][i][0]
        assert len(g) == len(ref)
        assert rms(g, ref) <= 1e-6 and np.max(np.abs(g.astype(np.float64) - ref)) <= 6e-7, (i, rms(g, ref))
        assert np.max(np.abs(g - got["generic"][i][0])) <= 3e-7, i   # the two kernels: a few ulps at full scale


@pytest.mark.parametrize("rate,bs", [(44100, 4096), (44100, 1152), (22050, 1152), (22050, 576)])
@pytest.mark.parametrize("segs", [1, 3])
def test_flac_rows_frame_by_frame(ctx, oracle, monkeypatch, rate, bs, segs):
    """FLAC's int16 finals read where the decoder left them, frame by frame (a window lies in one frame or two: blocksizes down to the window's own length),
    stereo -> highpass -> normalize -> mono in one pass (NW = 2) and per channel (NW = 1), with runs of tiles forced (AUKIT_RS_SEGS: a run warms up
    over the tiles before it)"""
    N, B = _mods()
    lens = (rate * 3 + 123, bs * 7 + 1, bs * 2, 700, 1)
    streams = []
    for i, n in enumerate(lens):
        p = np.stack([pcm16(n, rate, 5, 2 * i + c) for c in range(2)], 1).astype(np.int64)
        streams.append(oracle.gen_flac(p.ravel(), 2, 16, rate, bs))
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_FLAC)
    monkeypatch.setenv("AUKIT_RS_SEGS", str(segs))
    a = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32)
    B.effect(ctx, a, "highpass", 20.0)
    m = B.mono(ctx, a)
    name = ctx.last_kernel()[0]
    assert name in ("k_rsp<highpass,mono>", "k_rs_onepole<highpass,mono>"), name   # (k_rsp unless a frame is shorter than its window: 22.05 kHz at 576 is not)
    if not (rate == 22050 and bs == 576):
        assert name == "k_rsp<highpass,mono>", name
    mono = m.download()
    rows = a.download()   # (the stereo rows themselves: the filter paid per channel)
    for i, s in enumerate(streams):
        ref = oracle.fx_highpass(oracle.resample(oracle.flac(s), 48000, oracle.CUBIC), 20.0)
        rm = oracle.mono(ref).data[0]
        assert len(mono[i][0]) == len(rm)
        # (a forced run on a short row starts from a state 40 halvings old: 1e-12 of full scale)
        assert rms(mono[i][0], rm) <= 1e-6 and np.max(np.abs(mono[i][0].astype(np.float64) - rm), initial=0) <= 1e-6, (i, rms(mono[i][0], rm))
        for c in range(2):
            assert rms(rows[i][c], ref.data[c]) <= 1e-6, (i, c)


def test_shapes_k_rsp_leaves_to_the_general_kernel(ctx, oracle):
    """linear interpolation, 8 kHz (320 outputs do not advance it by whole samples at 5 outputs a lane ... they do: 1 / 6 — but its tap pattern is not built),
    32 kHz DFPWM (int8 rows): k_rs_onepole as before"""
    N, B = _mods()
    s = _ima(oracle, 5000, 3)
    bt = B.Batch.upload(ctx, [s])
    a = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), 48000, "linear", dtype=N.F32)
    B.effect(ctx, a, "lowpass", 11025.0)
    assert ctx.last_kernel()[0] == "k_rs_onepole<lowpass>"
    a = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 8000, block_align=512), 48000, "cubic", dtype=N.F32)
    B.effect(ctx, a, "lowpass", 3000.0)
    assert ctx.last_kernel()[0] == "k_rs_onepole<lowpass>"
    ref = oracle.fx_lowpass(oracle.resample(oracle.wav_adpcm(s, 512, 1, 8000), 48000, oracle.CUBIC), 3000.0).data[0]
    assert rms(a.download()[0][0], ref) <= 1e-6
