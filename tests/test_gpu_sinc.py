"""GPU: interpolate.sinc (aukit.lua:267-281).  sin() comes from the device libm, so parity is tolerance-level (≤ 1e-12 on
[-1,1] data); floored stream outputs may flip by one unit where the interpolated value is an integer to within 1e-13."""
import numpy as np
import pytest

from tests.util import pcm16, signal

pytestmark = pytest.mark.gpu


def test_audio_resample_sinc(ctx, oracle):
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    a = [[signal(6000, 22050, 9, 0)], [signal(333, 22050, 9, 1)]]
    ab = B.AudioBatch.upload(ctx, a, 22050, dtype=N.F64)
    for new_rate in (48000, 16000):
        got = B.resample(ctx, ab, new_rate, "sinc").download()
        for s in range(2):
            ref = oracle.resample(oracle.Audio(a[s], 22050), new_rate, oracle.SINC)
            assert len(got[s][0]) == len(ref.data[0])
            assert np.max(np.abs(got[s][0] - ref.data[0])) <= 1e-12
    ctx.set_sinc_window(30)  # LuaJIT hosts use a ±30 window (aukit.lua:129)
    oracle.set_sinc_window(30)
    try:
        got = B.resample(ctx, ab, 48000, "sinc").download()
        ref = oracle.resample(oracle.Audio(a[0], 22050), 48000, oracle.SINC)
        assert np.max(np.abs(got[0][0] - ref.data[0])) <= 1e-12
    finally:
        ctx.set_sinc_window(10)
        oracle.set_sinc_window(10)


def test_fused_and_stream_sinc(ctx, oracle):
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    s = pcm16(9000, 44100, 1, 2).tobytes()
    bt = B.Batch.upload(ctx, [s])
    got = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed"), 48000, "sinc", dtype=N.F64).download()[0][0]
    ref = oracle.resample(oracle.pcm(s, 16, oracle.SIGNED, 1, 44100), 48000, oracle.SINC)
    assert np.max(np.abs(got - ref.data[0])) <= 1e-12
    g = oracle.gen_g711(pcm16(12000, 8000, 2, 3), True)
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, [g]), B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True), "sinc", dtype=N.I8)
    ref = oracle.stream_g711(g, True, 1, 8000, False, oracle.SINC)
    d = out.download()[0][0] - ref.data[0]
    assert np.max(np.abs(d)) <= 1 and np.count_nonzero(d) <= 2


@pytest.mark.parametrize("dt", ["F64", "F32"])
def test_stream_flac_and_qoa_sinc(ctx, oracle, dt):
    """aukit.defaultInterpolation = "sinc" is legal for aukit.stream.flac (aukit.lua:3156) and aukit.stream.qoa (:3252): every block's table holds
    numbers at indices -1 .. #block (the two samples kept from the block before at -1 and 0), nil elsewhere, and interpolate.sinc skips the
    nils (:273).  Round 3 refused these with AUKIT_E_UNSUPPORTED (missing item 2)."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    dtype = getattr(N, dt)
    tol = 1e-9 if dt == "F64" else 2e-5   # [-128, 127] scale: sin() comes from the device libm; f32 storage adds its rounding
    pcms = [np.stack([pcm16(n, 44100, 6, 2 * i + c) for c in range(2)], 1) for i, n in enumerate((9000, 2500))]
    fl = [oracle.gen_flac(p.astype(np.int64).ravel(), 2, 16, 44100, 1152) for p in pcms]
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, fl), B.make_desc(N.CODEC_FLAC), "sinc", dtype=dtype)
    got = out.download()
    for s, f in enumerate(fl):
        ref = oracle.stream_flac(f, oracle.SINC)
        for c in range(2):
            assert len(got[s][c]) == len(ref.data[c])
            assert np.max(np.abs(got[s][c] - ref.data[c]), initial=0) <= tol, (s, c)
    qo = [oracle.gen_qoa(p.ravel(), 2, 44100) + b"\0" * 8 for p in pcms]
    for mono in (False, True):
        out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, qo), B.make_desc(N.CODEC_QOA, 2, 44100), "sinc", mono=mono, dtype=dtype)
        got = out.download()
        for s, q in enumerate(qo):
            ref = oracle.stream_qoa(q, mono, oracle.SINC)
            for c in range(1 if mono else 2):
                assert len(got[s][c]) == len(ref.data[c])
                assert np.max(np.abs(got[s][c] - ref.data[c]), initial=0) <= tol, (mono, s, c)


@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("rate", [44100, 22050])
def test_stream_msadpcm_stereo_sinc(ctx, oracle, mono, rate):
    """aukit.stream.msadpcm with two channels keeps the block before at table indices -N .. -1 and leaves index 0 nil (aukit.lua:2640-2643; `lastL`
    outlives the iterator call): interpolate.sinc is the one interpolation that reaches there.  Floored int8 outputs: sin() is the device's libm,
    so an output whose value lies within 1e-12 of an integer may land on the other side — counted, not tolerated in bulk."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    ba = 256
    spb = (ba - 14) + 2   # frames a stereo block decodes to
    streams = [oracle.gen_msadpcm(np.stack([pcm16(spb * nb, rate, 3, 60 + 2 * i + c) for c in range(2)], 1).ravel(), 2, ba) for i, nb in enumerate((40, 3, 1))]
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, streams), B.make_desc(N.CODEC_MSADPCM, 2, rate, block_align=ba), "sinc", mono=mono, dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_msadpcm(s, ba, 2, rate, mono, None, oracle.SINC)
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        for c in range(ref.channels):
            d = got[i][c].astype(np.int64) - ref.data[c].astype(np.int64)
            assert np.max(np.abs(d), initial=0) <= 1 and np.count_nonzero(d) <= 2, (i, c, int(np.count_nonzero(d)))
    # the history matters: the second block decoded as a stream's first block (no `lastL`) gives different outputs near its start
    two = oracle.stream_msadpcm(streams[0][:2 * ba], ba, 2, rate, mono, None, oracle.SINC)
    alone = oracle.stream_msadpcm(streams[0][ba:2 * ba], ba, 2, rate, mono, None, oracle.SINC)
    n1 = len(alone.data[0])
    assert len(two.data[0]) == 2 * n1 and not np.array_equal(two.data[0][n1:], alone.data[0])
