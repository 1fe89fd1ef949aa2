"""GPU parity for the sequential codecs (DFPWM, IMA / MS ADPCM, QOA, MDFPWM) vs the CPU oracle, through the C ABI.

Integer decode stages and floored stream outputs are compared bit-exactly; DFPWM re-encoded bytes bit-exactly.
"""
import numpy as np
import pytest

from tests.util import pcm16, rms, signal

pytestmark = pytest.mark.gpu


def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


def _dfpwm_bytes(oracle, n, cfg, stream):
    return oracle.dfpwm_encode(np.round(signal(n, 48000, cfg, stream) * 100))


# ---------------------------------------------------------------- DFPWM
@pytest.mark.parametrize("ch", [1, 2])
def test_dfpwm_decode(ctx, oracle, ch):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(3))
    streams = [_dfpwm_bytes(oracle, 48000 * 2, 4, 0), _dfpwm_bytes(oracle, 6000 * 8, 4, 1), _dfpwm_bytes(oracle, 6001 * 8 * 2, 4, 2),
               rng.integers(0, 256, 12002, dtype=np.uint8).tobytes(), b"\x55" * 2]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_DFPWM, ch, 48000)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.dfpwm(s, ch, 48000)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c])
    res = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_DFPWM, ch, 32000), 48000, "cubic", dtype=N.F64).download()
    ref = oracle.resample(oracle.dfpwm(streams[0], ch, 32000), 48000, oracle.CUBIC)
    assert np.max(np.abs(res[0][0] - ref.data[0])) <= 1e-15


def test_dfpwm_uneven_channels_is_an_error(ctx):
    B, N = _B(), _N()
    bt = B.Batch.upload(ctx, [b"\x55" * 7])  # 56 samples over 3 channels
    with pytest.raises(N.AukitError) as e:
        B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 3, 48000))
    assert "uneven amount of data per channel" in str(e.value)


@pytest.mark.parametrize("interleaved", [True, False])
def test_dfpwm_encode_bit_exact(ctx, oracle, interleaved):
    B, N = _B(), _N()
    a = [[signal(n, 48000, 4, 10 * i + c) * 0.9 for c in range(2)] for i, n in enumerate((48000, 1001, 8, 3))]
    ab = B.AudioBatch.upload(ctx, a, 48000, dtype=N.F64)
    got = B.dfpwm_encode(ctx, ab, interleaved).download()
    for s in range(len(a)):
        assert got[s] == oracle.audio_dfpwm(oracle.Audio(a[s], 48000), interleaved)


def test_dfpwm_saturating_inputs(ctx, oracle):
    """The corners of the step functions (dfpwm_dev.h): charge pinned at 127 / -128 (the encoder's `v == charge and v == 127`
    clause, the nudge with diff == 0), strength at its floor and at its ceiling, long runs and strict alternation — decoder, encoder
    and the fused transcode against the oracle, bit for bit."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(99))
    pats = [b"\xff" * 12000, b"\x00" * 12000, b"\xaa" * 12000, b"\x55" * 12004, b"\xff" * 3000 + b"\x00" * 3000 + b"\xaa" * 6000,
            bytes(rng.integers(0, 256, 12000, dtype=np.uint8)), b"\x0f" * 12000, b"\xfe\x01" * 6000]
    bt = B.Batch.upload(ctx, pats)
    for ch in (1, 2):
        got = B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, ch, 48000), dtype=N.F64).download()
        for s, g in zip(pats, got):
            ref = oracle.dfpwm(s, ch, 48000)
            for c in range(ch):
                assert np.array_equal(g[c], ref.data[c]), (s[:2], ch, c)
    fused = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    for s, f in zip(pats, fused):
        assert f == oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(s, 2, 48000)), True), s[:2]
    # the encoder on signals that sit on the rails, step between them, and dither around them
    n = 48000
    sq = np.where((np.arange(n) // 500) % 2 == 0, 1.0, -1.0)
    sigs = [np.full(n, 1.0), np.full(n, -1.0), sq, np.clip(sq + rng.uniform(-0.02, 0.02, n), -1, 1), np.zeros(n), np.full(n, 126 / 127), np.full(n, -127 / 128)]
    ab = B.AudioBatch.upload(ctx, [[x] for x in sigs], 48000, dtype=N.F64)
    enc = B.dfpwm_encode(ctx, ab, True).download()
    for x, e in zip(sigs, enc):
        assert e == oracle.audio_dfpwm(oracle.Audio([x], 48000), True)


def test_dfpwm_encode_out_of_range_raises(ctx):
    B, N = _B(), _N()
    ab = B.AudioBatch.upload(ctx, [[np.array([0.0, 0.5, 1.5, 0.0])]], 48000, dtype=N.F64)
    with pytest.raises(N.AukitError):
        B.dfpwm_encode(ctx, ab, True)


def test_config4_pipeline_and_fused_transcode(ctx, oracle, monkeypatch):
    """BASELINE config 4: a = aukit.dfpwm(d, 2, 48000); m = a:mono(); out = m:dfpwm() — unfused and fused, bit-exact."""
    B, N = _B(), _N()
    streams = []
    for i, n in enumerate((120000, 24000, 6000, 6002)):
        l, r = np.round(signal(n * 4, 48000, 4, 2 * i) * 100), np.round(signal(n * 4, 48000, 4, 2 * i + 1) * 90)
        streams.append(oracle.dfpwm_encode(np.stack([l, r], 1).ravel()))
    bt = B.Batch.upload(ctx, streams)
    au = B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), dtype=N.F64)
    unfused = B.dfpwm_encode(ctx, B.mono(ctx, au), True).download()
    fused = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    for s, u, f in zip(streams, unfused, fused):
        ref = oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(s, 2, 48000)), True)
        assert u == ref
        assert f == ref
    assert len(fused[0]) == 60010  # Q10: 120 000 B → 960 152 samples → 480 076 mono samples → 60 010 B
    # four short streams (16 seconds together): chunk-parallel exact decode + the exact parallel encoder — a few short streams are theirs (dfx_run)
    assert ctx.last_kernel()[0] == "k_df_chunks+k_dfe_*"
    monkeypatch.setenv("AUKIT_DFX_FEW", "0")   # ... unless told otherwise: a lane per (stream, chunk) decodes, mixes and encodes (dfpwm_spec.hip; tests/test_gpu_dfpwm_spec.py)
    assert B.dfpwm_transcode_mono(ctx, bt, 2).download() == fused
    assert ctx.last_kernel()[0] == "k_dfx_chunks"
    monkeypatch.delenv("AUKIT_DFX_FEW")
    monkeypatch.setenv("AUKIT_DFPWM_NOSPEC", "1")
    assert B.dfpwm_transcode_mono(ctx, bt, 2).download() == fused
    assert ctx.last_kernel()[0] == "k_df_chunks+k_dfe_*"
    monkeypatch.delenv("AUKIT_DFPWM_NOSPEC")
    # the same bytes through every schedule: one lane per stream; 2-byte blocks (a warm-up of 16 steps: most recorded start states
    # are wrong and the verify pass redoes the chunks); chunk and Q10 slice boundaries in odd positions
    for env in ({"AUKIT_DFPWM_SERIAL": "1"}, {"AUKIT_DFPWM_BLOCK": "2", "AUKIT_DFPWM_CHUNKS": "1000"}, {"AUKIT_DFPWM_BLOCK": "6002", "AUKIT_DFPWM_CHUNKS": "50"},
                {"AUKIT_DFPWM_BLOCK": "250", "AUKIT_DFPWM_CHUNKS": "7"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        again = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        dec = B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), dtype=N.F64).download()
        for k in env:
            monkeypatch.delenv(k)
        assert again == fused, env
        for x, y in zip(dec, au.download()):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), env
    # the one-launch transcode of large batches (k_df_fused: decoder waves on a ticket counter, one encoder wave per 64 streams following
    # them chunk by chunk), forced onto this small batch: default shape; 16-byte warm-up blocks (most recorded start states are wrong and
    # the encoder lanes decode their chunks again); odd chunk sizes; one chunk per block
    for env in ({}, {"AUKIT_DFPWM_BLOCK": "16", "AUKIT_DFPWM_CHUNKS": "1000"}, {"AUKIT_DFPWM_BLOCK": "6000", "AUKIT_DFPWM_CHUNKS": "50"},
                {"AUKIT_DFPWM_BLOCK": "250", "AUKIT_DFPWM_CHUNKS": "7"}, {"AUKIT_DFPWM_BLOCK": "48", "AUKIT_DFPWM_CHUNKS": "100000"}):
        monkeypatch.setenv("AUKIT_DFPWM_FUSED", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        again = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        assert ctx.last_kernel()[0] == "k_df_fused"
        monkeypatch.delenv("AUKIT_DFPWM_FUSED")
        for k in env:
            monkeypatch.delenv(k)
        assert again == fused, env
    monkeypatch.setenv("AUKIT_DFPWM_SERIAL", "1")
    B.dfpwm_transcode_mono(ctx, bt, 2)
    assert ctx.last_kernel()[0] == "k_dfpwm_transcode_stereo"
    monkeypatch.delenv("AUKIT_DFPWM_SERIAL")
    # mono and 3-channel DFPWM through the parallel loader
    for ch in (1, 3):
        sx = oracle.dfpwm_encode(np.round(signal(47976 if ch == 3 else 100000, 48000, 4, 40 + ch) * 100))  # 5997 B: fed bytes * 8 divisible by 3
        got = B.decode(ctx, B.Batch.upload(ctx, [sx]), B.make_desc(N.CODEC_DFPWM, ch, 48000), dtype=N.F64).download()[0]
        refx = oracle.dfpwm(sx, ch, 48000)
        for c in range(ch):
            assert np.array_equal(got[c], refx.data[c]), (ch, c)
    # a batch whose streams are not 16-byte aligned takes the generic kernel: same bytes
    bt2 = B.Batch.upload(ctx, [b"\x5a"] + streams)
    monkeypatch.setenv("AUKIT_DFPWM_SERIAL", "1")
    fused2 = B.dfpwm_transcode_mono(ctx, bt2, 2).download()
    assert ctx.last_kernel()[0] == "k_dfpwm_transcode_mono"
    monkeypatch.delenv("AUKIT_DFPWM_SERIAL")
    assert B.dfpwm_transcode_mono(ctx, bt2, 2).download() == fused2
    assert fused2[1:] == fused


def test_fused_transcode_ragged_groups(ctx, oracle, monkeypatch):
    """k_df_fused on 150 streams (three groups of 64, the last one partial) of six different lengths, empty and one-slice streams among
    them: the bytes of the one-lane-per-stream kernel and of the oracle."""
    B = _B()
    base = []
    for i, n in enumerate((30000, 0, 6000, 18016, 6001, 12345 * 2)):
        l, r = np.round(signal(n * 4, 48000, 4, 60 + 2 * i) * 100), np.round(signal(n * 4, 48000, 4, 61 + 2 * i) * 90)
        base.append(oracle.dfpwm_encode(np.stack([l, r], 1).ravel()) if n else b"")
    bt = B.Batch.upload(ctx, [base[(i * 5 + i // 7) % 6] for i in range(150)])
    monkeypatch.setenv("AUKIT_DFPWM_SERIAL", "1")
    want = B.dfpwm_transcode_mono(ctx, bt, 2).download()
    monkeypatch.delenv("AUKIT_DFPWM_SERIAL")
    # (the last one: sub-batches of one 64-stream group, the way batches beyond one group per CU are cut)
    for env in ({}, {"AUKIT_DFPWM_BLOCK": "32", "AUKIT_DFPWM_CHUNKS": "40"}, {"AUKIT_DFPWM_FUSED_GROUPS": "1"}, {"AUKIT_DFPWM_FUSED_GROUPS": "2", "AUKIT_DFPWM_BLOCK": "64"}):
        monkeypatch.setenv("AUKIT_DFPWM_FUSED", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = B.dfpwm_transcode_mono(ctx, bt, 2).download()
        assert ctx.last_kernel()[0] == "k_df_fused"
        monkeypatch.delenv("AUKIT_DFPWM_FUSED")
        for k in env:
            monkeypatch.delenv(k)
        assert got == want, env
    for c in range(6):
        i = [(i * 5 + i // 7) % 6 for i in range(150)].index(c)
        assert want[i] == (oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(base[c], 2, 48000)), True) if base[c] else b""), c


@pytest.mark.parametrize("ch,mono,rate", [(1, False, 48000), (2, False, 48000), (2, True, 48000), (1, False, 24000), (2, True, 32000)])
def test_stream_dfpwm(ctx, oracle, ch, mono, rate):
    B, N = _B(), _N()
    streams = [_dfpwm_bytes(oracle, 48000 * 2 + 16, 4, 3), _dfpwm_bytes(oracle, 6000 * 8 * ch, 4, 4), b"\xaa" * 13]
    bt = B.Batch.upload(ctx, streams)
    for interp in ("linear", "cubic"):
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, ch, rate), interp, mono=mono, dtype=N.F64)
        got = out.download()
        for i, s in enumerate(streams):
            ref = oracle.stream_dfpwm(s, rate, ch, mono, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks
            assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            assert np.array_equal(ck.pos[i][:ref.nchunks], ref.chunk_pos)
            for c in range(ref.channels):
                assert np.max(np.abs(got[i][c] - ref.data[c]), initial=0) <= 1e-13, (i, c)


@pytest.mark.parametrize("ch,mono,rate", [(1, False, 24000), (1, False, 32000), (2, False, 32000), (2, True, 32000), (1, False, 44100), (2, False, 22050), (3, True, 8000), (2, False, 24000)])
def test_stream_dfpwm_other_rates_on_the_wave_kernel(ctx, oracle, monkeypatch, ch, mono, rate):
    """stream.dfpwm at rates other than 48 kHz with F32 storage (aukit.lua:2471-2491; round 4): k_fast_wave_dfpwm against the oracle (tolerance stage:
    1e-6 RMS of the [-1, 1] scale = 1.28e-4 on these int8-range values) and against the reference-order kernel, chunk table unchanged."""
    B, N = _B(), _N()
    streams = [_dfpwm_bytes(oracle, 48000 * 2 + 16, 4, 3), _dfpwm_bytes(oracle, 6000 * 8 * ch, 4, 4), b"\xaa" * 13, _dfpwm_bytes(oracle, 6000 * 8 * ch + 8, 4, 5)]
    bt = B.Batch.upload(ctx, streams)
    for interp in ("linear", "cubic"):
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, ch, rate), interp, mono=mono, dtype=N.F32)
        assert ctx.last_kernel()[0].startswith("k_fast_wave_dfpwm"), ctx.last_kernel()
        got = out.download()
        monkeypatch.setenv("AUKIT_DFPWM_NO_WAVE", "1")
        out2, ck2 = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, ch, rate), interp, mono=mono, dtype=N.F32)
        monkeypatch.delenv("AUKIT_DFPWM_NO_WAVE")
        assert ctx.last_kernel()[0].startswith("k_resample<"), ctx.last_kernel()
        got2 = out2.download()
        for i, s in enumerate(streams):
            ref = oracle.stream_dfpwm(s, rate, ch, mono, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks
            assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            assert np.array_equal(ck.pos[i][:ref.nchunks], ref.chunk_pos)
            for c in range(ref.channels):
                assert len(got[i][c]) == len(ref.data[c])
                if len(ref.data[c]):
                    assert np.sqrt(np.mean((got[i][c] - ref.data[c]) ** 2)) <= 1.28e-4, (i, c)
                    assert np.max(np.abs(got[i][c] - ref.data[c])) <= 1e-3, (i, c)
                    assert np.max(np.abs(got2[i][c] - ref.data[c])) <= 2e-5, (i, c)   # (the reference-order kernel, rounded to f32)


def test_mdfpwm(ctx, oracle, monkeypatch):
    B, N = _B(), _N()
    e1, e2 = _dfpwm_bytes(oracle, 48000 * 3, 4, 5), _dfpwm_bytes(oracle, 48000 * 3, 4, 6)
    md = oracle.gen_mdfpwm(e1, e2, b"artist", b"title", b"album")
    bt = B.Batch.upload(ctx, [md, md[:len(md) - 12000]])
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_MDFPWM), dtype=N.F64).download()
    ref = oracle.mdfpwm(md)
    for c in range(2):
        assert np.array_equal(got[0][c], ref.data[c])
    # decoderL / decoderR go through the chunk-parallel exact decoder (run 6000, stride 12000): every schedule gives the same rows
    for env in ({"AUKIT_DFPWM_BLOCK": "2", "AUKIT_DFPWM_CHUNKS": "500"}, {"AUKIT_DFPWM_BLOCK": "6000", "AUKIT_DFPWM_CHUNKS": "3"},
                {"AUKIT_DFPWM_BLOCK": "1000", "AUKIT_DFPWM_CHUNKS": "7"}, {"AUKIT_DFPWM_SERIAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        alt = B.decode(ctx, bt, B.make_desc(N.CODEC_MDFPWM), dtype=N.F64).download()
        for k in env:
            monkeypatch.delenv(k)
        for i in range(2):
            for c in range(2):
                assert np.array_equal(alt[i][c], got[i][c]), (env, i, c)
    for mono in (False, True):
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_MDFPWM), "linear", mono=mono, dtype=N.I8)
        o = oracle.stream_mdfpwm(md, mono)
        assert ck.nchunks[0] == o.nchunks and ck.status[0] == o.final_status
        g = out.download()[0]
        for c in range(o.channels):
            assert np.array_equal(g[c], o.data[c])
    with pytest.raises(N.AukitError) as e:
        B.decode(ctx, B.Batch.upload(ctx, [b"RIFFxxxxWAVE"]), B.make_desc(N.CODEC_MDFPWM))
    assert "not a MDFPWM file" in str(e.value)


# ---------------------------------------------------------------- IMA ADPCM
@pytest.mark.parametrize("ch,ba", [(1, 512), (2, 1024), (1, 36), (2, 2048), (1, 2052)])
def test_ima_wav_audio_path(ctx, oracle, ch, ba):
    B, N = _B(), _N()
    spb = (ba - 4 * ch) * 2 // ch
    streams = [oracle.gen_ima(np.stack([pcm16(spb * nb, 22050, 3, 4 * i + c) for c in range(ch)], 1).ravel(), ch, ba, 15) for i, nb in enumerate((20, 1, 3))]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM_WAV, ch, 22050, block_align=ba)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.wav_adpcm(s, ba, ch, 22050)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c])
    res = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F64)
    B.effect(ctx, res, "lowpass", 11025.0)  # config 3b: aukit.wav → :resample(48000,"cubic") → effects.lowpass(a, 11025)
    ref = oracle.fx_lowpass(oracle.resample(oracle.wav_adpcm(streams[0], ba, ch, 22050), 48000, oracle.CUBIC), 11025.0)
    assert np.max(np.abs(res.download()[0][0] - ref.data[0])) <= 1e-12


def test_ima_wav_mono_masks_header_index_and_partial_block(ctx, oracle):
    B, N = _B(), _N()
    s = oracle.gen_ima(pcm16(1016 * 6, 22050, 3, 9), 1, 512, 88)  # full-range header step indices: masked with 0x0F by aukit.wav (Q8)
    s2 = s[:512 * 3 + 100]                                        # partial last block (mono: str_sub is just shorter)
    bt = B.Batch.upload(ctx, [s, s2])
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), dtype=N.F64).download()
    assert np.array_equal(got[0][0], oracle.wav_adpcm(s, 512, 1, 22050).data[0])
    assert np.array_equal(got[1][0], oracle.wav_adpcm(s2, 512, 1, 22050).data[0])


@pytest.mark.parametrize("ba", [8, 12, 36, 260, 512, 516, 1028])
def test_ima_wav_a_lane_per_block(ctx, oracle, monkeypatch, ba):
    """k_ima_lanes (codecs.hip, round 6): one-channel aukit.wav blocks (aukit.lua:1509-1548) decoded a lane each.  150 streams of 1 .. 5 blocks (a wave of
    64 blocks spans dozens of streams; rows start at every multiple of 16 bytes within a line), short last blocks of every length (a part word of 1 .. 3
    bytes, :1545 `str_sub` is simply shorter), against the oracle and against the wave-per-block kernel it replaces (AUKIT_IMA_ROWS_WAVE=1)."""
    B, N = _B(), _N()
    spb = (ba - 4) * 2
    rng = np.random.Generator(np.random.PCG64(100 + ba))
    streams = []
    for i in range(150):
        nb = 1 + i % 5
        s = oracle.gen_ima(pcm16(spb * nb, 22050, 3, i), 1, ba, 88)[:ba * nb]
        cut = int(rng.integers(0, ba - 3)) if i % 3 else 0      # (what is left of the last block: at least its 3 header bytes + ... the loader wants > 2)
        if cut and len(s) - cut > 3 and (len(s) - cut) % ba >= 3:
            s = s[:len(s) - cut]
        streams.append(s)
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=ba)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    monkeypatch.setenv("AUKIT_IMA_ROWS_WAVE", "1")
    old = B.decode(ctx, bt, desc, dtype=N.F64).download()
    monkeypatch.delenv("AUKIT_IMA_ROWS_WAVE")
    for i, s in enumerate(streams):
        ref = oracle.wav_adpcm(s, ba, 1, 22050).data[0]
        assert np.array_equal(got[i][0], ref), (i, len(s))
        assert np.array_equal(old[i][0], ref), (i, len(s))


@pytest.mark.parametrize("ch,interleaved,top_first", [(1, True, True), (1, True, False), (2, True, True), (2, False, False), (3, True, True)])
def test_ima_raw_adpcm(ctx, oracle, ch, interleaved, top_first):
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(17 + ch))
    streams = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (6000, 3, 1025)]
    pred, idx = [100 * (c + 1) for c in range(ch)], [10 * c + 3 for c in range(ch)]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_ADPCM, ch, 22050, interleaved=interleaved, top_first=top_first, predictor=pred, step_index=idx)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.adpcm(s, ch, 22050, top_first, interleaved, pred, idx)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c]), c


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
@pytest.mark.parametrize("ch,mono,ba", [(1, False, 512), (2, False, 1024), (2, True, 1024), (1, False, 260)])
def test_stream_adpcm_bit_exact(ctx, oracle, interp, ch, mono, ba):
    """BASELINE config 3a: stream.adpcm incl. the junk word after every non-final block and the dropped last word (Q6)."""
    B, N = _B(), _N()
    spb = (ba - 4 * ch) * 2 // ch
    streams = [oracle.gen_ima(np.stack([pcm16(spb * nb, 22050, 3, 4 * i + c) for c in range(ch)], 1).ravel(), ch, ba, 88) for i, nb in enumerate((50, 1, 23, 22))]
    streams.append(streams[0][: ba * 7 + 40])  # ragged tail: short final block
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, ch, 22050, block_align=ba), interp, mono=mono, dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_adpcm(s, ba, ch, 22050, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, (i, ck.nchunks[i], ref.nchunks)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        assert np.array_equal(ck.pos[i][:ref.nchunks], ref.chunk_pos)
        for c in range(ref.channels):
            assert np.array_equal(got[i][c], ref.data[c]), (i, c)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate", [48000, 24000, 16000, 44100, 32000, 8000])
def test_stream_adpcm_other_rates_bit_exact(ctx, oracle, rate, interp):
    """stream.adpcm on mono files at 48 kHz (equal rates: every position an integer — the three-tier kernel runs them as 2a / 2), at
    integer ratios and at 44.1 kHz: floored outputs bit for bit the oracle's, random-byte blocks (saturated predictors) included."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(rate))
    streams = [oracle.gen_ima(pcm16(1016 * nb, rate, 3, i), 1, 512, 88) for i, nb in enumerate((30, 1, 9))]
    noise = bytearray(rng.integers(0, 256, 512 * 6, dtype=np.uint8).tobytes())
    for b in range(6):
        noise[512 * b + 2] = int(rng.integers(0, 89)); noise[512 * b + 3] = 0  # a valid header step index
    streams.append(bytes(noise))
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, rate, block_align=512), interp, dtype=N.I8)
    assert ctx.last_kernel()[0] == "k_ima_stream_f32", ctx.last_kernel()
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_adpcm(s, 512, 1, rate, False, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        assert np.array_equal(got[i][0], ref.data[0]), i


def test_int_row_loaders_with_resample_f32_take_the_pcm_wave_kernels(ctx, oracle):
    """IMA / QOA (int16 rows) and DFPWM (int8 rows) loaders with a resample behind them and F32 storage: the rows are 16-bit / 8-bit mono
    strings to the wave kernels of the PCM path — ≤ 1e-6 RMS from the oracle, per channel."""
    B, N = _B(), _N()
    ima = [oracle.gen_ima(np.stack([pcm16(1017 * nb, 22050, 3, 4 * i + c) for c in range(2)], 1).ravel(), 2, 1024, 88) for i, nb in enumerate((20, 3))]
    qoa = [oracle.gen_qoa(np.stack([pcm16(n, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel(), 2, 44100) + b"\0" * 8 for i, n in enumerate((5120 * 3 + 777, 9000))]
    dfp = [oracle.dfpwm_encode(np.round(signal(48000 * 2, 24000, 4, 70 + i) * 100)) for i in range(2)]
    cases = [("ima", ima, B.make_desc(N.CODEC_ADPCM_WAV, 2, 22050, block_align=1024), lambda s: oracle.wav_adpcm(s, 1024, 2, 22050), 22050, "k_fast_wave"),
             ("qoa", qoa, B.make_desc(N.CODEC_QOA, 2, 44100), lambda s: oracle.qoa(s), 44100, "k_fast_wave"),
             ("dfpwm", dfp, B.make_desc(N.CODEC_DFPWM, 1, 24000), lambda s: oracle.dfpwm(s, 1, 24000), 24000, "k_fast_wave")]
    for name, streams, desc, ref_fn, rate, kernel in cases:
        for interp in ("linear", "cubic"):
            got = B.decode_resample(ctx, B.Batch.upload(ctx, streams), desc, 48000, interp, dtype=N.F32).download()
            assert ctx.last_kernel()[0].startswith(kernel), (name, ctx.last_kernel())
            for s, g in zip(streams, got):
                ref = oracle.resample(ref_fn(s), 48000, oracle.INTERP[interp])
                assert len(g) == ref.channels
                for c in range(ref.channels):
                    assert len(g[c]) == len(ref.data[c]), (name, c)
                    assert rms(g[c], ref.data[c]) <= 1e-6, (name, interp, c)


def test_stream_adpcm_config3_shape(ctx, oracle):
    """220 × 512-byte mono blocks @22 050 Hz → 219×2211 + 2194 = 486 403 outputs."""
    B, N = _B(), _N()
    s = oracle.gen_ima(pcm16(1016 * 220, 22050, 3, 0), 1, 512, 88)
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, [s]), B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), "cubic", dtype=N.I8)
    assert int(ck.lens[0].sum()) == 219 * 2211 + 2194 == 486403
    ref = oracle.stream_adpcm(s, 512, 1, 22050, False, oracle.CUBIC)
    assert np.array_equal(out.download()[0][0], ref.data[0])


# ---------------------------------------------------------------- MS ADPCM
@pytest.mark.parametrize("ch,ba", [(2, 1024), (1, 512), (2, 64)])
def test_msadpcm(ctx, oracle, ch, ba):
    B, N = _B(), _N()
    spb = (ba - 14) + 2 if ch == 2 else (ba - 7) * 2 + 2
    streams = [oracle.gen_msadpcm(np.stack([pcm16(spb * nb, 44100, 6, 4 * i + c) for c in range(ch)], 1).ravel(), ch, ba) for i, nb in enumerate((12, 1, 60))]
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_MSADPCM, ch, 44100, block_align=ba)
    got = B.decode(ctx, bt, desc, dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.msadpcm(s, ba, ch, 44100)
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c])
    for mono in ((False, True) if ch == 2 else (False,)):
        for interp in ("linear", "cubic"):
            out, ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=N.I8)
            g = out.download()
            for i, s in enumerate(streams):
                ref = oracle.stream_msadpcm(s, ba, ch, 44100, mono, None, oracle.INTERP[interp])
                assert ck.nchunks[i] == ref.nchunks
                assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
                assert np.array_equal(ck.pos[i][:ref.nchunks], ref.chunk_pos)
                for c in range(ref.channels):
                    assert np.array_equal(g[i][c], ref.data[c]), (i, c, mono, interp)


def test_msadpcm_random_bytes_follow_fp64_semantics(ctx, oracle):
    """Adversarial data drives `delta` out of the integer range; the reference computes in doubles, so do we."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(99))
    raw = bytearray(rng.integers(0, 256, 256 * 4, dtype=np.uint8).tobytes())
    for b in range(4):
        raw[256 * b] %= 7
        raw[256 * b + 1] %= 7
    bt = B.Batch.upload(ctx, [bytes(raw)])
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_MSADPCM, 2, 44100, block_align=256), dtype=N.F64).download()[0]
    ref = oracle.msadpcm(bytes(raw), 256, 2, 44100)
    for c in range(2):
        assert np.array_equal(got[c], ref.data[c], equal_nan=True)


# ---------------------------------------------------------------- QOA
@pytest.mark.parametrize("ch", [1, 2])
def test_qoa_audio_path(ctx, oracle, ch):
    B, N = _B(), _N()
    streams = [oracle.gen_qoa(np.stack([pcm16(n, 44100, 8, 4 * i + c) for c in range(ch)], 1).ravel(), ch, 44100) for i, n in enumerate((5120 * 3 + 777, 5120, 100))]
    streams = [s + b"\0" * 8 for s in streams]  # 8 trailing bytes: otherwise aukit.qoa drops the last frame (frame_size check, Q18)
    streams.append(streams[0][:-8])              # ... and here it does drop it
    bt = B.Batch.upload(ctx, streams)
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_QOA), dtype=N.F64).download()
    for s, g in zip(streams, got):
        ref = oracle.qoa(s)
        assert len(g) == ref.channels
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c])
    res = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_QOA), 48000, "cubic", dtype=N.F64).download()
    ref = oracle.resample(oracle.qoa(streams[0]), 48000, oracle.CUBIC)
    assert np.max(np.abs(res[0][0] - ref.data[0])) <= 1e-15


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
@pytest.mark.parametrize("ch,mono", [(1, False), (2, False), (2, True)])
def test_stream_qoa(ctx, oracle, ch, mono, interp):
    B, N = _B(), _N()
    streams = [oracle.gen_qoa(np.stack([pcm16(n, 44100, 8, 4 * i + c) for c in range(ch)], 1).ravel(), ch, 44100) for i, n in enumerate((44100 * 2 + 1234, 5120 * 9, 777))]
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA), interp, mono=mono, dtype=N.F64)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_qoa(s, mono, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, (i, ck.nchunks[i], ref.nchunks)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert np.array_equal(ck.pos[i][:ref.nchunks], ref.chunk_pos)
        assert ck.status[i] == ref.final_status
        for c in range(ref.channels):
            assert np.max(np.abs(got[i][c] - ref.data[c]), initial=0) <= 1e-12, (i, c)


@pytest.mark.parametrize("rate", [44100, 22050, 8000, 48000, 32000])
@pytest.mark.parametrize("ch,mono", [(1, False), (2, False), (2, True)])
def test_stream_qoa_f32_tail(ctx, oracle, monkeypatch, rs_kernel, ch, mono, rate):
    """F32 storage: stream.qoa's tail (interpolation, clamp, recursive low-pass, channel mean) runs in ONE launch from the int8 rows with the
    interpolation in f32 (k_iir_tail_fast, stream_tail.hip).  Tolerance path: 1e-6 RMS of the [-128, 127] scale, and no single
    sample off by more than 1e-4 of it; the chunk plan is the exact path's."""
    B, N = _B(), _N()
    streams = [oracle.gen_qoa(np.stack([pcm16(n, rate, 8, 4 * i + c) for c in range(ch)], 1).ravel(), ch, rate) for i, n in enumerate((rate * 2 + 1234, 5120 * 9, 777, 20))]
    bt = B.Batch.upload(ctx, streams)
    for interp in ("none", "linear", "cubic"):
        out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA), interp, mono=mono, dtype=N.F32)
        # linear / cubic: the tile chain of k_rs_onepole (state carried from tile to tile: round 4); "none": the warm-up tiles of k_iir_tail_fast
        # (round 6: at 44.1 kHz, cubic, the same chain with weights and tap offsets in registers — k_rsp<..., JOBS>, rs_periodic.hip — for long jobs and for the channels' mean)
        name = ctx.last_kernel()[0]
        assert name in (("k_iir_tail<qoa>",) if interp == "none" else ("k_rs_onepole<qoa>", "k_rsp<qoa>")), ctx.last_kernel()
        if interp != "none":
            assert (name == "k_rsp<qoa>") == (rs_kernel == "default" and interp == "cubic" and rate == 44100 and (ch == 2 and mono or name == "k_rsp<qoa>")), (name, rs_kernel)
        got = out.download()
        if interp != "none":
            monkeypatch.setenv("AUKIT_NO_RS_JOBS", "1")
            out2, _ = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA), interp, mono=mono, dtype=N.F32)
            monkeypatch.delenv("AUKIT_NO_RS_JOBS")
            assert ctx.last_kernel()[0] == "k_iir_tail<qoa>", ctx.last_kernel()
            got2 = out2.download()
            for i in range(len(streams)):
                for c in range(len(got[i])):
                    assert np.max(np.abs(got[i][c] - got2[i][c]), initial=0) <= 1e-4, (interp, i, c)   # (two f32 evaluations of the same taps; [-128, 127] scale)
        for i, s in enumerate(streams):
            ref = oracle.stream_qoa(s, mono, oracle.INTERP[interp])
            assert ck.nchunks[i] == ref.nchunks and list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            for c in range(ref.channels):
                assert rms(got[i][c] / 128, ref.data[c] / 128) <= 1e-6, (interp, i, c)
                assert np.max(np.abs(got[i][c] - ref.data[c]), initial=0) <= 128e-4, (interp, i, c)


def test_dfpwm_parallel_encoder_small_batches(ctx, oracle, monkeypatch):
    """Batches of a few long streams encode in parallel chunks (candidate start states from a 2048-sample warm-up of every
    (strength, previous bit) pair, true states chained through them): the bytes equal the serial encoder's and the oracle's for
    ordinary audio, rail-to-rail squares, silence, full-scale noise and constant rails, with 64 / 7 / 200 chunks per stream."""
    B, N = _B(), _N()
    rng = np.random.Generator(np.random.PCG64(21))
    n = 200000
    t = np.arange(n) / 48000
    sigs = [0.6 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.1, 0.1, n), np.where((np.arange(n) // 300) % 2 == 0, 1.0, -1.0), np.zeros(n), rng.uniform(-1, 1, n),
            np.full(n, 1.0), np.full(n, -1.0), 0.02 * np.sin(2 * np.pi * 50 * t), np.concatenate([np.zeros(70000), rng.uniform(-1, 1, 60001), np.full(69999, 0.5)])]
    sigs.append(sigs[0][:65536])
    sigs.append(sigs[3][:70003])
    ab = B.AudioBatch.upload(ctx, [[x] for x in sigs], 48000, dtype=N.F64)
    monkeypatch.setenv("AUKIT_DFPWM_NOSPEC", "1")   # (the candidate search of round 3: since round 5 the chunk-speculative encoder comes first, tests/test_gpu_dfpwm_spec.py)
    got = B.dfpwm_encode(ctx, ab, True).download()
    assert ctx.last_kernel()[0] == "k_dfpwm_quantize+k_dfe_*", ctx.last_kernel()
    for x, g in zip(sigs, got):
        assert g == oracle.audio_dfpwm(oracle.Audio([x], 48000), True)
    for env in ({"AUKIT_DFPWM_ENC_CHUNKS": "7"}, {"AUKIT_DFPWM_ENC_CHUNKS": "200"}, {"AUKIT_DFPWM_SERIAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        again = B.dfpwm_encode(ctx, ab, True).download()
        for k in env:
            monkeypatch.delenv(k)
        assert again == got, env
    monkeypatch.delenv("AUKIT_DFPWM_NOSPEC")
    assert B.dfpwm_encode(ctx, ab, True).download() == got   # the default schedule (probe, speculation or not): the same bytes
    # stereo, interleaved and channel after channel, through the same encoder
    st = [[sigs[0][:100000], sigs[3][:100000]]]
    ab2 = B.AudioBatch.upload(ctx, st, 48000, dtype=N.F64)
    for inter in (True, False):
        assert B.dfpwm_encode(ctx, ab2, inter).download()[0] == oracle.audio_dfpwm(oracle.Audio(st[0], 48000), inter)


def test_dfpwm_chunk_decoder_in_digital_silence(ctx, oracle):
    """DFPWM made from audio with stretches of digital silence (the idle pattern aukit.detect looks for, aukit.lua:2193 — 0x55 or 0xAA bytes, depending on
    where the silence starts): the chunk-parallel decoder's warm-ups converge there as they do on signal (measured: one chunk per stream decoded again,
    at a transition), the samples are the oracle's."""
    B, N = _B(), _N()
    n = 800000
    streams = []
    for k in range(6):   # silence from the first sample, from the second (the idle bytes are 0xAA in one case, 0x55 in the other), and inside the signal
        x = np.round(signal(n, 48000, 4, 70 + k) * 100)
        x[k % 2: n // 4 + k] = 0
        x[n // 2 + 3 * k: n // 2 + n // 5] = 0
        streams.append(oracle.dfpwm_encode(x))
    assert all(b"\x55" * 64 in s or b"\xaa" * 64 in s for s in streams)
    bt = B.Batch.upload(ctx, streams)
    ctx.set_option(N.OPT_COLLECT_STATS, 1)
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), dtype=N.F64).download()
    chunks, redone = ctx.counter(N.COUNTER_DFPWM_CHUNKS), ctx.counter(N.COUNTER_DFPWM_CHUNKS_REDONE)
    ctx.set_option(N.OPT_COLLECT_STATS, 0)
    for i, d in enumerate(streams):
        ref = oracle.dfpwm(d, 2, 48000)
        assert np.array_equal(got[i][0], ref.data[0]) and np.array_equal(got[i][1], ref.data[1])
    assert chunks > 10 and redone * 10 <= chunks, (chunks, redone)


@pytest.mark.parametrize("ch,ba", [(2, 4096), (2, 8192), (4, 4096)])
def test_stream_adpcm_large_blocks(ctx, oracle, ch, ba):
    """stream.adpcm on blocks whose decoded samples do not fit 64 KiB (blockAlign x channels from ~4 KiB on: refused until round 6) — the kernel takes
    the CU's whole LDS for a block; chunk for chunk the oracle's"""
    B, N = _B(), _N()
    spb = (ba - 4 * ch) * 2 // ch
    streams = [oracle.gen_ima(np.stack([pcm16(spb * nb, 22050, 3, 4 * i + c) for c in range(ch)], 1).ravel(), ch, ba, 88) for i, nb in enumerate((5, 1, 3))]
    streams.append(streams[0][: ba * 2 + 40])  # short final block
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, ch, 22050, block_align=ba), "cubic", dtype=N.I8)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_adpcm(s, ba, ch, 22050, False, oracle.CUBIC)
        assert ck.nchunks[i] == ref.nchunks, (i, ck.nchunks[i], ref.nchunks)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0]), i
        for c in range(ref.channels):
            assert np.array_equal(got[i][c], ref.data[c]), (i, c)


def test_stream_adpcm_back_to_back_calls(ctx, oracle):
    """stream.adpcm's header scan runs on the look-ahead stream and nothing waits for the kernel's own (unreachable) error flag: calls issued one behind the
    other on different batches — one with a stream whose third block carries a step index above 88 — give what they give one at a time, and that is the oracle's"""
    B, N = _B(), _N()
    batches = []
    for b in range(3):
        streams = [oracle.gen_ima(pcm16(1016 * (20 + 7 * i + b), 22050, 3, 10 * b + i), 1, 512, 88) for i in range(24)]
        if b == 1:
            bad = bytearray(streams[5]); bad[2 * 512 + 2] = 120; streams[5] = bytes(bad)
        batches.append((streams, B.Batch.upload(ctx, streams)))
    desc = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)

    def run(sync_each):
        res = []
        for k in range(9):
            out, ck = B.stream_decode(ctx, batches[k % 3][1], desc, "cubic", dtype=N.I8)
            res.append((out, ck))
            if sync_each:
                ctx.sync()
        return [(o.download(), list(c.nchunks), list(c.status)) for o, c in res]

    ref = run(True)
    for rep in range(3):
        got = run(False)
        for k in range(9):
            assert got[k][1] == ref[k][1] and got[k][2] == ref[k][2], (rep, k)
            for s in range(24):
                assert np.array_equal(got[k][0][s][0], ref[k][0][s][0]), (rep, k, s)
    for k in (0, 1):
        for s in (0, 5):
            r = oracle.stream_adpcm(batches[k][0][s], 512, 1, 22050, False, oracle.CUBIC)
            assert ref[k][1][s] == r.nchunks and (ref[k][2][s] != 0) == (r.final_status != 0)
            assert np.array_equal(ref[k][0][s][0], r.data[0])


def test_stream_qoa_back_to_back_calls(ctx, oracle):
    """stream.qoa's two walks run on the look-ahead stream into alternating sets of words: calls issued one behind the other on different batches (stereo, the
    channels' mean, one channel) give what they give one at a time, and that is the oracle's"""
    B, N = _B(), _N()
    batches = []
    for b in range(3):
        ch = 2 if b < 2 else 1
        streams = [oracle.gen_qoa(np.stack([pcm16(30000 + 4000 * i + 777 * b, 44100, 8, 40 * b + 2 * i + c) for c in range(ch)], 1).ravel(), ch, 44100) + b"\0" * 8 for i in range(12)]
        batches.append((streams, B.Batch.upload(ctx, streams), ch))

    def run(sync_each):
        res = []
        for k in range(9):
            streams, bt, ch = batches[k % 3]
            out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA), "cubic", mono=(k % 3 == 1), dtype=N.F32)
            res.append((out, ck))
            if sync_each:
                ctx.sync()
        return [(o.download(), list(c.nchunks)) for o, c in res]

    ref = run(True)
    for rep in range(3):
        got = run(False)
        for k in range(9):
            assert got[k][1] == ref[k][1], (rep, k)
            for s in range(12):
                for c in range(len(ref[k][0][s])):
                    assert np.array_equal(got[k][0][s][c], ref[k][0][s][c]), (rep, k, s, c)
    for k in range(3):
        r = oracle.stream_qoa(batches[k][0][3], k == 1, oracle.CUBIC)
        for c in range(r.channels):
            assert rms(ref[k][0][3][c] / 128, r.data[c] / 128) <= 1e-6, (k, c)


@pytest.mark.parametrize("scan", ["look-ahead scan", "kernel flag"])
def test_stream_msadpcm_predictor_index_beyond_the_table(ctx, oracle, monkeypatch, scan):
    """aukit.stream.msadpcm (aukit.lua:2588-2736): a block header's predictor index selects a coefficient pair; beyond the table `c1` is nil and the
    arithmetic raises.  One channel: every block uses the FIRST block's index (Q9) — the host looks at one byte per stream on the look-ahead stream
    (round 6) instead of waiting for the kernel's flag; both ways raise the reference's error, and a clean batch decodes to the oracle's bytes either way."""
    B, N = _B(), _N()
    if scan == "kernel flag":
        monkeypatch.setenv("AUKIT_MS_SCAN_OFF", "1")
    good = [oracle.gen_msadpcm(pcm16(2036 * (3 + i), 44100, 3, 70 + i), 1, 1024) for i in range(5)]
    desc = B.make_desc(N.CODEC_MSADPCM, 1, 44100, block_align=1024)
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, good), desc, "cubic", dtype=N.I8)
    got = out.download()
    for i, s in enumerate(good):
        r = oracle.stream_msadpcm(s, 1024, 1, 44100, False, None, oracle.CUBIC)
        assert ck.nchunks[i] == r.nchunks and np.array_equal(got[i][0], r.data[0]), i
    bad = bytearray(good[2]); bad[0] = 7          # the default table has seven pairs: 0 .. 6
    with pytest.raises(N.AukitError, match="local 'c1'"):
        B.stream_decode(ctx, B.Batch.upload(ctx, good[:2] + [bytes(bad)] + good[3:]), desc, "cubic", dtype=N.I8)
    later = bytearray(good[2]); later[1024] = 200  # a LATER block's index is never read (one channel: Q9)
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, [bytes(later)]), desc, "cubic", dtype=N.I8)
    r = oracle.stream_msadpcm(bytes(later), 1024, 1, 44100, False, None, oracle.CUBIC)
    assert np.array_equal(out.download()[0][0], r.data[0])
