"""GPU parity for FLAC (frame-parallel decode) vs the CPU oracle and vs the original PCM (losslessness)."""
import numpy as np
import pytest

from tests.util import pcm16, rms, signal, tail_kernel

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "stream"])
def _both_register_decoders(request, monkeypatch):
    """round 6: small batches decode with k_flac_pq (a parser and a predictor wave per 64 frames), batches that fill the chip with k_flac_stream
    (one wave) — every case of this module runs through both (AUKIT_FLAC_DECODER, read per call; the round-4 kernel stays behind "fused")"""
    if request.param != "auto":
        monkeypatch.setenv("AUKIT_FLAC_DECODER", request.param)



def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


def _pcm(n, ch, depth, cfg=5, stream=0):
    rng = np.random.Generator(np.random.PCG64(1000 * cfg + stream))
    cols = [(signal(n, 44100, cfg, 10 * stream + c) * (2 ** (depth - 1) - 1) * 0.9).astype(np.int64) for c in range(ch)]
    p = np.stack(cols, 1)
    if n > 16384:
        p[4096:8192] = 77                                 # CONSTANT
        p[8192:12288] = (p[8192:12288] >> 2) << 2          # wasted bits
        p[12288:16384] = rng.integers(-(2 ** (depth - 1)), 2 ** (depth - 1), (4096, ch))  # escape-coded partitions
    return p


@pytest.mark.parametrize("depth", [8, 16, 24])
@pytest.mark.parametrize("ch", [1, 2])
def test_flac_decode_bit_exact_and_lossless(ctx, oracle, depth, ch):
    B, N = _B(), _N()
    pcms = [_pcm(n, ch, depth, 5, i) for i, n in enumerate((4096 * 13 + 999, 5000, 4096, 100))]
    streams = [oracle.gen_flac(p.ravel(), ch, depth, 44100, 4096) for p in pcms]
    bt = B.Batch.upload(ctx, streams)
    got = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
    for p, s, g in zip(pcms, streams, got):
        ref = oracle.flac(s)
        assert len(g) == ch
        for c in range(ch):
            assert np.array_equal(g[c], ref.data[c])
            assert np.array_equal(np.round(g[c] * 2 ** depth).astype(np.int64), p[:, c])  # independent: FLAC is lossless


@pytest.mark.parametrize("depth", [16, 24])
def test_flac_both_prediction_kernels_agree(ctx, oracle, monkeypatch, depth):
    """int32 rows are predicted by the 24-bit multiply-add kernel where it can promise exactness (16-bit audio) and by the 64-bit one
    elsewhere (24-bit audio trips the bound, ragged last frames are declined up front); AUKIT_FLAC_SLOW_RESTORE sends every wave to the
    second.  Stereo so that every channel assignment, wasted bits and the decorrelation out of the partner lane take part."""
    B, N = _B(), _N()
    pcms = [_pcm(n, 2, depth, 5, 20 + i) for i, n in enumerate((4096 * 40 + 2728, 4096 * 3, 1152 * 9 + 4))]
    streams = [oracle.gen_flac(p.ravel(), 2, depth, 44100, bs) for p, bs in zip(pcms, (4096, 4096, 1152))] * 12   # several full waves of subframes
    bt = B.Batch.upload(ctx, streams)
    fast = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
    monkeypatch.setenv("AUKIT_FLAC_SLOW_RESTORE", "1")
    slow = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
    monkeypatch.delenv("AUKIT_FLAC_SLOW_RESTORE")
    for i, (f, s) in enumerate(zip(fast, slow)):
        p = pcms[i % 3]
        for c in range(2):
            assert np.array_equal(f[c], s[c])
            assert np.array_equal(np.round(f[c] * 2 ** depth).astype(np.int64), p[:, c])


def test_flac_blocksizes_and_config5_pipeline(ctx, oracle):
    B, N = _B(), _N()
    st = np.stack([pcm16(30000, 44100, 5, 0), pcm16(30000, 44100, 5, 1)], 1).astype(np.int64)
    for bs in (192, 576, 1024, 1000, 4608):
        s = oracle.gen_flac(st.ravel(), 2, 16, 44100, bs)
        got = B.decode(ctx, B.Batch.upload(ctx, [s]), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()[0]
        assert np.array_equal(np.round(got[0] * 65536).astype(np.int64), st[:, 0]) and np.array_equal(np.round(got[1] * 65536).astype(np.int64), st[:, 1]), bs
    # BASELINE config 5: aukit.flac(d):resample(48000,"cubic") → effects.highpass(a,20) → effects.normalize(a,0.8) → a:mono()
    s = oracle.gen_flac(st.ravel(), 2, 16, 44100, 4096)
    bt = B.Batch.upload(ctx, [s, s])
    a = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_FLAC), 48000, "cubic", dtype=N.F64)
    B.effect(ctx, a, "highpass", 20.0)
    B.effect(ctx, a, "normalize", 0.8)
    m = B.mono(ctx, a).download()
    ref = oracle.mono(oracle.fx_normalize(oracle.fx_highpass(oracle.resample(oracle.flac(s), 48000, oracle.CUBIC), 20.0), 0.8))
    assert np.max(np.abs(m[0][0] - ref.data[0])) <= 1e-11 and np.array_equal(m[0][0], m[1][0])
    a32 = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_FLAC), 48000, "cubic", dtype=N.F32)
    assert ctx.last_kernel()[0] == "(resample deferred)"       # F32 pipelines: the resample is owed (flac_tail.hip) ...
    a32_rows = a32.download()
    assert ctx.last_kernel()[0].startswith("k_fast_wave<i32")  # ... and a reader materialises it from the int32 rows with the f32 tolerance kernel
    assert rms(a32_rows[0][1], oracle.resample(oracle.flac(s), 48000, oracle.CUBIC).data[1]) <= 1e-6
    for depth, interp in ((24, "linear"), (8, "cubic")):
        p = _pcm(30000, 2, depth, 5, 3)
        sd = oracle.gen_flac(p.ravel(), 2, depth, 44100, 4096)
        g = B.decode_resample(ctx, B.Batch.upload(ctx, [sd]), B.make_desc(N.CODEC_FLAC), 48000, interp, dtype=N.F32).download()[0]
        r = oracle.resample(oracle.flac(sd), 48000, oracle.INTERP[interp])
        assert rms(g[0], r.data[0]) <= 1e-6 and rms(g[1], r.data[1]) <= 1e-6
    B.effect(ctx, a32, "highpass", 20.0)
    B.effect(ctx, a32, "normalize", 0.8)
    assert rms(B.mono(ctx, a32).download()[0][0], ref.data[0]) <= 1e-6


def test_flac_errors_match_reference(ctx, oracle):
    B, N = _B(), _N()
    st = np.stack([pcm16(9000, 44100, 5, 0), pcm16(9000, 44100, 5, 1)], 1).astype(np.int64)
    s = oracle.gen_flac(st.ravel(), 2, 16, 44100, 4096)
    for bad, msg in ((s[:-7], "nil"), (s + b"ID3\x00garbage", "Sync code expected"), (b"RIFF" + s[4:], "Invalid magic string")):
        with pytest.raises(oracle.OracleError):
            oracle.flac(bad)
        with pytest.raises(N.AukitError) as e:
            B.decode(ctx, B.Batch.upload(ctx, [bad]), B.make_desc(N.CODEC_FLAC))
        assert msg in str(e.value)
    # a sync-looking pattern inside the audio data must not confuse the chain: random payload frames
    rng = np.random.Generator(np.random.PCG64(4))
    noisy = rng.integers(-32768, 32768, (4096 * 6, 2))
    noisy[100:5000:7] = [-1, -8]  # 0xFFFF 0xFFF8 byte patterns → many false sync candidates in verbatim/escape data
    sn = oracle.gen_flac(noisy.ravel(), 2, 16, 44100, 4096)
    got = B.decode(ctx, B.Batch.upload(ctx, [sn]), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()[0]
    assert np.array_equal(np.round(got[0] * 65536).astype(np.int64), noisy[:, 0])


@pytest.mark.parametrize("interp", ["none", "linear", "cubic"])
def test_stream_flac(ctx, oracle, interp):
    B, N = _B(), _N()
    streams = []
    for i, (n, ch) in enumerate(((44100 * 2 + 500, 2), (4096 * 3, 2), (700, 2))):
        p = np.stack([pcm16(n, 44100, 5, 2 * i + c) for c in range(ch)], 1).astype(np.int64)
        streams.append(oracle.gen_flac(p.ravel(), ch, 16, 44100, 4096))
    streams.append(streams[0][: len(streams[0]) // 2])  # truncated: the decode error is swallowed, the stream just ends
    bt = B.Batch.upload(ctx, streams)
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_FLAC), interp, dtype=N.F64)
    got = out.download()
    for i, s in enumerate(streams):
        ref = oracle.stream_flac(s, oracle.INTERP[interp])
        assert ck.nchunks[i] == ref.nchunks, (i, ck.nchunks[i], ref.nchunks)
        assert list(ck.lens[i][:ref.nchunks]) == list(ref.chunk_len[:, 0])
        assert np.allclose(ck.pos[i][:ref.nchunks], ref.chunk_pos, rtol=0, atol=1e-12)
        for c in range(ref.channels):
            assert np.max(np.abs(got[i][c] - ref.data[c]), initial=0) <= 1e-12, (i, c)


@pytest.mark.parametrize("rate,bs", [(44100, 4096), (22050, 1152), (8000, 576), (48000, 4096)])
def test_stream_flac_f32_tail(ctx, oracle, monkeypatch, rate, bs):
    """F32 storage: stream.flac's per-block resample + recursive low-pass + scaling in one launch from the int32 rows, f32 interpolation
    (k_iir_tail_fast).  Tolerance path: 1e-6 RMS of the [-128, 127] scale."""
    B, N = _B(), _N()
    streams = []
    for i, (n, ch) in enumerate(((rate * 2 + 500, 2), (bs * 3, 1), (700, 2), (1, 1))):
        p = np.stack([pcm16(n, rate, 5, 2 * i + c) for c in range(ch)], 1).astype(np.int64)
        streams.append(oracle.gen_flac(p.ravel(), ch, 16, rate, bs))
    for s in streams:
        bt = B.Batch.upload(ctx, [s])
        for interp in ("none", "linear", "cubic"):
            out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_FLAC), interp, dtype=N.F32)
            # linear / cubic: the tile chain of k_rs_onepole (JOBS: a job per frame and channel, state carried from tile to tile; round 4) — "none": k_iir_tail_fast
            assert ctx.last_kernel()[0] == ("k_iir_tail<flac>" if interp == "none" else "k_rs_onepole<flac>"), ctx.last_kernel()
            got = out.download()[0]
            if interp != "none":
                monkeypatch.setenv("AUKIT_NO_RS_JOBS", "1")
                out2, _ = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_FLAC), interp, dtype=N.F32)
                monkeypatch.delenv("AUKIT_NO_RS_JOBS")
                assert ctx.last_kernel()[0] == "k_iir_tail<flac>", ctx.last_kernel()
                got2 = out2.download()[0]
                for c in range(len(got)):
                    assert np.max(np.abs(got[c] - got2[c]), initial=0) <= 1e-4, (interp, c)
            ref = oracle.stream_flac(s, oracle.INTERP[interp])
            assert ck.nchunks[0] == ref.nchunks and list(ck.lens[0][:ref.nchunks]) == list(ref.chunk_len[:, 0])
            for c in range(ref.channels):
                assert rms(got[c] / 128, ref.data[c] / 128) <= 1e-6, (interp, c)
                assert np.max(np.abs(got[c] - ref.data[c]), initial=0) <= 128e-4, (interp, c)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("rate,ch,bs", [(44100, 2, 4096), (22050, 1, 1152), (8000, 2, 576), (48000, 2, 4096)])
def test_deferred_resample_fused_into_the_filter_pass(ctx, oracle, monkeypatch, rs_kernel, rate, ch, bs, interp):
    """config 5's tail, round 3 (flac_tail.hip): aukit_decode_resample on FLAC with F32 storage leaves the resample OWED (the decoder's int32 rows
    move into the audio); effects.highpass / lowpass pay it inside their own pass (k_rs_onepole: same f32 interpolation as k_fast_wave<i32>, the
    recurrence in fp64 with an affine carry scan), every other reader materialises it with the ordinary kernel.  Observable results: the
    materialised rows are bit-identical to the eager path (AUKIT_NO_TAIL_FUSION=1); the fused filter agrees with resample-then-filter to a few f32
    ulps (another scan decomposition) and with the oracle to 1e-6 RMS; per-row maxima feed effects.normalize as before."""
    B, N = _B(), _N()
    lens = (rate * 2 + 777, bs * 3, 4097, 700, 2100)
    streams = []
    for i, n in enumerate(lens):
        p = np.stack([pcm16(n, rate, 5, 2 * i + c) for c in range(ch)], 1).astype(np.int64)
        streams.append(oracle.gen_flac(p.ravel(), ch, 16, rate, bs))
    bt = B.Batch.upload(ctx, streams)
    desc = B.make_desc(N.CODEC_FLAC)

    def chain(which):
        a = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32)
        name0 = ctx.last_kernel()[0]
        if which == "download":
            return a.download(), name0
        if which == "mono":
            return B.mono(ctx, a).download(), name0
        if which == "amplify":
            B.effect(ctx, a, "amplify", 0.5)
            return a.download(), name0
        B.effect(ctx, a, which, 300.0 if which == "lowpass" else 20.0)
        name1 = ctx.last_kernel()[0]
        B.effect(ctx, a, "normalize", 0.8)
        m_first = B.mono(ctx, a)   # two channels: resample + filter + mean in one pass, the normalize owed on the result; `a` keeps what it owes
        name2 = ctx.last_kernel()[0]
        m_first = m_first.download()
        return (a.download(), B.mono(ctx, a).download(), m_first), (name0, name1, name2)

    for which in ("download", "mono", "amplify"):
        got, name = chain(which)
        assert name == "(resample deferred)", name
        monkeypatch.setenv("AUKIT_NO_TAIL_FUSION", "1")
        plain, name_p = chain(which)
        monkeypatch.delenv("AUKIT_NO_TAIL_FUSION")
        assert name_p.startswith(("k_fast_wave<", "k_resample<")), name_p
        for s in range(len(streams)):
            for c in range(len(got[s])):
                assert np.array_equal(got[s][c], plain[s][c]), (which, s, c)
    per = interp == "cubic" and rate in (44100, 22050)   # k_rsp's shapes (the decoder's finals are int16 where the batch allows: either kernel's name passes, the values decide)
    for which in ("highpass", "lowpass"):
        (rows, mono, mono1), (n0, n1, n2) = chain(which)
        if ch == 2:   # round 4, late: the filter of a stereo audio is owed too, Audio:mono pays resample + filter + mean at once
            assert n0 == "(resample deferred)" and n1 == "(filter deferred)" and n2 in (tail_kernel(which, rs_kernel, per, True), tail_kernel(which, "generic", per, True)), (n0, n1, n2)
        else:
            assert n0 == "(resample deferred)" and n1 in (tail_kernel(which, rs_kernel, per), tail_kernel(which, "generic", per)) and n2.startswith("k_mono"), (n0, n1, n2)
        monkeypatch.setenv("AUKIT_NO_MONO_FUSION", "1")
        (rows_q, mono_q, mono1_q), (q0, q1, q2) = chain(which)
        monkeypatch.delenv("AUKIT_NO_MONO_FUSION")
        assert q1 in (tail_kernel(which, rs_kernel, per), tail_kernel(which, "generic", per)) and q2.startswith("k_mono"), (q1, q2)
        monkeypatch.setenv("AUKIT_NO_TAIL_FUSION", "1")
        (rows_p, mono_p, mono1_p), (p0, p1, p2) = chain(which)
        monkeypatch.delenv("AUKIT_NO_TAIL_FUSION")
        assert p1.startswith("k_onepole<"), p1
        for s_i, s in enumerate(streams):
            ref = oracle.resample(oracle.flac(s), 48000, oracle.INTERP[interp])
            ref = oracle.fx_highpass(ref, 20.0) if which == "highpass" else oracle.fx_lowpass(ref, 300.0)
            ref = oracle.fx_normalize(ref, 0.8)
            for c in range(ch):
                assert len(rows[s_i][c]) == len(ref.data[c])
                assert np.max(np.abs(rows[s_i][c] - rows_p[s_i][c]), initial=0) <= 4e-7, (which, s_i, c)
                assert rms(rows[s_i][c], ref.data[c]) <= 1e-6, (which, s_i, c)
            assert np.max(np.abs(mono[s_i][0] - mono_p[s_i][0]), initial=0) <= 4e-7
            assert rms(mono[s_i][0], oracle.mono(ref).data[0]) <= 1e-6
            for c in range(ch):
                assert np.array_equal(rows[s_i][c], rows_q[s_i][c]), (which, s_i, c)   # (paid by the same kernel whether the filter was owed or run at once)
            assert len(mono1[s_i][0]) == len(mono_p[s_i][0])
            assert np.max(np.abs(mono1[s_i][0] - mono_p[s_i][0]), initial=0) <= 6e-7, (which, s_i)
            assert rms(mono1[s_i][0], oracle.mono(ref).data[0]) <= 1e-6
    # the output audio reused for the next decode gives the rows' buffer back (no growth), and a deferred audio can be cloned / freed
    a = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32)
    a = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F32, out=a)
    b = a.clone()
    want = B.decode_resample(ctx, bt, desc, 48000, interp, dtype=N.F64).download()
    for s in range(len(streams)):
        for c in range(ch):
            assert rms(b.download()[s][c], want[s][c]) <= 1e-6
    a.free()


def test_back_to_back_calls_share_the_look_ahead_stream(ctx, oracle, monkeypatch):
    """Round 6: a FLAC call's search, decoder and chain walk run on the look-ahead stream while ctx->stream still works on the call BEFORE (its tile chain, its
    normalize); the frame scratch, the two table sets and the audios' deferred work are what the two streams share.  Calls issued back to back with nothing
    waited for in between — different batches, outputs kept, every way a scratch is consumed (the fused tail with and without the channels' mean, a
    download that gathers, a loader call without a resample) — must give what the same calls give one at a time."""
    B, N = _B(), _N()
    monkeypatch.setenv("AUKIT_FLAC_LOOKAHEAD", "1")
    rng = np.random.Generator(np.random.PCG64(77))
    batches = []
    for b in range(3):
        distinct = []
        for i in range(6):
            n = int(rng.integers(60000, 120000))
            p = np.stack([pcm16(n, 44100, 5, 10 * b + 2 * i + c) for c in range(2)], 1).astype(np.int64)
            distinct.append(oracle.gen_flac(p.ravel(), 2, 16, 44100, 4096, salt=b))
        streams = [distinct[i % 6] for i in range(256)]   # (kernels of a good part of a millisecond: the streams have something to overlap)
        batches.append((streams, B.Batch.upload(ctx, streams)))
    desc = B.make_desc(N.CODEC_FLAC)

    def run(order, sync_each):
        outs = []
        for k, (b, how) in enumerate(order):
            bt = batches[b][1]
            if how == "mono":
                a = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32)
                B.effect(ctx, a, "highpass", 20.0)
                B.effect(ctx, a, "normalize", 0.8)
                outs.append(B.mono(ctx, a))
            elif how == "tail":
                a = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32)
                B.effect(ctx, a, "lowpass", 3000.0)
                outs.append(a)
            elif how == "rows":
                outs.append(B.decode_resample(ctx, bt, desc, 48000, "linear", dtype=N.F32))   # (materialised at the download: the gather)
            else:
                outs.append(B.decode(ctx, bt, desc, dtype=N.F32))
            if sync_each:
                ctx.sync()
        return [o.download() for o in outs]

    order = [(0, "mono"), (1, "tail"), (2, "mono"), (0, "rows"), (1, "plain"), (2, "tail"), (0, "mono"), (1, "mono"), (2, "rows"), (0, "tail")]
    ref = run(order, True)
    for rep in range(3):
        got = run(order, False)
        for k in range(len(order)):
            for s in range(len(ref[k])):
                for c in range(len(ref[k][s])):
                    assert np.array_equal(got[k][s][c], ref[k][s][c]), (rep, k, order[k], s, c)
                    if s >= 6:
                        assert np.array_equal(got[k][s][c], got[k][s % 6][c]), (rep, k, s, c)   # (copies of a file: the same samples)
    # ... and the first of them against the oracle
    s0 = batches[0][0][0]
    r0 = oracle.mono(oracle.fx_normalize(oracle.fx_highpass(oracle.resample(oracle.flac(s0), 48000, oracle.CUBIC), 20.0), 0.8)).data[0]
    assert rms(ref[0][0][0], r0) <= 1e-6
