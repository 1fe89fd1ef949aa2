"""The product path sharded: a batch is cut with shard.partition, every shard is handed to the library from a device
tensor through aukit_batch_wrap_device (what a rank does with the bytes RCCL delivered), runs in its OWN context, and the shards' rows
put back in rank order are, bit for bit, the rows of the unsharded call.  One GPU is all a test box has, so the "ranks" are contexts
on cuda:0; the transport itself is tested over gloo (tests/test_host_math.py) and the N-rank launch in tests/test_bench_launch.py."""
import numpy as np
import pytest

from tests.util import pcm16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_single(ctx, world):
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    from aukit_amd import shard
    lens = [30000, 100, 4097, 1, 52000, 7, 9000, 1024, 2048, 33333, 5, 777]
    streams = [pcm16(n, 44100, 1, i).tobytes() for i, n in enumerate(lens)]
    desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
    whole = B.Batch.upload(ctx, streams)  # the job's bytes, resident in HBM
    starts = whole.offsets().astype(np.int64)
    for exact in (0, 1):
        ctx.set_option(N.OPT_EXACT_MATH, exact)
        try:
            single = B.decode_resample(ctx, whole, desc, 48000, "cubic", dtype=N.F32).download()
        finally:
            ctx.set_option(N.OPT_EXACT_MATH, 0)
        parts = shard.partition([len(s) for s in streams], world)
        assert parts[0][0] == 0 and parts[-1][1] == len(streams) and all(parts[g][1] == parts[g + 1][0] for g in range(world - 1))
        rows = []
        for lo, hi in parts:
            c2 = B.Context(0)  # a rank's own context (own stream, own scratch)
            c2.set_option(N.OPT_EXACT_MATH, exact)
            try:
                # what rank g holds after the scatter: `hi - lo` streams back to back in device memory, handed over with aukit_batch_wrap_device
                bt = B.Batch.wrap(c2, whole.device_ptr() + int(starts[lo]), (starts[lo:hi + 1] - starts[lo]).astype(np.uint64), keep=whole)
                rows += B.decode_resample(c2, bt, desc, 48000, "cubic", dtype=N.F32).download()
            finally:
                c2.close()
        assert len(rows) == len(single)
        for a, b in zip(rows, single):
            assert np.array_equal(a[0], b[0])


def test_device_views_and_one_rank_group():
    """shard.device_view / scatter_batch / gather_audio on the GPU, in a child process that loads torch BEFORE the library (torch bundles its
    own HIP runtime and must be the first to load it): a one-rank RCCL group, where the source's shard and the gathered rows are zero-copy
    views of the library's own device memory."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import torch.distributed as dist
torch.cuda.set_device(0)
from aukit_amd import _native as N, batch as B, shard
from tests.util import pcm16
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29877", rank=0, world_size=1, device_id=dev)
ctx = B.Context(0)
streams = [pcm16(n, 44100, 1, i).tobytes() for i, n in enumerate([5000, 300, 41, 0, 70000])]
bt = B.Batch.upload(ctx, streams)
v = shard.device_view(bt.device_ptr(), int(bt.offsets()[-1]), dev, keep=bt)
assert v.data_ptr() == bt.device_ptr() and v.cpu().numpy().tobytes() == b"".join(streams)
mine, (lo, hi) = shard.scatter_batch(ctx, bt, src=0, device=dev)
assert (lo, hi) == (0, 5) and mine.device_ptr() == bt.device_ptr()
desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
out = B.decode_resample(ctx, mine, desc, 48000, "cubic", dtype=N.F32)
ctx.sync()
got = shard.gather_audio(out, dst=0, device=dev)
assert len(got) == 1
t, m = got[0]
assert t.data_ptr() == out.device_ptr() and m["channels"] == 1 and m["dtype"] == N.F32 and m["rate"] == 48000.0
rows = t.view(torch.float32).cpu().numpy()
for s, g in enumerate(out.download()):
    assert int(m["lens"][s]) == len(g[0])
    assert np.array_equal(rows[int(m["row_off"][s]):int(m["row_off"][s]) + len(g[0])].astype(np.float64), g[0])
enc = B.dfpwm_encode(ctx, B.decode_resample(ctx, mine, desc, 48000, "cubic", dtype=N.F64)) if hasattr(B, "dfpwm_encode") else None
if enc is not None:
    gb = shard.gather_batch(enc, dst=0, device=dev)
    assert len(gb) == 1 and gb[0][0].cpu().numpy().tobytes() == b"".join(enc.download())
dist.destroy_process_group()
print("OK")
""" % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-3000:]


@pytest.mark.parametrize("world,root", [(1, 0), (2, 0), (3, 2), (8, 5)])
def test_group_scatter_compute_gather_through_the_c_abi(ctx, world, root):
    """sharding without Python's torch.distributed — aukit_partition + aukit_group_create / _scatter / _gather_* (csrc/group.hip).
    The members are `world` contexts on cuda:0 (a device may repeat in a group), so every line of the peer-copy transport runs on a one-GPU box:
    the batch lives on member `root`, every member gets its byte-balanced range device to device, runs the ordinary single-GPU calls in its own
    context, and the gathered rows / bytes are, bit for bit, those of the unsharded call."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    lens = [30000, 100, 4097, 1, 52000, 7, 9000, 1024, 2048, 33333, 5, 777, 0, 12]
    streams = [pcm16(n, 44100, 1, i).tobytes() for i, n in enumerate(lens)]
    desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
    single_ctx = ctx
    whole0 = B.Batch.upload(single_ctx, streams)
    single = B.decode_resample(single_ctx, whole0, desc, 48000, "cubic", dtype=N.F32).download()
    single_bytes = B.dfpwm_encode(single_ctx, B.decode_resample(single_ctx, whole0, desc, 48000, "cubic", dtype=N.F64), True).download()
    g = B.Group([0] * world, dtype=N.F32)
    try:
        assert g.transport() == "peer"
        whole = B.Batch.upload(g.contexts[root], streams)   # the job's bytes, resident on the root member
        shards, cuts = g.scatter(whole, root)
        assert cuts == B.partition([len(s) for s in streams], world)
        assert cuts[0][0] == 0 and cuts[-1][1] == len(streams) and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        for r, (lo, hi) in enumerate(cuts):
            assert shards[r].download() == streams[lo:hi]
        assert shards[root].device_ptr() == whole.device_ptr() + int(whole.offsets()[cuts[root][0]])   # the root's shard is a view
        parts = [B.decode_resample(g.contexts[r], shards[r], desc, 48000, "cubic", dtype=N.F32) for r in range(world)]
        got = g.gather_audio(parts, root)
        g.sync()
        rows = got.download()
        assert len(rows) == len(single)
        for a, b in zip(rows, single):
            assert np.array_equal(a[0], b[0])
        enc = [B.dfpwm_encode(g.contexts[r], B.decode_resample(g.contexts[r], shards[r], desc, 48000, "cubic", dtype=N.F64), True) for r in range(world)]
        allb = g.gather_batch(enc, root)
        g.sync()
        assert allb.download() == single_bytes
    finally:
        g.close()


def test_group_gather_of_deferred_audios(ctx, oracle):
    """A member's audio may still owe work when it is gathered — the resample `aukit_decode_resample` defers on FLAC (flac_tail.hip), a pending
    normalize: aukit_group_gather_audio finishes it in the member's context before the rows travel (round 3: it looked at the pending normalize only)."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    rng = np.random.default_rng(5)
    streams = []
    for i, n in enumerate((9000, 4097, 12000, 300)):
        x = np.stack([pcm16(n, 44100, 8, 2 * i + c) for c in range(2)], 1).ravel()
        streams.append(oracle.gen_flac(x, 2, 16, 44100, 1024))
    desc = B.make_desc(N.CODEC_FLAC, 2, 44100)
    single = B.decode_resample(ctx, B.Batch.upload(ctx, streams), desc, 48000, "cubic", dtype=N.F32)
    assert ctx.last_kernel()[0] == "(resample deferred)"
    single = single.download()
    g = B.Group([0, 0], dtype=N.F32)
    try:
        whole = B.Batch.upload(g.contexts[0], streams)
        shards, cuts = g.scatter(whole, 0)
        parts = [B.decode_resample(g.contexts[r], shards[r], desc, 48000, "cubic", dtype=N.F32) for r in range(2)]
        assert all(g.contexts[r].last_kernel()[0] == "(resample deferred)" for r in range(2))
        B.effect(g.contexts[1], parts[1], "normalize", 0.8)   # ... and a pending normalize on one member
        got = g.gather_audio(parts, 0)
        g.sync()
        rows = got.download()
        ref1 = B.decode_resample(ctx, B.Batch.upload(ctx, streams[cuts[1][0]:cuts[1][1]]), desc, 48000, "cubic", dtype=N.F32)
        B.effect(ctx, ref1, "normalize", 0.8)
        ref1 = ref1.download()
        assert len(rows) == len(single)
        for s in range(len(rows)):
            want = single[s] if s < cuts[0][1] else ref1[s - cuts[1][0]]
            for c in range(2):
                assert np.array_equal(rows[s][c], want[c]), (s, c)
    finally:
        g.close()


def test_group_rccl_transport_initialises(ctx, monkeypatch):
    """AUKIT_GROUP_TRANSPORT=rccl: librccl is dlopen'ed and ncclCommInitAll builds the group's communicators (distinct devices only — a one-GPU box
    has a group of one, whose scatter / gather move nothing: the messages between members cannot be exercised here, only the set-up and tear-down);
    a group with a repeated device keeps the peer-copy transport."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    monkeypatch.setenv("AUKIT_GROUP_TRANSPORT", "rccl")
    g = B.Group([0], dtype=N.F32)
    try:
        assert g.transport() == "rccl"
        streams = [pcm16(n, 44100, 1, i).tobytes() for i, n in enumerate([5000, 300, 41])]
        whole = B.Batch.upload(g.contexts[0], streams)
        shards, cuts = g.scatter(whole, 0)
        assert cuts == [(0, 3)] and shards[0].download() == streams
        desc = B.make_desc(N.CODEC_PCM, 1, 44100, 16, "signed")
        out = g.gather_audio([B.decode_resample(g.contexts[0], shards[0], desc, 48000, "cubic", dtype=N.F32)], 0)
        g.sync()
        want = B.decode_resample(ctx, B.Batch.upload(ctx, streams), desc, 48000, "cubic", dtype=N.F32).download()
        for a, b in zip(out.download(), want):
            assert np.array_equal(a[0], b[0])
    finally:
        g.close()
    g2 = B.Group([0, 0], dtype=N.F32)
    try:
        assert g2.transport() == "peer"
    finally:
        g2.close()


def test_group_run_members_side_by_side(ctx, oracle):
    """aukit_group_run.  Four members on cuda:0, each with a small FLAC shard — a codec whose entry point waits for its device
    five times per call (candidate count, chain, frame records ...).  Issued member after member from one host thread the four pipelines
    (aukit.flac -> resample -> highpass -> normalize -> mono: BASELINE config 5) run one AFTER another; through aukit_group_run the group's
    worker threads run them side by side: the members' intervals overlap, the wall time is nearer the longest member's than the sum, and the
    results are bit for bit those of the sequential calls."""
    import os, time
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "flac_44100_stereo_10s_0.bin")
    one = open(fx, "rb").read()
    W, per = 4, 24
    desc = B.make_desc(N.CODEC_FLAC)
    g = B.Group([0] * W, dtype=N.F32)
    try:
        whole = B.Batch.upload(g.contexts[0], [one] * (W * per))
        shards, cuts = g.scatter(whole, 0)
        g.sync()

        def sequential():
            outs = []
            for r in range(W):
                a = B.decode_resample(g.contexts[r], shards[r], desc, 48000, "cubic", dtype=N.F32)
                B.effect(g.contexts[r], a, "highpass", 20.0)
                B.effect(g.contexts[r], a, "normalize", 0.8)
                outs.append(B.mono(g.contexts[r], a))
            g.sync()
            return outs

        def lists(audios, monos):
            return [[{"op": "decode_resample", "batch": shards[r], "desc": desc, "new_rate": 48000, "interp": "cubic", "dtype": N.F32, "out": audios[r]},
                     {"op": "effect", "audio": audios[r], "name": "highpass", "args": (20.0,)},
                     {"op": "effect", "audio": audios[r], "name": "normalize", "args": (0.8,)},
                     {"op": "mono", "audio": audios[r], "out": monos[r]}] for r in range(W)]
        sequential()                                   # warm both ways: allocations, code objects
        audios = [B.AudioBatch(g.contexts[r]) for r in range(W)]
        monos = [B.AudioBatch(g.contexts[r]) for r in range(W)]
        # the effect entries name audios[r] before the decode has made it: run the decode alone once so that the handles exist
        g.run([[l[0]] for l in lists(audios, monos)])
        g.run(lists(audios, monos))
        t_seq = min(_timed(sequential) for _ in range(3))
        t_par = min(_timed(lambda: g.run(lists(audios, monos))) for _ in range(3))
        spans = g.last_run()
        assert max(s for s, _ in spans) < min(e for _, e in spans), spans      # every member was at work while every other one was
        longest = max(e - s for s, e in spans)
        assert (max(e for _, e in spans) - min(s for s, _ in spans)) < 1.5 * longest, spans
        assert t_par < 0.75 * t_seq, (t_par, t_seq, spans)
        want = sequential()
        for r in range(W):
            a, b = monos[r].download(), want[r].download()
            for s in range(len(a)):
                assert np.array_equal(a[s][0], b[s][0]), (r, s)
        # a failing member: its status and message come back, the others finish
        bad = B.Batch.upload(g.contexts[1], [b"not a flac file at all, but long enough to be read"])
        l2 = lists(audios, monos)
        l2[1][0]["batch"] = bad
        with pytest.raises(N.AukitError) as ei:
            g.run(l2)
        assert "Invalid magic string" in str(ei.value)
    finally:
        g.close()


def _timed(fn):
    import time
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_config4_and_config5_sharded_equal_single(ctx, oracle, world):
    """BASELINE configs 4 and 5 are worded as 8-GPU jobs: cut by input bytes (shard.partition), every shard through its own context, outputs back in
    rank order — the re-encoded DFPWM bytes (config 4: whatever the speculative transcoder's probe and rounds decide per shard, the bytes are the same) and
    the mono rows of the FLAC chain (config 5) equal the unsharded call's, and the oracle's."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    from aukit_amd import shard
    from tests.util import signal
    # config 4: eleven stereo DFPWM streams of six lengths (one with digital silence in front)
    st = []
    for i, n in enumerate((30000, 6000, 18016, 6001, 24690, 30000, 12000, 45000, 6002, 9000, 30001)):
        x = np.round(signal(8 * n, 48000, 4, 200 + i) * 100)
        if i == 5:
            x[: 8 * n // 3] = 0
        st.append(oracle.dfpwm_encode(x))
    whole = B.Batch.upload(ctx, st)
    starts = whole.offsets().astype(np.int64)
    single = B.dfpwm_transcode_mono(ctx, whole, 2).download()
    for s, g in zip(st, single):
        assert g == oracle.audio_dfpwm(oracle.mono(oracle.dfpwm(s, 2, 48000)), True)
    got = []
    for lo, hi in shard.partition([len(s) for s in st], world):
        if hi <= lo:   # (an empty share wraps nothing: the config-5 loop below has the same guard)
            continue
        c2 = B.Context(0)
        try:
            bt = B.Batch.wrap(c2, whole.device_ptr() + int(starts[lo]), (starts[lo:hi + 1] - starts[lo]).astype(np.uint64), keep=whole)
            got += B.dfpwm_transcode_mono(c2, bt, 2).download()
        finally:
            c2.close()
    assert got == single
    # config 5: five FLAC files through resample -> highpass -> normalize -> mono
    fl = [oracle.gen_flac(np.stack([pcm16(n, 44100, 5, 300 + 2 * i + c) for c in range(2)], 1).astype(np.int64).ravel(), 2, 16, 44100, 1152) for i, n in enumerate((9000, 3000, 12001, 4608, 7000))]
    whole = B.Batch.upload(ctx, fl)
    starts = whole.offsets().astype(np.int64)

    def chain(c, bt):
        a = B.decode_resample(c, bt, B.make_desc(N.CODEC_FLAC), 48000, "cubic", dtype=N.F32)
        B.effect(c, a, "highpass", 20.0)
        B.effect(c, a, "normalize", 0.8)
        return B.mono(c, a).download()

    single = chain(ctx, whole)
    for s, g in zip(fl, single):
        ref = oracle.mono(oracle.fx_normalize(oracle.fx_highpass(oracle.resample(oracle.flac(s), 48000, oracle.CUBIC), 20.0), 0.8)).data[0]
        assert len(g[0]) == len(ref) and np.sqrt(np.mean((g[0] - ref) ** 2)) <= 1e-6
    got = []
    for lo, hi in shard.partition([len(s) for s in fl], world):
        c2 = B.Context(0)
        try:
            if hi > lo:
                bt = B.Batch.wrap(c2, whole.device_ptr() + int(starts[lo]), (starts[lo:hi + 1] - starts[lo]).astype(np.uint64), keep=whole)
                got += chain(c2, bt)
        finally:
            c2.close()
    assert len(got) == len(single) and all(np.array_equal(a[0], b[0]) for a, b in zip(got, single))
