"""CPU: the exact-arithmetic shortcuts of the kernels, the sharding helpers, and the host-side chunk logic."""
import ctypes as C

import numpy as np
import pytest


def test_reciprocal_division_is_exact_for_common_rates(oracle):
    """x = (i-1)/ratio is computed on the GPU as fma(fma(-d, n*r, n), r, n*r) with r = RN(1/d): must equal n/d for every output."""
    L = oracle.lib()
    L.ork_check_div_rcp.restype = C.c_uint64
    for sr in (8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 192000, 37800, 7333):
        assert L.ork_check_div_rcp(C.c_double(48000 / sr), C.c_uint64(1 << 21)) == 0, sr
    for new in (44100, 8000, 22050):  # arbitrary target rates
        assert L.ork_check_div_rcp(C.c_double(new / 48000), C.c_uint64(1 << 20)) == 0


def test_sample_normalisation_by_reciprocal_is_exact(oracle):
    L = oracle.lib()
    L.ork_check_div_rcp.restype = C.c_uint64
    assert L.ork_check_div_rcp(C.c_double(32767.0), C.c_uint64(32768)) == 0   # s16: s / 32767
    assert L.ork_check_div_rcp(C.c_double(127.0), C.c_uint64(128)) == 0       # s8


def test_pow3_double_double_is_correctly_rounded(oracle):
    L = oracle.lib()
    L.ork_check_pow3.restype = C.c_uint64
    assert L.ork_check_pow3(C.c_uint64(1_000_000), C.c_uint64(7)) == 0


def test_rational_positions_agree_with_reference_doubles():
    """fast kernels: floor(x) and frac from (o*a) divmod b vs the reference's double x = o/ratio + 1 (differences ≤ 1e-10)."""
    for sr, new in ((44100, 48000), (8000, 48000), (22050, 48000), (48000, 44100), (11025, 48000)):
        g = np.gcd(sr, new)
        a, b = sr // g, new // g
        o = np.arange(0, 600000, dtype=np.int64)
        x = o.astype(np.float64) / (new / sr) + 1
        k, r = np.divmod(o * a, b)
        xr = k + 1 + r / b
        assert np.max(np.abs(x - xr)) < 1e-9
        magic = (2 ** 32 + b - 1) // b
        n = (r[:5000] + np.arange(5000) * a).astype(np.uint64)
        assert np.array_equal((n * np.uint64(magic)) >> np.uint64(32), n // np.uint64(b))


def test_dfpwm_step_algebra_is_the_published_step_for_every_state():
    """dfpwm_dev.h rewrites the DFPWM1a step for instruction count (median-of-three nudge, ±1 bits, n = -(2 charge + 1), low-pass
    with the lpf inside the floor, biased charge in the encoder).  Every identity it claims, checked exhaustively in integer numpy
    against the published form — independent of the GPU and of the C oracle."""
    def med3(a, b, c):
        return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))

    charge, strength, bit, prev = np.meshgrid(np.arange(-128, 128), np.concatenate([[0], np.arange(8, 1024)]), [0, 1], [0, 1], indexing="ij")
    charge, strength, bit, prev = [x.ravel().astype(np.int64) for x in (charge, strength, bit, prev)]
    # published predictor
    target = np.where(bit == 1, 127, -128)
    nxt = charge + ((strength * (target - charge) + 512) >> 10)
    nxt = np.where((nxt == charge) & (nxt != target), nxt + np.where(bit == 1, 1, -1), nxt)
    z = np.where(bit == prev, 1023, 0)
    ns = np.where(strength != z, strength + np.where(bit == prev, 1, -1), strength)
    ns = np.maximum(ns, 8)
    # decoder form
    b, pb, n = 2 * bit - 1, 2 * prev - 1, -(2 * charge + 1)
    diff2 = 255 * b + n
    assert np.array_equal(diff2, 2 * (target - charge))
    step = med3((strength * diff2 + 1024) >> 11, b, diff2)
    n2 = n - 2 * step
    assert np.array_equal((~n2) >> 1, nxt)
    assert np.array_equal(med3(b * pb + strength, 8, 1023), ns)
    # encoder form: biased charge, unsigned samples
    cu, diff = charge + 128, np.where(bit == 1, 255, 0) - (charge + 128)
    assert np.array_equal(cu + med3((strength * diff + 512) >> 10, b, diff) - 128, nxt)
    v, c = [x.ravel().astype(np.int64) for x in np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")]
    assert np.array_equal((v > c) | ((v == c) & (v == 127)), (v + 128) > np.minimum(c + 128, 254))
    # anti-jerk + low-pass: q = ceil((n + (same ? n : previous n)) / 4) = -(anti-jerked charge); lpf' = (116 lpf - 140 aj + 128) >> 8
    c1, c0, lpf, same = [x.ravel().astype(np.int64) for x in np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), np.arange(-128, 128, 3), [0, 1], indexing="ij")]
    aj = np.where(same == 1, c1, (c1 + c0 + 1) >> 1)
    ref = lpf + (((aj - lpf) * 140 + 0x80) >> 8)
    n1, n0 = -(2 * c1 + 1), -(2 * c0 + 1)
    q = (n1 + np.where(same == 1, n1, n0) + 3) >> 2
    assert np.array_equal(q, -aj)
    assert np.array_equal((116 * lpf + (-140 * q + 128)) >> 8, ref)


def test_partition_is_contiguous_and_balanced():
    from aukit_amd.shard import partition
    rng = np.random.default_rng(0)
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 4096):
            sizes = rng.integers(1, 100000, n)
            parts = partition(sizes, world)
            assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[g][1] == parts[g + 1][0] for g in range(world - 1))
            if n >= 64 * world:
                loads = [int(sizes[a:b].sum()) for a, b in parts]
                assert max(loads) - min(loads) <= 2 * int(sizes.max())
    assert partition([5, 5, 5, 5], 2) == [(0, 2), (2, 4)]
    assert partition(np.zeros(6), 3) == [(0, 2), (2, 4), (4, 6)]


def _worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aukit_amd import shard
    from oracle import oracle as O
    from tests.util import pcm16
    streams = [pcm16(2000 + 300 * i, 44100, 1, i).tobytes() for i in range(9)] if rank == 0 else None
    mine, (lo, hi) = shard.scatter_streams(streams, src=0)
    # stand-in for the per-GPU path: the CPU oracle (tests may use it as the checker and as the shard worker)
    outs = []
    for s in mine:
        r = O.resample(O.pcm(s, 16, O.SIGNED, 1, 44100), 48000, O.CUBIC)
        outs.append(r.data[0].astype(np.float32).tobytes())
    gathered = shard.gather_streams(outs, dst=0)
    if rank == 0:
        ref = [O.resample(O.pcm(s, 16, O.SIGNED, 1, 44100), 48000, O.CUBIC).data[0].astype(np.float32).tobytes() for s in streams]
        q.put((gathered == ref, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_process_gather_world2_gloo():
    """N > 1 path: gather(process(scatter(batch))) == process(batch), two processes over gloo on CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_worker, args=(r, 2, int(port), q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, lo, hi = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and lo == 0 and 0 < hi < 9


def _worker_transport(rank, world, port, q):
    """the device face's transport (aukit_amd.shard._scatter_bytes / _gather_bytes_with_table) on CPU tensors under gloo: the same code
    moves cuda tensors under nccl; here it must deliver exactly the partition's byte ranges, views for the source rank, and opaque tables"""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aukit_amd import shard
    dev = torch.device("cpu")
    rng = np.random.Generator(np.random.PCG64(99))
    sizes = [0, 700, 1, 1300, 0, 64, 5000, 3, 900][: 9 if world < 3 else 2]  # world 3: two streams only, one rank gets an empty shard
    blob = rng.integers(0, 256, int(sum(sizes)), dtype=np.uint8)
    flat = torch.from_numpy(blob.copy()) if rank == 0 else None
    mine, sz, (lo, hi) = shard._scatter_bytes(flat, sizes if rank == 0 else None, 0, dev, None)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    ok = list(sz) == sizes[lo:hi] and np.array_equal(mine.numpy(), blob[starts[lo]:starts[hi]])
    if rank == 0 and mine.numel():
        ok = ok and mine.data_ptr() == flat.data_ptr() + int(starts[lo])  # the source's own shard is a view, not a copy
    # every rank "processes" its shard (doubles every byte mod 256) and the results travel back with a layout table
    table = np.array([rank, lo, hi, -7], dtype=np.int64)
    got = shard._gather_bytes_with_table((mine * 2), table, 0, dev, None)
    if rank == 0:
        parts = shard.partition(sizes, world)
        whole = np.concatenate([g[0].numpy() for g in got])
        ok = ok and np.array_equal(whole, (blob * 2).astype(np.uint8)) and [list(g[1]) for g in got] == [[r, a, b, -7] for r, (a, b) in enumerate(parts)]
        q.put(bool(ok))
    else:
        assert got is None and ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_device_face_transport_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_worker_transport, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_partition_matches_the_numpy_statement():
    """aukit_partition against the midpoint rule written out in numpy (what shard.py computed before it called the library)"""
    from aukit_amd import batch as B
    rng = np.random.Generator(np.random.PCG64(5))
    for trial in range(200):
        n = int(rng.integers(0, 40))
        world = int(rng.integers(1, 10))
        sizes = rng.integers(0, 5000, n).astype(np.float64) * (rng.integers(0, 2, n) if trial % 3 == 0 else 1)
        if n == 0:
            want = [(0, 0)] * world
        elif sizes.sum() <= 0:
            cuts = [(n * g) // world for g in range(world + 1)]
            want = [(cuts[g], cuts[g + 1]) for g in range(world)]
        else:
            mid = np.cumsum(sizes) - sizes / 2
            owner = np.minimum((mid / sizes.sum() * world).astype(np.int64), world - 1)
            cuts = [int(np.searchsorted(owner, g, side="left")) for g in range(world)] + [n]
            want = [(cuts[g], cuts[g + 1]) for g in range(world)]
        assert B.partition(sizes.astype(np.uint64), world) == want, (trial, n, world)


def test_partition_numpy_fallback_agrees_with_the_library():
    """shard.partition falls back to numpy on a host without the built library: the two must cut alike"""
    from aukit_amd import batch as B
    from aukit_amd.shard import _partition_numpy
    rng = np.random.Generator(np.random.PCG64(9))
    for trial in range(300):
        n = int(rng.integers(0, 60))
        world = int(rng.integers(1, 10))
        sizes = rng.integers(0, 9000, n) * (rng.integers(0, 2, n) if trial % 4 == 0 else 1)
        assert _partition_numpy(sizes, world) == B.partition(sizes, world), (list(sizes), world)


def _worker_flows(rank, world, port, q):
    """BASELINE configs 4 and 5 as 8-GPU jobs: rank 0 holds the byte batch, shard.scatter_streams cuts it by input bytes, every rank
    runs the whole chain on its shard (stand-in for the GPU: the oracle), the outputs — re-encoded DFPWM bytes (shard.gather_batch's transport:
    sizes travel as a table) and mono f32 rows — come back in stream order.  gather(process(scatter(batch))) == process(batch), also with a rank
    whose shard is empty (world 3, two streams)."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aukit_amd import shard
    from oracle import oracle as O
    from tests.util import pcm16, signal
    nst = 5 if world < 3 else 2

    def transcode(s):   # config 4: a = aukit.dfpwm(d, 2, 48000); m = a:mono(); out = m:dfpwm()
        return O.audio_dfpwm(O.mono(O.dfpwm(s, 2, 48000)), True)

    def pipeline(s):    # config 5: aukit.flac(d):resample(48000, "cubic") -> highpass(20) -> normalize(0.8) -> mono
        a = O.fx_normalize(O.fx_highpass(O.resample(O.flac(s), 48000, O.CUBIC), 20.0), 0.8)
        return O.mono(a).data[0].astype(np.float32).tobytes()

    ok = True
    for make, work in ((lambda i: O.dfpwm_encode(np.round(signal(16 * (700 + 310 * i), 48000, 4, i) * 100)), transcode),
                       (lambda i: O.gen_flac(np.stack([pcm16(3000 + 900 * i, 44100, 5, 2 * i + c) for c in range(2)], 1).astype(np.int64).ravel(), 2, 16, 44100, 1152), pipeline)):
        streams = [make(i) for i in range(nst)] if rank == 0 else None
        mine, (lo, hi) = shard.scatter_streams(streams, src=0)
        if world == 3:
            ok = ok and (hi - lo) in (0, 1)
        gathered = shard.gather_streams([work(s) for s in mine], dst=0)
        if rank == 0:
            ok = ok and gathered == [work(s) for s in streams]
            ok = ok and [(a, b) for a, b in shard.partition([len(s) for s in streams], world)][0] == (lo, hi)
        else:
            ok = ok and gathered is None
    if rank == 0:
        q.put(bool(ok))
    else:
        assert ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config4_and_config5_flows_scatter_process_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_worker_flows, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_dfpwm_encoder_class_invariant():
    """what dfpwm_spec.hip's guesses rest on (its header): the DFPWM1a encoder's I = (strength - 2 [previous bit = 0] - t) mod 4 survives every step
    that does not clamp the strength.  A Python restatement of the encoder step (checked against the oracle's bytes first) run over noise, a tone and
    silence: the invariant holds on every unclamped step, and changes only where the strength was clamped at 8 or 1023."""
    from oracle import oracle as O
    rng = np.random.Generator(np.random.PCG64(42))
    n = 6000
    t = np.arange(n)
    for x in (rng.integers(-128, 128, n), np.round(100 * np.sin(t * 0.05)).astype(np.int64), np.zeros(n, dtype=np.int64),
              np.where((t // 700) % 2 == 0, np.round(90 * np.sin(t * 0.3)), 0).astype(np.int64)):
        charge, strength, prev = 0, 0, False
        bits, unclamped, kept, clamped, changed = [], 0, 0, 0, 0
        for i, v in enumerate(x):
            v = int(v)
            inv_before = (strength - (0 if prev else 2) - i) % 4
            bit = v > charge or (v == charge and v == 127)
            target = 127 if bit else -128
            nxt = charge + (strength * (target - charge) + 512) // 1024
            if nxt == charge and nxt != target:
                nxt += 1 if bit else -1
            raw = strength + (1 if bit == prev else -1)
            ns = min(max(raw, 8), 1023)
            charge, strength, prev = nxt, ns, bit
            bits.append(bit)
            inv_after = (strength - (0 if prev else 2) - (i + 1)) % 4
            if raw == ns:
                unclamped += 1
                kept += inv_after == inv_before
            else:
                clamped += 1
                changed += inv_after != inv_before
        by = bytes(sum(int(b) << k for k, b in enumerate(bits[j:j + 8])) for j in range(0, n, 8))
        assert by == O.dfpwm_encode(x.astype(np.float64))   # the restatement is the oracle's encoder
        assert kept == unclamped and changed == clamped, (unclamped, kept, clamped, changed)
