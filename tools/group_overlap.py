#!/usr/bin/env python3
"""aukit_group_run on one GPU: W members (contexts on cuda:0), each with a FLAC shard, the config-5 pipeline per member — member after member from
one host thread against the group's worker threads.  Prints wall times and the members' intervals (profiles/r04_group_overlap.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aukit_amd import _native as N, batch as B

one = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "flac_44100_stereo_10s_0.bin"), "rb").read()
desc = B.make_desc(N.CODEC_FLAC)
for W, per in ((4, 24), (4, 128), (8, 32), (2, 256)):
    g = B.Group([0] * W, dtype=N.F32)
    whole = B.Batch.upload(g.contexts[0], [one] * (W * per))
    shards, cuts = g.scatter(whole, 0)
    g.sync()
    audios = [B.AudioBatch(g.contexts[r]) for r in range(W)]
    monos = [B.AudioBatch(g.contexts[r]) for r in range(W)]

    def seq():
        for r in range(W):
            B.decode_resample(g.contexts[r], shards[r], desc, 48000, "cubic", dtype=N.F32, out=audios[r])
            B.effect(g.contexts[r], audios[r], "highpass", 20.0)
            B.effect(g.contexts[r], audios[r], "normalize", 0.8)
            B.mono(g.contexts[r], audios[r], out=monos[r])
        g.sync()

    lists = [[{"op": "decode_resample", "batch": shards[r], "desc": desc, "new_rate": 48000, "interp": "cubic", "dtype": N.F32, "out": audios[r]},
              {"op": "effect", "audio": audios[r], "name": "highpass", "args": (20.0,)},
              {"op": "effect", "audio": audios[r], "name": "normalize", "args": (0.8,)},
              {"op": "mono", "audio": audios[r], "out": monos[r]}] for r in range(W)]
    seq(); seq()
    g.run(lists); g.run(lists)

    def timed(f):
        t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
    ts = min(timed(seq) for _ in range(5))
    tp = min(timed(lambda: g.run(lists)) for _ in range(5))
    spans = g.last_run()
    print(f"{W} members x {per} FLAC streams (config-5 pipeline each): member after member {ts:7.2f} ms, aukit_group_run {tp:7.2f} ms ({ts / tp:.2f}x); "
          f"members' intervals [ms] " + " ".join(f"{s:.2f}-{e:.2f}" for s, e in spans))
    g.close()
