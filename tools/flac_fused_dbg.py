#!/usr/bin/env python3
"""debug aid: k_flac_decode (flac_fused.hip) against the first design (AUKIT_FLAC_NO_FUSED=1) and the oracle, first mismatches per stream / channel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import _native as N, batch as B
from oracle import oracle as O
O.build()
ctx = B.Context(0)


def pcm(n, ch, depth, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 44100
    full = 1 << (depth - 1)
    cols = []
    for c in range(ch):
        s = 0.4 * np.sin(2 * np.pi * (300 + 50 * c) * t) + rng.uniform(-0.2, 0.2, n)
        cols.append(np.clip(np.round(s * full), -full, full - 1).astype(np.int32))
    return np.stack(cols, 1)


bad = 0
for depth in (8, 16, 24):
    for ch in (1, 2):
        for bs in (4096, 1152, 256):
            ps = [pcm(n, ch, depth, 5 + i) for i, n in enumerate((bs * 13 + 999, 5000, bs, 100))]
            streams = [O.gen_flac(p.ravel(), ch, depth, 44100, bs) for p in ps]
            bt = B.Batch.upload(ctx, streams)
            os.environ.pop("AUKIT_FLAC_NO_FUSED", None)
            got = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
            print(f"depth {depth} ch {ch} bs {bs}: kernel {ctx.last_kernel()[0]} fused={ctx.counter(N.COUNTER_FLAC_FUSED)}")
            for si, (p, g) in enumerate(zip(ps, got)):
                for c in range(ch):
                    ref = p[:, c] / float(1 << depth)
                    if len(g[c]) != len(ref):
                        print(f"   stream {si} ch {c}: length {len(g[c])} vs {len(ref)}"); bad += 1; continue
                    w = np.nonzero(g[c] != ref)[0]
                    if len(w):
                        bad += 1
                        fr = w[0] // bs
                        print(f"   stream {si} ch {c}: {len(w)} wrong of {len(ref)}, first at {w[0]} (frame {fr}, index {w[0] % bs}), frames hit {sorted(set((w // bs).tolist()))[:12]}; got {g[c][w[0]] * (1 << depth)} want {ref[w[0]] * (1 << depth)}")
print("mismatching rows:", bad)
