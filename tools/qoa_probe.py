#!/usr/bin/env python3
"""a few calls of the QOA / FLAC stream paths for rocprofv3 --kernel-trace --stats (GPU box): python tools/qoa_probe.py [which] [streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16
which = sys.argv[1] if len(sys.argv) > 1 else "qoa1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = B.Context(0, dtype=N.F32)
base = [np.stack([pcm16(44100 * 10, 44100, 8, 4 * i + c) for c in range(2)], 1) for i in range(4)]
if which.startswith("qoa"):
    ch = int(which[3])
    qs = [O.gen_qoa(base[i][:, :ch].ravel(), ch, 44100) + b"\0" * 8 for i in range(4)]
    bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(n)])
    desc = B.make_desc(N.CODEC_QOA, ch, 44100)
    a = b = c = None
    for _ in range(4):
        a = B.decode(ctx, bt, desc, dtype=N.F32, out=a)
        b, ck = B.stream_decode(ctx, bt, desc, "cubic", mono=False, dtype=N.F32, out=b)
        if ch == 2: c, ck = B.stream_decode(ctx, bt, desc, "cubic", mono=True, dtype=N.F32, out=c)
    ctx.sync()
elif which == "flac":
    fs = [O.gen_flac(base[i].ravel(), 2, 16, 44100, 4096) for i in range(4)]
    bt = B.Batch.upload(ctx, [fs[i % 4] for i in range(n)])
    desc = B.make_desc(N.CODEC_FLAC, 2, 44100)
    b = None
    for _ in range(4):
        b, ck = B.stream_decode(ctx, bt, desc, "cubic", mono=False, dtype=N.F32, out=b)
    ctx.sync()
