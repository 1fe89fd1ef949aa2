#!/usr/bin/env python3
"""stream.flac on 1024 copies... of 4 encoder-made stereo files: wall time per call and (AUKIT_HOST_TIMING=1) the host laps (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = B.Context(0, dtype=N.F32)
one = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "flac_44100_stereo_10s_0.bin"), "rb").read()
bt = B.Batch.upload(ctx, [one] * n)
desc = B.make_desc(N.CODEC_FLAC, 2, 44100)
out = None
for i in range(6):
    ctx.sync(); t0 = time.time()
    out, ck = B.stream_decode(ctx, bt, desc, "cubic", dtype=N.F32, out=out)
    ctx.sync()
    print(f"call {i}: {(time.time() - t0) * 1e3:.2f} ms  ({ctx.last_kernel()[0]})", flush=True)
