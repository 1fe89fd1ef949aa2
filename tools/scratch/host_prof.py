import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16
which = sys.argv[1]
ctx = B.Context(0, dtype=N.F32)
n = 1024
base = [np.stack([pcm16(441000, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel() for i in range(4)]
if which == "qoa":
    qs = [O.gen_qoa(base[i], 2, 44100) + b"\0" * 8 for i in range(4)]
    bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(n)]); d = B.make_desc(N.CODEC_QOA, 2, 44100)
else:
    fs = [O.gen_flac(base[i], 2, 16, 44100, 4096) for i in range(4)]
    bt = B.Batch.upload(ctx, [fs[i % 4] for i in range(n)]); d = B.make_desc(N.CODEC_FLAC, 2, 44100)
for mono in (False, True):
    out = None
    for i in range(4):
        t0 = time.time()
        out, ck = B.stream_decode(ctx, bt, d, "cubic", mono=mono, dtype=N.F32, out=out)
        t1 = time.time(); ctx.sync(); t2 = time.time()
        print(which, "mono" if mono else "stereo", f"call returns after {(t1 - t0) * 1e3:.1f} ms, GPU done after {(t2 - t0) * 1e3:.1f} ms")
