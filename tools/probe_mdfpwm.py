import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = B.Context(0, dtype=N.F32)
rng = np.random.default_rng(5)
md = O.gen_mdfpwm(rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 60000, dtype=np.uint8).tobytes())
bt = B.Batch.upload(ctx, [md] * n)
d = B.make_desc(N.CODEC_MDFPWM)
out = None
for i in range(4):
    t0 = time.time(); out, ck = B.stream_decode(ctx, bt, d, "linear", mono=False, dtype=N.I8, out=out); t1 = time.time(); ctx.sync(); t2 = time.time()
    print(f"call {(t1-t0)*1e3:.2f} ms, +sync {(t2-t1)*1e3:.2f} ms", file=sys.stderr)
