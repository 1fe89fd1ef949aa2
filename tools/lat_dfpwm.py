"""Latency of aukit.dfpwm on ONE 10-second stream: chunk-parallel exact decoder vs one lane per stream (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from aukit_amd import batch as B, _native as N
ctx = B.Context()
rng = np.random.Generator(np.random.PCG64(3))
data = rng.integers(0, 256, 120000, dtype=np.uint8).tobytes()
bt = B.Batch.upload(ctx, [data])
desc = B.make_desc(N.CODEC_DFPWM, 2, 48000)
for mode in ("parallel", "serial"):
    if mode == "serial":
        os.environ["AUKIT_DFPWM_SERIAL"] = "1"
    out = B.AudioBatch(ctx)
    for _ in range(3):
        B.decode(ctx, bt, desc, dtype=N.F32, out=out)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        B.decode(ctx, bt, desc, dtype=N.F32, out=out)
    ctx.sync()
    print(mode, "%.3f ms per 120 000-byte stream" % ((time.perf_counter() - t0) / 20 * 1e3))
