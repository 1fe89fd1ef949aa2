import sys, time; sys.path.insert(0, ".")
import numpy as np
from aukit_amd import batch as B, _native as N
ctx = B.Context(0, dtype=N.F32)
rng = np.random.default_rng(5)
n = 4096
for bits, dt, ch, sr, mono in ((8, "unsigned", 1, 48000, False), (8, "unsigned", 2, 48000, False), (8, "unsigned", 2, 48000, True), (8, "unsigned", 1, 44100, False), (16, "signed", 1, 48000, False), (16, "signed", 2, 48000, False), (8, "signed", 1, 48000, False)):
    nbytes = sr * 10 * ch * bits // 8
    bt = B.Batch.upload(ctx, [rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()] * n)
    d = B.make_desc(N.CODEC_PCM, ch, sr, bits, dt)
    out = None
    for i in range(2): out, ck = B.stream_decode(ctx, bt, d, "linear", mono=mono, dtype=N.F32, out=out)
    ctx.sync(); t0 = time.time()
    for i in range(3): out, ck = B.stream_decode(ctx, bt, d, "linear", mono=mono, dtype=N.F32, out=out)
    ctx.sync(); dt_ = (time.time() - t0) / 3
    outs = float(np.sum(ck.lens)) * (1 if mono else ch)
    print(f"stream.pcm {bits}-bit {dt} {ch}ch {sr} Hz{' mono' if mono else ''}: {dt_ * 1e3:.2f} ms, {outs / dt_ / 1e9:.0f} G out-samples/s ({ctx.last_kernel()[0]})", flush=True)
    del bt, out
