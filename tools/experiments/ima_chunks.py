"""Would config 3b's int16 intermediate stay in the memory-side cache if the batch went through decode -> filter in slices?  Times the step of bench.py's
ima_pipeline as it is (one decode launch, one filter launch over 4096 streams) against the same streams in 2 .. 16 slices, each decoded and filtered before
the next (AUKIT_RS_SEGS keeps the filter's wave count at ~4096).  GPU box: python3 tools/experiments/ima_chunks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aukit_amd import _native as N, batch as B

ctx = B.Context(0)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
blobs = [open(os.path.join(root, "bench_data", f"ima_22050_220x512_{i}.bin"), "rb").read() for i in range(4)]
n = 4096
for slices in (1, 2, 4, 8, 16):
    per = n // slices
    data = b"".join(blobs[i % 4] for i in range(per))
    offs = np.cumsum([0] + [len(blobs[i % 4]) for i in range(per)]).astype(np.uint64)
    xs, bts, outs = [], [], []
    for s in range(slices):
        x = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        xs.append(x)
        bts.append(B.Batch.wrap(ctx, x.data_ptr(), offs, keep=x))
        outs.append(B.AudioBatch(ctx))
    d = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
    os.environ["AUKIT_RS_SEGS"] = str(slices)

    def step():
        for s in range(slices):
            B.decode_resample(ctx, bts[s], d, 48000, "cubic", dtype=N.F32, out=outs[s])
            B.effect(ctx, outs[s], "lowpass", 11025.0)
    for _ in range(3):
        step()
    ctx.sync()
    best = 1e9
    for w in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    print(f"slices {slices:2d}: {best:.3f} ms per {n} streams", flush=True)
    del xs, bts, outs
