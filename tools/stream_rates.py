#!/usr/bin/env python3
"""Throughput of every aukit.stream.* iterator (all calls of a batch at once, aukit_stream_decode) on one large batch (GPU box).
usage: python tools/stream_rates.py [streams=1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DT = N.F64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else N.F32
ctx = B.Context(0, dtype=DT)
rng = np.random.default_rng(5)
def rate(name, bt, desc, interp, mono, dtype):
    try:
        keep = [None]
        def f():
            keep[0], ck = B.stream_decode(ctx, bt, desc, interp, mono=mono, dtype=dtype, out=keep[0])
            return ck
        for _ in range(4): ck = f(); ctx.sync()  # untimed calls: buffers, pinned staging and the runtime's own pools grow on the first ones (round 4: two were not enough —
        # stream.mdfpwm's second call still spent 13 ms pinning its header staging, and the 8.7 ms per call the survey printed was that, not the 2.8 ms of its kernels)
        ts = []
        for _ in range(5):
            t0 = time.time(); ck = f(); ctx.sync(); ts.append(time.time() - t0)
        dt = sorted(ts)[len(ts) // 2]
        outs = float(np.sum(ck.lens)) * (1 if mono else max(1, desc.channels))
        print(f"{name:44s} {dt * 1e3:8.2f} ms  {outs / dt / 1e9:8.1f} G out-samples/s  ({ctx.last_kernel()[0]})", flush=True)
    except Exception as e:
        print(f"{name:44s} failed: {str(e)[:90]}", flush=True)
sec = 10
base_pcm = [np.stack([pcm16(44100 * sec, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel() for i in range(4)]
# stream.pcm stereo 16-bit 44.1k
bt = B.Batch.upload(ctx, [base_pcm[i % 4].astype("<i2").tobytes() for i in range(n)])
rate("stream.pcm s16le stereo 44.1k cubic f32", bt, B.make_desc(N.CODEC_PCM, 2, 44100, 16, "signed"), "cubic", False, DT)
rate("stream.pcm s16le stereo 44.1k cubic mono f32", bt, B.make_desc(N.CODEC_PCM, 2, 44100, 16, "signed"), "cubic", True, DT)
del bt
bt = B.Batch.upload(ctx, [rng.integers(0, 256, 44100 * sec * 3, dtype=np.uint8).tobytes()] * n)
rate("stream.pcm 24-bit mono 44.1k linear f32", bt, B.make_desc(N.CODEC_PCM, 1, 44100, 24, "signed"), "linear", False, DT)
del bt
bt = B.Batch.upload(ctx, [rng.integers(0, 256, 60000 * 2, dtype=np.uint8).tobytes()] * n)
rate("stream.dfpwm stereo 48k (f32)", bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), "linear", False, DT)
rate("stream.dfpwm stereo 48k mono (f32)", bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), "linear", True, DT)
rate("stream.dfpwm mono 32k cubic (f32)", bt, B.make_desc(N.CODEC_DFPWM, 1, 32000), "cubic", False, DT)
del bt
md = O.gen_mdfpwm(rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 60000, dtype=np.uint8).tobytes())
bt = B.Batch.upload(ctx, [md] * n)
rate("stream.mdfpwm (i8; both channels counted)", bt, B.make_desc(N.CODEC_MDFPWM, 2, 48000), "linear", False, N.I8)
rate("stream.mdfpwm mono (i8)", bt, B.make_desc(N.CODEC_MDFPWM, 2, 48000), "linear", True, N.I8)
del bt
for ch in (1, 2):
    ba = 1024
    raw = rng.integers(0, 256, (430, ba), dtype=np.uint8); raw[:, 0] = 0
    if ch == 2: raw[:, 1] = 1
    bt = B.Batch.upload(ctx, [raw.tobytes()] * n)
    rate(f"stream.msadpcm {ch}ch 44.1k cubic (i8)", bt, B.make_desc(N.CODEC_MSADPCM, ch, 44100, block_align=ba), "cubic", False, N.I8)
    del bt
qs = [O.gen_qoa(base_pcm[i], 2, 44100) + b"\0" * 8 for i in range(4)]
bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(n)])
rate("stream.qoa stereo 44.1k cubic (f32)", bt, B.make_desc(N.CODEC_QOA, 2, 44100), "cubic", False, DT)
rate("stream.qoa stereo 44.1k cubic mono (f32)", bt, B.make_desc(N.CODEC_QOA, 2, 44100), "cubic", True, DT)
del bt
fs = [O.gen_flac(base_pcm[i], 2, 16, 44100, 4096) for i in range(4)]
bt = B.Batch.upload(ctx, [fs[i % 4] for i in range(n)])
rate("stream.flac stereo 44.1k cubic (f32)", bt, B.make_desc(N.CODEC_FLAC, 2, 44100), "cubic", False, DT)
del bt
im = O.gen_ima(base_pcm[0][::2].copy(), 1, 512)
bt = B.Batch.upload(ctx, [im] * n)
rate("stream.adpcm mono 22.05k cubic (i8)", bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), "cubic", False, N.I8)
bt2 = B.Batch.upload(ctx, [rng.integers(0, 256, 80000, dtype=np.uint8).tobytes()] * n)
rate("stream.g711 stereo 8k cubic (i8)", bt2, B.make_desc(N.CODEC_G711, 2, 8000, ulaw=True), "cubic", False, N.I8)
