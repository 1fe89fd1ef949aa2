#!/usr/bin/env python3
"""Throughput of the loaders, alone and with a fused resample to 48 kHz (aukit_decode / aukit_decode_resample), per format, on one large
batch (GPU box).  usage: python tools/loader_rates.py [streams=1024] [f32|f64] [exact_math=0|1|2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dtype = N.F64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else N.F32
ctx = B.Context(0, dtype=dtype)
if len(sys.argv) > 3: ctx.set_option(N.OPT_EXACT_MATH, int(sys.argv[3]))
rng = np.random.default_rng(9)
def rate(name, bt, d, outs_per_stream_dec, rate_in):
    for label, fn, outs in (("decode", lambda o: B.decode(ctx, bt, d, out=o), outs_per_stream_dec),
                            ("decode+resample 48k cubic", lambda o: B.decode_resample(ctx, bt, d, 48000, "cubic", out=o), outs_per_stream_dec * 48000 / rate_in)):
        try:
            # (round 4: a deferred resample is PAID inside the timed call — device_ptr() materialises it — so that the line times the work, not the promise)
            o = fn(None); o.device_ptr(); ctx.sync()
            for _ in range(3): o = fn(o); o.device_ptr(); ctx.sync()
            ts = []
            for _ in range(5):
                t0 = time.time(); o = fn(o); o.device_ptr(); ctx.sync(); ts.append(time.time() - t0)
            dt = sorted(ts)[len(ts) // 2]
            print(f"{name:30s} {label:26s} {dt * 1e3:8.2f} ms  {n * outs / dt / 1e9:8.1f} G samples/s  ({ctx.last_kernel()[0]})", flush=True)
            del o
        except Exception as e:
            print(f"{name:30s} {label:26s} failed: {str(e)[:80]}", flush=True)
sec = 10
base = [np.stack([pcm16(44100 * sec, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel() for i in range(4)]
for bits, dt, ch in ((16, "signed", 2), (8, "unsigned", 1), (24, "signed", 2), (32, "float", 1), (16, "signed", 1)):
    nbytes = 44100 * sec * ch * bits // 8
    bt = B.Batch.upload(ctx, [rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes() if dt != "float" else rng.uniform(-1, 1, 44100 * sec * ch).astype("<f4").tobytes()] * n)
    rate(f"pcm {bits}-bit {dt} {ch}ch", bt, B.make_desc(N.CODEC_PCM, ch, 44100, bits, dt), 44100 * sec * ch, 44100)
    del bt
bt = B.Batch.upload(ctx, [rng.integers(0, 256, 80000 * 2, dtype=np.uint8).tobytes()] * n)
rate("g711 ulaw 2ch 8k", bt, B.make_desc(N.CODEC_G711, 2, 8000, ulaw=True), 160000, 8000); del bt
im = O.gen_ima(base[0][::2].copy(), 1, 512)
bt = B.Batch.upload(ctx, [im] * n)
rate("adpcm (IMA in WAV) mono", bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 44100, block_align=512), len(im) // 512 * 1016, 44100); del bt
raw = rng.integers(0, 256, (430, 1024), dtype=np.uint8); raw[:, 0] = 0; raw[:, 1] = 1
bt = B.Batch.upload(ctx, [raw.tobytes()] * n)
rate("msadpcm 2ch", bt, B.make_desc(N.CODEC_MSADPCM, 2, 44100, block_align=1024), 430 * 1012 * 2, 44100); del bt
bt = B.Batch.upload(ctx, [rng.integers(0, 256, 120000, dtype=np.uint8).tobytes()] * n)
rate("dfpwm 2ch", bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), 960152, 48000); del bt
qs = [O.gen_qoa(base[i], 2, 44100) + b"\0" * 8 for i in range(4)]
bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(n)])
rate("qoa 2ch", bt, B.make_desc(N.CODEC_QOA, 2, 44100), 441000 * 2, 44100); del bt
fs = [O.gen_flac(base[i], 2, 16, 44100, 4096) for i in range(4)]
bt = B.Batch.upload(ctx, [fs[i % 4] for i in range(n)])
rate("flac 2ch", bt, B.make_desc(N.CODEC_FLAC, 2, 44100), 441000 * 2, 44100); del bt
