#!/usr/bin/env python3
"""Throughput of every Audio method / effect / output op of the C ABI on one large batch (GPU box): spots the ops that sit far below the
HBM rate.  usage: python tools/surface_rates.py [streams=1024] [f32|f64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dtype = N.F64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else N.F32
esz = 8 if dtype == N.F64 else 4
ctx = B.Context(0, dtype=dtype)
rng = np.random.default_rng(3)
pcm = (rng.standard_normal(480000 * 2) * 6000).astype(np.int16).tobytes()
bt = B.Batch.upload(ctx, [pcm] * n)
d = B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed")
au = B.decode(ctx, bt, d)
elems = n * 2 * 480000
def rate(name, fn, gbytes, reps=3):
    try:
        fn(); ctx.sync()
        t0 = time.time()
        for _ in range(reps): fn()
        ctx.sync(); dt = (time.time() - t0) / reps
        print(f"{name:34s} {dt * 1e3:8.2f} ms  {gbytes / dt / 1e3:6.2f} TB/s  ({ctx.last_kernel()[0]})", flush=True)
    except Exception as e:
        print(f"{name:34s} failed: {str(e)[:80]}", flush=True)
rw = 2 * elems * esz / 1e9
keep = {}
def reuse(key, f):
    def g():
        keep[key] = f(keep.get(key))
    return g
for fx, args in (("amplify", (0.5,)), ("invert", ()), ("normalize", (0.8,)), ("center", (48000,)), ("fade", (1.0, 2.0, 0.2, 1.0)), ("lowpass", (4000.0,)), ("highpass", (20.0,)),
                 ("delay", (0.05, 0.5)), ("echo", (0.05, 0.5)), ("reverb", ())):
    rate("effects." + fx, lambda fx=fx, args=args: B.effect(ctx, au, fx, *args), rw)
rate("Audio:mono", reuse("m", lambda o: B.mono(ctx, au, out=o)), 1.5 * elems * esz / 1e9)
rate("Audio:mix (2 audios)", reuse("x", lambda o: B.mix(ctx, [au, au], 0.5, out=o)), 3 * elems * esz / 1e9)
rate("Audio:resample 44.1k linear", reuse("r", lambda o: B.resample(ctx, au, 44100, "linear", out=o)), (1 + 0.91875) * elems * esz / 1e9)
rate("Audio:resample 96k cubic", reuse("r2", lambda o: B.resample(ctx, au, 96000, "cubic", out=o)), 3 * elems * esz / 1e9)
rate("Audio:pcm 16 signed (numbers)", reuse("p", lambda o: B.encode_pcm(ctx, au, 16, "signed", True, out=o)), elems * (esz + 8) / 1e9)
rate("aukit.pack 16-bit LE interleaved", reuse("k", lambda o: B.pack_pcm(ctx, au, 16, "signed", False, True, out=o)), elems * (esz + 2) / 1e9)
rate("aukit.pack 8-bit unsigned", reuse("k8", lambda o: B.pack_pcm(ctx, au, 8, "unsigned", False, True, out=o)), elems * (esz + 1) / 1e9)
rate("aukit.pack 24-bit BE planar", reuse("k24", lambda o: B.pack_pcm(ctx, au, 24, "signed", True, False, out=o)), elems * (esz + 3) / 1e9)
rate("Audio:dfpwm", reuse("e", lambda o: B.dfpwm_encode(ctx, au, True, out=o)), elems * (esz + 0.125) / 1e9, reps=1)
rate("Audio:concat (2)", reuse("c", lambda o: B.concat(ctx, [au, au], out=o)), 4 * elems * esz / 1e9)
rate("Audio:sub 2 s .. 8 s", reuse("s", lambda o: B.sub(ctx, au, 2.0, 8.0, out=o)), 2 * 0.6 * elems * esz / 1e9)
rate("Audio:reverse", reuse("v", lambda o: B.reverse(ctx, au, out=o)), rw)
rate("Audio:rep 2", reuse("rp", lambda o: B.rep(ctx, au, 2, out=o)), 3 * elems * esz / 1e9)
rate("Audio:split ch 2", reuse("sp", lambda o: B.split(ctx, au, [2], out=o)), elems * esz / 1e9)
rate("aukit.tone sine", reuse("t", lambda o: B.tone(ctx, n, 440.0, 10.0, 0.5, "sine", 0.5, 2, 48000, out=o)), elems * esz / 1e9)
