#!/usr/bin/env python3
"""Rates of the floor-in-recurrence codecs (MS-ADPCM, QOA) on ENCODER-MADE input, per 1024 streams of 10 s — the sizes VERDICT r02 quotes.
Loader (aukit.msadpcm / aukit.qoa → Audio, F32), loader + resample, and the stream factories.  GPU box only.
usage: python tools/seq_codec_rates.py [streams=1024] [which=ms,qoa]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = (sys.argv[2] if len(sys.argv) > 2 else "ms,qoa").split(",")
ctx = B.Context(0, dtype=N.F32)
sec = 10


def timed(name, f, outs):
    try:
        f(); ctx.sync(); f(); ctx.sync()
        t0 = time.time()
        for _ in range(5): f()
        ctx.sync(); dt = (time.time() - t0) / 5
        print(f"{name:46s} {dt * 1e3:8.2f} ms  {outs() / dt / 1e9:8.1f} G samples/s  ({ctx.last_kernel()[0]})", flush=True)
    except Exception as e:
        print(f"{name:46s} failed: {str(e)[:100]}", flush=True)


base = [np.stack([pcm16(44100 * sec, 44100, 8, 4 * i + c) for c in range(2)], 1) for i in range(4)]
if "ms" in which:
    for ch in (1, 2):
        ba = 1024
        spb = (ba - 14) + 2 if ch == 2 else (ba - 7) * 2 + 2
        nblk = 44100 * sec // spb
        enc = [O.gen_msadpcm(base[i][: nblk * spb, :ch].ravel(), ch, ba) for i in range(4)]
        bt = B.Batch.upload(ctx, [enc[i % 4] for i in range(n)])
        desc = B.make_desc(N.CODEC_MSADPCM, ch, 44100, block_align=ba)
        keep = [None, None, None, None]
        def loader(): keep[0] = B.decode(ctx, bt, desc, dtype=N.F32, out=keep[0])
        def loader_rs(): keep[1] = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32, out=keep[1])
        cks = [None]
        def stream(): keep[2], cks[0] = B.stream_decode(ctx, bt, desc, "cubic", mono=False, dtype=N.I8, out=keep[2])
        timed(f"aukit.msadpcm {ch}ch -> f32", loader, lambda: n * nblk * spb * ch)
        timed(f"aukit.msadpcm {ch}ch + resample cubic f32", loader_rs, lambda: n * nblk * spb * ch * 48000 / 44100)
        timed(f"stream.msadpcm {ch}ch 44.1k cubic (i8)", stream, lambda: float(np.sum(cks[0].lens)) * ch)
        if ch == 2:
            def stream_m(): keep[3], cks[0] = B.stream_decode(ctx, bt, desc, "cubic", mono=True, dtype=N.I8, out=keep[3])
            timed("stream.msadpcm 2ch mono mix cubic (i8)", stream_m, lambda: float(np.sum(cks[0].lens)))
        del bt, keep
if "qoa" in which:
    for ch in (1, 2):
        qs = [O.gen_qoa(base[i][:, :ch].ravel(), ch, 44100) + b"\0" * 8 for i in range(4)]
        bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(n)])
        desc = B.make_desc(N.CODEC_QOA, ch, 44100)
        keep = [None, None, None, None]
        def loader(): keep[0] = B.decode(ctx, bt, desc, dtype=N.F32, out=keep[0])
        def loader_rs(): keep[1] = B.decode_resample(ctx, bt, desc, 48000, "cubic", dtype=N.F32, out=keep[1])
        cks = [None]
        def stream(): keep[2], cks[0] = B.stream_decode(ctx, bt, desc, "cubic", mono=False, dtype=N.F32, out=keep[2])
        timed(f"aukit.qoa {ch}ch -> f32", loader, lambda: n * 441000 * ch)
        timed(f"aukit.qoa {ch}ch + resample cubic f32", loader_rs, lambda: n * 480000 * ch)
        timed(f"stream.qoa {ch}ch 44.1k cubic (f32)", stream, lambda: float(np.sum(cks[0].lens)) * ch)
        if ch == 2:
            def stream_m(): keep[3], cks[0] = B.stream_decode(ctx, bt, desc, "cubic", mono=True, dtype=N.F32, out=keep[3])
            timed("stream.qoa 2ch mono mix cubic (f32)", stream_m, lambda: float(np.sum(cks[0].lens)))
        del bt, keep
