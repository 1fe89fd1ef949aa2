#!/usr/bin/env python3
"""Soak of the one-launch DFPWM transcode (k_df_fused): ITER batches of 16 384 x 120 000 random bytes (a new seed each time, generated on
the GPU), every result compared on the device, byte for byte, with the time-sliced version's (AUKIT_DFPWM_FUSED=0) — the hand-off between
decoder and encoder waves runs over release / acquire flags, and a visibility bug there would be rare and timing-dependent.
usage (GPU box): python tools/dfpwm_soak.py [ITER=20] [STREAMS=16384]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aukit_amd import batch as B, shard as S

it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda", 0)
ctx = B.Context(0)
offs = (np.arange(n + 1, dtype=np.uint64) * 120000)
bad = 0
for k in range(it):
    g = torch.Generator(device=dev); g.manual_seed(1000 + k)
    raw = torch.randint(0, 256, (n * 120000,), dtype=torch.uint8, device=dev, generator=g)
    if k % 3 == 1:  # long runs of equal bits: strengths at their ceiling
        raw[: raw.numel() // 2] = 0xFF
    torch.cuda.synchronize()  # the library runs on its own HIP stream: the input must be complete before it reads it
    bt = S.wrap_tensor(ctx, raw, offs)
    os.environ.pop("AUKIT_DFPWM_FUSED", None)
    a = B.dfpwm_transcode_mono(ctx, bt, 2)
    ctx.sync()  # ... and its result complete before torch reads it
    name = ctx.last_kernel()[0]
    ta = S.device_view(a.device_ptr(), int(a.offsets()[-1]), dev, keep=a).clone()
    os.environ["AUKIT_DFPWM_FUSED"] = "0"
    b = B.dfpwm_transcode_mono(ctx, bt, 2)
    ctx.sync()
    name_b = ctx.last_kernel()[0]
    tb = S.device_view(b.device_ptr(), int(b.offsets()[-1]), dev, keep=b)
    same = bool(torch.equal(ta, tb))
    bad += 0 if same else 1
    if not same:  # which of the two is wrong (the one-lane-per-stream kernel decides), where, and how much
        os.environ["AUKIT_DFPWM_SERIAL"] = "1"
        c = B.dfpwm_transcode_mono(ctx, bt, 2)
        ctx.sync()
        os.environ.pop("AUKIT_DFPWM_SERIAL")
        tc = S.device_view(c.device_ptr(), int(c.offsets()[-1]), dev, keep=c)
        per = int(a.offsets()[1])
        for nm, t in (("fused", ta), ("sliced", tb)):
            d = (t != tc).view(n, per)
            rows = torch.nonzero(d.any(dim=1)).flatten()
            print(f"   {nm} vs serial ({ctx.last_kernel()[0]}): {int(d.sum())} bytes differ in {rows.numel()} streams", flush=True)
            for r in rows[:6].tolist():
                cols = torch.nonzero(d[r]).flatten()
                print(f"      stream {r} (group {r // 64}, lane {r % 64}): {cols.numel()} bytes, first at {int(cols[0])} (mono sample {int(cols[0]) * 8}), last at {int(cols[-1])}", flush=True)
        del c, tc
    print(f"iter {k}: {name} vs {name_b}: {'equal' if same else 'DIFFERENT'} ({ta.numel()} bytes)", flush=True)
    del a, b, ta, tb, bt, raw
print("soak:", "ok" if bad == 0 else f"{bad} mismatching iterations")
sys.exit(1 if bad else 0)
