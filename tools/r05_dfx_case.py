"""one input class of the DFPWM transcode, a few steps (for rocprofv3): python tools/r05_dfx_case.py noise|gated|lead|signal 2048 [enc]
(enc: Audio:dfpwm on the mono mix of the same input instead of the transcode)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B, shard
kind, n = sys.argv[1], int(sys.argv[2])
enc_only = len(sys.argv) > 3 and sys.argv[3] == "enc"
dev = torch.device("cuda:0"); ctx = B.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
nb, frames, sub = 120000, 480000, 512
x = torch.empty(n * nb, dtype=torch.uint8, device=dev)
if kind == "noise":
    x = torch.randint(0, 256, (n * nb,), dtype=torch.uint8, device=dev)
else:
    au, enc = B.AudioBatch(ctx), B.Batch(ctx, __import__("ctypes").c_void_p())
    for s0 in range(0, n, sub):
        k = min(sub, n - s0)
        pcm = bench._sine_noise_s16(torch, dev, k, frames * 2, 48000, 77 + s0).view(k, frames * 2)
        if kind == "gated":
            pcm[:, : 2 * 96000] = 0
            pcm[:, 2 * 240000: 2 * 288000] = 0
        if kind == "lead":   # half a second of digital silence in front of the signal, and nothing else
            pcm[:, : 2 * 24000] = 0
        pcm = pcm.reshape(-1).contiguous()
        torch.cuda.synchronize()
        bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 4 for i in range(k + 1)], keep=pcm)
        B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"), dtype=N.F32, out=au)
        B.dfpwm_encode(ctx, au, True, out=enc)
        ctx.sync()
        x[s0 * nb:(s0 + k) * nb].copy_(shard.device_view(enc.device_ptr(), k * nb, dev, keep=enc))
        torch.cuda.synchronize()
bt = B.Batch.wrap(ctx, x.data_ptr(), [i * nb for i in range(n + 1)], keep=x)
out = B.Batch(ctx, __import__("ctypes").c_void_p())
if enc_only:
    mono = B.mono(ctx, B.decode(ctx, bt, B.make_desc(N.CODEC_DFPWM, 2, 48000), dtype=N.F32))
    for _ in range(2): B.dfpwm_encode(ctx, mono, True, out=out)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(10): B.dfpwm_encode(ctx, mono, True, out=out)
    ctx.sync(); print(kind, n, "enc ms/step", (time.perf_counter() - t0) / 10 * 1e3)
    sys.exit(0)
for _ in range(2): B.dfpwm_transcode_mono(ctx, bt, 2, out=out)
ctx.sync(); t0 = time.perf_counter()
for _ in range(5): B.dfpwm_transcode_mono(ctx, bt, 2, out=out)
ctx.sync(); print(kind, n, "ms/step", (time.perf_counter() - t0) / 5 * 1e3)
