"""DFPWM on FEW streams (the reference's own use: one): aukit.dfpwm (loader), the transcode, Audio:dfpwm — ms per call on 10-second stereo streams
of four input classes, the chunk-speculative engine against the older schedules:  python tools/r05_small_batches.py [streams ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B, shard
dev = torch.device("cuda:0"); ctx = B.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
nb, frames = 120000, 480000
torch.manual_seed(5)


def make(kind, n):
    if kind == "noise":
        return torch.randint(0, 256, (n * nb,), dtype=torch.uint8, device=dev)
    pcm = bench._sine_noise_s16(torch, dev, n, frames * 2, 48000, 77).view(n, frames * 2)
    if kind == "gated":
        pcm[:, : 2 * 96000] = 0
        pcm[:, 2 * 240000: 2 * 288000] = 0
    if kind == "lead":
        pcm[:, : 2 * 24000] = 0
    pcm = pcm.reshape(-1).contiguous()
    torch.cuda.synchronize()
    bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 4 for i in range(n + 1)], keep=pcm)
    au = B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"), dtype=N.F32)
    enc = B.dfpwm_encode(ctx, au, True)
    ctx.sync()
    return shard.device_view(enc.device_ptr(), n * nb, dev, keep=enc).clone()


def timed(f, reps=10):
    for _ in range(2): f()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for n in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64]:
    for kind in ("signal", "lead", "gated", "noise"):
        x = make(kind, n)
        bt = B.Batch.wrap(ctx, x.data_ptr(), [i * nb for i in range(n + 1)], keep=x)
        d = B.make_desc(N.CODEC_DFPWM, 2, 48000)
        a, out = B.AudioBatch(ctx), B.Batch(ctx, __import__("ctypes").c_void_p())
        mono = B.mono(ctx, B.decode(ctx, bt, d, dtype=N.F32))
        row = [f"{n:4d} x {kind:6s}", f"aukit.dfpwm {timed(lambda: B.decode(ctx, bt, d, dtype=N.F32, out=a)):6.2f}"]
        for tag, env in (("", {}), (" older", {"AUKIT_DFPWM_NOSPEC": "1"})):
            os.environ.update(env)
            row.append(f"transcode{tag} {timed(lambda: B.dfpwm_transcode_mono(ctx, bt, 2, out=out)):6.2f} ({ctx.last_kernel()[0][:14]})")
            row.append(f"Audio:dfpwm{tag} {timed(lambda: B.dfpwm_encode(ctx, mono, True, out=out)):6.2f}")
            for k in env: del os.environ[k]
        print("   ".join(row), flush=True)
