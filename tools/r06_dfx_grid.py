"""The chunk-speculative DFPWM engine against the schedules it replaced, over the whole grid VERDICT r05 item 3 names:
streams x input class (signal / lead / gated / noise) x entry point (aukit.dfpwm loader / transcode / Audio:dfpwm) — ms per call with the
default (speculative) engine and with AUKIT_DFPWM_NOSPEC=1, their ratio, and a mark where the default loses by more than 10 %.
    python tools/r06_dfx_grid.py [streams ...]      (default: 1 4 8 16 64 256 1024 2048 4096 8192 16384)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B, shard
dev = torch.device("cuda:0"); ctx = B.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
nb, frames, sub = 120000, 480000, 512
torch.manual_seed(5)


def make(kind, n):
    """n ten-second stereo DFPWM streams of one class, encoder-made on the device in sub-batches"""
    if kind == "noise":
        return torch.randint(0, 256, (n * nb,), dtype=torch.uint8, device=dev)
    x = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    au, enc = B.AudioBatch(ctx), B.Batch(ctx, ctypes.c_void_p())
    for s0 in range(0, n, sub):
        k = min(sub, n - s0)
        pcm = bench._sine_noise_s16(torch, dev, k, frames * 2, 48000, 77 + s0).view(k, frames * 2)
        if kind == "gated":   # two seconds of digital silence in front, one second inside
            pcm[:, : 2 * 96000] = 0
            pcm[:, 2 * 240000: 2 * 288000] = 0
        if kind == "lead":    # half a second in front, nothing else
            pcm[:, : 2 * 24000] = 0
        pcm = pcm.reshape(-1).contiguous()
        torch.cuda.synchronize()
        bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 4 for i in range(k + 1)], keep=pcm)
        B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"), dtype=N.F32, out=au)
        B.dfpwm_encode(ctx, au, True, out=enc)
        ctx.sync()
        x[s0 * nb:(s0 + k) * nb].copy_(shard.device_view(enc.device_ptr(), k * nb, dev, keep=enc))
        torch.cuda.synchronize()
    return x


def timed(f, n):
    reps = 10 if n <= 1024 else (5 if n <= 4096 else 3)
    for _ in range(2): f()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


worst = []
print(f"{'streams x class':>18s}  {'entry point':12s} {'default':>8s} {'NOSPEC':>8s}  ratio", flush=True)
for n in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 64, 256, 1024, 2048, 4096, 8192, 16384]:
    for kind in ("signal", "lead", "gated", "noise"):
        x = make(kind, n)
        bt = B.Batch.wrap(ctx, x.data_ptr(), [i * nb for i in range(n + 1)], keep=x)
        d = B.make_desc(N.CODEC_DFPWM, 2, 48000)
        a, out = B.AudioBatch(ctx), B.Batch(ctx, ctypes.c_void_p())
        ops = [("transcode", lambda: B.dfpwm_transcode_mono(ctx, bt, 2, out=out))]
        if n <= 4096:   # (the loader's and the encoder's rows of 16 384 streams are 63 GB of f32: the grid stops where the paths do)
            mono = B.mono(ctx, B.decode(ctx, bt, d, dtype=N.F32))
            ops += [("aukit.dfpwm", lambda: B.decode(ctx, bt, d, dtype=N.F32, out=a)), ("Audio:dfpwm", lambda: B.dfpwm_encode(ctx, mono, True, out=out))]
        for name, f in ops:
            t_def = timed(f, n)
            os.environ["AUKIT_DFPWM_NOSPEC"] = "1"
            t_old = timed(f, n)
            del os.environ["AUKIT_DFPWM_NOSPEC"]
            r = t_def / t_old
            mark = "  <-- loses" if r > 1.1 else ""
            if r > 1.1: worst.append((r, n, kind, name))
            print(f"{n:6d} x {kind:8s}  {name:12s} {t_def:8.2f} {t_old:8.2f}  {r:5.2f}{mark}", flush=True)
        del x, bt
        torch.cuda.empty_cache()
print("cells where the default loses by more than 10 %:", len(worst), sorted(worst, reverse=True)[:12])
