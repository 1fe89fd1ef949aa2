"""A/B timings of the DFPWM transcode schedules on one GPU: the config-4 signal (made by the product's own encoder, as bench.py does) and
random bytes (noise: the encoder speculation's worst case).  python tools/r05_dfx_sweep.py [streams ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B

def timeit(ctx, fn, steps=10, warm=2):
    for _ in range(warm): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / steps * 1e3

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 16384]
    dev = torch.device("cuda:0"); ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # (as bench.py does: the inputs are made on torch's stream)
    class A: pass
    for n in sizes:
        a = A(); a.streams = n; a.seconds = 10.0
        wl = bench.DfpwmTranscode().setup(torch, dev, ctx, a, 0, N, B)
        nb = 120000
        rnd = torch.randint(0, 256, (n * nb,), dtype=torch.uint8, device=dev)
        btr = B.Batch.wrap(ctx, rnd.data_ptr(), [i * nb for i in range(n + 1)], keep=rnd)
        out = B.Batch(ctx, __import__("ctypes").c_void_p())
        # the config-4 signal with digital silence in front (2 s) and inside (1 s at 5 s): the encoder of the transcode sits at its strength floor there
        frames, sub = 480000, 512
        gated = torch.empty(n * nb, dtype=torch.uint8, device=dev)
        au, enc = B.AudioBatch(ctx), B.Batch(ctx, __import__("ctypes").c_void_p())
        from aukit_amd import shard
        for s0 in range(0, n, sub):
            k = min(sub, n - s0)
            pcm = bench._sine_noise_s16(torch, dev, k, frames * 2, 48000, 77 + s0).view(k, frames * 2)
            pcm[:, : 2 * 96000] = 0
            pcm[:, 2 * 240000: 2 * 288000] = 0
            pcm = pcm.reshape(-1).contiguous()
            torch.cuda.synchronize()
            bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 4 for i in range(k + 1)], keep=pcm)
            B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"), dtype=N.F32, out=au)
            B.dfpwm_encode(ctx, au, True, out=enc)
            ctx.sync()
            gated[s0 * nb:(s0 + k) * nb].copy_(shard.device_view(enc.device_ptr(), k * nb, dev, keep=enc))
            torch.cuda.synchronize()
        btg = B.Batch.wrap(ctx, gated.data_ptr(), [i * nb for i in range(n + 1)], keep=gated)
        outg = B.Batch(ctx, __import__("ctypes").c_void_p())
        envs = [{}] + [json.loads(e) for e in os.environ.get("SWEEP", "").split(";") if e]
        for env in envs:
            for k, v in env.items(): os.environ[k] = str(v)
            t_sig = timeit(ctx, wl.step)
            t_rnd = timeit(ctx, lambda: B.dfpwm_transcode_mono(ctx, btr, 2, out=out), steps=3, warm=1) if os.environ.get("SWEEP_NOISE", "1") == "1" else float("nan")
            t_gat = timeit(ctx, lambda: B.dfpwm_transcode_mono(ctx, btg, 2, out=outg), steps=3, warm=1)
            for k in env: del os.environ[k]
            print(f"{n:6d} streams  {json.dumps(env):60s} config-4 signal {t_sig:8.3f} ms   with silence {t_gat:8.3f} ms   random bytes {t_rnd:8.3f} ms", flush=True)
main()
