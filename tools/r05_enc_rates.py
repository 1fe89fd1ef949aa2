"""Audio:dfpwm (aukit_dfpwm_encode) on batches of 10-second mono audios, the chunk-speculative encoder against the older schedules (AUKIT_DFPWM_NOSPEC=1):
python tools/r05_enc_rates.py [streams ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B
dev = torch.device("cuda:0"); ctx = B.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
frames = 480000
for n in [int(a) for a in sys.argv[1:]] or [1, 16, 64, 512, 2048, 4096]:
    pcm = bench._sine_noise_s16(torch, dev, n, frames, 48000, 5000 + n)
    torch.cuda.synchronize()
    bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 2 for i in range(n + 1)], keep=pcm)
    a = B.decode(ctx, bt, B.make_desc(N.CODEC_PCM, 1, 48000, 16, "signed"), dtype=N.F32)
    out = B.Batch(ctx, __import__("ctypes").c_void_p())
    res = {}
    for tag, env in (("spec", {}), ("older", {"AUKIT_DFPWM_NOSPEC": "1"})):
        os.environ.update(env)
        for _ in range(2): B.dfpwm_encode(ctx, a, True, out=out)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(20): B.dfpwm_encode(ctx, a, True, out=out)
        ctx.sync(); res[tag] = (time.perf_counter() - t0) / 20 * 1e3, ctx.last_kernel()[0]
        for k in env: del os.environ[k]
    print(f"{n:5d} streams x 10 s: spec {res['spec'][0]:8.3f} ms ({res['spec'][1]})   older {res['older'][0]:8.3f} ms ({res['older'][1]})   {n * frames / res['spec'][0] / 1e6:8.1f} G samples/s", flush=True)
