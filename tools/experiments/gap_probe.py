import os, sys, time, types
sys.path.insert(0, "/root/repo")
import torch, bench
from aukit_amd import _native as N, batch as B
dev = torch.device("cuda", 0)
for w, em in (("pcm16_stream", 1), ("pcm16_stream", 0), ("pcm16_cubic", 1)):
    args = types.SimpleNamespace(workload=w, streams=4096, seconds=10.0, dtype="f32", cpu_streams=0, interp="cubic", exact_math=em, store_x4=1)
    ctx = B.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_option(N.OPT_EXACT_MATH, em)
    wl = bench.WORKLOADS[w]().setup(torch, dev, ctx, args, 0, N, B)
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    host = []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter(); wl.step(); host.append(time.perf_counter() - t)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): wl.step()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / 20
    # GPU-side: events around 20 steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): wl.step()
    e1.record(); torch.cuda.synchronize()
    print(w, "exact", em, "host %.3f ms/call  step %.3f ms  events %.3f ms/step  last %s" % (sorted(host)[5]*1e3, tot*1e3, e0.elapsed_time(e1)/20, ctx.last_kernel()[0][:50]), flush=True)
    del wl, ctx
