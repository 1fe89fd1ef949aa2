#!/usr/bin/env python3
"""Host-side time of one call (no device sync) next to the synchronised step time, per bench workload: shows which workloads are
bound by the planner on the host rather than by their kernels.  usage (GPU box): python tools/host_time.py [workload ...]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aukit_amd import _native as N, batch as B

names = sys.argv[1:] or ["pcm16_stream", "pcm16_stereo_stream", "g711_stream", "ima_stream", "pcm16_cubic", "g711_cubic", "pcm16_stereo"]
dev = torch.device("cuda", 0)
for w in names:
    args = types.SimpleNamespace(workload=w, streams={"dfpwm_transcode": 16384, "flac_pipeline": 2048, "pcm16_stereo": 2048, "pcm16_stereo_stream": 2048}.get(w, 4096),
                                 seconds=10.0, dtype="f32", cpu_streams=0, interp="cubic", exact_math=1 if w == "pcm16_cubic" else 0, store_x4=1)
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option(N.OPT_EXACT_MATH, args.exact_math)
    wl = bench.WORKLOADS[w]().setup(torch, dev, ctx, args, 0, N, B)
    wl.step(); torch.cuda.synchronize()
    host = []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter(); wl.step(); host.append(time.perf_counter() - t)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): wl.step()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / 10
    wl.step(); torch.cuda.synchronize()
    name, kms, _ = ctx.last_kernel()     # HIP events around the last kernel of the last call
    print("%-22s host %.3f ms/call   step %.3f ms   last kernel %.3f ms (%s)" % (w, sorted(host)[5] * 1e3, tot * 1e3, kms, name[:40]), flush=True)
    del wl, ctx
