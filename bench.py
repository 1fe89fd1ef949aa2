#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s decoded + resampled to 48 kHz on a 4096-stream batch.

Workload (BASELINE.json metric): per GPU, 4096 independent 1-channel s16le
44.1 kHz streams of 10 s (441 000 samples = 882 000 B each, 3.61 GB of input resident in HBM) →
`aukit.pcm(d,16,"signed",1,44100):resample(48000,"cubic")` as ONE fused launch of aukit_decode_resample
(fp64 arithmetic in the reference's operation order, f32 store) → 4096 × 480 000 output samples.
One step = one pass over the whole batch.  Streams are independent, so N GPUs = N shards with no
data-path collective (weak scaling: every rank owns a 4096-stream shard; value is the whole-job rate).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6.29 TB/s is the measured copy ceiling
SRC_RATE, DST_RATE = 44100, 48000


def make_inputs(torch, dev, n_streams, n_samples, rank):
    """Synthetic s16le streams generated on the device: 0.5*sine(440 Hz) + uniform noise ±0.25."""
    g = torch.Generator(device=dev)
    g.manual_seed(0xA0C17 + 1000 * 1 + rank)
    out = torch.empty(n_streams * n_samples, dtype=torch.int16, device=dev)
    t = torch.arange(n_samples, device=dev, dtype=torch.float32) / SRC_RATE
    sine = 0.5 * torch.sin(2 * torch.pi * 440.0 * t)
    step = 256
    for s0 in range(0, n_streams, step):
        k = min(step, n_streams - s0)
        noise = (torch.rand((k, n_samples), generator=g, device=dev, dtype=torch.float32) - 0.5) * 0.5
        out[s0 * n_samples:(s0 + k) * n_samples] = torch.round((sine[None, :] + noise) * 32767.0).to(torch.int16).reshape(-1)
    return out


def cpu_baseline(n_streams, n_samples):
    """The CPU oracle (scalar fp64 C restatement of the reference's Lua loops) timed on a bounded sample of the
    same workload, 1 thread.  A reported baseline, not the target."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    rng = np.random.Generator(np.random.PCG64(0xA0C17 + 1000))
    t = np.arange(n_samples) / SRC_RATE
    sig = 0.5 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.25, 0.25, n_samples)
    data = np.round(sig * 32767).astype(np.int16).tobytes()
    done = 0
    t0 = time.perf_counter()
    for _ in range(n_streams):
        a = O.pcm(data, 16, O.SIGNED, 1, SRC_RATE)
        r = O.resample(a, DST_RATE, O.CUBIC)
        done += len(r.data[0])
    dt = time.perf_counter() - t0
    return {"value": done / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{n_streams} of the 4096 streams ({n_samples} samples each), scalar fp64 C oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-streams", type=int, default=256, help="streams timed on the CPU oracle (0 disables)")
    ap.add_argument("--store-x4", type=int, default=1, help="tuning: LDS-transposed 16-byte stores in the fast kernel")
    ap.add_argument("--exact-math", type=int, default=0, help="1: fp64 reference-order kernel even for f32 storage")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from aukit_amd import _native as N
    from aukit_amd import batch as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: aukit_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n_samples = int(round(args.seconds * SRC_RATE))
    dtype = N.F32 if args.dtype == "f32" else N.F64
    x = make_inputs(torch, dev, args.streams, n_samples, rank)
    torch.cuda.synchronize()

    ctx = B.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # launch on torch's current stream
    ctx.set_option(N.OPT_STORE_X4, args.store_x4)
    ctx.set_option(N.OPT_EXACT_MATH, args.exact_math)
    offs = [i * n_samples * 2 for i in range(args.streams + 1)]
    bt = B.Batch.wrap(ctx, x.data_ptr(), offs, keep=x)
    desc = B.make_desc(N.CODEC_PCM, 1, SRC_RATE, 16, "signed")
    out = B.AudioBatch(ctx)

    def step():
        B.decode_resample(ctx, bt, desc, DST_RATE, "cubic", dtype=dtype, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin()  # HIP events on the stream the kernel is launched on
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_end()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    lens, _, _ = out.layout()
    out_samples = int(lens.sum())
    name, _, alg_bytes = ctx.last_kernel()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ts = torch.tensor([out_samples], device=dev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_samples = float(ts.item())
    else:
        total_samples = float(out_samples)

    if rank == 0:
        kernel_ms = ev_ms / args.steps
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "Msamples/s decoded+resampled to 48kHz, 4096-stream batch",
            "value": total_samples * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # arithmetic type of the path: the f32-store fast kernels use exact integer positions + f32 FMA taps,
            # the reference-order kernels (f64 store or --exact-math 1) compute in fp64
            "dtype": "f32" if name.startswith("k_fast") else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.streams}x s16le 44.1kHz mono {args.seconds:g}s per GPU -> aukit.pcm:resample(48000,'cubic'), "
                                   f"{args.dtype} store (SURVEY 8d config T)",
                       "streams_per_gpu": args.streams, "seconds_per_stream": args.seconds, "interpolation": "cubic",
                       "store": args.dtype, "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_out_sample": alg_bytes / max(out_samples, 1)},
        }
        if world == 1 and args.cpu_streams > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_streams, n_samples)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
