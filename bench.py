#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s decoded + resampled to 48 kHz on a 4096-stream batch.

Default workload (BASELINE.json metric): per GPU, 4096 independent 1-channel s16le
44.1 kHz streams of 10 s (441 000 samples = 882 000 B each, 3.61 GB of input resident in HBM) →
`aukit.pcm(d,16,"signed",1,44100):resample(48000,"cubic")` as ONE fused launch of aukit_decode_resample
(f32 store) → 4096 × 480 000 output samples.  One step = one pass over the whole batch.  Streams are independent,
so N GPUs = N shards with no data-path collective (weak scaling: every rank owns a 4096-stream shard; value is the
whole-job rate).  Other BASELINE configs are parity-test cases; `--workload` times them too (not the driver's line).

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


def _sine_noise_s16(torch, dev, n_streams, n_samples, rate, seed):
    """0.5*sine(440 Hz) + uniform noise ±0.25, quantised to s16, generated on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty(n_streams * n_samples, dtype=torch.int16, device=dev)
    t = torch.arange(n_samples, device=dev, dtype=torch.float32) / rate
    sine = 0.5 * torch.sin(2 * torch.pi * 440.0 * t)
    step = 256
    for s0 in range(0, n_streams, step):
        k = min(step, n_streams - s0)
        noise = (torch.rand((k, n_samples), generator=g, device=dev, dtype=torch.float32) - 0.5) * 0.5
        out[s0 * n_samples:(s0 + k) * n_samples] = torch.round((sine[None, :] + noise) * 32767.0).to(torch.int16).reshape(-1)
    return out


def _random_bytes(torch, dev, n, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)


def _measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected
    as /opt/skills/guides/MI355X_MICROARCH.md prescribes), recorded in profiles/traffic.json; None when no matching measurement is committed."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as fh:
            for e in json.load(fh)["entries"]:
                if e["kernel"] == kernel and e["streams"] == args.streams and e["seconds"] == args.seconds:
                    return e["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


class Workload:
    """setup(torch, dev, ctx, args, rank) → self; step() runs one pass; out_samples = units per pass on this rank."""
    name = unit = desc = ""

    def cpu_baseline(self, args):
        return None


class Pcm16Cubic(Workload):
    name, unit = "pcm16_cubic", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.n_samples = int(round(args.seconds * SRC_RATE))
        self.x = _sine_noise_s16(torch, dev, args.streams, self.n_samples, SRC_RATE, 0xA0C17 + 1000 + rank)
        offs = [i * self.n_samples * 2 for i in range(args.streams + 1)]
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_PCM, 1, SRC_RATE, 16, "signed")
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.step = lambda: B.decode_resample(ctx, self.bt, self.d, DST_RATE, args.interp, dtype=self.dtype, out=self.out)
        self.desc = (f"{args.streams}x s16le 44.1kHz mono {args.seconds:g}s per GPU -> aukit.pcm:resample(48000,'cubic'), "
                     f"{args.dtype} store (SURVEY 8d config T)")
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())

    def cpu_baseline(self, args):
        import numpy as np
        from oracle import oracle as O
        O.build()
        rng = np.random.Generator(np.random.PCG64(0xA0C17 + 1000))
        t = np.arange(self.n_samples) / SRC_RATE
        data = np.round((0.5 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.25, 0.25, self.n_samples)) * 32767).astype(np.int16).tobytes()
        done, t0 = 0, time.perf_counter()
        for _ in range(args.cpu_streams):
            done += len(O.resample(O.pcm(data, 16, O.SIGNED, 1, SRC_RATE), DST_RATE, O.CUBIC).data[0])
        dt = time.perf_counter() - t0
        # (ii) every host core, one stream per task; ctypes releases the GIL around the C calls
        import concurrent.futures
        import os
        cores = os.cpu_count() or 1
        one = lambda _: len(O.resample(O.pcm(data, 16, O.SIGNED, 1, SRC_RATE), DST_RATE, O.CUBIC).data[0])
        t1 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(cores) as ex:
            n_all = max(args.cpu_streams, min(args.streams, 16 * cores))
            done_all = sum(ex.map(one, range(n_all)))
        dt_all = time.perf_counter() - t1
        return {"value": done / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"{args.cpu_streams} of the {args.streams} streams ({self.n_samples} samples each), scalar fp64 C oracle, {dt:.1f} s",
                "all_cores": {"value": done_all / dt_all / 1e6, "unit": "Msamples/s", "cores": cores, "streams": n_all, "seconds": round(dt_all, 1)}}


class Pcm16Stereo(Workload):
    """The WAV-file case: 16-bit stereo 44.1 kHz → 48 kHz through the Audio path (not a BASELINE config; same byte mix as config T)."""
    name, unit = "pcm16_stereo", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.n_samples = int(round(args.seconds * SRC_RATE))
        self.x = _sine_noise_s16(torch, dev, args.streams, self.n_samples * 2, SRC_RATE, 0xA0C17 + 6000 + rank)
        offs = [i * self.n_samples * 4 for i in range(args.streams + 1)]
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_PCM, 2, SRC_RATE, 16, "signed")
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.step = lambda: B.decode_resample(ctx, self.bt, self.d, DST_RATE, args.interp, dtype=self.dtype, out=self.out)
        self.desc = f"{args.streams}x s16le 44.1kHz STEREO {args.seconds:g}s -> aukit.pcm:resample(48000,'{args.interp}'), {args.dtype} store; unit = out-samples of both channels"
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum()) * 2


class Pcm16Stream(Workload):
    """config T through the stream path: aukit.stream.pcm(data, 16, "signed", 1, 44100) with defaultInterpolation = cubic, every iterator call."""
    name, unit = "pcm16_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.n_samples = int(round(args.seconds * SRC_RATE))
        self.x = _sine_noise_s16(torch, dev, args.streams, self.n_samples, SRC_RATE, 0xA0C17 + 1000 + rank)
        offs = [i * self.n_samples * 2 for i in range(args.streams + 1)]
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_PCM, 1, SRC_RATE, 16, "signed")
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=self.dtype, out=self.out)
        self.desc = (f"{args.streams}x s16le 44.1kHz mono {args.seconds:g}s per GPU -> aukit.stream.pcm ({args.interp}), all iterator calls, "
                     f"{args.dtype} store (config T, stream path)")
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class Pcm16StereoStream(Workload):
    """The austream case: a 16-bit stereo 44.1 kHz WAV through aukit.stream.pcm (both channels; not a BASELINE config)."""
    name, unit = "pcm16_stereo_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.n_samples = int(round(args.seconds * SRC_RATE))
        self.x = _sine_noise_s16(torch, dev, args.streams, self.n_samples * 2, SRC_RATE, 0xA0C17 + 7000 + rank)
        offs = [i * self.n_samples * 4 for i in range(args.streams + 1)]
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_PCM, 2, SRC_RATE, 16, "signed")
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=self.dtype, out=self.out)
        self.desc = (f"{args.streams}x s16le 44.1kHz STEREO {args.seconds:g}s -> aukit.stream.pcm ({args.interp}), all iterator calls, "
                     f"{args.dtype} store; unit = out-samples of both channels")
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum()) * 2


class G711Cubic(Workload):
    name, unit = "g711_cubic", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        n = int(round(args.seconds * 8000))
        self.x = _random_bytes(torch, dev, args.streams * n, 0xA0C17 + 2000 + rank)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.d = B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.decode_resample(ctx, self.bt, self.d, DST_RATE, "cubic", dtype=N.F32, out=self.out)
        self.desc = f"{args.streams}x G.711 u-law 8kHz {args.seconds:g}s -> aukit.g711:resample(48000,'cubic'), f32 store (config 2a)"
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class G711Stream(Workload):
    """config 2(b): aukit.stream.g711(d, true, 1, 8000, false), every iterator call, int8 out."""
    name, unit = "g711_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        n = int(round(args.seconds * 8000))
        self.x = _random_bytes(torch, dev, args.streams * n, 0xA0C17 + 2000 + rank)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.d = B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=N.I8, out=self.out)
        self.arith = "int decode + f64 resample"
        self.desc = f"{args.streams}x G.711 u-law 8kHz {args.seconds:g}s -> aukit.stream.g711 ({args.interp}), all iterator calls, int8 out (config 2b)"
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class ImaStream(Workload):
    name, unit = "ima_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        import numpy as np
        from oracle import oracle as O  # the oracle's IMA *encoder* only generates the synthetic input (the reference has none)
        blocks = int(round(args.seconds * 22))  # 220 blocks of 512 B ≈ 10.1 s @22 050 Hz
        n = blocks * 512
        # the config-1 style signal through the AUKit-variant encoder; 8 distinct streams, repeated
        kinds = []
        for i in range(8):
            rng = np.random.Generator(np.random.PCG64(0xA0C17 + 3000 + 8 * rank + i))
            t = np.arange(1016 * blocks) / 22050.0
            pcm = np.round((0.5 * np.sin(2 * np.pi * 440 * t) + rng.uniform(-0.25, 0.25, len(t))) * 32767).astype(np.int16)
            enc = O.gen_ima(pcm, 1, 512, 88)
            assert len(enc) == n, (len(enc), n)
            kinds.append(torch.frombuffer(bytearray(enc), dtype=torch.uint8))
        self.x = torch.cat(kinds).to(dev).repeat((args.streams + 7) // 8)[: args.streams * n].contiguous()
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.d = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=N.I8, out=self.out)
        self.arith = "i32 decode + f64 resample"
        self.desc = f"{args.streams}x IMA-ADPCM 22.05kHz mono {blocks}x512B -> stream.adpcm cubic, int8 out (config 3a)"
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class DfpwmTranscode(Workload):
    name, unit = "dfpwm_transcode", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        n = int(round(args.seconds * 12000))  # 120 000 B = 10 s of 2-channel interleaved DFPWM @48 kHz
        self.x = _random_bytes(torch, dev, args.streams * n, 0xA0C17 + 4000 + rank)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.outb = B.Batch(ctx, __import__("ctypes").c_void_p())
        self.step = lambda: B.dfpwm_transcode_mono(ctx, self.bt, 2, out=self.outb)
        self.desc = f"{args.streams}x DFPWM 48kHz stereo {args.seconds:g}s -> aukit.dfpwm:mono():dfpwm() fused (config 4); unit = mono samples"
        return self

    def out_samples(self):
        return int(self.outb.info()[1]) * 8


class FlacPipeline(Workload):
    name, unit = "flac_pipeline", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        import numpy as np
        from oracle import oracle as O  # the oracle's FLAC *encoder* only generates the synthetic input (the reference has none)
        n = int(round(args.seconds * SRC_RATE))
        rng = np.random.Generator(np.random.PCG64(0xA0C17 + 5000 + rank))
        t = np.arange(n) / SRC_RATE
        chans = [np.round((0.5 * np.sin(2 * np.pi * f * t) + rng.uniform(-0.25, 0.25, n)) * 32767 * 0.9) for f in (440.0, 330.0)]
        one = O.gen_flac(np.stack(chans, 1).astype(np.int32).ravel(), 2, 16, SRC_RATE, 4096)
        self.flac_bytes = len(one)
        blob = torch.frombuffer(bytearray(one), dtype=torch.uint8).to(dev)
        self.x = blob.repeat(args.streams)  # identical streams back to back (throughput does not depend on the content)
        offs = [i * len(one) for i in range(args.streams + 1)]
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_FLAC)
        self.a = B.AudioBatch(ctx)
        self.m = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.arith = "i32 decode + " + args.dtype + " resample/effects"

        def step():
            B.decode_resample(ctx, self.bt, self.d, DST_RATE, "cubic", dtype=self.dtype, out=self.a)
            B.effect(ctx, self.a, "highpass", 20.0)
            B.effect(ctx, self.a, "normalize", 0.8)
            B.mono(ctx, self.a, out=self.m)
        self.step = step
        self.desc = (f"{args.streams}x FLAC 44.1kHz stereo 16-bit {args.seconds:g}s ({self.flac_bytes} B each) -> aukit.flac:resample(48000,'cubic') "
                     f"-> highpass(20) -> normalize(0.8) -> mono, {args.dtype} store (config 5); unit = mono out-samples")
        return self

    def out_samples(self):
        return int(self.m.layout()[0].sum())


WORKLOADS = {w.name: w for w in (Pcm16Cubic, Pcm16Stereo, Pcm16Stream, Pcm16StereoStream, G711Cubic, G711Stream, ImaStream, DfpwmTranscode, FlacPipeline)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm", type=float, default=0.5, help="seconds of untimed steps before the warm-up steps (GPU power-state ramp); 0 disables")
    ap.add_argument("--workload", default="pcm16_cubic", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default 4096; 16384 for dfpwm_transcode)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-streams", type=int, default=1024, help="streams timed on the CPU oracle (0 disables)")
    ap.add_argument("--store-x4", type=int, default=1, help="tuning: LDS-transposed 16-byte stores in the v1 fast kernel")
    ap.add_argument("--exact-math", type=int, default=0, help="1: fp64 reference-order kernel even for f32 storage")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--interp", default="cubic", choices=["linear", "cubic"], help="tuning only: the metric is defined on cubic")
    args = ap.parse_args()
    if args.streams is None:
        args.streams = {"dfpwm_transcode": 16384, "flac_pipeline": 2048, "pcm16_stereo": 2048, "pcm16_stereo_stream": 2048}.get(args.workload, 4096)

    import torch
    import torch.distributed as dist
    from aukit_amd import _native as N
    from aukit_amd import batch as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: aukit_amd has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")

    ctx = B.Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # launch on torch's current stream
    ctx.set_option(N.OPT_STORE_X4, args.store_x4)
    ctx.set_option(N.OPT_EXACT_MATH, args.exact_math)
    wl = WORKLOADS[args.workload]().setup(torch, dev, ctx, args, rank, N, B)
    torch.cuda.synchronize()

    # untimed pre-warm (half a second of steps) so that the timed region never starts on a GPU that is still leaving its idle
    # power state — measured: no difference on the boxes of this pool (846-855 G samples/s either way), kept as insurance; then the
    # W warm-up steps
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        wl.step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin()  # HIP events on the stream the kernels are launched on
    for _ in range(args.steps):
        wl.step()
    ev_ms = ctx.timer_end()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    out_samples = wl.out_samples()
    name, _, alg_bytes = ctx.last_kernel()
    if world > 1:
        tt = torch.tensor([dt], device=rdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ts = torch.tensor([float(out_samples)], device=rdev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_samples = float(ts.item())
    else:
        total_samples = float(out_samples)

    if rank == 0:
        kernel_ms = ev_ms / args.steps
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        headline = args.workload == "pcm16_cubic"
        line = {
            "metric": "Msamples/s decoded+resampled to 48kHz, 4096-stream batch" if headline else f"Msamples/s ({args.workload})",
            "value": total_samples * args.steps / dt / 1e6,
            "unit": wl.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # arithmetic type of the path: the f32-store fast kernels use exact integer positions + f32 FMA taps,
            # the reference-order kernels (f64 store or --exact-math 1) compute in fp64; DFPWM / ADPCM decode is int32
            "dtype": getattr(wl, "arith", None) or ("f32" if name.startswith("k_fast") else ("i32" if "dfpwm" in name else "f64")),
            "data": "synthetic",
            "config": {"workload": wl.desc, "streams_per_gpu": args.streams, "seconds_per_stream": args.seconds,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _measured_traffic(name, args), "kernel": name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_out_sample": alg_bytes / max(out_samples, 1)},
        }
        if world == 1 and args.cpu_streams > 0:
            cb = wl.cpu_baseline(args)
            if cb:
                line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
