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


def _cpu_pcm16(n_samples, channels, rate, seed_base, count):
    """`count` distinct config-1 style s16 signals (0.5 sine(440 Hz) + uniform noise ±0.25) as byte strings, for the CPU leg"""
    import numpy as np
    t = np.arange(n_samples) / rate
    sine = 0.5 * np.sin(2 * np.pi * 440 * t)
    out = []
    for i in range(count):
        rng = np.random.Generator(np.random.PCG64(0xA0C17 + seed_base + i))
        x = sine[:, None] + rng.uniform(-0.25, 0.25, (n_samples, channels))
        out.append(np.round(x * 32767).astype(np.int16).tobytes())
    return out


def _cpu_random_bytes(n, seed_base, count):
    import numpy as np
    return [np.random.Generator(np.random.PCG64(0xA0C17 + seed_base + i)).integers(0, 256, n, dtype=np.uint8).tobytes() for i in range(count)]


def _measured_traffic_step(workload, args):
    """HBM bytes per STEP of a multi-launch workload: the sum over every kernel of the step of FETCH_SIZE x 2 + WRITE_SIZE from the committed PMC
    passes of this same command (profiles/traffic.json, `steps` entries); None when no matching measurement is committed."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as fh:
            for e in json.load(fh).get("steps", []):
                if e["workload"] == workload and e["streams"] == args.streams and e["seconds"] == args.seconds:
                    return e
    except (OSError, KeyError, ValueError):
        pass
    return None


def _fixture(name):
    """an encoder-made input cached under bench_data/ (tools/make_bench_inputs.py wrote it with the oracle's generators — once, not at bench
    time: the bench itself needs the checker only for the cpu_baseline leg)"""
    path = os.path.join(ROOT, "bench_data", name)
    if not os.path.exists(path):
        raise SystemExit(f"bench.py: {path} is missing: run `python tools/make_bench_inputs.py` (encoder-made inputs are cached fixtures)")
    with open(path, "rb") as fh:
        return fh.read()


def _tile_streams(torch, dev, blobs, streams):
    """`streams` byte strings on the device: the distinct fixture streams, cycled.  Returns (flat uint8 tensor, offsets)."""
    t = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in blobs]
    order = [i % len(t) for i in range(streams)]
    x = torch.cat([t[i] for i in order]).contiguous()
    offs = [0]
    for i in order:
        offs.append(offs[-1] + len(blobs[i]))
    return x, offs


def _gpu_clocks(pci_bus=None):
    """engine / memory clocks of the GPU as the kernel driver reports them right now (sysfs pp_dpm_sclk / pp_dpm_mclk: the level marked `*`),
    MHz — sampled while steps run, before and after the timed windows, so that a line from a box that ran below its clocks says so.  The card is
    the one whose PCI bus number is the device's (a box shows every card of its host); None where nothing can be read."""
    import glob
    out = {}
    base = None
    for f in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk")):
        d = os.path.dirname(f)
        addr = os.path.basename(os.path.realpath(d))   # 0000:5a:00.0
        try:
            bus = int(addr.split(":")[1], 16)
        except (IndexError, ValueError):
            bus = None
        if pci_bus is None or bus == pci_bus:
            base = d
            break
    if base is None:
        return None
    for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        try:
            with open(os.path.join(base, fn)) as fh:
                cur = [l for l in fh.read().splitlines() if l.strip().endswith("*")]
            out[key] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit())) if cur else None
        except (OSError, ValueError, IndexError):
            out[key] = None
    out["card"] = os.path.basename(os.path.dirname(base))
    return out if any(out.get(k) is not None for k in ("sclk_mhz", "mclk_mhz")) else None


def _copy_ceiling(torch, nbytes_traffic, reps=12):
    """what THIS box moves when nothing is computed: a float4 device-to-device copy (tools/bench_copy.hip: 16 bytes per lane and access, non-temporal,
    persistent grid) whose read + write traffic equals `nbytes_traffic`, timed with events on the current stream — the same process, the same
    moment, as the line it calibrates.  Returns (GB/s of traffic: median of `reps`, what copied)."""
    import ctypes
    n = max(int(nbytes_traffic) // 2 // 16 * 4, 1 << 20)   # float32 elements: n * 4 bytes read + n * 4 bytes written
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    lib = None
    path = os.path.join(ROOT, "tools", "libbench_copy.so")
    if os.path.exists(path):
        lib = ctypes.CDLL(path)
        lib.bench_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    cus = torch.cuda.get_device_properties(src.device).multi_processor_count
    shapes = [("plain, one float4 per lane and turn, 4 workgroups per CU", 0, 4), ("non-temporal, four float4 in flight per lane, 8 workgroups per CU", 1, 8),
              ("non-temporal, four float4 in flight per lane, 32 workgroups per CU", 1, 32)] if lib else [("torch copy_ (hipMemcpy device-to-device)", -1, 0)]
    best, how = 0.0, ""
    for name, shape, wgs in shapes:
        def once():
            if lib:
                rc = lib.bench_copy(dst.data_ptr(), src.data_ptr(), n * 4, torch.cuda.current_stream().cuda_stream, cus, shape, wgs)
                if rc:
                    raise RuntimeError(f"bench_copy: HIP error {rc}")
            else:
                dst.copy_(src)
        for _ in range(2):
            once()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            once()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        g = 2 * n * 4 / (ms[len(ms) // 2] * 1e-3) / 1e9
        if g > best:
            best, how = g, ("k_bench_copy (tools/bench_copy.hip), best of %d shapes: %s" % (len(shapes), name)) if lib else name
    ok = bool(torch.equal(dst[:4096], src[:4096]) and torch.equal(dst[-4096:], src[-4096:]))
    del src, dst
    if not ok:
        raise RuntimeError("bench_copy: the copy does not match its source")
    return best, how


class Workload:
    """setup(torch, dev, ctx, args, rank) → self; step() runs one pass; out_samples = units per pass on this rank.
    task_bytes(): END-TO-END algorithmic bytes of one step — the input read once + the final output written once (what the roofline
    fraction of a multi-launch step is measured against; intermediates are the implementation's business)."""
    name = unit = desc = ""
    distinct = None   # how many distinct streams the synthetic batch cycles through (None: every stream is its own)

    def cpu_one(self, args):
        """(one, sample): one(i) runs the oracle's entry points of this workload on the i-th of its distinct CPU inputs and returns the units it
        produced (the metric's unit); sample describes the inputs.  None: no CPU leg (selftest)."""
        return None

    def cpu_baseline(self, args):
        """The C oracle (reference arithmetic, scalar fp64: `kind: port`) timed on this box's host cores on a BOUNDED sample of the workload:
        one thread for about args.cpu_seconds, then every core (one stream per task) for about as long."""
        got = self.cpu_one(args)
        if not got:
            return None
        import concurrent.futures
        one, sample = got
        t0 = time.perf_counter()
        done = one(0)
        t_first = max(time.perf_counter() - t0, 1e-5)
        n1 = int(max(1, min(args.cpu_streams, args.cpu_seconds / t_first)))
        for i in range(1, n1):
            done += one(i)
        dt = time.perf_counter() - t0
        cores = os.cpu_count() or 1
        n_all = int(max(cores, min(args.cpu_streams, cores * (args.cpu_seconds / 2) / (dt / n1))))
        t1 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL around the C calls
            done_all = sum(ex.map(one, range(n_all)))
        dt_all = time.perf_counter() - t1
        return {"value": done / dt / 1e6, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"{n1} streams of the workload's shape ({sample}), scalar fp64 C oracle (reference arithmetic), {dt:.1f} s",
                "all_cores": {"value": done_all / dt_all / 1e6, "unit": self.unit, "cores": cores, "streams": n_all, "seconds": round(dt_all, 1)}}

    def task_bytes(self):
        return None

    # ---- the distribution leg (--distribute 1, SURVEY 8e's second curve): workloads that have one define
    #   dist_whole(torch, ctx, B, args, world) -> the Batch of EVERY rank's shard, resident on rank 0 (the same synthetic content `world` times);
    #   dist_pass(ctx, B, N, args, mine)       -> runs one pass on this rank's scattered shard; returns ("audio" | "batch", the output object);
    #   dist_units(kind, got)                  -> the metric's units in what rank 0 gathered.
    dist_whole = None


def _repeat_batch(torch, ctx, B, x, offs, world):
    """the byte tensor `x` (streams at offsets `offs`) `world` times back to back as one Batch on this rank"""
    x = x.view(torch.uint8)
    big = x.repeat(world) if world > 1 else x
    total = int(offs[-1])
    all_offs = [k * total + int(o) for k in range(world) for o in offs[:-1]] + [world * total]
    return B.Batch.wrap(ctx, big.data_ptr(), all_offs, keep=big)


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
        self.interp = args.interp
        self.step = lambda: B.decode_resample(ctx, self.bt, self.d, DST_RATE, args.interp, dtype=self.dtype, out=self.out)
        self.desc = (f"{args.streams}x s16le 44.1kHz mono {args.seconds:g}s per GPU -> aukit.pcm:resample(48000,'cubic'), "
                     f"{args.dtype} store (SURVEY 8d config T)")
        return self

    def out_samples(self):
        return int(self.out.layout()[0].sum())

    def task_bytes(self):
        return int(self.x.numel()) * 2 + self.out_samples() * (4 if self.dtype == 1 else 8)

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_pcm16(self.n_samples, 1, SRC_RATE, 1000, 64)   # 64 distinct streams = 56 MB of input + a fresh 3.8 MB fp64 row per call: not cache-resident
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: len(O.resample(O.pcm(distinct[i % len(distinct)], 16, O.SIGNED, 1, SRC_RATE), DST_RATE, mode).data[0])
        return one, f"{self.n_samples} samples each, {len(distinct)} distinct ones cycled"


    def dist_whole(self, torch, ctx, B, args, world):
        return _repeat_batch(torch, ctx, B, self.x, [i * self.n_samples * 2 for i in range(args.streams + 1)], world)

    def dist_pass(self, ctx, B, N, args, mine):
        self._dist_out = getattr(self, "_dist_out", None) or B.AudioBatch(ctx)
        B.decode_resample(ctx, mine, self.d, DST_RATE, args.interp, dtype=self.dtype, out=self._dist_out)
        return "audio", self._dist_out


class Pcm16Stereo(Workload):
    """The WAV-file case: 16-bit stereo 44.1 kHz → 48 kHz through the Audio path (not a BASELINE config; same byte mix as config T)."""
    name, unit = "pcm16_stereo", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.interp = args.interp
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

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_pcm16(self.n_samples, 2, SRC_RATE, 6000, 32)
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: 2 * len(O.resample(O.pcm(distinct[i % len(distinct)], 16, O.SIGNED, 2, SRC_RATE), DST_RATE, mode).data[0])
        return one, f"{self.n_samples} stereo frames each, {len(distinct)} distinct ones cycled"

    def out_samples(self):
        return int(self.out.layout()[0].sum()) * 2


class Pcm16Stream(Workload):
    """config T through the stream path: aukit.stream.pcm(data, 16, "signed", 1, 44100) with defaultInterpolation = cubic, every iterator call."""
    name, unit = "pcm16_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.interp = args.interp
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

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_pcm16(self.n_samples, 1, SRC_RATE, 1000, 64)
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: len(O.stream_pcm(distinct[i % len(distinct)], 16, O.SIGNED, 1, SRC_RATE, False, False, mode).data[0])
        return one, f"{self.n_samples} samples each, {len(distinct)} distinct ones cycled; every iterator call"

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class Pcm16StereoStream(Workload):
    """The austream case: a 16-bit stereo 44.1 kHz WAV through aukit.stream.pcm (both channels; not a BASELINE config)."""
    name, unit = "pcm16_stereo_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.interp = args.interp
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

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_pcm16(self.n_samples, 2, SRC_RATE, 7000, 32)
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: 2 * len(O.stream_pcm(distinct[i % len(distinct)], 16, O.SIGNED, 2, SRC_RATE, False, False, mode).data[0])
        return one, f"{self.n_samples} stereo frames each, {len(distinct)} distinct ones cycled; every iterator call"

    def out_samples(self):
        return int(self.out.layout()[0].sum()) * 2


class G711Cubic(Workload):
    name, unit = "g711_cubic", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        n = int(round(args.seconds * 8000))
        self.nbytes = n
        self.x = _random_bytes(torch, dev, args.streams * n, 0xA0C17 + 2000 + rank)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.d = B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.decode_resample(ctx, self.bt, self.d, DST_RATE, "cubic", dtype=N.F32, out=self.out)
        self.desc = f"{args.streams}x G.711 u-law 8kHz {args.seconds:g}s -> aukit.g711:resample(48000,'cubic'), f32 store (config 2a)"
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_random_bytes(self.nbytes, 2000, 64)
        one = lambda i: len(O.resample(O.g711(distinct[i % len(distinct)], True, 1, 8000), DST_RATE, O.CUBIC).data[0])
        return one, f"{self.nbytes} u-law bytes each, {len(distinct)} distinct ones cycled"

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class G711Stream(Workload):
    """config 2(b): aukit.stream.g711(d, true, 1, 8000, false), every iterator call, int8 out."""
    name, unit = "g711_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        n = int(round(args.seconds * 8000))
        self.nbytes = n
        self.interp = args.interp
        self.x = _random_bytes(torch, dev, args.streams * n, 0xA0C17 + 2000 + rank)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * n for i in range(args.streams + 1)], keep=self.x)
        self.d = B.make_desc(N.CODEC_G711, 1, 8000, ulaw=True)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=N.I8, out=self.out)
        self.arith = "int decode + f64 resample"
        self.desc = f"{args.streams}x G.711 u-law 8kHz {args.seconds:g}s -> aukit.stream.g711 ({args.interp}), all iterator calls, int8 out (config 2b)"
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        distinct = _cpu_random_bytes(self.nbytes, 2000, 64)
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        calls = -(-self.nbytes // 8000)   # one iterator call per second of input (with string input the iterator never returns nil: Q13)
        one = lambda i: len(O.stream_g711(distinct[i % len(distinct)], True, 1, 8000, False, mode, max_calls=calls).data[0])
        return one, f"{self.nbytes} u-law bytes each, {len(distinct)} distinct ones cycled; {calls} iterator calls"

    def out_samples(self):
        return int(self.out.layout()[0].sum())


class ImaStream(Workload):
    name, unit = "ima_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        blocks = 220  # 220 blocks of 512 B ≈ 10.1 s @22 050 Hz: the config-1 style signal through the AUKit-variant encoder
        if abs(args.seconds - 10.0) > 1e-9:
            raise SystemExit("ima_stream: the cached fixtures are 10 s long (tools/make_bench_inputs.py)")
        blobs = [_fixture(f"ima_22050_220x512_{i}.bin") for i in range(4)]
        self.blobs, self.interp = blobs, args.interp
        self.distinct = len(blobs)
        self.x, offs = _tile_streams(torch, dev, blobs, args.streams)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=N.I8, out=self.out)
        self.arith = "i32 decode + f32 resample with f64 / reference-order fallback under the floor (bit-exact)"
        self.desc = f"{args.streams}x IMA-ADPCM 22.05kHz mono {blocks}x512B ({self.distinct} distinct encoder-made streams, cycled) -> stream.adpcm cubic, int8 out (config 3a)"
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        blobs = self.blobs
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: len(O.stream_adpcm(blobs[i % len(blobs)], 512, 1, 22050, False, mode).data[0])
        return one, f"220 blocks of 512 B each, the {len(blobs)} fixture streams cycled; every iterator call"

    def out_samples(self):
        return int(self.out.layout()[0].sum())

    def task_bytes(self):
        return int(self.x.numel()) + self.out_samples()


class ImaPipeline(Workload):
    """BASELINE config 3 as written: IMA-ADPCM in WAV blocks -> aukit.wav(d):resample(48000, "cubic") -> aukit.effects.lowpass(a, 11025)
    (auplay.lua:11-34 for the call order).  Algorithmic bytes: 0.2316 B in + 4 B out per output sample."""
    name, unit = "ima_pipeline", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        if abs(args.seconds - 10.0) > 1e-9:
            raise SystemExit("ima_pipeline: the cached fixtures are 10 s long (tools/make_bench_inputs.py)")
        blobs = [_fixture(f"ima_22050_220x512_{i}.bin") for i in range(4)]
        self.blobs = blobs
        self.distinct = len(blobs)
        self.x, offs = _tile_streams(torch, dev, blobs, args.streams)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512)
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64

        def step():
            B.decode_resample(ctx, self.bt, self.d, DST_RATE, "cubic", dtype=self.dtype, out=self.out)
            B.effect(ctx, self.out, "lowpass", 11025.0)
            self.out.device_ptr()   # (nothing may stay owed on the rows when the step ends: a no-op for one channel, the filter is paid by the call above)
        self.step = step
        # (the recurrence's arithmetic is what the library reports of the launch it made — AUKIT_COUNTER_RECURRENCE_F32 — not a literal: round 4's last
        # commit moved it to f32 for slopes <= 1/2 and these labels went on saying f64)
        self.arith = (lambda ctx, N: "i32 decode + f32 interpolation, " + ("f32" if ctx.counter(N.COUNTER_RECURRENCE_F32) else "f64") + " recurrence and scan") if args.dtype == "f32" \
            else "i32 decode + f64 reference-order resample and filter"
        self.desc = (f"{args.streams}x IMA-ADPCM 22.05kHz mono 220x512B in WAV blocks ({self.distinct} distinct encoder-made streams, cycled) -> aukit.wav:resample(48000,'cubic') "
                     f"-> effects.lowpass(11025), {args.dtype} store (config 3b)")
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        blobs = self.blobs
        one = lambda i: len(O.fx_lowpass(O.resample(O.wav_adpcm(blobs[i % len(blobs)], 512, 1, 22050), DST_RATE, O.CUBIC), 11025.0).data[0])
        return one, f"220 blocks of 512 B each, the {len(blobs)} fixture streams cycled: aukit.wav -> resample -> lowpass"

    def out_samples(self):
        return int(self.out.layout()[0].sum())

    def task_bytes(self):
        return int(self.x.numel()) + self.out_samples() * (4 if self.dtype == 1 else 8)


class MsadpcmStream(Workload):
    """aukit.stream.msadpcm on encoder-made mono blocks of 1024 B @44.1 kHz (k_ms_wave): not a BASELINE config, a north-star codec."""
    name, unit = "msadpcm_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        if abs(args.seconds - 10.0) > 1e-9:
            raise SystemExit("msadpcm_stream: the cached fixtures are 10 s long (tools/make_bench_inputs.py)")
        blobs = [_fixture(f"msadpcm_44100_mono_216x1024_{i}.bin") for i in range(2)]
        self.blobs, self.interp = blobs, args.interp
        self.distinct = len(blobs)
        self.x, offs = _tile_streams(torch, dev, blobs, args.streams)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_MSADPCM, 1, 44100, block_align=1024)
        self.out = B.AudioBatch(ctx)
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=N.I8, out=self.out)
        self.arith = "i32 recurrence (fp64 per lane beyond the int32-safe range) + f32 resample with f64 / reference-order fallback under the floor (bit-exact)"
        self.desc = f"{args.streams}x MS-ADPCM 44.1kHz mono 216x1024B ({self.distinct} distinct encoder-made streams, cycled) -> stream.msadpcm {args.interp}, int8 out"
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        blobs = self.blobs
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: len(O.stream_msadpcm(blobs[i % len(blobs)], 1024, 1, 44100, False, None, mode).data[0])
        return one, f"216 blocks of 1024 B each, the {len(blobs)} fixture streams cycled; every iterator call"

    def out_samples(self):
        return int(self.out.layout()[0].sum())

    def task_bytes(self):
        return int(self.x.numel()) + self.out_samples()


class QoaStream(Workload):
    """aukit.stream.qoa on encoder-made stereo files @44.1 kHz: frame walk + k_qoa_wave (int8 rows) + k_iir_tail, f32 out."""
    name, unit = "qoa_stream", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        if abs(args.seconds - 10.0) > 1e-9:
            raise SystemExit("qoa_stream: the cached fixtures are 10 s long (tools/make_bench_inputs.py)")
        blobs = [_fixture(f"qoa_44100_stereo_10s_{i}.bin") for i in range(2)]
        self.blobs, self.interp = blobs, args.interp
        self.distinct = len(blobs)
        self.x, offs = _tile_streams(torch, dev, blobs, args.streams)
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), offs, keep=self.x)
        self.d = B.make_desc(N.CODEC_QOA, 2, 44100)
        self.out = B.AudioBatch(ctx)
        self.dtype = N.F32 if args.dtype == "f32" else N.F64
        self.step = lambda: B.stream_decode(ctx, self.bt, self.d, args.interp, dtype=self.dtype, out=self.out)
        self.arith = (lambda ctx, N: "i32 LMS decode + f32 interpolation, " + ("f32" if ctx.counter(N.COUNTER_RECURRENCE_F32) else "f64") + " recurrence and scan") if args.dtype == "f32" \
            else "i32 LMS decode + f64 reference-order tail"
        self.desc = (f"{args.streams}x QOA 44.1kHz stereo 10s ({self.distinct} distinct encoder-made files, cycled) -> stream.qoa {args.interp}, all iterator calls, "
                     f"{args.dtype} store; unit = out-samples of both channels")
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        blobs = self.blobs
        mode = O.CUBIC if self.interp == "cubic" else O.LINEAR
        one = lambda i: 2 * len(O.stream_qoa(blobs[i % len(blobs)], False, mode).data[0])
        return one, f"10 s stereo files, the {len(blobs)} fixture files cycled; every iterator call"

    def out_samples(self):
        return int(self.out.layout()[0].sum()) * 2

    def task_bytes(self):
        return int(self.x.numel()) + self.out_samples() * (4 if self.dtype == 1 else 8)


class DfpwmTranscode(Workload):
    name, unit = "dfpwm_transcode", "Msamples/s"
    @property
    def valu_per_unit(self):
        """the hot block's VALU instructions per mono sample, from the code this library was built from (tools/isa_count.py writes aukit_amd/isa_counts.json at
        build time: ADVICE r05 — the literal 731 of round 5 belonged to one build); None (no `issue` object in the line) where that file is missing"""
        try:
            with open(os.path.join(ROOT, "aukit_amd", "isa_counts.json")) as fh:
                c = json.load(fh)["dfx_chunks0_hot_valu"]
            return (c["value"] / 16, f"k_dfx_chunks<0>'s hot block: {c['value']} VALU instructions per source dword = 16 mono samples (2 x 15 decoder steps, mix index, 12 encoder "
                                     f"steps, bookkeeping); {c['definition']}, sources {c['src_sha16']}")
        except (OSError, KeyError, ValueError, TypeError):
            return None

    def setup(self, torch, dev, ctx, args, rank, N, B):
        # "produced by the build's DFPWM encoder" — the config-1 style signal, two channels, 48 kHz, through the product's own
        # aukit.pcm → Audio:dfpwm on the device (sub-batches of 512 streams; every stream has its own noise).  The chunk-parallel decoder's
        # warm-up / verify / redo step depends on what the bytes are, so the timed input is what the config names, not random bytes.
        frames = int(round(args.seconds * 48000))
        self.frames = frames
        nb = frames * 2 // 8
        self.x = torch.empty(args.streams * nb, dtype=torch.uint8, device=dev)
        pcm_desc = B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed")
        sub = 512
        a, enc = B.AudioBatch(ctx), B.Batch(ctx, __import__("ctypes").c_void_p())
        from aukit_amd import shard
        for s0 in range(0, args.streams, sub):
            k = min(sub, args.streams - s0)
            pcm = _sine_noise_s16(torch, dev, k, frames * 2, 48000, 0xA0C17 + 4000 + 97 * rank + s0)   # interleaved frames: the sine is common, the noise per sample
            bt = B.Batch.wrap(ctx, pcm.data_ptr(), [i * frames * 4 for i in range(k + 1)], keep=pcm)
            B.decode(ctx, bt, pcm_desc, dtype=N.F32, out=a)
            B.dfpwm_encode(ctx, a, True, out=enc)
            ctx.sync()
            n_e, tot_e = enc.info()
            assert n_e == k and tot_e == k * nb, (n_e, tot_e, k, nb)
            self.x[s0 * nb:(s0 + k) * nb].copy_(shard.device_view(enc.device_ptr(), tot_e, dev, keep=enc))
            torch.cuda.synchronize()
            del bt, pcm
        a.free(); enc.free()
        self.bt = B.Batch.wrap(ctx, self.x.data_ptr(), [i * nb for i in range(args.streams + 1)], keep=self.x)
        self.outb = B.Batch(ctx, __import__("ctypes").c_void_p())
        self.step = lambda: B.dfpwm_transcode_mono(ctx, self.bt, 2, out=self.outb)
        self.desc = (f"{args.streams}x DFPWM 48kHz stereo {args.seconds:g}s, made by the build's own encoder (aukit.pcm -> Audio:dfpwm on the device) from the config-1 "
                     f"style signal -> aukit.dfpwm:mono():dfpwm() fused (config 4); unit = mono samples")
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        frames = self.frames
        # the same route as the GPU input: the config-1 style stereo signal through the (oracle's) DFPWM encoder, 4 distinct streams
        enc = [O.audio_dfpwm(O.pcm(d, 16, O.SIGNED, 2, 48000), True) for d in _cpu_pcm16(frames, 2, 48000, 4000, 4)]
        one = lambda i: 8 * len(O.audio_dfpwm(O.mono(O.dfpwm(enc[i % len(enc)], 2, 48000)), True))
        return one, f"{len(enc[0])} DFPWM bytes each (stereo, {frames} frames), {len(enc)} distinct encoder-made streams cycled: aukit.dfpwm -> mono -> dfpwm"

    def out_samples(self):
        return int(self.outb.info()[1]) * 8

    def task_bytes(self):
        return int(self.x.numel()) + int(self.outb.info()[1])

    def extra(self, ctx, N):
        """the chunk-parallel decoder's redo count on this input (one more, untimed, step with the counters on)"""
        ctx.set_option(N.OPT_COLLECT_STATS, 1)
        self.step()
        ctx.sync()
        ctx.set_option(N.OPT_COLLECT_STATS, 0)
        return {"dfpwm_chunks": ctx.counter(N.COUNTER_DFPWM_CHUNKS), "dfpwm_chunks_redone": ctx.counter(N.COUNTER_DFPWM_CHUNKS_REDONE)}


    def dist_whole(self, torch, ctx, B, args, world):
        nb = self.frames * 2 // 8
        return _repeat_batch(torch, ctx, B, self.x, [i * nb for i in range(args.streams + 1)], world)

    def dist_pass(self, ctx, B, N, args, mine):
        self._dist_out = getattr(self, "_dist_out", None) or B.Batch(ctx, __import__("ctypes").c_void_p())
        B.dfpwm_transcode_mono(ctx, mine, 2, out=self._dist_out)
        return "batch", self._dist_out   # the re-encoded DFPWM bytes travel back (shard.gather_batch)


class FlacPipeline(Workload):
    name, unit = "flac_pipeline", "Msamples/s"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        if abs(args.seconds - 10.0) > 1e-9:
            raise SystemExit("flac_pipeline: the cached fixture is 10 s long (tools/make_bench_inputs.py)")
        # sixteen distinct encoder-made streams, cycled (round 6; the earlier rounds ran copies of ONE file): own seeds, own tones, own mix of
        # subframe types / predictor orders / partition orders / stereo modes per frame (tools/make_bench_inputs.py)
        blobs = [_fixture(f"flac_44100_stereo_10s_{i}.bin") for i in range(16)]
        self.blobs = blobs
        self.distinct = len(blobs)
        self.flac_bytes = sum(len(b) for b in blobs) // len(blobs)
        self.x, offs = _tile_streams(torch, dev, blobs, args.streams)
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
            self.m.device_ptr()   # the step ends with the mono rows FINAL in HBM: whatever the calls above left owed on them (a deferred normalize) is paid here
        self.step = step
        self.desc = (f"{args.streams}x FLAC 44.1kHz stereo 16-bit {args.seconds:g}s (about {self.flac_bytes} B each, {self.distinct} distinct encoder-made streams cycled) -> aukit.flac:resample(48000,'cubic') "
                     f"-> highpass(20) -> normalize(0.8) -> mono, {args.dtype} store (config 5); unit = mono out-samples")
        return self

    def cpu_one(self, args):
        from oracle import oracle as O
        O.build()
        blobs = self.blobs

        def one(i):
            a = O.resample(O.flac(blobs[i % len(blobs)]), DST_RATE, O.CUBIC)
            a = O.fx_normalize(O.fx_highpass(a, 20.0), 0.8)
            return len(O.mono(a).data[0])
        return one, f"the {len(blobs)} fixture streams cycled (about {self.flac_bytes} B each, 10 s stereo): aukit.flac -> resample -> highpass -> normalize -> mono"

    def out_samples(self):
        return int(self.m.layout()[0].sum())

    def task_bytes(self):
        return int(self.x.numel()) + self.out_samples() * (4 if self.dtype == 1 else 8)


    def dist_whole(self, torch, ctx, B, args, world):
        return _repeat_batch(torch, ctx, B, self.x, [i * self.flac_bytes for i in range(args.streams + 1)], world)

    def dist_pass(self, ctx, B, N, args, mine):
        if getattr(self, "_dist_a", None) is None:
            self._dist_a, self._dist_m = B.AudioBatch(ctx), B.AudioBatch(ctx)
        B.decode_resample(ctx, mine, self.d, DST_RATE, "cubic", dtype=self.dtype, out=self._dist_a)
        B.effect(ctx, self._dist_a, "highpass", 20.0)
        B.effect(ctx, self._dist_a, "normalize", 0.8)
        B.mono(ctx, self._dist_a, out=self._dist_m)
        return "audio", self._dist_m   # the mono rows travel back (shard.gather_audio pays whatever is still owed on them)


class SelftestNull(Workload):
    """No GPU, no kernel: a fixed sleep per step.  Exists so that the N-rank control flow of this file (self-launch of --gpus N, rendezvous,
    barriers, max-over-ranks time, summed units) runs on CPU under gloo (tests/test_bench_launch.py).  Never a measurement."""
    name, unit, arith = "selftest_null", "units/s", "none"

    def setup(self, torch, dev, ctx, args, rank, N, B):
        self.step = lambda: time.sleep(0.002 * (1 + rank))
        self.desc = "selftest: sleep(2 ms x (rank + 1)) per step, 1000 units per step and rank"
        return self

    def out_samples(self):
        return 1000


WORKLOADS = {w.name: w for w in (SelftestNull, Pcm16Cubic, Pcm16Stereo, Pcm16Stream, Pcm16StereoStream, G711Cubic, G711Stream, ImaStream, ImaPipeline, MsadpcmStream, QoaStream, DfpwmTranscode, FlacPipeline)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start N ranks of this same file through
    torch.distributed.run (one process per GPU) and hand their exit code back.  Runs BEFORE this process has made any GPU call
    (children are fresh processes: nothing that has initialised the GPU is ever re-exec'ed)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def reduce_over_ranks(dist, torch, rdev, dt, units):
    """(max over ranks of the timed region, sum over ranks of the units processed in it): the whole-job rate is units ÷ time."""
    tt = torch.tensor([dt], device=rdev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ts = torch.tensor([float(units)], device=rdev, dtype=torch.float64)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return float(tt.item()), float(ts.item())


XGMI_LINK_GBS = 153.0  # one xGMI link, per direction (the brief's figure): every peer's share crosses exactly one


def time_distribution(torch, dist, dev, ctx, wl, args, world, rank, N, B, sync):
    """SURVEY 8e's second curve: the job's input starts on rank 0 (all `world` shards back to back in its HBM), is scattered device to device
    (aukit_amd.shard.scatter_batch: RCCL point-to-point, each peer's share over its own xGMI link, received straight into the tensor
    the library then wraps with aukit_batch_wrap_device), every rank runs ONE pass of the workload's chain (Workload.dist_pass), and the outputs
    are gathered to rank 0 — rows (shard.gather_audio: sent straight from aukit_audio_device_ptr; config T, config 5's mono rows) or
    re-encoded bytes (shard.gather_batch; config 4).  Returns the three times (max over ranks) and the inclusive rate."""
    from aukit_amd import shard
    own_group = False
    if not dist.is_initialized():  # N = 1: a one-rank group, so that the same code runs (its scatter / gather are views, nothing moves)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
        own_group = True
    try:
        whole = wl.dist_whole(torch, ctx, B, args, world) if rank == 0 else None
        times = []
        def agreed(err):
            # every rank learns whether ANY rank failed its local stage before the next collective is entered: a rank that raised on its own
            # (out of memory in the pass, say) would otherwise leave the others waiting inside the gather until the launcher's timeout
            flag = torch.tensor([0 if err is None else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                raise RuntimeError(f"distribution pass failed on rank {rank}: {err}" if err is not None else "distribution pass failed on another rank")
        share_in = nbytes_out = nunits = 0
        for it in range(2):  # the first round pays RCCL's connection set-up: report the second
            sync(); dist.barrier(); sync()
            t0 = time.perf_counter()
            mine, (lo, hi) = shard.scatter_batch(ctx, whole, src=0, device=dev)
            sync(); t1 = time.perf_counter()
            err, kind, out = None, "audio", None
            try:
                kind, out = wl.dist_pass(ctx, B, N, args, mine)
                ctx.sync()
            except Exception as e:  # noqa: BLE001 — reported through agreed()
                err = f"{type(e).__name__}: {e}"
            sync(); t2 = time.perf_counter()
            agreed(err)   # (outside the timed stages' meaning: a 4-byte all-reduce, counted in the gather time)
            got = shard.gather_audio(out, dst=0, device=dev) if kind == "audio" else shard.gather_batch(out, dst=0, device=dev)
            sync(); dist.barrier(); sync()
            t3 = time.perf_counter()
            times.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
            share_in = int(mine.info()[1])
            if got is not None:
                nbytes_out = sum(int(t.numel()) for t, _ in got)
                nunits = sum(int(m["lens"].sum()) * int(m.get("channels", 1)) for _, m in got) if kind == "audio" else 8 * sum(int(sum(sz)) for _, sz in got)
            del got, mine
        tt = torch.tensor(times[-1], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sc, co, ga, tot = [float(v) for v in tt.tolist()]
        if rank != 0:
            return None
        share_out = nbytes_out / max(world, 1)
        return {"scatter_s": sc, "compute_s": co, "gather_s": ga, "total_s": tot, "value_inclusive": nunits / tot / 1e6, "unit": "Msamples/s",
                "bytes_scattered_per_peer": share_in, "bytes_gathered_per_peer": int(share_out), "peers": world - 1,
                "xgmi_link_GBs": XGMI_LINK_GBS, "link_bound_s": ((share_in + share_out) / (XGMI_LINK_GBS * 1e9)) if world > 1 else 0.0,
                "note": "input on rank 0 -> RCCL scatter (HBM to HBM) -> one pass per rank -> gather of the outputs (rows, or re-encoded bytes) to rank 0; max over ranks; "
                        "xGMI-bound by construction (SURVEY 8e), reported beside `value`, never as it"}
    finally:
        if own_group:
            dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm", type=float, default=0.5, help="seconds of untimed steps before the warm-up steps (GPU power-state ramp); 0 disables")
    ap.add_argument("--workload", default="pcm16_cubic", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default 4096; 16384 for dfpwm_transcode)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="STORAGE type of the output rows")
    ap.add_argument("--cpu-streams", type=int, default=1024, help="most streams timed on the CPU oracle (0 disables the cpu_baseline leg)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="about how long the one-thread CPU leg runs (the all-cores leg runs about half as long again)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --streams per GPU whatever N; strong: --streams is the TOTAL, cut into N contiguous shares (aukit_partition's rule)")
    ap.add_argument("--min-seconds", type=float, default=0.0,
                    help="raise --steps so that the timed region lasts at least this long (keeps the device busy long enough for an external sampler); 0: exactly --steps")
    ap.add_argument("--store-x4", type=int, default=1, help="tuning: LDS-transposed 16-byte stores in the v1 fast kernel")
    ap.add_argument("--exact-math", type=int, default=None,
                    help="arithmetic behind f32 storage: 1 = fp64 (the reference computes in doubles; default for pcm16_cubic, the driver's line), "
                         "2 = fp64 in the reference's operation order, 0 = f32 taps (default for the other workloads)")
    ap.add_argument("--extra-windows", type=int, default=4, help="further K-step windows timed after the contractual one (spread of the measurement)")
    ap.add_argument("--fast-line", type=int, default=1, help="pcm16_cubic: also time the f32-tap kernel and report it as roofline_fast")
    ap.add_argument("--copy-line", type=int, default=1, help="pcm16_cubic: also time a device copy of the same traffic on this box (roofline.copy_ceiling_GBs, frac_of_copy)")
    ap.add_argument("--distribute", type=int, default=0,
                    help="1: after the measurement, ALSO time the distribution variant (rank 0 holds every rank's input in its HBM: device-to-device "
                         "scatter over RCCL, one pass, gather of the output rows to rank 0) and report it as `distribute` — a second number, never `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--interp", default="cubic", choices=["linear", "cubic"], help="tuning only: the metric is defined on cubic")
    args = ap.parse_args(argv)
    if args.streams is None:
        args.streams = {"dfpwm_transcode": 16384, "flac_pipeline": 2048, "pcm16_stereo": 2048, "pcm16_stereo_stream": 2048}.get(args.workload, 4096)
    if args.exact_math is None:
        args.exact_math = 1 if args.workload in ("pcm16_cubic", "g711_cubic", "pcm16_stream") else 0   # the workloads that have an fp64-arithmetic wave kernel
    selftest = args.workload == "selftest_null"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let `python bench.py --gpus N` start them)")
    streams_total = args.streams * world
    if args.scaling == "strong":   # a fixed job cut into contiguous shares: rank g takes streams [total g / N, total (g + 1) / N) (equal-sized synthetic streams: the byte-balanced cut)
        streams_total = args.streams
        args.streams = streams_total * (rank + 1) // world - streams_total * rank // world
        if args.streams < 1:
            raise SystemExit(f"bench.py: --scaling strong with {streams_total} streams leaves rank {rank} of {world} without any")
    if selftest:
        N = B = ctx = None
        dev = torch.device("cpu")
        args.backend = "gloo" if args.backend == "nccl" else args.backend
        sync = lambda: None
    else:
        from aukit_amd import _native as N
        from aukit_amd import batch as B
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: aukit_amd has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {local_rank} has no GPU ({torch.cuda.device_count()} visible): --gpus must not exceed the GPUs of the node")
        dev_index = local_rank
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        sync = torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))   # a rank lost inside a collective ends the job in minutes, not at the launcher's limit
        else:
            dist.init_process_group(args.backend)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")

    if not selftest:
        ctx = B.Context(dev_index)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # launch on torch's current stream
        ctx.set_option(N.OPT_STORE_X4, args.store_x4)
        ctx.set_option(N.OPT_EXACT_MATH, args.exact_math)
    wl = WORKLOADS[args.workload]().setup(torch, dev, ctx, args, rank, N, B)
    sync()

    def timed_window():
        """K steps bracketed by barrier + synchronize on both sides; returns (wall seconds, HIP-event ms, launches, algorithmic bytes)"""
        # (the host's garbage collector out of the timed region: a full collection — 30 ms with the chunk tables of the stream workloads alive — once in
        # fifty-odd steps landed in one window of five and read as a step of 5.8 ms among steps of 2.4: profiles/r05_bench_lines.jsonl, r06)
        import gc
        gc.collect()
        if not os.environ.get("AUKIT_BENCH_KEEP_GC"):
            gc.disable()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        if ctx:
            ctx.timer_begin()  # HIP events on the stream the kernels are launched on
        for _ in range(args.steps):
            wl.step()
        ev_ms = ctx.timer_end() if ctx else 0.0
        sync()
        if world > 1:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        nl, nb = ctx.timer_stats() if ctx else (0, 0)
        return dt, ev_ms, nl, nb

    # untimed pre-warm (half a second of steps) so that the timed region never starts on a GPU that is still leaving its idle
    # power state; then the W warm-up steps
    # (steps queued eight at a time: with a sync behind every step the GPU idles between them and a short, instruction-bound step — stream.g711,
    # 1.5 ms — started its timed region 10-25 % below the clocks it reaches under back-to-back launches; the later windows showed the difference)
    # ... and a batch of a small workload (1024 streams: a step of 2 ms that does not fill the chip) needed more than eight steps in the queue: the
    # batches are sized to ~100 ms of queued work from the time of the first one.
    t_pre = time.perf_counter()
    per_sync = 8
    while not selftest and time.perf_counter() - t_pre < args.prewarm:
        t_b = time.perf_counter()
        for _ in range(per_sync):
            wl.step()
        sync()
        dt_b = max(time.perf_counter() - t_b, 1e-6)
        per_sync = int(min(256, max(8, per_sync * 0.1 / dt_b)))
    for _ in range(args.warmup):
        wl.step()
    if args.min_seconds > 0:   # as many steps as fill the asked-for time (the same count on every rank: rank 0's estimate)
        sync()
        t_e = time.perf_counter()
        wl.step()
        sync()
        want = int(args.min_seconds / max(time.perf_counter() - t_e, 1e-6)) + 1
        if world > 1:
            tw = torch.tensor([want], device=rdev, dtype=torch.int64)
            dist.broadcast(tw, 0)
            want = int(tw.item())
        args.steps = max(args.steps, want)
    def clocks_under_load():
        """the clocks while steps are running (an idle GPU drops to its lowest level at once: sampled behind a sync they say nothing)"""
        if selftest:
            return None
        for _ in range(max(4, min(64, per_sync))):
            wl.step()
        c = _gpu_clocks(getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None))
        sync()
        return c
    clocks_before = clocks_under_load()
    dt, ev_ms, n_launch, alg_total = timed_window()  # THE measurement: exactly K steps
    out_samples = wl.out_samples()
    name = ctx.last_kernel()[0] if ctx else "none"
    if world > 1:
        dt, total_samples = reduce_over_ranks(dist, torch, rdev, dt, out_samples)
    else:
        total_samples = float(out_samples)
    # spread of the measurement: further windows of the same K steps, each reduced like the first (max over ranks)
    windows = [dt / args.steps * 1e3]
    kernel_windows = [ev_ms / args.steps]
    for _ in range(max(args.extra_windows, 0)):
        d2, e2, _, _ = timed_window()
        if world > 1:
            d2, _ = reduce_over_ranks(dist, torch, rdev, d2, out_samples)
        windows.append(d2 / args.steps * 1e3)
        kernel_windows.append(e2 / args.steps)

    clocks_after = clocks_under_load()
    # every rank says who it is (device, what it sees of the job, its RCCL): the first run on N GPUs verifies itself from its own line
    ranks_info = None
    if not selftest:
        try:
            props = torch.cuda.get_device_properties(dev)
            me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
                  "uuid": str(getattr(props, "uuid", "")) or None, "cus": props.multi_processor_count,
                  "world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": args.backend if world > 1 else None,
                  "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                  "streams": args.streams, "out_samples": int(out_samples)}
        except Exception as e:  # never lose the line to the census
            me = {"rank": rank, "error": f"{type(e).__name__}: {e}"}
        if world > 1:
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, me)
        else:
            ranks_info = [me]
    copy_gbs, copy_how = None, None
    if not selftest and world == 1 and args.copy_line and args.workload == "pcm16_cubic":
        try:
            copy_gbs, copy_how = _copy_ceiling(torch, wl.task_bytes() if hasattr(wl, "task_bytes") and wl.task_bytes() else 11476992000)
        except Exception as e:
            copy_gbs, copy_how = None, f"{type(e).__name__}: {e}"

    fast = None
    if not selftest and args.workload == "pcm16_cubic" and args.exact_math != 0 and args.fast_line and args.dtype == "f32":
        ctx.set_option(N.OPT_EXACT_MATH, 0)  # the f32-tap kernel on the same batch: a secondary figure, never `value`
        for _ in range(args.warmup):
            wl.step()
        _, fe, _, fb = timed_window()
        fast = (ctx.last_kernel()[0], fe / args.steps, fb // args.steps)
        ctx.set_option(N.OPT_EXACT_MATH, args.exact_math)

    distribute = None
    if args.distribute and not selftest and wl.dist_whole is not None:
        try:
            distribute = time_distribution(torch, dist, dev, ctx, wl, args, world, rank, N, B, sync)
        except Exception as e:  # the distribution figure is optional: never lose the line to it
            distribute = {"error": f"{type(e).__name__}: {e}"}

    line = None
    if rank == 0:
        import statistics
        kernel_ms = ev_ms / args.steps
        alg_bytes = alg_total // max(args.steps, 1)      # algorithmic bytes of every launch of one step
        launches = n_launch // max(args.steps, 1)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        headline = args.workload == "pcm16_cubic"
        if getattr(wl, "arith", None):
            arith = wl.arith(ctx, N) if callable(wl.arith) else wl.arith
        elif name.startswith("k_fast"):
            arith = "f32"
        elif "dfpwm" in name or name.startswith("k_df"):
            arith = "i32"
        else:
            arith = "f64"
        line = {
            "metric": "Msamples/s decoded+resampled to 48kHz, 4096-stream batch" if headline else f"Msamples/s ({args.workload})",
            "value": total_samples * args.steps / dt / 1e6,
            "unit": wl.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            # arithmetic type of the path: "f64" = every tap, weight, product and sum is an fp64 value (the reference computes in Lua
            # doubles), whatever the storage type of the rows; the f32-tap kernels report "f32"; DFPWM / ADPCM decode is int32
            "dtype": arith,
            "data": "synthetic",
            "config": {"workload": wl.desc, "streams_per_gpu": args.streams, "streams_total": streams_total, "seconds_per_stream": args.seconds,
                       "parallelism": f"shard{world}", "storage": args.dtype, "exact_math": args.exact_math,
                       "arithmetic": ("fp64 phase-weight table on exact rational positions (k_wave_f64): the reference's arithmetic TYPE, not its operation order — "
                                      "within one f32 ulp of the reference-order kernel" if (headline and args.exact_math == 1) else arith)},
            "windows": {"ms_per_step": [round(w, 4) for w in windows], "median": statistics.median(windows), "min": min(windows), "max": max(windows),
                        "kernel_ms_median": statistics.median(kernel_windows), "note": "window 0 is the contractual K-step region `value` comes from"},
        }
        if not selftest:
            # One launch per step: its algorithmic bytes.  Several launches per step: the TASK's bytes — the input read once + the final output
            # written once (Workload.task_bytes) — over the whole step's kernel time; the per-launch sum (every intermediate counted on both
            # sides) is reported next to it, never as the fraction.
            task = wl.task_bytes() if launches > 1 else None
            roof_bytes = task if task else alg_bytes
            achieved = roof_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            tr = _measured_traffic(name, args) if launches == 1 else None
            tstep = _measured_traffic_step(args.workload, args) if (launches > 1 or tr is None) else None   # (a one-launch step with helper kernels around it is recorded as a step too)
            if tstep:
                tr = tstep["hbm_bytes_per_step"]
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                "traffic": tr, "kernel": name, "kernel_ms": kernel_ms,
                                "launches_per_step": launches, "algorithmic_bytes_per_launch": roof_bytes if launches == 1 else None,
                                "algorithmic_bytes_per_step": roof_bytes,
                                "bytes_per_out_sample": roof_bytes / max(out_samples, 1),
                                "frac_median_window": roof_bytes / (statistics.median(kernel_windows) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if copy_gbs:
                # the same box, the same process, minutes apart: a slow box shows in BOTH numbers, a regression of the kernel only in the first
                line["roofline"]["copy_ceiling_GBs"] = copy_gbs
                line["roofline"]["frac_of_copy"] = achieved / copy_gbs
                line["roofline"]["copy_note"] = copy_how + "; the same read + write traffic as the kernel's algorithmic bytes, median of 12, same process"
            if clocks_before or clocks_after:
                line["roofline"]["clocks"] = {"before": clocks_before, "after": clocks_after, "source": "sysfs pp_dpm_sclk / pp_dpm_mclk (active level) sampled while untimed steps of the workload run, MHz"}
            if tr is not None:
                line["roofline"]["traffic_ratio"] = tr / max(roof_bytes, 1)
            if tstep:
                line["roofline"]["traffic_by_kernel"] = tstep.get("kernels")
            if tr is not None:
                line["roofline"]["traffic_source"] = "committed PMC pass of this command (profiles/traffic.json; FETCH_SIZE / WRITE_SIZE corrected as the guide prescribes), not collected in this run"
            if launches > 1:
                line["roofline"]["launch_bytes_sum"] = alg_bytes
                line["roofline"]["note"] = ("several launches per step: frac = (input bytes + final output bytes) / the step's kernel time; launch_bytes_sum adds every "
                                            "launch's own input + output, intermediates included, and is not a roofline fraction of the task")
            if getattr(wl, "valu_per_unit", None):
                # a step bound by VALU issue, not by bytes (the HBM fraction above says so: it is tiny): the USEFUL instructions — the hot loop's
                # count per unit from the built code object, warm-ups, scans and repairs not counted — over the whole step's kernel
                # time, against what 1024 SIMDs of 16 lanes issue at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
                per, what = wl.valu_per_unit
                peak = 1024 * 16 * 2.4e9 / 1e12
                ach = out_samples * per / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
                line["issue"] = {"bound": "valu", "achieved": ach, "peak": peak, "unit": "T lane-instructions/s", "frac": ach / peak, "per_unit": per, "counted": what}
            if wl.distinct:
                line["config"]["distinct_streams"] = wl.distinct
            if hasattr(wl, "extra"):
                line["config"].update(wl.extra(ctx, N))
        if fast:
            fa = fast[2] / (fast[1] * 1e-3) / 1e9
            line["roofline_fast"] = {"bound": "hbm", "achieved": fa, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fa / HBM_PEAK_GBS, "kernel": fast[0],
                                     "kernel_ms": fast[1], "dtype": "f32", "traffic": _measured_traffic(fast[0], args),
                                     "note": "same batch through the f32-tap kernel (AUKIT_OPT_EXACT_MATH = 0): secondary figure"}
        if ranks_info:
            line["ranks"] = ranks_info
        if distribute:
            line["distribute"] = distribute
        if world == 1 and args.cpu_streams > 0:
            cb = wl.cpu_baseline(args)
            if cb:
                line["cpu_baseline"] = cb
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line, last: RCCL writes its version banner through C stdio, which would otherwise be flushed behind it at exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
