"""Batch-level host API over the C ABI: N independent streams per call.

This is the MI355X-shaped face of the hot path (one call = one fused kernel over the whole batch);
`aukit_amd.aukit` mirrors the reference's single-stream Lua API on top of it.
"""
import ctypes as C

import numpy as np

from . import _native as N


def make_desc(codec, channels=1, sample_rate=48000.0, bit_depth=8, data_type="signed", big_endian=False, interleaved=True,
              ulaw=True, top_first=True, block_align=0, coefficients=None, predictor=None, step_index=None):
    d = N.CodecDesc()
    d.codec = codec
    d.channels = int(channels)
    d.sample_rate = float(sample_rate)
    d.bit_depth = int(bit_depth)
    d.data_type = N.PCM_TYPE[data_type] if isinstance(data_type, str) else int(data_type)
    d.big_endian = int(bool(big_endian))
    d.interleaved = int(bool(interleaved))
    d.ulaw = int(bool(ulaw))
    d.top_first = int(bool(top_first))
    d.block_align = int(block_align)
    if coefficients:
        d.ncoef = len(coefficients[0])
        for i, v in enumerate(coefficients[0]):
            d.coef1[i] = int(v)
        for i, v in enumerate(coefficients[1]):
            d.coef2[i] = int(v)
    if predictor is not None:
        for i, v in enumerate(predictor):
            d.predictor[i] = int(v)
    if step_index is not None:
        for i, v in enumerate(step_index):
            d.step_index[i] = int(v)
    return d


class Context:
    """One GPU + one HIP stream (aukit_ctx)."""

    def __init__(self, device=0, dtype=N.F64):
        self._h = C.c_void_p()
        N.check(N.lib().aukit_ctx_create(C.byref(self._h), int(device)))
        self.device = device
        self.dtype = dtype
        N.check(N.lib().aukit_ctx_set_dtype(self._h, dtype))

    def close(self):
        if self._h:
            N.lib().aukit_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        N.check(N.lib().aukit_ctx_set_stream(self._h, C.c_void_p(int(hip_stream_ptr))))

    def sync(self):
        N.check(N.lib().aukit_ctx_sync(self._h))

    def timer_begin(self):
        N.check(N.lib().aukit_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        N.check(N.lib().aukit_timer_end(self._h, C.byref(ms)))
        return ms.value

    def set_kernel_timing(self, on=True):
        N.check(N.lib().aukit_ctx_set_kernel_timing(self._h, int(on)))

    def timer_stats(self):
        """(kernel launches, algorithmic bytes) since timer_begin"""
        nl, nb = C.c_uint64(), C.c_uint64()
        N.check(N.lib().aukit_timer_stats(self._h, C.byref(nl), C.byref(nb)))
        return nl.value, nb.value

    def last_kernel(self):
        name = C.c_char_p()
        ms = C.c_float()
        nb = C.c_uint64()
        N.check(N.lib().aukit_ctx_last_kernel(self._h, C.byref(name), C.byref(ms), C.byref(nb)))
        return (name.value or b"").decode(), ms.value, nb.value

    def set_option(self, option, value):
        N.check(N.lib().aukit_ctx_set_option(self._h, int(option), int(value)))

    def counter(self, which):
        """a counter of the most recent call that produced it (set_option(OPT_COLLECT_STATS, 1) first): N.COUNTER_*"""
        v = C.c_uint64()
        N.check(N.lib().aukit_ctx_get_counter(self._h, int(which), C.byref(v)))
        return v.value

    def set_sinc_window(self, w):
        N.check(N.lib().aukit_ctx_set_sinc_window(self._h, int(w)))


class _GroupContext(Context):
    """a member context of a Group: owned by the group (never destroyed on its own)"""

    def __init__(self, handle, device, dtype):
        self._h = C.c_void_p(handle)
        self.device = device
        self.dtype = dtype

    def close(self):
        self._h = C.c_void_p()


def partition(sizes, world):
    """aukit_partition: contiguous, byte-balanced stream ranges [(start, end)) per rank"""
    sz = np.ascontiguousarray(sizes, dtype=np.uint64)
    cuts = np.zeros(int(world) + 1, dtype=np.uint32)
    N.check(N.lib().aukit_partition(sz.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(sz)), C.c_uint32(int(world)), cuts.ctypes.data_as(C.POINTER(C.c_uint32))))
    return [(int(cuts[g]), int(cuts[g + 1])) for g in range(int(world))]


class Group:
    """several GPUs of one node in ONE process (aukit_group): a context per device, scatter of a batch's streams, gather of the results"""

    def __init__(self, devices, dtype=N.F64):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        N.check(N.lib().aukit_group_create(C.byref(self._h), devs, C.c_uint32(len(devices))))
        self.devices = list(devices)
        self.contexts = []
        for r, d in enumerate(devices):
            c = _GroupContext(N.lib().aukit_group_ctx(self._h, C.c_uint32(r)), d, dtype)
            N.check(N.lib().aukit_ctx_set_dtype(c._h, dtype))
            self.contexts.append(c)

    def transport(self):
        n, t = C.c_uint32(), C.c_int()
        N.check(N.lib().aukit_group_info(self._h, C.byref(n), C.byref(t)))
        return "rccl" if t.value else "peer"

    def scatter(self, whole, root=0):
        """→ ([Batch per member], [(start, end) per member]); the root's shard is a view of `whole`"""
        W = len(self.devices)
        hs = (C.c_void_p * W)()
        cuts = np.zeros(W + 1, dtype=np.uint32)
        # (on failure the library frees and nulls what it allocated in this call: nothing to release here — ADVICE r03)
        N.check(N.lib().aukit_group_scatter(self._h, C.c_uint32(root), whole._h, hs, cuts.ctypes.data_as(C.POINTER(C.c_uint32))))
        shards = []
        for r in range(W):
            b = Batch(self.contexts[r], C.c_void_p(hs[r]))
            b._keep = whole
            shards.append(b)
        return shards, [(int(cuts[g]), int(cuts[g + 1])) for g in range(W)]

    def gather_audio(self, parts, root=0, out=None):
        out = out if out is not None else AudioBatch(self.contexts[root])
        hs = (C.c_void_p * len(parts))(*[p._h for p in parts])
        N.check(N.lib().aukit_group_gather_audio(self._h, C.c_uint32(root), hs, C.byref(out._h)))
        return out

    def gather_batch(self, parts, root=0, out=None):
        out = out if out is not None else Batch(self.contexts[root], C.c_void_p())
        hs = (C.c_void_p * len(parts))(*[p._h for p in parts])
        N.check(N.lib().aukit_group_gather_batch(self._h, C.c_uint32(root), hs, C.byref(out._h)))
        return out

    def run(self, lists):
        """aukit_group_run: lists[r] = member r's calls, each a dict {"op": name, ...} with the entry point's arguments:
        decode / decode_resample / stream_decode: batch, desc, dtype, [new_rate, interp, mono], out (AudioBatch), [chunks_out];
        resample / mono / encode_pcm: audio, out; effect: audio, name, args; dfpwm_encode: audio, interleaved, out (Batch);
        dfpwm_transcode_mono: batch, channels, out (Batch).  The members' lists run side by side on the group's worker threads."""
        W = len(self.devices)
        if len(lists) != W:
            raise ValueError(f"Group.run: {len(lists)} call lists for a group of {W} members (one list per member, empty lists allowed)")
        n_per = max([len(l) for l in lists] + [0])
        calls = (N.GroupCall * (W * max(n_per, 1)))()
        keep = []
        for r in range(W):
            for k, c in enumerate(lists[r]):
                g = calls[r * n_per + k]
                g.op = N.GOP[c["op"]]
                g.dtype = int(c.get("dtype", self.contexts[r].dtype))
                it = c.get("interp", "linear")
                g.interp = N.INTERP[it] if isinstance(it, str) else int(it)
                g.mono = int(bool(c.get("mono", False)))
                if "batch" in c:
                    g.batch = c["batch"]._h.value
                if "desc" in c:
                    keep.append(c["desc"])
                    g.desc = C.addressof(c["desc"])
                if "audio" in c:
                    g.audio = c["audio"]._h.value
                out = c.get("out")
                if isinstance(out, AudioBatch):
                    g.out_audio = C.addressof(out._h)
                elif isinstance(out, Batch):
                    g.out_batch = C.addressof(out._h)
                if c.get("chunks_out") is not None:
                    g.out_chunks = C.addressof(c["chunks_out"])
                g.new_rate = float(c.get("new_rate", 0.0))
                if c["op"] == "effect":
                    g.effect_id = N.FX[c["name"]]
                    a = [float(x) for x in c.get("args", ())]
                    g.nargs = len(a)
                    for i, x in enumerate(a):
                        g.args[i] = x
                g.channels = int(c.get("channels", 1))
                g.interleaved = int(bool(c.get("interleaved", True)))
                g.bit_depth = int(c.get("bit_depth", 8))
                dt = c.get("data_type", "signed")
                g.data_type = N.PCM_TYPE[dt] if isinstance(dt, str) else int(dt)
        N.check(N.lib().aukit_group_run(self._h, calls, C.c_uint32(n_per)))

    def last_run(self):
        """→ [(start_ms, end_ms) per member] of the last run()"""
        W = len(self.devices)
        a, b = (C.c_double * W)(), (C.c_double * W)()
        N.check(N.lib().aukit_group_last_run(self._h, a, b))
        return [(a[r], b[r]) for r in range(W)]

    def sync(self):
        N.check(N.lib().aukit_group_sync(self._h))

    def close(self):
        if self._h:
            for c in self.contexts:
                c.close()
            N.lib().aukit_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """N byte strings resident on the device (aukit_batch)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        self._keep = None

    @classmethod
    def upload(cls, ctx, streams):
        streams = [bytes(s) if not isinstance(s, (bytes, bytearray, memoryview)) else s for s in streams]
        offs = np.zeros(len(streams) + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in streams], out=offs[1:])
        blob = b"".join(streams)
        buf = (C.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob if blob else b"\0")
        h = C.c_void_p()
        N.check(N.lib().aukit_batch_upload(ctx._h, C.byref(h), buf, offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(streams))))
        return cls(ctx, h)

    @classmethod
    def wrap(cls, ctx, device_ptr, offsets, keep=None):
        """Zero-copy view of bytes already on this GPU (e.g. a torch uint8 tensor's data_ptr())."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        N.check(N.lib().aukit_batch_wrap_device(ctx._h, C.byref(h), C.c_void_p(int(device_ptr)), offs.ctypes.data_as(C.c_void_p), C.c_uint32(len(offs) - 1)))
        b = cls(ctx, h)
        b._keep = keep
        return b

    def info(self):
        n = C.c_uint32()
        tot = C.c_uint64()
        N.check(N.lib().aukit_batch_info(self._h, C.byref(n), C.byref(tot)))
        return n.value, tot.value

    def offsets(self):
        n, _ = self.info()
        offs = np.zeros(n + 1, dtype=np.uint64)
        N.check(N.lib().aukit_batch_offsets(self._h, offs.ctypes.data_as(C.POINTER(C.c_uint64))))
        return offs

    def device_ptr(self):
        return N.lib().aukit_batch_device_ptr(self._h)

    def download(self):
        n, tot = self.info()
        buf = np.zeros(max(tot, 1), dtype=np.uint8)
        N.check(N.lib().aukit_batch_download(self.ctx._h, self._h, buf.ctypes.data_as(C.POINTER(C.c_uint8))))
        offs = self.offsets()
        return [bytes(buf[int(offs[i]):int(offs[i + 1])]) for i in range(n)]

    def free(self):
        if self._h:
            N.lib().aukit_batch_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class AudioBatch:
    """N aukit.Audio objects (same channel count and sample rate) resident on the device (aukit_audio)."""

    def __init__(self, ctx, handle=None):
        self.ctx = ctx
        self._h = handle if handle is not None else C.c_void_p()

    @classmethod
    def upload(cls, ctx, streams, sample_rate, dtype=None):
        """streams: list (per stream) of list (per channel) of 1-D arrays of equal length."""
        dtype = ctx.dtype if dtype is None else dtype
        channels = len(streams[0])
        lens = np.array([len(s[0]) for s in streams], dtype=np.uint64)
        flat = np.concatenate([np.asarray(ch, dtype=np.float64).ravel() for s in streams for ch in s]) if len(streams) else np.zeros(0)
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        if flat.size == 0:
            flat = np.zeros(1)
        h = C.c_void_p()
        N.check(N.lib().aukit_audio_upload(ctx._h, C.byref(h), flat.ctypes.data_as(C.POINTER(C.c_double)), lens.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           C.c_uint32(len(streams)), channels, C.c_double(sample_rate), dtype))
        return cls(ctx, h)

    def info(self):
        n = C.c_uint32()
        ch = C.c_int()
        rate = C.c_double()
        dt = C.c_int()
        tot = C.c_uint64()
        N.check(N.lib().aukit_audio_info(self._h, C.byref(n), C.byref(ch), C.byref(rate), C.byref(dt), C.byref(tot)))
        return dict(n=n.value, channels=ch.value, sample_rate=rate.value, dtype=dt.value, total_elems=tot.value)

    def layout(self):
        n = self.info()["n"]
        lens = np.zeros(n, dtype=np.uint64)
        off = np.zeros(n, dtype=np.uint64)
        stride = np.zeros(n, dtype=np.uint64)
        N.check(N.lib().aukit_audio_layout(self._h, lens.ctypes.data_as(C.POINTER(C.c_uint64)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           stride.ctypes.data_as(C.POINTER(C.c_uint64))))
        return lens, off, stride

    def device_ptr(self):
        return N.lib().aukit_audio_device_ptr(self._h)

    def download(self):
        """→ list (per stream) of list (per channel) of float64 arrays (exact for every storage dtype)."""
        inf = self.info()
        lens, _, _ = self.layout()
        total = int(lens.sum()) * inf["channels"]
        buf = np.zeros(max(total, 1), dtype=np.float64)
        N.check(N.lib().aukit_audio_download(self.ctx._h, self._h, buf.ctypes.data_as(C.POINTER(C.c_double))))
        out, p = [], 0
        for s in range(inf["n"]):
            chs = []
            for _ in range(inf["channels"]):
                chs.append(buf[p:p + int(lens[s])].copy())
                p += int(lens[s])
            out.append(chs)
        return out

    def clone(self):
        h = C.c_void_p()
        N.check(N.lib().aukit_audio_clone(self.ctx._h, self._h, C.byref(h)))
        return AudioBatch(self.ctx, h)

    def free(self):
        if self._h:
            N.lib().aukit_audio_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Chunks:
    """Per-stream chunk metadata of a stream.* run (lengths and the iterator's position values)."""

    def __init__(self, handle):
        n = C.c_uint32()
        mx = C.c_uint32()
        N.check(N.lib().aukit_chunks_info(handle, C.byref(n), C.byref(mx)))
        self.n, self.max_chunks = n.value, mx.value
        m = max(self.max_chunks, 1)
        self.nchunks = np.zeros(self.n, dtype=np.uint32)
        self.lens = np.zeros((self.n, m), dtype=np.uint32)
        self.pos = np.zeros((self.n, m), dtype=np.float64)
        self.status = np.zeros(self.n, dtype=np.int32)
        self.length_seconds = np.zeros(self.n, dtype=np.float64)
        N.check(N.lib().aukit_chunks_get(handle, self.nchunks.ctypes.data_as(C.POINTER(C.c_uint32)), self.lens.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         self.pos.ctypes.data_as(C.POINTER(C.c_double)), self.status.ctypes.data_as(C.POINTER(C.c_int32)),
                                         self.length_seconds.ctypes.data_as(C.POINTER(C.c_double))))
        N.lib().aukit_chunks_free(handle)


def _interp(i):
    return N.INTERP[i] if isinstance(i, str) else int(i)


def decode(ctx, batch, desc, dtype=None, out=None):
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_decode(ctx._h, batch._h, C.byref(desc), ctx.dtype if dtype is None else dtype, C.byref(out._h)))
    return out


def decode_table(ctx, tables, desc, out=None):
    """aukit.pcm on TABLES of numbers (aukit.lua:1077-1096): `tables` = one sequence of numbers per stream.  Storage = ctx.dtype."""
    import numpy as np
    out = out if out is not None else AudioBatch(ctx)
    arrs = [np.ascontiguousarray(t, dtype=np.float64).ravel() for t in tables]
    offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        offs[1:] = np.cumsum([a.size for a in arrs])
    vals = np.concatenate(arrs) if arrs and offs[-1] else np.zeros(1, dtype=np.float64)
    N.check(N.lib().aukit_decode_table(ctx._h, vals.ctypes.data_as(C.POINTER(C.c_double)), offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(arrs)),
                                       C.byref(desc), C.byref(out._h)))
    return out


def decode_nibbles(ctx, tables, desc, dtype=None, out=None):
    """aukit.adpcm on TABLES of nibbles (aukit.lua:1183-1184, :1232-1238): `tables` = one sequence of integers 0..15 per stream"""
    import numpy as np
    out = out if out is not None else AudioBatch(ctx)
    arrs = [np.ascontiguousarray(t, dtype=np.uint8).ravel() for t in tables]
    offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        offs[1:] = np.cumsum([len(a) for a in arrs])
    flat = np.concatenate(arrs) if arrs and int(offs[-1]) else np.zeros(1, dtype=np.uint8)
    N.check(N.lib().aukit_decode_nibbles(ctx._h, flat.ctypes.data_as(C.POINTER(C.c_uint8)), offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(arrs)), C.byref(desc),
                                         ctx.dtype if dtype is None else dtype, C.byref(out._h)))
    return out


def stream_decode_table(ctx, tables, desc, interp, mono=False, dtype=None, out=None):
    """aukit.stream.pcm on TABLES of numbers (aukit.lua:2255-2290): every iterator call at once, like stream_decode"""
    import numpy as np
    out = out if out is not None else AudioBatch(ctx)
    arrs = [np.ascontiguousarray(t, dtype=np.float64).ravel() for t in tables]
    offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        offs[1:] = np.cumsum([len(a) for a in arrs])
    flat = np.concatenate(arrs) if arrs and int(offs[-1]) else np.zeros(1)
    ch = C.c_void_p()
    N.check(N.lib().aukit_stream_decode_table(ctx._h, flat.ctypes.data_as(C.POINTER(C.c_double)), offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(arrs)), C.byref(desc),
                                              _interp(interp), int(bool(mono)), ctx.dtype if dtype is None else dtype, C.byref(out._h), C.byref(ch)))
    return out, Chunks(ch)


def decode_resample(ctx, batch, desc, new_rate, interp, dtype=None, out=None):
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_decode_resample(ctx._h, batch._h, C.byref(desc), C.c_double(new_rate), _interp(interp),
                                          ctx.dtype if dtype is None else dtype, C.byref(out._h)))
    return out


def resample(ctx, audio, new_rate, interp, out=None):
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_resample(ctx._h, audio._h, C.c_double(new_rate), _interp(interp), C.byref(out._h)))
    return out


def mono(ctx, audio, out=None):
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_mono(ctx._h, audio._h, C.byref(out._h)))
    return out


def mix(ctx, audios, amplifier=1.0, out=None):
    out = out if out is not None else AudioBatch(ctx)
    arr = (C.c_void_p * len(audios))(*[a._h for a in audios])
    N.check(N.lib().aukit_mix(ctx._h, arr, len(audios), C.c_double(amplifier), C.byref(out._h)))
    return out


def effect(ctx, audio, name, *args):
    a = (C.c_double * max(len(args), 1))(*[float(x) for x in args])
    N.check(N.lib().aukit_effect(ctx._h, audio._h, N.FX[name], a, len(args)))
    return audio


def dfpwm_encode(ctx, audio, interleaved=True, out=None):
    out = out if out is not None else Batch(ctx, C.c_void_p())
    N.check(N.lib().aukit_dfpwm_encode(ctx._h, audio._h, int(bool(interleaved)), C.byref(out._h)))
    return out


def dfpwm_transcode_mono(ctx, batch, channels, out=None):
    """Fused aukit.dfpwm(d, channels, sr):mono():dfpwm() over a batch."""
    out = out if out is not None else Batch(ctx, C.c_void_p())
    N.check(N.lib().aukit_dfpwm_transcode_mono(ctx._h, batch._h, int(channels), C.byref(out._h)))
    return out


def encode_pcm(ctx, audio, bit_depth=8, data_type="signed", interleaved=True, out=None):
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_encode_pcm(ctx._h, audio._h, int(bit_depth), N.PCM_TYPE[data_type] if isinstance(data_type, str) else int(data_type),
                                     int(bool(interleaved)), C.byref(out._h)))
    return out


def _group(audios):
    return (C.c_void_p * len(audios))(*[a._h for a in audios])


def concat(ctx, audios, out=None):  # Audio:concat  aukit.lua:695
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_concat(ctx._h, _group(audios), len(audios), C.byref(out._h)))
    return out


def sub(ctx, audio, start=0.0, last=0.0, out=None):  # Audio:sub  aukit.lua:725
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_sub(ctx._h, audio._h, C.c_double(start), C.c_double(last), C.byref(out._h)))
    return out


def combine(ctx, audios, out=None):  # Audio:combine  aukit.lua:751
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_combine(ctx._h, _group(audios), len(audios), C.byref(out._h)))
    return out


def split(ctx, audio, channels, out=None):  # one result of Audio:split  aukit.lua:781 (1-based channel numbers)
    out = out if out is not None else AudioBatch(ctx)
    arr = (C.c_int32 * max(len(channels), 1))(*[int(c) for c in channels])
    N.check(N.lib().aukit_split(ctx._h, audio._h, arr, len(channels), C.byref(out._h)))
    return out


def rep(ctx, audio, count, out=None):  # Audio:rep  aukit.lua:839
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_rep(ctx._h, audio._h, C.c_double(count), C.byref(out._h)))
    return out


def reverse(ctx, audio, out=None):  # Audio:reverse  aukit.lua:856
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_reverse(ctx._h, audio._h, C.byref(out._h)))
    return out


WAVES = {"none": N.WAVE_NONE, "sine": N.WAVE_SINE, "triangle": N.WAVE_TRIANGLE, "sawtooth": N.WAVE_SAWTOOTH, "square": N.WAVE_SQUARE}


def tone(ctx, n, frequency, duration, amplitude=1.0, wave="sine", duty=0.5, channels=1, sample_rate=48000.0, dtype=None, out=None):
    """aukit.tone (aukit.lua:1808); wave="none" is aukit.new (:1783).  `n` identical audios."""
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_tone(ctx._h, int(n), C.c_double(frequency), C.c_double(duration), C.c_double(amplitude), WAVES[wave] if isinstance(wave, str) else int(wave),
                               C.c_double(duty), int(channels), C.c_double(sample_rate), ctx.dtype if dtype is None else dtype, C.byref(out._h)))
    return out


def noise(ctx, n, duration, amplitude=1.0, channels=1, sample_rate=48000.0, seed=0, dtype=None, out=None):
    """aukit.noise (aukit.lua:1840) on the device: Philox4x32-10 keyed by `seed` — the reference's math.random stream is not reproducible."""
    out = out if out is not None else AudioBatch(ctx)
    N.check(N.lib().aukit_noise(ctx._h, int(n), C.c_double(duration), C.c_double(amplitude), int(channels), C.c_double(sample_rate), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                ctx.dtype if dtype is None else dtype, C.byref(out._h)))
    return out


def pack_pcm(ctx, audio, bit_depth=8, data_type="signed", big_endian=False, interleaved=True, int_mode=N.PACK_TRUNC, out=None):
    """aukit.pack(audio:pcm(...), ...) (aukit.lua:901, :1861) → Batch of byte strings."""
    out = out if out is not None else Batch(ctx, C.c_void_p())
    N.check(N.lib().aukit_pack_pcm(ctx._h, audio._h, int(bit_depth), N.PCM_TYPE[data_type] if isinstance(data_type, str) else int(data_type),
                                   int(bool(big_endian)), int(bool(interleaved)), int(int_mode), C.byref(out._h)))
    return out


def stream_decode(ctx, batch, desc, interp, mono=False, dtype=None, out=None):
    out = out if out is not None else AudioBatch(ctx)
    ch = C.c_void_p()
    N.check(N.lib().aukit_stream_decode(ctx._h, batch._h, C.byref(desc), _interp(interp), int(bool(mono)),
                                        ctx.dtype if dtype is None else dtype, C.byref(out._h), C.byref(ch)))
    return out, Chunks(ch)


class StreamHandle:
    """aukit.stream.<codec> fed piece by piece (aukit_stream_open / feed / finish / next): the chunks are those of the string version for the
    concatenation of everything fed, whatever the feeding pattern (include/aukit_hip.h)."""

    def __init__(self, ctx, desc, interp, mono=False, dtype=None, cap=1 << 20):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._cap = int(cap)
        N.check(N.lib().aukit_stream_open(ctx._h, C.byref(desc), _interp(interp), int(bool(mono)), ctx.dtype if dtype is None else dtype, C.byref(self._h)))

    def feed(self, data):
        b = bytes(data)
        N.check(N.lib().aukit_stream_feed(self._h, b, C.c_uint64(len(b))))

    def finish(self):
        N.check(N.lib().aukit_stream_finish(self._h))

    def next(self):
        """→ ("chunk", [per-channel float64 arrays], pos) | ("need_input", None, None) | ("end", None, None)"""
        if getattr(self, "_buf", None) is None:  # one buffer per handle (a fresh zero-filled 64 MB array per chunk was most of a chunk's cost)
            self._buf_ch = 2
            self._buf = np.empty(self._cap * self._buf_ch, dtype=np.float64)
        ln, ch, st, pos = C.c_uint32(), C.c_int32(), C.c_int32(), C.c_double()
        while True:
            buf = self._buf
            rc = N.lib().aukit_stream_next(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(buf.size), C.c_uint32(self._cap), C.byref(ln), C.byref(ch), C.byref(pos),
                                           C.byref(st))
            if rc == N.E_ARG and ch.value > self._buf_ch and ln.value <= self._cap:  # more channels than the buffer was sized for: the chunk is still there
                self._buf_ch = ch.value
                self._buf = np.empty(self._cap * self._buf_ch, dtype=np.float64)
                continue
            N.check(rc)
            break
        if st.value == N.STREAM_CHUNK:
            return "chunk", [buf[c * self._cap: c * self._cap + ln.value].copy() for c in range(ch.value)], pos.value
        return ("need_input" if st.value == N.STREAM_NEED_INPUT else "end"), None, None

    def length(self):
        s = C.c_double()
        N.check(N.lib().aukit_stream_length(self._h, C.byref(s)))
        return s.value

    def resident(self):
        """→ (bytes resident on the device, bytes dropped in front of them, input bytes of every decode so far summed)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(N.lib().aukit_stream_resident(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self._h:
            N.lib().aukit_stream_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


AUKIT_MAX_CH = N.MAX_CH
