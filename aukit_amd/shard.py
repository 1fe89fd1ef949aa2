"""Sharding a batch of independent streams over the GPUs of one node: one process per GPU, no data-path collective.

No reference function reads another stream, so a batch partitions by stream index and every rank
runs the same single-GPU path on its shard.  The only communication is optional *distribution*: one scatter of the
input byte strings from the rank that holds them and one gather of the outputs, over `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Each peer's share crosses exactly one
xGMI link (≈153 GB/s), ≈30× below the kernels' HBM rate, so callers that can load shards directly should do that
and skip scatter/gather entirely (bench.py does: shard-resident data, weak scaling; `bench.py --distribute` times
the scatter / gather beside it).

Two faces over one transport (`_scatter_bytes` / `_gather_bytes`, flat uint8 tensors + int64 size tables, point-to-point
messages because the shares differ in size):
  * `scatter_batch` / `gather_batch` / `gather_audio` — device to device: the payload tensors live in HBM, a received shard is
    handed to the library with `aukit_batch_wrap_device` (include/aukit_hip.h) and outputs are sent straight from
    `aukit_batch_device_ptr` / `aukit_audio_device_ptr`: no host copy anywhere in the data path;
  * `scatter_streams` / `gather_streams` — lists of Python byte strings (hosts that hold files, and the gloo tests).
"""
import numpy as np


def partition(sizes, world):
    """Contiguous stream ranges [(start, end)) per rank, balanced by input bytes.

    Rank g gets the streams whose cumulative byte midpoint falls in [g/world, (g+1)/world) of the total, which keeps
    the ranges contiguous (outputs concatenate in rank order) and within one stream of the ideal byte split.
    The arithmetic lives in the library (aukit_partition, csrc/group.hip: a Lua host shards with the same cuts); it needs no GPU.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    import os
    from . import _native
    if not os.path.exists(_native.LIB_PATH):   # a launcher host without the built library: the same arithmetic in numpy (tests assert they agree)
        import warnings
        warnings.warn(f"{_native.LIB_PATH} is missing: aukit_amd.shard.partition uses its numpy statement", RuntimeWarning, stacklevel=2)
        return _partition_numpy(sizes, world)
    from . import batch as B   # (a library that is there but broken — a failed dlopen, a missing symbol — must surface, not be papered over: ADVICE r04)
    return B.partition(sizes, world)


def _partition_numpy(sizes, world):
    import numpy as np
    sizes = np.asarray(sizes, dtype=np.float64)
    n = len(sizes)
    if n == 0:
        return [(0, 0)] * world
    total = float(sizes.sum())
    if not total > 0:
        cuts = [n * g // world for g in range(world + 1)]
        return [(cuts[g], cuts[g + 1]) for g in range(world)]
    cum = np.cumsum(sizes)
    owner = np.minimum(((cum - sizes / 2) / total * world).astype(np.int64), world - 1)
    cuts = [0] * (world + 1)
    g = 0
    for i in range(n):
        while g < owner[i]:
            g += 1
            cuts[g] = i
    while g < world:
        g += 1
        cuts[g] = n
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def _dist():
    import torch.distributed as dist
    return dist


# ---------------------------------------------------------------- transport (any device: cuda tensors under nccl, cpu tensors under gloo)
def _scatter_bytes(flat, sizes, src, dev, group):
    """`flat` (rank src: uint8 tensor on `dev` holding every stream back to back) and `sizes` (rank src: per-stream byte counts)
    → (this rank's slice as a uint8 tensor on `dev`, its per-stream sizes as a numpy int64 array, (lo, hi)).
    The slice of rank src is a view of `flat` (no copy); every other rank receives one message."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0] = len(sizes)
    dist.broadcast(meta, src, group=group)
    n = int(meta.item())
    tsz = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    if rank == src and n:
        tsz[:n] = torch.as_tensor(np.asarray(sizes, dtype=np.int64)).to(dev)
    dist.broadcast(tsz, src, group=group)
    sz = tsz[:n].cpu().numpy()  # the size table (a few KB) — the payload never leaves `dev`
    parts = partition(sz, world)
    starts = np.concatenate([[0], np.cumsum(sz)]).astype(np.int64)
    lo, hi = parts[rank]
    if rank == src:
        reqs = []
        for g, (a, b) in enumerate(parts):
            if g == src or starts[b] == starts[a]:
                continue
            reqs.append(dist.isend(flat[int(starts[a]):int(starts[b])], g, group=group))
        for r in reqs:
            r.wait()
        mine = flat[int(starts[lo]):int(starts[hi])]
    else:
        nbytes = int(starts[hi] - starts[lo])
        mine = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            dist.recv(mine, src, group=group)
    return mine, sz[lo:hi], (lo, hi)


def _gather_bytes(flat, sizes, dst, dev, group):
    """Inverse: every rank passes its flat uint8 tensor + per-item sizes; rank dst returns [(tensor, sizes)] in rank order (its own
    entry is `flat` itself), the others None."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = np.asarray(sizes, dtype=np.int64)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(sizes)], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    mine = torch.zeros(mx, dtype=torch.int64, device=dev)
    if len(sizes):
        mine[:len(sizes)] = torch.as_tensor(sizes).to(dev)
    allsz = [torch.zeros(mx, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, mine, group=group)
    if rank != dst:
        if flat.numel():
            dist.send(flat, dst, group=group)
        return None
    out = []
    for g in range(world):
        sz = allsz[g][:counts[g]].cpu().numpy()
        if g == dst:
            out.append((flat, sz))
            continue
        nbytes = int(sz.sum())
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            dist.recv(t, g, group=group)
        out.append((t, sz))
    return out


# ---------------------------------------------------------------- host face: lists of byte strings
def scatter_streams(streams, src=0, device=None, group=None):
    """Rank `src` passes the full list of byte strings; every rank returns (its shard as a list of bytes, (start, end))."""
    import torch
    dev = device if device is not None else torch.device("cpu")
    flat = sizes = None
    if _dist().get_rank(group) == src:
        sizes = [len(s) for s in streams]
        blob = b"".join(bytes(s) for s in streams)
        flat = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8)[:len(blob)].to(dev)
    mine, sz, (lo, hi) = _scatter_bytes(flat, sizes, src, dev, group)
    blob = mine.cpu().numpy().tobytes()
    out, p = [], 0
    for k in sz:
        out.append(blob[p:p + int(k)])
        p += int(k)
    return out, (lo, hi)


def gather_streams(local, dst=0, device=None, group=None):
    """Inverse of scatter_streams: rank `dst` returns the concatenation (in rank order) of every rank's list of byte strings."""
    import torch
    dev = device if device is not None else torch.device("cpu")
    blob = b"".join(bytes(s) for s in local)
    flat = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8)[:len(blob)].to(dev)
    got = _gather_bytes(flat, [len(s) for s in local], dst, dev, group)
    if got is None:
        return None
    out = []
    for t, sz in got:
        data = t.cpu().numpy().tobytes()
        p = 0
        for k in sz:
            out.append(data[p:p + int(k)])
            p += int(k)
    return out


# ---------------------------------------------------------------- device face: aukit batches / audios, HBM to HBM
class _DevMem:
    """(pointer, bytes) of device memory the library owns, as something `torch.as_tensor` can view without a copy"""

    def __init__(self, ptr, nbytes, keep):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        self._keep = keep


def device_view(ptr, nbytes, device, keep=None):
    """uint8 torch tensor over [ptr, ptr + nbytes) on `device` (zero-copy; `keep` is held alive with it)"""
    import torch
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    t = torch.as_tensor(_DevMem(ptr, nbytes, keep), device=device)
    t._aukit_keep = keep
    return t


def scatter_batch(ctx, batch, src=0, device=None, group=None):
    """Rank `src` passes an aukit_amd.batch.Batch (the others None); every rank returns (its shard as a Batch, (start, end)).
    The shard is the received device tensor wrapped with aukit_batch_wrap_device — rank src's is a view of its own bytes."""
    import torch
    from . import batch as B
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    flat = sizes = None
    if _dist().get_rank(group) == src:
        offs = batch.offsets()
        sizes = np.diff(offs.astype(np.int64))
        flat = device_view(batch.device_ptr(), int(offs[-1]), dev, keep=batch)
    mine, sz, (lo, hi) = _scatter_bytes(flat, sizes, src, dev, group)
    offs = np.concatenate([[0], np.cumsum(sz)]).astype(np.uint64)
    return wrap_tensor(ctx, mine, offs), (lo, hi)


def wrap_tensor(ctx, t, offsets):
    """a uint8 device tensor + stream offsets → Batch (zero-copy; the tensor is kept alive by the Batch)"""
    import torch
    from . import batch as B
    if t.numel() == 0:
        t = torch.zeros(16, dtype=torch.uint8, device=t.device)  # a valid pointer for an empty shard
    return B.Batch.wrap(ctx, t.data_ptr(), offsets, keep=t)


def gather_batch(batch, dst=0, device=None, group=None):
    """Every rank passes its output Batch (e.g. re-encoded DFPWM); rank `dst` returns [(uint8 device tensor, per-stream sizes)] in rank
    order — still in HBM — and the others None."""
    import torch
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    offs = batch.offsets()
    flat = device_view(batch.device_ptr(), int(offs[-1]), dev, keep=batch)
    return _gather_bytes(flat, np.diff(offs.astype(np.int64)), dst, dev, group)


def gather_audio(audio, dst=0, device=None, group=None):
    """Every rank passes its output AudioBatch; rank `dst` returns, in rank order, [(uint8 device tensor of the rows as the library laid
    them out, {"lens", "row_off", "row_stride", "channels", "dtype", "rate"})] — still in HBM — and the others None.
    Row r of stream s, channel c starts at element row_off[s] + c * row_stride[s] of the tensor viewed as `dtype`."""
    import torch
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    inf = audio.info()
    lens, off, stride = audio.layout()
    esz = {0: 8, 1: 4, 2: 1}[inf["dtype"]]
    total = int((off[-1] + stride[-1] * inf["channels"]) * esz) if len(lens) else 0
    flat = device_view(audio.device_ptr(), total, dev, keep=audio)
    # the layout travels as the "sizes" table: [n, channels, dtype, rate_bits, lens..., off..., stride...]
    table = np.concatenate([[len(lens), inf["channels"], inf["dtype"], np.float64(inf["sample_rate"]).view(np.int64)], lens.astype(np.int64), off.astype(np.int64),
                            stride.astype(np.int64)]).astype(np.int64)
    got = _gather_bytes_with_table(flat, table, dst, dev, group)
    if got is None:
        return None
    out = []
    for t, tb in got:
        n = int(tb[0])
        out.append((t, {"channels": int(tb[1]), "dtype": int(tb[2]), "rate": float(np.int64(tb[3]).view(np.float64)), "lens": tb[4:4 + n], "row_off": tb[4 + n:4 + 2 * n],
                        "row_stride": tb[4 + 2 * n:4 + 3 * n]}))
    return out


def _gather_bytes_with_table(flat, table, dst, dev, group):
    """_gather_bytes where the int64 table is opaque metadata and the payload length is flat.numel()"""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    hdr = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(hdr, torch.tensor([len(table), flat.numel()], dtype=torch.int64, device=dev), group=group)
    hdr = [h.cpu().numpy() for h in hdr]
    mx = max(max(int(h[0]) for h in hdr), 1)
    mine = torch.zeros(mx, dtype=torch.int64, device=dev)
    mine[:len(table)] = torch.as_tensor(np.asarray(table, dtype=np.int64)).to(dev)
    alltb = [torch.zeros(mx, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(alltb, mine, group=group)
    if rank != dst:
        if flat.numel():
            dist.send(flat, dst, group=group)
        return None
    out = []
    for g in range(world):
        tb = alltb[g][:int(hdr[g][0])].cpu().numpy()
        if g == dst:
            out.append((flat, tb))
            continue
        nbytes = int(hdr[g][1])
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            dist.recv(t, g, group=group)
        out.append((t, tb))
    return out
