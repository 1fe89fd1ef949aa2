"""Sharding a batch of independent streams over the GPUs of one node: one process per GPU, no data-path collective.

No reference function reads another stream, so a batch partitions by stream index and every rank
runs the same single-GPU path on its shard.  The only communication is optional *distribution*: one scatter of the
input byte strings from the rank that holds them and one gather of the outputs, over `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Each peer's share crosses exactly one
xGMI link (≈153 GB/s), ≈30× below the kernels' HBM rate, so callers that can load shards directly should do that
and skip scatter/gather entirely (bench.py does: shard-resident data, weak scaling).
"""
import numpy as np


def partition(sizes, world):
    """Contiguous stream ranges [(start, end)) per rank, balanced by input bytes.

    Rank g gets the streams whose cumulative byte midpoint falls in [g/world, (g+1)/world) of the total, which keeps
    the ranges contiguous (outputs concatenate in rank order) and within one stream of the ideal byte split.
    """
    sizes = np.asarray(sizes, dtype=np.float64)
    n = len(sizes)
    if world < 1:
        raise ValueError("world must be >= 1")
    if n == 0:
        return [(0, 0)] * world
    total = float(sizes.sum())
    if total <= 0:  # all empty: split by count
        cuts = [(n * g) // world for g in range(world + 1)]
    else:
        mid = np.cumsum(sizes) - sizes / 2
        owner = np.minimum((mid / total * world).astype(np.int64), world - 1)
        cuts = [int(np.searchsorted(owner, g, side="left")) for g in range(world)] + [n]
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def _dist():
    import torch.distributed as dist
    return dist


def scatter_streams(streams, src=0, device=None, group=None):
    """Rank `src` passes the full list of byte strings; every rank returns (its shard as a list of bytes, (start, end)).

    Sizes travel as one broadcast int64 tensor; payloads as one point-to-point uint8 message per peer (the
    variable-size scatter RCCL lacks), so each peer's bytes cross its own xGMI link once.
    """
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0] = len(streams)
    dist.broadcast(meta, src, group=group)
    n = int(meta.item())
    sizes = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    if rank == src:
        sizes[:n] = torch.tensor([len(s) for s in streams], dtype=torch.int64)
    dist.broadcast(sizes, src, group=group)
    sz = sizes[:n].cpu().numpy()
    parts = partition(sz, world)
    lo, hi = parts[rank]
    if rank == src:
        mine = None
        for g, (a, b) in enumerate(parts):
            blob = b"".join(bytes(s) for s in streams[a:b])
            if g == src:
                mine = blob
                continue
            t = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8).to(dev)
            dist.send(t, g, group=group)
        blob = mine
    else:
        nbytes = int(sz[lo:hi].sum())
        t = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        dist.recv(t, src, group=group)
        blob = bytes(t[:nbytes].cpu().numpy().tobytes())
    out, p = [], 0
    for k in range(lo, hi):
        out.append(blob[p:p + int(sz[k])])
        p += int(sz[k])
    return out, (lo, hi)


def gather_streams(local, dst=0, device=None, group=None):
    """Inverse of scatter_streams: rank `dst` returns the concatenation (in rank order) of every rank's list of byte strings."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(local)], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    mine = torch.zeros(mx, dtype=torch.int64, device=dev)
    mine[:len(local)] = torch.tensor([len(s) for s in local], dtype=torch.int64) if local else mine[:0]
    allsz = [torch.zeros(mx, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, mine, group=group)
    blob = b"".join(bytes(s) for s in local)
    if rank != dst:
        t = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8).to(dev)
        dist.send(t, dst, group=group)
        return None
    out = []
    for g in range(world):
        sz = allsz[g][:counts[g]].cpu().numpy()
        if g == dst:
            data = blob
        else:
            nbytes = int(sz.sum())
            t = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            dist.recv(t, g, group=group)
            data = bytes(t[:nbytes].cpu().numpy().tobytes())
        p = 0
        for k in range(counts[g]):
            out.append(data[p:p + int(sz[k])])
            p += int(sz[k])
    return out
