"""aukit_amd — MI355X-native (gfx950) implementation of AUKit's batched decode → resample → effects hot path.

Layers:
  csrc/            hand-written HIP kernels + the C ABI (include/aukit_hip.h) → libaukit_hip.so
  _native.py       ctypes loader (no CPU fallback: raises when the library or the GPU is missing)
  batch.py         batch-level host API (N streams per call)
  aukit.py         mirror of the reference's Lua API (aukit.pcm, aukit.stream.*, Audio, aukit.effects)
  shard.py         one-process-per-GPU sharding of a batch by stream index (no data-path collective)
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
