import numpy as np, sys
sys.path.insert(0, '.')
from tests.test_gpu_flac_edge import _odd_stream
from oracle import oracle as O
from aukit_amd import batch as B, _native as N
ctx = B.Context()
for seed in range(3):
    s, t = _odd_stream(seed)
    got = B.decode(ctx, B.Batch.upload(ctx, [s]), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()[0][0]
    ref = O.flac(s).data[0]
    bad = np.nonzero(got != ref)[0]
    print(seed, len(got), len(ref), "first bad", bad[:12], "n bad", len(bad))
    if len(bad):
        i = bad[0]
        print("  got", got[i:i+4] * 65536, "ref", ref[i:i+4] * 65536)
