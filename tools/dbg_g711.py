import numpy as np, sys
sys.path.insert(0, '.')
from tests.util import pcm16
from oracle import oracle as O
from aukit_amd import batch as B, _native as N
ctx = B.Context()
rate, alaw, interp = 22050, True, "linear"
s = O.gen_g711(pcm16(int(rate * 2.3), rate, 2, 5), not alaw)
bt = B.Batch.upload(ctx, [s])
desc = B.make_desc(N.CODEC_G711, 1, rate, ulaw=not alaw)
ref = O.stream_g711(s, not alaw, 1, rate, False, O.INTERP[interp]).data[0]
for dt in (N.I8, N.F64):
    out, ck = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
    g = out.download()[0][0]
    bad = np.nonzero(g != ref)[0]
    print("dtype", dt, ctx.last_kernel()[0], "n bad", len(bad), bad[:10], g[bad[:5]], ref[bad[:5]])
    ctx.set_option(N.OPT_EXACT_MATH, 1)
    out2, _ = B.stream_decode(ctx, bt, desc, interp, dtype=dt)
    g2 = out2.download()[0][0]
    ctx.set_option(N.OPT_EXACT_MATH, 0)
    bad2 = np.nonzero(g2 != ref)[0]
    print("   exact kernel", ctx.last_kernel()[0], "n bad", len(bad2), bad2[:10])
