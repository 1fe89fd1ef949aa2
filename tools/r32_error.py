#!/usr/bin/env python3
"""Error of the fused resample + low-pass (k_rs_onepole) against the fp64 oracle with the recurrence in f32 (default where the slope is <= 1/2) and in
fp64 (AUKIT_RS_F64=1): config 3b's shape (IMA-in-WAV 22.05 kHz -> resample(48000, cubic) -> effects.lowpass(11025)), F32 storage.  GPU box."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from aukit_amd import _native as N, batch as B
    from oracle import oracle as O
    from tests.util import pcm16
    ctx = B.Context(0, dtype=N.F32)
    s = O.gen_ima(pcm16(1016 * 300, 22050, 3, 4), 1, 512, 15)
    bt = B.Batch.upload(ctx, [s])
    res = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_ADPCM_WAV, 1, 22050, block_align=512), 48000, "cubic", dtype=N.F32)
    B.effect(ctx, res, "lowpass", 11025.0)
    k = ctx.last_kernel()[0]
    got = res.download()[0][0].astype(np.float64)
    ref = O.fx_lowpass(O.resample(O.wav_adpcm(s, 512, 1, 22050), 48000, O.CUBIC), 11025.0).data[0]
    e = got - ref
    print(f"{sys.argv[1]:5s} {k:28s} rms {np.sqrt(np.mean(e * e)):.3e}  max {np.max(np.abs(e)):.3e}  ({len(ref)} outputs)")
else:
    for tag, env in (("f32", {}), ("fp64", {"AUKIT_RS_F64": "1"})):
        subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env})
