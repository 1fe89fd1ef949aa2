#!/usr/bin/env python3
"""Generates tests/golden/*.npz — small input/output vectors produced by the CPU oracle (oracle/).

The reference has no fixtures and cannot run here (pure Lua for ComputerCraft, no interpreter in the image), so these
vectors pin the ORACLE's current behaviour (and, through tests/test_golden.py, the HIP path) against regressions;
the oracle itself is pinned by the independent known-answer tests in tests/test_oracle_kat.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.util import pcm16, signal  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)


def u8(b):
    return np.frombuffer(b, dtype=np.uint8)


def main():
    # config 1 / T shape: s16le 44.1 kHz mono → stream.pcm linear and Audio path cubic
    x = pcm16(6000, 44100, 1, 0).tobytes()
    s = O.stream_pcm(x, 16, O.SIGNED, 1, 44100, False, False, O.LINEAR)
    r = O.resample(O.pcm(x, 16, O.SIGNED, 1, 44100), 48000, O.CUBIC)
    save("pcm16_44k", data=u8(x), stream_linear=s.data[0], stream_chunk_len=s.chunk_len, resample_cubic=r.data[0])
    # config 2: µ-law 8 kHz
    g = O.gen_g711(pcm16(3000, 8000, 2, 0), True)
    save("g711_ulaw_8k", data=u8(g), audio_cubic=O.resample(O.g711(g, True, 1, 8000), 48000, O.CUBIC).data[0],
         stream_cubic=O.stream_g711(g, True, 1, 8000, False, O.CUBIC).data[0])
    # config 3: IMA 22.05 kHz mono, 4 blocks of 512 B
    im = O.gen_ima(pcm16(1016 * 4, 22050, 3, 0), 1, 512, 88)
    sa = O.stream_adpcm(im, 512, 1, 22050, False, O.CUBIC)
    save("ima_22k", data=u8(im), stream_cubic=sa.data[0], stream_chunk_len=sa.chunk_len,
         wav_lowpass=O.fx_lowpass(O.resample(O.wav_adpcm(im, 512, 1, 22050), 48000, O.CUBIC), 11025.0).data[0])
    # config 4: DFPWM stereo → mono → DFPWM
    l, rr = np.round(signal(52000, 48000, 4, 0) * 100), np.round(signal(52000, 48000, 4, 1) * 90)
    d = O.dfpwm_encode(np.stack([l, rr], 1).ravel())
    a = O.dfpwm(d, 2, 48000)
    save("dfpwm_stereo", data=u8(d), decoded_l=a.data[0][:4000], decoded_r=a.data[1][:4000], reencoded=u8(O.audio_dfpwm(O.mono(a), True)))
    # config 5: FLAC stereo → cubic → highpass → normalize → mono
    st = np.stack([pcm16(9000, 44100, 5, 0), pcm16(9000, 44100, 5, 1)], 1).astype(np.int32)
    fl = O.gen_flac(st.ravel(), 2, 16, 44100, 4096)
    fa = O.flac(fl)
    tail = O.mono(O.fx_normalize(O.fx_highpass(O.resample(fa, 48000, O.CUBIC), 20.0), 0.8))
    save("flac_stereo", data=u8(fl), pcm=st, decoded_l=fa.data[0], tail=tail.data[0])
    # MS-ADPCM and QOA
    ms = O.gen_msadpcm(np.stack([pcm16(1012 * 3, 44100, 6, 0), pcm16(1012 * 3, 44100, 6, 1)], 1).ravel(), 2, 1024)
    save("msadpcm_stereo", data=u8(ms), decoded_l=O.msadpcm(ms, 1024, 2, 44100).data[0], stream_cubic_mono=O.stream_msadpcm(ms, 1024, 2, 44100, True, None, O.CUBIC).data[0])
    # callers either side of the path: structural methods, tone, packing — from oracle/oracle_ops.py
    from oracle import oracle_ops as OPS
    xa = ([signal(300, 4800, 9, 0), signal(300, 4800, 9, 1)], 4800)
    xb = ([signal(120, 4800, 9, 2)], 4800)
    enc = OPS.encode_pcm(xa, 16, "signed", True)
    save("ops", a0=xa[0][0], a1=xa[0][1], b0=xb[0][0],
         concat0=OPS.concat([xa, xb])[0][0], concat1=OPS.concat([xa, xb])[0][1], combine2=OPS.combine([xa, xb])[0][2],
         rep_rev=OPS.reverse(OPS.rep(xb, 2))[0][0], tone_tri=OPS.tone(441, 0.01, 0.7, "triangle", 0.5, 1, 48000)[0][0],
         pack16_trunc=u8(OPS.pack(enc, 16, "signed", False, OPS.TRUNC)), pack16_floor_be=u8(OPS.pack(enc, 16, "signed", True, OPS.FLOOR)))
    q = O.gen_qoa(pcm16(5120 + 600, 44100, 8, 0), 1, 44100) + b"\0" * 8
    save("qoa_mono", data=u8(q), decoded=O.qoa(q).data[0], stream_cubic=O.stream_qoa(q[:-8], False, O.CUBIC).data[0])
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
