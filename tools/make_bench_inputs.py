#!/usr/bin/env python3
"""Encoder-made inputs of bench.py's codec workloads, cached as fixtures under bench_data/ so that the bench does not need the CPU oracle
(the checker) to run: the reference has no encoders, the product has none for IMA / MS-ADPCM / QOA / FLAC, so the oracle's generators
(oracle/ork_gen.c) make them ONCE, here.  SURVEY §8d signal: 0.5 sine(440 Hz) + uniform noise ±0.25, seeds 0xA0C17 + 1000 config + stream.
    python tools/make_bench_inputs.py        (≈ 4 MB of .bin files; deterministic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data")
os.makedirs(OUT, exist_ok=True)


def sig(n, rate, seed, f=440.0, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / rate
    return np.round((0.5 * np.sin(2 * np.pi * f * t) + rng.uniform(-0.25, 0.25, n)) * 32767 * scale).astype(np.int16)


def put(name, data):
    with open(os.path.join(OUT, name), "wb") as fh:
        fh.write(data)
    print(f"{name}: {len(data)} bytes")


O.build()
# config 3a: 220 blocks of 512 B, IMA mono 22 050 Hz (the AUKit-variant encoder), 4 distinct streams
for i in range(4):
    put(f"ima_22050_220x512_{i}.bin", O.gen_ima(sig(1016 * 220, 22050, 0xA0C17 + 3000 + i), 1, 512, 88))
# config 5: FLAC streams, 44.1 kHz stereo 16-bit, 10 s, blocks of 4096.  SIXTEEN distinct ones since round 6 (the bench ran 2048 copies
# of one file): stream i has its own seed (0xA0C17 + 5000 + i), its own pair of tones, and the generator's `salt` i — which frame
# gets which subframe type, predictor order, Rice partition order and stereo mode differs from file to file.  _0 is the round-5 fixture.
n = 441000
for i in range(16):
    fl, fr = 440.0 * (1 + 0.07 * i), 330.0 * (1 + 0.05 * i)
    ch = [sig(n, 44100, 0xA0C17 + 5000 + i, f, 0.9 - 0.03 * (i % 5)).astype(np.int32) for f in (fl, fr)]
    put(f"flac_44100_stereo_10s_{i}.bin", O.gen_flac(np.stack(ch, 1).ravel(), 2, 16, 44100, 4096, salt=i))
# MS-ADPCM mono 44.1 kHz, blocks of 1024 B (2036 samples), 216 blocks ≈ 10 s, 2 distinct streams
for i in range(2):
    put(f"msadpcm_44100_mono_216x1024_{i}.bin", O.gen_msadpcm(sig(2036 * 216, 44100, 0xA0C17 + 6000 + i), 1, 1024))
# QOA stereo 44.1 kHz 10 s, 2 distinct streams (8 trailing bytes: Q18)
for i in range(2):
    st = np.stack([sig(n, 44100, 0xA0C17 + 7000 + 2 * i + c, 440.0 if c == 0 else 330.0) for c in range(2)], 1).ravel()
    put(f"qoa_44100_stereo_10s_{i}.bin", O.gen_qoa(st, 2, 44100) + b"\0" * 8)
