"""writes the encoder inputs of the experiments (int8 files) into the current directory, with the oracle"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from oracle import oracle as O
O.build()
frames = 480000
q = lambda a: np.floor(np.asarray(a.data[0]) * np.where(np.asarray(a.data[0]) < 0, 128, 127)).astype(np.int8)
for i, d in enumerate(bench._cpu_pcm16(frames, 2, 48000, 4000, 4)):
    q(O.mono(O.dfpwm(O.audio_dfpwm(O.pcm(d, 16, O.SIGNED, 2, 48000), True), 2, 48000))).tofile(f"mono{i}.i8")
    if i == 0:
        a = np.frombuffer(d, dtype=np.int16).copy().reshape(frames, 2)
        a[:96000] = 0; a[240000:288000] = 0
        q(O.mono(O.dfpwm(O.audio_dfpwm(O.pcm(a.tobytes(), 16, O.SIGNED, 2, 48000), True), 2, 48000))).tofile("mono_gated.i8")
rb = np.random.default_rng(5).integers(0, 256, 120000, dtype=np.uint8).tobytes()
q(O.mono(O.dfpwm(rb, 2, 48000))).tofile("mono_rand.i8")
t = np.arange(frames) / 48000
np.floor((0.4 * np.sin(2 * np.pi * 220 * t) + 0.3 * np.sin(2 * np.pi * 1333 * t + 1) + 0.2 * np.sin(2 * np.pi * 5000 * t)) * 127).astype(np.int8).tofile("mono_sines.i8")
np.zeros(frames, dtype=np.int8).tofile("mono_zero.i8")
