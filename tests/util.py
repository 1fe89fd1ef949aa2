"""Shared synthetic-signal helpers for the tests (seed = 0xA0C17 + 1000*config + stream)."""
import numpy as np


def seed_for(config, stream):
    return 0xA0C17 + 1000 * config + stream


def signal(n, rate, config=1, stream=0, amp=0.5, noise=0.25):
    """0.5*sine(440 Hz) + uniform noise ±0.25, float64 in [-0.75, 0.75]."""
    rng = np.random.Generator(np.random.PCG64(seed_for(config, stream)))
    t = np.arange(n) / rate
    return amp * np.sin(2 * np.pi * 440 * t) + rng.uniform(-noise, noise, n)


def pcm16(n, rate, config=1, stream=0):
    return np.round(signal(n, rate, config, stream) * 32767).astype(np.int16)


def rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.sqrt(np.mean((a - b) ** 2))) if a.size else 0.0


def tail_kernel(which, rs_kernel, periodic_shape, mono=False):
    """the name `effects.<which>` leaves in ctx.last_kernel() when it pays an owed resample in its own pass (conftest.rs_kernel)"""
    k = "k_rsp" if (rs_kernel == "default" and periodic_shape) else "k_rs_onepole"
    return k + "<" + which + (",mono" if mono else "") + ">"
