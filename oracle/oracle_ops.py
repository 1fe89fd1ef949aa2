"""oracle_ops.py — CPU ORACLE (numpy) for the callers either side of the hot path.  TEST INFRASTRUCTURE ONLY.

Restates, index for index, the structural Audio methods, the generators and the packing of MCJack123/AUKit's aukit.lua
(cited as aukit.lua:LINE).  Like oracle/ork.h this is a checker: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; aukit_amd/ never does.

An audio is (list of 1-D float64 arrays — one per channel, Lua index i at [i-1] —, sample_rate).

PARITY PINNING: the reference ships no tests or vectors and cannot run here (see ork.h); these functions are pinned by the
hand-computed micro-vectors in tests/test_oracle_ops.py.  `pack` additionally depends on what the host VM's string.pack does
with a non-integer (not part of aukit.lua): the three plausible behaviours are modelled, none is pinned — **parity unpinned**.
"""
import math

import numpy as np


class LuaError(RuntimeError):
    pass


def concat(audios):  # Audio:concat  aukit.lua:695-718 (rates already equal)
    rate = audios[0][1]
    l = [len(a[0][0]) for a in audios]             # l[i] = #audios[i].data[1]
    cn = max(len(a[0]) for a in audios)
    out = []
    for c in range(cn):
        parts = []
        for a, n in zip(audios, l):
            parts.append(np.asarray(a[0][c][:n], dtype=np.float64) if c < len(a[0]) else np.zeros(n))
        out.append(np.concatenate(parts) if parts else np.zeros(0))
    return out, rate


def sub(audio, start=None, last=None):  # Audio:sub  aukit.lua:725-743
    data, rate = audio
    start = math.floor(start if start is not None else 0)
    last = math.floor(last if last is not None else 0)
    n = len(data[0])
    length = n / rate
    if start < 0:
        start = length + start
    if last <= 0:
        last = length + last
    for v in (start, last):
        if not (0 <= v <= length):
            raise LuaError("number outside of range")
    start, last = start * rate + 1, last * rate + 1
    out = []
    for sch in data:
        ch = []
        i = start
        while i <= last:                            # for i = start, last do ch[i-start+1] = sch[i] end
            if i != math.floor(i) or not (1 <= i <= len(sch)):
                break                               # sch[i] == nil: the table ends here
            ch.append(sch[int(i) - 1])
            i += 1
        out.append(np.asarray(ch, dtype=np.float64))
    return out, rate


def combine(audios):  # Audio:combine  aukit.lua:751-770
    rate = audios[0][1]
    n = max(len(a[0][0]) for a in audios)
    out = []
    for a in audios:
        for sch in a[0]:
            ch = np.zeros(n)
            m = min(n, len(sch))
            ch[:m] = sch[:m]                        # sch[i] or 0
            out.append(ch)
    return out, rate


def split(audio, *lists):  # Audio:split  aukit.lua:781-797
    data, rate = audio
    res = []
    for n, cl in enumerate(lists, 1):
        if len(cl) == 0:
            raise LuaError("bad argument #%d (cannot use empty table)" % n)
        chans = []
        for cs in cl:
            if not (1 <= cs <= len(data)):
                raise LuaError("channel %d (in argument %d) out of range" % (cs, n))
            chans.append(np.array(data[cs - 1], dtype=np.float64))
        res.append((chans, rate))
    return res


def rep(audio, count):  # Audio:rep  aukit.lua:839-852
    data, rate = audio
    reps = 0
    n = 0
    while n <= count - 1:                           # for n = 0, count - 1
        reps += 1
        n += 1
    return [np.tile(np.asarray(ch, dtype=np.float64), reps) for ch in data], rate


def reverse(audio):  # Audio:reverse  aukit.lua:856-866
    data, rate = audio
    return [np.asarray(ch, dtype=np.float64)[::-1].copy() for ch in data], rate


def _count(duration, rate):
    c = duration * rate                             # for i = 1, duration * sampleRate
    return int(math.floor(c)) if c >= 1 else 0


def new(duration, channels=1, sample_rate=48000):  # aukit.new  aukit.lua:1783-1796
    return [np.zeros(_count(duration, sample_rate)) for _ in range(channels)], sample_rate


def tone(frequency, duration, amplitude=1, wave="sine", duty=0.5, channels=1, sample_rate=48000):  # aukit.tone  aukit.lua:1808-1832, wavegen :286-299
    n = _count(duration, sample_rate)
    x = np.arange(1, n + 1, dtype=np.float64) / sample_rate
    if wave == "sine":
        v = np.sin(2 * x * math.pi * frequency) * amplitude
    elif wave == "triangle":
        v = 2.0 * np.abs(amplitude * np.fmod(2.0 * x * frequency + 1.5, 2.0) - amplitude) - amplitude
    elif wave == "sawtooth":
        v = amplitude * np.fmod(2.0 * x * frequency + 1.0, 2.0) - amplitude
    elif wave == "square":
        t = x * frequency
        v = np.where(t - np.floor(t) >= duty, -amplitude, amplitude).astype(np.float64)
    else:
        raise LuaError("bad argument #4 (invalid wave type)")
    return [v.copy() for _ in range(channels)], sample_rate


TRUNC, FLOOR, STRICT = 0, 1, 2


def pack(values, bit_depth=8, data_type="signed", big_endian=False, int_mode=TRUNC):
    """aukit.pack(data, bitDepth, dataType, bigEndian)  aukit.lua:1861-1878 on the numbers Audio:pcm returns (:868-910)."""
    v = np.asarray(values, dtype=np.float64)
    nb = bit_depth // 8
    if data_type == "float":
        raw = v.astype(np.float32).view(np.uint8).reshape(-1, 4)
        return bytes((raw[:, ::-1] if big_endian else raw).ravel())
    if int_mode == FLOOR:
        r = np.floor(v)
    elif int_mode == TRUNC:
        r = np.trunc(v)
    else:
        if np.any(v != np.floor(v)):
            raise LuaError("number has no integer representation")
        r = v
    q = r.astype(np.int64).view(np.uint64)
    raw = np.stack([((q >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8) for b in range(nb)], 1)
    return bytes((raw[:, ::-1] if big_endian else raw).ravel())


def encode_pcm(audio, bit_depth=8, data_type="signed", interleaved=True):  # Audio:pcm → encodePCM  aukit.lua:868-910
    data, _ = audio
    maxv = 2.0 ** (bit_depth - 1)
    add = maxv if data_type == "unsigned" else 0.0
    chans = []
    for ch in data:
        d = np.asarray(ch, dtype=np.float64)
        chans.append(d if data_type == "float" else d * np.where(d < 0, maxv, maxv - 1) + add)
    if interleaved:
        return np.stack(chans, 1).ravel()           # data[(n-1)*nc+c]
    return np.concatenate(chans)                    # data[(c-1)*len+n]
