"""Host-side mirror of the reference's Lua module table (`require "aukit"`, aukit.lua:97-113) over the HIP C ABI.

Same names, argument order, defaults and error strings as the reference so that tests read like its own API:

    import aukit_amd.aukit as aukit
    aukit.defaultInterpolation = "cubic"
    audio = aukit.pcm(data, 16, "signed", 1, 44100):  ->  aukit.pcm(data, 16, "signed", 1, 44100)
    audio.resample(48000).mono()                           (Lua `:` calls become Python methods)
    aukit.effects.normalize(audio, 0.8)                    (mutates and returns the same object, aukit.lua:3356-3618)
    for chunk, pos in aukit.stream.wav(data, mono):        (iterator of (chunk tables, position), aukit.lua:2927)

Every per-sample loop runs on the GPU; this file only parses container headers (host-side bytes),
checks arguments the way cc.expect does, and moves handles around.  Lua `data` strings are Python `bytes`.
Batch use (N streams per call) goes through `aukit_amd.batch` directly.
"""
import math
import os
import re
import struct

import numpy as np

from . import _native as N
from . import batch as B

_VERSION = "1.10.0"
defaultInterpolation = "linear"  # aukit.lua:99

_ctx = None


def context():
    """The process-wide GPU context (created on first use; raises without an MI355X — there is no CPU path)."""
    global _ctx
    if _ctx is None:
        _ctx = B.Context(0, dtype=N.F64)
    return _ctx


class LuaError(RuntimeError):
    """A Lua `error(...)` the reference would raise (message text preserved)."""


def _wrap(fn, *a, **k):
    try:
        return fn(*a, **k)
    except N.AukitError as e:
        raise LuaError(e.msg) from None


def _expect(n, v, *types):
    names = {"string": (bytes, bytearray, memoryview), "number": (int, float), "boolean": (bool,), "table": (list, tuple), "nil": (type(None),)}
    for t in types:
        if t == "number" and isinstance(v, bool):
            continue
        if isinstance(v, names[t]):
            return v
    got = "nil" if v is None else ("boolean" if isinstance(v, bool) else ("number" if isinstance(v, (int, float)) else ("string" if isinstance(v, (bytes, str)) else type(v).__name__)))
    want = " or ".join(types) if len(types) < 3 else ", ".join(types[:-1]) + ", or " + types[-1]
    raise LuaError(f"bad argument #{n} (expected {want}, got {got})")


def _expect_src(n, v):
    """aukit.stream.*: `string | function` (a reader returning strings, then nil)"""
    if callable(v):
        return v
    return _expect(n, v, "string", "function")


def _interp(name, argn=2):
    if name not in N.INTERP:
        raise LuaError(f"bad argument #{argn} (invalid interpolation type)")
    return N.INTERP[name]


class Audio:
    """aukit.Audio (aukit.lua:116-123, methods :631-1024) backed by a device-resident audio batch of one stream."""

    def __init__(self, handle, metadata=None, info=None):
        self._h = handle  # batch.AudioBatch with n == 1
        self.metadata = dict(metadata or {})
        self.info = dict(info or {})

    # -- data access (what `audio.data[c][i]` / `audio.sampleRate` are in Lua)
    @property
    def sampleRate(self):
        return self._h.info()["sample_rate"]

    @property
    def data(self):
        return self._h.download()[0]

    def len(self):  # Audio:len  :638
        lens, _, _ = self._h.layout()
        return int(lens[0]) / self.sampleRate

    def channels(self):  # Audio:channels  :644
        return self._h.info()["channels"]

    @classmethod
    def from_arrays(cls, channels, sample_rate):
        """Build an Audio from host arrays (what Lua code does by filling `data` tables by hand)."""
        return cls(B.AudioBatch.upload(context(), [[np.asarray(c, dtype=np.float64) for c in channels]], sample_rate, dtype=N.F64))

    # -- methods on the hot path
    def resample(self, sampleRate, interpolation=None):  # :653
        _expect(1, sampleRate, "number")
        interpolation = interpolation if interpolation is not None else defaultInterpolation
        ip = _interp(interpolation)
        return Audio(_wrap(B.resample, context(), self._h, float(sampleRate), ip), self.metadata, self.info)

    def mono(self):  # :677
        return Audio(_wrap(B.mono, context(), self._h), self.metadata, self.info)

    def mix(self, amplifier, *others):  # :804
        audios = [self] + list(others)
        if not isinstance(amplifier, (int, float)):
            if not isinstance(amplifier, Audio):
                raise LuaError("bad argument #1 (expected Audio, got " + type(amplifier).__name__ + ")")
            audios.insert(1, amplifier)
            amplifier = 1
        hs = []
        for a in audios:
            if not isinstance(a, Audio):
                raise LuaError("bad argument (expected Audio)")
            if a.sampleRate != self.sampleRate:  # :810 resamples with the default interpolation
                a = a.resample(self.sampleRate)
            hs.append(a._h)
        return Audio(_wrap(B.mix, context(), hs, float(amplifier)), self.metadata, self.info)

    def pcm(self, bitDepth=None, dataType=None, interleaved=None):  # :901 → table of numbers (unfloored, like the Lua)
        bitDepth = 8 if bitDepth is None else bitDepth
        dataType = "signed" if dataType is None else dataType
        interleaved = True if interleaved is None else interleaved
        if bitDepth not in (8, 16, 24, 32):
            raise LuaError("bad argument #2 (invalid bit depth)")
        if dataType not in ("signed", "unsigned", "float"):
            raise LuaError("bad argument #3 (invalid data type)")
        if dataType == "float" and bitDepth != 32:
            raise LuaError("bad argument #2 (float audio must have 32-bit depth)")
        return _wrap(B.encode_pcm, context(), self._h, bitDepth, dataType, interleaved).download()[0][0]

    def dfpwm(self, interleaved=None):  # :1005
        interleaved = True if interleaved is None else interleaved
        return _wrap(B.dfpwm_encode, context(), self._h, interleaved).download()[0]

    def stream(self, chunkSize=None, bitDepth=None, dataType=None):  # :921 → iterator of (chunk tables, seconds)
        chunkSize = 131072 if chunkSize is None else int(chunkSize)
        bitDepth = 8 if bitDepth is None else bitDepth
        dataType = "signed" if dataType is None else dataType
        planar = self.pcm(bitDepth, dataType, False)
        nc = self.channels()
        per = len(planar) // max(nc, 1)
        rate = self.sampleRate

        def it():
            pos = 1
            while pos <= per:
                yield [planar[c * per + pos - 1: c * per + pos - 1 + chunkSize] for c in range(nc)], pos / rate
                pos += chunkSize
        return it(), per / rate

    # -- structural methods (either side of the hot path): device-side row copies
    def _others(self, others, first_arg=1):
        hs = [self._h]
        for i, a in enumerate(others):
            if not isinstance(a, Audio):
                raise LuaError(f"bad argument #{first_arg + i} (expected Audio, got {type(a).__name__})")
            if a.sampleRate != self.sampleRate:  # :702 / :756 resample with the default interpolation
                a = a.resample(self.sampleRate)
            hs.append(a._h)
        return hs

    def concat(self, *others):  # :695
        return Audio(_wrap(B.concat, context(), self._others(others)), self.metadata, self.info)

    def sub(self, start=None, last=None):  # :725
        _expect(1, start, "number", "nil")
        _expect(2, last, "number", "nil")
        return Audio(_wrap(B.sub, context(), self._h, float(start or 0), float(last or 0)), self.metadata, self.info)

    def combine(self, *others):  # :751
        return Audio(_wrap(B.combine, context(), self._others(others)), self.metadata, self.info)

    def split(self, *lists):  # :781 → one Audio per channel list
        res = []
        for n, cl in enumerate(lists, 1):
            _expect(n, cl, "table")
            if len(cl) == 0:
                raise LuaError(f"bad argument #{n} (cannot use empty table)")
            for cs in cl:
                if not (1 <= cs <= self.channels()):
                    raise LuaError(f"channel {cs} (in argument {n}) out of range")
            res.append(Audio(_wrap(B.split, context(), self._h, list(cl)), self.metadata, self.info))
        return tuple(res)

    def rep(self, count):  # :839
        _expect(1, count, "number")
        return Audio(_wrap(B.rep, context(), self._h, float(count)), self.metadata, self.info)

    def reverse(self):  # :856
        return Audio(_wrap(B.reverse, context(), self._h), self.metadata, self.info)

    def _wav_list(self):
        """the "LIST" chunk Audio:wav writes for self.metadata (:946-956, :980-990): str_pack("!2<c4" .. ("c4s4Xh"):rep(k), "INFO", tag, tostring(value), ...)
        — every entry its four-letter tag, the value's length as a little-endian u32, the value, padded to an even offset — then "LIST" .. u32(#list) .. list.
        The reference walks `pairs(self.metadata)` (and `pairs(wavMetadata)` for the tag: trackNumber has two, IPRT and ITRK), whose order a Lua VM
        does not define: here the metadata's own order, and the first tag in the reference's table.  Keys without a tag are left out, as there."""
        if not self.metadata:
            return b""
        entries = b""
        for k, v in self.metadata.items():
            tag = _WAV_TAG.get(k)
            if tag is None:
                continue
            val = _lua_tostring(v).encode("latin-1", "replace") if not isinstance(v, (bytes, bytearray)) else bytes(v)
            entries += tag + struct.pack("<I", len(val)) + val
            if len(entries) % 2:
                entries += b"\0"   # Xh: align to 2 (the chunk starts at an even offset: "INFO" is four bytes)
        lst = b"INFO" + entries
        return b"LIST" + struct.pack("<I", len(lst)) + lst

    def wav(self, bitDepth=None, int_mode=N.PACK_TRUNC):  # :940-997
        bitDepth = 16 if bitDepth is None else _expect(1, bitDepth, "number")
        nc, rate, n = self.channels(), self.sampleRate, int(self._h.layout()[0][0])
        lst = self._wav_list()   # (the RIFF size field does not count it: `#str + 72` / `#str + 36` with or without, as the reference writes it)
        if bitDepth == 1:  # DFPWM in WAVE_FORMAT_EXTENSIBLE  :952-961, :981-985
            body = self.dfpwm(True)
            guid = bytes([0x3A, 0xC1, 0xFA, 0x38, 0x81, 0x1D, 0x43, 0x61, 0xA4, 0x0D, 0xCE, 0x53, 0xCA, 0x60, 0x7C, 0xD1])  # wavExtensible.dfpwm  :137
            chmask = {1: 0x04, 2: 0x03, 3: 0x07, 4: 0x33, 5: 0x37, 6: 0x3F, 7: 0x637, 8: 0x63F}.get(nc, 0)  # wavExtensibleChannels  :141-149
            return (struct.pack("<4sI4s4sIHHIIHHHHI16s4sII", b"RIFF", len(body) + 72, b"WAVE", b"fmt ", 40, 0xFFFE, nc, int(rate), int(rate * nc / 8),
                                int(math.ceil(nc / 8)), 1, 22, 1, chmask, guid, b"fact", 4, n) + lst + struct.pack("<4sI", b"data", len(body)) + body)
        if bitDepth not in (8, 16, 24, 32):
            raise LuaError("bad argument #2 (invalid bit depth)")
        body = _wrap(B.pack_pcm, context(), self._h, bitDepth, "unsigned" if bitDepth == 8 else "signed", False, True, int_mode).download()[0]
        return struct.pack("<4sI4s4sIHHIIHH", b"RIFF", len(body) + 36, b"WAVE", b"fmt ", 16, 1, nc, int(rate), int(rate * nc * bitDepth / 8),
                           int(nc * bitDepth / 8), bitDepth) + lst + struct.pack("<4sI", b"data", len(body)) + body

    def __str__(self):
        return f"Audio: {self.sampleRate} Hz, {self.channels()} channels, {self.len()} seconds"


# wavMetadata (aukit.lua:198-216), inverted: metadata key -> the tag Audio:wav writes (the first of the reference's table where a key has two)
_WAV_TAG = {"album": b"IPRD", "title": b"INAM", "artist": b"IART", "author": b"IWRI", "composer": b"IMUS", "producer": b"IPRO", "trackNumber": b"IPRT",
            "trackCount": b"IFRM", "partNumber": b"PRT1", "partCount": b"PRT2", "length": b"TLEN", "rating": b"IRTD", "date": b"ICRD", "encodedBy": b"ITCH",
            "encoder": b"ISFT", "media": b"ISRF", "genre": b"IGNR", "comment": b"ICMT", "copyright": b"ICOP", "language": b"ILNG"}


def _lua_tostring(v):
    """tostring(v) for the values metadata holds: strings as they are, integral numbers without a fraction (CC: Tweaked's VM prints 5, not 5.0),
    other numbers as %.14g"""
    if isinstance(v, str):
        return v
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return str(int(v)) if float(v).is_integer() else "%.14g" % v
    return str(v)


def _load(desc, data, dtype=None):
    ctx = context()
    bt = _wrap(B.Batch.upload, ctx, [data])
    return _wrap(B.decode, ctx, bt, desc, ctx.dtype if dtype is None else dtype)


# ---------------------------------------------------------------- generators and packing  (aukit.lua:1779-1878)
def new(duration, channels=None, sampleRate=None):  # :1783
    _expect(1, duration, "number")
    channels = 1 if channels is None else _expect(2, channels, "number")
    sampleRate = 48000 if sampleRate is None else _expect(3, sampleRate, "number")
    return Audio(_wrap(B.tone, context(), 1, 0.0, float(duration), 1.0, "none", 0.5, int(channels), float(sampleRate), N.F64))


def tone(frequency, duration, amplitude=None, waveType=None, duty=None, channels=None, sampleRate=None):  # :1808
    _expect(1, frequency, "number")
    _expect(2, duration, "number")
    amplitude = 1 if amplitude is None else _expect(3, amplitude, "number")
    waveType = "sine" if waveType is None else waveType
    duty = 0.5 if duty is None else _expect(5, duty, "number")
    channels = 1 if channels is None else _expect(6, channels, "number")
    sampleRate = 48000 if sampleRate is None else _expect(7, sampleRate, "number")
    if waveType not in ("sine", "triangle", "sawtooth", "square"):
        raise LuaError("bad argument #4 (invalid wave type)")
    return Audio(_wrap(B.tone, context(), 1, float(frequency), float(duration), float(amplitude), waveType, float(duty), int(channels), float(sampleRate), N.F64))


_noise_calls = [0]


def noise(duration, amplitude=None, channels=None, sampleRate=None, seed=None):  # :1840
    """White noise, (random() * 2 - 1) * amplitude per sample.  The reference draws from its VM's math.random — not reproducible by anyone;
    here the device draws (Philox4x32-10, aukit_noise): `seed` names the audio (the same seed, the same samples); without one every call
    draws a fresh audio, as the reference's calls do."""
    _expect(1, duration, "number")
    amplitude = 1 if amplitude is None else _expect(2, amplitude, "number")
    channels = 1 if channels is None else _expect(3, channels, "number")
    sampleRate = 48000 if sampleRate is None else _expect(4, sampleRate, "number")
    if seed is None:
        _noise_calls[0] += 1
        seed = (int.from_bytes(os.urandom(8), "little") ^ _noise_calls[0])
    return Audio(_wrap(B.noise, context(), 1, float(duration), float(amplitude), int(channels), float(sampleRate), int(seed), N.F64))


def pack(data, bitDepth=None, dataType=None, bigEndian=None, int_mode=N.PACK_TRUNC):  # :1861 on a table of numbers
    _expect(1, data, "string", "table")
    bitDepth = 8 if bitDepth is None else bitDepth
    dataType = "signed" if dataType is None else dataType
    if bitDepth not in (8, 16, 24, 32):
        raise LuaError("bad argument #2 (invalid bit depth)")
    if dataType not in ("signed", "unsigned", "float"):
        raise LuaError("bad argument #3 (invalid data type)")
    if dataType == "float" and bitDepth != 32:
        raise LuaError("bad argument #2 (float audio must have 32-bit depth)")
    if dataType == "float":
        tmp = Audio.from_arrays([np.asarray(data, dtype=np.float64)], 48000)
        return _wrap(B.pack_pcm, context(), tmp._h, 32, "float", bool(bigEndian), True, int_mode).download()[0]
    # integers: pack() receives already-encoded numbers, so undo nothing — convert and lay out the bytes (the device packer works on [-1, 1] audio)
    v = np.asarray(data, dtype=np.float64)
    if int_mode == N.PACK_STRICT and np.any(v != np.floor(v)):
        raise LuaError("bad argument #2 to 'pack' (number has no integer representation)")
    q = (np.floor(v) if int_mode == N.PACK_FLOOR else np.trunc(v)).astype(np.int64).view(np.uint64)
    nb = bitDepth // 8
    raw = np.stack([((q >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8) for b in range(nb)], 1)
    return bytes((raw[:, ::-1] if bigEndian else raw).ravel())


# ---------------------------------------------------------------- loaders  (aukit.lua:1049-1777)
def pcm(data, bitDepth=None, dataType=None, channels=None, sampleRate=None, interleaved=None, bigEndian=None):
    _expect(1, data, "string", "table")  # :1050 — a table of numbers is normalised as it is (:1077-1096)
    bitDepth = 8 if bitDepth is None else bitDepth
    dataType = "signed" if dataType is None else dataType
    channels = 1 if channels is None else channels
    sampleRate = 48000 if sampleRate is None else sampleRate
    interleaved = True if interleaved is None else interleaved
    if dataType not in ("signed", "unsigned", "float"):
        raise LuaError("bad argument #3 (invalid data type)")
    d = B.make_desc(N.CODEC_PCM, channels, sampleRate, bitDepth, dataType, bool(bigEndian), interleaved)
    if isinstance(data, (list, tuple)):
        return Audio(_wrap(B.decode_table, context(), [data], d), {}, {"bitDepth": bitDepth, "dataType": dataType})
    return Audio(_load(d, data), {}, {"bitDepth": bitDepth, "dataType": dataType})


def adpcm(data, channels=None, sampleRate=None, topFirst=None, interleaved=None, predictor=None, step_index=None):
    _expect(1, data, "string", "table")  # :1184 — a table holds one nibble per entry (:1232-1238)
    channels = 1 if channels is None else channels
    sampleRate = 48000 if sampleRate is None else sampleRate
    pred = [predictor] if isinstance(predictor, (int, float)) else predictor
    idx = [step_index] if isinstance(step_index, (int, float)) else step_index
    if pred is not None and channels > len(pred):
        raise LuaError("bad argument #6 (table too short)")
    if idx is not None and channels > len(idx):
        raise LuaError("bad argument #7 (table too short)")
    d = B.make_desc(N.CODEC_ADPCM, channels, sampleRate, top_first=True if topFirst is None else topFirst,
                    interleaved=True if interleaved is None else interleaved, predictor=pred, step_index=idx)
    if isinstance(data, (list, tuple)):
        for v in data:
            if not isinstance(v, (int, float)) or v != int(v) or not 0 <= v <= 15:
                raise LuaError("attempt to perform arithmetic on a nil value (field '?')")  # ima_index_table[nibble]
        return Audio(_wrap(B.decode_nibbles, context(), [[int(v) for v in data]], d), {}, {"bitDepth": 16, "dataType": "signed"})
    return Audio(_load(d, data), {}, {"bitDepth": 16, "dataType": "signed"})


def msadpcm(data, blockAlign, channels=None, sampleRate=None, coefficients=None):
    _expect(1, data, "string")
    _expect(2, blockAlign, "number")
    d = B.make_desc(N.CODEC_MSADPCM, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate, block_align=blockAlign, coefficients=coefficients)
    return Audio(_load(d, data), {}, {"bitDepth": 16, "dataType": "signed"})


def g711(data, ulaw, channels=None, sampleRate=None):
    _expect(1, data, "string")
    _expect(2, ulaw, "boolean")
    d = B.make_desc(N.CODEC_G711, 1 if channels is None else channels, 8000 if sampleRate is None else sampleRate, ulaw=ulaw)
    return Audio(_load(d, data), {"bitDepth": 14 if ulaw else 13, "dataType": "signed"}, {})


def dfpwm(data, channels=None, sampleRate=None):
    _expect(1, data, "string")
    d = B.make_desc(N.CODEC_DFPWM, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate)
    return Audio(_load(d, data), {}, {"bitDepth": 8, "dataType": "signed"})


def mdfpwm(data, head=None):
    _expect(1, data, "string")
    if bytes(data[:7]) != b"MDFPWM\x03":
        raise LuaError("bad argument #1 (not a MDFPWM file)")
    pos = 11
    meta = []
    for _ in range(3):
        ln = data[pos]
        meta.append(bytes(data[pos + 1:pos + 1 + ln]))
        pos += 1 + ln
    a = Audio(_load(B.make_desc(N.CODEC_MDFPWM), data), {}, {"bitDepth": 8, "dataType": "signed"})
    a.metadata = {"artist": meta[0], "title": meta[1], "album": meta[2]}
    return a


def qoa(data):
    _expect(1, data, "string")
    return Audio(_load(B.make_desc(N.CODEC_QOA), data), {}, {"bitDepth": 16, "dataType": "signed"})


def flac(data, head=None):
    _expect(1, data, "string")
    return Audio(_load(B.make_desc(N.CODEC_FLAC), data), {}, {"dataType": "signed"})


# ---- container front-ends: the header walk is the library's (aukit_parse_container, csrc/container.hip: host-side C, bound by the LuaJIT
# shim as well); this side only hands the payload range to the loaders (aukit.lua:1456-1651, :2927-3113) ----
def _parse(data, kind, stream=False):
    """→ (N.Container, payload bytes); raises the reference's errors (truncated chunk headers included)"""
    import ctypes as C
    buf = bytes(data)
    c = N.Container()
    rc = N.lib().aukit_parse_container(buf, C.c_uint64(len(buf)), kind, int(stream), C.byref(c))
    if rc:
        raise LuaError(N.lib().aukit_last_error().decode(errors="replace"))
    return c, buf[c.payload_off:c.payload_off + c.payload_len]


def _parse_first_piece(data, kind):
    """header walk on the first piece of a reader function (aukit_parse_container mode 2)"""
    return _parse(data, kind, stream=2)


def _desc_copy(c):
    d = N.CodecDesc()
    import ctypes as C
    C.memmove(C.byref(d), C.byref(c.desc), C.sizeof(d))
    return d


def wav(data, head=None):  # aukit.lua:1456
    _expect(1, data, "string")
    c, p = _parse(data, N.CONTAINER_WAV)
    a = Audio(_load(_desc_copy(c), p))
    a.metadata = {}
    a.info = {"dataType": N.WAVDT[c.wav_data_type], "bitDepth": c.bit_depth}
    return a


def aiff(data, head=None):  # aukit.lua:1580
    _expect(1, data, "string")
    c, p = _parse(data, N.CONTAINER_AIFF)
    d = _desc_copy(c)
    info = {"bitDepth": d.bit_depth, "dataType": ("signed", "unsigned", "float")[d.data_type]} if d.codec == N.CODEC_PCM else {"bitDepth": 8, "dataType": "signed"}  # what aukit.pcm / aukit.g711 set
    return Audio(_load(d, p), {}, info)


def au(data):  # aukit.lua:1639
    _expect(1, data, "string")
    c, p = _parse(data, N.CONTAINER_AU)
    d = _desc_copy(c)
    info = {"bitDepth": d.bit_depth, "dataType": ("signed", "unsigned", "float")[d.data_type]} if d.codec == N.CODEC_PCM else {"bitDepth": 8, "dataType": "signed"}
    return Audio(_load(d, p), {}, info)


# aukit.lua:2134-2146: (string.unpack format, bit depth, data type) in table order; every format reads 8 values
_DETECT_FMTS = (("<8b", 8, "signed"), ("<8B", 8, "unsigned"), ("<8h", 16, "signed"), ("<8i", 32, "signed"),
                ("<8f", 32, "float"), (None, 24, "signed"), ("<8I", 32, "unsigned"), (None, 24, "unsigned"),
                ("<8H", 16, "unsigned"))


def _detect_unpack(fmt, bits, typ, data, init):
    """pcall(string.unpack, fmt, data, init) → list of 8 numbers or None (Lua 5.3 lstrlib.c str_unpack position rules)."""
    n = len(data)
    if init < 0:
        init = 0 if -init > n else n + init + 1
    if init < 1 or init - 1 > n:  # "initial position out of string"
        return None
    nbytes = bits  # 8 values of bits/8 bytes
    raw = data[init - 1:init - 1 + nbytes]
    if len(raw) < nbytes:  # "data string too short"
        return None
    if fmt is not None:
        return list(struct.unpack(fmt, raw))
    return [int.from_bytes(raw[i:i + 3], "little", signed=(typ == "signed")) for i in range(0, 24, 3)]


def detect(data):  # aukit.lua:2156 → (type, bitDepth, dataType); host-side bytes only
    """Container magic, then the reference's "near silence at either end" PCM heuristic, then the DFPWM idle pattern."""
    _expect(1, data, "string")
    data = bytes(data)
    if re.match(rb"RIFF....WAVE", data, re.S):
        return "wav", None, None
    if re.match(rb"FORM....AIF[FC]", data, re.S):
        return "aiff", None, None
    if data[:4] == b".snd":
        return "au", None, None
    if data[:4] == b"fLaC":
        return "flac", None, None
    if data[:7] == b"MDFPWM\x03":
        return "mdfpwm", None, None
    if data[:4] == b"qoaf":
        return "qoa", None, None
    for fmt, bits, typ in _DETECT_FMTS:
        mid = 2.0 ** (bits - 1) if typ == "unsigned" else 0
        gap = 0.001 if typ == "float" else 8 * 2.0 ** (bits - 8)
        # :2172 reads from the start, :2183 from `#data - bitDepth` (the bit depth, not the byte count, is subtracted)
        for init in (1, len(data) - bits):
            nums = _detect_unpack(fmt, bits, typ, data, init)
            if nums is None:
                continue
            allzero, ok = True, True
            for v in nums:
                if v != mid:
                    allzero = False
                if v < mid - gap or v > mid + gap:
                    ok = False
                    break
            if ok and not allzero:
                return "pcm", bits, typ
    if b"\x55" * 12 in data or b"\xaa" * 12 in data:
        return "dfpwm", None, None
    return None, None, None


# ---------------------------------------------------------------- aukit.stream.*  (aukit.lua:2207-3337)
class _StreamNS:
    """Iterator factories: `it, length = aukit.stream.pcm(...)`; `for chunk, pos in it` (chunk = list of per-channel arrays)."""

    @staticmethod
    def _run_fn(desc, fn, first, mono, dtype, length_override=None):
        """reader-function input (aukit.lua:2776-2786 ...): `first` is what the first fn() call returned, the rest is pulled as chunks run out.
        The chunks are those of the string version for the concatenated input (the library's resumable handle)."""
        ctx = context()
        h = _wrap(B.StreamHandle, ctx, desc, _interp(defaultInterpolation, 0), bool(mono), dtype)
        _wrap(h.feed, first)

        def it():
            done = False
            while True:
                kind, chans, pos = _wrap(h.next)
                if kind == "chunk":
                    yield chans, pos
                elif kind == "end":
                    h.close()
                    return
                else:  # need input
                    piece = None if done else fn()
                    if piece is None:
                        done = True
                        _wrap(h.finish)
                    else:
                        _wrap(h.feed, piece)
        return it(), (_wrap(h.length) if length_override is None else length_override)

    @staticmethod
    def _run(desc, data, mono, dtype, length_override=None, endless_empty=False):
        if callable(data):
            first = data()
            _expect(1, first, "string")
            return _StreamNS._run_fn(desc, data, first, mono, dtype, length_override)
        ctx = context()
        bt = _wrap(B.Batch.upload, ctx, [data])
        out, ck = _wrap(B.stream_decode, ctx, bt, desc, _interp(defaultInterpolation, 0), bool(mono), dtype)
        return _StreamNS._chunk_iter(out, ck, length_override, endless_empty)

    @staticmethod
    def _chunk_iter(out, ck, length_override=None, endless_empty=False):
        chans = out.download()[0]
        n = int(ck.nchunks[0])
        lens, poss, status = ck.lens[0][:n], ck.pos[0][:n], int(ck.status[0])

        def it():
            off = 0
            for k in range(n):
                yield [c[off:off + int(lens[k])] for c in chans], float(poss[k])
                off += int(lens[k])
            if status == N.E_LUA:
                raise LuaError("the reference iterator raises a Lua error here (end of data inside the prefill / a malformed block)")
            while endless_empty:  # stream.g711 never returns nil with string input (Q13)
                yield [np.zeros(0) for _ in chans], float("nan")
        return it(), (float(ck.length_seconds[0]) if length_override is None else length_override)

    def pcm(self, data, bitDepth=None, dataType=None, channels=None, sampleRate=None, bigEndian=None, mono=None):
        if not isinstance(data, (list, tuple)):
            _expect_src(1, data)
        bitDepth = 8 if bitDepth is None else bitDepth
        dataType = "signed" if dataType is None else dataType
        if dataType not in ("signed", "unsigned", "float"):
            raise LuaError("bad argument #3 (invalid data type)")
        d = B.make_desc(N.CODEC_PCM, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate, bitDepth, dataType, bool(bigEndian))
        if isinstance(data, (list, tuple)):  # a table of numbers (:2255-2290): `read()` hands out data[pos], normalised like the string's samples
            ctx = context()
            out, ck = _wrap(B.stream_decode_table, ctx, [data], d, _interp(defaultInterpolation, 0), bool(mono), N.F64)
            return self._chunk_iter(out, ck)
        return self._run(d, data, mono, N.F64)

    def dfpwm(self, data, sampleRate=None, channels=None, mono=None):
        _expect_src(1, data)
        d = B.make_desc(N.CODEC_DFPWM, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate)
        return self._run(d, data, mono, N.F64)

    def mdfpwm(self, data, mono=None):
        _expect_src(1, data)
        if bytes(data[:7]) != b"MDFPWM\x03":
            raise LuaError("bad argument #1 (invalid MDFPWM data)")
        return self._run(B.make_desc(N.CODEC_MDFPWM), data, mono, N.I8)

    def msadpcm(self, input, blockAlign, channels=None, sampleRate=None, mono=None, coefficients=None):
        _expect_src(1, input)
        _expect(2, blockAlign, "number")
        d = B.make_desc(N.CODEC_MSADPCM, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate, block_align=blockAlign, coefficients=coefficients)
        return self._run(d, input, mono, N.I8)

    def adpcm(self, input, blockAlign, channels=None, sampleRate=None, mono=None):
        _expect_src(1, input)
        _expect(2, blockAlign, "number")
        d = B.make_desc(N.CODEC_ADPCM_WAV, 1 if channels is None else channels, 48000 if sampleRate is None else sampleRate, block_align=blockAlign)
        return self._run(d, input, mono, N.I8)

    def g711(self, input, ulaw, channels=None, sampleRate=None, mono=None):
        _expect_src(1, input)
        _expect(2, ulaw, "boolean")
        d = B.make_desc(N.CODEC_G711, 1 if channels is None else channels, 8000 if sampleRate is None else sampleRate, ulaw=ulaw)
        return self._run(d, input, mono, N.I8, endless_empty=True)

    def flac(self, data, mono=None):
        _expect_src(1, data)
        return self._run(B.make_desc(N.CODEC_FLAC), data, mono, N.F64)

    def qoa(self, data, mono=None):
        _expect_src(1, data)
        return self._run(B.make_desc(N.CODEC_QOA), data, mono, N.F64)

    @staticmethod
    def _strip_headers(fn, kind):
        """`ignoreHeader` in reader-function mode (:2983-2989, :3053-3060, :3097-3101): every LATER piece that starts with the container's magic
        has its header cut off with the reference's own patterns — including what they do to a header they do not fit (a WAV piece whose
        `data` chunk does not follow `WAVE` within one byte makes string.sub raise; AIFF takes the SSND chunk's SIZE for its offset)."""
        import re

        def wrapped():
            d = fn()
            if d is None:
                return None
            d = bytes(d)
            if kind == N.CONTAINER_WAV:
                if re.match(rb"RIFF.{4}WAVE", d, re.S):
                    m = re.match(rb"RIFF.{4}WAVE.?data.{4}", d, re.S)
                    if not m:
                        raise LuaError("bad argument #2 to 'sub' (number expected, got nil)")
                    return d[m.end():]
            elif kind == N.CONTAINER_AIFF:
                if re.match(rb"FORM.{4}AIF[FC]", d, re.S):
                    m = re.match(rb"FORM.{4}AIF[FC].*?SSND(.{4}).{4}", d, re.S)
                    if not m:
                        raise LuaError("bad argument #2 to 'unpack' (string expected, got nil)")
                    return d[m.end() + int.from_bytes(m.group(1), "big"):]
            else:
                if re.match(rb".snd", d, re.S):
                    if len(d) < 8:
                        raise LuaError("data string too short")
                    off = int.from_bytes(d[4:8], "big")
                    return d[max(off - 1, 0):]
            return d
        return wrapped

    def _container(self, data, kind, mono, ignoreHeader=None):
        fn = None
        if callable(data):  # "the first chunk MUST contain the ENTIRE header" (:2918): the header walk runs on it, the payload that follows it is the first piece
            fn, data = data, data()
            _expect(1, data, "string")
            if ignoreHeader:
                fn = self._strip_headers(fn, kind)
        c, p = _parse(data, kind, stream=True) if fn is None else _parse_first_piece(data, kind)
        d = _desc_copy(c)
        dtype = N.F64 if d.codec in (N.CODEC_PCM, N.CODEC_DFPWM) else N.I8  # what stream.pcm / .dfpwm vs .g711 / .adpcm / .msadpcm hand out
        if fn is not None:
            it, length = self._run_fn(d, fn, p, mono, dtype)
        else:
            it, length = self._run(d, p, mono, dtype, endless_empty=d.codec == N.CODEC_G711)
        return it, (length if math.isnan(c.length_seconds) else c.length_seconds)

    def wav(self, data, mono=None, ignoreHeader=None):  # :2927: header walk (library) + dispatch (:2992-2996)
        _expect_src(1, data)
        return self._container(data, N.CONTAINER_WAV, mono, ignoreHeader)

    def aiff(self, data, mono=None, ignoreHeader=None):  # :3016
        _expect_src(1, data)
        _expect(2, mono, "boolean", "nil")
        return self._container(data, N.CONTAINER_AIFF, mono, ignoreHeader)

    def au(self, data, mono=None, ignoreHeader=None):  # :3086
        _expect_src(1, data)
        _expect(2, mono, "boolean", "nil")
        return self._container(data, N.CONTAINER_AU, mono, ignoreHeader)


stream = _StreamNS()


# ---------------------------------------------------------------- aukit.effects.*  (aukit.lua:3349-3618)
class _EffectsNS:
    @staticmethod
    def _fx(audio, name, *args):
        if not isinstance(audio, Audio):
            raise LuaError("bad argument #1 (expected Audio, got " + type(audio).__name__ + ")")
        _wrap(B.effect, context(), audio._h, name, *args)
        return audio  # in place, returns the same object

    def amplify(self, audio, multiplier):
        _expect(2, multiplier, "number")
        return self._fx(audio, "amplify", multiplier)

    def speed(self, audio, multiplier):
        _expect(2, multiplier, "number")
        return self._fx(audio, "speed", multiplier, N.INTERP[defaultInterpolation])

    def fade(self, audio, startTime, startAmplitude, endTime, endAmplitude):
        for i, v in enumerate((startTime, startAmplitude, endTime, endAmplitude)):
            _expect(i + 2, v, "number")
        return self._fx(audio, "fade", startTime, startAmplitude, endTime, endAmplitude)

    def invert(self, audio):
        return self._fx(audio, "invert")

    def normalize(self, audio, peakAmplitude=None, independent=None):
        return self._fx(audio, "normalize", 1 if peakAmplitude is None else peakAmplitude, 1.0 if independent else 0.0)

    def center(self, audio):
        return self._fx(audio, "center")

    def trim(self, audio, threshold=None):
        return self._fx(audio, "trim", 1 / 65536 if threshold is None else threshold)

    def delay(self, audio, delay, multiplier=None):
        _expect(2, delay, "number")
        return self._fx(audio, "delay", delay, 0.5 if multiplier is None else multiplier)

    def echo(self, audio, delay=None, multiplier=None):
        return self._fx(audio, "echo", 1 if delay is None else delay, 0.5 if multiplier is None else multiplier)

    def reverb(self, audio, delay=None, decay=None, wetMultiplier=None, dryMultiplier=None):
        return self._fx(audio, "reverb", 100 if delay is None else delay, 0.3 if decay is None else decay, 1 if wetMultiplier is None else wetMultiplier,
                        0 if dryMultiplier is None else dryMultiplier)

    def lowpass(self, audio, frequency):
        _expect(2, frequency, "number")
        return self._fx(audio, "lowpass", frequency)

    def highpass(self, audio, frequency):
        _expect(2, frequency, "number")
        return self._fx(audio, "highpass", frequency)


effects = _EffectsNS()
