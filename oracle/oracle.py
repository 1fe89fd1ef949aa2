"""ctypes binding of the CPU ORACLE (oracle/libaukit_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package `aukit_amd` never imports it.
See oracle/ork.h for the parity-pinning status of the restatement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libaukit_oracle.so")
MAX_CH = 64

OK, E_ARG, E_LUA, E_NOMEM, E_UNSUPPORTED = 0, -1, -2, -3, -4
NONE, LINEAR, CUBIC, SINC = 0, 1, 2, 3
INTERP = {"none": 0, "linear": 1, "cubic": 2, "sinc": 3}
SIGNED, UNSIGNED, FLOAT = 0, 1, 2
DTYPE = {"signed": 0, "unsigned": 1, "float": 2}


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ork_core.c", "ork_codecs.c", "ork_stream.c", "ork_gen.c", "ork_check.c", "ork.h", "ork_internal.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


class _Audio(C.Structure):
    _fields_ = [("channels", C.c_int), ("sample_rate", C.c_double), ("len", C.c_size_t * MAX_CH), ("data", C.POINTER(C.c_double) * MAX_CH)]


class _Stream(C.Structure):
    _fields_ = [("channels", C.c_int), ("nchunks", C.c_int), ("chunk_len", C.POINTER(C.c_size_t)), ("chunk_pos", C.POINTER(C.c_double)),
                ("len", C.c_size_t * MAX_CH), ("data", C.POINTER(C.c_double) * MAX_CH), ("length_seconds", C.c_double), ("final_status", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.ork_last_error.restype = C.c_char_p
        _lib.ork_clamp.restype = C.c_double
        _lib.ork_clamp.argtypes = [C.c_double] * 3
    return _lib


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().ork_last_error().decode())


class Audio:
    """Mirror of aukit.Audio on the host: list of float64 numpy arrays (one per channel)."""

    def __init__(self, data, sample_rate):
        self.data = [np.ascontiguousarray(d, dtype=np.float64) for d in data]
        self.sample_rate = float(sample_rate)

    @property
    def channels(self):
        return len(self.data)

    def copy(self):
        return Audio([d.copy() for d in self.data], self.sample_rate)

    def _c(self):
        a = _Audio()
        a.channels = len(self.data)
        a.sample_rate = self.sample_rate
        for i, d in enumerate(self.data):
            a.len[i] = d.shape[0]
            a.data[i] = d.ctypes.data_as(C.POINTER(C.c_double))
        return a


def _take_audio(a):
    out = Audio([np.ctypeslib.as_array(a.data[c], shape=(max(a.len[c], 1),))[: a.len[c]].copy() for c in range(a.channels)], a.sample_rate)
    lib().ork_audio_free(C.byref(a))
    return out


class Stream:
    def __init__(self, s):
        self.channels = s.channels
        self.nchunks = s.nchunks
        self.length_seconds = s.length_seconds
        self.final_status = s.final_status
        self.chunk_len = np.array([s.chunk_len[i] for i in range(s.nchunks * s.channels)], dtype=np.int64).reshape(s.nchunks, s.channels)
        self.chunk_pos = np.array([s.chunk_pos[i] for i in range(s.nchunks)], dtype=np.float64)
        self.data = [np.ctypeslib.as_array(s.data[c], shape=(max(s.len[c], 1),))[: s.len[c]].copy() for c in range(s.channels)]

    def chunks(self):
        """List of chunks; each chunk is a list of per-channel arrays (what the Lua iterator returns)."""
        out, off = [], [0] * self.channels
        for k in range(self.nchunks):
            ch = []
            for c in range(self.channels):
                n = int(self.chunk_len[k, c])
                ch.append(self.data[c][off[c]: off[c] + n])
                off[c] += n
            out.append(ch)
        return out


def _take_stream(s):
    out = Stream(s)
    lib().ork_stream_free(C.byref(s))
    return out


def _bytes(b):
    b = bytes(b) if not isinstance(b, (bytes, bytearray)) else b
    n = len(b)
    buf = (C.c_uint8 * max(n, 1)).from_buffer_copy(b if n else b"\0")
    return buf, n


def _ints(v):
    if v is None:
        return None
    arr = (C.c_int * len(v))(*[int(x) for x in v])
    return arr


def set_sinc_window(w):
    lib().ork_set_sinc_window(int(w))


def interp(mode, data, x):
    d = np.ascontiguousarray(data, dtype=np.float64)
    out = C.c_double()
    _check(lib().ork_interp(int(mode), d.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(len(d)), C.c_double(x), C.byref(out)))
    return out.value


# ---- Audio methods ----
def resample(a, new_rate, mode):
    out = _Audio()
    ca = a._c()
    _check(lib().ork_resample(C.byref(ca), C.c_double(new_rate), int(mode), C.byref(out)))
    return _take_audio(out)


def mono(a):
    out = _Audio()
    ca = a._c()
    _check(lib().ork_mono(C.byref(ca), C.byref(out)))
    return _take_audio(out)


def mix(audios, amplifier=1.0):
    cs = [a._c() for a in audios]
    arr = (C.POINTER(_Audio) * len(cs))(*[C.pointer(c) for c in cs])
    out = _Audio()
    _check(lib().ork_mix(arr, len(cs), C.c_double(amplifier), C.byref(out)))
    return _take_audio(out)


def encode_pcm(a, bit_depth=8, data_type=SIGNED, interleaved=True):
    p = C.POINTER(C.c_double)()
    n = C.c_size_t()
    ca = a._c()
    _check(lib().ork_encode_pcm(C.byref(ca), bit_depth, data_type, int(interleaved), C.byref(p), C.byref(n)))
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[: n.value].copy()
    lib().ork_free(p)
    return out


def audio_dfpwm(a, interleaved=True):
    p = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    ca = a._c()
    _check(lib().ork_audio_dfpwm(C.byref(ca), int(interleaved), C.byref(p), C.byref(n)))
    out = bytes(np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[: n.value])
    lib().ork_free(p)
    return out


# ---- loaders ----
def pcm(data, bit_depth=8, data_type=SIGNED, channels=1, sample_rate=48000, interleaved=True, big_endian=False):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_pcm(buf, C.c_size_t(n), bit_depth, data_type, channels, C.c_double(sample_rate), int(interleaved), int(big_endian), C.byref(out)))
    return _take_audio(out)


def pcm_table(values, bit_depth=8, data_type=SIGNED, channels=1, sample_rate=48000, interleaved=True):
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = _Audio()
    _check(lib().ork_pcm_table(v.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(len(v)), bit_depth, data_type, channels, C.c_double(sample_rate), int(interleaved), C.byref(out)))
    return _take_audio(out)


def adpcm(data, channels=1, sample_rate=48000, top_first=True, interleaved=True, predictor=None, step_index=None):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_adpcm(buf, C.c_size_t(n), channels, C.c_double(sample_rate), int(top_first), int(interleaved), _ints(predictor), _ints(step_index), C.byref(out)))
    return _take_audio(out)


def wav_adpcm(data, block_align, channels, sample_rate):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_wav_adpcm(buf, C.c_size_t(n), block_align, channels, C.c_double(sample_rate), C.byref(out)))
    return _take_audio(out)


def msadpcm(data, block_align, channels=1, sample_rate=48000, coefficients=None):
    buf, n = _bytes(data)
    out = _Audio()
    c1 = _ints(coefficients[0]) if coefficients else None
    c2 = _ints(coefficients[1]) if coefficients else None
    nc = len(coefficients[0]) if coefficients else 0
    _check(lib().ork_msadpcm(buf, C.c_size_t(n), block_align, channels, C.c_double(sample_rate), c1, c2, nc, C.byref(out)))
    return _take_audio(out)


def g711(data, ulaw, channels=1, sample_rate=8000):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_g711(buf, C.c_size_t(n), int(ulaw), channels, C.c_double(sample_rate), C.byref(out)))
    return _take_audio(out)


def dfpwm(data, channels=1, sample_rate=48000):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_dfpwm(buf, C.c_size_t(n), channels, C.c_double(sample_rate), C.byref(out)))
    return _take_audio(out)


def mdfpwm(data):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_mdfpwm(buf, C.c_size_t(n), C.byref(out)))
    return _take_audio(out)


def qoa(data):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_qoa(buf, C.c_size_t(n), C.byref(out)))
    return _take_audio(out)


def flac(data):
    buf, n = _bytes(data)
    out = _Audio()
    _check(lib().ork_flac(buf, C.c_size_t(n), C.byref(out)))
    return _take_audio(out)


# ---- raw DFPWM codec ----
class _Dec(C.Structure):
    _fields_ = [("p", C.c_int * 3), ("low_pass_charge", C.c_int), ("previous_charge", C.c_int), ("previous_bit", C.c_int)]


class _Enc(C.Structure):
    _fields_ = [("p", C.c_int * 3), ("previous_charge", C.c_int)]


class DfpwmDecoder:
    """cc.audio.dfpwm.make_decoder(): carries state across calls."""

    def __init__(self):
        self.s = _Dec()
        lib().ork_dfpwm_dec_init(C.byref(self.s))

    def __call__(self, data):
        buf, n = _bytes(data)
        out = np.zeros(max(n * 8, 1), dtype=np.int8)
        lib().ork_dfpwm_decode(C.byref(self.s), buf, C.c_size_t(n), out.ctypes.data_as(C.POINTER(C.c_int8)))
        return out[: n * 8]


class DfpwmEncoder:
    """cc.audio.dfpwm.make_encoder(): carries state across calls."""

    def __init__(self):
        self.s = _Enc()
        lib().ork_dfpwm_enc_init(C.byref(self.s))

    def __call__(self, samples):
        v = np.ascontiguousarray(samples, dtype=np.float64)
        nb = (len(v) + 7) // 8
        out = np.zeros(max(nb, 1), dtype=np.uint8)
        _check(lib().ork_dfpwm_encode(C.byref(self.s), v.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(len(v)), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return bytes(out[:nb])


def dfpwm_encode(samples):
    return DfpwmEncoder()(samples)


# ---- effects (in place on a copy; returns the modified Audio) ----
def _fx(name, a, *args):
    ca = a._c()
    fn = getattr(lib(), "ork_fx_" + name)
    cargs = [C.c_double(x) if isinstance(x, float) else x for x in args]
    rc = fn(C.byref(ca), *cargs)
    if name == "speed" and rc == 0:
        # the C side replaced the channel buffers
        new = Audio([np.ctypeslib.as_array(ca.data[c], shape=(max(ca.len[c], 1),))[: ca.len[c]].copy() for c in range(ca.channels)], ca.sample_rate)
        for c in range(ca.channels):
            lib().ork_free(ca.data[c])
        a.data = new.data
    _check(rc)
    return a


def fx_amplify(a, m): return _fx("amplify", a, float(m))
def fx_invert(a): return _fx("invert", a)
def fx_fade(a, t0, a0, t1, a1): return _fx("fade", a, float(t0), float(a0), float(t1), float(a1))
def fx_normalize(a, peak=1.0, independent=False): return _fx("normalize", a, float(peak), int(bool(independent)))
def fx_center(a): return _fx("center", a)
def fx_trim(a, threshold=1 / 65536): return _fx("trim", a, float(threshold))
def fx_delay(a, delay, multiplier=0.5): return _fx("delay", a, float(delay), float(multiplier))
def fx_echo(a, delay=1.0, multiplier=0.5): return _fx("echo", a, float(delay), float(multiplier))
def fx_reverb(a, delay=100.0, decay=0.3, wet=1.0, dry=0.0): return _fx("reverb", a, float(delay), float(decay), float(wet), float(dry))
def fx_lowpass(a, f): return _fx("lowpass", a, float(f))
def fx_highpass(a, f): return _fx("highpass", a, float(f))


def fx_speed(a, m, default_interp=LINEAR):
    # ork_fx_speed frees/replaces the buffers it is given, so hand it malloc'd copies
    ca = _Audio()
    ca.channels = a.channels
    ca.sample_rate = a.sample_rate
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    for c, d in enumerate(a.data):
        p = libc.malloc(C.c_size_t(max(d.nbytes, 8)))
        C.memmove(p, d.ctypes.data, d.nbytes)
        ca.len[c] = len(d)
        ca.data[c] = C.cast(p, C.POINTER(C.c_double))
    rc = lib().ork_fx_speed(C.byref(ca), C.c_double(m), int(default_interp))
    res = _take_audio(ca)
    _check(rc)
    a.data = res.data
    return a


# ---- streams ----
def stream_pcm(data, bit_depth=8, data_type=SIGNED, channels=1, sample_rate=48000, big_endian=False, mono=False, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_pcm(buf, C.c_size_t(n), bit_depth, data_type, channels, C.c_double(sample_rate), int(big_endian), int(bool(mono)), int(interp), C.byref(out)))
    return _take_stream(out)


def stream_dfpwm(data, sample_rate=48000, channels=1, mono=False, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_dfpwm(buf, C.c_size_t(n), C.c_double(sample_rate), channels, int(bool(mono)), int(interp), C.byref(out)))
    return _take_stream(out)


def stream_mdfpwm(data, mono=False):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_mdfpwm(buf, C.c_size_t(n), int(bool(mono)), C.byref(out)))
    return _take_stream(out)


def stream_msadpcm(data, block_align, channels=1, sample_rate=48000, mono=False, coefficients=None, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    c1 = _ints(coefficients[0]) if coefficients else None
    c2 = _ints(coefficients[1]) if coefficients else None
    nc = len(coefficients[0]) if coefficients else 0
    _check(lib().ork_stream_msadpcm(buf, C.c_size_t(n), block_align, channels, C.c_double(sample_rate), int(bool(mono)), c1, c2, nc, int(interp), C.byref(out)))
    return _take_stream(out)


def stream_adpcm(data, block_align, channels=1, sample_rate=48000, mono=False, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_adpcm(buf, C.c_size_t(n), block_align, channels, C.c_double(sample_rate), int(bool(mono)), int(interp), C.byref(out)))
    return _take_stream(out)


def stream_g711(data, ulaw, channels=1, sample_rate=8000, mono=False, interp=LINEAR, max_calls=None):
    buf, n = _bytes(data)
    if max_calls is None:
        per = int(sample_rate) * channels
        max_calls = (n + per - 1) // per
    out = _Stream()
    _check(lib().ork_stream_g711(buf, C.c_size_t(n), int(bool(ulaw)), channels, C.c_double(sample_rate), int(bool(mono)), int(interp), int(max_calls), C.byref(out)))
    return _take_stream(out)


def stream_flac(data, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_flac(buf, C.c_size_t(n), int(interp), C.byref(out)))
    return _take_stream(out)


def stream_qoa(data, mono=False, interp=LINEAR):
    buf, n = _bytes(data)
    out = _Stream()
    _check(lib().ork_stream_qoa(buf, C.c_size_t(n), int(bool(mono)), int(interp), C.byref(out)))
    return _take_stream(out)


# ---- generators (ork_gen.c) ----
def gen_g711(pcm16, ulaw=True):
    p = np.ascontiguousarray(pcm16, dtype=np.int16)
    out = np.zeros(max(len(p), 1), dtype=np.uint8)
    lib().ork_gen_g711(p.ctypes.data_as(C.POINTER(C.c_int16)), C.c_size_t(len(p)), int(ulaw), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return bytes(out[: len(p)])


def gen_ima(pcm16_interleaved, channels=1, block_align=512, max_index=88):
    p = np.ascontiguousarray(pcm16_interleaved, dtype=np.int16)
    frames = len(p) // channels
    if block_align <= 4 * channels or (block_align - 4 * channels) % (4 * channels):
        raise ValueError("block_align must be 4*channels + a whole number of 4-byte words per channel")
    spb = (block_align - 4 * channels) * 2 // channels  # samples per block: small blocks spend most of their bytes on headers
    out = np.zeros((frames // spb + 2) * block_align + 64, dtype=np.uint8)
    f = lib().ork_gen_ima
    f.restype = C.c_size_t
    n = f(p.ctypes.data_as(C.POINTER(C.c_int16)), C.c_size_t(frames), channels, block_align, max_index, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return bytes(out[:n])


def gen_msadpcm(pcm16_interleaved, channels=1, block_align=1024):
    p = np.ascontiguousarray(pcm16_interleaved, dtype=np.int16)
    frames = len(p) // channels
    out = np.zeros(frames * channels + block_align * 4 + 64, dtype=np.uint8)
    f = lib().ork_gen_msadpcm
    f.restype = C.c_size_t
    n = f(p.ctypes.data_as(C.POINTER(C.c_int16)), C.c_size_t(frames), channels, block_align, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return bytes(out[:n])


def gen_qoa(pcm16_interleaved, channels=1, sample_rate=44100):
    p = np.ascontiguousarray(pcm16_interleaved, dtype=np.int16)
    frames = len(p) // channels
    nfr = (frames + 5119) // 5120
    out = np.zeros(8 + nfr * (8 + 16 * channels) + ((frames + 19) // 20 + nfr) * 8 * channels + 64, dtype=np.uint8)
    f = lib().ork_gen_qoa
    f.restype = C.c_size_t
    n = f(p.ctypes.data_as(C.POINTER(C.c_int16)), C.c_size_t(frames), channels, C.c_uint(sample_rate), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return bytes(out[:n])


def gen_flac(pcm_interleaved, channels=2, depth=16, sample_rate=44100, blocksize=4096, salt=0):
    """salt: which subframe type / predictor order / partition order / stereo mode each frame gets (0: as the golden fixtures know it)"""
    p = np.ascontiguousarray(pcm_interleaved, dtype=np.int32)
    frames = len(p) // channels
    f = lib().ork_gen_flac_salt
    f.restype = C.POINTER(C.c_uint8)
    n = C.c_size_t()
    ptr = f(p.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(frames), channels, depth, C.c_uint(sample_rate), blocksize, C.c_uint(salt), C.byref(n))
    out = bytes(np.ctypeslib.as_array(ptr, shape=(max(n.value, 1),))[: n.value])
    lib().ork_free(ptr)
    return out


def gen_mdfpwm(left_bytes, right_bytes, artist=b"", title=b"", album=b""):
    """Wrap two DFPWM byte strings (multiples of 6000 bytes each) as MDFPWMv3."""
    assert len(left_bytes) == len(right_bytes)
    body = b""
    for i in range(0, len(left_bytes), 6000):
        l, r = left_bytes[i:i + 6000], right_bytes[i:i + 6000]
        body += l.ljust(6000, b"\x55") + r.ljust(6000, b"\x55")
    import struct
    hdr = b"MDFPWM\x03" + struct.pack("<I", len(left_bytes) + len(right_bytes))
    for s in (artist, title, album):
        hdr += bytes([len(s)]) + s
    return hdr + body
