"""GPU parity for FLAC on streams no encoder writes but the reference decoder accepts (aukit.lua:380-470 has no sanity checks):
hand-assembled frames whose predictor order exceeds the block / the Rice partition size, frames with a bad header CRC-8
(the reference ignores it, :553), Rice escape partitions, and the double row path forced on ordinary streams."""
import numpy as np
import pytest

from tests.util import pcm16

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "stream"])
def _both_register_decoders(request, monkeypatch):
    """round 6: small batches decode with k_flac_pq (a parser and a predictor wave per 64 frames), batches that fill the chip with k_flac_stream
    (one wave) — every case of this module runs through both (AUKIT_FLAC_DECODER, read per call; the round-4 kernel stays behind "fused")"""
    if request.param != "auto":
        monkeypatch.setenv("AUKIT_FLAC_DECODER", request.param)



def _B():
    from aukit_amd import batch as B
    return B


def _N():
    from aukit_amd import _native as N
    return N


class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, v, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def s(self, v, n):
        self.u(v & ((1 << n) - 1), n)

    def rice(self, v, k):
        u = 2 * v if v >= 0 else -2 * v - 1
        self.bits.extend([0] * (u >> k))
        self.bits.append(1)
        if k:
            self.u(u & ((1 << k) - 1), k)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        self.align()
        a = np.array(self.bits, dtype=np.uint8).reshape(-1, 8)
        return bytes(np.packbits(a, axis=1).ravel())


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def streaminfo(rate, channels, depth, nsamples):
    w = BitWriter()
    w.u(16, 16); w.u(65535, 16); w.u(0, 24); w.u(0, 24)
    w.u(rate, 20); w.u(channels - 1, 3); w.u(depth - 1, 5); w.u(nsamples, 36)
    body = w.bytes() + bytes(16)
    assert len(body) == 34
    return b"fLaC" + bytes([0x80, 0, 0, 34]) + body


def frame(bs, subframes, chan_asgn=0, good_crc=True, number=0):
    """subframes: list of callables(BitWriter) writing one subframe each."""
    w = BitWriter()
    w.u(0x3FFE, 14); w.u(0, 2)
    w.u(6, 4)            # block size: 8-bit (blocksize - 1) follows
    w.u(9, 4)            # 44.1 kHz
    w.u(chan_asgn, 4); w.u(4, 3); w.u(0, 1)
    w.u(number, 8)       # "UTF-8" frame number, one byte
    w.u(bs - 1, 8)
    hdr = w.bytes()
    w.u(crc8(hdr) if good_crc else crc8(hdr) ^ 0x5A, 8)
    for sf in subframes:
        sf(w)
    w.align()
    w.u(0xBEEF, 16)      # CRC-16: ignored by the reference (:557)
    return w.bytes()


def lpc_subframe(order, depth, warm, precision, shift, coefs, porder, bs, params, residuals, escape_bits=None):
    def write(w):
        w.u(0, 1); w.u(31 + order, 6); w.u(0, 1)
        for v in warm:
            w.s(int(v), depth)
        w.u(precision - 1, 4); w.s(shift, 5)
        for c in coefs:
            w.s(int(c), precision)
        w.u(0, 2); w.u(porder, 4)
        nparts, psize = 1 << porder, bs >> porder
        it = iter(residuals)
        for p in range(nparts):
            start = p * psize + (order if p == 0 else 0)
            cnt = max(0, (p + 1) * psize - start)
            if escape_bits is not None and p % 2 == 1:
                w.u(15, 4); w.u(escape_bits, 5)
                for _ in range(cnt):
                    w.s(int(next(it)), escape_bits)
            else:
                w.u(params[p % len(params)], 4)
                for _ in range(cnt):
                    w.rice(int(next(it)), params[p % len(params)])
    return write


def fixed_subframe(order, depth, warm, porder, bs, param, residuals):
    def write(w):
        w.u(0, 1); w.u(8 + order, 6); w.u(0, 1)
        for v in warm:
            w.s(int(v), depth)
        w.u(0, 2); w.u(porder, 4)
        nparts, psize = 1 << porder, bs >> porder
        it = iter(residuals)
        for p in range(nparts):
            start = p * psize + (order if p == 0 else 0)
            w.u(param, 4)
            for _ in range(max(0, (p + 1) * psize - start)):
                w.rice(int(next(it)), param)
    return write


def _odd_stream(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    r = lambda n, a=40: rng.integers(-a, a + 1, n)
    frames, total = [], 0
    # 1. ordinary LPC frame
    frames.append(frame(64, [lpc_subframe(8, 16, r(8, 2000), 12, 9, r(8, 600), 2, 64, [3, 5], r(64))], number=0)); total += 64
    # 2. predictor order (20) larger than the block (16), one partition: only warm-up entries survive
    frames.append(frame(16, [lpc_subframe(20, 16, r(20, 2000), 10, 8, r(20, 200), 0, 16, [4], [])], number=1)); total += 16
    # 3. order 20 > block 16 with two partitions: the second one overwrites warm-up entries 9..16 (:400)
    frames.append(frame(16, [lpc_subframe(20, 16, r(20, 2000), 10, 8, r(20, 200), 1, 16, [4], r(8))], number=2)); total += 16
    # 4. partition size 4 < order 8: partitions 1.. overwrite warm-up entries before prediction starts; bad CRC-8 on top
    frames.append(frame(64, [lpc_subframe(8, 16, r(8, 2000), 12, 9, r(8, 600), 4, 64, [2, 6, 3], r(60))], good_crc=False, number=3)); total += 64
    # 5. fixed predictor with the same overlap, then an escape-coded LPC frame, then an ordinary frame (the chain must still line up)
    frames.append(frame(32, [fixed_subframe(4, 16, r(4, 3000), 4, 32, 3, r(30))], number=4)); total += 32
    frames.append(frame(128, [lpc_subframe(3, 16, r(3, 2000), 9, 6, r(3, 100), 3, 128, [5], r(125, 1000), escape_bits=12)], number=5)); total += 128
    frames.append(frame(64, [lpc_subframe(8, 16, r(8, 2000), 12, 9, r(8, 600), 2, 64, [3, 5], r(64))], number=6)); total += 64
    return streaminfo(44100, 1, 16, total) + b"".join(frames), total


@pytest.mark.parametrize("wide", [False, True])
def test_flac_frames_no_encoder_writes(ctx, oracle, monkeypatch, wide):
    B, N = _B(), _N()
    if wide:
        monkeypatch.setenv("AUKIT_FLAC_WIDE", "1")
    streams, totals = zip(*[_odd_stream(s) for s in range(6)])
    got = B.decode(ctx, B.Batch.upload(ctx, list(streams)), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
    for s, t, g in zip(streams, totals, got):
        ref = oracle.flac(s)
        assert len(ref.data[0]) == t
        assert np.array_equal(g[0], ref.data[0])
    out, ck = B.stream_decode(ctx, B.Batch.upload(ctx, list(streams)), B.make_desc(N.CODEC_FLAC), "cubic", dtype=N.F64)
    for s, g in zip(streams, out.download()):
        assert np.array_equal(g[0], oracle.stream_flac(s, oracle.CUBIC).data[0])


def test_flac_int64_rows_match_int32_rows(ctx, oracle, monkeypatch):
    """Ordinary streams through the double row path (taken for 32-bit audio / overflowing values) = the int32 path = the oracle."""
    B, N = _B(), _N()
    st = np.stack([pcm16(20000, 44100, 5, 0), pcm16(20000, 44100, 5, 1)], 1).astype(np.int64)
    st[5000:9000] = (st[5000:9000] >> 3) << 3
    streams = [oracle.gen_flac(st.ravel(), 2, 16, 44100, bs) for bs in (4096, 1000)] + [oracle.gen_flac((st[:, 0] * 200).ravel(), 1, 24, 48000, 576)]
    for group in (streams[:2], streams[2:]):
        bt = B.Batch.upload(ctx, group)
        a = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
        monkeypatch.setenv("AUKIT_FLAC_WIDE", "1")
        b = B.decode(ctx, bt, B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
        rs = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_FLAC), 48000, "cubic", dtype=N.F64).download()
        monkeypatch.delenv("AUKIT_FLAC_WIDE")
        rs32 = B.decode_resample(ctx, bt, B.make_desc(N.CODEC_FLAC), 48000, "cubic", dtype=N.F64).download()
        for s, x, y, u, v in zip(group, a, b, rs, rs32):
            ref = oracle.flac(s)
            for c in range(len(ref.data)):
                assert np.array_equal(x[c], ref.data[c]) and np.array_equal(y[c], ref.data[c])
                assert np.array_equal(u[c], v[c])


def test_flac_values_beyond_int32_fall_back_to_int64_rows(ctx, oracle):
    """A Rice code with a very long unary prefix yields a residual beyond 32 bits: the reference carries it in a double."""
    B, N = _B(), _N()
    def big(w):
        w.u(0, 1); w.u(8 + 0, 6); w.u(0, 1)          # FIXED order 0: samples = residuals
        w.u(0, 2); w.u(0, 4)                          # one partition
        w.u(14, 4)                                    # Rice parameter 14
        vals = [0] * 16
        vals[3] = (1 << 33) + 12345
        vals[7] = -(1 << 32) - 99
        for v in vals:
            u = 2 * v if v >= 0 else -2 * v - 1
            q = u >> 14
            w.bits.extend([0] * q); w.bits.append(1); w.u(u & 0x3FFF, 14)
    s = streaminfo(44100, 1, 16, 16) + frame(16, [big])
    ref = oracle.flac(s)
    got = B.decode(ctx, B.Batch.upload(ctx, [s]), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()[0]
    assert np.array_equal(got[0], ref.data[0])
    assert abs(ref.data[0][3]) > 2 ** 16  # far outside [-1, 1): really the wide path


def _lpc_stream(coefs, precision, shift, seed, nframes=8, bs=256):
    """An encoder's view: the residuals of a given predictor on a 16-bit signal (lossless whatever the coefficients are)."""
    x = pcm16(nframes * bs, 44100, 5, seed).astype(np.int64)
    order = len(coefs)
    frames = []
    for f in range(nframes):
        s = [int(v) for v in x[f * bs:(f + 1) * bs]]
        res = [s[i] - (sum(int(coefs[j]) * s[i - 1 - j] for j in range(order)) >> shift) for i in range(order, bs)]
        k = min(14, max(1, max(abs(r) for r in res).bit_length() - 1))
        frames.append(frame(bs, [lpc_subframe(order, 16, s[:order], precision, shift, coefs, 1, bs, [k], res)], number=f))
    return streaminfo(44100, 1, 16, nframes * bs) + b"".join(frames), x


@pytest.mark.parametrize("coefs, precision, shift", [
    ([900, -420, 30, -11, 7, -3, 2, -1], 12, 9),                       # sum |c| < 2^15: one multiply-add per tap (k_flac_restore_fast, ONE)
    ([16000, -15800, 9000, -2000, 500, -300, 100, -50], 15, 14),       # sum |c| = 43750 > 2^15: a tap is two 24-bit multiply-adds
    ([16383, 16383, -16384, 16383, -16384, 16383, 16383, -16384, 16383, 16383, -16384, 16383], 15, 14),   # the largest 15-bit taps
    ([4, -2, 1], 5, 0),                                                 # shift 0
])
def test_flac_predictors_small_and_large(ctx, oracle, coefs, precision, shift):
    """The int32 prediction kernel picks its arithmetic per wave from the size of the coefficients; every choice restores the encoder's samples."""
    B, N = _B(), _N()
    made = [_lpc_stream(coefs, precision, shift, seed) for seed in range(9)]
    got = B.decode(ctx, B.Batch.upload(ctx, [m[0] for m in made]), B.make_desc(N.CODEC_FLAC), dtype=N.F64).download()
    for (s, x), g in zip(made, got):
        ref = oracle.flac(s)
        assert np.array_equal(ref.data[0] * 65536.0, x.astype(np.float64))   # s / 2^sampleDepth (aukit.lua:505)
        assert np.array_equal(g[0], ref.data[0])
