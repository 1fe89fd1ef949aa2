"""CPU tests that pin the ORACLE with checks that do not depend on the restatement being right.

The reference ships no tests, fixtures or golden vectors, and no Lua interpreter exists in the
build image, so the oracle cannot be compared with the reference's own output.  These known-answer tests are
the independent anchors named in SURVEY.md §8c:
  * ITU-T G.711 expansion tables (µ-law / A-law), every code point;
  * FLAC is lossless: decode(encode(pcm)) == pcm for every subframe type / stereo mode / partition layout;
  * QOA: a straight int32 decoder written from the qoaformat.org specification agrees sample for sample;
  * DFPWM: encoder and decoder track the input (lossy), and the 1-bit stream round-trips through both states;
  * hand-computed micro-vectors for the reference quirks (§8.0).
"""
import numpy as np
import pytest

from tests.util import pcm16, signal


def _itu_ulaw(u):
    u = ~u & 0xFF
    t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4)
    return (0x84 - t) if (u & 0x80) else (t - 0x84)


def _itu_alaw(a):
    a ^= 0x55
    t = (a & 0x0F) << 4
    seg = (a & 0x70) >> 4
    if seg == 0:
        t += 8
    elif seg == 1:
        t += 0x108
    else:
        t = (t + 0x108) << (seg - 1)
    return t if (a & 0x80) else -t


def test_g711_matches_itu_tables(oracle):
    codes = bytes(range(256))
    u = oracle.g711(codes, True, 1, 8000).data[0]
    a = oracle.g711(codes, False, 1, 8000).data[0]
    assert np.array_equal(u, np.array([_itu_ulaw(b) / 32768.0 for b in range(256)]))
    assert np.array_equal(a, np.array([_itu_alaw(b) / 32768.0 for b in range(256)]))


def test_g711_encoder_decoder_roundtrip(oracle):
    x = pcm16(8000, 8000, 2, 0)
    for ulaw in (True, False):
        y = oracle.g711(oracle.gen_g711(x, ulaw), ulaw, 1, 8000).data[0] * 32768
        assert np.sqrt(np.mean((y - x) ** 2)) < 400  # companding error of a ±24 000 signal


@pytest.mark.parametrize("depth", [8, 16, 24])
@pytest.mark.parametrize("ch", [1, 2])
def test_flac_is_lossless(oracle, depth, ch):
    rng = np.random.Generator(np.random.PCG64(depth * 10 + ch))
    n = 4096 * 12 + 1234  # 13 frames: every (frame % 4) stereo mode, every subframe variant, a short explicit-size last frame
    base = (signal(n, 44100, 5, ch) * (2 ** (depth - 1) - 1) * 0.9).astype(np.int64)
    cols = [base]
    if ch == 2:
        cols.append((base * 0.5 + rng.integers(-40, 40, n)).astype(np.int64))
    pcm = np.stack(cols, 1)
    pcm[4096:8192] = 1234 if depth > 8 else 12        # CONSTANT subframes
    pcm[8192:12288] = (pcm[8192:12288] >> 3) << 3      # wasted bits
    pcm[12288:16384] = rng.integers(-(2 ** (depth - 1)), 2 ** (depth - 1), (4096, ch))  # incompressible → escape partitions
    data = oracle.gen_flac(pcm.ravel(), ch, depth, 44100, 4096)
    dec = oracle.flac(data)
    assert dec.channels == ch and dec.sample_rate == 44100
    got = np.stack([np.round(dec.data[c] * 2 ** depth) for c in range(ch)], 1).astype(np.int64)  # Q14: / 2^depth
    assert np.array_equal(got, pcm)


def test_flac_other_blocksizes(oracle):
    pcm = np.stack([pcm16(5000, 44100, 5, 0), pcm16(5000, 44100, 5, 1)], 1).astype(np.int64)
    for bs in (192, 576, 1152, 256, 1024, 1000):
        dec = oracle.flac(oracle.gen_flac(pcm.ravel(), 2, 16, 44100, bs))
        got = np.stack([np.round(dec.data[c] * 65536) for c in range(2)], 1).astype(np.int64)
        assert np.array_equal(got, pcm), bs


# ---- QOA: straight decoder from the specification (int32 arithmetic) ----
_QOA_DEQUANT = [[1, -1, 3, -3, 5, -5, 7, -7], [5, -5, 18, -18, 32, -32, 49, -49], [16, -16, 53, -53, 95, -95, 147, -147],
                [34, -34, 113, -113, 203, -203, 315, -315], [63, -63, 210, -210, 378, -378, 588, -588],
                [104, -104, 345, -345, 621, -621, 966, -966], [158, -158, 528, -528, 950, -950, 1477, -1477],
                [228, -228, 760, -760, 1368, -1368, 2128, -2128], [316, -316, 1053, -1053, 1895, -1895, 2947, -2947],
                [422, -422, 1405, -1405, 2529, -2529, 3934, -3934], [548, -548, 1828, -1828, 3290, -3290, 5117, -5117],
                [696, -696, 2320, -2320, 4176, -4176, 6496, -6496], [868, -868, 2893, -2893, 5207, -5207, 8099, -8099],
                [1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933], [1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005],
                [1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336]]


def _qoa_spec_decode(data):
    import struct
    assert data[:4] == b"qoaf"
    pos, out = 8, None
    while pos + 8 <= len(data):
        ch, = struct.unpack(">B", data[pos:pos + 1])
        samples, fsize = struct.unpack(">HH", data[pos + 4:pos + 8])
        if pos + fsize > len(data) or ch == 0:
            break
        if out is None:
            out = [[] for _ in range(ch)]
        p = pos + 8
        lms = []
        for _ in range(ch):
            h = list(struct.unpack(">4h", data[p:p + 8]))
            w = list(struct.unpack(">4h", data[p + 8:p + 16]))
            lms.append((h, w))
            p += 16
        frame = [[] for _ in range(ch)]
        for s0 in range(0, samples, 20):
            for c in range(ch):
                sl, = struct.unpack(">Q", data[p:p + 8])
                p += 8
                sf = sl >> 60
                h, w = lms[c]
                for k in range(min(20, samples - s0)):
                    q = (sl >> (57 - 3 * k)) & 7
                    pred = sum(a * b for a, b in zip(h, w)) >> 13
                    deq = _QOA_DEQUANT[sf][q]
                    rec = max(-32768, min(32767, pred + deq))
                    frame[c].append(rec)
                    d = deq >> 4
                    for i in range(4):
                        w[i] += -d if h[i] < 0 else d
                    h[:] = h[1:] + [rec]
        for c in range(ch):
            out[c] += frame[c]
        pos += fsize
    return out


@pytest.mark.parametrize("ch", [1, 2])
def test_qoa_matches_spec_decoder(oracle, ch):
    pcm = np.stack([pcm16(5120 * 2 + 20 * 17, 44100, 8, c) for c in range(ch)], 1)
    data = oracle.gen_qoa(pcm.ravel(), ch, 44100) + b"\0" * 8  # trailing bytes keep aukit.qoa's last frame (Q18)
    ref = _qoa_spec_decode(data)
    got = oracle.qoa(data)
    for c in range(ch):
        r = np.array(ref[c], dtype=np.float64)
        exp = np.where(r < 0, r / 32768, r / 32767)
        assert np.array_equal(got.data[c], exp)
    # the stream decoder emits floor(reconstructed / 256) of the same integers
    s = oracle.stream_qoa(data[:-8], False, oracle.NONE)
    assert s.nchunks >= 1


def test_qoa_last_frame_is_dropped_without_trailing_bytes(oracle):
    """aukit.lua:1735 compares frame_size with the bytes AFTER the 8-byte frame header, so the final frame never passes."""
    pcm = pcm16(5120 * 3, 44100, 8, 0)
    data = oracle.gen_qoa(pcm, 1, 44100)
    assert len(oracle.qoa(data).data[0]) == 5120 * 2
    assert len(oracle.qoa(data + b"\0" * 8).data[0]) == 5120 * 3


def test_dfpwm_roundtrip_and_state(oracle):
    x = np.round(signal(48000, 48000, 4, 0) * 100)
    enc = oracle.dfpwm_encode(x)
    assert len(enc) == 6000
    dec = oracle.DfpwmDecoder()(enc)
    assert np.sqrt(np.mean((dec - x) ** 2)) < 25           # lossy 1-bit codec tracks the input
    d2 = oracle.DfpwmDecoder()
    assert np.array_equal(np.concatenate([d2(enc[:1000]), d2(enc[1000:])]), dec)  # decoder state carries across calls
    e2 = oracle.DfpwmEncoder()
    assert e2(x[:8000]) + e2(x[8000:]) == enc               # encoder state too
    assert oracle.dfpwm_encode(np.zeros(3)) != b""         # a partial byte is padded with zero samples
    with pytest.raises(oracle.OracleError):
        oracle.dfpwm_encode(np.array([0.0, 200.0]))


def test_dfpwm_first_bits_by_hand(oracle):
    """DFPWM1a from charge = 0, strength = 0: bit 1 → target 127: next = 0 + floor((0·127 + 512)/1024) = 0 → nudged to 1;
    strength 0 → z = 0 (bit ≠ prev=false) → stays 0 → floor 8.  Decoder: antijerk (bit ≠ prevbit=false): floor((1+0+1)/2) = 1;
    lpf += floor((1·140 + 128)/256) = 1."""
    out = oracle.DfpwmDecoder()(bytes([0x01]))
    assert out[0] == 1
    # second bit 0: target -128, charge 1, strength 8: next = 1 + floor((8·(-129) + 512)/1024) = 1 + floor(-520/1024) = 0
    # antijerk: bit ≠ prev → floor((0 + 1 + 1)/2) = 1; lpf = 1 + floor(((1-1)·140 + 128)/256) = 1
    assert out[1] == 1


# ---- quirk micro-vectors ----
def test_q10_dfpwm_6001_byte_slices(oracle):
    data = bytes(range(256)) * 24  # 6144 bytes → slices [0,6001) and [6000,6144): byte 6000 is decoded twice
    a = oracle.dfpwm(data, 1, 48000)
    assert len(a.data[0]) == (6001 + 144) * 8
    d = oracle.DfpwmDecoder()
    exp = np.concatenate([d(data[:6001]), d(data[6000:])]).astype(np.float64)
    assert np.array_equal(a.data[0], np.where(exp < 0, exp / 128, exp / 127))


def test_q1_q2_stream_pcm_rebase_and_fir(oracle):
    rate = 44100
    x = pcm16(rate * 3, rate, 1, 0)
    s = oracle.stream_pcm(x.tobytes(), 16, oracle.SIGNED, 1, rate, False, False, oracle.LINEAR)
    assert list(s.chunk_len[:2, 0]) == [48000, 48000]
    v = np.where(x < 0, x / 32768.0, x / 32767.0)
    alpha = 1 - np.exp(-(rate / 96000) * 2 * np.pi)
    # first output of every chunk is alpha * s (ls restarts at 0); chunk c starts at input offset c * 44101 (linear)
    for c in range(2):
        ns = alpha * v[c * 44101]
        exp = max(-128.0, min(127.0, ns * (128 if ns < 0 else 127)))
        assert s.data[0][c * 48000] == exp
    sc = oracle.stream_pcm(x.tobytes(), 16, oracle.SIGNED, 1, rate, False, False, oracle.CUBIC)
    ns = alpha * v[1]  # cubic puts the FIRST sample at index 0, so output 1 is the 2nd sample
    assert sc.data[0][0] == ns * (128 if ns < 0 else 127)
    ns = alpha * v[44102 + 1]  # and re-bases by K = 44102
    assert sc.data[0][48000] == ns * (128 if ns < 0 else 127)


def test_q4_unsigned_normalisation(oracle):
    a = oracle.pcm(bytes([0, 127, 128, 255]), 8, oracle.UNSIGNED, 1, 8000)
    assert np.array_equal(a.data[0], np.array([-128 / 128, -1 / 128, 0 / 127, 127 / 127]))
    b = oracle.pcm(np.array([0, 200, 65535], dtype="<u2").tobytes(), 16, oracle.UNSIGNED, 1, 8000)
    assert np.array_equal(b.data[0], np.array([(0 - 128) / 32768, (200 - 128) / 32767, (65535 - 128) / 32767]))  # only right for 8 bit


def test_q5_ima_nibble_expansion(oracle):
    # predictor 0, step index 0 (step 7): nibble 7 → diff = (7*7 >> 2) + (7 >> 3) = 12 (standard IMA would give 11)
    a = oracle.adpcm(bytes([0x70]), 1, 8000, True, True, [0], [0])
    assert a.data[0][0] == 12 / 32767
    # nibble 0xF → -12, index moved 0 → 8 (step 16) by the first nibble: diff = (7*16 >> 2) + 2 = 30
    a = oracle.adpcm(bytes([0x7F]), 1, 8000, True, True, [0], [0])
    assert a.data[0][1] == (12 - 30) / 32768


def test_q6_stream_adpcm_junk_word_and_dropped_word(oracle):
    s = oracle.gen_ima(pcm16(1016 * 2, 22050, 3, 0), 1, 512, 88)
    r = oracle.stream_adpcm(s, 512, 1, 22050, False, oracle.CUBIC)
    assert int(r.chunk_len.sum()) == 2211 + int(np.floor(1008 * 48000 / 22050))  # block 1: 1016 real + 8 junk; block 2 loses its last word
    r1 = oracle.stream_adpcm(s[:512], 512, 1, 22050, False, oracle.CUBIC)
    n = int(np.floor(1008 * 48000 / 22050))
    # the junk nibbles only influence the interpolation look-ahead at the very end of block 1
    assert np.array_equal(r.data[0][:2000], oracle.stream_adpcm(s[:512] + s[512:516] + b"\0" * 508, 512, 1, 22050, False, oracle.CUBIC).data[0][:2000])
    assert len(r1.data[0]) == n


def test_q9_msadpcm_mono_reads_first_header_for_every_block(oracle):
    x = pcm16(1012 * 2, 44100, 6, 0)
    s = bytearray(oracle.gen_msadpcm(x, 1, 512))
    a = oracle.msadpcm(bytes(s), 512, 1, 44100)
    s2 = bytearray(s)
    s2[512:519] = b"\x06\x10\x00\x11\x11\x22\x22"  # rewrite block 2's header: ignored by the reference
    assert np.array_equal(oracle.msadpcm(bytes(s2), 512, 1, 44100).data[0], a.data[0])
    assert a.data[0][1012] == a.data[0][0] and a.data[0][1013] == a.data[0][1]  # block 2 starts from block 1's header samples


def test_q13_stream_g711_chunks_are_independent(oracle):
    g = oracle.gen_g711(pcm16(16000, 8000, 2, 0), True)
    full = oracle.stream_g711(g, True, 1, 8000, False, oracle.CUBIC)
    second = oracle.stream_g711(g[8000:], True, 1, 8000, False, oracle.CUBIC)
    assert np.array_equal(full.data[0][48000:], second.data[0])
    extra = oracle.stream_g711(g, True, 1, 8000, False, oracle.CUBIC, max_calls=4)
    assert list(extra.chunk_len[:, 0]) == [48000, 48000, 0, 0]  # never returns nil: empty chunks forever


def test_q14_flac_normalised_by_2_pow_depth(oracle):
    pcm = np.array([[32767, -32768], [1, -1]], dtype=np.int64)
    dec = oracle.flac(oracle.gen_flac(pcm.ravel(), 2, 16, 44100, 4096))
    assert dec.data[0][0] == 32767 / 65536 and dec.data[1][0] == -32768 / 65536


def test_q16_resample_integer_positions_copy_unclamped(oracle):
    a = oracle.Audio([np.array([0.0, 2.0, -3.0, 0.5])], 24000)
    r = oracle.resample(a, 48000, oracle.LINEAR)
    assert list(r.data[0][[0, 2, 4, 6]]) == [0.0, 2.0, -3.0, 0.5]  # x integer → copied as is
    assert r.data[0][1] == 1.0 and r.data[0][3] == -0.5             # interpolated → clamped to ±1


def test_q17_effects_that_raise(oracle):
    a = oracle.Audio([signal(1000, 8000, 1, 0)], 8000)
    with pytest.raises(oracle.OracleError):
        oracle.fx_trim(a)
    with pytest.raises(oracle.OracleError):
        oracle.fx_fade(a, 0.0, 1.0, 0.01, 0.0)
    z = oracle.fx_normalize(oracle.Audio([np.zeros(4)], 8000), 1.0)
    assert np.all(np.isnan(z.data[0]))  # peak / 0 = inf; 0 * inf = nan


def test_effects_against_direct_numpy(oracle):
    x = signal(5000, 22050, 7, 0)
    a = 1 - np.exp(-(3000 / 22050) * 2 * np.pi)
    y = x.copy()
    for i in range(1, len(y)):
        y[i] = y[i - 1] + a * (y[i] - y[i - 1])
    assert np.array_equal(oracle.fx_lowpass(oracle.Audio([x], 22050), 3000.0).data[0], y)
    ah = 1 / (2 * np.pi * (20 / 22050) + 1)
    z = x.copy()
    lx = z[0]
    for i in range(1, len(z)):
        llx = z[i]
        z[i] = ah * (z[i - 1] + llx - lx)
        lx = llx
    assert np.array_equal(oracle.fx_highpass(oracle.Audio([x], 22050), 20.0).data[0], z)


def test_cubic_interpolation_by_hand(oracle):
    d = np.array([0.0, 1.0, 4.0, 9.0, 16.0])
    # x = 2.5: p0..p3 = d[1..4]
    p0, p1, p2, p3, fx = 0.0, 1.0, 4.0, 9.0, 0.5
    exp = (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1
    assert oracle.interp(oracle.CUBIC, d, 2.5) == exp
    assert oracle.interp(oracle.CUBIC, d, 1.5) == (-0.5 * 0 + 1.5 * 0 - 1.5 * 1 + 0.5 * 4) * 0.125 + (0 - 0 + 2 - 2) * 0.25 + (0 + 0.5) * 0.5 + 0  # p0 := p1 at the left edge
    assert oracle.interp(oracle.LINEAR, d, 5.25) == 16.0  # data[ffx+1] or data[ffx]


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_q14_stream_flac_history_is_shared_across_channels_and_one_sample_blocks(oracle, interp):
    """stream.flac (aukit.lua:3157-3188) transliterated line by line in Python and run next to the C oracle on a stereo file whose
    last frame holds ONE sample: `last` is shared by the channels (channel 2 of a frame sees channel 1's tail) and for a block with
    #src == 1, `last = {src[#src-1], src[#src]}` picks up the block's own injected src[0].  (The random sweeps on the GPU found the
    HIP path wrong exactly here; this pins the oracle it was checked against.)"""
    import math
    rng = np.random.Generator(np.random.PCG64(77))
    rate, bs, ch, depth = 22050, 192, 2, 16
    n = 3 * bs + 1
    x = rng.integers(-20000, 20000, (n, ch)).astype(np.int64)
    ref = oracle.stream_flac(oracle.gen_flac(x.ravel(), ch, depth, rate, bs), oracle.INTERP[interp])

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def interpolate(src, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "linear":
            a, b = src.get(ffx), src.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx                                     # :257-260
        p0, p1, p2, p3 = src.get(ffx - 1), src.get(ffx), src.get(ffx + 1), src.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1  # :261-266

    ratio = 48000 / rate
    alpha = 1 - math.exp(-(rate / 96000) * 2 * math.pi)
    last, out, pos = [0, 0], [[] for _ in range(ch)], 0
    while pos < n:
        nfr = min(bs, n - pos)
        for c in range(ch):
            src = {i + 1: float(x[pos + i, c]) / (1 << depth) for i in range(nfr)}                   # :505 (Q14)
            src[0], src[-1] = last[1], last[0]                                                       # :3170-3171
            ls = last[1] / (128 if last[1] < 0 else 127)
            for i in range(1, math.floor(nfr * ratio) + 1):
                xx = ((i - 1) / ratio) + 1
                s = src[int(xx)] if xx % 1 == 0 else interpolate(src, xx)
                s = ls + alpha * (s - ls)
                ls = s
                out[c].append(clamp(s * (128 if s < 0 else 127), -128, 127))
            last = [src[nfr - 1], src[nfr]]                                                          # :3183
        pos += nfr
    for c in range(ch):
        assert len(out[c]) == len(ref.data[c])
        assert np.max(np.abs(np.array(out[c]) - ref.data[c])) <= 1e-12


_IMA_INDEX = [-1, -1, -1, -1, 2, 4, 6, 8] * 2
_IMA_STEP = [7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143, 157, 173, 190, 209,
             230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749,
             3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385,
             24623, 27086, 29794, 32767]  # the standard IMA tables (aukit.lua:156-171)


@pytest.mark.parametrize("ch,ba,mono,interp", [(1, 36, False, "cubic"), (1, 68, False, "linear"), (2, 72, False, "cubic"), (2, 72, True, "linear"), (2, 136, True, "cubic")])
def test_q6_stream_adpcm_transliterated(oracle, ch, ba, mono, interp):
    """stream.adpcm in string mode (aukit.lua:2771-2831) transliterated line by line — inclusive word loop (junk word after every
    non-final block, last word of the final block dropped), history written to the outer table (no effect), upvalue `newlen` that
    stays shrunk, unmasked header index, Q5 nibble expansion, Q7 scaling — next to the C oracle, chunk for chunk."""
    import math
    import struct as st
    rng = np.random.Generator(np.random.PCG64(ba + ch))
    rate = 22050
    spb = (ba - 4 * ch) * 2 // ch
    pcm = rng.integers(-9000, 9000, spb * 9 * ch).astype(np.int16)
    data = oracle.gen_ima(pcm, ch, ba, 88)
    data = data[: ba * 8 + ba // 2]  # a short final block
    ref = oracle.stream_adpcm(data, ba, ch, rate, mono, oracle.INTERP[interp])

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def length(t):  # Lua's # on a table filled from 1 upwards
        k = 0
        while (k + 1) in t:
            k += 1
        return k

    def interpolate(t, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "linear":
            a, b = t.get(ffx), t.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx
        p0, p1, p2, p3 = t.get(ffx - 1), t.get(ffx), t.get(ffx + 1), t.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    ratio = 48000 / rate
    n = 1
    samples_per_block = (ba - 4 * ch) * 2 / ch
    bytes_per_second = ba * math.ceil(rate / samples_per_block)
    newlen = math.floor(samples_per_block * ratio)
    chunks = []
    while True:
        target = n + bytes_per_second
        retval = [[] for _ in range(1 if mono else ch)]
        while n < target:
            if n + ch * 4 > len(data):
                break
            d = [dict() for _ in range(ch)]
            predictor = [st.unpack_from("<h", data, n - 1 + i * 4)[0] for i in range(ch)]
            step_index = [data[n - 1 + i * 4 + 2] for i in range(ch)]
            i = ch * 4
            while i <= ba:
                p = (i - ch * 4) // ch * 2 + 1
                if len(data) < n + i + ch * 4:
                    break
                for j in range(ch):
                    num = st.unpack_from("<I", data, n - 1 + i + j * 4)[0]
                    for k in range(8):
                        nibble = (num >> (4 * k)) & 15
                        step = _IMA_STEP[step_index[j]]
                        step_index[j] = clamp(step_index[j] + _IMA_INDEX[nibble], 0, 88)
                        diff = (((nibble % 8) * step) >> 2) + (step >> 3)
                        predictor[j] = clamp(predictor[j] - diff, -32768, 32767) if nibble >= 8 else clamp(predictor[j] + diff, -32768, 32767)
                        d[j][p + k] = predictor[j] / (128 if predictor[j] < 0 else 127)
                i += ch * 4
            if length(d[0]) < samples_per_block:
                newlen = math.floor(length(d[0]) * ratio)
            for i in range(1, newlen + 1):
                xx = (i - 1) / ratio + 1
                c = [d[j][int(xx)] if xx % 1 == 0 else interpolate(d[j], xx) for j in range(ch)]
                if mono:
                    acc = 0
                    for j in range(ch):
                        acc = acc + c[j]
                    retval[0].append(clamp(math.floor(acc / ch), -128, 127))
                else:
                    for j in range(ch):
                        retval[j].append(clamp(math.floor(c[j]), -128, 127))
            n += ba
        if not retval[0]:
            break
        chunks.append(retval)
    assert len(chunks) == ref.nchunks
    assert [len(c[0]) for c in chunks] == list(ref.chunk_len[:, 0])
    for c in range(ref.channels):
        assert np.array_equal(np.concatenate([np.array(k[c], dtype=np.float64) for k in chunks]), ref.data[c]), c


@pytest.mark.parametrize("ch,mono,interp,rate", [(1, False, "cubic", 44100), (1, False, "linear", 8000), (2, False, "cubic", 22050), (2, True, "linear", 44100), (2, True, "cubic", 32000)])
def test_q1_q2_q3_stream_pcm_transliterated(oracle, ch, mono, interp, rate):
    """aukit.stream.pcm on a 16-bit signed string (aukit.lua:2228-2424) transliterated line by line: the lazy `__index` tables that
    hand out the NEXT stream sample whatever index is asked (Q3), the eager prefill, the 2-tap low-pass on the raw previous sample
    with `ls` restarting at 0 (Q2), the window re-base that keeps d[-1], d[0] (Q1), end of data as an error inside pcall (truncated
    last chunk) or in the prefill (the iterator raises) — next to the C oracle, chunk for chunk."""
    import math
    rng = np.random.Generator(np.random.PCG64(rate + ch))
    nfr = rate * 2 + 100
    pcm = rng.integers(-32768, 32768, nfr * ch).astype("<i2")
    ref = oracle.stream_pcm(pcm.tobytes(), 16, oracle.SIGNED, ch, rate, False, mono, oracle.INTERP[interp])

    class LuaError(Exception):
        pass

    state = {"pos": 0}

    def read():
        if state["pos"] >= len(pcm):
            raise LuaError("attempt to compare nil with number")  # s = tmp[pos] is nil, `s < 0` raises
        s = int(pcm[state["pos"]])
        state["pos"] += 1
        return s / (32768 if s < 0 else 32767)

    if ch == 1:
        mono = False  # :2243

    class Lazy(dict):
        def __missing__(self, i):  # the __index metamethod :2367-2371
            if mono:
                v = 0
                for _ in range(ch):
                    v = v + read()
                v = v / ch
            else:
                v = read()
            self[i] = v
            return v

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def interpolate(t, xx):
        ffx = math.floor(xx)
        if interp == "linear":
            a = t[ffx]
            b = t[ffx + 1]
            return a + (b - a) * (xx - ffx)
        p0, p1, p2, p3, fx = t[ffx - 1], t[ffx], t[ffx + 1], t[ffx + 2], xx - ffx  # in this order: each miss reads the stream
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    def length(t):
        k = 0
        while (k + 1) in t:
            k += 1
        return k

    nd = 1 if mono else ch
    d = [Lazy() for _ in range(nd)]
    ratio = 48000 / rate
    alpha = 1 - math.exp(-(rate / 96000) * 2 * math.pi)
    istart, iend = {"linear": (1, 2), "cubic": (0, 3)}[interp]
    n, ok, chunks, raised = 0, True, [], False
    while ok:
        try:
            for i in range(istart if n == 0 else 1, iend + 1):  # the prefill :2376-2386 (outside pcall)
                if mono:
                    s = 0
                    for _ in range(ch):
                        s = s + read()
                    dict.__setitem__(d[0], i, s / ch)
                else:
                    for j in range(nd):
                        dict.__setitem__(d[j], i, read())
        except LuaError:
            raised = True
            break
        chunk = [dict() for _ in range(nd)]
        try:
            ls = [0.0] * nd  # chunk[y][0] is always nil
            for i in range(1, 48001):
                for y in range(nd):
                    xx = ((i - 1) / ratio) + 1
                    s = d[y][int(xx)] if xx % 1 == 0 else interpolate(d[y], xx)
                    ns = ls[y] + alpha * (s - ls[y])
                    chunk[y][i] = clamp(ns * (128 if ns < 0 else 127), -128, 127)
                    ls[y] = s
        except LuaError:
            ok = False
        if length(chunk[0]) == 0:
            break
        n += length(chunk[0])
        for y in range(nd):
            l = length(d[y])
            l2, l1 = d[y].get(l - 1), d[y].get(l)
            d[y] = Lazy()
            dict.__setitem__(d[y], -1, l2)
            dict.__setitem__(d[y], 0, l1)
        chunks.append(chunk)
    assert len(chunks) == ref.nchunks
    assert [length(c[0]) for c in chunks] == list(ref.chunk_len[:, 0])
    assert raised == (ref.final_status != 0)
    for y in range(nd):
        got = np.array([c[y][i] for c in chunks for i in range(1, length(c[y]) + 1)])
        assert np.max(np.abs(got - ref.data[y][:len(got)])) <= 1e-12, y


@pytest.mark.parametrize("ch,mono,ulaw,interp,rate", [(1, False, True, "cubic", 8000), (1, False, False, "linear", 11025), (2, True, True, "cubic", 8000), (3, False, False, "linear", 16000), (2, True, False, "none", 8000)])
def test_q13_stream_g711_transliterated(oracle, ch, mono, ulaw, interp, rate):
    """aukit.stream.g711 on a string (aukit.lua:2863-2911) transliterated: per-call slices of sampleRate * channels bytes, the G.711
    expansion with its ±0x40 divisor, independent chunks (the history copy lands in the outer table), floor + clamp, mono mean."""
    import math
    rng = np.random.Generator(np.random.PCG64(rate + 10 * ch))
    data = bytes(rng.integers(0, 256, rate * ch * 2 + 37 * ch, dtype=np.uint8))
    ref = oracle.stream_g711(data, ulaw, ch, rate, mono, oracle.INTERP[interp])

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def interpolate(t, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "none":
            return t.get(ffx)
        if interp == "linear":
            a, b = t.get(ffx), t.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx
        p0, p1, p2, p3 = t.get(ffx - 1), t.get(ffx), t.get(ffx + 1), t.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    xor = 0xFF if ulaw else 0x55
    ratio = 48000 / rate
    pos, outs = 1, [[] for _ in range(1 if mono else ch)]
    lens = []
    for _ in range(ref.nchunks):
        sl = data[pos - 1: pos - 1 + rate * ch]
        pos += rate * ch
        tabs = [dict() for _ in range(ch)]
        for k, byte in enumerate(sl):  # k = i + j - 2
            b = byte ^ xor
            m, e = b & 0x0F, (b >> 4) & 7
            m = m * 4 + 2 if (not ulaw and e == 0) else (m * 2 + 33) << e
            if ulaw:
                m -= 33
            tabs[k % ch][k // ch + 1] = m / (-0x40 if (bool(b & 0x80) == ulaw) else 0x40)
        n1 = 0
        while (n1 + 1) in tabs[0]:
            n1 += 1
        newlen = math.floor(n1 * ratio)
        lens.append(newlen)
        for i in range(1, newlen + 1):
            xx = (i - 1) / ratio + 1
            c = [tabs[j][int(xx)] if xx % 1 == 0 else interpolate(tabs[j], xx) for j in range(ch)]
            if mono:
                acc = 0
                for j in range(ch):
                    acc = acc + c[j]
                outs[0].append(clamp(math.floor(acc / ch), -128, 127))
            else:
                for j in range(ch):
                    outs[j].append(clamp(math.floor(c[j]), -128, 127))
    assert lens == list(ref.chunk_len[:, 0])
    for c in range(len(outs)):
        assert np.array_equal(np.array(outs[c], dtype=np.float64), ref.data[c]), c


@pytest.mark.parametrize("ch,mono,interp,rate", [(1, False, "linear", 48000), (1, False, "cubic", 24000), (2, False, "linear", 44100), (2, True, "cubic", 32000), (3, True, "linear", 48000)])
def test_q10_q11_stream_dfpwm_transliterated(oracle, ch, mono, interp, rate):
    """aukit.stream.dfpwm on a string (aukit.lua:2446-2493) transliterated around the oracle's decoder object (the codec arithmetic
    itself is the unpinned part): slices of 6000 * channels + 1 bytes advanced by 6000 * channels (Q10), audio[0] = the previous
    slice's last sample, `x` that ignores the channel index and the stride of `channels` over a fractional `newlen` (Q11), clamp on
    the interpolated branch only, mono mean."""
    import math
    rng = np.random.Generator(np.random.PCG64(rate + ch))
    data = bytes(rng.integers(0, 256, 6000 * ch * 2 + 777 * ch, dtype=np.uint8))
    ref = oracle.stream_dfpwm(data, rate, ch, mono, oracle.INTERP[interp])
    if ch == 1:
        mono = False

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def interpolate(t, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "linear":
            a, b = t.get(ffx), t.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx
        p0, p1, p2, p3 = t.get(ffx - 1), t.get(ffx), t.get(ffx + 1), t.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    dec = oracle.DfpwmDecoder()
    pos, last, chunks, positions = 1, 0, [], []
    while pos <= len(data):
        d = data[pos - 1: pos + 6000 * ch]
        a = dec(d)
        if len(a) == 0:
            break
        audio = {i + 1: float(v) for i, v in enumerate(a)}
        audio[0], last = last, audio[len(a)]
        ratio = 48000 / rate
        newlen = len(a) * ratio
        lines = [dict() for _ in range(1 if mono else ch)]
        i = 1
        while i <= newlen:
            n = 0
            for j in range(ch):
                xx = (i - 1) / ratio + 1
                s = audio[int(xx)] if xx % 1 == 0 else clamp(interpolate(audio, xx), -128, 127)
                if mono:
                    n = n + s
                else:
                    lines[j][math.ceil(i / ch)] = s
            if mono:
                lines[0][math.ceil(i / ch)] = n / ch
            i += ch
        chunks.append(lines)
        positions.append(pos * 8 / rate / ch)
        pos += 6000 * ch
    assert len(chunks) == ref.nchunks
    assert np.array_equal(np.array(positions), ref.chunk_pos)
    for c in range(ref.channels):
        got = np.array([k[c][i] for k in chunks for i in range(1, len(k[c]) + 1)])
        assert [len(k[c]) for k in chunks] == list(ref.chunk_len[:, 0])
        assert np.max(np.abs(got - ref.data[c]), initial=0) <= 1e-12, c


def test_dfpwm_codec_against_the_published_text(oracle):
    """DFPWM1a written out from the published description (PREC = 10, strength floor 8, anti-jerk, low-pass 140/256;
    encoder bit = v > charge or (v == charge and v == 127)) in plain Python, next to the C oracle's decoder and encoder on random
    bytes / random samples — the arithmetic stays "parity unpinned" (no reference copy of cc.audio.dfpwm exists here), but the three
    statements of it (this one, ork_codecs.c, dfpwm_dev.h via tests/test_host_math.py) agree."""
    rng = np.random.Generator(np.random.PCG64(4242))

    class Pred:
        def __init__(self):
            self.charge, self.strength, self.prev = 0, 0, False

        def step(self, bit):
            target = 127 if bit else -128
            nxt = self.charge + ((self.strength * (target - self.charge) + 512) >> 10)
            if nxt == self.charge and nxt != target:
                nxt += 1 if bit else -1
            z = 1023 if bit == self.prev else 0
            if self.strength != z:
                self.strength += 1 if bit == self.prev else -1
            if self.strength < 8:
                self.strength = 8
            self.charge, self.prev = nxt, bit
            return nxt

    data = bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + b"\xff" * 40 + b"\x00" * 40 + b"\xaa" * 40
    p, lpf, pcharge, pbit, out = Pred(), 0, 0, False, []
    for byte in data:
        for k in range(8):
            bit = bool((byte >> k) & 1)
            charge = p.step(bit)
            aj = (charge + pcharge + 1) >> 1 if bit != pbit else charge
            pcharge, pbit = charge, bit
            lpf += ((aj - lpf) * 140 + 0x80) >> 8
            out.append(lpf)
    assert np.array_equal(np.array(out, dtype=np.int8), oracle.DfpwmDecoder()(data))

    samples = np.concatenate([rng.integers(-128, 128, 8000), np.full(200, 127), np.full(200, -128), np.zeros(200, dtype=np.int64)])
    e, bits = Pred(), []
    for v in samples:
        bit = bool(v > e.charge or (v == e.charge and v == 127))
        e.step(bit)
        bits.append(bit)
    packed = bytes(sum(int(b) << k for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))
    assert packed == oracle.dfpwm_encode(samples.astype(np.float64))


@pytest.mark.parametrize("ch,mono,interp,tail", [(1, False, "linear", 0), (2, False, "cubic", 8), (2, True, "linear", 0), (1, False, "none", 8)])
def test_q15_stream_qoa_transliterated(oracle, ch, mono, interp, tail):
    """aukit.stream.qoa on a string (aukit.lua:3239-3336) transliterated, LMS included (weights are NOT wrapped to 16 bits, the sum
    wraps to int32 through bit32.arshift): frames are read until a call holds at least one second, every slice decodes 20 samples
    even when the frame's count is not a multiple of 20 (overwritten by the next frame or left as a tail), floor(reconstructed /
    256), interpolation clamp, recursive low-pass seeded with the previous call's last raw sample, position of the call."""
    import math
    import struct as st
    rng = np.random.Generator(np.random.PCG64(90 + ch))
    rate = 22050
    n = 5120 * 9 + 20 * 13 + 7  # the last frame's sample count is not a multiple of 20
    pcm = rng.integers(-12000, 12000, n * ch).astype(np.int16)
    data = oracle.gen_qoa(pcm, ch, rate) + b"\\0" * tail
    ref = oracle.stream_qoa(data, mono, oracle.INTERP[interp])

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def arsh32(a, b):  # signed_rshift: bit32.arshift on the value reduced to uint32, then back to a signed number
        a &= 0xFFFFFFFF
        if a & 0x80000000:
            a -= 0x100000000
        return a >> b

    def interpolate(t, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "none":
            return t.get(ffx)
        if interp == "linear":
            a, b = t.get(ffx), t.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx
        p0, p1, p2, p3 = t.get(ffx - 1), t.get(ffx), t.get(ffx + 1), t.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    def length(t):
        k = 0
        while (k + 1) in t:
            k += 1
        return k

    state = {"pos": 0}

    def read(k):
        if state["pos"] >= len(data):
            return None
        d = data[state["pos"]: state["pos"] + k]
        state["pos"] += k
        return d

    head = read(8)
    assert head[:4] == b"qoaf"
    fch, = st.unpack(">B", data[8:9])
    frate = int.from_bytes(data[9:12], "big")
    lms = [{"h": [0] * 4, "w": [0] * 4} for _ in range(fch)]
    last = [[0, 0] for _ in range(fch)]
    file_pos = 0
    ratio = 48000 / frate
    alpha = 1 - math.exp(-(frate / 96000) * 2 * math.pi)
    chunks, positions = [], []
    while True:
        chunk = [{-1: last[i][0], 0: last[i][1]} for i in range(fch)]
        sample_pos = 0
        while sample_pos < frate:
            d = read(8)
            if not d:
                break
            if len(d) < 8:
                raise AssertionError("short frame header")
            channels = d[0]
            samplerate = int.from_bytes(d[1:4], "big")
            samples, frame_size = st.unpack(">HH", d[4:8])
            data_size = frame_size - 8 - 4 * 4 * channels
            max_total = (data_size // 8) * 20
            if channels != fch or samplerate != frate or samples * channels > max_total:
                break
            for c in range(channels):
                lms[c]["h"] = list(st.unpack(">4h", read(8)))
                lms[c]["w"] = list(st.unpack(">4h", read(8)))
            for sample_index in range(1, samples + 1, 20):
                for c in range(channels):
                    hi, lo = st.unpack(">II", read(8))
                    sf = (hi >> 28) & 15
                    for si in range(sample_index, sample_index + 20):
                        w, h = lms[c]["w"], lms[c]["h"]
                        predicted = arsh32(w[0] * h[0] + w[1] * h[1] + w[2] * h[2] + w[3] * h[3], 13)
                        quantized = (hi >> 25) & 7
                        deq = _QOA_DEQUANT[sf][quantized]
                        rec = min(max(predicted + deq, -32768), 32767)
                        chunk[c][sample_pos + si] = math.floor(rec / 256)
                        hi = ((hi << 3) & 0xFFFFFFFF) + ((lo >> 29) & 7)
                        lo = (lo << 3) & 0xFFFFFFFF
                        delta = arsh32(deq, 4)
                        lms[c]["w"] = [w[i] + (-delta if h[i] < 0 else delta) for i in range(4)]
                        lms[c]["h"] = [h[1], h[2], h[3], rec]
            sample_pos += samples
        n1 = length(chunk[0])
        if n1 == 0:
            break
        newlen = n1 * ratio
        lines = [[] for _ in range(1 if mono else fch)]
        ls = [last[j][1] for j in range(fch)]
        i = 1
        while i <= newlen:
            acc = 0
            for j in range(fch):
                xx = (i - 1) / ratio + 1
                s = chunk[j][int(xx)] if xx % 1 == 0 else clamp(interpolate(chunk[j], xx), -128, 127)
                s = ls[j] + alpha * (s - ls[j])
                ls[j] = s
                if mono:
                    acc = acc + s
                else:
                    lines[j].append(s)
            if mono:
                lines[0].append(acc / fch)
            i += 1
        positions.append(file_pos / frate)
        file_pos += sample_pos
        for j in range(fch):
            l = length(chunk[j])
            last[j] = [chunk[j].get(l - 1), chunk[j].get(l)]
        chunks.append(lines)
    assert len(chunks) == ref.nchunks
    assert [len(c[0]) for c in chunks] == list(ref.chunk_len[:, 0])
    assert np.array_equal(np.array(positions), ref.chunk_pos)
    for c in range(ref.channels):
        got = np.concatenate([np.array(k[c], dtype=np.float64) for k in chunks])
        assert np.max(np.abs(got - ref.data[c])) <= 1e-12, c


_MS_ADAPT = {0: 230, 1: 230, 2: 230, 3: 230, 4: 307, 5: 409, 6: 512, 7: 614, -8: 768, -7: 614, -6: 512, -5: 409, -4: 307, -3: 230, -2: 230, -1: 230}


@pytest.mark.parametrize("ch,mono,interp", [(1, False, "linear"), (1, False, "cubic"), (2, False, "cubic"), (2, True, "linear")])
def test_q9_stream_msadpcm_transliterated(oracle, ch, mono, interp):
    """aukit.stream.msadpcm on a string (aukit.lua:2613-2733) transliterated: the mono branch reads EVERY block header from the start
    of the data (`str_unpack("<!1Bhhh", data)` has no position), stereo floors each sample and mixes l + r / 2, mono does not floor
    before interpolating, samplesPerBlock leaves the block's last two samples as look-ahead only."""
    import math
    import struct as st
    rng = np.random.Generator(np.random.PCG64(60 + ch))
    ba, rate = 256 * ch, 22050
    spb = (ba - 14) + 2 if ch == 2 else (ba - 7) * 2 + 2
    pcm = rng.integers(-6000, 6000, spb * 40 * ch).astype(np.int16)
    data = oracle.gen_msadpcm(pcm, ch, ba)
    ref = oracle.stream_msadpcm(data, ba, ch, rate, mono, None, oracle.INTERP[interp])
    c1t, c2t = [256, 512, 0, 192, 240, 460, 392], [0, -256, 0, 64, 0, -208, -232]

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def interpolate(t, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        if interp == "linear":
            a, b = t.get(ffx), t.get(ffx + 1)
            return a + ((b if b is not None else a) - a) * fx
        p0, p1, p2, p3 = t.get(ffx - 1), t.get(ffx), t.get(ffx + 1), t.get(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    def nib(v):
        return v - 16 if v >= 8 else v

    def scale(p):
        return p / (128 if p < 0 else 127)

    ratio = 48000 / rate
    spb_ref = ba - 14 if ch == 2 else (ba - 7) * 2
    bytes_per_second = ba * math.ceil(rate / spb_ref)
    newlen = math.floor(spb_ref * ratio)
    n, chunks = 1, []
    while True:
        target = n + bytes_per_second
        retval = [[] for _ in range(1 if (mono or ch == 1) else 2)]
        while n < target:
            if n > len(data):
                break
            if ch == 2:
                piL, piR, dL, dR, s1L, s1R, s2L, s2R = st.unpack_from("<BBhhhhhh", data, n - 1)
                left = {1: math.floor(scale(s2L)), 2: math.floor(scale(s1L))}
                right = {1: math.floor(scale(s2R)), 2: math.floor(scale(s1R))}
                for i in range(14, ba):
                    b = data[n - 1 + i]
                    hi, lo = nib(b >> 4), nib(b & 15)
                    p = clamp(math.floor((s1L * c1t[piL] + s2L * c2t[piL]) / 256) + hi * dL, -32768, 32767)
                    left[len(left) + 1] = math.floor(scale(p))
                    s2L, s1L = s1L, p
                    dL = max(math.floor(_MS_ADAPT[hi] * dL / 256), 16)
                    p = clamp(math.floor((s1R * c1t[piR] + s2R * c2t[piR]) / 256) + lo * dR, -32768, 32767)
                    right[len(right) + 1] = math.floor(scale(p))
                    s2R, s1R = s1R, p
                    dR = max(math.floor(_MS_ADAPT[lo] * dR / 256), 16)
                for i in range(1, newlen + 1):
                    xx = (i - 1) / ratio + 1
                    l, r = (left[int(xx)], right[int(xx)]) if xx % 1 == 0 else (interpolate(left, xx), interpolate(right, xx))
                    if mono:
                        retval[0].append(clamp(math.floor(l + r / 2), -128, 127))
                    else:
                        retval[0].append(clamp(math.floor(l), -128, 127))
                        retval[1].append(clamp(math.floor(r), -128, 127))
            else:
                pi, dl, s1, s2 = st.unpack_from("<Bhhh", data, 0)  # always the FIRST block's header (Q9)
                left = {1: scale(s2), 2: scale(s1)}
                for i in range(7, ba):
                    b = data[n - 1 + i]
                    for v in (nib(b >> 4), nib(b & 15)):
                        p = clamp(math.floor((s1 * c1t[pi] + s2 * c2t[pi]) / 256) + v * dl, -32768, 32767)
                        left[len(left) + 1] = scale(p)
                        s2, s1 = s1, p
                        dl = max(math.floor(_MS_ADAPT[v] * dl / 256), 16)
                for i in range(1, newlen + 1):
                    xx = (i - 1) / ratio + 1
                    retval[0].append(clamp(math.floor(left[int(xx)] if xx % 1 == 0 else interpolate(left, xx)), -128, 127))
            n += ba
        if not retval[0]:
            break
        chunks.append(retval)
    assert len(chunks) == ref.nchunks
    assert [len(c[0]) for c in chunks] == list(ref.chunk_len[:, 0])
    for c in range(ref.channels):
        assert np.array_equal(np.concatenate([np.array(k[c], dtype=np.float64) for k in chunks]), ref.data[c]), c


def test_effects_transliterated(oracle):
    """amplify / fade / normalize / center / delay / echo / reverb (aukit.lua:3356-3580) as plain Python loops over 1-based lists next
    to the C oracle — the in-place all-pass of reverb reading `sum[i + 20 - samples]` after earlier entries were overwritten, and
    `o[1..samples]` left dry (Q17), included"""
    import math
    rng = np.random.Generator(np.random.PCG64(31))
    rate, n = 8000, 5000
    x = rng.uniform(-1, 1, n)

    def clamp(v):
        return -1 if v < -1 else (1 if v > 1 else v)

    def lua(a):  # 1-based
        return [None] + list(a)

    def arr(t):
        return np.array(t[1:], dtype=np.float64)

    A = lambda: oracle.Audio([x.copy()], rate)
    # amplify
    assert np.array_equal(oracle.fx_amplify(A(), 1.7).data[0], np.array([clamp(v * 1.7) for v in x]))
    # fade 0.1 s .. 0.4 s from 1 to 0.2
    ch = lua(x)
    start, m = 0.1 * rate, (0.2 - 1.0) / ((0.4 - 0.1) * rate)
    i = start
    while i <= 0.4 * rate:
        ch[int(i)] = clamp(ch[int(i)] * (m * (i - start) + 1.0))
        i += 1
    assert np.array_equal(oracle.fx_fade(A(), 0.1, 1.0, 0.4, 0.2).data[0], arr(ch))
    # normalize (not independent)
    mx = 0
    for v in x:
        mx = max(mx, abs(v))
    mult = 0.8 / mx
    assert np.array_equal(oracle.fx_normalize(A(), 0.8).data[0], np.array([clamp(v * mult) for v in x]))
    # center: per second of samples
    ch = lua(x)
    i = 0
    while i <= n - 1:
        l = min(n - i, rate)
        avg = 0
        for j in range(1, l + 1):
            avg = avg + ch[i + j]
        avg = avg / l
        for j in range(1, l + 1):
            ch[i + j] = clamp(ch[i + j] - avg)
        i += rate
    assert np.array_equal(oracle.fx_center(A()).data[0], arr(ch))
    # delay / echo
    samples = math.floor(0.0123 * rate)
    o = lua(x)
    orig = list(o)
    for i in range(samples + 1, n + 1):
        o[i] = clamp(o[i] + orig[i - samples] * 0.6)
    assert np.array_equal(oracle.fx_delay(A(), 0.0123, 0.6).data[0], arr(o))
    o = lua(x)
    for i in range(samples + 1, n + 1):
        o[i] = clamp(o[i] + o[i - samples] * 0.6)
    assert np.array_equal(oracle.fx_echo(A(), 0.0123, 0.6).data[0], arr(o))
    # reverb
    delay, decay, wet, dry = 40.0, 0.35, 0.9, 0.2
    o = lua(x)
    sm = {}
    for k, (ds, dc) in enumerate(zip([0, -11.73, 19.31, -7.97], [0, 0.1313, 0.2743, 0.31])):
        comb = {}
        samples = math.floor((delay + ds) / 1000 * rate)
        mul = decay - dc
        for i in range(1, min(samples, n) + 1):
            comb[i] = o[i]
            sm[i] = sm.get(i, 0) + o[i]
        for i in range(samples + 1, n + 1):
            s = o[i] + comb[i - samples] * mul
            comb[i] = s
            sm[i] = sm.get(i, 0) + s
    for i in range(1, n + 1):
        sm[i] = sm[i] * wet + o[i] * dry
    samples = math.floor(0.08927 * rate)
    sm[samples + 1] = sm[samples + 1] - 0.131 * sm[1]
    for i in range(samples + 2, n + 1):
        sm[i] = sm[i] - 0.131 * sm[i - samples] + 0.131 * sm[i + 20 - samples]
    o[samples + 1] = clamp(sm[samples + 1] - 0.131 * sm[1])
    for i in range(samples + 2, n + 1):
        o[i] = clamp(sm[i] - 0.131 * sm[i - samples] + 0.131 * sm[i + 20 - samples])
    assert np.max(np.abs(oracle.fx_reverb(A(), delay, decay, wet, dry).data[0] - arr(o))) <= 1e-15


def test_audio_methods_transliterated(oracle):
    """Audio:resample (fractional `newlen` as the loop bound, copy at integer positions, clamp elsewhere), Audio:mono, Audio:mix
    (zero padding, missing channels, clamp of sum * amplifier) and Audio:pcm / encodePCM (interleaved and channel after channel)
    as plain Python next to the C oracle (aukit.lua:653-689, :804-835, :868-910)"""
    import math
    rng = np.random.Generator(np.random.PCG64(41))

    def clamp(v):
        return -1 if v < -1 else (1 if v > 1 else v)

    def cubic(c, xx):
        ffx = math.floor(xx)
        fx = xx - ffx
        g = lambda i: c[i - 1] if 1 <= i <= len(c) else None
        p0, p1, p2, p3 = g(ffx - 1), g(ffx), g(ffx + 1), g(ffx + 2)
        p0 = p1 if p0 is None else p0
        p2 = p1 if p2 is None else p2
        p3 = p2 if p3 is None else p3
        return (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * fx ** 3 + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * fx ** 2 + (-0.5 * p0 + 0.5 * p2) * fx + p1

    a = [rng.uniform(-1.2, 1.2, 1234), rng.uniform(-1, 1, 1234)]  # values above 1: copied unclamped at integer positions (Q16)
    for rate, new in ((44100, 48000), (8000, 48000), (48000, 32000)):
        ratio = new / rate
        newlen = len(a[0]) * ratio
        ref = oracle.resample(oracle.Audio(a, rate), new, oracle.CUBIC)
        for y in range(2):
            line = []
            i = 1
            while i <= newlen:
                xx = (i - 1) / ratio + 1
                line.append(a[y][int(xx) - 1] if xx % 1 == 0 else clamp(cubic(a[y], xx)))
                i += 1
            assert len(line) == len(ref.data[y])
            assert np.max(np.abs(np.array(line) - ref.data[y])) <= 1e-15, (rate, new, y)  # pow(fx, 3) is libm's in Python
    # mono
    s3 = [rng.uniform(-1, 1, 500) for _ in range(3)]
    mono = []
    for i in range(500):
        s = 0
        for c in range(3):
            s = s + s3[c][i]
        mono.append(s / 3)
    assert np.array_equal(oracle.mono(oracle.Audio(s3, 8000)).data[0], np.array(mono))
    # mix: three audios, different lengths and channel counts
    auds = [[rng.uniform(-1, 1, 300)], [rng.uniform(-1, 1, 450), rng.uniform(-1, 1, 450)], [rng.uniform(-1, 1, 20) for _ in range(3)]]
    amp = 0.9
    ln, cn = max(len(x[0]) for x in auds), max(len(x) for x in auds)
    ref = oracle.mix([oracle.Audio(x, 8000) for x in auds], amp)
    for c in range(cn):
        ch = []
        for i in range(ln):
            s = 0
            for x in auds:
                if c < len(x):
                    s = s + (x[c][i] if i < len(x[c]) else 0)
            ch.append(clamp(s * amp))
        assert np.array_equal(ref.data[c], np.array(ch)), c
    # Audio:pcm
    st = [rng.uniform(-1, 1, 77), rng.uniform(-1, 1, 77)]
    for bits, dt, odt in ((8, "unsigned", oracle.UNSIGNED), (16, "signed", oracle.SIGNED), (24, "signed", oracle.SIGNED), (32, "unsigned", oracle.UNSIGNED)):
        mv = 2 ** (bits - 1)
        add = mv if dt == "unsigned" else 0
        enc = lambda d: d * (mv if d < 0 else mv - 1) + add
        inter = [enc(st[c][n]) for n in range(77) for c in range(2)]
        planar = [enc(st[c][n]) for c in range(2) for n in range(77)]
        assert np.array_equal(oracle.encode_pcm(oracle.Audio(st, 8000), bits, odt, True), np.array(inter))
        assert np.array_equal(oracle.encode_pcm(oracle.Audio(st, 8000), bits, odt, False), np.array(planar))


@pytest.mark.parametrize("ch", [1, 2])
def test_q8_wav_ima_loader_transliterated(oracle, ch):
    """aukit.wav's IMA path (aukit.lua:1509-1548 → aukit.adpcm :1218-1274) transliterated: per block, mono masks the header step
    index with 0x0F (Q8) and decodes the bytes low nibble first; stereo re-orders the 4-byte words of both channels into an
    interleaved nibble table; the header predictor is not emitted; blocks are concatenated"""
    rng = np.random.Generator(np.random.PCG64(70 + ch))
    ba = 36 * ch
    spb = (ba - 4 * ch) * 2 // ch
    data = oracle.gen_ima(rng.integers(-9000, 9000, spb * 5 * ch).astype(np.int16), ch, ba, 88)
    ref = oracle.wav_adpcm(data, ba, ch, 22050)

    def clamp(v, lo, hi):
        return lo if v < lo else (hi if v > hi else v)

    def adpcm_nibbles(nibbles, pred, idx, channels):  # aukit.adpcm on a table, interleaved = true
        out = [[] for _ in range(channels)]
        pred, idx = list(pred), list(idx)
        for i in range(len(nibbles) // channels):
            for j in range(channels):
                nib = nibbles[i * channels + j]
                step = _IMA_STEP[idx[j]]
                idx[j] = clamp(idx[j] + _IMA_INDEX[nib], 0, 88)
                diff = (((nib % 8) * step) >> 2) + (step >> 3)
                pred[j] = clamp(pred[j] - diff, -32768, 32767) if nib >= 8 else clamp(pred[j] + diff, -32768, 32767)
                out[j].append(pred[j] / (32768 if pred[j] < 0 else 32767))
        return out

    import struct as st
    chans = [[] for _ in range(ch)]
    for n in range(0, len(data), ba):
        if ch == 2:
            pL, iL, pR, iR = st.unpack_from("<hBxhB", data, n)
            nib = {}
            for i in range(8, ba, 8):
                for k in range(4):  # left word → odd slots, right word → even slots (1-based)
                    b = data[n + i + k]
                    nib[(i - 7 + 2 * k) * 2 - 1] = b & 15
                    nib[(i - 6 + 2 * k) * 2 - 1] = b >> 4
                    b = data[n + i + 4 + k]
                    nib[(i - 7 + 2 * k) * 2] = b & 15
                    nib[(i - 6 + 2 * k) * 2] = b >> 4
            flat = [nib[k] for k in range(1, len(nib) + 1)]
            blk = adpcm_nibbles(flat, [pL, pR], [iL, iR], 2)
        else:
            p, ix = st.unpack_from("<hB", data, n)
            ix &= 0x0F
            body = data[n + 4: n + ba]
            flat = [v for b in body for v in (b & 15, b >> 4)]  # topFirst = false: low nibble first
            blk = adpcm_nibbles(flat, [p], [ix], 1)
        for c in range(ch):
            chans[c] += blk[c]
    for c in range(ch):
        assert np.array_equal(ref.data[c], np.array(chans[c])), c


def test_stream_pcm_float_string_end_of_data_by_hand(oracle):
    """Worked by hand from aukit.lua:2291-2311 / :2367-2371 / :228-266, 3 frames at 12 kHz (ratio 4), linear:
    float read() returns nil past the end instead of raising.  Per channel: outputs 10-12 (x = 3.25 .. 3.75) take
    `data[4] or data[3]` and equal d[3]; output 13 is `d[y][4]` = nil and `s - ls[y]` raises → 12 samples.  Mono over 2+ channels:
    the lazy __index computes `(rawget(self, i) or 0) + read()` → arithmetic on nil at output 10 → 9 samples, as for integers,
    whose read() raises at the same place on `s < 0`.  One channel forces mono off (:2243)."""
    for ch in (1, 2, 3):
        x = np.linspace(-0.5, 0.75, 3 * ch)
        f = x.astype("<f4").tobytes()
        per_channel = oracle.stream_pcm(f, 32, oracle.FLOAT, ch, 12000, False, False, oracle.LINEAR)
        assert per_channel.nchunks == 1 and list(per_channel.chunk_len[:, 0]) == [12] and per_channel.final_status == 0
        xf = x.astype("<f4").astype(np.float64).reshape(3, ch)
        alpha = 1 - np.exp(-(12000 / 96000) * 2 * np.pi)
        for c in range(ch):
            s = [xf[0, c] + (xf[1, c] - xf[0, c]) * k / 4 for k in range(4)] + [xf[1, c] + (xf[2, c] - xf[1, c]) * k / 4 for k in range(4)] + [xf[2, c]] * 4
            ls, want = 0.0, []
            for v in s:
                ns = ls + alpha * (v - ls)
                want.append(min(max(ns * (128 if ns < 0 else 127), -128), 127))
                ls = v
            assert np.max(np.abs(per_channel.data[c] - np.array(want))) <= 1e-12
        mixed = oracle.stream_pcm(f, 32, oracle.FLOAT, ch, 12000, False, True, oracle.LINEAR)
        assert list(mixed.chunk_len[:, 0]) == ([12] if ch == 1 else [9]) and mixed.final_status == 0
        i16 = (x * 30000).astype("<i2").tobytes()
        assert list(oracle.stream_pcm(i16, 16, oracle.SIGNED, ch, 12000, False, False, oracle.LINEAR).chunk_len[:, 0]) == [9]
