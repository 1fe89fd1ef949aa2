"""CPU: hand-computed micro-vectors that pin oracle/oracle_ops.py (the numpy restatement of the structural Audio methods,
generators and packing, aukit.lua:690-866, :1779-1878)."""
import math

import numpy as np
import pytest

from oracle import oracle_ops as OPS


def A(*chans, rate=4):
    return [np.array(c, dtype=np.float64) for c in chans], rate


def test_concat_pads_missing_channels_with_silence():
    out, rate = OPS.concat([A([1, 2, 3], [4, 5, 6]), A([7, 8])])
    assert rate == 4 and [list(c) for c in out] == [[1, 2, 3, 7, 8], [4, 5, 6, 0, 0]]


def test_sub_seconds_from_both_ends():
    a = A(list(range(1, 13)))  # 12 samples at 4 Hz = 3 s
    assert list(OPS.sub(a, 1, 2)[0][0]) == [5, 6, 7, 8, 9]       # indices 1*4+1 .. 2*4+1 inclusive (:737-738)
    assert list(OPS.sub(a)[0][0]) == list(range(1, 13))            # 0 .. len: index 13 does not exist, the table stops
    assert list(OPS.sub(a, -1)[0][0]) == [9, 10, 11, 12]           # start = 3 - 1 = 2 s
    assert list(OPS.sub(a, 0, -2)[0][0]) == [1, 2, 3, 4, 5]
    with pytest.raises(OPS.LuaError):
        OPS.sub(a, 4)
    b = A(list(range(1, 11)))  # 10 samples = 2.5 s
    assert list(OPS.sub(b, -1)[0][0]) == [7, 8, 9, 10]            # start = 1.5 s → index 7; last = 2.5 s → index 11 → stops after 10
    assert list(OPS.sub(b, 1, 0)[0][0]) == [5, 6, 7, 8, 9, 10]
    c = A(list(range(1, 11)), rate=2.5)  # 4 s at a fractional rate: start = 3 s → index 8.5 → sch[8.5] is nil at once
    assert list(OPS.sub(c, -1)[0][0]) == [] and list(OPS.sub(c, 2)[0][0]) == [6, 7, 8, 9, 10]


def test_combine_split_rep_reverse():
    out, _ = OPS.combine([A([1, 2, 3]), A([4], [5, 6])])
    assert [list(c) for c in out] == [[1, 2, 3], [4, 0, 0], [5, 6, 0]]
    l, r = OPS.split(A([1, 2], [3, 4], [5, 6]), [3], [2, 1])
    assert [list(c) for c in l[0]] == [[5, 6]] and [list(c) for c in r[0]] == [[3, 4], [1, 2]]
    with pytest.raises(OPS.LuaError):
        OPS.split(A([1]), [2])
    assert list(OPS.rep(A([1, 2]), 3)[0][0]) == [1, 2, 1, 2, 1, 2]
    assert list(OPS.rep(A([1, 2]), 2.5)[0][0]) == [1, 2, 1, 2] and list(OPS.rep(A([1, 2]), 0.5)[0][0]) == []
    assert list(OPS.reverse(A([1, 2, 3]))[0][0]) == [3, 2, 1]


def test_generators():
    assert len(OPS.new(0.5, 2, 10)[0]) == 2 and list(OPS.new(0.5, 2, 10)[0][1]) == [0] * 5
    assert len(OPS.new(0.05, 1, 10)[0][0]) == 0
    sq = OPS.tone(1, 1, 0.5, "square", 0.25, 1, 8)[0][0]          # x = i/8, (x*1) % 1 >= 0.25 → -amp
    assert list(sq) == [0.5, -0.5, -0.5, -0.5, -0.5, -0.5, -0.5, 0.5]
    saw = OPS.tone(1, 1, 1, "sawtooth", 0.5, 1, 4)[0][0]          # fmod(2x+1, 2) - 1 at x = .25 .5 .75 1
    assert np.allclose(saw, [0.5, -1.0, -0.5, 0.0])               # 1.5, fmod(2, 2) = 0, 0.5, 1
    tri = OPS.tone(1, 1, 1, "triangle", 0.5, 1, 4)[0][0]          # 2|fmod(2x+1.5,2) - 1| - 1
    assert np.allclose(tri, [1.0, 0.0, -1.0, 0.0])
    s = OPS.tone(2, 1, 0.5, "sine", 0.5, 2, 8)
    assert len(s[0]) == 2 and np.allclose(s[0][0], [0.5 * math.sin(2 * (i / 8) * math.pi * 2) for i in range(1, 9)])


def test_pack_modes_and_layouts():
    v = OPS.encode_pcm(A([1.0, -1.0, 0.5, -0.25]), 16, "signed")
    assert list(v) == [32767.0, -32768.0, 16383.5, -8192.0]
    assert OPS.pack(v, 16, "signed", False, OPS.TRUNC) == bytes([0xFF, 0x7F, 0x00, 0x80, 0xFF, 0x3F, 0x00, 0xE0])
    assert OPS.pack(v, 16, "signed", True, OPS.FLOOR)[4:6] == bytes([0x3F, 0xFF])
    neg = OPS.encode_pcm(A([-0.3]), 8, "signed")                   # -38.4: trunc → -38 (0xDA), floor → -39 (0xD9)
    assert OPS.pack(neg, 8, "signed", False, OPS.TRUNC) == b"\xda" and OPS.pack(neg, 8, "signed", False, OPS.FLOOR) == b"\xd9"
    with pytest.raises(OPS.LuaError):
        OPS.pack(neg, 8, "signed", False, OPS.STRICT)
    u = OPS.encode_pcm(A([0.0, 1.0], [-1.0, 0.5]), 8, "unsigned", True)   # d*(127|128)+128, interleaved l r l r
    assert list(u) == [128.0, 0.0, 255.0, 191.5]
    assert list(OPS.encode_pcm(A([0.0, 1.0], [-1.0, 0.5]), 8, "unsigned", False)) == [128.0, 255.0, 0.0, 191.5]
    f = OPS.pack(OPS.encode_pcm(A([0.5]), 32, "float"), 32, "float", True)
    assert f == bytes([0x3F, 0x00, 0x00, 0x00])
