// dfpwm_dev.h — DFPWM1a predictor / decoder / encoder steps (device).
//
// The arithmetic is NOT in the reference tree: aukit.lua:85 `require "cc.audio.dfpwm"` pulls a CC: Tweaked ROM module
// with no pinned version.  Restated from the published DFPWM1a algorithm (PREC = 10, strength floor 8, anti-jerk,
// low-pass 140/256) — **parity unpinned** (see DESIGN.md); the CPU oracle restates the same text independently in C.
//
// One decoder / encoder is a serial dependency chain and a lone wave issues one instruction at a time, so the step functions
// below are written for instruction count.  The published step is
//     target = bit ? 127 : -128
//     next   = charge + floor((strength * (target - charge) + 512) / 1024)
//     if next == charge and next ~= target then next = next + (bit ? 1 : -1)
//     z = (bit == prev) ? 1023 : 0;  if strength ~= z then strength += (bit == prev) ? 1 : -1;  strength = max(strength, 8)
// and both forms used here are algebraically the same integers:
//   * the nudge: with diff = target - charge, step = floor((strength * diff + 512) / 1024) has the sign of diff (or is 0) and
//     |step| <= |diff| for strength <= 1023, so "step, or ±1 if step is 0 and diff is not" = median(step, ±1, diff) with the
//     ±1 of the bit (diff = 0 gives step = 0 and median(0, ±1, 0) = 0)
//   * the strength: for strength >= 8, and for the reset value 0, the last line is clamp(strength ± 1, 8, 1023), and with bits
//     kept as b = ±1 the ± 1 is b * prev_b
//   * the decoder keeps n = -(2 charge + 1): 2 (target - charge) = 255 b + n, floor((s diff + 512) / 1024) =
//     floor((s (2 diff) + 1024) / 2048), and the anti-jerk mean floor((charge + previous + 1) / 2) = -ceil((n + previous n) / 4)
#pragma once
#include "common.h"

namespace aukit {

AUKIT_DEV int df_med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }  // one v_med3_i32

// ---- decoder side: bits are given
struct DfPred { int n = -1, strength = 0, pb = -1; };  // charge 0, strength 0, previous bit 0
AUKIT_DEV int df_charge(const DfPred &p) { return (~p.n) >> 1; }
AUKIT_DEV void df_set(DfPred &p, int charge, int strength, int prev_bit) { p.n = -(2 * charge + 1); p.strength = strength; p.pb = 2 * prev_bit - 1; }
// b = ±1; returns the new n
AUKIT_DEV int df_predict(DfPred &p, int b) {
    const int diff2 = __mul24(b, 255) + p.n;
    int step = (__mul24(p.strength, diff2) + 1024) >> 11;  // |strength * diff2| < 2^20: the 24-bit multiply is exact and full-rate
    step = df_med3(step, b, diff2);
    asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(p.n) : "v"(step), "v"(p.n));  // n -= 2 step in one instruction (the compiler makes it a shift and a subtract)
    p.strength = df_med3(__mul24(b, p.pb) + p.strength, 8, 1023);
    p.pb = b;
    return p.n;
}
struct DfDec { DfPred p; int lpf = 0, pn = -1; };
// bit k of a byte as ±1, from the complemented byte: one v_bfe_i32 (0 / -1) and an OR
AUKIT_DEV int df_pm1(unsigned not_byte, int k) { return ((int)(not_byte << (31 - k)) >> 31) | 1; }
AUKIT_DEV int df_decode_b(DfDec &d, int b) {  // b = ±1
    const bool same = b == d.p.pb;
    const int pn = d.pn;
    const int n = df_predict(d.p, b);
    const int q = (n + (same ? n : pn) + 3) >> 2;  // = -(anti-jerked charge: the mean with the previous charge when the bit flipped)
    d.pn = n;
    // lpf += floor(((aj - lpf) * 140 + 128) / 256) with the lpf moved inside the floor: two multiply-adds and a shift.  The 128
    // sits in a VGPR: gfx9 VOP3 takes one scalar operand and no literal, and left to itself the compiler emits mul, mul, add3.
    int t, u;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(q), "s"(-140), "v"(128));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(u) : "v"(d.lpf), "s"(116), "v"(t));
    d.lpf = u >> 8;
    return d.lpf;
}
AUKIT_DEV int df_decode_bit(DfDec &d, int bit) { return df_decode_b(d, 2 * bit - 1); }

// ---- encoder side: charge kept biased, cu = charge + 128 (0..255), samples as u = v + 128
struct DfEnc { int cu = 128, strength = 0, pb = -1; };
// returns 255 for a 1 bit, 0 for a 0 bit
AUKIT_DEV unsigned df_encode_u(DfEnc &e, unsigned u) {
    // bit = v > charge or (v == charge and v == 127): a sample of 127 always gives 1, which is u > min(cu, 254)
    const bool bit = u > (unsigned)min(e.cu, 254);
    const int target = bit ? 255 : 0, b = bit ? 1 : -1;
    const int diff = target - e.cu;
    int step = (__mul24(e.strength, diff) + 512) >> 10;
    step = df_med3(step, b, diff);
    e.cu += step;
    e.strength = df_med3(__mul24(b, e.pb) + e.strength, 8, 1023);
    e.pb = b;
    return (unsigned)target;
}
AUKIT_DEV int df_encode_sample(DfEnc &e, int v) { return (int)(df_encode_u(e, (unsigned)(v + 128)) & 1u); }  // v in [-128, 127]

}  // namespace aukit
