// dfpwm_dev.h — DFPWM1a predictor / decoder / encoder steps (device).
//
// The arithmetic is NOT in the reference tree: aukit.lua:85 `require "cc.audio.dfpwm"` pulls a CC: Tweaked ROM module
// with no pinned version.  Restated from the published DFPWM1a algorithm (PREC = 10, strength floor 8, anti-jerk,
// low-pass 140/256) — **parity unpinned** (see DESIGN.md); the CPU oracle restates the same text independently in C.
#pragma once
#include "common.h"

namespace aukit {

struct DfPred { int charge, strength, prev; };
// math.floor(a / 2^sh) on two's-complement ints is an arithmetic shift
AUKIT_DEV int df_predict(DfPred &p, int bit) {
    // branch-free form of:  target = bit ? 127 : -128;  next = charge + floor((strength * (target - charge) + 512) / 1024);
    //   if next == charge and next ~= target then next = next + (bit ? 1 : -1);
    //   z = (bit == prev) ? 1023 : 0;  if strength ~= z then strength += (bit == prev) ? 1 : -1;  strength = max(strength, 8)
    // (for strength >= 8, and for the reset value 0, the last line is max(min(strength ± 1, 1023), 8))
    const int diff = bit * 255 - 128 - p.charge;
    const int step = (__mul24(p.strength, diff) + 512) >> 10;  // |strength * diff| < 2^18: the 24-bit multiply is exact and full-rate
    const int nudge = (step == 0 && diff != 0) ? 2 * bit - 1 : 0;
    const int next = p.charge + step + nudge;
    const int ns = p.strength + (bit == p.prev ? 1 : -1);
    p.strength = max(min(ns, 1023), 8);
    p.charge = next;
    p.prev = bit;
    return next;
}
struct DfDec { DfPred p; int lpf, pcharge, pbit; };
AUKIT_DEV int df_decode_bit(DfDec &d, int bit) {
    const int charge = df_predict(d.p, bit);
    int aj = charge;
    if (bit != d.pbit) aj = (charge + d.pcharge + 1) >> 1;
    d.pcharge = charge; d.pbit = bit;
    d.lpf += ((aj - d.lpf) * 140 + 0x80) >> 8;
    return d.lpf;
}
struct DfEnc { DfPred p; int pcharge; };
AUKIT_DEV int df_encode_sample(DfEnc &e, int v) {
    const int bit = (v > e.pcharge) || (v == e.pcharge && v == 127);
    e.pcharge = df_predict(e.p, bit);
    return bit;
}

}  // namespace aukit
