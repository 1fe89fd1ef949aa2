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
    const int target = bit ? 127 : -128;
    int next = p.charge + ((p.strength * (target - p.charge) + 512) >> 10);
    if (next == p.charge && next != target) next += bit ? 1 : -1;
    const int z = (bit == p.prev) ? 1023 : 0;
    int ns = p.strength;
    if (ns != z) ns += (bit == p.prev) ? 1 : -1;
    if (ns < 8) ns = 8;
    p.charge = next; p.strength = ns; p.prev = bit;
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
