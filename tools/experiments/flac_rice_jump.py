"""Groundwork for a WAVE-per-frame FLAC decoder (gap 1): where the serial work of a frame sits.

The fused decoder gives a frame to ONE lane, which reads its bits code by code: 3.4 ms for a frame of 2 x 4096 samples however small the batch.
Inside a Rice partition with parameter k the start of the code after the one at bit p is next(p) = p + zeros(p) + 1 + k — a function of the BIT
POSITION alone, so next[] can be filled for every position of the partition with independent instructions (count-leading-zeros on a
sliding window) and only the walk p -> next[p] along the chain of code starts stays serial: one dependent LDS read per sample.  This script parses frames of the bench fixture, checks that the walk over next[] lands on exactly the code starts the
sequential reader finds (remainder bits that look like unary runs and all), and counts what a wave would have to do:

    python tools/experiments/flac_rice_jump.py [frames=12]

Nothing here is built or run by the tests."""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
data = open(os.path.join(ROOT, "bench_data", "flac_44100_stereo_10s_0.bin"), "rb").read()
bits = bin(int.from_bytes(data, "big"))[2:].zfill(8 * len(data))
NFR = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def u(p, n):
    return int(bits[p:p + n], 2) if n else 0


def s(p, n):
    v = u(p, n)
    return v - (1 << n) if n and v >> (n - 1) else v


# metadata
assert data[:4] == b"fLaC"
p = 32
while True:
    last, ln = u(p, 1), u(p + 8, 24)
    if u(p + 1, 7) == 0:
        depth = u(p + 32 + 80 + 20 + 3, 5) + 1
    p += 32 + 8 * ln
    if last:
        break

tot = dict(frames=0, samples=0, rice_samples=0, rice_bits=0, partitions=0, header_bits=0, walk_ok=0, walk_bad=0, max_part_bits=0, max_zeros=0, other_samples=0)
part_bits, part_n = [], []
for _ in range(NFR):
    f0 = p
    assert u(p, 14) == 0x3FFE, (p, hex(u(p, 16)))
    bsc, src, asg, ssz = u(p + 16, 4), u(p + 20, 4), u(p + 24, 4), u(p + 28, 3)
    p += 32
    t = u(p, 8)
    n1 = 0
    while t & (0x80 >> n1):
        n1 += 1
    p += 8 * max(n1, 1)
    if bsc == 6: bs = u(p, 8) + 1; p += 8
    elif bsc == 7: bs = u(p, 16) + 1; p += 16
    elif bsc == 1: bs = 192
    elif 2 <= bsc <= 5: bs = 576 << (bsc - 2)
    else: bs = 256 << (bsc - 8)
    if src == 12: p += 8
    elif src in (13, 14): p += 16
    p += 8   # CRC-8
    nsub = 2 if asg >= 8 else asg + 1
    for ch in range(nsub):
        h0 = p
        typ = u(p + 1, 6)
        wasted = 0
        if u(p + 7, 1):
            q = p + 8
            while bits[q] == "0":
                q += 1
            wasted = q - (p + 8) + 1
            p = q + 1
        else:
            p += 8
        sd = depth - wasted + (1 if (asg == 8 and ch == 1) or (asg == 9 and ch == 0) or (asg == 10 and ch == 1) else 0)
        if typ == 0:
            p += sd; tot["other_samples"] += bs; tot["header_bits"] += p - h0; continue
        if typ == 1:
            p += sd * bs; tot["other_samples"] += bs; tot["header_bits"] += p - h0; continue
        order = typ - 8 if typ <= 12 else typ - 31
        p += sd * order
        if typ >= 32:
            prec = u(p, 4) + 1
            p += 9 + prec * order
        method, porder = u(p, 2), u(p + 2, 4)
        p += 6
        tot["header_bits"] += p - h0
        pb = 4 if method == 0 else 5
        for part in range(1 << porder):
            k = u(p, pb); p += pb
            n = (bs >> porder) - (order if part == 0 else 0)
            tot["header_bits"] += pb
            if k == (1 << pb) - 1:
                raw = u(p, 5); p += 5 + raw * n; tot["other_samples"] += n; continue
            # sequential reader: the code starts
            starts, q = [], p
            for _i in range(n):
                starts.append(q)
                z = bits.index("1", q) - q
                tot["max_zeros"] = max(tot["max_zeros"], z)
                q += z + 1 + k
            end = q
            # next[] for EVERY bit position of the partition (what a wave fills in parallel), then the walk
            nxt = {}
            one = end   # position of the next 1 bit at or after r, from the right
            # (a position whose unary run leaves the partition has no successor inside it)
            ones = [i for i in range(p, end) if bits[i] == "1"]
            import bisect
            for r in range(p, end):
                j = bisect.bisect_left(ones, r)
                nxt[r] = (ones[j] + 1 + k) if j < len(ones) else None
            w, ok = p, True
            for _i in range(n):
                ok = ok and w == starts[_i]
                w = nxt[w] if w in nxt else None
                if w is None and _i + 1 < n:
                    ok = False; break
            ok = ok and (w == end or (w is None and n == 0))
            tot["walk_ok" if ok else "walk_bad"] += 1
            tot["partitions"] += 1; tot["rice_samples"] += n; tot["rice_bits"] += end - p
            tot["max_part_bits"] = max(tot["max_part_bits"], end - p)
            part_bits.append(end - p); part_n.append(n)
            p = end
    p = (p + 7) & ~7
    p += 16
    tot["frames"] += 1; tot["samples"] += bs * nsub

print("frames parsed %d (%d samples), Rice-coded %d samples in %d partitions, %d bits (%.2f bits / sample); constant / verbatim / escaped %d samples; headers %d bits" % (
    tot["frames"], tot["samples"], tot["rice_samples"], tot["partitions"], tot["rice_bits"], tot["rice_bits"] / max(tot["rice_samples"], 1), tot["other_samples"], tot["header_bits"]))
print("walk over next[] == the sequential reader's code starts in %d of %d partitions; longest partition %d bits (%d B of LDS as 16-bit next[]), longest unary run %d" % (
    tot["walk_ok"], tot["walk_ok"] + tot["walk_bad"], tot["max_part_bits"], 2 * tot["max_part_bits"], tot["max_zeros"]))
spf = tot["rice_samples"] / max(tot["frames"], 1)
bpf = tot["rice_bits"] / max(tot["frames"], 1)
print("per frame: %.0f coded samples, %.0f bit positions: filling next[] is ~3 instructions per position = %.0f per coded sample (the sequential reader: ~25, dependent);" % (spf, bpf, 3 * bpf / max(spf, 1)))
print("           the walk is one dependent LDS read per code (~100 cycles) where the reader's own chain (align, count zeros, add) is ~50 - 80: no shorter.")
print("           k_flac_decode spends 88 instructions per sample, most of them NOT on that chain — the idea does not pay as it stands (DESIGN.md, gap 1).")
