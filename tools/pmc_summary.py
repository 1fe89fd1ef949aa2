#!/usr/bin/env python3
"""Print one line per kernel (first dispatch) from the PMC CSVs that tools/profile_bench.sh leaves in gpurun_out/prof/."""
import csv, sys
tag, pat = sys.argv[1], sys.argv[2:]
for f in ("pmc_sq", "pmc_lds"):
    seen = set()
    try:
        rows = list(csv.DictReader(open("gpurun_out/prof/%s_%s.csv" % (tag, f))))
    except OSError:
        continue
    for row in rows:
        k = row["kernel"][:44]
        if any(p in k for p in pat) and k not in seen:
            seen.add(k)
            print(f, k, {a.replace("SQ_", ""): "%.3g" % float(b) for a, b in row.items() if a not in ("kernel", "dispatch") and b})
