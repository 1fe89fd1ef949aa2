#!/bin/bash
# tools/pmc_kernel2.sh PATTERN "COUNTERS..." SCRIPT [args...] — any PMC counter set for the kernels matching PATTERN (last dispatch), GPU box
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
pat="$1"; shift
ctrs="$1"; shift
RAW=/tmp/aukit_pmc_$$; rm -rf $RAW; mkdir -p $RAW
timeout ${KT_TIMEOUT:-300} rocprofv3 --pmc $ctrs --output-format csv -d $RAW -o pm -- python3 "$@" > $RAW/log.txt 2>&1 < /dev/null
python3 - "$RAW" "$pat" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        print(f"  {k}  {c:26s} {v[-1]:.4g}  (x{len(v)})")
PY
tail -3 $RAW/log.txt | cut -c1-300
rm -rf $RAW
