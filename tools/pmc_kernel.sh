#!/bin/bash
# tools/pmc_kernel.sh PATTERN SCRIPT [args...] — SQ_INSTS_VALU / SALU / LDS / WAVE_CYCLES of the kernels matching PATTERN (last dispatch), GPU box
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
pat="$1"; shift
RAW=/tmp/aukit_pmc_$$; rm -rf $RAW; mkdir -p $RAW
timeout ${KT_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $RAW -o pm -- python3 "$@" > $RAW/log.txt 2>&1 < /dev/null
python3 - "$RAW" "$pat" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        print(f"  {k}  {c:22s} {v[-1]:.4g}  (x{len(v)})")
PY
rm -rf $RAW
