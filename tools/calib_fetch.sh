#!/bin/bash
# tools/calib_fetch.sh — builds tools/calib_fetch.hip on the GPU box and reads FETCH_SIZE / WRITE_SIZE for its known byte counts (separate PMC passes)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_fetch tools/calib_fetch.hip || exit 1
timeout 120 /tmp/calib_fetch ${CALIB_WORK:-40}
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  RAW=/tmp/aukit_calib_$$; rm -rf $RAW; mkdir -p $RAW
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $RAW -o pm -- /tmp/calib_fetch ${CALIB_WORK:-40} > $RAW/log.txt 2>&1 < /dev/null
  python3 - "$RAW" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(f)):
        d.setdefault((r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        print(f"  {k:40s} {c:24s} {v[-1]:.6g}")
PY
  rm -rf $RAW
done
