#!/bin/bash
# tools/r06_ab.sh "<env assignments>" ... : config 5 at 2048 (and 256) streams under each environment, kernel averages (one box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r06
for E in "$@"; do
  for S in ${AB_STREAMS:-2048 256}; do
    echo "== [$E] streams $S"
    env $E KT_TIMEOUT=200 bash tools/kt.sh bench.py --workload flac_pipeline --streams $S --steps ${AB_STEPS:-6} --warmup 2 --extra-windows 0 --cpu-streams 0 2>&1 | grep -v "^rc=" | head -${AB_LINES:-2}
  done
done 2>&1 | tee -a gpurun_out/r06/ab.txt
