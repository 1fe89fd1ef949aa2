#!/bin/bash
# round / turn census of k_flac_decode (a -DAUKIT_FLAC_STATS build: tools/build_variant.sh stats flac_fused.hip "-DAUKIT_FLAC_STATS")
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r06
for S in 2048 256; do
  echo "== streams $S"
  AUKIT_LIB=tools/ab/libaukit_stats.so AUKIT_FLAC_STATS=1 python3 bench.py --workload flac_pipeline --streams $S --steps 1 --warmup 0 --extra-windows 0 --cpu-streams 0 2>&1 | grep "flac stats" | tail -1
done | tee gpurun_out/r06/flac_stats.txt
