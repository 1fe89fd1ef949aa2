#!/bin/bash
# per-phase ablation of k_flac_decode on one box: AUKIT_FLAC_FUSED_DBG = 0 (all), 1 (no prediction), 2 (no stores), 3 (neither); results are wrong by design for dbg != 0
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r06
for S in 2048 256; do
  for D in 0 1 2 3; do
    echo "== streams $S dbg $D"
    AUKIT_FLAC_FUSED_DBG=$D KT_TIMEOUT=200 bash tools/kt.sh bench.py --workload flac_pipeline --streams $S --steps 10 --warmup 2 --extra-windows 0 --cpu-streams 0 2>&1 | grep -v "^rc=" | head -4
  done
done 2>&1 | tee gpurun_out/r06/flac_phase.txt
python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 | tee gpurun_out/r06/headline0.json
