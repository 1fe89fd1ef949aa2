#!/bin/bash
# correctness of the FLAC decoders on one box, then the config-5 step with each (tools/kt.sh: kernel averages)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r06
{
timeout 900 python3 -m pytest tests/test_gpu_flac.py tests/test_gpu_flac_edge.py -x -q -m gpu 2>&1 | tail -15
for DEC in stream fused; do
  for S in 2048 256; do
    echo "== decoder $DEC streams $S"
    AUKIT_FLAC_DECODER=$DEC KT_TIMEOUT=200 bash tools/kt.sh bench.py --workload flac_pipeline --streams $S --steps 10 --warmup 2 --extra-windows 0 --cpu-streams 0 2>&1 | head -5
  done
done
} 2>&1 | tee gpurun_out/r06/flac_try.txt
