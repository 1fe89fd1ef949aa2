#!/bin/bash
# soak of the chunk-speculative DFPWM engine: tests/test_gpu_fuzz.py::test_fuzz_dfpwm_speculation (+ the older DFPWM sweeps) at shifted seed ranges.  usage: r05_spec_soak.sh FIRST LAST [STEP]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
first=${1:-100}; last=${2:-1000}; step=${3:-100}
mkdir -p gpurun_out
: > gpurun_out/spec_soak.log
for off in $(seq $first $step $last); do
  AUKIT_FUZZ_SEED_OFFSET=$off timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "dfpwm" 2>&1 | tail -25 > /tmp/soak_one.txt
  echo "offset $off: $(tail -1 /tmp/soak_one.txt)" | tee -a gpurun_out/spec_soak.log
  if ! tail -1 /tmp/soak_one.txt | grep -q " passed" || tail -1 /tmp/soak_one.txt | grep -q "failed"; then cat /tmp/soak_one.txt >> gpurun_out/spec_soak.log; fi
done
