#!/bin/bash
# A/B of k_wave_f64 on the headline workload (same box, interleaved): tile size, grid size, Horner form instead of the phase table
mkdir -p gpurun_out
one() { python bench.py --steps 30 --warmup 5 --cpu-streams 0 --fast-line 0 --extra-windows 2 "${@:2}" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), 'ms', round(d['roofline']['frac'],4), d['windows']['ms_per_step'])"; }
for rep in 1 2; do
  AUKIT_F64_TILE=512 one t512
  AUKIT_F64_TILE=1024 one t1024
  AUKIT_F64_TILE=512 AUKIT_F64_HORNER=1 one h512
  for g in 6 12 24 32 48 64; do AUKIT_F64_TILE=512 AUKIT_FAST_BLOCKS_PER_CU=$g one t512g$g; done
  for g in 6 32; do AUKIT_F64_TILE=1024 AUKIT_FAST_BLOCKS_PER_CU=$g one t1024g$g; done
  one f32 --exact-math 0
done
