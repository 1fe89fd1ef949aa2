timeout 900 python -m pytest tests/test_gpu_dfpwm_spec.py tests/test_gpu_codecs.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-250
SWEEP_NOISE=0 python tools/r05_dfx_sweep.py 2048 4096 8192 16384 2>&1 | grep -v '^\[dfpwm' | tee gpurun_out/r05b/sweep10.txt
