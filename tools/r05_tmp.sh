timeout 900 python -m pytest tests/test_gpu_dfpwm_spec.py tests/test_gpu_codecs.py tests/test_gpu_fuzz.py -x -q -m gpu -k "dfpwm or transcode or config4 or encoder" 2>&1 | tail -15 | cut -c1-300
