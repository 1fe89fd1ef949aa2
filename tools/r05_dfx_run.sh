mkdir -p gpurun_out/r05b; rm -f gpurun_out/r05b/lines.jsonl
timeout 900 python -m pytest tests/test_gpu_dfpwm_spec.py tests/test_gpu_codecs.py -x -q -m gpu -k "dfpwm or transcode or config4" > gpurun_out/r05b/tests.log 2>&1; tail -15 gpurun_out/r05b/tests.log
for s in ${SIZES:-2048 4096 8192 16384}; do AUKIT_DFPWM_STATS=1 timeout 300 python bench.py --workload dfpwm_transcode --streams $s --extra-windows 1 --cpu-streams 0 2>gpurun_out/r05b/df_$s.err | tail -1 >> gpurun_out/r05b/lines.jsonl; tail -1 gpurun_out/r05b/df_$s.err | cut -c1-500; done
python - <<'PY'
import json
for l in open('gpurun_out/r05b/lines.jsonl'):
    try: j=json.loads(l); print(j['config']['workload'][:12], j['ms_per_step'], j.get('roofline',{}).get('kernel'))
    except Exception as e: print('bad line', l[:200])
PY
