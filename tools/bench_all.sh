#!/bin/bash
# runs every bench workload once, appends the JSON lines to gpurun_out/bench_lines.jsonl and prints a compact summary (GPU box)
mkdir -p gpurun_out
: > gpurun_out/bench_lines.jsonl
python bench.py 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
for w in pcm16_stream pcm16_stereo pcm16_stereo_stream g711_cubic g711_stream ima_stream msadpcm_stream qoa_stream dfpwm_transcode flac_pipeline; do
  python bench.py --workload $w --steps ${STEPS:-5} --warmup 1 --cpu-streams 0 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines.jsonl"):
    try:
        d = json.loads(l); r = d["roofline"]
        print("%-70s %9.1f Gsamples/s  %8.3f ms/step  last kernel %s (%.0f GB/s)" % (d["config"]["workload"][:70], d["value"] / 1e3, d["ms_per_step"], r["kernel"], r["achieved"]))
    except Exception as e:
        print("bad line", e)
PY
