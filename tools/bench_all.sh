#!/bin/bash
# runs every bench workload once, appends the JSON lines to gpurun_out/bench_lines.jsonl and prints a compact summary (GPU box)
mkdir -p gpurun_out
: > gpurun_out/bench_lines.jsonl
python bench.py --cpu-seconds ${CPU_SECONDS:-4} 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
for w in pcm16_stream pcm16_stereo pcm16_stereo_stream g711_cubic g711_stream ima_stream ima_pipeline msadpcm_stream qoa_stream dfpwm_transcode flac_pipeline; do
  python bench.py --workload $w --steps ${STEPS:-5} --warmup 3 --cpu-seconds ${CPU_SECONDS:-4} 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines.jsonl"):
    try:
        d = json.loads(l); r = d["roofline"]
        c = d.get("cpu_baseline") or {}
        print("%-62s %8.1f Gsamples/s %7.3f ms/step frac %.3f launches %d traffic x%s cpu %s/%s Msamples/s (1/%s cores) last %s" % (
            d["config"]["workload"][:62], d["value"] / 1e3, d["ms_per_step"], r["frac"], r["launches_per_step"],
            ("%.2f" % r["traffic_ratio"]) if r.get("traffic_ratio") else "-", ("%.1f" % c["value"]) if c else "-",
            ("%.1f" % c["all_cores"]["value"]) if c else "-", c.get("all_cores", {}).get("cores", "-"), r["kernel"][:40]))
    except Exception as e:
        print("bad line", e)
PY
