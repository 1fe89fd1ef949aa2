#!/usr/bin/env python3
"""Copy the summaries tools/profile_all.sh left in gpurun_out/ into profiles/ (r1_<tag>_* -> r<NN>_<tag>_*), and rebuild
profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes (first dispatch of the dominant kernel of each tag).

usage: python tools/collect_profiles.py [round=01]
"""
import csv, glob, json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(src, "r1_*.csv")) + glob.glob(os.path.join(src, "r1_*_timeline.txt"))):
    shutil.copy(f, os.path.join(dst, "r%s_%s" % (rnd, os.path.basename(f)[3:])))
    n += 1
shutil.copy(os.path.join(root, "gpurun_out", "bench_lines.jsonl"), os.path.join(dst, "r%s_bench_lines.jsonl" % rnd))

# kernel name as bench.py reports it (ctx.last_kernel) -> (tag, substring of the rocprofv3 kernel name)
DOMINANT = [("k_wave_f64<pcm_s16le_mono,cubic,tile640,phase_regs>", "wavef64", "k_wave_f64", 4096),
            ("k_fast_wave<pcm_s16le_mono,cubic,nv2>", "fastwave", "k_fast_wave<", 4096),
            ("k_wave_f64<pcm_s16le_mono,cubic,tile640,phase_regs,stream_pcm>", "pcmstream", "k_wave_f64", 4096),
            ("k_wave_coef_f64<g711_mono,cubic>", "g711cubic", "k_wave_coef_f64<", 4096),
            ("k_fast_wave_s16x2<cubic,nv4>", "stereo", "k_fast_wave_s16x2<", 2048),
            ("k_fast_wave_stream_s16x2<cubic,nv4,stereo>", "stereostream", "k_fast_wave_stream_s16x2<", 2048),
            ("k_floor_wave_g711<cubic>", "g711stream", "k_floor_wave_g711<", 4096),
            ("k_ima_stream_f32", "ima", "k_ima_stream_f32<", 4096),
            ("k_ms_wave", "msadpcm", "k_ms_wave<", 4096)]


def first(tag, which, pat, col):
    with open(os.path.join(dst, "r%s_%s_pmc_%s.csv" % (rnd, tag, which))) as fh:
        for row in csv.DictReader(fh):
            if pat in row["kernel"] and row.get(col):
                return float(row[col])
    return None


entries = []
for kernel, tag, pat, nstreams in DOMINANT:
    try:
        fk, wk = first(tag, "fetch", pat, "FETCH_SIZE"), first(tag, "write", pat, "WRITE_SIZE")
    except OSError:
        continue
    if fk is None or wk is None:
        continue
    entries.append({"kernel": kernel, "streams": nstreams, "seconds": 10.0, "tag": tag, "fetch_kb": fk, "write_kb": wk,
                    "hbm_bytes_per_launch": int(fk * 1024 * 2 + wk * 1024)})
# multi-launch workloads: HBM bytes of ONE step = every aukit kernel's FETCH_SIZE x 2 + WRITE_SIZE summed over the run, divided by the number of
# steps the run made = the dispatches of a kernel that is launched once per step (the anchor)
STEPS = [("flac_pipeline", "flac", "k_flac_find", 2048), ("flac_pipeline", "flac256", "k_flac_find", 256), ("qoa_stream", "qoa", "k_qoa_wave", 4096),
         ("dfpwm_transcode", "dfpwm", "k_dfx_onset<0>", 16384), ("dfpwm_transcode", "dfpwm2048", "k_dfx_onset<0>", 2048), ("ima_pipeline", "imapipe", "k_ima_lanes", 4096)]


def step_traffic(tag, anchor):
    tot, per, n_anchor = 0.0, {}, 0
    for which, col, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
        with open(os.path.join(dst, "r%s_%s_pmc_%s.csv" % (rnd, tag, which))) as fh:
            na = 0
            for row in csv.DictReader(fh):
                if not row.get(col):
                    continue
                k = row["kernel"].replace("void ", "").replace("aukit::", "").split("(")[0][:60]
                if tag.startswith("dfpwm") and any(x in k for x in ("k_pcm_unpack", "k_dfpwm_quantize", "k_dfpwm_encode_i8", "<1>", "k_dfx_lut")):
                    continue   # the input is MADE by the product's encoder (bench.py DfpwmTranscode.setup): not part of a step
                per[k] = per.get(k, 0.0) + float(row[col]) * mul
                tot += float(row[col]) * mul
                if anchor in row["kernel"]:
                    na += 1
            n_anchor = max(n_anchor, na)
    if not n_anchor:
        return None
    return int(tot / n_anchor), {k: int(v / n_anchor) for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v / n_anchor >= 1e6}, n_anchor


steps = []
for workload, tag, anchor, nstreams in STEPS:
    try:
        got = step_traffic(tag, anchor)
    except OSError:
        continue
    if got:
        steps.append({"workload": workload, "streams": nstreams, "seconds": 10.0, "tag": tag, "anchor": anchor, "steps_in_run": got[2], "hbm_bytes_per_step": got[0], "kernels": got[1]})
note = ("HBM bytes per launch from rocprofv3 PMC passes of `python3 bench.py --steps 5 --warmup 1 --cpu-streams 0 [--workload W]` "
        "(tools/profile_bench.sh, separate passes for FETCH_SIZE and WRITE_SIZE): FETCH_SIZE [KB] x 1024 x 2 (gfx950 reports wide "
        "coalesced reads as 64-B requests: MI355X_MICROARCH.md, HBM/rocprofv3 section) + WRITE_SIZE [KB] x 1024. Source rows: "
        "profiles/r%s_<tag>_pmc_fetch.csv / _pmc_write.csv; rebuilt by tools/collect_profiles.py." % rnd)
with open(os.path.join(dst, "traffic.json"), "w") as fh:
    json.dump({"note": note, "entries": entries, "steps": steps}, fh, indent=1)
print("copied %d csv files, %d traffic entries" % (n, len(entries)))
for e in entries:
    print("  %-60s %.3f GB" % (e["kernel"], e["hbm_bytes_per_launch"] / 1e9))
for e in steps:
    print("  step %-55s %.3f GB over %d steps" % (e["workload"], e["hbm_bytes_per_step"] / 1e9, e["steps_in_run"]))
