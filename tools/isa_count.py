#!/usr/bin/env python3
"""Instruction counts bench.py quotes, read from the code the library is built from (the `issue` fraction of dfpwm_transcode hard-coded 731
VALU instructions per source dword, counted by hand from one build): compiles a translation unit to gfx950 assembly with the library's flags and
reports, for a kernel, the longest straight-line run of VALU instructions (no label, no branch in between) — for k_dfx_chunks<0> that is the hot
block: one source dword = 16 mono samples through 2 x 15 decoder steps, the mix look-up and 12 encoder steps.
    python tools/isa_count.py            -> writes aukit_amd/isa_counts.json (called by __graft_entry__.build())"""
import hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aukit_amd", "csrc")
OUT = os.path.join(ROOT, "aukit_amd", "isa_counts.json")
WHAT = [("dfx_chunks0_hot_valu", "dfpwm_spec.hip", "_ZN5aukit12k_dfx_chunksILi0E")]


def src_hash(unit):
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f == unit or f.endswith(".h"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def longest_valu_run(asm, mangled_prefix):
    lines = asm.splitlines()
    start = next((i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and ":" in l), None)
    if start is None:
        return None
    best = cur = 0
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", l) or t.startswith("s_cbranch") or t.startswith("s_branch"):
            best, cur = max(best, cur), 0
        elif t.startswith("v_"):
            cur += 1
    return max(best, cur)


def main():
    have = {}
    if os.path.exists(OUT):
        try:
            have = json.load(open(OUT))
        except ValueError:
            have = {}
    out = {}
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    for key, unit, prefix in WHAT:
        sha = src_hash(unit)
        if have.get(key, {}).get("src_sha16") == sha:
            out[key] = have[key]
            continue
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "k.s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                            "--cuda-device-only", "-o", s, os.path.join(CSRC, unit)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            n = longest_valu_run(open(s).read(), prefix)
        out[key] = {"value": n, "unit": unit, "kernel": prefix, "src_sha16": sha, "definition": "longest straight-line run of VALU instructions in the kernel's gfx950 code"}
    json.dump(out, open(OUT, "w"), indent=1)
    return out


if __name__ == "__main__":
    print(json.dumps(main(), indent=1))
