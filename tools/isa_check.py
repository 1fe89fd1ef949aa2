#!/usr/bin/env python3
"""Build-time check of the hand-scheduled memory pipelines (wave_f64.hip's Row::issue / wait, wave_coef_f64.hip's CoefRow, dfpwm_par.hip's AUKIT_DFF_ISSUE /
WAIT, flac.hip / fast_wave_dev.h's "landed" statements are NOT covered: those pass through compiler-visible loads).

The kernels issue `ds_read_*` / `global_load_*` in one inline-asm statement and wait for them in a LATER one (`s_waitcnt lgkmcnt(N)` /
`vmcnt(0)`); hipcc believes an asm statement's outputs are valid the moment it ends, so nothing in the source stops it from copying,
spilling or reusing such a register before the data has landed — correctness rests on the register allocation of the day.  This script
disassembles the translation unit and walks every kernel in text order with the hardware's counters:

  * every LDS / scalar-memory instruction enters the lgkm queue, every vector-memory instruction the vm queue; the hand-issued loads
    enter with their destination registers, everything else anonymously;
  * `s_waitcnt lgkmcnt(K)` retires all but the newest K entries of the lgkm queue (LDS returns in order); `vmcnt(0)` empties the vm
    queue (loads and stores return out of order with respect to each other: a partial vmcnt retires nothing here);
  * any instruction that reads or writes a register of an entry still in its queue is a violation.

Straight-line approximation: labels and branches do not reset the queues (a loop's back edge is checked as if it fell through), which
is the conservative direction for the loops in question — their bodies end in a full wait.

usage: python tools/isa_check.py [--cfg] file.hip [more.hip ...]     (exit code 1 on a violation; --cfg: check_vm_cfg, see there)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
VM = re.compile(r"vmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_asm(src):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "aukit_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def check(asm):
    """-> (violations, hand_issued_loads): violations as (kernel, line number, text, registers)."""
    viol, issued = [], 0
    kernel, in_asm = None, False
    lgkm, vm = [], []   # queues of frozenset(dest regs) (empty = anonymous)
    for ln, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw and not raw[0].isspace() and raw.rstrip().endswith(":") is False and ":" in raw and raw.startswith("_Z"):
            kernel, lgkm, vm = raw.split(":")[0], [], []
            continue
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        s = line.strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], parts[1] if len(parts) > 1 else ""
        if op == "s_endpgm":
            lgkm, vm = [], []
            continue
        touched = regs_of(args)
        # 1. does this instruction touch a register that is still in flight?
        busy = set()
        for q in (lgkm, vm):
            for e in q:
                busy |= e
        hit = touched & busy
        is_load = op.startswith(("ds_read", "global_load", "buffer_load", "flat_load", "scratch_load"))
        if hit and not (in_asm and is_load and not (regs_of(args.split(",")[0]) & busy)):
            # (an address register may legitimately be shared; a destination may not)
            viol.append((kernel, ln, s, sorted(hit)))
        # 2. the counters
        if op == "s_waitcnt":
            m = LGKM.search(args)
            if m:
                k = int(m.group(1))
                lgkm = lgkm[len(lgkm) - k:] if k else []
            m = VM.search(args)
            if m and int(m.group(1)) == 0:
                vm = []
            if not LGKM.search(args) and not VM.search(args) and re.fullmatch(r"\s*0x[0-9a-fA-F]+|\d+", args.strip() or "x"):
                lgkm, vm = [], []   # a raw immediate: assume a full wait only for 0
            continue
        dest = frozenset(regs_of(args.split(",")[0])) if (in_asm and is_load and "lds" not in op) else frozenset()
        if in_asm and dest:
            issued += 1
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            lgkm.append(dest)
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vm.append(dest)
            if op.startswith("flat_"):
                lgkm.append(frozenset())
    return viol, issued


def check_vm_cfg(asm):
    """The vector-memory half of check() on the kernel's control-flow graph instead of in text order, for kernels whose hand-issued
    `global_load` sits in one arm of a fork (k_rs_onepole, flac_tail.hip): in text order the OTHER arm's instructions follow the load and
    every register they reuse looks like a violation.  Per basic block: IN = union of the predecessors' OUT (registers of hand-issued
    loads still in flight on SOME path), `s_waitcnt vmcnt(0)` empties the set, a hand-issued load adds its destination; iterated to the
    fixed point, then every instruction that touches a register of its IN-flight set is a violation.  -> (violations, loads)."""
    viol, issued = [], 0
    kernels, cur = [], None
    for ln, raw in enumerate(asm.splitlines(), 1):
        if raw.startswith("_Z") and ":" in raw and not raw[0].isspace():
            cur = (raw.split(":")[0], [])
            kernels.append(cur)
            continue
        if cur is not None:
            cur[1].append((ln, raw))
    for name, lines in kernels:
        # instructions and labels
        ins, labels, in_asm = [], {}, False
        for ln, raw in lines:
            if ";;#ASMSTART" in raw: in_asm = True; continue
            if ";;#ASMEND" in raw: in_asm = False; continue
            line = raw.split(";")[0].rstrip()
            t = line.strip()
            if not t: continue
            if t.endswith(":") and not raw[0].isspace():
                labels[t[:-1]] = len(ins)
                continue
            if t.startswith("."): continue
            parts = t.split(None, 1)
            ins.append((ln, parts[0], parts[1] if len(parts) > 1 else "", in_asm, t))
            if parts[0] == "s_endpgm": break
        if not ins: continue
        if any(i[1].startswith(("s_setpc", "s_swappc", "s_call")) for i in ins) and any(i[3] and i[1].startswith(("global_load", "buffer_load")) for i in ins):
            viol.append((name, ins[0][0], "indirect branch in a kernel with hand-issued loads: the graph is incomplete", []))   # (sound, not clever)
        # basic blocks
        leaders = {0} | set(labels.values())
        for i, (_, op, args, _, _) in enumerate(ins):
            if op.startswith(("s_branch", "s_cbranch", "s_endpgm")) and i + 1 < len(ins): leaders.add(i + 1)
        starts = sorted(x for x in leaders if x < len(ins))
        bidx = {st: k for k, st in enumerate(starts)}
        blocks = [(st, (starts[k + 1] if k + 1 < len(starts) else len(ins))) for k, st in enumerate(starts)]
        succ = []
        for st, en in blocks:
            _, op, args, _, _ = ins[en - 1]
            out = []
            if op.startswith("s_branch") or op.startswith("s_cbranch"):
                tgt = labels.get(args.strip())
                if tgt is not None and tgt in bidx: out.append(bidx[tgt])
            if not op.startswith("s_branch") and op != "s_endpgm" and en < len(ins): out.append(bidx[en])
            succ.append(out)

        def run(k, state, report):
            nonlocal issued
            st, en = blocks[k]
            fl = set(state)
            for i in range(st, en):
                ln, op, args, ia, text = ins[i]
                if op == "s_waitcnt":
                    m = VM.search(args)
                    if m and int(m.group(1)) == 0: fl = set()
                    continue
                touched = regs_of(args)
                hand = ia and op.startswith(("global_load", "buffer_load"))
                dest = regs_of(args.split(",")[0]) if hand else set()
                hit = touched & fl
                if report and hit and not (hand and not (dest & fl)):
                    viol.append((name, ln, text, sorted(hit)))
                if hand:
                    fl |= dest
                    if report: issued += 1
            return fl

        IN = [set() for _ in blocks]
        work = list(range(len(blocks)))
        while work:
            k = work.pop()
            out = run(k, IN[k], False)
            for j in succ[k]:
                if not out <= IN[j]:
                    IN[j] |= out
                    if j not in work: work.append(j)
        for k in range(len(blocks)): run(k, IN[k], True)
    return viol, issued


def main(argv):
    bad = 0
    cfg = "--cfg" in argv
    for src in [a for a in argv if a != "--cfg"]:
        v, n = (check_vm_cfg if cfg else check)(compile_asm(src))
        print(f"{os.path.basename(src)}: {n} hand-issued loads, {len(v)} violations")
        for k, ln, s, regs in v[:20]:
            print(f"  {k[:60]} line {ln}: `{s}` touches in-flight v{regs}")
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
