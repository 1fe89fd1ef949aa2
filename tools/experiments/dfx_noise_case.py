"""One cell of tools/r06_dfx_grid.py with the engine's own account of it (AUKIT_DFPWM_STATS): python3 tools/experiments/dfx_noise_case.py [streams] [class] [entry]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
spec = importlib.util.spec_from_file_location("grid", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r06_dfx_grid.py"))
src = open(spec.origin).read().split("worst = []")[0]
ns = {"__name__": "grid", "__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
n = int(args[0]) if args else 256
kind = args[1] if len(args) > 1 else "noise"
entry = args[2] if len(args) > 2 else "Audio:dfpwm"
B, N, ctx = ns["B"], ns["N"], ns["ctx"]
x = ns["make"](kind, n)
bt = B.Batch.wrap(ctx, x.data_ptr(), [i * ns["nb"] for i in range(n + 1)], keep=x)
d = B.make_desc(N.CODEC_DFPWM, 2, 48000)
out = B.Batch(ctx, ctypes.c_void_p())
if entry == "transcode":
    f = lambda: B.dfpwm_transcode_mono(ctx, bt, 2, out=out)
else:
    mono = B.mono(ctx, B.decode(ctx, bt, d, dtype=N.F32))
    f = lambda: B.dfpwm_encode(ctx, mono, True, out=out)
f(); ctx.sync()
os.environ["AUKIT_DFPWM_STATS"] = "1"
f(); ctx.sync()
del os.environ["AUKIT_DFPWM_STATS"]
print(f"{n} x {kind} {entry}: default {ns['timed'](f, n):.2f} ms", flush=True)
os.environ["AUKIT_DFPWM_NOSPEC"] = "1"
print(f"{n} x {kind} {entry}: NOSPEC  {ns['timed'](f, n):.2f} ms", flush=True)
