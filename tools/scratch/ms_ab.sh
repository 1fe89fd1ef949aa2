timeout 600 python -m pytest tests -m gpu -x -q -k "msadpcm or golden or fuzz or stream_handle" 2>&1 | tail -2
cat > /tmp/m.py <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from aukit_amd import batch as B, _native as N
rng = np.random.default_rng(3)
ctx = B.Context(0, dtype=N.F64)
for ch in (1, 2):
    ba = 1024
    raw = rng.integers(0, 256, ba * 216, dtype=np.uint8)
    raw = raw.reshape(216, ba)
    raw[:, 0] = 0
    if ch == 2: raw[:, 1] = 1
    data = raw.tobytes()
    bt = B.Batch.upload(ctx, [data] * 2048)
    d = B.make_desc(N.CODEC_MSADPCM, ch, 44100, block_align=ba)
    for it in range(3):
        out = B.decode(ctx, bt, d); ctx.sync()
    del out
PY
for lib in tools/variants/libaukit_oldms.so aukit_amd/libaukit_hip.so; do
rm -rf /tmp/dl; PYTHONPATH=$PWD AUKIT_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dl -o dl -- python3 /tmp/m.py 2>&1 | grep -i "error\|Traceback" | head -3
python3 - $lib <<'PY'
import csv, glob, sys
ms = []
for f in glob.glob("/tmp/dl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_msadpcm" in r["Kernel_Name"]: ms.append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2))
print(sys.argv[1][-14:], ms)
PY
done
