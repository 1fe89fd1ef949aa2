cat > /tmp/dl.py <<'PY'
import numpy as np, time, sys
from aukit_amd import batch as B, _native as N
ctx = B.Context(0, dtype=N.F32)
rng = np.random.default_rng(1)
data = [rng.integers(0, 256, 120000, dtype=np.uint8).tobytes()] * 16384
bt = B.Batch.upload(ctx, data)
for ch in (1, 2):
    d = B.make_desc(N.CODEC_DFPWM, ch, 48000)
    for it in range(4):
        out = B.decode(ctx, bt, d); ctx.sync()
    del out
PY
for lib in tools/variants/libaukit_oldrows.so aukit_amd/libaukit_hip.so; do
echo "== $lib"
rm -rf /tmp/dl; PYTHONPATH=$PWD AUKIT_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o dl -- python3 /tmp/dl.py 2>&1 | tail -3
python3 tools/kstats.py /tmp/dl 5
done
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/dl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blockmaps" in r["Kernel_Name"] or "k_df_chunks" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:30], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
PY
