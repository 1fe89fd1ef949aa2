cat > /tmp/qs.py <<'PY'
import sys; sys.path.insert(0, ".")
import numpy as np
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16
ctx = B.Context(0, dtype=N.F32)
base = [np.stack([pcm16(441000, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel() for i in range(4)]
qs = [O.gen_qoa(base[i], 2, 44100) + b"\0" * 8 for i in range(4)]
bt = B.Batch.upload(ctx, [qs[i % 4] for i in range(1024)])
out = None
for i in range(3):
    out, ck = B.stream_decode(ctx, bt, B.make_desc(N.CODEC_QOA, 2, 44100), "cubic", mono=False, dtype=N.F32, out=out)
ctx.sync()
PY
rm -rf /tmp/qq; PYTHONPATH=$PWD timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/qq -o qq -- python3 /tmp/qs.py > /dev/null 2>&1
python3 tools/kstats.py /tmp/qq 8
ls /tmp/qq/*/ | head; python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/qq/**/*memory_copy_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): print(r)
PY
