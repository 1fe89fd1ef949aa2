timeout 600 python -m pytest tests -m gpu -x -q -k "qoa or golden or fuzz" 2>&1 | tail -2
cat > /tmp/q.py <<'PY'
import numpy as np, time, sys
sys.path.insert(0, ".")
from aukit_amd import batch as B, _native as N
from oracle import oracle as O
from tests.util import pcm16
ctx = B.Context(0, dtype=N.F32)
base = [O.gen_qoa(np.stack([pcm16(441000, 44100, 8, 4 * i + c) for c in range(2)], 1).ravel(), 2, 44100) + b"\0" * 8 for i in range(4)]
bt = B.Batch.upload(ctx, [base[i % 4] for i in range(2048)])
d = B.make_desc(N.CODEC_QOA, 2, 44100)
for it in range(4):
    out = B.decode(ctx, bt, d); ctx.sync()
PY
for lib in tools/variants/libaukit_oldqoa.so aukit_amd/libaukit_hip.so; do
rm -rf /tmp/dl; PYTHONPATH=$PWD AUKIT_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o dl -- python3 /tmp/q.py > /dev/null 2>&1
echo "== $lib"; python3 tools/kstats.py /tmp/dl 3
done
