cat > /tmp/rv.py <<'PY'
import sys; sys.path.insert(0, ".")
import numpy as np
from aukit_amd import batch as B, _native as N
ctx = B.Context(0, dtype=N.F32)
rng = np.random.default_rng(3)
pcm = (rng.standard_normal(480000 * 2) * 6000).astype(np.int16).tobytes()
au = B.decode(ctx, B.Batch.upload(ctx, [pcm] * 1024), B.make_desc(N.CODEC_PCM, 2, 48000, 16, "signed"))
for i in range(3): B.effect(ctx, au, "reverb")
B.effect(ctx, au, "delay", 0.05, 0.5); B.effect(ctx, au, "delay", 0.05, 0.5)
ctx.sync()
PY
rm -rf /tmp/rv; PYTHONPATH=$PWD timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rv -o rv -- python3 /tmp/rv.py > /dev/null 2>&1
python3 tools/kstats.py /tmp/rv 10
