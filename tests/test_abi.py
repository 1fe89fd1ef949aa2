"""CPU: the C-ABI library loads and exports every symbol include/aukit_hip.h declares; there is no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "aukit_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aukit_[a-z0-9_]+)\s*\(", src)))


def test_header_and_export_list_agree():
    from aukit_amd import _native as N
    assert sorted(N.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol():
    from aukit_amd import _native as N
    N.build()
    L = N.lib()
    missing = [s for s in _declared() if not hasattr(L, s)]
    assert not missing, missing
    assert L.aukit_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(N.AukitError) as e:
        B.Context(0)
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aukit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libaukit_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)


def test_codec_desc_layout_matches_header():
    """ctypes mirror of aukit_codec_desc: field order / sizes as in the header (4+4+8+7*4+4+64+64+32+32 = 240 bytes)."""
    import ctypes as C
    from aukit_amd import _native as N
    assert C.sizeof(N.CodecDesc) == 240
    assert N.CodecDesc.sample_rate.offset == 8 and N.CodecDesc.coef1.offset == 48 and N.CodecDesc.predictor.offset == 176
