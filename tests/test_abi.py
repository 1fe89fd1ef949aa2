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
    assert L.aukit_abi_version() == 2


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
    """ctypes mirror of aukit_codec_desc: field order / sizes as in the header (4+4+8+7*4+4+64+64+256+256 = 688 bytes; ABI 2: the predictor /
    step-index arrays hold AUKIT_MAX_PLANAR_CHANNELS entries)."""
    import ctypes as C
    from aukit_amd import _native as N
    assert C.sizeof(N.CodecDesc) == 688
    assert N.CodecDesc.sample_rate.offset == 8 and N.CodecDesc.coef1.offset == 48 and N.CodecDesc.predictor.offset == 176


def _norm_proto(p):
    p = re.sub(r"\s+", " ", p.strip())
    p = re.sub(r"\b(const )?(\w+( \w+)?) ?(\*+) ?(const )?\*? ?\w*", lambda m: m.group(0), p)
    return p


def _params(proto):
    """parameter TYPES of a C prototype (names dropped, whitespace / `const` placement normalised)"""
    inner = proto[proto.index("(") + 1:proto.rindex(")")].strip()
    if inner in ("void", ""):
        return []
    out = []
    for a in inner.split(","):
        a = re.sub(r"/\*.*?\*/", "", a).strip()
        stars = a.count("*")
        a = a.replace("*", " ")
        words = [w for w in a.split() if w != "const"]
        # the last word is the parameter's name unless the declaration is unnamed (type only)
        base_types = {"int", "double", "uint32_t", "uint64_t", "int32_t", "uint8_t", "void", "float", "char", "aukit_ctx", "aukit_batch", "aukit_audio", "aukit_chunks",
                      "aukit_codec_desc", "aukit_container", "aukit_group", "aukit_group_call"}
        if len(words) > 1 and words[-1] not in base_types:
            words = words[:-1]
        out.append(" ".join(words) + "*" * stars)
    return out


def test_lua_shim_declarations_match_the_header():
    """the LuaJIT shim cannot run here (no Lua in the image): at least its ffi.cdef must declare only functions the library exports, with the
    header's parameter types, and its two structs must list the header's fields in the header's order"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aukit_hip.h")).read(), flags=re.S)
    lua = open(os.path.join(ROOT, "aukit_amd", "lua", "aukit.lua")).read()
    cdef = lua[lua.index("ffi.cdef [["):lua.index("]]", lua.index("ffi.cdef [["))]
    hprotos = {m.group(2): m.group(0) for m in re.finditer(r"(?:int|void|const char \*|void \*|aukit_ctx \*)\s*\*?(aukit_[a-z0-9_]+)\s*\([^;{]*\)\s*;", hdr) for m in [re.match(r"(.*?)(aukit_[a-z0-9_]+)\s*\(.*", m.group(0), re.S)]}
    lprotos = {m.group(2): m.group(0) for m in re.finditer(r"(?:int|void|const char \*|aukit_ctx \*)\s*\*?(aukit_[a-z0-9_]+)\s*\([^;]*\)\s*;", cdef) for m in [re.match(r"(.*?)(aukit_[a-z0-9_]+)\s*\(.*", m.group(0), re.S)]}
    assert len(lprotos) >= 30
    for name, proto in lprotos.items():
        assert name in hprotos, name
        assert _params(proto) == _params(hprotos[name]), (name, _params(proto), _params(hprotos[name]))
    # every C call the shim makes is declared in its cdef
    used = set(re.findall(r"\bC\.(aukit_[a-z0-9_]+)", lua))
    assert used <= set(lprotos), used - set(lprotos)

    def fields(text, name):
        end = re.search(r"\}\s*" + name + r"\s*;", text).start()
        body = text[text.rfind("{", 0, end) + 1:end]
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.sub(r"\[.*?\]", "", first.split()[-1]).strip("*"))
            names += [re.sub(r"\[.*?\]", "", r.strip()) for r in rest]
        return names
    for st in ("aukit_codec_desc", "aukit_container", "aukit_group_call"):
        assert fields(cdef, st) == fields(hdr, st), st


def test_container_struct_layout():
    import ctypes as C
    from aukit_amd import _native as N
    assert C.sizeof(N.GroupCall) == 160 and N.GroupCall.new_rate.offset == 64 and N.GroupCall.args.offset == 80   # static_assert in group.hip
    assert C.sizeof(N.Container) == 688 + 8 + 8 + 4 + 4 + 8 and N.Container.payload_off.offset == 688 and N.Container.length_seconds.offset == 712
