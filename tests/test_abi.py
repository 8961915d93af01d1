"""The C-ABI shared library: builds in-tree, loads without a GPU and exports every symbol include/como_hip.h declares."""
import ctypes
import os
import re

from tests.conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "como_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(como_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from como_amd import _lib, build
    build.build()
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/como_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in como_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert L.como_abi_version() == 2
    assert L.como_select_workspace_bytes() == 6 * 2048 * 4
    assert L.como_ba_partials_elems(14, 55, 64) == 14 * 55 * 3936


def test_ba_args_struct_matches_header_field_order():
    from como_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "como_hip.h")).read()
    body = hdr[hdr.index("typedef struct como_ba_args {"):hdr.index("} como_ba_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        for nm in " ".join(names[1:]).split(","):
            nm = nm.strip().split()[-1] if nm.strip() else ""
            if nm and nm not in ("const", "void", "int", "long", "double", "uint8_t"):
                fields.append(nm)
    assert fields == [f[0] for f in _lib.BAArgs._fields_]


def test_win_args_struct_matches_header_field_order():
    from como_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "como_hip.h")).read()
    body = hdr[hdr.index("typedef struct como_win_args {"):hdr.index("} como_win_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).split("{", 1)[1]
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        toks = decl.replace("*", " ").replace(",", " ").split()
        fields += [t for t in toks if t not in ("const", "void", "int", "long", "double", "uint8_t")]
    assert fields == [f[0] for f in _lib.WinArgs._fields_]


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "como_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, f"product code imports the oracle: {bad}"


def test_missing_library_fails_loudly(monkeypatch):
    from como_amd import _lib
    import pytest
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcomo_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
