"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/mvtracker_hip.h declares;
the ctypes table of mvtracker_amd.hip covers exactly that set with matching argument counts."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "mvtracker_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|long long|const char\*|void\*)\s+(mvt_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_library_exports_every_declared_symbol():
    from mvtracker_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    decl = declared()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    from mvtracker_amd import hip
    decl = declared()
    assert set(decl) == set(hip.SIGNATURES)
    for name, n in decl.items():
        assert len(hip.SIGNATURES[name]) == n, name


def test_introspection_without_gpu():
    from mvtracker_amd import hip
    assert hip.abi_version() == 6 and hip.build_arch() == "gfx950"
