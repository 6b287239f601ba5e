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
    assert hip.abi_version() == 7 and hip.build_arch() == "gfx950"


def test_ctypes_structs_match_header_layout(tmp_path):
    """Every struct hip.py mirrors with ctypes has the size and the field offsets the C compiler gives the header's struct:
    a field appended to the header and forgotten in the binding (or reordered) shows here, not as a wrong pointer on the GPU.
    gcc compiles a probe that includes include/mvtracker_hip.h as plain C and prints sizeof / offsetof."""
    import shutil
    import subprocess
    import pytest
    from mvtracker_amd import hip
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"mvt_block_next": hip.BlockNext, "mvt_block_attn": hip.BlockAttn, "mvt_knn_level": hip.KnnLevel, "mvt_lin_frag": hip.LinFrag,
             "mvt_lin_rows": hip.LinRows, "mvt_updater_block": hip.UpdaterBlock, "mvt_updater_weights": hip.UpdaterWeights,
             "mvt_token_inputs": hip.TokenInputs, "mvt_conv_weights": hip.ConvWeights, "mvt_encoder_weights": hip.EncoderWeights}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mvtracker_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr  # (also: the header is valid C and its field names are the binding's)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = {tuple(ln.split()[:2]): int(ln.split()[2]) for ln in out if ln.strip()}
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), (cname, got[(cname, "size")], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
