"""CPU: the C-ABI library loads, exports every symbol include/emap_hip.h declares, and fails loudly without a GPU."""
import ctypes as ct
import os
import re

import pytest

from conftest import HAS_GPU, ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "emap_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emap_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from elevation_mapping_cupy_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libemap_hip.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "python binding table and header disagree"
    assert lib.emap_abi_version() == _lib.ABI_VERSION == 3


def test_struct_layouts_match_header():
    from elevation_mapping_cupy_amd import _lib
    from oracle import emap_oracle as eo
    assert ct.sizeof(_lib.EmapParams) == 8 * 4 + 28 * 8 + (36 * 3 + 12) * 4
    assert ct.sizeof(_lib.EmapParams) == ct.sizeof(eo.EoParams)
    assert ct.sizeof(_lib.EmapStrip) == 16
    assert ct.sizeof(_lib.EmapStats) == 40


def test_invalid_arguments_are_rejected_without_touching_a_device():
    from elevation_mapping_cupy_amd import _lib
    lib = _lib.load()
    ctx = ct.c_void_p()
    P = _lib.EmapParams()
    P.cell_n = 4000; P.mode = 0; P.resolution = 0.04           # fp16 index mode cannot address 4000 cells
    assert lib.emap_create(ct.byref(P), None, 0, None, ct.byref(ctx)) == -1 and not ctx.value
    assert lib.emap_create(None, None, 0, None, ct.byref(ctx)) == -1
    assert lib.emap_destroy(None) == 0
    assert lib.emap_update(None, None, None, ct.c_double(0), ct.c_double(0), None) == -1


@pytest.mark.skipif(HAS_GPU, reason="only meaningful on a machine without a HIP device")
def test_no_cpu_fallback_without_gpu():
    from elevation_mapping_cupy_amd import ElevationMap, Parameter
    from elevation_mapping_cupy_amd._lib import EmapError
    p = Parameter(); p.update()
    with pytest.raises(EmapError):
        ElevationMap(p)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "elevation_mapping_cupy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "emap_oracle" not in txt, f


def test_product_imports_neither_torch_nor_test_code():
    """north_star: "no PyTorch, no CuPy, no Triton" -- the package and the compat layer import none of them, and nothing from tests/
    (the torch.distributed communicator of the strip tests lives in tests/_torch_strips.py)"""
    for top in ("elevation_mapping_cupy_amd", "compat"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+(torch|cupy|triton)\b", txt, flags=re.M), os.path.join(dirpath, f)
                    assert not re.search(r"^\s*(from|import)\s+(_fixtures|_util|_torch_strips|conftest)\b", txt, flags=re.M), os.path.join(dirpath, f)
                    assert "\"tests\"" not in txt and "'tests'" not in txt, os.path.join(dirpath, f)


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md section D lists every entry point of the header next to the reference interface it replaces: a symbol added to
    the ABI without its row there fails here"""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _declared() if "`%s`" % n not in txt]
    assert not missing, "INTEGRATION.md does not name: %s" % ", ".join(missing)


def test_product_opens_nothing_under_oracle_or_the_reference_tree():
    """the product may MENTION the oracle in comments; it must not open, execute or load anything under oracle/ or /root/reference:
    no such path may appear in a string literal of the package, the compat layer or the header"""
    lit = re.compile(r"\"([^\"\n]*)\"|'([^'\n]*)'")
    for top in ("elevation_mapping_cupy_amd", "compat", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".hip", ".h", ".cpp")):
                    continue
                for ln in open(os.path.join(dirpath, f)).read().splitlines():
                    code = ln.split("//")[0] if f.endswith((".hip", ".h", ".cpp")) else ln.split("#")[0]
                    for m in lit.finditer(code):
                        s = m.group(1) or m.group(2) or ""
                        assert "/root/reference" not in s and "oracle/" not in s and "oracle." not in s, "%s: %r" % (f, s)


def test_the_ctypes_stub_of_integration_md_matches_the_header_and_the_binding():
    """INTEGRATION.md (B) shows a maintainer the ctypes Structure for emap_params: its field list must be the header's, in order, and the
    one the shipped binding uses"""
    from elevation_mapping_cupy_amd import _lib
    h = open(os.path.join(ROOT, "include", "emap_hip.h")).read()
    body = re.search(r"typedef struct emap_params \{(.*?)\} emap_params;", h, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    header = [re.sub(r"\[.*\]", "", f.strip()) for decl in body.split(";") if decl.strip() for f in decl.strip().split(None, 1)[1].split(",")]
    t = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = re.findall(r'"(\w+)"', t[t.index("class emap_params(ct.Structure)"):t.index("ctx = ct.c_void_p()")])
    assert stub == header
    assert [f[0] for f in _lib.EmapParams._fields_] == header


def test_the_header_is_plain_c():
    """the drop-in boundary is a C ABI: include/emap_hip.h must compile as C99 (and as C++11) on its own, warnings as errors, and a C
    program that only includes it must link against the library"""
    import subprocess
    import tempfile
    hdr = os.path.join(ROOT, "include", "emap_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    from elevation_mapping_cupy_amd import _lib
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write('#include "emap_hip.h"\n#include <stdio.h>\nint main(void) { printf("%d\\n", emap_abi_version()); return emap_destroy(0); }\n')
        exe = os.path.join(td, "t")
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH),
                               "-Wl,-rpath," + libdir])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and out.stdout.strip() == str(_lib.ABI_VERSION), (out.returncode, out.stdout, out.stderr[-500:])
