"""CPU: the C-ABI library loads and exports every symbol include/hypdiff.h declares
(no compute call is made here: there is no GPU in the build container)."""
import os
import re

from conftest import ROOT
from hypdiff_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hypdiff.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"HD_DECL2\(\s*(hd_\w+)", text):
        names.update({m.group(1) + "_f32", m.group(1) + "_f64"})
    for m in re.finditer(r"^\s*(?:int|size_t|int64_t)\s+(hd_\w+)\s*\(", text, flags=re.M):
        names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    decl = declared_symbols()
    assert len(decl) > 50
    for n in decl:
        assert hasattr(lib, n), f"{n} declared in include/hypdiff.h but not exported"
    assert set(_lib.exported_names()) == decl, set(_lib.exported_names()) ^ decl
    assert lib.hd_version() >= 100


def test_no_oracle_in_product():
    """The product never imports the oracle (or /root/reference)."""
    for dp, _, fns in os.walk(os.path.join(ROOT, "hypdiff_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cuh", ".cpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src and "/root/reference" not in src, os.path.join(dp, fn)
