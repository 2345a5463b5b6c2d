"""The C-ABI library loads on a machine without a GPU and exports every symbol include/hsk.h
declares; the compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests import util


def _declared_symbols():
    txt = open(os.path.join(util.ROOT, "include", "hsk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hsk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from hysortk_amd import _lib
    lib = _lib.load()
    decl = _declared_symbols()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), "libhsk.so does not export " + name
    assert sorted(_lib.SYMBOLS) == decl, "hysortk_amd/_lib.py SYMBOLS out of sync with include/hsk.h"
    assert lib.hsk_abi_version() == 4


def test_struct_sizes_match_header():
    """ctypes mirrors of hsk_config / hsk_result / hsk_stats must have the C layout (checked by compiling
    a tiny C program against the header)."""
    import subprocess
    import tempfile
    from hysortk_amd import _lib
    src = '#include <stdio.h>\n#include "hsk.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(hsk_config), sizeof(hsk_result), sizeof(hsk_stats));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(util.ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert [int(x) for x in out] == [C.sizeof(_lib.Config), C.sizeof(_lib.Result), C.sizeof(_lib.Stats)]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import hysortk_amd as H
    with pytest.raises(H.HskError) as e:
        H.Context()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under hysortk_amd/ or include/ may reference it."""
    bad = []
    for base in ("hysortk_amd", "include", "examples"):
        for dp, _, files in os.walk(os.path.join(util.ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"hsk_oracle|from oracle|import oracle|hsko_|oracle/", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
