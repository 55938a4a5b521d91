"""CPU: the C-ABI library loads, exports every symbol include/te_hip.h declares, and refuses to
run without a GPU (no CPU fallback). No compute entry point is called here."""
import ctypes as C
import os
import re

import pytest

from pressurepoissonsolver_amd import capi
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "te_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(te_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = declared_symbols()
    assert len(decl) >= 50
    assert sorted(capi.SYMBOLS) == decl


def test_every_symbol_exported():
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_error_convention_and_no_cpu_fallback():
    with pytest.raises(capi.TeError) as e:
        capi.Mesh.read("/nonexistent/mesh.bin", 3)
    assert e.value.code == capi.TE_EIO
    with pytest.raises(capi.TeError) as e:
        capi.Hierarchy(util.mesh("2uni.bin"), 7)  # odd n
    assert e.value.code == capi.TE_EINVAL
    import torch
    if not torch.cuda.is_available():
        H = capi.Hierarchy(util.mesh("2uni.bin"), 8)
        with pytest.raises(capi.TeError) as e:
            capi.GMG(H)
        assert e.value.code == capi.TE_EHIP  # fails loudly: no device, no fallback


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pressurepoissonsolver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "te_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
