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


def test_no_getenv_on_launch_paths():
    """the TE_* switches are read once (te_gmg_create -> Cfg::fromEnv; te_hier_build for the partition), never per launch"""
    src = open(os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", "gmg.hip")).read()
    calls = [ln for ln in src.splitlines() if "getenv(" in ln]
    assert len(calls) == 1 and "optName[o]" in calls[0], calls
    for hdr in os.listdir(os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc")):
        if hdr.endswith(".hpp"):
            assert "getenv(" not in open(os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", hdr)).read(), hdr


def test_every_extern_c_entry_has_an_exception_barrier():
    """no C++ exception may unwind into a C caller: every int-returning entry point of gmg.hip runs inside guarded()"""
    src = open(os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", "gmg.hip")).read()
    body = src[src.index('extern "C" {'):]
    entries = re.findall(r"^int\s+(te_\w+)\(", body, flags=re.M)
    assert len(entries) >= 40
    for name in entries:
        at = re.search(r"^int\s+" + name + r"\([^{;]*\)\s*\{(.{0,80})", body, flags=re.M | re.S)
        assert at and "guarded(" in at.group(1), name


@pytest.mark.gpu
def test_set_option_and_release_workspace():
    H = capi.Hierarchy(util.mesh("2uni.bin"), 8)
    g = capi.GMG(H)
    g.set_option("TE_NO_FCORR", "1")
    g.set_option("TE_NO_FCORR", None)
    with pytest.raises(capi.TeError) as e:
        g.set_option("TE_NO_SUCH_SWITCH", "1")
    assert e.value.code == capi.TE_EINVAL
    with pytest.raises(capi.TeError) as e:
        g.set_option("TE_2D_SIMPLE", "1")  # shapes the level tables: fixed at creation
    assert e.value.code == capi.TE_ESTATE
    from pressurepoissonsolver_amd import problems
    f, _ = problems.init_dirichlet(H.tables(0), 8)
    x = g.new_vector(0)
    its, rr = g.bicgstab(x, g.new_vector(0, f), g.default_opts())
    g.release_workspace()
    x2 = g.new_vector(0)
    its2, rr2 = g.bicgstab(x2, g.new_vector(0, f), g.default_opts())  # allocates them again
    assert its2 == its and (x2.download() == x.download()).all()
