"""Host-side checks that need no GPU: the C-ABI library loads, exports every symbol include/mnav.h
declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from mesh_navigation_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    lib = build.build_lib()
    assert os.path.exists(lib)
    L = ctypes.CDLL(lib)
    header = open(os.path.join(ROOT, "include", "mnav.h")).read()
    declared = set(re.findall(r"\b(mnav_[a-z_]+)\s*\(", header))
    declared -= {"mnav_ctx", "mnav_stats"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.MnavContext(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mesh_navigation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "from oracle" not in txt and "import oracle" not in txt and "mnav_oracle" not in txt, f
