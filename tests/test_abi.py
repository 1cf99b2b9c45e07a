"""CPU: the C-ABI library loads and exports every symbol include/pt_hip.h declares (no compute calls)."""
import os
import re

from conftest import ROOT
from path_tracer_amd import capi


def test_library_exports_every_declared_symbol():
    L = capi.load()
    hdr = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pt_version() >= 1 and L.pt_kernel_name().decode().startswith("pt_trace_items")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "path_tracer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "pt_oracle" not in txt and "libpt_oracle" not in txt, f
