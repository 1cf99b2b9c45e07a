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


import subprocess
import sys

import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["load,torch,capi", "capi,torch,capi", "torch,capi"])
def test_one_hip_runtime_whatever_the_import_order(order):
    """torch wheels bundle a HIP runtime of their own; capi.load() must end up on the same one as torch in either import
    order, or the second of the two to touch the GPU finds no device (tools/hip_order_diag.py, fresh process per order)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hip_order_diag.py"), order], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "capi ok" in r.stdout and "torch ok" in r.stdout and "FAILED" not in r.stdout, r.stdout
    last = r.stdout.strip().splitlines()[-1]
    assert last.count("libamdhip64") == 1, last
