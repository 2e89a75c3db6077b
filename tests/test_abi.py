"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/sac_hip.h declares (no compute calls without a GPU)."""
import os
import re

import sac_amd
from sac_amd.lib import EXPORTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sac_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sac_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = sac_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/sac_hip.h but not exported"
    assert sorted(EXPORTS) == syms


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    try:
        sac_amd.SacHandle(22, 60, 2001, 0, 2000)
    except sac_amd.SacError as e:
        assert "hip" in str(e).lower()
    else:
        raise AssertionError("creating a handle without a GPU must fail")


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sac_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "libsac_oracle" not in src and "ora_" not in src.replace("flora_", "")
                assert "import oracle" not in src and "from oracle" not in src
                assert "sac_oracle.h" not in src
