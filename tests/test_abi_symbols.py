"""CPU test: the C-ABI library loads and exports every symbol include/ncf_abi.h declares (no compute calls)."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "ncf_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ncf_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from deeprecommendation_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = native.load_library()
    declared = _declared()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ncf_abi.h but not exported"
        assert name in native.SIGNATURES, f"{name} has no ctypes signature"
    assert set(native.SIGNATURES) == set(declared)
    assert lib.ncf_version() == 1
    assert lib.ncf_build_arch() == b"gfx950"


def test_missing_library_fails_loudly(tmp_path):
    from deeprecommendation_amd import native
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load_library(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected_in_eval():
    import torch
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    m = BasicNCF(item_dim=10, user_dim=12, item_emb=8, user_emb=8, mlp_dense_layers=[16]).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(3, dtype=torch.long), torch.zeros(3, dtype=torch.long))
