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


def test_option_table_matches_the_binding():
    """ncf_set_option / ncf_get_option are host-only: every symbolic value the Python binding knows is accepted by the library's option
    table and reads back; a value outside an option's range and an unknown option are refused with the error string set."""
    from deeprecommendation_amd import native
    lib = native.load_library()
    try:
        for name, values in native.OPTION_VALUES.items():
            for sym, val in values.items():
                native.set_option(name, sym)
                assert native.get_option(name) == val, (name, sym)
            top = max(values.values())
            assert lib.ncf_set_option(name.encode(), top + 50) != native.NCF_OK
            assert name.encode() in lib.ncf_last_error()
        assert lib.ncf_set_option(b"no_such_option", 1) != native.NCF_OK
        with pytest.raises(ValueError):
            native.set_option("bf16_kernel", "no_such_kernel")
    finally:
        for name in native.OPTION_VALUES:
            native.set_option(name, "auto")


def test_library_carries_the_id_of_its_sources_and_a_stale_one_is_refused(tmp_path, monkeypatch):
    """ncf_build_id(): the library embeds a hash of every source / header / flag it was built from; needs_build() compares ids (not
    mtimes), and the binding refuses a library whose id is not that of the sources next to it."""
    from deeprecommendation_amd import native
    from deeprecommendation_amd.csrc import build as b
    lib = native.load_library()
    assert lib.ncf_build_id().decode() == b.source_id() == b.library_id()
    assert not b.needs_build()
    monkeypatch.setattr(b, "source_id", lambda: "0123456789abcdef")       # as if a source had been edited
    assert b.needs_build()
    with pytest.raises(RuntimeError, match="stale"):
        native._check_build_id(lib, "libncf_hip.so")
