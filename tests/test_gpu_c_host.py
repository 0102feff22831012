"""The C ABI is usable without Python or torch: a C++ host program (tests/c_host/abi_host.cpp) that includes only
include/ncf_abi.h and the HIP runtime is compiled with hipcc against the in-tree libncf_hip.so and run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program_links_and_runs(gpu, tmp_path):
    from deeprecommendation_amd import native
    from deeprecommendation_amd.csrc import build as B
    hipcc = B._hipcc()
    lib_dir = os.path.dirname(native.LIB_PATH)
    exe = str(tmp_path / "abi_host")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", os.path.join(ROOT, "tests", "c_host", "abi_host.cpp"), "-o", exe,
                           "-L" + lib_dir, "-l:libncf_hip.so", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ABI HOST OK" in out.stdout
