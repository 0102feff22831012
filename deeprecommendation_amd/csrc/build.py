"""Builds libncf_hip.so (gfx950 device code + C ABI) in-tree with hipcc.  No torch, no cmake.

    python -m deeprecommendation_amd.csrc.build            # build if sources are newer than the .so
    python -m deeprecommendation_amd.csrc.build --force
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["abi.hip", "gather.hip", "mlp_fused.hip", "linear.hip", "spmm.hip", "attn.hip", "attn_split.hip", "attn_cand.hip", "mlp_bf16.hip", "mlp_bf16_ws8.hip", "backward.hip", "exchange.hip", "dense_csr.hip"]
LIB = os.path.join(PKG, "libncf_hip.so")
ARCH = "gfx950"
# per-file flags.  mlp_bf16.hip: MFMA accumulators in VGPRs instead of AGPRs (the ReLU / bf16 conversion of the hidden
# layers reads them with VALU instructions; in AGPR form hipcc copies whole 16-register tuples first) — measured with
# tools/ab_bf16.py, bit-identical results: weight-stationary kernel 222 -> 216 us (1 M pairs), streaming kernel 20.4 -> 19.9 us
EXTRA_FLAGS = {"mlp_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "mlp_bf16_ws8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(HERE, "ncf_common.h"), os.path.join(HERE, "mlp_bf16.h"), os.path.join(HERE, "attn_util.h"), os.path.join(HERE, "group_pairs.h"), os.path.join(PKG, "..", "include", "ncf_abi.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
