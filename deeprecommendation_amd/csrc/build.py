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
SOURCES = ["abi.hip", "gather.hip", "mlp_fused.hip", "linear.hip", "spmm.hip", "attn.hip", "attn_split.hip", "attn_cand.hip", "attn_tail.hip", "mlp_bf16.hip", "mlp_bf16_ws8.hip", "backward.hip", "exchange.hip", "dense_csr.hip", "probe.hip"]
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


BASE_FLAGS = ["-O3", "-fPIC", "-std=c++17"]


def dependencies():
    import glob
    return sources() + sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(PKG, "..", "include", "ncf_abi.h")]


def source_id() -> str:
    """16 hex digits over everything the library is built from: every source and header (content, by name), the compiler flags and
    the target.  Embedded in the library (ncf_build_id()); a library with another id is stale, whatever its mtime says."""
    import hashlib
    h = hashlib.sha256()
    h.update(("|".join([ARCH] + BASE_FLAGS) + "|" + repr(sorted(EXTRA_FLAGS.items()))).encode())
    for f in dependencies():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def library_id(path=LIB):
    """The id embedded in a built library, read from the file (no dlopen); None if absent / unstamped."""
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    k = data.find(b"NCF_BUILD_ID=")
    if k < 0:
        return None
    v = data[k + 13:k + 13 + 16]
    return v.decode("ascii", "replace") if len(v) == 16 and all(c in b"0123456789abcdef" for c in v) else None


def needs_build():
    return library_id() != source_id()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    sid = source_id()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    import hashlib
    headers = [f for f in dependencies() if f.endswith(".h")]
    hdr = hashlib.sha256(b"".join(open(f, "rb").read() for f in headers)).hexdigest()
    stamps = {}
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [_hipcc(), f"--offload-arch={ARCH}"] + BASE_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if os.path.basename(src) == "abi.hip":
            cmd.insert(-4, f'-DNCF_BUILD_ID="{sid}"')
        # an object is reused when its source, the headers and its command line are what they were when it was compiled
        stamp = hashlib.sha256((hdr + " ".join(cmd)).encode() + open(src, "rb").read()).hexdigest()
        stamps[obj] = stamp
        try:
            fresh = not force and os.path.exists(obj) and open(obj + ".id").read() == stamp
        except OSError:
            fresh = False
        if fresh:
            continue
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd)))
    for src, obj, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        open(obj + ".id", "w").write(stamps[obj])
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if library_id() != sid:
        raise RuntimeError(f"built {LIB} does not carry the id of its sources ({library_id()} != {sid})")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
