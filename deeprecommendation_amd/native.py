"""ctypes binding of libncf_hip.so (include/ncf_abi.h) + thin torch-tensor marshalling.

PyTorch is used for device memory and streams only: every function here takes CUDA(=HIP) tensors, passes raw
device pointers + sizes + the current HIP stream to the C ABI and returns tensors it allocated for the result.
There is NO fallback: if the shared library is missing or a tensor is not on a GPU, these functions raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NCF_HIP_LIBRARY") or os.path.join(_HERE, "libncf_hip.so")   # the override is for A/B builds (tools/ab_build.sh)

NCF_F32, NCF_BF16 = 0, 1
NCF_OK, NCF_EINVAL, NCF_EUNSUPPORTED, NCF_ELAUNCH, NCF_EWORKSPACE = 0, -1, -2, -3, -4
ATT_MLP, ATT_LINEAR, ATT_COS, ATT_MLP_SCALED = 0, 1, 2, 3
ATT_SCALE_LOG2 = 64   # include/ncf_abi.h NCF_ATT_SCALE_LOG2

_c_i64 = ctypes.c_int64
_c_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_size = ctypes.c_size_t

# name -> (restype, argtypes); the parity tests check that every symbol declared in include/ncf_abi.h is here
# and exported by the library.
SIGNATURES = {
    "ncf_version": (_c_int, []),
    "ncf_last_error": (ctypes.c_char_p, []),
    "ncf_build_arch": (ctypes.c_char_p, []),
    "ncf_build_id": (ctypes.c_char_p, []),
    "ncf_probe_mfma_bf16": (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "ncf_probe_copy": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "ncf_probe_gather_read": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_i64, _c_int, _c_int, _c_p, _c_p]),
    "ncf_set_option": (_c_int, [ctypes.c_char_p, _c_int]),
    "ncf_get_option": (_c_int, [ctypes.c_char_p, _c_p]),
    "ncf_bucket_ids": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "ncf_bucket_dedup_table_slots": (_c_size, [_c_i64]),
    "ncf_bucket_ids_dedup": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "ncf_gather_buckets": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_int, _c_i64, _c_int, _c_p, _c_i64, _c_p, _c_p]),
    "ncf_gather_concat": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_int, _c_int,
                                   _c_p, _c_i64, _c_p, _c_p]),
    "ncf_gather_dot": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p]),
    "ncf_mlp_workspace_bytes": (_c_size, [_c_int, _c_i64, _c_int, _c_p]),
    "ncf_mlp_forward": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p, _c_i64, _c_p]),
    "ncf_score_fused_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_p]),
    "ncf_mlp_packed_bytes": (_c_size, [_c_int, _c_int, _c_p]),
    "ncf_mlp_pack": (_c_int, [_c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p]),
    "ncf_score_fused": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_int, _c_int,
                                 _c_int, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "ncf_spmm_csr": (_c_int, [_c_int, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_i64, _c_p,
                              _c_i64, _c_p, _c_int, _c_p]),
    "ncf_spmm_csr_dropout": (_c_int, [_c_int, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_i64, _c_p,
                                      _c_i64, _c_p, _c_int, _c_p, ctypes.c_uint32, ctypes.c_float, _c_p]),
    "ncf_degree_accumulate": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "ncf_edge_coef": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "ncf_scale_rows": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, ctypes.c_float, _c_p, _c_i64, _c_p]),
    "ncf_attn_forward": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, ctypes.c_float, _c_p, _c_p, _c_p,
                                  _c_i64, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "ncf_attn_forward_dropout": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, ctypes.c_float, _c_p, _c_p, _c_p,
                                          _c_i64, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_i64, _c_p, ctypes.c_uint32, ctypes.c_float, _c_p]),
    "ncf_attn_backward": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_int,
                                   _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, ctypes.c_uint32, ctypes.c_float, _c_p]),
    "ncf_attn_forward_grouped": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, ctypes.c_float, _c_p, _c_p, _c_p,
                                          _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_i64, _c_int, _c_p, _c_p,
                                          _c_i64, _c_p, _c_p, _c_p]),
    "ncf_group_pairs_workspace_bytes": (_c_size, [_c_i64]),
    "ncf_group_pairs": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p, _c_p]),
    "ncf_group_pairs_rows": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p, _c_p]),
    "ncf_dense_csr_workspace_bytes": (_c_size, [_c_i64]),
    "ncf_dense_csr_rows": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_size, _c_p]),
    "ncf_dense_csr_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "ncf_attn_candidates_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "ncf_attn_candidates_workspace_bytes": (_c_size, [_c_i64]),
    "ncf_attn_candidates": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_i64, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_i64, _c_p, _c_i64,
                                     _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p, _c_p]),
    "ncf_attn_candidates_pack_floats": (_c_size, [_c_int, _c_int]),
    "ncf_attn_candidates_packed_workspace_bytes": (_c_size, [_c_i64, _c_int, _c_i64]),
    "ncf_attn_candidates_pack": (_c_int, [_c_p, _c_i64, _c_int, _c_int, _c_p, _c_p]),
    "ncf_attn_candidates_packed": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_i64, _c_p, _c_i64,
                                            _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_size, _c_p, _c_p]),
    "ncf_attn_split_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "ncf_attn_split_workspace_bytes": (_c_size, [_c_i64, _c_int, _c_int]),
    "ncf_attn_forward_split": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, ctypes.c_float, _c_p, _c_p, _c_p,
                                        _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_i64, _c_int, _c_p, _c_p,
                                        _c_i64, _c_int, _c_int, _c_p, _c_size, _c_p]),
    "ncf_attn_tail_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "ncf_attn_tail_pack_weight": (_c_int, [_c_p, _c_int, _c_int, _c_p, _c_p]),
    "ncf_attn_tail": (_c_int, [_c_p, _c_i64, _c_int, _c_p, _c_int, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int,
                               _c_p, ctypes.c_float, _c_int, _c_p, _c_i64, _c_p]),
    "ncf_edge_softmax_csr": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "ncf_edge_softmax_segmented_workspace_bytes": (_c_size, [_c_i64, _c_i64]),
    "ncf_edge_softmax_segmented": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_size, _c_p]),
    "ncf_score_folded_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "ncf_score_folded": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_int, _c_int, _c_p,
                                  _c_p, _c_p, _c_p]),
    "ncf_linear_forward": (_c_int, [_c_int, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_i64, _c_p]),
    "ncf_gemm_tn_workspace_bytes": (_c_size, [_c_i64, _c_int, _c_int]),
    "ncf_gemm_tn": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_int, _c_p, _c_i64, _c_p, _c_size, _c_p]),
    "ncf_colsum_workspace_bytes": (_c_size, [_c_i64, _c_int]),
    "ncf_colsum": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_size, _c_p]),
    "ncf_relu_backward": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_p]),
    "ncf_relu_backward_out": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, ctypes.c_float, _c_p]),
    "ncf_scatter_add_rows": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "ncf_l2_normalize_rows": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_p, _c_i64, _c_p]),
    "ncf_gather_cols": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_int, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "ncf_scatter_add_cols": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "ncf_adam_step": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                               ctypes.c_float, _c_i64, _c_p]),
}

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libncf_hip: {msg} (status {code})")
        self.code = code


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the HIP library and bind every ABI symbol.  Raises if it is absent — there is no CPU path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc). deeprecommendation_amd has no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        if not os.environ.get("NCF_HIP_LIBRARY"):        # an A/B build named by the override is loaded as it is
            _check_build_id(lib, p)
        _lib = lib
        _options_from_environment(lib)
    return lib


def _check_build_id(lib, path):
    """A library built from other sources than the ones next to it is refused (mtime said nothing about a checkout or a copy)."""
    try:
        from .csrc import build as _build
        want = _build.source_id()
    except (ImportError, OSError):        # sources not shipped: nothing to compare with
        return
    have = lib.ncf_build_id().decode()
    if have != want:
        raise RuntimeError(f"{path} is stale: built from sources {have}, the sources here are {want}. "
                           "Rebuild: python -m deeprecommendation_amd.csrc.build")


# ncf_set_option values by name (include/ncf_abi.h); 0 / "auto" = choose by shape
OPTION_VALUES = {
    "bf16_kernel": {"auto": 0, "ws": 1, "stream": 2, "ws8": 3},
    "linear_kernel": {"auto": 0, "rs": 1, "rsp": 2},
    "linear_kslices": {"auto": 0, "4": 4, "8": 8},
    "attn_grouped_kernel": {"auto": 0, "lds": 1, "scalar": 2},
    "gather_kernel": {"auto": 0, "step": 1, "persistent": 2},
}
_ENV_OPTIONS = {"NCF_BF16_KERNEL": "bf16_kernel", "NCF_LINEAR_KERNEL": "linear_kernel", "NCF_LINEAR_KS": "linear_kslices",
                "NCF_ATTN_GROUPED_KERNEL": "attn_grouped_kernel", "NCF_GATHER_KERNEL": "gather_kernel"}


def set_option(name: str, value) -> None:
    """Kernel-selection override (A/B tools, tests that drive every variant).  ``value``: the symbolic name
    ("ws", "rsp", …, "auto") or the integer of include/ncf_abi.h.  Process-wide; "auto" / 0 restores the default."""
    lib = load_library()
    if isinstance(value, str):
        try:
            value = OPTION_VALUES[name][value]
        except KeyError:
            raise ValueError(f"option {name!r} has no value {value!r}") from None
    _check(lib.ncf_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    _check(load_library().ncf_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def _options_from_environment(lib) -> None:
    """The library never reads the environment; the dev tools' NCF_* variables are applied HERE, once, at load."""
    for var, name in _ENV_OPTIONS.items():
        v = os.environ.get(var)
        if v:
            if v not in OPTION_VALUES[name]:
                raise ValueError(f"{var}={v!r}: expected one of {sorted(OPTION_VALUES[name])}")
            rc = lib.ncf_set_option(name.encode(), OPTION_VALUES[name][v])
            if rc != NCF_OK:
                raise NativeError(rc, lib.ncf_last_error().decode())


def _check(rc: int):
    if rc != NCF_OK:
        raise NativeError(rc, load_library().ncf_last_error().decode())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> int:
    """torch's current stream on t's device as a raw hipStream_t (every launch of the library goes there).  The raw accessor costs
    0.3 us where torch.cuda.current_stream() builds a Stream object (2.3 us, five times per AttentionNCF forward)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU (got {t.device}); deeprecommendation_amd has no CPU path")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return NCF_F32
    if t.dtype == torch.bfloat16:
        return NCF_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor, what: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{what} must be 2-D with unit inner stride")
    return t.shape[0], t.shape[1], t.stride(0)


def _idx(t: Optional[torch.Tensor], B: Optional[int] = None):
    if t is None:
        return None
    if t.dtype != torch.int64 or t.dim() != 1 or not t.is_contiguous():
        raise ValueError("index tensors must be contiguous 1-D int64")
    return t


_oob_flags = {}


def _oob_flag(device) -> torch.Tensor:
    f = _oob_flags.get(device)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _oob_flags[device] = f
    return f


def check_oob(device):
    """Synchronising check of the sticky out-of-range flag (torch indexing raises IndexError in the reference)."""
    f = _oob_flag(device)
    if int(f.item()) != 0:
        f.zero_()
        raise IndexError("index out of range in an embedding gather")


# ------------------------------------------------------------------ K1
def gather_concat(tabA: torch.Tensor, idxA: Optional[torch.Tensor], tabB: Optional[torch.Tensor] = None,
                  idxB: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, B: Optional[int] = None):
    lib = load_library()
    _dev(tabA, "tabA")
    rowsA, EA, ldA = _rows2d(tabA, "tabA")
    rowsB = EB = ldB = 0
    if tabB is not None:
        _dev(tabB, "tabB")
        rowsB, EB, ldB = _rows2d(tabB, "tabB")
        if tabB.dtype != tabA.dtype:
            raise TypeError("tables must share a dtype")
    idxA, idxB = _idx(idxA), _idx(idxB)
    if B is None:
        B = idxA.numel() if idxA is not None else (idxB.numel() if idxB is not None else rowsA)
    if out is None:
        out = torch.empty((B, EA + EB), dtype=tabA.dtype, device=tabA.device)
    _, _, ldo = _rows2d(out, "out")
    _check(lib.ncf_gather_concat(_dt(tabA), _ptr(tabA), rowsA, ldA, _ptr(tabB), rowsB, ldB, _ptr(idxA), _ptr(idxB), B, EA, EB,
                                 _ptr(out), ldo, _ptr(_oob_flag(tabA.device)), _stream(tabA)))
    return out


def gather_dot(tabA: torch.Tensor, idxA, tabB: torch.Tensor, idxB, B: Optional[int] = None):
    lib = load_library()
    _dev(tabA, "tabA"), _dev(tabB, "tabB")
    rowsA, EA, ldA = _rows2d(tabA, "tabA")
    rowsB, EB, ldB = _rows2d(tabB, "tabB")
    if EA != EB or tabA.dtype != tabB.dtype:
        raise ValueError("gather_dot needs equal widths and dtypes")
    idxA, idxB = _idx(idxA), _idx(idxB)
    if B is None:
        B = idxA.numel() if idxA is not None else idxB.numel()
    out = torch.empty((B, 1), dtype=torch.float32, device=tabA.device)
    _check(lib.ncf_gather_dot(_dt(tabA), _ptr(tabA), rowsA, ldA, _ptr(tabB), rowsB, ldB, _ptr(idxA), _ptr(idxB), B, EA,
                              _ptr(out), _ptr(_oob_flag(tabA.device)), _stream(tabA)))
    return out


def bucket_ids(idx: torch.Tensor, rows_per_rank: int, total_rows: int, world: int, cap: int, send: torch.Tensor,
               slot: torch.Tensor, counts: torch.Tensor, overflow: torch.Tensor):
    """Owner bucketing of a batch's ids for a row-sharded table (ncf_bucket_ids): fills ``send`` (world*cap,) int64 with
    local row ids (unused slots 0), ``slot`` (B,) int64 with each pair's row in the exchanged buffer (-1 = dropped),
    ``counts`` (world,) int32; sets the sticky out-of-range flag / ``overflow`` (int32 (1,)) on the device.  No host sync."""
    lib = load_library()
    _dev(idx, "idx")
    idx = _idx(idx)
    B = idx.numel()
    if send.dtype != torch.int64 or send.numel() < world * cap or slot.dtype != torch.int64 or slot.numel() < B \
            or counts.dtype != torch.int32 or counts.numel() < world or overflow.dtype != torch.int32:
        raise ValueError("bucket_ids: send / slot must be int64 of world*cap / B elements, counts / overflow int32")
    _check(lib.ncf_bucket_ids(_ptr(idx), B, int(rows_per_rank), int(total_rows), int(world), int(cap), _ptr(send), _ptr(slot),
                              _ptr(counts), _ptr(_oob_flag(idx.device)), _ptr(overflow), _stream(idx)))
    return send, slot, counts


def bucket_dedup_table_slots(B: int) -> int:
    return int(load_library().ncf_bucket_dedup_table_slots(int(B)))


def bucket_ids_dedup(idx: torch.Tensor, rows_per_rank: int, total_rows: int, world: int, cap: int, send: torch.Tensor,
                     slot: torch.Tensor, counts: torch.Tensor, overflow: torch.Tensor, hkeys: torch.Tensor, hvals: torch.Tensor):
    """ncf_bucket_ids_dedup: owner bucketing with every DISTINCT id listed once.  ``send`` (world*(cap+1),) int64 = buckets of
    [count, ids...]; ``slot`` (B,) int64 = each pair's row in the exchanged (world*cap)-row buffer (-1 = dropped); ``counts`` (world,)
    int32 = distinct ids per owner (may exceed cap); ``hkeys`` / ``hvals``: int64 scratch of bucket_dedup_table_slots(B) slots.
    Sticky flags on the device, no host sync."""
    lib = load_library()
    _dev(idx, "idx")
    idx = _idx(idx)
    B = idx.numel()
    H = min(hkeys.numel(), hvals.numel())
    if (send.dtype != torch.int64 or send.numel() < world * (cap + 1) or slot.dtype != torch.int64 or slot.numel() < B
            or counts.dtype != torch.int32 or counts.numel() < world or overflow.dtype != torch.int32
            or hkeys.dtype != torch.int64 or hvals.dtype != torch.int64):
        raise ValueError("bucket_ids_dedup: send / slot / hkeys / hvals must be int64 (world*(cap+1) / B / table slots), counts / overflow int32")
    Hp = 1 << (H.bit_length() - 1)        # the largest power of two the scratch holds
    _check(lib.ncf_bucket_ids_dedup(_ptr(idx), B, int(rows_per_rank), int(total_rows), int(world), int(cap), _ptr(hkeys), _ptr(hvals), Hp,
                                    _ptr(send), _ptr(slot), _ptr(counts), _ptr(_oob_flag(idx.device)), _ptr(overflow), _stream(idx)))
    return send, slot, counts


def gather_buckets(table: torch.Tensor, recv: torch.Tensor, world: int, cap: int, out: torch.Tensor) -> torch.Tensor:
    """ncf_gather_buckets: out[r * cap + k] = table[recv[r][1 + k]] for k < recv[r][0] (buckets of [count, ids...]); padding untouched."""
    lib = load_library()
    _dev(table, "table")
    rows, E, ld = _rows2d(table, "table")
    if recv.dtype != torch.int64 or recv.numel() < world * (cap + 1) or out.dtype != table.dtype or out.shape[0] < world * cap or out.shape[1] != E:
        raise ValueError("gather_buckets: recv must be int64 (world*(cap+1),) and out (world*cap, E) of the table's dtype")
    _check(lib.ncf_gather_buckets(_dt(table), _ptr(table), rows, ld, _ptr(recv), int(world), int(cap), E, _ptr(out), out.stride(0),
                                  _ptr(_oob_flag(table.device)), _stream(table)))
    return out


# ------------------------------------------------------------------ K2 generic
def _dims_array(dims: Sequence[int]):
    return (ctypes.c_int * len(dims))(*[int(d) for d in dims])


def _ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def mlp_forward(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act chain: Linear, (ReLU, Linear)* — no activation after the last layer.  Any dims, fp32."""
    lib = load_library()
    _dev(x, "x")
    if x.dtype != torch.float32:
        raise TypeError(f"mlp_forward computes in fp32; got activations of dtype {x.dtype}")
    B, K0, ldx = _rows2d(x, "x")
    dims = [K0] + [int(w.shape[0]) for w in weights]
    for i, w in enumerate(weights):
        _dev(w, "weight")
        if w.dtype != torch.float32 or not w.is_contiguous() or w.shape[1] != dims[i]:
            raise ValueError(f"weight {i} must be contiguous fp32 [{dims[i + 1]}, {dims[i]}]")
    biases = [None if b is None else b.contiguous() for b in biases]
    n = len(weights)
    d = _dims_array(dims)
    ws_bytes = lib.ncf_mlp_workspace_bytes(NCF_F32, B, n, d)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty((B, dims[-1]), dtype=torch.float32, device=x.device)
    _, _, ldo = _rows2d(out, "out")
    _check(lib.ncf_mlp_forward(NCF_F32, _ptr(x), B, ldx, n, d, _ptr_array(weights), _ptr_array(biases), _ptr(ws), ws_bytes,
                               _ptr(out), ldo, _stream(x)))
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, out=None) -> torch.Tensor:
    return mlp_forward(x, [weight], [bias], out=out)


# ------------------------------------------------------------------ K1+K2 fused
class PackedMLP:
    """Weights of an MLP ending in a 1-wide layer, pre-packed for ncf_score_fused (built once per model)."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], dtype=torch.float32):
        """``dtype`` selects the kernel family: float32 (exact fp32 MFMA chain) or bfloat16 (weights rounded to bf16
        at pack time, bf16 tables, fp32 accumulate).  Sources are always fp32 [out][in] tensors."""
        lib = load_library()
        self.dims = [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]
        self.n_layers = len(weights)
        self.device = weights[0].device
        self.dtype = dtype
        self.dt = NCF_F32 if dtype == torch.float32 else NCF_BF16
        d = _dims_array(self.dims)
        nbytes = lib.ncf_mlp_packed_bytes(self.dt, self.n_layers, d)
        if nbytes == 0:
            raise NativeError(NCF_EUNSUPPORTED, "MLP shape cannot be packed for the fused kernel")
        ws = [w.detach().to(torch.float32).contiguous() for w in weights]
        bs = [None if b is None else b.detach().to(torch.float32).contiguous() for b in biases]
        for w in ws:
            _dev(w, "weight")
        self.blob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _check(lib.ncf_mlp_pack(self.dt, self.n_layers, d, _ptr_array(ws), _ptr_array(bs), _ptr(self.blob), nbytes,
                                _stream(ws[0])))

    def supports(self, EA: int, EB: int) -> bool:
        return bool(load_library().ncf_score_fused_supported(self.dt, EA, EB, self.n_layers, _dims_array(self.dims)))


def fused_supported(EA: int, EB: int, dims: Sequence[int], dtype=torch.float32) -> bool:
    dt = NCF_F32 if dtype == torch.float32 else NCF_BF16
    return bool(load_library().ncf_score_fused_supported(dt, EA, EB, len(dims) - 1, _dims_array(dims)))


def score_fused(tabA: torch.Tensor, idxA, tabB: Optional[torch.Tensor], idxB, packed: PackedMLP,
                out: Optional[torch.Tensor] = None, B: Optional[int] = None) -> torch.Tensor:
    lib = load_library()
    _dev(tabA, "tabA")
    rowsA, EA, ldA = _rows2d(tabA, "tabA")
    rowsB = EB = ldB = 0
    if tabB is not None:
        _dev(tabB, "tabB")
        rowsB, EB, ldB = _rows2d(tabB, "tabB")
    idxA, idxB = _idx(idxA), _idx(idxB)
    if B is None:
        B = idxA.numel() if idxA is not None else (idxB.numel() if idxB is not None else rowsA)
    if out is None:
        out = torch.empty((B, 1), dtype=torch.float32, device=tabA.device)
    if tabA.dtype != packed.dtype:
        raise TypeError(f"tables are {tabA.dtype} but the MLP was packed for {packed.dtype}")
    _check(lib.ncf_score_fused(_dt(tabA), _ptr(tabA), rowsA, ldA, _ptr(tabB), rowsB, ldB, _ptr(idxA), _ptr(idxB), B, EA, EB,
                               packed.n_layers, _dims_array(packed.dims), _ptr(packed.blob), _ptr(out),
                               _ptr(_oob_flag(tabA.device)), _stream(tabA)))
    return out


# ------------------------------------------------------------------ K4 / K5
def spmm_csr(segptr: torch.Tensor, row_of: Optional[torch.Tensor], col: torch.Tensor, coef: Optional[torch.Tensor],
             z: torch.Tensor, N: int, y: Optional[torch.Tensor] = None, acc_sum: Optional[torch.Tensor] = None,
             partial: Optional[torch.Tensor] = None, fixup: bool = True, dropout: Optional[tuple] = None) -> torch.Tensor:
    """``dropout`` = (p, seed, edge_id or None): per-(edge, feature) message dropout regenerated in the kernel (training)."""
    lib = load_library()
    _dev(z, "z")
    Nz, D, ldz = _rows2d(z, "z")
    n_seg = segptr.numel() - 1
    if segptr.dtype != torch.int64 or col.dtype != torch.int32:
        raise TypeError("segptr must be int64 and col int32")
    if row_of is not None and row_of.dtype != torch.int32:
        raise TypeError("row_of must be int32")
    if y is None:
        y = torch.empty((N, D), dtype=torch.float32, device=z.device)
    if row_of is not None and partial is None:
        partial = torch.empty((n_seg, D), dtype=torch.float32, device=z.device)
    if dropout is not None and dropout[0] > 0:
        p, seed, eid = dropout
        if eid is not None and (eid.dtype != torch.int32 or eid.numel() != col.numel()):
            raise TypeError("edge ids must be int32, one per CSR entry")
        _check(lib.ncf_spmm_csr_dropout(NCF_F32, _ptr(segptr), _ptr(row_of), n_seg, _ptr(col), _ptr(coef), _ptr(z), Nz, ldz, D, _ptr(y),
                                        y.stride(0), _ptr(acc_sum), 0 if acc_sum is None else acc_sum.stride(0), _ptr(partial),
                                        1 if fixup else 0, _ptr(eid), int(seed) & 0xFFFFFFFF, float(p), _stream(z)))
        return y
    _check(lib.ncf_spmm_csr(NCF_F32, _ptr(segptr), _ptr(row_of), n_seg, _ptr(col), _ptr(coef), _ptr(z), Nz, ldz, D, _ptr(y),
                            y.stride(0), _ptr(acc_sum), 0 if acc_sum is None else acc_sum.stride(0), _ptr(partial),
                            1 if fixup else 0, _stream(z)))
    return y


class SegmentedCSR:
    """A CSR-by-destination matrix prepared for ncf_spmm_csr: rows longer than ``seg_len`` edges are split into
    segments (load balance), and rows with more than one segment are finished by re-applying the SAME kernel to the
    partial sums with ``fan``-wide segments, level after level, until one segment per row is left — an ordered
    log-depth tree (a hub item with 4 M in-edges: 8 k partials -> 125 -> 2 -> 1) instead of one lane group walking
    8 k partials serially.  Deterministic: every level adds in index order."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, coef: Optional[torch.Tensor], seg_len: int = 512, fan: int = 64):
        self.n_rows = rowptr.numel() - 1
        self.col, self.coef = col, coef
        counts = rowptr[1:] - rowptr[:-1]
        self._build_tree(rowptr, counts, seg_len, fan)

    @staticmethod
    def _split(rowptr, row_ids, counts, seg_len):
        """Segments of at most seg_len edges per row.  Returns (segptr, row_of or None, nseg per row)."""
        dev = rowptr.device
        nseg = torch.clamp((counts + seg_len - 1) // seg_len, min=1)
        n = counts.numel()
        if n == 0 or int(nseg.max()) == 1:
            return rowptr, None, nseg
        local_row = torch.repeat_interleave(torch.arange(n, device=dev), nseg)
        first = torch.cumsum(nseg, 0) - nseg
        local = torch.arange(local_row.numel(), device=dev) - first[local_row]
        segptr = torch.empty(local_row.numel() + 1, dtype=torch.int64, device=dev)
        segptr[:-1] = rowptr[local_row] + local * seg_len
        segptr[-1] = rowptr[-1]
        return segptr, row_ids[local_row].to(torch.int32).contiguous(), nseg

    def _build_tree(self, rowptr, counts, seg_len, fan):
        dev = rowptr.device
        self.levels = []
        row_ids = torch.arange(self.n_rows, device=dev)
        segptr, row_of, nseg = self._split(rowptr, row_ids, counts, seg_len)
        self.levels.append((segptr, row_of, None))
        while row_of is not None:
            # segments of this level that belong to multi-segment rows are the next level's "edges"
            local_row = torch.repeat_interleave(torch.arange(nseg.numel(), device=dev), nseg)
            multi_seg = (nseg > 1)[local_row]
            edge_ids = torch.nonzero(multi_seg).flatten()                 # ids into this level's partial buffer
            keep_rows = nseg > 1
            row_ids = row_ids[keep_rows]
            counts = nseg[keep_rows]
            rp = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
            rp[1:] = torch.cumsum(counts, 0)
            segptr, row_of, nseg = self._split(rp, row_ids, counts, fan)
            if row_of is None:  # every remaining row fits one segment: the kernel still needs the row map
                row_of_last = row_ids.to(torch.int32).contiguous()
                self.levels.append((segptr, row_of_last, edge_ids.to(torch.int32).contiguous()))
                break
            self.levels.append((segptr, row_of, edge_ids.to(torch.int32).contiguous()))
        self._partials = {}

    def spmm(self, z: torch.Tensor, y: Optional[torch.Tensor] = None, acc_sum: Optional[torch.Tensor] = None,
             coef: Optional[torch.Tensor] = None, dropout: Optional[tuple] = None) -> torch.Tensor:
        """``coef`` overrides the per-edge coefficients for this call (LightGAT recomputes them every layer; a training step
        masks target edges).  ``dropout`` = (p, seed, edge_id or None): message dropout at the edge level (see spmm_csr)."""
        D = z.shape[1]
        coef = self.coef if coef is None else coef
        segptr, row_of, _ = self.levels[0]
        if y is None:
            y = torch.empty((self.n_rows, D), dtype=torch.float32, device=z.device)
        if len(self.levels) == 1:
            return spmm_csr(segptr, row_of, self.col, coef, z, self.n_rows, y=y, acc_sum=acc_sum, dropout=dropout)
        bufs = self._partials.get(D)
        if bufs is None:
            bufs = [torch.empty((lv[0].numel() - 1, D), dtype=torch.float32, device=z.device) for lv in self.levels[:-1]]
            self._partials[D] = bufs
        spmm_csr(segptr, row_of, self.col, coef, z, self.n_rows, y=y, acc_sum=acc_sum, partial=bufs[0], fixup=False, dropout=dropout)
        for li in range(1, len(self.levels)):
            segptr, row_of, edge_ids = self.levels[li]
            last = li == len(self.levels) - 1
            spmm_csr(segptr, row_of, edge_ids, None, bufs[li - 1], self.n_rows, y=y, acc_sum=acc_sum,
                     partial=None if last else bufs[li], fixup=False)
        return y


def degree_accumulate(dst: torch.Tensor, N: int, deg: torch.Tensor):
    lib = load_library()
    _dev(dst, "dst")
    _check(lib.ncf_degree_accumulate(_ptr(dst), dst.numel(), N, _ptr(deg), _ptr(_oob_flag(dst.device)), _stream(dst)))
    return deg


def edge_coef(src: torch.Tensor, dst: torch.Tensor, attr: Optional[torch.Tensor], deg: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    _dev(src, "src")
    coef = torch.empty(src.numel(), dtype=torch.float32, device=src.device)
    _check(lib.ncf_edge_coef(_ptr(src), _ptr(dst), _ptr(attr), _ptr(deg), src.numel(), deg.numel(), _ptr(coef), _stream(src)))
    return coef


def scale_rows(x: torch.Tensor, divisor: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load_library()
    _dev(x, "x")
    N, D, ld = _rows2d(x, "x")
    if out is None:
        out = torch.empty((N, D), dtype=torch.float32, device=x.device)
    _check(lib.ncf_scale_rows(_ptr(x), ld, N, D, float(divisor), _ptr(out), out.stride(0), _stream(x)))
    return out


# ------------------------------------------------------------------ K3
def attn_forward(mode: int, pc: torch.Tensor, pr: torch.Tensor, w1: Optional[torch.Tensor], b1: float,
                 rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, feat: torch.Tensor,
                 out_bias: Optional[torch.Tensor] = None, dropout: Optional[tuple] = None):
    """Returns (out_feat (B, Fdim), weights (nnz,)).  ``dropout`` = (p, seed): AttentionNet's hidden dropout (training)."""
    lib = load_library()
    _dev(pc, "pc")
    B, A, ldpc = _rows2d(pc, "pc")
    I, A2, ldpr = _rows2d(pr, "pr")
    I2, Fdim, ldf = _rows2d(feat, "feat")
    if A != A2 or I != I2:
        raise ValueError("attention operand shapes disagree")
    if rowptr.dtype != torch.int64 or col.dtype != torch.int32 or val.dtype != torch.float32:
        raise TypeError("CSR must be (int64 rowptr, int32 col, fp32 val)")
    out = torch.empty((B, Fdim), dtype=torch.float32, device=pc.device)
    wts = torch.empty(max(col.numel(), 1), dtype=torch.float32, device=pc.device)
    if dropout is not None and dropout[0] > 0:
        _check(lib.ncf_attn_forward_dropout(mode, _ptr(pc), ldpc, _ptr(pr), ldpr, A, _ptr(w1), float(b1), _ptr(rowptr), _ptr(col), _ptr(val),
                                            B, I, _ptr(feat), ldf, Fdim, _ptr(out_bias), _ptr(out), out.stride(0), _ptr(wts),
                                            int(dropout[1]) & 0xFFFFFFFF, float(dropout[0]), _stream(pc)))
        return out, wts[:col.numel()]
    _check(lib.ncf_attn_forward(mode, _ptr(pc), ldpc, _ptr(pr), ldpr, A, _ptr(w1), float(b1), _ptr(rowptr), _ptr(col), _ptr(val),
                                B, I, _ptr(feat), ldf, Fdim, _ptr(out_bias), _ptr(out), out.stride(0), _ptr(wts), _stream(pc)))
    return out, wts[:col.numel()]


def attn_backward(mode: int, pc, pr, w1, rowptr, col, val, feat, wts, dout, dropout: Optional[tuple] = None):
    """Gradients of attn_forward: (d_pc (B, A), d_pr (I, A), d_w1 (A,) or None, d_feat (I, Fdim)); d b1 is exactly zero."""
    lib = load_library()
    _dev(pc, "pc")
    B, A, ldpc = _rows2d(pc, "pc")
    I, _, ldpr = _rows2d(pr, "pr")
    _, Fdim, ldf = _rows2d(feat, "feat")
    dout = dout.contiguous()
    d_pc = torch.empty((B, A), dtype=torch.float32, device=pc.device)
    d_pr = torch.zeros((I, A), dtype=torch.float32, device=pc.device)
    d_feat = torch.zeros((I, Fdim), dtype=torch.float32, device=pc.device)
    mlp = mode in (ATT_MLP, ATT_MLP_SCALED)
    part = torch.empty((B, A), dtype=torch.float32, device=pc.device) if mlp else None
    scratch = torch.empty(max(col.numel(), 1), dtype=torch.float32, device=pc.device)
    p, seed = (dropout if dropout is not None else (0.0, 0))
    _check(lib.ncf_attn_backward(mode, _ptr(pc), ldpc, _ptr(pr), ldpr, A, _ptr(w1), _ptr(rowptr), _ptr(col), _ptr(val), B, I, _ptr(feat), ldf, Fdim,
                                 _ptr(wts), _ptr(dout), dout.stride(0), _ptr(d_pc), A, _ptr(d_pr), A, _ptr(part), _ptr(d_feat), Fdim,
                                 _ptr(scratch), int(seed) & 0xFFFFFFFF, float(p), _stream(pc)))
    return d_pc, d_pr, (colsum(part) if mlp and B > 0 else (torch.zeros(A, device=pc.device) if mlp else None)), d_feat


def attn_grouped_supported(mode: int, A: int, Fdim: int) -> bool:
    """Shapes the LDS-tiled grouped kernel takes (mirrors ncf_attn_forward_grouped's NCF_EUNSUPPORTED conditions)."""
    return mode in (ATT_MLP, ATT_COS, ATT_MLP_SCALED) and A % 4 == 0 and A <= 256 and Fdim <= 256


def attn_backward_supported(mode: int, A: int, Fdim: int) -> bool:
    """Shapes the training pair ``ncf_attn_forward`` + ``ncf_attn_backward`` takes: mirrors ncf_attn_backward's own
    NCF_EUNSUPPORTED conditions (csrc/attn.hip: MLP / cosine only; A and Fdim multiples of 4 and <= 256 — the leading
    dimensions are those of contiguous (.., A) / (.., Fdim) tensors, so they are multiples of 4 when A and Fdim are).  The
    plain forward accepts any Fdim through its generic path, so a gate on the forward's conditions would let
    ``loss.backward()`` raise for e.g. user_emb = 50."""
    return mode in (ATT_MLP, ATT_COS, ATT_MLP_SCALED) and A % 4 == 0 and Fdim % 4 == 0 and 0 < A <= 256 and 0 < Fdim <= 256


def default_pairs_per_wg(B: int) -> int:
    """Pairs of one rated set per workgroup of the grouped attention kernel (4 per wave).  32 (512 threads) stages a tile
    for twice as many pairs as 16 (256 threads): half the gathered bytes per pair, and the tile DMAs hold a wave for about a
    third of a tile's time.  Measured (tools/ab_attn_grouped.py, scalar-operand form, 8 / 16 / 32): config 3 (4096 pairs, 64
    users) 59.2 / 37.9 / 34.4 us; 16 384 pairs 137.8 / 94.4 / 97.1; one user x 65 536 candidates 450.6 / 261.8 / 240.9."""
    return 32 if B >= 2048 else (16 if B >= 512 else 8)


def group_pairs(pair_row: torch.Tensor, n_rows: int, pairs_per_wg: int):
    """Pairs listed row by row for ncf_attn_forward_grouped: (grp_ptr (R+1), pair_ids (B), wg_ptr (R+1)), all int64 on
    pair_row's device — a counting sort on the stream (ncf_group_pairs), no host synchronisation; an out-of-range row
    raises at the next check_oob()."""
    lib = load_library()
    _dev(pair_row, "pair_row")
    if pair_row.dtype != torch.int64 or pair_row.dim() != 1 or not pair_row.is_contiguous():
        raise ValueError("pair_row must be contiguous 1-D int64")
    B, R, dev = pair_row.numel(), int(n_rows), pair_row.device
    grp_ptr = torch.empty(R + 1, dtype=torch.int64, device=dev)
    wg_ptr = torch.empty(R + 1, dtype=torch.int64, device=dev)
    pair_ids = torch.empty(max(B, 1), dtype=torch.int64, device=dev)
    # workgroup -> CSR row (the grid bound of the grouped kernels: every non-empty row adds at most one partly filled workgroup)
    wg_row = torch.empty((B + int(pairs_per_wg) - 1) // int(pairs_per_wg) + min(R, B) + 1, dtype=torch.int32, device=dev)
    nbytes = lib.ncf_group_pairs_workspace_bytes(R)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _check(lib.ncf_group_pairs_rows(_ptr(pair_row), B, R, int(pairs_per_wg), _ptr(grp_ptr), _ptr(pair_ids), _ptr(wg_ptr), _ptr(wg_row),
                                    _ptr(ws), nbytes, _ptr(_oob_flag(dev)), _stream(pair_row)))
    return Grouping(grp_ptr, pair_ids[:B], wg_ptr, wg_row)


class Grouping(tuple):
    """(grp_ptr, pair_ids, wg_ptr) of ncf_group_pairs — unpacks as that triple — plus ``wg_row`` (workgroup -> CSR row)."""

    def __new__(cls, grp_ptr, pair_ids, wg_ptr, wg_row=None):
        self = super().__new__(cls, (grp_ptr, pair_ids, wg_ptr))
        self.wg_row = wg_row
        return self


DENSE_CSR_MAX_ENTRIES = 1 << 26     # col / val are sized for the worst case B * I: beyond 64 M entries the host-sized path is used


def dense_to_csr(user_matrix: torch.Tensor, share_rows: bool = True):
    """(rowptr (B+1) int64, col int32, val fp32, pair_row (B) int64) of a dense (B, I) user_matrix, entirely on the stream
    (ncf_dense_csr_rows + cumsum + ncf_dense_csr_fill; no host read).  The CSR has B rows; with ``share_rows`` a row that equals an
    earlier one is empty and pair_row points at the earlier one.  col / val are sized B * I (only rowptr[B] entries are written)."""
    lib = load_library()
    _dev(user_matrix, "user_matrix")
    if user_matrix.dtype != torch.float32:
        raise TypeError("dense_to_csr takes an fp32 matrix")
    B, I, ld = _rows2d(user_matrix, "user_matrix")
    dev = user_matrix.device
    pair_row = torch.empty(B, dtype=torch.int64, device=dev)
    col = torch.empty(max(B * I, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(B * I, 1), dtype=torch.float32, device=dev)
    if B == 0:
        return torch.zeros(1, dtype=torch.int64, device=dev), col, val, pair_row
    rowptr = torch.empty(B + 1, dtype=torch.int64, device=dev)     # [0, counts...] from the kernels; the cumulative sum runs in place
    nbytes = lib.ncf_dense_csr_workspace_bytes(B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _check(lib.ncf_dense_csr_rows(_ptr(user_matrix), ld, B, I, 1 if share_rows else 0, _ptr(pair_row), _ptr(rowptr), _ptr(ws), nbytes, _stream(user_matrix)))
    torch.cumsum(rowptr, 0, out=rowptr)
    _check(lib.ncf_dense_csr_fill(_ptr(user_matrix), ld, B, I, _ptr(rowptr), _ptr(pair_row), _ptr(col), _ptr(val), _stream(user_matrix)))
    return rowptr, col, val, pair_row


def _events_us(fn, reps, settle):
    for _ in range(settle):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def probe_gather_ceilings(table: torch.Tensor, idx: torch.Tensor, copy_bytes: int, reps: int = 100):
    """What the chip delivers, now, for the two patterns the standalone gather is made of (ncf_probe_copy / ncf_probe_gather_read):
    a streaming copy of ``copy_bytes`` (read + write counted) and random whole-row reads of ``table`` rows named by ``idx`` (read
    bytes only) at 1 / 2 / 4 / 8 rows in flight per lane group and two grid sizes.  GB/s; the best of the sweep is the ceiling."""
    lib = load_library()
    _dev(table, "table")
    rows, E, ld = _rows2d(table, "table")
    rb = E * table.element_size()
    dev = table.device
    st = torch.cuda.current_stream(dev).cuda_stream
    src = torch.empty(copy_bytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    us = _events_us(lambda: _check(lib.ncf_probe_copy(_ptr(src), _ptr(dst), copy_bytes, st)), reps, 30)
    out = {"copy": {"bytes_read_plus_written": 2 * copy_bytes, "us": us, "GBps": 2 * copy_bytes / (us * 1e-6) / 1e9}, "random_row_read": {}}
    n = idx.numel()
    best = 0.0
    for blocks in (2048, 8192):
        sink = torch.empty(blocks * 256, dtype=torch.int32, device=dev)
        for u in (1, 2, 4, 8):
            us = _events_us(lambda: _check(lib.ncf_probe_gather_read(_ptr(table), rows, ld * table.element_size(), rb, _ptr(idx), n, u, blocks,
                                                                     _ptr(sink), st)), reps, 20)
            gbs = n * rb / (us * 1e-6) / 1e9
            out["random_row_read"][f"blocks{blocks}_inflight{u}"] = {"us": us, "GBps": gbs}
            best = max(best, gbs)
    out["random_row_read_best_GBps"] = best
    out["row_bytes"] = rb
    return out


def probe_mfma_bf16(device, seconds: float = 2.0, with_lds: bool = False, blocks: int = 256, iters: int = 20000):
    """The bf16 MFMA rate the chip sustains on random operands (ncf_probe_mfma_bf16): back-to-back launches for ``seconds`` (the
    clock needs about two seconds of load to settle), the LAST launches timed with HIP events.  Returns {"TFLOPs", "clock_GHz",
    "us_per_launch", ...}."""
    lib = load_library()
    g = torch.Generator(device=device).manual_seed(1234)
    rnd = (torch.randn(32768, device=device, generator=g) * 0.5).to(torch.bfloat16).contiguous()      # 64 KiB of finite bf16
    sink = torch.empty(blocks * 512, dtype=torch.float32, device=device)
    clk = torch.zeros(blocks * 8 * 2, dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream

    def launch():
        _check(lib.ncf_probe_mfma_bf16(_ptr(rnd), int(iters), int(blocks), 1 if with_lds else 0, _ptr(sink), _ptr(clk), stream))

    flop = blocks * 8 * iters * 32 * 16384
    launch()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record()
    torch.cuda.synchronize(device)
    one = e0.elapsed_time(e1) * 1e-3
    n = max(4, int(seconds / max(one, 1e-6)))
    for _ in range(n - 4):
        launch()
    e0.record()
    for _ in range(4):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    us = e0.elapsed_time(e1) * 1e3 / 4
    c = clk.view(-1, 2).double().cpu()
    ghz = float((c[:, 0] / c[:, 1].clamp_min(1)).median()) * 0.1
    return {"TFLOPs": flop / (us * 1e-6) / 1e12, "clock_GHz": ghz, "us_per_launch": us, "launches": n, "blocks": blocks,
            "iters": iters, "with_lds_reads": bool(with_lds), "mfma": "v_mfma_f32_16x16x32_bf16, 2 waves per SIMD, register operands, random data",
            "cycles_per_mfma_per_simd": float(c[:, 0].median()) / (iters * 32 * 2)}      # two waves share a SIMD's matrix pipe


def attn_candidates_supported(K: int, N1: int, N2: int) -> bool:
    return bool(load_library().ncf_attn_candidates_supported(int(K), int(N1), int(N2)))


class PackedCandidateWeight:
    """ItemEmbeddings' (N1, K) weight in MFMA operand order for attn_candidates (ncf_attn_candidates_pack): built once per weight
    version (the model keeps it beside its other weight-derived caches)."""

    def __init__(self, Wi: torch.Tensor):
        lib = load_library()
        _dev(Wi, "Wi")
        if Wi.dtype != torch.float32:
            raise TypeError("attn_candidates computes in fp32")
        self.N1, self.K, ldw = _rows2d(Wi, "Wi")
        self.src = Wi
        # Both widths take the packed entry point at every batch size: measured (tools/ab_cand.py, us per call incl. the grouping;
        # LDS-staged -> packed): N1 = 64: 4096 rows 25.1 -> 20.5, 16 384 rows 96.7 -> 74.0; N1 = 128: 512 rows 29.3 -> 22.2 (K range
        # over workgroups), 4096 rows 41.3 -> 38.6, 16 384 rows 159.0 -> 141.5.  ``use_packed = False`` selects the LDS-staged kernel.
        self.use_packed = True
        self.data = torch.empty(lib.ncf_attn_candidates_pack_floats(self.K, self.N1), dtype=torch.float32, device=Wi.device)
        _check(lib.ncf_attn_candidates_pack(_ptr(Wi), ldw, self.K, self.N1, _ptr(self.data), _stream(Wi)))


def attn_candidates(x: torch.Tensor, Wi, bi: Optional[torch.Tensor], Wc: torch.Tensor, b0: Optional[torch.Tensor],
                    pair_row: Optional[torch.Tensor] = None, n_rows: int = 0, pairs_per_wg: int = 32):
    """ncf_attn_candidates: (emb (B, N1), pc (B, N2), grouping or None) — the candidates' ItemEmbeddings and their half of
    AttentionNet.0 in one launch; with ``pair_row`` (B,) int64 the same launch also lists the pairs by rated set (the ``Grouping``
    group_pairs() returns).  ``Wi``: the (N1, K) weight, or a PackedCandidateWeight of it (ncf_attn_candidates_packed: the faster
    form, bit-identical results).  Shapes: attn_candidates_supported(); the fused grouping needs B, n_rows <= 32768."""
    lib = load_library()
    _dev(x, "x")
    if isinstance(Wi, PackedCandidateWeight) and not Wi.use_packed:
        Wi = Wi.src
    packed = isinstance(Wi, PackedCandidateWeight)
    if x.dtype != torch.float32 or (not packed and Wi.dtype != torch.float32) or Wc.dtype != torch.float32:
        raise TypeError("attn_candidates computes in fp32")
    B, K, ldx = _rows2d(x, "x")
    if packed:
        N1, K2, ldw = Wi.N1, Wi.K, 0
    else:
        N1, K2, ldw = _rows2d(Wi, "Wi")
    N2 = int(Wc.shape[0])
    if K2 != K or Wc.shape[1] != N1 or not Wc.is_contiguous():
        raise ValueError("attn_candidates: Wi must be (N1, K) and Wc contiguous (N2, N1)")
    dev = x.device
    emb = torch.empty((B, N1), dtype=torch.float32, device=dev)
    pc = torch.empty((B, N2), dtype=torch.float32, device=dev)
    grp = None
    grp_ptr = pair_ids = wg_ptr = wg_row = ws = None
    nbytes = 0
    R = int(n_rows)
    if pair_row is not None:
        if pair_row.dtype != torch.int64 or pair_row.dim() != 1 or not pair_row.is_contiguous() or pair_row.numel() != B:
            raise ValueError("pair_row must be contiguous 1-D int64 with one entry per row of x")
        grp_ptr = torch.empty(R + 1, dtype=torch.int64, device=dev)
        wg_ptr = torch.empty(R + 1, dtype=torch.int64, device=dev)
        pair_ids = torch.empty(max(B, 1), dtype=torch.int64, device=dev)
        wg_row = torch.empty((B + int(pairs_per_wg) - 1) // int(pairs_per_wg) + min(R, B) + 1, dtype=torch.int32, device=dev)
        nbytes = lib.ncf_attn_candidates_workspace_bytes(R)
        grp = Grouping(grp_ptr, pair_ids[:B], wg_ptr, wg_row)
    if packed:
        nbytes = lib.ncf_attn_candidates_packed_workspace_bytes(B, N1, R if pair_row is not None else -1)
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tail = (_ptr(bi), N1, _ptr(Wc), _ptr(b0), N2, _ptr(emb), emb.stride(0), _ptr(pc), pc.stride(0), _ptr(pair_row), R, int(pairs_per_wg),
            _ptr(grp_ptr), _ptr(pair_ids), _ptr(wg_ptr), _ptr(wg_row), _ptr(ws), nbytes,
            _ptr(_oob_flag(dev)) if pair_row is not None else None, _stream(x))
    if packed:
        _check(lib.ncf_attn_candidates_packed(_ptr(x), B, ldx, K, _ptr(Wi.data), *tail))
    else:
        _check(lib.ncf_attn_candidates(_ptr(x), B, ldx, K, _ptr(Wi), ldw, *tail))
    return emb, pc, grp


class AttnPartials:
    """The entry-split attention's un-merged softmax partials: (B, nsplit, Fdim + 4) floats in ``ws`` — what attn_tail() merges."""

    def __init__(self, ws, nsplit, Fdim, B):
        self.ws, self.nsplit, self.Fdim, self.B = ws, int(nsplit), int(Fdim), int(B)


def attn_tail_supported(EA: int, UE: int, N1: int, N2: int) -> bool:
    return bool(load_library().ncf_attn_tail_supported(int(EA), int(UE), int(N1), int(N2)))


class PackedTailWeight:
    """A hidden layer's (N, K) weight of the tail MLP in MFMA operand order (ncf_attn_tail_pack_weight), built once per weight version."""

    def __init__(self, W: torch.Tensor):
        lib = load_library()
        _dev(W, "W")
        if W.dtype != torch.float32 or W.dim() != 2 or not W.is_contiguous():
            raise TypeError("PackedTailWeight takes a contiguous fp32 (N, K) weight")
        self.shape = tuple(W.shape)
        self.data = torch.empty_like(W)
        _check(lib.ncf_attn_tail_pack_weight(_ptr(W), int(W.shape[0]), int(W.shape[1]), _ptr(self.data), _stream(W)))

    def is_contiguous(self):
        return True


def attn_tail(cand_emb: torch.Tensor, user, ubias: Optional[torch.Tensor], W1, b1, W2, b2, w3, b3: float) -> torch.Tensor:
    """ncf_attn_tail: merge of the attention partials (``user`` an AttnPartials; ``ubias`` = UserEmbeddings' bias) or finished user
    embeddings (``user`` a (B, UE) tensor, ``ubias`` None), cat(candidate_emb, user_emb), MLP -> (B, 1).  W1 / W2: both tensors in
    the reference's row-major layout, or both PackedTailWeight (the faster form, bit-identical)."""
    lib = load_library()
    _dev(cand_emb, "cand_emb")
    B, EA, ldc = _rows2d(cand_emb, "cand_emb")
    packed = isinstance(W1, PackedTailWeight)
    if packed != isinstance(W2, PackedTailWeight):
        raise TypeError("attn_tail: W1 and W2 must both be packed or both be plain tensors")
    N1, N2 = int(W1.shape[0]), int(W2.shape[0])
    out = torch.empty((B, 1), dtype=torch.float32, device=cand_emb.device)
    if isinstance(user, AttnPartials):
        part, ns, uptr, ldu, UE = user.ws, user.nsplit, None, 0, user.Fdim
    else:
        _, UE, ldu = _rows2d(user, "user_emb")
        part, ns, uptr = None, 1, user
    if W1.shape[1] != EA + UE or W2.shape[1] != N1 or w3.numel() != N2 or not (W1.is_contiguous() and W2.is_contiguous() and w3.is_contiguous()):
        raise ValueError("attn_tail: MLP weights do not match cat(candidate_emb, user_emb)")
    w1p, w2p = (W1.data, W2.data) if packed else (W1, W2)
    _check(lib.ncf_attn_tail(_ptr(cand_emb), ldc, EA, _ptr(part), ns, _ptr(uptr), ldu, UE, _ptr(ubias), _ptr(w1p), _ptr(b1), N1, _ptr(w2p), _ptr(b2), N2,
                             _ptr(w3), float(b3), 1 if packed else 0, _ptr(out), B, _stream(cand_emb)))
    return out


def attn_split_supported(mode: int, A: int, Fdim: int, pairs_per_wg: int) -> bool:
    return bool(load_library().ncf_attn_split_supported(int(mode), int(A), int(Fdim), int(pairs_per_wg)))


def default_attn_nsplit(B: int, n_rows: int, nnz: int, pairs_per_wg: int) -> int:
    """Slices of a rated set for the entry-split attention kernel, from sizes the host already holds (no device read): enough
    (group, slice) workgroups to give every CU two, at most one slice per 64-entry tile of an average row, at most 8."""
    if B <= 0 or n_rows <= 0:
        return 1
    groups = (B + pairs_per_wg - 1) // pairs_per_wg + min(n_rows, B) // 2          # between the bound and its half
    tiles = max(1, (nnz // max(n_rows, 1) + 63) // 64)
    want = (2 * num_cus() + groups - 1) // groups
    return int(max(1, min(want, tiles, 8)))


def num_cus() -> int:
    return 256


def attn_forward_grouped(mode: int, pc: torch.Tensor, pr: torch.Tensor, w1: Optional[torch.Tensor], b1: float,
                         rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, pair_row: torch.Tensor,
                         feat: torch.Tensor, out_bias: Optional[torch.Tensor] = None, pairs_per_wg: Optional[int] = None,
                         grouping=None, return_weights: bool = False, nsplit: Optional[int] = None, nnz_hint: Optional[int] = None,
                         leave_partials: bool = False):
    """LDS-tiled attention for pairs that share rated sets: CSR row ``pair_row[b]`` is pair b's set.  Returns out_feat
    (B, Fdim), or (out_feat, weights) with ``return_weights``: the attention weights in the layout of the expanded
    per-pair CSR (``SparseRatings.expanded()``).  ``grouping`` = (group_pairs(pair_row, R, ppw), ppw) computed earlier
    for this batch skips the sort.  Where ncf_attn_split_supported() holds (and no weights are asked for) the call takes the
    entry-split form (``nsplit`` slices of each rated set; default from the batch's sizes; 1 = one workgroup per group)."""
    lib = load_library()
    _dev(pc, "pc")
    B, A, ldpc = _rows2d(pc, "pc")
    I, A2, ldpr = _rows2d(pr, "pr")
    I2, Fdim, ldf = _rows2d(feat, "feat")
    if A != A2 or I != I2:
        raise ValueError("attention operand shapes disagree")
    if rowptr.dtype != torch.int64 or col.dtype != torch.int32 or val.dtype != torch.float32:
        raise TypeError("CSR must be (int64 rowptr, int32 col, fp32 val)")
    if pair_row.numel() != B:
        raise ValueError("pair_row must name one CSR row per pair")
    R = rowptr.numel() - 1
    if grouping is not None:
        grp, pairs_per_wg = grouping
    else:
        if pairs_per_wg is None:
            pairs_per_wg = default_pairs_per_wg(B)
        grp = group_pairs(pair_row.to(torch.int64).contiguous(), R, pairs_per_wg)
    grp_ptr, pair_ids, wg_ptr = grp
    out = torch.empty((B, Fdim), dtype=torch.float32, device=pc.device)
    wg_row = getattr(grp, "wg_row", None)
    if (not return_weights and get_option("attn_grouped_kernel") == 0
            and lib.ncf_attn_split_supported(mode, A, Fdim, int(pairs_per_wg)) and I > 0):
        # entry-split form (round 3): (group of pairs) x (slice of the rated set) workgroups + a merge of the softmax partials
        ns = int(nsplit) if nsplit is not None else default_attn_nsplit(B, R, col.numel() if nnz_hint is None else int(nnz_hint), int(pairs_per_wg))
        nbytes = lib.ncf_attn_split_workspace_bytes(B, Fdim, ns)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=pc.device) if ns > 1 else None
        _check(lib.ncf_attn_forward_split(mode, _ptr(pc), ldpc, _ptr(pr), ldpr, A, _ptr(w1), float(b1), _ptr(rowptr), _ptr(col),
                                          _ptr(val), R, I, _ptr(grp_ptr), _ptr(pair_ids), _ptr(wg_ptr), _ptr(wg_row), B,
                                          int(pairs_per_wg), _ptr(feat), ldf, Fdim, _ptr(out_bias), _ptr(out), out.stride(0),
                                          ns, 0 if (leave_partials and ns > 1) else 1, _ptr(ws), nbytes, _stream(pc)))
        if leave_partials and ns > 1:
            return AttnPartials(ws, ns, Fdim, B)
        return out
    wts = wts_off = None
    if return_weights:
        lens = (rowptr[1:] - rowptr[:-1])[pair_row.to(torch.int64)]
        wts_off = (torch.cumsum(lens, 0) - lens).contiguous()
        # expanded nnz without a host sync is not available: size by the upper bound B * (longest row) would be wasteful,
        # so this one size is read back (return_weights is the explanation / plotting path, not the scoring loop)
        wts = torch.empty(max(int(lens.sum().item()), 1), dtype=torch.float32, device=pc.device)
    _check(lib.ncf_attn_forward_grouped(mode, _ptr(pc), ldpc, _ptr(pr), ldpr, A, _ptr(w1), float(b1), _ptr(rowptr), _ptr(col),
                                        _ptr(val), R, I, _ptr(grp_ptr), _ptr(pair_ids), _ptr(wg_ptr), B, int(pairs_per_wg),
                                        _ptr(feat), ldf, Fdim, _ptr(out_bias), _ptr(out), out.stride(0), _ptr(wts), _ptr(wts_off),
                                        _stream(pc)))
    return (out, wts) if return_weights else out


def l2_normalize_rows(x: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    _dev(x, "x")
    R, E, ld = _rows2d(x, "x")
    out = torch.empty((R, E), dtype=torch.float32, device=x.device)
    _check(lib.ncf_l2_normalize_rows(_ptr(x), ld, R, E, _ptr(out), out.stride(0), _stream(x)))
    return out


def folded_supported(N1: int, N2: int) -> bool:
    return bool(load_library().ncf_score_folded_supported(NCF_F32, int(N1), int(N2)))


def score_folded(PA: torch.Tensor, idxA, PB: torch.Tensor, idxB, packed_tail: PackedMLP, out: Optional[torch.Tensor] = None,
                 B: Optional[int] = None) -> torch.Tensor:
    """Folded-first-layer scoring: relu(PA[idxA] + PB[idxB]) -> tail MLP [N1 -> N2 -> 1] (packed_tail)."""
    lib = load_library()
    _dev(PA, "PA"), _dev(PB, "PB")
    rowsA, N1, ldA = _rows2d(PA, "PA")
    rowsB, N1b, ldB = _rows2d(PB, "PB")
    if N1 != N1b or packed_tail.dims[0] != N1 or packed_tail.n_layers != 2:
        raise ValueError("folded tables and tail MLP disagree")
    idxA, idxB = _idx(idxA), _idx(idxB)
    if B is None:
        B = idxA.numel() if idxA is not None else (idxB.numel() if idxB is not None else rowsA)
    if out is None:
        out = torch.empty((B, 1), dtype=torch.float32, device=PA.device)
    _check(lib.ncf_score_folded(NCF_F32, _ptr(PA), rowsA, ldA, _ptr(PB), rowsB, ldB, _ptr(idxA), _ptr(idxB), B, N1,
                                packed_tail.dims[1], _ptr(packed_tail.blob), _ptr(out), _ptr(_oob_flag(PA.device)), _stream(PA)))
    return out


def edge_softmax_csr(rowptr: torch.Tensor, col: torch.Tensor, attr: Optional[torch.Tensor], s: torch.Tensor,
                     out: Optional[torch.Tensor] = None, segments=None) -> torch.Tensor:
    """Per-destination softmax of the source scores times the edge weight (LightGAT).  ``segments`` = (segptr, row_of, seg_first) of
    the SpMM's split rows: the segmented form (parallel over segments: hub rows do not serialise on one wave)."""
    lib = load_library()
    _dev(s, "s")
    if out is None:
        out = torch.empty(max(col.numel(), 1), dtype=torch.float32, device=s.device)[:col.numel()]
    if segments is not None and segments[1] is not None:
        segptr, row_of, seg_first = segments
        n_seg, n_rows = segptr.numel() - 1, rowptr.numel() - 1
        nbytes = lib.ncf_edge_softmax_segmented_workspace_bytes(n_seg, n_rows)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=s.device)
        _check(lib.ncf_edge_softmax_segmented(_ptr(segptr), _ptr(row_of), n_seg, _ptr(seg_first), n_rows, _ptr(col), _ptr(attr), _ptr(s), s.numel(),
                                              _ptr(out), _ptr(ws), nbytes, _stream(s)))
        return out
    _check(lib.ncf_edge_softmax_csr(_ptr(rowptr), _ptr(col), _ptr(attr), _ptr(s), rowptr.numel() - 1, s.numel(), _ptr(out), _stream(s)))
    return out


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """One layer with a selectable fused ReLU (the autograd wrappers keep every layer's output)."""
    lib = load_library()
    _dev(x, "x")
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise TypeError("linear_act needs fp32 activations and a contiguous fp32 weight")
    M, K, ldx = _rows2d(x, "x")
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError("weight / activation widths disagree")
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _check(lib.ncf_linear_forward(NCF_F32, _ptr(x), M, ldx, _ptr(weight), _ptr(bias), K, N, 1 if relu else 0, _ptr(out), N, _stream(x)))
    return out


# ------------------------------------------------------------------ backward (training step)
def gemm_tn(A: torch.Tensor, Bm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (N1, N2) = A^T @ Bm for A (M, N1), Bm (M, N2): the weight gradient dW = dY^T X."""
    lib = load_library()
    _dev(A, "A"), _dev(Bm, "B")
    M, N1, lda = _rows2d(A, "A")
    M2, N2, ldb = _rows2d(Bm, "B")
    if M != M2 or A.dtype != torch.float32 or Bm.dtype != torch.float32:
        raise ValueError("gemm_tn needs fp32 operands with the same number of rows")
    if out is None:
        out = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
    nb = lib.ncf_gemm_tn_workspace_bytes(M, N1, N2)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=A.device)
    _check(lib.ncf_gemm_tn(_ptr(A), lda, _ptr(Bm), ldb, M, N1, N2, _ptr(out), out.stride(0), _ptr(ws), nb, _stream(A)))
    return out


def colsum(X: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    _dev(X, "X")
    M, N, ld = _rows2d(X, "X")
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    nb = lib.ncf_colsum_workspace_bytes(M, N)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=X.device)
    _check(lib.ncf_colsum(_ptr(X), ld, M, N, _ptr(out), _ptr(ws), nb, _stream(X)))
    return out


def relu_backward_(dY: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    M, N, ldd = _rows2d(dY, "dY")
    _, _, ldy = _rows2d(Y, "Y")
    _check(lib.ncf_relu_backward(_ptr(dY), ldd, _ptr(Y), ldy, M, N, _stream(dY)))
    return dY


def relu_backward(dY: torch.Tensor, Y: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """scale * dY masked by Y > 0 into a NEW tensor (dY untouched)."""
    lib = load_library()
    _dev(dY, "dY"), _dev(Y, "Y")
    M, N, ldd = _rows2d(dY, "dY")
    M2, N2, ldy = _rows2d(Y, "Y")
    if (M, N) != (M2, N2):
        raise ValueError("dY and Y shapes disagree")
    out = torch.empty((M, N), dtype=torch.float32, device=dY.device)
    _check(lib.ncf_relu_backward_out(_ptr(dY), ldd, _ptr(Y), ldy, _ptr(out), N, M, N, float(scale), _stream(dY)))
    return out


def scatter_add_rows(src: torch.Tensor, idx: Optional[torch.Tensor], dst: torch.Tensor) -> torch.Tensor:
    """dst[idx[p], :] += src[p, :] (src may be a column slice of a wider matrix)."""
    lib = load_library()
    _dev(src, "src"), _dev(dst, "dst")
    B, E, lds = _rows2d(src, "src")
    rows, E2, ldd = _rows2d(dst, "dst")
    if E != E2:
        raise ValueError("row widths disagree")
    _check(lib.ncf_scatter_add_rows(_ptr(src), lds, _ptr(_idx(idx)), B, E, _ptr(dst), ldd, rows, _ptr(_oob_flag(src.device)), _stream(src)))
    return dst


# ------------------------------------------------------------------ training step: Linear-layout embeddings, Adam
def gather_cols(W: torch.Tensor, bias: Optional[torch.Tensor], idx: torch.Tensor) -> torch.Tensor:
    """out[p, :] = W[:, idx[p]] + bias  (W is the nn.Linear weight [E, U])."""
    lib = load_library()
    _dev(W, "W")
    E, U, ldw = _rows2d(W, "W")
    idx = _idx(idx)
    out = torch.empty((idx.numel(), E), dtype=torch.float32, device=W.device)
    _check(lib.ncf_gather_cols(_ptr(W), ldw, _ptr(bias), _ptr(idx), idx.numel(), E, U, _ptr(out), out.stride(0) if idx.numel() else E,
                               _ptr(_oob_flag(W.device)), _stream(W)))
    return out


def scatter_add_cols(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[:, idx[p]] += src[p, :]  (dst is [E, U])."""
    lib = load_library()
    _dev(src, "src")
    B, E, lds = _rows2d(src, "src")
    E2, U, ldd = _rows2d(dst, "dst")
    if E != E2:
        raise ValueError("scatter_add_cols: widths disagree")
    _check(lib.ncf_scatter_add_cols(_ptr(src), lds, _ptr(_idx(idx)), B, E, _ptr(dst), ldd, U, _ptr(_oob_flag(src.device)), _stream(src)))
    return dst


def adam_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int):
    lib = load_library()
    _dev(p, "p")
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adam_step_: contiguous fp32 tensors of one size")
    _check(lib.ncf_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                             float(weight_decay), int(step), _stream(p)))
