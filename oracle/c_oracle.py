"""ctypes binding of oracle/libncf_oracle_c.so (the kernel-order C restatement) — TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libncf_oracle_c.so")


def _load():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "ncf_oracle_c.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    lib = ctypes.CDLL(_LIB)
    p, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.oracle_score_fused_f32.restype = ci
    lib.oracle_score_fused_f32.argtypes = [p, i64, p, i64, p, p, i64, ci, ci, p, p, ci, p, p, ci, p, p, p]
    return lib


def score_fused_f32(tabA, tabB, idxA, idxB, weights, biases):
    """Same arguments as the table-form forward; weights/biases are the MLP's (W, b) in order, last layer 1 wide."""
    lib = _load()
    c = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())
    ta, tb = c(tabA.float()), c(tabB.float())
    ia, ib = c(idxA.long()), c(idxB.long())
    ws = [c(w.float()) for w in weights]
    bs = [c(b.float()) for b in biases]
    B = ia.shape[0]
    out = np.empty(B, dtype=np.float32)
    two = len(ws) == 3
    W2 = ws[1] if two else np.zeros(1, np.float32)
    b2 = bs[1] if two else np.zeros(1, np.float32)
    rc = lib.oracle_score_fused_f32(ta.ctypes.data, ta.shape[1], tb.ctypes.data, tb.shape[1], ia.ctypes.data, ib.ctypes.data, B,
                                    ta.shape[1], tb.shape[1], ws[0].ctypes.data, bs[0].ctypes.data, ws[0].shape[0],
                                    W2.ctypes.data, b2.ctypes.data, ws[1].shape[0] if two else 0,
                                    ws[-1].ctypes.data, bs[-1].ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("shape not tileable like the fused kernel")
    return torch.from_numpy(out).view(-1, 1)
