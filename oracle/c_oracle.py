"""ctypes bindings of the C oracle (oracle/ceeg_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libceeg_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ceeg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libceeg_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def vq_forward_f32(z, W, beta):
    """oracle_vq_forward_f32 -> dict(z_q, idx, loss, perplexity, hist, dmin)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    B, D = z.shape
    n_e = W.shape[0]
    assert W.shape[1] == D
    z_q = np.empty_like(z)
    idx = np.empty(B, dtype=np.int64)
    hist = np.empty(n_e, dtype=np.uint32)
    dmin = np.empty(B, dtype=np.float32)
    loss = C.c_float()
    ppl = C.c_float()
    lib().oracle_vq_forward_f32(
        _p(z, C.c_float), _p(W, C.c_float), C.c_int64(B), C.c_int(n_e), C.c_int(D), C.c_float(beta),
        _p(z_q, C.c_float), _p(idx, C.c_int64), C.byref(loss), C.byref(ppl), _p(hist, C.c_uint32),
        _p(dmin, C.c_float))
    return dict(z_q=z_q, idx=idx, loss=np.float32(loss.value), perplexity=np.float32(ppl.value),
                hist=hist, dmin=dmin)


def philox_normal(seed, n):
    eps = np.empty(n, dtype=np.float32)
    lib().oracle_philox_normal(C.c_uint64(seed), C.c_int64(n), _p(eps, C.c_float))
    return eps


def philox_raw(seed, ngroups, tag):
    out = np.empty(ngroups * 4, dtype=np.uint32)
    lib().oracle_philox_raw(C.c_uint64(seed), C.c_int64(ngroups), C.c_uint32(tag), _p(out, C.c_uint32))
    return out


def neg_sample_uniform(rowptr, col, N, num, seed):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    s = np.empty(num, dtype=np.int32)
    d = np.empty(num, dtype=np.int32)
    lib().oracle_neg_sample_uniform(_p(rowptr, C.c_int32), _p(col, C.c_int32), C.c_int64(N), C.c_int64(num),
                                    C.c_uint64(seed), _p(s, C.c_int32), _p(d, C.c_int32))
    return s, d


def mfma_bf16_dot(a_bits, b_bits, c):
    """One dot product as v_mfma_f32_16x16x32_bf16 computes it (chained 8-product steps): a, b uint16 bf16 bit patterns."""
    a = np.ascontiguousarray(a_bits, dtype=np.uint16)
    b = np.ascontiguousarray(b_bits, dtype=np.uint16)
    out = C.c_float()
    lib().oracle_mfma_bf16_dot(_p(a, C.c_uint16), _p(b, C.c_uint16), C.c_int(a.shape[0]), C.c_float(c), C.byref(out))
    return np.float32(out.value)


def vq_forward_bf16(z_bits, W, beta):
    """oracle_vq_forward_bf16: z as uint16 bf16 bit patterns [B, D], W fp32 -> dict(z_q (bits), idx, loss, perplexity, hist, best)."""
    z = np.ascontiguousarray(z_bits, dtype=np.uint16)
    W = np.ascontiguousarray(W, dtype=np.float32)
    B, D = z.shape
    n_e = W.shape[0]
    assert W.shape[1] == D
    z_q = np.empty_like(z)
    idx = np.empty(B, dtype=np.int64)
    hist = np.empty(n_e, dtype=np.uint32)
    best = np.empty(B, dtype=np.float32)
    loss = C.c_float()
    ppl = C.c_float()
    lib().oracle_vq_forward_bf16(
        _p(z, C.c_uint16), _p(W, C.c_float), C.c_int64(B), C.c_int(n_e), C.c_int(D), C.c_float(beta),
        _p(z_q, C.c_uint16), _p(idx, C.c_int64), C.byref(loss), C.byref(ppl), _p(hist, C.c_uint32), _p(best, C.c_float))
    return dict(z_q=z_q, idx=idx, loss=np.float32(loss.value), perplexity=np.float32(ppl.value), hist=hist, best=best)
