"""ctypes front-end of oracle/fv_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (fastvocoder_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "fv_oracle.c")
_LIB = os.path.join(_HERE, "libfv_oracle.so")

PAD_ZERO = 0
PAD_REFLECT = 1


def build(force=False):
    """gcc-compile the C restatement next to its source (idempotent)."""
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC)):
        return _LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _LIB, "-lm"]
    subprocess.check_call(cmd)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        fp = ctypes.POINTER(ctypes.c_float)
        i, f, l = ctypes.c_int, ctypes.c_float, ctypes.c_long
        L.fvo_conv1d.argtypes = [fp, fp, fp, fp, i, i, i, i, i, i, i, i, f]
        L.fvo_conv_transpose1d.argtypes = [fp, fp, fp, fp, i, i, i, i, i, i, i, i, f]
        L.fvo_weight_norm_fold.argtypes = [fp, fp, fp, i, l]
        L.fvo_pqmf_synthesis.argtypes = [fp, fp, fp, i, i, i, i]
        L.fvo_pqmf_analysis.argtypes = [fp, fp, fp, i, i, i, i]
        L.fvo_basis_ola.argtypes = [fp, fp, fp, i, i, i, i, i, f]
        L.fvo_tanh.argtypes = [fp, fp, l]
        for name in ("fvo_conv1d", "fvo_conv_transpose1d", "fvo_weight_norm_fold",
                     "fvo_pqmf_synthesis", "fvo_pqmf_analysis", "fvo_basis_ola", "fvo_tanh"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def conv1d(x, w, b, dil=1, pad=0, pad_mode=PAD_ZERO, pre_slope=1.0):
    """x [B,Cin,T], w [Cout,Cin,k] -> [B,Cout,Tout]."""
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    B, Cin, T = x.shape
    Cout, Cin2, k = w.shape
    assert Cin == Cin2
    Tout = T + 2 * pad - dil * (k - 1)
    y = np.empty((B, Cout, Tout), np.float32)
    lib().fvo_conv1d(_p(x), _p(w), _p(b), _p(y), B, Cin, Cout, T, k, dil, pad,
                     pad_mode, float(pre_slope))
    return y


def conv_transpose1d(x, w, b, stride, pad, out_pad, pre_slope=1.0):
    """x [B,Cin,T], w [Cin,Cout,k] -> [B,Cout,(T-1)s-2p+k+op]."""
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    B, Cin, T = x.shape
    Cin2, Cout, k = w.shape
    assert Cin == Cin2
    Tout = (T - 1) * stride - 2 * pad + k + out_pad
    y = np.empty((B, Cout, Tout), np.float32)
    lib().fvo_conv_transpose1d(_p(x), _p(w), _p(b), _p(y), B, Cin, Cout, T, k,
                               stride, pad, out_pad, float(pre_slope))
    return y


def weight_norm_fold(v, g):
    """v [dim0, ...], g [dim0, 1, ...] -> w, norm over all dims but 0."""
    v = _f32(v)
    g = _f32(g).reshape(-1)
    w = np.empty_like(v)
    lib().fvo_weight_norm_fold(_p(v), _p(g), _p(w), v.shape[0], int(np.prod(v.shape[1:])))
    return w


def pqmf_synthesis(x, h):
    """x [B,S,Tsub], h [S,ntaps] -> [B, S*Tsub]."""
    x, h = _f32(x), _f32(h)
    B, S, Tsub = x.shape
    y = np.empty((B, S * Tsub), np.float32)
    lib().fvo_pqmf_synthesis(_p(x), _p(h), _p(y), B, S, h.shape[1], Tsub)
    return y


def pqmf_analysis(xin, ha):
    """xin [B,T], ha [S,ntaps] -> [B,S,T//S]."""
    xin, ha = _f32(xin), _f32(ha)
    B, T = xin.shape
    S = ha.shape[0]
    Tsub = (T - S) // S + 1
    x = np.empty((B, S, Tsub), np.float32)
    lib().fvo_pqmf_analysis(_p(xin), _p(ha), _p(x), B, S, ha.shape[1], T)
    return x


def basis_ola(wt, W, hop, pre_slope=1.0):
    """wt [B,C,F] (trunk layout), W [L,C] -> [B,(F-1)hop+L]."""
    wt, W = _f32(wt), _f32(W)
    B, C, F = wt.shape
    L = W.shape[0]
    y = np.empty((B, (F - 1) * hop + L), np.float32)
    lib().fvo_basis_ola(_p(wt), _p(W), _p(y), B, C, F, L, hop, float(pre_slope))
    return y


def tanh(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().fvo_tanh(_p(x), _p(y), x.size)
    return y


def lrelu(x, slope):
    return np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)
