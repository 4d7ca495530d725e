"""ctypes loader for oracle/slfp_oracle.c -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy, layouts are the reference's (NCHW activations, OIHW weights).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libslfp_oracle.so")

FMT_ACT8, FMT_W8, FMT_SFP7, FMT_EXT = 0, 1, 2, 4

_lib = None


def build(force=False):
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "slfp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        bp = ctypes.POINTER(ctypes.c_uint8)
        L.slfp_oracle_quantize.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_float, ctypes.c_int]
        L.slfp_oracle_quantize.restype = None
        L.slfp_oracle_encode.argtypes = [fp, bp, ctypes.c_size_t, ctypes.c_float, ctypes.c_int]
        L.slfp_oracle_encode.restype = None
        L.slfp_oracle_layerout.argtypes = [fp, fp, ctypes.c_size_t]
        L.slfp_oracle_layerout.restype = None
        L.slfp_oracle_decode.argtypes = [bp, fp, ctypes.c_size_t, ctypes.c_int]
        L.slfp_oracle_decode.restype = None
        i64, ci = ctypes.c_int64, ctypes.c_int
        L.slfp_oracle_conv2d.argtypes = [fp, i64, i64, i64, i64, fp, i64, i64, i64, fp,
                                         ci, ci, ci, ci, ci, ci, ci,
                                         ctypes.c_float, ctypes.c_float, ci, fp, fp, fp, fp]
        L.slfp_oracle_conv2d.restype = ci
        L.slfp_oracle_linear.argtypes = [fp, i64, i64, fp, i64, fp, ctypes.c_float,
                                         ctypes.c_float, ci, fp, fp]
        L.slfp_oracle_linear.restype = ci
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty)) if a is not None else None


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def quantize(x, scale_div, fmt):
    """Q_fmt(x / scale_div) as float32 -- what the reference's qfn.forward returns."""
    x = _f32(x)
    y = np.empty_like(x)
    lib().slfp_oracle_quantize(_p(x), _p(y), x.size, np.float32(scale_div), fmt)
    return y


def layerout(x):
    """quantize_layerout(k <= 8).forward (utils/sfp_quant.py:108-127), SFP<4,4> with the reference's quirks."""
    x = _f32(x)
    y = np.empty_like(x)
    lib().slfp_oracle_layerout(_p(x), _p(y), x.size)
    return y


def encode(x, scale_div, fmt):
    x = _f32(x)
    c = np.empty(x.shape, dtype=np.uint8)
    lib().slfp_oracle_encode(_p(x), _p(c, ctypes.c_uint8), x.size, np.float32(scale_div), fmt)
    return c


def decode(code, fmt):
    code = np.ascontiguousarray(code, dtype=np.uint8)
    y = np.empty(code.shape, dtype=np.float32)
    lib().slfp_oracle_decode(_p(code, ctypes.c_uint8), _p(y), code.size, fmt)
    return y


def conv2d(x, w, bias, stride, padding, dilation, groups, Ka, Kw, qbits, want_q=False):
    """Conv2d_Q.forward (utils/conv2d_func.py:20-25 / :41-47) on NCHW / OIHW numpy arrays."""
    x, w = _f32(x), _f32(w)
    b = _f32(bias) if bias is not None else None
    N, C, H, W = x.shape
    O, Cg, KH, KW = w.shape
    assert Cg * groups == C, (x.shape, w.shape, groups)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    y = np.empty((N, O, Ho, Wo), dtype=np.float32)
    xq = np.empty_like(x)
    wq = np.empty_like(w)
    rc = lib().slfp_oracle_conv2d(_p(x), N, C, H, W, _p(w), O, KH, KW, _p(b), sh, sw, ph, pw,
                                  dh, dw, groups, np.float32(Ka), np.float32(Kw), qbits,
                                  _p(y), _p(xq), _p(wq), None)
    if rc != 0:
        raise ValueError("slfp_oracle_conv2d: bad arguments")
    return (y, xq, wq) if want_q else y


def linear(x, w, bias, Ka, Kw, qbits):
    """Linear_Q.forward (utils/conv2d_func.py:60-65)."""
    x, w = _f32(x), _f32(w)
    b = _f32(bias) if bias is not None else None
    B, I = x.shape
    O = w.shape[0]
    y = np.empty((B, O), dtype=np.float32)
    scratch = np.empty(B * I + O * I, dtype=np.float32)
    rc = lib().slfp_oracle_linear(_p(x), B, I, _p(w), O, _p(b), np.float32(Ka), np.float32(Kw),
                                  qbits, _p(y), _p(scratch))
    if rc != 0:
        raise ValueError("slfp_oracle_linear: bad arguments")
    return y
