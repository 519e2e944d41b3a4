"""ctypes wrapper around oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def msda_forward(value, shapes, level_start, loc, attn, acc_double: bool = False) -> np.ndarray:
    value, pv = _f(value); loc, pl = _f(loc); attn, pa = _f(attn)
    shapes, ps = _i64(shapes); level_start, pst = _i64(level_start)
    B, S, H, D = value.shape
    Nq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = np.empty((B, Nq, H * D), dtype=np.float32)
    rc = lib().oracle_msda_forward_f32(pv, ps, pst, pl, pa, B, S, H, D, L, Nq, P, int(acc_double),
                                       out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def msda_backward(value, shapes, level_start, loc, attn, grad_out):
    value, pv = _f(value); loc, pl = _f(loc); attn, pa = _f(attn); grad_out, pg = _f(grad_out)
    shapes, ps = _i64(shapes); level_start, pst = _i64(level_start)
    B, S, H, D = value.shape
    Nq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    gv = np.empty_like(value); gl = np.empty_like(loc); ga = np.empty_like(attn)
    rc = lib().oracle_msda_backward_f32(pv, ps, pst, pl, pa, pg, B, S, H, D, L, Nq, P,
                                        gv.ctypes.data_as(ctypes.c_void_p), gl.ctypes.data_as(ctypes.c_void_p),
                                        ga.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return gv, gl, ga


def relation_bias(src, tgt, Wp, bp, num_pos_feats=16, scale=100.0, temperature=10000.0, eps=1e-5) -> np.ndarray:
    src, p1 = _f(src); tgt, p2 = _f(tgt)
    Wp, pw = _f(np.asarray(Wp).reshape(np.asarray(Wp).shape[0], -1)); bp, pb = _f(bp)
    B, N1, _ = src.shape
    N2 = tgt.shape[1]
    Hh = Wp.shape[0]
    out = np.empty((B, Hh, N1, N2), dtype=np.float32)
    rc = lib().oracle_relation_bias_f32(p1, p2, pw, pb, B, N1, N2, Hh, num_pos_feats, ctypes.c_float(scale),
                                        ctypes.c_float(temperature), ctypes.c_float(eps),
                                        out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def bias_softmax(scores, bias=None, mask=None) -> np.ndarray:
    scores = np.array(scores, dtype=np.float32, order="C", copy=True)
    BH, N1, N2 = scores.shape
    pb = None
    if bias is not None:
        bias, pb = _f(bias)
    pm = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        pm = mask.ctypes.data_as(ctypes.c_void_p)
    rc = lib().oracle_bias_softmax_f32(scores.ctypes.data_as(ctypes.c_void_p), pb, pm, BH, N1, N2)
    assert rc == 0, rc
    return scores
