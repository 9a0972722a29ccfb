"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of ``oracle/hfem_oracle.c``.

numpy in, numpy out; fp64; full (assembled) node arrays.  Used by tests/ as the
fast checker at sizes where the autograd restatement (``ref_chain``) takes too
long, and by ``bench.py``'s ``cpu_baseline`` leg is NOT this file (that leg
times ``ref_chain``, the op-for-op port).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.oracle_tri3_energy.restype = C.c_double
        _lib.oracle_tri3_energy.argtypes = [_dp, _dp, _ip, C.c_int64, _dp, C.c_double, _dp, _dp, _dp]
        _lib.oracle_quad4_energy.restype = C.c_double
        _lib.oracle_quad4_energy.argtypes = [_dp, _dp, _ip, C.c_int64, _dp, _dp, _dp, _dp]
        _lib.oracle_edge2_energy.restype = C.c_double
        _lib.oracle_edge2_energy.argtypes = [_dp, _dp, _ip, C.c_int64, _dp, _dp, _dp, _dp]
        _lib.oracle_tri3_eval.restype = None
        _lib.oracle_tri3_eval.argtypes = [_dp, _dp, _ip, _dp, _ip, C.c_int64, _dp, _dp, _dp]
        _lib.oracle_grid_param_fwd.restype = None
        _lib.oracle_grid_param_fwd.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, _dp]
        _lib.oracle_grid_param_bwd.restype = None
        _lib.oracle_grid_param_bwd.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, _dp, _dp]
        _lib.oracle_line2.restype = None
        _lib.oracle_line2.argtypes = [_dp, _dp, C.c_int64, _dp, C.c_int64, _dp, _dp, _dp, _dp, _dp]
        _lib.oracle_bar_energy.restype = C.c_double
        _lib.oracle_bar_energy.argtypes = [_dp, _dp, C.c_int64, _dp, _dp, _dp, C.c_int64, C.c_double, _dp, _dp]
        _lib.oracle_rectq4.restype = None
        _lib.oracle_rectq4.argtypes = [_dp, C.c_int64, _dp, C.c_int64, _dp, _dp, C.c_int64,
                                       _dp, _dp, _dp, _dp, _dp, _dp]
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def plane_stress(E=10e9, nu=0.3):
    """{c11, c12, c22, c33} of loss.py:29-32, multiplied in fp64 as torch does."""
    f = E / (1 - nu ** 2)
    return np.array([1.0 * f, nu * f, 1.0 * f, (1.0 - nu) / 2.0 * f])


def tri3_energy(X, U, conn, mat, W, Bk=None, grads=True):
    """-> (domain energy, gX[Nn,2], gU[Nn,2])."""
    X, U = _f64(X), _f64(U)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    Bk = np.zeros(6) if Bk is None else _f64(Bk).reshape(6)
    gX = np.zeros_like(X) if grads else None
    gU = np.zeros_like(U) if grads else None
    e = lib().oracle_tri3_energy(_d(X), _d(U), _i(conn), conn.shape[0], _d(_f64(mat)), float(W),
                                 _d(Bk), _d(gX), _d(gU))
    return e, gX, gU


def quad4_energy(X, U, conn4, mat, Bq=None, grads=True):
    """QUAD4-iso extension (parity unpinned by the reference) -> (domain energy, gX, gU);
    ``Bq [4,2]``: body force at the 2x2 Gauss points (reference coordinates, order (-,-) (+,-) (-,+) (+,+))."""
    X, U = _f64(X), _f64(U)
    conn4 = np.ascontiguousarray(conn4, dtype=np.int64)
    Bq = None if Bq is None else _f64(Bq).reshape(8)
    gX = np.zeros_like(X) if grads else None
    gU = np.zeros_like(U) if grads else None
    e = lib().oracle_quad4_energy(_d(X), _d(U), _i(conn4), conn4.shape[0], _d(_f64(mat)), _d(Bq), _d(gX), _d(gU))
    return e, gX, gU


def edge2_energy(X, U, edges, T=None, Tconst=None, gX=None, gU=None):
    """-> edge work (to subtract); gradient of (-work) ACCUMULATED into gX/gU if given."""
    X, U = _f64(X), _f64(U)
    edges = np.ascontiguousarray(edges, dtype=np.int64)
    T = None if T is None else _f64(T).reshape(-1, 4)
    Tc = None if Tconst is None else _f64(Tconst).reshape(4)
    return lib().oracle_edge2_energy(_d(X), _d(U), _i(edges), edges.shape[0], _d(T), _d(Tc), _d(gX), _d(gU))


def tri3_eval(X, U, conn, x_eval, elem_id):
    X, U, x_eval = _f64(X), _f64(U), _f64(x_eval)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    elem_id = np.ascontiguousarray(elem_id, dtype=np.int64)
    m = elem_id.shape[0]
    u_h, detJ, grad_u = np.empty((m, 2)), np.empty(m), np.empty((m, 2, 2))
    lib().oracle_tri3_eval(_d(X), _d(U), _i(conn), _d(x_eval), _i(elem_id), m, _d(u_h), _d(detJ), _d(grad_u))
    return u_h, detJ, grad_u


def grid_param_fwd(p, x0, xN):
    p = _f64(p)
    grid = np.empty(p.shape[0] + 1)
    lib().oracle_grid_param_fwd(_d(p), p.shape[0], float(x0), float(xN), _d(grid))
    return grid


def grid_param_bwd(p, x0, xN, ggrid):
    p, ggrid = _f64(p), _f64(ggrid)
    gp = np.empty_like(p)
    lib().oracle_grid_param_bwd(_d(p), p.shape[0], float(x0), float(xN), _d(ggrid), _d(gp))
    return gp


def line2(grid, u, x_eval, cot=None):
    """-> pred, (ggrid, gu, gx_eval) if cot is given."""
    grid, u, x_eval = _f64(grid), _f64(u), _f64(x_eval).ravel()
    pred = np.empty_like(x_eval)
    if cot is None:
        lib().oracle_line2(_d(grid), _d(u), grid.shape[0], _d(x_eval), x_eval.shape[0], _d(pred),
                           None, None, None, None)
        return pred
    cot = _f64(cot).ravel()
    gg, gu, gx = np.zeros_like(grid), np.zeros_like(u), np.empty_like(x_eval)
    lib().oracle_line2(_d(grid), _d(u), grid.shape[0], _d(x_eval), x_eval.shape[0], _d(pred),
                       _d(cot), _d(gg), _d(gu), _d(gx))
    return pred, gg, gu, gx


def bar_energy(grid, u, xq, wq, bq, E, grads=True):
    grid, u = _f64(grid), _f64(u)
    xq, wq, bq = _f64(xq).ravel(), _f64(wq).ravel(), _f64(bq).ravel()
    gg = np.zeros_like(grid) if grads else None
    gu = np.zeros_like(u) if grads else None
    e = lib().oracle_bar_energy(_d(grid), _d(u), grid.shape[0], _d(xq), _d(wq), _d(bq), xq.shape[0],
                                float(E), _d(gg), _d(gu))
    return e, gg, gu


def rectq4(gx, gy, u, x_eval, cot=None):
    gx, gy, u, x_eval = _f64(gx), _f64(gy), _f64(u), _f64(x_eval)
    m = x_eval.shape[0]
    pred = np.empty(m)
    if cot is None:
        lib().oracle_rectq4(_d(gx), gx.shape[0], _d(gy), gy.shape[0], _d(u), _d(x_eval), m, _d(pred),
                            None, None, None, None, None)
        return pred
    cot = _f64(cot)
    ggx, ggy, gu, gxe = np.zeros_like(gx), np.zeros_like(gy), np.zeros_like(u), np.empty_like(x_eval)
    lib().oracle_rectq4(_d(gx), gx.shape[0], _d(gy), gy.shape[0], _d(u), _d(x_eval), m, _d(pred),
                        _d(cot), _d(ggx), _d(ggy), _d(gu), _d(gxe))
    return pred, ggx, ggy, gu, gxe
