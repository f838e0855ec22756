"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so (the C restatement of the reference's PCG,
see pcg_oracle.c) plus small numpy helpers used to pin it: dense assembly of the
compressed block-tridiagonal layout and dense fp64 solves.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  Nothing under gbd-pcg_amd/ does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

FMA = 1
TREE = 2
DEFAULT = FMA | TREE

_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


_CT = {np.dtype(np.float32): (ctypes.c_float, "f32"), np.dtype(np.float64): (ctypes.c_double, "f64")}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def spmv(n, N, M, x, flags=DEFAULT, batch=1, nthreads=1):
    """y = M x, block-tridiagonal [L|D|R] column-major layout (utils.cuh:46-85)."""
    dtype = np.dtype(M.dtype)
    cty, suf = _CT[dtype]
    M = _c(M, dtype).reshape(-1)
    x = _c(x, dtype).reshape(-1)
    assert M.size == batch * 3 * n * n * N and x.size == batch * n * N
    y = np.empty_like(x)
    fn = getattr(lib(), f"oracle_spmv_batch_{suf}")
    rc = fn(ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch), _ptr(M), _ptr(x),
            _ptr(y), ctypes.c_int(flags), ctypes.c_int(nthreads))
    assert rc == 0
    return y


def pcg(n, N, S, Pinv, gamma, lambda0=None, tol=1e-6, max_iter=25, flags=DEFAULT, trace=False):
    """Single-problem solve.  Returns dict(lambda_, r, p, iters, max_iter_exit[, eta])."""
    dtype = np.dtype(S.dtype)
    cty, suf = _CT[dtype]
    S = _c(S, dtype).reshape(-1)
    Pinv = None if Pinv is None else _c(Pinv, dtype).reshape(-1)
    gamma = _c(gamma, dtype).reshape(-1)
    lam = np.zeros(n * N, dtype) if lambda0 is None else np.array(lambda0, dtype=dtype).reshape(-1).copy()
    assert S.size == 3 * n * n * N and gamma.size == n * N and lam.size == n * N
    r = np.empty(n * N, dtype)
    p = np.empty(n * N, dtype)
    iters = ctypes.c_uint32(0)
    mie = ctypes.c_uint8(0)
    eta = np.full(max_iter + 1, np.nan, dtype) if trace else None
    fn = getattr(lib(), f"oracle_pcg_{suf}")
    rc = fn(ctypes.c_uint32(n), ctypes.c_uint32(N), _ptr(S), _ptr(Pinv), _ptr(gamma), _ptr(lam),
            _ptr(r), _ptr(p), cty(tol), ctypes.c_uint32(max_iter), ctypes.byref(iters),
            ctypes.byref(mie), _ptr(eta), ctypes.c_int(flags))
    assert rc == 0
    out = dict(lambda_=lam, r=r, p=p, iters=int(iters.value), max_iter_exit=bool(mie.value))
    if trace:
        out["eta"] = eta
    return out


def pcg_batch(n, N, batch, S, Pinv, gamma, lambda0=None, tol=1e-6, max_iter=25, flags=DEFAULT,
              nthreads=1):
    dtype = np.dtype(S.dtype)
    cty, suf = _CT[dtype]
    S = _c(S, dtype).reshape(-1)
    Pinv = None if Pinv is None else _c(Pinv, dtype).reshape(-1)
    gamma = _c(gamma, dtype).reshape(-1)
    lam = (np.zeros(batch * n * N, dtype) if lambda0 is None
           else np.array(lambda0, dtype=dtype).reshape(-1).copy())
    assert S.size == batch * 3 * n * n * N and gamma.size == batch * n * N
    r = np.empty(batch * n * N, dtype)
    p = np.empty(batch * n * N, dtype)
    iters = np.zeros(batch, np.uint32)
    mie = np.zeros(batch, np.uint8)
    fn = getattr(lib(), f"oracle_pcg_batch_{suf}")
    rc = fn(ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch), _ptr(S), _ptr(Pinv),
            _ptr(gamma), _ptr(lam), _ptr(r), _ptr(p), cty(tol), ctypes.c_uint32(max_iter),
            _ptr(iters), _ptr(mie), ctypes.c_int(flags), ctypes.c_int(nthreads))
    assert rc == 0
    return dict(lambda_=lam.reshape(batch, -1), r=r.reshape(batch, -1), p=p.reshape(batch, -1),
                iters=iters, max_iter_exit=mie.astype(bool))


# --------------------------------------------------------------------------- numpy helpers
def dense_from_bt(n, N, M):
    """Dense nN x nN fp64 matrix of one problem in the [L|D|R] column-major layout.

    Element (r,c) of block b of block-row k sits at k*3n^2 + b*n^2 + c*n + r
    (pcg.cuh:104-110, utils.cuh:80); L_0 and R_{N-1} are ignored, as the kernel
    never reads them (pcg.cuh:105-106).
    """
    blk = np.asarray(M, dtype=np.float64).reshape(N, 3, n, n).transpose(0, 1, 3, 2)  # [k,b,r,c]
    A = np.zeros((n * N, n * N))
    for k in range(N):
        for b in range(3):
            kc = k + b - 1
            if 0 <= kc < N:
                A[k * n:(k + 1) * n, kc * n:(kc + 1) * n] = blk[k, b]
    return A


def readme_system(dtype=np.float64):
    """The one fixture the reference holds: the n=2, N=3 INPUT system of
    examples/pcg_solve.cu:14-25 (identical in pcg_solve_dp.cu:14-25)."""
    S = np.array([0, 0, 0, 0,
                  -.999, 0, 0, -.999,
                  .999, .0999, -.98, .999,
                  .999, -.98, .0999, .999,
                  -2.008, .8801, .8801, -3.0584,
                  .999, .0999, -.98, .999,
                  .999, -.98, .0999, .999,
                  -1.019, .8801, .8801, -2.0694,
                  0, 0, 0, 0], dtype=dtype)
    gamma = np.array([3.1385, 0, 0, 3.0788, .0031, 3.0788], dtype=dtype)
    return 2, 3, S, gamma
