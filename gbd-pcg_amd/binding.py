"""ctypes view of libgbdpcg.so (the C ABI of include/gbdpcg.h) over torch device tensors.

Plumbing for tests and bench.py only: torch supplies device memory and streams, every
computation goes through the C ABI into the hand-written HIP kernels.  There is NO fallback:
a missing library raises at load(), and Solver() raises without a gfx950 device.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# GBDPCG_LIB: alternative build of the same library (A/B tuning runs on one device)
LIB_PATH = os.environ.get("GBDPCG_LIB") or os.path.join(CSRC, "libgbdpcg.so")

OK = 0
PATH_AUTO, PATH_FUSED, PATH_SPLIT, PATH_PERSISTENT, PATH_PERSISTENT_1R = 0, 1, 2, 3, 4
PINV_IDENTITY, PINV_BLOCK_JACOBI, PINV_STAIR = 0, 1, 2

# every symbol include/gbdpcg.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "gbdpcg_create", "gbdpcg_destroy", "gbdpcg_status_string", "gbdpcg_last_hip_error",
    "gbdpcg_last_hip_error_string", "gbdpcg_set_path", "gbdpcg_choose_path", "gbdpcg_cluster_members", "gbdpcg_set_symmetric",
    "gbdpcg_check_symmetric_f32", "gbdpcg_check_symmetric_f64",
    "gbdpcg_pcg_shared_mem_size", "gbdpcg_check_occupancy", "gbdpcg_workspace_bytes",
    "gbdpcg_reserve", "gbdpcg_spmv_f32", "gbdpcg_spmv_f64", "gbdpcg_solve_f32", "gbdpcg_solve_f64",
    "gbdpcg_solve_blocking_f32", "gbdpcg_solve_blocking_f64", "gbdpcg_solve_host_f32",
    "gbdpcg_solve_host_f64", "gbdpcg_graph_create_solve_f32", "gbdpcg_graph_create_solve_f64",
    "gbdpcg_graph_launch", "gbdpcg_graph_destroy", "gbdpcg_form_pinv_f32", "gbdpcg_form_pinv_f64",
    "gbdpcg_form_pinv_solve_f32", "gbdpcg_form_pinv_solve_f64",
    "gbdpcg_graph_create_form_pinv_solve_f32", "gbdpcg_graph_create_form_pinv_solve_f64",
    "gbdpcg_form_schur_f32", "gbdpcg_form_schur_f64", "gbdpcg_recover_primal_f32", "gbdpcg_recover_primal_f64",
    "gbdpcg_kkt_step_f32", "gbdpcg_kkt_step_f64", "gbdpcg_graph_create_kkt_step_f32", "gbdpcg_graph_create_kkt_step_f64",
    "gbdpcg_csr_to_bt_f32", "gbdpcg_csr_to_bt_f64", "gbdpcg_version",
]

_lib = None


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C gbd-pcg_amd/csrc` "
                               "(there is no CPU fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        lib.gbdpcg_status_string.restype = ctypes.c_char_p
        lib.gbdpcg_last_hip_error_string.restype = ctypes.c_char_p
        lib.gbdpcg_version.restype = ctypes.c_char_p
        lib.gbdpcg_pcg_shared_mem_size.restype = ctypes.c_size_t
        lib.gbdpcg_workspace_bytes.restype = ctypes.c_size_t
        _lib = lib
    return _lib


class GbdPcgError(RuntimeError):
    pass


def _suffix(t):
    import torch
    if t.dtype == torch.float32:
        return "f32", ctypes.c_float
    if t.dtype == torch.float64:
        return "f64", ctypes.c_double
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class Solver:
    """One gbdpcg handle on one device."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.h = ctypes.c_void_p()
        st = self.lib.gbdpcg_create(ctypes.byref(self.h), ctypes.c_int(device))
        if st != OK:
            raise GbdPcgError(f"gbdpcg_create: {self.lib.gbdpcg_status_string(st).decode()}")
        self.device = device

    def close(self):
        if self.h:
            self.lib.gbdpcg_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, what):
        if st != OK:
            msg = self.lib.gbdpcg_status_string(st).decode()
            hip = self.lib.gbdpcg_last_hip_error_string(self.h).decode()
            raise GbdPcgError(f"{what}: {msg} (status {st}; last HIP error: {hip})")

    @staticmethod
    def _stream(stream):
        import torch
        s = torch.cuda.current_stream() if stream is None else stream
        return ctypes.c_void_p(s.cuda_stream)

    def set_path(self, path: int):
        self._check(self.lib.gbdpcg_set_path(self.h, ctypes.c_int(path)), "set_path")

    def set_symmetric(self, mode):
        """0 / False: never, 1 / True: assume, 2: test on the device before every solve (the default)."""
        self._check(self.lib.gbdpcg_set_symmetric(self.h, ctypes.c_int(int(mode))), "set_symmetric")

    def check_symmetric(self, n, N, batch, M, stream=None):
        """uint8 tensor [batch]: 1 where L_{k+1} == R_k^T bit for bit for every knot."""
        import torch
        suf, _ = _suffix(M)
        flags = torch.empty(batch, dtype=torch.uint8, device=M.device)
        fn = getattr(self.lib, f"gbdpcg_check_symmetric_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(M), _p(flags),
                       self._stream(stream)), "check_symmetric")
        return flags

    def choose_path(self, elem_size, n, N, batch) -> int:
        return self.lib.gbdpcg_choose_path(self.h, ctypes.c_uint32(elem_size), ctypes.c_uint32(n),
                                           ctypes.c_uint32(N), ctypes.c_uint32(batch))

    def cluster_members(self, elem_size, n, N) -> int:
        return self.lib.gbdpcg_cluster_members(ctypes.c_uint32(elem_size), ctypes.c_uint32(n), ctypes.c_uint32(N))

    def reserve(self, elem_size, n, N, batch):
        self._check(self.lib.gbdpcg_reserve(self.h, ctypes.c_uint32(elem_size), ctypes.c_uint32(n),
                                            ctypes.c_uint32(N), ctypes.c_uint32(batch)), "reserve")

    def spmv(self, n, N, batch, M, x, y=None, stream=None):
        import torch
        suf, _ = _suffix(M)
        assert M.is_cuda and M.is_contiguous() and x.is_contiguous()
        assert M.numel() == batch * 3 * n * n * N and x.numel() == batch * n * N
        if y is None:
            y = torch.empty_like(x)
        fn = getattr(self.lib, f"gbdpcg_spmv_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch),
                       _p(M), _p(x), _p(y), self._stream(stream)), "spmv")
        return y

    def solve_args(self, n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters, mie):
        suf, cty = _suffix(S)
        for t, cnt in ((S, 3 * n * n * N), (Pinv, 3 * n * n * N), (gamma, n * N), (lam, n * N),
                       (r, n * N), (p, n * N)):
            if t is not None:
                assert t.is_cuda and t.is_contiguous() and t.numel() == batch * cnt and t.dtype == S.dtype
        assert iters.numel() == batch and mie is None or mie.numel() == batch
        return suf, (self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(S),
                     _p(Pinv), _p(gamma), _p(lam), _p(r), _p(p), cty(tol), ctypes.c_uint32(max_iter),
                     _p(iters), _p(mie))

    def solve(self, n, N, batch, S, Pinv, gamma, lam, r=None, p=None, tol=1e-6, max_iter=25,
              iters=None, max_iter_exit=None, stream=None):
        """Asynchronous batched solve (gbdpcg_solve_*).  lam is in/out.  Returns (iters, flags)."""
        import torch
        if iters is None:
            iters = torch.empty(batch, dtype=torch.int32, device=S.device)
        if max_iter_exit is None:
            max_iter_exit = torch.empty(batch, dtype=torch.uint8, device=S.device)
        suf, args = self.solve_args(n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters,
                                    max_iter_exit)
        fn = getattr(self.lib, f"gbdpcg_solve_{suf}")
        self._check(fn(*args, self._stream(stream)), "solve")
        return iters, max_iter_exit

    def solve_blocking(self, n, N, S, Pinv, gamma, lam, r=None, p=None, tol=1e-6, max_iter=25):
        """gbdpcg_solve_blocking_* : the reference's device-pointer overload. Returns (iters, flag)."""
        suf, cty = _suffix(S)
        it = ctypes.c_uint32(0)
        fl = ctypes.c_uint8(0)
        fn = getattr(self.lib, f"gbdpcg_solve_blocking_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), _p(S), _p(Pinv), _p(gamma),
                       _p(lam), _p(r), _p(p), cty(tol), ctypes.c_uint32(max_iter), ctypes.byref(it),
                       ctypes.byref(fl)), "solve_blocking")
        return int(it.value), bool(fl.value)

    def solve_host(self, n, N, S, Pinv, gamma, lam, tol=1e-6, max_iter=25):
        """gbdpcg_solve_host_* on numpy arrays (lam in/out). Returns (iters, flag)."""
        import numpy as np
        suf, cty = {np.dtype(np.float32): ("f32", ctypes.c_float),
                    np.dtype(np.float64): ("f64", ctypes.c_double)}[S.dtype]
        it = ctypes.c_uint32(0)
        fl = ctypes.c_uint8(0)

        def hp(a):
            return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        fn = getattr(self.lib, f"gbdpcg_solve_host_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), hp(S), hp(Pinv), hp(gamma),
                       hp(lam), cty(tol), ctypes.c_uint32(max_iter), ctypes.byref(it),
                       ctypes.byref(fl)), "solve_host")
        return int(it.value), bool(fl.value)

    def graph_solve(self, n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters, max_iter_exit):
        """Capture a solve into a hipGraph; returns a Graph whose launch() replays it."""
        suf, args = self.solve_args(n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters,
                                    max_iter_exit)
        g = ctypes.c_void_p()
        fn = getattr(self.lib, f"gbdpcg_graph_create_solve_{suf}")
        self._check(fn(*args, ctypes.byref(g)), "graph_create_solve")
        return Graph(self, g, keep=(S, Pinv, gamma, lam, r, p, iters, max_iter_exit))

    def _form_solve_args(self, n, N, batch, S, Pinv, kind, gamma, lam, r, p, tol, max_iter, iters, mie):
        suf, a = self.solve_args(n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters, mie)
        # (h, n, N, batch, S, Pinv, | kind, | gamma, lam, r, p, tol, max_iter, iters, mie)
        return suf, a[:6] + (ctypes.c_int(kind),) + a[6:]

    def form_pinv_solve(self, n, N, batch, S, Pinv, gamma, lam, kind=PINV_STAIR, r=None, p=None, tol=1e-6,
                        max_iter=25, iters=None, max_iter_exit=None, stream=None):
        """gbdpcg_form_pinv_solve_*: Pinv (output) formed from S, then the solve, one call.  Returns (iters, flags)."""
        import torch
        if iters is None:
            iters = torch.empty(batch, dtype=torch.int32, device=S.device)
        if max_iter_exit is None:
            max_iter_exit = torch.empty(batch, dtype=torch.uint8, device=S.device)
        suf, args = self._form_solve_args(n, N, batch, S, Pinv, kind, gamma, lam, r, p, tol, max_iter, iters,
                                          max_iter_exit)
        fn = getattr(self.lib, f"gbdpcg_form_pinv_solve_{suf}")
        self._check(fn(*args, self._stream(stream)), "form_pinv_solve")
        return iters, max_iter_exit

    def graph_form_pinv_solve(self, n, N, batch, S, Pinv, gamma, lam, r, p, tol, max_iter, iters, max_iter_exit,
                              kind=PINV_STAIR):
        """Capture Pinv formation + solve into one hipGraph (gbdpcg_graph_create_form_pinv_solve_*)."""
        suf, args = self._form_solve_args(n, N, batch, S, Pinv, kind, gamma, lam, r, p, tol, max_iter, iters,
                                          max_iter_exit)
        g = ctypes.c_void_p()
        fn = getattr(self.lib, f"gbdpcg_graph_create_form_pinv_solve_{suf}")
        self._check(fn(*args, ctypes.byref(g)), "graph_create_form_pinv_solve")
        return Graph(self, g, keep=(S, Pinv, gamma, lam, r, p, iters, max_iter_exit))

    def form_pinv(self, n, N, batch, S, kind=PINV_STAIR, Pinv=None, stream=None):
        import torch
        suf, _ = _suffix(S)
        if Pinv is None:
            Pinv = torch.empty_like(S)
        fn = getattr(self.lib, f"gbdpcg_form_pinv_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(n), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(S),
                       _p(Pinv), ctypes.c_int(kind), self._stream(stream)), "form_pinv")
        return Pinv

    def form_schur(self, nx, nu, N, batch, G, C, g, c, S=None, gamma=None, Ginv=None, want_ginv=True, stream=None):
        """gbdpcg_form_schur_*: packed KKT blocks (layouts in include/gbdpcg.h) -> S, gamma and (optionally) G^-1."""
        import torch
        suf, _ = _suffix(G)
        if S is None:
            S = torch.empty(batch * 3 * nx * nx * N, dtype=G.dtype, device=G.device)
        if gamma is None:
            gamma = torch.empty(batch * nx * N, dtype=G.dtype, device=G.device)
        if Ginv is None and want_ginv:
            Ginv = torch.empty_like(G)
        fn = getattr(self.lib, f"gbdpcg_form_schur_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(nx), ctypes.c_uint32(nu), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(G),
                       _p(C), _p(g), _p(c), _p(S), _p(gamma), _p(Ginv), self._stream(stream)), "form_schur")
        return S, gamma, Ginv

    def recover_primal(self, nx, nu, N, batch, Ginv, C, g, lam, z=None, stream=None):
        """gbdpcg_recover_primal_*: z = -G^-1 (g + C' lambda), the layout of g."""
        import torch
        suf, _ = _suffix(Ginv)
        if z is None:
            z = torch.empty_like(g)
        fn = getattr(self.lib, f"gbdpcg_recover_primal_{suf}")
        self._check(fn(self.h, ctypes.c_uint32(nx), ctypes.c_uint32(nu), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(Ginv),
                       _p(C), _p(g), _p(lam), _p(z), self._stream(stream)), "recover_primal")
        return z


    def _kkt_args(self, nx, nu, N, batch, G, C, g, c, S, gamma, Ginv, Pinv, kind, lam, r, p, tol, max_iter, iters, mie, z):
        suf, ft = _suffix(G)
        return suf, (self.h, ctypes.c_uint32(nx), ctypes.c_uint32(nu), ctypes.c_uint32(N), ctypes.c_uint32(batch), _p(G), _p(C),
                     _p(g), _p(c), _p(S), _p(gamma), _p(Ginv), _p(Pinv), ctypes.c_int(kind), _p(lam), _p(r), _p(p), ft(tol),
                     ctypes.c_uint32(max_iter), _p(iters), _p(mie), _p(z))

    def kkt_step(self, nx, nu, N, batch, G, C, g, c, S, gamma, Ginv, Pinv, lam, z, kind=PINV_STAIR, r=None, p=None, tol=1e-6,
                 max_iter=25, iters=None, max_iter_exit=None, stream=None):
        """gbdpcg_kkt_step_*: KKT blocks -> S, gamma, G^-1 -> Pinv -> PCG (warm start from lam) -> primal step z, one call."""
        import torch
        if iters is None:
            iters = torch.zeros(batch, dtype=torch.int32, device=G.device)
        if max_iter_exit is None:
            max_iter_exit = torch.zeros(batch, dtype=torch.uint8, device=G.device)
        suf, args = self._kkt_args(nx, nu, N, batch, G, C, g, c, S, gamma, Ginv, Pinv, kind, lam, r, p, tol, max_iter, iters,
                                   max_iter_exit, z)
        fn = getattr(self.lib, f"gbdpcg_kkt_step_{suf}")
        self._check(fn(*args, self._stream(stream)), "kkt_step")
        return iters, max_iter_exit

    def graph_kkt_step(self, nx, nu, N, batch, G, C, g, c, S, gamma, Ginv, Pinv, lam, r, p, tol, max_iter, iters, max_iter_exit, z,
                       kind=PINV_STAIR):
        """Capture the whole step into one hipGraph (gbdpcg_graph_create_kkt_step_*)."""
        suf, args = self._kkt_args(nx, nu, N, batch, G, C, g, c, S, gamma, Ginv, Pinv, kind, lam, r, p, tol, max_iter, iters,
                                   max_iter_exit, z)
        gr = ctypes.c_void_p()
        fn = getattr(self.lib, f"gbdpcg_graph_create_kkt_step_{suf}")
        self._check(fn(*args, ctypes.byref(gr)), "graph_create_kkt_step")
        return Graph(self, gr, keep=(G, C, g, c, S, gamma, Ginv, Pinv, lam, r, p, iters, max_iter_exit, z))


class Graph:
    def __init__(self, solver, g, keep):
        self.solver, self.g, self.keep = solver, g, keep

    def launch(self, stream=None):
        self.solver._check(self.solver.lib.gbdpcg_graph_launch(self.g, Solver._stream(stream)),
                           "graph_launch")

    def close(self):
        if self.g:
            self.solver.lib.gbdpcg_graph_destroy(self.g)
            self.g = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csr_to_bt(n, N, row_ptr, col_ind, val):
    """gbdpcg_csr_to_bt_* on numpy arrays -> flat [L|D|R] array.  Host-only, no GPU needed."""
    import numpy as np
    lib = load()
    suf = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[val.dtype]
    row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
    col_ind = np.ascontiguousarray(col_ind, np.uint32)
    val = np.ascontiguousarray(val)
    M = np.empty(3 * n * n * N, val.dtype)
    st = getattr(lib, f"gbdpcg_csr_to_bt_{suf}")(
        ctypes.c_uint32(n), ctypes.c_uint32(N), row_ptr.ctypes.data_as(ctypes.c_void_p),
        col_ind.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p),
        M.ctypes.data_as(ctypes.c_void_p))
    if st != OK:
        raise GbdPcgError(f"csr_to_bt: {lib.gbdpcg_status_string(st).decode()}")
    return M
