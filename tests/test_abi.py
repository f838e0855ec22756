"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/gbdpcg.h
declares, its host-only helpers work, and without a GPU it refuses to run (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from gbd_pcg_amd import binding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    binding.build()
    return binding.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "gbdpcg.h")).read()
    declared = set(re.findall(r"\b(gbdpcg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(binding.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_status_strings(lib):
    assert b"gfx950" in lib.gbdpcg_version()
    assert lib.gbdpcg_status_string(0) == b"ok"
    assert lib.gbdpcg_status_string(12) == b"not implemented"


@pytest.mark.parametrize("es,n,N,want", [(8, 2, 3, 400), (4, 14, 64, 7056), (4, 14, 128, 7056),
                                         (8, 36, 256, 93312)])
def test_pcg_shared_mem_size_formula(lib, es, n, N, want):
    """pcgSharedMemSize<T> (pcg.cuh:13-20); expected values from SURVEY.md section 8a7."""
    assert lib.gbdpcg_pcg_shared_mem_size(ctypes.c_uint32(es), ctypes.c_uint32(n), ctypes.c_uint32(N)) == want


def test_no_gpu_means_no_solver(lib):
    """Without a gfx950 device gbdpcg_create fails; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.gbdpcg_create(ctypes.byref(h), 0) == 3  # GBDPCG_ERR_NO_DEVICE
    assert not h
    with pytest.raises(binding.GbdPcgError):
        binding.Solver(0)


def test_null_handle_is_invalid(lib):
    assert lib.gbdpcg_solve_f32(None, 14, 8, 1, None, None, None, None, None, None,
                                ctypes.c_float(0), 1, None, None, None) == 1
    assert lib.gbdpcg_spmv_f64(None, 14, 8, 1, None, None, None, None) == 1
    assert lib.gbdpcg_destroy(None) == 1


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csr_to_bt_roundtrip(orc, dtype):
    """f3: CSR (types.cuh:7-15) -> [L|D|R]; the stub at interface.cuh:8-20 made real."""
    import scipy.sparse as sp
    n, N = 5, 6
    d = synth.gen_numpy(n, N, seed=4, dtype=dtype)
    A = orc.dense_from_bt(n, N, d["S"][0]).astype(dtype)
    csr = sp.csr_matrix(A)
    M = binding.csr_to_bt(n, N, csr.indptr, csr.indices, csr.data.astype(dtype))
    assert np.array_equal(orc.dense_from_bt(n, N, M).astype(dtype), A)
    # an entry outside the block-tridiagonal pattern is rejected
    A[0, 3 * n] = 1.0
    csr = sp.csr_matrix(A)
    with pytest.raises(binding.GbdPcgError):
        binding.csr_to_bt(n, N, csr.indptr, csr.indices, csr.data.astype(dtype))


def test_shipped_library_has_no_fault_injection_hooks():
    """The hooks that make a workgroup of the cluster / persistent kernels vanish or arrive late, shorten the spin bound or
    switch the in-kernel rescue off exist in csrc/variants/libgbdpcg_hooks.so only (-DGBDPCG_TEST_HOOKS):
    the shipped library must not even contain their names."""
    from gbd_pcg_amd import binding
    blob = open(binding.LIB_PATH, "rb").read()
    for name in (b"DROP_WG", b"SPIN_LIMIT", b"RESCUE_OFF", b"HOLD_US"):
        assert name not in blob, name
    hooks = os.path.join(os.path.dirname(binding.LIB_PATH), "variants", "libgbdpcg_hooks.so")
    assert os.path.exists(hooks), "make -C gbd-pcg_amd/csrc builds it next to the shipped library"
    hb = open(hooks, "rb").read()
    assert b"GBDPCG_CLUSTER_DROP_WG" in hb and b"GBDPCG_PERSIST_HOLD_US" in hb and b"GBDPCG_RESCUE_OFF" in hb
