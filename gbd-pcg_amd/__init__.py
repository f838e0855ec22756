"""gbd-pcg_amd: MI355X-native block-tridiagonal PCG (drop-in for A2R-Lab/GBD-PCG's solve path).

The product is the C-ABI shared library built from csrc/ (include/gbdpcg.h) and the C++
drop-in headers in include/.  The Python here is plumbing for tests and bench.py:
  binding : ctypes view of libgbdpcg.so taking torch device tensors
  synth   : synthetic Schur-system generator
"""
__all__ = ["binding", "synth"]
