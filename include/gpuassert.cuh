// gpuassert.cuh -- file-name compatibility only.  Callers of A2R-Lab/GBD-PCG write
// '#include "gpuassert.cuh"'; everything that header declared now lives in gbdpcg.hpp, a host-only
// C++ layer over the C ABI of libgbdpcg.so (no CUDA, no device code in headers).
#pragma once
#include "gbdpcg.hpp"
