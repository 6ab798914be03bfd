// Explicit instantiations of the block-scaled fp8 kernels (FLAG_FP8 | FLAG_LEAN | FLAG_MXA / FLAG_MXC): a translation unit
// of its own (see gemm_tiles.h).
#include "gemm_core.h"
#include "gemm_tiles.h"

#define X(BM, BN, WM, WN, NS, PIPE) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXA>(const GemmParams);
FLUXHIP_TILES_MXA(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXC>(const GemmParams);
FLUXHIP_TILES_MXC(X)
#undef X
