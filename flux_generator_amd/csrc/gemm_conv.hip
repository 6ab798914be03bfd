// Explicit instantiations of the implicit-GEMM (conv, AMODE 1) bf16 kernels of every tile configuration: a translation
// unit of its own so that it compiles in parallel with gemm.hip (the table, the launchers and the dense kernels).
#include "gemm_core.h"
#include "gemm_tiles.h"

#define X(BM, BN, WM, WN, NS, PIPE, FL) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FL>(const GemmParams);
FLUXHIP_TILES(X)
#undef X
