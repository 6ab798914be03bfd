// Explicit instantiations of the float16-storage (FLAG_F16) dense kernels and their GEGLU-pair twins: a translation unit of
// its own so that it compiles in parallel with gemm.hip (which declares them `extern template`).
#include "gemm_core.h"
#include "gemm_tiles.h"

#define X(BM, BN, WM, WN, NS, PIPE) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16>(const GemmParams);
FLUXHIP_TILES_F16_DENSE(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) \
  template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16 | FLAG_LEAN | ((EPI_GEGLU_PAIR + 1) << 8)>(const GemmParams);
FLUXHIP_TILES_F16_PAIR(X)
#undef X
