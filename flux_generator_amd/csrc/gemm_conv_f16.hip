// Explicit instantiations of the float16-storage (FLAG_F16) implicit-GEMM (conv, AMODE 1) kernels.
#include "gemm_core.h"
#include "gemm_tiles.h"

#define X(BM, BN, WM, WN, NS, PIPE) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FLAG_F16>(const GemmParams);
FLUXHIP_TILES_F16_CONV(X)
#undef X
