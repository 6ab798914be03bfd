// Explicit instantiations of the fp32-faithful (FLAG_SPLIT, dense + conv) and fp8 (FLAG_FP8, dense) kernels: a
// translation unit of its own (see gemm_tiles.h).
#include "gemm_core.h"
#include "gemm_tiles.h"

#define X(BM, BN, WM, WN, NS, PIPE)                                                                    \
  template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_SPLIT>(const GemmParams); \
  template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FLAG_SPLIT>(const GemmParams);
FLUXHIP_TILES_X3(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8>(const GemmParams);
FLUXHIP_TILES_F8(X)
#undef X
// diagnostic: the two ping-pong fp32-faithful conv tiles with phase stamps (fluxhip_gemm_set_trace; tools/conv_phase_trace.py)
template __global__ void gemm_nt_kernel<256, 256, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>(const GemmParams);
template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>(const GemmParams);
// the halo-tile loader of the 3 x 3 fp32-faithful convs whose tile is 256 pixels of one image row (FLAG_DXR, gemm_core.h), + its stamped twin
template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR>(const GemmParams);
template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR | FLAG_TIMED>(const GemmParams);
