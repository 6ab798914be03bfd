// Tile configurations of gemm_nt_kernel as X-macro lists: (BM, BN, WM, WN, NSTAGE, PIPE, FLAGS), each distinct
// instantiation once.  gemm.hip owns the table (kCfgs) and instantiates the dense bf16 kernels; the implicit-GEMM (conv)
// kernels and the FLAG_SPLIT / FLAG_FP8 variants are explicitly instantiated in gemm_conv.hip / gemm_x3f8.hip and
// declared `extern template` in gemm.hip, so the three translation units compile in parallel (one unit took > 5 min).
#pragma once

#define FLUXHIP_TILES(X)                                                                                      \
  X(128, 128, 2, 2, 2, 0, 0) X(128, 64, 2, 2, 2, 0, 0) X(64, 128, 2, 2, 2, 0, 0) X(64, 64, 2, 2, 2, 0, 0)     \
  X(256, 128, 4, 2, 2, 0, 0) X(256, 256, 2, 4, 2, 0, 0) X(128, 128, 2, 2, 2, 1, 0) X(128, 64, 2, 2, 2, 1, 0)  \
  X(64, 128, 2, 2, 2, 1, 0) X(256, 128, 4, 2, 2, 1, 0) X(256, 256, 2, 4, 2, 1, 0) X(256, 128, 4, 2, 3, 1, 0)  \
  X(128, 128, 2, 2, 3, 1, 0) X(128, 256, 2, 4, 2, 1, 0) X(256, 256, 4, 2, 2, 1, 0) X(128, 128, 2, 4, 3, 1, 0) \
  X(128, 128, 2, 2, 4, 1, 0) X(256, 224, 4, 2, 2, 1, 0) X(256, 192, 4, 2, 2, 1, 0) X(128, 128, 4, 2, 3, 1, 0) \
  X(128, 192, 2, 2, 3, 1, 0) X(256, 256, 2, 2, 2, 1, 0) X(256, 160, 4, 2, 3, 1, 0) X(256, 256, 4, 2, 2, 3, 0) \
  X(256, 224, 4, 2, 2, 3, 0) X(256, 192, 4, 2, 2, 3, 0) X(256, 256, 2, 4, 2, 3, 0) X(128, 128, 2, 4, 3, 3, 0) \
  X(256, 128, 4, 2, 3, 3, 0) X(256, 160, 4, 2, 2, 3, 0) X(128, 256, 2, 4, 2, 3, 0) X(128, 128, 2, 2, 2, 3, 0) \
  X(128, 128, 2, 4, 2, 3, 0) X(256, 256, 4, 2, 2, 4, 0) X(256, 224, 4, 2, 2, 4, 0) X(256, 192, 4, 2, 2, 4, 0) \
  X(256, 256, 2, 4, 2, 4, 0) X(128, 128, 2, 4, 3, 4, 0) X(256, 128, 4, 2, 3, 4, 0) X(256, 256, 4, 2, 2, 5, 0) \
  X(256, 224, 4, 2, 2, 5, 0) X(256, 192, 4, 2, 2, 5, 0) X(256, 128, 4, 2, 3, 5, 0) X(128, 128, 2, 4, 3, 5, 0) \
  X(256, 256, 4, 2, 2, 5, 1) X(256, 256, 4, 2, 2, 6, 0) X(256, 224, 4, 2, 2, 6, 0) X(256, 192, 4, 2, 2, 6, 0) \
  X(256, 128, 4, 2, 2, 6, 0) X(128, 128, 2, 4, 2, 6, 0) X(256, 160, 4, 2, 2, 6, 0) X(128, 256, 2, 4, 2, 6, 0) \
  X(256, 192, 4, 2, 2, 6, 1) X(128, 160, 4, 2, 2, 6, 0)

// tiles that also have fp32-faithful (FLAG_SPLIT) kernels, dense and conv
#define FLUXHIP_TILES_X3(X)                                                                                   \
  X(64, 64, 2, 2, 2, 0) X(128, 128, 2, 2, 2, 1) X(128, 64, 2, 2, 2, 1) X(64, 128, 2, 2, 2, 1)                 \
  X(256, 128, 4, 2, 2, 1) X(256, 256, 4, 2, 2, 1) X(128, 128, 2, 4, 3, 5) X(256, 256, 4, 2, 2, 6)             \
  X(128, 256, 2, 4, 2, 6) X(256, 128, 4, 2, 2, 6)

// tiles that also have an fp8 (FLAG_FP8) dense kernel
#define FLUXHIP_TILES_F8(X)                                                                                   \
  X(128, 128, 2, 2, 2, 0) X(128, 64, 2, 2, 2, 0) X(64, 128, 2, 2, 2, 0) X(64, 64, 2, 2, 2, 0)                 \
  X(256, 256, 2, 4, 2, 0) X(256, 128, 4, 2, 2, 6) X(128, 128, 2, 4, 2, 6) X(256, 160, 4, 2, 2, 6)             \
  X(128, 256, 2, 4, 2, 6) X(256, 224, 4, 2, 2, 6) X(256, 192, 4, 2, 2, 6) X(256, 256, 4, 2, 2, 6)

// block-scaled fp8 (gemm_mx.hip): FLAG_MXA = activation operand with E8M0 block scales (the tiles the N = 3072 projections
// run on), FLAG_MXC = e4m3 + block-scale output (32-column blocks must sit inside a wave's sub-tile: BN / WN % 32 == 0)
#define FLUXHIP_TILES_MXA(X) X(256, 256, 4, 2, 2, 6) X(256, 224, 4, 2, 2, 6) X(256, 192, 4, 2, 2, 6) X(256, 128, 4, 2, 2, 6) \
  X(128, 256, 2, 4, 2, 6) X(128, 128, 2, 4, 2, 6)
#define FLUXHIP_TILES_MXC(X) X(256, 256, 4, 2, 2, 6) X(256, 192, 4, 2, 2, 6) X(256, 128, 4, 2, 2, 6) \
  X(128, 256, 2, 4, 2, 6) X(128, 128, 2, 4, 2, 6)

// float16-storage (FLAG_F16) kernels: the tiles the dense / conv pickers can return (kCands, kConvCands in gemm.hip), the
// plain-ring 256 x 256 conv tile of the fused-upsample loader, and the two tiles that carry the GEGLU pair epilogue.
// (BM, BN, WM, WN, NSTAGE, PIPE); instantiated in gemm_f16.hip (dense, pair) and gemm_conv_f16.hip (conv).
#define FLUXHIP_TILES_F16_DENSE(X)                                                                            \
  X(256, 256, 4, 2, 2, 6) X(256, 224, 4, 2, 2, 6) X(256, 192, 4, 2, 2, 6) X(256, 160, 4, 2, 2, 6)             \
  X(256, 128, 4, 2, 3, 5) X(128, 256, 2, 4, 2, 6) X(128, 128, 2, 4, 3, 5) X(128, 128, 2, 2, 2, 1)             \
  X(128, 64, 2, 2, 2, 1) X(64, 128, 2, 2, 2, 1) X(64, 64, 2, 2, 2, 0) X(128, 160, 4, 2, 2, 6)
#define FLUXHIP_TILES_F16_CONV(X)                                                                             \
  X(256, 256, 4, 2, 2, 6) X(256, 224, 4, 2, 2, 6) X(256, 192, 4, 2, 2, 6) X(256, 160, 4, 2, 2, 6)             \
  X(256, 128, 4, 2, 2, 1) X(128, 256, 2, 4, 2, 6) X(128, 128, 2, 2, 2, 1) X(128, 64, 2, 2, 2, 1)              \
  X(64, 128, 2, 2, 2, 1) X(64, 64, 2, 2, 2, 0) X(256, 256, 4, 2, 2, 1)
#define FLUXHIP_TILES_F16_PAIR(X) X(256, 256, 4, 2, 2, 6) X(128, 256, 2, 4, 2, 6)
