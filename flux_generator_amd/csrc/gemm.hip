// Host launchers for the bf16 NT GEMM / implicit-GEMM conv (kernel: gemm_core.h).
#include "../../include/fluxhip.h"
#include <cmath>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "gemm_core.h"
#include "gemm_tiles.h"

// instantiated in gemm_conv.hip / gemm_x3f8.hip
#define X(BM, BN, WM, WN, NS, PIPE, FL) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FL>(const GemmParams);
FLUXHIP_TILES(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE)                                                                           \
  extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_SPLIT>(const GemmParams); \
  extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FLAG_SPLIT>(const GemmParams);
FLUXHIP_TILES_X3(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8>(const GemmParams);
FLUXHIP_TILES_F8(X)
#undef X

extern template __global__ void gemm_nt_kernel<256, 256, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>(const GemmParams);
extern template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>(const GemmParams);
extern template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR>(const GemmParams);
extern template __global__ void gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR | FLAG_TIMED>(const GemmParams);

// instantiated in gemm_mx.hip
#define X(BM, BN, WM, WN, NS, PIPE) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXA>(const GemmParams);
FLUXHIP_TILES_MXA(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXC>(const GemmParams);
FLUXHIP_TILES_MXC(X)
#undef X

// instantiated in gemm_f16.hip / gemm_conv_f16.hip
#define X(BM, BN, WM, WN, NS, PIPE) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16>(const GemmParams);
FLUXHIP_TILES_F16_DENSE(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FLAG_F16>(const GemmParams);
FLUXHIP_TILES_F16_CONV(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) \
  extern template __global__ void gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16 | FLAG_LEAN | ((EPI_GEGLU_PAIR + 1) << 8)>(const GemmParams);
FLUXHIP_TILES_F16_PAIR(X)
#undef X

namespace {

struct TileCfg {
  int bm, bn, threads, lds;
  void (*dense)(const GemmParams);
  void (*conv)(const GemmParams);
  void (*dense_x3)(const GemmParams);   // FLAG_SPLIT (bf16x3, fp32-faithful) instantiations; null for most tiles
  void (*conv_x3)(const GemmParams);
  void (*dense_f8)(const GemmParams);   // FLAG_FP8 (e4m3 x e4m3 on the 16x16x128 block-scaled MFMA); simple-ring and ping-pong tiles
  int wm = 1;                           // waves along M (rows per wave = bm / wm)
  void (*dense_rs)(const GemmParams) = nullptr;   // FLAG_RS | FLAG_LEAN (split-K reduce-scatter hand-off) variant: 256 x 256 / 256 x 192 ping-pong
  void (*dense_lean)(const GemmParams) = nullptr; // FLAG_LEAN: the same dense kernel with only the transformer-block epilogues compiled in (lean_ok)
  void (*dense_f8_lean)(const GemmParams) = nullptr;   // FLAG_FP8 | FLAG_LEAN: the four transformer-block epilogues (runtime switch)
  void (*dense_f8_mxa)(const GemmParams) = nullptr;    // ... with a block-scaled activation operand (attached by shape below)
  void (*dense_f8_mxc)(const GemmParams) = nullptr;    // ... with an e4m3 + block-scale output
  void (*dense_pair)(const GemmParams) = nullptr;      // FLAG_LEAN with EPI_GEGLU_PAIR compiled in (the UNet's fused GEGLU Linears)
  void (*dense_f16)(const GemmParams) = nullptr;       // FLAG_F16 (float16 storage) twins: the tiles of kCands / kConvCands
  void (*conv_f16)(const GemmParams) = nullptr;
  void (*dense_pair_f16)(const GemmParams) = nullptr;
  int lean_epi = -1;                              // >= 0: the lean kernel has exactly this epilogue compiled in
  int rs_epi = -1;                                //       (and the reduce-scatter kernel this one)
};

template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE, int FLAGS = 0>
constexpr TileCfg make_cfg() {
  return TileCfg{BM, BN, WM * WN * 64, (NSTAGE * BM + (PIPE >= 3 ? NSTAGE + 1 : NSTAGE) * BN) * 64 * 2,
                 gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAGS>,
                 gemm_nt_kernel<BM, BN, WM, WN, 1, NSTAGE, PIPE, FLAGS>, nullptr, nullptr, nullptr, WM};
}
// the same tile with the fp32-faithful (FLAG_SPLIT) kernels as well: only the tiles the VAE decoders use
template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE>
constexpr TileCfg make_cfg_x3() {
  TileCfg c = make_cfg<BM, BN, WM, WN, NSTAGE, PIPE>();
  c.dense_x3 = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_SPLIT>;
  c.conv_x3 = gemm_nt_kernel<BM, BN, WM, WN, 1, NSTAGE, PIPE, FLAG_SPLIT>;
  return c;
}
template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE>
constexpr TileCfg make_cfg_f8() {
  TileCfg c = make_cfg<BM, BN, WM, WN, NSTAGE, PIPE>();
  c.dense_f8 = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_FP8>;
  return c;
}
template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE>
constexpr TileCfg make_cfg_x3_f8() {
  TileCfg c = make_cfg_x3<BM, BN, WM, WN, NSTAGE, PIPE>();
  c.dense_f8 = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_FP8>;
  return c;
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE, int EXTRA = 0>
constexpr TileCfg with_rs(TileCfg c) {      // split-K launches are the K >= 5120 projections back into the residual stream
  c.dense_rs = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_RS | FLAG_LEAN | ((EPI_GATE_RES + 1) << 8) | EXTRA>;
  c.rs_epi = EPI_GATE_RES;
  return c;
}
// Lean twins of the tiles the Flux / transformer-block launches run on.  The epilogue of gemm_nt_kernel is one runtime switch
// over every fused form any caller needs (9 activations / gates, row or column bias, addvec, float32 output, the direct-store
// path for unaligned operands, the split-K chain); code that is merely PRESENT costs the launches that never take it - the
// reduce-scatter hand-off compiled into the shared kernel cost the non-split launches 2-4 %.  A FLAG_LEAN
// instantiation keeps bias + {none, GELU-tanh, gate-residual, split-GELU} on the LDS-transposed epilogue and nothing else.
// (A second twin with the four epilogues behind the runtime switch, for the launches of the multi-round plans - Flux-dev
//  1024^2, batch 4 - the single-epilogue kernels do not match, measured level with the generic kernel there and is not built.)
template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE, int EPI>
constexpr TileCfg with_lean(TileCfg c) {
  c.dense_lean = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_LEAN | ((EPI + 1) << 8)>;
  c.lean_epi = EPI;
  return c;
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE>
constexpr TileCfg with_pair(TileCfg c) {      // EPI_GEGLU_PAIR exists in these instantiations only (no generic kernel carries it)
  c.dense_pair = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_LEAN | ((EPI_GEGLU_PAIR + 1) << 8)>;
  return c;
}
template <int BM, int BN, int WM, int WN, int NSTAGE, int PIPE>
constexpr TileCfg with_lean_f8(TileCfg c) {   // fp8: at C5's batch the 256x224 / 256x256 tiles serve several epilogues each
  c.dense_f8_lean = gemm_nt_kernel<BM, BN, WM, WN, 0, NSTAGE, PIPE, FLAG_FP8 | FLAG_LEAN>;
  return c;
}

// index 0 is unused ("auto")
TileCfg kCfgs[] = {
    TileCfg{0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
    make_cfg_f8<128, 128, 2, 2, 2, 0>(),  // 1: 64 KiB LDS, 2 blocks/CU
    make_cfg_f8<128, 64, 2, 2, 2, 0>(),   // 2: more blocks for N=3072 outputs at small M
    make_cfg_f8<64, 128, 2, 2, 2, 0>(),   // 3
    make_cfg_x3_f8<64, 64, 2, 2, 2, 0>(),    // 4: tiny problems (N=64 final layer)
    make_cfg<256, 128, 4, 2, 2, 0>(),  // 5: 8 waves, 96 KiB
    make_cfg_f8<256, 256, 2, 4, 2, 0>(),  // 6: 8 waves x (128x64), 128 KiB (fp8: the one 256x256 tile that fits 256 registers)
    make_cfg_x3<128, 128, 2, 2, 2, 1>(),  // 7: fragment-pipelined variants of 1,2,3,5,6
    make_cfg_x3<128, 64, 2, 2, 2, 1>(),   // 8
    make_cfg_x3<64, 128, 2, 2, 2, 1>(),   // 9
    make_cfg_x3<256, 128, 4, 2, 2, 1>(),  // 10
    make_cfg<256, 256, 2, 4, 2, 1>(),  // 11
    make_cfg<256, 128, 4, 2, 3, 1>(),  // 12: 3-deep ring, 144 KiB
    make_cfg<128, 128, 2, 2, 3, 1>(),  // 13
    make_cfg<128, 256, 2, 4, 2, 1>(),  // 14
    make_cfg_x3<256, 256, 4, 2, 2, 1>(),  // 15: 8 waves x (64x128)
    make_cfg<128, 128, 2, 4, 3, 1>(),  // 16: 8 waves x (64x32), 3-deep ring
    make_cfg<128, 128, 2, 2, 4, 1>(),  // 17: 4-deep ring, 128 KiB
    make_cfg<256, 224, 4, 2, 2, 1>(),  // 18: 21504 = 96 x 224 -> 480 tiles at M = 1280
    make_cfg<256, 192, 4, 2, 2, 1>(),  // 19: 9216 = 48 x 192 -> 240 tiles at M = 1280
    make_cfg<128, 128, 4, 2, 3, 1>(),  // 20: 8 waves x (32x64), 3-deep ring
    make_cfg<128, 192, 2, 2, 3, 1>(),  // 21: 4 waves x (64x96), 120 KiB
    make_cfg<256, 256, 2, 2, 2, 1>(),  // 22: 4 waves x (128x128): one wave per SIMD, 512-register waves
    make_cfg<256, 160, 4, 2, 3, 1>(),  // 23: 3-deep ring, 156 KiB
    make_cfg<256, 256, 4, 2, 2, 3>(),  // 24: cfg 15 with a 3-deep WEIGHT ring (2 x 32 + 3 x 32 KiB = 160 KiB)
    make_cfg<256, 224, 4, 2, 2, 3>(),  // 25: cfg 18 "  (2 x 32 + 3 x 28 KiB)
    make_cfg<256, 192, 4, 2, 2, 3>(),  // 26: cfg 19 "
    make_cfg<256, 256, 2, 4, 2, 3>(),  // 27: cfg 11 "
    make_cfg<128, 128, 2, 4, 3, 3>(),  // 28: cfg 16 with a 4-deep weight ring
    make_cfg<256, 128, 4, 2, 3, 3>(),  // 29: cfg 12 "
    make_cfg<256, 160, 4, 2, 2, 3>(),  // 30: 256x160, 2 + 3
    make_cfg<128, 256, 2, 4, 2, 3>(),  // 31: cfg 14, 2 + 3
    make_cfg<128, 128, 2, 2, 2, 3>(),  // 32: cfg 7, 2 + 3 (80 KiB: 2 blocks/CU)
    make_cfg<128, 128, 2, 4, 2, 3>(),  // 33: 128x128, 8 waves, 2 + 3
    make_cfg<256, 256, 4, 2, 2, 1>(),     // 34: (= 15; slot kept so later indices stay stable)
    make_cfg<256, 256, 4, 2, 2, 3>(),     // 35: (= 24)
    make_cfg<256, 256, 4, 2, 2, 4>(),     // 36: cfg 24 with the LDS-DMA pieces spread between the MFMAs
    make_cfg<256, 224, 4, 2, 2, 4>(),     // 37: cfg 25 "
    make_cfg<256, 192, 4, 2, 2, 4>(),     // 38: cfg 26 "
    make_cfg<256, 256, 2, 4, 2, 4>(),     // 39: cfg 27 "
    make_cfg<128, 128, 2, 4, 3, 4>(),     // 40: cfg 28 "
    make_cfg<256, 128, 4, 2, 3, 4>(),     // 41: cfg 29 "
    make_cfg<256, 256, 4, 2, 2, 4>(),     // 42: (= 36)
    make_cfg<256, 256, 4, 2, 2, 5>(),     // 43: cfg 36 with the fragment reads spread between the MFMAs too
    make_cfg<256, 224, 4, 2, 2, 5>(),     // 44: cfg 37 "
    make_cfg<256, 192, 4, 2, 2, 5>(),     // 45: cfg 38 "
    make_cfg<256, 128, 4, 2, 3, 5>(),     // 46: cfg 41 "
    with_lean<128, 128, 2, 4, 3, 5, EPI_GATE_RES>(make_cfg_x3<128, 128, 2, 4, 3, 5>()),     // 47: cfg 40 "
    make_cfg<256, 256, 4, 2, 2, 5, 1>(),  // 48: cfg 43 with phase stamps (diagnostic: fluxhip_gemm_set_trace, tools/gemm_phase_trace.py)
    with_pair<256, 256, 4, 2, 2, 6>(with_lean_f8<256, 256, 4, 2, 2, 6>(with_lean<256, 256, 4, 2, 2, 6, EPI_GELU_TANH>(with_rs<256, 256, 4, 2, 2, 6>(make_cfg_x3_f8<256, 256, 4, 2, 2, 6>())))),     // 49: cfg 43 with the ping-pong schedule (one MFMA-issuing wave per SIMD per phase)
    with_lean_f8<256, 224, 4, 2, 2, 6>(with_lean<256, 224, 4, 2, 2, 6, EPI_SPLIT_GELU>(make_cfg_f8<256, 224, 4, 2, 2, 6>())),     // 50: cfg 44 "
    with_lean<256, 192, 4, 2, 2, 6, EPI_BIAS>(with_rs<256, 192, 4, 2, 2, 6>(make_cfg_f8<256, 192, 4, 2, 2, 6>())),     // 51: cfg 45 "
    make_cfg_x3_f8<256, 128, 4, 2, 2, 6>(),     // 52: 256x128, ping-pong, 2 + 3 ring (112 KiB)
    make_cfg_f8<128, 128, 2, 4, 2, 6>(),     // 53: 128x128, ping-pong (80 KiB)
    make_cfg_f8<256, 160, 4, 2, 2, 6>(),     // 54: 256x160, ping-pong
    with_pair<128, 256, 2, 4, 2, 6>(make_cfg_x3_f8<128, 256, 2, 4, 2, 6>()),     // 55: 128x256, ping-pong
    with_rs<256, 192, 4, 2, 2, 6, 1>(make_cfg<256, 192, 4, 2, 2, 6, 1>()),        // 56: cfg 51 with stamps around the main loop, the split-K reduce-scatter segments and the epilogue (diagnostic)
    // 57 (round 6): 128x160, ping-pong, 8 waves x (32x80).  1280 = 8 x 160: the M = 4096, N = 1280 projections of the SDXL transformer
    // blocks (313 launches per UNet step) are 32 x 8 = 256 tiles - one full round - where 128x256 gives 160 tiles on 256 CUs
    with_lean<128, 160, 4, 2, 2, 6, EPI_GATE_RES>(make_cfg<128, 160, 4, 2, 2, 6>()),
    // (the same tile with a 3 + 4 ring on the spread schedule and with a 4-deep plain ring were built and swept with it: level
    //  (620 / 709 / 869 TFLOP/s on the three shapes below) and 8 % slower - the K-step is not waiting for its loads - removed)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// block-scaled fp8 twins, attached by tile shape (ping-pong tiles only)
const bool g_mx_attached = [] {
  auto attach = [](int bm, int bn, void (*k)(const GemmParams), bool mxa) {
    for (int i = 49; i <= 55; ++i) {          // the ping-pong tiles
      TileCfg& c = kCfgs[i];
      if (c.bm == bm && c.bn == bn && c.dense_f8) (mxa ? c.dense_f8_mxa : c.dense_f8_mxc) = k;
    }
  };
#define X(BM, BN, WM, WN, NS, PIPE) attach(BM, BN, gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXA>, true);
  FLUXHIP_TILES_MXA(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) attach(BM, BN, gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_FP8 | FLAG_LEAN | FLAG_MXC>, false);
  FLUXHIP_TILES_MXC(X)
#undef X
  return true;
}();

// float16-storage twins, attached by tile shape: a table entry gets the FLAG_F16 kernel of its own (BM, BN, waves, ring, PIPE)
const bool g_f16_attached = [] {
  for (int i = 1; i < kNumCfgs; ++i) {
    TileCfg& c = kCfgs[i];
#define X(BM, BN, WM, WN, NS, PIPE) \
    if (c.dense == gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, 0>) c.dense_f16 = gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16>;
    FLUXHIP_TILES_F16_DENSE(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE) \
    if (c.dense == gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, 0>) c.conv_f16 = gemm_nt_kernel<BM, BN, WM, WN, 1, NS, PIPE, FLAG_F16>;
    FLUXHIP_TILES_F16_CONV(X)
#undef X
#define X(BM, BN, WM, WN, NS, PIPE)                                 \
    if (c.dense == gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, 0> && c.dense_pair) \
      c.dense_pair_f16 = gemm_nt_kernel<BM, BN, WM, WN, 0, NS, PIPE, FLAG_F16 | FLAG_LEAN | ((EPI_GEGLU_PAIR + 1) << 8)>;
    FLUXHIP_TILES_F16_PAIR(X)
#undef X
  }
  return true;
}();

bool g_attr_set[kNumCfgs][14] = {};
unsigned long long* g_trace = nullptr;   // fluxhip_gemm_set_trace
// fluxhip_conv_set_x3_tile: forced tile of the fp32-faithful convs (0: the picker; env FLUXHIP_CONV_X3_CFG) and the halo-tile loader
// (FLAG_DXR; env FLUXHIP_CONV_DXR=0 switches it off) - A/B timing and tests
// (atomics: ctypes callers may launch from several host threads - the GIL is released during a call)
std::atomic<int> g_conv_x3_cfg{[] { const char* e = getenv("FLUXHIP_CONV_X3_CFG"); return e ? atoi(e) : 0; }()};
std::atomic<bool> g_conv_dxr{[] { const char* e = getenv("FLUXHIP_CONV_DXR"); return !(e && e[0] == '0'); }()};
std::atomic<long long> g_conv_dxr_launches{0};
std::once_flag g_dxr_attr_once[2];        // LDS-size attribute of the two halo-tile kernels: set once, like g_attr_set for the table

// Tile choice: a time model per tile, fitted to tools/gemm_tune.py sweeps.
//   time = t_launch + R(x) * g(x) * (K/64 * t_step + t_fixed),   x = tiles / (256 CUs x blocks/CU)
// t_step = one K-step of the main loop with every CU busy, t_fixed = prologue fill + epilogue of one tile round
// (tools/fit_tiles.py).  Round 3 refit of the dense bf16 table on 111 shapes (Flux at 512^2 / 1024^2 / batch 4, T5, CLIP and
// the SD / SDXL transformer GEMMs at batch 1..16), each timed from a replayed hipGraph (profiles/r03_gemm_tune_*.txt):
//   g(x) = phi + (1 - phi) min(1, x): a round that leaves CUs idle runs faster (clocks, L2 / fabric contention) - phi = 0.6-0.7
//          for every one-block-per-CU tile, which the round-1/2 model (g = 1) did not have: it over-priced the large tiles on
//          the under-filled SDXL launches (M = 4096, N = 1280: picked 128x128 x 2 blocks/CU at 35 us, 128x256 takes 25.7 us);
//   R(x) = 1 for x <= 1, else (1 - beta) x + beta ceil(x): the last partial round costs less than a whole one.
// Mean regret (time of the pick / time of the best candidate) over the 111 shapes 1.069 -> 1.005, worst 1.45 -> 1.15; the
// in-situ choices of the Flux plans (256x192 split-K 3 for mlp2 / linear2, 256x224 for linear1, ...) are unchanged.
// The conv / fp32-faithful / fp8 tables below keep the round-1/2 form (phi = 1, beta = 1, t_launch = 0).
struct Cand { int cfg; int bpc; float t_step_us; float t_fixed_us; float phi = 1.0f; };
struct CostForm { float t_launch_us, beta; };
constexpr CostForm kPlainForm = {0.0f, 1.0f}, kDenseForm = {1.79f, 0.46f};
const Cand kCands[] = {
    {49, 1, 1.590f, 9.17f, 0.56f},    // 256x256, ping-pong schedule, 2 + 3 ring
    {50, 1, 1.424f, 10.63f, 0.67f},   // 256x224  "
    {51, 1, 1.197f, 9.51f, 0.64f},    // 256x192  "
    {54, 1, 1.223f, 8.38f, 0.62f},    // 256x160  "
    {46, 1, 1.023f, 7.18f, 0.62f},    // 256x128, spread LDS-DMA + fragment reads, 3 + 4 ring
    {55, 1, 0.990f, 5.06f, 0.68f},    // 128x256, ping-pong
    {57, 1, 0.720f, 5.50f, 0.68f},    // 128x160, ping-pong (round 6; float16 sweep at M = 4096, profiles/r06_gemm_tune_sdxl_f16.txt: N = K = 1280
                                      //  21.7 us = 619 TFLOP/s where 128x256 takes 23.6; K = 5120 62.0 us = 866 against 75.0)
    {47, 1, 0.606f, 3.69f, 0.69f},    // 128x128, 8 waves, spread reads
    // two blocks per CU (per-step time with both blocks resident)
    {7, 2, 1.244f, 5.95f, 0.63f},     // 128x128, 4 waves
    {8, 2, 0.938f, 2.34f, 0.58f},     // 128x64
    {9, 2, 0.737f, 2.34f, 0.78f},     // 64x128
    {4, 2, 0.546f, 1.06f, 0.92f},     // 64x64
};

// The implicit-GEMM (conv) loader has its own table: its K-step carries the tap / border address generation, so the
// plain 2-deep rings win there and the spread-DMA variants lose.  Round 3: refitted in the dense table's form on the SD / SDXL
// UNet conv shapes at batch 1, 2, 4 and 16 (tools/conv_tune_unet.py, profiles/r03_conv_tune_unet_b*.txt), and the 256x160 /
// 256x192 / 256x224 ping-pong tiles joined the candidates: 320 = 2 x 160 and 640 = 4 x 160 output channels fill them exactly
// (64x64x320 -> 320 at batch 16: 204 -> 159 us; 32x32x640 -> 640: 150 -> 127 us).
constexpr CostForm kConvForm = {4.39f, 0.22f};
const Cand kConvCands[] = {
    {49, 1, 1.657f, 7.58f, 0.75f},    // 256x256, ping-pong (the fused-upsample loader stays on the plain ring, cfg 15, see fluxhip_conv2d_*)
    {50, 1, 1.516f, 8.58f, 0.78f},    // 256x224  "
    {51, 1, 1.433f, 7.69f, 0.77f},    // 256x192  "
    {54, 1, 1.352f, 6.92f, 0.77f},    // 256x160  "
    {10, 1, 1.161f, 5.82f, 0.72f},    // 256x128
    {55, 1, 1.114f, 2.80f, 0.75f},    // 128x256, ping-pong (the address generation runs in the memory phase, off the MFMA wave)
    {7, 2, 1.370f, 5.04f, 0.59f},     // 128x128, 2 blocks/CU
    {8, 2, 0.767f, 5.47f, 0.75f},     // 128x64
    {9, 2, 0.797f, 1.93f, 0.76f},     // 64x128
    {4, 2, 0.484f, 1.78f, 1.00f},     // 64x64
};

// fp32-faithful (FLAG_SPLIT) kernels exist for these tiles only; same time model with 3 x the K-steps.
const Cand kX3Cands[] = {
    {49, 1, 1.072f, 22.1f}, {55, 1, 0.787f, 10.9f}, {47, 1, 0.564f, 5.71f}, {7, 2, 0.903f, 10.2f},
    {8, 2, 0.847f, 0.30f},  {9, 2, 0.672f, 2.90f},  {4, 2, 0.483f, 1.15f},
};
// (round 3, tools/conv_tune_x3.py, profiles/r03_conv_tune_x3_*.txt: 128x256 nudged 0.98 -> 1.05 so that the 128^2 x 512 -> 512
//  layers (M = 16384: 128 tiles of 256x256) take 256x256 split-K 2, 236 us, instead of 128x256 unsplit, 284 us.  The 3-pass loop
//  really costs 1.28 us per K-step on the 256x128 / 128x256 tiles, but entering that value sends the 64^2 and the 128-channel
//  layers to worse tiles - the other rows of this table are as mis-scaled - so until the whole table is refitted in the dense
//  table's form only the one ranking that was wrong is corrected.)
const Cand kX3ConvCands[] = {
    {49, 1, 1.850f, 6.0f}, {52, 1, 1.100f, 4.8f}, {10, 1, 1.110f, 4.8f}, {55, 1, 1.050f, 8.2f}, {7, 2, 1.155f, 4.0f},
    {8, 2, 0.847f, 0.30f}, {9, 2, 0.672f, 2.90f}, {4, 2, 0.483f, 1.15f},
};   // (52 = the 256x128 tile on the ping-pong schedule, round 4: takes exactly the layers the plain-ring 256x128 tile had - the
     //  128-output-channel convs at 512^2: 518 vs 544 us and 286 vs 290 us, tools/conv_tune_x3.py - and nothing else)

// fp8 kernels: the ping-pong tiles (time per K-step as measured for bf16: a K-step is the same 128 bytes per row) and
// the simple-ring tiles for small shapes.
// (the 256-wide ping-pong tiles fit 256 registers since the fp8 loop reads and consumes a step's fragments inside one
//  iteration - the rotated group-0 loop in gemm_core.h - and the staging sources are 32-bit offsets from a scalar base;
//  the 256x256 one keeps 76 bytes of scratch in its epilogue only)
const Cand kF8Cands[] = {          // (t_step per 128-byte K-step, t_fixed) fitted to TUNE_FP8=1 tools/gemm_tune.py, M = 17408 / 1280
    {49, 1, 1.398f, 22.8f},  // 256x256, ping-pong: 1.9-2.2 PFLOP/s on the N >= 9216 shapes at M = 17408
    {50, 1, 1.460f, 13.2f},  // 256x224, ping-pong: 1.7-2.6 PFLOP/s
    {51, 1, 1.360f, 11.2f},  // 256x192, ping-pong (fit 1.30; nudged so that 256x224 keeps the shapes where both need the same rounds)
    {6, 1, 1.571f, 20.6f},   // 256x256, simple ring: 1.7-2.1 PFLOP/s on the large shapes
    {55, 1, 0.927f, 10.1f},  // 128x256, ping-pong
    {54, 1, 1.178f, 13.1f},  // 256x160  "
    {52, 1, 1.137f, 11.1f},  // 256x128  "
    {53, 1, 0.815f, 2.8f},   // 128x128  "  (the N = 3072 shapes at batch 1: 1.0-1.5 PFLOP/s)
    // two blocks per CU: the per-step times are for BOTH blocks resident (each runs at about half the single-block rate)
    {1, 2, 1.70f, 9.2f},     // 128x128, simple ring
    {2, 2, 1.70f, 0.30f},  {3, 2, 1.35f, 2.90f},  {4, 2, 0.97f, 1.15f},
};

// block-scaled fp8 (FLAG_MXA / FLAG_MXC): the ping-pong tiles that carry those kernels, unsplit (rows of kF8Cands)
const Cand kF8MxaCands[] = {{49, 1, 1.398f, 22.8f}, {50, 1, 1.460f, 13.2f}, {51, 1, 1.360f, 11.2f}, {52, 1, 1.137f, 11.1f},
                             {55, 1, 0.927f, 10.1f}, {53, 1, 0.815f, 2.8f}};
const Cand kF8MxcCands[] = {{49, 1, 1.398f, 22.8f}, {51, 1, 1.360f, 11.2f}, {52, 1, 1.137f, 11.1f},
                             {55, 1, 0.927f, 10.1f}, {53, 1, 0.815f, 2.8f}};

// Split-K workspace (fluxhip_set_workspace): [kSkMaxTiles] int32 hand-off counters (reduce-scatter mode: arrival counters
// in the first half, departure counters in the second), then fp32 partial tiles.
constexpr int kSkMaxTiles = 16384;
constexpr long long kSkFlagBytes = (long long)kSkMaxTiles * 4;
char* g_ws = nullptr;
long long g_ws_bytes = 0;
constexpr float kHopUs = 20.0f, kHopNextUs = 8.0f;   // measured cost of the first / each further hand-off of a chain
constexpr float kRsHopUs = 9.0f, kRsHopNextUs = 4.0f; // reduce-scatter hand-off (all S exchanges concurrent)
int g_num_cus = 0;                                    // CUs of the bound device (reduce-scatter needs the whole grid resident)
// FLUXHIP_SPLITK=chain (or fluxhip_gemm_set_splitk_mode(1)) keeps every split-K launch on the chain (A/B runs, diagnostics)
bool g_rs_enabled = [] { const char* e = getenv("FLUXHIP_SPLITK"); return !(e && e[0] == 'c'); }();
bool g_rs_any_grid = false;      // fluxhip_gemm_set_splitk_mode(2), tests: reduce-scatter also for grids larger than the chip
long long g_rs_launches = 0;
// how long a block of a reduce-scatter tile polls for its peers before it orphans its slice and exits (gemm_core.h, hand-off):
// 100 MHz ticks.  100 us is 10-20x the skew between the blocks of a resident grid; a false timeout only costs time (the
// wait-free completion gives the same bits).  FLUXHIP_RS_TIMEOUT_US / fluxhip_gemm_set_rs_timeout_us override (tests use 0
// to force every early block through the orphan path).
int g_rs_timeout_ticks = [] { const char* e = getenv("FLUXHIP_RS_TIMEOUT_US"); return (e && *e) ? atoi(e) * 100 : 10000; }();
// The split-K workspace (counters + partial slabs) belongs to ONE launch at a time.  Launches on one stream are ordered by
// the stream; when the stream changes, the new one is made to wait for everything the previous one has enqueued so far.
hipStream_t g_ws_stream = nullptr;
bool g_ws_stream_set = false;
hipEvent_t g_ws_event = nullptr;

std::mutex g_ws_mutex;      // the remembered stream / event are process-global: two host threads launching split-K GEMMs
                            // (ctypes releases the GIL during a call) must not interleave inside this function
int serialize_workspace_user(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_ws_stream_set && g_ws_stream == s) return FLUXHIP_OK;
  if (g_ws_stream_set) {
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(g_ws_stream, &a);
    (void)hipStreamIsCapturing(s, &b);
    (void)hipGetLastError();
    // a capturing stream cannot be joined from outside the capture (and a graph is ordered by the stream that replays it)
    if (a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone) {
      if (!g_ws_event && hipEventCreateWithFlags(&g_ws_event, hipEventDisableTiming) != hipSuccess) return FLUXHIP_ELAUNCH;
      if (hipEventRecord(g_ws_event, g_ws_stream) != hipSuccess || hipStreamWaitEvent(s, g_ws_event, 0) != hipSuccess) {
        (void)hipGetLastError();      // the previous stream was destroyed: whatever it ran has been synchronised by its owner
      }
    }
  }
  g_ws_stream = s;
  g_ws_stream_set = true;
  return FLUXHIP_OK;
}
// FLUXHIP_LEAN=0 (or fluxhip_gemm_set_lean(0)): every launch on the generic kernels (A/B runs, tests)
bool g_lean_enabled = [] { const char* e = getenv("FLUXHIP_LEAN"); return !(e && e[0] == '0'); }();
long long g_lean_launches = 0;

// Can a split-K launch of `cfg` with S splits over `tiles` output tiles use the reduce-scatter hand-off?
bool rs_ok(int cfg, int S, long long tiles, bool conv, bool x3, bool f8) {
  if (!g_rs_enabled || conv || x3 || f8 || cfg <= 0 || cfg >= kNumCfgs || S < 2) return false;
  const TileCfg& t = kCfgs[cfg];
  if (!t.dense_rs) return false;
  const int mi = t.bm / t.wm / 16, nj = t.bn / (t.threads / 64 / t.wm) / 16;
  if (mi % S && nj % S) return false;
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
    g_num_cus = pr.multiProcessorCount;
  }
  // (performance, not safety: a grid that is not resident as a whole completes through the orphan path of the hand-off)
  if ((tiles * S > g_num_cus && !g_rs_any_grid) || tiles > kSkMaxTiles / 2) return false;
  return kSkFlagBytes + tiles * S * t.bm * t.bn * 4LL <= g_ws_bytes;
}

// returns cfg | (splits << 8)
template <int NC, bool RS = false, int MAXS = 4>
int pick_from(const Cand (&cands)[NC], const int* group_m, int ngroups, int nbatch, int N, int K,
              const CostForm& form = kPlainForm) {
  float best = 3.4e38f;
  int best_cfg = 4;
  const int nkt = K / 64;
  for (const Cand& c : cands) {
    const TileCfg& t = kCfgs[c.cfg];
    long long tiles = 0;
    for (int g = 0; g < ngroups; ++g) tiles += (long long)((group_m[g] + t.bm - 1) / t.bm) * nbatch;
    tiles *= (N + t.bn - 1) / t.bn;
    const long long slots = 256LL * c.bpc;
    for (int S = 1; S <= MAXS; ++S) {
      if (S > 1) {   // only under-filled grids with a long K loop, and only if the workspace can hold the partials
        if (c.bpc != 1 || tiles * S > slots || nkt / S < 16 || tiles > kSkMaxTiles) break;
        if (kSkFlagBytes + tiles * t.bm * t.bn * 4LL > g_ws_bytes) break;
      }
      const float x = (float)(tiles * S) / (float)slots;
      const float whole = (float)((tiles * S + slots - 1) / slots);
      const float rounds = x <= 1.f ? 1.f : (1.f - form.beta) * x + form.beta * whole;
      const float fill = c.phi + (1.f - c.phi) * (x < 1.f ? x : 1.f);
      const float hop = S == 1 ? 0.f
                        : (RS && rs_ok(c.cfg, S, tiles, false, false, false)) ? kRsHopUs + (float)(S - 2) * kRsHopNextUs
                                                                             : kHopUs + (float)(S - 2) * kHopNextUs;
      const float cost =
          form.t_launch_us + rounds * fill * ((float)((nkt + S - 1) / S) * c.t_step_us + c.t_fixed_us) + hop;
      if (cost < best) { best = cost; best_cfg = c.cfg | (S << 8); }
    }
  }
  return best_cfg;
}

// rs: the launch can take the reduce-scatter split-K kernels (lean-eligible gate-residual epilogue on bf16), so a split is
// priced with the cheap hand-off; every other launch that splits runs the chain and is priced with it
int pick_cfg(const int* group_m, int ngroups, int nbatch, int N, int K, bool conv = false, bool x3 = false,
             bool f8 = false, bool rs = false, int mx = 0) {
  if (f8 && mx == 1) return pick_from<sizeof(kF8MxaCands) / sizeof(Cand), false, 1>(kF8MxaCands, group_m, ngroups, nbatch, N, K / 2);
  if (f8 && mx == 2) return pick_from<sizeof(kF8MxcCands) / sizeof(Cand), false, 1>(kF8MxcCands, group_m, ngroups, nbatch, N, K / 2);
  if (f8) return pick_from(kF8Cands, group_m, ngroups, nbatch, N, K / 2);   // 128 elements per K-step
  if (x3)   // three passes over K
    return conv ? pick_from(kX3ConvCands, group_m, ngroups, nbatch, N, 3 * K)
                : pick_from(kX3Cands, group_m, ngroups, nbatch, N, 3 * K);
  if (!conv) {
    // Skinny launches (the text towers: M = 256 rows against 34-84 MB of weights) are HBM streaming, not MFMA work, and the cost
    // model - fitted at M >= 1280 - under-prices a split for them: at M = 256 the 128 x 128 tile covers N = 4096 / 8192 with 64 /
    // 128 blocks, and splitting K until ~256 blocks are in flight is what raises the weight stream (tools/gemm_tune.py, cold
    // weights: q/k 423 -> 511 TFLOP/s with S = 2, out-proj 212 -> 281 and wo 369 -> 426 with S = 3).  M = 512 is left to the model.
    long long m_total = 0;
    for (int g = 0; g < ngroups; ++g) m_total += (long long)group_m[g] * nbatch;
    long long tiles = 0;
    for (int g = 0; g < ngroups; ++g) tiles += (long long)((group_m[g] + 127) / 128) * nbatch;
    tiles *= (N + 127) / 128;
    if (m_total <= 256 && K >= 2048 && N >= 1024 && tiles <= 128) {
      int S = (int)(256 / tiles);
      S = S > 3 ? 3 : S;
      while (S > 1 && (K / 64) / S < 16) --S;
      if (S > 1 && kSkFlagBytes + tiles * 128 * 128 * 4LL <= g_ws_bytes) return 47 | (S << 8);
    }
  }
  return conv ? pick_from(kConvCands, group_m, ngroups, nbatch, N, K, kConvForm)
              : rs ? pick_from<sizeof(kCands) / sizeof(kCands[0]), true>(kCands, group_m, ngroups, nbatch, N, K, kDenseForm)
                   : pick_from(kCands, group_m, ngroups, nbatch, N, K, kDenseForm);
}

// mx (fp8 only): 0 = per-token activation scales; 1 = block-scaled activation operand (FLAG_MXA); 2 = e4m3 + block-scale output (FLAG_MXC)
int launch(GemmParams& p, int cfg_code, bool conv, hipStream_t s, bool x3 = false, bool f8 = false, bool f16 = false, int mx = 0) {
  int cfg_idx = cfg_code & 0xff;
  if (!conv && cfg_idx >= 49 && cfg_idx <= 57) {
    // the ping-pong tiles address their dense operands as scalar base + 32-bit byte offset per (group, batch); an
    // operand of 4 GiB or more (no product shape comes near: 65536 x 5120 bf16 is 0.67 GB) goes to the plain-ring
    // tile of the same shape (fp8: the 256 x 256 simple ring)
    const long long esz = f8 ? 1 : 2;
    bool ok = (long long)p.N * p.K * esz < (1ll << 32);
    for (int g = 0; g < p.ngroups; ++g) ok = ok && (long long)p.g[g].M * p.lda * esz < (1ll << 32);
    static const int plain[9] = {15, 18, 19, 10, 7, 23, 14, 19, 7};   // 49..57 -> same tile shape, PIPE 1 (53: 128 x 128 is cfg 7; 57: 128 x 160 has no plain twin, 128 x 128)
    if (!ok) cfg_idx = f8 ? 6 : (x3 && cfg_idx != 49 ? 15 : plain[cfg_idx - 49]);
  }
  int splits = cfg_code >> 8;
  if (splits < 1) splits = 1;
  if (cfg_idx <= 0 || cfg_idx >= kNumCfgs || splits > 16) return FLUXHIP_EINVAL;
  const TileCfg& c = kCfgs[cfg_idx];
  int tm_total = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    p.g[g].tiles_m = (p.g[g].M + c.bm - 1) / c.bm;
    tm_total += p.g[g].tiles_m * p.nbatch;
  }
  if (p.ngroups == 1) p.g[1] = p.g[0];
  p.tiles_m_total = tm_total;
  p.tiles_n = (p.N + c.bn - 1) / c.bn;
  auto fn = f16 ? (conv ? c.conv_f16 : c.dense_f16)
            : f8 ? c.dense_f8 : x3 ? (conv ? c.conv_x3 : c.dense_x3) : (conv ? c.conv : c.dense);
  if (!fn) return FLUXHIP_EINVAL;                   // this tile has no fp32-faithful / fp8 / float16 instantiation

  const int slot = f16 ? 9 + (int)conv : f8 ? 4 : (int)conv + 2 * (int)x3;
  if (!g_attr_set[cfg_idx][slot]) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds) !=
        hipSuccess)
      return FLUXHIP_ELAUNCH;
    g_attr_set[cfg_idx][slot] = true;
  }
  p.trace = g_trace;
  if (conv) {
    // buffer-addressed activation loader (ping-pong conv tiles, gemm_core.h ConvGeom::buf): the image, its lo plane and the bias
    // that keeps lane offsets non-negative must fit the resource's 32-bit num_records with room for the out-of-range marker
    static const bool buf_off = [] { const char* e = getenv("FLUXHIP_CONV_BUF"); return e && e[0] == '0'; }();
    const long long bias_b = (2LL * p.cv.Ws + 2) * p.cv.Cin * 2;
    p.cv.buf = (!buf_off && !p.cv.ups && p.cv.x_extent > 0 && p.cv.x_extent + bias_b + 4096 < 0xFFF00000LL) ? 1 : 0;
  }
  // LDS-transposed epilogue needs every output-side operand addressable in 16-byte units
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  bool wide = !x3 && !p.out_f32 && p.N % 8 == 0 && p.ldc % 8 == 0;
  for (int g = 0; g < p.ngroups && wide; ++g) {
    const GemmGroup& t = p.g[g];
    wide = a16(t.C) && t.c_bstride % 8 == 0 && a16(t.res) && a16(t.gate) && t.gate_bstride % 8 == 0;
  }
  if (p.epi == EPI_SPLIT_GELU)
    wide = wide && a16(p.C2) && p.n_split % 8 == 0 && p.ldc2 % 8 == 0 && p.c2_coloff % 8 == 0 && p.c2_bstride % 8 == 0;
  p.wide_epi = wide;
  p.splits = splits;
  // what a FLAG_LEAN instantiation can do (with_lean above): the transformer-block epilogues on the LDS-transposed path
  const bool lean_any = !conv && !x3 && wide && !p.addvec && !p.row_bias && !p.out_f32 &&
                        (p.epi == EPI_BIAS || p.epi == EPI_GELU_TANH || p.epi == EPI_GATE_RES || p.epi == EPI_SPLIT_GELU);
  const bool lean_ok = lean_any && !f8 && !f16;      // (float16 launches: generic kernels, chain split-K)
  const bool lean_on = g_lean_enabled;
  auto use = [&](void (*k)(const GemmParams), int slot_) {      // a variant kernel of this tile: dynamic LDS attribute once
    if (!g_attr_set[cfg_idx][slot_]) {
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds) != hipSuccess) return false;
      g_attr_set[cfg_idx][slot_] = true;
    }
    fn = k;
    return true;
  };
  void (*const generic)(const GemmParams) = fn;
  if (p.epi == EPI_GEGLU_PAIR) {                  // lives in its own instantiations: no generic fallback
    if (conv || x3 || f8 || splits != 1 || !wide || p.addvec || p.row_bias || p.out_f32 || !c.dense_pair || p.N % 32) return FLUXHIP_EINVAL;
    if (f16 && !c.dense_pair_f16) return FLUXHIP_EINVAL;
    if (!(f16 ? use(c.dense_pair_f16, 11) : use(c.dense_pair, 8))) return FLUXHIP_ELAUNCH;
  } else
  if (mx) {                                       // block-scaled fp8: lean kernels of their own, no generic fallback
    void (*const k)(const GemmParams) = mx == 1 ? c.dense_f8_mxa : c.dense_f8_mxc;
    if (!f8 || !k || splits != 1 || !lean_any) return FLUXHIP_EINVAL;
    if (mx == 2 && p.epi != EPI_GELU_TANH && p.epi != EPI_SPLIT_GELU) return FLUXHIP_EINVAL;
    if (!use(k, 11 + mx)) return FLUXHIP_ELAUNCH;
  } else
  if (splits == 1 && lean_on) {
    bool ok = true;
    if (lean_ok && c.dense_lean && p.epi == c.lean_epi) ok = use(c.dense_lean, 6);
    else if (f8 && !f16 && lean_any && c.dense_f8_lean) ok = use(c.dense_f8_lean, 7);
    if (!ok) return FLUXHIP_ELAUNCH;
    g_lean_launches += fn != generic;
  }
  if (splits > 1) {
    const long long tiles = (long long)tm_total * p.tiles_n;
    if (splits > (f8 ? p.K / 128 : (x3 ? 3 : 1) * (p.K / 64)) || tiles > kSkMaxTiles ||
        kSkFlagBytes + tiles * c.bm * c.bn * 4LL > g_ws_bytes)
      return FLUXHIP_EINVAL;                        // no (or too small a) split-K workspace
    p.sk_flag = (int*)g_ws;
    p.sk_part = (float*)(g_ws + kSkFlagBytes);
    p.sk_mode = (lean_ok && p.epi == c.rs_epi && rs_ok(cfg_idx, splits, tiles, conv, x3, f8)) ? 1 : 0;   // (the reduce-scatter kernels are lean: LDS-transposed epilogue, transformer-block epilogues)
    p.sk_depart = p.sk_flag + kSkMaxTiles / 2;
    p.sk_timeout = g_rs_timeout_ticks;
    if (serialize_workspace_user(s) != FLUXHIP_OK) return FLUXHIP_ELAUNCH;
    g_rs_launches += p.sk_mode != 0;
    if (p.sk_mode && !use(c.dense_rs, 5)) return FLUXHIP_ELAUNCH;
  }
  int lds = c.lds;
  // Halo-tile loader (FLAG_DXR, gemm_core.h): 3 x 3 / stride 1 / pad 1 fp32-faithful convs on the 256 x 128 tile whose tiles are 256
  // pixels of one image row - the N = 128 layers of the VAE decoders at 512 x 512 and up.  FLUXHIP_CONV_DXR=0: the tap-by-tap loader (A/B).
  const bool dxr = conv && x3 && cfg_idx == 52 && splits == 1 && g_conv_dxr && p.cv.ksize == 3 && p.cv.stride == 1 && p.cv.pad == 1 &&
                   !p.cv.ups && !p.cv.sub2 && p.cv.buf && p.cv.Ws % 256 == 0 && p.cv.Ho == p.cv.Hs && p.cv.Wo == p.cv.Ws && p.nbatch == 1;
  if (dxr) {
    void (*const dk)(const GemmParams) = g_trace ? gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR | FLAG_TIMED>
                                                 : gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_DXR>;
    lds = 2 * 33 * 1024 + 3 * c.bn * 128;       // two halo slots of 264 rows + the weight ring
    g_conv_dxr_launches.fetch_add(1, std::memory_order_relaxed);
    bool attr_ok = true;
    std::call_once(g_dxr_attr_once[g_trace ? 1 : 0], [&] {
      attr_ok = hipFuncSetAttribute((const void*)dk, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    });
    if (!attr_ok) return FLUXHIP_ELAUNCH;
    fn = dk;
  } else if (conv && x3 && g_trace && (cfg_idx == 49 || cfg_idx == 52)) {      // diagnostic: the stamped twin of the fp32-faithful conv tile
    void (*const tk)(const GemmParams) = cfg_idx == 49 ? gemm_nt_kernel<256, 256, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>
                                                       : gemm_nt_kernel<256, 128, 4, 2, 1, 2, 6, FLAG_SPLIT | FLAG_TIMED>;
    if (hipFuncSetAttribute((const void*)tk, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds) != hipSuccess) return FLUXHIP_ELAUNCH;
    fn = tk;
  }
  dim3 grid(tm_total * p.tiles_n * splits), block(c.threads);
  hipLaunchKernelGGL(fn, grid, block, lds, s, p);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// Arms the GroupNorm-statistics output of a FLAG_SPLIT conv launch (GemmParams::gn_ws) when the chosen tile allows it:
// the per-image pixel count of the GEMM's M space must be a multiple of the tile height and the workspace must hold
// [B][nchunks][N][2] partials plus the [B][64][2] statistics gn_finalize appends.  Returns the chunk count (0 = not armed:
// the consumer computes the partials itself).
int arm_gn_stats(GemmParams& p, int cfg_code, int B, int hw, int parities, void* gn_ws, int64_t gn_ws_bytes) {
  if (!gn_ws) return 0;
  const int idx = cfg_code & 0xff;      // e.g. a forced FLUXHIP_CONV_X3_CFG: launch() rejects it, do not index the table first
  if (idx <= 0 || idx >= kNumCfgs || !kCfgs[idx].conv_x3) return 0;
  const TileCfg& c = kCfgs[idx];
  const int wtm = c.bm / c.wm;
  if (hw % c.bm || p.N % 4) return 0;
  const int nchunks = parities * (hw / wtm);
  if (((int64_t)B * nchunks * p.N + (int64_t)B * 64) * 2 * (int64_t)sizeof(float) > gn_ws_bytes) return 0;
  p.gn_ws = (float*)gn_ws;
  p.gn_hw = hw;
  p.gn_nchunks = nchunks;
  return nchunks;
}

}  // namespace

// fluxhip_gemm_desc -> GemmParams (shared by the bf16 and the fp8 entry points); kalign = K granularity
static int params_from_desc(const fluxhip_gemm_desc* d, GemmParams& p, int kalign) {
  if (!d || d->ngroups < 1 || d->ngroups > 2 || d->nbatch < 1) return FLUXHIP_EINVAL;
  if (d->K <= 0 || d->K % kalign || d->N <= 0 || d->N % 4 || d->lda % (kalign == 128 ? 16 : 8) || d->ldc % 4)
    return FLUXHIP_EINVAL;
  for (int g = 0; g < d->ngroups; ++g) {
    const fluxhip_gemm_group& s = d->g[g];
    if (!s.A || !s.W || !s.C || s.M <= 0) return FLUXHIP_EINVAL;
    if ((d->epi == FLUXHIP_EPI_GATE_RES || d->epi == FLUXHIP_EPI_GEGLU) && !s.res) return FLUXHIP_EINVAL;
    GemmGroup& t = p.g[g];
    t.A = (const bf16_t*)s.A;
    t.W = (const bf16_t*)s.W;
    t.bias = (const bf16_t*)s.bias;
    t.C = (bf16_t*)s.C;
    t.res = (const bf16_t*)s.res;
    t.gate = (const bf16_t*)s.gate;
    t.a_bstride = s.a_bstride;
    t.c_bstride = s.c_bstride;
    t.gate_bstride = s.gate_bstride;
    t.w_bstride = s.w_bstride;
    t.M = s.M;
    t.addm = (const bf16_t*)s.add;
    t.addm_bstride = s.add_bstride;
    if ((s.add != nullptr) != (d->g[0].add != nullptr)) return FLUXHIP_EINVAL;     // all groups or none
  }
  if (d->g[0].add) {
    if (d->ld_add < d->N || d->ld_add % 4 || d->row_bias || d->out_f32) return FLUXHIP_EINVAL;
    p.addvec = (const bf16_t*)d->g[0].add;      // non-null: selects the addend epilogue; the kernel reads the group's matrix
    p.addvec_rows = 1;
    p.addvec_stride = d->ld_add;
  }
  if (d->epi == FLUXHIP_EPI_SPLIT_GELU && (!d->C2 || d->n_split % 4 || d->ldc2 % 4 || d->c2_coloff % 4))
    return FLUXHIP_EINVAL;
  p.ngroups = d->ngroups;
  p.nbatch = d->nbatch;
  p.N = d->N;
  p.K = d->K;
  p.lda = d->lda;
  p.ldc = d->ldc;
  p.epi = d->epi;
  p.row_bias = d->row_bias;
  p.n_split = d->n_split;
  p.C2 = (bf16_t*)d->C2;
  p.ldc2 = d->ldc2;
  p.c2_bstride = d->c2_bstride;
  p.c2_coloff = d->c2_coloff;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.out_f32 = d->out_f32;
  if (p.out_f32 && d->epi != FLUXHIP_EPI_BIAS) return FLUXHIP_EINVAL;
  return FLUXHIP_OK;
}

// does launch() take the reduce-scatter kernel when this descriptor splits?  (what launch() tests: lean_ok && epi == rs_epi)
static bool rs_epilogue(const fluxhip_gemm_desc* d) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  bool ok = d->epi == FLUXHIP_EPI_GATE_RES && !d->g[0].add && !d->row_bias && !d->out_f32 && d->N % 8 == 0 && d->ldc % 8 == 0;
  for (int g = 0; g < d->ngroups && ok; ++g)
    ok = a16(d->g[g].C) && d->g[g].c_bstride % 8 == 0 && a16(d->g[g].res) && a16(d->g[g].gate) && d->g[g].gate_bstride % 8 == 0;
  return ok;
}

static int gemm_dense16(const fluxhip_gemm_desc* d, void* stream, bool f16) {
  GemmParams p{};
  if (int rc = params_from_desc(d, p, 64)) return rc;
  const int gm[2] = {d->g[0].M, d->ngroups > 1 ? d->g[1].M : 0};
  int cfg = d->tile_cfg > 0 ? d->tile_cfg : pick_cfg(gm, d->ngroups, d->nbatch, d->N, d->K, false, false, false, !f16 && rs_epilogue(d));
  if (d->epi == FLUXHIP_EPI_GEGLU_PAIR && d->tile_cfg <= 0) {
    // the pair epilogue exists for 256x256 and 128x256 only: the large tile once it fills most of the chip
    long long t49 = 0;
    for (int g = 0; g < d->ngroups; ++g) t49 += (long long)((gm[g] + 255) / 256) * d->nbatch;
    t49 *= (d->N + 255) / 256;
    cfg = t49 >= 192 ? 49 : 55;
  }
  return launch(p, cfg, false, (hipStream_t)stream, false, false, f16);
}

extern "C" int fluxhip_gemm_bf16(const fluxhip_gemm_desc* d, void* stream) { return gemm_dense16(d, stream, false); }
// the same operator on IEEE float16 storage (v_mfma_f32_16x16x32_f16, fp32 accumulate): the stable_diffusion/ models with float16=True
extern "C" int fluxhip_gemm_f16(const fluxhip_gemm_desc* d, void* stream) { return gemm_dense16(d, stream, true); }

extern "C" int fluxhip_gemm_fp8(const fluxhip_gemm_desc* d, const fluxhip_fp8_scales* sc, void* stream) {
  GemmParams p{};
  if (!sc) return FLUXHIP_EINVAL;
  if (int rc = params_from_desc(d, p, 128)) return rc;
  for (int g = 0; g < d->ngroups; ++g) {
    if (!sc->a_scale[g] || !sc->w_scale[g]) return FLUXHIP_EINVAL;
    p.a_scale[g] = (const float*)sc->a_scale[g];
    p.w_scale[g] = (const float*)sc->w_scale[g];
  }
  if (d->ngroups == 1) { p.a_scale[1] = p.a_scale[0]; p.w_scale[1] = p.w_scale[0]; }
  p.a_sc_bstride = sc->a_scale_bstride;
  const int gm[2] = {d->g[0].M, d->ngroups > 1 ? d->g[1].M : 0};
  int cfg = d->tile_cfg > 0 ? d->tile_cfg : pick_cfg(gm, d->ngroups, d->nbatch, d->N, d->K, false, false, true);
  return launch(p, cfg, false, (hipStream_t)stream, false, true);
}

// Block-scaled (MX) form: see include/fluxhip.h.
extern "C" int fluxhip_gemm_fp8_mx(const fluxhip_gemm_desc* d, const fluxhip_fp8_scales* sc, const fluxhip_fp8_mx* mx, void* stream) {
  GemmParams p{};
  if (!sc || !mx) return FLUXHIP_EINVAL;
  const bool mxa = mx->a_mx != nullptr, mxc = mx->c_mx != nullptr;
  if (mxa == mxc) return FLUXHIP_EINVAL;          // one of the two per launch (the producer's own input comes out of LayerNorm: per token)
  if (int rc = params_from_desc(d, p, 128)) return rc;
  for (int g = 0; g < d->ngroups; ++g) {
    if (!sc->w_scale[g] || (!mxa && !sc->a_scale[g])) return FLUXHIP_EINVAL;
    if (d->g[g].M % 64) return FLUXHIP_EINVAL;    // whole 64-row groups of the scale tiling per wave
    p.a_scale[g] = (const float*)sc->a_scale[g];
    p.w_scale[g] = (const float*)sc->w_scale[g];
  }
  if (d->ngroups == 1) { p.a_scale[1] = p.a_scale[0]; p.w_scale[1] = p.w_scale[0]; }
  p.a_sc_bstride = sc->a_scale_bstride;
  const int ng = d->ngroups;
  if (mxa) {
    if (mx->a_mx_bstride % 64 || mx->a_mx_kstride % 64 || mx->a_mx_kstride <= 0 || ((uintptr_t)mx->a_mx & 3)) return FLUXHIP_EINVAL;
    for (int g = 0; g < ng; ++g) {
      if (mx->a_mx_row0[g] < 0 || mx->a_mx_row0[g] % 64) return FLUXHIP_EINVAL;
      if (mx->a_mx_row0[g] + (d->nbatch - 1) * mx->a_mx_bstride + d->g[g].M > mx->a_mx_kstride) return FLUXHIP_EINVAL;
      p.a_mx_row0[g] = mx->a_mx_row0[g];
    }
    if (ng == 1) p.a_mx_row0[1] = p.a_mx_row0[0];
    if ((long long)(d->K / 128) * mx->a_mx_kstride * 4 >= (1ll << 32)) return FLUXHIP_EINVAL;   // 32-bit offsets in the K loop
    p.a_mx = (const uint32_t*)mx->a_mx;
    p.a_mx_bstride = mx->a_mx_bstride;
    p.a_mx_kstride = mx->a_mx_kstride;
  } else {
    const bool split = d->epi == FLUXHIP_EPI_SPLIT_GELU;
    if (d->epi != FLUXHIP_EPI_GELU_TANH && !split) return FLUXHIP_EINVAL;
    if (d->N % 32 || mx->ldc8 % 8 || mx->c8_bstride % 8 || mx->c_mx_bstride % 64 || mx->c_mx_kstride % 64 || mx->c_mx_kstride <= 0)
      return FLUXHIP_EINVAL;
    if (split && (d->n_split % 32 || mx->c8_coloff % 32 || mx->c8_coloff < 0)) return FLUXHIP_EINVAL;
    for (int g = 0; g < ng; ++g) {
      if (!mx->c8[g] || ((uintptr_t)mx->c8[g] & 7) || mx->c_mx_row0[g] < 0 || mx->c_mx_row0[g] % 64) return FLUXHIP_EINVAL;
      if (mx->c_mx_row0[g] + (d->nbatch - 1) * mx->c_mx_bstride + d->g[g].M > mx->c_mx_kstride) return FLUXHIP_EINVAL;
      p.c8[g] = (uint8_t*)mx->c8[g];
      p.c_mx_row0[g] = mx->c_mx_row0[g];
    }
    if (ng == 1) { p.c8[1] = p.c8[0]; p.c_mx_row0[1] = p.c_mx_row0[0]; }
    p.c8_bstride = mx->c8_bstride;
    p.ldc8 = mx->ldc8;
    p.c8_coloff = split ? mx->c8_coloff : 0;
    p.c_mx = (uint8_t*)mx->c_mx;
    p.c_mx_bstride = mx->c_mx_bstride;
    p.c_mx_kstride = mx->c_mx_kstride;
  }
  const int gm[2] = {d->g[0].M, ng > 1 ? d->g[1].M : 0};
  const int mode = mxa ? 1 : 2;
  const int cfg = d->tile_cfg > 0 ? d->tile_cfg : pick_cfg(gm, ng, d->nbatch, d->N, d->K, false, false, true, false, mode);
  return launch(p, cfg, false, (hipStream_t)stream, false, true, false, mode);
}

extern "C" int fluxhip_gemm_fp8_tile_cfg(const fluxhip_gemm_desc* d) {
  if (!d || d->ngroups < 1 || d->ngroups > 2) return FLUXHIP_EINVAL;
  if (d->tile_cfg > 0) return d->tile_cfg < kNumCfgs ? d->tile_cfg : FLUXHIP_EINVAL;
  const int gm[2] = {d->g[0].M, d->ngroups > 1 ? d->g[1].M : 0};
  return pick_cfg(gm, d->ngroups, d->nbatch, d->N, d->K, false, false, true);
}

extern "C" int fluxhip_gemm_x3(const fluxhip_gemm_x3_desc* d, void* stream) {
  if (!d || d->nbatch < 1 || !d->A || !d->W || !d->C || d->M <= 0) return FLUXHIP_EINVAL;
  if (d->K <= 0 || d->K % 64 || d->N <= 0 || d->N % 4 || d->lda % 8 || d->ldc % 4) return FLUXHIP_EINVAL;
  if (d->epi != FLUXHIP_EPI_BIAS && d->epi != FLUXHIP_EPI_GATE_RES) return FLUXHIP_EINVAL;
  if (d->epi == FLUXHIP_EPI_GATE_RES && (!d->res || d->out_f32)) return FLUXHIP_EINVAL;
  if ((d->a_lo | d->w_lo) % 8 || (d->c_lo | d->res_lo) % 4) return FLUXHIP_EINVAL;   // 16-byte staging / 8-byte stores
  GemmParams p{};
  GemmGroup& t = p.g[0];
  t.A = (const bf16_t*)d->A;
  t.W = (const bf16_t*)d->W;
  t.bias = (const bf16_t*)d->bias;                  // float32 in this mode (the kernel casts back)
  t.C = (bf16_t*)d->C;
  t.res = (const bf16_t*)d->res;
  t.a_bstride = d->a_bstride;
  t.c_bstride = d->c_bstride;
  t.w_bstride = d->w_bstride;
  t.M = d->M;
  p.ngroups = 1;
  p.nbatch = d->nbatch;
  p.N = d->N;
  p.K = d->K;
  p.lda = d->lda;
  p.ldc = d->ldc;
  p.epi = d->epi;
  p.row_bias = d->row_bias;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.out_f32 = d->out_f32;
  p.a_lo = d->a_lo;
  p.w_lo = d->w_lo;
  p.c_lo = d->c_lo;
  p.res_lo = d->res_lo;
  int cfg = d->tile_cfg > 0 ? d->tile_cfg : pick_cfg(&t.M, 1, d->nbatch, d->N, d->K, false, true);
  return launch(p, cfg, false, (hipStream_t)stream, true);
}

extern "C" int fluxhip_gemm_tile_cfg(const fluxhip_gemm_desc* d) {
  if (!d || d->ngroups < 1 || d->ngroups > 2) return FLUXHIP_EINVAL;
  if (d->tile_cfg > 0) return d->tile_cfg < kNumCfgs ? d->tile_cfg : FLUXHIP_EINVAL;
  const int gm[2] = {d->g[0].M, d->ngroups > 1 ? d->g[1].M : 0};
  return pick_cfg(gm, d->ngroups, d->nbatch, d->N, d->K, false, false, false, rs_epilogue(d));
}

extern "C" int fluxhip_set_workspace(void* ws, int64_t bytes) {
  if (ws && (bytes < (int64_t)kSkFlagBytes || ((uintptr_t)ws & 255))) return FLUXHIP_EINVAL;
  g_ws = (char*)ws;
  g_ws_bytes = ws ? bytes : 0;
  return FLUXHIP_OK;
}

extern "C" int fluxhip_gemm_set_splitk_mode(int mode) {
  if (mode < 0 || mode > 2) return FLUXHIP_EINVAL;
  g_rs_enabled = mode != 1;
  g_rs_any_grid = mode == 2;
  return FLUXHIP_OK;
}

extern "C" int64_t fluxhip_gemm_rs_launches(void) { return g_rs_launches; }

extern "C" int fluxhip_gemm_set_rs_timeout_us(int us) {
  if (us < 0 || us > 10000000) return FLUXHIP_EINVAL;
  g_rs_timeout_ticks = us * 100;
  return FLUXHIP_OK;
}

extern "C" int fluxhip_gemm_set_lean(int on) {
  g_lean_enabled = on != 0;
  return FLUXHIP_OK;
}

extern "C" int64_t fluxhip_gemm_lean_launches(void) { return g_lean_launches; }

extern "C" int fluxhip_conv_set_x3_tile(int cfg, int dxr) {
  if (cfg < 0 || (cfg & 0xff) >= kNumCfgs) return FLUXHIP_EINVAL;
  g_conv_x3_cfg = cfg;
  if (dxr >= 0) g_conv_dxr = dxr != 0;
  return FLUXHIP_OK;
}
extern "C" int64_t fluxhip_conv_dxr_launches(void) { return g_conv_dxr_launches; }

extern "C" int fluxhip_gemm_set_trace(void* buf) {
  g_trace = (unsigned long long*)buf;
  return FLUXHIP_OK;
}

extern "C" int fluxhip_gemm_tile_shape(int cfg, int* bm, int* bn, int* threads) {
  cfg &= 0xff;                                      // accepts the cfg | splits << 8 code of fluxhip_gemm_tile_cfg
  if (cfg <= 0 || cfg >= kNumCfgs || !bm || !bn || !threads) return FLUXHIP_EINVAL;
  *bm = kCfgs[cfg].bm;
  *bn = kCfgs[cfg].bn;
  *threads = kCfgs[cfg].threads;
  return FLUXHIP_OK;
}

static int conv2d_16(const void* x, const void* w, const void* bias, const void* res,
                     const void* addvec, void* out, int B, int Hs, int Ws, int Cin,
                     int Cout, int ksize, int stride, int pad, int ups, int epi,
                     const void* zero16, void* stream, bool f16) {
  if (!x || !w || !out || !zero16) return FLUXHIP_EINVAL;
  if (Cin % 64 || Cout % 4 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return FLUXHIP_EINVAL;
  if (epi != FLUXHIP_EPI_BIAS && epi != FLUXHIP_EPI_GATE_RES && epi != FLUXHIP_EPI_SILU)
    return FLUXHIP_EINVAL;
  if (epi == FLUXHIP_EPI_GATE_RES && !res) return FLUXHIP_EINVAL;
  const int Hl = ups ? Hs * 2 : Hs, Wl = ups ? Ws * 2 : Ws;
  const int Ho = (Hl + 2 * pad - ksize) / stride + 1;
  const int Wo = (Wl + 2 * pad - ksize) / stride + 1;
  GemmParams p{};
  p.cv.X = (const bf16_t*)x;
  p.cv.zero = (const bf16_t*)zero16;
  p.cv.Hs = Hs; p.cv.Ws = Ws; p.cv.Ho = Ho; p.cv.Wo = Wo;
  p.cv.Cin = Cin; p.cv.ksize = ksize; p.cv.stride = stride; p.cv.pad = pad; p.cv.ups = ups;
  p.cv.x_extent = (long long)B * Hs * Ws * Cin * 2;
  GemmGroup& t = p.g[0];
  t.A = (const bf16_t*)x;
  t.W = (const bf16_t*)w;
  t.bias = (const bf16_t*)bias;
  t.C = (bf16_t*)out;
  t.res = (const bf16_t*)res;
  t.gate = nullptr;
  t.M = B * Ho * Wo;
  p.ngroups = 1;
  p.nbatch = 1;
  p.N = Cout;
  p.K = ksize * ksize * Cin;
  p.lda = Cin;
  p.ldc = Cout;
  p.epi = epi;
  p.alpha = 1.f;
  p.addvec = (const bf16_t*)addvec;
  p.addvec_rows = Ho * Wo;
  p.addvec_stride = Cout;
  static const int forced = [] { const char* e = getenv("FLUXHIP_CONV_CFG"); return e ? atoi(e) : 0; }();   // tuning knob
  int cfg = forced > 0 ? forced : pick_cfg(&t.M, 1, 1, Cout, p.K, true);
  if (forced <= 0 && ups && (cfg & 0xff) == 49) cfg = (cfg & ~0xff) | 15;     // fused-upsample loader: the plain ring is 2-4 % faster
  return launch(p, cfg, true, (hipStream_t)stream, false, false, f16);
}

extern "C" int fluxhip_conv2d_bf16(const void* x, const void* w, const void* bias, const void* res,
                                   const void* addvec, void* out, int B, int Hs, int Ws, int Cin,
                                   int Cout, int ksize, int stride, int pad, int ups, int epi,
                                   const void* zero16, void* stream) {
  return conv2d_16(x, w, bias, res, addvec, out, B, Hs, Ws, Cin, Cout, ksize, stride, pad, ups, epi, zero16, stream, false);
}
extern "C" int fluxhip_conv2d_f16(const void* x, const void* w, const void* bias, const void* res,
                                  const void* addvec, void* out, int B, int Hs, int Ws, int Cin,
                                  int Cout, int ksize, int stride, int pad, int ups, int epi,
                                  const void* zero16, void* stream) {
  return conv2d_16(x, w, bias, res, addvec, out, B, Hs, Ws, Cin, Cout, ksize, stride, pad, ups, epi, zero16, stream, true);
}

extern "C" int fluxhip_conv2d_x3(const void* x, int64_t x_lo, const void* w, int64_t w_lo, const void* bias,
                                 const void* res, int64_t res_lo, void* out, int64_t out_lo, int B, int Hs,
                                 int Ws, int Cin, int Cout, int ksize, int stride, int pad, int ups,
                                 void* gn_ws, int64_t gn_ws_bytes, int* gn_nchunks, const void* zero16,
                                 void* stream) {
  if (gn_nchunks) *gn_nchunks = 0;
  if (!x || !w || !out || !zero16 || (gn_ws && !gn_nchunks)) return FLUXHIP_EINVAL;
  if (Cin % 64 || Cout % 4 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return FLUXHIP_EINVAL;
  if ((x_lo | w_lo) % 8 || (out_lo | res_lo) % 4) return FLUXHIP_EINVAL;
  const int Hl = ups ? Hs * 2 : Hs, Wl = ups ? Ws * 2 : Ws;
  const int Ho = (Hl + 2 * pad - ksize) / stride + 1;
  const int Wo = (Wl + 2 * pad - ksize) / stride + 1;
  GemmParams p{};
  p.cv.X = (const bf16_t*)x;
  p.cv.zero = (const bf16_t*)zero16;
  p.cv.Hs = Hs; p.cv.Ws = Ws; p.cv.Ho = Ho; p.cv.Wo = Wo;
  p.cv.Cin = Cin; p.cv.ksize = ksize; p.cv.stride = stride; p.cv.pad = pad; p.cv.ups = ups;
  p.cv.x_extent = x_lo >= 0 ? (x_lo + (long long)B * Hs * Ws * Cin) * 2 : 0;       // hi plane, then the lo plane x_lo elements on
  GemmGroup& t = p.g[0];
  t.A = (const bf16_t*)x;
  t.W = (const bf16_t*)w;
  t.bias = (const bf16_t*)bias;                     // float32 [Cout]
  t.C = (bf16_t*)out;
  t.res = (const bf16_t*)res;
  t.M = B * Ho * Wo;
  p.ngroups = 1;
  p.nbatch = 1;
  p.N = Cout;
  p.K = ksize * ksize * Cin;
  p.lda = Cin;
  p.ldc = Cout;
  p.epi = res ? EPI_GATE_RES : EPI_BIAS;
  p.alpha = 1.f;
  p.a_lo = x_lo;
  p.w_lo = w_lo;
  p.c_lo = out_lo;
  p.res_lo = res_lo;
  const int forced = g_conv_x3_cfg;   // tuning knob
  int cfg = forced > 0 ? forced : pick_cfg(&t.M, 1, 1, Cout, p.K, true, true);
  if (forced <= 0 && ups && (cfg & 0xff) == 49) cfg = (cfg & ~0xff) | 15;
  if (gn_ws) *gn_nchunks = arm_gn_stats(p, cfg, B, Ho * Wo, 1, gn_ws, gn_ws_bytes);
  return launch(p, cfg, true, (hipStream_t)stream, true);
}

// (nearest-upsample x2 -> 3x3 conv, pad 1) as four 2x2 convs of the low-res input, one per output-pixel parity:
// the 3x3 taps that read the same source pixel are pre-summed in `w4` ([4][Cout][2][2][Cin], parity = 2*dy + dx;
// ops.subpixel_weights builds it in float32 before the hi/lo split), so the MFMA work is 4/9 of the fused-upsample
// loader's.  Replaces Upsample + Conv2d of flux/autoencoder.py:117-122 (and stable_diffusion/vae.py) exactly up to
// float32 rounding of the pre-summed weights.
extern "C" int fluxhip_conv_up2x_x3(const void* x, int64_t x_lo, const void* w4, int64_t w_lo, const void* bias,
                                    void* out, int64_t out_lo, int B, int Hs, int Ws, int Cin, int Cout,
                                    void* gn_ws, int64_t gn_ws_bytes, int* gn_nchunks, const void* zero16,
                                    void* stream) {
  if (gn_nchunks) *gn_nchunks = 0;
  if (!x || !w4 || !out || !zero16 || (gn_ws && !gn_nchunks)) return FLUXHIP_EINVAL;
  if (Cin % 64 || Cout % 4 || (x_lo | w_lo) % 8 || out_lo % 4) return FLUXHIP_EINVAL;
  static const int forced = [] { const char* e = getenv("FLUXHIP_CONV_X3_CFG"); return e ? atoi(e) : 0; }();
  GemmParams p{};
  p.cv.X = (const bf16_t*)x;
  p.cv.zero = (const bf16_t*)zero16;
  p.cv.Hs = Hs; p.cv.Ws = Ws; p.cv.Ho = Hs; p.cv.Wo = Ws;
  p.cv.Cin = Cin; p.cv.ksize = 2; p.cv.stride = 1; p.cv.pad = 0; p.cv.ups = 0;
  p.cv.sub2 = 1;
  p.cv.x_extent = x_lo >= 0 ? (x_lo + (long long)B * Hs * Ws * Cin) * 2 : 0;
  GemmGroup& t = p.g[0];
  t.A = (const bf16_t*)x;
  t.W = (const bf16_t*)w4;
  t.w_bstride = (long long)Cout * 4 * Cin;          // one 2x2 weight set per parity ("batch")
  t.bias = (const bf16_t*)bias;
  t.C = (bf16_t*)out;
  t.M = B * Hs * Ws;
  p.ngroups = 1;
  p.nbatch = 4;
  p.N = Cout;
  p.K = 4 * Cin;
  p.lda = Cin;
  p.ldc = Cout;
  p.epi = EPI_BIAS;
  p.alpha = 1.f;
  p.a_lo = x_lo;
  p.w_lo = w_lo;
  p.c_lo = out_lo;
  int cfg = forced > 0 ? forced : pick_cfg(&t.M, 1, 4, Cout, p.K, true, true);
  if (gn_ws) *gn_nchunks = arm_gn_stats(p, cfg, B, Hs * Ws, 4, gn_ws, gn_ws_bytes);
  return launch(p, cfg, true, (hipStream_t)stream, true);
}
