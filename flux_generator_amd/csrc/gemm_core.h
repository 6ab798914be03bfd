// bf16 "NT" GEMM main loop for gfx950:  C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue)
//
// This is the contraction behind every Linear on the Flux denoise path
// (reference: flux/layers.py:104-106,163-165,250-252 and flux/model.py:56,64) and,
// with the implicit-GEMM A loader, behind the VAE / UNet 3x3 convolutions
// (reference: flux/autoencoder.py:70-81,117-122).
//
// Design (MI355X-first, not a translation of anything):
//  * 64-lane waves, v_mfma_f32_16x16x32_bf16, fp32 accumulators.
//  * Both operands are K-contiguous, so A and W tiles are staged with
//    global_load_lds_dwordx4 (16 B / lane, 1 KiB / wave-instruction) straight
//    into LDS, double-buffered, one barrier per K-step of 64.
//  * LDS image is lane-linear (hardware constraint of the LDS-DMA), so the bank
//    swizzle lives on the *source* address: 16-B chunk c of tile row r is fetched
//    into physical chunk c ^ (r & 7); fragment reads apply the same XOR. With
//    128-B rows this makes every ds_read_b128 lane group hit 16 distinct slots.
//  * MFMA operands are swapped (W fragment as "A", activation fragment as "B") so
//    that each lane ends up holding 4 *consecutive output columns* of one row:
//    epilogue loads/stores are 8-byte vectors, bias/gate are read as 4 bf16.
//  * Workgroup -> tile mapping is XCD-aware: the 8 XCDs (private L2 each) get
//    contiguous ranges of N-tiles with all M-tiles of that range, so every weight
//    panel is pulled from HBM by exactly one XCD.
#pragma once
#include "common.h"

enum GemmEpi : int {
  EPI_BIAS = 0,       // C = acc + bias
  EPI_GELU_TANH = 1,  // C = gelu_tanh(acc + bias)
  EPI_GATE_RES = 2,   // C = res + gate[b,n] * (acc + bias)      (gate may be null -> 1)
  EPI_SPLIT_GELU = 3, // n <  n_split: C  = acc + bias
                      // n >= n_split: C2[:, n - n_split + c2_coloff] = gelu_tanh(acc + bias)
  EPI_SILU = 4,       // C = silu(acc + bias)
  EPI_GEGLU = 5,      // C = res * gelu_erf(acc + bias)       (UNet GEGLU: linear1(y) * gelu(linear2(y)))
  EPI_QUICK_GELU = 6, // C = v * sigmoid(1.702 v), v = acc + bias   (CLIP "quick_gelu", flux/clip.py:9)
  EPI_GELU_ERF = 7,   // C = gelu_erf(acc + bias)   (nn.gelu: OpenCLIP text towers of SD 2.1 / SDXL, stable_diffusion/.../clip.py:11)
  EPI_GEGLU_PAIR = 8, // lean kernels only: W holds [value rows | gate rows] interleaved in blocks of 16, C[m][16 f + c] = v * gelu_erf(g) with
                      // v / g = column 32 f + c / 32 f + 16 + c of (acc + bias): the two GEGLU Linears of the UNet as ONE launch, N / 2 outputs
};

struct GemmGroup {
  const bf16_t* A;     // [nbatch][M][lda]   (dense A loader)
  const bf16_t* W;     // [N][K]
  const bf16_t* bias;  // [N] or null  (row_bias: [M])
  bf16_t* C;           // [nbatch][M][ldc]
  const bf16_t* res;   // residual, same indexing as C (EPI_GATE_RES)
  const bf16_t* gate;  // [nbatch][gate_bstride] or null
  long long a_bstride; // elements between batches of A
  long long c_bstride; // elements between batches of C / res
  long long gate_bstride;
  long long w_bstride; // elements between batches of W (0: shared weight)
  int M;               // rows per batch
  int tiles_m;         // ceil(M / BM)
  const bf16_t* addm;  // optional matrix addend [nbatch][M][addvec_stride] (GemmParams::addvec is then non-null as well)
  long long addm_bstride;
};

struct ConvGeom {        // implicit-GEMM A loader (NHWC activations)
  const bf16_t* X;       // [B][Hs][Ws][Cin]
  const bf16_t* zero;    // >= 16 zero bytes (border taps read this)
  int Hs, Ws;            // stored (source) resolution
  int Ho, Wo;            // output resolution
  int Cin;               // multiple of 64
  int ksize;             // 1, 2 or 3
  int stride;            // 1 or 2
  int pad;               // 0 or 1
  int ups;               // 1: input is nearest-upsampled x2 on the fly
  // Sub-pixel form of (nearest-upsample x2 -> 3x3 conv): output pixels of parity (sdy, sdx) are a 2x2 conv of the
  // LOW-res input (the 3x3 taps that land on the same source pixel are pre-summed in the weights: 4/9 of the
  // MFMA work).  The four parities are the "batches" of one launch (nbatch = 4, w_bstride = one 2x2 weight set,
  // parity innermost in the tile order so that the four tiles sharing an input window run together): the window's
  // top-left is (y + sdy - 1, x + sdx - 1) and output row m = (img, y, x) is stored at (img, 2y + sdy, 2x + sdx) of
  // the [B][2Ho][2Wo] output.  FLAG_SPLIT kernels only.
  int sub2;
  // Round 5, ping-pong conv tiles: the activation image addressed through a BUFFER RESOURCE (buffer_load ... lds).  A piece's
  // source is then (32-bit lane offset, constant over the K loop) + (scalar offset of the step's tap / channel chunk / plane), and
  // a border tap is a lane offset that fails the resource's range check - the hardware writes zeros - instead of a 64-bit
  // address rebuilt and selected against a zero page per piece (~10 VALU per piece in the memory phase of group 0, the critical
  // path of the loop).  x_extent = bytes from X to the end of the last plane the loader may touch (hi [+ lo]); buf = 1 when the
  // launcher found x_extent + the bias below to fit 32 bits.
  long long x_extent;
  int buf;
};

DEVINL int conv_tap_row(int ksize, int tap) { return ksize == 3 ? tap / 3 : (ksize == 2 ? tap >> 1 : 0); }

struct GemmParams {
  GemmGroup g[2];
  ConvGeom cv;
  int ngroups, nbatch;
  int N, K;
  int lda, ldc;
  int epi;
  int row_bias;          // bias indexed by output row instead of column
  int n_split;
  bf16_t* C2;
  int ldc2;
  long long c2_bstride;
  int c2_coloff;
  int tiles_m_total, tiles_n;
  float alpha;           // scales acc before bias (attention logits etc.)
  int out_f32;           // EPI_BIAS only: C is float32 (logits of the single-head VAE attention)
  const bf16_t* addvec;  // optional [.., addvec_stride] vector added per group of addvec_rows output rows
  int addvec_rows;
  long long addvec_stride;
  int splits;            // split-K factor S (1 = off): S blocks share one output tile, see the split-K note below
  float* sk_part;        // [tiles][BM*BN] fp32 partial tiles (caller-provided workspace)
  int* sk_flag;          // [tiles] hand-off counters, zero between launches
  int sk_mode;           // 0: chain (block s adds the partial of block s-1);  1: reduce-scatter (ping-pong tiles only, every
                         //    split block resident at once: grid <= CUs) — see the split-K note at the hand-off
  int* sk_depart;        // sk_mode 1: [tiles] departure counters (zero between launches; the last block to leave resets both)
  int sk_timeout;        // sk_mode 1: how long (100 MHz ticks) a block polls for its peers before it orphans its slice and exits
  int wide_epi;          // outputs / residual / gate are 16-byte addressable: LDS-transposed epilogue
  unsigned long long* trace;  // FLAG_TIMED kernels only: [nblk][nwaves][8] summed segment cycles
  // FLAG_SPLIT kernels only ("bf16x3": fp32-faithful products on the bf16 matrix cores).  Every operand is a pair
  // of bf16 planes (hi = bf16(x), lo = bf16(x - hi)); the lo plane sits `*_lo` ELEMENTS after the hi pointer.
  // bias is float32.  See the FLAG_SPLIT note at the kernel.
  long long a_lo, w_lo, c_lo, res_lo;
  // FLAG_SPLIT convs only: GroupNorm statistics of the OUTPUT produced by the epilogue (the consumer of every VAE
  // conv is a GroupNorm): per wave row-block and channel quad, (sum, sum of squares) of the float32 results, written
  // in the per-channel partial layout gn_finalize_kernel reads ([image][chunk][C][2]) on the quad's FIRST channel only
  // (the other three slots are never written nor read: gn_finalize_kernel's channel step is 4 for these tables).  gn_hw = output pixels per image of the GEMM's M space (a multiple of BM, so
  // a tile never straddles two images); chunk = parity * gn_hw / WTM + row-block (parity: sub-pixel convs).
  float* gn_ws;
  int gn_hw, gn_nchunks;
  // FLAG_FP8 kernels only: per-row dequantisation scales of the fp8 (e4m3fn) operands, per group:
  // C = (acc * a_scale[m] * w_scale[n]) + bias.  a_scale is indexed like the rows of A ([nbatch][M], batch stride
  // a_sc_bstride), w_scale like the rows of W ([N]).
  const float* a_scale[2];
  const float* w_scale[2];
  long long a_sc_bstride;
  // FLAG_MXA kernels (fp8 with a block-scaled activation operand): E8M0 scale bytes of A, one per (row, 32 consecutive K
  // elements), tiled so that lane (r16, q4) of a wave reads the bytes of its four 16-row fragments for one K-step as ONE
  // dword (the block-scaled MFMA takes its scale from a byte of a per-lane register, op_sel picks the byte):
  //     dword  ks * a_mx_kstride + (row >> 6) * 64 + q4 * 16 + (row & 15),   byte (row >> 4) & 3,
  // q4 = K block inside the 128-element step ks, row = a_mx_row0[group] + batch * a_mx_bstride + m counted over the whole
  // scale buffer (every term a multiple of 64).  a_scale is not read: the epilogue applies w_scale[n] only.
  const uint32_t* a_mx;
  int a_mx_row0[2];
  long long a_mx_bstride, a_mx_kstride;
  // FLAG_MXC kernels: the GELU'd output (EPI_GELU_TANH: every column; EPI_SPLIT_GELU: the columns from n_split on, at
  // column n - n_split + c8_coloff) leaves as e4m3 bytes [M][ldc8] + block scales in the tiling above - the next GEMM's A.
  uint8_t* c8[2];
  long long c8_bstride;
  int ldc8, c8_coloff;
  uint8_t* c_mx;
  int c_mx_row0[2];
  long long c_mx_bstride, c_mx_kstride;
};

enum GemmFlags : int {
  FLAG_TIMED = 1,     // s_memtime stamps around the phases of the pipelined loop (diagnostic tile configs only)
  // fp32-faithful mode for the VAE decoders, which the reference runs in fp32 (flux/utils.py:137-143 keeps the
  // checkpoint dtype; stable_diffusion/__init__.py:25 load_autoencoder(model, False)).  An fp32 value x is held as
  // two bf16 planes x = hi + lo (+ O(2^-17 |x|)); a product a*w is evaluated as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi
  // (the dropped a_lo*w_lo is O(2^-16) relative), all three on v_mfma_f32_16x16x32_bf16 into the SAME fp32
  // accumulators: 3x the MFMA work at 16x the fp32-MFMA rate.  The main loop is unchanged: its K-step index g
  // runs over 3 * K/64 steps, g -> (operand step g / 3, pass g % 3), and the pass only selects which plane the
  // staging addresses point at (the hi tiles are fetched twice back to back: the second fetch hits L2).
  FLAG_SPLIT = 2,
  // fp8 operands on the block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, the only fp8 MFMA
  // that runs at twice the bf16 rate on gfx950; unit E8M0 block scales).  A and W are OCP e4m3fn bytes with one
  // float32 scale per row (per token / per output channel), applied in the epilogue.  A K-step is still 128 BYTES
  // per row (128 fp8 elements instead of 64 bf16), so the LDS image, the LDS-DMA staging, the swizzle and the whole
  // barrier / vmcnt schedule are those of the bf16 kernel; only the fragment chunk pairing differs: a lane's 32
  // consecutive K bytes are the two 16-byte chunks 2q and 2q+1 of its row (bf16: chunk q for MFMA 0, 4+q for MFMA 1),
  // and the pair feeds ONE 16x16x128 MFMA where the bf16 loop issues two 16x16x32.  Same cycles per K-step, 2x K.
  FLAG_FP8 = 4,
  // split-K reduce-scatter hand-off (see the hand-off after the main loop): its own instantiations.  The ownership-aware
  // epilogue (runtime trip counts, owned-fragment tests) and the brick block map cost the launches that do NOT split 2 % when
  // they share the kernel (build-to-build A/B, tools/lib_ab.py: mlp0 87.1 -> 88.7 us, linear1 150.0 -> 153.0 us), so only the
  // tiles the picker splits (256 x 192, 256 x 256) carry a second kernel with it.
  FLAG_RS = 16,
  // float16 storage (the stable_diffusion/ path with float16=True, the reference's flux_app.py:77-79): operands, bias,
  // residual, gate, addvec and outputs are IEEE half; v_mfma_f32_16x16x32_f16 issues at the bf16 rate; accumulation stays fp32
  FLAG_F16 = 64,
  // fp8 with MX-style block scales (OCP e4m3 elements, one E8M0 scale per 32 consecutive K elements of a row), so that a
  // producer's epilogue can quantise its own output tile (a per-token scale needs the whole row).  FLAG_MXA: the activation
  // operand carries block scales, fed to the MFMA's scale operand (GemmParams::a_mx); FLAG_MXC: the epilogue writes e4m3 +
  // block scales instead of bf16 (GemmParams::c8 / c_mx).  Ping-pong fp8 lean kernels only.  (tools/probes/mfma_mx_probe*.hip:
  // the scale byte of lane (r, kb) scales k in [32 kb, 32 kb + 32) of row r, and the instruction's own K order is the one
  // the fp8 loop already uses - lane group kb supplies the 16-byte chunks kb and 4 + kb - so memory K order = MFMA K order.)
  FLAG_MXA = 8,
  FLAG_MXC = 128,
  // fp32-faithful 3 x 3 / stride 1 / pad 1 convs whose tile is 256 consecutive pixels of ONE image row (Ws % 256 == 0), 256 x 128
  // ping-pong tile: the three horizontal taps of a filter row read the SAME activation rows shifted by one pixel, so the LDS slot
  // holds the row segment once with a one-pixel halo (258 rows of 128 B) and the taps are fragment-read offsets - a third of the
  // activation LDS-DMA of the tap-by-tap loader.  See the DXR block in the ping-pong loop.
  FLAG_DXR = 4096,
  FLAG_LEAN = 32,    // dense bf16 only: the epilogues, operands and hand-offs the Flux / transformer-block launches use, nothing else compiled in (see gemm.hip lean_ok)
};

// one 16x16x32 MFMA on 16-bit operands of the kernel's storage type (bf16, or float16 under FLAG_F16)
template <bool H>
DEVINL f32x4 mfma16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int N>
DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NSTAGE-deep LDS ring: the loads of K-step kt+NSTAGE-1 are issued right after the barrier that
// opens step kt, so they have NSTAGE-1 compute phases to land; waits are COUNTED (vmcnt(N), never a
// drain in steady state) and the barrier is a raw s_barrier, because __syncthreads() would drain
// the in-flight LDS-DMA (cdna guide §5, "Pipelining across barriers").
template <int BM, int BN, int WM, int WN, int AMODE, int NSTAGE, int PIPE, int FLAGS = 0>
// 8-wave tiles with more than half of the CU's LDS own the CU: two waves per SIMD is all they can ever have, so the
// register allocator is told not to squeeze below 256 VGPRs for a third wave that can never be resident (it spilled
// 9-25 dwords into scratch doing so on the conv variants).
#define GEMM_OWNS_CU (WM * WN == 8 && (NSTAGE * BM + (PIPE >= 3 ? NSTAGE + 1 : NSTAGE) * BN) * 128 > 80 * 1024)
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(GEMM_OWNS_CU ? 2 : 1, GEMM_OWNS_CU ? 2 : 8)))
void gemm_nt_kernel(const GemmParams p) {
#undef GEMM_OWNS_CU
  constexpr int BK = 64;
  constexpr int NWAVES = WM * WN;
  constexpr bool PP = PIPE == 6;                   // ping-pong schedule
  constexpr int PIPEX = PIPE;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  constexpr int APW = BM / 8 / NWAVES;
  // B pieces need not divide evenly over the waves (BN = 192, 224): the surplus slots re-fetch the
  // last piece (same bytes to the same LDS address), which keeps the per-wave vmcnt arithmetic uniform.
  constexpr int BPIECES = BN / 8, BPW = (BPIECES + NWAVES - 1) / NWAVES;
  static_assert(APW >= 1 && BM % (8 * NWAVES) == 0 && BN % 16 == 0, "tile / wave-count mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long t_start = 0, rt_start = 0, t_loop0 = 0, t_loop1 = 0, t_epiA = 0, t_epiB = 0;
  if constexpr ((FLAGS & FLAG_TIMED) != 0) {
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t_start), "=s"(rt_start)::"memory");
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware, bijective block -> tile map -------------------------------
  // Split-K (p.splits = S > 1): S consecutive-in-dispatch blocks of the same XCD own one output tile and
  // one K range each.  Block s adds the partial tile of block s-1 (fp32, through L2) to its accumulators
  // and either hands the sum on (s < S-1) or runs the epilogue (s = S-1): a fixed summation order, so the
  // result is deterministic.  Block s only ever waits for a block with a LOWER block id, which the
  // dispatcher has already started, so the chain cannot deadlock even when the grid exceeds the chip.
  // FLAG_LEAN: a non-RS lean kernel is never launched with a split (the chain code below folds away); a lean RS kernel is
  // only ever launched in reduce-scatter mode
  constexpr bool LEAN = (FLAGS & FLAG_LEAN) != 0;
  const int S = (LEAN && (FLAGS & FLAG_RS) == 0) ? 1 : p.splits;
  int nblk = gridDim.x, bid = blockIdx.x, sidx = 0;
  // Reduce-scatter split-K (sk_mode 1, see the hand-off after the main loop) exchanges its partials through memory, so the
  // S blocks of a tile need not share an XCD — and should not: with K = 12288 / 15360 the main loop of the N = 3072
  // projections is FABRIC-bound (phase trace: 1.0 us per K-step against 0.8 us of MFMA issue), because under the tile-major
  // map below every XCD walks all M-tiles over all K, i.e. the 39 MB activation panel crosses the fabric 8 times (~410 MB per
  // launch for 134 MB of operands).  Here the (split, tile) space is cut into one contiguous brick per XCD in
  // (split-major, N-tile, M-tile) order — 240 blocks: every XCD owns ONE K range, 6 N-tiles and all 5 M-tiles — so an XCD
  // reads a third of the activation panel (once per brick row) and its own weight panels: ~215 MB per launch.
  bool rs_map = false;
  if constexpr ((FLAGS & FLAG_RS) != 0) rs_map = LEAN || (S > 1 && p.sk_mode != 0);
  int swz;
  if (rs_map) {
    const int T = nblk / S;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // position in XCD-brick order
    sidx = lin / T;
    bid = lin - sidx * T;          // tile id (flags, slab)
    swz = bid;
  } else {
  if (S > 1) {
    const int T = nblk / S, full = T >> 3, grp = 8 * S;
    if (bid < full * grp) {
      const int g = bid / grp, r = bid - g * grp;
      sidx = r >> 3;
      bid = g * 8 + (r & 7);
    } else {
      const int r = bid - full * grp, rem = T - full * 8;
      sidx = r / rem;
      bid = full * 8 + (r - sidx * rem);
    }
    nblk = T;
  }
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int TM = p.tiles_m_total;
  int tm = swz % TM;
  int tn = swz / TM;
  // Many M-tiles (Flux-dev 1024^2: 18, the fp8 configuration at batch 4: 68): bands of GN N-tiles, inside a band the N-tile runs
  // fastest - the ~32 tiles an XCD has in flight are then ~8 M-tiles x 4 N-tiles (8 activation + 4 weight panels in its L2)
  // instead of 32 M-tiles of ONE N-tile (32 + 1 panels): the activation operand, which no longer fits any cache at these
  // sizes, is read once per band instead of once per N-tile.  (At batch 1 - 5 M-tiles - the order below is unchanged.)
#ifndef FLUXHIP_TILE_BAND
#define FLUXHIP_TILE_BAND 4
#endif
  if (FLUXHIP_TILE_BAND > 1 && TM >= 16 && !rs_map) {      // (convs: the N-tiles of one pixel window share its im2col reads)
    // (long K - the N = 3072 projections back into the residual stream - measured 3 % better with bands of 8: their weight panel
    //  per N-tile is the larger operand stream)
    const int GN = p.K >= 8192 ? 2 * FLUXHIP_TILE_BAND : FLUXHIP_TILE_BAND;
    const int band = swz / (TM * GN), first = band * GN;
    const int gn = min(GN, p.tiles_n - first);
    const int r = swz - band * (TM * GN);
    tm = r / gn;
    tn = first + (r - tm * gn);
  }

  const int tm0 = p.nbatch * p.g[0].tiles_m;
  const bool g1 = tm >= tm0;
  if (g1) tm -= tm0;
  const bf16_t* gA = g1 ? p.g[1].A : p.g[0].A;
  const bf16_t* gW = g1 ? p.g[1].W : p.g[0].W;
  const bf16_t* gBias = g1 ? p.g[1].bias : p.g[0].bias;
  bf16_t* gC = g1 ? p.g[1].C : p.g[0].C;
  const bf16_t* gRes = g1 ? p.g[1].res : p.g[0].res;
  const bf16_t* gGate = g1 ? p.g[1].gate : p.g[0].gate;
  const bf16_t* gAddm = g1 ? p.g[1].addm : p.g[0].addm;
  const long long addm_bs = g1 ? p.g[1].addm_bstride : p.g[0].addm_bstride;
  const long long a_bs = g1 ? p.g[1].a_bstride : p.g[0].a_bstride;
  const long long c_bs = g1 ? p.g[1].c_bstride : p.g[0].c_bstride;
  const long long gate_bs = g1 ? p.g[1].gate_bstride : p.g[0].gate_bstride;
  const long long w_bs = g1 ? p.g[1].w_bstride : p.g[0].w_bstride;
  const int Mg = g1 ? p.g[1].M : p.g[0].M;
  const int tpb = g1 ? p.g[1].tiles_m : p.g[0].tiles_m;
  int b = tm / tpb;
  int m0 = (tm % tpb) * BM;
  int sdy = 0, sdx = 0, c_oy = 0, c_ox = 0;         // sub-pixel conv: parity of this tile's output pixels
  if (AMODE == 1 && p.cv.sub2) {
    b = tm & 3;
    m0 = (tm >> 2) * BM;
    sdy = b >> 1; sdx = b & 1;
    c_oy = sdy - 1; c_ox = sdx - 1;
  }
  const int n0 = tn * BN;
  const int N = p.N, K = p.K;
  constexpr bool X3 = (FLAGS & FLAG_SPLIT) != 0;
  constexpr bool F8 = (FLAGS & FLAG_FP8) != 0;
  constexpr bool F16 = (FLAGS & FLAG_F16) != 0;    // 16-bit storage type: false = bfloat16, true = float16
  static_assert(!(F16 && (X3 || F8 || (FLAGS & FLAG_RS) != 0)), "float16 storage: plain dense / conv kernels");
  constexpr int ESZ = F8 ? 1 : 2;                  // bytes per operand element
  static_assert(!(F8 && (X3 || AMODE != 0)), "fp8: dense operands, no split mode");
  static_assert(!F8 || PIPE == 0 || PIPE == 6, "fp8 variants exist for the simple ring and the ping-pong schedule");
  constexpr bool MXA = (FLAGS & FLAG_MXA) != 0, MXC = (FLAGS & FLAG_MXC) != 0;
  static_assert(!(MXA || MXC) || (F8 && PIPE == 6 && (FLAGS & FLAG_LEAN) != 0), "block-scaled fp8: lean ping-pong kernels");
  const int nkt_all = X3 ? 3 * (K / BK) : K / (BK * 2 / ESZ);
  // K range of this block (even split; skewing the ranges so that the producer finishes early did not
  // pay: the release fence of the early block slows the L2 for the blocks still in their main loop)
  const int kbase = (int)((long long)sidx * nkt_all / S);
  const int nkt = (int)((long long)(sidx + 1) * nkt_all / S) - kbase;

  // logical K-step kt of this block -> operand K-step; FLAG_SPLIT: also the byte offsets that move the activation /
  // weight source to its lo plane for this pass (pass 0: hi*hi, 1: hi*lo, 2: lo*hi).  All wave-uniform.
  auto kstep = [&](int kt, long long& a_adj, long long& w_adj) {
    const int g = kt + kbase;
    if constexpr (X3) {
      const int kk = g / 3, pass = g - kk * 3;
      a_adj = pass == 2 ? p.a_lo * 2 : 0;
      w_adj = pass == 1 ? p.w_lo * 2 : 0;
      return kk;
    } else {
      a_adj = 0;
      w_adj = 0;
      return g;
    }
  };

  // ---- per-lane staging sources ----------------------------------------------
  const int lr = lane >> 3;             // row inside an 8-row piece
  const int lc = (lane & 7) ^ lr;       // logical 16-B chunk fetched by this lane
  const char* asrc[APW];
  int arow[APW];                        // conv: packed output-pixel coords
  const char* bsrc[BPW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    int row = (wave + i * NWAVES) * 8 + lr;
    int grow = min(m0 + row, Mg - 1);
    if (AMODE == 0) {
      asrc[i] = (const char*)gA + ((long long)b * a_bs + (long long)grow * p.lda) * ESZ + lc * 16;
      arow[i] = 0;
    } else {
      asrc[i] = nullptr;
      arow[i] = grow;
    }
  }
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    int row = min(wave + i * NWAVES, BPIECES - 1) * 8 + lr;
    int n = min(n0 + row, N - 1);
    bsrc[i] = (const char*)gW + ((long long)b * w_bs + (long long)n * K) * ESZ + lc * 16;
  }

  // conv geometry decode (per staged row), hoisted out of the K loop, 2 registers per piece (the 256-wide
  // tiles have no VGPRs to spare):
  //  * plain convs: cimg = signed offset, in 16-byte units, of the (possibly out-of-image) top-left tap of the
  //    row's window; cyx = 9-bit mask of the taps that fall inside the image.  A K-step then adds one
  //    wave-uniform (tap, channel chunk) byte offset - no per-lane coordinate arithmetic in the main loop.
  //  * fused nearest-upsample convs: cimg = image offset (16-byte units), cyx = packed top-left tap coords;
  //    the source pixel is ((y+dy)>>1, (x+dx)>>1), decoded per K-step.
  int cyx[APW];
  uint32_t cimg[APW];
  const char* const cX = (const char*)p.cv.X + lc * 16;
  if (AMODE == 1) {
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      int pix = arow[i];
      int hw = p.cv.Ho * p.cv.Wo;
      int bb = pix / hw;
      int rem = pix - bb * hw;
      int y = rem / p.cv.Wo;
      int x = rem - y * p.cv.Wo;
      const int y0 = y * p.cv.stride - p.cv.pad + c_oy, x0 = x * p.cv.stride - p.cv.pad + c_ox;
      const long long img = (long long)bb * p.cv.Hs * p.cv.Ws * p.cv.Cin;      // elements; Cin % 64 == 0
      if (p.cv.ups) {
        cyx[i] = (y0 << 16) | (x0 & 0xffff);
        cimg[i] = (uint32_t)(img >> 3);
      } else {
        int mask = 0;
        for (int t = 0; t < p.cv.ksize * p.cv.ksize; ++t) {
          const int ty = conv_tap_row(p.cv.ksize, t);
          const int yy = y0 + ty, xx = x0 + t - ty * p.cv.ksize;
          mask |= ((yy >= 0) & (yy < p.cv.Hs) & (xx >= 0) & (xx < p.cv.Ws)) << t;
        }
        cyx[i] = mask;
        cimg[i] = (uint32_t)(int)((img + ((long long)y0 * p.cv.Ws + x0) * p.cv.Cin) >> 3);
      }
    }
  }

  // LDS layout: NSA activation slots of A_BYTES, then NSW weight slots of B_BYTES.  The weight ring may be
  // one slot deeper than the activation ring (PIPE 3): weights are the cold HBM stream (every line is a
  // compulsory miss for the XCD), activations are re-read by every N-tile and mostly hit L2, and 2 x 32 KiB
  // + 3 x 32 KiB is exactly the 160 KiB of a CU for the 256 x 256 tile.
  constexpr int NSA = NSTAGE, NSW = (PIPEX >= 3) ? NSTAGE + 1 : NSTAGE;
  constexpr int W_BASE = NSA * A_BYTES;
  // [i0, i1) = the subset of this wave's pieces to issue (PIPE 4 spreads them between MFMAs)
  auto stage_a = [&](int kt, int slot, int i0 = 0, int i1 = 64) {
    char* sa = smem + slot * A_BYTES;
    long long a_adj, w_adj_unused;
    const int ks = kstep(kt, a_adj, w_adj_unused);
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < APW; ++i)
        if (i >= i0 && i < i1) glds16(asrc[i] + (long long)ks * (BK * 2) + a_adj, sa + (wave + i * NWAVES) * 1024);
    } else {
      // K order = channel chunk outer, filter tap inner: the 9 taps of one 64-channel chunk are
      // staged back to back, so their overlapping input rows are still in L1/L2 (a tap-major order
      // re-streams the whole input tile 9 times through the XCD's L2).
      const int ntap = p.cv.ksize * p.cv.ksize;
      const int cch = ks / ntap;
      const int tap = ks - cch * ntap;
      const int c0 = cch << 6;
      const int dy = conv_tap_row(p.cv.ksize, tap);
      const int dx = tap - dy * p.cv.ksize;
      if (!p.cv.ups) {
        const long long uoff = ((long long)(dy * p.cv.Ws + dx) * p.cv.Cin + c0) * 2 + a_adj;   // wave-uniform
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          if (i < i0 || i >= i1) continue;
          uint32_t img = cimg[i];
          asm volatile("" : "+v"(img));      // keep the 64-bit base out of loop-invariant hoisting (VGPR budget)
          const char* src = ((cyx[i] >> tap) & 1) ? cX + (long long)(int)img * 16 + uoff : (const char*)p.cv.zero;
          glds16(src, sa + (wave + i * NWAVES) * 1024);
        }
        return;
      }
      const int Hl = p.cv.Hs * 2, Wl = p.cv.Ws * 2;      // logical input grid of the fused upsample
#pragma unroll
      for (int i = 0; i < APW; ++i) {
        if (i < i0 || i >= i1) continue;
        int yy = (cyx[i] >> 16) + dy, xx = (int)(short)(cyx[i] & 0xffff) + dx;
        bool ok = (yy >= 0) & (yy < Hl) & (xx >= 0) & (xx < Wl);
        uint32_t img = cimg[i];
        asm volatile("" : "+v"(img));
        const char* src = ok ? cX + (long long)img * 16 + ((long long)((yy >> 1) * p.cv.Ws + (xx >> 1)) * p.cv.Cin + c0) * 2 + a_adj
                             : (const char*)p.cv.zero;
        glds16(src, sa + (wave + i * NWAVES) * 1024);
      }
    }
  };
  auto stage_w = [&](int kt, int slot, int i0 = 0, int i1 = 64) {
    char* sb = smem + W_BASE + slot * B_BYTES;
    long long a_adj_unused, w_adj;
    const int ks = kstep(kt, a_adj_unused, w_adj);
    int koff = ks * BK;                      // K offset (elements) of this step inside a W row
    if (AMODE == 1) {                        // conv: weight column = tap*Cin + c0 (pure index remap)
      const int ntap = p.cv.ksize * p.cv.ksize;
      const int cch = ks / ntap;
      koff = (ks - cch * ntap) * p.cv.Cin + (cch << 6);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i)
      if (i >= i0 && i < i1) glds16(bsrc[i] + (long long)koff * 2 + w_adj, sb + min(wave + i * NWAVES, BPIECES - 1) * 1024);
  };

  // ---- fragment read offsets (same XOR as the staging source swizzle) ---------
  const int r16 = lane & 15, q4 = lane >> 4;
  int foff[2];
  // FLAG_FP8: the lane's 32-byte operand of the 16x16x128 MFMA is chunks q4 and 4 + q4 of its row as well - the same two
  // reads as bf16 - not the contiguous pair 2 q4, 2 q4 + 1: the K order inside a step is free as long as both operands use
  // the same one, and with chunk 2 q4 the ds_read_b128 lane group {0-3, 12-15, 20-27} put two lanes on every 16-byte slot
  // (rocprofv3: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE in the fp8 kernels against 2 % in the bf16 ones).
  foff[0] = r16 * 128 + ((q4 ^ (lane & 7)) << 4);
  foff[1] = r16 * 128 + (((4 + q4) ^ (lane & 7)) << 4);

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int G = APW + BPW;                       // LDS-DMA instructions per wave per K-step
  static_assert((NSTAGE - 1) * G + BPW <= 63, "vmcnt field");

  auto mma = [&](const bf16x8(&af)[MI], const bf16x8(&wf)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = mfma16<F16>(wf[j], af[i], acc[i][j]);
  };

  // FLAG_FP8: the two 16-byte halves of every fragment make one 32-byte operand of the 16x16x128 fp8 MFMA
  auto mma8 = [&](const bf16x8(&af0)[MI], const bf16x8(&af1)[MI], const bf16x8(&wf0)[NJ], const bf16x8(&wf1)[NJ]) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    i32x8 av[MI], wv[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      av[i] = __builtin_shufflevector(__builtin_bit_cast(i32x4, af0[i]), __builtin_bit_cast(i32x4, af1[i]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      wv[j] = __builtin_shufflevector(__builtin_bit_cast(i32x4, wf0[j]), __builtin_bit_cast(i32x4, wf1[j]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv[j], av[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                     0x7f7f7f7f);   // e4m3 x e4m3, unit block scales
  };
  auto mma_step = [&](const bf16x8(&af0)[MI], const bf16x8(&wf0)[NJ], const bf16x8(&af1)[MI], const bf16x8(&wf1)[NJ]) {
    if constexpr (F8) {
      mma8(af0, af1, wf0, wf1);
    } else {
      mma(af0, wf0);
      mma(af1, wf1);
    }
  };

  if constexpr (PIPEX == 0) {
    // simple ring: wait -> barrier -> refill the freed slot -> read fragments -> MFMA
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nkt) { stage_a(s, s); stage_w(s, s); }
    int cur = 0, nxt = NSTAGE - 1;                   // ring slots of step kt and step kt+NSTAGE-1
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + NSTAGE - 2 < nkt) wait_vmcnt<(NSTAGE - 2) * G>();
      else wait_vmcnt<0>();                          // tail: fewer steps in flight than the ring holds
      __builtin_amdgcn_s_barrier();                  // step kt landed for every wave; slot nxt is free
      if (kt + NSTAGE - 1 < nkt) { stage_a(kt + NSTAGE - 1, nxt); stage_w(kt + NSTAGE - 1, nxt); }
      const char* sa = smem + cur * A_BYTES + (wm * WTM) * 128;
      const char* sb = smem + W_BASE + cur * B_BYTES + (wn * WTN) * 128;
      if constexpr (F8) {
        bf16x8 af0[MI], wf0[NJ], af1[MI], wf1[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          af0[i] = *(const bf16x8*)(sa + i * 2048 + foff[0]);
          af1[i] = *(const bf16x8*)(sa + i * 2048 + foff[1]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          wf0[j] = *(const bf16x8*)(sb + j * 2048 + foff[0]);
          wf1[j] = *(const bf16x8*)(sb + j * 2048 + foff[1]);
        }
        mma8(af0, af1, wf0, wf1);
      } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + i * 2048 + foff[kk]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *(const bf16x8*)(sb + j * 2048 + foff[kk]);
        mma(af, wf);
      }
      }
      cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
      nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    }
  } else {
    // fragment-double-buffered ring: every MFMA cluster runs with the NEXT cluster's ds_reads in
    // flight; the barrier (and the wait for the next K-step's LDS-DMA) sits between the two
    // clusters of a step, so LDS latency and barrier skew hide under 32..64 MFMAs.
    // The fragment reads are inline asm with hand-counted lgkmcnt: hipcc's own bookkeeping drains
    // lgkmcnt(0) at the loop head, which serialises [reads -> wait -> MFMAs] inside each wave.
    // lgkmcnt is a 4-bit field: with 16 fragment reads per cluster the wait below asks for <= 15
    // outstanding, i.e. it also waits for the first read of the NEXT cluster (reads retire in order).
    constexpr int NF = (MI + NJ) < 15 ? (MI + NJ) : 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_rd = lds0 + (wm * WTM) * 128;             // + slotA*A_BYTES + foff[kk]
    const uint32_t b_rd = lds0 + W_BASE + (wn * WTN) * 128;    // + slotW*B_BYTES + foff[kk]
    auto read_frags = [&](bf16x8(&af)[MI], bf16x8(&wf)[NJ], int slot_a, int slot_w, int kk) {
      const uint32_t aa = a_rd + slot_a * A_BYTES + foff[kk];
      const uint32_t bb = b_rd + slot_w * B_BYTES + foff[kk];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(aa), "n"(i * 2048) : "memory");
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j]) : "v"(bb), "n"(j * 2048) : "memory");
    };
    // LDS-DMA issue order (it fixes the vmcnt arithmetic): prologue A0 W0 A1 W1 .. then the extra weight
    // steps W(NSA)..W(NSW-1); iteration j issues A(j+NSA) then W(j+NSW).  When step kt+1 is needed, the
    // loads issued after A(kt+1) are W(kt+1-NSA+NSW) plus NSA-2 whole iterations:
    constexpr bool TIMED = (FLAGS & FLAG_TIMED) != 0;
    constexpr int PENDING = BPW * (NSW - NSA) + (NSA - 2) * G;   // == (NSTAGE-2)*G for a uniform ring
    // phase stamps (FLAG_TIMED, PIPE 5 only): taken where the loop drains lgkmcnt anyway, so they add no wait
    unsigned long long t_prev = 0;
    uint32_t tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (TIMED) {
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_loop0)::"memory");
      t_prev = t_loop0;
    }
#define GEMM_STAMP(i)                                                                         \
    if constexpr (TIMED) {                                                                    \
      unsigned long long t_now;                                                               \
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory");            \
      tsum[i] += (uint32_t)(t_now - t_prev);                                                  \
      t_prev = t_now;                                                                         \
    }
    if constexpr (PP) {
// Ping-pong schedule (PIPE 6).  The 8 waves form two groups, one wave of each per SIMD (waves w and w+4
      // share a SIMD).  A K-step has two phases separated by block barriers; in each phase one group issues its 64
      // MFMAs back to back while the other does its memory work (24 fragment reads for its next MFMA phase plus
      // its share of the LDS-DMA: group 0 stages the activation tiles, group 1 the weight tiles), then they swap:
      //      group 0:  MFMA(kt) | B1 | read(kt+1), A(kt+2) -> slot kt%2      | B2
      //      group 1:  read(kt), W(kt+2) -> slot (kt+2)%3, wait W(kt+1) | B1 | MFMA(kt) | B2
      // so a SIMD always has exactly one wave feeding the matrix pipe, and that wave has nothing else to issue.
      // (The spread loops above let both waves of a SIMD run the same mix; the phase trace showed one of them
      //  winning the pipe, finishing early and idling at the barrier while the other ran alone at ~2/3 rate.)
      // Hazards: a slot is refilled one barrier after its last reader drained lgkmcnt; a group waits (vmcnt) for
      // its own pieces before the barrier that publishes them.  See DESIGN.md 3.1.
      static_assert(NSA == 2 && NSW == 3 && NWAVES == 8, "ping-pong: 8 waves, 2 + 3 ring");
      constexpr int LW = NWAVES / 2;
      constexpr int PA = BM / 8 / LW, PB = (BPIECES + LW - 1) / LW;
      static_assert(BM % (8 * LW) == 0, "activation pieces per loader wave");
      const bool g0 = wave < LW;                                   // wave-uniform
      const int lw = g0 ? wave : wave - LW;
      bf16x8 a0[MI], w0[NJ], a1[MI], w1[NJ];
      // FLAG_FP8: a fragment is born as the 8-register operand of the 16x16x128 MFMA (its two 16-byte halves are
      // joined where they are read, while both are short-lived temporaries), not as two 4-register values joined at
      // the MFMA: hipcc does not coalesce loop-carried 4-register values into a tuple and kept a second copy of every
      // fragment (+80 VGPRs on the 256 x 192 tile: scratch, and a spilled ds_read destination is copied before its data lands).
      typedef __attribute__((ext_vector_type(4))) int pp_i32x4;
      typedef __attribute__((ext_vector_type(8))) int pp_i32x8;
      pp_i32x8 A8[F8 ? MI : 1], W8[F8 ? NJ : 1];
      auto read_both = [&](int slot_a, int slot_w) {
        if constexpr (F8) {
          const uint32_t aa0 = a_rd + slot_a * A_BYTES + foff[0], aa1 = a_rd + slot_a * A_BYTES + foff[1];
          const uint32_t bb0 = b_rd + slot_w * B_BYTES + foff[0], bb1 = b_rd + slot_w * B_BYTES + foff[1];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            pp_i32x4 lo, hi;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lo) : "v"(aa0), "n"(i * 2048) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hi) : "v"(aa1), "n"(i * 2048) : "memory");
            A8[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            pp_i32x4 lo, hi;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lo) : "v"(bb0), "n"(j * 2048) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hi) : "v"(bb1), "n"(j * 2048) : "memory");
            W8[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        } else {
          read_frags(a0, w0, slot_a, slot_w, 0);
          read_frags(a1, w1, slot_a, slot_w, 1);
        }
      };
      // Interleaved memory phase (round 5, dense bf16 / f16 operands): the phase trace of the loop (tools/gemm_phase_trace2.py) shows
      // the memory phase of group 0 - 20 fragment reads, then 8 LDS-DMA pieces - taking ~960 cycles against the ~800 of the MFMA
      // phase it runs beside: the four waves of a group leave the barrier together, so all of them queue at the LDS for their
      // reads and THEN all of them queue at the CU's one texture-address path for their pieces (16 cycles per piece x 4 waves).
      // Issued alternately - a few reads, one piece - the LDS serves one wave's reads while another wave's piece holds the
      // address path.  read_flat(r): fragment read number r of the 2 (MI + NJ) of a K-step, compile-time indexed.
      // Measured (same box, interleaved runs, profiles/r05_pp_interleave_ab.json): group 0's phase 1148 -> 1134 cycles, group 1's
      // 864 -> 826, denoise step 17.43 -> 17.29 ms - the reads and the pieces mostly serialise on the LDS port anyway (a K-step
      // of this tile is 56 pieces x 16 + 160 reads x 4 = 1536 LDS cycles, as many as its MFMAs take), so the gain is small.
      constexpr bool ILV = !X3 && !F8 && AMODE == 0;
      constexpr int NRD2 = 2 * (MI + NJ);
      auto read_flat = [&](auto R, int slot_a, int slot_w, bf16x8(&a0)[MI], bf16x8(&w0)[NJ], bf16x8(&a1)[MI], bf16x8(&w1)[NJ]) {
        constexpr int r = decltype(R)::value, kk = r / (MI + NJ), q = r % (MI + NJ);
        const uint32_t aa = a_rd + slot_a * A_BYTES + foff[kk], bb = b_rd + slot_w * B_BYTES + foff[kk];
        if constexpr (kk == 0 && q < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a0[q]) : "v"(aa), "n"(q * 2048) : "memory");
        else if constexpr (kk == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w0[q - MI]) : "v"(bb), "n"((q - MI) * 2048) : "memory");
        else if constexpr (q < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a1[q]) : "v"(aa), "n"(q * 2048) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w1[q - MI]) : "v"(bb), "n"((q - MI) * 2048) : "memory");
      };
      // FLAG_MXA: the block scales of this wave's 64 rows for one K-step are one dword per lane (byte i = fragment i), fetched
      // a step ahead by an untracked (inline-asm) global load that the loops' own vmcnt waits cover: issued BEFORE the LDS-DMA
      // pieces of the same phase, so every counted wait that lands those pieces lands it too.  sc_cur feeds the MFMAs of the
      // current step, sc_nxt is in flight / landed for the next one.
      static_assert(!MXA || MI == 4, "block-scaled A: 64-row wave tiles (one scale byte per fragment in a dword)");
      uint32_t sc_cur = 0x7f7f7f7fu, sc_nxt = 0x7f7f7f7fu, mx_off = 0;
      if constexpr (MXA) {
        const int rowb = (g1 ? p.a_mx_row0[1] : p.a_mx_row0[0]) + b * (int)p.a_mx_bstride + min(m0 + wm * WTM, Mg - WTM);
        mx_off = (uint32_t)(((long long)kbase * p.a_mx_kstride + rowb + lane) * 4);     // < 4 GiB: checked by the launcher
      }
      const uint32_t mx_step = MXA ? (uint32_t)(p.a_mx_kstride * 4) : 0u;
      auto mx_load = [&](uint32_t& dst) {        // scales of the next K-step not yet requested
        asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(mx_off), "s"(p.a_mx) : "memory");
        mx_off += mx_step;
      };
      auto mma_both = [&]() {
        if constexpr (F8 && MXA) {
          static_for<0, MI>([&](auto I) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W8[j], A8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, i,
                                                                           (int)sc_cur);
          });
        } else if constexpr (F8) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W8[j], A8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                           0x7f7f7f7f);
        } else {
          mma_step(a0, w0, a1, w1);
        }
      };
      // Two separate code paths, each with its own staging sources, prologue and loop (not one loop with a group
      // branch inside): every fragment and accumulator register is then written unconditionally in its loop -
      // which keeps hipcc from holding second copies of them - and a wave only carries the address registers
      // of the operand it stages.
      constexpr bool DXR = (FLAGS & FLAG_DXR) != 0;
      if constexpr (DXR) {
        // ---- fp32-faithful 3 x 3 conv, horizontal taps from ONE halo tile (FLAG_DXR) --------------------------------------------
        // The phase trace of the N = 128 layers at 512 x 512 (tools/conv_phase_trace.py) has BOTH memory phases at twice the MFMA
        // phases: group 0 issues 8 activation pieces per wave in two of three steps, and a tenth of those lines are first touches
        // that hold the CU's memory queue for the weight pieces as well.  Per channel chunk and filter row dy the nine K-steps
        // (3 taps x 3 passes) need only TWO activation tiles - the hi and the lo plane of pixels x0 - 1 .. x0 + 256 of image row
        // y + dy - 1 - if a tap dx reads its fragments dx rows further down.  K order inside a (chunk, dy) group:
        //     s = 0..5: tap dx = s / 2, pass s % 2 (hi x hi, hi x lo)   <- slot 0 (hi plane),  s = 6..8: tap dx = s - 6, pass 2 (lo x hi)  <- slot 1
        // so the hi tile of the NEXT group is staged while the lo tile is consumed (steps 5, 6, 7) and the next lo tile while the
        // hi tile is (steps 8, 1, 3): 66 pieces per 9 steps instead of 192, at most 3 per wave and phase.  Same sum, another order
        // of the fp32 accumulation than the tap-by-tap loader.  Launcher: ksize 3, stride 1, pad 1, Ws % 256 == 0, no split-K.
        static_assert(X3 && AMODE == 1 && BM == 256 && !F8, "FLAG_DXR: fp32-faithful conv, 256-row tile");
        constexpr int AD_BYTES = 33 * 1024;            // 264 LDS rows (258 used): pixels x0 - 1 .. x0 + 256
        constexpr int WD_BASE = 2 * AD_BYTES;
        const int nR = nkt / 9;                         // (chunk, dy) groups of this block (kbase = 0)
        auto read_dxr = [&](int slot, int dx, int slot_w) {
          const uint32_t t = (uint32_t)(r16 + dx), sw = t & 7u;
          const uint32_t ab = lds0 + slot * AD_BYTES + (wm * WTM) * 128 + t * 128u;
          const uint32_t aa0 = ab + ((((uint32_t)q4) ^ sw) << 4), aa1 = ab + ((((uint32_t)(4 + q4)) ^ sw) << 4);
          const uint32_t bb0 = lds0 + WD_BASE + slot_w * B_BYTES + (wn * WTN) * 128 + foff[0];
          const uint32_t bb1 = lds0 + WD_BASE + slot_w * B_BYTES + (wn * WTN) * 128 + foff[1];
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a0[i]) : "v"(aa0), "n"(i * 2048) : "memory");
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w0[j]) : "v"(bb0), "n"(j * 2048) : "memory");
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a1[i]) : "v"(aa1), "n"(i * 2048) : "memory");
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w1[j]) : "v"(bb1), "n"(j * 2048) : "memory");
        };
        // step s of a group -> (activation slot, tap column)
        auto slot_of = [](int s_) { return s_ < 6 ? 0 : 1; };
        auto dx_of = [](int s_) { return s_ < 6 ? (s_ >> 1) : s_ - 6; };
        if (g0) {
          // this tile: 256 pixels of image row (img, y) from x0 on
          const int hw = p.cv.Ho * p.cv.Wo;
          const int bb = m0 / hw, rem = m0 - bb * hw;
          const int y = rem / p.cv.Wo, x0 = rem - y * p.cv.Wo;
          const uint32_t cbias = (uint32_t)((2 * p.cv.Ws + 2) * p.cv.Cin * 2);
          const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.cv.X - cbias), 0, (int)(uint32_t)(p.cv.x_extent + cbias), 0x00020000);
          // 9 piece slots per wave (33 pieces over 4 waves: the surplus slots re-fetch piece 32): lane offset of LDS row rho = 8 id + lr,
          // i.e. pixel (y - 1, x0 - 1 + rho), constant over the loop; a pixel outside the row (or an unused row) is out of range for good
          uint32_t vo[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const int id = min(lw + i * LW, 32);
            const int rho = id * 8 + lr, x = x0 - 1 + rho;
            const long long pix = ((long long)bb * p.cv.Hs + (y - 1)) * p.cv.Ws + x;
            vo[i] = (x >= 0 && x < p.cv.Ws && rho < 258) ? (uint32_t)(pix * p.cv.Cin * 2 + lc * 16 + (long long)cbias) : 0xFFFFFFF0u;
          }
          // pieces [3 part, 3 part + 3) of the tile (group R, plane) -> slot
          auto stage_part = [&](int R, int plane, auto part_c) {      // part: compile time, so the unrolled loop below is three straight pieces
            constexpr int part = decltype(part_c)::value;
            const int cch = R / 3, dy = R - cch * 3;
            const bool row_ok = (unsigned)(y + dy - 1) < (unsigned)p.cv.Hs;                    // wave-uniform: a filter row outside the image is zeros
            const uint32_t soffs = (uint32_t)(((long long)dy * p.cv.Ws * p.cv.Cin + (cch << 6)) * 2 + (plane ? p.a_lo * 2 : 0));
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              if (i / 3 != part) continue;
              const int id = min(lw + i * LW, 32);
              __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void*)(smem + plane * AD_BYTES + id * 1024), 16,
                                                       row_ok ? vo[i] : 0xFFFFFFF0u, soffs, 0, 0);
            }
          };
          static_for<0, 3>([&](auto P) { stage_part(0, 0, P); stage_part(0, 1, P); });
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          read_dxr(0, 0, 0);
          int R = 0, s_ = 0, sw = 0;
          for (int kt = 0; kt < nkt; ++kt) {
            const int s1 = s_ == 8 ? 0 : s_ + 1, sw1 = sw == 2 ? 0 : sw + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GEMM_STAMP(0)
            __builtin_amdgcn_sched_barrier(0);
            mma_both();
            __builtin_amdgcn_sched_barrier(0);
            GEMM_STAMP(1)
            // my pieces: the hi tile of the next group (issued at steps 5, 6, 7) is first read after B1 of step 8, this group's lo
            // tile (steps 8, 1, 3) after B1 of step 5 - the only two barriers that have to publish landed pieces
            if (s_ == 8 || s_ == 5) wait_vmcnt<0>();
            GEMM_STAMP(2)
            __builtin_amdgcn_s_barrier();            // B1
            GEMM_STAMP(3)
            read_dxr(slot_of(s1), dx_of(s1), sw1);   // (past the last step: a valid slot into registers nobody uses)
            // hi tile of the next group while this group's lo tile is consumed (its last hi reader was step 5, before B1 above);
            // lo tile of THIS group during steps 1 and 3, of the next one at step 8 (the last lo reader was step 8, before B1 above)
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            if (s_ == 5) { if (R + 1 < nR) stage_part(R + 1, 0, I0{}); }
            else if (s_ == 6) { if (R + 1 < nR) stage_part(R + 1, 0, I1{}); }
            else if (s_ == 7) { if (R + 1 < nR) stage_part(R + 1, 0, I2{}); }
            else if (s_ == 8) { if (R + 1 < nR) stage_part(R + 1, 1, I0{}); }
            else if (s_ == 1) { if (R > 0) stage_part(R, 1, I1{}); }
            else if (s_ == 3) { if (R > 0) stage_part(R, 1, I2{}); }
            __builtin_amdgcn_s_barrier();            // B2
            if (s_ == 8) ++R;
            s_ = s1;
            sw = sw1;
          }
        } else {
          uint32_t woff[PB];
          const char* const wbase = (const char*)gW + (long long)b * w_bs * ESZ;
#pragma unroll
          for (int i = 0; i < PB; ++i) {
            const int row = min(lw + i * LW, BPIECES - 1) * 8 + lr;
            woff[i] = (uint32_t)min(n0 + row, N - 1) * (uint32_t)(K * ESZ) + lc * 16;
          }
          // W(kt): group R = kt / 9 -> (chunk, dy), step s -> (dx, pass); the weight column of a tap is tap * Cin + chunk * 64
          auto stage = [&](int kt, int slot) {
            const int R = kt / 9, s_ = kt - R * 9;
            const int cch = R / 3, dy = R - cch * 3;
            const int dx = s_ < 6 ? (s_ >> 1) : s_ - 6;
            const bool lo = s_ < 6 && (s_ & 1);
            const long long koff = ((long long)(dy * 3 + dx) * p.cv.Cin + (cch << 6)) * 2 + (lo ? p.w_lo * 2 : 0);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
              uint32_t wo = woff[i];
              asm volatile("" : "+v"(wo));
              glds16(wbase + koff + (size_t)wo, smem + WD_BASE + slot * B_BYTES + min(lw + i * LW, BPIECES - 1) * 1024);
            }
          };
          stage(0, 0);
          if (nkt > 1) {
            stage(1, 1);
            wait_vmcnt<PB>();
          } else {
            wait_vmcnt<0>();
          }
          __builtin_amdgcn_s_barrier();
          int s_ = 0, sw = 0;
          for (int kt = 0; kt < nkt; ++kt) {
            const int sw1 = sw == 2 ? 0 : sw + 1, sw2 = sw1 == 2 ? 0 : sw1 + 1;
            read_dxr(slot_of(s_), dx_of(s_), sw);
            if (kt + 2 < nkt) {
              stage(kt + 2, sw2);
              wait_vmcnt<PB>();                      // W(kt+1) landed; W(kt+2) may still fly
            } else {
              wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GEMM_STAMP(4)
            __builtin_amdgcn_s_barrier();            // B1
            GEMM_STAMP(5)
            __builtin_amdgcn_sched_barrier(0);
            mma_both();
            __builtin_amdgcn_sched_barrier(0);
            GEMM_STAMP(6)
            __builtin_amdgcn_s_barrier();            // B2
            GEMM_STAMP(7)
            s_ = s_ == 8 ? 0 : s_ + 1;
            sw = sw1;
          }
        }
      } else
      if (g0) {
        // ---- group 0: stages the activation tile.  Dense rows as pointers; conv rows as the two packed
        //      geometry registers of the implicit-GEMM loader (same encoding as cyx / cimg above).
        uint32_t soff[PA];                 // dense rows: byte offset from the (wave-uniform) base of this group / batch
        const char* const abase = (const char*)gA + (long long)b * a_bs * ESZ;
        int gyx[PA];
        uint32_t gimg[PA];
        const uint32_t cbias = AMODE == 1 ? (uint32_t)((2 * p.cv.Ws + 2) * p.cv.Cin * 2) : 0u;
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            AMODE == 1 ? (void*)((const char*)p.cv.X - cbias) : nullptr, 0, AMODE == 1 ? (int)(uint32_t)(p.cv.x_extent + cbias) : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int row = (lw + i * LW) * 8 + lr;
          const int grow = min(m0 + row, Mg - 1);
          if (AMODE == 0) {
            soff[i] = (uint32_t)grow * (uint32_t)(p.lda * ESZ) + lc * 16;      // < 4 GiB per (group, batch): checked by the launcher
          } else {
            const int hw = p.cv.Ho * p.cv.Wo;
            const int bb = grow / hw, rem = grow - bb * hw;
            const int y = rem / p.cv.Wo, x = rem - y * p.cv.Wo;
            const int y0 = y * p.cv.stride - p.cv.pad + c_oy, x0 = x * p.cv.stride - p.cv.pad + c_ox;
            const long long img = (long long)bb * p.cv.Hs * p.cv.Ws * p.cv.Cin;
            if (p.cv.ups) {
              gyx[i] = (y0 << 16) | (x0 & 0xffff);
              gimg[i] = (uint32_t)(img >> 3);
            } else {
              int mask = 0;
              for (int t = 0; t < p.cv.ksize * p.cv.ksize; ++t) {
                const int ty = conv_tap_row(p.cv.ksize, t);
          const int yy = y0 + ty, xx = x0 + t - ty * p.cv.ksize;
                mask |= ((yy >= 0) & (yy < p.cv.Hs) & (xx >= 0) & (xx < p.cv.Ws)) << t;
              }
              gyx[i] = mask;
              gimg[i] = (uint32_t)(int)((img + ((long long)y0 * p.cv.Ws + x0) * p.cv.Cin) >> 3);
              // buffer mode: byte offset from (X - cbias), never negative (the window's top-left is at most two rows and two
              // pixels before the image), this lane's 16-byte chunk included
              if (p.cv.buf) gimg[i] = gimg[i] * 16u + (uint32_t)(lc * 16) + cbias;
            }
          }
        }
        auto stage = [&](int kt, int slot) {       // A(kt) -> activation slot `slot`
          int tap = 0, c0 = 0, dy = 0, dx = 0;
          long long a_adj, w_adj_unused;
          const int ks = kstep(kt, a_adj, w_adj_unused);
          if (AMODE == 1) {                        // K order: channel chunk outer, filter tap inner
            const int ntap = p.cv.ksize * p.cv.ksize;
            const int cch = ks / ntap;
            tap = ks - cch * ntap;
            c0 = cch << 6;
            dy = conv_tap_row(p.cv.ksize, tap);
            dx = tap - dy * p.cv.ksize;
          }
          // One wave-uniform decision per STAGE, then a straight run of PA pieces.  (As one loop with the loader variants selected
          // inside it, hipcc kept the selection per piece: ~6 scalar branches around every LDS-DMA instruction, and the phase trace
          // of the fp32-faithful conv tiles showed this group's memory phase at ~2000 cycles against ~1100 for the dense loader.)
          if constexpr (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
              uint32_t so = soff[i];
              asm volatile("" : "+v"(so));           // opaque: keeps the 64-bit sum out of loop-invariant hoisting
              glds16(abase + ((long long)ks * (BK * 2) + a_adj) + (size_t)so, smem + slot * A_BYTES + (lw + i * LW) * 1024);   // scalar base + 32-bit lane offset
            }
          } else if (!p.cv.ups && p.cv.buf) {
            // border taps: an offset beyond num_records -> the LDS-DMA writes zeros (tools/probes/buffer_lds_oob_probe.hip)
            const uint32_t soffs = (uint32_t)(((long long)(dy * p.cv.Ws + dx) * p.cv.Cin + c0) * 2 + a_adj);    // wave-uniform
#pragma unroll
            for (int i = 0; i < PA; ++i) {
              const uint32_t vo = ((gyx[i] >> tap) & 1) ? gimg[i] : 0xFFFFFFF0u;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void*)(smem + slot * A_BYTES + (lw + i * LW) * 1024),
                                                       16, vo, soffs, 0, 0);
            }
          } else if (!p.cv.ups) {
            const long long uoff = ((long long)(dy * p.cv.Ws + dx) * p.cv.Cin + c0) * 2 + a_adj;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
              // opaque copy: otherwise hipcc hoists the loop-invariant 64-bit cX + img * 16 of every piece out of the
              // K loop, spills the 8 pairs to scratch and reloads one per piece per stage - VMEM loads whose vmcnt(0)
              // waits drain the LDS-DMA pieces in flight
              uint32_t img = gimg[i];
              asm volatile("" : "+v"(img));
              const char* s_ = ((gyx[i] >> tap) & 1) ? cX + (long long)(int)img * 16 + uoff : (const char*)p.cv.zero;
              glds16(s_, smem + slot * A_BYTES + (lw + i * LW) * 1024);
            }
          } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
              const int yy = (gyx[i] >> 16) + dy, xx = (int)(short)(gyx[i] & 0xffff) + dx;
              const bool ok = (yy >= 0) & (yy < p.cv.Hs * 2) & (xx >= 0) & (xx < p.cv.Ws * 2);
              uint32_t img = gimg[i];
              asm volatile("" : "+v"(img));
              const char* s_ = ok ? cX + (long long)img * 16 + ((long long)((yy >> 1) * p.cv.Ws + (xx >> 1)) * p.cv.Cin + c0) * 2 + a_adj
                                  : (const char*)p.cv.zero;
              glds16(s_, smem + slot * A_BYTES + (lw + i * LW) * 1024);
            }
          }
        };
        // Only step 0 has to be in LDS before the first MFMA: waiting for step 1 as well made every CU of the
        // chip sit through a second 64-KiB fetch of the cold-start burst.  Step 1 is waited for where it is needed.
        // FLAG_SPLIT (REUSE_HI): passes 0 and 1 of an operand step multiply the SAME activation plane (hi x hi, hi x lo), so the
        // hi tile is staged once and read for two steps: activation slot 0 holds the current hi tile, slot 1 the current lo
        // tile (pass 2), and the group issues two activation stages per three steps instead of three - a third less LDS-DMA
        // and, for the implicit-GEMM loader, a third less address generation in a loop that runs at ~57 % of the MFMA rate
        // because of exactly that work.  pk = pass of step kt (a split-K block may start mid-triple).
        constexpr bool REUSE_HI = X3;
        int pk = REUSE_HI ? kbase % 3 : 0;
        if constexpr (REUSE_HI) {
          const int j = pk == 0 ? 2 : 1;           // first step with another activation tile than step 0
          stage(0, pk == 2);
          if (j < nkt) {
            stage(j, (pk + j) % 3 == 2);
            wait_vmcnt<PA>();
          } else {
            wait_vmcnt<0>();
          }
        } else {
        if constexpr (MXA) {                       // scales of steps 0 and 1: older than every piece below
          mx_load(sc_cur);
          if (nkt > 1) mx_load(sc_nxt);
        }
        stage(0, 0);
        if (nkt > 1) {
          stage(1, 1);
          wait_vmcnt<PA>();
        } else {
          wait_vmcnt<0>();
        }
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (MXA) asm volatile("" : "+v"(sc_cur));       // landed (the wait above): readable from here on
        read_both(REUSE_HI ? (pk == 2) : 0, 0);
        int sa = REUSE_HI ? (pk == 2) : 0, sw = 0; // ring slots of step kt
        if constexpr (F8) {
          // The same schedule with the loop rotated (first MFMA phase peeled): the fragments of step kt are read and
          // consumed inside ONE iteration.  As loop-carried values (read at the bottom for the next trip) hipcc kept
          // every 8-register operand twice - read into one set, v_mov'ed into another during the MFMA phase.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          mma_both();
          __builtin_amdgcn_sched_barrier(0);
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();            // B1 of step 0
          for (int kt = 1; kt < nkt; ++kt) {
            const int sprev = sa;
            sa ^= 1;
            sw = sw == 2 ? 0 : sw + 1;
            if constexpr (MXA) {                   // scales of step kt: landed at the wait_vmcnt<0> that closed the previous trip
              asm volatile("" : "+v"(sc_nxt));
              sc_cur = sc_nxt;
            }
            read_both(sa, sw);                     // memory phase of step kt - 1: fragments of step kt ...
            if (kt + 1 < nkt) {
              if constexpr (MXA) mx_load(sc_nxt);   // scales of step kt + 1
              stage(kt + 1, sprev);                // ... and A(kt + 1) into the slot step kt - 1 has drained
            }
            __builtin_amdgcn_s_barrier();          // B2 of step kt - 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_both();
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<0>();                       // my A(kt+1) pieces
            __builtin_amdgcn_s_barrier();          // B1 of step kt
          }
          __builtin_amdgcn_s_barrier();            // B2 of the last step (group 1's MFMA phase)
        } else
        for (int kt = 0; kt < nkt; ++kt) {
          const int sa1 = REUSE_HI ? (pk == 1) : sa ^ 1, sw1 = sw == 2 ? 0 : sw + 1;   // REUSE_HI: step kt + 1 is a pass 2 iff this is a pass 1
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          GEMM_STAMP(0)                            // (FLAG_TIMED) memory phase of the previous step: reads + A pieces issued, B2, fragments landed
          __builtin_amdgcn_sched_barrier(0);
          mma_both();
          __builtin_amdgcn_sched_barrier(0);
          GEMM_STAMP(1)                            // 2 x MI x NJ MFMAs issued
          wait_vmcnt<0>();                         // my A(kt+1) pieces (issued one phase ago)
          GEMM_STAMP(2)
          __builtin_amdgcn_s_barrier();            // B1
          GEMM_STAMP(3)                            // wait at B1 (for group 1's memory phase)
          // (past the last step this re-reads a valid slot into registers nobody uses: keeps the body branch-free)
          if constexpr (ILV) {
            if (kt + 2 < nkt) {
              long long a_adj, w_adj_unused;
              const int ks = kstep(kt + 2, a_adj, w_adj_unused);
              const char* const src0 = abase + ((long long)ks * (BK * 2) + a_adj);
              static_for<0, PA>([&](auto I) {
                constexpr int i = decltype(I)::value;
                static_for<i * NRD2 / PA, (i + 1) * NRD2 / PA>([&](auto R) { read_flat(R, sa1, sw1, a0, w0, a1, w1); });
                uint32_t so = soff[i];
                asm volatile("" : "+v"(so));
                glds16(src0 + (size_t)so, smem + sa * A_BYTES + (lw + i * LW) * 1024);
                __builtin_amdgcn_sched_barrier(0);
              });
            } else {
              read_both(sa1, sw1);
            }
          } else {
          read_both(sa1, sw1);
          if constexpr (REUSE_HI) {
            // pass 0: slot 0 is read again by step kt + 1, nothing to stage; pass 1: slot 0 has had its last reader
            // (group 1 read step kt before B1) -> the next hi tile, needed at kt + 2; pass 2: slot 1 -> the next lo tile (kt + 3)
            if (pk == 1) { if (kt + 2 < nkt) stage(kt + 2, 0); }
            else if (pk == 2) { if (kt + 3 < nkt) stage(kt + 3, 1); }
            pk = pk == 2 ? 0 : pk + 1;
          } else {
            if (kt + 2 < nkt) stage(kt + 2, sa);
          }
          }
          __builtin_amdgcn_s_barrier();            // B2
          sa = sa1;
          sw = sw1;
        }
      } else {
        // ---- group 1: stages the weight tile (conv: weight column = tap * Cin + channel chunk)
        uint32_t woff[PB];                 // byte offset of the piece's row from the weight base of this group / batch
        const char* const wbase = (const char*)gW + (long long)b * w_bs * ESZ;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int row = min(lw + i * LW, BPIECES - 1) * 8 + lr;
          woff[i] = (uint32_t)min(n0 + row, N - 1) * (uint32_t)(K * ESZ) + lc * 16;
        }
        auto stage = [&](int kt, int slot) {       // W(kt) -> weight slot `slot`
          long long a_adj_unused, w_adj;
          const int ks = kstep(kt, a_adj_unused, w_adj);
          long long koff = (long long)ks * (BK * 2);
          if (AMODE == 1) {
            const int ntap = p.cv.ksize * p.cv.ksize;
            const int cch = ks / ntap;
            koff = ((long long)(ks - cch * ntap) * p.cv.Cin + (cch << 6)) * 2;
          }
          koff += w_adj;
#pragma unroll
          for (int i = 0; i < PB; ++i)
          {
            uint32_t wo = woff[i];
            asm volatile("" : "+v"(wo));
            glds16(wbase + koff + (size_t)wo, smem + W_BASE + slot * B_BYTES + min(lw + i * LW, BPIECES - 1) * 1024);
          }
        };
        if constexpr (MXA) mx_load(sc_cur);        // scales of step 0: older than every piece below
        stage(0, 0);
        if (nkt > 1) {
          stage(1, 1);
          wait_vmcnt<PB>();
        } else {
          wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (MXA) asm volatile("" : "+v"(sc_cur));
        int sa = 0, sw = 0;
        int pk = X3 ? kbase % 3 : 0;               // FLAG_SPLIT: the activation slot of a step is its pass (see group 0)
        for (int kt = 0; kt < nkt; ++kt) {
          const int sw1 = sw == 2 ? 0 : sw + 1, sw2 = sw1 == 2 ? 0 : sw1 + 1;
          if constexpr (X3) sa = pk == 2;
          if constexpr (ILV) {
            if (kt + 2 < nkt) {
              long long a_adj_unused, w_adj;
              const int ks = kstep(kt + 2, a_adj_unused, w_adj);
              const char* const src0 = wbase + ((long long)ks * (BK * 2) + w_adj);
              static_for<0, PB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                static_for<i * NRD2 / PB, (i + 1) * NRD2 / PB>([&](auto R) { read_flat(R, sa, sw, a0, w0, a1, w1); });
                uint32_t wo = woff[i];
                asm volatile("" : "+v"(wo));
                glds16(src0 + (size_t)wo, smem + W_BASE + sw2 * B_BYTES + min(lw + i * LW, BPIECES - 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
              });
              wait_vmcnt<PB>();
            } else {
              read_both(sa, sw);
              wait_vmcnt<0>();
            }
          } else {
          read_both(sa, sw);
          if constexpr (MXA) { if (kt + 1 < nkt) mx_load(sc_nxt); }   // scales of step kt + 1, ahead of W(kt + 2): the wait below lands them
          if (kt + 2 < nkt) {
            stage(kt + 2, sw2);
            wait_vmcnt<PB>();                      // W(kt+1) landed; W(kt+2) may still fly
          } else {
            wait_vmcnt<0>();
          }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of slot kt are done before it is refilled
          GEMM_STAMP(4)                            // (FLAG_TIMED) memory phase: reads + W pieces issued, W(kt+1) landed, fragments landed
          __builtin_amdgcn_s_barrier();            // B1
          GEMM_STAMP(5)                            // wait at B1 (for group 0's MFMA phase)
          __builtin_amdgcn_sched_barrier(0);
          mma_both();
          __builtin_amdgcn_sched_barrier(0);
          GEMM_STAMP(6)                            // MFMAs issued
          __builtin_amdgcn_s_barrier();            // B2
          GEMM_STAMP(7)                            // wait at B2 (for group 0's memory phase)
          if constexpr (MXA) {
            asm volatile("" : "+v"(sc_nxt));
            sc_cur = sc_nxt;
          }
          sa ^= 1;
          sw = sw1;
          if constexpr (X3) pk = pk == 2 ? 0 : pk + 1;
        }
      }
    } else if constexpr (PIPEX >= 4) {
      // Spread LDS-DMA issue.  The CU's texture-address path takes ~16 cycles per 1-KiB piece, so the 64
      // pieces of a K-step issued back to back after the barrier (PIPE 1/3) hold every wave in the issue
      // queue for ~700 cycles with the MFMA pipes idle (tools/gemm_phase_trace.py).  Here each piece is
      // issued between MFMAs: the weight pieces of step kt+NSW-1 inside the first MFMA cluster (their slot
      // was drained before the previous barrier), the activation pieces of step kt+NSA inside the second.
      // Issue order: prologue A0 W0 .. A(NSA-1) W(NSA-1) W(NSA) .. W(NSW-2); iteration j: W(j+NSW-1), [wait,
      // barrier], A(j+NSA).  When step kt+1 is needed, the loads younger than A(kt+1) are NSA-1 weight
      // steps and NSA-2 activation steps.
      // PIPE 5 also spreads the fragment reads: the 12..16 ds_reads of the next MFMA cluster are issued one
      // per MFMA inside the current cluster instead of in a block before it (each block of reads kept the
      // matrix pipe waiting ~150-250 cycles, twice per K-step).
      constexpr bool RDS = PIPEX == 5;
      static_assert(NSW == NSA + 1, "PIPE 4/5 need the deeper weight ring");
      constexpr int PEND4 = (NSA - 1) * BPW + (NSA - 2) * APW;
      constexpr int NM = MI * NJ;
      // (compile-time indices throughout: a fragment that is only conditionally written costs a second live copy)
      auto no_read = [](auto) {};
      auto mma_spread = [&](const bf16x8(&af)[MI], const bf16x8(&wf)[NJ], auto&& piece, auto np_tag, auto&& rd,
                            auto nrd_tag) {
        constexpr int np = decltype(np_tag)::value, nrd = decltype(nrd_tag)::value;
        static_for<0, NM>([&](auto I) {
          constexpr int i = I.value / NJ, j = I.value % NJ, idx = I.value + 1;
          acc[i][j] = mfma16<F16>(wf[j], af[i], acc[i][j]);
          if constexpr (idx <= nrd) {
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, idx - 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (np > 0) {
            if constexpr ((idx - NM / (2 * np)) % (NM / np) == 0 && idx >= NM / (2 * np)) {
              __builtin_amdgcn_sched_barrier(0);
              piece((idx - NM / (2 * np)) / (NM / np));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      };
      // fragment read number r of a cluster: r < MI -> activation fragment r, else weight fragment r - MI
      auto read_one = [&](bf16x8(&af)[MI], bf16x8(&wf)[NJ], uint32_t aa, uint32_t bb, auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        if constexpr (r < MI)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[r]) : "v"(aa), "n"(r * 2048) : "memory");
        else
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[r - MI]) : "v"(bb), "n"((r - MI) * 2048) : "memory");
      };
      constexpr int NRD = (MI + NJ) <= NM ? (MI + NJ) : NM;
      static_assert(!RDS || MI + NJ <= NM, "spread reads need at least as many MFMAs as fragments");
#pragma unroll
      for (int s = 0; s < NSA; ++s)
        if (s < nkt) { stage_a(s, s); stage_w(s, s); }
#pragma unroll
      for (int s = NSA; s < NSW - 1; ++s)
        if (s < nkt) stage_w(s, s);
      if (nkt >= NSW) wait_vmcnt<(NSA - 1) * G + (NSW - 1 - NSA) * BPW>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      bf16x8 a0[MI], w0[NJ], a1[MI], w1[NJ];
      read_frags(a0, w0, 0, 0, 0);
      int ca = 0, cw = 0, pw = NSW - 1;                  // slots of step kt (A, W) and of W step kt+NSW-1
      for (int kt = 0; kt < nkt; ++kt) {
        const int na = (ca + 1 == NSA) ? 0 : ca + 1;
        const int nw = (cw + 1 == NSW) ? 0 : cw + 1;
        const bool more_w = kt + NSW - 1 < nkt, more_a = kt + NSA < nkt;
        if constexpr (RDS) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // a0/w0 (requested inside the previous cluster)
        } else {
          read_frags(a1, w1, ca, cw, 1);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF) : "memory");
        }
        if constexpr (RDS) { GEMM_STAMP(3) if (TIMED) tsum[7] += 1; }   // second cluster of the previous step + loop turn
        __builtin_amdgcn_sched_barrier(0);
        using IC0 = std::integral_constant<int, 0>;
        if constexpr (RDS) {
          const uint32_t aa = a_rd + ca * A_BYTES + foff[1], bb = b_rd + cw * B_BYTES + foff[1];
          mma_spread(a0, w0, [&](int q) { if (more_w) stage_w(kt + NSW - 1, pw, q, q + 1); },
                     std::integral_constant<int, BPW>{}, [&](auto r) { read_one(a1, w1, aa, bb, r); },
                     std::integral_constant<int, NRD>{});
        } else {
          mma_spread(a0, w0, [&](int q) { if (more_w) stage_w(kt + NSW - 1, pw, q, q + 1); },
                     std::integral_constant<int, BPW>{}, no_read, IC0{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RDS) { GEMM_STAMP(0) }                                  // first cluster (MFMAs + reads + weight pieces)
        if (kt + 1 < nkt) {
          if (more_w) wait_vmcnt<PEND4>();
          else wait_vmcnt<0>();
          if constexpr (RDS) { GEMM_STAMP(1) }                                // counted LDS-DMA wait
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if constexpr (RDS) { GEMM_STAMP(2) }                                // barrier
          if constexpr (!RDS) read_frags(a0, w0, na, nw, 0);
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RDS) {
          // (past the last step this re-reads a valid slot into registers nobody uses: keeps the loop branch-free)
          const uint32_t aa = a_rd + na * A_BYTES + foff[0], bb = b_rd + nw * B_BYTES + foff[0];
          mma_spread(a1, w1, [&](int q) { if (more_a) stage_a(kt + NSA, ca, q, q + 1); },
                     std::integral_constant<int, APW>{}, [&](auto r) { read_one(a0, w0, aa, bb, r); },
                     std::integral_constant<int, NRD>{});
        } else {
          mma_spread(a1, w1, [&](int q) { if (more_a) stage_a(kt + NSA, ca, q, q + 1); },
                     std::integral_constant<int, APW>{}, no_read, IC0{});
        }
        __builtin_amdgcn_sched_barrier(0);
        ca = na;
        pw = cw;
        cw = nw;
      }
    } else {
#pragma unroll
    for (int s = 0; s < NSA; ++s)
      if (s < nkt) { stage_a(s, s); stage_w(s, s); }
#pragma unroll
    for (int s = NSA; s < NSW; ++s)
      if (s < nkt) stage_w(s, s);
    if (nkt >= NSW) wait_vmcnt<(NSA - 1) * G + (NSW - NSA) * BPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[MI], w0[NJ], a1[MI], w1[NJ];
    read_frags(a0, w0, 0, 0, 0);
    int ca = 0, cw = 0;                                // ring slots of step kt
    for (int kt = 0; kt < nkt; ++kt) {
      const int na = (ca + 1 == NSA) ? 0 : ca + 1;
      const int nw = (cw + 1 == NSW) ? 0 : cw + 1;
      read_frags(a1, w1, ca, cw, 1);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF) : "memory");   // a0/w0 landed, a1/w1 in flight
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, w0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkt) {
        if (kt + NSW - 1 < nkt) wait_vmcnt<PENDING>();                // step kt+1 landed (my pieces)
        else wait_vmcnt<0>();                                          // tail: fewer steps in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my reads of the current slots are done
        __builtin_amdgcn_s_barrier();                                   // ... and everybody else's
        if (kt + NSA < nkt) stage_a(kt + NSA, ca);                      // refill the slots just drained
        if (kt + NSW < nkt) stage_w(kt + NSW, cw);
        read_frags(a0, w0, na, nw, 0);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, w1);
      __builtin_amdgcn_sched_barrier(0);
      ca = na;
      cw = nw;
    }
    }
    if constexpr (TIMED) {
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_loop1)::"memory");
      if (p.trace && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.trace[((long long)blockIdx.x * NWAVES + wave) * 16 + i] = tsum[i];
      }
    }
#undef GEMM_STAMP
  }

  // ---- operands of the epilogue that depend on nothing the split-K hand-off produces (bias, dequantisation scales, residual and
  //      gate): declared here so that a reduce-scatter block can request them while it waits for its peers
  const float alpha = p.alpha;
  constexpr int LEPI = (FLAGS >> 8) & 15;      // FLAG_LEAN kernels may fix their ONE epilogue at compile time (see the epilogue)
  const int epi = LEPI ? LEPI - 1 : p.epi;
  // bias operands are fetched up front (one batch of loads in flight, not one dependent load per MFMA tile): by a reduce-scatter
  // block right after the drain of its partial stores (a vmcnt(0) that would otherwise wait for these loads as well), by every other
  // launch where the epilogue starts
  u32x2 bcol[NJ];
  float brow[MI];
  bool bias_loaded = false;
  auto load_bias = [&]() {
    const bool colb = gBias && (LEAN || !p.row_bias), rowb = !LEAN && gBias && p.row_bias;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n4 = min(n0 + wn * WTN + j * 16 + q4 * 4, N - 4);
      bcol[j] = colb ? *(const u32x2*)(gBias + n4) : u32x2{0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) brow[i] = rowb ? e2f<F16>(gBias[min(m0 + wm * WTM + i * 16 + r16, Mg - 1)]) : 0.f;
    bias_loaded = true;
  };
  // FLAG_FP8: dequantisation scales of this lane's rows (activation, per token) and columns (weight, per channel)
  float asc[F8 ? MI : 1];
  f32x4 wsc[F8 ? NJ : 1];
  if constexpr (F8) {
    const float* as_ = (g1 ? p.a_scale[1] : p.a_scale[0]) + (long long)b * p.a_sc_bstride;
    const float* ws_ = g1 ? p.w_scale[1] : p.w_scale[0];
#pragma unroll
    for (int i = 0; i < MI; ++i) asc[i] = MXA ? alpha : as_[min(m0 + wm * WTM + i * 16 + r16, Mg - 1)] * alpha;   // MXA: the MFMA applied the block scales
#pragma unroll
    for (int j = 0; j < NJ; ++j) wsc[j] = *(const f32x4*)(ws_ + min(n0 + wn * WTN + j * 16 + q4 * 4, N - 4));
  }
  // Gate-residual epilogue (phase B of the LDS-transposed path): item t * 64 + lane of this wave = 8 consecutive columns of one
  // row of its sub-tile (of the part a reduce-scatter block owns: rows [r_lo, r_lo + nr) x chunks [c_lo, c_lo + ncw)).  gr_load
  // requests the residual and gate operands of FOUR items per lane before any is used (clamped addresses, no branch around a
  // load): as one item per trip - LDS read, residual + gate loads, arithmetic, store - the loop was a chain of dependent global
  // round trips, ~6 us for the third of a tile a reduce-scatter block finishes against 4.3 us for a WHOLE tile of the bias-only
  // epilogue.  A reduce-scatter block requests its first four items BEFORE the exchange (its ownership is known from its split
  // index), so their round trip runs under the hand-off; the residual may alias the output: only this block writes these elements.
  constexpr int NCH_E = WTN / 8;
  struct GrItems { bool ok[4]; int rowc[4], cc[4]; long long off[4]; u32x4 rv[4], gv[4]; };
  auto gr_load = [&](GrItems& it, int t0, int nit, bool sub, int r_lo, int nr, int c_lo, int ncw) {
    const bf16_t* const resb = gRes + (long long)b * c_bs;
    const bf16_t* const gateb = gGate ? gGate + (long long)b * gate_bs : nullptr;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = (t0 + u) * 64 + lane;
      int row, c;
      bool v_ = t0 + u < nit;
      if (sub) {
        const int rr = (int)((unsigned)idx / (unsigned)ncw);
        row = r_lo + rr;
        c = c_lo + idx - rr * ncw;
        v_ = v_ && rr < nr;
      } else {
        row = idx / NCH_E;
        c = idx - row * NCH_E;
      }
      const int m = m0 + wm * WTM + row, n8 = n0 + wn * WTN + c * 8;
      v_ = v_ && row < WTM && m < Mg && n8 < N;
      it.ok[u] = v_;
      it.rowc[u] = v_ ? row : 0;
      it.cc[u] = v_ ? c : 0;
      const int mc = v_ ? m : min(m0, Mg - 1), nc = v_ ? n8 : 0;       // (an item that is switched off reads a valid address and drops the value)
      it.off[u] = (long long)mc * p.ldc + nc;
      it.rv[u] = *(const u32x4*)(resb + it.off[u]);
      it.gv[u] = gateb ? *(const u32x4*)(gateb + nc) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  GrItems gr_pf;
  bool gr_prefetched = false;
  // ---- split-K hand-off (see the block map above) ---------------------------------------------
  // Owned fragment range of this block: everything unless the reduce-scatter hand-off below narrows it.
  int oi_lo = 0, oi_hi = MI, oj_lo = 0, oj_hi = NJ;
  // Mode 1, REDUCE-SCATTER (ping-pong bf16 tiles; every split block of the grid resident at once).  The chain below costs
  // ~20 us for its first hop and ~8 us for each further one: S - 1 serial {store the whole fp32 tile, release fence =
  // buffer_wbl2 walking an L2 that 30 other blocks are dirtying, flag, acquire, load the whole tile}.  Here the S blocks of
  // a tile finish their K ranges together and EACH ends up owning 1/S of the tile (fragment rows if MI % S == 0, else
  // fragment columns): a block stores the (S-1)/S it does not own WRITE-THROUGH (sc1: the bytes leave this XCD's L2 as they
  // are written, so no release fence and no L2 walk), every storing wave drains vmcnt, the block arrives on the tile's
  // counter, one lane polls it (relaxed) until all S have arrived, ONE agent acquire drops this CU's stale L1 lines, and the
  // block adds its peers' partials of the fragments it owns (fixed peer order: deterministic) and runs the epilogue on
  // them only.  All S hand-offs are concurrent; bytes per block: (S-1)/S of a tile out, the same in.  Placement
  // independent (cdna guide, Guideline 16 R1).  The last block to leave a tile zeroes both counters.
  // The blocks of a tile wait for EACH OTHER, which is only live when all of them are resident.  The launcher selects the
  // mode when grid <= CUs, but residency is not its to guarantee (a second process on the GPU, masked CUs, a partition
  // mode), so the wait is BOUNDED and the protocol has a wait-free completion: the tile's counter word carries the
  // arrivals in bits 0-7 and an ORPHAN mask above.  A block whose peers have not all arrived after p.sk_timeout publishes
  // the partial of its OWN slice as well (the slab has the slot), sets its orphan bit with the same kind of atomic and
  // EXITS, freeing its CU.  The block whose arrival completes the count is the one party that never waits: it finishes
  // its own slice and then every slice whose orphan bit was set BEFORE its arrival (the atomic's return value) from the
  // slab — same summation order (owner's partial first, then the peers by index), same bf16 epilogue arithmetic, so the
  // result is bit-identical whichever path a slice took.  An orphan bit set AFTER the last arrival is seen by its own
  // setter (count == S in the value its atomic returns), who then simply continues on the fast path.  Nobody waits
  // unboundedly, nobody depends on co-residency.
  constexpr bool RS_CAPABLE = (FLAGS & FLAG_RS) != 0;
  static_assert(!RS_CAPABLE || (PP && !X3 && !F8 && AMODE == 0), "reduce-scatter split-K: ping-pong dense bf16 tiles");
  bool rs = false;
  int rs_orphans = 0;                                   // last arriver only: slices whose owners gave up waiting (bit s)
  if constexpr (RS_CAPABLE) {
    if (LEAN || (S > 1 && p.sk_mode != 0)) {
      rs = true;
      const bool by_rows = (MI % S) == 0;             // the launcher guarantees MI % S == 0 or NJ % S == 0
      if (by_rows) { oi_lo = sidx * (MI / S); oi_hi = oi_lo + MI / S; }
      else { oj_lo = sidx * (NJ / S); oj_hi = oj_lo + NJ / S; }
      constexpr int F = MI * NJ;                        // fragments (1 KiB of fp32 each) per wave
      // slab of this tile: [S sources][NWAVES][F][64 lanes] f32x4 (a source never writes the fragments it owns)
      float* slab = p.sk_part + (size_t)bid * ((size_t)S * BM * BN);
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, S * BM * BN * 4, 0x00020000);
      const int my_off = ((sidx * NWAVES + wave) * F) * 1024 + lane * 16;
      // (Measured and dropped: when all S blocks verifiably share one XCD — they do under the observed dispatch order —
      //  plain stores would keep the exchange inside that XCD's L2.  It is SLOWER in situ, 6.32 vs 6.12 ms for the 57 split
      //  launches of a forward: 30 blocks x 131 KB of dirty partials per XCD evict the operand panels from the 4 MB L2 and
      //  still have to be written back at the end of the kernel.  Write-through keeps the L2 clean and needs no placement.)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (i >= oi_lo && i < oi_hi && j >= oj_lo && j < oj_hi) continue;      // wave-uniform
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc, my_off + (i * NJ + j) * 1024, 0,
                                                 /*sc1: write-through*/ 16);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores
      unsigned long long t_rs1 = 0, t_rs2 = 0, t_rs3 = 0;
      if constexpr ((FLAGS & FLAG_TIMED) != 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_rs1)::"memory");
      __syncthreads();
      int* arrive = p.sk_flag + bid;
      int* const bcast = (int*)smem;                  // the operand ring is dead (block barrier above)
      int old = 0;
      if (tid == 0) old = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // Epilogue operands of the part this block owns (bias; residual + gate of its first four items, see gr_load): requested HERE -
      // after the arrival (the peers are not kept waiting) and behind the block barrier above (a __syncthreads() ahead of them would
      // wait for them: measured, the whole hand-off moved out by their latency) - so they land while lane 0 polls for the peers.
      load_bias();
      if (epi == EPI_GATE_RES) {
        const int nr_ = (oi_hi - oi_lo) * 16, ncw_ = (oj_hi - oj_lo) * 2;
        gr_load(gr_pf, 0, (nr_ * ncw_ + 63) >> 6, true, oi_lo * 16, nr_, oj_lo * 2, ncw_);
        gr_prefetched = true;
      }
      if (tid == 0) {
        int gave_up = 0, orphans = 0;
        if ((old & 255) + 1 == S) {
          orphans = old >> 8;                           // last arriver: never waits, finishes what was orphaned before it came
        } else {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          while ((__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 255) < S) {
            if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)p.sk_timeout) { gave_up = 1; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        bcast[0] = gave_up;
        bcast[1] = orphans;
        // (no acquire on the fast path: the peers' partials are read with sc1 loads below - write-through stores on the producing
        //  side and L1-bypassing loads on this one is a complete hand-off, cdna guide Guideline 16 / correctness table; the fence was a
        //  buffer_inv on the critical path of every split launch, ~1.7 us behind the last arrival)
      }
      __syncthreads();
      int gave_up = __builtin_amdgcn_readfirstlane(bcast[0]);
      rs_orphans = __builtin_amdgcn_readfirstlane(bcast[1]);
      if (gave_up) {                                    // rare: peers not resident (shared / partitioned GPU)
        __syncthreads();                                // everybody has read the broadcast words
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (i < oi_lo || i >= oi_hi || j < oj_lo || j >= oj_hi) continue;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc, my_off + (i * NJ + j) * 1024, 0, 16);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const int old2 = __hip_atomic_fetch_add(arrive, 1 << (8 + sidx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int still = (old2 & 255) < S;           // the last arriver is yet to come and will see the bit
          bcast[0] = still;
          if (!still) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        gave_up = __builtin_amdgcn_readfirstlane(bcast[0]);
        if (gave_up) {
          if (tid == 0) {
            const int left = __hip_atomic_fetch_add(p.sk_depart + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == S - 1) {
              __hip_atomic_store(p.sk_depart + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          return;
        }
      }
      if constexpr ((FLAGS & FLAG_TIMED) != 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_rs2)::"memory");
      // The peers' partials of the fragments this block owns, added in peer order (fixed summation order).  Ownership depends on
      // (S, split index), run-time values, so every fragment sits behind an "owned?" test - and around a LOAD into registers hipcc
      // then waits vmcnt(0) before the add that follows: 16 dependent round trips per wave, 7.4 us of every split launch (the
      // disassembly; the MI355X guide's "register or load" trap).  Dispatching once on (S, split index) makes the owned set a
      // compile-time list, but its nine instantiations cost the kernel 56 registers and spills into the main loop's budget
      // (measured: the step got slower).  So the partials travel by LDS-DMA instead (a fragment of a wave is exactly one 1-KiB
      // piece; sc1 like the loads they replace): a piece needs no destination register and no wait, the tests around them cost
      // scalar time only, ONE vmcnt(0) lands them all in this wave's own 16 KiB of the dead operand ring, and the adds read them
      // back from LDS (~100 cycles each instead of a memory round trip).  Up to 16 fragments per round: all peers at once for
      // S <= 3 on these tiles, peer by peer otherwise.  Same operands in the same order: bit-identical to the register form.
      {
        static_assert(4096 + NWAVES * 16384 <= NSA * A_BYTES + NSW * B_BYTES, "reduce-scatter gather area inside the operand ring");
        char* const gl = smem + 4096 + wave * 16384;          // (the first words of smem are the broadcast words above)
        const int own = (oi_hi - oi_lo) * (oj_hi - oj_lo);
        const bool one_round = (S - 1) * own <= 16;
        const char* const slab_b = (const char*)slab;
        for (int r0 = 0; r0 < S; r0 = one_round ? S : r0 + 1) {
          const int r1 = one_round ? S : r0 + 1;
          int n = 0;
          for (int s2 = r0; s2 < r1; ++s2) {
            if (s2 == sidx) continue;
            const size_t src_off = (size_t)((s2 * NWAVES + wave) * F) * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                if (i < oi_lo || i >= oi_hi || j < oj_lo || j >= oj_hi) continue;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slab_b + src_off + (i * NJ + j) * 1024),
                                                 (__attribute__((address_space(3))) void*)(gl + n * 1024), 16, 0, /*sc1*/ 16);
                ++n;
              }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces have landed (only this wave reads them: no barrier)
          n = 0;
          for (int s2 = r0; s2 < r1; ++s2) {
            if (s2 == sidx) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                if (i < oi_lo || i >= oi_hi || j < oj_lo || j >= oj_hi) continue;
                acc[i][j] += *(const f32x4*)(gl + n * 1024 + lane * 16);
                ++n;
              }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next round (or the epilogue) reuses the area
        }
      }
      if constexpr ((FLAGS & FLAG_TIMED) != 0) {
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_rs3)::"memory");
        if (p.trace && lane == 0) {
          unsigned long long* t = p.trace + ((long long)blockIdx.x * NWAVES + wave) * 16;
          t[0] = t_rs1 - t_loop1;      // write-through stores issued + drained
          t[1] = t_rs2 - t_rs1;        // block barrier, arrival, poll for the peers, acquire, block barrier
          t[2] = t_rs3 - t_rs2;        // peers' partials loaded and added
          t[3] = sidx;
        }
      }
      __syncthreads();                                       // every wave has consumed its peers' partials
      if (tid == 0) {
        const int left = __hip_atomic_fetch_add(p.sk_depart + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == S - 1) {                                 // everybody has passed the poll: clean for the next launch
          __hip_atomic_store(p.sk_depart + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  // Mode 0, CHAIN.  The producer publishes write-through (see below), the consumer acquires at agent scope, ONE lane per block.
  if (!(LEAN && RS_CAPABLE) && S > 1 && !rs) {
    f32x4* part = (f32x4*)(p.sk_part + (size_t)bid * (BM * BN)) + (size_t)wave * (MI * NJ) * 64 + lane;
    int* flag = p.sk_flag + bid;
    if (sidx > 0) {
      if (tid == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sidx) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv only: drop this CU's stale L1 lines
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 prev = __builtin_nontemporal_load(part + (i * NJ + j) * 64);
          acc[i][j] += prev;
        }
    }
    if (sidx < S - 1) {
      // The partial leaves WRITE-THROUGH (sc1, like the reduce-scatter slabs): the bytes go to memory as they are written, every
      // storing wave drains its own stores, the block barrier collects the drains and lane 0 moves the counter - no release fence.
      // (The fence is a buffer_wbl2 that walks an L2 which the other blocks of the XCD are still dirtying: ~6 us per hop.  The
      // phase trace of the fp32-faithful 64 x 64 convs of the VAE, 256 x 128 tiles with S = 4: 29 us of a 92 us launch in the
      // three hops.)  cdna guide, Guideline 16 R1; placement independent.
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.sk_part + (size_t)bid * (BM * BN), 0, BM * BN * 4, 0x00020000);
      const int poff = (wave * (MI * NJ)) * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), prs, poff + (i * NJ + j) * 1024, 0, /*sc1*/ 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains its write-through stores
      __syncthreads();                                     // ... all of them have
      if (tid == 0) __hip_atomic_store(flag, sidx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __syncthreads();                                // every wave has consumed the partial
    if (tid == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
  }

  // ---- epilogue ------------------------------------------------------------------------------
  // After the MFMAs a lane holds C[m][n4 .. n4+3] for 16 different rows per wave-instruction: stored
  // directly that is 32-byte runs, and the 256 x 256 tile's store tail costs ~40 % of its main loop
  // (tools/gemm_phase_trace.py).  So the tile goes through LDS once (the operand ring is dead by now):
  // phase A adds bias (+ addvec), rounds to bf16 and writes 8-byte pieces into the wave's own
  // [WTM][WTN] region; phase B reads 16-byte chunks back row-major, applies the activation / gate /
  // residual on 8 consecutive columns and stores 16 B per lane = whole 128-byte lines per row.
  // Every fused epilogue starts from the bf16-rounded (acc + bias), so the LDS round trip is exact.
  // The direct path remains for float32 outputs and for operands that are not 16-byte aligned.
  // FLAG_LEAN kernels may also fix their ONE epilogue at compile time (FLAGS bits 8-11 = epilogue code + 1): every
  // `epi ==` below folds and the kernel carries a single store path
  if constexpr (X3) {
    // FLAG_SPLIT epilogue: v = alpha * acc + bias (float32 bias, by column or by row) [+ residual (hi + lo planes)],
    // then either float32 out (attention logits) or the hi / lo bf16 planes of v.  Direct stores from the MFMA
    // layout (8 bytes per lane and plane): the 3-pass main loop is three times as long as the bf16 one, so the
    // store tail weighs a third of what it does there.
    const float* fbias = (const float*)gBias;
    float gs[NJ], gq[NJ];                          // GroupNorm partials of this lane's column quads (p.gn_ws)
#pragma unroll
    for (int j = 0; j < NJ; ++j) gs[j] = gq[j] = 0.f;
    // Operands first, arithmetic after (round 5).  With the bias and residual loads inside the (i, j) loops behind the row / column
    // range tests, hipcc branched around every load and waited vmcnt(0) before its use: two dependent round trips per fragment,
    // 32 per wave on the 256 x 128 tile = the 10 us (256 x 256: 21 us) this epilogue took per tile in tools/conv_phase_trace.py.
    // Now: the column bias of all NJ fragments in one batch (clamped column, no branch), and per row block the 2 NJ residual
    // loads in one batch (clamped row / column); only the stores are predicated.
    f32x4 bj[NJ];
    bool jok[NJ];
    int n4c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n4 = n0 + wn * WTN + j * 16 + q4 * 4;
      jok[j] = n4 < N;
      n4c[j] = jok[j] ? n4 : 0;
      bj[j] = (fbias && !p.row_bias) ? *(const f32x4*)(fbias + n4c[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float rbi[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rbi[i] = (fbias && p.row_bias) ? fbias[min(m0 + wm * WTM + i * 16 + r16, Mg - 1)] : 0.f;
    const bool with_res = !p.out_f32 && epi == EPI_GATE_RES;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WTM + i * 16 + r16;
      const bool iok = m < Mg;
      const int mc = iok ? m : Mg - 1;
      long long orow = mc;
      if (AMODE == 1 && p.cv.sub2) {               // sub-pixel conv: (b, y, x) -> (b, 2y + sdy, 2x + sdx)
        const int hw = p.cv.Ho * p.cv.Wo;
        const int bb = mc / hw, rem = mc - bb * hw;
        const int y = rem / p.cv.Wo, x = rem - y * p.cv.Wo;
        orow = ((long long)bb * (2 * p.cv.Ho) + 2 * y + sdy) * (2 * p.cv.Wo) + 2 * x + sdx;
      }
      const long long rowbase = (long long)b * c_bs + orow * p.ldc;
      u32x2 rh[NJ], rl[NJ];
      if (with_res) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          rh[j] = *(const u32x2*)(gRes + rowbase + n4c[j]);
          rl[j] = *(const u32x2*)(gRes + rowbase + n4c[j] + p.res_lo);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 v = acc[i][j] * alpha;
        v += bj[j];
        v += rbi[i];
        const long long idx = rowbase + n4c[j];
        const bool ok = iok && jok[j];
        if (p.out_f32) {
          if (ok) *(f32x4*)((float*)gC + idx) = v;
          continue;
        }
        if (with_res) {
          v[0] += e_lo<F16>(rh[j][0]) + e_lo<F16>(rl[j][0]);
          v[1] += e_hi<F16>(rh[j][0]) + e_hi<F16>(rl[j][0]);
          v[2] += e_lo<F16>(rh[j][1]) + e_lo<F16>(rl[j][1]);
          v[3] += e_hi<F16>(rh[j][1]) + e_hi<F16>(rl[j][1]);
        }
        u32x2 oh, ol;
        oh[0] = e_pack<F16>(v[0], v[1]);
        oh[1] = e_pack<F16>(v[2], v[3]);
        ol[0] = e_pack<F16>(v[0] - e_lo<F16>(oh[0]), v[1] - e_hi<F16>(oh[0]));
        ol[1] = e_pack<F16>(v[2] - e_lo<F16>(oh[1]), v[3] - e_hi<F16>(oh[1]));
        if (ok) {
          *(u32x2*)(gC + idx) = oh;
          *(u32x2*)(gC + idx + p.c_lo) = ol;
          gs[j] += (v[0] + v[1]) + (v[2] + v[3]);
          gq[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
      }
    }
    if (AMODE == 1 && p.gn_ws) {
      // reduce over the 16 rows a 16-lane group holds (and the MI row blocks summed above): fixed order, deterministic
      const int img = m0 / p.gn_hw;
      const int rb = (m0 - img * p.gn_hw) / WTM + wm + (p.cv.sub2 ? b * (p.gn_hw / WTM) : 0);
      float* dst = p.gn_ws + (((long long)img * p.gn_nchunks + rb) * N) * 2;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float s_ = row16_sum(gs[j]), q_ = row16_sum(gq[j]);
        const int n4 = n0 + wn * WTN + j * 16 + q4 * 4;
        if (r16 == 0 && n4 < N) *(float2*)(dst + (long long)n4 * 2) = float2{s_, q_};
      }
    }
    if constexpr ((FLAGS & FLAG_TIMED) != 0) {       // diagnostic fp32-faithful conv tiles (tools/conv_phase_trace.py)
      unsigned long long t_end, rt_end;
      asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t_end), "=s"(rt_end)::"memory");
      if (p.trace && lane == 0) {
        unsigned long long* t = p.trace + ((long long)blockIdx.x * NWAVES + wave) * 16;
        t[8] = t_end - t_start;
        t[9] = rt_end - rt_start;
        t[10] = t_loop0 - t_start;
        t[11] = t_end - t_loop1;
      }
    }
    return;
  }
  constexpr int NCH = WTN / 8;                      // 16-byte chunks per row of the wave's sub-tile
  const bool wide = LEAN || p.wide_epi != 0;
  char* const my_lds = smem + wave * (WTM * WTN * 2);
  static_assert(BM * BN * 2 <= NSA * A_BYTES + NSW * B_BYTES, "epilogue staging must fit in the operand ring");
  if (wide) __builtin_amdgcn_s_barrier();           // every wave is done reading the last operand slots
  unsigned long long t_epi0 = 0;
  if constexpr ((FLAGS & FLAG_TIMED) != 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_epi0)::"memory");

  if (!bias_loaded) load_bias();
  // (ADDVEC is a compile-time tag so the common path carries no per-tile branch or integer division)
  // (aw_pre: the addend of this fragment already in registers - phase A of the LDS-transposed epilogue requests the NJ addends of a
  //  row block in one batch; as a load inside the fragment loop every one of them was followed by vmcnt(0), see phase_a)
  auto biased = [&](auto addvec_tag, int i, int j, int m, int n4, float (&v)[4], const u32x2* aw_pre = nullptr) {
    if constexpr (F8) {
      v[0] = acc[i][j][0] * (asc[i] * wsc[j][0]) + (e_lo<F16>(bcol[j][0]) + brow[i]);
      v[1] = acc[i][j][1] * (asc[i] * wsc[j][1]) + (e_hi<F16>(bcol[j][0]) + brow[i]);
      v[2] = acc[i][j][2] * (asc[i] * wsc[j][2]) + (e_lo<F16>(bcol[j][1]) + brow[i]);
      v[3] = acc[i][j][3] * (asc[i] * wsc[j][3]) + (e_hi<F16>(bcol[j][1]) + brow[i]);
    } else {
    v[0] = acc[i][j][0] * alpha + (e_lo<F16>(bcol[j][0]) + brow[i]);
    v[1] = acc[i][j][1] * alpha + (e_hi<F16>(bcol[j][0]) + brow[i]);
    v[2] = acc[i][j][2] * alpha + (e_lo<F16>(bcol[j][1]) + brow[i]);
    v[3] = acc[i][j][3] * alpha + (e_hi<F16>(bcol[j][1]) + brow[i]);
    }
    if constexpr (decltype(addvec_tag)::value) {   // per-image vector added after the bias (ResnetBlock2D: + temb[:, None, None, :])
      // ... or a whole matrix addend of this group (the low-rank branch of an unfused LoRA layer): row m of batch b
      u32x2 aw = aw_pre ? *aw_pre
                 : gAddm ? *(const u32x2*)(gAddm + (long long)b * addm_bs + (long long)m * p.addvec_stride + n4)
                         : *(const u32x2*)(p.addvec + (long long)(m / p.addvec_rows) * p.addvec_stride + n4);
      v[0] = e_rnd<F16>(v[0]) + e_lo<F16>(aw[0]);
      v[1] = e_rnd<F16>(v[1]) + e_hi<F16>(aw[0]);
      v[2] = e_rnd<F16>(v[2]) + e_lo<F16>(aw[1]);
      v[3] = e_rnd<F16>(v[3]) + e_hi<F16>(aw[1]);
    }
  };
  // activation / gate / residual on NV consecutive columns of row m starting at column n
  auto finish = [&](float* v, const int NV, int m, int n, bf16_t*& dst) {
    dst = gC + (long long)b * c_bs + (long long)m * p.ldc + n;
    if (epi == EPI_GELU_TANH) {
      for (int r = 0; r < NV; ++r) v[r] = gelu_tanh_f(v[r]);
    } else if (!LEAN && epi == EPI_SILU) {
      for (int r = 0; r < NV; ++r) v[r] = silu_f(v[r]);
    } else if (!LEAN && epi == EPI_QUICK_GELU) {
      for (int r = 0; r < NV; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[r]));
    } else if (!LEAN && epi == EPI_GELU_ERF) {
      for (int r = 0; r < NV; ++r) v[r] = gelu_erf_f(v[r]);
    } else if (epi == EPI_GATE_RES) {
      const uint32_t* rp = (const uint32_t*)(gRes + (long long)b * c_bs + (long long)m * p.ldc + n);
      if (gGate) {
        const uint32_t* gp = (const uint32_t*)(gGate + (long long)b * gate_bs + n);
        for (int r = 0; r < NV; r += 2) {
          const uint32_t rw = rp[r >> 1], gw = gp[r >> 1];
          v[r] = e_lo<F16>(rw) + e_rnd<F16>(e_lo<F16>(gw) * v[r]);
          v[r + 1] = e_hi<F16>(rw) + e_rnd<F16>(e_hi<F16>(gw) * v[r + 1]);
        }
      } else {
        for (int r = 0; r < NV; r += 2) {
          const uint32_t rw = rp[r >> 1];
          v[r] = e_lo<F16>(rw) + v[r];
          v[r + 1] = e_hi<F16>(rw) + v[r + 1];
        }
      }
    } else if (!LEAN && epi == EPI_GEGLU) {   // out = res * gelu_erf(acc + bias)   (y_a * nn.gelu(y_b))
      const uint32_t* rp = (const uint32_t*)(gRes + (long long)b * c_bs + (long long)m * p.ldc + n);
      for (int r = 0; r < NV; r += 2) {
        const uint32_t rw = rp[r >> 1];
        v[r] = e_lo<F16>(rw) * e_rnd<F16>(gelu_erf_f(v[r]));
        v[r + 1] = e_hi<F16>(rw) * e_rnd<F16>(gelu_erf_f(v[r + 1]));
      }
    } else if (epi == EPI_SPLIT_GELU) {
      if (n >= p.n_split) {
        for (int r = 0; r < NV; ++r) v[r] = gelu_tanh_f(v[r]);
        dst = p.C2 + (long long)b * p.c2_bstride + (long long)m * p.ldc2 + (n - p.n_split + p.c2_coloff);
      }
    }
  };

  // chunk swizzle by row: keeps the 16 rows of a phase-A write and the row pairs of a phase-B read on distinct banks
  auto cswz = [](int c, int row) { return (NCH & (NCH - 1)) == 0 ? (c ^ (row & (NCH - 1))) : (c + row) % NCH; };
  if (wide) {
    // phase A: registers -> LDS
    // sub_tag: only the fragments this block owns after a reduce-scatter split-K hand-off (wave-uniform ranges)
    auto phase_a = [&](auto addvec_tag, auto sub_tag) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if constexpr (decltype(sub_tag)::value) {
          if (i < oi_lo || i >= oi_hi) continue;
        }
        const int row = i * 16 + r16;
        const int m = min(m0 + wm * WTM + row, Mg - 1);
        // the addends of this row block (ResnetBlock2D's temb vector / an unfused LoRA branch): one batch of NJ loads, one row
        // decode - inside the fragment loop each load was followed by vmcnt(0) and each fragment repeated the division
        u32x2 awv[decltype(addvec_tag)::value ? NJ : 1];
        if constexpr (decltype(addvec_tag)::value) {
          const bf16_t* const ap = gAddm ? gAddm + (long long)b * addm_bs + (long long)m * p.addvec_stride
                                         : p.addvec + (long long)(m / p.addvec_rows) * p.addvec_stride;
#pragma unroll
          for (int j = 0; j < NJ; ++j) awv[j] = *(const u32x2*)(ap + min(n0 + wn * WTN + j * 16 + q4 * 4, N - 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (decltype(sub_tag)::value) {
            if (j < oj_lo || j >= oj_hi) continue;
          }
          const int n4 = min(n0 + wn * WTN + j * 16 + q4 * 4, N - 4);
          float v[4];
          if constexpr (decltype(addvec_tag)::value) biased(addvec_tag, i, j, m, n4, v, &awv[j]);
          else
          biased(addvec_tag, i, j, m, n4, v);
          u32x2 o;
          o[0] = e_pack<F16>(v[0], v[1]);
          o[1] = e_pack<F16>(v[2], v[3]);
          const int ch = cswz(j * 2 + (q4 >> 1), row);
          *(u32x2*)(my_lds + row * (NCH * 16) + ch * 16 + (q4 & 1) * 8) = o;
        }
      }
    };
    if constexpr (RS_CAPABLE) {
      if (LEAN || rs) phase_a(std::false_type{}, std::true_type{});          // (the launcher keeps addvec GEMMs on the chain)
      else if (p.addvec) phase_a(std::true_type{}, std::false_type{});
      else phase_a(std::false_type{}, std::false_type{});
    } else {
      if (!LEAN && p.addvec) phase_a(std::true_type{}, std::false_type{});
      else phase_a(std::false_type{}, std::false_type{});
    }
    if constexpr ((FLAGS & FLAG_TIMED) != 0) asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_epiA)::"memory");
    if constexpr (LEPI == EPI_GEGLU_PAIR + 1) {
      // phase B, pair form: out chunk oc of a row = value chunk (4 f + h) * gelu_erf(gate chunk (4 f + 2 + h)), f = oc / 2, h = oc & 1
      static_assert(NJ % 2 == 0, "GEGLU pair epilogue: an even number of 16-column fragments per wave");
      constexpr int NOC = NCH / 2;
      constexpr int NITP = (WTM * NOC + 63) / 64;
      const int nhalf = N >> 1;
#pragma unroll 4
      for (int t = 0; t < NITP; ++t) {
        const int idx = t * 64 + lane;
        const int row = idx / NOC, oc = idx - row * NOC;
        const int ca = (oc >> 1) * 4 + (oc & 1);
        const int m = m0 + wm * WTM + row, nout = ((n0 + wn * WTN) >> 1) + (oc >> 1) * 16 + (oc & 1) * 8;
        if (row >= WTM || m >= Mg || nout >= nhalf) continue;
        const u32x4 xa = *(const u32x4*)(my_lds + row * (NCH * 16) + cswz(ca, row) * 16);
        const u32x4 xg = *(const u32x4*)(my_lds + row * (NCH * 16) + cswz(ca + 2, row) * 16);
        u32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = e_pack<F16>(e_lo<F16>(xa[r]) * e_rnd<F16>(gelu_erf_f(e_lo<F16>(xg[r]))), e_hi<F16>(xa[r]) * e_rnd<F16>(gelu_erf_f(e_hi<F16>(xg[r]))));
        *(u32x4*)(gC + (long long)b * c_bs + (long long)m * p.ldc + nout) = o;
      }
    } else {
    // phase B: LDS -> 8 consecutive columns per lane -> global
    constexpr int NIT = (WTM * NCH + 63) / 64;
    // reduce-scatter split-K: the owned rows [r_lo, r_lo + nr) x chunks [c_lo, c_lo + ncw) of the wave's sub-tile only
    const int r_lo = oi_lo * 16, nr = (oi_hi - oi_lo) * 16, c_lo = oj_lo * 2, ncw = (oj_hi - oj_lo) * 2;
    const int nit = rs ? (nr * ncw + 63) >> 6 : NIT;
    if (!MXC && epi == EPI_GATE_RES) {
      // gate-residual form: four items per lane and trip, operands requested together (gr_load above the split-K hand-off)
      const bool gated = gGate != nullptr;
      for (int t0 = 0; t0 < nit; t0 += 4) {
        GrItems cur;
        if (t0 == 0 && gr_prefetched) cur = gr_pf;
        else gr_load(cur, t0, nit, RS_CAPABLE && rs, r_lo, nr, c_lo, ncw);
        u32x4 xs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xs[u] = *(const u32x4*)(my_lds + cur.rowc[u] * (NCH * 16) + cswz(cur.cc[u], cur.rowc[u]) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!cur.ok[u]) continue;
          u32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x0 = e_lo<F16>(xs[u][r]), x1 = e_hi<F16>(xs[u][r]);
            float y0, y1;
            if (gated) {
              y0 = e_lo<F16>(cur.rv[u][r]) + e_rnd<F16>(e_lo<F16>(cur.gv[u][r]) * x0);
              y1 = e_hi<F16>(cur.rv[u][r]) + e_rnd<F16>(e_hi<F16>(cur.gv[u][r]) * x1);
            } else {
              y0 = e_lo<F16>(cur.rv[u][r]) + x0;
              y1 = e_hi<F16>(cur.rv[u][r]) + x1;
            }
            o[r] = e_pack<F16>(y0, y1);
          }
          *(u32x4*)(gC + (long long)b * c_bs + cur.off[u]) = o;
        }
      }
    } else
#pragma unroll 4
    for (int t = 0; t < nit; ++t) {
      const int idx = t * 64 + lane;
      int row, c;
      if (RS_CAPABLE && rs) {
        const int rr = (int)((unsigned)idx / (unsigned)ncw);
        row = r_lo + rr;
        c = c_lo + idx - rr * ncw;
        if (rr >= nr) continue;
      } else {
        row = idx / NCH;
        c = idx - row * NCH;
      }
      const int m = m0 + wm * WTM + row, n8 = n0 + wn * WTN + c * 8;
      if (row >= WTM || m >= Mg || n8 >= N) continue;
      const u32x4 x = *(const u32x4*)(my_lds + row * (NCH * 16) + cswz(c, row) * 16);
      if constexpr (MXC) {
        // FLAG_MXC: GELU, then e4m3 with one E8M0 scale per 32 consecutive columns = the 4 lanes of a quad (idx, n8 and the
        // tests above are quad-uniform: N, n_split and the column offsets are multiples of 32).  Scale 2^e, e the smallest
        // exponent with max|v| / 2^e <= 448 (the e4m3 maximum): no element saturates, the conversion rounds to nearest even.
        const bool gelu_half = epi == EPI_GELU_TANH || (epi == EPI_SPLIT_GELU && n8 >= p.n_split);
        if (gelu_half) {
          static_assert(!MXC || (NCH % 4 == 0 && WTN % 32 == 0 && BN % 32 == 0), "block-scaled output: 32-column blocks inside a wave's sub-tile");
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[2 * r] = gelu_tanh_f(e_lo<F16>(x[r])); v[2 * r + 1] = gelu_tanh_f(e_hi<F16>(x[r])); }
          float am = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) am = fmaxf(am, fabsf(v[r]));
          am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, am), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
          am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, am), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
          const uint32_t ab = __builtin_bit_cast(uint32_t, am);
          int e8 = (int)(ab >> 23) - 8 + (int)((ab & 0x7fffffu) > 0x600000u);      // 448 = 1.75 * 2^8
          e8 = min(max(e8, 1), 253);
          const float mul = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);   // 2^(127 - e8)
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * mul, v[1] * mul, w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * mul, v[3] * mul, w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * mul, v[5] * mul, w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * mul, v[7] * mul, w1, true);
          const int col = epi == EPI_SPLIT_GELU ? n8 - p.n_split + p.c8_coloff : n8;
          *(u32x2*)((g1 ? p.c8[1] : p.c8[0]) + (long long)b * p.c8_bstride + (long long)m * p.ldc8 + col) = u32x2{(uint32_t)w0, (uint32_t)w1};
          if ((lane & 3) == 0) {
            const long long mrow = (g1 ? p.c_mx_row0[1] : p.c_mx_row0[0]) + (long long)b * p.c_mx_bstride + m;
            const int kb = col >> 5;
            p.c_mx[((((long long)(kb >> 2) * p.c_mx_kstride + (mrow >> 6) * 64 + (kb & 3) * 16 + (mrow & 15)) << 2) + ((mrow >> 4) & 3))] = (uint8_t)e8;
          }
          continue;
        }
      }
      bf16_t* dst;
      u32x4 o = x;
      if (epi != EPI_BIAS) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[2 * r] = e_lo<F16>(x[r]); v[2 * r + 1] = e_hi<F16>(x[r]); }
        finish(v, 8, m, n8, dst);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = e_pack<F16>(v[2 * r], v[2 * r + 1]);
      } else {
        dst = gC + (long long)b * c_bs + (long long)m * p.ldc + n8;
      }
      *(u32x4*)dst = o;
    }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WTM + i * 16 + r16;
      if (m >= Mg) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n4 = n0 + wn * WTN + j * 16 + q4 * 4;
        if (n4 >= N) continue;
        float v[4];
        if (p.addvec) biased(std::true_type{}, i, j, m, n4, v);
        else biased(std::false_type{}, i, j, m, n4, v);
        if (p.out_f32) {
          float* fdst = (float*)gC + (long long)b * c_bs + (long long)m * p.ldc + n4;
          *(f32x4*)fdst = f32x4{v[0], v[1], v[2], v[3]};
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = e_rnd<F16>(v[r]);
        bf16_t* dst;
        finish(v, 4, m, n4, dst);
        u32x2 o;
        o[0] = e_pack<F16>(v[0], v[1]);
        o[1] = e_pack<F16>(v[2], v[3]);
        *(u32x2*)dst = o;
      }
    }
  }
  // ---- reduce-scatter split-K, wait-free completion: the last arriver of a tile finishes the slices whose owners gave up
  // waiting (see the hand-off).  Rare path: direct stores from the MFMA layout; the arithmetic is that of the LDS-transposed
  // epilogue (bf16-rounded acc * alpha + bias, then the fused form), the summation order that of the owner.
  if constexpr (RS_CAPABLE) {
    if (rs_orphans != 0) {
      constexpr int F = MI * NJ;
      const bool by_rows = (MI % S) == 0;
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.sk_part + (size_t)bid * ((size_t)S * BM * BN), 0, S * BM * BN * 4, 0x00020000);
      auto slab_ld = [&](int src, int frag) {      // fragment `frag` of this wave in the partial of split `src`: sc1 load (L1 bypass, like the fast path)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, ((src * NWAVES + wave) * F + frag) * 1024 + lane * 16, 0, 16));
      };
      for (int so = 0; so < S; ++so) {
        if (((rs_orphans >> so) & 1) == 0) continue;
        int ai_lo = 0, ai_hi = MI, aj_lo = 0, aj_hi = NJ;
        if (by_rows) { ai_lo = so * (MI / S); ai_hi = ai_lo + MI / S; }
        else { aj_lo = so * (NJ / S); aj_hi = aj_lo + NJ / S; }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (i < ai_lo || i >= ai_hi || j < aj_lo || j >= aj_hi) continue;      // wave-uniform
            f32x4 sum = slab_ld(so, i * NJ + j);
            for (int s2 = 0; s2 < S; ++s2)
              if (s2 != so) sum += slab_ld(s2, i * NJ + j);
            acc[i][j] = sum;
            const int m = m0 + wm * WTM + i * 16 + r16, n4 = n0 + wn * WTN + j * 16 + q4 * 4;
            if (m >= Mg || n4 >= N) continue;
            float v[4];
            biased(std::false_type{}, i, j, m, n4, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = e_rnd<F16>(v[r]);
            bf16_t* dst;
            finish(v, 4, m, n4, dst);
            u32x2 o;
            o[0] = e_pack<F16>(v[0], v[1]);
            o[1] = e_pack<F16>(v[2], v[3]);
            *(u32x2*)dst = o;
          }
      }
    }
  }
  if constexpr ((FLAGS & FLAG_TIMED) != 0) {
    unsigned long long t_end, rt_end;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_epiB)::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t_end), "=s"(rt_end)::"memory");
    if (p.trace && lane == 0) {
      unsigned long long* t = p.trace + ((long long)blockIdx.x * NWAVES + wave) * 16;
      t[8] = t_end - t_start;        // whole wave, shader cycles
      t[9] = rt_end - rt_start;      // whole wave, 100 MHz ticks
      t[10] = t_loop0 - t_start;     // address setup
      t[11] = t_end - t_loop1;       // epilogue
      t[12] = t_epiA - t_loop1;      // epilogue phase A (barrier, bias, LDS writes)
      t[13] = t_epiB - t_epiA;       // epilogue phase B issue (LDS reads, fused math, store issue)
      t[14] = t_epi0 - t_loop1;      // barrier after the main loop (wave skew)
    }
  }
}