// Memory-bound normalisation kernels of the Flux denoise path (HBM roofline, wave-shuffle reductions,
// 16-byte bf16 vector loads; no LDS except for the V transpose tile). GroupNorm lives in groupnorm.hip.
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// adaLN:  out = (1 + scale) * LayerNorm(x) + shift     (flux/layers.py:192-193,202-203,222,228,
// 267,300; nn.LayerNorm(affine=False, eps=1e-6): biased variance, fp32 statistics)
// WPR waves share one token row (NCH = 16-byte chunks per lane kept in registers, chunk c of a row belongs to thread
// c mod (64 WPR) of the row's group).  WPR = 1: the whole row in one wave (small D).  WPR = 4 (D >= 2048): a 3072-wide
// row is 1.5 chunks per lane instead of 6 — at batch 1 the launch is 1280 rows on 256 CUs, i.e. latency-bound, and
// one-wave-per-row left every CU with 5 waves each walking 18 dependent 16-byte loads and ~700 VALU instructions;
// four waves per row put 20 waves on a CU with a quarter of the serial work each (statistics combined through LDS).
// ---------------------------------------------------------------------------------------------
// wave-wide sum without the LDS crossbar: DPP inside each row of 16 lanes, then the four row totals through SGPRs
DEVINL float wave_sum_dpp(float v) {
  v = row16_sum(v);
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}
DEVINL float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}
DEVINL float wave_max_dpp(float v) {      // v >= 0
  v = row16_max(v);
  const int i = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48))));
}
// sum / max over the WPR waves of a row group (fixed order: deterministic); `red` = one float per wave of the block
template <int WPR, bool MAX = false>
DEVINL float rowgroup_reduce(float v, float* red, int wave) {
  v = MAX ? wave_max_dpp(v) : wave_sum_dpp(v);
  if constexpr (WPR == 1) return v;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  const int w0 = wave & ~(WPR - 1);
  float r = red[w0];
#pragma unroll
  for (int i = 1; i < WPR; ++i) r = MAX ? fmaxf(r, red[w0 + i]) : r + red[w0 + i];
  return r;
}
// Q8: the modulated row is written as OCP e4m3fn bytes with ONE float32 scale per row (max|row| / 448) instead of bf16 —
// the per-token activation quantisation of the fp8 path (fluxhip_gemm_fp8) fused into its producer: the row is already in
// this wave's registers, so the separate quantise pass (read bf16 row, write fp8 row) and its launch disappear.  The values
// are rounded to bf16 first, exactly like the two-kernel sequence, so both give bit-identical bytes and scales.
template <int NCH, int WPR, bool Q8 = false>
__global__ __launch_bounds__(512) void ln_modulate_kernel(
    const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int Tr, int D, int S,
    long long x_bstride, long long out_bstride, const bf16_t* __restrict__ shift_txt,
    const bf16_t* __restrict__ scale_txt, const bf16_t* __restrict__ shift_img,
    const bf16_t* __restrict__ scale_img, long long mod_bstride, float eps, float* __restrict__ row_scale = nullptr) {
  __shared__ float red[3][8];
  constexpr int LPR = 64 * WPR;                                  // lanes (threads) per row
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & (LPR - 1);                      // position inside the row group
  const uint32_t nrows = (uint32_t)B * (uint32_t)Tr;          // < 2^31 (checked by the launcher): 32-bit index math, a
  uint32_t row = blockIdx.x * (blockDim.x / LPR) + threadIdx.x / LPR;   // 64-bit divide costs more than the row's arithmetic
  const bool live = row < nrows;       // a dead row group (last block) walks the last row and stores nothing: no early
  if (!live) row = nrows - 1;          // return, every wave reaches the block barriers of rowgroup_reduce
  const int b = (int)(row / (uint32_t)Tr);
  const int t = (int)(row - (uint32_t)b * (uint32_t)Tr);
  const bf16_t* xr = x + b * x_bstride + (long long)t * D;
  bf16_t* orow = out + b * out_bstride + (long long)t * D;
  const bool txt = t < S;
  const bf16_t* sh = (txt ? shift_txt : shift_img) + b * mod_bstride;
  const bf16_t* sc = (txt ? scale_txt : scale_img) + b * mod_bstride;
  const int nchunk = D >> 3;

  float v[NCH][8];
  float sum = 0.f;
  // shift / scale do not depend on the statistics: request them with the row so one latency covers both
  u32x4 shw[NCH], scw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min(lane + i * LPR, nchunk - 1);
    shw[i] = *((const u32x4*)sh + c);
    scw[i] = *((const u32x4*)sc + c);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * LPR;
    if (c < nchunk) {
      u32x4 w = *((const u32x4*)xr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = bf_lo(w[e]);
        v[i][2 * e + 1] = bf_hi(w[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[i][e];
  }
  const float mean = rowgroup_reduce<WPR>(sum, red[0], wave) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * LPR;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(rowgroup_reduce<WPR>(sq, red[1], wave) / (float)D + eps);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * LPR;
    if (c < nchunk) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // reference op boundaries: LN -> bf16, (1+scale) -> bf16, product -> bf16, + shift -> bf16
        float n0 = rbf((v[i][2 * e] - mean) * rstd);
        float n1 = rbf((v[i][2 * e + 1] - mean) * rstd);
        float y0 = rbf(rbf(1.f + bf_lo(scw[i][e])) * n0) + bf_lo(shw[i][e]);
        float y1 = rbf(rbf(1.f + bf_hi(scw[i][e])) * n1) + bf_hi(shw[i][e]);
        o[e] = pack_bf16x2(y0, y1);
        if constexpr (Q8) {
          v[i][2 * e] = bf_lo(o[e]);                 // the bf16-rounded outputs replace the inputs in registers
          v[i][2 * e + 1] = bf_hi(o[e]);
          amax = fmaxf(amax, fmaxf(fabsf(v[i][2 * e]), fabsf(v[i][2 * e + 1])));
        }
      }
      if constexpr (!Q8) {
        if (live) *((u32x4*)orow + c) = o;
      }
    }
  }
  if constexpr (Q8) {
    amax = rowgroup_reduce<WPR, true>(amax, red[2], wave);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0 && live) row_scale[row] = sc;
    uint8_t* qrow = (uint8_t*)out + (b * out_bstride + (long long)t * D);      // strides in ELEMENTS = bytes here
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + i * LPR;
      if (c < nchunk && live) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, w1, true);
        *(u32x2*)(qrow + c * 8) = u32x2{(uint32_t)w0, (uint32_t)w1};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// QKNorm (RMSNorm(128), eps 1e-5, learned scale; flux/layers.py:88-95) + RoPE
// (flux/layers.py:29-33) for q and k, head-major output; plus V -> V^T tiles.
// 16 lanes own one (token, head, q|k) row of 128: 16 B per lane = 4 rotation pairs.
// Blocks >= n_qk_blocks transpose V instead (64 tokens x 128 dims per block through LDS).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_norm_rope_vt_kernel(
    const bf16_t* __restrict__ qkv, int ld, int B, int T, int S, int H,
    const bf16_t* __restrict__ qw_txt, const bf16_t* __restrict__ kw_txt,
    const bf16_t* __restrict__ qw_img, const bf16_t* __restrict__ kw_img,
    const bf16_t* __restrict__ rope, long long rope_bstride, bf16_t* __restrict__ Q,
    bf16_t* __restrict__ Kout, bf16_t* __restrict__ Vt, int Tpad, float eps, int n_vt_blocks, int blocks_per_token) {
  __shared__ bf16_t vt_tile[64][130];
  const int tid = threadIdx.x;
  // The V^T blocks (LDS transpose: the longer ones) come FIRST in the grid so they never form the tail of the launch.
  // All index arithmetic is 32-bit and per BLOCK (uniform): a q / k block is 16 of the 2H rows of ONE token.
  if ((int)blockIdx.x >= n_vt_blocks) {
    const uint32_t q = blockIdx.x - (uint32_t)n_vt_blocks;
    const uint32_t bt = q / (uint32_t)blocks_per_token;         // token index b * T + t
    const int part = (int)(q - bt * (uint32_t)blocks_per_token);
    const int b = (int)(bt / (uint32_t)T);
    const int t = (int)(bt - (uint32_t)b * (uint32_t)T);
    const int sub = tid & 15;                                   // 16-byte chunk inside the row
    const int r = part * 16 + (tid >> 4);                       // row of this token: 2 h + which
    const bool live = r < 2 * H;
    const int rr = live ? r : 2 * H - 1;
    const int which = rr & 1;                                   // 0 = q, 1 = k
    const int h = rr >> 1;
    const bf16_t* src = qkv + (long long)bt * ld + which * H * 128 + h * 128 + sub * 8;
    u32x4 w = *(const u32x4*)src;
    const bool txt = t < S;
    const bf16_t* wp = (which ? (txt ? kw_txt : kw_img) : (txt ? qw_txt : qw_img)) + sub * 8;
    u32x4 ww = *(const u32x4*)wp;
    u32x4 cs = *(const u32x4*)(rope + b * rope_bstride + ((long long)t * 64 + sub * 4) * 2);  // 4 x (cos, sin)
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bf_lo(w[e]);
      v[2 * e + 1] = bf_hi(w[e]);
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    ss = row16_sum(ss);                                         // the 16 lanes of a row are one DPP row
    const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x0 = rbf(v[2 * e] * rstd * bf_lo(ww[e]));      // RMSNorm output is a bf16 tensor
      float x1 = rbf(v[2 * e + 1] * rstd * bf_hi(ww[e]));
      float c = bf_lo(cs[e]), s = bf_hi(cs[e]);
      // Plain (unpacked) VALU ops, opaque to the compiler's packed-FP32 formation.  Written as C, hipcc turned this rotation
      // into v_pk_mul_f32 / v_pk_fma_f32 sequences whose LOW results came out wrong in lanes 48-63 (the K rows of the odd heads)
      // whenever another process shared the GPU - bit-stable alone, different on every run next to a co-tenant
      // (tools/flux_contention_bisect.py located it: this launch, this statement; tools/contention_platform_probe.py shows
      // the co-tenant need not be this library).  Two rewrites that stayed packed failed the same way; these four
      // instructions per pair do not.  The library is also built with packed FP32 formation off (build.sh).
      float t0, t1, r0, r1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(s), "v"(x1));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(s), "v"(x0));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(r0) : "v"(c), "v"(x0), "v"(t0));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(c), "v"(x1), "v"(t1));
      o[e] = pack_bf16x2(r0, r1);
    }
    if (live) {
      bf16_t* dst = (which ? Kout : Q) + (((long long)b * H + h) * T + t) * 128 + sub * 8;
      *(u32x4*)dst = o;
    }
  } else {
    // V^T tile: blockIdx -> (bh, token tile of 64)
    const uint32_t vb = blockIdx.x;
    const uint32_t ntt = (uint32_t)Tpad >> 6;
    const uint32_t bh = vb / ntt;
    const int tt = (int)(vb - bh * ntt);
    const int b = (int)(bh / (uint32_t)H), h = (int)(bh - (uint32_t)b * (uint32_t)H);
    const int t0 = tt * 64;
    // load: 64 rows x 16 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + i * 256;
      int r = idx >> 4, c = idx & 15;
      int t = t0 + r;
      u32x4 w = u32x4{0, 0, 0, 0};
      if (t < T)
        w = *(const u32x4*)(qkv + ((long long)b * T + t) * ld + 2LL * H * 128 + h * 128 + c * 8);
      uint32_t* d = (uint32_t*)&vt_tile[r][c * 8];
      d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    __syncthreads();
    // store: 128 d-rows x 8 chunks of 8 tokens
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + i * 256;
      int d = idx >> 3, c = idx & 7;
      uint32_t o[4];
      // key permutation inside every aligned group of 16 keys: stored order [0-3, 8-11, 4-7, 12-15] (what one PV
      // MFMA fragment of the attention kernel consumes is then one 16-byte chunk; see attn_kernel, VP)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pp = (c & 1) * 8 + 2 * e;                      // stored position inside the 16-key group (even)
        const int pg = pp >> 2, g = ((pg & 1) << 1) | (pg >> 1);  // 4-key sub-group stored there
        const int tk = (c >> 1) * 16 + g * 4 + (pp & 3);          // token (inside the 64-token tile) of that position
        o[e] = (uint32_t)vt_tile[tk][d] | ((uint32_t)vt_tile[tk + 1][d] << 16);
      }
      bf16_t* dst = Vt + (((long long)b * H + h) * 128 + d) * Tpad + t0 + c * 8;
      *(u32x4*)dst = u32x4{o[0], o[1], o[2], o[3]};
    }
  }
}

}  // namespace

static int ln_modulate_launch(const void* x, void* out, float* row_scale, int B, int Tr, int D, int S,
                              int64_t x_bstride, int64_t out_bstride, const void* shift_txt, const void* scale_txt,
                              const void* shift_img, const void* scale_img, int64_t mod_bstride, float eps, void* stream) {
  if (!x || !out || !shift_img || !scale_img || B < 1 || Tr < 1 || D < 8 || D % 8 || D > 4096)
    return FLUXHIP_EINVAL;
  if (S > 0 && (!shift_txt || !scale_txt)) return FLUXHIP_EINVAL;
  const long long rows = (long long)B * Tr;
  if (rows >= (1LL << 31)) return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid, block;
#define LN_LAUNCH(NCH, WPR)                                                                             \
  do {                                                                                                  \
    if (row_scale)                                                                                      \
      hipLaunchKernelGGL((ln_modulate_kernel<NCH, WPR, true>), grid, block, 0, s, (const bf16_t*)x,     \
                         (bf16_t*)out, B, Tr, D, S, (long long)x_bstride, (long long)out_bstride,       \
                         (const bf16_t*)shift_txt, (const bf16_t*)scale_txt, (const bf16_t*)shift_img,  \
                         (const bf16_t*)scale_img, (long long)mod_bstride, eps, row_scale);             \
    else                                                                                                \
      hipLaunchKernelGGL((ln_modulate_kernel<NCH, WPR, false>), grid, block, 0, s, (const bf16_t*)x,    \
                         (bf16_t*)out, B, Tr, D, S, (long long)x_bstride, (long long)out_bstride,       \
                         (const bf16_t*)shift_txt, (const bf16_t*)scale_txt, (const bf16_t*)shift_img,  \
                         (const bf16_t*)scale_img, (long long)mod_bstride, eps, (float*)nullptr);       \
  } while (0)
  if (D >= 2048 && rows >= 8192) {
    // many rows (the fp8 configuration at batch 4: 17408; at 4608 rows the row-per-workgroup form is level): ONE wave per row, four rows per workgroup - no
    // block barrier between the three reductions of a row; with a row per workgroup the launch was a queue of 17408
    // latency-bound workgroups (56 us for 160 MB at C5's shape = 2.8 TB/s)
    grid = dim3((unsigned)((rows + 3) / 4)); block = dim3(256);
    if (D <= 3072) LN_LAUNCH(6, 1);
    else LN_LAUNCH(8, 1);
  } else if (D >= 2048) {
    // four waves per row, one row per 256-thread workgroup: 1280 rows -> 1280 workgroups, 20 waves per CU
    grid = dim3((unsigned)rows); block = dim3(256);
    LN_LAUNCH(2, 4);
  } else {
    // one wave per row; waves per workgroup chosen so that the grid is ~one workgroup per CU
    int wpb = (int)((rows + 255) / 256);
    wpb = wpb < 1 ? 1 : wpb > 8 ? 8 : wpb;
    grid = dim3((unsigned)((rows + wpb - 1) / wpb)); block = dim3(wpb * 64);
    const int nch = (D + 511) / 512;
    if (nch <= 1) LN_LAUNCH(1, 1);
    else if (nch <= 2) LN_LAUNCH(2, 1);
    else LN_LAUNCH(4, 1);
  }
#undef LN_LAUNCH
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_ln_modulate_bf16(const void* x, void* out, int B, int Tr, int D, int S,
                                        int64_t x_bstride, int64_t out_bstride,
                                        const void* shift_txt, const void* scale_txt,
                                        const void* shift_img, const void* scale_img,
                                        int64_t mod_bstride, float eps, void* stream) {
  return ln_modulate_launch(x, out, nullptr, B, Tr, D, S, x_bstride, out_bstride, shift_txt, scale_txt, shift_img,
                            scale_img, mod_bstride, eps, stream);
}

extern "C" int fluxhip_ln_modulate_fp8(const void* x, void* out, void* row_scale, int B, int Tr, int D, int S,
                                       int64_t x_bstride, int64_t out_bstride, const void* shift_txt,
                                       const void* scale_txt, const void* shift_img, const void* scale_img,
                                       int64_t mod_bstride, float eps, void* stream) {
  if (!row_scale || D % 16) return FLUXHIP_EINVAL;
  return ln_modulate_launch(x, out, (float*)row_scale, B, Tr, D, S, x_bstride, out_bstride, shift_txt, scale_txt,
                            shift_img, scale_img, mod_bstride, eps, stream);
}

extern "C" int fluxhip_qk_norm_rope_bf16(const void* qkv, int ld, int B, int T, int S, int H,
                                         const void* qw_txt, const void* kw_txt, const void* qw_img,
                                         const void* kw_img, const void* rope, int64_t rope_bstride,
                                         void* Q, void* Kout, void* Vt, int Tpad, float eps,
                                         void* stream) {
  if (!qkv || !qw_img || !kw_img || !rope || !Q || !Kout || !Vt) return FLUXHIP_EINVAL;
  if (S > 0 && (!qw_txt || !kw_txt)) return FLUXHIP_EINVAL;
  if (B < 1 || T < 1 || H < 1 || ld % 8 || Tpad % 64 || Tpad < T) return FLUXHIP_EINVAL;
  const int bpt = (2 * H + 15) / 16;                      // q / k blocks per token (16 rows of 128 each)
  const long long n_qk = (long long)B * T * bpt;
  const long long n_vt = (long long)B * H * (Tpad / 64);
  if (n_qk + n_vt >= (1LL << 31)) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(qk_norm_rope_vt_kernel, dim3((unsigned)(n_qk + n_vt)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qkv, ld, B, T, S, H, (const bf16_t*)qw_txt,
                     (const bf16_t*)kw_txt, (const bf16_t*)qw_img, (const bf16_t*)kw_img,
                     (const bf16_t*)rope, (long long)rope_bstride, (bf16_t*)Q, (bf16_t*)Kout,
                     (bf16_t*)Vt, Tpad, eps, (int)n_vt, bpt);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
