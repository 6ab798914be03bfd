// Memory-bound normalisation kernels of the denoise path (HBM roofline, wave-shuffle reductions,
// 16-byte bf16 vector loads; no LDS except for the V transpose tile).
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// adaLN:  out = (1 + scale) * LayerNorm(x) + shift     (flux/layers.py:192-193,202-203,222,228,
// 267,300; nn.LayerNorm(affine=False, eps=1e-6): biased variance, fp32 statistics)
// One wave per token row; NCH = ceil(D / 512) 16-byte chunks per lane kept in registers.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_kernel(
    const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int Tr, int D, int S,
    long long x_bstride, long long out_bstride, const bf16_t* __restrict__ shift_txt,
    const bf16_t* __restrict__ scale_txt, const bf16_t* __restrict__ shift_img,
    const bf16_t* __restrict__ scale_img, long long mod_bstride, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * Tr) return;
  const int b = (int)(row / Tr);
  const int t = (int)(row - (long long)b * Tr);
  const bf16_t* xr = x + b * x_bstride + (long long)t * D;
  bf16_t* orow = out + b * out_bstride + (long long)t * D;
  const bool txt = t < S;
  const bf16_t* sh = (txt ? shift_txt : shift_img) + b * mod_bstride;
  const bf16_t* sc = (txt ? scale_txt : scale_img) + b * mod_bstride;
  const int nchunk = D >> 3;

  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
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
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nchunk) {
      u32x4 shw = *((const u32x4*)sh + c);
      u32x4 scw = *((const u32x4*)sc + c);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // reference op boundaries: LN -> bf16, (1+scale) -> bf16, product -> bf16, + shift -> bf16
        float n0 = rbf((v[i][2 * e] - mean) * rstd);
        float n1 = rbf((v[i][2 * e + 1] - mean) * rstd);
        float y0 = rbf(rbf(1.f + bf_lo(scw[e])) * n0) + bf_lo(shw[e]);
        float y1 = rbf(rbf(1.f + bf_hi(scw[e])) * n1) + bf_hi(shw[e]);
        o[e] = pack_bf16x2(y0, y1);
      }
      *((u32x4*)orow + c) = o;
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
    bf16_t* __restrict__ Kout, bf16_t* __restrict__ Vt, int Tpad, float eps, int n_qk_blocks) {
  __shared__ bf16_t vt_tile[64][130];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < n_qk_blocks) {
    const int sub = tid & 15;                                   // 16-byte chunk inside the row
    const long long row = (long long)blockIdx.x * 16 + (tid >> 4);
    const long long nrows = (long long)B * T * H * 2;
    const bool live = row < nrows;
    const long long rr = live ? row : nrows - 1;
    const int which = (int)(rr & 1);                            // 0 = q, 1 = k
    const long long th = rr >> 1;
    const int h = (int)(th % H);
    const long long bt = th / H;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const bf16_t* src = qkv + bt * ld + (long long)which * H * 128 + h * 128 + sub * 8;
    u32x4 w = *(const u32x4*)src;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bf_lo(w[e]);
      v[2 * e + 1] = bf_hi(w[e]);
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
    const bool txt = t < S;
    const bf16_t* wp = (which ? (txt ? kw_txt : kw_img) : (txt ? qw_txt : qw_img)) + sub * 8;
    u32x4 ww = *(const u32x4*)wp;
    u32x4 cs = *(const u32x4*)(rope + b * rope_bstride + ((long long)t * 64 + sub * 4) * 2);  // 4 x (cos, sin)
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x0 = rbf(v[2 * e] * rstd * bf_lo(ww[e]));      // RMSNorm output is a bf16 tensor
      float x1 = rbf(v[2 * e + 1] * rstd * bf_hi(ww[e]));
      float c = bf_lo(cs[e]), s = bf_hi(cs[e]);
      o[e] = pack_bf16x2(x0 * c - x1 * s, x0 * s + x1 * c);
    }
    if (live) {
      bf16_t* dst = (which ? Kout : Q) + (((long long)b * H + h) * T + t) * 128 + sub * 8;
      *(u32x4*)dst = o;
    }
  } else {
    // V^T tile: blockIdx -> (bh, token tile of 64)
    const int vb = blockIdx.x - n_qk_blocks;
    const int ntt = Tpad >> 6;
    const int tt = vb % ntt;
    const int bh = vb / ntt;
    const int b = bh / H, h = bh - b * H;
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
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = (uint32_t)vt_tile[c * 8 + 2 * e][d] | ((uint32_t)vt_tile[c * 8 + 2 * e + 1][d] << 16);
      bf16_t* dst = Vt + (((long long)b * H + h) * 128 + d) * Tpad + t0 + c * 8;
      *(u32x4*)dst = u32x4{o[0], o[1], o[2], o[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) on NHWC bf16 (nn.GroupNorm(pytorch_compatible=True): G groups of C/G
// contiguous channels, statistics over H*W*(C/G), biased variance).
// Pass 1: per (batch, pixel-chunk) partial sums per group  -> ws[b][chunk][G][2]  (deterministic)
// Pass 2: reduce the partials, normalise, affine, optional SiLU.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x,
                                                         float* __restrict__ ws, int HW, int C,
                                                         int G, int nchunks, int ppb) {
  // block = (chunk, b); thread t handles 16-byte channel chunk (t % (C/8)) over a strided pixel set
  __shared__ float red[256 * 2];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpr = C >> 3;                       // 16-B chunks per pixel
  const int tid = threadIdx.x;
  const int cc = tid % cpr;                     // channel chunk owned (fixed -> fixed group)
  const int prow = tid / cpr, pstep = 256 / cpr;  // requires cpr | 256
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, HW);
  float s = 0.f, q = 0.f;
#pragma unroll 4
  for (int p = p0 + prow; p < p1; p += pstep) {
    u32x4 w = *((const u32x4*)(x + ((long long)b * HW + p) * C) + cc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = bf_lo(w[e]), c = bf_hi(w[e]);
      s += a + c;
      q += a * a + c * c;
    }
  }
  red[tid] = s;
  red[256 + tid] = q;
  __syncthreads();
  // group g covers channel chunks [g*cpg8, (g+1)*cpg8) where cpg8 = (C/G)/8 (>= 1 when C/G >= 8);
  // when C/G < 8 one 16-B chunk spans several groups: handled by the fine path below.
  const int cg = C / G;
  if (cg >= 8) {
    const int cpg8 = cg >> 3;
    if (tid < G) {
      float ts = 0.f, tq = 0.f;
      for (int r = 0; r < pstep; ++r)
        for (int k = 0; k < cpg8; ++k) {
          int idx = r * cpr + tid * cpg8 + k;
          ts += red[idx];
          tq += red[256 + idx];
        }
      float* o = ws + (((long long)b * nchunks + chunk) * G + tid) * 2;
      o[0] = ts;
      o[1] = tq;
    }
  }
}

// fine path for C/G == 4 (C=128, G=32): per-thread chunk holds two groups
__global__ __launch_bounds__(256) void gn_partial_cg4_kernel(const bf16_t* __restrict__ x,
                                                             float* __restrict__ ws, int HW, int C,
                                                             int G, int nchunks, int ppb) {
  __shared__ float red[256 * 4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpr = C >> 3;
  const int tid = threadIdx.x;
  const int cc = tid % cpr;
  const int prow = tid / cpr, pstep = 256 / cpr;
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, HW);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll 4
  for (int p = p0 + prow; p < p1; p += pstep) {
    u32x4 w = *((const u32x4*)(x + ((long long)b * HW + p) * C) + cc);
    float a0 = bf_lo(w[0]), a1 = bf_hi(w[0]), a2 = bf_lo(w[1]), a3 = bf_hi(w[1]);
    float c0 = bf_lo(w[2]), c1 = bf_hi(w[2]), c2 = bf_lo(w[3]), c3 = bf_hi(w[3]);
    s0 += a0 + a1 + a2 + a3;
    q0 += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    s1 += c0 + c1 + c2 + c3;
    q1 += c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
  }
  red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
  __syncthreads();
  if (tid < G) {
    const int ccg = tid >> 1, half = tid & 1;
    float ts = 0.f, tq = 0.f;
    for (int r = 0; r < pstep; ++r) {
      int idx = (r * cpr + ccg) * 4 + half * 2;
      ts += red[idx];
      tq += red[idx + 1];
    }
    float* o = ws + (((long long)b * nchunks + chunk) * G + tid) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x,
                                                       const float* __restrict__ ws,
                                                       const bf16_t* __restrict__ gamma,
                                                       const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ out, int HW, int C,
                                                       int G, int nchunks, float eps, int silu, int ppb) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < G) {   // (mean, rstd) per group, reduced once by gn_finalize_kernel
    const float* st = ws + ((long long)gridDim.y * nchunks * G + (long long)b * G + tid) * 2;
    s_mean[tid] = st[0];
    s_rstd[tid] = st[1];
  }
  __syncthreads();
  const int cpr = C >> 3;
  const int cc = tid % cpr;
  const int prow = tid / cpr, pstep = 256 / cpr;
  const int cg = C / G;
  u32x4 gw = *((const u32x4*)gamma + cc);
  u32x4 bw = *((const u32x4*)beta + cc);
  float mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int g = (cc * 8 + e) / cg;
    mu[e] = s_mean[g];
    rs[e] = s_rstd[g];
  }
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, HW);
  // fold normalisation + affine into one fma per element: y = x * a + c
  float fa[8], fc[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    fa[2 * e] = rs[2 * e] * bf_lo(gw[e]);
    fa[2 * e + 1] = rs[2 * e + 1] * bf_hi(gw[e]);
    fc[2 * e] = bf_lo(bw[e]) - mu[2 * e] * fa[2 * e];
    fc[2 * e + 1] = bf_hi(bw[e]) - mu[2 * e + 1] * fa[2 * e + 1];
  }
  constexpr int U = 4;   // independent 16-B loads in flight per thread
  for (int p = p0 + prow; p < p1; p += pstep * U) {
    u32x4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = min(p + u * pstep, p1 - 1);
      w[u] = *((const u32x4*)(x + ((long long)b * HW + pp) * C) + cc);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * pstep;
      if (pp >= p1) break;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = fmaf(bf_lo(w[u][e]), fa[2 * e], fc[2 * e]);
        float y1 = fmaf(bf_hi(w[u][e]), fa[2 * e + 1], fc[2 * e + 1]);
        if (silu) {
          y0 = silu_f(y0);
          y1 = silu_f(y1);
        }
        o[e] = pack_bf16x2(y0, y1);
      }
      *((u32x4*)(out + ((long long)b * HW + pp) * C) + cc) = o;
    }
  }
}

// one wave per (batch, group): deterministic (fixed-order) reduction of the per-chunk partials
__global__ __launch_bounds__(64) void gn_finalize_kernel(float* __restrict__ ws, int B, int G,
                                                         int nchunks, float cnt, float eps) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int b = i / G, g = i - b * G;
  float ts = 0.f, tq = 0.f;
  for (int k = lane; k < nchunks; k += 64) {
    const float* o = ws + (((long long)b * nchunks + k) * G + g) * 2;
    ts += o[0];
    tq += o[1];
  }
  ts = wave_sum(ts);
  tq = wave_sum(tq);
  if (lane == 0) {
    const float mean = ts / cnt;
    const float var = fmaxf(tq / cnt - mean * mean, 0.f);
    float* st = ws + ((long long)B * nchunks * G + i) * 2;
    st[0] = mean;
    st[1] = rsqrtf(var + eps);
  }
}

}  // namespace

extern "C" int fluxhip_ln_modulate_bf16(const void* x, void* out, int B, int Tr, int D, int S,
                                        int64_t x_bstride, int64_t out_bstride,
                                        const void* shift_txt, const void* scale_txt,
                                        const void* shift_img, const void* scale_img,
                                        int64_t mod_bstride, float eps, void* stream) {
  if (!x || !out || !shift_img || !scale_img || B < 1 || Tr < 1 || D < 8 || D % 8 || D > 4096)
    return FLUXHIP_EINVAL;
  if (S > 0 && (!shift_txt || !scale_txt)) return FLUXHIP_EINVAL;
  const long long rows = (long long)B * Tr;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nch = (D + 511) / 512;
#define LN_LAUNCH(NCH)                                                                            \
  hipLaunchKernelGGL((ln_modulate_kernel<NCH>), grid, block, 0, s, (const bf16_t*)x,              \
                     (bf16_t*)out, B, Tr, D, S, (long long)x_bstride, (long long)out_bstride,     \
                     (const bf16_t*)shift_txt, (const bf16_t*)scale_txt, (const bf16_t*)shift_img, \
                     (const bf16_t*)scale_img, (long long)mod_bstride, eps)
  if (nch <= 1) LN_LAUNCH(1);
  else if (nch <= 2) LN_LAUNCH(2);
  else if (nch <= 4) LN_LAUNCH(4);
  else if (nch <= 6) LN_LAUNCH(6);
  else LN_LAUNCH(8);
#undef LN_LAUNCH
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_qk_norm_rope_bf16(const void* qkv, int ld, int B, int T, int S, int H,
                                         const void* qw_txt, const void* kw_txt, const void* qw_img,
                                         const void* kw_img, const void* rope, int64_t rope_bstride,
                                         void* Q, void* Kout, void* Vt, int Tpad, float eps,
                                         void* stream) {
  if (!qkv || !qw_img || !kw_img || !rope || !Q || !Kout || !Vt) return FLUXHIP_EINVAL;
  if (S > 0 && (!qw_txt || !kw_txt)) return FLUXHIP_EINVAL;
  if (B < 1 || T < 1 || H < 1 || ld % 8 || Tpad % 64 || Tpad < T) return FLUXHIP_EINVAL;
  const long long nrows = (long long)B * T * H * 2;
  const int n_qk = (int)((nrows + 15) / 16);
  const int n_vt = B * H * (Tpad / 64);
  hipLaunchKernelGGL(qk_norm_rope_vt_kernel, dim3(n_qk + n_vt), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qkv, ld, B, T, S, H, (const bf16_t*)qw_txt,
                     (const bf16_t*)kw_txt, (const bf16_t*)qw_img, (const bf16_t*)kw_img,
                     (const bf16_t*)rope, (long long)rope_bstride, (bf16_t*)Q, (bf16_t*)Kout,
                     (bf16_t*)Vt, Tpad, eps, n_qk);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_groupnorm_silu_bf16(const void* x, const void* gamma, const void* beta,
                                           void* out, int B, int HW, int C, int G, float eps,
                                           int silu, void* ws, int64_t ws_bytes, void* stream) {
  if (!x || !gamma || !beta || !out || !ws) return FLUXHIP_EINVAL;
  if (B < 1 || HW < 1 || C % 8 || G < 1 || G > 64 || C % G) return FLUXHIP_EINVAL;
  const int cpr = C / 8;
  if (cpr > 256 || 256 % cpr) return FLUXHIP_EINVAL;
  const int cg = C / G;
  if (!(cg == 4 || (cg >= 8 && cg % 8 == 0))) return FLUXHIP_EINVAL;
  // pixels per block: aim for >= 512 blocks, keep the partial table small
  int ppb = 1024;
  while (ppb > 32 && (long long)B * ((HW + ppb - 1) / ppb) < 512) ppb >>= 1;
  const int nchunks = (HW + ppb - 1) / ppb;
  if (ws_bytes < ((int64_t)B * nchunks * G + (int64_t)B * G) * 2 * (int64_t)sizeof(float))
    return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(nchunks, B), block(256);
  if (cg == 4)
    hipLaunchKernelGGL(gn_partial_cg4_kernel, grid, block, 0, s, (const bf16_t*)x, (float*)ws, HW,
                       C, G, nchunks, ppb);
  else
    hipLaunchKernelGGL(gn_partial_kernel, grid, block, 0, s, (const bf16_t*)x, (float*)ws, HW, C,
                       G, nchunks, ppb);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(64), 0, s, (float*)ws, B, G, nchunks,
                     (float)HW * (float)cg, eps);
  hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, s, (const bf16_t*)x, (const float*)ws,
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)out, HW, C, G, nchunks,
                     eps, silu, ppb);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
