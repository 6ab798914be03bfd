// GroupNorm (+SiLU) on NHWC bf16 — nn.GroupNorm(G, C, eps, affine, pytorch_compatible=True):
// G groups of C/G contiguous channels, statistics over H*W*(C/G), biased variance, fp32 statistics.
// Reference call sites: flux/autoencoder.py:29-35,62-78,266 (eps 1e-6) and
// stable_diffusion/.../unet.py:98,139,145,391, vae.py:19,137,204 (eps 1e-5, C/G = 10,20,30,40,...).
//
// HBM-bound, three launches, deterministic (no atomics):
//   1. gn_partial : per (batch, pixel chunk) PER-CHANNEL (sum, sumsq) -> ws[b][chunk][C][2]
//                   (per-channel partials make any C/G work, incl. groups that straddle 16-B chunks)
//   2. gn_finalize: one block per (batch, group): fixed-order reduction -> (mean, rstd)
//   3. gn_apply   : y = x*a + c per element (a, c fold mean/rstd/gamma/beta), optional SiLU
// Threads own a fixed 16-byte channel chunk and stride over pixels with 4 loads in flight.
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

// 8 channels of one pixel as float32: a bf16 chunk, or (SPLIT) the sum of the hi and lo plane chunks
// (H: the 16-bit storage type is IEEE float16 instead of bfloat16; never together with SPLIT)
template <bool SPLIT, bool H = false>
DEVINL void load8(const bf16_t* p, long long lo_off, float (&v)[8]) {
  const u32x4 h = *(const u32x4*)p;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = e_lo<H>(h[e]);
    v[2 * e + 1] = e_hi<H>(h[e]);
  }
  if constexpr (SPLIT) {
    const u32x4 l = *(const u32x4*)(p + lo_off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] += bf_lo(l[e]);
      v[2 * e + 1] += bf_hi(l[e]);
    }
  }
}

// CB = channel chunks (of 8 channels) handled per block along blockIdx.z; 256 % CB == 0
template <int CB, bool SPLIT = false, bool H = false>
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x,
                                                         float* __restrict__ ws, int HW, int C,
                                                         int nchunks, int ppb, long long x_lo = 0) {
  constexpr int ROWS = 256 / CB;
  __shared__ float red[256][17];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int cc = blockIdx.z * CB + (tid % CB);   // 16-B channel chunk owned by this thread
  const int prow = tid / CB;
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, HW);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  constexpr int U = 4;
  for (int p = p0 + prow; p < p1; p += ROWS * U) {
    float w[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * ROWS;
      if (pp < p1) {
        load8<SPLIT, H>(x + ((long long)b * HW + pp) * C + cc * 8, x_lo, w[u]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[u][e] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += w[u][e];
        q[e] += w[u][e] * w[u][e];
      }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[tid][e] = s[e];
    red[tid][8 + e] = q[e];
  }
  __syncthreads();
  // CB*16 (channel, stat) pairs, each summed over ROWS pixel rows in a fixed order
  for (int i = tid; i < CB * 16; i += 256) {
    const int c8 = i >> 4, k = i & 15;
    float t = 0.f;
    for (int r = 0; r < ROWS; ++r) t += red[r * CB + c8][k];
    const int ch = (blockIdx.z * CB + c8) * 8 + (k & 7);
    ws[((((long long)b * nchunks + chunk) * C) + ch) * 2 + (k >> 3)] = t;
  }
}

// cstep: 1 = one partial per channel (gn_partial_kernel); 4 = one partial per aligned channel quad, stored on the quad's
// first channel (the conv epilogue's tables, gemm_core.h GemmParams::gn_ws; needs (C / G) % 4 == 0)
__global__ __launch_bounds__(256) void gn_finalize_kernel(float* __restrict__ ws, int B, int C, int G,
                                                          int nchunks, float cnt, float eps, int cstep = 1) {
  // one block per (batch, group); fixed thread -> item assignment and a fixed reduction tree: deterministic
  __shared__ float red[2][4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const int b = i / G, g = i - b * G;
  const int cg = C / G, cq = cg / cstep;
  float ts = 0.f, tq = 0.f;
  const int items = nchunks * cq;
  constexpr int U = 4;
  for (int k0 = tid; k0 < items; k0 += 256 * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 256;
      const int chunk = k / cq, c = g * cg + (k - chunk * cq) * cstep;
      v[u] = k < items ? *(const float2*)(ws + (((long long)b * nchunks + chunk) * C + c) * 2) : float2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ts += v[u].x;
      tq += v[u].y;
    }
  }
  ts = wave_sum(ts);
  tq = wave_sum(tq);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ts;
    red[1][tid >> 6] = tq;
  }
  __syncthreads();
  if (tid == 0) {
    ts = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    tq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float mean = ts / cnt;
    const float var = fmaxf(tq / cnt - mean * mean, 0.f);
    float* st = ws + ((long long)B * nchunks * C + i) * 2;
    st[0] = mean;
    st[1] = rsqrtf(var + eps);
  }
}

// SPLIT: x / out are hi+lo plane pairs, gamma / beta are float32
template <int CB, bool SPLIT = false, bool H = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x,
                                                       const float* __restrict__ ws,
                                                       const bf16_t* __restrict__ gamma,
                                                       const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ out, int HW, int C, int G,
                                                       int nchunks, int silu, int ppb, long long x_lo = 0,
                                                       long long out_lo = 0, int stat_chunks = 0) {
  constexpr int ROWS = 256 / CB;
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < G) {
    // the statistics follow the partial table, whose chunk count differs from this kernel's pixel partition when
    // the partials came out of the producing conv's epilogue (stat_chunks)
    const float* st = ws + ((long long)gridDim.y * (stat_chunks ? stat_chunks : nchunks) * C + (long long)b * G + tid) * 2;
    s_mean[tid] = st[0];
    s_rstd[tid] = st[1];
  }
  __syncthreads();
  const int cc = blockIdx.z * CB + (tid % CB);
  const int prow = tid / CB;
  const int cg = C / G;
  float gaf[8], bef[8];
  if constexpr (SPLIT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gaf[e] = ((const float*)gamma)[cc * 8 + e];
      bef[e] = ((const float*)beta)[cc * 8 + e];
    }
  } else {
    const u32x4 gw = *((const u32x4*)gamma + cc);
    const u32x4 bw = *((const u32x4*)beta + cc);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gaf[e] = (e & 1) ? e_hi<H>(gw[e >> 1]) : e_lo<H>(gw[e >> 1]);
      bef[e] = (e & 1) ? e_hi<H>(bw[e >> 1]) : e_lo<H>(bw[e >> 1]);
    }
  }
  float fa[8], fc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (cc * 8 + e) / cg;
    fa[e] = s_rstd[g] * gaf[e];
    fc[e] = bef[e] - s_mean[g] * fa[e];
  }
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, HW);
  constexpr int U = 4;
  for (int p = p0 + prow; p < p1; p += ROWS * U) {
    float w[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = min(p + u * ROWS, p1 - 1);
      load8<SPLIT, H>(x + ((long long)b * HW + pp) * C + cc * 8, x_lo, w[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * ROWS;
      if (pp >= p1) break;
      u32x4 o, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = fmaf(w[u][2 * e], fa[2 * e], fc[2 * e]);
        float y1 = fmaf(w[u][2 * e + 1], fa[2 * e + 1], fc[2 * e + 1]);
        if (silu) {
          // (round 5: the fp32-faithful path shares the exp2 / rcp form: libm's expf + an IEEE division were ~25 VALU
          //  instructions per element against 5 - the apply pass is 8 values per lane per 32 bytes moved and was VALU-bound, not
          //  HBM-bound.  v_exp_f32 / v_rcp_f32 are 1 ulp each, |y| log2(e) rounds once more: <= 1e-6 relative for |y| <= 16, an
          //  order of magnitude inside the 2^-16 the dropped lo x lo products already cost the convolutions around it.)
          y0 = silu_f(y0);
          y1 = silu_f(y1);
        }
        o[e] = e_pack<H>(y0, y1);
        if constexpr (SPLIT) ol[e] = pack_bf16x2(y0 - bf_lo(o[e]), y1 - bf_hi(o[e]));
      }
      bf16_t* dst = out + ((long long)b * HW + pp) * C + cc * 8;
      *(u32x4*)dst = o;
      if constexpr (SPLIT) *(u32x4*)(dst + out_lo) = ol;
    }
  }
}

}  // namespace

template <bool H>
static int run_groupnorm_silu(const void* x, const void* gamma, const void* beta,
                              void* out, int B, int HW, int C, int G, float eps,
                              int silu, void* ws, int64_t ws_bytes, void* stream) {
  if (!x || !gamma || !beta || !out || !ws) return FLUXHIP_EINVAL;
  if (B < 1 || HW < 1 || C % 8 || G < 1 || G > 64 || C % G) return FLUXHIP_EINVAL;
  const int cpr = C / 8;
  const int cb = cpr % 64 == 0 ? 64 : cpr % 32 == 0 ? 32 : cpr % 16 == 0 ? 16 : cpr % 8 == 0 ? 8 : 0;
  if (!cb) return FLUXHIP_EINVAL;   // C must be a multiple of 64
  // pixels per block: aim for >= 512 blocks, keep the partial table small
  int ppb = 1024;
  while (ppb > 32 && (long long)B * ((HW + ppb - 1) / ppb) * (cpr / cb) < 512) ppb >>= 1;
  const int nchunks = (HW + ppb - 1) / ppb;
  if (ws_bytes < ((int64_t)B * nchunks * C + (int64_t)B * G) * 2 * (int64_t)sizeof(float))
    return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(nchunks, B, cpr / cb), block(256);
#define GN_RUN(CB)                                                                                   \
  do {                                                                                               \
    hipLaunchKernelGGL((gn_partial_kernel<CB, false, H>), grid, block, 0, s, (const bf16_t*)x, (float*)ws, HW, \
                       C, nchunks, ppb);                                                             \
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(256), 0, s, (float*)ws, B, C, G,         \
                       nchunks, (float)HW * (float)(C / G), eps);                                    \
    hipLaunchKernelGGL((gn_apply_kernel<CB, false, H>), grid, block, 0, s, (const bf16_t*)x, (const float*)ws, \
                       (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)out, HW, C, G, nchunks,   \
                       silu, ppb);                                                                   \
  } while (0)
  if (cb == 64) GN_RUN(64);
  else if (cb == 32) GN_RUN(32);
  else if (cb == 16) GN_RUN(16);
  else GN_RUN(8);
#undef GN_RUN
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_groupnorm_silu_bf16(const void* x, const void* gamma, const void* beta,
                                           void* out, int B, int HW, int C, int G, float eps,
                                           int silu, void* ws, int64_t ws_bytes, void* stream) {
  return run_groupnorm_silu<false>(x, gamma, beta, out, B, HW, C, G, eps, silu, ws, ws_bytes, stream);
}
// float16 storage (x, gamma, beta, out are IEEE half; statistics float32 as before)
extern "C" int fluxhip_groupnorm_silu_f16(const void* x, const void* gamma, const void* beta,
                                          void* out, int B, int HW, int C, int G, float eps,
                                          int silu, void* ws, int64_t ws_bytes, void* stream) {
  return run_groupnorm_silu<true>(x, gamma, beta, out, B, HW, C, G, eps, silu, ws, ws_bytes, stream);
}

extern "C" int fluxhip_groupnorm_silu_x3(const void* x, int64_t x_lo, const void* gamma, const void* beta,
                                         void* out, int64_t out_lo, int B, int HW, int C, int G, float eps,
                                         int silu, void* ws, int64_t ws_bytes, void* stream) {
  if (!x || !gamma || !beta || !out || !ws) return FLUXHIP_EINVAL;
  if (B < 1 || HW < 1 || C % 8 || G < 1 || G > 64 || C % G || (x_lo | out_lo) % 8) return FLUXHIP_EINVAL;
  const int cpr = C / 8;
  const int cb = cpr % 64 == 0 ? 64 : cpr % 32 == 0 ? 32 : cpr % 16 == 0 ? 16 : cpr % 8 == 0 ? 8 : 0;
  if (!cb) return FLUXHIP_EINVAL;
  int ppb = 1024;
  while (ppb > 32 && (long long)B * ((HW + ppb - 1) / ppb) * (cpr / cb) < 512) ppb >>= 1;
  const int nchunks = (HW + ppb - 1) / ppb;
  if (ws_bytes < ((int64_t)B * nchunks * C + (int64_t)B * G) * 2 * (int64_t)sizeof(float))
    return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(nchunks, B, cpr / cb), block(256);
#define GN_RUN3(CB)                                                                                  \
  do {                                                                                               \
    hipLaunchKernelGGL((gn_partial_kernel<CB, true>), grid, block, 0, s, (const bf16_t*)x,           \
                       (float*)ws, HW, C, nchunks, ppb, (long long)x_lo);                            \
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(256), 0, s, (float*)ws, B, C, G,         \
                       nchunks, (float)HW * (float)(C / G), eps);                                    \
    hipLaunchKernelGGL((gn_apply_kernel<CB, true>), grid, block, 0, s, (const bf16_t*)x,             \
                       (const float*)ws, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)out,    \
                       HW, C, G, nchunks, silu, ppb, (long long)x_lo, (long long)out_lo);            \
  } while (0)
  if (cb == 64) GN_RUN3(64);
  else if (cb == 32) GN_RUN3(32);
  else if (cb == 16) GN_RUN3(16);
  else GN_RUN3(8);
#undef GN_RUN3
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// GroupNorm [+ SiLU] of a split tensor whose per-channel partial sums already sit in `ws` ([B][nchunks][C][2] floats,
// written by the epilogue of the conv that produced x: fluxhip_conv2d_x3 / fluxhip_conv_up2x_x3 with gn_ws): only the
// finalize and apply launches run, and x is read once instead of twice.
extern "C" int fluxhip_groupnorm_apply_x3(const void* x, int64_t x_lo, const void* gamma, const void* beta,
                                          void* out, int64_t out_lo, int B, int HW, int C, int G, float eps,
                                          int silu, void* ws, int64_t ws_bytes, int nchunks, void* stream) {
  if (!x || !gamma || !beta || !out || !ws || nchunks < 1) return FLUXHIP_EINVAL;
  if (B < 1 || HW < 1 || C % 8 || G < 1 || G > 64 || C % G || (C / G) % 4 || (x_lo | out_lo) % 8) return FLUXHIP_EINVAL;
  const int cpr = C / 8;
  const int cb = cpr % 64 == 0 ? 64 : cpr % 32 == 0 ? 32 : cpr % 16 == 0 ? 16 : cpr % 8 == 0 ? 8 : 0;
  if (!cb) return FLUXHIP_EINVAL;
  if (ws_bytes < ((int64_t)B * nchunks * C + (int64_t)B * G) * 2 * (int64_t)sizeof(float)) return FLUXHIP_EINVAL;
  int ppb = 1024;
  while (ppb > 32 && (long long)B * ((HW + ppb - 1) / ppb) * (cpr / cb) < 512) ppb >>= 1;
  const int achunks = (HW + ppb - 1) / ppb;       // pixel partition of the apply pass (independent of the partial table)
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(achunks, B, cpr / cb), block(256);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(256), 0, s, (float*)ws, B, C, G, nchunks,
                     (float)HW * (float)(C / G), eps, 4);
#define GN_APPLY3(CB)                                                                                \
  hipLaunchKernelGGL((gn_apply_kernel<CB, true>), grid, block, 0, s, (const bf16_t*)x, (const float*)ws, \
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)out, HW, C, G, achunks, silu, ppb, \
                     (long long)x_lo, (long long)out_lo, nchunks)
  if (cb == 64) GN_APPLY3(64);
  else if (cb == 32) GN_APPLY3(32);
  else if (cb == 16) GN_APPLY3(16);
  else GN_APPLY3(8);
#undef GN_APPLY3
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
