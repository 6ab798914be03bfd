// Small-M linear (GEMV-class): out[b,n] = x'[b,:] . W[n,:] + bias[n], M = batch <= 16.
//
// Replaces MLPEmbedder (flux/layers.py:78-85: time_in / vector_in / guidance_in) and the
// Modulation.lin of every block (flux/layers.py:134-137). The 95 modulation layers of one
// denoise step depend only on `vec`, so the host concatenates their weights row-wise once at
// load time and a single launch of this kernel produces every shift/scale/gate of the step.
//
// HBM-bound on W (6.5 GB of bf16 per step at Flux size): one wave per ROWS output rows,
// 16-byte loads, ROWS*K/512 loads in flight per lane, no LDS (the weight is streamed exactly
// once and shared by nobody; x is a few KB and lives in L1/L2).
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

template <int ROWS, int MAXB, bool H = false>
__global__ __launch_bounds__(256) void small_linear_kernel(const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ W,
                                                           const bf16_t* __restrict__ bias,
                                                           bf16_t* __restrict__ out, int B, int N,
                                                           int K, int silu_in, int accum) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n0 = wave_global * ROWS;
  if (n0 >= N) return;
  const int nchunk = K >> 3;  // 16-byte chunks per row

  float acc[ROWS][MAXB];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[r][b] = 0.f;

  for (int c = lane; c < nchunk; c += 64) {
    u32x4 wv[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int n = min(n0 + r, N - 1);
      wv[r] = __builtin_nontemporal_load((const u32x4*)(W + (long long)n * K) + c);
    }
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      if (b < B) {
        u32x4 xv = *((const u32x4*)(x + (long long)b * K) + c);
        float xf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xf[2 * e] = e_lo<H>(xv[e]);
          xf[2 * e + 1] = e_hi<H>(xv[e]);
        }
        if (silu_in) {
#pragma unroll
          for (int e = 0; e < 8; ++e) xf[e] = e_rnd<H>(silu_f(xf[e]));
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[r][b] += e_lo<H>(wv[r][e]) * xf[2 * e];
            acc[r][b] += e_hi<H>(wv[r][e]) * xf[2 * e + 1];
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      if (b < B) {
        float s = wave_sum(acc[r][b]);
        int n = n0 + r;
        if (lane == 0 && n < N) {
          if (bias) s += e2f<H>(bias[n]);
          long long o = (long long)b * N + n;
          if (accum) s = e2f<H>(out[o]) + e_rnd<H>(s);
          out[o] = f2e<H>(s);
        }
      }
    }
  }
}

}  // namespace

template <bool H>
static int small_linear_16(const void* x, const void* W, const void* bias, void* out,
                           int B, int N, int K, int silu_in, int accum, void* stream) {
  if (!x || !W || !out || B < 1 || B > 16 || N < 1 || K < 8 || K % 8) return FLUXHIP_EINVAL;
  constexpr int ROWS = 4;
  const int waves = (N + ROWS - 1) / ROWS;
  dim3 grid((waves + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (B <= 1)
    hipLaunchKernelGGL((small_linear_kernel<ROWS, 1, H>), grid, block, 0, s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)out, B, N, K, silu_in, accum);
  else if (B <= 4)
    hipLaunchKernelGGL((small_linear_kernel<ROWS, 4, H>), grid, block, 0, s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)out, B, N, K, silu_in, accum);
  else
    hipLaunchKernelGGL((small_linear_kernel<2, 16, H>), dim3(((N + 1) / 2 + 3) / 4), block, 0, s,
                       (const bf16_t*)x, (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)out, B, N,
                       K, silu_in, accum);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_small_linear_bf16(const void* x, const void* W, const void* bias, void* out,
                                         int B, int N, int K, int silu_in, int accum, void* stream) {
  return small_linear_16<false>(x, W, bias, out, B, N, K, silu_in, accum, stream);
}
// float16 storage (stable_diffusion/ with float16=True): same kernel, IEEE half elements
extern "C" int fluxhip_small_linear_f16(const void* x, const void* W, const void* bias, void* out,
                                        int B, int N, int K, int silu_in, int accum, void* stream) {
  return small_linear_16<true>(x, W, bias, out, B, N, K, silu_in, accum, stream);
}
