// Memory-bound glue kernels of the stable_diffusion/ UNet + sampler path (HBM roofline class).
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

// nn.LayerNorm(dims) with affine weight/bias, eps 1e-5 (TransformerBlock.norm1/2/3,
// stable_diffusion/.../unet.py:45,50,57).  WPR waves share a row (NCH 16-byte chunks per lane); the launcher uses more than
// one wave per row only for short inputs (see run_layernorm).  The row statistics cross the waves through LDS in a fixed
// order (deterministic).
template <int NCH, int WPR, bool H = false>
__global__ __launch_bounds__(256) void layernorm_affine_kernel(const bf16_t* __restrict__ x,
                                                               bf16_t* __restrict__ out,
                                                               long long rows, int D,
                                                               const bf16_t* __restrict__ gamma,
                                                               const bf16_t* __restrict__ beta,
                                                               float eps, int rms) {
  // rms != 0: nn.RMSNorm (x * rsqrt(mean(x^2) + eps) * gamma, no mean subtraction, no beta;
  // flux/t5.py:196-197,216)
  __shared__ float red[2][4];
  constexpr int LPR = 64 * WPR;                         // lanes per row
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & (LPR - 1);
  long long row = (long long)blockIdx.x * (4 / WPR) + threadIdx.x / LPR;
  const bool live = row < rows;                         // a dead row group walks the last row and stores nothing: every
  if (!live) row = rows - 1;                            // wave reaches the block barriers below
  const bf16_t* xr = x + row * D;
  bf16_t* orow = out + row * D;
  const int nchunk = D >> 3;
  auto group_sum = [&](float v, float* r) {
    v = wave_sum(v);
    if constexpr (WPR == 1) return v;
    if ((threadIdx.x & 63) == 0) r[wave] = v;
    __syncthreads();
    const int w0 = wave & ~(WPR - 1);
    float t = r[w0];
#pragma unroll
    for (int i = 1; i < WPR; ++i) t += r[w0 + i];
    return t;
  };
  // gamma / beta do not depend on the statistics: the multi-wave form (short inputs, latency-bound) requests them with
  // the row; one wave per row (large grids, bandwidth-bound) reads them where they are used and keeps the registers
  constexpr bool EARLY = WPR > 1;
  u32x4 gw[EARLY ? NCH : 1], bw[EARLY ? NCH : 1];
  if constexpr (EARLY) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = min(lane + i * LPR, nchunk - 1);
      gw[i] = *((const u32x4*)gamma + c);
      bw[i] = beta ? *((const u32x4*)beta + c) : u32x4{0, 0, 0, 0};
    }
  }
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * LPR;
    if (c < nchunk) {
      u32x4 w = *((const u32x4*)xr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = e_lo<H>(w[e]);
        v[i][2 * e + 1] = e_hi<H>(w[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[i][e];
  }
  const float mean = rms ? 0.f : group_sum(sum, red[0]) / (float)D;
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
  const float rstd = rsqrtf(group_sum(sq, red[1]) / (float)D + eps);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * LPR;
    if (c < nchunk) {
      u32x4 o, g4, b4;
      if constexpr (EARLY) {
        g4 = gw[i];
        b4 = bw[i];
      } else {
        g4 = *((const u32x4*)gamma + c);
        b4 = beta ? *((const u32x4*)beta + c) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = e_pack<H>((v[i][2 * e] - mean) * rstd * e_lo<H>(g4[e]) + e_lo<H>(b4[e]),
                           (v[i][2 * e + 1] - mean) * rstd * e_hi<H>(g4[e]) + e_hi<H>(b4[e]));
      *((u32x4*)orow + c) = o;
    }
  }
}

// mx.concatenate([x, res], axis=-1) on NHWC (UNetBlock2D skip connections, unet.py:250) and channel
// zero-padding (b == nullptr). 4-byte granularity: Ca, Cb even.
__global__ __launch_bounds__(256) void concat_channels_kernel(const uint32_t* __restrict__ a,
                                                              const uint32_t* __restrict__ b,
                                                              uint32_t* __restrict__ out,
                                                              long long npix, int ca2, int cb2) {
  const int c2 = ca2 + cb2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * c2) return;
  long long p = i / c2;
  int c = (int)(i - p * c2);
  out[i] = c < ca2 ? a[p * ca2 + c] : (b ? b[p * cb2 + (c - ca2)] : 0u);
}

// out = ca*x + cb*y + cc*z  (SimpleEulerSampler / SimpleEulerAncestralSampler.step with host-computed
// sigma coefficients, sampler.py:76-105; and CFG: eps_neg + w (eps_text - eps_neg), __init__.py:77-78)
template <bool H = false>
__global__ __launch_bounds__(256) void axpbypcz_kernel(const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ y,
                                                       const bf16_t* __restrict__ z,
                                                       bf16_t* __restrict__ out, long long n, float ca,
                                                       float cb, float cc, const float* __restrict__ coef) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= n) return;
  if (coef) {          // coefficients in device memory: one captured hipGraph serves every sampler step
    ca = coef[0];
    cb = coef[1];
    cc = coef[2];
  }
  if (i + 1 < n) {
    uint32_t xv = *(const uint32_t*)(x + i), yv = *(const uint32_t*)(y + i);
    float o0 = ca * e_lo<H>(xv) + cb * e_lo<H>(yv), o1 = ca * e_hi<H>(xv) + cb * e_hi<H>(yv);
    if (z) {
      uint32_t zv = *(const uint32_t*)(z + i);
      o0 += cc * e_lo<H>(zv);
      o1 += cc * e_hi<H>(zv);
    }
    *(uint32_t*)(out + i) = e_pack<H>(o0, o1);
  } else {
    float o0 = ca * e2f<H>(x[i]) + cb * e2f<H>(y[i]) + (z ? cc * e2f<H>(z[i]) : 0.f);
    out[i] = f2e<H>(o0);
  }
}

// Per-pixel tiny Linear (Autoencoder.post_quant_proj 4 -> 4 fused with z / scaling_factor, vae.py:256-258)
// with optional zero padding of the output channels to Cpad (so the following 3x3 conv sees Cin % 8 == 0).
template <bool H = false>
__global__ __launch_bounds__(256) void pixel_linear_kernel(const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ bias,
                                                           bf16_t* __restrict__ out, long long npix,
                                                           int Cin, int Cout, int Cpad, float in_div) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * Cpad) return;
  long long p = i / Cpad;
  int co = (int)(i - p * Cpad);
  float acc = 0.f;
  if (co < Cout) {
    acc = bias ? e2f<H>(bias[co]) : 0.f;
    for (int c = 0; c < Cin; ++c) acc += (e2f<H>(x[p * Cin + c]) / in_div) * e2f<H>(w[co * Cin + c]);
  }
  out[i] = f2e<H>(acc);
}

// nn.SinusoidalPositionalEncoding(cos_first=True): out[n] = [cos(x[n] * sig) | sin(x[n] * sig)]
// (unet.py:283-292,301-313). x is float32 (timesteps are not bf16-representable), sig comes from the host.
template <bool H = false>
__global__ __launch_bounds__(256) void sincos_embed_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ sig,
                                                           bf16_t* __restrict__ out, int n, int half) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * half) return;
  int r = i / half, k = i - r * half;
  float a = x[r] * sig[k];
  out[(long long)r * 2 * half + k] = f2e<H>(cosf(a));
  out[(long long)r * 2 * half + half + k] = f2e<H>(sinf(a));
}

// nn.Embedding lookup (+ optional learned position embedding added per position): out[i] = table[idx[i]]
// (+ pos[i % T]).  flux/t5.py:229,243 (wte), flux/clip.py:83-84,134-135; also used to gather the pooled
// EOS rows (flux/clip.py:148).  16-byte chunks, one thread per chunk.
template <bool H = false>
__global__ __launch_bounds__(256) void embedding_kernel(const int* __restrict__ idx,
                                                        const bf16_t* __restrict__ table,
                                                        const bf16_t* __restrict__ pos,
                                                        bf16_t* __restrict__ out, long long n, int D,
                                                        int T, int V) {
  const int cpr = D >> 3;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * cpr) return;
  long long r = i / cpr;
  int c = (int)(i - r * cpr);
  int id = min(max(idx[r], 0), V - 1);
  u32x4 w = *((const u32x4*)(table + (long long)id * D) + c);
  if (pos) {
    u32x4 pw = *((const u32x4*)(pos + (long long)(r % T) * D) + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = e_pack<H>(e_lo<H>(w[e]) + e_lo<H>(pw[e]), e_hi<H>(w[e]) + e_hi<H>(pw[e]));
  }
  *((u32x4*)(out + r * D) + c) = w;
}

}  // namespace

// H = false: bfloat16 storage; H = true: IEEE float16 storage (the stable_diffusion/ models with float16=True)
template <bool H>
static int run_layernorm(const void* x, void* out, int64_t rows, int D, const void* gamma, const void* beta,
                         float eps, int rms, void* stream) {
  if (!x || !out || !gamma || (!beta && !rms) || rows < 1 || D < 8 || D % 8 || D > 4096) return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = D / 8;
  // waves per row: one when the grid fills the chip anyway (the row's 2-3 chunk loads per lane are independent, and the
  // cross-wave exchange costs two block barriers: 4096 x 1280 takes 6.0 us with one wave per row, 7.6 us with four -
  // tools/ln_ab.py); two or four for short inputs, where more waves in flight hide the latency (512 x 4096, T5: 6.5 -> 3.8 us)
  const int wpr = rows > 1536 ? 1 : nchunk > 128 ? 4 : nchunk > 64 ? 2 : 1;
  const int nch = (nchunk + 64 * wpr - 1) / (64 * wpr);
  dim3 grid((unsigned)((rows * wpr + 3) / 4)), block(256);
#define LNA(NCH, WPR)                                                                             \
  hipLaunchKernelGGL((layernorm_affine_kernel<NCH, WPR, H>), grid, block, 0, s, (const bf16_t*)x, \
                     (bf16_t*)out, (long long)rows, D, (const bf16_t*)gamma, (const bf16_t*)beta, eps, rms)
  if (wpr == 4) {
    if (nch <= 1) LNA(1, 4);
    else LNA(2, 4);
  } else if (wpr == 2) LNA(1, 2);
  else if (nch <= 1) LNA(1, 1);
  else if (nch <= 2) LNA(2, 1);
  else if (nch <= 3) LNA(3, 1);
  else if (nch <= 4) LNA(4, 1);
  else if (nch <= 6) LNA(6, 1);
  else LNA(8, 1);
#undef LNA
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_layernorm_affine_bf16(const void* x, void* out, int64_t rows, int D,
                                             const void* gamma, const void* beta, float eps,
                                             void* stream) {
  return run_layernorm<false>(x, out, rows, D, gamma, beta, eps, 0, stream);
}
extern "C" int fluxhip_layernorm_affine_f16(const void* x, void* out, int64_t rows, int D,
                                            const void* gamma, const void* beta, float eps,
                                            void* stream) {
  return run_layernorm<true>(x, out, rows, D, gamma, beta, eps, 0, stream);
}

extern "C" int fluxhip_rmsnorm_bf16(const void* x, void* out, int64_t rows, int D, const void* gamma,
                                    float eps, void* stream) {
  return run_layernorm<false>(x, out, rows, D, gamma, nullptr, eps, 1, stream);
}

// (a 16-bit copy: the element type does not matter, float16 callers use this entry point too)
extern "C" int fluxhip_concat_channels_bf16(const void* a, const void* b, void* out, int64_t npix,
                                            int Ca, int Cb, void* stream) {
  if (!a || !out || npix < 1 || Ca < 2 || Cb < 0 || (Ca & 1) || (Cb & 1)) return FLUXHIP_EINVAL;
  long long total = npix * ((Ca + Cb) / 2);
  hipLaunchKernelGGL(concat_channels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const uint32_t*)a, (const uint32_t*)b, (uint32_t*)out,
                     (long long)npix, Ca / 2, Cb / 2);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

template <bool H>
static int run_axpbypcz(const void* x, const void* y, const void* z, void* out, int64_t n, float ca, float cb, float cc,
                        const void* coef, void* stream) {
  if (!x || !y || !out || n < 1) return FLUXHIP_EINVAL;
  long long threads = (n + 1) / 2;
  hipLaunchKernelGGL(axpbypcz_kernel<H>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)z,
                     (bf16_t*)out, (long long)n, ca, cb, cc, (const float*)coef);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_axpbypcz_bf16(const void* x, const void* y, const void* z, void* out,
                                     int64_t n, float ca, float cb, float cc, void* stream) {
  return run_axpbypcz<false>(x, y, z, out, n, ca, cb, cc, nullptr, stream);
}
extern "C" int fluxhip_axpbypcz_f16(const void* x, const void* y, const void* z, void* out,
                                    int64_t n, float ca, float cb, float cc, void* stream) {
  return run_axpbypcz<true>(x, y, z, out, n, ca, cb, cc, nullptr, stream);
}

extern "C" int fluxhip_axpbypcz_dev_bf16(const void* x, const void* y, const void* z, void* out,
                                         int64_t n, const void* coef, void* stream) {
  if (!coef) return FLUXHIP_EINVAL;
  return run_axpbypcz<false>(x, y, z, out, n, 0.f, 0.f, 0.f, coef, stream);
}
extern "C" int fluxhip_axpbypcz_dev_f16(const void* x, const void* y, const void* z, void* out,
                                        int64_t n, const void* coef, void* stream) {
  if (!coef) return FLUXHIP_EINVAL;
  return run_axpbypcz<true>(x, y, z, out, n, 0.f, 0.f, 0.f, coef, stream);
}

template <bool H>
static int run_pixel_linear(const void* x, const void* w, const void* bias, void* out, int64_t npix, int Cin, int Cout,
                            int Cpad, float in_div, void* stream) {
  if (!x || !w || !out || npix < 1 || Cin < 1 || Cin > 64 || Cout < 1 || Cpad < Cout) return FLUXHIP_EINVAL;
  long long total = npix * Cpad;
  hipLaunchKernelGGL(pixel_linear_kernel<H>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias,
                     (bf16_t*)out, (long long)npix, Cin, Cout, Cpad, in_div == 0.f ? 1.f : in_div);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_pixel_linear_bf16(const void* x, const void* w, const void* bias, void* out,
                                         int64_t npix, int Cin, int Cout, int Cpad, float in_div,
                                         void* stream) {
  return run_pixel_linear<false>(x, w, bias, out, npix, Cin, Cout, Cpad, in_div, stream);
}

template <bool H>
static int run_sincos(const void* x, const void* sig, void* out, int n, int half, void* stream) {
  if (!x || !sig || !out || n < 1 || half < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(sincos_embed_kernel<H>, dim3((n * half + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, (const float*)x, (const float*)sig, (bf16_t*)out, n, half);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_sincos_embed_f32(const void* x, const void* sig, void* out, int n, int half,
                                        void* stream) {
  return run_sincos<false>(x, sig, out, n, half, stream);
}
// float32 positions in, float16 table out
extern "C" int fluxhip_sincos_embed_f32_f16(const void* x, const void* sig, void* out, int n, int half,
                                            void* stream) {
  return run_sincos<true>(x, sig, out, n, half, stream);
}

template <bool H>
static int run_embedding(const void* idx, const void* table, const void* pos, void* out, int64_t n, int D, int T, int V,
                         void* stream) {
  if (!idx || !table || !out || n < 1 || D < 8 || D % 8 || V < 1 || (pos && T < 1)) return FLUXHIP_EINVAL;
  long long total = n * (D / 8);
  hipLaunchKernelGGL(embedding_kernel<H>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const int*)idx, (const bf16_t*)table, (const bf16_t*)pos,
                     (bf16_t*)out, (long long)n, D, T > 0 ? T : 1, V);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_embedding_bf16(const void* idx, const void* table, const void* pos, void* out,
                                      int64_t n, int D, int T, int V, void* stream) {
  return run_embedding<false>(idx, table, pos, out, n, D, T, V, stream);
}
extern "C" int fluxhip_embedding_f16(const void* idx, const void* table, const void* pos, void* out,
                                     int64_t n, int D, int T, int V, void* stream) {
  return run_embedding<true>(idx, table, pos, out, n, D, T, V, stream);
}
