// float32-faithful glue kernels of the stable_diffusion/ UNet and CLIP text towers under float16=False - the reference's
// DEFAULT arithmetic for those classes (stable_diffusion/stable_diffusion/__init__.py:19-25: float32 UNet / CLIP unless
// float16=True).  The contractions run on the "bf16x3" GEMM / conv kernels the VAE decoders already use (gemm_core.h FLAG_SPLIT:
// every float32 tensor is a pair of bf16 planes hi + lo, three MFMA passes, float32 accumulation); what is here is the rest of a
// transformer / resnet block in float32 arithmetic on such split tensors: LayerNorm, the activations (SiLU, exact-erf GELU,
// quick-GELU, GEGLU), the per-image time-embedding add, the sinusoidal embedding, the embedding lookup, the causal softmax of the
// text towers, and the sampler update on float32 latents.  HBM-roofline class; none of it is on a timed path of bench.py.
//
// A split tensor is addressed as (hi plane pointer, offset of the lo plane in ELEMENTS), like every *_x3 entry point.
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

DEVINL float ld_split(const bf16_t* hi, long long lo, long long i) { return bf2f(hi[i]) + bf2f(hi[i + lo]); }
DEVINL void st_split(bf16_t* hi, long long lo, long long i, float v) {
  const bf16_t h = f2bf(v);
  hi[i] = h;
  hi[i + lo] = f2bf(v - bf2f(h));
}
// four consecutive elements (8-byte vectors on both planes)
DEVINL void ld_split4(const bf16_t* hi, long long lo, long long i, float (&v)[4]) {
  const u32x2 h = *(const u32x2*)(hi + i), l = *(const u32x2*)(hi + i + lo);
  v[0] = bf_lo(h[0]) + bf_lo(l[0]);
  v[1] = bf_hi(h[0]) + bf_hi(l[0]);
  v[2] = bf_lo(h[1]) + bf_lo(l[1]);
  v[3] = bf_hi(h[1]) + bf_hi(l[1]);
}
DEVINL void st_split4(bf16_t* hi, long long lo, long long i, const float (&v)[4]) {
  u32x2 h, l;
  h[0] = pack_bf16x2(v[0], v[1]);
  h[1] = pack_bf16x2(v[2], v[3]);
  l[0] = pack_bf16x2(v[0] - bf_lo(h[0]), v[1] - bf_hi(h[0]));
  l[1] = pack_bf16x2(v[2] - bf_lo(h[1]), v[3] - bf_hi(h[1]));
  *(u32x2*)(hi + i) = h;
  *(u32x2*)(hi + i + lo) = l;
}

// nn.LayerNorm(D) with affine float32 weight / bias (TransformerBlock.norm1/2/3, unet.py:45,50,57; CLIP layer norms,
// clip.py:35-60): one wave per row, the row in registers (D <= 4096: 16 four-element chunks per lane), mean and the variance
// about the mean in two passes over the registers, summed in a fixed lane order (deterministic).
constexpr int LN_MAXCH = 16;
__global__ __launch_bounds__(256) void layernorm_x3_kernel(const bf16_t* __restrict__ x, long long x_lo,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ out, long long out_lo, long long rows, int D,
                                                           float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long base = row * D;
  const int nch = D >> 2;
  float v[LN_MAXCH][4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      ld_split4(x, x_lo, base + 4LL * c, v[i]);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const f32x4 g = *((const f32x4*)gamma + c);
      const f32x4 b = beta ? *((const f32x4*)beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      st_split4(out, out_lo, base + 4LL * c, o);
    }
  }
}

// activations in float32 with libm's exp / erf (no approximate v_exp / v_rcp forms here: this path states float32)
DEVINL float act_f32(float x, int mode) {
  switch (mode) {
    case 0: return x / (1.0f + expf(-x));                               // SiLU (nn.silu)
    case 1: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));   // exact GELU (nn.gelu)
    case 2: return x / (1.0f + expf(-1.702f * x));                      // quick-GELU (x * sigmoid(1.702 x), clip.py:9)
    default: return x;
  }
}

// out[r, c] = act(a[r, c])                              mode 0 / 1 / 2
// out[r, c] = a[r, c] * gelu_erf(a[r, gate_off + c])     mode 3 (GEGLU, unet.py:74-78: linear1(y) * gelu(linear2(y)) with
//                                                        the two Linears evaluated as ONE GEMM over [linear1; linear2] rows)
__global__ __launch_bounds__(256) void act_x3_kernel(const bf16_t* __restrict__ a, long long a_lo, long long lda,
                                                     bf16_t* __restrict__ out, long long out_lo, long long ldo,
                                                     long long rows, int cols, int mode, int gate_off) {
  const int c4 = cols >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * c4) return;
  const long long r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  float v[4], o[4];
  ld_split4(a, a_lo, r * lda + c, v);
  if (mode == 3) {
    float g[4];
    ld_split4(a, a_lo, r * lda + gate_off + c, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * act_f32(g[e], 1);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_f32(v[e], mode);
  }
  st_split4(out, out_lo, r * ldo + c, o);
}

// x[b, p, c] += v[b, c]  (ResnetBlock2D: y + time_emb_proj(silu(temb))[:, None, None, :], unet.py:158-160), in place
__global__ __launch_bounds__(256) void addvec_x3_kernel(bf16_t* __restrict__ x, long long x_lo, const bf16_t* __restrict__ v,
                                                        long long v_lo, long long total4, long long hw, int C) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long e0 = i * 4;
  const long long pix = e0 / C;
  const int c = (int)(e0 - pix * C);
  const long long b = pix / hw;
  float xv[4], vv[4];
  ld_split4(x, x_lo, e0, xv);
  ld_split4(v, v_lo, b * C + c, vv);
#pragma unroll
  for (int e = 0; e < 4; ++e) xv[e] += vv[e];
  st_split4(x, x_lo, e0, xv);
}

// nn.SinusoidalPositionalEncoding(cos_first=True) in float32 -> split: out[n] = [cos(x[n] sig) | sin(x[n] sig)] (unet.py:283-313)
__global__ __launch_bounds__(256) void sincos_x3_kernel(const float* __restrict__ x, const float* __restrict__ sig,
                                                        bf16_t* __restrict__ out, long long out_lo, int n, int half) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * half) return;
  const int r = i / half, k = i - r * half;
  const float a = x[r] * sig[k];
  st_split(out, out_lo, (long long)r * 2 * half + k, cosf(a));
  st_split(out, out_lo, (long long)r * 2 * half + half + k, sinf(a));
}

// out = ca x + cb y + cc z on float32 tensors (the sampler step and the CFG combine of a float16=False pipeline: float32
// latents, sampler.py:76-105, __init__.py:77-78)
__global__ __launch_bounds__(256) void axpbypcz_f32_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, float* __restrict__ out, long long n,
                                                           float ca, float cb, float cc, const float* __restrict__ coef) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (coef) {
    ca = coef[0];
    cb = coef[1];
    cc = coef[2];
  }
  float o = ca * x[i] + cb * y[i];
  if (z) o += cc * z[i];
  out[i] = o;
}

// softmax over the first `cols` entries of each float32 logit row, split probabilities; T > 0: causal rows of a [.., T, ld]
// logit tensor - row r attends to columns [0, r % T] (the text towers' additive causal mask, clip.py:127-137) - the masked and
// the padding columns up to `ld` are written as zeros.  Full-precision exp (libm).
__global__ __launch_bounds__(256) void softmax_rows_masked_x3_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                                     long long p_lo, int cols, int ld, float scale, int T) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + row * ld;
  const long long pb = row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int live = T > 0 ? min(cols, (int)(row % T) + 1) : cols;
  float mx = -3.0e38f;
  for (int c = tid; c < live; c += 256) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < live; c += 256) sum += expf((sr[c] - mx) * scale);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
  for (int c = tid; c < ld; c += 256) st_split(p, p_lo, pb + c, c < live ? expf((sr[c] - mx) * scale) * inv : 0.f);
}

// nn.Embedding lookup in float32 (+ learned position embedding) -> split (clip.py:83-84,134-135)
__global__ __launch_bounds__(256) void embedding_x3_kernel(const int* __restrict__ idx, const float* __restrict__ table,
                                                           const float* __restrict__ pos, bf16_t* __restrict__ out,
                                                           long long out_lo, long long n, int D, int T, int V) {
  const int cpr = D >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * cpr) return;
  const long long r = i / cpr;
  const int c = (int)(i - r * cpr);
  const int id = min(max(idx[r], 0), V - 1);
  f32x4 w = *((const f32x4*)(table + (long long)id * D) + c);
  if (pos) {
    const f32x4 pw = *((const f32x4*)(pos + (long long)(r % T) * D) + c);
    w += pw;
  }
  const float o[4] = {w[0], w[1], w[2], w[3]};
  st_split4(out, out_lo, r * D + 4LL * c, o);
}

// Autoencoder.decode's z / scaling_factor -> post_quant_proj (vae.py:256-258) on FLOAT32 latents (float16=False pipelines)
__global__ __launch_bounds__(256) void pixel_linear_x3_f32in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                    long long out_lo, long long npix, int Cin, int Cout,
                                                                    int Cpad, float in_div) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * Cpad) return;
  const long long p = i / Cpad;
  const int co = (int)(i - p * Cpad);
  float acc = 0.f;
  if (co < Cout) {
    acc = bias ? bias[co] : 0.f;
    for (int c = 0; c < Cin; ++c) acc += (x[p * Cin + c] / in_div) * w[co * Cin + c];
  }
  st_split(out, out_lo, i, acc);
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
inline bool launched() { return hipGetLastError() == hipSuccess; }

}  // namespace

extern "C" int fluxhip_layernorm_x3(const void* x, int64_t x_lo, const float* gamma, const float* beta, void* out,
                                    int64_t out_lo, int64_t rows, int D, float eps, void* stream) {
  if (!x || !out || !gamma || rows < 1 || D < 4 || D % 4 || D > 4 * 64 * LN_MAXCH || (x_lo | out_lo) % 4) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(layernorm_x3_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (long long)x_lo, gamma, beta, (bf16_t*)out, (long long)out_lo, (long long)rows, D, eps);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_act_x3(const void* a, int64_t a_lo, int64_t lda, void* out, int64_t out_lo, int64_t ldo, int64_t rows,
                              int cols, int mode, int gate_off, void* stream) {
  if (!a || !out || rows < 1 || cols < 4 || cols % 4 || mode < 0 || mode > 3 || (a_lo | out_lo | lda | ldo) % 4 ||
      (mode == 3 && (gate_off % 4 || gate_off < cols)))
    return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(act_x3_kernel, dim3(blocks_for(rows * (cols / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (long long)a_lo, (long long)lda, (bf16_t*)out, (long long)out_lo, (long long)ldo, (long long)rows, cols, mode,
                     gate_off);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_addvec_x3(void* x, int64_t x_lo, const void* v, int64_t v_lo, int B, int64_t hw, int C, void* stream) {
  if (!x || !v || B < 1 || hw < 1 || C < 4 || C % 4 || (x_lo | v_lo) % 4) return FLUXHIP_EINVAL;
  const long long total4 = (long long)B * hw * C / 4;
  hipLaunchKernelGGL(addvec_x3_kernel, dim3(blocks_for(total4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, (long long)x_lo,
                     (const bf16_t*)v, (long long)v_lo, total4, (long long)hw, C);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_sincos_embed_x3(const float* x, const float* sig, void* out, int64_t out_lo, int n, int half,
                                       void* stream) {
  if (!x || !sig || !out || n < 1 || half < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(sincos_x3_kernel, dim3(blocks_for((long long)n * half)), dim3(256), 0, (hipStream_t)stream, x, sig,
                     (bf16_t*)out, (long long)out_lo, n, half);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_axpbypcz_f32(const float* x, const float* y, const float* z, float* out, int64_t n, float ca, float cb,
                                    float cc, const float* coef, void* stream) {
  if (!x || !y || !out || n < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(axpbypcz_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, z, out, (long long)n, ca,
                     cb, cc, coef);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_softmax_rows_masked_x3(const float* s, void* p, int64_t p_lo, int64_t rows, int cols, int ld,
                                              float scale, int causal_T, void* stream) {
  if (!s || !p || rows < 1 || cols < 1 || cols > ld || causal_T < 0) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(softmax_rows_masked_x3_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p,
                     (long long)p_lo, cols, ld, scale, causal_T);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_embedding_x3(const int* idx, const float* table, const float* pos, void* out, int64_t out_lo, int64_t n,
                                    int D, int T, int V, void* stream) {
  if (!idx || !table || !out || n < 1 || D < 4 || D % 4 || V < 1 || (pos && T < 1) || out_lo % 4) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(embedding_x3_kernel, dim3(blocks_for(n * (D / 4))), dim3(256), 0, (hipStream_t)stream, idx, table, pos,
                     (bf16_t*)out, (long long)out_lo, (long long)n, D, T, V);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_pixel_linear_x3_f32in(const float* x, const float* w, const float* bias, void* out, int64_t out_lo,
                                             int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream) {
  if (!x || !w || !out || npix < 1 || Cin < 1 || Cout < 1 || Cpad < Cout || in_div == 0.f) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(pixel_linear_x3_f32in_kernel, dim3(blocks_for(npix * Cpad)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     (bf16_t*)out, (long long)out_lo, (long long)npix, Cin, Cout, Cpad, in_div);
  return launched() ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
