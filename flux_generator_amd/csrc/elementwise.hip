// Small memory-bound kernels around the denoise / decode path.
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

// FluxSampler.step (flux/sampler.py:56-57): x + (t_prev - t) * pred, bf16 tensors, python-float dt.
// MLX op boundaries: dt*pred -> bf16, sum -> bf16.
__global__ __launch_bounds__(256) void euler_kernel(const bf16_t* __restrict__ x,
                                                    const bf16_t* __restrict__ pred,
                                                    bf16_t* __restrict__ out, long long n8,
                                                    long long n, float dt) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n8) {
    u32x4 a = *((const u32x4*)x + i), p = *((const u32x4*)pred + i), o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(bf_lo(a[e]) + rbf(dt * bf_lo(p[e])), bf_hi(a[e]) + rbf(dt * bf_hi(p[e])));
    *((u32x4*)out + i) = o;
  }
  if (i == 0) {  // scalar tail
    for (long long k = n8 * 8; k < n; ++k) out[k] = f2bf(bf2f(x[k]) + rbf(dt * bf2f(pred[k])));
  }
}

// _prepare_latent_images (flux/flux.py:57-58): [B,h,w,C] -> [B,(h/2)(w/2), C*4], feature = c*4+dy*2+dx
__global__ __launch_bounds__(256) void pack_kernel(const bf16_t* __restrict__ x,
                                                   bf16_t* __restrict__ out, int B, int h, int w,
                                                   int C) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)B * h * w * C;
  if (i >= total) return;
  int f = (int)(i % (C * 4));
  long long tok = i / (C * 4);
  int w2 = w / 2, h2 = h / 2;
  int tx = (int)(tok % w2);
  int ty = (int)((tok / w2) % h2);
  int b = (int)(tok / ((long long)w2 * h2));
  int c = f >> 2, dy = (f >> 1) & 1, dx = f & 1;
  out[i] = x[(((long long)b * h + (ty * 2 + dy)) * w + (tx * 2 + dx)) * C + c];
}

// FluxPipeline.decode unpack (flux/flux.py:159-160) fused with AutoEncoder.decode's affine
// z / scale_factor + shift_factor (flux/autoencoder.py:353): out[b,y,x,c] NHWC.
__global__ __launch_bounds__(256) void unpack_kernel(const bf16_t* __restrict__ x,
                                                     bf16_t* __restrict__ out, int B, int h, int w,
                                                     int C, float inv_scale, float shift) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)B * h * w * C;
  if (i >= total) return;
  int c = (int)(i % C);
  long long pix = i / C;
  int xx = (int)(pix % w);
  int yy = (int)((pix / w) % h);
  int b = (int)(pix / ((long long)w * h));
  int w2 = w / 2, h2 = h / 2;
  long long tok = ((long long)b * h2 + (yy >> 1)) * w2 + (xx >> 1);
  int f = c * 4 + (yy & 1) * 2 + (xx & 1);
  out[i] = f2bf(bf2f(x[tok * (C * 4) + f]) * inv_scale + shift);
}


// timestep_embedding (flux/layers.py:46-57) for a bf16 timestep vector: 1000*t is a bf16 product
// (python scalar x bf16 array), the multiply with the fp32 frequencies promotes to fp32, the
// [cos | sin] result is cast back to bf16.
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const bf16_t* __restrict__ t,
                                                                 bf16_t* __restrict__ out, int B,
                                                                 int dim, float time_factor,
                                                                 float neg_log_period) {
  const int half = dim >> 1;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * half) return;
  int b = i / half, k = i - b * half;
  float tv = rbf(time_factor * bf2f(t[b]));
  float f = expf(((float)k / (float)half) * neg_log_period);
  float a = tv * f;
  out[(long long)b * dim + k] = f2bf(cosf(a));
  out[(long long)b * dim + half + k] = f2bf(sinf(a));
}

// EmbedND / _rope (flux/layers.py:12-21,60-75): per token the (cos, sin) of pos_axis * omega_j for
// the 64 rotation pairs of a 128-wide head, rounded to bf16 like pe.astype(bf16) (flux/model.py:124).
__global__ __launch_bounds__(256) void rope_table_kernel(const int* __restrict__ ids,
                                                         bf16_t* __restrict__ out, long long ntok,
                                                         int n_axes, int a0, int a1, int a2,
                                                         float theta) {
  const int npairs = (a0 + a1 + a2) >> 1;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntok * npairs) return;
  long long tok = i / npairs;
  int j = (int)(i - tok * npairs);
  int axis, jj, dim;
  if (j < (a0 >> 1)) { axis = 0; jj = j; dim = a0; }
  else if (j < ((a0 + a1) >> 1)) { axis = 1; jj = j - (a0 >> 1); dim = a1; }
  else { axis = 2; jj = j - ((a0 + a1) >> 1); dim = a2; }
  float scale = (float)(2 * jj) / (float)dim;
  float omega = 1.0f / powf(theta, scale);
  float x = (float)ids[tok * n_axes + axis] * omega;
  uint32_t w = pack_bf16x2(cosf(x), sinf(x));
  *((uint32_t*)out + i) = w;
}

// Row softmax over float32 logits (single-head VAE attention, flux/autoencoder.py:49), bf16 out.
// One 256-thread block per row, three passes over L2-resident data.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s,
                                                           bf16_t* __restrict__ p, int cols, int ld,
                                                           float scale_log2) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + row * ld;
  bf16_t* pr = p + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nch = cols >> 2;
  float mx = -1e30f;
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
    mx = fmaxf(mx, fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mneg = -mx * scale_log2;
  float sum = 0.f;
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += __builtin_amdgcn_exp2f(fmaf(w[e], scale_log2, mneg));
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
    u32x2 o;
    o[0] = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(w[0], scale_log2, mneg)) * inv,
                       __builtin_amdgcn_exp2f(fmaf(w[1], scale_log2, mneg)) * inv);
    o[1] = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(w[2], scale_log2, mneg)) * inv,
                       __builtin_amdgcn_exp2f(fmaf(w[3], scale_log2, mneg)) * inv);
    *((u32x2*)pr + c) = o;
  }
}

// Direct 3x3 conv (pad 1, stride 1) for tiny channel counts: conv_in 16->512 and conv_out 128->3
// of the Flux decoder (flux/autoencoder.py:224-226,269). One thread per (pixel, output-channel
// group of COPT); weights [Cout][3][3][Cin]. fp32 accumulate.
template <int COPT>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
    void* __restrict__ out, int B, int H, int W, int Cin, int Cout, int out_f32, int clip01) {
  const int cgroups = (Cout + COPT - 1) / COPT;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)B * H * W * cgroups;
  if (i >= total) return;
  const int cgp = (int)(i % cgroups);
  long long pix = i / cgroups;
  const int xx = (int)(pix % W);
  const int yy = (int)((pix / W) % H);
  const int b = (int)(pix / ((long long)W * H));
  float acc[COPT];
#pragma unroll
  for (int o = 0; o < COPT; ++o) {
    int co = cgp * COPT + o;
    acc[o] = (bias && co < Cout) ? bf2f(bias[co]) : 0.f;
  }
  for (int ky = 0; ky < 3; ++ky) {
    int y = yy + ky - 1;
    if (y < 0 || y >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      int xq = xx + kx - 1;
      if (xq < 0 || xq >= W) continue;
      const bf16_t* xp = x + (((long long)b * H + y) * W + xq) * Cin;
      for (int c8 = 0; c8 < Cin; c8 += 8) {
        u32x4 xv = *(const u32x4*)(xp + c8);
#pragma unroll
        for (int o = 0; o < COPT; ++o) {
          int co = min(cgp * COPT + o, Cout - 1);
          u32x4 wv = *(const u32x4*)(w + (((long long)co * 3 + ky) * 3 + kx) * Cin + c8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[o] += bf_lo(xv[e]) * bf_lo(wv[e]);
            acc[o] += bf_hi(xv[e]) * bf_hi(wv[e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COPT; ++o) {
    int co = cgp * COPT + o;
    if (co >= Cout) continue;
    float v = acc[o];
    if (clip01) v = fminf(fmaxf(v + 1.f, 0.f), 2.f) * 0.5f;
    long long oi = pix * Cout + co;
    if (out_f32) ((float*)out)[oi] = v;
    else ((bf16_t*)out)[oi] = f2bf(v);
  }
}

// Coalesced variant for wide inputs / very few outputs (conv_out 128 -> 3): LPP = Cin/8 lanes share
// one output pixel, lane c owns the 16-byte channel chunk c of every tap, so each tap is one fully
// coalesced Cin*2-byte row read; partial dot products are combined with xor-shuffles inside the lane
// group.  The weights (Cout*9*Cin bf16, a few KB) stay in L1.
template <int LPP, int COUT>
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
    void* __restrict__ out, int B, int H, int W, int Cout, int out_f32, int clip01) {
  // A lane group walks a horizontal run of RUN output pixels: its weight slice (COUT x 9 taps x 8 channels)
  // lives in registers for the whole run and the 3x3 input window slides, so each new pixel costs 3 loads.
  constexpr int Cin = LPP * 8, RUN = 16;
  const int tid = threadIdx.x;
  const int c = tid % LPP;
  const int runs_per_row = (W + RUN - 1) / RUN;
  const long long run = (long long)blockIdx.x * (256 / LPP) + tid / LPP;
  const long long nruns = (long long)B * H * runs_per_row;
  const long long rr = run < nruns ? run : nruns - 1;
  const int x0 = (int)(rr % runs_per_row) * RUN;
  const int yy = (int)((rr / runs_per_row) % H);
  const int b = (int)(rr / ((long long)runs_per_row * H));
  u32x4 wv[COUT][9];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wv[o][t] = *((const u32x4*)(w + ((long long)min(o, Cout - 1) * 9 + t) * Cin) + c);
  float bv[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) bv[o] = bias ? bf2f(bias[min(o, Cout - 1)]) : 0.f;
  auto load_col = [&](int xq, u32x4(&col)[3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int y = yy + ky - 1;
      const bool ok = (y >= 0) & (y < H) & (xq >= 0) & (xq < W);
      col[ky] = ok ? *((const u32x4*)(x + (((long long)b * H + y) * W + xq) * Cin) + c) : u32x4{0, 0, 0, 0};
    }
  };
  u32x4 win[3][3];                       // win[kx][ky]
  load_col(x0 - 1, win[0]);
  load_col(x0, win[1]);
#pragma unroll 1
  for (int i = 0; i < RUN; ++i) {
    const int xx = x0 + i;
    if (xx >= W) break;
    load_col(xx + 1, win[2]);
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (v_dot2c_f32_bf16 would halve the VALU work, but a dependent chain of them returned wrong sums
            //  on gfx950 under ROCm 7.2 - tools probe, 2026-09 - so this stays on unpack + FMA.)
            acc[o] += bf_lo(win[kx][ky][e]) * bf_lo(wv[o][ky * 3 + kx][e]);
            acc[o] += bf_hi(win[kx][ky][e]) * bf_hi(wv[o][ky * 3 + kx][e]);
          }
#pragma unroll
    for (int o = 0; o < COUT; ++o)
#pragma unroll
      for (int s = LPP >> 1; s > 0; s >>= 1) acc[o] += __shfl_xor(acc[o], s, 64);
    if (c == 0 && run < nruns) {
      const long long pix = ((long long)b * H + yy) * W + xx;
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        if (o >= Cout) break;
        float v = acc[o] + bv[o];
        if (clip01) v = fminf(fmaxf(v + 1.f, 0.f), 2.f) * 0.5f;
        const long long oi = pix * Cout + o;
        if (out_f32) ((float*)out)[oi] = v;
        else ((bf16_t*)out)[oi] = f2bf(v);
      }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      win[0][ky] = win[1][ky];
      win[1][ky] = win[2][ky];
    }
  }
}


// ---- fp32-faithful ("bf16x3") VAE path: split tensors (include/fluxhip.h) -------------------------
DEVINL void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - bf_lo(hi), b - bf_hi(hi));
}

__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi,
                                                        bf16_t* __restrict__ lo, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const bf16_t h = f2bf(v);
  hi[i] = h;
  lo[i] = f2bf(v - bf2f(h));
}

__global__ __launch_bounds__(256) void join_f32_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo,
                                                       float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = bf2f(hi[i]) + bf2f(lo[i]);
}

// unpack (flux/flux.py:159-160) + z / scale + shift (flux/autoencoder.py:353) in float32 -> split, channels
// zero-padded to Cpad (one 64-channel K-step of the implicit-GEMM loader)
__global__ __launch_bounds__(256) void unpack_x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                        long long out_lo, int B, int h, int w, int C, int Cpad,
                                                        float scale, float shift) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * h * w * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long long pix = i / Cpad;
  float v = 0.f;
  if (c < C) {
    const int xx = (int)(pix % w);
    const int yy = (int)((pix / w) % h);
    const int b = (int)(pix / ((long long)w * h));
    const long long tok = ((long long)b * (h / 2) + (yy >> 1)) * (w / 2) + (xx >> 1);
    v = bf2f(x[tok * (C * 4) + c * 4 + (yy & 1) * 2 + (xx & 1)]) / scale + shift;
  }
  const bf16_t hh = f2bf(v);
  out[i] = hh;
  out[i + out_lo] = f2bf(v - bf2f(hh));
}

// XH: x is IEEE float16 (latents of a float16=True pipeline).  The reference divides by the scaling factor in the latents'
// dtype before the float32 Linear promotes (vae.py:256-258: z / scaling_factor on a float16 array), so the quotient is rounded
// to float16 here as well.
template <bool XH = false>
__global__ __launch_bounds__(256) void pixel_linear_x3_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
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
    for (int c = 0; c < Cin; ++c) {
      const float zc = XH ? e_rnd<true>(h2f(x[p * Cin + c]) / in_div) : bf2f(x[p * Cin + c]) / in_div;
      acc += zc * w[co * Cin + c];
    }
  }
  const bf16_t hh = f2bf(acc);
  out[i] = hh;
  out[i + out_lo] = f2bf(acc - bf2f(hh));
}

// softmax_rows_kernel with a split (hi / lo) probability matrix and full-precision exp2
__global__ __launch_bounds__(256) void softmax_rows_x3_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                              long long p_lo, int cols, int ld, float scale_log2) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + row * ld;
  bf16_t* pr = p + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nch = cols >> 2;
  float mx = -1e30f;
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
    mx = fmaxf(mx, fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mneg = -mx * scale_log2;
  float sum = 0.f;
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += exp2f(fmaf(w[e], scale_log2, mneg));
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
  for (int c = tid; c < nch; c += 256) {
    f32x4 w = *((const f32x4*)sr + c);
    float e0 = exp2f(fmaf(w[0], scale_log2, mneg)) * inv, e1 = exp2f(fmaf(w[1], scale_log2, mneg)) * inv;
    float e2 = exp2f(fmaf(w[2], scale_log2, mneg)) * inv, e3 = exp2f(fmaf(w[3], scale_log2, mneg)) * inv;
    uint32_t h0, l0, h1, l1;
    split2(e0, e1, h0, l0);
    split2(e2, e3, h1, l1);
    const u32x2 oh = {h0, h1}, ol = {l0, l1};
    *((u32x2*)pr + c) = oh;
    *((u32x2*)(pr + p_lo) + c) = ol;
  }
}

// conv3x3_fewout_kernel for a split input and float32 weights: one output channel per lane group (blockIdx.y),
// LPP = Cin/8 lanes per pixel, float32 FMAs on hi + lo.  The three output channels of a pixel run re-read the
// same window from L1 / L2.
template <int LPP>
__global__ __launch_bounds__(256) void conv3x3_fewout_x3_kernel(const bf16_t* __restrict__ x, long long x_lo,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out, int B, int H, int W,
                                                                int Cout, int clip01) {
  constexpr int Cin = LPP * 8, RUN = 16;
  const int tid = threadIdx.x;
  const int c = tid % LPP;
  const int co = blockIdx.y;
  const int runs_per_row = (W + RUN - 1) / RUN;
  const long long run = (long long)blockIdx.x * (256 / LPP) + tid / LPP;
  const long long nruns = (long long)B * H * runs_per_row;
  const long long rr = run < nruns ? run : nruns - 1;
  const int x0 = (int)(rr % runs_per_row) * RUN;
  const int yy = (int)((rr / runs_per_row) % H);
  const int b = (int)(rr / ((long long)runs_per_row * H));
  float wv[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* wp = w + ((long long)co * 9 + t) * Cin + c * 8;
    const f32x4 a = *(const f32x4*)wp, d = *(const f32x4*)(wp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wv[t][e] = a[e];
      wv[t][4 + e] = d[e];
    }
  }
  const float bv = bias ? bias[co] : 0.f;
  auto load_col = [&](int xq, float (&col)[3][8]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int y = yy + ky - 1;
      const bool ok = (y >= 0) & (y < H) & (xq >= 0) & (xq < W);
      u32x4 h = {0, 0, 0, 0}, l = {0, 0, 0, 0};
      if (ok) {
        const bf16_t* p = x + (((long long)b * H + y) * W + xq) * Cin + c * 8;
        h = *(const u32x4*)p;
        l = *(const u32x4*)(p + x_lo);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        col[ky][2 * e] = bf_lo(h[e]) + bf_lo(l[e]);
        col[ky][2 * e + 1] = bf_hi(h[e]) + bf_hi(l[e]);
      }
    }
  };
  float win[3][3][8];                    // win[kx][ky][channel]
  load_col(x0 - 1, win[0]);
  load_col(x0, win[1]);
#pragma unroll 1
  for (int i = 0; i < RUN; ++i) {
    const int xx = x0 + i;
    if (xx >= W) break;
    load_col(xx + 1, win[2]);
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(win[kx][ky][e], wv[ky * 3 + kx][e], acc);
#pragma unroll
    for (int s_ = LPP >> 1; s_ > 0; s_ >>= 1) acc += __shfl_xor(acc, s_, 64);
    if (c == 0 && run < nruns) {
      float v = acc + bv;
      if (clip01) v = fminf(fmaxf(v + 1.f, 0.f), 2.f) * 0.5f;
      out[(((long long)b * H + yy) * W + xx) * Cout + co] = v;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        win[0][ky][e] = win[1][ky][e];
        win[1][ky][e] = win[2][ky][e];
      }
  }
}


// ---- fp8 (e4m3fn) row quantiser: one 256-thread block per row, the row stays in registers between the amax
// reduction and the conversion (K <= 16384)
template <bool F32SRC>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const void* __restrict__ x, uint8_t* __restrict__ out,
                                                                float* __restrict__ scale, int K, long long ld) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const int tid = threadIdx.x, nch = K >> 3;           // 8-element chunks
  constexpr int MAXC = 8;                              // chunks per thread: K <= 8 * 256 * 8
  float v[MAXC][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      if constexpr (F32SRC) {
        const float* p = (const float*)x + row * ld + c * 8;
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
      } else {
        const u32x4 w = *(const u32x4*)((const bf16_t*)x + row * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][2 * e] = bf_lo(w[e]); v[i][2 * e + 1] = bf_hi(w[e]); }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
    }
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (tid == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      u32x2 o;
      int w0 = 0, w1 = 0;
      // v_cvt_pk_fp8_f32: two floats -> two OCP e4m3fn bytes (round to nearest even), packed into the selected half
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, w1, true);
      o[0] = (uint32_t)w0;
      o[1] = (uint32_t)w1;
      *(u32x2*)(out + row * (long long)K + c * 8) = o;
    }
  }
}

}  // namespace

extern "C" int fluxhip_euler_step_bf16(const void* x, const void* pred, void* out, int64_t n,
                                       float dt, void* stream) {
  if (!x || !pred || !out || n < 1) return FLUXHIP_EINVAL;
  long long n8 = n / 8;
  unsigned blocks = (unsigned)((n8 + 255) / 256);
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(euler_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)pred, (bf16_t*)out, n8, (long long)n, dt);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_pack_latents_bf16(const void* x, void* out, int B, int h, int w, int C,
                                         void* stream) {
  if (!x || !out || B < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || C < 1) return FLUXHIP_EINVAL;
  long long total = (long long)B * h * w * C;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, B, h, w, C);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_unpack_latents_bf16(const void* x, void* out, int B, int h, int w, int C,
                                           float scale, float shift, void* stream) {
  if (!x || !out || B < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || C < 1 || scale == 0.f)
    return FLUXHIP_EINVAL;
  long long total = (long long)B * h * w * C;
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, B, h, w, C, 1.f / scale,
                     shift);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_softmax_rows_f32(const void* s, void* p, int64_t rows, int cols, int ld,
                                        float scale, void* stream) {
  if (!s || !p || rows < 1 || cols < 4 || cols % 4 || ld % 4 || ld < cols) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const float*)s, (bf16_t*)p, cols, ld, scale * 1.4426950408889634f);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_conv2d_small(const void* x, const void* w, const void* bias, void* out,
                                    int B, int H, int W, int Cin, int Cout, int out_f32, int clip01,
                                    void* stream) {
  if (!x || !w || !out || B < 1 || H < 1 || W < 1 || Cin % 8 || Cout < 1) return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (Cout <= 4 && (Cin == 128 || Cin == 256 || Cin == 512 || Cin == 64)) {
    const long long npix = (long long)B * H * ((W + 15) / 16);   // runs of 16 pixels
#define FEWOUT(LPP)                                                                                \
  hipLaunchKernelGGL((conv3x3_fewout_kernel<LPP, 4>),                                              \
                     dim3((unsigned)((npix + (256 / LPP) - 1) / (256 / LPP))), dim3(256), 0, s,    \
                     (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, out, B, H, W, Cout,  \
                     out_f32, clip01)
    if (Cin == 64) FEWOUT(8);
    else if (Cin == 128) FEWOUT(16);
    else if (Cin == 256) FEWOUT(32);
    else FEWOUT(64);
#undef FEWOUT
  } else if (Cout <= 4) {
    long long total = (long long)B * H * W;
    hipLaunchKernelGGL((conv3x3_small_kernel<4>), dim3((unsigned)((total + 255) / 256)), dim3(256),
                       0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, out, B, H, W,
                       Cin, Cout, out_f32, clip01);
  } else {
    long long total = (long long)B * H * W * ((Cout + 7) / 8);
    hipLaunchKernelGGL((conv3x3_small_kernel<8>), dim3((unsigned)((total + 255) / 256)), dim3(256),
                       0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, out, B, H, W,
                       Cin, Cout, out_f32, clip01);
  }
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_timestep_embedding_bf16(const void* t, void* out, int B, int dim,
                                               float time_factor, float max_period, void* stream) {
  if (!t || !out || B < 1 || dim < 2 || (dim & 1)) return FLUXHIP_EINVAL;
  int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)t, (bf16_t*)out, B, dim, time_factor,
                     -logf(max_period));
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_rope_table_bf16(const void* ids, void* out, int64_t ntok, int n_axes, int a0,
                                       int a1, int a2, float theta, void* stream) {
  if (!ids || !out || ntok < 1 || n_axes != 3 || (a0 & 1) || (a1 & 1) || (a2 & 1))
    return FLUXHIP_EINVAL;
  long long n = ntok * ((a0 + a1 + a2) / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const int*)ids, (bf16_t*)out, (long long)ntok, n_axes,
                     a0, a1, a2, theta);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// ---- nn.silu on a bf16 vector (Modulation: lin(silu(vec)), flux/layers.py:136) --------------------
// The GEMV applies silu to its input on the fly when it serves one step (silu_in = 1); when one pass serves several
// steps' vectors (Flux.modulation_tables) the activation is taken out of the weight-streaming loop: same function,
// same bf16 rounding, so both orders give identical bits.
namespace {
template <bool H>
__global__ __launch_bounds__(256) void silu16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f2e<H>(silu_f(e2f<H>(x[i])));
}
}  // namespace

extern "C" int fluxhip_silu_bf16(const void* x, void* out, int64_t n, void* stream) {
  if (!x || !out || n < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(silu16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, (long long)n);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
extern "C" int fluxhip_silu_f16(const void* x, void* out, int64_t n, void* stream) {
  if (!x || !out || n < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(silu16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, (long long)n);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// ---- fp32-faithful VAE path entry points ---------------------------------------------------------
extern "C" int fluxhip_split_f32(const void* x, void* hi, void* lo, int64_t n, void* stream) {
  if (!x || !hi || !lo || n < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (bf16_t*)hi, (bf16_t*)lo, (long long)n);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_join_f32(const void* hi, const void* lo, void* out, int64_t n, void* stream) {
  if (!hi || !lo || !out || n < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(join_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)hi, (const bf16_t*)lo, (float*)out, (long long)n);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_unpack_latents_x3(const void* x, void* out, int64_t out_lo, int B, int h, int w, int C,
                                         int Cpad, float scale, float shift, void* stream) {
  if (!x || !out || B < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || C < 1 || Cpad < C || scale == 0.f)
    return FLUXHIP_EINVAL;
  const long long total = (long long)B * h * w * Cpad;
  hipLaunchKernelGGL(unpack_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, (long long)out_lo, B, h, w, C, Cpad, scale, shift);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

template <bool XH>
static int run_pixel_linear_x3(const void* x, const void* w, const void* bias, void* out, int64_t out_lo,
                               int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream) {
  if (!x || !w || !out || npix < 1 || Cin < 1 || Cin > 64 || Cout < 1 || Cpad < Cout) return FLUXHIP_EINVAL;
  const long long total = npix * Cpad;
  hipLaunchKernelGGL(pixel_linear_x3_kernel<XH>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const float*)w, (const float*)bias, (bf16_t*)out, (long long)out_lo,
                     (long long)npix, Cin, Cout, Cpad, in_div == 0.f ? 1.f : in_div);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_pixel_linear_x3(const void* x, const void* w, const void* bias, void* out, int64_t out_lo,
                                       int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream) {
  return run_pixel_linear_x3<false>(x, w, bias, out, out_lo, npix, Cin, Cout, Cpad, in_div, stream);
}
// x float16 (the latents of a float16=True stable_diffusion/ pipeline); everything else as above
extern "C" int fluxhip_pixel_linear_x3_f16in(const void* x, const void* w, const void* bias, void* out, int64_t out_lo,
                                             int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream) {
  return run_pixel_linear_x3<true>(x, w, bias, out, out_lo, npix, Cin, Cout, Cpad, in_div, stream);
}

extern "C" int fluxhip_softmax_rows_x3(const void* s, void* p, int64_t p_lo, int64_t rows, int cols, int ld,
                                       float scale, void* stream) {
  if (!s || !p || rows < 1 || cols < 4 || cols % 4 || ld % 4 || ld < cols || p_lo % 4) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(softmax_rows_x3_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const float*)s, (bf16_t*)p, (long long)p_lo, cols, ld, scale * 1.4426950408889634f);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// The same conv, LDS-tiled: a workgroup owns an 8 x 32 tile of output pixels (one per thread) and walks the input
// channels in chunks of 32.  A chunk of the (8+2) x (32+2) halo window is fetched ONCE (hi + lo planes summed to
// float32, borders zero-filled) and stored as channel planes, so that a thread's tap reads are stride-1 across lanes
// (no bank conflicts) and every value feeds all COUT outputs; the chunk's weights sit in LDS too and are read as
// broadcasts (scalar loads share the LDS's in-order counter and stalled every tap).  The input is read from HBM once (the one-output-per-launch-row
// kernel above read it COUT times and went through L1 for every tap): 512 x 512 x 128 -> 3: 214 -> 70 us.
namespace {
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_fewout_x3_tiled_kernel(const bf16_t* __restrict__ x, long long x_lo,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ out, int B, int H, int W,
                                                                      int Cin, int clip01) {
  constexpr int TH = 8, TW = 32, HW_ = (TH + 2) * (TW + 2), PL = HW_ + 1, CC = 32;   // PL: odd plane stride
  __shared__ float tile[CC * PL];
  __shared__ __attribute__((aligned(16))) float wl[COUT * 9 * CC];     // this chunk's weights [co][tap][32 channels]
  const int tid = threadIdx.x;
  const int tx = tid & (TW - 1), ty = tid >> 5;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int bx = bid % tiles_x;
  bid /= tiles_x;
  const int by = bid % tiles_y;
  const int b = bid / tiles_y;
  const int x0 = bx * TW, y0 = by * TH;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  // ---- fill: 4 lanes per halo pixel, 8 channels (16 B of each plane) per lane.  The element offsets do not depend on
  // the channel chunk; the raw loads of chunk k+1 are issued before the compute of chunk k and committed to LDS after it.
  constexpr int NIT = (HW_ * 4 + 255) / 256;
  long long goff[NIT];                              // element offset of this lane's 8 channels in chunk 0, or -1 (border)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + it * 256;
    const int pp = idx >> 2, g = idx & 3;
    const int py = pp / (TW + 2), px = pp - py * (TW + 2);
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool ok = (idx < HW_ * 4) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
    goff[it] = ok ? (((long long)b * H + gy) * W + gx) * Cin + g * 8 : -1;
  }
  u32x4 rh[NIT], rl[NIT];
  constexpr int WQ = COUT * 9 * CC / 4, WIT = (WQ + 255) / 256;      // float4 pieces of the weight slab
  f32x4 rw[WIT];
  auto issue = [&](int c0) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int q = tid + it * 256;
      if (q < WQ) rw[it] = *(const f32x4*)(w + (long long)(q >> 3) * Cin + c0 + (q & 7) * 4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      rh[it] = u32x4{0u, 0u, 0u, 0u};
      rl[it] = u32x4{0u, 0u, 0u, 0u};
      if (goff[it] >= 0) {
        const bf16_t* src = x + goff[it] + c0;
        rh[it] = *(const u32x4*)src;
        rl[it] = *(const u32x4*)(src + x_lo);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int q = tid + it * 256;
      if (q < WQ) *(f32x4*)(wl + q * 4) = rw[it];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      if (idx < HW_ * 4) {
        const int pp = idx >> 2, g = idx & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          tile[(g * 8 + 2 * e) * PL + pp] = bf_lo(rh[it][e]) + bf_lo(rl[it][e]);
          tile[(g * 8 + 2 * e + 1) * PL + pp] = bf_hi(rh[it][e]) + bf_hi(rl[it][e]);
        }
      }
    }
  };
  issue(0);
  for (int c0 = 0; c0 < Cin; c0 += CC) {
    commit();
    __syncthreads();
    if (c0 + CC < Cin) issue(c0 + CC);
    // ---- compute: this thread's pixel, 32 channels x 9 taps x COUT outputs
#pragma unroll 1
    for (int c8 = 0; c8 < CC; c8 += 8) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float* tp = tile + c8 * PL + (ty + ky) * (TW + 2) + tx + kx;
        float xs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] = tp[e * PL];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float* wp = wl + (co * 9 + tap) * CC + c8;                      // same address in every lane: LDS broadcast
          const f32x4 wa = *(const f32x4*)wp, wb = *(const f32x4*)(wp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[co] = fmaf(xs[e], wa[e], acc[co]);
            acc[co] = fmaf(xs[4 + e], wb[e], acc[co]);
          }
        }
      }
    }
    __syncthreads();
  }
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx < W && gy < H) {
    float* dst = out + (((long long)b * H + gy) * W + gx) * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co] + (bias ? bias[co] : 0.f);
      if (clip01) v = fminf(fmaxf(v + 1.f, 0.f), 2.f) * 0.5f;
      dst[co] = v;
    }
  }
}
}  // namespace

extern "C" int fluxhip_conv2d_small_x3(const void* x, int64_t x_lo, const void* w, const void* bias, void* out,
                                       int B, int H, int W, int Cin, int Cout, int clip01, void* stream) {
  if (!x || !w || !out || B < 1 || H < 1 || W < 1 || Cout < 1 || Cout > 4 || x_lo % 8) return FLUXHIP_EINVAL;
  if (Cin != 64 && Cin != 128 && Cin != 256 && Cin != 512) return FLUXHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (H * (long long)W >= 4096) {      // LDS-tiled kernel (8 x 32 pixel tiles): anything but toy images
    const long long nblk = (long long)B * ((H + 7) / 8) * ((W + 31) / 32);
#define FEWOUT3T(CO)                                                                                          \
    hipLaunchKernelGGL((conv3x3_fewout_x3_tiled_kernel<CO>), dim3((unsigned)nblk), dim3(256), 0, s,            \
                       (const bf16_t*)x, (long long)x_lo, (const float*)w, (const float*)bias, (float*)out, B, H, W, \
                       Cin, clip01)
    if (Cout == 1) FEWOUT3T(1);
    else if (Cout == 2) FEWOUT3T(2);
    else if (Cout == 3) FEWOUT3T(3);
    else FEWOUT3T(4);
#undef FEWOUT3T
    return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
  }
  const long long npix = (long long)B * H * ((W + 15) / 16);   // runs of 16 pixels
#define FEWOUT3(LPP)                                                                                   \
  hipLaunchKernelGGL((conv3x3_fewout_x3_kernel<LPP>),                                                   \
                     dim3((unsigned)((npix + (256 / LPP) - 1) / (256 / LPP)), Cout), dim3(256), 0, s,   \
                     (const bf16_t*)x, (long long)x_lo, (const float*)w, (const float*)bias, (float*)out, B, H, \
                     W, Cout, clip01)
  if (Cin == 64) FEWOUT3(8);
  else if (Cin == 128) FEWOUT3(16);
  else if (Cin == 256) FEWOUT3(32);
  else FEWOUT3(64);
#undef FEWOUT3
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// ---- block-scaled (MX) fp8 quantiser: e4m3 elements + one E8M0 scale per 32 consecutive columns, scale bytes in the tiled
// layout the block-scaled GEMM reads (include/fluxhip.h).  A thread converts 8 columns, the 4 lanes of a quad share a block.
// The arithmetic is the one of the FLAG_MXC GEMM epilogue (gemm_core.h): 2^e = smallest power of two with max|v| / 2^e <= 448.
__global__ __launch_bounds__(256) void quantize_mx_fp8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ out,
                                                             uint8_t* __restrict__ mx, long long rows, int K, long long ld,
                                                             long long ld_out, int col0, long long row0, long long kstride) {
  const int cpr = K >> 3;                                  // 8-column chunks per row (a multiple of 4)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = idx / cpr;
  const int c = (int)(idx - r * cpr);
  if (r >= rows) return;                                   // quad-uniform
  const u32x4 w = *(const u32x4*)(x + r * ld + c * 8);
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(w[e]); v[2 * e + 1] = bf_hi(w[e]); }
  float am = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[e]));
  am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, am), 0xB1, 0xf, 0xf, true)));
  am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, am), 0x4E, 0xf, 0xf, true)));
  const uint32_t ab = __builtin_bit_cast(uint32_t, am);
  int e8 = (int)(ab >> 23) - 8 + (int)((ab & 0x7fffffu) > 0x600000u);
  e8 = min(max(e8, 1), 253);
  const float mul = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * mul, v[1] * mul, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * mul, v[3] * mul, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * mul, v[5] * mul, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * mul, v[7] * mul, w1, true);
  const int col = col0 + c * 8;
  *(u32x2*)(out + r * ld_out + col) = u32x2{(uint32_t)w0, (uint32_t)w1};
  if ((threadIdx.x & 3) == 0) {
    const long long mrow = row0 + r;
    const int kb = col >> 5;
    mx[((((long long)(kb >> 2) * kstride + (mrow >> 6) * 64 + (kb & 3) * 16 + (mrow & 15)) << 2) + ((mrow >> 4) & 3))] = (uint8_t)e8;
  }
}

extern "C" int fluxhip_quantize_mx_fp8(const void* x, void* out, void* mx, int64_t rows, int K, int64_t ld, int64_t ld_out,
                                       int col0, int64_t row0, int64_t kstride, void* stream) {
  if (!x || !out || !mx || rows < 1 || K < 32 || K % 32 || ld < K || ld % 8 || ld_out % 8 || col0 < 0 || col0 % 32 ||
      ld_out < col0 + K || row0 < 0 || row0 % 64 || kstride % 64 || kstride < row0 + rows)
    return FLUXHIP_EINVAL;
  const long long n = rows * (long long)(K >> 3);
  hipLaunchKernelGGL(quantize_mx_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (uint8_t*)out, (uint8_t*)mx, (long long)rows, K, (long long)ld, (long long)ld_out, col0,
                     (long long)row0, (long long)kstride);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// ---- diagnostic: 64-bit sum of the 32-bit words of a buffer into out[0] (tools/rare_divergence_hunt.py: per-launch checksums
// without leaving the library's own kernels) ------------------------------------------------------------------------------
__global__ void debug_zero_kernel(unsigned long long* out) { *out = 0ull; }
__global__ __launch_bounds__(256) void debug_checksum_kernel(const uint32_t* __restrict__ x, long long n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
// the same reduction staged through `LDS_WORDS` words of LDS per workgroup (tools/cotenant_fault_bisect.py: does LDS use alone make a
// kernel take part in the shared-GPU fault of DESIGN.md 3.8b?)
template <int LDS_WORDS>
__global__ __launch_bounds__(256) void debug_checksum_lds_kernel(const uint32_t* __restrict__ x, long long n, unsigned long long* out) {
  __shared__ uint32_t stage[LDS_WORDS];
  unsigned long long acc = 0;
  for (long long i0 = (long long)blockIdx.x * 256; i0 < n; i0 += (long long)gridDim.x * 256) {
    const long long i = i0 + threadIdx.x;
    stage[(threadIdx.x * 17) % LDS_WORDS] = i < n ? x[i] : 0u;       // a conflict-free permutation of the first 256 words' slots
    __syncthreads();
    acc += stage[((255 - threadIdx.x) * 17) % LDS_WORDS];
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
extern "C" int fluxhip_debug_checksum_lds(const void* x, int64_t nwords, void* out, int lds_kib, void* stream) {
  if (!x || !out || nwords < 1 || (lds_kib != 16 && lds_kib != 60)) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(debug_zero_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
  if (lds_kib == 16)
    hipLaunchKernelGGL(debug_checksum_lds_kernel<4096>, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)x,
                       (long long)nwords, (unsigned long long*)out);
  else
    hipLaunchKernelGGL(debug_checksum_lds_kernel<15360>, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)x,
                       (long long)nwords, (unsigned long long*)out);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_debug_checksum(const void* x, int64_t nwords, void* out, void* stream) {
  if (!x || !out || nwords < 1) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL(debug_zero_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
  hipLaunchKernelGGL(debug_checksum_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)x, (long long)nwords,
                     (unsigned long long*)out);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

// ---- fp8 row quantiser entry points ---------------------------------------------------------------
extern "C" int fluxhip_quantize_rows_fp8(const void* x, void* out, void* scale, int64_t rows, int K, int64_t ld,
                                         void* stream) {
  if (!x || !out || !scale || rows < 1 || K < 16 || K % 16 || K > 16384 || ld < K || ld % 8) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL((quantize_rows_fp8_kernel<false>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x,
                     (uint8_t*)out, (float*)scale, K, (long long)ld);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

extern "C" int fluxhip_quantize_rows_fp8_f32(const void* x, void* out, void* scale, int64_t rows, int K, int64_t ld,
                                             void* stream) {
  if (!x || !out || !scale || rows < 1 || K < 16 || K % 16 || K > 16384 || ld < K || ld % 4) return FLUXHIP_EINVAL;
  hipLaunchKernelGGL((quantize_rows_fp8_kernel<true>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x,
                     (uint8_t*)out, (float*)scale, K, (long long)ld);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
