/* libfluxhip — C ABI of the MI355X (gfx950) latent-diffusion denoise/decode hot path.
 *
 * The reference (voipnuggets/flux-generator) has no FFI/plugin layer: its hot path is
 * Python calling MLX ops. This header is the operator boundary a maintainer would bind
 * instead of those MLX calls (see INTEGRATION.md for the ctypes stub). Each entry point
 * cites the reference call it replaces (paths relative to the reference tree).
 *
 * Conventions (inner seam):
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the Python side);
 *   - bf16 tensors are raw uint16 bfloat16, row-major, last dim contiguous;
 *   - no allocation, no implicit synchronisation: work is enqueued on `stream` (hipStream_t);
 *   - return 0 on success, negative on error (-1 bad argument, -2 launch failure); never throws;
 *   - thread-safe iff callers use distinct streams / output buffers.
 */
#ifndef FLUXHIP_H
#define FLUXHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLUXHIP_ABI_VERSION 9

int fluxhip_abi_version(void);
/* "gfx950" — the only architecture this library is built for. */
const char* fluxhip_arch(void);

/* ---- epilogues of fluxhip_gemm_bf16 ------------------------------------------------ */
enum {
  FLUXHIP_EPI_BIAS = 0,       /* C = A W^T + b                       nn.Linear                 */
  FLUXHIP_EPI_GELU_TANH = 1,  /* C = gelu_tanh(A W^T + b)            nn.GELU(approx="tanh")    */
  FLUXHIP_EPI_GATE_RES = 2,   /* C = res + gate * (A W^T + b)        x + mod.gate * proj(...)  */
  FLUXHIP_EPI_SPLIT_GELU = 3, /* cols <  n_split -> C ; cols >= n_split -> gelu_tanh -> C2     */
  FLUXHIP_EPI_SILU = 4,       /* C = silu(A W^T + b)                                            */
  FLUXHIP_EPI_GEGLU = 5,      /* C = res * gelu_erf(A W^T + b)       UNet GEGLU (unet.py:74-78) */
  FLUXHIP_EPI_QUICK_GELU = 6, /* C = v * sigmoid(1.702 v)            CLIP quick_gelu (flux/clip.py:9) */
  FLUXHIP_EPI_GELU_ERF = 7,   /* C = gelu_erf(A W^T + b)             nn.gelu: OpenCLIP text towers (stable_diffusion/.../clip.py:11) */
  FLUXHIP_EPI_GEGLU_PAIR = 8  /* the two Linears of the UNet GEGLU (unet.py:74-78) as ONE launch: W = [N][K] with the value rows and the gate rows
                               * interleaved in blocks of 16 (rows 32 f .. 32 f + 15 = value rows 16 f .., rows 32 f + 16 .. = gate rows 16 f ..),
                               * bias likewise, C = [M][N / 2]: C[m][16 f + c] = v * gelu_erf(g).  N % 32 == 0, ldc >= N / 2; bit-identical to
                               * FLUXHIP_EPI_BIAS followed by FLUXHIP_EPI_GEGLU */
};

/* One operand group of a (possibly grouped) GEMM. Two groups share N, K, the epilogue and the
 * launch: this is how the txt and img streams of a DoubleStreamBlock (different weights, same
 * shapes; flux/layers.py:192-208,220-229) run as ONE launch over the packed [txt;img] buffer. */
typedef struct fluxhip_gemm_group {
  const void* A;         /* bf16 [nbatch][M][lda]                                         */
  const void* W;         /* bf16 [N][K]  (nn.Linear weight layout, [out,in])              */
  const void* bias;      /* bf16 [N] or NULL                                              */
  void* C;               /* bf16 [nbatch][M][ldc]                                         */
  const void* res;       /* bf16 residual, indexed like C (EPI_GATE_RES; may alias C)     */
  const void* gate;      /* bf16 [nbatch][gate_bstride] or NULL (EPI_GATE_RES)            */
  int64_t a_bstride;     /* elements between batches of A                                 */
  int64_t c_bstride;     /* elements between batches of C / res                           */
  int64_t gate_bstride;  /* elements between batches of gate                              */
  int64_t w_bstride;     /* elements between batches of W (0 = one shared weight)         */
  int32_t M;             /* rows per batch                                                */
  int32_t _pad;
  const void* add;       /* optional matrix addend [nbatch][M][ld_add] (16-bit storage type): C = epi(round16(A W^T + b) + add).
                          * The low-rank branch of an UNFUSED LoRA layer, flux/lora.py:73-76: y + (scale * z).astype(x.dtype)  */
  int64_t add_bstride;   /* elements between batches of add                               */
} fluxhip_gemm_group;

typedef struct fluxhip_gemm_desc {
  fluxhip_gemm_group g[2];
  int32_t ngroups;       /* 1 or 2                                                        */
  int32_t nbatch;
  int32_t N, K;          /* K % 64 == 0, N % 4 == 0                                       */
  int32_t lda, ldc;      /* row strides (elements), multiples of 8 / 4                    */
  int32_t epi;           /* FLUXHIP_EPI_*                                                 */
  int32_t row_bias;      /* bias indexed by output row (used for V^T = Wv Y^T)            */
  int32_t n_split;       /* EPI_SPLIT_GELU                                                */
  int32_t ldc2;
  void* C2;
  int64_t c2_bstride;
  int32_t c2_coloff;
  int32_t tile_cfg;      /* 0 = auto; otherwise index into the compiled tile configs      */
  float alpha;           /* acc scale before bias (1.0 for Linear; 0 means 1.0)           */
  int32_t out_f32;       /* EPI_BIAS only: C is float32 [..][ldc] (VAE attention logits)  */
  int32_t ld_add;        /* row stride (elements, multiple of 4) of the groups' `add` matrices; all groups or none carry one */
  int32_t _pad2;
} fluxhip_gemm_desc;

/* Replaces nn.Linear (+ fused activation / gated residual) on the Flux path:
 * flux/layers.py:104,106 (qkv, proj), :163-165,176-178 (MLP), :250,252 (linear1/linear2),
 * :295 (final linear); flux/model.py:56,64 (img_in, txt_in). */
int fluxhip_gemm_bf16(const fluxhip_gemm_desc* d, void* stream);
/* Introspection for profiling/bench: which compiled tile configuration fluxhip_gemm_bf16 uses for
 * `d` (>= 1), and that configuration's block tile / threads. No device work. */
int fluxhip_gemm_tile_cfg(const fluxhip_gemm_desc* d);
int fluxhip_gemm_tile_shape(int cfg, int* bm, int* bn, int* threads);
/* Split-K workspace.  GEMMs whose output has too few tiles to fill the chip (N = 3072 projections at batch 1,
 * the 64x64 VAE convs) are split along K over 2..4 blocks per tile; the blocks pass an fp32 partial tile through
 * this buffer in a fixed order (deterministic).  `ws` must be zero-filled, 256-byte aligned device memory that
 * stays alive and is only used by one stream at a time; the first 64 KiB hold hand-off counters (they return to
 * zero after every launch).  Without a workspace (or with one too small for a shape) split-K is not used.
 * A forced tile_cfg may carry the split factor in bits 8+ (cfg | splits << 8); fluxhip_gemm_tile_cfg reports the
 * same encoding. */
int fluxhip_set_workspace(void* ws, int64_t bytes);
/* Split-K hand-off mode.  0 (default): "reduce-scatter" wherever a launch allows it — a ping-pong bf16 tile whose whole
 * split grid is resident at once (tiles x splits <= CUs): the S blocks of a tile exchange write-through partials
 * concurrently and each finishes 1/S of the tile; everything else uses the chain (block s adds the partial of block s-1).
 * 1: chain only (diagnostics, A/B timing).  Both are deterministic; they differ in fp32 summation order.
 * 2 (tests): like 0 without the "grid <= CUs" condition - exercises the hand-off's completion path for blocks that are
 * not co-resident on an exclusive GPU.
 * fluxhip_gemm_rs_launches: number of reduce-scatter launches issued so far by this process (tests, profiling).
 * Neither mode depends on the grid being co-resident: the chain only waits for lower block ids, and a reduce-scatter block
 * polls for its peers for at most fluxhip_gemm_set_rs_timeout_us (default 100 us; env FLUXHIP_RS_TIMEOUT_US), then
 * publishes its slice, exits, and the tile's last arriver finishes it from the workspace - bit-identical either way, so a
 * GPU shared with other processes (or with CUs masked) is slower, never wrong and never stuck.
 * The split-K workspace serves one launch at a time: launches on one stream are ordered by it, and when the launching
 * stream changes the library makes the new stream wait for the work enqueued on the previous one (outside stream capture;
 * a captured graph must be replayed on a stream ordered with other split-K users by the caller). */
int fluxhip_gemm_set_splitk_mode(int mode);
int64_t fluxhip_gemm_rs_launches(void);
int fluxhip_gemm_set_rs_timeout_us(int us);
/* Lean kernels.  The tiles the transformer-block launches run on also exist as instantiations with ONE epilogue compiled
 * in (bias; bias + GELU-tanh; gate-residual; split-GELU) on the LDS-transposed store path and nothing else - no other
 * activation, no row bias / addvec / float32 output, no split-K chain; a launch that fits one takes it (same arithmetic,
 * bit-identical results; the shared kernel's unused code cost those launches 1-2 %).  fluxhip_gemm_set_lean(0) keeps every
 * launch on the generic kernels (A/B timing, tests); fluxhip_gemm_lean_launches counts the launches that took a lean one. */
int fluxhip_gemm_set_lean(int on);
int64_t fluxhip_gemm_lean_launches(void);
/* Diagnostic: device buffer of [blocks][waves][16] u64 that the phase-timed tile configurations fill with
 * summed s_memtime deltas per main-loop phase (7 phases, iteration count, whole-wave cycles and 100 MHz ticks, setup and epilogue cycles); NULL disables. */
int fluxhip_gemm_set_trace(void* buf);
/* Diagnostic (ABI 8): tile of the fp32-faithful convolutions.  cfg > 0 forces a tile configuration (| split-K factor << 8) on
 * fluxhip_conv2d_x3, 0 returns the choice to the picker.  dxr: 0 / 1 switches the halo-tile loader off / on (-1 leaves it): 3 x 3,
 * stride 1, pad 1 convolutions on the 256 x 128 tile whose tiles are 256 pixels of ONE image row (Ws % 256 == 0) stage a row segment
 * once with a one-pixel halo and take the three horizontal taps as fragment-read offsets - a third of the activation traffic of the
 * tap-by-tap loader, same sum in another fp32 accumulation order.  fluxhip_conv_dxr_launches counts the launches that took it. */
int fluxhip_conv_set_x3_tile(int cfg, int dxr);
int64_t fluxhip_conv_dxr_launches(void);

/* Implicit-GEMM convolution, NHWC bf16, weight [Cout][kh][kw][Cin] (MLX nn.Conv2d layout).
 * ksize 1|3, stride 1|2, pad 0|1, ups=1 fuses upsample_nearest(x,(2,2)) into the loader.
 * out = epi(conv(x) + bias [+ res]).  Replaces nn.Conv2d / Upsample in
 * flux/autoencoder.py:70-81,117-122,224-226,269 and the UNet/VAE convs of stable_diffusion/.
 * addvec: optional bf16 [B][Cout] added per image after the bias (ResnetBlock2D's time embedding,
 * stable_diffusion/.../unet.py:161-162).
 * Cin % 64 == 0, Cout % 4 == 0 (the 16->512 conv_in and 128->3 conv_out use the two
 * dedicated entry points below). */
int fluxhip_conv2d_bf16(const void* x, const void* w, const void* bias, const void* res,
                        const void* addvec, void* out, int B, int Hs, int Ws, int Cin, int Cout,
                        int ksize, int stride, int pad, int ups, int epi, const void* zero16,
                        void* stream);
/* Direct conv for tiny channel counts (Cin*9 not a multiple of 64, or Cout < 4). fp32 accumulate.
 * out_f32 != 0 writes float32 (the decoder's final image), optionally clip((y+1),0,2)*0.5
 * (flux/flux.py:162) when clip01 != 0. */
int fluxhip_conv2d_small(const void* x, const void* w, const void* bias, void* out, int B, int H,
                         int W, int Cin, int Cout, int out_f32, int clip01, void* stream);

/* Small-M linear (M = batch <= 16): out[b,n] = (accum ? out[b,n] : 0) + x'[b,:] . W[n,:] + bias[n],
 * x' = silu(x) when silu_in. HBM-bound on W. Replaces MLPEmbedder (flux/layers.py:78-85) and every
 * Modulation.lin (flux/layers.py:134-137) — all 95 modulation layers of a step run as ONE call on
 * the row-concatenated weight. */
int fluxhip_small_linear_bf16(const void* x, const void* W, const void* bias, void* out, int B,
                              int N, int K, int silu_in, int accum, void* stream);
/* nn.silu on n bf16 values (Modulation: lin(silu(vec)), flux/layers.py:136) as its own launch — for a GEMV pass
 * that serves several steps' vectors at once (then called with silu_in = 0 on this output); bit-identical to the
 * on-the-fly silu_in = 1 of fluxhip_small_linear_bf16. */
int fluxhip_silu_bf16(const void* x, void* out, int64_t n, void* stream);

/* out = (1 + scale) * LayerNorm(x, eps, no affine) + shift over rows of width D (D % 8 == 0,
 * D <= 4096).  B batches of Tr rows; row (b,t) is read at x + b*x_bstride + t*D and written at
 * out + b*out_bstride + t*D.  Rows t < S use the txt shift/scale, the others img; each is a
 * per-batch vector ([B][mod_bstride]).  Replaces nn.LayerNorm(affine=False) + modulate at
 * flux/layers.py:192-193,202-203,222,228,267,300. */
int fluxhip_ln_modulate_bf16(const void* x, void* out, int B, int Tr, int D, int S,
                             int64_t x_bstride, int64_t out_bstride, const void* shift_txt,
                             const void* scale_txt, const void* shift_img, const void* scale_img,
                             int64_t mod_bstride, float eps, void* stream);

/* QKNorm (RMSNorm over head_dim=128, learned scale; flux/layers.py:88-95) + RoPE
 * (flux/layers.py:29-33) on q,k, plus the V transpose the attention kernel wants.
 * qkv: bf16 [B*T][ld], q at col 0, k at col H*128, v at col 2*H*128 (head-major inside each).
 * rope: bf16 [T][64][2] = (cos, sin) per rotation pair (already rounded to bf16 like the
 * reference's pe.astype(bf16), flux/model.py:124); batch b reads rope + b*rope_bstride (0 = shared).
 * Outputs: Q,K bf16 [B][H][T][128]; Vt bf16 [B][H][128][Tpad] with zero padding for t >= T, keys stored PERMUTED inside
 * every aligned group of 16 as [0-3, 8-11, 4-7, 12-15] — the order in which fluxhip_attention_d128_bf16's PV MFMA
 * consumes them (one 16-byte LDS read per fragment). */
int fluxhip_qk_norm_rope_bf16(const void* qkv, int ld, int B, int T, int S, int H,
                              const void* qw_txt, const void* kw_txt, const void* qw_img,
                              const void* kw_img, const void* rope, int64_t rope_bstride, void* Q,
                              void* Kout, void* Vt, int Tpad, float eps, void* stream);

/* Joint (txt+img) non-causal attention, head_dim 128: O[b,t,h*128:(h+1)*128] =
 * softmax(scale * Q K^T) V.  Replaces mx.fast.scaled_dot_product_attention + the
 * transpose/reshape at flux/layers.py:36-43.  O is token-major with row stride ldo.  Vt in the key-permuted layout
 * fluxhip_qk_norm_rope_bf16 writes (see there). */
int fluxhip_attention_d128_bf16(const void* Q, const void* K, const void* Vt, void* O, int ldo,
                                int B, int H, int T, int Tpad, float scale, void* stream);
/* Diagnostic / tests: kernel variant of fluxhip_attention_d128_bf16 (0 = automatic; see csrc/attention.hip). */
int fluxhip_attention_set_variant(int variant);

/* timestep_embedding(t, dim) (flux/layers.py:46-57) for a bf16 timestep vector t[B]:
 * out[b] = [cos(a) | sin(a)], a = bf16(time_factor * t[b]) * exp(-ln(max_period) * k / (dim/2)). */
int fluxhip_timestep_embedding_bf16(const void* t, void* out, int B, int dim, float time_factor,
                                    float max_period, void* stream);

/* EmbedND (flux/layers.py:60-75) reduced to what attention needs: ids int32 [ntok][3] ->
 * out bf16 [ntok][(a0+a1+a2)/2][2] = (cos, sin) of ids[axis] * theta^(-2j/a_axis). */
int fluxhip_rope_table_bf16(const void* ids, void* out, int64_t ntok, int n_axes, int a0, int a1,
                            int a2, float theta, void* stream);

/* x_out = x + dt * pred  (bf16; FluxSampler.step, flux/sampler.py:56-57). */
int fluxhip_euler_step_bf16(const void* x, const void* pred, void* out, int64_t n, float dt,
                            void* stream);

/* 2x2 pixel-unshuffle pack [B,h,w,C] -> [B,(h/2)(w/2),4C] (flux/flux.py:57-58) and its inverse
 * (flux/flux.py:159-160).  Packed feature index = c*4 + dy*2 + dx. */
int fluxhip_pack_latents_bf16(const void* x, void* out, int B, int h, int w, int C, void* stream);
int fluxhip_unpack_latents_bf16(const void* x, void* out, int B, int h, int w, int C, float scale,
                                float shift, void* stream);

/* GroupNorm(G groups, eps, affine) [+ SiLU] on NHWC bf16 (nn.GroupNorm(pytorch_compatible=True),
 * flux/autoencoder.py:29-35,62-78,266).  ws: float workspace of >= (B*ceil(HW/32)*G + B*G)*2 floats.
 * Two launches (partial statistics, then normalise) on `stream`. */
int fluxhip_groupnorm_silu_bf16(const void* x, const void* gamma, const void* beta, void* out,
                                int B, int HW, int C, int G, float eps, int silu, void* ws,
                                int64_t ws_bytes, void* stream);

/* Row softmax of float32 logits scaled by `scale`: P(bf16) = softmax(scale * S), fp32 inside.
 * Used by the single-head VAE AttnBlock (flux/autoencoder.py:49).  ld: row stride of S and P. */
int fluxhip_softmax_rows_f32(const void* s, void* p, int64_t rows, int cols, int ld, float scale,
                             void* stream);


/* ---- stable_diffusion/ UNet path ---------------------------------- */

/* Attention with explicit (batch, head, row) element strides for Q and K and head_dim 64 or 128:
 * O[b, t, h*hd:(h+1)*hd] = softmax(scale * Q K^T) V, V given transposed as Vt [B][H*hd][Tkpad]
 * (zero padded keys).  Self- and cross-attention of nn.MultiHeadAttention in TransformerBlock
 * (stable_diffusion/stable_diffusion/unet.py:46-54,64-71; no masks are ever passed, unet.py:403-411). */
int fluxhip_attention_strided_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                   const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                   const void* Vt, void* O, int ldo, int B, int H, int head_dim,
                                   int Tq, int Tk, int Tkpad, float scale, void* stream);

/* The same with an explicit element stride vt_bs >= H*head_dim*Tkpad between the V^T images of two batches: the
 * cross-attention K / V^T of every transformer layer of the UNet depend on the text only, so the host projects them for
 * all layers in two launches (concatenated key_proj / value_proj weights) and each layer reads its [H*hd][Tkpad] slice of
 * the [B][sum C][Tkpad] image (unet.py:46-54: the per-layer key_proj / value_proj of the encoder states). */
int fluxhip_attention_strided_vt_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                      const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                      const void* Vt, int64_t vt_bs, void* O, int ldo, int B, int H,
                                      int head_dim, int Tq, int Tk, int Tkpad, float scale, void* stream);

/* nn.LayerNorm(D) with affine gamma/beta (unet.py:45,50,57), rows of width D <= 4096. */
int fluxhip_layernorm_affine_bf16(const void* x, void* out, int64_t rows, int D, const void* gamma,
                                  const void* beta, float eps, void* stream);

/* out[p] = [a[p, :Ca] | b[p, :Cb]] (mx.concatenate(axis=-1), unet.py:250); b == NULL pads with zeros. */
int fluxhip_concat_channels_bf16(const void* a, const void* b, void* out, int64_t npix, int Ca,
                                 int Cb, void* stream);

/* out = ca*x + cb*y + cc*z (z may be NULL).  SimpleEuler(Ancestral)Sampler.step with host-computed
 * sigma coefficients (sampler.py:76-105) and classifier-free guidance (__init__.py:77-78). */
int fluxhip_axpbypcz_bf16(const void* x, const void* y, const void* z, void* out, int64_t n, float ca,
                          float cb, float cc, void* stream);
/* Same, with (ca, cb, cc) read from device memory (float32[3]): the coefficients of sampler step i are written
 * into a static buffer before a captured hipGraph of the whole UNet step is replayed, so ONE graph serves every
 * (t, t_prev) of a run instead of one capture per step. */
int fluxhip_axpbypcz_dev_bf16(const void* x, const void* y, const void* z, void* out, int64_t n,
                              const void* coef, void* stream);

/* Per-pixel Linear for tiny channel counts: out[p, :Cout] = W (x[p] / in_div) + bias, zero padded to
 * Cpad channels.  Autoencoder.decode's z / scaling_factor + post_quant_proj (vae.py:256-258). */
int fluxhip_pixel_linear_bf16(const void* x, const void* w, const void* bias, void* out, int64_t npix,
                              int Cin, int Cout, int Cpad, float in_div, void* stream);

/* nn.SinusoidalPositionalEncoding(cos_first=True): out[n] = [cos(x[n]*sig) | sin(x[n]*sig)], x and
 * sig float32, out bf16 [n][2*half] (unet.py:283-292,301-313,413,419). */
int fluxhip_sincos_embed_f32(const void* x, const void* sig, void* out, int n, int half, void* stream);

/* ---- fp32-faithful ("bf16x3") VAE decode path -------------------------------------------------
 * The reference decodes both VAEs in float32: flux/utils.py:137-143 loads ae.safetensors in the checkpoint
 * dtype and never casts (bf16 latents x fp32 weights promote to fp32), and
 * stable_diffusion/stable_diffusion/__init__.py:25 calls load_autoencoder(model, False).  gfx950's fp32 MFMA
 * runs at 1/16 of the bf16 rate, so this path keeps every fp32 value x as TWO bf16 planes
 *     hi = bf16(x),  lo = bf16(x - hi)        (x = hi + lo up to 2^-17 |x|)
 * and evaluates a product as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16 matrix cores with fp32
 * accumulation (three MFMA passes; the dropped lo*lo term is 2^-16 relative).  Norm statistics, softmax and
 * the final 128->3 conv are plain fp32 arithmetic on hi + lo.  In every entry point below a "split tensor"
 * is the hi-plane pointer plus the ELEMENT offset of its lo plane (`*_lo`); biases / norm affines are float32. */

/* x float32 [n] -> hi / lo bf16 planes (weights at load time, test inputs), and back (hi + lo). */
int fluxhip_split_f32(const void* x, void* hi, void* lo, int64_t n, void* stream);
int fluxhip_join_f32(const void* hi, const void* lo, void* out, int64_t n, void* stream);

typedef struct fluxhip_gemm_x3_desc {
  const void* A;         /* split [nbatch][M][lda]                                        */
  const void* W;         /* split [N][K]                                                  */
  const void* bias;      /* float32 [N] (or [M] with row_bias) or NULL                    */
  void* C;               /* split [nbatch][M][ldc], or float32 when out_f32               */
  const void* res;       /* split residual indexed like C (EPI_GATE_RES, gate = 1)        */
  int64_t a_lo, w_lo, c_lo, res_lo;   /* element offsets of the lo planes (a_lo, w_lo % 8 == 0; c_lo, res_lo % 4 == 0) */
  int64_t a_bstride, c_bstride, w_bstride;
  int32_t M, nbatch, N, K, lda, ldc;
  int32_t epi;           /* FLUXHIP_EPI_BIAS or FLUXHIP_EPI_GATE_RES                      */
  int32_t row_bias, out_f32, tile_cfg;
  float alpha;
  int32_t _pad;
} fluxhip_gemm_x3_desc;
/* nn.Linear in float32 (AttnBlock q/k/v/proj_out, flux/autoencoder.py:36-39; the QK^T and PV products of its
 * single-head attention, :49; vae.py:25-42). */
int fluxhip_gemm_x3(const fluxhip_gemm_x3_desc* d, void* stream);
/* nn.Conv2d in float32 (same geometry rules as fluxhip_conv2d_bf16; res != NULL adds the split residual). */
/* gn_ws != NULL (gn_ws_bytes of float32 scratch, gn_nchunks != NULL): the epilogue also writes the GroupNorm
 * partial sums of the OUTPUT (every VAE conv feeds a GroupNorm) as [B][*gn_nchunks][Cout][2] floats, to be consumed by
 * fluxhip_groupnorm_apply_x3 — the statistics pass over the tensor disappears.  *gn_nchunks = 0 when the chosen tile
 * cannot do it (pixels per image not a multiple of the tile height, or the scratch is too small): use
 * fluxhip_groupnorm_silu_x3 then. */
int fluxhip_conv2d_x3(const void* x, int64_t x_lo, const void* w, int64_t w_lo, const void* bias,
                      const void* res, int64_t res_lo, void* out, int64_t out_lo, int B, int Hs, int Ws,
                      int Cin, int Cout, int ksize, int stride, int pad, int ups, void* gn_ws,
                      int64_t gn_ws_bytes, int* gn_nchunks, const void* zero16, void* stream);
/* Upsample (nearest x2) + Conv2d 3x3 pad 1 in float32 (flux/autoencoder.py:117-122; stable_diffusion/vae.py
 * upsample path) in sub-pixel form: four 2x2 convs of the low-res input, one per output-pixel parity, on weights
 * w4 = [4][Cout][2][2][Cin] in which the 3x3 taps that read the same source pixel are pre-summed (parity =
 * 2*dy + dx; dy = 0: rows {w0, w1+w2}, dy = 1: rows {w0+w1, w2}; same along x).  4/9 of the MFMA work of the
 * fused-upsample loader.  x: split [B][Hs][Ws][Cin]; out: split [B][2Hs][2Ws][Cout]; bias float32. */
int fluxhip_conv_up2x_x3(const void* x, int64_t x_lo, const void* w4, int64_t w_lo, const void* bias, void* out,
                         int64_t out_lo, int B, int Hs, int Ws, int Cin, int Cout, void* gn_ws,
                         int64_t gn_ws_bytes, int* gn_nchunks, const void* zero16, void* stream);
/* GroupNorm [+ SiLU] on a split NHWC tensor with float32 gamma / beta; float32 arithmetic throughout. */
int fluxhip_groupnorm_silu_x3(const void* x, int64_t x_lo, const void* gamma, const void* beta, void* out,
                              int64_t out_lo, int B, int HW, int C, int G, float eps, int silu, void* ws,
                              int64_t ws_bytes, void* stream);
/* The same GroupNorm when the partial sums of x are already in ws (written by the producing conv's epilogue, see
 * fluxhip_conv2d_x3): finalize + apply only. */
int fluxhip_groupnorm_apply_x3(const void* x, int64_t x_lo, const void* gamma, const void* beta, void* out,
                               int64_t out_lo, int B, int HW, int C, int G, float eps, int silu, void* ws,
                               int64_t ws_bytes, int nchunks, void* stream);
/* softmax(scale * S) of float32 logits -> split P. */
int fluxhip_softmax_rows_x3(const void* s, void* p, int64_t p_lo, int64_t rows, int cols, int ld, float scale,
                            void* stream);
/* Final 3x3 conv to <= 4 channels (Cin in {64,128,256,512}) of a split tensor with float32 weights
 * [Cout][3][3][Cin] and bias -> float32 image, optional clip((y+1),0,2)*0.5 (flux/flux.py:162). */
int fluxhip_conv2d_small_x3(const void* x, int64_t x_lo, const void* w, const void* bias, void* out, int B,
                            int H, int W, int Cin, int Cout, int clip01, void* stream);
/* fluxhip_unpack_latents_bf16 producing a split tensor zero-padded to Cpad channels (conv_in's K-step). */
int fluxhip_unpack_latents_x3(const void* x, void* out, int64_t out_lo, int B, int h, int w, int C, int Cpad,
                              float scale, float shift, void* stream);
/* fluxhip_pixel_linear_bf16 with float32 weights / bias producing a split tensor (SD VAE post_quant_proj). */
int fluxhip_pixel_linear_x3(const void* x, const void* w, const void* bias, void* out, int64_t out_lo,
                            int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream);

/* ---- fp8 weight / activation path (BASELINE.json configs[4]: "Flux-schnell fp8 weights on CDNA4 fp8 MFMA") ------
 * Replaces the reference's `--quantize` (txt2image.py:26-28,79-82: nn.quantize of every Linear whose input width is
 * a multiple of 512).  Weights: OCP e4m3fn with one float32 scale per OUTPUT CHANNEL, quantised once at load.
 * Activations: e4m3fn with one float32 scale per TOKEN (row), quantised on the fly by fluxhip_quantize_rows_fp8.
 * Products run on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; twice the bf16 MFMA rate), accumulate in
 * float32, and the epilogue applies a_scale[m] * w_scale[n] before the bias and the fused activation / gate /
 * residual, which are the same as fluxhip_gemm_bf16's.  Outputs stay bf16. */

/* x bf16 [rows][ld] (first K columns) -> out e4m3fn [rows][K], scale float32 [rows]:
 * scale[r] = max|x[r,:]| / 448 (1 if the row is zero), out = round_to_nearest_even(x / scale).  K % 16 == 0. */
int fluxhip_quantize_rows_fp8(const void* x, void* out, void* scale, int64_t rows, int K, int64_t ld, void* stream);
/* float32 [rows][K] source (weights kept in a wider master dtype), same output. */
int fluxhip_quantize_rows_fp8_f32(const void* x, void* out, void* scale, int64_t rows, int K, int64_t ld, void* stream);

/* fluxhip_ln_modulate_bf16 with the per-token e4m3 quantisation fused: out is e4m3fn [B][Tr][D] (out_bstride in
 * elements = bytes), row_scale float32 [B*Tr] (row (b,t) at b*Tr + t).  Bit-identical to ln_modulate_bf16 followed by
 * fluxhip_quantize_rows_fp8, without the bf16 round trip through HBM and without the second launch. */
int fluxhip_ln_modulate_fp8(const void* x, void* out, void* row_scale, int B, int Tr, int D, int S,
                            int64_t x_bstride, int64_t out_bstride, const void* shift_txt, const void* scale_txt,
                            const void* shift_img, const void* scale_img, int64_t mod_bstride, float eps,
                            void* stream);

typedef struct fluxhip_fp8_scales {
  const void* a_scale[2];   /* float32 [nbatch][M] per group: one per row of A                     */
  const void* w_scale[2];   /* float32 [N] per group: one per row of W (output channel)            */
  int64_t a_scale_bstride;  /* elements between batches of a_scale                                  */
} fluxhip_fp8_scales;
/* `d` as for fluxhip_gemm_bf16 with A and W pointing at e4m3fn bytes (lda, strides and K in ELEMENTS; K % 128 == 0,
 * lda % 16 == 0); bias / C / res / gate / C2 stay bf16. */
int fluxhip_gemm_fp8(const fluxhip_gemm_desc* d, const fluxhip_fp8_scales* sc, void* stream);
int fluxhip_gemm_fp8_tile_cfg(const fluxhip_gemm_desc* d);

/* diagnostic (tools/rare_divergence_hunt.py): out[0] (uint64) = sum of the nwords 32-bit words at x */
int fluxhip_debug_checksum(const void* x, int64_t nwords, void* out, void* stream);
/* ... staged through lds_kib (16 or 60) KiB of LDS per workgroup (diagnostic: DESIGN.md 3.8b) */
int fluxhip_debug_checksum_lds(const void* x, int64_t nwords, void* out, int lds_kib, void* stream);

/* ---- block-scaled ("MX") fp8: quantisation fused into the producing kernel ---------------------------------------
 * A per-token scale needs the whole row before the first byte can be written, which forces a stand-alone quantise pass
 * between a producer (GELU epilogue, attention) and the next Linear.  With one power-of-two scale per 32 consecutive
 * elements of a row (OCP MX: e4m3 elements, E8M0 scales = biased exponent bytes, value 2^(byte - 127)) a producer can
 * quantise the tile it holds, and v_mfma_scale_f32_16x16x128_f8f6f4 applies the scales itself.
 * Block scale of 32 values: the smallest power of two 2^e with max|v| / 2^e <= 448; elements = RNE(v / 2^e) in e4m3fn
 * (nothing saturates); an all-zero block gets byte 1.
 * Scale bytes are stored TILED so that one lane of the GEMM reads the four bytes it needs for a 128-element K-step as one
 * dword: with `row` counted over the whole scale buffer (row0 of the group + batch * bstride + m; row0, bstride and
 * kstride multiples of 64, kstride >= the buffer's row count) and kb = column / 32,
 *     byte offset = (((kb >> 2) * kstride + (row >> 6) * 64 + (kb & 3) * 16 + (row & 15)) << 2) + ((row >> 4) & 3).
 * Buffer size: (K / 128) * kstride * 4 bytes. */
typedef struct fluxhip_fp8_mx {
  /* consumer (exactly one of a_mx / c_mx is set per launch): A is e4m3 [M][lda] with block scales a_mx; sc->a_scale is not read */
  const void* a_mx;
  int32_t a_mx_row0[2];     /* first scale-buffer row of each group                                  */
  int64_t a_mx_bstride;     /* scale-buffer rows between batches                                     */
  int64_t a_mx_kstride;     /* dwords between K-steps of 128 elements (= rows of the scale buffer)   */
  /* producer (epi GELU_TANH: every column; SPLIT_GELU: the columns >= n_split, stored at n - n_split + c8_coloff; the
   * other columns go to C in bf16 as usual): e4m3 bytes c8[g] [M][ldc8] + block scales c_mx instead of bf16          */
  void* c8[2];
  int64_t c8_bstride;       /* bytes between batches of c8                                           */
  int32_t ldc8, c8_coloff;
  void* c_mx;
  int32_t c_mx_row0[2];
  int64_t c_mx_bstride, c_mx_kstride;
} fluxhip_fp8_mx;
/* fluxhip_gemm_fp8 with a block-scaled activation operand or a block-scaled output.  Every group's M must be a multiple
 * of 64; runs unsplit on the ping-pong tiles (tile_cfg 49-53, 55).  N % 32 == 0 for the producer form. */
int fluxhip_gemm_fp8_mx(const fluxhip_gemm_desc* d, const fluxhip_fp8_scales* sc, const fluxhip_fp8_mx* mx, void* stream);
/* x bf16 [rows][ld] (K columns) -> e4m3 out[rows][ld_out] at column col0 + block scales (tiling above; scale-buffer row =
 * row0 + r).  K % 32 == 0, col0 % 32 == 0, row0 % 64 == 0, ld % 8 == 0, ld_out % 8 == 0. */
int fluxhip_quantize_mx_fp8(const void* x, void* out, void* mx, int64_t rows, int K, int64_t ld, int64_t ld_out, int col0,
                            int64_t row0, int64_t kstride, void* stream);
/* fluxhip_attention_d128_bf16 whose output leaves as the block-scaled operand of the next Linear: out8 e4m3 [B*T][ld8] (head h
 * at columns [128 h, 128 h + 128)), mx = tiled block scales with scale-buffer row b * T + t (mx_kstride >= B * T). */
int fluxhip_attention_d128_mx(const void* Q, const void* K, const void* Vt, void* out8, int ld8, void* mx, int64_t mx_kstride,
                              int B, int H, int T, int Tpad, float scale, void* stream);

/* ---- text encoders (flux/t5.py, flux/clip.py) ------------------------- */

/* head_dim-64 attention with either an additive per-head bias [H][Tq][Tk] bf16 (T5: relative position
 * bias passed as the SDPA mask with scale 1.0, flux/t5.py:153-155,220-224; pads are attended) or a
 * causal mask (CLIP text model, flux/clip.py:91-95,138).  Exactly one of bias / causal must be set. */
int fluxhip_attention_masked_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K,
                                  int64_t k_bs, int64_t k_hs, int64_t k_rs, const void* Vt, void* O,
                                  int ldo, int B, int H, int Tq, int Tk, int Tkpad, float scale,
                                  const void* bias, int causal, void* stream);

/* nn.RMSNorm(D, eps) with learned scale (flux/t5.py:196-197,216). */
int fluxhip_rmsnorm_bf16(const void* x, void* out, int64_t rows, int D, const void* gamma, float eps,
                         void* stream);

/* nn.Embedding: out[i] = table[idx[i]] (+ pos[i % T] when pos != NULL); idx int32 [n], rows of D bf16
 * (flux/t5.py:229,243; flux/clip.py:83-84,134-135,148). */
int fluxhip_embedding_bf16(const void* idx, const void* table, const void* pos, void* out, int64_t n,
                           int D, int T, int V, void* stream);

/* ---- float16 storage (ABI 6) -------------------------------------------------------------------------------------
 * The reference runs the stable_diffusion/ UNet and text towers in float16 when the caller passes float16=True
 * (stable_diffusion/__init__.py:20-27; flux_app.py:77-79 does) and in float32 otherwise.  The entry points below are the
 * float16 twins of the bf16 operators the UNet / CLIP / sampler path uses: SAME signatures, SAME kernels, every 16-bit
 * operand (activations, weights, biases, residuals, outputs) is IEEE half instead of bfloat16, the products run on
 * v_mfma_f32_16x16x32_f16 / 32x32x16_f16 (the bf16 issue rate), accumulation / statistics / softmax stay float32.  Values
 * beyond 65504 become inf exactly as in the reference's float16 arrays.  fluxhip_concat_channels_bf16 is a 16-bit copy and
 * serves both types.  fluxhip_gemm_f16 supports every epilogue of fluxhip_gemm_bf16 (incl. FLUXHIP_EPI_GEGLU_PAIR) on the
 * tiles the pickers choose; a forced tile_cfg without a float16 instantiation returns FLUXHIP_EINVAL.  Split-K launches take
 * the chain hand-off.  fluxhip_attention_*_f16: head_dim 64; fluxhip_attention_masked_f16: causal only.
 * fluxhip_sincos_embed_f32_f16: float32 positions in, float16 table out.  fluxhip_pixel_linear_x3_f16in: the fp32-faithful
 * VAE's first op on float16 latents (z / scaling_factor rounded to float16 like the reference's array division, then the
 * float32 post_quant_proj). */
int fluxhip_gemm_f16(const fluxhip_gemm_desc* d, void* stream);
int fluxhip_conv2d_f16(const void* x, const void* w, const void* bias, const void* res,
                       const void* addvec, void* out, int B, int Hs, int Ws, int Cin, int Cout,
                       int ksize, int stride, int pad, int ups, int epi, const void* zero16,
                       void* stream);
int fluxhip_small_linear_f16(const void* x, const void* W, const void* bias, void* out, int B,
                             int N, int K, int silu_in, int accum, void* stream);
int fluxhip_silu_f16(const void* x, void* out, int64_t n, void* stream);
int fluxhip_groupnorm_silu_f16(const void* x, const void* gamma, const void* beta, void* out,
                               int B, int HW, int C, int G, float eps, int silu, void* ws,
                               int64_t ws_bytes, void* stream);
int fluxhip_attention_strided_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                  const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                  const void* Vt, void* O, int ldo, int B, int H, int head_dim,
                                  int Tq, int Tk, int Tkpad, float scale, void* stream);
int fluxhip_attention_strided_vt_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                     const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                     const void* Vt, int64_t vt_bs, void* O, int ldo, int B, int H,
                                     int head_dim, int Tq, int Tk, int Tkpad, float scale, void* stream);
int fluxhip_attention_masked_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K,
                                 int64_t k_bs, int64_t k_hs, int64_t k_rs, const void* Vt, void* O, int ldo,
                                 int B, int H, int Tq, int Tk, int Tkpad, float scale, const void* bias,
                                 int causal, void* stream);
int fluxhip_layernorm_affine_f16(const void* x, void* out, int64_t rows, int D, const void* gamma,
                                 const void* beta, float eps, void* stream);
int fluxhip_axpbypcz_f16(const void* x, const void* y, const void* z, void* out, int64_t n, float ca,
                         float cb, float cc, void* stream);
int fluxhip_axpbypcz_dev_f16(const void* x, const void* y, const void* z, void* out, int64_t n,
                             const void* coef, void* stream);
int fluxhip_sincos_embed_f32_f16(const void* x, const void* sig, void* out, int n, int half, void* stream);
int fluxhip_embedding_f16(const void* idx, const void* table, const void* pos, void* out, int64_t n,
                          int D, int T, int V, void* stream);
int fluxhip_pixel_linear_x3_f16in(const void* x, const void* w, const void* bias, void* out, int64_t out_lo,
                                  int64_t npix, int Cin, int Cout, int Cpad, float in_div, void* stream);

/* ---- float32 arithmetic for the stable_diffusion/ UNet and text towers (ABI 9) ------------------------------------
 * `StableDiffusion(model)` / `StableDiffusionXL(model)` with the reference's DEFAULT float16=False run UNet and CLIP in
 * float32 (stable_diffusion/stable_diffusion/__init__.py:19-25, model_io.py:171-174).  Here that arithmetic is the
 * float32-faithful "bf16x3" form the VAE decoders use: a float32 tensor is a SPLIT tensor (hi + lo bf16 planes, addressed
 * as hi pointer + offset of the lo plane in elements), every Linear / conv is fluxhip_gemm_x3 / fluxhip_conv2d_x3 (three
 * MFMA passes, float32 accumulation), GroupNorm is fluxhip_groupnorm_silu_x3, and the entry points below are the rest of
 * a ResnetBlock2D / TransformerBlock / CLIP layer in float32 on split tensors.  gamma / beta / tables are float32.
 *   fluxhip_layernorm_x3        nn.LayerNorm(D) rows (unet.py:45,50,57; clip.py), D % 4 == 0, D <= 4096
 *   fluxhip_act_x3              mode 0 SiLU, 1 exact-erf GELU, 2 quick-GELU (x sigmoid(1.702 x)), 3 GEGLU: out[r, c] =
 *                               a[r, c] * gelu(a[r, gate_off + c]) (unet.py:74-78 with linear1 / linear2 as one GEMM); libm exp / erf
 *   fluxhip_addvec_x3           x[b, p, c] += v[b, c] in place (the time-embedding add of ResnetBlock2D, unet.py:158-160)
 *   fluxhip_sincos_embed_x3     [cos(x sig) | sin(x sig)] (unet.py:283-313), float32 in, split out
 *   fluxhip_axpbypcz_f32        out = ca x + cb y + cc z on float32 latents (sampler.py:76-105, CFG __init__.py:77-78);
 *                               coef != NULL: (ca, cb, cc) read from a float32[3] device buffer
 *   fluxhip_softmax_rows_masked_x3   softmax(scale * s[r, :cols]) of float32 logits -> split probabilities [rows][ld];
 *                               causal_T > 0: row r sees columns [0, r % causal_T] (clip.py:127-137); other columns are zeros
 *   fluxhip_embedding_x3        out[i] = table[idx[i]] (+ pos[i % T]) from float32 tables (clip.py:83-84,134-135)
 *   fluxhip_pixel_linear_x3_f32in    fluxhip_pixel_linear_x3 on float32 latents (vae.py:256-258) */
int fluxhip_layernorm_x3(const void* x, int64_t x_lo, const float* gamma, const float* beta, void* out, int64_t out_lo,
                         int64_t rows, int D, float eps, void* stream);
int fluxhip_act_x3(const void* a, int64_t a_lo, int64_t lda, void* out, int64_t out_lo, int64_t ldo, int64_t rows, int cols,
                   int mode, int gate_off, void* stream);
int fluxhip_addvec_x3(void* x, int64_t x_lo, const void* v, int64_t v_lo, int B, int64_t hw, int C, void* stream);
int fluxhip_sincos_embed_x3(const float* x, const float* sig, void* out, int64_t out_lo, int n, int half, void* stream);
int fluxhip_axpbypcz_f32(const float* x, const float* y, const float* z, float* out, int64_t n, float ca, float cb, float cc,
                         const float* coef, void* stream);
int fluxhip_softmax_rows_masked_x3(const float* s, void* p, int64_t p_lo, int64_t rows, int cols, int ld, float scale,
                                   int causal_T, void* stream);
int fluxhip_embedding_x3(const int* idx, const float* table, const float* pos, void* out, int64_t out_lo, int64_t n, int D,
                         int T, int V, void* stream);
int fluxhip_pixel_linear_x3_f32in(const float* x, const float* w, const float* bias, void* out, int64_t out_lo, int64_t npix,
                                  int Cin, int Cout, int Cpad, float in_div, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLUXHIP_H */
