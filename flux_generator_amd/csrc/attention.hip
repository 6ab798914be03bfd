// Non-causal flash attention, head_dim 128 (Flux joint txt+img attention) or 64 (SD/SDXL UNet
// self- and cross-attention), bf16 in / fp32 softmax / bf16 out.
// Replaces mx.fast.scaled_dot_product_attention(q, k, v, scale=D**-0.5) and the transpose/reshape
// after it (reference flux/layers.py:36-43; joint [txt;img] token order from flux/layers.py:212-214)
// and nn.MultiHeadAttention's softmax(q k^T / sqrt(d)) v (stable_diffusion/.../unet.py:46-54,64-71).
//
// MI355X design (64-lane waves, v_mfma_f32_32x32x16_bf16):
//  * one wave owns 32 query rows; NW waves per workgroup share the K / V^T tiles in LDS.
//  * both products are computed TRANSPOSED so that every per-query quantity is lane-local:
//        S^T[key][q] = K (A, from LDS) x Q^T (B, registers)        -> lane holds 16 keys of query (lane&31)
//        O^T[d][q]   = V^T (A, from LDS) x P^T (B, registers)       -> lane holds HD/2 d's of query (lane&31)
//    the S^T accumulator layout *is* the B-operand layout of the second product (up to a fixed key
//    permutation that is applied to the V^T fragment address instead), so P never leaves registers
//    and the online-softmax rescale of O needs no cross-lane traffic; row max / row sum need one
//    exchange with lane^32.
//  * K tile [64 keys][HD] and V^T tile [HD][64 keys] arrive by LDS-DMA (global_load_lds_dwordx4),
//    double-buffered, one barrier per KV tile. The LDS image is lane-linear, so the bank swizzle is
//    applied to the per-lane *source* address and again on the fragment reads
//    (256-B rows: 16-B chunk ^= row&15; 128-B rows: chunk ^= (row>>1)&7 -> conflict-free b128 reads).
//  * Q and K are addressed through (batch, head, row) strides, so both the head-major [B,H,T,HD]
//    buffers of the Flux path and token-major [B,T,H*HD] projections of the UNet path are read in
//    place; V^T ([B][H*HD][Tkpad], zero padded keys) comes from fluxhip_qk_norm_rope_bf16 (Flux) or
//    directly from the value-projection GEMM written transposed (UNet).
#include "../../include/fluxhip.h"
#include <cstdlib>
#include "common.h"

namespace {

constexpr int KV = 64;  // keys per tile

struct AttnParams {
  const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
  long long q_bs, q_hs, q_rs;   // element strides: batch, head, row
  long long k_bs, k_hs, k_rs;
  int ldo, H, Tq, Tk, Tkpad, nqb;
  long long vt_bs;          // element stride between the V^T images of two batches (H * HD * Tkpad when dense)
  float scale_log2;
  float scale;              // MODE 1: logits = scale * q.k + bias
  const bf16_t* bias;       // MODE 1: additive bias [H][Tq][Tk] (T5 relative position bias)
  int force_slow;           // attn128_w64_kernel: take the online-rescale fallback (tests)
  // MXO kernels: the output leaves as e4m3 bytes O8 [B * Tq][ldo] + E8M0 block scales in the tiled layout of the block-scaled
  // fp8 GEMM (include/fluxhip.h, fluxhip_fp8_mx): scale-buffer row = b * Tq + q, 32-column block = h * HD / 32 + db
  uint8_t* O8;
  uint8_t* mx;
  long long mx_kstride;
};

// MODE 0: plain; MODE 1: additive per-head bias (flux/t5.py:70-116,153-155: scale 1.0, bias passed as
// the SDPA mask); MODE 2: causal (CLIP text model, flux/clip.py:91-95: key index > query index masked)

// KS = 2: a second set of NW waves takes every other KV tile of the same 128 queries and the two partial
// (max, sum, O) states are merged through LDS at the end.  At batch 1 the Flux grid is 240 workgroups for
// 256 CUs: with KS = 1 that is one wave per SIMD, and the MFMA pipe idles through every softmax phase; with
// KS = 2 each SIMD holds two waves in different phases.
// VP = 1: the keys of V^T are stored permuted inside every aligned group of 16 as [0-3, 8-11, 4-7, 12-15] (written that
// way by fluxhip_qk_norm_rope_bf16): the 8 keys a lane feeds to one PV MFMA - (0-3, 8-11) for lanes 0-31, (4-7, 12-15)
// for lanes 32-63 - are then ONE 16-byte chunk, i.e. one conflict-free ds_read_b128 per fragment instead of two
// ds_read_b64 (whose 8-byte accesses were 2-way bank conflicted: 30 % of the kernel's LDS cycles in the round-1 PMC).
template <bool H>
DEVINL f32x16 mfma32(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// F16: the 16-bit storage type is IEEE float16 (stable_diffusion/ with float16=True) instead of bfloat16: the S and PV products
// run on v_mfma_f32_32x32x16_f16, P is rounded to float16 (<= 256 under the lazy rescale, far inside its range)
template <int HD, int NW, int MODE, int KS = 1, int VP = 0, bool F16 = false, bool MXO = false>
__global__ __launch_bounds__(NW * KS * 64, KS == 1 ? 2 : 1) void attn_kernel(const AttnParams p) {
  constexpr int RB = HD * 2;                  // bytes per K row
  constexpr int CPR = RB / 16;                // 16-B chunks per K row
  constexpr int KT_BYTES = KV * RB;           // K tile
  constexpr int VT_BYTES = HD * KV * 2;       // V^T tile (HD rows of 128 B)
  constexpr int STAGE = KT_BYTES + VT_BYTES;
  constexpr int NDS = HD / 16;                // d-steps of S^T
  constexpr int NDB = HD / 32;                // d-blocks of O^T
  constexpr int PPT_K = KT_BYTES / 1024;      // 1-KiB pieces per K tile
  constexpr int PPT_V = HD / 8;               // ... per V^T tile
  constexpr int KPW = PPT_K / NW;             // K pieces per wave (KS tiles over NW*KS waves)
  constexpr int VPW = PPT_V / NW;             // V^T pieces per wave
  constexpr int KROWS = 1024 / RB;            // K rows per 1-KiB piece

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int H = p.H, Tq = p.Tq, Tk = p.Tk, Tkpad = p.Tkpad;

  // XCD-aware map: all query blocks of one (b,h) land on the same XCD (they share K/V in its L2)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / p.nqb;
  const int qb = logical - bh * p.nqb;
  const int b = bh / H, h = bh - b * H;

  const bf16_t* Qh = p.Q + b * p.q_bs + h * p.q_hs;
  const bf16_t* Kh = p.K + b * p.k_bs + h * p.k_hs;
  const bf16_t* Vh = p.Vt + b * p.vt_bs + (long long)h * HD * Tkpad;

  const int qw = wave % NW, kp = wave / NW;   // query group of this wave, KV-tile parity
  const int q0 = qb * (NW * 32) + qw * 32;
  const int qrow = min(q0 + ql, Tq - 1);

  // Q^T fragments (B operand): lane (q = lane&31, hi) holds d = ds*16 + hi*8 .. +8
  bf16x8 qf[NDS];
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds)
    qf[ds] = *(const bf16x8*)(Qh + (long long)qrow * p.q_rs + ds * 16 + hi * 8);

  // staging sources
  // Round 5: every LDS-DMA piece is addressed as (wave-uniform 64-bit base of the STAGE) + (32-bit lane offset that does not
  // change from stage to stage) - the saddr form of global_load_lds.  The per-piece address used to be rebuilt from the key
  // index every time: min / sign-extend / two quarter-rate 32-bit multiplies / 64-bit multiply-add / shift-add per K piece and a
  // 64-bit compare / select chain per V^T piece, ~185 VALU + ~100 SALU instructions per KV tile and wave next to the ~130 the
  // softmax itself needs, in a kernel that is bound by instruction issue.  Only a stage that reaches past Tk /
  // Tkpad (the last one, when Tk is not a multiple of KS * 64) needs clamped per-lane keys: its offsets are rebuilt once, against
  // the head's base (uniform branch, never taken for the Flux shapes).  The launcher checks that a head's K / V^T image is < 4 GiB.
  const int kr = lane / CPR, kc = lane % CPR;   // K piece: KROWS rows x CPR chunks
  const int vr = lane >> 3, vc = lane & 7;      // V^T piece: 8 rows x 8 chunks
  // piece id = wave + i * (NW*KS): tile = id / pieces-per-tile, piece inside the tile = id % pieces-per-tile
  const uint32_t k_rowbytes = (uint32_t)(p.k_rs * 2), v_rowbytes = (uint32_t)Tkpad * 2u;
  auto k_piece = [&](int i, int& krow_, int& chunk_) {      // key row of this lane inside the stage, swizzled source chunk
    const int id = wave + i * (NW * KS);
    const int row = (id % PPT_K) * KROWS + kr;
    krow_ = row + (id / PPT_K) * KV;
    chunk_ = kc ^ (HD == 128 ? (row & 15) : ((row >> 1) & 7));
  };
  auto v_piece = [&](int i, int& d_, int& chunk_, int& tile_) {
    const int id = wave + i * (NW * KS);
    d_ = (id % PPT_V) * 8 + vr;
    chunk_ = vc ^ ((d_ >> 1) & 7);
    tile_ = id / PPT_V;
  };
  uint32_t koffs[KPW], voffs[VPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    int kr_, ch_;
    k_piece(i, kr_, ch_);
    koffs[i] = (uint32_t)kr_ * k_rowbytes + ch_ * 16;
  }
#pragma unroll
  for (int i = 0; i < VPW; ++i) {
    int d_, ch_, t_;
    v_piece(i, d_, ch_, t_);
    voffs[i] = (uint32_t)d_ * v_rowbytes + ch_ * 16 + t_ * (KV * 2);
  }
  const char* kbase = (const char*)Kh;          // wave-uniform bases of the stage being issued (stage_addr)
  const char* vbase = (const char*)Vh;
  auto stage_addr = [&](int it) {
    const int key0 = it * (KS * KV);
    if (key0 + KS * KV <= Tk) {
      kbase = (const char*)Kh + (long long)key0 * p.k_rs * 2;
      vbase = (const char*)Vh + (long long)key0 * 2;
    } else {                                    // keys past Tk - 1 re-read row Tk - 1 (masked later); V^T tiles stay inside the row
      kbase = (const char*)Kh;
      vbase = (const char*)Vh;
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        int kr_, ch_;
        k_piece(i, kr_, ch_);
        koffs[i] = (uint32_t)min(key0 + kr_, Tk - 1) * k_rowbytes + ch_ * 16;
      }
#pragma unroll
      for (int i = 0; i < VPW; ++i) {
        int d_, ch_, t_;
        v_piece(i, d_, ch_, t_);
        // (a V^T tile past Tkpad is only ever read for fully masked keys; clamp the source inside the row)
        voffs[i] = (uint32_t)d_ * v_rowbytes + ch_ * 16 + (uint32_t)min(key0 + t_ * KV, Tkpad - KV) * 2;
      }
    }
  };
  // one stage = KS consecutive KV tiles: [KS][K tile | V^T tile]
  // [p0, p1): which of this wave's KPW + VPW pieces to issue (the main loop issues one between MFMAs); stage_addr(it) first
  auto stage = [&](int buf, int p0 = 0, int p1 = 64) {
    char* s0 = smem + buf * (KS * STAGE);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      if (i < p0 || i >= p1) continue;
      const int id = wave + i * (NW * KS);
      glds16(kbase + (size_t)koffs[i], s0 + (id / PPT_K) * STAGE + (id % PPT_K) * 1024);
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
      if (KPW + i < p0 || KPW + i >= p1) continue;
      const int id = wave + i * (NW * KS);
      glds16(vbase + (size_t)voffs[i], s0 + (id / PPT_V) * STAGE + KT_BYTES + (id % PPT_V) * 1024);
    }
  };

  f32x16 oT[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // fragment read offsets
  // K (A operand): row = kb*32 + ql, logical chunk = ds*2 + hi, phys = chunk ^ swz(row)
  const int k_rowoff = ql * RB;
  const int k_sw = (HD == 128) ? (ql & 15) : ((ql >> 1) & 7);
  // V^T (A operand): row d = db*32 + ql ; bytes: ((kb*4 + j*2 + {0,1}) ^ ((d>>1)&7))*16 + hi*8
  const int v_rowoff = ql * 128 + (VP ? 0 : hi * 8);
  const int v_sw = (ql >> 1) & 7;
  // LDS byte offsets of the fragments inside a stage (the swizzle is an XOR, so one register per chunk)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t koff[NDS], voff[8];
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) koff[ds] = k_rowoff + (((ds * 2 + hi) ^ k_sw) << 4);
#pragma unroll
  for (int c = 0; c < 8; ++c) voff[c] = KT_BYTES + v_rowoff + (((VP ? (c | hi) : c) ^ v_sw) << 4);   // VP: only even c are used

  int ntiles = (Tk + KV - 1) / KV;
  // causal: keys beyond the workgroup's last query row are all masked (block-uniform trip count:
  // every wave must reach the same barriers)
  if (MODE == 2) ntiles = min(ntiles, min(qb * (NW * 32) + NW * 32 - 1, Tq - 1) / KV + 1);
  const float scale_log2 = (MODE == 1) ? 1.4426950408889634f : p.scale_log2;
  const bf16_t* bias_q = (MODE == 1) ? p.bias + ((long long)h * Tq + qrow) * Tk : nullptr;
  stage_addr(0);
  stage(0);
  wait_vm0();
  __syncthreads();

  const int nstages = (ntiles + KS - 1) / KS;
  // Static priority for the second-dispatched wave set (MI355X guide, "two waves per SIMD", item 4): the younger wave of a
  // SIMD loses every VALU arbitration to its older partner; one s_setprio for that half before the loop (no per-phase
  // flips) lets it take the older half's timing.  FLUXHIP_ATTN_PRIO=0 at build time of this experiment removes it.
  if (KS == 2 && kp == 1) __builtin_amdgcn_s_setprio(1);
  for (int st = 0; st < nstages; ++st) {
    const int cur = st & 1;
    const bool more = st + 1 < nstages;
    static_assert(KPW + VPW == NDS, "one LDS-DMA piece per d-step of the S product");
    if (more) stage_addr(st + 1);
    if (more && !(KS == 1 || st * KS + kp < ntiles)) stage(cur ^ 1);   // idle wave set: no MFMAs to hide behind
    const int it = st * KS + kp;                 // this wave's KV tile
    const char* sk = smem + cur * (KS * STAGE) + kp * STAGE;
    const char* sv = sk + KT_BYTES;
    if (KS == 1 || it < ntiles) {

    // ---- S^T = K Q^T -------------------------------------------------------
    // All K fragments are requested up front (inline asm: hipcc would wait for each read right before
    // its MFMA) and the two key blocks alternate, so no MFMA waits on the accumulator of the previous one.
    const uint32_t sbase = lds0 + cur * (KS * STAGE) + kp * STAGE;
    bf16x8 kf[NDS][2];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ds][kb]) : "v"(sbase + koff[ds]), "n"(kb * 32 * RB) : "memory");
    f32x16 sT[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[kb][r] = 0.f;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NDS - 1 - ds) > 15 ? 15 : 2 * (NDS - 1 - ds)) : "memory");
      __builtin_amdgcn_sched_barrier(0);            // (MFMAs have no memory operands: keep them behind the wait)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        sT[kb] = mfma32<F16>(kf[ds][kb], qf[ds], sT[kb]);
      // next stage's K / V^T pieces, one per d-step: issued back to back after the barrier they cost every
      // wave ~1000 cycles per stage in the address path with the MFMA pipe idle
      __builtin_amdgcn_sched_barrier(0);
      if (more) stage(cur ^ 1, ds, ds + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // V^T fragments of key block 0 travel while the softmax runs (P register pair 8j+e <-> k-slot e)
    union VFrag { bf16x8 v; u32x2 h[2]; };
    VFrag vf0[2][NDB], vf1[2][NDB];
    auto read_v = [&](VFrag(&vf)[2][NDB], int kb) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          if constexpr (VP) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[j][db].v) : "v"(sbase + voff[kb * 4 + j * 2]), "n"(db * 32 * 128) : "memory");
          } else {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(vf[j][db].h[0]) : "v"(sbase + voff[kb * 4 + j * 2]), "n"(db * 32 * 128) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(vf[j][db].h[1]) : "v"(sbase + voff[kb * 4 + j * 2 + 1]), "n"(db * 32 * 128) : "memory");
          }
        }
    };
    read_v(vf0, 0);
    const int key0 = it * KV;
    if (MODE == 1) {   // logits = scale * s + bias[h][q][key]; 4 consecutive keys per 8-byte load
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int key = key0 + kb * 32 + 8 * rg + 4 * hi;
          float b4[4] = {0.f, 0.f, 0.f, 0.f};
          if (key + 3 < Tk) {
            u32x2 bw = *(const u32x2*)(bias_q + key);
            b4[0] = e_lo<F16>(bw[0]); b4[1] = e_hi<F16>(bw[0]); b4[2] = e_lo<F16>(bw[1]); b4[3] = e_hi<F16>(bw[1]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (key + e < Tk) b4[e] = e2f<F16>(bias_q[key + e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) sT[kb][rg * 4 + e] = fmaf(sT[kb][rg * 4 + e], p.scale, b4[e]);
        }
    }
    if (MODE == 2) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key > q0 + ql) sT[kb][r] = -1e30f;
        }
    }
    // ---- mask the key tail (last tile only) ----------------------------------
    if (key0 + KV > Tk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= Tk) sT[kb][r] = -1e30f;
        }
    }
    // ---- online softmax (per query = per lane pair (l, l^32)) -----------------
    float mx = sT[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kb][r]);
    mx = pair32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    // Lazy rescale: softmax is shift-invariant, so the running reference only has to keep exp2 in range.
    // It moves when some query's maximum grew by more than 8 in the log2 domain (P <= 2^8 otherwise);
    // on typical logits that is the first tile or two, instead of nearly every tile.
    if (__any((m_new - m_run) * scale_log2 > 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
      m_run = m_new;
    }
    const float mneg = -m_run * scale_log2;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(fmaf(sT[kb][r], scale_log2, mneg));
        sT[kb][r] = pv;
        psum += pv;
      }
    l_run += psum;

    // ---- O^T += V^T P^T -------------------------------------------------------
    auto pv = [&](const VFrag(&vf)[2][NDB], int kb) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pf.u[e] = e_pack<F16>(sT[kb][8 * j + 2 * e], sT[kb][8 * j + 2 * e + 1]);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
          oT[db] = mfma32<F16>(vf[j][db].v, pf.v, oT[db]);
      }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_v(vf1, 1);                                   // lands under the MFMAs of key block 0
    pv(vf0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pv(vf1, 1);
    }
    wait_vm0();
    __syncthreads();
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l for the d-blocks [LO, HI) of this wave ---------------------
  // Stores: a lane holds 4 consecutive d of its query per register group, its partner (lane ^ 32) the next 4; one
  // v_permlane32_swap per dword and group PAIR gives every lane 16 contiguous bytes - half the store instructions of the
  // 8-byte form (the store tail of this layout is issue-bound: MI355X guide, T21).
  auto finalize = [&](auto LOc, auto HIc) {
    constexpr int LO = decltype(LOc)::value, HI = decltype(HIc)::value;
    const float l_tot = pair32_sum(l_run);
    const float inv = 1.f / l_tot;
    const int q = q0 + ql;
    if constexpr (MXO) {
      // e4m3 + one E8M0 scale per 32 output columns: block db of query q is the 16 values of this lane and the 16 of lane ^ 32
      // (the arithmetic of the FLAG_MXC GEMM epilogue: 2^e = smallest power of two with max|v| / 2^e <= 448)
      const long long grow = (long long)b * Tq + min(q, Tq - 1);
      uint8_t* orow8 = p.O8 + grow * p.ldo + h * HD;
#pragma unroll
      for (int db = LO; db < HI; ++db) {
        float v[16];
        float am = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[r] = oT[db][r] * inv; am = fmaxf(am, fabsf(v[r])); }
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const uint32_t ab = __builtin_bit_cast(uint32_t, am);
        int e8 = (int)(ab >> 23) - 8 + (int)((ab & 0x7fffffu) > 0x600000u);
        e8 = min(max(e8, 1), 253);
        const float mul = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);
        if (q < Tq) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[rg * 4 + 0] * mul, v[rg * 4 + 1] * mul, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[rg * 4 + 2] * mul, v[rg * 4 + 3] * mul, w, true);
            *(uint32_t*)(orow8 + db * 32 + 8 * rg + 4 * hi) = (uint32_t)w;
          }
          if (hi == 0) {
            const int kb = h * (HD / 32) + db;
            p.mx[((((long long)(kb >> 2) * p.mx_kstride + (grow >> 6) * 64 + (kb & 3) * 16 + (grow & 15)) << 2) + ((grow >> 4) & 3))] = (uint8_t)e8;
          }
        }
      }
    } else {
      bf16_t* orow = p.O + ((long long)b * Tq + min(q, Tq - 1)) * p.ldo + h * HD;
      const bool wide_o = ((p.ldo & 7) == 0) && (((uintptr_t)p.O & 15) == 0);      // 16-byte addressable rows (wave-uniform)
      if (!wide_o) {                             // the interface only asks for ldo % 4 == 0: 8-byte stores from the MFMA layout
#pragma unroll
        for (int db = LO; db < HI; ++db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            u32x2 o;
            o[0] = e_pack<F16>(oT[db][rg * 4 + 0] * inv, oT[db][rg * 4 + 1] * inv);
            o[1] = e_pack<F16>(oT[db][rg * 4 + 2] * inv, oT[db][rg * 4 + 3] * inv);
            if (q < Tq) *(u32x2*)(orow + db * 32 + 8 * rg + 4 * hi) = o;
          }
        return;
      }
#pragma unroll
      for (int db = LO; db < HI; ++db)
#pragma unroll
        for (int rg = 0; rg < 4; rg += 2) {
          u32x2 oa, ob;                          // groups rg (d = 8 rg + 4 hi ..) and rg + 1 of this lane
          oa[0] = e_pack<F16>(oT[db][rg * 4 + 0] * inv, oT[db][rg * 4 + 1] * inv);
          oa[1] = e_pack<F16>(oT[db][rg * 4 + 2] * inv, oT[db][rg * 4 + 3] * inv);
          ob[0] = e_pack<F16>(oT[db][rg * 4 + 4] * inv, oT[db][rg * 4 + 5] * inv);
          ob[1] = e_pack<F16>(oT[db][rg * 4 + 6] * inv, oT[db][rg * 4 + 7] * inv);
          // lanes 0-31 end up with [own group rg | partner's group rg], lanes 32-63 with [partner's group rg + 1 | own group rg + 1]
          const auto s0 = __builtin_amdgcn_permlane32_swap(oa[0], ob[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(oa[1], ob[1], false, false);
          if (q < Tq) *(u32x4*)(orow + db * 32 + 8 * (rg + hi)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
    }
  };
  using D0 = std::integral_constant<int, 0>;
  using DH = std::integral_constant<int, NDB / 2>;
  using DN = std::integral_constant<int, NDB>;

  // ---- KS = 2: the two wave sets hold partial (max, sum, O^T) states of the same queries over the even / odd KV tiles.
  // Each set FINISHES HALF of the head dimension: it hands the other half of its O^T (and its max / sum) to its partner through
  // LDS in 16-byte pieces, merges the half it keeps and stores it - both sets work, each moves 34 values per lane instead of
  // one set writing 66 and the other reading them back one dword at a time (round 5; the tail of a 20-tile workgroup at T = 1280
  // is a tenth of its time).
  if constexpr (KS == 2) {
    static_assert(NDB % 2 == 0, "two wave sets split the head dimension in halves");
    constexpr int HB = NDB / 2, SLOTS = HB * 4 + 1;          // f32x4 slots per lane: HB d-blocks of 16 values + (max, sum)
    f32x4* const mine = (f32x4*)smem + (size_t)((qw * 2 + kp) * SLOTS) * 64 + lane;        // [qw][writer set][slot][lane], after the last barrier
    const f32x4* const theirs = (const f32x4*)smem + (size_t)((qw * 2 + (1 - kp)) * SLOTS) * 64 + lane;
    auto hand_over = [&](auto GIVEc) {
      constexpr int GIVE = decltype(GIVEc)::value;
#pragma unroll
      for (int i = 0; i < HB; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          mine[(i * 4 + r4) * 64] = f32x4{oT[GIVE + i][r4 * 4], oT[GIVE + i][r4 * 4 + 1], oT[GIVE + i][r4 * 4 + 2], oT[GIVE + i][r4 * 4 + 3]};
      mine[(HB * 4) * 64] = f32x4{m_run, l_run, 0.f, 0.f};
    };
    if (kp == 0) hand_over(DH{}); else hand_over(D0{});      // set 0 keeps the low half of d, set 1 the high half
    __syncthreads();
    const f32x4 ml = theirs[(HB * 4) * 64];
    const float m_new = fmaxf(m_run, ml[0]);
    const float fa = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);
    const float fb = __builtin_amdgcn_exp2f((ml[0] - m_new) * scale_log2);
    l_run = l_run * fa + ml[1] * fb;
    auto take = [&](auto KEEPc) {
      constexpr int KEEP = decltype(KEEPc)::value;
#pragma unroll
      for (int i = 0; i < HB; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 t = theirs[(i * 4 + r4) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) oT[KEEP + i][r4 * 4 + e] = oT[KEEP + i][r4 * 4 + e] * fa + t[e] * fb;
        }
    };
    if (kp == 0) { take(D0{}); finalize(D0{}, DH{}); }
    else { take(DH{}); finalize(DH{}, DN{}); }
  } else {
    finalize(D0{}, DN{});
  }
}


// -------------------------------------------------------------------------------------------------------------------
// head_dim 128, 64 QUERIES PER WAVE, one wave per SIMD, software-pipelined by program order (round 3).
// Same transposed products, LDS-DMA staging, swizzles and key-permuted V^T layout as attn_kernel above; what changes is the
// work per wave and who overlaps with whom.  r02 counters of attn_kernel at T = 1280: matrix pipe busy 26 %, the MFMA and
// VALU instruction classes hardly overlapping — two lock-stepped waves per SIMD, each a serial chain S -> softmax -> PV.
//  * a wave owns TWO 32-query blocks (A, B) and the whole 512-entry register file of its SIMD (4 waves per workgroup,
//    __launch_bounds__(256, 1)), and overlaps their chains IN PROGRAM ORDER — an in-order wave only overlaps what is
//    interleaved instruction by instruction, so every MFMA of a phase is followed by one slice of the other block's work:
//        phase 1   S_A            16 MFMAs   | K fragment reads (8 ahead), LDS-DMA pieces of the next stage
//        phase 2   S_B            16 MFMAs   | softmax_A in 16 slices, K re-reads, V^T fragment reads
//        -- lgkmcnt(0), vmcnt(0), barrier: this stage's LDS buffer is dead, the next stage is visible --
//        phase 3   PV_A           16 MFMAs   | softmax_B in 16 slices
//        phase 4   PV_B           16 MFMAs   | first 8 K fragments of this wave's NEXT tile
//    V^T fragments (64 registers) are read once and feed both blocks; K fragments go through an 8-deep register ring and
//    are read twice (the LDS has the bandwidth, the arch VGPRs do not have the room for all 16).
//  * register classes are pinned with inline-asm MFMAs (hipcc, left alone, parks the S accumulators in AccVGPRs and moves
//    hundreds of registers per tile across the files): S accumulators / P in arch VGPRs (the softmax is VALU work), O^T
//    accumulators (128 registers) and Q^T fragments (64) in AccVGPRs.  An asm MFMA is opaque to hipcc's hazard recogniser:
//    the required wait states (MFMA result -> VALU read, VALU write -> MFMA operand) are s_nop's inside the asm strings.
//  * no O rescale in the loop: VALU code touching the O accumulators makes hipcc carry them in arch VGPRs (128 copies per
//    tile).  Softmax is shift-invariant, so the reference of a query is simply the row maximum of its FIRST tile and never
//    moves; exp2 then overflows only if a later score exceeds it by > ~100 in the log2 domain (a raw logit gap of ~800 at
//    head_dim 128).  That case is DETECTED (sticky flag, block-wide OR through LDS) and the workgroup recomputes its queries
//    with the plain online-rescale loop (`slow`, compiler-scheduled): correct for any input, never taken on sane logits.
//  * KS = 2 (small grids, e.g. batch 1 at 512 x 512: 24 heads x 10 blocks of 128 queries = 240 workgroups): waves 0-1 take
//    the even KV tiles of the workgroup's 128 queries, waves 2-3 the odd ones, merged through LDS at the end.  KS = 1:
//    256 queries per workgroup.
DEVINL void mfma_s0(f32x16& acc, const bf16x8& a, const bf16x8& b) {     // first MFMA of a chain: C = 0
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b));
}
DEVINL void mfma_s(f32x16& acc, const bf16x8& a, const bf16x8& b) {      // acc: arch VGPRs; b (Q^T fragment): AccVGPRs
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
DEVINL void mfma_o(f32x16& acc, const bf16x8& a, const bf16x8& b) {      // acc (O^T): AccVGPRs; b = freshly packed P
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// max without the canonicalising v_max x, x that fmaxf() puts in front of values hipcc cannot prove quiet (asm MFMA outputs)
DEVINL float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEVINL float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// MFMA result (arch VGPRs) -> first VALU read: 16-pass worst case 19 wait states (gfx940 hazard table)
DEVINL void mfma_settle(f32x16& a, f32x16& b) { asm("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }

template <int KS>
__global__ __launch_bounds__(256, 1) void attn128_w64_kernel(const AttnParams p) {
  constexpr int HD = 128, NW = 4, NQW = NW / KS;      // query-owning waves per workgroup
  constexpr int RB = HD * 2, KT_BYTES = KV * RB, VT_BYTES = HD * KV * 2, STAGE = KT_BYTES + VT_BYTES;
  constexpr int NDS = HD / 16, NDB = HD / 32;
  constexpr int NPIECE = 8 * KS;                      // LDS-DMA pieces per wave and stage (KS tiles x 32 pieces over 4 waves)
  constexpr int FLAG_OFF = 2 * KS * STAGE;            // one LDS word after the stage buffers: block-wide overflow flag

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int H = p.H, Tq = p.Tq, Tk = p.Tk, Tkpad = p.Tkpad;

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / p.nqb;
  const int qb = logical - bh * p.nqb;
  const int b = bh / H, h = bh - b * H;
  // dense head-major operands (fluxhip_attention_d128_bf16): Q, K [B][H][T][128], V^T [B][H][128][Tkpad]
  const bf16_t* Qh = p.Q + ((long long)bh * Tq) * HD;
  const char* Kh = (const char*)(p.K + ((long long)bh * Tk) * HD);
  const char* Vh = (const char*)(p.Vt + (long long)bh * HD * Tkpad);

  const int qw = wave % NQW, kp = wave / NQW;
  const int q0 = qb * (NQW * 64) + qw * 64;           // first query of this wave; block x covers q0 + 32 x + [0, 32)

  // staging: piece id = wave + 4 i (i < NPIECE / 2 for K, then for V^T) over the KS tiles of a stage.
  //   K piece   = 4 key rows x 16 chunks: row = 16 (i & 3) + 4 wave + (lane >> 4)  ->  row & 15 is a lane constant
  //   V^T piece = 8 d rows x 8 chunks:    d   = 32 (i & 3) + 8 wave + (lane >> 3)  ->  (d >> 1) & 7 is a lane constant
  const int krl = 4 * wave + (lane >> 4);
  const uint32_t k_lane = (uint32_t)krl * RB + (uint32_t)(((lane & 15) ^ krl) << 4);            // byte offset inside 16 key rows
  const int dl = 8 * wave + (lane >> 3);
  const char* v_lane = Vh + (long long)dl * Tkpad * 2 + (((lane & 7) ^ ((dl >> 1) & 7)) << 4);
  auto stage1 = [&](int st_next, int buf, int i) {     // piece i of this wave for stage st_next -> LDS buffer buf
    char* s0 = smem + buf * (KS * STAGE);
    const int key0 = st_next * (KS * KV);
    if (i < NPIECE / 2) {
      const int tile = i >> 2, sub = i & 3;
      const int key = min(key0 + tile * KV + 16 * sub + krl, Tk - 1);       // (clamped rows are masked or never used)
      glds16(Kh + (long long)key * RB + (k_lane - (uint32_t)krl * RB), s0 + tile * STAGE + (wave + 4 * sub) * 1024);
    } else {
      const int j = i - NPIECE / 2, tile = j >> 2, sub = j & 3;
      const long long koff = min((long long)key0 + tile * KV, (long long)Tkpad - KV);
      glds16(v_lane + (long long)(32 * sub) * Tkpad * 2 + koff * 2, s0 + tile * STAGE + KT_BYTES + (wave + 4 * sub) * 1024);
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // fragment addresses inside a tile: K: row kb * 32 + ql, chunk (2 ds + hi) ^ (ql & 15); V^T: row db * 32 + ql, chunk (4 kb + 2 j) | hi
  uint32_t koff[NDS], voff[4];
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) koff[ds] = ql * RB + (((ds * 2 + hi) ^ (ql & 15)) << 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) voff[c] = KT_BYTES + ql * 128 + ((((2 * c) | hi) ^ ((ql >> 1) & 7)) << 4);

  const int ntiles = (Tk + KV - 1) / KV;
  const float scale_log2 = p.scale_log2;
  const int nstages = (ntiles + KS - 1) / KS;

  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  f32x16 oT[2][NDB];
  bool ovf = p.force_slow != 0;

  if (!ovf) {
    // ================================================== fast path =======================================================
    bf16x8 qf[2][NDS];                     // Q^T fragments of both query blocks (AccVGPRs: only ever MFMA B operands)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int qrow = min(q0 + 32 * x + ql, Tq - 1);
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) qf[x][ds] = *(const bf16x8*)(Qh + (long long)qrow * HD + ds * 16 + hi * 8);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[x][i][r] = 0.f;

    bf16x8 kw[8], vf[4][NDB];              // K fragment ring (fragment f = ds * 2 + kb sits in slot f & 7), V^T fragments [kb * 2 + j][db]
    f32x16 sT[2][2];                       // [query block][key block]: 16 keys of query (lane & 31) per key block
    bf16x8 pf[2][4];                       // packed P^T fragments [query block][kb * 2 + j]
    float pm[2][4], mneg[2], psum[2], excess[2] = {0.f, 0.f};
    auto read_k = [&](uint32_t sbase, auto fc) {          // K fragment f of the tile at sbase -> ring slot f & 7
      constexpr int f = decltype(fc)::value & 15, ds = f >> 1, kb = f & 1;
      bf16x8& dst = kw[f & 7];                            // (named outside the asm: operands of an asm statement inside a generic
      const uint32_t addr = sbase + koff[ds];             //  lambda do not count as captures for clang)
      if constexpr (kb == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(32 * RB) : "memory");
    };
    auto read_v = [&](uint32_t sbase, auto ic) {          // V^T fragment i = c * 4 + db
      constexpr int i = decltype(ic)::value, c = i >> 2, db = i & 3;
      bf16x8& dst = vf[c][db];
      const uint32_t addr = sbase + voff[c];
      if constexpr (db == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(db * 32 * 128) : "memory");
    };
    // softmax of one query block in 16 slices (one per MFMA gap of the phase it hides in)
    auto sm_slice = [&](auto xc, auto ic, auto firstc, auto lastc, int key0) {
      constexpr int x = decltype(xc)::value, i = decltype(ic)::value;
      if constexpr (i < 4) {               // partial maxima over 8 scores each (key-tail mask first: last stage only)
        constexpr int kb = i >> 1, r0 = (i & 1) * 8;
        if constexpr (decltype(lastc)::value) {
          if (key0 + KV > Tk) {
#pragma unroll
            for (int r = r0; r < r0 + 8; ++r)
              if (key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Tk) sT[x][kb][r] = -1e30f;
          }
        }
        const float m0 = vmax3(sT[x][kb][r0], sT[x][kb][r0 + 1], sT[x][kb][r0 + 2]);
        const float m1 = vmax3(sT[x][kb][r0 + 3], sT[x][kb][r0 + 4], sT[x][kb][r0 + 5]);
        pm[x][i] = vmax3(vmax(sT[x][kb][r0 + 6], sT[x][kb][r0 + 7]), m0, m1);
      } else if constexpr (i == 4) {       // row maximum: reference of the query (first tile) / overflow watch (later tiles)
        const float mh = vmax(vmax(pm[x][0], pm[x][1]), vmax(pm[x][2], pm[x][3]));
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mh), __float_as_uint(mh), false, false);
        const float mx = vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if constexpr (decltype(firstc)::value) m_run[x] = mx;
        else excess[x] = vmax(excess[x], mx - m_run[x]);       // branch-free: judged once, after the loop
        mneg[x] = -m_run[x] * scale_log2;
        psum[x] = 0.f;
      } else if constexpr (i < 13) {       // 8 slices of 4 scores: p = exp2(s * scale - m)
        constexpr int e0 = (i - 5) * 4, kb = e0 >> 4, r0 = e0 & 15;
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(sT[x][kb][r], scale_log2, mneg[x]));
          sT[x][kb][r] = pv;
          psum[x] += pv;
        }
      } else if constexpr (i < 15) {       // pack to bf16: fragment c = kb * 2 + j holds P[kb][8 j .. 8 j + 7]
#pragma unroll
        for (int c = (i - 13) * 2; c < (i - 13) * 2 + 2; ++c) {
          union { bf16x8 v; uint32_t u[4]; } t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t.u[e] = pack_bf16x2(sT[x][c >> 1][8 * (c & 1) + 2 * e], sT[x][c >> 1][8 * (c & 1) + 2 * e + 1]);
          pf[x][c] = t.v;
        }
      } else {
        l_run[x] += psum[x];
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    constexpr std::integral_constant<int, 0> XA{};
    constexpr std::integral_constant<int, 1> XB{};

    // one stage of this wave: its tile `it`; FIRST: first tile of the wave, MORE: another stage follows
    auto body = [&](int st, auto firstc, auto morec) {
      constexpr bool MORE = decltype(morec)::value;
      using LAST = std::integral_constant<bool, !MORE>;
      const int cur = st & 1;
      const int it = st * KS + kp;
      const bool active = KS == 1 || it < ntiles;
      const uint32_t sbase = lds0 + cur * (KS * STAGE) + kp * STAGE;
      const uint32_t sbn = lds0 + (cur ^ 1) * (KS * STAGE) + kp * STAGE;
      const int key0 = it * KV;
      if (active) {
        // ---- phase 1: S_A | K ring refills, LDS-DMA pieces of the next stage ----------------------------------------
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, ds = i >> 1, kb = i & 1;
          asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");         // fragment i has landed (7 younger reads in flight)
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ds == 0) mfma_s0(sT[0][kb], kw[i & 7], qf[0][ds]);
          else mfma_s(sT[0][kb], kw[i & 7], qf[0][ds]);
          read_k(sbase, std::integral_constant<int, i + 8>{});        // i < 8: fragments 8..15; then fragments 0..7 again (S_B)
          if constexpr (MORE && (KS == 2 || (i & 1) == 0)) stage1(st + 1, cur ^ 1, KS == 2 ? i : i >> 1);
          __builtin_amdgcn_sched_barrier(0);
        });
        mfma_settle(sT[0][0], sT[0][1]);
        // ---- phase 2: S_B | softmax_A, K re-reads (fragments 8..15), V^T fragment reads ----------------------------------
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, ds = i >> 1, kb = i & 1;
          // LDS ops younger than the read of fragment i: see the schedule above (K, then V^T, per gap)
          constexpr int younger = i < 8 ? 7 + i : 23 - i;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger > 15 ? 15 : younger) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ds == 0) mfma_s0(sT[1][kb], kw[i & 7], qf[1][ds]);
          else mfma_s(sT[1][kb], kw[i & 7], qf[1][ds]);
          if constexpr (i < 8) read_k(sbase, std::integral_constant<int, i + 8>{});
          read_v(sbase, ic);
          sm_slice(XA, ic, firstc, LAST{}, key0);
          __builtin_amdgcn_sched_barrier(0);
        });
        mfma_settle(sT[1][0], sT[1][1]);
      } else if constexpr (MORE) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage1(st + 1, cur ^ 1, i);    // idle wave set (odd tile count): just stage
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // V^T fragments are in registers:
      wait_vm0();                                                        // ... this stage's LDS buffer is dead for this wave, and
      __syncthreads();                                                   // after the barrier for everybody; the next stage is visible
      if (active) {
        // ---- phase 3: PV_A | softmax_B ----------------------------------------------------------------------------------
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, c = i >> 2, db = i & 3;
          mfma_o(oT[0][db], vf[c][db], pf[0][c]);
          sm_slice(XB, ic, firstc, LAST{}, key0);
          __builtin_amdgcn_sched_barrier(0);
        });
        // ---- phase 4: PV_B | first 8 K fragments of this wave's next tile ------------------------------------------------
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, c = i >> 2, db = i & 3;
          mfma_o(oT[1][db], vf[c][db], pf[1][c]);
          if constexpr (MORE && i < 8) read_k(sbn, ic);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    };

#pragma unroll
    for (int i = 0; i < NPIECE; ++i) stage1(0, 0, i);
    wait_vm0();
    __syncthreads();
    static_for<0, 8>([&](auto ic) { read_k(lds0 + kp * STAGE, ic); });
    if (nstages == 1) {
      body(0, T_{}, F_{});
    } else {
      body(0, T_{}, T_{});
      for (int st = 1; st + 1 < nstages; ++st) body(st, F_{}, T_{});
      body(nstages - 1, F_{}, F_{});
    }
    // O^T accumulators: last MFMA -> first read; outstanding K prefetches of a tile that does not exist
    asm volatile("s_nop 15\n\ts_nop 3\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    // block-wide overflow verdict (uniform: every wave takes the same path below)
    int* flag = (int*)(smem + FLAG_OFF);
    if (tid == 0) *flag = 0;
    __syncthreads();
    ovf = vmax(excess[0], excess[1]) * scale_log2 > 100.0f;
    if (__any(ovf) && lane == 0) atomicOr(flag, 1);
    __syncthreads();
    ovf = *flag != 0;
    __syncthreads();
  }

  if (ovf) {
    // ================================================== slow path =======================================================
    // plain online softmax with rescale (the algorithm of attn_kernel, two query blocks per wave), compiler-scheduled
    m_run[0] = m_run[1] = -1e30f;
    l_run[0] = l_run[1] = 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[x][i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) stage1(0, 0, i);
    wait_vm0();
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
      const int cur = st & 1;
      const int it = st * KS + kp;
      if (st + 1 < nstages) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage1(st + 1, cur ^ 1, i);
      }
      if (KS == 1 || it < ntiles) {
        const char* sk = smem + cur * (KS * STAGE) + kp * STAGE;
        const int key0 = it * KV;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int qrow = min(q0 + 32 * x + ql, Tq - 1);
          f32x16 s2[2];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s2[kb][r] = 0.f;
#pragma unroll
          for (int ds = 0; ds < NDS; ++ds) {
            const bf16x8 qv = *(const bf16x8*)(Qh + (long long)qrow * HD + ds * 16 + hi * 8);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              const bf16x8 kv = *(const bf16x8*)(sk + koff[ds] + kb * 32 * RB);
              s2[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv, qv, s2[kb], 0, 0, 0);
            }
          }
          float mx = -1e30f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Tk) s2[kb][r] = -1e30f;
              mx = fmaxf(mx, s2[kb][r]);
            }
          mx = pair32_max(mx);
          const float m_new = fmaxf(m_run[x], mx);
          const float alpha = __builtin_amdgcn_exp2f((m_run[x] - m_new) * scale_log2);
          m_run[x] = m_new;
          l_run[x] *= alpha;
#pragma unroll
          for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[x][i][r] *= alpha;
          float ps = 0.f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float pv = __builtin_amdgcn_exp2f((s2[kb][r] - m_new) * scale_log2);
              s2[kb][r] = pv;
              ps += pv;
            }
          l_run[x] += ps;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            union { bf16x8 v; uint32_t u[4]; } t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t.u[e] = pack_bf16x2(s2[c >> 1][8 * (c & 1) + 2 * e], s2[c >> 1][8 * (c & 1) + 2 * e + 1]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
              const bf16x8 vv = *(const bf16x8*)(sk + voff[c] + db * 32 * 128);
              oT[x][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vv, t.v, oT[x][db], 0, 0, 0);
            }
          }
        }
      }
      wait_vm0();
      __syncthreads();
    }
  }

  // ---- KS = 2: fold the odd-tile state into the even-tile wave of the same queries ----------------------------------
  if (KS == 2) {
    constexpr int NREG = 2 * (NDB * 16 + 2);
    float* mrg = (float*)smem + (size_t)qw * NREG * 64 + lane;      // [qw][reg][lane], after the last barrier
    if (kp == 1) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) mrg[(x * (NDB * 16 + 2) + i * 16 + r) * 64] = oT[x][i][r];
        mrg[(x * (NDB * 16 + 2) + NDB * 16) * 64] = m_run[x];
        mrg[(x * (NDB * 16 + 2) + NDB * 16 + 1) * 64] = l_run[x];
      }
    }
    __syncthreads();
    if (kp == 1) return;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float m_b = mrg[(x * (NDB * 16 + 2) + NDB * 16) * 64], l_b = mrg[(x * (NDB * 16 + 2) + NDB * 16 + 1) * 64];
      const float m_new = fmaxf(m_run[x], m_b);
      const float fa = __builtin_amdgcn_exp2f((m_run[x] - m_new) * scale_log2);
      const float fb = __builtin_amdgcn_exp2f((m_b - m_new) * scale_log2);
      l_run[x] = l_run[x] * fa + l_b * fb;
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[x][i][r] = oT[x][i][r] * fa + mrg[(x * (NDB * 16 + 2) + i * 16 + r) * 64] * fb;
    }
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l -------------------------------------------------------------------------------
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float inv = 1.f / pair32_sum(l_run[x]);
    const int q = q0 + 32 * x + ql;
    if (q < Tq) {
      bf16_t* orow = p.O + ((long long)b * Tq + q) * p.ldo + h * HD;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int d0 = db * 32 + 8 * rg + 4 * hi;
          u32x2 o;
          o[0] = pack_bf16x2(oT[x][db][rg * 4 + 0] * inv, oT[x][db][rg * 4 + 1] * inv);
          o[1] = pack_bf16x2(oT[x][db][rg * 4 + 2] * inv, oT[x][db][rg * 4 + 3] * inv);
          *(u32x2*)(orow + d0) = o;
        }
    }
  }
}

// kernel variant of fluxhip_attention_d128_bf16: 0 auto; 2 / 3: attn_kernel with one / two wave sets; 4 / 5 / 6: the 64-queries-
// per-wave kernel with 256 / 128 queries per workgroup / chosen by grid size; + 0x100: its online-rescale fallback path.
// FLUXHIP_ATTN in the environment or fluxhip_attention_set_variant() (tools/attn_bench.py, tests).
int g_attn_variant = [] { const char* e = getenv("FLUXHIP_ATTN"); return e ? (int)strtol(e, nullptr, 0) : 0; }();

template <int KS>
int launch_attn128_w64(AttnParams p, int B, hipStream_t s) {
  constexpr int lds = 2 * KS * (KV * 128 * 2 + 128 * KV * 2) + 16;      // + the overflow flag word
  auto fn = attn128_w64_kernel<KS>;
  p.force_slow = (g_attn_variant & 0x100) != 0;       // tests: force the online-rescale fallback
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return FLUXHIP_ELAUNCH;
    done = true;
  }
  p.nqb = (p.Tq + (256 / KS) - 1) / (256 / KS);
  hipLaunchKernelGGL(fn, dim3(B * p.H * p.nqb), dim3(256), lds, s, p);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

template <int HD, int MODE, int KS = 1, int VP = 0, bool F16 = false, bool MXO = false>
int launch_attn(const AttnParams& p, int B, hipStream_t s) {
  constexpr int NW = 4;
  constexpr int lds = 2 * KS * (KV * HD * 2 + HD * KV * 2);
  auto fn = attn_kernel<HD, NW, MODE, KS, VP, F16, MXO>;
  // the staging addresses are 32-bit lane offsets from a per-head base (see attn_kernel): one head's K / V^T image < 4 GiB
  if ((long long)p.Tk * p.k_rs * 2 >= (1ll << 32) || (long long)HD * p.Tkpad * 2 >= (1ll << 32)) return FLUXHIP_EINVAL;
  static bool done = false;            // one flag per template instantiation
  if (!done) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return FLUXHIP_ELAUNCH;
    done = true;
  }
  hipLaunchKernelGGL(fn, dim3(B * p.H * p.nqb), dim3(NW * KS * 64), lds, s, p);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}

}  // namespace

extern "C" int fluxhip_attention_d128_bf16(const void* Q, const void* K, const void* Vt, void* O,
                                           int ldo, int B, int H, int T, int Tpad, float scale,
                                           void* stream) {
  if (!Q || !K || !Vt || !O || B < 1 || H < 1 || T < 1 || Tpad % 64 || Tpad < T || ldo % 4)
    return FLUXHIP_EINVAL;
  AttnParams p{};
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.q_rs = p.k_rs = 128;
  p.q_hs = p.k_hs = (long long)T * 128;
  p.q_bs = p.k_bs = (long long)H * T * 128;
  p.ldo = ldo; p.H = H; p.Tq = T; p.Tk = T; p.Tkpad = Tpad;
  p.vt_bs = (long long)H * 128 * Tpad;
  p.nqb = (T + 127) / 128;
  p.scale_log2 = scale * 1.4426950408889634f;
  // fewer workgroups than two per CU: split the KV tiles over a second wave set instead (see attn_kernel)
  const int variant = g_attn_variant & 0xff;
  if (variant == 4) return launch_attn128_w64<1>(p, B, (hipStream_t)stream);                       // 64 queries per wave, 256 per workgroup
  if (variant == 5 && T > 2 * KV) return launch_attn128_w64<2>(p, B, (hipStream_t)stream);         // ... 128 per workgroup, two KV wave sets
  if (variant == 6) {                                                                               // ... chosen by grid size
    if ((long long)B * H * ((T + 255) / 256) < 256 && T > 2 * KV) return launch_attn128_w64<2>(p, B, (hipStream_t)stream);
    return launch_attn128_w64<1>(p, B, (hipStream_t)stream);
  }
  if (variant == 2) return launch_attn<128, 0, 1, 1>(p, B, (hipStream_t)stream);
  if (variant == 3 && T > 2 * KV) return launch_attn<128, 0, 2, 1>(p, B, (hipStream_t)stream);   // two wave sets even on large grids
  if ((long long)B * H * p.nqb < 384 && T > 2 * KV) return launch_attn<128, 0, 2, 1>(p, B, (hipStream_t)stream);
  // (the 64-queries-per-wave kernel, variants 4 - 6, stays opt-in: interleaved A/B runs of tools/attn_bench.py put it within
  //  +-4 % of attn_kernel on every Flux shape — 257 vs 263 us at T = 4352, 273 vs 283 at T = 4608, 945 vs 921 at B = 4 —
  //  both are bound by instruction issue, ~13 non-MFMA instructions per MFMA in its S phases)
  return launch_attn<128, 0, 1, 1>(p, B, (hipStream_t)stream);
}

// fluxhip_attention_d128_bf16 with the fp8 quantisation of the NEXT Linear's operand fused into the finalize step: out8 is
// e4m3 [B * T][ld8] (head h at columns [128 h, 128 h + 128)), mx the tiled E8M0 block scales (row = b * T + t).
extern "C" int fluxhip_attention_d128_mx(const void* Q, const void* K, const void* Vt, void* out8, int ld8, void* mx,
                                         int64_t mx_kstride, int B, int H, int T, int Tpad, float scale, void* stream) {
  if (!Q || !K || !Vt || !out8 || !mx || B < 1 || H < 1 || T < 1 || Tpad % 64 || Tpad < T || ld8 % 4 || ld8 < H * 128 ||
      mx_kstride % 64 || mx_kstride < (int64_t)B * T || ((uintptr_t)out8 & 3))
    return FLUXHIP_EINVAL;
  AttnParams p{};
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt;
  p.O8 = (uint8_t*)out8; p.mx = (uint8_t*)mx; p.mx_kstride = mx_kstride;
  p.q_rs = p.k_rs = 128;
  p.q_hs = p.k_hs = (long long)T * 128;
  p.q_bs = p.k_bs = (long long)H * T * 128;
  p.ldo = ld8; p.H = H; p.Tq = T; p.Tk = T; p.Tkpad = Tpad;
  p.vt_bs = (long long)H * 128 * Tpad;
  p.nqb = (T + 127) / 128;
  p.scale_log2 = scale * 1.4426950408889634f;
  if ((long long)B * H * p.nqb < 384 && T > 2 * KV) return launch_attn<128, 0, 2, 1, false, true>(p, B, (hipStream_t)stream);
  return launch_attn<128, 0, 1, 1, false, true>(p, B, (hipStream_t)stream);
}

extern "C" int fluxhip_attention_set_variant(int v) {
  g_attn_variant = v;
  return FLUXHIP_OK;
}

template <bool H>
static int attention_strided_vt(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                const void* Vt, int64_t vt_bs, void* O, int ldo, int B, int Hh,
                                int head_dim, int Tq, int Tk, int Tkpad, float scale, void* stream) {
  if (!Q || !K || !Vt || !O || B < 1 || Hh < 1 || Tq < 1 || Tk < 1 || Tkpad % 64 || Tkpad < Tk)
    return FLUXHIP_EINVAL;
  if ((head_dim != 64 && head_dim != 128) || ldo % 4 || q_rs % 8 || k_rs % 8 || q_hs % 8 || k_hs % 8 ||
      q_bs % 8 || k_bs % 8 || vt_bs % 8 || vt_bs < (int64_t)Hh * head_dim * Tkpad)
    return FLUXHIP_EINVAL;
  if (H && head_dim != 64) return FLUXHIP_EINVAL;          // float16: the UNet / CLIP head size
  AttnParams p{};
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
  p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
  p.vt_bs = vt_bs;
  p.ldo = ldo; p.H = Hh; p.Tq = Tq; p.Tk = Tk; p.Tkpad = Tkpad;
  p.nqb = (Tq + 127) / 128;
  p.scale_log2 = scale * 1.4426950408889634f;
  if constexpr (H) return launch_attn<64, 0, 1, 0, true>(p, B, (hipStream_t)stream);
  else return head_dim == 128 ? launch_attn<128, 0>(p, B, (hipStream_t)stream)
                              : launch_attn<64, 0>(p, B, (hipStream_t)stream);
}

extern "C" int fluxhip_attention_strided_vt_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                                 const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                                 const void* Vt, int64_t vt_bs, void* O, int ldo, int B, int H,
                                                 int head_dim, int Tq, int Tk, int Tkpad, float scale,
                                                 void* stream) {
  return attention_strided_vt<false>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt, vt_bs, O, ldo, B, H, head_dim, Tq, Tk,
                                     Tkpad, scale, stream);
}
// float16 storage, head_dim 64 (the UNet's nn.MultiHeadAttention with float16=True)
extern "C" int fluxhip_attention_strided_vt_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                                const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                                const void* Vt, int64_t vt_bs, void* O, int ldo, int B, int H,
                                                int head_dim, int Tq, int Tk, int Tkpad, float scale,
                                                void* stream) {
  return attention_strided_vt<true>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt, vt_bs, O, ldo, B, H, head_dim, Tq, Tk,
                                    Tkpad, scale, stream);
}

extern "C" int fluxhip_attention_strided_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                              const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                              const void* Vt, void* O, int ldo, int B, int H,
                                              int head_dim, int Tq, int Tk, int Tkpad, float scale,
                                              void* stream) {
  return attention_strided_vt<false>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt,
                                     (int64_t)H * head_dim * Tkpad, O, ldo, B, H, head_dim, Tq, Tk, Tkpad,
                                     scale, stream);
}
extern "C" int fluxhip_attention_strided_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                             const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                             const void* Vt, void* O, int ldo, int B, int H,
                                             int head_dim, int Tq, int Tk, int Tkpad, float scale,
                                             void* stream) {
  return attention_strided_vt<true>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt,
                                    (int64_t)H * head_dim * Tkpad, O, ldo, B, H, head_dim, Tq, Tk, Tkpad,
                                    scale, stream);
}

template <bool F16>
static int attention_masked(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                            const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                            const void* Vt, void* O, int ldo, int B, int H, int Tq,
                            int Tk, int Tkpad, float scale, const void* bias,
                            int causal, void* stream) {
  if (!Q || !K || !Vt || !O || B < 1 || H < 1 || Tq < 1 || Tk < 1 || Tkpad % 64 || Tkpad < Tk)
    return FLUXHIP_EINVAL;
  if (ldo % 4 || q_rs % 8 || k_rs % 8 || q_hs % 8 || k_hs % 8 || q_bs % 8 || k_bs % 8) return FLUXHIP_EINVAL;
  if ((bias != nullptr) == (causal != 0)) return FLUXHIP_EINVAL;   // exactly one of the two
  if (bias && (Tk % 4)) return FLUXHIP_EINVAL;
  AttnParams p{};
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
  p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
  p.ldo = ldo; p.H = H; p.Tq = Tq; p.Tk = Tk; p.Tkpad = Tkpad;
  p.vt_bs = (long long)H * 64 * Tkpad;
  p.nqb = (Tq + 127) / 128;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.scale = scale;
  p.bias = (const bf16_t*)bias;
  if constexpr (F16) return bias ? FLUXHIP_EINVAL : launch_attn<64, 2, 1, 0, true>(p, B, (hipStream_t)stream);   // (T5's bias mode: bf16 only)
  else return bias ? launch_attn<64, 1>(p, B, (hipStream_t)stream) : launch_attn<64, 2>(p, B, (hipStream_t)stream);
}

extern "C" int fluxhip_attention_masked_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                             const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                             const void* Vt, void* O, int ldo, int B, int H, int Tq,
                                             int Tk, int Tkpad, float scale, const void* bias,
                                             int causal, void* stream) {
  return attention_masked<false>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt, O, ldo, B, H, Tq, Tk, Tkpad, scale, bias, causal, stream);
}
// float16 storage, causal only (the CLIP text towers of the stable_diffusion/ pipelines with float16=True)
extern "C" int fluxhip_attention_masked_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs,
                                            const void* K, int64_t k_bs, int64_t k_hs, int64_t k_rs,
                                            const void* Vt, void* O, int ldo, int B, int H, int Tq,
                                            int Tk, int Tkpad, float scale, const void* bias,
                                            int causal, void* stream) {
  return attention_masked<true>(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, Vt, O, ldo, B, H, Tq, Tk, Tkpad, scale, bias, causal, stream);
}
