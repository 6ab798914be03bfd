// Joint (txt+img) non-causal flash attention, head_dim 128, bf16 in / fp32 softmax / bf16 out.
// Replaces mx.fast.scaled_dot_product_attention(q, k, v, scale=D**-0.5) and the
// transpose/reshape after it (reference flux/layers.py:36-43; joint [txt;img] token order from
// flux/layers.py:212-214).
//
// MI355X design (64-lane waves, v_mfma_f32_32x32x16_bf16):
//  * one wave owns 32 query rows; NW waves per workgroup share the K / V^T tiles in LDS.
//  * both products are computed TRANSPOSED so that every per-query quantity is lane-local:
//        S^T[key][q] = K (A, from LDS) x Q^T (B, registers)        -> lane holds 16 keys of query (lane&31)
//        O^T[d][q]   = V^T (A, from LDS) x P^T (B, registers)       -> lane holds 64 d's of query (lane&31)
//    the S^T accumulator layout *is* the B-operand layout of the second product (up to a fixed key
//    permutation that is applied to the V^T fragment address instead), so P never leaves registers
//    and the online-softmax rescale of O needs no cross-lane traffic; row max / row sum need one
//    exchange with lane^32.
//  * K tile [64 keys][128 d] and V^T tile [128 d][64 keys] arrive by LDS-DMA
//    (global_load_lds_dwordx4), double-buffered, one barrier per KV tile. The LDS image is
//    lane-linear, so the bank swizzle is applied to the per-lane *source* address and again on
//    the fragment reads (K: 16-B chunk ^= row&15 -> conflict-free ds_read_b128;
//    V^T: chunk ^= (row>>1)&7).
//  * V^T ([B][H][128][Tpad], zero padded) is produced by fluxhip_qk_norm_rope_bf16.
#include "../../include/fluxhip.h"
#include "common.h"

namespace {

constexpr int KV = 64;                 // keys per tile
constexpr int KT_BYTES = KV * 256;     // K tile
constexpr int VT_BYTES = 128 * KV * 2; // V^T tile
constexpr int STAGE = KT_BYTES + VT_BYTES;

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_d128_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
    bf16_t* __restrict__ O, int ldo, int H, int T, int Tpad, float scale_log2, int nqb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;

  // XCD-aware map: all query blocks of one (b,h) land on the same XCD (they share K/V in its L2)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = logical / nqb;
  const int qb = logical - bh * nqb;
  const int b = bh / H, h = bh - b * H;

  const bf16_t* Qh = Q + (long long)bh * T * 128;
  const bf16_t* Kh = K + (long long)bh * T * 128;
  const bf16_t* Vh = Vt + (long long)bh * 128 * Tpad;

  const int q0 = qb * (NW * 32) + wave * 32;
  const int qrow = min(q0 + ql, T - 1);

  // Q^T fragments (B operand): lane (q = lane&31, hi) holds d = ds*16 + hi*8 .. +8
  bf16x8 qf[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds)
    qf[ds] = *(const bf16x8*)(Qh + (long long)qrow * 128 + ds * 16 + hi * 8);

  // staging sources
  constexpr int KPW = 16 / NW;  // 1-KiB pieces per wave per tile (K: 16 pieces, V^T: 16 pieces)
  const int kr = lane >> 4, kc = lane & 15;   // K piece: 4 rows x 16 chunks
  const int vr = lane >> 3, vc = lane & 7;    // V^T piece: 8 rows x 8 chunks
  const char* vsrc[KPW];
  int krow[KPW], kchunk[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    int piece = wave + i * NW;
    int row = piece * 4 + kr;
    krow[i] = row;
    kchunk[i] = kc ^ (row & 15);
    int d = piece * 8 + vr;
    int lchunk = vc ^ ((d >> 1) & 7);
    vsrc[i] = (const char*)(Vh + (long long)d * Tpad) + lchunk * 16;
  }
  auto stage = [&](int it, int buf) {
    char* sk = smem + buf * STAGE;
    char* sv = sk + KT_BYTES;
    const int key0 = it * KV;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      int key = min(key0 + krow[i], T - 1);
      glds16((const char*)(Kh + (long long)key * 128) + kchunk[i] * 16, sk + (wave + i * NW) * 1024);
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) glds16(vsrc[i] + (long long)key0 * 2, sv + (wave + i * NW) * 1024);
  };

  f32x16 oT[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // fragment read offsets
  // K (A operand): row = kb*32 + ql, logical chunk = ds*2 + hi, phys = chunk ^ (row & 15)
  const int k_rowoff = ql * 256;
  const int k_sw = ql & 15;
  // V^T (A operand): row d = dblk*32 + ql ; bytes: ((kb*4 + j*2 + {0,1}) ^ ((d>>1)&7))*16 + hi*8
  const int v_rowoff = ql * 128 + hi * 8;
  const int v_sw = (ql >> 1) & 7;   // (dblk*32 + ql) >> 1 & 7 == (ql>>1)&7 since 32/2 = 16 = 0 mod 8

  const int ntiles = (T + KV - 1) / KV;
  stage(0, 0);
  wait_vm0();
  __syncthreads();

  for (int it = 0; it < ntiles; ++it) {
    const int cur = it & 1;
    if (it + 1 < ntiles) stage(it + 1, cur ^ 1);
    const char* sk = smem + cur * STAGE;
    const char* sv = sk + KT_BYTES;

    // ---- S^T = K Q^T -------------------------------------------------------
    f32x16 sT[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[kb][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        bf16x8 kf = *(const bf16x8*)(sk + kb * 32 * 256 + k_rowoff + (((ds * 2 + hi) ^ k_sw) << 4));
        sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sT[kb], 0, 0, 0);
      }
    }
    // ---- mask the key tail (last tile only) ----------------------------------
    const int key0 = it * KV;
    if (key0 + KV > T) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= T) sT[kb][r] = -1e30f;
        }
    }
    // ---- online softmax (per query = per lane pair (l, l^32)) -----------------
    float mx = sT[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (!__all(m_new == m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
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
        float p = __builtin_amdgcn_exp2f(fmaf(sT[kb][r], scale_log2, mneg));
        sT[kb][r] = p;
        psum += p;
      }
    l_run += psum;

    // ---- O^T += V^T P^T -------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // B operand: k-slot e <-> P reg 8j+e of key block kb
        union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pf.u[e] = pack_bf16x2(sT[kb][8 * j + 2 * e], sT[kb][8 * j + 2 * e + 1]);
        const int c1 = kb * 4 + j * 2;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const char* base = sv + db * 32 * 128 + v_rowoff;
          union { bf16x8 v; u32x2 h[2]; } vf;
          vf.h[0] = *(const u32x2*)(base + (((c1) ^ v_sw) << 4));
          vf.h[1] = *(const u32x2*)(base + (((c1 + 1) ^ v_sw) << 4));
          oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oT[db], 0, 0, 0);
        }
      }
    }
    wait_vm0();
    __syncthreads();
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l -----------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int q = q0 + ql;
  if (q < T) {
    bf16_t* orow = O + ((long long)b * T + q) * ldo + h * 128;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = db * 32 + 8 * rg + 4 * hi;
        u32x2 o;
        o[0] = pack_bf16x2(oT[db][rg * 4 + 0] * inv, oT[db][rg * 4 + 1] * inv);
        o[1] = pack_bf16x2(oT[db][rg * 4 + 2] * inv, oT[db][rg * 4 + 3] * inv);
        *(u32x2*)(orow + d0) = o;
      }
  }
}

bool g_attr_done = false;

}  // namespace

extern "C" int fluxhip_attention_d128_bf16(const void* Q, const void* K, const void* Vt, void* O,
                                           int ldo, int B, int H, int T, int Tpad, float scale,
                                           void* stream) {
  if (!Q || !K || !Vt || !O || B < 1 || H < 1 || T < 1 || Tpad % 64 || Tpad < T || ldo % 4)
    return FLUXHIP_EINVAL;
  constexpr int NW = 4;
  const int nqb = (T + NW * 32 - 1) / (NW * 32);
  const int lds = 2 * STAGE;
  auto fn = attn_d128_kernel<NW>;
  if (!g_attr_done) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return FLUXHIP_ELAUNCH;
    g_attr_done = true;
  }
  const float scale_log2 = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(fn, dim3(B * H * nqb), dim3(NW * 64), lds, (hipStream_t)stream,
                     (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, T,
                     Tpad, scale_log2, nqb);
  return hipGetLastError() == hipSuccess ? FLUXHIP_OK : FLUXHIP_ELAUNCH;
}
