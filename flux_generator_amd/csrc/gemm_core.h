// bf16 "NT" GEMM main loop for gfx950:  C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue)
//
// This is the contraction behind every Linear on the Flux denoise path
// (reference: flux/layers.py:104-106,163-165,250-252 and flux/model.py:56,64) and,
// with the implicit-GEMM A loader, behind the VAE / UNet 3x3 convolutions
// (reference: flux/autoencoder.py:70-81,117-122).
//
// Design (MI355X-first, not a translation of anything):
//  * 64-lane waves, v_mfma_f32_16x16x32_bf16, fp32 accumulators.
//  * Both operands are K-contiguous, so A and W tiles are staged with
//    global_load_lds_dwordx4 (16 B / lane, 1 KiB / wave-instruction) straight
//    into LDS, double-buffered, one barrier per K-step of 64.
//  * LDS image is lane-linear (hardware constraint of the LDS-DMA), so the bank
//    swizzle lives on the *source* address: 16-B chunk c of tile row r is fetched
//    into physical chunk c ^ (r & 7); fragment reads apply the same XOR. With
//    128-B rows this makes every ds_read_b128 lane group hit 16 distinct slots.
//  * MFMA operands are swapped (W fragment as "A", activation fragment as "B") so
//    that each lane ends up holding 4 *consecutive output columns* of one row:
//    epilogue loads/stores are 8-byte vectors, bias/gate are read as 4 bf16.
//  * Workgroup -> tile mapping is XCD-aware: the 8 XCDs (private L2 each) get
//    contiguous ranges of N-tiles with all M-tiles of that range, so every weight
//    panel is pulled from HBM by exactly one XCD.
#pragma once
#include "common.h"

enum GemmEpi : int {
  EPI_BIAS = 0,       // C = acc + bias
  EPI_GELU_TANH = 1,  // C = gelu_tanh(acc + bias)
  EPI_GATE_RES = 2,   // C = res + gate[b,n] * (acc + bias)      (gate may be null -> 1)
  EPI_SPLIT_GELU = 3, // n <  n_split: C  = acc + bias
                      // n >= n_split: C2[:, n - n_split + c2_coloff] = gelu_tanh(acc + bias)
  EPI_SILU = 4,       // C = silu(acc + bias)
  EPI_GEGLU = 5,      // C = res * gelu_erf(acc + bias)       (UNet GEGLU: linear1(y) * gelu(linear2(y)))
  EPI_QUICK_GELU = 6, // C = v * sigmoid(1.702 v), v = acc + bias   (CLIP "quick_gelu", flux/clip.py:9)
};

struct GemmGroup {
  const bf16_t* A;     // [nbatch][M][lda]   (dense A loader)
  const bf16_t* W;     // [N][K]
  const bf16_t* bias;  // [N] or null  (row_bias: [M])
  bf16_t* C;           // [nbatch][M][ldc]
  const bf16_t* res;   // residual, same indexing as C (EPI_GATE_RES)
  const bf16_t* gate;  // [nbatch][gate_bstride] or null
  long long a_bstride; // elements between batches of A
  long long c_bstride; // elements between batches of C / res
  long long gate_bstride;
  long long w_bstride; // elements between batches of W (0: shared weight)
  int M;               // rows per batch
  int tiles_m;         // ceil(M / BM)
};

struct ConvGeom {        // implicit-GEMM A loader (NHWC activations)
  const bf16_t* X;       // [B][Hs][Ws][Cin]
  const bf16_t* zero;    // >= 16 zero bytes (border taps read this)
  int Hs, Ws;            // stored (source) resolution
  int Ho, Wo;            // output resolution
  int Cin;               // multiple of 64
  int ksize;             // 1 or 3
  int stride;            // 1 or 2
  int pad;               // 0 or 1
  int ups;               // 1: input is nearest-upsampled x2 on the fly
};

struct GemmParams {
  GemmGroup g[2];
  ConvGeom cv;
  int ngroups, nbatch;
  int N, K;
  int lda, ldc;
  int epi;
  int row_bias;          // bias indexed by output row instead of column
  int n_split;
  bf16_t* C2;
  int ldc2;
  long long c2_bstride;
  int c2_coloff;
  int tiles_m_total, tiles_n;
  float alpha;           // scales acc before bias (attention logits etc.)
  int out_f32;           // EPI_BIAS only: C is float32 (logits of the single-head VAE attention)
  const bf16_t* addvec;  // optional [.., addvec_stride] vector added per group of addvec_rows output rows
  int addvec_rows;
  long long addvec_stride;
};

template <int N>
DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NSTAGE-deep LDS ring: the loads of K-step kt+NSTAGE-1 are issued right after the barrier that
// opens step kt, so they have NSTAGE-1 compute phases to land; waits are COUNTED (vmcnt(N), never a
// drain in steady state) and the barrier is a raw s_barrier, because __syncthreads() would drain
// the in-flight LDS-DMA (cdna guide §5, "Pipelining across barriers").
template <int BM, int BN, int WM, int WN, int AMODE, int NSTAGE, int PIPE>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(const GemmParams p) {
  constexpr int BK = 64;
  constexpr int NWAVES = WM * WN;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  constexpr int APW = BM / 8 / NWAVES;
  // B pieces need not divide evenly over the waves (BN = 192, 224): the surplus slots re-fetch the
  // last piece (same bytes to the same LDS address), which keeps the per-wave vmcnt arithmetic uniform.
  constexpr int BPIECES = BN / 8, BPW = (BPIECES + NWAVES - 1) / NWAVES;
  static_assert(APW >= 1 && BM % (8 * NWAVES) == 0 && BN % 16 == 0, "tile / wave-count mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware, bijective block -> tile map -------------------------------
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int TM = p.tiles_m_total;
  int tm = swz % TM;
  const int tn = swz / TM;

  const int tm0 = p.nbatch * p.g[0].tiles_m;
  const bool g1 = tm >= tm0;
  if (g1) tm -= tm0;
  const bf16_t* gA = g1 ? p.g[1].A : p.g[0].A;
  const bf16_t* gW = g1 ? p.g[1].W : p.g[0].W;
  const bf16_t* gBias = g1 ? p.g[1].bias : p.g[0].bias;
  bf16_t* gC = g1 ? p.g[1].C : p.g[0].C;
  const bf16_t* gRes = g1 ? p.g[1].res : p.g[0].res;
  const bf16_t* gGate = g1 ? p.g[1].gate : p.g[0].gate;
  const long long a_bs = g1 ? p.g[1].a_bstride : p.g[0].a_bstride;
  const long long c_bs = g1 ? p.g[1].c_bstride : p.g[0].c_bstride;
  const long long gate_bs = g1 ? p.g[1].gate_bstride : p.g[0].gate_bstride;
  const long long w_bs = g1 ? p.g[1].w_bstride : p.g[0].w_bstride;
  const int Mg = g1 ? p.g[1].M : p.g[0].M;
  const int tpb = g1 ? p.g[1].tiles_m : p.g[0].tiles_m;
  const int b = tm / tpb;
  const int m0 = (tm % tpb) * BM;
  const int n0 = tn * BN;
  const int N = p.N, K = p.K;

  // ---- per-lane staging sources ----------------------------------------------
  const int lr = lane >> 3;             // row inside an 8-row piece
  const int lc = (lane & 7) ^ lr;       // logical 16-B chunk fetched by this lane
  const char* asrc[APW];
  int arow[APW];                        // conv: packed output-pixel coords
  const char* bsrc[BPW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    int row = (wave + i * NWAVES) * 8 + lr;
    int grow = min(m0 + row, Mg - 1);
    if (AMODE == 0) {
      asrc[i] = (const char*)(gA + (long long)b * a_bs + (long long)grow * p.lda) + lc * 16;
      arow[i] = 0;
    } else {
      asrc[i] = nullptr;
      arow[i] = grow;
    }
  }
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    int row = min(wave + i * NWAVES, BPIECES - 1) * 8 + lr;
    int n = min(n0 + row, N - 1);
    bsrc[i] = (const char*)(gW + (long long)b * w_bs + (long long)n * K) + lc * 16;
  }

  // conv geometry decode (per staged row), hoisted out of the K loop
  int cy[APW], cx[APW];
  const char* cbase[APW];
  if (AMODE == 1) {
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      int pix = arow[i];
      int hw = p.cv.Ho * p.cv.Wo;
      int bb = pix / hw;
      int rem = pix - bb * hw;
      int y = rem / p.cv.Wo;
      int x = rem - y * p.cv.Wo;
      cy[i] = y * p.cv.stride - p.cv.pad;
      cx[i] = x * p.cv.stride - p.cv.pad;
      cbase[i] = (const char*)(p.cv.X + (long long)bb * p.cv.Hs * p.cv.Ws * p.cv.Cin) + lc * 16;
    }
  }

  // LDS layout: NSA activation slots of A_BYTES, then NSW weight slots of B_BYTES.  The weight ring may be
  // one slot deeper than the activation ring (PIPE 3): weights are the cold HBM stream (every line is a
  // compulsory miss for the XCD), activations are re-read by every N-tile and mostly hit L2, and 2 x 32 KiB
  // + 3 x 32 KiB is exactly the 160 KiB of a CU for the 256 x 256 tile.
  constexpr int NSA = NSTAGE, NSW = (PIPE == 3) ? NSTAGE + 1 : NSTAGE;
  constexpr int W_BASE = NSA * A_BYTES;
  auto stage_a = [&](int kt, int slot) {
    char* sa = smem + slot * A_BYTES;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < APW; ++i)
        glds16(asrc[i] + (long long)kt * (BK * 2), sa + (wave + i * NWAVES) * 1024);
    } else {
      // K order = channel chunk outer, filter tap inner: the 9 taps of one 64-channel chunk are
      // staged back to back, so their overlapping input rows are still in L1/L2 (a tap-major order
      // re-streams the whole input tile 9 times through the XCD's L2).
      const int ntap = p.cv.ksize * p.cv.ksize;
      const int cch = kt / ntap;
      const int tap = kt - cch * ntap;
      const int c0 = cch << 6;
      const int dy = (p.cv.ksize == 3) ? tap / 3 : 0;
      const int dx = (p.cv.ksize == 3) ? tap - dy * 3 : 0;
      const int Hl = p.cv.ups ? p.cv.Hs * 2 : p.cv.Hs;   // logical input grid
      const int Wl = p.cv.ups ? p.cv.Ws * 2 : p.cv.Ws;
#pragma unroll
      for (int i = 0; i < APW; ++i) {
        int yy = cy[i] + dy, xx = cx[i] + dx;
        bool ok = (yy >= 0) & (yy < Hl) & (xx >= 0) & (xx < Wl);
        int ys = p.cv.ups ? (yy >> 1) : yy;
        int xs = p.cv.ups ? (xx >> 1) : xx;
        const char* src = ok ? cbase[i] + ((long long)(ys * p.cv.Ws + xs) * p.cv.Cin + c0) * 2
                             : (const char*)p.cv.zero;
        glds16(src, sa + (wave + i * NWAVES) * 1024);
      }
    }
  };
  auto stage_w = [&](int kt, int slot) {
    char* sb = smem + W_BASE + slot * B_BYTES;
    int koff = kt * BK;                      // K offset (elements) of this step inside a W row
    if (AMODE == 1) {                        // conv: weight column = tap*Cin + c0 (pure index remap)
      const int ntap = p.cv.ksize * p.cv.ksize;
      const int cch = kt / ntap;
      koff = (kt - cch * ntap) * p.cv.Cin + (cch << 6);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i)
      glds16(bsrc[i] + (long long)koff * 2, sb + min(wave + i * NWAVES, BPIECES - 1) * 1024);
  };

  // ---- fragment read offsets (same XOR as the staging source swizzle) ---------
  const int r16 = lane & 15, q4 = lane >> 4;
  int foff[2];
  foff[0] = r16 * 128 + (((0 + q4) ^ (lane & 7)) << 4);
  foff[1] = r16 * 128 + (((4 + q4) ^ (lane & 7)) << 4);

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  constexpr int G = APW + BPW;                       // LDS-DMA instructions per wave per K-step
  static_assert((NSTAGE - 1) * G + BPW <= 63, "vmcnt field");

  auto mma = [&](const bf16x8(&af)[MI], const bf16x8(&wf)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };

  if constexpr (PIPE == 0) {
    // simple ring: wait -> barrier -> refill the freed slot -> read fragments -> MFMA
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nkt) { stage_a(s, s); stage_w(s, s); }
    int cur = 0, nxt = NSTAGE - 1;                   // ring slots of step kt and step kt+NSTAGE-1
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + NSTAGE - 2 < nkt) wait_vmcnt<(NSTAGE - 2) * G>();
      else wait_vmcnt<0>();                          // tail: fewer steps in flight than the ring holds
      __builtin_amdgcn_s_barrier();                  // step kt landed for every wave; slot nxt is free
      if (kt + NSTAGE - 1 < nkt) { stage_a(kt + NSTAGE - 1, nxt); stage_w(kt + NSTAGE - 1, nxt); }
      const char* sa = smem + cur * A_BYTES + (wm * WTM) * 128;
      const char* sb = smem + W_BASE + cur * B_BYTES + (wn * WTN) * 128;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + i * 2048 + foff[kk]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *(const bf16x8*)(sb + j * 2048 + foff[kk]);
        mma(af, wf);
      }
      cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
      nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    }
  } else {
    // fragment-double-buffered ring: every MFMA cluster runs with the NEXT cluster's ds_reads in
    // flight; the barrier (and the wait for the next K-step's LDS-DMA) sits between the two
    // clusters of a step, so LDS latency and barrier skew hide under 32..64 MFMAs.
    // The fragment reads are inline asm with hand-counted lgkmcnt: hipcc's own bookkeeping drains
    // lgkmcnt(0) at the loop head, which serialises [reads -> wait -> MFMAs] inside each wave.
    // lgkmcnt is a 4-bit field: with 16 fragment reads per cluster the wait below asks for <= 15
    // outstanding, i.e. it also waits for the first read of the NEXT cluster (reads retire in order).
    constexpr int NF = (MI + NJ) < 15 ? (MI + NJ) : 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_rd = lds0 + (wm * WTM) * 128;             // + slotA*A_BYTES + foff[kk]
    const uint32_t b_rd = lds0 + W_BASE + (wn * WTN) * 128;    // + slotW*B_BYTES + foff[kk]
    auto read_frags = [&](bf16x8(&af)[MI], bf16x8(&wf)[NJ], int slot_a, int slot_w, int kk) {
      const uint32_t aa = a_rd + slot_a * A_BYTES + foff[kk];
      const uint32_t bb = b_rd + slot_w * B_BYTES + foff[kk];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(aa), "n"(i * 2048) : "memory");
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j]) : "v"(bb), "n"(j * 2048) : "memory");
    };
    // LDS-DMA issue order (it fixes the vmcnt arithmetic): prologue A0 W0 A1 W1 .. then the extra weight
    // steps W(NSA)..W(NSW-1); iteration j issues A(j+NSA) then W(j+NSW).  When step kt+1 is needed, the
    // loads issued after A(kt+1) are W(kt+1-NSA+NSW) plus NSA-2 whole iterations:
    constexpr int PENDING = BPW * (NSW - NSA) + (NSA - 2) * G;   // == (NSTAGE-2)*G for a uniform ring
#pragma unroll
    for (int s = 0; s < NSA; ++s)
      if (s < nkt) { stage_a(s, s); stage_w(s, s); }
#pragma unroll
    for (int s = NSA; s < NSW; ++s)
      if (s < nkt) stage_w(s, s);
    if (nkt >= NSW) wait_vmcnt<(NSA - 1) * G + (NSW - NSA) * BPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[MI], w0[NJ], a1[MI], w1[NJ];
    read_frags(a0, w0, 0, 0, 0);
    int ca = 0, cw = 0;                                // ring slots of step kt
    for (int kt = 0; kt < nkt; ++kt) {
      const int na = (ca + 1 == NSA) ? 0 : ca + 1;
      const int nw = (cw + 1 == NSW) ? 0 : cw + 1;
      read_frags(a1, w1, ca, cw, 1);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF) : "memory");      // a0/w0 landed, a1/w1 in flight
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, w0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkt) {
        if (kt + NSW - 1 < nkt) wait_vmcnt<PENDING>();                // step kt+1 landed (my pieces)
        else wait_vmcnt<0>();                                          // tail: fewer steps in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my reads of the current slots are done
        __builtin_amdgcn_s_barrier();                                   // ... and everybody else's
        if (kt + NSA < nkt) stage_a(kt + NSA, ca);                      // refill the slots just drained
        if (kt + NSW < nkt) stage_w(kt + NSW, cw);
        read_frags(a0, w0, na, nw, 0);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, w1);
      __builtin_amdgcn_sched_barrier(0);
      ca = na;
      cw = nw;
    }
  }

  // ---- epilogue: lane holds C[m][n4 .. n4+3] -----------------------------------
  const int epi = p.epi;
  const float alpha = p.alpha;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WTM + i * 16 + r16;
    if (m >= Mg) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n4 = n0 + wn * WTN + j * 16 + q4 * 4;
      if (n4 >= N) continue;
      float v[4];
      if (gBias) {
        if (p.row_bias) {
          float bv = bf2f(gBias[m]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha + bv;
        } else {
          u32x2 bw = *(const u32x2*)(gBias + n4);
          v[0] = acc[i][j][0] * alpha + bf_lo(bw[0]);
          v[1] = acc[i][j][1] * alpha + bf_hi(bw[0]);
          v[2] = acc[i][j][2] * alpha + bf_lo(bw[1]);
          v[3] = acc[i][j][3] * alpha + bf_hi(bw[1]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha;
      }
      if (p.addvec) {   // per-image vector added after the bias (ResnetBlock2D: + temb[:, None, None, :])
        u32x2 aw = *(const u32x2*)(p.addvec + (long long)(m / p.addvec_rows) * p.addvec_stride + n4);
        v[0] = rbf(v[0]) + bf_lo(aw[0]);
        v[1] = rbf(v[1]) + bf_hi(aw[0]);
        v[2] = rbf(v[2]) + bf_lo(aw[1]);
        v[3] = rbf(v[3]) + bf_hi(aw[1]);
      }
      if (p.out_f32) {
        float* fdst = (float*)gC + (long long)b * c_bs + (long long)m * p.ldc + n4;
        *(f32x4*)fdst = f32x4{v[0], v[1], v[2], v[3]};
        continue;
      }
      bf16_t* dst = gC + (long long)b * c_bs + (long long)m * p.ldc + n4;
      if (epi == EPI_GELU_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(rbf(v[r]));
      } else if (epi == EPI_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(rbf(v[r]));
      } else if (epi == EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = rbf(v[r]);
          v[r] = t / (1.0f + __expf(-1.702f * t));
        }
      } else if (epi == EPI_GATE_RES) {
        const bf16_t* rp = gRes + (long long)b * c_bs + (long long)m * p.ldc + n4;
        u32x2 rw = *(const u32x2*)rp;
        float rr[4] = {bf_lo(rw[0]), bf_hi(rw[0]), bf_lo(rw[1]), bf_hi(rw[1])};
        if (gGate) {
          u32x2 gw = *(const u32x2*)(gGate + (long long)b * gate_bs + n4);
          float gg[4] = {bf_lo(gw[0]), bf_hi(gw[0]), bf_lo(gw[1]), bf_hi(gw[1])};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rr[r] + rbf(gg[r] * rbf(v[r]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rr[r] + rbf(v[r]);
        }
      } else if (epi == EPI_GEGLU) {   // out = res * gelu_erf(acc + bias)   (y_a * nn.gelu(y_b))
        const bf16_t* rp = gRes + (long long)b * c_bs + (long long)m * p.ldc + n4;
        u32x2 rw = *(const u32x2*)rp;
        v[0] = bf_lo(rw[0]) * rbf(gelu_erf_f(rbf(v[0])));
        v[1] = bf_hi(rw[0]) * rbf(gelu_erf_f(rbf(v[1])));
        v[2] = bf_lo(rw[1]) * rbf(gelu_erf_f(rbf(v[2])));
        v[3] = bf_hi(rw[1]) * rbf(gelu_erf_f(rbf(v[3])));
      } else if (epi == EPI_SPLIT_GELU) {
        if (n4 >= p.n_split) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(rbf(v[r]));
          dst = p.C2 + (long long)b * p.c2_bstride + (long long)m * p.ldc2 +
                (n4 - p.n_split + p.c2_coloff);
        }
      }
      u32x2 o;
      o[0] = pack_bf16x2(v[0], v[1]);
      o[1] = pack_bf16x2(v[2], v[3]);
      *(u32x2*)dst = o;
    }
  }
}
