// Probe of the per-lane E8M0 block scales of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (mfma_mx_probe2.hip found the mapping):
// the instruction's K order is  lane group kb = l >> 4 supplies k in [16 kb, +16) and [64 + 16 kb, +16), and the scale byte of lane
// (r, kb), selected by op_sel, scales k in [32 kb, 32 kb + 32) of row r - NOT the 32 elements that lane supplies.
// hipcc --offload-arch=gfx950 -O2 mfma_mx_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -r : r;
}

template <int OPA, int OPB>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint32_t* SA, const uint32_t* SB, float* C) {
  const int l = threadIdx.x, r = l & 15, kb = l >> 4;
  i32x8 a, b;
  // the lane's 32 operand bytes = K elements [16 kb, 16 kb + 16) and [64 + 16 kb, 64 + 16 kb + 16) of its row (the two
  // 16-byte chunks the fp8 GEMM reads); its scale byte scales the CONTIGUOUS block [32 kb, 32 kb + 32) of that row
  const int* ap = (const int*)(A + r * 128 + kb * 16);
  const int* bp = (const int*)(B + r * 128 + kb * 16);
  for (int i = 0; i < 4; ++i) { a[i] = ap[i]; b[i] = bp[i]; a[4 + i] = ap[16 + i]; b[4 + i] = bp[16 + i]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPA, (int)SA[l], OPB, (int)SB[l]);
  for (int i = 0; i < 4; ++i) C[l * 4 + i] = acc[i];
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  srand(1);
  for (auto& v : A) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; v &= 0xbf; }
  for (auto& v : B) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; v &= 0xbf; }
  // per (row, K block) exponents, 4 different sets in the 4 bytes of the lane's scale register
  std::vector<uint32_t> SA(64), SB(64);
  int ea[4][16][4], eb[4][16][4];
  for (int by = 0; by < 4; ++by) for (int r = 0; r < 16; ++r) for (int kb = 0; kb < 4; ++kb) {
    ea[by][r][kb] = (rand() % 9) - 4; eb[by][r][kb] = (rand() % 9) - 4;
  }
  for (int l = 0; l < 64; ++l) {
    uint32_t a = 0, b = 0;
    for (int by = 0; by < 4; ++by) { a |= (uint32_t)(127 + ea[by][l & 15][l >> 4]) << (8 * by); b |= (uint32_t)(127 + eb[by][l & 15][l >> 4]) << (8 * by); }
    SA[l] = a; SB[l] = b;
  }
  uint8_t *dA, *dB; uint32_t *dSA, *dSB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 1024); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
  auto check = [&](int opa, int opb) {
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double e = 0, nr = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      int m = (l >> 4) * 4 + i, n = l & 15;       // acc[i] of lane l = C[a-row (l>>4)*4+i][b-row l&15]
      double s = 0;
      for (int k = 0; k < 128; ++k)
        s += (double)e4m3_to_f(A[m * 128 + k]) * e4m3_to_f(B[n * 128 + k]) * ldexp(1.0, ea[opa][m][k >> 5] + eb[opb][n][k >> 5]);
      e += (C[l * 4 + i] - s) * (C[l * 4 + i] - s); nr += s * s;
    }
    printf("op_sel a=%d b=%d: relerr %.3e\n", opa, opb, sqrt(e / nr));
  };
  probe<0, 0><<<1, 64>>>(dA, dB, dSA, dSB, dC); check(0, 0);
  probe<1, 2><<<1, 64>>>(dA, dB, dSA, dSB, dC); check(1, 2);
  probe<3, 1><<<1, 64>>>(dA, dB, dSA, dSB, dC); check(3, 1);
  probe<2, 3><<<1, 64>>>(dA, dB, dSA, dSB, dC); check(2, 3);
  return 0;
}
