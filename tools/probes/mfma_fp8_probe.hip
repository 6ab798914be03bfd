// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: operand byte layout, E8M0 scale encoding, C/D layout.
// hipcc --offload-arch=gfx950 -O2 mfma_fp8_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// e4m3fn decode (OCP)
static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -r : r;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int scale_a, int scale_b) {
  // A [16][128] row-major (row = output row m, k contiguous), B [16][128] (row = output col n, k contiguous)
  const int l = threadIdx.x, r = l & 15, kb = l >> 4;
  i32x8 a, b;
  const int* ap = (const int*)(A + r * 128 + kb * 32);
  const int* bp = (const int*)(B + r * 128 + kb * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
  // hypothesis: acc[reg] = C[row = (l>>4)*4 + reg][col = l&15] where "a" rows are M and "b" rows are N
  for (int i = 0; i < 4; ++i) C[l * 4 + i] = acc[i];
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  srand(1);
  for (auto& v : A) v = (uint8_t)(rand() & 0xff);
  for (auto& v : B) v = (uint8_t)(rand() & 0xff);
  for (auto& v : A) if ((v & 0x7f) == 0x7f) v = 0x38;    // no NaN
  for (auto& v : B) if ((v & 0x7f) == 0x7f) v = 0x38;
  for (auto& v : A) v &= 0xbf;                             // keep exponents small: |x| < 2
  for (auto& v : B) v &= 0xbf;
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<double> ref(256);
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double s = 0; for (int k = 0; k < 128; ++k) s += (double)e4m3_to_f(A[m * 128 + k]) * e4m3_to_f(B[n * 128 + k]);
    ref[m * 16 + n] = s;
  }
  int scales[] = {0x7f7f7f7f, 0x7f, 0, (int)0x80808080};
  for (int sc : scales) {
    probe<<<1, 64>>>(dA, dB, dC, sc, sc);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    // try both layouts
    double e1 = 0, e2 = 0, nr = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      int row = (l >> 4) * 4 + i, col = l & 15;
      double v = C[l * 4 + i];
      e1 += (v - ref[row * 16 + col]) * (v - ref[row * 16 + col]);     // a = M rows, b = N cols
      e2 += (v - ref[col * 16 + row]) * (v - ref[col * 16 + row]);     // swapped
      nr += ref[row * 16 + col] * ref[row * 16 + col];
    }
    printf("scale 0x%08x: relerr(M=a,N=b) %.3e  relerr(swapped) %.3e   C[0]=%g ref=%g ratio %g\n", sc, sqrt(e1 / nr), sqrt(e2 / nr), C[0], ref[0], C[0] / ref[0]);
  }
  return 0;
}
