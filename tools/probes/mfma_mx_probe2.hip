// Which C entries does lane t's E8M0 scale byte touch?  All operand elements are 1.0 in K block kbA only (others 0) so the K block is visible too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int OPA, int OPB>
__global__ void probe(const uint32_t* SA, const uint32_t* SB, float* C, int kb_on) {
  const int l = threadIdx.x, kb = l >> 4;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = kb == kb_on ? 0x38383838 : 0; b[i] = a[i]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPA, (int)SA[l], OPB, (int)SB[l]);
  for (int i = 0; i < 4; ++i) C[l * 4 + i] = acc[i];
}
int main() {
  uint32_t *dSA, *dSB; float* dC;
  hipMalloc(&dC, 1024); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256);
  for (int which = 0; which < 2; ++which)
    for (int kb_on = 0; kb_on < 4; kb_on += 3)
      for (int t = 0; t < 64; t += 1) {
        std::vector<uint32_t> SA(64, 0x7f7f7f7f), SB(64, 0x7f7f7f7f);
        (which ? SB : SA)[t] = 0x7f7f7f80;            // byte 0 = 2^1
        hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
        probe<0, 0><<<1, 64>>>(dSA, dSB, dC, kb_on);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int rows = 0, cols = 0, cnt = 0;   // bit sets of (a-row = (l>>4)*4+i, b-row = l&15) whose value != 32
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (C[l * 4 + i] != 32.f) { rows |= 1 << ((l >> 4) * 4 + i); cols |= 1 << (l & 15); ++cnt; }
        if (cnt) printf("%s lane %2d (r %2d kb %d), operand K block %d on: %3d entries changed, a-rows %04x b-rows %04x\n", which ? "scale_b" : "scale_a", t, t & 15, t >> 4, kb_on, cnt, rows, cols);
      }
  return 0;
}
