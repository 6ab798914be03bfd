// What does a CU's LDS deliver to ds_read_b128 - the fragment reads of the GEMM main loop?  DESIGN.md 3.1 priced a K-step of the
// 256 x 192 ping-pong tile at "56 pieces x 16 + 160 reads x 4 = 1536 LDS cycles" (256 B/clk for a conflict-free ds_read_b128);
// the phase trace has group 0's memory phase - 80 reads + 32 LDS-DMA pieces - at ~960 cycles, which fits 8 cycles per read
// (128 B/clk) better than 4.  This probe measures it: W waves per CU (4 = one ping-pong group, 8 = the whole workgroup) issue the
// loop's own access pattern (16 rows x 128 B with the XOR chunk swizzle: 16 distinct 16-byte slots per 16 lanes) back to back,
// NRD reads per lgkmcnt(0) wait, and s_memtime brackets the run.
//   mode 0: the swizzled fragment pattern (conflict-free by construction)   mode 1: lane-linear 16 B (also conflict-free)
//   mode 2: all lanes of a 16-lane group on ONE 128-byte row, unswizzled (bank conflicts: the yardstick for "conflicted")
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/lds_read_rate_probe.hip -o /tmp/ldsrd && /tmp/ldsrd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((int*)smem)[i] = i;
  __syncthreads();
  const int r16 = lane & 15, q4 = lane >> 4;
  uint32_t off;
  if (MODE == 0) off = r16 * 128 + ((q4 ^ (lane & 7)) << 4);
  else if (MODE == 1) off = lane * 16;
  else off = q4 * 128 + (r16 >> 3) * 16;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 2048 * 4 + off;
  i4v v[10];
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(base), "n"((r % 4) * 2048) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  int s = 0;
#pragma unroll
  for (int r = 0; r < 10; ++r) s += v[r].x;
  if (s == 0x7fffffff) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(int waves, int iters, unsigned long long* d, int* sink, const char* what) {
  unsigned long long h[256 * 8];
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 65536, 0, iters, d, sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) mean += (double)h[b * 8 + w];
  mean /= 256.0 * waves;
  const double reads = (double)iters * 10 * waves;           // wave-level ds_read_b128 per CU
  printf("%-52s %d waves/CU: %.2f cycles per wave-read (s_memtime ticks = shader clocks?), %.1f B per tick per CU\n", what, waves,
         mean / reads, reads * 1024.0 / mean);
}

int main() {
  unsigned long long* d;
  int* sink;
  hipMalloc(&d, 256 * 8 * 8);
  hipMalloc(&sink, 4);
  for (int waves : {1, 4, 8}) {
    run<0>(waves, 2000, d, sink, "fragment pattern (XOR-swizzled 128-B rows)");
    run<1>(waves, 2000, d, sink, "lane-linear 16 B");
    run<2>(waves, 2000, d, sink, "16 lanes on two 16-B slots of one row (conflicts)");
  }
  return 0;
}
