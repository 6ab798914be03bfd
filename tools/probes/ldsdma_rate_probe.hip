// How many bytes per clock can ONE CU pull in through its vector memory path when the data sit in the XCD's L2 (the
// steady state of the GEMM main loop: a K-step of the 256 x 192 tile is 56 KiB per CU, each line shared by 5-6 CUs of the XCD)?
// The ping-pong loop's per-K-step time follows the BYTES of the tile (256x256 / 224 / 192: 2140 / 2020 / 1750 cycles for
// 64 / 60 / 56 KiB), not its MFMAs - ~32 cycles per 1-KiB LDS-DMA piece.  This probe measures the ceiling directly:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, what the loop uses), 8 waves per CU, a counted number of pieces in flight
//   mode 1  global_load_dwordx4 into VGPRs (discarded)            - is the limit the LDS-DMA write side or the L1 / TA path?
//   mode 2  LDS-DMA with the nt bit, mode 3 with sc1 (L1 bypass policies)
//   mode 4  LDS-DMA, every wave re-reading ONE 8-KiB window (L1 hits)   - the path without L2 latency
// footprint F per XCD (all CUs of an XCD walk the same region, like GEMM tiles sharing operand panels): 512 KiB (L2) / 32 MiB (MALL)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/ldsdma_rate_probe.hip -o /tmp/ldsrate && /tmp/ldsrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* base, long long foot, int iters, int inflight_pieces, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const char* region = base + (long long)xcd * foot;
  // piece p of this wave at step it: 1 KiB, rows like a GEMM tile (8 rows x 128 B); the CUs of an XCD are spread over 6 "tiles"
  // so that ~5 CUs read the same bytes at about the same time
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float sink = 0.f;
  const long long tile_off = (long long)(cu % 6) * 65536;
  if constexpr (MODE == 1) {
    // two register sets of 8 x 16 B per lane, alternated: a set is only named again (an empty asm use) after the counted wait
    // that lands it, so hipcc keeps its registers allocated while the loads fly
    f4v va[8], vb[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) vb[p] = f4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const char* src = region + (tile_off + ((long long)it * 8 + p) * 8192 + wave * 1024 + lane * 16) % foot;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(va[p]) : "v"(src) : "memory");
      }
      if (inflight_pieces >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(vb[p]));
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const char* src = region + (tile_off + ((long long)(it + 1) * 8 + p) * 8192 + wave * 1024 + lane * 16) % foot;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vb[p]) : "v"(src) : "memory");
      }
      if (inflight_pieces >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(va[p]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p) { asm volatile("" : "+v"(vb[p])); sink += vb[p].x; }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      long long off = MODE == 4 ? (long long)(p * 1024 + lane * 16)
                                : (tile_off + ((long long)it * 8 + p) * 8192 + wave * 1024 + lane * 16) % foot;
      const char* src = region + off;
      char* dst = smem + (wave * 8 + p) * 1024;
      if constexpr (MODE == 0 || MODE == 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      else if constexpr (MODE == 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
      else if constexpr (MODE == 3)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 16);
    }
    // keep `inflight_pieces` of this wave's pieces in flight (counted wait, never a drain in steady state)
    if (inflight_pieces >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (sink == 1.2345f) cyc[0] = 0;
}

template <int MODE>
void run(const char* name, const char* d, long long foot, int nwaves, int inflight, unsigned long long* dc) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(nwaves * 64), nwaves * 8 * 1024, 0, d, foot, iters, inflight, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nwaves; ++w) mean += (double)h[b * 8 + w];
  mean /= 256.0 * nwaves;
  const double bytes_cu = (double)iters * 8 * 1024 * nwaves;
  printf("%-34s foot/XCD %6lld KiB  %d waves/CU  in flight/wave %2d: %7.1f us  %6.2f TB/s chip  %5.1f B/clk/CU (s_memtime)  %.0f cycles per KiB piece per CU\n",
         name, foot >> 10, nwaves, inflight >= 8 ? 16 : 8, ms * 1e3, bytes_cu * 256 / (ms * 1e-3) / 1e12, bytes_cu / mean, mean / (iters * 8.0 * nwaves));
}

int main() {
  const long long total = 8LL * (64 << 20);
  char* d; unsigned long long* dc;
  hipMalloc(&d, total); hipMalloc(&dc, 256 * 8 * 8);
  hipMemset(d, 1, total);
  for (long long foot : {512LL << 10, 32LL << 20}) {
    for (int nwaves : {8, 4}) {
      run<0>("LDS-DMA global_load_lds_dwordx4", d, foot, nwaves, 8, dc);
      run<0>("LDS-DMA global_load_lds_dwordx4", d, foot, nwaves, 0, dc);
      run<1>("global_load_dwordx4 -> VGPR", d, foot, nwaves, 8, dc);
      run<2>("LDS-DMA nt", d, foot, nwaves, 8, dc);
      run<3>("LDS-DMA sc1", d, foot, nwaves, 8, dc);
    }
  }
  run<4>("LDS-DMA, L1-resident 8 KiB window", d, 512 << 10, 8, 8, dc);
  run<4>("LDS-DMA, L1-resident 8 KiB window", d, 512 << 10, 4, 8, dc);
  return 0;
}
