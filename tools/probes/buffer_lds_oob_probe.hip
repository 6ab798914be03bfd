// What does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) write for lanes whose offset fails the
// resource's range check?  The implicit-GEMM conv loader wants ZEROS there (border taps), instead of a select between the image
// and a zero page per piece.  Prefills LDS with 0xAB bytes, loads 64 lanes x 16 B with odd lanes out of range, prints what landed.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const char* x, int nbytes, uint32_t* out, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 1024 / 4; i += 64) ((uint32_t*)smem)[i] = 0xABABABABu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0xFFFFFFF0u;                 // fails the range check
  if (threadIdx.x == 2) voff = nbytes - 8;                 // straddles the end: 8 bytes in, 8 bytes out
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((uint32_t*)smem)[i];
}
int main() {
  const int n = 4096;
  std::vector<uint32_t> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x10000000u + i;
  char* d; uint32_t* o;
  hipMalloc(&d, n); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  for (int soff : {0, 256}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, n, o, soff);
    std::vector<uint32_t> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("soffset %d:\n", soff);
    for (int lane = 0; lane < 6; ++lane)
      printf("  lane %d: %08x %08x %08x %08x\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    int zeros = 0, stale = 0;
    for (int lane = 1; lane < 64; lane += 2) for (int e = 0; e < 4; ++e) { zeros += r[lane * 4 + e] == 0; stale += r[lane * 4 + e] == 0xABABABABu; }
    printf("  odd (out-of-range) lanes: %d dwords zero, %d dwords stale of %d\n", zeros, stale, 32 * 4);
  }
  return 0;
}
