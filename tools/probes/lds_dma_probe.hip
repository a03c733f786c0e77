// What buffer_load_dword ... lds does on gfx950, measured (hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe):
// every lane of every wave DMA-loads 24 dwords from pseudo-random places of a global array into its wave's LDS slot (one
// 256-byte block per dword index: lane l's word expected at block + 4 l), waits, reads them back and compares with the same
// words loaded into registers.  Variants: the word's byte address carried by voffset / by soffset / by the instruction offset
// (does the instruction offset move the LDS address too?), and a wait of vmcnt(0) against vmcnt(2) with two younger register
// loads behind the DMA batch (do LDS-bound and register-bound loads retire in order?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr;

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void k(const unsigned *src, int n, unsigned *mism, int rounds)
{
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *slot = lds + wave * 24 * 256;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, n * 4, 0x00020000);
  unsigned bad = 0;
  unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  for (int it = 0; it < rounds; it++) {
    int idx[24];
#pragma unroll
    for (int v = 0; v < 24; v++) { seed = seed * 1664525u + 1013904223u; idx[v] = 2 + (int)((seed >> 8) % (unsigned)(n - 8)); }
#pragma unroll
    for (int v = 0; v < 24; v++) {
      if (VARIANT == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4, 0, 0, 0);
      if (VARIANT == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4 - 8, 8, 0, 0);
      if (VARIANT == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4 - 8, 0, 8, 0);
      if (VARIANT == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4, 0, 0, 0);
    }
    unsigned y0 = 0, y1 = 0;
    if (VARIANT == 3) {      // two younger register loads, then wait for all but two
      y0 = __builtin_amdgcn_raw_buffer_load_b32(r, idx[0] * 4 + 4, 0, 0);
      y1 = __builtin_amdgcn_raw_buffer_load_b32(r, idx[1] * 4 + 4, 0, 0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned *words = reinterpret_cast<const unsigned *>(slot) + lane;
    unsigned got[24];
#pragma unroll
    for (int v = 0; v < 24; v++) got[v] = words[v * 64];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int v = 0; v < 24; v++) bad += got[v] != src[idx[v]];
    bad += (y0 != src[idx[0] + 1]) + (VARIANT == 3 ? (y1 != src[idx[1] + 1]) : 0u);
  }
  if (bad) atomicAdd(mism, bad);
}

int main()
{
  const int n = 1 << 24;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; i++) h[i] = (unsigned)i * 2246822519u + 7u;
  unsigned *d, *m;
  hipMalloc(&d, n * 4); hipMalloc(&m, 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const char *names[4] = {"address in voffset, vmcnt(0)", "8 bytes of it in soffset", "8 bytes of it in the instruction offset",
                          "address in voffset, two younger register loads, vmcnt(2)"};
  for (int v = 0; v < 4; v++) {
    hipMemset(m, 0, 4);
    const int rounds = 2000;
    if (v == 0) hipLaunchKernelGGL(k<0>, dim3(768), dim3(256), 4 * 24 * 256, 0, d, n, m, rounds);
    if (v == 1) hipLaunchKernelGGL(k<1>, dim3(768), dim3(256), 4 * 24 * 256, 0, d, n, m, rounds);
    if (v == 2) hipLaunchKernelGGL(k<2>, dim3(768), dim3(256), 4 * 24 * 256, 0, d, n, m, rounds);
    if (v == 3) hipLaunchKernelGGL(k<3>, dim3(768), dim3(256), 4 * 24 * 256, 0, d, n, m, rounds);
    unsigned bad = 0;
    hipError_t e = hipMemcpy(&bad, m, 4, hipMemcpyDeviceToHost);
    printf("variant %d (%s): %u mismatching words of %llu%s\n", v, names[v], bad, 768ull * 256 * 24 * rounds,
           e == hipSuccess ? "" : "  [HIP error]");
  }
  return 0;
}
