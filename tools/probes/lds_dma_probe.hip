// What buffer_load_dword ... lds does on gfx950, measured (hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe).
// Every lane of every wave DMA-loads 24 dwords from pseudo-random places of a global array into its wave's LDS slot (one
// 256-byte block per dword index: lane l's word expected at block + 4 l; the slot is filled with a poison word first), waits,
// reads them back and compares with the same words loaded into registers.
//   0  the word's byte address in voffset, s_waitcnt vmcnt(0)                      -> the mechanism itself
//   1  8 bytes of the address in soffset                                            -> soffset addresses memory only?
//   2  8 bytes of the address in the INSTRUCTION offset                             -> does it move the LDS address too?
//   3  24 LDS-bound loads, then 2 register-bound loads, s_waitcnt vmcnt(2)          -> may the 2 YOUNGER register loads retire first?
//   4  2 register-bound loads (raw asm, poisoned registers), then 24 LDS-bound loads, s_waitcnt vmcnt(24), registers copied at once
//                                                                                    -> may YOUNGER LDS-bound loads retire first?
//   5  12 + 12 LDS-bound loads, s_waitcnt vmcnt(12), first twelve checked           -> do LDS-bound loads retire in order among themselves?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr;
constexpr unsigned POISON = 0xdeadbeefu;

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void k(const unsigned *src, int n, unsigned long long *mism, int rounds)
{
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *slot = lds + wave * 24 * 256;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, n * 4, 0x00020000);
  unsigned long long bad = 0;
  unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  unsigned *mine = reinterpret_cast<unsigned *>(slot) + lane;
  for (int it = 0; it < rounds; it++) {
    int idx[24];
#pragma unroll
    for (int v = 0; v < 24; v++) { seed = seed * 1664525u + 1013904223u; idx[v] = 2 + (int)((seed >> 8) % (unsigned)(n - 8)); }
#pragma unroll
    for (int v = 0; v < 24; v++) mine[v * 64] = POISON;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned y0 = POISON, y1 = POISON, snap0 = 0, snap1 = 0;
    if (VARIANT == 4) {        // the compiler must not know these are loads: it would wait for them before the copy below
      const int o0 = idx[0] * 4 + 4, o1 = idx[1] * 4 + 4;
      asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_dword %1, %3, %4, 0 offen"
                   : "+v"(y0), "+v"(y1) : "v"(o0), "v"(o1), "s"(r) : "memory");
    }
#pragma unroll
    for (int v = 0; v < 24; v++) {
      if (VARIANT == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4 - 8, 8, 0, 0);
      else if (VARIANT == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4 - 8, 0, 8, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(slot + v * 256), 4, idx[v] * 4, 0, 0, 0);
    }
    int first_checked = 0, last_checked = 24;
    if (VARIANT == 3) {
      y0 = __builtin_amdgcn_raw_buffer_load_b32(r, idx[0] * 4 + 4, 0, 0);
      y1 = __builtin_amdgcn_raw_buffer_load_b32(r, idx[1] * 4 + 4, 0, 0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else if (VARIANT == 4) {
      asm volatile("s_waitcnt vmcnt(24)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(snap0), "=v"(snap1) : "v"(y0), "v"(y1) : "memory");
      first_checked = 24;      // (the LDS words are not what this variant is about)
    } else if (VARIANT == 5) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      last_checked = 12;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned got[24];
#pragma unroll
    for (int v = 0; v < 24; v++) got[v] = mine[v * 64];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int v = 0; v < 24; v++) bad += (v >= first_checked && v < last_checked && got[v] != src[idx[v]]) ? 1u : 0u;
    if (VARIANT == 4) bad += (snap0 != src[idx[0] + 1]) + (snap1 != src[idx[1] + 1]);
  }
  if (bad) atomicAdd(mism, bad);
}

int main()
{
  const int n = 1 << 24;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; i++) h[i] = (unsigned)i * 2246822519u + 7u;
  unsigned *d;
  unsigned long long *m;
  if (hipMalloc(&d, (size_t)n * 4) != hipSuccess || hipMalloc(&m, 8) != hipSuccess) return 1;
  if (hipMemcpy(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  const char *names[6] = {"address in voffset, vmcnt(0)", "8 bytes of it in soffset", "8 bytes of it in the instruction offset",
                          "24 LDS-bound loads, then 2 register loads, vmcnt(2): stale LDS words",
                          "2 register loads, then 24 LDS-bound loads, vmcnt(24): registers not yet written",
                          "12 + 12 LDS-bound loads, vmcnt(12): stale words among the first twelve"};
  const int rounds = 2000;
  for (int v = 0; v < 6; v++) {
    (void)hipMemset(m, 0, 8);
    const dim3 g(768), b(256);
    const size_t lds = 4 * 24 * 256;
    switch (v) {
      case 0: hipLaunchKernelGGL(k<0>, g, b, lds, 0, d, n, m, rounds); break;
      case 1: hipLaunchKernelGGL(k<1>, g, b, lds, 0, d, n, m, rounds); break;
      case 2: hipLaunchKernelGGL(k<2>, g, b, lds, 0, d, n, m, rounds); break;
      case 3: hipLaunchKernelGGL(k<3>, g, b, lds, 0, d, n, m, rounds); break;
      case 4: hipLaunchKernelGGL(k<4>, g, b, lds, 0, d, n, m, rounds); break;
      default: hipLaunchKernelGGL(k<5>, g, b, lds, 0, d, n, m, rounds); break;
    }
    unsigned long long bad = 0;
    const hipError_t e = hipMemcpy(&bad, m, 8, hipMemcpyDeviceToHost);
    printf("variant %d (%s): %llu mismatches in %llu lane-rounds%s\n", v, names[v], bad, 768ull * 256 * rounds,
           e == hipSuccess ? "" : "  [HIP error]");
  }
  return 0;
}
