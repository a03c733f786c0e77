// Probe (tools/ only, not part of the library): what does v_mfma_f64_4x4x4 cost on gfx950 next to v_fma_f64, do the two
// pipes overlap between waves of one SIMD, and which lane holds which element.  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_probe.hip -o tools/probes/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// mode 0: MFMA only, 1: FMA only, 2: even waves MFMA / odd waves FMA, 3: every wave 1 MFMA + 4 FMA interleaved
__global__ __launch_bounds__(512, 4) void k_rate(double *out, int n, int mode, double seed)
{
  const int wave = threadIdx.x >> 6;
  double a = seed + threadIdx.x, b = seed * 0.5 + 1.0;
  double c[8], f[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { c[i] = 0.0; f[i] = (double)i; }
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0) || mode == 3;
  const bool do_fma = mode == 1 || (mode == 2 && (wave & 1) == 1);
  if (mode == 3) {
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) f[(i + j) & 7] = fma(f[(i + j) & 7], a, b);
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    }
  } else if (do_fma) {
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = fma(f[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += c[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_layout(const double *a, const double *b, double *d)
{
  d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}

int main()
{
  double *out;
  const int blocks = 512, threads = 512;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n = 20000;
  const char *names[4] = {"mfma only (8 per trip, all waves)", "fma only (32 per trip, all waves)",
                          "even waves mfma x8, odd waves fma x32", "every wave: 8 x (1 mfma + 4 fma)"};
  for (int mode = 0; mode < 4; mode++) {
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, out, 100, mode, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, out, n, mode, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // 512 blocks x 8 waves = 4096 waves on 1024 SIMDs: 4 waves per SIMD
    const double waves_per_simd = blocks * (threads / 64) / 1024.0;
    double mf = 0, fm = 0;
    if (mode == 0) mf = 8.0 * n * waves_per_simd;
    if (mode == 1) fm = 32.0 * n * waves_per_simd;
    if (mode == 2) { mf = 8.0 * n * waves_per_simd / 2; fm = 32.0 * n * waves_per_simd / 2; }
    if (mode == 3) { mf = 8.0 * n * waves_per_simd; fm = 32.0 * n * waves_per_simd; }
    printf("%-45s %8.3f ms: per SIMD %.0f mfma + %.0f fma -> %.2f ns per mfma-slot, %.2f ns per fma-slot\n", names[mode], ms,
           mf, fm, mf > 0 ? ms * 1e6 / mf : 0.0, fm > 0 ? ms * 1e6 / fm : 0.0);
  }
  // lane layout with exact integer data: A lane l = 1000 + l, B lane l = (l == probe)
  std::vector<double> ha(64), hb(64), hd(64);
  double *da, *db, *dd;
  CHECK(hipMalloc(&da, 512)); CHECK(hipMalloc(&db, 512)); CHECK(hipMalloc(&dd, 512));
  for (int l = 0; l < 64; l++) ha[l] = 1000 + l;
  CHECK(hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice));
  printf("layout: for B = unit at lane p, D lanes that are non-zero and the A value they carry\n");
  for (int p = 0; p < 64; p += 1) {
    for (int l = 0; l < 64; l++) hb[l] = l == p ? 1.0 : 0.0;
    CHECK(hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    CHECK(hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost));
    printf("B lane %2d:", p);
    for (int l = 0; l < 64; l++) if (hd[l] != 0.0) printf("  D[%d]=A[%d]", l, (int)hd[l] - 1000);
    printf("\n");
  }
  return 0;
}
