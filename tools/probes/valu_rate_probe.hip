// Probe (tools/ only, not part of the library): issue cost of the vector instructions of the level kernels' pixel loops on
// gfx950, in cycles per wave64 instruction, one wave per SIMD and four (as the kernels run).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o tools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 8 independent chains, 32 instructions per trip
#define BODY(ASM, CONSTRAINT_D, CONSTRAINT_S, TYPE_D, TYPE_S)                                                   \
  {                                                                                                              \
    TYPE_D d[8];                                                                                                 \
    TYPE_S s[8];                                                                                                 \
    for (int i = 0; i < 8; i++) { d[i] = (TYPE_D)(seed + i); s[i] = (TYPE_S)(seed * 3 + i + threadIdx.x); }      \
    for (int it = 0; it < n; it++) {                                                                             \
      _Pragma("unroll") for (int r = 0; r < 4; r++)                                                              \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : CONSTRAINT_D(d[i]) : CONSTRAINT_S(s[i])); \
    }                                                                                                            \
    double acc = 0;                                                                                              \
    for (int i = 0; i < 8; i++) acc += (double)d[i];                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                            \
  }

__global__ __launch_bounds__(256, 4) void k_probe(double *out, int n, int mode, double seed)
{
  switch (mode) {
    case 0: BODY("v_fma_f64 %0, %1, %1, %0", "+v", "v", double, double) break;
    case 1: BODY("v_mul_f64 %0, %1, %0", "+v", "v", double, double) break;
    case 2: BODY("v_add_f64 %0, %1, %0", "+v", "v", double, double) break;
    case 3: BODY("v_rcp_f64 %0, %1", "=v", "v", double, double) break;
    case 4: BODY("v_trunc_f64 %0, %1", "=v", "v", double, double) break;
    case 5: BODY("v_cvt_i32_f64 %0, %1", "=v", "v", int, double) break;
    case 6: BODY("v_mul_lo_u32 %0, %1, %0", "+v", "v", int, int) break;
    case 7: BODY("v_mad_i32_i24 %0, %1, %1, %0", "+v", "v", int, int) break;
    case 8: BODY("v_cmp_lt_f64 vcc, %1, %1", "+v", "v", double, double) break;
    case 9: BODY("v_cndmask_b32 %0, %1, %0, vcc", "+v", "v", int, int) break;
    case 10: BODY("v_mov_b64 %0, %1", "=v", "v", double, double) break;
    case 11: BODY("v_cvt_f64_i32 %0, %1", "=v", "v", double, int) break;
    case 12: BODY("v_floor_f64 %0, %1", "=v", "v", double, double) break;
    case 13: BODY("v_lshl_add_u32 %0, %1, 2, %0", "+v", "v", int, int) break;
    case 14: BODY("v_readlane_b32 s20, %1, 3", "+v", "v", int, int) break;
    case 15: BODY("v_fma_f32 %0, %1, %1, %0", "+v", "v", float, float) break;
    case 16: BODY("v_cndmask_b32_e64 %0, %1, %0, s[20:21]", "+v", "v", int, int) break;
    case 17: BODY("v_fmac_f64 %0, %1, %1", "+v", "v", double, double) break;
    case 18: BODY("v_fma_f64 %0, %1, s[20:21], %0", "+v", "v", double, double) break;
    case 19: BODY("v_cndmask_b32 %0, %1, %0, vcc\n\tv_add_u32 %0, %1, %0", "+v", "v", int, int) break;
    case 20: BODY("v_cmp_lt_i32 vcc, -1, %1\n\tv_cndmask_b32 %0, 0, %0, vcc", "+v", "v", int, int) break;
    case 21: BODY("v_and_b32 %0, %1, %0", "+v", "v", int, int) break;
  }
}

int main()
{
  const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_trunc_f64", "v_cvt_i32_f64", "v_mul_lo_u32",
                         "v_mad_i32_i24", "v_cmp_lt_f64", "v_cndmask_b32", "v_mov_b64", "v_cvt_f64_i32", "v_floor_f64",
                         "v_lshl_add_u32", "v_readlane_b32", "v_fma_f32", "v_cndmask_b32_e64 sgpr", "v_fmac_f64", "v_fma_f64 sgpr src",
                         "cndmask+add_u32 (2)", "cmp_i32+cndmask (2)", "v_and_b32"};
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double *out;
  CHECK(hipMalloc(&out, sizeof(double) * 256 * 4 * cus));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int n = 20000;
  printf("clock %d kHz, %d CUs; cycles per wave instruction at the reported clock (the effective one under load is lower)\n", prop.clockRate, cus);
  for (int wgs_per_cu = 4; wgs_per_cu <= 4; wgs_per_cu *= 4) {          // 256 threads = 1 wave per SIMD; x 4 = 4 waves per SIMD
    for (int mode = 0; mode < 22; mode++) {
      hipLaunchKernelGGL(k_probe, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 100, mode, 1.0);
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_probe, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, n, mode, 1.0);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_simd = (double)n * 32.0 * wgs_per_cu;       // waves per SIMD x instructions per wave
      printf("%d wave(s)/SIMD  %-16s %7.3f ms  %6.2f cycles per instruction per SIMD\n", wgs_per_cu, names[mode], ms,
             ms * 1e-3 * prop.clockRate * 1e3 / instr_per_simd);
    }
  }
  return 0;
}
