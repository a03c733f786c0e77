// Gauss-Newton level kernel: the hot path of
//   phovo::Analytic::CPhotoconsistencyOdometryAnalytic::Optimize()
//   (phovo/include/CPhotoconsistencyOdometryAnalytic.h:500-563) with
//   ComputeResidualsAndJacobians (:191-367) and TestTerminationCriteria (:376-392) fused in.
//
// One workgroup owns one frame pair for one pyramid level and runs the WHOLE iteration loop of
// that level on the device: no host round trip per iteration, no materialised J[N x 6] / r[N]
// (the reference allocates, zeroes and re-reads 56 B/pixel of them every iteration).
//
// Per iteration (all fp64, wave64):
//   pass 1  every source pixel i: depth gate, unproject, SE(3) transform, project, C round(),
//           bounds -> atomicMax(owner[target], i).  The reference's serial raster loop lets the
//           LAST source pixel that lands on a target win the residual slot (:358); "largest source
//           index" is the same rule.  A 64-bit ballot per 64-pixel chunk remembers which pixels
//           were in bounds.
//   pass 2  every in-bounds pixel k: r_k = I1[k] - I0[owner[k]] (0 if nobody landed on k),
//           J_k = gx1[k]*Ju + gy1[k]*Jv from pixel k's own depth (the reference reads the gradient at
//           the SOURCE index, :346-347, and writes J at the source row, :351-356, while r is scattered,
//           so row k of J meets residual k in J^T r, :538), then 21 + 6 FMAs into the upper triangle of
//           J^T J and J^T r held in registers.
//   reduce  per-wave transposed butterfly (32 shuffles for 32 values instead of 6 x 27), one LDS row
//           per wave, fixed-order sum across waves -> bitwise reproducible.
//   solve   wave 0: 6x6 Gaussian elimination with partial pivoting, state -= lambda * H^-1 g,
//           termination test, pose constants of the next iteration.
// No MFMA: J^T J is a 6 x N by N x 6 contraction, a reduction, not a GEMM tile.
//
// The reference's `temp11 = cos(pitch)*cos(yaw)+x` transcription bug (:253) is reproduced on purpose:
// parity with the reference's poses requires it ("read this first" item 4).

#include <hip/hip_runtime.h>

#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

constexpr int WAVE = 64;
constexpr int NACC = 27;     // 21 upper-triangular J^T J + 6 J^T r
constexpr int NRED = 32;     // padded for the butterfly

// Indices into the pose-constant block in LDS.
enum {
  C_X = 0, C_Y, C_Z, C_R01, C_R02, C_R11, C_R12,
  C_T1, C_T2, C_T3, C_T4, C_T5, C_T6, C_T8, C_T10, C_T11, C_T12, C_T13, C_T14, C_T15,
  C_T16, C_T17, C_T18, C_T19, C_T20, C_T24, C_COUNT
};

// Control words in LDS.
enum { CTL_DONE = 0, CTL_FLAGS = 1, CTL_COUNT = 4 };

__device__ __forceinline__ double uniform_f64(double v)
{
  // The value is identical in every lane: move it to scalar registers so that the per-pixel
  // math reads it as an SGPR operand instead of burning two VGPRs per constant.
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Pose constants from the state: Rt (:219-241) and temp1..temp24 (:243-266), written with the
// reference's association.  temp7 = -temp6, temp9 = -temp8, temp21 = -temp5, temp22 = temp2,
// temp23 = temp1 hold exactly in IEEE arithmetic and are not stored; Rt(0,0) = temp15,
// Rt(1,0) = temp14, Rt(2,0) = -temp3, Rt(2,1) = temp1, Rt(2,2) = temp2 likewise.
__device__ void write_pose_constants(const double s[6], double *cst)
{
  const double x = s[0], y = s[1], z = s[2];
  double sy, cy, sp, cp, sr, cr;
  sincos(s[3], &sy, &cy);
  sincos(s[4], &sp, &cp);
  sincos(s[5], &sr, &cr);
  cst[C_X] = x; cst[C_Y] = y; cst[C_Z] = z;
  cst[C_R01] = cy * sp * sr - sy * cr;
  cst[C_R02] = cy * sp * cr + sy * sr;
  cst[C_R11] = sy * sp * sr + cy * cr;
  cst[C_R12] = sy * sp * cr - cy * sr;
  cst[C_T1] = cp * sr;
  cst[C_T2] = cp * cr;
  cst[C_T3] = sp;
  cst[C_T4] = sr * sy + sp * cr * cy;
  cst[C_T5] = sp * sr * cy - cr * sy;
  cst[C_T6] = sp * sr * sy + cr * cy;
  cst[C_T8] = sr * cy - sp * cr * sy;
  cst[C_T10] = cp * sr * cy;
  cst[C_T11] = cp * cy + x;          // the reference's bug, kept (:253)
  cst[C_T12] = cp * cr * cy;
  cst[C_T13] = sp * cy;
  cst[C_T14] = cp * sy;
  cst[C_T15] = cp * cy;
  cst[C_T16] = sp * sr;
  cst[C_T17] = sp * cr;
  cst[C_T18] = cp * sr * sy;
  cst[C_T19] = cp * cr * sy;
  cst[C_T20] = sp * sy;
  cst[C_T24] = cp;
}

// One butterfly stage of the transposed wave reduction: N values in, N/2 out.
template <int N>
__device__ __forceinline__ void reduce_stage(double (&v)[NRED], int lane, int dist)
{
  const bool up = (lane & dist) != 0;
#pragma unroll
  for (int i = 0; i < N / 2; i++) {
    const double send = up ? v[i] : v[i + N / 2];
    const double keep = up ? v[i + N / 2] : v[i];
    v[i] = keep + __shfl_xor(send, dist, WAVE);
  }
}

// 6x6 solve by Gaussian elimination with partial pivoting, fully unrolled so that every index is a
// compile-time constant (no scratch).  a is the augmented matrix [H | g]; returns H^-1 g in x.
__device__ void solve6(double (&a)[6][7], double (&x)[6])
{
#pragma unroll
  for (int k = 0; k < 6; k++) {
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      const bool sw = fabs(a[r][k]) > fabs(a[k][k]);
#pragma unroll
      for (int c = k; c < 7; c++) {
        const double u = a[k][c], l = a[r][c];
        a[k][c] = sw ? l : u;
        a[r][c] = sw ? u : l;
      }
    }
    const double piv = a[k][k];
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      const double f = a[r][k] / piv;
#pragma unroll
      for (int c = k + 1; c < 7; c++) a[r][c] -= f * a[k][c];
    }
  }
#pragma unroll
  for (int r = 5; r >= 0; r--) {
    double s = a[r][6];
#pragma unroll
    for (int c = r + 1; c < 6; c++) s -= a[r][c] * x[c];
    x[r] = s / a[r][r];
  }
}

template <int T, bool SRC_LDS, bool OWNER_LDS>
__global__ __launch_bounds__(T) void gn_level_kernel(const GNLevelArgs A)
{
  constexpr int NW = T / WAVE;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // LDS carve-up (all offsets multiples of 8):
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW * NRED);             // [CTL_COUNT]
  unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(s_ctl + CTL_COUNT);  // [n_chunks]
  unsigned char *p = reinterpret_cast<unsigned char *>(s_mask + A.n_chunks);
  int *s_owner = reinterpret_cast<int *>(p);                           // [n]      (OWNER_LDS)
  if (OWNER_LDS) p += sizeof(int) * ((A.n + 1) & ~1);
  double *s_i0 = reinterpret_cast<double *>(p);                        // [n]      (SRC_LDS)

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = tid / WAVE;
  const int pair = blockIdx.x;
  const int n = A.n, W = A.w, H = A.h;
  const float inv_w = 1.0f / (float)W;

  const size_t fstride = (size_t)PLANES_PER_FRAME * (size_t)n;
  const double *src_frame = A.planes + (size_t)A.src[pair] * fstride;
  const double *tgt_frame = A.planes + (size_t)A.tgt[pair] * fstride;
  const double *__restrict__ I0 = src_frame + (size_t)PLANE_I * n;
  const double *__restrict__ D0 = src_frame + (size_t)PLANE_D * n;
  const double *__restrict__ I1 = tgt_frame + (size_t)PLANE_I * n;
  const double *__restrict__ GX = tgt_frame + (size_t)PLANE_GX * n;
  const double *__restrict__ GY = tgt_frame + (size_t)PLANE_GY * n;
  int *g_owner = OWNER_LDS ? nullptr : A.g_owner + (size_t)pair * (size_t)n;

  // ---- level prologue -------------------------------------------------------------------
  if (OWNER_LDS) {
    for (int k = tid; k < n; k += T) s_owner[k] = -1;
  }   // the global owner map is cleared by the host before the launch and re-cleared in pass 2
  if (SRC_LDS) {
    for (int k = tid; k < n; k += T) s_i0[k] = I0[k];
  }
  if (wave == 0) {
    double st[6];
#pragma unroll
    for (int j = 0; j < 6; j++) st[j] = A.states[(size_t)pair * 6 + j];
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) s_state[j] = st[j];
      write_pose_constants(st, s_cst);
      s_ctl[CTL_DONE] = 0;
      s_ctl[CTL_FLAGS] = 0;
    }
  }
  __syncthreads();

  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double min_d = A.min_depth, max_d = A.max_depth;
  const double dW = (double)W, dH = (double)H;

  int iteration = 0;
  double last_gnorm = 0.0;
  while (true) {
    // ---- constants of this iteration (uniform -> SGPRs) -----------------------------------
    const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
    const double r01 = uniform_f64(s_cst[C_R01]), r02 = uniform_f64(s_cst[C_R02]);
    const double r11 = uniform_f64(s_cst[C_R11]), r12 = uniform_f64(s_cst[C_R12]);
    const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
    const double t14 = uniform_f64(s_cst[C_T14]), t15 = uniform_f64(s_cst[C_T15]);

    // ---- pass 1: warp every source pixel, resolve who owns each target pixel -------------
    for (int chunk = wave; chunk < A.n_chunks; chunk += NW) {
      const int k = chunk * WAVE + lane;
      bool inb = false;
      if (k < n) {
        const double pz = D0[k];                                        // :279
        if (min_d < pz && pz < max_d) {                                 // :280
          int r = (int)(((float)k + 0.5f) * inv_w);
          int c = k - r * W;
          if (c < 0) { r -= 1; c += W; }
          if (c >= W) { r += 1; c -= W; }
          const double px = ((double)c - ox) * pz * ifx;                // :282
          const double py = ((double)r - oy) * pz * ify;                // :283
          const double X = ((t15 * px + r01 * py) + r02 * pz) + cx;     // Rt*point3D  :291
          const double Y = ((t14 * px + r11 * py) + r12 * pz) + cyy;
          const double Z = ((-t3 * px + t1 * py) + t2 * pz) + cz;
          const double iz = 1.0 / Z;                                    // :294
          const double tc = (X * fx) * iz + ox;                         // :295
          const double tr = (Y * fy) * iz + oy;                         // :296
          const double rr = round(tr), rc = round(tc);                  // C round(), half away  :297-298
          if (rr >= 0.0 && rr < dH && rc >= 0.0 && rc < dW) {           // :302-303 (NaN fails)
            inb = true;
            const int t = (int)rr * W + (int)rc;
            if (OWNER_LDS) atomicMax(&s_owner[t], k);                   // last raster writer wins  :358
            else atomicMax(&g_owner[t], k);
          }
        }
      }
      const unsigned long long m = __ballot(inb);
      if (lane == 0) s_mask[chunk] = m;
    }
    __syncthreads();

    // ---- pass 2: residual, Jacobian row, normal-equation accumulation ---------------------
    const double t4 = uniform_f64(s_cst[C_T4]), t5 = uniform_f64(s_cst[C_T5]), t6 = uniform_f64(s_cst[C_T6]);
    const double t8 = uniform_f64(s_cst[C_T8]), t10 = uniform_f64(s_cst[C_T10]), t11 = uniform_f64(s_cst[C_T11]);
    const double t12 = uniform_f64(s_cst[C_T12]), t13 = uniform_f64(s_cst[C_T13]);
    const double t16 = uniform_f64(s_cst[C_T16]), t17 = uniform_f64(s_cst[C_T17]), t18 = uniform_f64(s_cst[C_T18]);
    const double t19 = uniform_f64(s_cst[C_T19]), t20 = uniform_f64(s_cst[C_T20]), t24 = uniform_f64(s_cst[C_T24]);
    const double t7 = -t6, t9 = -t8, t21 = -t5;

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;

    for (int chunk = wave; chunk < A.n_chunks; chunk += NW) {
      const int k = chunk * WAVE + lane;
      int o = -1;
      if (k < n) {
        if (OWNER_LDS) {
          o = s_owner[k];
          s_owner[k] = -1;                        // ready for the next iteration
        } else {
          o = __hip_atomic_load(&g_owner[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&g_owner[k], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const unsigned long long m = s_mask[chunk];
      if (m == 0ull) continue;                    // wave-uniform
      if ((m >> lane) & 1ull) {
        const double pz = D0[k];
        const double gxi = GX[k];                 // gradient at the SOURCE index  :346-347
        const double gyi = GY[k];
        double res = 0.0;
        if (o >= 0) {
          const double pixel1 = SRC_LDS ? s_i0[o] : I0[o];              // :308 of the owning source pixel
          const double pixel2 = I1[k];                                  // :309
          res = pixel2 - pixel1;                                        // :358
        }
        int r = (int)(((float)k + 0.5f) * inv_w);
        int c = k - r * W;
        if (c < 0) { r -= 1; c += W; }
        if (c >= W) { r += 1; c -= W; }
        const double px = ((double)c - ox) * pz * ifx;
        const double py = ((double)r - oy) * pz * ify;

        const double t25 = 1.0 / (cz + py * t1 + pz * t2 - px * t3);    // :313
        const double t26 = t25 * t25;                                   // :314
        const double Au = pz * t4 + py * t5 + px * t11;                 // (pz*temp4+py*temp5+px*temp11)
        const double Bv = py * t6 + pz * t9 + px * t14 + cyy;           // (py*temp6+pz*temp9+px*temp14+y)
        const double Cm = -py * t16 - pz * t17 - px * t24;              // d(Z)/d(pitch)
        const double Dm = py * t2 - pz * t1;                            // (py*temp22-pz*temp23)

        const double ju0 = fx * t25;                                                    // :317
        const double jv1 = fy * t25;                                                    // :322
        const double ju2 = -fx * Au * t26;                                              // :325
        const double jv2 = -fy * Bv * t26;                                              // :326
        const double ju3 = fx * (py * t7 + pz * t8 - px * t14) * t25;                   // :329
        const double jv3 = fy * (pz * t4 + py * t5 + px * t15) * t25;                   // :330
        const double ju4 = fx * (py * t10 + pz * t12 - px * t13) * t25 - fx * Cm * Au * t26;   // :333-334
        const double jv4 = fy * (py * t18 + pz * t19 - px * t20) * t25 - fy * Cm * Bv * t26;   // :335-336
        const double ju5 = fx * (py * t4 + pz * t21) * t25 - fx * Dm * Au * t26;        // :339-340
        const double jv5 = fy * (pz * t7 + py * t9) * t25 - fy * Dm * Bv * t26;         // :341-342

        double J[6];                                                                    // :348
        J[0] = gxi * ju0 + gyi * 0.0;
        J[1] = gxi * 0.0 + gyi * jv1;
        J[2] = gxi * ju2 + gyi * jv2;
        J[3] = gxi * ju3 + gyi * jv3;
        J[4] = gxi * ju4 + gyi * jv4;
        J[5] = gxi * ju5 + gyi * jv5;

        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
          for (int b = a; b < 6; b++) {
            acc[q] = fma(J[a], J[b], acc[q]);                                           // J^T J  :540
            q++;
          }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] = fma(J[a], res, acc[21 + a]);           // J^T r  :538
      }
    }

    // ---- wave-level transposed butterfly: 32 shuffles, lane l ends with value index idx(l) ----
    reduce_stage<32>(acc, lane, 32);
    reduce_stage<16>(acc, lane, 16);
    reduce_stage<8>(acc, lane, 8);
    reduce_stage<4>(acc, lane, 4);
    reduce_stage<2>(acc, lane, 2);
    {
      const double total = acc[0] + __shfl_xor(acc[0], 1, WAVE);
      const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 +
                      ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
      if ((lane & 1) == 0) s_red[wave * NRED + idx] = total;
    }
    __syncthreads();

    // ---- wave 0: cross-wave sum (fixed order), solve, update, terminate ------------------------
    if (wave == 0) {
      double v = 0.0;
      if (lane < NRED) {
#pragma unroll 4
        for (int w2 = 0; w2 < NW; w2++) v += s_red[w2 * NRED + lane];
      }
      double a[6][7];
      {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
          for (int j = i; j < 6; j++) {
            const double hij = __shfl(v, q, WAVE);
            a[i][j] = hij;
            a[j][i] = hij;
            q++;
          }
        }
      }
      double g[6];
#pragma unroll
      for (int i = 0; i < 6; i++) { g[i] = __shfl(v, 21 + i, WAVE); a[i][6] = g[i]; }

      double step[6];
      solve6(a, step);
      double st[6];
      bool finite = true;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        st[i] = s_state[i] - A.lambda * step[i];                                        // :539
        finite = finite && (fabs(st[i]) <= 1.79769313486231570815e308);
      }
      double gn2 = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) gn2 += g[i] * g[i];
      const double gnorm = sqrt(gn2);                                                   // :380
      const int it = iteration + 1;                                                     // :547
      bool done = false;
      if (it >= A.max_iter) done = true;                                                // :383
      else if (gnorm < A.min_grad_norm) done = true;                                    // :388
      if (!finite) done = true;     // the reference would keep iterating on NaN; the result is the same NaN
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) s_state[i] = st[i];
        s_ctl[CTL_DONE] = done ? 1 : 0;
        if (!finite) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_NONFINITE;
        if (!done) write_pose_constants(st, s_cst);
      }
      last_gnorm = gnorm;
    }
    __syncthreads();
    iteration++;
    if (s_ctl[CTL_DONE]) break;
  }

  // ---- epilogue: state and report back to HBM -------------------------------------------------
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 6; j++) A.states[(size_t)pair * 6 + j] = s_state[j];
    if (A.reports) {
      A.reports[pair].iterations[A.level] = iteration;
      A.reports[pair].gradient_norm = last_gnorm;
      A.reports[pair].flags |= (uint32_t)s_ctl[CTL_FLAGS];
    }
  }
}

size_t lds_fixed_bytes(int threads, int n_chunks)
{
  const int nw = threads / WAVE;
  return sizeof(double) * (32 + 8 + (size_t)nw * NRED) + sizeof(int) * CTL_COUNT +
         sizeof(unsigned long long) * (size_t)n_chunks;
}

constexpr size_t LDS_LIMIT = 160 * 1024;   // MI355X: 160 KiB per CU, one workgroup may take all of it

template <int T, bool SRC_LDS, bool OWNER_LDS>
hipError_t launch_inst(const GNLevelArgs &a, int n_pairs, size_t lds, hipStream_t stream)
{
  hipLaunchKernelGGL((gn_level_kernel<T, SRC_LDS, OWNER_LDS>), dim3(n_pairs), dim3(T), lds, stream, a);
  return hipGetLastError();
}

template <int T, bool SRC_LDS, bool OWNER_LDS>
hipError_t prepare_inst()
{
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&gn_level_kernel<T, SRC_LDS, OWNER_LDS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
}

}  // namespace

bool gn_plan_level(int n, GNLaunchPlan *plan)
{
  const int n_chunks = (n + WAVE - 1) / WAVE;
  // Workgroup size: enough waves to cover the level a few times over, at most 1024 threads.
  int threads = 1024;
  if (n <= 64 * 4) threads = 256;
  else if (n <= 64 * 16) threads = 512;
  const size_t fixed = lds_fixed_bytes(threads, n_chunks);
  if (fixed > LDS_LIMIT) return false;
  const size_t owner = sizeof(int) * (size_t)((n + 1) & ~1);
  const size_t src = sizeof(double) * (size_t)n;
  plan->threads = threads;
  plan->owner_in_lds = fixed + owner <= LDS_LIMIT;
  plan->source_in_lds = plan->owner_in_lds && (fixed + owner + src <= LDS_LIMIT);
  plan->lds_bytes = (int)(fixed + (plan->owner_in_lds ? owner : 0) + (plan->source_in_lds ? src : 0));
  return true;
}

hipError_t gn_prepare_kernels()
{
  hipError_t e;
#define PHOVO_PREP(T)                                                     \
  if ((e = prepare_inst<T, true, true>()) != hipSuccess) return e;        \
  if ((e = prepare_inst<T, false, true>()) != hipSuccess) return e;       \
  if ((e = prepare_inst<T, false, false>()) != hipSuccess) return e;
  PHOVO_PREP(256)
  PHOVO_PREP(512)
  PHOVO_PREP(1024)
#undef PHOVO_PREP
  return hipSuccess;
}

hipError_t gn_launch_level(const GNLevelArgs &a, const GNLaunchPlan &plan, int n_pairs,
                           hipStream_t stream)
{
  if (n_pairs <= 0) return hipSuccess;
  const size_t lds = (size_t)plan.lds_bytes;
#define PHOVO_DISPATCH(T)                                                                   \
  if (plan.threads == T) {                                                                  \
    if (plan.source_in_lds) return launch_inst<T, true, true>(a, n_pairs, lds, stream);     \
    if (plan.owner_in_lds) return launch_inst<T, false, true>(a, n_pairs, lds, stream);     \
    return launch_inst<T, false, false>(a, n_pairs, lds, stream);                           \
  }
  PHOVO_DISPATCH(256)
  PHOVO_DISPATCH(512)
  PHOVO_DISPATCH(1024)
#undef PHOVO_DISPATCH
  return hipErrorInvalidValue;
}

}  // namespace phovo_hip
