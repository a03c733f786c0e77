// EXTENSION, NOT IN THE REFERENCE'S ANALYTIC PATH (PHOVO_SAMPLING_BILINEAR): forward-additive alignment with bilinear
// sampling, taps gathered from global memory (any level size), 256 threads x 2 workgroups per CU.
// Round 5, fp16 planes only: a tap is ONE RECORD {I, GX, GY, pad} of the target frame -- 8 bytes, one load -- instead of three
// 2-byte gathers from three planes (pyr_build_tap_records keeps the records behind the planes, same bits): 4 loads per pixel
// where there were 12, 186-196 k -> 224 k alignments/s (+14 %).  The same records on fp32 planes (16 bytes, 4 loads instead
// of 6 pair loads) lost 5 %, on fp64 planes (32 bytes: a 16- and an 8-byte load per tap, 8 instead of 12) changed nothing
// (178.1 k against 178.3 k: 80x60 5 % faster, 160x120 3 % slower): it is not the NUMBER of loads that holds this kernel at
// 0.54 on fp64 planes, and more bytes per tap cost what fewer instructions buy (profiles/r05_runs/bilinear_records_ab.txt).
// Those two storages keep their gathers.
// Also measured and not kept (round 4, profiles/r04_runs/bilinear_roles_ab.txt): two kinds of waves as in the sliding-window
// kernel -- samplers (warp, taps, interpolation) leave depth, 1/Z', residual and both gradients in LDS, accumulators build
// the Jacobian row and the 27 sums a band later.  Parity-green; 512 threads x 2 per CU with one chunk of taps in flight per
// sampler: 162 k alignments/s against this kernel's 178 k on fp64 planes, 188 k against 186 k on fp16 planes (two chunks in
// flight need 168 registers for fp64 taps: 126-150 k at three waves per SIMD).  What the sliding-window kernel gained from
// the split -- a wave of arithmetic beside two of memory traffic -- this kernel already has: its two pipeline stages are that.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

// One tap record {I, GX, GY, pad} as the load leaves it in registers; the three values are widened where they are used.
template <typename T> struct tap_rec;
template <> struct tap_rec<double> { };                              // (fp64 / fp32 planes keep their gathers: see above)
template <> struct tap_rec<float> { };
template <> struct tap_rec<__half> {                                  // 8 bytes: one load
  u32x2 a;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int idx) { a = __builtin_amdgcn_raw_buffer_load_b64(r, idx * 8, 0, 0); }
  __device__ __forceinline__ double i() const { return (double)__half2float(__ushort_as_half((unsigned short)(a.x & 0xffffu))); }
  __device__ __forceinline__ double gx() const { return (double)__half2float(__ushort_as_half((unsigned short)(a.x >> 16))); }
  __device__ __forceinline__ double gy() const { return (double)__half2float(__ushort_as_half((unsigned short)(a.y & 0xffffu))); }
};

// EXTENSION, NOT IN THE REFERENCE'S ANALYTIC PATH (PHOVO_SAMPLING_BILINEAR): forward-additive alignment with
// bilinear sampling.  Every valid source pixel i is warped to the real-valued (tr, tc); the target intensity and
// its two gradients are sampled bilinearly there (clamp-to-edge taps; in bounds iff the nearest pixel is), the residual
// r_i = I1(tr,tc) - I0(i) and the Jacobian row J_i = gx(tr,tc)*Ju + gy(tr,tc)*Jv both belong to source pixel i.
// No scatter, hence no owner map and a single pass per iteration.  CORRECTED selects the true warp Jacobian
// (temp11 = temp15, i.e. without the reference's `+x` transcription slip, ...Analytic.h:253) instead of the
// reference's.  Huber weights and narrow storages combine with it.  The reference's only bilinear sampler lives in
// its Ceres path (third_party/sample.h:53-99, out of scope); this one uses pixel-centre integer coordinates like
// the analytic path's round().
template <int T, int WPS, typename TI, typename TD, bool CORRECTED>
__global__ __launch_bounds__(T, WPS) void gn_level_kernel_bilinear(const GNLevelArgs A)
{
  constexpr int NW = T / WAVE;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW * NRED);             // [CTL_COUNT]

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  for (;;) {                                // work queue, as in gn_level_kernel
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // see gn_level_kernel
  __syncthreads();
  const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
  if (pair >= A.n_pairs) break;
  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const unsigned char *tgt_frame = A.planes + (size_t)A.tgt[pair] * A.frame_bytes;
  const __amdgpu_buffer_rsrc_t rI0 = plane_rsrc<TI>(src_frame + A.plane_off[PLANE_I], n);
  const __amdgpu_buffer_rsrc_t rD0 = plane_rsrc<TD>(src_frame + A.plane_off[PLANE_D], n);
  constexpr bool REC = sizeof(TI) == 2;             // fp16 planes: taps come from the target's records {I, GX, GY, pad}
  const __amdgpu_buffer_rsrc_t rRec = plane_rsrc<TI>(tgt_frame + (REC ? A.rec_off : 0), REC ? 4 * n : 0);
  const __amdgpu_buffer_rsrc_t rI1 = plane_rsrc<TI>(tgt_frame + A.plane_off[PLANE_I], n);
  const __amdgpu_buffer_rsrc_t rGX = plane_rsrc<TI>(tgt_frame + A.plane_off[PLANE_GX], n);
  const __amdgpu_buffer_rsrc_t rGY = plane_rsrc<TI>(tgt_frame + A.plane_off[PLANE_GY], n);

  if (wave == 0) {
    double st[6];
#pragma unroll
    for (int j = 0; j < 6; j++) st[j] = A.states[(size_t)pair * 6 + j];
    write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], s_cst, lane);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) s_state[j] = st[j];
      s_ctl[CTL_DONE] = 0;
      s_ctl[CTL_FLAGS] = 0;
    }
  }
  __syncthreads();

  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double min_d = A.min_depth, max_d = A.max_depth;
  const double wlim = (double)W - 0.5, hlim = (double)H - 0.5;
  const double huber_delta = A.huber_delta;
  const bool huber_on = huber_delta > 0.0;
  const int k0 = wave * WAVE + lane;
  const int r0 = k0 / W, c0 = k0 - r0 * W;
  const int step_r = (NW * WAVE) / W, step_c = (NW * WAVE) - step_r * W;
  const RowColStep rc_step = make_rowcol_step(step_r, step_c, W);
  const double cd0 = (double)c0, rd0 = (double)r0;

  int iteration = 0;
  double last_gnorm = 0.0;
  int last_valid = 0;
  while (true) {
    const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
    const double r01 = uniform_f64(s_cst[C_R01]), r02 = uniform_f64(s_cst[C_R02]);
    const double r11 = uniform_f64(s_cst[C_R11]), r12 = uniform_f64(s_cst[C_R12]);
    const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
    const double t4 = uniform_f64(s_cst[C_T4]), t5 = uniform_f64(s_cst[C_T5]), t6 = uniform_f64(s_cst[C_T6]);
    const double t8 = uniform_f64(s_cst[C_T8]), t14 = uniform_f64(s_cst[C_T14]), t15 = uniform_f64(s_cst[C_T15]);
    const double t16 = uniform_f64(s_cst[C_T16]), t17 = uniform_f64(s_cst[C_T17]), t24 = uniform_f64(s_cst[C_T24]);
    const double cosy = uniform_f64(s_cst[C_CY]), siny = uniform_f64(s_cst[C_SY]);
    const double t7 = -t6, t9 = -t8, t21 = -t5;

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;

    // Two-stage software pipeline over the wave's chunks.  Without it the kernel was latency-bound (three quarters of
    // every wave's cycles parked on s_waitcnt, 2.8 TB/s at the HBM side): a chunk's twelve taps can only be requested
    // once its warp is known, and were consumed right behind the request.  Now stage `warp` of chunk i+1 (depth and
    // source intensity requested a chunk earlier; the twelve taps go out at its end) runs BEFORE stage `consume` of
    // chunk i (interpolation, Jacobian row, accumulation), so every tap has a whole chunk of arithmetic to arrive in.
    // Two register sets alternate (no copies); the arithmetic of a pixel is unchanged.
    struct Warped {
      double px, py, pz, Zr, t25, ax, ay, i0;
      tap_rec<TI> rec[REC ? 4 : 1];         // REC: the records at p00, p01, p10, p11
      double tap[REC ? 1 : 12];             // else: I1, GX, GY x (p00, p01, p10, p11)
      unsigned long long m;                 // lanes that are valid and land in bounds
    };
    int k = k0;
    double cd = cd0, rd = rd0;
    double pz_next = plane_load<TD>(rD0, k);                              // past the plane: 0
    double i0_next = plane_load<TI>(rI0, k);
    auto warp = [&](Warped &w) {
      const double pz = pz_next;
      w.i0 = i0_next;
      pz_next = plane_load<TD>(rD0, k + NW * WAVE);
      i0_next = plane_load<TI>(rI0, k + NW * WAVE);
      const double px = (cd - ox) * pz * ifx;                             // :282
      const double py = (rd - oy) * pz * ify;                             // :283
      const double X = ((t15 * px + r01 * py) + r02 * pz) + cx;           // :291
      const double Y = ((t14 * px + r11 * py) + r12 * pz) + cyy;
      const double Zr = py * t1 + pz * t2 - px * t3;
      const double t25 = fast_rcp(cz + Zr);                               // :294 and :313 are the same quantity
      const double tc = (X * fx) * t25 + ox;                              // :295
      const double tr = (Y * fy) * t25 + oy;                              // :296
      // depth gate (:280), and in bounds iff the NEAREST pixel is inside -- the same region as the reference's
      // round() test (:297-303), so a zero-motion start never sits on the boundary; in the outer half-pixel band the
      // taps are clamped to the edge row / column (NaN fails the comparisons).  One ballot per comparison, ANDed on
      // the scalar unit.
      w.m = __builtin_amdgcn_ballot_w64(k < n) & __builtin_amdgcn_ballot_w64(min_d < pz) &
            __builtin_amdgcn_ballot_w64(pz < max_d) & __builtin_amdgcn_ballot_w64(tc > -0.5) &
            __builtin_amdgcn_ballot_w64(tc < wlim) & __builtin_amdgcn_ballot_w64(tr > -0.5) &
            __builtin_amdgcn_ballot_w64(tr < hlim);
      w.px = px; w.py = py; w.pz = pz; w.Zr = Zr; w.t25 = t25;
      if (__builtin_amdgcn_inverse_ballot_w64(w.m)) {
        const double fc = floor(tc), fr = floor(tr);
        w.ax = tc - fc;
        w.ay = tr - fr;
        const int ic = (int)fc, ir = (int)fr;
        const int r0w = __mul24(max(ir, 0), W), r1w = __mul24(min(ir + 1, H - 1), W);
        if constexpr (REC) {
          // clamp-to-edge taps (in the outer half-pixel band both taps of a row / a column are the edge pixel)
          const int c0i = max(ic, 0), c1i = min(ic + 1, W - 1);
          w.rec[0].load(rRec, r0w + c0i); w.rec[1].load(rRec, r0w + c1i);
          w.rec[2].load(rRec, r1w + c0i); w.rec[3].load(rRec, r1w + c1i);
        } else
        if (sizeof(TI) < sizeof(double) && W >= 2) {                      // (compile-time and wave-uniform)
          // Narrow plane storages: the two horizontal taps of a row are neighbours in memory and go out as a PAIR -- one
          // 8-byte load for two fp32 taps, one address for two fp16 taps (fp32 planes 99 -> 176 k alignments/s, fp16
          // 110 -> 181 k at 8192 pairs per step).  Not for fp64 planes: a 16-byte gather that is only 8-byte aligned cost
          // a quarter of the rate (165 -> 126 k), twelve single loads stay.  In the outer half-pixel band both taps are the
          // edge pixel (clamp to edge): the pair is then loaded one column inside and the edge value copied over the other.
          const int cb = min(max(ic, 0), W - 2);
          const int oa = r0w + cb, ob = r1w + cb;
          plane_load2<TI>(rI1, oa, w.tap[0], w.tap[1]); plane_load2<TI>(rI1, ob, w.tap[2], w.tap[3]);
          plane_load2<TI>(rGX, oa, w.tap[4], w.tap[5]); plane_load2<TI>(rGX, ob, w.tap[6], w.tap[7]);
          plane_load2<TI>(rGY, oa, w.tap[8], w.tap[9]); plane_load2<TI>(rGY, ob, w.tap[10], w.tap[11]);
          const bool left = ic < 0, right = ic > W - 2;
          if (__builtin_amdgcn_ballot_w64(left || right)) {               // rare: a lane of the wave sits in that band
#pragma unroll
            for (int t = 0; t < 12; t += 2) {      // (selects, not conditional stores: those sent two taps through scratch)
              const double a = w.tap[t], b = w.tap[t + 1];
              w.tap[t] = right ? b : a;                                   // both taps: column W - 1
              w.tap[t + 1] = left ? a : b;                                // both taps: column 0
            }
          }
        } else {
          const int c0i = max(ic, 0), c1i = min(ic + 1, W - 1);
          const int o00 = r0w + c0i, o01 = r0w + c1i, o10 = r1w + c0i, o11 = r1w + c1i;
          w.tap[0] = plane_load<TI>(rI1, o00); w.tap[1] = plane_load<TI>(rI1, o01);
          w.tap[2] = plane_load<TI>(rI1, o10); w.tap[3] = plane_load<TI>(rI1, o11);
          w.tap[4] = plane_load<TI>(rGX, o00); w.tap[5] = plane_load<TI>(rGX, o01);
          w.tap[6] = plane_load<TI>(rGX, o10); w.tap[7] = plane_load<TI>(rGX, o11);
          w.tap[8] = plane_load<TI>(rGY, o00); w.tap[9] = plane_load<TI>(rGY, o01);
          w.tap[10] = plane_load<TI>(rGY, o10); w.tap[11] = plane_load<TI>(rGY, o11);
        }
      }
      k += NW * WAVE;
      rowcol_advance(cd, rd, rc_step);
    };
    int n_rows = 0;
    auto consume = [&](const Warped &w) {
      n_rows += __builtin_popcountll(w.m);
      if (__builtin_amdgcn_inverse_ballot_w64(w.m)) {
        const double px = w.px, py = w.py, pz = w.pz, Zr = w.Zr, t25 = w.t25, ax = w.ax, ay = w.ay;
        auto sample = [&](double p00, double p01, double p10, double p11) {
          return (1.0 - ay) * ((1.0 - ax) * p00 + ax * p01) + ay * ((1.0 - ax) * p10 + ax * p11);
        };
        double res, gxi, gyi;
        if constexpr (REC) {
          res = sample(w.rec[0].i(), w.rec[1].i(), w.rec[2].i(), w.rec[3].i()) - w.i0;
          gxi = sample(w.rec[0].gx(), w.rec[1].gx(), w.rec[2].gx(), w.rec[3].gx());
          gyi = sample(w.rec[0].gy(), w.rec[1].gy(), w.rec[2].gy(), w.rec[3].gy());
        } else {
          res = sample(w.tap[0], w.tap[1], w.tap[2], w.tap[3]) - w.i0;
          gxi = sample(w.tap[4], w.tap[5], w.tap[6], w.tap[7]);
          gyi = sample(w.tap[8], w.tap[9], w.tap[10], w.tap[11]);
        }

        const double base = pz * t4 + py * t5 + px * t15;                 // (pz*temp4+py*temp5+px*temp15) = X - x
        const double Au = CORRECTED ? base + cx : base + px * cx;         // reference: px*(temp15 + x)  (:253)
        const double Bv = py * t6 + pz * t9 + px * t14 + cyy;
        const double Cm = -py * t16 - pz * t17 - px * t24;
        const double Dm = py * t2 - pz * t1;
        double J[6];
        J[0] = (gxi * fx) * t25;
        J[1] = (gyi * fy) * t25;
        J[2] = -(J[0] * Au + J[1] * Bv) * t25;
        J[3] = J[0] * (cyy - Bv) + J[1] * base;
        J[4] = (J[0] * cosy + J[1] * siny) * Zr + Cm * J[2];
        J[5] = J[0] * (py * t4 + pz * t21) + J[1] * (pz * t7 + py * t9) + Dm * J[2];
        double wgt = 1.0;
        if (huber_on) {
          const double ar = fabs(res);
          wgt = ar <= huber_delta ? 1.0 : huber_delta / ar;
        }
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const double jw = J[a] * wgt;
#pragma unroll
          for (int b = a; b < 6; b++) {
            acc[q] = fma(jw, J[b], acc[q]);
            q++;
          }
          acc[21 + a] = fma(jw, res, acc[21 + a]);
        }
      }
    };
    {
      Warped w0, w1;
      int chunk = wave;                                                   // wave-uniform loop control throughout
      if (chunk < A.n_chunks) {
        warp(w0);
        for (;;) {
          chunk += NW;
          const bool more1 = chunk < A.n_chunks;
          if (more1) warp(w1);
          consume(w0);
          if (!more1) break;
          chunk += NW;
          const bool more0 = chunk < A.n_chunks;
          if (more0) warp(w0);
          consume(w1);
          if (!more0) break;
        }
      }
    }
    acc[RED_VALID] = lane == 0 ? (double)n_rows : 0.0;
    reduce_solve_update<NW>(acc, lane, wave, s_red, s_state, s_cst, s_ctl, A.lambda, A.max_iter, A.min_grad_norm,
                            iteration, last_gnorm, last_valid);
    iteration++;
    if (s_ctl[CTL_DONE]) break;
  }
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 6; j++) A.states[(size_t)pair * 6 + j] = s_state[j];
    if (A.reports) {
      A.reports[pair].iterations[A.level] = iteration;
      A.reports[pair].gradient_norm = last_gnorm;
      A.reports[pair].valid_pixels[A.level] = last_valid;
      A.reports[pair].flags |= (uint32_t)s_ctl[CTL_FLAGS];
    }
    s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  }
  }   // next pair
}

}  // namespace

#ifndef PHOVO_BILINEAR_WPS
#define PHOVO_BILINEAR_WPS 2
#endif
// 256-thread workgroups per CU = waves per SIMD.  The pipelined kernel keeps two chunks' worth of taps in registers
// (252 VGPRs): 2 -> 164 k / 175 k alignments/s (2048 pairs, fixed iterations, fp64 / fp16 planes), 3 -> 98 k / 87 k (84
// registers spilled into the pixel loop).  Before the pipeline: 4 -> 93 k / 126 k, 3 -> 116 k / 159 k, 2 -> 97 k / 134 k.
constexpr int BILINEAR_WPS = PHOVO_BILINEAR_WPS;

template <typename TI, typename TD>
hipError_t launch_bilinear_storage(const GNLevelArgs &a, bool corrected, int n_blocks, hipStream_t stream)
{
  const dim3 grid((unsigned)n_blocks), block(256);
  const size_t lds = lds_fixed_bytes(256);
  if (corrected) hipLaunchKernelGGL((gn_level_kernel_bilinear<256, BILINEAR_WPS, TI, TD, true>), grid, block, lds, stream, a);
  else hipLaunchKernelGGL((gn_level_kernel_bilinear<256, BILINEAR_WPS, TI, TD, false>), grid, block, lds, stream, a);
  return hipGetLastError();
}

int gn_bilinear_wgs_per_cu() { return BILINEAR_WPS; }

hipError_t gn_launch_level_bilinear(const GNLevelArgs &a, int storage, bool corrected, int cu_count,
                                    hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  const int resident = cu_count * BILINEAR_WPS;                  // persistent grid: as many workgroups as stay resident
  const int n_pairs = a.n_pairs < resident ? a.n_pairs : resident;
  switch (storage) {
    case PHOVO_STORAGE_F64: return launch_bilinear_storage<double, double>(a, corrected, n_pairs, stream);
    case PHOVO_STORAGE_F32: return launch_bilinear_storage<float, float>(a, corrected, n_pairs, stream);
    case PHOVO_STORAGE_F16: return launch_bilinear_storage<__half, float>(a, corrected, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}


}  // namespace phovo_hip
