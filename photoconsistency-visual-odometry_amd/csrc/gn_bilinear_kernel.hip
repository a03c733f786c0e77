// EXTENSION, NOT IN THE REFERENCE'S ANALYTIC PATH (PHOVO_SAMPLING_BILINEAR): forward-additive alignment with bilinear
// sampling (any level size, no owner map, one pass per iteration).  Two kernels, chosen by the plane storage:
//   * fp64 / fp32 planes: gn_level_kernel_bilinear_dma -- a chunk's twelve taps are DMA-ed straight into LDS
//     (buffer_load ... lds) while the chunk before is consumed; 168 registers, three waves per SIMD (round 5);
//   * fp16 planes: gn_level_kernel_bilinear -- a tap is one 8-byte RECORD {I, GX, GY, pad} of the target frame
//     (pyr_build_tap_records) held in registers; two waves per SIMD (round 5).
// What the numbers were (8192 pairs, fixed 50 + 20 iterations, alignments/s; profiles/r05_runs/bilinear_*.txt):
//   rounds 2-4, taps gathered plane by plane into registers (12 loads per pixel, 252 registers, 2 waves per SIMD):
//     fp64 178 k, fp32 195 k, fp16 186-196 k; vector units 55 % busy, half of all wave cycles parked;
//   round 4, measured and not kept: target planes staged in LDS by the workgroup (whole level: same speed; sliding ring of
//     rows: 2.5 x slower), sampler / accumulator waves (162 k fp64): profiles/r04_runs/bilinear_{lds,roles}_ab.txt;
//   round 5, tap records: fp16 +14 % (224 k: 4 loads per pixel instead of 12), fp32 -5 %, fp64 +-0 (8 loads instead of 12
//     changed nothing: it was never the NUMBER of gathers) -> kept for fp16;
//   round 5, taps through LDS-DMA: fp32 223 k (+14 %), fp64 119 k with 24 dword loads per chunk (a CU has one address
//     path: twice the instructions cost a third of the rate) and 187-193 k (+5...8 %) with the two taps of a row as ONE
//     16-byte load (6 per chunk) -> kept for fp64 and fp32;
//   round 5, that form's vector instructions (93 % busy at three waves per SIMD, a quarter of them no arithmetic): loop-only
//     pose constants read from LDS where they are used instead of 30 v_readlane per chunk of spilled scalar registers, the
//     outer half-pixel band as a weight of 0 or 1 instead of 24 v_cndmask per chunk: fp64 195 -> 215-218 k (+11-12 %),
//     fp32 +-0 (bilinear_valu_ab.txt).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

// One tap record {I, GX, GY, pad} as the load leaves it in registers; the three values are widened where they are used.
template <typename T> struct tap_rec;
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
  static_assert(sizeof(TI) == 2, "the record form serves fp16 planes (fp64 / fp32: gn_level_kernel_bilinear_dma)");
  const __amdgpu_buffer_rsrc_t rRec = plane_rsrc<TI>(tgt_frame + A.rec_off, 4 * n);      // the target's tap records {I, GX, GY, pad}

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
      tap_rec<TI> rec[4];                   // the records at p00, p01, p10, p11
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
        // clamp-to-edge taps (in the outer half-pixel band both taps of a row / a column are the edge pixel)
        const int c0i = max(ic, 0), c1i = min(ic + 1, W - 1);
        w.rec[0].load(rRec, r0w + c0i); w.rec[1].load(rRec, r0w + c1i);
        w.rec[2].load(rRec, r1w + c0i); w.rec[3].load(rRec, r1w + c1i);
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
        const double res = sample(w.rec[0].i(), w.rec[1].i(), w.rec[2].i(), w.rec[3].i()) - w.i0;
        const double gxi = sample(w.rec[0].gx(), w.rec[1].gx(), w.rec[2].gx(), w.rec[3].gx());
        const double gyi = sample(w.rec[0].gy(), w.rec[1].gy(), w.rec[2].gy(), w.rec[3].gy());

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


// A wave-uniform constant read out of LDS where it is used (volatile: never hoisted into registers that the loop then carries).
typedef const volatile __attribute__((address_space(3))) double *lds_cst_ptr;
__device__ __forceinline__ lds_cst_ptr lds_cst(const double *p) { return (lds_cst_ptr)p; }

// ---------------------------------------------------------------------------------------------------------------------------
// The same alignment with the taps landing in LDS instead of registers (fp64 / fp32 planes; round 5).  A form that keeps two
// chunks' worth of taps in registers needs 252 of them, i.e. two waves per SIMD, and with two waves a SIMD has nothing to
// issue while both wait for their taps (rounds 2-4: vector units 55 % busy, half of all wave cycles parked; fewer loads per tap
// changed nothing: profiles/r05_runs/bilinear_records_ab.txt).  gfx950's buffer loads can write straight to LDS
// (buffer_load_dword ... lds: lane l's dword lands at M0 + 4 l), so here a chunk's twelve taps are DMA-ed into a per-wave LDS
// slot while the chunk before is consumed, and read back -- each lane its own words, no barrier -- when their turn comes:
// the taps cost registers only while they are being interpolated, the kernel fits 168 registers and a third wave per SIMD.
// Same per-pixel arithmetic as the gather form, with one exception on fp64 planes: in the outer half-pixel band a row's
// value is 1 * e + 0 * x instead of (1 - ax) * e + ax * e (e the edge pixel; see warp()), equal up to that sum's rounding.
template <int T, int WPS, typename TI, typename TD, bool CORRECTED>
__global__ __launch_bounds__(T, WPS) void gn_level_kernel_bilinear_dma(const GNLevelArgs A)
{
  static_assert(sizeof(TI) == 8 || sizeof(TI) == 4, "fp64 or fp32 planes (fp16 planes take the record form)");
  constexpr int NW = T / WAVE;
  constexpr bool PAIRS = sizeof(TI) == 8;                                // fp64: a row's two taps are one 16-byte load
  constexpr int SLOT_BYTES = 12 * (int)sizeof(TI) * WAVE;                // a chunk's taps: I1, GX, GY x (p00, p01, p10, p11), all 64 lanes
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW * NRED);             // [CTL_COUNT]
  unsigned char *s_taps = lds_raw + ((lds_fixed_bytes(T) + 15) & ~(size_t)15);      // [NW][2][SLOT_BYTES]

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  unsigned char *const my_slots = s_taps + (size_t)wave * 2 * SLOT_BYTES;
  // (bounds the warp compares against: kept behind the pose constants in LDS, read back per chunk -- see consume())
  enum { C_MIN_D = 24, C_MAX_D, C_WLIM, C_HLIM };
  static_assert(C_MIN_D >= C_COUNT && C_HLIM < 32, "spare slots of the constant block");
  if (tid == 0) {
    s_cst[C_MIN_D] = A.min_depth; s_cst[C_MAX_D] = A.max_depth;
    s_cst[C_WLIM] = (double)A.w - 0.5; s_cst[C_HLIM] = (double)A.h - 0.5;
    s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  }
  for (;;) {                                // work queue, as in gn_level_kernel
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // see gn_level_kernel
  __syncthreads();
  const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
  if (pair >= A.n_pairs) break;
  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const unsigned char *tgt_frame = A.planes + (size_t)A.tgt[pair] * A.frame_bytes;
  const __amdgpu_buffer_rsrc_t rI0 = plane_rsrc<TI>(src_frame + A.plane_off[PLANE_I], n);
  const __amdgpu_buffer_rsrc_t rD0 = plane_rsrc<TD>(src_frame + A.plane_off[PLANE_D], n);
  // one descriptor for the target frame; a plane is chosen by the load's scalar offset
  const __amdgpu_buffer_rsrc_t rT = frame_rsrc(tgt_frame, A.frame_bytes);
  const int o_plane[3] = {(int)A.plane_off[PLANE_I], (int)A.plane_off[PLANE_GX], (int)A.plane_off[PLANE_GY]};

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
    const double t14 = uniform_f64(s_cst[C_T14]), t15 = uniform_f64(s_cst[C_T15]);

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;

    struct Warped {
      double px, py, pz, Zr, t25, ax, ay, i0;
      int off[4];                           // byte offsets in a plane: fp32 the taps p00, p01, p10, p11; fp64 [0], [1]: the two PAIRS
      unsigned long long m;                 // lanes that are valid and land in bounds
    };
    int k = k0;
    double cd = cd0, rd = rd0;
    double pz_next = plane_load<TD>(rD0, k);                              // past the plane: 0
    double i0_next = plane_load<TI>(rI0, k);
    // (the depth and source intensity of the chunk after the one warp() has just worked on: requested by prefetch(), which the
    // loop calls right IN FRONT of a batch of LDS-bound loads, never behind one -- see the wait in the loop)
    auto prefetch = [&]() {
      pz_next = plane_load<TD>(rD0, k);
      i0_next = plane_load<TI>(rI0, k);
    };
    auto warp = [&](Warped &w) {            // geometry of the wave's next chunk; the taps are requested by issue()
      const double pz = pz_next;
      w.i0 = i0_next;
      const double min_d = lds_cst(s_cst)[C_MIN_D], max_d = lds_cst(s_cst)[C_MAX_D];
      const double wlim = lds_cst(s_cst)[C_WLIM], hlim = lds_cst(s_cst)[C_HLIM];
      const double px = (cd - ox) * pz * ifx;                             // :282
      const double py = (rd - oy) * pz * ify;                             // :283
      const double X = ((t15 * px + r01 * py) + r02 * pz) + cx;           // :291
      const double Y = ((t14 * px + r11 * py) + r12 * pz) + cyy;
      const double Zr = py * t1 + pz * t2 - px * t3;
      const double t25 = fast_rcp(cz + Zr);                               // :294 and :313 are the same quantity
      const double tc = (X * fx) * t25 + ox;                              // :295
      const double tr = (Y * fy) * t25 + oy;                              // :296
      w.m = __builtin_amdgcn_ballot_w64(k < n) & __builtin_amdgcn_ballot_w64(min_d < pz) &
            __builtin_amdgcn_ballot_w64(pz < max_d) & __builtin_amdgcn_ballot_w64(tc > -0.5) &
            __builtin_amdgcn_ballot_w64(tc < wlim) & __builtin_amdgcn_ballot_w64(tr > -0.5) &
            __builtin_amdgcn_ballot_w64(tr < hlim);
      w.px = px; w.py = py; w.pz = pz; w.Zr = Zr; w.t25 = t25;
      const double fc = floor(tc), fr = floor(tr);
      w.ay = tr - fr;
      // (lanes outside w.m: whatever the conversions give; their taps are range-checked reads nobody uses)
      const int ic = (int)fc, ir = (int)fr;
      const int r0w = __mul24(min(max(ir, 0), H - 1), W), r1w = __mul24(min(max(ir + 1, 0), H - 1), W);
      if (PAIRS) {
        // fp64 planes: the two horizontal taps of a row are neighbours in memory and travel as ONE 16-byte load (24 dword loads
        // per chunk made this form a third SLOWER than the gathers: a CU has one address path).  In the outer half-pixel band
        // both taps are the edge pixel e: the pair is loaded one column inside and the horizontal weight is set to 0 (left
        // band: the pair is (e, x)) or 1 (right band: (x, e)), so that the row gives 1 * e + 0 * x.  That is the value of
        // (1 - ax) * e + ax * e up to its rounding, i.e. within one ulp of the clamped-tap form the other storages and the oracle
        // use, in that band only; selecting the taps instead cost 24 v_cndmask per chunk in a kernel whose vector units are
        // 93 % busy (profiles/r05_bilinear_pmc_sq.json).  x is a pixel of the plane, finite whenever the plane is.
        const int cb = min(max(ic, 0), max(W - 2, 0));
        w.off[0] = (r0w + cb) * 8; w.off[1] = (r1w + cb) * 8; w.off[2] = w.off[3] = 0;
        w.ax = (ic < 0 || W < 2) ? 0.0 : (ic > W - 2 ? 1.0 : tc - fc);       // (a one-column image: both taps are that column)
      } else {
        const int c0i = min(max(ic, 0), W - 1), c1i = min(max(ic + 1, 0), W - 1);      // clamp-to-edge taps
        w.off[0] = (r0w + c0i) * (int)sizeof(TI); w.off[1] = (r0w + c1i) * (int)sizeof(TI);
        w.off[2] = (r1w + c0i) * (int)sizeof(TI); w.off[3] = (r1w + c1i) * (int)sizeof(TI);
        w.ax = tc - fc;
      }
      k += NW * WAVE;
      rowcol_advance(cd, rd, rc_step);
    };
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto issue = [&](const Warped &w, int slot) {                         // the chunk's twelve taps, straight to LDS
      unsigned char *base = my_slots + slot * SLOT_BYTES;
#pragma unroll
      for (int p = 0; p < 3; p++) {
        if (PAIRS) {                         // lane l's 16 bytes (two doubles) land at block + 16 l; (never the instruction's own
#pragma unroll                               //  offset field: it moves the LDS address along, tools/probes/lds_dma_probe.hip)
          for (int r = 0; r < 2; r++) {
            // (the 16-byte form exists on gfx950 only, and the HOST pass of this file -- which needs the kernel's body for
            // nothing but its launch stub -- checks the size against a generic target, rejects it and silently drops the stub)
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (lds_ptr)(base + (p * 2 + r) * 1024), 16, w.off[r], o_plane[p], 0, 0);
#endif
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (lds_ptr)(base + (p * 4 + t) * 256), 4, w.off[t], o_plane[p], 0, 0);
        }
      }
    };
    int n_rows = 0;
    // A chunk is consumed in two steps with the NEXT chunk's loads issued in between: first its taps come out of LDS into
    // registers (the wait in front of these reads is for its own loads, issued a chunk ago), then the next chunk's twelve
    // loads go out, then the arithmetic runs while those travel.  (Issued in front of the reads, the new loads would be what
    // the reads wait for: the hardware counts a wave's loads in order.)
    auto read_taps = [&](int slot, const Warped &w, double (&smp)[3]) {      // -> the three bilinear samples I1, GX, GY
      // this lane's words of the slot (written by this lane's loads: no other wave, no other lane touches them)
      const unsigned char *base = my_slots + slot * SLOT_BYTES;
      const unsigned *words = reinterpret_cast<const unsigned *>(base) + lane;
      const double ax = w.ax, ay = w.ay;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double p00, p01, p10, p11;
        if (PAIRS) {
          const double *row0 = reinterpret_cast<const double *>(base + (c * 2) * 1024) + 2 * lane;
          const double *row1 = reinterpret_cast<const double *>(base + (c * 2 + 1) * 1024) + 2 * lane;
          p00 = row0[0]; p01 = row0[1]; p10 = row1[0]; p11 = row1[1];   // (outer half-pixel band: see the weight in warp())
        } else {
          p00 = (double)__uint_as_float(words[(4 * c) * 64]); p01 = (double)__uint_as_float(words[(4 * c + 1) * 64]);
          p10 = (double)__uint_as_float(words[(4 * c + 2) * 64]); p11 = (double)__uint_as_float(words[(4 * c + 3) * 64]);
        }
        smp[c] = (1.0 - ay) * ((1.0 - ax) * p00 + ax * p01) + ay * ((1.0 - ax) * p10 + ax * p11);
      }
    };
    auto consume = [&](const Warped &w, const double (&smp)[3]) {
      n_rows += __builtin_popcountll(w.m);
      if (__builtin_amdgcn_inverse_ballot_w64(w.m)) {
        const double px = w.px, py = w.py, pz = w.pz, Zr = w.Zr, t25 = w.t25;
        const double res = smp[0] - w.i0;
        const double gxi = smp[1], gyi = smp[2];
        // The pose constants only this block uses come out of LDS each time (a broadcast read, the LDS pipe has room), not out
        // of scalar registers: with them the kernel wanted ~30 more scalar registers than a wave has, and what the compiler
        // then parks in lanes of a vector register comes back through v_readlane -- 30 of them per chunk, on vector units
        // that are the kernel's bound.
        const lds_cst_ptr vc = lds_cst(s_cst);
        const double t4 = vc[C_T4], t5 = vc[C_T5], t6 = vc[C_T6], t8 = vc[C_T8];
        const double t16 = vc[C_T16], t17 = vc[C_T17], t24 = vc[C_T24], cosy = vc[C_CY], siny = vc[C_SY];
        const double t7 = -t6, t9 = -t8, t21 = -t5;

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
      double smp[3];
      int chunk = wave;                                                   // wave-uniform loop control throughout
      if (chunk < A.n_chunks) {
        warp(w0);
        prefetch();
        issue(w0, 0);
        for (;;) {
          chunk += NW;
          const bool more1 = chunk < A.n_chunks;
          if (more1) warp(w1);
          // The taps about to be read were requested a chunk ago, and NOTHING has been requested since.  That is on purpose,
          // and the wait is for everything: on gfx950 a register-bound load issued BEHIND LDS-bound ones may retire before them
          // (tools/probes/lds_dma_probe.hip, variant 3: `s_waitcnt vmcnt(2)` with two younger register loads outstanding left
          // the last two LDS words unwritten in 98 % of all trials), so a count that lets the youngest loads stay out is
          // only sound when those are LDS-bound too (variant 5) or the register-bound ones are the OLDER (variant 4: that is
          // how warp() gets its depth -- prefetch() runs in front of issue()).  The wait is explicit: what the compiler infers
          // about a ds_read behind such loads is not something to rest a result on.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          read_taps(0, w0, smp);
          if (more1) { prefetch(); issue(w1, 1); }
          consume(w0, smp);
          if (!more1) break;
          chunk += NW;
          const bool more0 = chunk < A.n_chunks;
          if (more0) warp(w0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          read_taps(1, w1, smp);
          if (more0) { prefetch(); issue(w0, 0); }
          consume(w1, smp);
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

// The record form (fp16 planes): 256-thread workgroups per CU = waves per SIMD.  It keeps two chunks' worth of records in
// registers (228): 2 workgroups per CU; 3 spill 32 registers into the pixel loop.
constexpr int BILINEAR_WPS = 2;
// The LDS-landing form (fp64 / fp32 planes): 256 threads x 3 workgroups per CU = 3 waves per SIMD, two tap slots per wave
// (3 x 50.5 KB of the CU's 160 KB of LDS).
constexpr int BILINEAR_DMA_WPS = 3;
template <typename TI>
constexpr size_t bilinear_dma_lds_bytes()
{
  return ((lds_fixed_bytes(256) + 15) & ~(size_t)15) + (size_t)(256 / WAVE) * 2 * 12 * sizeof(TI) * WAVE;
}

template <typename TI, typename TD>
hipError_t launch_bilinear_dma(const GNLevelArgs &a, bool corrected, int cu_count, hipStream_t stream)
{
  const int resident = cu_count * BILINEAR_DMA_WPS;
  const dim3 grid((unsigned)(a.n_pairs < resident ? a.n_pairs : resident)), block(256);
  const size_t lds = bilinear_dma_lds_bytes<TI>();
  if (corrected) hipLaunchKernelGGL((gn_level_kernel_bilinear_dma<256, BILINEAR_DMA_WPS, TI, TD, true>), grid, block, lds, stream, a);
  else hipLaunchKernelGGL((gn_level_kernel_bilinear_dma<256, BILINEAR_DMA_WPS, TI, TD, false>), grid, block, lds, stream, a);
  return hipGetLastError();
}

int gn_bilinear_wgs_per_cu(int storage) { return storage == PHOVO_STORAGE_F16 ? BILINEAR_WPS : BILINEAR_DMA_WPS; }

hipError_t gn_launch_level_bilinear(const GNLevelArgs &a, int storage, bool corrected, int cu_count,
                                    hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  switch (storage) {
    case PHOVO_STORAGE_F64: return launch_bilinear_dma<double, double>(a, corrected, cu_count, stream);
    case PHOVO_STORAGE_F32: return launch_bilinear_dma<float, float>(a, corrected, cu_count, stream);
    case PHOVO_STORAGE_F16: {
      if (!a.rec_off) return hipErrorInvalidValue;             // (the pool carries tap records exactly in this mode: engine.cpp)
      const int resident = cu_count * BILINEAR_WPS;                // persistent grid: as many workgroups as stay resident
      const dim3 grid((unsigned)(a.n_pairs < resident ? a.n_pairs : resident)), block(256);
      const size_t lds = lds_fixed_bytes(256);
      if (corrected) hipLaunchKernelGGL((gn_level_kernel_bilinear<256, BILINEAR_WPS, __half, float, true>), grid, block, lds, stream, a);
      else hipLaunchKernelGGL((gn_level_kernel_bilinear<256, BILINEAR_WPS, __half, float, false>), grid, block, lds, stream, a);
      return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
  }
}


}  // namespace phovo_hip
