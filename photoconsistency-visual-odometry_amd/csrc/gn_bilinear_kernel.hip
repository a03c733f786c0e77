// EXTENSION, NOT IN THE REFERENCE'S ANALYTIC PATH (PHOVO_SAMPLING_BILINEAR): forward-additive alignment with bilinear
// sampling, in two forms that compute the same thing:
//   gn_level_kernel_bilinear      taps gathered from global memory (any level size), 256 threads x 2 workgroups per CU;
//   gn_level_kernel_bilinear_lds  the target's I1 / GX / GY rows staged in LDS -- the whole level when it fits (80x60 fp64,
//                                 160x120 fp16: 115 KB), a ring of rows that slides down the image otherwise -- taps read
//                                 with ds_read2; 1024 threads, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

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
      double tap[12];                       // I1, GX, GY x (p00, p01, p10, p11)
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
        if (sizeof(TI) < sizeof(double) && W >= 2) {                      // (compile-time and wave-uniform)
          // Narrow plane storages: the two horizontal taps of a row are neighbours in memory and go out as a PAIR -- one
          // 8-byte load for two fp32 taps, one address for two fp16 taps (fp32 planes 99 -> 139 k alignments/s, fp16
          // 110 -> 144 k at 2048 pairs per step).  Not for fp64 planes: a 16-byte gather that is only 8-byte aligned cost
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
        auto sample = [&](int b) {
          const double p00 = w.tap[b], p01 = w.tap[b + 1], p10 = w.tap[b + 2], p11 = w.tap[b + 3];
          return (1.0 - ay) * ((1.0 - ax) * p00 + ax * p01) + ay * ((1.0 - ax) * p10 + ax * p11);
        };
        const double res = sample(0) - w.i0;
        const double gxi = sample(4), gyi = sample(8);

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


// Two horizontally adjacent taps of a row staged in LDS (ds_read2_b64 / ds_read2_b32 / two ds_read_u16), widened to fp64.
template <typename TI>
__device__ __forceinline__ void lds_tap2(const TI *p, double &a, double &b)
{
  a = (double)p[0];
  b = (double)p[1];
}
template <>
__device__ __forceinline__ void lds_tap2<__half>(const __half *p, double &a, double &b)
{
  a = (double)__half2float(p[0]);
  b = (double)__half2float(p[1]);
}
template <typename TI>
__device__ __forceinline__ TI plane_load_raw(__amdgpu_buffer_rsrc_t r, int idx, int soff);
template <>
__device__ __forceinline__ double plane_load_raw<double>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  return plane_load<double>(r, idx, soff);
}
template <>
__device__ __forceinline__ float plane_load_raw<float>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, idx * 4, soff, 0));
}
template <>
__device__ __forceinline__ __half plane_load_raw<__half>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  return __ushort_as_half(__builtin_amdgcn_raw_buffer_load_b16(r, idx * 2, soff, 0));
}

constexpr int BIL_PF = 4;            // elements of the row prefetch a thread carries per block (sliding ring)

// The same alignment as gn_level_kernel_bilinear with the target's three planes staged in LDS.  The twelve taps of a
// pixel have 4x spatial reuse and are re-read in every one of 20-50 iterations; gathered from global memory they kept half
// of every wave's cycles parked on the texture path (round 3: vector units 56 % busy).  Here rows of the target frame --
// [row][I1 | GX | GY][column -1 .. W], element type TI, the edge columns stored twice so that the clamp-to-edge taps of the
// outer half-pixel band need no selects -- sit in a ring of `ring_rows` rows:
//   * RESIDENT (ring_rows = H): the whole level, loaded once per pair; an iteration then reads only the source frame's
//     depth and intensity from memory (16 of the 40 algorithmic bytes per pixel);
//   * otherwise ring_rows is a power of two and the ring slides: the image is walked in blocks of T pixels (one chunk per
//     wave), block b may read rows [base(b+1), base(b) + ring_rows) from LDS, base(b) = first source row of the block minus
//     half the ring, clamped; while a block is processed every thread fetches its share of the rows block b + 1 adds --
//     into registers, written to the slots of the rows block b - 1 needed last once the chunk is done -- and ONE barrier
//     per block orders it all.  A lane whose taps fall outside that range (a motion of more than about a third of the
//     ring) takes them from global memory instead: any motion gives the same result, only the speed differs.
template <int T, typename TI, typename TD, bool CORRECTED, bool RESIDENT>
__global__ __launch_bounds__(T, T / 256) void gn_level_kernel_bilinear_lds(const GNLevelArgs A, const int ring_rows)
{
  constexpr int NW = T / WAVE;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW * NRED);             // [CTL_COUNT]
  TI *s_ring = reinterpret_cast<TI *>(lds_raw + ((lds_fixed_bytes(T) + 15) & ~(size_t)15));   // [ring_rows][3][W + 2]

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  const int slot_mask = RESIDENT ? -1 : ring_rows - 1;
  const int WP = W + 2;                           // a plane row in the ring: column -1 (= column 0), 0 .. W - 1, W (= column W - 1)
  const int row_elems = 3 * WP;                   // elements of one ring row
  // sliding ring: base(b) = clamp(first source row of block b - ring_rows / 2, 0, H - ring_rows)
  auto base_of = [&](int b) {
    const int r = (b * T) / W - ring_rows / 2;
    return min(max(r, 0), H - ring_rows);
  };
  const int n_blocks = (A.n_chunks + NW - 1) / NW;
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  for (;;) {                                // work queue, as in gn_level_kernel
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // see gn_level_kernel
  __syncthreads();
  const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
  if (pair >= A.n_pairs) break;
  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const unsigned char *tgt_frame = A.planes + (size_t)A.tgt[pair] * A.frame_bytes;
  const __amdgpu_buffer_rsrc_t rS = frame_rsrc(src_frame, A.frame_bytes), rT = frame_rsrc(tgt_frame, A.frame_bytes);
  const int oI = (int)A.plane_off[PLANE_I], oD = (int)A.plane_off[PLANE_D];
  const int oGX = (int)A.plane_off[PLANE_GX], oGY = (int)A.plane_off[PLANE_GY];
  // element e of the rows [ra, ...) a workgroup loads: its ring row, its position in that row and its value
  auto ring_element = [&](int ra, int e, int &slot_elem) {
    const int rr = e / row_elems, rem = e - rr * row_elems;
    const int p = rem / WP, cp = rem - p * WP;
    const int row = ra + rr, c = min(max(cp - 1, 0), W - 1);
    slot_elem = (row & slot_mask) * row_elems + rem;
    return plane_load_raw<TI>(rT, row * W + c, p == 0 ? oI : (p == 1 ? oGX : oGY));
  };
  auto load_rows = [&](int ra, int rb) {       // rows [ra, rb) into their ring slots (every thread takes elements tid, tid + T, ...)
    const int count = (rb - ra) * row_elems;
    for (int e = tid; e < count; e += T) {
      int at;
      const TI v = ring_element(ra, e, at);
      s_ring[at] = v;
    }
  };

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
  if (RESIDENT) load_rows(0, H);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double min_d = A.min_depth, max_d = A.max_depth;
  const double wlim = (double)W - 0.5, hlim = (double)H - 0.5;
  const double dW = (double)W;
  const double oxi = uniform_f64(-(ox + 0.5) * ifx), oyi = uniform_f64(-oy * ify);
  const double inv_w = uniform_f64(1.0 / dW);
  const double huber_delta = A.huber_delta;
  const bool huber_on = huber_delta > 0.0;

  int iteration = 0;
  double last_gnorm = 0.0;
  int last_valid = 0;
  while (true) {
    // the constants of the warp in scalar registers; those of the Jacobian are read from LDS where they are used (one
    // broadcast ds_read each, no vector instruction): all of them at once do not fit the scalar file, and a spilled one
    // costs a v_readlane per use
    const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
    const double r01 = uniform_f64(s_cst[C_R01]), r02 = uniform_f64(s_cst[C_R02]);
    const double r11 = uniform_f64(s_cst[C_R11]), r12 = uniform_f64(s_cst[C_R12]);
    const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
    const double t14 = uniform_f64(s_cst[C_T14]), t15 = uniform_f64(s_cst[C_T15]);

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;
    int n_rows = 0;

    if (!RESIDENT) {                        // the ring starts over at the top of the image
      load_rows(0, ring_rows);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
    int base = 0;                           // first row of the ring during this block
    int k = wave * WAVE + lane;
    double kd = (double)k + 0.5;
    double pz_next = plane_load<TD>(rS, k, oD);
    double i0_next = plane_load<TI>(rS, k, oI);
    for (int b = 0; b < n_blocks; b++) {    // wave-uniform loop control throughout
      const int chunk = b * NW + wave;
      // what block b + 1 adds to the ring: rows [base + ring_rows, next_base + ring_rows), into registers now
      const int next_base = RESIDENT ? 0 : base_of(b + 1);
      const int pf_count = (next_base - base) * row_elems;
      TI pf[BIL_PF];
      int pf_at[BIL_PF];
      if (!RESIDENT) {
#pragma unroll
        for (int j = 0; j < BIL_PF; j++) {
          const int e = tid + j * T;
          if (e < pf_count) pf[j] = ring_element(base + ring_rows, e, pf_at[j]);
        }
      }
      // rows this block may read from LDS (the rows below next_base are being replaced while it runs)
      const int lo = next_base, hi = RESIDENT ? H : base + ring_rows;

      if (chunk < A.n_chunks) {
        const double pz = pz_next, i0 = i0_next;
        pz_next = plane_load<TD>(rS, k + T, oD);                            // (past the plane: masked out)
        i0_next = plane_load<TI>(rS, k + T, oI);
        const double rd = trunc(kd * inv_w), cd = fma(-rd, dW, kd);         // row, column + 0.5 (see gn_level_kernel)
        const double px = fma(cd, ifx, oxi) * pz;                           // :282
        const double py = fma(rd, ify, oyi) * pz;                           // :283
        const double X = ((t15 * px + r01 * py) + r02 * pz) + cx;           // :291
        const double Y = ((t14 * px + r11 * py) + r12 * pz) + cyy;
        const double Zr = py * t1 + pz * t2 - px * t3;
        const double t25 = fast_rcp(cz + Zr);                               // :294 and :313 are the same quantity
        const double tc = (X * fx) * t25 + ox;                              // :295
        const double tr = (Y * fy) * t25 + oy;                              // :296
        // depth gate (:280), in bounds iff the NEAREST pixel is inside (gn_level_kernel_bilinear)
        const unsigned long long m =
            __builtin_amdgcn_ballot_w64(k < n) & __builtin_amdgcn_ballot_w64(min_d < pz) &
            __builtin_amdgcn_ballot_w64(pz < max_d) & __builtin_amdgcn_ballot_w64(tc > -0.5) &
            __builtin_amdgcn_ballot_w64(tc < wlim) & __builtin_amdgcn_ballot_w64(tr > -0.5) &
            __builtin_amdgcn_ballot_w64(tr < hlim);
        n_rows += __builtin_popcountll(m);
        if (__builtin_amdgcn_inverse_ballot_w64(m)) {
          const double fc = floor(tc), fr = floor(tr);
          const double ax = tc - fc, ay = tr - fr;
          const int ic = (int)fc, ir = (int)fr;                             // -1 <= ic <= W - 1: the ring's padded columns
          const int r0 = max(ir, 0), r1 = min(ir + 1, H - 1);
          const bool in_ring = RESIDENT || (r0 >= lo && r1 < hi);
          const unsigned long long outside = RESIDENT ? 0ull : __builtin_amdgcn_ballot_w64(!in_ring);
          const TI *p0 = s_ring + (r0 & slot_mask) * row_elems + (ic + 1), *p1 = s_ring + (r1 & slot_mask) * row_elems + (ic + 1);
          // one plane at a time (four taps live, not twelve)
          auto sample = [&](const int lds_off, const int plane_off) {
            double p00, p01, p10, p11;
            if (in_ring) {
              lds_tap2<TI>(p0 + lds_off, p00, p01);
              lds_tap2<TI>(p1 + lds_off, p10, p11);
            }
            if (!RESIDENT && outside) {                                     // rare: a motion larger than the ring covers
              if (!in_ring) {
                const int c0 = max(ic, 0), c1 = min(ic + 1, W - 1);
                p00 = plane_load<TI>(rT, r0 * W + c0, plane_off); p01 = plane_load<TI>(rT, r0 * W + c1, plane_off);
                p10 = plane_load<TI>(rT, r1 * W + c0, plane_off); p11 = plane_load<TI>(rT, r1 * W + c1, plane_off);
              }
            }
            const double top = fma(ax, p01 - p00, p00), bot = fma(ax, p11 - p10, p10);
            return fma(ay, bot - top, top);
          };
          const double res = sample(0, oI) - i0;
          const double gxi = sample(WP, oGX), gyi = sample(2 * WP, oGY);

          // (volatile: read HERE, every chunk -- hoisted out of the loop they would sit in eighteen vector registers)
          typedef const volatile __attribute__((address_space(3))) double lds_cst_t;
          lds_cst_t *const vc = (lds_cst_t *)s_cst;
          const double t4 = vc[C_T4], t5 = vc[C_T5], t6 = vc[C_T6], t8 = vc[C_T8];
          const double t16 = vc[C_T16], t17 = vc[C_T17], t24 = vc[C_T24];
          const double cosy = vc[C_CY], siny = vc[C_SY];
          const double bs = pz * t4 + py * t5 + px * t15;                   // (pz*temp4+py*temp5+px*temp15) = X - x
          const double Au = CORRECTED ? bs + cx : bs + px * cx;             // reference: px*(temp15 + x)  (:253)
          const double Bv = py * t6 - pz * t8 + px * t14 + cyy;             // (temp9 = -temp8)
          const double Cm = -py * t16 - pz * t17 - px * t24;
          const double Dm = py * t2 - pz * t1;
          double J[6];
          J[0] = (gxi * fx) * t25;
          J[1] = (gyi * fy) * t25;
          J[2] = -(J[0] * Au + J[1] * Bv) * t25;
          J[3] = J[0] * (cyy - Bv) + J[1] * bs;
          J[4] = (J[0] * cosy + J[1] * siny) * Zr + Cm * J[2];
          J[5] = J[0] * (py * t4 - pz * t5) - J[1] * (pz * t6 + py * t8) + Dm * J[2];      // (temp21 = -temp5, temp7 = -temp6, temp9 = -temp8)
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
            for (int c = a; c < 6; c++) {
              acc[q] = fma(jw, J[c], acc[q]);
              q++;
            }
            acc[21 + a] = fma(jw, res, acc[21 + a]);
          }
        }
      }
      k += T;
      kd += (double)T;
      if (!RESIDENT) {
        // the prefetched rows go into the slots of rows [base, next_base): nobody reads those during this block (lo), and
        // the barrier below separates these writes from the next block's reads
#pragma unroll
        for (int j = 0; j < BIL_PF; j++) {
          const int e = tid + j * T;
          if (e < pf_count) s_ring[pf_at[j]] = pf[j];
        }
        base = next_base;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
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


// ---- the LDS-staged form -------------------------------------------------------------------------------------------
namespace {

constexpr int BIL_LDS_THREADS = 768;

size_t bilinear_ring_offset() { return (lds_fixed_bytes(BIL_LDS_THREADS) + 15) & ~(size_t)15; }

template <typename TI, typename TD>
hipError_t prepare_bilinear_lds()
{
  hipError_t e = hipSuccess;
#define PHOVO_PREP(C, R)                                                                                                       \
  if (e == hipSuccess)                                                                                                         \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gn_level_kernel_bilinear_lds<BIL_LDS_THREADS, TI, TD, C, R>),     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
  PHOVO_PREP(true, true) PHOVO_PREP(true, false) PHOVO_PREP(false, true) PHOVO_PREP(false, false)
#undef PHOVO_PREP
  return e;
}

template <typename TI, typename TD>
hipError_t launch_bilinear_lds(const GNLevelArgs &a, bool corrected, int ring_rows, int n_blocks, hipStream_t stream)
{
  const dim3 grid((unsigned)n_blocks), block(BIL_LDS_THREADS);
  const size_t lds = bilinear_ring_offset() + (size_t)ring_rows * 3 * (size_t)(a.w + 2) * sizeof(TI);
  const bool resident = ring_rows >= a.h;
#define PHOVO_GO(C, R) hipLaunchKernelGGL((gn_level_kernel_bilinear_lds<BIL_LDS_THREADS, TI, TD, C, R>), grid, block, lds, stream, a, ring_rows)
  if (corrected) { if (resident) PHOVO_GO(true, true); else PHOVO_GO(true, false); }
  else { if (resident) PHOVO_GO(false, true); else PHOVO_GO(false, false); }
#undef PHOVO_GO
  return hipGetLastError();
}

}  // namespace

int gn_bilinear_ring_rows(int w, int h, int storage)
{
  const size_t elem = storage_elem_size(storage, false);
  const size_t row = 3 * (size_t)(w + 2) * elem, room = LDS_LIMIT - bilinear_ring_offset();
  if (w < 2 || row == 0) return 0;
  if ((size_t)h * row <= room) return h;                               // the whole level stays in LDS
  int r = 1;
  while ((size_t)(2 * r) * row <= room) r *= 2;                        // a sliding ring: a power of two rows
  // a block of 1024 pixels spans 1024 / w + 2 rows and the ring moves by up to 1024 / w + 1 per block: it must hold both
  // with rows to spare either side, and a thread's share of a block's new rows must fit its prefetch registers
  const int span = BIL_LDS_THREADS / w + 2, advance = BIL_LDS_THREADS / w + 1;
  if (r < 2 * span + 4) return 0;
  if ((size_t)advance * 3 * (size_t)(w + 2) > (size_t)BIL_PF * BIL_LDS_THREADS) return 0;
  return r;
}

hipError_t gn_prepare_bilinear_kernels()
{
  hipError_t e;
  if ((e = prepare_bilinear_lds<double, double>()) != hipSuccess) return e;
  if ((e = prepare_bilinear_lds<float, float>()) != hipSuccess) return e;
  return prepare_bilinear_lds<__half, float>();
}

hipError_t gn_launch_level_bilinear_lds(const GNLevelArgs &a, int storage, bool corrected, int cu_count, hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  const int ring_rows = gn_bilinear_ring_rows(a.w, a.h, storage);
  if (ring_rows <= 0) return hipErrorInvalidValue;
  const int n_blocks = a.n_pairs < cu_count ? a.n_pairs : cu_count;      // persistent grid, one workgroup per CU
  switch (storage) {
    case PHOVO_STORAGE_F64: return launch_bilinear_lds<double, double>(a, corrected, ring_rows, n_blocks, stream);
    case PHOVO_STORAGE_F32: return launch_bilinear_lds<float, float>(a, corrected, ring_rows, n_blocks, stream);
    case PHOVO_STORAGE_F16: return launch_bilinear_lds<__half, float>(a, corrected, ring_rows, n_blocks, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace phovo_hip
