// Gauss-Newton level kernel for levels whose owner map does not fit LDS (more than ~39 k pixels: 320x240 of a
// 1280x960 pyramid, 640x480 level 0): the SLIDING-WINDOW form.
//
// Same per-pixel arithmetic and the same reference semantics as gn_level_kernel (gn_kernels.hip; reference:
// phovo/include/CPhotoconsistencyOdometryAnalytic.h:191-367, 376-392, 500-563), one workgroup per frame pair, whole
// iteration loop of the level on the device.  What differs is where the scatter of the residuals (:358, last raster
// writer wins) is resolved.  gn_level_kernel's HUGE variant keeps the owner map in HBM (tagged global atomics, 60 B per
// pixel-iteration instead of 40, 0.51 of the roofline).  Here the map is a RING in LDS that slides down the image, and
// the two passes of an iteration run AT THE SAME TIME in different waves of the workgroup:
//
//   * the image is cut into bands of BAND_PX pixels; an iteration is a sequence of phases with one workgroup barrier each;
//   * the first NW1 waves of the workgroup only ever run PASS 1 (warp, atomicMax into the ring): in phase s each of them
//     takes B1 chunks of source band s;
//   * the other NW2 waves only ever run PASS 2 (residual, Jacobian row, 27 sums): in phase s each of them takes B2 chunks
//     of target band s - m - 1, whose owners -- and the source intensities they point at, and its four planes -- it
//     requested a phase earlier, when band s - m had just become final;
//   * the ring holds 32768 int32 (128 KiB): while band s is warped, targets may fall into bands s-m+1 .. s+m.  Rotations
//     and translations of the sizes Gauss-Newton steps take move a pixel by a few rows; m is chosen per level by the
//     host (GNLevelArgs::slide_m), at most M_MAX;
//   * a source pixel whose target falls OUTSIDE the window sets a flag.  The iteration is then void: the state is left
//     as it was, the pair is put on GNLevelArgs::handover_out and the engine's follow-up launch of gn_level_kernel (HBM
//     owner map, exact for any motion) continues that pair from the same iteration.  Results are therefore exactly
//     the reference's whatever the motion; only the speed depends on the window.
//
// Why two kinds of waves.  Rounds 2-3 ran both passes interleaved in every wave: 27 sums, both passes' operands a phase
// ahead and all 32 pose constants live at once need 256 registers, i.e. 2 waves per SIMD, and pass 1 (arithmetic and LDS
// atomics, 8 bytes per pixel) and pass 2 (40 bytes per pixel) then overlap only inside one wave's instruction stream:
// vector units 69-73 % busy, 0.63-0.68 of the roofline.  With one kind of work per wave each fits 168 registers (pass 2:
// the sums and a phase of prefetched operands; pass 1: almost nothing), each keeps only its own constants in scalar
// registers, a SIMD holds one arithmetic-bound and two memory-bound waves, and the hardware interleaves them.
// Depth is read by both passes (m + 1 bands apart: the second read is an L2 / Infinity Cache hit); no global atomics, no
// owner traffic in HBM.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

constexpr int SLIDE_RING_PX = 32768;                          // entries of the ring, a power of two (128 KiB)
constexpr int SLIDE_MASK_RING = 512;                          // in-bounds ballots of the chunks between their two passes (4 KiB)
static_assert((SLIDE_RING_PX & (SLIDE_RING_PX - 1)) == 0 && (SLIDE_MASK_RING & (SLIDE_MASK_RING - 1)) == 0, "ring indices are masks");

// Geometry of one instantiation: T threads of which NW1 waves run pass 1 on B1 chunks per phase and the others pass 2 on
// B2 chunks per phase.
template <int T_, int NW1_, int B1_, int B2_>
struct SlideGeom {
  static constexpr int T = T_, NW = T_ / WAVE, NW1 = NW1_, NW2 = NW - NW1_, B1 = B1_, B2 = B2_;
  static constexpr int BAND_CHUNKS = NW1 * B1;
  static constexpr int BAND_PX = BAND_CHUNKS * WAVE;
  static constexpr int RING_BANDS = SLIDE_RING_PX / BAND_PX;
  // Source band s may write target bands s-m+1 .. s+m; band s-m is final when phase s begins (pass 2 reads and resets
  // its owners then, a phase ahead of their use): 2m + 1 ring slots.
  static constexpr int M_MAX = (RING_BANDS - 1) / 2;
  static_assert(NW1 * B1 == NW2 * B2, "both kinds of waves cover one band per phase");
  static_assert(NW2 % 2 == 0, "the cross-wave sum splits the rows of s_red in two halves");
  static_assert((M_MAX + 2) * BAND_CHUNKS <= SLIDE_MASK_RING, "ballots of the bands between pass 1 and pass 2");
};

template <int T, int NW1, int B1, int B2, typename TI, typename TD>
__global__ __launch_bounds__(T, T / 256) void gn_level_kernel_slide(const GNLevelArgs A)
{
  using G = SlideGeom<T, NW1, B1, B2>;
  constexpr int NW2 = G::NW2, BAND_CHUNKS = G::BAND_CHUNKS, BAND_PX = G::BAND_PX;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW2][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW2 * NRED);            // [CTL_COUNT]
  unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(s_ctl + CTL_COUNT);      // [SLIDE_MASK_RING]
  int *s_owner = reinterpret_cast<int *>(s_mask + SLIDE_MASK_RING);    // [SLIDE_RING_PX]

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  const int n_bands = (A.n_chunks + BAND_CHUNKS - 1) / BAND_CHUNKS;
  const int m_run = A.slide_m;                                         // 1 .. M_MAX (gn_launch_level_slide)
  const int n_phases = n_bands + m_run + 1;
  const int oI = (int)A.plane_off[PLANE_I], oD = (int)A.plane_off[PLANE_D];
  const int oGX = (int)A.plane_off[PLANE_GX], oGY = (int)A.plane_off[PLANE_GY];
  // Work queue and loop shape exactly as in gn_level_kernel (one exit every wave reaches; the next ticket is drawn in the
  // block that writes the finished pair back; explicit LDS wait in front of the barrier at the loop head).
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  for (;;) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
  if (pair >= A.n_pairs) break;

  // one descriptor per frame, the plane chosen by a scalar offset (gn_device.hpp, plane_load)
  const __amdgpu_buffer_rsrc_t rS = frame_rsrc(A.planes + (size_t)A.src[pair] * A.frame_bytes, A.frame_bytes);
  const __amdgpu_buffer_rsrc_t rT = frame_rsrc(A.planes + (size_t)A.tgt[pair] * A.frame_bytes, A.frame_bytes);

  // ---- pair prologue: empty ring, pose constants --------------------------------------------------------
  for (int k = tid; k < SLIDE_RING_PX; k += T) s_owner[k] = -1;
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
      s_ctl[CTL_OOW] = 0;
    }
  }
  __syncthreads();

  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double dW = (double)W;
  // (row, column) from the linear index carried as the double k + 0.5, and the half pixel folded into the unprojection:
  // gn_kernels.hip, level_body
  const double oxi = uniform_f64(-(ox + 0.5) * ifx), oyi = uniform_f64(-oy * ify);
  const double inv_w = uniform_f64(1.0 / dW);

  int iteration = 0;
  double last_gnorm = 0.0;
  int last_valid = 0;
  bool handed_over = false;
  while (true) {
    if (wave < NW1) {
      // ================= the waves of PASS 1: warp, bounds, window, atomicMax into the ring  (:279-303, 358) ==========
      const double min_d = A.min_depth, max_d = A.max_depth;
      const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
      const double fr00 = uniform_f64(uniform_f64(s_cst[C_T15]) * fx), fr01 = uniform_f64(uniform_f64(s_cst[C_R01]) * fx);
      const double fr02 = uniform_f64(uniform_f64(s_cst[C_R02]) * fx);
      const double fr10 = uniform_f64(uniform_f64(s_cst[C_T14]) * fy), fr11 = uniform_f64(uniform_f64(s_cst[C_R11]) * fy);
      const double fr12 = uniform_f64(uniform_f64(s_cst[C_R12]) * fy);
      const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
      // the translation in vector registers: an fma takes one scalar operand, and the rotation entry already is one
      double cxv = cx * fx, cyv = cyy * fy, czv = cz, oxv = oxi, oyv = oyi;
      asm volatile("" : "+v"(cxv), "+v"(cyv), "+v"(czv), "+v"(oxv), "+v"(oyv));
      const double half_below = __hiloint2double(0x3fdfffff, (int)0xffffffff);      // the largest double below one half
      constexpr int STEP = NW1 * WAVE;
      int k1 = wave * WAVE + lane, chunk1 = wave;
      double kd1 = (double)k1 + 0.5;
      // software prefetch, one whole phase ahead: chunk b of a phase takes its depth from slot b and refills the slot
      // with chunk b of the NEXT phase
      double pzb[B1];
#pragma unroll
      for (int b = 0; b < B1; b++) pzb[b] = plane_load<TD>(rS, k1 + b * STEP, oD);
      for (int s = 0; s < n_bands; s++) {                               // wave-uniform trip count; no other branch in the body
        const int win_lo = (s - m_run + 1) * BAND_PX;
        const unsigned win_span = (unsigned)(2 * m_run * BAND_PX);
#pragma unroll
        for (int b = 0; b < B1; b++) {
          const double pz = pzb[b];                                     // :279
          pzb[b] = plane_load<TD>(rS, k1 + B1 * STEP, oD);              // (past the plane: masked out below)
          const double rd = trunc(kd1 * inv_w), cd = fma(-rd, dW, kd1);
          const double px = fma(cd, ifx, oxv) * pz;                     // :282
          const double py = fma(rd, ify, oyv) * pz;                     // :283
          const double Xf = fma(fr02, pz, fma(fr01, py, fma(fr00, px, cxv)));      // fx * (Rt*point3D).x  :291,295
          const double Yf = fma(fr12, pz, fma(fr11, py, fma(fr10, px, cyv)));
          const double Z = fma(t2, pz, fma(t1, py, fma(-t3, px, czv)));
          const double iz = fast_rcp(Z);                                // :294
          const double tc = fma(Xf, iz, ox);                            // :295
          const double tr = fma(Yf, iz, oy);                            // :296
          // C round() (:297-298) and the bounds (:302-303) as in level_body: conversion of v + (1/2 - ulp) truncates = floor
          const int ri = __double2int_rz(tr + half_below), ci = __double2int_rz(tc + half_below);
          const int lanes_left = n - chunk1 * WAVE;                     // (a property of the chunk: scalar unit)
          const unsigned long long in_image =
              lanes_left >= WAVE ? ~0ull : (lanes_left > 0 ? ((1ull << lanes_left) - 1ull) : 0ull);
          unsigned long long m =
              in_image & __builtin_amdgcn_ballot_w64(min_d < pz) & __builtin_amdgcn_ballot_w64(pz < max_d) &
              __builtin_amdgcn_ballot_w64(tr > -0.5) & __builtin_amdgcn_ballot_w64(ri < H) &
              __builtin_amdgcn_ballot_w64(tc > -0.5) & __builtin_amdgcn_ballot_w64(ci < W);          // :280, :302-303
          const int t = mad24_uniform_b(ri, W, ci);                           // (gn_device.hpp)
          // inside the window of this phase?  (unsigned compare: below win_lo wraps to a huge value)
          const unsigned long long inside = __builtin_amdgcn_ballot_w64((unsigned)(t - win_lo) < win_span);
          if (m & ~inside) {                                            // wave-uniform, rare: this iteration is void
            if (lane == 0) s_ctl[CTL_OOW] = 1;
            m &= inside;
          }
          if (__builtin_amdgcn_inverse_ballot_w64(m)) atomicMax(&s_owner[t & (SLIDE_RING_PX - 1)], k1);   // :358
          if (lane == 0) s_mask[chunk1 & (SLIDE_MASK_RING - 1)] = m;    // the chunk's Jacobian rows, for pass 2
          k1 += STEP;
          chunk1 += NW1;
          kd1 += (double)STEP;
        }
        __syncthreads();
      }
      for (int s = n_bands; s < n_phases; s++) __syncthreads();         // pass 2 finishes the last m + 1 bands
    } else {
      // ================= the waves of PASS 2: residual, Jacobian row, accumulation  (:308-356, 538-540) ===============
      const int wave2 = wave - NW1;
      const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
      const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
      const double t14 = uniform_f64(s_cst[C_T14]);
      const double t4 = uniform_f64(s_cst[C_T4]), t5 = uniform_f64(s_cst[C_T5]), t6 = uniform_f64(s_cst[C_T6]);
      const double t8 = uniform_f64(s_cst[C_T8]), t11 = uniform_f64(s_cst[C_T11]);
      const double t16 = uniform_f64(s_cst[C_T16]), t17 = uniform_f64(s_cst[C_T17]), t24 = uniform_f64(s_cst[C_T24]);
      const double cosy = uniform_f64(s_cst[C_CY]), siny = uniform_f64(s_cst[C_SY]);
      const double t7 = -t6, t9 = -t8, t21 = -t5;
      const double huber_delta = A.huber_delta;
      const bool huber_on = huber_delta > 0.0;
      constexpr int STEP = NW2 * WAVE;

      double acc[NRED];
#pragma unroll
      for (int j = 0; j < NRED; j++) acc[j] = 0.0;
      int n_rows = 0;                 // Jacobian rows this wave fills in this iteration (popcount of the ballots, scalar unit)

      for (int s = 0; s < m_run; s++) __syncthreads();                  // nothing is final yet
      // Phase m: band 0 is final (pass 1 is on band m and reaches back to band 1).  Everything a phase consumes is
      // requested a phase earlier: chunk b of a band takes its operands from slot b and refills the slot with chunk b of
      // the next band -- owner (read and reset for the band that reuses the ring slot), the gather of the source
      // intensity it points at, and the four planes.
      int k2 = wave2 * WAVE + lane, chunk2 = wave2;
      double kd2 = (double)k2 + 0.5;
      // Two sets of slots that swap roles from phase to phase (one is consumed while the other is being filled): with ONE set,
      // refilled in place, the compiler lands the loads in fresh registers and copies them into the slot's registers on the
      // loop's back edge -- 19 v_mov per phase, behind an s_waitcnt vmcnt(0) that ended every load's flight at the phase's end.
      struct Slots {
        int own[B2];
        double i0[B2], pz[B2], gx[B2], gy[B2], i1[B2];
      };
      Slots set_a, set_b;
      auto request = [&](Slots &S, const int b, const int kk) {
        const int o = s_owner[kk & (SLIDE_RING_PX - 1)];                // (lanes past the image: a slot of a band long consumed)
        s_owner[kk & (SLIDE_RING_PX - 1)] = -1;
        S.own[b] = o;
        S.pz[b] = plane_load<TD>(rS, kk, oD);
        S.gx[b] = plane_load<TI>(rT, kk, oGX);                          // gradient at the SOURCE index  :346-347
        S.gy[b] = plane_load<TI>(rT, kk, oGY);
        S.i1[b] = plane_load<TI>(rT, kk, oI);                           // :309
        S.i0[b] = plane_load<TI>(rS, o, oI);                            // :308 (owner -1: past the frame -> 0)
      };
#pragma unroll
      for (int b = 0; b < B2; b++) request(set_a, b, k2 + b * STEP);
      __syncthreads();

      // (three constants that meet another scalar inside one fma sit in vector registers, as in gn_kernels.hip)
      double oxv2 = oxi, oyv2 = oyi, cyv2 = cyy;
      asm volatile("" : "+v"(oxv2), "+v"(oyv2), "+v"(cyv2));
      auto one_phase = [&](const Slots &cur, Slots &nxt, auto huber_tag) {
        constexpr bool HUBER = decltype(huber_tag)::value;
        {
#pragma unroll
          for (int b = 0; b < B2; b++) {
            const int o = cur.own[b];
            const double pz = cur.pz[b], gxi = cur.gx[b], gyi = cur.gy[b], pixel2 = cur.i1[b], pixel1 = cur.i0[b];
            request(nxt, b, k2 + B2 * STEP);                            // this slot's chunk of the next band (final: see above)
            const unsigned long long mbits = s_mask[chunk2 & (SLIDE_MASK_RING - 1)];
            const unsigned long long mrow = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mbits >> 32)) << 32) |
                                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)mbits);
            n_rows += __builtin_popcountll(mrow);
            if (__builtin_amdgcn_inverse_ballot_w64(mrow)) {
              const double res = (o >= 0) ? (pixel2 - pixel1) : 0.0;    // :358
              const double rd = trunc(kd2 * inv_w), cd = fma(-rd, dW, kd2);
              const double px = fma(cd, ifx, oxv2) * pz;
              const double py = fma(rd, ify, oyv2) * pz;
              // factored Jacobian, derivation in gn_kernels.hip (pass 2)
              const double Zr = py * t1 + pz * t2 - px * t3;
              const double t25 = fast_rcp<1>(cz + Zr);                  // :313
              const double Au = pz * t4 + py * t5 + px * t11;           // temp11 = temp15 + x: the reference's slip (:253), kept
              const double Bv = fma(py, t6, fma(pz, t9, fma(px, t14, cyv2)));
              const double Cm = -py * t16 - pz * t17 - px * t24;
              const double Dm = py * t2 - pz * t1;
              double J[6];
              J[0] = (gxi * fx) * t25;                                  // :317
              J[1] = (gyi * fy) * t25;                                  // :322
              J[2] = -(J[0] * Au + J[1] * Bv) * t25;                    // :325-326
              J[3] = J[0] * (cyy - Bv) + J[1] * (Au - px * cx);         // :329-330
              J[4] = (J[0] * cosy + J[1] * siny) * Zr + Cm * J[2];      // :333-336
              J[5] = J[0] * (py * t4 + pz * t21) + J[1] * (pz * t7 + py * t9) + Dm * J[2];     // :339-342
              double Jw[6];
#pragma unroll
              for (int a = 0; a < 6; a++) Jw[a] = J[a];
              if (HUBER) {             // extension, not in the reference: IRLS weight of the Huber loss
                const double ar = fabs(res);
                const double wgt = ar <= huber_delta ? 1.0 : huber_delta / ar;
#pragma unroll
                for (int a = 0; a < 6; a++) Jw[a] = J[a] * wgt;
              }
              int q = 0;
#pragma unroll
              for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = a; c < 6; c++) {
                  acc[q] = fma(Jw[a], J[c], acc[q]);                    // J^T (W) J  :540
                  q++;
                }
              }
#pragma unroll
              for (int a = 0; a < 6; a++) acc[21 + a] = fma(Jw[a], res, acc[21 + a]);   // J^T (W) r  :538
            }
            k2 += STEP;
            chunk2 += NW2;
            kd2 += (double)STEP;
          }
          __syncthreads();
        }
      };
      auto run_phases = [&](auto huber_tag) {
        for (int s = m_run + 1; s < n_phases; s += 2) {                 // band s - m - 1; wave-uniform trip count
          one_phase(set_a, set_b, huber_tag);
          if (s + 1 >= n_phases) break;
          one_phase(set_b, set_a, huber_tag);
        }
      };
      // (two compiled copies, with and without the Huber weights, chosen outside the loop)
      if (huber_on) run_phases(std::true_type{}); else run_phases(std::false_type{});
      acc[RED_VALID] = lane == 0 ? (double)n_rows : 0.0;

      // ---- wave-level transposed butterfly (as gn_level_kernel) ----------------------------------------------------
      reduce_stage_swap<32, false>(acc);
      reduce_stage_swap<16, true>(acc);
      reduce_stage<8, 4>(acc, lane, 8);
      reduce_stage<4, 4>(acc, lane, 4);
      reduce_stage<2, 4>(acc, lane, 2);
      {
        const double total = acc[0] + __shfl_xor(acc[0], 1, WAVE);
        const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 +
                        ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if ((lane & 1) == 0) s_red[wave2 * NRED + idx] = total;
      }
    }
    __syncthreads();

    // ---- cross-wave sum, solve, update, terminate (as gn_level_kernel) --------------------------------------------
    if (wave == 0) {
      double v = 0.0;
      {
        const int j = lane & (NRED - 1);
        const int w0 = (lane >> 5) * (NW2 / 2);
#pragma unroll
        for (int w2 = 0; w2 < NW2 / 2; w2++) v += s_red[(w0 + w2) * NRED + j];
        v += __shfl_xor(v, 32, WAVE);
      }
      double h[21], g[6];
#pragma unroll
      for (int q = 0; q < 21; q++) h[q] = __shfl(v, q, WAVE);
#pragma unroll
      for (int i = 0; i < 6; i++) g[i] = __shfl(v, 21 + i, WAVE);
      const int n_valid = (int)__shfl(v, RED_VALID, WAVE);
      const bool void_iteration = s_ctl[CTL_OOW] != 0;                  // wave-uniform
      double step[6];
      solve6_ldlt(h, g, step);
      double st[6];
      bool finite = true;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        st[i] = s_state[i] - A.lambda * step[i];                        // :539
        finite = finite && (fabs(st[i]) <= 1.79769313486231570815e308);
      }
      double gn2 = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) gn2 += g[i] * g[i];
      const double gnorm = sqrt(gn2);                                   // :380
      const int it = iteration + 1;                                     // :547
      bool done = false;
      if (it >= A.max_iter) done = true;                                // :383
      else if (gnorm < A.min_grad_norm) done = true;                    // :388
      if (!finite) done = true;
      if (void_iteration) {
        // a source pixel left the window: nothing of this iteration counts; the pair goes to the exact kernel
        if (lane == 0) { s_ctl[CTL_DONE] = 2; }
      } else {
        if (!done) write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], s_cst, lane);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 6; i++) s_state[i] = st[i];
          s_ctl[CTL_DONE] = done ? 1 : 0;
          if (!finite) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_NONFINITE;
          if (n_valid < 6) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_RANK_DEFICIENT;
        }
        last_gnorm = gnorm;
        last_valid = n_valid;
      }
    }
    __syncthreads();
    const int done_word = s_ctl[CTL_DONE];
    if (done_word == 2) { handed_over = true; break; }
    iteration++;
    if (done_word) break;
  }

  // ---- epilogue: state and report back to HBM -------------------------------------------------------------
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 6; j++) A.states[(size_t)pair * 6 + j] = s_state[j];
    if (A.reports) {
      A.reports[pair].iterations[A.level] = iteration;                  // completed iterations (the exact kernel resumes here)
      A.reports[pair].gradient_norm = last_gnorm;
      A.reports[pair].valid_pixels[A.level] = last_valid;
      A.reports[pair].flags |= (uint32_t)s_ctl[CTL_FLAGS];
    }
    if (handed_over) handover_append(A, pair);
    s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  }
  }   // next pair
}

}  // namespace

// The instantiation: 768 threads = 3 waves per SIMD and 168 registers; 4 waves of pass 1 with 6 chunks per phase, 8 waves
// of pass 2 with 3 (a chunk of pass 2 costs about 1.6 chunks of pass 1): bands of 24 chunks = 1536 pixels, 21 ring slots.
// Waves are dealt to the SIMDs round-robin, so every SIMD holds one wave of pass 1 and two of pass 2.
#define PHOVO_SLIDE_GEOM 768, 4, 6, 3
using SlideShipped = SlideGeom<PHOVO_SLIDE_GEOM>;

size_t gn_slide_lds_bytes()
{
  return sizeof(double) * (32 + 8 + (size_t)SlideShipped::NW2 * NRED) + sizeof(int) * CTL_COUNT +
         sizeof(unsigned long long) * SLIDE_MASK_RING + sizeof(int) * (size_t)SLIDE_RING_PX;
}

int gn_slide_threads() { return SlideShipped::T; }

// Bands a target may lie away from its source's band (GNLevelArgs::slide_m).  Pass 2 follows pass 1 at m + 1 bands and
// the first and last m + 1 phases of an iteration run one kind of wave only, so m is no larger than the motions need:
// enough bands for max(16 rows, a twelfth of the image height) on top of the band the source pixel itself may lie at the
// end of -- at most what the ring holds.
int gn_slide_reach_bands(int w, int h)
{
  const int rows = h / 12 > 16 ? h / 12 : 16;
  int m = (rows * w + SlideShipped::BAND_PX - 1) / SlideShipped::BAND_PX + 1;
  if (m > SlideShipped::M_MAX) m = SlideShipped::M_MAX;
  return m < 2 ? 2 : m;
}

hipError_t gn_prepare_slide_kernels()
{
  hipError_t e;
#define PHOVO_PREP_SLIDE(GEOM, TI, TD)                                                                             \
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gn_level_kernel_slide<GEOM, TI, TD>),                    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)gn_slide_lds_bytes());                  \
  if (e != hipSuccess) return e;
  PHOVO_PREP_SLIDE(PHOVO_SLIDE_GEOM, double, double)
  PHOVO_PREP_SLIDE(PHOVO_SLIDE_GEOM, float, float)
  PHOVO_PREP_SLIDE(PHOVO_SLIDE_GEOM, __half, float)
#undef PHOVO_PREP_SLIDE
  return hipSuccess;
}

template <int T, int NW1, int B1, int B2>
static hipError_t launch_slide_geom(const GNLevelArgs &a, int storage, int n_blocks, hipStream_t stream)
{
  const dim3 grid((unsigned)n_blocks), block((unsigned)T);
  const size_t lds = gn_slide_lds_bytes();
  switch (storage) {
    case PHOVO_STORAGE_F64: hipLaunchKernelGGL((gn_level_kernel_slide<T, NW1, B1, B2, double, double>), grid, block, lds, stream, a); break;
    case PHOVO_STORAGE_F32: hipLaunchKernelGGL((gn_level_kernel_slide<T, NW1, B1, B2, float, float>), grid, block, lds, stream, a); break;
    case PHOVO_STORAGE_F16: hipLaunchKernelGGL((gn_level_kernel_slide<T, NW1, B1, B2, __half, float>), grid, block, lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t gn_launch_level_slide(const GNLevelArgs &a, int storage, int cu_count, hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  if (!a.handover_out || a.handover_in) return hipErrorInvalidValue;
  if (a.slide_m < 1 || a.slide_m > SlideShipped::M_MAX) return hipErrorInvalidValue;
  const int n_blocks = a.n_pairs < cu_count ? a.n_pairs : cu_count;     // persistent grid, one workgroup per CU
  return launch_slide_geom<PHOVO_SLIDE_GEOM>(a, storage, n_blocks, stream);
}

}  // namespace phovo_hip
