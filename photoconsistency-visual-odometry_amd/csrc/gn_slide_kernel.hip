// Gauss-Newton level kernel for levels whose owner map does not fit LDS (more than ~39 k pixels: 320x240 of a
// 1280x960 pyramid, 640x480 level 0): the SLIDING-WINDOW form.
//
// Same per-pixel arithmetic and the same reference semantics as gn_level_kernel (gn_kernels.hip; reference:
// phovo/include/CPhotoconsistencyOdometryAnalytic.h:191-367, 376-392, 500-563), one workgroup per frame pair, whole
// iteration loop of the level on the device.  What differs is where the scatter of the residuals (:358, last raster
// writer wins) is resolved.  gn_level_kernel's HUGE variant keeps the owner map in HBM (tagged global atomics, 60 B per
// pixel-iteration instead of 40, 0.51 of the roofline).  Here the map is a RING in LDS that slides down the image:
//
//   * the image is cut into bands of 2048 pixels (32 chunks of 64; each of the 8 waves owns 4 chunks of a band);
//   * an iteration is a sequence of phases; in phase s every wave runs PASS 1 (warp, atomicMax into the ring) on its
//     chunks of source band s and then PASS 2 (residual, Jacobian row, 27 sums) on its chunks of target band s - 8,
//     one workgroup barrier per phase;
//   * the ring holds 16 bands (32768 int32 = 128 KiB): while band s is warped, targets may fall into bands s-6 .. s+7;
//     band s - 7 is final when the phase begins -- its owners are read then, and the source intensities they point at
//     are gathered, a whole phase before pass 2 needs them -- and band s - 8 is consumed by pass 2.  Rotations and
//     translations of the sizes Gauss-Newton steps take move a pixel by a few rows; 6 bands are 38 rows at 320 px width.
//   * a source pixel whose target falls OUTSIDE the window sets a flag.  The iteration is then void: the state is left
//     as it was, the pair is put on GNLevelArgs::handover_out and the engine's follow-up launch of gn_level_kernel (HBM
//     owner map, exact for any motion) continues that pair from the same iteration.  Results are therefore exactly
//     the reference's whatever the motion; only the speed depends on the window.
//
// A wave walks its chunks in the same order as gn_level_kernel would with 8 waves (wave, wave + 8, ...).  Depth is read by
// both passes (eight bands apart: the second read is an L2 / Infinity Cache hit); no global atomics, no owner traffic in
// HBM.  512 threads = 2 waves per SIMD and 256 registers: the two passes interleaved need them (27 sums + both passes'
// operands a phase ahead), see the notes in the kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

constexpr int SLIDE_RING_PX = 32768;                          // entries of the ring, a power of two (128 KiB)
static_assert((SLIDE_RING_PX & (SLIDE_RING_PX - 1)) == 0, "ring index is a mask");

// Geometry of one instantiation: T threads, B chunks per wave and band.
template <int T_, int B_>
struct SlideGeom {
  static constexpr int T = T_, B = B_, NW = T_ / WAVE;
  static constexpr int BAND_CHUNKS = NW * B_;
  static constexpr int BAND_PX = BAND_CHUNKS * WAVE;
  static constexpr int RING_BANDS = SLIDE_RING_PX / BAND_PX;
  // Source band s may write target bands s-M+1 .. s+M; band s-M is final when phase s begins (its owners are read
  // then, a phase ahead of their use), band s-M-1 is consumed by pass 2 during phase s.
  static constexpr int M = (RING_BANDS - 1) / 2;
  static_assert(2 * M + 1 <= RING_BANDS, "the band being read and the 2M bands being written must be distinct ring slots");
  static_assert((M + 2) * B_ <= 64, "in-bounds ballots of the chunks between pass 1 and pass 2 live in 64 register lanes");
};

template <int T, int B, typename TI, typename TD>
__global__ __launch_bounds__(T, T / 256) void gn_level_kernel_slide(const GNLevelArgs A)
{
  using G = SlideGeom<T, B>;
  constexpr int NW = G::NW, SLIDE_M = G::M, SLIDE_BAND_CHUNKS = G::BAND_CHUNKS, SLIDE_BAND_PX = G::BAND_PX;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *s_cst = reinterpret_cast<double *>(lds_raw);                 // [32]
  double *s_state = s_cst + 32;                                        // [8]
  double *s_red = s_state + 8;                                         // [NW][NRED]
  int *s_ctl = reinterpret_cast<int *>(s_red + NW * NRED);             // [CTL_COUNT]
  int *s_owner = s_ctl + CTL_COUNT;                                    // [SLIDE_RING_PX]

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  const int n_bands = (A.n_chunks + SLIDE_BAND_CHUNKS - 1) / SLIDE_BAND_CHUNKS;
  // Work queue and loop shape exactly as in gn_level_kernel (one exit every wave reaches; the next ticket is drawn in the
  // block that writes the finished pair back; explicit LDS wait in front of the barrier at the loop head).
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  for (;;) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
  const double min_d = A.min_depth, max_d = A.max_depth;
  const double dW = (double)W, dH = (double)H;
  const double huber_delta = A.huber_delta;
  const bool huber_on = huber_delta > 0.0;

  const int k0 = wave * WAVE + lane;
  // (row, column) of a cursor's pixel from its linear index, carried as a double (gn_device.hpp, rowcol_from_index_floor:
  // pass 2's cursor starts (M + 1) bands in front of the image, at negative indices -- rows counted downwards from 0)
  const RowColFromIndex rc_map = make_rowcol_from_index(W);
  const double kd_step = (double)(NW * WAVE);
  const double kd1_0 = (double)k0, kd2_0 = (double)(k0 - (SLIDE_M + 1) * SLIDE_BAND_PX);

  int iteration = 0;
  double last_gnorm = 0.0;
  int last_valid = 0;
  bool handed_over = false;
  while (true) {
    // ---- constants of this iteration ----------------------------------------------------------------------
    // Pass 1 and pass 2 run interleaved here, so all 22 pose constants and the 10 intrinsics are live at once.  All in
    // SGPRs (gn_level_kernel keeps each pass's own set there) they do not fit next to the descriptors and the loop state:
    // they were spilled to register lanes and came back one v_readlane at a time, and every instruction with two of
    // them as operands needed a copy first (one constant-bus read per instruction) -- 382 vector instructions per pair
    // of chunks instead of 174.  So the twelve that meet another constant inside one instruction (the translation, the
    // rotation entries of pass 1, temp1..3, temp14/15) live in vector registers -- with 2 waves per SIMD there are 256 --
    // and the other twenty stay scalar.
    auto vreg = [](double v) { asm volatile("" : "+v"(v)); return v; };
    const double cx = vreg(s_cst[C_X]), cyy = vreg(s_cst[C_Y]), cz = vreg(s_cst[C_Z]);
    const double r01 = vreg(s_cst[C_R01]), r02 = vreg(s_cst[C_R02]);
    const double r11 = vreg(s_cst[C_R11]), r12 = vreg(s_cst[C_R12]);
    const double t1 = vreg(s_cst[C_T1]), t2 = vreg(s_cst[C_T2]), t3 = vreg(s_cst[C_T3]);
    const double t14 = vreg(s_cst[C_T14]), t15 = vreg(s_cst[C_T15]);
    const double t4 = uniform_f64(s_cst[C_T4]), t5 = uniform_f64(s_cst[C_T5]), t6 = uniform_f64(s_cst[C_T6]);
    const double t8 = uniform_f64(s_cst[C_T8]), t11 = uniform_f64(s_cst[C_T11]);
    const double t16 = uniform_f64(s_cst[C_T16]), t17 = uniform_f64(s_cst[C_T17]), t24 = uniform_f64(s_cst[C_T24]);
    const double cosy = uniform_f64(s_cst[C_CY]), siny = uniform_f64(s_cst[C_SY]);
    const double t7 = -t6, t9 = -t8, t21 = -t5;

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;

    // Two cursors walk the wave's chunks (wave, wave + NW, ...): pass 1 leads, pass 2 follows (M + 1) bands behind and the
    // owner reads M bands behind.  EVERY phase runs both passes on B chunks each, with no branch around a chunk: in
    // the first M + 1 phases pass 2 walks "virtual" chunks in front of the image (negative pixel indices) and in the last
    // M + 1 pass 1 walks past its end.  Such chunks load zeros (a buffer load outside its descriptor returns 0), ballot
    // to an empty mask and do nothing.  A branch around a chunk body would cost far more than those idle chunks: the
    // loads a chunk issues for the NEXT phase land in registers that are live around the loop, and at the join behind a
    // conditional chunk the compiler parks each of them in a temporary, waits for it (s_waitcnt vmcnt(0)) and copies
    // it -- which serialises every chunk behind its own prefetch (measured: 8.0 ms per launch either way, prefetch or not).
    int k1 = k0, j1 = 0;
    double kd1 = kd1_0;
    int k2 = k0 - (SLIDE_M + 1) * SLIDE_BAND_PX, j2 = -(SLIDE_M + 1) * B;
    double kd2 = kd2_0;
    // chunk j's "valid and landed in bounds" ballot lives in lane (j & 63) of two registers from pass 1 to pass 2
    int inb_lo = 0, inb_hi = 0;
    int n_rows = 0;                 // Jacobian rows this wave fills in this iteration (popcount of the ballots, scalar unit)
    // Software prefetch, B chunks (one whole phase) ahead in each pass: chunk b of a phase takes its operands from slot b
    // and refills the slot with chunk b of the NEXT phase.  Every wave then has 4 + 16 plane loads and 4 gathers in
    // flight at all times (~12 KB; 8 waves per CU, ~96 KB per CU): the level is streamed from HBM -- little of it stays in
    // the Infinity Cache between iterations (2048 pairs x 3 MB) -- and with 2 waves per SIMD there are few other waves to
    // hide an HBM miss behind, so the distance is a whole phase rather than one chunk.
    double pzb[B], pz_s[B], gx_s[B], gy_s[B], i1_s[B];
#pragma unroll
    for (int b = 0; b < B; b++) {
      pzb[b] = plane_load<TD>(rD0, k0 + b * NW * WAVE);
      pz_s[b] = gx_s[b] = gy_s[b] = i1_s[b] = 0.0;                      // pass 2 starts on virtual chunks; its first real
    }                                                                   // ones are requested a phase ahead like all others


    // ---- pass 1 on one chunk: warp, bounds, window, atomicMax into the ring  (:279-303, 358) ---------------
    auto pass1_chunk = [&](const int win_lo, const unsigned win_span, double &slot) {
      const double pz = slot;                                           // :279
      slot = plane_load<TD>(rD0, k1 + B * NW * WAVE);                   // this slot's chunk of the next phase (past the plane: 0)
      double cd1, rd1;
      rowcol_from_index_floor(kd1, rc_map, cd1, rd1);
      const double px = (cd1 - ox) * pz * ifx;                          // :282
      const double py = (rd1 - oy) * pz * ify;                          // :283
      const double X = fma(r02, pz, fma(r01, py, fma(t15, px, cx)));    // Rt*point3D  :291
      const double Y = fma(r12, pz, fma(r11, py, fma(t14, px, cyy)));
      const double Z = fma(t2, pz, fma(t1, py, fma(-t3, px, cz)));
      const double iz = fast_rcp(Z);                                    // :294
      const double tc = (X * fx) * iz + ox;                             // :295
      const double tr = (Y * fy) * iz + oy;                             // :296
      const double rr = round_half_up_from(tr), rc = round_half_up_from(tc);       // :297-298 (arguments > -0.5)
      unsigned long long m =
          __builtin_amdgcn_ballot_w64(k1 < n) & __builtin_amdgcn_ballot_w64(min_d < pz) &
          __builtin_amdgcn_ballot_w64(pz < max_d) & __builtin_amdgcn_ballot_w64(tr > -0.5) &
          __builtin_amdgcn_ballot_w64(rr < dH) & __builtin_amdgcn_ballot_w64(tc > -0.5) &
          __builtin_amdgcn_ballot_w64(rc < dW);                         // :280, :302-303
      const int t = (int)fma(rr, dW, rc);
      // inside the window of this phase?  (unsigned compare: below win_lo wraps to a huge value)
      const unsigned long long inside = __builtin_amdgcn_ballot_w64((unsigned)(t - win_lo) < win_span);
      if (m & ~inside) {                                                // wave-uniform, rare: this iteration is void
        if (lane == 0) s_ctl[CTL_OOW] = 1;
        m &= inside;
      }
      n_rows += __builtin_popcountll(m);
      if (__builtin_amdgcn_inverse_ballot_w64(m)) atomicMax(&s_owner[t & (SLIDE_RING_PX - 1)], k1);   // :358
      inb_lo = writelane_b32(inb_lo, (int)(unsigned)m, j1 & 63);
      inb_hi = writelane_b32(inb_hi, (int)(unsigned)(m >> 32), j1 & 63);
      k1 += NW * WAVE;
      j1++;
      kd1 += kd_step;
    };

    // ---- pass 2 on one chunk: residual, Jacobian row, accumulation  (:308-356, 538-540) -------------------
    auto pass2_chunk = [&](auto huber_tag, const int o, const double pixel1, double &s_pz, double &s_gx, double &s_gy,
                           double &s_i1) {
      constexpr bool HUBER = decltype(huber_tag)::value;
      const double pz = s_pz, gxi = s_gx, gyi = s_gy, pixel2 = s_i1;
      {
        const int kk = k2 + B * NW * WAVE;                              // this slot's chunk of the next phase
        s_pz = plane_load<TD>(rD0, kk);
        s_gx = plane_load<TI>(rGX, kk);                                 // gradient at the SOURCE index  :346-347
        s_gy = plane_load<TI>(rGY, kk);
        s_i1 = plane_load<TI>(rI1, kk);                                 // :309
      }
      const unsigned long long mbits =
          ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(inb_hi, j2 & 63) << 32) |
          (unsigned long long)(unsigned)__builtin_amdgcn_readlane(inb_lo, j2 & 63);
      if (__builtin_amdgcn_inverse_ballot_w64(mbits)) {
        const double res = (o >= 0) ? (pixel2 - pixel1) : 0.0;          // :358
        double cd2, rd2;
        rowcol_from_index_floor(kd2, rc_map, cd2, rd2);
        const double px = (cd2 - ox) * pz * ifx;
        const double py = (rd2 - oy) * pz * ify;
        // factored Jacobian, derivation in gn_kernels.hip (pass 2)
        const double Zr = py * t1 + pz * t2 - px * t3;
        const double t25 = fast_rcp(cz + Zr);                           // :313
        const double Au = pz * t4 + py * t5 + px * t11;                 // temp11 = temp15 + x: the reference's slip (:253), kept
        const double Bv = py * t6 + pz * t9 + px * t14 + cyy;
        const double Cm = -py * t16 - pz * t17 - px * t24;
        const double Dm = py * t2 - pz * t1;
        double J[6];
        J[0] = (gxi * fx) * t25;                                        // :317
        J[1] = (gyi * fy) * t25;                                        // :322
        J[2] = -(J[0] * Au + J[1] * Bv) * t25;                          // :325-326
        J[3] = J[0] * (cyy - Bv) + J[1] * (Au - px * cx);               // :329-330
        J[4] = (J[0] * cosy + J[1] * siny) * Zr + Cm * J[2];            // :333-336
        J[5] = J[0] * (py * t4 + pz * t21) + J[1] * (pz * t7 + py * t9) + Dm * J[2];     // :339-342
        double Jw[6];
#pragma unroll
        for (int a = 0; a < 6; a++) Jw[a] = J[a];
        if (HUBER) {                 // extension, not in the reference: IRLS weight of the Huber loss
          const double ar = fabs(res);
          const double wgt = ar <= huber_delta ? 1.0 : huber_delta / ar;
#pragma unroll
          for (int a = 0; a < 6; a++) Jw[a] = J[a] * wgt;
        }
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
          for (int b = a; b < 6; b++) {
            acc[q] = fma(Jw[a], J[b], acc[q]);                          // J^T (W) J  :540
            q++;
          }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] = fma(Jw[a], res, acc[21 + a]);   // J^T (W) r  :538
      }
      k2 += NW * WAVE;
      j2++;
      kd2 += kd_step;
    };

    // ---- the phases of this iteration ---------------------------------------------------------------------
    // (two compiled copies, with and without the Huber weights, chosen outside the loop: the reference path carries none
    // of the extension's instructions and the 27 sums are not shuffled through a join after every chunk)
    auto run_phases = [&](auto huber_tag) {
    // Everything a phase consumes was requested a phase earlier: the depth of pass 1's chunks and the four planes of
    // pass 2's (inside the chunk bodies, one chunk ahead), and -- here -- the owners of the band pass 2 takes NEXT phase
    // together with the gathers of the source intensities they point at.
    int own_n[B];
    double i0_n[B];
#pragma unroll
    for (int b = 0; b < B; b++) { own_n[b] = -1; i0_n[b] = 0.0; }
    int k3 = k0 - SLIDE_M * SLIDE_BAND_PX;                              // this wave's first pixel of the band whose owners are read
    for (int s = 0; s < n_bands + SLIDE_M + 1; s++) {                   // wave-uniform trip count; no other branch in the body
      int own_c[B];
      double i0_c[B];
#pragma unroll
      for (int b = 0; b < B; b++) { own_c[b] = own_n[b]; i0_c[b] = i0_n[b]; }
      // band s - M: nobody writes it any more (this phase's pass 1 reaches back to s - M + 1 only)
#pragma unroll
      for (int b = 0; b < B; b++) {
        const int kk = k3 + b * NW * WAVE;
        own_n[b] = -1;
        if ((unsigned)kk < (unsigned)n) {
          own_n[b] = s_owner[kk & (SLIDE_RING_PX - 1)];
          s_owner[kk & (SLIDE_RING_PX - 1)] = -1;                       // ready for the band that reuses this slot
        }
      }
#pragma unroll
      for (int b = 0; b < B; b++) i0_n[b] = plane_load<TI>(rI0, own_n[b]);     // :308 (owner -1: past the plane -> 0)
      k3 += SLIDE_BAND_PX;
      const int win_lo = (s - SLIDE_M + 1) * SLIDE_BAND_PX;
      const unsigned win_span = (unsigned)(2 * SLIDE_M * SLIDE_BAND_PX);
#pragma unroll
      for (int b = 0; b < B; b++) pass1_chunk(win_lo, win_span, pzb[b]);
#pragma unroll
      for (int b = 0; b < B; b++) pass2_chunk(huber_tag, own_c[b], i0_c[b], pz_s[b], gx_s[b], gy_s[b], i1_s[b]);
      __syncthreads();
    }
    };
    if (huber_on) run_phases(std::true_type{}); else run_phases(std::false_type{});
    acc[RED_VALID] = lane == 0 ? (double)n_rows : 0.0;

    // ---- wave-level transposed butterfly, cross-wave sum, solve, update, terminate (as gn_level_kernel) ---
    reduce_stage_swap<32, false>(acc);
    reduce_stage_swap<16, true>(acc);
    reduce_stage<8, 4>(acc, lane, 8);
    reduce_stage<4, 4>(acc, lane, 4);
    reduce_stage<2, 4>(acc, lane, 2);
    {
      const double total = acc[0] + __shfl_xor(acc[0], 1, WAVE);
      const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 +
                      ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
      if ((lane & 1) == 0) s_red[wave * NRED + idx] = total;
    }
    __syncthreads();
    if (wave == 0) {
      double v = 0.0;
      {
        const int j = lane & (NRED - 1);
        const int w0 = (lane >> 5) * (NW / 2);
#pragma unroll
        for (int w2 = 0; w2 < NW / 2; w2++) v += s_red[(w0 + w2) * NRED + j];
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

// The instantiation: 512 threads, four chunks per wave and band.  Measured alternatives (1280x960 level 2, 2048 pairs x 5
// iterations; the exact kernel with the owner map in HBM takes 7.78 ms): 1024 threads / one chunk (4 waves per SIMD, 128
// registers: the fused passes spill 176 registers into the pixel loop) 47 ms; 512 threads / two chunks 6.63 ms; this one 6.41 ms.
#define PHOVO_SLIDE_GEOM 512, 4

size_t gn_slide_lds_bytes()
{
  return sizeof(double) * (32 + 8 + (size_t)(512 / WAVE) * NRED) + sizeof(int) * (CTL_COUNT + (size_t)SLIDE_RING_PX);
}

int gn_slide_window_pixels() { return (SlideGeom<PHOVO_SLIDE_GEOM>::M - 1) * SlideGeom<PHOVO_SLIDE_GEOM>::BAND_PX; }

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

template <int T, int B>
static hipError_t launch_slide_geom(const GNLevelArgs &a, int storage, int n_blocks, hipStream_t stream)
{
  const dim3 grid((unsigned)n_blocks), block((unsigned)T);
  const size_t lds = gn_slide_lds_bytes();
  switch (storage) {
    case PHOVO_STORAGE_F64: hipLaunchKernelGGL((gn_level_kernel_slide<T, B, double, double>), grid, block, lds, stream, a); break;
    case PHOVO_STORAGE_F32: hipLaunchKernelGGL((gn_level_kernel_slide<T, B, float, float>), grid, block, lds, stream, a); break;
    case PHOVO_STORAGE_F16: hipLaunchKernelGGL((gn_level_kernel_slide<T, B, __half, float>), grid, block, lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t gn_launch_level_slide(const GNLevelArgs &a, int storage, int cu_count, hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  if (!a.handover_out || a.handover_in) return hipErrorInvalidValue;
  const int n_blocks = a.n_pairs < cu_count ? a.n_pairs : cu_count;     // persistent grid, one workgroup per CU
  return launch_slide_geom<PHOVO_SLIDE_GEOM>(a, storage, n_blocks, stream);
}

}  // namespace phovo_hip
