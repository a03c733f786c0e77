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
//   solve   wave 0: 6x6 LDL^T solve, state -= lambda * H^-1 g,
//           termination test, pose constants of the next iteration.
// No MFMA: J^T J is a 6 x N by N x 6 contraction, a reduction, not a GEMM tile.
//
// The reference's `temp11 = cos(pitch)*cos(yaw)+x` transcription bug (:253) is reproduced on purpose:
// parity with the reference's poses requires it ("read this first" item 4).

//
// Two kernels share the per-(pair, level) body below (level_body): gn_level_kernel runs ONE level for every pair it draws
// (one launch per active level), gn_fused_kernel runs a pair through SEVERAL consecutive levels back to back inside the
// workgroup that drew it -- the reference's own loop structure, per pair over levels (:500-563) -- so that with
// data-dependent termination a level boundary is not a chip-wide launch boundary.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

#ifdef PHOVO_TIMELINE
// Diagnostic build only (tools/timeline.py): when a pair started and ended on the 100 MHz wall clock (low 31 bits) and in
// which workgroup, left in the report slots of levels 15 / 14, which no configuration of the tool uses.
#define PHOVO_TIMELINE_BEGIN const int tl_begin = (int)(wall_clock64() & 0x7fffffffull);
#define PHOVO_TIMELINE_END(rep) { (rep).valid_pixels[15] = tl_begin; (rep).valid_pixels[14] = (int)(wall_clock64() & 0x7fffffffull); (rep).iterations[15] = (int)blockIdx.x; }
#else
#define PHOVO_TIMELINE_BEGIN
#define PHOVO_TIMELINE_END(rep)
#endif

// LDS blocks of a workgroup (offsets are multiples of 8; s_owner is sized for the largest level the workgroup will run).
struct LevelLds {
  double *cst;                 // [32]  pose constants of the running iteration
  double *state;               // [8]   state vector
  double *red;                 // [NW][NRED] one row of wave sums per wave
  int *ctl;                    // [CTL_COUNT]
  unsigned long long *mask;    // [n_chunks] in-bounds ballots (!MASK_REG)
  int *owner;                  // owner map (OWNER_LDS) or its leading n_lds entries (owner map in HBM)
  double *i0;                  // [n] source intensity (SRC_LDS)
};

// One frame pair at one pyramid level: every Gauss-Newton iteration of that level, from iteration `iteration` (0, or the
// count an earlier launch had reached) until the termination test stops it (:547-549).  Called by ALL threads of the
// workgroup with workgroup-uniform arguments; the state comes from A.states (HBM) or, with state_in_lds, from L.state,
// where the level before left it; on return L.state holds the result and `iteration` the iterations executed in all.
// T threads per workgroup.  SRC_LDS: source intensity plane staged in LDS.  OWNER_LDS: owner map in LDS (else in global
// memory).  MASK_REG: the per-pixel "warped in bounds" flags of a lane live in one 64-bit register (needs <= 64 chunks
// per wave), else in an LDS ballot array.  TI / TD: storage type of the intensity+gradient planes / of the depth plane
// (double = reference-exact).  PARK (owner map in LDS, no SRC_LDS): whatever LDS the geometry leaves unused keeps the depth of
// the image's leading A.depth_lds_chunks chunks from pass 1 to pass 2 (the block L.i0).
template <int T, bool SRC_LDS, bool OWNER_LDS, bool MASK_REG, typename TI, typename TD, bool PARK = false>
__device__ __forceinline__ void level_body(const GNLevelArgs &A, const LevelLds &L, const int pair, const bool state_in_lds,
                                           int &iteration, double &last_gnorm, int &last_valid)
{
  constexpr int NW = T / WAVE;
#ifdef PHOVO_PHASE_STAMPS
  const unsigned long long stamp_entry = wall_clock64();
#endif
  double *const s_cst = L.cst, *const s_state = L.state, *const s_red = L.red, *const s_i0 = L.i0;
  int *const s_ctl = L.ctl, *const s_owner = L.owner;
  // (ballots in global memory: written and read back by the same wave, chunk by chunk -- program order is all it needs)
  unsigned long long *const s_mask = (!MASK_REG && A.g_mask) ? A.g_mask + (size_t)pair * (size_t)A.n_chunks : L.mask;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  // wave-uniform by construction; saying so lets the chunk loops run on the scalar unit (s_cmp / s_cbranch)
  // instead of exec-masked vector compares
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  const int n_lds = OWNER_LDS ? 0 : A.n_lds;      // owner map in HBM: targets below n_lds are resolved in LDS all the same
  const int depth_chunks = PARK ? A.depth_lds_chunks : 0;

  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const unsigned char *tgt_frame = A.planes + (size_t)A.tgt[pair] * A.frame_bytes;
  int *g_owner = OWNER_LDS ? nullptr : A.g_owner + (size_t)pair * (size_t)n;
  // ONE descriptor per frame, the plane chosen by a scalar offset (plane_load): the hardware checks voffset + soffset
  // against num_records, so the bound is the frame's -- an index past its plane reads the next plane of the same frame
  // (the chunk-ahead prefetches do that: those lanes are masked out of every use) and never leaves the frame; the index
  // -1 of "no owner" wraps to the top of the address range and still reads 0.
  const __amdgpu_buffer_rsrc_t rS = frame_rsrc(src_frame, A.frame_bytes), rT = frame_rsrc(tgt_frame, A.frame_bytes);
  const int oI = (int)A.plane_off[PLANE_I], oD = (int)A.plane_off[PLANE_D];
  const int oGX = (int)A.plane_off[PLANE_GX], oGY = (int)A.plane_off[PLANE_GY];

  // ---- level prologue -------------------------------------------------------------------
  // (a level that follows another one in the same workgroup: every wave must have read the previous level's last
  // CTL_DONE before wave 0 clears it below)
  if (state_in_lds) __syncthreads();
  if (!OWNER_LDS) {
    // Owner map in HBM: entries carry the iteration they were written in (OWNER_TAG_SHIFT), so nothing has to be
    // reset between iterations; it is wiped once per pair here (the barrier below waits for the stores, and the
    // map is touched by this workgroup only).
    for (int k = n_lds + tid; k < n; k += T) g_owner[k] = -1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  if (SRC_LDS) {
    for (int k = tid; k < n; k += T) s_i0[k] = plane_load<TI>(rS, k, oI);
  }
  if (wave == 0) {
    double st[6];
#pragma unroll
    for (int j = 0; j < 6; j++) st[j] = state_in_lds ? s_state[j] : A.states[(size_t)pair * 6 + j];
    write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], s_cst, lane);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) s_state[j] = st[j];
      s_ctl[CTL_DONE] = 0;
      if (!state_in_lds) s_ctl[CTL_FLAGS] = 0;
    }
  }
  __syncthreads();

  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double min_d = A.min_depth, max_d = A.max_depth;
  const double dW = (double)W;
  // px = (c - ox) * pz * ifx (:282) as fma(c + 0.5, ifx, -(ox + 0.5) * ifx) * pz: one fma and one product
  const double oxi = uniform_f64(-(ox + 0.5) * ifx), oyi = uniform_f64(-oy * ify);
  const double huber_delta = A.huber_delta;
  const bool huber_on = huber_delta > 0.0;

  // A wave walks the image in chunks of 64 consecutive pixels, NW chunks apart; (row, column) of a
  // lane's pixel is carried along instead of divided out per pixel.
  const int k0 = wave * WAVE + lane;
  // (row, column) of a lane's pixel from its linear index, carried as the double k + 0.5: row = trunc((k + 0.5) / W) -- the
  // quotient lies at least 0.5 / W >= 2.4e-7 away from every integer, its product form is off by less than 1e-12, so the
  // truncation is exact -- and column + 0.5 = (k + 0.5) - row * W, an exact integer fma; the half pixel is folded into the
  // constant of the unprojection below.  Four instructions per chunk (add, mul, trunc, fma), three scalar constants.
  const double inv_w = uniform_f64(1.0 / dW);
  const double kd0 = (double)k0 + 0.5, kd_step = (double)(NW * WAVE);
#define PHOVO_ROWCOL_BEGIN double kd = kd0, cd, rd;
#define PHOVO_ROWCOL_HERE { rd = trunc(kd * inv_w); cd = fma(-rd, dW, kd); }
#define PHOVO_ROWCOL_NEXT kd += kd_step;

#ifdef PHOVO_PHASE_STAMPS
  // Diagnostic build only (make ... EXTRA=-DPHOVO_PHASE_STAMPS): where the iterations of ONE pair spend their time, per wave,
  // in ticks of the 100 MHz wall clock: pass 1, the barrier behind it, pass 2, the butterfly, the barrier in front of the
  // solve, the solve (wave 0) or the wait for it.  Printed by workgroup 0 for the first pair it draws.
  unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0}, stamp_last = wall_clock64();
  const unsigned long long stamp_prologue = stamp_last - stamp_entry;
  unsigned long long solve_sum[4] = {0, 0, 0, 0};
#define PHOVO_STAMP(i) { const unsigned long long t_ = wall_clock64(); stamp_sum[i] += t_ - stamp_last; stamp_last = t_; }
#else
#define PHOVO_STAMP(i)
#endif
  while (true) {
#ifdef PHOVO_PHASE_STAMPS
    stamp_last = wall_clock64();
#endif
    // ---- constants of this iteration (uniform -> SGPRs) -----------------------------------
    const double cx = uniform_f64(s_cst[C_X]), cyy = uniform_f64(s_cst[C_Y]), cz = uniform_f64(s_cst[C_Z]);
    const double r01 = uniform_f64(s_cst[C_R01]), r02 = uniform_f64(s_cst[C_R02]);
    const double r11 = uniform_f64(s_cst[C_R11]), r12 = uniform_f64(s_cst[C_R12]);
    // the two projected rows of Rt already multiplied by the focal lengths (pass 1 only): (X*fx)*iz + ox (:295) becomes
    // one fma on X*fx built directly -- eight products per iteration and wave instead of two per pixel
    const double fr00 = uniform_f64(uniform_f64(s_cst[C_T15]) * fx), fr01 = uniform_f64(r01 * fx), fr02 = uniform_f64(r02 * fx);
    const double fr10 = uniform_f64(uniform_f64(s_cst[C_T14]) * fy), fr11 = uniform_f64(r11 * fy), fr12 = uniform_f64(r12 * fy);
    const double t1 = uniform_f64(s_cst[C_T1]), t2 = uniform_f64(s_cst[C_T2]), t3 = uniform_f64(s_cst[C_T3]);
    const double t14 = uniform_f64(s_cst[C_T14]);

    // Owner map in HBM: this iteration's tag (1..OWNER_TAG_PERIOD); when the tags start over the map is wiped.
    int owner_tag = 0;
    if constexpr (!OWNER_LDS) {
      const int tg = iteration % OWNER_TAG_PERIOD + 1;
      if (iteration > 0 && tg == 1) {                                     // uniform: every wave takes it
        for (int kk = n_lds + tid; kk < n; kk += T) g_owner[kk] = -1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
      }
      owner_tag = tg << OWNER_TAG_SHIFT;
    }

    // ---- pass 1: warp every source pixel, resolve who owns each target pixel -------------
    // MASK_REG: chunk j's "landed in bounds" ballot lives in lane j of two registers (v_writelane here, v_readlane
    // + exec mask in pass 2): no per-lane shifting and masking in either pass
    int inb_lo = 0, inb_hi = 0;
    int n_rows = 0;                 // Jacobian rows this wave fills in this iteration (!MASK_REG: popcount of the ballots as they come)
    {
      int k = k0, j = 0;
      PHOVO_ROWCOL_BEGIN
      // software prefetch: the depth of the NEXT chunk is requested before this chunk is processed, so
      // every wave keeps a load in flight while it computes (the passes are bound by bytes in flight per CU)
      double pz_next = plane_load<TD>(rS, k, oD);
      // the translation sits in vector registers during this pass (pass 1 has registers to spare): an fma takes one
      // scalar operand, and the rotation entry already is one
      double cxv = cx * fx, cyv = cyy * fy, czv = cz, oxv = oxi, oyv = oyi;
      asm volatile("" : "+v"(cxv), "+v"(cyv), "+v"(czv), "+v"(oxv), "+v"(oyv));
      // warp of one 64-pixel chunk: ballot of "valid and landed in bounds" and the target index of every lane
      auto warp_chunk = [&](const double pz, const int chunk, unsigned long long &m_out, int &t_out) {
        // No branch around the arithmetic: a lane that fails the depth gate computes on whatever it loaded and is
        // dropped by `valid` below (a whole wave of invalid pixels is rare), and the ballot of a flat condition is
        // the AND of the compare masks -- scalar work only.
        // depth gate: k < n, min_d < pz < max_d  (:280), folded into the ballot below
        PHOVO_ROWCOL_HERE
        const double px = fma(cd, ifx, oxv) * pz;                         // :282  ((c - ox) * ifx as one fma)
        const double py = fma(rd, ify, oyv) * pz;                         // :283
        const double Xf = fma(fr02, pz, fma(fr01, py, fma(fr00, px, cxv)));      // fx * (Rt*point3D).x  :291,295
        const double Yf = fma(fr12, pz, fma(fr11, py, fma(fr10, px, cyv)));
        const double Z = fma(t2, pz, fma(t1, py, fma(-t3, px, czv)));
        const double iz = fast_rcp(Z);                                    // :294
        const double tc = fma(Xf, iz, ox);                                // :295
        const double tr = fma(Yf, iz, oy);                                // :296
        // C round(), half away from zero (:297-298), then 0 <= . < size (:302-303): round(v) >= 0 iff v > -0.5
        // (NaN fails every comparison)
        // round() of a coordinate that passes `> -0.5` is floor(v + p) with p the largest double below one half
        // (round_half_up_from, gn_device.hpp) and v + p >= 0 there, so the conversion to int -- which truncates -- IS that
        // floor: one add + one conversion per coordinate, the upper bounds as 32-bit compares, the target index as one
        // integer multiply-add.  A coordinate beyond the int range saturates and fails the upper bound; NaN fails `> -0.5`.
        const double half_below = __hiloint2double(0x3fdfffff, (int)0xffffffff);
        const int ri = __double2int_rz(tr + half_below), ci = __double2int_rz(tc + half_below);
        // One ballot per comparison, ANDed on the scalar unit: the ballot of an AND of comparisons would be
        // rebuilt lane by lane (v_cndmask + v_cmp) by this compiler.
        // (k < n is a property of the chunk, not of the lane: the lanes of the image's last, partial chunk -- scalar unit)
        const int lanes_left = n - chunk * WAVE;
        const unsigned long long in_image = lanes_left >= WAVE ? ~0ull : ((1ull << lanes_left) - 1ull);
        const unsigned long long m =
            in_image & __builtin_amdgcn_ballot_w64(min_d < pz) &
            __builtin_amdgcn_ballot_w64(pz < max_d) & __builtin_amdgcn_ballot_w64(tr > -0.5) &
            __builtin_amdgcn_ballot_w64(ri < H) & __builtin_amdgcn_ballot_w64(tc > -0.5) &
            __builtin_amdgcn_ballot_w64(ci < W);
        m_out = m;
        t_out = mad24_uniform_b(ri, W, ci);           // (one multiply-add + one shift-add to the LDS address: gn_device.hpp)
      };
      auto keep_mask = [&](const unsigned long long m, const int chunk) {
        if (!MASK_REG) n_rows += __builtin_popcountll(m);
        if (MASK_REG) {
          inb_lo = writelane_b32(inb_lo, (int)(unsigned)m, j);
          inb_hi = writelane_b32(inb_hi, (int)(unsigned)(m >> 32), j);
        } else {
          if (lane == 0) s_mask[chunk] = m;
        }
      };
      if constexpr (OWNER_LDS) {
        auto chunk_body = [&](const int chunk) {
          const double pz = pz_next;                                      // :279
          pz_next = plane_load<TD>(rS, k + NW * WAVE, oD);                   // (past the plane: masked out)
          // The depth plane is the one plane both passes read.  Whatever LDS this geometry leaves unused keeps the depth of
          // the image's LEADING chunks for pass 2 (the same lane reads back what it wrote); pass 2 walks backwards, so the
          // trailing chunks -- read last here -- still come from the XCD's L2.  The kernel is bound by bytes at the fabric.
          if (PARK && chunk < depth_chunks) s_i0[k] = pz;                 // (wave-uniform; s_i0: the block behind the owner map)
          unsigned long long m;
          int t;
          warp_chunk(pz, chunk, m, t);
          if (__builtin_amdgcn_inverse_ballot_w64(m)) atomicMax(&s_owner[t], k);     // last raster writer wins  :358
          keep_mask(m, chunk);
          k += NW * WAVE;
          j++;
          PHOVO_ROWCOL_NEXT
        };
        // two chunks per trip, written out by hand: the ballot / lane accesses are convergent operations, which the
        // compiler will not duplicate for a run-time trip count (#pragma unroll is refused); the bounds are wave-uniform
        int chunk = wave;
        for (; chunk + NW < A.n_chunks; chunk += 2 * NW) {
          chunk_body(chunk);
          chunk_body(chunk + NW);
        }
        if (chunk < A.n_chunks) chunk_body(chunk);
      } else {
        // Owner map in HBM.  The memory counter of a wave retires in order, so a load issued behind a global atomic
        // waits for that atomic (about 2800 cycles with every CU issuing them): a loop of load - compute - atomic per
        // chunk ran at the atomic's latency (1600 cycles per chunk against 740 with the map in LDS).  Here four chunks
        // go together: their depths were requested a group ago, the next group's are requested first, then the four
        // are warped, and only then do the four atomics go out.
        constexpr int G = 4;
        double pzg[G], pzn[G];
        int pend_t[G], pend_v[G];
        unsigned long long pend_m[G];
        pzg[0] = pz_next;
#pragma unroll
        for (int g = 1; g < G; g++) pzg[g] = plane_load<TD>(rS, k + g * NW * WAVE, oD);
        int chunk = wave;
        while (chunk < A.n_chunks) {
#pragma unroll
          for (int g = 0; g < G; g++) pzn[g] = plane_load<TD>(rS, k + (G + g) * NW * WAVE, oD);     // (past the plane: masked out)
          int count = 0;
#pragma unroll
          for (int g = 0; g < G; g++) {
            if (chunk < A.n_chunks) {                                     // wave-uniform
              warp_chunk(pzg[g], chunk, pend_m[g], pend_t[g]);
              pend_v[g] = owner_tag | k;
              keep_mask(pend_m[g], chunk);
              count = g + 1;
              chunk += NW;
              k += NW * WAVE;
              j++;
              PHOVO_ROWCOL_NEXT
            }
          }
#pragma unroll
          for (int g = 0; g < G; g++) {
            if (g < count && __builtin_amdgcn_inverse_ballot_w64(pend_m[g])) {
              // last raster writer of THIS iteration wins (:358); the leading n_lds targets are resolved in LDS (plain
              // source index, reset by pass 2), the others in HBM (tagged)
              if (pend_t[g] < n_lds) atomicMax(&s_owner[pend_t[g]], pend_v[g] & OWNER_INDEX_MASK);
              else atomicMax(&g_owner[pend_t[g]], pend_v[g]);
            }
          }
#pragma unroll
          for (int g = 0; g < G; g++) pzg[g] = pzn[g];
        }
      }
    }
    PHOVO_STAMP(0)
    __syncthreads();
    PHOVO_STAMP(1)

    // ---- pass 2: residual, Jacobian row, normal-equation accumulation ---------------------
    const double t4 = uniform_f64(s_cst[C_T4]), t5 = uniform_f64(s_cst[C_T5]), t6 = uniform_f64(s_cst[C_T6]);
    const double t8 = uniform_f64(s_cst[C_T8]), t11 = uniform_f64(s_cst[C_T11]);
    const double t16 = uniform_f64(s_cst[C_T16]), t17 = uniform_f64(s_cst[C_T17]), t24 = uniform_f64(s_cst[C_T24]);
    const double cosy = uniform_f64(s_cst[C_CY]), siny = uniform_f64(s_cst[C_SY]);
    const double t7 = -t6, t9 = -t8, t21 = -t5;

    double acc[NRED];
#pragma unroll
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;

    // Pass 2 exists in two compiled copies, selected by a wave-uniform branch OUTSIDE the pixel loop, so that the
    // reference path (no Huber weights) carries none of the extension's instructions.
    auto pass2 = [&](auto huber_tag) {
      constexpr bool HUBER = decltype(huber_tag)::value;
      // With the owner map in LDS a wave walks its chunks BACKWARDS here: pass 1 has just read the depth plane front to
      // back and the next iteration's pass 1 will start at the front again, so the depth lines this pass meets first and
      // last are the ones an XCD's L2 (4 MB shared by 64 workgroups: a line lives a few microseconds) still holds from
      // the pass before / keeps for the pass after.  The kernel is bound by bytes at the fabric; the
      // second read of the depth plane is the only plane traffic that is not algorithmic.
      constexpr bool REV = OWNER_LDS;
      const int my_chunks = wave < A.n_chunks ? (A.n_chunks - 1 - wave) / NW + 1 : 0;       // wave-uniform
      const int first = REV ? my_chunks - 1 : 0;                                            // position of the first chunk taken
      const int k_step = REV ? -NW * WAVE : NW * WAVE;
      int k = k0 + first * (NW * WAVE), j = first;
      double kd = kd0 + (double)first * kd_step, cd, rd;
      // software prefetch, as in pass 1: owner + four planes of the NEXT chunk are requested (and the
      // gathered source intensity right behind them) before this chunk's arithmetic starts.
      int o_n = -1;
      double pz_n = 0.0, gx_n = 0.0, gy_n = 0.0, i1_n = 0.0, i0_n = 0.0;
      // three constants that meet another scalar inside one fma sit in vector registers (an instruction reads one scalar
      // operand; the compiler otherwise copies them in front of every use: three v_mov_b64 per chunk)
      double oxv2 = oxi, oyv2 = oyi, cyv2 = cyy;
      asm volatile("" : "+v"(oxv2), "+v"(oyv2), "+v"(cyv2));
      // Owner map in HBM: the entry is requested TWO chunks ahead (o_raw), so that the gather of the source
      // intensity one chunk ahead starts from an index that has already arrived instead of stalling on it; entries
      // of earlier iterations fail the tag comparison, so nothing is written back (a store per chunk would hold up
      // every later load of the wave: the memory counter retires in order).
      int o_raw = -1;
      auto owner_request = [&](int kk) {       // (entries below n_lds are in LDS: nothing to request)
        o_raw = (kk >= n_lds && kk < n) ? __hip_atomic_load(&g_owner[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
      };
      if (!OWNER_LDS) owner_request(k);
      // Owner map in LDS: the entry is read a chunk EARLIER than the rest of the chunk's operands (o_ahead), so that the
      // gather of the source intensity starts from an index that has already arrived: read in the same fetch it made
      // every wave wait out an LDS round trip (s_waitcnt lgkmcnt) in front of the five loads of every chunk.
      int o_ahead = -1;
      auto fetch = [&](int kk, auto parked_tag, const bool another_behind) {   // parked_tag: the chunk's depth was parked in
        o_n = -1;                                                              // LDS by pass 1; another_behind: wave-uniform
        if (OWNER_LDS) {
          o_n = o_ahead;
          o_ahead = -1;
          if (another_behind) {                     // (walking backwards: not the chunk in front of the first)
            const int k2 = kk + k_step;
            o_ahead = s_owner[k2];                  // (past the image: the padding, -1)
            s_owner[k2] = -1;                       // ready for the next iteration
          }
        } else {
          if (kk < n_lds) {                         // n_lds is a multiple of 64: the whole chunk is on one side
            o_n = s_owner[kk];
            s_owner[kk] = -1;
          } else {
            o_n = (o_raw & ~OWNER_INDEX_MASK) == owner_tag ? (o_raw & OWNER_INDEX_MASK) : -1;
          }
          owner_request(kk + NW * WAVE);
        }
        if constexpr (decltype(parked_tag)::value) pz_n = s_i0[max(kk, 0)];      // (kk < 0: the chunk in front of the first, unused)
        else pz_n = plane_load<TD>(rS, kk, oD);
        gx_n = plane_load<TI>(rT, kk, oGX);             // gradient at the SOURCE index  :346-347
        gy_n = plane_load<TI>(rT, kk, oGY);
        i1_n = plane_load<TI>(rT, kk, oI);             // :309
        if (SRC_LDS) { if (o_n >= 0) i0_n = s_i0[o_n]; }
        else i0_n = plane_load<TI>(rS, o_n, oI);       // :308 of the owning source pixel (o = -1: offset past the plane -> 0)
      };
      if (OWNER_LDS && my_chunks > 0) {
        o_ahead = s_owner[k];
        s_owner[k] = -1;
      }
      if (PARK && wave + first * NW < depth_chunks) fetch(k, std::true_type{}, my_chunks >= 2);
      else fetch(k, std::false_type{}, my_chunks >= 2);
      // `behind`: chunks of this wave that follow the one the body's fetch is for (wave-uniform)
      auto chunk_body = [&](const int chunk, auto next_parked_tag, const int behind) {
        const int o = o_n;
        const double pz = pz_n, gxi = gx_n, gyi = gy_n, pixel2 = i1_n, pixel1 = i0_n;
        fetch(k + k_step, next_parked_tag, behind > 0);
        const unsigned long long mbits =
            MASK_REG ? (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(inb_hi, j) << 32) |
                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane(inb_lo, j))
                     : s_mask[chunk];
        if (__builtin_amdgcn_inverse_ballot_w64(mbits)) {                 // the ballot becomes the exec mask
          const double res = (o >= 0) ? (pixel2 - pixel1) : 0.0;          // :358
          PHOVO_ROWCOL_HERE
          const double px = fma(cd, ifx, oxv2) * pz;
          const double py = fma(rd, ify, oyv2) * pz;

          // The 2x6 warp Jacobian (:312-342) contracted with the image gradient (:348), with the common
          // factors pulled out and the reference's temps folded by exact algebraic identities:
          //   temp25 = 1/Zd, Zd = z+py*temp1+pz*temp2-px*temp3, temp26 = temp25^2,
          //   Au = (pz*temp4+py*temp5+px*temp11)   [temp11 = temp15 + x carries the reference's bug, :253]
          //   Bv = (py*temp6+pz*temp9+px*temp14+y)
          //   Cm = (-py*temp16-pz*temp17-px*temp24), Dm = (py*temp22-pz*temp23)
          //   (py*temp7+pz*temp8-px*temp14) = y - Bv          since temp7 = -temp6, temp8 = -temp9
          //   (pz*temp4+py*temp5+px*temp15) = Au - px*x       since temp15 = temp11 - x
          //   (py*temp10+pz*temp12-px*temp13) = cos(yaw)*(Zd - z), (py*temp18+pz*temp19-px*temp20) = sin(yaw)*(Zd - z)
          //   J0 = gx*fx*temp25, J1 = gy*fy*temp25, J2 = -(J0*Au + J1*Bv)*temp25,
          //   J3 = J0*(y - Bv) + J1*(Au - px*x),
          //   J4 = (J0*cos(yaw) + J1*sin(yaw))*(Zd - z) + Cm*J2,
          //   J5 = J0*(py*temp4+pz*temp21) + J1*(pz*temp7+py*temp9) + Dm*J2.
          const double Zr = py * t1 + pz * t2 - px * t3;                  // Zd - z
          const double t25 = fast_rcp<1>(cz + Zr);                        // :313  (one Newton step: 2^-48, a Jacobian factor)
          const double Au = pz * t4 + py * t5 + px * t11;
          const double Bv = fma(py, t6, fma(pz, t9, fma(px, t14, cyv2)));
          const double Cm = -py * t16 - pz * t17 - px * t24;
          const double Dm = py * t2 - pz * t1;
          double J[6];
          J[0] = (gxi * fx) * t25;                                        // :317
          J[1] = (gyi * fy) * t25;                                        // :322
          J[2] = -(J[0] * Au + J[1] * Bv) * t25;                          // :325-326
          J[3] = J[0] * (cyy - Bv) + J[1] * (Au - px * cx);               // :329-330
          J[4] = (J[0] * cosy + J[1] * siny) * Zr + Cm * J[2];            // :333-336
          J[5] = J[0] * (py * t4 + pz * t21) + J[1] * (pz * t7 + py * t9) + Dm * J[2];                   // :339-342

          double Jw[6];
#pragma unroll
          for (int a = 0; a < 6; a++) Jw[a] = J[a];
          if (HUBER) {         // extension, not in the reference: IRLS weight of the Huber loss
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
              acc[q] = fma(Jw[a], J[b], acc[q]);                                        // J^T (W) J  :540
              q++;
            }
          }
#pragma unroll
          for (int a = 0; a < 6; a++) acc[21 + a] = fma(Jw[a], res, acc[21 + a]);        // J^T (W) r  :538
        }
        k += k_step;
        j += REV ? -1 : 1;
        kd += REV ? -kd_step : kd_step;
      };
      // two chunks per trip by hand, as in pass 1
      if (REV) {
        // positions c_hi .. c_lo of this wave's chunks, downwards; `tag`: is the chunk BEHIND each of them parked?
        auto run = [&](int c_hi, int c_lo, auto tag) {
          for (int c = c_hi; c >= c_lo; c -= 2) {
            chunk_body(wave + c * NW, tag, c - 1);           // (its fetch is for position c - 1: c - 1 positions follow that)
            if (c > c_lo) chunk_body(wave + (c - 1) * NW, tag, c - 2);
          }
        };
        if (PARK) {
          // position c's successor is chunk wave + (c - 1) * NW: parked iff that is below depth_chunks (position 0's lies
          // in front of the image: read from the parked block too, clamped, unused)
          // (no chunk of this wave parked: -1, everything in the first run)
          const int c_split = depth_chunks > wave ? min(my_chunks - 1, (depth_chunks - 1 - wave) / NW + 1) : -1;
          run(my_chunks - 1, c_split + 1, std::false_type{});
          run(min(my_chunks - 1, c_split), 0, std::true_type{});
        } else {
          run(my_chunks - 1, 0, std::false_type{});
        }
      } else {
        for (int chunk = wave; chunk < A.n_chunks; chunk += 2 * NW) {
          chunk_body(chunk, std::false_type{}, 0);
          if (chunk + NW < A.n_chunks) chunk_body(chunk + NW, std::false_type{}, 0);
        }
      }
    };
    if (huber_on) pass2(std::true_type{}); else pass2(std::false_type{});
    PHOVO_STAMP(2)
    // The number of Jacobian rows filled rides through the reduction in a spare slot.  MASK_REG: lane j still holds chunk
    // j's ballot -- its popcount is that chunk's rows, and the sum over lanes and waves is the pair's (three vector
    // instructions per iteration instead of two scalar ones per chunk).
    acc[RED_VALID] = MASK_REG ? (double)(__popc((unsigned)inb_lo) + __popc((unsigned)inb_hi)) : (lane == 0 ? (double)n_rows : 0.0);


    // ---- wave-level transposed butterfly: 32 shuffles, lane l ends with value index idx(l) ----
    reduce_stage_swap<32, false>(acc);
    reduce_stage_swap<16, true>(acc);
    reduce_stage<8, 4>(acc, lane, 8);
    reduce_stage<4, 4>(acc, lane, 4);
    reduce_stage<2, 4>(acc, lane, 2);
    {
      const double total = acc[0] + __shfl_xor(acc[0], 1, WAVE);
      // (the lane is made opaque here: the row address below is loop-invariant, and hoisted out of the iteration loop it
      // does not survive pass 2's register pressure -- it came back as a scratch reload behind an s_waitcnt vmcnt(0),
      // in every wave, right in front of the barrier wave 0's solve waits at; five integer instructions instead)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int idx = ((ln >> 5) & 1) * 16 + ((ln >> 4) & 1) * 8 + ((ln >> 3) & 1) * 4 + ((ln >> 2) & 1) * 2 + ((ln >> 1) & 1);
      if ((ln & 1) == 0) s_red[wave * NRED + idx] = total;
    }
    PHOVO_STAMP(3)
    __syncthreads();
    PHOVO_STAMP(4)

    // ---- wave 0: cross-wave sum (fixed order), solve, update, terminate ------------------------
    if (wave == 0) {
      // lane l sums value (l & 31) over half of the waves, the halves meet in one shuffle
      double v = 0.0;
      if constexpr (NW == 1) {
        v = s_red[lane & (NRED - 1)];                     // a single wave: its own row is the total
      } else {
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

      last_valid = (int)__shfl(v, RED_VALID, WAVE);
#ifdef PHOVO_PHASE_STAMPS
      asm volatile("" :: "v"(h[0]), "v"(g[5]));
      const unsigned long long solve_t0 = wall_clock64();
#endif
      double step[6];
      solve6_ldlt(h, g, step);
#ifdef PHOVO_PHASE_STAMPS
      asm volatile("" :: "v"(step[0]), "v"(step[5]));
      const unsigned long long solve_t1 = wall_clock64();
      solve_sum[0] += solve_t0 - stamp_last;       // cross-wave sum and broadcasts
      solve_sum[1] += solve_t1 - solve_t0;         // LDL^T
#endif
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
#ifdef PHOVO_PHASE_STAMPS
      asm volatile("" :: "v"(gnorm));
      const unsigned long long solve_t2 = wall_clock64();
      solve_sum[2] += solve_t2 - solve_t1;         // update, norm, termination
#endif
      if (!done) write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], s_cst, lane);   // wave-uniform branch
#ifdef PHOVO_PHASE_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      solve_sum[3] += wall_clock64() - solve_t2;   // sincos and pose constants
#endif
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) s_state[i] = st[i];
        s_ctl[CTL_DONE] = done ? 1 : 0;
        if (!finite) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_NONFINITE;
        if (last_valid < 6) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_RANK_DEFICIENT;
      }
      last_gnorm = gnorm;
    }
    __syncthreads();
    PHOVO_STAMP(5)
    iteration++;
    if (s_ctl[CTL_DONE]) break;
  }
#ifdef PHOVO_PHASE_STAMPS
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 1 || wave == NW / 2 || wave == NW - 1))
    printf("stamps T=%d level n=%d it=%d wave %2d: pass1 %llu barrier %llu pass2 %llu butterfly %llu barrier %llu solve/wait %llu prologue %llu (10 ns ticks)\n",
           T, n, iteration, wave, stamp_sum[0], stamp_sum[1], stamp_sum[2], stamp_sum[3], stamp_sum[4], stamp_sum[5], stamp_prologue);
  if (blockIdx.x == 0 && lane == 0 && wave == 0)
    printf("stamps T=%d level n=%d it=%d inside wave 0's serial section: row sums %llu  LDL^T %llu  update+norm %llu  pose constants %llu\n",
           T, n, iteration, solve_sum[0], solve_sum[1], solve_sum[2], solve_sum[3]);
#endif
#undef PHOVO_STAMP
#undef PHOVO_ROWCOL_BEGIN
#undef PHOVO_ROWCOL_HERE
#undef PHOVO_ROWCOL_NEXT
}


// LDS carve-up of a workgroup of T threads whose largest level has n_max pixels in n_chunks_max chunks.
template <int T, bool SRC_LDS, bool OWNER_LDS, bool MASK_REG>
__device__ __forceinline__ LevelLds carve_lds(unsigned char *lds_raw, int n_max, int n_chunks_max)
{
  constexpr int NW = T / WAVE;
  LevelLds L;
  L.cst = reinterpret_cast<double *>(lds_raw);                            // [32]
  L.state = L.cst + 32;                                                   // [8]
  L.red = L.state + 8;                                                    // [NW][NRED]
  L.ctl = reinterpret_cast<int *>(L.red + NW * NRED);                     // [CTL_COUNT]
  L.mask = reinterpret_cast<unsigned long long *>(L.ctl + CTL_COUNT);     // [n_chunks] (!MASK_REG)
  unsigned char *p = reinterpret_cast<unsigned char *>(L.mask + (MASK_REG ? 0 : n_chunks_max));
  L.owner = reinterpret_cast<int *>(p);                                   // [n] (OWNER_LDS) or [n_lds] (owner map in HBM)
  if (OWNER_LDS) p += sizeof(int) * (size_t)owner_lds_entries(n_max, T);
  L.i0 = reinterpret_cast<double *>(p);                                   // [n]      (SRC_LDS)
  return L;
}

// One level per launch.  WPS = waves per SIMD the register allocator must leave room for (2 workgroups of 512 threads
// per CU <=> 4).
// Work queue: the grid has as many workgroups as fit the chip; each draws pair after pair from an atomic counter.
// Pairs stop after data-dependent iteration counts and the hardware deals blocks to the 8 XCDs round-robin, so a
// one-block-per-pair grid leaves whole XCDs idle while another one still works through its long pairs.
// Thread 0 draws the next index in the same block that writes the finished pair back, so that the only thing between
// the iteration loop and the barrier at the head of the work loop is one if-block (two adjacent `if (tid == 0)`
// blocks, one either side of the back edge, were threaded together by the compiler into a loop that reached that
// barrier with thread 0 parked: a hang).
// (draw_pair_any: with handover_in the pairs come from the list the sliding-window kernel left behind)
// The owner map in LDS is wiped ONCE per workgroup: pass 2 resets every slot it reads, and it reads all of them.
// It is padded to whole chunks plus one round of the workgroup (owner_lds_entries): pass 2 fetches the owner a chunk
// ahead without asking whether that chunk still exists -- the padding reads -1, "nobody", and pass 1 never writes there.
template <int T, int WPS, bool SRC_LDS, bool OWNER_LDS, bool MASK_REG, typename TI, typename TD, bool PARK = false>
__global__ __launch_bounds__(T, WPS) void gn_level_kernel(const GNLevelArgs A)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const LevelLds L = carve_lds<T, SRC_LDS, OWNER_LDS, MASK_REG>(lds_raw, A.n, A.g_mask ? 0 : A.n_chunks);
  int *const s_ctl = L.ctl;
  const int tid = threadIdx.x;
  for (int k = tid; k < (OWNER_LDS ? owner_lds_entries(A.n, T) : A.n_lds); k += T) L.owner[k] = -1;
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair_any(A);
  for (;;) {
    // The ticket thread 0 has just stored must have LEFT its LDS queue before any wave is released: the compiler omits
    // the wait in front of this one barrier (it relies on LDS operations being ordered across waves), and on the GPU
    // about one wave in 10^5 then read the previous ticket and aligned the wrong pair's pixels into this pair's sums --
    // or, on the last round, would never have left the loop.  Found by test_work_queue_results_do_not_depend_on_...
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
    if (pair >= A.n_pairs) break;           // uniform: every wave of the workgroup leaves together

    int iteration = 0;                // continuing a pair the sliding-window kernel handed over: its completed iterations count
    if (A.handover_in) iteration = __builtin_amdgcn_readfirstlane(A.reports[pair].iterations[A.level]);
    double last_gnorm = 0.0;
    int last_valid = 0;
    PHOVO_TIMELINE_BEGIN
    level_body<T, SRC_LDS, OWNER_LDS, MASK_REG, TI, TD, PARK>(A, L, pair, false, iteration, last_gnorm, last_valid);

    // ---- epilogue: state and report back to HBM -------------------------------------------------
    if (tid == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) A.states[(size_t)pair * 6 + j] = L.state[j];
      if (A.reports) {
        A.reports[pair].iterations[A.level] = iteration;
        A.reports[pair].gradient_norm = last_gnorm;
        A.reports[pair].valid_pixels[A.level] = last_valid;
        // (a flag is rare: an atomic without return instead of a read-modify-write that thread 0 would wait for)
        const uint32_t new_flags = (uint32_t)s_ctl[CTL_FLAGS] | (A.handover_in ? A.takeover_flag : 0u);
        if (new_flags) atomicOr(&A.reports[pair].flags, new_flags);
        PHOVO_TIMELINE_END(A.reports[pair])
      }
      s_ctl[CTL_PAIR] = draw_pair_any(A);
    }
  }   // next pair
}

// SEVERAL consecutive levels per launch: a workgroup draws a pair and runs it through F.n_levels levels, coarse to fine,
// back to back -- the loop of Optimize() (:502-563) as the reference has it, per pair over levels.  The state stays in
// LDS between levels; iteration counts and valid-pixel counts go to the pair's report level by level.  Why: with the
// shipped thresholds a pair stops after a data-dependent number of iterations (:376-392), and one launch per level makes
// every level boundary a boundary for the whole batch -- each persistent launch drains at falling occupancy and the few
// pairs that run to max_num_iterations finish alone, once per level.  Here a long pair of one level runs beside other
// pairs' work of any level, and the batch has ONE drain.  All levels take the geometry of the largest one (T threads,
// owner map in LDS sized for F.n_max pixels, register masks): per level the arithmetic and its order are those of
// gn_level_kernel<T, ...> with the same T, so the two paths are bit-identical (tests/test_gpu_fused.py).
template <int T, int WPS, typename TI, typename TD>
__global__ __launch_bounds__(T, WPS) void gn_fused_kernel(const GNFusedArgs F)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const LevelLds L = carve_lds<T, false, true, true>(lds_raw, F.n_max, 0);
  int *const s_ctl = L.ctl;
  const int tid = threadIdx.x;
  for (int k = tid; k < owner_lds_entries(F.n_max, T); k += T) L.owner[k] = -1;
  if (tid == 0) s_ctl[CTL_PAIR] = draw_pair(F.work_counter, F.n_queues, F.n_pairs);
  for (;;) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // see gn_level_kernel
    __syncthreads();
    const int pair = __builtin_amdgcn_readfirstlane(s_ctl[CTL_PAIR]);
    if (pair >= F.n_pairs) break;           // uniform: every wave of the workgroup leaves together

    double last_gnorm = 0.0;
    PHOVO_TIMELINE_BEGIN
    for (int li = 0; li < F.n_levels; li++) {               // coarse to fine  :502-503
      const GNLevelArgs &A = F.lv[li];
      int iteration = 0, last_valid = 0;
      // A level smaller than the largest leaves the tail of the owner map unused: the depth of its leading chunks is parked
      // there between the passes (level_body, PARK; gn_launch_fused sizes A.depth_lds_chunks), and the tail gets its -1 back
      // before the next level may need it as owner map (that level's prologue has the barriers).
      LevelLds Ll = L;
      int *const tail = L.owner + owner_lds_entries(A.n, T);
      Ll.i0 = reinterpret_cast<double *>(tail);
      level_body<T, false, true, true, TI, TD, true>(A, Ll, pair, li > 0, iteration, last_gnorm, last_valid);
      for (int k = tid; k < A.depth_lds_chunks * (int)(WAVE * sizeof(double) / sizeof(int)); k += T) tail[k] = -1;
      if (tid == 0 && A.reports) {
        A.reports[pair].iterations[A.level] = iteration;
        A.reports[pair].valid_pixels[A.level] = last_valid;
      }
    }

    if (tid == 0) {
      const GNLevelArgs &A = F.lv[0];
#pragma unroll
      for (int j = 0; j < 6; j++) A.states[(size_t)pair * 6 + j] = L.state[j];
      if (A.reports) {
        A.reports[pair].gradient_norm = last_gnorm;
        const uint32_t new_flags = (uint32_t)s_ctl[CTL_FLAGS];
        if (new_flags) atomicOr(&A.reports[pair].flags, new_flags);
        PHOVO_TIMELINE_END(A.reports[pair])
      }
      s_ctl[CTL_PAIR] = draw_pair(F.work_counter, F.n_queues, F.n_pairs);
    }
  }   // next pair
}


constexpr size_t LDS_HALF = LDS_LIMIT / 2; // two workgroups per CU

static_assert((1 << OWNER_TAG_SHIFT) - 1 == OWNER_INDEX_MASK, "index mask and tag shift belong together");

// The instantiations that exist (each one is a separate kernel in the code object):
//   MID    512 threads, 2 workgroups/CU, owner map in LDS, register mask
//   WIDE   1024 threads, 1 workgroup/CU, owner map in LDS (80..160 KB), register mask
//   HUGE   1024 threads, owner map in global memory, ballot mask in LDS
//   TINY   256 threads, 4 workgroups/CU, everything in LDS (levels of <= 2048 pixels)
//   QUAD   256 threads, 4 workgroups/CU, owner map in LDS, source intensity gathered from L2
enum Variant { V_TINY = 0, V_MID, V_WIDE, V_HUGE, V_QUAD, V_SOLO };

// ... times the three plane storages (fp64 = reference-exact; fp32; fp16 images + fp32 depth).
//   SOLO   64 threads, 16 workgroups/CU: one wave per pair (levels of <= 2048 pixels in a throughput launch)
#define PHOVO_KERNEL_TINY(TI, TD)  gn_level_kernel<256, 4, true, true, true, TI, TD>
#define PHOVO_KERNEL_MID(TI, TD)   gn_level_kernel<512, 4, false, true, true, TI, TD, true>
#define PHOVO_KERNEL_WIDE(TI, TD)  gn_level_kernel<1024, 4, false, true, true, TI, TD, true>
#define PHOVO_KERNEL_HUGE(TI, TD)  gn_level_kernel<1024, 4, false, false, false, TI, TD>
#define PHOVO_KERNEL_QUAD(TI, TD)  gn_level_kernel<256, 4, false, true, true, TI, TD, true>
#define PHOVO_KERNEL_SOLO(TI, TD)  gn_level_kernel<64, 4, false, true, true, TI, TD, true>
#define PHOVO_KERNEL_FUSED(TI, TD) gn_fused_kernel<512, 4, TI, TD>

}  // namespace

bool gn_plan_level(int n, GNLaunchPlan *plan, int prefer_latency, bool bound_by_bytes)
{
  plan->owner_lds_entries = 0;
  plan->mask_in_hbm = false;
  plan->depth_lds_chunks = 0;
  const size_t n_chunks = (size_t)(n + WAVE - 1) / WAVE;
  auto owner_bytes = [n](int threads) { return sizeof(int) * (size_t)owner_lds_entries(n, threads); };
  // leftover LDS of a geometry with `per_cu` workgroups per CU -> leading chunks whose depth pass 1 parks there (PARK)
  auto park_depth = [&](size_t used, int per_cu) {
    used = (used + 7) & ~(size_t)7;
    const size_t room = LDS_LIMIT / (size_t)per_cu;
    const size_t chunks = room > used ? (room - used) / (sizeof(double) * WAVE) : 0;
    plan->depth_lds_chunks = (int)(chunks < n_chunks ? chunks : n_chunks);
    return used + sizeof(double) * WAVE * (size_t)plan->depth_lds_chunks;
  };
  const size_t src = sizeof(double) * (size_t)n;
  // Levels of <= 2048 pixels in a throughput launch: ONE WAVE per pair, 16 workgroups per CU.  An iteration of such a level
  // is short (40x30: 19 chunks) and a third of a 256-thread workgroup's time per iteration is wave 0's serial section
  // (solve, sincos, pose constants) with the other three waves waiting at the barrier; with one wave per pair nobody
  // waits -- the other 15 pairs of the CU fill the SIMDs meanwhile.
  if (n <= 2048 && !prefer_latency && !tuning_switch("PHOVO_GN_NO_SOLO") && n_chunks <= 64) {
    plan->variant = V_SOLO; plan->threads = 64; plan->wgs_per_cu = 16; plan->owner_in_lds = true; plan->source_in_lds = false;
    plan->lds_bytes = (int)park_depth(lds_fixed_bytes(64) + owner_bytes(64), 16);
    return true;
  }
  if (n <= 2048) {
    plan->variant = V_TINY; plan->threads = 256; plan->wgs_per_cu = 4; plan->owner_in_lds = true; plan->source_in_lds = true;
    plan->lds_bytes = (int)(lds_fixed_bytes(256) + owner_bytes(256) + src);
    return true;
  }
  const size_t f256 = lds_fixed_bytes(256);
  // Four 256-thread workgroups per CU beat two of 512 on levels this small (80x60: 4.31 vs 4.64 ms per
  // 2048 pairs x 50 iterations): while one workgroup's wave 0 solves, three others keep the SIMDs busy.
  // (one pair alone on a CU: 12.4 us per 80x60 iteration with 256 threads, 10.1 us with 512 -- prefer_latency)
  if (!tuning_switch("PHOVO_GN_NO_QUAD") && !prefer_latency && n_chunks <= 64 * 4 && f256 + owner_bytes(256) <= LDS_LIMIT / 4) {
    plan->variant = V_QUAD; plan->threads = 256; plan->wgs_per_cu = 4; plan->owner_in_lds = true; plan->source_in_lds = false;
    plan->lds_bytes = (int)park_depth(f256 + owner_bytes(256), 4);
    return true;
  }
  const size_t f512 = lds_fixed_bytes(512), f1024 = lds_fixed_bytes(1024);
  const bool reg512 = n_chunks <= 64 * 8, reg1024 = n_chunks <= 64 * 16;
  const bool fits_wide = reg1024 && f1024 + owner_bytes(1024) <= LDS_LIMIT;
  if (!tuning_switch("PHOVO_GN_FORCE_WIDE") && reg512 && f512 + owner_bytes(512) <= LDS_HALF) {
    plan->variant = V_MID; plan->threads = 512; plan->wgs_per_cu = 2; plan->owner_in_lds = true; plan->source_in_lds = false;
    plan->lds_bytes = (int)park_depth(f512 + owner_bytes(512), 2);
    // A level whose owner map fills a half of LDS (160x120: 77 of 80 KB) leaves two workgroups per CU no room to park
    // depth, while ONE workgroup of 1024 threads parks half the image next to the same map: 19.25 -> 18.98 ms per 512 pairs
    // x 50 iterations of 160x120 (profiles/r04_runs/park_depth_ab.txt).  Throughput launches on fp64 planes only: with the
    // narrow storages the level is bound by the vector unit, bytes saved buy nothing and one workgroup per CU costs 14 %.
    const bool mid_parks_little = (size_t)plan->depth_lds_chunks * 4 < n_chunks;
    if (!(mid_parks_little && fits_wide && !prefer_latency && bound_by_bytes && !tuning_switch("PHOVO_GN_NO_WIDE_PARK")))
      return true;
  }
  if (fits_wide) {
    plan->variant = V_WIDE; plan->threads = 1024; plan->wgs_per_cu = 1; plan->owner_in_lds = true; plan->source_in_lds = false;
    plan->lds_bytes = (int)park_depth(f1024 + owner_bytes(1024), 1);
    return true;
  }
  size_t mask = sizeof(unsigned long long) * n_chunks;
  if (n > OWNER_INDEX_MASK) return false;          // the tagged entries of the HBM owner map hold 21-bit indices
  if (f1024 + mask > LDS_LIMIT / 2) {              // the ballots would take most of LDS (or do not fit): global memory
    plan->mask_in_hbm = true;
    mask = 0;
  }
  // whatever LDS the ballot masks leave free holds the leading part of the owner map (whole 64-pixel chunks)
  const size_t spare = LDS_LIMIT - f1024 - mask;
  plan->owner_lds_entries = tuning_switch("PHOVO_GN_NO_OWNER_SPLIT") ? 0 : (int)((spare / sizeof(int)) / WAVE * WAVE);
  if (plan->owner_lds_entries > n) plan->owner_lds_entries = (n / WAVE) * WAVE;
  plan->variant = V_HUGE; plan->threads = 1024; plan->wgs_per_cu = 1; plan->owner_in_lds = false; plan->source_in_lds = false;
  plan->lds_bytes = (int)(f1024 + mask + sizeof(int) * (size_t)plan->owner_lds_entries);
  return true;
}

bool gn_level_fusable(int n)
{
  const size_t n_chunks = (size_t)(n + WAVE - 1) / WAVE;
  return n > 2048 && n_chunks <= 64 * 8 &&
         lds_fixed_bytes(512) + sizeof(int) * (size_t)owner_lds_entries(n, 512) <= LDS_HALF;
}

bool gn_plan_fused_geometry(int n, GNLaunchPlan *plan)
{
  if (!gn_level_fusable(n)) return false;
  plan->variant = V_MID; plan->threads = 512; plan->wgs_per_cu = 2; plan->owner_in_lds = true; plan->source_in_lds = false;
  plan->owner_lds_entries = 0; plan->mask_in_hbm = false; plan->depth_lds_chunks = 0;
  plan->lds_bytes = gn_fused_lds_bytes(n);
  return true;
}

int gn_fused_lds_bytes(int n_max)
{
  return (int)(lds_fixed_bytes(512) + sizeof(int) * (size_t)owner_lds_entries(n_max, 512));
}

template <typename TI, typename TD>
hipError_t prepare_storage()
{
  hipError_t e;
#define PHOVO_PREP(K)                                                                                      \
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&K(TI, TD)), hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)LDS_LIMIT);                                                                 \
  if (e != hipSuccess) return e;
  PHOVO_PREP(PHOVO_KERNEL_TINY)
  PHOVO_PREP(PHOVO_KERNEL_MID)
  PHOVO_PREP(PHOVO_KERNEL_WIDE)
  PHOVO_PREP(PHOVO_KERNEL_HUGE)
  PHOVO_PREP(PHOVO_KERNEL_QUAD)
  PHOVO_PREP(PHOVO_KERNEL_SOLO)
  PHOVO_PREP(PHOVO_KERNEL_FUSED)
#undef PHOVO_PREP
  return hipSuccess;
}

template <typename TI, typename TD>
hipError_t launch_storage(const GNLevelArgs &a, const GNLaunchPlan &plan, int n_blocks, hipStream_t stream)
{
  const size_t lds = (size_t)plan.lds_bytes;
  const dim3 grid((unsigned)n_blocks), block((unsigned)plan.threads);
  switch (plan.variant) {
    case V_TINY:  hipLaunchKernelGGL(PHOVO_KERNEL_TINY(TI, TD), grid, block, lds, stream, a); break;
    case V_MID:   hipLaunchKernelGGL(PHOVO_KERNEL_MID(TI, TD), grid, block, lds, stream, a); break;
    case V_WIDE:  hipLaunchKernelGGL(PHOVO_KERNEL_WIDE(TI, TD), grid, block, lds, stream, a); break;
    case V_HUGE:  hipLaunchKernelGGL(PHOVO_KERNEL_HUGE(TI, TD), grid, block, lds, stream, a); break;
    case V_QUAD:  hipLaunchKernelGGL(PHOVO_KERNEL_QUAD(TI, TD), grid, block, lds, stream, a); break;
    case V_SOLO:  hipLaunchKernelGGL(PHOVO_KERNEL_SOLO(TI, TD), grid, block, lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename TI, typename TD>
hipError_t launch_fused_storage(const GNFusedArgs &f, int n_blocks, hipStream_t stream)
{
  hipLaunchKernelGGL(PHOVO_KERNEL_FUSED(TI, TD), dim3((unsigned)n_blocks), dim3(512), (size_t)gn_fused_lds_bytes(f.n_max),
                     stream, f);
  return hipGetLastError();
}

hipError_t gn_prepare_kernels()
{
  hipError_t e;
  if ((e = prepare_storage<double, double>()) != hipSuccess) return e;
  if ((e = prepare_storage<float, float>()) != hipSuccess) return e;
  return prepare_storage<__half, float>();
}

hipError_t gn_launch_level(const GNLevelArgs &a, const GNLaunchPlan &plan, int storage, int cu_count,
                           hipStream_t stream)
{
  if (a.n_pairs <= 0) return hipSuccess;
  const int slots = cu_count * plan.wgs_per_cu;                                // what the chip holds at once
  const int n_pairs = a.n_pairs < slots ? a.n_pairs : slots;                   // persistent grid size
  switch (storage) {
    case PHOVO_STORAGE_F64: return launch_storage<double, double>(a, plan, n_pairs, stream);
    case PHOVO_STORAGE_F32: return launch_storage<float, float>(a, plan, n_pairs, stream);
    case PHOVO_STORAGE_F16: return launch_storage<__half, float>(a, plan, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t gn_launch_fused(const GNFusedArgs &f, int storage, int cu_count, hipStream_t stream)
{
  if (f.n_pairs <= 0 || f.n_levels <= 0) return hipSuccess;
  if (f.n_levels > GN_MAX_FUSED_LEVELS) return hipErrorInvalidValue;
  for (int i = 0; i < f.n_levels; i++)
    if (!gn_level_fusable(f.lv[i].n) || f.lv[i].n > f.n_max) return hipErrorInvalidValue;
  const int slots = cu_count * 2;                                              // two 512-thread workgroups per CU
  const int n_blocks = f.n_pairs < slots ? f.n_pairs : slots;
  // depth parked in the part of the owner map a level does not use (gn_fused_kernel)
  GNFusedArgs g = f;
  for (int i = 0; i < g.n_levels; i++) {
    const int spare = owner_lds_entries(g.n_max, 512) - owner_lds_entries(g.lv[i].n, 512);        // int32 entries
    const int chunks = (int)((size_t)spare * sizeof(int) / (sizeof(double) * WAVE));
    g.lv[i].depth_lds_chunks = chunks < g.lv[i].n_chunks ? chunks : g.lv[i].n_chunks;
  }
  switch (storage) {
    case PHOVO_STORAGE_F64: return launch_fused_storage<double, double>(g, n_blocks, stream);
    case PHOVO_STORAGE_F32: return launch_fused_storage<float, float>(g, n_blocks, stream);
    case PHOVO_STORAGE_F16: return launch_fused_storage<__half, float>(g, n_blocks, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace phovo_hip
