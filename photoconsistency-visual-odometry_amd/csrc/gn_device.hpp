// Device-side building blocks shared by the Gauss-Newton kernels (gn_kernels.hip, gn_wide_kernels.hip):
// pose constants, wave / workgroup reductions, the 6x6 solve, typed plane loads.  Everything is
// __forceinline__ in an anonymous namespace: each translation unit gets its own copy.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

constexpr int WAVE = 64;
constexpr size_t LDS_LIMIT = 160 * 1024;   // MI355X: 160 KiB per CU, one workgroup may take all of it
constexpr int NRED = 32;     // padded for the butterfly
constexpr int RED_VALID = 27;   // slot behind the 21 + 6 sums: the number of Jacobian rows filled (lane 0 of every wave puts
                                // its wave's count there as a double -- exact -- and the reduction adds them up for free)

// Indices into the pose-constant block in LDS.
enum {
  C_X = 0, C_Y, C_Z, C_R01, C_R02, C_R11, C_R12,
  C_T1, C_T2, C_T3, C_T4, C_T5, C_T6, C_T8, C_T11, C_T14, C_T15,
  C_T16, C_T17, C_T24, C_CY, C_SY, C_COUNT
};

// Control words in LDS.
enum { CTL_DONE = 0, CTL_FLAGS = 1, CTL_PAIR = 2, CTL_OOW = 3, CTL_COUNT = 4 };      // CTL_OOW: sliding-window kernel only

// LDS every level kernel needs beside its maps: pose constants [32], state [8], one row of NRED wave sums per wave, control words
__host__ __device__ constexpr size_t lds_fixed_bytes(int threads)
{
  return sizeof(double) * (32 + 8 + (size_t)(threads / WAVE) * NRED) + sizeof(int) * CTL_COUNT;
}

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
// Rt(1,0) = temp14, Rt(2,0) = -temp3, Rt(2,1) = temp1, Rt(2,2) = temp2 likewise.  temp10, temp12, temp13 are
// temp1, temp2, temp3 times cos(yaw) and temp18, temp19, temp20 the same times sin(yaw): the kernel applies
// those two factors to the sum instead (see pass 2), so only cos(yaw) and sin(yaw) are stored.
__device__ __forceinline__ void write_pose_constants(double x, double y, double z, double yaw, double pitch,
                                                     double roll, double *cst, int lane)
{
  // Lanes 0, 1, 2 take yaw, pitch, roll: ONE sincos issue instead of three dependent ones; the six
  // results are then broadcast (the whole wave executes this with a wave-uniform state).
  const double ang = (lane == 1) ? pitch : ((lane == 2) ? roll : yaw);
  double sn, cs;
  // Euler angles of a frame-to-frame motion are small: up to pi/4 the two minimax polynomials of fdlibm's __kernel_sin /
  // __kernel_cos (errors below one ulp, as the library's) need no argument reduction; anything larger takes the device
  // library's sincos (wave-uniform branch: lanes 3.. carry yaw).  The library call alone cost 4.7 k cycles per iteration
  // for a reason that has nothing to do with arithmetic: its polynomial coefficients are VGPR immediates that the
  // compiler hoists to the top of the kernel, cannot keep under the 128-register cap and SPILLS -- nine dependent
  // scratch reloads, each a full trip to memory behind an s_waitcnt vmcnt(0), in the middle of wave 0's serial section
  // (and nine more in every pair's prologue).  Here the coefficients are scalar operands (opaque to the hoisting).
  const bool small_angles = __builtin_amdgcn_ballot_w64(!(fabs(ang) <= 0.78539816339744828)) == 0;
  if (small_angles) {
    auto k = [](double v) { asm volatile("" : "+s"(v)); return v; };       // a coefficient as an SGPR pair
    const double z = ang * ang;
    double ps = fma(z, k(1.58969099521155010221e-10), k(-2.50507602534068634195e-08));
    double pc = fma(z, k(-1.13596475577881948265e-11), k(2.08757232129817482790e-09));
    ps = fma(z, ps, k(2.75573137070700676789e-06));
    pc = fma(z, pc, k(-2.75573143513906633035e-07));
    ps = fma(z, ps, k(-1.98412698298579493134e-04));
    pc = fma(z, pc, k(2.48015872894767294178e-05));
    ps = fma(z, ps, k(8.33333333332248946124e-03));
    pc = fma(z, pc, k(-1.38888888888741095749e-03));
    ps = fma(z, ps, k(-1.66666666666666324348e-01));
    pc = fma(z, pc, k(4.16666666666666019037e-02));
    sn = fma(ang * z, ps, ang);                         // x + x^3 (S1 + z (S2 + ...)): 0.63 ulp at worst
    // cos = 1 - z/2 + z^2 (C1 + z (C2 + ...)), summed as fdlibm does: from 0.3 upwards a quarter of |x| (truncated to
    // its high word, so that 1 - qx and z/2 - qx are exact) is taken out of both big terms first: 0.77 ulp at worst
    // (1.2 ulp summed naively)
    const double ax = fabs(ang);
    const double q4 = __hiloint2double(__double2hiint(ax) - 0x00200000, 0);
    const double qx = ax < 0.3 ? 0.0 : (ax > 0.78125 ? 0.28125 : q4);
    cs = (1.0 - qx) - fma(-z, z * pc, 0.5 * z - qx);
  } else {
    sincos(ang, &sn, &cs);
  }
  const double sy = __shfl(sn, 0, WAVE), cy = __shfl(cs, 0, WAVE);
  const double sp = __shfl(sn, 1, WAVE), cp = __shfl(cs, 1, WAVE);
  const double sr = __shfl(sn, 2, WAVE), cr = __shfl(cs, 2, WAVE);
  if (lane != 0) return;
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
  cst[C_T11] = cp * cy + x;          // the reference's bug, kept (:253)
  cst[C_T14] = cp * sy;
  cst[C_T15] = cp * cy;
  cst[C_T16] = sp * sr;
  cst[C_T17] = sp * cr;
  cst[C_T24] = cp;
  cst[C_CY] = cy;
  cst[C_SY] = sy;
}

// One butterfly stage of the transposed wave reduction: N values in, N/2 out.  The stage is issued in
// groups of G exchanges with a scheduling fence between groups: left alone, the scheduler hoists all
// N/2 select pairs in front of the shuffles and the live range grows by 4 VGPRs per exchange
// (179 VGPRs and spills under the 128-register budget of a 1024-thread workgroup).
template <int N, int G>
__device__ __forceinline__ void reduce_stage(double (&v)[NRED], int lane, int dist)
{
  const bool up = (lane & dist) != 0;
#pragma unroll
  for (int base = 0; base < N / 2; base += G) {
#pragma unroll
    for (int i = base; i < base + G && i < N / 2; i++) {
      const double send = up ? v[i] : v[i + N / 2];
      const double keep = up ? v[i + N / 2] : v[i];
      v[i] = keep + __shfl_xor(send, dist, WAVE);
    }
    if (N / 2 > G) __builtin_amdgcn_sched_barrier(0);
  }
}

// The two widest butterfly stages without selects or LDS-crossbar traffic: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the upper half (odd 16-lane rows) of one register with
// the lower half (even rows) of another, so for the pair (v[i], v[i+N/2]) one swap per dword leaves
//   newA = { own v[i] in the low lanes,  partner's v[i+N/2] in the high lanes }
//   newB = { partner's v[i] in the low lanes,  own v[i+N/2] in the high lanes }
// and newA + newB is exactly "keep + received" of reduce_stage with the same lane -> index map.
template <int N, bool ROW16>
__device__ __forceinline__ void reduce_stage_swap(double (&v)[NRED])
{
#pragma unroll
  for (int i = 0; i < N / 2; i++) {
    const unsigned alo = (unsigned)__double2loint(v[i]), ahi = (unsigned)__double2hiint(v[i]);
    const unsigned blo = (unsigned)__double2loint(v[i + N / 2]), bhi = (unsigned)__double2hiint(v[i + N / 2]);
    const auto lo = ROW16 ? __builtin_amdgcn_permlane16_swap(alo, blo, false, false)
                          : __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = ROW16 ? __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false)
                          : __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
}

// 6x6 solve of the normal equations H x = g by an unpivoted LDL^T factorisation: H = J^T J is
// symmetric positive semi-definite, for which LDL^T is backward stable, needs 6 reciprocals instead of
// the 21 divisions + 15 row swaps of pivoted elimination and only the 21 upper-triangular sums.
// (The reference forms H^-1 with Eigen's PartialPivLU, ...Analytic.h:540; both are accurate to
// cond(H)*eps, far inside the 1e-5 pose bar -- tests/test_gpu_parity.py holds 1e-9.)
// h is the upper triangle, row-major: h[idx(i,j)], i <= j.  Fully unrolled: every index is a
// compile-time constant, nothing goes to scratch.
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

__device__ __forceinline__ void solve6_ldlt(const double (&h)[21], const double (&g)[6], double (&x)[6])
{
  double L[6][6];       // strictly lower part used
  double Ld[6][6];      // L[i][k] * d[k]
  double inv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = h[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) dj = fma(-L[j][k], Ld[j][k], dj);
    inv[j] = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = h[tri(j, i)];
#pragma unroll
      for (int k = 0; k < j; k++) t = fma(-L[i][k], Ld[j][k], t);
      Ld[i][j] = t;
      L[i][j] = t * inv[j];
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double t = g[i];
#pragma unroll
    for (int k = 0; k < i; k++) t = fma(-L[i][k], y[k], t);
    y[i] = t;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double t = y[i] * inv[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t = fma(-L[k][i], x[k], t);
    x[i] = t;
  }
}

// Plane loads go through buffer descriptors: the four planes of a chunk share ONE 32-bit byte offset
// (8*k) in a VGPR while the bases sit in SGPRs, and a read past the plane returns 0 instead of needing
// a branch (raw buffer, num_records = plane bytes).  Flat loads cost a 64-bit VGPR address per plane.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const unsigned char *p, int n)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(p), 0, n * (int)sizeof(T), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const unsigned char *frame, size_t frame_bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(frame), 0, (int)frame_bytes, 0x00020000);
}
// Element `idx` of a plane stored as T, widened to fp64 (all arithmetic stays fp64; narrower storage is the
// opt-in extension of include/phovo_hip.h, PHOVO_STORAGE_*).  idx = -1 or past the plane reads 0.
// soff: a wave-uniform byte offset (the instruction's scalar offset; it takes part in the range check): one descriptor
// per FRAME (frame_rsrc) serves every plane of it -- base = the frame, soff = the plane's offset in it.  Four SGPRs per
// frame plus one per plane instead of four per plane: the level kernels run out of scalar registers, and every spilled
// one costs a v_readlane per use in the pixel loops.
template <typename T>
__device__ __forceinline__ double plane_load(__amdgpu_buffer_rsrc_t r, int idx, int soff = 0);
template <>
__device__ __forceinline__ double plane_load<double>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, idx * 8, soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
template <>
__device__ __forceinline__ double plane_load<float>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  return (double)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, idx * 4, soff, 0));
}
template <>
__device__ __forceinline__ double plane_load<__half>(__amdgpu_buffer_rsrc_t r, int idx, int soff)
{
  const unsigned short bits = __builtin_amdgcn_raw_buffer_load_b16(r, idx * 2, soff, 0);
  return (double)__half2float(__ushort_as_half(bits));
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 1/x to within one ulp: v_rcp_f64 seeds two Newton steps.  The IEEE-exact division sequence is
// 11 instructions, this is 5; the half-ulp it gives up is far below the fp64 noise floor of the sums
// that follow (tests/test_gpu_parity.py holds the poses to 1e-9 against the oracle's exact divisions).
template <int STEPS = 2>
__device__ __forceinline__ double fast_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  if (STEPS >= 2) {
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
  }
  return r;
}

// Entries of an owner map kept in HBM: (iteration tag << 21) | source index.  atomicMax still picks the largest
// source index among the entries of the running iteration, and every entry of an earlier one is smaller than any of
// them, so the map needs no reset between iterations (it is wiped per pair and whenever the 1023 tags start over).
constexpr int OWNER_TAG_SHIFT = 21;
constexpr int OWNER_INDEX_MASK = (1 << OWNER_TAG_SHIFT) - 1;      // levels of up to 2 097 151 pixels
constexpr int OWNER_TAG_PERIOD = 1023;

// Entries of an owner map kept in LDS for a level of n pixels and a workgroup of `threads` threads: whole 64-pixel chunks
// plus one round of the workgroup's chunks (see gn_level_kernel: pass 2 reads a chunk ahead, unguarded).  Even.
__host__ __device__ __forceinline__ constexpr int owner_lds_entries(int n, int threads)
{
  return ((n + 63) & ~63) + threads;
}

// a * b + c as ONE instruction on the 24-bit integer multiplier: row * width + column of a pixel index.  b is wave-uniform.
// What it saves is instructions, not a slow multiply (v_mul_lo_u32 issues at full rate on gfx950,
// profiles/r04_runs/valu_rate_probe.txt): written as row * W + column the compiler distributes the shift of the byte
// address over both terms -- multiply, two shifts, add3 -- where this and one shift-add do.  (HIP has __mul24 but no mad.)
__device__ __forceinline__ int mad24_uniform_b(int a, int b, int c)
{
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}

// C round() -- half away from zero (...Analytic.h:297-298) -- for arguments > -0.5, which is all the bounds test lets
// through, in TWO instructions: floor(v + p) with p = 0.49999999999999994, the largest double below one half.  Exact, not
// approximately right (tests/test_oracle_properties.py checks every neighbour of every tie against exact arithmetic):
//   * a tie v = n + 0.5 gives n + 1 - 2^-54, which rounds to n + 1 (the next double below n + 1 is at least 2^-53 away for
//     n >= 1; for n = 0 the sum is halfway between 1 - 2^-53 and 1 and ties-to-even picks 1): round() says n + 1;
//   * the largest double below a tie, n + 0.5 - ulp, gives a sum that rounds to at most n + 1 - ulp: floor is n -- also
//     for 0.49999999999999994 itself, where the obvious floor(v + 0.5) goes wrong (the sum rounds up to 1.0);
//   * anything else is further than an ulp from a tie and the sum's rounding cannot carry it across an integer;
//   * (-0.5, 0) gives a sum in [0, 0.5): pixel 0, as round()'s -0.
// The trunc / fraction / compare / select / add form this replaces was five instructions per coordinate: 6 of the 50 of a
// pass-1 chunk (256 k -> 261 k alignments/s; the 5-level configuration, whose 40x30 level is vector-bound, +5 %).
__device__ __forceinline__ double round_half_up_from(double v)
{
  return floor(v + __hiloint2double(0x3fdfffff, (int)0xffffffff));
}

// Next pair of the work queue for the calling workgroup (one thread calls this), or n_pairs when everything is taken.
// With 8 queues, queue q serves the q-th contiguous eighth of the pair list and a workgroup starts with the queue of
// its XCD -- the hardware deals workgroups to the 8 XCDs round-robin, so blockIdx & 7 -- which keeps consecutive pairs
// of a sequence (they share a frame: the target of one is the source of the next) on one XCD's L2 at the same time;
// an empty queue sends the caller on to the next one.
// Every head has a cache line of its own (QUEUE_HEAD_STRIDE, round 3): the draws are device-scope atomics with a result
// and are served one after the other per line -- with all eight heads in ONE line the 8192 draws of a launch of short
// pairs stood in a single file (40x30 with the shipped thresholds, one iteration per pair: 1.05 -> 0.66 ms per launch;
// 80x60, every plane streamed once: 0.385 -> 0.32 ms).  Handing every workgroup its FIRST pair without an atomic on top
// of that was measured too and is not kept: nothing gained where the draws were the limit, -4 % with the shipped thresholds.
__device__ __forceinline__ int draw_pair(int *heads, int n_queues, int n_pairs)
{
  const int per = (n_pairs + n_queues - 1) / n_queues;
  int q = (int)(blockIdx.x & (unsigned)(n_queues - 1));                 // (n_queues is 1 or 8)
  for (int tries = 0; tries < n_queues; tries++) {
    const int first = q * per;
    const int size = first >= n_pairs ? 0 : (first + per > n_pairs ? n_pairs - first : per);
    if (size > 0) {
      const int t = atomicAdd(heads + q * QUEUE_HEAD_STRIDE, 1);
      if (t < size) return first + t;
    }
    q = (q + 1) & (n_queues - 1);
  }
  return n_pairs;
}

// Next pair for the calling workgroup: with GNLevelArgs::handover_in an index into the list the sliding-window launch of
// the level left behind (length at [n_pairs], final since that launch has ended) and the pair stored there; otherwise the
// plain queue.  n_pairs = nothing left.
__device__ __forceinline__ int draw_pair_any(const GNLevelArgs &A)
{
  if (!A.handover_in) return draw_pair(A.work_counter, A.n_queues, A.n_pairs);
  const int count = __hip_atomic_load(&A.handover_in[A.n_pairs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int i = atomicAdd(A.work_counter, 1);
  return i < count ? __hip_atomic_load(&A.handover_in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : A.n_pairs;
}

// The calling thread puts `pair` on handover_out (after its state and iteration count have been stored; the kernel
// boundary makes all of it visible to the next launch).
__device__ __forceinline__ void handover_append(const GNLevelArgs &A, int pair)
{
  const int slot = atomicAdd(&A.handover_out[A.n_pairs], 1);
  A.handover_out[slot] = pair;
}

// v_writelane_b32: lane `lane` (wave-uniform) of `old` becomes `value` (wave-uniform); the other lanes keep theirs.
// clang has builtins for readlane / readfirstlane but none for writelane, so the LLVM intrinsic is declared by name
// (the device libraries do the same for intrinsics without a builtin).  Inline assembly is not an option: the lane
// select travels in M0 and the assembler's operand check then counts two scalar sources.
extern "C" __device__ int phovo_llvm_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int writelane_b32(int old, int value, int lane)
{
  return phovo_llvm_writelane_i32(value, lane, old);
}

// (row, column) of a lane's pixel, carried along as integer-valued doubles (exact), NW*64 pixels per step:
// column += step_c, and on running past the row end column -= W, row += 1.  W, step_r and step_r + 1 are small
// integers, so as doubles their low words are zero and each select is one v_cndmask on the high word.
struct RowColStep {
  double step_c, w;
  int hi_w, hi_step_r, hi_step_r1;
};
__device__ __forceinline__ RowColStep make_rowcol_step(int step_r, int step_c, int W)
{
  RowColStep s;
  s.step_c = (double)step_c;
  s.w = (double)W;
  s.hi_w = __double2hiint((double)W);
  s.hi_step_r = __double2hiint((double)step_r);
  s.hi_step_r1 = __double2hiint((double)(step_r + 1));
  return s;
}
__device__ __forceinline__ void rowcol_advance(double &cd, double &rd, const RowColStep &s)
{
  cd += s.step_c;
  const bool wrap = cd >= s.w;
  cd -= __hiloint2double(wrap ? s.hi_w : 0, 0);
  rd += __hiloint2double(wrap ? s.hi_step_r1 : s.hi_step_r, 0);
}

// (row, column) of pixel k from k itself, all in fp64: row = trunc((k + 0.5) / W), column = k - row * W.  Exact: k < 2^21
// and W are integers, (k + 0.5) / W lies at least 0.5 / W >= 2.4e-7 away from every integer while the fma below is off by
// less than (k / W) * 2^-52 < 1e-12, so the truncation cannot land on the wrong side; the column is an exact integer fma.
struct RowColFromIndex {
  double inv_w, half_inv_w, w;
};
__device__ __forceinline__ RowColFromIndex make_rowcol_from_index(int W)
{
  RowColFromIndex m;
  m.w = (double)W;
  m.inv_w = uniform_f64(1.0 / m.w);
  m.half_inv_w = 0.5 * m.inv_w;
  return m;
}
__device__ __forceinline__ void rowcol_from_index(double kd, const RowColFromIndex &m, double &cd, double &rd)
{
  rd = trunc(fma(kd, m.inv_w, m.half_inv_w));
  cd = fma(-rd, m.w, kd);
}

// The same for a cursor that may stand in front of the image (negative index): floor instead of trunc.
__device__ __forceinline__ void rowcol_from_index_floor(double kd, const RowColFromIndex &m, double &cd, double &rd)
{
  rd = floor(fma(kd, m.inv_w, m.half_inv_w));
  cd = fma(-rd, m.w, kd);
}

// A wave's 27 sums (+ the row count) folded over its 64 lanes and written to row `row` of s_red (transposed butterfly).
__device__ __forceinline__ void reduce_wave_to_row(double (&acc)[NRED], int lane, int row, double *s_red)
{
  reduce_stage_swap<32, false>(acc);
  reduce_stage_swap<16, true>(acc);
  reduce_stage<8, 4>(acc, lane, 8);
  reduce_stage<4, 4>(acc, lane, 4);
  reduce_stage<2, 4>(acc, lane, 2);
  const double total = acc[0] + __shfl_xor(acc[0], 1, WAVE);
  const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 +
                  ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
  if ((lane & 1) == 0) s_red[row * NRED + idx] = total;
}

// Called by ONE wave behind the barrier that follows reduce_wave_to_row: sum of the NROWS rows, solve, update, termination
// (as gn_level_kernel's tail); the caller closes with a barrier and reads s_ctl[CTL_DONE].
template <int NROWS>
__device__ __forceinline__ void sum_rows_solve_update(int lane, const double *s_red, double *s_state, double *s_cst, int *s_ctl,
                                                      double lambda, int max_iter, double min_grad_norm,
                                                      int iteration, double &last_gnorm, int &last_valid)
{
  static_assert(NROWS % 2 == 0, "the rows are summed in two halves");
  double v = 0.0;
  {
    const int j = lane & (NRED - 1);
    const int w0 = (lane >> 5) * (NROWS / 2);
#pragma unroll
    for (int w2 = 0; w2 < NROWS / 2; w2++) v += s_red[(w0 + w2) * NRED + j];
    v += __shfl_xor(v, 32, WAVE);
  }
  double h[21], g[6];
#pragma unroll
  for (int q = 0; q < 21; q++) h[q] = __shfl(v, q, WAVE);
#pragma unroll
  for (int i = 0; i < 6; i++) g[i] = __shfl(v, 21 + i, WAVE);
  last_valid = (int)__shfl(v, RED_VALID, WAVE);
  double step[6];
  solve6_ldlt(h, g, step);
  double st[6];
  bool finite = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    st[i] = s_state[i] - lambda * step[i];                                            // :539
    finite = finite && (fabs(st[i]) <= 1.79769313486231570815e308);
  }
  double gn2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) gn2 += g[i] * g[i];
  const double gnorm = sqrt(gn2);                                                     // :380
  bool done = false;
  if (iteration + 1 >= max_iter) done = true;                                         // :383
  else if (gnorm < min_grad_norm) done = true;                                        // :388
  if (!finite) done = true;
  if (!done) write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], s_cst, lane);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) s_state[i] = st[i];
    s_ctl[CTL_DONE] = done ? 1 : 0;
    if (!finite) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_NONFINITE;
    if (last_valid < 6) s_ctl[CTL_FLAGS] |= (int)PHOVO_PAIR_RANK_DEFICIENT;
  }
  last_gnorm = gnorm;
}

// Wave butterfly + cross-wave sum + solve + update + termination, as in gn_level_kernel's tail, for kernels
// that keep their 27 sums in `acc` in every wave.  Returns after the closing barrier; the caller reads s_ctl[CTL_DONE].
template <int NW>
__device__ __forceinline__ void reduce_solve_update(double (&acc)[NRED], int lane, int wave, double *s_red,
                                                    double *s_state, double *s_cst, int *s_ctl,
                                                    double lambda, int max_iter, double min_grad_norm,
                                                    int iteration, double &last_gnorm, int &last_valid)
{
  reduce_wave_to_row(acc, lane, wave, s_red);
  __syncthreads();
  if (wave == 0)
    sum_rows_solve_update<NW>(lane, s_red, s_state, s_cst, s_ctl, lambda, max_iter, min_grad_norm, iteration, last_gnorm,
                              last_valid);
  __syncthreads();
}

}  // namespace

}  // namespace phovo_hip
