// "Wide" form of the Gauss-Newton level: MANY workgroups per frame pair, one launch per phase.
//
// gn_level_kernel (gn_kernels.hip) gives every pair one workgroup, which is what a batch of hundreds of pairs
// wants.  One pair -- the case the reference's FrameAlignment app times
// (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:99-102) -- then runs on ONE of 256 CUs:
// 0.5 ms per iteration of a 640x480 level.  Here the pixels of a pair are cut into tiles of 1024, each tile is a
// 256-thread workgroup, and an iteration is TWO launches on one stream:
//   k_wide_pass1   (from the second iteration on) EVERY tile workgroup first finishes the iteration before: fixed-order
//                  sum of that iteration's tile sums, LDL^T, state update, termination (:539-549), pose constants of
//                  the new state -- redundantly, bit-identical in every workgroup by construction; tile 0 of the pair
//                  records it -- then warps its source pixels, atomicMax into the owner map in HBM (:279-303,358)
//   k_wide_pass2   residual / Jacobian rows / 27 partial sums per tile (:308-356, :538-540)
// and ONE k_wide_finish behind the last iteration (the last solve, final state back to the engine's buffer).  Round 2
// ran the solve as a third launch per iteration (one workgroup per pair): a dependent kernel boundary costs about as
// much as the solve itself, and this is the form the reference's own timer sees (one pair per Optimize() call).
// There is no in-kernel grid barrier: the kernel boundary is the synchronisation, so nothing can hang; the host
// looks at the per-pair "done" words every few iterations.  Same arithmetic as the persistent kernel (same helper
// functions), same reference semantics, fp64 planes only (the narrow storages and Huber weights are served by
// the persistent kernel).
#include <hip/hip_runtime.h>

#include "gn_device.hpp"
#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

constexpr int WT = 256;                 // threads per tile workgroup
constexpr int TILE_CHUNKS = 16;         // 64-pixel chunks per tile (1024 pixels), 4 per wave
constexpr int WNW = WT / WAVE;

struct PoseRegs {
  double cx, cyy, cz, r01, r02, r11, r12, t1, t2, t3, t4, t5, t6, t8, t11, t14, t15, t16, t17, t24, cosy, siny;
};

__device__ __forceinline__ PoseRegs load_pose(const double *c)
{
  PoseRegs p;
  p.cx = c[C_X]; p.cyy = c[C_Y]; p.cz = c[C_Z];
  p.r01 = c[C_R01]; p.r02 = c[C_R02]; p.r11 = c[C_R11]; p.r12 = c[C_R12];
  p.t1 = c[C_T1]; p.t2 = c[C_T2]; p.t3 = c[C_T3]; p.t4 = c[C_T4]; p.t5 = c[C_T5]; p.t6 = c[C_T6];
  p.t8 = c[C_T8]; p.t11 = c[C_T11]; p.t14 = c[C_T14]; p.t15 = c[C_T15]; p.t16 = c[C_T16]; p.t17 = c[C_T17];
  p.t24 = c[C_T24]; p.cosy = c[C_CY]; p.siny = c[C_SY];
  return p;
}

// ctl words per pair
enum { W_DONE = 0, W_FLAGS = 1, W_ITER = 2, W_COUNT = 4 };

// The state of a pair is double-buffered in the workspace: the solve at the head of iteration i's pass 1 reads
// g_st[pair][(i - 1) & 1] in every tile workgroup while tile 0 writes g_st[pair][i & 1].
__global__ __launch_bounds__(WAVE) void k_wide_init(const GNLevelArgs A, double *g_cst, double *g_st, int *g_ctl)
{
  const int pair = blockIdx.x, lane = threadIdx.x;
  double st[6];
#pragma unroll
  for (int j = 0; j < 6; j++) st[j] = A.states[(size_t)pair * 6 + j];
  write_pose_constants(st[0], st[1], st[2], st[3], st[4], st[5], g_cst + (size_t)pair * 32, lane);
  if (lane < 6) g_st[(size_t)pair * 12 + lane] = st[lane];
  if (lane < W_COUNT) g_ctl[pair * W_COUNT + lane] = 0;
}

constexpr int SOLVE_T = 256;            // threads that sum the tile sums: 8 interleaved tile subsets x 32 values
static_assert(SOLVE_T == WT, "the solve runs in the tile workgroups");

// The Gauss-Newton step of iteration `it` (1-based count after it) from the tile sums of that iteration, executed by a
// whole 256-thread workgroup; returns in every thread of wave 0 what the step decided.  Tile sums are added in a fixed
// order without a 300-deep dependent chain: thread (s, j) adds tiles s, s+8, s+16, ... of value j (independent loads,
// pipelined), then the 8 subset sums are added in subset order -- every workgroup that runs this for the same pair
// gets the same bits.
struct WideStep {
  double st[6];
  double gnorm;
  int n_valid;
  bool done, finite;
};
__device__ __forceinline__ WideStep wide_solve(const GNLevelArgs &A, const double *g_part_pair, int tiles,
                                                const double *st_prev, int it, double *s_part)
{
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  {
    const int j = tid & (NRED - 1), sub = tid / NRED;
    const double *base = g_part_pair + j;
    double v0 = 0.0;
#pragma unroll 4
    for (int t = sub; t < tiles; t += SOLVE_T / NRED) v0 += base[(size_t)t * NRED];
    s_part[sub * NRED + j] = v0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  WideStep w{};
  if (tid >= WAVE) return w;                    // (the caller lets only wave 0 look at the result)
  double v = 0.0;
  if (lane < NRED) {
#pragma unroll
    for (int sub = 0; sub < SOLVE_T / NRED; sub++) v += s_part[sub * NRED + lane];
  }
  double h[21], g[6];
#pragma unroll
  for (int q = 0; q < 21; q++) h[q] = __shfl(v, q, WAVE);
#pragma unroll
  for (int i = 0; i < 6; i++) g[i] = __shfl(v, 21 + i, WAVE);
  w.n_valid = (int)__shfl(v, RED_VALID, WAVE);
  double step[6];
  solve6_ldlt(h, g, step);
  w.finite = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    w.st[i] = st_prev[i] - A.lambda * step[i];                                               // :539
    w.finite = w.finite && (fabs(w.st[i]) <= 1.79769313486231570815e308);
  }
  double gn2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) gn2 += g[i] * g[i];
  w.gnorm = sqrt(gn2);                                                                       // :380
  w.done = false;
  if (it >= A.max_iter) w.done = true;                                                       // :383
  else if (w.gnorm < A.min_grad_norm) w.done = true;                                         // :388
  if (!w.finite) w.done = true;
  return w;
}

// What tile 0 of a pair (or k_wide_finish) records of a step: state into the OTHER buffer, control words, report.
__device__ __forceinline__ void wide_record(const GNLevelArgs &A, int pair, const WideStep &w, int it, double *g_st,
                                            int *ctl)
{
#pragma unroll
  for (int i = 0; i < 6; i++) g_st[(size_t)pair * 12 + (size_t)(it & 1) * 6 + i] = w.st[i];
  ctl[W_ITER] = it;
  ctl[W_DONE] = w.done ? 1 : 0;
  if (!w.finite) ctl[W_FLAGS] |= (int)PHOVO_PAIR_NONFINITE;
  if (A.reports) {
    A.reports[pair].iterations[A.level] = it;
    A.reports[pair].gradient_norm = w.gnorm;
    A.reports[pair].valid_pixels[A.level] = w.n_valid;
    if (!w.finite) A.reports[pair].flags |= PHOVO_PAIR_NONFINITE;
    if (w.n_valid < 6) A.reports[pair].flags |= PHOVO_PAIR_RANK_DEFICIENT;
  }
}

// `it` = iterations completed before this launch (0 for the first): with it > 0 the step of iteration `it` is taken first.
__global__ __launch_bounds__(WT) void k_wide_pass1(const GNLevelArgs A, double *g_cst, double *g_st, int *g_ctl,
                                                   const double *g_part, int tiles, int it, unsigned long long *g_mask)
{
  const int pair = blockIdx.y;
  __shared__ double s_part[(SOLVE_T / NRED) * NRED];
  __shared__ double s_cst[32];
  __shared__ int s_done;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  // Tile 0 of this pair -- possibly running right now -- may record this very step as the pair's last: a workgroup that
  // reads that leaves, one that does not reaches the same decision itself.  ONE thread looks, so that the whole
  // workgroup goes the same way (the word can change while the threads of a workgroup read it).
  if (tid == 0) s_done = g_ctl[pair * W_COUNT + W_DONE];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (s_done) return;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                               // (s_done is written again below)
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w, H = A.h;
  const double *cst = g_cst + (size_t)pair * 32;
  if (it > 0) {                                                           // uniform
    const WideStep w = wide_solve(A, g_part + (size_t)pair * tiles * NRED, tiles,
                                  g_st + (size_t)pair * 12 + (size_t)((it - 1) & 1) * 6, it, s_part);
    if (wave == 0) {
      if (!w.done) write_pose_constants(w.st[0], w.st[1], w.st[2], w.st[3], w.st[4], w.st[5], s_cst, lane);
      if (lane == 0) s_done = w.done ? 1 : 0;
      if (blockIdx.x == 0) {                                              // the pair's recorder
        if (!w.done) write_pose_constants(w.st[0], w.st[1], w.st[2], w.st[3], w.st[4], w.st[5], g_cst + (size_t)pair * 32, lane);
        if (lane == 0) wide_record(A, pair, w, it, g_st, g_ctl + pair * W_COUNT);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (s_done) return;
    cst = s_cst;
  }
  const PoseRegs P = load_pose(cst);
  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const __amdgpu_buffer_rsrc_t rD0 = plane_rsrc<double>(src_frame + A.plane_off[PLANE_D], n);
  int *g_owner = A.g_owner + (size_t)pair * (size_t)n;
  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double min_d = A.min_depth, max_d = A.max_depth, dW = (double)W, dH = (double)H;
  const RowColFromIndex rc_map = make_rowcol_from_index(W);
  // The four depths of this wave are requested before anything else: a wave's memory counter retires in order, so a
  // load issued behind one of the global atomics below would wait for that atomic.
  constexpr int CPW = TILE_CHUNKS / WNW;      // chunks per wave
  double pzs[CPW];
#pragma unroll
  for (int j = 0; j < CPW; j++)
    pzs[j] = plane_load<double>(rD0, (blockIdx.x * TILE_CHUNKS + j * WNW + wave) * WAVE + lane);   // past the plane: 0
#pragma unroll
  for (int j = 0; j < CPW; j++) {
    const int chunk = blockIdx.x * TILE_CHUNKS + j * WNW + wave;
    if (chunk >= A.n_chunks) break;
    const int k = chunk * WAVE + lane;
    bool inb = false;
    const double pz = pzs[j];                                             // :279
    if (k < n && min_d < pz && pz < max_d) {                              // :280
      double cd, rd;                                                      // (row, column) without an integer division
      rowcol_from_index((double)k, rc_map, cd, rd);
      const double px = (cd - ox) * pz * ifx;                             // :282
      const double py = (rd - oy) * pz * ify;                             // :283
      const double X = ((P.t15 * px + P.r01 * py) + P.r02 * pz) + P.cx;   // :291
      const double Y = ((P.t14 * px + P.r11 * py) + P.r12 * pz) + P.cyy;
      const double Z = ((-P.t3 * px + P.t1 * py) + P.t2 * pz) + P.cz;
      const double iz = fast_rcp(Z);                                      // :294
      const double tc = (X * fx) * iz + ox;                               // :295
      const double tr = (Y * fy) * iz + oy;                               // :296
      const double rr = round(tr), rc = round(tc);                        // :297-298
      if (rr >= 0.0 && rr < dH && rc >= 0.0 && rc < dW) {                 // :302-303
        inb = true;
        atomicMax(&g_owner[__mul24((int)rr, W) + (int)rc], k);            // last raster writer wins  :358
      }
    }
    const unsigned long long m = __ballot(inb);
    if (lane == 0) g_mask[(size_t)pair * A.n_chunks + chunk] = m;
  }
}

__global__ __launch_bounds__(WT) void k_wide_pass2(const GNLevelArgs A, const double *g_cst, const int *g_ctl,
                                                   const unsigned long long *g_mask, double *g_part, int tiles)
{
  const int pair = blockIdx.y;
  if (g_ctl[pair * W_COUNT + W_DONE]) return;
  __shared__ double s_red[WNW * NRED];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int n = A.n, W = A.w;
  const PoseRegs P = load_pose(g_cst + (size_t)pair * 32);
  const unsigned char *src_frame = A.planes + (size_t)A.src[pair] * A.frame_bytes;
  const unsigned char *tgt_frame = A.planes + (size_t)A.tgt[pair] * A.frame_bytes;
  const __amdgpu_buffer_rsrc_t rI0 = plane_rsrc<double>(src_frame + A.plane_off[PLANE_I], n);
  const __amdgpu_buffer_rsrc_t rD0 = plane_rsrc<double>(src_frame + A.plane_off[PLANE_D], n);
  const __amdgpu_buffer_rsrc_t rI1 = plane_rsrc<double>(tgt_frame + A.plane_off[PLANE_I], n);
  const __amdgpu_buffer_rsrc_t rGX = plane_rsrc<double>(tgt_frame + A.plane_off[PLANE_GX], n);
  const __amdgpu_buffer_rsrc_t rGY = plane_rsrc<double>(tgt_frame + A.plane_off[PLANE_GY], n);
  int *g_owner = A.g_owner + (size_t)pair * (size_t)n;
  const double fx = A.fx, fy = A.fy, ox = A.ox, oy = A.oy, ifx = A.ifx, ify = A.ify;
  const double t7 = -P.t6, t9 = -P.t8, t21 = -P.t5;

  double acc[NRED];
#pragma unroll
  for (int j = 0; j < NRED; j++) acc[j] = 0.0;
  const RowColFromIndex rc_map = make_rowcol_from_index(W);
  int n_rows = 0;                   // Jacobian rows this wave fills (popcount of its chunks' ballots)
  // Every load of the wave's four chunks goes out first -- owners, then the planes, then the gathers that need the
  // owners -- and the owner slots are reset only after the last load has been issued: issued in between, each store and
  // each dependent gather would hold up everything behind it (the memory counter retires in order).
  constexpr int CPW = TILE_CHUNKS / WNW;      // chunks per wave
  int os[CPW];
  unsigned long long ms[CPW];
  double pzs[CPW], gxs[CPW], gys[CPW], i1s[CPW], i0s[CPW];
#pragma unroll
  for (int j = 0; j < CPW; j++) {
    const int chunk = blockIdx.x * TILE_CHUNKS + j * WNW + wave;
    const int k = chunk * WAVE + lane;
    os[j] = k < n ? g_owner[k] : -1;
    ms[j] = chunk < A.n_chunks ? g_mask[(size_t)pair * A.n_chunks + chunk] : 0ull;
  }
#pragma unroll
  for (int j = 0; j < CPW; j++) {
    const int k = (blockIdx.x * TILE_CHUNKS + j * WNW + wave) * WAVE + lane;
    pzs[j] = plane_load<double>(rD0, k);                                  // past the plane: 0
    gxs[j] = plane_load<double>(rGX, k);                                  // gradient at the SOURCE index  :346-347
    gys[j] = plane_load<double>(rGY, k);
    i1s[j] = plane_load<double>(rI1, k);                                  // :309
  }
#pragma unroll
  for (int j = 0; j < CPW; j++) i0s[j] = plane_load<double>(rI0, os[j]);  // :308 (owner -1: past the plane -> 0)
#pragma unroll
  for (int j = 0; j < CPW; j++) {                 // every slot is read once and made ready for the next pass 1
    const int k = (blockIdx.x * TILE_CHUNKS + j * WNW + wave) * WAVE + lane;
    if (k < n) g_owner[k] = -1;
  }
#pragma unroll
  for (int j = 0; j < CPW; j++) {
    const int chunk = blockIdx.x * TILE_CHUNKS + j * WNW + wave;
    if (chunk >= A.n_chunks) break;
    const int k = chunk * WAVE + lane;
    const int o = os[j];
    const unsigned long long m = ms[j];
    n_rows += __builtin_popcountll(m);
    if (!((m >> lane) & 1ull)) continue;
    const double pz = pzs[j];
    const double gxi = gxs[j], gyi = gys[j];
    const double res = o >= 0 ? i1s[j] - i0s[j] : 0.0;                    // :308-309,358
    double cd, rd;
    rowcol_from_index((double)k, rc_map, cd, rd);
    const double px = (cd - ox) * pz * ifx;
    const double py = (rd - oy) * pz * ify;
    // same factored Jacobian as gn_level_kernel (see the derivation there)
    const double Zr = py * P.t1 + pz * P.t2 - px * P.t3;
    const double t25 = fast_rcp(P.cz + Zr);                               // :313
    const double Au = pz * P.t4 + py * P.t5 + px * P.t11;                 // temp11 = temp15 + x: the reference's slip, kept
    const double Bv = py * P.t6 + pz * t9 + px * P.t14 + P.cyy;
    const double Cm = -py * P.t16 - pz * P.t17 - px * P.t24;
    const double Dm = py * P.t2 - pz * P.t1;
    double J[6];
    J[0] = (gxi * fx) * t25;
    J[1] = (gyi * fy) * t25;
    J[2] = -(J[0] * Au + J[1] * Bv) * t25;
    J[3] = J[0] * (P.cyy - Bv) + J[1] * (Au - px * P.cx);
    J[4] = (J[0] * P.cosy + J[1] * P.siny) * Zr + Cm * J[2];
    J[5] = J[0] * (py * P.t4 + pz * t21) + J[1] * (pz * t7 + py * t9) + Dm * J[2];
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int b = a; b < 6; b++) {
        acc[q] = fma(J[a], J[b], acc[q]);                                 // J^T J  :540
        q++;
      }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] = fma(J[a], res, acc[21 + a]);  // J^T r  :538
  }
  acc[RED_VALID] = lane == 0 ? (double)n_rows : 0.0;
  // tile sums: wave butterfly, then the four waves in fixed order
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
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < NRED) {
    double v = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < WNW; w2++) v += s_red[w2 * NRED + tid];
    g_part[((size_t)pair * tiles + blockIdx.x) * NRED + tid] = v;
  }
}

// Behind the last iteration: the step of that iteration for the pairs still running (as at the head of pass 1), and the
// final state of EVERY pair from whichever buffer holds it back into the engine's array.  One workgroup per pair.
__global__ __launch_bounds__(SOLVE_T) void k_wide_finish(const GNLevelArgs A, double *g_st, int *g_ctl,
                                                         const double *g_part, int tiles, int it)
{
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1);
  __shared__ double s_part[(SOLVE_T / NRED) * NRED];
  int *ctl = g_ctl + pair * W_COUNT;
  if (!ctl[W_DONE] && it > 0) {                                           // uniform
    const WideStep w = wide_solve(A, g_part + (size_t)pair * tiles * NRED, tiles,
                                  g_st + (size_t)pair * 12 + (size_t)((it - 1) & 1) * 6, it, s_part);
    if (tid == 0) wide_record(A, pair, w, it, g_st, ctl);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int done_it = ctl[W_ITER];                                      // iterations the pair completed: its state is in buffer done_it & 1
#pragma unroll
    for (int i = 0; i < 6; i++) A.states[(size_t)pair * 6 + i] = g_st[(size_t)pair * 12 + (size_t)(done_it & 1) * 6 + i];
    (void)lane;
  }
}

}  // namespace

size_t gn_wide_workspace_bytes(int n, int n_pairs)
{
  const size_t n_chunks = (size_t)(n + WAVE - 1) / WAVE;
  const size_t tiles = (n_chunks + TILE_CHUNKS - 1) / TILE_CHUNKS;
  return (size_t)n_pairs * ((32 + 12) * sizeof(double) + W_COUNT * sizeof(int) + tiles * NRED * sizeof(double) +
                            n_chunks * sizeof(unsigned long long)) + 256;
}

// Runs one level for n_pairs pairs with the wide kernels.  `workspace` has gn_wide_workspace_bytes() bytes,
// a.g_owner is an [n_pairs][n] int32 map holding -1 everywhere (and again on return).  Synchronises the stream
// every `check_every` iterations to read the per-pair done words.
hipError_t gn_run_level_wide(const GNLevelArgs &a, int n_pairs, void *workspace, int *h_done_scratch,
                             hipStream_t stream)
{
  if (n_pairs <= 0) return hipSuccess;
  const int tiles = (a.n_chunks + TILE_CHUNKS - 1) / TILE_CHUNKS;
  unsigned char *w = static_cast<unsigned char *>(workspace);
  double *g_cst = reinterpret_cast<double *>(w);              w += (size_t)n_pairs * 32 * sizeof(double);
  double *g_st = reinterpret_cast<double *>(w);               w += (size_t)n_pairs * 12 * sizeof(double);
  double *g_part = reinterpret_cast<double *>(w);             w += (size_t)n_pairs * tiles * NRED * sizeof(double);
  unsigned long long *g_mask = reinterpret_cast<unsigned long long *>(w);
  w += (size_t)n_pairs * a.n_chunks * sizeof(unsigned long long);
  int *g_ctl = reinterpret_cast<int *>(w);
  const dim3 grid((unsigned)tiles, (unsigned)n_pairs);
  hipLaunchKernelGGL(k_wide_init, dim3(n_pairs), dim3(WAVE), 0, stream, a, g_cst, g_st, g_ctl);
  // Iterations whose pairs have all stopped still cost two (empty) launches each, and asking the device costs a host
  // round trip: with a gradient threshold the first look comes after 3 iterations (pairs typically stop after 2-4),
  // then every 8; without one (min_gradient_norm = 0: every pair runs max_iter iterations) nobody asks at all.  The step
  // of iteration i is taken at the head of pass 1 of iteration i + 1: the done words the host sees after launching
  // iteration `it` are those of iteration it - 1.
  const bool may_stop_early = a.min_grad_norm > 0.0;
  int next_check = 4;
  int launched = 0;
  for (int it = 0; it < a.max_iter; it++) {
    hipLaunchKernelGGL(k_wide_pass1, grid, dim3(WT), 0, stream, a, g_cst, g_st, g_ctl, g_part, tiles, it, g_mask);
    hipLaunchKernelGGL(k_wide_pass2, grid, dim3(WT), 0, stream, a, g_cst, g_ctl, g_mask, g_part, tiles);
    launched = it + 1;
    if (may_stop_early && it + 1 >= next_check && a.max_iter - (it + 1) >= 4) {     // a look costs about three empty iterations
      next_check = it + 1 + 8;
      hipError_t e = hipMemcpyAsync(h_done_scratch, g_ctl, sizeof(int) * W_COUNT * (size_t)n_pairs,
                                    hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return e;
      e = hipStreamSynchronize(stream);
      if (e != hipSuccess) return e;
      bool all = true;
      for (int p = 0; p < n_pairs; p++) all = all && h_done_scratch[p * W_COUNT + W_DONE] != 0;
      if (all) break;
    }
  }
  hipLaunchKernelGGL(k_wide_finish, dim3(n_pairs), dim3(SOLVE_T), 0, stream, a, g_st, g_ctl, g_part, tiles, launched);
  return hipGetLastError();
}

}  // namespace phovo_hip
