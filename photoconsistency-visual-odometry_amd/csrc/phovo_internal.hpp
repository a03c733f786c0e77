// Internal declarations shared by the host engine and the HIP kernels (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "phovo_hip.h"

namespace phovo_hip {

// Planes of one frame at one level, in pool order.
enum Plane { PLANE_I = 0, PLANE_D = 1, PLANE_GX = 2, PLANE_GY = 3, PLANES_PER_FRAME = 4 };

// Everything one Gauss-Newton level launch needs (passed by value as the kernel argument,
// so the scalars land in SGPRs).
struct GNLevelArgs {
  int w, h, n;              // level size, n = w*h
  int level;
  int max_iter;             // max_num_iterations[level] (> 0)
  int n_chunks;             // ceil(n / 64)
  double lambda;            // lambda_optimization_step[level]
  double min_grad_norm;     // min_gradient_norm[level]
  double fx, fy, ox, oy, ifx, ify;   // level-scaled intrinsics (...Analytic.h:203-209)
  double min_depth, max_depth;
  double huber_delta;       // > 0: Huber IRLS weights (extension, not in the reference); <= 0: off
  const unsigned char *planes;   // pool of this level: frame f at planes + f*frame_bytes,
  size_t frame_bytes;            //   plane p of a frame at + plane_off[p], n elements of the storage type
  size_t plane_off[PLANES_PER_FRAME];
  const int *src;           // [pairs] source frame of each pair
  const int *tgt;           // [pairs] target frame of each pair
  double *states;           // [pairs][6] in: initial / previous level, out: updated
  phovo_pair_report *reports;   // [pairs]
  int *g_owner;             // [pairs][n] owner map in global memory (only when it does not fit LDS)
  unsigned long long *g_mask;   // [pairs][n_chunks] in-bounds ballots in global memory (only when they do not fit LDS either:
                                // levels above ~1.27 M pixels), else null
  int n_pairs;              // pairs of this launch
  int *work_counter;        // [QUEUES_PER_LEVEL] heads QUEUE_HEAD_STRIDE ints apart, zeroed before the launch: workgroups draw pair indices from them
  int n_queues;             // 1: one queue for the whole grid; 8: one per XCD over a contiguous eighth of the pairs (+ stealing)
  int n_lds;                // owner map in HBM only: its first n_lds entries (a multiple of 64) live in LDS instead
  int slide_m;              // sliding-window kernel: bands a target may lie away from its source's band (gn_slide_reach_bands)
  int depth_lds_chunks;     // owner map in LDS: the depth of the first this-many 64-pixel chunks is kept in leftover LDS by pass 1
                            // and read from there by pass 2 (0: none)
  // Hand-over of pairs from the sliding-window launch of a level to the exact launch right behind it (same stream).  A
  // list is [n_pairs + 1] ints, zeroed before the first launch: pair indices, and at [n_pairs] their number.
  //   handover_out  non-null (sliding-window kernel): a pair whose warp leaves the window is appended there (its state and
  //                 its completed iteration count, reports[p].iterations[level], are in place)
  //   handover_in   non-null: the launch works on that list instead of the pairs 0..n_pairs-1 and continues every pair at
  //                 its stored iteration count; takeover_flag is OR-ed into reports[p].flags of every pair taken
  const int *handover_in;
  int *handover_out;
  unsigned takeover_flag;
  size_t rec_off;           // bilinear sampling: byte offset, in a frame, of its TAP RECORDS {I, GX, GY, pad} (pyr_build_tap_records)
};

// Several consecutive levels of one launch (gn_fused_kernel): lv[0] is the coarsest.  The pair list, states, reports and
// the queue are those of lv[0]; every level shares them.
constexpr int GN_MAX_FUSED_LEVELS = 3;
struct GNFusedArgs {
  GNLevelArgs lv[GN_MAX_FUSED_LEVELS];
  int n_levels;
  int n_max;                // pixels of the largest level: sizes the owner map in LDS
  int n_pairs;
  int n_queues;
  int *work_counter;
};

constexpr int QUEUES_PER_LEVEL = 8;      // one per XCD
constexpr int QUEUE_HEAD_STRIDE = 32;    // ints between two queue heads: a 128-byte line each (the draws are atomics, served per line)
constexpr int QUEUE_HEADS_INTS = QUEUES_PER_LEVEL * QUEUE_HEAD_STRIDE;     // the heads of one launch

struct GNLaunchPlan {
  int variant;              // which instantiation of the level kernel (gn_kernels.hip)
  int threads;              // workgroup size
  int wgs_per_cu;           // workgroups of this geometry that fit one CU (sizes the persistent grid)
  int lds_bytes;            // dynamic LDS
  bool owner_in_lds;
  bool source_in_lds;
  int owner_lds_entries;    // owner map in HBM: how many of its leading entries the leftover LDS holds (GNLevelArgs::n_lds)
  bool mask_in_hbm;         // owner map in HBM and a level so large that the per-chunk ballots do not fit LDS either
  int depth_lds_chunks;     // leading 64-pixel chunks whose depth pass 1 parks in the LDS this geometry leaves unused (GNLevelArgs)
};

// Chooses the launch geometry for a level of n pixels.  Returns false if the level cannot be
// handled (inbound-mask does not fit LDS).
// prefer_latency 1: the geometry for a handful of pairs (each alone on a CU): 512 threads instead of four workgroups of 256.
// bound_by_bytes: the planes are fp64 (the level kernels are then bound by bytes at the fabric and a geometry may be chosen
// for the depth it can park in LDS; with narrow storages they are bound by the vector unit)
bool gn_plan_level(int n, GNLaunchPlan *plan, int prefer_latency = 0, bool bound_by_bytes = true);
// args.n_pairs pairs, args.work_counter zeroed on the stream beforehand; the grid is min(pairs, CUs x workgroups/CU).
hipError_t gn_launch_level(const GNLevelArgs &args, const GNLaunchPlan &plan, int storage, int cu_count,
                           hipStream_t stream);
// Several consecutive levels in ONE persistent launch (a pair flows through them inside the workgroup that drew it): every
// level must be gn_level_fusable (owner map in LDS beside a second 512-thread workgroup on the CU).
bool gn_level_fusable(int n_pixels);
int gn_fused_lds_bytes(int n_max);
bool gn_plan_fused_geometry(int n_pixels, GNLaunchPlan *plan);      // the per-level kernel in the fused launch's geometry
hipError_t gn_launch_fused(const GNFusedArgs &args, int storage, int cu_count, hipStream_t stream);
// Extension (PHOVO_SAMPLING_BILINEAR): single-pass kernels, 256 threads, no owner map, any level size (fp64 / fp32 planes:
// taps through LDS-DMA; fp16 planes: tap records).
hipError_t gn_launch_level_bilinear(const GNLevelArgs &args, int storage, bool corrected, int cu_count,
                                    hipStream_t stream);
int gn_bilinear_wgs_per_cu(int storage);      // workgroups of the bilinear kernel for that plane storage that stay resident per CU
// Wide form (gn_wide_kernels.hip): many workgroups per pair, three launches per iteration; for a handful of
// pairs on large levels.  fp64 planes, reference semantics only.
size_t gn_wide_workspace_bytes(int n, int n_pairs);
hipError_t gn_run_level_wide(const GNLevelArgs &args, int n_pairs, void *workspace, int *h_done_scratch,
                             hipStream_t stream);
hipError_t gn_prepare_kernels();   // raises the dynamic-LDS limit of every instantiation
// Sliding-window form for levels whose owner map exceeds LDS (gn_slide_kernel.hip): owner ring in LDS; pairs whose
// motion leaves the window are appended to args.handover_out for a follow-up gn_launch_level that takes that list.
hipError_t gn_prepare_slide_kernels();
hipError_t gn_launch_level_slide(const GNLevelArgs &args, int storage, int cu_count, hipStream_t stream);
size_t gn_slide_lds_bytes();
int gn_slide_threads();
int gn_slide_reach_bands(int w, int h);      // GNLevelArgs::slide_m for a level of w x h pixels

// Pyramid producers (SetSourceFrame / SetTargetFrame, ...Analytic.h:466-491), batched over `frames`
// consecutive frames: frame f reads src + f*src_frame_stride and writes dst + f*dst_frame_stride (elements).
hipError_t pyr_intensity_level(const uint8_t *gray, size_t src_frame_stride, int frames, int w, int h, int level,
                               int lw, int lh, double *dst, size_t dst_frame_stride, hipStream_t stream);
hipError_t pyr_depth_level(const double *depth, size_t src_frame_stride, int frames, int w, int h, int level,
                           int lw, int lh, double *dst, size_t dst_frame_stride, hipStream_t stream);
hipError_t pyr_depth_level_u16(const uint16_t *depth, size_t src_frame_stride, double scale, int frames, int w,
                               int h, int level, int lw, int lh, double *dst, size_t dst_frame_stride,
                               hipStream_t stream);
hipError_t pyr_scharr(const double *base, size_t frame_stride, size_t img_off, size_t gx_off, size_t gy_off,
                      int frames, int w, int h, double scale, hipStream_t stream);
// fp64 <-> storage-type plane conversion (storage = PHOVO_STORAGE_*, is_depth selects the depth element type).
size_t storage_elem_size(int storage, bool is_depth);
hipError_t pyr_store_plane(const double *src, size_t src_frame_stride, int frames, int n, unsigned char *dst,
                           size_t dst_frame_bytes, int storage, bool is_depth, hipStream_t stream);
hipError_t pyr_load_plane(const unsigned char *src, int n, double *dst, int storage, bool is_depth,
                          hipStream_t stream);
// Bilinear sampling (extension): one interleaved record {I, GX, GY, pad} per pixel of a target frame, in the planes' own
// storage type (the same bits as the three planes), so that a bilinear tap is one or two wide loads instead of three
// gathers: frames consecutive frames of a level pool, planes at plane_off[] and the records at rec_off in every frame.
hipError_t pyr_build_tap_records(unsigned char *pool, size_t frame_bytes, const size_t plane_off[PLANES_PER_FRAME],
                                 size_t rec_off, int frames, int n, int storage, hipStream_t stream);
hipError_t pyr_gaussian_blur(double *img, double *tmp, int w, int h, int ksize,
                             const double *d_kernel, hipStream_t stream);
hipError_t fill_i32(int *dst, size_t n, int value, hipStream_t stream);

// Diagnostic switches read from the environment exist only in a -DPHOVO_TUNING build (tools/); the release library reads
// PHOVO_VISUALIZE_DIR and nothing else.
#ifdef PHOVO_TUNING
bool tuning_switch(const char *name);
#else
inline bool tuning_switch(const char *) { return false; }
#endif

// Error plumbing
void set_last_error(const std::string &msg);
int fail(int status, const std::string &msg);

// OpenCV-FileStorage-dialect reader (yml_config.cpp)
int read_config_file(const char *path, phovo_config *cfg);
int read_extensions_file(const char *path, phovo_extensions *ext);

}  // namespace phovo_hip
