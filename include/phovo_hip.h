/*
 * phovo_hip.h -- C ABI of the MI355X-native analytic Gauss-Newton RGB-D alignment path.
 *
 * This is the drop-in boundary for ONE path of MiguelAlgaba/photoconsistency-visual-odometry:
 *   phovo::Analytic::CPhotoconsistencyOdometryAnalytic<unsigned char,double>
 *   (phovo/include/CPhotoconsistencyOdometryAnalytic.h:57-608), i.e. the abstract surface of
 *   phovo/include/CPhotoconsistencyOdometry.h:137-179 plus SetMinDepth/SetMaxDepth/
 *   ReadConfigurationFile.
 * The reference has no FFI: its "plugin API" is that C++ template class.  The functions
 * below are what a binding of that class binds (include/phovo/CPhotoconsistencyOdometryAnalytic.h
 * in this repository is such a binding; INTEGRATION.md shows it next to the reference's apps).
 * Plain pointers and sizes only; every function returns a phovo_status (0 = ok) and never throws.
 * All images are row-major; strides are in BYTES; the library copies what it is given (the
 * caller keeps ownership of every buffer, as with the reference's const& arguments,
 * ...Analytic.h:466-491).  All arithmetic is fp64 on the device.
 *
 * Two layers:
 *   phovo_odometry_*  one frame pair at a time -- 1:1 with the reference's methods.
 *   phovo_engine_*    the batched form the throughput path uses: a pool of frames whose
 *                     pyramids live in HBM and a list of (source, target) frame pairs that
 *                     are aligned by one launch per active pyramid level.
 * A process drives one engine per GPU; engines are independent (one HIP stream each).
 */
#ifndef PHOVO_HIP_H
#define PHOVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHOVO_MAX_LEVELS 16

typedef enum phovo_status {
  PHOVO_OK = 0,
  PHOVO_E_INVALID_ARGUMENT = 1,  /* NULL pointer, index out of range                         */
  PHOVO_E_CONFIG = 2,            /* missing / malformed key, per-level array too short        */
  PHOVO_E_SHAPE = 3,             /* frame size does not match the pool, level too large       */
  PHOVO_E_HIP = 4,               /* a HIP runtime call failed (message: phovo_last_error())   */
  PHOVO_E_NOT_READY = 5,         /* call order violated (e.g. Optimize before Set*Frame)      */
  PHOVO_E_IO = 6,                /* file cannot be opened                                     */
  PHOVO_E_UNSUPPORTED = 7        /* a mode the device path does not implement                 */
} phovo_status;

/* Flags in phovo_pair_report.flags */
#define PHOVO_PAIR_NONFINITE 1u  /* J^T J was singular / the state became inf or NaN: the reference
                                    propagates inf/NaN silently (...Analytic.h:540); so does this
                                    library, but it says so here.                                 */
#define PHOVO_PAIR_WINDOW_FALLBACK 2u  /* informational: on a level whose owner map exceeds LDS this pair's warp left
                                    the sliding window of the fast kernel (a displacement of more than about 20 rows of
                                    the level -- 13 800 pixels in linear index at 640x480, 7 700 at 320x240 -- e.g. a
                                    large in-plane rotation) and the exact kernel with the map in HBM finished it.  The
                                    result is the same; only the time differs.                                   */
#define PHOVO_PAIR_RANK_DEFICIENT 4u  /* fewer than six Jacobian rows were filled in SOME iteration of some level (the
                                    flag is sticky; phovo_pair_report.valid_pixels holds the count of each level's last
                                    iteration): J^T J was singular by construction there and the step taken from it
                                    rounding noise -- finite or not -- exactly as in the reference, which inverts such a
                                    matrix without a word (...Analytic.h:540).                                        */

/* The per-level parameter vectors of the reference (...Analytic.h:91-103), as filled by
 * ReadConfigurationFile (:581-607) or by the constructor defaults (:430-443). */
typedef struct phovo_config {
  int    num_levels;                                       /* numOptimizationLevels                      */
  int    blur_filter_size[PHOVO_MAX_LEVELS];               /* blurFilterSize (at each level)             */
  double image_gradients_scaling_factor[PHOVO_MAX_LEVELS]; /* imageGradientsScalingFactor (at each level)*/
  double lambda_optimization_step[PHOVO_MAX_LEVELS];       /* lambda_optimization_step (at each level)   */
  int    max_num_iterations[PHOVO_MAX_LEVELS];             /* max_num_iterations (at each level)         */
  double min_gradient_norm[PHOVO_MAX_LEVELS];              /* min_gradient_norm (at each level)          */
  int    visualize_iterations;                             /* visualizeIterations: the reference shows |I1 - warped|
                                                              in a window after every iteration (:551-557); here
                                                              phovo_odometry_optimize writes that image to
                                                              $PHOVO_VISUALIZE_DIR/optimize_imgDiff_level<L>_iteration<N>.pgm
                                                              when the variable is set, and ignores the key otherwise */
} phovo_config;

/* What Optimize() did for one pair (the reference only prints this behind
 * ENABLE_PRINT_CONSOLE_OPTIMIZATION_PROGRESS, ...Analytic.h:396-422). */
typedef struct phovo_pair_report {
  int      iterations[PHOVO_MAX_LEVELS]; /* m_Iteration when each level stopped (:547-549)      */
  double   gradient_norm;                /* ||J^T r|| of the last executed iteration (:380)      */
  uint32_t flags;                        /* PHOVO_PAIR_*                                         */
  uint32_t reserved;
  int32_t  valid_pixels[PHOVO_MAX_LEVELS]; /* rows of J filled in the LAST executed iteration of each level: source
                                            pixels that passed the depth gate (:280) and whose warp landed inside the
                                            image (:302-303); 0 for a level with max_num_iterations == 0.  A handful of
                                            them (a rank-deficient J^T J) is what precedes PHOVO_PAIR_NONFINITE.   */
} phovo_pair_report;

/* ---- extensions that are NOT in the reference (BASELINE.json configs[4]); all off by default -------------
 * plane_storage: how the pyramid planes are kept in HBM.  Arithmetic is fp64 in every mode; pyramids are built
 * in fp64 and rounded once when stored (fp64 -> fp32 by round-to-nearest-even, fp16 via fp32).
 *   PHOVO_STORAGE_F64  reference-exact (cv::Mat_<double>, ...Analytic.h:73-85)
 *   PHOVO_STORAGE_F32  all four planes fp32  (half the HBM footprint and traffic)
 *   PHOVO_STORAGE_F16  intensity and gradients fp16, depth fp32  (10 instead of 32 bytes per pixel)
 * huber_delta[L] > 0 turns the least-squares step of level L into an IRLS step of the Huber loss: residual k gets
 * the weight 1 if |r_k| <= delta, delta/|r_k| otherwise, in both J^T W J and J^T W r (and hence in the gradient norm
 * of the termination test).  <= 0: off. */
#define PHOVO_STORAGE_F64 0
#define PHOVO_STORAGE_F32 1
#define PHOVO_STORAGE_F16 2
/* sampling: how the target frame is looked up.
 *   PHOVO_SAMPLING_NEAREST_SCATTER  the reference's analytic path: C round() of the warped position, residual
 *                                   scattered to the target index, Jacobian row at the source index (...Analytic.h:297-358)
 *   PHOVO_SAMPLING_BILINEAR         forward-additive alignment with bilinear samples of intensity and gradients at the
 *                                   real-valued warped position (in bounds iff the nearest pixel is, taps clamped to
 *                                   the edge); residual and Jacobian row both belong to the source pixel: no scatter.
 * jacobian_corrected (bilinear sampling only): 1 = the true warp Jacobian, i.e. without the reference's
 * `temp11 = cos(pitch)*cos(yaw)+x` transcription slip (...Analytic.h:253); 0 = the reference's Jacobian. */
#define PHOVO_SAMPLING_NEAREST_SCATTER 0
#define PHOVO_SAMPLING_BILINEAR 1
typedef struct phovo_extensions {
  int    plane_storage;                     /* PHOVO_STORAGE_*            yml: "plane_storage_bits: 64|32|16" */
  int    sampling;                          /* PHOVO_SAMPLING_*           yml: "sampling_bilinear: 0|1"       */
  int    jacobian_corrected;                /* 0 | 1                      yml: "jacobian_corrected: 0|1"      */
  int    reserved;
  double huber_delta[PHOVO_MAX_LEVELS];     /* optional yml key "huber_delta (at each level)"             */
} phovo_extensions;

typedef struct phovo_engine phovo_engine;
typedef struct phovo_odometry phovo_odometry;

/* ---- library ---------------------------------------------------------------------------- */
const char *phovo_version(void);
const char *phovo_status_string(int status);
/* Message of the most recent failure on the calling thread ("" if none). */
const char *phovo_last_error(void);
/* Number of HIP devices visible (0 if none / no driver). */
int phovo_device_count(void);

/* ---- configuration: ...Analytic.h:430-443 (defaults), :581-607 (ReadConfigurationFile) ---- */
int phovo_config_default(phovo_config *cfg);
/* Parses the reference's config_files/ *.yml unchanged (OpenCV FileStorage "%YAML:1.0" dialect,
 * keys with spaces and parentheses, per-level arrays that may be longer than num_levels). */
int phovo_config_read_file(const char *path, phovo_config *cfg);

int phovo_extensions_default(phovo_extensions *ext);
/* Optional keys in the same yml file (the reference's cv::FileStorage lookups ignore keys they do not ask for,
 * so such a file still loads there): "huber_delta (at each level): [..]", "plane_storage_bits: 64|32|16",
 * "sampling_bilinear: 0|1", "jacobian_corrected: 0|1".
 * Absent keys leave the defaults (everything off). */
int phovo_extensions_read_file(const char *path, phovo_extensions *ext);

/* eigenPose, CPhotoconsistencyOdometry.h:47-71: (x,y,z,yaw,pitch,roll) -> row-major 4x4. */
int phovo_eigen_pose(const double state[6], double rt[16]);

/* The VisualOdometry app's pose chain and trajectory line
 * (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:233-243):
 *   pose *= Rt^-1 per pair, starting from `pose_io` (row-major 4x4; identity at the start of a sequence), with
 *   Rt = eigenPose(states[p]); poses_out[p] (may be NULL) receives the pose after pair p, pose_io the last one.
 * Host arithmetic only (no GPU): one implementation shared by the app's loop, its --batch path and the sharded
 * sequence driver, so that their trajectory files are byte-identical. */
int phovo_trajectory_chain(int n_pairs, const double *states /* [n_pairs][6] */, double pose_io[16],
                           double *poses_out /* [n_pairs][16] or NULL */);
/* `timestamp tx ty tz qx qy qz qw` with digits10 + 1 = 16 significant digits (:240-243), quaternion from the
 * rotation block as Eigen::Quaternion(Matrix3) builds it (:237); no newline.  Returns PHOVO_E_INVALID_ARGUMENT if
 * `capacity` is too small (256 always suffices). */
int phovo_trajectory_format_pose(double timestamp, const double pose[16], char *line, size_t capacity);

/* warpImage, CPhotoconsistencyOdometry.h:73-134 -- the forward warp both reference apps call after Optimize()
 * (...FrameAlignment.cpp:108, ...VisualOdometry.cpp:248-250) to show |I1 - warp(I0)|.  Host buffers in and out,
 * strides in bytes; rt row-major 4x4, k row-major 3x3, level scales the intrinsics by 2^-level as the reference does.
 * Reference semantics: depth > 0 gate, truncating cast of the projected position, the last source pixel in raster
 * order that lands on a target pixel stays, zeros elsewhere. */
int phovo_warp_image(int device, const uint8_t *intensity, size_t intensity_stride_bytes,
                     const double *depth, size_t depth_stride_bytes, int w, int h, const double rt[16],
                     const double k[9], int level, uint8_t *warped, size_t warped_stride_bytes);

/* ---- single pair: 1:1 with CPhotoconsistencyOdometryAnalytic<unsigned char,double> ------- */
int phovo_odometry_create(int device, phovo_odometry **out);               /* ctor  :430-443 */
int phovo_odometry_destroy(phovo_odometry *o);                             /* dtor  :445     */
int phovo_odometry_read_configuration_file(phovo_odometry *o, const char *path);   /* :581 */
int phovo_odometry_set_config(phovo_odometry *o, const phovo_config *cfg);
int phovo_odometry_set_extensions(phovo_odometry *o, const phovo_extensions *ext);   /* not in the reference */
/* not in the reference: 1 = Optimize() may take the forms that finish soonest for ONE pair (phovo_engine_set_latency_forms:
 * last bits may then differ from the same pair aligned in a batch); default 0 */
int phovo_odometry_set_latency_forms(phovo_odometry *o, int on);
int phovo_odometry_set_min_depth(phovo_odometry *o, double min_depth);     /* :448 */
int phovo_odometry_set_max_depth(phovo_odometry *o, double max_depth);     /* :454 */
int phovo_odometry_set_intrinsic_matrix(phovo_odometry *o, const double k[9]);     /* :460, row-major 3x3 */
/* SetSourceFrame :466-476 -- intensity u8, depth fp64 metres; builds the intensity and depth pyramids. */
int phovo_odometry_set_source_frame(phovo_odometry *o,
                                    const uint8_t *intensity, size_t intensity_stride,
                                    const double *depth, size_t depth_stride,
                                    int width, int height);
/* SetTargetFrame :479-491 -- depth is ignored (may be NULL); builds intensity + Scharr pyramids. */
int phovo_odometry_set_target_frame(phovo_odometry *o,
                                    const uint8_t *intensity, size_t intensity_stride,
                                    const double *depth, size_t depth_stride,
                                    int width, int height);
int phovo_odometry_set_initial_state_vector(phovo_odometry *o, const double state[6]);     /* :494 */
int phovo_odometry_optimize(phovo_odometry *o);                                            /* :500 */
int phovo_odometry_get_optimal_state_vector(const phovo_odometry *o, double state[6]);     /* :566 */
int phovo_odometry_get_optimal_rigid_transformation_matrix(const phovo_odometry *o, double rt[16]); /* :572 */
int phovo_odometry_get_report(const phovo_odometry *o, phovo_pair_report *report);
/* Device time of the last Optimize() in milliseconds (HIP events; the reference wraps the same
 * call in cv::TickMeter, apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:99-102). */
int phovo_odometry_last_optimize_ms(const phovo_odometry *o, double *ms);

/* ---- batched engine ------------------------------------------------------------------------ */
#define PHOVO_ROLE_SOURCE 1   /* needs intensity + depth pyramids   (SetSourceFrame) */
#define PHOVO_ROLE_TARGET 2   /* needs intensity + gradient pyramids (SetTargetFrame) */
#define PHOVO_ROLE_BOTH   3

int phovo_engine_create(int device, phovo_engine **out);
int phovo_engine_destroy(phovo_engine *e);
int phovo_engine_set_config(phovo_engine *e, const phovo_config *cfg);
int phovo_engine_get_config(const phovo_engine *e, phovo_config *cfg);
/* Changing plane_storage drops the frame pool (like a configuration that changes the levels). */
int phovo_engine_set_extensions(phovo_engine *e, const phovo_extensions *ext);
int phovo_engine_get_extensions(const phovo_engine *e, phovo_extensions *ext);
int phovo_engine_set_intrinsic_matrix(phovo_engine *e, const double k[9]);
int phovo_engine_set_depth_range(phovo_engine *e, double min_depth, double max_depth);
/* 0 (default): only levels with max_num_iterations > 0 are built and kept in HBM (the others are
 * never read by Optimize()).  1: every level, as the reference does (:474-475,487-490). */
int phovo_engine_set_build_all_levels(phovo_engine *e, int on);

/* How a level is run.  The persistent form gives every pair ONE workgroup for all iterations of a level (the
 * throughput form).  The wide form cuts a pair into tiles of 1024 pixels, one workgroup each, with two launches
 * per iteration and a host look at the "done" words every 8 iterations (the latency form for a handful of pairs on
 * a large level; reference-exact configuration only).  policy: 0 = automatic, 1 = wide wherever possible, -1 = never.
 * Automatic: wide iff n_pairs <= 32 and the level's owner map does not fit LDS (more than ~39 k pixels: 320x240,
 * 640x480 -- one workgroup would need hundreds of microseconds per iteration there); never with
 * phovo_engine_set_batch_invariant.  Same iteration counts and poses within the parity bar either way (the forms sum in
 * different orders). */
int phovo_engine_set_wide_policy(phovo_engine *e, int policy);
/* Levels whose owner map exceeds LDS (more than ~39 k pixels) run the sliding-window kernel (owner ring in LDS) followed
 * by the exact kernel (owner map in HBM) for the pairs whose warp left the window.  policy: 0 = automatic (that), -1 =
 * exact kernel only.  Results are the same either way (tests/test_gpu_parity.py). */
int phovo_engine_set_slide_policy(phovo_engine *e, int policy);
/* Consecutive pyramid levels in ONE launch.  The reference's Optimize() loops per pair over levels
 * (CPhotoconsistencyOdometryAnalytic.h:502-563); with a gradient threshold (min_gradient_norm > 0, :388) a pair leaves a
 * level after a data-dependent number of iterations, and one launch per level would make every level boundary a boundary
 * for the whole batch.  PHOVO_FUSION_AUTO (default): where two or more consecutive active levels each fit the 512-thread
 * scatter kernel with its owner map in half a CU's LDS (more than 2048 and up to about 19 700 pixels: 80x60 and 160x120 of
 * a 640x480 pyramid) and at least one of them has a gradient threshold, those levels are one persistent launch in which a
 * workgroup runs a pair through all of them back to back.  PHOVO_FUSION_OFF: one launch per level, each in the geometry
 * that suits it alone (what a configuration without thresholds gets anyway).  PHOVO_FUSION_SPLIT: one launch per level in
 * the geometry of the fused launch -- bit-identical to PHOVO_FUSION_AUTO, for tests.  Iteration counts are the same in all
 * three; poses agree to the parity bar between AUTO and OFF (other summation order on the smaller levels). */
enum { PHOVO_FUSION_AUTO = 0, PHOVO_FUSION_OFF = -1, PHOVO_FUSION_SPLIT = -2 };
int phovo_engine_set_level_fusion(phovo_engine *e, int mode);
/* One arithmetic per pair.  The reference has one (CPhotoconsistencyOdometryAnalytic.h:500-563), and so has this library
 * wherever it costs little: on every level whose owner map fits LDS (up to ~39 k pixels -- every active level of the
 * shipped 4- and 5-level files on 640x480) a pair runs the SAME kernel in the same geometry whether it is aligned alone
 * through phovo_odometry_optimize or as one of thousands in a batch, so its state vector is the same bit for bit
 * (PhotoconsistencyVisualOdometry: the pair-by-pair loop and --batch write the same file).  Two switches move that line:
 *   phovo_engine_set_latency_forms(e, 1)   a handful of pairs (<= 8 / <= 32) may take the forms that finish soonest on
 *       those levels too: 512-thread workgroups for levels of <= 9.5 k pixels, the wide form from 16 384 pixels (160x120:
 *       12 instead of 27 us per iteration).  Same iteration counts, poses within the parity bar of the batch forms, last
 *       bits may differ.  Default 0.
 *   phovo_engine_set_batch_invariant(e, 1) also levels ABOVE ~39 k pixels take the batch forms for every batch size (no
 *       automatic wide form), so that a sequence cut into shards of any sizes gives bit-identical poses on every
 *       configuration.  What the sequence drivers set (apps/PhotoconsistencyVisualOdometry --batch, sequence.py,
 *       bench.py).  Default 0: one pair on 640x480 level 0 takes 22 us per iteration in the wide form, 520 us in one
 *       workgroup. */
int phovo_engine_set_latency_forms(phovo_engine *e, int on);
int phovo_engine_set_batch_invariant(phovo_engine *e, int on);
/* 1 if `level` would run in the wide form for a batch of n_pairs under the current settings. */
int phovo_engine_level_uses_wide(const phovo_engine *e, int level, int n_pairs);

/* Page-locks (and releases) a host buffer the caller will hand to the upload entry points repeatedly: uploads from
 * registered memory are direct DMA at the link rate instead of going through the runtime's bounce buffers.  Optional;
 * hipHostRegister / hipHostUnregister behind the C ABI for hosts that do not link the HIP runtime themselves, with the
 * library's own bookkeeping: registering a range that overlaps a registered one, and unregistering anything but the
 * start of a registered range, return PHOVO_E_INVALID_ARGUMENT (thread-safe). */
int phovo_host_register(void *ptr, size_t bytes);
int phovo_host_unregister(void *ptr);

/* (Re)allocates the frame pool: n_frames frames of width x height.  Uses the current config. */
int phovo_engine_reserve_frames(phovo_engine *e, int n_frames, int width, int height);
int phovo_engine_level_size(const phovo_engine *e, int level, int *width, int *height);
/* 1 if `level` is resident in the pool. */
int phovo_engine_level_is_stored(const phovo_engine *e, int level);

/* Copies one raw frame to the device and builds its pyramids there. depth may be NULL for a
 * pure target frame. */
int phovo_engine_upload_frame(phovo_engine *e, int frame, int roles,
                              const uint8_t *intensity, size_t intensity_stride,
                              const double *depth, size_t depth_stride);
/* Same with 16-bit depth (TUM / Kinect PNG) converted on the device as double(u16) * depth_scale
 * (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:163,208,220). */
int phovo_engine_upload_frame_u16(phovo_engine *e, int frame, int roles,
                                  const uint8_t *intensity, size_t intensity_stride,
                                  const uint16_t *depth, size_t depth_stride, double depth_scale);
/* Batched forms: `count` consecutive pool slots starting at first_frame, frames `*_frame_stride` BYTES
 * apart in host memory (rows `*_stride` bytes apart).  One host->device copy and one producer launch per
 * pyramid level for up to 32 frames at a time, instead of a copy, six launches and a sync per frame. */
int phovo_engine_upload_frames(phovo_engine *e, int first_frame, int count, int roles,
                               const uint8_t *intensity, size_t intensity_stride, size_t intensity_frame_stride,
                               const double *depth, size_t depth_stride, size_t depth_frame_stride);
int phovo_engine_upload_frames_u16(phovo_engine *e, int first_frame, int count, int roles,
                                   const uint8_t *intensity, size_t intensity_stride, size_t intensity_frame_stride,
                                   const uint16_t *depth, size_t depth_stride, size_t depth_frame_stride,
                                   double depth_scale);
/* Direct access to the planes of one level of one frame as fp64 (w*h doubles each, NULL = skip): lets a
 * caller supply pyramids built elsewhere (e.g. by OpenCV) or read back the device-built ones.  With a narrower
 * plane_storage, set rounds to the storage type and get returns the stored (rounded) values. */
int phovo_engine_set_level_planes(phovo_engine *e, int frame, int level,
                                  const double *intensity, const double *depth,
                                  const double *grad_x, const double *grad_y);
int phovo_engine_get_level_planes(const phovo_engine *e, int frame, int level,
                                  double *intensity, double *depth,
                                  double *grad_x, double *grad_y);

/* Optimize() for n_pairs independent (source, target) frame pairs.
 *   init_states  n_pairs x 6 (SetInitialStateVector) or NULL for all-zero
 *   out_states   n_pairs x 6 optimal state vectors
 *   reports      n_pairs entries or NULL
 * Synchronous: returns when the results are in host memory. */
int phovo_engine_align_pairs(phovo_engine *e, int n_pairs,
                             const int *source_frames, const int *target_frames,
                             const double *init_states, double *out_states,
                             phovo_pair_report *reports);
/* Split form: enqueue without waiting, then wait, then fetch.  The argument arrays are copied before the call
 * returns.  (Levels that run in the wide form synchronise the stream every 8 iterations to look at the "done" words,
 * so for them the call returns when the level has finished.)  phovo_engine_synchronize waits for everything the engine
 * has in flight; fetch_results / results_device_ptr / last_align_ms / last_launches speak of the LAST enqueue. */
int phovo_engine_enqueue_align(phovo_engine *e, int n_pairs,
                               const int *source_frames, const int *target_frames,
                               const double *init_states);
int phovo_engine_synchronize(phovo_engine *e);
int phovo_engine_fetch_results(phovo_engine *e, int n_pairs, double *out_states,
                               phovo_pair_report *reports);
/* Device pointer to the n_pairs x 6 fp64 result of the last enqueue (for an RCCL gather). */
int phovo_engine_results_device_ptr(phovo_engine *e, void **states);

/* Pipelining.  Pairs are independent, and with data-dependent termination a batch ends with a few long pairs on an
 * otherwise idle chip.  The engine therefore keeps PHOVO_ENQUEUE_DEPTH enqueues in flight, each with its own stream, pair
 * buffers and pinned mirrors: phovo_engine_enqueue_align returns at once, and the kernels of enqueue k + 1 fill the CUs
 * that the tail of enqueue k leaves free.  Every enqueue gets a ticket (1, 2, ...; phovo_engine_last_ticket right after the
 * call); a ticket stays valid -- its results fetchable -- until PHOVO_ENQUEUE_DEPTH later enqueues have been issued.  The
 * slot an enqueue takes is waited for inside phovo_engine_enqueue_align, so a caller that never looks at tickets sees the
 * behaviour of a single stream, except that two consecutive enqueues may overlap on the device.  Uploads, plane writes
 * and configuration changes wait for every enqueue in flight before they touch device memory.
 *     for (k = 0; k < steps; k++) {
 *       phovo_engine_enqueue_align(e, n, src[k], tgt[k], NULL);  t[k] = phovo_engine_last_ticket(e);
 *       if (k > 0) phovo_engine_fetch(e, t[k - 1], n, states[k - 1], NULL);      // waits for enqueue k - 1 only
 *     }
 *     phovo_engine_fetch(e, t[steps - 1], n, states[steps - 1], NULL);
 * Results do not depend on what else is in flight (same kernels, same arithmetic per pair).
 * A refused enqueue (bad argument, a level that cannot run) takes no ticket and evicts nothing; one that fails later (an
 * allocation, a launch) leaves its slot empty, and every ticket-taking call refuses it.  An enqueue of zero pairs is an
 * enqueue (ticket, nothing to fetch, no device buffer).
 * Footprint: pair data (src, tgt, states, reports: 208 B per pair) and its pinned mirrors exist per slot; the large scratch
 * -- owner maps of levels above ~39 k pixels (4 B per pixel and pair: 10 GB for 8192 pairs on 640x480 level 0), ballots,
 * the wide form's workspace -- exists ONCE as long as the caller has one enqueue in flight at a time (it changes hands
 * between the slots) and twice only while two enqueues really overlap. */
#define PHOVO_ENQUEUE_DEPTH 2
int phovo_engine_last_ticket(const phovo_engine *e);                       /* 0 before the first enqueue */
int phovo_engine_wait(phovo_engine *e, int ticket);                        /* host wait for that enqueue alone */
int phovo_engine_fetch(phovo_engine *e, int ticket, int n_pairs, double *out_states, phovo_pair_report *reports);
int phovo_engine_device_states(phovo_engine *e, int ticket, void **states);    /* n_pairs x 6 fp64 in HBM */
int phovo_engine_align_ms(const phovo_engine *e, int ticket, double *total_ms, double level_ms[PHOVO_MAX_LEVELS]);

/* Device time (ms, HIP events on the enqueue's stream) of the last enqueue: first launch to last, and per level (0 for
 * levels that were not launched; a fused launch is reported at the coarsest level it covers). */
int phovo_engine_last_align_ms(const phovo_engine *e, double *total_ms,
                               double level_ms[PHOVO_MAX_LEVELS]);
/* What the last enqueue launched, in launch order: one record per kernel launch (the wide form: per level). */
enum { PHOVO_LAUNCH_PERSISTENT = 0,       /* gn_level_kernel: one level, one workgroup per pair at a time */
       PHOVO_LAUNCH_FUSED = 1,            /* gn_fused_kernel: levels level_first..level_last (coarse to fine) per pair */
       PHOVO_LAUNCH_SLIDE = 2,            /* gn_level_kernel_slide: owner ring in LDS */
       PHOVO_LAUNCH_SLIDE_FALLBACK = 3,   /* gn_level_kernel on the pairs the sliding-window launch handed over */
       PHOVO_LAUNCH_WIDE = 4,             /* k_wide_pass1 / k_wide_pass2 per iteration, many workgroups per pair */
       PHOVO_LAUNCH_BILINEAR = 5 };       /* gn_level_kernel_bilinear (extension) */
typedef struct phovo_launch_record {
  int level_first, level_last;            /* pyramid levels the launch covers (level_first >= level_last) */
  int kind;                               /* PHOVO_LAUNCH_* */
  int threads, lds_bytes, workgroups;     /* launch geometry (workgroups: grid size) */
} phovo_launch_record;
/* count receives the number of launches; up to `capacity` of them are written to out (may be NULL). */
int phovo_engine_last_launches(const phovo_engine *e, phovo_launch_record *out, int capacity, int *count);
/* Launch geometry chosen for `level` when it is launched alone: threads per workgroup and dynamic LDS bytes. */
int phovo_engine_level_launch_info(const phovo_engine *e, int level, int *threads, int *lds_bytes,
                                   int *owner_in_lds, int *source_in_lds);

#ifdef __cplusplus
}
#endif
#endif
