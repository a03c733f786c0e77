/*
 * phovo_oracle.h -- CPU oracle for the analytic Gauss-Newton RGB-D alignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference (MiguelAlgaba/photoconsistency-visual-odometry)
 * ships no tests, golden vectors or fixtures, and it cannot be compiled in this
 * image (OpenCV, Eigen and Boost are absent).  This file is a plain-C, fp64
 * restatement of
 *   phovo/include/CPhotoconsistencyOdometryAnalytic.h:115-189  (pyramids, via OpenCV semantics)
 *   phovo/include/CPhotoconsistencyOdometryAnalytic.h:191-367  (ComputeResidualsAndJacobians)
 *   phovo/include/CPhotoconsistencyOdometryAnalytic.h:376-392  (TestTerminationCriteria)
 *   phovo/include/CPhotoconsistencyOdometryAnalytic.h:500-563  (Optimize)
 *   phovo/include/CPhotoconsistencyOdometry.h:47-71            (eigenPose)
 *   phovo/include/CPhotoconsistencyOdometry.h:73-134           (warpImage)
 * What pins it instead: an independent numpy restatement (oracle/numpy_twin.py)
 * whose outputs are committed under tests/golden/, and a symbolic known-answer
 * test of the warp Jacobian against the model in
 * phovo/Maxima/derivatives_photoconsistency.wxm:5-19 (tests/test_jacobian_kat.py).
 * The arithmetic that lives in Eigen / OpenCV (matrix products, 6x6 inverse,
 * resize, Scharr) is restated from those libraries' documented behaviour; the
 * summation order inside Eigen's products is not reproduced bit for bit.
 */
#ifndef PHOVO_ORACLE_H
#define PHOVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHOVO_ORACLE_MAX_LEVELS 16

/* Per-level optimisation parameters: the vectors filled by ReadConfigurationFile
 * (...Analytic.h:581-607) or by the constructor defaults (:430-443). */
typedef struct phovo_oracle_config {
  int    num_levels;
  int    blur_filter_size[PHOVO_ORACLE_MAX_LEVELS];
  double image_gradients_scaling_factor[PHOVO_ORACLE_MAX_LEVELS];
  double lambda_optimization_step[PHOVO_ORACLE_MAX_LEVELS];
  int    max_num_iterations[PHOVO_ORACLE_MAX_LEVELS];
  double min_gradient_norm[PHOVO_ORACLE_MAX_LEVELS];
  double min_depth;   /* m_MinDepth, default 0.3 (:430) */
  double max_depth;   /* m_MaxDepth, default 5.0 (:430) */
} phovo_oracle_config;

/* One pyramid level of one alignment problem: five row-major fp64 planes. */
typedef struct phovo_oracle_level {
  int w, h;
  const double *i0;   /* source intensity   m_IntensityPyramid0[L]          */
  const double *d0;   /* source depth       m_DepthPyramid0[L]              */
  const double *i1;   /* target intensity   m_IntensityPyramid1[L]          */
  const double *gx1;  /* target gradient x  m_IntensityGradientXPyramid1[L] */
  const double *gy1;  /* target gradient y  m_IntensityGradientYPyramid1[L] */
} phovo_oracle_level;

/* One record per executed GN iteration (levels with max_num_iterations == 0
 * produce none). */
typedef struct phovo_oracle_trace_entry {
  int    level;
  int    iteration;      /* m_Iteration after the increment (:547) */
  double gradient[6];    /* J^T r (:538) */
  double hessian[36];    /* J^T J, row-major (:540) */
  double state[6];       /* state after the update (:539-540) */
  int    valid_pixels;   /* rows of J this pass filled: depth gate (:280) and bounds (:302-303) passed */
  int    reserved;
} phovo_oracle_trace_entry;

void phovo_oracle_default_config(phovo_oracle_config *cfg);

void phovo_oracle_eigen_pose(const double state[6], double rt[16]);

/* intensity u8 -> fp64 * (1./255)  (...Analytic.h:471,484) */
void phovo_oracle_convert_intensity(const uint8_t *src, int n, double *dst);

/* Level size = round(w * 2^-L) as cv::resize(Size(0,0), f, f) does (:132). */
void phovo_oracle_level_size(int w, int h, int level, int *lw, int *lh);

/* cv::resize(img, out, Size(0,0), 2^-L, 2^-L) with the default INTER_LINEAR
 * for fp64 images, always from level 0 (:126-137). dst has level_size elements. */
void phovo_oracle_resize_level(const double *src, int w, int h, int level, double *dst);

/* cv::GaussianBlur(img, img, Size(k,k), 3) applied twice (:146-147). In place. */
void phovo_oracle_gaussian_blur_twice(double *img, int w, int h, int ksize);

/* cv::Scharr dx and dy with scale, delta 0, BORDER_DEFAULT (:181-187). */
void phovo_oracle_scharr(const double *img, int w, int h, double scale,
                         double *gx, double *gy);

/* ...Analytic.h:191-367.  residuals has n = w*h entries, jacobians is n x 6
 * COLUMN-major (Matrix.h:114-121).  Both must be zeroed by the caller exactly
 * as Optimize does (:519-524).  warped (n entries, may be NULL) receives the
 * forward-warped source intensities (:359-362). */
void phovo_oracle_compute_residuals_and_jacobians(
    const phovo_oracle_level *lv, int level, const double k[9],
    const double state[6], double min_depth, double max_depth,
    double *residuals, double *jacobians, double *warped);

/* ...Analytic.h:500-563.  levels[L] for L = 0..num_levels-1.  state is the
 * initial state on entry (SetInitialStateVector) and the optimum on return.
 * iterations_per_level (may be NULL) receives m_Iteration at the end of each
 * level.  trace (may be NULL) receives up to trace_capacity entries; the return
 * value is the number of GN iterations executed (may exceed trace_capacity). */
int phovo_oracle_optimize(const phovo_oracle_config *cfg, const double k[9],
                          const phovo_oracle_level *levels, double state[6],
                          int *iterations_per_level,
                          phovo_oracle_trace_entry *trace, int trace_capacity);

/* EXTENSION, NOT IN THE REFERENCE: Optimize() with Huber IRLS weights (see phovo_oracle.c). */
int phovo_oracle_optimize_huber(const phovo_oracle_config *cfg, const double k[9],
                                const phovo_oracle_level *levels, double state[6],
                                int *iterations_per_level,
                                phovo_oracle_trace_entry *trace, int trace_capacity,
                                const double *huber_delta);

/* EXTENSION, NOT IN THE REFERENCE: + bilinear forward-additive sampling, optionally with the corrected Jacobian. */
int phovo_oracle_optimize_ext(const phovo_oracle_config *cfg, const double k[9],
                              const phovo_oracle_level *levels, double state[6],
                              int *iterations_per_level,
                              phovo_oracle_trace_entry *trace, int trace_capacity,
                              const double *huber_delta, int bilinear, int corrected);

/* CPhotoconsistencyOdometry.h:73-134 (truncating cast, depth>0 gate). */
void phovo_oracle_warp_image(const uint8_t *intensity, const double *depth,
                             int w, int h, const double rt[16], const double k[9],
                             int level, uint8_t *warped);

/* How often the branches tagged UNVERIFIED-vs-OpenCV in phovo_oracle.c have executed since the last reset (process-wide,
 * not thread-safe: the pyramid producers run on one thread in every test): 0 = scale-2 resize, clipped block at an odd
 * border; 1 = scale >= 4 resize, tap clipped to the last row / column; 2 = GaussianBlur.  -1 for any other index. */
long phovo_oracle_unverified_hits(int which);
void phovo_oracle_unverified_reset(void);

#ifdef __cplusplus
}
#endif
#endif
