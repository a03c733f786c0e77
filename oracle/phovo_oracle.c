/*
 * phovo_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, PARITY UNPINNED).
 * See phovo_oracle.h for what this restates and what pins it.
 *
 * All citations are into the reference tree (MiguelAlgaba/photoconsistency-visual-odometry).
 * Build: oracle/Makefile (gcc -O3 -mtune=native -ffp-contract=off, the reference's own
 * flags from CMakeLists.txt:58-60 plus contraction off so results do not depend on
 * the host having FMA).
 */
#include "phovo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Defaults: CPhotoconsistencyOdometryAnalytic.h:430-443                      */
/* ------------------------------------------------------------------------- */
void phovo_oracle_default_config(phovo_oracle_config *cfg)
{
  memset(cfg, 0, sizeof(*cfg));
  cfg->num_levels = 5;
  for (int l = 0; l < PHOVO_ORACLE_MAX_LEVELS; l++) {
    cfg->blur_filter_size[l] = 0;
    cfg->image_gradients_scaling_factor[l] = 0.0625;
    cfg->lambda_optimization_step[l] = 1.0;
    cfg->max_num_iterations[l] = 0;
    cfg->min_gradient_norm[l] = 300.0;
  }
  cfg->max_num_iterations[2] = 5;
  cfg->max_num_iterations[3] = 20;
  cfg->max_num_iterations[4] = 50;
  cfg->min_depth = 0.3;
  cfg->max_depth = 5.0;
}

/* ------------------------------------------------------------------------- */
/* eigenPose: CPhotoconsistencyOdometry.h:47-71                               */
/* ------------------------------------------------------------------------- */
void phovo_oracle_eigen_pose(const double s[6], double rt[16])
{
  const double x = s[0], y = s[1], z = s[2], yaw = s[3], pitch = s[4], roll = s[5];
  rt[0] = cos(yaw) * cos(pitch);
  rt[1] = cos(yaw) * sin(pitch) * sin(roll) - sin(yaw) * cos(roll);
  rt[2] = cos(yaw) * sin(pitch) * cos(roll) + sin(yaw) * sin(roll);
  rt[3] = x;
  rt[4] = sin(yaw) * cos(pitch);
  rt[5] = sin(yaw) * sin(pitch) * sin(roll) + cos(yaw) * cos(roll);
  rt[6] = sin(yaw) * sin(pitch) * cos(roll) - cos(yaw) * sin(roll);
  rt[7] = y;
  rt[8] = -sin(pitch);
  rt[9] = cos(pitch) * sin(roll);
  rt[10] = cos(pitch) * cos(roll);
  rt[11] = z;
  rt[12] = 0; rt[13] = 0; rt[14] = 0; rt[15] = 1;
}

/* ------------------------------------------------------------------------- */
/* Pyramid producers (OpenCV semantics; ...Analytic.h:115-189,466-491)        */
/* ------------------------------------------------------------------------- */
/* WHAT THIS ORACLE CANNOT VOUCH FOR.  OpenCV is not in /root/reference and not in the image: everything in this section
 * is restated from recalled OpenCV 2.4 behaviour and pinned only against oracle/numpy_twin.py, never against OpenCV.
 * The branches below that no BASELINE shape reaches are tagged UNVERIFIED-vs-OpenCV and counted when they execute
 * (phovo_oracle_unverified_hits), so that tests/test_oracle_properties.py can assert that 640x480 and 1280x960 pyramids of
 * every shipped yml stay out of them:
 *   [0] resize, scale 2: the clipped 2x2 block at an odd border (sum / count in double; OpenCV's ResizeAreaFast tail is
 *       recalled to go through the generic area path, possibly with float casts)
 *   [1] resize, scale >= 4: a tap clipped to the last row / column (weights 1 / 0; how OpenCV builds its index and
 *       coefficient tables at the border is recalled, not checked)
 *   [2] GaussianBlur (any blurFilterSize > 0): kernel coefficients, their normalisation, the row / column summation
 *       order and the border mode.  No shipped analytic yml sets blurFilterSize > 0.
 * Not counted because every run takes them: the summation order of the scale-2 area mean, of the bilinear taps and of
 * the Scharr row / column filters.  Those are restated too -- a different association changes the last bit of a plane,
 * not the algorithm -- and that is why parity tests feed the SAME pyramids (the build's own) to both sides and a caller
 * who needs OpenCV's own planes hands them over with phovo_engine_set_level_planes. */
static long g_unverified[3];
long phovo_oracle_unverified_hits(int which) { return (which >= 0 && which < 3) ? g_unverified[which] : -1; }
void phovo_oracle_unverified_reset(void) { g_unverified[0] = g_unverified[1] = g_unverified[2] = 0; }
void phovo_oracle_convert_intensity(const uint8_t *src, int n, double *dst)
{
  const double a = 1. / 255;                 /* convertTo(..., 1./255)  :471 */
  for (int i = 0; i < n; i++) dst[i] = (double)src[i] * a;
}

void phovo_oracle_level_size(int w, int h, int level, int *lw, int *lh)
{
  const double f = 1.0 / (double)(1 << level);   /* factor = factor/2 per level :161 */
  *lw = (int)rint((double)w * f);                /* saturate_cast<int>(cols*fx) = cvRound */
  *lh = (int)rint((double)h * f);
}

/* cv::resize, INTER_LINEAR, fp64, integer scale s = 2^L.
 *   s == 2: OpenCV switches to the fast INTER_AREA path, which for double runs the
 *           generic loop  sum = (((a + b) + c) + d) * 0.25  (rows y,y+1; cols x,x+1).
 *   s >= 4: bilinear at (d + 0.5) * s - 0.5 = s*d + s/2 - 0.5, i.e. taps s*d + s/2 - 1
 *           and + 1 with weights 0.5/0.5 (float coefficients, exact); horizontal pass
 *           first, then vertical:  (a*.5 + b*.5)*.5 + (c*.5 + d*.5)*.5.
 * Taps are clipped to the last row/column the way resize builds its tables. */
void phovo_oracle_resize_level(const double *src, int w, int h, int level, double *dst)
{
  int lw, lh;
  phovo_oracle_level_size(w, h, level, &lw, &lh);
  if (level == 0) {                              /* imgAux = img  :136 */
    memcpy(dst, src, sizeof(double) * (size_t)w * (size_t)h);
    return;
  }
  const int s = 1 << level;
  if (s == 2) {
    for (int dy = 0; dy < lh; dy++) {
      for (int dx = 0; dx < lw; dx++) {
        const int sx = dx * 2, sy = dy * 2;
        if (sx + 1 < w && sy + 1 < h) {
          const double a = src[(size_t)sy * w + sx], b = src[(size_t)sy * w + sx + 1];
          const double c = src[(size_t)(sy + 1) * w + sx], d = src[(size_t)(sy + 1) * w + sx + 1];
          dst[(size_t)dy * lw + dx] = (((a + b) + c) + d) * 0.25;
        } else {                                  /* clipped block at an odd border: UNVERIFIED-vs-OpenCV [0] */
          g_unverified[0]++;
          double sum = 0; int count = 0;
          for (int yy = 0; yy < 2; yy++) {
            if (sy + yy >= h) break;
            for (int xx = 0; xx < 2; xx++) {
              if (sx + xx >= w) break;
              sum += src[(size_t)(sy + yy) * w + sx + xx];
              count++;
            }
          }
          dst[(size_t)dy * lw + dx] = count ? sum / count : 0.0;
        }
      }
    }
    return;
  }
  const int half = s / 2 - 1;
  for (int dy = 0; dy < lh; dy++) {
    int sy = dy * s + half; double wy1 = 0.5;
    if (sy >= h - 1) { sy = h - 1; wy1 = 0.0; g_unverified[1]++; }      /* UNVERIFIED-vs-OpenCV [1] */
    for (int dx = 0; dx < lw; dx++) {
      int sx = dx * s + half; double wx1 = 0.5;
      if (sx >= w - 1) { sx = w - 1; wx1 = 0.0; g_unverified[1]++; }    /* UNVERIFIED-vs-OpenCV [1] */
      double top, bot;
      if (wx1 != 0.0) {
        top = src[(size_t)sy * w + sx] * 0.5 + src[(size_t)sy * w + sx + 1] * 0.5;
      } else {
        top = src[(size_t)sy * w + sx] * 1.0;
      }
      if (wy1 != 0.0) {
        if (wx1 != 0.0)
          bot = src[(size_t)(sy + 1) * w + sx] * 0.5 + src[(size_t)(sy + 1) * w + sx + 1] * 0.5;
        else
          bot = src[(size_t)(sy + 1) * w + sx] * 1.0;
        dst[(size_t)dy * lw + dx] = top * 0.5 + bot * 0.5;
      } else {
        dst[(size_t)dy * lw + dx] = top * 1.0 + top * 0.0;
      }
    }
  }
}

static inline int reflect101(int p, int len)
{
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

/* cv::Scharr(src, dst, CV_64F, dx, dy, scale, 0, BORDER_DEFAULT) for (1,0) and (0,1).
 * getScharrKernels: derivative kernel [-1,0,1], smoothing kernel [3,10,3]; the
 * smoothing kernel is the one multiplied by `scale`.  sepFilter2D runs the row
 * filter (generic RowFilter: k0*S0 + k1*S1 + k2*S2 in that order) and then the
 * column filter (SymmColumnFilter: centre*f0 + f1*(below + above), or for the
 * antisymmetric kernel f1*(below - above)). */
void phovo_oracle_scharr(const double *img, int w, int h, double scale,
                         double *gx, double *gy)
{
  const double k3 = 3.0 * scale, k10 = 10.0 * scale;
  double *tx = (double *)malloc(sizeof(double) * (size_t)w * (size_t)h);
  double *ty = (double *)malloc(sizeof(double) * (size_t)w * (size_t)h);
  for (int y = 0; y < h; y++) {
    const double *row = img + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      const double a = row[reflect101(x - 1, w)], b = row[x], c = row[reflect101(x + 1, w)];
      tx[(size_t)y * w + x] = ((-1.0 * a) + (0.0 * b)) + (1.0 * c);
      ty[(size_t)y * w + x] = ((k3 * a) + (k10 * b)) + (k3 * c);
    }
  }
  for (int y = 0; y < h; y++) {
    const int yu = reflect101(y - 1, h), yd = reflect101(y + 1, h);
    for (int x = 0; x < w; x++) {
      gx[(size_t)y * w + x] = (k10 * tx[(size_t)y * w + x]) +
                              (k3 * (tx[(size_t)yd * w + x] + tx[(size_t)yu * w + x]));
      gy[(size_t)y * w + x] = 0.0 + 1.0 * (ty[(size_t)yd * w + x] - ty[(size_t)yu * w + x]);
    }
  }
  free(tx);
  free(ty);
}

/* cv::GaussianBlur(img, img, Size(k,k), 3) twice.  getGaussianKernel(k, 3, CV_64F):
 * t_i = exp(-0.5/sigma^2 * (i-(k-1)/2)^2), normalised by 1/sum.  Row filter generic,
 * column filter symmetric, BORDER_DEFAULT (reflect-101). */
static void gaussian_blur_once(double *img, int w, int h, int ksize)
{
  if (ksize <= 1) return;
  const double sigma = 3.0;
  double *kern = (double *)malloc(sizeof(double) * (size_t)ksize);
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < ksize; i++) {
    const double x = i - (ksize - 1) * 0.5;
    const double t = exp(scale2x * x * x);
    kern[i] = t; sum += t;
  }
  sum = 1. / sum;
  for (int i = 0; i < ksize; i++) kern[i] *= sum;
  const int r = ksize / 2;
  double *tmp = (double *)malloc(sizeof(double) * (size_t)w * (size_t)h);
  for (int y = 0; y < h; y++) {
    const double *row = img + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      double s0 = kern[0] * row[reflect101(x - r, w)];
      for (int k = 1; k < ksize; k++) s0 += kern[k] * row[reflect101(x - r + k, w)];
      tmp[(size_t)y * w + x] = s0;
    }
  }
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      double s0 = kern[r] * tmp[(size_t)y * w + x];
      for (int k = 1; k <= r; k++)
        s0 += kern[r + k] * (tmp[(size_t)reflect101(y + k, h) * w + x] +
                             tmp[(size_t)reflect101(y - k, h) * w + x]);
      img[(size_t)y * w + x] = s0;
    }
  }
  free(tmp);
  free(kern);
}

void phovo_oracle_gaussian_blur_twice(double *img, int w, int h, int ksize)
{
  if (ksize <= 0) return;                        /* if( blurFilterSize>0 )  :144 */
  g_unverified[2]++;                             /* UNVERIFIED-vs-OpenCV [2]: the whole filter */
  gaussian_blur_once(img, w, h, ksize);
  gaussian_blur_once(img, w, h, ksize);
}

/* Rows of J the last ComputeResidualsAndJacobians / compute_bilinear call on this thread filled (what the device
 * path reports as phovo_pair_report.valid_pixels; the reference keeps no such count). */
static _Thread_local int g_rows_filled;

/* ------------------------------------------------------------------------- */
/* ComputeResidualsAndJacobians: ...Analytic.h:191-367                        */
/* ------------------------------------------------------------------------- */
void phovo_oracle_compute_residuals_and_jacobians(
    const phovo_oracle_level *lv, int level, const double k[9],
    const double state[6], double min_depth, double max_depth,
    double *residuals, double *jacobians, double *warped)
{
  const int nRows = lv->h, nCols = lv->w;
  const size_t n = (size_t)nRows * (size_t)nCols;

  const double scaleFactor = 1.0 / pow(2, level);           /* :203 */
  const double fx = k[0] * scaleFactor;                      /* :204 */
  const double fy = k[4] * scaleFactor;
  const double ox = k[2] * scaleFactor;
  const double oy = k[5] * scaleFactor;
  const double inv_fx = 1.f / fx;                            /* :208 */
  const double inv_fy = 1.f / fy;

  const double x = state[0], y = state[1], z = state[2];
  const double yaw = state[3], pitch = state[4], roll = state[5];

  const double sin_yaw = sin(yaw), cos_yaw = cos(yaw);       /* :220-225 */
  const double sin_pitch = sin(pitch), cos_pitch = cos(pitch);
  const double sin_roll = sin(roll), cos_roll = cos(roll);
  double Rt[16];
  Rt[0] = cos_yaw * cos_pitch;                               /* :226-241 */
  Rt[1] = cos_yaw * sin_pitch * sin_roll - sin_yaw * cos_roll;
  Rt[2] = cos_yaw * sin_pitch * cos_roll + sin_yaw * sin_roll;
  Rt[3] = x;
  Rt[4] = sin_yaw * cos_pitch;
  Rt[5] = sin_yaw * sin_pitch * sin_roll + cos_yaw * cos_roll;
  Rt[6] = sin_yaw * sin_pitch * cos_roll - cos_yaw * sin_roll;
  Rt[7] = y;
  Rt[8] = -sin_pitch;
  Rt[9] = cos_pitch * sin_roll;
  Rt[10] = cos_pitch * cos_roll;
  Rt[11] = z;
  Rt[12] = 0.0; Rt[13] = 0.0; Rt[14] = 0.0; Rt[15] = 1.0;

  const double temp1 = cos(pitch) * sin(roll);               /* :243-266 */
  const double temp2 = cos(pitch) * cos(roll);
  const double temp3 = sin(pitch);
  const double temp4 = (sin(roll) * sin(yaw) + sin(pitch) * cos(roll) * cos(yaw));
  const double temp5 = (sin(pitch) * sin(roll) * cos(yaw) - cos(roll) * sin(yaw));
  const double temp6 = (sin(pitch) * sin(roll) * sin(yaw) + cos(roll) * cos(yaw));
  const double temp7 = (-sin(pitch) * sin(roll) * sin(yaw) - cos(roll) * cos(yaw));
  const double temp8 = (sin(roll) * cos(yaw) - sin(pitch) * cos(roll) * sin(yaw));
  const double temp9 = (sin(pitch) * cos(roll) * sin(yaw) - sin(roll) * cos(yaw));
  const double temp10 = cos(pitch) * sin(roll) * cos(yaw);
  const double temp11 = cos(pitch) * cos(yaw) + x;           /* the transcription bug, kept  :253 */
  const double temp12 = cos(pitch) * cos(roll) * cos(yaw);
  const double temp13 = sin(pitch) * cos(yaw);
  const double temp14 = cos(pitch) * sin(yaw);
  const double temp15 = cos(pitch) * cos(yaw);
  const double temp16 = sin(pitch) * sin(roll);
  const double temp17 = sin(pitch) * cos(roll);
  const double temp18 = cos(pitch) * sin(roll) * sin(yaw);
  const double temp19 = cos(pitch) * cos(roll) * sin(yaw);
  const double temp20 = sin(pitch) * sin(yaw);
  const double temp21 = (cos(roll) * sin(yaw) - sin(pitch) * sin(roll) * cos(yaw));
  const double temp22 = cos(pitch) * cos(roll);
  const double temp23 = cos(pitch) * sin(roll);
  const double temp24 = cos(pitch);

  g_rows_filled = 0;
  for (int r = 0; r < nRows; r++) {                          /* raster order  :271-273 */
    for (int c = 0; c < nCols; c++) {
      const size_t i = (size_t)nCols * r + c;                /* :275 */
      double point3D[4];
      point3D[2] = lv->d0[i];                                /* :279 */
      if (min_depth < point3D[2] && point3D[2] < max_depth) {/* :280 */
        point3D[0] = (c - ox) * point3D[2] * inv_fx;         /* :282 */
        point3D[1] = (r - oy) * point3D[2] * inv_fy;         /* :283 */
        point3D[3] = 1.0;
        const double px = point3D[0], py = point3D[1], pz = point3D[2];

        double tp[4];                                        /* Rt*point3D  :291 */
        for (int a = 0; a < 4; a++)
          tp[a] = ((Rt[4 * a + 0] * point3D[0] + Rt[4 * a + 1] * point3D[1]) +
                   Rt[4 * a + 2] * point3D[2]) + Rt[4 * a + 3] * point3D[3];

        const double inv_transformedPz = 1.0 / tp[2];        /* :294 */
        const double transformed_c = (tp[0] * fx) * inv_transformedPz + ox;  /* :295 */
        const double transformed_r = (tp[1] * fy) * inv_transformedPz + oy;  /* :296 */
        /* static_cast<int>(round(.)) (:297-298).  The comparison is done on the
         * rounded double so that non-finite / huge values are simply out of bounds
         * instead of undefined behaviour in the cast. */
        const double rr = round(transformed_r), rc = round(transformed_c);
        if ((rr >= 0 && rr < (double)nRows) & (rc >= 0 && rc < (double)nCols)) {  /* :302-303 */
          const int transformed_r_int = (int)rr, transformed_c_int = (int)rc;
          const double pixel1 = lv->i0[i];                                        /* :308 */
          const double pixel2 = lv->i1[(size_t)transformed_r_int * nCols + transformed_c_int]; /* :309 */

          double J[2][6];
          const double temp25 = 1.0 / (z + py * temp1 + pz * temp2 - px * temp3); /* :313 */
          const double temp26 = temp25 * temp25;

          J[0][0] = fx * temp25;                                                  /* :317 */
          J[1][0] = 0.0;
          J[0][1] = 0.0;
          J[1][1] = fy * temp25;                                                  /* :322 */
          J[0][2] = -fx * (pz * temp4 + py * temp5 + px * temp11) * temp26;       /* :325 */
          J[1][2] = -fy * (py * temp6 + pz * temp9 + px * temp14 + y) * temp26;   /* :326 */
          J[0][3] = fx * (py * temp7 + pz * temp8 - px * temp14) * temp25;        /* :329 */
          J[1][3] = fy * (pz * temp4 + py * temp5 + px * temp15) * temp25;        /* :330 */
          J[0][4] = fx * (py * temp10 + pz * temp12 - px * temp13) * temp25       /* :333-334 */
                    - fx * (-py * temp16 - pz * temp17 - px * temp24) * (pz * temp4 + py * temp5 + px * temp11) * temp26;
          J[1][4] = fy * (py * temp18 + pz * temp19 - px * temp20) * temp25       /* :335-336 */
                    - fy * (-py * temp16 - pz * temp17 - px * temp24) * (py * temp6 + pz * temp9 + px * temp14 + y) * temp26;
          J[0][5] = fx * (py * temp4 + pz * temp21) * temp25                      /* :339-340 */
                    - fx * (py * temp22 - pz * temp23) * (pz * temp4 + py * temp5 + px * temp11) * temp26;
          J[1][5] = fy * (pz * temp7 + py * temp9) * temp25                       /* :341-342 */
                    - fy * (py * temp22 - pz * temp23) * (py * temp6 + pz * temp9 + px * temp14 + y) * temp26;

          const double gxi = lv->gx1[i];     /* gradient at the SOURCE linear index  :346 */
          const double gyi = lv->gy1[i];     /* :347 */
          for (int j = 0; j < 6; j++)        /* 1x2 * 2x6  :348 ; rows of J are column-major planes  :351-356 */
            jacobians[(size_t)j * n + i] = gxi * J[0][j] + gyi * J[1][j];
          g_rows_filled++;

          residuals[(size_t)nCols * transformed_r_int + transformed_c_int] = pixel2 - pixel1;  /* scatter  :358 */
          if (warped) warped[(size_t)nCols * transformed_r_int + transformed_c_int] = pixel1;   /* :361 */
        }
      }
    }
  }
}

/* EXTENSION, NOT IN THE REFERENCE'S ANALYTIC PATH: forward-additive residuals and Jacobians with bilinear
 * sampling of the target intensity and gradients at the real-valued warped position.  Same warp, same temps as
 * ComputeResidualsAndJacobians above; residual AND Jacobian row both live at the source index i (no scatter).
 * corrected != 0 uses temp11 = cos(pitch)*cos(yaw) (the true derivative) instead of the reference's `+x`. */
static void compute_bilinear(const phovo_oracle_level *lv, int level, const double k[9], const double state[6],
                             double min_depth, double max_depth, int corrected,
                             double *residuals, double *jacobians)
{
  const int nRows = lv->h, nCols = lv->w;
  const size_t n = (size_t)nRows * (size_t)nCols;
  const double scaleFactor = 1.0 / pow(2, level);
  const double fx = k[0] * scaleFactor, fy = k[4] * scaleFactor;
  const double ox = k[2] * scaleFactor, oy = k[5] * scaleFactor;
  const double inv_fx = 1.f / fx, inv_fy = 1.f / fy;
  const double x = state[0], y = state[1], z = state[2];
  const double yaw = state[3], pitch = state[4], roll = state[5];
  const double sy = sin(yaw), cy = cos(yaw), sp = sin(pitch), cp = cos(pitch), sr = sin(roll), cr = cos(roll);
  const double R00 = cy * cp, R01 = cy * sp * sr - sy * cr, R02 = cy * sp * cr + sy * sr;
  const double R10 = sy * cp, R11 = sy * sp * sr + cy * cr, R12 = sy * sp * cr - cy * sr;
  const double temp1 = cp * sr, temp2 = cp * cr, temp3 = sp;
  const double temp4 = (sr * sy + sp * cr * cy), temp5 = (sp * sr * cy - cr * sy);
  const double temp6 = (sp * sr * sy + cr * cy), temp7 = (-sp * sr * sy - cr * cy);
  const double temp8 = (sr * cy - sp * cr * sy), temp9 = (sp * cr * sy - sr * cy);
  const double temp10 = cp * sr * cy, temp12 = cp * cr * cy, temp13 = sp * cy;
  const double temp14 = cp * sy, temp15 = cp * cy, temp16 = sp * sr, temp17 = sp * cr;
  const double temp18 = cp * sr * sy, temp19 = cp * cr * sy, temp20 = sp * sy;
  const double temp21 = (cr * sy - sp * sr * cy), temp22 = cp * cr, temp23 = cp * sr, temp24 = cp;
  g_rows_filled = 0;
  for (int r = 0; r < nRows; r++) {
    for (int c = 0; c < nCols; c++) {
      const size_t i = (size_t)nCols * r + c;
      const double pz = lv->d0[i];
      if (!(min_depth < pz && pz < max_depth)) continue;
      const double px = (c - ox) * pz * inv_fx, py = (r - oy) * pz * inv_fy;
      const double X = ((R00 * px + R01 * py) + R02 * pz) + x;
      const double Y = ((R10 * px + R11 * py) + R12 * pz) + y;
      const double temp25 = 1.0 / (z + py * temp1 + pz * temp2 - px * temp3);
      const double temp26 = temp25 * temp25;
      const double tc = (X * fx) * temp25 + ox, tr = (Y * fy) * temp25 + oy;
      /* in bounds iff the nearest pixel is inside (the region of the reference's round() test); taps are
       * clamped to the edge row / column in the outer half-pixel band */
      if (!(tc > -0.5 && tc < (double)nCols - 0.5 && tr > -0.5 && tr < (double)nRows - 0.5)) continue;
      const double fc = floor(tc), fr = floor(tr);
      const double ax = tc - fc, ay = tr - fr;
      const int ic = (int)fc, ir = (int)fr;
      const int c0 = ic < 0 ? 0 : ic, c1 = ic + 1 > nCols - 1 ? nCols - 1 : ic + 1;
      const int r0 = ir < 0 ? 0 : ir, r1 = ir + 1 > nRows - 1 ? nRows - 1 : ir + 1;
      double smp[3];
      const double *pl[3] = {lv->i1, lv->gx1, lv->gy1};
      for (int q = 0; q < 3; q++) {
        const double *P = pl[q];
        smp[q] = (1.0 - ay) * ((1.0 - ax) * P[(size_t)r0 * nCols + c0] + ax * P[(size_t)r0 * nCols + c1]) +
                 ay * ((1.0 - ax) * P[(size_t)r1 * nCols + c0] + ax * P[(size_t)r1 * nCols + c1]);
      }
      const double A = corrected ? (pz * temp4 + py * temp5 + px * temp15 + x)
                                 : (pz * temp4 + py * temp5 + px * (temp15 + x));
      const double B = (py * temp6 + pz * temp9 + px * temp14 + y);
      const double Cc = (-py * temp16 - pz * temp17 - px * temp24);
      const double D = (py * temp22 - pz * temp23);
      double J[2][6];
      J[0][0] = fx * temp25;  J[1][0] = 0.0;
      J[0][1] = 0.0;          J[1][1] = fy * temp25;
      J[0][2] = -fx * A * temp26;
      J[1][2] = -fy * B * temp26;
      J[0][3] = fx * (py * temp7 + pz * temp8 - px * temp14) * temp25;
      J[1][3] = fy * (pz * temp4 + py * temp5 + px * temp15) * temp25;
      J[0][4] = fx * (py * temp10 + pz * temp12 - px * temp13) * temp25 - fx * Cc * A * temp26;
      J[1][4] = fy * (py * temp18 + pz * temp19 - px * temp20) * temp25 - fy * Cc * B * temp26;
      J[0][5] = fx * (py * temp4 + pz * temp21) * temp25 - fx * D * A * temp26;
      J[1][5] = fy * (pz * temp7 + py * temp9) * temp25 - fy * D * B * temp26;
      for (int j = 0; j < 6; j++) jacobians[(size_t)j * n + i] = smp[1] * J[0][j] + smp[2] * J[1][j];
      residuals[i] = smp[0] - lv->i0[i];
      g_rows_filled++;
    }
  }
}

/* 6x6 inverse by LU with partial pivoting (Eigen's fixed-size .inverse() for
 * sizes > 4 goes through PartialPivLU; ...Analytic.h:540). */
static void inverse6(const double a[36], double inv[36])
{
  double lu[36];
  int perm[6];
  memcpy(lu, a, sizeof(lu));
  for (int i = 0; i < 6; i++) perm[i] = i;
  for (int kcol = 0; kcol < 6; kcol++) {
    int piv = kcol; double best = fabs(lu[kcol * 6 + kcol]);
    for (int r = kcol + 1; r < 6; r++) {
      const double v = fabs(lu[r * 6 + kcol]);
      if (v > best) { best = v; piv = r; }
    }
    if (piv != kcol) {
      for (int c = 0; c < 6; c++) {
        const double t = lu[kcol * 6 + c]; lu[kcol * 6 + c] = lu[piv * 6 + c]; lu[piv * 6 + c] = t;
      }
      const int t = perm[kcol]; perm[kcol] = perm[piv]; perm[piv] = t;
    }
    const double d = lu[kcol * 6 + kcol];
    for (int r = kcol + 1; r < 6; r++) {
      lu[r * 6 + kcol] /= d;
      const double f = lu[r * 6 + kcol];
      for (int c = kcol + 1; c < 6; c++) lu[r * 6 + c] -= f * lu[kcol * 6 + c];
    }
  }
  for (int col = 0; col < 6; col++) {
    double yv[6];
    for (int r = 0; r < 6; r++) {               /* forward: L y = P e_col */
      double s = (perm[r] == col) ? 1.0 : 0.0;
      for (int c = 0; c < r; c++) s -= lu[r * 6 + c] * yv[c];
      yv[r] = s;
    }
    for (int r = 5; r >= 0; r--) {              /* backward: U x = y */
      double s = yv[r];
      for (int c = r + 1; c < 6; c++) s -= lu[r * 6 + c] * inv[c * 6 + col];
      inv[r * 6 + col] = s / lu[r * 6 + r];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Optimize: ...Analytic.h:500-563, TestTerminationCriteria :376-392          */
/* ------------------------------------------------------------------------- */
static int optimize_impl(const phovo_oracle_config *cfg, const double k[9],
                         const phovo_oracle_level *levels, double state[6],
                         int *iterations_per_level,
                         phovo_oracle_trace_entry *trace, int trace_capacity,
                         const double *huber_delta, int bilinear, int corrected);

int phovo_oracle_optimize(const phovo_oracle_config *cfg, const double k[9],
                          const phovo_oracle_level *levels, double state[6],
                          int *iterations_per_level,
                          phovo_oracle_trace_entry *trace, int trace_capacity)
{
  return optimize_impl(cfg, k, levels, state, iterations_per_level, trace, trace_capacity, NULL, 0, 0);
}

/* EXTENSION, NOT IN THE REFERENCE (BASELINE.json configs[4]): Optimize() with Huber IRLS weights.  Row k of the
 * normal equations gets w_k = 1 if |r_k| <= delta_L, delta_L/|r_k| otherwise: g = J^T W r, H = J^T W J.
 * huber_delta[L] <= 0 leaves level L exactly as the reference.  Used only to check the device extension. */
int phovo_oracle_optimize_huber(const phovo_oracle_config *cfg, const double k[9],
                                const phovo_oracle_level *levels, double state[6],
                                int *iterations_per_level,
                                phovo_oracle_trace_entry *trace, int trace_capacity,
                                const double *huber_delta)
{
  return optimize_impl(cfg, k, levels, state, iterations_per_level, trace, trace_capacity, huber_delta, 0, 0);
}

/* EXTENSION, NOT IN THE REFERENCE: all opt-in modes at once (Huber deltas may be NULL). */
int phovo_oracle_optimize_ext(const phovo_oracle_config *cfg, const double k[9],
                              const phovo_oracle_level *levels, double state[6],
                              int *iterations_per_level,
                              phovo_oracle_trace_entry *trace, int trace_capacity,
                              const double *huber_delta, int bilinear, int corrected)
{
  return optimize_impl(cfg, k, levels, state, iterations_per_level, trace, trace_capacity, huber_delta, bilinear,
                       corrected);
}

static int optimize_impl(const phovo_oracle_config *cfg, const double k[9],
                         const phovo_oracle_level *levels, double state[6],
                         int *iterations_per_level,
                         phovo_oracle_trace_entry *trace, int trace_capacity,
                         const double *huber_delta, int bilinear, int corrected)
{
  double gradients[6] = {0, 0, 0, 0, 0, 0};    /* m_Gradients persists across levels */
  int executed = 0;
  for (int level = cfg->num_levels - 1; level >= 0; level--) {         /* :502-503 */
    const phovo_oracle_level *lv = &levels[level];
    const size_t nPoints = (size_t)lv->w * (size_t)lv->h;               /* :505-507 */
    int iteration = 0;                                                  /* :509 */
    while (1) {
      /* residuals / jacobians are allocated and zeroed on EVERY pass, also on
       * levels that are skipped (:519-524) -- kept because it is part of the
       * reference's cost structure. */
      double *residuals = (double *)malloc(sizeof(double) * nPoints);
      double *jacobians = (double *)malloc(sizeof(double) * nPoints * 6);
      memset(residuals, 0, sizeof(double) * nPoints);
      memset(jacobians, 0, sizeof(double) * nPoints * 6);

      if (cfg->max_num_iterations[level] > 0) {                         /* :526 */
        if (bilinear)
          compute_bilinear(lv, level, k, state, cfg->min_depth, cfg->max_depth, corrected, residuals, jacobians);
        else
          phovo_oracle_compute_residuals_and_jacobians(lv, level, k, state,
              cfg->min_depth, cfg->max_depth, residuals, jacobians, NULL);

        double H[36];
        const double delta = huber_delta ? huber_delta[level] : 0.0;
        if (delta > 0.0) {        /* extension: scale row k of J by w_k once; J^T (W r) and (W J)^T J follow */
          for (size_t i = 0; i < nPoints; i++) {
            const double ar = fabs(residuals[i]);
            const double wgt = ar <= delta ? 1.0 : delta / ar;
            if (wgt != 1.0) {
              /* keep an unweighted copy in the residual's place is not needed: g uses w*J*r, H uses w*J*J */
              for (int a = 0; a < 6; a++) jacobians[(size_t)a * nPoints + i] *= sqrt(wgt);
              residuals[i] *= sqrt(wgt);
            }
          }
        }
        for (int a = 0; a < 6; a++) {                                   /* J^T r  :538 */
          const double *ja = jacobians + (size_t)a * nPoints;
          double s = 0;
          for (size_t i = 0; i < nPoints; i++) s += ja[i] * residuals[i];
          gradients[a] = s;
        }
        for (int a = 0; a < 6; a++) {                                   /* J^T J  :540 */
          const double *ja = jacobians + (size_t)a * nPoints;
          for (int b = a; b < 6; b++) {
            const double *jb = jacobians + (size_t)b * nPoints;
            double s = 0;
            for (size_t i = 0; i < nPoints; i++) s += ja[i] * jb[i];
            H[a * 6 + b] = s; H[b * 6 + a] = s;
          }
        }
        double Hinv[36], step[6];
        inverse6(H, Hinv);
        for (int a = 0; a < 6; a++) {
          double s = 0;
          for (int b = 0; b < 6; b++) s += Hinv[a * 6 + b] * gradients[b];
          step[a] = s;
        }
        for (int a = 0; a < 6; a++)                                     /* :539 */
          state[a] = state[a] - cfg->lambda_optimization_step[level] * step[a];

        if (trace && executed < trace_capacity) {
          phovo_oracle_trace_entry *e = &trace[executed];
          e->level = level; e->iteration = iteration + 1;
          memcpy(e->gradient, gradients, sizeof(gradients));
          memcpy(e->hessian, H, sizeof(H));
          memcpy(e->state, state, sizeof(double) * 6);
          e->valid_pixels = g_rows_filled;
          e->reserved = 0;
        }
        executed++;
      }
      free(residuals);
      free(jacobians);

      iteration++;                                                      /* :547 */

      /* TestTerminationCriteria  :376-392 */
      double gradientNorm = 0;
      for (int a = 0; a < 6; a++) gradientNorm += gradients[a] * gradients[a];
      gradientNorm = sqrt(gradientNorm);
      if (iteration >= cfg->max_num_iterations[level]) break;           /* :383 */
      else if (gradientNorm < cfg->min_gradient_norm[level]) break;     /* :388 */
    }
    if (iterations_per_level) iterations_per_level[level] = iteration;
  }
  return executed;
}

/* ------------------------------------------------------------------------- */
/* warpImage: CPhotoconsistencyOdometry.h:73-134                              */
/* ------------------------------------------------------------------------- */
void phovo_oracle_warp_image(const uint8_t *intensity, const double *depth,
                             int w, int h, const double rt[16], const double k[9],
                             int level, uint8_t *warped)
{
  const double fx = k[0] / pow(2, level);      /* :89-94 */
  const double fy = k[4] / pow(2, level);
  const double inv_fx = 1.f / fx;
  const double inv_fy = 1.f / fy;
  const double ox = k[2] / pow(2, level);
  const double oy = k[5] / pow(2, level);
  memset(warped, 0, (size_t)w * (size_t)h);    /* zeros  :98 */
  for (int r = 0; r < h; r++) {
    for (int c = 0; c < w; c++) {
      const double dz = depth[(size_t)r * w + c];
      if (dz > 0) {                            /* :107 */
        double p[4];
        p[2] = dz;
        p[0] = (c - ox) * p[2] * inv_fx;
        p[1] = (r - oy) * p[2] * inv_fy;
        p[3] = 1.0;
        double tp[4];
        for (int a = 0; a < 4; a++)
          tp[a] = ((rt[4 * a + 0] * p[0] + rt[4 * a + 1] * p[1]) + rt[4 * a + 2] * p[2]) + rt[4 * a + 3] * p[3];
        /* static_cast<int>: truncation toward zero, not rounding  :119-122 */
        const double tcd = ((tp[0] * fx) / tp[2]) + ox;
        const double trd = ((tp[1] * fy) / tp[2]) + oy;
        if (!(tcd > -2147483648.0 && tcd < 2147483647.0 && trd > -2147483648.0 && trd < 2147483647.0))
          continue;                            /* non-finite / huge: treated as out of bounds */
        const int tc = (int)tcd, tr = (int)trd;
        if (tr >= 0 && tr < h && tc >= 0 && tc < w)
          warped[(size_t)tr * w + tc] = intensity[(size_t)r * w + c];
      }
    }
  }
}
