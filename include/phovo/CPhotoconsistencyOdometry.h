// Abstract alignment surface + eigenPose + warpImage, for builds WITHOUT OpenCV / Eigen.
//
// With OpenCV and Eigen installed, use the reference's own phovo/include/CPhotoconsistencyOdometry.h
// (define PHOVO_HIP_USE_REFERENCE_TYPES before including phovo/CPhotoconsistencyOdometryAnalytic.h,
// see INTEGRATION.md): the class below then is not needed.  Without them, this header supplies the
// same interface (reference: CPhotoconsistencyOdometry.h:137-179) on the small containers of
// phovo/compat/.
#ifndef PHOVO_HIP_CPHOTOCONSISTENCY_ODOMETRY_H
#define PHOVO_HIP_CPHOTOCONSISTENCY_ODOMETRY_H

#include <cmath>

#include "phovo/compat/Image.h"
#include "phovo/compat/Numeric.h"
#include "phovo_hip.h"

namespace phovo {

// (x, y, z, yaw, pitch, roll) -> 4x4 rigid transform, R = Rz(yaw) Ry(pitch) Rx(roll)
// (reference: CPhotoconsistencyOdometry.h:47-71).
template <class T>
void eigenPose(const T x, const T y, const T z, const T yaw, const T pitch, const T roll,
               Numeric::Matrix44RowMajor<T> &pose)
{
  const double s[6] = {(double)x, (double)y, (double)z, (double)yaw, (double)pitch, (double)roll};
  double rt[16];
  phovo_eigen_pose(s, rt);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) pose(i, j) = (T)rt[4 * i + j];
}

// Forward-warps the source intensities into the target view for display
// (reference: CPhotoconsistencyOdometry.h:73-134): depth > 0 gate, truncating cast, last writer wins.
template <class TPixel, class TCoordinate>
void warpImage(const compat::Mat_<TPixel> &intensityImage, const compat::Mat_<TCoordinate> &depthImage,
               compat::Mat_<TPixel> &warpedIntensityImage, const Numeric::Matrix44RowMajor<TCoordinate> &Rt,
               const Numeric::Matrix33RowMajor<TCoordinate> &intrinsicMatrix, const int level = 0)
{
  const TCoordinate s = (TCoordinate)std::pow(2, level);
  const TCoordinate fx = intrinsicMatrix(0, 0) / s, fy = intrinsicMatrix(1, 1) / s;
  const TCoordinate ox = intrinsicMatrix(0, 2) / s, oy = intrinsicMatrix(1, 2) / s;
  const TCoordinate inv_fx = 1.f / fx, inv_fy = 1.f / fy;
  const int H = intensityImage.rows, W = intensityImage.cols;
  warpedIntensityImage = compat::Mat_<TPixel>::zeros(H, W);
  for (int r = 0; r < H; r++) {
    for (int c = 0; c < W; c++) {
      const TCoordinate d = depthImage(r, c);
      if (!(d > 0)) continue;
      const TCoordinate p[4] = {(c - ox) * d * inv_fx, (r - oy) * d * inv_fy, d, 1};
      TCoordinate q[3];
      for (int a = 0; a < 3; a++)
        q[a] = ((Rt(a, 0) * p[0] + Rt(a, 1) * p[1]) + Rt(a, 2) * p[2]) + Rt(a, 3) * p[3];
      const TCoordinate tcd = ((q[0] * fx) / q[2]) + ox, trd = ((q[1] * fy) / q[2]) + oy;
      if (!(tcd > -2147483648.0 && tcd < 2147483647.0 && trd > -2147483648.0 && trd < 2147483647.0)) continue;
      const int tc = (int)tcd, tr = (int)trd;          // truncation, not rounding
      if (tr >= 0 && tr < H && tc >= 0 && tc < W) warpedIntensityImage(tr, tc) = intensityImage(r, c);
    }
  }
}

template <class TPixel, class TCoordinate>
class CPhotoconsistencyOdometry {
 public:
  typedef TPixel PixelType;
  typedef compat::Mat_<PixelType> IntensityImageType;
  typedef TCoordinate CoordinateType;
  typedef compat::Mat_<CoordinateType> DepthImageType;
  typedef Numeric::Matrix33RowMajor<CoordinateType> Matrix33Type;
  typedef Numeric::Matrix44RowMajor<CoordinateType> Matrix44Type;
  typedef Numeric::VectorCol6<CoordinateType> Vector6Type;
  typedef Numeric::VectorCol4<CoordinateType> Vector4Type;

  virtual ~CPhotoconsistencyOdometry() {}
  virtual void SetIntrinsicMatrix(const Matrix33Type &intrinsicMatrix) = 0;
  virtual void SetSourceFrame(const IntensityImageType &intensityImage, const DepthImageType &depthImage) = 0;
  virtual void SetTargetFrame(const IntensityImageType &intensityImage, const DepthImageType &depthImage) = 0;
  virtual void SetInitialStateVector(const Vector6Type &initialStateVector) = 0;
  virtual void Optimize() = 0;
  virtual Vector6Type GetOptimalStateVector() const = 0;
  virtual Matrix44Type GetOptimalRigidTransformationMatrix() const = 0;
};

}  // namespace phovo
#endif
