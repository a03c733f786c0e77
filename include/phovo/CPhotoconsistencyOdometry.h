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
#include <stdexcept>
#include <string>
#include <type_traits>

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
// Runs on the device (phovo_warp_image, csrc/warp_kernels.hip); like the class it is instantiable for
// <unsigned char, double>, which is what both reference apps use, and throws when the device call fails.
template <class TPixel, class TCoordinate>
void warpImage(const compat::Mat_<TPixel> &intensityImage, const compat::Mat_<TCoordinate> &depthImage,
               compat::Mat_<TPixel> &warpedIntensityImage, const Numeric::Matrix44RowMajor<TCoordinate> &Rt,
               const Numeric::Matrix33RowMajor<TCoordinate> &intrinsicMatrix, const int level = 0,
               const int device = 0)
{
  static_assert(std::is_same<TPixel, unsigned char>::value && std::is_same<TCoordinate, double>::value,
                "the MI355X path implements warpImage<unsigned char, double>");
  if (depthImage.rows != intensityImage.rows || depthImage.cols != intensityImage.cols)
    throw std::runtime_error("warpImage: intensity and depth sizes differ");
  warpedIntensityImage = compat::Mat_<TPixel>::zeros(intensityImage.rows, intensityImage.cols);   // :98
  const int status = phovo_warp_image(device, intensityImage.data, intensityImage.step, depthImage.data,
                                      depthImage.step, intensityImage.cols, intensityImage.rows, Rt.data(),
                                      intrinsicMatrix.data(), level, warpedIntensityImage.data,
                                      warpedIntensityImage.step);
  if (status != PHOVO_OK) throw std::runtime_error(std::string("warpImage: ") + phovo_last_error());
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
