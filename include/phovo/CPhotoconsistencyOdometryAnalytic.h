// Drop-in for the reference's phovo/include/CPhotoconsistencyOdometryAnalytic.h: the same namespace,
// class name, template parameters and public methods (reference :428-607), but every method
// forwards to the MI355X library through the C ABI of phovo_hip.h -- nothing is computed on the host.
//
//   #include "phovo/CPhotoconsistencyOdometryAnalytic.h"
//   phovo::Analytic::CPhotoconsistencyOdometryAnalytic<unsigned char, double> odometry;
//
// Two ways to get the image / matrix types:
//   * default: the small containers of phovo/compat/ (no OpenCV, no Eigen needed);
//   * PHOVO_HIP_USE_REFERENCE_TYPES: include the reference's own CPhotoconsistencyOdometry.h
//     (cv::Mat_, Eigen-based Matrix.h) first; this class then derives from the reference's abstract
//     base and the reference's apps compile against it unchanged.
// Only <unsigned char, double> is instantiable: the device path is fp64 on u8 intensities, which
// is what both reference apps use (...FrameAlignment.cpp:58-63, ...VisualOdometry.cpp:122-131).
//
// The reference's methods return void and report nothing; here a failed call throws
// std::runtime_error carrying phovo_last_error() (e.g. no GPU present: there is no CPU path).
#ifndef PHOVO_HIP_CPHOTOCONSISTENCY_ODOMETRY_ANALYTIC_H
#define PHOVO_HIP_CPHOTOCONSISTENCY_ODOMETRY_ANALYTIC_H

#include <stdexcept>
#include <string>
#include <type_traits>

#ifdef PHOVO_HIP_USE_REFERENCE_TYPES
#include "CPhotoconsistencyOdometry.h"          // the reference's header (needs OpenCV + Eigen)
#else
#include "phovo/CPhotoconsistencyOdometry.h"
#endif
#include "phovo_hip.h"

namespace phovo {
namespace Analytic {

template <class TPixel, class TCoordinate>
class CPhotoconsistencyOdometryAnalytic : public CPhotoconsistencyOdometry<TPixel, TCoordinate> {
  static_assert(std::is_same<TPixel, unsigned char>::value && std::is_same<TCoordinate, double>::value,
                "the MI355X path implements CPhotoconsistencyOdometryAnalytic<unsigned char, double>");

 public:
  typedef CPhotoconsistencyOdometry<TPixel, TCoordinate> Superclass;
  typedef typename Superclass::CoordinateType CoordinateType;
  typedef typename Superclass::IntensityImageType IntensityImageType;
  typedef typename Superclass::DepthImageType DepthImageType;
  typedef typename Superclass::Matrix33Type Matrix33Type;
  typedef typename Superclass::Matrix44Type Matrix44Type;
  typedef typename Superclass::Vector6Type Vector6Type;
  typedef typename Superclass::Vector4Type Vector4Type;

  explicit CPhotoconsistencyOdometryAnalytic(int device = 0) : m_Handle(nullptr)
  {
    Check(phovo_odometry_create(device, &m_Handle), "CPhotoconsistencyOdometryAnalytic()");
  }
  ~CPhotoconsistencyOdometryAnalytic() { phovo_odometry_destroy(m_Handle); }
  CPhotoconsistencyOdometryAnalytic(const CPhotoconsistencyOdometryAnalytic &) = delete;
  CPhotoconsistencyOdometryAnalytic &operator=(const CPhotoconsistencyOdometryAnalytic &) = delete;

  void SetMinDepth(const CoordinateType minD) { Check(phovo_odometry_set_min_depth(m_Handle, minD), "SetMinDepth"); }
  void SetMaxDepth(const CoordinateType maxD) { Check(phovo_odometry_set_max_depth(m_Handle, maxD), "SetMaxDepth"); }

  void SetIntrinsicMatrix(const Matrix33Type &intrinsicMatrix)
  {
    double k[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) k[3 * i + j] = intrinsicMatrix(i, j);
    Check(phovo_odometry_set_intrinsic_matrix(m_Handle, k), "SetIntrinsicMatrix");
  }

  void SetSourceFrame(const IntensityImageType &intensityImage, const DepthImageType &depthImage)
  {
    if (intensityImage.rows != depthImage.rows || intensityImage.cols != depthImage.cols)
      throw std::runtime_error("SetSourceFrame: intensity and depth sizes differ");
    Check(phovo_odometry_set_source_frame(m_Handle, intensityImage.data, static_cast<size_t>(intensityImage.step),
                                          reinterpret_cast<const double *>(depthImage.data),
                                          static_cast<size_t>(depthImage.step),
                                          intensityImage.cols, intensityImage.rows), "SetSourceFrame");
  }

  // "Depth image is ignored" (reference :478).
  void SetTargetFrame(const IntensityImageType &intensityImage, const DepthImageType &depthImage)
  {
    Check(phovo_odometry_set_target_frame(m_Handle, intensityImage.data, static_cast<size_t>(intensityImage.step),
                                          reinterpret_cast<const double *>(depthImage.data),
                                          static_cast<size_t>(depthImage.step),
                                          intensityImage.cols, intensityImage.rows), "SetTargetFrame");
  }

  void SetInitialStateVector(const Vector6Type &initialStateVector)
  {
    double s[6];
    for (int i = 0; i < 6; i++) s[i] = initialStateVector(i);
    Check(phovo_odometry_set_initial_state_vector(m_Handle, s), "SetInitialStateVector");
  }

  void Optimize() { Check(phovo_odometry_optimize(m_Handle), "Optimize"); }

  Vector6Type GetOptimalStateVector() const
  {
    double s[6];
    Check(phovo_odometry_get_optimal_state_vector(m_Handle, s), "GetOptimalStateVector");
    Vector6Type v;
    for (int i = 0; i < 6; i++) v(i) = s[i];
    return v;
  }

  Matrix44Type GetOptimalRigidTransformationMatrix() const
  {
    double rt[16];
    Check(phovo_odometry_get_optimal_rigid_transformation_matrix(m_Handle, rt), "GetOptimalRigidTransformationMatrix");
    Matrix44Type m;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = rt[4 * i + j];
    return m;
  }

  void ReadConfigurationFile(const std::string &fileName)
  {
    Check(phovo_odometry_read_configuration_file(m_Handle, fileName.c_str()), "ReadConfigurationFile");
  }

  // Not in the reference: narrow plane storage / Huber IRLS weights (phovo_extensions in phovo_hip.h).
  // ReadConfigurationFile() also picks them up from the two optional yml keys.
  void SetExtensions(const phovo_extensions &ext) { Check(phovo_odometry_set_extensions(m_Handle, &ext), "SetExtensions"); }
  // Not in the reference: Optimize() may take the forms that finish soonest for one pair (last bits may then differ from the
  // same pair aligned in a batch); default off.
  void SetLatencyForms(bool on) { Check(phovo_odometry_set_latency_forms(m_Handle, on ? 1 : 0), "SetLatencyForms"); }

  // Not in the reference: what Optimize() did (iterations per level, last gradient norm, flags) and
  // its device time.
  phovo_pair_report GetReport() const
  {
    phovo_pair_report r;
    Check(phovo_odometry_get_report(m_Handle, &r), "GetReport");
    return r;
  }
  double GetLastOptimizeMilliseconds() const
  {
    double ms = 0;
    Check(phovo_odometry_last_optimize_ms(m_Handle, &ms), "GetLastOptimizeMilliseconds");
    return ms;
  }

 private:
  static void Check(int status, const char *where)
  {
    if (status != PHOVO_OK)
      throw std::runtime_error(std::string(where) + ": " + phovo_status_string(status) + " -- " + phovo_last_error());
  }
  phovo_odometry *m_Handle;
};

}  // namespace Analytic
}  // namespace phovo
#endif
