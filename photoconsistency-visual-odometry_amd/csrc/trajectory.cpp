// Host-side tail of the VisualOdometry loop (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:
// 233-243): pose *= Rt^-1, quaternion from the rotation block, one trajectory line per pair.  No GPU work; it lives
// behind the C ABI so that the app's pair-by-pair loop, its --batch path and the multi-rank sequence driver
// (photoconsistency-visual-odometry_amd/sequence.py) all print through the same arithmetic.
#include <iomanip>
#include <limits>
#include <sstream>
#include <string>

#include "phovo/compat/Numeric.h"
#include "phovo_internal.hpp"

using namespace phovo_hip;

typedef phovo::Numeric::Matrix44RowMajor<double> Matrix44;
typedef phovo::Numeric::Matrix33RowMajor<double> Matrix33;

extern "C" {

int phovo_trajectory_chain(int n_pairs, const double *states, double pose_io[16], double *poses_out)
{
  if (n_pairs < 0 || !pose_io || (n_pairs > 0 && !states)) return fail(PHOVO_E_INVALID_ARGUMENT, "trajectory_chain: null");
  Matrix44 pose;
  for (int i = 0; i < 16; i++) pose(i) = pose_io[i];
  for (int p = 0; p < n_pairs; p++) {
    double rt[16];
    const int st = phovo_eigen_pose(states + (size_t)p * 6, rt);                  // GetOptimalRigidTransformationMatrix  :232
    if (st != PHOVO_OK) return st;
    Matrix44 Rt;
    for (int i = 0; i < 16; i++) Rt(i) = rt[i];
    pose *= Rt.inverse();                                                         // :233-234
    if (poses_out) for (int i = 0; i < 16; i++) poses_out[(size_t)p * 16 + i] = pose(i);
  }
  for (int i = 0; i < 16; i++) pose_io[i] = pose(i);
  return PHOVO_OK;
}

int phovo_trajectory_format_pose(double timestamp, const double pose[16], char *line, size_t capacity)
{
  if (!pose || !line) return fail(PHOVO_E_INVALID_ARGUMENT, "trajectory_format_pose: null");
  Matrix33 R;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = pose[4 * i + j];
  const phovo::Numeric::Quaternion<double> q(R);                                  // :237
  std::ostringstream os;
  os << std::setprecision(std::numeric_limits<double>::digits10 + 1) << timestamp << " "      // :240-243
     << pose[3] << " " << pose[7] << " " << pose[11] << " "
     << q.x() << " " << q.y() << " " << q.z() << " " << q.w();
  const std::string s = os.str();
  if (s.size() + 1 > capacity) return fail(PHOVO_E_INVALID_ARGUMENT, "trajectory_format_pose: buffer too small");
  s.copy(line, s.size());
  line[s.size()] = '\0';
  return PHOVO_OK;
}

}  // extern "C"
