// Minimal fixed-size matrix types used when Eigen is not available.
//
// The reference's phovo/include/Matrix.h:44-483 derives ~20 typedef-classes from Eigen::Matrix; the
// alignment path and the two apps only need a few operations of four of them (Matrix33RowMajor,
// Matrix44RowMajor, VectorCol3/4/6): element access with operator(), comma initialisation
// (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:68-71), Identity/Zero,
// product, 4x4 inverse, 3x3 block -> quaternion (apps/PhotoconsistencyVisualOdometry/
// PhotoconsistencyVisualOdometry.cpp:233-237) and stream output.  Storage is row-major.
#ifndef PHOVO_COMPAT_NUMERIC_H
#define PHOVO_COMPAT_NUMERIC_H

#include <cmath>
#include <cstddef>
#include <ostream>

namespace phovo {
namespace Numeric {

template <class T, int R, int C>
class FixedMatrixRowMajor {
 public:
  typedef T Scalar;
  FixedMatrixRowMajor() { for (int i = 0; i < R * C; i++) m_[i] = T(0); }

  T &operator()(int r, int c) { return m_[r * C + c]; }
  const T &operator()(int r, int c) const { return m_[r * C + c]; }
  T &operator()(int i) { return m_[i]; }                 // vectors (and linear access)
  const T &operator()(int i) const { return m_[i]; }
  T *data() { return m_; }
  const T *data() const { return m_; }
  static int rows() { return R; }
  static int cols() { return C; }

  static FixedMatrixRowMajor Zero() { return FixedMatrixRowMajor(); }
  static FixedMatrixRowMajor Identity()
  {
    FixedMatrixRowMajor m;
    for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = T(1);
    return m;
  }
  void setZero() { *this = Zero(); }

  // `m << a, b, c, ...;`  (row-major fill, as Eigen's comma initialiser)
  class CommaInit {
   public:
    CommaInit(FixedMatrixRowMajor &m, T first) : m_(m), n_(0) { m_.m_[n_++] = first; }
    CommaInit &operator,(T v) { if (n_ < R * C) m_.m_[n_++] = v; return *this; }
   private:
    FixedMatrixRowMajor &m_;
    int n_;
  };
  CommaInit operator<<(T first) { return CommaInit(*this, first); }

  template <int C2>
  FixedMatrixRowMajor<T, R, C2> operator*(const FixedMatrixRowMajor<T, C, C2> &b) const
  {
    FixedMatrixRowMajor<T, R, C2> out;
    for (int i = 0; i < R; i++)
      for (int j = 0; j < C2; j++) {
        T s = T(0);
        for (int k = 0; k < C; k++) s += (*this)(i, k) * b(k, j);
        out(i, j) = s;
      }
    return out;
  }
  FixedMatrixRowMajor &operator*=(const FixedMatrixRowMajor<T, C, C> &b) { *this = (*this) * b; return *this; }

  FixedMatrixRowMajor<T, C, R> transpose() const
  {
    FixedMatrixRowMajor<T, C, R> t;
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t(j, i) = (*this)(i, j);
    return t;
  }

  // General inverse by Gauss-Jordan with partial pivoting (square matrices only).
  FixedMatrixRowMajor inverse() const
  {
    static_assert(R == C, "inverse() needs a square matrix");
    FixedMatrixRowMajor a = *this, inv = Identity();
    for (int k = 0; k < R; k++) {
      int piv = k;
      for (int r = k + 1; r < R; r++) if (std::fabs(a(r, k)) > std::fabs(a(piv, k))) piv = r;
      if (piv != k) for (int c = 0; c < C; c++) { T t = a(k, c); a(k, c) = a(piv, c); a(piv, c) = t; t = inv(k, c); inv(k, c) = inv(piv, c); inv(piv, c) = t; }
      const T d = a(k, k);
      for (int c = 0; c < C; c++) { a(k, c) /= d; inv(k, c) /= d; }
      for (int r = 0; r < R; r++) {
        if (r == k) continue;
        const T f = a(r, k);
        for (int c = 0; c < C; c++) { a(r, c) -= f * a(k, c); inv(r, c) -= f * inv(k, c); }
      }
    }
    return inv;
  }

  template <int BR, int BC>
  FixedMatrixRowMajor<T, BR, BC> block(int r0, int c0) const
  {
    FixedMatrixRowMajor<T, BR, BC> b;
    for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) b(i, j) = (*this)(r0 + i, c0 + j);
    return b;
  }

 private:
  T m_[R * C];
};

template <class T, int R, int C>
std::ostream &operator<<(std::ostream &os, const FixedMatrixRowMajor<T, R, C> &m)
{
  for (int i = 0; i < R; i++) {
    for (int j = 0; j < C; j++) os << (j ? " " : "") << m(i, j);
    if (i + 1 < R) os << "\n";
  }
  return os;
}

template <class T> using Matrix33RowMajor = FixedMatrixRowMajor<T, 3, 3>;   // Matrix.h:314-333
template <class T> using Matrix44RowMajor = FixedMatrixRowMajor<T, 4, 4>;   // Matrix.h:271-290
template <class T> using VectorCol3 = FixedMatrixRowMajor<T, 3, 1>;
template <class T> using VectorCol4 = FixedMatrixRowMajor<T, 4, 1>;        // Matrix.h:421-440
template <class T> using VectorCol6 = FixedMatrixRowMajor<T, 6, 1>;        // Matrix.h:378-397

// Rotation matrix -> unit quaternion, the algorithm of Eigen::Quaternion(const Matrix3&)
// (used at ...VisualOdometry.cpp:237).
template <class T>
struct Quaternion {
  T x_, y_, z_, w_;
  explicit Quaternion(const Matrix33RowMajor<T> &R)
  {
    const T t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > T(0)) {
      T s = std::sqrt(t + T(1));
      w_ = T(0.5) * s;
      s = T(0.5) / s;
      x_ = (R(2, 1) - R(1, 2)) * s;
      y_ = (R(0, 2) - R(2, 0)) * s;
      z_ = (R(1, 0) - R(0, 1)) * s;
    } else {
      int i = 0;
      if (R(1, 1) > R(0, 0)) i = 1;
      if (R(2, 2) > R(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      T s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + T(1));
      T q[3];
      q[i] = T(0.5) * s;
      s = T(0.5) / s;
      w_ = (R(k, j) - R(j, k)) * s;
      q[j] = (R(j, i) + R(i, j)) * s;
      q[k] = (R(k, i) + R(i, k)) * s;
      x_ = q[0]; y_ = q[1]; z_ = q[2];
    }
  }
  T x() const { return x_; }
  T y() const { return y_; }
  T z() const { return z_; }
  T w() const { return w_; }
};

}  // namespace Numeric
}  // namespace phovo
#endif
