// Minimal image container used when OpenCV is not available: the subset of cv::Mat_<T> the alignment
// path and the apps touch (rows, cols, step, data, (r,c) access, zeros, deep copy).  Rows are
// contiguous; `step` is the row pitch in bytes, as in cv::Mat.
#ifndef PHOVO_COMPAT_IMAGE_H
#define PHOVO_COMPAT_IMAGE_H

#include <cstddef>
#include <memory>
#include <vector>

namespace phovo {
namespace compat {

template <class T>
class Mat_ {
 public:
  typedef T value_type;
  int rows, cols;
  size_t step;
  T *data;

  Mat_() : rows(0), cols(0), step(0), data(nullptr) {}
  Mat_(int r, int c) { create(r, c); }
  static Mat_ zeros(int r, int c) { return Mat_(r, c); }

  void create(int r, int c)
  {
    rows = r; cols = c; step = sizeof(T) * (size_t)c;
    store_ = std::make_shared<std::vector<T> >((size_t)r * (size_t)c, T(0));
    data = store_->data();
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  T &operator()(int r, int c) { return data[(size_t)r * cols + c]; }
  const T &operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
  T &operator()(int i) { return data[i]; }
  const T &operator()(int i) const { return data[i]; }

  Mat_ clone() const
  {
    Mat_ m;
    if (!empty()) { m.create(rows, cols); *m.store_ = *store_; }
    return m;
  }
  template <class U>
  Mat_<U> convertTo(double scale) const            // cv::Mat::convertTo(type, alpha)
  {
    Mat_<U> m(rows, cols);
    for (size_t i = 0; i < (size_t)rows * cols; i++) m.data[i] = (U)((double)data[i] * scale);
    return m;
  }

 private:
  std::shared_ptr<std::vector<T> > store_;       // shallow copies share pixels, like cv::Mat
};

}  // namespace compat
}  // namespace phovo
#endif
