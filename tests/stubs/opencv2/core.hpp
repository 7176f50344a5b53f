// Syntax-only stand-in for <opencv2/core.hpp> (see tests/stubs/README.md).  Same member names and meanings as the
// real cv::Mat for the subset the adapters touch; reference-counted storage like the real one.
#ifndef SVO_TEST_STUB_OPENCV_CORE_HPP_
#define SVO_TEST_STUB_OPENCV_CORE_HPP_
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;
template <typename T> struct Point3_ {
  T x, y, z;
  Point3_() : x(0), y(0), z(0) {}
  Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<float> Point3f;

class Mat {
 public:
  struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
    size_t operator[](int) const { return v; }
  };
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  MatStep step;

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t ext_step = 0) : rows(r), cols(c), data(static_cast<uchar*>(ext)), type_(type) {
    step.v = ext_step ? ext_step : (size_t)c * elem(type);
  }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  static Mat eye(int r, int c, int type) {
    Mat m(r, c, type);
    for (int i = 0; i < r && i < c; ++i) {
      if ((type & 7) == CV_32F) m.at<float>(i, i) = 1.f;
      else m.at<uchar>(i, i) = 1;
    }
    return m;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data && store_) return;
    rows = r; cols = c; type_ = type;
    step.v = (size_t)c * elem(type);
    store_ = std::make_shared<std::vector<uchar>>((size_t)r * step.v, (uchar)0);
    data = store_->data();
  }
  int type() const { return type_; }
  int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step.v == (size_t)cols * elem(type_); }
  Mat clone() const {
    Mat m;
    if (empty()) return m;
    m.create(rows, cols, type_);
    for (int r = 0; r < rows; ++r) memcpy(m.data + (size_t)r * m.step.v, data + (size_t)r * step.v, (size_t)cols * elem(type_));
    return m;
  }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step.v); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step.v); }

 private:
  static size_t elem(int type) { return (size_t)(((type & 7) == CV_32F) ? 4 : 1) * (size_t)((type >> CV_CN_SHIFT) + 1); }
  int type_ = 0;
  std::shared_ptr<std::vector<uchar>> store_;
};

}  // namespace cv
#endif
