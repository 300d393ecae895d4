// test stub: the cv:: names the shims use (see tests/stubs/README.md)
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(x) ((void)(x))
namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
class Mat {
 public:
  Mat() {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t) {}
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  int type() const { return type_; }
  bool empty() const { return rows == 0; }
  Mat row(int) const { return *this; }
  Mat rowRange(int, int) const { return *this; }
  template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
  template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }
  template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[r * cols + c]; }
  template <class T> T* ptr(int = 0) { return reinterpret_cast<T*>(data); }
  template <class T> const T* ptr(int = 0) const { return reinterpret_cast<const T*>(data); }
  void create(int r, int c, int t) { rows = r; cols = c; type_ = t; }
  void copyTo(class _OutputArray&) const {}
  void release() {}
  bool isContinuous() const { return true; }
 private:
  int type_ = 0;
};
class _InputArray { public: _InputArray() {} _InputArray(const Mat& m) : m_(m) {} bool empty() const { return m_.empty(); } Mat getMat() const { return m_; } private: Mat m_; };
class _OutputArray { public: _OutputArray() {} _OutputArray(Mat&) {} void release() {} };
typedef const _InputArray& InputArray;
typedef _OutputArray& OutputArray;
}  // namespace cv
