// test mock: the cv:: names the shims use (see tests/stubs/README.md).  cv::Mat is a reference-counted byte matrix (8-bit
// rows x cols) -- enough for images, N x 32 descriptor tables and their row() views.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) ((void)(x))
namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
class _OutputArray;
class Mat {
 public:
  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(int r, int c, int t, void* ext, size_t step_ = 0) : rows(r), cols(c), data(static_cast<uint8_t*>(ext)), step(step_ ? step_ : (size_t)c * esz(t)), type_(t) {}
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  int type() const { return type_; }
  bool empty() const { return rows == 0 || cols == 0 || !data; }
  Mat row(int i) const { Mat m = *this; m.rows = 1; m.data = data + (size_t)i * step; return m; }
  Mat rowRange(int a, int b) const { Mat m = *this; m.rows = b - a; m.data = data + (size_t)a * step; return m; }
  Mat clone() const { Mat m(rows, cols, type_); for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz(type_)); return m; }
  template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
  template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }
  template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data + (size_t)r * step)[c]; }
  template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data + (size_t)r * step)[c]; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  void create(int r, int c, int t) {
    rows = r; cols = c; type_ = t; step = (size_t)c * esz(t);
    own_.reset(new std::vector<uint8_t>((size_t)r * step + 1));
    data = own_->data();
  }
  inline void copyTo(_OutputArray& o) const;
  void release() { own_.reset(); data = nullptr; rows = cols = 0; }
  bool isContinuous() const { return step == (size_t)cols * esz(type_); }
 private:
  static size_t esz(int t) { return t == CV_32F ? 4 : 1; }
  int type_ = 0;
  std::shared_ptr<std::vector<uint8_t>> own_;
};
class _InputArray { public: _InputArray() {} _InputArray(const Mat& m) : m_(m) {} bool empty() const { return m_.empty(); } Mat getMat() const { return m_; } private: Mat m_; };
// an output array bound to a cv::Mat: create() allocates it, getMat() hands the header out (data pointer shared)
class _OutputArray {
 public:
  _OutputArray() {}
  _OutputArray(Mat& m) : m_(&m) {}
  void release() { if (m_) m_->release(); }
  void create(int r, int c, int t) { if (m_) m_->create(r, c, t); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 private:
  Mat* m_ = nullptr;
};
inline void Mat::copyTo(_OutputArray& o) const {
  o.create(rows, cols, type_);
  Mat d = o.getMat();
  for (int r = 0; r < rows; r++) std::memcpy(d.data + (size_t)r * d.step, data + (size_t)r * step, (size_t)cols * esz(type_));
}
typedef const _InputArray& InputArray;
typedef _OutputArray& OutputArray;
}  // namespace cv
