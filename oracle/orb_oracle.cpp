// oracle/orb_oracle.cpp -- CPU restatement of the reference ORB front end.
//
// TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED against the real reference
// binary: the reference cannot be built here and pins no result in any test; the OpenCV primitives
// are restated from the published OpenCV 4.x algorithms.
//
// Reference files followed (all under /root/reference/src/slam_system/orb_slam3/):
//   src/ORBextractor.cc:75-99    IC_Angle                    -> ic_angle()
//   src/ORBextractor.cc:102-143  computeOrbDescriptor        -> brief_descriptor()
//   src/ORBextractor.cc:282-339  ORBextractor::ORBextractor  -> OrbOracle::OrbOracle()
//   src/ORBextractor.cc:348-400  ExtractorNode::DivideNode   -> split_node()
//   src/ORBextractor.cc:402-417  compareNodes                -> node_less()
//   src/ORBextractor.cc:419-610  DistributeOctTree           -> distribute_octree()
//   src/ORBextractor.cc:612-715  ComputeKeyPointsOctTree     -> OrbOracle::detect()
//   src/ORBextractor.cc:876-955  operator()                  -> OrbOracle::extract()
//   src/ORBextractor.cc:957-976  ComputePyramid              -> OrbOracle::build_pyramid()
// Compile with -ffp-contract=off (no FMA contraction): every float op is a single IEEE rounding.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

const int kPatch = 31, kHalfPatch = 15, kEdge = 19;

const int kPattern[1024] = {
#include "../dvm_slam_amd/csrc/orb_pattern_31.inc"
};

// ---- cvRound / cvFloor / cvCeil (OpenCV fast_math.hpp): round-half-to-even under default FE mode
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}

// ---- cv::fastAtan2 scalar path (OpenCV mathfuncs_core: atan_f32), degrees, float, no FMA.
float fast_atan2(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  static const float p1 = 0.9997878412794807f * scale;
  static const float p3 = -0.3258083974640975f * scale;
  static const float p5 = 0.1555786518463281f * scale;
  static const float p7 = -0.04432655554792128f * scale;
  const float eps = (float)2.2204460492503131e-16;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- (float)cos(angle), (float)sin(angle) for angle = angle_deg * (float)(pi/180).
// Shared-spec implementation (DESIGN.md "sincos spec"): double Cody-Waite reduction by pi/2 and the
// fdlibm kernel polynomials, evaluated in Horner form with separate mul/add, rounded to float.
void sincos_deg(float angle_deg, float* c_out, float* s_out) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);  // ORBextractor.cc:101
  float angf = angle_deg * factorPI;                                           // :103
  double x = (double)angf;
  const double two_over_pi = 0.63661977236758134308;
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  double kd = std::nearbyint(x * two_over_pi);
  int k = (int)kd;
  double r = (x - kd * pio2_hi) - kd * pio2_lo;
  double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  double s = r + r * (z * ps);
  double c = (1.0 - 0.5 * z) + (z * z) * pc;
  double cs, sn;
  switch (k & 3) {
    case 0: cs = c; sn = s; break;
    case 1: cs = -s; sn = c; break;
    case 2: cs = -c; sn = -s; break;
    default: cs = s; sn = -c; break;
  }
  *c_out = (float)cs;
  *s_out = (float)sn;
}

// ---- umax table, ORBextractor.cc:320-338
void build_umax(int umax[16]) {
  int v, v0;
  int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

// ---- IC_Angle, ORBextractor.cc:75-99.  (cx,cy) = (cvRound(pt.x), cvRound(pt.y)) done by caller.
float ic_angle(const uint8_t* img, int stride, int cx, int cy, const int umax[16]) {
  int m01 = 0, m10 = 0;
  const uint8_t* center = img + (std::ptrdiff_t)cy * stride + cx;
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
  for (int v = 1; v <= kHalfPatch; ++v) {
    int vsum = 0, d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int vp = center[u + v * stride], vm = center[u - v * stride];
      vsum += (vp - vm);
      m10 += u * (vp + vm);
    }
    m01 += v * vsum;
  }
  return fast_atan2((float)m01, (float)m10);
}

// ---- computeOrbDescriptor, ORBextractor.cc:102-143
void brief_descriptor(const uint8_t* img, int stride, int cx, int cy, float angle_deg, uint8_t out[32]) {
  float a, b;
  sincos_deg(angle_deg, &a, &b);
  const uint8_t* center = img + (std::ptrdiff_t)cy * stride + cx;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int bit = 0; bit < 8; ++bit) {
      const int* p = kPattern + (i * 8 + bit) * 4;
      float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
      int t0 = center[cv_round_f(x0 * b + y0 * a) * stride + cv_round_f(x0 * a - y0 * b)];
      int t1 = center[cv_round_f(x1 * b + y1 * a) * stride + cv_round_f(x1 * a - y1 * b)];
      val |= (t0 < t1) << bit;
    }
    out[i] = (uint8_t)val;
  }
}

// ---- cv::resize(..., INTER_LINEAR) for CV_8UC1 (OpenCV resize.cpp: resizeGeneric_ with
// HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>).
inline short sat_short_from_float(float v) {
  int i = cv_round_f(v);
  return (short)std::min(std::max(i, -32768), 32767);
}
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short_from_float((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = sat_short_from_float(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = sat_short_from_float((1.f - fy) * 2048);
    ibeta[2 * dy + 1] = sat_short_from_float(fy * 2048);
  }
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = src + (std::ptrdiff_t)sy * sstride;
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      out[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
    }
    for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * 2048;
  };
  int have0 = -1, have1 = -1;  // source rows currently held in row0 / row1 (OpenCV keeps the same ring)
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    if (have1 == sy0) { row0.swap(row1); std::swap(have0, have1); }
    if (have0 != sy0) { hresize(sy0, row0); have0 = sy0; }
    if (have1 != sy1) { if (sy1 == sy0) row1 = row0; else hresize(sy1, row1); have1 = sy1; }
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst + (std::ptrdiff_t)dy * dstride;
    for (int x = 0; x < dw; x++) {
      int v = (((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2;
      D[x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

// ---- BORDER_REFLECT_101 index (OpenCV borderInterpolate)
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

// ---- cv::GaussianBlur(7x7, sigma 2) CV_8U fixed-point path (OpenCV smooth.dispatch.cpp:
// getGaussianKernelFixedPoint_ED -> 8.8 kernel; hline 8.8, vline 16.16, round-to-nearest).
void gaussian_kernel7_q8(int32_t k[7]) {
  const int n = 7;
  const double sigma = 2.0;
  double kf[7], sum = 0;
  const double scale2X = -0.5 / (sigma * sigma);
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    kf[i] = std::exp(scale2X * x * x);
    sum += kf[i];
  }
  for (int i = 0; i < n; i++) kf[i] = kf[i] * (1.0 / sum);
  double err = 0;
  long long acc = 0;
  for (int i = 0; i < n / 2; i++) {
    double adj = kf[i] * 256.0 + err;
    long long v = (long long)std::nearbyint(adj);
    err = adj - (double)v;
    k[i] = k[n - 1 - i] = (int32_t)v;
    acc += v;
  }
  k[n / 2] = (int32_t)(256 - 2 * acc);
}
void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  int32_t k[7];
  gaussian_kernel7_q8(k);
  // REFLECT_101-padded copy (3 px), then plain separable loops: same arithmetic, no per-tap index math
  const int pw = w + 6, ph = h + 6;
  std::vector<uint8_t> pad((size_t)pw * ph);
  for (int y = 0; y < ph; y++) {
    const uint8_t* S = src + (std::ptrdiff_t)reflect101(y - 3, h) * sstride;
    uint8_t* P = &pad[(size_t)y * pw];
    for (int x = 0; x < pw; x++) P[x] = S[reflect101(x - 3, w)];
  }
  std::vector<uint16_t> tmp((size_t)w * ph);
  for (int y = 0; y < ph; y++) {
    const uint8_t* P = &pad[(size_t)y * pw];
    uint16_t* T = &tmp[(size_t)y * w];
    for (int x = 0; x < w; x++) {
      uint32_t acc = (uint32_t)k[0] * (P[x] + P[x + 6]) + (uint32_t)k[1] * (P[x + 1] + P[x + 5]) +
                     (uint32_t)k[2] * (P[x + 2] + P[x + 4]) + (uint32_t)k[3] * P[x + 3];
      T[x] = (uint16_t)std::min(acc, 65535u);
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst + (std::ptrdiff_t)y * dstride;
    const uint16_t *r0 = &tmp[(size_t)y * w], *r1 = r0 + w, *r2 = r1 + w, *r3 = r2 + w, *r4 = r3 + w, *r5 = r4 + w, *r6 = r5 + w;
    for (int x = 0; x < w; x++) {
      uint32_t acc = (uint32_t)k[0] * ((uint32_t)r0[x] + r6[x]) + (uint32_t)k[1] * ((uint32_t)r1[x] + r5[x]) +
                     (uint32_t)k[2] * ((uint32_t)r2[x] + r4[x]) + (uint32_t)k[3] * r3[x];
      D[x] = (uint8_t)std::min((acc + 32768u) >> 16, 255u);
    }
  }
}

// ---- cv::FAST(img, kps, threshold, true) TYPE_9_16 (OpenCV fast.cpp FAST_t<16>, fast_score.cpp
// cornerScore<16>).  Brute-force form: A = max over the 16 nine-pixel arcs of min(v - p),
// B = same of min(p - v); corner iff max(A,B) > t; score = max(A,B) - 1; NMS keeps strict maxima of
// the 8-neighbourhood, scores outside the evaluated area (3-px frame) or of non-corners count 0.
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
inline int fast_strength(const uint8_t* p, int stride) {  // max(A,B); p points at the centre pixel
  int v = p[0], d[25];
  for (int k = 0; k < 16; k++) d[k] = v - p[kCircle[k][0] + kCircle[k][1] * stride];
  for (int k = 16; k < 25; k++) d[k] = d[k - 16];
  int A = -256, B = -256;
  for (int s = 0; s < 16; s++) {
    int mn = 255, mx = -255;
    for (int k = s; k < s + 9; k++) {
      mn = std::min(mn, d[k]);
      mx = std::max(mx, d[k]);
    }
    A = std::max(A, mn);
    B = std::max(B, -mx);
  }
  return std::max(A, B);
}
int fast9_16(const uint8_t* roi, int w, int h, int stride, int threshold, std::vector<int>& xs, std::vector<int>& ys,
             std::vector<int>& sc) {
  xs.clear(); ys.clear(); sc.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  if (w < 7 || h < 7) return 0;
  std::vector<uint8_t> score((size_t)w * h, 0);
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = roi + (std::ptrdiff_t)y * stride + x;
      // quick reject as in OpenCV's FAST_t (opposing ring pixels 0/8 and 4/12): a 9-arc contains at least
      // one pixel of every opposing pair, so both pairs need a pixel darker than v-t or both a brighter one
      const int v = p[0], lo = v - threshold, hi = v + threshold;
      const int n = p[3 * stride], s_ = p[-3 * stride], e = p[3], w_ = p[-3];
      const bool dark = (n < lo || s_ < lo) && (e < lo || w_ < lo);
      const bool bright = (n > hi || s_ > hi) && (e > hi || w_ > hi);
      if (!dark && !bright) continue;
      int m = fast_strength(p, stride);
      if (m > threshold) score[(size_t)y * w + x] = (uint8_t)(m - 1);
    }
  // note: a corner at threshold 0 with max(A,B)=1 has score 0 and can never pass the strict NMS,
  // exactly as in OpenCV where curr[j]=0 fails 'score > prev[j+1]' ... only if a neighbour is >=0,
  // i.e. always.  Thresholds used here are >= 7.
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int s = score[(size_t)y * w + x];
      if (s == 0) continue;
      bool keep = true;
      for (int dy = -1; dy <= 1 && keep; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (!dx && !dy) continue;
          if (s <= score[(size_t)(y + dy) * w + (x + dx)]) { keep = false; break; }
        }
      if (keep) { xs.push_back(x); ys.push_back(y); sc.push_back(s); }
    }
  return (int)xs.size();
}


// ---- DistributeOctTree, ORBextractor.cc:419-610 (+ DivideNode :348-400, compareNodes :402-417).
// A node is the rectangle [x0,x1) x [y0,y1) (the reference's UL/UR/BL/BR are its four corners).
// Candidates are referred to by index into the caller's arrays; `keys` keeps the parent's order.
struct QNode {
  int x0, y0, x1, y1;
  std::vector<int> keys;
  bool no_more = false;
  std::list<QNode>::iterator self;
};
struct Cand { float x, y, response; };

void split_node(const QNode& n, const std::vector<Cand>& c, QNode ch[4]) {
  const int halfX = (int)std::ceil(static_cast<float>(n.x1 - n.x0) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(n.y1 - n.y0) / 2);
  const int xm = n.x0 + halfX, ym = n.y0 + halfY;
  ch[0].x0 = n.x0; ch[0].y0 = n.y0; ch[0].x1 = xm;   ch[0].y1 = ym;
  ch[1].x0 = xm;   ch[1].y0 = n.y0; ch[1].x1 = n.x1; ch[1].y1 = ym;
  ch[2].x0 = n.x0; ch[2].y0 = ym;   ch[2].x1 = xm;   ch[2].y1 = n.y1;
  ch[3].x0 = xm;   ch[3].y0 = ym;   ch[3].x1 = n.x1; ch[3].y1 = n.y1;
  for (int k : n.keys) {
    const Cand& kp = c[k];
    int q = (kp.x < (float)xm) ? ((kp.y < (float)ym) ? 0 : 2) : ((kp.y < (float)ym) ? 1 : 3);
    ch[q].keys.push_back(k);
  }
  for (int q = 0; q < 4; q++) ch[q].no_more = (ch[q].keys.size() == 1);
}

typedef std::pair<int, QNode*> SizedNode;
bool node_less(const SizedNode& a, const SizedNode& b) {
  if (a.first != b.first) return a.first < b.first;
  return a.second->x0 < b.second->x0;
}

std::vector<int> distribute_octree(const std::vector<Cand>& c, int minX, int maxX, int minY, int maxY, int N) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<QNode> nodes;
  std::vector<QNode*> roots(nIni);
  for (int i = 0; i < nIni; i++) {
    QNode r;
    r.x0 = (int)(hX * static_cast<float>(i));
    r.x1 = (int)(hX * static_cast<float>(i + 1));
    r.y0 = 0;
    r.y1 = maxY - minY;
    nodes.push_back(r);
    roots[i] = &nodes.back();
  }
  for (int i = 0; i < (int)c.size(); i++) roots[(int)(c[i].x / hX)]->keys.push_back(i);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->no_more = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }

  std::vector<SizedNode> expandable;
  // pushes the non-empty children of *it to the front (order n1..n4) and records those with >1 key
  auto emit_children = [&](const QNode& parent, int* n_to_expand) {
    QNode ch[4];
    split_node(parent, c, ch);
    for (int q = 0; q < 4; q++) {
      if (ch[q].keys.empty()) continue;
      nodes.push_front(ch[q]);
      if (ch[q].keys.size() > 1) {
        if (n_to_expand) ++*n_to_expand;
        expandable.push_back(std::make_pair((int)ch[q].keys.size(), &nodes.front()));
        nodes.front().self = nodes.begin();
      }
    }
  };

  bool finish = false;
  while (!finish) {
    int prev_size = (int)nodes.size();
    int n_to_expand = 0;
    expandable.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->no_more) { ++it; continue; }
      emit_children(*it, &n_to_expand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) {
      finish = true;
    } else if ((int)nodes.size() + n_to_expand * 3 > N) {
      while (!finish) {
        prev_size = (int)nodes.size();
        std::vector<SizedNode> prev = expandable;
        expandable.clear();
        std::sort(prev.begin(), prev.end(), node_less);
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          emit_children(*prev[j].second, nullptr);
          nodes.erase(prev[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) finish = true;
      }
    }
  }

  std::vector<int> out;
  out.reserve(nodes.size());
  for (const QNode& n : nodes) {
    int best = n.keys[0];
    float best_r = c[best].response;
    for (size_t k = 1; k < n.keys.size(); k++)
      if (c[n.keys[k]].response > best_r) { best = n.keys[k]; best_r = c[best].response; }
    out.push_back(best);
  }
  return out;
}

}  // namespace

// ===================================================================================== OrbOracle
struct orc_orb {
  orc_orb_params p;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> nfeat;
  int umax[16];
  // per-extract state
  struct Level {
    int rows = 0, cols = 0;
    std::vector<uint8_t> buf;      // (rows+38) x (cols+38), image at offset (19,19)
    std::vector<uint8_t> blurred;  // rows x cols, empty when the level has no keypoint
    std::vector<int> cx, cy, cs;   // vToDistributeKeys
    std::vector<orc_keypoint> kps; // after octree + orientation, level coordinates
    int stride() const { return cols + 2 * kEdge; }
    const uint8_t* img() const { return buf.data() + (size_t)kEdge * stride() + kEdge; }
    uint8_t* img() { return buf.data() + (size_t)kEdge * stride() + kEdge; }
  };
  std::vector<Level> lv;

  explicit orc_orb(const orc_orb_params& pp) : p(pp) {
    const int L = p.nlevels;
    const double scaleFactor = (double)p.scale_factor;  // member is `double scaleFactor`, ORBextractor.h:84
    scale.resize(L); inv_scale.resize(L); sigma2.resize(L); inv_sigma2.resize(L); nfeat.resize(L);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
      scale[i] = (float)(scale[i - 1] * scaleFactor);
      sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < L; i++) {
      inv_scale[i] = 1.0f / scale[i];
      inv_sigma2[i] = 1.0f / sigma2[i];
    }
    float factor = (float)(1.0f / scaleFactor);
    float desired = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
      nfeat[l] = cv_round_f(desired);
      sum += nfeat[l];
      desired *= factor;
    }
    nfeat[L - 1] = std::max(p.nfeatures - sum, 0);
    build_umax(umax);
    lv.resize(L);
  }

  void fill_border(Level& l) {  // copyMakeBorder(..., 19, BORDER_REFLECT_101), ORBextractor.cc:969-973
    const int st = l.stride();
    for (int Y = 0; Y < l.rows + 2 * kEdge; Y++) {
      int sy = reflect101(Y - kEdge, l.rows);
      for (int X = 0; X < l.cols + 2 * kEdge; X++) {
        int sx = reflect101(X - kEdge, l.cols);
        if (sx == X - kEdge && sy == Y - kEdge) continue;
        l.buf[(size_t)Y * st + X] = l.img()[(size_t)sy * st + sx];
      }
    }
  }

  void build_pyramid(const uint8_t* img, int rows, int cols, int stride) {  // ORBextractor.cc:957-976
    for (int level = 0; level < p.nlevels; ++level) {
      float s = inv_scale[level];
      Level& l = lv[level];
      l.cols = cv_round_f((float)cols * s);
      l.rows = cv_round_f((float)rows * s);
      l.buf.assign((size_t)(l.rows + 2 * kEdge) * (l.cols + 2 * kEdge), 0);
      l.blurred.clear(); l.cx.clear(); l.cy.clear(); l.cs.clear(); l.kps.clear();
      if (level == 0) {
        for (int y = 0; y < rows; y++) std::memcpy(l.img() + (size_t)y * l.stride(), img + (size_t)y * stride, cols);
      } else {
        Level& pl = lv[level - 1];
        resize_linear_u8(pl.img(), pl.cols, pl.rows, pl.stride(), l.img(), l.cols, l.rows, l.stride());
      }
      fill_border(l);
    }
  }

  void detect() {  // ORBextractor.cc:612-715
    const float W = 35;
    std::vector<int> fx, fy, fs;
    for (int level = 0; level < p.nlevels; ++level) {
      Level& l = lv[level];
      const int minBorderX = kEdge - 3, minBorderY = minBorderX;
      const int maxBorderX = l.cols - kEdge + 3, maxBorderY = l.rows - kEdge + 3;
      const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
          const uint8_t* roi = l.img() + (std::ptrdiff_t)y0 * l.stride() + x0;
          int n = fast9_16(roi, x1 - x0, y1 - y0, l.stride(), p.ini_th_fast, fx, fy, fs);
          if (n == 0) n = fast9_16(roi, x1 - x0, y1 - y0, l.stride(), p.min_th_fast, fx, fy, fs);
          for (int k = 0; k < n; k++) {
            l.cx.push_back(fx[k] + j * wCell);
            l.cy.push_back(fy[k] + i * hCell);
            l.cs.push_back(fs[k]);
          }
        }
      }
      std::vector<Cand> cand(l.cx.size());
      for (size_t k = 0; k < cand.size(); k++) cand[k] = Cand{(float)l.cx[k], (float)l.cy[k], (float)l.cs[k]};
      std::vector<int> sel;
      if (!cand.empty()) sel = distribute_octree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, nfeat[level]);
      const int scaledPatchSize = (int)(kPatch * scale[level]);
      for (int idx : sel) {
        orc_keypoint kp;
        kp.x = cand[idx].x + minBorderX;
        kp.y = cand[idx].y + minBorderY;
        kp.size = (float)scaledPatchSize;
        kp.angle = -1.f;
        kp.response = cand[idx].response;
        kp.octave = level;
        kp.class_id = -1;
        l.kps.push_back(kp);
      }
    }
    for (int level = 0; level < p.nlevels; ++level) {  // computeOrientation, :341-346
      Level& l = lv[level];
      for (orc_keypoint& kp : l.kps)
        kp.angle = ic_angle(l.img(), l.stride(), cv_round_f(kp.x), cv_round_f(kp.y), umax);
    }
  }

  int extract(const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, orc_keypoint* out_kps,
              uint8_t* out_desc, int cap, int* mono_index) {  // ORBextractor.cc:876-955
    if (!img || rows <= 0 || cols <= 0) return -1;
    build_pyramid(img, rows, cols, stride);
    detect();
    int total = 0;
    for (auto& l : lv) total += (int)l.kps.size();
    if (total > cap) return -2;
    int mono = 0, stereo = total - 1;
    for (int level = 0; level < p.nlevels; ++level) {
      Level& l = lv[level];
      if (l.kps.empty()) continue;
      l.blurred.resize((size_t)l.rows * l.cols);
      gaussian_blur7(l.img(), l.cols, l.rows, l.stride(), l.blurred.data(), l.cols);
      const float s = scale[level];
      for (const orc_keypoint& k0 : l.kps) {
        uint8_t d[32];
        brief_descriptor(l.blurred.data(), l.cols, cv_round_f(k0.x), cv_round_f(k0.y), k0.angle, d);
        orc_keypoint kp = k0;
        if (level != 0) { kp.x = kp.x * s; kp.y = kp.y * s; }
        int dst = (kp.x >= (float)lap0 && kp.x <= (float)lap1) ? stereo-- : mono++;
        out_kps[dst] = kp;
        std::memcpy(out_desc + (size_t)dst * 32, d, 32);
      }
    }
    if (mono_index) *mono_index = mono;
    return total;
  }
};

// ========================================================================================= C API
extern "C" {

orc_orb* orc_orb_create(const orc_orb_params* p) { return new orc_orb(*p); }
void orc_orb_destroy(orc_orb* h) { delete h; }
void orc_orb_tables(const orc_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* nfeat, int32_t* umax16) {
  for (int i = 0; i < h->p.nlevels; i++) {
    if (scale) scale[i] = h->scale[i];
    if (inv_scale) inv_scale[i] = h->inv_scale[i];
    if (sigma2) sigma2[i] = h->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
    if (nfeat) nfeat[i] = h->nfeat[i];
  }
  if (umax16) for (int i = 0; i < 16; i++) umax16[i] = h->umax[i];
}
int orc_orb_extract(orc_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* mono_index) {
  return h->extract(img, rows, cols, stride, lap0, lap1, kps, desc, cap, mono_index);
}
int orc_orb_level_dims(const orc_orb* h, int level, int* rows, int* cols) {
  if (level < 0 || level >= h->p.nlevels) return -1;
  *rows = h->lv[level].rows; *cols = h->lv[level].cols;
  return 0;
}
int orc_orb_get_level(const orc_orb* h, int level, uint8_t* out) {
  const auto& l = h->lv[level];
  for (int y = 0; y < l.rows; y++) std::memcpy(out + (size_t)y * l.cols, l.img() + (size_t)y * l.stride(), l.cols);
  return 0;
}
int orc_orb_get_level_bordered(const orc_orb* h, int level, uint8_t* out) {
  const auto& l = h->lv[level];
  std::memcpy(out, l.buf.data(), l.buf.size());
  return 0;
}
int orc_orb_get_blurred(const orc_orb* h, int level, uint8_t* out) {
  const auto& l = h->lv[level];
  if (l.blurred.empty()) return -1;
  std::memcpy(out, l.blurred.data(), l.blurred.size());
  return 0;
}
int orc_orb_get_candidates(const orc_orb* h, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap) {
  const auto& l = h->lv[level];
  int n = (int)l.cx.size();
  for (int i = 0; i < n && i < cap; i++) { xs[i] = l.cx[i]; ys[i] = l.cy[i]; scores[i] = l.cs[i]; }
  return n;
}
int orc_orb_get_level_keypoints(const orc_orb* h, int level, orc_keypoint* kps, int cap) {
  const auto& l = h->lv[level];
  int n = (int)l.kps.size();
  for (int i = 0; i < n && i < cap; i++) kps[i] = l.kps[i];
  return n;
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_gaussian_blur7_s2_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gaussian_blur7(src, w, h, sstride, dst, dstride);
}
void orc_gaussian_kernel7_s2_q8(int32_t k[7]) { gaussian_kernel7_q8(k); }
int orc_fast9_16(const uint8_t* roi, int w, int h, int stride, int threshold, int32_t* xs, int32_t* ys,
                 int32_t* scores, int cap) {
  std::vector<int> x, y, s;
  int n = fast9_16(roi, w, h, stride, threshold, x, y, s);
  for (int i = 0; i < n && i < cap; i++) { xs[i] = x[i]; ys[i] = y[i]; scores[i] = s[i]; }
  return n;
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_cv_round(float v) { return cv_round_f(v); }
void orc_sincos_deg(float angle_deg, float* c, float* s) { sincos_deg(angle_deg, c, s); }
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy) {
  int um[16];
  build_umax(um);
  return ic_angle(img, stride, cx, cy, um);
}
void orc_brief_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t out[32]) {
  brief_descriptor(blurred, stride, cx, cy, angle_deg, out);
}
int orc_distribute_octree(const int32_t* xs, const int32_t* ys, const int32_t* scores, int n, int minX, int maxX,
                          int minY, int maxY, int N, int32_t* out_idx, int cap) {
  std::vector<Cand> c(n);
  for (int i = 0; i < n; i++) c[i] = Cand{(float)xs[i], (float)ys[i], (float)scores[i]};
  if (n == 0) return 0;
  std::vector<int> sel = distribute_octree(c, minX, maxX, minY, maxY, N);
  for (int i = 0; i < (int)sel.size() && i < cap; i++) out_idx[i] = sel[i];
  return (int)sel.size();
}

}  // extern "C"
