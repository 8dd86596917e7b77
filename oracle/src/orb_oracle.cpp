// oracle/src/orb_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference's ORB extraction path.
// In-tree logic is restated from /root/reference/src/ssvio/orbextractor.cpp (lines cited per function).
// Where the reference calls OpenCV 3.2 (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound) --
// which is neither vendored nor installed -- the published OpenCV 3.x algorithm is restated from its
// documentation/behaviour and the result is PARITY UNPINNED (no OpenCV build, test or golden vector exists
// to check against); the integer/fixed-point choices made are listed in DESIGN.md.
//
// Deliberate, documented deviations (both needed for a deterministic spec):
//  * octree tie-break: the reference sorts (size, ExtractorNode*) pairs (orbextractor.cpp:486), i.e. equal
//    sizes are ordered by heap address; here equal sizes are ordered by node creation sequence.
//  * sin/cos of the keypoint angle (orbextractor.cpp:49-50 calls libm cosf/sinf): computed by orc_sincos_deg,
//    a fixed sequence of IEEE double operations rounded to float, so that the GPU can reproduce it bit for
//    bit; tests/test_oracle_orb.py measures its agreement with libm.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>

#include "oracle.h"

namespace {

const int PATCH_SIZE = 31;
const int HALF_PATCH_SIZE = 15;
const int EDGE_THRESHOLD = 19;

const int8_t kPattern[256 * 4] = {
#include "brief_pattern.inc"
};

// cvRound: round half to even (x86 cvtsd2si), via the default rounding mode
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

struct Img {
  const uint8_t* p; int stride, rows, cols;
  uint8_t at(int y, int x) const { return p[(size_t)y * stride + x]; }
};

// ring offsets of the 16-pixel Bresenham circle, orbextractor.cpp:95-123 (same table as OpenCV's)
const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// FAST-9/16 segment test at one pixel: >= 9 contiguous ring pixels darker than v-t or brighter than v+t
// (orbextractor.cpp:194-268 isFastCorner == the test inside cv::FAST)
bool segment_test(const uint8_t* c, int stride, int t)
{
  const int v = c[0];
  int ring[25];
  for (int k = 0; k < 25; ++k) ring[k] = c[kRing[k % 16][0] + kRing[k % 16][1] * stride];
  int cd = 0, cb = 0;
  for (int k = 0; k < 25; ++k) {
    if (ring[k] < v - t) { if (++cd > 8) return true; } else cd = 0;
    if (ring[k] > v + t) { if (++cb > 8) return true; } else cb = 0;
  }
  return false;
}

// OpenCV cornerScore<16>: the largest threshold for which the pixel is still a FAST-9 corner, minus 1
int corner_score(const uint8_t* c, int stride, int t)
{
  int d[25];
  const int v = c[0];
  for (int k = 0; k < 25; ++k) d[k] = v - c[kRing[k % 16][0] + kRing[k % 16][1] * stride];
  int a0 = t;
  for (int k = 0; k < 16; ++k) {   // every arc of 9 contiguous ring pixels starting at k
    int a = d[k];
    for (int q = 1; q < 9; ++q) a = std::min(a, d[k + q]);
    a0 = std::max(a0, a);
  }
  int b0 = -a0;
  for (int k = 0; k < 16; ++k) {
    int b = d[k];
    for (int q = 1; q < 9; ++q) b = std::max(b, d[k + q]);
    b0 = std::min(b0, b);
  }
  return -b0 - 1;
}

// cv::FAST(roi, kps, threshold, nonmaxSuppression=true) -- OpenCV 3.x FAST_t<16>: corners on rows
// 3..rows-4 and cols 3..cols-4 of the ROI, score stored as uchar, 3x3 NMS with strict '>' against score
// buffers that are zero outside the ROI interior; output row-major.
struct Cand { int x, y, score; };
void fast_roi(const Img& im, int x0, int y0, int w, int h, int threshold, std::vector<Cand>& out)
{
  out.clear();
  if (w < 7 || h < 7) return;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> sc((size_t)w * h, 0);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      const uint8_t* c = im.p + (size_t)(y0 + y) * im.stride + (x0 + x);
      if (segment_test(c, im.stride, threshold)) sc[(size_t)y * w + x] = (uint8_t)corner_score(c, im.stride, threshold);
    }
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      const uint8_t* c = im.p + (size_t)(y0 + y) * im.stride + (x0 + x);
      if (!segment_test(c, im.stride, threshold)) continue;
      const int s = sc[(size_t)y * w + x];
      const uint8_t* r = &sc[(size_t)y * w + x];
      if (s > r[1] && s > r[-1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] && s > r[w - 1] && s > r[w] && s > r[w + 1])
        out.push_back({x, y, s});
    }
}

// grid FAST of ORBextractor::Detect / ComputeKeyPointsOctTree (orbextractor.cpp:765-829, 575-647)
void grid_fast(const Img& im, const Img* mask, int ini_th, int min_th, std::vector<orc_keypoint>& cands)
{
  const float W = 30;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = im.cols - EDGE_THRESHOLD + 3, maxBorderY = im.rows - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols < 1 || nRows < 1) return;
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
  std::vector<Cand> cell;
  for (int i = 0; i < nRows; ++i) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; ++j) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      const int rx = (int)iniX, ry = (int)iniY, rw = (int)maxX - (int)iniX, rh = (int)maxY - (int)iniY;
      fast_roi(im, rx, ry, rw, rh, ini_th, cell);
      if (cell.empty()) fast_roi(im, rx, ry, rw, rh, min_th, cell);
      for (const Cand& c : cell) {
        orc_keypoint kp;
        kp.x = (float)c.x + (float)(j * wCell);
        kp.y = (float)c.y + (float)(i * hCell);
        kp.size = 7.f; kp.angle = -1.f; kp.response = (float)c.score; kp.octave = 0; kp.class_id = -1;
        // the mask is indexed with the un-bordered cell coordinates (orbextractor.cpp:818-823): ref quirk
        if (mask && mask->at(cv_round(kp.y), cv_round(kp.x)) == 0) continue;
        cands.push_back(kp);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// DistributeOctTree (orbextractor.cpp:340-568) + ExtractorNode::DivideNode (:282-338)
// ---------------------------------------------------------------------------------------------
struct Node {
  int ulx, uly, brx, bry;        // UL and BR corners (UR = (brx,uly), BL = (ulx,bry))
  std::vector<int> keys;         // candidate indices, in insertion order
  bool no_more = false;
  long seq = 0;                  // creation sequence (deterministic tie-break)
  std::list<Node>::iterator self;
};

void divide(const Node& n, const std::vector<orc_keypoint>& c, Node out[4])
{
  const int halfX = (int)std::ceil((float)(n.brx - n.ulx) / 2);
  const int halfY = (int)std::ceil((float)(n.bry - n.uly) / 2);
  const int mx = n.ulx + halfX, my = n.uly + halfY;
  out[0] = Node{n.ulx, n.uly, mx, my, {}, false, 0, {}};
  out[1] = Node{mx, n.uly, n.brx, my, {}, false, 0, {}};
  out[2] = Node{n.ulx, my, mx, n.bry, {}, false, 0, {}};
  out[3] = Node{mx, my, n.brx, n.bry, {}, false, 0, {}};
  for (int k : n.keys) {
    const orc_keypoint& kp = c[k];
    const int q = (kp.x < mx) ? ((kp.y < my) ? 0 : 2) : ((kp.y < my) ? 1 : 3);
    out[q].keys.push_back(k);
  }
  for (int q = 0; q < 4; ++q)
    if (out[q].keys.size() == 1) out[q].no_more = true;
}

void octree(const std::vector<orc_keypoint>& c, int minX, int maxX, int minY, int maxY, int N,
            std::vector<orc_keypoint>& result)
{
  result.clear();
  if (c.empty()) return;
  const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
  if (nIni < 1) return;
  const float hX = (float)(maxX - minX) / nIni;
  std::list<Node> nodes;
  std::vector<Node*> ini(nIni);
  long seq = 0;
  for (int i = 0; i < nIni; ++i) {
    Node n{(int)(hX * (float)i), 0, (int)(hX * (float)(i + 1)), maxY - minY, {}, false, seq++, {}};
    nodes.push_back(n);
    ini[i] = &nodes.back();
  }
  for (size_t i = 0; i < c.size(); ++i) {
    int idx = (int)(c[i].x / hX);
    if (idx >= nIni) idx = nIni - 1;   // guard (the reference would index out of bounds)
    ini[idx]->keys.push_back((int)i);
  }
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->no_more = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  typedef std::pair<int, Node*> SizeNode;
  std::vector<SizeNode> expandable;
  auto push_children = [&](Node ch[4]) {
    for (int q = 0; q < 4; ++q) {
      if (ch[q].keys.empty()) continue;
      ch[q].seq = seq++;
      nodes.push_front(ch[q]);
      nodes.front().self = nodes.begin();
      if (nodes.front().keys.size() > 1) expandable.push_back({(int)nodes.front().keys.size(), &nodes.front()});
    }
  };
  bool finish = false;
  while (!finish) {
    int prevSize = (int)nodes.size();
    int nToExpand = 0;
    expandable.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->no_more) { ++it; continue; }
      Node ch[4];
      divide(*it, c, ch);
      const size_t before = expandable.size();
      push_children(ch);
      nToExpand += (int)(expandable.size() - before);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
      finish = true;
    } else if ((int)nodes.size() + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<SizeNode> prev = expandable;
        expandable.clear();
        // ascending (size, creation seq); walked from the back = largest first, newest first among equals
        std::sort(prev.begin(), prev.end(), [](const SizeNode& a, const SizeNode& b) {
          return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq;
        });
        for (int j = (int)prev.size() - 1; j >= 0; --j) {
          Node ch[4];
          divide(*prev[j].second, c, ch);
          push_children(ch);
          nodes.erase(prev[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  // best response per node, first one wins ties (orbextractor.cpp:549-565)
  for (const Node& n : nodes) {
    int best = n.keys[0];
    for (size_t k = 1; k < n.keys.size(); ++k)
      if (c[n.keys[k]].response > c[best].response) best = n.keys[k];
    result.push_back(c[best]);
  }
}

// ---------------------------------------------------------------------------------------------
// constructor tables (orbextractor.cpp:127-192)
// ---------------------------------------------------------------------------------------------
struct Tables {
  std::vector<float> scale, inv_scale;
  std::vector<int> feats;
  int umax[HALF_PATCH_SIZE + 1];
};

void make_tables(int nfeatures, float scaleFactor, int nlevels, Tables& t)
{
  t.scale.assign(nlevels, 1.0f); t.inv_scale.assign(nlevels, 1.0f);
  for (int i = 1; i < nlevels; ++i) t.scale[i] = t.scale[i - 1] * scaleFactor;
  for (int i = 0; i < nlevels; ++i) t.inv_scale[i] = 1.0f / t.scale[i];
  t.feats.assign(nlevels, 0);
  const float factor = 1.0f / scaleFactor;
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; ++level) {
    t.feats[level] = cv_round(nDesired);
    sum += t.feats[level];
    nDesired *= factor;
  }
  t.feats[nlevels - 1] = std::max(nfeatures - sum, 0);
  int v, v0;
  const int vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
  const int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
    t.umax[v] = v0;
    ++v0;
  }
}

// ---------------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1 -- OpenCV 3.x fixed-point bilinear:
// 11-bit coefficients (INTER_RESIZE_COEF_SCALE = 2048), horizontal pass into int, vertical pass
// ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
// ---------------------------------------------------------------------------------------------
void resize_linear(const uint8_t* src, int sstride, int srows, int scols, uint8_t* dst, int dstride,
                   int drows, int dcols)
{
  const double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dcols), yofs(drows);
  std::vector<short> ialpha(2 * dcols), ibeta(2 * drows);
  int xmax = dcols;
  for (int dx = 0; dx < dcols; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= scols) {
      xmax = std::min(xmax, dx);
      if (sx >= scols - 1) { fx = 0; sx = scols - 1; }
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)std::min(std::max(cv_round((1.f - fx) * 2048), -32768), 32767);
    ialpha[2 * dx + 1] = (short)std::min(std::max(cv_round(fx * 2048), -32768), 32767);
  }
  for (int dy = 0; dy < drows; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = (short)std::min(std::max(cv_round((1.f - fy) * 2048), -32768), 32767);
    ibeta[2 * dy + 1] = (short)std::min(std::max(cv_round(fy * 2048), -32768), 32767);
  }
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  std::vector<int> row0(dcols), row1(dcols);
  auto hresize = [&](int sy, std::vector<int>& D) {
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dcols; ++dx) {
      const int sx = xofs[dx];
      if (dx < xmax) D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
      else D[dx] = S[sx] * 2048;
    }
  };
  for (int dy = 0; dy < drows; ++dy) {
    const int sy0 = clip(yofs[dy], 0, srows), sy1 = clip(yofs[dy] + 1, 0, srows);
    hresize(sy0, row0);
    hresize(sy1, row1);
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dcols; ++x)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) for 8UC1 -- OpenCV 3.x: float kernel
// from getGaussianKernel, converted to 8-bit fixed point (x256), separable integer row pass, column pass
// with rounding shift by 16.
// ---------------------------------------------------------------------------------------------
void gauss_kernel_q8(int k[7])
{
  const int n = 7;
  const double sigma = 2.0;
  const double scale2X = -0.5 / (sigma * sigma);
  float cf[7];
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cv_round((double)cf[i] * 256.0);
  }
}

inline int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}

void gauss7(const uint8_t* src, int sstride, int rows, int cols, uint8_t* dst, int dstride)
{
  int k[7];
  gauss_kernel_q8(k);
  std::vector<int> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int s = 0;
      for (int q = -3; q <= 3; ++q) s += k[q + 3] * src[(size_t)y * sstride + reflect101(x + q, cols)];
      tmp[(size_t)y * cols + x] = s;
    }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int s = 0;
      for (int q = -3; q <= 3; ++q) s += k[q + 3] * tmp[(size_t)reflect101(y + q, rows) * cols + x];
      const int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
}

// cv::fastAtan2 (degrees), OpenCV 3.x scalar polynomial
float fast_atan2(float y, float x)
{
  const float s = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
  const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// IC_Angle, orbextractor.cpp:15-43
float ic_angle(const Img& im, float px, float py, const int* umax)
{
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = im.p + (size_t)cv_round(py) * im.stride + cv_round(px);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  const int step = im.stride;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// deterministic sin/cos of an angle given in degrees as float (see header comment).  radians are formed
// exactly as orbextractor.cpp:45-49 does: float(angle) * float(CV_PI/180.f).
void sincos_deg(float angle_deg, float* c, float* s)
{
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float ang = angle_deg * factorPI;
  const double r = (double)ang;
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
  const double kq = std::floor(r * two_over_pi + 0.5);
  const int q = ((int)kq) & 3;
  const double y = (r - kq * pio2_hi) - kq * pio2_lo;
  const double z = y * y;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  const double sn = y + (y * z) * ps;
  const double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
  double so, co;
  switch (q) {
    case 0: so = sn; co = cs; break;
    case 1: so = cs; co = -sn; break;
    case 2: so = -sn; co = -cs; break;
    default: so = -cs; co = sn; break;
  }
  *c = (float)co;
  *s = (float)so;
}

// computeOrbDescriptor, orbextractor.cpp:46-91
void brief(const Img& im, float px, float py, float angle_deg, uint8_t* desc)
{
  float a, b;
  sincos_deg(angle_deg, &a, &b);
  const uint8_t* center = im.p + (size_t)cv_round(py) * im.stride + cv_round(px);
  const int step = im.stride;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int bit = 0; bit < 8; ++bit) {
      const int8_t* t = &kPattern[(i * 8 + bit) * 4];
      const int t0 = center[cv_round((float)t[0] * b + (float)t[1] * a) * step + cv_round((float)t[0] * a - (float)t[1] * b)];
      const int t1 = center[cv_round((float)t[2] * b + (float)t[3] * a) * step + cv_round((float)t[2] * a - (float)t[3] * b)];
      val |= (t0 < t1) << bit;
    }
    desc[i] = (uint8_t)val;
  }
}

struct Pyramid {
  std::vector<std::vector<uint8_t>> img, mask;
  std::vector<int> rows, cols;
};

// ComputePyramid, orbextractor.cpp:993-1027
void build_pyramid(const Img& im, const Img* mask, const Tables& t, int nlevels, Pyramid& py)
{
  py.img.resize(nlevels); py.mask.resize(nlevels); py.rows.resize(nlevels); py.cols.resize(nlevels);
  for (int l = 0; l < nlevels; ++l) {
    const float sc = t.inv_scale[l];
    py.cols[l] = cv_round((float)im.cols * sc);
    py.rows[l] = cv_round((float)im.rows * sc);
    py.img[l].resize((size_t)py.rows[l] * py.cols[l]);
    py.mask[l].resize((size_t)py.rows[l] * py.cols[l]);
    if (l == 0) {
      for (int y = 0; y < im.rows; ++y) {
        std::memcpy(&py.img[0][(size_t)y * im.cols], im.p + (size_t)y * im.stride, im.cols);
        if (mask) std::memcpy(&py.mask[0][(size_t)y * im.cols], mask->p + (size_t)y * mask->stride, im.cols);
        else std::memset(&py.mask[0][(size_t)y * im.cols], 255, im.cols);
      }
    } else {
      resize_linear(py.img[l - 1].data(), py.cols[l - 1], py.rows[l - 1], py.cols[l - 1], py.img[l].data(), py.cols[l],
                    py.rows[l], py.cols[l]);
      resize_linear(py.mask[l - 1].data(), py.cols[l - 1], py.rows[l - 1], py.cols[l - 1], py.mask[l].data(), py.cols[l],
                    py.rows[l], py.cols[l]);
    }
  }
}

}  // namespace

extern "C" {

const int8_t* orc_brief_pattern(void) { return kPattern; }

int orc_fast_roi(const uint8_t* img, int stride, int rows, int cols, int threshold, int cap, int32_t* xs,
                 int32_t* ys, int32_t* scores)
{
  Img im{img, stride, rows, cols};
  std::vector<Cand> out;
  fast_roi(im, 0, 0, cols, rows, threshold, out);
  const int n = std::min<int>((int)out.size(), cap);
  for (int i = 0; i < n; ++i) { xs[i] = out[i].x; ys[i] = out[i].y; scores[i] = out[i].score; }
  return (int)out.size();
}

int orc_is_fast_corner(const uint8_t* img, int stride, int x, int y, int threshold)
{
  threshold = std::min(std::max(threshold, 0), 255);
  return segment_test(img + (size_t)y * stride + x, stride, threshold) ? 1 : 0;
}

int orc_orb_grid_fast(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                      int ini_th, int min_th, int cap, orc_keypoint* out)
{
  Img im{img, stride, rows, cols}, mk{mask, mask_stride, rows, cols};
  std::vector<orc_keypoint> c;
  grid_fast(im, mask ? &mk : nullptr, ini_th, min_th, c);
  const int n = std::min<int>((int)c.size(), cap);
  for (int i = 0; i < n; ++i) out[i] = c[i];
  return (int)c.size();
}

int orc_octree(const orc_keypoint* cand, int n, int minX, int maxX, int minY, int maxY, int N, int cap,
               orc_keypoint* out)
{
  std::vector<orc_keypoint> c(cand, cand + n), r;
  octree(c, minX, maxX, minY, maxY, N, r);
  const int m = std::min<int>((int)r.size(), cap);
  for (int i = 0; i < m; ++i) out[i] = r[i];
  return (int)r.size();
}

// ORBextractor::Detect, orbextractor.cpp:755-842
int orc_orb_detect(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                   const orc_orb_params* prm, int cap, orc_keypoint* kps_out)
{
  if (!img || rows <= 0 || cols <= 0) return 0;   // silently returns on empty input (:758-759)
  Img im{img, stride, rows, cols}, mk{mask, mask_stride, rows, cols};
  std::vector<orc_keypoint> c, r;
  grid_fast(im, mask ? &mk : nullptr, prm->ini_th_fast, prm->min_th_fast, c);
  const int minB = EDGE_THRESHOLD - 3;
  octree(c, minB, cols - EDGE_THRESHOLD + 3, minB, rows - EDGE_THRESHOLD + 3, prm->nfeatures, r);
  for (auto& k : r) { k.x += minB; k.y += minB; }
  const int m = std::min<int>((int)r.size(), cap);
  for (int i = 0; i < m; ++i) kps_out[i] = r[i];
  return (int)r.size();
}

void orc_level_sizes(int rows, int cols, float scale_factor, int nlevels, int32_t* rows_out, int32_t* cols_out)
{
  Tables t;
  make_tables(100, scale_factor, nlevels, t);
  for (int l = 0; l < nlevels; ++l) {
    cols_out[l] = cv_round((float)cols * t.inv_scale[l]);
    rows_out[l] = cv_round((float)rows * t.inv_scale[l]);
  }
}

void orc_features_per_level(int nfeatures, float scale_factor, int nlevels, int32_t* out)
{
  Tables t;
  make_tables(nfeatures, scale_factor, nlevels, t);
  for (int l = 0; l < nlevels; ++l) out[l] = t.feats[l];
}

void orc_umax(int32_t* out16)
{
  Tables t;
  make_tables(100, 1.2f, 8, t);
  for (int i = 0; i <= HALF_PATCH_SIZE; ++i) out16[i] = t.umax[i];
}

void orc_resize_linear(const uint8_t* src, int sstride, int srows, int scols, uint8_t* dst, int dstride, int drows,
                       int dcols)
{
  resize_linear(src, sstride, srows, scols, dst, dstride, drows, dcols);
}

void orc_gauss7(const uint8_t* src, int sstride, int rows, int cols, uint8_t* dst, int dstride)
{
  gauss7(src, sstride, rows, cols, dst, dstride);
}

float orc_ic_angle(const uint8_t* img, int stride, float x, float y)
{
  Tables t;
  make_tables(100, 1.2f, 8, t);
  Img im{img, stride, 0, 0};
  return ic_angle(im, x, y, t.umax);
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orc_sincos_deg(float angle_deg, float* c, float* s) { sincos_deg(angle_deg, c, s); }

// computeOrbDescriptor with the reference's OWN sine / cosine (orbextractor.cpp:49-50: `(float)cos(angle)`, `(float)sin(angle)`
// of a float angle -- cosf / sinf through <cmath>'s overloads, mode 0, or the double functions of <math.h> rounded to float,
// mode 1): what tests/test_oracle_orb.py compares the deterministic sincos_deg with, bit by bit of the descriptor
void orc_brief_libm(const uint8_t* blurred, int stride, float px, float py, float angle_deg, int mode, uint8_t* desc, float* ab_out)
{
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float angle = angle_deg * factorPI;
  const float a = mode == 0 ? cosf(angle) : (float)cos((double)angle), b = mode == 0 ? sinf(angle) : (float)sin((double)angle);
  if (ab_out) { ab_out[0] = a; ab_out[1] = b; }
  const uint8_t* center = blurred + (size_t)cv_round(py) * stride + cv_round(px);
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int bit = 0; bit < 8; ++bit) {
      const int8_t* t = &kPattern[(i * 8 + bit) * 4];
      const int t0 = center[cv_round((float)t[0] * b + (float)t[1] * a) * stride + cv_round((float)t[0] * a - (float)t[1] * b)];
      const int t1 = center[cv_round((float)t[2] * b + (float)t[3] * a) * stride + cv_round((float)t[2] * a - (float)t[3] * b)];
      val |= (t0 < t1) << bit;
    }
    desc[i] = (uint8_t)val;
  }
}

// n keypoints at once: bits of the descriptors that differ between sincos_deg and libm, and how many (cos, sin) pairs differ
void orc_brief_libm_census(const uint8_t* blurred, int stride, int n, const float* xya, int mode, int64_t* out3)
{
  int64_t bits = 0, descs = 0, pairs = 0;
  for (int i = 0; i < n; ++i) {
    uint8_t d0[32], d1[32];
    float ab[2], c, s;
    Img im{blurred, stride, 0, 0};
    brief(im, xya[3 * i], xya[3 * i + 1], xya[3 * i + 2], d0);
    orc_brief_libm(blurred, stride, xya[3 * i], xya[3 * i + 1], xya[3 * i + 2], mode, d1, ab);
    sincos_deg(xya[3 * i + 2], &c, &s);
    pairs += (c != ab[0] || s != ab[1]) ? 1 : 0;
    int diff = 0;
    for (int k = 0; k < 32; ++k) diff += __builtin_popcount((unsigned)(d0[k] ^ d1[k]));
    bits += diff; descs += diff ? 1 : 0;
  }
  out3[0] = bits; out3[1] = descs; out3[2] = pairs;
}

void orc_brief(const uint8_t* blurred, int stride, float x, float y, float angle_deg, uint8_t* desc32)
{
  Img im{blurred, stride, 0, 0};
  brief(im, x, y, angle_deg, desc32);
}

// ORBextractor::DetectAndCompute, orbextractor.cpp:687-753 (+ ComputeKeyPointsOctTree :572-676)
int orc_orb_extract(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                    const orc_orb_params* prm, int cap, orc_keypoint* kps_out, uint8_t* desc_out)
{
  if (!img || rows <= 0 || cols <= 0) return 0;
  Tables t;
  make_tables(prm->nfeatures, prm->scale_factor, prm->nlevels, t);
  Img im{img, stride, rows, cols}, mk{mask, mask_stride, rows, cols};
  Pyramid py;
  build_pyramid(im, mask ? &mk : nullptr, t, prm->nlevels, py);
  int total = 0;
  const int minB = EDGE_THRESHOLD - 3;
  for (int l = 0; l < prm->nlevels; ++l) {
    Img li{py.img[l].data(), py.cols[l], py.rows[l], py.cols[l]};
    Img lm{py.mask[l].data(), py.cols[l], py.rows[l], py.cols[l]};
    std::vector<orc_keypoint> c, r;
    grid_fast(li, &lm, prm->ini_th_fast, prm->min_th_fast, c);
    octree(c, minB, li.cols - EDGE_THRESHOLD + 3, minB, li.rows - EDGE_THRESHOLD + 3, t.feats[l], r);
    const int scaledPatchSize = (int)(PATCH_SIZE * t.scale[l]);
    for (auto& k : r) {
      k.x += minB; k.y += minB; k.octave = l; k.size = (float)scaledPatchSize;
      k.angle = ic_angle(li, k.x, k.y, t.umax);
    }
    if (r.empty()) continue;
    std::vector<uint8_t> blur((size_t)li.rows * li.cols);
    gauss7(li.p, li.stride, li.rows, li.cols, blur.data(), li.cols);
    Img bi{blur.data(), li.cols, li.rows, li.cols};
    for (auto& k : r) {
      if (total < cap) {
        brief(bi, k.x, k.y, k.angle, desc_out + 32 * (size_t)total);
        orc_keypoint o = k;
        if (l != 0) { o.x *= t.scale[l]; o.y *= t.scale[l]; }
        kps_out[total] = o;
      }
      ++total;
    }
  }
  return total;
}

// ScreenAndComputeKPsParams (orbextractor.cpp:844-894) followed by CalcDescriptors (:943-991)
int orc_orb_describe_at(const uint8_t* img, int stride, int rows, int cols, const orc_orb_params* prm,
                        const orc_keypoint* kps_in, int n_in, orc_keypoint* kps_out, uint8_t* desc_out)
{
  if (!img || rows <= 0 || cols <= 0 || n_in <= 0) return 0;
  Tables t;
  make_tables(prm->nfeatures, prm->scale_factor, prm->nlevels, t);
  Img im{img, stride, rows, cols};
  Pyramid py;
  build_pyramid(im, nullptr, t, prm->nlevels, py);
  std::vector<std::vector<uint8_t>> blur(prm->nlevels);
  int n = 0;
  for (int i = 0; i < n_in; ++i) {
    orc_keypoint k = kps_in[i];
    const int l = k.octave;
    if (l < 0 || l >= prm->nlevels) continue;
    const float sc = t.scale[l];
    Img li{py.img[l].data(), py.cols[l], py.rows[l], py.cols[l]};
    k.x /= sc; k.y /= sc;
    if (!(k.y - EDGE_THRESHOLD >= 0 && k.y + EDGE_THRESHOLD < li.rows && k.x - EDGE_THRESHOLD >= 0 &&
          k.x + EDGE_THRESHOLD < li.cols))
      continue;
    const int th = std::min(std::max(prm->min_th_fast, 0), 255);
    if (!segment_test(li.p + (size_t)cv_round(k.y) * li.stride + cv_round(k.x), li.stride, th)) continue;
    k.angle = ic_angle(li, k.x, k.y, t.umax);
    k.size = PATCH_SIZE * sc;
    k.x *= sc; k.y *= sc;
    // CalcDescriptors: divide by scale again, describe on the blurred level
    orc_keypoint q = k;
    q.x /= sc; q.y /= sc;
    if (blur[l].empty()) {
      blur[l].resize((size_t)li.rows * li.cols);
      gauss7(li.p, li.stride, li.rows, li.cols, blur[l].data(), li.cols);
    }
    Img bi{blur[l].data(), li.cols, li.rows, li.cols};
    brief(bi, q.x, q.y, q.angle, desc_out + 32 * (size_t)n);
    kps_out[n++] = k;
  }
  return n;
}

}  // extern "C"
