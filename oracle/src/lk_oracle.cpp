// oracle/src/lk_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of cv::calcOpticalFlowPyrLK as the reference calls it:
//   frontend.cpp:156-166 (TrackLastFrame: last left -> current left) and frontend.cpp:374-384 (FindFeaturesInRight:
//   left -> right), both with winSize 11x11, maxLevel 3, TermCriteria(COUNT+EPS, 30, 0.01),
//   OPTFLOW_USE_INITIAL_FLOW, default minEigThreshold 1e-4, err = mean absolute patch difference / 32.
// OpenCV is neither vendored nor installed (cmake/packages.cmake:9), so this follows the published OpenCV 3.x
// algorithm (modules/video/src/lkpyramid.cpp: buildOpticalFlowPyramid, calcSharrDeriv, LKTrackerInvoker;
// modules/imgproc/src/pyramids.cpp: pyrDown 8u) -- PARITY UNPINNED vs OpenCV.
//
// One deliberate, documented choice: the 2x2 normal matrix and the mismatch vector are sums of INTEGER products
// (OpenCV accumulates them in float, in an order that depends on its SIMD build).  They are accumulated exactly
// in 64-bit integers here and converted to float once, which makes the result independent of the summation
// order (and therefore reproducible by a parallel reduction on the GPU); vs OpenCV's scalar float accumulation
// the normal matrix differs by ~1e-7 relative.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline int refl101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
  return i;
}
inline int cv_round(float v) { return (int)std::nearbyintf(v); }     // RNE (default rounding mode)
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

struct Img {
  int rows = 0, cols = 0;
  std::vector<uint8_t> px;         // rows x cols
  std::vector<int16_t> dxy;        // rows x cols x 2 (Scharr), filled for the "previous" pyramid only
  inline int at(int y, int x) const { return px[(size_t)refl101(y, rows) * cols + refl101(x, cols)]; }   // BORDER_REFLECT_101 border of the pyramid
  inline int dx(int y, int x) const { return (y < 0 || y >= rows || x < 0 || x >= cols) ? 0 : dxy[((size_t)y * cols + x) * 2]; }
  inline int dy(int y, int x) const { return (y < 0 || y >= rows || x < 0 || x >= cols) ? 0 : dxy[((size_t)y * cols + x) * 2 + 1]; }
};

// pyrDown, 8-bit: separable 1-4-6-4-1, (sum + 128) >> 8, BORDER_REFLECT_101, dst = ((cols+1)/2, (rows+1)/2)
void pyr_down(const Img& s, Img& d)
{
  d.rows = (s.rows + 1) / 2; d.cols = (s.cols + 1) / 2;
  d.px.assign((size_t)d.rows * d.cols, 0);
  static const int w[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < d.rows; ++y)
    for (int x = 0; x < d.cols; ++x) {
      int sum = 0;
      for (int ky = 0; ky < 5; ++ky) {
        const int sy = refl101(2 * y + ky - 2, s.rows);
        int row = 0;
        for (int kx = 0; kx < 5; ++kx) row += w[kx] * s.px[(size_t)sy * s.cols + refl101(2 * x + kx - 2, s.cols)];
        sum += w[ky] * row;
      }
      d.px[(size_t)y * d.cols + x] = (uint8_t)((sum + 128) >> 8);
    }
}

// calcSharrDeriv: dx = [3 10 3]^T (x) [-1 0 1], dy = [-1 0 1]^T (x) [3 10 3], rows/columns reflected (101) at the image edge
void scharr(Img& im)
{
  im.dxy.assign((size_t)im.rows * im.cols * 2, 0);
  std::vector<int> t0(im.cols + 2), t1(im.cols + 2);
  for (int y = 0; y < im.rows; ++y) {
    const int y0 = y > 0 ? y - 1 : (im.rows > 1 ? 1 : 0), y2 = y < im.rows - 1 ? y + 1 : (im.rows > 1 ? im.rows - 2 : 0);
    const uint8_t *r0 = &im.px[(size_t)y0 * im.cols], *r1 = &im.px[(size_t)y * im.cols], *r2 = &im.px[(size_t)y2 * im.cols];
    for (int x = 0; x < im.cols; ++x) {
      t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
      t1[x + 1] = r2[x] - r0[x];
    }
    const int xl = im.cols > 1 ? 1 : 0, xr = im.cols > 1 ? im.cols - 2 : 0;
    t0[0] = t0[xl + 1]; t1[0] = t1[xl + 1];
    t0[im.cols + 1] = t0[xr + 1]; t1[im.cols + 1] = t1[xr + 1];
    for (int x = 0; x < im.cols; ++x) {
      im.dxy[((size_t)y * im.cols + x) * 2] = (int16_t)(t0[x + 2] - t0[x]);
      im.dxy[((size_t)y * im.cols + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
    }
  }
}

// buildOpticalFlowPyramid: stops before a level not larger than the window
int build_pyramid(const uint8_t* img, int stride, int rows, int cols, int win, int max_level, std::vector<Img>& pyr)
{
  pyr.assign(1, Img());
  pyr[0].rows = rows; pyr[0].cols = cols; pyr[0].px.resize((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) memcpy(&pyr[0].px[(size_t)y * cols], img + (size_t)y * stride, cols);
  for (int level = 1; level <= max_level; ++level) {
    const int w = (pyr[level - 1].cols + 1) / 2, h = (pyr[level - 1].rows + 1) / 2;
    if (w <= win || h <= win) return level - 1;
    pyr.emplace_back();
    pyr_down(pyr[level - 1], pyr[level]);
  }
  return max_level;
}

}  // namespace

extern "C" {

void orc_lk_default_params(orc_lk_params* p)
{
  p->win = 11; p->max_level = 3; p->max_iters = 30; p->eps = 0.01; p->min_eig_threshold = 1e-4f; p->use_initial_flow = 1;
}

void orc_lk_pyr_down(const uint8_t* src, int sstride, int rows, int cols, uint8_t* dst, int dstride)
{
  Img s, d;
  s.rows = rows; s.cols = cols; s.px.resize((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) memcpy(&s.px[(size_t)y * cols], src + (size_t)y * sstride, cols);
  pyr_down(s, d);
  for (int y = 0; y < d.rows; ++y) memcpy(dst + (size_t)y * dstride, &d.px[(size_t)y * d.cols], d.cols);
}

void orc_lk_scharr(const uint8_t* src, int sstride, int rows, int cols, int16_t* dxy)
{
  Img s;
  s.rows = rows; s.cols = cols; s.px.resize((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) memcpy(&s.px[(size_t)y * cols], src + (size_t)y * sstride, cols);
  scharr(s);
  memcpy(dxy, s.dxy.data(), sizeof(int16_t) * s.dxy.size());
}

int orc_lk_track(const uint8_t* prev, int pstride, const uint8_t* next, int nstride, int rows, int cols, int n,
                 const float* prev_pts, float* next_pts, uint8_t* status, float* err, const orc_lk_params* prm)
{
  const int win = prm->win;
  std::vector<Img> P, Q;
  const int l1 = build_pyramid(prev, pstride, rows, cols, win, prm->max_level, P);
  const int l2 = build_pyramid(next, nstride, rows, cols, win, prm->max_level, Q);
  const int max_level = l1 < l2 ? l1 : l2;
  const int max_count = prm->max_iters < 0 ? 0 : (prm->max_iters > 100 ? 100 : prm->max_iters);
  double eps = prm->eps < 0. ? 0. : (prm->eps > 10. ? 10. : prm->eps);
  eps *= eps;
  for (int i = 0; i < n; ++i) { status[i] = 1; if (err) err[i] = 0.f; }
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  std::vector<int> Iw((size_t)win * win), Ix((size_t)win * win), Iy((size_t)win * win);
  for (int level = max_level; level >= 0; --level) {
    Img& I = P[level];
    const Img& J = Q[level];
    scharr(I);
    for (int i = 0; i < n; ++i) {
      float px = prev_pts[2 * i] * (float)(1. / (1 << level)), py = prev_pts[2 * i + 1] * (float)(1. / (1 << level));
      float nx, ny;
      if (level == max_level) {
        if (prm->use_initial_flow) { nx = next_pts[2 * i] * (float)(1. / (1 << level)); ny = next_pts[2 * i + 1] * (float)(1. / (1 << level)); }
        else { nx = px; ny = py; }
      } else { nx = next_pts[2 * i] * 2.f; ny = next_pts[2 * i + 1] * 2.f; }
      next_pts[2 * i] = nx; next_pts[2 * i + 1] = ny;
      px -= half; py -= half;
      const int ipx = cv_floor(px), ipy = cv_floor(py);
      // a non-finite position is outside every image (OpenCV leaves this case to the float -> int conversion,
      // which is undefined for NaN / inf; the kernel and this restatement agree on "outside")
      if (!(std::isfinite(px) && std::isfinite(py)) || ipx < -win || ipx >= I.cols || ipy < -win || ipy >= I.rows) {
        if (level == 0) { status[i] = 0; if (err) err[i] = 0.f; }
        continue;
      }
      float a = px - ipx, b = py - ipy;
      int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS)), iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cv_round((1.f - a) * b * (1 << W_BITS)), iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      long long sA11 = 0, sA12 = 0, sA22 = 0;
      for (int y = 0; y < win; ++y)
        for (int x = 0; x < win; ++x) {
          const int yy = y + ipy, xx = x + ipx;
          const int ival = descale(I.at(yy, xx) * iw00 + I.at(yy, xx + 1) * iw01 + I.at(yy + 1, xx) * iw10 + I.at(yy + 1, xx + 1) * iw11, W_BITS - 5);
          const int ixv = descale(I.dx(yy, xx) * iw00 + I.dx(yy, xx + 1) * iw01 + I.dx(yy + 1, xx) * iw10 + I.dx(yy + 1, xx + 1) * iw11, W_BITS);
          const int iyv = descale(I.dy(yy, xx) * iw00 + I.dy(yy, xx + 1) * iw01 + I.dy(yy + 1, xx) * iw10 + I.dy(yy + 1, xx + 1) * iw11, W_BITS);
          Iw[y * win + x] = (int16_t)ival; Ix[y * win + x] = (int16_t)ixv; Iy[y * win + x] = (int16_t)iyv;
          sA11 += (long long)ixv * ixv; sA12 += (long long)ixv * iyv; sA22 += (long long)iyv * iyv;
        }
      const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if (minEig < prm->min_eig_threshold || D < 1.1920928955078125e-7f) {
        if (level == 0) status[i] = 0;
        continue;
      }
      D = 1.f / D;
      nx -= half; ny -= half;
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < max_count; ++j) {
        const int inx = cv_floor(nx), iny = cv_floor(ny);
        if (!(std::isfinite(nx) && std::isfinite(ny)) || inx < -win || inx >= J.cols || iny < -win || iny >= J.rows) {
          if (level == 0) status[i] = 0;
          break;
        }
        a = nx - inx; b = ny - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS)); iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS)); iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        long long sb1 = 0, sb2 = 0;
        for (int y = 0; y < win; ++y)
          for (int x = 0; x < win; ++x) {
            const int yy = y + iny, xx = x + inx;
            const int diff = descale(J.at(yy, xx) * iw00 + J.at(yy, xx + 1) * iw01 + J.at(yy + 1, xx) * iw10 + J.at(yy + 1, xx + 1) * iw11, W_BITS - 5) - Iw[y * win + x];
            sb1 += (long long)diff * Ix[y * win + x]; sb2 += (long long)diff * Iy[y * win + x];
          }
        const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
        nx += dx; ny += dy;
        next_pts[2 * i] = nx + half; next_pts[2 * i + 1] = ny + half;
        if ((double)dx * dx + (double)dy * dy <= eps) break;
        if (j > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) {
          next_pts[2 * i] -= dx * 0.5f; next_pts[2 * i + 1] -= dy * 0.5f;
          break;
        }
        pdx = dx; pdy = dy;
      }
      if (status[i] && err && level == 0) {
        const float ex = next_pts[2 * i] - half, ey = next_pts[2 * i + 1] - half;
        const int inx = cv_floor(ex), iny = cv_floor(ey);
        if (!(std::isfinite(ex) && std::isfinite(ey)) || inx < -win || inx >= J.cols || iny < -win || iny >= J.rows) { status[i] = 0; continue; }
        const float aa = ex - inx, bb = ey - iny;
        iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS)); iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS)); iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        long long e = 0;
        for (int y = 0; y < win; ++y)
          for (int x = 0; x < win; ++x) {
            const int yy = y + iny, xx = x + inx;
            const int diff = descale(J.at(yy, xx) * iw00 + J.at(yy, xx + 1) * iw01 + J.at(yy + 1, xx) * iw10 + J.at(yy + 1, xx + 1) * iw11, W_BITS - 5) - Iw[y * win + x];
            e += std::abs(diff);
          }
        err[i] = (float)e * (1.f / (float)(32 * win * win));
      }
    }
  }
  return max_level;
}

}  // extern "C"
