// oracle/src/stereo_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
//  * orc_triangulate: ssvio::triangulation (/root/reference/include/ssvio/algorithm.hpp:23-45) for the stereo rig of
//    System::GenerateSteroCamera (src/ssvio/system.cpp:63,71) with the acceptance test of FrontEnd::BuidInitMap /
//    TriangulateNewPoints (src/ssvio/frontend.cpp:466,528) and Camera::pixel2camera (src/ssvio/camera.cpp:25-30).
//    The reference uses Eigen's bdcSvd; here the 4x4 SVD is a one-sided Jacobi (Hestenes) iteration, pinned
//    against the real function through oracle/_ref (tests/test_oracle_geom.py).
//  * orc_stereo_match: the row-band Hamming stereo matcher BASELINE.json's north_star asks for.  The reference
//    has NO such function (its stereo association is cv::calcOpticalFlowPyrLK, frontend.cpp:374-384); the
//    semantics are defined here on top of the only Hamming matching the reference does,
//    OpenCV BruteForce-Hamming match() (src/ssvio/loopclosing.cpp:24,108): minimum distance, lowest train
//    index wins ties.
//  * orc_bf_match: that brute-force matcher itself (loopclosing.cpp:105-110).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline void quat_rotate(const double* q, const double* p, double* out)
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  double ux = y * p[2] - z * p[1], uy = z * p[0] - x * p[2], uz = x * p[1] - y * p[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = p[0] + w * ux + (y * uz - z * uy);
  out[1] = p[1] + w * uy + (z * ux - x * uz);
  out[2] = p[2] + w * uz + (x * uy - y * ux);
}

// one-sided Jacobi SVD of a 4x4 (row-major A): singular values (descending) and V (columns)
void svd4(const double* A, double* sv, double* V)
{
  double U[16];
  for (int i = 0; i < 16; ++i) { U[i] = A[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 4; ++r) {
          alpha += U[r * 4 + p] * U[r * 4 + p];
          beta += U[r * 4 + q] * U[r * 4 + q];
          gamma += U[r * 4 + p] * U[r * 4 + q];
        }
        if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta) || gamma == 0.0) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 4; ++r) {
          const double up = U[r * 4 + p], uq = U[r * 4 + q];
          U[r * 4 + p] = c * up - s * uq;
          U[r * 4 + q] = s * up + c * uq;
          const double vp = V[r * 4 + p], vq = V[r * 4 + q];
          V[r * 4 + p] = c * vp - s * vq;
          V[r * 4 + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double n[4];
  for (int c = 0; c < 4; ++c) {
    double s = 0;
    for (int r = 0; r < 4; ++r) s += U[r * 4 + c] * U[r * 4 + c];
    n[c] = std::sqrt(s);
  }
  // sort descending (stable selection), permuting V's columns
  int ord[4] = {0, 1, 2, 3};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 4; ++j)
      if (n[ord[j]] > n[ord[i]]) { int tmp = ord[i]; ord[i] = ord[j]; ord[j] = tmp; }
  double Vs[16];
  for (int c = 0; c < 4; ++c) {
    sv[c] = n[ord[c]];
    for (int r = 0; r < 4; ++r) Vs[r * 4 + c] = V[r * 4 + ord[c]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

inline int popcount256(const uint8_t* a, const uint8_t* b)
{
  int d = 0;
  for (int k = 0; k < 32; ++k) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
  return d;
}

}  // namespace

extern "C" {

void orc_triangulate(int n, const double* uvL, const double* uvR, double fx, double fy, double cx, double cy,
                     double baseline, const double* T_wc7, double* xyz_out, uint8_t* ok_out, double* ratio_out)
{
  for (int i = 0; i < n; ++i) {
    // pixel2camera with depth 1 (camera.cpp:25-30)
    const double x1 = (uvL[2 * i] - cx) / fx * 1.0, y1 = (uvL[2 * i + 1] - cy) / fy * 1.0;
    const double x2 = (uvR[2 * i] - cx) / fx * 1.0, y2 = (uvR[2 * i + 1] - cy) / fy * 1.0;
    // rows x*m2 - m0, y*m2 - m1 of the 3x4 poses [I|0] and [I|(-baseline,0,0)]  (algorithm.hpp:30-35)
    const double tx = -baseline;
    const double A[16] = {-1, 0, x1, 0,
                          0, -1, y1, 0,
                          -1, 0, x2, x2 * 0.0 - tx,
                          0, -1, y2, y2 * 0.0 - 0.0};
    double sv[4], V[16];
    svd4(A, sv, V);
    const double w = V[3 * 4 + 3];
    double p[3] = {V[0 * 4 + 3] / w, V[1 * 4 + 3] / w, V[2 * 4 + 3] / w};
    const double ratio = sv[3] / sv[2];
    // no positive disparity = a point at or behind infinity: w (and the sign of z) is rounding noise when uL == uR, so
    // the callers' z > 0 test is decided on the disparity and the point is zeroed (same rule in the kernel)
    const bool positive_disparity = uvL[2 * i] - uvR[2 * i] > 0.0;
    const bool ok = (ratio < 1e-2) && (p[2] > 0) && positive_disparity;
    if (!positive_disparity) { p[0] = 0.0; p[1] = 0.0; p[2] = 0.0; }
    if (T_wc7) {  // new map points: world = T_wc * p_cam1 (frontend.cpp:503,531)
      double r[3];
      quat_rotate(T_wc7, p, r);
      p[0] = r[0] + T_wc7[4]; p[1] = r[1] + T_wc7[5]; p[2] = r[2] + T_wc7[6];
    }
    xyz_out[3 * i] = p[0]; xyz_out[3 * i + 1] = p[1]; xyz_out[3 * i + 2] = p[2];
    ok_out[i] = ok ? 1 : 0;
    if (ratio_out) ratio_out[i] = ratio;
  }
}

void orc_stereo_match(const orc_keypoint* kL, const uint8_t* dL, int nL, const orc_keypoint* kR, const uint8_t* dR,
                      int nR, const orc_match_params* prm, int32_t* match_idx, int32_t* dist)
{
  float scale[32];
  scale[0] = 1.0f;
  for (int i = 1; i < 32; ++i) scale[i] = scale[i - 1] * prm->scale_factor;   // mvScaleFactor, orbextractor.cpp:135-143
  for (int i = 0; i < nL; ++i) {
    const int ol = kL[i].octave < 0 ? 0 : (kL[i].octave > 31 ? 31 : kL[i].octave);
    const float band = prm->band_px * scale[ol];
    int best = 257, bj = -1;
    for (int j = 0; j < nR; ++j) {
      const float dv = kL[i].y - kR[j].y;
      if (dv > band || -dv > band) continue;
      int doct = kL[i].octave - kR[j].octave;
      if (doct < 0) doct = -doct;
      if (doct > prm->max_octave_diff) continue;
      const float disp = kL[i].x - kR[j].x;
      if (disp < prm->min_disp || disp > prm->max_disp) continue;
      const int d = popcount256(dL + 32 * (size_t)i, dR + 32 * (size_t)j);
      if (d < best) { best = d; bj = j; }
    }
    dist[i] = best;
    match_idx[i] = (bj >= 0 && best <= prm->max_dist) ? bj : -1;
  }
}

void orc_bf_match(const uint8_t* dq, int nq, const uint8_t* dt, int nt, int32_t* idx, int32_t* dist)
{
  for (int i = 0; i < nq; ++i) {
    int best = 257, bj = -1;
    for (int j = 0; j < nt; ++j) {
      const int d = popcount256(dq + 32 * (size_t)i, dt + 32 * (size_t)j);
      if (d < best) { best = d; bj = j; }
    }
    idx[i] = bj;
    dist[i] = best;
  }
}

}  // extern "C"
