// oracle/src/pg_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of LoopClosing::PoseGraphOptimization:
//   /root/reference/src/ssvio/loopclosing.cpp:458-539   graph: VertexPose per keyframe (some fixed), EdgePoseGraph
//                                                       per temporal / loop constraint, information = I, no kernel,
//                                                       BlockSolver<6,6> + LinearSolverEigen + LM, optimize(20)
//   /root/reference/include/ssvio/g2otypes.hpp:164-199  EdgePoseGraph: e = log(M^-1 * T0 * T1^-1); linearizeOplus is
//                                                       commented out -> g2o's central differences, delta = 1e-9
//                                                       (thirdparty/g2o/g2o/core/base_binary_edge.hpp:61-141)
//   thirdparty/sophus/sophus/se3.hpp:208-256, so3.hpp:218-286   inverse(), log()
//   thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:58-175   the LM iteration (same as the BA oracle)
// PINNED against the real reference arithmetic: oracle/_ref/libssvio_ref.so : ref_pose_graph and
// tests/golden/ref_golden.npz (tests/test_oracle_pg.py).  The linear solve is a dense Cholesky here (the reference
// uses Eigen's SimplicialLDLT): same solution up to rounding.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle.h"

namespace {

#include "se3_oracle.h"

// SE3::inverse, se3.hpp:208-211: SO3(conjugate) re-normalises the quaternion (so3.hpp:502)
inline void se3_inverse(const double* T, double* out)
{
  double q[4] = {-T[0], -T[1], -T[2], T[3]};
  const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= len;
  const double nt[3] = {T[4] * -1.0, T[5] * -1.0, T[6] * -1.0};
  double r[3];
  quat_rotate(q, nt, r);
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = r[0]; out[5] = r[1]; out[6] = r[2];
}

// SE3::log, se3.hpp:223-256 with SO3::logAndTheta, so3.hpp:245-286; tangent = (upsilon, omega)
inline void se3_log(const double* T, double* out)
{
  const double eps = 1e-10;
  const double squared_n = T[0] * T[0] + T[1] * T[1] + T[2] * T[2];
  const double w = T[3];
  double two_atan_nbyw_by_n, theta;
  if (squared_n < eps * eps) {
    const double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * (squared_n) / (w * squared_w);
    theta = 2.0 * squared_n / w;
  } else {
    const double n = std::sqrt(squared_n);
    if (std::fabs(w) < eps) two_atan_nbyw_by_n = (w > 0.0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
    else two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
    theta = two_atan_nbyw_by_n * n;
  }
  const double om[3] = {two_atan_nbyw_by_n * T[0], two_atan_nbyw_by_n * T[1], two_atan_nbyw_by_n * T[2]};
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += O[r * 3 + k] * O[k * 3 + c];
      O2[r * 3 + c] = s;
    }
  double Vinv[9];
  if (std::fabs(theta) < eps) {
    for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + (1. / 12.) * O2[i];
  } else {
    const double half_theta = 0.5 * theta;
    const double c = (1.0 - theta * std::cos(half_theta) / (2.0 * std::sin(half_theta))) / (theta * theta);
    for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c * O2[i];
  }
  for (int r = 0; r < 3; ++r) out[r] = Vinv[r * 3] * T[4] + Vinv[r * 3 + 1] * T[5] + Vinv[r * 3 + 2] * T[6];
  out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

// EdgePoseGraph::computeError, g2otypes.hpp:169-176
inline void pg_error(const double* M, const double* T0, const double* T1, double* e)
{
  double Mi[7], T1i[7], A[7], B[7];
  se3_inverse(M, Mi);
  se3_inverse(T1, T1i);
  se3_mul(Mi, T0, A);
  se3_mul(A, T1i, B);
  se3_log(B, e);
}

// base_binary_edge.hpp:61-141: central differences on the oplus of one vertex, delta = 1e-9
inline void pg_jac_numeric(const double* M, const double* T0, const double* T1, int which, double* J /*6x6 row-major*/)
{
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  for (int d = 0; d < 6; ++d) {
    double add[6] = {0, 0, 0, 0, 0, 0}, Tp[7], Tm[7], ep[6], em[6];
    add[d] = delta;
    pose_oplus(which == 0 ? T0 : T1, add, Tp);
    add[d] = -delta;
    pose_oplus(which == 0 ? T0 : T1, add, Tm);
    if (which == 0) { pg_error(M, Tp, T1, ep); pg_error(M, Tm, T1, em); }
    else { pg_error(M, T0, Tp, ep); pg_error(M, T0, Tm, em); }
    for (int r = 0; r < 6; ++r) J[r * 6 + d] = scalar * (ep[r] - em[r]);
  }
}

bool chol_solve_pg(std::vector<double>& A, int n, const double* b, double* x)
{
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * y[k];
    y[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

struct PG {
  int P, E;
  double* poses;
  const uint8_t* fixed;
  const int32_t *ei, *ej;
  const double* meas;
  std::vector<int> idx;      // free index of a pose or -1
  int nP = 0;
  std::vector<double> err, H, b, x;
  std::vector<uint8_t> active;

  void setup()
  {
    idx.assign(P, -1);
    for (int i = 0; i < P; ++i) if (!fixed[i]) idx[i] = nP++;
    err.assign(6 * (size_t)E, 0.0);
    active.assign(E, 0);
    // an edge whose vertices are all fixed is not active (sparse_optimizer.cpp:237)
    for (int k = 0; k < E; ++k) active[k] = (idx[ei[k]] >= 0 || idx[ej[k]] >= 0);
    x.assign(6 * (size_t)nP, 0.0);
  }
  void compute_errors()
  {
    for (int k = 0; k < E; ++k)
      if (active[k]) pg_error(meas + 7 * (size_t)k, poses + 7 * (size_t)ei[k], poses + 7 * (size_t)ej[k], &err[6 * (size_t)k]);
  }
  double chi2() const
  {
    double s = 0;
    for (int k = 0; k < E; ++k)
      if (active[k]) { double c = 0; for (int r = 0; r < 6; ++r) c += err[6 * (size_t)k + r] * err[6 * (size_t)k + r]; s += c; }
    return s;
  }
  // linearizeOplus + constructQuadraticForm (base_binary_edge.hpp:61-212) with Omega = I, no robust kernel
  void build_system()
  {
    const int n = 6 * nP;
    H.assign((size_t)n * n, 0.0);
    b.assign(n, 0.0);
    for (int k = 0; k < E; ++k) {
      if (!active[k]) continue;
      const int a = idx[ei[k]], c = idx[ej[k]];
      const double* e = &err[6 * (size_t)k];
      double Ji[36], Jj[36];
      if (a >= 0) pg_jac_numeric(meas + 7 * (size_t)k, poses + 7 * (size_t)ei[k], poses + 7 * (size_t)ej[k], 0, Ji);
      if (c >= 0) pg_jac_numeric(meas + 7 * (size_t)k, poses + 7 * (size_t)ei[k], poses + 7 * (size_t)ej[k], 1, Jj);
      if (a >= 0) {
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          for (int m = 0; m < 6; ++m) s += Ji[m * 6 + r] * (-e[m]);
          b[6 * a + r] += s;
          for (int q = 0; q < 6; ++q) {
            double h = 0;
            for (int m = 0; m < 6; ++m) h += Ji[m * 6 + r] * Ji[m * 6 + q];
            H[(size_t)(6 * a + r) * n + 6 * a + q] += h;
          }
        }
      }
      if (c >= 0) {
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          for (int m = 0; m < 6; ++m) s += Jj[m * 6 + r] * (-e[m]);
          b[6 * c + r] += s;
          for (int q = 0; q < 6; ++q) {
            double h = 0;
            for (int m = 0; m < 6; ++m) h += Jj[m * 6 + r] * Jj[m * 6 + q];
            H[(size_t)(6 * c + r) * n + 6 * c + q] += h;
          }
        }
      }
      if (a >= 0 && c >= 0 && a != c) {
        for (int r = 0; r < 6; ++r)
          for (int q = 0; q < 6; ++q) {
            double h = 0;
            for (int m = 0; m < 6; ++m) h += Ji[m * 6 + r] * Jj[m * 6 + q];
            H[(size_t)(6 * a + r) * n + 6 * c + q] += h;
            H[(size_t)(6 * c + q) * n + 6 * a + r] += h;
          }
      }
    }
  }
  double lambda_init() const
  {
    const int n = 6 * nP;
    double mx = 0;
    for (int j = 0; j < n; ++j) mx = std::max(std::fabs(H[(size_t)j * n + j]), mx);
    return 1e-5 * mx;
  }
  bool solve(double lambda)
  {
    const int n = 6 * nP;
    std::vector<double> A(H);
    for (int j = 0; j < n; ++j) A[(size_t)j * n + j] += lambda;
    return chol_solve_pg(A, n, b.data(), x.data());
  }
  void apply_update()
  {
    for (int i = 0; i < P; ++i) {
      if (idx[i] < 0) continue;
      double out[7];
      pose_oplus(poses + 7 * (size_t)i, &x[6 * (size_t)idx[i]], out);
      std::memcpy(poses + 7 * (size_t)i, out, sizeof(out));
    }
  }
  double compute_scale(double lambda) const
  {
    double s = 0;
    for (size_t j = 0; j < 6 * (size_t)nP; ++j) s += x[j] * (lambda * x[j] + b[j]);
    return s;
  }
};

}  // namespace

extern "C" {

void orc_se3_log(const double* pose7, double* tangent6) { se3_log(pose7, tangent6); }
void orc_se3_inverse(const double* pose7, double* out7) { se3_inverse(pose7, out7); }
void orc_pg_edge_eval(const double* meas7, const double* T0, const double* T1, double* err6, double* Ji36, double* Jj36)
{
  pg_error(meas7, T0, T1, err6);
  if (Ji36) pg_jac_numeric(meas7, T0, T1, 0, Ji36);
  if (Jj36) pg_jac_numeric(meas7, T0, T1, 1, Jj36);
}

int orc_pose_graph_opt(int P, double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                       const double* meas7, int iterations, double* edge_err_out, int stats_cap, int* stats_n,
                       double* stats_chi2, double* stats_lambda, int* stats_trials)
{
  PG g;
  g.P = P; g.E = E; g.poses = poses; g.fixed = fixed; g.ei = ei; g.ej = ej; g.meas = meas7;
  g.setup();
  if (stats_n) *stats_n = 0;
  if (g.nP == 0) return -1;
  double lambda = -1, ni = 2;
  int done = 0;
  std::vector<double> bak(7 * (size_t)P);
  for (int it = 0; it < iterations; ++it) {
    g.compute_errors();
    double currentChi = g.chi2();
    double tempChi = currentChi;
    g.build_system();
    if (it == 0) { lambda = g.lambda_init(); ni = 2; }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      std::memcpy(bak.data(), poses, sizeof(double) * 7 * P);
      const bool ok2 = g.solve(lambda);
      g.apply_update();
      g.compute_errors();
      tempChi = g.chi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = g.compute_scale(lambda);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        std::memcpy(poses, bak.data(), sizeof(double) * 7 * P);
        if (!std::isfinite(lambda)) { lambda_bad = true; break; }
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    ++done;
    if (stats_n && *stats_n < stats_cap) {
      const int k = *stats_n;
      if (stats_chi2) stats_chi2[k] = g.chi2();
      if (stats_lambda) stats_lambda[k] = lambda;
      if (stats_trials) stats_trials[k] = qmax;
      *stats_n = k + 1;
    }
    if (qmax == 10 || rho == 0 || lambda_bad) break;
  }
  if (edge_err_out) std::memcpy(edge_err_out, g.err.data(), sizeof(double) * 6 * (size_t)E);
  return done;
}

}  // extern "C"
