// oracle/src/ba_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference's bundle-adjustment arithmetic, plain C++17, no Eigen/g2o/Sophus.
// Each block cites the reference lines it restates (paths relative to /root/reference).
// Pinned against oracle/_ref/libssvio_ref.so (the real reference code) by tests/test_oracle_ba.py.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <algorithm>
#include <vector>

#include "oracle.h"

namespace {

#include "se3_oracle.h"

// ---------------------------------------------------------------------------------------------
// Edge model
// ---------------------------------------------------------------------------------------------
struct Cam { double fx, fy, cx, cy; };

// EdgeProjection::computeError, g2otypes.hpp:123-131:  e = z - hnorm(K * (ext * (T * p)))
inline void edge_error(const double* T, const double* p, const double* ext, const Cam& K,
                       const double* uv, double* e, double* pc_out = nullptr)
{
  double p1[3], pc[3];
  se3_act(T, p, p1);
  se3_act(ext, p1, pc);
  // K * pc (full 3x3 product incl. the zero terms), then division by the third component
  const double hx = K.fx * pc[0] + 0.0 * pc[1] + K.cx * pc[2];
  const double hy = 0.0 * pc[0] + K.fy * pc[1] + K.cy * pc[2];
  const double hz = 0.0 * pc[0] + 0.0 * pc[1] + 1.0 * pc[2];
  e[0] = uv[0] - hx / hz;
  e[1] = uv[1] - hy / hz;
  if (pc_out) { pc_out[0] = pc[0]; pc_out[1] = pc[1]; pc_out[2] = pc[2]; }
}

// Numeric Jacobians: thirdparty/g2o/g2o/core/base_binary_edge.hpp:144-212 (delta 1e-9, central).
inline void edge_jac_numeric(const double* T, const double* p, const double* ext, const Cam& K,
                             const double* uv, bool pose_free, bool point_free, double* Ji, double* Jj)
{
  const double delta = 1e-9;
  const double scalar = 1 / (2 * delta);
  if (pose_free) {
    double add[6] = {0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 6; ++d) {
      double Tp[7], e1[2], e2[2];
      add[d] = delta;
      pose_oplus(T, add, Tp);
      edge_error(Tp, p, ext, K, uv, e1);
      add[d] = -delta;
      pose_oplus(T, add, Tp);
      edge_error(Tp, p, ext, K, uv, e2);
      add[d] = 0.0;
      Ji[0 * 6 + d] = scalar * (e1[0] - e2[0]);
      Ji[1 * 6 + d] = scalar * (e1[1] - e2[1]);
    }
  }
  if (point_free) {
    for (int d = 0; d < 3; ++d) {
      double pp[3] = {p[0], p[1], p[2]}, e1[2], e2[2];
      pp[d] = p[d] + delta;
      edge_error(T, pp, ext, K, uv, e1);
      // g2o: pop() restores the estimate, then oplus(-delta)
      pp[d] = p[d] + (-delta);
      edge_error(T, pp, ext, K, uv, e2);
      Jj[0 * 3 + d] = scalar * (e1[0] - e2[0]);
      Jj[1 * 3 + d] = scalar * (e1[1] - e2[1]);
    }
  }
}

// Analytic Jacobians: the formula the reference left commented out (g2otypes.hpp:133-153), generalised
// to a non-identity cam_ext (for ext = I it is exactly that formula):
//   d e / d xi = -dproj(pc) * R_ext * [ I | -[T p]x ],    d e / d p = -dproj(pc) * R_ext * R_T
inline void edge_jac_analytic(const double* T, const double* p, const double* ext, const Cam& K,
                              double* Ji, double* Jj)
{
  double p1[3], pc[3];
  se3_act(T, p, p1);
  se3_act(ext, p1, pc);
  const double X = pc[0], Y = pc[1], Z = pc[2];
  const double Zinv = 1.0 / (Z + 1e-18);
  const double Zinv2 = Zinv * Zinv;
  // A = d e / d pc  (2x3)
  const double A[6] = {-K.fx * Zinv, 0, K.fx * X * Zinv2, 0, -K.fy * Zinv, K.fy * Y * Zinv2};
  double Re[9], Rt[9];
  quat_to_R(ext, Re);
  quat_to_R(T, Rt);
  double AR[6];  // A * R_ext
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      AR[r * 3 + c] = A[r * 3] * Re[c] + A[r * 3 + 1] * Re[3 + c] + A[r * 3 + 2] * Re[6 + c];
  // -[p1]x
  const double H[9] = {0, p1[2], -p1[1], -p1[2], 0, p1[0], p1[1], -p1[0], 0};
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = AR[r * 3 + c];
      Ji[r * 6 + 3 + c] = AR[r * 3] * H[c] + AR[r * 3 + 1] * H[3 + c] + AR[r * 3 + 2] * H[6 + c];
      Jj[r * 3 + c] = AR[r * 3] * Rt[c] + AR[r * 3 + 1] * Rt[3 + c] + AR[r * 3 + 2] * Rt[6 + c];
    }
  }
}

// RobustKernelHuber::robustify, thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-78
inline void huber(double e2, double delta, double* rho)
{
  const double dsqr = delta * delta;
  if (e2 <= dsqr) {
    rho[0] = e2; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e2);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e2;
  }
}

// 3x3 inverse by cofactors (what Eigen's fixed-size inverse() does, block_solver.hpp:356)
inline bool inv3(const double* m, double* o)
{
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[5] * m[6] - m[3] * m[8];
  const double c20 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
  const double invdet = 1.0 / det;
  o[0] = c00 * invdet;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * invdet;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * invdet;
  o[3] = c10 * invdet;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * invdet;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * invdet;
  o[6] = c20 * invdet;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * invdet;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * invdet;
  return std::isfinite(invdet);
}

// Dense Cholesky solve (LL^T) of a symmetric system; false when not positive definite -- the role
// of LinearSolverCSparse::solve (solvers/csparse/linear_solver_csparse.h:106-142) / LinearSolverDense.
bool chol_solve(std::vector<double>& A, int n, const double* b, double* x)
{
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i * n + k] * y[k];
    y[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * x[k];
    x[i] = s / A[i * n + i];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// The optimizer: SparseOptimizer + BlockSolver<6,3> + OptimizationAlgorithmLevenberg restated.
// ---------------------------------------------------------------------------------------------
struct BA {
  int P, L, E;
  double* poses; double* points;
  const uint8_t *pose_fixed, *point_fixed;
  const int32_t *edge_pose, *edge_point;
  const double* edge_uv; const uint8_t* edge_cam;
  Cam K; const double* ext;
  double huber_delta; int jac_mode;
  bool use_kernel = true;

  // index mapping (sparse_optimizer.cpp:168-192, buildIndexMapping: poses first, then landmarks)
  std::vector<int> pose_idx, lm_idx;  // -1 when fixed
  int nP = 0, nL = 0;
  // per-landmark unique (pose, block) lists, sorted by pose index (Hpl column, block_solver.hpp:196-222)
  struct PL { int pidx; int slot; };
  std::vector<std::vector<PL>> lm_cols;   // indexed by free-landmark index
  std::vector<int> edge_slot;             // edge -> Hpl slot (or -1)
  int nslots = 0;

  std::vector<double> err, Hpp, bp, Hll, bl, Hpl, x;
  std::vector<uint8_t> active;            // edge active (level 0)

  void init_structure()
  {
    pose_idx.assign(P, -1); lm_idx.assign(L, -1);
    nP = nL = 0;
    for (int i = 0; i < P; ++i) if (!(pose_fixed && pose_fixed[i])) pose_idx[i] = nP++;
    for (int j = 0; j < L; ++j) if (!(point_fixed && point_fixed[j])) lm_idx[j] = nL++;
    lm_cols.assign(nL, {});
    edge_slot.assign(E, -1);
    nslots = 0;
    for (int e = 0; e < E; ++e) {
      if (!active[e]) continue;
      const int pi = pose_idx[edge_pose[e]], li = lm_idx[edge_point[e]];
      if (pi < 0 || li < 0) continue;
      auto& col = lm_cols[li];
      int slot = -1;
      for (auto& pl : col) if (pl.pidx == pi) slot = pl.slot;
      if (slot < 0) { slot = nslots++; col.push_back({pi, slot}); }
      edge_slot[e] = slot;
    }
    for (auto& col : lm_cols)
      std::sort(col.begin(), col.end(), [](const PL& a, const PL& b) { return a.pidx < b.pidx; });
    err.assign(2 * (size_t)E, 0.0);
    Hpp.assign(36 * (size_t)nP, 0.0); bp.assign(6 * (size_t)nP, 0.0);
    Hll.assign(9 * (size_t)nL, 0.0); bl.assign(3 * (size_t)nL, 0.0);
    Hpl.assign(18 * (size_t)nslots, 0.0);
    x.assign(6 * (size_t)nP + 3 * (size_t)nL, 0.0);
  }

  // SparseOptimizer::computeActiveErrors (sparse_optimizer.cpp:63-90)
  void compute_errors()
  {
    for (int e = 0; e < E; ++e) {
      if (!active[e]) continue;
      edge_error(poses + 7 * edge_pose[e], points + 3 * edge_point[e], ext + 7 * (edge_cam ? edge_cam[e] : 0),
                 K, edge_uv + 2 * e, &err[2 * e]);
    }
  }
  double edge_chi2(int e) const { return err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]; }

  // SparseOptimizer::activeRobustChi2 (sparse_optimizer.cpp:102-116)
  double robust_chi2() const
  {
    double chi = 0;
    for (int e = 0; e < E; ++e) {
      if (!active[e]) continue;
      if (use_kernel) { double rho[3]; huber(edge_chi2(e), huber_delta, rho); chi += rho[0]; }
      else chi += edge_chi2(e);
    }
    return chi;
  }

  // BlockSolver::buildSystem (block_solver.hpp:463-521) + constructQuadraticForm (base_binary_edge.hpp:61-134)
  void build_system()
  {
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
    std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
    std::fill(Hpl.begin(), Hpl.end(), 0.0);
    for (int e = 0; e < E; ++e) {
      if (!active[e]) continue;
      const int pi = pose_idx[edge_pose[e]], li = lm_idx[edge_point[e]];
      if (pi < 0 && li < 0) continue;
      const double* T = poses + 7 * edge_pose[e];
      const double* p = points + 3 * edge_point[e];
      const double* ex = ext + 7 * (edge_cam ? edge_cam[e] : 0);
      double Ji[12] = {0}, Jj[6] = {0};
      if (jac_mode == 1) edge_jac_numeric(T, p, ex, K, edge_uv + 2 * e, pi >= 0, li >= 0, Ji, Jj);
      else edge_jac_analytic(T, p, ex, K, Ji, Jj);
      const double* er = &err[2 * e];
      double w = 1.0;
      if (use_kernel) { double rho[3]; huber(edge_chi2(e), huber_delta, rho); w = rho[1]; }
      // omega_r = -Omega e, scaled by rho[1]; weightedOmega = rho[1] * I
      const double r0 = -er[0] * w, r1 = -er[1] * w;
      if (pi >= 0) {
        double* H = &Hpp[36 * (size_t)pi]; double* b = &bp[6 * (size_t)pi];
        for (int a = 0; a < 6; ++a) {
          b[a] += Ji[a] * r0 + Ji[6 + a] * r1;
          for (int c = 0; c < 6; ++c) H[a * 6 + c] += Ji[a] * w * Ji[c] + Ji[6 + a] * w * Ji[6 + c];
        }
        if (li >= 0) {
          double* W = &Hpl[18 * (size_t)edge_slot[e]];
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 3; ++c) W[a * 3 + c] += Ji[a] * w * Jj[c] + Ji[6 + a] * w * Jj[3 + c];
        }
      }
      if (li >= 0) {
        double* H = &Hll[9 * (size_t)li]; double* b = &bl[3 * (size_t)li];
        for (int a = 0; a < 3; ++a) {
          b[a] += Jj[a] * r0 + Jj[3 + a] * r1;
          for (int c = 0; c < 3; ++c) H[a * 3 + c] += Jj[a] * w * Jj[c] + Jj[3 + a] * w * Jj[3 + c];
        }
      }
    }
  }

  // OptimizationAlgorithmLevenberg::computeLambdaInit (optimization_algorithm_levenberg.cpp:152-166)
  double lambda_init() const
  {
    double m = 0;
    for (int i = 0; i < nP; ++i) for (int d = 0; d < 6; ++d) m = std::max(std::fabs(Hpp[36 * (size_t)i + d * 7]), m);
    for (int j = 0; j < nL; ++j) for (int d = 0; d < 3; ++d) m = std::max(std::fabs(Hll[9 * (size_t)j + d * 4]), m);
    return 1e-5 * m;
  }

  // BlockSolver::setLambda + solve (block_solver.hpp:525-548, 315-447).  false = not SPD.
  bool solve(double lambda)
  {
    const int n = 6 * nP;
    std::fill(x.begin(), x.end(), 0.0);
    if (nL == 0) {  // !_doSchur
      if (n == 0) return true;
      std::vector<double> S((size_t)n * n, 0.0);
      for (int i = 0; i < nP; ++i)
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c)
            S[(size_t)(6 * i + a) * n + 6 * i + c] = Hpp[36 * (size_t)i + a * 6 + c] + (a == c ? lambda : 0.0);
      return chol_solve(S, n, bp.data(), x.data());
    }
    std::vector<double> S((size_t)n * n, 0.0), coeff(n, 0.0), Dinv(9 * (size_t)nL);
    for (int i = 0; i < nP; ++i)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 6; ++c)
          S[(size_t)(6 * i + a) * n + 6 * i + c] = Hpp[36 * (size_t)i + a * 6 + c] + (a == c ? lambda : 0.0);
    for (int j = 0; j < nL; ++j) {
      double D[9];
      for (int k = 0; k < 9; ++k) D[k] = Hll[9 * (size_t)j + k] + ((k % 4 == 0) ? lambda : 0.0);
      double* Di = &Dinv[9 * (size_t)j];
      inv3(D, Di);
      const double* b = &bl[3 * (size_t)j];
      double db[3];
      for (int r = 0; r < 3; ++r) db[r] = Di[r * 3] * b[0] + Di[r * 3 + 1] * b[1] + Di[r * 3 + 2] * b[2];
      const auto& col = lm_cols[j];
      for (size_t o = 0; o < col.size(); ++o) {
        const int i1 = col[o].pidx;
        const double* Bi = &Hpl[18 * (size_t)col[o].slot];
        double BD[18];
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 3; ++c)
            BD[a * 3 + c] = Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c] + Bi[a * 3 + 2] * Di[6 + c];
        for (int a = 0; a < 6; ++a)
          coeff[6 * i1 + a] += Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
        for (size_t o2 = o; o2 < col.size(); ++o2) {
          const int i2 = col[o2].pidx;
          const double* Bj = &Hpl[18 * (size_t)col[o2].slot];
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 6; ++c)
              S[(size_t)(6 * i1 + a) * n + 6 * i2 + c] -=
                  BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1] + BD[a * 3 + 2] * Bj[c * 3 + 2];
        }
      }
    }
    // mirror the upper block triangle to the lower one (the linear solver reads the upper part)
    for (int r = 0; r < n; ++r)
      for (int c = r + 1; c < n; ++c)
        if (r / 6 != c / 6) S[(size_t)c * n + r] = S[(size_t)r * n + c];
    std::vector<double> bs(n);
    for (int i = 0; i < n; ++i) bs[i] = bp[i] - coeff[i];
    if (n > 0 && !chol_solve(S, n, bs.data(), x.data())) return false;
    // landmarks: xl = Dinv * (bl - Hpl^T xp)   (block_solver.hpp:422-442)
    for (int j = 0; j < nL; ++j) {
      double cl[3] = {bl[3 * (size_t)j], bl[3 * (size_t)j + 1], bl[3 * (size_t)j + 2]};
      for (const auto& pl : lm_cols[j]) {
        const double* B = &Hpl[18 * (size_t)pl.slot];
        const double* xp = &x[6 * (size_t)pl.pidx];
        for (int c = 0; c < 3; ++c)
          for (int a = 0; a < 6; ++a) cl[c] -= B[a * 3 + c] * xp[a];
      }
      const double* Di = &Dinv[9 * (size_t)j];
      double* xl = &x[6 * (size_t)nP + 3 * (size_t)j];
      for (int r = 0; r < 3; ++r) xl[r] = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
    }
    return true;
  }

  // SparseOptimizer::update (sparse_optimizer.cpp:433-446)
  void apply_update()
  {
    for (int i = 0; i < P; ++i) {
      if (pose_idx[i] < 0) continue;
      double out[7];
      pose_oplus(poses + 7 * i, &x[6 * (size_t)pose_idx[i]], out);
      std::memcpy(poses + 7 * i, out, sizeof(out));
    }
    for (int j = 0; j < L; ++j) {
      if (lm_idx[j] < 0) continue;
      const double* d = &x[6 * (size_t)nP + 3 * (size_t)lm_idx[j]];
      points[3 * j] += d[0]; points[3 * j + 1] += d[1]; points[3 * j + 2] += d[2];
    }
  }

  // computeScale (optimization_algorithm_levenberg.cpp:168-175)
  double compute_scale(double lambda) const
  {
    double s = 0;
    const size_t np = 6 * (size_t)nP;
    for (size_t j = 0; j < np; ++j) s += x[j] * (lambda * x[j] + bp[j]);
    for (size_t j = 0; j < 3 * (size_t)nL; ++j) s += x[np + j] * (lambda * x[np + j] + bl[j]);
    return s;
  }

  struct Stats { std::vector<double> chi2, lambda; std::vector<int> trials; };

  // SparseOptimizer::optimize (sparse_optimizer.cpp:366-431) driving
  // OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:58-150)
  int optimize(int iterations, Stats* st)
  {
    if (nP + nL == 0) return -1;
    double lambda = -1, ni = 2;
    int done = 0;
    std::vector<double> poses_bak(7 * (size_t)P), points_bak(3 * (size_t)L);
    for (int it = 0; it < iterations; ++it) {
      compute_errors();
      double currentChi = robust_chi2();
      double tempChi = currentChi;
      build_system();
      if (it == 0) { lambda = lambda_init(); ni = 2; }
      double rho = 0;
      int qmax = 0;
      const int maxTrials = 10;
      bool lambda_bad = false;
      do {
        std::memcpy(poses_bak.data(), poses, sizeof(double) * 7 * P);       // push
        std::memcpy(points_bak.data(), points, sizeof(double) * 3 * L);
        const bool ok2 = solve(lambda);
        apply_update();
        compute_errors();
        tempChi = robust_chi2();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = (currentChi - tempChi);
        double scale = compute_scale(lambda);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          const double scaleFactor = std::max(1. / 3., alpha);
          lambda *= scaleFactor;
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          std::memcpy(poses, poses_bak.data(), sizeof(double) * 7 * P);     // pop
          std::memcpy(points, points_bak.data(), sizeof(double) * 3 * L);
          if (!std::isfinite(lambda)) { lambda_bad = true; break; }
        }
        qmax++;
      } while (rho < 0 && qmax < maxTrials);
      ++done;
      if (st) {
        // NB: like g2o, `err` holds the errors of the LAST TRIAL evaluated (pop() restores vertices only,
        // never the edges' _error); when that trial was rejected this is not the chi2 of the kept state.
        st->chi2.push_back(robust_chi2());
        st->lambda.push_back(lambda);
        st->trials.push_back(qmax);
      }
      if (qmax == maxTrials || rho == 0 || lambda_bad) break;  // Terminate
    }
    return done;
  }
};

Cam cam_from(const double* K4) { return Cam{K4[0], K4[1], K4[2], K4[3]}; }

}  // namespace

extern "C" {

void orc_se3_exp(const double* tangent6, double* pose7) { se3_exp(tangent6, pose7); }
void orc_pose_oplus(const double* pose7, const double* delta6, double* out7) { pose_oplus(pose7, delta6, out7); }
void orc_se3_act(const double* pose7, const double* p3, double* out3) { se3_act(pose7, p3, out3); }

void orc_edge_eval(const double* pose7, const double* p3, const double* uv2, const double* K4,
                   const double* ext7, double huber_delta, int jac_mode, double* err2, double* Ji12,
                   double* Jj6, double* chi2, double* rho3)
{
  Cam K = cam_from(K4);
  edge_error(pose7, p3, ext7, K, uv2, err2);
  if (jac_mode == 1) edge_jac_numeric(pose7, p3, ext7, K, uv2, true, true, Ji12, Jj6);
  else edge_jac_analytic(pose7, p3, ext7, K, Ji12, Jj6);
  *chi2 = err2[0] * err2[0] + err2[1] * err2[1];
  huber(*chi2, huber_delta, rho3);
}

static void ba_setup(BA& ba, int P, double* poses, const uint8_t* pose_fixed, int L, double* points,
                     const uint8_t* point_fixed, int E, const int32_t* edge_pose, const int32_t* edge_point,
                     const double* edge_uv, const uint8_t* edge_cam, const double* K4, const double* cam_ext14,
                     double huber_delta, int jac_mode)
{
  ba.P = P; ba.L = L; ba.E = E; ba.poses = poses; ba.points = points;
  ba.pose_fixed = pose_fixed; ba.point_fixed = point_fixed;
  ba.edge_pose = edge_pose; ba.edge_point = edge_point; ba.edge_uv = edge_uv; ba.edge_cam = edge_cam;
  ba.K = cam_from(K4); ba.ext = cam_ext14; ba.huber_delta = huber_delta; ba.jac_mode = jac_mode;
  // SparseOptimizer::initializeOptimization (sparse_optimizer.cpp:237): edges whose vertices are ALL fixed
  // are not active -- they contribute neither to chi2 nor to the system.  (g2o never computes their _error,
  // so the reference would read uninitialised memory through edge->chi2(); we report their chi2 at the
  // input state instead, see orc_ba_solve.)
  ba.active.assign(E, 1);
  for (int e = 0; e < E; ++e)
    if (pose_fixed && pose_fixed[edge_pose[e]] && point_fixed && point_fixed[edge_point[e]]) ba.active[e] = 0;
}

// Backend::OptimizeActiveMap outer loop, src/ssvio/backend.cpp:175-203.
int orc_ba_solve(int P, double* poses, const uint8_t* pose_fixed, int L, double* points,
                 const uint8_t* point_fixed, int E, const int32_t* edge_pose, const int32_t* edge_point,
                 const double* edge_uv, const uint8_t* edge_cam, const double* K4, const double* cam_ext14,
                 const orc_ba_options* opt, double* edge_chi2_out, uint8_t* edge_outlier_out,
                 int stats_cap, int* stats_n, double* stats_chi2, double* stats_lambda, int* stats_trials)
{
  BA ba;
  ba_setup(ba, P, poses, pose_fixed, L, points, point_fixed, E, edge_pose, edge_point, edge_uv, edge_cam,
           K4, cam_ext14, opt->huber_delta, opt->jac_mode);
  BA::Stats st;
  // errors of the inactive (all-fixed) edges: constant, evaluated once
  std::vector<double> inactive_err(2 * (size_t)E, 0.0);
  for (int e = 0; e < E; ++e)
    if (!ba.active[e])
      edge_error(poses + 7 * edge_pose[e], points + 3 * edge_point[e], cam_ext14 + 7 * (edge_cam ? edge_cam[e] : 0),
                 ba.K, edge_uv + 2 * e, &inactive_err[2 * e]);
  int round = 0, rounds_done = 0;
  while (round < opt->outer_rounds) {
    ba.init_structure();          // initializeOptimization()
    for (int e = 0; e < E; ++e)
      if (!ba.active[e]) { ba.err[2 * e] = inactive_err[2 * e]; ba.err[2 * e + 1] = inactive_err[2 * e + 1]; }
    ba.optimize(opt->iters, &st);
    ++rounds_done;
    int cnt_outlier = 0, cnt_inlier = 0;
    for (int e = 0; e < E; ++e) {
      if (ba.edge_chi2(e) > opt->chi2_th) ++cnt_outlier; else ++cnt_inlier;
    }
    const double ratio = cnt_inlier / double(cnt_inlier + cnt_outlier);
    if (ratio > opt->inlier_ratio) break;
    ++round;
  }
  for (int e = 0; e < E; ++e) {
    if (edge_chi2_out) edge_chi2_out[e] = ba.edge_chi2(e);
    if (edge_outlier_out) edge_outlier_out[e] = ba.edge_chi2(e) > opt->chi2_th;
  }
  if (stats_n) {
    int n = std::min<int>((int)st.chi2.size(), stats_cap);
    *stats_n = n;
    for (int i = 0; i < n; ++i) {
      if (stats_chi2) stats_chi2[i] = st.chi2[i];
      if (stats_lambda) stats_lambda[i] = st.lambda[i];
      if (stats_trials) stats_trials[i] = st.trials[i];
    }
  }
  return rounds_done;
}

int orc_ba_linearize(int P, const double* poses, const uint8_t* pose_fixed, int L, const double* points,
                     const uint8_t* point_fixed, int E, const int32_t* edge_pose, const int32_t* edge_point,
                     const double* edge_uv, const uint8_t* edge_cam, const double* K4, const double* cam_ext14,
                     double huber_delta, int jac_mode, double* Hpp, double* bp, double* Hll, double* bl,
                     double* Hpl, double* edge_err2, double* chi2_robust)
{
  BA ba;
  ba_setup(ba, P, const_cast<double*>(poses), pose_fixed, L, const_cast<double*>(points), point_fixed, E,
           edge_pose, edge_point, edge_uv, edge_cam, K4, cam_ext14, huber_delta, jac_mode);
  ba.init_structure();
  ba.compute_errors();
  ba.build_system();
  if (chi2_robust) *chi2_robust = ba.robust_chi2();
  // scatter to full-size (fixed vertices get zeros)
  if (Hpp) std::memset(Hpp, 0, sizeof(double) * 36 * P);
  if (bp) std::memset(bp, 0, sizeof(double) * 6 * P);
  if (Hll) std::memset(Hll, 0, sizeof(double) * 9 * L);
  if (bl) std::memset(bl, 0, sizeof(double) * 3 * L);
  if (Hpl) std::memset(Hpl, 0, sizeof(double) * 18 * E);
  for (int i = 0; i < P; ++i) {
    const int pi = ba.pose_idx[i];
    if (pi < 0) continue;
    if (Hpp) std::memcpy(Hpp + 36 * (size_t)i, &ba.Hpp[36 * (size_t)pi], sizeof(double) * 36);
    if (bp) std::memcpy(bp + 6 * (size_t)i, &ba.bp[6 * (size_t)pi], sizeof(double) * 6);
  }
  for (int j = 0; j < L; ++j) {
    const int li = ba.lm_idx[j];
    if (li < 0) continue;
    if (Hll) std::memcpy(Hll + 9 * (size_t)j, &ba.Hll[9 * (size_t)li], sizeof(double) * 9);
    if (bl) std::memcpy(bl + 3 * (size_t)j, &ba.bl[3 * (size_t)li], sizeof(double) * 3);
  }
  if (Hpl)   // per-edge slot value (edges sharing a (pose,landmark) pair share the summed block)
    for (int e = 0; e < E; ++e)
      if (ba.edge_slot[e] >= 0) std::memcpy(Hpl + 18 * (size_t)e, &ba.Hpl[18 * (size_t)ba.edge_slot[e]], sizeof(double) * 18);
  if (edge_err2) std::memcpy(edge_err2, ba.err.data(), sizeof(double) * 2 * E);
  return 0;
}

// FrontEnd::EstimateCurrentPose core, src/ssvio/frontend.cpp:184-270, with EdgeProjectionPoseOnly
// (g2otypes.hpp:67-110: analytic Jacobian, Zinv = 1/(Z+1e-18)) and LinearSolverDense.
int orc_pose_only(double* pose7, const double* K4, int M, const double* xyz, const double* uv,
                  int rounds, int iters, double chi2_th, double huber_delta, uint8_t* inlier_out)
{
  const Cam K = cam_from(K4);
  std::vector<uint8_t> level(M, 0), is_outlier(M, 0);
  std::vector<double> err(2 * (size_t)M, 0.0);
  bool use_kernel = true;
  auto compute_error = [&](int i) {
    double pc[3];
    se3_act(pose7, xyz + 3 * i, pc);
    const double hx = K.fx * pc[0] + 0.0 * pc[1] + K.cx * pc[2];
    const double hy = 0.0 * pc[0] + K.fy * pc[1] + K.cy * pc[2];
    const double hz = 0.0 * pc[0] + 0.0 * pc[1] + 1.0 * pc[2];
    err[2 * i] = uv[2 * i] - hx / hz;
    err[2 * i + 1] = uv[2 * i + 1] - hy / hz;
  };
  auto chi2_of = [&](int i) { return err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]; };
  auto robust_chi2 = [&]() {
    double c = 0;
    for (int i = 0; i < M; ++i) {
      if (level[i]) continue;
      if (use_kernel) { double rho[3]; huber(chi2_of(i), huber_delta, rho); c += rho[0]; }
      else c += chi2_of(i);
    }
    return c;
  };
  int cnt_outliers = 0;
  for (int round = 0; round < rounds; ++round) {
    // initializeOptimization(0): only level-0 edges are active.  optimize(iters):
    int n_active = 0;
    for (int i = 0; i < M; ++i) n_active += !level[i];
    double lambda = -1, ni = 2;
    for (int it = 0; it < iters && n_active > 0; ++it) {
      for (int i = 0; i < M; ++i) if (!level[i]) compute_error(i);
      double currentChi = robust_chi2(), tempChi = currentChi;
      double H[36] = {0}, b[6] = {0};
      for (int i = 0; i < M; ++i) {
        if (level[i]) continue;
        double pc[3];
        se3_act(pose7, xyz + 3 * i, pc);
        const double X = pc[0], Y = pc[1], Z = pc[2];
        const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;
        const double J[12] = {-K.fx * Zinv, 0, K.fx * X * Zinv2, K.fx * X * Y * Zinv2, -K.fx - K.fx * X * X * Zinv2,
                              K.fx * Y * Zinv, 0, -K.fy * Zinv, K.fy * Y * Zinv2, K.fy + K.fy * Y * Y * Zinv2,
                              -K.fy * X * Y * Zinv2, -K.fy * X * Zinv};
        double w = 1.0;
        if (use_kernel) { double rho[3]; huber(chi2_of(i), huber_delta, rho); w = rho[1]; }
        for (int a = 0; a < 6; ++a) {
          b[a] -= w * (J[a] * err[2 * i] + J[6 + a] * err[2 * i + 1]);
          for (int c = 0; c < 6; ++c) H[a * 6 + c] += J[a] * w * J[c] + J[6 + a] * w * J[6 + c];
        }
      }
      if (it == 0) {
        double m = 0;
        for (int d = 0; d < 6; ++d) m = std::max(std::fabs(H[d * 7]), m);
        lambda = 1e-5 * m; ni = 2;
      }
      double rho = 0; int qmax = 0; bool lambda_bad = false;
      do {
        double bak[7]; std::memcpy(bak, pose7, sizeof(bak));
        std::vector<double> S(36);
        for (int k = 0; k < 36; ++k) S[k] = H[k] + ((k % 7 == 0) ? lambda : 0.0);
        double xs[6] = {0};
        const bool ok2 = chol_solve(S, 6, b, xs);
        double out[7]; pose_oplus(pose7, xs, out); std::memcpy(pose7, out, sizeof(out));
        for (int i = 0; i < M; ++i) if (!level[i]) compute_error(i);
        tempChi = robust_chi2();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < 6; ++j) scale += xs[j] * (lambda * xs[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2; currentChi = tempChi;
        } else {
          lambda *= ni; ni *= 2;
          std::memcpy(pose7, bak, sizeof(bak));
          if (!std::isfinite(lambda)) { lambda_bad = true; break; }
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0 || lambda_bad) break;
    }
    // NB: after a rejected last trial g2o leaves _error of the active edges at the TRIAL state
    // (computeActiveErrors ran after update(); pop() restores vertices only).  frontend.cpp:247-251
    // recomputes the error only for features flagged outlier; mirror that exactly:
    cnt_outliers = 0;
    for (int i = 0; i < M; ++i) {
      if (is_outlier[i]) compute_error(i);
      if (chi2_of(i) > chi2_th) { is_outlier[i] = 1; level[i] = 1; cnt_outliers++; }
      else { is_outlier[i] = 0; level[i] = 0; }
    }
    if (round == rounds - 2) use_kernel = false;
  }
  if (inlier_out) for (int i = 0; i < M; ++i) inlier_out[i] = !is_outlier[i];
  return M - cnt_outliers;
}

}  // extern "C"
