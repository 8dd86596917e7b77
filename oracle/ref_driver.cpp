// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin C-ABI driver around the REAL reference arithmetic: it #includes the reference's own
// headers where they lie under /root/reference (ssvio/g2otypes.hpp, ssvio/algorithm.hpp, the
// vendored g2o / Sophus / Eigen) and is linked against the reference's vendored g2o + CSparse
// sources compiled by oracle/Makefile into oracle/_ref/.  Nothing from /root/reference is copied
// into this repository; only this driver (ours) and the build recipe are committed.
//
// What it exposes (all flat arrays, pose = [qx qy qz qw tx ty tz], Sophus SE3d::data() order):
//   ref_ba_solve      graph assembly + outer robust loop of Backend::OptimizeActiveMap
//                     (/root/reference/src/ssvio/backend.cpp:78-203) on a flat problem.
//   ref_pose_only     FrontEnd::EstimateCurrentPose optimisation core
//                     (/root/reference/src/ssvio/frontend.cpp:184-270).
//   ref_triangulate   ssvio::triangulation (/root/reference/include/ssvio/algorithm.hpp:23-45) with the
//                     stereo rig of System::GenerateSteroCamera (src/ssvio/system.cpp:54-113) and the
//                     acceptance test of FrontEnd::BuidInitMap (src/ssvio/frontend.cpp:466).
//   ref_se3_exp / ref_pose_oplus   Sophus::SE3d::exp, VertexPose::oplusImpl (g2otypes.hpp:36-41).
//   ref_edge_eval     EdgeProjection error + g2o numeric Jacobian + Huber weights for ONE edge.
//   ref_pose_graph    LoopClosing::PoseGraphOptimization (/root/reference/src/ssvio/loopclosing.cpp:458-539):
//                     VertexPose + EdgePoseGraph, BlockSolver<6,6>, LinearSolverEigen, LM, optimize(20).
//   ref_bow_vector    the BowVector a TemplatedVocabulary::transform builds from per-feature (word, weight) pairs
//                     (/root/reference/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1083-1122), on the reference's OWN
//                     DBoW2::BowVector (BowVector.cpp:30-84 compiled where it lies: addWeight / addIfNotExist / normalize).
//                     TemplatedVocabulary.h itself includes OpenCV and cannot be compiled: the tree descent stays unpinned.
//
// backend.cpp / frontend.cpp themselves cannot be compiled here (they need OpenCV + Pangolin), so the
// graph assembly is restated from the cited lines; all arithmetic is the reference's.
#include <cstring>
#include <map>
#include <vector>

#include "DBoW2/BowVector.h"
#include "ssvio/algorithm.hpp"
#include "ssvio/g2otypes.hpp"
#include "g2o/solvers/eigen/linear_solver_eigen.h"

namespace {

using ssvio::EdgeProjection;
using ssvio::EdgeProjectionPoseOnly;
using ssvio::VertexPose;
using ssvio::VertexXYZ;

Sophus::SE3d pose_from(const double* p)
{
  Eigen::Quaterniond q(p[3], p[0], p[1], p[2]);  // (w,x,y,z)
  return Sophus::SE3d(q, Eigen::Vector3d(p[4], p[5], p[6]));
}

void pose_to(const Sophus::SE3d& T, double* p)
{
  const Eigen::Quaterniond& q = T.unit_quaternion();
  p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); p[3] = q.w();
  p[4] = T.translation()[0]; p[5] = T.translation()[1]; p[6] = T.translation()[2];
}

Eigen::Matrix3d K_from(const double* k)
{
  Eigen::Matrix3d K = Eigen::Matrix3d::Identity();
  K(0, 0) = k[0]; K(1, 1) = k[1]; K(0, 2) = k[2]; K(1, 2) = k[3];
  return K;
}

// Records what g2o exposes after each LM iteration, without side effects: robust chi2 of the edges'
// current _error, lambda, #trials (levenbergIteration()).
struct IterRecorder : public g2o::HyperGraphAction
{
  g2o::SparseOptimizer* opt = nullptr;
  g2o::OptimizationAlgorithmLevenberg* lm = nullptr;
  std::vector<double> chi2, lambda;
  std::vector<int> trials;
  HyperGraphAction* operator()(const g2o::HyperGraph*, Parameters* prm = 0) override
  {
    auto* pi = dynamic_cast<ParametersIteration*>(prm);
    if (pi && pi->iteration < 0) return this;  // the postIteration(-1) call of initializeOptimization()
    // no computeActiveErrors() here: it would overwrite the edges' _error and change what
    // Backend::OptimizeActiveMap later reads through edge->chi2() after a rejected final trial.
    chi2.push_back(opt->activeRobustChi2());
    lambda.push_back(lm->currentLambda());
    trials.push_back(lm->levenbergIteration());
    return this;
  }
};

}  // namespace

extern "C" {

// Returns the number of outer rounds executed.  stats_* are optional (may be NULL); stats_cap is the
// capacity in iterations; *stats_n receives the number of LM iterations recorded.
int ref_ba_solve(int P, double* poses, const unsigned char* pose_fixed, int L, double* points,
                 const unsigned char* point_fixed, int E, const int* edge_pose, const int* edge_point,
                 const double* edge_uv, const unsigned char* edge_cam, const double* K4,
                 const double* cam_ext14, int outer_rounds, int iters, double chi2_th, double huber_delta,
                 double inlier_ratio_th, double* edge_chi2_out, int stats_cap, int* stats_n,
                 double* stats_chi2, double* stats_lambda, int* stats_trials)
{
  typedef g2o::BlockSolver_6_3 BlockSolverType;
  typedef g2o::LinearSolverCSparse<BlockSolverType::PoseMatrixType> LinearSolverType;
  auto solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
  g2o::SparseOptimizer optimizer;
  optimizer.setAlgorithm(solver);

  IterRecorder rec;
  rec.opt = &optimizer;
  rec.lm = solver;
  if (stats_n) optimizer.addPostIterationAction(&rec);

  std::vector<VertexPose*> vp(P);
  for (int i = 0; i < P; ++i) {
    VertexPose* v = new VertexPose();
    v->setId(i);
    v->setEstimate(pose_from(poses + 7 * i));
    if (pose_fixed && pose_fixed[i]) v->setFixed(true);
    optimizer.addVertex(v);
    vp[i] = v;
  }
  Eigen::Matrix3d K = K_from(K4);
  Sophus::SE3d ext[2] = {pose_from(cam_ext14), pose_from(cam_ext14 + 7)};

  std::vector<VertexXYZ*> vl(L);
  for (int j = 0; j < L; ++j) {
    VertexXYZ* v = new VertexXYZ;
    v->setEstimate(Eigen::Vector3d(points[3 * j], points[3 * j + 1], points[3 * j + 2]));
    v->setId(P + j);
    v->setMarginalized(true);
    if (point_fixed && point_fixed[j]) v->setFixed(true);
    optimizer.addVertex(v);
    vl[j] = v;
  }
  std::vector<EdgeProjection*> edges(E);
  for (int e = 0; e < E; ++e) {
    EdgeProjection* edge = new EdgeProjection(K, ext[edge_cam ? edge_cam[e] : 0]);
    edge->setId(e + 1);
    edge->setVertex(0, vp[edge_pose[e]]);
    edge->setVertex(1, vl[edge_point[e]]);
    edge->setMeasurement(Eigen::Vector2d(edge_uv[2 * e], edge_uv[2 * e + 1]));
    edge->setInformation(Eigen::Matrix2d::Identity());
    auto rk = new g2o::RobustKernelHuber();
    rk->setDelta(huber_delta);
    edge->setRobustKernel(rk);
    optimizer.addEdge(edge);
    edges[e] = edge;
  }

  int round = 0, rounds_done = 0;
  while (round < outer_rounds) {
    optimizer.initializeOptimization();
    optimizer.optimize(iters);
    ++rounds_done;
    int cnt_outlier = 0, cnt_inlier = 0;
    for (int e = 0; e < E; ++e) {
      if (edges[e]->chi2() > chi2_th) ++cnt_outlier; else ++cnt_inlier;
    }
    double ratio = cnt_inlier / double(cnt_inlier + cnt_outlier);
    if (ratio > inlier_ratio_th) break;
    ++round;
  }

  for (int i = 0; i < P; ++i) pose_to(vp[i]->estimate(), poses + 7 * i);
  for (int j = 0; j < L; ++j) {
    const Eigen::Vector3d& p = vl[j]->estimate();
    points[3 * j] = p[0]; points[3 * j + 1] = p[1]; points[3 * j + 2] = p[2];
  }
  if (edge_chi2_out)
    for (int e = 0; e < E; ++e) edge_chi2_out[e] = edges[e]->chi2();
  if (stats_n) {
    int n = (int)rec.chi2.size();
    if (n > stats_cap) n = stats_cap;
    *stats_n = n;
    for (int i = 0; i < n; ++i) {
      if (stats_chi2) stats_chi2[i] = rec.chi2[i];
      if (stats_lambda) stats_lambda[i] = rec.lambda[i];
      if (stats_trials) stats_trials[i] = rec.trials[i];
    }
  }
  return rounds_done;
}

// Pose-only optimisation, frontend.cpp:184-270.  inlier_out[m] = 1 if the feature ends as inlier.
int ref_pose_only(double* pose7, const double* K4, int M, const double* xyz, const double* uv,
                  int rounds, int iters, double chi2_th, unsigned char* inlier_out)
{
  typedef g2o::BlockSolver_6_3 BlockSolverType;
  typedef g2o::LinearSolverDense<BlockSolverType::PoseMatrixType> LinearSolverType;
  auto solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
  g2o::SparseOptimizer optimizer;
  optimizer.setAlgorithm(solver);

  VertexPose* vertex_pose = new VertexPose();
  vertex_pose->setId(0);
  vertex_pose->setEstimate(pose_from(pose7));
  optimizer.addVertex(vertex_pose);
  Eigen::Matrix3d K = K_from(K4);

  std::vector<EdgeProjectionPoseOnly*> edges(M);
  std::vector<unsigned char> is_outlier(M, 0);
  for (int i = 0; i < M; ++i) {
    EdgeProjectionPoseOnly* edge = new EdgeProjectionPoseOnly(
        Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), K);
    edge->setId(i + 1);
    edge->setVertex(0, vertex_pose);
    edge->setMeasurement(Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]));
    edge->setInformation(Eigen::Matrix2d::Identity());
    edge->setRobustKernel(new g2o::RobustKernelHuber);
    edges[i] = edge;
    optimizer.addEdge(edge);
  }
  int cnt_outliers = 0;
  for (int iteration = 0; iteration < rounds; iteration++) {
    optimizer.initializeOptimization();
    optimizer.optimize(iters);
    cnt_outliers = 0;
    for (int i = 0; i < M; ++i) {
      auto e = edges[i];
      if (is_outlier[i]) e->computeError();
      if (e->chi2() > chi2_th) {
        is_outlier[i] = 1;
        e->setLevel(1);
        cnt_outliers++;
      } else {
        is_outlier[i] = 0;
        e->setLevel(0);
      }
      if (iteration == rounds - 2) e->setRobustKernel(nullptr);
    }
  }
  pose_to(vertex_pose->estimate(), pose7);
  if (inlier_out)
    for (int i = 0; i < M; ++i) inlier_out[i] = !is_outlier[i];
  return M - cnt_outliers;
}

// Stereo triangulation with the rig of system.cpp:63,71 (left = I, right = (I, (-baseline,0,0))).
// ok_out[i] = triangulation()==true && z>0 (frontend.cpp:466); xyz_out always holds the DLT point.
void ref_triangulate(int n, const double* uvL, const double* uvR, double fx, double fy, double cx,
                     double cy, double baseline, double* xyz_out, unsigned char* ok_out,
                     double* ratio_out)
{
  std::vector<Sophus::SE3d> poses{Sophus::SE3d(Sophus::SO3d(), Eigen::Vector3d::Zero()),
                                  Sophus::SE3d(Sophus::SO3d(), Eigen::Vector3d(-baseline, 0, 0))};
  for (int i = 0; i < n; ++i) {
    std::vector<Eigen::Vector3d> pts{
        Eigen::Vector3d((uvL[2 * i] - cx) / fx * 1.0, (uvL[2 * i + 1] - cy) / fy * 1.0, 1.0),
        Eigen::Vector3d((uvR[2 * i] - cx) / fx * 1.0, (uvR[2 * i + 1] - cy) / fy * 1.0, 1.0)};
    Eigen::Vector3d pw = Eigen::Vector3d::Zero();
    bool ok = ssvio::triangulation(poses, pts, pw);
    xyz_out[3 * i] = pw[0]; xyz_out[3 * i + 1] = pw[1]; xyz_out[3 * i + 2] = pw[2];
    ok_out[i] = (ok && pw[2] > 0) ? 1 : 0;
    if (ratio_out) {
      // recompute the ratio the reference tests (sigma3/sigma2) for fixture diagnostics
      Eigen::Matrix<double, 4, 4> A;
      for (int c = 0; c < 2; ++c) {
        Eigen::Matrix<double, 3, 4> m = poses[c].matrix3x4();
        A.row(2 * c) = pts[c][0] * m.row(2) - m.row(0);
        A.row(2 * c + 1) = pts[c][1] * m.row(2) - m.row(1);
      }
      Eigen::MatrixXd Ad = A;
      auto svd = Ad.bdcSvd(Eigen::ComputeThinU | Eigen::ComputeThinV);
      ratio_out[i] = svd.singularValues()[3] / svd.singularValues()[2];
    }
  }
}

void ref_se3_exp(const double* tangent6, double* pose7_out)
{
  Eigen::Matrix<double, 6, 1> a;
  for (int i = 0; i < 6; ++i) a[i] = tangent6[i];
  pose_to(Sophus::SE3d::exp(a), pose7_out);
}

void ref_pose_oplus(const double* pose7, const double* delta6, double* pose7_out)
{
  VertexPose v;
  v.setEstimate(pose_from(pose7));
  v.oplus(delta6);
  pose_to(v.estimate(), pose7_out);
}

void ref_se3_act(const double* pose7, const double* p3, double* out3)
{
  Eigen::Vector3d r = pose_from(pose7) * Eigen::Vector3d(p3[0], p3[1], p3[2]);
  out3[0] = r[0]; out3[1] = r[1]; out3[2] = r[2];
}

// One EdgeProjection: error (2), g2o numeric Jacobians Ji (2x6 row-major), Jj (2x3 row-major),
// chi2 and Huber rho[3].
void ref_edge_eval(const double* pose7, const double* p3, const double* uv2, const double* K4,
                   const double* ext7, double huber_delta, double* err2, double* Ji12, double* Jj6,
                   double* chi2, double* rho3)
{
  typedef g2o::BlockSolver_6_3 BlockSolverType;
  typedef g2o::LinearSolverCSparse<BlockSolverType::PoseMatrixType> LinearSolverType;
  auto solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
  g2o::SparseOptimizer optimizer;
  optimizer.setAlgorithm(solver);
  VertexPose* v0 = new VertexPose();
  v0->setId(0);
  v0->setEstimate(pose_from(pose7));
  optimizer.addVertex(v0);
  VertexXYZ* v1 = new VertexXYZ;
  v1->setId(1);
  v1->setEstimate(Eigen::Vector3d(p3[0], p3[1], p3[2]));
  v1->setMarginalized(true);
  optimizer.addVertex(v1);
  EdgeProjection* edge = new EdgeProjection(K_from(K4), pose_from(ext7));
  edge->setId(1);
  edge->setVertex(0, v0);
  edge->setVertex(1, v1);
  edge->setMeasurement(Eigen::Vector2d(uv2[0], uv2[1]));
  edge->setInformation(Eigen::Matrix2d::Identity());
  auto rk = new g2o::RobustKernelHuber();
  rk->setDelta(huber_delta);
  edge->setRobustKernel(rk);
  optimizer.addEdge(edge);
  optimizer.initializeOptimization();
  optimizer.computeActiveErrors();
  edge->linearizeOplus(optimizer.jacobianWorkspace());
  err2[0] = edge->error()[0];
  err2[1] = edge->error()[1];
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 6; ++c) Ji12[r * 6 + c] = edge->jacobianOplusXi()(r, c);
    for (int c = 0; c < 3; ++c) Jj6[r * 3 + c] = edge->jacobianOplusXj()(r, c);
  }
  *chi2 = edge->chi2();
  g2o::Vector3 rho;
  rk->robustify(*chi2, rho);
  rho3[0] = rho[0]; rho3[1] = rho[1]; rho3[2] = rho[2];
}


// Pose-graph optimisation, loopclosing.cpp:458-539.  poses[P][7] in/out, fixed[P], edges: vertex 0 = ei, vertex 1 = ej,
// measurement = relative pose (T_i * T_j^-1 at the time of creation), information = identity.
// edge_err_out (optional, 6 per edge) = the edges' _error after the optimisation.
int ref_pose_graph(int P, double* poses, const unsigned char* fixed, int E, const int* ei, const int* ej,
                   const double* meas7, int iterations, double* edge_err_out, int stats_cap, int* stats_n,
                   double* stats_chi2, double* stats_lambda, int* stats_trials)
{
  typedef g2o::BlockSolver<g2o::BlockSolverTraits<6, 6>> BlockSolverType;
  typedef g2o::LinearSolverEigen<BlockSolverType::PoseMatrixType> LinearSolverType;
  auto solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
  g2o::SparseOptimizer optimizer;
  optimizer.setAlgorithm(solver);
  std::vector<VertexPose*> vs(P);
  for (int i = 0; i < P; ++i) {
    VertexPose* v = new VertexPose();
    v->setId(i);
    v->setEstimate(pose_from(poses + 7 * i));
    v->setMarginalized(false);
    if (fixed[i]) v->setFixed(true);
    optimizer.addVertex(v);
    vs[i] = v;
  }
  std::vector<ssvio::EdgePoseGraph*> es(E);
  for (int k = 0; k < E; ++k) {
    ssvio::EdgePoseGraph* e = new ssvio::EdgePoseGraph();
    e->setId(k);
    e->setVertex(0, vs[ei[k]]);
    e->setVertex(1, vs[ej[k]]);
    e->setMeasurement(pose_from(meas7 + 7 * k));
    e->setInformation(Eigen::Matrix<double, 6, 6>::Identity());
    optimizer.addEdge(e);
    es[k] = e;
  }
  IterRecorder rec;
  rec.opt = &optimizer; rec.lm = solver;
  optimizer.addPostIterationAction(&rec);
  optimizer.initializeOptimization();
  const int done = optimizer.optimize(iterations);
  for (int i = 0; i < P; ++i) pose_to(vs[i]->estimate(), poses + 7 * i);
  if (edge_err_out)
    for (int k = 0; k < E; ++k)
      for (int r = 0; r < 6; ++r) edge_err_out[6 * k + r] = es[k]->error()[r];
  int n = 0;
  for (size_t k = 0; k < rec.chi2.size() && (int)k < stats_cap; ++k, ++n) {
    if (stats_chi2) stats_chi2[k] = rec.chi2[k];
    if (stats_lambda) stats_lambda[k] = rec.lambda[k];
    if (stats_trials) stats_trials[k] = rec.trials[k];
  }
  if (stats_n) *stats_n = n;
  optimizer.removePostIterationAction(&rec);
  return done;
}


// One EdgePoseGraph: error (g2otypes.hpp:169-176) and g2o's numeric Jacobians (base_binary_edge.hpp:61-141).
void ref_pg_edge_eval(const double* meas7, const double* T0, const double* T1, double* err6, double* Ji36, double* Jj36)
{
  VertexPose v0, v1;
  v0.setId(0); v1.setId(1);
  v0.setEstimate(pose_from(T0)); v1.setEstimate(pose_from(T1));
  ssvio::EdgePoseGraph e;
  e.setVertex(0, &v0); e.setVertex(1, &v1);
  e.setMeasurement(pose_from(meas7));
  e.setInformation(Eigen::Matrix<double, 6, 6>::Identity());
  e.computeError();
  for (int r = 0; r < 6; ++r) err6[r] = e.error()[r];
  g2o::JacobianWorkspace ws;
  ws.updateSize(&e);
  ws.allocate();
  e.linearizeOplus(ws);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      if (Ji36) Ji36[r * 6 + c] = e.jacobianOplusXi()(r, c);
      if (Jj36) Jj36[r * 6 + c] = e.jacobianOplusXj()(r, c);
    }
}

// TemplatedVocabulary::transform after the tree descent (TemplatedVocabulary.h:1083-1122): weighting 0 TF_IDF / 1 TF ->
// addWeight + division by the feature count unless the scoring normalises (L1_NORM: it does); 2 IDF / 3 BINARY ->
// addIfNotExist; then normalize(L1).  -> number of entries (ids ascending: std::map order)
int ref_bow_vector(int n, const int32_t* word, const double* weight, int weighting, int cap, int32_t* ids_out, double* vals_out)
{
  DBoW2::BowVector v;
  for (int i = 0; i < n; ++i) {
    if (!(weight[i] > 0)) continue;
    if (weighting == 0 || weighting == 1) v.addWeight((DBoW2::WordId)word[i], weight[i]);
    else v.addIfNotExist((DBoW2::WordId)word[i], weight[i]);
  }
  v.normalize(DBoW2::L1);
  int m = 0;
  for (const auto& kv : v) {
    if (m < cap) { ids_out[m] = (int32_t)kv.first; vals_out[m] = kv.second; }
    ++m;
  }
  return m;
}

}  // extern "C"
