// include/ssx_shim.hpp -- header-only C++ wrappers over the C ABI (ssx.h) with the method names and argument
// meaning of the reference classes they replace, for a maintainer who wants to swap ssvio's compute bodies
// without touching its callers.  No OpenCV / g2o / Sophus types: images are (pointer, stride, rows, cols),
// keypoints are ssx_keypoint (binary layout of cv::KeyPoint), poses are 7 doubles in Sophus::SE3d::data() order.
// INTEGRATION.md shows the two-line adapters from cv::Mat / std::vector<cv::KeyPoint> / Sophus::SE3d.
//
//   ssx::Context                 one GPU + one stream (create one per ssvio thread: front-end, backend)
//   ssx::ORBextractor            ssvio::ORBextractor (include/ssvio/orbextractor.hpp:44-59)
//   ssx::triangulation           ssvio::triangulation (include/ssvio/algorithm.hpp:23-25) for the stereo rig
//   ssx::BundleAdjuster          the optimisation of Backend::OptimizeActiveMap (src/ssvio/backend.cpp:78-245)
//   ssx::StereoFrontEnd          DetectFeatures + FindFeaturesInRight + triangulation in one device-resident call
//   ssx::calcOpticalFlowPyrLK    cv::calcOpticalFlowPyrLK as frontend.cpp:156-166 / :374-384 call it
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ssx.h"

namespace ssx {

class Context {
 public:
  explicit Context(int device = 0, void* hip_stream = nullptr)
  {
    if (ssx_abi_check(SSX_VERSION, sizeof(ssx_config), sizeof(ssx_ba_problem), sizeof(ssx_ba_options), sizeof(ssx_ba_result),
                      sizeof(ssx_ba_window_update)) != SSX_OK)
      throw std::runtime_error("libssx.so was built from another version of include/ssx.h than this caller");
    ssx_config cfg{};
    cfg.device = device; cfg.stream = hip_stream;
    const ssx_status st = ssx_ctx_create(&cfg, &ctx_);
    if (st != SSX_OK) throw std::runtime_error("ssx_ctx_create failed (no gfx950 device? there is no CPU fallback)");
  }
  ~Context() { ssx_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  ssx_ctx* get() const { return ctx_; }
  // the reference's compute code signals errors by assert / LOG(FATAL); the shim throws instead of aborting
  void check(ssx_status st) const
  {
    if (st != SSX_OK) throw std::runtime_error(std::string("ssx: ") + ssx_last_error(ctx_));
  }

 private:
  ssx_ctx* ctx_ = nullptr;
};

// ssvio::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
class ORBextractor {
 public:
  ORBextractor(Context& ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : ctx_(ctx), prm_{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST}
  {
  }

  // void Detect(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints)
  // (orbextractor.cpp:755-842).  Empty image: returns with `keypoints` untouched, like the reference.
  void Detect(const uint8_t* image, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
              std::vector<ssx_keypoint>& keypoints)
  {
    if (!image || rows <= 0 || cols <= 0) return;
    std::vector<ssx_keypoint> out(capacity());
    int32_t n = 0;
    ctx_.check(ssx_orb_detect(ctx_.get(), image, stride, rows, cols, mask, mask_stride, &prm_, (int32_t)out.size(), out.data(), &n));
    out.resize(n);
    keypoints.swap(out);
  }

  // void DetectAndCompute(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors)
  // (orbextractor.cpp:687-753); descriptors = N x 32 bytes, row-major (CV_8U N x 32)
  void DetectAndCompute(const uint8_t* image, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                        std::vector<ssx_keypoint>& keypoints, std::vector<uint8_t>& descriptors)
  {
    if (!image || rows <= 0 || cols <= 0) return;
    std::vector<ssx_keypoint> out(capacity());
    std::vector<uint8_t> desc(out.size() * 32);
    int32_t n = 0;
    ctx_.check(ssx_orb_extract(ctx_.get(), image, stride, rows, cols, mask, mask_stride, &prm_, (int32_t)out.size(), out.data(),
                               desc.data(), &n));
    out.resize(n);
    desc.resize((size_t)n * 32);
    keypoints.swap(out);
    descriptors.swap(desc);
  }

  int GetLevels() const { return prm_.nlevels; }            // orbextractor.hpp:73
  float GetScaleFactor() const { return prm_.scale_factor; }  // orbextractor.hpp:75
  const ssx_orb_params& params() const { return prm_; }

 private:
  // a level can return up to max(budget + 3, 4 * nIni <= 256) keypoints (the first quadtree subdivision)
  size_t capacity() const { return (size_t)prm_.nfeatures + 260 * (size_t)prm_.nlevels + 64; }
  Context& ctx_;
  ssx_orb_params prm_;
};

// bool triangulation(const std::vector<SE3d>& poses, const std::vector<Vector3d> points, Vector3d& pt_world)
// for poses = {left = identity, right = (I, (-baseline,0,0))} (system.cpp:63,71) and pixel inputs; `ok` already
// includes the z > 0 test the callers add (frontend.cpp:466,528).  T_wc (nullable) = current pose inverse.
inline void triangulation(Context& ctx, const ssx_stereo_rig& rig, const std::vector<double>& uvL, const std::vector<double>& uvR,
                          const double* T_wc, std::vector<double>& xyz, std::vector<uint8_t>& ok)
{
  const int32_t n = (int32_t)(uvL.size() / 2);
  xyz.resize((size_t)n * 3);
  ok.resize(n);
  ctx.check(ssx_triangulate(ctx.get(), n, uvL.data(), uvR.data(), &rig, T_wc, xyz.data(), ok.data()));
}

// The optimisation of Backend::OptimizeActiveMap: the caller marshals its active keyframes / map points into flat
// arrays exactly where backend.cpp:88-169 creates vertices and edges, calls Optimize(), and writes the results back
// where backend.cpp:207-244 does.
class BundleAdjuster {
 public:
  explicit BundleAdjuster(Context& ctx) : ctx_(ctx) { ssx_ba_default_options(&opt); }
  ssx_ba_options opt;   // outer_rounds 5, iters 10, chi2 5.891, Huber 5.891, inlier ratio 0.7 (backend.cpp:109,163,175-195)

  // poses / points are updated in place; edge_outlier[e] = 1 where backend.cpp:209 would unlink the observation
  void Optimize(std::vector<double>& poses7, const std::vector<uint8_t>& pose_fixed, std::vector<double>& points3,
                const std::vector<uint8_t>& point_fixed, const std::vector<int32_t>& edge_pose,
                const std::vector<int32_t>& edge_point, const std::vector<double>& edge_uv, const std::vector<uint8_t>& edge_cam,
                const double K[4], const double cam_ext[14], std::vector<uint8_t>& edge_outlier, ssx_ba_result* stats = nullptr)
  {
    ssx_ba_problem p{};
    p.P = (int32_t)(poses7.size() / 7); p.poses = poses7.data(); p.pose_fixed = pose_fixed.empty() ? nullptr : pose_fixed.data();
    p.L = (int32_t)(points3.size() / 3); p.points = points3.data(); p.point_fixed = point_fixed.empty() ? nullptr : point_fixed.data();
    p.E = (int32_t)edge_pose.size(); p.edge_pose = edge_pose.data(); p.edge_point = edge_point.data(); p.edge_uv = edge_uv.data();
    p.edge_cam = edge_cam.empty() ? nullptr : edge_cam.data();
    for (int i = 0; i < 4; ++i) p.K[i] = K[i];
    for (int i = 0; i < 14; ++i) p.cam_ext[i] = cam_ext[i];
    std::vector<double> po(poses7.size()), pt(points3.size());
    edge_outlier.assign(edge_pose.size(), 0);
    ssx_ba_result local{};
    ssx_ba_result* r = stats ? stats : &local;
    r->poses_out = po.data(); r->points_out = pt.data(); r->edge_chi2 = nullptr; r->edge_outlier = edge_outlier.data();
    ctx_.check(ssx_ba_solve(ctx_.get(), &p, &opt, r));
    poses7.swap(po);
    points3.swap(pt);
  }

 private:
  Context& ctx_;
};

// DetectFeatures + FindFeaturesInRight + BuidInitMap/TriangulateNewPoints of the north_star pipeline in one call
class StereoFrontEnd {
 public:
  StereoFrontEnd(Context& ctx, const ssx_orb_params& orb, const ssx_stereo_rig& rig) : ctx_(ctx), orb_(orb), rig_(rig)
  {
    ssx_match_default_params(&mp);
    mp.scale_factor = orb.scale_factor;
  }
  ssx_match_params mp;

  struct Result {
    std::vector<ssx_keypoint> kpsL, kpsR;
    std::vector<uint8_t> descL, descR, ok;
    std::vector<int32_t> match_idx, match_dist;
    std::vector<double> xyz;
    int n_matched = 0, n_triangulated = 0;
  };

  Result Process(const uint8_t* imgL, const uint8_t* imgR, int stride, int rows, int cols, const double* T_wc = nullptr)
  {
    const size_t cap = (size_t)orb_.nfeatures + 260 * (size_t)orb_.nlevels + 64;   // see ORBextractor::capacity()
    Result r;
    r.kpsL.resize(cap); r.kpsR.resize(cap); r.descL.resize(cap * 32); r.descR.resize(cap * 32);
    r.match_idx.resize(cap); r.match_dist.resize(cap); r.xyz.resize(cap * 3); r.ok.resize(cap);
    ssx_stereo_frame_out o{};
    o.cap = (int32_t)cap; o.kpsL = r.kpsL.data(); o.kpsR = r.kpsR.data(); o.descL = r.descL.data(); o.descR = r.descR.data();
    o.match_idx = r.match_idx.data(); o.match_dist = r.match_dist.data(); o.xyz = r.xyz.data(); o.ok = r.ok.data();
    ctx_.check(ssx_stereo_frame(ctx_.get(), imgL, imgR, stride, rows, cols, &orb_, &mp, &rig_, T_wc, &o));
    r.kpsL.resize(o.nL); r.descL.resize((size_t)o.nL * 32); r.kpsR.resize(o.nR); r.descR.resize((size_t)o.nR * 32);
    r.match_idx.resize(o.nL); r.match_dist.resize(o.nL); r.xyz.resize((size_t)o.nL * 3); r.ok.resize(o.nL);
    r.n_matched = o.n_matched; r.n_triangulated = o.n_triangulated;
    return r;
  }

 private:
  Context& ctx_;
  ssx_orb_params orb_;
  ssx_stereo_rig rig_;
};

// cv::calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, status, err, Size(win, win), maxLevel,
//                          TermCriteria(COUNT+EPS, maxCount, epsilon), OPTFLOW_USE_INITIAL_FLOW)
// with cv::Point2f passed as interleaved floats (same layout).  nextPts must hold the initial guesses (as both call
// sites of the reference prepare them); status / err are resized.
inline void calcOpticalFlowPyrLK(Context& ctx, const uint8_t* prevImg, int prevStep, const uint8_t* nextImg, int nextStep,
                                 int rows, int cols, const std::vector<float>& prevPts, std::vector<float>& nextPts,
                                 std::vector<uint8_t>& status, std::vector<float>& err, int win = 11, int maxLevel = 3,
                                 int maxCount = 30, double epsilon = 0.01, bool useInitialFlow = true)
{
  if (prevPts.size() != nextPts.size() || (prevPts.size() & 1)) throw std::invalid_argument("calcOpticalFlowPyrLK: point vectors");
  const int n = (int)(prevPts.size() / 2);
  status.assign(n, 0); err.assign(n, 0.f);
  ssx_lk_params p;
  ssx_lk_default_params(&p);
  p.win = win; p.max_level = maxLevel; p.max_iters = maxCount; p.eps = epsilon; p.use_initial_flow = useInitialFlow ? 1 : 0;
  ctx.check(ssx_lk_track(ctx.get(), prevImg, prevStep, nextImg, nextStep, rows, cols, n, prevPts.data(), nextPts.data(),
                         status.data(), err.data(), &p, nullptr));
}

// The same call for consecutive frames (TrackLastFrame): prevImg is the nextImg of the last call on ctx and is not
// passed again -- its pyramid is still on the device (ssx_lk_track_next).
inline void calcOpticalFlowPyrLKNext(Context& ctx, const uint8_t* nextImg, int nextStep, int rows, int cols,
                                     const std::vector<float>& prevPts, std::vector<float>& nextPts,
                                     std::vector<uint8_t>& status, std::vector<float>& err, int win = 11, int maxLevel = 3,
                                     int maxCount = 30, double epsilon = 0.01, bool useInitialFlow = true)
{
  if (prevPts.size() != nextPts.size() || (prevPts.size() & 1)) throw std::invalid_argument("calcOpticalFlowPyrLK: point vectors");
  const int n = (int)(prevPts.size() / 2);
  status.assign(n, 0); err.assign(n, 0.f);
  ssx_lk_params p;
  ssx_lk_default_params(&p);
  p.win = win; p.max_level = maxLevel; p.max_iters = maxCount; p.eps = epsilon; p.use_initial_flow = useInitialFlow ? 1 : 0;
  ctx.check(ssx_lk_track_next(ctx.get(), nextImg, nextStep, rows, cols, n, prevPts.data(), nextPts.data(), status.data(),
                              err.data(), &p, nullptr));
}

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ssvio/orbvocabulary.hpp:10) as LoopClosing
// uses it: loadFromTextFile, transform(descriptors, BowVector), score(BowVector, BowVector).  A BowVector here is the
// sorted (word id, value) pairs of DBoW2::BowVector (a std::map<WordId, WordValue>).
struct BowVector {
  std::vector<int32_t> ids;
  std::vector<double> values;
  bool empty() const { return ids.empty(); }
};

class ORBVocabulary {
 public:
  explicit ORBVocabulary(Context& ctx) : ctx_(ctx) {}
  ~ORBVocabulary() { ssx_voc_destroy(voc_); }
  ORBVocabulary(const ORBVocabulary&) = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;

  bool loadFromTextFile(const std::string& filename)                 // loopclosing.cpp:33-41
  {
    ssx_voc_destroy(voc_);
    voc_ = nullptr;
    return ssx_voc_load_text(ctx_.get(), filename.c_str(), &voc_) == SSX_OK;
  }
  // descriptors: n x 32 bytes (the rows of the CV_8U n x 32 descriptor matrix)
  void transform(const uint8_t* descriptors, int n, BowVector& v) const
  {
    v.ids.assign((size_t)std::max(n, 1), 0);
    v.values.assign((size_t)std::max(n, 1), 0.0);
    int32_t m = 0;
    if (voc_) ctx_.check(ssx_voc_transform(voc_, descriptors, n, nullptr, nullptr, (int32_t)v.ids.size(), v.ids.data(), v.values.data(), &m));
    v.ids.resize(m);
    v.values.resize(m);
  }
  double score(const BowVector& a, const BowVector& b) const
  {
    return ssx_bow_score_l1((int32_t)a.ids.size(), a.ids.data(), a.values.data(), (int32_t)b.ids.size(), b.ids.data(), b.values.data());
  }

 private:
  Context& ctx_;
  ssx_vocabulary* voc_ = nullptr;
};

}  // namespace ssx
