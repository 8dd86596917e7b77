// ssvio_amd/host/compute.hpp -- the five compute calls the reference's front-end and backend make, as one interface
// the host layer is written against:
//   Detect          ORBextractor::Detect              (frontend.cpp:302-344 DetectFeatures)
//   TrackLK         cv::calcOpticalFlowPyrLK          (frontend.cpp:156-166 TrackLastFrame, :374-384 FindFeaturesInRight)
//   PoseOnly        the g2o part of EstimateCurrentPose (frontend.cpp:196-270)
//   Triangulate     ssvio::triangulation + z > 0      (frontend.cpp:448-544)
//   BundleAdjust    the g2o part of OptimizeActiveMap (backend.cpp:78-205)
// SsxCompute is the product implementation: the C ABI of libssx.so on an MI355X, nothing else (no CPU fallback; the
// constructor throws without a gfx950 device).  The interface exists so tests can run the same host logic against
// the CPU oracle and compare whole trajectories.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/ssx.h"
#include "dataset.hpp"

namespace ssx::host {

class Compute {
 public:
  virtual ~Compute() = default;
  virtual void Detect(const Image& img, const uint8_t* mask, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) = 0;
  // 11x11 window, 3 levels, (COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW: next_pts holds the guesses going in.
  // `temporal` marks the frame-to-frame call (the implementation may keep the previous frame's pyramid).
  virtual void TrackLK(const Image& prev, const Image& next, const std::vector<float>& prev_pts, std::vector<float>& next_pts,
                       std::vector<uint8_t>& status, bool temporal) = 0;
  // 4 rounds x 10 iterations, chi2 5.991, Huber 1.0; returns features.size() - outliers
  virtual int PoseOnly(double* pose_io, const double* K4, int M, const double* xyz, const double* uv, uint8_t* inlier) = 0;
  virtual void Triangulate(int n, const double* uvL, const double* uvR, const ssx_stereo_rig& rig, const double* T_wc, double* xyz,
                           uint8_t* ok) = 0;
  // res.poses_out / points_out / edge_outlier point at caller storage
  virtual void BundleAdjust(const ssx_ba_problem& prob, const ssx_ba_options& opt, ssx_ba_result& res) = 0;
};

// device: GPU ordinal.  Three contexts (streams): per-frame work, the temporal LK chain, the backend.
std::unique_ptr<Compute> MakeSsxCompute(int device = 0);

}  // namespace ssx::host
