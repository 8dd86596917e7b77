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

// The active window of the backend RESIDENT on the device (ssx_ba_window, include/ssx.h): the backend mirrors every change of
// its map -- a keyframe in, a keyframe out, outlier observations unlinked, condemned map points deleted
// (map.cpp:18-58, 89-194; backend.cpp:205-244) -- instead of re-marshalling the whole window at every keyframe.  Keyframes and
// map points are named by their ids; the window solves in id order, so its result is, bit for bit, the one BundleAdjust returns
// for the re-marshalled map (Backend::Marshal).
class BaWindow {
 public:
  virtual ~BaWindow() = default;
  virtual void Push(int64_t kf_id, const double* pose7, int n_new, const int64_t* new_ids, const double* new_xyz, const uint8_t* new_fixed,
                    int n_obs, const int64_t* obs_lm, const double* obs_uv, const uint8_t* obs_cam) = 0;
  virtual void Pop(int64_t kf_id) = 0;
  virtual void RemoveLandmarks(int n, const int64_t* lm_ids) = 0;
  virtual void RemoveFlagged(int n_obs, const uint8_t* flags) = 0;             // flags in the order of Export / of the last Solve
  virtual void Size(int& n_keyframes, int& n_landmarks, int& n_observations) = 0;
  // ids ascending; edge_pose / edge_point index into them; point_fixed = the flags the next solve uses (any may be null)
  virtual void Export(int64_t* kf_ids, int64_t* lm_ids, uint8_t* point_fixed, int32_t* edge_pose, int32_t* edge_point, double* edge_uv) = 0;
  virtual void Solve(ssx_ba_result& res) = 0;                                   // res.* in the order of Export
};

class Compute {
 public:
  virtual ~Compute() = default;
  virtual void Detect(const Image& img, const uint8_t* mask, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) = 0;
  // FrontEnd::DetectFeatures' mask (frontend.cpp:302-312) as its rectangles: boxes = n x (x0, y0, x1, y1), corners inclusive and inside
  // the image.  Default: rasterise on the host and call Detect (the CPU oracle's Compute); the GPU library takes the rectangles
  // themselves (ssx_orb_detect_boxes: 16 bytes per tracked feature over PCIe instead of a 466 KB mask).
  virtual void DetectBoxes(const Image& img, const std::vector<int32_t>& boxes, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps)
  {
    std::vector<uint8_t> mask((size_t)img.rows * img.cols, 255);
    for (size_t b = 0; b + 3 < boxes.size(); b += 4)
      for (long y = boxes[b + 1]; y <= boxes[b + 3]; ++y)
        for (long x = boxes[b]; x <= boxes[b + 2]; ++x) mask[(size_t)y * img.cols + x] = 0;
    Detect(img, mask.data(), prm, kps);
  }
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
  // a resident window with the first-observer rule of backend.cpp:125-130 switched on, or null when the implementation has
  // none (the backend then re-marshals its map per keyframe)
  virtual std::unique_ptr<BaWindow> MakeBaWindow(const double* K4, const double* cam_ext14, const ssx_ba_options& opt)
  {
    (void)K4; (void)cam_ext14; (void)opt;
    return nullptr;
  }
};

// device: GPU ordinal.  Three contexts (streams): per-frame work, the temporal LK chain, the backend.
std::unique_ptr<Compute> MakeSsxCompute(int device = 0);

}  // namespace ssx::host
