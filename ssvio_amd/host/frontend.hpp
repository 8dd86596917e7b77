// ssvio_amd/host/frontend.hpp -- the tracking state machine around the compute calls:
//   Camera     /root/reference/src/ssvio/camera.cpp:9-50, include/ssvio/camera.hpp
//   FrontEnd   /root/reference/src/ssvio/frontend.cpp:34-128 (GrabSteroImage / Track), :130-182 TrackLastFrame,
//              :184-300 EstimateCurrentPose, :302-344 DetectFeatures, :346-428 FindFeaturesInRight, :430-446 SteroInit,
//              :448-498 BuidInitMap, :500-544 TriangulateNewPoints, :546-580 InsertKeyFrame
// Every arithmetic step of those functions is a Compute call; what is restated here is the bookkeeping between them.
// Not carried over: the viewer hooks (Pangolin), cv::imshow debugging, image undistortion (KITTI is rectified;
// Camera.NeedUndistortion != 0 is refused), and the mutexes -- the headless runner is single-threaded.
#pragma once
#include <memory>
#include <vector>

#include "compute.hpp"
#include "map.hpp"
#include "setting.hpp"

namespace ssx::host {

struct Camera {
  double fx = 0, fy = 0, cx = 0, cy = 0, baseline = 0;
  SE3 pose;                                          // extrinsic: camera <- rig
  void world2pixel(const double* p_w, const SE3& T_c_w, float* uv) const;   // cv::Point2f(p.x(), p.y()) of Camera::world2pixel
};

class Backend;

enum class FrontendStatus { INITING, TRACKING_GOOD, TRACKING_BAD, LOST };

struct StageTimes {                                   // wall-clock seconds spent inside the compute calls
  double detect = 0, lk_temporal = 0, lk_stereo = 0, pose_only = 0, triangulate = 0, bundle_adjust = 0;
  long n_detect = 0, n_lk_temporal = 0, n_lk_stereo = 0, n_pose_only = 0, n_triangulate = 0, n_bundle_adjust = 0;
};

class FrontEnd {
 public:
  FrontEnd(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right);
  void SetBackend(Backend* backend) { backend_ = backend; }

  bool GrabSteroImage(ImagePtr left, ImagePtr right, double timestamp);

  FrontendStatus status() const { return track_status_; }
  const FramePtr& current_frame() const { return current_frame_; }
  const KeyFramePtr& reference_kf() const { return reference_kf_; }
  StageTimes& times() { return times_; }

 private:
  bool SteroInit();
  bool BuidInitMap();
  bool Track();
  int TrackLastFrame();
  int EstimateCurrentPose();
  int DetectFeatures();
  int FindFeaturesInRight();
  int TriangulateNewPoints();
  bool InsertKeyFrame();

  Compute& compute_;
  std::shared_ptr<Map> map_;
  Backend* backend_ = nullptr;
  Camera left_camera_, right_camera_;
  ssx_orb_params orb_, orb_init_;
  ssx_stereo_rig rig_;

  FrontendStatus track_status_ = FrontendStatus::INITING;
  FramePtr current_frame_, last_frame_;
  KeyFramePtr reference_kf_;
  SE3 relative_motion_;

  int num_features_init_good_, num_features_tracking_good_, num_features_tracking_bad_;
  unsigned min_init_landmark_;
  bool open_backend_optimization_;
  std::vector<int32_t> boxes_;               // DetectFeatures: the mask's rectangles (x0, y0, x1, y1)
  StageTimes times_;
};

}  // namespace ssx::host
