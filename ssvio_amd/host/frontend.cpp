// ssvio_amd/host/frontend.cpp -- see frontend.hpp
#include "frontend.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "backend.hpp"

namespace ssx::host {
namespace {

struct Stopwatch {
  double& acc; long& count;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  Stopwatch(double& a, long& c) : acc(a), count(c) {}
  ~Stopwatch() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++count; }
};

}  // namespace

void Camera::world2pixel(const double* p_w, const SE3& T_c_w, float* uv) const
{
  double pc[3], pr[3];
  T_c_w.act(p_w, pc);
  pose.act(pc, pr);                                                    // pose_ * T_c_w * p_w (camera.cpp:11)
  uv[0] = (float)(fx * pr[0] / pr[2] + cx);
  uv[1] = (float)(fy * pr[1] / pr[2] + cy);
}

FrontEnd::FrontEnd(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right)
    : compute_(compute), map_(std::move(map)), left_camera_(left), right_camera_(right)
{
  num_features_init_good_ = cfg.Get<int>("numFeatures.initGood");
  num_features_tracking_good_ = cfg.Get<int>("numFeatures.trackingGood");
  num_features_tracking_bad_ = cfg.Get<int>("numFeatures.trackingBad");
  min_init_landmark_ = (unsigned)cfg.Get<int>("Min.Init.Landmark.Num");
  open_backend_optimization_ = cfg.Get<int>("Backend.Open") != 0;
  if (cfg.Get<int>("Camera.NeedUndistortion") != 0)
    throw std::runtime_error("FrontEnd: Camera.NeedUndistortion != 0 is not supported (rectified input only)");
  if (left.fx != right.fx || left.fy != right.fy || left.cx != right.cx || left.cy != right.cy)
    throw std::runtime_error("FrontEnd: the two cameras must share their intrinsics (rectified stereo rig)");
  // System::GenerateORBextractor (system.cpp:115-128): nNewFeatures for keyframes, nInitFeatures for initialisation
  orb_.nfeatures = cfg.Get<int>("ORBextractor.nNewFeatures");
  orb_.scale_factor = cfg.Get<float>("ORBextractor.scaleFactor");
  orb_.nlevels = cfg.Get<int>("ORBextractor.nLevels");
  orb_.ini_th_fast = cfg.Get<int>("ORBextractor.iniThFAST");
  orb_.min_th_fast = cfg.Get<int>("ORBextractor.minThFAST");
  orb_init_ = orb_;
  orb_init_.nfeatures = cfg.Get<int>("ORBextractor.nInitFeatures");
  rig_ = ssx_stereo_rig{left.fx, left.fy, left.cx, left.cy, right.baseline};
}

// frontend.cpp:34-80
bool FrontEnd::GrabSteroImage(ImagePtr left, ImagePtr right, double timestamp)
{
  std::lock_guard<std::mutex> map_lock(map_->update_mutex);           // frontend.cpp:49: the whole frame is tracked under the map mutex
  current_frame_ = map_->NewFrame(std::move(left), std::move(right), timestamp);
  switch (track_status_) {
    case FrontendStatus::INITING: SteroInit(); break;
    case FrontendStatus::TRACKING_BAD:
    case FrontendStatus::TRACKING_GOOD: Track(); break;
    case FrontendStatus::LOST: break;                                  // the reference has no relocalisation either
  }
  last_frame_ = current_frame_;
  return true;
}

// frontend.cpp:82-128
bool FrontEnd::Track()
{
  if (last_frame_) current_frame_->relative_pose_to_kf = relative_motion_ * last_frame_->relative_pose_to_kf;   // constant velocity
  TrackLastFrame();
  const int inliers = EstimateCurrentPose();
  if (inliers > num_features_tracking_good_) {
    track_status_ = FrontendStatus::TRACKING_GOOD;
  } else if (inliers > num_features_tracking_bad_) {
    track_status_ = FrontendStatus::TRACKING_BAD;
  } else {
    track_status_ = FrontendStatus::LOST;
    std::fprintf(stderr, "[frontend] frame %lu: tracking lost with %d inliers\n", current_frame_->frame_id, inliers);
  }
  relative_motion_ = current_frame_->relative_pose_to_kf * last_frame_->relative_pose_to_kf.inverse();
  if (track_status_ == FrontendStatus::TRACKING_BAD) {                 // new features, new map points, new keyframe
    DetectFeatures();
    FindFeaturesInRight();
    TriangulateNewPoints();
    InsertKeyFrame();
  }
  return true;
}

// frontend.cpp:130-182
int FrontEnd::TrackLastFrame()
{
  const auto& last = last_frame_->features_left;
  const size_t n = last.size();
  if (n == 0) return 0;
  const SE3 T_cw = current_frame_->relative_pose_to_kf * reference_kf_->pose;
  std::vector<float> kps_last(2 * n), kps_current(2 * n);
  std::vector<char> has_point(n, 0);
  for (size_t i = 0; i < n; ++i) {
    kps_last[2 * i] = last[i]->x; kps_last[2 * i + 1] = last[i]->y;
    if (MapPointPtr mp = map_->Lock(last[i])) {
      has_point[i] = 1;
      left_camera_.world2pixel(mp->position, T_cw, &kps_current[2 * i]);
    } else {
      kps_current[2 * i] = last[i]->x; kps_current[2 * i + 1] = last[i]->y;
    }
  }
  std::vector<uint8_t> status;
  {
    Stopwatch sw(times_.lk_temporal, times_.n_lk_temporal);
    compute_.TrackLK(*last_frame_->left_image, *current_frame_->left_image, kps_last, kps_current, status, true);
  }
  int num_good_track = 0;
  for (size_t i = 0; i < n; ++i)
    if (status[i] && has_point[i]) {                                   // only triangulated points carry a BA constraint
      auto f = std::make_shared<Feature>();
      f->x = kps_current[2 * i]; f->y = kps_current[2 * i + 1];
      f->map_point = last[i]->map_point;
      current_frame_->features_left.push_back(std::move(f));
      ++num_good_track;
    }
  return num_good_track;
}

// frontend.cpp:184-300
int FrontEnd::EstimateCurrentPose()
{
  const double K4[4] = {left_camera_.fx, left_camera_.fy, left_camera_.cx, left_camera_.cy};
  SE3 est = current_frame_->relative_pose_to_kf * reference_kf_->pose;
  std::vector<FeaturePtr> features;
  std::vector<double> xyz, uv;
  for (auto& feat : current_frame_->features_left) {
    MapPointPtr mp = map_->Lock(feat);
    if (mp && !mp->is_outlier) {
      features.push_back(feat);
      xyz.insert(xyz.end(), mp->position, mp->position + 3);
      uv.push_back(feat->x); uv.push_back(feat->y);
    }
  }
  const int M = (int)features.size();
  std::vector<uint8_t> inlier(M, 0);
  int n_inliers = 0;
  if (M > 0) {
    Stopwatch sw(times_.pose_only, times_.n_pose_only);
    n_inliers = compute_.PoseOnly(est.data(), K4, M, xyz.data(), uv.data(), inlier.data());
  }
  current_frame_->pose = est;
  current_frame_->relative_pose_to_kf = est * reference_kf_->pose.inverse();
  for (int i = 0; i < M; ++i)
    if (!inlier[i]) {
      // an outlier seen within two frames of the reference keyframe condemns the map point itself (:283-288)
      MapPointPtr mp = map_->Lock(features[i]);
      if (mp && current_frame_->frame_id - reference_kf_->frame_id <= 2) {
        mp->is_outlier = true;
        map_->AddOutlierMapPoint(mp->id);
      }
      features[i]->map_point = kNoMapPoint;
      features[i]->is_outlier = false;
    }
  return n_inliers;
}

// frontend.cpp:302-344.  The mask boxes are cv::rectangle(pt - (10,10), pt + (10,10), 0, FILLED): corners rounded to
// the nearest integer (ties to even), both inclusive, clipped to the image.
int FrontEnd::DetectFeatures()
{
  const Image& img = *current_frame_->left_image;
  // cv::Mat mask(size, CV_8UC1, 255); cv::rectangle(mask, pt - (10, 10), pt + (10, 10), 0, FILLED) per tracked feature
  // (frontend.cpp:304-311): the rectangles themselves go to the compute layer, corners inclusive as cv::rectangle draws them
  boxes_.clear();
  for (const auto& feat : current_frame_->features_left) {
    const long x0 = std::max(0l, std::lrintf(feat->x - 10.f)), x1 = std::min((long)img.cols - 1, std::lrintf(feat->x + 10.f));
    const long y0 = std::max(0l, std::lrintf(feat->y - 10.f)), y1 = std::min((long)img.rows - 1, std::lrintf(feat->y + 10.f));
    if (x1 < x0 || y1 < y0) continue;
    boxes_.push_back((int32_t)x0); boxes_.push_back((int32_t)y0); boxes_.push_back((int32_t)x1); boxes_.push_back((int32_t)y1);
  }
  std::vector<ssx_keypoint> kps;
  {
    Stopwatch sw(times_.detect, times_.n_detect);
    compute_.DetectBoxes(img, boxes_, track_status_ == FrontendStatus::INITING ? orb_init_ : orb_, kps);
  }
  for (const auto& kp : kps) {
    auto f = std::make_shared<Feature>();
    f->x = kp.x; f->y = kp.y;
    current_frame_->features_left.push_back(std::move(f));
  }
  return (int)kps.size();
}

// frontend.cpp:346-428
int FrontEnd::FindFeaturesInRight()
{
  const auto& feats = current_frame_->features_left;
  const size_t n = feats.size();
  current_frame_->features_right.assign(n, nullptr);
  if (n == 0) return 0;
  std::vector<float> left(2 * n), right(2 * n);
  SE3 T_cw;
  if (reference_kf_) T_cw = current_frame_->relative_pose_to_kf * reference_kf_->pose;
  for (size_t i = 0; i < n; ++i) {
    left[2 * i] = feats[i]->x; left[2 * i + 1] = feats[i]->y;
    MapPointPtr mp = map_->Lock(feats[i]);
    if (mp && reference_kf_) {
      right_camera_.world2pixel(mp->position, T_cw, &right[2 * i]);    // a mapped point predicts its right-image position
    } else {
      right[2 * i] = feats[i]->x; right[2 * i + 1] = feats[i]->y;
    }
  }
  std::vector<uint8_t> status;
  {
    Stopwatch sw(times_.lk_stereo, times_.n_lk_stereo);
    compute_.TrackLK(*current_frame_->left_image, *current_frame_->right_image, left, right, status, false);
  }
  int num_good_points = 0;
  for (size_t i = 0; i < n; ++i)
    if (status[i]) {
      auto f = std::make_shared<Feature>();
      f->x = right[2 * i]; f->y = right[2 * i + 1];
      f->is_on_left_frame = false;
      current_frame_->features_right[i] = std::move(f);
      ++num_good_points;
    }
  return num_good_points;
}

// frontend.cpp:430-446
bool FrontEnd::SteroInit()
{
  DetectFeatures();
  const int tracked = FindFeaturesInRight();
  if (tracked < num_features_init_good_) {
    std::fprintf(stderr, "[frontend] frame %lu: too few stereo features to initialise (%d)\n", current_frame_->frame_id, tracked);
    return false;
  }
  if (BuidInitMap()) {
    track_status_ = FrontendStatus::TRACKING_GOOD;
    return true;
  }
  return false;
}

// frontend.cpp:448-498
bool FrontEnd::BuidInitMap()
{
  std::vector<size_t> idx;
  std::vector<double> uvL, uvR;
  const auto& fl = current_frame_->features_left;
  const auto& fr = current_frame_->features_right;
  for (size_t i = 0; i < fl.size(); ++i)
    if (fr[i]) {
      idx.push_back(i);
      uvL.push_back(fl[i]->x); uvL.push_back(fl[i]->y);
      uvR.push_back(fr[i]->x); uvR.push_back(fr[i]->y);
    }
  std::vector<double> xyz(3 * idx.size());
  std::vector<uint8_t> ok(idx.size(), 0);
  if (!idx.empty()) {
    Stopwatch sw(times_.triangulate, times_.n_triangulate);
    compute_.Triangulate((int)idx.size(), uvL.data(), uvR.data(), rig_, nullptr, xyz.data(), ok.data());
  }
  size_t cnt_init_landmarks = 0;
  for (size_t k = 0; k < idx.size(); ++k)
    if (ok[k]) {
      MapPointPtr mp = map_->NewMapPoint(&xyz[3 * k]);
      fl[idx[k]]->map_point = (long)mp->id;
      fr[idx[k]]->map_point = (long)mp->id;
      map_->InsertMapPoint(mp);
      ++cnt_init_landmarks;
    }
  if (cnt_init_landmarks < min_init_landmark_) {
    std::fprintf(stderr, "[frontend] initial map has %zu points, %u needed\n", cnt_init_landmarks, min_init_landmark_);
    return false;
  }
  InsertKeyFrame();
  return true;
}

// frontend.cpp:500-544
int FrontEnd::TriangulateNewPoints()
{
  const SE3 T_wc = (current_frame_->relative_pose_to_kf * reference_kf_->pose).inverse();
  std::vector<size_t> idx;
  std::vector<double> uvL, uvR;
  const auto& fl = current_frame_->features_left;
  const auto& fr = current_frame_->features_right;
  for (size_t i = 0; i < fl.size(); ++i) {
    if (map_->Lock(fl[i])) continue;                                   // already a map point
    if (!fr[i]) continue;                                              // LK failed
    idx.push_back(i);
    uvL.push_back(fl[i]->x); uvL.push_back(fl[i]->y);
    uvR.push_back(fr[i]->x); uvR.push_back(fr[i]->y);
  }
  if (idx.empty()) return 0;
  std::vector<double> xyz(3 * idx.size());
  std::vector<uint8_t> ok(idx.size(), 0);
  {
    Stopwatch sw(times_.triangulate, times_.n_triangulate);
    compute_.Triangulate((int)idx.size(), uvL.data(), uvR.data(), rig_, T_wc.data(), xyz.data(), ok.data());
  }
  int cnt = 0;
  for (size_t k = 0; k < idx.size(); ++k)
    if (ok[k]) {
      MapPointPtr mp = map_->NewMapPoint(&xyz[3 * k]);
      fl[idx[k]]->map_point = (long)mp->id;
      fr[idx[k]]->map_point = (long)mp->id;
      map_->InsertMapPoint(mp);
      ++cnt;
    }
  return cnt;
}

// frontend.cpp:546-580
bool FrontEnd::InsertKeyFrame()
{
  KeyFramePtr kf = map_->CreateKF(current_frame_);
  if (track_status_ == FrontendStatus::INITING) {
    kf->pose = SE3();
  } else {
    kf->pose = current_frame_->relative_pose_to_kf * reference_kf_->pose;
    kf->last_key_frame = (long)reference_kf_->key_frame_id;
    kf->relative_pose_to_last_kf = current_frame_->relative_pose_to_kf;
  }
  if (backend_) {
    Stopwatch sw(times_.bundle_adjust, times_.n_bundle_adjust);
    backend_->InsertKeyFrame(kf, open_backend_optimization_);
  }
  reference_kf_ = kf;
  current_frame_->relative_pose_to_kf = SE3();
  return true;
}

}  // namespace ssx::host
