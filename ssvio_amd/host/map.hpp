// ssvio_amd/host/map.hpp -- the bookkeeping either side of the compute path: features, map points, frames, keyframes
// and the active-window map of the reference, restated for the headless runner:
//   Feature    /root/reference/include/ssvio/feature.hpp:17-37
//   MapPoint   /root/reference/src/ssvio/mappoint.cpp:10-79      (observation lists, active observation count)
//   Frame      /root/reference/src/ssvio/frame.cpp:9-43
//   KeyFrame   /root/reference/src/ssvio/keyframe.cpp:11-58      (CreateKF links features and observations)
//   Map        /root/reference/src/ssvio/map.cpp:13-217          (InsertKeyFrame, the sliding active window)
// Ownership differs from the reference on purpose: map points are owned by the Map and referred to by id (the
// reference's weak_ptr "expired" == the id is no longer in the map), so the active window marshals into the flat
// arrays of ssx_ba_problem without chasing pointers.  Keyframes and map points stay in std::unordered_map keyed by id
// like the reference: Map::RemoveOldActiveKeyframe depends on that container's iteration order.
#pragma once
#include <list>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "dataset.hpp"
#include "se3.hpp"

namespace ssx::host {

constexpr long kNoMapPoint = -1;

struct Feature {
  float x = 0, y = 0;               // kp_position_.pt (cv::KeyPoint; size 7 and the other fields are never read)
  long map_point = kNoMapPoint;     // id of the associated map point
  long keyframe = -1;               // id of the keyframe that holds the feature (set by CreateKF)
  bool is_outlier = false;
  bool is_on_left_frame = true;
};
using FeaturePtr = std::shared_ptr<Feature>;

struct MapPoint {
  unsigned long id = 0;
  double position[3] = {0, 0, 0};
  bool is_outlier = false;
  std::list<FeaturePtr> observations, active_observations;
  int observed_times = 0, active_observed_times = 0;

  void AddObservation(const FeaturePtr& f) { observations.push_back(f); ++observed_times; }
  void AddActiveObservation(const FeaturePtr& f) { active_observations.push_back(f); ++active_observed_times; }
  void RemoveActiveObservation(const FeaturePtr& f);
  void RemoveObservation(const FeaturePtr& f);       // also clears f->map_point
};
using MapPointPtr = std::shared_ptr<MapPoint>;

struct Frame {
  unsigned long frame_id = 0;
  double timestamp = 0;
  ImagePtr left_image, right_image;
  SE3 pose;                          // T_cw
  SE3 relative_pose_to_kf;           // T_c,kf
  std::vector<FeaturePtr> features_left, features_right;   // features_right[i] pairs features_left[i]; null = LK failed
};
using FramePtr = std::shared_ptr<Frame>;

struct KeyFrame {
  unsigned long key_frame_id = 0, frame_id = 0;
  double timestamp = 0;
  SE3 pose;                          // T_cw
  std::vector<FeaturePtr> features_left;
  long last_key_frame = -1;
  SE3 relative_pose_to_last_kf;
};
using KeyFramePtr = std::shared_ptr<KeyFrame>;

class Map {
 public:
  using KeyFramesType = std::unordered_map<unsigned long, KeyFramePtr>;
  using MapPointsType = std::unordered_map<unsigned long, MapPointPtr>;

  explicit Map(unsigned num_active_key_frames) : num_active_key_frames_(num_active_key_frames) {}

  // factories: the reference numbers frames, keyframes and map points with function-local static counters
  FramePtr NewFrame(ImagePtr left, ImagePtr right, double timestamp);
  MapPointPtr NewMapPoint(const double* position);
  KeyFramePtr CreateKF(const FramePtr& frame);                       // KeyFrame::CreateKF

  // the map point a feature refers to, or null when there is none or it has been removed from the map
  MapPointPtr Lock(long map_point_id) const;
  MapPointPtr Lock(const FeaturePtr& f) const { return f ? Lock(f->map_point) : nullptr; }

  void InsertKeyFrame(const KeyFramePtr& kf);
  void InsertMapPoint(const MapPointPtr& mp);
  void InsertActiveMapPoint(const MapPointPtr& mp);
  void RemoveOldActiveKeyframe();
  void RemoveOldActiveMapPoints();
  void RemoveMapPoint(const MapPointPtr& mp);
  void AddOutlierMapPoint(unsigned long id) { outlier_map_points_.push_back(id); }
  void RemoveAllOutlierMapPoints();

  // Map::mmutex_map_update_ of the reference (map.hpp:66): the front-end holds it while it tracks a frame, the
  // backend while it inserts a keyframe, reads the active window and writes an optimisation back.  Only contended
  // when the backend runs on its own thread (Backend.Async).
  std::mutex update_mutex;

  const MapPointsType& GetAllMapPoints() const { return all_map_points_; }
  const KeyFramesType& GetAllKeyFrames() const { return all_key_frames_; }
  const MapPointsType& GetActiveMapPoints() const { return active_map_points_; }
  const KeyFramesType& GetActiveKeyFrames() const { return active_key_frames_; }
  // for a backend that mirrors the map into a resident window: the keyframe the last InsertKeyFrame dropped from the active
  // window (-1: none), and the map points condemned since the last RemoveAllOutlierMapPoints
  long last_removed_keyframe() const { return last_removed_keyframe_; }
  const std::list<unsigned long>& outlier_map_points() const { return outlier_map_points_; }

 private:
  MapPointsType all_map_points_, active_map_points_;
  KeyFramesType all_key_frames_, active_key_frames_;
  std::list<unsigned long> outlier_map_points_;
  KeyFramePtr current_keyframe_;
  long last_removed_keyframe_ = -1;
  unsigned num_active_key_frames_;
  unsigned long next_frame_id_ = 0, next_key_frame_id_ = 0, next_map_point_id_ = 0;
};

}  // namespace ssx::host
