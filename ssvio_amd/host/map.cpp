// ssvio_amd/host/map.cpp -- see map.hpp
#include "map.hpp"

#include <stdexcept>

namespace ssx::host {

void MapPoint::RemoveActiveObservation(const FeaturePtr& f)
{
  for (auto it = active_observations.begin(); it != active_observations.end(); ++it)
    if (*it == f) {                                     // one entry per feature (mappoint.cpp:33-46)
      active_observations.erase(it);
      --active_observed_times;
      return;
    }
}

void MapPoint::RemoveObservation(const FeaturePtr& f)
{
  for (auto it = observations.begin(); it != observations.end(); ++it)
    if (*it == f) {
      observations.erase(it);
      f->map_point = kNoMapPoint;                       // mappoint.cpp:56
      --observed_times;
      return;
    }
}

FramePtr Map::NewFrame(ImagePtr left, ImagePtr right, double timestamp)
{
  auto f = std::make_shared<Frame>();
  f->left_image = std::move(left);
  f->right_image = std::move(right);
  f->timestamp = timestamp;
  f->frame_id = next_frame_id_++;
  return f;
}

MapPointPtr Map::NewMapPoint(const double* position)
{
  auto mp = std::make_shared<MapPoint>();
  mp->id = next_map_point_id_++;
  for (int i = 0; i < 3; ++i) mp->position[i] = position[i];
  return mp;
}

// keyframe.cpp:11-50: the keyframe shares the frame's left feature objects, becomes their holder, and every feature
// with a live map point is appended to that point's observations
KeyFramePtr Map::CreateKF(const FramePtr& frame)
{
  auto kf = std::make_shared<KeyFrame>();
  kf->key_frame_id = next_key_frame_id_++;
  kf->frame_id = frame->frame_id;
  kf->timestamp = frame->timestamp;
  kf->features_left = frame->features_left;
  for (auto& feat : kf->features_left) {
    feat->keyframe = (long)kf->key_frame_id;
    if (MapPointPtr mp = Lock(feat)) mp->AddObservation(feat);
  }
  return kf;
}

MapPointPtr Map::Lock(long id) const
{
  if (id < 0) return nullptr;
  auto it = all_map_points_.find((unsigned long)id);
  return it == all_map_points_.end() ? nullptr : it->second;
}

// map.cpp:18-58
void Map::InsertKeyFrame(const KeyFramePtr& kf)
{
  current_keyframe_ = kf;
  last_removed_keyframe_ = -1;
  if (all_key_frames_.count(kf->key_frame_id)) throw std::logic_error("Map::InsertKeyFrame: keyframe id inserted twice");
  all_key_frames_.insert({kf->key_frame_id, kf});
  active_key_frames_.insert({kf->key_frame_id, kf});
  for (auto& feat : kf->features_left)
    if (MapPointPtr mp = Lock(feat)) {
      mp->AddActiveObservation(feat);
      InsertActiveMapPoint(mp);
    }
  if (active_key_frames_.size() > num_active_key_frames_) {
    RemoveOldActiveKeyframe();
    RemoveOldActiveMapPoints();
  }
}

void Map::InsertMapPoint(const MapPointPtr& mp)
{
  if (!all_map_points_.insert({mp->id, mp}).second) throw std::logic_error("Map::InsertMapPoint: map point id inserted twice");
}

void Map::InsertActiveMapPoint(const MapPointPtr& mp) { active_map_points_[mp->id] = mp; }

// map.cpp:89-146.  The window drops the keyframe closest to the current one when it is closer than 0.2 (|log| of the
// relative pose), otherwise the farthest.  The running minimum is only updated in the `else` of the maximum test and
// the ids start at 0 -- kept as they are, together with the unordered_map iteration order they depend on.
void Map::RemoveOldActiveKeyframe()
{
  if (!current_keyframe_) return;
  double max_dis = 0, min_dis = 9999;
  unsigned long max_id = 0, min_id = 0;
  const SE3 Twc = current_keyframe_->pose.inverse();
  for (auto& kv : active_key_frames_) {
    if (kv.second == current_keyframe_) continue;
    const double dis = (kv.second->pose * Twc).log_norm();
    if (dis > max_dis) {
      max_dis = dis; max_id = kv.first;
    } else if (dis < min_dis) {
      min_dis = dis; min_id = kv.first;
    }
  }
  const double min_dis_th = 0.2;
  KeyFramePtr victim = active_key_frames_.at(min_dis < min_dis_th ? min_id : max_id);
  active_key_frames_.erase(victim->key_frame_id);
  last_removed_keyframe_ = (long)victim->key_frame_id;
  for (auto& feat : victim->features_left)
    if (MapPointPtr mp = Lock(feat)) mp->RemoveActiveObservation(feat);
}

void Map::RemoveOldActiveMapPoints()
{
  for (auto it = active_map_points_.begin(); it != active_map_points_.end();)
    it = it->second->active_observed_times == 0 ? active_map_points_.erase(it) : std::next(it);
}

void Map::RemoveMapPoint(const MapPointPtr& mp)
{
  all_map_points_.erase(mp->id);
  active_map_points_.erase(mp->id);
}

void Map::RemoveAllOutlierMapPoints()
{
  for (unsigned long id : outlier_map_points_) {
    all_map_points_.erase(id);
    active_map_points_.erase(id);
  }
  outlier_map_points_.clear();
}

}  // namespace ssx::host
