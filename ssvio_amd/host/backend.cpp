// ssvio_amd/host/backend.cpp -- see backend.hpp
#include "backend.hpp"

#include <algorithm>
#include <cstring>

namespace ssx::host {

Backend::Backend(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right)
    : compute_(compute), map_(std::move(map)), camera_left_(left), camera_right_(right)
{
  // the reference's EdgeProjection leaves linearizeOplus to g2o's numeric differentiation (g2otypes.hpp:133-153);
  // Backend.Jacobian.Numeric: 1 reproduces that, the default is the analytic Jacobian (same optimum, fewer flops)
  jac_mode_ = cfg.Get<int>("Backend.Jacobian.Numeric") != 0 ? SSX_JAC_NUMERIC_G2O : SSX_JAC_ANALYTIC;
}

// backend.cpp:57-78: ProcessNewKeyFrame + the optimisation the worker loop would start next
void Backend::InsertKeyFrame(const KeyFramePtr& kf, bool optimization)
{
  map_->InsertKeyFrame(kf);
  if (optimization) OptimizeActiveMap();
}

// backend.cpp:78-245.  Vertices: every active keyframe (none fixed) and every active, non-outlier map point (fixed when
// the keyframe of its first observation has left the window); edges: the active observations held by active keyframes.
// Keyframes and map points are marshalled in ascending id order -- the order g2o gives its vertices.
void Backend::OptimizeActiveMap()
{
  const auto& active_kfs = map_->GetActiveKeyFrames();
  const auto& active_mps = map_->GetActiveMapPoints();

  std::vector<KeyFramePtr> kfs;
  for (auto& kv : active_kfs) kfs.push_back(kv.second);
  std::sort(kfs.begin(), kfs.end(), [](const KeyFramePtr& a, const KeyFramePtr& b) { return a->key_frame_id < b->key_frame_id; });
  std::unordered_map<unsigned long, int> kf_index;
  std::vector<double> poses(7 * kfs.size());
  for (size_t i = 0; i < kfs.size(); ++i) {
    kf_index[kfs[i]->key_frame_id] = (int)i;
    std::memcpy(&poses[7 * i], kfs[i]->pose.data(), 7 * sizeof(double));
  }

  std::vector<MapPointPtr> candidates;
  for (auto& kv : active_mps)
    if (!kv.second->is_outlier) candidates.push_back(kv.second);
  std::sort(candidates.begin(), candidates.end(), [](const MapPointPtr& a, const MapPointPtr& b) { return a->id < b->id; });

  std::vector<MapPointPtr> mps;
  std::vector<double> points;
  std::vector<uint8_t> point_fixed, edge_cam;
  std::vector<int32_t> edge_pose, edge_point;
  std::vector<double> edge_uv;
  std::vector<FeaturePtr> edge_feature;
  for (auto& mp : candidates) {
    const size_t first_edge = edge_feature.size();
    for (auto& feat : mp->active_observations) {
      auto it = kf_index.find((unsigned long)feat->keyframe);
      if (feat->keyframe < 0 || it == kf_index.end() || feat->is_outlier) continue;
      edge_pose.push_back(it->second);
      edge_point.push_back((int32_t)mps.size());
      edge_uv.push_back(feat->x); edge_uv.push_back(feat->y);
      edge_cam.push_back(feat->is_on_left_frame ? 0 : 1);
      edge_feature.push_back(feat);
    }
    if (edge_feature.size() == first_edge) continue;                  // no edge: g2o leaves such a vertex out of the active set
    const bool fixed = mp->observations.empty() || kf_index.find((unsigned long)mp->observations.front()->keyframe) == kf_index.end();
    mps.push_back(mp);
    points.insert(points.end(), mp->position, mp->position + 3);
    point_fixed.push_back(fixed ? 1 : 0);
  }
  if (kfs.empty() || edge_feature.empty()) return;

  ssx_ba_problem prob{};
  prob.P = (int32_t)kfs.size(); prob.poses = poses.data(); prob.pose_fixed = nullptr;
  prob.L = (int32_t)mps.size(); prob.points = points.data(); prob.point_fixed = point_fixed.data();
  prob.E = (int32_t)edge_feature.size();
  prob.edge_pose = edge_pose.data(); prob.edge_point = edge_point.data(); prob.edge_uv = edge_uv.data(); prob.edge_cam = edge_cam.data();
  prob.K[0] = camera_left_.fx; prob.K[1] = camera_left_.fy; prob.K[2] = camera_left_.cx; prob.K[3] = camera_left_.cy;
  std::memcpy(prob.cam_ext, camera_left_.pose.data(), 7 * sizeof(double));
  std::memcpy(prob.cam_ext + 7, camera_right_.pose.data(), 7 * sizeof(double));

  ssx_ba_options opt;
  ssx_ba_default_options(&opt);                                        // 5 rounds x optimize(10), chi2 / Huber 5.891, inlier ratio 0.7
  opt.jac_mode = jac_mode_;
  std::vector<double> poses_out(poses.size()), points_out(points.size());
  std::vector<uint8_t> edge_outlier(edge_feature.size(), 0);
  ssx_ba_result res{};
  res.poses_out = poses_out.data(); res.points_out = points_out.data(); res.edge_outlier = edge_outlier.data();
  compute_.BundleAdjust(prob, opt, res);
  stats_.windows++; stats_.lm_iterations += res.n_iters; stats_.edges += prob.E;

  // outlier edges lose their observation; a map point without observations is condemned (backend.cpp:205-228)
  for (size_t e = 0; e < edge_feature.size(); ++e) {
    const FeaturePtr& feat = edge_feature[e];
    if (edge_outlier[e]) {
      stats_.outlier_edges++;
      feat->is_outlier = true;
      MapPointPtr mp = mps[edge_point[e]];
      mp->RemoveActiveObservation(feat);
      mp->RemoveObservation(feat);
      if (mp->observations.empty()) {
        mp->is_outlier = true;
        map_->AddOutlierMapPoint(mp->id);
      }
      feat->map_point = kNoMapPoint;
    } else {
      feat->is_outlier = false;
    }
  }
  for (size_t i = 0; i < kfs.size(); ++i) kfs[i]->pose = SE3(&poses_out[7 * i]);
  for (size_t j = 0; j < mps.size(); ++j) std::memcpy(mps[j]->position, &points_out[3 * j], 3 * sizeof(double));
  map_->RemoveAllOutlierMapPoints();
  map_->RemoveOldActiveMapPoints();
}

}  // namespace ssx::host
