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
  async_ = cfg.Get<int>("Backend.Async") != 0;
  if (async_) worker_ = std::thread([this] { Worker(); });
}

Backend::~Backend()
{
  if (worker_.joinable()) {
    {
      std::lock_guard<std::mutex> lk(queue_mutex_);
      stop_ = true;
    }
    queue_cv_.notify_all();
    worker_.join();
  }
}

// backend.cpp:57-78.  Synchronous: ProcessNewKeyFrame + the optimisation the worker loop would start next, inline.
// Asynchronous: the keyframe goes to the queue (Backend::InsertKeyFrame of the reference) and the worker does the rest.
void Backend::InsertKeyFrame(const KeyFramePtr& kf, bool optimization)
{
  if (!async_) {
    map_->InsertKeyFrame(kf);
    if (optimization) OptimizeActiveMap();
    return;
  }
  {
    std::lock_guard<std::mutex> lk(queue_mutex_);
    RethrowWorkerError();                                            // a failed solve surfaces at the next insertion
    queue_.emplace_back(kf, optimization);
  }
  queue_cv_.notify_one();
}

void Backend::WaitIdle()
{
  if (!async_) return;
  std::unique_lock<std::mutex> lk(queue_mutex_);
  // (first the worker comes to rest -- also after a failure: the caller must never unwind, or touch the map, while the
  // worker is still inside a later batch; the synchronous mode this mirrors cannot do that either -- then the parked error)
  idle_cv_.wait(lk, [this] { return queue_.empty() && !busy_; });
  RethrowWorkerError();
}

// queue_mutex_ held.  The worker thread cannot let an exception escape (std::terminate): it parks it here and the
// caller thread gets it from WaitIdle() / the next InsertKeyFrame(), like the synchronous mode through its call stack.
void Backend::RethrowWorkerError()
{
  if (!worker_error_) return;
  std::exception_ptr e = worker_error_;
  worker_error_ = nullptr;
  std::rethrow_exception(e);
}

// backend.cpp:24-55 (BackendLoop) without the polling sleep: process every queued keyframe, optimise the window once the
// queue is empty (need_optimization_ is the flag of the LAST keyframe inserted, backend.cpp:72-77)
void Backend::Worker()
{
  for (;;) {
    std::deque<std::pair<KeyFramePtr, bool>> batch;
    {
      std::unique_lock<std::mutex> lk(queue_mutex_);
      queue_cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;                                    // stop requested and nothing left to do
      batch.swap(queue_);
      busy_ = true;
    }
    std::exception_ptr err;
    try {
      Window w;
      bool optimize = false;
      {
        std::lock_guard<std::mutex> map_lock(map_->update_mutex);
        for (auto& item : batch) { map_->InsertKeyFrame(item.first); optimize = item.second; }
        if (optimize) Marshal(w);
      }
      if (optimize && !w.empty()) {
        Solve(w);                                                    // throws on a HIP / argument error of the BA call
        std::lock_guard<std::mutex> map_lock(map_->update_mutex);
        Apply(w);
      }
    } catch (...) {
      err = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(queue_mutex_);
      busy_ = false;
      if (err && !worker_error_) worker_error_ = err;
    }
    idle_cv_.notify_all();
  }
}

// backend.cpp:78-245.  Vertices: every active keyframe (none fixed) and every active, non-outlier map point (fixed when
// the keyframe of its first observation has left the window); edges: the active observations held by active keyframes.
// Keyframes and map points are marshalled in ascending id order -- the order g2o gives its vertices.
void Backend::OptimizeActiveMap()
{
  Window w;
  Marshal(w);
  if (w.empty()) return;
  Solve(w);
  Apply(w);
}

void Backend::Marshal(Window& w) const
{
  const auto& active_kfs = map_->GetActiveKeyFrames();
  const auto& active_mps = map_->GetActiveMapPoints();

  for (auto& kv : active_kfs) w.kfs.push_back(kv.second);
  std::sort(w.kfs.begin(), w.kfs.end(), [](const KeyFramePtr& a, const KeyFramePtr& b) { return a->key_frame_id < b->key_frame_id; });
  std::unordered_map<unsigned long, int> kf_index;
  w.poses.resize(7 * w.kfs.size());
  for (size_t i = 0; i < w.kfs.size(); ++i) {
    kf_index[w.kfs[i]->key_frame_id] = (int)i;
    std::memcpy(&w.poses[7 * i], w.kfs[i]->pose.data(), 7 * sizeof(double));
  }

  std::vector<MapPointPtr> candidates;
  for (auto& kv : active_mps)
    if (!kv.second->is_outlier) candidates.push_back(kv.second);
  std::sort(candidates.begin(), candidates.end(), [](const MapPointPtr& a, const MapPointPtr& b) { return a->id < b->id; });

  for (auto& mp : candidates) {
    const size_t first_edge = w.edge_feature.size();
    for (auto& feat : mp->active_observations) {
      auto it = kf_index.find((unsigned long)feat->keyframe);
      if (feat->keyframe < 0 || it == kf_index.end() || feat->is_outlier) continue;
      w.edge_pose.push_back(it->second);
      w.edge_point.push_back((int32_t)w.mps.size());
      w.edge_uv.push_back(feat->x); w.edge_uv.push_back(feat->y);
      w.edge_cam.push_back(feat->is_on_left_frame ? 0 : 1);
      w.edge_feature.push_back(feat);
    }
    if (w.edge_feature.size() == first_edge) continue;                // no edge: g2o leaves such a vertex out of the active set
    const bool fixed = mp->observations.empty() || kf_index.find((unsigned long)mp->observations.front()->keyframe) == kf_index.end();
    w.mps.push_back(mp);
    w.points.insert(w.points.end(), mp->position, mp->position + 3);
    w.point_fixed.push_back(fixed ? 1 : 0);
  }
}

void Backend::Solve(Window& w)
{
  ssx_ba_problem prob{};
  prob.P = (int32_t)w.kfs.size(); prob.poses = w.poses.data(); prob.pose_fixed = nullptr;
  prob.L = (int32_t)w.mps.size(); prob.points = w.points.data(); prob.point_fixed = w.point_fixed.data();
  prob.E = (int32_t)w.edge_feature.size();
  prob.edge_pose = w.edge_pose.data(); prob.edge_point = w.edge_point.data(); prob.edge_uv = w.edge_uv.data(); prob.edge_cam = w.edge_cam.data();
  prob.K[0] = camera_left_.fx; prob.K[1] = camera_left_.fy; prob.K[2] = camera_left_.cx; prob.K[3] = camera_left_.cy;
  std::memcpy(prob.cam_ext, camera_left_.pose.data(), 7 * sizeof(double));
  std::memcpy(prob.cam_ext + 7, camera_right_.pose.data(), 7 * sizeof(double));

  ssx_ba_options opt;
  ssx_ba_default_options(&opt);                                        // 5 rounds x optimize(10), chi2 / Huber 5.891, inlier ratio 0.7
  opt.jac_mode = jac_mode_;
  w.poses_out.resize(w.poses.size()); w.points_out.resize(w.points.size());
  w.edge_outlier.assign(w.edge_feature.size(), 0);
  ssx_ba_result res{};
  res.poses_out = w.poses_out.data(); res.points_out = w.points_out.data(); res.edge_outlier = w.edge_outlier.data();
  compute_.BundleAdjust(prob, opt, res);
  w.lm_iterations = res.n_iters;
}

void Backend::Apply(Window& w)
{
  stats_.windows++; stats_.lm_iterations += w.lm_iterations; stats_.edges += (long)w.edge_feature.size();
  // outlier edges lose their observation; a map point without observations is condemned (backend.cpp:205-228)
  for (size_t e = 0; e < w.edge_feature.size(); ++e) {
    const FeaturePtr& feat = w.edge_feature[e];
    if (w.edge_outlier[e]) {
      stats_.outlier_edges++;
      feat->is_outlier = true;
      MapPointPtr mp = w.mps[w.edge_point[e]];
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
  for (size_t i = 0; i < w.kfs.size(); ++i) w.kfs[i]->pose = SE3(&w.poses_out[7 * i]);
  for (size_t j = 0; j < w.mps.size(); ++j) std::memcpy(w.mps[j]->position, &w.points_out[3 * j], 3 * sizeof(double));
  map_->RemoveAllOutlierMapPoints();
  map_->RemoveOldActiveMapPoints();
}

}  // namespace ssx::host
