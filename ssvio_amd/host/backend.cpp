// ssvio_amd/host/backend.cpp -- see backend.hpp
#include "backend.hpp"

#include <chrono>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>

namespace ssx::host {

namespace {
struct Timed {
  double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Timed(double& a) : acc(a) {}
  ~Timed() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

Backend::Backend(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right)
    : compute_(compute), map_(std::move(map)), camera_left_(left), camera_right_(right)
{
  // the reference's EdgeProjection leaves linearizeOplus to g2o's numeric differentiation (g2otypes.hpp:133-153);
  // Backend.Jacobian.Numeric: 1 reproduces that, the default is the analytic Jacobian (same optimum, fewer flops)
  jac_mode_ = cfg.Get<int>("Backend.Jacobian.Numeric") != 0 ? SSX_JAC_NUMERIC_G2O : SSX_JAC_ANALYTIC;
  async_ = cfg.Get<int>("Backend.Async") != 0;
  if (!cfg.Has("Backend.Window") || cfg.Get<int>("Backend.Window") != 0) {
    ssx_ba_options opt;
    ssx_ba_default_options(&opt);                                      // 5 rounds x optimize(10), chi2 / Huber 5.891, inlier ratio 0.7
    opt.jac_mode = jac_mode_;
    const double K4[4] = {camera_left_.fx, camera_left_.fy, camera_left_.cx, camera_left_.cy};
    double ext[14];
    std::memcpy(ext, camera_left_.pose.data(), 7 * sizeof(double));
    std::memcpy(ext + 7, camera_right_.pose.data(), 7 * sizeof(double));
    window_ = compute_.MakeBaWindow(K4, ext, opt);                     // null: this Compute has no resident window
  }
  window_check_ = cfg.Get<int>("Backend.Window.Check") != 0;
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
    InsertIntoMap(kf);
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
      WindowResult wr;
      bool optimize = false;
      {
        std::lock_guard<std::mutex> map_lock(map_->update_mutex);
        for (auto& item : batch) { InsertIntoMap(item.first); optimize = item.second; }
        if (optimize && window_) WindowDropCondemned();
        if (optimize && !window_) Marshal(w);
      }
      if (optimize && window_) {
        WindowSolve(wr);                                             // the window is this thread's alone: no lock
        if (wr.solved) {
          std::lock_guard<std::mutex> map_lock(map_->update_mutex);
          WindowApply(wr);
        }
      } else if (optimize && !w.empty()) {
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
  if (window_) {
    WindowDropCondemned();
    WindowResult r;
    WindowSolve(r);
    if (r.solved) WindowApply(r);
    return;
  }
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

// ---- the resident window ---------------------------------------------------------------------------------------------

void Backend::InsertIntoMap(const KeyFramePtr& kf)
{
  Timed tm(stats_.t_insert);
  map_->InsertKeyFrame(kf);
  if (window_) WindowMirrorInsert(kf);
}

// Map::InsertKeyFrame on the window: the keyframe with the observations Marshal would turn into edges (live, not condemned map
// points), the map points the window does not hold yet -- new ones, or ones that come BACK after all their observers had left
// or been unlinked: those enter with the flag of backend.cpp:125-130 as the map gives it --, then the keyframe the map dropped.
void Backend::WindowMirrorInsert(const KeyFramePtr& kf)
{
  const auto& active_kfs = map_->GetActiveKeyFrames();
  const auto& active_mps = map_->GetActiveMapPoints();
  for (auto it = in_window_.begin(); it != in_window_.end();) it = active_mps.count(*it) ? std::next(it) : in_window_.erase(it);
  std::vector<int64_t> new_ids, obs_lm;
  std::vector<double> new_xyz, obs_uv;
  std::vector<uint8_t> new_fixed, obs_cam;
  auto& feats = window_feats_[kf->key_frame_id];
  for (auto& feat : kf->features_left) {
    MapPointPtr mp = map_->Lock(feat);
    if (!mp || mp->is_outlier || feat->is_outlier) continue;
    if (!in_window_.count(mp->id)) {
      in_window_.insert(mp->id);
      new_ids.push_back((int64_t)mp->id);
      new_xyz.insert(new_xyz.end(), mp->position, mp->position + 3);
      const bool fixed = mp->observations.empty() || active_kfs.find((unsigned long)mp->observations.front()->keyframe) == active_kfs.end();
      new_fixed.push_back(fixed ? 1 : 0);
    }
    obs_lm.push_back((int64_t)mp->id);
    obs_uv.push_back(feat->x); obs_uv.push_back(feat->y);
    obs_cam.push_back(feat->is_on_left_frame ? 0 : 1);
    // (one left feature per (keyframe, map point): WindowApply resolves an edge to its Feature through this pair; a keyframe that
    // carried two would leave the second one's flags stale, so the assumption is enforced instead of overwritten silently)
    if (!feats.emplace(mp->id, feat).second)
      throw std::logic_error("Backend: keyframe " + std::to_string(kf->key_frame_id) + " holds two left features of map point " + std::to_string(mp->id));
  }
  window_->Push((int64_t)kf->key_frame_id, kf->pose.data(), (int)new_ids.size(), new_ids.data(), new_xyz.data(), new_fixed.data(), (int)obs_lm.size(),
                obs_lm.data(), obs_uv.data(), obs_cam.data());
  const long victim = map_->last_removed_keyframe();
  if (victim >= 0) {
    window_->Pop((int64_t)victim);
    window_feats_.erase((unsigned long)victim);
    for (auto it = in_window_.begin(); it != in_window_.end();) it = active_mps.count(*it) ? std::next(it) : in_window_.erase(it);
  }
  if (window_check_) WindowCheckAgainstMap();
}

// backend.cpp:116 `if (mp->is_outlier_) continue`: map points the front-end condemned since the last optimisation leave the graph
void Backend::WindowDropCondemned()
{
  std::vector<int64_t> ids;
  for (unsigned long id : map_->outlier_map_points())
    if (in_window_.erase(id)) ids.push_back((int64_t)id);
  if (!ids.empty()) window_->RemoveLandmarks((int)ids.size(), ids.data());
  if (window_check_) WindowCheckAgainstMap();
}

void Backend::WindowSolve(WindowResult& r)
{
  Timed tm(stats_.t_solve);
  int nk = 0, nl = 0, no = 0;
  window_->Size(nk, nl, no);
  if (nk == 0 || no == 0) return;
  r.kf_ids.resize(nk); r.lm_ids.resize(nl); r.edge_pose.resize(no); r.edge_point.resize(no);
  r.poses.resize(7 * (size_t)nk); r.points.resize(3 * (size_t)nl); r.edge_outlier.assign(no, 0);
  window_->Export(r.kf_ids.data(), r.lm_ids.data(), nullptr, r.edge_pose.data(), r.edge_point.data(), nullptr);
  ssx_ba_result res{};
  res.poses_out = r.poses.data(); res.points_out = r.points.data(); res.edge_outlier = r.edge_outlier.data();
  window_->Solve(res);
  r.lm_iterations = res.n_iters;
  r.solved = true;
}

// backend.cpp:205-244 on the map, and the same edits on the window
void Backend::WindowApply(WindowResult& r)
{
  Timed tm(stats_.t_apply);
  stats_.windows++; stats_.lm_iterations += r.lm_iterations; stats_.edges += (long)r.edge_outlier.size();
  const auto& all_mps = map_->GetAllMapPoints();
  for (size_t e = 0; e < r.edge_outlier.size(); ++e) {
    const unsigned long kf_id = (unsigned long)r.kf_ids[r.edge_pose[e]], mp_id = (unsigned long)r.lm_ids[r.edge_point[e]];
    auto kf_it = window_feats_.find(kf_id);
    if (kf_it == window_feats_.end()) throw std::logic_error("Backend: the window holds an observation of keyframe " + std::to_string(kf_id) + " the backend never pushed");
    auto f_it = kf_it->second.find(mp_id);
    if (f_it == kf_it->second.end()) throw std::logic_error("Backend: the window holds an observation the backend never pushed");
    const FeaturePtr& feat = f_it->second;
    if (!r.edge_outlier[e]) { feat->is_outlier = false; continue; }
    stats_.outlier_edges++;
    feat->is_outlier = true;
    auto mp_it = all_mps.find(mp_id);
    if (mp_it != all_mps.end()) {                                     // (async: the front-end may have deleted it meanwhile)
      MapPointPtr mp = mp_it->second;
      mp->RemoveActiveObservation(feat);
      mp->RemoveObservation(feat);
      if (mp->observations.empty()) {
        mp->is_outlier = true;
        map_->AddOutlierMapPoint(mp->id);
      }
    }
    feat->map_point = kNoMapPoint;
    kf_it->second.erase(f_it);
  }
  window_->RemoveFlagged((int)r.edge_outlier.size(), r.edge_outlier.data());
  const auto& all_kfs = map_->GetAllKeyFrames();
  for (size_t i = 0; i < r.kf_ids.size(); ++i) all_kfs.at((unsigned long)r.kf_ids[i])->pose = SE3(&r.poses[7 * i]);
  for (size_t j = 0; j < r.lm_ids.size(); ++j) {
    auto it = all_mps.find((unsigned long)r.lm_ids[j]);
    if (it != all_mps.end()) std::memcpy(it->second->position, &r.points[3 * j], 3 * sizeof(double));
  }
  // map points condemned during the solve (asynchronous front-end) are still in the window: out with them before the map forgets them
  WindowDropCondemned();
  map_->RemoveAllOutlierMapPoints();
  map_->RemoveOldActiveMapPoints();
  const auto& active_mps = map_->GetActiveMapPoints();
  for (auto it = in_window_.begin(); it != in_window_.end();) it = active_mps.count(*it) ? std::next(it) : in_window_.erase(it);
  if (window_check_) WindowCheckAgainstMap();
}

// Backend.Window.Check: the graph Marshal builds from the map against what the window exports -- keyframe ids, map-point ids,
// fixed flags, and the observations as (keyframe, map point, pixel) triples.  Map points the front-end has condemned but
// WindowDropCondemned has not seen yet are the one legitimate difference and are taken out of the comparison.
void Backend::WindowCheckAgainstMap()
{
  Window w;
  Marshal(w);
  int nk = 0, nl = 0, no = 0;
  window_->Size(nk, nl, no);
  std::vector<int64_t> kf_ids(nk + 1), lm_ids(nl + 1);
  std::vector<uint8_t> fixed(nl + 1);
  std::vector<int32_t> ep(no + 1), el(no + 1);
  std::vector<double> uv(2 * (size_t)no + 2);
  window_->Export(kf_ids.data(), lm_ids.data(), fixed.data(), ep.data(), el.data(), uv.data());
  auto fail = [&](const std::string& what) { throw std::logic_error("Backend.Window.Check: the resident window and the map disagree: " + what); };
  std::unordered_set<unsigned long> pending;                          // condemned, still in the window
  for (unsigned long id : map_->outlier_map_points()) if (in_window_.count(id)) pending.insert(id);
  if ((size_t)nk != w.kfs.size()) fail("keyframe count " + std::to_string(nk) + " vs " + std::to_string(w.kfs.size()));
  for (int i = 0; i < nk; ++i) if ((unsigned long)kf_ids[i] != w.kfs[i]->key_frame_id) fail("keyframe ids");
  using Tri = std::tuple<unsigned long, unsigned long, double, double>;
  std::vector<Tri> a, b;
  std::vector<std::pair<unsigned long, int>> la, lb;
  for (int j = 0; j < nl; ++j) if (!pending.count((unsigned long)lm_ids[j])) la.emplace_back((unsigned long)lm_ids[j], fixed[j]);
  for (size_t j = 0; j < w.mps.size(); ++j) lb.emplace_back(w.mps[j]->id, w.point_fixed[j]);
  if (la != lb) fail("map points or their fixed flags (" + std::to_string(la.size()) + " vs " + std::to_string(lb.size()) + ")");
  for (int e = 0; e < no; ++e)
    if (!pending.count((unsigned long)lm_ids[el[e]])) a.emplace_back((unsigned long)kf_ids[ep[e]], (unsigned long)lm_ids[el[e]], uv[2 * (size_t)e], uv[2 * (size_t)e + 1]);
  for (size_t e = 0; e < w.edge_feature.size(); ++e)
    b.emplace_back(w.kfs[w.edge_pose[e]]->key_frame_id, w.mps[w.edge_point[e]]->id, w.edge_uv[2 * e], w.edge_uv[2 * e + 1]);
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  if (a != b) fail("observations (" + std::to_string(a.size()) + " vs " + std::to_string(b.size()) + ")");
}

}  // namespace ssx::host
