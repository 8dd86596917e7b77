// ssvio_amd/host/backend.hpp -- Backend::InsertKeyFrame / ProcessNewKeyFrame / OptimizeActiveMap
// (/root/reference/src/ssvio/backend.cpp:24-245).  Two modes:
//   synchronous (default)   FrontEnd::InsertKeyFrame runs the keyframe insertion and the window optimisation inline:
//                           the window is optimised before the next frame is tracked -- deterministic, what the
//                           parity tests use;
//   Backend.Async: 1        the reference's layout (backend.cpp:24-55): a worker thread takes keyframes from a queue,
//                           inserts them into the map and optimises the active window on its own GPU context while
//                           the front-end keeps tracking; map access is serialised by Map::update_mutex exactly where
//                           the reference takes mmutex_map_update_ (front-end: one frame; backend: insertion + read,
//                           and the write-back of the result -- not the solve).  Results then depend on timing,
//                           as they do in the reference.
// Loop closing is not attached.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <exception>

#include "compute.hpp"
#include "frontend.hpp"
#include "map.hpp"

namespace ssx::host {

class Backend {
 public:
  Backend(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right);
  ~Backend();
  Backend(const Backend&) = delete;
  Backend& operator=(const Backend&) = delete;
  // called by the front-end WITH Map::update_mutex held (it is inside FrontEnd::GrabSteroImage)
  void InsertKeyFrame(const KeyFramePtr& kf, bool optimization);
  void OptimizeActiveMap();                 // synchronous: marshal + solve + apply, caller holds the map mutex
  void WaitIdle();                          // asynchronous mode: returns when the queue is empty and the worker idle
  bool async() const { return async_; }

  struct Stats { long windows = 0, lm_iterations = 0, edges = 0, outlier_edges = 0; };
  const Stats& stats() const { return stats_; }

 private:
  // the active window as flat arrays (ssx_ba_problem) + what is needed to write the result back
  struct Window {
    std::vector<KeyFramePtr> kfs;
    std::vector<MapPointPtr> mps;
    std::vector<FeaturePtr> edge_feature;
    std::vector<double> poses, points, edge_uv, poses_out, points_out;
    std::vector<uint8_t> point_fixed, edge_cam, edge_outlier;
    std::vector<int32_t> edge_pose, edge_point;
    int lm_iterations = 0;
    bool empty() const { return kfs.empty() || edge_feature.empty(); }
  };
  void Marshal(Window& w) const;            // map -> arrays              (map mutex held)
  void Solve(Window& w);                    // the GPU call               (no lock)
  void Apply(Window& w);                    // arrays -> map, outliers    (map mutex held)
  void Worker();
  void RethrowWorkerError();                // queue_mutex_ held

  Compute& compute_;
  std::shared_ptr<Map> map_;
  Camera camera_left_, camera_right_;
  int jac_mode_;
  Stats stats_;
  bool async_ = false;
  std::thread worker_;
  std::mutex queue_mutex_;
  std::condition_variable queue_cv_, idle_cv_;
  std::deque<std::pair<KeyFramePtr, bool>> queue_;
  bool stop_ = false, busy_ = false;
  std::exception_ptr worker_error_;         // first failure of the worker thread, handed to the caller thread
};

}  // namespace ssx::host
