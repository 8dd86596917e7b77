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
// The active window itself lives in one of two places (setting Backend.Window, default 1):
//   1  RESIDENT on the device (Compute::MakeBaWindow -> ssx_ba_window): every change the reference makes to its active map --
//      Map::InsertKeyFrame, RemoveOldActiveKeyframe, RemoveOldActiveMapPoints (map.cpp:18-58, 89-160), the outlier edges
//      OptimizeActiveMap unlinks and the map points it deletes (backend.cpp:205-244), map points the front-end condemned
//      (frontend.cpp:283-288) -- is mirrored as an edit of the window; a keyframe's pose, its new map points and its
//      observations are all that crosses PCIe, and the fixed flags of backend.cpp:125-130 are kept by the window itself;
//   0  re-marshalled from the map at every keyframe (Marshal: the reference's own way, backend.cpp:88-169).
// Both give the same bits (the window solves in id order): tests/test_host_gpu.py runs every sequence both ways and compares
// the per-frame logs and trajectory files byte for byte; Backend.Window.Check: 1 re-marshals the map beside the window at every
// keyframe and throws when the two graphs differ.  An implementation without a resident window (the CPU oracle of the tests)
// falls back to 0.
// Loop closing is not attached.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

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
  bool resident_window() const { return window_ != nullptr; }

  struct Stats {
    long windows = 0, lm_iterations = 0, edges = 0, outlier_edges = 0;
    // seconds on the host side of a keyframe: the map + window edits of an insertion, the export + solve call, the write-back
    double t_insert = 0, t_solve = 0, t_apply = 0;
  };
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
  // the resident window (map mutex held except for WindowSolve)
  void InsertIntoMap(const KeyFramePtr& kf);        // Map::InsertKeyFrame + its mirror on the window
  void WindowMirrorInsert(const KeyFramePtr& kf);
  void WindowDropCondemned();
  struct WindowResult {
    std::vector<int64_t> kf_ids, lm_ids;
    std::vector<int32_t> edge_pose, edge_point;
    std::vector<double> poses, points;
    std::vector<uint8_t> edge_outlier;
    int lm_iterations = 0;
    bool solved = false;
  };
  void WindowSolve(WindowResult& r);        // the GPU call               (no lock)
  void WindowApply(WindowResult& r);        // result -> map, outliers -> window and map
  void WindowCheckAgainstMap();             // Backend.Window.Check

  Compute& compute_;
  std::shared_ptr<Map> map_;
  Camera camera_left_, camera_right_;
  int jac_mode_;
  std::unique_ptr<BaWindow> window_;        // null: Marshal / Solve / Apply per keyframe
  bool window_check_ = false;
  std::unordered_set<unsigned long> in_window_;                                       // map points the window holds
  std::unordered_map<unsigned long, std::unordered_map<unsigned long, FeaturePtr>> window_feats_;   // keyframe -> map point -> the feature pushed
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
