// ssvio_amd/host/backend.hpp -- Backend::InsertKeyFrame / ProcessNewKeyFrame / OptimizeActiveMap
// (/root/reference/src/ssvio/backend.cpp:57-245).  The reference runs these on a worker thread that polls a keyframe
// queue; the headless runner calls them synchronously from FrontEnd::InsertKeyFrame, i.e. the window is optimised
// before the next frame is tracked (deterministic; the GPU solve takes ~1-2 ms).  Loop closing is not attached.
#pragma once
#include <memory>

#include "compute.hpp"
#include "frontend.hpp"
#include "map.hpp"

namespace ssx::host {

class Backend {
 public:
  Backend(const Setting& cfg, Compute& compute, std::shared_ptr<Map> map, const Camera& left, const Camera& right);
  void InsertKeyFrame(const KeyFramePtr& kf, bool optimization);
  void OptimizeActiveMap();

  struct Stats { long windows = 0, lm_iterations = 0, edges = 0, outlier_edges = 0; };
  const Stats& stats() const { return stats_; }

 private:
  Compute& compute_;
  std::shared_ptr<Map> map_;
  Camera camera_left_, camera_right_;
  int jac_mode_;
  Stats stats_;
};

}  // namespace ssx::host
