// ssvio_amd/host/system.hpp -- ssvio::System without the viewer (/root/reference/src/ssvio/system.cpp:9-128):
// reads the settings file, builds the stereo rig (GenerateSteroCamera :54-113), the two extractor settings
// (GenerateORBextractor :115-128), Map, Backend, FrontEnd; RunStep = FrontEnd::GrabSteroImage (:46-52).
// SaveTrajectoryTUM is the trajectory writer the reference keeps in its viewer
// (/root/reference/src/ui/pangolin_window_impl.cpp:362-395): every keyframe in id order,
// "timestamp tx ty tz qx qy qz qw" of T_wc, fixed notation with 6 decimals.
#pragma once
#include <memory>
#include <string>

#include "backend.hpp"
#include "compute.hpp"
#include "frontend.hpp"
#include "map.hpp"
#include "setting.hpp"

namespace ssx::host {

class System {
 public:
  // compute: the implementation to run on; null = MakeSsxCompute(device)
  explicit System(const std::string& config_file_path, std::unique_ptr<Compute> compute = nullptr, int device = 0);

  bool RunStep(ImagePtr left, ImagePtr right, double timestamp);
  // One synthetic keyframe + two tracked frames of rows x cols through every compute call the loop makes (masked detection, stereo
  // and chained temporal LK, triangulation, pose-only LM, a window that grows to Map.ActiveMap.Size keyframes and slides once), results discarded: the GPU library loads
  // its kernels and sizes its workspaces here (15 - 20 ms for the first window solve alone) instead of inside the first frames.
  // Touches neither the map nor the tracker; a run with and without it writes the same trajectory.
  void Warmup(int rows, int cols);
  void SaveTrajectoryTUM(const std::string& path = std::string()) const;   // empty: Trajectory.Save.Path of the settings

  const Setting& setting() const { return setting_; }
  Map& map() { return *map_; }
  FrontEnd& frontend() { return *frontend_; }
  Backend& backend() { return *backend_; }

 private:
  Setting setting_;
  std::unique_ptr<Compute> compute_;
  Camera left_camera_, right_camera_;
  std::shared_ptr<Map> map_;
  std::unique_ptr<Backend> backend_;
  std::unique_ptr<FrontEnd> frontend_;
};

}  // namespace ssx::host
