// tests/host/oracle_runner.cpp -- TEST INFRASTRUCTURE: ssx_run_kitti's loop on the CPU oracle.
//   oracle_runner <config yaml> <sequence dir> <trajectory out> [max frames]
// Prints one line per frame: "frame <id> status <s> features <n> keyframes <k> points <p>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "../../ssvio_amd/host/system.hpp"
#include "oracle_compute.hpp"

int main(int argc, char** argv)
{
  using namespace ssx::host;
  if (argc < 4) return 2;
  try {
    std::vector<std::string> left, right;
    std::vector<double> ts;
    LoadKittiImagesTimestamps(argv[2], left, right, ts);
    size_t n = left.size();
    if (argc > 4) n = std::min(n, (size_t)std::atol(argv[4]));
    std::unique_ptr<Compute> compute;
    if (!std::getenv("SSX_HOST_TEST_GPU")) compute = std::make_unique<OracleCompute>();   // unset: the oracle; set: libssx.so
    System system(argv[1], std::move(compute));
    double t_steps = 0, t_first = 0;
    for (size_t i = 0; i < n; ++i) {
      ImagePtr l = imread_gray(left[i]), r = imread_gray(right[i]);
      const auto t0 = std::chrono::steady_clock::now();
      system.RunStep(l, r, ts[i]);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      (i == 0 ? t_first : t_steps) += dt;
      std::lock_guard<std::mutex> map_lock(system.map().update_mutex);     // the backend may run on its own thread (Backend.Async)
      // camera centre of the frame: T_cw = relative pose to the reference keyframe * that keyframe's pose
      SE3 T_wc;
      if (system.frontend().reference_kf()) T_wc = (system.frontend().current_frame()->relative_pose_to_kf * system.frontend().reference_kf()->pose).inverse();
      std::printf("frame %zu status %d features %zu keyframes %zu points %zu active_kfs %zu active_points %zu centre %.6f %.6f %.6f\n", i,
                  (int)system.frontend().status(), system.frontend().current_frame()->features_left.size(), system.map().GetAllKeyFrames().size(),
                  system.map().GetAllMapPoints().size(), system.map().GetActiveKeyFrames().size(), system.map().GetActiveMapPoints().size(),
                  T_wc.d[4], T_wc.d[5], T_wc.d[6]);
    }
    std::printf("runstep_seconds first %.6f rest %.6f\n", t_first, t_steps);
    system.SaveTrajectoryTUM(argv[3]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  return 0;
}
