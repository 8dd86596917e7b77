// ssvio_amd/host/system.cpp -- see system.hpp
#include "system.hpp"

#include <fstream>
#include <iomanip>
#include <map>
#include <stdexcept>

namespace ssx::host {

System::System(const std::string& config_file_path, std::unique_ptr<Compute> compute, int device) : setting_(config_file_path)
{
  compute_ = compute ? std::move(compute) : MakeSsxCompute(device);
  // system.cpp:54-113: intrinsics are read as float and widened; baseline = Camera.Base.Line / fx in float arithmetic;
  // the left camera sits at the rig origin, the right one at (-baseline, 0, 0)
  const float fx_l = setting_.Get<float>("Camera1.fx"), fy_l = setting_.Get<float>("Camera1.fy");
  const float cx_l = setting_.Get<float>("Camera1.cx"), cy_l = setting_.Get<float>("Camera1.cy");
  const float fx_r = setting_.Get<float>("Camera2.fx"), fy_r = setting_.Get<float>("Camera2.fy");
  const float cx_r = setting_.Get<float>("Camera2.cx"), cy_r = setting_.Get<float>("Camera2.cy");
  const float bf = setting_.Get<float>("Camera.Base.Line");
  if (!(fx_l > 0) || !(fx_r > 0)) throw std::runtime_error("System: Camera1.fx / Camera2.fx missing in " + config_file_path);
  const float baseline = bf / fx_r;
  left_camera_ = Camera{fx_l, fy_l, cx_l, cy_l, 0.0, SE3()};
  right_camera_ = Camera{fx_r, fy_r, cx_r, cy_r, baseline, SE3::translation(-(double)baseline, 0, 0)};

  map_ = std::make_shared<Map>((unsigned)setting_.Get<int>("Map.ActiveMap.Size"));
  backend_ = std::make_unique<Backend>(setting_, *compute_, map_, left_camera_, right_camera_);
  frontend_ = std::make_unique<FrontEnd>(setting_, *compute_, map_, left_camera_, right_camera_);
  frontend_->SetBackend(backend_.get());
}

bool System::RunStep(ImagePtr left, ImagePtr right, double timestamp)
{
  if (!left || !right || left->empty() || right->empty() || timestamp < 0) throw std::invalid_argument("System::RunStep: empty image or negative timestamp");
  if (left->rows != right->rows || left->cols != right->cols) throw std::invalid_argument("System::RunStep: left / right image sizes differ");
  return frontend_->GrabSteroImage(std::move(left), std::move(right), timestamp);
}

void System::SaveTrajectoryTUM(const std::string& path_in) const
{
  backend_->WaitIdle();                                               // asynchronous backend: let the last windows finish
  std::lock_guard<std::mutex> map_lock(map_->update_mutex);
  const std::string path = path_in.empty() ? setting_.Get<std::string>("Trajectory.Save.Path") : path_in;
  std::ofstream out(path, std::ios_base::out | std::ios_base::trunc);
  if (!out.is_open()) throw std::runtime_error("SaveTrajectoryTUM: cannot write " + path);
  out << std::fixed;
  std::map<unsigned long, KeyFramePtr> ordered(map_->GetAllKeyFrames().begin(), map_->GetAllKeyFrames().end());
  for (auto& kv : ordered) {
    const SE3 T_wc = kv.second->pose.inverse();
    double q[4];
    T_wc.rotation_quaternion(q);
    out << std::setprecision(6) << kv.second->timestamp << " " << T_wc.d[4] << " " << T_wc.d[5] << " " << T_wc.d[6] << " " << q[0] << " "
        << q[1] << " " << q[2] << " " << q[3] << std::endl;
  }
}

}  // namespace ssx::host
