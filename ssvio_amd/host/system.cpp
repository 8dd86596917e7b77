// ssvio_amd/host/system.cpp -- see system.hpp
#include "system.hpp"

#include <algorithm>
#include <cstring>
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

void System::Warmup(int rows, int cols)
{
  if (rows < 64 || cols < 64) return;
  // textured images (bilinear value noise in two octaves); the second and third are the first shifted by a few pixels
  auto make = [&](double dx, uint64_t id) {
    auto im = std::make_shared<Image>();
    im->rows = rows; im->cols = cols; im->id = id; im->data.resize((size_t)rows * cols);
    auto h = [](int x, int y, int o) {
      uint32_t v = (uint32_t)(x * 73856093) ^ (uint32_t)(y * 19349663) ^ (uint32_t)(o * 83492791);
      v ^= v >> 13; v *= 0x5bd1e995u; v ^= v >> 15;
      return (double)(v & 0xffff) / 65535.0;
    };
    const double cell[2] = {18.0, 5.0}, amp[2] = {130.0, 90.0};
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols; ++x) {
        double acc = 15.0;
        for (int o = 0; o < 2; ++o) {
          const double fx = (x + dx) / cell[o] + 64.0, fy = y / cell[o] + 64.0;
          const int ix = (int)fx, iy = (int)fy;
          const double ax = fx - ix, ay = fy - iy;
          acc += amp[o] * ((1 - ay) * ((1 - ax) * h(ix, iy, o) + ax * h(ix + 1, iy, o)) + ay * ((1 - ax) * h(ix, iy + 1, o) + ax * h(ix + 1, iy + 1, o)));
        }
        im->data[(size_t)y * cols + x] = (uint8_t)(acc > 255.0 ? 255.0 : acc);
      }
    return im;
  };
  // (ids far from the loader's: a resident pyramid or a pinned copy of these images is never mistaken for a frame's)
  const uint64_t id0 = ~uint64_t(0) - 16;
  ImagePtr L0 = make(0.0, id0), R0 = make(9.0, id0 + 1), L1 = make(1.5, id0 + 2), L2 = make(3.0, id0 + 3);
  ssx_orb_params prm{};
  prm.nfeatures = std::max(setting_.Get<int>("ORBextractor.nInitFeatures"), setting_.Get<int>("ORBextractor.nNewFeatures"));
  prm.scale_factor = setting_.Get<float>("ORBextractor.scaleFactor"); prm.nlevels = setting_.Get<int>("ORBextractor.nLevels");
  prm.ini_th_fast = setting_.Get<int>("ORBextractor.iniThFAST"); prm.min_th_fast = setting_.Get<int>("ORBextractor.minThFAST");
  std::vector<ssx_keypoint> kps;
  const std::vector<int32_t> boxes = {cols / 4, rows / 4, cols / 4 + 20, rows / 4 + 20};
  compute_->DetectBoxes(*L0, boxes, prm, kps);
  const int n = (int)kps.size();
  if (n < 8) return;
  std::vector<float> pts(2 * (size_t)n), right(2 * (size_t)n), next1, next2;
  for (int i = 0; i < n; ++i) { pts[2 * i] = kps[i].x; pts[2 * i + 1] = kps[i].y; }
  right = pts;
  std::vector<uint8_t> st;
  compute_->TrackLK(*L0, *R0, pts, right, st, false);
  std::vector<double> uvL(pts.begin(), pts.end()), uvR(right.begin(), right.end()), xyz(3 * (size_t)n, 0.0);
  std::vector<uint8_t> ok(n, 0);
  const ssx_stereo_rig rig{left_camera_.fx, left_camera_.fy, left_camera_.cx, left_camera_.cy, right_camera_.baseline};
  const SE3 T;
  compute_->Triangulate(n, uvL.data(), uvR.data(), rig, T.data(), xyz.data(), ok.data());
  next1 = pts;
  compute_->TrackLK(*L0, *L1, pts, next1, st, true);
  next2 = next1;
  compute_->TrackLK(*L1, *L2, next1, next2, st, true);
  // pose-only on points in front of the camera that project to the tracked pixels exactly (any well-posed problem will do)
  const double K4[4] = {left_camera_.fx, left_camera_.fy, left_camera_.cx, left_camera_.cy};
  std::vector<double> P(3 * (size_t)n), uv(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const double z = 8.0 + (i % 17);
    P[3 * i] = (pts[2 * i] - K4[2]) / K4[0] * z; P[3 * i + 1] = (pts[2 * i + 1] - K4[3]) / K4[1] * z; P[3 * i + 2] = z;
    uv[2 * i] = pts[2 * i]; uv[2 * i + 1] = pts[2 * i + 1];
  }
  double pose[7] = {0, 0, 0, 1, 0.01, -0.01, 0.02};
  std::vector<uint8_t> inl(n, 0);
  compute_->PoseOnly(pose, K4, n, P.data(), uv.data(), inl.data());
  // a window of three keyframes over the same points (the resident window if the Compute has one, else the marshalled problem)
  ssx_ba_options opt;
  ssx_ba_default_options(&opt);
  opt.jac_mode = setting_.Get<int>("Backend.Jacobian.Numeric") != 0 ? SSX_JAC_NUMERIC_G2O : SSX_JAC_ANALYTIC;
  double ext[14];
  std::memcpy(ext, left_camera_.pose.data(), 7 * sizeof(double));
  std::memcpy(ext + 7, right_camera_.pose.data(), 7 * sizeof(double));
  // the window grows to Map.ActiveMap.Size keyframes, is solved at 3 and when full, then slides once (pop + push) and is solved again:
  // the arenas reach the size of a full window, the kernels of multi-chunk windows and of the pop path are loaded
  const int full = std::max(3, std::min(16, setting_.Get<int>("Map.ActiveMap.Size")));
  const int nk = full + 1;
  std::vector<double> poses(7 * (size_t)nk, 0.0), obs_uv;
  std::vector<int64_t> ids(n), obs_lm;
  std::vector<uint8_t> fixed(n, 0), cam(n, 0);
  for (int i = 0; i < n; ++i) ids[i] = i;
  std::vector<int32_t> e_pose, e_point;
  for (int k = 0; k < nk; ++k) {
    poses[7 * k + 3] = 1.0; poses[7 * k + 4] = -0.3 * k;                  // T_cw: the camera moves 0.3 m along x per keyframe
    for (int i = 0; i < n; ++i) {
      const double xc = P[3 * i] - 0.3 * k, yc = P[3 * i + 1], zc = P[3 * i + 2];
      e_pose.push_back(k); e_point.push_back(i);
      obs_uv.push_back(K4[0] * xc / zc + K4[2] + 0.3 * ((i + k) % 3 - 1)); obs_uv.push_back(K4[1] * yc / zc + K4[3]);
    }
  }
  std::vector<double> poses_out(poses.size()), points_out(P.size());
  std::vector<uint8_t> outl(e_pose.size(), 0);
  ssx_ba_result res{};
  res.poses_out = poses_out.data(); res.points_out = points_out.data(); res.edge_outlier = outl.data();
  if (auto win = compute_->MakeBaWindow(K4, ext, opt)) {
    obs_lm.assign(ids.begin(), ids.end());
    for (int k = 0; k < nk; ++k) {
      if (k == full) win->Pop(0);
      win->Push(k, &poses[7 * k], k == 0 ? n : 0, ids.data(), P.data(), fixed.data(), n, obs_lm.data(), &obs_uv[2 * (size_t)n * k], cam.data());
      if (k == 2 || k >= full - 1) win->Solve(res);
    }
  } else {
    ssx_ba_problem prob{};
    prob.P = full; prob.poses = poses.data(); prob.L = n; prob.points = P.data(); prob.point_fixed = fixed.data(); prob.E = (int32_t)((size_t)n * full);
    prob.edge_pose = e_pose.data(); prob.edge_point = e_point.data(); prob.edge_uv = obs_uv.data();
    std::memcpy(prob.K, K4, sizeof(K4)); std::memcpy(prob.cam_ext, ext, sizeof(ext));
    compute_->BundleAdjust(prob, opt, res);
  }
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
