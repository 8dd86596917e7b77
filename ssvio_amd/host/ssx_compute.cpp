// ssvio_amd/host/ssx_compute.cpp -- Compute on libssx.so (see compute.hpp)
#include <stdexcept>
#include <string>

#include "../../include/ssx_shim.hpp"
#include "compute.hpp"

namespace ssx::host {
namespace {

class SsxBaWindow final : public BaWindow {
 public:
  SsxBaWindow(ssx::Context& ctx, const double* K4, const double* cam_ext14, const ssx_ba_options& opt) : ctx_(ctx)
  {
    ctx_.check(ssx_ba_window_create(ctx_.get(), &opt, K4, cam_ext14, &win_));
    ctx_.check(ssx_ba_window_set_fix_rule(win_, 1));
  }
  ~SsxBaWindow() override { ssx_ba_window_destroy(win_); }
  void Push(int64_t kf_id, const double* pose7, int n_new, const int64_t* new_ids, const double* new_xyz, const uint8_t* new_fixed, int n_obs,
            const int64_t* obs_lm, const double* obs_uv, const uint8_t* obs_cam) override
  {
    ctx_.check(ssx_ba_window_push_keyframe(win_, kf_id, pose7, 0, n_new, new_ids, new_xyz, new_fixed, n_obs, obs_lm, obs_uv, obs_cam));
  }
  void Pop(int64_t kf_id) override { ctx_.check(ssx_ba_window_pop_keyframe(win_, kf_id)); }
  void RemoveLandmarks(int n, const int64_t* lm_ids) override { ctx_.check(ssx_ba_window_remove_landmarks(win_, n, lm_ids, nullptr)); }
  void RemoveFlagged(int n_obs, const uint8_t* flags) override { ctx_.check(ssx_ba_window_remove_flagged(win_, n_obs, flags, nullptr)); }
  void Size(int& nk, int& nl, int& no) override
  {
    int32_t a = 0, b = 0, c = 0;
    ctx_.check(ssx_ba_window_size(win_, &a, &b, &c));
    nk = a; nl = b; no = c;
  }
  void Export(int64_t* kf_ids, int64_t* lm_ids, uint8_t* point_fixed, int32_t* edge_pose, int32_t* edge_point, double* edge_uv) override
  {
    ctx_.check(ssx_ba_window_export(win_, kf_ids, nullptr, nullptr, lm_ids, nullptr, point_fixed, edge_pose, edge_point, edge_uv, nullptr));
  }
  void Solve(ssx_ba_result& res) override { ctx_.check(ssx_ba_window_solve(win_, &res)); }

 private:
  ssx::Context& ctx_;
  ssx_ba_window* win_ = nullptr;
};

class SsxCompute final : public Compute {
 public:
  explicit SsxCompute(int device) : frame_(device), chain_(device), backend_(device) {}

  void Detect(const Image& img, const uint8_t* mask, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) override
  {
    // Detect is single-level: <= max(N + 3, 4 * nIni <= 256) keypoints by the bound of ssx.h; should a grid ever return
    // more, the call reports the size it needs (SSX_ERR_CAPACITY, n > capacity) and is repeated once with that size
    kps.assign((size_t)prm.nfeatures + 260 + 64, ssx_keypoint{});
    int32_t n = 0;
    ssx_status st = ssx_orb_detect(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, mask, img.cols, &prm, (int32_t)kps.size(),
                                   kps.data(), &n);
    if (st == SSX_ERR_CAPACITY && n > (int32_t)kps.size()) {
      kps.assign((size_t)n, ssx_keypoint{});
      st = ssx_orb_detect(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, mask, img.cols, &prm, (int32_t)kps.size(), kps.data(), &n);
    }
    frame_.check(st);
    kps.resize(n);
  }

  void DetectBoxes(const Image& img, const std::vector<int32_t>& boxes, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) override
  {
    kps.assign((size_t)prm.nfeatures + 260 + 64, ssx_keypoint{});
    int32_t n = 0;
    const int32_t nb = (int32_t)(boxes.size() / 4);
    ssx_status st = ssx_orb_detect_boxes(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, boxes.data(), nb, &prm, (int32_t)kps.size(), kps.data(), &n);
    if (st == SSX_ERR_CAPACITY && n > (int32_t)kps.size()) {
      kps.assign((size_t)n, ssx_keypoint{});
      st = ssx_orb_detect_boxes(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, boxes.data(), nb, &prm, (int32_t)kps.size(), kps.data(), &n);
    }
    frame_.check(st);
    kps.resize(n);
  }

  void TrackLK(const Image& prev, const Image& next, const std::vector<float>& prev_pts, std::vector<float>& next_pts,
               std::vector<uint8_t>& status, bool temporal) override
  {
    const int n = (int)(prev_pts.size() / 2);
    status.assign(n, 0);
    ssx_lk_params p;
    ssx_lk_default_params(&p);
    p.win = 11; p.max_level = 3; p.max_iters = 30; p.eps = 0.01; p.use_initial_flow = 1;
    if (!temporal) {
      frame_.check(ssx_lk_track(frame_.get(), prev.ptr(), prev.cols, next.ptr(), next.cols, prev.rows, prev.cols, n, prev_pts.data(),
                                next_pts.data(), status.data(), nullptr, &p, nullptr));
      return;
    }
    // consecutive frames: the pyramid of `prev` is still on the device when it was the `next` image of the last call
    if (prev.id != 0 && prev.id == chain_next_id_ && prev.rows == chain_rows_ && prev.cols == chain_cols_) {
      chain_.check(ssx_lk_track_next(chain_.get(), next.ptr(), next.cols, next.rows, next.cols, n, prev_pts.data(), next_pts.data(),
                                     status.data(), nullptr, &p, nullptr));
    } else {
      chain_.check(ssx_lk_track(chain_.get(), prev.ptr(), prev.cols, next.ptr(), next.cols, prev.rows, prev.cols, n, prev_pts.data(),
                                next_pts.data(), status.data(), nullptr, &p, nullptr));
    }
    chain_next_id_ = next.id; chain_rows_ = next.rows; chain_cols_ = next.cols;
  }

  int PoseOnly(double* pose_io, const double* K4, int M, const double* xyz, const double* uv, uint8_t* inlier) override
  {
    int32_t n_in = 0;
    frame_.check(ssx_pose_only_opt(frame_.get(), pose_io, K4, M, xyz, uv, 4, 10, 5.991, 1.0, inlier, &n_in));
    return n_in;
  }

  void Triangulate(int n, const double* uvL, const double* uvR, const ssx_stereo_rig& rig, const double* T_wc, double* xyz,
                   uint8_t* ok) override
  {
    frame_.check(ssx_triangulate(frame_.get(), n, uvL, uvR, &rig, T_wc, xyz, ok));
  }

  void BundleAdjust(const ssx_ba_problem& prob, const ssx_ba_options& opt, ssx_ba_result& res) override
  {
    backend_.check(ssx_ba_solve(backend_.get(), &prob, &opt, &res));
  }

  std::unique_ptr<BaWindow> MakeBaWindow(const double* K4, const double* cam_ext14, const ssx_ba_options& opt) override
  {
    return std::make_unique<SsxBaWindow>(backend_, K4, cam_ext14, opt);    // (the window must be destroyed before this Compute)
  }

 private:
  ssx::Context frame_, chain_, backend_;
  uint64_t chain_next_id_ = 0;
  int chain_rows_ = 0, chain_cols_ = 0;
};

}  // namespace

std::unique_ptr<Compute> MakeSsxCompute(int device) { return std::make_unique<SsxCompute>(device); }

}  // namespace ssx::host
