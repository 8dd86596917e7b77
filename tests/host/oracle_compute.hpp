// tests/host/oracle_compute.hpp -- TEST INFRASTRUCTURE: the host layer's Compute interface on the CPU oracle
// (oracle/liboracle.so), so the state machine of ssvio_amd/host can be exercised without a GPU and whole trajectories
// of the GPU library can be compared with it.  Never part of the product: only tests/ builds this file.
#pragma once
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../oracle/src/oracle.h"
#include "../../ssvio_amd/host/compute.hpp"

namespace ssx::host {

class OracleCompute final : public Compute {
 public:
  void Detect(const Image& img, const uint8_t* mask, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) override
  {
    static_assert(sizeof(orc_keypoint) == sizeof(ssx_keypoint), "keypoint layouts");
    orc_orb_params p{prm.nfeatures, prm.scale_factor, prm.nlevels, prm.ini_th_fast, prm.min_th_fast};
    kps.assign((size_t)prm.nfeatures + 260 + 64, ssx_keypoint{});
    const int n = orc_orb_detect(img.ptr(), img.cols, img.rows, img.cols, mask, img.cols, &p, (int)kps.size(),
                                 reinterpret_cast<orc_keypoint*>(kps.data()));
    if (n < 0) throw std::runtime_error("orc_orb_detect failed");
    kps.resize(n);
  }
  void TrackLK(const Image& prev, const Image& next, const std::vector<float>& prev_pts, std::vector<float>& next_pts,
               std::vector<uint8_t>& status, bool) override
  {
    const int n = (int)(prev_pts.size() / 2);
    status.assign(n, 0);
    orc_lk_params p;
    orc_lk_default_params(&p);
    p.win = 11; p.max_level = 3; p.max_iters = 30; p.eps = 0.01; p.use_initial_flow = 1;
    orc_lk_track(prev.ptr(), prev.cols, next.ptr(), next.cols, prev.rows, prev.cols, n, prev_pts.data(), next_pts.data(), status.data(),
                 nullptr, &p);
  }
  int PoseOnly(double* pose_io, const double* K4, int M, const double* xyz, const double* uv, uint8_t* inlier) override
  {
    return orc_pose_only(pose_io, K4, M, xyz, uv, 4, 10, 5.991, 1.0, inlier);
  }
  void Triangulate(int n, const double* uvL, const double* uvR, const ssx_stereo_rig& rig, const double* T_wc, double* xyz,
                   uint8_t* ok) override
  {
    orc_triangulate(n, uvL, uvR, rig.fx, rig.fy, rig.cx, rig.cy, rig.baseline, T_wc, xyz, ok, nullptr);
  }
  void BundleAdjust(const ssx_ba_problem& pr, const ssx_ba_options& opt, ssx_ba_result& res) override
  {
    std::vector<double> poses(pr.poses, pr.poses + 7 * (size_t)pr.P), points(pr.points, pr.points + 3 * (size_t)pr.L);
    orc_ba_options o{opt.outer_rounds, opt.iters, opt.chi2_th, opt.huber_delta, opt.inlier_ratio, opt.jac_mode};
    int stats_n = 0;
    std::vector<double> chi2(SSX_BA_MAX_STATS), lambda(SSX_BA_MAX_STATS);
    std::vector<int> trials(SSX_BA_MAX_STATS);
    const int rounds = orc_ba_solve(pr.P, poses.data(), pr.pose_fixed, pr.L, points.data(), pr.point_fixed, pr.E, pr.edge_pose,
                                    pr.edge_point, pr.edge_uv, pr.edge_cam, pr.K, pr.cam_ext, &o, res.edge_chi2, res.edge_outlier,
                                    SSX_BA_MAX_STATS, &stats_n, chi2.data(), lambda.data(), trials.data());
    if (rounds < 0) throw std::runtime_error("orc_ba_solve failed");
    res.rounds = rounds; res.n_iters = stats_n;
    if (res.poses_out) std::memcpy(res.poses_out, poses.data(), poses.size() * sizeof(double));
    if (res.points_out) std::memcpy(res.points_out, points.data(), points.size() * sizeof(double));
  }
};

}  // namespace ssx::host
