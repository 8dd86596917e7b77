// ssvio_amd/host/se3.hpp -- the rigid-body type of the host layer: Sophus::SE3d as the reference's Frame / KeyFrame /
// Camera use it (unit quaternion x y z w + translation, 7 doubles in Sophus::SE3d::data() order, which is also the
// pose layout of ssx.h).  The group operations are the ones the kernels use (ssvio_amd/csrc/se3.hpp), compiled for
// the host.
#pragma once
#include <cmath>
#include <cstring>

#include "../csrc/se3.hpp"

namespace ssx::host {

struct SE3 {
  double d[7] = {0, 0, 0, 1, 0, 0, 0};

  SE3() = default;
  explicit SE3(const double* p7) { std::memcpy(d, p7, sizeof(d)); }
  static SE3 translation(double x, double y, double z)
  {
    SE3 T;
    T.d[4] = x; T.d[5] = y; T.d[6] = z;
    return T;
  }
  const double* data() const { return d; }
  double* data() { return d; }

  SE3 operator*(const SE3& o) const
  {
    SE3 r;
    ssx::se3_mul(d, o.d, r.d);
    return r;
  }
  SE3 inverse() const
  {
    SE3 r;
    ssx::se3_inverse(d, r.d);
    return r;
  }
  // T * p
  void act(const double* p, double* out) const { ssx::se3_act(d, p, out); }
  // |log(T)|: the 6-vector norm Map::RemoveOldActiveKeyframe compares (map.cpp:106)
  double log_norm() const
  {
    double v[6];
    ssx::se3_log(d, v);
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
  }
  // Eigen::Quaterniond(rotationMatrix()).coeffs() -- the quaternion the TUM writer prints (x y z w)
  void rotation_quaternion(double* q_xyzw) const
  {
    double R[9];
    ssx::quat_to_R(d, R);
    double t = R[0] + R[4] + R[8];
    double q[4];                                           // x y z w
    if (t > 0) {
      t = std::sqrt(t + 1.0);
      q[3] = 0.5 * t;
      t = 0.5 / t;
      q[0] = (R[7] - R[5]) * t;
      q[1] = (R[2] - R[6]) * t;
      q[2] = (R[3] - R[1]) * t;
    } else {
      int i = 0;
      if (R[4] > R[0]) i = 1;
      if (R[8] > R[i * 4]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
      q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
      q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    std::memcpy(q_xyzw, q, sizeof(q));
  }
};

}  // namespace ssx::host
