/* oracle/src/se3_oracle.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * SE3 arithmetic with Sophus semantics, shared by the BA and the pose-graph restatements.  Included INSIDE an
 * anonymous namespace by each translation unit.  pose = (qx,qy,qz,qw, tx,ty,tz). */
// ---------------------------------------------------------------------------------------------
// SE3 with Sophus semantics.  pose = (qx,qy,qz,qw, tx,ty,tz)
// ---------------------------------------------------------------------------------------------
// thirdparty/sophus/sophus/so3.hpp:352-361  (q * p): uv = qv x p; uv += uv; p + w*uv + qv x uv
inline void quat_rotate(const double* q, const double* p, double* out)
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  double uvx = y * p[2] - z * p[1];
  double uvy = z * p[0] - x * p[2];
  double uvz = x * p[1] - y * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  out[0] = p[0] + w * uvx + (y * uvz - z * uvy);
  out[1] = p[1] + w * uvy + (z * uvx - x * uvz);
  out[2] = p[2] + w * uvz + (x * uvy - y * uvx);
}

inline void se3_act(const double* T, const double* p, double* out)
{
  double r[3];
  quat_rotate(T, p, r);
  out[0] = r[0] + T[4]; out[1] = r[1] + T[5]; out[2] = r[2] + T[6];
}

// Eigen::Quaternion::toRotationMatrix
inline void quat_to_R(const double* q, double R[9])
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// so3.hpp:593-622 expAndTheta, epsilon = 1e-10 (common.hpp:110-111)
inline void so3_exp(const double* om, double* q, double* theta)
{
  const double eps = 1e-10;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double imag, real;
  if (theta_sq < eps * eps) {
    *theta = 0;
    const double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    *theta = std::sqrt(theta_sq);
    const double half = 0.5 * (*theta);
    imag = std::sin(half) / (*theta);
    real = std::cos(half);
  }
  q[0] = imag * om[0]; q[1] = imag * om[1]; q[2] = imag * om[2]; q[3] = real;
}

// se3.hpp:763-784
inline void se3_exp(const double* a, double* T)
{
  const double eps = 1e-10;
  const double* om = a + 3;
  double theta;
  so3_exp(om, T, &theta);
  // Omega = hat(omega), Omega_sq = Omega*Omega
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += O[r * 3 + k] * O[k * 3 + c];
      O2[r * 3 + c] = s;
    }
  double V[9];
  if (theta < eps) {
    quat_to_R(T, V);
  } else {
    const double theta_sq = theta * theta;
    const double c1 = (1.0 - std::cos(theta)) / theta_sq;
    const double c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int r = 0; r < 3; ++r) T[4 + r] = V[r * 3] * a[0] + V[r * 3 + 1] * a[1] + V[r * 3 + 2] * a[2];
}

// so3.hpp:322-334 (product) + :502 (normalize in the quaternion ctor), se3.hpp:308-312
inline void se3_mul(const double* A, const double* B, double* out)
{
  const double ax = A[0], ay = A[1], az = A[2], aw = A[3];
  const double bx = B[0], by = B[1], bz = B[2], bw = B[3];
  double w = aw * bw - ax * bx - ay * by - az * bz;
  double x = aw * bx + ax * bw + ay * bz - az * by;
  double y = aw * by + ay * bw + az * bx - ax * bz;
  double z = aw * bz + az * bw + ax * by - ay * bx;
  const double len = std::sqrt(x * x + y * y + z * z + w * w);
  double r[3];
  quat_rotate(A, B + 4, r);
  out[0] = x / len; out[1] = y / len; out[2] = z / len; out[3] = w / len;
  out[4] = A[4] + r[0]; out[5] = A[5] + r[1]; out[6] = A[6] + r[2];
}

// VertexPose::oplusImpl, include/ssvio/g2otypes.hpp:36-41: T <- exp(delta) * T
inline void pose_oplus(const double* T, const double* d, double* out)
{
  double ex[7];
  se3_exp(d, ex);
  se3_mul(ex, T, out);
}
