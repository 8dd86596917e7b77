// ssvio_amd/csrc/se3.hpp -- SE3 / projection math for the BA and pose-only kernels (f64, __host__ __device__).
//
// Semantics follow what the reference's vertices and edges compute (paths relative to /root/reference):
//   pose update  T <- exp(delta) * T        include/ssvio/g2otypes.hpp:36-41  (VertexPose::oplusImpl)
//   SE3 exp / SO3 exp / group product        thirdparty/sophus/sophus/se3.hpp:763-784, so3.hpp:593-622,322-334
//   residual     e = z - hnorm(K (ext (T p))) include/ssvio/g2otypes.hpp:123-131 (EdgeProjection::computeError)
//   Huber        rho(e2)                      thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-78
// pose = qx qy qz qw tx ty tz.
#pragma once
#include <hip/hip_runtime.h>

#define SSX_HD __host__ __device__ __forceinline__

namespace ssx {

struct Cam { double fx, fy, cx, cy; };

SSX_HD void quat_rotate(const double* q, const double* p, double* out)
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  double ux = y * p[2] - z * p[1];
  double uy = z * p[0] - x * p[2];
  double uz = x * p[1] - y * p[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = p[0] + w * ux + (y * uz - z * uy);
  out[1] = p[1] + w * uy + (z * ux - x * uz);
  out[2] = p[2] + w * uz + (x * uy - y * ux);
}

SSX_HD void se3_act(const double* T, const double* p, double* out)
{
  double r[3];
  quat_rotate(T, p, r);
  out[0] = r[0] + T[4]; out[1] = r[1] + T[5]; out[2] = r[2] + T[6];
}

SSX_HD void quat_to_R(const double* q, double* R)
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

// exp of se(3): a = (upsilon, omega)
SSX_HD void se3_exp(const double* a, double* T)
{
  const double eps = 1e-10;
  const double ox = a[3], oy = a[4], oz = a[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  double imag, real, theta, sh = 0, ch = 1;
  if (theta_sq < eps * eps) {
    theta = 0;
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
  } else {
    theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    sincos(half, &sh, &ch);
    imag = sh / theta;
    real = ch;
  }
  T[0] = imag * ox; T[1] = imag * oy; T[2] = imag * oz; T[3] = real;
  // V = I + c1 * Omega + c2 * Omega^2   (V = R when theta < eps)
  double V[9];
  if (theta < eps) {
    quat_to_R(T, V);
  } else {
    // 1 - cos(theta) = 2 sin^2(theta/2), sin(theta) = 2 sin(theta/2) cos(theta/2): two trigonometric calls instead of
    // four on the serial path of every LM trial (and no cancellation in c1)
    const double c1 = 2.0 * sh * sh / theta_sq;
    const double c2 = (theta - 2.0 * sh * ch) / (theta_sq * theta);
    // Omega^2 = omega omega^T - theta^2 I
    V[0] = 1.0 + c2 * (-(oy * oy + oz * oz));
    V[1] = c1 * (-oz) + c2 * (ox * oy);
    V[2] = c1 * (oy) + c2 * (ox * oz);
    V[3] = c1 * (oz) + c2 * (ox * oy);
    V[4] = 1.0 + c2 * (-(ox * ox + oz * oz));
    V[5] = c1 * (-ox) + c2 * (oy * oz);
    V[6] = c1 * (-oy) + c2 * (ox * oz);
    V[7] = c1 * (ox) + c2 * (oy * oz);
    V[8] = 1.0 + c2 * (-(ox * ox + oy * oy));
  }
  T[4] = V[0] * a[0] + V[1] * a[1] + V[2] * a[2];
  T[5] = V[3] * a[0] + V[4] * a[1] + V[5] * a[2];
  T[6] = V[6] * a[0] + V[7] * a[1] + V[8] * a[2];
}

// out = A * B with quaternion re-normalisation (Sophus SO3 ctor from quaternion)
SSX_HD void se3_mul(const double* A, const double* B, double* out)
{
  const double ax = A[0], ay = A[1], az = A[2], aw = A[3];
  const double bx = B[0], by = B[1], bz = B[2], bw = B[3];
  const double w = aw * bw - ax * bx - ay * by - az * bz;
  const double x = aw * bx + ax * bw + ay * bz - az * by;
  const double y = aw * by + ay * bw + az * bx - ax * bz;
  const double z = aw * bz + az * bw + ax * by - ay * bx;
  const double len = sqrt(x * x + y * y + z * z + w * w);
  double r[3];
  quat_rotate(A, B + 4, r);
  out[0] = x / len; out[1] = y / len; out[2] = z / len; out[3] = w / len;
  out[4] = A[4] + r[0]; out[5] = A[5] + r[1]; out[6] = A[6] + r[2];
}

SSX_HD void pose_oplus(const double* T, const double* d, double* out)
{
  double ex[7];
  se3_exp(d, ex);
  se3_mul(ex, T, out);
}

// Sophus SE3::inverse (se3.hpp:208-211): the SO3 constructor re-normalises the conjugate quaternion
SSX_HD void se3_inverse(const double* T, double* out)
{
  double q[4] = {-T[0], -T[1], -T[2], T[3]};
  const double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= len; q[1] /= len; q[2] /= len; q[3] /= len;
  const double nt[3] = {T[4] * -1.0, T[5] * -1.0, T[6] * -1.0};
  double r[3];
  quat_rotate(q, nt, r);
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = r[0]; out[5] = r[1]; out[6] = r[2];
}

// Sophus SE3::log (se3.hpp:223-256) with SO3::logAndTheta (so3.hpp:245-286); tangent = (upsilon, omega)
SSX_HD void se3_log(const double* T, double* out)
{
  const double eps = 1e-10;
  const double squared_n = T[0] * T[0] + T[1] * T[1] + T[2] * T[2];
  const double w = T[3];
  double k, theta;                       // k = 2 atan(n / w) / n
  if (squared_n < eps * eps) {
    k = 2.0 / w - (2.0 / 3.0) * (squared_n) / (w * (w * w));
    theta = 2.0 * squared_n / w;
  } else {
    const double n = sqrt(squared_n);
    if (fabs(w) < eps) k = (w > 0.0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
    else k = 2.0 * atan(n / w) / n;
    theta = k * n;
  }
  const double ox = k * T[0], oy = k * T[1], oz = k * T[2];
  double c;
  if (fabs(theta) < eps) c = 1. / 12.;
  else {
    const double half_theta = 0.5 * theta;
    c = (1.0 - theta * cos(half_theta) / (2.0 * sin(half_theta))) / (theta * theta);
  }
  // V^-1 = I - 0.5 Omega + c Omega^2,  Omega^2 = omega omega^T - |omega|^2 I
  const double o2 = ox * ox + oy * oy + oz * oz;
  const double t0 = T[4], t1 = T[5], t2 = T[6];
  const double od = ox * t0 + oy * t1 + oz * t2;
  out[0] = t0 - 0.5 * (oy * t2 - oz * t1) + c * (ox * od - o2 * t0);
  out[1] = t1 - 0.5 * (oz * t0 - ox * t2) + c * (oy * od - o2 * t1);
  out[2] = t2 - 0.5 * (ox * t1 - oy * t0) + c * (oz * od - o2 * t2);
  out[3] = ox; out[4] = oy; out[5] = oz;
}

// EdgePoseGraph::computeError (g2otypes.hpp:169-176): e = log(M^-1 * T0 * T1^-1)
SSX_HD void pg_error(const double* M, const double* T0, const double* T1, double* e)
{
  double Mi[7], T1i[7], A[7], B[7];
  se3_inverse(M, Mi);
  se3_inverse(T1, T1i);
  se3_mul(Mi, T0, A);
  se3_mul(A, T1i, B);
  se3_log(B, e);
}

// g2o's central differences on the oplus of vertex `which` (base_binary_edge.hpp:61-141), delta = 1e-9; J 6x6 row-major
SSX_HD void pg_jac_numeric(const double* M, const double* T0, const double* T1, int which, double* J)
{
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  for (int d = 0; d < 6; ++d) {
    double add[6] = {0, 0, 0, 0, 0, 0}, Tp[7], Tm[7], ep[6], em[6];
    add[d] = delta;
    pose_oplus(which == 0 ? T0 : T1, add, Tp);
    add[d] = -delta;
    pose_oplus(which == 0 ? T0 : T1, add, Tm);
    if (which == 0) { pg_error(M, Tp, T1, ep); pg_error(M, Tm, T1, em); }
    else { pg_error(M, T0, Tp, ep); pg_error(M, T0, Tm, em); }
    for (int r = 0; r < 6; ++r) J[r * 6 + d] = scalar * (ep[r] - em[r]);
  }
}

// e = uv - hnorm(K * (ext * (T * p)));   p1 = T*p and pc = ext*p1 are returned for the Jacobians
SSX_HD void edge_error(const double* T, const double* p, const double* ext, const Cam& K,
                       double u, double v, double* e, double* p1, double* pc)
{
  se3_act(T, p, p1);
  se3_act(ext, p1, pc);
  const double hx = K.fx * pc[0] + K.cx * pc[2];
  const double hy = K.fy * pc[1] + K.cy * pc[2];
  e[0] = u - hx / pc[2];
  e[1] = v - hy / pc[2];
}

// analytic Jacobians (the formula commented out at g2otypes.hpp:133-153, generalised to ext != I):
//   Ji (2x6) = A * R_ext * [ I | -[p1]x ],  Jj (2x3) = A * R_ext * R_T,  A = d e / d pc
// (Re = quat_to_R(ext), row-major: the rotation of the camera extrinsic is the same for every edge of a camera -- callers that
// linearise thousands of edges keep it, k_linearize measured no gain from it: profiles/r05/lin_schur_ab.md)
SSX_HD void edge_jac_analytic_R(const double* T, const double* Re, const Cam& K, const double* p1,
                                const double* pc, double* Ji, double* Jj)
{
  const double X = pc[0], Y = pc[1], Z = pc[2];
  const double Zinv = 1.0 / (Z + 1e-18);
  const double Zinv2 = Zinv * Zinv;
  const double A[6] = {-K.fx * Zinv, 0.0, K.fx * X * Zinv2, 0.0, -K.fy * Zinv, K.fy * Y * Zinv2};
  double Rt[9];
  quat_to_R(T, Rt);
  double AR[6];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      AR[r * 3 + c] = A[r * 3] * Re[c] + A[r * 3 + 1] * Re[3 + c] + A[r * 3 + 2] * Re[6 + c];
  const double H[9] = {0.0, p1[2], -p1[1], -p1[2], 0.0, p1[0], p1[1], -p1[0], 0.0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = AR[r * 3 + c];
      Ji[r * 6 + 3 + c] = AR[r * 3] * H[c] + AR[r * 3 + 1] * H[3 + c] + AR[r * 3 + 2] * H[6 + c];
      Jj[r * 3 + c] = AR[r * 3] * Rt[c] + AR[r * 3 + 1] * Rt[3 + c] + AR[r * 3 + 2] * Rt[6 + c];
    }
  }
}
SSX_HD void edge_jac_analytic(const double* T, const double* ext, const Cam& K, const double* p1,
                              const double* pc, double* Ji, double* Jj)
{
  double Re[9];
  quat_to_R(ext, Re);
  edge_jac_analytic_R(T, Re, K, p1, pc, Ji, Jj);
}

// g2o's numeric Jacobians: central differences, delta = 1e-9, through oplus
// (thirdparty/g2o/g2o/core/base_binary_edge.hpp:144-212)
SSX_HD void edge_jac_numeric(const double* T, const double* p, const double* ext, const Cam& K,
                             double u, double v, double* Ji, double* Jj)
{
  const double delta = 1e-9;
  const double scalar = 1.0 / (2 * delta);
  double p1[3], pc[3];
  for (int d = 0; d < 6; ++d) {
    double add[6] = {0, 0, 0, 0, 0, 0};
    double Tp[7], e1[2], e2[2];
    add[d] = delta;
    pose_oplus(T, add, Tp);
    edge_error(Tp, p, ext, K, u, v, e1, p1, pc);
    add[d] = -delta;
    pose_oplus(T, add, Tp);
    edge_error(Tp, p, ext, K, u, v, e2, p1, pc);
    Ji[d] = scalar * (e1[0] - e2[0]);
    Ji[6 + d] = scalar * (e1[1] - e2[1]);
  }
  for (int d = 0; d < 3; ++d) {
    double pp[3] = {p[0], p[1], p[2]}, e1[2], e2[2];
    pp[d] = p[d] + delta;
    edge_error(T, pp, ext, K, u, v, e1, p1, pc);
    pp[d] = p[d] + (-delta);
    edge_error(T, pp, ext, K, u, v, e2, p1, pc);
    Jj[d] = scalar * (e1[0] - e2[0]);
    Jj[3 + d] = scalar * (e1[1] - e2[1]);
  }
}

// Huber: rho0 = robustified cost, rho1 = weight
SSX_HD void huber(double e2, double delta, double& rho0, double& rho1)
{
  const double dsqr = delta * delta;
  if (e2 <= dsqr) {
    rho0 = e2; rho1 = 1.0;
  } else {
    const double s = sqrt(e2);
    rho0 = 2 * s * delta - dsqr;
    rho1 = delta / s;
  }
}

// inverse of a symmetric 3x3 given as (a00 a01 a02 a11 a12 a22) by cofactors -> full 9
SSX_HD void inv3_sym(const double* s, double* o)
{
  const double m0 = s[0], m1 = s[1], m2 = s[2], m4 = s[3], m5 = s[4], m8 = s[5];
  const double c00 = m4 * m8 - m5 * m5;
  const double c10 = m5 * m2 - m1 * m8;
  const double c20 = m1 * m5 - m4 * m2;
  const double det = m0 * c00 + m1 * c10 + m2 * c20;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c10 * id; o[2] = c20 * id;
  o[3] = o[1];     o[4] = (m0 * m8 - m2 * m2) * id; o[5] = (m2 * m1 - m0 * m5) * id;
  o[6] = o[2];     o[7] = o[5]; o[8] = (m0 * m4 - m1 * m1) * id;
}

}  // namespace ssx
