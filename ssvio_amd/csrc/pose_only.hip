// ssvio_amd/csrc/pose_only.hip -- per-frame pose-only robust optimisation on gfx950 (ssx_pose_only_opt).
//
// Replaces the optimisation core of FrontEnd::EstimateCurrentPose (/root/reference/src/ssvio/frontend.cpp:184-270):
// one VertexPose, one EdgeProjectionPoseOnly per tracked map point (analytic 2x6 Jacobian,
// include/ssvio/g2otypes.hpp:67-110), Huber kernel (delta 1.0 = g2o's default ctor), LinearSolverDense, LM;
// 4 rounds x optimize(10) with chi2 > 5.991 => outlier (level 1) after every round and the kernels removed before
// the last round (frontend.cpp:236-269).
//
// The problem is tiny (50-300 edges, 6 unknowns) and strictly sequential across LM trials, so the WHOLE procedure --
// all rounds, iterations and LM trials -- runs inside ONE launch of ONE 256-thread workgroup: no host round trip,
// no second kernel.  Edges are strided over the threads; the 6x6 normal equations (21 + 6 + chi2 values) are reduced
// with one fixed-shape LDS tree per linearisation (deterministic); thread 0 runs the 6x6 Cholesky and the LM
// bookkeeping of OptimizationAlgorithmLevenberg::solve (thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:58-175).
#include <cmath>

#include "ctx.hpp"
#include "se3.hpp"

namespace {

constexpr int PT = 256;
constexpr int NRED = 28;   // 21 (upper 6x6) + 6 (b) + 1 (chi2)

struct PoDev {
  int M, rounds, iters;
  double chi2_th, huber_delta;
  ssx::Cam K;
  const double* xyz;   // M x 3
  const double* uv;    // M x 2
  double* err;         // M x 2 (last computed error of each edge, like g2o's _error)
  uint8_t* level;      // M: 1 = outlier level (not optimised)
  uint8_t* outlier;    // M: features[i]->is_outlier_
  double* pose;        // 7 in/out
  int* n_inliers;
};

__device__ __forceinline__ void po_error(const double* T, const PoDev& d, int i, double* e, double* pc)
{
  const double X[3] = {d.xyz[3 * i], d.xyz[3 * i + 1], d.xyz[3 * i + 2]};
  ssx::se3_act(T, X, pc);
  const double hx = d.K.fx * pc[0] + d.K.cx * pc[2];
  const double hy = d.K.fy * pc[1] + d.K.cy * pc[2];
  e[0] = d.uv[2 * i] - hx / pc[2];
  e[1] = d.uv[2 * i + 1] - hy / pc[2];
}

// all threads: reduce `n` values per thread (column-major sRed[k][t]) with one tree; result in sRed[k][0]
__device__ __forceinline__ void tree_reduce(double (*sRed)[PT], int n)
{
  const int t = threadIdx.x;
  __syncthreads();
  for (int o = PT / 2; o > 0; o >>= 1) {
    if (t < o)
      for (int k = 0; k < n; ++k) sRed[k][t] += sRed[k][t + o];
    __syncthreads();
  }
}

__global__ __launch_bounds__(PT) void k_pose_only(PoDev d)
{
  __shared__ double sRed[NRED][PT];
  __shared__ double sT[7], sTbak[7], sX[6];
  __shared__ double sCtl[8];      // 0 lambda, 1 ni, 2 currentChi, 3 rho, 4 qmax, 5 stop flag, 6 accepted
  __shared__ int sUseKernel;
  const int t = threadIdx.x;
  if (t < 7) sT[t] = d.pose[t];
  if (t == 0) sUseKernel = 1;
  for (int i = t; i < d.M; i += PT) { d.level[i] = 0; d.outlier[i] = 0; d.err[2 * i] = 0.0; d.err[2 * i + 1] = 0.0; }
  __syncthreads();

  // errors of the active (level 0) edges at the current pose + robust chi2 -> sRed[27][0]
  auto errors_and_chi2 = [&]() {
    double T[7];
    for (int k = 0; k < 7; ++k) T[k] = sT[k];
    double chi = 0.0;
    for (int i = t; i < d.M; i += PT) {
      if (d.level[i]) continue;
      double e[2], pc[3];
      po_error(T, d, i, e, pc);
      d.err[2 * i] = e[0]; d.err[2 * i + 1] = e[1];
      const double c2 = e[0] * e[0] + e[1] * e[1];
      double r0 = c2, w = 1.0;
      if (sUseKernel) ssx::huber(c2, d.huber_delta, r0, w);
      chi += r0;
    }
    sRed[27][t] = chi;
    tree_reduce(sRed + 27, 1);
    return sRed[27][0];
  };

  int cnt_outliers = 0;
  for (int round = 0; round < d.rounds; ++round) {
    // initializeOptimization(0): only level-0 edges are active
    int mine = 0;
    for (int i = t; i < d.M; i += PT) mine += !d.level[i];
    sRed[0][t] = (double)mine;
    tree_reduce(sRed, 1);
    const int n_active = (int)sRed[0][0];
    __syncthreads();
    for (int it = 0; it < d.iters && n_active > 0; ++it) {
      const double chi_now = errors_and_chi2();
      __syncthreads();
      // ---- linearise: H (upper 21), b (6) ----
      {
        double T[7];
        for (int k = 0; k < 7; ++k) T[k] = sT[k];
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
        for (int i = t; i < d.M; i += PT) {
          if (d.level[i]) continue;
          const double Xw[3] = {d.xyz[3 * i], d.xyz[3 * i + 1], d.xyz[3 * i + 2]};
          double pc[3];
          ssx::se3_act(T, Xw, pc);
          const double X = pc[0], Y = pc[1], Z = pc[2];
          const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;
          // EdgeProjectionPoseOnly::linearizeOplus, g2otypes.hpp:86-101
          const double J[12] = {-d.K.fx * Zinv, 0, d.K.fx * X * Zinv2, d.K.fx * X * Y * Zinv2, -d.K.fx - d.K.fx * X * X * Zinv2,
                                d.K.fx * Y * Zinv, 0, -d.K.fy * Zinv, d.K.fy * Y * Zinv2, d.K.fy + d.K.fy * Y * Y * Zinv2,
                                -d.K.fy * X * Y * Zinv2, -d.K.fy * X * Zinv};
          const double e0 = d.err[2 * i], e1 = d.err[2 * i + 1];
          double r0, w = 1.0;
          if (sUseKernel) ssx::huber(e0 * e0 + e1 * e1, d.huber_delta, r0, w);
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = a; c < 6; ++c) acc[q++] += J[a] * w * J[c] + J[6 + a] * w * J[6 + c];
          }
#pragma unroll
          for (int a = 0; a < 6; ++a) acc[21 + a] -= w * (J[a] * e0 + J[6 + a] * e1);
        }
#pragma unroll
        for (int k = 0; k < 27; ++k) sRed[k][t] = acc[k];
        tree_reduce(sRed, 27);
      }
      if (t == 0) {
        if (it == 0) {
          // computeLambdaInit: 1e-5 * max |diag(H)|; diagonal entries of the upper layout: 0,6,11,15,18,20
          const int dg[6] = {0, 6, 11, 15, 18, 20};
          double m = 0.0;
          for (int k = 0; k < 6; ++k) m = fmax(m, fabs(sRed[dg[k]][0]));
          sCtl[0] = 1e-5 * m; sCtl[1] = 2.0;
        }
        sCtl[2] = chi_now; sCtl[4] = 0.0; sCtl[5] = 0.0;
      }
      __syncthreads();
      // ---- LM trials ----
      while (true) {
        if (t == 0) {
          for (int k = 0; k < 7; ++k) sTbak[k] = sT[k];
          // (H + lambda I) x = b by Cholesky (LinearSolverDense: Eigen LDLT, false when not positive)
          double A[36];
          int q = 0;
          for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) { const double v = sRed[q++][0]; A[a * 6 + c] = v; A[c * 6 + a] = v; }
          for (int a = 0; a < 6; ++a) A[a * 6 + a] += sCtl[0];
          bool ok = true;
          for (int j = 0; j < 6 && ok; ++j) {
            double dj = A[j * 6 + j];
            for (int k = 0; k < j; ++k) dj -= A[j * 6 + k] * A[j * 6 + k];
            if (!(dj > 0.0) || !isfinite(dj)) { ok = false; break; }
            dj = sqrt(dj);
            A[j * 6 + j] = dj;
            for (int i = j + 1; i < 6; ++i) {
              double s = A[i * 6 + j];
              for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
              A[i * 6 + j] = s / dj;
            }
          }
          double y[6], x[6] = {0, 0, 0, 0, 0, 0};
          if (ok) {
            for (int i = 0; i < 6; ++i) {
              double s = sRed[21 + i][0];
              for (int k = 0; k < i; ++k) s -= A[i * 6 + k] * y[k];
              y[i] = s / A[i * 6 + i];
            }
            for (int i = 5; i >= 0; --i) {
              double s = y[i];
              for (int k = i + 1; k < 6; ++k) s -= A[k * 6 + i] * x[k];
              x[i] = s / A[i * 6 + i];
            }
          }
          for (int k = 0; k < 6; ++k) sX[k] = x[k];
          double T[7], out[7];
          for (int k = 0; k < 7; ++k) T[k] = sT[k];
          ssx::pose_oplus(T, x, out);
          for (int k = 0; k < 7; ++k) sT[k] = out[k];
          sCtl[6] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        // the H/b in sRed[0..26][0] must survive the chi2 reduction: it only touches row 27
        double tempChi = errors_and_chi2();
        if (t == 0) {
          if (sCtl[6] == 0.0) tempChi = 1.7976931348623157e308;
          double rho = sCtl[2] - tempChi;
          double scale = 0.0;
          for (int j = 0; j < 6; ++j) scale += sX[j] * (sCtl[0] * sX[j] + sRed[21 + j][0]);
          scale += 1e-3;
          rho /= scale;
          bool lambda_bad = false;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow((2 * rho - 1), 3.0);
            alpha = fmin(alpha, 2. / 3.);
            sCtl[0] *= fmax(1. / 3., alpha);
            sCtl[1] = 2.0;
            sCtl[2] = tempChi;
          } else {
            sCtl[0] *= sCtl[1];
            sCtl[1] *= 2.0;
            for (int k = 0; k < 7; ++k) sT[k] = sTbak[k];     // pop(): vertices only, errors stay at the trial state
            if (!isfinite(sCtl[0])) lambda_bad = true;
          }
          sCtl[4] += lambda_bad ? 0.0 : 1.0;
          const double qmax = sCtl[4];
          // do { } while (rho < 0 && qmax < 10);  then Terminate if qmax == 10 || rho == 0 || lambda non-finite
          const bool again = !lambda_bad && (rho < 0) && (qmax < 10.0);
          sCtl[3] = again ? 1.0 : 0.0;
          sCtl[5] = (qmax == 10.0 || rho == 0 || lambda_bad) ? 1.0 : 0.0;
        }
        __syncthreads();
        if (sCtl[3] == 0.0) break;
      }
      if (sCtl[5] != 0.0) break;   // Terminate: optimize() stops iterating
      __syncthreads();
    }
    __syncthreads();
    // frontend.cpp:243-268: recompute the error only for features flagged outlier, classify, set levels
    {
      double T[7];
      for (int k = 0; k < 7; ++k) T[k] = sT[k];
      int co = 0;
      for (int i = t; i < d.M; i += PT) {
        if (d.outlier[i]) {
          double e[2], pc[3];
          po_error(T, d, i, e, pc);
          d.err[2 * i] = e[0]; d.err[2 * i + 1] = e[1];
        }
        const double c2 = d.err[2 * i] * d.err[2 * i] + d.err[2 * i + 1] * d.err[2 * i + 1];
        if (c2 > d.chi2_th) { d.outlier[i] = 1; d.level[i] = 1; ++co; }
        else { d.outlier[i] = 0; d.level[i] = 0; }
      }
      sRed[0][t] = (double)co;
      tree_reduce(sRed, 1);
      cnt_outliers = (int)sRed[0][0];
      __syncthreads();
      if (t == 0 && round == d.rounds - 2) sUseKernel = 0;   // e->setRobustKernel(nullptr)
      __syncthreads();
    }
  }
  if (t < 7) d.pose[t] = sT[t];
  if (t == 0) *d.n_inliers = d.M - cnt_outliers;
}

}  // namespace

struct PoWorkspace { DevBuf arena; HostBuf stage; };

extern "C" ssx_status ssx_pose_only_opt(ssx_ctx* ctx, double* pose_io, const double* K4, int32_t M, const double* xyz,
                                        const double* uv, int32_t rounds, int32_t iters, double chi2_th, double huber_delta,
                                        uint8_t* inlier_out, int32_t* n_inliers)
{
  if (!ctx || !pose_io || !K4 || M < 0 || (M && (!xyz || !uv)) || rounds < 0 || iters < 0) return SSX_ERR_INVALID_ARG;
  if (M == 0) { if (n_inliers) *n_inliers = 0; return SSX_OK; }
  static thread_local int dummy = 0; (void)dummy;
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // reuse the BA-independent scratch: a small private arena hung off the ctx through the generic slot
  static_assert(sizeof(double) == 8, "");
  DevBuf& arena = ctx->po_arena;
  HostBuf& stage = ctx->po_stage;
  Layout lay;
  const size_t o_xyz = lay.take(sizeof(double) * 3 * (size_t)M);
  const size_t o_uv = lay.take(sizeof(double) * 2 * (size_t)M);
  const size_t o_pose = lay.take(sizeof(double) * 8);
  const size_t in_bytes = lay.off;
  const size_t o_err = lay.take(sizeof(double) * 2 * (size_t)M);
  const size_t o_level = lay.take((size_t)M);
  const size_t o_out = lay.take((size_t)M);
  const size_t o_n = lay.take(sizeof(int) * 2);
  SSX_HIP_TRY(ctx, arena.reserve(lay.off));
  SSX_HIP_TRY(ctx, stage.reserve(lay.off));
  char* hs = stage.as<char>();
  memcpy(hs + o_xyz, xyz, sizeof(double) * 3 * M);
  memcpy(hs + o_uv, uv, sizeof(double) * 2 * M);
  memcpy(hs + o_pose, pose_io, sizeof(double) * 7);
  char* base = arena.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  PoDev d;
  d.M = M; d.rounds = rounds; d.iters = iters; d.chi2_th = chi2_th; d.huber_delta = huber_delta;
  d.K = ssx::Cam{K4[0], K4[1], K4[2], K4[3]};
  d.xyz = (const double*)(base + o_xyz); d.uv = (const double*)(base + o_uv);
  d.err = (double*)(base + o_err); d.level = (uint8_t*)(base + o_level); d.outlier = (uint8_t*)(base + o_out);
  d.pose = (double*)(base + o_pose); d.n_inliers = (int*)(base + o_n);
  SSX_PROF(ctx, KID_POSE_ONLY, hipLaunchKernelGGL(k_pose_only, dim3(1), dim3(PT), 0, ctx->stream, d));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_pose, base + o_pose, sizeof(double) * 7, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_out, base + o_out, M, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_n, base + o_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(pose_io, hs + o_pose, sizeof(double) * 7);
  if (inlier_out) for (int i = 0; i < M; ++i) inlier_out[i] = !reinterpret_cast<uint8_t*>(hs + o_out)[i];
  if (n_inliers) *n_inliers = *reinterpret_cast<int*>(hs + o_n);
  return SSX_OK;
}
