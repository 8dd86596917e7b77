// ssvio_amd/csrc/pose_only.hip -- per-frame pose-only robust optimisation on gfx950 (ssx_pose_only_opt).
//
// Replaces the optimisation core of FrontEnd::EstimateCurrentPose (/root/reference/src/ssvio/frontend.cpp:184-270):
// one VertexPose, one EdgeProjectionPoseOnly per tracked map point (analytic 2x6 Jacobian,
// include/ssvio/g2otypes.hpp:67-110), Huber kernel (delta 1.0 = g2o's default ctor), LinearSolverDense, LM;
// 4 rounds x optimize(10) with chi2 > 5.991 => outlier (level 1) after every round and the kernels removed before
// the last round (frontend.cpp:236-269).
//
// The problem is tiny (50-400 edges, 6 unknowns) and strictly sequential across LM trials, so the WHOLE procedure --
// all rounds, iterations and LM trials -- runs inside ONE launch of ONE 256-thread workgroup: no host round trip,
// no second kernel.  k_pose_only<EPT> (M <= 256 * EPT) keeps every edge (map point, pixel, last error, level) in the
// registers of its thread for the whole launch; the 6x6 normal equations (21 + 6 values) and every chi2 are summed by
// a wave butterfly + one LDS exchange between the 4 waves (ONE barrier per sum, double-buffered), and every thread
// then runs the 6x6 Cholesky, the pose update and the LM bookkeeping of OptimizationAlgorithmLevenberg::solve
// (thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:58-175) redundantly on identical values -- no
// broadcast, no second barrier.  k_pose_only_generic (any M) is the same procedure with the edges in global memory
// and an LDS tree reduction.
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
  double* pose;        // 7 in (the generic kernel also writes its result here)
  double* pose_out;    // 7 out
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

__global__ __launch_bounds__(PT) void k_pose_only_generic(PoDev d)
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
  if (t < 7) d.pose_out[t] = sT[t];
  if (t == 0) *d.n_inliers = d.M - cnt_outliers;
}

// ---- register-resident variant ------------------------------------------------------------------------------------
constexpr int NW = PT / 64;

// lane exchange inside a row of 16 lanes on the VALU (DPP) -- a shuffle through the LDS crossbar costs ~100 cycles of
// latency per step, and the 27-value reduction below is on the critical path of every LM iteration
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane_f64(double v, int lane)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over the 64 lanes, the same value in every lane
__device__ __forceinline__ double wave_sum_f64(double v)
{
  v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f64<0x141>(v);    // row_half_mirror: the other quad of the 8-lane half
  v += dpp_f64<0x140>(v);    // row_mirror: the other half of the 16-lane row
  return (read_lane_f64(v, 0) + read_lane_f64(v, 16)) + (read_lane_f64(v, 32) + read_lane_f64(v, 48));
}

// sum of N values per thread over the workgroup, returned to every thread.  buf alternates between calls, so a single
// barrier per call is enough: a thread can only reach the next call on the same buffer after every thread has passed
// the barrier of the call in between, i.e. has finished reading.
template <int N>
__device__ __forceinline__ void block_sum(double* v, double (*sPart)[NRED][NW], int& buf)
{
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum_f64(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sPart[buf][k][wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double s = sPart[buf][k][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += sPart[buf][k][w];
    v[k] = s;
  }
  buf ^= 1;
}

// The 27 sums of a linearisation (upper H + b).  A butterfly over all lanes costs ~30 instructions per value on the
// critical path; here every thread drops its 27 values into LDS ([value][thread], conflict-free), thread (value k,
// part p) of the first 216 adds 32 of them (reads rotated by the lane so a wave touches every bank pair at most
// twice), 3 DPP steps fold the 8 parts, and every thread reads the 27 totals back: ~130 instructions, two barriers.
// sT is private to this function (the alternating buffers of block_sum are not touched).
__device__ __forceinline__ void block_sum27(double* v, double (*sT)[PT], double* sTot)
{
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 27; ++k) sT[k][t] = v[k];
  __syncthreads();
  if (t < 27 * 8) {
    const int k = t >> 3, part = t & 7;
    const double* row = &sT[k][part * 32];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
      s0 += row[(j + t) & 31];
      s1 += row[(j + 1 + t) & 31];
    }
    double s = s0 + s1;
    s += dpp_f64<0xB1>(s);
    s += dpp_f64<0x4E>(s);
    s += dpp_f64<0x141>(s);
    if (part == 0) sTot[k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = sTot[k];
}

// 1 / sqrt(d): v_rsq_f64 + two Newton steps instead of an IEEE sqrt followed by an IEEE divide (~45 dependent
// instructions -> 9) on the serial pivot chain
__device__ __forceinline__ double rsqrt_nr(double d)
{
  double y = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = fma(-(d * y), y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  return y;
}

// (H + lambda I) x = b for the upper-triangular H[21] by Cholesky, fully unrolled; false when a pivot is not positive
// (LinearSolverDense: Eigen's LDLT reports failure the same way)
__device__ __forceinline__ bool solve6(const double* Hb, double lambda, double* x)
{
  double A[6][6];
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int c = a; c < 6; ++c) { A[c][a] = Hb[q]; ++q; }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) A[a][a] += lambda;
  bool ok = true;
  double rd[6];                                              // 1 / L[j][j]
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dj = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) dj -= A[j][k] * A[j][k];
    if (!(dj > 0.0) || !isfinite(dj)) { ok = false; dj = 1.0; }
    const double inv = rsqrt_nr(dj);
    rd[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double sum = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) sum -= A[i][k] * A[j][k];
      A[i][j] = sum * inv;
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double sum = Hb[21 + i];
#pragma unroll
    for (int k = 0; k < i; ++k) sum -= A[i][k] * y[k];
    y[i] = sum * rd[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sum = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) sum -= A[k][i] * x[k];
    x[i] = sum * rd[i];
  }
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
  }
  return ok;
}

// dv: one descriptor per problem of the batch (blockIdx.x = problem); ssx_pose_only_opt is a batch of one.  The descriptors
// and a problem's inputs live in a pinned host block the kernel reads DIRECTLY (each value once, at the top), its results go
// straight back into that block: one launch and one synchronisation per batch, no copy in either direction.
template <int EPT>
__global__ __launch_bounds__(PT) void k_pose_only(const PoDev* __restrict__ dv)
{
  const PoDev d = dv[blockIdx.x];
  __shared__ double sPart[2][NRED][NW];
  __shared__ double sT[27][PT];
  __shared__ double sTot[27];
  const int t = threadIdx.x;
  int buf = 0;
  // this thread's edges: i = t + k * PT
  double X[EPT][3], z[EPT][2], e[EPT][2];
  bool valid[EPT], level[EPT], outl[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = t + k * PT;
    valid[k] = i < d.M;
    level[k] = false; outl[k] = false;
    e[k][0] = 0.0; e[k][1] = 0.0;
    const int ii = valid[k] ? i : 0;
    X[k][0] = d.xyz[3 * ii]; X[k][1] = d.xyz[3 * ii + 1]; X[k][2] = d.xyz[3 * ii + 2];
    z[k][0] = d.uv[2 * ii]; z[k][1] = d.uv[2 * ii + 1];
  }
  double T[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) T[k] = d.pose[k];
  bool use_kernel = true;
  const ssx::Cam K = d.K;
  const double delta = d.huber_delta;

  // errors of the active (level 0) edges at pose T + their robust chi2, summed over the workgroup
  auto errors_and_chi2 = [&]() {
    double chi = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      double pc[3];
      ssx::se3_act(T, X[k], pc);
      const double hx = K.fx * pc[0] + K.cx * pc[2], hy = K.fy * pc[1] + K.cy * pc[2];
      const double e0 = z[k][0] - hx / pc[2], e1 = z[k][1] - hy / pc[2];
      const bool act = valid[k] && !level[k];
      e[k][0] = act ? e0 : e[k][0];
      e[k][1] = act ? e1 : e[k][1];
      const double c2 = e0 * e0 + e1 * e1;
      double r0 = c2, w = 1.0;
      if (use_kernel) ssx::huber(c2, delta, r0, w);
      chi += act ? r0 : 0.0;
    }
    block_sum<1>(&chi, sPart, buf);
    return chi;
  };

  int cnt_outliers = 0;
  for (int round = 0; round < d.rounds; ++round) {
    double na = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) na += (valid[k] && !level[k]) ? 1.0 : 0.0;
    block_sum<1>(&na, sPart, buf);
    const bool any_active = na > 0.0;                       // initializeOptimization(0): only level-0 edges are active
    double lambda = 0.0, ni = 2.0, current_chi = 0.0;
    for (int it = 0; it < d.iters && any_active; ++it) {
      // solve() starts with computeActiveErrors() + activeRobustChi2(); after an accepted trial (the only way to get
      // here with it > 0) the errors and the chi2 of that trial ARE those values -- no second pass over the edges
      if (it == 0) current_chi = errors_and_chi2();
      // ---- linearise: H (upper 21) and b (6) ----
      double Hb[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) Hb[k] = 0.0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        double pc[3];
        ssx::se3_act(T, X[k], pc);
        const double Xc = pc[0], Yc = pc[1], Zc = pc[2];
        const double Zinv = 1.0 / (Zc + 1e-18), Zinv2 = Zinv * Zinv;
        // EdgeProjectionPoseOnly::linearizeOplus, g2otypes.hpp:86-101
        const double J[12] = {-K.fx * Zinv, 0, K.fx * Xc * Zinv2, K.fx * Xc * Yc * Zinv2, -K.fx - K.fx * Xc * Xc * Zinv2, K.fx * Yc * Zinv,
                              0, -K.fy * Zinv, K.fy * Yc * Zinv2, K.fy + K.fy * Yc * Yc * Zinv2, -K.fy * Xc * Yc * Zinv2, -K.fy * Xc * Zinv};
        const double e0 = e[k][0], e1 = e[k][1];
        double r0, w = 1.0;
        if (use_kernel) ssx::huber(e0 * e0 + e1 * e1, delta, r0, w);
        w = (valid[k] && !level[k]) ? w : 0.0;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = a; c < 6; ++c) { Hb[q] += J[a] * w * J[c] + J[6 + a] * w * J[6 + c]; ++q; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) Hb[21 + a] -= w * (J[a] * e0 + J[6 + a] * e1);
      }
      block_sum27(Hb, sT, sTot);
      if (it == 0) {
        // computeLambdaInit: 1e-5 * max |diag(H)|; diagonal entries of the upper layout: 0,6,11,15,18,20
        const double m = fmax(fmax(fmax(fabs(Hb[0]), fabs(Hb[6])), fmax(fabs(Hb[11]), fabs(Hb[15]))), fmax(fabs(Hb[18]), fabs(Hb[20])));
        lambda = 1e-5 * m; ni = 2.0;
      }
      // ---- LM trials ----
      int qmax = 0;
      bool stop = false;
      while (true) {
        double Tbak[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) Tbak[k] = T[k];
        double x[6];
        const bool ok = solve6(Hb, lambda, x);
        double Tn[7];
        ssx::pose_oplus(T, x, Tn);
#pragma unroll
        for (int k = 0; k < 7; ++k) T[k] = Tn[k];
        double temp_chi = errors_and_chi2();
        if (!ok) temp_chi = 1.7976931348623157e308;
        double scale = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + Hb[21 + j]);
        scale += 1e-3;
        const double rho = (current_chi - temp_chi) / scale;
        bool lambda_bad = false;
        if (rho > 0 && isfinite(temp_chi)) {
          const double c = 2 * rho - 1;
          double alpha = 1. - c * c * c;
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2.0;
          current_chi = temp_chi;
        } else {
          lambda *= ni;
          ni *= 2.0;
#pragma unroll
          for (int k = 0; k < 7; ++k) T[k] = Tbak[k];          // pop(): vertices only, errors stay at the trial state
          if (!isfinite(lambda)) lambda_bad = true;
        }
        qmax += lambda_bad ? 0 : 1;
        // do { } while (rho < 0 && qmax < 10);  then Terminate if qmax == 10 || rho == 0 || lambda non-finite
        stop = (qmax == 10 || rho == 0 || lambda_bad);
        if (!(!lambda_bad && rho < 0 && qmax < 10)) break;
      }
      if (stop) break;                                         // Terminate: optimize() stops iterating
    }
    // frontend.cpp:243-268: recompute the error only for features flagged outlier, classify, set levels
    double co = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (outl[k]) {
        double pc[3];
        ssx::se3_act(T, X[k], pc);
        const double hx = K.fx * pc[0] + K.cx * pc[2], hy = K.fy * pc[1] + K.cy * pc[2];
        e[k][0] = z[k][0] - hx / pc[2]; e[k][1] = z[k][1] - hy / pc[2];
      }
      const double c2 = e[k][0] * e[k][0] + e[k][1] * e[k][1];
      const bool out = valid[k] && c2 > d.chi2_th;
      outl[k] = out; level[k] = out;
      co += out ? 1.0 : 0.0;
    }
    block_sum<1>(&co, sPart, buf);
    cnt_outliers = (int)co;
    if (round == d.rounds - 2) use_kernel = false;             // e->setRobustKernel(nullptr)
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k)
    if (valid[k]) d.outlier[t + k * PT] = outl[k] ? 1 : 0;
  if (t < 7) d.pose_out[t] = T[t];
  if (t == 0) *d.n_inliers = d.M - cnt_outliers;
}

}  // namespace

struct PoWorkspace { DevBuf arena; HostBuf stage; };

namespace {

// the generic kernel (M > 6 x 256 edges: never a front-end's frame) keeps its edges in device memory: one problem per call
ssx_status pose_only_generic(ssx_ctx* ctx, double* pose_io, const double* K4, int32_t M, const double* xyz, const double* uv, int32_t rounds,
                             int32_t iters, double chi2_th, double huber_delta, uint8_t* inlier_out, int32_t* n_inliers)
{
  DevBuf& arena = ctx->po_arena;
  HostBuf& stage = ctx->po_stage;
  Layout lay;
  const size_t o_xyz = lay.take(sizeof(double) * 3 * (size_t)M);
  const size_t o_uv = lay.take(sizeof(double) * 2 * (size_t)M);
  const size_t o_pose = lay.take(sizeof(double) * 8);
  const size_t in_bytes = lay.off;
  const size_t o_err = lay.take(sizeof(double) * 2 * (size_t)M);
  const size_t o_level = lay.take((size_t)M);
  // results, contiguous so that one copy brings them back: pose | inlier count | outlier flags
  const size_t o_res = lay.take(sizeof(double) * 8 + sizeof(int) * 2 + (size_t)M);
  const size_t o_pose_out = o_res, o_n = o_res + sizeof(double) * 8, o_out = o_n + sizeof(int) * 2;
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
  d.pose = (double*)(base + o_pose); d.pose_out = (double*)(base + o_pose_out); d.n_inliers = (int*)(base + o_n);
  SSX_PROF(ctx, KID_POSE_ONLY, hipLaunchKernelGGL(k_pose_only_generic, dim3(1), dim3(PT), 0, ctx->stream, d));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_res, base + o_res, sizeof(double) * 8 + sizeof(int) * 2 + (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(pose_io, hs + o_pose_out, sizeof(double) * 7);
  if (inlier_out) for (int i = 0; i < M; ++i) inlier_out[i] = !reinterpret_cast<uint8_t*>(hs + o_out)[i];
  if (n_inliers) *n_inliers = *reinterpret_cast<int*>(hs + o_n);
  return SSX_OK;
}

}  // namespace

// n problems in one call (one frame of each of n streams: FrontEnd::EstimateCurrentPose, frontend.cpp:184-300): one workgroup per
// problem, ONE launch per register class of the kernel.  Per problem the bits of ssx_pose_only_opt (which is a batch of one).
extern "C" ssx_status ssx_pose_only_opt_batch(ssx_ctx* ctx, int32_t n, const ssx_pose_only_job* jobs)
{
  if (!ctx || n < 0 || (n > 0 && !jobs)) return SSX_ERR_INVALID_ARG;
  for (int j = 0; j < n; ++j) {
    const ssx_pose_only_job& q = jobs[j];
    if (!q.pose_io || !q.K4 || q.M < 0 || (q.M && (!q.xyz || !q.uv)) || q.rounds < 0 || q.iters < 0) return SSX_ERR_INVALID_ARG;
  }
  if (n == 0) return SSX_OK;
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // classes: 0 = up to 2 edges per thread, 1 = up to 6 (the register-resident kernels), 2 = generic, 3 = empty
  std::vector<int> order[2];
  for (int j = 0; j < n; ++j) {
    const int M = jobs[j].M;
    if (M == 0) { if (jobs[j].n_inliers) *jobs[j].n_inliers = 0; continue; }
    if (M <= PT * 2) order[0].push_back(j); else if (M <= PT * 6) order[1].push_back(j);
  }
  const size_t nb = order[0].size() + order[1].size();
  if (nb) {
    // one pinned block: [descriptors | per problem: xyz, uv, pose in | pose out, inlier count, outlier flags]
    Layout lay;
    const size_t o_desc = lay.take(sizeof(PoDev) * nb);
    std::vector<size_t> o_in(nb), o_out(nb);
    size_t k = 0;
    for (int c = 0; c < 2; ++c)
      for (int j : order[c]) {
        const size_t M = (size_t)jobs[j].M;
        o_in[k] = lay.take(sizeof(double) * (5 * M + 8));
        o_out[k] = lay.take(sizeof(double) * 8 + sizeof(int) * 2 + M);
        ++k;
      }
    HostBuf& stage = ctx->po_stage;
    SSX_HIP_TRY(ctx, stage.reserve(lay.off, 2.0));
    char* hs = stage.as<char>();
    PoDev* dv = reinterpret_cast<PoDev*>(hs + o_desc);
    k = 0;
    for (int c = 0; c < 2; ++c)
      for (int j : order[c]) {
        const ssx_pose_only_job& q = jobs[j];
        const size_t M = (size_t)q.M;
        double* in = reinterpret_cast<double*>(hs + o_in[k]);
        memcpy(in, q.xyz, sizeof(double) * 3 * M);
        memcpy(in + 3 * M, q.uv, sizeof(double) * 2 * M);
        memcpy(in + 5 * M, q.pose_io, sizeof(double) * 7);
        PoDev& d = dv[k];
        d.M = q.M; d.rounds = q.rounds; d.iters = q.iters; d.chi2_th = q.chi2_th; d.huber_delta = q.huber_delta;
        d.K = ssx::Cam{q.K4[0], q.K4[1], q.K4[2], q.K4[3]};
        d.xyz = in; d.uv = in + 3 * M; d.pose = in + 5 * M;
        d.err = nullptr; d.level = nullptr;                          // (the register-resident kernels keep both in registers)
        d.pose_out = reinterpret_cast<double*>(hs + o_out[k]);
        d.n_inliers = reinterpret_cast<int*>(hs + o_out[k] + sizeof(double) * 8);
        d.outlier = reinterpret_cast<uint8_t*>(hs + o_out[k] + sizeof(double) * 8 + sizeof(int) * 2);
        ++k;
      }
    if (!order[0].empty())
      SSX_PROF(ctx, KID_POSE_ONLY, hipLaunchKernelGGL(k_pose_only<2>, dim3((unsigned)order[0].size()), dim3(PT), 0, ctx->stream, (const PoDev*)dv));
    if (!order[1].empty())
      SSX_PROF(ctx, KID_POSE_ONLY, hipLaunchKernelGGL(k_pose_only<6>, dim3((unsigned)order[1].size()), dim3(PT), 0, ctx->stream, (const PoDev*)(dv + order[0].size())));
    SSX_HIP_TRY(ctx, hipGetLastError());
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    k = 0;
    for (int c = 0; c < 2; ++c)
      for (int j : order[c]) {
        const ssx_pose_only_job& q = jobs[j];
        const char* o = hs + o_out[k];
        memcpy(q.pose_io, o, sizeof(double) * 7);
        const uint8_t* ol = reinterpret_cast<const uint8_t*>(o + sizeof(double) * 8 + sizeof(int) * 2);
        if (q.inlier_out) for (int i = 0; i < q.M; ++i) q.inlier_out[i] = !ol[i];
        if (q.n_inliers) *q.n_inliers = *reinterpret_cast<const int*>(o + sizeof(double) * 8);
        ++k;
      }
  }
  for (int j = 0; j < n; ++j) {
    const ssx_pose_only_job& q = jobs[j];
    if (q.M <= PT * 6) continue;
    const ssx_status st = pose_only_generic(ctx, q.pose_io, q.K4, q.M, q.xyz, q.uv, q.rounds, q.iters, q.chi2_th, q.huber_delta, q.inlier_out, q.n_inliers);
    if (st != SSX_OK) return st;
  }
  return SSX_OK;
}

extern "C" ssx_status ssx_pose_only_opt(ssx_ctx* ctx, double* pose_io, const double* K4, int32_t M, const double* xyz,
                                        const double* uv, int32_t rounds, int32_t iters, double chi2_th, double huber_delta,
                                        uint8_t* inlier_out, int32_t* n_inliers)
{
  if (!ctx || !pose_io || !K4 || M < 0 || (M && (!xyz || !uv)) || rounds < 0 || iters < 0) return SSX_ERR_INVALID_ARG;
  ssx_pose_only_job q;
  q.pose_io = pose_io; q.K4 = K4; q.M = M; q.xyz = xyz; q.uv = uv; q.rounds = rounds; q.iters = iters; q.chi2_th = chi2_th; q.huber_delta = huber_delta;
  q.inlier_out = inlier_out; q.n_inliers = n_inliers;
  return ssx_pose_only_opt_batch(ctx, 1, &q);
}
