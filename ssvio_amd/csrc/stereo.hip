// ssvio_amd/csrc/stereo.hip -- stereo association and triangulation on gfx950.
//
//   k_row_bucket     counting sort of the right keypoints by image row (LDS histogram + scan)
//   k_match          ROW-BAND Hamming matcher, one wave per left keypoint: only the right keypoints whose row lies
//                    in the band are visited; 256-bit XOR + 4 x popcount per candidate; wave-wide min-reduce on
//                    (distance << 16 | right index) = minimum distance, lowest index wins ties
//                    (the semantics of OpenCV BruteForce-Hamming match(), /root/reference/src/ssvio/loopclosing.cpp:24,108)
//   k_bf_match       the unrestricted brute-force matcher of loop closing (loopclosing.cpp:105-110)
//   k_triangulate    ssvio::triangulation (/root/reference/include/ssvio/algorithm.hpp:23-45): DLT rows of the two 3x4
//                    poses, 4x4 SVD by one-sided Jacobi in registers (f64), sigma3/sigma2 < 1e-2 and z > 0
//
// The reference has no row-band matcher (its stereo association is LK optical flow, frontend.cpp:374-384); the
// matching rule is the one defined with the oracle (oracle/src/stereo_oracle.cpp) and is bit-exact against it.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ctx.hpp"
#include "orb_ws.hpp"

namespace ssxorb {

namespace {

struct MatchDev {
  int pairs, out_cap, rows;
  const ssx_keypoint* kps;   // [2*pairs][out_cap]
  const uint8_t* desc;       // [2*pairs][out_cap][32]
  const int* n;              // [2*pairs]
  int* row_ptr;              // [pairs][rows + 2]
  int* sorted;               // [pairs][out_cap]
  int* match_idx;            // [pairs][out_cap]
  int* match_dist;           // [pairs][out_cap]
  double* xyz;               // [pairs][out_cap][3]
  uint8_t* ok;               // [pairs][out_cap]
  int* counts;               // [pairs][4]
  ssx_match_params mp;
  ssx_stereo_rig rig;
  double T_wc[7];
  int has_T;
  float scale[32];
};

constexpr int BUCKET_ROWS_MAX = 4096;

__global__ __launch_bounds__(1024) void k_row_bucket(MatchDev m)
{
  __shared__ int hist[BUCKET_ROWS_MAX + 1];
  __shared__ int wsum[17];
  const int pair = blockIdx.x, t = threadIdx.x;
  const int nR = min(m.n[2 * pair + 1], m.out_cap);
  const ssx_keypoint* kR = m.kps + (size_t)(2 * pair + 1) * m.out_cap;
  int* row_ptr = m.row_ptr + (size_t)pair * (m.rows + 2);
  int* sorted = m.sorted + (size_t)pair * m.out_cap;
  const int R = m.rows + 1;
  for (int i = t; i <= R; i += 1024) hist[i] = 0;
  __syncthreads();
  for (int j = t; j < nR; j += 1024) {
    int r = (int)kR[j].y;
    r = r < 0 ? 0 : (r >= R ? R - 1 : r);
    atomicAdd(&hist[r], 1);
  }
  __syncthreads();
  // exclusive scan of hist[0..R) by 1024 threads: contiguous chunks + wave scan
  const int chunk = (R + 1023) / 1024;
  const int lo = min(t * chunk, R), hi = min(lo + chunk, R);
  int mine = 0;
  for (int i = lo; i < hi; ++i) mine += hist[i];
  int inc = mine;
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (t == 0) { int run = 0; for (int w = 0; w < 16; ++w) { const int x = wsum[w]; wsum[w] = run; run += x; } wsum[16] = run; }
  __syncthreads();
  int run = wsum[wave] + inc - mine;
  for (int i = lo; i < hi; ++i) { const int c = hist[i]; row_ptr[i] = run; hist[i] = run; run += c; }
  if (t == 0) row_ptr[R] = wsum[16];
  __syncthreads();
  for (int j = t; j < nR; j += 1024) {
    int r = (int)kR[j].y;
    r = r < 0 ? 0 : (r >= R ? R - 1 : r);
    sorted[atomicAdd(&hist[r], 1)] = j;   // order inside a row is irrelevant: the reduction key carries the index
  }
}

__device__ __forceinline__ int hamming256(const uint8_t* a, const uint8_t* b)
{
  const unsigned long long* x = reinterpret_cast<const unsigned long long*>(a);
  const unsigned long long* y = reinterpret_cast<const unsigned long long*>(b);
  return __popcll(x[0] ^ y[0]) + __popcll(x[1] ^ y[1]) + __popcll(x[2] ^ y[2]) + __popcll(x[3] ^ y[3]);
}

__global__ __launch_bounds__(256) void k_match(MatchDev m)
{
  const int pair = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
  const int i = blockIdx.x * 4 + wave;
  const int nL = min(m.n[2 * pair], m.out_cap), nR = min(m.n[2 * pair + 1], m.out_cap);
  if (i >= nL) return;
  const ssx_keypoint* kL = m.kps + (size_t)(2 * pair) * m.out_cap;
  const ssx_keypoint* kR = m.kps + (size_t)(2 * pair + 1) * m.out_cap;
  const uint8_t* dL = m.desc + (size_t)(2 * pair) * m.out_cap * 32;
  const uint8_t* dR = m.desc + (size_t)(2 * pair + 1) * m.out_cap * 32;
  const int* row_ptr = m.row_ptr + (size_t)pair * (m.rows + 2);
  const int* sorted = m.sorted + (size_t)pair * m.out_cap;
  const ssx_keypoint a = kL[i];
  const int ol = a.octave < 0 ? 0 : (a.octave > 31 ? 31 : a.octave);
  const float band = m.mp.band_px * m.scale[ol];
  const int R = m.rows + 1;
  int r0 = (int)floorf(a.y - band), r1 = (int)floorf(a.y + band);
  r0 = r0 < 0 ? 0 : (r0 >= R ? R - 1 : r0);
  r1 = r1 < 0 ? 0 : (r1 >= R ? R - 1 : r1);
  const int c0 = (nR > 0) ? row_ptr[r0] : 0, c1 = (nR > 0) ? row_ptr[r1 + 1] : 0;
  // my descriptor in registers
  const unsigned long long* da = reinterpret_cast<const unsigned long long*>(dL + (size_t)i * 32);
  const unsigned long long a0 = da[0], a1 = da[1], a2 = da[2], a3 = da[3];
  unsigned best = (257u << 16) | 0xFFFFu;
  for (int c = c0 + lane; c < c1; c += 64) {
    const int j = sorted[c];
    const ssx_keypoint b = kR[j];
    const float dv = a.y - b.y;
    if (dv > band || -dv > band) continue;
    int doct = a.octave - b.octave;
    doct = doct < 0 ? -doct : doct;
    if (doct > m.mp.max_octave_diff) continue;
    const float disp = a.x - b.x;
    if (disp < m.mp.min_disp || disp > m.mp.max_disp) continue;
    const unsigned long long* db = reinterpret_cast<const unsigned long long*>(dR + (size_t)j * 32);
    const int d = __popcll(a0 ^ db[0]) + __popcll(a1 ^ db[1]) + __popcll(a2 ^ db[2]) + __popcll(a3 ^ db[3]);
    const unsigned key = ((unsigned)d << 16) | (unsigned)j;
    best = min(best, key);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
  if (lane == 0) {
    const int d = (int)(best >> 16), j = (int)(best & 0xFFFFu);
    m.match_dist[(size_t)pair * m.out_cap + i] = d;
    m.match_idx[(size_t)pair * m.out_cap + i] = (d <= 256 && d <= m.mp.max_dist) ? j : -1;
  }
}

// generic brute force: query block x all train descriptors (host-array entry point ssx_bf_match)
__global__ __launch_bounds__(256) void k_bf_match(const uint8_t* dq, int nq, const uint8_t* dt, int nt, int* idx, int* dist)
{
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
  const int i = blockIdx.x * 4 + wave;
  if (i >= nq) return;
  unsigned best = (257u << 16) | 0xFFFFu;
  for (int j = lane; j < nt; j += 64) {
    const int d = hamming256(dq + (size_t)i * 32, dt + (size_t)j * 32);
    best = min(best, ((unsigned)d << 16) | (unsigned)j);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
  if (lane == 0) {
    const int d = (int)(best >> 16);
    dist[i] = d;
    idx[i] = d <= 256 ? (int)(best & 0xFFFFu) : -1;
  }
}

// ---- 4x4 SVD by one-sided (Hestenes) Jacobi, everything in registers (loops fully unrolled: static indices) ----
// frontend.cpp:458-465 / algorithm.hpp:23-45 take the right singular vector of the smallest singular value from
// Eigen::JacobiSVD; any convergent Jacobi ordering gives it to ~1e-15 relative.  This one is laid out for the
// latency of ONE thread (a stereo frame has ~1000 matches: the kernel is a few hundred waves, all latency):
//   * round-robin ordering {(0,1),(2,3)}, {(0,2),(1,3)}, {(0,3),(1,2)}: the two rotations of a round touch disjoint
//     columns, so their dependent sqrt/div chains interleave;
//   * t = 2g / (d + sign(d) * hypot(d, 2g)), c = rsqrt(1 + t^2): three slow fp64 operations per rotation instead of
//     the textbook five (zeta, sqrt, 1/x, sqrt, 1/x);
//   * explicit fma (the file is compiled with -ffp-contract=off for the float code that must match the CPU bit
//     for bit; this routine is compared at 1e-9 relative, see tests);
//   * a sweep whose largest |g| / sqrt(a b) was below 1e-8 ends the iteration: Jacobi converges quadratically, so
//     the off-diagonal mass after it is at rounding level and the usual extra "nothing rotated" sweep is skipped.
__device__ __forceinline__ void jacobi_moments(const double* U, int p, int q, double& a, double& b, double& g)
{
  a = U[p] * U[p]; b = U[q] * U[q]; g = U[p] * U[q];
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    a = fma(U[r * 4 + p], U[r * 4 + p], a);
    b = fma(U[r * 4 + q], U[r * 4 + q], b);
    g = fma(U[r * 4 + p], U[r * 4 + q], g);
  }
}

__device__ __forceinline__ void jacobi_rotation(double a, double b, double g, double& c, double& s, bool& big)
{
  const double g2 = g * g, ab = a * b;
  const bool on = g2 > 1e-30 * ab;            // |g| > 1e-15 sqrt(a b)
  big = big || (g2 > 1e-16 * ab);
  const double d = b - a, tg = g + g;
  const double h = sqrt(fma(d, d, tg * tg));
  const double t = tg / (d + copysign(h, d));
  const double cc = rsqrt(fma(t, t, 1.0));
  c = on ? cc : 1.0;
  s = on ? cc * t : 0.0;
}

__device__ __forceinline__ void jacobi_apply(double* U, double* V, int p, int q, double c, double s)
{
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double up = U[r * 4 + p], uq = U[r * 4 + q];
    U[r * 4 + p] = fma(c, up, -(s * uq));
    U[r * 4 + q] = fma(s, up, c * uq);
    const double vp = V[r * 4 + p], vq = V[r * 4 + q];
    V[r * 4 + p] = fma(c, vp, -(s * vq));
    V[r * 4 + q] = fma(s, vp, c * vq);
  }
}

__device__ __forceinline__ void jacobi_round(double* U, double* V, int p0, int q0, int p1, int q1, bool& big)
{
  double a0, b0, g0, a1, b1, g1, c0, s0, c1, s1;
  jacobi_moments(U, p0, q0, a0, b0, g0);
  jacobi_moments(U, p1, q1, a1, b1, g1);
  jacobi_rotation(a0, b0, g0, c0, s0, big);
  jacobi_rotation(a1, b1, g1, c1, s1, big);
  jacobi_apply(U, V, p0, q0, c0, s0);
  jacobi_apply(U, V, p1, q1, c1, s1);
}

__device__ __forceinline__ void triangulate_one(double uL, double vL, double uR, double vR, const ssx_stereo_rig& rig,
                                                const double* T_wc, double* xyz, uint8_t* ok)
{
  const double x1 = (uL - rig.cx) / rig.fx * 1.0, y1 = (vL - rig.cy) / rig.fy * 1.0;
  const double x2 = (uR - rig.cx) / rig.fx * 1.0, y2 = (vR - rig.cy) / rig.fy * 1.0;
  const double tx = -rig.baseline;
  double U[16] = {-1, 0, x1, 0, 0, -1, y1, 0, -1, 0, x2, x2 * 0.0 - tx, 0, -1, y2, y2 * 0.0 - 0.0};
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool big = false;
    jacobi_round(U, V, 0, 1, 2, 3, big);
    jacobi_round(U, V, 0, 2, 1, 3, big);
    jacobi_round(U, V, 0, 3, 1, 2, big);
    if (!big) break;
  }
  double n[4];   // SQUARED singular values
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double s = U[c] * U[c];
#pragma unroll
    for (int r = 1; r < 4; ++r) s = fma(U[r * 4 + c], U[r * 4 + c], s);
    n[c] = s;
  }
  // smallest and second smallest singular value, ties resolved like the oracle's stable descending sort:
  // ord = [0,1,2,3]; swap when n[ord[j]] > n[ord[i]] (j > i)
  int ord[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) {
      const bool sw = n[ord[j]] > n[ord[i]];
      const int a = ord[i], b = ord[j];
      ord[i] = sw ? b : a; ord[j] = sw ? a : b;
    }
  const int cmin = ord[3], c2 = ord[2];
  double vx = 0, vy = 0, vz = 0, vw = 0, smin = 0, s2 = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c == cmin) { vx = V[0 * 4 + c]; vy = V[1 * 4 + c]; vz = V[2 * 4 + c]; vw = V[3 * 4 + c]; smin = n[c]; }
    if (c == c2) s2 = n[c];
  }
  const double iw = 1.0 / vw;
  double p[3] = {vx * iw, vy * iw, vz * iw};
  // sigma3 / sigma2 < 1e-2 on the squared values (algorithm.hpp:38).  A pair without positive disparity is a point at
  // or behind infinity: with uL == uR the homogeneous w is rounding noise and the sign of z with it, so the z > 0 test
  // of the callers (frontend.cpp:466,528) is taken on the disparity there -- deterministic, and the point is zeroed.
  const bool positive_disparity = uL - uR > 0.0;
  *ok = ((smin < 1e-4 * s2) && (p[2] > 0) && positive_disparity) ? 1 : 0;
  if (!positive_disparity) { p[0] = 0.0; p[1] = 0.0; p[2] = 0.0; }
  if (T_wc) {
    const double qx = T_wc[0], qy = T_wc[1], qz = T_wc[2], qw = T_wc[3];
    double ux = qy * p[2] - qz * p[1], uy = qz * p[0] - qx * p[2], uz = qx * p[1] - qy * p[0];
    ux += ux; uy += uy; uz += uz;
    const double rx = p[0] + qw * ux + (qy * uz - qz * uy);
    const double ry = p[1] + qw * uy + (qz * ux - qx * uz);
    const double rz = p[2] + qw * uz + (qx * uy - qy * ux);
    p[0] = rx + T_wc[4]; p[1] = ry + T_wc[5]; p[2] = rz + T_wc[6];
  }
  xyz[0] = p[0]; xyz[1] = p[1]; xyz[2] = p[2];
}

__global__ __launch_bounds__(256) void k_triangulate_matches(MatchDev m)
{
  const int pair = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nL = min(m.n[2 * pair], m.out_cap);
  if (i >= nL) return;
  const int j = m.match_idx[(size_t)pair * m.out_cap + i];
  double* xyz = m.xyz + ((size_t)pair * m.out_cap + i) * 3;
  uint8_t* ok = m.ok + (size_t)pair * m.out_cap + i;
  if (j < 0) { xyz[0] = 0; xyz[1] = 0; xyz[2] = 0; *ok = 0; return; }
  const ssx_keypoint a = m.kps[(size_t)(2 * pair) * m.out_cap + i];
  const ssx_keypoint b = m.kps[(size_t)(2 * pair + 1) * m.out_cap + j];
  // cv::Point2f widened to double (frontend.cpp:458-465)
  triangulate_one((double)a.x, (double)a.y, (double)b.x, (double)b.y, m.rig, m.has_T ? m.T_wc : nullptr, xyz, ok);
}

__global__ __launch_bounds__(256) void k_triangulate_uv(int n, const double* uvL, const double* uvR, ssx_stereo_rig rig,
                                                        MatchDev m, double* xyz, uint8_t* ok)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  triangulate_one(uvL[2 * i], uvL[2 * i + 1], uvR[2 * i], uvR[2 * i + 1], rig, m.has_T ? m.T_wc : nullptr, xyz + 3 * (size_t)i,
                  ok + i);
}

// n_jobs triangulation calls in one launch (ssx_triangulate_batch): blockIdx.y = job, its points at [p0, p0 + n) of the call's arrays;
// the job table, the inputs and the outputs live in one pinned host block the kernel reads and writes directly
struct TriJob { int n, p0, has_T, pad; ssx_stereo_rig rig; double T_wc[7]; };
__global__ __launch_bounds__(256) void k_triangulate_uv_b(const TriJob* __restrict__ jobs, const double* __restrict__ uvL, const double* __restrict__ uvR,
                                                          double* __restrict__ xyz, uint8_t* __restrict__ ok)
{
  const TriJob jb = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= jb.n) return;
  const size_t g = (size_t)jb.p0 + i;
  triangulate_one(uvL[2 * g], uvL[2 * g + 1], uvR[2 * g], uvR[2 * g + 1], jb.rig, jb.has_T ? jb.T_wc : nullptr, xyz + 3 * g, ok + g);
}

__global__ __launch_bounds__(256) void k_pair_counts(MatchDev m)
{
  __shared__ int s[2];
  const int pair = blockIdx.x, t = threadIdx.x;
  const int nL = min(m.n[2 * pair], m.out_cap);
  if (t < 2) s[t] = 0;
  __syncthreads();
  int nm = 0, nt = 0;
  for (int i = t; i < nL; i += 256) {
    nm += m.match_idx[(size_t)pair * m.out_cap + i] >= 0;
    nt += m.ok[(size_t)pair * m.out_cap + i] != 0;
  }
  atomicAdd(&s[0], nm);
  atomicAdd(&s[1], nt);
  __syncthreads();
  if (t == 0) {
    int* c = m.counts + 4 * (size_t)pair;
    c[0] = nL; c[1] = min(m.n[2 * pair + 1], m.out_cap); c[2] = s[0]; c[3] = s[1];
  }
}

// Single-frame path: the counts of k_pair_counts (workgroup 0) and, beside it, every result array of pair 0 written
// straight into pinned host memory -- one launch instead of k_pair_counts + nine device-to-host copies (5 us against 73).
struct FrameSlots { size_t cnt, kps, desc, idx, dist, xyz, ok; };   // byte offsets in the pinned block, capacity based

__device__ __forceinline__ void copy_words(uint32_t* dst, const uint32_t* src, int words, int tid, int nthr)
{
  for (int i = tid; i < words; i += nthr) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_pack_frame(MatchDev m, const int* status, char* host, FrameSlots o)
{
  const int t = threadIdx.x;
  const int nL = min(m.n[0], m.out_cap), nR = min(m.n[1], m.out_cap);
  if (blockIdx.x == 0) {
    __shared__ int s[2];
    if (t < 2) s[t] = 0;
    __syncthreads();
    int nm = 0, nt = 0;
    for (int i = t; i < nL; i += 256) { nm += m.match_idx[i] >= 0; nt += m.ok[i] != 0; }
    atomicAdd(&s[0], nm);
    atomicAdd(&s[1], nt);
    __syncthreads();
    if (t == 0) {
      int* c = m.counts;
      c[0] = nL; c[1] = nR; c[2] = s[0]; c[3] = s[1];
      int* h = reinterpret_cast<int*>(host + o.cnt);
      h[0] = nL; h[1] = nR; h[2] = s[0]; h[3] = s[1]; h[4] = m.n[0]; h[5] = m.n[1]; h[6] = status[0]; h[7] = status[1];
    }
    return;
  }
  const int tid = (blockIdx.x - 1) * 256 + t, nthr = (gridDim.x - 1) * 256;
  const size_t cap = m.out_cap;
  const uint32_t* kps = reinterpret_cast<const uint32_t*>(m.kps);
  const uint32_t* desc = reinterpret_cast<const uint32_t*>(m.desc);
  copy_words(reinterpret_cast<uint32_t*>(host + o.kps), kps, nL * 7, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.kps) + cap * 7, kps + cap * 7, nR * 7, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.desc), desc, nL * 8, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.desc) + cap * 8, desc + cap * 8, nR * 8, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.idx), reinterpret_cast<const uint32_t*>(m.match_idx), nL, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.dist), reinterpret_cast<const uint32_t*>(m.match_dist), nL, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.xyz), reinterpret_cast<const uint32_t*>(m.xyz), nL * 6, tid, nthr);
  copy_words(reinterpret_cast<uint32_t*>(host + o.ok), reinterpret_cast<const uint32_t*>(m.ok), (nL + 3) / 4, tid, nthr);
}
static_assert(sizeof(ssx_keypoint) == 28, "k_pack_frame copies keypoints as seven words");

void fill_scale(MatchDev& m)
{
  m.scale[0] = 1.0f;
  for (int i = 1; i < 32; ++i) m.scale[i] = m.scale[i - 1] * m.mp.scale_factor;   // mvScaleFactor
}

// allocate the stereo result buffers for `pairs` pairs and build the device view
ssx_status make_match_dev(ssx_ctx* ctx, int pairs, const ssx_match_params& mp, const ssx_stereo_rig& rig, const double* T_wc,
                          MatchDev& m)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  if (d.lvl_rows[0] + 2 > BUCKET_ROWS_MAX) { ctx->set_error("ssx_stereo: image too tall for the row buckets"); return SSX_ERR_UNSUPPORTED; }
  if (d.out_cap > 65535) { ctx->set_error("ssx_stereo: more than 65535 keypoints per image"); return SSX_ERR_UNSUPPORTED; }
  Layout lay;
  const size_t o_rp = lay.take(sizeof(int) * (size_t)pairs * (d.lvl_rows[0] + 2));
  const size_t o_sorted = lay.take(sizeof(int) * (size_t)pairs * d.out_cap);
  const size_t o_idx = lay.take(sizeof(int) * (size_t)pairs * d.out_cap);
  const size_t o_dist = lay.take(sizeof(int) * (size_t)pairs * d.out_cap);
  const size_t o_xyz = lay.take(sizeof(double) * 3 * (size_t)pairs * d.out_cap);
  const size_t o_ok = lay.take((size_t)pairs * d.out_cap);
  const size_t o_cnt = lay.take(sizeof(int) * 4 * (size_t)pairs);
  SSX_HIP_TRY(ctx, ws->stereo.reserve(lay.off));
  char* base = ws->stereo.as<char>();
  m.pairs = pairs; m.out_cap = d.out_cap; m.rows = d.lvl_rows[0];
  m.kps = reinterpret_cast<const ssx_keypoint*>(d.out_kps); m.desc = d.out_desc; m.n = d.out_n;
  m.row_ptr = (int*)(base + o_rp); m.sorted = (int*)(base + o_sorted);
  m.match_idx = (int*)(base + o_idx); m.match_dist = (int*)(base + o_dist);
  m.xyz = (double*)(base + o_xyz); m.ok = (uint8_t*)(base + o_ok); m.counts = (int*)(base + o_cnt);
  m.mp = mp; m.rig = rig;
  m.has_T = T_wc ? 1 : 0;
  for (int i = 0; i < 7; ++i) m.T_wc[i] = T_wc ? T_wc[i] : (i == 3 ? 1.0 : 0.0);
  fill_scale(m);
  ws->match_idx = m.match_idx; ws->match_dist = m.match_dist; ws->xyz = m.xyz; ws->tri_ok = m.ok; ws->pair_counts = m.counts;
  return SSX_OK;
}

ssx_status launch_stereo(ssx_ctx* ctx, const MatchDev& m, char* frame_host = nullptr, const FrameSlots* slots = nullptr)
{
  hipStream_t s = ctx->stream;
  if (frame_host) {
    SSX_PROF(ctx, KID_ST_BUCKET, hipLaunchKernelGGL(k_row_bucket, dim3(m.pairs), dim3(1024), 0, s, m));
    SSX_PROF(ctx, KID_ST_MATCH, hipLaunchKernelGGL(k_match, dim3((m.out_cap + 3) / 4, m.pairs), dim3(256), 0, s, m));
    SSX_PROF(ctx, KID_ST_TRIANGULATE, hipLaunchKernelGGL(k_triangulate_matches, dim3((m.out_cap + 255) / 256, m.pairs), dim3(256), 0, s, m));
    SSX_PROF(ctx, KID_ST_MISC, hipLaunchKernelGGL(k_pack_frame, dim3(1 + 128), dim3(256), 0, s, m, get_ws(ctx)->dev.status, frame_host, *slots));
    SSX_HIP_TRY(ctx, hipGetLastError());
    return SSX_OK;
  }
  SSX_PROF(ctx, KID_ST_BUCKET, hipLaunchKernelGGL(k_row_bucket, dim3(m.pairs), dim3(1024), 0, s, m));
  SSX_PROF(ctx, KID_ST_MATCH, hipLaunchKernelGGL(k_match, dim3((m.out_cap + 3) / 4, m.pairs), dim3(256), 0, s, m));
  SSX_PROF(ctx, KID_ST_TRIANGULATE, hipLaunchKernelGGL(k_triangulate_matches, dim3((m.out_cap + 255) / 256, m.pairs), dim3(256), 0, s, m));
  SSX_PROF(ctx, KID_ST_MISC, hipLaunchKernelGGL(k_pair_counts, dim3(m.pairs), dim3(256), 0, s, m));
  SSX_HIP_TRY(ctx, hipGetLastError());
  return SSX_OK;
}

ssx_status fetch_pair(ssx_ctx* ctx, int pair, ssx_stereo_frame_out* out)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  int c[4];
  SSX_HIP_TRY(ctx, hipMemcpyAsync(c, ws->pair_counts + 4 * (size_t)pair, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  int32_t nL = 0, nR = 0;
  ssx_status st = fetch_image(ctx, 2 * pair, out->cap, out->kpsL, out->descL, &nL);
  if (st != SSX_OK) return st;
  st = fetch_image(ctx, 2 * pair + 1, out->cap, out->kpsR, out->descR, &nR);
  if (st != SSX_OK) return st;
  out->nL = nL; out->nR = nR; out->n_matched = c[2]; out->n_triangulated = c[3];
  if (nL > 0) {
    if (out->match_idx) SSX_HIP_TRY(ctx, hipMemcpyAsync(out->match_idx, ws->match_idx + (size_t)pair * d.out_cap, sizeof(int) * nL, hipMemcpyDeviceToHost, ctx->stream));
    if (out->match_dist) SSX_HIP_TRY(ctx, hipMemcpyAsync(out->match_dist, ws->match_dist + (size_t)pair * d.out_cap, sizeof(int) * nL, hipMemcpyDeviceToHost, ctx->stream));
    if (out->xyz) SSX_HIP_TRY(ctx, hipMemcpyAsync(out->xyz, ws->xyz + (size_t)pair * d.out_cap * 3, sizeof(double) * 3 * nL, hipMemcpyDeviceToHost, ctx->stream));
    if (out->ok) SSX_HIP_TRY(ctx, hipMemcpyAsync(out->ok, ws->tri_ok + (size_t)pair * d.out_cap, nL, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SSX_OK;
}

// One stereo pair's results with ONE stream synchronisation: k_pack_frame has written them into pinned memory; wait, check,
// and hand them to the caller's (pageable) arrays.  (fetch_pair takes five synchronisations and nine copies.)
FrameSlots frame_slots(size_t cap, size_t* total)
{
  Layout lay;
  FrameSlots o;
  o.cnt = lay.take(sizeof(int) * 8);
  o.kps = lay.take(sizeof(ssx_keypoint) * 2 * cap); o.desc = lay.take((size_t)64 * cap);
  o.idx = lay.take(sizeof(int) * cap); o.dist = lay.take(sizeof(int) * cap); o.xyz = lay.take(sizeof(double) * 3 * cap); o.ok = lay.take(cap + 4);
  *total = lay.off;
  return o;
}

ssx_status fetch_frame_fast(ssx_ctx* ctx, const FrameSlots& o, ssx_stereo_frame_out* out)
{
  OrbWorkspace* ws = get_ws(ctx);
  const size_t cap = ws->dev.out_cap;
  const char* hf = ws->fetch.as<char>();
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const int* c = reinterpret_cast<const int*>(hf + o.cnt);
  const int nL = c[4], nR = c[5];
  if (c[6] != 0 || c[7] != 0) {
    ctx->set_error("ssx_orb: internal capacity exceeded (status bits %d / %d: 1=candidates 2=octree nodes 4=outputs)", c[6], c[7]);
    return SSX_ERR_CAPACITY;
  }
  out->nL = nL; out->nR = nR; out->n_matched = c[2]; out->n_triangulated = c[3];
  if (nL > out->cap || nR > out->cap) { ctx->set_error("ssx_orb: %d / %d keypoints but capacity %d", nL, nR, out->cap); return SSX_ERR_CAPACITY; }
  if (out->kpsL) memcpy(out->kpsL, hf + o.kps, sizeof(ssx_keypoint) * nL);
  if (out->kpsR) memcpy(out->kpsR, hf + o.kps + sizeof(ssx_keypoint) * cap, sizeof(ssx_keypoint) * nR);
  if (out->descL) memcpy(out->descL, hf + o.desc, (size_t)32 * nL);
  if (out->descR) memcpy(out->descR, hf + o.desc + (size_t)32 * cap, (size_t)32 * nR);
  if (out->match_idx) memcpy(out->match_idx, hf + o.idx, sizeof(int) * nL);
  if (out->match_dist) memcpy(out->match_dist, hf + o.dist, sizeof(int) * nL);
  if (out->xyz) memcpy(out->xyz, hf + o.xyz, sizeof(double) * 3 * nL);
  if (out->ok) memcpy(out->ok, hf + o.ok, nL);
  return SSX_OK;
}

}  // namespace
}  // namespace ssxorb

using namespace ssxorb;

extern "C" {

void ssx_match_default_params(ssx_match_params* p)
{
  if (!p) return;
  p->band_px = 2.0f; p->min_disp = 0.0f; p->max_disp = 120.0f; p->max_dist = 80; p->max_octave_diff = 1; p->scale_factor = 1.2f;
}

ssx_status ssx_stereo_match(ssx_ctx* ctx, const ssx_keypoint* kL, const uint8_t* dL, int32_t nL, const ssx_keypoint* kR,
                            const uint8_t* dR, int32_t nR, const ssx_match_params* prm, int32_t* match_idx, int32_t* dist)
{
  if (!ctx || !prm || nL < 0 || nR < 0 || (nL && (!kL || !dL || !match_idx || !dist)) || (nR && (!kR || !dR)))
    return SSX_ERR_INVALID_ARG;
  if (nL == 0) return SSX_OK;
  if (nL > 65535 || nR > 65535) { ctx->set_error("ssx_stereo_match: more than 65535 keypoints"); return SSX_ERR_UNSUPPORTED; }
  OrbWorkspace* ws = get_ws(ctx);
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int cap = std::max(nL, std::max(nR, 1));
  // rows covered by the buckets: max right y + 2
  float ymax = 0.f;
  for (int j = 0; j < nR; ++j) ymax = std::max(ymax, kR[j].y);
  for (int i = 0; i < nL; ++i) ymax = std::max(ymax, kL[i].y);
  const int rows = std::min((int)ymax + 2, BUCKET_ROWS_MAX - 2);
  Layout lay;
  const size_t o_k = lay.take(sizeof(ssx_keypoint) * 2 * (size_t)cap);
  const size_t o_d = lay.take((size_t)64 * cap);
  const size_t o_n = lay.take(sizeof(int) * 2);
  const size_t in_bytes = lay.off;
  const size_t o_rp = lay.take(sizeof(int) * (rows + 2));
  const size_t o_sorted = lay.take(sizeof(int) * (size_t)cap);
  const size_t o_idx = lay.take(sizeof(int) * (size_t)cap);
  const size_t o_dist = lay.take(sizeof(int) * (size_t)cap);
  SSX_HIP_TRY(ctx, ws->input.reserve(lay.off));
  SSX_HIP_TRY(ctx, ws->stage.reserve(lay.off));
  char* hs = ws->stage.as<char>();
  memcpy(hs + o_k, kL, sizeof(ssx_keypoint) * nL);
  if (nR) memcpy(hs + o_k + sizeof(ssx_keypoint) * cap, kR, sizeof(ssx_keypoint) * nR);
  memcpy(hs + o_d, dL, (size_t)32 * nL);
  if (nR) memcpy(hs + o_d + (size_t)32 * cap, dR, (size_t)32 * nR);
  int nn[2] = {nL, nR};
  memcpy(hs + o_n, nn, sizeof(nn));
  char* base = ws->input.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  MatchDev m{};
  m.pairs = 1; m.out_cap = cap; m.rows = rows;
  m.kps = (const ssx_keypoint*)(base + o_k); m.desc = (const uint8_t*)(base + o_d); m.n = (const int*)(base + o_n);
  m.row_ptr = (int*)(base + o_rp); m.sorted = (int*)(base + o_sorted);
  m.match_idx = (int*)(base + o_idx); m.match_dist = (int*)(base + o_dist);
  m.mp = *prm;
  fill_scale(m);
  SSX_PROF(ctx, KID_ST_BUCKET, hipLaunchKernelGGL(k_row_bucket, dim3(1), dim3(1024), 0, ctx->stream, m));
  SSX_PROF(ctx, KID_ST_MATCH, hipLaunchKernelGGL(k_match, dim3((cap + 3) / 4, 1), dim3(256), 0, ctx->stream, m));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(match_idx, m.match_idx, sizeof(int) * nL, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(dist, m.match_dist, sizeof(int) * nL, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

ssx_status ssx_bf_match(ssx_ctx* ctx, const uint8_t* dq, int32_t nq, const uint8_t* dt, int32_t nt, int32_t* idx,
                        int32_t* dist)
{
  if (!ctx || nq < 0 || nt < 0 || (nq && (!dq || !idx || !dist)) || (nt && !dt)) return SSX_ERR_INVALID_ARG;
  if (nq == 0) return SSX_OK;
  if (nt > 65535) { ctx->set_error("ssx_bf_match: more than 65535 train descriptors"); return SSX_ERR_UNSUPPORTED; }
  OrbWorkspace* ws = get_ws(ctx);
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  Layout lay;
  const size_t o_q = lay.take((size_t)32 * nq);
  const size_t o_t = lay.take((size_t)32 * std::max(nt, 1));
  const size_t in_bytes = lay.off;
  const size_t o_i = lay.take(sizeof(int) * nq);
  const size_t o_d = lay.take(sizeof(int) * nq);
  SSX_HIP_TRY(ctx, ws->input.reserve(lay.off));
  SSX_HIP_TRY(ctx, ws->stage.reserve(lay.off));
  char* hs = ws->stage.as<char>();
  memcpy(hs + o_q, dq, (size_t)32 * nq);
  if (nt) memcpy(hs + o_t, dt, (size_t)32 * nt);
  char* base = ws->input.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  SSX_PROF(ctx, KID_ST_MISC, hipLaunchKernelGGL(k_bf_match, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, (const uint8_t*)(base + o_q), nq,
                     (const uint8_t*)(base + o_t), nt, (int*)(base + o_i), (int*)(base + o_d)));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(idx, base + o_i, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(dist, base + o_d, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

ssx_status ssx_triangulate(ssx_ctx* ctx, int32_t n, const double* uvL, const double* uvR, const ssx_stereo_rig* rig,
                           const double* T_wc, double* xyz_out, uint8_t* ok_out)
{
  if (!ctx || !rig || n < 0 || (n && (!uvL || !uvR || !xyz_out || !ok_out))) return SSX_ERR_INVALID_ARG;
  if (n == 0) return SSX_OK;
  OrbWorkspace* ws = get_ws(ctx);
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  Layout lay;
  const size_t o_l = lay.take(sizeof(double) * 2 * (size_t)n);
  const size_t o_r = lay.take(sizeof(double) * 2 * (size_t)n);
  const size_t in_bytes = lay.off;
  const size_t o_x = lay.take(sizeof(double) * 3 * (size_t)n);
  const size_t o_k = lay.take((size_t)n);
  SSX_HIP_TRY(ctx, ws->input.reserve(lay.off));
  SSX_HIP_TRY(ctx, ws->stage.reserve(lay.off));
  char* hs = ws->stage.as<char>();
  memcpy(hs + o_l, uvL, sizeof(double) * 2 * n);
  memcpy(hs + o_r, uvR, sizeof(double) * 2 * n);
  char* base = ws->input.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  MatchDev m{};
  m.has_T = T_wc ? 1 : 0;
  for (int i = 0; i < 7; ++i) m.T_wc[i] = T_wc ? T_wc[i] : (i == 3 ? 1.0 : 0.0);
  SSX_PROF(ctx, KID_ST_MISC, hipLaunchKernelGGL(k_triangulate_uv, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (const double*)(base + o_l),
                     (const double*)(base + o_r), *rig, m, (double*)(base + o_x), (uint8_t*)(base + o_k)));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(xyz_out, base + o_x, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(ok_out, base + o_k, n, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

// n calls of ssx_triangulate in one launch and one synchronisation (one keyframe of each of n streams: FrontEnd::TriangulateNewPoints /
// BuidInitMap, frontend.cpp:448-544).  Per job the bits of ssx_triangulate.
ssx_status ssx_triangulate_batch(ssx_ctx* ctx, int32_t n_jobs, const ssx_triangulate_job* jobs)
{
  if (!ctx || n_jobs < 0 || (n_jobs > 0 && !jobs)) return SSX_ERR_INVALID_ARG;
  size_t total = 0;
  int max_n = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const ssx_triangulate_job& q = jobs[j];
    if (!q.rig || q.n < 0 || (q.n && (!q.uvL || !q.uvR || !q.xyz_out || !q.ok_out))) return SSX_ERR_INVALID_ARG;
    total += (size_t)q.n; max_n = std::max(max_n, q.n);
  }
  if (total == 0) return SSX_OK;
  OrbWorkspace* ws = get_ws(ctx);
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  Layout lay;
  const size_t o_tab = lay.take(sizeof(TriJob) * (size_t)n_jobs);
  const size_t o_l = lay.take(sizeof(double) * 2 * total), o_r = lay.take(sizeof(double) * 2 * total);
  const size_t o_x = lay.take(sizeof(double) * 3 * total), o_k = lay.take(total);
  SSX_HIP_TRY(ctx, ws->fetch.reserve(lay.off, 1.5));
  char* hs = ws->fetch.as<char>();
  TriJob* tab = reinterpret_cast<TriJob*>(hs + o_tab);
  size_t p0 = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const ssx_triangulate_job& q = jobs[j];
    TriJob& t = tab[j];
    t.n = q.n; t.p0 = (int)p0; t.has_T = q.T_wc ? 1 : 0; t.pad = 0; t.rig = *q.rig;
    for (int i = 0; i < 7; ++i) t.T_wc[i] = q.T_wc ? q.T_wc[i] : (i == 3 ? 1.0 : 0.0);
    if (q.n) { memcpy(hs + o_l + sizeof(double) * 2 * p0, q.uvL, sizeof(double) * 2 * (size_t)q.n); memcpy(hs + o_r + sizeof(double) * 2 * p0, q.uvR, sizeof(double) * 2 * (size_t)q.n); }
    p0 += (size_t)q.n;
  }
  SSX_PROF(ctx, KID_ST_MISC, hipLaunchKernelGGL(k_triangulate_uv_b, dim3((max_n + 255) / 256, n_jobs), dim3(256), 0, ctx->stream, (const TriJob*)tab,
                                                (const double*)(hs + o_l), (const double*)(hs + o_r), (double*)(hs + o_x), (uint8_t*)(hs + o_k)));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  p0 = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const ssx_triangulate_job& q = jobs[j];
    if (q.n) { memcpy(q.xyz_out, hs + o_x + sizeof(double) * 3 * p0, sizeof(double) * 3 * (size_t)q.n); memcpy(q.ok_out, hs + o_k + p0, (size_t)q.n); }
    p0 += (size_t)q.n;
  }
  return SSX_OK;
}

ssx_status ssx_stereo_frame(ssx_ctx* ctx, const uint8_t* imgL, const uint8_t* imgR, int32_t stride, int32_t rows,
                            int32_t cols, const ssx_orb_params* orb, const ssx_match_params* mp, const ssx_stereo_rig* rig,
                            const double* T_wc, ssx_stereo_frame_out* out)
{
  if (!ctx || !imgL || !imgR || !orb || !mp || !rig || !out) return SSX_ERR_INVALID_ARG;
  ssx_status st = plan(ctx, rows, cols, 2, *orb, false, false);
  if (st != SSX_OK) return st;
  OrbWorkspace* ws = get_ws(ctx);
  const int pitch = (cols + 7) & ~7;   // 8-byte rows: k_copy_level0 reads the staging copy with one load per 8 pixels
  const size_t bytes = (size_t)rows * pitch;
  SSX_HIP_TRY(ctx, ws->stage.reserve(2 * bytes + 512));
  uint8_t* hs = ws->stage.as<uint8_t>();
  for (int y = 0; y < rows; ++y) {
    memcpy(hs + (size_t)y * pitch, imgL + (size_t)y * stride, cols);
    memcpy(hs + bytes + (size_t)y * pitch, imgR + (size_t)y * stride, cols);
  }
  // level 0 is built straight from the pinned staging copy (k_copy_level0 reads it over PCIe once): no separate upload
  st = stage_level0(ctx, hs, pitch, bytes, nullptr, 0, 0);
  if (st != SSX_OK) return st;
  st = run_pipeline(ctx);
  if (st != SSX_OK) return st;
  MatchDev m{};
  st = make_match_dev(ctx, 1, *mp, *rig, T_wc, m);
  if (st != SSX_OK) return st;
  size_t fetch_bytes = 0;
  const FrameSlots slots = frame_slots(ws->dev.out_cap, &fetch_bytes);
  SSX_HIP_TRY(ctx, ws->fetch.reserve(fetch_bytes));
  st = launch_stereo(ctx, m, ws->fetch.as<char>(), &slots);
  if (st != SSX_OK) return st;
  return fetch_frame_fast(ctx, slots, out);
}

ssx_status ssx_stereo_batch_enqueue(ssx_ctx* ctx)
{
  if (!ctx || !ctx->orb || !ctx->orb->batch_imgs) return SSX_ERR_INVALID_ARG;
  OrbWorkspace* ws = ctx->orb;
  const size_t img_bytes = (size_t)ws->rows * ws->batch_stride;
  ssx_status st = stage_level0(ctx, ws->batch_imgs, ws->batch_stride, img_bytes, nullptr, 0, 0);
  if (st != SSX_OK) return st;
  st = run_pipeline(ctx);
  if (st != SSX_OK) return st;
  MatchDev m{};
  st = make_match_dev(ctx, ws->batch_pairs, ws->batch_mp, ws->batch_rig, nullptr, m);
  if (st != SSX_OK) return st;
  return launch_stereo(ctx, m);
}

ssx_status ssx_stereo_batch_dev(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_dev, int32_t stride, int32_t rows,
                                int32_t cols, const ssx_orb_params* orb, const ssx_match_params* mp,
                                const ssx_stereo_rig* rig, int32_t* counts_out)
{
  if (!ctx || pairs < 1 || !imgs_dev || !orb || !mp || !rig || stride < cols) return SSX_ERR_INVALID_ARG;
  ssx_status st = plan(ctx, rows, cols, 2 * pairs, *orb, false, false);
  if (st != SSX_OK) return st;
  OrbWorkspace* ws = get_ws(ctx);
  ws->batch_imgs = imgs_dev; ws->batch_pairs = pairs; ws->batch_stride = stride;
  ws->batch_orb = *orb; ws->batch_mp = *mp; ws->batch_rig = *rig;
  st = ssx_stereo_batch_enqueue(ctx);
  if (st != SSX_OK) return st;
  if (counts_out) {
    std::vector<int> status(2 * (size_t)pairs);
    SSX_HIP_TRY(ctx, hipMemcpyAsync(counts_out, ws->pair_counts, sizeof(int) * 4 * pairs, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipMemcpyAsync(status.data(), ws->dev.status, sizeof(int) * 2 * pairs, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2 * pairs; ++i)
      if (status[i]) { ctx->set_error("ssx_stereo_batch: internal capacity exceeded on image %d (bits %d)", i, status[i]); return SSX_ERR_CAPACITY; }
  }
  return SSX_OK;
}

// Batches that arrive from the HOST (test/test_system.cpp:36-47: the reference's loop reads a pair from disk and hands it to
// System::RunStep at every step).  The images go up on a copy stream of the ctx's own into one of two device buffers;
// a batch's pipeline waits for its own upload only.
//   ssx_stereo_batch_upload   start the upload of a batch (at most two may be pending: the one being processed next and the
//                             one after it) and return: the copy runs beside whatever the GPU is doing
//   ssx_stereo_batch_run      enqueue the pipeline on the OLDEST uploaded batch (no synchronisation)
//   ssx_stereo_batch_host     upload + run in one call
//   ssx_stereo_batch_counts   wait for the OLDEST batch that was run and not collected yet (at most two wait) and return its counts
//   ssx_stereo_batch_fetch    one pair's arrays of the batch run LAST (after run(k + 1); counts() -> k the two name different batches)
// A server keeps one upload ahead: upload(k + 1); run(k); ...; counts(k) -- batch k + 1 crosses PCIe while batch k's kernels run.
static ssx_status batch_ingest_init(ssx_ctx* ctx, OrbWorkspace* ws)
{
  if (ws->copy_stream) return SSX_OK;
  SSX_HIP_TRY(ctx, ctx->make_stream(&ws->copy_stream, false));
  for (int b = 0; b < 2; ++b) {
    SSX_HIP_TRY(ctx, hipEventCreateWithFlags(&ws->ev_up[b], hipEventDisableTiming));
    SSX_HIP_TRY(ctx, hipEventCreateWithFlags(&ws->ev_free[b], hipEventDisableTiming));
  }
  return SSX_OK;
}

ssx_status ssx_stereo_batch_upload(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_host, int32_t stride, int32_t rows, int32_t cols)
{
  if (!ctx || pairs < 1 || !imgs_host || stride < cols || rows < 1) return SSX_ERR_INVALID_ARG;
  OrbWorkspace* ws = get_ws(ctx);
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ssx_status st = batch_ingest_init(ctx, ws);
  if (st != SSX_OK) return st;
  if (ws->up_count >= 2) { ctx->set_error("ssx_stereo_batch_upload: two uploaded batches are waiting for ssx_stereo_batch_run already"); return SSX_ERR_INVALID_ARG; }
  const int b = (ws->up_first + ws->up_count) & 1;
  const size_t bytes = 2 * (size_t)pairs * (size_t)rows * stride;
  if (bytes > ws->ingest[b].cap) {                                    // (grows once; the batch that used the buffer last may still be read)
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    SSX_HIP_TRY(ctx, ws->ingest[b].reserve(bytes));
    ws->free_pending[b] = false;
  }
  if (ws->free_pending[b]) SSX_HIP_TRY(ctx, hipStreamWaitEvent(ws->copy_stream, ws->ev_free[b], 0));   // level 0 of the batch that used it is staged
  SSX_HIP_TRY(ctx, hipMemcpyAsync(ws->ingest[b].p, imgs_host, bytes, hipMemcpyHostToDevice, ws->copy_stream));
  SSX_HIP_TRY(ctx, hipEventRecord(ws->ev_up[b], ws->copy_stream));
  ws->up_shape[b][0] = pairs; ws->up_shape[b][1] = stride; ws->up_shape[b][2] = rows; ws->up_shape[b][3] = cols;
  ws->up_count++;
  return SSX_OK;
}

ssx_status ssx_stereo_batch_run(ssx_ctx* ctx, const ssx_orb_params* orb, const ssx_match_params* mp, const ssx_stereo_rig* rig)
{
  if (!ctx || !orb || !mp || !rig) return SSX_ERR_INVALID_ARG;
  OrbWorkspace* ws = get_ws(ctx);
  if (ws->up_count < 1) { ctx->set_error("ssx_stereo_batch_run: no uploaded batch (ssx_stereo_batch_upload first)"); return SSX_ERR_INVALID_ARG; }
  const int b = ws->up_first;
  const int pairs = ws->up_shape[b][0], stride = ws->up_shape[b][1], rows = ws->up_shape[b][2], cols = ws->up_shape[b][3];
  ssx_status st = plan(ctx, rows, cols, 2 * pairs, *orb, false, false);
  if (st != SSX_OK) return st;
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ws->cnt_count >= 2) { ctx->set_error("ssx_stereo_batch_run: the counts of two batches are waiting for ssx_stereo_batch_counts already"); return SSX_ERR_INVALID_ARG; }
  if (ws->cnt_count > 0 && ws->cnt_pairs[ws->cnt_first] != pairs) { ctx->set_error("ssx_stereo_batch_run: collect the counts of the batches that were run before changing the batch size"); return SSX_ERR_INVALID_ARG; }
  SSX_HIP_TRY(ctx, ws->counts_pinned.reserve(2 * sizeof(int) * 6 * (size_t)pairs));
  for (int q = 0; q < 2; ++q) if (!ws->ev_counts[q]) SSX_HIP_TRY(ctx, hipEventCreateWithFlags(&ws->ev_counts[q], hipEventDisableTiming));
  SSX_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ws->ev_up[b], 0));
  const size_t img_bytes = (size_t)rows * stride;
  ws->batch_imgs = ws->ingest[b].as<uint8_t>(); ws->batch_pairs = pairs; ws->batch_stride = stride;
  ws->batch_orb = *orb; ws->batch_mp = *mp; ws->batch_rig = *rig;
  st = stage_level0(ctx, ws->batch_imgs, stride, img_bytes, nullptr, 0, 0);
  if (st != SSX_OK) return st;
  SSX_HIP_TRY(ctx, hipEventRecord(ws->ev_free[b], ctx->stream));
  ws->free_pending[b] = true;
  ws->up_first ^= 1; ws->up_count--;
  // From here on the batch is CONSUMED (its buffer is promised to the next upload): whatever happens below, it takes its slot in
  // the counts FIFO, so a caller that pipelines upload / run / counts stays aligned -- a failure is reported by this call AND by
  // the ssx_stereo_batch_counts call that collects this batch, never as another batch's counts.
  const int slot = (ws->cnt_first + ws->cnt_count) & 1;
  auto rest = [&]() -> ssx_status {
    ssx_status r = run_pipeline(ctx);
    if (r != SSX_OK) return r;
    MatchDev m{};
    r = make_match_dev(ctx, pairs, *mp, *rig, nullptr, m);
    if (r != SSX_OK) return r;
    r = launch_stereo(ctx, m);
    if (r != SSX_OK) return r;
    // the counts of this batch leave the device before the next batch's kernels overwrite them (stream order); an event per batch
    // lets ssx_stereo_batch_counts wait for THIS batch only -- a caller may run the next batch first and collect one batch behind
    int* hc = ws->counts_pinned.as<int>() + (size_t)slot * 6 * pairs;
    SSX_HIP_TRY(ctx, hipMemcpyAsync(hc, ws->pair_counts, sizeof(int) * 4 * pairs, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipMemcpyAsync(hc + 4 * (size_t)pairs, ws->dev.status, sizeof(int) * 2 * pairs, hipMemcpyDeviceToHost, ctx->stream));
    return SSX_OK;
  };
  st = rest();
  const hipError_t ev_err = hipEventRecord(ws->ev_counts[slot], ctx->stream);
  if (st == SSX_OK && ev_err != hipSuccess) { ctx->set_error("ssx_stereo_batch_run: hipEventRecord: %s", hipGetErrorString(ev_err)); st = SSX_ERR_HIP; }
  ws->cnt_pairs[slot] = pairs;
  ws->cnt_fail[slot] = st;
  ws->cnt_count++;
  return st;
}

ssx_status ssx_stereo_batch_host(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_host, int32_t stride, int32_t rows, int32_t cols,
                                 const ssx_orb_params* orb, const ssx_match_params* mp, const ssx_stereo_rig* rig)
{
  if (!ctx || !orb || !mp || !rig) return SSX_ERR_INVALID_ARG;
  OrbWorkspace* ws = get_ws(ctx);
  if (ws->up_count != 0) { ctx->set_error("ssx_stereo_batch_host: %d uploaded batch(es) are waiting for ssx_stereo_batch_run", ws->up_count); return SSX_ERR_INVALID_ARG; }
  ssx_status st = ssx_stereo_batch_upload(ctx, pairs, imgs_host, stride, rows, cols);
  if (st != SSX_OK) return st;
  return ssx_stereo_batch_run(ctx, orb, mp, rig);
}

// the counts of the OLDEST batch that was run and not collected yet (at most two can be waiting): waits for that batch only
ssx_status ssx_stereo_batch_counts(ssx_ctx* ctx, int32_t* counts_out)
{
  if (!ctx || !ctx->orb) return SSX_ERR_INVALID_ARG;
  OrbWorkspace* ws = ctx->orb;
  if (ws->cnt_count < 1) { ctx->set_error("ssx_stereo_batch_counts: no batch has been run (ssx_stereo_batch_run first)"); return SSX_ERR_INVALID_ARG; }
  const int slot = ws->cnt_first;
  const hipError_t sync_err = hipEventSynchronize(ws->ev_counts[slot]);
  ws->cnt_first ^= 1; ws->cnt_count--;                               // (collected either way: the FIFO stays aligned with the runs)
  if (ws->cnt_fail[slot] != SSX_OK) { ctx->set_error("ssx_stereo_batch_counts: the run of this batch failed (status %d)", (int)ws->cnt_fail[slot]); return ws->cnt_fail[slot]; }
  SSX_HIP_TRY(ctx, sync_err);
  const int pairs = ws->cnt_pairs[slot];
  const int* hc = ws->counts_pinned.as<int>() + (size_t)slot * 6 * pairs;
  for (int i = 0; i < 2 * pairs; ++i)
    if (hc[4 * (size_t)pairs + i]) { ctx->set_error("ssx_stereo_batch: internal capacity exceeded on image %d (bits %d)", i, hc[4 * (size_t)pairs + i]); return SSX_ERR_CAPACITY; }
  if (counts_out) memcpy(counts_out, hc, sizeof(int) * 4 * (size_t)pairs);
  return SSX_OK;
}

ssx_status ssx_stereo_batch_fetch(ssx_ctx* ctx, int32_t pair, ssx_stereo_frame_out* out)
{
  if (!ctx || !ctx->orb || !out || pair < 0 || pair >= ctx->orb->batch_pairs) return SSX_ERR_INVALID_ARG;
  return fetch_pair(ctx, pair, out);
}

}  // extern "C"
