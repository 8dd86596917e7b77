// ssvio_amd/csrc/lk.hip -- pyramidal Lucas-Kanade tracker on gfx950.
//
// Replaces the two cv::calcOpticalFlowPyrLK calls of the reference front-end
// (/root/reference/src/ssvio/frontend.cpp:156-166 TrackLastFrame, :374-384 FindFeaturesInRight; 11x11 window,
// maxLevel 3, TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW).  The arithmetic is the fixed-point
// scheme of OpenCV 3.x lkpyramid.cpp (14-bit bilinear weights, Scharr derivatives in int16, patch values scaled
// by 32) with ONE documented difference, shared with the CPU oracle (oracle/src/lk_oracle.cpp): the 2x2 normal
// matrix and the mismatch vector are sums of integer products and are accumulated EXACTLY in 64-bit integers (then
// converted to float once) instead of in float -- the result does not depend on the order of the parallel
// reduction, so the device is bit-identical to the CPU restatement.
//
//   k_lk_pyramid      calls of <= 16 jobs: padded pyramid + derivative images of one image in ONE launch (LDS tiles with halo)
//   k_lk_pad_level0   (wider calls, > 4 levels) image -> level 0 with a BORDER_REFLECT_101 border of win+1 pixels (what
//                     buildOpticalFlowPyramid's copyMakeBorder produces), pitch a multiple of 64
//   k_lk_pyr_down     level l-1 -> level l (+ border): 5x5 binomial, (sum + 128) >> 8, one thread per padded pixel
//   k_lk_scharr       Scharr dx|dy packed as int16x2 per pixel, ZERO border (BORDER_CONSTANT, as calcOpticalFlowPyrLK)
//   k_lk_track        ONE WAVE PER POINT, all pyramid levels inside the kernel: lane l owns window pixels l and
//                     l + 64 (121 of 128 slots for 11x11); the template patch and its derivatives stay in
//                     registers; every iteration is 2 x 4 bilinear taps per lane, one 64-bit wave reduction of
//                     the mismatch vector (xor butterfly), and the 2x2 solve done redundantly by every lane.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "../../include/ssx_test_hooks.h"

namespace {

constexpr int LK_MAX_LEVELS = 8;
constexpr int LK_MAX_WIN = 15;          // win * win <= 4 slots x 64 lanes would allow 16; 15 keeps the border small

struct LkDev {
  int levels;                            // pyramid levels built (top level = levels - 1)
  int win, pad;                          // window side, border = win + 1
  int rows[LK_MAX_LEVELS], cols[LK_MAX_LEVELS], pitch[LK_MAX_LEVELS];
  size_t off[LK_MAX_LEVELS];             // byte offset of a padded level inside one image pyramid
  size_t doff[LK_MAX_LEVELS];            // int32 offset of a padded derivative level
  // tracking
  int max_iters, use_initial_flow;
  double eps2;
  float min_eig;
  const float* prev_pts;                 // the points of ALL jobs of the call, job after job (LkJob::pt0)
  float* next_pts;
  uint8_t* status;
  float* err;
};

// One tracking job of a call (blockIdx.z / .y = job): its two pyramids + derivative images (a SLOT of the context: a stream keeps
// its slot, so the pyramid of its last `next` image is still there), the level-0 sources and its points.  ssx_lk_track /
// ssx_lk_track_next are a call with one job on slot 0.
struct LkJob {
  uint8_t* pyr[2];                       // [0] previous image, [1] next image
  uint32_t* deriv;                       // previous image: dx | dy << 16 (int16 each), same padded geometry
  uint32_t* deriv1;                      // the same of the next image, written by k_lk_pyramid for the job that will be chained to this one
  const uint8_t* img[2];                 // level-0 sources (device memory, or pinned host memory read over PCIe); img[0] null: chained
  int stride[2];
  int n, pt0;
};

struct LkSlot {
  DevBuf mem;
  uint8_t* pyr[2] = {nullptr, nullptr};
  uint32_t* deriv = nullptr, *deriv1 = nullptr;
  bool have_next = false;                // pyr[1] holds the pyramid of the slot's last `next` image
  bool have_next_deriv = false;          // ... and deriv1 its derivative images (fused pyramid kernel)
};

struct LkWorkspace {
  std::vector<LkSlot> slots;
  DevBuf io;
  HostBuf stage;
  LkDev dev{};
  size_t pyr_bytes = 0, deriv_words = 0;
  int rows = 0, cols = 0, win = 0, max_level = -1;
  bool planned = false;
  bool fused = false;                    // k_lk_pyramid builds the pyramids (<= 4 levels, border <= 16)
};

void lk_ws_free(void* p)
{
  LkWorkspace* w = static_cast<LkWorkspace*>(p);
  if (!w) return;
  for (LkSlot& sl : w->slots) sl.mem.release();
  w->io.release(); w->stage.release();
  delete w;
}

LkWorkspace* lk_ws(ssx_ctx* ctx)
{
  if (!ctx->lk) { ctx->lk = new LkWorkspace(); ctx->lk_free = lk_ws_free; }
  return static_cast<LkWorkspace*>(ctx->lk);
}

__device__ __forceinline__ int refl101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void k_lk_pad_level0(LkDev d, const LkJob* __restrict__ jobs, int which)
{
  const LkJob jb = jobs[blockIdx.z];
  const uint8_t* img = jb.img[which];
  if (!img) return;                                                   // chained job: its previous pyramid is resident
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;      // padded coordinates
  const int pw = d.cols[0] + 2 * d.pad;
  if (x >= pw) return;
  const int sy = refl101(y - d.pad, d.rows[0]), sx = refl101(x - d.pad, d.cols[0]);
  jb.pyr[which][d.off[0] + (size_t)y * d.pitch[0] + x] = img[(size_t)sy * jb.stride[which] + sx];
}

// pyrDown (imgproc/pyramids.cpp, 8u): separable 1-4-6-4-1, (sum + 128) >> 8, BORDER_REFLECT_101; the padded
// border of the source level already holds the reflected pixels.  Border pixels of the destination are the
// reflected destination pixels, computed redundantly.
__global__ __launch_bounds__(256) void k_lk_pyr_down(LkDev d, const LkJob* __restrict__ jobs, int which, int level)
{
  const LkJob jb = jobs[blockIdx.z];
  if (!jb.img[which]) return;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  const int pw = d.cols[level] + 2 * d.pad;
  if (x >= pw) return;
  const int dy = refl101(y - d.pad, d.rows[level]), dx = refl101(x - d.pad, d.cols[level]);
  const uint8_t* s = jb.pyr[which] + d.off[level - 1];
  const int sp = d.pitch[level - 1];
  const int w[5] = {1, 4, 6, 4, 1};
  int sum = 0;
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    const uint8_t* row = s + (size_t)(2 * dy + ky - 2 + d.pad) * sp + (2 * dx - 2 + d.pad);
    int r = 0;
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) r += w[kx] * row[kx];
    sum += w[ky] * r;
  }
  jb.pyr[which][d.off[level] + (size_t)y * d.pitch[level] + x] = (uint8_t)((sum + 128) >> 8);
}

// calcSharrDeriv (video/lkpyramid.cpp): dx = [3 10 3]^T (x) [-1 0 1], dy = [-1 0 1]^T (x) [3 10 3]; rows and
// columns reflect (101) at the image edge = the padded border; the derivative image itself has a zero border.
__global__ __launch_bounds__(256) void k_lk_scharr(LkDev d, const LkJob* __restrict__ jobs, int level)
{
  const LkJob jb = jobs[blockIdx.z];
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;      // padded coordinates
  const int pw = d.cols[level] + 2 * d.pad;
  if (x >= pw) return;
  const int iy = y - d.pad, ix = x - d.pad;
  uint32_t out = 0;
  if (iy >= 0 && iy < d.rows[level] && ix >= 0 && ix < d.cols[level]) {
    const uint8_t* c = jb.pyr[0] + d.off[level] + (size_t)y * d.pitch[level] + x;
    const int p = d.pitch[level];
    const int a00 = c[-p - 1], a01 = c[-p], a02 = c[-p + 1], a10 = c[-1], a12 = c[1], a20 = c[p - 1], a21 = c[p], a22 = c[p + 1];
    const int gx = ((a02 + a22) * 3 + a12 * 10) - ((a00 + a20) * 3 + a10 * 10);
    const int gy = ((a20 - a00) + (a22 - a02)) * 3 + (a21 - a01) * 10;
    out = ((uint32_t)gx & 0xFFFFu) | ((uint32_t)gy << 16);
  }
  jb.deriv[d.doff[level] + (size_t)y * d.pitch[level] + x] = out;
}

// ---- one launch per image instead of 1 + (levels - 1) + levels --------------------------------------------------------------
// k_lk_pyramid: the padded pyramid AND the derivative images of one image in ONE launch (k_lk_pad_level0, k_lk_pyr_down per level
// and k_lk_scharr per level are eight dependent launches of 5 - 17 us each for a 1241x376 frame: 78 us of kernels + their
// boundaries, against the ~0.1 ms the tracking kernel itself takes).  A workgroup owns FT x FT pixels of the PADDED top level and,
// nested below them, the 2^k times larger blocks of the padded lower levels (ext coordinate e = padded coordinate - border; level
// l - 1 block = [2 lo, 2 hi + 1]).  It reads the patch of the source image everything it owns depends on (101 x 101 pixels for four
// levels) into LDS once, and goes up level by level inside LDS: outputs of level l (padded pixels = BORDER_REFLECT_101 of the level,
// Scharr pair with a zero border) from the level's LDS patch, then the patch of level l + 1 by the 5x5 binomial, (sum + 128) >> 8.
// The patches hold IMAGE coordinates; every read folds its coordinate (refl101), as the kernels it replaces do through the padded
// borders -- same integers, same results (tests/test_lk_gpu.py compares every level and derivative image with the oracle's).
// The halo is recomputed per workgroup (level 0 is read 2.5 times, in L2), nothing crosses workgroups.
constexpr int FT = 8;
constexpr int FUSED_MAX_LEVELS = 4;
constexpr int FUSED_MAX_JOBS = 16;
constexpr int FD_0 = FT + 2, FD_1 = 2 * FD_0 + 3, FD_2 = 2 * FD_1 + 3, FD_3 = 2 * FD_2 + 3;       // patch sides by distance from the top: 10, 23, 49, 101
constexpr int FOFF_0 = 0, FOFF_1 = FOFF_0 + ((FD_0 * FD_0 + 15) & ~15), FOFF_2 = FOFF_1 + ((FD_1 * FD_1 + 15) & ~15), FOFF_3 = FOFF_2 + ((FD_2 * FD_2 + 15) & ~15);
constexpr int FUSED_LDS = FOFF_3 + ((FD_3 * FD_3 + 15) & ~15);
static_assert(FD_3 <= 128, "k_lk_pyramid loads its level-0 patch one column per thread of half a workgroup");

struct Iv { int lo, hi; };                                             // closed interval; lo > hi = empty
__device__ __forceinline__ Iv iv_hull(Iv a, Iv b)
{
  if (a.lo > a.hi) return b;
  if (b.lo > b.hi) return a;
  return Iv{min(a.lo, b.lo), max(a.hi, b.hi)};
}
// the image coordinates the ext interval folds onto (BORDER_REFLECT_101 on [0, n)), as one interval
__device__ __forceinline__ Iv iv_fold(Iv e, int n)
{
  if (e.lo > e.hi) return e;
  if (n == 1) return Iv{0, 0};
  if (e.lo < -(n - 1) || e.hi > 2 * (n - 1)) return Iv{0, n - 1};       // (more than one fold: tiny levels under a wide border)
  Iv r{max(e.lo, 0), min(e.hi, n - 1)};
  if (e.lo < 0) r = iv_hull(r, Iv{max(1, -e.hi), -e.lo});
  if (e.hi > n - 1) r = iv_hull(r, Iv{2 * (n - 1) - e.hi, 2 * (n - 1) - max(e.lo, n)});
  return Iv{max(r.lo, 0), min(r.hi, n - 1)};
}
__device__ __forceinline__ int fold1(int i, int n) { return (unsigned)i < (unsigned)n ? i : refl101(i, n); }

__global__ __launch_bounds__(256) void k_lk_pyramid(LkDev d, const LkJob* __restrict__ jobs)
{
  __shared__ __attribute__((aligned(16))) uint8_t sP[FUSED_LDS];
  const LkJob& jb = jobs[blockIdx.z >> 1];                             // by reference: a private copy indexed by `which` lives in scratch memory
  const int which = blockIdx.z & 1;
  const uint8_t* img = jb.img[which];
  if (!img) return;                                                   // chained job: the previous image's pyramid and derivatives are resident
  uint8_t* pyr = jb.pyr[which];
  uint32_t* der = which ? jb.deriv1 : jb.deriv;
  const int top = d.levels - 1, pad = d.pad, t = threadIdx.x, tx = t & 63, ty = t >> 6;
  // what this workgroup owns (ext coordinates, clipped to the padded level) and the patch (image coordinates) it needs, per level
  Iv ox[FUSED_MAX_LEVELS], oy[FUSED_MAX_LEVELS], vx[FUSED_MAX_LEVELS], vy[FUSED_MAX_LEVELS];
  {
    const int lx = (int)blockIdx.x * FT - pad, ly = (int)blockIdx.y * FT - pad;
#pragma unroll
    for (int l = FUSED_MAX_LEVELS - 1; l >= 0; --l) {
      if (l > top) continue;
      const int k = top - l, f = 1 << k;
      ox[l] = Iv{max(lx * f, -pad), min((lx + FT) * f - 1, d.cols[l] - 1 + pad)};
      oy[l] = Iv{max(ly * f, -pad), min((ly + FT) * f - 1, d.rows[l] - 1 + pad)};
      if (ox[l].lo > ox[l].hi || oy[l].lo > oy[l].hi) { ox[l] = Iv{0, -1}; oy[l] = Iv{0, -1}; }
      const bool own = ox[l].lo <= ox[l].hi;
      Iv nx = own ? iv_fold(Iv{ox[l].lo - 1, ox[l].hi + 1}, d.cols[l]) : Iv{0, -1};
      Iv ny = own ? iv_fold(Iv{oy[l].lo - 1, oy[l].hi + 1}, d.rows[l]) : Iv{0, -1};
      if (l < top && vx[l + 1].lo <= vx[l + 1].hi) {
        nx = iv_hull(nx, iv_fold(Iv{2 * vx[l + 1].lo - 2, 2 * vx[l + 1].hi + 2}, d.cols[l]));
        ny = iv_hull(ny, iv_fold(Iv{2 * vy[l + 1].lo - 2, 2 * vy[l + 1].hi + 2}, d.rows[l]));
      }
      vx[l] = nx; vy[l] = ny;
    }
  }
  if (vx[0].lo > vx[0].hi) return;                                    // (a tile beyond the padded top level owns nothing)
  auto patch = [&](int l) -> uint8_t* {
    const int k = top - l;
    return sP + (k == 0 ? FOFF_0 : k == 1 ? FOFF_1 : k == 2 ? FOFF_2 : FOFF_3);
  };
  // level 0 patch <- the image
  {
    const int W = vx[0].hi - vx[0].lo + 1, H = vy[0].hi - vy[0].lo + 1;
    uint8_t* S = patch(0);
    const uint8_t* src = img + (size_t)vy[0].lo * jb.stride[which] + vx[0].lo;
    // (thread = one column of the patch, every second row; 16 loads in flight per round: a loop of load -> store pairs waits a memory
    // round trip per pixel, 40 of them in a row)
    const int x = t & 127, y0 = t >> 7;
    const bool xin = x < W;
    for (int j0 = 0; j0 < H; j0 += 32) {
      uint8_t v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int y = j0 + y0 + 2 * u;
        v[u] = (xin && y < H) ? src[(size_t)y * jb.stride[which] + x] : (uint8_t)0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int y = j0 + y0 + 2 * u;
        if (xin && y < H) S[y * W + x] = v[u];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < FUSED_MAX_LEVELS; ++l) {
    if (l > top) continue;
    const int W = vx[l].hi - vx[l].lo + 1;
    const int rows = d.rows[l], cols = d.cols[l], pitch = d.pitch[l];
    const uint8_t* S = patch(l);
    // the owned padded pixels of level l and their Scharr pairs
    if (ox[l].lo <= ox[l].hi) {
      uint8_t* P = pyr + d.off[l];
      uint32_t* G = der + d.doff[l];
      for (int ey = oy[l].lo + ty; ey <= oy[l].hi; ey += 4) {
        const int iy = fold1(ey, rows), ym = fold1(iy - 1, rows), yp = fold1(iy + 1, rows);
        const uint8_t* r0 = S + (ym - vy[l].lo) * W - vx[l].lo;
        const uint8_t* r1 = S + (iy - vy[l].lo) * W - vx[l].lo;
        const uint8_t* r2 = S + (yp - vy[l].lo) * W - vx[l].lo;
        for (int ex = ox[l].lo + tx; ex <= ox[l].hi; ex += 64) {
          const int ix = fold1(ex, cols);
          const size_t o = (size_t)(ey + pad) * pitch + (ex + pad);
          P[o] = r1[ix];
          uint32_t g = 0;
          if (ey == iy && ex == ix) {                                  // inside the image: the derivative; the border of G is zero
            const int xm = fold1(ix - 1, cols), xp = fold1(ix + 1, cols);
            const int a00 = r0[xm], a01 = r0[ix], a02 = r0[xp], a10 = r1[xm], a12 = r1[xp], a20 = r2[xm], a21 = r2[ix], a22 = r2[xp];
            const int gx = ((a02 + a22) * 3 + a12 * 10) - ((a00 + a20) * 3 + a10 * 10);
            const int gy = ((a20 - a00) + (a22 - a02)) * 3 + (a21 - a01) * 10;
            g = ((uint32_t)gx & 0xFFFFu) | ((uint32_t)gy << 16);
          }
          G[o] = g;
        }
      }
    }
    // the patch of level l + 1
    if (l < top) {
      const int W1 = vx[l + 1].hi - vx[l + 1].lo + 1, H1 = vy[l + 1].hi - vy[l + 1].lo + 1;
      uint8_t* D = patch(l + 1);
      for (int y = ty; y < H1; y += 4) {
        const int cy = 2 * (vy[l + 1].lo + y);
        const uint8_t* rr[5];
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) rr[ky] = S + (fold1(cy + ky - 2, rows) - vy[l].lo) * W - vx[l].lo;
        for (int x = tx; x < W1; x += 64) {
          const int cx = 2 * (vx[l + 1].lo + x);
          int xi[5];
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) xi[kx] = fold1(cx + kx - 2, cols);
          int sum = 0;
#pragma unroll
          for (int ky = 0; ky < 5; ++ky) {
            const uint8_t* r = rr[ky];
            const int h = r[xi[0]] + r[xi[4]] + 4 * (r[xi[1]] + r[xi[3]]) + 6 * r[xi[2]];
            sum += (ky == 0 || ky == 4) ? h : (ky == 2 ? 6 * h : 4 * h);
          }
          D[y * W1 + x] = (uint8_t)((sum + 128) >> 8);
        }
      }
      __syncthreads();
    }
  }
}

// exact 64-bit sum over the wave, the same value in every lane.  Lane exchanges inside a row of 16 run on the VALU (DPP:
// ~10 cycles a step instead of a ~100-cycle trip through the LDS crossbar per __shfl), the four row totals are read with
// v_readlane; integer addition is associative, so the order does not matter.
template <int CTRL>
__device__ __forceinline__ long long dpp_i64(long long v)
{
  int lo = (int)(unsigned)(unsigned long long)v, hi = (int)(unsigned)((unsigned long long)v >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long read_lane_i64(long long v, int lane)
{
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_sum(long long v)
{
  v += dpp_i64<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_i64<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_i64<0x141>(v);     // row_half_mirror
  v += dpp_i64<0x140>(v);     // row_mirror
  return (read_lane_i64(v, 0) + read_lane_i64(v, 16)) + (read_lane_i64(v, 32) + read_lane_i64(v, 48));
}
__device__ __forceinline__ int cv_floor(float v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

struct Weights { int w00, w01, w10, w11; };
__device__ __forceinline__ Weights bilinear(float a, float b)
{
  Weights w;
  w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
  w.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
  w.w10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
  w.w11 = (1 << 14) - w.w00 - w.w01 - w.w10;
  return w;
}

constexpr int SLOTS = 4;     // window pixels per lane (win <= 15 -> 225 <= 256)
constexpr int RG = 32;       // side of the per-wave LDS copy of the search neighbourhood in the next image

// The iterations of one level read a (win+1)^2 window of J whose position moves by a fraction of a pixel per step:
// a 32x32 byte neighbourhood around the starting position is copied to LDS once (16 loads per lane, all in flight
// together) and every iteration reads from there (LDS latency instead of an L2 round trip on the critical path of
// each of up to 30 dependent iterations); the copy is redone only when the window leaves it.  Pixels outside the
// padded image are never part of a valid window and are written as 0.
struct Region { int ox, oy; };
__device__ __forceinline__ Region stage_region(uint8_t* sR, const uint8_t* J, int pitch, int rows, int cols, int pad, int win, int inx, int iny, int lane)
{
  const int margin = (RG - (win + 1)) >> 1;
  Region r{inx - margin, iny - margin};
  __builtin_amdgcn_wave_barrier();                 // earlier reads of the old copy are done
  uint8_t v[RG * RG / 64];
#pragma unroll
  for (int k = 0; k < RG * RG / 64; ++k) {
    const int idx = lane + 64 * k, y = r.oy + (idx >> 5), x = r.ox + (idx & (RG - 1));
    const bool in = x >= -pad && x < cols + pad && y >= -pad && y < rows + pad;
    v[k] = in ? J[(size_t)(y + pad) * pitch + (x + pad)] : (uint8_t)0;
  }
#pragma unroll
  for (int k = 0; k < RG * RG / 64; ++k) sR[lane + 64 * k] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  return r;
}
__device__ __forceinline__ bool region_holds(const Region& r, int win, int inx, int iny)
{
  return inx >= r.ox && iny >= r.oy && inx + win + 1 <= r.ox + RG && iny + win + 1 <= r.oy + RG;
}

__global__ __launch_bounds__(256) void k_lk_track(LkDev d, const LkJob* __restrict__ jobs)
{
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
  __shared__ uint8_t sRegion[4][RG * RG];
  const LkJob jb = jobs[blockIdx.y];
  const int ij = blockIdx.x * 4 + wave;                               // point of the job
  if (ij >= jb.n) return;
  const int i = jb.pt0 + ij;                                          // ... and of the call
  uint8_t* sR = sRegion[wave];
  const int win = d.win, nwin = win * win, pad = d.pad;
  const float half = (float)(win - 1) * 0.5f;
  int wy[SLOTS], wx[SLOTS];
  bool on[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = lane + 64 * s;
    on[s] = idx < nwin;
    wy[s] = on[s] ? idx / win : 0;
    wx[s] = on[s] ? idx - wy[s] * win : 0;
  }
  const float ppx = d.prev_pts[2 * i], ppy = d.prev_pts[2 * i + 1];
  float outx = d.next_pts[2 * i], outy = d.next_pts[2 * i + 1];       // nextPts[ptidx] as the levels go by
  int status = 1;
  float errv = 0.f;
  const int top = d.levels - 1;
  for (int level = top; level >= 0; --level) {
    const uint8_t* I = jb.pyr[0] + d.off[level];
    const uint8_t* J = jb.pyr[1] + d.off[level];
    const uint32_t* G = jb.deriv + d.doff[level];
    const int pitch = d.pitch[level], rows = d.rows[level], cols = d.cols[level];
    const float sc = (float)(1. / (double)(1 << level));
    float px = ppx * sc, py = ppy * sc;
    float nx, ny;
    if (level == top) {
      if (d.use_initial_flow) { nx = outx * sc; ny = outy * sc; }
      else { nx = px; ny = py; }
    } else { nx = outx * 2.f; ny = outy * 2.f; }
    outx = nx; outy = ny;
    px -= half; py -= half;
    const int ipx = cv_floor(px), ipy = cv_floor(py);
    // a non-finite position is outside every image (float -> int conversion of NaN / inf differs between CPU and GPU)
    if (!(isfinite(px) && isfinite(py)) || ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
      if (level == 0) { status = 0; errv = 0.f; }
      continue;
    }
    Weights w = bilinear(px - (float)ipx, py - (float)ipy);
    int Iw[SLOTS], Ix[SLOTS], Iy[SLOTS];
    long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      Iw[s] = 0; Ix[s] = 0; Iy[s] = 0;
      if (on[s]) {
        const size_t o = (size_t)(wy[s] + ipy + pad) * pitch + (wx[s] + ipx + pad);
        const uint8_t* p = I + o;
        Iw[s] = (int)(short)descale(p[0] * w.w00 + p[1] * w.w01 + p[pitch] * w.w10 + p[pitch + 1] * w.w11, 14 - 5);
        const uint32_t g00 = G[o], g01 = G[o + 1], g10 = G[o + pitch], g11 = G[o + pitch + 1];
        Ix[s] = (int)(short)descale((int)(short)(g00 & 0xFFFF) * w.w00 + (int)(short)(g01 & 0xFFFF) * w.w01 +
                                    (int)(short)(g10 & 0xFFFF) * w.w10 + (int)(short)(g11 & 0xFFFF) * w.w11, 14);
        Iy[s] = (int)(short)descale(((int)g00 >> 16) * w.w00 + ((int)g01 >> 16) * w.w01 + ((int)g10 >> 16) * w.w10 +
                                    ((int)g11 >> 16) * w.w11, 14);
        sA11 += (long long)Ix[s] * Ix[s]; sA12 += (long long)Ix[s] * Iy[s]; sA22 += (long long)Iy[s] * Iy[s];
      }
    }
    sA11 = wave_sum(sA11); sA12 = wave_sum(sA12); sA22 = wave_sum(sA22);
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < d.min_eig || D < 1.1920928955078125e-7f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    Region reg{0, 0};
    bool have_region = false;
    for (int j = 0; j < d.max_iters; ++j) {
      const int inx = cv_floor(nx), iny = cv_floor(ny);
      if (!(isfinite(nx) && isfinite(ny)) || inx < -win || inx >= cols || iny < -win || iny >= rows) {
        if (level == 0) status = 0;
        break;
      }
      if (!have_region || !region_holds(reg, win, inx, iny)) {          // wave-uniform: nx, ny are the same in every lane
        reg = stage_region(sR, J, pitch, rows, cols, pad, win, inx, iny, lane);
        have_region = true;
      }
      w = bilinear(nx - (float)inx, ny - (float)iny);
      long long sb1 = 0, sb2 = 0;
      const int rbase = (iny - reg.oy) * RG + (inx - reg.ox);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (on[s]) {
          const uint8_t* p = sR + rbase + wy[s] * RG + wx[s];
          const int diff = descale(p[0] * w.w00 + p[1] * w.w01 + p[RG] * w.w10 + p[RG + 1] * w.w11, 14 - 5) - Iw[s];
          sb1 += (long long)diff * Ix[s]; sb2 += (long long)diff * Iy[s];
        }
      sb1 = wave_sum(sb1); sb2 = wave_sum(sb2);
      const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
      const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      outx = nx + half; outy = ny + half;
      if ((double)dx * (double)dx + (double)dy * (double)dy <= d.eps2) break;
      if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
        outx -= dx * 0.5f; outy -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    if (status && level == 0) {
      const float ex = outx - half, ey = outy - half;
      const int inx = cv_floor(ex), iny = cv_floor(ey);
      if (!(isfinite(ex) && isfinite(ey)) || inx < -win || inx >= cols || iny < -win || iny >= rows) { status = 0; continue; }
      if (!have_region || !region_holds(reg, win, inx, iny)) reg = stage_region(sR, J, pitch, rows, cols, pad, win, inx, iny, lane);
      w = bilinear(ex - (float)inx, ey - (float)iny);
      long long e = 0;
      const int rbase = (iny - reg.oy) * RG + (inx - reg.ox);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (on[s]) {
          const uint8_t* p = sR + rbase + wy[s] * RG + wx[s];
          const int diff = descale(p[0] * w.w00 + p[1] * w.w01 + p[RG] * w.w10 + p[RG + 1] * w.w11, 14 - 5) - Iw[s];
          e += diff < 0 ? -diff : diff;
        }
      e = wave_sum(e);
      errv = (float)e * (1.f / (float)(32 * win * win));
    }
  }
  if (lane == 0) {
    d.next_pts[2 * i] = outx; d.next_pts[2 * i + 1] = outy;
    d.status[i] = (uint8_t)status;
    if (d.err) d.err[i] = errv;
  }
}

ssx_status lk_plan(ssx_ctx* ctx, int rows, int cols, const ssx_lk_params& prm)
{
  LkWorkspace* ws = lk_ws(ctx);
  if (prm.win < 3 || prm.win > LK_MAX_WIN || (prm.win & 1) == 0 || prm.max_level < 0 || prm.max_level >= LK_MAX_LEVELS) {
    ctx->set_error("ssx_lk: unsupported window %d (odd, 3..%d) or max_level %d (0..%d)", prm.win, LK_MAX_WIN, prm.max_level, LK_MAX_LEVELS - 1);
    return SSX_ERR_INVALID_ARG;
  }
  if (rows < 2 || cols < 2 || rows > 8192 || cols > 8192) {
    ctx->set_error("ssx_lk: image %dx%d outside the supported range", cols, rows);
    return SSX_ERR_INVALID_ARG;
  }
  if (ws->planned && ws->rows == rows && ws->cols == cols && ws->win == prm.win && ws->max_level == prm.max_level) return SSX_OK;
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  LkDev d{};
  d.win = prm.win; d.pad = prm.win + 1;
  // buildOpticalFlowPyramid: a level not larger than the window ends the pyramid
  d.levels = 1;
  d.rows[0] = rows; d.cols[0] = cols;
  for (int l = 1; l <= prm.max_level; ++l) {
    const int w = (d.cols[l - 1] + 1) / 2, h = (d.rows[l - 1] + 1) / 2;
    if (w <= prm.win || h <= prm.win) break;
    d.rows[l] = h; d.cols[l] = w; d.levels = l + 1;
  }
  size_t off = 0, doff = 0;
  for (int l = 0; l < d.levels; ++l) {
    d.pitch[l] = (d.cols[l] + 2 * d.pad + 63) & ~63;
    d.off[l] = off; d.doff[l] = doff;
    off += (size_t)d.pitch[l] * (d.rows[l] + 2 * d.pad) + 64;
    doff += (size_t)d.pitch[l] * (d.rows[l] + 2 * d.pad) + 64;
  }
  ws->pyr_bytes = (off + 255) & ~size_t(255);
  ws->deriv_words = doff;
  ws->dev = d;
  // k_lk_pyramid: at most four levels in LDS, and every level at least border + 2 pixels a side (one fold per coordinate)
  static const bool unfused = getenv("SSX_LK_UNFUSED") != nullptr;     // (A/B: the per-level kernels)
  ws->fused = !unfused && d.levels <= FUSED_MAX_LEVELS && std::min(d.rows[d.levels - 1], d.cols[d.levels - 1]) >= d.pad + 2;
  ws->rows = rows; ws->cols = cols; ws->win = prm.win; ws->max_level = prm.max_level;
  ws->planned = true;
  for (LkSlot& sl : ws->slots) { sl.have_next = sl.have_next_deriv = false; sl.pyr[0] = sl.pyr[1] = nullptr; sl.deriv = sl.deriv1 = nullptr; }   // (their memory is re-carved on use)
  return SSX_OK;
}

// the slot's two pyramids and its derivative images, carved from a buffer of its own
ssx_status lk_slot(ssx_ctx* ctx, LkWorkspace* ws, int slot, LkSlot** out)
{
  if (slot < 0 || slot >= 4096) { ctx->set_error("ssx_lk: slot %d outside 0..4095", slot); return SSX_ERR_INVALID_ARG; }
  if ((size_t)slot >= ws->slots.size()) ws->slots.resize((size_t)slot + 1);
  LkSlot& sl = ws->slots[slot];
  if (!sl.pyr[0]) {
    Layout lay;
    const size_t o_p0 = lay.take(ws->pyr_bytes), o_p1 = lay.take(ws->pyr_bytes), o_g = lay.take(sizeof(uint32_t) * ws->deriv_words);
    const size_t o_g1 = lay.take(sizeof(uint32_t) * ws->deriv_words);
    SSX_HIP_TRY(ctx, sl.mem.reserve(lay.off, 1.0));
    char* base = sl.mem.as<char>();
    sl.pyr[0] = (uint8_t*)(base + o_p0); sl.pyr[1] = (uint8_t*)(base + o_p1); sl.deriv = (uint32_t*)(base + o_g); sl.deriv1 = (uint32_t*)(base + o_g1);
    sl.have_next = sl.have_next_deriv = false;
  }
  *out = &sl;
  return SSX_OK;
}

// One tracking CALL of nj jobs (ssx_lk_job).  A job with prev == nullptr is chained: its previous image is the next image of its
// slot's last job, whose pyramid is still in pyr[1] -- the two pyramid pointers of the slot swap roles and only the new image is
// read and reduced.  images_on_device: the image pointers are readable by the GPU (device memory, or pinned host memory that the
// level-0 kernel then reads over PCIe): nothing is staged; else they go through pinned staging and one copy, with the points.
ssx_status lk_run(ssx_ctx* ctx, int nj, const ssx_lk_job* jobs, int32_t rows, int32_t cols, const ssx_lk_params* prm_in, bool images_on_device,
                  int32_t* top_level)
{
  ssx_lk_params prm;
  if (prm_in) prm = *prm_in; else ssx_lk_default_params(&prm);
  LkWorkspace* w0 = lk_ws(ctx);
  bool any_chained = false, any_fresh = false;
  for (int j = 0; j < nj; ++j) {
    const ssx_lk_job& q = jobs[j];
    if (!q.next || q.n < 0 || (q.n > 0 && (!q.prev_pts || !q.next_pts || !q.status))) return SSX_ERR_INVALID_ARG;
    if ((q.prev && q.prev_stride < cols) || q.next_stride < cols) { ctx->set_error("ssx_lk: stride smaller than the image width"); return SSX_ERR_INVALID_ARG; }
    if (q.slot < 0 || q.slot >= 4096) { ctx->set_error("ssx_lk: slot %d outside 0..4095", q.slot); return SSX_ERR_INVALID_ARG; }   // (before any slot is touched)
    for (int k = 0; k < j; ++k) if (jobs[k].slot == q.slot) { ctx->set_error("ssx_lk_track_batch: slot %d twice in one call", q.slot); return SSX_ERR_INVALID_ARG; }
    if (!q.prev) {
      any_chained = true;
      const bool ok = w0->planned && w0->rows == rows && w0->cols == cols && w0->win == prm.win && w0->max_level == prm.max_level && q.slot >= 0 &&
                      (size_t)q.slot < w0->slots.size() && w0->slots[q.slot].pyr[0] && w0->slots[q.slot].have_next;
      if (!ok) {
        ctx->set_error("ssx_lk_track_next: no previous ssx_lk_track call with the same image size, window and max_level on this context (slot %d)", q.slot);
        return SSX_ERR_INVALID_ARG;
      }
    } else any_fresh = true;
  }
  (void)any_chained;
  ssx_status st = lk_plan(ctx, rows, cols, prm);
  if (st != SSX_OK) return st;
  LkWorkspace* ws = lk_ws(ctx);
  std::vector<LkSlot*> slot(nj);
  {
    int top = -1;
    for (int j = 0; j < nj; ++j) top = std::max(top, jobs[j].slot);
    if (top >= 0 && top < 4096 && (size_t)top >= ws->slots.size()) ws->slots.resize((size_t)top + 1);   // (before any pointer into it is taken)
  }
  // One launch per image (k_lk_pyramid) for calls of a few jobs -- a single stream's frame is a chain of dependent launches, and the
  // fused kernel replaces eight of them (chained call 0.158 -> 0.122 ms, two fresh images 0.242 -> 0.151 ms); in a call of many jobs
  // the per-level kernels are wide launches already and do less work per pixel (64 jobs: 0.53 against 0.75 ms): they stay for those
  // (profiles/r06/lk_pyramid_ab.txt).
  const bool use_fused = ws->fused && nj <= FUSED_MAX_JOBS;
  bool scharr_now = !use_fused;                                       // (the derivative images of the previous image: per-level kernel, this call)
  for (int j = 0; j < nj; ++j) {
    st = lk_slot(ctx, ws, jobs[j].slot, &slot[j]);
    if (st != SSX_OK) return st;
    if (!jobs[j].prev) {
      if (!slot[j]->have_next_deriv) scharr_now = true;               // its last call was a wide one: no derivative images were kept
      std::swap(slot[j]->pyr[0], slot[j]->pyr[1]); std::swap(slot[j]->deriv, slot[j]->deriv1);
    }
    slot[j]->have_next = slot[j]->have_next_deriv = false;
  }
  LkDev d = ws->dev;
  hipStream_t s = ctx->stream;
  // one pinned block: [job table | (staged images) | prev points | next points] -> one copy -> device; results come back in one
  const size_t img_bytes = (size_t)rows * cols;
  size_t n_pts = 0;
  int max_n = 0;
  for (int j = 0; j < nj; ++j) { n_pts += (size_t)jobs[j].n; max_n = std::max(max_n, jobs[j].n); }
  Layout io;
  const size_t o_tab = io.take(sizeof(LkJob) * (size_t)nj);
  std::vector<size_t> o_img0(nj, 0), o_img1(nj, 0);
  if (!images_on_device)
    for (int j = 0; j < nj; ++j) { o_img1[j] = io.take(img_bytes); if (jobs[j].prev) o_img0[j] = io.take(img_bytes); }
  const size_t o_pp = io.take(sizeof(float) * 2 * std::max<size_t>(n_pts, 1));
  const size_t o_np = io.take(sizeof(float) * 2 * std::max<size_t>(n_pts, 1));
  const size_t in_bytes = io.off;
  const size_t o_st = io.take(std::max<size_t>(n_pts, 1));
  const size_t o_er = io.take(sizeof(float) * std::max<size_t>(n_pts, 1));
  const size_t host_end = io.off;                                    // (what has a pinned mirror: inputs by copy + results)
  // images_on_device: when the callers' `next` images lie at a constant distance from each other (one pinned arena, a slice per stream:
  // ssvio_amd/host/stream_batcher.cpp) ONE copy on the DMA engine brings them into a staging block in HBM -- 54 GB/s, against 32 GB/s for
  // the level-0 kernel reading the host pixels itself (0.95 ms of a 1.4 ms call for 64 frames; a kernel with 16-byte loads: no better --
  // profiles/r06/c5_kernel_stats.txt).  Images elsewhere are read where they lie.
  auto span = [&](int stride) { return (size_t)(rows - 1) * (size_t)stride + (size_t)cols; };
  size_t arena_bytes = 0, o_arena = 0;                              // the range [jobs[0].next, last job's end) when it is worth one copy
  if (images_on_device && nj >= 2) {                                 // (one job: stage_each below, if the image is in host memory)
    bool ok = true;
    for (int j = 1; ok && j < nj; ++j)
      ok = jobs[j].next_stride == jobs[0].next_stride && jobs[j].next > jobs[j - 1].next && (size_t)(jobs[j].next - jobs[j - 1].next) >= span(jobs[0].next_stride);
    const size_t range = ok ? (size_t)(jobs[nj - 1].next - jobs[0].next) + span(jobs[0].next_stride) : 0;
    if (ok && range <= 2 * (size_t)nj * span(jobs[0].next_stride)) { arena_bytes = range; o_arena = io.take(range + 16); }   // (gaps: streams that sit this call out)
  }
  // k_lk_pyramid reads its patches with a halo (level 0 about 2.5 times): images that lie in HOST memory (pinned, not in one arena) come
  // over by one DMA copy each instead of being read through PCIe by the kernel
  bool stage_each = false;
  size_t o_each = 0, each_span = 0;
  if (images_on_device && use_fused && !arena_bytes) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, jobs[0].next) == hipSuccess) stage_each = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
    if (stage_each) {
      for (int j = 0; j < nj; ++j) each_span = std::max(each_span, std::max(span(jobs[j].next_stride), jobs[j].prev ? span(jobs[j].prev_stride) : (size_t)0));
      each_span = (each_span + 63) & ~size_t(63);
      o_each = io.take(2 * (size_t)nj * each_span);
    }
  }
  SSX_HIP_TRY(ctx, ws->io.reserve(io.off, 1.5));
  SSX_HIP_TRY(ctx, ws->stage.reserve(host_end, 1.5));
  char* hs = ws->stage.as<char>();
  char* db = ws->io.as<char>();
  LkJob* tab = reinterpret_cast<LkJob*>(hs + o_tab);
  size_t pt0 = 0;
  for (int j = 0; j < nj; ++j) {
    const ssx_lk_job& q = jobs[j];
    LkJob& t = tab[j];
    t.pyr[0] = slot[j]->pyr[0]; t.pyr[1] = slot[j]->pyr[1]; t.deriv = slot[j]->deriv; t.deriv1 = slot[j]->deriv1;
    if (images_on_device) {
      t.img[0] = q.prev; t.img[1] = q.next; t.stride[0] = q.prev_stride; t.stride[1] = q.next_stride;
      if (arena_bytes) t.img[1] = (const uint8_t*)(db + o_arena) + (size_t)(q.next - jobs[0].next);
      if (stage_each) {
        t.img[1] = (const uint8_t*)(db + o_each + (size_t)(2 * j) * each_span);
        if (q.prev) t.img[0] = (const uint8_t*)(db + o_each + (size_t)(2 * j + 1) * each_span);
      }
    } else {
      for (int y = 0; y < rows; ++y) {
        if (q.prev) memcpy(hs + o_img0[j] + (size_t)y * cols, q.prev + (size_t)y * q.prev_stride, cols);
        memcpy(hs + o_img1[j] + (size_t)y * cols, q.next + (size_t)y * q.next_stride, cols);
      }
      t.img[0] = q.prev ? (const uint8_t*)(db + o_img0[j]) : nullptr; t.img[1] = (const uint8_t*)(db + o_img1[j]);
      t.stride[0] = t.stride[1] = cols;
    }
    t.n = q.n; t.pt0 = (int)pt0;
    if (q.n > 0) {
      memcpy(hs + o_pp + sizeof(float) * 2 * pt0, q.prev_pts, sizeof(float) * 2 * q.n);
      memcpy(hs + o_np + sizeof(float) * 2 * pt0, q.next_pts, sizeof(float) * 2 * q.n);
    }
    pt0 += (size_t)q.n;
  }
  SSX_HIP_TRY(ctx, hipMemcpyAsync(db, hs, in_bytes, hipMemcpyHostToDevice, s));
  const LkJob* dtab = reinterpret_cast<const LkJob*>(db + o_tab);
  if (arena_bytes) SSX_HIP_TRY(ctx, hipMemcpyAsync(db + o_arena, jobs[0].next, arena_bytes, hipMemcpyDefault, s));
  if (stage_each)
    for (int j = 0; j < nj; ++j) {
      SSX_HIP_TRY(ctx, hipMemcpyAsync(db + o_each + (size_t)(2 * j) * each_span, jobs[j].next, span(jobs[j].next_stride), hipMemcpyHostToDevice, s));
      if (jobs[j].prev) SSX_HIP_TRY(ctx, hipMemcpyAsync(db + o_each + (size_t)(2 * j + 1) * each_span, jobs[j].prev, span(jobs[j].prev_stride), hipMemcpyHostToDevice, s));
    }
  if (use_fused) {
    const int top = d.levels - 1;
    const dim3 g((d.cols[top] + 2 * d.pad + FT - 1) / FT, (d.rows[top] + 2 * d.pad + FT - 1) / FT, 2 * nj);   // z = job x {previous, next} image
    hipLaunchKernelGGL(k_lk_pyramid, g, dim3(256), 0, s, d, dtab);
  }
  for (int which = use_fused ? 2 : any_fresh ? 0 : 1; which < 2; ++which) {
    const dim3 g0((d.cols[0] + 2 * d.pad + 255) / 256, d.rows[0] + 2 * d.pad, nj);
    hipLaunchKernelGGL(k_lk_pad_level0, g0, dim3(256), 0, s, d, dtab, which);
    for (int l = 1; l < d.levels; ++l) {
      const dim3 g((d.cols[l] + 2 * d.pad + 255) / 256, d.rows[l] + 2 * d.pad, nj);
      hipLaunchKernelGGL(k_lk_pyr_down, g, dim3(256), 0, s, d, dtab, which, l);
    }
  }
  for (int l = scharr_now ? 0 : d.levels; l < d.levels; ++l) {
    const dim3 g((d.cols[l] + 2 * d.pad + 255) / 256, d.rows[l] + 2 * d.pad, nj);
    hipLaunchKernelGGL(k_lk_scharr, g, dim3(256), 0, s, d, dtab, l);
  }
  if (n_pts > 0) {
    d.max_iters = std::min(std::max(prm.max_iters, 0), 100);          // TermCriteria clamps of calcOpticalFlowPyrLK
    const double e = std::min(std::max(prm.eps, 0.), 10.);
    d.eps2 = e * e;
    d.min_eig = prm.min_eig_threshold;
    d.use_initial_flow = prm.use_initial_flow;
    d.prev_pts = (const float*)(db + o_pp); d.next_pts = (float*)(db + o_np);
    d.status = (uint8_t*)(db + o_st); d.err = (float*)(db + o_er);
    hipLaunchKernelGGL(k_lk_track, dim3((max_n + 3) / 4, nj), dim3(256), 0, s, d, dtab);
    SSX_HIP_TRY(ctx, hipGetLastError());
    SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_np, db + o_np, host_end - o_np, hipMemcpyDeviceToHost, s));
  }
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipStreamSynchronize(s));
  pt0 = 0;
  for (int j = 0; j < nj; ++j) {
    const ssx_lk_job& q = jobs[j];
    slot[j]->have_next = true;
    slot[j]->have_next_deriv = use_fused;
    if (q.n > 0) {
      memcpy(q.next_pts, hs + o_np + sizeof(float) * 2 * pt0, sizeof(float) * 2 * q.n);
      memcpy(q.status, hs + o_st + pt0, (size_t)q.n);
      if (q.err) memcpy(q.err, hs + o_er + sizeof(float) * pt0, sizeof(float) * q.n);
    }
    pt0 += (size_t)q.n;
  }
  if (top_level) *top_level = d.levels - 1;
  return SSX_OK;
}

}  // namespace

extern "C" {

void ssx_lk_default_params(ssx_lk_params* p)
{
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->win = 11; p->max_level = 3; p->max_iters = 30; p->eps = 0.01; p->min_eig_threshold = 1e-4f; p->use_initial_flow = 1;
}

ssx_status ssx_lk_track(ssx_ctx* ctx, const uint8_t* prev, int32_t prev_stride, const uint8_t* next, int32_t next_stride,
                        int32_t rows, int32_t cols, int32_t n, const float* prev_pts, float* next_pts, uint8_t* status,
                        float* err, const ssx_lk_params* prm_in, int32_t* top_level)
{
  if (!ctx || !prev || !next || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status))) return SSX_ERR_INVALID_ARG;
  const ssx_lk_job q{0, prev, prev_stride, next, next_stride, n, prev_pts, next_pts, status, err};
  return lk_run(ctx, 1, &q, rows, cols, prm_in, false, top_level);
}

ssx_status ssx_lk_track_next(ssx_ctx* ctx, const uint8_t* next, int32_t next_stride, int32_t rows, int32_t cols, int32_t n,
                             const float* prev_pts, float* next_pts, uint8_t* status, float* err,
                             const ssx_lk_params* prm_in, int32_t* top_level)
{
  if (!ctx || !next || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status))) return SSX_ERR_INVALID_ARG;
  const ssx_lk_job q{0, nullptr, 0, next, next_stride, n, prev_pts, next_pts, status, err};
  return lk_run(ctx, 1, &q, rows, cols, prm_in, false, top_level);
}

// n_jobs tracking problems in one call (one frame of each of n_jobs streams: FrontEnd::TrackLastFrame, frontend.cpp:130-182): every
// kernel of the tracker once for all of them.  Per job the bits of ssx_lk_track / ssx_lk_track_next.
ssx_status ssx_lk_track_batch(ssx_ctx* ctx, int32_t n_jobs, const ssx_lk_job* jobs, int32_t rows, int32_t cols, const ssx_lk_params* prm_in,
                              int32_t images_on_device)
{
  if (!ctx || n_jobs < 0 || (n_jobs > 0 && !jobs)) return SSX_ERR_INVALID_ARG;
  if (n_jobs == 0) return SSX_OK;
  return lk_run(ctx, n_jobs, jobs, rows, cols, prm_in, images_on_device != 0, nullptr);
}

#ifndef SSX_NO_TEST_HOOKS   // kernel taps of the parity tests (include/ssx_test_hooks.h)
// test / debug access to the pyramid and derivative images of the last ssx_lk_track call
ssx_status ssx_lk_stage_level(ssx_ctx* ctx, int32_t which, int32_t level, uint8_t* out, int32_t out_cap, int32_t* rows, int32_t* cols)
{
  if (!ctx || !ctx->lk || !rows || !cols) return SSX_ERR_INVALID_ARG;
  LkWorkspace* ws = static_cast<LkWorkspace*>(ctx->lk);
  if (!ws->planned || which < 0 || which > 1 || level < 0 || level >= ws->dev.levels) return SSX_ERR_INVALID_ARG;
  const LkDev& d = ws->dev;
  *rows = d.rows[level]; *cols = d.cols[level];
  if (!out) return SSX_OK;
  if (out_cap < d.rows[level] * d.cols[level]) return SSX_ERR_CAPACITY;
  if (ws->slots.empty() || !ws->slots[0].pyr[0]) return SSX_ERR_INVALID_ARG;
  const uint8_t* src = ws->slots[0].pyr[which] + d.off[level] + (size_t)d.pad * d.pitch[level] + d.pad;
  SSX_HIP_TRY(ctx, hipMemcpy2DAsync(out, d.cols[level], src, d.pitch[level], d.cols[level], d.rows[level], hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

ssx_status ssx_lk_stage_deriv(ssx_ctx* ctx, int32_t level, int16_t* out, int32_t out_cap, int32_t* rows, int32_t* cols)
{
  if (!ctx || !ctx->lk || !rows || !cols) return SSX_ERR_INVALID_ARG;
  LkWorkspace* ws = static_cast<LkWorkspace*>(ctx->lk);
  if (!ws->planned || level < 0 || level >= ws->dev.levels) return SSX_ERR_INVALID_ARG;
  const LkDev& d = ws->dev;
  *rows = d.rows[level]; *cols = d.cols[level];
  if (!out) return SSX_OK;
  if (out_cap < 2 * d.rows[level] * d.cols[level]) return SSX_ERR_CAPACITY;
  if (ws->slots.empty() || !ws->slots[0].deriv) return SSX_ERR_INVALID_ARG;
  const uint32_t* src = ws->slots[0].deriv + d.doff[level] + (size_t)d.pad * d.pitch[level] + d.pad;
  SSX_HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)d.cols[level] * 4, src, (size_t)d.pitch[level] * 4, (size_t)d.cols[level] * 4, d.rows[level],
                                    hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

#endif  // SSX_NO_TEST_HOOKS

}  // extern "C"
