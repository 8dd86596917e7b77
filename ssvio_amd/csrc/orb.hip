// ssvio_amd/csrc/orb.hip -- ORB extraction on gfx950: pyramid, grid FAST-9/16 + NMS, octree selection,
// intensity-centroid orientation, 7x7 Gaussian blur, steered BRIEF-256.
//
// Replaces the arithmetic of ssvio::ORBextractor (/root/reference/src/ssvio/orbextractor.cpp) and of the OpenCV 3.2
// calls it makes (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound).  Integer / f32 results are
// bit-identical to the CPU oracle (oracle/src/orb_oracle.cpp): this file is compiled with -ffp-contract=off and
// every float expression keeps the operation order of the scalar code.
//
// Everything is batched over I images (a stereo pair is I = 2; ssx_stereo_batch_dev runs I = 2 x pairs):
// blockIdx.z (or the leading grid dimension) is the image, so one launch sequence serves the whole batch.
//
//   k_resize         level l from level l-1, one thread per destination pixel, fixed-point bilinear (A5)
//   k_fast_cells     ONE WORKGROUP PER GRID CELL: ROI (<= 72x72 bytes) staged in LDS from coalesced row reads,
//                    segment test + cornerScore at iniThFAST, fallback to minThFAST when the cell is empty,
//                    3x3 NMS inside the cell, mask test, ordered (row-major) compaction                     (A1+A2)
//   k_octree         ONE WORKGROUP PER (image, level): data-parallel DistributeOctTree (A3); the formulation is
//                    tools/octree_model.py -- stable 4-way partitions by packed prefix sums, list order by scans
//   k_orient         one wave per keypoint: 31x31 patch in LDS, integer moments, fastAtan2 polynomial        (A6)
//   k_gauss7         separable 7x7 sigma=2 in Q8 fixed point, 64x16 tiles with 3-px halo in LDS             (A7)
//   k_brief          one wave per keypoint: 37x37 blurred patch in LDS, 4 tests per lane, ballot-packed      (A7)
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "ctx.hpp"
#include "orb_ws.hpp"

namespace ssxorb {

__constant__ int8_t c_pattern[256 * 4] = {
#include "brief_pattern.inc"
};
// umax of the circular patch (orbextractor.cpp:176-191): closed form checked in tests
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// getGaussianKernel(7, 2) * 256 rounded (OpenCV 8U fixed-point path): sums to 257, checked in tests
__constant__ int c_gauss[7] = {18, 34, 49, 55, 49, 34, 18};
__constant__ int c_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int c_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// ------------------------------------------------------------------------------------------------
// A5: cv::resize INTER_LINEAR 8UC1 (11-bit coefficients, vertical ((b*(S>>4))>>16 ... +2)>>2)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base,
                                                size_t img_stride_bytes, int srows, int scols, int spitch,
                                                int drows, int dcols, int dpitch, double scale_x, double scale_y)
{
  // scale_x = 1. / ((double)dcols / scols) exactly as cv::resize forms it (computed once on the host)
  const int dx = blockIdx.x * 256 + threadIdx.x;
  const int dy = blockIdx.y;
  const uint8_t* src = src_base + (size_t)blockIdx.z * img_stride_bytes;
  uint8_t* dst = dst_base + (size_t)blockIdx.z * img_stride_bytes;
  if (dx >= dcols) return;
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = cv_floor_f(fx);
  fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  const bool tail = (sx + 1 >= scols);   // dx >= xmax
  if (sx >= scols - 1) { fx = 0; sx = scols - 1; }
  const int a0 = __float2int_rn((1.f - fx) * 2048), a1 = __float2int_rn(fx * 2048);
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = cv_floor_f(fy);
  fy -= sy;
  const int b0 = __float2int_rn((1.f - fy) * 2048), b1 = __float2int_rn(fy * 2048);
  const int sy0 = sy < 0 ? 0 : (sy < srows ? sy : srows - 1);
  const int sy1 = (sy + 1) < 0 ? 0 : ((sy + 1) < srows ? (sy + 1) : srows - 1);
  const uint8_t* r0 = src + (size_t)sy0 * spitch;
  const uint8_t* r1 = src + (size_t)sy1 * spitch;
  int S0, S1;
  if (!tail) {
    S0 = r0[sx] * a0 + r0[sx + 1] * a1;
    S1 = r1[sx] * a0 + r1[sx + 1] * a1;
  } else {
    S0 = r0[sx] * 2048;
    S1 = r1[sx] * 2048;
  }
  dst[(size_t)dy * dpitch + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

// copy a pitched host-layout image into level 0 of the pyramid (pitch change) and optionally fill a constant
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ in, int in_stride, size_t in_img_bytes,
                                                     uint8_t* __restrict__ dst_base, size_t img_stride_bytes,
                                                     int rows, int cols, int dpitch, int fill, int fill_value)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= cols) return;
  uint8_t* dst = dst_base + (size_t)blockIdx.z * img_stride_bytes;
  dst[(size_t)y * dpitch + x] = fill ? (uint8_t)fill_value : in[(size_t)blockIdx.z * in_img_bytes + (size_t)y * in_stride + x];
}

// ------------------------------------------------------------------------------------------------
// A1+A2: grid FAST.  One workgroup per cell.
// ------------------------------------------------------------------------------------------------
constexpr int ROI_MAX = 72;   // largest supported cell ROI side

__device__ __forceinline__ bool arc9(unsigned m)   // 9 contiguous set bits in a circular 16-bit mask
{
  m |= m << 16;
  unsigned r = m & (m >> 1);
  r &= r >> 2;       // 4 contiguous
  r &= r >> 4;       // 8 contiguous
  r &= m >> 8;       // 9 contiguous
  return (r & 0xFFFFu) != 0;
}

// segment test + cornerScore<16> (OpenCV) for the pixel at LDS address c (row pitch p): returns -1 when the
// pixel is not a FAST-9 corner at threshold t, else the score (largest threshold that still passes, minus 1... i.e.
// -b0-1 of OpenCV's cornerScore) which is >= t-1 and <= 254.
__device__ __forceinline__ int fast_score(const uint8_t* c, int p, int t)
{
  const int v = c[0];
  int d[16];
  unsigned dark = 0, bright = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = c[c_ring_dx[k] + c_ring_dy[k] * p];
    d[k] = v - r;
    dark |= (unsigned)(r < v - t) << k;
    bright |= (unsigned)(r > v + t) << k;
  }
  if (!arc9(dark) && !arc9(bright)) return -1;
  // sliding-window (9) min / max over the circular ring by doubling: 2, 4, 8, then +1
  int mn[16], mx[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { mn[k] = min(d[k], d[(k + 1) & 15]); mx[k] = max(d[k], d[(k + 1) & 15]); }
  int mn4[16], mx4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { mn4[k] = min(mn[k], mn[(k + 2) & 15]); mx4[k] = max(mx[k], mx[(k + 2) & 15]); }
  int a0 = t, bmin = 255;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int m8 = min(mn4[k], mn4[(k + 4) & 15]);
    const int x8 = max(mx4[k], mx4[(k + 4) & 15]);
    a0 = max(a0, min(m8, d[(k + 8) & 15]));
    bmin = min(bmin, max(x8, d[(k + 8) & 15]));
  }
  const int b0 = min(-a0, bmin);
  return -b0 - 1;
}

// ring pixel k of the Bresenham circle relative to c (LDS row pitch p)
#define RING(k) ((int)c[c_ring_dx[k] + c_ring_dy[k] * p])

// cheap necessary condition: an arc of 9 contiguous ring pixels contains at least 2 of the 4 compass pixels
// (ring indices 0, 4, 8, 12), so a corner needs >= 2 of them darker than v-t or >= 2 brighter than v+t.
__device__ __forceinline__ bool fast_quick(const uint8_t* c, int p, int t)
{
  const int v = c[0];
  const int lo = v - t, hi = v + t;
  const int p0 = c[3 * p], p8 = c[-3 * p], p4 = c[3], p12 = c[-3];
  const int nd = (p0 < lo) + (p8 < lo) + (p4 < lo) + (p12 < lo);
  const int nb = (p0 > hi) + (p8 > hi) + (p4 > hi) + (p12 > hi);
  return nd >= 2 || nb >= 2;
}

// ONE WAVE PER CELL (4 cells per 256-thread workgroup, no workgroup barrier anywhere):
//   1. the ROI is staged in LDS with aligned dword loads (level pitch is a multiple of 128 bytes);
//   2. every interior pixel takes the 4-load quick test; survivors are compacted with ballot/popcount;
//   3. the compacted survivors take the full segment test + cornerScore (dense lanes);
//   4. the corners (again a compacted list) take the 3x3 strict NMS, the mask test and an ORDER-PRESERVING
//      compaction (ballot prefix inside a round; rounds walk the pixels in row-major order).
__global__ __launch_bounds__(256) void k_fast_cells(OrbDev o)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cell = blockIdx.x * 4 + wave, img = blockIdx.y;
  if (cell >= o.n_cells) return;
  const Cell c = o.cells[cell];
  const uint8_t* lvl = o.pyr + (size_t)img * o.pyr_bytes + o.lvl_off[c.level];
  const int pitch = o.lvl_pitch[c.level];
  const int w = c.w, h = c.h;
  const int tx0 = c.x0 & ~3;                       // aligned tile origin
  const int xoff = c.x0 - tx0;
  const int tw = (xoff + w + 3) & ~3;              // LDS row pitch (bytes), multiple of 4
  const int ndw = tw >> 2;
  uint8_t* sImg = smem + (size_t)wave * o.fast_lds_per_wave;
  uint8_t* sScore = sImg + o.fast_tile_bytes;
  uint16_t* sList = reinterpret_cast<uint16_t*>(sScore + o.fast_tile_bytes);   // compacted pixel indices
  for (int i = lane; i < h * ndw; i += 64) {
    const int y = i / ndw, xd = i - y * ndw;
    reinterpret_cast<uint32_t*>(sImg)[i] = *reinterpret_cast<const uint32_t*>(lvl + (size_t)(c.y0 + y) * pitch + tx0 + 4 * xd);
  }
  const int iw = w - 6, ih = h - 6;                // cv::FAST ignores a 3-px border of the ROI
  const int npx = (iw > 0 && ih > 0) ? iw * ih : 0;
  const float inv_iw = iw > 0 ? 1.0f / (float)iw : 0.f;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int* cell_count = o.cell_count + (size_t)img * o.n_cells + cell;
  uint32_t* cell_cand = o.cell_cand + ((size_t)img * o.n_cells + cell) * CELL_CAP;
  const uint8_t* mk = o.has_mask ? o.maskpyr + (size_t)img * o.pyr_bytes + o.lvl_off[c.level] : nullptr;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int n_out = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int th = min(max(pass == 0 ? o.ini_th : o.min_th, 0), 255);
    for (int i = lane; i < h * ndw; i += 64) reinterpret_cast<uint32_t*>(sScore)[i] = 0;
    // -- stage 1: quick test on every interior pixel, compaction of the survivors --
    int n_surv = 0;
    for (int base = 0; base < npx; base += 64) {
      const int pxi = base + lane;
      bool pass_q = false;
      int off = 0;
      if (pxi < npx) {
        const int yy = (int)(((float)pxi + 0.5f) * inv_iw);
        off = (3 + yy) * tw + xoff + 3 + (pxi - yy * iw);
        pass_q = fast_quick(&sImg[off], tw, th);
      }
      const unsigned long long bal = __ballot(pass_q);
      if (pass_q) sList[n_surv + __popcll(bal & lt_mask)] = (uint16_t)off;
      n_surv += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // -- stage 2: full segment test + score on the survivors; corners compacted in place (order kept) --
    int n_corner = 0;
    for (int base = 0; base < n_surv; base += 64) {
      const int k = base + lane;
      int off = 0, sc = -1;
      if (k < n_surv) {
        off = sList[k];
        sc = fast_score(&sImg[off], tw, th);
      }
      // OpenCV keeps the score as uchar; a corner with score 0 can never win the strict '>' NMS
      const bool is_c = sc > 0;
      if (is_c) sScore[off] = (uint8_t)sc;
      const unsigned long long bal = __ballot(is_c);
      __builtin_amdgcn_wave_barrier();   // all lanes have read sList[base..base+63] before it is overwritten
      if (is_c) sList[n_corner + __popcll(bal & lt_mask)] = (uint16_t)off;   // n_corner + rank <= k: in-place safe
      n_corner += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // -- stage 3: NMS (strict >), emptiness test BEFORE the mask (orbextractor.cpp:803-808), mask test at the
    //    UN-bordered coordinates (orbextractor.cpp:816-823, reference quirk), ordered output --
    int n_keep_pre = 0;
    n_out = 0;
    for (int base = 0; base < n_corner; base += 64) {
      const int k = base + lane;
      bool keep = false;
      int off = 0;
      if (k < n_corner) {
        off = sList[k];
        const uint8_t* r = &sScore[off];
        const int sc = r[0];
        keep = sc > r[1] && sc > r[-1] && sc > r[-tw - 1] && sc > r[-tw] && sc > r[-tw + 1] && sc > r[tw - 1] &&
               sc > r[tw] && sc > r[tw + 1];
      }
      n_keep_pre += __popcll(__ballot(keep));
      const int yl = off / tw, xl = off - yl * tw - xoff;           // ROI-local coordinates
      if (keep && mk) keep = mk[(size_t)(yl + c.oy) * pitch + (xl + c.ox)] != 0;
      const unsigned long long bal = __ballot(keep);
      const int pos = n_out + __popcll(bal & lt_mask);
      if (keep && pos < CELL_CAP)
        cell_cand[pos] = (uint32_t)(xl + c.ox) | ((uint32_t)(yl + c.oy) << 12) | ((uint32_t)sScore[off] << 24);
      n_out += __popcll(bal);
    }
    if (n_keep_pre > 0) break;      // keypoints found at this threshold: no retry (wave-uniform)
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    *cell_count = min(n_out, CELL_CAP);
    if (n_out > CELL_CAP) atomicOr(&o.status[img], 1);
  }
}
#undef RING

// ------------------------------------------------------------------------------------------------
// A7: cv::GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, 8-bit fixed point (row pass exact ints, column pass
// (sum + 2^15) >> 16).  One 256-thread workgroup computes a 64x16 output tile from a 70x22 LDS tile.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
  return i;
}

constexpr int GT_W = 128, GT_H = 32;            // output tile
constexpr int GT_PITCH = GT_W + 8;              // input tile pitch: 4 bytes of halo left, 4 right (3 needed)

// ONE launch for all levels and images: blockIdx.x walks the tiles of every level (o.gauss_tile0[]), blockIdx.y
// is the image.  Input rows are fetched as aligned dwords (interior tiles) into LDS; the row pass produces four
// Q8 sums per thread from three LDS dwords; the column pass reads ten int4 rows and stores one dword per row.
__global__ __launch_bounds__(256) void k_gauss7(OrbDev o)
{
  __shared__ __attribute__((aligned(16))) uint8_t sIn[(GT_H + 6) * GT_PITCH];
  __shared__ __attribute__((aligned(16))) int sRow[(GT_H + 6) * GT_W];
  const int img = blockIdx.y, t = threadIdx.x;
  int level = 0;
  while (level + 1 < o.nlevels && (int)blockIdx.x >= o.gauss_tile0[level + 1]) ++level;
  const int tile = blockIdx.x - o.gauss_tile0[level];
  const int rows = o.lvl_rows[level], cols = o.lvl_cols[level], pitch = o.lvl_pitch[level];
  const int tiles_x = (cols + GT_W - 1) / GT_W;
  const int ty_ = tile / tiles_x, tx_ = tile - ty_ * tiles_x;
  const int x0 = tx_ * GT_W, y0 = ty_ * GT_H;
  const uint8_t* src = o.pyr + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  uint8_t* dst = o.blur + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  constexpr int NDW = GT_PITCH / 4;   // 34 dwords per tile row
  for (int i = t; i < (GT_H + 6) * NDW; i += 256) {
    const int r = i / NDW, dwi = i - r * NDW;
    const int gy = reflect101(y0 + r - 3, rows);
    const int gx = x0 - 4 + 4 * dwi;
    uint32_t v;
    if (gx >= 0 && gx + 3 < cols) {
      v = *reinterpret_cast<const uint32_t*>(src + (size_t)gy * pitch + gx);
    } else {
      const uint8_t* row = src + (size_t)gy * pitch;
      v = (uint32_t)row[reflect101(gx, cols)] | ((uint32_t)row[reflect101(gx + 1, cols)] << 8) |
          ((uint32_t)row[reflect101(gx + 2, cols)] << 16) | ((uint32_t)row[reflect101(gx + 3, cols)] << 24);
    }
    reinterpret_cast<uint32_t*>(sIn)[i] = v;
  }
  __syncthreads();
  // row pass: thread -> (tile row r, group g of 4 output pixels); bytes 4g+1 .. 4g+10 of the tile row
  for (int i = t; i < (GT_H + 6) * (GT_W / 4); i += 256) {
    const int r = i / (GT_W / 4), g = i - r * (GT_W / 4);
    const uint32_t* rowdw = reinterpret_cast<const uint32_t*>(sIn + r * GT_PITCH) + g;
    const uint32_t d0 = rowdw[0], d1 = rowdw[1], d2 = rowdw[2];
    int b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { b[k] = (d0 >> (8 * k)) & 0xFF; b[4 + k] = (d1 >> (8 * k)) & 0xFF; b[8 + k] = (d2 >> (8 * k)) & 0xFF; }
    int acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = 0;
#pragma unroll
      for (int q = 0; q < 7; ++q) acc[j] += c_gauss[q] * b[1 + j + q];
    }
    *reinterpret_cast<int4*>(&sRow[r * GT_W + 4 * g]) = make_int4(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  // column pass: thread -> 4 pixels x 4 rows
  {
    const int g = t & 31, rq = t >> 5;             // 32 groups x 8 row-quads
    int4 rowv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) rowv[k] = *reinterpret_cast<const int4*>(&sRow[(4 * rq + k) * GT_W + 4 * g]);
    const int x = x0 + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + 4 * rq + j;
      int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        a0 += c_gauss[q] * rowv[j + q].x; a1 += c_gauss[q] * rowv[j + q].y;
        a2 += c_gauss[q] * rowv[j + q].z; a3 += c_gauss[q] * rowv[j + q].w;
      }
      // sums are non-negative, so only the upper clamp is needed.  Written with UNSIGNED shifts/min on purpose:
      // hipcc (ROCm 7.2) fuses med3(ashr(x,16),0,255) pairs into v_ashr_pk_u8_i32 and then assumes the upper
      // 16 bits of its result are zero, which they are not on gfx950 (byte 0 leaked into byte 2).
      const uint32_t o0 = min(((uint32_t)a0 + (1u << 15)) >> 16, 255u);
      const uint32_t o1 = min(((uint32_t)a1 + (1u << 15)) >> 16, 255u);
      const uint32_t o2 = min(((uint32_t)a2 + (1u << 15)) >> 16, 255u);
      const uint32_t o3 = min(((uint32_t)a3 + (1u << 15)) >> 16, 255u);
      // the pitch is a multiple of the tile width: the dword store stays inside the row's padding
      if (y < rows && x < pitch) *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// A6: IC_Angle (orbextractor.cpp:15-43) + cv::fastAtan2.  One wave per keypoint, 4 keypoints per workgroup.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
  const float s = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
  const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__global__ __launch_bounds__(256) void k_orient(OrbDev o)
{
  __shared__ uint8_t sPatch[4][31 * 32];
  const int level = blockIdx.y, img = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int il = img * o.nlevels + level;
  const int n = o.sel_count[il];
  const int k = blockIdx.x * 4 + wave;
  const bool active = k < n;
  const uint32_t p = active ? o.sel[(size_t)il * SEL_CAP + k] : 0u;
  const int minB = EDGE_THRESHOLD - 3;
  const int cx = (int)(p & 0xFFF) + minB, cy = (int)((p >> 12) & 0xFFF) + minB;   // cvRound of integral coords
  const uint8_t* src = o.pyr + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  const int pitch = o.lvl_pitch[level];
  uint8_t* sp = sPatch[wave];
  // stage the 31x31 patch: each iteration the wave reads two 31-byte row segments
  if (active)
    for (int i = lane; i < 31 * 32; i += 64) {
      const int r = i >> 5, cc = i & 31;
      if (cc < 31) sp[i] = src[(size_t)(cy - 15 + r) * pitch + (cx - 15 + cc)];
    }
  __syncthreads();
  if (!active) return;
  int m10 = 0, m01 = 0;
  if (lane < 31) {
    const int v = lane - 15;
    const int dmax = c_umax[v < 0 ? -v : v];
    int sum = 0;
    for (int u = -dmax; u <= dmax; ++u) {
      const int val = sp[lane * 32 + (u + 15)];
      m10 += u * val;
      sum += val;
    }
    m01 = v * sum;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { m10 += __shfl_xor(m10, off); m01 += __shfl_xor(m01, off); }
  if (lane == 0) o.sel_angle[(size_t)il * SEL_CAP + k] = fast_atan2_deg((float)m01, (float)m10);
}

// ------------------------------------------------------------------------------------------------
// A7: computeOrbDescriptor (orbextractor.cpp:46-91) + keypoint finalisation (:660-670, :741-749).
// One wave per keypoint; 37x37 blurred patch in LDS; lane l evaluates tests l, 64+l, 128+l, 192+l and the four
// ballots ARE the descriptor's four 64-bit words (bit i of the descriptor = test i, LSB first).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_deg(float angle_deg, float* c, float* s)
{
  // fixed sequence of IEEE double operations (no libm, no FMA): bit-identical on CPU and GPU; see oracle.
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float ang = angle_deg * factorPI;
  const double r = (double)ang;
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
  const double kq = floor(r * two_over_pi + 0.5);
  const int q = ((int)kq) & 3;
  const double y = (r - kq * pio2_hi) - kq * pio2_lo;
  const double z = y * y;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  const double sn = y + (y * z) * ps;
  const double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
  double so, co;
  switch (q) {
    case 0: so = sn; co = cs; break;
    case 1: so = cs; co = -sn; break;
    case 2: so = -sn; co = -cs; break;
    default: so = -cs; co = sn; break;
  }
  *c = (float)co;
  *s = (float)so;
}

constexpr int BP = 37, BR = 18;   // blurred patch side / radius: |rotated pattern point| <= 18.4 -> rounds to <= 18

__global__ __launch_bounds__(256) void k_brief(OrbDev o)
{
  __shared__ uint8_t sPatch[4][BP * 40];
  const int level = blockIdx.y, img = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int il = img * o.nlevels + level;
  const int n = o.sel_count[il];
  const int k = blockIdx.x * 4 + wave;
  // output slot: levels are concatenated in order (orbextractor.cpp:722-752)
  int base = 0;
  for (int l = 0; l < level; ++l) base += o.sel_count[img * o.nlevels + l];
  const int slot = base + k;
  const bool active = (k < n) && (slot < o.out_cap);
  if (k < n && slot >= o.out_cap && lane == 0) atomicOr(&o.status[img], 4);
  const uint32_t p = active ? o.sel[(size_t)il * SEL_CAP + k] : 0u;
  const int minB = EDGE_THRESHOLD - 3;
  const int cx = (int)(p & 0xFFF) + minB, cy = (int)((p >> 12) & 0xFFF) + minB;
  const float angle = active ? o.sel_angle[(size_t)il * SEL_CAP + k] : 0.f;
  const uint8_t* src = o.blur + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  const int pitch = o.lvl_pitch[level];
  uint8_t* sp = sPatch[wave];
  if (active)
    for (int i = lane; i < BP * 40; i += 64) {
      const int r = i / 40, cc = i - r * 40;
      if (cc < BP) sp[i] = src[(size_t)(cy - BR + r) * pitch + (cx - BR + cc)];
    }
  __syncthreads();
  if (!active) return;
  float a, b;
  sincos_deg(angle, &a, &b);
  unsigned long long words[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int8_t* tp = &c_pattern[(j * 64 + lane) * 4];
    const float x0 = (float)tp[0], y0 = (float)tp[1], x1 = (float)tp[2], y1 = (float)tp[3];
    const int r0 = __float2int_rn(x0 * b + y0 * a), c0 = __float2int_rn(x0 * a - y0 * b);
    const int r1 = __float2int_rn(x1 * b + y1 * a), c1 = __float2int_rn(x1 * a - y1 * b);
    const int t0 = sp[(r0 + BR) * 40 + (c0 + BR)];
    const int t1 = sp[(r1 + BR) * 40 + (c1 + BR)];
    words[j] = __ballot(t0 < t1);
  }
  if (lane < 4) {
    unsigned long long* dd = reinterpret_cast<unsigned long long*>(o.out_desc + ((size_t)img * o.out_cap + slot) * 32);
    dd[lane] = words[lane];
  }
  if (lane == 0) {
    ssx_keypoint kp;
    const float sc = o.scale[level];
    kp.x = (float)cx; kp.y = (float)cy;
    if (level != 0) { kp.x *= sc; kp.y *= sc; }
    kp.size = (float)(int)(31 * sc);
    kp.angle = angle;
    kp.response = (float)(p >> 24);
    kp.octave = level;
    kp.class_id = -1;
    reinterpret_cast<ssx_keypoint*>(o.out_kps)[(size_t)img * o.out_cap + slot] = kp;
  }
}

// ------------------------------------------------------------------------------------------------
// A8: ORBextractor::ScreenAndComputeKPsParams (orbextractor.cpp:844-894) + CalcDescriptors (:943-991), the
// loop-closing variant: keypoints GIVEN (position at level 0, octave), keep those that are >= 19 px inside their
// level and pass the FAST segment test at minThFAST, then orientation + size + descriptor.  One wave per keypoint.
// ------------------------------------------------------------------------------------------------
struct DescribeAt {
  const ssx_keypoint* in;   // n_in
  ssx_keypoint* out;        // n_in (valid where keep)
  uint8_t* desc;            // n_in x 32
  uint8_t* keep;            // n_in
  int n_in;
};

__global__ __launch_bounds__(256) void k_describe_at(OrbDev o, DescribeAt a)
{
  __shared__ uint8_t sPatch[4][BP * 40];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + wave;
  bool active = k < a.n_in;
  ssx_keypoint kp{};
  int level = 0, cx = 0, cy = 0;
  float sc = 1.f;
  if (active) {
    kp = a.in[k];
    level = kp.octave;
    active = level >= 0 && level < o.nlevels;
  }
  if (active) {
    sc = o.scale[level];
    kp.x = kp.x / sc; kp.y = kp.y / sc;                       // kps.pt /= scale
    const int rows = o.lvl_rows[level], cols = o.lvl_cols[level];
    active = (kp.y - EDGE_THRESHOLD >= 0 && kp.y + EDGE_THRESHOLD < rows && kp.x - EDGE_THRESHOLD >= 0 &&
              kp.x + EDGE_THRESHOLD < cols);
    cx = __float2int_rn(kp.x); cy = __float2int_rn(kp.y);     // cvRound
  }
  const uint8_t* lvl = o.pyr + o.lvl_off[level];
  const int pitch = o.lvl_pitch[level];
  if (active) active = fast_score(lvl + (size_t)cy * pitch + cx, pitch, min(max(o.min_th, 0), 255)) >= 0;   // isFastCorner
  uint8_t* sp = sPatch[wave];
  // orientation patch (raw level)
  if (active)
    for (int i = lane; i < 31 * 32; i += 64) {
      const int r = i >> 5, cc = i & 31;
      if (cc < 31) sp[i] = lvl[(size_t)(cy - 15 + r) * pitch + (cx - 15 + cc)];
    }
  __syncthreads();
  float angle = 0.f;
  if (active) {
    int m10 = 0, m01 = 0;
    if (lane < 31) {
      const int v = lane - 15;
      const int dmax = c_umax[v < 0 ? -v : v];
      int sum = 0;
      for (int u = -dmax; u <= dmax; ++u) { const int val = sp[lane * 32 + (u + 15)]; m10 += u * val; sum += val; }
      m01 = v * sum;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m10 += __shfl_xor(m10, off); m01 += __shfl_xor(m01, off); }
    angle = fast_atan2_deg((float)m01, (float)m10);
  }
  __syncthreads();
  // CalcDescriptors: pt (already multiplied back by scale) is divided by scale AGAIN before describing
  float ox = kp.x * sc, oy = kp.y * sc;                        // kps.pt *= scale  (the output coordinates)
  const float dx = ox / sc, dy = oy / sc;
  const int bx = __float2int_rn(dx), by = __float2int_rn(dy);
  const uint8_t* blur = o.blur + o.lvl_off[level];
  if (active)
    for (int i = lane; i < BP * 40; i += 64) {
      const int r = i / 40, cc = i - r * 40;
      if (cc < BP) sp[i] = blur[(size_t)(by - BR + r) * pitch + (bx - BR + cc)];
    }
  __syncthreads();
  if (!active) { if (k < a.n_in && lane == 0) a.keep[k] = 0; return; }
  float ca, sb;
  sincos_deg(angle, &ca, &sb);
  unsigned long long words[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int8_t* tp = &c_pattern[(j * 64 + lane) * 4];
    const float x0 = (float)tp[0], y0 = (float)tp[1], x1 = (float)tp[2], y1 = (float)tp[3];
    const int r0 = __float2int_rn(x0 * sb + y0 * ca), c0 = __float2int_rn(x0 * ca - y0 * sb);
    const int r1 = __float2int_rn(x1 * sb + y1 * ca), c1 = __float2int_rn(x1 * ca - y1 * sb);
    words[j] = __ballot(sp[(r0 + BR) * 40 + (c0 + BR)] < sp[(r1 + BR) * 40 + (c1 + BR)]);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(a.desc + (size_t)k * 32)[lane] = words[lane];
  if (lane == 0) {
    ssx_keypoint okp = kp;
    okp.x = ox; okp.y = oy;
    okp.angle = angle;
    okp.size = 31 * sc;           // PATCH_SIZE * mvScaleFactor[level]  (float, :888)
    a.out[k] = okp;
    a.keep[k] = 1;
  }
}

// ORBextractor::Detect output: octree selection of level 0 + border, size 7, angle -1, octave 0 (cv::FAST keypoints)
__global__ __launch_bounds__(256) void k_finalize_detect(OrbDev o)
{
  const int img = blockIdx.y;
  const int il = img * o.nlevels;
  const int n = o.sel_count[il];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  if (k >= o.out_cap) { atomicOr(&o.status[img], 4); return; }
  const uint32_t p = o.sel[(size_t)il * SEL_CAP + k];
  const int minB = EDGE_THRESHOLD - 3;
  ssx_keypoint kp;
  kp.x = (float)(p & 0xFFF) + (float)minB;
  kp.y = (float)((p >> 12) & 0xFFF) + (float)minB;
  kp.size = 7.f; kp.angle = -1.f; kp.response = (float)(p >> 24); kp.octave = 0; kp.class_id = -1;
  reinterpret_cast<ssx_keypoint*>(o.out_kps)[(size_t)img * o.out_cap + k] = kp;
}

__global__ void k_counts(OrbDev o)
{
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= o.I) return;
  int n = 0;
  const int nl = o.detect_only ? 1 : o.nlevels;
  for (int l = 0; l < nl; ++l) n += o.sel_count[img * o.nlevels + l];
  o.out_n[img] = min(n, o.out_cap);
}


// ================================================================================================
// host side
// ================================================================================================
static void orb_ws_free(OrbWorkspace* w)
{
  if (!w) return;
  w->arena.release(); w->input.release(); w->stereo.release(); w->stage.release();
  delete w;
}

OrbWorkspace* get_ws(ssx_ctx* ctx)
{
  if (!ctx->orb) { ctx->orb = new OrbWorkspace(); ctx->orb_free = orb_ws_free; }
  return ctx->orb;
}

namespace {

inline int h_round(double v) { return (int)std::lrint(v); }

// grid of ORBextractor::Detect / ComputeKeyPointsOctTree (orbextractor.cpp:765-801, 575-612): float arithmetic
// kept exactly as the reference writes it.
void make_cells(int rows, int cols, int level, std::vector<Cell>& out)
{
  const float W = 30;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols < 1 || nRows < 1) return;
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
  for (int i = 0; i < nRows; ++i) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; ++j) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      Cell c;
      c.x0 = (int16_t)(int)iniX; c.y0 = (int16_t)(int)iniY;
      c.w = (int16_t)((int)maxX - (int)iniX); c.h = (int16_t)((int)maxY - (int)iniY);
      c.ox = (int16_t)(j * wCell); c.oy = (int16_t)(i * hCell);
      c.level = (int16_t)level; c.pad = 0;
      out.push_back(c);
    }
  }
}

}  // namespace

ssx_status plan(ssx_ctx* ctx, int rows, int cols, int I, const ssx_orb_params& prm, bool has_mask, bool detect_only)
{
  OrbWorkspace* ws = get_ws(ctx);
  const int nlevels = detect_only ? 1 : prm.nlevels;
  if (rows <= 2 * EDGE_THRESHOLD || cols <= 2 * EDGE_THRESHOLD || rows > 4000 || cols > 4000) {
    ctx->set_error("ssx_orb: image %dx%d outside the supported range (40..4000 per side)", cols, rows);
    return SSX_ERR_INVALID_ARG;
  }
  if (nlevels < 1 || nlevels > MAX_LEVELS || !(prm.scale_factor > 1.0f) || prm.nfeatures < 1 || prm.nfeatures > SEL_CAP - 8 || I < 1) {
    ctx->set_error("ssx_orb: unsupported parameters (nlevels=%d scale=%g nfeatures=%d)", prm.nlevels, (double)prm.scale_factor, prm.nfeatures);
    return SSX_ERR_INVALID_ARG;
  }
  if (ws->planned && ws->rows == rows && ws->cols == cols && ws->I == I && ws->nlevels == nlevels &&
      ws->nfeatures == prm.nfeatures && ws->ini_th == prm.ini_th_fast && ws->min_th == prm.min_th_fast &&
      ws->has_mask == (int)has_mask && ws->detect_only == (int)detect_only && ws->scale_factor == prm.scale_factor)
    return SSX_OK;
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // re-planning invalidates buffers in flight
  OrbDev d{};
  d.I = I; d.nlevels = nlevels; d.ini_th = prm.ini_th_fast; d.min_th = prm.min_th_fast;
  d.has_mask = has_mask; d.detect_only = detect_only;
  // ORBextractor ctor tables (orbextractor.cpp:133-168)
  float scale[MAX_LEVELS], inv[MAX_LEVELS];
  scale[0] = 1.0f;
  for (int i = 1; i < nlevels; ++i) scale[i] = scale[i - 1] * prm.scale_factor;
  for (int i = 0; i < nlevels; ++i) inv[i] = 1.0f / scale[i];
  if (detect_only) {
    d.feat[0] = prm.nfeatures;
  } else {
    const float factor = 1.0f / prm.scale_factor;
    float nDesired = prm.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
      d.feat[l] = h_round(nDesired);
      sum += d.feat[l];
      nDesired *= factor;
    }
    d.feat[nlevels - 1] = std::max(prm.nfeatures - sum, 0);
  }
  std::vector<Cell> cells;
  size_t off = 0;
  int out_cap = 32;
  for (int l = 0; l < nlevels; ++l) {
    d.scale[l] = scale[l];
    d.lvl_cols[l] = h_round((float)cols * inv[l]);   // ComputePyramid, orbextractor.cpp:999-1001
    d.lvl_rows[l] = h_round((float)rows * inv[l]);
    d.lvl_pitch[l] = (d.lvl_cols[l] + 127) & ~127;
    d.lvl_off[l] = off;
    off += (size_t)d.lvl_pitch[l] * d.lvl_rows[l];
    d.lvl_cell0[l] = (int)cells.size();
    make_cells(d.lvl_rows[l], d.lvl_cols[l], l, cells);
    out_cap += d.feat[l] + 4;
    if (d.feat[l] * 4 + 8 > NODE_CAP) {
      ctx->set_error("ssx_orb: %d features on level %d exceed the octree node capacity", d.feat[l], l);
      return SSX_ERR_UNSUPPORTED;
    }
  }
  d.lvl_cell0[nlevels] = (int)cells.size();
  for (int l = nlevels + 1; l <= MAX_LEVELS; ++l) d.lvl_cell0[l] = (int)cells.size();
  d.n_cells = (int)cells.size();
  d.pyr_bytes = (off + 255) & ~size_t(255);
  d.out_cap = out_cap;
  {
    int tile = 16, npx = 1, t0 = 0;
    for (const Cell& c : cells) {
      const int tw = ((c.x0 & 3) + c.w + 3) & ~3;
      tile = std::max(tile, tw * (int)c.h);
      npx = std::max(npx, std::max(c.w - 6, 0) * std::max(c.h - 6, 0));
    }
    d.fast_tile_bytes = (tile + 15) & ~15;
    d.fast_lds_per_wave = 2 * d.fast_tile_bytes + ((2 * npx + 15) & ~15);
    for (int l = 0; l < nlevels; ++l) {
      d.gauss_tile0[l] = t0;
      t0 += ((d.lvl_cols[l] + GT_W - 1) / GT_W) * ((d.lvl_rows[l] + GT_H - 1) / GT_H);
    }
    for (int l = nlevels; l <= MAX_LEVELS; ++l) d.gauss_tile0[l] = t0;
    for (int l = 1; l < nlevels; ++l) {   // cv::resize: inv_scale = dsize/ssize, scale = 1/inv_scale
      d.rs_scale_x[l] = 1. / ((double)d.lvl_cols[l] / d.lvl_cols[l - 1]);
      d.rs_scale_y[l] = 1. / ((double)d.lvl_rows[l] / d.lvl_rows[l - 1]);
    }
  }
  for (const Cell& c : cells)
    if (c.w > ROI_MAX || c.h > ROI_MAX) {
      ctx->set_error("ssx_orb: grid cell %dx%d exceeds the %d-px LDS tile", c.w, c.h, ROI_MAX);
      return SSX_ERR_UNSUPPORTED;
    }
  Layout lay;
  const size_t o_cells = lay.take(sizeof(Cell) * std::max<size_t>(cells.size(), 1));
  const size_t o_pyr = lay.take(d.pyr_bytes * I);
  const size_t o_mask = lay.take(has_mask ? d.pyr_bytes * I : 256);
  const size_t o_blur = lay.take(d.pyr_bytes * I);
  const size_t o_ccount = lay.take(sizeof(int) * (size_t)I * std::max(d.n_cells, 1));
  const size_t o_ccand = lay.take(sizeof(uint32_t) * (size_t)I * std::max(d.n_cells, 1) * CELL_CAP);
  const size_t o_oct = lay.take(OctLayout::total * (size_t)I * nlevels);
  const size_t o_ncand = lay.take(sizeof(int) * (size_t)I * nlevels);
  const size_t o_selc = lay.take(sizeof(int) * (size_t)I * nlevels);
  const size_t o_sel = lay.take(sizeof(uint32_t) * (size_t)I * nlevels * SEL_CAP);
  const size_t o_ang = lay.take(sizeof(float) * (size_t)I * nlevels * SEL_CAP);
  const size_t o_status = lay.take(sizeof(int) * (size_t)I);
  const size_t o_kps = lay.take(sizeof(ssx_keypoint) * (size_t)I * out_cap);
  const size_t o_desc = lay.take((size_t)32 * I * out_cap);
  const size_t o_n = lay.take(sizeof(int) * (size_t)I);
  SSX_HIP_TRY(ctx, ws->arena.reserve(lay.off));
  char* base = ws->arena.as<char>();
  d.cells = (const Cell*)(base + o_cells);
  d.pyr = (uint8_t*)(base + o_pyr);
  d.maskpyr = (uint8_t*)(base + o_mask);
  d.blur = (uint8_t*)(base + o_blur);
  d.cell_count = (int*)(base + o_ccount);
  d.cell_cand = (uint32_t*)(base + o_ccand);
  d.oct = (uint8_t*)(base + o_oct);
  d.lvl_ncand = (int*)(base + o_ncand);
  d.sel_count = (int*)(base + o_selc);
  d.sel = (uint32_t*)(base + o_sel);
  d.sel_angle = (float*)(base + o_ang);
  d.status = (int*)(base + o_status);
  d.out_kps = (uint8_t*)(base + o_kps);
  d.out_desc = (uint8_t*)(base + o_desc);
  d.out_n = (int*)(base + o_n);
  if (!cells.empty())
    SSX_HIP_TRY(ctx, hipMemcpyAsync(base + o_cells, cells.data(), sizeof(Cell) * cells.size(), hipMemcpyHostToDevice, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // `cells` is a pageable temporary
  ws->dev = d;
  ws->rows = rows; ws->cols = cols; ws->I = I; ws->nlevels = nlevels; ws->nfeatures = prm.nfeatures;
  ws->ini_th = prm.ini_th_fast; ws->min_th = prm.min_th_fast; ws->has_mask = has_mask; ws->detect_only = detect_only;
  ws->scale_factor = prm.scale_factor;
  ws->planned = true;
  return SSX_OK;
}

ssx_status stage_level0(ssx_ctx* ctx, const uint8_t* imgs_dev, int stride, size_t img_bytes, const uint8_t* masks_dev,
                        int mask_stride, size_t mask_bytes)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  const dim3 grid((d.lvl_cols[0] + 255) / 256, d.lvl_rows[0], d.I);
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0, grid, dim3(256), 0, ctx->stream, imgs_dev, stride, img_bytes, d.pyr, d.pyr_bytes,
                     d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0], 0, 0));
  if (d.has_mask)
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0, grid, dim3(256), 0, ctx->stream, masks_dev, mask_stride, mask_bytes, d.maskpyr,
                       d.pyr_bytes, d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0], 0, 0));
  SSX_HIP_TRY(ctx, hipGetLastError());
  return SSX_OK;
}

ssx_status run_pipeline(ssx_ctx* ctx)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  hipStream_t s = ctx->stream;
  SSX_HIP_TRY(ctx, hipMemsetAsync(d.status, 0, sizeof(int) * d.I, s));
  // pyramid (ComputePyramid): level l from level l-1
  for (int l = 1; l < d.nlevels; ++l) {
    const dim3 grid((d.lvl_cols[l] + 255) / 256, d.lvl_rows[l], d.I);
    SSX_PROF(ctx, KID_ORB_RESIZE, hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, d.pyr + d.lvl_off[l - 1], d.pyr + d.lvl_off[l], d.pyr_bytes,
                       d.lvl_rows[l - 1], d.lvl_cols[l - 1], d.lvl_pitch[l - 1], d.lvl_rows[l], d.lvl_cols[l], d.lvl_pitch[l],
                       d.rs_scale_x[l], d.rs_scale_y[l]));
    if (d.has_mask)
      SSX_PROF(ctx, KID_ORB_RESIZE, hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, d.maskpyr + d.lvl_off[l - 1], d.maskpyr + d.lvl_off[l], d.pyr_bytes,
                         d.lvl_rows[l - 1], d.lvl_cols[l - 1], d.lvl_pitch[l - 1], d.lvl_rows[l], d.lvl_cols[l], d.lvl_pitch[l],
                         d.rs_scale_x[l], d.rs_scale_y[l]));
  }
  // the blur only depends on the pyramid: it runs on the auxiliary stream, concurrently with detection
  // (HBM-streaming blur next to the latency-bound octree), and is joined before the descriptors.
  const bool fork = !d.detect_only && ctx->aux != nullptr;
  if (!d.detect_only) {
    hipStream_t gs = fork ? ctx->aux : s;
    if (fork) {
      SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
      SSX_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    }
    SSX_PROF_ON(ctx, gs, KID_ORB_GAUSS, hipLaunchKernelGGL(k_gauss7, dim3(d.gauss_tile0[d.nlevels], d.I), dim3(256), 0, gs, d));
    if (fork) SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
  }
  if (4 * (size_t)d.fast_lds_per_wave > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_cells), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(4 * (size_t)d.fast_lds_per_wave));
  if (d.n_cells > 0)
    SSX_PROF(ctx, KID_ORB_FAST, hipLaunchKernelGGL(k_fast_cells, dim3((d.n_cells + 3) / 4, d.I), dim3(256), 4 * (size_t)d.fast_lds_per_wave, s, d));
  SSX_PROF(ctx, KID_ORB_OCTREE, launch_octree(d, s));
  if (d.detect_only) {
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_finalize_detect, dim3((SEL_CAP + 255) / 256, d.I), dim3(256), 0, s, d));
  } else {
    int maxfeat = 0;
    for (int l = 0; l < d.nlevels; ++l) maxfeat = std::max(maxfeat, d.feat[l] + 4);
    const dim3 kgrid((maxfeat + 3) / 4, d.nlevels, d.I);
    SSX_PROF(ctx, KID_ORB_ORIENT, hipLaunchKernelGGL(k_orient, kgrid, dim3(256), 0, s, d));
    if (fork) SSX_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    SSX_PROF(ctx, KID_ORB_BRIEF, hipLaunchKernelGGL(k_brief, kgrid, dim3(256), 0, s, d));
  }
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_counts, dim3((d.I + 63) / 64), dim3(64), 0, s, d));
  SSX_HIP_TRY(ctx, hipGetLastError());
  return SSX_OK;
}

namespace {

// upload one host image (+ optional mask) and run; shared by ssx_orb_detect / ssx_orb_extract
ssx_status run_host_image(ssx_ctx* ctx, const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask,
                          int mask_stride, const ssx_orb_params& prm, bool detect_only)
{
  ssx_status st = plan(ctx, rows, cols, 1, prm, mask != nullptr, detect_only);
  if (st != SSX_OK) return st;
  OrbWorkspace* ws = get_ws(ctx);
  const size_t bytes = (size_t)rows * cols;
  SSX_HIP_TRY(ctx, ws->input.reserve(2 * bytes + 512));
  SSX_HIP_TRY(ctx, ws->stage.reserve(2 * bytes + 512));
  uint8_t* hs = ws->stage.as<uint8_t>();
  for (int y = 0; y < rows; ++y) memcpy(hs + (size_t)y * cols, img + (size_t)y * stride, cols);
  if (mask)
    for (int y = 0; y < rows; ++y) memcpy(hs + bytes + (size_t)y * cols, mask + (size_t)y * mask_stride, cols);
  SSX_HIP_TRY(ctx, hipMemcpyAsync(ws->input.p, hs, mask ? 2 * bytes : bytes, hipMemcpyHostToDevice, ctx->stream));
  st = stage_level0(ctx, ws->input.as<uint8_t>(), cols, bytes, ws->input.as<uint8_t>() + bytes, cols, bytes);
  if (st != SSX_OK) return st;
  return run_pipeline(ctx);
}

ssx_status fetch_image_result(ssx_ctx* ctx, int image, int cap, ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  int hn[2] = {0, 0};
  SSX_HIP_TRY(ctx, hipMemcpyAsync(&hn[0], d.out_n + image, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(&hn[1], d.status + image, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (hn[1] != 0) {
    ctx->set_error("ssx_orb: internal capacity exceeded (status bits %d: 1=candidates 2=octree nodes 4=outputs)", hn[1]);
    return SSX_ERR_CAPACITY;
  }
  *n = hn[0];
  if (hn[0] > cap) {
    ctx->set_error("ssx_orb: %d keypoints but capacity %d", hn[0], cap);
    return SSX_ERR_CAPACITY;
  }
  if (hn[0] > 0) {
    if (kps_out)
      SSX_HIP_TRY(ctx, hipMemcpyAsync(kps_out, d.out_kps + (size_t)image * d.out_cap * sizeof(ssx_keypoint),
                                      sizeof(ssx_keypoint) * hn[0], hipMemcpyDeviceToHost, ctx->stream));
    if (desc_out && !d.detect_only)
      SSX_HIP_TRY(ctx, hipMemcpyAsync(desc_out, d.out_desc + (size_t)image * d.out_cap * 32, (size_t)32 * hn[0],
                                      hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SSX_OK;
}

}  // namespace

ssx_status fetch_image(ssx_ctx* ctx, int image, int cap, ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n)
{
  return fetch_image_result(ctx, image, cap, kps_out, desc_out, n);
}

}  // namespace ssxorb

using namespace ssxorb;

extern "C" {

void ssx_orb_default_params(ssx_orb_params* p)
{
  if (!p) return;
  p->nfeatures = 2000; p->scale_factor = 1.2f; p->nlevels = 8; p->ini_th_fast = 20; p->min_th_fast = 7;
}

ssx_status ssx_orb_detect(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                          const uint8_t* mask, int32_t mask_stride, const ssx_orb_params* prm, int32_t cap,
                          ssx_keypoint* kps_out, int32_t* n)
{
  if (!ctx || !prm || !n) return SSX_ERR_INVALID_ARG;
  *n = 0;
  if (!img || rows <= 0 || cols <= 0) return SSX_OK;   // `if (_image.empty()) return;` orbextractor.cpp:758
  ssx_status st = run_host_image(ctx, img, stride, rows, cols, mask, mask_stride, *prm, true);
  if (st != SSX_OK) return st;
  return fetch_image_result(ctx, 0, cap, kps_out, nullptr, n);
}

ssx_status ssx_orb_extract(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                           const uint8_t* mask, int32_t mask_stride, const ssx_orb_params* prm, int32_t cap,
                           ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n)
{
  if (!ctx || !prm || !n) return SSX_ERR_INVALID_ARG;
  *n = 0;
  if (!img || rows <= 0 || cols <= 0) return SSX_OK;   // orbextractor.cpp:691
  ssx_status st = run_host_image(ctx, img, stride, rows, cols, mask, mask_stride, *prm, false);
  if (st != SSX_OK) return st;
  return fetch_image_result(ctx, 0, cap, kps_out, desc_out, n);
}

ssx_status ssx_orb_describe_at(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                               const ssx_orb_params* prm, const ssx_keypoint* kps_in, int32_t n_in, ssx_keypoint* kps_out,
                               uint8_t* desc_out, int32_t* n)
{
  if (!ctx || !prm || !n) return SSX_ERR_INVALID_ARG;
  *n = 0;
  if (!img || rows <= 0 || cols <= 0 || n_in <= 0 || !kps_in) return SSX_OK;   // LOG(ERROR) + return in the reference
  ssx_status st = plan(ctx, rows, cols, 1, *prm, false, false);
  if (st != SSX_OK) return st;
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  const size_t bytes = (size_t)rows * cols;
  Layout lay;
  const size_t o_img = lay.take(bytes);
  const size_t o_in = lay.take(sizeof(ssx_keypoint) * (size_t)n_in);
  const size_t in_bytes = lay.off;
  const size_t o_out = lay.take(sizeof(ssx_keypoint) * (size_t)n_in);
  const size_t o_desc = lay.take((size_t)32 * n_in);
  const size_t o_keep = lay.take((size_t)n_in);
  SSX_HIP_TRY(ctx, ws->input.reserve(lay.off));
  SSX_HIP_TRY(ctx, ws->stage.reserve(lay.off));
  char* hs = ws->stage.as<char>();
  for (int y = 0; y < rows; ++y) memcpy(hs + o_img + (size_t)y * cols, img + (size_t)y * stride, cols);
  memcpy(hs + o_in, kps_in, sizeof(ssx_keypoint) * n_in);
  char* base = ws->input.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  st = stage_level0(ctx, (const uint8_t*)(base + o_img), cols, bytes, nullptr, 0, 0);
  if (st != SSX_OK) return st;
  hipStream_t s = ctx->stream;
  for (int l = 1; l < d.nlevels; ++l) {   // ComputePyramid(image), orbextractor.cpp:1012-1027
    const dim3 grid((d.lvl_cols[l] + 255) / 256, d.lvl_rows[l], 1);
    SSX_PROF(ctx, KID_ORB_RESIZE, hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, d.pyr + d.lvl_off[l - 1], d.pyr + d.lvl_off[l], d.pyr_bytes,
                       d.lvl_rows[l - 1], d.lvl_cols[l - 1], d.lvl_pitch[l - 1], d.lvl_rows[l], d.lvl_cols[l], d.lvl_pitch[l],
                       d.rs_scale_x[l], d.rs_scale_y[l]));
  }
  SSX_PROF(ctx, KID_ORB_GAUSS, hipLaunchKernelGGL(k_gauss7, dim3(d.gauss_tile0[d.nlevels], 1), dim3(256), 0, s, d));
  DescribeAt a;
  a.in = (const ssx_keypoint*)(base + o_in); a.out = (ssx_keypoint*)(base + o_out); a.desc = (uint8_t*)(base + o_desc);
  a.keep = (uint8_t*)(base + o_keep); a.n_in = n_in;
  SSX_PROF(ctx, KID_ORB_BRIEF, hipLaunchKernelGGL(k_describe_at, dim3((n_in + 3) / 4), dim3(256), 0, s, d, a));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_out, base + o_out, lay.off - o_out, hipMemcpyDeviceToHost, s));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(s));
  // order-preserving compaction of the kept keypoints (out_keypoints.push_back order, :893)
  const ssx_keypoint* ok = (const ssx_keypoint*)(hs + o_out);
  const uint8_t* od = (const uint8_t*)(hs + o_desc);
  const uint8_t* keep = (const uint8_t*)(hs + o_keep);
  int m = 0;
  for (int i = 0; i < n_in; ++i) {
    if (!keep[i]) continue;
    if (kps_out) kps_out[m] = ok[i];
    if (desc_out) memcpy(desc_out + (size_t)32 * m, od + (size_t)32 * i, 32);
    ++m;
  }
  *n = m;
  return SSX_OK;
}

ssx_status ssx_orb_stage_level(ssx_ctx* ctx, int32_t image, int32_t level, int32_t blurred, uint8_t* out,
                               int32_t out_cap, int32_t* rows, int32_t* cols)
{
  if (!ctx || !ctx->orb || !ctx->orb->planned) return SSX_ERR_INVALID_ARG;
  const OrbDev& d = ctx->orb->dev;
  if (image < 0 || image >= d.I || level < 0 || level >= d.nlevels) return SSX_ERR_INVALID_ARG;
  const int r = d.lvl_rows[level], c = d.lvl_cols[level];
  if (rows) *rows = r;
  if (cols) *cols = c;
  if (!out) return SSX_OK;
  if (out_cap < r * c) return SSX_ERR_CAPACITY;
  const uint8_t* src = (blurred ? d.blur : d.pyr) + (size_t)image * d.pyr_bytes + d.lvl_off[level];
  SSX_HIP_TRY(ctx, hipMemcpy2DAsync(out, c, src, d.lvl_pitch[level], c, r, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

ssx_status ssx_orb_stage_candidates(ssx_ctx* ctx, int32_t image, int32_t level, int32_t cap, ssx_keypoint* out,
                                    int32_t* n)
{
  if (!ctx || !ctx->orb || !ctx->orb->planned || !n) return SSX_ERR_INVALID_ARG;
  const OrbDev& d = ctx->orb->dev;
  if (image < 0 || image >= d.I || level < 0 || level >= d.nlevels) return SSX_ERR_INVALID_ARG;
  const int c0 = d.lvl_cell0[level], c1 = d.lvl_cell0[level + 1];
  const int nc = c1 - c0;
  std::vector<int> cnt(std::max(nc, 1));
  std::vector<uint32_t> cand((size_t)std::max(nc, 1) * CELL_CAP);
  if (nc > 0) {
    SSX_HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), d.cell_count + (size_t)image * d.n_cells + c0, sizeof(int) * nc, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipMemcpyAsync(cand.data(), d.cell_cand + ((size_t)image * d.n_cells + c0) * CELL_CAP,
                                    sizeof(uint32_t) * (size_t)nc * CELL_CAP, hipMemcpyDeviceToHost, ctx->stream));
    SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  int m = 0;
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < cnt[c]; ++k, ++m) {
      if (m >= cap || !out) continue;
      const uint32_t p = cand[(size_t)c * CELL_CAP + k];
      ssx_keypoint kp;
      kp.x = (float)(p & 0xFFF); kp.y = (float)((p >> 12) & 0xFFF); kp.size = 7.f; kp.angle = -1.f;
      kp.response = (float)(p >> 24); kp.octave = 0; kp.class_id = -1;
      out[m] = kp;
    }
  *n = m;
  return (out && m > cap) ? SSX_ERR_CAPACITY : SSX_OK;
}

}  // extern "C"
