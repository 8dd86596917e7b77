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
//   k_resize         level l from level l-1, host-tabulated indices/weights, 4 pixels per thread           (A5)
//   k_fast_cells     ONE WORKGROUP PER GRID CELL: ROI (<= 72x72 bytes) staged in LDS from coalesced row reads,
//                    segment test + cornerScore at iniThFAST, fallback to minThFAST when the cell is empty,
//                    3x3 NMS inside the cell, mask test, ordered (row-major) compaction                     (A1+A2)
//   k_octree         ONE WORKGROUP PER (image, level): data-parallel DistributeOctTree (A3); the formulation is
//                    tools/octree_model.py -- stable 4-way partitions by packed prefix sums, list order by scans
//   k_gauss7         separable 7x7 sigma=2 in Q8 fixed point, 128x32 tiles with halo in LDS                  (A7)
//   k_orient_brief   one wave per output keypoint: raw 31x31 + blurred 37x37 patches staged in LDS as aligned
//                    dwords; integer moments + fastAtan2 polynomial (A6), then the steered BRIEF tests, 4 per
//                    lane, ballot-packed (A7), and the keypoint record
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "ctx.hpp"
#include "orb_ws.hpp"
#include "../../include/ssx_test_hooks.h"

namespace ssxorb {

__constant__ __attribute__((aligned(16))) int8_t c_pattern[256 * 4] = {
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
// The source indices and the two 11-bit weights of a destination column depend on the column only (rows likewise),
// so plan() tabulates them once per level on the host with the exact float sequence of cv::resize.  One thread =
// 4 destination pixels of one row = one QUAD row of the column table (ResizeQuad): the 8 source bytes starting at
// s0 (ONE unaligned 8-byte load per source row) hold every left/right neighbour of the quad, two v_perm_b32 with
// the tabulated selectors gather them, then the integer blend and one aligned dword store; a thread does this
// for RS_ROWS consecutive rows with all row loads in flight together.  A 64x4 block covers 256 columns x 16 rows.  WIDE8 = false (a level whose quads span more than 8 source bytes, scale factor > 2)
// falls back to byte loads.
struct ResizeQuad {          // 32 bytes
  uint32_t s0;               // first source column of the quad
  uint32_t sel0, sel1;       // byte k: (sx[k] - s0), (sx1[k] - s0)      (v_perm selectors when all <= 7)
  uint32_t pad;
  uint32_t w[4];             // a0 | a1 << 16 per destination column (0 for columns past dcols)
};

constexpr int RS_ROWS = 4;   // destination rows per thread: their 8 source-row loads are issued together
constexpr int RS_LOOP = 4;   // such groups per wave, one after the other, in a batch (a frame or two: 1 -- the chain of seven levels is latency bound)
template <bool WIDE8>
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base,
                                                size_t img_stride_bytes, int spitch, int drows, int dcols, int dpitch,
                                                const ResizeQuad* __restrict__ xtab, const uint2* __restrict__ ytab, int loops)
{
  const int q = blockIdx.x * 64 + threadIdx.x;      // quad of destination columns
  if (4 * q >= dcols) return;
  const uint8_t* src = src_base + (size_t)blockIdx.z * img_stride_bytes;
  uint8_t* dst = dst_base + (size_t)blockIdx.z * img_stride_bytes;
  const uint4 ta = reinterpret_cast<const uint4*>(xtab)[2 * q], tw = reinterpret_cast<const uint4*>(xtab)[2 * q + 1];
  // RS_LOOP groups of RS_ROWS rows per wave: a quarter of the workgroups (the big levels launched 12 800 of them for 82 us:
  // their rate, not their work, set the time) and one column-table fetch for 16 rows instead of 4
#pragma unroll 1
  for (int it = 0; it < loops; ++it) {
  // a wave = RS_ROWS consecutive destination rows of one quad column range: row tables and bases are scalar
  const int dy0 = ((blockIdx.y * loops + it) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * RS_ROWS;
  if (dy0 >= drows) break;
  // the kernel is bound by dependent memory round trips (table -> source rows), not by bandwidth or VALU: issue
  // all source-row loads of the RS_ROWS rows before the first use
  uint32_t p00[RS_ROWS], p01[RS_ROWS], p10[RS_ROWS], p11[RS_ROWS];
  int b0[RS_ROWS], b1[RS_ROWS];
  uint2 w0[RS_ROWS], w1[RS_ROWS];
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint2 yt = ytab[min(dy0 + r, drows - 1)];
    b0[r] = (int)(yt.y & 0xFFFFu); b1[r] = (int)(yt.y >> 16);
    const uint8_t* r0 = src + (size_t)(yt.x & 0xFFFFu) * spitch + ta.x;
    const uint8_t* r1 = src + (size_t)(yt.x >> 16) * spitch + ta.x;
    if (WIDE8) {
      __builtin_memcpy(&w0[r], r0, 8);
      __builtin_memcpy(&w1[r], r1, 8);
    } else {
      p00[r] = p01[r] = p10[r] = p11[r] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o0 = (ta.y >> (8 * k)) & 0xFF, o1 = (ta.z >> (8 * k)) & 0xFF;
        p00[r] |= (uint32_t)r0[o0] << (8 * k); p01[r] |= (uint32_t)r0[o1] << (8 * k);
        p10[r] |= (uint32_t)r1[o0] << (8 * k); p11[r] |= (uint32_t)r1[o1] << (8 * k);
      }
    }
  }
  const uint32_t wa[4] = {tw.x, tw.y, tw.z, tw.w};
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    if (WIDE8) {
      p00[r] = __builtin_amdgcn_perm(w0[r].y, w0[r].x, ta.y); p01[r] = __builtin_amdgcn_perm(w0[r].y, w0[r].x, ta.z);
      p10[r] = __builtin_amdgcn_perm(w1[r].y, w1[r].x, ta.y); p11[r] = __builtin_amdgcn_perm(w1[r].y, w1[r].x, ta.z);
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a0 = (int)(wa[k] & 0xFFFFu), a1 = (int)(wa[k] >> 16);
      const int S0 = (int)((p00[r] >> (8 * k)) & 0xFF) * a0 + (int)((p01[r] >> (8 * k)) & 0xFF) * a1;
      const int S1 = (int)((p10[r] >> (8 * k)) & 0xFF) * a0 + (int)((p11[r] >> (8 * k)) & 0xFF) * a1;
      const uint32_t v = (uint32_t)((((b0[r] * (S0 >> 4)) >> 16) + ((b1[r] * (S1 >> 4)) >> 16) + 2) >> 2);
      out |= (v & 0xFFu) << (8 * k);
    }
    if (dy0 + r < drows) *reinterpret_cast<uint32_t*>(dst + (size_t)(dy0 + r) * dpitch + 4 * q) = out;
  }
  }   // row groups
}

// copy a pitched host-layout image into level 0 of the pyramid (pitch change): 8 destination bytes per thread
// (the source rows of a 1241-px image are not dword aligned, the destination rows always are)
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ in, int in_stride, size_t in_img_bytes,
                                                     uint8_t* __restrict__ dst_base, size_t img_stride_bytes,
                                                     int rows, int cols, int dpitch)
{
  const int x = (blockIdx.x * 64 + threadIdx.x) * 8, y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (x >= cols || y >= rows) return;
  const uint8_t* srow = in + (size_t)blockIdx.z * in_img_bytes + (size_t)y * in_stride;
  uint32_t w[2] = {0u, 0u};
  if (((in_stride | in_img_bytes | reinterpret_cast<uintptr_t>(in)) & 7) == 0) {   // 8-byte rows (may run into the row's padding)
    const uint2 v = *reinterpret_cast<const uint2*>(srow + x);
    w[0] = v.x; w[1] = v.y;
    const int valid = cols - x;   // >= 1
    if (valid < 8) { const uint64_t keep = (~0ull) >> (8 * (8 - valid)); w[0] &= (uint32_t)keep; w[1] &= (uint32_t)(keep >> 32); }
  } else {
    // rows that start anywhere (a 1241-pixel row): the three ALIGNED dwords that hold the eight bytes, cut with v_alignbyte
    // (eight byte loads per thread before: 0.10 ms per 256 images for 0.03 ms of traffic).  A dword that holds one valid byte is
    // inside the allocation; the ones behind it are clamped to the dword of the input's last byte.
    const uintptr_t p = reinterpret_cast<uintptr_t>(srow + x), a = p & ~uintptr_t(3);
    const uintptr_t last = (reinterpret_cast<uintptr_t>(in) + (size_t)(gridDim.z - 1) * in_img_bytes + (size_t)(rows - 1) * in_stride + (size_t)(cols - 1)) & ~uintptr_t(3);
    const uint32_t d0 = *reinterpret_cast<const uint32_t*>(a);
    const uint32_t d1 = *reinterpret_cast<const uint32_t*>(a + 4 < last ? a + 4 : last);
    const uint32_t d2 = *reinterpret_cast<const uint32_t*>(a + 8 < last ? a + 8 : last);
    const uint32_t sh = (uint32_t)(p & 3);
    w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const int valid = cols - x;   // >= 1
    if (valid < 8) { const uint64_t keep = (~0ull) >> (8 * (8 - valid)); w[0] &= (uint32_t)keep; w[1] &= (uint32_t)(keep >> 32); }
  }
  *reinterpret_cast<uint2*>(dst_base + (size_t)blockIdx.z * img_stride_bytes + (size_t)y * dpitch + x) = make_uint2(w[0], w[1]);
}

// the same for a batch whose images lie in buffers of their own (one pointer per image: device memory, or pinned host memory read
// over PCIe): ssx_orb_detect_boxes_batch.  Rows may start anywhere: the aligned-dword path of k_copy_level0, clamped per image.
__global__ __launch_bounds__(256) void k_copy_level0_ptrs(const uint8_t* const* __restrict__ imgs, int in_stride, uint8_t* __restrict__ dst_base,
                                                          size_t img_stride_bytes, int rows, int cols, int dpitch)
{
  const int x = (blockIdx.x * 64 + threadIdx.x) * 8, y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (x >= cols || y >= rows) return;
  const uint8_t* in = imgs[blockIdx.z];
  const uint8_t* srow = in + (size_t)y * in_stride;
  const uintptr_t p = reinterpret_cast<uintptr_t>(srow + x), a = p & ~uintptr_t(3);
  const uintptr_t last = (reinterpret_cast<uintptr_t>(in) + (size_t)(rows - 1) * in_stride + (size_t)(cols - 1)) & ~uintptr_t(3);
  const uint32_t d0 = *reinterpret_cast<const uint32_t*>(a);
  const uint32_t d1 = *reinterpret_cast<const uint32_t*>(a + 4 < last ? a + 4 : last);
  const uint32_t d2 = *reinterpret_cast<const uint32_t*>(a + 8 < last ? a + 8 : last);
  const uint32_t sh = (uint32_t)(p & 3);
  uint32_t w[2];
  w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
  w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
  const int valid = cols - x;   // >= 1
  if (valid < 8) { const uint64_t keep = (~0ull) >> (8 * (8 - valid)); w[0] &= (uint32_t)keep; w[1] &= (uint32_t)(keep >> 32); }
  *reinterpret_cast<uint2*>(dst_base + (size_t)blockIdx.z * img_stride_bytes + (size_t)y * dpitch + x) = make_uint2(w[0], w[1]);
}

// A4, the mask of FrontEnd::DetectFeatures (frontend.cpp:302-312) rasterised on the device: level 0 of the mask pyramid was
// set to 255; one workgroup per box clears its rectangle [x0, x1] x [y0, y1] (inclusive, already clipped by the caller's
// contract -- clipped again here).  16 bytes per tracked feature cross PCIe instead of rows x cols bytes of mask.
__global__ __launch_bounds__(64) void k_mask_boxes(const int4* __restrict__ boxes, int n_boxes, uint8_t* __restrict__ mask0, int rows, int cols, int pitch)
{
  const int4 b = boxes[blockIdx.x];
  const int x0 = max(b.x, 0), y0 = max(b.y, 0), x1 = min(b.z, cols - 1), y1 = min(b.w, rows - 1);
  const int w = x1 - x0 + 1, h = y1 - y0 + 1;
  if (w <= 0 || h <= 0) return;
  for (int i = threadIdx.x; i < w * h; i += 64) {
    const int r = i / w, c = i - r * w;
    mask0[(size_t)(y0 + r) * pitch + x0 + c] = 0;
  }
}

// the boxes of a BATCH of images: box b belongs to image box_img[b] (ssx_orb_detect_boxes_batch)
__global__ __launch_bounds__(64) void k_mask_boxes_b(const int4* __restrict__ boxes, const int* __restrict__ box_img, uint8_t* __restrict__ maskpyr,
                                                     size_t img_stride_bytes, int rows, int cols, int pitch)
{
  const int4 b = boxes[blockIdx.x];
  uint8_t* mask0 = maskpyr + (size_t)box_img[blockIdx.x] * img_stride_bytes;
  const int x0 = max(b.x, 0), y0 = max(b.y, 0), x1 = min(b.z, cols - 1), y1 = min(b.w, rows - 1);
  const int w = x1 - x0 + 1, h = y1 - y0 + 1;
  if (w <= 0 || h <= 0) return;
  for (int i = threadIdx.x; i < w * h; i += 64) {
    const int r = i / w, c = i - r * w;
    mask0[(size_t)(y0 + r) * pitch + x0 + c] = 0;
  }
}
__global__ __launch_bounds__(256) void k_fill_level0_b(uint8_t* __restrict__ base, size_t img_stride_bytes, size_t level_bytes, uint32_t v)
{
  uint32_t* p = reinterpret_cast<uint32_t*>(base + (size_t)blockIdx.y * img_stride_bytes);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < level_bytes / 4; i += (size_t)gridDim.x * 256) p[i] = v;
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

// segment test + cornerScore<16> (OpenCV) for the pixel at address c (row pitch p): returns -1 when the pixel is
// not a FAST-9 corner at threshold t, else OpenCV's cornerScore (the largest threshold at which the pixel is still
// a corner), which is >= t and <= 254.
//
// With d[k] = v - ring[k], a0 = max over the 16 arcs of (min of d over the 9-arc) and b0 = min over arcs of (max
// over the arc): the pixel is a dark corner at t iff a0 > t and a bright one iff b0 < -t, and cornerScore is
// max(a0, -b0) - 1 once both are clamped at t - so the segment test needs no bit masks, it falls out of the
// score.  The ring is held as 8 packed i16 pairs (d[k], d[k+8]); every sliding-window min/max then runs on
// v_pk_min_i16 / v_pk_max_i16 and the "+8" rotations are half swaps (op_sel, free).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ s16x2 hswap(s16x2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

__device__ __forceinline__ int fast_score(const uint8_t* c, int p, int t)
{
  const short v = (short)c[0];
  // seven row bases (one add each), the column offsets 0..6 ride in the load instructions' immediate field
  constexpr int RDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int RDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const uint8_t* rowp[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) rowp[j] = c - 3 + (j - 3) * p;
  s16x2 D[12];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const short r0 = (short)rowp[RDY[k] + 3][RDX[k] + 3];
    const short r1 = (short)rowp[RDY[k + 8] + 3][RDX[k + 8] + 3];
    D[k] = s16x2{v, v} - s16x2{r0, r1};
  }
  // index j >= 8 of any packed array is the half swap of index j - 8
  s16x2 mn[12], mx[12];
  D[8] = hswap(D[0]);
#pragma unroll
  for (int k = 0; k < 8; ++k) { mn[k] = pmin(D[k], D[k + 1]); mx[k] = pmax(D[k], D[k + 1]); }
  mn[8] = hswap(mn[0]); mn[9] = hswap(mn[1]); mx[8] = hswap(mx[0]); mx[9] = hswap(mx[1]);
  s16x2 mn4[12], mx4[12];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mn4[k] = pmin(mn[k], mn[k + 2]); mx4[k] = pmax(mx[k], mx[k + 2]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { mn4[8 + k] = hswap(mn4[k]); mx4[8 + k] = hswap(mx4[k]); }
  s16x2 a0 = s16x2{(short)t, (short)t}, b0 = s16x2{255, 255};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const s16x2 m8 = pmin(mn4[k], mn4[k + 4]);      // min of d over k .. k+7 (and k+8 .. k+15)
    const s16x2 x8 = pmax(mx4[k], mx4[k + 4]);
    const s16x2 d8 = hswap(D[k]);                   // d[k+8], d[k+16]
    a0 = pmax(a0, pmin(m8, d8));
    b0 = pmin(b0, pmax(x8, d8));
  }
  const int a = max((int)a0.x, (int)a0.y), b = min((int)b0.x, (int)b0.y);
  const int sc = max(a, -b) - 1;                    // == -min(-a, b) - 1 of OpenCV with a0 seeded at t
  return sc >= t ? sc : -1;
}

// ONE WAVE PER CELL (4 cells per 256-thread workgroup, no workgroup barrier anywhere):
//   1. the ROI is staged in LDS with aligned dword loads (level pitch is a multiple of 128 bytes);
//   2. every interior pixel takes the quick test (an arc of 9 contiguous ring pixels contains at least 2 of the 4
//      compass pixels, so a corner needs >= 2 of them darker than v-t or >= 2 brighter than v+t), four pixels per
//      lane on packed u16 pairs; survivors are compacted with ballot/popcount;
//   3. the compacted survivors take the full segment test + cornerScore (dense lanes);
//   4. the corners (again a compacted list) take the 3x3 strict NMS, the mask test and an ORDER-PRESERVING
//      compaction (ballot prefix inside a round; rounds walk the pixels in row-major order).
__global__ __launch_bounds__(256) void k_fast_cells(OrbDev o, int cell_begin, int cell_end)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
  const int cell = cell_begin + blockIdx.x * (blockDim.x >> 6) + wave, img = blockIdx.y;   // 1..4 cells per workgroup
  if (cell >= cell_end) return;
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
  {
    // lane -> (row, dword) once; every further round advances by 64 dwords = q64 rows + r64 dwords (wave-uniform)
    const float inv_ndw = 1.0f / (float)ndw;
    const int y0l = (int)(((float)lane + 0.5f) * inv_ndw);           // exact: lane < 64, ndw <= 19
    int xd = lane - y0l * ndw;
    const int q64 = 64 / ndw, r64 = 64 - q64 * ndw;
    const uint8_t* src = lvl + (size_t)(c.y0 + y0l) * pitch + tx0 + 4 * xd;
    const int step = q64 * pitch + 4 * r64, wrap = pitch - 4 * ndw;
    // eight rounds of loads in flight before the first LDS store: one memory round trip per 512 dwords instead of one
    // per 64 (the staging was 40 % of a wave's life: 9 600 of 23 500 cycles for a 37x38 ROI)
    const int total = h * ndw;
    for (int i0 = lane; i0 < total; i0 += 8 * 64) {
      uint32_t v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = (i0 + 64 * r < total) ? *reinterpret_cast<const uint32_t*>(src) : 0u;
        xd += r64;
        src += step;
        if (xd >= ndw) { xd -= ndw; src += wrap; }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (i0 + 64 * r < total) reinterpret_cast<uint32_t*>(sImg)[i0 + 64 * r] = v[r];
    }
  }
  const int iw = w - 6, ih = h - 6;                // cv::FAST ignores a 3-px border of the ROI
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
    // -- stage 1: quick test on every interior pixel, FOUR pixels of one row per lane, compaction of the
    //    survivors in row-major order.  Five aligned LDS dwords (centre row x4-4, x4, x4+4; rows y-3 and y+3 at x4)
    //    hold the centre and the four compass pixels of the quad; as u16 pairs (pixels 0|2 and 1|3) the second
    //    smallest / second largest compass pixel come from a 4-input min/max network on v_pk_min/max_u16, and
    //    "two compass pixels darker than v-t" is  sat(sat(v-t) - second_smallest) != 0.
    int n_surv = 0;
    {
      const int xs = xoff + 3, xe = xs + iw;            // interior columns of the tile
      const int qx0 = xs & ~3;
      const int nq = (iw > 0 && ih > 0) ? ((xe - qx0 + 3) >> 2) : 0;   // quads per interior row (<= 18)
      // lane -> (row inside a round, quad) ONCE: a round covers R = 64 / nq whole rows, so the quad column, its
      // border mask and the dword index are loop invariants and a round only adds R rows
      const int R = nq > 0 ? 64 / nq : 0;
      const float inv_nq = nq > 0 ? 1.0f / (float)nq : 0.f;
      const int rl = (int)(((float)lane + 0.5f) * inv_nq), ql = lane - rl * nq;
      const bool lane_on = nq > 0 && rl < R;
      const int x4 = qx0 + 4 * (lane_on ? ql : 0);
      bool vk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) vk[k] = lane_on && (x4 + k >= xs) && (x4 + k < xe);
      const uint32_t t2 = (uint32_t)th | ((uint32_t)th << 16);
      const uint32_t* dw = reinterpret_cast<const uint32_t*>(sImg);
      int ci = (3 + (lane_on ? rl : 0)) * ndw + (x4 >> 2);
      int off0 = (3 + (lane_on ? rl : 0)) * tw + x4;
      for (int row0 = 0; row0 < ih; row0 += R, ci += R * ndw, off0 += R * tw) {
        const bool in = lane_on && (row0 + rl < ih);
        const int cj = in ? ci : 3 * ndw + 1;                          // a harmless address for idle lanes
        const uint32_t C0 = dw[cj - 1], C1 = dw[cj], C2 = dw[cj + 1], Tp = dw[cj - 3 * ndw], Bt = dw[cj + 3 * ndw];
        const uint32_t Lf = __builtin_amdgcn_alignbyte(C1, C0, 1);     // pixels x-3 of the quad
        const uint32_t Rt = __builtin_amdgcn_alignbyte(C2, C1, 3);     // pixels x+3
        uint32_t flag[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {                            // pixels 0|2, then 1|3
          const int sh = 8 * par;
          const u16x2 v = as_u16x2((C1 >> sh) & 0x00FF00FFu), tp = as_u16x2((Tp >> sh) & 0x00FF00FFu);
          const u16x2 bt = as_u16x2((Bt >> sh) & 0x00FF00FFu), lf = as_u16x2((Lf >> sh) & 0x00FF00FFu);
          const u16x2 rt = as_u16x2((Rt >> sh) & 0x00FF00FFu);
          const u16x2 a = __builtin_elementwise_min(tp, bt), bmx = __builtin_elementwise_max(tp, bt);
          const u16x2 c2 = __builtin_elementwise_min(lf, rt), d2 = __builtin_elementwise_max(lf, rt);
          const u16x2 m1 = __builtin_elementwise_max(a, c2), m2 = __builtin_elementwise_min(bmx, d2);
          const u16x2 s2 = __builtin_elementwise_min(m1, m2), s3 = __builtin_elementwise_max(m1, m2);
          const u16x2 lo = __builtin_elementwise_sub_sat(v, as_u16x2(t2)), hi = v + as_u16x2(t2);
          const u16x2 f = __builtin_elementwise_sub_sat(lo, s2) | __builtin_elementwise_sub_sat(s3, hi);
          flag[par] = __builtin_bit_cast(uint32_t, f);
        }
        bool fk[4];
        fk[0] = (flag[0] & 0xFFFFu) != 0; fk[1] = (flag[1] & 0xFFFFu) != 0;
        fk[2] = (flag[0] >> 16) != 0;     fk[3] = (flag[1] >> 16) != 0;
        int before = n_surv, mine = 0, total = 0;
        unsigned long long bal[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fk[k] = fk[k] && in && vk[k];
          bal[k] = __ballot(fk[k]);
          before += __popcll(bal[k] & lt_mask);
          total += __popcll(bal[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (fk[k]) sList[before + mine] = (uint16_t)(off0 + k);
          mine += fk[k];
        }
        n_surv += total;
      }
    }
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
// is the image.  Input rows are fetched as aligned dwords into LDS (interior tiles take a branch-free path).
//   row pass     thread = (row PAIR, quad of 4 pixels): a pixel's 7 taps are two v_dot4_u32_u8 on byte windows cut
//                with v_alignbyte from three LDS dwords; the sums (<= 255 * 257, exact) of the two rows are packed
//                as u16 pairs, one int4 store per quad;
//   column pass  thread = 4 pixels x 4 rows: five int4 loads give the ten input rows as vertical u16 pairs, and
//                every output pixel is four v_dot2_u32_u16 (the weight pairs depend on the row parity), seeded
//                with the 2^15 rounding term.
// The fixed-point result is exact integer arithmetic, so the evaluation order is free (OpenCV: row sums exact,
// column pass (sum + 2^15) >> 16, saturate).
__device__ __forceinline__ uint32_t udot2(uint32_t pair, uint32_t w, uint32_t acc)
{
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, w), acc, false);
}

constexpr int GT_RUN = 4;                       // consecutive tiles of one tile row per workgroup of a batch (OrbDev::gauss_run): the next tile's loads fly during this tile's passes
constexpr int GT_NLD = ((GT_H + 6) * (GT_PITCH / 4) + 255) / 256;   // dwords of an input tile per thread (6)

// the dwords thread t holds of input tile (x0, y0): aligned loads, every tile alike.  Rows outside the image: the load goes to the
// BORDER_REFLECT_101 row.  Columns outside it: the dword index is clamped into the row, and the (at most three) bytes left of
// column 0 / right of the last column are mirrored INSIDE LDS once the tile is there (gauss_fix_edges: 38 x 6 byte copies and
// one more barrier, only in tiles that touch the left / right edge).  One reflection is exact for the rows / columns that reach a
// stored pixel of the image (at most 3 outside it); what lies further out only feeds outputs that are never stored or land in
// the row padding.
// [Until round 4 the first 3 x 38 threads of a border tile -- 45 % of all tiles -- assembled the dwords that straddle the left /
// right edge from single reflected bytes read from global memory, and interior tiles took a branch of their own: 108 VGPRs,
// 0.326 ms per 256 images alone on the chip against 0.21 with every tile read like an interior one.]
__device__ __forceinline__ void gauss_load_tile(const uint8_t* __restrict__ src, int rows, int cols, int pitch, int x0, int y0, int t, uint32_t (&v)[GT_NLD])
{
  constexpr int NDW = GT_PITCH / 4;   // 34 dwords per tile row
  if (rows >= 8 && cols >= 8) {
    const int gx_max = (cols - 1) & ~3;                     // the dword of the last column (the pitch is a multiple of 128: it is inside the row)
#pragma unroll
    for (int k = 0; k < GT_NLD; ++k) {
      const int i = t + 256 * k;
      const int r = i / NDW, dwi = i - r * NDW;
      int gy = y0 + r - 3;
      gy = gy < 0 ? -gy : gy;
      gy = gy >= rows ? 2 * rows - 2 - gy : gy;
      gy = min(max(gy, 0), rows - 1);
      const int gx = min(max(x0 - 4 + 4 * dwi, 0), gx_max);
      v[k] = i < (GT_H + 6) * NDW ? *reinterpret_cast<const uint32_t*>(src + (size_t)gy * pitch + gx) : 0u;
    }
  } else {
    // (levels of a few pixels: the general reflection, byte by byte)
#pragma unroll 1
    for (int k = 0; k < GT_NLD; ++k) {
      const int i = t + 256 * k;
      const int r = i / NDW, dwi = i - r * NDW;
      uint32_t w = 0u;
      if (i < (GT_H + 6) * NDW) {
        const int gy = reflect101(y0 + r - 3, rows);
        const int gx = x0 - 4 + 4 * dwi;
        const uint8_t* row = src + (size_t)gy * pitch;
        w = (uint32_t)row[reflect101(gx, cols)] | ((uint32_t)row[reflect101(gx + 1, cols)] << 8) |
            ((uint32_t)row[reflect101(gx + 2, cols)] << 16) | ((uint32_t)row[reflect101(gx + 3, cols)] << 24);
      }
      v[k] = w;
    }
  }
}

// does tile x0 hold a column left of 0 or right of cols - 1 that a stored output needs?  (block-uniform)
__device__ __forceinline__ bool gauss_tile_at_edge(int rows, int cols, int x0)
{
  return rows >= 8 && cols >= 8 && (x0 == 0 || x0 + GT_W + 4 > cols);
}
// mirror the bytes of columns -3 .. -1 and cols .. cols + 2 inside the LDS tile (tile byte b of a row = column x0 - 4 + b)
__device__ __forceinline__ void gauss_fix_edges(uint8_t* sIn, int cols, int x0, int t)
{
  if (t < 6 * (GT_H + 6)) {
    const int r = t / 6, k = t - 6 * r;
    const int x = k < 3 ? k - 3 : cols + k - 3;             // the column to fill
    const int xs = k < 3 ? -x : 2 * cols - 2 - x;           // its BORDER_REFLECT_101 source (inside the image: cols >= 8)
    const int b = x - x0 + 4, bs = xs - x0 + 4;
    if (b >= 0 && b < GT_PITCH && bs >= 0 && bs < GT_PITCH) sIn[r * GT_PITCH + b] = sIn[r * GT_PITCH + bs];
  }
}

__global__ __launch_bounds__(256) void k_gauss7(OrbDev o)
{
  __shared__ __attribute__((aligned(16))) uint8_t sIn[(GT_H + 6) * GT_PITCH];
  __shared__ __attribute__((aligned(16))) uint32_t sRowP[(GT_H + 6) / 2 * GT_W];   // [row pair][x]: sum(2rp) | sum(2rp+1) << 16
  static_assert((GT_H + 6) % 2 == 0, "rows are processed in pairs");
  const int img = blockIdx.y, t = threadIdx.x;
  int level = 0;
  while (level + 1 < o.nlevels && (int)blockIdx.x >= o.gauss_tile0[level + 1]) ++level;
  const int grp = blockIdx.x - o.gauss_tile0[level];             // a run of GT_RUN tiles of one tile row
  const int rows = o.lvl_rows[level], cols = o.lvl_cols[level], pitch = o.lvl_pitch[level];
  const int tiles_x = (cols + GT_W - 1) / GT_W;
  const int run = o.gauss_run;
  const int groups_x = (tiles_x + run - 1) / run;
  const int ty_ = grp / groups_x, tx0 = (grp - ty_ * groups_x) * run;
  const int tx1 = min(tx0 + run, tiles_x);
  const int y0 = ty_ * GT_H;
  const uint8_t* src = o.pyr + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  uint8_t* dst = o.blur + (size_t)img * o.pyr_bytes + o.lvl_off[level];
  constexpr int NDW = GT_PITCH / 4;   // 34 dwords per tile row
  uint32_t ld[GT_NLD];
  gauss_load_tile(src, rows, cols, pitch, tx0 * GT_W, y0, t, ld);
  // The results of a tile stay in registers and are stored one tile LATE, behind the next tile's loads: loads and stores share
  // one counter on this target (vmcnt) and complete out of order with respect to each other, so the wait for a tile's input is
  // s_waitcnt vmcnt(0) -- with the stores issued right after the column pass it also waited for their completion, every tile.
  uint32_t outv[4];
  const int og = t & 31, orq = t >> 5;                                // (the column pass's pixel quad and row quad)
  auto store_tile = [&](int x0s) {
    const int x = x0s + 4 * og;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + 4 * orq + j;
      // the pitch is a multiple of the tile width: the dword store stays inside the row's padding
      if (y < rows && x < pitch) *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x) = outv[j];
    }
  };
#pragma unroll 1
  for (int tx_ = tx0; tx_ < tx1; ++tx_) {
  const int x0 = tx_ * GT_W;
  if (tx_ > tx0) __syncthreads();                                  // the previous tile's column pass has read sRowP, its row pass sIn
#pragma unroll
  for (int k = 0; k < GT_NLD; ++k) {
    const int i = t + 256 * k;
    if (i < (GT_H + 6) * NDW) reinterpret_cast<uint32_t*>(sIn)[i] = ld[k];
  }
  if (tx_ + 1 < tx1) gauss_load_tile(src, rows, cols, pitch, x0 + GT_W, y0, t, ld);   // in flight during the two passes below
  if (tx_ > tx0) store_tile(x0 - GT_W);
  if (gauss_tile_at_edge(rows, cols, x0)) {                        // (block-uniform)
    __syncthreads();
    gauss_fix_edges(sIn, cols, x0, t);
  }
  __syncthreads();
  // row pass: tile column x sits at byte x + 4 of a tile row, so the taps of pixels 4g .. 4g+3 are bytes
  // 4g+1 .. 4g+10 = bytes 1 .. 10 of the dwords g, g+1, g+2
  constexpr uint32_t W_LO = 18u | (34u << 8) | (49u << 16) | (55u << 24);   // taps 0..3
  constexpr uint32_t W_HI = 49u | (34u << 8) | (18u << 16);                  // taps 4..6 (+ a zero weight)
  for (int i = t; i < (GT_H + 6) / 2 * (GT_W / 4); i += 256) {
    const int rp = i / (GT_W / 4), g = i - rp * (GT_W / 4);
    uint32_t sum[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t* rowdw = reinterpret_cast<const uint32_t*>(sIn + (2 * rp + h) * GT_PITCH) + g;
      const uint32_t d0 = rowdw[0], d1 = rowdw[1], d2 = rowdw[2];
      sum[h][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), W_LO, 0u, false);
      sum[h][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), W_HI, sum[h][0], false);
      sum[h][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), W_LO, 0u, false);
      sum[h][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), W_HI, sum[h][1], false);
      sum[h][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), W_LO, 0u, false);
      sum[h][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), W_HI, sum[h][2], false);
      sum[h][3] = __builtin_amdgcn_udot4(d1, W_LO, 0u, false);
      sum[h][3] = __builtin_amdgcn_udot4(d2, W_HI, sum[h][3], false);
    }
    uint4 pk;
    pk.x = sum[0][0] | (sum[1][0] << 16); pk.y = sum[0][1] | (sum[1][1] << 16);
    pk.z = sum[0][2] | (sum[1][2] << 16); pk.w = sum[0][3] | (sum[1][3] << 16);
    *reinterpret_cast<uint4*>(&sRowP[rp * GT_W + 4 * g]) = pk;
  }
  __syncthreads();
  // column pass: thread -> 4 pixels x 4 rows (output rows 4rq .. 4rq+3 need tile rows 4rq .. 4rq+9 = pairs 2rq .. 2rq+4)
  {
    const int g = t & 31, rq = t >> 5;             // 32 groups x 8 row-quads
    uint4 P[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) P[k] = *reinterpret_cast<const uint4*>(&sRowP[(2 * rq + k) * GT_W + 4 * g]);
    // weights of the vertical pairs: an even output row starts on a pair, an odd one in the middle of a pair
    constexpr uint32_t E0 = 18u | (34u << 16), E1 = 49u | (55u << 16), E2 = 49u | (34u << 16), E3 = 18u;
    constexpr uint32_t O0 = 18u << 16, O1 = 34u | (49u << 16), O2 = 55u | (49u << 16), O3 = 34u | (18u << 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = j >> 1;
      const bool odd = j & 1;
      const uint32_t w0 = odd ? O0 : E0, w1 = odd ? O1 : E1, w2 = odd ? O2 : E2, w3 = odd ? O3 : E3;
      uint32_t a[4];
      const uint32_t p0[4] = {P[k0].x, P[k0].y, P[k0].z, P[k0].w}, p1[4] = {P[k0 + 1].x, P[k0 + 1].y, P[k0 + 1].z, P[k0 + 1].w};
      const uint32_t p2[4] = {P[k0 + 2].x, P[k0 + 2].y, P[k0 + 2].z, P[k0 + 2].w}, p3[4] = {P[k0 + 3].x, P[k0 + 3].y, P[k0 + 3].z, P[k0 + 3].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t acc = udot2(p0[c], w0, 1u << 15);
        acc = udot2(p1[c], w1, acc);
        acc = udot2(p2[c], w2, acc);
        acc = udot2(p3[c], w3, acc);
        // unsigned shift/min on purpose: hipcc (ROCm 7.2) fuses med3(ashr(x,16),0,255) pairs into v_ashr_pk_u8_i32
        // and then assumes the upper 16 bits of its result are zero, which they are not on gfx950
        a[c] = min(acc >> 16, 255u);
      }
      outv[j] = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
    }
  }
  }   // tiles of the run
  store_tile((tx1 - 1) * GT_W);
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

// Stage ROWS patch rows as NDW ALIGNED dwords each (level pitches and offsets are multiples of 128 bytes), starting
// at the dword that holds column x0 of row y0; returns x0 & 3, the byte offset of x0 inside each staged row.
// (MUL, SH): i / NDW == (i * MUL) >> SH for i < 64 (the first round; later rounds advance row / dword / byte offset
// incrementally: 64 = (64 / NDW) rows + (64 % NDW) dwords).
template <int ROWS, int NDW, int MUL, int SH>
__device__ __forceinline__ int load_patch(uint32_t (&v)[(ROWS * NDW + 63) / 64], const uint8_t* src, int pitch, int x0, int y0, int lane)
{
  const int ax = x0 & ~3;
  const uint8_t* base = src + (size_t)y0 * pitch + ax;
  int d = 0;
  {
    const int r = (lane * MUL) >> SH;
    d = lane - r * NDW;
    base += r * pitch + 4 * d;
  }
  const int step = (64 / NDW) * pitch + 4 * (64 % NDW), wrap = pitch - 4 * NDW;
#pragma unroll
  for (int i0 = 0; i0 < ROWS * NDW; i0 += 64) {
    const int i = i0 + lane;
    v[i0 / 64] = i < ROWS * NDW ? *reinterpret_cast<const uint32_t*>(base) : 0u;
    d += 64 % NDW;
    base += step;
    if (d >= NDW) { d -= NDW; base += wrap; }
  }
  return x0 - ax;
}
template <int ROWS, int NDW>
__device__ __forceinline__ void write_patch(uint32_t* sp, const uint32_t (&v)[(ROWS * NDW + 63) / 64], int lane)
{
#pragma unroll
  for (int i0 = 0; i0 < ROWS * NDW; i0 += 64)
    if (i0 + lane < ROWS * NDW) sp[i0 + lane] = v[i0 / 64];
}
template <int ROWS, int NDW, int MUL, int SH>
__device__ __forceinline__ int stage_patch(uint32_t* sp, const uint8_t* src, int pitch, int x0, int y0, int lane)
{
  uint32_t v[(ROWS * NDW + 63) / 64];
  const int off = load_patch<ROWS, NDW, MUL, SH>(v, src, pitch, x0, y0, lane);
  write_patch<ROWS, NDW>(sp, v, lane);
  return off;
}

constexpr int OP_NDW = 9;            // orientation patch: 31 rows x 9 dwords (31 bytes + <= 3 bytes of alignment)
constexpr int BP = 37, BR = 18;      // blurred patch side / radius: |rotated pattern point| <= 18.4 -> rounds to <= 18
constexpr int BP_NDW = 10;           // 37 bytes + <= 3 bytes of alignment
constexpr int PATCH_LDS_DW = 31 * OP_NDW + BP * BP_NDW;   // both patches of one keypoint

// IC_Angle (orbextractor.cpp:25-44): integer moments of the circular patch.  Two lanes per patch row (bytes 0..15 |
// 16..31 of the row); a lane's 16 pixels are four dwords cut from the staged row with v_alignbyte, and
//   sum u * I = sum (u + 15) * I - 15 * sum I
// is two v_dot4_u32_u8 per dword against the tabulated byte weights (u + 15, and 1) of the row's circular mask.
struct IcTables { uint2 w[16][8]; };   // [|v|][dword of the 32-byte row]: x = weights u+15, y = mask, zero outside |u| <= umax(|v|)

__device__ __forceinline__ void ic_tables_init(IcTables& tb)
{
  const int t = threadIdx.x;
  if (t < 128) {
    const int av = t >> 3, k = t & 7, dmax = c_umax[av];
    uint32_t w1 = 0, w0 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = 4 * k + j - 15;
      if (u >= -dmax && u <= dmax) { w1 |= (uint32_t)(u + 15) << (8 * j); w0 |= 1u << (8 * j); }
    }
    tb.w[av][k] = make_uint2(w1, w0);
  }
}

__device__ __forceinline__ float ic_angle(const uint32_t* sp, int off, int lane, const IcTables& tb)
{
  int m10 = 0, m01 = 0;
  if (lane < 62) {
    const int row = lane >> 1, half = lane & 1;
    const int v = row - 15, av = v < 0 ? -v : v;
    const uint32_t* rp = sp + row * OP_NDW + 4 * half;
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = rp[k];
    uint32_t s1 = 0, s0 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t px = __builtin_amdgcn_alignbyte(d[k + 1], d[k], off);
      const uint2 w = tb.w[av][4 * half + k];
      s1 = __builtin_amdgcn_udot4(px, w.x, s1, false);
      s0 = __builtin_amdgcn_udot4(px, w.y, s0, false);
    }
    m10 = (int)s1 - 15 * (int)s0;
    m01 = v * (int)s0;
  }
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) { m10 += __shfl_xor(m10, o2); m01 += __shfl_xor(m01, o2); }
  return fast_atan2_deg((float)m01, (float)m10);
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

// steered BRIEF: lane l evaluates tests l, 64+l, 128+l, 192+l on the staged blurred patch (byte offset `off` in
// each row of 4 * BP_NDW bytes); the four ballots ARE the descriptor's four 64-bit words (bit i = test i).
// pat[j] = the four signed bytes (x0, y0, x1, y1) of test 64 j + lane (brief_pattern_of_lane).
__device__ __forceinline__ void brief_pattern_of_lane(int lane, uint32_t pat[4])
{
#pragma unroll
  for (int j = 0; j < 4; ++j) pat[j] = *reinterpret_cast<const uint32_t*>(&c_pattern[(j * 64 + lane) * 4]);
}
__device__ __forceinline__ void brief_words(const uint8_t* sp, int off, float angle, const uint32_t pat[4], unsigned long long words[4])
{
  float a, b;
  sincos_deg(angle, &a, &b);
  const uint8_t* c = sp + BR * (4 * BP_NDW) + BR + off;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = (int)pat[j];
    const float x0 = (float)((w << 24) >> 24), y0 = (float)((w << 16) >> 24), x1 = (float)((w << 8) >> 24), y1 = (float)(w >> 24);
    const int r0 = __float2int_rn(x0 * b + y0 * a), c0 = __float2int_rn(x0 * a - y0 * b);
    const int r1 = __float2int_rn(x1 * b + y1 * a), c1 = __float2int_rn(x1 * a - y1 * b);
    const int t0 = c[r0 * (4 * BP_NDW) + c0];
    const int t1 = c[r1 * (4 * BP_NDW) + c1];
    words[j] = __ballot(t0 < t1);
  }
}

// One wave per OB_KPW consecutive OUTPUT SLOTS (levels concatenated in order, orbextractor.cpp:722-752): orientation from the raw
// 31x31 patch, descriptor from the blurred 37x37 patch, keypoint record.  No workgroup barrier: a wave owns its LDS patches.
// The counts and the OB_KPW candidate records are fetched once per wave, the lane's four BRIEF tests once per wave; the patches
// of keypoint j + 1 are requested into registers BEFORE keypoint j is worked on (they reach LDS when j is done), and the results
// of keypoint j are stored one keypoint late, behind the next patch request (loads and stores share the vmcnt counter: a wait for
// the patches right after the stores would wait for the stores' completion as well).
// What bounds it (round 4, profiles/r04/orient_brief_bound.md): the PATCH LOADS.  Per 256 images: 0.49 ms; 0.24 with the loads
// replaced by constants; 0.50 with the loads but without BRIEF or without IC_Angle; 0.08 with none of the three.  A keypoint is
// 68 row segments of 36-40 bytes, each its own 128-byte line; the time does not depend on the batch (64 images = 196 MB, inside
// the 256 MB MALL: the same per image) nor on which XCD's L2 an image lands in -- it is the L2 -> L1 line traffic of ~11 KB per
// keypoint for 2.3 KB of pixels.  Variants measured and not kept: 8 keypoints per wave with the slot -> level search on vector
// lanes, per-lane IC_Angle weights in registers instead of the LDS table, one 32-bit offset per load (330 -> 60 scalar
// instructions per keypoint, but 74 VGPRs = 6 waves per SIMD): 0.487 ms; the same forced to 64 VGPRs (40 B of scratch): 0.84.
constexpr int OB_KPW = 4;
__global__ __launch_bounds__(256) void k_orient_brief(OrbDev o)
{
  __shared__ uint32_t sPatch[4][PATCH_LDS_DW];
  __shared__ IcTables sIc;
  ic_tables_init(sIc);
  __syncthreads();                                      // the only workgroup barrier; waves leave only after it
  const int img = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
  const int slot0 = (blockIdx.x * 4 + wave) * OB_KPW;
  // levels of the output slots: all per-level counts are fetched at once (independent scalar loads) -- a search loop
  // that loads one count per iteration costs one dependent memory round trip per level
  int cnt[MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < MAX_LEVELS; ++l) cnt[l] = l < o.nlevels ? o.sel_count[img * o.nlevels + l] : 0;
  uint32_t pat[4];
  brief_pattern_of_lane(lane, pat);
  int lvl_of[OB_KPW];
  uint32_t cand[OB_KPW];
  bool ok[OB_KPW];
#pragma unroll
  for (int j = 0; j < OB_KPW; ++j) {
    const int slot = slot0 + j;
    int level = 0, base = 0, n = cnt[0];
#pragma unroll
    for (int l = 1; l < MAX_LEVELS; ++l) {
      const bool next = l < o.nlevels && level == l - 1 && slot >= base + n;
      base = next ? base + n : base;
      level = next ? l : level;
      n = next ? cnt[l] : n;
    }
    const int k = slot - base;
    ok[j] = k < n && slot < o.out_cap;                  // (wave-uniform; false from the first slot past the last level's keypoints on)
    if (k < n && slot >= o.out_cap && lane == 0) atomicOr(&o.status[img], 4);   // a keypoint past the output capacity
    lvl_of[j] = level;
    cand[j] = ok[j] ? o.sel[(size_t)(img * o.nlevels + level) * SEL_CAP + k] : 0u;
  }
  uint32_t* sp = sPatch[wave];
  const int minB = EDGE_THRESHOLD - 3;
  uint32_t vr[(31 * OP_NDW + 63) / 64], vb[(BP * BP_NDW + 63) / 64];
  int off_o = 0, off_b = 0;
  auto request = [&](int j) {
    const uint32_t p = cand[j];
    const int cx = (int)(p & 0xFFF) + minB, cy = (int)((p >> 12) & 0xFFF) + minB;   // cvRound of integral coords
    const size_t lvl = (size_t)img * o.pyr_bytes + o.lvl_off[lvl_of[j]];
    const int pitch = o.lvl_pitch[lvl_of[j]];
    off_o = load_patch<31, OP_NDW, 57, 9>(vr, o.pyr + lvl, pitch, cx - 15, cy - 15, lane);
    off_b = load_patch<BP, BP_NDW, 205, 11>(vb, o.blur + lvl, pitch, cx - BR, cy - BR, lane);
  };
  unsigned long long word_prev = 0;                     // lane < 4: word `lane` of the previous keypoint's descriptor
  float angle_prev = 0.f;
  auto store = [&](int j) {                             // results of slot j (held in word_prev / angle_prev)
    const int slot = slot0 + j, level = lvl_of[j];
    const uint32_t p = cand[j];
    if (lane < 4) reinterpret_cast<unsigned long long*>(o.out_desc + ((size_t)img * o.out_cap + slot) * 32)[lane] = word_prev;
    if (lane == 0) {
      ssx_keypoint kp;
      const float sc = o.scale[level];
      kp.x = (float)((int)(p & 0xFFF) + minB); kp.y = (float)((int)((p >> 12) & 0xFFF) + minB);
      if (level != 0) { kp.x *= sc; kp.y *= sc; }
      kp.size = (float)(int)(31 * sc);
      kp.angle = angle_prev;
      kp.response = (float)(p >> 24);
      kp.octave = level;
      kp.class_id = -1;
      reinterpret_cast<ssx_keypoint*>(o.out_kps)[(size_t)img * o.out_cap + slot] = kp;
    }
  };
  if (!ok[0]) return;
  request(0);
#pragma unroll
  for (int j = 0; j < OB_KPW; ++j) {
    if (!ok[j]) break;
    const int oo = off_o, ob = off_b;
    write_patch<31, OP_NDW>(sp, vr, lane);
    write_patch<BP, BP_NDW>(sp + 31 * OP_NDW, vb, lane);
    if (j + 1 < OB_KPW && ok[j + 1]) request(j + 1);
    if (j > 0) store(j - 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float angle = ic_angle(sp, oo, lane, sIc);
    unsigned long long words[4];
    brief_words(reinterpret_cast<const uint8_t*>(sp + 31 * OP_NDW), ob, angle, pat, words);
    word_prev = (lane & 2) ? ((lane & 1) ? words[3] : words[2]) : ((lane & 1) ? words[1] : words[0]);
    angle_prev = angle;
    __builtin_amdgcn_wave_barrier();                    // every lane has read the patches before the next keypoint's overwrite them
    if (j + 1 == OB_KPW || !ok[j + 1]) store(j);
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
  __shared__ uint32_t sPatch[4][PATCH_LDS_DW];
  __shared__ IcTables sIc;
  ic_tables_init(sIc);                                  // published by the __syncthreads() after the patch staging
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR: everything derived from it is scalar
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
  uint32_t* sp = sPatch[wave];
  // orientation patch (raw level)
  int off_o = 0;
  if (active) off_o = stage_patch<31, OP_NDW, 57, 9>(sp, lvl, pitch, cx - 15, cy - 15, lane);
  __syncthreads();
  float angle = 0.f;
  if (active) angle = ic_angle(sp, off_o, lane, sIc);
  // CalcDescriptors: pt (already multiplied back by scale) is divided by scale AGAIN before describing
  float ox = kp.x * sc, oy = kp.y * sc;                        // kps.pt *= scale  (the output coordinates)
  const float dx = ox / sc, dy = oy / sc;
  const int bx = __float2int_rn(dx), by = __float2int_rn(dy);
  const uint8_t* blur = o.blur + o.lvl_off[level];
  int off_b = 0;
  if (active) off_b = stage_patch<BP, BP_NDW, 205, 11>(sp + 31 * OP_NDW, blur, pitch, bx - BR, by - BR, lane);
  __syncthreads();
  if (!active) { if (k < a.n_in && lane == 0) a.keep[k] = 0; return; }
  unsigned long long words[4];
  uint32_t pat[4];
  brief_pattern_of_lane(lane, pat);
  brief_words(reinterpret_cast<const uint8_t*>(sp + 31 * OP_NDW), off_b, angle, pat, words);
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
  w->arena.release(); w->input.release(); w->stereo.release(); w->stage.release(); w->fetch.release();
  w->ingest[0].release(); w->ingest[1].release(); w->counts_pinned.release();
  for (int b = 0; b < 2; ++b) { if (w->ev_up[b]) (void)hipEventDestroy(w->ev_up[b]); if (w->ev_free[b]) (void)hipEventDestroy(w->ev_free[b]); if (w->ev_counts[b]) (void)hipEventDestroy(w->ev_counts[b]); }
  if (w->copy_stream) { (void)hipStreamSynchronize(w->copy_stream); (void)hipStreamDestroy(w->copy_stream); }
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
    // a level yields at most max(N + 3, 4 * nIni) keypoints: the quadtree stops within 3 nodes of its budget N, but
    // its very first subdivision already makes up to 4 * nIni nodes (nIni = aspect ratio of the level) however
    // small N is (DistributeOctTree, orbextractor.cpp:347-349, 394-470)
    {
      const int bw = d.lvl_cols[l] - 2 * (EDGE_THRESHOLD - 3), bh = d.lvl_rows[l] - 2 * (EDGE_THRESHOLD - 3);
      const int n_ini = bh > 0 ? (int)std::lround((double)bw / (double)bh) : 0;
      out_cap += std::max(d.feat[l] + 4, 4 * std::max(n_ini, 0) + 4);
    }
    if (d.feat[l] + 8 > SEL_CAP) {
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
    // octree node-table capacity: a round of phase 1 never ends above N nodes (it is only entered while
    // size + 3 * expandable <= N) except the first one (<= 4 * nIni <= 256), phase 2 stops within N + 2
    int maxN = 256;
    for (int l = 0; l < (detect_only ? 1 : nlevels); ++l) maxN = std::max(maxN, d.feat[l]);
    d.oct_ln = (maxN + 8 + 7) & ~7;
    d.oct_max_cells = 1;
    for (int l = 0; l < nlevels; ++l) d.oct_max_cells = std::max(d.oct_max_cells, d.lvl_cell0[l + 1] - d.lvl_cell0[l]);
    // per-key state: in LDS for images up to ~0.6 Mpx (16384 candidates per level), else in global scratch
    d.oct_global_keys = (size_t)d.lvl_rows[0] * d.lvl_cols[0] > 600000;
    d.oct_cand_cap = d.oct_global_keys ? CAND_CAP_BIG : CAND_CAP;
    const size_t key_lds = d.oct_global_keys ? 0 : 6 * (size_t)CAND_CAP;
    if (4 * (size_t)(d.oct_max_cells + 8) + key_lds > (size_t)OCT_LDS_BUDGET) {
      ctx->set_error("ssx_orb: %d grid cells on one level exceed the octree workgroup's LDS", d.oct_max_cells);
      return SSX_ERR_UNSUPPORTED;
    }
    d.oct_global_tab = (size_t)OCT_NODE_BYTES * d.oct_ln + 4 * (size_t)(d.oct_max_cells + 8) + key_lds > (size_t)OCT_LDS_BUDGET;
    d.oct_stride = ((d.oct_global_keys ? 6 * (size_t)CAND_CAP_BIG : 0) + (d.oct_global_tab ? (size_t)OCT_NODE_BYTES * d.oct_ln : 0) + 255) & ~size_t(255);
    if (!d.oct_global_keys && !d.oct_global_tab) d.oct_stride = 0;
  }
  {
    int tile = 16, npx = 1, t0 = 0;
    for (const Cell& c : cells) {
      const int tw = ((c.x0 & 3) + c.w + 3) & ~3;
      tile = std::max(tile, tw * (int)c.h);
      npx = std::max(npx, std::max(c.w - 6, 0) * std::max(c.h - 6, 0));
    }
    d.fast_tile_bytes = (tile + 15) & ~15;
    d.fast_lds_per_wave = 2 * d.fast_tile_bytes + ((2 * npx + 15) & ~15);
    d.gauss_run = I > 8 ? GT_RUN : 1;
    for (int l = 0; l < nlevels; ++l) {
      d.gauss_tile0[l] = t0;
      t0 += ((((d.lvl_cols[l] + GT_W - 1) / GT_W) + d.gauss_run - 1) / d.gauss_run) * ((d.lvl_rows[l] + GT_H - 1) / GT_H);   // runs of tiles
    }
    for (int l = nlevels; l <= MAX_LEVELS; ++l) d.gauss_tile0[l] = t0;
  }
  // cv::resize INTER_LINEAR tables (resize.cpp: inv_scale = dsize/ssize, scale = 1/inv_scale; fx = (dx+0.5)*scale-0.5
  // in float, sx = floor, clamps, cvRound of the weights * 2048): one (index, weight) pair row per destination
  // column / row of every level.  Column tables are padded to a multiple of 4 with zero rows.
  std::vector<ResizeQuad> xtab;
  std::vector<uint2> ytab;
  for (int l = 1; l < nlevels; ++l) {
    const int scols = d.lvl_cols[l - 1], srows = d.lvl_rows[l - 1], dcols = d.lvl_cols[l], drows = d.lvl_rows[l];
    const double scale_x = 1. / ((double)dcols / scols), scale_y = 1. / ((double)drows / srows);
    d.rs_xoff[l] = (int)xtab.size();
    d.rs_yoff[l] = (int)ytab.size();
    d.rs_wide8[l] = 1;
    for (int q = 0; 4 * q < dcols; ++q) {
      ResizeQuad rq{};
      for (int k = 0; k < 4 && 4 * q + k < dcols; ++k) {
        const int dx = 4 * q + k;
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= scols - 1) { fx = 0; sx = scols - 1; }       // dx >= xmax: the right neighbour gets weight 0
        const int a0 = h_round((1.f - fx) * 2048), a1 = h_round(fx * 2048);
        const int sx1 = std::min(sx + 1, scols - 1);
        if (k == 0) rq.s0 = (uint32_t)sx;
        const int o0 = sx - (int)rq.s0, o1 = sx1 - (int)rq.s0;
        if (o1 > 7) d.rs_wide8[l] = 0;
        if (o1 > 255) {
          ctx->set_error("ssx_orb: pyramid scale factor %g is too large for the resize tables", (double)prm.scale_factor);
          return SSX_ERR_UNSUPPORTED;
        }
        rq.sel0 |= (uint32_t)o0 << (8 * k);
        rq.sel1 |= (uint32_t)o1 << (8 * k);
        rq.w[k] = (uint32_t)a0 | ((uint32_t)a1 << 16);
      }
      xtab.push_back(rq);
    }
    for (int dy = 0; dy < drows; ++dy) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      const int sy = (int)std::floor(fy);
      fy -= sy;
      const int b0 = h_round((1.f - fy) * 2048), b1 = h_round(fy * 2048);
      const int sy0 = std::min(std::max(sy, 0), srows - 1), sy1 = std::min(std::max(sy + 1, 0), srows - 1);
      ytab.push_back(make_uint2((uint32_t)sy0 | ((uint32_t)sy1 << 16), (uint32_t)b0 | ((uint32_t)b1 << 16)));
    }
  }
  for (const Cell& c : cells)
    if (c.w > ROI_MAX || c.h > ROI_MAX) {
      ctx->set_error("ssx_orb: grid cell %dx%d exceeds the %d-px LDS tile", c.w, c.h, ROI_MAX);
      return SSX_ERR_UNSUPPORTED;
    }
  Layout lay;
  const size_t o_cells = lay.take(sizeof(Cell) * std::max<size_t>(cells.size(), 1));
  const size_t o_xtab = lay.take(sizeof(ResizeQuad) * std::max<size_t>(xtab.size(), 1));
  const size_t o_ytab = lay.take(sizeof(uint2) * std::max<size_t>(ytab.size(), 4));
  const size_t o_pyr = lay.take(d.pyr_bytes * I);
  const size_t o_mask = lay.take(has_mask ? d.pyr_bytes * I : 256);
  const size_t o_blur = lay.take(d.pyr_bytes * I);
  const size_t o_ccount = lay.take(sizeof(int) * (size_t)I * std::max(d.n_cells, 1));
  const size_t o_ccand = lay.take(sizeof(uint32_t) * (size_t)I * std::max(d.n_cells, 1) * CELL_CAP);
  const size_t o_oct = lay.take(std::max<size_t>(d.oct_stride * (size_t)I * nlevels, 256));
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
  d.rs_xtab = (const ResizeQuad*)(base + o_xtab);
  d.rs_ytab = (const uint2*)(base + o_ytab);
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
  if (!xtab.empty()) {
    SSX_HIP_TRY(ctx, hipMemcpyAsync(base + o_xtab, xtab.data(), sizeof(ResizeQuad) * xtab.size(), hipMemcpyHostToDevice, ctx->stream));
    SSX_HIP_TRY(ctx, hipMemcpyAsync(base + o_ytab, ytab.data(), sizeof(uint2) * ytab.size(), hipMemcpyHostToDevice, ctx->stream));
  }
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // `cells` and the tables are pageable temporaries
  ws->dev = d;
  ws->rows = rows; ws->cols = cols; ws->I = I; ws->nlevels = nlevels; ws->nfeatures = prm.nfeatures;
  ws->ini_th = prm.ini_th_fast; ws->min_th = prm.min_th_fast; ws->has_mask = has_mask; ws->detect_only = detect_only;
  ws->scale_factor = prm.scale_factor;
  ws->planned = true;
  return SSX_OK;
}

// pyramid (ComputePyramid, orbextractor.cpp:993-1027): level l from level l-1, images (and masks) of the batch
static void launch_pyramid(ssx_ctx* ctx, const OrbDev& d, hipStream_t s, int images)
{
  for (int l = 1; l < d.nlevels; ++l) {
    if (d.lvl_cols[l] <= 0 || d.lvl_rows[l] <= 0) break;     // a tiny image runs out of pixels before it runs out of levels: nothing there
    const int loops = images > 8 ? RS_LOOP : 1;
    const dim3 grid((d.lvl_cols[l] + 255) / 256, (d.lvl_rows[l] + 4 * RS_ROWS * loops - 1) / (4 * RS_ROWS * loops), images);
    auto kern = d.rs_wide8[l] ? k_resize<true> : k_resize<false>;
    for (int m = 0; m < (d.has_mask ? 2 : 1); ++m) {
      uint8_t* pyr = m ? d.maskpyr : d.pyr;
      SSX_PROF(ctx, KID_ORB_RESIZE, hipLaunchKernelGGL(kern, grid, dim3(64, 4), 0, s, pyr + d.lvl_off[l - 1], pyr + d.lvl_off[l], d.pyr_bytes,
                         d.lvl_pitch[l - 1], d.lvl_rows[l], d.lvl_cols[l], d.lvl_pitch[l], d.rs_xtab + d.rs_xoff[l], d.rs_ytab + d.rs_yoff[l], loops));
    }
  }
}

ssx_status stage_level0(ssx_ctx* ctx, const uint8_t* imgs_dev, int stride, size_t img_bytes, const uint8_t* masks_dev,
                        int mask_stride, size_t mask_bytes)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  const dim3 grid((d.lvl_cols[0] + 511) / 512, (d.lvl_rows[0] + 3) / 4, d.I);
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0, grid, dim3(64, 4), 0, ctx->stream, imgs_dev, stride, img_bytes, d.pyr, d.pyr_bytes,
                     d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  if (d.has_mask)
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0, grid, dim3(64, 4), 0, ctx->stream, masks_dev, mask_stride, mask_bytes, d.maskpyr,
                       d.pyr_bytes, d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  SSX_HIP_TRY(ctx, hipGetLastError());
  return SSX_OK;
}

ssx_status run_pipeline(ssx_ctx* ctx)
{
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  hipStream_t s = ctx->stream;
  SSX_HIP_TRY(ctx, hipMemsetAsync(d.status, 0, sizeof(int) * d.I, s));
  // cells (waves) per workgroup.  A workgroup keeps its LDS until its slowest cell is done and cells differ 8x in work:
  // the kernel alone takes 0.615 / 0.430 / 0.410 ms per 128 images with 4 / 2 / 1 cells per workgroup.  The front-end as a
  // whole does not care (1.16 / 1.19 / 1.20 ms: its two streams hide the difference), the composite step with the BA on
  // its own streams does a little: 17.96 / 18.36 / 18.20 k frames/s.
  constexpr int FAST_WPW = 2;
  if (FAST_WPW * (size_t)d.fast_lds_per_wave > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_cells), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(FAST_WPW * (size_t)d.fast_lds_per_wave));
  auto launch_fast = [&](hipStream_t st, int c0, int c1) {
    if (c1 > c0)
      SSX_PROF_ON(ctx, st, KID_ORB_FAST, hipLaunchKernelGGL(k_fast_cells, dim3((c1 - c0 + FAST_WPW - 1) / FAST_WPW, d.I), dim3(64 * FAST_WPW),
                                                            FAST_WPW * (size_t)d.fast_lds_per_wave, st, d, c0, c1));
  };
  // Two streams.  The pyramid is a chain of seven dependent, mostly small launches that leaves the chip idle, so
  // level 0 of the detection (a third of the cells) runs beside it on the auxiliary stream; the blur only needs
  // the pyramid and runs beside the detection of the upper levels and the octree.  Joined before the octree
  // (level-0 candidates) and before the descriptors (blur).
  // (One or two images: the chain is latency bound either way and the event hand-offs cost more than the overlap
  // returns -- 0.319 against 0.312 ms for a single stereo pair -- so a single frame stays on one stream.)
  static const bool no_fork_env = getenv("SSX_ORB_NO_FORK") != nullptr;   // (tools: every kernel alone on one stream, for per-kernel times)
  const bool fork = !d.detect_only && ctx->aux != nullptr && d.I > 2 && !no_fork_env;
  if (fork) {
    SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
    SSX_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    launch_fast(ctx->aux, 0, d.lvl_cell0[1]);
    SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_fast0, ctx->aux));
    launch_pyramid(ctx, d, s, d.I);
    SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_pyr, s));
    SSX_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_pyr, 0));
    SSX_PROF_ON(ctx, ctx->aux, KID_ORB_GAUSS, hipLaunchKernelGGL(k_gauss7, dim3(d.gauss_tile0[d.nlevels], d.I), dim3(256), 0, ctx->aux, d));
    SSX_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    launch_fast(s, d.lvl_cell0[1], d.n_cells);
    SSX_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_fast0, 0));
  } else {
    launch_pyramid(ctx, d, s, d.I);
    if (!d.detect_only)
      SSX_PROF(ctx, KID_ORB_GAUSS, hipLaunchKernelGGL(k_gauss7, dim3(d.gauss_tile0[d.nlevels], d.I), dim3(256), 0, s, d));
    launch_fast(s, 0, d.n_cells);
  }
  SSX_PROF(ctx, KID_ORB_OCTREE, launch_octree(d, s));
  if (d.detect_only) {
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_finalize_detect, dim3((SEL_CAP + 255) / 256, d.I), dim3(256), 0, s, d));
  } else {
    // one wave per OB_KPW output slots; a keypoint past out_cap raises the capacity flag from the wave that holds its slot
    const dim3 kgrid((d.out_cap + 4 + 4 * OB_KPW - 1) / (4 * OB_KPW), d.I);
    if (fork) SSX_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    SSX_PROF(ctx, KID_ORB_BRIEF, hipLaunchKernelGGL(k_orient_brief, kgrid, dim3(256), 0, s, d));
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
  if (stride < cols || (mask && mask_stride < cols) || cap < 0 || (cap > 0 && !kps_out)) {
    ctx->set_error("ssx_orb_detect: stride smaller than the image width, negative capacity or no output array");
    return SSX_ERR_INVALID_ARG;
  }
  ssx_status st = run_host_image(ctx, img, stride, rows, cols, mask, mask_stride, *prm, true);
  if (st != SSX_OK) return st;
  return fetch_image_result(ctx, 0, cap, kps_out, nullptr, n);
}

// ORBextractor::Detect with the mask of FrontEnd::DetectFeatures given as its RECTANGLES (frontend.cpp:302-312: 255 everywhere,
// cv::rectangle(mask, pt - (10, 10), pt + (10, 10), 0, FILLED) per tracked feature): boxes_xyxy = n_boxes x (x0, y0, x1, y1), corners
// inclusive.  The mask is built on the device (k_mask_boxes); results equal ssx_orb_detect on the rasterised mask bit for bit.
ssx_status ssx_orb_detect_boxes(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols, const int32_t* boxes_xyxy,
                                int32_t n_boxes, const ssx_orb_params* prm, int32_t cap, ssx_keypoint* kps_out, int32_t* n)
{
  if (!ctx || !prm || !n) return SSX_ERR_INVALID_ARG;
  *n = 0;
  if (!img || rows <= 0 || cols <= 0) return SSX_OK;   // `if (_image.empty()) return;` orbextractor.cpp:758
  if (stride < cols || cap < 0 || (cap > 0 && !kps_out) || n_boxes < 0 || (n_boxes > 0 && !boxes_xyxy)) {
    ctx->set_error("ssx_orb_detect_boxes: stride smaller than the image width, negative capacity / box count or a missing array");
    return SSX_ERR_INVALID_ARG;
  }
  ssx_status st = plan(ctx, rows, cols, 1, *prm, true, true);
  if (st != SSX_OK) return st;
  OrbWorkspace* ws = get_ws(ctx);
  const OrbDev& d = ws->dev;
  const size_t bytes = (size_t)rows * cols, box_off = (bytes + 15) & ~size_t(15), box_bytes = sizeof(int32_t) * 4 * (size_t)n_boxes;
  SSX_HIP_TRY(ctx, ws->input.reserve(box_off + box_bytes + 512));
  SSX_HIP_TRY(ctx, ws->stage.reserve(box_off + box_bytes + 512));
  uint8_t* hs = ws->stage.as<uint8_t>();
  for (int y = 0; y < rows; ++y) memcpy(hs + (size_t)y * cols, img + (size_t)y * stride, cols);
  if (n_boxes) memcpy(hs + box_off, boxes_xyxy, box_bytes);
  SSX_HIP_TRY(ctx, hipMemcpyAsync(ws->input.p, hs, box_off + box_bytes, hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((d.lvl_cols[0] + 511) / 512, (d.lvl_rows[0] + 3) / 4, 1);
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0, grid, dim3(64, 4), 0, ctx->stream, ws->input.as<uint8_t>(), cols, bytes, d.pyr, d.pyr_bytes,
                     d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  SSX_HIP_TRY(ctx, hipMemsetAsync(d.maskpyr + d.lvl_off[0], 255, (size_t)d.lvl_pitch[0] * d.lvl_rows[0], ctx->stream));
  if (n_boxes)
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_mask_boxes, dim3(n_boxes), dim3(64), 0, ctx->stream, reinterpret_cast<const int4*>(ws->input.as<uint8_t>() + box_off),
                       n_boxes, d.maskpyr + d.lvl_off[0], d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  SSX_HIP_TRY(ctx, hipGetLastError());
  st = run_pipeline(ctx);
  if (st != SSX_OK) return st;
  return fetch_image_result(ctx, 0, cap, kps_out, nullptr, n);
}

// ssx_orb_detect_boxes for n images in ONE call (one keyframe of each of n streams: FrontEnd::DetectFeatures, frontend.cpp:302-344):
// every kernel of the detection once for all images.  Per image the bits of ssx_orb_detect_boxes.  The workspace is planned for
// the largest batch seen (a power of two) and run with the images of the call.
ssx_status ssx_orb_detect_boxes_batch(ssx_ctx* ctx, int32_t n, const ssx_orb_detect_job* jobs, int32_t rows, int32_t cols, const ssx_orb_params* prm,
                                      int32_t images_on_device)
{
  if (!ctx || !prm || n < 0 || (n > 0 && !jobs)) return SSX_ERR_INVALID_ARG;
  if (n == 0) return SSX_OK;
  if (rows <= 0 || cols <= 0) return SSX_ERR_INVALID_ARG;
  size_t n_boxes = 0;
  for (int j = 0; j < n; ++j) {
    const ssx_orb_detect_job& q = jobs[j];
    if (!q.n_out || !q.img || q.stride < cols || q.cap < 0 || (q.cap > 0 && !q.kps_out) || q.n_boxes < 0 || (q.n_boxes > 0 && !q.boxes_xyxy)) {
      ctx->set_error("ssx_orb_detect_boxes_batch: job %d: missing image / output, stride smaller than the width, or a negative count", j);
      return SSX_ERR_INVALID_ARG;
    }
    if (q.stride != jobs[0].stride) { ctx->set_error("ssx_orb_detect_boxes_batch: the images of a call share one stride"); return SSX_ERR_INVALID_ARG; }
    *q.n_out = 0;
    n_boxes += (size_t)q.n_boxes;
  }
  OrbWorkspace* ws = get_ws(ctx);
  int cap_I = 1;
  while (cap_I < n) cap_I *= 2;
  if (ws->planned && ws->rows == rows && ws->cols == cols && ws->detect_only && ws->has_mask && ws->nfeatures == prm->nfeatures && ws->ini_th == prm->ini_th_fast &&
      ws->min_th == prm->min_th_fast && ws->I > cap_I)
    cap_I = ws->I;                                                   // (a smaller batch runs on the larger plan)
  ssx_status st = plan(ctx, rows, cols, cap_I, *prm, true, true);
  if (st != SSX_OK) return st;
  struct RestoreI { OrbWorkspace* w; int I; ~RestoreI() { w->dev.I = I; } } restore{ws, ws->dev.I};
  ws->dev.I = n;
  const OrbDev& d = ws->dev;
  hipStream_t s = ctx->stream;
  // one pinned block: [image pointers | (staged images) | boxes | box -> image] -> one copy
  const size_t bytes = (size_t)rows * cols;
  Layout lay;
  const size_t o_ptr = lay.take(sizeof(void*) * (size_t)n);
  const size_t o_box = lay.take(sizeof(int32_t) * 4 * std::max<size_t>(n_boxes, 1));
  const size_t o_bimg = lay.take(sizeof(int) * std::max<size_t>(n_boxes, 1));
  const size_t o_img = images_on_device ? 0 : lay.take(bytes * (size_t)n + 16);
  SSX_HIP_TRY(ctx, ws->input.reserve(lay.off + 512, 1.5));
  SSX_HIP_TRY(ctx, ws->stage.reserve(lay.off + 512, 1.5));
  char* hs = ws->stage.as<char>();
  char* db = ws->input.as<char>();
  const uint8_t** ptrs = reinterpret_cast<const uint8_t**>(hs + o_ptr);
  int32_t* hbox = reinterpret_cast<int32_t*>(hs + o_box);
  int* hbimg = reinterpret_cast<int*>(hs + o_bimg);
  size_t b0 = 0;
  int in_stride = jobs[0].stride;
  for (int j = 0; j < n; ++j) {
    const ssx_orb_detect_job& q = jobs[j];
    if (images_on_device) ptrs[j] = q.img;
    else {
      uint8_t* dst = reinterpret_cast<uint8_t*>(hs + o_img) + bytes * (size_t)j;
      for (int y = 0; y < rows; ++y) memcpy(dst + (size_t)y * cols, q.img + (size_t)y * q.stride, cols);
      ptrs[j] = reinterpret_cast<const uint8_t*>(db + o_img) + bytes * (size_t)j;
    }
    if (q.n_boxes) memcpy(hbox + 4 * b0, q.boxes_xyxy, sizeof(int32_t) * 4 * (size_t)q.n_boxes);
    for (int b = 0; b < q.n_boxes; ++b) hbimg[b0 + b] = j;
    b0 += (size_t)q.n_boxes;
  }
  if (!images_on_device) in_stride = cols;
  SSX_HIP_TRY(ctx, hipMemcpyAsync(db, hs, lay.off, hipMemcpyHostToDevice, s));
  const dim3 grid((d.lvl_cols[0] + 511) / 512, (d.lvl_rows[0] + 3) / 4, n);
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_copy_level0_ptrs, grid, dim3(64, 4), 0, s, reinterpret_cast<const uint8_t* const*>(db + o_ptr), in_stride, d.pyr,
                                                 d.pyr_bytes, d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  const size_t lvl0_bytes = (size_t)d.lvl_pitch[0] * d.lvl_rows[0];   // (pitch: a multiple of 128)
  SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_fill_level0_b, dim3(64, n), dim3(256), 0, s, d.maskpyr + d.lvl_off[0], d.pyr_bytes, lvl0_bytes, 0xFFFFFFFFu));
  if (n_boxes)
    SSX_PROF(ctx, KID_ORB_MISC, hipLaunchKernelGGL(k_mask_boxes_b, dim3((unsigned)n_boxes), dim3(64), 0, s, reinterpret_cast<const int4*>(db + o_box),
                                                   reinterpret_cast<const int*>(db + o_bimg), d.maskpyr + d.lvl_off[0], d.pyr_bytes, d.lvl_rows[0], d.lvl_cols[0], d.lvl_pitch[0]));
  SSX_HIP_TRY(ctx, hipGetLastError());
  st = run_pipeline(ctx);
  if (st != SSX_OK) return st;
  // results: counts + status words + the keypoint block of the n images, one synchronisation
  Layout out;
  const size_t r_n = out.take(sizeof(int) * (size_t)n), r_st = out.take(sizeof(int) * (size_t)n), r_k = out.take(sizeof(ssx_keypoint) * (size_t)n * d.out_cap);
  SSX_HIP_TRY(ctx, ws->fetch.reserve(out.off, 1.5));
  char* hf = ws->fetch.as<char>();
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hf + r_n, d.out_n, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hf + r_st, d.status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hf + r_k, d.out_kps, sizeof(ssx_keypoint) * (size_t)n * d.out_cap, hipMemcpyDeviceToHost, s));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(s));
  const int* hn = reinterpret_cast<const int*>(hf + r_n);
  const int* hst = reinterpret_cast<const int*>(hf + r_st);
  ssx_status first = SSX_OK;
  for (int j = 0; j < n; ++j) {
    const ssx_orb_detect_job& q = jobs[j];
    if (hst[j] != 0) {
      ctx->set_error("ssx_orb: internal capacity exceeded (image %d of the batch, status bits %d: 1=candidates 2=octree nodes 4=outputs)", j, hst[j]);
      if (first == SSX_OK) first = SSX_ERR_CAPACITY;
      continue;
    }
    *q.n_out = hn[j];
    if (hn[j] > q.cap) {
      ctx->set_error("ssx_orb: %d keypoints but capacity %d (image %d of the batch)", hn[j], q.cap, j);
      if (first == SSX_OK) first = SSX_ERR_CAPACITY;
      continue;
    }
    if (hn[j] > 0) memcpy(q.kps_out, hf + r_k + sizeof(ssx_keypoint) * (size_t)j * d.out_cap, sizeof(ssx_keypoint) * (size_t)hn[j]);
  }
  return first;
}

ssx_status ssx_orb_extract(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                           const uint8_t* mask, int32_t mask_stride, const ssx_orb_params* prm, int32_t cap,
                           ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n)
{
  if (!ctx || !prm || !n) return SSX_ERR_INVALID_ARG;
  *n = 0;
  if (!img || rows <= 0 || cols <= 0) return SSX_OK;   // orbextractor.cpp:691
  if (stride < cols || (mask && mask_stride < cols) || cap < 0 || (cap > 0 && (!kps_out || !desc_out))) {
    ctx->set_error("ssx_orb_extract: stride smaller than the image width, negative capacity or no output arrays");
    return SSX_ERR_INVALID_ARG;
  }
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
  if (stride < cols || !kps_out || !desc_out) {
    ctx->set_error("ssx_orb_describe_at: stride smaller than the image width or no output arrays");
    return SSX_ERR_INVALID_ARG;
  }
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
  launch_pyramid(ctx, d, s, 1);           // ComputePyramid(image), orbextractor.cpp:1012-1027
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

#ifndef SSX_NO_TEST_HOOKS   // kernel taps of the parity tests (include/ssx_test_hooks.h)
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

#endif  // SSX_NO_TEST_HOOKS

}  // extern "C"
