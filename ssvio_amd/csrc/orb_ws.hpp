// ssvio_amd/csrc/orb_ws.hpp -- device-side view and per-ctx workspace of the ORB / stereo front-end.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "ctx.hpp"

namespace ssxorb {

constexpr int MAX_LEVELS = 8;
constexpr int CELL_CAP = 256;      // candidates kept per grid cell (31x32 interior: NMS leaves < 250)
constexpr int CAND_CAP = 16384;    // candidates per (image, level) the octree keeps in LDS (16 per thread)
constexpr int CAND_CAP_BIG = 65536;  // ... and in global scratch for larger images (64 per thread)
constexpr int SEL_CAP = 4096;      // selected keypoints per (image, level)
constexpr int EDGE_THRESHOLD = 19; // orbextractor.cpp:13
constexpr int OCT_THREADS = 1024;

struct Cell {
  int16_t x0, y0;   // ROI origin in the level image
  int16_t w, h;     // ROI size (cell + 6, clipped)
  int16_t ox, oy;   // j*wCell, i*hCell added to ROI-local coordinates (orbextractor.cpp:816-817)
  int16_t level, pad;
};

// octree (octree.hip): OCT_NODE_BYTES per node-table entry; the tables live in LDS, or in a per (image, level) global
// scratch block when the per-level budget is too large for LDS (OrbDev::oct_global_tab).
constexpr int OCT_NODE_BYTES = 50;
constexpr int OCT_LDS_BUDGET = 158 * 1024;   // dynamic LDS the octree workgroup may ask for (160 KB per CU, ~1 KB static)

struct ResizeQuad;

struct OrbDev {
  // geometry
  int I;                         // images in the batch
  int nlevels;
  int lvl_rows[MAX_LEVELS], lvl_cols[MAX_LEVELS], lvl_pitch[MAX_LEVELS];
  size_t lvl_off[MAX_LEVELS];    // byte offset of a level inside one image's pyramid
  size_t pyr_bytes;              // bytes of one image's pyramid
  float scale[MAX_LEVELS];       // mvScaleFactor
  int feat[MAX_LEVELS];          // mnFeaturesPerLevel (or nfeatures for the single-level Detect)
  int lvl_cell0[MAX_LEVELS + 1]; // first cell of each level
  int n_cells;
  int ini_th, min_th;
  int has_mask;
  int detect_only;               // ORBextractor::Detect: level 0 only, no orientation / descriptors
  int out_cap;                   // keypoints per image in the output arrays
  int fast_tile_bytes;           // LDS bytes of one ROI tile (max over cells, pitch rounded to 4)
  int fast_lds_per_wave;         // image tile + score tile + compaction list
  int gauss_tile0[MAX_LEVELS + 1]; // first blur tile RUN of each level (one launch covers all levels)
  int gauss_run;                 // tiles per workgroup of k_gauss7: 4 for a batch (the next tile's loads fly during a tile's passes), 1 for a
                                 // frame or two (the chip is empty: every tile its own workgroup is the shorter chain)
  int rs_xoff[MAX_LEVELS], rs_yoff[MAX_LEVELS];   // first row of level l in the cv::resize tables below
  int rs_wide8[MAX_LEVELS];      // every quad of the level reads <= 8 consecutive source bytes
  // buffers
  const Cell* cells;
  const ResizeQuad* rs_xtab;     // per quad of destination columns (levels concatenated), see k_resize
  const uint2* rs_ytab;          // per destination row:    sy0 | sy1<<16, b0 | b1<<16
  uint8_t* pyr;                  // [I][pyr_bytes]
  uint8_t* maskpyr;              // [I][pyr_bytes] (only when has_mask)
  uint8_t* blur;                 // [I][pyr_bytes]
  int* cell_count;               // [I][n_cells]
  uint32_t* cell_cand;           // [I][n_cells][CELL_CAP]   x | y<<12 | score<<24 (relative to the 16-px border)
  uint8_t* oct;                  // [I][nlevels][oct_stride] (only with oct_global_tab)
  size_t oct_stride;
  int oct_max_cells;             // most grid cells on one level (cell-prefix array in LDS)
  int oct_cand_cap;              // candidates per (image, level): CAND_CAP (keys in LDS) or CAND_CAP_BIG (keys in global scratch)
  int oct_global_keys;
  int oct_ln;                    // node-table capacity: max(N, 256) + 8 over the levels (N = per-level budget)
  int oct_global_tab;            // node tables in global scratch (budgets too large for LDS)
  int* lvl_ncand;                // [I][nlevels]
  int* sel_count;                // [I][nlevels]
  uint32_t* sel;                 // [I][nlevels][SEL_CAP] packed like cell_cand
  float* sel_angle;              // [I][nlevels][SEL_CAP]
  int* status;                   // [I] bit0: candidate overflow, bit1: node overflow, bit2: output overflow
  // outputs
  uint8_t* out_kps;              // [I][out_cap] x 28 bytes (ssx_keypoint)
  uint8_t* out_desc;             // [I][out_cap][32]
  int* out_n;                    // [I]
};

void launch_octree(const OrbDev& o, hipStream_t s);   // octree.hip

}  // namespace ssxorb

// host-side plan + buffers (one per ctx; re-planned when the geometry / parameters change)
struct OrbWorkspace {
  DevBuf arena;        // pyramids, candidates, octree scratch, outputs
  DevBuf input;        // uploaded host images (host-pointer entry points)
  DevBuf stereo;       // match / triangulation results of the batch entry points
  HostBuf stage;       // pinned staging
  HostBuf fetch;       // pinned: everything ssx_stereo_frame returns, fetched with ONE synchronisation
  ssxorb::OrbDev dev{};
  // plan key
  int rows = 0, cols = 0, I = 0, nlevels = 0, nfeatures = 0, ini_th = 0, min_th = 0, has_mask = 0, detect_only = 0;
  float scale_factor = 0.f;
  bool planned = false;
  // batch state (ssx_stereo_batch_dev / _enqueue / _fetch)
  const uint8_t* batch_imgs = nullptr;
  int batch_pairs = 0, batch_stride = 0;
  ssx_orb_params batch_orb{};
  ssx_match_params batch_mp{};
  ssx_stereo_rig batch_rig{};
  // stereo result views inside `stereo`
  int* match_idx = nullptr;
  int* match_dist = nullptr;
  double* xyz = nullptr;
  uint8_t* tri_ok = nullptr;
  int* pair_counts = nullptr;   // [pairs][4]
  // streaming batches (ssx_stereo_batch_host): two device buffers filled from the host on a copy stream of their own, so that
  // batch k + 1 crosses PCIe while batch k is being processed
  DevBuf ingest[2];
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
  bool free_pending[2] = {false, false};
  int up_first = 0, up_count = 0;  // uploaded batches waiting for ssx_stereo_batch_run: buffers up_first, up_first ^ 1
  int up_shape[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // pairs, stride, rows, cols of an uploaded batch
  HostBuf counts_pinned;        // [pairs][4] counts + [2 pairs] status words of the last enqueued batch
  hipEvent_t ev_counts[2] = {nullptr, nullptr};   // per run batch: its counts have reached counts_pinned
  int cnt_first = 0, cnt_count = 0, cnt_pairs[2] = {0, 0};
  ssx_status cnt_fail[2] = {SSX_OK, SSX_OK};           // a run that failed after consuming its batch still owns a slot of the FIFO
};

namespace ssxorb {
OrbWorkspace* get_ws(ssx_ctx* ctx);
// plan (allocate + upload cell tables) for I images of rows x cols; returns SSX_OK or an error
ssx_status plan(ssx_ctx* ctx, int rows, int cols, int I, const ssx_orb_params& prm, bool has_mask, bool detect_only);
// run the extraction pipeline on level-0 images already placed in the pyramid buffers
ssx_status run_pipeline(ssx_ctx* ctx);
// stage level 0 from a device/host-layout buffer [I][rows][stride]
// download the keypoints / descriptors of one image of the last run (synchronises the stream)
ssx_status fetch_image(ssx_ctx* ctx, int image, int cap, ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n);
ssx_status stage_level0(ssx_ctx* ctx, const uint8_t* imgs_dev, int stride, size_t img_bytes, const uint8_t* masks_dev,
                        int mask_stride, size_t mask_bytes);
}  // namespace ssxorb
