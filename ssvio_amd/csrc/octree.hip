// ssvio_amd/csrc/octree.hip -- DistributeOctTree on the device, one 1024-thread workgroup per (image, level).
//
// Replaces ORBextractor::DistributeOctTree + ExtractorNode::DivideNode
// (/root/reference/src/ssvio/orbextractor.cpp:340-568, 282-338).  The reference walks a std::list and splits
// nodes one at a time, moving the keypoints of a node into its children.  What it SELECTS depends only on
//   (a) the order of the node list,
//   (b) how many keypoints every node holds (bNoMore, "expandable", the largest-first order of phase 2), and
//   (c) inside a node, the keypoint with the largest response, the FIRST such keypoint in the node's vector.
// A node's vector is always in the original candidate order (the root assignment and every DivideNode are stable),
// so (c) is "largest response, then smallest candidate index", and the keypoints never have to move:
//   * every thread owns up to 16 CONSECUTIVE candidates, kept in LDS in a conflict-free [slot][thread] layout
//     (packed x|y|score and the id of the node they are in);
//   * a round = one pass over the keys that adds the quadrant of every key of a divided node to that node's packed
//     4 x 16-bit counter (64-bit LDS atomics, one per run of equal nodes in a thread's keys), node-level
//     bookkeeping, and a second pass that renames each key's node id to the child (or to the survivor's new
//     position);
//   * the node table (boxes, counts, processing list, child ids, expandable lists) lives in LDS, IN LIST ORDER:
//     after a round the children of the i-th processed node sit at [T - P_i - cc_i, T - P_i) in quadrant order
//     4,3,2,1 (push_front semantics) and the untouched nodes follow in their old relative order
//     (tools/octree_model.py is the executable statement of this bookkeeping, checked against the sequential
//     oracle);
//   * phase 2 orders the expandable nodes by (size, creation index) descending by counting ranks, and cuts the
//     list at the first prefix that reaches N nodes;
//   * the winner of every final node is one LDS atomicMax on (response << 16 | 65535 - candidate index).
// Tie-break of equal sizes in phase 2: creation order (the reference compares heap pointers,
// orbextractor.cpp:486, which is not reproducible); identical to the oracle.
//
// History: v1 kept per-key 64-bit scan values and unpacked coordinates in global memory (rocprof: 400 MB of
// L2<->HBM traffic per launch for 1.3 MB of algorithmic bytes); v2 packed the per-key streams but still moved the
// keys with a stable partition every round through global memory (~150 us per level-0 workgroup, bound by ~30
// barriers and a dozen dependent global round trips per round).  This version touches global memory twice: the
// cell lists in, the selection out.
#include "orb_ws.hpp"

namespace ssxorb {

namespace {

typedef unsigned long long u64;
constexpr int T = OCT_THREADS;
constexpr int NW = T / 64;
constexpr unsigned NONE = 0xFFFFu;
struct Box { int16_t ulx, uly, brx, bry; };   // node corners (UL, BR), relative to the level's border

// node-level arrays, LN entries each (OCT_NODE_BYTES per entry): LDS, or global scratch for very large budgets
struct Tab {
  u64* c4;            // per PROCESSED node: 4 x 16-bit quadrant counts
  u64* kid4;          // per processed node: 4 x 16-bit child ids (new list positions)
  Box* box;           // per node, [2][LN] (ping-pong over rounds; indexed, never selected by pointer, so the
                      // compiler keeps the LDS address space and emits ds_ instead of flat_ instructions)
  uint32_t* key;      // phase-2 sort keys; reused for the per-node winner at the end
  uint16_t* cnt;      // keypoints per node, [2][LN]
  uint16_t* pidx;     // per node: index in the processing list of this round, NONE = not divided
  uint16_t* newpos;   // per surviving node: position in the new list
  uint16_t* proc;     // processing list -> node
  uint16_t* exp;      // expandable children (count > 1) in creation order, [2][LN]
};

__device__ __forceinline__ Tab carve(uint8_t* base, int LN)
{
  Tab o;
  const size_t n = (size_t)LN;
  o.c4 = (u64*)base;
  o.kid4 = o.c4 + n;
  o.box = (Box*)(o.kid4 + n);
  o.key = (uint32_t*)(o.box + 2 * n);
  o.cnt = (uint16_t*)(o.key + n);
  o.pidx = o.cnt + 2 * n;
  o.newpos = o.pidx + n;
  o.proc = o.newpos + n;
  o.exp = o.proc + n;
  return o;
}
static_assert(OCT_NODE_BYTES == 8 + 8 + 16 + 4 + 4 + 2 + 2 + 2 + 4, "Tab layout and OCT_NODE_BYTES disagree");

// block-wide exclusive scan of one packed value per thread (thread order); total returned to every thread.
// ONE barrier: the wave totals go to s_w[par] and every thread adds up the waves before its own; par alternates,
// so a buffer is rewritten only after the barrier of the following call.
__device__ __forceinline__ u64 block_scan(u64 v, u64 (*s_w)[NW], int& par, u64& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_w[par][wave] = inc;
  __syncthreads();
  u64 before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const u64 x = s_w[par][w];
    tot += x;
    if (w < wave) before += x;
  }
  par ^= 1;
  total = tot;
  return before + (inc - v);
}

__device__ __forceinline__ int quadrant(int x, int y, const Box& n)
{
  // DivideNode: halfX = ceil((UR.x - UL.x) / 2), children split at UL + half (orbextractor.cpp:285-327);
  // ceil(w / 2.f) == (w + 1) >> 1 for the non-negative integer widths that occur
  const int mx = n.ulx + ((n.brx - n.ulx + 1) >> 1);
  const int my = n.uly + ((n.bry - n.uly + 1) >> 1);
  return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}

__device__ __forceinline__ unsigned f16(u64 v, int q) { return (unsigned)((v >> (16 * q)) & 0xFFFFu); }
__device__ __forceinline__ int pk_x(uint32_t p) { return (int)(p & 0xFFF); }
__device__ __forceinline__ int pk_y(uint32_t p) { return (int)((p >> 12) & 0xFFF); }
__device__ __forceinline__ uint32_t pk_r(uint32_t p) { return p >> 24; }
__device__ __forceinline__ int nonzero4(u64 c) { return (f16(c, 0) > 0) + (f16(c, 1) > 0) + (f16(c, 2) > 0) + (f16(c, 3) > 0); }
__device__ __forceinline__ int above1_4(u64 c) { return (f16(c, 0) > 1) + (f16(c, 1) > 1) + (f16(c, 2) > 1) + (f16(c, 3) > 1); }

// GLOBAL_TAB = false: node tables in dynamic LDS (OCT_NODE_BYTES * LN bytes); true: in the global scratch block.
// GLOBAL_KEYS = false: per-key state (6 bytes per candidate, <= 16384 candidates per level) in LDS; true: in the
// global scratch block (images above ~0.6 Mpx: up to 65536 candidates per level, 64 keys per thread).
template <bool GLOBAL_TAB, bool GLOBAL_KEYS>
__global__ __launch_bounds__(OCT_THREADS) void k_octree(OrbDev d)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ u64 s_w[2][NW];
  __shared__ int s_root[64], s_rootpos[64];
  __shared__ int s_i[2];
  const int img = blockIdx.x, level = blockIdx.y, t = threadIdx.x;   // see launch_octree for the order
  const int lane = t & 63;
  if (d.detect_only && level > 0) return;
  const int il = img * d.nlevels + level;
  const int LN = d.oct_ln;
  Tab tb;
  uint32_t* s_cellpos;                                                 // [cells of the level + 1], dynamic LDS
  const int CAP = d.oct_cand_cap;
  uint8_t* scratch = d.oct + (size_t)il * d.oct_stride;                // [keys (GLOBAL_KEYS) | node tables (GLOBAL_TAB)]
  if constexpr (GLOBAL_TAB) {
    tb = carve(scratch + (GLOBAL_KEYS ? 6 * (size_t)CAP : 0), LN);
    s_cellpos = reinterpret_cast<uint32_t*>(smem);
  } else {
    tb = carve(smem, LN);
    s_cellpos = reinterpret_cast<uint32_t*>(smem + (((size_t)OCT_NODE_BYTES * LN + 15) & ~size_t(15)));
  }
  // per-key state, slot k of thread t at [k * T + t]: packed candidate, and node id | quadrant << 14
  uint32_t* s_kpk;                                                     // [CAP] x | y << 12 | score << 24
  uint16_t* s_kn;                                                      // [CAP]
  if constexpr (GLOBAL_KEYS) {
    s_kpk = reinterpret_cast<uint32_t*>(scratch);
    s_kn = reinterpret_cast<uint16_t*>(s_kpk + CAP);
  } else {
    s_kpk = s_cellpos + ((d.oct_max_cells + 1 + 3) & ~3);
    s_kn = reinterpret_cast<uint16_t*>(s_kpk + CAP);
  }
  const int N = d.feat[level];
  const int cell0 = d.lvl_cell0[level], cell1 = d.lvl_cell0[level + 1];
  const int ncell = cell1 - cell0;
  const int* ccount = d.cell_count + (size_t)img * d.n_cells + cell0;
  const uint32_t* __restrict__ ccand = d.cell_cand + ((size_t)img * d.n_cells + cell0) * CELL_CAP;
  int par = 0;

  // ---- gather the candidates of this level in reference order (cell-row-major, row-major inside a cell) ----
  // s_cellpos[c] = index of cell c's first candidate in reference order; the candidates themselves stay where
  // k_fast_cells wrote them and are read straight into registers below.
  int M;
  {
    const int chunk = (ncell + T - 1) / T;
    const int lo = min(t * chunk, ncell), hi = min(lo + chunk, ncell);
    u64 mine = 0;
    for (int c = lo; c < hi; ++c) mine += (u64)ccount[c];
    u64 total;
    uint32_t pos = (uint32_t)block_scan(mine, s_w, par, total);
    for (int c = lo; c < hi; ++c) { s_cellpos[c] = pos; pos += (uint32_t)ccount[c]; }
    if (t == 0) {
      s_cellpos[ncell] = (uint32_t)total;
      d.lvl_ncand[il] = (int)total;
      if (total > (u64)CAP) atomicOr(&d.status[img], 1);
    }
    M = (int)min(total, (u64)CAP);
    if (t < 64) s_root[t] = 0;
  }
  __syncthreads();
  int* sel_count = d.sel_count + il;
  uint32_t* sel = d.sel + (size_t)il * SEL_CAP;
  if (M == 0) { if (t == 0) *sel_count = 0; return; }

  // ---- root nodes (orbextractor.cpp:345-392) ----
  const int minX = EDGE_THRESHOLD - 3, minY = minX;
  const int maxX = d.lvl_cols[level] - EDGE_THRESHOLD + 3, maxY = d.lvl_rows[level] - EDGE_THRESHOLD + 3;
  const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
  if (nIni < 1 || nIni > 64) { if (t == 0) *sel_count = 0; return; }
  const float hX = (float)(maxX - minX) / (float)nIni;
  // my candidates: the CONTIGUOUS range [p0, p0 + nk) of the reference order.  Neighbouring candidates mostly sit
  // in the same node, so a thread folds runs of equal targets in registers and issues one LDS atomic per run
  // (one atomic per key made every round LDS-atomic bound: ~4 clocks per conflicting lane, 13 us per round).
  const int kchunk = (M + T - 1) / T;
  const int p0 = min(t * kchunk, M);
  const int nk = min(kchunk, M - p0);
  {
    // cell of p0 by binary search, then walk (empty cells are skipped)
    int c = 0, off = 0, cend = 0;
    if (nk > 0) {
      int lo = 0, hi = ncell;                          // invariant: s_cellpos[lo] <= p0 < s_cellpos[hi]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_cellpos[mid] <= (uint32_t)p0) lo = mid; else hi = mid; }
      c = lo;
      off = p0 - (int)s_cellpos[c];
      cend = (int)(s_cellpos[c + 1] - s_cellpos[c]);
    }
    int run_root = -1, run_n = 0;
    for (int k = 0; k < nk; ++k) {
      while (off >= cend) { ++c; off = 0; cend = (int)(s_cellpos[c + 1] - s_cellpos[c]); }
      const uint32_t v = ccand[(size_t)c * CELL_CAP + off];
      ++off;
      s_kpk[k * T + t] = v;
      const int rt = (int)((float)pk_x(v) / hX);
      const int root = rt >= nIni ? nIni - 1 : rt;
      s_kn[k * T + t] = (uint16_t)root;
      if (root != run_root) {
        if (run_n > 0) atomicAdd(&s_root[run_root], run_n);
        run_root = root; run_n = 0;
      }
      ++run_n;
    }
    if (run_n > 0) atomicAdd(&s_root[run_root], run_n);
  }
  __syncthreads();
  if (t < 64) {
    // the list keeps the non-empty roots in creation order
    const int c = t < nIni ? s_root[t] : 0;
    const u64 bal = __ballot(c > 0);
    const int pos = __popcll(bal & ((1ull << lane) - 1ull));
    s_rootpos[t] = pos;
    if (c > 0) {
      Box b;
      b.ulx = (int16_t)(int)(hX * (float)t); b.uly = 0;
      b.brx = (int16_t)(int)(hX * (float)(t + 1)); b.bry = (int16_t)(maxY - minY);
      tb.box[pos] = b;
      tb.cnt[pos] = (uint16_t)c;
    }
    if (t == 0) s_i[0] = __popcll(bal);
  }
  __syncthreads();
  int nNodes = s_i[0];
  for (int k = 0; k < nk; ++k) s_kn[k * T + t] = (uint16_t)s_rootpos[s_kn[k * T + t]];

  int cur = 0, ecur = 0, nExp = 0, phase = 1, guard = 0;
  bool finish = false;
  while (!finish && ++guard < 256) {
    const Box* bx = tb.box + cur * LN;
    const uint16_t* ct = tb.cnt + cur * LN;
    Box* nbx = tb.box + (cur ^ 1) * LN;
    uint16_t* nct = tb.cnt + (cur ^ 1) * LN;
    const uint16_t* exp_cur = tb.exp + ecur * LN;
    uint16_t* exp_nxt = tb.exp + (ecur ^ 1) * LN;
    const int prevSize = nNodes;
    const int nchunk = (nNodes + T - 1) / T;
    const int nlo = min(t * nchunk, nNodes), nhi = min(nlo + nchunk, nNodes);
    int m = 0;   // length of the processing list
    // ---------------- build the processing list ----------------
    if (phase == 1) {
      u64 mine = 0;
      for (int j = nlo; j < nhi; ++j) mine += ct[j] > 1;   // !bNoMore
      u64 total;
      uint32_t pos = (uint32_t)block_scan(mine, s_w, par, total);
      for (int j = nlo; j < nhi; ++j) {
        if (ct[j] > 1) { tb.proc[pos] = (uint16_t)j; tb.pidx[j] = (uint16_t)pos; tb.c4[pos] = 0; ++pos; }
        else tb.pidx[j] = (uint16_t)NONE;
      }
      m = (int)total;
    } else {
      // expandable nodes by (size, creation index) descending: key = size << 16 | index, rank by counting
      // (keys are unique and non-zero; the array is padded with zeros to a multiple of 4 for 16-byte reads)
      const int nExp4 = (nExp + 3) & ~3;
      for (int i = t; i < nExp4; i += T) tb.key[i] = i < nExp ? (((uint32_t)ct[exp_cur[i]] << 16) | (uint32_t)i) : 0u;
      for (int j = nlo; j < nhi; ++j) tb.pidx[j] = (uint16_t)NONE;
      __syncthreads();
      for (int i = t; i < nExp; i += T) {
        const uint32_t ki = tb.key[i];
        int rank = 0;
#pragma unroll 4
        for (int l = 0; l < nExp4; l += 4) {
          const uint4 kk = *reinterpret_cast<const uint4*>(&tb.key[l]);
          rank += (kk.x > ki) + (kk.y > ki) + (kk.z > ki) + (kk.w > ki);
        }
        const int j = exp_cur[i];
        tb.proc[rank] = (uint16_t)j; tb.pidx[j] = (uint16_t)rank; tb.c4[rank] = 0;
      }
      m = nExp;
    }
    __syncthreads();
    // ---------------- key pass 1: quadrant counts of the divided nodes ----------------
    {
      unsigned run_pi = NONE;
      u64 run_c = 0;
      for (int k = 0; k < nk; ++k) {
        const int j = s_kn[k * T + t] & 0x3FFF;
        const uint32_t v = s_kpk[k * T + t];
        const unsigned pi = tb.pidx[j];
        const int q = quadrant(pk_x(v), pk_y(v), bx[j]);
        s_kn[k * T + t] = (uint16_t)(j | (q << 14));                     // the quadrant rides in the top two bits
        if (pi != run_pi) {
          if (run_pi != NONE) atomicAdd(&tb.c4[run_pi], run_c);
          run_pi = pi; run_c = 0;
        }
        run_c += 1ull << (16 * q);
      }
      if (run_pi != NONE) atomicAdd(&tb.c4[run_pi], run_c);
    }
    __syncthreads();
    int pchunk = (m + T - 1) / T;
    int plo = min(t * pchunk, m), phi = min(plo + pchunk, m);
    if (phase == 2) {
      // cut the list at the first prefix that brings the node count to >= N (orbextractor.cpp:487-539)
      u64 mine = 0;
      for (int i = plo; i < phi; ++i) mine += (u64)(nonzero4(tb.c4[i]) - 1);
      u64 total;
      uint32_t run = (uint32_t)block_scan(mine, s_w, par, total);
      if (t == 0) s_i[0] = m;
      __syncthreads();
      for (int i = plo; i < phi; ++i) {
        const uint32_t before = run;
        run += (uint32_t)(nonzero4(tb.c4[i]) - 1);
        // first index whose inclusive prefix reaches N: size0 + before < N <= size0 + run
        if ((int)(nNodes + before) < N && (int)(nNodes + run) >= N) s_i[0] = i + 1;
      }
      __syncthreads();
      const int mt = s_i[0];
      for (int i = mt + t; i < m; i += T) tb.pidx[tb.proc[i]] = (uint16_t)NONE;   // not processed in this round
      m = mt;
      __syncthreads();
      pchunk = (m + T - 1) / T;
      plo = min(t * pchunk, m); phi = min(plo + pchunk, m);
    }
    // ---------------- one packed scan: children (bits 0-15), expandable children (16-31), survivors (32-47) ----
    uint32_t Tchild, EXtot, Kkeep;
    uint32_t run_cc, run_ex, run_k;
    {
      u64 mine = 0;
      for (int i = plo; i < phi; ++i) {
        const u64 c = tb.c4[i];
        mine += (u64)nonzero4(c) | ((u64)above1_4(c) << 16);
      }
      for (int j = nlo; j < nhi; ++j) mine += (u64)(tb.pidx[j] == NONE) << 32;
      u64 total;
      const u64 run = block_scan(mine, s_w, par, total);
      Tchild = (uint32_t)(total & 0xFFFFu); EXtot = (uint32_t)((total >> 16) & 0xFFFFu); Kkeep = (uint32_t)((total >> 32) & 0xFFFFu);
      run_cc = (uint32_t)(run & 0xFFFFu); run_ex = (uint32_t)((run >> 16) & 0xFFFFu); run_k = (uint32_t)((run >> 32) & 0xFFFFu);
    }
    const int nNew = (int)(Tchild + Kkeep);
    if (nNew > LN) {
      if (t == 0) atomicOr(&d.status[img], 2);
      break;
    }
    // ---------------- create the children (list order = push_front order) and move the survivors ----------------
    for (int i = plo; i < phi; ++i) {
      const int j = tb.proc[i];
      const Box x = bx[j];
      const u64 c = tb.c4[i];
      const int cc = nonzero4(c);
      const int mx = x.ulx + ((x.brx - x.ulx + 1) >> 1);
      const int my = x.uly + ((x.bry - x.uly + 1) >> 1);
      const int first = (int)Tchild - (int)run_cc - cc;   // block of this node's children in the new list
      int above = cc;
      u64 kid = 0;
      for (int q = 0; q < 4; ++q) {
        const int n = (int)f16(c, q);
        if (n == 0) continue;
        --above;                      // children with a larger quadrant index come first (push_front)
        const int id = first + above;
        Box ch;
        ch.ulx = (int16_t)((q & 1) ? mx : x.ulx); ch.brx = (int16_t)((q & 1) ? x.brx : mx);
        ch.uly = (int16_t)((q & 2) ? my : x.uly); ch.bry = (int16_t)((q & 2) ? x.bry : my);
        nbx[id] = ch;
        nct[id] = (uint16_t)n;
        kid |= (u64)id << (16 * q);
        if (n > 1) exp_nxt[run_ex++] = (uint16_t)id;
      }
      tb.kid4[i] = kid;
      run_cc += (uint32_t)cc;
    }
    for (int j = nlo; j < nhi; ++j)
      if (tb.pidx[j] == NONE) {
        const uint32_t np = Tchild + run_k++;
        tb.newpos[j] = (uint16_t)np;
        nbx[np] = bx[j];
        nct[np] = ct[j];
      }
    __syncthreads();
    // ---------------- key pass 2: rename every key's node ----------------
    for (int k = 0; k < nk; ++k) {
      const int kn = s_kn[k * T + t];
      const int j = kn & 0x3FFF, q = kn >> 14;
      const unsigned pi = tb.pidx[j];
      s_kn[k * T + t] = (pi != NONE) ? (uint16_t)f16(tb.kid4[pi], q) : tb.newpos[j];
    }
    __syncthreads();
    cur ^= 1;
    ecur ^= 1;
    nNodes = nNew;
    nExp = (int)EXtot;
    // ---------------- termination (orbextractor.cpp:472-476, 541-543) ----------------
    if (nNodes >= N || nNodes == prevSize) finish = true;
    else if (phase == 1 && nNodes + 3 * nExp > N) phase = 2;
  }
  __syncthreads();
  // ---- best response per node, first wins ties (orbextractor.cpp:549-565), in list order ----
  {
    uint32_t* best = tb.key;
    for (int j = t; j < nNodes; j += T) best[j] = 0;
    __syncthreads();
    {
      int run_j = -1;
      uint32_t run_best = 0;
      for (int k = 0; k < nk; ++k) {
        const int j = s_kn[k * T + t] & 0x3FFF;          // (quadrant bits may be left over when the loop ended on a capacity break)
        const uint32_t v = (pk_r(s_kpk[k * T + t]) << 16) | (uint32_t)(0xFFFF - (p0 + k));
        if (j != run_j) {
          if (run_j >= 0) atomicMax(&best[run_j], run_best);
          run_j = j; run_best = 0;
        }
        run_best = max(run_best, v);
      }
      if (run_j >= 0) atomicMax(&best[run_j], run_best);
    }
    __syncthreads();
    const int nOut = min(nNodes, SEL_CAP);
    for (int k = 0; k < nk; ++k) {
      const int j = s_kn[k * T + t] & 0x3FFF;
      const uint32_t c = s_kpk[k * T + t];
      if (j < nOut && best[j] == ((pk_r(c) << 16) | (uint32_t)(0xFFFF - (p0 + k)))) sel[j] = c;
    }
    if (t == 0) {
      *sel_count = nOut;
      if (nNodes > SEL_CAP) atomicOr(&d.status[img], 4);
    }
  }
}

}  // namespace

void launch_octree(const OrbDev& o, hipStream_t s)
{
  // image-major grid: workgroups are dealt round-robin to the 8 XCDs, so with the level as the fastest index
  // (8 levels) every level-0 workgroup - the long one - would land on the same XCD.  Level-major order starts
  // all level-0 workgroups first and spreads them over the whole chip.
  const dim3 grid(o.I, o.detect_only ? 1 : o.nlevels);
  const size_t tab = o.oct_global_tab ? 0 : (((size_t)OCT_NODE_BYTES * o.oct_ln + 15) & ~size_t(15));
  const size_t lds = tab + 4 * (size_t)((o.oct_max_cells + 1 + 3) & ~3) + (o.oct_global_keys ? 0 : 6 * (size_t)o.oct_cand_cap);
  void (*kern)(OrbDev) = o.oct_global_tab ? (o.oct_global_keys ? k_octree<true, true> : k_octree<true, false>)
                                          : (o.oct_global_keys ? k_octree<false, true> : k_octree<false, false>);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(OCT_THREADS), lds, s, o);
}

}  // namespace ssxorb
