// ssvio_amd/csrc/octree.hip -- DistributeOctTree on the device, one 1024-thread workgroup per (image, level).
//
// Replaces ORBextractor::DistributeOctTree + ExtractorNode::DivideNode
// (/root/reference/src/ssvio/orbextractor.cpp:340-568, 282-338).  The reference walks a std::list and splits
// nodes one at a time; the selection it produces depends only on (a) the order of the list, (b) the insertion
// order of the keypoints inside every node and (c) the processing order of the "largest first" phase.  All
// three are reproduced with flat arrays and prefix sums (tools/octree_model.py is the executable statement
// of this formulation, checked against the sequential oracle):
//   * every node owns a contiguous range of the key array; a division is a STABLE 4-way partition of that range,
//     computed for all divided nodes at once from one packed (4 x 16 bit) exclusive scan over the key positions;
//   * the node table is kept IN LIST ORDER: after a round the children of the i-th processed node sit at
//     [T - P_i - cc_i, T - P_i) in quadrant order 4,3,2,1 (push_front semantics) and the untouched nodes follow
//     in their old relative order;
//   * phase 2 sorts the expandable nodes by (size, creation index) descending with an LDS bitonic sort and
//     cuts the list at the first prefix that reaches N nodes.
// Tie-break of equal sizes: creation order (the reference compares heap pointers, orbextractor.cpp:486, which is
// not reproducible); identical to the oracle.
//
// Memory: per key only two u32 streams exist (packed candidate x|y|score, and key|node); the quadrant of every key
// lives in LDS for the duration of a round; each thread walks a contiguous run of key positions, so the node
// record is re-read only when the run crosses a node boundary.  (Round 1 kept a 64-bit scan value, separate key /
// node-id arrays and unpacked coordinates per key in global memory: rocprof showed 400 MB of L2<->HBM traffic per
// launch for 1.3 MB of algorithmic bytes.)
#include "orb_ws.hpp"

namespace ssxorb {

namespace {

typedef unsigned long long u64;
constexpr int T = OCT_THREADS;
constexpr int NW = T / 64;

struct Oct {
  uint32_t* pk;            // packed candidates
  uint32_t* kn[2];         // key | node << 16
  OctNode* nodes[2];
  u64* eb;                 // packed scan value at the first key of each node
  uint16_t *proc, *expa, *expb, *newpos;
  u64 *c4, *kid4;
  uint32_t *ccp, *exp, *sortk;
};

__device__ __forceinline__ Oct carve(uint8_t* base)
{
  Oct o;
  o.pk = (uint32_t*)(base + OctLayout::candpk);
  o.kn[0] = (uint32_t*)(base + OctLayout::keynode); o.kn[1] = o.kn[0] + CAND_CAP;
  o.nodes[0] = (OctNode*)(base + OctLayout::nodes); o.nodes[1] = o.nodes[0] + NODE_CAP;
  o.eb = (u64*)(base + OctLayout::ebeg);
  o.proc = (uint16_t*)(base + OctLayout::proc);
  o.expa = (uint16_t*)(base + OctLayout::expa);
  o.expb = (uint16_t*)(base + OctLayout::expb);
  o.c4 = (u64*)(base + OctLayout::c4);
  o.kid4 = (u64*)(base + OctLayout::kid4);
  o.ccp = (uint32_t*)(base + OctLayout::ccp);
  o.exp = (uint32_t*)(base + OctLayout::exp_);
  o.newpos = (uint16_t*)(base + OctLayout::newpos);
  o.sortk = (uint32_t*)(base + OctLayout::sortk);
  return o;
}

// block-wide exclusive scan of one value per thread (thread order); total returned to every thread.
template <class V>
__device__ __forceinline__ V block_excl_scan(V v, V* s_wave /*[NW+1]*/, V& total)
{
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  V inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const V up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (t == 0) {
    V run = 0;
    for (int w = 0; w < NW; ++w) { const V x = s_wave[w]; s_wave[w] = run; run += x; }
    s_wave[NW] = run;
  }
  __syncthreads();
  const V res = s_wave[wave] + (inc - v);
  total = s_wave[NW];
  __syncthreads();
  return res;
}

__device__ __forceinline__ int quadrant(int x, int y, const OctNode& n)
{
  // DivideNode: halfX = ceil((UR.x - UL.x) / 2), children split at UL + half (orbextractor.cpp:285-327)
  const int mx = n.ulx + (int)ceilf((float)(n.brx - n.ulx) / 2);
  const int my = n.uly + (int)ceilf((float)(n.bry - n.uly) / 2);
  return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}

__device__ __forceinline__ unsigned f16(u64 v, int q) { return (unsigned)((v >> (16 * q)) & 0xFFFFu); }
__device__ __forceinline__ int pk_x(uint32_t p) { return (int)(p & 0xFFF); }
__device__ __forceinline__ int pk_y(uint32_t p) { return (int)((p >> 12) & 0xFFF); }
__device__ __forceinline__ int pk_r(uint32_t p) { return (int)(p >> 24); }

}  // namespace

__global__ __launch_bounds__(OCT_THREADS) void k_octree(OrbDev d)
{
  __shared__ u64 s_w64[NW + 1];
  __shared__ uint32_t s_w32[NW + 1];
  __shared__ uint32_t s_sort[4096];
  __shared__ uint8_t s_q[CAND_CAP];     // quadrant of every key position in the current round (4 = node not divided)
  __shared__ int s_i[8];
  const int img = blockIdx.x, level = blockIdx.y, t = threadIdx.x;   // see launch_octree for the order
  if (d.detect_only && level > 0) return;
  const int il = img * d.nlevels + level;
  Oct o = carve(d.oct + (size_t)il * OctLayout::total);
  const int N = d.feat[level];
  const int cell0 = d.lvl_cell0[level], cell1 = d.lvl_cell0[level + 1];
  const int ncell = cell1 - cell0;
  const int* ccount = d.cell_count + (size_t)img * d.n_cells + cell0;
  const uint32_t* ccand = d.cell_cand + ((size_t)img * d.n_cells + cell0) * CELL_CAP;

  // ---- gather the candidates of this level in reference order (cell-row-major, row-major inside a cell) ----
  int M;
  {
    const int chunk = (ncell + T - 1) / T;
    const int lo = min(t * chunk, ncell), hi = min(lo + chunk, ncell);
    uint32_t mine = 0;
    for (int c = lo; c < hi; ++c) mine += (uint32_t)ccount[c];
    uint32_t total;
    uint32_t pos = block_excl_scan<uint32_t>(mine, s_w32, total);
    for (int c = lo; c < hi; ++c) {
      const int n = ccount[c];
      for (int k = 0; k < n; ++k, ++pos)
        if (pos < (uint32_t)CAND_CAP) o.pk[pos] = ccand[(size_t)c * CELL_CAP + k];
    }
    M = (int)min(total, (uint32_t)CAND_CAP);
    if (t == 0) {
      d.lvl_ncand[il] = (int)total;
      if (total > (uint32_t)CAND_CAP) atomicOr(&d.status[img], 1);
    }
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
  const int kchunk = (M + T - 1) / T;
  const int klo = min(t * kchunk, M), khi = min(klo + kchunk, M);
  int cur = 0;
  int nNodes = 0;
  {
    // root of every candidate once (s_q doubles as scratch), then one stable pass per root
    for (int p = klo; p < khi; ++p) {
      int rt = (int)((float)pk_x(o.pk[p]) / hX);
      s_q[p] = (uint8_t)(rt >= nIni ? nIni - 1 : rt);
    }
    int base = 0;
    for (int r = 0; r < nIni; ++r) {
      uint32_t mine = 0;
      for (int p = klo; p < khi; ++p) mine += (s_q[p] == r);
      uint32_t total;
      uint32_t pos = block_excl_scan<uint32_t>(mine, s_w32, total);
      if (total > 0) {
        for (int p = klo; p < khi; ++p)
          if (s_q[p] == r) { o.kn[cur][base + pos] = (uint32_t)p | ((uint32_t)nNodes << 16); ++pos; }
        if (t == 0) {
          OctNode n;
          n.b = (uint16_t)base; n.e = (uint16_t)(base + total);
          n.ulx = (int16_t)(int)(hX * (float)r); n.uly = 0;
          n.brx = (int16_t)(int)(hX * (float)(r + 1)); n.bry = (int16_t)(maxY - minY);
          n.pidx = 0; n.no_more = (total == 1); n.div = 0;
          o.nodes[cur][nNodes] = n;
        }
        ++nNodes;
        base += (int)total;
      }
      __syncthreads();
    }
  }

  uint16_t* exp_cur = o.expa;
  uint16_t* exp_nxt = o.expb;
  int nExp = 0;
  bool finish = false;
  int phase = 1;
  int guard = 0;
  while (!finish && ++guard < 256) {
    OctNode* nd = o.nodes[cur];
    OctNode* nn = o.nodes[cur ^ 1];
    const uint32_t* kn = o.kn[cur];
    uint32_t* kn_next = o.kn[cur ^ 1];
    const int prevSize = nNodes;
    const int nchunk = (nNodes + T - 1) / T;
    const int nlo = min(t * nchunk, nNodes), nhi = min(nlo + nchunk, nNodes);
    int m = 0;   // length of the processing list
    // ---------------- build the processing list ----------------
    if (phase == 1) {
      uint32_t mine = 0;
      for (int j = nlo; j < nhi; ++j) mine += !nd[j].no_more;
      uint32_t total;
      uint32_t pos = block_excl_scan<uint32_t>(mine, s_w32, total);
      for (int j = nlo; j < nhi; ++j)
        if (!nd[j].no_more) { o.proc[pos] = (uint16_t)j; nd[j].pidx = (uint16_t)pos; nd[j].div = 1; ++pos; }
      m = (int)total;
    } else {
      // sort the expandable nodes by (size, creation index) descending: key = size << 16 | index
      int n2 = 1;
      while (n2 < nExp) n2 <<= 1;
      uint32_t* sk = (n2 <= 4096) ? s_sort : o.sortk;
      for (int i = t; i < n2; i += T) {
        uint32_t key = 0;
        if (i < nExp) { const OctNode& x = nd[exp_cur[i]]; key = ((uint32_t)(x.e - x.b) << 16) | (uint32_t)i; }
        sk[i] = key;
      }
      __syncthreads();
      for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = t; i < n2; i += T) {
            const int l = i ^ j;
            if (l > i) {
              const uint32_t a = sk[i], b = sk[l];
              const bool desc = ((i & k) == 0);
              if (desc ? (a < b) : (a > b)) { sk[i] = b; sk[l] = a; }
            }
          }
          __syncthreads();
        }
      for (int i = t; i < nExp; i += T) {
        const int j = exp_cur[sk[i] & 0xFFFFu];
        o.proc[i] = (uint16_t)j; nd[j].pidx = (uint16_t)i; nd[j].div = 1;
      }
      m = nExp;
    }
    __syncthreads();
    // ---------------- quadrant of every key (LDS) + packed exclusive scan over the key positions ----------------
    u64 my_prefix;
    u64 Etotal;
    {
      u64 mine = 0;
      int cn = -1;
      OctNode x{};
      for (int p = klo; p < khi; ++p) {
        const uint32_t v = kn[p];
        const int node = (int)(v >> 16);
        if (node != cn) { cn = node; x = nd[node]; }
        int q = 4;
        if (x.div) { const uint32_t c = o.pk[v & 0xFFFFu]; q = quadrant(pk_x(c), pk_y(c), x); mine += 1ull << (16 * q); }
        s_q[p] = (uint8_t)q;
      }
      my_prefix = block_excl_scan<u64>(mine, s_w64, Etotal);
      // scan value at the first key of every node (the stable rank of a key = running value - value at node start)
      u64 run = my_prefix;
      cn = -1;
      for (int p = klo; p < khi; ++p) {
        const int node = (int)(kn[p] >> 16);
        if (node != cn) { cn = node; if (nd[node].b == p) o.eb[node] = run; }
        const int q = s_q[p];
        if (q < 4) run += 1ull << (16 * q);
      }
    }
    __syncthreads();
    // quadrant counts of node x = E(x.e) - E(x.b), with E(pos) the scan value at the node that starts at pos
    auto E_at = [&](int pos) -> u64 { return pos >= M ? Etotal : o.eb[kn[pos] >> 16]; };
    // ---------------- quadrant counts / child counts per processed node ----------------
    const int pchunk = (m + T - 1) / T;
    int plo = min(t * pchunk, m), phi = min(plo + pchunk, m);
    for (int i = plo; i < phi; ++i) {
      const OctNode& x = nd[o.proc[i]];
      o.c4[i] = E_at(x.e) - o.eb[o.proc[i]];
    }
    if (phase == 2) {
      // cut the list at the first prefix that brings the node count to >= N (orbextractor.cpp:487-539)
      uint32_t mine = 0;
      for (int i = plo; i < phi; ++i) {
        const u64 c = o.c4[i];
        const int cc = (f16(c, 0) > 0) + (f16(c, 1) > 0) + (f16(c, 2) > 0) + (f16(c, 3) > 0);
        mine += (uint32_t)(cc - 1);
      }
      uint32_t total;
      uint32_t run = block_excl_scan<uint32_t>(mine, s_w32, total);
      if (t == 0) s_i[0] = m;
      __syncthreads();
      for (int i = plo; i < phi; ++i) {
        const u64 c = o.c4[i];
        const int cc = (f16(c, 0) > 0) + (f16(c, 1) > 0) + (f16(c, 2) > 0) + (f16(c, 3) > 0);
        const uint32_t before = run;
        run += (uint32_t)(cc - 1);
        // first index whose inclusive prefix reaches N: size0 + before < N <= size0 + run
        if ((int)(nNodes + before) < N && (int)(nNodes + run) >= N) s_i[0] = i + 1;
      }
      __syncthreads();
      const int mt = s_i[0];
      for (int i = mt + t; i < m; i += T) nd[o.proc[i]].div = 0;   // not processed in this round
      m = mt;
      __syncthreads();
      const int pc2 = (m + T - 1) / T;
      plo = min(t * pc2, m); phi = min(plo + pc2, m);
    }
    uint32_t Tchild, EXtot, Kkeep;
    {
      uint32_t mine_cc = 0, mine_ex = 0;
      for (int i = plo; i < phi; ++i) {
        const u64 c = o.c4[i];
        for (int q = 0; q < 4; ++q) { mine_cc += f16(c, q) > 0; mine_ex += f16(c, q) > 1; }
      }
      uint32_t run_cc = block_excl_scan<uint32_t>(mine_cc, s_w32, Tchild);
      uint32_t run_ex = block_excl_scan<uint32_t>(mine_ex, s_w32, EXtot);
      for (int i = plo; i < phi; ++i) {
        const u64 c = o.c4[i];
        o.ccp[i] = run_cc; o.exp[i] = run_ex;
        for (int q = 0; q < 4; ++q) { run_cc += f16(c, q) > 0; run_ex += f16(c, q) > 1; }
      }
      uint32_t mine_k = 0;
      for (int j = nlo; j < nhi; ++j) mine_k += !nd[j].div;
      uint32_t run_k = block_excl_scan<uint32_t>(mine_k, s_w32, Kkeep);
      for (int j = nlo; j < nhi; ++j)
        if (!nd[j].div) { o.newpos[j] = (uint16_t)min(Tchild + run_k, (uint32_t)(NODE_CAP - 1)); ++run_k; }
    }
    const int nNew = (int)(Tchild + Kkeep);
    if (nNew > NODE_CAP) {
      if (t == 0) atomicOr(&d.status[img], 2);
      break;
    }
    __syncthreads();
    // ---------------- create the children (list order = push_front order) and move the survivors ----------------
    for (int i = plo; i < phi; ++i) {
      const OctNode x = nd[o.proc[i]];
      const u64 c = o.c4[i];
      const int cc = (f16(c, 0) > 0) + (f16(c, 1) > 0) + (f16(c, 2) > 0) + (f16(c, 3) > 0);
      const int mx = x.ulx + (int)ceilf((float)(x.brx - x.ulx) / 2);
      const int my = x.uly + (int)ceilf((float)(x.bry - x.uly) / 2);
      const int first = (int)Tchild - (int)o.ccp[i] - cc;   // block of this node's children in the new list
      int off = x.b, above = cc, ex = (int)o.exp[i];
      u64 kid = 0;
      for (int q = 0; q < 4; ++q) {
        const int n = (int)f16(c, q);
        if (n == 0) continue;
        --above;                      // children with a larger quadrant index come first (push_front)
        const int id = first + above;
        OctNode ch;
        ch.b = (uint16_t)off; ch.e = (uint16_t)(off + n);
        ch.ulx = (int16_t)((q & 1) ? mx : x.ulx); ch.brx = (int16_t)((q & 1) ? x.brx : mx);
        ch.uly = (int16_t)((q & 2) ? my : x.uly); ch.bry = (int16_t)((q & 2) ? x.bry : my);
        ch.pidx = 0; ch.no_more = (n == 1); ch.div = 0;
        nn[id] = ch;
        kid |= (u64)id << (16 * q);
        if (n > 1) exp_nxt[ex++] = (uint16_t)id;
        off += n;
      }
      o.kid4[i] = kid;
    }
    for (int j = nlo; j < nhi; ++j)
      if (!nd[j].div) { OctNode x = nd[j]; x.pidx = 0; nn[o.newpos[j]] = x; }
    __syncthreads();
    // ---------------- move the keys (stable partition) ----------------
    {
      u64 run = my_prefix;
      int cn = -1;
      OctNode x{};
      u64 c = 0, kid = 0, ebn = 0;
      uint32_t np = 0;
      for (int p = klo; p < khi; ++p) {
        const uint32_t v = kn[p];
        const int node = (int)(v >> 16);
        if (node != cn) {
          cn = node; x = nd[node];
          if (x.div) { c = o.c4[x.pidx]; kid = o.kid4[x.pidx]; ebn = o.eb[node]; }
          else np = o.newpos[node];
        }
        const int q = s_q[p];
        if (q < 4 && x.div) {
          int before = 0;
          for (int qq = 0; qq < q; ++qq) before += (int)f16(c, qq);
          const int rank = (int)f16(run, q) - (int)f16(ebn, q);
          kn_next[x.b + before + rank] = (v & 0xFFFFu) | ((uint32_t)f16(kid, q) << 16);
        } else {
          kn_next[p] = (v & 0xFFFFu) | (np << 16);
        }
        // keys of phase-2 candidates that were cut from this round still sit in the scan: keep `run` in step
        if (q < 4) run += 1ull << (16 * q);
      }
    }
    __syncthreads();
    cur ^= 1;
    nNodes = nNew;
    nExp = (int)EXtot;
    { uint16_t* tmp = exp_cur; exp_cur = exp_nxt; exp_nxt = tmp; }
    // ---------------- termination (orbextractor.cpp:472-476, 541-543) ----------------
    if (nNodes >= N || nNodes == prevSize) finish = true;
    else if (phase == 1 && nNodes + 3 * nExp > N) phase = 2;
  }
  __syncthreads();
  // ---- best response per node, first wins ties (orbextractor.cpp:549-565), in list order ----
  {
    const OctNode* nd = o.nodes[cur];
    const uint32_t* kn = o.kn[cur];
    const int nOut = min(nNodes, SEL_CAP);
    for (int j = t; j < nOut; j += T) {
      const OctNode x = nd[j];
      uint32_t best = o.pk[kn[x.b] & 0xFFFFu];
      for (int p = x.b + 1; p < x.e; ++p) {
        const uint32_t c = o.pk[kn[p] & 0xFFFFu];
        if (pk_r(c) > pk_r(best)) best = c;
      }
      sel[j] = best;
    }
    if (t == 0) {
      *sel_count = nOut;
      if (nNodes > SEL_CAP) atomicOr(&d.status[img], 4);
    }
  }
}

void launch_octree(const OrbDev& o, hipStream_t s)
{
  // image-major grid: workgroups are dealt round-robin to the 8 XCDs, so with the level as the fastest index
  // (8 levels) every level-0 workgroup - the long one - would land on the same XCD.  Level-major order starts
  // all level-0 workgroups first and spreads them over the whole chip.
  hipLaunchKernelGGL(k_octree, dim3(o.I, o.detect_only ? 1 : o.nlevels), dim3(OCT_THREADS), 0, s, o);
}

}  // namespace ssxorb
