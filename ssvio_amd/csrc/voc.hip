// ssvio_amd/csrc/voc.hip -- bag-of-words side of the loop closing on gfx950.
//
// ssvio turns a keyframe's ORB descriptors into a DBoW2 BowVector (/root/reference/src/ssvio/loopclosing.cpp:633,
// `dbow2_vocabulary_->transform(desc, bow2_vec_)`) and ranks the keyframe database with `dbow2_vocabulary_->score`
// (:84).  ORBVocabulary = TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ssvio/orbvocabulary.hpp:10), a k-ary
// tree of 32-byte descriptors: a feature descends from the root to the child with the smallest Hamming distance (first
// one on ties, thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1260) until a leaf = its word.
//
// k_voc_words: one thread per feature, the query in 8 registers, every child descriptor as two 16-byte loads and eight
// v_bcnt; the vocabulary (ORBvoc: 1 082 073 nodes = 35 MB of descriptors) stays resident in HBM for the life of the
// object.  The BowVector itself (a sorted map with weights summed in feature order and an L1 normalisation in word order,
// TemplatedVocabulary.h:1065-1124, BowVector.cpp:62-84) is a few hundred entries and is assembled on the host in
// exactly that order; L1Scoring::score (ScoringObject.cpp:23-68) is a host function.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "ctx.hpp"

struct ssx_vocabulary {
  ssx_ctx* ctx = nullptr;
  int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0;
  DevBuf arena, io;
  HostBuf stage;
  const uint8_t* d_desc = nullptr;      // [n_nodes][32]
  const double* d_weight = nullptr;     // [n_nodes]
  const int32_t* d_child_ptr = nullptr; // [n_nodes + 1]
  const int32_t* d_child = nullptr;     // [n_nodes - 1] children of every node, id order
  const int32_t* d_word = nullptr;      // [n_nodes] word id of a leaf, -1 otherwise
};

namespace {

__global__ __launch_bounds__(256) void k_voc_words(const uint8_t* desc, const double* weight, const int32_t* child_ptr, const int32_t* child,
                                                   const int32_t* word_of, const uint8_t* feat, int n, int32_t* word_out, double* weight_out)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const uint4* q4 = reinterpret_cast<const uint4*>(feat + 32 * (size_t)f);
  const uint4 qa = q4[0], qb = q4[1];
  int id = 0;
  int c0 = child_ptr[0], c1 = child_ptr[1];
  if (c1 == c0) { word_out[f] = -1; weight_out[f] = 0.0; return; }
  while (c1 > c0) {
    int best = 0x7fffffff, bid = 0;
    for (int c = c0; c < c1; ++c) {
      const int node = child[c];
      const uint4* d4 = reinterpret_cast<const uint4*>(desc + 32 * (size_t)node);
      const uint4 da = d4[0], db = d4[1];
      const int dist = __popc(qa.x ^ da.x) + __popc(qa.y ^ da.y) + __popc(qa.z ^ da.z) + __popc(qa.w ^ da.w) + __popc(qb.x ^ db.x) +
                       __popc(qb.y ^ db.y) + __popc(qb.z ^ db.z) + __popc(qb.w ^ db.w);
      if (dist < best) { best = dist; bid = node; }            // strict '<': the first child with the minimum wins
    }
    id = bid;
    c0 = child_ptr[id]; c1 = child_ptr[id + 1];
  }
  word_out[f] = word_of[id];
  weight_out[f] = weight[id];
}

}  // namespace

extern "C" {

ssx_status ssx_voc_create(ssx_ctx* ctx, int32_t k, int32_t L, int32_t scoring, int32_t weighting, int32_t n_nodes, const int32_t* parent,
                          const uint8_t* is_leaf, const uint8_t* desc, const double* weight, ssx_vocabulary** out)
{
  if (!ctx || !out || n_nodes < 1 || (n_nodes > 1 && (!parent || !is_leaf || !desc || !weight))) return SSX_ERR_INVALID_ARG;
  *out = nullptr;
  if (scoring != 0) { ctx->set_error("ssx_voc: only L1_NORM scoring (0) is supported, got %d", scoring); return SSX_ERR_UNSUPPORTED; }
  if (weighting < 0 || weighting > 3) { ctx->set_error("ssx_voc: weighting %d (0 TF_IDF, 1 TF, 2 IDF, 3 BINARY)", weighting); return SSX_ERR_INVALID_ARG; }
  std::vector<int32_t> cnt(n_nodes + 1, 0), word(n_nodes, -1);
  int n_words = 0;
  for (int i = 1; i < n_nodes; ++i) {
    if (parent[i] < 0 || parent[i] >= i) { ctx->set_error("ssx_voc: node %d has parent %d (a parent precedes its children)", i, parent[i]); return SSX_ERR_INVALID_ARG; }
    cnt[parent[i] + 1]++;
    if (is_leaf[i]) word[i] = n_words++;
  }
  for (int i = 0; i < n_nodes; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> child(std::max(n_nodes - 1, 1), 0), fill(cnt.begin(), cnt.end() - 1);
  for (int i = 1; i < n_nodes; ++i) child[fill[parent[i]]++] = i;
  for (int i = 1; i < n_nodes; ++i)
    if ((cnt[i + 1] == cnt[i]) != (is_leaf[i] != 0)) { ctx->set_error("ssx_voc: node %d: leaf flag and children disagree", i); return SSX_ERR_INVALID_ARG; }
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ssx_vocabulary* v = new ssx_vocabulary();
  v->ctx = ctx; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes; v->n_words = n_words;
  Layout lay;
  const size_t o_desc = lay.take((size_t)32 * n_nodes), o_w = lay.take(sizeof(double) * (size_t)n_nodes);
  const size_t o_cp = lay.take(sizeof(int32_t) * ((size_t)n_nodes + 1)), o_c = lay.take(sizeof(int32_t) * child.size());
  const size_t o_word = lay.take(sizeof(int32_t) * (size_t)n_nodes);
  hipError_t e = v->arena.reserve(lay.off);
  if (e != hipSuccess) { delete v; ctx->set_error("ssx_voc: device allocation of %zu bytes failed", lay.off); return SSX_ERR_HIP; }
  char* base = v->arena.as<char>();
  std::vector<uint8_t> d0(32, 0);
  std::vector<double> w0(n_nodes, 0.0);
  if (n_nodes > 1) memcpy(w0.data() + 1, weight + 1, sizeof(double) * (size_t)(n_nodes - 1));
  e = hipMemcpy(base + o_desc, d0.data(), 32, hipMemcpyHostToDevice);
  if (e == hipSuccess && n_nodes > 1) e = hipMemcpy(base + o_desc + 32, desc + 32, (size_t)32 * (n_nodes - 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(base + o_w, w0.data(), sizeof(double) * (size_t)n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(base + o_cp, cnt.data(), sizeof(int32_t) * ((size_t)n_nodes + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(base + o_c, child.data(), sizeof(int32_t) * child.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(base + o_word, word.data(), sizeof(int32_t) * (size_t)n_nodes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { v->arena.release(); delete v; ctx->set_error("ssx_voc: upload failed: %s", hipGetErrorString(e)); return SSX_ERR_HIP; }
  v->d_desc = (const uint8_t*)(base + o_desc); v->d_weight = (const double*)(base + o_w);
  v->d_child_ptr = (const int32_t*)(base + o_cp); v->d_child = (const int32_t*)(base + o_c); v->d_word = (const int32_t*)(base + o_word);
  *out = v;
  return SSX_OK;
}

// TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1337-1420): "k L scoring weighting", then one node per line
ssx_status ssx_voc_load_text(ssx_ctx* ctx, const char* path, ssx_vocabulary** out)
{
  if (!ctx || !path || !out) return SSX_ERR_INVALID_ARG;
  *out = nullptr;
  std::ifstream f(path);
  if (!f.is_open()) { ctx->set_error("ssx_voc_load_text: cannot open %s", path); return SSX_ERR_INVALID_ARG; }
  std::string line;
  if (!std::getline(f, line)) { ctx->set_error("ssx_voc_load_text: %s is empty", path); return SSX_ERR_INVALID_ARG; }
  int k = -1, L = -1, n1 = -1, n2 = -1;
  { std::stringstream ss(line); ss >> k >> L >> n1 >> n2; }
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
    ctx->set_error("ssx_voc_load_text: %s is not a DBoW2 text vocabulary (header '%s')", path, line.c_str());
    return SSX_ERR_INVALID_ARG;
  }
  std::vector<int32_t> parent(1, -1);
  std::vector<uint8_t> leaf(1, 0), desc(32, 0);
  std::vector<double> weight(1, 0.0);
  while (std::getline(f, line)) {
    if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::stringstream ss(line);
    int pid = -1, is_leaf = 0;
    ss >> pid >> is_leaf;
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) { int b = -1; ss >> b; if (b < 0 || b > 255 || !ss) { ctx->set_error("ssx_voc_load_text: bad node line %zu", parent.size()); return SSX_ERR_INVALID_ARG; } d[i] = (uint8_t)b; }
    double w = 0.0;
    ss >> w;
    if (!ss) { ctx->set_error("ssx_voc_load_text: bad node line %zu", parent.size()); return SSX_ERR_INVALID_ARG; }
    parent.push_back(pid); leaf.push_back(is_leaf > 0 ? 1 : 0); weight.push_back(w);
    desc.insert(desc.end(), d, d + 32);
  }
  return ssx_voc_create(ctx, k, L, n1, n2, (int32_t)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

void ssx_voc_destroy(ssx_vocabulary* v)
{
  if (!v) return;
  v->arena.release(); v->io.release(); v->stage.release();
  delete v;
}

ssx_status ssx_voc_info(const ssx_vocabulary* v, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words, int32_t* weighting)
{
  if (!v) return SSX_ERR_INVALID_ARG;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->n_nodes;
  if (n_words) *n_words = v->n_words;
  if (weighting) *weighting = v->weighting;
  return SSX_OK;
}

ssx_status ssx_voc_transform(ssx_vocabulary* v, const uint8_t* desc, int32_t n, int32_t* words_out, double* weights_out, int32_t cap,
                             int32_t* ids_out, double* vals_out, int32_t* n_entries)
{
  if (!v || n < 0 || (n > 0 && !desc) || cap < 0 || (cap > 0 && (!ids_out || !vals_out))) return SSX_ERR_INVALID_ARG;
  ssx_ctx* ctx = v->ctx;
  if (n_entries) *n_entries = 0;
  if (n == 0) return SSX_OK;
  if (v->n_nodes <= 1) {                                    // empty vocabulary: an empty BowVector (TemplatedVocabulary.h:1071-1074)
    for (int32_t i = 0; i < n; ++i) {                       // and the documented per-feature outputs: word -1, weight 0
      if (words_out) words_out[i] = -1;
      if (weights_out) weights_out[i] = 0.0;
    }
    return SSX_OK;
  }
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  Layout lay;
  const size_t o_f = lay.take((size_t)32 * n);
  const size_t in_bytes = lay.off;
  const size_t o_word = lay.take(sizeof(int32_t) * (size_t)n), o_w = lay.take(sizeof(double) * (size_t)n);
  SSX_HIP_TRY(ctx, v->io.reserve(lay.off));
  SSX_HIP_TRY(ctx, v->stage.reserve(lay.off));
  char* hs = v->stage.as<char>();
  char* db = v->io.as<char>();
  memcpy(hs + o_f, desc, (size_t)32 * n);
  SSX_HIP_TRY(ctx, hipMemcpyAsync(db, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_voc_words, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, v->d_desc, v->d_weight, v->d_child_ptr, v->d_child, v->d_word,
                     (const uint8_t*)(db + o_f), n, (int32_t*)(db + o_word), (double*)(db + o_w));
  SSX_HIP_TRY(ctx, hipGetLastError());
  SSX_HIP_TRY(ctx, hipMemcpyAsync(hs + o_word, db + o_word, lay.off - o_word, hipMemcpyDeviceToHost, ctx->stream));
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t* word = reinterpret_cast<const int32_t*>(hs + o_word);
  const double* w = reinterpret_cast<const double*>(hs + o_w);
  if (words_out) memcpy(words_out, word, sizeof(int32_t) * (size_t)n);
  if (weights_out) memcpy(weights_out, w, sizeof(double) * (size_t)n);
  // BowVector: TF / TF_IDF add the weight per occurrence, IDF / BINARY keep the first; stopped words (weight 0) are skipped;
  // L1 scoring normalises (TemplatedVocabulary.h:1083-1124, BowVector.cpp:62-84): accumulation in feature order, norm in word order
  std::map<int32_t, double> bow;
  for (int f = 0; f < n; ++f) {
    if (!(w[f] > 0) || word[f] < 0) continue;
    auto it = bow.lower_bound(word[f]);
    const bool have = it != bow.end() && it->first == word[f];
    if (v->weighting == 0 || v->weighting == 1) {
      if (have) it->second += w[f]; else bow.insert(it, {word[f], w[f]});
    } else if (!have) {
      bow.insert(it, {word[f], w[f]});
    }
  }
  double norm = 0.0;
  for (auto& kv : bow) norm += std::fabs(kv.second);
  if (norm > 0.0)
    for (auto& kv : bow) kv.second /= norm;
  if (n_entries) *n_entries = (int32_t)bow.size();
  if ((int64_t)bow.size() > cap) {
    if (cap == 0 && !ids_out) return SSX_OK;               // the caller only asked for the per-feature words / the size
    ctx->set_error("ssx_voc_transform: %zu words but capacity %d", bow.size(), cap);
    return SSX_ERR_CAPACITY;
  }
  int k = 0;
  for (auto& kv : bow) { ids_out[k] = kv.first; vals_out[k] = kv.second; ++k; }
  return SSX_OK;
}

// L1Scoring::score (ScoringObject.cpp:23-68) on two BowVectors given as sorted (id, value) arrays
double ssx_bow_score_l1(int32_t n1, const int32_t* id1, const double* v1, int32_t n2, const int32_t* id2, const double* v2)
{
  if (n1 < 0 || n2 < 0 || (n1 > 0 && (!id1 || !v1)) || (n2 > 0 && (!id2 || !v2))) return 0.0;
  double score = 0.0;
  int i = 0, j = 0;
  while (i < n1 && j < n2) {
    if (id1[i] == id2[j]) {
      score += std::fabs(v1[i] - v2[j]) - std::fabs(v1[i]) - std::fabs(v2[j]);
      ++i; ++j;
    } else if (id1[i] < id2[j]) {
      while (i < n1 && id1[i] < id2[j]) ++i;
    } else {
      while (j < n2 && id2[j] < id1[i]) ++j;
    }
  }
  return -score / 2.0;
}

}  // extern "C"
