// oracle/src/voc_oracle.cpp -- TEST INFRASTRUCTURE (CPU restatement, never used by the product).
//
// Bag-of-words side of the loop closing: ssvio turns a keyframe's ORB descriptors into a
// DBoW2 BowVector (loopclosing.cpp:633 `dbow2_vocabulary_->transform(desc, bow2_vec_)`) and ranks database keyframes
// with `dbow2_vocabulary_->score(a, b)` (loopclosing.cpp:84).  ORBVocabulary = TemplatedVocabulary<FORB::TDescriptor,
// FORB> (include/ssvio/orbvocabulary.hpp:10).  Restated from the vendored sources:
//   tree descent of one descriptor   thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1260  (first child with the
//                                    smallest Hamming distance wins: strict '<' over the children in file order)
//   FORB::distance                   thirdparty/DBoW2/DBoW2/FORB.cpp:81-101  (Hamming over 256 bits)
//   transform of a descriptor set    TemplatedVocabulary.h:1065-1124  (TF / TF_IDF: addWeight; IDF / BINARY:
//                                    addIfNotExist; words of weight 0 are "stopped"; then normalise when the scoring needs it)
//   BowVector::normalize (L1)        thirdparty/DBoW2/DBoW2/BowVector.cpp:62-84
//   L1Scoring::score                 thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68
//   text vocabulary                  TemplatedVocabulary.h:1337-1420 (loadFromTextFile: "k L scoring weighting", then one
//                                    line per node: parent isLeaf d0..d31 weight; node ids in line order from 1, word ids in
//                                    leaf order)
// PARITY UNPINNED vs the real DBoW2: it cannot be compiled here (FORB is written on cv::Mat, OpenCV is absent) and the
// reference holds no test or fixture for it; pinned by hand-checkable known answers (tests/test_oracle_voc.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "oracle.h"

namespace {

int hamming256(const uint8_t* a, const uint8_t* b)
{
  int d = 0;
  for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

}  // namespace

extern "C" {

// Vocabulary as flat arrays: node 0 is the root; node i >= 1: parent[i], is_leaf[i], desc[i][32], weight[i] (entries 0
// unused).  Children of a node are its nodes in increasing id order.  word_of[i] = word id of leaf i (leaf order), -1 else.
// Returns the number of words, or -1 on a malformed tree.
int orc_voc_words(int n_nodes, const int32_t* parent, const uint8_t* is_leaf, int32_t* word_of)
{
  int words = 0;
  for (int i = 0; i < n_nodes; ++i) word_of[i] = -1;
  for (int i = 1; i < n_nodes; ++i) {
    if (parent[i] < 0 || parent[i] >= i) return -1;          // loadFromTextFile appends children to an existing parent
    if (is_leaf[i]) word_of[i] = words++;
  }
  return words;
}

// per-descriptor word id and weight (weight 0 = stopped word, -1 = no vocabulary)
void orc_voc_transform_features(int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                                const uint8_t* feat, int n, int32_t* word_out, double* weight_out)
{
  std::vector<std::vector<int>> children(n_nodes);
  std::vector<int32_t> word_of(n_nodes);
  orc_voc_words(n_nodes, parent, is_leaf, word_of.data());
  for (int i = 1; i < n_nodes; ++i) children[parent[i]].push_back(i);
  for (int f = 0; f < n; ++f) {
    int id = 0;
    if (children[0].empty()) { word_out[f] = -1; weight_out[f] = 0.0; continue; }
    do {
      const std::vector<int>& ch = children[id];
      id = ch[0];
      int best = hamming256(feat + 32 * (size_t)f, desc + 32 * (size_t)id);
      for (size_t c = 1; c < ch.size(); ++c) {
        const int dd = hamming256(feat + 32 * (size_t)f, desc + 32 * (size_t)ch[c]);
        if (dd < best) { best = dd; id = ch[c]; }
      }
    } while (!children[id].empty());
    word_out[f] = word_of[id];
    weight_out[f] = weight[id];
  }
}

// BowVector of a descriptor set: sorted unique word ids with their values; weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY;
// L1 scoring (the vector is L1-normalised).  Returns the number of entries (<= cap).
int orc_bow_vector(int n, const int32_t* word, const double* weight, int weighting, int cap, int32_t* ids_out, double* vals_out)
{
  std::map<int32_t, double> v;
  for (int f = 0; f < n; ++f) {
    if (!(weight[f] > 0) || word[f] < 0) continue;            // stopped
    auto it = v.lower_bound(word[f]);
    const bool have = it != v.end() && it->first == word[f];
    if (weighting == 0 || weighting == 1) {                   // addWeight
      if (have) it->second += weight[f]; else v.insert(it, {word[f], weight[f]});
    } else if (!have) {
      v.insert(it, {word[f], weight[f]});                     // addIfNotExist
    }
  }
  double norm = 0.0;
  for (auto& kv : v) norm += std::fabs(kv.second);
  if (norm > 0.0)
    for (auto& kv : v) kv.second /= norm;
  int k = 0;
  for (auto& kv : v) {
    if (k >= cap) break;
    ids_out[k] = kv.first; vals_out[k] = kv.second; ++k;
  }
  return (int)v.size();
}

double orc_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2)
{
  double score = 0.0;
  int i = 0, j = 0;
  while (i < n1 && j < n2) {
    if (id1[i] == id2[j]) {
      score += std::fabs(v1[i] - v2[j]) - std::fabs(v1[i]) - std::fabs(v2[j]);
      ++i; ++j;
    } else if (id1[i] < id2[j]) {
      while (i < n1 && id1[i] < id2[j]) ++i;                   // lower_bound
    } else {
      while (j < n2 && id2[j] < id1[i]) ++j;
    }
  }
  return -score / 2.0;
}

}  // extern "C"
