// ssvio_amd/csrc/ctx.hpp -- ssx_ctx: one GPU, one HIP stream, grow-only scratch arenas.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ssx.h"

// Wrap a kernel launch: when profiling is on, bracket it with two HIP events on the ctx stream.
#define SSX_PROF(ctx, kid, stmt)                                              \
  do {                                                                        \
    if ((ctx)->prof.on) {                                                     \
      SsxProf::Rec _r{(kid), (ctx)->prof.get(), (ctx)->prof.get()};           \
      (void)hipEventRecord(_r.a, (ctx)->stream);                              \
      stmt;                                                                   \
      (void)hipEventRecord(_r.b, (ctx)->stream);                              \
      (ctx)->prof.recs.push_back(_r);                                         \
    } else {                                                                  \
      stmt;                                                                   \
    }                                                                         \
  } while (0)

#define SSX_PROF_ON(ctx, strm, kid, stmt)                                     \
  do {                                                                        \
    if ((ctx)->prof.on) {                                                     \
      SsxProf::Rec _r{(kid), (ctx)->prof.get(), (ctx)->prof.get()};           \
      (void)hipEventRecord(_r.a, (strm));                                     \
      stmt;                                                                   \
      (void)hipEventRecord(_r.b, (strm));                                     \
      (ctx)->prof.recs.push_back(_r);                                         \
    } else {                                                                  \
      stmt;                                                                   \
    }                                                                         \
  } while (0)

#define SSX_HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));    \
      return SSX_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)

// A device buffer that only ever grows (288 GB of HBM: keep scratch resident between calls instead of
// paying hipMalloc/hipFree on the hot path).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // (grow = what a reallocation asks for, in units of the request: a hipFree synchronises the whole device)
  hipError_t reserve(size_t bytes, double grow = 1.25)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = (size_t)((double)bytes * grow) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Pinned host staging buffer (async H2D/D2H without the runtime's bounce copy).
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, double grow = 1.25)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    size_t want = (size_t)((double)bytes * grow) + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// bump allocator for carving one arena into 256-byte aligned sub-buffers
struct Layout {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};

// ---- per-kernel timing with HIP events on the ctx stream (ssx_profile_begin / ssx_profile_end) ----
enum SsxKernelId {
  KID_BA_LINEARIZE = 0, KID_BA_REDUCE_LIN, KID_BA_SCHUR, KID_BA_REDUCE_SCHUR, KID_BA_SOLVE, KID_BA_BACKSUB,
  KID_BA_REDUCE_TRIAL, KID_ORB_RESIZE, KID_ORB_FAST, KID_ORB_OCTREE, KID_ORB_ORIENT, KID_ORB_GAUSS, KID_ORB_BRIEF,
  KID_ORB_MISC, KID_ST_BUCKET, KID_ST_MATCH, KID_ST_TRIANGULATE, KID_ST_MISC, KID_POSE_ONLY, KID_BA_COMM, KID_BA_LIN_SCHUR, KID_COUNT
};
static const char* const kSsxKernelNames[KID_COUNT] = {
  "k_linearize", "k_reduce_lin", "k_schur", "k_reduce_schur", "k_solve", "k_backsub_residual", "k_reduce_trial",
  "k_resize", "k_fast_cells", "k_octree", "k_orient", "k_gauss7", "k_orient_brief", "orb_misc", "k_row_bucket", "k_match",
  "k_triangulate_matches", "stereo_misc", "k_pose_only", "ba_allreduce", "k_lin_schur"};

struct SsxProf {
  bool on = false;
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<Rec> recs;
  hipEvent_t get()
  {
    if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
  }
};

struct BaWorkspace;   // ba.hip
struct OrbWorkspace;  // orb.hip

struct ssx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  uint32_t cu_mask[16] = {0};                // ssx_config.cu_first / cu_count as a bit mask; cu_mask_words == 0: no mask
  int cu_mask_words = 0;
  // every stream the ctx creates: CU-masked when the ctx is, lowest priority on request
  hipError_t make_stream(hipStream_t* out, bool low_priority)
  {
    if (cu_mask_words > 0) return hipExtStreamCreateWithCUMask(out, (uint32_t)cu_mask_words, cu_mask);
    if (low_priority) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (hipStreamCreateWithPriority(out, hipStreamNonBlocking, least) == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  char err[512] = {0};
  BaWorkspace* ba = nullptr;
  OrbWorkspace* orb = nullptr;
  void (*ba_free)(BaWorkspace*) = nullptr;    // set by the module that allocates the workspace
  void (*orb_free)(OrbWorkspace*) = nullptr;
  void* lk = nullptr;                        // lk.hip workspace
  void (*lk_free)(void*) = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_spec = nullptr;
  hipStream_t aux = nullptr;                 // second stream: independent stages overlap (blur || detect)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pyr = nullptr, ev_fast0 = nullptr;
  hipStream_t grp[3] = {nullptr, nullptr, nullptr};   // batched BA: the groups of windows beside the main stream (created on first use)
  hipEvent_t grp_ev[3] = {nullptr, nullptr, nullptr};
  int ba_batch_groups = 0;                   // ssx_ba_set_batch_groups: groups of windows of this ctx's one-shot batched solves (0: the default)
  SsxProf prof;
  DevBuf po_arena;                           // pose-only optimisation scratch
  HostBuf po_stage;

  std::mutex err_mu;                         // worker threads of a batched call may report failures concurrently

  void set_error(const char* fmt, ...)
  {
    std::lock_guard<std::mutex> lk(err_mu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, sizeof(err), fmt, ap);
    va_end(ap);
  }
};

