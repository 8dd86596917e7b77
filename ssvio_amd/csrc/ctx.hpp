// ssvio_amd/csrc/ctx.hpp -- ssx_ctx: one GPU, one HIP stream, grow-only scratch arenas.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssx.h"

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
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
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
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
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

struct BaWorkspace;   // ba.hip
struct OrbWorkspace;  // orb.hip

struct ssx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  char err[512] = {0};
  BaWorkspace* ba = nullptr;
  OrbWorkspace* orb = nullptr;
  void (*ba_free)(BaWorkspace*) = nullptr;    // set by the module that allocates the workspace
  void (*orb_free)(OrbWorkspace*) = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  void set_error(const char* fmt, ...)
  {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, sizeof(err), fmt, ap);
    va_end(ap);
  }
};

