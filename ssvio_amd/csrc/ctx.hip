// ssvio_amd/csrc/ctx.hip -- context lifetime for libssx.so (include/ssx.h).
#include "ctx.hpp"

extern "C" {

int ssx_version(void) { return SSX_VERSION; }

ssx_status ssx_abi_check(int header_version, size_t sizeof_config, size_t sizeof_ba_problem, size_t sizeof_ba_options, size_t sizeof_ba_result,
                         size_t sizeof_window_update)
{
  return (header_version == SSX_VERSION && sizeof_config == sizeof(ssx_config) && sizeof_ba_problem == sizeof(ssx_ba_problem) &&
          sizeof_ba_options == sizeof(ssx_ba_options) && sizeof_ba_result == sizeof(ssx_ba_result) &&
          sizeof_window_update == sizeof(ssx_ba_window_update)) ? SSX_OK : SSX_ERR_UNSUPPORTED;
}

int ssx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

ssx_status ssx_ctx_create(const ssx_config* cfg, ssx_ctx** out)
{
  if (!out) return SSX_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SSX_ERR_NO_DEVICE;  // no CPU fallback, by design
  const int dev = cfg ? cfg->device : 0;
  if (dev < 0 || dev >= n) return SSX_ERR_INVALID_ARG;
  if (hipSetDevice(dev) != hipSuccess) return SSX_ERR_HIP;
  ssx_ctx* c = new ssx_ctx();
  c->device = dev;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  if (cfg && cfg->cu_count > 0) {
    const int n_cu = c->num_cus > 0 ? c->num_cus : 256;
    if (cfg->cu_first < 0 || cfg->cu_first + cfg->cu_count > n_cu || n_cu > 512) { delete c; return SSX_ERR_INVALID_ARG; }
    c->cu_mask_words = (n_cu + 31) / 32;
    for (int i = cfg->cu_first; i < cfg->cu_first + cfg->cu_count; ++i) c->cu_mask[i >> 5] |= 1u << (i & 31);
  }
  if (cfg && cfg->stream) {
    c->stream = static_cast<hipStream_t>(cfg->stream);
    c->own_stream = false;
  } else {
    if (c->make_stream(&c->stream, false) != hipSuccess) {
      delete c;
      return SSX_ERR_HIP;
    }
    c->own_stream = true;
  }
  (void)hipEventCreate(&c->ev0);
  (void)hipEventCreate(&c->ev1);
  {
    // the auxiliary stream carries work that fills the gaps of the main stream's dependent chain: lowest priority
    if (c->make_stream(&c->aux, true) != hipSuccess) (void)hipGetLastError();
  }
  (void)hipEventCreateWithFlags(&c->ev_spec, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_pyr, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_fast0, hipEventDisableTiming);
  *out = c;
  return SSX_OK;
}

void ssx_ctx_destroy(ssx_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ba && ctx->ba_free) ctx->ba_free(ctx->ba);
  if (ctx->orb && ctx->orb_free) ctx->orb_free(ctx->orb);
  if (ctx->lk && ctx->lk_free) ctx->lk_free(ctx->lk);
  ctx->po_arena.release();
  ctx->po_stage.release();
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
  if (ctx->ev_spec) (void)hipEventDestroy(ctx->ev_spec);
  for (int g = 0; g < 3; ++g) {
    if (ctx->grp[g]) { (void)hipStreamSynchronize(ctx->grp[g]); (void)hipStreamDestroy(ctx->grp[g]); }
    if (ctx->grp_ev[g]) (void)hipEventDestroy(ctx->grp_ev[g]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_pyr) (void)hipEventDestroy(ctx->ev_pyr);
  if (ctx->ev_fast0) (void)hipEventDestroy(ctx->ev_fast0);
  for (hipEvent_t e : ctx->prof.pool) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* ssx_last_error(const ssx_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

ssx_status ssx_ctx_synchronize(ssx_ctx* ctx)
{
  if (!ctx) return SSX_ERR_INVALID_ARG;
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SSX_OK;
}

void* ssx_ctx_stream(ssx_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

// Page-locked host memory the GPU can read and write directly (fine-grained: coherent with the CPU).  What a caller hands to the
// entry points that take `images_on_device` (ssx_lk_track_batch): the level-0 kernel then reads the image over PCIe, nothing is staged.
void* ssx_host_alloc(size_t bytes)
{
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void ssx_host_free(void* p) { if (p) (void)hipHostFree(p); }

ssx_status ssx_profile_begin(ssx_ctx* ctx)
{
  if (!ctx) return SSX_ERR_INVALID_ARG;
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof.on = true;
  ctx->prof.used = 0;
  ctx->prof.recs.clear();
  return SSX_OK;
}

ssx_status ssx_profile_end(ssx_ctx* ctx, ssx_kernel_time* out, int32_t cap, int32_t* n)
{
  if (!ctx || !n) return SSX_ERR_INVALID_ARG;
  SSX_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof.on = false;
  double total[KID_COUNT] = {0};
  int calls[KID_COUNT] = {0};
  for (const auto& r : ctx->prof.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total[r.id] += ms; calls[r.id]++; }
  }
  int k = 0;
  for (int id = 0; id < KID_COUNT; ++id) {
    if (!calls[id]) continue;
    if (out && k < cap) {
      memset(&out[k], 0, sizeof(out[k]));
      strncpy(out[k].name, kSsxKernelNames[id], sizeof(out[k].name) - 1);
      out[k].calls = calls[id];
      out[k].total_ms = total[id];
    }
    ++k;
  }
  *n = k;
  ctx->prof.recs.clear();
  ctx->prof.used = 0;
  return SSX_OK;
}

}  // extern "C"
