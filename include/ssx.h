/* include/ssx.h -- C ABI of libssx.so, the MI355X (gfx950) compute core that drops in behind
 * ssvio's ORBextractor / FrontEnd / Backend class surfaces.
 *
 * The reference (weihaoysgs/ssvio, /root/reference) has no plugin/FFI layer: its seams are the
 * non-virtual C++ methods listed per entry point below.  A maintainer
 * replaces the BODY of each cited method with a marshalling call into this ABI (INTEGRATION.md shows
 * the stubs; include/ssx_shim.hpp ships ready-made C++ wrappers with the reference's method names).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  All array arguments are HOST pointers
 *     unless the parameter name ends in _dev (device pointers, for callers that keep data in HBM).
 *   - every function returns an ssx_status (0 = OK, <0 = error) and never aborts or throws;
 *     ssx_last_error(ctx) returns a human-readable message for the last failure on that ctx.
 *   - the caller owns every buffer; outputs are caller-allocated with a capacity + *count.
 *   - one ssx_ctx = one GPU + one HIP stream.  A ctx is single-threaded; different ctxs may be used
 *     concurrently from different threads (the reference's front-end thread and backend thread each
 *     hold their own; multi-GPU = one ctx per device, one process per GPU under torch.distributed).
 *   - pose layout = Sophus::SE3d::data(): qx qy qz qw tx ty tz (T_cw, world -> camera).
 *   - there is NO CPU fallback: if no gfx950 device is present ssx_ctx_create fails with
 *     SSX_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef SSX_H
#define SSX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSX_VERSION 100 /* 0.1.0 */
#if defined(__GNUC__)
#define SSX_API __attribute__((visibility("default")))
#else
#define SSX_API
#endif

typedef enum {
  SSX_OK = 0,
  SSX_ERR_INVALID_ARG = -1,
  SSX_ERR_NO_DEVICE = -2,
  SSX_ERR_HIP = -3,        /* a HIP runtime call failed; see ssx_last_error */
  SSX_ERR_CAPACITY = -4,   /* an output capacity was too small; *count holds the needed size */
  SSX_ERR_UNSUPPORTED = -5,
  SSX_ERR_COMM = -6        /* the all-reduce callback reported a failure */
} ssx_status;

typedef struct ssx_ctx ssx_ctx;

/* Construction parameters the reference reads from its YAML through Setting::Get
 * (src/ssvio/system.cpp:59-70,117-128; config/kitti_00.yaml). */
typedef struct {
  int device;          /* HIP device ordinal */
  void* stream;        /* hipStream_t to run on, or NULL to create a private non-blocking stream */
  int max_width, max_height; /* largest image the ctx will see (scratch sizing); 0 = grow on demand */
} ssx_config;

SSX_API int ssx_version(void);
SSX_API int ssx_device_count(void);
SSX_API ssx_status ssx_ctx_create(const ssx_config* cfg, ssx_ctx** out);
SSX_API void ssx_ctx_destroy(ssx_ctx* ctx);
SSX_API const char* ssx_last_error(const ssx_ctx* ctx);
SSX_API ssx_status ssx_ctx_synchronize(ssx_ctx* ctx);
/* hipStream_t the ctx enqueues on (for callers that bracket calls with their own events). */
SSX_API void* ssx_ctx_stream(ssx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Local bundle adjustment -- replaces the body of Backend::OptimizeActiveMap
 * (src/ssvio/backend.cpp:78-245): g2o BlockSolver_6_3 + LinearSolverCSparse + Levenberg-Marquardt
 * with EdgeProjection / VertexPose / VertexXYZ (include/ssvio/g2otypes.hpp:28-65,112-162), Huber kernel,
 * landmarks marginalised (Schur complement, thirdparty/g2o/g2o/core/block_solver.hpp:315-447).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t P;                 /* poses (active keyframes, backend.cpp:88-103) */
  const double* poses;       /* P x 7 */
  const uint8_t* pose_fixed; /* P, nullable (reference: none fixed) */
  int32_t L;                 /* landmarks (backend.cpp:113-132) */
  const double* points;      /* L x 3 */
  const uint8_t* point_fixed;/* L, nullable (backend.cpp:125-130) */
  int32_t E;                 /* observations / EdgeProjection edges (backend.cpp:135-168) */
  const int32_t* edge_pose;  /* E, index into poses */
  const int32_t* edge_point; /* E, index into points */
  const double* edge_uv;     /* E x 2 measured pixel (cv::Point2f widened to double, backend.cpp:80,160) */
  const uint8_t* edge_cam;   /* E, 0 = left extrinsic, 1 = right (backend.cpp:148-155); nullable = all 0 */
  double K[4];               /* fx fy cx cy (Camera::getK) */
  double cam_ext[14];        /* 2 x 7: left / right camera extrinsic (Camera::getPose, system.cpp:63,71) */
} ssx_ba_problem;

typedef enum { SSX_JAC_ANALYTIC = 0, SSX_JAC_NUMERIC_G2O = 1 } ssx_jac_mode;

/* Sum-all-reduce hook for landmark-sharded multi-GPU BA.  `buf_dev` is a DEVICE pointer to `count`
 * doubles on the ctx's device; the callee must enqueue an in-place sum over all ranks ordered after
 * work already enqueued on `stream` and make later work on `stream` wait for it (torch.distributed
 * all_reduce on a tensor aliasing buf_dev does exactly that when the ctx runs on torch's current
 * stream).  Return 0 on success. */
typedef int (*ssx_allreduce_fn)(void* user, double* buf_dev, size_t count, void* stream);

typedef struct {
  int32_t outer_rounds;   /* backend.cpp:175  while (iteration < 5)            default 5     */
  int32_t iters;          /* backend.cpp:178  optimizer.optimize(10)           default 10    */
  double chi2_th;         /* backend.cpp:109                                    default 5.891 */
  double huber_delta;     /* backend.cpp:163  rk->setDelta(chi2_th)             default 5.891 */
  double inlier_ratio;    /* backend.cpp:195                                    default 0.7   */
  int32_t jac_mode;       /* ssx_jac_mode; the reference uses g2o's numeric Jacobians
                             (linearizeOplus is commented out, g2otypes.hpp:133-153)           */
  /* multi-GPU: this rank holds a landmark shard (all edges of its landmarks, every pose replicated);
   * allreduce sums the reduced pose system and the chi2/scale scalars across ranks.  NULL = 1 GPU. */
  ssx_allreduce_fn allreduce;
  void* allreduce_user;
  int32_t rank, world_size; /* of this shard; world_size <= 1 = single GPU */
} ssx_ba_options;

#define SSX_BA_MAX_STATS 128
typedef struct {
  double* poses_out;        /* P x 7, nullable */
  double* points_out;       /* L x 3, nullable */
  double* edge_chi2;        /* E, nullable: edge->chi2() as backend.cpp:185,209 reads it */
  uint8_t* edge_outlier;    /* E, nullable: chi2 > chi2_th (backend.cpp:209-227) */
  int32_t rounds;           /* outer rounds executed */
  int32_t n_iters;          /* LM iterations executed over all rounds (<= SSX_BA_MAX_STATS recorded) */
  double iter_chi2[SSX_BA_MAX_STATS];   /* robust chi2 after each LM iteration */
  double iter_lambda[SSX_BA_MAX_STATS]; /* lambda after each LM iteration */
  int32_t iter_trials[SSX_BA_MAX_STATS];/* LM trials of each iteration */
  int32_t n_inliers, n_outliers;        /* of the last round (backend.cpp:181-194) */
  /* phase timing of this call, milliseconds of GPU time (HIP events on the ctx stream); the schema
   * follows g2o's G2OBatchStatistics (thirdparty/g2o/g2o/core/batch_stats.h) */
  float ms_total, ms_setup;
} ssx_ba_result;

SSX_API void ssx_ba_default_options(ssx_ba_options* opt);
SSX_API ssx_status ssx_ba_solve(ssx_ctx* ctx, const ssx_ba_problem* prob, const ssx_ba_options* opt,
                        ssx_ba_result* res);

/* One linearisation of the problem at its current state (no update): the blocks the kernels build,
 * for kernel-level parity tests and profiling.  Any output may be NULL.
 *   Hpp P x 36 (row-major 6x6), bp P x 6, Hll L x 9, bl L x 3, Hpl E x 18 (6x3 row-major, per edge),
 *   err E x 2, chi2 = robust chi2.  Rows of fixed vertices are zero. */
SSX_API ssx_status ssx_ba_linearize(ssx_ctx* ctx, const ssx_ba_problem* prob, double huber_delta, int32_t jac_mode,
                            double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* err,
                            double* chi2);

#ifdef __cplusplus
}
#endif
#endif /* SSX_H */
