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
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSX_VERSION 120 /* 0.1.20: + ssx_lk_track_batch, ssx_pose_only_opt_batch, ssx_host_alloc / _free; the kernel taps (ssx_*_stage_*,
                           * ssx_ba_linearize) moved to ssx_test_hooks.h, ssx_ba_batch_set_persistent (a no-op since round 4) removed.  0.1.10: ssx_config grew by cu_first / cu_count, ssx_ba_result by ms_comm, ssx_ba_window_update by the
                           * removal fields -- see ssx_abi_check */
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
  /* Partition the chip between contexts that run side by side (the front-end's ctx and the backend's): when cu_count > 0
   * every stream the ctx CREATES (stream == NULL above, its auxiliary and group streams) is restricted to cu_count
   * compute units starting at cu_first (hipExtStreamCreateWithCUMask; bit i of the mask lands on XCD i mod 8, so any
   * contiguous range is spread evenly over the eight XCDs).  0 / 0 = the whole chip.  Ignored for a caller's stream.
   * NOTE: hipExtStreamCreateWithCUMask takes no flags -- the streams of a masked ctx have default flags and normal priority:
   * they are not hipStreamNonBlocking (they synchronise implicitly with the legacy NULL stream, e.g. torch's default stream) and
   * the auxiliary stream loses the lowest priority an unmasked ctx gives it. */
  int cu_first, cu_count;
} ssx_config;

SSX_API int ssx_version(void);
/* The structs of this header carry no size field: a caller built against another version of it would hand the library a
 * shorter ssx_config (cu_count read from garbage) or receive ms_comm past the end of its ssx_ba_result.  Call this once after
 * loading the library with SSX_VERSION and the sizeof of the structs AS THE CALLER WAS COMPILED:
 *   ssx_abi_check(SSX_VERSION, sizeof(ssx_config), sizeof(ssx_ba_problem), sizeof(ssx_ba_options), sizeof(ssx_ba_result),
 *                 sizeof(ssx_ba_window_update))
 * SSX_OK when they are the library's, SSX_ERR_UNSUPPORTED otherwise (include/ssx_shim.hpp and ssvio_amd/_lib.py do it when they
 * load the library).  Every struct must be zero-initialised before its fields are set: later versions append fields whose
 * zero value means "as before". */
SSX_API ssx_status ssx_abi_check(int header_version, size_t sizeof_config, size_t sizeof_ba_problem, size_t sizeof_ba_options,
                                 size_t sizeof_ba_result, size_t sizeof_window_update);
SSX_API int ssx_device_count(void);
SSX_API ssx_status ssx_ctx_create(const ssx_config* cfg, ssx_ctx** out);
SSX_API void ssx_ctx_destroy(ssx_ctx* ctx);
SSX_API const char* ssx_last_error(const ssx_ctx* ctx);
SSX_API ssx_status ssx_ctx_synchronize(ssx_ctx* ctx);
/* hipStream_t the ctx enqueues on (for callers that bracket calls with their own events). */
SSX_API void* ssx_ctx_stream(ssx_ctx* ctx);
/* Page-locked (pinned, GPU-readable) host memory for callers without a HIP toolchain of their own: images kept in such a buffer can
 * be handed to the batched entry points with images_on_device = 1 (the kernels read them over PCIe, nothing is staged).  NULL when
 * the allocation fails.  (hipHostMalloc / hipHostFree.) */
SSX_API void* ssx_host_alloc(size_t bytes);
SSX_API void ssx_host_free(void* p);

/* Per-kernel GPU time, measured with HIP events recorded on the ctx stream around every kernel launch
 * between ssx_profile_begin and ssx_profile_end (the counterpart of g2o's G2OBatchStatistics; bench.py uses it
 * for the roofline of the dominant kernel).  Event pairs add a little launch overhead: time throughput with
 * profiling off. */
typedef struct {
  char name[48];
  int32_t calls;
  double total_ms;
} ssx_kernel_time;
SSX_API ssx_status ssx_profile_begin(ssx_ctx* ctx);
SSX_API ssx_status ssx_profile_end(ssx_ctx* ctx, ssx_kernel_time* out, int32_t cap, int32_t* n);

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
typedef enum { SSX_LARGE_SOLVER_AUTO = 0, SSX_LARGE_SOLVER_TILES = 1, SSX_LARGE_SOLVER_BAND = 2 } ssx_large_solver;

/* Sum-all-reduce hook for landmark-sharded multi-GPU BA.  `buf_dev` is a DEVICE pointer to `count`
 * doubles on the ctx's device; the callee must enqueue an in-place sum over all ranks ordered after
 * work already enqueued on `stream` and make later work on `stream` wait for it (torch.distributed
 * all_reduce on a tensor aliasing buf_dev does exactly that when the ctx runs on torch's current
 * stream).  Return 0 on success. */
typedef int (*ssx_allreduce_fn)(void* user, double* buf_dev, size_t count, void* stream);

/* RCCL inside the library (the reference is a single process, there is no counterpart): one
 * process per GPU, one communicator per process.  Rank 0 calls ssx_comm_unique_id and hands the 128 bytes to the other
 * ranks by whatever channel the host has (MPI, a file, torch.distributed); every rank then calls ssx_comm_init
 * (collective: ncclCommInitRank on the ctx's device).  ssx_comm_wrap adopts an ncclComm_t the host already owns.
 * With ssx_ba_options.comm set, ssx_ba_solve sums its exchange buffers with ncclAllReduce(f64, sum) enqueued on the
 * ctx stream -- no callback, no host round trip.  librccl.so.1 is bound at run time (the copy already loaded in the
 * process, else /opt/rocm/lib); SSX_ERR_COMM when it cannot be found. */
typedef struct ssx_comm ssx_comm;
typedef struct { char bytes[128]; } ssx_comm_id;      /* an ncclUniqueId */
SSX_API ssx_status ssx_comm_unique_id(ssx_ctx* ctx, ssx_comm_id* out);
SSX_API ssx_status ssx_comm_init(ssx_ctx* ctx, const ssx_comm_id* id, int32_t rank, int32_t world_size, ssx_comm** out);
SSX_API ssx_status ssx_comm_wrap(ssx_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world_size, ssx_comm** out);
SSX_API void ssx_comm_destroy(ssx_comm* comm);
SSX_API ssx_status ssx_comm_info(const ssx_comm* comm, int32_t* rank, int32_t* world_size);
/* in-place f64 sum of `count` doubles at the DEVICE pointer buf_dev, ordered on the ctx stream */
SSX_API ssx_status ssx_comm_allreduce_sum(ssx_ctx* ctx, ssx_comm* comm, double* buf_dev, size_t count);

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
  /* the native path: an ssx_comm (RCCL).  When set it takes precedence over `allreduce`, and rank / world_size are
   * the communicator's.  The callback above stays for hosts with their own collective layer and for tests. */
  ssx_comm* comm;
  int32_t collect_stats;    /* 1 = time every phase with HIP events (fills ssx_ba_result.ms_*; a few us per launch) */
  /* windows with more than 16 free keyframes, the reduced pose system: SSX_LARGE_SOLVER_AUTO picks the band solver
   * (sliding-window block Cholesky + nested dissection along the trajectory) when every keyframe shares landmarks only
   * with keyframes within 5 positions along the (possibly closed) trajectory (block bandwidth w <= 5) and the window
   * has at least 2 w + 2 free keyframes, else the 64x64-tile sparse Cholesky; _TILES / _BAND force one (BAND fails
   * with SSX_ERR_UNSUPPORTED on a wider co-visibility or a shorter window). */
  int32_t large_solver;
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
  /* GPU time of this call, milliseconds (HIP events on the ctx stream): upload .. last download */
  float ms_total, ms_setup;
  /* with ssx_ba_options.collect_stats: GPU time per phase summed over the call, the fields of g2o's
   * G2OBatchStatistics (thirdparty/g2o/g2o/core/batch_stats.h:38-68) they correspond to in brackets */
  float ms_linearize;       /* residuals + Jacobians + quadratic form   [timeResiduals + timeLinearize + timeQuadraticForm] */
  float ms_schur;           /* landmark elimination                     [timeSchurComplement]                               */
  float ms_linear_solution; /* reduced-system factorisation + solves    [timeLinearSolution]                                */
  float ms_update;          /* back-substitution, state update, trial residuals  [timeUpdate + the next computeActiveErrors] */
  float ms_reduce;          /* cross-chunk reductions                   [no g2o counterpart: single-threaded]               */
  float ms_comm;            /* the all-reduces of a landmark-sharded solve (enqueue to completion on the ctx stream)         */
} ssx_ba_result;

SSX_API void ssx_ba_default_options(ssx_ba_options* opt);
SSX_API ssx_status ssx_ba_solve(ssx_ctx* ctx, const ssx_ba_problem* prob, const ssx_ba_options* opt,
                        ssx_ba_result* res);

/* n windows in one call: what a batch of stereo pairs / a set of concurrent streams hands to its backends together.
 * Per window the arithmetic and the result are those of ssx_ba_solve (bit-identical); the kernels run once for all
 * windows (one grid dimension is the window), each window's device-driven LM loop advances on its own, one upload and
 * one download carry all of them, and the host marshalling runs on several threads.  Windows with more than 16 free
 * keyframes, a collective in `opt`, or n == 1 are solved one after the other.  `opt` applies to every window. */
SSX_API ssx_status ssx_ba_solve_batch(ssx_ctx* ctx, int32_t n, const ssx_ba_problem* probs, const ssx_ba_options* opt,
                                      ssx_ba_result* results);

/* A RESIDENT batch: marshalled and uploaded once, the windows stay in HBM; every ssx_ba_batch_solve optimises all of them
 * again from the uploaded state.  results may be NULL (nothing is downloaded; *lm_iterations_total, optional, still
 * counts the LM iterations run); with_edge_errors = 1 keeps what ssx_ba_result.edge_chi2 / edge_outlier need.
 * A batch belongs to its ctx and must be destroyed before it. */
typedef struct ssx_ba_batch ssx_ba_batch;
SSX_API ssx_status ssx_ba_batch_create(ssx_ctx* ctx, int32_t n, const ssx_ba_problem* probs, const ssx_ba_options* opt,
                                       int32_t with_edge_errors, ssx_ba_batch** out);
SSX_API ssx_status ssx_ba_batch_solve(ssx_ba_batch* batch, ssx_ba_result* results, int32_t* lm_iterations_total);
SSX_API int32_t ssx_ba_batch_size(const ssx_ba_batch* batch);
/* groups of windows ssx_ba_batch_solve runs side by side, each on its own stream (every batched kernel is launched once
   per group; 1 for fewer than 8 windows; SSX_BA_GROUPS in the environment overrides the default of 2) */
SSX_API int32_t ssx_ba_batch_groups(const ssx_ba_batch* batch);
/* 1 .. 4 groups for the following solves; 0 = back to the default (per-kernel profiles want 1: one launch per kernel and
   LM slot, nothing else on the chip beside it) */
SSX_API void ssx_ba_batch_set_groups(ssx_ba_batch* batch, int32_t groups);
SSX_API void ssx_ba_batch_destroy(ssx_ba_batch* batch);
/* Process-wide: the device phases of batched solves (ssx_ba_solve_batch, ssx_ba_batch_solve, ssx_ba_window_solve_batch) of DIFFERENT
 * contexts run one after the other on the device, in the order they were enqueued: the kernels of an LM round wait, on the device
 * (hipStreamWaitEvent), for the end of the round enqueued before it, whichever context that was.  Nothing is serialised on the host
 * but the enqueueing.  For servers that drive groups of windows from several host threads, one context each (the backend threads of
 * concurrent streams, backend.cpp:57-76): without turns the device phases of the groups interleave on the chip, end together, and
 * the groups' host phases (pending uploads, counting tables, unpacking) then coincide with the device idle; with turns one group's
 * host phase lies beside the other's kernels.  Off by default; no effect on results. */
SSX_API void ssx_ba_device_turns(int32_t enable);
/* Groups of windows (1 .. 4, each on a stream of its own; 0 = the default, 2 from 8 windows on) the ONE-SHOT batched solves of this ctx
 * -- ssx_ba_solve_batch, ssx_ba_window_solve_batch -- are run in, as ssx_ba_batch_set_groups does for a resident batch.  A server that
 * already drives several contexts side by side wants 1: the GPU has four hardware queues, and three backend contexts of one stream
 * each beside the front-end's measured 18 k stereo frames/s where three contexts of two streams each measured 14 k
 * (profiles/r05/live_backend_orchestration.md).  No effect on results. */
SSX_API ssx_status ssx_ba_set_batch_groups(ssx_ctx* ctx, int32_t groups);

/* ------------------------------------------------------------------------------------------------
 * A sliding local-BA window that STAYS in HBM.  Backend::OptimizeActiveMap (src/ssvio/backend.cpp:88-169) rebuilds its
 * graph from the active map at every keyframe, and that map changes by ONE keyframe between two optimisations
 * (Map::InsertKeyFrame, RemoveOldActiveKeyframe, RemoveOldActiveMapPoints: src/ssvio/map.cpp:27-56, 89-160).  The window
 * mirrors that: push a keyframe (its pose, the landmarks it introduces, its observations: the only data that crosses PCIe),
 * pop a keyframe (its observations go, and every landmark nobody observes any more), solve what is resident.  The device
 * sorts the observations by landmark itself; per solve the host counts observations per landmark (one pass over the
 * window) and uploads ~3.6 bytes per observation + 13 per landmark of tables.  Keyframes and landmarks are named by the
 * caller's 64-bit ids (KeyFrame::key_frame_id_, MapPoint::id_).  <= 16 free keyframes (the reference keeps 12:
 * config/kitti_00.yaml:30).  A window belongs to its ctx and must be destroyed before it.
 * ORDER: a solve gives its vertices the order of the caller's ids -- keyframes ascending by id, landmarks ascending by id, as
 * g2o orders its vertices and as a caller that re-marshals its map per keyframe does -- so its bits depend on what the window
 * holds, never on the storage slots it reused.  ssx_ba_window_export lists the window in that order, and the result of
 * ssx_ba_window_solve is, bit for bit, that of ssx_ba_solve on the exported problem.
 * BETWEEN TWO OPTIMISATIONS the reference changes its map by more than one keyframe in / one out (backend.cpp:205-244,
 * map.cpp:142-194): outlier observations are unlinked (ssx_ba_window_remove_flagged / _remove_observations), condemned map
 * points are deleted (ssx_ba_window_remove_landmarks), and a map point is FIXED while the keyframe of its first remaining
 * observation is outside the window (backend.cpp:125-130: ssx_ba_window_set_fix_rule).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ssx_ba_window ssx_ba_window;
SSX_API ssx_status ssx_ba_window_create(ssx_ctx* ctx, const ssx_ba_options* opt, const double* K4, const double* cam_ext14,
                                        ssx_ba_window** out);
SSX_API void ssx_ba_window_destroy(ssx_ba_window* win);
/* new_ids / new_xyz / new_fixed: the n_new landmarks this keyframe brings into the window (ids not yet in it);
 * obs_lm / obs_uv (n_obs x 2) / obs_cam (nullable = all left): its observations, of those or of landmarks already in
 * the window (backend.cpp:135-168).  Nothing is changed when the call fails. */
SSX_API ssx_status ssx_ba_window_push_keyframe(ssx_ba_window* win, int64_t kf_id, const double* pose7, int32_t pose_fixed,
                                               int32_t n_new, const int64_t* new_ids, const double* new_xyz,
                                               const uint8_t* new_fixed, int32_t n_obs, const int64_t* obs_lm,
                                               const double* obs_uv, const uint8_t* obs_cam);
/* The same for a caller that keeps, with every map point, the SLOT the window gave it (new_slots_out[i] = slot of the i-th
 * new landmark, valid until the landmark leaves the window): obs_slot[j] >= 0 = that slot, -1 - i = the i-th new landmark
 * of this call.  No id is looked up: a 2000-observation keyframe costs ~10 us of host time instead of ~70. */
SSX_API ssx_status ssx_ba_window_push_keyframe_slots(ssx_ba_window* win, int64_t kf_id, const double* pose7, int32_t pose_fixed,
                                                     int32_t n_new, const int64_t* new_ids, const double* new_xyz,
                                                     const uint8_t* new_fixed, int32_t* new_slots_out, int32_t n_obs,
                                                     const int32_t* obs_slot, const double* obs_uv, const uint8_t* obs_cam);
SSX_API ssx_status ssx_ba_window_pop_keyframe(ssx_ba_window* win, int64_t kf_id);
/* backend.cpp:205-227 (`if (ef.first->chi2() > chi2_th)`: mappoint->RemoveActiveObservation + RemoveObservation): unlink
 * observations; a landmark left without any leaves the window (Map::RemoveOldActiveMapPoints, and map.cpp:175-194 when the
 * reference deletes it altogether).
 *   _remove_flagged       flags[i] != 0 removes the i-th observation in the order of ssx_ba_window_export: the
 *                         ssx_ba_result.edge_outlier of the solve that just ran can be handed back as it is; n_obs must be the
 *                         window's observation count
 *   _remove_observations  the observations keyframe kf_id holds of the landmarks lm_ids[i] (camera cams[i] != 0 = right;
 *                         cams NULL = either); pairs the window does not hold are skipped, an unknown kf_id is an error
 * *n_removed (nullable) receives the number of observations removed. */
SSX_API ssx_status ssx_ba_window_remove_flagged(ssx_ba_window* win, int32_t n_obs, const uint8_t* flags, int32_t* n_removed);
SSX_API ssx_status ssx_ba_window_remove_observations(ssx_ba_window* win, int64_t kf_id, int32_t n, const int64_t* lm_ids,
                                                     const uint8_t* cams, int32_t* n_removed);
/* Map::RemoveAllOutlierMapPoints (map.cpp:175-194; the map points FrontEnd::EstimateCurrentPose, frontend.cpp:283-288, or the
 * backend condemned): the landmarks leave with all their observations.  Ids the window does not hold are skipped;
 * *n_removed (nullable) = landmarks removed. */
SSX_API ssx_status ssx_ba_window_remove_landmarks(ssx_ba_window* win, int32_t n, const int64_t* lm_ids, int32_t* n_removed);
/* rule 0 (default): a landmark is fixed when the caller says so (new_fixed, ssx_ba_window_set_landmark).
 * rule 1: backend.cpp:125-130 as well -- `mp->GetObservations().front()`'s keyframe is not active: a landmark is fixed while the
 * earliest-pushed keyframe among its remaining observations (MapPoint::observations_ keeps those of keyframes that left the
 * window, and loses those removed as outliers) is no longer in the window; ssx_ba_window_pop_keyframe and the removals keep
 * that up to date.  A landmark that comes BACK into a window it had left is pushed with new_fixed = what the caller's map says. */
SSX_API ssx_status ssx_ba_window_set_fix_rule(ssx_ba_window* win, int32_t rule);
/* One keyframe replaced in each of n windows of ONE ctx, in one call, the windows spread over the ctx's host threads (the
 * windows of concurrent streams all change at every keyframe; the edits are independent host work, ~25-70 us per window).
 * Per window, in this order: n_remove_flags > 0 -> ssx_ba_window_remove_flagged (what the LAST optimisation decided: its flags
 * refer to the window as that solve saw it); n_remove_lm > 0 -> ssx_ba_window_remove_landmarks; pop != 0 ->
 * ssx_ba_window_pop_keyframe(pop_kf_id); push != 0 -> ssx_ba_window_push_keyframe (obs_lm) or
 * _push_keyframe_slots (obs_lm NULL, obs_slot) with the remaining fields.  status_out (nullable) receives every window's own
 * status; the call returns the first one that is not SSX_OK (a failing window is left as its own failing call leaves it, the
 * others are updated).  The windows must be distinct. */
typedef struct ssx_ba_window_update {
  int32_t pop, push;
  int64_t pop_kf_id, kf_id;
  const double* pose7;
  int32_t pose_fixed, n_new;
  const int64_t* new_ids;
  const double* new_xyz;
  const uint8_t* new_fixed;
  int32_t* new_slots_out;       /* slots form only */
  int32_t n_obs, reserved;
  const int64_t* obs_lm;        /* observations by landmark id, or NULL and ... */
  const int32_t* obs_slot;      /* ... by slot (see ssx_ba_window_push_keyframe_slots) */
  const double* obs_uv;
  const uint8_t* obs_cam;
  int32_t n_remove_flags, n_remove_lm;
  const uint8_t* remove_flags;  /* n_remove_flags = the window's observation count before this update */
  const int64_t* remove_lm_ids;
} ssx_ba_window_update;
SSX_API ssx_status ssx_ba_window_update_batch(int32_t n, ssx_ba_window* const* wins, const ssx_ba_window_update* updates,
                                              ssx_status* status_out);
/* overwrite the estimate / the fixed flag of a keyframe or landmark of the window (fixed < 0: unchanged; xyz NULL: unchanged) */
SSX_API ssx_status ssx_ba_window_set_pose(ssx_ba_window* win, int64_t kf_id, const double* pose7, int32_t fixed);
SSX_API ssx_status ssx_ba_window_set_landmark(ssx_ba_window* win, int64_t lm_id, const double* xyz, int32_t fixed);
SSX_API ssx_status ssx_ba_window_size(const ssx_ba_window* win, int32_t* n_keyframes, int32_t* n_landmarks,
                                      int32_t* n_observations);
/* the window as an ordinary ssx_ba_problem (current estimate): arrays of ssx_ba_window_size() entries, any may be NULL.
 * Keyframes ascending by id, landmarks ascending by id, observations in the order they were pushed; point_fixed = the flags
 * the next solve uses (fix rule included). */
SSX_API ssx_status ssx_ba_window_export(const ssx_ba_window* win, int64_t* kf_ids, double* poses, uint8_t* pose_fixed,
                                        int64_t* lm_ids, double* points, uint8_t* point_fixed, int32_t* edge_pose,
                                        int32_t* edge_point, double* edge_uv, uint8_t* edge_cam);
/* res->poses_out / points_out / edge_chi2 / edge_outlier (nullable) in the order of ssx_ba_window_export; the window's
 * state becomes the result (the next solve starts from it) */
SSX_API ssx_status ssx_ba_window_solve(ssx_ba_window* win, ssx_ba_result* res);
/* n windows of ONE ctx in one call (one launch sequence for all of them, like ssx_ba_solve_batch): the windows of the
 * concurrent streams of BASELINE configs[4], or of a batch of stereo pairs.  Per window the bits of ssx_ba_window_solve.
 * The options of the first window apply. */
SSX_API ssx_status ssx_ba_window_solve_batch(int32_t n, ssx_ba_window* const* wins, ssx_ba_result* results);

/* (test and tools hooks -- ssx_ba_window_selftest, ssx_debug_*, ssx_ba_debug_*, the kernel taps ssx_ba_linearize / ssx_orb_stage_* /
 * ssx_lk_stage_* -- are NOT part of this ABI: include/ssx_test_hooks.h,
 * compiled out of the library by -DSSX_NO_TEST_HOOKS / SSX_PRODUCT_BUILD=1 python -m ssvio_amd.build) */

/* ------------------------------------------------------------------------------------------------
 * Pose-only robust optimisation -- replaces the g2o part of FrontEnd::EstimateCurrentPose
 * (src/ssvio/frontend.cpp:184-270): one VertexPose, EdgeProjectionPoseOnly per tracked map point
 * (include/ssvio/g2otypes.hpp:67-110), Huber delta 1.0 (g2o default), `rounds` x optimize(`iters`) with
 * chi2 > chi2_th => outlier after every round and the robust kernels dropped before the last round.
 * The whole procedure (all rounds, iterations, LM trials) is ONE kernel launch.
 *   pose_io: initial estimate in, result out;  xyz: M x 3 map points;  uv: M x 2 measured pixels (cv::Point2f
 *   widened to double);  inlier_out[i] = 1 if the feature ends as inlier;  *n_inliers = features.size() - outliers.
 *   Reference defaults: rounds 4, iters 10, chi2_th 5.991, huber_delta 1.0.
 * ------------------------------------------------------------------------------------------------ */
SSX_API ssx_status ssx_pose_only_opt(ssx_ctx* ctx, double* pose_io, const double* K4, int32_t M, const double* xyz,
                                     const double* uv, int32_t rounds, int32_t iters, double chi2_th,
                                     double huber_delta, uint8_t* inlier_out, int32_t* n_inliers);

/* n problems in ONE call -- one frame of each of n streams (BASELINE configs[4]; FrontEnd::EstimateCurrentPose is one such problem
 * per frame, frontend.cpp:184-300): one workgroup per problem, one launch, one synchronisation.  Per problem the arguments of
 * ssx_pose_only_opt and, bit for bit, its results. */
typedef struct ssx_pose_only_job {
  double* pose_io; const double* K4; int32_t M; const double* xyz; const double* uv;
  int32_t rounds, iters; double chi2_th, huber_delta;
  uint8_t* inlier_out;        /* nullable */
  int32_t* n_inliers;         /* nullable */
} ssx_pose_only_job;
SSX_API ssx_status ssx_pose_only_opt_batch(ssx_ctx* ctx, int32_t n, const ssx_pose_only_job* jobs);

/* ------------------------------------------------------------------------------------------------
 * ORB extraction -- replaces the bodies of ssvio::ORBextractor::Detect / DetectAndCompute
 * (include/ssvio/orbextractor.hpp:50-59, src/ssvio/orbextractor.cpp:755-842, 687-753) and everything
 * they call: cv::FAST per grid cell, DistributeOctTree, ComputePyramid (cv::resize), IC_Angle
 * (cv::fastAtan2), cv::GaussianBlur 7x7 sigma 2, computeOrbDescriptor (steered BRIEF-256).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {          /* binary layout of cv::KeyPoint (28 bytes) */
  float x, y;             /* pt */
  float size, angle, response;
  int32_t octave, class_id;
} ssx_keypoint;

typedef struct {          /* ORBextractor ctor arguments (orbextractor.hpp:44-45, system.cpp:117-128) */
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast, min_th_fast;
} ssx_orb_params;

SSX_API void ssx_orb_default_params(ssx_orb_params* p); /* 2000, 1.2, 8, 20, 7 (config/kitti_00.yaml:41-49) */

/* Output capacity `cap` of the two calls below: a pyramid level returns at most max(N_level + 3, 4 * nIni) keypoints
 * (the quadtree stops within 3 nodes of its budget, but its first subdivision already makes up to 4 * nIni <= 256
 * nodes however small the budget is); nfeatures + 260 * nlevels is always enough.  SSX_ERR_CAPACITY otherwise.
 *
 * ORBextractor::Detect: single-level grid FAST + octree.  mask may be NULL (all 255).  Empty image:
 * returns SSX_OK with *n = 0 (the reference silently returns, orbextractor.cpp:758-759).
 * Output: keypoints {pt, size 7, angle -1, response = FAST score, octave 0, class_id -1}. */
SSX_API ssx_status ssx_orb_detect(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                                  const uint8_t* mask, int32_t mask_stride, const ssx_orb_params* prm,
                                  int32_t cap, ssx_keypoint* kps_out, int32_t* n);
/* The same with the mask of FrontEnd::DetectFeatures (src/ssvio/frontend.cpp:302-312: 255 everywhere, a FILLED cv::rectangle of
 * pt -+ (10, 10) set to 0 per tracked feature) handed over as its rectangles: boxes_xyxy = n_boxes x (x0, y0, x1, y1), corners
 * INCLUSIVE (cv::rectangle's convention), clipped to the image by the library.  16 bytes per tracked feature cross PCIe instead
 * of rows x cols bytes; the mask is rasterised on the device.  Same keypoints as ssx_orb_detect on the rasterised mask. */
SSX_API ssx_status ssx_orb_detect_boxes(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                                        const int32_t* boxes_xyxy, int32_t n_boxes, const ssx_orb_params* prm,
                                        int32_t cap, ssx_keypoint* kps_out, int32_t* n);

/* ssx_orb_detect_boxes for n images in ONE call -- one keyframe of each of n streams (BASELINE configs[4]).  All images rows x cols with
 * one stride and one parameter set; images_on_device != 0: the image pointers are GPU-readable (device or pinned host memory) and are
 * read where they lie.  Per image the bits of ssx_orb_detect_boxes; SSX_ERR_CAPACITY if any image's keypoints do not fit its cap
 * (the others are still returned, *n_out of every job is set). */
typedef struct ssx_orb_detect_job {
  const uint8_t* img; int32_t stride;
  const int32_t* boxes_xyxy; int32_t n_boxes;
  int32_t cap; ssx_keypoint* kps_out; int32_t* n_out;
} ssx_orb_detect_job;
SSX_API ssx_status ssx_orb_detect_boxes_batch(ssx_ctx* ctx, int32_t n, const ssx_orb_detect_job* jobs, int32_t rows, int32_t cols,
                                              const ssx_orb_params* prm, int32_t images_on_device);

/* ORBextractor::DetectAndCompute: 8-level pyramid ORB.  desc_out: cap x 32 bytes (CV_8U N x 32). */
SSX_API ssx_status ssx_orb_extract(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                                   const uint8_t* mask, int32_t mask_stride, const ssx_orb_params* prm,
                                   int32_t cap, ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n);

/* ORBextractor::ScreenAndComputeKPsParams + CalcDescriptors (orbextractor.hpp:54-55,69-71; the loop-closing use,
 * src/ssvio/loopclosing.cpp:622-629): keypoints are GIVEN (level-0 position + octave); those at least 19 px inside
 * their pyramid level that pass the FAST segment test at minThFAST are kept (input order), get angle / size, and
 * are described on the blurred level.  kps_out / desc_out need capacity n_in. */
SSX_API ssx_status ssx_orb_describe_at(ssx_ctx* ctx, const uint8_t* img, int32_t stride, int32_t rows, int32_t cols,
                                       const ssx_orb_params* prm, const ssx_keypoint* kps_in, int32_t n_in,
                                       ssx_keypoint* kps_out, uint8_t* desc_out, int32_t* n);

/* ------------------------------------------------------------------------------------------------
 * Stereo association + triangulation.
 * The north_star asks for row-band Hamming matching; the reference itself associates by LK optical flow
 * (FrontEnd::FindFeaturesInRight, src/ssvio/frontend.cpp:346-428) and only matches descriptors in loop
 * closing with OpenCV BruteForce-Hamming (src/ssvio/loopclosing.cpp:105-145) -- whose semantics (minimum
 * distance, lowest train index wins ties) this matcher keeps inside the row band.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  float band_px;            /* |vL - vR| <= band_px * scale_factor^octave_L */
  float min_disp, max_disp; /* uL - uR in [min_disp, max_disp] */
  int32_t max_dist;         /* accept iff best Hamming distance <= max_dist */
  int32_t max_octave_diff;  /* |octave_L - octave_R| <= this */
  float scale_factor;
} ssx_match_params;

SSX_API void ssx_match_default_params(ssx_match_params* p); /* 2, 0, 120, 80, 1, 1.2 */

/* match_idx[i] = index of the right keypoint or -1; dist[i] = best distance found (257 = no candidate). */
SSX_API ssx_status ssx_stereo_match(ssx_ctx* ctx, const ssx_keypoint* kL, const uint8_t* dL, int32_t nL,
                                    const ssx_keypoint* kR, const uint8_t* dR, int32_t nR,
                                    const ssx_match_params* prm, int32_t* match_idx, int32_t* dist);

/* Brute-force Hamming match() of loopclosing.cpp:108 (query -> train), next-row N2. */
SSX_API ssx_status ssx_bf_match(ssx_ctx* ctx, const uint8_t* dq, int32_t nq, const uint8_t* dt, int32_t nt,
                                int32_t* idx, int32_t* dist);

typedef struct {           /* the rig System::GenerateSteroCamera builds (src/ssvio/system.cpp:54-113) */
  double fx, fy, cx, cy;
  double baseline;         /* Camera.Base.Line / fx */
} ssx_stereo_rig;

/* ssvio::triangulation (include/ssvio/algorithm.hpp:23-45) for left = [I|0], right = [I|(-baseline,0,0)] with
 * Camera::pixel2camera (src/ssvio/camera.cpp:25-30); ok = sigma3/sigma2 < 1e-2 && z > 0 (frontend.cpp:466,528).
 * A pair without positive disparity (uL <= uR: a point at or behind infinity, where the sign of z is rounding noise)
 * is reported ok = 0 with a zeroed camera-frame point.
 * T_wc (nullable, 7 doubles) maps the camera-frame point to the world (frontend.cpp:503,531). */
SSX_API ssx_status ssx_triangulate(ssx_ctx* ctx, int32_t n, const double* uvL, const double* uvR,
                                   const ssx_stereo_rig* rig, const double* T_wc, double* xyz_out,
                                   uint8_t* ok_out);
/* n calls of ssx_triangulate in one launch (one keyframe of each of n streams).  Per job the arguments and, bit for bit, the results
 * of ssx_triangulate. */
typedef struct ssx_triangulate_job {
  int32_t n; const double* uvL; const double* uvR; const ssx_stereo_rig* rig; const double* T_wc;   /* T_wc nullable */
  double* xyz_out; uint8_t* ok_out;
} ssx_triangulate_job;
SSX_API ssx_status ssx_triangulate_batch(ssx_ctx* ctx, int32_t n_jobs, const ssx_triangulate_job* jobs);

/* One stereo frame, fully on the device: extract left+right (one batched launch sequence), row-band match,
 * triangulate the matches.  Replaces DetectFeatures + FindFeaturesInRight + BuidInitMap/TriangulateNewPoints
 * of the front-end (frontend.cpp:302-544) for the north_star pipeline.  All outputs are optional. */
typedef struct {
  int32_t cap;                    /* capacity of every per-keypoint array below */
  ssx_keypoint *kpsL, *kpsR;
  uint8_t *descL, *descR;         /* cap x 32 */
  int32_t nL, nR;                 /* out */
  int32_t* match_idx;             /* nL */
  int32_t* match_dist;            /* nL */
  double* xyz;                    /* nL x 3 (valid where ok) */
  uint8_t* ok;                    /* nL */
  int32_t n_matched, n_triangulated;  /* out */
} ssx_stereo_frame_out;

SSX_API ssx_status ssx_stereo_frame(ssx_ctx* ctx, const uint8_t* imgL, const uint8_t* imgR, int32_t stride,
                                    int32_t rows, int32_t cols, const ssx_orb_params* orb,
                                    const ssx_match_params* mp, const ssx_stereo_rig* rig, const double* T_wc,
                                    ssx_stereo_frame_out* out);

/* Batched, device-resident variant for throughput: `pairs` stereo pairs already in HBM
 * (imgs_dev = [pairs][2][rows][stride] u8, DEVICE pointer).  Results stay on the device inside the ctx;
 * counts_out (host, pairs x 4 int32: nL, nR, n_matched, n_triangulated) is the only download.
 * Use ssx_stereo_batch_fetch to copy one pair's results out afterwards. */
SSX_API ssx_status ssx_stereo_batch_dev(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_dev, int32_t stride,
                                        int32_t rows, int32_t cols, const ssx_orb_params* orb,
                                        const ssx_match_params* mp, const ssx_stereo_rig* rig,
                                        int32_t* counts_out);
SSX_API ssx_status ssx_stereo_batch_fetch(ssx_ctx* ctx, int32_t pair, ssx_stereo_frame_out* out);
/* Batches that arrive from the HOST, as the reference's loop hands System::RunStep a fresh pair at every step
 * (test/test_system.cpp:36-47).  imgs_host = [pairs][2][rows][stride] u8 in host memory (pinned: the upload is then
 * asynchronous).  The images go up on a copy stream of the ctx's own into one of two device buffers; a batch's pipeline waits
 * for its own upload only.
 *   ssx_stereo_batch_upload  start the upload of a batch and return (at most two uploaded batches may be waiting); imgs_host
 *                            must stay untouched until the batch has been run and a later call of these functions returned
 *   ssx_stereo_batch_run     enqueue the whole front-end on the OLDEST uploaded batch; nothing is synchronised
 *   ssx_stereo_batch_host    upload + run in one call
 *   ssx_stereo_batch_counts  the counts of the OLDEST batch that was run and not collected yet (at most two may be waiting): waits
 *                            for that batch only; counts_out = pairs x 4 int32: nL, nR, n_matched, n_triangulated.
 *                            A run that failed after it consumed its batch still owns its place in this FIFO: the counts call
 *                            that collects it returns the run's status, and the batches around it keep their own counts.
 *                            ssx_stereo_batch_fetch copies one pair's keypoints / matches / points of the batch run LAST -- after
 *                            run(k + 1); counts() -> batch k, a fetch reads batch k + 1: fetch a batch before the next one is run.
 * A server keeps one upload ahead and collects one batch behind -- upload(k + 1); run(k); ...; counts() -> batch k - 1 -- so that
 * batch k + 1 crosses PCIe and batch k's front-end runs while the host waits for nothing but its own backend (bench.py's
 * headline region); counts() right after run(k) returns batch k's counts once it is done. */
SSX_API ssx_status ssx_stereo_batch_upload(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_host, int32_t stride, int32_t rows,
                                           int32_t cols);
SSX_API ssx_status ssx_stereo_batch_run(ssx_ctx* ctx, const ssx_orb_params* orb, const ssx_match_params* mp,
                                        const ssx_stereo_rig* rig);
SSX_API ssx_status ssx_stereo_batch_host(ssx_ctx* ctx, int32_t pairs, const uint8_t* imgs_host, int32_t stride, int32_t rows,
                                         int32_t cols, const ssx_orb_params* orb, const ssx_match_params* mp,
                                         const ssx_stereo_rig* rig);
SSX_API ssx_status ssx_stereo_batch_counts(ssx_ctx* ctx, int32_t* counts_out);
/* Timing hook: enqueue the batch again on the already-resident inputs WITHOUT any host synchronisation or
 * download (bench.py brackets a run of these with HIP events). */
SSX_API ssx_status ssx_stereo_batch_enqueue(ssx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * N1: pyramidal Lucas-Kanade tracking.
 * Replaces the two cv::calcOpticalFlowPyrLK calls of the reference front-end:
 *   FrontEnd::TrackLastFrame        /root/reference/src/ssvio/frontend.cpp:130-182 (call at :156-166)
 *   FrontEnd::FindFeaturesInRight   /root/reference/src/ssvio/frontend.cpp:346-428 (call at :374-384)
 * both with cv::Size(11,11), maxLevel 3, TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW.
 * Semantics of OpenCV 3.x calcOpticalFlowPyrLK: buildOpticalFlowPyramid (pyrDown, REFLECT_101 border, the pyramid
 * stops before a level not larger than the window), Scharr derivatives, 14-bit bilinear weights, status = 0 when
 * the window leaves the image or the smaller eigenvalue of the normal matrix falls below the threshold, err =
 * mean absolute patch difference / 32 at level 0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ssx_lk_params {
  int32_t win;                /* winSize (square, odd, 3..15): 11 */
  int32_t max_level;          /* maxLevel: 3 */
  int32_t max_iters;          /* TermCriteria COUNT: 30 */
  double eps;                 /* TermCriteria EPS: 0.01 (compared squared against |delta|^2) */
  float min_eig_threshold;    /* minEigThreshold: 1e-4 */
  int32_t use_initial_flow;   /* OPTFLOW_USE_INITIAL_FLOW: next_pts holds the initial guesses */
} ssx_lk_params;
SSX_API void ssx_lk_default_params(ssx_lk_params* p);
/* prev / next: host images rows x cols (8-bit, strides in bytes); prev_pts: n x 2 floats (x, y); next_pts: n x 2
 * floats, in (initial guess when use_initial_flow) and out; status: n bytes; err: n floats or NULL;
 * top_level (optional): the top pyramid level actually used. */
SSX_API ssx_status ssx_lk_track(ssx_ctx* ctx, const uint8_t* prev, int32_t prev_stride, const uint8_t* next,
                                int32_t next_stride, int32_t rows, int32_t cols, int32_t n, const float* prev_pts,
                                float* next_pts, uint8_t* status, float* err, const ssx_lk_params* prm,
                                int32_t* top_level);
/* Frame-to-frame chaining (FrontEnd::TrackLastFrame calls LK on (last, current) every frame, frontend.cpp:156-166):
 * the previous image is the `next` image of the last ssx_lk_track / ssx_lk_track_next call on this context, whose
 * pyramid is still on the device -- only the new image is uploaded and reduced. Same results as ssx_lk_track on the
 * two images; SSX_ERR_INVALID_ARG when there is no such call or the size / window / max_level changed. A context
 * holds one chain: keep the left-right stereo LK calls of a keyframe on a second context. */
SSX_API ssx_status ssx_lk_track_next(ssx_ctx* ctx, const uint8_t* next, int32_t next_stride, int32_t rows,
                                     int32_t cols, int32_t n, const float* prev_pts, float* next_pts,
                                     uint8_t* status, float* err, const ssx_lk_params* prm, int32_t* top_level);
/* n_jobs tracking problems in ONE call -- one frame of each of n_jobs streams (BASELINE configs[4]): every kernel of the tracker runs
 * once for all jobs.  A job names the SLOT of the context that holds its pyramids (0 .. 4095; ssx_lk_track / _next use slot 0; a
 * slot at most once per call); prev == NULL chains it to the slot's last job like ssx_lk_track_next.  All jobs share rows x cols
 * and prm.  images_on_device != 0: the image pointers are readable by the GPU -- device memory, or pinned host memory
 * (hipHostMalloc / hipHostRegister) that the level-0 kernel reads over PCIe -- and nothing is staged; 0: ordinary host memory.
 * Per job the bits of ssx_lk_track / ssx_lk_track_next. */
typedef struct ssx_lk_job {
  int32_t slot;
  const uint8_t* prev; int32_t prev_stride;      /* NULL: the slot's last `next` image */
  const uint8_t* next; int32_t next_stride;
  int32_t n; const float* prev_pts; float* next_pts; uint8_t* status; float* err;   /* err nullable */
} ssx_lk_job;
SSX_API ssx_status ssx_lk_track_batch(ssx_ctx* ctx, int32_t n_jobs, const ssx_lk_job* jobs, int32_t rows, int32_t cols,
                                      const ssx_lk_params* prm, int32_t images_on_device);
/* ------------------------------------------------------------------------------------------------
 * N3: pose-graph optimisation.
 * Replaces the optimisation of LoopClosing::PoseGraphOptimization
 * (/root/reference/src/ssvio/loopclosing.cpp:458-539): one VertexPose per keyframe (pose_fixed = the initial, the
 * active and the loop keyframes, :484-489), one EdgePoseGraph per temporal (:502-514) or loop (:515-528) constraint
 * with error log(M^-1 T_i T_j^-1) (include/ssvio/g2otypes.hpp:164-176), identity information, no robust kernel,
 * g2o numeric Jacobians, Levenberg-Marquardt, optimizer.optimize(iterations = 20).
 * poses are T_cw as (qx qy qz qw tx ty tz), in/out (fixed keyframes are returned unchanged).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ssx_pose_graph_problem {
  int32_t n_poses;
  int32_t n_edges;
  double* poses;                /* n_poses x 7, in/out */
  const uint8_t* pose_fixed;    /* n_poses */
  const int32_t* edge_i;        /* n_edges: vertex 0 of the edge (the keyframe itself) */
  const int32_t* edge_j;        /* n_edges: vertex 1 (its predecessor / its loop keyframe) */
  const double* edge_meas;      /* n_edges x 7: the measured relative pose T_i * T_j^-1 */
  double* edge_err_out;         /* optional, n_edges x 6: the edges' errors of the last evaluation */
  int32_t stats_cap;            /* optional per-iteration statistics (chi2 after the iteration, lambda, trials) */
  double* stats_chi2;
  double* stats_lambda;
  int32_t* stats_trials;
} ssx_pose_graph_problem;
typedef struct ssx_pose_graph_result {
  int32_t n_iters;              /* LM iterations executed (0: nothing to optimise) */
  int32_t stats_n;
  double chi2_initial;
  double chi2_final;
} ssx_pose_graph_result;
SSX_API ssx_status ssx_pose_graph_opt(ssx_ctx* ctx, const ssx_pose_graph_problem* prob, int32_t iterations,
                                      ssx_pose_graph_result* res);

/* ------------------------------------------------------------------------------------------------
 * Bag of words for loop detection -- replaces ORBVocabulary::transform / ::score, i.e.
 * DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ssvio/orbvocabulary.hpp:10) as LoopClosing uses it:
 *   dbow2_vocabulary_->transform(desc, keyframe->bow2_vec_)      src/ssvio/loopclosing.cpp:633
 *   dbow2_vocabulary_->score(current->bow2_vec_, db->bow2_vec_)  src/ssvio/loopclosing.cpp:84
 * The vocabulary is a k-ary tree of 32-byte descriptors kept in HBM; a descriptor descends to the child with the
 * smallest Hamming distance (the first one on ties) until it reaches a leaf = its word
 * (thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1260).  L1_NORM scoring only (what ORBvoc.txt declares).
 * A vocabulary belongs to the context it was created on and must be destroyed before it.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ssx_vocabulary ssx_vocabulary;
/* Flat tree: node 0 = root; node i >= 1: parent[i] (< i), is_leaf[i], desc[32 i .. 32 i + 31], weight[i]; entry 0 of
 * each array is ignored.  Word ids number the leaves in node order.  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY
 * (DBoW2::WeightingType); scoring must be 0 (L1_NORM). */
SSX_API ssx_status ssx_voc_create(ssx_ctx* ctx, int32_t k, int32_t L, int32_t scoring, int32_t weighting, int32_t n_nodes,
                                  const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                                  const double* weight, ssx_vocabulary** out);
/* TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1337-1420): the ORBvoc.txt format */
SSX_API ssx_status ssx_voc_load_text(ssx_ctx* ctx, const char* path, ssx_vocabulary** out);
SSX_API void ssx_voc_destroy(ssx_vocabulary* voc);
SSX_API ssx_status ssx_voc_info(const ssx_vocabulary* voc, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words,
                                int32_t* weighting);
/* transform(): desc = n x 32 bytes.  words_out / weights_out (n each, nullable): word id and node weight of every
 * feature (-1 / 0 when the vocabulary is empty).  ids_out / vals_out (capacity cap): the BowVector -- word ids in
 * ascending order with their L1-normalised values; *n_entries = its size (SSX_ERR_CAPACITY when cap is too small,
 * except cap == 0 with ids_out == NULL: only the size and the per-feature outputs are wanted). */
SSX_API ssx_status ssx_voc_transform(ssx_vocabulary* voc, const uint8_t* desc, int32_t n, int32_t* words_out,
                                     double* weights_out, int32_t cap, int32_t* ids_out, double* vals_out,
                                     int32_t* n_entries);
/* L1Scoring::score (thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) of two BowVectors; host arithmetic, in [0, 1] */
SSX_API double ssx_bow_score_l1(int32_t n1, const int32_t* id1, const double* v1, int32_t n2, const int32_t* id2,
                                const double* v2);

#ifdef __cplusplus
}
#endif
#endif /* SSX_H */
