/* include/ssx_test_hooks.h -- hooks of the TESTS and TOOLS of this repository into libssx.so.  NOT part of the product ABI
 * (include/ssx.h): they may change or disappear at any time, and a product build leaves them out altogether
 * (-DSSX_NO_TEST_HOOKS, which `SSX_PRODUCT_BUILD=1 python -m ssvio_amd.build` passes).  The default build carries them because the
 * driver's GPU tests load the very library that ships (tests/test_ba_gpu.py, tests/test_window_host.py, tools/kernel_resources.py,
 * tools/asan_prepare.py). */
#ifndef SSX_TEST_HOOKS_H
#define SSX_TEST_HOOKS_H
#include "ssx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook, needs no GPU: `steps` random pushes / pops / removals of observations and landmarks on a window without a device (fix
 * rule 1), its contents, order and fixed flags checked against a plain model after every step, and two twin windows that receive the same edits through ssx_ba_window_update_batch (two windows per call: the
 * threaded path) against the window itself; 0 = all steps agree, else the first step that does not */
SSX_API int32_t ssx_ba_window_selftest(uint32_t seed, int32_t steps);

/* tools hook, needs no GPU: dynamic LDS bytes a BA kernel is launched with (-1: depends on the problem); the compiler's
 * resource report and rocprofv3's dispatch rows only know static __shared__ arrays (tools/kernel_resources.py) */
SSX_API int64_t ssx_debug_kernel_dynamic_lds(const char* kernel);
/* tests hook: 1 = the linearise / Schur kernels write, and the reductions read, every entry of the per-chunk partial sums (round 3's
 * dense slabs); 0 (default) = only the blocks of the reduced system and the poses a chunk contributes to; < 0 = the environment's
 * choice (SSX_BA_DENSE_SLABS).  Same bits either way (tests/test_ba_gpu.py::test_sparse_slabs_equal_dense_slabs); applies to problems
 * uploaded after the call. */
SSX_API void ssx_debug_set_dense_slabs(int32_t mode);

/* tests hook: who completes an LM trial of a small window on one GPU.  1 (default) = the last chunk of k_backsub_residual to publish
 * its three sums adds them and takes the LM step (agent-scope stores + a ticket, no fence); 0 = a launch of k_reduce_trial does (rounds
 * 1-4).  Same bits either way: tests/test_ba_gpu.py::test_trial_finish_litmus. */
SSX_API void ssx_debug_set_trial_finish(int32_t mode);

/* tools hook, needs no GPU: seconds of host marshalling (edge sort by landmark, chunks, index lists) for one problem */
SSX_API double ssx_ba_debug_prepare_seconds(const ssx_ba_problem* prob, int32_t reps);

/* tools / tests hook, needs no GPU: how ssx_ba_solve / ssx_ba_solve_batch would send this problem's observation arrays
 * across PCIe (lossless narrowing): bit 0 = keyframe indices as bytes, bit 1 = landmark indices as 16-bit words, bit 2 =
 * pixel coordinates as floats (every edge_uv value is a float's value, as the reference's cv::KeyPoint::pt measurements
 * are); 0 = as handed over (large windows, SSX_BA_WIDE_UPLOAD / SSX_BA_HOST_PREP set); -1 = invalid problem */
SSX_API int32_t ssx_ba_debug_upload_format(const ssx_ba_problem* prob);

/* test hook, needs no GPU: FNV-1a digest of the host marshalling of a LARGE window (per-landmark offsets, every observation's rank
 * inside its landmark, per-keyframe counts, chunk cuts) with the observation pass on `threads` host threads (>= 65 536 observations
 * take it on the worker pool; SSX_BA_PREP_THREADS, default min(8, cores)).  The digest must not depend on `threads`.  0 = invalid. */
SSX_API uint64_t ssx_ba_debug_prepare_digest(const ssx_ba_problem* prob, int32_t threads);

/* ---- kernel taps (moved here from ssx.h in 0.1.20): intermediate results of the kernels, for the parity tests and tools ---- */

/* One linearisation of the problem at its current state (no update): the blocks the kernels build,
 * for kernel-level parity tests and profiling.  Any output may be NULL.
 *   Hpp P x 36 (row-major 6x6), bp P x 6, Hll L x 9, bl L x 3, Hpl E x 18 (6x3 row-major, per edge),
 *   err E x 2, chi2 = robust chi2.  Rows of fixed vertices are zero. */
SSX_API ssx_status ssx_ba_linearize(ssx_ctx* ctx, const ssx_ba_problem* prob, double huber_delta, int32_t jac_mode,
                            double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* err,
                            double* chi2);

/* Parity / profiling hooks: copies of intermediate buffers of the LAST ssx_orb_extract / ssx_orb_detect /
 * ssx_stereo_* call on this ctx, image `image` of that call (0 = left / only image, 1 = right ...).
 *   level image (u8, rows x cols returned), blurred level image, and the grid-FAST candidates of a level
 *   (keypoints relative to the 16-px border, reference order = cell-row-major then row-major in the cell). */
SSX_API ssx_status ssx_orb_stage_level(ssx_ctx* ctx, int32_t image, int32_t level, int32_t blurred,
                                       uint8_t* out, int32_t out_cap, int32_t* rows, int32_t* cols);
SSX_API ssx_status ssx_orb_stage_candidates(ssx_ctx* ctx, int32_t image, int32_t level, int32_t cap,
                                            ssx_keypoint* out, int32_t* n);

/* Test access to the pyramids (which = 0 previous, 1 next) and the Scharr images (int16 dx, dy interleaved) of
 * the last ssx_lk_track call. */
SSX_API ssx_status ssx_lk_stage_level(ssx_ctx* ctx, int32_t which, int32_t level, uint8_t* out, int32_t out_cap,
                                      int32_t* rows, int32_t* cols);
SSX_API ssx_status ssx_lk_stage_deriv(ssx_ctx* ctx, int32_t level, int16_t* out, int32_t out_cap, int32_t* rows,
                                      int32_t* cols);

#ifdef __cplusplus
}
#endif
#endif /* SSX_TEST_HOOKS_H */
