/* oracle/src/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU restatement ("oracle") of the reference's hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the product
 * (ssvio_amd/, include/ssx.h) never includes, links or calls anything in this directory.
 *
 * Parity status (also in DESIGN.md):
 *   - BA / pose-only / triangulation / SE3: PINNED against the real reference arithmetic compiled
 *     from /root/reference by oracle/Makefile (oracle/_ref/libssvio_ref.so) and against the golden
 *     vectors that library produced (tests/golden/, generator tests/golden/make_golden.py).
 *   - ORB (FAST / NMS / octree / IC_Angle / BRIEF / pyramid / blur) and Hamming matching: restated
 *     from the in-tree sources (file:line cited at each function) where they exist and from the
 *     published OpenCV 3.2 algorithms where the reference calls OpenCV (not vendored, not installed):
 *     PARITY UNPINNED vs OpenCV -- the reference holds no test, fixture or golden vector for them.
 */
#ifndef SSVIO_ORACLE_H
#define SSVIO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- SE3 (Sophus semantics; pose = qx qy qz qw tx ty tz) ---------------- */
void orc_se3_exp(const double* tangent6, double* pose7);
void orc_pose_oplus(const double* pose7, const double* delta6, double* out7);
void orc_se3_act(const double* pose7, const double* p3, double* out3);

/* ---------------- BA (A12-A17) ---------------- */
typedef struct {
  int outer_rounds;      /* backend.cpp:175  (5)   */
  int iters;             /* backend.cpp:178  (10)  */
  double chi2_th;        /* backend.cpp:109  (5.891) */
  double huber_delta;    /* backend.cpp:163  (5.891) */
  double inlier_ratio;   /* backend.cpp:195  (0.7) */
  int jac_mode;          /* 0 = analytic (g2otypes.hpp:133-153, commented-out formula), 1 = g2o numeric 1e-9 */
} orc_ba_options;

/* Returns outer rounds executed (>=1) or <0 on error.  poses/points are updated in place.
 * stats arrays (optional) receive one entry per LM iteration: robust chi2 after the iteration,
 * lambda after the iteration, number of trials. */
int orc_ba_solve(int P, double* poses, const uint8_t* pose_fixed, int L, double* points,
                 const uint8_t* point_fixed, int E, const int32_t* edge_pose, const int32_t* edge_point,
                 const double* edge_uv, const uint8_t* edge_cam, const double* K4, const double* cam_ext14,
                 const orc_ba_options* opt, double* edge_chi2_out, uint8_t* edge_outlier_out,
                 int stats_cap, int* stats_n, double* stats_chi2, double* stats_lambda, int* stats_trials);

/* One linearisation at the current state: fills Hpp (P x 36, row-major 6x6 blocks), bp (P x 6),
 * Hll (L x 9), bl (L x 3), Hpl (E x 18, 6x3 row-major, per edge), robust chi2; used to check kernels. */
int orc_ba_linearize(int P, const double* poses, const uint8_t* pose_fixed, int L, const double* points,
                     const uint8_t* point_fixed, int E, const int32_t* edge_pose, const int32_t* edge_point,
                     const double* edge_uv, const uint8_t* edge_cam, const double* K4, const double* cam_ext14,
                     double huber_delta, int jac_mode, double* Hpp, double* bp, double* Hll, double* bl,
                     double* Hpl, double* edge_err2, double* chi2_robust);

/* Single edge: residual, Jacobians (2x6, 2x3 row-major), chi2, Huber rho[3]. */
void orc_edge_eval(const double* pose7, const double* p3, const double* uv2, const double* K4,
                   const double* ext7, double huber_delta, int jac_mode, double* err2, double* Ji12,
                   double* Jj6, double* chi2, double* rho3);

/* Pose-only (A11).  Returns number of inliers. */
int orc_pose_only(double* pose7, const double* K4, int M, const double* xyz, const double* uv,
                  int rounds, int iters, double chi2_th, double huber_delta, uint8_t* inlier_out);

/* ---------------- Triangulation (A10) ---------------- */
void orc_triangulate(int n, const double* uvL, const double* uvR, double fx, double fy, double cx,
                     double cy, double baseline, const double* T_wc7 /*nullable*/, double* xyz_out,
                     uint8_t* ok_out, double* ratio_out /*nullable*/);

/* ---------------- ORB (A1-A8) ---------------- */
typedef struct {       /* mirrors cv::KeyPoint (28 bytes) */
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orc_keypoint;

typedef struct {
  int nfeatures; float scale_factor; int nlevels; int ini_th_fast; int min_th_fast;
} orc_orb_params;

/* cv::FAST(img, kps, thr, true) on one ROI (A2).  Output (x,y,score) row-major order. */
int orc_fast_roi(const uint8_t* img, int stride, int rows, int cols, int threshold, int cap,
                 int32_t* xs, int32_t* ys, int32_t* scores);
/* ORBextractor::Detect (A1): single-level grid FAST + octree.  mask may be NULL (= all 255). */
int orc_orb_detect(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                   const orc_orb_params* prm, int cap, orc_keypoint* kps_out);
/* Candidates of the grid-FAST stage only (before the octree), in reference order. */
int orc_orb_grid_fast(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                      int ini_th, int min_th, int cap, orc_keypoint* out);
/* DistributeOctTree (A3) on candidate list. */
int orc_octree(const orc_keypoint* cand, int n, int minX, int maxX, int minY, int maxY, int N,
               int cap, orc_keypoint* out);
/* ORBextractor::DetectAndCompute (A5-A7). desc_out: cap x 32 bytes. */
int orc_orb_extract(const uint8_t* img, int stride, int rows, int cols, const uint8_t* mask, int mask_stride,
                    const orc_orb_params* prm, int cap, orc_keypoint* kps_out, uint8_t* desc_out);
/* Building blocks exposed for kernel-level parity tests. */
void orc_level_sizes(int rows, int cols, float scale_factor, int nlevels, int32_t* rows_out, int32_t* cols_out);
void orc_features_per_level(int nfeatures, float scale_factor, int nlevels, int32_t* out);
void orc_umax(int32_t* out16);
void orc_resize_linear(const uint8_t* src, int sstride, int srows, int scols, uint8_t* dst, int dstride,
                       int drows, int dcols);
void orc_gauss7(const uint8_t* src, int sstride, int rows, int cols, uint8_t* dst, int dstride);
float orc_ic_angle(const uint8_t* img, int stride, float x, float y);
float orc_fast_atan2(float y, float x);
void orc_brief(const uint8_t* blurred, int stride, float x, float y, float angle_deg, uint8_t* desc32);
void orc_sincos_deg(float angle_deg, float* c, float* s);
/* the descriptor with libm's cosf / sinf (mode 0) or cos / sin rounded to float (mode 1) as orbextractor.cpp:49-50 calls them */
void orc_brief_libm(const uint8_t* blurred, int stride, float x, float y, float angle_deg, int mode, uint8_t* desc32, float* ab_out);
/* xya = n x (x, y, angle in degrees); out3 = differing descriptor bits, descriptors with any differing bit, (cos, sin) pairs that differ */
void orc_brief_libm_census(const uint8_t* blurred, int stride, int n, const float* xya, int mode, int64_t* out3);
int orc_is_fast_corner(const uint8_t* img, int stride, int x, int y, int threshold);
const int8_t* orc_brief_pattern(void); /* 256 x 4 int8: x0 y0 x1 y1 */
/* ScreenAndComputeKPsParams + CalcDescriptors (A8) */
int orc_orb_describe_at(const uint8_t* img, int stride, int rows, int cols, const orc_orb_params* prm,
                        const orc_keypoint* kps_in, int n_in, orc_keypoint* kps_out, uint8_t* desc_out);

/* ---------------- Stereo row-band Hamming matcher (A9, ours) ---------------- */
typedef struct {
  float band_px;        /* |vL - vR| <= band_px * scale[octave_L] */
  float min_disp, max_disp;  /* uL - uR in [min_disp, max_disp] */
  int max_dist;         /* accept iff best distance <= max_dist */
  int max_octave_diff;  /* |octave_L - octave_R| <= this */
  float scale_factor;
} orc_match_params;
void orc_stereo_match(const orc_keypoint* kL, const uint8_t* dL, int nL, const orc_keypoint* kR,
                      const uint8_t* dR, int nR, const orc_match_params* prm, int32_t* match_idx,
                      int32_t* dist);
/* BruteForce-Hamming match() + the loopclosing.cpp:111-120 filter (N2 semantic source) */
void orc_bf_match(const uint8_t* dq, int nq, const uint8_t* dt, int nt, int32_t* idx, int32_t* dist);

/* ---------------- Pyramidal Lucas-Kanade tracker (N1: frontend.cpp:156-166, 374-384; OpenCV calcOpticalFlowPyrLK) -- */
typedef struct {
  int win;                  /* winSize 11 */
  int max_level;            /* 3 */
  int max_iters;            /* TermCriteria COUNT 30 */
  double eps;               /* TermCriteria EPS 0.01 (squared internally) */
  float min_eig_threshold;  /* 1e-4 */
  int use_initial_flow;     /* OPTFLOW_USE_INITIAL_FLOW */
} orc_lk_params;
void orc_lk_default_params(orc_lk_params* p);
void orc_lk_pyr_down(const uint8_t* src, int sstride, int rows, int cols, uint8_t* dst, int dstride);
void orc_lk_scharr(const uint8_t* src, int sstride, int rows, int cols, int16_t* dxy);
/* returns the top pyramid level used; next_pts in/out, status/err out (err may be NULL) */
int orc_lk_track(const uint8_t* prev, int pstride, const uint8_t* next, int nstride, int rows, int cols, int n,
                 const float* prev_pts, float* next_pts, uint8_t* status, float* err, const orc_lk_params* prm);

/* ---------------- Pose-graph optimisation (N3: loopclosing.cpp:458-539, g2otypes.hpp:164-199) -- PINNED ---------------- */
void orc_se3_log(const double* pose7, double* tangent6);
void orc_se3_inverse(const double* pose7, double* out7);
void orc_pg_edge_eval(const double* meas7, const double* T0, const double* T1, double* err6, double* Ji36, double* Jj36);
/* poses in/out; returns LM iterations executed (<0: nothing to optimise); stats like orc_ba_solve */
int orc_pose_graph_opt(int P, double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                       const double* meas7, int iterations, double* edge_err_out, int stats_cap, int* stats_n,
                       double* stats_chi2, double* stats_lambda, int* stats_trials);

/* ---------------- bag of words (N2: DBoW2 transform / score, loopclosing.cpp:84,633) ---------------- */
int orc_voc_words(int n_nodes, const int32_t* parent, const uint8_t* is_leaf, int32_t* word_of);
void orc_voc_transform_features(int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                                const uint8_t* feat, int n, int32_t* word_out, double* weight_out);
int orc_bow_vector(int n, const int32_t* word, const double* weight, int weighting, int cap, int32_t* ids_out, double* vals_out);
double orc_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2);

#ifdef __cplusplus
}
#endif
#endif
