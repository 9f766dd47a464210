/*
 * oracle.h -- C interface of the CPU parity oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * This directory is a dependency-free CPU restatement of the hot path of
 * lixiny/ORB-SLAM2-DualCam (ORB extraction, Hamming matching, dual-camera local BA). It exists
 * only as (1) the checker the HIP kernels are held to in tests/, __graft_entry__.smoke() and
 * (2) the `cpu_baseline` leg of bench.py. Nothing under orb-slam2-dualcam_amd/ may include, link
 * or call it.
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors and cannot be built here (needs
 * OpenCV >= 3.2 and Eigen3, both absent). The OpenCV/Eigen arithmetic the
 * reference leans on (FAST, resize, GaussianBlur, fastAtan2, cvRound, SimplicialLDLT) is
 * restated from the published OpenCV 3.3/3.4.0 non-IPP semantics written down in SURVEY.md
 * Appendix A. Each function cites the reference file:line it follows.
 */
#ifndef ORB_DUALCAM_ORACLE_H
#define ORB_DUALCAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout (28 bytes), SURVEY.md Appendix E */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* pre-quadtree FAST candidate, level coordinates relative to minBorder (ORBextractor.cc:820-824) */
typedef struct orc_candidate {
    int16_t x, y;
    int32_t score;
} orc_candidate;

typedef struct orc_orb orc_orb;

/* ---- extraction (reference: src/ORBextractor.cc) ---- */
orc_orb* orc_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void     orc_orb_destroy(orc_orb*);
/* tables of the ctor (ORBextractor.cc:410-470); each array has nlevels entries (umax: 16) */
void     orc_orb_tables(const orc_orb*, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                        int32_t* n_per_level, int32_t* umax16);
/* ORBextractor::operator() (ORBextractor.cc:1043-1105). returns 0, or -1 if cap too small. */
int      orc_orb_extract(orc_orb*, const uint8_t* img, int rows, int cols, int stride,
                         orc_keypoint* kp, uint8_t* desc, int cap, int* n_out);
/* stage accessors (valid after orc_orb_extract) for stage-by-stage parity tests */
int      orc_orb_level_dims(const orc_orb*, int level, int* w, int* h);
int      orc_orb_level_copy(const orc_orb*, int level, int blurred, uint8_t* dst /* w*h */);
int      orc_orb_level_candidates(const orc_orb*, int level, orc_candidate* dst, int cap);
int      orc_orb_level_keypoints(const orc_orb*, int level, orc_keypoint* dst, int cap);

/* stand-alone stages */
/* cv::resize INTER_LINEAR 8UC1 (OpenCV 3.3/3.4.0 non-IPP), SURVEY A.2 */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
/* cv::GaussianBlur 7x7 sigma=2 BORDER_REFLECT_101, legacy 8-bit fixed point, SURVEY A.5 */
void orc_gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* cv::FAST(roi, T, nonmax=true) TYPE_9_16, SURVEY A.3. out: ROI-relative (x,y,score). returns count */
int  orc_fast_roi(const uint8_t* roi, int w, int h, int stride, int threshold, orc_candidate* out, int cap);
/* raw FAST-9/16 score of one pixel (max arc-min |diff| - 1; threshold independent) */
int  orc_fast_score(const uint8_t* p, int stride);
/* cv::fastAtan2 (OpenCV 3.x scalar polynomial), SURVEY A.4 */
float orc_fast_atan2(float y, float x);
/* IC_Angle (ORBextractor.cc:77-104) on an image with given stride, at integer (x,y) */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y);
/* computeOrbDescriptor (ORBextractor.cc:108-147) */
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32);
/* DistributeOctTree (ORBextractor.cc:539-763) on candidates (in emission order); returns count */
int  orc_distribute_octree(const orc_candidate* cand, int n, int minX, int maxX, int minY, int maxY,
                           int N, orc_candidate* out, int cap);

/* ---- matching (reference: src/ORBmatcher.cc) ---- */
/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2015-2031) */
int  orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* best / second-best loop of ORBmatcher.cc:208-231 over ALL train descriptors (brute force);
   t_mask[j] != 0 skips candidate j (may be NULL). best_idx = -1 when no candidate. */
void orc_distinctive_descriptors(const uint8_t* pool, const int32_t* off, const int32_t* idx, int n_points, int32_t* best);
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_mask,
              int32_t* best_idx, int32_t* best_d, int32_t* second_d);
/* same, restricted to CSR buckets: group g matches queries q_idx[q_off[g]..q_off[g+1]) against
   candidates t_idx[t_off[g]..t_off[g+1]) (stateless variant: no "already matched" skipping) */
void orc_knn2_grouped(const uint8_t* q, int nq, const uint8_t* t, int nt, int n_groups,
                      const int32_t* q_off, const int32_t* q_idx, const int32_t* t_off, const int32_t* t_idx,
                      int32_t* best_idx, int32_t* best_d, int32_t* second_d);
/* accept test (ORBmatcher.cc:233-236) + rotation histogram (ORBmatcher.cc:241-251, 272-290,
   1969-2010). match[i] = train index or -1. th_strict: use best < th (KF-KF variant :366). */
int  orc_ratio_rot_filter(int nq, const int32_t* best_idx, const int32_t* best_d, const int32_t* second_d,
                          int th, int th_strict, float ratio, int check_ori,
                          const float* q_angle, const float* t_angle, int32_t* match);
/* faithful greedy SearchByBoWCrossCam(F,cF,KF,cKF) (ORBmatcher.cc:162-294) on flat inputs.
   KF side = queries (kf_valid[i] != 0 means "has a good MapPoint"), F side = candidates.
   feature vectors as sorted node ids + CSR. match_f[j] = KF local index or -1. returns nmatches */
int  orc_search_by_bow_crosscam(const uint8_t* desc_kf, const float* ang_kf, const uint8_t* kf_valid, int n_kf,
                                const uint8_t* desc_f, const float* ang_f, int n_f,
                                const int32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx, int kf_n_nodes,
                                const int32_t* f_nodes, const int32_t* f_off, const int32_t* f_idx, int f_n_nodes,
                                float ratio, int check_ori, int32_t* match_f);

/* ---- projection-guided matching: ORBmatcher::SearchByProjection(F, local map points, th) (ORBmatcher.cc:539-624) and
   SearchByProjectionOnCam(Fcur, cam, Flast, th) (:954-1113), on top of Frame::GetFeaturesInArea (Frame.cc:316-376).
   The geometry that produces the window of every query (isInFrustum, Frame.cc:244-312; the motion-model projection,
   ORBmatcher.cc:990-1027) stays with the caller. Queries are processed IN ORDER: a feature matched by an earlier query
   is no longer available (mvpMapPoints[idx] && Observations() > 0). */
#define ORC_GRID_COLS 64   /* FRAME_GRID_COLS (Frame.h:40) */
#define ORC_GRID_ROWS 48   /* FRAME_GRID_ROWS (Frame.h:39) */
typedef struct orc_proj_frame {
    int32_t n_cams;
    const int32_t* cam_off;      /* [n_cams+1] global index of each camera's first feature */
    const float*   kp_x;         /* [N] mvvkeysUnTemp[c][i].pt.x in global index order */
    const float*   kp_y;
    const int32_t* kp_octave;
    const float*   kp_angle;     /* read when check_orientation */
    const uint8_t* desc;         /* [N][32] */
    const uint8_t* taken;        /* [N] mvpMapPoints[i] && Observations() > 0 before the call */
    const float*   min_x;        /* [n_cams] mvMinX */
    const float*   min_y;
    const float*   grid_w_inv;   /* [n_cams] mvfGridElementWidthInv */
    const float*   grid_h_inv;
    const int32_t* grid_off;     /* [n_cams*64*48 + 1] CSR over (c, ix, iy) of mvGrids[c][ix][iy] */
    const int32_t* grid_idx;     /* camera-local feature indices in insertion order */
} orc_proj_frame;

typedef struct orc_proj_queries {
    int32_t n;
    const uint8_t* valid;        /* [n] passes the caller's gating (mbTrackInView && !isBad / :994-1013) */
    const int32_t* cam;          /* [n] mTrackProjCamera / query camera */
    const float*   u;            /* [n] mTrackProjX */
    const float*   v;
    const float*   radius;       /* [n] r * mvScaleFactors[level] (:565) / th * mvScaleFactors[octave] (:1036) */
    const int32_t* min_level;    /* [n] level - 1 */
    const int32_t* max_level;    /* [n] level + 1 */
    const uint8_t* desc;         /* [n][32] MapPoint::GetDescriptor */
    const float*   angle;        /* [n] last frame's keypoint angle (check_orientation) */
} orc_proj_queries;

/* nn_ratio > 0: best/second + "same level" ratio rule of SearchByProjection (:606-613); nn_ratio <= 0: best only (OnCam).
   check_orientation: rotation histogram + three maxima of the OnCam variant (:1072-1101).
   match_of_query[n] = global feature index or -1; query_of_feature[N] = query index or -1. */
void orc_search_by_projection(const orc_proj_frame* f, const orc_proj_queries* q, int th_high, float nn_ratio, int check_orientation,
                              int32_t* match_of_query, int32_t* query_of_feature, int32_t* n_matches);
void orc_three_maxima(const int32_t* sizes, int L, int32_t* ind /* [3] */);
/* ORBmatcher::SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536): sequential, KeyFrame window
   (KeyFrame.cc:728-765 incl. :756), octave gate in the loop, best only. */
void orc_search_by_projection_kf(const orc_proj_frame* f, const orc_proj_queries* q, int th, int32_t* match_of_query, int32_t* query_of_feature,
                                 int32_t* n_matches);

/* Window searches with independent queries: Fuse x2 (ORBmatcher.cc:1431-1556, 1560-1706), SearchBySim3CrossCam (:1713-1965),
   SearchByProjection(KF, vpMapPoints, sAlreadyFound, th, ORBdist) (:693-799), and :416-536 with vpMatched as a snapshot.
   kf_area: candidates come from KeyFrame::GetFeaturesInArea (KeyFrame.cc:728-765, local index read as global, :756);
   chi2_inv_sigma2 (may be NULL): Fuse's e2 * mvInvLevelSigma2[octave] > 5.99 gate. best_dist may be NULL. */
int orc_search_in_window(const orc_proj_frame* f, const orc_proj_queries* q, int th, int kf_area, const float* chi2_inv_sigma2,
                         int32_t* match_of_query, int32_t* best_dist);
/* SearchForInitialization (ORBmatcher.cc:1117-1251): vMatchedDistance gate + match stealing + the stale rotation histogram */
int orc_search_for_initialization(const orc_proj_frame* f2, const orc_proj_queries* q, float nn_ratio, int check_orientation,
                                  int32_t* match12);
/* Frame::PosInGrid + the grid fill of the Frame constructor (Frame.cc:180-196, 380-390): CSR over (c, ix, iy); features whose
   cell falls outside the 64 x 48 grid are left out. grid_off[n_cams*64*48+1], grid_idx[<= N]; returns the entries written. */
int orc_frame_grid(int n_cams, const int32_t* cam_off, const float* kp_x, const float* kp_y, const float* min_x, const float* min_y,
                   const float* grid_w_inv, const float* grid_h_inv, int32_t* grid_off, int32_t* grid_idx);
/* Frame::GetFeaturesInArea (Frame.cc:316-376): camera-local indices in visiting order; returns the count */
int orc_features_in_area(const orc_proj_frame* f, int c, float x, float y, float r, int min_level, int max_level, int32_t* out, int cap);

/* ---- local BA (reference: src/Optimizer.cc:407-696 + vendored g2o) ---- */
typedef struct orc_ba_camera {
    double fx, fy, cx, cy;
    double ext[7];   /* extrinsic T_c (rig -> camera c): tx,ty,tz,qx,qy,qz,qw */
    double adj[36];  /* row-major 6x6 "adjoint" as built by Cameras::setExtrinsics (Cameras.cc:27-37) */
} orc_ba_camera;

typedef struct orc_ba_problem {
    int32_t n_poses, n_points, n_edges, n_cams;
    const double*  poses;        /* [P][7] tx,ty,tz,qx,qy,qz,qw (world -> rig), ascending KF id */
    const uint8_t* pose_fixed;   /* [P] */
    const double*  points;       /* [L][3], ascending point id */
    const int32_t* edge_pose;    /* [E] */
    const int32_t* edge_point;   /* [E] */
    const int32_t* edge_cam;     /* [E] */
    const double*  obs;          /* [E][2] */
    const double*  inv_sigma2;   /* [E] */
    const orc_ba_camera* cams;   /* [n_cams] */
    double huber_delta;          /* sqrt(5.991) (Optimizer.cc:515) */
    double chi2_th;              /* 5.991 (Optimizer.cc:607) */
    int32_t iters1, iters2;      /* 5, 10 (Optimizer.cc:587,619) */
} orc_ba_problem;

typedef struct orc_ba_result {
    double*  poses;          /* [P][7] */
    double*  points;         /* [L][3] */
    double*  edge_chi2;      /* [E] chi2 of the last computeActiveErrors (inactive edges: stale value) */
    uint8_t* edge_outlier;   /* [E] final check Optimizer.cc:653 */
    uint8_t* edge_level1;    /* [E] excluded after round 1 (Optimizer.cc:607-610) */
    int32_t  n_iters[2];     /* LM iterations run in each round */
    int32_t  n_trials[2];    /* linear solves (trials) in each round */
    double   lambda[2];      /* lambda at the end of each round */
    double   chi2_trace[32]; /* robust chi2 at the end of each iteration (round1 then round2) */
} orc_ba_result;

int orc_ba_local(const orc_ba_problem* prob, const volatile uint8_t* stop_flag, orc_ba_result* res);
/* the H / b blocks of the first linearisation (parity tap for linearizeOplus + constructQuadraticForm); see ba_oracle.cpp */
int orc_ba_linearize(const orc_ba_problem* prob, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, int32_t* pose_idx, int* n_free);
/* RobustKernelHuber::setDelta(delta) + ::robustify (Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91): rho = {rho(e), rho'(e), rho''(e)} of
   an edge's chi2 e. The one Huber every solver of the oracle uses; pinned against the reference's own statements (oracle/_ref, tests/test_oracle_ref.py). */
void orc_robust_huber(double delta, double e, double rho[3]);

/* ---- Optimizer::PoseOptimization (src/Optimizer.cc:250-405) for a batch of independent frames:
   one pose vertex per frame, unary EdgeSE3ProjectXYZOnlyPose edges (types_six_dof_expmap.cpp:200-255), dense 6x6 solve,
   4 rounds restarting from the initial pose, re-classification of every edge after each round. */
typedef struct orc_pose_problem {
    int32_t n_frames, n_cams;
    const double*  poses;        /* [F][7] pFrame->mTcw as SE3Quat */
    const int32_t* edge_off;     /* [F+1] edges of frame f: edge_off[f] .. edge_off[f+1] (features with a MapPoint, ascending i) */
    const double*  xw;           /* [E][3] MapPoint world position (float in the reference) */
    const double*  obs;          /* [E][2] mvTotalKeysUn[i].pt */
    const double*  inv_sigma2;   /* [E] mvInvLevelSigma2[octave] */
    const int32_t* edge_cam;     /* [E] keypointToCam[i] */
    const orc_ba_camera* cams;   /* [n_cams] */
    double  huber_delta;         /* (float)sqrt(5.991) (:284) */
    float   chi2_th[4];          /* {5.991f x4} (:352); compared in float (:375-377) */
    int32_t its[4];              /* {10,10,10,10} (:354) */
} orc_pose_problem;

typedef struct orc_pose_result {
    double*  poses;          /* [F][7] */
    uint8_t* outlier;        /* [E] pFrame->mvbOutlier of the edge's feature */
    int32_t* n_inliers;      /* [F] return value: nInitialCorrespondences - nBad (0 when < 3 correspondences) */
    double*  edge_chi2;      /* [E] e->chi2() as read by the last classification (NULL allowed) */
    int32_t* n_iters;        /* [F][4] LM iterations per round (NULL allowed) */
} orc_pose_result;

int orc_pose_optimization(const orc_pose_problem* prob, orc_pose_result* res);

/* edge pieces, for unit tests (types_six_dof_expmap.cpp:109-169) */
void orc_ba_edge_error(const double pose[7], const double point[3], const orc_ba_camera* cam,
                       const double obs[2], double err[2], double* depth);
void orc_ba_edge_jacobian(const double pose[7], const double point[3], const orc_ba_camera* cam,
                          double J_pose[12] /* 2x6 row-major */, double J_point[6] /* 2x3 */);
/* SE3Quat::exp(update) * T (types_six_dof_expmap.h:73-76, se3quat.h:223-257) */
void orc_se3_oplus(const double pose_in[7], const double update[6], double pose_out[7]);
/* Cameras::setExtrinsics adjoint (Cameras.cc:27-37) from a float 4x4 T (row-major), LL block = 0 */
void orc_rig_adjoint(const float T44[16], int exact, double adj36[36], double ext7[7]);

/* Frame::isInFrustum (Frame.cc:244-312) + PredictScale (MapPoint.cc:440-455) + the search window of SearchByProjection
   (ORBmatcher.cc:65-71, 557-565) for n map points; per-camera matrices as the caller's cv::Mat code forms them. */
typedef struct orc_frustum_frame {
    int32_t n_cams;
    const float* Rsw; const float* tsw; const float* Ow;
    const float* fx; const float* fy; const float* cx; const float* cy;
    const float* min_x; const float* max_x; const float* min_y; const float* max_y;
    float log_scale_factor; int32_t n_scale_levels; const float* scale_factors;
} orc_frustum_frame;
void orc_is_in_frustum(const orc_frustum_frame* f, int n, const float* pos, const float* normal, const float* min_dist,
                       const float* max_dist, const uint8_t* candidate, float viewing_cos_limit, float th, uint8_t* in_view,
                       int32_t* cam, float* u, float* v, float* view_cos, int32_t* level, float* radius);
/* cv::undistortPoints(src, dst, K, dist, Mat(), K) as Frame::UndistortKeyPoints calls it (Frame.cc:410-441); OpenCV 3.3 / 3.4.0 cvUndistortPoints restated */
void orc_undistort_points(int n, const float* xy, const float K4[4], const float* dist, int n_dist, float* out);
/* the geometry of ORBmatcher::SearchByProjectionOnCam (ORBmatcher.cc:962-968, 990-1036) for the last frame's features that hold a good map point */
void orc_motion_model_queries(const orc_frustum_frame* f, int n, const float* pos, const int32_t* q_cam, const int32_t* q_octave, float th,
                              uint8_t* valid, float* u, float* v, float* radius, int32_t* min_level, int32_t* max_level);

/* ---------------------------------------------------------------------------------------------------------------
   BoW front half: DBoW2 vocabulary tree, transform -> BowVector + FeatureVector, L1 score.
   The vocabulary is given as the columns of the reference's text format (TemplatedVocabulary.h:1362-1446): row i
   describes node i + 1 (node 0 = root): parent id, leaf flag, 32-byte descriptor, weight. Children of a node are
   visited in row order (m_nodes[pid].children.push_back(nid), :1416); word ids count the leaves in row order (:1434). */
typedef struct orc_vocab orc_vocab;
orc_vocab* orc_vocab_create(int k, int L, int scoring, int weighting, int n_rows, const int32_t* parent, const uint8_t* is_leaf,
                            const uint8_t* desc, const double* weight);
void orc_vocab_destroy(orc_vocab* v);
int  orc_vocab_words(const orc_vocab* v);
/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (:1151-1228) for one image.
   word[n] / node[n]: per-feature word id and node id at level L - levelsup (-1 / -1 for a stopped word);
   BowVector = (bow_word ascending, bow_val), FeatureVector = (fv_node ascending, fv_off[n_nodes+1], fv_idx).
   All arrays sized n (fv_off n + 1). Returns 0, or -1 for an unsupported scoring type. */
int  orc_bow_transform(const orc_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node,
                       int32_t* bow_word, double* bow_val, int* n_words, int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx, int* n_nodes);
/* L1Scoring::score (ScoringObject.cpp:23-67) of one BowVector against n_db BowVectors stored as CSR */
void orc_bow_score_l1(const int32_t* q_word, const double* q_val, int nq, const int32_t* db_off, const int32_t* db_word,
                      const double* db_val, int n_db, double* score);

/* KeyFrameDatabase::DetectLoopCandidatesForCam (src/KeyFrameDatabase.cc:111-235) / DetectRelocalizationCandidates (:237-372) of one
   camera pair on flat arrays with real inverted files; st_* = the key frames' mn*Query / mn*Words / m*Score members (in/out) */
int orc_detect_candidates(int loop, int query_id, const int32_t* q_word, const double* q_val, int nq, int n_db, const int32_t* db_off,
                          const int32_t* db_word, const double* db_val, const uint8_t* dead, const uint8_t* connected, float min_score,
                          const int32_t* covis_off, const int32_t* covis_idx, int32_t* st_query, int32_t* st_words, float* st_score,
                          int32_t* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
