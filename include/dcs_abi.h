/*
 * dcs_abi.h -- C ABI of the MI355X-native dual-camera SLAM hot path ("dcs").
 *
 * Drop-in boundary for the three seams of lixiny/ORB-SLAM2-DualCam; the reference
 * has no FFI layer, so each entry point cites the C++ member it replaces:
 *   extraction : ORBextractor::ORBextractor / operator()     include/ORBextractor.h:51-61,
 *                                                             src/ORBextractor.cc:410-470, 1043-1105
 *   matching   : ORBmatcher::DescriptorDistance + the best/second-best/ratio/rot-hist kernel shared
 *                by every Search... / Fuse member                include/ORBmatcher.h:49-281,
 *                                                             src/ORBmatcher.cc:162-294, 1969-2031
 *   local BA   : Optimizer::LocalBundleAdjustment             include/Optimizer.h:55,
 *                                                             src/Optimizer.cc:407-696 (+ vendored g2o)
 *
 * Conventions: every function returns DCS_OK (0) or a negative dcs_status; nothing throws or
 * exits across the boundary (the reference's exit() calls, ORBmatcher.cc:1168..., are not
 * reproduced). Plain pointers + sizes only. Functions without a `_device` suffix take HOST
 * buffers (what cv::Mat / std::vector hand over in the reference) and synchronise before
 * returning; `_device` functions take HBM-resident buffers plus a hipStream_t (as void*) and
 * only enqueue work (outputs are valid once the stream is synchronised).
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * DCS_ERR_NO_DEVICE.
 *
 * Threading contract (mirrors the reference): a dcs_orb handle is NOT thread-safe
 * (ORBextractor is stateful, one per camera, Tracking thread only); dcs_match_* / dcs_hamming_*
 * are re-entrant (ORBmatcher is stack-constructed concurrently on 3 threads); a dcs_ba handle
 * has a single caller; `stop_flag` may be written asynchronously (LocalMapping.cc:141).
 */
#ifndef DCS_ABI_H
#define DCS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dcs_status {
    DCS_OK = 0,
    DCS_ERR_INVALID = -1,     /* bad argument */
    DCS_ERR_CAPACITY = -2,    /* caller buffer too small; *n_out holds the required size */
    DCS_ERR_HIP = -3,         /* HIP runtime error (dcs_last_error() has the text) */
    DCS_ERR_NO_DEVICE = -4,   /* no gfx950 device visible: there is no CPU path */
    DCS_ERR_UNSUPPORTED = -5
} dcs_status;

const char* dcs_last_error(void);          /* thread-local text of the last failure */
const char* dcs_version(void);
int dcs_device_count(void);

/* ---- chip partitioning for config C5 (one dual-camera stream extracting / matching next to its LocalMapping thread's local BA,
 * src/LocalMapping.cc:97-104 beside src/Tracking.cc:236-269). The bundle adjustment is a chain of short kernels around a one-workgroup
 * factorisation that needs a CU's whole LDS; time-sliced against a front end that fills every CU it waits for CUs to drain. A CU mask
 * gives each side its own compute units (hipExtStreamCreateWithCUMask; mask bit k selects a CU of XCD k mod 8, so a prefix of the bits
 * spreads evenly over the eight XCDs).
 *   dcs_stream_create_cu_range  a stream restricted to the CUs [first_cu, first_cu + n_cus) of the mask order, for the caller's front-end
 *                               launches (the _device entry points run on the caller's stream)
 *   dcs_ba_set_cu_range         every solver stream created AFTER the call (new host threads, or after dcs_ba_release_thread) is
 *                               restricted to that range; n_cus = 0 removes the restriction.
 *   dcs_ba_release_thread       frees the calling thread's solver context (arena, pinned words, streams); the next call rebuilds it
 *   dcs_host_alloc / _free      page-locked host memory for the caller's frame ring (the cv::Mat headers of src/Frame.cc:141-149 can wrap it:
 *                               cv::Mat(rows, cols, CV_8UC1, ptr, stride)). dcs_orb_extract_batch recognises images that lie in page-locked
 *                               memory (this call, hipHostMalloc or hipHostRegister) at equal spacing with a 4-byte aligned stride and lets
 *                               the DMA read them in place -- no staging copy on the host
 *   dcs_streams_share_queue     HIP maps a process's streams onto a few hardware queues (four per priority by default), and two streams on one
 *                               queue run strictly one after the other. *shared = 1 when work on b waits for work on a (measured with two
 *                               probe kernels, ~0.1 ms on idle streams; b is synchronised first, the probe takes its turn behind a's backlog);
 *                               *shared = -1 when a's backlog kept the probe kernel from starting within 50 ms: nothing was measured.
 *                               BOTH calls ENQUEUE WORK on the streams they probe -- a one-wave kernel is parked on stream_a / on every
 *                               avoid[] stream for up to a millisecond (a null entry = the legacy default stream: every blocking stream of
 *                               the process waits with it): never call them while one of those streams is being captured into a graph, and
 *                               create streams at set-up time, not inside a latency-critical section
 *   dcs_stream_create_apart     a non-blocking stream that shares its hardware queue with none of avoid[] (*apart = 0 when the process has
 *                               fewer queues than that needs, or when a backlogged avoid[] stream could not be probed): for the stream the matcher / a second extraction lane / the solver runs on
 *                               next to the caller's front-end stream. The library's own streams are created this way.
 *   dcs_ba_avoid_streams        the solver's streams created AFTER the call (new host threads, or after dcs_ba_release_thread) keep off the
 *                               hardware queues of these streams (the Tracking thread's extraction and matcher streams; n = 0 clears) */
int  dcs_streams_share_queue(void* stream_a, void* stream_b, int* shared);
int  dcs_stream_create_apart(void* const* avoid, int n_avoid, void** stream, int* apart);
int  dcs_ba_avoid_streams(void* const* avoid, int n_avoid);
int  dcs_stream_create_cu_range(int first_cu, int n_cus, void** stream);
void dcs_stream_destroy(void* stream);
int  dcs_host_alloc(void** ptr, size_t bytes);
int  dcs_host_free(void* ptr);
int  dcs_ba_set_cu_range(int first_cu, int n_cus);
int  dcs_ba_release_thread(void);

/* cv::KeyPoint layout, 28 bytes: pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct dcs_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} dcs_keypoint;

/* FAST candidate before the quadtree: level coordinates relative to minBorder (ORBextractor.cc:820-824) */
typedef struct dcs_candidate {
    int16_t x, y;
    int32_t score;
} dcs_candidate;

/* ------------------------------------------------------------------ options
   The library's tuning and A/B switches (fused blur, FAST launch grouping, emitting FAST, quadtree mode, LDL^T variant, small-call graph, host
   chunk size, ...: the table in csrc/config.h). An option is named like the environment variable that seeds it ("DCS_ORB_FUSED_BLUR"; the
   prefix may be omitted) and holds an integer. The environment is read once per process; dcs_option_set changes the process-wide value from
   then on: handle-less entry points (matcher, solver, tracking chain) read it per call, an extractor handle copies the extraction-pipeline
   options when it is created -- set, create handle A, set again, create handle B gives two handles that differ. Unknown name: DCS_ERR_INVALID. */
int  dcs_option_count(void);
const char* dcs_option_name(int index);
int  dcs_option_get(const char* name, int64_t* value);
int  dcs_option_set(const char* name, int64_t value);

/* ------------------------------------------------------------------ extraction */
typedef struct dcs_orb dcs_orb;

typedef struct dcs_orb_params {
    int32_t nfeatures;        /* ORBextractor ctor args (ORBextractor.h:51-52) */
    float   scale_factor;
    int32_t nlevels;          /* 1..16 */
    int32_t ini_th_fast;
    int32_t min_th_fast;      /* >= 1 */
    int32_t device;           /* HIP ordinal, -1 = current device */
    int32_t max_images;       /* images per batched call this handle is sized for (>= 1) */
    int32_t host_threads;     /* worker threads for the host-side quadtree stage, 0 = auto */
} dcs_orb_params;

int  dcs_orb_create(const dcs_orb_params* p, dcs_orb** out);
void dcs_orb_destroy(dcs_orb* h);

/* getters GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
   GetInverseScaleSigmaSquares (ORBextractor.h:63-83) + mnFeaturesPerLevel; arrays of nlevels */
int  dcs_orb_tables(const dcs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* n_per_level);

/* Smallest `cap` the extract calls accept for rows x cols images: the reference returns up to
   sum over levels of max(N_level + 3, 4 * round(w/h)) keypoints (DistributeOctTree splits every initial node once
   before it looks at N, ORBextractor.cc:594-673); *cap is that bound with a little slack per level. */
int  dcs_orb_required_cap(const dcs_orb* h, int rows, int cols, int* cap);

/* ORBextractor::operator()(image, mask (ignored), keypoints, descriptors).
   image: 8-bit single channel, `stride` bytes per row. kp[cap], desc[cap*32] caller-owned.
   rows/cols == 0 or image == NULL -> *n_out = 0, DCS_OK (ORBextractor.cc:1046-1047). */
int  dcs_orb_extract(dcs_orb* h, const uint8_t* image, int rows, int cols, int stride,
                     dcs_keypoint* kp, uint8_t* desc, int cap, int* n_out);

/* n_images equally sized images in one pass (dual frame: n_images = 2; multi-stream: more).
   Outputs are slotted: image i -> kp[i*cap ...], desc[i*cap*32 ...], n_out[i]. */
int  dcs_orb_extract_batch(dcs_orb* h, const uint8_t* const* images, int n_images, int rows, int cols,
                           int stride, dcs_keypoint* kp, uint8_t* desc, int cap, int* n_out);

/* HBM-resident variant: d_images = n_images * rows * stride bytes; outputs in HBM, slotted as
   above; d_n_out[n_images]. Work is enqueued on `stream` (hipStream_t); nothing is read back, so the one
   run-time failure -- the batch's FAST candidates exceed the handle's candidate buffer (1/16 of the pyramid pixels per
   image, shared by the batch; a scene of pure salt-and-pepper corners) -- is reported in band: every d_n_out[i] of that
   call is DCS_ERR_CAPACITY (negative) instead of a count, and no keypoint of the call is valid. The host-buffer entry points
   return DCS_ERR_CAPACITY for the same condition. d_images may have any alignment and stride; d_desc must be 16-byte aligned and
   d_kp / d_n_out 4-byte aligned (descriptors are written and later read as 16-byte words; hipMalloc returns 256-byte alignment). `cap` < dcs_orb_required_cap() is rejected before anything is enqueued. */
int  dcs_orb_extract_batch_device(dcs_orb* h, const uint8_t* d_images, int n_images, int rows, int cols,
                                  int stride, dcs_keypoint* d_kp, uint8_t* d_desc, int cap,
                                  int32_t* d_n_out, void* stream);

/* stage taps for parity tests (valid after an extract call on the same handle; host buffers) */
/* test tap: the cosf / sinf the describe kernel evaluates for the steering coefficients (libm's float overloads, which is
   what `cos(angle)` / `sin(angle)` on a float are in src/ORBextractor.cc:112-113), for n angles in radians (|x| < 120). */
int  dcs_debug_sincosf(const float* x, int n, float* cos_out, float* sin_out);
int  dcs_orb_debug_level_dims(const dcs_orb* h, int level, int* w, int* h_out);
/* blurred = 1: the GaussianBlur'ed level (src/ORBextractor.cc:1081-1083). When the last call described its keypoints with the fused
   kernel (no blurred pyramid exists in the product path then) the level is blurred on demand from the raw pyramid -- for level 0
   of the _device API that is the caller's buffer, which must still be alive. */
int  dcs_orb_debug_level(dcs_orb* h, int image, int level, int blurred, uint8_t* dst /* w*h */);
int  dcs_orb_debug_candidates(dcs_orb* h, int image, int level, dcs_candidate* dst, int cap, int* n);
/* number of (image, level) quadtrees of the last call that left the LDS histogram fast path for the general
   sort-based kernel (device-quadtree mode; 0 in host-quadtree mode) */
int  dcs_orb_debug_quadtree_fallbacks(dcs_orb* h, int* n);
/* which way the last dcs_orb_extract_batch went: *direct = 1 when the DMA read the caller's page-locked frames in place (0: packed into the
   library's staging), *graph_replayed = 1 when a 1-2 image call was replayed as the handle's executable graph (either may be NULL) */
int  dcs_orb_debug_host_path(const dcs_orb* h, int* direct, int* graph_replayed);
/* *fast_hw = 1: this handle's k_fast_cells launches use the two hardware-specific instruction forms (the zeroing ds_read_u8_d16_hi, the
   v_cmpx append), which a one-wave probe verified on the handle's device when it was created; 0: the probe found the device behaving
   differently (or DCS_FAST_HW_PROBE=fail asked for it) and the plain forms run -- same results. */
int  dcs_orb_debug_fast_hw(const dcs_orb* h, int* fast_hw);
/* *levels = pyramid levels 1 .. *levels of the handle's LAST call were written by the FAST cells of the level below (k_fast_cells<EMIT>:
   ComputePyramid fused into the per-cell FAST pass); 0: the k_resize chain produced the pyramid (small batches, unaligned level 0, separate
   blur kernels, option DCS_ORB_EMIT = 0). */
int  dcs_orb_debug_emit_levels(const dcs_orb* h, int* levels);
/* per-stage time of the last TIMED extraction in microseconds (hipEvents on the streams the kernels ran on):
   resize chain, k_fast_cells, scan+gather, k_blur, quadtree, k_describe, whole call (7 floats).
   Which extractions are timed: calls of MORE than two images under timing mode 1 / 2 (dcs_orb_set_timing). A call of one or two images
   records no markers (they would cost 45 us of a 160-us dual-frame call) so after such a
   call this function fails with DCS_ERR_INVALID "no timing available". A host-buffer call of >= 128 images runs as a pipeline of chunks,
   each chunk an extraction of its own: the figures describe the LAST CHUNK only. */
int  dcs_orb_last_timing(dcs_orb* h, float* us7);
/* sums of the same 7 stage times over every TIMED extraction since the last reset; *n_calls counts extractions, i.e. CHUNKS for a
   chunked host-buffer call (sum / n_calls = per chunk; the sums themselves cover the whole call). Event sets live in a ring and are read
   lazily, so asynchronous (_device) callers are never stalled by the instrumentation. */
int  dcs_orb_timing_totals(dcs_orb* h, double* sum_us7, int64_t* n_calls, int reset);
/* what the stage markers above cost: every hipEventRecord is a packet between two kernels on the stream, a call of more than two images
   records 7 of them, and on the 512-image benchmark step that is 2.7 % of the step (1.405 instead of 1.443 ms without them). mode 2 (the
   default): every stage; mode 1: only the FAST stage (the roofline kernel of bench.py) is bracketed, the other entries of the timing
   arrays read 0; mode 0: no markers (dcs_orb_last_timing then fails with "no timing available"). Pending event sets are read first. */
int  dcs_orb_set_timing(dcs_orb* h, int mode);

/* DistributeOctTree (ORBextractor.cc:539-763) alone, host buffers (used by tests) */
int  dcs_distribute_octree(const dcs_candidate* cand, int n, int min_x, int max_x, int min_y, int max_y,
                           int n_target, dcs_candidate* out, int cap, int* n_out);

/* ------------------------------------------------------------------ matching */
#define DCS_TH_LOW 50          /* ORBmatcher.cc:58 */
#define DCS_TH_HIGH 100        /* ORBmatcher.cc:57 */
#define DCS_HISTO_LENGTH 30    /* ORBmatcher.cc:59 */

/* best / second-best Hamming loop (ORBmatcher.cc:208-231) of every query against every train
   descriptor (brute force). t_mask[j] != 0 skips j (NULL = none). best_idx = -1, best = second =
   256 when there is no candidate. First minimum wins ties (strict <). Descriptors are 32-byte rows. */
int  dcs_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_mask,
                      int32_t* best_idx, int32_t* best_d, int32_t* second_d);

/* same restricted to CSR buckets (a DBoW2 FeatureVector per side, Frame.cc:400-402): group g matches
   queries q_idx[q_off[g]..q_off[g+1]) against t_idx[t_off[g]..t_off[g+1]). Queries in no group get -1/256/256. */
int  dcs_hamming_knn2_grouped(const uint8_t* q, int nq, const uint8_t* t, int nt, int n_groups,
                              const int32_t* q_off, const int32_t* q_idx, const int32_t* t_off, const int32_t* t_idx,
                              int32_t* best_idx, int32_t* best_d, int32_t* second_d);

/* accept test best <= th (th_strict: best < th) and (float)best < ratio*(float)second
   (ORBmatcher.cc:233-236, :366) + 30-bin rotation histogram with ComputeThreeMaxima
   (ORBmatcher.cc:241-251, 272-290, 1969-2010). match[i] = train index or -1. */
int  dcs_match_filter(int nq, const int32_t* best_idx, const int32_t* best_d, const int32_t* second_d,
                      int th, int th_strict, float ratio, int check_ori,
                      const float* q_angle, const float* t_angle, int32_t* match, int* n_matches);

/* fused brute-force matcher "Hamming BF + ratio test across the two camera streams": knn2 + filter.
   Angles are taken from the keypoints. */
/* ---- projection-guided matching: ORBmatcher::SearchByProjection(F, local map points, th) (ORBmatcher.cc:539-624) and
   SearchByProjectionOnCam(Fcur, cam, Flast, th) (:954-1113) on top of Frame::GetFeaturesInArea (Frame.cc:316-376).
   The geometry that produces every query's window (Frame::isInFrustum, Frame.cc:244-312; the motion-model projection,
   ORBmatcher.cc:990-1027) stays with the caller. Queries are honoured IN ORDER: a feature matched by an earlier query is no
   longer available (mvpMapPoints[idx] && Observations() > 0), exactly like the reference's sequential loop. */
#define DCS_GRID_COLS 64   /* FRAME_GRID_COLS (Frame.h:40) */
#define DCS_GRID_ROWS 48   /* FRAME_GRID_ROWS (Frame.h:39) */
typedef struct dcs_proj_frame {
    int32_t n_cams;
    const int32_t* cam_off;      /* [n_cams+1] global index of each camera's first feature (prefix of mvN) */
    const float*   kp_x;         /* [N] mvvkeysUnTemp[c][i].pt.x, global index order */
    const float*   kp_y;
    const int32_t* kp_octave;
    const float*   kp_angle;     /* read when check_orientation */
    const uint8_t* desc;         /* [N][32] */
    const uint8_t* taken;        /* [N] mvpMapPoints[i] && Observations() > 0 before the call */
    const float*   min_x;        /* [n_cams] mvMinX */
    const float*   min_y;
    const float*   grid_w_inv;   /* [n_cams] mvfGridElementWidthInv */
    const float*   grid_h_inv;
    const int32_t* grid_off;     /* [n_cams*64*48 + 1] CSR over (c, ix, iy) of mvGrids[c][ix][iy] (dcs_frame_grid) */
    const int32_t* grid_idx;     /* camera-local feature indices in insertion order */
} dcs_proj_frame;

typedef struct dcs_proj_queries {
    int32_t n;
    const uint8_t* valid;        /* [n] passes the caller's gating (mbTrackInView && !isBad / ORBmatcher.cc:994-1013) */
    const int32_t* cam;          /* [n] mTrackProjCamera / the query camera */
    const float*   u;            /* [n] mTrackProjX */
    const float*   v;
    const float*   radius;       /* [n] r * mvScaleFactors[level] (:565) / th * mvScaleFactors[octave] (:1036) */
    const int32_t* min_level;    /* [n] level - 1 */
    const int32_t* max_level;    /* [n] level + 1 */
    const uint8_t* desc;         /* [n][32] MapPoint::GetDescriptor */
    const float*   angle;        /* [n] the last frame's keypoint angle (check_orientation) */
} dcs_proj_queries;

/* Frame::PosInGrid + grid fill (Frame.cc:180-196, 380-390) as CSR; pure host helper. grid_off[n_cams*64*48+1],
   grid_idx[N]; *n_entries = features that fell inside the grid. */
int  dcs_frame_grid(int n_cams, const int32_t* cam_off, const float* kp_x, const float* kp_y, const float* min_x, const float* min_y,
                    const float* grid_w_inv, const float* grid_h_inv, int32_t* grid_off, int32_t* grid_idx, int* n_entries);
/* nn_ratio > 0: best / second + the "same level" ratio rule of SearchByProjection (:606-613), TH_HIGH = th_high;
   nn_ratio <= 0: best only (SearchByProjectionOnCam). check_orientation: rotation histogram + three maxima (:1072-1101).
   match_of_query[n] = global feature index or -1; query_of_feature[N] = query index or -1. */
int  dcs_search_by_projection(const dcs_proj_frame* frame, const dcs_proj_queries* queries, int th_high, float nn_ratio,
                              int check_orientation, int32_t* match_of_query, int32_t* query_of_feature, int* n_matches);

/* The same call also is SearchByProjectionOnCam(F, query, KF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:812-951: best only,
   taken = pF->mvpMapPoints[g] != NULL, th_high = ORBdist, levels nPredictedLevel -+ 1, check_orientation); it marks the matched
   feature as taken for the queries that follow.

   SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536, loop closing) searches a KEY FRAME: its
   candidates come from KeyFrame::GetFeaturesInArea (KeyFrame.cc:728-765: no level argument, and the |dx|,|dy| < r test reads
   mvTotalKeysUn[camera-LOCAL index], :756 -- for cameras > 0 the position of another key point decides; reproduced), the octave
   gate nPredictedLevel - 1 <= octave <= nPredictedLevel sits in the loop (:510-513: min_level / max_level of the queries), best
   only, bestDist <= th (TH_LOW), a matched feature is taken for the queries that follow (:506, :527). frame->taken[g] =
   (vpMatched[g - cam_off[query]] != NULL): the reference indexes vpMatched with the camera-local index. */
int  dcs_search_by_projection_kf(const dcs_proj_frame* frame, const dcs_proj_queries* queries, int th,
                                 int32_t* match_of_query, int32_t* query_of_feature, int* n_matches);

/*

   Window searches whose queries do NOT see each other's results -- the candidate loops of
     Fuse(KF, vpMapPoints, th)                        ORBmatcher.cc:1431-1556  kf_area = 1, chi2 gate, levels pred - 1 .. pred, th = TH_LOW
     Fuse(KF, Scw, vpPoints, th, vpReplacePoint)      :1560-1706               kf_area = 1, no chi2 gate, levels pred - 1 .. pred, TH_LOW
     SearchBySim3CrossCam (each direction)            :1713-1965               kf_area = 1, levels pred - 1 .. pred, th = TH_HIGH
     SearchByProjection(KF, vpMapPoints, sAlreadyFound, th, ORBdist)  :693-799 kf_area = 1, levels pred - 1 .. pred + 1, th = ORBdist
   one wave per query. kf_area != 0: candidates as KeyFrame::GetFeaturesInArea returns them (KeyFrame.cc:728-765: no level
   argument, and the |dx|,|dy| < r test reads mvTotalKeysUn[camera-LOCAL index], :756 -- reproduced); the octave gate is the
   loop's own (min_level <= octave <= max_level). chi2_inv_sigma2 (NULL = off; n_levels entries = mvInvLevelSigma2): Fuse's
   reprojection gate e2 * mvInvLevelSigma2[octave] > 5.99 (:1503-1509). frame->taken may be NULL (nothing to skip).
   match_of_query[n] = GLOBAL feature index or -1 (bestDist <= th), best_dist[n] (NULL allowed) = bestDist (256: no candidate).
   What the reference then does with a match (Replace / AddObservation / agreement check of the two Sim3 directions) is
   bookkeeping on its map and stays with the caller. */
int  dcs_search_in_window(const dcs_proj_frame* frame, const dcs_proj_queries* queries, int th, int kf_area,
                          const float* chi2_inv_sigma2, int n_levels, int32_t* match_of_query, int32_t* best_dist, int* n_matches);

/* ORBmatcher::SearchForInitialization (ORBmatcher.cc:1117-1251). frame2 = F2 (its grid; `taken` is ignored), queries = F1's key
   points in global order: valid[i] = "camera CAP and octave 0" (:1142-1149), (u, v) = vbPrevMatched[i], radius = windowSize,
   min_level = max_level = the key point's octave (:1152), desc / angle = F1's. In-loop state reproduced: a candidate held by an
   earlier query with a distance <= this one's is skipped (vMatchedDistance, :1176), an accepted query takes the feature from
   its previous owner (vnMatches21, :1194-1198), TH_LOW and bestDist < bestDist2 * nn_ratio (:1190-1192), and the rotation
   histogram keeps robbed entries (:1206, 1228-1240). match12[n] = global F2 feature or -1, *n_matches = the return value. */
int  dcs_search_for_initialization(const dcs_proj_frame* frame2, const dcs_proj_queries* queries, float nn_ratio, int check_orientation,
                                   int32_t* match12, int* n_matches);

/* Frame::isInFrustum (Frame.cc:244-312) for a batch of map points + the window of SearchByProjection (ORBmatcher.cc:557-565):
   the geometry gate in front of dcs_search_by_projection. The caller supplies the per-camera matrices exactly as the reference
   forms them with cv::Mat (Tsw = mvExtrinsics[c] * mTcw -> Rsw, tsw, Frame.cc:252-256; GetCameraCenter(c), :222-235);
   cameras are tried in order and the first that sees the point wins (n_cams = 1 when bForAllCam is false). */
typedef struct dcs_frustum_frame {
    int32_t n_cams;
    const float* Rsw;            /* [n_cams][9] row-major */
    const float* tsw;            /* [n_cams][3] */
    const float* Ow;             /* [n_cams][3] camera centres in the world */
    const float* fx; const float* fy; const float* cx; const float* cy;          /* [n_cams] mvfx ... */
    const float* min_x; const float* max_x; const float* min_y; const float* max_y;   /* [n_cams] mvMinX ... */
    float   log_scale_factor;    /* mfLogScaleFactor */
    int32_t n_scale_levels;      /* mnScaleLevels */
    const float* scale_factors;  /* [n_scale_levels] mvScaleFactors */
} dcs_frustum_frame;
/* pos / normal [n][3] (GetWorldPos, GetNormal), min_dist / max_dist [n] (mfMinDistance, mfMaxDistance: the 0.8 / 1.2
   invariance factors are applied inside, MapPoint.cc:411-421), candidate [n] or NULL (0 = skip: bad / already matched).
   Outputs [n]: in_view (mbTrackInView), cam (mTrackProjCamera), u, v (mTrackProjX/Y), view_cos (mTrackViewCos), level
   (mnTrackScaleLevel = PredictScale, MapPoint.cc:440-455), radius = RadiusByViewingCos(view_cos) [* th when th != 1] *
   scale_factors[level] -- i.e. the valid / cam / u / v / radius / level -+ 1 columns of dcs_proj_queries. */
int  dcs_is_in_frustum(const dcs_frustum_frame* frame, int n, const float* pos, const float* normal, const float* min_dist,
                       const float* max_dist, const uint8_t* candidate, float viewing_cos_limit, float th, uint8_t* in_view,
                       int32_t* cam, float* u, float* v, float* view_cos, int32_t* level, float* radius);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-340), batched: map point p owns the descriptors
   pool[idx[off[p] .. off[p+1])] (the rows the reference gathers from its observations, in that order); best[p] = position
   inside that list of the descriptor with the least median Hamming distance to the others (median = sorted[(int)(0.5 (N-1))]
   of its row including the zero self-distance, first minimum wins), or -1 for an empty list. */
int  dcs_distinctive_descriptors(const uint8_t* pool, int n_pool, const int32_t* off, const int32_t* idx, int n_points,
                                 int32_t* best);

int  dcs_match_bf(const uint8_t* q, const dcs_keypoint* q_kp, int nq,
                  const uint8_t* t, const dcs_keypoint* t_kp, int nt,
                  int th, float ratio, int check_ori, int32_t* match, int* n_matches);

/* HBM-resident batch: feature slots as written by dcs_orb_extract_batch_device (slot s at
   d_desc + s*cap*32, d_kp + s*cap, count d_n[s]). pair p matches slot pairs[2p] (queries) against
   slot pairs[2p+1] (train). Outputs d_match[p*cap + i], d_n_matches[p], and the best / second-best
   distances d_best_d / d_second_d [n_pairs*cap] (required; same slotting). */
int  dcs_match_bf_batch_device(const uint8_t* d_desc, const dcs_keypoint* d_kp, const int32_t* d_n, int cap,
                               const int32_t* d_pairs, int n_pairs, int th, float ratio, int check_ori,
                               int32_t* d_match, int32_t* d_n_matches, int32_t* d_best_d, int32_t* d_second_d,
                               void* stream);

/* faithful SearchByBoWCrossCam(F,cF,KF,cKF) (ORBmatcher.cc:162-294): greedy, sequential over the KF
   features of each shared vocabulary node (already-claimed F features are skipped, :216).
   Feature vectors = ascending node ids + CSR. match_f[j] = KF index or -1. */
int  dcs_search_by_bow(const uint8_t* desc_kf, const float* ang_kf, const uint8_t* kf_valid, int n_kf,
                       const uint8_t* desc_f, const float* ang_f, int n_f,
                       const int32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx, int kf_n_nodes,
                       const int32_t* f_nodes, const int32_t* f_off, const int32_t* f_idx, int f_n_nodes,
                       float ratio, int check_ori, int32_t* match_f, int* n_matches);

/* ------------------------------------------------------------------ cross-GPU feature exchange
   One process per GPU; frame pairs / streams are sharded with no data-path collective. The only exchange: every rank
   contributes its newest feature slots and receives everybody's, so that it can match its cameras against features
   extracted on the other GPUs -- the multi-GPU analogue of SearchByBoWCrossCam(curFrame, camS, KF, CAP) (reference
   src/Tracking.cc:822; the reference itself has no collective). RCCL over xGMI, bound at run time (librccl.so.1).
   Bootstrap like NCCL's: rank 0 calls dcs_comm_unique_id and hands the 128 bytes to the other ranks over whatever channel
   the host has (file, socket, MPI ...); every rank then calls dcs_comm_create. */
#define DCS_COMM_ID_BYTES 128
typedef struct dcs_comm dcs_comm;
int  dcs_comm_unique_id(uint8_t id[DCS_COMM_ID_BYTES]);
int  dcs_comm_create(const uint8_t id[DCS_COMM_ID_BYTES], int rank, int world, dcs_comm** out);
void dcs_comm_destroy(dcs_comm*);
int  dcs_comm_info(const dcs_comm*, int* rank, int* world);
/* d_kp [n_slots][cap], d_desc [n_slots][cap][32], d_n [n_slots] of this rank (device pointers, the extractor's slot layout)
   -> *_all [world * n_slots] ..., rank-major: slot s of rank r lands at r * n_slots + s. Asynchronous on `stream`
   (hipStream_t); three grouped ncclAllGather calls on the slot arrays in place, no packing. */
int  dcs_features_allgather(dcs_comm*, const dcs_keypoint* d_kp, const uint8_t* d_desc, const int32_t* d_n, int n_slots, int cap,
                            dcs_keypoint* d_kp_all, uint8_t* d_desc_all, int32_t* d_n_all, void* stream);

/* SearchByBoWCrossCam(KF1, c1, KF2, c2, vpMatches12) (ORBmatcher.cc:297-414, LoopClosing.cc:300): like dcs_search_by_bow, but
   both sides are key frames: valid1 / valid2 = "has a good MapPoint", a KF2 feature matched once stays claimed (vbMatched2,
   also when the rotation histogram later drops the match), best < TH_LOW is STRICT. match12[i] = KF2 feature (camera-local)
   whose MapPoint KF1 feature i takes, or -1. */
int  dcs_search_by_bow_kf(const uint8_t* desc1, const float* ang1, const uint8_t* valid1, int n1,
                          const uint8_t* desc2, const float* ang2, const uint8_t* valid2, int n2,
                          const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                          const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2,
                          float ratio, int check_ori, int32_t* match12, int* n_matches);

/* SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, camS) (ORBmatcher.cc:1253-1427, LocalMapping.cc:321) with
   CheckDistEpipolarLine (:74-91): features WITHOUT a MapPoint (free1 / free2) of camera camS, per shared vocabulary node, best
   candidate only: distance <= TH_LOW, not within 10 * sqrt(scale) px of the epipole (ex, ey), on the epipolar line x1' F12
   (3.84 * mvLevelSigma2[octave]); of equal distances the LAST one of the node list wins (`dist > bestDist` skips). The caller
   computes F12 and the epipole with its cv::Mat code (:1262-1268). match12[i] = camera-local KF2 feature or -1; the global
   index pairs of vMatchedPairs follow by adding the camera offsets (:1413-1424). */
typedef struct dcs_epipolar {
    float F12[9];                 /* row-major */
    float ex, ey;                 /* epipole in the second image */
    const float* kp1_x; const float* kp1_y;     /* [n1] pKF1->mvvkeysUnTemp[camS][i].pt */
    const float* kp2_x; const float* kp2_y;     /* [n2] */
    const int32_t* kp2_octave;    /* [n2] */
    const float* level_sigma2;    /* [n_levels] pKF2->mvLevelSigma2 */
    const float* scale_factors;   /* [n_levels] pKF2->mvScaleFactors */
    int32_t n_levels;
} dcs_epipolar;
int  dcs_search_for_triangulation(const uint8_t* desc1, const float* ang1, const uint8_t* free1, int n1,
                                  const uint8_t* desc2, const float* ang2, const uint8_t* free2, int n2,
                                  const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                                  const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2,
                                  const dcs_epipolar* epi, int check_ori, int32_t* match12, int* n_matches);

/* ------------------------------------------------------------------ local BA */
typedef struct dcs_ba_camera {
    double fx, fy, cx, cy;      /* e->fx.. (Optimizer.cc:561-564) */
    double ext[7];              /* rig -> camera extrinsic SE3Quat: tx,ty,tz,qx,qy,qz,qw (:565-571) */
    double adj[36];             /* row-major 6x6 mVertexSE3CamExtAdj (Cameras.cc:27-37) */
} dcs_ba_camera;

typedef struct dcs_ba_problem {
    int32_t n_poses, n_points, n_edges, n_cams;
    const double*  poses;        /* [P][7] tx,ty,tz,qx,qy,qz,qw world -> rig, ascending KF id */
    const uint8_t* pose_fixed;   /* [P] setFixed (Optimizer.cc:483,496) */
    const double*  points;       /* [L][3] ascending MapPoint id */
    const int32_t* edge_pose;    /* [E] */
    const int32_t* edge_point;   /* [E] at most one edge per (pose, point) pair */
    const int32_t* edge_cam;     /* [E] */
    const double*  obs;          /* [E][2] kpUn.pt */
    const double*  inv_sigma2;   /* [E] mvInvLevelSigma2[octave] */
    const dcs_ba_camera* cams;   /* [n_cams] */
    double  huber_delta;         /* sqrt(5.991); <= 0: no robust kernel in round 1 (BundleAdjustment with bRobust = false) */
    double  chi2_th;             /* 5.991 */
    int32_t iters1, iters2;      /* 5, 10; iters2 = 0: single round (BundleAdjustment, Optimizer.cc:70-248) */
} dcs_ba_problem;

typedef struct dcs_ba_result {
    double*  poses;          /* [P][7] */
    double*  points;         /* [L][3] */
    double*  edge_chi2;      /* [E] chi2 of the last error evaluation (NULL allowed) */
    uint8_t* edge_outlier;   /* [E] chi2 > chi2_th || depth <= 0 after optimisation (Optimizer.cc:653) */
    uint8_t* edge_level1;    /* [E] excluded after round 1 (:607-610) (NULL allowed) */
    int32_t  n_iters[2];     /* LM iterations per round */
    int32_t  n_trials[2];    /* linear solves per round */
    double   lambda[2];
    double   chi2_trace[32]; /* robust chi2 after each iteration */
    float    gpu_ms;         /* device time of the optimise phase */
} dcs_ba_result;

/* Optimizer::LocalBundleAdjustment numerics on a flat problem (host buffers).
   stop_flag (may be NULL) is polled between LM iterations and trials like g2o does.
   Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (Optimizer.cc:61-248) is the same edge type and solver with one
   round: iters1 = nIterations, iters2 = 0, huber_delta = sqrt(3.99) (:107) or <= 0 when bRobust is false, only fixId fixed.
   The window is as large as the caller makes it (Optimizer.cc:415-422 takes every covisible key frame): up to 42 free poses the reduced camera
   system is factored in one workgroup's registers, beyond that by a blocked LDL^T over the whole chip (one launch per 16 columns; tested against
   the oracle at 60 free poses and at a 200-key-frame / 20 000-point / 160 000-edge map). More than 1 365 free poses: DCS_ERR_UNSUPPORTED. */
int  dcs_ba_local(const dcs_ba_problem* prob, const volatile uint8_t* stop_flag, dcs_ba_result* res);

/* The same solver for n_problems INDEPENDENT problems at once -- BASELINE config C5: one LocalMapping thread per
   dual-camera stream, each issuing Optimizer::LocalBundleAdjustment (src/LocalMapping.cc:97-104 -> src/Optimizer.cc:407-696).
   Every kernel of an LM step covers all problems (one launch, blockIdx.y = problem; the reduced camera systems are factored
   on n_problems compute units at once) and the accept / reject logic of optimization_algorithm_levenberg.cpp:104-164 runs on
   the device, so the host never waits for a trial. dcs_ba_local IS this function with n_problems = 1: results are identical.
   stop_flags may be NULL, and so may any stop_flags[b]; a flag already set at entry leaves that problem untouched
   (estimates copied through, no outliers, zero iterations: Optimizer.cc:582-585). Problems may differ in every size.
   Bit-for-bit equality with a solo call holds as long as no problem of the call has more than 512 poses: such a problem moves the
   problems that share its stream group to the two-launch update + error path, whose sums are ordered differently (same values to ~1e-15
   relative; tests: 1e-9 on the translations). A problem whose pose-pair tables would exceed 1 GB / 4 M workgroups: DCS_ERR_UNSUPPORTED. */
int  dcs_ba_local_batch(int n_problems, const dcs_ba_problem* const* probs, const volatile uint8_t* const* stop_flags,
                        dcs_ba_result* const* results);
/* Parity tap for rows a14 / a15 (like dcs_orb_debug_level for the extraction stages): the blocks the solver holds after the FIRST
   linearisation of the problem -- computeActiveErrors, linearizeOplus (Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:123-161) and
   constructQuadraticForm with the robust kernel (g2o/core/base_binary_edge.hpp:55-120) at the initial estimates, no lambda added, no
   solve. Hpp [n_free][36] and bp [n_free][6] in free-pose order (pose_idx[p] = index of pose p or -1: fixed poses and poses without an
   edge), Hll [n_points][9] and bl [n_points][3], Hpl [n_edges][18] = 6 x 3 row-major (pose rows, point columns; zero for edges of fixed
   poses). Caller-owned arrays sized for n_poses / n_points / n_edges. */
int  dcs_ba_debug_linearize(const dcs_ba_problem* prob, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, int32_t* pose_idx, int* n_free);
/* Measurement hook for bench.py (per calling thread): on != 0 makes the following dcs_ba_local[_batch] calls of this thread
   bracket every launch of the reduced-camera-system factorisation (k_ldlt_mfma, the reference's LinearSolverEigen::solve,
   Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124) and every LM step with hipEvents on the solver's stream.
   out (may be NULL) receives and resets {factorisation us, launches timed, step us, steps timed}. */
int  dcs_ba_timing(int on, double out[4]);

/* Optimizer::PoseOptimization (src/Optimizer.cc:250-405) for a batch of independent frames (one per stream / camera
   rig); the whole 4-round Levenberg-Marquardt procedure of a frame runs inside one workgroup, no host round trips.
   Frame f owns the edges edge_off[f] .. edge_off[f+1] (features with a MapPoint, ascending feature index).
   Poses agree with the reference's arithmetic to ~1e-11, flags and inlier counts exactly; a round's LM iteration count can be one off where
   two trials differ by rounding only (the convergence plateau). Option DCS_POSE_EXACT_EDGE = 1 selects the build that forms every edge's
   camera-frame point and residual with the reference's own operations: such rounds become about half as frequent, for 17-30 % more time. */
typedef struct dcs_pose_problem {
    int32_t n_frames, n_cams;
    const double*  poses;        /* [F][7] pFrame->mTcw (dcs_pose_from_matrix) */
    const int32_t* edge_off;     /* [F+1] */
    const double*  xw;           /* [E][3] MapPoint::GetWorldPos */
    const double*  obs;          /* [E][2] mvTotalKeysUn[i].pt */
    const double*  inv_sigma2;   /* [E] mvInvLevelSigma2[octave] */
    const int32_t* edge_cam;     /* [E] keypointToCam[i] */
    const dcs_ba_camera* cams;   /* [n_cams] */
    double  huber_delta;         /* (float)sqrt(5.991) (:284) */
    float   chi2_th[4];          /* {5.991f x 4} (:352), compared in float like the reference (:375-377) */
    int32_t its[4];              /* {10,10,10,10} (:354) */
} dcs_pose_problem;

typedef struct dcs_pose_result {
    double*  poses;          /* [F][7] */
    uint8_t* outlier;        /* [E] pFrame->mvbOutlier of the edge's feature */
    int32_t* n_inliers;      /* [F] the reference's return value (0 when fewer than 3 correspondences: pose untouched) */
    double*  edge_chi2;      /* [E] NULL allowed */
    int32_t* n_iters;        /* [F][4] LM iterations per round, NULL allowed */
} dcs_pose_result;

int  dcs_pose_optimization(const dcs_pose_problem* prob, dcs_pose_result* res);

/* ---- the steady-state per-frame chain of the Tracking thread, device-resident and batched over frames (one per stream / rig):
   Tracking::SearchLocalPoints (src/Tracking.cc:1617-1680) = Frame::isInFrustum for every local map point (Frame.cc:244-312, viewing
   cosine limit 0.5) + ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th) (ORBmatcher.cc:539-624), then
   Optimizer::PoseOptimization (src/Optimizer.cc:250-405; Tracking.cc:1321) over every feature that holds a map point afterwards.
   The three stages are dcs_is_in_frustum -> dcs_search_by_projection -> dcs_pose_optimization with the intermediate arrays kept in HBM:
   the frustum stage writes the search's queries (valid = in view, camera, u, v, radius, level -+ 1), the search's assignment
   (F.mvpMapPoints[bestIdx] = pMP, :617) becomes the optimiser's edge list (ascending feature index, Optimizer.cc:288-350: obs = the
   undistorted key point, Xw = the map point's position as double, information = mvInvLevelSigma2[octave], camera = keypointToCam[i]).
   Results equal the three calls issued one after the other (tests/test_gpu_track.py). */
typedef struct dcs_track_frame {
    dcs_proj_frame    features;     /* the frame: key points, descriptors, grid; taken[i] = mvpMapPoints[i] && Observations() > 0 before the call */
    const uint8_t*    has_point;    /* [N] mvpMapPoints[i] != NULL before the call (NULL: same as features.taken) */
    const float*      point_xw;     /* [N][3] GetWorldPos of the map point feature i holds (read where has_point) */
    dcs_frustum_frame view;         /* the matrices of the pose guess exactly as for dcs_is_in_frustum; view.n_cams == features.n_cams */
    const double*     pose;         /* [7] mTcw: the optimiser's initial estimate (dcs_pose_from_matrix) */
    int32_t           n_points;     /* local map points offered to the search */
    const float*      pos;          /* [n_points][3] */
    const float*      normal;       /* [n_points][3] */
    const float*      min_dist;     /* [n_points] */
    const float*      max_dist;
    const uint8_t*    candidate;    /* [n_points] or NULL: 0 = skip (bad, or already matched in this frame: Tracking.cc:1625-1640) */
    const uint8_t*    desc;         /* [n_points][32] MapPoint::GetDescriptor */
} dcs_track_frame;
typedef struct dcs_track_params {
    float   viewing_cos_limit;      /* 0.5 (Tracking.cc:1655) */
    float   th;                     /* 1; 3 / 5 right after a relocalisation (Tracking.cc:1667-1672) */
    int32_t th_high;                /* TH_HIGH = 100 */
    float   nn_ratio;               /* 0.8 (:1664) */
    int32_t n_levels;
    const float* inv_level_sigma2;  /* [n_levels] mvInvLevelSigma2 */
    int32_t n_cams;
    const dcs_ba_camera* cams;      /* [n_cams] as for dcs_pose_optimization */
    double  huber_delta;            /* (float)sqrt(5.991) */
    float   chi2_th[4];
    int32_t its[4];
} dcs_track_params;
typedef struct dcs_track_result {
    double*  poses;                   /* [F][7] */
    int32_t* n_inliers;               /* [F] PoseOptimization's return value */
    int32_t* n_matches;               /* [F] SearchByProjection's return value */
    int32_t* const* match_of_point;   /* [F] -> [n_points]: global feature index the map point was assigned to, or -1 */
    int32_t* const* point_of_feature; /* [F] -> [N]: >= 0 local map point assigned by this call, -2 the feature keeps the map point it held, -1 none */
    uint8_t* const* outlier;          /* [F] -> [N]: mvbOutlier[i] after the optimisation (0 for features without a map point) */
} dcs_track_result;
int  dcs_track_local_map(int n_frames, const dcs_track_frame* frames, const dcs_track_params* prm, dcs_track_result* res);

/* ---- the same chain FED FROM THE EXTRACTOR'S SLOTS (round 5): the frame's key points and descriptors stay where
   dcs_orb_extract_batch_device left them in HBM one call earlier; what the Frame constructor does with them on the host
   (src/Frame.cc:141-196: concatenate the cameras' key points into mvTotalKeysUn, PosInGrid + the 64 x 48 grid per camera) runs on
   the device. Two modes, the two per-frame stages of the reference's steady state:
     mode 0  Tracking::TrackLocalMap's search + optimisation: exactly dcs_track_local_map (isInFrustum -> SearchByProjection(F,
             local map points, th) -> PoseOptimization);
     mode 1  Tracking::TrackWithMotionModel (src/Tracking.cc:1384-1427): ORBmatcher(0.8, true).SearchByProjection(mCurrentFrame,
             mpLastFrame, th) = SearchByProjectionOnCam per camera (src/ORBmatcher.cc:954-1113: project the last frame's map points with
             the predicted pose, window th * mvScaleFactors[last octave], levels octave -+ 1, best distance <= TH_HIGH, a rotation
             histogram per camera), then PoseOptimization over the new matches. The queries are the last frame's features that hold
             a good map point, in ascending feature order: pos = GetWorldPos, desc = MapPoint::GetDescriptor, q_cam = keypointToCam[i],
             q_octave / q_angle = the last frame's key point. mvpMapPoints of the current frame is empty in this stage (Tracking.cc:1396).
             PRECONDITION: every query's map point has Observations() > 0 (true of the points Tracking keeps in the last frame outside
             localisation-only mode): a matched feature is then taken for the queries that follow, as :1046-1048 skips it. For a point WITHOUT
             observations the reference lets a later query overwrite the feature and counts the match twice (:1062); that is not reproduced.
             A query with zs == 0 is treated as not visible (the reference continues with inf / NaN coordinates, which fail its bounds test too).
   Frame::UndistortKeyPoints (Frame.cc:410-441) runs on the device too: with K and dist given (the rig file's fx fy cx cy and k1 k2 p1 p2 k3;
   Dual-LenaCV.yaml has k1 = -0.37) every key point goes through cv::undistortPoints' arithmetic (dcs_undistort_points below) when the camera's
   k1 != 0, exactly the reference's test (:414); dist == NULL: the key points are taken as they are. Feature g of the outputs = the frame's compact index: camera c's features at [off_c, off_c + n_c) with
   n_c = min(count of slot first_slot + c, cap) -- n_features reports n_c. Map-side arrays come from the host (they live in the reference's
   map). One synchronisation, at the end. `stream` = the stream the extraction was enqueued on (the chain waits for it; NULL = legacy stream). */
typedef struct dcs_dev_frame {
    int32_t n_cams, cap, first_slot;    /* camera c = slot first_slot + c of the arrays below */
    const dcs_keypoint* d_kp;           /* DEVICE [slots][cap] */
    const uint8_t*      d_desc;         /* DEVICE [slots][cap][32] */
    const int32_t*      d_n;            /* DEVICE [slots] */
    const float* min_x; const float* min_y; const float* grid_w_inv; const float* grid_h_inv;   /* host [n_cams]: mvMinX, mvMinY, mvfGridElementWidthInv / HeightInv */
    const float* K;                     /* host [n_cams][4] fx, fy, cx, cy of the rig file (read with dist) */
    const float* dist;                  /* host [n_cams][5] k1, k2, p1, p2, k3 (0 when the file has no k3), or NULL: no undistortion */
} dcs_dev_frame;
typedef struct dcs_track_dev_frame {
    dcs_dev_frame     features;
    dcs_frustum_frame view;         /* the matrices of the pose guess (mode 1 reads Rsw, tsw, fx .. cy, min / max, scale_factors) */
    const double*     pose;         /* [7] */
    int32_t           n_held;       /* entries of the three arrays below (the frame's feature count as an earlier stage reported it; 0 with NULL arrays: nothing held) */
    const uint8_t*    taken;        /* host [n_held] as dcs_proj_frame.taken */
    const uint8_t*    has_point;    /* host [n_held] or NULL (= taken) */
    const float*      point_xw;     /* host [n_held][3] */
    int32_t           n_points;     /* queries: local map points (mode 0) / the last frame's features with a map point (mode 1) */
    const float* pos; const float* normal; const float* min_dist; const float* max_dist; const uint8_t* candidate;   /* normal .. candidate: mode 0 only */
    const uint8_t* desc;            /* [n_points][32] */
    const int32_t* q_cam; const int32_t* q_octave; const float* q_angle;   /* mode 1 only, [n_points] */
} dcs_track_dev_frame;
typedef struct dcs_track_dev_result {
    dcs_track_result r;             /* arrays per frame sized for n_cams * cap features / n_points queries */
    int32_t* n_features;            /* features per camera as assembled, frame after frame: frame k's n_cams counts start at the sum of n_cams of frames 0..k-1
                                       (= [F][n_cams] when every frame has the same number of cameras) */
} dcs_track_dev_result;
/* cv::undistortPoints(xy, out, K, dist, cv::Mat(), K) on n float points, the call of Frame::UndistortKeyPoints and Frame::ComputeImageBounds (Frame.cc:430,
   468): OpenCV 3.3 / 3.4.0's five fixed-point iterations in double. dist5[0] == 0: copied through like the reference does (:414). Pure host helper (the
   host-buffer chain dcs_track_local_map takes undistorted key points; the image bounds are four such points); the device chain runs the same arithmetic. */
int  dcs_undistort_points(int n, const float* xy, const float K4[4], const float dist5[5], float* out);
int  dcs_track_frame_device(int n_frames, const dcs_track_dev_frame* frames, const dcs_track_params* prm, int mode, int check_orientation,
                            dcs_track_dev_result* res, void* stream);


/* Cameras::setExtrinsics (Cameras.cc:17-37) + Converter::toSE3Quat/toMatrix6d: float 4x4 (row-major)
   -> ext[7], adj[36]. exact = 0: reference matrix [[R, R t^],[0, R]] in float;
   exact = 1: g2o's SE3Quat::adj() [[R,0],[t^R,R]]. Pure host helper. */
int  dcs_rig_adjoint(const float T44[16], int exact, double ext7[7], double adj36[36]);
/* Converter::toSE3Quat (Converter.cc:58-68) / toCvMat(SE3Quat) (:70-74): 4x4 float <-> pose[7] */
int  dcs_pose_from_matrix(const float T44[16], double pose7[7]);
int  dcs_pose_to_matrix(const double pose7[7], float T44[16]);

/* ---------------------------------------------------------------------------------------------------------------
   BoW front half: DBoW2 vocabulary tree in HBM, transform -> BowVector + FeatureVector, L1 score.
   Replaces ORBVocabulary::transform as called by Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:393-406,
   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1151-1283) and ORBVocabulary::score (ScoringObject.cpp:23-67) as called by
   KeyFrameDatabase (src/KeyFrameDatabase.cc:250-372). */
typedef struct dcs_vocab dcs_vocab;

/* The vocabulary as the columns of the reference's text format (loadFromTextFile, TemplatedVocabulary.h:1362-1446): row i
   describes node i + 1 (node 0 = root): parent id, leaf flag, 32-byte descriptor, weight. scoring / weighting are the DBoW2
   enums (L1_NORM = 0 ..., TF_IDF = 0, TF, IDF, BINARY). A row's leaf flag must agree with "has no children". */
int  dcs_vocab_create(int k, int L, int scoring, int weighting, int n_rows, const int32_t* parent, const uint8_t* is_leaf,
                      const uint8_t* desc, const double* weight, dcs_vocab** out);
void dcs_vocab_destroy(dcs_vocab* v);
int  dcs_vocab_info(const dcs_vocab* v, int* k, int* L, int* n_nodes, int* n_words);

/* transform(features, BowVector, FeatureVector, levelsup) for n_images images whose descriptors sit in HBM in the extractor's
   slotted layout (image i: d_desc + i*cap*32, count d_n[i]); cap <= 4096. Outputs, slotted the same way:
     d_word / d_node / d_weight [n_images*cap]   per feature: word id, node id at level L - levelsup, word weight
                                                 (-1 / -1 for stopped words, weight <= 0)
     d_bow_word / d_bow_val [n_images*cap], d_bow_n[n_images]            BowVector: ascending word ids, normalised values
     d_fv_node [n_images*cap], d_fv_off [n_images*(cap+1)], d_fv_idx [n_images*cap], d_fv_n[n_images]
                                                 FeatureVector as CSR: ascending node ids, feature indices ascending per node
   (the input format of dcs_search_by_bow / dcs_hamming_knn2_grouped). Values are bit-identical to the reference's doubles. */
int  dcs_bow_transform_device(const dcs_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int n_images, int cap, int levelsup,
                              int32_t* d_word, int32_t* d_node, double* d_weight, int32_t* d_bow_word, double* d_bow_val,
                              int32_t* d_bow_n, int32_t* d_fv_node, int32_t* d_fv_off, int32_t* d_fv_idx, int32_t* d_fv_n,
                              void* stream);
/* one image from host memory; arrays sized n (fv_off n + 1); word / node may be NULL */
int  dcs_bow_transform(const dcs_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node,
                       int32_t* bow_word, double* bow_val, int* n_words, int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx,
                       int* n_nodes);
/* L1Scoring::score of one BowVector against n_db BowVectors stored as CSR (db_off[n_db+1]); score[n_db] in [0, 1] */
int  dcs_bow_score_l1(const int32_t* q_word, const double* q_val, int nq, const int32_t* db_off, const int32_t* db_word,
                      const double* db_val, int n_db, double* score);

/* ---------------------------------------------------------------------------------------------------------------
   KeyFrameDatabase of one camera (src/KeyFrameDatabase.cc:46-110: mvvInvertedFiles[c]) resident in HBM, and the query half of
   DetectLoopCandidatesForCam (:111-235) / DetectRelocalizationCandidates (:237-372). The database keeps the entries' BowVectors in
   order of insertion; an inverted list of the reference is "the live entries that hold the word, ascending entry id" (its
   push_back order), so it is never materialised. One handle per camera; not thread-safe per handle (the reference holds mMutex). */
typedef struct dcs_kfdb dcs_kfdb;
int  dcs_kfdb_create(dcs_kfdb** out);
void dcs_kfdb_destroy(dcs_kfdb* db);
/* KeyFrameDatabase::add (:63-71): the BowVector (word ids strictly ascending, as a std::map iterates) of the new key frame;
   *entry_id = its index (0, 1, 2 ... in order of insertion) */
int  dcs_kfdb_add(dcs_kfdb* db, const int32_t* word, const double* val, int n, int* entry_id);
/* KeyFrameDatabase::erase (:73-97): the entry leaves every inverted list; ids of the others do not change */
int  dcs_kfdb_erase(dcs_kfdb* db, int entry_id);
int  dcs_kfdb_clear(dcs_kfdb* db);                      /* :99-108 */
int  dcs_kfdb_size(const dcs_kfdb* db, int* n_entries); /* entries ever added since the last clear (erased ones included) */
/* For every entry (arrays of dcs_kfdb_size elements): common = words shared with the query = the value the walk over the
   inverted files leaves in mnLoopWords / mnRelocWords (:128-149, :257-272; 0 for erased entries), first_word = the smallest shared
   word id (-1 if none) -- entries enter lKFsSharingWords in (first_word, entry id) order -- and score = (float)L1Scoring::score
   (query, entry), the `float si = mpVoc->score(...)` of :175, :305 (bit-identical to dcs_bow_score_l1). The thresholds
   (0.8f * maxCommonWords, minScore), the covisibility accumulation and the 0.75f * bestAccScore cut work on these three arrays
   and the caller's covisibility graph: KeyFrameDatabase::Detect*Candidates in orb-slam2-dualcam_amd/host/KeyFrameDatabase.h. */
int  dcs_kfdb_query(dcs_kfdb* db, const int32_t* q_word, const double* q_val, int nq, int32_t* common, int32_t* first_word, float* score);

#ifdef __cplusplus
}
#endif
#endif /* DCS_ABI_H */
