// config.h -- the library's tuning / A-B switches as OPTIONS (round 5; the round-4 verdict counted 39 getenv calls across csrc/, most of them
// read once into function-local statics: two handles in a process could not differ and every switch needed a child process to test).
// An option has a name (the environment variable it used to be), an integer value and a default. The environment is read ONCE per process, in
// config.cpp -- the only getenv of the library -- to seed the process-wide values; dcs_option_set() changes them at run time. Handle-less entry
// points (matcher, solver, tracking) read the process-wide value per call; an extractor handle copies what it needs when it is created
// (so: set, create handle A, set again, create handle B -> two handles that differ).
#pragma once
#include <cstdint>

namespace dcs {

#define DCS_OPTIONS(X)                                                                                                                   \
    /* name                       default   what */                                                                                      \
    X(ORB_FUSED_BLUR,             -1)   /* -1 auto (fused), 0 separate blur kernels + k_describe<false>, 1 k_describe<FUSED> */          \
    X(ORB_FAST_SPLIT,              0)   /* k > 0: FAST of levels [0, k) starts on a stream of its own next to the resize chain */        \
    X(ORB_EMIT,                   -1)   /* 0 resize chain; n > 0 cells of levels [0, n) emit the next level whatever the batch; -1 auto */ \
    X(ORB_NO_OVERLAP,              0)   /* 1: every extraction kernel alone on the main stream (profiling) */                            \
    X(ORB_DENSE_CAP,               0)   /* > 0: candidates the handle's dense buffer holds (test hook: provokes the overflow path) */    \
    X(ORB_HOST_CHUNK,             -1)   /* images per chunk of the host-image pipeline; 0 one-shot; -1 auto */                           \
    X(ORB_HOST_DIRECT,             1)   /* 0: page-locked frames are packed like pageable ones */                                        \
    X(ORB_HOST_TRACE,              0)   /* 1: per-stage host times of dcs_orb_extract_batch on stderr */                                 \
    X(ORB_SMALL_GRAPH,            -1)   /* 0 / 1: one-or-two-image calls replayed as an executable graph; -1 auto (on) */                \
    X(ORB_STAGING_THREADS,         0)   /* > 0: packing threads of the host-image pipeline; 0 auto */                                    \
    X(FAST_HW_PROBE_FAIL,          0)   /* 1: new handles behave as if the start-up probe had failed (plain k_fast_cells forms) */       \
    X(FAST_EXACT,                  2)   /* bit 0: 40-byte LDS rows, bit 1: 44-byte rows */                                               \
    X(BLUR_FOLD,                  -1)   /* separate blur: 1 k_blur_fold, 0 k_blur + k_blur_edge_cols, -1 auto */                         \
    X(OCTREE_FORCE_GENERAL,        0)   /* 1: the sort-based k_octree for every task (test hook) */                                      \
    X(KNN2_I8,                     0)   /* 1: the i8 matrix-core matcher instead of the FP4 one */                                       \
    X(POSE_FAST,                   1)   /* 0: every frame to the round-4 k_pose_opt */                                                   \
    X(POSE_EXACT_EDGE,             0)   /* 1: k_pose_opt2 forms every edge's point and residual with the oracle's own operations: rounds one LM iteration from the oracle 16.8 -> 9.2 % of batches, +17 % time (NOTES R6.3) */ \
    X(BA_SCHUR_WAVE,               1)   /* Schur launch: 1 = by group size (k_schur<28> for 1-2 problems, k_schur_w beyond), 0 / 2 = always the former / the latter */ \
    X(BA_TRACE,                    0)   /* 1: host time per phase of dcs_ba_local_batch on stderr */                                     \
    X(BA_FORCE_BLOCKED_LDLT,       0)   /* test hook: the n > 256 factorisation at small n */                                            \
    X(BA_GROUPS,                   0)   /* > 0: stream groups of a batch */                                                              \
    X(BA_PAIRS_SIDE,               1)   /* 0: pose-pair lists built in front of the first step instead of beside it */                   \
    X(BA_FUSED_UPDATE,             1)   /* 0: k_solve_update + k_error<1> as two launches */                                             \
    X(BA_GRAPH,                    0)   /* 1: an LM step replayed as an executable graph */                                              \
    X(BA_LOOKAHEAD,                0)   /* LM steps enqueued ahead of the progress word; 0 = auto (2, or 1 for batches of four or more problems) */ \
    X(BA_DL_STREAM,                1)   /* 0: results come down on the solver's stream */                                                \

enum Opt : int {
#define DCS_OPT_ENUM(name, def) OPT_##name,
    DCS_OPTIONS(DCS_OPT_ENUM)
#undef DCS_OPT_ENUM
    OPT_COUNT
};

long long opt(Opt o);                          // the process-wide value
int opt_find(const char* name);                // "DCS_ORB_EMIT" or "ORB_EMIT" -> index, -1 unknown
const char* opt_name(int i);                   // "DCS_..." form
void opt_set(int i, long long v);
long long opt_default(int i);

}  // namespace dcs
