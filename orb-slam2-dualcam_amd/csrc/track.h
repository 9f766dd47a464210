// track.h -- internal seam between proj_kernels.hip (frustum + projection search + the tracking chain) and ba_solver.hip (k_pose_opt):
// PoseOptimization on arrays that already live in HBM, enqueued on the caller's stream (no copies, no synchronisation).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcs_abi.h"

namespace dcs {

constexpr int kPoseMaxCams = 4;      // cameras of a rig the optimiser kernels hold by value (== kMaxCams of ba_solver.hip)

struct PoseOptDevice {
    const double* poses;          // [F][7]
    const int32_t* edge_off;      // [F] first edge of frame f
    const int32_t* edge_cnt;      // [F] its number of edges (NULL: edge_off is a CSR of F + 1 entries)
    const double *xw, *obs, *w;   // [E][3], [E][2], [E]
    const int32_t* cam;           // [E]
    double huber; float chi2_th[4]; int its[4];
    double* err; uint8_t* level;  // scratch [E][2], [E]
    double* out_poses; uint8_t* outlier; int32_t* n_inliers; double* edge_chi2 /* may be NULL */; int32_t* n_iters /* [F][4], may be NULL */;
};
int launch_pose_opt_device(const PoseOptDevice& p, const dcs_ba_camera* cams, int n_cams, int n_frames, int max_edges_bound /* no frame has more edges than this */, hipStream_t st);

}  // namespace dcs
