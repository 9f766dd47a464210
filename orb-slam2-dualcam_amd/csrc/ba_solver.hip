// ba_solver.hip -- Optimizer::LocalBundleAdjustment numerics on gfx950 (f64).
//
// Replaces (reference files): src/Optimizer.cc:582-621, 645-658 (two LM rounds, outlier flags) and the
// g2o arithmetic they execute -- types/types_six_dof_expmap.cpp:109-169 (dual-camera edge error and
// Jacobians with extrinsic + "adjoint"), core/base_binary_edge.hpp:55-120 + robust_kernel_impl.cpp:78-91
// (Huber-weighted J^T W J blocks), core/block_solver.hpp:354-604 (Schur complement, lambda on every
// diagonal, landmark back-substitution), solvers/linear_solver_eigen.h:94-124 (LDL^T of the reduced
// camera system), core/optimization_algorithm_levenberg.cpp:61-189 (LM control), se3quat.h:223-257 (exp).
//
// Flat SoA problem in HBM, no graph objects. Launches per LM step, 7 (every list walk is a latency chain, so each
// reduction uses many short chunks with several loads in flight, and small kernels share launches):
//   k_begin         edge-parallel: what happens BETWEEN iterations (round change: outlier flags, level-1 set, robust kernel off;
//                   errors of a stale state), then the Jacobians -> per-edge H_pl (6x3) and per-edge pose / point contributions
//   k_reduce_pose   one launch: segmented sums of the pose blocks (37 chunks per free pose) and of the landmark blocks,
//                   CSR order, no atomics (reproducible)
//   per trial (5 launches + one 40-byte read-back):
//   k_prep          BD[e] = H_pl[e] (H_ll + lambda I)^-1 per edge, D^-1 and D^-1 b_l per landmark
//   k_schur         workgroup per pose pair over a precomputed (edge, edge) list: 28 partial sums x 36 entries (4 waves with 4
//                   accumulators per thread, or 16 waves for groups of <= 2 problems; same bits) + the reduced right-hand side
//   k_ldlt_mfma     LDL^T + both triangular solves of the reduced camera system in ONE workgroup: 16x16 tiles in the
//                   registers of 11 worker waves, trailing updates on the f64 matrix cores (v_mfma_f64_16x16x4_f64), the
//                   diagonal blocks factored one step ahead by a twelfth wave
//                   (k_ldlt_step / k_ldlt_back, one launch per block column:
//                   multi-launch fallback for n > 256)
//   k_solve_update  landmark back-substitution, push + manifold update of all estimates, computeScale partials
//   k_error<1>      edge-parallel residual + chi2 + Huber rho of the trial estimates; the problem's last block adds the block
//                   partials (and the scale partials) in index order and runs g2o's accept / reject logic ON THE DEVICE;
//                   the grid's last block publishes the step to the host's pinned progress word
// Every kernel covers a whole batch of problems (blockIdx.y); the host only enqueues steps and watches a pinned progress word.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

#include "common.h"
#include "config.h"
#include "track.h"

namespace dcs {

constexpr int kMaxCams = 4;
static_assert(kMaxCams == dcs::kPoseMaxCams, "track.h and ba_solver.hip disagree on the cameras per rig");
constexpr int kNB = 16;

struct DCam { double fx, fy, cx, cy, t[3], q[4], adj[36]; };
struct DCams { DCam c[kMaxCams]; };

// ------------------------------------------------------------------ small SE3 math (Eigen semantics)
__host__ __device__ inline void cross3(const double a[3], const double b[3], double o[3])
{ o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }

// q = (x, y, z, w); Eigen QuaternionBase::_transformVector
__host__ __device__ inline void qrot(const double q[4], const double v[3], double o[3])
{
    double uv[3]; cross3(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3]; cross3(q, uv, c);
    o[0] = v[0] + q[3] * uv[0] + c[0]; o[1] = v[1] + q[3] * uv[1] + c[1]; o[2] = v[2] + q[3] * uv[2] + c[2];
}
__host__ __device__ inline void qmul(const double a[4], const double b[4], double o[4])
{
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__host__ __device__ inline void qnormalize(double q[4])      // SE3Quat::normalizeRotation
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__host__ __device__ inline void qtoR(const double q[4], double R[9])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void qfromR(const double R[9], double q[4])
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        // Eigen: i = the largest diagonal entry (first wins ties), j = i + 1, k = j + 1 cyclic. Written out per case: indexing a local array
        // with i at run time sends it to scratch memory on the device.
        const bool b1 = R[4] > R[0];
        const bool b2 = R[8] > (b1 ? R[4] : R[0]);
        if (b2) {                                             // i = 2, j = 0, k = 1
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[2] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
        } else if (b1) {                                      // i = 1, j = 2, k = 0
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[1] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
        } else {                                              // i = 0, j = 1, k = 2
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[0] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
        }
    }
}
// pose layout: tx,ty,tz,qx,qy,qz,qw
__host__ __device__ inline void pose_map(const double* T, const double X[3], double o[3])
{ qrot(T + 3, X, o); o[0] += T[0]; o[1] += T[1]; o[2] += T[2]; }

__host__ __device__ inline void cam_point(const double* pose, const double* X, const DCam& c, double pc[3])
{
    double pm[3];
    pose_map(pose, X, pm);
    qrot(c.q, pm, pc);
    pc[0] += c.t[0]; pc[1] += c.t[1]; pc[2] += c.t[2];
}

// exp(update) * T   (VertexSE3Expmap::oplusImpl, SE3Quat::exp, SE3Quat::operator*)
__host__ __device__ inline void pose_oplus(const double* T, const double* u, double* out)
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0 ? 1.0 : 0.0); R[i] = I + a * O[i] + b * O2[i]; V[i] = I + b * O[i] + c * O2[i]; }
    }
    double qe[4], te[3];
    qfromR(R, qe); qnormalize(qe);
    for (int i = 0; i < 3; ++i) te[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    double rt[3]; qrot(qe, T, rt);
    out[0] = te[0] + rt[0]; out[1] = te[1] + rt[1]; out[2] = te[2] + rt[2];
    double qo[4]; qmul(qe, T + 3, qo); qnormalize(qo);
    out[3] = qo[0]; out[4] = qo[1]; out[5] = qo[2]; out[6] = qo[3];
}

__device__ inline void inv3(const double m[9], double o[9])     // Eigen cofactor inverse
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (c00 * m[0] + c10 * m[1] + c20 * m[2]);
    o[0] = c00 * id; o[3] = c10 * id; o[6] = c20 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// deterministic block sum (256 threads): result valid in thread 0
__device__ inline double block_sum_256(double v, double* s /*[256]*/)
{
    s[threadIdx.x] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    return s[0];
}

// ------------------------------------------------------------------ batched problem descriptors
// Every BA kernel is launched once for a whole batch of independent problems: blockIdx.y = problem, blockIdx.x runs up
// to the largest problem's block count (smaller problems exit). The Levenberg-Marquardt state of a problem (BaCtl)
// lives in HBM and is advanced ON THE DEVICE -- by the finisher of the trial's error kernel and by k_round_ctl -- so
// the host enqueues identical "steps" back to back and never waits for a trial (it only watches a progress word in
// pinned memory). A kernel whose problem is not in a state that needs it returns at once.
enum : int { ST_NEW_ITER = 0, ST_RETRY = 1, ST_ROUND_END = 2, ST_DONE = 3 };

struct BaCtl {
    int state, round, it, qmax, nBad, cur, errors_current, robust, trace, stopped, n_active, pad0;
    double mult, ni, currentChi, iniChi, maxdiag, ok, scale, pad1;
    int n_iters[2], n_trials[2];
    double lambda[2], chi2_trace[32];
};

struct BaProb {
    int P, L, E, np, n, n_pad, ld, n_pairs, nblk, nb_pts, nb_pose, use_reg;
    int iters[2], robust0, pad;
    double delta, chi2_th;
    double *poses[2], *points[2];                  // estimates are double-buffered: a trial writes [cur ^ 1], accept flips cur
    const int32_t *epose, *epoint, *ecam;          // [E]
    const double *obs, *w;                         // [E][2], [E]
    uint8_t *active, *flag, *level1;               // [E]
    double *err, *chi2;                            // [E][2], [E]
    double *Hpl[2], *BD, *cpose[2], *cpoint[2];    // [E][18], [E][18], [E][27], [E][9]; the linearisation is double-buffered like the estimates: [b] belongs to poses[b] / points[b]
                                                   // (k_update_error linearises the trial estimates it has just evaluated; [1] == [0] when a group runs the six-launch step)
    double *Hll[2], *bl[2], *Dinv, *db, *xl;       // per landmark; H_ll / b_l / pt_active belong to a linearisation: [b] like Hpl[b]
    double *Hpp, *bp, *bsch, *xp;                  // per free pose
    double *S, *W, *Dgf;                           // reduced camera system (ld x ld; n > 256: (n_pad + 16) x ld with the right-hand side below), the panels L D of two block columns, U^-1 and 1 / d of the diagonal blocks
    const int32_t *pose_idx, *pt_off, *pt_edges, *pt_pi, *ps_off, *ps_edges;   // pt_pi[k] = free-pose index (or -1) of edge pt_edges[k]
    const int32_t* pair_ij;                          // [n_pairs][2]: every pair i1 <= i2 of free poses, the np diagonal pairs first
    int32_t *pair_off, *pair_e;                      // lists of the pairs, BUILT ON THE DEVICE (k_pairs_*); pair_e: (e1, e2) interleaved, one 8-byte load per entry
    uint32_t* pt_bits;                               // [np][pt_words] bit l of row i: free pose i observes point l (zeroed region)
    int32_t* edge_of;                                // [np][L] the edge of (free pose, point), valid where the bit is set
    int pt_words, pad2;
    double *partial, *scale_part, *maxd_part;      // block partials: chi2, computeScale, max |diagonal| (np + nb_pts entries)
    uint8_t* pt_active[2];
    unsigned* ticket;                              // [0] error kernels, [1] k_reduce_pose, [2] k_begin
    const DCams* cams;
    double *out_poses, *out_points;
};

__device__ __forceinline__ void load_cams(DCams* dst, const DCams* src)
{
    const double* s = reinterpret_cast<const double*>(src);
    double* d = reinterpret_cast<double*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(DCams) / sizeof(double)); i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// ------------------------------------------------------------------ kernels
// residuals (computeError), chi2, robust rho0. Block partial sums go to partial[]; the last block of the problem to
// finish (device-scope ticket) adds them in index order, so the total is reproducible run to run. That finisher also
// owns the LM control flow (OptimizationAlgorithmLevenberg::solve, optimization_algorithm_levenberg.cpp:61-164):
//   TRIAL = 0  computeActiveErrors at the top of an LM iteration whose errors are stale (first iteration of a round, or
//              after a rejected trial): evaluated at the current estimates, sets currentChi
//   TRIAL = 1  errors of the trial estimates [cur ^ 1], + computeScale, then accept / reject, lambda update, termination
// Progress for the host, ONE pinned 64-bit word {finished problems : step} per group of problems, written with a system-scope release
// store, so the host decides from a single acquire load and "all done" can never be seen ahead of the results. Only the threads that
// END something publish -- no grid-wide arrival counter, no extra barrier in the other blocks (a grid-wide ticket in every block cost
// 2 us per kernel):
//   * a problem ends in k_begin: its finisher (the problem's last block, which has acquired every other block's release) bumps
//     grid_ticket[2] = finished problems of the group and publishes it under the current step number;
//   * a step ends in k_error<1>: the problems still live (B - finished: none of them can be in ROUND_END there) each have exactly one
//     finisher thread, the last of THOSE to arrive (grid_ticket[0]) advances grid_ticket[1] = steps finished and publishes.
// The counters live on the device so that every launch of a step has the same arguments (the step can be replayed as one hipGraph).
__device__ __forceinline__ void publish_word(int* __restrict__ h_progress, unsigned done, unsigned step)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(h_progress), ((unsigned long long)done << 32) | step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_step_end(int B, int* __restrict__ h_progress, unsigned* __restrict__ grid_ticket)      // one thread per live problem
{
    const unsigned done = __hip_atomic_load(grid_ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned live = (unsigned)B - done;
    if (live > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(grid_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != live - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(grid_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned step = grid_ticket[1] + 1;
    grid_ticket[1] = step;
    publish_word(h_progress, done, step);
}
__device__ __forceinline__ void publish_problem_end(int* __restrict__ h_progress, unsigned* __restrict__ grid_ticket)          // the finisher of a problem that ended
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(grid_ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Several problems may end in the same launch: every finisher publishes the count it READS and repeats until the count has not moved
    // after its store has completed -- the store of the largest count is then the last one to land, whatever order the finishers ran in
    // (without the re-check a batch whose problems all end in one launch could leave the word one problem short for ever: the no-op
    // steps behind it publish nothing).
    const unsigned step = grid_ticket[1];
    unsigned v;
    do {
        v = __hip_atomic_load(grid_ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish_word(h_progress, v, step);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } while (__hip_atomic_load(grid_ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != v);
}

// accept / reject of a trial (optimization_algorithm_levenberg.cpp:104-164): ONE thread of the problem, after the chi2 of the trial
// estimates (tot) and computeScale (sc) are known. Always returns true.
__device__ __forceinline__ bool lm_accept(const BaProb& pb, BaCtl& ctl, const volatile int* __restrict__ stop_words, double tot, double sc)
{
    ++ctl.n_trials[ctl.round];
    double tempChi = tot;
    if (ctl.ok == 0.0) tempChi = 1.7976931348623157e308;
    double rho = ctl.currentChi - tempChi;
    const double scale = sc + 1e-3;
    rho /= scale;
    double mult = ctl.mult, ni = ctl.ni, currentChi = ctl.currentChi;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        mult *= fmax(1. / 3., alpha);
        ni = 2; currentChi = tempChi;
        ctl.cur ^= 1; ctl.errors_current = 1;       // discardTop: the trial estimates become the current ones
    } else {
        mult *= ni; ni *= 2;
        ctl.errors_current = 0;                     // pop: the other buffer still holds the estimates; err / chi2 stay stale like g2o's
    }
    ctl.mult = mult; ctl.ni = ni; ctl.currentChi = currentChi; ctl.scale = sc;
    const int qmax = ++ctl.qmax;
    const bool stop = stop_words && __hip_atomic_load(const_cast<const int*>(stop_words + blockIdx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    if (stop) ctl.stopped = 1;
    if (rho < 0 && qmax < 10 && !stop) { ctl.state = ST_RETRY; return true; }
    const int round = ctl.round;
    ++ctl.n_iters[round];
    if (ctl.trace < 32) ctl.chi2_trace[ctl.trace++] = currentChi;
    ctl.lambda[round] = 1e-5 * ctl.maxdiag * mult;
    bool term = (qmax == 10 || rho == 0);           // Terminate
    if (!term) {
        if ((ctl.iniChi - currentChi) * 1e3 < ctl.iniChi) ++ctl.nBad; else ctl.nBad = 0;
        if (ctl.nBad >= 3) term = true;
    }
    const int it = ++ctl.it;
    if (term || stop || it >= pb.iters[round]) ctl.state = ST_ROUND_END;
    else { ctl.state = ST_NEW_ITER; ctl.qmax = 0; ctl.iniChi = currentChi; }
    return true;
}

template <int TRIAL>
__device__ __forceinline__ bool error_body(const BaProb& pb, BaCtl& ctl, const volatile int* __restrict__ stop_words, double* s /*[256]*/, DCams& cams, bool& last)
{   // returns true in the ONE thread of the problem that ran the accept / reject logic
    if ((int)blockIdx.x >= pb.nblk) return false;
    if (TRIAL ? ctl.state > ST_RETRY : (ctl.state != ST_NEW_ITER || ctl.errors_current)) return false;
    const int robust = ctl.robust, E = pb.E;
    const double delta = pb.delta;
    const double* __restrict__ poses = pb.poses[TRIAL ? ctl.cur ^ 1 : ctl.cur];
    const double* __restrict__ points = pb.points[TRIAL ? ctl.cur ^ 1 : ctl.cur];
    load_cams(&cams, pb.cams);
    const int e = blockIdx.x * 256 + threadIdx.x;
    double rho0 = 0;
    if (e < E && pb.active[e]) {
        double pc[3];
        const DCam& c = cams.c[pb.ecam[e]];
        cam_point(poses + 7 * pb.epose[e], points + 3 * pb.epoint[e], c, pc);
        const double e0 = pb.obs[2 * e] - (pc[0] / pc[2] * c.fx + c.cx);
        const double e1 = pb.obs[2 * e + 1] - (pc[1] / pc[2] * c.fy + c.cy);
        const double w = pb.w[e];
        const double x2 = e0 * (w * e0) + e1 * (w * e1);
        pb.err[2 * e] = e0; pb.err[2 * e + 1] = e1; pb.chi2[e] = x2;
        if (robust && x2 > delta * delta) rho0 = 2 * sqrt(x2) * delta - delta * delta; else rho0 = x2;
    }
    const double t = block_sum_256(rho0, s);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pb.partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(pb.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (prev == (unsigned)pb.nblk - 1);
    }
    __syncthreads();
    if (!last) return false;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double v = 0;
    for (int i = threadIdx.x; i < pb.nblk; i += 256) v += __hip_atomic_load(&pb.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double tot = block_sum_256(v, s);
    if (threadIdx.x == 0) __hip_atomic_store(pb.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!TRIAL) {
        if (threadIdx.x == 0) { ctl.currentChi = tot; ctl.iniChi = tot; ctl.errors_current = 1; }
        return false;
    }
    __syncthreads();
    double w = 0;                                   // computeScale: block partials of k_solve_update (an earlier launch), fixed order
    for (int i = threadIdx.x; i < pb.nb_pts + pb.nb_pose; i += 256) w += pb.scale_part[i];
    const double sc = block_sum_256(w, s);
    if (threadIdx.x != 0) return false;
    return lm_accept(pb, ctl, stop_words, tot, sc);
}

// k_error<1> is the LAST kernel of a step: the last problem to finish its trial publishes the step to the host (publish_step_end).
template <int TRIAL>
__global__ __launch_bounds__(256) void k_error(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls, const volatile int* __restrict__ stop_words, int B,
                                               int* __restrict__ h_progress, unsigned* __restrict__ grid_ticket)
{
    __shared__ double s[256];
    __shared__ DCams cams;
    __shared__ bool last;
    const bool finisher = error_body<TRIAL>(probs[blockIdx.y], ctls[blockIdx.y], stop_words, s, cams, last);
    if (TRIAL && finisher) publish_step_end(B, h_progress, grid_ticket);
}

// FIRST kernel of every step, thread per edge: (1) everything that happens BETWEEN LM iterations, then (2) the linearisation of the
// iteration the step starts -- one launch, because both are edge-parallel over the same edges (until round 3 part 1 was a kernel of
// its own, k_post, at the END of the step: a no-op launch in 10 of 13 steps of a C4 solve).
// (1) state == ROUND_END   outlier flags chi2 (last evaluation) > th || depth <= 0 with the CURRENT estimates (Optimizer.cc:607,
//                          653); after round 0 they become the level-1 set of round 1 (:607-612) unless the stop flag was seen
//                          (:597-600), and the same thread evaluates the edge's error for round 1's first iteration (robust kernel
//                          off); the current estimates are copied to the output arrays. The problem's last block (ticket) then
//                          starts round 1 (lambda is re-initialised by k_reduce_pose) or marks the problem done.
//     state == NEW_ITER with stale errors (first iteration of round 0; an iteration that ended on a rejected trial)
//                          computeActiveErrors at the current estimates, chi2 total -> currentChi.
// (2) linearizeOplus + constructQuadraticForm of the edge if the problem is (or has just been put) at the top of an LM iteration:
//     cpoint[e] = {Hll 00,01,02,11,12,22, bl0..2}, cpose[e] = {21 upper entries of Hpp row-major, bp0..5}, Hpl[e] = 6x3 row-major
//     (pose rows, point cols). A block cannot wait for the problem's last block, so it PREDICTS what that block will write from the
//     control word it read on entry: round 1 starts after round 0 unless stopped (robust kernel off). The one case it cannot see --
//     round 1 without a single active edge, which ends the problem -- leaves a linearisation nobody reads: every later kernel of the
//     step tests ctl.state.
// The end of a problem is decided here: its last block publishes it to the host (publish_problem_end).
// linearizeOplus + constructQuadraticForm of ONE edge at the estimates of buffer `lb` (the pose T, the camera-frame point pc, the edge's
// error / chi2 there): cpoint[e] = {Hll 00,01,02,11,12,22, bl0..2}, cpose[e] = {21 upper entries of Hpp row-major, bp0..5}, Hpl[e] = 6x3
// row-major (pose rows, point cols). Shared by k_begin and by the trial kernel (which linearises the estimates it has just evaluated).
__device__ __forceinline__ void linearize_edge(const BaProb& pb, int lb, int e, bool act, bool free_pose, const DCam& c, const double* T, const double* pc,
                                               double err0, double err1, double x2, int robust, double delta, double* cp /* the edge's 9 landmark terms: cpoint[lb] + 9 e, or LDS */)
{
    if (!act) {                                            // level-1 edge: adds nothing to any block (the CSR lists still name it)
        for (int i = 0; i < 9; ++i) cp[i] = 0;
        if (free_pose) {
            double* cq = pb.cpose[lb] + (size_t)e * 27;
            for (int i = 0; i < 27; ++i) cq[i] = 0;
            double* h = pb.Hpl[lb] + (size_t)e * 18;
            for (int i = 0; i < 18; ++i) h[i] = 0;
        }
        return;
    }
    const double x = pc[0], y = pc[1], z = pc[2];
    const double sz = -1. / z;
    const double st[6] = {sz * c.fx, sz * 0.0, sz * (-x / z * c.fx), sz * 0.0, sz * c.fy, sz * (-y / z * c.fy)};
    const double J3[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    double A[12], Jp[12], Jx[6];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) A[i * 6 + j] = st[i * 3] * J3[j] + st[i * 3 + 1] * J3[6 + j] + st[i * 3 + 2] * J3[12 + j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) {
        double acc = 0;
        for (int k = 0; k < 6; ++k) acc += A[i * 6 + k] * c.adj[k * 6 + j];
        Jp[i * 6 + j] = acc;
    }
    double qt[4], R[9];
    qmul(c.q, T + 3, qt); qnormalize(qt); qtoR(qt, R);         // rotation of T_ext * T_mcs
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Jx[i * 3 + j] = st[i * 3] * R[j] + st[i * 3 + 1] * R[3 + j] + st[i * 3 + 2] * R[6 + j];
    double w = pb.w[e];
    double r0 = -w * err0, r1 = -w * err1;
    if (robust) {
        const double rho1 = x2 <= delta * delta ? 1.0 : delta / sqrt(x2);
        r0 *= rho1; r1 *= rho1; w = rho1 * w;
    }
    int k = 0;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) cp[k++] = Jx[i] * w * Jx[j] + Jx[3 + i] * w * Jx[3 + j];
    for (int i = 0; i < 3; ++i) cp[6 + i] = Jx[i] * r0 + Jx[3 + i] * r1;
    if (free_pose) {
        double* cq = pb.cpose[lb] + (size_t)e * 27;
        k = 0;
        for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) cq[k++] = Jp[i] * w * Jp[j] + Jp[6 + i] * w * Jp[6 + j];
        for (int i = 0; i < 6; ++i) cq[21 + i] = Jp[i] * r0 + Jp[6 + i] * r1;
        double* h = pb.Hpl[lb] + (size_t)e * 18;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) h[i * 3 + j] = Jp[i] * w * Jx[j] + Jp[6 + i] * w * Jx[3 + j];
    }
}

// sums over the 256-thread sub-blocks of a workgroup of 256 VB threads, each with block_sum_256's tree (the same bits): the total of the
// caller's sub-block comes back in all of its threads
template <int VB>
__device__ __forceinline__ double vblock_sum_256(double v, double* s /*[256 VB]*/)
{
    const int t = threadIdx.x;
    s[t] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if ((t & 255) < d) s[t] += s[t + d]; __syncthreads(); }
    const double r = s[t & ~255];
    __syncthreads();
    return r;
}

// The body of k_begin for a workgroup of 256 VB threads = VB of its 256-edge blocks (VB = 1 is the only form left: round 5's k_front, which ran it with
// VB = 2 inside a fused launch, was bit-identical, slower and is gone). bx = the workgroup's index among the problem's ceil(nblk / VB).
template <int VB>
__device__ __forceinline__ void begin_body(const BaProb& pb, BaCtl& ctl, int bx, double* s /*[256 VB]*/, DCams& cams, int& s_cnt, bool& last, int* __restrict__ h_progress,
                                           unsigned* __restrict__ grid_ticket)
{
    const int nwg = (pb.nblk + VB - 1) / VB, t = threadIdx.x;
    // the control word as of the end of the previous step: read ONCE -- the problem's last block rewrites it while other blocks
    // are still in part (2)
    const int state = ctl.state, round = ctl.round, stopped = ctl.stopped, cur = ctl.cur, robust_in = ctl.robust;
    const bool round_end = state == ST_ROUND_END, pre = state == ST_NEW_ITER && !ctl.errors_current;
    const bool next_round = round_end && round == 0 && !stopped && pb.iters[1] > 0;
    const bool lin = state == ST_NEW_ITER || next_round;           // (2) runs
    const int robust = round_end ? 0 : robust_in;
    if (!(round_end || pre || lin)) return;                          // workgroup-uniform
    const double* __restrict__ poses = pb.poses[cur];
    const double* __restrict__ points = pb.points[cur];
    const double delta = pb.delta;
    if (t == 0) s_cnt = 0;
    load_cams(&cams, pb.cams);
    const int vb = bx * VB + (t >> 8);                               // the 256-edge block of this thread
    const int e = bx * (256 * VB) + t;
    const bool in = e < pb.E;
    double rho0 = 0, pc[3] = {0, 0, 1}, err0 = 0, err1 = 0, x2 = 0;
    bool act = false;
    int ps = 0, cam_id = 0;
    if (in) {
        ps = pb.epose[e]; cam_id = pb.ecam[e];
        cam_point(poses + 7 * ps, points + 3 * pb.epoint[e], cams.c[cam_id], pc);
        act = pb.active[e] != 0;
    }
    if (round_end || pre) {
        if (in) {
            const DCam& c = cams.c[cam_id];
            if (round_end) {
                const uint8_t f = (pb.chi2[e] > pb.chi2_th || !(pc[2] > 0.0)) ? 1 : 0;
                pb.flag[e] = f;
                if (round == 0) {
                    pb.level1[e] = stopped ? 0 : f;
                    if (!stopped) { act = !f; pb.active[e] = act; if (act) atomicAdd(&s_cnt, 1); }
                }
            }
            if ((pre || next_round) && act) {
                err0 = pb.obs[2 * e] - (pc[0] / pc[2] * c.fx + c.cx);
                err1 = pb.obs[2 * e + 1] - (pc[1] / pc[2] * c.fy + c.cy);
                const double w = pb.w[e];
                x2 = err0 * (w * err0) + err1 * (w * err1);
                pb.err[2 * e] = err0; pb.err[2 * e + 1] = err1; pb.chi2[e] = x2;
                if (robust && x2 > delta * delta) rho0 = 2 * sqrt(x2) * delta - delta * delta; else rho0 = x2;
            }
        }
        if (round_end) {
            for (int i = bx * (256 * VB) + t; i < 7 * pb.P; i += nwg * (256 * VB)) pb.out_poses[i] = poses[i];
            for (int i = bx * (256 * VB) + t; i < 3 * pb.L; i += nwg * (256 * VB)) pb.out_points[i] = points[i];
        }
    } else if (in && act) {                                    // errors are current (written by the accepted trial's k_error<1>)
        err0 = pb.err[2 * e]; err1 = pb.err[2 * e + 1]; x2 = pb.chi2[e];
    }
    // ---- (2) linearizeOplus + constructQuadraticForm
    if (lin && in) linearize_edge(pb, cur, e, act, pb.pose_idx[ps] >= 0, cams.c[cam_id], poses + 7 * ps, pc, err0, err1, x2, robust, delta, pb.cpoint[cur] + (size_t)e * 9);
    if (!(round_end || pre)) return;                                 // workgroup-uniform
    const double tsum = vblock_sum_256<VB>(rho0, s);                 // (its barriers also order every thread's stores before thread 0's release)
    if ((t & 255) == 0 && vb < pb.nblk) __hip_atomic_store(&pb.partial[vb], tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (VB > 1) __syncthreads();                                     // the sub-blocks' partials are stored before thread 0 releases them
    if (t == 0) {
        if (s_cnt) atomicAdd(&ctl.n_active, s_cnt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(pb.ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1;
    }
    __syncthreads();
    if (!last) return;                                               // workgroup-uniform
    // ---- the problem's last workgroup: the chi2 total in block order, the round change / the end of the problem
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double v = 0;
    if (t < 256) for (int i = t; i < pb.nblk; i += 256) v += __hip_atomic_load(&pb.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double tot = vblock_sum_256<VB>(v, s);                     // sub-block 0 = block_sum_256 over threads 0..255
    if (t == 0) {
        __hip_atomic_store(pb.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int n_act = __hip_atomic_load(&ctl.n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool done = false;
        if (pre || (next_round && n_act > 0)) {
            if (next_round) { ctl.round = 1; ctl.it = 0; ctl.qmax = 0; ctl.nBad = 0; ctl.robust = 0; }   // Optimizer.cc:612-621
            ctl.currentChi = tot; ctl.iniChi = tot; ctl.errors_current = 1; ctl.state = ST_NEW_ITER;
        } else { ctl.state = ST_DONE; done = true; }
        if (done) publish_problem_end(h_progress, grid_ticket);
    }
}

__global__ __launch_bounds__(256) void k_begin(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls, int B, int* __restrict__ h_progress,
                                               unsigned* __restrict__ grid_ticket)
{
    __shared__ double s[256];
    __shared__ DCams cams;
    __shared__ int s_cnt;
    __shared__ bool last;
    const BaProb& pb = probs[blockIdx.y];
    if ((int)blockIdx.x < pb.nblk) begin_body<1>(pb, ctls[blockIdx.y], blockIdx.x, s, cams, s_cnt, last, h_progress, grid_ticket);
}

// block (1024 threads) per free pose: 37 edge chunks x 27 components, combined in chunk order (deterministic). The
// list walk is latency-bound (index load -> value load), so many short chunks with 4 loads in flight each.
constexpr int kPoseChunks = 37;
// computeLambdaInit (optimization_algorithm_levenberg.cpp:170-181) rides along: every block leaves the max |diagonal| of its
// Hessian blocks in maxd_part[], and at the first iteration of a round the last block to finish (ticket) takes the maximum of
// those np + nb_pts numbers -- a max is order-independent, so the result is reproducible -- and resets the LM multipliers.
// Called by the first wave of a block with a wave-uniform `md`.
__device__ __forceinline__ void reduce_finish(const BaProb& pb, BaCtl& ctl, double md, int slot, int n_blk)
{
    if (ctl.it != 0) return;                                 // block-uniform: lambda is only initialised at the first iteration
    const int lane = threadIdx.x & 63;
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&pb.maxd_part[slot], md, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(pb.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)n_blk - 1;
    }
    last = __shfl(last, 0);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double m = 0.0;
    for (int i = lane; i < n_blk; i += 64) m = fmax(m, __hip_atomic_load(&pb.maxd_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d));
    if (lane == 0) {
        ctl.maxdiag = m; ctl.mult = 1.0; ctl.ni = 2; ctl.nBad = 0;       // lambda = tau * max diagonal, tau = 1e-5
        __hip_atomic_store(pb.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(1024) void k_reduce_pose(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls)
{
    __shared__ double part[kPoseChunks][27];
    __shared__ double s[27];
    const BaProb& pb = probs[blockIdx.y];
    if (ctls[blockIdx.y].state != ST_NEW_ITER) return;
    const int np = pb.np, cur = ctls[blockIdx.y].cur;
    if ((int)blockIdx.x >= np + pb.nb_pts) return;
    if ((int)blockIdx.x >= np) {                             // blocks past the free poses: 64 landmarks each (thread per landmark)
        if (threadIdx.x >= 64) return;
        const int l = (blockIdx.x - np) * 64 + threadIdx.x;
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int n_act = 0;
        for (int k = l < pb.L ? pb.pt_off[l] : 0, k1 = l < pb.L ? pb.pt_off[l + 1] : 0; k < k1; ++k) {
            const int e = pb.pt_edges[k];
            n_act += pb.active[e];
            const double* c = pb.cpoint[cur] + (size_t)e * 9;
            for (int i = 0; i < 9; ++i) a[i] += c[i];
        }
        // NOTE: threads of this branch must all reach the block's finisher below (no early return for l >= L)
        double md = 0.0;
        if (l < pb.L) {
        pb.pt_active[cur][l] = n_act > 0;            // a landmark without active edges is not part of this round
        double* H = pb.Hll[cur] + (size_t)l * 9;
        H[0] = a[0]; H[1] = a[1]; H[2] = a[2]; H[3] = a[1]; H[4] = a[3]; H[5] = a[4]; H[6] = a[2]; H[7] = a[4]; H[8] = a[5];
        pb.bl[cur][3 * l] = a[6]; pb.bl[cur][3 * l + 1] = a[7]; pb.bl[cur][3 * l + 2] = a[8];
        if (n_act > 0) md = fmax(fmax(fabs(a[0]), fabs(a[3])), fabs(a[5]));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) md = fmax(md, __shfl_xor(md, d));
        reduce_finish(pb, ctls[blockIdx.y], md, blockIdx.x, pb.np + pb.nb_pts);
        return;
    }
    const int32_t* __restrict__ ps_off = pb.ps_off;
    const int32_t* __restrict__ ps_edges = pb.ps_edges;
    const double* __restrict__ cpose = pb.cpose[cur];
    const int i = blockIdx.x, t = threadIdx.x;
    const int c = t % 27, q = t / 27;
    if (q < kPoseChunks) {
        double a = 0;
        const int k1 = ps_off[i + 1];
#pragma unroll 4
        for (int k = ps_off[i] + q; k < k1; k += kPoseChunks) a += cpose[(size_t)ps_edges[k] * 27 + c];
        part[q][c] = a;
    }
    __syncthreads();
    if (t < 27) { double a = 0; for (int q2 = 0; q2 < kPoseChunks; ++q2) a += part[q2][t]; s[t] = a; }
    __syncthreads();
    if (t < 36) {
        const int r = t / 6, qq = t % 6, lo = min(r, qq), hi = max(r, qq);
        pb.Hpp[(size_t)i * 36 + t] = s[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
    }
    if (t < 6) pb.bp[i * 6 + t] = s[21 + t];
    if (t < 64) {                                            // wave 0: max |diagonal| of this pose block (entries 0, 6, 11, 15, 18, 20 of the packed upper triangle)
        const double md = fmax(fmax(fmax(fabs(s[0]), fabs(s[6])), fmax(fabs(s[11]), fabs(s[15]))), fmax(fabs(s[18]), fabs(s[20])));
        reduce_finish(pb, ctls[blockIdx.y], md, blockIdx.x, pb.np + pb.nb_pts);
    }
}

// setLambda for the landmark blocks, one launch:
//   blocks [0, nblk_e)  thread per edge of a free pose: BD[e] = Hpl[e] Dinv[point(e)] (zero for level-1 edges); the 3x3
//                       inverse is recomputed per edge (40 flops) so that the edges do not wait for a per-point pass
//   blocks [nblk_e, ..) thread per point: Dinv = (Hll + lambda I)^-1, db = Dinv bl (read by k_schur / k_solve_update)
__global__ __launch_bounds__(256) void k_prep(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls)
{
    const BaProb& pb = probs[blockIdx.y];
    BaCtl& ctl = ctls[blockIdx.y];
    if (ctl.state > ST_RETRY) return;
    const int nblk_e = pb.np ? pb.nblk : 0, L = pb.L;
    if ((int)blockIdx.x >= nblk_e + (L + 255) / 256) return;
    const double lambda = 1e-5 * ctl.maxdiag * ctl.mult;     // computeLambdaInit (tau = 1e-5) x the LM multiplier
    const double* __restrict__ Hll = pb.Hll[ctl.cur];
    if ((int)blockIdx.x < nblk_e) {
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e >= pb.E || pb.pose_idx[pb.epose[e]] < 0) return;
        double* o = pb.BD + (size_t)e * 18;
        if (!pb.active[e]) {                                 // level-1 edge: still named by the pair lists, contributes zero
            for (int i = 0; i < 18; ++i) o[i] = 0.0;
            return;
        }
        const int l = pb.epoint[e];
        double H[9], D[9];
        for (int i = 0; i < 9; ++i) H[i] = Hll[(size_t)l * 9 + i];
        H[0] += lambda; H[4] += lambda; H[8] += lambda;
        inv3(H, D);
        const double* B = pb.Hpl[ctl.cur] + (size_t)e * 18;
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) o[r * 3 + c] = B[r * 3] * D[c] + B[r * 3 + 1] * D[3 + c] + B[r * 3 + 2] * D[6 + c];
        return;
    }
    const int l = (blockIdx.x - nblk_e) * 256 + threadIdx.x;
    if (l == 0) ctl.ok = 1.0;                                // reset the "factorisation succeeded" flag of this trial
    if (l >= L || !pb.pt_active[ctl.cur][l]) return;
    double H[9], D[9];
    for (int i = 0; i < 9; ++i) H[i] = Hll[(size_t)l * 9 + i];
    H[0] += lambda; H[4] += lambda; H[8] += lambda;
    inv3(H, D);
    const double* bl = pb.bl[ctl.cur];
    for (int i = 0; i < 9; ++i) pb.Dinv[(size_t)l * 9 + i] = D[i];
    for (int i = 0; i < 3; ++i) pb.db[3 * l + i] = D[i * 3] * bl[3 * l] + D[i * 3 + 1] * bl[3 * l + 1] + D[i * 3 + 2] * bl[3 * l + 2];
}

// Storage of the reduced camera system for k_ldlt_mfma (use_reg == 1): the lower triangle as 16x16 TILES in the order and the register
// layout the factorisation keeps them in -- tile t = I (I + 1) / 2 + K (block row I >= block column K), 256 doubles, element (row, col)
// of the tile at (row >> 2) * 64 + (row & 3) * 16 + col = register row >> 2 of lane (row & 3) * 16 + col in the MFMA accumulator
// layout -- so that a worker wave loads a tile with four fully coalesced 512-byte reads and no transposition through LDS. Diagonal
// tiles hold both triangles. (The other two solvers read S row-major, ld x ld.)
__host__ __device__ inline size_t s_tile_off(int x, int y)       // element (x, y) of the matrix, block(x) >= block(y)
{
    const int I = x >> 4, K = y >> 4, row = x & 15, col = y & 15;
    return (size_t)(I * (I + 1) / 2 + K) * 256 + (row >> 2) * 64 + (row & 3) * 16 + col;
}

// workgroup per pose pair (i1 <= i2): S block = [i1==i2](Hpp + lambda I) - sum over shared points BD[e1] Hpl[e2]^T.
// C list chunks x 36 block entries per workgroup (4 list entries in flight per thread); partials combined in fixed order. A C4 problem
// has ~860 such workgroups per trial and a batch of 8 problems ~6 900: 16-wave workgroups (C = 28) for every group size needed 13
// rounds of the chip's wave slots for 8 problems (97 us), 4-wave ones (C = 7) 4 rounds (65 us); a single problem is the other way round.
constexpr int kSchurFine = 28;                           // partial sums per S entry: list entry k belongs to partial k % 28, whatever the workgroup size
constexpr int kSchurThreads = 256;
constexpr int kSchurRhsChunks = 42;                      // 42 x 6 rows = 252 threads for the reduced right-hand side
// C = list chunks per workgroup: 28 (16 waves, one partial per thread) is the form that runs -- for one or two problems, where the pose pairs are too
// few to fill the chip and the shortest chain per thread wins; k_schur_w below takes the larger groups. (C = 7 / 14, 4 / 8 waves with several partials
// per thread, were the round-4 forms for batches.) Every form adds the 28 partials in index order: the same bits whichever one runs, so a batch still
// equals its problems solved one by one.
// one partial of a pose's reduced right-hand side row: sum over the list entries k0, k0 + kSchurRhsChunks, ... of H_pl[e] row . db[point(e)], in list order.
// Three dependent loads per entry (list -> edge -> point -> db): eight entries go through each level together (one at a time the ~11 entries of a
// C4 pose made this row the longest chain of the whole Schur launch).
__device__ __forceinline__ double schur_rhs_partial(int k0, int k1, const int32_t* __restrict__ ps_edges, const int32_t* __restrict__ e_point,
                                                    const double* __restrict__ Hrow /* Hpl + 3 r */, const double* __restrict__ db)
{
    constexpr int kU = 8;
    double a = 0;
    for (int kb = k0; kb < k1; kb += kU * kSchurRhsChunks) {
        int e[kU], pt[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int k = kb + u * kSchurRhsChunks; e[u] = ps_edges[k < k1 ? k : k0]; }
#pragma unroll
        for (int u = 0; u < kU; ++u) pt[u] = e_point[e[u]];
        double B[kU][3], d[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const double* Bp = Hrow + (size_t)e[u] * 18;
            const double* dp = db + 3 * pt[u];
#pragma unroll
            for (int j = 0; j < 3; ++j) { B[u][j] = Bp[j]; d[u][j] = dp[j]; }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) if (kb + u * kSchurRhsChunks < k1) a += B[u][0] * d[u][0] + B[u][1] * d[u][1] + B[u][2] * d[u][2];
    }
    return a;
}

template <int C>
__global__ __launch_bounds__(C == 7 ? 256 : C == 14 ? 512 : 1024) void k_schur(const BaProb* __restrict__ probs, const BaCtl* __restrict__ ctls)
{
    static_assert(C == 7 || C == 14 || C == 28, "28 fine partials split evenly");
    __shared__ double part[kSchurFine][36];
    const BaProb& pb = probs[blockIdx.y];
    const BaCtl& ctl = ctls[blockIdx.y];
    if (ctl.state > ST_RETRY || pb.np == 0) return;
    const int n_pairs = pb.n_pairs;
    if ((int)blockIdx.x >= n_pairs + pb.np) return;
    const double* __restrict__ Hpl = pb.Hpl[ctl.cur];
    if ((int)blockIdx.x >= n_pairs) {                        // blocks past the pair list: reduced right-hand side of one free pose (dispatching them FIRST was measured: 20.4 vs 19.6 us)
        // bsch = bp - sum_e Hpl[e] db[point(e)]; kSchurRhsChunks edge chunks x 6 rows, combined in chunk order
        static_assert(kSchurRhsChunks * 6 <= kSchurFine * 36 && kSchurRhsChunks * 6 <= kSchurThreads, "rhs partials live in `part`");
        double (*bpart)[6] = reinterpret_cast<double (*)[6]>(&part[0][0]);
        const int32_t* __restrict__ ps_off = pb.ps_off;
        const int32_t* __restrict__ ps_edges = pb.ps_edges;
        const int32_t* __restrict__ e_point = pb.epoint;
        const double* __restrict__ db = pb.db;
        const int i = blockIdx.x - n_pairs, t = threadIdx.x;
        const int r = t % 6, q = t / 6;
        if (q < kSchurRhsChunks) bpart[q][r] = schur_rhs_partial(ps_off[i] + q, ps_off[i + 1], ps_edges, e_point, Hpl + r * 3, db);
        __syncthreads();
        if (t < 6) { double a = 0; for (int q2 = 0; q2 < kSchurRhsChunks; ++q2) a += bpart[q2][t]; pb.bsch[i * 6 + t] = pb.bp[i * 6 + t] - a; }
        return;
    }
    const double lambda = 1e-5 * ctl.maxdiag * ctl.mult;
    const int32_t* __restrict__ pair_off = pb.pair_off;
    const int2* __restrict__ pair_e = reinterpret_cast<const int2*>(pb.pair_e);
    const double* __restrict__ BD = pb.BD;
    const int p = blockIdx.x, t = threadIdx.x;
    const int el = t % 36, q = t / 36;
    const int r = el / 6, c = el % 6;
    const int k0 = pair_off[p], k1 = pair_off[p + 1];
    if (k0 == k1 && p >= pb.np) return;                      // two poses without a common point: their block of S stays zero (the arena's zeroed region)
    auto term = [&](int k) {
        const int2 pe = pair_e[k];
        const double* a = BD + (size_t)pe.x * 18 + r * 3;
        const double* b = Hpl + (size_t)pe.y * 18 + c * 3;
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    };
    if (q < C) {
        constexpr int A = kSchurFine / C;                    // accumulators per thread
        double acc[A];
#pragma unroll
        for (int m = 0; m < A; ++m) acc[m] = 0;
        int k = k0 + q;
        for (; k + (A - 1) * C < k1; k += kSchurFine) {
#pragma unroll
            for (int m = 0; m < A; ++m) acc[m] += term(k + C * m);
        }
#pragma unroll
        for (int m = 0; m < A - 1; ++m) if (k + C * m < k1) acc[m] += term(k + C * m);
#pragma unroll
        for (int m = 0; m < A; ++m) part[q + C * m][el] = acc[m];
    }
    __syncthreads();
    if (t >= 36) return;
    const int i1 = pb.pair_ij[2 * p], i2 = pb.pair_ij[2 * p + 1], ld = pb.ld;
    double acc = 0;
    for (int q2 = 0; q2 < kSchurFine; ++q2) acc += part[q2][t];
    double v = -acc;
    if (i1 == i2) v += pb.Hpp[(size_t)i1 * 36 + t] + (r == c ? lambda : 0.0);
    const int xa = i1 * 6 + r, ya = i2 * 6 + c;
    if (pb.use_reg == 1) {
        // Tiles of the lower triangle (diagonal tiles: both triangles), filled from the UPPER entries only -- like g2o, which hands
        // Eigen the upper triangle (linear_solver_eigen.h:94-124, selfadjointView<Upper>): entry (r, c) and entry (c, r) of a diagonal
        // pose block are the same number on paper but two differently rounded sums here.
        if (xa <= ya) {
            pb.S[s_tile_off(ya, xa)] = v;
            if (xa != ya && (xa >> 4) == (ya >> 4)) pb.S[s_tile_off(xa, ya)] = v;
        }
        return;
    }
    pb.S[(size_t)xa * ld + ya] = v;
    if (i1 != i2) pb.S[(size_t)ya * ld + xa] = v;
}

// ---- k_schur_w (round 5): TWO WAVES per pose pair, a lane per (partial, 3 x 3 quarter of the 6 x 6 block). k_schur gives every list entry to 36 threads that
// each load 7 values for 3 multiply-adds (an int2, a row of BD, a row of H_pl): 16 waves per pair, and a batch of 4 problems is 3 440 pairs = 27 500 waves, 3.8
// rounds of the chip at 11 us. Here lane (q, hr, hc) walks the entries of partial q (k = q, q + 28, ...: k_schur's own partials, so the sums keep their order and
// their BITS) with rows 3 hr .., columns 3 hc .. of the block in 9 accumulators: an entry costs it 18 loads (three rows of BD[e1], three of H_pl[e2]) for 27
// multiply-adds. The walk is a chain of dependent loads (list entry -> blocks), and the diagonal pairs hold ~16 entries per partial, so FOUR entries' blocks
// are requested together and the next four index pairs ahead of them. The 28 partials are added in index order through LDS (rows of 37 doubles), the block goes
// to S exactly as k_schur writes it. Workgroups of 4 waves = 2 pairs; the workgroups behind the pairs are k_schur's right-hand-side rows.
constexpr int kSchurWPairs = 2;
__global__ __launch_bounds__(128 * kSchurWPairs) void k_schur_w(const BaProb* __restrict__ probs, const BaCtl* __restrict__ ctls)
{
    __shared__ double s_part[kSchurWPairs][kSchurFine][37];
    const BaProb& pb = probs[blockIdx.y];
    const BaCtl& ctl = ctls[blockIdx.y];
    if (ctl.state > ST_RETRY || pb.np == 0) return;
    const int n_pairs = pb.n_pairs, n_pb = (n_pairs + kSchurWPairs - 1) / kSchurWPairs;
    if ((int)blockIdx.x >= n_pb + pb.np) return;
    const double* __restrict__ Hpl = pb.Hpl[ctl.cur];
    if ((int)blockIdx.x >= n_pb) {                           // the reduced right-hand side of one free pose: k_schur's code (kSchurRhsChunks x 6 threads)
        double (*bpart)[6] = reinterpret_cast<double (*)[6]>(&s_part[0][0][0]);
        static_assert(kSchurRhsChunks * 6 <= 128 * kSchurWPairs && kSchurRhsChunks * 6 <= kSchurFine * 37, "rhs partials live in s_part");
        const int32_t* __restrict__ ps_off = pb.ps_off;
        const int32_t* __restrict__ ps_edges = pb.ps_edges;
        const int32_t* __restrict__ e_point = pb.epoint;
        const double* __restrict__ db = pb.db;
        const int i = blockIdx.x - n_pb, t = threadIdx.x;
        const int r = t % 6, q = t / 6;
        if (q < kSchurRhsChunks) bpart[q][r] = schur_rhs_partial(ps_off[i] + q, ps_off[i + 1], ps_edges, e_point, Hpl + r * 3, db);
        __syncthreads();
        if (t < 6) { double a = 0; for (int q2 = 0; q2 < kSchurRhsChunks; ++q2) a += bpart[q2][t]; pb.bsch[i * 6 + t] = pb.bp[i * 6 + t] - a; }
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pl = wave >> 1;
    const int p = min(blockIdx.x * kSchurWPairs + pl, n_pairs - 1);                  // (an odd pair count: the last workgroup's second half repeats its pair and stores nothing)
    const bool real = (int)(blockIdx.x * kSchurWPairs + pl) < n_pairs;
    const int k0 = pb.pair_off[p], k1 = pb.pair_off[p + 1];
    const bool skip = !real || (k0 == k1 && p >= pb.np);     // two poses without a common point: their block of S stays zero (the arena's zeroed region)
    const int2* __restrict__ pair_e = reinterpret_cast<const int2*>(pb.pair_e);
    const double* __restrict__ BD = pb.BD;
    const int ql = lane >> 2, hr = (lane >> 1) & 1, hc = lane & 1, q = 14 * (wave & 1) + ql;
    double (*part)[37] = s_part[pl];
    if (!skip && ql < 14) {
        double acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = 0;
        constexpr int kD = 4;                                // entries in flight per lane
        auto idx = [&](int k) { return pair_e[k < k1 ? k : k0]; };
        int2 pn[kD];
        int k = k0 + q;
#pragma unroll
        for (int d = 0; d < kD; ++d) pn[d] = idx(k + kSchurFine * d);
        while (k < k1) {
            int2 pc[kD];
#pragma unroll
            for (int d = 0; d < kD; ++d) { pc[d] = pn[d]; pn[d] = idx(k + kSchurFine * (kD + d)); }
            double av[kD][9], bv[kD][9];
#pragma unroll
            for (int d = 0; d < kD; ++d) {
                const double* a = BD + (size_t)pc[d].x * 18 + 9 * hr;          // rows 3 hr .. 3 hr + 2 of BD[e1]
                const double* b = Hpl + (size_t)pc[d].y * 18 + 9 * hc;         // rows 3 hc .. 3 hc + 2 of H_pl[e2]
#pragma unroll
                for (int i = 0; i < 9; ++i) { av[d][i] = a[i]; bv[d][i] = b[i]; }
            }
#pragma unroll
            for (int d = 0; d < kD; ++d) {
                if (k + kSchurFine * d < k1) {
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc)
                            acc[rr * 3 + cc] += av[d][rr * 3] * bv[d][cc * 3] + av[d][rr * 3 + 1] * bv[d][cc * 3 + 1] + av[d][rr * 3 + 2] * bv[d][cc * 3 + 2];
                }
            }
            k += kSchurFine * kD;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) part[q][(3 * hr + rr) * 6 + 3 * hc + cc] = acc[rr * 3 + cc];
    }
    __syncthreads();
    if (skip || (wave & 1) || lane >= 36) return;
    const int t = lane, r = t / 6, c = t % 6;
    const int i1 = pb.pair_ij[2 * p], i2 = pb.pair_ij[2 * p + 1], ld = pb.ld;
    double acc = 0;
    for (int q2 = 0; q2 < kSchurFine; ++q2) acc += part[q2][t];
    double v = -acc;
    const double lambda = 1e-5 * ctl.maxdiag * ctl.mult;
    if (i1 == i2) v += pb.Hpp[(size_t)i1 * 36 + t] + (r == c ? lambda : 0.0);
    const int xa = i1 * 6 + r, ya = i2 * 6 + c;
    if (pb.use_reg == 1) {                                   // (tiles of the lower triangle, filled from the upper entries: see k_schur)
        if (xa <= ya) {
            pb.S[s_tile_off(ya, xa)] = v;
            if (xa != ya && (xa >> 4) == (ya >> 4)) pb.S[s_tile_off(xa, ya)] = v;
        }
        return;
    }
    pb.S[(size_t)xa * ld + ya] = v;
    if (i1 != i2) pb.S[(size_t)ya * ld + xa] = v;
}

// ---- pose-pair lists on the device, once per call (until round 4 the host built and uploaded them: 0.8 of the 1.25 MB a C4 problem
// uploads, and two passes over ~100 k (point, pose, pose) triples per problem on the host).
// List of the pair (i1 <= i2) = the points both poses observe, in POINT order, as (edge of i1, edge of i2): the AND of two bit rows.
//   k_pairs_mark   thread per edge of a free pose: bit (pose, point), edge_of[pose][point]   (at most one edge per (pose, point):
//                  build_round's duplicate test)
//   k_pairs_count  thread per pair: popcount of the pair's AND
//   k_pairs_scan   one workgroup per problem: exclusive scan of the counts in pair order -> pair_off
//   k_pairs_fill   one wave per pair: the set bits of the AND in ascending order -> pair_e
__global__ __launch_bounds__(256) void k_pairs_mark(const BaProb* __restrict__ probs)
{
    const BaProb& pb = probs[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pb.E) return;
    const int pi = pb.pose_idx[pb.epose[e]];
    if (pi < 0) return;
    const int l = pb.epoint[e];
    atomicOr(pb.pt_bits + (size_t)pi * pb.pt_words + (l >> 5), 1u << (l & 31));
    pb.edge_of[(size_t)pi * pb.L + l] = e;
}

// entries of every pair's list = popcount of the AND of the two poses' point rows: thread per pair over the whole chip (round 6: k_pairs_scan counted them
// itself, one workgroup reading 100 MB of rows at the global-BA shape -- 1.28 ms of a 7.6-ms solve), left in pair_off for the scan
__global__ __launch_bounds__(256) void k_pairs_count(const BaProb* __restrict__ probs)
{
    const BaProb& pb = probs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x, W = pb.pt_words;
    if (p >= pb.n_pairs) return;
    const uint4* a = reinterpret_cast<const uint4*>(pb.pt_bits + (size_t)pb.pair_ij[2 * p] * W);        // pt_words is a multiple of 4
    const uint4* b = reinterpret_cast<const uint4*>(pb.pt_bits + (size_t)pb.pair_ij[2 * p + 1] * W);
    int c = 0;
#pragma unroll 4
    for (int w = 0; w < W / 4; ++w) {
        const uint4 x = a[w], y = b[w];
        c += __popc(x.x & y.x) + __popc(x.y & y.y) + __popc(x.z & y.z) + __popc(x.w & y.w);
    }
    pb.pair_off[p] = c;
}

__global__ __launch_bounds__(1024) void k_pairs_scan(const BaProb* __restrict__ probs)
{
    __shared__ int s_w[16];
    const BaProb& pb = probs[blockIdx.x];
    const int n_pairs = pb.n_pairs, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int running = 0;
    for (int p0 = 0; p0 < n_pairs; p0 += 1024) {
        const int p = p0 + t;
        const int c = p < n_pairs ? pb.pair_off[p] : 0;          // k_pairs_count's result, replaced by the exclusive sum
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int v = s_w[w]; if (w < wave) off += v; tot += v; }
        if (p < n_pairs) pb.pair_off[p] = running + off + inc - c;
        running += tot;
        __syncthreads();
    }
    if (t == 0) pb.pair_off[n_pairs] = running;
}

__global__ __launch_bounds__(64) void k_pairs_fill(const BaProb* __restrict__ probs)
{
    const BaProb& pb = probs[blockIdx.y];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= pb.n_pairs) return;
    const int W = pb.pt_words, L = pb.L;
    const int i1 = pb.pair_ij[2 * p], i2 = pb.pair_ij[2 * p + 1];
    const uint32_t* a = pb.pt_bits + (size_t)i1 * W;
    const uint32_t* b = pb.pt_bits + (size_t)i2 * W;
    const int32_t* e1 = pb.edge_of + (size_t)i1 * L;
    const int32_t* e2 = pb.edge_of + (size_t)i2 * L;
    int2* out = reinterpret_cast<int2*>(pb.pair_e);
    int base = pb.pair_off[p];
    for (int w0 = 0; w0 < W; w0 += 64) {
        const int w = w0 + lane;
        uint32_t m = w < W ? (a[w] & b[w]) : 0u;
        const int c = __popc(m);
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        int pos = base + inc - c;
        while (m) {
            const int l = w * 32 + __builtin_ctz(m);
            m &= m - 1;
            out[pos++] = make_int2(e1[l], e2[l]);
        }
        base += __shfl(inc, 63);
    }
}

// 1/d to full double precision without the IEEE division sequence: v_rcp_f64 + one third-order correction
__device__ __forceinline__ double fast_recip(double d)
{
    const double r = __builtin_amdgcn_rcp(d), e = fma(-d, r, 1.0);       // v_rcp_f64: 2^-24.4; r (1 + e + e^2): error e^3 (scratch/probe/rcp_probe.hip)
    return fma(r, fma(e, e, e), r);
}

// (use_reg: 1 = k_ldlt_mfma, 0 = blocked multi-launch fallback; one workgroup per problem. The column-by-column VALU predecessor of k_ldlt_mfma,
// k_ldlt_reg, was deleted in round 5: nothing selected it but an A/B switch.)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- single-workgroup blocked LDL^T + solve on the f64 matrix cores (n <= 256) ---------------------------------------
// 12 waves: 11 WORKERS hold the lower triangle in their VGPRs as 16x16 tiles in the MFMA accumulator layout (lane l, register r
// holds element (row = (l >> 4) + 4 r, col = l & 15)); tile t = I (I + 1) / 2 + K of the row-major triangle belongs to worker
// t % 11, slot t / 11 (11 slots for <= 15 block rows, 13 for 16: two builds; 7 workers with 18 / 20 slots: 67 instead of 64 us, 15: 64). The twelfth wave is the CHAIN wave: it owns no tile
// and runs the only sequential part, the diagonal blocks.
// Block step J, two workgroup barriers:
//   workers, rows    the panel below the diagonal block is W = A U^-T, L = W D^-1 with the INVERSE of the unit factor of the
//                    diagonal block: 4 v_mfma_f64_16x16x4_f64 per 16-row tile (operand maps: A[i = l & 15][k = l >> 4],
//                    B[k = l >> 4][j = l & 15]), tiles dealt round-robin; L and -W go to LDS in the operand layout of the update
//   workers, update  every worker updates its tiles C(I, K) -= L(I, J) W(K, J)^T with 4 MFMAs each (the slots that have work in
//                    step J come from a per-column bit mask: one scalar test per idle slot); the tiles of block column J + 1 are
//                    published as the next panel, tiles (J + 2, J + 1) and (J + 2, J + 2) also to the chain wave's inbox; the
//                    finished L tiles of column J return to registers
//   chain wave       diagonal block J + 1 needs only panel tile (J + 1, J) and diagonal tile (J + 1, J + 1) as of step J - 1 (its
//                    inbox): it computes that panel tile itself, applies it, forward-substitutes the block's right-hand side
//                    and factors the block -- one ROW per lane, pivot row / column broadcast with DPP row_newbcast
//                    (v_mov_b64_dpp: no SGPR traffic); the same elimination applied to an identity yields U^-1 -- all of it
//                    while the workers run step J. The workers only ever wait for U^-1 of the next block.
// Back substitution L^T x = D^-1 y walks the block columns backwards: tile owners reduce L(I, J)^T x_I with two
// cross-lane adds, the chain wave sums the workers' partial vectors in fixed order and multiplies by U^-T (16 DPP broadcasts).
// Measured (n = 234, scratch/ldlt, 7 workers): 67 us = staging 9 + 14 steps 52 (7 700 cycles each for the first steps, matrix-core bound at
// ~70 % of one CU's rate; 5 200 for the last ones = the chain wave's 1 300 + 3 500) + back substitution 9. f64 MFMA 16x16x4 issues
// every 64 cycles, dependent or not, the waves of a SIMD share the pipe (scratch/probe/mfma_f64_probe.hip): with 3 instead of 2 waves
// per SIMD the operand loads of one wave hide behind the others' MFMAs (64 us).
constexpr int kLS = 17;                        // padded LDS row stride of the 16-wide panels (doubles)
#ifndef DCS_LDLT_WORKERS                       // tuning hook (scratch/ldlt)
#define DCS_LDLT_WORKERS 11
#endif
constexpr int kLdltWorkers = DCS_LDLT_WORKERS; // waves that own tiles; one more wave runs the chain of diagonal blocks
constexpr int kLdltSlotsSmall = (120 + kLdltWorkers - 1) / kLdltWorkers;       // <= 15 block rows (n <= 240): 120 tiles
constexpr int kLdltSlotsBig = (136 + kLdltWorkers - 1) / kLdltWorkers;         // 16 block rows: 136 tiles
constexpr int kLdltThreads = 64 * (kLdltWorkers + 1);
constexpr int kDiagWave = kLdltWorkers;

__device__ __forceinline__ double dbl_of(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double readlane_f64(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane), hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// DPP row_newbcast: lane K of every 16-lane row feeds all lanes of that row. v_mov_b64_dpp broadcasts a double,
// v_fmac_f64_dpp folds the broadcast into the FMA (acc += src[lane K of the row] * mul): no SGPR traffic, no temporaries.
// Everything on the pivot chain is volatile asm so that the hand-made order below (and with it the register footprint)
// survives the scheduler. The compiler does not track hazards inside asm: a VGPR written by one of the two preceding VALU
// instructions must not be the DPP source (2 wait states), which the NOP variants / the op order take care of.
template <int K, bool NOP> __device__ __forceinline__ double bcast16(double v)
{
    double r;
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K, bool NOP> __device__ __forceinline__ void fmac_bcast16(double& acc, double src, double mul)      // acc += src[lane K of the row] * mul
{
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
__device__ __forceinline__ double asm_rcp(double x) { double r; asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ double asm_fnma1(double d, double r) { double e; asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(d), "v"(r)); return e; }   // 1 - d r
__device__ __forceinline__ double asm_fma(double a, double b, double c) { double r; asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ double asm_nmul(double a, double b) { double r; asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }        // -(a b)

// Factor the 16x16 diagonal block whose raw rows sit in P (row stride kLS): A = U D U^T, U unit lower. One wave, lane
// (l & 15) = row (the four 16-lane rows of the wave work redundantly). Writes U^-1 (row-major, stride kLS) and 1/d, and
// forward-substitutes the block's right-hand side yv (lane = row); returns it. ok is cleared on a zero / non-finite pivot.
//
// Pivot k = a dependent chain (broadcast d_k -> v_rcp_f64 -> two Newton steps -> -l_ik) followed by 16 independent
// updates: a_ij -= l_ik (d_k l_jk) for j > k, y_i -= l_ik y_k, and row_i(U^-1) -= l_ik row_k(U^-1) (the elimination applied
// to an identity). Only the update of a_{k+1,k+1} feeds the next pivot, so the other 15 are issued BETWEEN the chain
// instructions of pivot k + 1, two per latency bubble (software pipeline, spelled out because a single wave issues in
// order). Compile-time recursion instead of unrolled loops: every register index and DPP lane is a constant, and the code
// is branch-free (with a branch per pivot the compiler sinks the updates into later blocks and keeps every broadcast
// alive: 200+ VGPRs).
struct Diag16 { double (&ar)[16]; double (&xr)[16]; double yv, nlik, nlikm, myinvd; int lo; };      // ar / xr: caller-provided registers

// the IDX-th of the 16 updates of pivot M: columns M+1 .. 15, then y, then U^-1 columns 0 .. M-1
template <int M, int IDX> __device__ __forceinline__ void diag16_bulk(Diag16& d)
{
    if constexpr (IDX < 15 - M) fmac_bcast16<M + 1 + IDX, false>(d.ar[M + 1 + IDX], d.ar[M], d.nlik);
    else if constexpr (IDX == 15 - M) fmac_bcast16<M, false>(d.yv, d.yv, d.nlikm);
    else fmac_bcast16<M, (IDX - (16 - M) == M - 1)>(d.xr[IDX - (16 - M)], d.xr[IDX - (16 - M)], d.nlikm);   // xr[M-1] was set by plain VALU code: keep its distance
}
template <int M, int I0> __device__ __forceinline__ void diag16_bulk2(Diag16& d)
{
    if constexpr (M >= 0) { diag16_bulk<M, I0>(d); diag16_bulk<M, I0 + 1>(d); }
}
template <int K> __device__ __forceinline__ void diag16_pivot(Diag16& d)
{
    if constexpr (K > 0) { diag16_bulk<K - 1, 0>(d); diag16_bulk<K - 1, 1>(d); diag16_bulk<K - 1, 2>(d); }   // a_kk first; two more keep the DPP read of a_kk two slots away
    const double dk = bcast16<K, K == 0>(d.ar[K]);
    diag16_bulk2<K - 1, 3>(d);
    const double r = asm_rcp(dk);
    diag16_bulk2<K - 1, 5>(d);
    // ONE correction of third order instead of two Newton steps: v_rcp_f64 is good to 2^-24.4 (scratch/probe/rcp_probe.hip), e = 1 - d r, and
    // r (1 + e + e^2) leaves e^3 ~ 1e-22 -- a dependent operation less on the pivot chain, which is what a diagonal block's time is made of
    const double e = asm_fnma1(dk, r);
    diag16_bulk2<K - 1, 7>(d);
    diag16_bulk2<K - 1, 9>(d);
    const double t = asm_fma(e, e, e);
    diag16_bulk2<K - 1, 11>(d);
    const double invd = asm_fma(r, t, r);
    diag16_bulk2<K - 1, 13>(d);
    if constexpr (K > 0) diag16_bulk<K - 1, 15>(d);
    d.myinvd = d.lo == K ? invd : d.myinvd;
    d.nlik = asm_nmul(d.ar[K], invd);                                      // -l_ik, meaningful for rows below k
    d.nlikm = d.lo > K ? d.nlik : 0.0;
    d.xr[K] += d.nlikm;                                                    // column k of U^-1 starts as e_k (set before the loop)
}
template <int... Ks> __device__ __forceinline__ void diag16_pivots(Diag16& d, std::integer_sequence<int, Ks...>) { (diag16_pivot<Ks>(d), ...); }
template <int... Is> __device__ __forceinline__ void diag16_tail(Diag16& d, std::integer_sequence<int, Is...>) { (diag16_bulk<15, Is>(d), ...); }

__device__ __forceinline__ double ldlt_diag16(double (&ar)[16], double (&xr)[16], const double* P, double* Uinv, double* invd_out, double yv, double* ok, int lane)
{
    Diag16 d{ar, xr, yv, 0.0, 0.0, 0.0, lane & 15};
#pragma unroll
    for (int k = 0; k < 16; ++k) { d.ar[k] = P[d.lo * kLS + k]; d.xr[k] = d.lo == k ? 1.0 : 0.0; }
    diag16_pivots(d, std::make_integer_sequence<int, 16>{});
    diag16_tail(d, std::make_integer_sequence<int, 16>{});
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Uinv[d.lo * kLS + k] = d.xr[k];
        invd_out[d.lo] = d.myinvd;
    }
    // a zero / infinite / NaN pivot shows in its reciprocal (inf / 0 / NaN): one test per lane instead of one per pivot
    if (__any(lane < 16 && (!isfinite(d.myinvd) || d.myinvd == 0.0)) && lane == 0) *ok = 0.0;
    return d.yv;
}

// x_j = sum_k uc[k] rhs[lane k]: four chains of four v_fmac_f64_dpp instead of one of sixteen
template <int K> __device__ __forceinline__ void bsub4(double& xa, double& xb, double& xc, double& xd, double rhs, const double* uc)
{
    fmac_bcast16<K, false>(xa, rhs, uc[K]); fmac_bcast16<K + 1, false>(xb, rhs, uc[K + 1]);
    fmac_bcast16<K + 2, false>(xc, rhs, uc[K + 2]); fmac_bcast16<K + 3, false>(xd, rhs, uc[K + 3]);
}

#ifdef LDLT_PROF
__device__ long long g_ldlt_prof[512];
#define LP(J, k, w) do { if (blockIdx.x == 0 && lane == 0 && wave == (w)) g_ldlt_prof[(J) * 8 + (k)] = clock64(); } while (0)
#else
#define LP(J, k, w)
#endif
struct LdltShared {
    double Lp[2][256 * kLS];        // panel (double-buffered): raw columns of block J, then the finished L rows
    double Wn[256 * kLS];           // -(L D) rows of the panel
    double Ui[16][16 * kLS];        // inverse of the unit lower factor of every diagonal block
    double Inb[2][2][16 * kLS];     // chain wave's inbox, [block & 1]: raw tile (K, K - 1) and diagonal tile (K, K) as of step K - 2
    double Dsc[2][16 * kLS];        // chain wave's own panel tile: L(K, K - 1) and -W(K, K - 1)
    double invd[256], y[256], x[256];
    double part[kLdltWorkers][16];
};

// The chain wave: diagonal block J + 1 only needs panel tile (J + 1, J) and diagonal tile (J + 1, J + 1) as of step J - 1, which
// the workers drop into its inbox one step early. It computes that panel tile itself, applies it to the diagonal tile, forward-
// substitutes the block's right-hand side and factors the block while the workers are still busy with the trailing update of
// step J: the only thing the workers ever wait for is U^-1 of the next block. Its barriers mirror the workers' one for one.
__device__ __forceinline__ void ldlt_chain_wave(LdltShared& sh, int NT, double* ok, int lane, const double* __restrict__ S, const double* __restrict__ b, int ld, int n)
{
    const int lo = lane & 15, hi = lane >> 4;
    const int offC = hi * kLS + lo, offA = lo * kLS + hi;
    [[maybe_unused]] const int wave = kDiagWave;                                // LDLT_PROF builds
    double ar[16], xr[16];
    {   // diagonal block 0 straight from memory (lane lo = row lo of the mirrored triangle) while the workers stage their tiles
        double* const T = &sh.Inb[0][1][0];     // the inbox of block 2: first written in step 0
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = S[(lo >> 2) * 64 + (lo & 3) * 16 + k];          // tile 0, row lo: 128 contiguous bytes per lane
        const double yv0 = b[min(lo, n - 1)];
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) T[lo * kLS + k] = (lo < n && k < n) ? v[k] : (lo == k ? 1.0 : 0.0);
        }
        const double yv = ldlt_diag16(ar, xr, T, &sh.Ui[0][0], sh.invd, lo < n ? yv0 : 0.0, ok, lane);
        if (lane < 16) sh.y[lo] = yv;
    }
    __syncthreads();                            // workers: block column 0 published
    for (int J = 0; J + 1 < NT; ++J) {
        const int par = (J + 1) & 1;
        const double* const R = &sh.Inb[par][0][0];
        double* const Dg = &sh.Inb[par][1][0];
        const double* const UiB = &sh.Ui[J][offA];
        const double invd_c = sh.invd[16 * J + lo];
        double4_t w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) w = __builtin_amdgcn_mfma_f64_16x16x4f64(R[offA + 4 * sl], UiB[4 * sl], w, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { sh.Dsc[0][offC + 4 * r * kLS] = w[r] * invd_c; sh.Dsc[1][offC + 4 * r * kLS] = -w[r]; }
        double4_t dg;
#pragma unroll
        for (int r = 0; r < 4; ++r) dg[r] = Dg[offC + 4 * r * kLS];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.Dsc[0][offA + 4 * sl], sh.Dsc[1][offA + 4 * sl], dg, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Dg[offC + 4 * r * kLS] = dg[r];
        double yv = sh.y[16 * (J + 1) + lo];
#pragma unroll
        for (int j = 0; j < 16; ++j) yv = fma(-sh.Dsc[0][lo * kLS + j], sh.y[16 * J + j], yv);
        LP(J, 5, kDiagWave);
        __syncthreads();
        LP(J, 6, kDiagWave);
        // (all 64 lanes run it: with only the 16 useful lanes active it takes the same 3 500 cycles -- EXEC does not shorten a pass)
        yv = ldlt_diag16(ar, xr, Dg, &sh.Ui[J + 1][0], sh.invd + 16 * (J + 1), yv, ok, lane);
        if (lane < 16) sh.y[16 * (J + 1) + lo] = yv;
        LP(J, 7, kDiagWave);
        __syncthreads();
    }
    __syncthreads();
    for (int J = NT - 1; J >= 0; --J) {                                       // x_J = U^-T (z_J - sum_I L(I, J)^T x_I)
        double uc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) uc[k] = sh.Ui[J][k * kLS + lo];          // U^-1[k][lo]: in flight while the owners reduce
        __syncthreads();
        double sum = sh.part[0][lo];
#pragma unroll
        for (int w = 1; w < kLdltWorkers; ++w) sum += sh.part[w][lo];
        double rhs = sh.x[16 * J + lo] - sum;
        asm volatile("s_nop 1" : "+v"(rhs));                                    // rhs is a DPP source right away
        double xa = 0.0, xb = 0.0, xc = 0.0, xd = 0.0;
        bsub4<0>(xa, xb, xc, xd, rhs, uc); bsub4<4>(xa, xb, xc, xd, rhs, uc); bsub4<8>(xa, xb, xc, xd, rhs, uc); bsub4<12>(xa, xb, xc, xd, rhs, uc);
        if (lane < 16) sh.x[16 * J + lo] = (xa + xb) + (xc + xd);
        __syncthreads();
    }
}

template <int kLdltSlots>       // kLdltSlotsSmall: at most 15 block rows (n <= 240, 120 tiles), kLdltSlotsBig: 16 block rows (136 tiles)
__global__ __launch_bounds__(kLdltThreads) void k_ldlt_mfma(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls)
{
    const BaProb& pb = probs[blockIdx.x];
    if (ctls[blockIdx.x].state > ST_RETRY || pb.np == 0 || pb.use_reg != 1) return;
    const double* __restrict__ S = pb.S;
    const double* __restrict__ b = pb.bsch;
    double* __restrict__ x = pb.xp;
    double* __restrict__ ok = &ctls[blockIdx.x].ok;
    const int ld = pb.ld, n = pb.n;
    __shared__ LdltShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = (n + 15) >> 4, n_pad = NT << 4;
#ifdef LDLT_PROF
    if (blockIdx.x == 0 && lane == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_ldlt_prof[21 * 8 + wave] = 1000 + ((hw >> 4) & 3); }
#endif
    LP(20, 0, 0);
    if (wave == kDiagWave) { ldlt_chain_wave(sh, NT, ok, lane, S, b, ld, n); return; }
    // ---- workers
    const int lo = lane & 15, hi = lane >> 4;
    // every LDS address below is a lane base + a wave-uniform tile offset + a compile-time constant
    const int offC = hi * kLS + lo;            // accumulator layout: element (hi + 4 r, lo) of a tile at offC + 4 r kLS
    const int offA = lo * kLS + hi;            // MFMA operand layout: element (row lo, k = hi + 4 sl) at offA + 4 sl
    // slot s of worker w holds tile t = kLdltWorkers s + w of the row-major lower triangle, t = I (I + 1) / 2 + K
    int tIK[kLdltSlots];                       // I | K << 8 (one SGPR per slot), -1 = no tile
#define tI(s) (tIK[s] < 0 ? -1 : (tIK[s] & 255))
#define tK(s) (tIK[s] >> 8)
#pragma unroll
    for (int s = 0; s < kLdltSlots; ++s) {
        const int t = kLdltWorkers * s + wave;
        int I = (int)((__builtin_amdgcn_sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);      // raw v_sqrt_f32: the two corrections below absorb its error
        if ((I + 1) * (I + 2) / 2 <= t) ++I;
        if (I * (I + 1) / 2 > t) --I;
        const int K = t - I * (I + 1) / 2;
        tIK[s] = __builtin_amdgcn_readfirstlane(I >= NT ? -1 : (I | (K << 8)));
    }
    // slot sets per block column, one bit per slot, column k in lane k (fetched with v_readlane in step k): the tiles of column k
    // below the diagonal, and every tile right of column k. A step only visits the slots it has work for -- walking all 20 slots
    // with scalar tests cost more than the matrix-core instructions themselves.
    unsigned colv = 0, gtv = 0;
#pragma unroll
    for (int s = 0; s < kLdltSlots; ++s) {
        const int I = tI(s), K = tK(s);
        if (K >= 0) {
            if (K == lo && I > lo) colv |= 1u << s;
            if (K > lo) gtv |= 1u << s;
        }
    }
    // Tiles come in straight from the tile storage k_schur writes (s_tile_off): register r of a tile = 64 consecutive doubles, all loads
    // in flight at once (unconditional: a branch around them makes the compiler sink each one to its use; slots without a tile read
    // tile 0). Rows / columns past n become the identity.
    LP(20, 2, 0);
    double4_t acc[kLdltSlots];
#pragma unroll
    for (int s = 0; s < kLdltSlots; ++s) {
        const int I = max(tI(s), 0), K = max(tK(s), 0);
        const double* const T = S + (size_t)(I * (I + 1) / 2 + K) * 256 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = T[64 * r];
    }
    LP(20, 3, 0);
    if (n < n_pad) {                                                             // wave-uniform: only a padded system has tiles to fix up
#pragma unroll
        for (int s = 0; s < kLdltSlots; ++s) {
            const int I = max(tI(s), 0), K = max(tK(s), 0);
            if (I < NT - 1) continue;                                            // padding lives in the last block row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + hi + 4 * r, j = 16 * K + lo;
                if (!(i < n && j < n)) acc[s][r] = i == j ? 1.0 : 0.0;
            }
        }
    }
    LP(20, 4, 0);
    LP(20, 5, 0);
    // ---- prologue: publish block column 0 and the chain wave's inputs for block 1; the chain wave factors diagonal block 0
#pragma unroll
    for (int s = 0; s < kLdltSlots; ++s) {
        const int I = tI(s), K = tK(s);
        if (K == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.Lp[0][16 * I * kLS + offC + 4 * r * kLS] = acc[s][r];
        }
        if (I == 1) {                                                            // tiles (1, 0) and (1, 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.Inb[1][K][offC + 4 * r * kLS] = acc[s][r];
        }
    }
    if (tid >= 16 && tid < 256) sh.y[tid] = tid < n ? b[tid] : 0.0;             // rows 0 .. 15 belong to the chain wave
    LP(20, 1, 0);
    __syncthreads();
    for (int J = 0; J + 1 < NT; ++J) {
        double* const Lc = sh.Lp[J & 1];           // panel of this step
        double* const Ln = sh.Lp[(J & 1) ^ 1];     // panel of the next step (raw columns of block J + 1)
        LP(J, 0, 0);
        {   // ---- rows below the diagonal block: W = A U^-T on the matrix cores, L = W D^-1; tiles dealt round-robin to the workers
            const double* const UiB = &sh.Ui[J][offA];                          // B[k = 4 sl + hi][j = lo] = U^-1[lo][4 sl + hi]
            const double invd_c = sh.invd[16 * J + lo];
            for (int I = J + 1 + wave; I < NT; I += kLdltWorkers) {            // wave-uniform
                double* const T = Lc + 16 * I * kLS;
                double4_t w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) w = __builtin_amdgcn_mfma_f64_16x16x4f64(T[offA + 4 * sl], UiB[4 * sl], w, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { T[offC + 4 * r * kLS] = w[r] * invd_c; sh.Wn[16 * I * kLS + offC + 4 * r * kLS] = -w[r]; }
            }
        }
        LP(J, 1, 0);
        __syncthreads();
        LP(J, 2, 0);
        // ---- update: C(I, K) -= L(I, J) W(K, J)^T for every owned tile right of column J; the tiles of column J + 1 are
        // published as the next panel, tiles (J + 2, J + 1) and (J + 2, J + 2) also to the chain wave's inbox; the finished L
        // tiles of column J return to registers (back substitution)
        const unsigned m_eq = (unsigned)__builtin_amdgcn_readlane((int)colv, J), m_gt = (unsigned)__builtin_amdgcn_readlane((int)gtv, J);
        // (two loops: with the reload and the update of a slot in one if / else the register allocator routes both through a
        // temporary tile and drains the matrix pipe twice per tile to copy it. Requesting the next slot's operands ahead of this
        // slot's matrix-core instructions was measured twice: the double buffer pushes the kernel into spills, 71 -> 89 us.)
#pragma unroll
        for (int s = 0; s < kLdltSlots; ++s) {
            if (!(m_gt >> s & 1)) continue;                                      // one scalar bit test per idle slot
            const int I = tI(s), K = tK(s);                                      // wave-uniform
            const double* const Ar = Lc + 16 * I * kLS + offA;
            const double* const Br = sh.Wn + 16 * K * kLS + offA;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[4 * sl], Br[4 * sl], acc[s], 0, 0, 0);
            if (K == J + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Ln[16 * I * kLS + offC + 4 * r * kLS] = acc[s][r];
                if (I == J + 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.Inb[J & 1][0][offC + 4 * r * kLS] = acc[s][r];
                }
            } else if (I == J + 2 && K == J + 2) {                               // the diagonal tile after next
#pragma unroll
                for (int r = 0; r < 4; ++r) sh.Inb[J & 1][1][offC + 4 * r * kLS] = acc[s][r];
            }
        }
#pragma unroll
        for (int s = 0; s < kLdltSlots; ++s) {
            if (!(m_eq >> s & 1)) continue;
            const int I = tI(s);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = Lc[16 * I * kLS + offC + 4 * r * kLS];
        }
        {   // ---- right-hand side of the rows below block J + 1 (one thread per row)
            const int row = 16 * (J + 2) + tid;
            if (tid < 256 && row < n_pad) {
                double yacc = sh.y[row];
#pragma unroll
                for (int j = 0; j < 16; ++j) yacc = fma(-Lc[row * kLS + j], sh.y[16 * J + j], yacc);
                sh.y[row] = yacc;
            }
        }
        LP(J, 4, 0);
        __syncthreads();
    }
    LP(NT, 0, 0);
    if (tid < n_pad) sh.x[tid] = sh.y[tid] * sh.invd[tid];                    // z = D^-1 y
    __syncthreads();
    for (int J = NT - 1; J >= 0; --J) {
        double c = 0.0;                                                       // owners of block column J: L(I, J)^T x_I
        const unsigned m_eq = (unsigned)__builtin_amdgcn_readlane((int)colv, J);
#pragma unroll
        for (int s = 0; s < kLdltSlots; ++s) {
            if (!(m_eq >> s & 1)) continue;
            const int I = tI(s);
#pragma unroll
            for (int r = 0; r < 4; ++r) c = fma(acc[s][r], sh.x[16 * I + hi + 4 * r], c);
        }
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        if (lane < 16) sh.part[wave][lane] = c;
        __syncthreads();
        __syncthreads();
    }
    LP(NT + 1, 0, 0);
    if (tid < n) x[tid] = sh.x[tid];
#undef tI
#undef tK
}

// x = S^-1 b with S = L D L^T already factored in place (unit lower L below the diagonal, D on it). single block.
// ---- n > 256 (more than 42 free poses: Optimizer.cc:415-422 takes EVERY covisible key frame, BundleAdjustment the whole map): blocked LDL^T
// over many workgroups, ONE launch per block column and no hand-over between workgroups of a launch (round 6). S is a row-major (n_pad + 16) x ld
// array of 16 x 16 tiles: block rows 0 .. m - 1 the matrix (both triangles filled by k_schur), block row m the right-hand side in its first
// row (the factorisation of [S b; b^T .] leaves z = D^-1 L^-1 b there: the forward substitution costs one more tile row).
// Launch s makes block column s final; one wave per tile (I, J), s <= J <= I <= m:
//   every tile     C(I, J) -= W(I, s - 1) L(J, s - 1)^T          the update of column s - 1: 4 v_mfma_f64_16x16x4_f64
//   tiles J == s   additionally: the wave forms the updated DIAGONAL tile (s, s) itself (4 more MFMAs on three tiles every such wave reads anyway),
//                  factors it (ldlt_diag16: one row per lane, DPP broadcasts, ~1.5 us; gives U^-1 and 1 / d) and finishes its own tile:
//                  W(I, s) = C(I, s) U^-T (4 MFMAs), L(I, s) = W D^-1. The m - s waves of the column repeat the same diagonal factorisation
//                  instead of waiting for one another: a hand-over inside a launch costs 1 - 4 us (MI355X_MICROARCH.md price list), the
//                  redundant factorisation 1.5 us and no ordering assumption. Wave (s, s) stores U^-1 and 1 / d for the back substitution
//                  and owns the pivot test; it never writes S(s, s), which the others read.
// W is double-buffered by the parity of s (launch s reads column s - 1 and writes column s). The predecessor (one 256-thread workgroup per
// panel + one tile update launch per panel + a one-workgroup substitution) spent 21 us per panel in its panel kernel and 1.35 ms in the
// substitution at n = 1 188: 3.76 ms per trial against ~0.6 ms here (scratch/time_ba_large.py).
struct StepShared { double R[16 * kLS], Dg[16 * kLS], Ui[16 * kLS], invd[16], ok_other; };
constexpr int kDiagRec = 16 * 16 + 16;             // per diagonal block: U^-1 (row-major 16 x 16), 1 / d

__global__ __launch_bounds__(64) void k_ldlt_step(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls, int s)
{
    const BaProb& pb = probs[blockIdx.z];
    if (ctls[blockIdx.z].state > ST_RETRY || pb.np == 0 || pb.use_reg != 0) return;
    const int m = pb.n_pad / kNB;
    const int I = s + (int)blockIdx.x, J = s + (int)blockIdx.y;
    if (s >= m || J > I || I > m || J >= m) return;
    __shared__ StepShared sh;
    double* __restrict__ S = pb.S;
    const int ld = pb.ld, n = pb.n, lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
    const size_t w_rows = (size_t)pb.n_pad + kNB;
    const double* __restrict__ Wprev = pb.W + (size_t)((s + 1) & 1) * w_rows * kNB;      // column s - 1
    double* __restrict__ Wnext = pb.W + (size_t)(s & 1) * w_rows * kNB;
    const int i0 = kNB * I, j0 = kNB * J, k0 = kNB * (s - 1);
    // C(I, J) as the matrix cores hold it: lane l, register r = element (row (l >> 4) + 4 r, column l & 15). The right-hand side's tile row is
    // read from bsch until it has been written once (launch 0 touches column 0 only, launch 1 every column)
    auto load_tile = [&](int ti0, int tj0, bool rhs_fresh) {
        double4_t c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (rhs_fresh) c[r] = (hi + 4 * r == 0 && tj0 + lo < n) ? pb.bsch[tj0 + lo] : 0.0;
            else c[r] = S[(size_t)(ti0 + hi + 4 * r) * ld + tj0 + lo];
        }
        return c;
    };
    auto update = [&](double4_t c, int ti0, int tj0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double a = -Wprev[(size_t)(ti0 + lo) * kNB + 4 * kk + hi];
            const double b = S[(size_t)(tj0 + lo) * ld + k0 + 4 * kk + hi];             // L(J, s - 1)[lo][k]
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
        return c;
    };
    double4_t acc = load_tile(i0, j0, I == m && s <= 1);
    if (s > 0) acc = update(acc, i0, j0);
    if (J > s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(size_t)(i0 + hi + 4 * r) * ld + j0 + lo] = acc[r];
        return;
    }
    // ---- column s: the diagonal tile (formed here unless this IS its wave), its factorisation, this wave's panel tile
    const int offC = hi * kLS + lo, offA = lo * kLS + hi;
    double4_t dg = acc;
    if (I != s) { dg = load_tile(j0, j0, false); if (s > 0) dg = update(dg, j0, j0); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sh.R[offC + 4 * r * kLS] = acc[r]; sh.Dg[offC + 4 * r * kLS] = dg[r]; }
    __syncthreads();
    double ar[16], xr[16];
    (void)ldlt_diag16(ar, xr, sh.Dg, sh.Ui, sh.invd, 0.0, I == s ? &ctls[blockIdx.z].ok : &sh.ok_other, lane);
    __syncthreads();
    if (I == s) {
        double* rec = pb.Dgf + (size_t)s * kDiagRec;
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[(hi + 4 * r) * 16 + lo] = sh.Ui[offC + 4 * r * kLS];
        if (lane < 16) rec[256 + lane] = sh.invd[lane];
        return;
    }
    double4_t w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.R[offA + 4 * sl], sh.Ui[offA + 4 * sl], w, 0, 0, 0);   // W = C U^-T
    const double invd_c = sh.invd[lo];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        S[(size_t)(i0 + hi + 4 * r) * ld + j0 + lo] = w[r] * invd_c;                     // L(I, s)
        Wnext[(size_t)(i0 + hi + 4 * r) * kNB + lo] = w[r];
    }
}

// L^T x = z, block columns backwards, one workgroup per problem: z (the first row of S's last block row) sits in LDS; step J forms
// x_J = U_J^-T z_J on 16 lanes and every thread takes x_J out of its columns of z: z_c -= sum_r L(16 J + r, c) x_J[r] (tile row J of S, read
// along the rows: coalesced). What a step reads from memory does not depend on the steps before it and is requested before the step's barriers.
constexpr int kBackT = 1024;
__global__ __launch_bounds__(kBackT) void k_ldlt_back(const BaProb* __restrict__ probs, const BaCtl* __restrict__ ctls)
{
    const BaProb& pb = probs[blockIdx.x];
    if (ctls[blockIdx.x].state > ST_RETRY || pb.np == 0 || pb.use_reg != 0) return;
    const double* __restrict__ S = pb.S;
    double* __restrict__ x = pb.xp;
    const int ld = pb.ld, n_pad = pb.n_pad, n = pb.n, m = n_pad / kNB;
    extern __shared__ double zz[];                 // n_pad
    __shared__ double xj[kNB];
    const int t = threadIdx.x;
    for (int c = t; c < n_pad; c += kBackT) zz[c] = S[(size_t)n_pad * ld + c];
    for (int J = m - 1; J >= 0; --J) {
        const int r0 = kNB * J, c0 = t, c1 = t + kBackT;
        const bool h0 = c0 < r0, h1 = c1 < r0;
        double u[kNB], l0[kNB], l1[kNB];
        if (t < kNB) {
            const double* rec = pb.Dgf + (size_t)J * kDiagRec;
#pragma unroll
            for (int k = 0; k < kNB; ++k) u[k] = rec[k * 16 + t];                        // U^-1[k][t]
        }
#pragma unroll
        for (int r = 0; r < kNB; ++r) { l0[r] = h0 ? S[(size_t)(r0 + r) * ld + c0] : 0.0; l1[r] = h1 ? S[(size_t)(r0 + r) * ld + c1] : 0.0; }
        __syncthreads();                           // z_J is final: every earlier step has taken its x out
        if (t < kNB) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kNB; ++k) a = fma(u[k], zz[r0 + k], a);
            xj[t] = a;
            if (r0 + t < n) x[r0 + t] = a;
        }
        __syncthreads();
        if (h0) { double a = zz[c0];
#pragma unroll
            for (int r = 0; r < kNB; ++r) a = fma(-l0[r], xj[r], a); zz[c0] = a; }
        if (h1) { double a = zz[c1];
#pragma unroll
            for (int r = 0; r < kNB; ++r) a = fma(-l1[r], xj[r], a); zz[c1] = a; }
        for (int c = t + 2 * kBackT; c < r0; c += kBackT) {                              // (more than 2 048 columns: 341 free poses)
            double a = zz[c];
            for (int r = 0; r < kNB; ++r) a = fma(-S[(size_t)(r0 + r) * ld + c], xj[r], a);
            zz[c] = a;
        }
    }
}

// Landmark back-substitution + oplus of every estimate into the OTHER estimate buffer (g2o's push + update: the current
// buffer is the backup, nothing has to be copied back on a rejected trial) and the block partials of computeScale, one
// launch of 64-thread blocks: blocks [0, nb_pts) thread per landmark (xl = Dinv (bl - sum_e Hpl[e]^T xp[pose(e)])),
// blocks [nb_pts, ..) thread per pose (pose <- exp(dx) * pose). scale = sum_j x_j (lambda x_j + b_j) over pose and
// active landmark entries; the block sums are added in index order by k_error<1>'s finisher.
__global__ __launch_bounds__(64) void k_solve_update(const BaProb* __restrict__ probs, const BaCtl* __restrict__ ctls)
{
    const BaProb& pb = probs[blockIdx.y];
    const BaCtl& ctl = ctls[blockIdx.y];
    if (ctl.state > ST_RETRY) return;
    const int nb_pts = pb.nb_pts, L = pb.L, P = pb.P;
    if ((int)blockIdx.x >= nb_pts + pb.nb_pose) return;
    const double lambda = 1e-5 * ctl.maxdiag * ctl.mult;
    const int32_t* __restrict__ pose_idx = pb.pose_idx;
    const double* __restrict__ xp = pb.xp;
    double sc = 0;
    if ((int)blockIdx.x < nb_pts) {
        const int l = blockIdx.x * 64 + threadIdx.x;
        if (l < L) {
            const double* __restrict__ bl = pb.bl[ctl.cur];
            double x[3] = {0, 0, 0};
            if (pb.pt_active[ctl.cur][l]) {
                double c[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
                if (pb.np) {
                    // (edge, free-pose index) come from two parallel CSR arrays: one hop to the operands instead of three through
                    // epose / pose_idx; the index pair of the next entry is already in flight while this one is applied
                    const int32_t* __restrict__ pt_edges = pb.pt_edges;
                    const int32_t* __restrict__ pt_pi = pb.pt_pi;
                    const int k1 = pb.pt_off[l + 1];
                    int k = pb.pt_off[l];
                    int e = k < k1 ? pt_edges[k] : 0, pi = k < k1 ? pt_pi[k] : -1;
                    for (; k < k1; ++k) {
                        const int e_n = k + 1 < k1 ? pt_edges[k + 1] : 0, pi_n = k + 1 < k1 ? pt_pi[k + 1] : -1;
                        if (pi >= 0) {
                            const double* B = pb.Hpl[ctl.cur] + (size_t)e * 18;
                            const double* xq = xp + pi * 6;
                            for (int j = 0; j < 3; ++j) for (int r = 0; r < 6; ++r) c[j] -= B[r * 3 + j] * xq[r];
                        }
                        e = e_n; pi = pi_n;
                    }
                }
                const double* D = pb.Dinv + (size_t)l * 9;
                for (int i = 0; i < 3; ++i) {
                    x[i] = D[i * 3] * c[0] + D[i * 3 + 1] * c[1] + D[i * 3 + 2] * c[2];
                    sc += x[i] * (lambda * x[i] + bl[3 * l + i]);
                }
            }
            const double* __restrict__ src = pb.points[ctl.cur];
            double* __restrict__ dst = pb.points[ctl.cur ^ 1];
            for (int k = 0; k < 3; ++k) { pb.xl[3 * l + k] = x[k]; dst[3 * l + k] = src[3 * l + k] + x[k]; }
        }
    } else {
        const int i = (blockIdx.x - nb_pts) * 64 + threadIdx.x;
        if (i < P) {
            const double* __restrict__ src = pb.poses[ctl.cur];
            double* __restrict__ dst = pb.poses[ctl.cur ^ 1];
            double T[7];
            for (int k = 0; k < 7; ++k) T[k] = src[7 * i + k];
            const int pi = pose_idx[i];
            if (pi >= 0) {
                double o[7];
                pose_oplus(T, xp + 6 * pi, o);
                for (int k = 0; k < 7; ++k) dst[7 * i + k] = o[k];
                for (int k = 0; k < 6; ++k) sc += xp[6 * pi + k] * (lambda * xp[6 * pi + k] + pb.bp[6 * pi + k]);
            } else {
                for (int k = 0; k < 7; ++k) dst[7 * i + k] = T[k];
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sc += __shfl_xor(sc, d);
    if (threadIdx.x == 0) pb.scale_part[blockIdx.x] = sc;
}

// k_solve_update + k_error<1> in ONE launch (a launch boundary costs a step ~4 us plus the second kernel's ramp; the step was 7 launches):
// workgroup per 64 landmarks, 320 threads.
//   (a) wave 4 of every workgroup maps ALL poses into LDS (pose <- exp(dx) * pose: a few hundred flops per pose, cheaper than a pass through HBM and a
//       launch boundary); workgroup 0 also stores them and owns the poses' computeScale terms
//   (b) waves 0-3, four lanes per landmark: the back-substitution of k_solve_update; the new point goes to LDS and to the other buffer
//   (c) all threads: the edges of the workgroup's landmarks are ONE contiguous run of the point CSR -- residuals and chi2 of the trial
//       estimates, thread per entry (k_error<1>'s arithmetic per edge; only the ORDER of the chi2 sum differs: CSR order, partial per
//       workgroup, partials added in index order by the finisher)
//   (d) the problem's last workgroup (ticket) runs the accept / reject logic and publishes the step like k_error<1>.
// Problems with more than kFusedMaxPoses poses keep the two launches.
constexpr int kFusedMaxPoses = 512;
constexpr int kFusedThreads = 320;                       // waves 0-3: four lanes per landmark, wave 4: the poses
// two sums at once over a workgroup of kFusedThreads threads (fixed tree; the totals come back in thread 0)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* s /*[512]*/, double* s2 /*[512]*/)
{
    const int t = threadIdx.x;
    s[t] = a; s2[t] = b;
    if (t < 512 - kFusedThreads) { s[kFusedThreads + t] = 0; s2[kFusedThreads + t] = 0; }
    __syncthreads();
    for (int d = 256; d >= 1; d >>= 1) { if (t < d) { s[t] += s[t + d]; s2[t] += s2[t + d]; } __syncthreads(); }
    a = s[0]; b = s2[0];
}
__global__ __launch_bounds__(kFusedThreads) void k_update_error(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls, const volatile int* __restrict__ stop_words, int B,
                                                      int* __restrict__ h_progress, unsigned* __restrict__ grid_ticket)
{
    __shared__ double s[512], s2[512];
    __shared__ DCams cams;
    __shared__ double s_pose[kFusedMaxPoses * 7];
    __shared__ double s_pt[64 * 3];
    __shared__ bool last;
    const BaProb& pb = probs[blockIdx.y];
    BaCtl& ctl = ctls[blockIdx.y];
    if (ctl.state > ST_RETRY) return;
    const int nb_pts = pb.nb_pts, L = pb.L, P = pb.P, t = threadIdx.x;
    if ((int)blockIdx.x >= nb_pts) return;
    // the control word as of the start of the launch: the finisher rewrites it, but only after every workgroup of the problem has arrived
    const int cur = ctl.cur, robust = ctl.robust;
    const double lambda = 1e-5 * ctl.maxdiag * ctl.mult, delta = pb.delta;
    const int32_t* __restrict__ pose_idx = pb.pose_idx;
    const double* __restrict__ xp = pb.xp;
    load_cams(&cams, pb.cams);
    // ---- (a)
    double sc_pose = 0;
    if (t >= 256 && t < kFusedThreads) {                     // wave 4, while waves 0-3 back-substitute
        const double* __restrict__ src = pb.poses[cur];
        double* __restrict__ dst = pb.poses[cur ^ 1];
        for (int i = t - 256; i < P; i += 64) {
            double T[7], o[7];
            for (int k = 0; k < 7; ++k) T[k] = src[7 * i + k];
            const int pi = pose_idx[i];
            if (pi >= 0) {
                pose_oplus(T, xp + 6 * pi, o);
                for (int k = 0; k < 6; ++k) sc_pose += xp[6 * pi + k] * (lambda * xp[6 * pi + k] + pb.bp[6 * pi + k]);
            } else {
                for (int k = 0; k < 7; ++k) o[k] = T[k];
            }
            for (int k = 0; k < 7; ++k) s_pose[7 * i + k] = o[k];
            if (blockIdx.x == 0) for (int k = 0; k < 7; ++k) dst[7 * i + k] = o[k];
        }
    }
    // ---- (b) four lanes per landmark: a thread of its own walks a landmark's ~15 edges one dependent load pair at a time (index -> block
    //      of H_pl), which was most of k_solve_update's 9.7 us; lane `sub` takes the entries k = sub (mod 4) and the four partial
    //      sums are added as (p0 + p1) + (p2 + p3)
    double sc = 0;
    if (t < 256) {
        const int lt = t >> 2, sub = t & 3;
        const int l = blockIdx.x * 64 + lt;
        const bool on = l < L && pb.pt_active[cur][l];
        double acc[3] = {0, 0, 0};
        if (on && pb.np) {
            const int32_t* __restrict__ pt_edges = pb.pt_edges;
            const int32_t* __restrict__ pt_pi = pb.pt_pi;
            const int k1 = pb.pt_off[l + 1];
            for (int k = pb.pt_off[l] + sub; k < k1; k += 4) {
                const int pi = pt_pi[k];
                if (pi < 0) continue;
                const double* Bm = pb.Hpl[cur] + (size_t)pt_edges[k] * 18;
                const double* xq = xp + pi * 6;
                for (int j = 0; j < 3; ++j) for (int r = 0; r < 6; ++r) acc[j] += Bm[r * 3 + j] * xq[r];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { acc[j] += __shfl_xor(acc[j], 1); acc[j] += __shfl_xor(acc[j], 2); }
        if (sub == 0 && l < L) {
            const double* __restrict__ bl = pb.bl[cur];
            double x[3] = {0, 0, 0};
            if (on) {
                const double c[3] = {bl[3 * l] - acc[0], bl[3 * l + 1] - acc[1], bl[3 * l + 2] - acc[2]};
                const double* D = pb.Dinv + (size_t)l * 9;
                for (int i = 0; i < 3; ++i) {
                    x[i] = D[i * 3] * c[0] + D[i * 3 + 1] * c[1] + D[i * 3 + 2] * c[2];
                    sc += x[i] * (lambda * x[i] + bl[3 * l + i]);
                }
            }
            const double* __restrict__ src = pb.points[cur];
            double* __restrict__ dst = pb.points[cur ^ 1];
            for (int k = 0; k < 3; ++k) {
                const double v = src[3 * l + k] + x[k];
                pb.xl[3 * l + k] = x[k]; dst[3 * l + k] = v; s_pt[3 * lt + k] = v;
            }
        }
    }
    if (blockIdx.x == 0 && t >= 256 && t < kFusedThreads) {  // the poses' share of computeScale: wave 4's lanes in a fixed tree
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sc_pose += __shfl_xor(sc_pose, d);
        if (t == 256) pb.scale_part[nb_pts] = sc_pose;
    }
    __syncthreads();
    // ---- (c)
    double rho0 = 0;
    {
        const int l0 = blockIdx.x * 64, l1 = min(l0 + 64, L);
        const int k0 = pb.pt_off[l0], k1 = pb.pt_off[l1];
        const int32_t* __restrict__ pt_edges = pb.pt_edges;
        for (int k = k0 + t; k < k1; k += kFusedThreads) {
            const int e = pt_edges[k];
            if (!pb.active[e]) continue;
            double pc[3];
            const DCam& c = cams.c[pb.ecam[e]];
            cam_point(s_pose + 7 * pb.epose[e], s_pt + 3 * (pb.epoint[e] - l0), c, pc);
            const double e0 = pb.obs[2 * e] - (pc[0] / pc[2] * c.fx + c.cx);
            const double e1 = pb.obs[2 * e + 1] - (pc[1] / pc[2] * c.fy + c.cy);
            const double w = pb.w[e];
            const double x2 = e0 * (w * e0) + e1 * (w * e1);
            pb.err[2 * e] = e0; pb.err[2 * e + 1] = e1; pb.chi2[e] = x2;
            rho0 += (robust && x2 > delta * delta) ? 2 * sqrt(x2) * delta - delta * delta : x2;
        }
    }
    block_sum2(rho0, sc, s, s2);                       // thread 0: chi2 and computeScale partials of this workgroup
    // ---- (d)
    if (t == 0) {
        pb.scale_part[blockIdx.x] = sc;
        __hip_atomic_store(&pb.partial[blockIdx.x], rho0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(pb.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (prev == (unsigned)nb_pts - 1);
    }
    __syncthreads();
    if (!last) return;
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double tot = 0, sc_tot = 0;
    if (t < kFusedThreads) {
        for (int i = t; i < nb_pts; i += kFusedThreads) tot += __hip_atomic_load(&pb.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = t; i <= nb_pts; i += kFusedThreads) sc_tot += __hip_atomic_load(&pb.scale_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum2(tot, sc_tot, s, s2);
    if (t != 0) return;
    __hip_atomic_store(pb.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lm_accept(pb, ctl, stop_words, tot, sc_tot);
    publish_step_end(B, h_progress, grid_ticket);
}

__global__ __launch_bounds__(1024) void k_ctl_init(const BaProb* __restrict__ probs, BaCtl* __restrict__ ctls, int B)
{
    for (int b = threadIdx.x; b < B; b += 1024) {
        BaCtl c{};
        c.state = probs[b].iters[0] > 0 ? ST_NEW_ITER : ST_ROUND_END;
        c.robust = probs[b].robust0;
        c.mult = 1.0; c.ni = 2; c.ok = 1.0;
        ctls[b] = c;
    }
}

// n > 256 fallback: the blocked factorisation works in place, so S is cleared and its padding rows get a unit diagonal
__global__ void k_pad_identity(const BaProb* __restrict__ probs, const BaCtl* __restrict__ ctls)
{
    const BaProb& pb = probs[blockIdx.y];
    if (ctls[blockIdx.y].state > ST_RETRY || pb.np == 0 || pb.use_reg != 0) return;
    const int i = pb.n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pb.n_pad) pb.S[(size_t)i * pb.ld + i] = 1.0;
}

// ------------------------------------------------------------------ Optimizer::PoseOptimization (Optimizer.cc:250-405)
// One workgroup per frame runs the whole procedure on the device: 4 rounds x <= 10 LM iterations x <= 10 trials, each
// trial = edge-parallel residuals / Jacobians (threads stride over the frame's unary edges), a fixed-order block
// reduction of the 21 + 6 entries of the 6x6 system, a dense LDL^T by thread 0, the manifold update and the chi2 of
// the trial -- g2o's accept / reject logic is evaluated by thread 0 and broadcast through LDS. Frames are independent,
// so a batch (one frame per stream) fills the GPU; a single frame costs a few hundred microseconds of latency.
struct PoseArgs {
    const double* poses; const int32_t* edge_off; const double* xw; const double* obs; const double* w; const int32_t* cam;
    const int32_t* edge_cnt;                     // optional: frame f owns edge_cnt[f] edges from edge_off[f] (slotted layout of the tracking chain) instead of a CSR
    double huber; float chi2_th[4]; int its[4];
    double* err; uint8_t* level;                 // scratch per edge
    double* out_poses; uint8_t* outlier; int32_t* n_inliers; double* edge_chi2; int32_t* n_iters;
    int fast_max;                                // frames of up to fast_max edges belong to k_pose_opt2, larger ones to k_pose_opt (-1: every frame to k_pose_opt)
};

// Jacobian of the projection w.r.t. the rig pose (EdgeSE3ProjectXYZOnlyPose::linearizeOplus, types_six_dof_expmap.cpp:218-246)
__device__ inline void pose_jacobian(const double* T, const double* X, const DCam& c, double Jp[12])
{
    double pc[3];
    cam_point(T, X, c, pc);
    const double x = pc[0], y = pc[1], z = pc[2];
    const double s = -1. / z;
    const double st[6] = {s * c.fx, s * 0.0, s * (-x / z * c.fx), s * 0.0, s * c.fy, s * (-y / z * c.fy)};
    const double J3[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    double A[12];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) A[i * 6 + j] = st[i * 3] * J3[j] + st[i * 3 + 1] * J3[6 + j] + st[i * 3 + 2] * J3[12 + j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) {
        double acc = 0;
        for (int k = 0; k < 6; ++k) acc += A[i * 6 + k] * c.adj[k * 6 + j];
        Jp[i * 6 + j] = acc;
    }
}

// dense LDL^T solve of a 6x6 system (LinearSolverDense; fails like isPositive() on a non-positive pivot)
__device__ inline bool solve6(const double* H, double lambda, const double* b, double* x)
{
    double A[36];
    for (int i = 0; i < 36; ++i) A[i] = H[i];
    for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
    double dd[6];
    for (int j = 0; j < 6; ++j) {
        double v = A[j * 6 + j];
        for (int k = 0; k < j; ++k) v -= A[j * 6 + k] * A[j * 6 + k] * dd[k];
        if (!(v > 0.0) || !isfinite(v)) return false;
        dd[j] = v;
        for (int i = j + 1; i < 6; ++i) {
            double a = A[i * 6 + j];
            for (int k = 0; k < j; ++k) a -= A[i * 6 + k] * A[j * 6 + k] * dd[k];
            A[i * 6 + j] = a / v;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double a = b[i]; for (int k = 0; k < i; ++k) a -= A[i * 6 + k] * y[k]; y[i] = a; }
    for (int i = 0; i < 6; ++i) y[i] /= dd[i];
    for (int i = 5; i >= 0; --i) { double a = y[i]; for (int k = i + 1; k < 6; ++k) a -= A[k * 6 + i] * x[k]; x[i] = a; }
    return true;
}

__global__ __launch_bounds__(256) void k_pose_opt(PoseArgs a, DCams cams)
{
    __shared__ double s_red[4][28];
    __shared__ double s_T[7], s_x[6], s_tot[28];
    __shared__ int s_ctl;                                    // decision of thread 0: 0 = next trial, 1 = iteration done, 2 = round done
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = a.edge_off[f], n = a.edge_cnt ? a.edge_cnt[f] : a.edge_off[f + 1] - e0;
    if (n <= a.fast_max && a.n_inliers[f] != INT_MIN) return;   // k_pose_opt2's frame, unless that kernel declined it (kPoseDeclined: a camera share beyond its registers)
    const double* Tin = a.poses + 7 * f;
    for (int k = tid; k < n; k += 256) { a.outlier[e0 + k] = 0; a.level[e0 + k] = 0; if (a.edge_chi2) a.edge_chi2[e0 + k] = 0; }
    if (tid < 4 && a.n_iters) a.n_iters[4 * f + tid] = 0;
    if (n < 3) {                                              // :343-344
        if (tid < 7) a.out_poses[7 * f + tid] = Tin[tid];
        if (tid == 0) a.n_inliers[f] = 0;
        return;
    }
    // fixed-order block sum of NV values per thread: wave shuffles, then the 4 wave partials in order; result in s_tot
    auto block_sum = [&](double* v, int nv) {
        for (int i = 0; i < nv; ++i) {
            double t = v[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
            if (lane == 0) s_red[wave][i] = t;
        }
        __syncthreads();
        if (tid < nv) s_tot[tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        __syncthreads();
    };
    const double delta = a.huber, dsqr = delta * delta;
    auto chi2_of = [&](int k) { const double w = a.w[e0 + k]; const double ex = a.err[2 * (size_t)(e0 + k)], ey = a.err[2 * (size_t)(e0 + k) + 1]; return ex * (w * ex) + ey * (w * ey); };
    bool robust = true;
    int n_bad_edges = 0;
    for (int it = 0; it < 4; ++it) {
        if (tid < 7) s_T[tid] = Tin[tid];                      // :360 every round restarts from the frame's pose
        __syncthreads();
        // errors of the active edges at s_T + robust chi2 -> s_tot[0]; also the number of active edges in s_tot[1]
        auto errors_and_chi = [&]() {
            double v[2] = {0, 0};
            for (int k = tid; k < n; k += 256) {
                if (a.level[e0 + k]) continue;
                const DCam& c = cams.c[a.cam[e0 + k]];
                double pc[3];
                cam_point(s_T, a.xw + 3 * (size_t)(e0 + k), c, pc);
                const double ex = a.obs[2 * (size_t)(e0 + k)] - (pc[0] / pc[2] * c.fx + c.cx);
                const double ey = a.obs[2 * (size_t)(e0 + k) + 1] - (pc[1] / pc[2] * c.fy + c.cy);
                a.err[2 * (size_t)(e0 + k)] = ex; a.err[2 * (size_t)(e0 + k) + 1] = ey;
                const double w = a.w[e0 + k], x2 = ex * (w * ex) + ey * (w * ey);
                v[0] += (robust && x2 > dsqr) ? 2 * sqrt(x2) * delta - dsqr : x2;
                v[1] += 1.0;
            }
            block_sum(v, 2);
        };
        // LM state lives in thread 0's registers
        double lambda = -1, ni = 2, currentChi = 0, iniChi = 0;
        int nBad = 0, n_it = 0;
        bool round_done = false;
        errors_and_chi();
        if (s_tot[1] == 0.0) round_done = true;                // no active edge: optimize() does nothing
        for (int i = 0; i < a.its[it] && !round_done; ++i) {
            if (i > 0) errors_and_chi();                        // computeActiveErrors at the top of every LM iteration
            if (tid == 0) { currentChi = s_tot[0]; iniChi = currentChi; }
            {   // build the 6x6 system
                double v[27];
                for (int q = 0; q < 27; ++q) v[q] = 0;
                for (int k = tid; k < n; k += 256) {
                    if (a.level[e0 + k]) continue;
                    double Jp[12];
                    pose_jacobian(s_T, a.xw + 3 * (size_t)(e0 + k), cams.c[a.cam[e0 + k]], Jp);
                    double w = a.w[e0 + k];
                    const double ex = a.err[2 * (size_t)(e0 + k)], ey = a.err[2 * (size_t)(e0 + k) + 1];
                    double r0 = -w * ex, r1 = -w * ey;
                    if (robust) {
                        const double x2 = ex * (w * ex) + ey * (w * ey);
                        const double rho1 = x2 <= dsqr ? 1.0 : delta / sqrt(x2);
                        r0 *= rho1; r1 *= rho1; w = rho1 * w;
                    }
                    int q = 0;
                    for (int r = 0; r < 6; ++r) for (int c2 = r; c2 < 6; ++c2) v[q++] += Jp[r] * w * Jp[c2] + Jp[6 + r] * w * Jp[6 + c2];
                    for (int r = 0; r < 6; ++r) v[21 + r] += Jp[r] * r0 + Jp[6 + r] * r1;
                }
                block_sum(v, 27);
            }
            double H[36], b[6];
            if (tid == 0) {
                int q = 0;
                for (int r = 0; r < 6; ++r) for (int c2 = r; c2 < 6; ++c2) { H[r * 6 + c2] = s_tot[q]; H[c2 * 6 + r] = s_tot[q]; ++q; }
                for (int r = 0; r < 6; ++r) b[r] = s_tot[21 + r];
                if (i == 0) {                                   // computeLambdaInit
                    double md = 0;
                    for (int d = 0; d < 6; ++d) md = fmax(fabs(H[d * 7]), md);
                    lambda = 1e-5 * md; ni = 2; nBad = 0;
                }
            }
            double rho = 0, bk[7];
            int qmax = 0;
            bool ok2 = true;
            for (;;) {                                          // trials
                if (tid == 0) {
                    for (int d = 0; d < 7; ++d) bk[d] = s_T[d];                          // push
                    double x[6];
                    ok2 = solve6(H, lambda, b, x);
                    if (!ok2) for (int d = 0; d < 6; ++d) x[d] = 0;
                    double o[7];
                    pose_oplus(s_T, x, o);
                    for (int d = 0; d < 7; ++d) s_T[d] = o[d];
                    for (int d = 0; d < 6; ++d) s_x[d] = x[d];
                }
                __syncthreads();
                errors_and_chi();
                if (tid == 0) {
                    double tempChi = ok2 ? s_tot[0] : 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    for (int j = 0; j < 6; ++j) scale += s_x[j] * (lambda * s_x[j] + b[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        double alpha = 1. - pow((2 * rho - 1), 3);
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                    } else {
                        lambda *= ni; ni *= 2;
                        for (int d = 0; d < 7; ++d) s_T[d] = bk[d];                      // pop (errors stay stale, as in g2o)
                    }
                    ++qmax;
                    int ctl = (rho < 0 && qmax < 10) ? 0 : 1;
                    if (ctl == 1) {
                        ++n_it;
                        if (qmax == 10 || rho == 0) ctl = 2;
                        else { if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0; if (nBad >= 3) ctl = 2; }
                    }
                    s_ctl = ctl;
                }
                __syncthreads();
                if (s_ctl != 0) break;
            }
            if (s_ctl == 2) round_done = true;
        }
        if (tid == 0 && a.n_iters) a.n_iters[4 * f + it] = n_it;
        __syncthreads();
        // classification of every edge (:365-390): previous outliers are re-evaluated at the final pose, inliers keep the
        // error of the last evaluation
        double cnt[1] = {0};
        for (int k = tid; k < n; k += 256) {
            if (a.outlier[e0 + k]) {
                const DCam& c = cams.c[a.cam[e0 + k]];
                double pc[3];
                cam_point(s_T, a.xw + 3 * (size_t)(e0 + k), c, pc);
                a.err[2 * (size_t)(e0 + k)] = a.obs[2 * (size_t)(e0 + k)] - (pc[0] / pc[2] * c.fx + c.cx);
                a.err[2 * (size_t)(e0 + k) + 1] = a.obs[2 * (size_t)(e0 + k) + 1] - (pc[1] / pc[2] * c.fy + c.cy);
            }
            const float chi2 = (float)chi2_of(k);
            const bool bad = chi2 > a.chi2_th[it];
            a.outlier[e0 + k] = bad; a.level[e0 + k] = bad;
            cnt[0] += bad;
        }
        block_sum(cnt, 1);
        n_bad_edges = (int)s_tot[0];
        if (it == 2) robust = false;                           // :388-389
        if (n < 10) break;                                     // :392
    }
    if (tid < 7) a.out_poses[7 * f + tid] = s_T[tid];
    if (tid == 0) a.n_inliers[f] = n - n_bad_edges;
    if (a.edge_chi2) for (int k = tid; k < n; k += 256) a.edge_chi2[e0 + k] = chi2_of(k);
}


// ------------------------------------------------------------------ k_pose_opt2 (round 6: rebuilt as a compact kernel)
// The same procedure (Optimizer.cc:250-405, types_six_dof_expmap.cpp:200-255) for frames of up to kPoseFastMax edges, one workgroup of 256
// threads (one wave per SIMD) per frame; k_pose_opt stays for larger frames. Round 5's version of this kernel compiled to 940 KB of code with 13 375 spilled SGPRs
// (a control wave whose every stage was force-inlined three times and broadcast through v_readlane): its passes waited on the instruction cache,
// not on arithmetic. This one has ONE code path that every wave runs:
//  * the edges live in registers (point, last chi2) and LDS (observation, weight), grouped by camera once at the start (counting sort in LDS)
//    so that a WAVE works on one camera: its intrinsics and rig -> camera transform are wave-uniform. Two builds of the per-edge arithmetic
//    (template argument, option DCS_POSE_EXACT_EDGE): the composed world -> camera transform M_c = R_c R(T), m_c = R_c t(T) + t_c, an edge's
//    point 9 FMAs (default); or the point and the residual through the oracle's own operations (cam_point_regs(), quotient(): see there);
//  * J = A adj_c with A the 2 x 6 projection Jacobian in the camera frame (two structural zeros). adj_c is the reference's 6 x 6 matrix as
//    given (NOT the SE3 adjoint, so it cannot be folded into the geometry): the wave accumulates A^T W A (21) and A^T r (6), reduces
//    them across its lanes (one transposed reduction on v_permlane32/16_swap), and applies adj_c^T ( . ) adj_c to its own 27 sums as ONE constant
//    27 x 27 linear map (row q on lane q, built once per call in LDS) before they meet the other waves' -- the same sums as 72 multiply-adds
//    per edge, associated differently;
//  * ONE sweep per trial: a trial's errors and the linearisation at the trial's pose. g2o recomputes exactly those when it accepts (and a
//    rejection re-solves the kept system with a larger lambda);
//  * NO control wave: after the one workgroup barrier of a pass every wave adds the 4 partial systems in wave order and then runs the scalar part
//    -- LM rule, 6 x 6 LDL^T, exp map -- itself, every lane the same arithmetic on the same numbers (a wave64 f64 instruction costs the same
//    for one lane as for 64). Nothing is broadcast, no second
//    barrier, no v_readlane; the partial sums are double-buffered so that a fast wave's next pass cannot overwrite what a slow one still reads;
//  * a round's classification of the edges (:365-390) is one more pass of the same sweep code at the round's final pose, told by a flag.
// Parity bar unchanged (tests/test_gpu_ba.py::test_pose_optimization_vs_oracle, tests/test_gpu_track.py): poses 1e-7 / 1e-8, flags, counts +-1.
constexpr int kPoT = 256, kPoW = kPoT / 64, kPoEpt = 12;     // ONE wave per SIMD: two waves on a SIMD run this f64 code one after the other, not interleaved (tools/pose_timeline.py)
constexpr int kPoseFastMax = 2304;      // (the most a dual rig can split over 12 slots: see the static_asserts; a frame of 2 x 1 048 feature slots stays below it)
// A camera's edges are shared by the waves the greedy split below gives it; a frame whose largest share does not fit kPoEpt slots per lane is left to
// k_pose_opt. The dual rig always fits (the split is 2 + 2 waves when the smaller camera has at least half the larger one's edges, else 3 + 1: 11 slots
// for up to 2 048 edges, 12 for up to 2 304, tests/test_pose_split.py walks them all); a third or fourth camera fits while no camera that is left
// with one wave holds more than 64 * kPoEpt = 768 edges. The slot count is the kernel's code size: the sweep is unrolled over it, 2.8 KB a slot.
static_assert(128 * kPoEpt >= (2 * kPoseFastMax + 2) / 3, "two cameras, two waves each: the larger has at most two thirds of the edges");
static_assert(192 * kPoEpt >= kPoseFastMax && 64 * kPoEpt >= (kPoseFastMax + 2) / 3, "two cameras, three waves and one: the smaller has less than a third");
constexpr int kPoChunks = (kPoseFastMax + kPoT - 1) / kPoT;
constexpr int kPoLine = 48;          // a wave's exchange line: H 0..20, b 22..27, zeros 28..43, chi2 44, active edges 45

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 64 lanes of each of 32 values: lanes 2q and 2q + 1 return the total of v[q]. One fixed tree: halves are exchanged, so every
// step moves half as many values as the one before (16 + 8 + 4 + 2 + 1 + 1 exchanges). The two widest levels -- 24 of the 32 exchanges -- are
// gfx950's v_permlane32_swap / v_permlane16_swap (the upper half of one register trades places with the lower half of the other: exactly
// this step, no LDS round trip, no selects); the narrow ones are DPP moves inside a row of 16 lanes (round 6; ds_bpermute before: four LDS round trips).
__device__ __forceinline__ double swap_add32(double p, double q)
{
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(p), (unsigned)__double2loint(q), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(p), (unsigned)__double2hiint(q), false, false);
    return dbl_of(a[0], b[0]) + dbl_of(a[1], b[1]);
}
__device__ __forceinline__ double swap_add16(double p, double q)
{
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p), (unsigned)__double2loint(q), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p), (unsigned)__double2hiint(q), false, false);
    return dbl_of(a[0], b[0]) + dbl_of(a[1], b[1]);
}
// the value of lane (l ^ 8), (l ^ 4), (l ^ 2), (l ^ 1): DPP moves inside a row of 16 lanes (no LDS round trip as ds_bpermute has)
template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double x)
{
    return dbl_of((unsigned)__builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANKS, false),
                  (unsigned)__builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANKS, false));
}
__device__ __forceinline__ double lane_xor8(double x) { return dpp_f64<0x128>(x, x); }                    // row_ror:8
__device__ __forceinline__ double lane_xor4(double x) { return dpp_f64<0x114, 0xa>(dpp_f64<0x104, 0x5>(x, x), x); }   // row_shl:4 into lanes 0-3, 8-11; row_shr:4 into 4-7, 12-15
__device__ __forceinline__ double lane_xor2(double x) { return dpp_f64<0x4e>(x, x); }                     // quad_perm [2, 3, 0, 1]
__device__ __forceinline__ double lane_xor1(double x) { return dpp_f64<0xb1>(x, x); }                     // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double wave_sum32(double (&v)[32], int lane)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = swap_add32(v[i], v[i + 16]);     // lanes 0-31: v[i] over (l, l + 32); lanes 32-63: v[i + 16]
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = swap_add16(v[i], v[i + 8]);
#define DCS_PO_STEP(N, BIT, XOR) _Pragma("unroll") for (int i = 0; i < N; ++i) { const bool hi = (lane & BIT) != 0; const double send = hi ? v[i] : v[i + N], keep = hi ? v[i + N] : v[i]; v[i] = keep + XOR(send); }
    DCS_PO_STEP(4, 8, lane_xor8)
    DCS_PO_STEP(2, 4, lane_xor4)
    DCS_PO_STEP(1, 2, lane_xor2)
#undef DCS_PO_STEP
    return v[0] + lane_xor1(v[0]);
}

// entry q of the upper triangle of a 6 x 6 matrix, rows first: (0,0) (0,1) .. (0,5) (1,1) .. (5,5)
__device__ inline void tri6(int q, int& r, int& c)
{
    r = 0;
    int base = 0;
    while (q >= base + (6 - r)) { base += 6 - r; ++r; }
    c = r + (q - base);
}

struct alignas(16) PoseShared {
    double part[2][kPoW][32];               // a pass's partial systems, wave by wave (double-buffered)
    double line[kPoW][kPoLine];             // per wave: its own sums of a pass, laid out for the adjoint map
    double tot[kPoW][32];                   // per wave: the totals of a pass for its own lanes
    double keep[kPoW][36];                  // per wave: the adopted system (21 + 6 at 0..26) and the pushed pose (28..34)
    double K[kMaxCams][27][22];             // adj_c^T ( . ) adj_c as a linear map on the 21 + 6 sums: row = output entry
    double ed[3][kPoEpt][kPoT];             // observation (x, y) and weight of every resident edge (72 KB; one workgroup per CU anyway)
    DCam cam[kMaxCams];                     // the rig's cameras (a by-value kernel argument indexed at run time would be copied to scratch)
    double rc[kMaxCams][12];                // (composed-transform build) rig -> camera: rotation matrix, translation
    double series[34];                      // kPoSeries
    uint16_t list[kPoseFastMax];
    int cnt[kMaxCams][kPoChunks * kPoW];
    int ncam[kMaxCams], off[kMaxCams], W[kMaxCams], wcam[kPoW], wu[kPoW], bad[2][kPoW];
};

// 1 / sqrt(x) to full double precision: v_rsq_f64 + two Newton steps (x > 0)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}
// exp(update) * T like pose_oplus(), shaped for the dependent chain between two sweeps:
//   * the three coefficients sin(t)/t, (1 - cos t)/t^2, (t - sin t)/t^3 as even series in t^2 (an LM step rotates by far less than 0.5 rad: 11
//     terms leave < 1e-17; no square root, no sincos, no division -- and no cancellation, which the closed forms have at small t);
//     t >= 0.5 takes the closed forms
//   * normalisations multiply by 1 / sqrt (v_rsq_f64 + two Newton steps) instead of sqrt followed by four IEEE divisions
// Differences to pose_oplus() are of the order of an ulp per operation: the estimates agree to ~1e-15, which moves an accept / reject decision on a
// plateau no more often than the reduction order does (tests: poses 1e-7 / 1e-8, flags exact, iteration counts +- 1).
__device__ inline void quat_normalize_fast(double q[4])
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double r = fast_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}
// coefficients of the three series, highest power first: 1 / (2k+1)!, 1 / (2k+2)!, 1 / (2k+3)! for k = 10 .. 0. The kernel copies them to LDS and
// reads them where it needs them: as literals the compiler hoists all 33 out of the trial loop into scalar register pairs and spills them.
__device__ const double kPoSeries[33] = {
    1.0 / 51090942171709440000.0, 1.0 / 1124000727777607680000.0, 1.0 / 25852016738884976640000.0,
    1.0 / 121645100408832000.0, 1.0 / 2432902008176640000.0, 1.0 / 51090942171709440000.0,
    1.0 / 355687428096000.0, 1.0 / 6402373705728000.0, 1.0 / 121645100408832000.0,
    1.0 / 1307674368000.0, 1.0 / 20922789888000.0, 1.0 / 355687428096000.0,
    1.0 / 6227020800.0, 1.0 / 87178291200.0, 1.0 / 1307674368000.0,
    1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0,
    1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0,
    1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
    1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
    1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0,
    1.0, 1.0 / 2.0, 1.0 / 6.0};
__device__ inline void pose_oplus_fast(const double* T, const double* u, double* out, const double* series)
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double t2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    // O = [om]x and O2 = O O with the structural zeros written out (pose_oplus() multiplies and adds them: the same values -- a product with an
    // exact zero adds nothing -- for 9 instructions instead of 45; likewise R and V below)
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    const double s00 = om[0] * om[0], s11 = om[1] * om[1], s22 = om[2] * om[2], p01 = om[1] * om[0], p02 = om[2] * om[0], p12 = om[2] * om[1];
    const double O2[9] = {-s22 - s11, p01, p02, p01, -s22 - s00, p12, p02, p12, -s11 - s00};
    double R[9], V[9];
    if (t2 < 1e-10) {                                         // theta < 0.00001 (pose_oplus's first-order branch)
        for (int i = 0; i < 9; ++i) { R[i] = i % 4 == 0 ? 1.0 + O2[i] : O[i] + O2[i]; V[i] = R[i]; }
    } else {
        double a, b, c, h = 1.0, ts = t2;                     // ts = (theta h)^2 < 0.25 with h = 2^-halvings
        int halvings = 0;
        while (!(ts < 0.25) && halvings < 1100) { ts *= 0.25; h *= 0.5; ++halvings; }
        {
            // a = sum (-1)^k t2^k / (2k+1)!,  b = sum (-1)^k t2^k / (2k+2)!,  c = sum (-1)^k t2^k / (2k+3)!   (Horner, k = 10 .. 0)
            double sc[33];
#pragma unroll
            for (int k = 0; k < 33; ++k) sc[k] = series[k];
            a = sc[0]; b = sc[1]; c = sc[2];
#pragma unroll
            for (int k = 1; k < 11; ++k) { a = fma(-ts, a, sc[3 * k]); b = fma(-ts, b, sc[3 * k + 1]); c = fma(-ts, c, sc[3 * k + 2]); }
        }
        if (!(t2 < 0.25)) {
            // a rotation of half a radian or more in ONE LM step (never seen; kept exact): sine and cosine of theta / 2^k from the series above,
            // doubled k times, then the closed forms. (libm's sincos here costs 6 KB of code and two dozen scalar constants for a branch that
            // does not run.)
            const double theta = sqrt(t2);
            double sn = a * (theta * h), cs = 1.0 - b * (t2 * (h * h));
            for (; halvings > 0; --halvings) { const double s2 = 2.0 * sn * cs; cs = 1.0 - 2.0 * sn * sn; sn = s2; }
            a = sn / theta; b = (1 - cs) / t2; c = (theta - sn) / (t2 * theta);
        }
        for (int i = 0; i < 9; ++i) {
            if (i % 4 == 0) { R[i] = 1.0 + b * O2[i]; V[i] = 1.0 + c * O2[i]; }
            else { R[i] = a * O[i] + b * O2[i]; V[i] = b * O[i] + c * O2[i]; }
        }
    }
    double qe[4], te[3];
    {   // qfromR with 1 / sqrt in place of sqrt + 0.5 / t
        double tr = R[0] + R[4] + R[8];
        if (tr > 0) {
            const double r = fast_rsqrt(tr + 1.0), t = (tr + 1.0) * r, h = 0.5 * r;
            qe[3] = 0.5 * t;
            qe[0] = (R[7] - R[5]) * h; qe[1] = (R[2] - R[6]) * h; qe[2] = (R[3] - R[1]) * h;
        } else qfromR(R, qe);
    }
    quat_normalize_fast(qe);
    for (int i = 0; i < 3; ++i) te[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    double rt[3]; qrot(qe, T, rt);
    out[0] = te[0] + rt[0]; out[1] = te[1] + rt[1]; out[2] = te[2] + rt[2];
    double qo[4]; qmul(qe, T + 3, qo); quat_normalize_fast(qo);
    out[3] = qo[0]; out[4] = qo[1]; out[5] = qo[2]; out[6] = qo[3];
}

// (H + lambda I) x = b by LDL^T, H given as the 21 entries of its upper triangle; right-looking, the terms of every entry subtracted in
// ascending k as solve6()'s inner products are. Column j before its scaling holds W_ij = L_ij d_j, so the update of entry (i, m) is ONE fused
// multiply-add  A_im -= L_ij W_mj  (solve6(): (L_ij L_mj) d_j in three roundings), a division by a pivot is a multiplication by its reciprocal
// (fast_recip), the substitutions are fused multiply-adds: 110 f64 instructions instead of 210 on the chain between two sweeps, the solution an ulp or
// two from solve6()'s. false = a pivot that is not positive and finite (LinearSolverDense's isPositive()).
__device__ __forceinline__ bool solve6_fast(const double (&Hs)[21], const double (&b)[6], double lambda, double (&x)[6])
{
    double A[6][6];
    {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) A[c][r] = Hs[q++];                           // lower triangle: A[i][j], j <= i
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) A[d][d] += lambda;
    double inv[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double dj = A[j][j];
        if (!(dj > 0.0) || !isfinite(dj)) ok = false;
        inv[j] = fast_recip(dj);
        double W[6];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) { W[i] = A[i][j]; A[i][j] = W[i] * inv[j]; }   // L[i][j]
#pragma unroll
        for (int m = j + 1; m < 6; ++m)
#pragma unroll
            for (int i = m; i < 6; ++i) A[i][m] = fma(-A[i][j], W[m], A[i][m]);
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { double acc = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) acc = fma(-A[i][k], y[k], acc); y[i] = acc; }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] *= inv[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) { double acc = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) acc = fma(-A[k][i], x[k], acc); x[i] = acc; }
    return ok;
}

// a value every lane holds, moved to a scalar register pair (v_readfirstlane): the sweep's wave-uniform operands then cost no vector registers
__device__ __forceinline__ double uniform_f64(double x)
{
    return dbl_of((unsigned)__builtin_amdgcn_readfirstlane(__double2loint(x)), (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x)));
}

// (kExactEdge = false) world -> camera at the pose T for the camera whose rotation matrix and translation sit at rc[0..8], rc[9..11] (LDS: twelve
// loads per pass are cheaper than 24 registers held across the sweep): M = Rc R(T), m = Rc t(T) + tc; an edge's point is then 9 FMAs
__device__ __forceinline__ void pose_compose(const double (&T)[7], const double* rc, double (&M)[12])
{
    double Rc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Rc[i] = rc[i];
    double R[9];
    qtoR(&T[3], R);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = fma(Rc[i * 3], R[j], fma(Rc[i * 3 + 1], R[3 + j], Rc[i * 3 + 2] * R[6 + j]));
        M[9 + i] = fma(Rc[i * 3], T[0], fma(Rc[i * 3 + 1], T[1], fma(Rc[i * 3 + 2], T[2], Rc[9 + i])));
    }
}

// The camera-frame point through the oracle's own operations (cam_point(): two quaternion rotations, 66 f64 instructions instead of the 9 of a composed
// matrix), and the residual's quotients x / z, y / z, the Huber kernel's square root and quotient as values that equal the IEEE results in all but
// vanishingly rare halfway cases, for a third of the IEEE sequences' instructions: v_rcp_f64 / v_rsq_f64 seeds, a correction, ONE remainder step.
// Why (round 6, scratch/pose_flip_stats.py, profiles/r06_pose_flip_stats.txt): an accept / reject decision on the convergence plateau follows the
// rounding of chi2. With per-edge terms that differ from the oracle's at 1e-14 (composed matrix, reciprocal-multiply quotients) 16.8 % of random
// batches ended a round one LM iteration apart; with the oracle's point 12.2 %, with its residual arithmetic as well 9.2 % (the per-edge kernel
// k_pose_opt: 8.2 %; what remains is the order of the sums). Costs 37 us of a 253-us frame (898 edges), 80 of 344 at 1 781: the sweep is bound by
// the f64 issue rate. Hence kExactEdge: the default build keeps the composed transform, option DCS_POSE_EXACT_EDGE selects this arithmetic.
__device__ __forceinline__ void cam_point_regs(const double (&T)[7], const double (&X)[3], const double (&cq)[4], const double (&ct)[3], double (&pc)[3])
{
    double pm[3];
    qrot(&T[3], X, pm);
    pm[0] += T[0]; pm[1] += T[1]; pm[2] += T[2];
    qrot(cq, pm, pc);
    pc[0] += ct[0]; pc[1] += ct[1]; pc[2] += ct[2];
}
__device__ __forceinline__ double quotient(double x, double r /* ~ 1 / z to an ulp */, double z)
{
    const double q0 = x * r;
    return fma(fma(-q0, z, x), r, q0);
}
__device__ __forceinline__ double sqrt_seeded(double x, double rs /* ~ 1 / sqrt(x) to an ulp */)
{
    const double s0 = x * rs;
    return fma(fma(-s0, s0, x), 0.5 * rs, s0);
}

#ifdef DCS_POSE_PROF
// side builds only (-DDCS_POSE_PROF, tools/pose_timeline.py): frame 0 leaves, per wave, the clock at every stage boundary of ONE pass and the kernel's
// span on both clocks (s_memtime = shader clock, s_memrealtime = 100 MHz) in a global array that dcs_debug_pose_prof() copies out -- no printf in
// the kernel: its code and registers would distort what is measured
__device__ unsigned long long g_pose_prof[kPoT / 64][16];
// (the stamps go straight to memory: nine of them held to the kernel's end were 18 scalar registers, spilled, and the spills stretched every stage)
#define DCS_PO_TICK(k) { if (prof_on) { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane_l == 0) g_pose_prof[wave][k] = now_; } }
#else
#define DCS_PO_TICK(k)
#endif

constexpr int kPoseDeclined = INT_MIN;                    // n_inliers of a frame k_pose_opt2 left to k_pose_opt
struct PoseKernargs { PoseArgs a; DCams cams; };          // the kernel's argument list as the kernarg segment lays it out
template <bool kExactEdge>
__global__ __launch_bounds__(kPoT) void k_pose_opt2(PoseArgs a, DCams cams)
{
    __shared__ PoseShared S;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e0 = a.edge_off[f], n = a.edge_cnt ? a.edge_cnt[f] : a.edge_off[f + 1] - e0;
    if (n > a.fast_max) return;                               // k_pose_opt's frame (the host launches it when a frame can be that large)
    for (int k = tid; k < n; k += kPoT) { a.outlier[e0 + k] = 0; if (a.edge_chi2) a.edge_chi2[e0 + k] = 0; }
    if (tid < 4 && a.n_iters) a.n_iters[4 * f + tid] = 0;
    if (n < 3) {                                              // :343-344
        const __attribute__((address_space(4))) char* k0 = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(k0));
        const __attribute__((address_space(4))) PoseArgs& early = *(const __attribute__((address_space(4))) PoseArgs*)k0;
        int f_e = f;
        asm volatile("" : "+s"(f_e));
        if (tid < 7) early.out_poses[7 * f_e + tid] = early.poses[7 * f_e + tid];
        if (tid == 0) early.n_inliers[f_e] = 0;
        return;
    }
    // ---- edges by camera: stable counting sort of the edge indices into S.list (camera c at [S.off[c], S.off[c] + S.ncam[c]))
    {
        int my_pos[kPoChunks];
        for (int i = tid; i < kMaxCams * kPoChunks * kPoW; i += kPoT) (&S.cnt[0][0])[i] = 0;
        for (int i = tid; i < kPoW * kPoLine; i += kPoT) (&S.line[0][0])[i] = 0.0;
        if (tid < 33) S.series[tid] = kPoSeries[tid];
        {   // the rig's cameras: thread i copies word i of the kernel argument straight from the kernarg segment (indexing the by-value argument at run
            // time would copy it to scratch; round 5's "if (tid == i) dst[i] = src[i]" over constant i compiled to 900 KB of exec-mask bookkeeping)
            constexpr int kCamWords = (int)(sizeof(DCams) / sizeof(double));
            static_assert(kCamWords <= kPoT, "one thread per word");
            typedef const __attribute__((address_space(4))) double* KernargWords;
            static_assert(offsetof(PoseKernargs, cams) % sizeof(double) == 0, "word offset");
            const KernargWords src = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PoseKernargs, cams) / sizeof(double);
            double* dst = reinterpret_cast<double*>(&S.cam[0]);
            if (tid < kCamWords) dst[tid] = src[tid];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kPoChunks; ++m) {
            const int k = m * kPoT + tid;
            const int c = k < n ? a.cam[e0 + k] : -1;
            my_pos[m] = -1;
#pragma unroll
            for (int cc = 0; cc < kMaxCams; ++cc) {
                const unsigned long long mk = __ballot(c == cc);
                if (c == cc) my_pos[m] = (cc << 16) | __popcll(mk & ((1ull << lane) - 1ull));
                if (lane == 0) S.cnt[cc][m * kPoW + wave] = __popcll(mk);
            }
            asm volatile("" : "+v"(my_pos[m]));                // (computed HERE: sunk below the table set-up, the 36 ballot masks it is made from stay in scalar registers, and spill)
        }
        // the adjoint map of every camera: H'(i, j) = sum_{k <= m} K[(i, j)][(k, m)] HA(k, m), b'(r) = sum_k adj[k][r] ba(k)
        for (int idx = tid; idx < kMaxCams * 27 * 22; idx += kPoT) {
            const int c = idx / (27 * 22), r = (idx / 22) % 27, t = idx % 22;
            const double* adj = S.cam[c].adj;
            double val = 0.0;
            if (r < 21 && t < 21) {
                int i, j, k, m;
                tri6(r, i, j); tri6(t, k, m);
                val = k == m ? adj[k * 6 + i] * adj[k * 6 + j] : adj[k * 6 + i] * adj[m * 6 + j] + adj[m * 6 + i] * adj[k * 6 + j];
            } else if (r >= 21 && t < 6) val = adj[t * 6 + (r - 21)];
            S.K[c][r][t] = val;
        }
        if constexpr (!kExactEdge) {
            if (tid < kMaxCams) {
                double Rc[9];
                qtoR(S.cam[tid].q, Rc);
                for (int i = 0; i < 9; ++i) S.rc[tid][i] = Rc[i];
                for (int i = 0; i < 3; ++i) S.rc[tid][9 + i] = S.cam[tid].t[i];
            }
        }
        __syncthreads();
        if (tid < kMaxCams) {                                 // exclusive prefix over (chunk, wave) in edge order
            int run = 0;
            for (int i = 0; i < kPoChunks * kPoW; ++i) { const int v = S.cnt[tid][i]; S.cnt[tid][i] = run; run += v; }
            S.ncam[tid] = run;
        }
        __syncthreads();
        if (tid == 0) {
            int off = 0, active = 0;
            for (int c = 0; c < kMaxCams; ++c) { S.off[c] = off; off += S.ncam[c]; S.W[c] = S.ncam[c] > 0 ? 1 : 0; active += S.W[c]; }
            // the remaining waves go, one at a time, to the camera with the most edges per wave (ties: the lower index)
            for (int r = active; r < kPoW; ++r) {
                int best = 0; long long bn = -1, bd = 1;
                for (int c = 0; c < kMaxCams; ++c) if (S.W[c] > 0 && (long long)S.ncam[c] * bd > bn * S.W[c]) { best = c; bn = S.ncam[c]; bd = S.W[c]; }
                ++S.W[best];
            }
            int w = 0;
            for (int c = 0; c < kMaxCams; ++c) for (int u = 0; u < S.W[c]; ++u) { S.wcam[w] = c; S.wu[w] = u; ++w; }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kPoChunks; ++m)
            if (my_pos[m] >= 0) { const int cc = my_pos[m] >> 16; S.list[S.off[cc] + S.cnt[cc][m * kPoW + wave] + (my_pos[m] & 0xffff)] = (uint16_t)(m * kPoT + tid); }
        __syncthreads();
    }
    // ---- this wave's camera and its share of the camera's edges
    const int wc = __builtin_amdgcn_readfirstlane(S.wcam[wave]), wu = __builtin_amdgcn_readfirstlane(S.wu[wave]), wW = __builtin_amdgcn_readfirstlane(S.W[wc]);
    const int nc = __builtin_amdgcn_readfirstlane(S.ncam[wc]), coff = __builtin_amdgcn_readfirstlane(S.off[wc]);
    const int ept = (nc + 64 * wW - 1) / (64 * wW);
    {   // a camera whose share does not fit the registers (only a rig of three or four cameras with more than 768 edges on one of them can do that): the
        // frame is k_pose_opt's, told by the sentinel in n_inliers (the host launches k_pose_opt behind this kernel for such rigs)
        int worst = 0;
        for (int c = 0; c < kMaxCams; ++c) if (S.W[c] > 0) worst = max(worst, (S.ncam[c] + 64 * S.W[c] - 1) / (64 * S.W[c]));
        if (worst > kPoEpt) {
            int tid_d = tid, f_d = f;
            asm volatile("" : "+v"(tid_d), "+s"(f_d));
            if (tid_d == 0) a.n_inliers[f_d] = kPoseDeclined;
            return;
        }
    }
    const double fx = uniform_f64(S.cam[wc].fx), fy = uniform_f64(S.cam[wc].fy), cx = uniform_f64(S.cam[wc].cx), cy = uniform_f64(S.cam[wc].cy);
    double X[kPoEpt][3], c2[kPoEpt];                          // the points and the chi2 of their last evaluation (observation, weight: S.ed)
    unsigned valid = 0, outl = 0;
#pragma unroll
    for (int j = 0; j < kPoEpt; ++j) {
        const int p = (j * wW + wu) * 64 + lane;
        const bool v = j < ept && p < nc;
        const size_t e = (size_t)e0 + (v ? S.list[coff + p] : 0);
        X[j][0] = v ? a.xw[3 * e] : 0.0; X[j][1] = v ? a.xw[3 * e + 1] : 0.0; X[j][2] = v ? a.xw[3 * e + 2] : 1.0;
        S.ed[0][j][tid] = v ? a.obs[2 * e] : 0.0; S.ed[1][j][tid] = v ? a.obs[2 * e + 1] : 0.0; S.ed[2][j][tid] = v ? a.w[e] : 0.0; c2[j] = 0.0;
        valid |= (v ? 1u : 0u) << j;
    }
    const double delta = a.huber, dsqr = uniform_f64(delta * delta);
    const double2* const my_line = reinterpret_cast<const double2*>(S.line[wave]) + (lane < 21 ? 0 : 11);     // lanes 21..26 read b (22..27) and zeros
    const double2* const my_krow = reinterpret_cast<const double2*>(S.K[wc][lane < 27 ? lane : 0]);
    const int line_slot = (lane >> 1) < 21 ? (lane >> 1) : ((lane >> 1) < 27 ? (lane >> 1) + 1 : (lane >> 1) + 17);   // where the sum of v[lane / 2] goes
    bool robust = true;
    int n_bad_edges = 0, buf = 0, n_its = 0;
    double T[7];
#ifdef DCS_POSE_PROF
    int n_pass = 0;
    bool prof_on = false;
    if (lane == 0 && f == 0) { g_pose_prof[wave][9] = __builtin_readcyclecounter(); g_pose_prof[wave][11] = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int it = 0; it < 4; ++it) {
        // (the round's iteration limit and chi2 gate are read from the kernarg segment when the round starts: eight values held from the top
        // of the kernel are eight scalar registers too many)
        const __attribute__((address_space(4))) char* kr = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kr));
        const int its_it = ((const __attribute__((address_space(4))) PoseArgs*)kr)->its[it];
        const float th_it = ((const __attribute__((address_space(4))) PoseArgs*)kr)->chi2_th[it];
        const double* const Tin = ((const __attribute__((address_space(4))) PoseArgs*)kr)->poses + 7 * f;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));                      // (lane == 0 as a mask held since the sort would be a scalar register pair across the loop)
#pragma unroll
        for (int d = 0; d < 7; ++d) T[d] = Tin[d];            // :360 every round restarts from the frame's pose
        // LM state: the same in every lane of every wave
        double lambda = -1, ni = 2, currentChi = 0, iniChi = 0, inv_scale = 1;
        int nBad = 0, n_it = 0, qmax = 0, it_i = 0, phase = 0;
        bool ok2 = true, classify = false;
        for (;;) {
#ifdef DCS_POSE_PROF
            ++n_pass;
            prof_on = f == 0 && n_pass == DCS_POSE_PROF;          // the pass to record: -DDCS_POSE_PROF=<pass number>
            DCS_PO_TICK(8)
#endif
            // ---- one sweep: errors of the active edges at T, robust chi2, and the linearisation there
            double v[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = 0;
            double cq[4], ct[3], M[12];                         // the wave's camera, rig -> camera (vector registers, read per sweep: as scalars they spill)
            if constexpr (kExactEdge) {
#pragma unroll
                for (int d = 0; d < 4; ++d) cq[d] = S.cam[wc].q[d];
#pragma unroll
                for (int d = 0; d < 3; ++d) ct[d] = S.cam[wc].t[d];
            } else pose_compose(T, S.rc[wc], M);
            int ept_l = ept;
            unsigned live_l = classify ? outl : valid & ~outl, valid_l = valid, bad = 0;      // (classifying: only the previous outliers are evaluated)
            asm volatile("" : "+s"(ept_l), "+v"(live_l), "+v"(valid_l));   // (tested afresh per slot: hoisted out of the loop the slots' comparisons and masks hold 28 scalar registers)
#pragma unroll
            for (int j = 0; j < kPoEpt; ++j) {
                if (j >= ept_l) continue;                       // wave-uniform (no early exit: a 16-fold unrolled loop with 16 exits is not unrolled at all)
                const bool act = (live_l >> j) & 1u;
                double x, y, z;
                if constexpr (kExactEdge) {
                    double pcv[3];
                    cam_point_regs(T, X[j], cq, ct, pcv);
                    x = pcv[0]; y = pcv[1]; z = pcv[2];
                } else {
                    x = fma(M[0], X[j][0], fma(M[1], X[j][1], fma(M[2], X[j][2], M[9])));
                    y = fma(M[3], X[j][0], fma(M[4], X[j][1], fma(M[5], X[j][2], M[10])));
                    z = fma(M[6], X[j][0], fma(M[7], X[j][1], fma(M[8], X[j][2], M[11])));
                }
                if (!act) { x = 0; y = 0; z = 1; }              // an excluded edge contributes exact zeros below, whatever its point does
                const double iz = fast_recip(z);                // (v_rcp_f64 + a third-order correction: an ulp from the IEEE reciprocal, a quarter of its instructions)
                double xz, yz, ex, ey;
                if constexpr (kExactEdge) {
                    xz = quotient(x, iz, z); yz = quotient(y, iz, z);
                    ex = S.ed[0][j][tid] - (xz * fx + cx); ey = S.ed[1][j][tid] - (yz * fy + cy);     // (unfused: the oracle's obs - (x / z * fx + cx))
                } else {
                    xz = x * iz; yz = y * iz;
                    ex = S.ed[0][j][tid] - fma(xz, fx, cx); ey = S.ed[1][j][tid] - fma(yz, fy, cy);
                }
                const double w = act ? S.ed[2][j][tid] : 0.0;
                const double x2 = ex * (w * ex) + ey * (w * ey);
                if (act) c2[j] = x2;
                // (the classifying pass runs on through the accumulation, whose sums nobody reads: a skip here is a second way into the next slot,
                // and the compiler then copies all 28 accumulators at the end of every slot)
                if (classify && ((valid_l >> j) & 1u) && (float)c2[j] > th_it) bad |= 1u << j;
                const bool big = robust && x2 > dsqr;
                double rho0 = x2, we = w;
                if (__builtin_amdgcn_ballot_w64(big)) {        // wave-uniform: only when some edge of the wave is beyond the Huber width
                    const double rs = fast_rsqrt(big ? x2 : 1.0);
                    if constexpr (kExactEdge) {
                        const double sq = sqrt_seeded(big ? x2 : 1.0, rs);
                        if (big) { rho0 = 2 * sq * delta - dsqr; we = quotient(delta, rs, sq) * w; }     // (delta / sqrt(x2)) * w
                    } else if (big) { rho0 = 2 * (x2 * rs) * delta - dsqr; we = (delta * rs) * w; }       // sqrt(x2) = x2 rs, delta / sqrt(x2) = delta rs
                }
                v[27] += rho0;
                v[28] += act ? 1.0 : 0.0;
                const double r0 = -(we * ex), r1 = -(we * ey);
                // A = -(1/z) [fx 0 -x/z fx; 0 fy -y/z fy] [-[p]x | I]  (types_six_dof_expmap.cpp:218-246 before the adjoint)
                const double st0 = -(fx * iz), st2 = (fx * xz) * iz, st4 = -(fy * iz), st5 = (fy * yz) * iz;
                const double a00 = st2 * y, a01 = fma(st0, z, -(st2 * x)), a02 = -(st0 * y), a03 = st0, a05 = st2;
                const double a10 = fma(st5, y, -(st4 * z)), a11 = -(st5 * x), a12 = st4 * x, a14 = st4, a15 = st5;
                const double p00 = we * a00, p01 = we * a01, p02 = we * a02, p03 = we * a03, p05 = we * a05;
                const double p10 = we * a10, p11 = we * a11, p12 = we * a12, p14 = we * a14, p15 = we * a15;
                v[0] = fma(p00, a00, fma(p10, a10, v[0]));   v[1] = fma(p00, a01, fma(p10, a11, v[1]));   v[2] = fma(p00, a02, fma(p10, a12, v[2]));
                v[3] = fma(p00, a03, v[3]);                  v[4] = fma(p10, a14, v[4]);                  v[5] = fma(p00, a05, fma(p10, a15, v[5]));
                v[6] = fma(p01, a01, fma(p11, a11, v[6]));   v[7] = fma(p01, a02, fma(p11, a12, v[7]));   v[8] = fma(p01, a03, v[8]);
                v[9] = fma(p11, a14, v[9]);                  v[10] = fma(p01, a05, fma(p11, a15, v[10]));
                v[11] = fma(p02, a02, fma(p12, a12, v[11])); v[12] = fma(p02, a03, v[12]);                v[13] = fma(p12, a14, v[13]);
                v[14] = fma(p02, a05, fma(p12, a15, v[14]));
                v[15] = fma(p03, a03, v[15]);                v[17] = fma(p03, a05, v[17]);
                v[18] = fma(p14, a14, v[18]);                v[19] = fma(p14, a15, v[19]);
                v[20] = fma(p05, a05, fma(p15, a15, v[20]));
                v[21] = fma(a00, r0, fma(a10, r1, v[21]));   v[22] = fma(a01, r0, fma(a11, r1, v[22]));   v[23] = fma(a02, r0, fma(a12, r1, v[23]));
                v[24] = fma(a03, r0, v[24]);                 v[25] = fma(a14, r1, v[25]);                 v[26] = fma(a05, r0, fma(a15, r1, v[26]));
            }
            if (classify) {
                outl = bad;
                int cnt = __popc(bad);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
                if (lane_l == 0) S.bad[it & 1][wave] = cnt;
                break;
            }
            DCS_PO_TICK(0)
            // ---- the wave's 29 sums, then adj^T ( . ) adj on them: lane q < 21 forms entry q of H, lanes 21..26 the entries of b
            {
                const double2 k0 = my_krow[0], k1 = my_krow[1], k2 = my_krow[2], k3 = my_krow[3], k4 = my_krow[4], k5 = my_krow[5], k6 = my_krow[6], k7 = my_krow[7], k8 = my_krow[8],
                              k9 = my_krow[9], k10 = my_krow[10];        // this lane's row of the adjoint map: asked for now, needed after the reduction
                const double sum = wave_sum32(v, lane);         // lanes 2q, 2q + 1: total of v[q]
                if (!(lane & 1) && lane < 58) S.line[wave][line_slot] = sum;
                DCS_PO_TICK(1)
                wave_sync();
                const double2 d0 = my_line[0], d1 = my_line[1], d2 = my_line[2], d3 = my_line[3], d4 = my_line[4], d5 = my_line[5], d6 = my_line[6], d7 = my_line[7], d8 = my_line[8],
                              d9 = my_line[9], d10 = my_line[10];
                const double pass_through = S.line[wave][lane == 28 ? 45 : 44];
                double c0 = k0.x * d0.x, c1 = k0.y * d0.y, c2 = k1.x * d1.x, c3 = k1.y * d1.y;
                c0 = fma(k2.x, d2.x, c0); c1 = fma(k2.y, d2.y, c1); c2 = fma(k3.x, d3.x, c2); c3 = fma(k3.y, d3.y, c3);
                c0 = fma(k4.x, d4.x, c0); c1 = fma(k4.y, d4.y, c1); c2 = fma(k5.x, d5.x, c2); c3 = fma(k5.y, d5.y, c3);
                c0 = fma(k6.x, d6.x, c0); c1 = fma(k6.y, d6.y, c1); c2 = fma(k7.x, d7.x, c2); c3 = fma(k7.y, d7.y, c3);
                c0 = fma(k8.x, d8.x, c0); c1 = fma(k8.y, d8.y, c1); c2 = fma(k9.x, d9.x, c2); c3 = fma(k9.y, d9.y, c3);
                c0 = fma(k10.x, d10.x, c0); c1 = fma(k10.y, d10.y, c1);
                const double mine = lane < 27 ? (c0 + c1) + (c2 + c3) : pass_through;
                if (lane < 29) S.part[buf][wave][lane] = mine;
            }
            DCS_PO_TICK(2)
            __syncthreads();                                    // the one barrier of a pass
            DCS_PO_TICK(3)
            double Hn[21], bn[6], chi, cnt;
            {
                double t = S.part[buf][0][lane & 31];
#pragma unroll
                for (int w = 1; w < kPoW; ++w) t += S.part[buf][w][lane & 31];          // wave order
                buf ^= 1;
                if (lane < 32) S.tot[wave][lane] = t;
                wave_sync();
                const double2* src = reinterpret_cast<const double2*>(S.tot[wave]);
                double2 u[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) u[i] = src[i];
#pragma unroll
                for (int i = 0; i < 21; ++i) Hn[i] = (i & 1) ? u[i >> 1].y : u[i >> 1].x;
#pragma unroll
                for (int i = 0; i < 6; ++i) bn[i] = ((21 + i) & 1) ? u[(21 + i) >> 1].y : u[(21 + i) >> 1].x;
                chi = u[13].y; cnt = u[14].x;
            }
            DCS_PO_TICK(4)
            // ---- the LM rule (optimization_algorithm_levenberg.cpp:61-164 as Optimizer.cc:360-364 drives it). The floating-point side is
            // branch-free (both outcomes, then selects); its five comparisons become bits of a SCALAR register, so that the counters and the
            // control flow below live on the scalar unit: every lane holds the same values, but only this tells the compiler
            const double tempChi = ok2 ? chi : 1.7976931348623157e308;
#ifdef DCS_PO_EXACT_RULE
            const double rho = (currentChi - tempChi) / (1.0 / inv_scale);
#else
            const double rho = (currentChi - tempChi) * inv_scale;   // (1 / scale: divided when the trial was made, off this chain)
#endif
            const bool acc_v = rho > 0 && isfinite(tempChi);
            const double t3 = 2 * rho - 1;
            const double alpha = fmin(1. - t3 * t3 * t3, 2. / 3.);      // g2o: pow(2 rho - 1, 3)
            const double lam_acc = lambda * fmax(1. / 3., alpha), lam_rej = lambda * ni;
            const double chi_after = acc_v ? tempChi : currentChi;
            const int bits = __builtin_amdgcn_readfirstlane((acc_v ? 1 : 0) | (rho < 0 ? 2 : 0) | (rho == 0 ? 4 : 0) | ((iniChi - chi_after) * 1e3 < iniChi ? 8 : 0) | (cnt == 0.0 ? 16 : 0));
            int ctl = 0, adopt = 0, solve = 0;
            if (phase == 0) {                                   // evaluation at the current pose (round start, or after a rejected trial that did not end the round)
                if ((it_i == 0 && (bits & 16)) || it_i >= its_it) ctl = 2;                // no active edge / no iterations: optimize() does nothing
                else { currentChi = chi; iniChi = chi; adopt = 1; solve = 1; qmax = 0; }
            } else {
                const bool accepted = bits & 1;
                if (accepted) { lambda = lam_acc; ni = 2; currentChi = tempChi; }
                else {
                    lambda = lam_rej; ni *= 2;
#pragma unroll
                    for (int d = 0; d < 7; ++d) T[d] = S.keep[wave][28 + d];               // pop (the edges keep the rejected trial's errors, as in g2o)
                }
                ++qmax;
                if ((bits & 2) && qmax < 10) solve = 1;                                   // next trial of this iteration: same system, larger lambda
                else {
                    ++n_it;
                    if (qmax == 10 || (bits & 4)) ctl = 2;
                    else { if (bits & 8) ++nBad; else nBad = 0; if (nBad >= 3) ctl = 2; }
                    ++it_i;
                    if (it_i >= its_it) ctl = 2;
                    if (ctl != 2) {
                        if (accepted) { adopt = 1; solve = 1; iniChi = currentChi; qmax = 0; }   // the trial's sweep IS computeActiveErrors + the next linearisation
                        else phase = 0;                                                   // (rho is NaN) re-evaluate at the restored pose
                    }
                }
            }
            if (solve && it_i == 0 && phase == 0) {             // computeLambdaInit on the system about to be adopted
                double md = 0;
                md = fmax(fabs(Hn[0]), md); md = fmax(fabs(Hn[6]), md); md = fmax(fabs(Hn[11]), md);
                md = fmax(fabs(Hn[15]), md); md = fmax(fabs(Hn[18]), md); md = fmax(fabs(Hn[20]), md);
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            const int flags = ctl | (adopt << 2) | (solve << 3);
            DCS_PO_TICK(5)
            if (flags & 8) {
                if (flags & 4) {                                // adopt: the wave keeps its copy for the re-solves after rejected trials
                    if (lane < 27) S.keep[wave][lane] = S.tot[wave][lane];
                } else {
                    wave_sync();
#pragma unroll
                    for (int i = 0; i < 21; ++i) Hn[i] = S.keep[wave][i];
#pragma unroll
                    for (int i = 0; i < 6; ++i) bn[i] = S.keep[wave][21 + i];
                }
                if (lane_l == 0) {
#pragma unroll
                    for (int d = 0; d < 7; ++d) S.keep[wave][28 + d] = T[d];              // push
                }
                double xs[6];
#ifdef DCS_PO_EXACT_SOLVE
                {
                    double Hf[36];
                    int q = 0;
                    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { Hf[r * 6 + c] = Hn[q]; Hf[c * 6 + r] = Hn[q]; ++q; }
                    ok2 = __builtin_amdgcn_readfirstlane(solve6(Hf, lambda, bn, xs) ? 1 : 0) != 0;
                }
#else
                ok2 = __builtin_amdgcn_readfirstlane(solve6_fast(Hn, bn, lambda, xs) ? 1 : 0) != 0;
#endif
                if (!ok2) {
#pragma unroll
                    for (int d = 0; d < 6; ++d) xs[d] = 0;
                }
                double scale = 0;
#pragma unroll
                for (int j = 0; j < 6; ++j) scale += xs[j] * (lambda * xs[j] + bn[j]);
                scale += 1e-3;
                inv_scale = 1.0 / scale;
                DCS_PO_TICK(6)
                double o[7];
#ifdef DCS_PO_EXACT_EXP
                pose_oplus(T, xs, o);
#else
                pose_oplus_fast(T, xs, o, S.series);
#endif
#pragma unroll
                for (int d = 0; d < 7; ++d) T[d] = o[d];
                phase = 1;
            }
            DCS_PO_TICK(7)
            // ---- the round is over: classification of every edge (:365-390) is one more pass of the sweep's evaluation at the final pose, in which
            // the previous outliers are re-evaluated and the inliers keep the chi2 of their last evaluation (a second copy of the evaluation for
            // it would be another 17 KB of code)
            if ((flags & 3) == 2) classify = true;
        }
        n_its = it == 0 ? n_it : n_its | (n_it << (8 * it));                          // (<= 10 each: a byte per round)
        __syncthreads();
        n_bad_edges = 0;
#pragma unroll
        for (int w = 0; w < kPoW; ++w) n_bad_edges += S.bad[it & 1][w];
        if (it == 2) robust = false;                            // :388-389
        if (n < 10) break;                                      // :392
    }
#ifdef DCS_POSE_PROF
    if (lane == 0 && f == 0) {
        g_pose_prof[wave][10] = __builtin_readcyclecounter();
        g_pose_prof[wave][12] = __builtin_amdgcn_s_memrealtime();
        g_pose_prof[wave][13] = (unsigned long long)n_pass; g_pose_prof[wave][14] = (unsigned long long)n; g_pose_prof[wave][15] = (unsigned long long)ept;
    }
#endif
    // the output pointers are read from the kernarg segment again here (through a pointer the compiler cannot connect to the argument list): held
    // from the top of the kernel they would sit in scalar registers across the whole trial loop, and spill
    const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const __attribute__((address_space(4))) PoseArgs& late = *(const __attribute__((address_space(4))) PoseArgs*)ka;
    int tid_l = tid, f_l = f, wave_l = wave;
    asm volatile("" : "+v"(tid_l), "+s"(f_l), "+s"(wave_l));
    if (tid_l == 0) {
        double* out_poses = late.out_poses;
#pragma unroll
        for (int d = 0; d < 7; ++d) out_poses[7 * f_l + d] = T[d];
        late.n_inliers[f_l] = n - n_bad_edges;
        int32_t* n_iters = late.n_iters;
        if (n_iters) { n_iters[4 * f_l] = n_its & 255; n_iters[4 * f_l + 1] = (n_its >> 8) & 255; n_iters[4 * f_l + 2] = (n_its >> 16) & 255; n_iters[4 * f_l + 3] = (n_its >> 24) & 255; }
    }
    uint8_t* const outlier = late.outlier;
    double* const edge_chi2 = late.edge_chi2;
    {   // (the wave's share again from LDS and the masks from their bits, for the same reason)
        const int wc2 = __builtin_amdgcn_readfirstlane(S.wcam[wave_l]), wu2 = __builtin_amdgcn_readfirstlane(S.wu[wave_l]), wW2 = __builtin_amdgcn_readfirstlane(S.W[wc2]);
        const int nc2 = __builtin_amdgcn_readfirstlane(S.ncam[wc2]), coff2 = __builtin_amdgcn_readfirstlane(S.off[wc2]);
        const int ept2 = (nc2 + 64 * wW2 - 1) / (64 * wW2);
        const int e0_2 = late.edge_off[f_l];
        unsigned valid_l = valid, outl_l = outl;
        asm volatile("" : "+v"(valid_l), "+v"(outl_l));
#pragma unroll
        for (int j = 0; j < kPoEpt; ++j) {
            if (j >= ept2) continue;
            if ((valid_l >> j) & 1u) {
                const size_t e = (size_t)e0_2 + S.list[coff2 + (j * wW2 + wu2) * 64 + lane];
                outlier[e] = (outl_l >> j) & 1u;
                if (edge_chi2) edge_chi2[e] = c2[j];
            }
        }
    }
}

}  // namespace dcs

using namespace dcs;

namespace {

// CU range of the solver's streams: process-wide, set by dcs_ba_set_cu_range or DCS_BA_CUS="first:count" (read once)
static void ba_cu_range(int& first, int& count, bool set)
{
    static std::mutex mu;
    static int s_first = 0, s_count = -1;
    std::lock_guard<std::mutex> lk(mu);
    if (set) { s_first = first; s_count = count; return; }
    if (s_count < 0) {
        s_first = 0; s_count = 0;
    }
    first = s_first; count = s_count;
}

// streams of the front end the solver's compute streams keep off (dcs_ba_avoid_streams): process-wide
static void ba_avoid_list(std::vector<hipStream_t>& v, bool set)
{
    static std::mutex mu;
    static std::vector<hipStream_t> s_list;
    std::lock_guard<std::mutex> lk(mu);
    if (set) s_list = v; else v = s_list;
}

// Per host thread and device, kept across solves: the device arena (grow-only), the pinned staging image of its
// upload / download regions, the pinned progress + stop words and the stream. hipMalloc + hipFree of ~30 MB,
// hipHostMalloc/Free and stream create/destroy cost ~1.4 ms per call together -- a third of a local BA.
struct BaContext {
    char* base = nullptr; size_t cap = 0; int device = -1;
    char* h_stage = nullptr; size_t stage_cap = 0;
    int* h_words = nullptr; size_t words_cap = 0;      // [0] last finished step, [1] problems done, [16 + b] stop word of problem b
    hipStream_t stream = nullptr;
    // a batch is split into groups of problems, each fed through its own stream: while one group's reduced camera systems are
    // factored (one workgroup per problem) the other groups' wide kernels have the rest of the chip
    static constexpr int kMaxGroups = 8;
    hipStream_t aux[kMaxGroups - 1] = {nullptr, nullptr, nullptr};
    hipStream_t dl = nullptr;                          // results come down on their own stream once every problem has reported done
    // A call that took its results down on `dl` returns while the steps it had queued ahead (no-ops on a finished batch) are still in
    // the solver's queues -- and the last block of each of their k_begin / k_error launches still stores to the pinned progress words. Nothing
    // may reset those words, or free / reuse what the queues reference, before they have drained: drain() is the first thing prepare()
    // and release() do (by then the tail has long run under the caller's own work; found as a bug of the first version of the dl path:
    // the next call zeroed the words on the host, a late progress store of the previous call reported "all problems done", and the next call
    // would have downloaded results before running a step).
    bool tail_pending = false;
    void drain()
    {
        if (!tail_pending) return;
#ifdef DCS_BA_NO_DRAIN                                  // side builds only: shows that tests/test_gpu_ba.py catches the missing drain
        tail_pending = false;
        return;
#endif
        if (stream) (void)hipStreamSynchronize(stream);
        for (hipStream_t a : aux) if (a) (void)hipStreamSynchronize(a);
        if (dl) (void)hipStreamSynchronize(dl);             // the pair-list kernels of a call that ended before its first k_schur
        tail_pending = false;
    }
    hipEvent_t ev_up = nullptr, ev_done[kMaxGroups - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_staged = nullptr, ev_pairs = nullptr;   // pair lists on the side stream: upload done -> lists done
    // one LM step of a group = 8 dependent launches with the same arguments every time: optionally replayed as an executable hipGraph
    // (DCS_BA_GRAPH=1, see dcs_ba_local_batch)
    struct StepGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; std::vector<hipGraphNode_t> nodes; unsigned shape = 0; };
    StepGraph step_graph[kMaxGroups];
    void drop_graph(StepGraph& g)
    {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        g.exec = nullptr; g.graph = nullptr; g.nodes.clear(); g.shape = 0;
    }
    // measurement hook (dcs_ba_timing): hipEvents around every LDL^T launch and every step of the device loop
    bool timing = false;
    std::vector<hipEvent_t> events;
    double t_ldlt_us = 0, n_ldlt = 0, t_step_us = 0, n_step = 0;
    void release()
    {
        drain();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        if (base) (void)hipFree(base);
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_words) (void)hipHostFree(h_words);
        if (stream) (void)hipStreamDestroy(stream);
        for (StepGraph& g : step_graph) drop_graph(g);
        for (hipStream_t& a : aux) { if (a) (void)hipStreamDestroy(a); a = nullptr; }
        if (dl) (void)hipStreamDestroy(dl);
        dl = nullptr;
        for (hipEvent_t& e : ev_done) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (ev_up) (void)hipEventDestroy(ev_up);
        ev_up = nullptr;
        if (ev_staged) (void)hipEventDestroy(ev_staged);
        if (ev_pairs) (void)hipEventDestroy(ev_pairs);
        ev_staged = ev_pairs = nullptr;
        base = nullptr; cap = 0; h_stage = nullptr; stage_cap = 0; h_words = nullptr; words_cap = 0; stream = nullptr; device = -1;
    }
    ~BaContext() { release(); }
    // the pinned staging image alone, before the lists (and so the arena's size) exist: the caller's arrays are copied into it while the lists are built
    int prepare_stage(size_t stage_bytes)
    {
        drain();
        int dev = 0;
        DCS_HIP(hipGetDevice(&dev));
        if (device != dev) release();
        device = dev;
        if (h_stage && stage_cap < stage_bytes) { (void)hipHostFree(h_stage); h_stage = nullptr; stage_cap = 0; }
        if (!h_stage) { DCS_HIP(hipHostMalloc((void**)&h_stage, stage_bytes, hipHostMallocDefault)); stage_cap = stage_bytes; }
        return DCS_OK;
    }
    int prepare(size_t arena_bytes, size_t stage_bytes, size_t n_words)
    {
        drain();                                            // queued-ahead steps of the previous call (see tail_pending)
        int dev = 0;
        DCS_HIP(hipGetDevice(&dev));
        if (device != dev) release();
        device = dev;
        if (base && cap < arena_bytes) { (void)hipFree(base); base = nullptr; cap = 0; }
        if (!base) { DCS_HIP(hipMalloc((void**)&base, arena_bytes)); cap = arena_bytes; }
        if (h_stage && stage_cap < stage_bytes) { (void)hipHostFree(h_stage); h_stage = nullptr; stage_cap = 0; }
        if (!h_stage && stage_bytes) { DCS_HIP(hipHostMalloc((void**)&h_stage, stage_bytes, hipHostMallocDefault)); stage_cap = stage_bytes; }
        if (h_words && words_cap < n_words) { (void)hipHostFree(h_words); h_words = nullptr; words_cap = 0; }
        if (!h_words) {
            const size_t w = std::max<size_t>(n_words, 64);
            DCS_HIP(hipHostMalloc((void**)&h_words, w * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
            words_cap = w;
        }
        if (!stream) { const int rc_s = create_stream(&stream); if (rc_s) return rc_s; }
        return DCS_OK;
    }
    // Default priority. DCS_BA_STREAM_PRIORITY=1 (measurement aid) creates the solver's streams with the highest priority: next to a
    // front end that fills the chip (config C5) that HALVES the extraction rate (85 -> 41 kfeatures/s) for +20 % BA iterations --
    // the short, gap-ridden kernels of the solver keep preempting the dispatch of the front end's waves.
    // The streams of one context run CONCURRENTLY (groups of a batch, the result download over the queued-ahead steps), and so do the
    // solver and the front end of config C5: each new stream is created on a hardware queue of its own -- apart from the context's
    // other streams and from the streams named with dcs_ba_avoid_streams (common.cpp: create_stream_apart).
    int create_stream(hipStream_t* s)
    {
        int first = 0, count = 0;
        ba_cu_range(first, count, false);
        hipError_t e = hipSuccess;
        if (count > 0) e = create_cu_range_stream(s, first, count);      // config C5: the solver keeps its own compute units (dcs_ba_set_cu_range)
        else {
            {
                // first choice: a queue shared with nobody; when the process has too few (three besides the legacy default stream's), a queue
                // shared with one of the context's OWN streams -- never the front end's: a download or a group's kernels queued behind an
                // extraction that the Tracking thread enqueues steps ahead would wait for all of it
                std::vector<hipStream_t> front, all;
                ba_avoid_list(front, false);
                all = front;
                if (stream) all.push_back(stream);
                for (hipStream_t a : aux) if (a) all.push_back(a);
                bool apart = false;
                int rc_a = create_stream_apart(s, all.data(), (int)all.size(), &apart);
                if (rc_a || apart || front.empty() || front.size() == all.size()) return rc_a;
                (void)hipStreamDestroy(*s);
                *s = nullptr;
                return create_stream_apart(s, front.data(), (int)front.size(), nullptr);
            }
        }
        if (e != hipSuccess) { set_error("solver stream: %s", hipGetErrorString(e)); (void)hipGetLastError(); return DCS_ERR_HIP; }
        return DCS_OK;
    }
    int prepare_groups(int n_groups)
    {
        if (n_groups > 1 && !ev_up) DCS_HIP(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
        for (int g = 1; g < n_groups; ++g) {
            if (!aux[g - 1]) { const int rc_s = create_stream(&aux[g - 1]); if (rc_s) return rc_s; }
            if (!ev_done[g - 1]) DCS_HIP(hipEventCreateWithFlags(&ev_done[g - 1], hipEventDisableTiming));
        }
        return DCS_OK;
    }
};
inline BaContext& ba_context() { static thread_local BaContext c; return c; }

// carves 256-byte aligned arrays out of [base, ...): run once with a null base to size the arena, once for real
struct Carver {
    char* base = nullptr; size_t off = 0;
    template <typename T> T* get(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += std::max<size_t>(n, 1) * sizeof(T);
        return p;
    }
};

// the Schur launch by group size: k_schur<28> (16 waves per pose pair: the shortest chains) for 1-2 problems of C4 size, k_schur_w (2 waves per pair: a batch
// fits the chip in one round) beyond; DCS_BA_SCHUR_WAVE = 0 / 2 forces one of them (same bits)
static bool schur_use_wave(int nb, int pairs_per_problem)
{
    const int m = (int)opt(OPT_BA_SCHUR_WAVE);
    // (round 6: also for ONE problem with many pose pairs -- 60 free poses = 1 830 pairs +3 %, the 200-KF map = 19 900 pairs +9 %)
    return m == 2 || (m == 1 && (nb > 2 || (long long)nb * pairs_per_problem >= 1800));
}

// A few persistent host threads for the per-problem list building of a batch. One job at a time: a second caller (the header
// promises re-entrancy) that finds the pool busy simply does its own work inline. The workers are never joined (they sleep on a
// condition variable until the process ends), so there is no destruction-order problem at exit.
class HostPool {
public:
    template <class F> void run(int n, F&& f)
    {
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock() || n < 2) { for (int i = 0; i < n; ++i) f(i); return; }
        std::function<void(int)> fn = std::ref(f);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_ = 0; left_ = n; ++gen_;
            while ((int)threads_.size() < std::min(n - 1, 7)) { threads_.emplace_back([this] { loop(); }); threads_.back().detach(); }
        }
        cv_.notify_all();
        for (;;) {                                  // the caller works too
            int i;
            { std::lock_guard<std::mutex> lk(mu_); if (next_ >= n_) break; i = next_++; }
            fn(i);
            std::lock_guard<std::mutex> lk(mu_); --left_;
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return left_ == 0; });
        fn_ = nullptr;
    }
private:
    void loop()
    {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return gen_ != seen && fn_ && next_ < n_; });
            seen = gen_;
            while (fn_ && next_ < n_) {
                const int i = next_++;
                lk.unlock(); (*fn_)(i); lk.lock();
                if (--left_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, next_ = 0, left_ = 0;
    unsigned gen_ = 0;
};
static HostPool& host_pool() { static HostPool* p = new HostPool; return *p; }

struct Round {                                  // structure of one optimisation round (buildIndexMapping + buildStructure)
    std::vector<int32_t> pose_idx, pt_off, pt_edges, pt_pi, ps_off, ps_edges, pair_ij;
    std::vector<int32_t> cur_pt, cur_ps;
    std::vector<uint8_t> pose_act;
    int np = 0, n = 0, n_pad = 0, n_pairs = 0, n_active = 0;
    size_t n_pair_entries = 0;                      // total length of the pose-pair lists (the device builds them: k_pairs_*)
};

// returns -1, or the id of an edge that repeats a (pose, point) pair (the pair lists assume at most one, like g2o's hash of Hpl blocks)
//
// Host cost matters: the lists are rebuilt on every call (the local map changes between calls) and were 0.26 ms of a 1.95 ms C4 solve.
// The host keeps the two counting sorts over the edges and the per-point ordering + duplicate test; the pose-pair lists of k_schur
// (~L * obs^2 / 2 entries) are built on the device from the uploaded edges (k_pairs_mark / _scan / _fill) -- the host only sizes them.
int build_round(const dcs_ba_problem* pb, Round& r)
{
    const int P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
    const int32_t* __restrict__ e_pose = pb->edge_pose;
    const int32_t* __restrict__ e_point = pb->edge_point;
    r.n_active = E;                                           // round 0 structure: every edge (the second round only masks edges)
    // ONE pass over the edges for the range check, the edges per point and the edges per pose (by pose id: which poses are free is only known
    // once every edge has named its pose); then the free poses' counts are packed. (Round 6: the check, the activity flags and the two counts
    // were two passes, and the sorted runs went through a scratch array: 0.086 of a 1.67-ms C4 solve on the host before the first upload.)
    r.pt_off.assign(L + 1, 0);
    std::vector<int32_t>& pose_cnt = r.cur_ps;                // (scratch: P counts now, the free poses' write cursors below)
    pose_cnt.assign((size_t)P + 1, 0);
    {
        const int32_t* __restrict__ e_cam = pb->edge_cam;
        const int n_cams = pb->n_cams;
        int32_t* __restrict__ pt_cnt = r.pt_off.data() + 1;
        int32_t* __restrict__ ps_cnt = pose_cnt.data();
        for (int e = 0; e < E; ++e) {
            const int ps = e_pose[e], pt = e_point[e];
            if ((unsigned)ps >= (unsigned)P || (unsigned)pt >= (unsigned)L || (unsigned)e_cam[e] >= (unsigned)n_cams) return -2 - e;
            ++pt_cnt[pt];
            ++ps_cnt[ps];
        }
    }
    r.pose_idx.assign(P, -1);
    r.pose_act.assign(P, 0);
    r.np = 0;
    r.ps_off.assign(1, 0);
    for (int p = 0; p < P; ++p) {
        r.pose_act[p] = pose_cnt[p] > 0;
        if (pose_cnt[p] > 0 && !pb->pose_fixed[p]) { r.pose_idx[p] = r.np++; r.ps_off.push_back(r.ps_off.back() + pose_cnt[p]); }
    }
    const int np = r.np;
    const int32_t* __restrict__ pose_idx = r.pose_idx.data();
    r.n = np * 6; r.n_pad = ((r.n + kNB - 1) / kNB) * kNB;
    for (int l = 0; l < L; ++l) r.pt_off[l + 1] += r.pt_off[l];
    r.pt_edges.resize(E); r.pt_pi.resize(E);
    r.ps_edges.resize(r.ps_off[np]);
    // A point's run is ordered by (free-pose index, edge id), fixed poses (index -1) first: the order the kernels sum in. No sort: the edges of
    // fixed poses go into the runs first, in edge order, while the free poses' edges are bucketed by pose (ps_edges: edge order inside a pose);
    // then the buckets are emptied into the runs pose by pose. A repeated (pose, point) pair shows up as two neighbours with the same index
    // (free poses) or, in the short fixed prefix, by comparing pose ids pairwise.
    size_t entries = 0;
    {
        std::vector<int32_t>& cur_pt = r.cur_pt;
        std::vector<int32_t>& cur_ps = r.cur_ps;
        cur_pt.assign(r.pt_off.begin(), r.pt_off.end() - 1); cur_ps.assign(r.ps_off.begin(), r.ps_off.end() - 1);
        int32_t* __restrict__ pe = r.pt_edges.data();
        int32_t* __restrict__ pp = r.pt_pi.data();
        int32_t* __restrict__ ps_edges = r.ps_edges.data();
        int32_t* __restrict__ cpt = cur_pt.data();
        for (int e = 0; e < E; ++e) {
            const int pi = pose_idx[e_pose[e]];
            if (pi >= 0) ps_edges[cur_ps[pi]++] = e;
            else { const int k = cpt[e_point[e]]++; pe[k] = e; pp[k] = -1; }
        }
        for (int l = 0; l < L; ++l) {                         // the fixed prefix of every run: duplicates by pose id
            const int k0 = r.pt_off[l], nf = cpt[l] - k0;
            for (int a = 1; a < nf; ++a)
                for (int b2 = 0; b2 < a; ++b2) if (e_pose[pe[k0 + a]] == e_pose[pe[k0 + b2]]) return pe[k0 + a];
            const size_t f = (size_t)(r.pt_off[l + 1] - cpt[l]);
            entries += f * (f + 1) / 2;
        }
        for (int pi = 0; pi < np; ++pi)
            for (int q = r.ps_off[pi], q1 = r.ps_off[pi + 1]; q < q1; ++q) {
                const int e = ps_edges[q], l = e_point[e], k = cpt[l]++;
                if (k > r.pt_off[l] && pp[k - 1] == pi) return e;                  // the same free pose twice on this point
                pe[k] = e; pp[k] = pi;
            }
    }
    r.n_pair_entries = entries;
    // EVERY pose pair (i1 <= i2) gets a workgroup of k_schur; one without a common point finds an empty list and leaves its block of S
    // zero. The DIAGONAL pairs come first: a pose's own list (every edge of the pose) is several times longer than any list it shares
    // with another pose, k_schur runs one workgroup per pair in list order, and a long workgroup that starts in the second round of
    // the chip's wave slots is the kernel's tail. (The order of the pairs decides nothing else: every pair writes its own block of S.)
    r.n_pairs = np * (np + 1) / 2;
    r.pair_ij.resize((size_t)2 * r.n_pairs);
    {
        int32_t* ij = r.pair_ij.data();
        for (int i = 0; i < np; ++i) { *ij++ = i; *ij++ = i; }
        for (int i1 = 0; i1 < np; ++i1)
            for (int i2 = i1 + 1; i2 < np; ++i2) { *ij++ = i1; *ij++ = i2; }
    }
    return -1;
}

// k_pose_opt on arrays that already live in HBM (the tracking chain of proj_kernels.hip): declared in track.h
static DCams pose_cams_of(const dcs_ba_camera* cams_in, int n_cams)
{
    DCams cams{};
    for (int c = 0; c < n_cams; ++c) {
        DCam& d = cams.c[c];
        const dcs_ba_camera& s = cams_in[c];
        d.fx = s.fx; d.fy = s.fy; d.cx = s.cx; d.cy = s.cy;
        d.t[0] = s.ext[0]; d.t[1] = s.ext[1]; d.t[2] = s.ext[2];
        d.q[0] = s.ext[3]; d.q[1] = s.ext[4]; d.q[2] = s.ext[5]; d.q[3] = s.ext[6];
        memcpy(d.adj, s.adj, sizeof(d.adj));
    }
    return cams;
}

// dcs_ba_debug_linearize: the blocks of the FIRST linearisation (rows a14 / a15) handed back instead of a solve
struct BaTap { double *Hpp, *bp, *Hll, *bl, *Hpl; int32_t* pose_idx; int* np; };
thread_local const BaTap* tl_tap = nullptr;

}  // namespace

// both kernels over the frames: k_pose_opt2 takes the frames of up to kPoseFastMax edges, k_pose_opt the rest (launched only when the host's
// bound on a frame's edges says one may exist). DCS_POSE_FAST=0: every frame to k_pose_opt (the round-4 kernel: A/B and tests).
static int launch_pose_kernels(PoseArgs a, const DCams& cams, int n_cams, int n_frames, int max_edges_bound, hipStream_t st)
{
    const bool fast = opt(OPT_POSE_FAST) != 0;
    a.fast_max = fast ? kPoseFastMax : -1;
    if (fast) {
        if (opt(OPT_POSE_EXACT_EDGE) != 0) hipLaunchKernelGGL(k_pose_opt2<true>, dim3(n_frames), dim3(kPoT), 0, st, a, cams);
        else hipLaunchKernelGGL(k_pose_opt2<false>, dim3(n_frames), dim3(kPoT), 0, st, a, cams);
    }
    // k_pose_opt behind it when a frame can be beyond k_pose_opt2: more edges than kPoseFastMax, or a rig of more than two cameras with more than
    // 64 * kPoEpt edges on one of them (k_pose_opt2 marks such a frame: kPoseDeclined)
    if (!fast || max_edges_bound > kPoseFastMax || (n_cams > 2 && max_edges_bound > 64 * kPoEpt)) hipLaunchKernelGGL(k_pose_opt, dim3(n_frames), dim3(256), 0, st, a, cams);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

int dcs::launch_pose_opt_device(const PoseOptDevice& p, const dcs_ba_camera* cams_in, int n_cams, int n_frames, int max_edges_bound, hipStream_t st)
{
    if (n_cams < 1 || n_cams > kMaxCams) { set_error("pose optimisation: n_cams must be 1..%d", kMaxCams); return DCS_ERR_INVALID; }
    if (n_frames <= 0) return DCS_OK;
    PoseArgs a{};
    a.poses = p.poses; a.edge_off = p.edge_off; a.edge_cnt = p.edge_cnt; a.xw = p.xw; a.obs = p.obs; a.w = p.w; a.cam = p.cam;
    a.huber = p.huber;
    for (int i = 0; i < 4; ++i) { a.chi2_th[i] = p.chi2_th[i]; a.its[i] = p.its[i]; }
    a.err = p.err; a.level = p.level; a.out_poses = p.out_poses; a.outlier = p.outlier; a.n_inliers = p.n_inliers; a.edge_chi2 = p.edge_chi2; a.n_iters = p.n_iters;
    return launch_pose_kernels(a, pose_cams_of(cams_in, n_cams), n_cams, n_frames, max_edges_bound, st);
}

extern "C" {

int dcs_ba_local_batch(int n_problems, const dcs_ba_problem* const* problems, const volatile uint8_t* const* stop_flags, dcs_ba_result* const* results)
{
    const int B = n_problems;
    if (B < 0 || (B && (!problems || !results))) { set_error("bad BA batch"); return DCS_ERR_INVALID; }
    if (B == 0) return DCS_OK;
    for (int b = 0; b < B; ++b) {
        const dcs_ba_problem* pb = problems[b];
        const dcs_ba_result* res = results[b];
        if (!pb || !res || !res->poses || !res->points || !res->edge_outlier || pb->n_poses < 1 || pb->n_points < 1 || pb->n_edges < 1 ||
            pb->n_cams < 1 || pb->n_cams > kMaxCams || !pb->poses || !pb->pose_fixed || !pb->points || !pb->edge_pose || !pb->edge_point ||
            !pb->edge_cam || !pb->obs || !pb->inv_sigma2 || !pb->cams) {
            set_error("bad BA problem %d (n_cams must be 1..%d)", b, kMaxCams); return DCS_ERR_INVALID;
        }
    }
    int rc = ensure_device();
    if (rc) return rc;
    const bool trace_t = opt(OPT_BA_TRACE) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    const auto t_call0 = now();
    auto stop_requested = [&](int b) { return stop_flags && stop_flags[b] && *stop_flags[b]; };

    // A problem whose stop flag is already set is not optimised at all (Optimizer.cc:582-585: the reference returns before
    // it touches anything): estimates pass through, no outlier is reported. The others form the device batch.
    std::vector<int> live;
    for (int b = 0; b < B; ++b) {
        dcs_ba_result* res = results[b];
        res->n_iters[0] = res->n_iters[1] = 0; res->n_trials[0] = res->n_trials[1] = 0; res->lambda[0] = res->lambda[1] = 0; res->gpu_ms = 0;
        for (int i = 0; i < 32; ++i) res->chi2_trace[i] = 0;
        if (stop_requested(b)) {
            const dcs_ba_problem* pb = problems[b];
            memcpy(res->poses, pb->poses, sizeof(double) * 7 * pb->n_poses);
            memcpy(res->points, pb->points, sizeof(double) * 3 * pb->n_points);
            memset(res->edge_outlier, 0, pb->n_edges);
            if (res->edge_level1) memset(res->edge_level1, 0, pb->n_edges);
            if (res->edge_chi2) memset(res->edge_chi2, 0, sizeof(double) * pb->n_edges);
        } else live.push_back(b);
    }
    const int NB = (int)live.size();
    if (NB == 0) return DCS_OK;

    // ---- structure of every problem (index mapping, CSR lists, pose-pair lists): built ONCE from all edges. The second
    // round only changes the per-edge active mask; inactive edges contribute exact zeros, landmarks without an active
    // edge are skipped (pt_active) and poses without one see a decoupled lambda*I block (zero update), which is
    // what g2o's re-indexing of the active subgraph amounts to.
    // (one list set per slot of the calling thread, kept across calls: fresh vectors of this size are mmap'ed and page-faulted on every call)
    thread_local std::vector<Round> tl_rounds;
    if ((int)tl_rounds.size() < NB) tl_rounds.resize(NB);
    std::vector<Round>& rounds = tl_rounds;
    std::vector<int> dup(NB, -1);
    // The arrays that go up as the caller gave them (estimates, edges, observations, cameras) lie at the FRONT of the upload image, every problem's
    // before any problem's lists: their places depend on the problem sizes alone, so the worker that builds a problem's lists also copies its arrays
    // into the pinned image -- 0.79 MB per C4 problem, 0.25 ms of a 3-ms batch of 8 when one thread copied them after the builds. The pinned
    // buffer is sized beforehand for the worst case of the lists (every pose free, every edge in a pose list).
    // Their LISTS follow, in slots sized for the worst case (every pose free, every edge in a pose list: a few per cent more than a round needs),
    // so that those places do not wait for the builds either: a worker puts its lists into the image as soon as it has built them.
    auto raw_layout = [&](Carver& c, int i, BaProb& q) {
        const dcs_ba_problem* pb = problems[live[i]];
        const size_t P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
        q.poses[0] = c.get<double>(7 * P); q.points[0] = c.get<double>(3 * L);
        q.epose = c.get<int32_t>(E); q.epoint = c.get<int32_t>(E); q.ecam = c.get<int32_t>(E);
        q.obs = c.get<double>(2 * E); q.w = c.get<double>(E); q.active = c.get<uint8_t>(E);
        q.cams = c.get<DCams>(1);
    };
    auto list_layout = [&](Carver& c, int i, BaProb& q) {
        const dcs_ba_problem* pb = problems[live[i]];
        const size_t P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
        q.pose_idx = c.get<int32_t>(P); q.pt_off = c.get<int32_t>(L + 1); q.pt_edges = c.get<int32_t>(E); q.pt_pi = c.get<int32_t>(E);
        q.ps_off = c.get<int32_t>(P + 1); q.ps_edges = c.get<int32_t>(E);
        q.pair_ij = c.get<int32_t>(P * (P + 1));
    };
    BaContext& ctx = ba_context();
    std::vector<BaProb> at(NB);                              // the same places as offsets into the image (Carver with a null base)
    {
        Carver c0;
        size_t dl_bound = 512 + sizeof(BaCtl) * (size_t)NB;
        for (int i = 0; i < NB; ++i) raw_layout(c0, i, at[i]);
        for (int i = 0; i < NB; ++i) {
            list_layout(c0, i, at[i]);
            const dcs_ba_problem* pb = problems[live[i]];
            const size_t P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
            dl_bound += sizeof(double) * (7 * P + 3 * L + E) + 2 * E + 5 * 256;                          // out_poses, out_points, chi2, flag, level1
        }
        const size_t stage_bound = c0.off + sizeof(BaProb) * (size_t)NB + 1024 + dl_bound;
        if ((rc = ctx.prepare_stage(stage_bound))) return rc;
    }
    char* const hs0 = ctx.h_stage;
    {
        auto work = [&](int i) {
            const dcs_ba_problem* pb = problems[live[i]];
            const BaProb& a = at[i];
            auto img = [&](const void* off) { return hs0 + reinterpret_cast<size_t>(off); };
            const size_t P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
            memcpy(img(a.poses[0]), pb->poses, sizeof(double) * 7 * P);
            memcpy(img(a.points[0]), pb->points, sizeof(double) * 3 * L);
            memcpy(img(a.epose), pb->edge_pose, sizeof(int32_t) * E);
            memcpy(img(a.epoint), pb->edge_point, sizeof(int32_t) * E);
            memcpy(img(a.ecam), pb->edge_cam, sizeof(int32_t) * E);
            memcpy(img(a.obs), pb->obs, sizeof(double) * 2 * E);
            memcpy(img(a.w), pb->inv_sigma2, sizeof(double) * E);
            memset(img(a.active), 1, E);
            DCams* cams = reinterpret_cast<DCams*>(img(a.cams));
            memset(cams, 0, sizeof(DCams));
            for (int c = 0; c < pb->n_cams; ++c) {
                DCam& d = cams->c[c];
                const dcs_ba_camera& sc = pb->cams[c];
                d.fx = sc.fx; d.fy = sc.fy; d.cx = sc.cx; d.cy = sc.cy;
                d.t[0] = sc.ext[0]; d.t[1] = sc.ext[1]; d.t[2] = sc.ext[2];
                d.q[0] = sc.ext[3]; d.q[1] = sc.ext[4]; d.q[2] = sc.ext[5]; d.q[3] = sc.ext[6];
                memcpy(d.adj, sc.adj, sizeof(d.adj));
            }
            const Round& r = rounds[i];
            if ((dup[i] = build_round(pb, rounds[i])) != -1) return;
            auto put = [&](const int32_t* off, const std::vector<int32_t>& v) { if (!v.empty()) memcpy(img(off), v.data(), sizeof(int32_t) * v.size()); };
            put(a.pose_idx, r.pose_idx); put(a.pt_off, r.pt_off); put(a.pt_edges, r.pt_edges); put(a.pt_pi, r.pt_pi); put(a.ps_off, r.ps_off); put(a.ps_edges, r.ps_edges);
            put(a.pair_ij, r.pair_ij);
        };
        if (NB == 1) work(0);
        else host_pool().run(NB, work);            // persistent workers: creating 8 threads per call cost more than the lists themselves
        for (int i = 0; i < NB; ++i) {                       // build_round: -1 fine, e >= 0 a duplicate at edge e, -2 - e edge e out of range
            if (dup[i] <= -2) { set_error("problem %d: edge %d out of range", live[i], -2 - dup[i]); return DCS_ERR_INVALID; }
            if (dup[i] >= 0) { set_error("problem %d: more than one edge between pose and point of edge %d", live[i], dup[i]); return DCS_ERR_INVALID; }
        }
        // The device-built pair lists take edge_of[free poses][points] (int32) and one k_schur workgroup per pose PAIR, whatever the
        // covisibility: fine for local BA (tens of poses) and the global BA sizes tested (10 x 70 000, 60 x 800), quadratic beyond. A problem
        // that would need more than 1 GB or 4 M workgroups per step is refused instead of growing silently.
        for (int i = 0; i < NB; ++i) {
            // (the substitution of the n > 256 path keeps z in the 64 KB of dynamic LDS every launch may ask for: n <= 8 192 = 1 365 free poses)
            if ((size_t)rounds[i].n_pad * sizeof(double) > 65536) {
                set_error("problem %d: %d free poses (reduced camera system n = %d) exceed the 8 192 rows of this solver's substitution kernel", live[i], rounds[i].np, rounds[i].n);
                return DCS_ERR_UNSUPPORTED;
            }
            const size_t cells = (size_t)rounds[i].np * (size_t)std::max(problems[live[i]]->n_points, 1);
            if (cells * sizeof(int32_t) > ((size_t)1 << 30) || rounds[i].n_pairs > (4 << 20)) {
                set_error("problem %d: %d free poses x %d points exceeds the pose-pair tables of this solver (edge_of %zu MB, %d pair workgroups)", live[i], rounds[i].np,
                          problems[live[i]]->n_points, cells * sizeof(int32_t) >> 20, rounds[i].n_pairs);
                return DCS_ERR_UNSUPPORTED;
            }
        }
    }
    const double t_build = ms_since(t_call0);
    const bool force_blocked = opt(OPT_BA_FORCE_BLOCKED_LDLT) != 0;   // test hook: the n > 256 path at small n

    // ---- arena layout: [upload | zeroed | scratch | download]
    struct Regions { size_t upload_end, zero_begin, zero_end, dl_begin, dl_end; };
    std::vector<BaProb> hp(NB);
    BaProb* d_probs = nullptr; BaCtl* d_ctls = nullptr;
    // groups of problems = contiguous ranges [g_begin[g], g_begin[g + 1]) of the live list, one stream each. The event timing
    // of dcs_ba_timing brackets launches on ONE stream, so a timed call runs as a single group.
    // Two groups by default: alone, 4 groups of 2 are as fast (18.8 k LM iterations/s for 8 C4 problems, 17.5 k with one group), but
    // next to a busy front end (config C5) four queues of short kernels lose against its long ones (9 k vs 16 k iterations/s).
    int G = NB >= 4 ? 2 : 1;
    if (opt(OPT_BA_GROUPS) > 0) G = std::max(1, std::min({(int)opt(OPT_BA_GROUPS), (int)BaContext::kMaxGroups, NB}));
    if (ctx.timing) G = 1;
    int g_begin[BaContext::kMaxGroups + 1];
    for (int g = 0; g <= G; ++g) g_begin[g] = (int)((long long)NB * g / G);
    unsigned* d_grid_ticket[BaContext::kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
    auto layout = [&](Carver& c, Regions& rg) {
        for (int i = 0; i < NB; ++i) raw_layout(c, i, hp[i]);           // (first, and by the same function as the places the workers copied to)
        for (int i = 0; i < NB; ++i) {
            const dcs_ba_problem* pb = problems[live[i]];
            const Round& r = rounds[i];
            BaProb& q = hp[i];
            const size_t P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
            q.P = (int)P; q.L = (int)L; q.E = (int)E; q.np = r.np; q.n = r.n; q.n_pad = r.n_pad; q.ld = std::max(r.n_pad, kNB); q.n_pairs = r.n_pairs;
            q.nblk = (int)((E + 255) / 256); q.nb_pts = (int)((L + 63) / 64); q.nb_pose = (int)((P + 63) / 64);
            q.use_reg = (r.n <= 256 && !force_blocked) ? 1 : 0;
            q.iters[0] = pb->iters1; q.iters[1] = pb->iters2; q.robust0 = pb->huber_delta > 0.0 ? 1 : 0; q.pad = 0;   // BundleAdjustment(bRobust = false): no kernel
            q.delta = pb->huber_delta; q.chi2_th = pb->chi2_th;
            list_layout(c, i, q);
            q.pt_words = (int)(((L + 31) / 32 + 3) & ~(size_t)3);      // rows are read 16 bytes at a time
        }
        d_probs = c.get<BaProb>(NB);
        rg.upload_end = c.off;
        c.off = (c.off + 255) & ~(size_t)255;
        rg.zero_begin = c.off;
        for (int i = 0; i < NB; ++i) {
            BaProb& q = hp[i];
            q.ticket = c.get<unsigned>(4);
            for (int g = 0; g < G; ++g) if (i == g_begin[g]) d_grid_ticket[g] = c.get<unsigned>(4);
            q.pt_bits = c.get<uint32_t>((size_t)q.np * q.pt_words);
            q.S = q.use_reg ? c.get<double>((size_t)q.ld * q.ld) : nullptr;       // pairs without shared points stay 0
        }
        rg.zero_end = c.off;
        for (int i = 0; i < NB; ++i) {
            BaProb& q = hp[i];
            const size_t P = q.P, L = q.L, E = q.E;
            if (!q.use_reg) q.S = c.get<double>(((size_t)q.n_pad + kNB) * q.ld);
            q.W = c.get<double>(q.use_reg ? 1 : 2 * ((size_t)q.n_pad + kNB) * kNB);
            q.Dgf = c.get<double>(q.use_reg ? 1 : (size_t)(q.n_pad / kNB) * kDiagRec);
            q.poses[1] = c.get<double>(7 * P); q.points[1] = c.get<double>(3 * L);
            q.err = c.get<double>(2 * E);
            q.Hpl[0] = c.get<double>(18 * E); q.BD = c.get<double>(18 * E); q.cpose[0] = c.get<double>(27 * E); q.cpoint[0] = c.get<double>(9 * E);
            q.cpoint[1] = q.cpoint[0];                     // (only k_begin's edge pass writes them to memory, and its readers run in the same step)
            q.Hpl[1] = q.Hpl[0]; q.cpose[1] = q.cpose[0];          // (one linearisation at a time: [1] is [0]; the two indices are what is left of round 5's trial kernel that linearised ahead)
            q.Hll[0] = c.get<double>(9 * L); q.bl[0] = c.get<double>(3 * L); q.Dinv = c.get<double>(9 * L); q.db = c.get<double>(3 * L); q.xl = c.get<double>(3 * L);
            q.Hpp = c.get<double>(36 * P); q.bp = c.get<double>(6 * P); q.bsch = c.get<double>(6 * P); q.xp = c.get<double>(6 * P);
            q.partial = c.get<double>(std::max(q.nblk, q.nb_pts)); q.scale_part = c.get<double>(q.nb_pts + q.nb_pose); q.maxd_part = c.get<double>(q.np + q.nb_pts);
            q.pt_active[0] = c.get<uint8_t>(L);
            q.Hll[1] = q.Hll[0]; q.bl[1] = q.bl[0]; q.pt_active[1] = q.pt_active[0];
            q.edge_of = c.get<int32_t>((size_t)q.np * L);
            q.pair_off = c.get<int32_t>((size_t)q.n_pairs + 1); q.pair_e = c.get<int32_t>(2 * rounds[i].n_pair_entries);
        }
        c.off = (c.off + 255) & ~(size_t)255;
        rg.dl_begin = c.off;
        d_ctls = c.get<BaCtl>(NB);
        for (int i = 0; i < NB; ++i) {
            BaProb& q = hp[i];
            const size_t P = q.P, L = q.L, E = q.E;
            q.out_poses = c.get<double>(7 * P); q.out_points = c.get<double>(3 * L);
            q.chi2 = c.get<double>(E); q.flag = c.get<uint8_t>(E); q.level1 = c.get<uint8_t>(E);
        }
        rg.dl_end = c.off;
    };
    Regions rg{};
    { Carver dry; layout(dry, rg); }
    const size_t arena_bytes = rg.dl_end + 256, dl_bytes = rg.dl_end - rg.dl_begin;
    if ((rc = ctx.prepare(arena_bytes, rg.upload_end + dl_bytes, 16 + (size_t)NB)) || (rc = ctx.prepare_groups(G))) return rc;
    Carver real; real.base = ctx.base;
    layout(real, rg);
    hipStream_t st = ctx.stream;
    char* const hs = ctx.h_stage;
    char* const h_dl = ctx.h_stage + rg.upload_end;
    auto stage = [&](const void* dptr) { return hs + (reinterpret_cast<const char*>(dptr) - ctx.base); };
    auto landed = [&](const void* dptr) { return h_dl + (reinterpret_cast<const char*>(dptr) - (ctx.base + rg.dl_begin)); };

    // ---- the table of the problems completes the staging image (the workers have filled the rest), then ONE copy
    if (hs != hs0 || rg.upload_end + dl_bytes > ctx.stage_cap) { set_error("BA staging image: bound exceeded"); return DCS_ERR_HIP; }    // (cannot happen: prepare_stage's bound)
    for (int i = 0; i < NB; ++i)
        if ((size_t)(reinterpret_cast<const char*>(hp[i].poses[0]) - ctx.base) != reinterpret_cast<size_t>(at[i].poses[0]) ||
            (size_t)(reinterpret_cast<const char*>(hp[i].pair_ij) - ctx.base) != reinterpret_cast<size_t>(at[i].pair_ij)) {
            set_error("BA staging image: layout changed between the two passes"); return DCS_ERR_HIP;
        }
    memcpy(stage(d_probs), hp.data(), sizeof(BaProb) * NB);
    int* const h_words = ctx.h_words;                   // [2g] last finished step of group g, [2g + 1] its finished problems, [16 + i] stop word of problem i
    for (int k = 0; k < 16; ++k) h_words[k] = 0;
    for (int i = 0; i < NB; ++i) h_words[16 + i] = 0;
    const auto t_opt0 = now();
    // From here on the solver's queues hold work that references the arena and the pinned words: EVERY return below -- the error
    // returns of the feed loop included -- leaves the context marked, and the next prepare() / release() drains the streams before it
    // resets or frees anything. Only the ordered download path, which ends with the queues empty, clears the mark.
    ctx.tail_pending = true;
    DCS_HIP(hipMemcpyAsync(ctx.base, hs, rg.upload_end, hipMemcpyHostToDevice, st));
    DCS_HIP(hipMemsetAsync(ctx.base + rg.zero_begin, 0, rg.zero_end - rg.zero_begin, st));
    for (int i = 0; i < NB; ++i)
        if (hp[i].iters[0] <= 0) DCS_HIP(hipMemsetAsync(hp[i].chi2, 0, sizeof(double) * hp[i].E, st));    // no error evaluation will ever write it
    hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(1024), 0, st, (const BaProb*)d_probs, d_ctls, NB);
    // The pose-pair lists of k_schur, built where the edges already are. Nothing before the first k_schur reads them, so they are built on
    // the context's side stream (the one the results come down on) UNDER k_begin / k_reduce_pose / k_prep of step 1: three short launches
    // in front of the step cost a C4 solve 35 us. (A timed call and the parity tap keep to one stream.)
    bool pairs_on_side = false;
    {
        int max_nblk = 0, max_pairs = 0;
        for (int i = 0; i < NB; ++i) { max_nblk = std::max(max_nblk, hp[i].nblk); if (hp[i].np) max_pairs = std::max(max_pairs, hp[i].n_pairs); }
        if (max_pairs) {
            const bool side_env = opt(OPT_BA_PAIRS_SIDE) != 0;
            hipStream_t ps = st;
            if (side_env && !ctx.timing && !tl_tap) {
                if (!ctx.dl) { const int rc_s = ctx.create_stream(&ctx.dl); if (rc_s) return rc_s; }
                if (!ctx.ev_staged) DCS_HIP(hipEventCreateWithFlags(&ctx.ev_staged, hipEventDisableTiming));
                if (!ctx.ev_pairs) DCS_HIP(hipEventCreateWithFlags(&ctx.ev_pairs, hipEventDisableTiming));
                DCS_HIP(hipEventRecord(ctx.ev_staged, st));
                DCS_HIP(hipStreamWaitEvent(ctx.dl, ctx.ev_staged, 0));
                ps = ctx.dl; pairs_on_side = true;
            }
            hipLaunchKernelGGL(k_pairs_mark, dim3(max_nblk, NB), dim3(256), 0, ps, (const BaProb*)d_probs);
            hipLaunchKernelGGL(k_pairs_count, dim3((max_pairs + 255) / 256, NB), dim3(256), 0, ps, (const BaProb*)d_probs);
            hipLaunchKernelGGL(k_pairs_scan, dim3(NB), dim3(1024), 0, ps, (const BaProb*)d_probs);
            hipLaunchKernelGGL(k_pairs_fill, dim3(max_pairs, NB), dim3(64), 0, ps, (const BaProb*)d_probs);
            if (pairs_on_side) DCS_HIP(hipEventRecord(ctx.ev_pairs, ps));
        }
    }
    DCS_CHECK_LAUNCH();
    if (G > 1) {
        DCS_HIP(hipEventRecord(ctx.ev_up, st));
        for (int g = 1; g < G; ++g) DCS_HIP(hipStreamWaitEvent(ctx.aux[g - 1], ctx.ev_up, 0));
    }

    if (const BaTap* tap = tl_tap) {
        // parity tap: the first two launches of step 1 -- k_begin (errors of the initial estimates, linearizeOplus, constructQuadraticForm per
        // edge) and k_reduce_pose (the per-vertex sums) -- then the blocks come down as they lie in the arena and the call ends
        const BaProb& q = hp[0];
        hipLaunchKernelGGL(k_begin, dim3(q.nblk, 1), dim3(256), 0, st, (const BaProb*)d_probs, d_ctls, 1, h_words, d_grid_ticket[0]);
        hipLaunchKernelGGL(k_reduce_pose, dim3(q.np + q.nb_pts, 1), dim3(1024), 0, st, (const BaProb*)d_probs, d_ctls);
        DCS_CHECK_LAUNCH();
        DCS_HIP(hipStreamSynchronize(st));
        ctx.tail_pending = false;
        if (q.np) {
            DCS_HIP(hipMemcpy(tap->Hpp, q.Hpp, sizeof(double) * 36 * q.np, hipMemcpyDeviceToHost));
            DCS_HIP(hipMemcpy(tap->bp, q.bp, sizeof(double) * 6 * q.np, hipMemcpyDeviceToHost));
        }
        DCS_HIP(hipMemcpy(tap->Hll, q.Hll[0], sizeof(double) * 9 * q.L, hipMemcpyDeviceToHost));
        DCS_HIP(hipMemcpy(tap->bl, q.bl[0], sizeof(double) * 3 * q.L, hipMemcpyDeviceToHost));
        DCS_HIP(hipMemcpy(tap->Hpl, q.Hpl[0], sizeof(double) * 18 * q.E, hipMemcpyDeviceToHost));
        const dcs_ba_problem* pb0 = problems[live[0]];
        for (int e = 0; e < q.E; ++e)                          // edges of fixed poses have no H_pl block (the kernel never writes their slot)
            if (rounds[0].pose_idx[pb0->edge_pose[e]] < 0) for (int k = 0; k < 18; ++k) tap->Hpl[(size_t)e * 18 + k] = 0.0;
        memcpy(tap->pose_idx, rounds[0].pose_idx.data(), sizeof(int32_t) * q.P);
        *tap->np = q.np;
        return DCS_OK;
    }

    // ---- launch geometry of one step of a group (its largest problem decides; smaller ones exit early)
    struct Group {
        hipStream_t st; const BaProb* dp; BaCtl* ctls; int nb, off; int* words; unsigned* ticket;
        int g_edges = 0, g_reduce = 0, g_prep = 0, g_schur = 0, g_update = 0, g_pts = 0, max_npad_blocked = 0;
        bool any_mfma = false, any_blocked = false, finished = false, fused_update = true;
        int g_schur_w = 0;
        int max_n_mfma = 0;
    };
    const bool no_fused_update = opt(OPT_BA_FUSED_UPDATE) == 0;   // A/B: the two launches
    std::vector<Group> groups((size_t)G);
    int max_steps = 0;
    for (int g = 0; g < G; ++g) {
        Group& gr = groups[g];
        gr.st = g == 0 ? st : ctx.aux[g - 1]; gr.off = g_begin[g]; gr.nb = g_begin[g + 1] - g_begin[g];
        gr.dp = d_probs + gr.off; gr.ctls = d_ctls + gr.off; gr.words = h_words + 2 * g; gr.ticket = d_grid_ticket[g];
        for (int i = gr.off; i < gr.off + gr.nb; ++i) {
            const BaProb& q = hp[i];
            gr.g_edges = std::max(gr.g_edges, q.nblk);
            gr.g_reduce = std::max(gr.g_reduce, q.np + q.nb_pts);
            gr.g_prep = std::max(gr.g_prep, (q.np ? q.nblk : 0) + (q.L + 255) / 256);
            if (q.np) { gr.g_schur = std::max(gr.g_schur, q.n_pairs + q.np); gr.g_schur_w = std::max(gr.g_schur_w, (q.n_pairs + kSchurWPairs - 1) / kSchurWPairs + q.np); }
            gr.g_update = std::max(gr.g_update, q.nb_pts + q.nb_pose);
            gr.g_pts = std::max(gr.g_pts, q.nb_pts);
            if (q.P > kFusedMaxPoses || no_fused_update) gr.fused_update = false;
            max_steps = std::max(max_steps, (std::max(q.iters[0], 0) + std::max(q.iters[1], 0)) * 10 + 2);
            if (q.np) {
                if (q.use_reg == 1) gr.max_n_mfma = std::max(gr.max_n_mfma, q.n);
                gr.any_mfma |= q.use_reg == 1; gr.any_blocked |= q.use_reg == 0;
                if (q.use_reg == 0) gr.max_npad_blocked = std::max(gr.max_npad_blocked, q.n_pad);
            }
        }
    }
    // (the errors of the initial estimates -- or, with iters1 <= 0, the flags straight away -- are the first thing k_begin of step 1 does)
    const bool timing = ctx.timing;
    auto event_at = [&](size_t i) -> hipEvent_t {
        while (ctx.events.size() <= i) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; ctx.events.push_back(e); }
        return ctx.events[i];
    };
    auto mark = [&](int step, int k) { if (timing) { hipEvent_t e = event_at((size_t)(step - 1) * 4 + k); if (e) (void)hipEventRecord(e, st); } };
    // ---- the step as a hipGraph: OPT-IN (DCS_BA_GRAPH=1). Measured on MI355X / ROCm 7.0: replaying the 8-kernel step as one
    // hipGraphLaunch is SLOWER than 8 plain launches -- 2.36 vs 2.28 ms per C4 solve, 3.61 vs 3.39 ms for 8 problems -- because the host
    // is not the bottleneck here (it feeds the queue two steps ahead and spends 80 % of a solve waiting) and the graph's kernel
    // nodes are dispatched no faster than stream launches. The graph is rebuilt per call: hipGraphExecKernelNodeSetParams did not
    // take new grid sizes reliably (a problem of another size then never finished).
    const bool use_graph_env = opt(OPT_BA_GRAPH) != 0;
    std::vector<hipGraphExec_t> step_exec((size_t)G, nullptr);
    if (use_graph_env && !timing) {
        for (int g = 0; g < G; ++g) {
            const Group& gr = groups[g];
            if (gr.any_blocked) continue;
            const BaProb* dp = gr.dp; BaCtl* ctls = gr.ctls; const BaCtl* cctls = gr.ctls; int nb = gr.nb;
            const volatile int* d_stop = h_words + 16 + gr.off; int* words = gr.words; unsigned* ticket = gr.ticket;
            void* a_cc[] = {(void*)&dp, (void*)&cctls};                                   // (probs, const ctls)
            void* a_c[] = {(void*)&dp, (void*)&ctls};                                     // (probs, ctls)
            void* a_err[] = {(void*)&dp, (void*)&ctls, (void*)&d_stop, (void*)&nb, (void*)&words, (void*)&ticket};
            void* a_begin[] = {(void*)&dp, (void*)&ctls, (void*)&nb, (void*)&words, (void*)&ticket};
            struct Spec { void* fn; dim3 grid, block; void** args; };
            std::vector<Spec> spec;
            spec.push_back({(void*)k_begin, dim3(gr.g_edges, nb), dim3(256), a_begin});
            spec.push_back({(void*)k_reduce_pose, dim3(gr.g_reduce, nb), dim3(1024), a_c});
            spec.push_back({(void*)k_prep, dim3(gr.g_prep, nb), dim3(256), a_c});
            if (gr.g_schur && schur_use_wave(nb, gr.g_schur)) spec.push_back({(void*)k_schur_w, dim3(gr.g_schur_w, nb), dim3(128 * kSchurWPairs), a_cc});
            else if (gr.g_schur) spec.push_back({(void*)k_schur<28>, dim3(gr.g_schur, nb), dim3(1024), a_cc});
            if (gr.any_mfma) spec.push_back({gr.max_n_mfma <= 240 ? (void*)k_ldlt_mfma<kLdltSlotsSmall> : (void*)k_ldlt_mfma<kLdltSlotsBig>, dim3(nb), dim3(kLdltThreads), a_c});
            if (gr.fused_update) spec.push_back({(void*)k_update_error, dim3(gr.g_pts, nb), dim3(kFusedThreads), a_err});
            else {
                spec.push_back({(void*)k_solve_update, dim3(gr.g_update, nb), dim3(64), a_cc});
                spec.push_back({(void*)k_error<1>, dim3(gr.g_edges, nb), dim3(256), a_err});
            }
            const unsigned shape = (gr.g_schur ? 1u : 0u) | (gr.any_mfma ? 2u : 0u) | 8u;
            BaContext::StepGraph& sg = ctx.step_graph[g];
            auto params_of = [](const Spec& sp) { hipKernelNodeParams kp{}; kp.func = sp.fn; kp.gridDim = sp.grid; kp.blockDim = sp.block; kp.sharedMemBytes = 0; kp.kernelParams = sp.args; kp.extra = nullptr; return kp; };
            bool ok = true;
            {
                ctx.drop_graph(sg);
                ok = hipGraphCreate(&sg.graph, 0) == hipSuccess;
                hipGraphNode_t prev = nullptr;
                for (size_t k = 0; k < spec.size() && ok; ++k) {
                    hipKernelNodeParams kp = params_of(spec[k]);
                    hipGraphNode_t node = nullptr;
                    ok = hipGraphAddKernelNode(&node, sg.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp) == hipSuccess;
                    sg.nodes.push_back(node); prev = node;
                }
                ok = ok && hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0) == hipSuccess;
                if (ok) sg.shape = shape; else { (void)hipGetLastError(); ctx.drop_graph(sg); }
            }
            step_exec[(size_t)g] = ok ? sg.exec : nullptr;            // nullptr: fall back to plain launches for this group
        }
    }
    auto enqueue_step = [&](const Group& gr, int step) -> int {
        hipStream_t gs = gr.st;
        const BaProb* dp = gr.dp;
        BaCtl* ctls = gr.ctls;
        const int nb = gr.nb;
        const volatile int* d_stop = h_words + 16 + gr.off;
        if (hipGraphExec_t ge = step_exec[(size_t)(&gr - groups.data())]) {
            if (step == 1 && pairs_on_side) DCS_HIP(hipStreamWaitEvent(gs, ctx.ev_pairs, 0));
            DCS_HIP(hipGraphLaunch(ge, gs));
            return DCS_OK;
        }
        mark(step, 0);
        hipLaunchKernelGGL(k_begin, dim3(gr.g_edges, nb), dim3(256), 0, gs, dp, ctls, nb, gr.words, gr.ticket);            // round change / stale errors, then buildSystem
        hipLaunchKernelGGL(k_reduce_pose, dim3(gr.g_reduce, nb), dim3(1024), 0, gs, dp, ctls);                             // + computeLambdaInit (first iteration)
        hipLaunchKernelGGL(k_prep, dim3(gr.g_prep, nb), dim3(256), 0, gs, dp, ctls);                                      // setLambda + solve (Schur)
        DCS_CHECK_LAUNCH();
        if (gr.any_blocked) {                             // the blocked fallback factors S in place: rebuild it every trial
            for (int i = gr.off; i < gr.off + nb; ++i)
                if (hp[i].np && hp[i].use_reg == 0) DCS_HIP(hipMemsetAsync(hp[i].S, 0, sizeof(double) * ((size_t)hp[i].n_pad + kNB) * hp[i].ld, gs));
            hipLaunchKernelGGL(k_pad_identity, dim3(1, nb), dim3(64), 0, gs, dp, (const BaCtl*)ctls);
        }
        if (gr.g_schur) {
            if (step == 1 && pairs_on_side) DCS_HIP(hipStreamWaitEvent(gs, ctx.ev_pairs, 0));      // the pair lists (side stream)
            if (schur_use_wave(nb, gr.g_schur)) hipLaunchKernelGGL(k_schur_w, dim3(gr.g_schur_w, nb), dim3(128 * kSchurWPairs), 0, gs, dp, (const BaCtl*)ctls);
            else hipLaunchKernelGGL(k_schur<28>, dim3(gr.g_schur, nb), dim3(1024), 0, gs, dp, (const BaCtl*)ctls);
        }
        mark(step, 1);
        if (gr.any_mfma) {
            if (gr.max_n_mfma <= 240) hipLaunchKernelGGL(k_ldlt_mfma<kLdltSlotsSmall>, dim3(nb), dim3(kLdltThreads), 0, gs, dp, ctls);
            else hipLaunchKernelGGL(k_ldlt_mfma<kLdltSlotsBig>, dim3(nb), dim3(kLdltThreads), 0, gs, dp, ctls);
        }
        if (gr.any_blocked) {
            const int m = gr.max_npad_blocked / kNB;      // block columns; launch s: tiles (I, J), s <= J < m, J <= I <= m (block row m = the right-hand side)
            for (int s2 = 0; s2 < m; ++s2) hipLaunchKernelGGL(k_ldlt_step, dim3(m - s2 + 1, s2 == 0 ? 1 : m - s2, nb), dim3(64), 0, gs, dp, ctls, s2);
            hipLaunchKernelGGL(k_ldlt_back, dim3(nb), dim3(kBackT), sizeof(double) * gr.max_npad_blocked, gs, dp, (const BaCtl*)ctls);
        }
        DCS_CHECK_LAUNCH();
        mark(step, 2);
        if (gr.fused_update) {
            hipLaunchKernelGGL(k_update_error, dim3(gr.g_pts, nb), dim3(kFusedThreads), 0, gs, dp, ctls, d_stop, nb, gr.words, gr.ticket);   // back-substitution, oplus, chi2 of the trial + computeScale + accept / reject, progress
        } else {
            hipLaunchKernelGGL(k_solve_update, dim3(gr.g_update, nb), dim3(64), 0, gs, dp, (const BaCtl*)ctls);
            hipLaunchKernelGGL(k_error<1>, dim3(gr.g_edges, nb), dim3(256), 0, gs, dp, ctls, d_stop, nb, gr.words, gr.ticket);   // chi2 of the trial + computeScale + accept / reject, progress
        }
        mark(step, 3);
        DCS_CHECK_LAUNCH();
        return DCS_OK;
    };

    // ---- the host only feeds the queues: kLookahead steps ahead of what the device has reported finished, group by group
    // Steps already in the queue when the device reports the last problem done still run, as 8 no-op launches of ~5 us each, in front
    // of the download (75 us of a 2 ms C4 solve at two steps ahead). One step ahead (DCS_BA_LOOKAHEAD=1) measured +1 % for one C4 problem
    // and nothing for a batch; two stays the default because small problems (a step of 50 us) need the slack: enqueueing a step
    // costs the host 30 - 50 us per group.
    // (0 = by batch size. Round 6, after the host side stopped dominating a batch: the steps of four or more problems are long enough -- 200 us
    // for eight C4 problems against 80 us of enqueueing for two groups -- that ONE step ahead keeps the queues fed, and the call behind this one
    // waits for one no-op step less: 2.86 -> 2.77 ms per batch of 8. One problem: 1.61 vs 1.62 ms, stays at two.)
    const int kLookahead = opt(OPT_BA_LOOKAHEAD) > 0 ? (int)opt(OPT_BA_LOOKAHEAD) : (NB >= 4 ? 1 : 2);
    auto load_words = [&](const Group& gr, int& step_done, int& n_done) {
        const unsigned long long w = __atomic_load_n(reinterpret_cast<const unsigned long long*>(gr.words), __ATOMIC_ACQUIRE);   // {done : step}, one word
        step_done = (int)(unsigned)w;
        n_done = (int)(w >> 32);
    };
    auto refresh_stop = [&] { for (int i = 0; i < NB; ++i) if (stop_requested(live[i])) __atomic_store_n(h_words + 16 + i, 1, __ATOMIC_RELAXED); };
    int steps = 0, n_finished = 0;
    double t_wait = 0;
    while (n_finished < G && steps < max_steps) {
        refresh_stop();
        ++steps;
        for (Group& gr : groups) if (!gr.finished && (rc = enqueue_step(gr, steps))) return rc;
        const auto tw0 = now();
        for (Group& gr : groups) {
            if (gr.finished) continue;
            for (;;) {
                int sd, nd;
                load_words(gr, sd, nd);
                if (nd == gr.nb) { gr.finished = true; ++n_finished; break; }
                if (sd >= steps - kLookahead) break;
                refresh_stop();
                const hipError_t qe = hipStreamQuery(gr.st);
                if (qe == hipSuccess) {                   // queue drained: the words are final
                    load_words(gr, sd, nd);
                    if (nd == gr.nb) { gr.finished = true; ++n_finished; }
                    else if (sd < steps) { set_error("BA device loop lost step %d (reported %d)", steps, sd); return DCS_ERR_HIP; }
                    break;
                }
                if (qe != hipErrorNotReady) { set_error("BA stream: %s", hipGetErrorString(qe)); return DCS_ERR_HIP; }
            }
        }
        t_wait += ms_since(tw0);
    }
    // When every group has REPORTED its problems done (the progress words are stored with a system-scope release by the last block
    // of k_begin / k_error<1>, after every result array and every BaCtl of the step), the steps still in the queues are no-ops: they return on
    // state > ST_RETRY before they write anything but the progress words. The results then come down on a stream of their own
    // instead of behind those 2 x 8 launches (~75 us of a 2 ms C4 solve). DCS_BA_DL_STREAM=0, or a batch that ran into max_steps,
    // takes the ordered path: the download behind everything on the solver's streams.
    const bool dl_own = opt(OPT_BA_DL_STREAM) != 0;
    const auto t_dl0 = now();
    if (dl_own && n_finished == G) {
        if (!ctx.dl) { const int rc_s = ctx.create_stream(&ctx.dl); if (rc_s) return rc_s; }
        DCS_HIP(hipMemcpyAsync(h_dl, ctx.base + rg.dl_begin, dl_bytes, hipMemcpyDeviceToHost, ctx.dl));
        DCS_HIP(hipStreamSynchronize(ctx.dl));
    } else {
        for (int g = 1; g < G; ++g) {                     // the download waits for every group
            DCS_HIP(hipEventRecord(ctx.ev_done[g - 1], ctx.aux[g - 1]));
            DCS_HIP(hipStreamWaitEvent(st, ctx.ev_done[g - 1], 0));
        }
        DCS_HIP(hipMemcpyAsync(h_dl, ctx.base + rg.dl_begin, dl_bytes, hipMemcpyDeviceToHost, st));
        DCS_HIP(hipStreamSynchronize(st));
        if (pairs_on_side) DCS_HIP(hipStreamSynchronize(ctx.dl));
        ctx.tail_pending = false;                           // every group's queue was waited for: nothing is in flight
    }
    const float opt_ms = (float)ms_since(t_opt0);
    const double dl_ms = ms_since(t_dl0);
    const BaCtl* h_ctls = reinterpret_cast<const BaCtl*>(landed(d_ctls));
    if (timing) {                                         // only the steps in which at least one problem ran a trial
        int real = 0;
        for (int i = 0; i < NB; ++i) real = std::max(real, h_ctls[i].n_trials[0] + h_ctls[i].n_trials[1]);
        for (int sidx = 0; sidx < std::min(real, steps); ++sidx) {
            float a = 0, b = 0;
            if ((size_t)sidx * 4 + 3 < ctx.events.size() && hipEventElapsedTime(&a, ctx.events[sidx * 4 + 1], ctx.events[sidx * 4 + 2]) == hipSuccess &&
                hipEventElapsedTime(&b, ctx.events[sidx * 4], ctx.events[sidx * 4 + 3]) == hipSuccess) {
                ctx.t_ldlt_us += 1e3 * a; ctx.n_ldlt += 1; ctx.t_step_us += 1e3 * b; ctx.n_step += 1;
            }
        }
    }
    for (int i = 0; i < NB; ++i) {
        const BaProb& q = hp[i];
        dcs_ba_result* res = results[live[i]];
        const BaCtl& c = h_ctls[i];
        if (c.state != ST_DONE) { set_error("BA problem %d did not finish within %d steps", live[i], max_steps); return DCS_ERR_HIP; }
        memcpy(res->poses, landed(q.out_poses), sizeof(double) * 7 * q.P);
        memcpy(res->points, landed(q.out_points), sizeof(double) * 3 * q.L);
        memcpy(res->edge_outlier, landed(q.flag), q.E);
        if (res->edge_level1) memcpy(res->edge_level1, landed(q.level1), q.E);
        if (res->edge_chi2) memcpy(res->edge_chi2, landed(q.chi2), sizeof(double) * q.E);
        for (int k = 0; k < 2; ++k) { res->n_iters[k] = c.n_iters[k]; res->n_trials[k] = c.n_trials[k]; res->lambda[k] = c.lambda[k]; }
        for (int k = 0; k < 32; ++k) res->chi2_trace[k] = c.chi2_trace[k];
        res->gpu_ms = opt_ms;
    }
    if (trace_t)
        fprintf(stderr, "[dcs_ba] %d problems, total %.3f ms: build_round %.3f, layout + staging %.3f, optimise %.3f (%d steps enqueued, host waited %.3f, download %.3f)\n",
                NB, ms_since(t_call0), t_build, ms_since(t_call0) - t_build - opt_ms, (double)opt_ms, steps, t_wait, dl_ms);
    return DCS_OK;
}

// diagnostic (not part of the public header): average host time in ms of the structure build of one problem, no GPU needed
double dcs_debug_ba_build_ms(const dcs_ba_problem* pb, int reps)
{
    if (!pb || reps < 1) return -1.0;
    Round r;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) if (build_round(pb, r) != -1) return -2.0;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int dcs_ba_set_cu_range(int first_cu, int n_cus)
{
    if (first_cu < 0 || n_cus < 0) { set_error("dcs_ba_set_cu_range: bad range"); return DCS_ERR_INVALID; }
    ba_cu_range(first_cu, n_cus, true);
    return DCS_OK;
}

int dcs_ba_avoid_streams(void* const* avoid, int n_avoid)
{
    if (n_avoid < 0 || (n_avoid && !avoid)) { set_error("dcs_ba_avoid_streams: bad argument"); return DCS_ERR_INVALID; }
    std::vector<hipStream_t> v;
    for (int i = 0; i < n_avoid; ++i) v.push_back((hipStream_t)avoid[i]);          // a null entry = the legacy default stream: it has a queue too
    ba_avoid_list(v, true);
    return DCS_OK;
}

int dcs_ba_release_thread(void)
{
    ba_context().release();
    return DCS_OK;
}

int dcs_ba_timing(int on, double out[4])
{
    BaContext& c = ba_context();
    if (out) { out[0] = c.t_ldlt_us; out[1] = c.n_ldlt; out[2] = c.t_step_us; out[3] = c.n_step; }
    c.t_ldlt_us = c.n_ldlt = c.t_step_us = c.n_step = 0;
    c.timing = on != 0;
    return DCS_OK;
}

int dcs_ba_local(const dcs_ba_problem* pb, const volatile uint8_t* stop_flag, dcs_ba_result* res)
{
    return dcs_ba_local_batch(1, &pb, &stop_flag, &res);
}

#ifdef DCS_POSE_PROF
// side builds only: the timeline k_pose_opt2 left for frame 0 of its last launch ([8 waves][16] clock values, see g_pose_prof)
extern "C" int dcs_debug_pose_prof(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_prof), sizeof(unsigned long long) * (kPoT / 64) * 16) == hipSuccess ? 0 : -1;
}
#endif

int dcs_ba_debug_linearize(const dcs_ba_problem* pb, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, int32_t* pose_idx, int* n_free)
{
    if (!pb || !Hpp || !bp || !Hll || !bl || !Hpl || !pose_idx || !n_free) { set_error("dcs_ba_debug_linearize: null argument"); return DCS_ERR_INVALID; }
    std::vector<double> poses((size_t)7 * std::max(pb->n_poses, 1)), points((size_t)3 * std::max(pb->n_points, 1));
    std::vector<uint8_t> flags((size_t)std::max(pb->n_edges, 1));
    dcs_ba_result res{};
    res.poses = poses.data(); res.points = points.data(); res.edge_outlier = flags.data();
    dcs_ba_result* rp = &res;
    // a problem the batch call answers without linearising (no free pose, no live edge: estimates copied through) never fires the tap: the
    // caller then reads a defined "nothing" instead of whatever its buffers held
    *n_free = -1;
    for (int i = 0; i < pb->n_poses; ++i) pose_idx[i] = -1;
    const BaTap tap{Hpp, bp, Hll, bl, Hpl, pose_idx, n_free};
    tl_tap = &tap;
    const int rc = dcs_ba_local_batch(1, &pb, nullptr, &rp);
    tl_tap = nullptr;
    if (rc == DCS_OK && *n_free < 0) { *n_free = 0; set_error("dcs_ba_debug_linearize: the problem has nothing to linearise (no free pose or no edge)"); return DCS_ERR_INVALID; }
    return rc;
}

int dcs_pose_optimization(const dcs_pose_problem* pb, dcs_pose_result* res)
{
    if (!pb || !res || pb->n_frames < 0 || pb->n_cams < 1 || pb->n_cams > kMaxCams || !pb->cams || (pb->n_frames && (!pb->poses || !pb->edge_off ||
        !res->poses || !res->n_inliers))) { set_error("bad pose-optimisation problem (n_cams must be 1..%d)", kMaxCams); return DCS_ERR_INVALID; }
    const int F = pb->n_frames;
    if (F == 0) return DCS_OK;
    const int E = pb->edge_off[F];
    for (int f = 0; f < F; ++f) if (pb->edge_off[f + 1] < pb->edge_off[f]) { set_error("edge_off not ascending at frame %d", f); return DCS_ERR_INVALID; }
    if (E && (!pb->xw || !pb->obs || !pb->inv_sigma2 || !pb->edge_cam || !res->outlier)) { set_error("null edge array"); return DCS_ERR_INVALID; }
    for (int e = 0; e < E; ++e) if (pb->edge_cam[e] < 0 || pb->edge_cam[e] >= pb->n_cams) { set_error("edge %d: camera out of range", e); return DCS_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    DCams cams{};
    for (int c = 0; c < pb->n_cams; ++c) {
        DCam& d = cams.c[c];
        const dcs_ba_camera& s = pb->cams[c];
        d.fx = s.fx; d.fy = s.fy; d.cx = s.cx; d.cy = s.cy;
        d.t[0] = s.ext[0]; d.t[1] = s.ext[1]; d.t[2] = s.ext[2];
        d.q[0] = s.ext[3]; d.q[1] = s.ext[4]; d.q[2] = s.ext[5]; d.q[3] = s.ext[6];
        memcpy(d.adj, s.adj, sizeof(d.adj));
    }
    const size_t Ee = std::max(E, 1);
    auto carve = [&](Carver& ar, double*& d_poses, double*& d_out, int32_t*& d_off, int32_t*& d_ninl, int32_t*& d_nit, double*& d_xw, double*& d_obs,
                     double*& d_w, double*& d_err, double*& d_chi, int32_t*& d_cam, uint8_t*& d_level, uint8_t*& d_outl) {
        d_poses = ar.get<double>(7 * (size_t)F); d_out = ar.get<double>(7 * (size_t)F);
        d_off = ar.get<int32_t>(F + 1); d_ninl = ar.get<int32_t>(F); d_nit = ar.get<int32_t>(4 * (size_t)F);
        d_xw = ar.get<double>(3 * Ee); d_obs = ar.get<double>(2 * Ee); d_w = ar.get<double>(Ee);
        d_err = ar.get<double>(2 * Ee); d_chi = ar.get<double>(Ee);
        d_cam = ar.get<int32_t>(Ee); d_level = ar.get<uint8_t>(Ee); d_outl = ar.get<uint8_t>(Ee);
    };
    double *d_poses, *d_out, *d_xw, *d_obs, *d_w, *d_err, *d_chi;
    int32_t *d_off, *d_ninl, *d_nit, *d_cam;
    uint8_t *d_level, *d_outl;
    Carver dry;
    carve(dry, d_poses, d_out, d_off, d_ninl, d_nit, d_xw, d_obs, d_w, d_err, d_chi, d_cam, d_level, d_outl);
    BaContext& ctx = ba_context();
    if ((rc = ctx.prepare(dry.off + 256, 0, 16))) return rc;
    Carver ar; ar.base = ctx.base;
    carve(ar, d_poses, d_out, d_off, d_ninl, d_nit, d_xw, d_obs, d_w, d_err, d_chi, d_cam, d_level, d_outl);
    hipStream_t st = ctx.stream;
    DCS_HIP(hipMemcpyAsync(d_poses, pb->poses, sizeof(double) * 7 * F, hipMemcpyHostToDevice, st));
    DCS_HIP(hipMemcpyAsync(d_off, pb->edge_off, sizeof(int32_t) * (F + 1), hipMemcpyHostToDevice, st));
    if (E) {
        DCS_HIP(hipMemcpyAsync(d_xw, pb->xw, sizeof(double) * 3 * E, hipMemcpyHostToDevice, st));
        DCS_HIP(hipMemcpyAsync(d_obs, pb->obs, sizeof(double) * 2 * E, hipMemcpyHostToDevice, st));
        DCS_HIP(hipMemcpyAsync(d_w, pb->inv_sigma2, sizeof(double) * E, hipMemcpyHostToDevice, st));
        DCS_HIP(hipMemcpyAsync(d_cam, pb->edge_cam, sizeof(int32_t) * E, hipMemcpyHostToDevice, st));
    }
    PoseArgs a{};
    a.poses = d_poses; a.edge_off = d_off; a.xw = d_xw; a.obs = d_obs; a.w = d_w; a.cam = d_cam;
    a.huber = pb->huber_delta;
    for (int i = 0; i < 4; ++i) { a.chi2_th[i] = pb->chi2_th[i]; a.its[i] = pb->its[i]; }
    a.err = d_err; a.level = d_level; a.out_poses = d_out; a.outlier = d_outl; a.n_inliers = d_ninl;
    a.edge_chi2 = res->edge_chi2 ? d_chi : nullptr; a.n_iters = res->n_iters ? d_nit : nullptr;
    int max_edges = 0;
    for (int f = 0; f < F; ++f) max_edges = std::max(max_edges, pb->edge_off[f + 1] - pb->edge_off[f]);
    if ((rc = launch_pose_kernels(a, cams, pb->n_cams, F, max_edges, st))) return rc;
    DCS_HIP(hipMemcpyAsync(res->poses, d_out, sizeof(double) * 7 * F, hipMemcpyDeviceToHost, st));
    DCS_HIP(hipMemcpyAsync(res->n_inliers, d_ninl, sizeof(int32_t) * F, hipMemcpyDeviceToHost, st));
    if (E) DCS_HIP(hipMemcpyAsync(res->outlier, d_outl, E, hipMemcpyDeviceToHost, st));
    if (E && res->edge_chi2) DCS_HIP(hipMemcpyAsync(res->edge_chi2, d_chi, sizeof(double) * E, hipMemcpyDeviceToHost, st));
    if (res->n_iters) DCS_HIP(hipMemcpyAsync(res->n_iters, d_nit, sizeof(int32_t) * 4 * F, hipMemcpyDeviceToHost, st));
    DCS_HIP(hipStreamSynchronize(st));
    return DCS_OK;
}

int dcs_rig_adjoint(const float T[16], int exact, double ext7[7], double adj[36])
{
    if (!T || !ext7 || !adj) { set_error("null argument"); return DCS_ERR_INVALID; }
    double R[9], t[3], q[4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
    qfromR(R, q); qnormalize(q);
    ext7[0] = t[0]; ext7[1] = t[1]; ext7[2] = t[2]; ext7[3] = q[0]; ext7[4] = q[1]; ext7[5] = q[2]; ext7[6] = q[3];
    for (int i = 0; i < 36; ++i) adj[i] = 0;
    if (!exact) {                                      // Cameras::setExtrinsics: float [[R, R t^],[?, R]], LL := 0
        const float th[9] = {0, -T[11], T[7], T[11], 0, -T[3], -T[7], T[3], 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            float acc = 0;
            for (int k = 0; k < 3; ++k) acc += T[i * 4 + k] * th[k * 3 + j];
            adj[i * 6 + j] = T[i * 4 + j]; adj[(i + 3) * 6 + j + 3] = T[i * 4 + j]; adj[i * 6 + j + 3] = acc;
        }
    } else {                                           // g2o SE3Quat::adj(): [[R, 0],[t^ R, R]]
        double Rn[9]; qtoR(q, Rn);
        const double th[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            adj[i * 6 + j] = Rn[i * 3 + j]; adj[(i + 3) * 6 + j + 3] = Rn[i * 3 + j];
            adj[(i + 3) * 6 + j] = th[i * 3] * Rn[j] + th[i * 3 + 1] * Rn[3 + j] + th[i * 3 + 2] * Rn[6 + j];
        }
    }
    return DCS_OK;
}

int dcs_pose_from_matrix(const float T[16], double p[7])      // Converter::toSE3Quat (Converter.cc:58-68)
{
    if (!T || !p) { set_error("null argument"); return DCS_ERR_INVALID; }
    double R[9], q[4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j]; p[i] = T[i * 4 + 3]; }
    qfromR(R, q); qnormalize(q);
    p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
    return DCS_OK;
}

int dcs_pose_to_matrix(const double p[7], float T[16])        // Converter::toCvMat(SE3Quat) (Converter.cc:70-74)
{
    if (!T || !p) { set_error("null argument"); return DCS_ERR_INVALID; }
    double R[9];
    qtoR(p + 3, R);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j]; T[i * 4 + 3] = (float)p[i]; }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    return DCS_OK;
}

}  // extern "C"
