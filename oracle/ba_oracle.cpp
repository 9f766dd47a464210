/*
 * ba_oracle.cpp -- CPU restatement of Optimizer::LocalBundleAdjustment numerics
 * (TEST INFRASTRUCTURE ONLY; see oracle.h).
 *
 * Follows /root/reference/src/Optimizer.cc:407-696 and the arithmetic of the vendored, modified
 * g2o that LocalBA actually executes (all paths relative to Thirdparty/g2o/g2o/):
 *   types/types_six_dof_expmap.cpp:109-169 (dual-camera edge error / Jacobians),
 *   types/se3quat.h:41-296 (SE3Quat: map, operator*, exp, normalizeRotation),
 *   core/base_binary_edge.hpp:55-120 (quadratic form), core/robust_kernel_impl.cpp:78-91 (Huber),
 *   core/block_solver.hpp:354-604 (Schur solve, lambda handling),
 *   core/optimization_algorithm_levenberg.cpp:61-189 (LM control flow),
 *   core/sparse_optimizer.cpp:61-114,166-267,354-435 (errors, index mapping, optimize loop),
 *   solvers/linear_solver_eigen.h:94-124 (LDL^T; here dense, natural order -- Eigen's sparse
 *   SimplicialLDLT with AMD ordering differs from it by rounding only).
 * Eigen is not vendored by the reference: PARITY UNPINNED; quaternion/matrix helpers restate
 * Eigen 3's published algorithms. Canonical edge order = caller's order (Q4).
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

/* RobustKernelHuber::robustify with dsqr = delta * delta as setDelta leaves it (robust_kernel_impl.cpp:65-69, 78-91) */
extern "C" void orc_robust_huber(double delta, double e, double rho[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sqrte = std::sqrt(e); rho[0] = 2 * sqrte * delta - dsqr; rho[1] = delta / sqrte; rho[2] = -0.5 * rho[1] / e; }
}

namespace {

struct Quat { double x, y, z, w; };
struct Pose { double t[3]; Quat q; };

inline void cross(const double a[3], const double b[3], double o[3])
{ o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }

/* Eigen QuaternionBase::_transformVector */
inline void qrot(const Quat& q, const double v[3], double o[3])
{
    const double u[3] = {q.x, q.y, q.z};
    double uv[3]; cross(u, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3]; cross(u, uv, c);
    o[0] = v[0] + q.w * uv[0] + c[0]; o[1] = v[1] + q.w * uv[1] + c[1]; o[2] = v[2] + q.w * uv[2] + c[2];
}
inline Quat qmul(const Quat& a, const Quat& b)
{
    return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
             a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
}
/* SE3Quat::normalizeRotation (se3quat.h:278-283) */
inline void qnormalize(Quat& q)
{
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
/* Eigen QuaternionBase::toRotationMatrix, row-major 3x3 */
inline void qtoR(const Quat& q, double R[9])
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen quaternion-from-rotation-matrix (Shoemake), row-major R */
inline Quat qfromR(const double R[9])
{
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
inline void pose_map(const Pose& T, const double X[3], double o[3])
{ qrot(T.q, X, o); o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2]; }

/* SE3Quat::operator* (se3quat.h:101-107) */
inline Pose pose_mul(const Pose& a, const Pose& b)
{
    Pose r = a;
    double rt[3]; qrot(a.q, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    r.q = qmul(a.q, b.q);
    qnormalize(r.q);
    return r;
}
inline void mat3mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
/* SE3Quat::exp (se3quat.h:223-257), update = [omega, upsilon] */
inline Pose pose_exp(const double u[6])
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9]; mat3mul(O, O, O2);
    double R[9], V[9];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + a * O[i] + b * O2[i]; V[i] = I[i] + b * O[i] + c * O2[i]; }
    }
    Pose T;
    T.q = qfromR(R);
    for (int i = 0; i < 3; ++i) T.t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    qnormalize(T.q);
    return T;
}

struct Cam { double fx, fy, cx, cy; Pose ext; double adj[36]; };

inline Pose pose_from7(const double p[7]) { Pose T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q = {p[3], p[4], p[5], p[6]}; return T; }
inline void pose_to7(const Pose& T, double p[7]) { p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2]; p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w; }
inline Cam cam_from(const orc_ba_camera& c)
{ Cam k; k.fx = c.fx; k.fy = c.fy; k.cx = c.cx; k.cy = c.cy; k.ext = pose_from7(c.ext); memcpy(k.adj, c.adj, sizeof(k.adj)); return k; }

/* EdgeSE3ProjectXYZ::computeError (types_six_dof_expmap.cpp:109-114, 163-169) */
inline void edge_error(const Pose& T, const double X[3], const Cam& c, const double obs[2], double e[2], double* z)
{
    double pm[3], pc[3];
    pose_map(T, X, pm); pose_map(c.ext, pm, pc);
    e[0] = obs[0] - (pc[0] / pc[2] * c.fx + c.cx);
    e[1] = obs[1] - (pc[1] / pc[2] * c.fy + c.cy);
    *z = pc[2];
}

/* EdgeSE3ProjectXYZ::linearizeOplus (types_six_dof_expmap.cpp:123-161) */
inline void edge_jacobian(const Pose& T, const double X[3], const Cam& c, double Jp[12], double Jx[6])
{
    double pm[3], pc[3];
    pose_map(T, X, pm); pose_map(c.ext, pm, pc);
    const double x = pc[0], y = pc[1], z = pc[2];
    const double tmp[6] = {c.fx, 0, -x / z * c.fx, 0, c.fy, -y / z * c.fy};
    const double J3[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    const double s = -1. / z;
    /* Eigen evaluates ((-1/z * tmp) * J3) * Adj left to right */
    double st[6]; for (int i = 0; i < 6; ++i) st[i] = s * tmp[i];
    double A[12];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j)
        A[i * 6 + j] = st[i * 3] * J3[j] + st[i * 3 + 1] * J3[6 + j] + st[i * 3 + 2] * J3[12 + j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) {
        double acc = 0;
        for (int k = 0; k < 6; ++k) acc += A[i * 6 + k] * c.adj[k * 6 + j];
        Jp[i * 6 + j] = acc;
    }
    const Pose TT = pose_mul(c.ext, T);
    double R[9]; qtoR(TT.q, R);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j)
        Jx[i * 3 + j] = st[i * 3] * R[j] + st[i * 3 + 1] * R[3 + j] + st[i * 3 + 2] * R[6 + j];
}

/* Matrix3d::inverse (Eigen cofactor formula); returns false when det == 0 */
inline void inv3(const double m[9], double o[9])
{
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c10 = m[5] * m[6] - m[3] * m[8];
    const double c20 = m[3] * m[7] - m[4] * m[6];
    const double det = c00 * m[0] + c10 * m[1] + c20 * m[2];
    const double id = 1.0 / det;
    o[0] = c00 * id; o[3] = c10 * id; o[6] = c20 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* dense LDL^T on the upper triangle (row-major n x n), in place; fails on an exactly-zero pivot
   like Eigen's SimplicialLDLT (linear_solver_eigen.h:105). Solves A x = b. */
bool ldlt_solve(std::vector<double>& A, int n, const double* b, double* x)
{
    /* work on L = lower (we mirror upper into lower first) */
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i];
    std::vector<double> d(n);
    for (int j = 0; j < n; ++j) {
        double dj = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) dj -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[k];
        d[j] = dj;
        if (dj == 0.0 || !std::isfinite(dj)) return false;
        for (int i = j + 1; i < n; ++i) {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[k];
            A[(size_t)i * n + j] = v / dj;
        }
    }
    for (int i = 0; i < n; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * x[k];
        x[i] = v;
    }
    for (int i = 0; i < n; ++i) x[i] /= d[i];
    for (int i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * x[k];
        x[i] = v;
    }
    return true;
}

struct Solver {
    const orc_ba_problem* pb;
    const volatile uint8_t* stop;
    std::vector<Pose> poses;
    std::vector<double> points;           /* 3L */
    std::vector<Cam> cams;
    std::vector<uint8_t> level1;          /* per edge */
    std::vector<double> err;              /* 2E, last computeActiveErrors */
    bool robust = true;
    /* per round */
    std::vector<int> act;                 /* active edge ids, ascending (EdgeIDCompare) */
    std::vector<int> pose_idx, point_idx; /* hessian index or -1 */
    std::vector<int> idx_pose, idx_point; /* inverse maps */
    std::vector<std::vector<int>> point_edges;   /* per point index: active edges with a free pose, sorted by pose index */
    int np = 0, nl = 0;
    std::vector<double> Hpp, Hll, Hpl, b, x, S, coeff, bsch, Dinv;
    std::vector<int> hpl_of_edge;
    double lambda = -1, ni = 2;
    int nBad = 0;

    bool terminate() const { return stop ? (*stop != 0) : false; }

    void compute_errors()
    {
        for (int e : act) {
            double z;
            edge_error(poses[pb->edge_pose[e]], &points[3 * pb->edge_point[e]], cams[pb->edge_cam[e]],
                       pb->obs + 2 * e, &err[2 * e], &z);
        }
    }
    double chi2(int e) const { const double w = pb->inv_sigma2[e]; return err[2 * e] * (w * err[2 * e]) + err[2 * e + 1] * (w * err[2 * e + 1]); }
    void huber(double e2, double rho[3]) const { orc_robust_huber(pb->huber_delta, e2, rho); }
    double robust_chi2() const
    {
        double chi = 0.0, rho[3];
        for (int e : act) { if (robust) { huber(chi2(e), rho); chi += rho[0]; } else chi += chi2(e); }
        return chi;
    }

    /* SparseOptimizer::initializeOptimization(level 0) + buildIndexMapping + buildStructure */
    bool init_round()
    {
        const int P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
        act.clear();
        std::vector<uint8_t> pa(P, 0), la(L, 0);
        for (int e = 0; e < E; ++e) if (!level1[e]) { act.push_back(e); pa[pb->edge_pose[e]] = 1; la[pb->edge_point[e]] = 1; }
        if (act.empty()) return false;
        pose_idx.assign(P, -1); point_idx.assign(L, -1); idx_pose.clear(); idx_point.clear();
        for (int p = 0; p < P; ++p) if (pa[p] && !pb->pose_fixed[p]) { pose_idx[p] = (int)idx_pose.size(); idx_pose.push_back(p); }
        for (int l = 0; l < L; ++l) if (la[l]) { point_idx[l] = (int)idx_point.size(); idx_point.push_back(l); }
        np = (int)idx_pose.size(); nl = (int)idx_point.size();
        point_edges.assign(nl, {});
        hpl_of_edge.assign(E, -1);
        int nhpl = 0;
        for (int e : act) if (pose_idx[pb->edge_pose[e]] >= 0) { hpl_of_edge[e] = nhpl++; point_edges[point_idx[pb->edge_point[e]]].push_back(e); }
        for (auto& v : point_edges)
            std::stable_sort(v.begin(), v.end(), [&](int a, int c) { return pose_idx[pb->edge_pose[a]] < pose_idx[pb->edge_pose[c]]; });
        Hpp.assign((size_t)np * 36, 0); Hll.assign((size_t)nl * 9, 0); Hpl.assign((size_t)nhpl * 18, 0);
        b.assign((size_t)np * 6 + (size_t)nl * 3, 0); x.assign(b.size(), 0);
        S.assign((size_t)np * 6 * np * 6, 0); coeff.assign((size_t)np * 6, 0); bsch.assign((size_t)np * 6, 0);
        Dinv.assign((size_t)nl * 9, 0);
        return true;
    }

    /* BlockSolver::buildSystem (block_solver.hpp:502-560) + constructQuadraticForm */
    void build_system()
    {
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(Hpl.begin(), Hpl.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
        for (int e : act) {
            const int p = pb->edge_pose[e], l = pb->edge_point[e];
            double Jp[12], Jx[6];
            edge_jacobian(poses[p], &points[3 * l], cams[pb->edge_cam[e]], Jp, Jx);
            double w = pb->inv_sigma2[e];
            double r[2] = {-w * err[2 * e], -w * err[2 * e + 1]};          /* omega_r = -Omega e */
            if (robust) {
                double rho[3]; huber(chi2(e), rho);
                r[0] *= rho[1]; r[1] *= rho[1];
                w = rho[1] * w;                                            /* robustInformation */
            }
            const int li = point_idx[l], pi = pose_idx[p];
            double* bl = &b[(size_t)np * 6 + (size_t)li * 3];
            double* hl = &Hll[(size_t)li * 9];
            for (int i = 0; i < 3; ++i) {
                bl[i] += Jx[i] * r[0] + Jx[3 + i] * r[1];
                for (int j = 0; j < 3; ++j) hl[i * 3 + j] += Jx[i] * w * Jx[j] + Jx[3 + i] * w * Jx[3 + j];
            }
            if (pi >= 0) {
                double* bp = &b[(size_t)pi * 6];
                double* hp = &Hpp[(size_t)pi * 36];
                double* hpl = &Hpl[(size_t)hpl_of_edge[e] * 18];           /* 6x3 = Jp^T W Jx */
                for (int i = 0; i < 6; ++i) {
                    bp[i] += Jp[i] * r[0] + Jp[6 + i] * r[1];
                    for (int j = 0; j < 6; ++j) hp[i * 6 + j] += Jp[i] * w * Jp[j] + Jp[6 + i] * w * Jp[6 + j];
                    for (int j = 0; j < 3; ++j) hpl[i * 3 + j] += Jp[i] * w * Jx[j] + Jp[6 + i] * w * Jx[3 + j];
                }
            }
        }
    }

    /* BlockSolver::solve, Schur branch (block_solver.hpp:354-486); lam already on the diagonals */
    bool solve()
    {
        const int n = np * 6;
        std::fill(S.begin(), S.end(), 0.0);
        for (int i = 0; i < np; ++i) for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
            S[(size_t)(i * 6 + r) * n + i * 6 + c] = Hpp[(size_t)i * 36 + r * 6 + c];
        std::fill(coeff.begin(), coeff.end(), 0.0);
        for (int li = 0; li < nl; ++li) {
            double* Di = &Dinv[(size_t)li * 9];
            inv3(&Hll[(size_t)li * 9], Di);
            const double* bl = &b[(size_t)n + (size_t)li * 3];
            double db[3];
            for (int i = 0; i < 3; ++i) db[i] = Di[i * 3] * bl[0] + Di[i * 3 + 1] * bl[1] + Di[i * 3 + 2] * bl[2];
            const std::vector<int>& col = point_edges[li];
            for (size_t a = 0; a < col.size(); ++a) {
                const int i1 = pose_idx[pb->edge_pose[col[a]]];
                const double* Bi = &Hpl[(size_t)hpl_of_edge[col[a]] * 18];
                double BD[18];
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c)
                    BD[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
                for (int r = 0; r < 6; ++r) coeff[(size_t)i1 * 6 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
                for (size_t c2 = a; c2 < col.size(); ++c2) {
                    const int i2 = pose_idx[pb->edge_pose[col[c2]]];
                    const double* Bj = &Hpl[(size_t)hpl_of_edge[col[c2]] * 18];
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
                        S[(size_t)(i1 * 6 + r) * n + i2 * 6 + c] -= BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
                }
            }
        }
        for (int i = 0; i < n; ++i) bsch[i] = b[i] - coeff[i];
        if (n > 0 && !ldlt_solve(S, n, bsch.data(), x.data())) return false;
        for (int li = 0; li < nl; ++li) {
            double cl[3] = {b[(size_t)n + li * 3], b[(size_t)n + li * 3 + 1], b[(size_t)n + li * 3 + 2]};
            for (int e : point_edges[li]) {
                const int i1 = pose_idx[pb->edge_pose[e]];
                const double* B = &Hpl[(size_t)hpl_of_edge[e] * 18];
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * x[(size_t)i1 * 6 + r];
            }
            const double* Di = &Dinv[(size_t)li * 9];
            for (int i = 0; i < 3; ++i) x[(size_t)n + li * 3 + i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
        }
        return true;
    }

    void add_lambda(double lam)
    {
        for (int i = 0; i < np; ++i) for (int d = 0; d < 6; ++d) Hpp[(size_t)i * 36 + d * 7] += lam;
        for (int i = 0; i < nl; ++i) for (int d = 0; d < 3; ++d) Hll[(size_t)i * 9 + d * 4] += lam;
    }

    /* OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164).
       returns 0 = OK, 1 = Terminate */
    int lm_iteration(int iteration, int* trials)
    {
        compute_errors();
        double currentChi = robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        build_system();
        if (iteration == 0) {
            double maxDiag = 0;                                   /* computeLambdaInit, tau = 1e-5 */
            for (int i = 0; i < np; ++i) for (int d = 0; d < 6; ++d) maxDiag = std::max(std::fabs(Hpp[(size_t)i * 36 + d * 7]), maxDiag);
            for (int i = 0; i < nl; ++i) for (int d = 0; d < 3; ++d) maxDiag = std::max(std::fabs(Hll[(size_t)i * 9 + d * 4]), maxDiag);
            lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        std::vector<Pose> bk_poses; std::vector<double> bk_points, dp(6 * (size_t)np), dl(3 * (size_t)nl);
        do {
            bk_poses = poses; bk_points = points;                 /* push */
            for (int i = 0; i < np; ++i) for (int d = 0; d < 6; ++d) dp[(size_t)i * 6 + d] = Hpp[(size_t)i * 36 + d * 7];
            for (int i = 0; i < nl; ++i) for (int d = 0; d < 3; ++d) dl[(size_t)i * 3 + d] = Hll[(size_t)i * 9 + d * 4];
            add_lambda(lambda);
            const bool ok2 = solve();
            ++*trials;
            for (int i = 0; i < np; ++i) poses[idx_pose[i]] = pose_mul(pose_exp(&x[(size_t)i * 6]), poses[idx_pose[i]]);
            for (int i = 0; i < nl; ++i) for (int d = 0; d < 3; ++d) points[(size_t)idx_point[i] * 3 + d] += x[(size_t)np * 6 + (size_t)i * 3 + d];
            for (int i = 0; i < np; ++i) for (int d = 0; d < 6; ++d) Hpp[(size_t)i * 36 + d * 7] = dp[(size_t)i * 6 + d];
            for (int i = 0; i < nl; ++i) for (int d = 0; d < 3; ++d) Hll[(size_t)i * 9 + d * 4] = dl[(size_t)i * 3 + d];
            compute_errors();
            tempChi = robust_chi2();
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;                                     /* computeScale */
            for (size_t j = 0; j < x.size(); ++j) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double sf = std::max(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                poses = bk_poses; points = bk_points;             /* pop (errors stay stale, as in g2o) */
            }
            ++qmax;
        } while (rho < 0 && qmax < 10 && !terminate());
        last_chi = currentChi;
        if (qmax == 10 || rho == 0) return 1;
        if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0;
        if (nBad >= 3) return 1;
        return 0;
    }
    double last_chi = 0;
};

}  // namespace

extern "C" {

int orc_ba_local(const orc_ba_problem* pb, const volatile uint8_t* stop_flag, orc_ba_result* res)
{
    Solver s;
    s.pb = pb; s.stop = stop_flag;
    const int P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
    s.poses.resize(P);
    for (int p = 0; p < P; ++p) { s.poses[p] = pose_from7(pb->poses + 7 * p); }
    s.points.assign(pb->points, pb->points + 3 * (size_t)L);
    s.cams.resize(pb->n_cams);
    for (int c = 0; c < pb->n_cams; ++c) s.cams[c] = cam_from(pb->cams[c]);
    s.level1.assign(E, 0);
    s.err.assign(2 * (size_t)E, 0.0);
    res->n_iters[0] = res->n_iters[1] = 0; res->n_trials[0] = res->n_trials[1] = 0;
    res->lambda[0] = res->lambda[1] = 0;
    for (int i = 0; i < 32; ++i) res->chi2_trace[i] = 0;
    int trace = 0;

    auto run = [&](int round, int iters) {
        if (!s.init_round()) return;
        for (int i = 0; i < iters && !s.terminate(); ++i) {
            const int r = s.lm_iteration(i, &res->n_trials[round]);
            ++res->n_iters[round];
            if (trace < 32) res->chi2_trace[trace++] = s.last_chi;
            if (r != 0) break;
        }
        res->lambda[round] = s.lambda;
    };

    if (!(stop_flag && *stop_flag)) {                              /* Optimizer.cc:582-584 */
        s.robust = pb->huber_delta > 0.0;                          /* BundleAdjustment(bRobust = false): no kernel (:137-142) */
        run(0, pb->iters1);
        bool more = !(stop_flag && *stop_flag);                    /* :589-593 */
        if (more) {
            for (int e = 0; e < E; ++e) {                          /* :599-613 */
                double z, tmp[2];
                edge_error(s.poses[pb->edge_pose[e]], &s.points[3 * pb->edge_point[e]], s.cams[pb->edge_cam[e]], pb->obs + 2 * e, tmp, &z);
                if (s.chi2(e) > pb->chi2_th || !(z > 0.0)) s.level1[e] = 1;
            }
            s.robust = false;
            run(1, pb->iters2);
        }
    }
    for (int e = 0; e < E; ++e) {                                  /* :645-658 */
        double z, tmp[2];
        edge_error(s.poses[pb->edge_pose[e]], &s.points[3 * pb->edge_point[e]], s.cams[pb->edge_cam[e]], pb->obs + 2 * e, tmp, &z);
        res->edge_chi2[e] = s.chi2(e);
        res->edge_outlier[e] = (s.chi2(e) > pb->chi2_th || !(z > 0.0)) ? 1 : 0;
        res->edge_level1[e] = s.level1[e];
    }
    for (int p = 0; p < P; ++p) pose_to7(s.poses[p], res->poses + 7 * p);
    memcpy(res->points, s.points.data(), sizeof(double) * 3 * (size_t)L);
    return 0;
}

/* Parity tap: the blocks of the first linearisation of orc_ba_local -- initializeOptimization, computeActiveErrors, then buildSystem =
   linearizeOplus (types_six_dof_expmap.cpp:123-161) + constructQuadraticForm with robustInformation (base_binary_edge.hpp:55-120,
   base_edge.h:96-102) at the initial estimates; no lambda, no solve. Outputs by GLOBAL ids: Hpp / bp in free-pose order with
   pose_idx[p] = index or -1, Hll / bl per point, Hpl per edge (6 x 3 row-major, zero where the edge's pose is fixed). */
int orc_ba_linearize(const orc_ba_problem* pb, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, int32_t* pose_idx, int* n_free)
{
    Solver s;
    s.pb = pb; s.stop = nullptr;
    const int P = pb->n_poses, L = pb->n_points, E = pb->n_edges;
    s.poses.resize(P);
    for (int p = 0; p < P; ++p) s.poses[p] = pose_from7(pb->poses + 7 * p);
    s.points.assign(pb->points, pb->points + 3 * (size_t)L);
    s.cams.resize(pb->n_cams);
    for (int c = 0; c < pb->n_cams; ++c) s.cams[c] = cam_from(pb->cams[c]);
    s.level1.assign(E, 0);
    s.err.assign(2 * (size_t)E, 0.0);
    s.robust = pb->huber_delta > 0.0;
    if (!s.init_round()) return -1;
    s.compute_errors();
    s.build_system();
    *n_free = s.np;
    for (int p = 0; p < P; ++p) pose_idx[p] = s.pose_idx[p];
    memcpy(Hpp, s.Hpp.data(), sizeof(double) * 36 * (size_t)s.np);
    memcpy(bp, s.b.data(), sizeof(double) * 6 * (size_t)s.np);
    memset(Hll, 0, sizeof(double) * 9 * (size_t)L); memset(bl, 0, sizeof(double) * 3 * (size_t)L);
    for (int li = 0; li < s.nl; ++li) {
        const int l = s.idx_point[li];
        memcpy(Hll + 9 * (size_t)l, &s.Hll[(size_t)li * 9], sizeof(double) * 9);
        memcpy(bl + 3 * (size_t)l, &s.b[(size_t)s.np * 6 + (size_t)li * 3], sizeof(double) * 3);
    }
    memset(Hpl, 0, sizeof(double) * 18 * (size_t)E);
    for (int e = 0; e < E; ++e) if (s.hpl_of_edge[e] >= 0) memcpy(Hpl + 18 * (size_t)e, &s.Hpl[(size_t)s.hpl_of_edge[e] * 18], sizeof(double) * 18);
    return 0;
}

/* Optimizer::PoseOptimization (Optimizer.cc:250-405). Per frame: vertex = pose, unary edges; the LM driver is the same
   optimization_algorithm_levenberg.cpp:61-164 as in orc_ba_local with a single 6x6 block and no Schur complement
   (BlockSolver::solve non-Schur branch, block_solver.hpp:354-372; LinearSolverDense = dense LDL^T, isPositive check). */
int orc_pose_optimization(const orc_pose_problem* pb, orc_pose_result* res)
{
    std::vector<Cam> cams(pb->n_cams);
    for (int c = 0; c < pb->n_cams; ++c) cams[c] = cam_from(pb->cams[c]);
    for (int f = 0; f < pb->n_frames; ++f) {
        const int e0 = pb->edge_off[f], n = pb->edge_off[f + 1] - e0;
        const Pose init = pose_from7(pb->poses + 7 * f);
        Pose T = init;
        for (int k = 0; k < n; ++k) res->outlier[e0 + k] = 0;                       /* :300 */
        if (res->n_iters) for (int r = 0; r < 4; ++r) res->n_iters[4 * f + r] = 0;
        if (n < 3) {                                                                  /* :343-344: pose untouched, returns 0 */
            res->n_inliers[f] = 0;
            pose_to7(T, res->poses + 7 * f);
            if (res->edge_chi2) for (int k = 0; k < n; ++k) res->edge_chi2[e0 + k] = 0;
            continue;
        }
        std::vector<uint8_t> level(n, 0);
        std::vector<double> err(2 * (size_t)n, 0.0);                                   /* _error of every edge */
        bool robust = true;                                                            /* kernels removed after round 3 (:388-389) */
        auto chi2_of = [&](int k) { const double w = pb->inv_sigma2[e0 + k]; return err[2 * k] * (w * err[2 * k]) + err[2 * k + 1] * (w * err[2 * k + 1]); };
        auto huber = [&](double e2, double rho[3]) { orc_robust_huber(pb->huber_delta, e2, rho); };
        int nBadEdges = 0;
        for (int it = 0; it < 4; ++it) {
            T = init;                                                                  /* :360 */
            std::vector<int> act;
            for (int k = 0; k < n; ++k) if (!level[k]) act.push_back(k);              /* initializeOptimization(0) */
            auto compute_errors = [&]() {
                for (int k : act) { double z; edge_error(T, pb->xw + 3 * (size_t)(e0 + k), cams[pb->edge_cam[e0 + k]], pb->obs + 2 * (size_t)(e0 + k), &err[2 * k], &z); }
            };
            auto robust_chi2 = [&]() {
                double chi = 0, rho[3];
                for (int k : act) { if (robust) { huber(chi2_of(k), rho); chi += rho[0]; } else chi += chi2_of(k); }
                return chi;
            };
            double lambda = -1, ni = 2;
            int nBad = 0;
            if (!act.empty()) {                                                        /* optimize(): no active vertex -> nothing happens */
                for (int i = 0; i < pb->its[it]; ++i) {
                    compute_errors();
                    double currentChi = robust_chi2(), tempChi = currentChi;
                    const double iniChi = currentChi;
                    double H[36], b[6], x[6];
                    for (double& v : H) v = 0;
                    for (double& v : b) v = 0;
                    for (int k : act) {                                                /* linearizeOplus + constructQuadraticForm (base_unary_edge.hpp) */
                        double Jp[12], Jx[6];
                        edge_jacobian(T, pb->xw + 3 * (size_t)(e0 + k), cams[pb->edge_cam[e0 + k]], Jp, Jx);
                        double w = pb->inv_sigma2[e0 + k];
                        double r[2] = {-w * err[2 * k], -w * err[2 * k + 1]};
                        if (robust) { double rho[3]; huber(chi2_of(k), rho); r[0] *= rho[1]; r[1] *= rho[1]; w = rho[1] * w; }
                        for (int a = 0; a < 6; ++a) {
                            b[a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
                            for (int c = 0; c < 6; ++c) H[a * 6 + c] += Jp[a] * w * Jp[c] + Jp[6 + a] * w * Jp[6 + c];
                        }
                    }
                    if (i == 0) {                                                      /* computeLambdaInit */
                        double maxDiag = 0;
                        for (int d = 0; d < 6; ++d) maxDiag = std::max(std::fabs(H[d * 7]), maxDiag);
                        lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
                    }
                    double rho = 0;
                    int qmax = 0;
                    do {
                        const Pose bk = T;                                             /* push */
                        std::vector<double> A(H, H + 36);
                        for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
                        const bool ok2 = ldlt_solve(A, 6, b, x);
                        T = pose_mul(pose_exp(x), T);
                        compute_errors();
                        tempChi = robust_chi2();
                        if (!ok2) tempChi = DBL_MAX;
                        rho = currentChi - tempChi;
                        double scale = 0;
                        for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
                        scale += 1e-3;
                        rho /= scale;
                        if (rho > 0 && std::isfinite(tempChi)) {
                            double alpha = 1. - std::pow((2 * rho - 1), 3);
                            alpha = std::min(alpha, 2. / 3.);
                            lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                        } else { lambda *= ni; ni *= 2; T = bk; }                      /* pop: errors stay stale, as in g2o */
                        ++qmax;
                    } while (rho < 0 && qmax < 10);
                    if (res->n_iters) ++res->n_iters[4 * f + it];
                    if (qmax == 10 || rho == 0) break;
                    if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0;
                    if (nBad >= 3) break;
                }
            }
            nBadEdges = 0;
            for (int k = 0; k < n; ++k) {                                              /* :365-390 */
                if (res->outlier[e0 + k]) { double z; edge_error(T, pb->xw + 3 * (size_t)(e0 + k), cams[pb->edge_cam[e0 + k]], pb->obs + 2 * (size_t)(e0 + k), &err[2 * k], &z); }
                const float chi2 = (float)chi2_of(k);
                if (chi2 > pb->chi2_th[it]) { res->outlier[e0 + k] = 1; level[k] = 1; ++nBadEdges; }
                else { res->outlier[e0 + k] = 0; level[k] = 0; }
            }
            if (it == 2) robust = false;
            if (n < 10) break;                                                         /* optimizer.edges().size() < 10 (:392) */
        }
        pose_to7(T, res->poses + 7 * f);
        res->n_inliers[f] = n - nBadEdges;
        if (res->edge_chi2) for (int k = 0; k < n; ++k) res->edge_chi2[e0 + k] = chi2_of(k);
    }
    return 0;
}

void orc_ba_edge_error(const double pose[7], const double point[3], const orc_ba_camera* cam,
                       const double obs[2], double err[2], double* depth)
{ edge_error(pose_from7(pose), point, cam_from(*cam), obs, err, depth); }

void orc_ba_edge_jacobian(const double pose[7], const double point[3], const orc_ba_camera* cam,
                          double J_pose[12], double J_point[6])
{ edge_jacobian(pose_from7(pose), point, cam_from(*cam), J_pose, J_point); }

void orc_se3_oplus(const double pose_in[7], const double update[6], double pose_out[7])
{ pose_to7(pose_mul(pose_exp(update), pose_from7(pose_in)), pose_out); }

/* Cameras::setExtrinsics (Cameras.cc:17-37) + Converter::toSE3Quat/toMatrix6d (Converter.cc:58-68,
   104-112): float 4x4 -> Adj = [[R, R*t^],[0, R]] computed in float then widened (Q1: the
   reference leaves the lower-left block uninitialised; we use 0). exact != 0 gives g2o's own
   SE3Quat::adj() = [[R,0],[t^R,R]] (se3quat.h:259-268) in double. */
void orc_rig_adjoint(const float T[16], int exact, double adj[36], double ext7[7])
{
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
    Quat q = qfromR(R); qnormalize(q);
    ext7[0] = t[0]; ext7[1] = t[1]; ext7[2] = t[2]; ext7[3] = q.x; ext7[4] = q.y; ext7[5] = q.z; ext7[6] = q.w;
    for (int i = 0; i < 36; ++i) adj[i] = 0;
    if (!exact) {
        const float th[9] = {0, -T[11], T[7], T[11], 0, -T[3], -T[7], T[3], 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            float acc = 0;                                      /* cv::Mat float GEMM, k-ordered */
            for (int k = 0; k < 3; ++k) acc += T[i * 4 + k] * th[k * 3 + j];
            adj[i * 6 + j] = T[i * 4 + j]; adj[(i + 3) * 6 + j + 3] = T[i * 4 + j];
            adj[i * 6 + j + 3] = acc;
        }
    } else {
        double Rn[9]; qtoR(q, Rn);
        const double th[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double tR[9]; mat3mul(th, Rn, tR);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            adj[i * 6 + j] = Rn[i * 3 + j]; adj[(i + 3) * 6 + j + 3] = Rn[i * 3 + j];
            adj[(i + 3) * 6 + j] = tR[i * 3 + j];
        }
    }
}

}  // extern "C"
