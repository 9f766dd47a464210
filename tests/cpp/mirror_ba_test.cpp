// Drives the C++ mirrors of the optimiser and of the projection-guided matcher (orb-slam2-dualcam_amd/host/Optimizer.h,
// ORBmatcher.h) the way LocalMapping::Run -> Optimizer::LocalBundleAdjustment (src/LocalMapping.cc:97-104),
// Tracking -> Optimizer::PoseOptimization (src/Tracking.cc:1321) and Tracking::SearchLocalPoints -> isInFrustum +
// ORBmatcher::SearchByProjection (src/Tracking.cc:1617-1680) do, on flat problems written by tests/test_gpu_cpp_mirror.py.
// Prints checksums; the Python side compares them with the ctypes path and the oracle.
// blob format: repeated { u32 name_len, name, u64 n_bytes, data }.  usage: mirror_ba_test problem.blob
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ORBmatcher.h"
#include "Optimizer.h"

static uint64_t fnv(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <typename T> static uint64_t fnv(const std::vector<T>& v) { return fnv(v.data(), v.size() * sizeof(T)); }

struct Blob {
    std::map<std::string, std::vector<char>> a;
    bool load(const char* path)
    {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        for (;;) {
            uint32_t nl; uint64_t nb;
            if (fread(&nl, 4, 1, f) != 1) break;
            std::string name(nl, ' ');
            if (fread(&name[0], 1, nl, f) != nl || fread(&nb, 8, 1, f) != 1) return false;
            std::vector<char> d(nb);
            if (nb && fread(d.data(), 1, nb, f) != nb) return false;
            a[name] = std::move(d);
        }
        fclose(f);
        return true;
    }
    template <typename T> std::vector<T> get(const std::string& n) const
    {
        const std::vector<char>& d = a.at(n);
        std::vector<T> v(d.size() / sizeof(T));
        if (!d.empty()) memcpy(v.data(), d.data(), d.size());
        return v;
    }
};

int main(int argc, char** argv)
{
    Blob b;
    if (argc < 2 || !b.load(argv[1])) return 2;
    try {
        using namespace ORB_SLAM2;
        const std::vector<dcs_ba_camera> cams = b.get<dcs_ba_camera>("cams");
        {   // ---- LocalBundleAdjustment, then the same call with the stop flag already raised
            LocalBAProblem p;
            p.poses = b.get<double>("ba.poses"); p.poseFixed = b.get<uint8_t>("ba.fixed"); p.points = b.get<double>("ba.points");
            p.edgePose = b.get<int32_t>("ba.edge_pose"); p.edgePoint = b.get<int32_t>("ba.edge_point"); p.edgeCam = b.get<int32_t>("ba.edge_cam");
            p.obs = b.get<double>("ba.obs"); p.invSigma2 = b.get<double>("ba.inv_sigma2"); p.cams = cams;
            LocalBAResult r;
            bool stop = false;
            Optimizer::LocalBundleAdjustment(p, &stop, r);
            printf("lba %d %d %016llx %016llx %016llx\n", r.iterations[0], r.iterations[1], (unsigned long long)fnv(r.poses),
                   (unsigned long long)fnv(r.points), (unsigned long long)fnv(r.edgeOutlier));
            stop = true;
            LocalBAResult r2;
            Optimizer::LocalBundleAdjustment(p, &stop, r2);
            printf("lba_stopped %d %d %d\n", r2.iterations[0], r2.iterations[1], (int)(r2.poses == p.poses && r2.points == p.points));
            LocalBAResult r3;
            Optimizer::BundleAdjustment(p, 5, nullptr, true, r3);
            printf("gba %d %d %016llx\n", r3.iterations[0], r3.iterations[1], (unsigned long long)fnv(r3.poses));
        }
        {   // ---- PoseOptimization
            PoseProblem p;
            p.poses = b.get<double>("po.poses"); p.edgeOff = b.get<int32_t>("po.edge_off"); p.xw = b.get<double>("po.xw"); p.obs = b.get<double>("po.obs");
            p.invSigma2 = b.get<double>("po.inv_sigma2"); p.edgeCam = b.get<int32_t>("po.edge_cam"); p.cams = cams;
            std::vector<double> out;
            std::vector<uint8_t> outl;
            const std::vector<int> inl = Optimizer::PoseOptimization(p, out, outl);
            printf("po %zu %016llx %016llx %016llx\n", inl.size(), (unsigned long long)fnv(inl), (unsigned long long)fnv(out), (unsigned long long)fnv(outl));
        }
        {   // ---- SearchByProjection / SearchByProjectionOnCam on a frame grid built with dcs_frame_grid
            const std::vector<int32_t> cam_off = b.get<int32_t>("pr.cam_off"), oct = b.get<int32_t>("pr.kp_octave");
            const std::vector<float> kx = b.get<float>("pr.kp_x"), ky = b.get<float>("pr.kp_y"), ka = b.get<float>("pr.kp_angle");
            const std::vector<uint8_t> desc = b.get<uint8_t>("pr.desc"), taken = b.get<uint8_t>("pr.taken");
            const std::vector<float> mnx = b.get<float>("pr.min_x"), mny = b.get<float>("pr.min_y"), wi = b.get<float>("pr.grid_w_inv"), hi = b.get<float>("pr.grid_h_inv");
            const int C = (int)cam_off.size() - 1, N = cam_off[C];
            std::vector<int32_t> goff((size_t)C * DCS_GRID_COLS * DCS_GRID_ROWS + 1), gidx(N > 0 ? N : 1);
            int n_entries = 0;
            if (dcs_frame_grid(C, cam_off.data(), kx.data(), ky.data(), mnx.data(), mny.data(), wi.data(), hi.data(), goff.data(), gidx.data(), &n_entries) != DCS_OK)
                throw std::runtime_error(dcs_last_error());
            dcs_proj_frame fr{};
            fr.n_cams = C; fr.cam_off = cam_off.data(); fr.kp_x = kx.data(); fr.kp_y = ky.data(); fr.kp_octave = oct.data(); fr.kp_angle = ka.data();
            fr.desc = desc.data(); fr.taken = taken.data(); fr.min_x = mnx.data(); fr.min_y = mny.data(); fr.grid_w_inv = wi.data(); fr.grid_h_inv = hi.data();
            fr.grid_off = goff.data(); fr.grid_idx = gidx.data();
            const std::vector<uint8_t> qv = b.get<uint8_t>("q.valid"), qd = b.get<uint8_t>("q.desc");
            const std::vector<int32_t> qc = b.get<int32_t>("q.cam"), qmin = b.get<int32_t>("q.min_level"), qmax = b.get<int32_t>("q.max_level");
            const std::vector<float> qu = b.get<float>("q.u"), qvv = b.get<float>("q.v"), qr = b.get<float>("q.radius"), qa = b.get<float>("q.angle");
            dcs_proj_queries q{};
            q.n = (int)qc.size(); q.valid = qv.data(); q.cam = qc.data(); q.u = qu.data(); q.v = qvv.data(); q.radius = qr.data();
            q.min_level = qmin.data(); q.max_level = qmax.data(); q.desc = qd.data(); q.angle = qa.data();
            std::vector<int32_t> mq, qf;
            ORBmatcher m08(0.8f, true);
            const int n1 = m08.SearchByProjection(fr, q, mq, qf);
            printf("proj %d %d %016llx %016llx\n", n_entries, n1, (unsigned long long)fnv(mq), (unsigned long long)fnv(qf));
            const int n2 = m08.SearchByProjectionOnCam(fr, q, mq, qf);
            printf("proj_oncam %d %016llx %016llx\n", n2, (unsigned long long)fnv(mq), (unsigned long long)fnv(qf));
        }
        {   // ---- isInFrustum for every local map point
            dcs_frustum_frame f{};
            const std::vector<float> Rsw = b.get<float>("fr.Rsw"), tsw = b.get<float>("fr.tsw"), Ow = b.get<float>("fr.Ow"), fx = b.get<float>("fr.fx"),
                                     fy = b.get<float>("fr.fy"), cx = b.get<float>("fr.cx"), cy = b.get<float>("fr.cy"), bx0 = b.get<float>("fr.min_x"),
                                     bx1 = b.get<float>("fr.max_x"), by0 = b.get<float>("fr.min_y"), by1 = b.get<float>("fr.max_y"), sf = b.get<float>("fr.scale_factors");
            const std::vector<float> misc = b.get<float>("fr.log_scale_factor");
            f.n_cams = (int)fx.size(); f.Rsw = Rsw.data(); f.tsw = tsw.data(); f.Ow = Ow.data(); f.fx = fx.data(); f.fy = fy.data(); f.cx = cx.data(); f.cy = cy.data();
            f.min_x = bx0.data(); f.max_x = bx1.data(); f.min_y = by0.data(); f.max_y = by1.data(); f.scale_factors = sf.data(); f.n_scale_levels = (int)sf.size();
            f.log_scale_factor = misc[0];
            ORBmatcher::FrustumResult r;
            ORBmatcher::IsInFrustum(f, b.get<float>("pt.pos"), b.get<float>("pt.normal"), b.get<float>("pt.min_dist"), b.get<float>("pt.max_dist"), {}, 0.5f, 1.0f, r);
            printf("frustum %016llx %016llx %016llx %016llx %016llx\n", (unsigned long long)fnv(r.inView), (unsigned long long)fnv(r.cam),
                   (unsigned long long)fnv(r.u), (unsigned long long)fnv(r.v), (unsigned long long)fnv(r.viewCos));
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
