// One dual frame from images to pose, driven from C++ through the C ABI only -- the reference's per-frame steady state as a host would bind it
// (INTEGRATION.md section 2b''): Frame ctor (src/Frame.cc:141-196: ExtractORB per camera -> dcs_orb_extract_batch_device, slots stay in HBM),
// Tracking::TrackWithMotionModel (src/Tracking.cc:1384-1448: dcs_track_frame_device mode 1 + the outlier bookkeeping on the host) and
// TrackLocalMap's search + optimisation (src/Tracking.cc:1617-1680, 1321: mode 0). The map comes from tests/test_gpu_cpp_mirror.py
// (synth.scene_from_features). Prints counts, poses and checksums; the Python side compares them with the ctypes path bit for bit.
// blob format: repeated { u32 name_len, name, u64 n_bytes, data }.  usage: frame_pipeline_test scene.blob
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dcs_abi.h"

static uint64_t fnv(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
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
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, dcs_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    Blob b;
    if (argc < 2 || !b.load(argv[1])) return 2;
    const std::vector<int32_t> dims = b.get<int32_t>("dims");            // rows, cols, nfeatures
    const int rows = dims[0], cols = dims[1], nfeat = dims[2], C = 2;
    const std::vector<uint8_t> images = b.get<uint8_t>("images");
    // ---- Frame ctor: both cameras' images -> slots in HBM
    dcs_orb_params op = {nfeat, 1.2f, 8, 20, 7, -1, 2, 0};
    dcs_orb* orb = nullptr;
    CK(dcs_orb_create(&op, &orb));
    int cap = 0;
    CK(dcs_orb_required_cap(orb, rows, cols, &cap));
    uint8_t* d_img = nullptr; dcs_keypoint* d_kp = nullptr; uint8_t* d_desc = nullptr; int32_t* d_n = nullptr;
    hipStream_t st = nullptr;
    HK(hipStreamCreate(&st));
    HK(hipMalloc((void**)&d_img, images.size())); HK(hipMalloc((void**)&d_kp, sizeof(dcs_keypoint) * C * cap)); HK(hipMalloc((void**)&d_desc, (size_t)C * cap * 32));
    HK(hipMalloc((void**)&d_n, sizeof(int32_t) * C));
    HK(hipMemcpyAsync(d_img, images.data(), images.size(), hipMemcpyHostToDevice, st));
    CK(dcs_orb_extract_batch_device(orb, d_img, C, rows, cols, cols, d_kp, d_desc, cap, d_n, st));
    // ---- the frame as dcs_track_frame_device sees it
    const std::vector<float> Rsw = b.get<float>("Rsw"), tsw = b.get<float>("tsw"), Ow = b.get<float>("Ow"), fx = b.get<float>("fx"), fy = b.get<float>("fy"), cx = b.get<float>("cx"),
                             cy = b.get<float>("cy"), vmin_x = b.get<float>("min_x"), vmax_x = b.get<float>("max_x"), vmin_y = b.get<float>("min_y"), vmax_y = b.get<float>("max_y"),
                             scale = b.get<float>("scale_factors"), gwi = b.get<float>("grid_w_inv"), ghi = b.get<float>("grid_h_inv"), sig = b.get<float>("inv_level_sigma2");
    const std::vector<double> pose = b.get<double>("pose");
    const std::vector<dcs_ba_camera> cams = b.get<dcs_ba_camera>("cams");
    dcs_track_dev_frame f;
    memset(&f, 0, sizeof f);
    f.features.n_cams = C; f.features.cap = cap; f.features.first_slot = 0; f.features.d_kp = d_kp; f.features.d_desc = d_desc; f.features.d_n = d_n;
    f.features.min_x = vmin_x.data(); f.features.min_y = vmin_y.data(); f.features.grid_w_inv = gwi.data(); f.features.grid_h_inv = ghi.data();
    f.view.n_cams = C; f.view.Rsw = Rsw.data(); f.view.tsw = tsw.data(); f.view.Ow = Ow.data(); f.view.fx = fx.data(); f.view.fy = fy.data(); f.view.cx = cx.data(); f.view.cy = cy.data();
    f.view.min_x = vmin_x.data(); f.view.max_x = vmax_x.data(); f.view.min_y = vmin_y.data(); f.view.max_y = vmax_y.data();
    f.view.log_scale_factor = b.get<float>("log_scale_factor")[0]; f.view.n_scale_levels = (int)scale.size(); f.view.scale_factors = scale.data();
    f.pose = pose.data();
    dcs_track_params prm;
    memset(&prm, 0, sizeof prm);
    prm.viewing_cos_limit = 0.5f; prm.th = 7.f; prm.th_high = 100; prm.nn_ratio = 0.f; prm.n_levels = (int)sig.size(); prm.inv_level_sigma2 = sig.data();
    prm.n_cams = C; prm.cams = cams.data(); prm.huber_delta = b.get<double>("huber_delta")[0];
    for (int i = 0; i < 4; ++i) { prm.chi2_th[i] = 5.991f; prm.its[i] = 10; }
    // ---- TrackWithMotionModel: the last frame's features with a good map point are the queries; mvpMapPoints is empty (:1396)
    const std::vector<float> mm_pos = b.get<float>("mm.pos"), mm_ang = b.get<float>("mm.q_angle");
    const std::vector<uint8_t> mm_desc = b.get<uint8_t>("mm.desc");
    const std::vector<int32_t> mm_cam = b.get<int32_t>("mm.q_cam"), mm_oct = b.get<int32_t>("mm.q_octave"), mm_point = b.get<int32_t>("mm.point");
    const int nq = (int)mm_cam.size(), Ncap = C * cap;
    f.n_held = 0; f.n_points = nq; f.pos = mm_pos.data(); f.desc = mm_desc.data(); f.q_cam = mm_cam.data(); f.q_octave = mm_oct.data(); f.q_angle = mm_ang.data();
    std::vector<double> out_pose(7);
    std::vector<int32_t> mop((size_t)std::max(nq, 1)), pof((size_t)Ncap), nfeat_c(C);
    std::vector<uint8_t> outl((size_t)Ncap);
    int32_t n_inl = 0, n_match = 0;
    int32_t* p_mop = mop.data(); int32_t* p_pof = pof.data(); uint8_t* p_outl = outl.data();
    dcs_track_dev_result res;
    res.r.poses = out_pose.data(); res.r.n_inliers = &n_inl; res.r.n_matches = &n_match; res.r.match_of_point = &p_mop; res.r.point_of_feature = &p_pof; res.r.outlier = &p_outl;
    res.n_features = nfeat_c.data();
    CK(dcs_track_frame_device(1, &f, &prm, 1, 1, &res, st));
    const int N = nfeat_c[0] + nfeat_c[1];
    printf("features %d %d\n", nfeat_c[0], nfeat_c[1]);
    printf("mm %d %d %016llx %016llx %016llx", n_match, n_inl, (unsigned long long)fnv(mop.data(), sizeof(int32_t) * nq), (unsigned long long)fnv(pof.data(), sizeof(int32_t) * N),
           (unsigned long long)fnv(outl.data(), N));
    for (double v : out_pose) printf(" %.17g", v);
    printf("\n");
    // ---- the bookkeeping of Tracking.cc:1429-1448 (outliers lose their point) and the candidate flags of SearchLocalPoints (:1625-1640)
    const std::vector<float> lm_pos = b.get<float>("lm.pos"), lm_nrm = b.get<float>("lm.normal"), lm_min = b.get<float>("lm.min_dist"), lm_max = b.get<float>("lm.max_dist");
    const std::vector<uint8_t> lm_desc = b.get<uint8_t>("lm.desc");
    const int nl = (int)lm_min.size();
    std::vector<uint8_t> held((size_t)N, 0), cand((size_t)nl, 1);
    std::vector<float> held_xw((size_t)3 * N, 0.f);
    for (int g = 0; g < N; ++g)
        if (pof[g] >= 0 && !outl[g]) { held[g] = 1; memcpy(&held_xw[3 * (size_t)g], &mm_pos[3 * (size_t)pof[g]], 12); cand[mm_point[pof[g]]] = 0; }
    f.n_held = N; f.taken = held.data(); f.has_point = held.data(); f.point_xw = held_xw.data();
    f.n_points = nl; f.pos = lm_pos.data(); f.normal = lm_nrm.data(); f.min_dist = lm_min.data(); f.max_dist = lm_max.data(); f.candidate = cand.data(); f.desc = lm_desc.data();
    f.q_cam = nullptr; f.q_octave = nullptr; f.q_angle = nullptr;
    prm.th = 1.f; prm.nn_ratio = 0.8f;
    std::vector<int32_t> mop2((size_t)std::max(nl, 1));
    p_mop = mop2.data();
    CK(dcs_track_frame_device(1, &f, &prm, 0, 0, &res, st));
    printf("lm %d %d %016llx %016llx %016llx", n_match, n_inl, (unsigned long long)fnv(mop2.data(), sizeof(int32_t) * nl), (unsigned long long)fnv(pof.data(), sizeof(int32_t) * N),
           (unsigned long long)fnv(outl.data(), N));
    for (double v : out_pose) printf(" %.17g", v);
    printf("\n");
    dcs_orb_destroy(orb);
    (void)hipFree(d_img); (void)hipFree(d_kp); (void)hipFree(d_desc); (void)hipFree(d_n); (void)hipStreamDestroy(st);
    return 0;
}
