// orb_extract.cpp -- extractor handle and per-call orchestration (HIP streams/events, no kernels here).
//
// Mirrors ORBextractor (reference include/ORBextractor.h:45-113, src/ORBextractor.cc:1043-1105) for a
// BATCH of equally sized images. HBM layout per handle (sized for max_images B):
//   pyramid slab   B x slab_bytes   u8, level l at lv[l].offset, row pitch = roundup64(w_l)
//   blurred slab   B x slab_bytes   same geometry
//   FAST slots     B x n_slots      dcs_candidate, fixed capacity per 30-px cell (no atomics: order matters)
//   dense cands    one array for the whole batch, (image, level)-major, in the reference's emission order
// Stream plan per call: [resize l=1..L-1] -> {FAST cells -> scan -> gather -> D2H} on the main stream,
// Gaussian blur of all levels concurrently on an auxiliary stream, host quadtree (thread pool), then
// one orientation+rBRIEF launch that waits for the blur event.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "config.h"
#include "orb_host.h"
#include "orb_kernels.h"

namespace dcs {

// Upper bound of the quadtree's node list for one level: the first pass splits all nIni = round(w/h) initial nodes before N
// is looked at (<= 4 nIni nodes, ORBextractor.cc:594-673); every later pass stops at N + 3 at the latest.
static int octree_list_bound(int width, int height, int n_target)
{
    const int n_ini = height > 0 ? (int)std::round((float)width / (float)height) : 0;
    return std::max(n_target + 8, 4 * std::min(std::max(n_ini, 1), 255));
}


// ------------------------------------------------------------------ tiny persistent thread pool
class Pool {
public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // runs fn(i) for i in [0, n) on the workers + the calling thread
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (workers_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)workers_.size(); ++epoch_;
        }
        cv_.notify_all();
        run();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void run() { for (int i; (i = next_.fetch_add(1)) < n_;) (*fn_)(i); }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            run();
            { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

}  // namespace dcs

using namespace dcs;

constexpr bool kSmallGraphDefault = true;       // one dual frame per call 0.174 -> 0.165 ms (DCS_ORB_SMALL_GRAPH=0 switches it off)

struct dcs_orb {
    OrbTables t;
    dcs_orb_params prm;
    int device = 0;
    PyramidGeom g;
    bool configured = false;

    DevBuf<uint8_t> d_pyr, d_blur;
    DevBuf<int16_t> d_rtab;
    struct RTab { size_t xofs, xa, yofs, ya; } rtab[kMaxLevels];
    // Round 5, "FAST emits the next level": per level l < L - 1 the packed row table {sy, b0 | b1 << 16} of level l + 1 and the frame of
    // level l + 1 that no cell of level l produces (k_resize on four rectangles). emit_ok: the geometry allows it (every cell's taps inside its ROI).
    DevBuf<int32_t> d_erows;
    size_t erows_off[kMaxLevels] = {};
    ResizeRects frame[kMaxLevels] = {};
    FastEmit femit[kMaxLevels] = {};               // frame rectangles + block counts per emitting level (pointers filled per call)
    bool emit_ok = false;
    long long opt_dense_cap = 0;
    bool fast_hw = true;                           // k_fast_cells' hardware-specific forms passed the start-up probe on this device
    int emit_mode = -1;                            // DCS_ORB_EMIT when the handle is created: 0 = off, n > 0 = levels [0, n) emit whatever the batch; unset = all levels, from emit_min_pixels level-0 pixels per call
    double emit_min_pixels = 2.5e7;
    std::vector<CellDesc> h_cells;
    std::vector<int32_t> h_level_cell_begin;
    DevBuf<CellDesc> d_cells;
    DevBuf<int32_t> d_level_cell_begin;
    DevBuf<dcs_candidate> d_slots, d_dense;
    size_t dense_cap = 0;
    DevBuf<int32_t> d_cell_count, d_cell_off, d_lvl_total, d_lvl_off;
    PinnedBuf<int32_t> h_lvl_off;
    PinnedBuf<dcs_candidate> h_dense;
    PinnedBuf<SelKp> h_sel;
    DevBuf<SelKp> d_sel;
    PinnedBuf<int32_t> h_img_off;
    DevBuf<int32_t> d_img_off;
    // device quadtree (default): per-(image, level) output slots + HBM scratch
    bool device_octree = true;
    bool no_overlap = false;
    OctLevels oct{};
    DevBuf<int32_t> d_lvl_cnt, d_oct_flag;
    DevBuf<uint32_t> d_ic_mask;
    std::unique_ptr<Pool> stage_pool;
    DevBuf<uint8_t> d_stage;
    PinnedBuf<dcs_keypoint> h_kp_out;
    PinnedBuf<uint8_t> h_desc_out;
    DevBuf<unsigned long long> d_oct_u64[3];
    DevBuf<unsigned> d_oct_u32[2];
    DevBuf<int> d_oct_i32[8];
    DevBuf<unsigned char> d_oct_u8;
    bool host_copy_valid = false;       // h_lvl_off / h_dense mirror the last call (debug taps)
    int last_tasks = 0;
    // staging for the host-buffer API
    DevBuf<dcs_keypoint> d_kp;
    DevBuf<uint8_t> d_desc;
    DevBuf<uint8_t> d_out;               // one-shot host calls: [counts | key points | descriptors] in one block, one download
    PinnedBuf<uint8_t> h_out;
    DevBuf<int32_t> d_n;
    PinnedBuf<dcs_keypoint> h_kp;
    PinnedBuf<uint8_t> h_desc;
    PinnedBuf<int32_t> h_n;
    PinnedBuf<uint8_t> h_img;

    hipStream_t s_main = nullptr, s_aux = nullptr, s_fast = nullptr;
    // host-buffer batches run as a pipeline of image chunks: upload of chunk k + 1 || kernels of chunk k || download of chunk k - 1
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    // side streams are made on first use, each on a hardware queue of its own next to the streams it runs beside (common.cpp:
    // create_stream_apart -- two streams on one hardware queue run one after the other, and which streams share one is an accident of how
    // many the process has alive)
    int side_stream(hipStream_t& s, hipStream_t beside_a, hipStream_t beside_b = nullptr)
    {
        if (s) return DCS_OK;
        const hipStream_t avoid[2] = {beside_a, beside_b};                 // beside_a may be the legacy default stream (0): it has a queue too
        return create_stream_apart(&s, avoid, beside_b ? 2 : 1, nullptr);
    }
    std::vector<hipEvent_t> ev_chunk;              // 3 per chunk: uploaded, computed, downloaded
    hipEvent_t ev_lvl = nullptr, ev_fast_early = nullptr;
    int fast_split = 0;                            // DCS_ORB_FAST_SPLIT: FAST of levels [0, fast_split) starts on its own stream as soon as they exist
    hipEvent_t ev_pyr = nullptr, ev_blur = nullptr;
    // per-call timing events live in a small ring so that asynchronous callers are never stalled: a set is harvested
    // (elapsed times read and accumulated) only when it is about to be reused or when the totals are requested
    static constexpr int kRing = 8;
    struct EvSet { hipEvent_t t[7] = {}, b[2] = {}, f[2] = {}; float host_us = 0; bool pending = false, dev_oct = true, has_f = false, has_b = false, all = true; } ring[kRing];
    int timing_mode = 2;                                     // dcs_orb_set_timing: 0 no stage markers, 1 the FAST stage only, 2 every stage
    hipEvent_t* ev_t = nullptr; hipEvent_t* ev_b = nullptr; hipEvent_t* ev_f = nullptr;     // the set of the call in flight
    long n_calls = 0;
    float last_us[7] = {0, 0, 0, 0, 0, 0, 0};
    double sum_us[7] = {0, 0, 0, 0, 0, 0, 0};
    long n_harvested = 0;
    float host_us = 0;
    bool timing_valid = false;
    int harvest(EvSet& es);
    std::unique_ptr<Pool> pool;
    std::vector<std::vector<dcs_candidate>> task_out;

    // last call (debug taps)
    LevelSet last_raw{}, last_blur{};
    bool last_blur_valid = false;                  // fused describe: the blurred pyramid is only made when dcs_orb_debug_level asks for it
    const uint8_t* locked_lo = nullptr;            // host range last found page-locked by dcs_orb_extract_batch (read in place by the DMA)
    const uint8_t* locked_hi = nullptr;
    const uint8_t* pageable_lo = nullptr;          // ... and the last range found pageable (no second query for a caller that re-uses its buffers)
    const uint8_t* pageable_hi = nullptr;
    // one or two host images per call: the call's launches replayed as one executable graph (dcs_orb_extract_batch)
    struct SmallGraphKey {
        const void* stage; const void* d_out; const void* h_out; size_t img_bytes, total; int n, rows, cols, pitch, cap; long generation;
        bool operator==(const SmallGraphKey& o) const
        { return generation == o.generation && stage == o.stage && d_out == o.d_out && h_out == o.h_out && img_bytes == o.img_bytes && total == o.total && n == o.n && rows == o.rows && cols == o.cols && pitch == o.pitch && cap == o.cap; }
    };
    SmallGraphKey small_graph_key{};
    hipGraphExec_t small_graph_exec = nullptr;
    int small_graph_seen = 0;
    bool small_graph_broken = false;
    int last_host_path = -1;                       // dcs_orb_debug_host_path: 1 = the last host-buffer call let the DMA read the caller's page-locked frames in place, 0 = packed
    int last_small_graph = 0;                      // ... and 1 = it was replayed as the small-call graph
    long config_generation = 0;                    // bumped by every configure() that rebuilds the buffers (part of the graph key)
    int fused_mode = -1;                           // DCS_ORB_FUSED_BLUR when the handle is created: 0 / 1, unset = choose per call
    int last_n_images = 0;
    int last_emit_levels = 0;                      // pyramid levels the FAST cells of the last call produced (0: the resize chain did)

    ~dcs_orb() {
        if (s_main) (void)hipStreamDestroy(s_main);
        if (small_graph_exec) (void)hipGraphExecDestroy(small_graph_exec);
        if (s_h2d) (void)hipStreamDestroy(s_h2d);
        if (s_d2h) (void)hipStreamDestroy(s_d2h);
        for (hipEvent_t e : ev_chunk) (void)hipEventDestroy(e);
        if (s_aux) (void)hipStreamDestroy(s_aux);
        if (s_fast) (void)hipStreamDestroy(s_fast);
        if (ev_lvl) (void)hipEventDestroy(ev_lvl);
        if (ev_fast_early) (void)hipEventDestroy(ev_fast_early);
        if (ev_pyr) (void)hipEventDestroy(ev_pyr);
        if (ev_blur) (void)hipEventDestroy(ev_blur);
        for (auto& es : ring) { for (auto& e : es.t) if (e) (void)hipEventDestroy(e); for (auto& e : es.b) if (e) (void)hipEventDestroy(e); for (auto& e : es.f) if (e) (void)hipEventDestroy(e); }
    }

    int configure(int rows, int cols);
    LevelSet make_levels(const uint8_t* slab, const uint8_t* level0, size_t level0_img_stride, int level0_pitch) const;
    int run(const uint8_t* d_level0, size_t level0_img_stride, int level0_pitch, int n_images,
            dcs_keypoint* d_kp_out, uint8_t* d_desc_out, int cap, int32_t* d_n_out, hipStream_t stream,
            int* h_counts /* may be null */);
};

int dcs_orb::configure(int rows, int cols)
{
    if (configured && g.rows == rows && g.cols == cols) return DCS_OK;
    if (rows < 2 * kEdgeThreshold || cols < 2 * kEdgeThreshold || rows > 16384 || cols > 16384) {
        set_error("image size %dx%d outside supported range", cols, rows);
        return DCS_ERR_INVALID;
    }
    // Everything below frees and reallocates the handle's buffers: an executable graph captured for the previous shape holds their old
    // addresses (d_pyr, d_cells, d_slots, d_dense, d_sel, d_rtab, d_ic_mask ...) and must never be replayed again -- the key of the small-call
    // graph only covers the staging / output blocks, and a shape change can arrive through dcs_orb_extract_batch_device, which never looks at it.
    if (small_graph_exec) { (void)hipGraphExecDestroy(small_graph_exec); small_graph_exec = nullptr; }
    small_graph_seen = 0; small_graph_key = SmallGraphKey{};
    ++config_generation;
    configured = false;
    g.build(t, rows, cols);
    const int B = prm.max_images, L = t.nlevels;
    for (int l = 0; l < L; ++l) {
        if (g.lv[l].w < 2 * kEdgeThreshold + 1 || g.lv[l].h < 2 * kEdgeThreshold + 1) {
            set_error("pyramid level %d (%dx%d) is smaller than the 19-px border allows", l, g.lv[l].w, g.lv[l].h);
            return DCS_ERR_INVALID;
        }
        if (g.lv[l].w_cell + 6 > 65 || g.lv[l].h_cell + 6 > 65) { set_error("cell too large"); return DCS_ERR_UNSUPPORTED; }
    }
    int rc;
    if ((rc = d_pyr.resize((size_t)B * g.slab_bytes + 256))) return rc;
    if ((rc = d_blur.resize((size_t)B * g.slab_bytes + 256))) return rc;
    // resize coefficient tables, one set per level >= 1
    std::vector<int16_t> tab;
    for (int l = 1; l < L; ++l) {
        ResizeTable rt;
        rt.build(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h);
        while (tab.size() % 8) tab.push_back(0);                 // 16-byte aligned packed columns {sx, 0, a0, a1}: an emitting FAST cell reads four of them as two 16-byte loads
        rtab[l].xofs = tab.size();
        for (size_t x = 0; x < rt.xofs.size(); ++x) { tab.push_back(rt.xofs[x]); tab.push_back(0); tab.push_back(rt.xa[2 * x]); tab.push_back(rt.xa[2 * x + 1]); }
        rtab[l].xa = tab.size();
        rtab[l].yofs = tab.size(); tab.insert(tab.end(), rt.yofs.begin(), rt.yofs.end());
        while (tab.size() % 2) tab.push_back(0);                 // k_resize reads a row's two coefficients as one dword
        rtab[l].ya = tab.size();   tab.insert(tab.end(), rt.ya.begin(), rt.ya.end());
    }
    if ((rc = d_rtab.resize(std::max<size_t>(tab.size(), 1)))) return rc;
    if (!tab.empty()) DCS_HIP(hipMemcpy(d_rtab.p, tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    // FAST cells (ORBextractor.cc:789-806)
    h_cells.clear(); h_level_cell_begin.assign(L + 1, 0);
    for (int l = 0; l < L; ++l) {
        const LevelGeom& lg = g.lv[l];
        h_level_cell_begin[l] = (int)h_cells.size();
        const int max_bx = lg.w - kEdgeThreshold + 3, max_by = lg.h - kEdgeThreshold + 3;
        for (int i = 0; i < lg.n_rows; ++i) {
            const int ini_y = kMinBorder + i * lg.h_cell;
            int max_y = ini_y + lg.h_cell + 6;
            const bool skip_row = ini_y >= max_by - 3;
            if (max_y > max_by) max_y = max_by;
            for (int j = 0; j < lg.n_cols; ++j) {
                const int ini_x = kMinBorder + j * lg.w_cell;
                int max_x = ini_x + lg.w_cell + 6;
                const bool skip = skip_row || ini_x >= max_bx - 6;
                if (max_x > max_bx) max_x = max_bx;
                CellDesc c{};
                c.level = (int16_t)l; c.x0 = (int16_t)ini_x; c.y0 = (int16_t)ini_y;
                c.rw = (int16_t)(skip ? 0 : max_x - ini_x); c.rh = (int16_t)(skip ? 0 : max_y - ini_y);
                c.ox = (int16_t)(j * lg.w_cell); c.oy = (int16_t)(i * lg.h_cell);
                c.cap = (int16_t)lg.cell_cap;
                c.slot_base = (int32_t)(lg.slot_base + (size_t)(i * lg.n_cols + j) * lg.cell_cap);
                h_cells.push_back(c);
            }
        }
    }
    h_level_cell_begin[L] = (int)h_cells.size();
    // ---- which part of level l + 1 each cell of level l produces (k_fast_cells<EMIT>), and the frame left to k_resize.
    // Columns: cell column j owns the destination dwords k whose first source column sx(4k) lies in [x0_j, x0_{j+1}) (the last emitting column:
    // up to its ROI's end), provided every tap of the dword's four pixels lies inside the ROI; rows likewise with the upper source row. The ROI
    // reaches 6 pixels into the next cell, a dword's taps span <= 3 * scale + 2 columns: always inside for scale <= 1.33; anything else
    // (a cell that cannot hold its share, skipped or degenerate cells inside the grid) switches the mode off for this image size.
    emit_ok = L > 1;
    std::vector<int32_t> erows;
    for (int l = 0; l + 1 < L && emit_ok; ++l) {
        const LevelGeom& lg = g.lv[l];
        const LevelGeom& ld = g.lv[l + 1];
        ResizeTable rt;
        rt.build(lg.w, lg.h, ld.w, ld.h);
        erows_off[l] = erows.size();
        for (int dy = 0; dy < ld.h; ++dy) { erows.push_back(rt.yofs[dy]); erows.push_back((int32_t)((uint32_t)(uint16_t)rt.ya[2 * dy] | ((uint32_t)(uint16_t)rt.ya[2 * dy + 1] << 16))); }
        const int n_x4 = (ld.w + 3) / 4;
        const int c0 = h_level_cell_begin[l];
        // emitting columns / rows: the leading run of cells with a usable ROI
        int ncol = 0, nrow = 0;
        while (ncol < lg.n_cols && h_cells[c0 + ncol].rw >= 7) ++ncol;
        while (nrow < lg.n_rows && h_cells[c0 + nrow * lg.n_cols].rh >= 7) ++nrow;
        for (int j = ncol; j < lg.n_cols; ++j) if (h_cells[c0 + j].rw >= 7) emit_ok = false;                     // a gap inside the grid
        for (int i = nrow; i < lg.n_rows; ++i) if (h_cells[c0 + i * lg.n_cols].rh >= 7) emit_ok = false;
        if (ncol == 0 || nrow == 0) emit_ok = false;
        if (!emit_ok) break;
        std::vector<int> kx_begin(ncol + 1), dy_begin(nrow + 1);
        auto dword_fits = [&](int k, int lo, int hi) {           // all taps of destination pixels 4k .. 4k + 3 inside source columns [lo, hi)
            if (4 * k + 3 >= ld.w) return false;                 // a partial last dword belongs to the frame
            for (int i = 0; i < 4; ++i) {
                const int sx = rt.xofs[4 * k + i];
                if (sx < lo || sx + 1 >= hi || sx < rt.xofs[4 * k]) return false;
            }
            return rt.xofs[4 * k + 3] + 1 - rt.xofs[4 * k] <= 7;          // one 8-byte LDS window per source row holds every tap
        };
        {
            int k = 0;
            for (int j = 0; j < ncol; ++j) {
                const CellDesc& c = h_cells[c0 + j];
                const int own_hi = j + 1 < ncol ? h_cells[c0 + j + 1].x0 : c.x0 + c.rw;
                while (k < n_x4 && 4 * k < ld.w && rt.xofs[4 * k] < c.x0) ++k;           // (only before the first column: the frame's left part)
                kx_begin[j] = k;
                while (k < n_x4 && 4 * k < ld.w && rt.xofs[4 * k] < own_hi) {
                    if (!dword_fits(k, c.x0, c.x0 + c.rw)) { if (j + 1 < ncol) emit_ok = false; break; }   // the last column simply stops here
                    ++k;
                }
                if (j + 1 < ncol && (4 * k >= ld.w || rt.xofs[4 * k] < own_hi)) emit_ok = false;
            }
            kx_begin[ncol] = k;
        }
        {
            int d = 0;
            for (int i = 0; i < nrow; ++i) {
                const CellDesc& c = h_cells[c0 + i * lg.n_cols];
                const int own_hi = i + 1 < nrow ? h_cells[c0 + (i + 1) * lg.n_cols].y0 : c.y0 + c.rh;
                while (d < ld.h && rt.yofs[d] < c.y0) ++d;
                dy_begin[i] = d;
                while (d < ld.h && rt.yofs[d] < own_hi) {
                    if (rt.yofs[d] + 1 >= c.y0 + c.rh) { if (i + 1 < nrow) emit_ok = false; break; }
                    ++d;
                }
                if (i + 1 < nrow && (d >= ld.h || rt.yofs[d] < own_hi)) emit_ok = false;
            }
            dy_begin[nrow] = d;
        }
        for (int j = 0; j + 1 < ncol && emit_ok; ++j) if (kx_begin[j + 1] - kx_begin[j] > 64) emit_ok = false;
        if (!emit_ok) break;
        for (int i = 0; i < nrow; ++i)
            for (int j = 0; j < ncol; ++j) {
                CellDesc& c = h_cells[c0 + i * lg.n_cols + j];
                const int nkx = kx_begin[j + 1] - kx_begin[j], ndy = dy_begin[i + 1] - dy_begin[i];
                if (nkx <= 0 || ndy <= 0 || nkx > 64) continue;
                c.ekx0 = (int16_t)kx_begin[j]; c.enkx = (int16_t)nkx; c.edy0 = (int16_t)dy_begin[i]; c.endy = (int16_t)ndy;
                c.eG = (int16_t)(64 / nkx); c.erounds = (int16_t)((ndy + c.eG - 1) / c.eG); c.emul = (65536 + nkx - 1) / nkx;
            }
        ResizeRects& fr = frame[l];
        const int KX0 = kx_begin[0], KX1 = kx_begin[ncol], DY0 = dy_begin[0], DY1 = dy_begin[nrow];
        fr.n = 4;
        fr.r[0] = ResizeRect{0, n_x4, 0, DY0};                   // top
        fr.r[1] = ResizeRect{0, n_x4, DY1, ld.h};                // bottom
        fr.r[2] = ResizeRect{0, KX0, DY0, DY1};                  // left
        fr.r[3] = ResizeRect{KX1, n_x4 - KX1, DY0, DY1};         // right
        FastEmit& fe = femit[l];
        fe = FastEmit{};
        int blocks = 0;
        for (int k = 0; k < 4; ++k) {
            const int nx = std::max(fr.r[k].x4_count, 0), nr = std::max(fr.r[k].row_end - fr.r[k].row_begin, 0);
            FrameRect& q = fe.fr[k];
            q.x4_begin = fr.r[k].x4_begin; q.x4_count = std::max(nx, 1); q.row_begin = fr.r[k].row_begin; q.row_end = fr.r[k].row_end;
            q.count = nx * ((nr + 3) / 4); q.blk_begin = blocks;                     // a lane = one dword column x 4 rows
            q.magic = nx > 1 ? (unsigned)((0x100000000ull + (unsigned)nx - 1) / (unsigned)nx) : 0u;
            if (nx == 1 && nr > 0) emit_ok = false;                                   // (i / 1 needs a 33-bit magic; never seen: the frame is ~3 dwords wide)
            blocks += (q.count + 63) / 64;
        }
        fe.n_frame_blocks = blocks; fe.src_level = l;
    }
    if (!emit_ok) for (CellDesc& c : h_cells) { c.enkx = 0; c.endy = 0; }
    if ((rc = d_erows.resize(std::max<size_t>(erows.size(), 2)))) return rc;
    if (!erows.empty()) DCS_HIP(hipMemcpy(d_erows.p, erows.data(), erows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    const int n_cells = (int)h_cells.size();
    if ((rc = d_cells.resize(std::max(n_cells, 1)))) return rc;
    if (n_cells) DCS_HIP(hipMemcpy(d_cells.p, h_cells.data(), sizeof(CellDesc) * n_cells, hipMemcpyHostToDevice));
    if ((rc = d_level_cell_begin.resize(L + 1))) return rc;
    DCS_HIP(hipMemcpy(d_level_cell_begin.p, h_level_cell_begin.data(), sizeof(int32_t) * (L + 1), hipMemcpyHostToDevice));
    if ((rc = d_slots.resize(std::max<size_t>((size_t)B * g.n_slots, 1)))) return rc;
    if ((rc = d_cell_count.resize(std::max<size_t>((size_t)B * n_cells, 1)))) return rc;
    if ((rc = d_cell_off.resize(std::max<size_t>((size_t)B * n_cells, 1)))) return rc;
    if ((rc = d_lvl_total.resize((size_t)B * L))) return rc;
    if ((rc = d_lvl_off.resize((size_t)B * L + 1))) return rc;
    if ((rc = h_lvl_off.resize((size_t)B * L + 1))) return rc;
    // dense candidates: generous (1/16 of the pixels), the call fails loudly if an image exceeds it
    size_t px = 0;
    for (int l = 0; l < L; ++l) px += (size_t)g.lv[l].w * g.lv[l].h;
    dense_cap = std::min<size_t>((size_t)B * g.n_slots, (size_t)B * std::max<size_t>(px / 16, 4096));
    dense_cap = std::max<size_t>(dense_cap, 1);
    if (opt_dense_cap > 0) dense_cap = (size_t)opt_dense_cap;      // test hook (option DCS_ORB_DENSE_CAP when the handle was created): provoke the overflow path
    if ((rc = d_dense.resize(dense_cap))) return rc;
    if ((rc = h_dense.resize(dense_cap))) return rc;
    oct = OctLevels();
    oct.nlevels = L;
    int out_total = 0;
    for (int l = 0; l < L; ++l) {
        OctLevel& ol = oct.lv[l];
        ol.width = (g.lv[l].w - kEdgeThreshold + 3) - kMinBorder; ol.height = (g.lv[l].h - kEdgeThreshold + 3) - kMinBorder;
        ol.n_target = t.n_per_level[l];
        ol.out_base = out_total; ol.out_cap = octree_list_bound(ol.width, ol.height, ol.n_target);
        out_total += ol.out_cap;
    }
    oct.out_per_image = out_total;
    {
        std::vector<uint32_t> mask(kIcMaskWords);
        build_ic_mask(t.umax, mask.data());
        if ((rc = d_ic_mask.resize(kIcMaskWords))) return rc;
        DCS_HIP(hipMemcpy(d_ic_mask.p, mask.data(), sizeof(uint32_t) * kIcMaskWords, hipMemcpyHostToDevice));
    }
    if (device_octree) {
        if ((rc = d_lvl_cnt.resize((size_t)B * L))) return rc;
        if ((rc = d_oct_flag.resize((size_t)B * L))) return rc;
        for (auto& b : d_oct_u64) if ((rc = b.resize(2 * dense_cap))) return rc;
        for (auto& b : d_oct_u32) if ((rc = b.resize(2 * dense_cap))) return rc;
        for (auto& b : d_oct_i32) if ((rc = b.resize(2 * dense_cap))) return rc;
        if ((rc = d_oct_u8.resize(2 * dense_cap))) return rc;
    }
    const size_t max_sel = std::max((size_t)B * ((size_t)t.nfeatures + 4 * L + 64) * 2, (size_t)B * out_total);
    if ((rc = h_sel.resize(max_sel))) return rc;
    if ((rc = d_sel.resize(max_sel))) return rc;
    if ((rc = h_img_off.resize(B + 1))) return rc;
    if ((rc = d_img_off.resize(B + 1))) return rc;
    task_out.resize((size_t)B * L);
    configured = true;
    return DCS_OK;
}

LevelSet dcs_orb::make_levels(const uint8_t* slab, const uint8_t* level0, size_t level0_img_stride, int level0_pitch) const
{
    LevelSet s{};
    s.nlevels = t.nlevels;
    for (int l = 0; l < t.nlevels; ++l) {
        s.lv[l].base = slab + g.lv[l].offset;
        s.lv[l].img_stride = g.slab_bytes;
        s.lv[l].w = g.lv[l].w; s.lv[l].h = g.lv[l].h; s.lv[l].pitch = g.lv[l].pitch;
    }
    if (level0) { s.lv[0].base = level0; s.lv[0].img_stride = level0_img_stride; s.lv[0].pitch = level0_pitch; }
    return s;
}

int dcs_orb::run(const uint8_t* d_level0, size_t level0_img_stride, int level0_pitch, int n_images,
                 dcs_keypoint* d_kp_out, uint8_t* d_desc_out, int cap, int32_t* d_n_out, hipStream_t stream, int* h_counts)
{
    const int L = t.nlevels, n_cells = (int)h_cells.size();
    const LevelSet raw = make_levels(d_pyr.p, d_level0, level0_img_stride, level0_pitch);
    const LevelSet blur = make_levels(d_blur.p, nullptr, 0, 0);
    last_raw = raw; last_blur = blur; last_n_images = n_images;
    timing_valid = false;
    int rc;
    EvSet& es = ring[n_calls % kRing];
    if (es.pending && (rc = harvest(es))) return rc;
    ev_t = es.t; ev_b = es.b; ev_f = es.f; es.dev_oct = device_octree;
    ++n_calls;
    // Stage timing (dcs_orb_timing_totals) brackets every stage with hipEvents; each record is a marker packet between two kernels and
    // costs 5 - 10 us of queue latency -- 45 us of a 220-us dual-frame call. Calls of one or two images (a frame per call) skip them (DCS_ORB_TIMING=1 keeps them).
    const bool timed = timing_mode > 0 && (n_images > 2 || no_overlap);
    const bool timed_all = timed && timing_mode >= 2;        // mode 1: only the two markers around FAST (the roofline kernel) are recorded
    es.all = timed_all; es.has_f = false; es.has_b = false;
#define DCS_MARK(e, s) do { if (timed_all) DCS_HIP(hipEventRecord(e, s)); } while (0)
#define DCS_MARK_FAST(e, s) do { if (timed) DCS_HIP(hipEventRecord(e, s)); } while (0)
    DCS_MARK(ev_t[0], stream);
    FastFootprint fp_all;                                    // every cell of the pyramid in one launch
    for (const CellDesc& c : h_cells) fast_footprint_add(fp_all, c.rw, c.rh);
    // Early FAST (DCS_ORB_FAST_SPLIT=k when the handle is created; OPT-IN): the cells of levels [0, k) -- most of the pixels -- start on
    // their own stream as soon as those levels exist and run next to the rest of the resize chain (7 dependent launches at half the
    // chip's issue rate) instead of after it. Measured: an extraction running ALONE gains 6 % with the fused describe (k = 2: 1 740 ->
    // 1 640 us per 512 images; 1: 1 700, 3: 1 650, 4: 1 700) and 2 % with the separate blur kernels; in the benchmark's pipeline, where
    // the matcher of the previous step already fills those idle issue slots, it LOSES 3-7 % (fused + 2: 277 k, separate + 3: 267 k
    // against 287 k kfeatures/s): the steady state is bound by the number of vector instructions, not by idle time.
    const int split = (no_overlap || L < 3) ? 0 : std::min(fast_split, L - 1);
    if (split > 0 && (rc = side_stream(s_fast, stream))) return rc;
    const int cells_early = split > 0 ? h_level_cell_begin[split] : 0;
    const bool fused_blur = fused_mode >= 0 ? fused_mode != 0 : true;
    // Round 5: no resize chain -- the FAST cells of level l write level l + 1 from their LDS tiles (k_fast_cells<EMIT>), k_resize only the
    // frame around them; one FAST launch per level, each behind the one that produced its level. Needs the fused describe (the separate blur
    // kernels want the whole pyramid before FAST starts). DCS_ORB_EMIT=0 (read when the handle is created) keeps the round-4 pipeline.
    // (the frame workgroups read level 0 as aligned dwords like k_resize<true>: a caller's buffer that is not 4-byte aligned keeps the resize chain)
    const bool level0_aligned = ((reinterpret_cast<uintptr_t>(raw.lv[0].base) | (uintptr_t)raw.lv[0].img_stride | (uintptr_t)raw.lv[0].pitch) & 3) == 0 && raw.lv[0].pitch >= 12;
    // Small batches keep the chain as well: they are bound by launch latency, and eight dependent FAST launches whose waves each run ~12 us are no
    // shorter than seven small resizes + one FAST launch (per call alone, 640 x 480: 2 images 114 against 103 us, 16 images 168 / 159, 64 images
    // 265 / 260; 512 images 1 253 / 1 331 and 128 images of 1280 x 720 1 075 / 1 130 the other way). DCS_ORB_EMIT=n forces the mode on for any batch.
    const bool emit = emit_ok && split == 0 && fused_blur && emit_mode != 0 && level0_aligned &&
                      (emit_mode > 0 || (double)n_images * g.rows * g.cols >= emit_min_pixels);
    const int E = emit ? (emit_mode > 0 ? std::min(emit_mode, L - 1) : L - 1) : 0;
    last_emit_levels = E;      // levels [0, E) emit levels [1, E]; the rest of the chain is k_resize
    for (int l = 1; l < L && !emit; ++l) {
        if ((rc = launch_resize(raw.lv[l - 1], raw.lv[l], d_rtab.p + rtab[l].xofs,
                                d_rtab.p + rtab[l].yofs, d_rtab.p + rtab[l].ya, n_images, stream))) return rc;
        if (cells_early > 0 && l == std::max(split - 1, 1)) {              // levels 0 .. split - 1 are complete (split == 1: level 0 needs no resize)
            DCS_HIP(hipEventRecord(ev_lvl, stream));
            DCS_HIP(hipStreamWaitEvent(s_fast, ev_lvl, 0));
            es.has_f = timed_all;
            DCS_MARK(ev_f[0], s_fast);
            if ((rc = launch_fast_cells(raw, d_cells.p, n_cells, n_images, t.ini_th, t.min_th, d_slots.p, g.n_slots,
                                        d_cell_count.p, fp_all, s_fast, 0, cells_early, nullptr, fast_hw))) return rc;
            DCS_MARK(ev_f[1], s_fast);
            DCS_HIP(hipEventRecord(ev_fast_early, s_fast));
        }
    }
    DCS_MARK_FAST(ev_t[1], stream);
    // blur on the auxiliary stream, overlapping FAST (DCS_ORB_NO_OVERLAP=1 serialises it for clean timings).
    // DCS_ORB_BLUR_LATE=1 starts it after FAST instead (measured slower: it then collides with the latency-bound
    // compaction + quadtree kernels, 1.64 vs 1.55 ms per 128 dual frames).
    // fused_blur (the DEFAULT since round 3): no blur kernels, no blurred pyramid, no auxiliary stream -- k_describe<FUSED> blurs every
    // keypoint's 43 x 43 raw patch itself (horizontal pass on the matrix cores, vertical pass at the tested pixels only;
    // dcs_orb_debug_level makes the blurred level on demand). Bit-exact either way. Measured on the headline workload (640 x 480 / 1000
    // features, 512 images per step): 313 k kfeatures/s fused against 289 k with the separate kernels -- FAST runs at its solo speed once
    // the blur no longer competes for the vector ALUs (800 -> 619 us) and the fused describe costs 455 instead of 388 us.
    // DCS_ORB_FUSED_BLUR=0 (read when the handle is created) selects the separate blur kernels (the debug / A-B path).
    last_blur_valid = !fused_blur;
    const bool blur_early = true;                            // (separate blur kernels start beside FAST; the "late" placement of round 2 lost and its switch is gone)
    if (!fused_blur && !no_overlap && (rc = side_stream(s_aux, stream, s_fast))) return rc;
    hipStream_t sb = no_overlap ? stream : s_aux;
    auto blur_stage = [&]() -> int {
        if (fused_blur) return DCS_OK;                       // no blur kernels, no markers (every record is a packet on the stream)
        es.has_b = timed_all;
        if (!no_overlap) { DCS_HIP(hipEventRecord(ev_pyr, stream)); DCS_HIP(hipStreamWaitEvent(s_aux, ev_pyr, 0)); }
        DCS_MARK(ev_b[0], sb);
        int r = launch_blur(raw, blur, n_images, sb);
        if (r) return r;
        DCS_MARK(ev_b[1], sb);
        DCS_HIP(hipEventRecord(ev_blur, sb));
        return DCS_OK;
    };
    if (no_overlap || blur_early) {
        if ((rc = blur_stage())) return rc;
        if (no_overlap) DCS_MARK_FAST(ev_t[1], stream);      // FAST timing starts after the blur
    }

    if (emit) {
        for (int l = 0; l < E; ++l) {
            FastFootprint lf;
            const int c0 = h_level_cell_begin[l], c1 = h_level_cell_begin[l + 1];
            for (int c = c0; c < c1; ++c) fast_footprint_add(lf, h_cells[c].rw, h_cells[c].rh);
            FastEmit fe = femit[l];
            fe.dst = raw.lv[l + 1]; fe.cols = d_rtab.p + rtab[l + 1].xofs; fe.rows = d_erows.p + erows_off[l];
            if (c1 > c0 && (rc = launch_fast_cells(raw, d_cells.p, n_cells, n_images, t.ini_th, t.min_th, d_slots.p, g.n_slots,
                                                   d_cell_count.p, lf, stream, c0, c1 - c0, &fe, fast_hw))) return rc;
        }
        for (int l = E + 1; l < L; ++l)
            if ((rc = launch_resize(raw.lv[l - 1], raw.lv[l], d_rtab.p + rtab[l].xofs, d_rtab.p + rtab[l].yofs, d_rtab.p + rtab[l].ya, n_images, stream))) return rc;
    }
    if (cells_early > 0) {
        if ((rc = launch_fast_cells(raw, d_cells.p, n_cells, n_images, t.ini_th, t.min_th, d_slots.p, g.n_slots,
                                    d_cell_count.p, fp_all, stream, cells_early, n_cells - cells_early, nullptr, fast_hw))) return rc;
    } else {
        // Launches by LDS footprint: a cell's workgroup (one wave) holds its ROI, score map and survivor list in LDS, sized for the
        // largest ROI of the LAUNCH, and that footprint decides how many cells a CU holds -- 5 104 B for the 38 x 38 ROIs of levels 0-3 of
        // the 640 x 480 pyramid = 32 waves per CU, 5.2-5.4 KB for levels 4 / 5 / 6 = 30 / 30 / 31, 5.7 KB for the 43-wide cells of level 7
        // = 28 (survivor list and score map sized for the largest CELL of the launch, not for the largest width x the largest height). One launch sized for level 7's twelve cells held every level at 24 (DCS_ORB_FAST_GROUPS=0 restores it: 603 us per 512
        // images); one launch per footprint class (five) ran in 547 us, but three of them were 1.5-3.6 rounds of the chip long and paid
        // ramp, tail and launch gap for that. The partition of the levels into consecutive groups is now chosen by cost: a group costs
        // its cells / (waves the chip holds at the group's footprint), each level's share raised by TWICE its relative loss of occupancy
        // against a launch of its own (a level running below its own occupancy loses more than the proportion: [0-4] at 30 per CU
        // measured 558 us), plus a fixed 1.3 rounds per launch. Headline pyramid: [0-3] 32, [4-6] 29, [7] 24 = 525 us.
        // Small batches are latency-bound -- one launch there: 16 images of 1280 x 720 (config C5's step) run at 89 k kfeatures/s with
        // one launch against 80 k with three.
        const bool grouped = n_images >= 64;
        auto wg_per_cu = [](const FastFootprint& f) { return std::min(32, 163840 / std::max(fast_cells_lds_bytes(f), 1)); };
        FastFootprint lfp[kMaxLevels];
        int start_of[kMaxLevels + 1];
        auto merged = [](FastFootprint a, const FastFootprint& b) {
            a.max_rw = std::max(a.max_rw, b.max_rw); a.max_rh = std::max(a.max_rh, b.max_rh);
            a.list_entries = std::max(a.list_entries, b.list_entries); a.sc_bytes = std::max(a.sc_bytes, b.sc_bytes);
            return a;
        };
        for (int l = 0; l < L; ++l)
            for (int c = h_level_cell_begin[l]; c < h_level_cell_begin[l + 1]; ++c) fast_footprint_add(lfp[l], h_cells[c].rw, h_cells[c].rh);
        if (!grouped) { for (int i = 0; i <= L; ++i) start_of[i] = E; }
        else {
            double best[kMaxLevels + 1];
            best[E] = 0;
            for (int i = E + 1; i <= L; ++i) {               // best[i] = cheapest partition of levels [E, i); the last group is [start_of[i], i)
                best[i] = 1e300; start_of[i] = E;
                FastFootprint gf;
                for (int j = i - 1; j >= E; --j) {
                    gf = merged(gf, lfp[j]);
                    const double occ = wg_per_cu(gf);
                    double cost = 1.3;
                    for (int l = j; l < i; ++l) {
                        const double own = wg_per_cu(lfp[l]);
                        const double rounds = (double)(h_level_cell_begin[l + 1] - h_level_cell_begin[l]) * n_images / (256.0 * occ);
                        cost += rounds * (1.0 + 2.0 * (own - occ) / own);
                    }
                    if (best[j] + cost < best[i]) { best[i] = best[j] + cost; start_of[i] = j; }
                }
            }
        }
        int bounds[kMaxLevels + 1], nb = 0;                  // group boundaries, last to first
        for (int i = L; i > E; i = start_of[i]) bounds[nb++] = i;
        int l0 = E;
        for (int k = nb - 1; k >= 0; --k) {
            const int l1 = bounds[k];
            FastFootprint gf;
            for (int l = l0; l < l1; ++l) gf = merged(gf, lfp[l]);
            const int c0 = h_level_cell_begin[l0], c1 = h_level_cell_begin[l1];
            if (c1 > c0 && (rc = launch_fast_cells(raw, d_cells.p, n_cells, n_images, t.ini_th, t.min_th, d_slots.p, g.n_slots,
                                                   d_cell_count.p, gf, stream, c0, c1 - c0, nullptr, fast_hw))) return rc;
            l0 = l1;
        }
    }
    DCS_MARK_FAST(ev_t[2], stream);
    if (cells_early > 0) DCS_HIP(hipStreamWaitEvent(stream, ev_fast_early, 0));      // the compaction needs every cell's count
    if (!(no_overlap || blur_early) && (rc = blur_stage())) return rc;
    if ((rc = launch_compact(d_cells.p, d_level_cell_begin.p, L, n_images, n_cells, d_slots.p, g.n_slots, d_cell_count.p,
                             d_cell_off.p, d_lvl_total.p, d_lvl_off.p, d_dense.p, dense_cap, stream))) return rc;
    const int n_tasks = n_images * L;
    DCS_MARK(ev_t[3], stream);
    last_tasks = n_tasks; host_copy_valid = false;
    DescribeParams dp{};
    for (int l = 0; l < L; ++l) { dp.scale[l] = t.scale[l]; dp.scaled_patch[l] = g.lv[l].scaled_patch; dp.out_base[l] = oct.lv[l].out_base; }
    for (int v = 0; v <= kHalfPatch; ++v) { dp.umax[v] = t.umax[v]; dp.umax_packed |= (unsigned long long)(t.umax[v] & 15) << (4 * v); }
    dp.out_per_image = oct.out_per_image; dp.nlevels = L; dp.ic_mask = d_ic_mask.p;
    if (device_octree) {
        // fully asynchronous: quadtree on the device, no host round trip
        if (cap < oct.out_per_image) {
            set_error("cap %d < %d (sum over levels of max(N_level + 8, 4 * initial nodes)): required by the device quadtree path", cap, oct.out_per_image);
            return DCS_ERR_CAPACITY;
        }
        OctScratch sc{d_oct_u64[0].p, d_oct_u8.p, d_oct_i32[0].p, d_oct_i32[1].p, d_oct_i32[2].p, d_oct_i32[3].p, d_oct_i32[4].p,
                      d_oct_i32[5].p, d_oct_i32[6].p, d_oct_i32[7].p, d_oct_u64[1].p, d_oct_u32[0].p, d_oct_u64[2].p, d_oct_u32[1].p};
        if ((rc = launch_octree(d_dense.p, d_lvl_off.p, oct, sc, n_tasks, (int)dense_cap, d_sel.p, d_lvl_cnt.p, d_oct_flag.p, stream))) return rc;
        DCS_MARK(ev_t[6], stream);
        host_us = -1.f;
        if (!fused_blur) DCS_HIP(hipStreamWaitEvent(stream, ev_blur, 0));
        DCS_MARK(ev_t[4], stream);
        if ((rc = launch_describe(raw, blur, dp, d_sel.p, nullptr, d_lvl_cnt.p, n_images, oct.out_per_image, d_kp_out, d_desc_out, cap,
                                  d_n_out, stream, d_lvl_off.p + n_tasks, (int)dense_cap, fused_blur))) return rc;
    } else {
        DCS_HIP(hipMemcpyAsync(h_lvl_off.p, d_lvl_off.p, sizeof(int32_t) * (n_tasks + 1), hipMemcpyDeviceToHost, stream));
        DCS_HIP(hipStreamSynchronize(stream));
        const size_t total = (size_t)h_lvl_off.p[n_tasks];
        if (total > dense_cap) { set_error("FAST candidates (%zu) exceed the dense buffer (%zu)", total, dense_cap); return DCS_ERR_CAPACITY; }
        if (total) {
            DCS_HIP(hipMemcpyAsync(h_dense.p, d_dense.p, sizeof(dcs_candidate) * total, hipMemcpyDeviceToHost, stream));
            DCS_HIP(hipStreamSynchronize(stream));
        }
        host_copy_valid = true;
        // host quadtree per (image, level)
        const auto t0 = std::chrono::steady_clock::now();
        pool->parallel_for(n_tasks, [&](int k) {
            const int l = k % L;
            const int b = h_lvl_off.p[k], e = h_lvl_off.p[k + 1];
            distribute_octree(h_dense.p + b, e - b, oct.lv[l].width, oct.lv[l].height, t.n_per_level[l], task_out[k]);
        });
        size_t n_sel = 0;
        int max_per_image = 0;
        bool over = false;
        for (int i = 0; i < n_images; ++i) {
            h_img_off.p[i] = (int32_t)n_sel;
            for (int l = 0; l < L; ++l) {
                for (const dcs_candidate& c : task_out[(size_t)i * L + l]) {
                    if (n_sel >= h_sel.n) { over = true; break; }
                    SelKp sk;
                    sk.x = (int16_t)(c.x + kMinBorder); sk.y = (int16_t)(c.y + kMinBorder);
                    sk.score = (int16_t)c.score; sk.level = (int8_t)l; sk.pad = 0;
                    h_sel.p[n_sel++] = sk;
                }
            }
            const int n_i = (int)(n_sel - h_img_off.p[i]);
            if (h_counts) h_counts[i] = n_i;
            max_per_image = std::max(max_per_image, n_i);
        }
        h_img_off.p[n_images] = (int32_t)n_sel;
        host_us = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (over) { set_error("selected keypoints exceed internal capacity"); return DCS_ERR_CAPACITY; }
        if (max_per_image > cap) {
            set_error("an image yields %d keypoints but cap is %d", max_per_image, cap);
            return DCS_ERR_CAPACITY;
        }
        if (n_sel) DCS_HIP(hipMemcpyAsync(d_sel.p, h_sel.p, sizeof(SelKp) * n_sel, hipMemcpyHostToDevice, stream));
        DCS_HIP(hipMemcpyAsync(d_img_off.p, h_img_off.p, sizeof(int32_t) * (n_images + 1), hipMemcpyHostToDevice, stream));
        if (!fused_blur) DCS_HIP(hipStreamWaitEvent(stream, ev_blur, 0));
        DCS_MARK(ev_t[4], stream);
        if ((rc = launch_describe(raw, blur, dp, d_sel.p, d_img_off.p, nullptr, n_images, max_per_image, d_kp_out, d_desc_out, cap,
                                  d_n_out, stream, nullptr, 0, fused_blur))) return rc;
    }
    DCS_MARK(ev_t[5], stream);
    es.host_us = host_us; es.pending = timed;
    timing_valid = timed;
    return DCS_OK;
#undef DCS_MARK
#undef DCS_MARK_FAST
}

int dcs_orb::harvest(EvSet& es)
{
    float ms, us[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!es.all) {                                           // timing mode 1: only the FAST stage was bracketed
        DCS_HIP(hipEventSynchronize(es.t[2]));
        DCS_HIP(hipEventElapsedTime(&ms, es.t[1], es.t[2])); us[1] = ms * 1000.f;
    } else {
        DCS_HIP(hipEventSynchronize(es.t[5]));
        DCS_HIP(hipEventElapsedTime(&ms, es.t[0], es.t[1])); us[0] = ms * 1000.f;   // resize chain
        DCS_HIP(hipEventElapsedTime(&ms, es.t[1], es.t[2])); us[1] = ms * 1000.f;   // k_fast_cells
        if (es.has_f) {
            DCS_HIP(hipEventSynchronize(es.f[1]));
            DCS_HIP(hipEventElapsedTime(&ms, es.f[0], es.f[1])); us[1] += ms * 1000.f;  // + its early launch (levels [0, fast_split)) on s_fast
        }
        DCS_HIP(hipEventElapsedTime(&ms, es.t[2], es.t[3])); us[2] = ms * 1000.f;   // scan + offsets + gather
        if (es.has_b) {
            DCS_HIP(hipEventSynchronize(es.b[1]));
            DCS_HIP(hipEventElapsedTime(&ms, es.b[0], es.b[1])); us[3] = ms * 1000.f;   // k_blur (aux stream)
        }
        if (es.dev_oct) { DCS_HIP(hipEventElapsedTime(&ms, es.t[3], es.t[6])); us[4] = ms * 1000.f; }   // k_octree
        else us[4] = es.host_us;                                                    // host quadtree (wall)
        DCS_HIP(hipEventElapsedTime(&ms, es.t[4], es.t[5])); us[5] = ms * 1000.f;   // k_describe
        DCS_HIP(hipEventElapsedTime(&ms, es.t[0], es.t[5])); us[6] = ms * 1000.f;   // whole call on the main stream
    }
    for (int i = 0; i < 7; ++i) { last_us[i] = us[i]; sum_us[i] += us[i]; }
    ++n_harvested;
    es.pending = false;
    return DCS_OK;
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int dcs_orb_create(const dcs_orb_params* p, dcs_orb** out)
{
    if (!p || !out) { set_error("null argument"); return DCS_ERR_INVALID; }
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > kMaxLevels || p->nfeatures < 1 || p->min_th_fast < 1 ||
        p->ini_th_fast < p->min_th_fast || p->ini_th_fast > 254 || !(p->scale_factor > 1.0f)) {
        set_error("bad ORB parameters");
        return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (p->device >= 0) DCS_HIP(hipSetDevice(p->device));
    std::unique_ptr<dcs_orb> h(new dcs_orb);
    h->prm = *p;
    if (h->prm.max_images < 1) h->prm.max_images = 1;
    DCS_HIP(hipGetDevice(&h->device));
    h->t.build(p->nfeatures, p->scale_factor, p->nlevels, p->ini_th_fast, p->min_th_fast);
    DCS_HIP(hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking));
    DCS_HIP(hipEventCreateWithFlags(&h->ev_pyr, hipEventDisableTiming));
    DCS_HIP(hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming));
    DCS_HIP(hipEventCreateWithFlags(&h->ev_lvl, hipEventDisableTiming));
    DCS_HIP(hipEventCreateWithFlags(&h->ev_fast_early, hipEventDisableTiming));
    // options are copied when the handle is created (config.h): dcs_option_set between two dcs_orb_create calls gives two handles that differ
    h->fast_split = (int)opt(OPT_ORB_FAST_SPLIT);
    for (auto& es : h->ring) { for (auto& e : es.t) DCS_HIP(hipEventCreate(&e)); for (auto& e : es.b) DCS_HIP(hipEventCreate(&e)); for (auto& e : es.f) DCS_HIP(hipEventCreate(&e)); }
    h->no_overlap = opt(OPT_ORB_NO_OVERLAP) != 0;
    h->fused_mode = opt(OPT_ORB_FUSED_BLUR) < 0 ? -1 : (opt(OPT_ORB_FUSED_BLUR) != 0);
    h->opt_dense_cap = opt(OPT_ORB_DENSE_CAP);
    h->emit_mode = (int)opt(OPT_ORB_EMIT);
    {   // the probe runs once per device and process; DCS_FAST_HW_PROBE=fail (read per handle: tests) makes this handle take the fallback
        static std::mutex probe_m;
        static int probed[64] = {0};                   // 0 = not yet, 1 = fast forms, 2 = plain forms
        std::lock_guard<std::mutex> g(probe_m);
        const int di = h->device >= 0 && h->device < 64 ? h->device : 0;
        if (!probed[di]) {
            bool ok = false;
            char why[160] = "";
            if ((rc = fast_hw_probe(&ok, why, sizeof why))) return rc;
            probed[di] = ok ? 1 : 2;
            if (!ok) fprintf(stderr, "[dcs] device %d: %s -- k_fast_cells runs its plain forms (byte loads, ballot append)\n", h->device, why);
        }
        h->fast_hw = probed[di] == 1 && opt(OPT_FAST_HW_PROBE_FAIL) == 0;
    }     // 0: round-4 resize chain; n > 0: the cells of levels [0, n) emit; unset: all of them
    h->device_octree = p->host_threads <= 0;          // host_threads > 0 selects the host quadtree with that many workers
    h->pool.reset(new Pool(std::max(0, p->host_threads - 1)));
    {   // staging threads of the host-buffer API (DCS_ORB_STAGING_THREADS, default 4; 1 = pack on the calling thread)
        const int nt = opt(OPT_ORB_STAGING_THREADS) > 0 ? (int)opt(OPT_ORB_STAGING_THREADS) : (h->prm.max_images >= 64 ? 8 : 4);    // 8 packers keep up with PCIe 5 (~45 GB/s of image bytes) on large batches
        if (nt > 1 && h->prm.max_images > 2) h->stage_pool.reset(new Pool(nt - 1));
    }
    *out = h.release();
    return DCS_OK;
}

void dcs_orb_destroy(dcs_orb* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    delete h;
}

int dcs_orb_tables(const dcs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* n_per_level)
{
    if (!h) { set_error("null handle"); return DCS_ERR_INVALID; }
    for (int i = 0; i < h->t.nlevels; ++i) {
        if (scale) scale[i] = h->t.scale[i];
        if (inv_scale) inv_scale[i] = h->t.inv_scale[i];
        if (sigma2) sigma2[i] = h->t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->t.inv_sigma2[i];
        if (n_per_level) n_per_level[i] = h->t.n_per_level[i];
    }
    return DCS_OK;
}

int dcs_debug_sincosf(const float* x, int n, float* cos_out, float* sin_out)
{
    if (n < 0 || (n && (!x || !cos_out || !sin_out))) { set_error("bad argument"); return DCS_ERR_INVALID; }
    int rc = ensure_device();
    if (rc || n == 0) return rc;
    Scratch s;
    const float* d_x; float *d_c, *d_s;
    if ((rc = s.upload(&d_x, x, (size_t)n)) || (rc = s.alloc(&d_c, (size_t)n)) || (rc = s.alloc(&d_s, (size_t)n))) return rc;
    if ((rc = launch_debug_sincosf(d_x, n, d_c, d_s, s.st))) return rc;
    if ((rc = s.download_bytes(cos_out, d_c, sizeof(float) * n))) return rc;
    if ((rc = s.download_bytes(sin_out, d_s, sizeof(float) * n))) return rc;
    return s.finish();
}

int dcs_orb_required_cap(const dcs_orb* h, int rows, int cols, int* cap)
{
    if (!h || !cap || rows < 1 || cols < 1) { set_error("dcs_orb_required_cap: bad arguments"); return DCS_ERR_INVALID; }
    PyramidGeom g;
    g.build(h->t, rows, cols);
    int total = 0;
    for (int l = 0; l < h->t.nlevels; ++l)
        total += octree_list_bound((g.lv[l].w - kEdgeThreshold + 3) - kMinBorder, (g.lv[l].h - kEdgeThreshold + 3) - kMinBorder, h->t.n_per_level[l]);
    *cap = total;
    return DCS_OK;
}

int dcs_orb_extract_batch_device(dcs_orb* h, const uint8_t* d_images, int n_images, int rows, int cols, int stride,
                                 dcs_keypoint* d_kp, uint8_t* d_desc, int cap, int32_t* d_n_out, void* stream)
{
    if (!h || !d_images || !d_kp || !d_desc || !d_n_out || n_images < 1 || cap < 1 || stride < cols) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    if (n_images > h->prm.max_images) { set_error("n_images %d > max_images %d", n_images, h->prm.max_images); return DCS_ERR_INVALID; }
    DCS_HIP(hipSetDevice(h->device));
    int rc = h->configure(rows, cols);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;            // NULL = the legacy default stream, exactly what the caller passed
    return h->run(d_images, (size_t)rows * stride, stride, n_images, d_kp, d_desc, cap, d_n_out, s, nullptr);
}

int dcs_orb_extract_batch(dcs_orb* h, const uint8_t* const* images, int n_images, int rows, int cols, int stride,
                          dcs_keypoint* kp, uint8_t* desc, int cap, int* n_out)
{
    if (!h || !n_out || n_images < 1) { set_error("bad argument"); return DCS_ERR_INVALID; }
    for (int i = 0; i < n_images; ++i) n_out[i] = 0;
    if (!images || rows <= 0 || cols <= 0) return DCS_OK;          // _image.empty() (:1046-1047)
    for (int i = 0; i < n_images; ++i) if (!images[i]) return DCS_OK;
    if (!kp || !desc || cap < 1 || stride < cols) { set_error("bad argument"); return DCS_ERR_INVALID; }
    if (n_images > h->prm.max_images) { set_error("n_images %d > max_images %d", n_images, h->prm.max_images); return DCS_ERR_INVALID; }
    DCS_HIP(hipSetDevice(h->device));
    int rc = h->configure(rows, cols);
    if (rc) return rc;
    // Host images -> pinned staging (rows packed at a 4-byte aligned pitch) -> ONE device staging buffer that the kernels
    // read in place as level 0, exactly like the _device entry point.
    // Images that already sit in page-locked memory (dcs_host_alloc, hipHostMalloc / hipHostRegister of the caller's frame ring) at equal
    // spacing and a 4-byte aligned stride need no staging copy: the DMA reads the caller's memory and the kernels take the caller's
    // stride as the pitch of level 0. Two pointer queries, remembered per handle; pageable images take the packing path below. DCS_ORB_HOST_DIRECT=0 disables.
    bool direct = false;
    size_t spacing = (size_t)rows * stride;
    if (stride % 4 == 0 && (reinterpret_cast<uintptr_t>(images[0]) & 3) == 0 && opt(OPT_ORB_HOST_DIRECT) != 0) {
        bool even = true;
        if (n_images > 1) {
            // equally spaced AND close: the spacing decides the size of the device staging and of the one DMA that reads the whole range,
            // so it is capped at two images' worth (side-by-side frames, a ring with a header per frame); anything wider is packed
            even = images[1] > images[0] && (size_t)(images[1] - images[0]) >= spacing && (size_t)(images[1] - images[0]) <= 2 * spacing &&
                   ((images[1] - images[0]) & 3) == 0;
            const size_t d = even ? (size_t)(images[1] - images[0]) : 0;
            for (int i = 2; i < n_images && even; ++i) even = images[i] == images[0] + (size_t)i * d;
            if (even) spacing = d;
        }
        if (even) {
            // The WHOLE range [first byte of image 0, last byte of the last image] must lie inside ONE page-locked allocation: the DMA reads
            // the gaps between the images too, and two frames that were pinned separately (two hipHostRegister'd cv::Mats, two dcs_host_alloc
            // blocks) are "equally spaced" by construction while the memory between them may be unregistered or unmapped. The allocation's
            // base and size come from the runtime; a range it cannot vouch for is packed like pageable memory.
            auto locked_range = [](const uint8_t* lo, const uint8_t* hi) {
                hipPointerAttribute_t a;
                if (hipPointerGetAttributes(&a, lo) != hipSuccess) { (void)hipGetLastError(); return false; }
                if (a.type != hipMemoryTypeHost) return false;
                void* base = nullptr; size_t size = 0;
                if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)lo) != hipSuccess ||
                    hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)lo) != hipSuccess || !base) {
                    (void)hipGetLastError();
                    return false;
                }
                const uint8_t* b = static_cast<const uint8_t*>(base);
                return lo >= b && hi <= b + size;
            };
            // the queries cost tens of microseconds (more for a large block): the handle remembers the last range it found page-locked -- a
            // frame ring is handed over again and again. (A stale entry -- the block freed and the addresses reused by pageable memory --
            // costs speed only: hipMemcpyAsync stages pageable sources itself.)
            const uint8_t* lo = images[0];
            const uint8_t* hi = images[n_images - 1] + (size_t)rows * stride;
            if (lo >= h->locked_lo && hi <= h->locked_hi) direct = true;
            else if (lo == h->pageable_lo && hi == h->pageable_hi) direct = false;           // asked before: pageable (a caller that re-uses its buffers)
            else if ((direct = locked_range(lo, hi))) { h->locked_lo = lo; h->locked_hi = hi; }
            else { h->pageable_lo = lo; h->pageable_hi = hi; }
        }
    }
    h->last_host_path = direct ? 1 : 0; h->last_small_graph = 0;
    const int pitch_s = direct ? stride : (cols + 3) & ~3;
    const size_t img_bytes = direct ? spacing : (size_t)rows * pitch_s;
    // (direct: the last image is copied up to its last row only -- the caller's block need not extend to a full spacing behind it)
    const size_t last_img_bytes = direct ? (size_t)rows * stride : img_bytes;
    if ((!direct && (rc = h->h_img.resize(img_bytes * n_images))) || (rc = h->d_stage.resize(img_bytes * n_images))) return rc;
    const uint8_t* const up_src = direct ? images[0] : h->h_img.p;
    auto up_bytes = [&](int i0, int m) { return img_bytes * (size_t)(m - 1) + (i0 + m == n_images ? last_img_bytes : img_bytes); };
    const bool nopack = direct;
    auto pack = [&](int i) {
        if (nopack) return;
        uint8_t* dst = h->h_img.p + i * img_bytes;
        if (stride == cols && pitch_s == cols) memcpy(dst, images[i], img_bytes);
        else for (int y = 0; y < rows; ++y) memcpy(dst + (size_t)y * pitch_s, images[i] + (size_t)y * stride, cols);
    };
    // ---- large batches (what a host that buffers frames hands over): a three-stage pipeline over chunks of images. While the staging
    // threads pack chunk k + 1 into pinned memory and its DMA runs on the upload stream, the kernels of chunk k run on the main stream
    // and the slots of chunk k - 1 come down on the download stream and are scattered into the caller's arrays: the call costs the
    // slowest stage (PCIe: 157 MB of images per 512) instead of the sum. Results are the same bytes as the one-shot path (every chunk
    // is an ordinary extraction of its images).
    // images per chunk: 96 from 192 images, half the call from 128 (0 = one-shot path; read per call: tests compare the two paths).
    // Measured per 512 images of 640 x 480 on one box, three alternating rounds: 64 -> 4.90 / 4.94 / 6.04 ms, 96 -> 4.22 / 4.24 / 5.79,
    // 80 -> 5.77 / 4.63 / 5.33, 128 -> 6.6, 170 -> 5.2, 32 -> 8.4 (the compute stream is the slow stage: every chunk is a complete
    // extraction, and small ones fill the chip badly; large ones start late and leave a long tail). A second extraction lane (odd
    // chunks on a twin handle with its own scratch and stream, underneath the even ones) was measured and changes nothing -- 4.24 / 4.75 /
    // 4.65 ms with two lanes against 5.13 / 4.72 / 4.68 with one: what the call waits for after the last image is packed (2.1 of
    // 4.3 ms) is the image DMA, 157 MB at an effective ~45 GB/s next to the packing threads' traffic, then one chunk's kernels, download and scatter.
    const bool chunk_s = opt(OPT_ORB_HOST_CHUNK) >= 0;                      // set explicitly (read per call)
    const int chunk_env = chunk_s ? (int)opt(OPT_ORB_HOST_CHUNK) : std::min(96, n_images / 2);
    if (h->device_octree && chunk_env > 0 && n_images >= 2 * chunk_env && (chunk_s || n_images >= 128)) {
        const int C = chunk_env;
        const size_t slots = (size_t)n_images * cap;
        if ((rc = h->d_kp.resize(slots)) || (rc = h->d_desc.resize(slots * 32)) || (rc = h->d_n.resize(n_images)) || (rc = h->h_n.resize(n_images)) ||
            (rc = h->h_kp_out.resize(slots)) || (rc = h->h_desc_out.resize(slots * 32))) return rc;
        if ((rc = h->side_stream(h->s_h2d, h->s_main)) || (rc = h->side_stream(h->s_d2h, h->s_main, h->s_h2d))) return rc;
        // Chunk boundaries: a short first chunk starts the DMA early, a short last one keeps the tail (kernels + download + scatter of
        // the final chunk, which nothing overlaps) small.
        std::vector<int> c_begin;
        {
            int i = 0;
            const int head = std::max(8, C / 4);
            c_begin.push_back(0);
            if (n_images > 4 * C) { i = head; c_begin.push_back(i); }
            while (n_images - i > C + head) { i += C; c_begin.push_back(i); }
            if (n_images - i > head && n_images > 4 * C) { c_begin.push_back(n_images - head); }
            c_begin.push_back(n_images);
        }
        const int NC = (int)c_begin.size() - 1;
        while (h->ev_chunk.size() < (size_t)3 * NC) { hipEvent_t e; DCS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_chunk.push_back(e); }
        const bool trace = opt(OPT_ORB_HOST_TRACE) != 0;           // stderr: host time per stage of this call
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_call = now();
        // The calling thread only ENQUEUES (copies, ~25 launches and 3 events per chunk: ~0.13 ms of API time each). A helper thread owns
        // the staging pool: it packs the chunks in order into pinned memory (flag packed[k]), then scatters every chunk that has landed
        // (flag enqueued[k] -> its download event) into the caller's arrays -- packing, DMA, kernels and scattering all overlap.
        std::vector<std::atomic<int>> packed(NC), enqueued(NC);
        for (int k = 0; k < NC; ++k) { packed[k].store(0); enqueued[k].store(0); }
        std::atomic<int> failed{DCS_OK}, abort_flag{0};
        double t_pack = 0, t_scatter = 0;
        auto scatter = [&](int k) {                                // chunk k has landed in pinned memory: hand its valid prefixes to the caller
            const int i0 = c_begin[k], m = c_begin[k + 1] - i0;
            auto one = [&](int j) {
                const int i = i0 + j, c = h->h_n.p[i];
                if (c > 0) {
                    memcpy(kp + (size_t)i * cap, h->h_kp_out.p + (size_t)i * cap, sizeof(dcs_keypoint) * std::min(c, cap));
                    memcpy(desc + (size_t)i * cap * 32, h->h_desc_out.p + (size_t)i * cap * 32, (size_t)32 * std::min(c, cap));
                }
                n_out[i] = std::max(c, 0);
            };
            for (int j = 0; j < m; ++j) if (h->h_n.p[i0 + j] < 0) failed.store(h->h_n.p[i0 + j]);     // k_describe reports DCS_ERR_CAPACITY in place of a count
            if (h->stage_pool) h->stage_pool->parallel_for(m, one); else for (int j = 0; j < m; ++j) one(j);
        };
        const int dev = h->device;
        std::thread helper([&] {
            (void)hipSetDevice(dev);
            const auto tp0 = now();
            for (int k = 0; k < NC && !abort_flag.load(); ++k) {
                const int i0 = c_begin[k], m = c_begin[k + 1] - i0;
                if (h->stage_pool) h->stage_pool->parallel_for(m, [&](int j) { pack(i0 + j); }); else for (int j = 0; j < m; ++j) pack(i0 + j);
                packed[k].store(1, std::memory_order_release);
            }
            const auto tp1 = now();
            t_pack = ms(tp0, tp1);
            for (int k = 0; k < NC; ++k) {
                while (!enqueued[k].load(std::memory_order_acquire)) { if (abort_flag.load()) return; std::this_thread::yield(); }
                if (hipEventSynchronize(h->ev_chunk[3 * k + 2]) != hipSuccess) { failed.store(DCS_ERR_HIP); return; }
                scatter(k);
            }
            t_scatter = ms(tp1, now());
        });
        // on any error: the helper stops, and nothing may stay in flight that reads the pinned / device staging of this handle
        auto bail = [&](int code) {
            abort_flag.store(1);
            helper.join();
            (void)hipStreamSynchronize(h->s_h2d); (void)hipStreamSynchronize(h->s_main); (void)hipStreamSynchronize(h->s_d2h);
            return code;
        };
#define DCS_PIPE(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return bail(DCS_ERR_HIP); } } while (0)
        double t_enq = 0, t_wait_pack = 0;
        for (int k = 0; k < NC; ++k) {
            const int i0 = c_begin[k], m = c_begin[k + 1] - i0;
            hipEvent_t e_up = h->ev_chunk[3 * k], e_done = h->ev_chunk[3 * k + 1], e_dl = h->ev_chunk[3 * k + 2];
            const auto tw0 = now();
            while (!packed[k].load(std::memory_order_acquire)) std::this_thread::yield();
            const auto tw1 = now();
            t_wait_pack += ms(tw0, tw1);
            DCS_PIPE(hipMemcpyAsync(h->d_stage.p + i0 * img_bytes, up_src + i0 * img_bytes, up_bytes(i0, m), hipMemcpyHostToDevice, h->s_h2d));
            DCS_PIPE(hipEventRecord(e_up, h->s_h2d));
            DCS_PIPE(hipStreamWaitEvent(h->s_main, e_up, 0));
            if ((rc = h->run(h->d_stage.p + i0 * img_bytes, img_bytes, pitch_s, m, h->d_kp.p + (size_t)i0 * cap, h->d_desc.p + (size_t)i0 * cap * 32, cap,
                             h->d_n.p + i0, h->s_main, nullptr))) return bail(rc);
            DCS_PIPE(hipEventRecord(e_done, h->s_main));
            DCS_PIPE(hipStreamWaitEvent(h->s_d2h, e_done, 0));
            DCS_PIPE(hipMemcpyAsync(h->h_n.p + i0, h->d_n.p + i0, sizeof(int32_t) * m, hipMemcpyDeviceToHost, h->s_d2h));
            DCS_PIPE(hipMemcpyAsync(h->h_kp_out.p + (size_t)i0 * cap, h->d_kp.p + (size_t)i0 * cap, sizeof(dcs_keypoint) * (size_t)m * cap, hipMemcpyDeviceToHost, h->s_d2h));
            DCS_PIPE(hipMemcpyAsync(h->h_desc_out.p + (size_t)i0 * cap * 32, h->d_desc.p + (size_t)i0 * cap * 32, (size_t)m * cap * 32, hipMemcpyDeviceToHost, h->s_d2h));
            DCS_PIPE(hipEventRecord(e_dl, h->s_d2h));
            enqueued[k].store(1, std::memory_order_release);
            t_enq += ms(tw1, now());
        }
#undef DCS_PIPE
        const auto t_loop = now();
        helper.join();
        DCS_HIP(hipStreamSynchronize(h->s_main));                  // timing events of the last chunk
        if (trace) fprintf(stderr, "[dcs_orb_extract_batch] %d images in %d chunks: %.3f ms (calling thread: enqueue %.3f, waited for packing %.3f, loop done at %.3f; "
                                   "helper: pack %.3f, then scatter + waiting for downloads %.3f)\n", n_images, NC, ms(t_call, now()), t_enq, t_wait_pack, ms(t_call, t_loop), t_pack, t_scatter);
        if (failed.load() == DCS_ERR_CAPACITY) { set_error("FAST candidates exceed the dense buffer of the handle"); return DCS_ERR_CAPACITY; }
        if (failed.load()) { set_error("host pipeline: a download did not complete"); return failed.load(); }
        return DCS_OK;
    }
    // ---- small batches: groups of 8 images are packed by the staging threads and their DMA (one copy per group) is enqueued while the
    // next group is packed; one launch sequence, one synchronisation.
    for (int i0 = 0; i0 < n_images; i0 += 8) {
        const int m = std::min(8, n_images - i0);
        if (m > 1 && h->stage_pool) h->stage_pool->parallel_for(m, [&](int k) { pack(i0 + k); });
        else for (int k = 0; k < m; ++k) pack(i0 + k);
        DCS_HIP(hipMemcpyAsync(h->d_stage.p + i0 * img_bytes, up_src + i0 * img_bytes, up_bytes(i0, m), hipMemcpyHostToDevice, h->s_main));
    }
    const size_t slots = (size_t)n_images * cap;
    if (h->device_octree) {
        // The results of the call live in ONE device block [counts | key points | descriptors] and come down in ONE copy behind the
        // kernels: one synchronisation for the whole call (a dual frame per call is latency-bound: every DMA operation counts).
        const size_t o_kp = ((sizeof(int32_t) * n_images + 255) & ~(size_t)255), o_desc = o_kp + ((sizeof(dcs_keypoint) * slots + 255) & ~(size_t)255),
                     total = o_desc + slots * 32;
        if ((rc = h->d_out.resize(total)) || (rc = h->h_out.resize(total))) return rc;
        int32_t* d_cnt = reinterpret_cast<int32_t*>(h->d_out.p);
        dcs_keypoint* d_kps = reinterpret_cast<dcs_keypoint*>(h->d_out.p + o_kp);
        uint8_t* d_dsc = h->d_out.p + o_desc;
        // One frame (or dual frame) per call is bound by the HOST's launch rate: ~15 launches of 4-20 us kernels, and the queue runs dry
        // between them. The launches of such a call are the same every time (same buffers, same shapes), so from the third call of a shape on
        // they are replayed as one executable graph (captured from the very code below; any failure to capture switches the handle back to
        // plain launches). DCS_ORB_SMALL_GRAPH=0 / 1 forces it off / on.
        const int graph_env = (int)opt(OPT_ORB_SMALL_GRAPH);                 // read per call (tests switch it)
        const bool graph_ok = (graph_env < 0 ? kSmallGraphDefault : graph_env != 0) && n_images <= 2 && !h->no_overlap && !h->small_graph_broken;
        const dcs_orb::SmallGraphKey key{h->d_stage.p, h->d_out.p, h->h_out.p, img_bytes, total, n_images, rows, cols, pitch_s, cap, h->config_generation};
        bool replayed = false;
        if (graph_ok && h->small_graph_exec && key == h->small_graph_key) {
            DCS_HIP(hipGraphLaunch(h->small_graph_exec, h->s_main));
            replayed = true;
        } else if (graph_ok && key == h->small_graph_key && ++h->small_graph_seen >= 2) {
            if (h->small_graph_exec) { (void)hipGraphExecDestroy(h->small_graph_exec); h->small_graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(h->s_main, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int rc_run = h->run(h->d_stage.p, img_bytes, pitch_s, n_images, d_kps, d_dsc, cap, d_cnt, h->s_main, nullptr);
                const hipError_t e_cp = hipMemcpyAsync(h->h_out.p, h->d_out.p, total, hipMemcpyDeviceToHost, h->s_main);
                const hipError_t e_end = hipStreamEndCapture(h->s_main, &g);
                ok = rc_run == DCS_OK && e_cp == hipSuccess && e_end == hipSuccess && g != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&h->small_graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
            if (ok) ok = hipGraphLaunch(h->small_graph_exec, h->s_main) == hipSuccess;
            if (ok) replayed = true;
            else {                                              // back to plain launches for good; nothing of the capture has run
                (void)hipGetLastError();
                if (h->small_graph_exec) { (void)hipGraphExecDestroy(h->small_graph_exec); h->small_graph_exec = nullptr; }
                h->small_graph_broken = true;
            }
        } else if (!(key == h->small_graph_key)) {
            h->small_graph_key = key; h->small_graph_seen = 0;
            if (h->small_graph_exec) { (void)hipGraphExecDestroy(h->small_graph_exec); h->small_graph_exec = nullptr; }
        }
        h->last_small_graph = replayed ? 1 : 0;
        if (!replayed) {
            if ((rc = h->run(h->d_stage.p, img_bytes, pitch_s, n_images, d_kps, d_dsc, cap, d_cnt, h->s_main, nullptr))) return rc;
            DCS_HIP(hipMemcpyAsync(h->h_out.p, h->d_out.p, total, hipMemcpyDeviceToHost, h->s_main));
        }
        DCS_HIP(hipStreamSynchronize(h->s_main));
        const int32_t* cnt = reinterpret_cast<const int32_t*>(h->h_out.p);
        for (int i = 0; i < n_images; ++i)
            if (cnt[i] < 0) { set_error("FAST candidates exceed the dense buffer of the handle (%zu)", h->dense_cap); return cnt[i]; }      // k_describe reports DCS_ERR_CAPACITY in place of a count
        for (int i = 0; i < n_images; ++i) {
            const int c = std::min(cnt[i], cap);
            if (c) {
                memcpy(kp + (size_t)i * cap, h->h_out.p + o_kp + sizeof(dcs_keypoint) * (size_t)i * cap, sizeof(dcs_keypoint) * c);
                memcpy(desc + (size_t)i * cap * 32, h->h_out.p + o_desc + (size_t)i * cap * 32, (size_t)32 * c);
            }
            n_out[i] = c;
        }
        return DCS_OK;
    }
    if ((rc = h->d_kp.resize(slots)) || (rc = h->d_desc.resize(slots * 32)) || (rc = h->d_n.resize(n_images)) || (rc = h->h_n.resize(n_images))) return rc;
    std::vector<int> counts(n_images, 0);
    rc = h->run(h->d_stage.p, img_bytes, pitch_s, n_images, h->d_kp.p, h->d_desc.p, cap, h->d_n.p, h->s_main, counts.data());
    if (rc) { for (int i = 0; i < n_images; ++i) n_out[i] = counts[i]; return rc; }
    // host-quadtree mode: counts are known on the host already
    size_t total = 0;
    for (int i = 0; i < n_images; ++i) total += (size_t)counts[i];
    if ((rc = h->h_kp_out.resize(total)) || (rc = h->h_desc_out.resize(total * 32))) return rc;
    size_t at = 0;
    for (int i = 0; i < n_images; ++i) {
        if (!counts[i]) continue;
        DCS_HIP(hipMemcpyAsync(h->h_kp_out.p + at, h->d_kp.p + (size_t)i * cap, sizeof(dcs_keypoint) * counts[i], hipMemcpyDeviceToHost, h->s_main));
        DCS_HIP(hipMemcpyAsync(h->h_desc_out.p + at * 32, h->d_desc.p + (size_t)i * cap * 32, (size_t)32 * counts[i], hipMemcpyDeviceToHost, h->s_main));
        at += (size_t)counts[i];
    }
    DCS_HIP(hipStreamSynchronize(h->s_main));
    at = 0;
    for (int i = 0; i < n_images; ++i) {
        if (counts[i]) {
            memcpy(kp + (size_t)i * cap, h->h_kp_out.p + at, sizeof(dcs_keypoint) * counts[i]);
            memcpy(desc + (size_t)i * cap * 32, h->h_desc_out.p + at * 32, (size_t)32 * counts[i]);
            at += (size_t)counts[i];
        }
        n_out[i] = counts[i];
    }
    return DCS_OK;
}

int dcs_orb_extract(dcs_orb* h, const uint8_t* image, int rows, int cols, int stride, dcs_keypoint* kp, uint8_t* desc,
                    int cap, int* n_out)
{
    const uint8_t* imgs[1] = {image};
    return dcs_orb_extract_batch(h, imgs, 1, rows, cols, stride, kp, desc, cap, n_out);
}

int dcs_orb_debug_level_dims(const dcs_orb* h, int level, int* w, int* h_out)
{
    if (!h || !h->configured || level < 0 || level >= h->t.nlevels) { set_error("bad argument"); return DCS_ERR_INVALID; }
    *w = h->g.lv[level].w; *h_out = h->g.lv[level].h;
    return DCS_OK;
}

int dcs_orb_debug_level(dcs_orb* h, int image, int level, int blurred, uint8_t* dst)
{
    if (!h || !h->configured || level < 0 || level >= h->t.nlevels || image < 0 || image >= h->last_n_images || !dst) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    DCS_HIP(hipSetDevice(h->device));
    DCS_HIP(hipDeviceSynchronize());
    if (blurred && !h->last_blur_valid) {           // the product path never needed it (the caller's level-0 buffer must still be alive)
        int r = launch_blur(h->last_raw, h->last_blur, h->last_n_images, h->s_main);
        if (r) return r;
        DCS_HIP(hipDeviceSynchronize());
        h->last_blur_valid = true;
    }
    const LevelView& v = blurred ? h->last_blur.lv[level] : h->last_raw.lv[level];
    DCS_HIP(hipMemcpy2D(dst, v.w, v.base + (size_t)image * v.img_stride, v.pitch, v.w, v.h, hipMemcpyDeviceToHost));
    return DCS_OK;
}

int dcs_orb_debug_quadtree_fallbacks(dcs_orb* h, int* n)
{
    if (!h || !h->configured || !n) { set_error("bad argument"); return DCS_ERR_INVALID; }
    *n = 0;
    if (!h->device_octree || h->last_tasks <= 0) return DCS_OK;
    DCS_HIP(hipSetDevice(h->device));
    DCS_HIP(hipDeviceSynchronize());
    std::vector<int32_t> f((size_t)h->last_tasks);
    DCS_HIP(hipMemcpy(f.data(), h->d_oct_flag.p, sizeof(int32_t) * f.size(), hipMemcpyDeviceToHost));
    for (int32_t v : f) *n += v != 0;
    return DCS_OK;
}

int dcs_orb_debug_emit_levels(const dcs_orb* h, int* levels)
{
    if (!h || !levels) { set_error("dcs_orb_debug_emit_levels: null argument"); return DCS_ERR_INVALID; }
    *levels = h->last_emit_levels;
    return DCS_OK;
}

int dcs_orb_debug_fast_hw(const dcs_orb* h, int* fast_hw)
{
    if (!h || !fast_hw) { set_error("dcs_orb_debug_fast_hw: null argument"); return DCS_ERR_INVALID; }
    *fast_hw = h->fast_hw ? 1 : 0;
    return DCS_OK;
}

int dcs_orb_debug_host_path(const dcs_orb* h, int* direct, int* graph_replayed)
{
    if (!h || h->last_host_path < 0) { set_error("dcs_orb_debug_host_path: no host-buffer call yet"); return DCS_ERR_INVALID; }
    if (direct) *direct = h->last_host_path;
    if (graph_replayed) *graph_replayed = h->last_small_graph;
    return DCS_OK;
}

int dcs_orb_debug_candidates(dcs_orb* h, int image, int level, dcs_candidate* dst, int cap, int* n)
{
    if (!h || !h->configured || level < 0 || level >= h->t.nlevels || image < 0 || image >= h->last_n_images || !n) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    if (!h->host_copy_valid) {
        DCS_HIP(hipSetDevice(h->device));
        DCS_HIP(hipDeviceSynchronize());
        DCS_HIP(hipMemcpy(h->h_lvl_off.p, h->d_lvl_off.p, sizeof(int32_t) * (h->last_tasks + 1), hipMemcpyDeviceToHost));
        const size_t total = std::min((size_t)h->h_lvl_off.p[h->last_tasks], h->dense_cap);
        if (total) DCS_HIP(hipMemcpy(h->h_dense.p, h->d_dense.p, sizeof(dcs_candidate) * total, hipMemcpyDeviceToHost));
        h->host_copy_valid = true;
    }
    const int k = image * h->t.nlevels + level;
    const int b = h->h_lvl_off.p[k], e = h->h_lvl_off.p[k + 1];
    *n = e - b;
    if (dst) memcpy(dst, h->h_dense.p + b, sizeof(dcs_candidate) * std::min(cap, e - b));
    return DCS_OK;
}

int dcs_orb_last_timing(dcs_orb* h, float* us7)
{
    if (!h || !us7 || !h->timing_valid) { set_error("no timing available"); return DCS_ERR_INVALID; }
    dcs_orb::EvSet& es = h->ring[(h->n_calls - 1) % dcs_orb::kRing];
    int rc;
    if (es.pending && (rc = h->harvest(es))) return rc;
    for (int i = 0; i < 7; ++i) us7[i] = h->last_us[i];
    return DCS_OK;
}

int dcs_orb_set_timing(dcs_orb* h, int mode)
{
    if (!h || mode < 0 || mode > 2) { set_error("timing mode must be 0 (off), 1 (FAST stage only) or 2 (every stage)"); return DCS_ERR_INVALID; }
    int rc;
    for (auto& es : h->ring) if (es.pending && (rc = h->harvest(es))) return rc;      // sets recorded under the old mode are read under it
    h->timing_mode = mode;
    return DCS_OK;
}

int dcs_orb_timing_totals(dcs_orb* h, double* sum_us7, int64_t* n_calls, int reset)
{
    if (!h || !sum_us7 || !n_calls) { set_error("null argument"); return DCS_ERR_INVALID; }
    int rc;
    for (auto& es : h->ring) if (es.pending && (rc = h->harvest(es))) return rc;
    for (int i = 0; i < 7; ++i) sum_us7[i] = h->sum_us[i];
    *n_calls = h->n_harvested;
    if (reset) { for (auto& v : h->sum_us) v = 0; h->n_harvested = 0; }
    return DCS_OK;
}

int dcs_distribute_octree(const dcs_candidate* cand, int n, int min_x, int max_x, int min_y, int max_y, int n_target,
                          dcs_candidate* out, int cap, int* n_out)
{
    if (!n_out || n < 0 || (n && !cand)) { set_error("bad argument"); return DCS_ERR_INVALID; }
    std::vector<dcs_candidate> sel;
    const int m = distribute_octree(cand, n, max_x - min_x, max_y - min_y, n_target, sel);
    *n_out = m;
    if (m > cap) return DCS_ERR_CAPACITY;
    if (m && out) memcpy(out, sel.data(), sizeof(dcs_candidate) * m);
    return DCS_OK;
}

}  // extern "C"
