// orb_kernels.h -- launch interface of the extraction kernels (orb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcs_abi.h"
#include "orb_host.h"

namespace dcs {

// one pyramid level across a batch: image i lives at base + i * img_stride
struct LevelView {
    const uint8_t* base;
    size_t img_stride;
    int w, h, pitch;
};

struct LevelSet {                       // passed by value to kernels (<= 16 levels)
    LevelView lv[kMaxLevels];
    int nlevels;
};

// FAST cell descriptor (ORBextractor.cc:789-827): ROI origin/size in level coordinates, offset
// added to ROI-relative keypoint coordinates (j*wCell, i*hCell), output slots.
struct CellDesc {
    int16_t level, x0, y0, rw, rh, ox, oy, cap;
    int32_t slot_base;                  // first candidate slot of this cell inside one image
    // Round 5: the part of pyramid level `level + 1` this cell produces from its ROI in LDS (ComputePyramid :1107-1132 fused into the per-cell
    // FAST of :789-827): destination dwords [ekx0, ekx0 + enkx) x rows [edy0, edy0 + endy). A cell owns the destination dword whose FIRST
    // source column lies in [x0, next cell's x0) and the rows whose upper source row lies in [y0, next cell's y0): all taps then fall inside
    // its ROI (the ROI overlaps the next cell by 6 pixels). enkx == 0: nothing to emit. eG = 64 / enkx row groups per round, erounds = ceil(endy / eG), emul = ceil(65536 / enkx).
    int16_t ekx0, enkx, edy0, endy, eG, erounds;
    int32_t emul;
};

// what an emitting FAST launch needs besides its cells: the level it writes and that level's resize tables
// The frame of the next level that no cell's ROI reaches (its outer ~11 pixels: sources in the 13-pixel band FAST never loads) is written by
// extra workgroups of the same launch, one destination dword per lane straight from global memory (blockIdx.y >= n_cell_blocks): a separate
// k_resize launch for it costs a wave's latency (>= 13 us) per level whatever its size. Four rectangles; a block belongs to one of them.
struct FrameRect { int x4_begin, x4_count, row_begin, row_end, count, blk_begin; unsigned magic; };   // count = lanes (a lane = one dword column x 4 rows); magic = ceil(2^32 / x4_count)
struct FastEmit {
    LevelView dst;
    const int16_t* cols;                // {sx, 0, a0, a1} per destination column (16-byte aligned)
    const int32_t* rows;                // {sy, b0 | b1 << 16} per destination row
    FrameRect fr[4];
    int n_cell_blocks, n_frame_blocks, src_level;
};

// destination rectangle of a k_resize launch, in dwords (4 pixels) x rows
struct ResizeRect { int x4_begin, x4_count, row_begin, row_end; int blk_begin, nbx; };   // blk_begin / nbx: filled by launch_resize (first block of the rectangle in the 1-D grid, blocks per strip row)
struct ResizeRects { ResizeRect r[4]; int n; };

// keypoint selected by the quadtree, level coordinates (already + minBorder)
struct SelKp {
    int16_t x, y, score;
    int8_t level, pad;
};

struct DescribeParams {
    float scale[kMaxLevels];
    int scaled_patch[kMaxLevels];
    int umax[kHalfPatch + 1];
    unsigned long long umax_packed;      // umax[v] in bits [4v, 4v+4)
    // device-quadtree mode: keypoints of (image, level) live at sel[image * out_per_image + out_base[level] ...]
    int out_base[kMaxLevels];
    int out_per_image, nlevels;
    const uint32_t* ic_mask;             // device table built by build_ic_mask()
};

constexpr int kIcMaskWords = 256;        // 32 rows (31 + one empty) x 8 dwords: byte b of dword c of row r covers u = -15 + 4 c + b, v = r - 15
void build_ic_mask(const int* umax, uint32_t* out /* kIcMaskWords */);

// per-level parameters of the device quadtree (k_octree)
struct OctLevel { int width, height, n_target, out_base, out_cap; };
struct OctLevels { OctLevel lv[kMaxLevels]; int nlevels; int out_per_image; };
// HBM scratch of k_octree: every array has 2 * dense_cap elements
struct OctScratch {
    unsigned long long* keys; unsigned char* lcp;
    int *i0, *i1, *i2, *i3, *i4, *i5, *i6, *i7;
    unsigned long long* fkey; unsigned* fval; unsigned long long* tkey; unsigned* tval;
};

int launch_resize(const LevelView& src, const LevelView& dst, const int16_t* d_cols /* {sx, 0, a0, a1} per destination column */,
                  const int16_t* d_yofs, const int16_t* d_ya, int n_images, hipStream_t s,
                  const ResizeRects* rects = nullptr /* NULL: the whole level; else only these rectangles (the frame no FAST cell produces) */);

// what decides the LDS of a cell's workgroup in a launch: the largest ROI (pixel map: max_rh rows at the pitch class of max_rw), the
// longest survivor list and the largest score map among the launch's cells
struct FastFootprint { int max_rw = 7, max_rh = 7, list_entries = 0, sc_bytes = 0; };
void fast_footprint_add(FastFootprint& f, int rw, int rh);
int fast_cells_lds_bytes(const FastFootprint& f);
int launch_fast_cells(const LevelSet& levels, const CellDesc* d_cells, int n_cells, int n_images,
                      int ini_th, int min_th, dcs_candidate* d_slots, size_t slots_per_image,
                      int32_t* d_cell_count, const FastFootprint& fp, hipStream_t s,
                      int cell0 = 0, int n_launch = -1 /* the launch covers cells [cell0, cell0 + n_launch); -1: to the end */,
                      const FastEmit* emit = nullptr /* the cells of this launch (ONE level) also write their part of the next level */,
                      bool fast_hw = true /* the d16_hi score loads + the v_cmpx append (fast_hw_probe() said yes for this device) or the plain forms */);
// one-wave probe of the two hardware behaviours the fast form relies on; *ok = both behave, else why[] says which one does not
int fast_hw_probe(bool* ok, char* why, size_t why_len);

// per (image, level): scan the cell counts, then gather the slots into one dense array for the whole
// batch. d_lvl_off[n_images*nlevels + 1] = exclusive offsets (image-major, level-minor).
int launch_compact(const CellDesc* d_cells, const int32_t* d_level_cell_begin /* nlevels+1 */, int nlevels,
                   int n_images, int n_cells, const dcs_candidate* d_slots, size_t slots_per_image,
                   const int32_t* d_cell_count, int32_t* d_cell_off, int32_t* d_lvl_total,
                   int32_t* d_lvl_off, dcs_candidate* d_dense, size_t dense_cap, hipStream_t s);

int launch_blur(const LevelSet& src, const LevelSet& dst, int n_images, hipStream_t s);

// d_img_off != NULL: host-quadtree mode (sel dense, image i at [img_off[i], img_off[i+1]));
// d_lvl_cnt != NULL: device-quadtree mode (per (image, level) counts, layout in prm.out_base)
int launch_describe(const LevelSet& raw, const LevelSet& blurred, const DescribeParams& prm,
                    const SelKp* d_sel, const int32_t* d_img_off /* n_images+1 */, const int32_t* d_lvl_cnt, int n_images,
                    int max_per_image, dcs_keypoint* d_kp, uint8_t* d_desc, int cap, int32_t* d_n_out, hipStream_t s,
                    const int32_t* d_dense_total = nullptr /* device quadtree: total candidates of the batch */, int dense_cap = 0,
                    bool fused = false /* no blurred pyramid: k_describe blurs every patch itself */);

int launch_octree(const dcs_candidate* d_dense, const int32_t* d_lvl_off, const OctLevels& levels, const OctScratch& scratch,
                  int n_tasks, int dense_cap, SelKp* d_sel, int32_t* d_lvl_cnt, int32_t* d_need_general, hipStream_t s);

int upload_pattern();   // copies the rBRIEF table to constant memory of the current device

int launch_debug_sincosf(const float* d_x, int n, float* d_c, float* d_s, hipStream_t st);

}  // namespace dcs
