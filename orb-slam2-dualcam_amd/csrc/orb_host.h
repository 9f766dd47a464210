// orb_host.h -- host-side geometry, tables and the quadtree stage of the extractor.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/dcs_abi.h"

namespace dcs {

constexpr int kPatchSize = 31;       // ORBextractor.cc:72
constexpr int kHalfPatch = 15;       // ORBextractor.cc:73
constexpr int kEdgeThreshold = 19;   // ORBextractor.cc:74
constexpr int kMinBorder = kEdgeThreshold - 3;   // 16 (ORBextractor.cc:773)
constexpr int kMaxLevels = 16;

// ORBextractor ctor state (ORBextractor.cc:410-470)
struct OrbTables {
    int nfeatures = 0, nlevels = 0, ini_th = 0, min_th = 0;
    double scale_factor = 0;
    float scale[kMaxLevels], inv_scale[kMaxLevels], sigma2[kMaxLevels], inv_sigma2[kMaxLevels];
    int n_per_level[kMaxLevels];
    int umax[kHalfPatch + 1];
    void build(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
};

// per-level geometry for one image size (ComputePyramid :1107-1132, cell grid :769-806)
struct LevelGeom {
    int w = 0, h = 0, pitch = 0;
    size_t offset = 0;            // byte offset of this level inside one image's pyramid slab
    int n_cols = 0, n_rows = 0, w_cell = 0, h_cell = 0;
    int cell_base = 0;            // first cell id of this level in the per-image cell list
    int cell_cap = 0;             // candidate slots per cell (max strict local maxima)
    size_t slot_base = 0;         // first candidate slot of this level in the per-image slot array
    int scaled_patch = 0;         // (int)(31 * scale[level])  (:837)
};

struct PyramidGeom {
    int rows = 0, cols = 0, nlevels = 0;
    LevelGeom lv[kMaxLevels];
    size_t slab_bytes = 0;        // bytes of one image's pyramid (all levels, pitched)
    int n_cells = 0;              // cells per image over all levels
    size_t n_slots = 0;           // candidate slots per image over all levels
    void build(const OrbTables& t, int rows, int cols);
};

// fixed-point bilinear tables of cv::resize INTER_LINEAR 8UC1 (OpenCV 3.3/3.4.0)
struct ResizeTable {
    std::vector<int16_t> xofs, yofs;      // source index (x already clamped like OpenCV)
    std::vector<int16_t> xa, ya;          // 2 coefficients per destination index, 11 fractional bits
    void build(int sw, int sh, int dw, int dh);
};

// DistributeOctTree (ORBextractor.cc:539-763) as a sort/scan formulation (see octree.cpp).
// cand: level coordinates relative to minBorder, in emission order. Returns the number kept.
int distribute_octree(const dcs_candidate* cand, int n, int width, int height, int n_target,
                      std::vector<dcs_candidate>& out);

}  // namespace dcs
