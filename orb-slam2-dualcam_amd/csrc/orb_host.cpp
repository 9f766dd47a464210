// orb_host.cpp -- extractor constants computed once on the host and uploaded to HBM.
// Reference behaviour: src/ORBextractor.cc:410-470 (ctor tables), :1107-1132 (level sizes),
// :769-806 (cell grid); OpenCV resize coefficient tables per SURVEY.md Appendix A.2.
#include "orb_host.h"

#include <algorithm>
#include <cmath>

namespace dcs {

namespace {
inline int round_half_even(float v) { return (int)lrintf(v); }     // cvRound(float)
inline int round_half_even(double v) { return (int)lrint(v); }     // cvRound(double)
inline int floor_int(double v) { int i = (int)v; return i - (i > v); }
inline int ceil_int(double v) { int i = (int)v; return i + (i < v); }
inline int16_t coef11(float v)
{
    int i = round_half_even(v);
    return (int16_t)std::min(32767, std::max(-32768, i));
}
}  // namespace

void OrbTables::build(int nf, float sf, int nl, int ini, int mn)
{
    nfeatures = nf; nlevels = nl; ini_th = ini; min_th = mn;
    scale_factor = sf;                                     // double member initialised from a float
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; ++i) {
        scale[i] = (float)((double)scale[i - 1] * scale_factor);
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; ++i) { inv_scale[i] = 1.0f / scale[i]; inv_sigma2[i] = 1.0f / sigma2[i]; }
    const float factor = (float)(1.0f / scale_factor);
    float per_scale = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; ++l) {
        n_per_level[l] = round_half_even(per_scale);
        sum += n_per_level[l];
        per_scale *= factor;
    }
    n_per_level[nl - 1] = std::max(nf - sum, 0);
    // rows of the circular orientation patch
    const int vmax = floor_int(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = ceil_int(kHalfPatch * std::sqrt(2.f) / 2);
    const double r2 = (double)kHalfPatch * kHalfPatch;
    for (int v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
    for (int v = 0; v <= vmax; ++v) umax[v] = round_half_even(std::sqrt(r2 - (double)v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

void PyramidGeom::build(const OrbTables& t, int rows_, int cols_)
{
    rows = rows_; cols = cols_; nlevels = t.nlevels;
    size_t off = 0, slots = 0;
    int cells = 0;
    const float W = 30;
    for (int l = 0; l < nlevels; ++l) {
        LevelGeom& g = lv[l];
        g = LevelGeom();
        g.w = round_half_even((float)cols * t.inv_scale[l]);
        g.h = round_half_even((float)rows * t.inv_scale[l]);
        g.pitch = (g.w + 63) & ~63;
        g.offset = off;
        off += (size_t)g.pitch * g.h;
        off = (off + 255) & ~(size_t)255;
        g.scaled_patch = (int)(kPatchSize * t.scale[l]);
        const int max_bx = g.w - kEdgeThreshold + 3, max_by = g.h - kEdgeThreshold + 3;
        const float width = (float)(max_bx - kMinBorder), height = (float)(max_by - kMinBorder);
        g.cell_base = cells; g.slot_base = slots;
        if (width >= W && height >= W) {
            g.n_cols = (int)(width / W); g.n_rows = (int)(height / W);
            g.w_cell = (int)std::ceil(width / g.n_cols); g.h_cell = (int)std::ceil(height / g.n_rows);
            // strict 8-neighbour local maxima cannot touch: at most one per 2x2 block
            g.cell_cap = ((g.w_cell + 1) / 2) * ((g.h_cell + 1) / 2);
            cells += g.n_cols * g.n_rows;
            slots += (size_t)g.n_cols * g.n_rows * g.cell_cap;
        }
    }
    slab_bytes = off; n_cells = cells; n_slots = slots;
}

void ResizeTable::build(int sw, int sh, int dw, int dh)
{
    const float ONE = 2048.f;
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    xofs.resize(dw); xa.resize(2 * (size_t)dw); yofs.resize(dh); ya.resize(2 * (size_t)dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = floor_int(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }      // single-tap column (second tap weight 0)
        xofs[dx] = (int16_t)sx;
        xa[2 * dx] = coef11((1.f - fx) * ONE);
        xa[2 * dx + 1] = coef11(fx * ONE);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = floor_int(fy);
        fy -= sy;
        yofs[dy] = (int16_t)sy;                          // rows sy, sy+1 are clamped at use
        ya[2 * dy] = coef11((1.f - fy) * ONE);
        ya[2 * dy + 1] = coef11(fy * ONE);
    }
}

}  // namespace dcs
