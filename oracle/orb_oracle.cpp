/*
 * orb_oracle.cpp -- CPU restatement of ORB extraction (TEST INFRASTRUCTURE ONLY; see oracle.h).
 *
 * Follows /root/reference/src/ORBextractor.cc (cited per function) and the OpenCV 3.3/3.4.0
 * non-IPP semantics of FAST / resize / GaussianBlur / fastAtan2 / cvRound as written down in
 * SURVEY.md Appendix A (OpenCV is not vendored by the reference: PARITY UNPINNED).
 * Canonical choices where the reference is address-dependent are marked Q3.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>

namespace {

const int PATCH_SIZE = 31;        /* ORBextractor.cc:72 */
const int HALF_PATCH_SIZE = 15;   /* ORBextractor.cc:73 */
const int EDGE_THRESHOLD = 19;    /* ORBextractor.cc:74 */

const int8_t kPattern[1024] = {
#include "brief_pattern.inc"
};

/* cvRound: round-half-to-even (SSE cvtss2si / lrint under the default rounding mode) */
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline short sat_s16_rne(float v) { int i = cv_round(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); }

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;   /* stride == w (the reference's clone()/ROI without border) */
    void alloc(int w_, int h_) { w = w_; h = h_; px.assign((size_t)w * h, 0); }
};

/* ---------------------------------------------------------------- resize */
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    const int ONE = 2048;                        /* INTER_RESIZE_COEF_SCALE, 11 bits */
    const double scale_x = 1.0 / ((double)dw / sw);
    const double scale_y = 1.0 / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_s16_rne((1.f - fx) * ONE);
        alpha[2 * dx + 1] = sat_s16_rne(fx * ONE);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = sat_s16_rne((1.f - fy) * ONE);
        beta[2 * dy + 1] = sat_s16_rne(fy * ONE);
    }
    /* horizontal pass for every source row (int, 11 fractional bits), then the vertical pass */
    std::vector<int> hbuf((size_t)sh * dw);
    for (int sy = 0; sy < sh; ++sy) {
        const uint8_t* S = src + (size_t)sy * sstride;
        int* out = &hbuf[(size_t)sy * dw];
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            if (sx >= sw - 1) out[dx] = S[sx] * ONE;    /* dx >= xmax: single tap */
            else out[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
        }
    }
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        const int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const int* row0 = &hbuf[(size_t)sy0 * dw];
        const int* row1 = &hbuf[(size_t)sy1 * dw];
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

/* ---------------------------------------------------------------- Gaussian 7x7 */
inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

void gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    /* getGaussianKernel(7, 2) as float, then *256 rounded -> {18,34,49,55,49,34,18} (sum 257) */
    static const int K[7] = {18, 34, 49, 55, 49, 34, 18};
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* S = src + (size_t)y * sstride;
        int* T = &tmp[(size_t)y * w];
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            if (x >= 3 && x < w - 3) {
                for (int k = 0; k < 7; ++k) acc += K[k] * S[x + k - 3];
            } else {
                for (int k = 0; k < 7; ++k) acc += K[k] * S[reflect101(x + k - 3, w)];
            }
            T[x] = acc;
        }
    }
    for (int y = 0; y < h; ++y) {
        const int* R[7];
        for (int k = 0; k < 7; ++k) R[k] = &tmp[(size_t)reflect101(y + k - 3, h) * w];
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += K[k] * R[k][x];
            D[x] = sat_u8((acc + 32768) >> 16);
        }
    }
}

/* ---------------------------------------------------------------- FAST-9/16 */
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* max over the 16 contiguous 9-arcs of min(sign-consistent |centre - ring|), minus 1.
   "corner at T" <=> fast_score >= T ... equivalently arc-min > T (OpenCV cornerScore<16>). */
inline int fast_score_px(const uint8_t* p, const int* off)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 16; ++k) d[k] = v - p[off[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int best = -256;
    for (int s = 0; s < 16; ++s) {
        int mn = d[s], mx = d[s];
        for (int k = 1; k < 9; ++k) { mn = std::min(mn, d[s + k]); mx = std::max(mx, d[s + k]); }
        best = std::max(best, std::max(mn, -mx));
    }
    return best - 1;
}

/* is (p) a FAST-9 corner at threshold T? quick reject on opposite ring pairs first. */
inline bool fast_is_corner(const uint8_t* p, const int* off, int T)
{
    const int v = p[0], hi = v + T, lo = v - T;
    auto cls = [&](int k) -> int { const int r = p[off[k]]; return r > hi ? 2 : (r < lo ? 1 : 0); };
    /* a 9-arc of 16 contains at least one pixel of every opposite pair (k, k+8) */
    int m = cls(0) | cls(8);
    if (!m) return false;
    m &= cls(2) | cls(10); if (!m) return false;
    m &= cls(4) | cls(12); if (!m) return false;
    m &= cls(6) | cls(14); if (!m) return false;
    m &= cls(1) | cls(9);  if (!m) return false;
    m &= cls(3) | cls(11); if (!m) return false;
    m &= cls(5) | cls(13); if (!m) return false;
    m &= cls(7) | cls(15); if (!m) return false;
    for (int pass = 1; pass <= 2; ++pass) {
        if (!(m & pass)) continue;
        int run = 0;
        for (int k = 0; k < 25; ++k) {
            if (cls(k & 15) == pass) { if (++run >= 9) return true; } else run = 0;
        }
    }
    return false;
}

/* cv::FAST(roi, kps, T, true): detection on rows/cols [3, dim-3), score map, strict 8-neighbour
   NMS where non-corners / outside pixels count 0; emission row-major. */
int fast_roi(const uint8_t* roi, int w, int h, int stride, int T, orc_candidate* out, int cap,
             std::vector<int>& score /* scratch w*h */)
{
    if (w < 7 || h < 7) return 0;
    int off[16];
    for (int k = 0; k < 16; ++k) off[k] = kRingDy[k] * stride + kRingDx[k];
    score.assign((size_t)w * h, 0);
    bool any = false;
    for (int y = 3; y < h - 3; ++y) {
        const uint8_t* row = roi + (size_t)y * stride;
        for (int x = 3; x < w - 3; ++x) {
            if (fast_is_corner(row + x, off, T)) { score[(size_t)y * w + x] = fast_score_px(row + x, off); any = true; }
        }
    }
    if (!any) return 0;
    int n = 0;
    for (int y = 3; y < h - 3; ++y) {
        for (int x = 3; x < w - 3; ++x) {
            const int s = score[(size_t)y * w + x];
            if (s <= 0) continue;   /* non-corner (0); a score-0 corner can never beat its neighbours */
            const int* c = &score[(size_t)y * w + x];
            if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                s > c[w - 1] && s > c[w] && s > c[w + 1]) {
                if (n < cap) { out[n].x = (int16_t)x; out[n].y = (int16_t)y; out[n].score = s; }
                ++n;
            }
        }
    }
    return n;
}

/* ---------------------------------------------------------------- fastAtan2 */
float fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct Tables {
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor;                    /* ORBextractor.h:100: double member from a float arg */
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> n_per_level;
    int umax[HALF_PATCH_SIZE + 1];
};

/* ORBextractor::ORBextractor (ORBextractor.cc:410-470) */
void build_tables(Tables& t, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    t.nfeatures = nfeatures; t.nlevels = nlevels; t.ini_th = ini_th; t.min_th = min_th;
    t.scale_factor = scale_factor;
    t.scale.assign(nlevels, 1.f); t.sigma2.assign(nlevels, 1.f);
    for (int i = 1; i < nlevels; ++i) {
        t.scale[i] = (float)(t.scale[i - 1] * t.scale_factor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    t.inv_scale.resize(nlevels); t.inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) { t.inv_scale[i] = 1.0f / t.scale[i]; t.inv_sigma2[i] = 1.0f / t.sigma2[i]; }
    t.n_per_level.assign(nlevels, 0);
    const float factor = (float)(1.0f / t.scale_factor);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        t.n_per_level[l] = cv_round(desired);
        sum += t.n_per_level[l];
        desired *= factor;
    }
    t.n_per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    /* umax (ORBextractor.cc:454-469) */
    int v, v0;
    const int vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    const int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= HALF_PATCH_SIZE; ++v) t.umax[v] = 0;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
}

/* IC_Angle (ORBextractor.cc:77-104) */
float ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

/* glibc >= 2.28 sinf / cosf (sysdeps/ieee754/flt-32/s_sincosf.h, the ARM optimized-routines code) for |x| < 120: argument
   reduction by pi/2 and two minimax polynomials, all in double. Exported so that the tests can hold it against the host's
   libm (exhaustively equal on [0, 6.5) -- every angle an ORB keypoint can have -- with and without FMA contraction); the
   GPU kernel evaluates the same expressions (csrc/orb_kernels.hip, glibc_sincosf). */
namespace {
struct SinCosTab { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
const SinCosTab kSinCos[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
inline uint32_t abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
inline float sincos_poly(double x, double x2, const SinCosTab& p, int n)
{
    if ((n & 1) == 0) { const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1; return (float)(s + x7 * s1); }
    const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}
}  // namespace

extern "C" void orc_sincosf_restated(const float* x, int n, float* cos_out, float* sin_out)
{
    for (int i = 0; i < n; ++i) {
        const float y = x[i];
        double xd = y;
        if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {                           /* |y| < pi/4 */
            const double x2 = xd * xd;
            const bool tiny = abstop12(y) < abstop12(0x1p-12f);
            cos_out[i] = tiny ? 1.0f : sincos_poly(xd, x2, kSinCos[0], 1);
            sin_out[i] = tiny ? y : sincos_poly(xd, x2, kSinCos[0], 0);
            continue;
        }
        const double r = xd * kSinCos[0].hpi_inv;                               /* valid for |y| < 120 */
        const int q = ((int32_t)r + 0x800000) >> 24;
        xd = xd - q * kSinCos[0].hpi;
        const double sgn = kSinCos[0].sign[q & 3];
        const SinCosTab& p = kSinCos[(q & 2) ? 1 : 0];
        cos_out[i] = sincos_poly(xd * sgn, xd * xd, p, q ^ 1);
        sin_out[i] = sincos_poly(xd * sgn, xd * xd, p, q);
    }
}

/* the host libm's float overloads: what the reference's cos(angle) / sin(angle) call (ORBextractor.cc:112-113) */
extern "C" void orc_libm_sincosf(const float* x, int n, float* cos_out, float* sin_out)
{
    for (int i = 0; i < n; ++i) { cos_out[i] = cosf(x[i]); sin_out[i] = sinf(x[i]); }
}

/* computeOrbDescriptor (ORBextractor.cc:108-147). `cos(angle)` / `sin(angle)` on a float with <cmath> and `using namespace
   std` in scope (:66-67, :112-113) resolve to the float overloads, i.e. libm's cosf / sinf -- NOT a double cosine narrowed
   to float as SURVEY A.5 has it (the two differ by one ulp for 0.3 % / 0.6 % of the angles). The oracle calls the same
   functions the reference calls; sincosf_restated() below is glibc's (>= 2.28) algorithm spelled out, which the GPU runs. */
void brief256(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    const float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img + (size_t)y * stride + x;
    const int8_t* pat = kPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            const float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1];
            const float x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
            const float r0 = x0 * b, r0b = y0 * a, c0 = x0 * a, c0b = y0 * b;
            const float r1 = x1 * b, r1b = y1 * a, c1 = x1 * a, c1b = y1 * b;
            const int t0 = center[cv_round(r0 + r0b) * stride + cv_round(c0 - c0b)];
            const int t1 = center[cv_round(r1 + r1b) * stride + cv_round(c1 - c1b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- quadtree (ORBextractor.cc:481-763) */
struct Node {
    std::vector<int> keys;             /* indices into the candidate array, original order */
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool no_more = false;
    long seq = 0;                      /* Q3: creation sequence number replaces the heap address */
};

void divide_node(const Node& p, const orc_candidate* c, Node& n1, Node& n2, Node& n3, Node& n4)
{
    const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (int idx : p.keys) {
        const float px = (float)c[idx].x, py = (float)c[idx].y;
        if (px < n1.URx) { if (py < n1.BRy) n1.keys.push_back(idx); else n3.keys.push_back(idx); }
        else if (py < n1.BRy) n2.keys.push_back(idx);
        else n4.keys.push_back(idx);
    }
    if (n1.keys.size() == 1) n1.no_more = true;
    if (n2.keys.size() == 1) n2.no_more = true;
    if (n3.keys.size() == 1) n3.no_more = true;
    if (n4.keys.size() == 1) n4.no_more = true;
}

struct SizeSeqNode { int size; long seq; Node* node; };
inline bool operator<(const SizeSeqNode& a, const SizeSeqNode& b)
{ return a.size != b.size ? a.size < b.size : a.seq < b.seq; }

int distribute_octree(const orc_candidate* c, int n, int minX, int maxX, int minY, int maxY, int N,
                      orc_candidate* out, int cap)
{
    if (n <= 0) return 0;
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) return 0;            /* reference divides by zero here (portrait images) */
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; ++i) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (int i = 0; i < n; ++i) {
        int k = (int)((float)c[i].x / hX);
        if (k >= nIni) k = nIni - 1;   /* cannot happen for in-range x; guards the UB */
        ini[k]->keys.push_back(i);
    }
    for (auto lit = nodes.begin(); lit != nodes.end();) {
        if (lit->keys.size() == 1) { lit->no_more = true; ++lit; }
        else if (lit->keys.empty()) lit = nodes.erase(lit);
        else ++lit;
    }
    bool finish = false;
    std::vector<SizeSeqNode> todo;
    auto push_child = [&](Node& ch, int* n_expand) {
        if (ch.keys.empty()) return;
        ch.seq = seq++;
        nodes.push_front(ch);
        if (ch.keys.size() > 1) {
            if (n_expand) ++*n_expand;
            todo.push_back({(int)ch.keys.size(), nodes.front().seq, &nodes.front()});
            nodes.front().lit = nodes.begin();
        }
    };
    while (!finish) {
        const int prev = (int)nodes.size();
        int n_expand = 0;
        todo.clear();
        for (auto lit = nodes.begin(); lit != nodes.end();) {
            if (lit->no_more) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, c, n1, n2, n3, n4);
            push_child(n1, &n_expand); push_child(n2, &n_expand);
            push_child(n3, &n_expand); push_child(n4, &n_expand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev) {
            finish = true;
        } else if ((int)nodes.size() + n_expand * 3 > N) {
            while (!finish) {
                const int prev2 = (int)nodes.size();
                std::vector<SizeSeqNode> prev_todo = todo;
                todo.clear();
                std::sort(prev_todo.begin(), prev_todo.end());
                for (int j = (int)prev_todo.size() - 1; j >= 0; --j) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev_todo[j].node, c, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    nodes.erase(prev_todo[j].node->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev2) finish = true;
            }
        }
    }
    int m = 0;
    for (auto& nd : nodes) {
        int best = nd.keys[0];
        float max_resp = (float)c[best].score;
        for (size_t k = 1; k < nd.keys.size(); ++k) {
            if ((float)c[nd.keys[k]].score > max_resp) { best = nd.keys[k]; max_resp = (float)c[best].score; }
        }
        if (m < cap) out[m] = c[best];
        ++m;
    }
    return m;
}

}  // namespace

struct orc_orb {
    Tables t;
    std::vector<Image> pyr, blur;
    std::vector<std::vector<orc_candidate>> cand;
    std::vector<std::vector<orc_keypoint>> kps;   /* level coordinates, before the final scaling */
    std::vector<int> score_scratch;
};

extern "C" {

orc_orb* orc_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nfeatures < 0) return nullptr;
    orc_orb* o = new orc_orb;
    build_tables(o->t, nfeatures, scale_factor, nlevels, ini_th, min_th);
    o->pyr.resize(nlevels); o->blur.resize(nlevels); o->cand.resize(nlevels); o->kps.resize(nlevels);
    return o;
}

void orc_orb_destroy(orc_orb* o) { delete o; }

void orc_orb_tables(const orc_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* n_per_level, int32_t* umax16)
{
    for (int i = 0; i < o->t.nlevels; ++i) {
        if (scale) scale[i] = o->t.scale[i];
        if (inv_scale) inv_scale[i] = o->t.inv_scale[i];
        if (sigma2) sigma2[i] = o->t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = o->t.inv_sigma2[i];
        if (n_per_level) n_per_level[i] = o->t.n_per_level[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = o->t.umax[i];
}

/* ORBextractor::operator() (ORBextractor.cc:1043-1105) */
int orc_orb_extract(orc_orb* o, const uint8_t* img, int rows, int cols, int stride,
                    orc_keypoint* kp, uint8_t* desc, int cap, int* n_out)
{
    const Tables& t = o->t;
    *n_out = 0;
    if (!img || rows <= 0 || cols <= 0) return 0;           /* _image.empty() -> return (:1046) */

    /* ComputePyramid (:1107-1132): cascaded resize; the 19-px border is dead data for this path */
    for (int l = 0; l < t.nlevels; ++l) {
        const float s = t.inv_scale[l];
        const int w = cv_round((float)cols * s), h = cv_round((float)rows * s);
        o->pyr[l].alloc(w, h);
        if (l == 0) {
            for (int y = 0; y < rows; ++y) memcpy(&o->pyr[0].px[(size_t)y * cols], img + (size_t)y * stride, cols);
        } else {
            resize_linear_u8(o->pyr[l - 1].px.data(), o->pyr[l - 1].w, o->pyr[l - 1].h, o->pyr[l - 1].w,
                             o->pyr[l].px.data(), w, h, w);
        }
    }

    /* ComputeKeyPointsOctTree (:765-853) */
    const float W = 30;
    std::vector<orc_candidate> cell(4096), sel;
    for (int l = 0; l < t.nlevels; ++l) {
        const Image& im = o->pyr[l];
        std::vector<orc_candidate>& cand = o->cand[l];
        cand.clear(); o->kps[l].clear();
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
        const int maxBX = im.w - EDGE_THRESHOLD + 3, maxBY = im.h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
            const int iniY = minBY + i * hCell;
            int maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = maxBY;
            for (int j = 0; j < nCols; ++j) {
                const int iniX = minBX + j * wCell;
                int maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = maxBX;
                const uint8_t* roi = &im.px[(size_t)iniY * im.w + iniX];
                int n = fast_roi(roi, maxX - iniX, maxY - iniY, im.w, t.ini_th, cell.data(), (int)cell.size(), o->score_scratch);
                if (n == 0)
                    n = fast_roi(roi, maxX - iniX, maxY - iniY, im.w, t.min_th, cell.data(), (int)cell.size(), o->score_scratch);
                for (int k = 0; k < n; ++k) {
                    orc_candidate cc = cell[k];
                    cc.x = (int16_t)(cc.x + j * wCell);
                    cc.y = (int16_t)(cc.y + i * hCell);
                    cand.push_back(cc);
                }
            }
        }
        sel.resize(std::max<size_t>(cand.size(), 1));
        const int m = distribute_octree(cand.data(), (int)cand.size(), minBX, maxBX, minBY, maxBY,
                                        t.n_per_level[l], sel.data(), (int)sel.size());
        const int scaledPatchSize = (int)(PATCH_SIZE * t.scale[l]);
        for (int k = 0; k < m; ++k) {
            orc_keypoint q;
            q.x = (float)sel[k].x + minBX; q.y = (float)sel[k].y + minBY;
            q.size = (float)scaledPatchSize; q.angle = -1.f; q.response = (float)sel[k].score;
            q.octave = l; q.class_id = -1;
            o->kps[l].push_back(q);
        }
    }
    /* computeOrientation (:472-479, 851-852) on the unblurred level */
    for (int l = 0; l < t.nlevels; ++l)
        for (auto& q : o->kps[l])
            q.angle = ic_angle(o->pyr[l].px.data(), o->pyr[l].w, cv_round(q.x), cv_round(q.y), t.umax);

    int total = 0;
    for (int l = 0; l < t.nlevels; ++l) total += (int)o->kps[l].size();
    if (total > cap) { *n_out = total; return -1; }

    int offset = 0;
    for (int l = 0; l < t.nlevels; ++l) {
        o->blur[l].w = o->blur[l].h = 0; o->blur[l].px.clear();
        if (o->kps[l].empty()) continue;
        o->blur[l].alloc(o->pyr[l].w, o->pyr[l].h);
        gauss7_u8(o->pyr[l].px.data(), o->pyr[l].w, o->pyr[l].h, o->pyr[l].w, o->blur[l].px.data(), o->pyr[l].w);
        for (const auto& q : o->kps[l]) {
            brief256(o->blur[l].px.data(), o->blur[l].w, cv_round(q.x), cv_round(q.y), q.angle, desc + (size_t)offset * 32);
            orc_keypoint r = q;
            if (l != 0) { r.x = q.x * t.scale[l]; r.y = q.y * t.scale[l]; }
            kp[offset++] = r;
        }
    }
    *n_out = total;
    return 0;
}

int orc_orb_level_dims(const orc_orb* o, int level, int* w, int* h)
{
    if (level < 0 || level >= o->t.nlevels) return -1;
    *w = o->pyr[level].w; *h = o->pyr[level].h;
    return 0;
}

int orc_orb_level_copy(const orc_orb* o, int level, int blurred, uint8_t* dst)
{
    if (level < 0 || level >= o->t.nlevels) return -1;
    const Image& im = blurred ? o->blur[level] : o->pyr[level];
    if (im.px.empty()) return 0;
    memcpy(dst, im.px.data(), im.px.size());
    return (int)im.px.size();
}

int orc_orb_level_candidates(const orc_orb* o, int level, orc_candidate* dst, int cap)
{
    if (level < 0 || level >= o->t.nlevels) return -1;
    const int n = (int)o->cand[level].size();
    if (dst) memcpy(dst, o->cand[level].data(), sizeof(orc_candidate) * std::min(n, cap));
    return n;
}

int orc_orb_level_keypoints(const orc_orb* o, int level, orc_keypoint* dst, int cap)
{
    if (level < 0 || level >= o->t.nlevels) return -1;
    const int n = (int)o->kps[level].size();
    if (dst) memcpy(dst, o->kps[level].data(), sizeof(orc_keypoint) * std::min(n, cap));
    return n;
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{ resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride); }

void orc_gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{ gauss7_u8(src, w, h, sstride, dst, dstride); }

int orc_fast_roi(const uint8_t* roi, int w, int h, int stride, int threshold, orc_candidate* out, int cap)
{ std::vector<int> scratch; return fast_roi(roi, w, h, stride, threshold, out, cap, scratch); }

int orc_fast_score(const uint8_t* p, int stride)
{
    int off[16];
    for (int k = 0; k < 16; ++k) off[k] = kRingDy[k] * stride + kRingDx[k];
    return fast_score_px(p, off);
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }

float orc_ic_angle(const uint8_t* img, int stride, int x, int y)
{
    Tables t; build_tables(t, 1000, 1.2f, 8, 20, 7);
    return ic_angle(img, stride, x, y, t.umax);
}

void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32)
{ brief256(blurred, stride, x, y, angle_deg, desc32); }

int orc_distribute_octree(const orc_candidate* cand, int n, int minX, int maxX, int minY, int maxY,
                          int N, orc_candidate* out, int cap)
{ return distribute_octree(cand, n, minX, maxX, minY, maxY, N, out, cap); }

}  // extern "C"
