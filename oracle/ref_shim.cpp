// ref_shim.cpp -- C entry points around the pieces of the REAL reference that compile here without OpenCV / Eigen
// (test infrastructure, built into oracle/_ref/libref.so by `make ref`, only where /root/reference exists):
//   * Thirdparty/DBoW2/DBoW2/BowVector.cpp, FeatureVector.cpp                       compiled from where they lie, unmodified
//   * Thirdparty/DBoW2/DBoW2/ScoringObject.cpp                                      unmodified except that its one
//       `#include "TemplatedVocabulary.h"` (which pulls OpenCV in) is dropped by sed at build time; the declarations it needs
//       come from the reference's own ScoringObject.h (-include)
//   * the loops of ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2015-2031) and FORB::distance
//       (Thirdparty/DBoW2/DBoW2/FORB.cpp:82-102): their statements from `int dist=0;` to `return dist;` are cut out by awk at build
//       time into _ref/*.inc (the two lines before them only fetch `const int*` row pointers from cv::Mat)
//   * ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1969-2010): the statements between the braces of its definition, cut out
//       by awk the same way; the wrapper below repeats the reference's parameter list (histo, L, ind1, ind2, ind3)
//   * RobustKernelHuber::setDelta / ::robustify (Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91): the statements between the
//       braces of the two definitions, cut out by awk; the struct below supplies the two members they use (dsqr, _delta) and the
//       parameter list (double e, rho indexable 0..2 like Eigen::Vector3d)
// Nothing of the reference is copied into the repository: the generated files live in oracle/_ref/ (git-ignored).
// The oracle's restatements are checked against these functions in tests/test_oracle_ref.py, and golden vectors produced
// by them are committed (tests/golden/ref_dbow2.npz, make_golden_ref.py) so that the pin travels to machines without the reference.
#include <cstdint>
#include <cstring>
#include <vector>

#include "BowVector.h"
#include "FeatureVector.h"
#include "ScoringObject.h"

static int ref_orbmatcher_distance(const int* pa, const int* pb)
{
#include "orbmatcher_distance_body.inc"
}
static int ref_forb_distance_impl(const int* pa, const int* pb)
{
#include "forb_distance_body.inc"
}

static void ref_three_maxima_impl(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
{
    using namespace std;
#include "three_maxima_body.inc"
}

struct RefHuber {
    double dsqr = 0, _delta = 0;
    void setDelta(double delta)
    {
#include "huber_setdelta_body.inc"
    }
    void robustify(double e, double* rho) const
    {
#include "huber_robustify_body.inc"
    }
};

extern "C" {

// rho[3] = {rho(e), rho'(e), rho''(e)} of g2o's Huber kernel with setDelta(delta), e = chi2 of the edge
void ref_huber(double delta, double e, double* rho)
{
    RefHuber k;
    k.setDelta(delta);
    k.robustify(e, rho);
}

// histogram given as bin sizes; ind[3] = {ind1, ind2, ind3} as the callers initialise them (-1) and the function leaves them
void ref_three_maxima(const int32_t* sizes, int L, int32_t* ind)
{
    std::vector<std::vector<int>> h((size_t)L);
    for (int i = 0; i < L; ++i) h[i].assign((size_t)sizes[i], 0);
    int i1 = -1, i2 = -1, i3 = -1;
    ref_three_maxima_impl(h.data(), L, i1, i2, i3);
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
}

int ref_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int32_t ra[8], rb[8];
    memcpy(ra, a, 32); memcpy(rb, b, 32);
    return ref_orbmatcher_distance(ra, rb);
}
int ref_forb_distance(const uint8_t* a, const uint8_t* b)
{
    int32_t ra[8], rb[8];
    memcpy(ra, a, 32); memcpy(rb, b, 32);
    return ref_forb_distance_impl(ra, rb);
}

// TemplatedVocabulary::transform's accumulation (TemplatedVocabulary.h:1186-1237): per feature addWeight (TF_IDF, TF) or
// addIfNotExist (IDF, BINARY) in feature order, then BowVector::normalize. norm: 0 none, 1 L1, 2 L2. Returns the size.
int ref_bow_vector(int n, const uint32_t* word, const double* weight, int if_not_exist, int norm, uint32_t* out_word, double* out_val)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;                       // transform(): `if(w > 0) v.addWeight(id, w);`
        if (if_not_exist) v.addIfNotExist(word[i], weight[i]); else v.addWeight(word[i], weight[i]);
    }
    if (norm == 1) v.normalize(DBoW2::L1); else if (norm == 2) v.normalize(DBoW2::L2);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { out_word[k] = it->first; out_val[k] = it->second; }
    return k;
}

// kind: 0 L1, 1 L2, 2 ChiSquare, 3 KL, 4 Bhattacharyya, 5 DotProduct (the ScoringType order of TemplatedVocabulary.h)
double ref_score(int kind, int n1, const uint32_t* w1, const double* v1, int n2, const uint32_t* w2, const double* v2)
{
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; ++i) a.insert(a.end(), std::make_pair(w1[i], v1[i]));
    for (int i = 0; i < n2; ++i) b.insert(b.end(), std::make_pair(w2[i], v2[i]));
    switch (kind) {
    case 0: return DBoW2::L1Scoring().score(a, b);
    case 1: return DBoW2::L2Scoring().score(a, b);
    case 2: return DBoW2::ChiSquareScoring().score(a, b);
    case 3: return DBoW2::KLScoring().score(a, b);
    case 4: return DBoW2::BhattacharyyaScoring().score(a, b);
    default: return DBoW2::DotProductScoring().score(a, b);
    }
}

// FeatureVector::addFeature for features 0 .. n-1 in order -> CSR (ascending node ids, offsets, feature indices). Returns #nodes.
int ref_feature_vector(int n, const uint32_t* node, uint32_t* out_node, int32_t* out_off, uint32_t* out_idx)
{
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; ++i) fv.addFeature(node[i], (unsigned)i);
    int k = 0, at = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        out_node[k] = it->first; out_off[k] = at;
        for (size_t j = 0; j < it->second.size(); ++j) out_idx[at++] = it->second[j];
    }
    out_off[k] = at;
    return k;
}

}  // extern "C"
