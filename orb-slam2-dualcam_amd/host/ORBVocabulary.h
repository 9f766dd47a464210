// ORBVocabulary.h -- header-only C++ mirror of ORB_SLAM2::ORBVocabulary (reference include/ORBVocabulary.h:31-32 =
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) over the C ABI: the text loader the reference added to DBoW2
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1362-1446), transform (…:1151-1228, as called by Frame::ComputeBoW,
// src/Frame.cc:393-406) and score (ScoringObject.cpp:23-67, as called by KeyFrameDatabase.cc:250-372). The tree lives in HBM;
// BowVector / FeatureVector come back flat (ascending ids), which is also what dcs_search_by_bow consumes.
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBmatcher.h"
#include "dcs_abi.h"

namespace ORB_SLAM2 {

// flat DBoW2::BowVector (BowVector.h:58-59: std::map<WordId, WordValue>): ascending word ids + values
struct BowVectorFlat { std::vector<int32_t> words; std::vector<double> values; };

class ORBVocabulary {
public:
    ORBVocabulary() = default;
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;
    ~ORBVocabulary() { dcs_vocab_destroy(h_); }

    // loadFromTextFile: "k L scoring weighting" then one line per node: "parent leaf b0 .. b31 weight". Returns false like the
    // reference when the header is not a vocabulary's; throws when the GPU library rejects the tree.
    bool loadFromTextFile(const std::string& filename)
    {
        std::ifstream f(filename.c_str());
        if (!f) return false;
        std::string s;
        std::getline(f, s);
        std::stringstream ss(s);
        int k = -1, L = -1, n1 = -1, n2 = -1;
        ss >> k >> L >> n1 >> n2;
        if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return false;
        std::vector<int32_t> parent;
        std::vector<uint8_t> leaf, desc;
        std::vector<double> weight;
        while (std::getline(f, s)) {
            if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;
            std::stringstream sn(s);
            int pid = 0, is_leaf = 0;
            sn >> pid >> is_leaf;
            parent.push_back(pid); leaf.push_back(is_leaf > 0 ? 1 : 0);
            for (int i = 0; i < 32; ++i) { int b = 0; sn >> b; desc.push_back((uint8_t)b); }
            double w = 0;
            sn >> w;
            weight.push_back(w);
        }
        dcs_vocab_destroy(h_); h_ = nullptr;
        const int rc = dcs_vocab_create(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), &h_);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_vocab_create: ") + dcs_last_error());
        return true;
    }

    bool empty() const { return h_ == nullptr; }
    unsigned int size() const { int n = 0; if (h_) dcs_vocab_info(h_, nullptr, nullptr, nullptr, &n); return (unsigned)n; }

    // transform(features, v, fv, levelsup): descriptors = n x 32 bytes (one image / camera)
    void transform(const std::vector<uint8_t>& descriptors, BowVectorFlat& v, FeatureVectorCSR& fv, int levelsup) const
    {
        const int n = (int)(descriptors.size() / 32);
        v.words.assign(n > 0 ? n : 1, 0); v.values.assign(n > 0 ? n : 1, 0.0);
        fv.nodes.assign(n > 0 ? n : 1, 0); fv.off.assign(n + 1, 0); fv.idx.assign(n > 0 ? n : 1, 0);
        int nw = 0, nn = 0;
        if (!h_) { v.words.clear(); v.values.clear(); fv.nodes.clear(); fv.off.assign(1, 0); fv.idx.clear(); return; }      // empty(): nothing (:1156-1159)
        const int rc = dcs_bow_transform(h_, descriptors.data(), n, levelsup, nullptr, nullptr, v.words.data(), v.values.data(), &nw, fv.nodes.data(),
                                         fv.off.data(), fv.idx.data(), &nn);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_bow_transform: ") + dcs_last_error());
        v.words.resize(nw); v.values.resize(nw);
        fv.nodes.resize(nn); fv.off.resize(nn + 1); fv.idx.resize(fv.off[nn]);
    }

    // score(v1, v2), L1 (the scoring ORBvoc.txt declares)
    double score(const BowVectorFlat& v1, const BowVectorFlat& v2) const
    {
        const int32_t off[2] = {0, (int32_t)v2.words.size()};
        double s = 0;
        const int rc = dcs_bow_score_l1(v1.words.data(), v1.values.data(), (int)v1.words.size(), off, v2.words.data(), v2.values.data(), 1, &s);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_bow_score_l1: ") + dcs_last_error());
        return s;
    }

    const dcs_vocab* handle() const { return h_; }

private:
    dcs_vocab* h_ = nullptr;
};

}  // namespace ORB_SLAM2
