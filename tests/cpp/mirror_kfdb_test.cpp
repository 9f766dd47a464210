// Drives host/KeyFrameDatabase.h like LoopClosing::DetectLoop / Tracking::Relocalization do (src/LoopClosing.cc:138-170,
// src/Tracking.cc:1950-1960): a database filled key frame by key frame, some erased, then queries. Input = one binary file
// written by tests/test_gpu_cpp_mirror.py; prints the candidate lists.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "KeyFrameDatabase.h"

template <typename T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short file\n"); exit(2); } return v; }

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    const std::vector<int32_t> hdr = rd<int32_t>(f, 3);              // n_db, n_queries, n_dead
    const int n_db = hdr[0], n_q = hdr[1], n_dead = hdr[2];
    ORB_SLAM2::KeyFrameDatabase db;
    std::vector<std::vector<int>> covis((size_t)n_db);
    for (int k = 0; k < n_db; ++k) {
        const int nw = rd<int32_t>(f, 1)[0];
        const std::vector<int32_t> w = rd<int32_t>(f, nw);
        const std::vector<double> v = rd<double>(f, nw);
        if (db.add(w, v) != k) return 3;
        const int nc = rd<int32_t>(f, 1)[0];
        const std::vector<int32_t> c = rd<int32_t>(f, nc);
        covis[k].assign(c.begin(), c.end());
    }
    for (int d : rd<int32_t>(f, n_dead)) db.erase(d);
    for (int loop = 0; loop < 2; ++loop) {
        ORB_SLAM2::KeyFrameDatabase::State st;
        long pos = ftell(f);
        for (int q = 0; q < n_q; ++q) {
            const std::vector<int32_t> h = rd<int32_t>(f, 2);        // query id, n words
            const std::vector<int32_t> w = rd<int32_t>(f, h[1]);
            const std::vector<double> v = rd<double>(f, h[1]);
            const std::vector<uint8_t> conn = rd<uint8_t>(f, n_db);
            const std::vector<int> got = loop ? db.DetectLoopCandidates(h[0], w, v, conn, covis, 0.05f, st) : db.DetectRelocalizationCandidates(h[0], w, v, covis, st);
            printf("%s %d:", loop ? "loop" : "reloc", q);
            for (int k : got) printf(" %d", k);
            printf("\n");
        }
        if (loop == 0) fseek(f, pos, SEEK_SET);
    }
    fclose(f);
    printf("size %d\n", db.size());
    return 0;
}
