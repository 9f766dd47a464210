// Micro-benchmark of the single-workgroup LDL^T kernels (not part of the library): random SPD system, B copies,
// residual check, average time per launch, optional in-kernel phase timestamps (-DLDLT_PROF).
#include "../../orb-slam2-dualcam_amd/csrc/ba_solver.hip"
#include <random>
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 234, B = argc > 2 ? atoi(argv[2]) : 1, which = argc > 3 ? atoi(argv[3]) : 1;
    const int n_pad = (n + 15) / 16 * 16, ld = n_pad;
    std::mt19937_64 rng(5);
    std::normal_distribution<double> N(0, 1);
    std::vector<double> A((size_t)n * n), S((size_t)ld * ld, 0.0), b(n), x(n);
    for (auto& v : A) v = N(rng);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0; for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * A[(size_t)j * n + k];
        if (i == j) s += n;
        S[(size_t)i * ld + j] = s; S[(size_t)j * ld + i] = s;
    }
    for (auto& v : b) v = N(rng);
    std::vector<double> St((size_t)ld * ld, 0.0);              // what k_schur writes for k_ldlt_mfma: 16x16 tiles of the lower triangle (s_tile_off)
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if ((i >> 4) >= (j >> 4)) St[s_tile_off(i, j)] = S[(size_t)i * ld + j];
    double *dS, *db, *dx; BaProb* dp; BaCtl* dc;
    hipMalloc(&dS, sizeof(double) * ld * ld * B); hipMalloc(&db, sizeof(double) * n_pad * B); hipMalloc(&dx, sizeof(double) * n_pad * B);
    hipMalloc(&dp, sizeof(BaProb) * B); hipMalloc(&dc, sizeof(BaCtl) * B);
    std::vector<BaProb> hp(B); std::vector<BaCtl> hc(B);
    for (int i = 0; i < B; ++i) {
        hipMemcpy(dS + (size_t)i * ld * ld, which == 1 ? St.data() : S.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice);
        hipMemcpy(db + (size_t)i * n_pad, b.data(), sizeof(double) * n, hipMemcpyHostToDevice);
        memset(&hp[i], 0, sizeof(BaProb)); memset(&hc[i], 0, sizeof(BaCtl));
        hp[i].np = n / 6 ? n / 6 : 1; hp[i].n = n; hp[i].n_pad = n_pad; hp[i].ld = ld; hp[i].use_reg = which;
        hp[i].S = dS + (size_t)i * ld * ld; hp[i].bsch = db + (size_t)i * n_pad; hp[i].xp = dx + (size_t)i * n_pad;
        hc[i].state = ST_NEW_ITER; hc[i].ok = 1.0;
    }
    hipMemcpy(dp, hp.data(), sizeof(BaProb) * B, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), sizeof(BaCtl) * B, hipMemcpyHostToDevice);
    auto launch = [&] {
        if (which == 1 && n <= 240) hipLaunchKernelGGL(k_ldlt_mfma<kLdltSlotsSmall>, dim3(B), dim3(kLdltThreads), 0, 0, (const BaProb*)dp, dc);
        else hipLaunchKernelGGL(k_ldlt_mfma<kLdltSlotsBig>, dim3(B), dim3(kLdltThreads), 0, 0, (const BaProb*)dp, dc);
    };
    launch();
    hipDeviceSynchronize();
    double worst = 0;
    for (int i = 0; i < B; ++i) {
        hipMemcpy(x.data(), dx + (size_t)i * n_pad, sizeof(double) * n, hipMemcpyDeviceToHost);
        for (int r = 0; r < n; ++r) { double s = -b[r]; for (int c = 0; c < n; ++c) s += S[(size_t)r * ld + c] * x[c]; worst = std::max(worst, fabs(s)); }
    }
    hipMemcpy(hc.data(), dc, sizeof(BaCtl) * B, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("n=%d B=%d kernel=%s: %.2f us per launch, max |Sx-b| = %.3e, ok=%g\n", n, B, which == 1 ? "mfma" : "valu", ms * 1e3 / reps, worst, hc[0].ok);
#ifdef LDLT_PROF
    long long h[512]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ldlt_prof), sizeof(h));
    for (int i = 0; i < 24; ++i) { if (!h[i * 8]) continue; printf("J=%2d:", i); for (int k = 0; k < 8; ++k) printf(" %7lld", h[i * 8 + k] - h[0]); printf("\n"); }
#endif
    return worst < 1e-8 ? 0 : 1;
}
