// hardware probe: issue interval of v_mfma_f64_16x16x4_f64 on gfx950 -- dependent chain vs independent accumulators, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    const int iters = 1000;
    auto run = [&](auto kern, int nacc, int threads) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("nacc=%d threads=%d: %.1f clock64 ticks per MFMA per wave, %.1f ns per MFMA per wave (kernel %.1f us)\n", nacc, threads, (double)h / (iters * nacc), ms * 1e6 / (iters * nacc), ms * 1e3);
    };
    run(k<1>, 1, 64); run(k<4>, 4, 64); run(k<1>, 1, 256); run(k<4>, 4, 256); run(k<1>, 1, 512); run(k<4>, 4, 512); run(k<2>, 2, 512);
    return 0;
}
