// accuracy of v_rcp_f64 and of one / two Newton steps behind it (relative error against 1 / x in IEEE division), and of v_rsq_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    o[4 * i] = r;
    r = fma(fma(-d, r, 1.0), r, r); o[4 * i + 1] = r;
    r = fma(fma(-d, r, 1.0), r, r); o[4 * i + 2] = r;
    o[4 * i + 3] = __builtin_amdgcn_rsq(fabs(d));
}
int main()
{
    const int n = 1 << 20;
    std::mt19937_64 g(7); std::uniform_real_distribution<double> u(-40, 40);
    std::vector<double> x(n), o(4 * n);
    for (auto& v : x) { v = std::exp2(u(g)) * (g() & 1 ? 1 : -1) * (1.0 + (g() >> 11) * 0x1p-53); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const double t = 1.0 / x[i];
        e0 = std::max(e0, std::fabs((o[4 * i] - t) / t)); e1 = std::max(e1, std::fabs((o[4 * i + 1] - t) / t)); e2 = std::max(e2, std::fabs((o[4 * i + 2] - t) / t));
        const double s = 1.0 / std::sqrt(std::fabs(x[i])); e3 = std::max(e3, std::fabs((o[4 * i + 3] - s) / s));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); one Newton step %.3e; two %.3e; v_rsq_f64 %.3e (2^%.1f)\n", e0, std::log2(e0), e1, e2, e3, std::log2(e3));
    return 0;
}
