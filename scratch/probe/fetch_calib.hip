// hardware probe: what does the FETCH_SIZE counter report for a kernel that reads a KNOWN number of bytes from HBM, by width of the
// per-lane load? Every kernel streams the same buffer (far larger than L2 + Infinity Cache... the first pass is cold) exactly once with
// 1-, 4-, 8-, 12- (dwordx3) and 16-byte loads per lane, consecutive lanes on consecutive addresses, and writes one word per workgroup.
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (tools/fetch_calib.sh); reported KB / true KB per width = the correction bench.py
// applies to `roofline.traffic` (MI355X_MICROARCH.md: on gfx950 the counter tallies 128-byte requests at 64 B).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename T> __device__ unsigned fold(T v);
template <> __device__ unsigned fold(uint8_t v) { return v; }
template <> __device__ unsigned fold(uint32_t v) { return v; }
template <> __device__ unsigned fold(uint2 v) { return v.x ^ v.y; }
template <> __device__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
struct u3 { uint32_t a, b, c; };
template <> __device__ unsigned fold(u3 v) { return v.a ^ v.b ^ v.c; }
template <typename T> __global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, size_t n, unsigned* __restrict__ out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= fold(p[i]);
    if (acc == 0x12345677u) out[blockIdx.x] = acc;               // never true for this data: keeps the loads alive without a store per thread
}
// rows of 38 x 8 bytes at a 704-byte pitch, 6 lanes x 10 rows per wave: the access shape of k_fast_cells' ROI loads
__global__ __launch_bounds__(64) void k_read_roi(const uint8_t* __restrict__ p, int pitch, int rows_total, unsigned* __restrict__ out)
{
    const int lane = threadIdx.x, r0 = lane / 6, c = lane - 6 * r0;
    const size_t roi = blockIdx.x;                                 // ROI b: 38 rows starting at row 30 * (b % tiles_y), column 30 * ...: here simply consecutive row bands
    unsigned acc = 0;
    if (lane < 60)
        for (int i = 0; i < 4; ++i) {
            const size_t r = (roi * 40 + r0 + 10 * i) % rows_total;
            const uint2 v = *reinterpret_cast<const uint2*>(p + r * pitch + 8 * c + 48 * (roi % 14));
            acc ^= v.x ^ v.y;
        }
    if (acc == 0x12345677u) out[blockIdx.x & 1023] = acc;
}
int main()
{
    const size_t bytes = (size_t)1 << 30;                         // 1 GiB: four times the Infinity Cache
    uint8_t* d; unsigned* o;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4 * 65536));
    CK(hipMemset(d, 0x5a, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read<uint8_t>, dim3(grid), dim3(256), 0, 0, (const uint8_t*)d, bytes / 4, o);          // 256 MiB through byte loads
        hipLaunchKernelGGL(k_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)d, bytes / 4, o);
        hipLaunchKernelGGL(k_read<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)d, bytes / 8, o);
        hipLaunchKernelGGL(k_read<u3>, dim3(grid), dim3(256), 0, 0, (const u3*)d, bytes / 12, o);
        hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)d, bytes / 16, o);
        hipLaunchKernelGGL(k_read_roi, dim3(1 << 20), dim3(64), 0, 0, (const uint8_t*)d, 704, (int)(bytes / 704), o);
        CK(hipDeviceSynchronize());
    }
    printf("true bytes per launch: u8 %zu, u32 %zu, u64 %zu, u96 %zu, u128 %zu, roi %zu (requested; 8-byte columns of 48-byte row pieces)\n",
           bytes / 4, bytes, bytes, bytes / 12 * 12, bytes, (size_t)(1 << 20) * 60 * 4 * 8);
    return 0;
}
