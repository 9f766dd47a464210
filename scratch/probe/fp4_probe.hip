// hardware probe: v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (E2M1) operands as an exact AND-popcount engine.
//   A nibble = bit ? 0x2 (+1.0) : 0, B nibble = bit ? 0xC (-2.0) : 0, scale_a = 2^12 (E8M0 139), scale_b = 2^0 (127):
//   D[r][c] = C[r][c] - 8192 * popcount(a_r & b_c) over the 128 bits a lane group holds.
// Both operands are expanded by the same function, so the order of the K elements inside the instruction does not matter; what the
// probe pins is the row / column <-> lane map, the K-group <-> lane map (lane >> 4), the scale operands and exactness.
// Build: hipcc --offload-arch=gfx950 -O2 -o fp4_probe fp4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ unsigned spread8(unsigned b, unsigned code)          // 8 bits -> 8 nibbles (bit i -> nibble i), each set nibble = code
{
    unsigned x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x * code;
}

__global__ void k(const uint32_t* A /*16 rows x 4 dwords (128 bits: K-group g = dword g)*/, const uint32_t* B, const float* C, float* D, int sa, int sb)
{
    const int lane = threadIdx.x, rc = lane & 15, g = lane >> 4;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    // 32 elements per lane = 32 bits of the row: ONE dword of the descriptor slice per lane and K-group
    const uint32_t wa = A[rc * 4 + g], wb = B[rc * 4 + g];      // row rc, K-group g: 32 bits
    for (int w = 0; w < 4; ++w) { a[w] = (int)spread8(wa >> (8 * w), 0x2u); b[w] = (int)spread8(wb >> (8 * w), 0xCu); }
    v4f c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * g + r) * 16 + rc];          // standard C/D map: col = lane & 15, row = 4 (lane >> 4) + r
    v4f d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4 /*A: fp4*/, 4 /*B: fp4*/, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + rc] = d[r];
}

int main()
{
    uint32_t hA[16 * 4], hB[16 * 4]; float hC[256], hD[256];
    srand(7);
    for (int i = 0; i < 64; ++i) { hA[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); hB[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); }
    for (int i = 0; i < 256; ++i) hC[i] = (float)((300 + i) * 4096 + i);
    uint32_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; ++variant) {
        // scale operand: the builtin takes an int whose selected byte is the E8M0 scale; try the byte replicated
        const int sa = variant == 0 ? 139 * 0x01010101 : variant == 1 ? 127 * 0x01010101 : 133 * 0x01010101;
        const int sb = variant == 2 ? 133 * 0x01010101 : 127 * 0x01010101;
        const double scale = variant == 1 ? 1.0 : 4096.0;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, sa, sb);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad_rowA = 0, bad_rowB = 0;
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            int pcAB = 0, pcBA = 0;
            for (int w = 0; w < 4; ++w) { pcAB += __builtin_popcount(hA[r * 4 + w] & hB[c * 4 + w]); pcBA += __builtin_popcount(hA[c * 4 + w] & hB[r * 4 + w]); }
            const double eAB = (double)hC[r * 16 + c] - 2.0 * scale * pcAB, eBA = (double)hC[r * 16 + c] - 2.0 * scale * pcBA;
            bad_rowA += hD[r * 16 + c] != (float)eAB; bad_rowB += hD[r * 16 + c] != (float)eBA;
        }
        printf("variant %d (sa %d sb %d): mismatches with A = rows: %d, with B = rows: %d   D[0][1] = %.1f C = %.1f\n", variant, sa & 255, sb & 255, bad_rowA, bad_rowB, hD[1], hC[1]);
    }
    return 0;
}
