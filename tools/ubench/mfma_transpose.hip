// Is v_mfma_f32_32x32x16_bf16 symmetric in its operands?  D1 = A x B and D2 = B^T x A^T (operand registers swapped) must satisfy
// D1[m][n] == D2[n][m] bit for bit if the k-summation order does not depend on the operand role.  Also: three dependent MFMAs
// (lo x hi, hi x lo, hi x hi) accumulated in one register, both ways.   hipcc --offload-arch=gfx950 -O3 mfma_transpose.hip -o mfma_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const uint4 *A, const uint4 *B, float *D1, float *D2, int steps)
{
    const int lane = threadIdx.x;
    f32x16 c1, c2;
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; }
    for (int s = 0; s < steps; ++s) {
        union { uint4 u; bf16x8 b; } a, b;
        a.u = A[s * 64 + lane]; b.u = B[s * 64 + lane];
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b, a.b, c2, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        D1[row * 32 + col] = c1[r];      // D1[m][n]
        D2[col * 32 + row] = c2[r];      // D2[m'][n'] with m' = row index of (B^T A^T) = n of D1: store transposed
    }
}
int main()
{
    const int steps = 32;
    std::vector<uint16_t> ha(steps * 64 * 8), hb(steps * 64 * 8);
    srand(3);
    auto rnd = [] { float x = (float)rand() / RAND_MAX * 4.f - 2.f; uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); };
    for (auto &v : ha) v = rnd();
    for (auto &v : hb) v = rnd();
    uint4 *A, *B; float *D1, *D2;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&D1, 4096); hipMalloc(&D2, 4096);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D1, D2, steps);
    float h1[1024], h2[1024];
    hipMemcpy(h1, D1, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2, D2, 4096, hipMemcpyDeviceToHost);
    int diff = 0; double mx = 0;
    for (int i = 0; i < 1024; ++i) { if (memcmp(&h1[i], &h2[i], 4)) ++diff; double d = h1[i] - h2[i]; if (d < 0) d = -d; if (d > mx) mx = d; }
    printf("32 accumulated k-steps: %d of 1024 entries differ between A x B and (B x A)^T, max |diff| %.3g (values up to ~%.3g)\n", diff, mx, (double)h1[0]);
    return 0;
}
