// Issue rate of v_mfma_f32_32x32x2_f32 (and the 16-bit 32x32x16 form) from ONE wave per SIMD: eight independent accumulators, operands in
// registers, optionally one LDS read pair / one plain VALU instruction between consecutive MFMAs (what k_edge_f32m interleaves).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[8];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b[8];
    for (int n = 0; n < 8; ++n) b[n] = a + n;
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(a + e); bh[e] = (_Float16)(0.5f * e); }
    float v = a;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (MODE == 3) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
            if (MODE == 1) { b[n] = lds[(threadIdx.x + n * 32 + it) & 4095]; }
            if (MODE == 2) { v = __builtin_fmaf(v, 1.0001f, 0.5f); v = __builtin_fmaf(v, 1.0001f, 0.5f); v = __builtin_fmaf(v, 1.0001f, 0.5f); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = v;
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const char *names[] = {"mfma_f32_32x32x2_f32 back to back", "... + one LDS read per MFMA", "... + three v_fma per MFMA", "mfma_f32_32x32x16_f16 back to back"};
    const int iters = 4000;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 2; ++wps) {
            dim3 g(cus), bl(256 * wps);
            hipEventRecord(e0);
            if (wps == 1) { switch (mode) { case 0: k<0><<<g, 256>>>(out, iters, cyc); break; case 1: k<1><<<g, 256>>>(out, iters, cyc); break; case 2: k<2><<<g, 256>>>(out, iters, cyc); break; case 3: k<3><<<g, 256>>>(out, iters, cyc); break; } }
            else { dim3 g2(2 * cus); switch (mode) { case 0: k<0><<<g2, 256>>>(out, iters, cyc); break; case 1: k<1><<<g2, 256>>>(out, iters, cyc); break; case 2: k<2><<<g2, 256>>>(out, iters, cyc); break; case 3: k<3><<<g2, 256>>>(out, iters, cyc); break; } }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double per_simd = (double)iters * 8 * wps;
            const double flop = mode == 3 ? 32768.0 : 4096.0;
            printf("%-40s waves/SIMD %d : clock64 %.1f cycles per MFMA per wave | wall %.3f ms -> %.1f cycles per MFMA per SIMD @2.4 GHz = %.0f TFLOP/s\n",
                   names[mode], wps, (double)h / iters / 8, ms, ms * 1e6 / per_simd * 2.4, flop * per_simd * cus * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
