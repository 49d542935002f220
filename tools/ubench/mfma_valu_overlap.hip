// Do VALU instructions issue in the shadow of an MFMA of the same / another wave on gfx950?  For the two MFMA forms the engine uses
// (v_mfma_f32_32x32x2_f32: 16 passes; v_mfma_f32_32x32x16_f16: 8 passes), K independent v_fma_f32 (or K/2 v_exp_f32) between consecutive
// MFMAs, 1 and 2 waves per SIMD.  Overlap: cycles per MFMA = max(MFMA, VALU); no overlap: the sum.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int F16, int KV, int TRANS> __global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 acc[8];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b[8], v[16];
    for (int n = 0; n < 8; ++n) b[n] = a + n;
    for (int n = 0; n < 16; ++n) v[n] = a + 0.1f * n;
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(a + e); bh[e] = (_Float16)(0.5f * e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (F16) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < KV; ++q) {
                if (TRANS) v[q & 15] = __builtin_amdgcn_exp2f(v[q & 15]);
                else v[q & 15] = __builtin_fmaf(v[q & 15], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int n = 0; n < 16; ++n) s += v[n];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int F16, int KV, int TRANS> void run(float *out, int cus)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            k<F16, KV, TRANS><<<dim3(cus * wps), 256>>>(out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double per_simd = (double)iters * 8 * wps;
        printf("%s + %2d %s between MFMAs, waves/SIMD %d : %.1f cycles@2.4GHz per MFMA per SIMD (VALU alone would need %.0f)\n",
               F16 ? "mfma_f32_32x32x16_f16" : "mfma_f32_32x32x2_f32 ", KV, TRANS ? "v_exp_f32" : "v_fma_f32", wps, best * 1e6 / per_simd * 2.4, KV * (TRANS ? 12.2 : 3.2) * wps / wps);
    }
}
int main()
{
    float *out; (void)hipMalloc(&out, 1 << 24);
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    k<0, 0, 0><<<dim3(cus), 256>>>(out, 10); (void)hipDeviceSynchronize();      // warm-up: code object load
    run<0, 0, 0>(out, cus); run<0, 2, 0>(out, cus); run<0, 4, 0>(out, cus); run<0, 8, 0>(out, cus); run<0, 16, 0>(out, cus); run<0, 4, 1>(out, cus);
    run<1, 0, 0>(out, cus); run<1, 2, 0>(out, cus); run<1, 4, 0>(out, cus); run<1, 8, 0>(out, cus); run<1, 16, 0>(out, cus); run<1, 2, 1>(out, cus); run<1, 4, 1>(out, cus);
    return 0;
}
