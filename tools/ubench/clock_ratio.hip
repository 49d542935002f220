// s_memtime (shader-clock counter) against s_memrealtime (constant 100 MHz) inside one kernel: the clock the chip holds under a given load.
//   hipcc --offload-arch=gfx950 -O3 -o clock_ratio clock_ratio.hip && ./clock_ratio
// Loads: 0 idle chip, one wave spinning on dependent FMAs; 1 every SIMD of every CU with two waves of independent v_fma chains; 2 the same with
// v_exp / v_rcp; 3 two waves per SIMD of v_mfma_f32_32x32x16_f16 back to back; 4 MFMA + VALU + LDS reads mixed (the message kernel's diet).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int LOAD> __global__ __launch_bounds__(512) void k(unsigned long long *out, float *sink, int iters)
{
    __shared__ h8 lds[2048];
    for (int q = threadIdx.x; q < 2048; q += blockDim.x) lds[q] = (h8){1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    float a[8];
    for (int q = 0; q < 8; ++q) a[q] = threadIdx.x * 1e-3f + q;
    f16v acc0 = {0}, acc1 = {0};
    h8 x = lds[threadIdx.x], y = lds[threadIdx.x + 512];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LOAD == 0 || LOAD == 1 || LOAD == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = __builtin_fmaf(a[q], 0.999f, 0.001f);
        }
        if (LOAD == 2 || LOAD == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a[q]));
        }
        if (LOAD == 3 || LOAD == 4) {
            if (LOAD == 4) { x = lds[(threadIdx.x + it) & 2047]; y = lds[(threadIdx.x * 3 + it) & 2047]; }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = acc0[0] + acc1[3];
    for (int q = 0; q < 8; ++q) s += a[q];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int LOAD> void run(const char *name, int grid, int block, int iters)
{
    unsigned long long *out; float *sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<LOAD>, dim3(grid), dim3(block), 0, 0, out, sink, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-44s shader-clock counts %12llu  100 MHz counts %10llu  -> %7.1f MHz over %.2f ms\n", name, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 1e5);
    hipFree(out); hipFree(sink);
}

int main()
{
    run<0>("one wave, dependent FMAs (idle chip)", 1, 64, 400000);
    run<1>("256 CUs x 8 waves, v_fma chains", 256, 512, 400000);
    run<2>("256 CUs x 8 waves, v_exp + v_rcp", 256, 512, 200000);
    run<3>("256 CUs x 8 waves, MFMA 32x32x16 f16", 256, 512, 200000);
    run<4>("256 CUs x 8 waves, MFMA + VALU + trans + LDS", 256, 512, 100000);
    run<0>("one wave again", 1, 64, 400000);
    return 0;
}
