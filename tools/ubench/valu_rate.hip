// Issue-rate microbenchmark for the VALU instruction classes the edge kernel is made of (gfx950):
// cycles per wave64 instruction with 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(float *out, int iters, long long *cyc)
{
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i] + 0.5f}; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 3) p[i] = p[i] * (f2){1.0001f, 0.9999f} + (f2){0.5f, 0.25f};
            if (OP == 4) p[i] = p[i] * (f2){1.0001f, 0.9999f};
            if (OP == 5) { _Float16 h = (_Float16)a[i]; a[i] = (float)h + 1.0f; }
            if (OP == 6) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[(i + 1) & 7] = __builtin_fmaf(a[(i + 1) & 7], 1.0001f, 0.5f); }
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const char *names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_fma_f32", "v_pk_mul_f32", "cvt f32->f16->f32 + add", "exp + fma interleaved (2 instr)"};
    const int iters = 20000;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 7; ++op)
        for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD: one block of 256*wps threads per CU (8: two blocks of 1024)
            dim3 g(wps == 8 ? 2 * cus : cus), b(wps == 8 ? 1024 : 256 * wps);
            hipEventRecord(e0);
            switch (op) {
            case 0: k<0><<<g, b>>>(out, iters, cyc); break; case 1: k<1><<<g, b>>>(out, iters, cyc); break;
            case 2: k<2><<<g, b>>>(out, iters, cyc); break; case 3: k<3><<<g, b>>>(out, iters, cyc); break;
            case 4: k<4><<<g, b>>>(out, iters, cyc); break; case 5: k<5><<<g, b>>>(out, iters, cyc); break;
            case 6: k<6><<<g, b>>>(out, iters, cyc); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double instr_per_simd = (double)iters * 8 * wps * (op == 6 ? 2 : (op == 5 ? 3 : 1));
            printf("%-34s waves/SIMD %d : clock64 %.2f ticks/instr/wave | wall %.3f ms -> %.2f ns per wave64 instr per SIMD (%.2f cycles @2.4 GHz)\n",
                   names[op], wps, (double)h / iters / 8 / (op == 6 ? 2 : (op == 5 ? 3 : 1)), ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
