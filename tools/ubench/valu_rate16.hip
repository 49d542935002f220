// Issue-rate microbenchmark, 16-bit instruction classes (gfx950): is the f16 transcendental path faster than f32?
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate16.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(float *out, int iters)
{
    float a[8]; _Float16 hx[8]; h2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i * 0.1f; hx[i] = (_Float16)a[i]; p[i] = (h2){hx[i], (_Float16)(a[i] * 0.5f)}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (OP == 1) asm volatile("v_exp_f16 %0, %1" : "=v"(hx[i]) : "v"(hx[i]));
            if (OP == 2) asm volatile("v_rcp_f16 %0, %1" : "=v"(hx[i]) : "v"(hx[i]));
            if (OP == 3) p[i] = p[i] * (h2){(_Float16)1.001f, (_Float16)0.999f} + (h2){(_Float16)0.5f, (_Float16)0.25f};
            if (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)hx[i] + (float)p[i].x + (float)p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 1 << 24);
    const char *names[] = {"v_exp_f32", "v_exp_f16", "v_rcp_f16", "v_pk_fma_f16", "v_rcp_f32"};
    const int iters = 20000;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 5; ++op)
        for (int wps = 2; wps <= 8; wps *= 2) {
            dim3 g(wps == 8 ? 2 * cus : cus), b(wps == 8 ? 1024 : 256 * wps);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (op) {
                case 0: k<0><<<g, b>>>(out, iters); break; case 1: k<1><<<g, b>>>(out, iters); break;
                case 2: k<2><<<g, b>>>(out, iters); break; case 3: k<3><<<g, b>>>(out, iters); break;
                case 4: k<4><<<g, b>>>(out, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 8 * wps;
            printf("%-22s waves/SIMD %d : %.2f cycles per wave64 instruction per SIMD (@2.4 GHz)\n", names[op], wps, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
