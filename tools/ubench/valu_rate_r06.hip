// Issue cost of every VALU instruction class in k_edge_msg's hot loops (gfx950), 2 waves per SIMD (the kernel's occupancy) and 8:
// wall-clock per wave64 instruction per SIMD, 8 independent chains per wave.  r01's table had v_fma / v_exp / v_rcp / v_pk_*_f32;
// this adds the mixed-precision, conversion and packed-fp16 forms the producer is made of, and DPP / permute forms of the epilogue.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate_r06.hip -o valu_rate_r06
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ void k(float *out, int iters)
{
    float a[8]; unsigned h[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; h[i] = 0x3c003800u + threadIdx.x + i; }
    const float w = out[0], r = out[1];
    const float ws = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, out[2])));
    const unsigned long long mask = __builtin_amdgcn_read_exec() >> 1;
    const double wpair = __builtin_bit_cast(double, ((unsigned long long)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, out[3])) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, out[2])));
    for (int it = 0; it < iters; ++it) {
#define OPI(i) \
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(r)); \
        if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(a[i]) : "v"(w), "v"(a[i]), "v"(h[i])); \
        if (OP == 2) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a[i]) : "v"(h[i]), "v"(a[i])); \
        if (OP == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 7])); \
        if (OP == 4) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(a[(i + 1) & 7])); \
        if (OP == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(a[(i + 1) & 7])); \
        if (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w)); \
        if (OP == 9) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
        if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[i & 6])) : "v"(*reinterpret_cast<const double *>(&a[(i + 2) & 6]))); \
        if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); \
        if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w)); \
        if (OP == 13) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 7])); \
        if (OP == 14) asm volatile("v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w)); \
        if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 7])); \
        if (OP == 16) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(ws)); \
        if (OP == 17) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(ws)); \
        if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "s"(mask)); \
        if (OP == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(*reinterpret_cast<double *>(&a[i & 6])) : "s"(wpair)); \
        if (OP == 20) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w) : "vcc"); \
        if (OP == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
        REP8(OPI)
#undef OPI
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
}
int main()
{
    float *out; (void)hipMalloc(&out, 1 << 24); (void)hipMemset(out, 0, 1 << 24);
    const char *names[] = {"v_fma_f32", "v_fma_mix_f32 (f32,f32,f16)", "v_fma_mix_f32 (f16 hi,1.0,f32)", "v_pk_add_f16", "v_cvt_pkrtz_f16_f32", "v_cvt_pk_f16_f32",
                           "v_exp_f32", "v_rcp_f32", "v_mul_f32", "v_add_f32 dpp quad_perm", "v_pk_mul_f32", "v_mov_b32", "v_cndmask_b32", "v_pk_max_f16",
                           "v_exp + s_nop + dependent v_fma (2 instr)", "v_add_u32",
                           "v_fma_f32 with an SGPR operand", "v_mul_f32 with an SGPR operand", "v_cndmask_b32_e64 (SGPR-pair condition)", "v_pk_fma_f32 with an SGPR-pair operand",
                           "v_cmp + v_cndmask vcc (2 instr)", "v_max_f32"};
    const int iters = 20000;
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void (*ks[])(float *, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>};
    for (int op = 0; op < 22; ++op)
        for (int wps = 2; wps <= 8; wps *= 4) {
            dim3 g(wps == 8 ? 2 * cus : cus), b(wps == 8 ? 1024 : 256 * wps);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(ks[op], g, b, 0, 0, out, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double n = (double)iters * 8 * wps * ((op == 14 || op == 20) ? 2 : 1);
            printf("%-44s waves/SIMD %d : %.2f cycles @2.4 GHz per wave64 instruction per SIMD\n", names[op], wps, best * 1e6 / n * 2.4);
        }
    return 0;
}
