// Stand-alone hunt for the AGGRESSOR side of the packed-fp32 op_sel = [0,1] miscompute (profiles/r06_concurrency.txt): the victim is the
// minimal kernel of pkmul_victim.hip (v_pk_mul_f32 with op_sel:[0,1] on known data, no LDS); the aggressor here is SYNTHETIC - a persistent
// 512-thread workgroup per CU holding 160 KiB of LDS that loops one instruction mix - instead of the engine's message kernel.  Which mix
// (if any) makes the victim fail tells what in the message kernel sets the erratum off.
// Build: hipcc --offload-arch=gfx950 -O3 pk_erratum.hip -o pk_erratum ; run: ./pk_erratum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM> __global__ __launch_bounds__(256) void k_victim(const f2 *__restrict__ A, const f2 *__restrict__ Bv, long long n, unsigned long long *bad, int rounds)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    f2 a = A[t], b = Bv[t];
    unsigned wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const f2 a0 = a, b0 = b;
        f2 e = FORM == 0 ? (f2){a0.x * b0.y, a0.y * b0.y} : (FORM == 1 ? (f2){a0.x * b0.x, a0.y * b0.x} : (f2){a0.x * b0.x, a0.y * b0.y});
        asm volatile("" : "+v"(e));
        f2 o;
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(o) : "v"(a), "v"(b));          // the failing form
        if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(o) : "v"(a), "v"(b));       // the library's form
        if (FORM == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(o) : "v"(a), "v"(b));
        if (FORM >= 3) {      // the other VOP3P forms the library's message kernel executes next to its partner wave's MFMAs all the time
            const unsigned hb = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b0.x, b0.y));      // (lo, hi) fp16 of b
            const float blo = (float)__builtin_bit_cast(_Float16, (unsigned short)(hb & 0xffffu)), bhi = (float)__builtin_bit_cast(_Float16, (unsigned short)(hb >> 16));
            if (FORM == 3) { e = (f2){a0.x + blo, a0.y + bhi};      // add_half_lo / add_half_hi
                asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(o.x) : "v"(hb), "v"(a.x));
                asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(o.y) : "v"(hb), "v"(a.y)); }
            if (FORM == 4) { e = (f2){__builtin_fmaf(a0.x, a0.y, blo), __builtin_fmaf(a0.y, a0.x, bhi)};      // fma_half_lo / fma_half_hi
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=&v"(o.x) : "v"(a.x), "v"(a.y), "v"(hb));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(o.y) : "v"(a.y), "v"(a.x), "v"(hb)); }
            if (FORM == 5) { e = (f2){__builtin_fmaf(a0.x, b0.x, b0.x), __builtin_fmaf(a0.y, b0.x, b0.x)};      // v_pk_fma_f32 op_sel_hi:[1,0,0]
                asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel_hi:[1,0,0]" : "=&v"(o) : "v"(a), "v"(b)); }
            if (FORM == 6) { e = (f2){__builtin_fmaf(a0.x, b0.x, b0.x), __builtin_fmaf(a0.x, b0.y, b0.y)};      // v_pk_fma_f32 op_sel_hi:[0,1,1]
                asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel_hi:[0,1,1]" : "=&v"(o) : "v"(a), "v"(b)); }
            if (FORM == 7) { e = (f2){a0.x + 1.0f, a0.y + 1.0f};                                                  // v_pk_add_f32 with an inline constant, op_sel_hi:[1,0]
                asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=&v"(o) : "v"(a)); }
            if (FORM == 8) {      // v_pk_add_f16 / v_pk_max_f16 op_sel_hi:[1,0] on packed halves, checked bit for bit against per-half arithmetic
                const unsigned ha = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0.x, a0.y));
                const _Float16 al = __builtin_bit_cast(_Float16, (unsigned short)(ha & 0xffffu)), ah2 = __builtin_bit_cast(_Float16, (unsigned short)(ha >> 16));
                const _Float16 bl = __builtin_bit_cast(_Float16, (unsigned short)(hb & 0xffffu)), bh2 = __builtin_bit_cast(_Float16, (unsigned short)(hb >> 16));
                const _Float16 s0 = al + bl, s1 = ah2 + bh2;
                const _Float16 m0 = s0 > bl ? s0 : bl, m1 = s1 > bl ? s1 : bl;
                e = (f2){(float)m0, (float)m1};
                unsigned r1, r2;
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=&v"(r1) : "v"(ha), "v"(hb));
                asm volatile("v_pk_max_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r2) : "v"(r1), "v"(hb));
                o = (f2){(float)__builtin_bit_cast(_Float16, (unsigned short)(r2 & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(r2 >> 16))}; }
            asm volatile("" : "+v"(e));
        }
        wrong += (__float_as_uint(o.x) != __float_as_uint(e.x)) || (__float_as_uint(o.y) != __float_as_uint(e.y));
        b = (f2){e.y * 0.25f + 0.3f, e.x * 0.25f + 0.2f};
        b.x = b.x > 4.0f ? 0.5f : b.x; b.y = b.y > 4.0f ? 0.75f : b.y;
        a = (f2){a0.y, a0.x};
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

template <int MIX> __global__ __launch_bounds__(512) void k_aggr(float *out, int iters, volatile int *stop)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    lds[tid] = (float)tid;
    __syncthreads();
    f2 p[4]; float s[4]; unsigned h[4];
    for (int i = 0; i < 4; ++i) { p[i] = (f2){1.0f + tid * 1e-4f + i, 0.5f + i}; s[i] = 1.0f + i * 0.25f; h[i] = 0x3c003800u + tid + i; }
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 ah, bh; for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(0.01f * e); bh[e] = (_Float16)(0.02f * e); }
    const f2 c = {0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MIX == 1 || MIX == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
            if (MIX == 2 || MIX == 10) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(c));
            if (MIX == 3 || MIX == 10) asm volatile("v_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(p[i]));
            if (MIX == 4 || MIX == 10) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(c));
            if (MIX == 6 || MIX == 10) {
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
                s[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s[i]), 0x401F));
                s[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((tid & 63) * 4 ^ 4, __builtin_bit_cast(int, s[i])));
            }
            if (MIX == 7 || MIX == 10) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s[i]) : "v"(h[i]));
            if (MIX == 8 || MIX == 10) { asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 3])); asm volatile("v_pk_max_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(h[i]) : "v"(h[(i + 2) & 3])); }
            if (MIX == 9 || MIX == 10) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(s[i]), "v"(s[(i + 1) & 3]));
            if (MIX == 11 || MIX == 10) { s[i] = __builtin_amdgcn_exp2f(s[i] * 0.001f); s[i] = __builtin_amdgcn_rcpf(1.0f + s[i]); }
            p[i] = (f2){p[i].x * 0.5f + 0.7f, p[i].y * 0.5f + 0.6f};
            s[i] = s[i] * 0.5f + 0.6f;
        }
        if (MIX == 5 || MIX == 10) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if (MIX == 13) { typedef __bf16 bf16x8 __attribute__((ext_vector_type(8))); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0); }
        if (MIX == 14) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[0], s[1], acc, 0, 0, 0);
        if (MIX == 15) { typedef float f32x4 __attribute__((ext_vector_type(4))); f32x4 a4 = {acc[0], acc[1], acc[2], acc[3]}; a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a4, 0, 0, 0); acc[0] = a4[0]; acc[1] = a4[1]; acc[2] = a4[2]; acc[3] = a4[3]; }
        if (MIX == 12 || MIX == 10) s[0] += lds[(tid * 33 + it) & 8191];
        if (MIX == 0) __builtin_amdgcn_s_sleep(16);
        if ((it & 1023) == 0 && *stop) break;
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += p[i].x + p[i].y + s[i] + (float)h[i];
    for (int k = 0; k < 16; ++k) r += acc[k];
    out[blockIdx.x * 512 + tid] = r;
}

static const long long N = 948000;
int main(int argc, char **)
{
    std::vector<f2> a(N), b(N);
    unsigned sd = 12345u;
    auto rnd = [&] { sd = sd * 1664525u + 1013904223u; return 0.5f + (float)(sd >> 8) / 16777216.0f; };
    for (long long i = 0; i < N; ++i) { a[i] = (f2){rnd(), rnd()}; b[i] = (f2){rnd(), rnd()}; }
    f2 *dA, *dB; unsigned long long *dbad; float *dout; int *dstop;
    (void)hipMalloc(&dA, N * 8); (void)hipMalloc(&dB, N * 8); (void)hipMalloc(&dbad, 8); (void)hipMalloc(&dout, 1 << 22); (void)hipHostMalloc(&dstop, 4);
    (void)hipMemcpy(dA, a.data(), N * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dB, b.data(), N * 8, hipMemcpyHostToDevice);
    hipStream_t sa, sv; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int LDS = 160 * 1024;
    const char *names[] = {"idle spin (s_sleep), 160 KiB LDS held", "v_pk_mul_f32 plain", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_add_f32 1.0 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel_hi:[0,1,1]",
                           "v_mfma_f32_32x32x16_f16", "DPP add + ds_swizzle + ds_bpermute", "v_fma_mix_f32", "v_pk_add_f16 + v_pk_max_f16 op_sel_hi:[1,0]", "v_cvt_pkrtz_f16_f32",
                           "everything together", "v_exp_f32 + v_rcp_f32", "LDS reads", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_f16"};
    void (*ks[])(float *, int, volatile int *) = {k_aggr<0>, k_aggr<1>, k_aggr<2>, k_aggr<3>, k_aggr<4>, k_aggr<5>, k_aggr<6>, k_aggr<7>, k_aggr<8>, k_aggr<9>, k_aggr<10>, k_aggr<11>, k_aggr<12>, k_aggr<13>, k_aggr<14>, k_aggr<15>};
    const bool only_mfma = argc > 1;
    for (int mix = -1; mix < 16; ++mix) {
        if (only_mfma && !(mix < 0 || mix == 5 || mix == 13)) continue;
        for (int lds_try = 0; lds_try < (only_mfma ? 1 : 2); ++lds_try) {
            const int lds = lds_try == 0 ? LDS : 64 * 1024;
            if (mix < 0 && lds_try) continue;
            *dstop = 0;
            if (mix >= 0) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks[mix]), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                hipLaunchKernelGGL(ks[mix], dim3(cus), dim3(512), lds, sa, dout, 1 << 30, dstop);
            }
            const bool mfma = mix < 0 || mix == 5 || mix == 13 || mix == 14 || mix == 15;      // (no aggressor: every victim form must be clean)
            for (int form = 0; form < (mfma ? 9 : 1); ++form) for (int vlds = 0; vlds <= (mfma && form == 0 ? 64 : 0); vlds += 64) {
                int bad_launches = 0; long long tot = 0;
                for (int r = 0; r < 32; ++r) {
                    (void)hipMemsetAsync(dbad, 0, 8, sv);
                    const dim3 g((unsigned)((N + 255) / 256)), bl(256);
                    if (form == 0) hipLaunchKernelGGL(k_victim<0>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 1) hipLaunchKernelGGL(k_victim<1>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 2) hipLaunchKernelGGL(k_victim<2>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 3) hipLaunchKernelGGL(k_victim<3>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 4) hipLaunchKernelGGL(k_victim<4>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 5) hipLaunchKernelGGL(k_victim<5>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 6) hipLaunchKernelGGL(k_victim<6>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 7) hipLaunchKernelGGL(k_victim<7>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    if (form == 8) hipLaunchKernelGGL(k_victim<8>, g, bl, vlds, sv, dA, dB, N, dbad, 64);
                    unsigned long long hbad = 0;
                    (void)hipMemcpyAsync(&hbad, dbad, 8, hipMemcpyDeviceToHost, sv);
                    (void)hipStreamSynchronize(sv);
                    if (hbad) { ++bad_launches; tot += (long long)hbad; }
                }
                printf("aggressor %-46s LDS %3d KiB | victim %-28s LDS %2d B: wrong in %2d of 32 launches (%lld element-rounds)\n", mix < 0 ? "none" : names[mix], mix < 0 ? 0 : lds / 1024,
                       form == 0 ? "v_pk_mul_f32 op_sel:[0,1]" : (form == 1 ? "v_pk_mul_f32 op_sel_hi:[1,0]" : (form == 2 ? "v_pk_mul_f32 plain" : (form == 3 ? "v_fma_mix (f16 lo / hi) + f32" :
                       (form == 4 ? "v_fma_mix f32*f32 + f16 lo/hi" : (form == 5 ? "v_pk_fma_f32 op_sel_hi:[1,0,0]" : (form == 6 ? "v_pk_fma_f32 op_sel_hi:[0,1,1]" : (form == 7 ? "v_pk_add_f32 1.0 op_sel_hi" : "v_pk_add_f16 + v_pk_max_f16"))))))), vlds, bad_launches, tot);
                fflush(stdout);
            }
            *dstop = 1;
            (void)hipStreamSynchronize(sa);
        }
    }
    return 0;
}
