// Minimal victim for the cross-handle miscompute of r05 / r06 (profiles/r06_concurrency.txt).  The ISA-level bisection of k_edge_feat<0>
// (tools/asm_variant.py) ends at ONE instruction of the SLP-vectorised dihedral:
//     v_pk_mul_f32 v[24:25], v[2:3], v[24:25] op_sel:[0,1] op_sel_hi:[1,0]
// a packed fp32 multiply whose DESTINATION pair is also a SOURCE pair read with CROSSED halves (D.lo = S0.lo * S1.hi, D.hi = S0.hi * S1.lo).
// This library runs exactly that form on known data next to whatever else the process has on the GPU (the real engine in another thread,
// tools/pkmul_probe.py) and counts the elements whose packed result differs from two plain v_mul_f32 on the same inputs:
//   mode 0  in place, crossed halves (the instruction of the bisection)
//   mode 1  crossed halves, separate destination registers
//   mode 2  in place, straight halves
//   mode 3  in place, crossed, S0 / S1 roles swapped (v_pk_mul_f32 D, D, S op_sel:[1,0] op_sel_hi:[0,1])
//   modes 4..9 (separate destination): which selector bit matters -
//     4  op_sel:[0,1]                  lo = S0.lo * S1.HI, hi = S0.hi * S1.hi        5  op_sel_hi:[1,0]   lo = S0.lo * S1.lo, hi = S0.hi * S1.LO (the library's broadcast form)
//     6  op_sel:[1,0]                  lo = S0.HI * S1.lo, hi = S0.hi * S1.hi        7  v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]
//     8  v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (+ 0.25 in both halves)     9  op_sel:[1,1] op_sel_hi:[0,0]   both sources crossed
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC pkmul_victim.hip -o libpkmul.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void k_pk(const f2 *__restrict__ A, const f2 *__restrict__ Bv, long long n, unsigned long long *bad, int rounds)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    f2 a = A[t], b = Bv[t];
    unsigned wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const f2 a0 = a, b0 = b;
        f2 e;      // expected, from plain multiplies
        if (MODE == 2) { e.x = a0.x * b0.x; e.y = a0.y * b0.y; }
        else if (MODE == 4) { e.x = a0.x * b0.y; e.y = a0.y * b0.y; }
        else if (MODE == 5) { e.x = a0.x * b0.x; e.y = a0.y * b0.x; }
        else if (MODE == 6) { e.x = a0.y * b0.x; e.y = a0.y * b0.y; }
        else if (MODE == 7) { e.x = a0.x + b0.y; e.y = a0.y + b0.x; }
        else if (MODE == 8) { e.x = __builtin_fmaf(a0.x, b0.y, 0.25f); e.y = __builtin_fmaf(a0.y, b0.x, 0.25f); }
        else if (MODE == 9) { e.x = a0.y * b0.y; e.y = a0.x * b0.x; }
        else { e.x = a0.x * b0.y; e.y = a0.y * b0.x; }
        asm volatile("" : "+v"(e));
        f2 d = b;
        if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(a));
        if (MODE == 1) { f2 o; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(o) : "v"(a), "v"(d)); d = o; }
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(d) : "v"(a));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(d) : "v"(a));
        if (MODE >= 4) {
            f2 o; const f2 q = {0.25f, 0.25f};
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(o) : "v"(a), "v"(d));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(o) : "v"(a), "v"(d));
            if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(o) : "v"(a), "v"(d));
            if (MODE == 7) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(o) : "v"(a), "v"(d));
            if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(o) : "v"(a), "v"(d), "v"(q));
            if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=&v"(o) : "v"(a), "v"(d));
            d = o;
        }
        wrong += (__float_as_uint(d.x) != __float_as_uint(e.x)) || (__float_as_uint(d.y) != __float_as_uint(e.y));
        // next round's operands: keep magnitudes near 1 (no overflow / denormals over many rounds)
        b = (f2){e.y * 0.5f + 0.3f, e.x * 0.5f + 0.2f};
        a = (f2){a0.y, a0.x};
        b.x = b.x > 4.0f ? b.x * 0.125f : b.x; b.y = b.y > 4.0f ? b.y * 0.125f : b.y;
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

static const long long N = 948000;
static f2 *dA, *dB; static unsigned long long *dbad; static hipStream_t sa;

extern "C" int pk_init()
{
    std::vector<f2> a(N), b(N);
    unsigned s = 12345u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return 0.5f + (float)(s >> 8) / 16777216.0f; };
    for (long long i = 0; i < N; ++i) { a[i] = (f2){rnd(), rnd()}; b[i] = (f2){rnd(), rnd()}; }
    if (hipMalloc(&dA, N * 8) || hipMalloc(&dB, N * 8) || hipMalloc(&dbad, 8)) return -1;
    hipMemcpy(dA, a.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), N * 8, hipMemcpyHostToDevice);
    if (hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)) return -2;
    return 0;
}
// launches the victim `reps` times with `lds` bytes of dynamic LDS; returns the number of launches with at least one wrong element, *wrong_total = wrong elements in all
extern "C" int pk_run(int mode, int reps, int lds, int rounds, long long *wrong_total)
{
    int bad_launches = 0; long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(dbad, 0, 8, sa);
        const dim3 g((unsigned)((N + 255) / 256)), b(256);
        if (mode == 0) hipLaunchKernelGGL(k_pk<0>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 1) hipLaunchKernelGGL(k_pk<1>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 2) hipLaunchKernelGGL(k_pk<2>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 3) hipLaunchKernelGGL(k_pk<3>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 4) hipLaunchKernelGGL(k_pk<4>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 5) hipLaunchKernelGGL(k_pk<5>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 6) hipLaunchKernelGGL(k_pk<6>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 7) hipLaunchKernelGGL(k_pk<7>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 8) hipLaunchKernelGGL(k_pk<8>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        if (mode == 9) hipLaunchKernelGGL(k_pk<9>, g, b, lds, sa, dA, dB, N, dbad, rounds);
        unsigned long long h = 0;
        hipMemcpyAsync(&h, dbad, 8, hipMemcpyDeviceToHost, sa);
        hipStreamSynchronize(sa);
        if (h) { ++bad_launches; tot += (long long)h; }
    }
    if (wrong_total) *wrong_total = tot;
    return bad_launches;
}
