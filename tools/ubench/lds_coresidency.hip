// Stand-alone reproducer attempt for the cross-stream interference of r05 (profiles/r05_concurrency.txt): does a kernel WITHOUT any
// LDS allocation ("victim": per-thread gathers + a dihedral through atan2f, like k_edge_feat<0>) compute different results while a
// persistent kernel holding ALL of a CU's LDS ("aggressor") runs on another stream?
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lds_coresidency.hip -o lds_coresidency
//   run:   ./lds_coresidency            -> per aggressor flavour and victim LDS size: calls of the victim that differ from its solo result
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct v3 { float x, y, z; };
__device__ inline v3 vsub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline v3 vcross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ inline float vnorm(v3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ inline v3 vdivs(v3 a, float s) { return v3{a.x / s, a.y / s, a.z / s}; }
__device__ inline float dihedral_deg(v3 a, v3 b, v3 c, v3 d)
{
    const v3 b1 = vsub(a, b), b2 = vsub(b, c), b3 = vsub(c, d);
    v3 n1 = vcross(b1, b2); n1 = vdivs(n1, vnorm(n1));
    v3 n2 = vcross(b2, b3); n2 = vdivs(n2, vnorm(n2));
    const v3 m1 = vcross(n1, vdivs(b2, vnorm(b2)));
    return atan2f(vdot(m1, n2), vdot(n1, n2)) * 180.0f / 3.14159265358979323846f;
}
__device__ inline int bin24(float a) { int b = 0; for (int i = 0; i < 23; ++i) b += (a > -180.0f + 360.0f / 22.0f * (float)i) ? 1 : 0; return b; }

// victim: one thread per edge (i = e / K, j = edges[e]); three float4 arrays; writes a packed code of three angle bins
__global__ __launch_bounds__(256) void k_victim(const float4 *__restrict__ n4, const float4 *__restrict__ ca4, const float4 *__restrict__ cb4,
                                                const int *__restrict__ edges, long long total, int K, unsigned *__restrict__ codes)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e / K), j = edges[e];
    const float4 ni = n4[i], cai = ca4[i], caj = ca4[j], cbi = cb4[i], cbj = cb4[j];
    const v3 N{ni.x, ni.y, ni.z}, Ci{cai.x, cai.y, cai.z}, Cj{caj.x, caj.y, caj.z}, Bi{cbi.x, cbi.y, cbi.z}, Bj{cbj.x, cbj.y, cbj.z};
    const float om = dihedral_deg(Ci, Bi, Bj, Cj), th = dihedral_deg(N, Ci, Bi, Bj);
    const v3 v1 = vsub(Ci, Bi), v2 = vsub(Bj, Bi);
    const float ph = acosf(vdot(v1, v2) / (vnorm(v1) * vnorm(v2))) * 180.0f / 3.14159265358979323846f;
    codes[e] = (unsigned)bin24(om) | ((unsigned)bin24(th) << 5) | ((unsigned)(int)(ph / 18.0f) << 10);
}

// aggressor: persistent, one 512-thread workgroup per CU holding `lds_bytes` of dynamic LDS for ~`iters` rounds of
//   flavour 0: LDS traffic + transcendental VALU work;  1: the same + MFMA;  2: the same + streaming global loads / nt stores
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_aggressor(int flavour, int iters, const float4 *__restrict__ src, long long nsrc, float *__restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    float a = tid * 1e-3f + 1.0f;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 fa, fb;
    for (int r = 0; r < 8; ++r) { fa[r] = (_Float16)(0.01f * (tid + r)); fb[r] = (_Float16)(0.02f * (tid - r)); }
    long long p = ((long long)blockIdx.x * 512 + tid) % nsrc;
    for (int it = 0; it < iters; ++it) {
        lds[(tid * 33 + it) & 32767] = a;
        __syncthreads();
        a = a * 0.999f + lds[(tid * 17 + 5 * it) & 32767];
        a = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-a * 1e-3f));
        if (flavour >= 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        if (flavour >= 2) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src + p));
            a += v.x * 1e-6f;
            p = (p + 4099) % nsrc;
        }
        __syncthreads();
    }
    asm volatile("" ::: "v255");      // the whole 256-register budget, like the message kernel: two of these waves fill a SIMD's register file
    float s = a;
    for (int r = 0; r < 16; ++r) s += acc[r] * 1e-9f;
    __builtin_nontemporal_store(s, sink + (size_t)blockIdx.x * 512 + tid);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int N = 15800, K = 60;
    const long long E = (long long)N * K;
    std::vector<float4> n4(N), ca(N), cb(N);
    std::vector<int> edges(E);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    for (int i = 0; i < N; ++i) {
        ca[i] = make_float4(40 * rnd(), 40 * rnd(), 40 * rnd(), 0);
        n4[i] = make_float4(ca[i].x + rnd() - 0.5f, ca[i].y + 1.2f, ca[i].z + rnd() - 0.5f, 0);
        cb[i] = make_float4(ca[i].x + 1.1f, ca[i].y + rnd() - 0.5f, ca[i].z + rnd() - 0.5f, 0);
    }
    for (long long e = 0; e < E; ++e) edges[e] = rand() % N;
    float4 *dn, *dca, *dcb, *dsrc; int *de; unsigned *dcodes; float *dsink;
    const long long nsrc = 1 << 22;
    CK(hipMalloc(&dn, N * 16)); CK(hipMalloc(&dca, N * 16)); CK(hipMalloc(&dcb, N * 16)); CK(hipMalloc(&de, E * 4));
    CK(hipMalloc(&dcodes, E * 4)); CK(hipMalloc(&dsrc, nsrc * 16)); CK(hipMalloc(&dsink, 512 * 512 * 4));
    CK(hipMemcpy(dn, n4.data(), N * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dca, ca.data(), N * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcb, cb.data(), N * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(de, edges.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dsrc, 0, nsrc * 16));
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<unsigned> solo(E), got(E);
    const unsigned grid = (unsigned)((E + 255) / 256);
    hipLaunchKernelGGL(k_victim, dim3(grid), dim3(256), 0, sa, dn, dca, dcb, de, E, K, dcodes);
    CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(solo.data(), dcodes, E * 4, hipMemcpyDeviceToHost));
    const int lds_sizes[] = {160 * 1024, 128 * 1024, 96 * 1024};
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20000;
    for (int flavour = 0; flavour < 3; ++flavour)
        for (int li = 0; li < 3; ++li)
            for (int vlds = 0; vlds <= 64; vlds += 64) {
                int bad = 0; long long worst = 0;
                const int reps = 16;
                for (int rep = 0; rep < reps; ++rep) {
                    hipLaunchKernelGGL(k_aggressor, dim3(cus), dim3(512), lds_sizes[li], sb, flavour, iters, dsrc, nsrc, dsink);
                    CK(hipMemsetAsync(dcodes, 0, E * 4, sa));
                    hipLaunchKernelGGL(k_victim, dim3(grid), dim3(256), vlds, sa, dn, dca, dcb, de, E, K, dcodes);
                    CK(hipStreamSynchronize(sa));
                    CK(hipMemcpy(got.data(), dcodes, E * 4, hipMemcpyDeviceToHost));
                    long long nd = 0;
                    for (long long e = 0; e < E; ++e) nd += got[e] != solo[e];
                    bad += nd != 0; worst = nd > worst ? nd : worst;
                    CK(hipStreamSynchronize(sb));
                }
                printf("aggressor flavour %d (LDS %3d KiB per workgroup), victim LDS %2d B: %2d of %d victim calls differ from solo (worst: %lld of %lld codes)\n",
                       flavour, lds_sizes[li] / 1024, vlds, bad, reps, worst, E);
                fflush(stdout);
            }
    return 0;
}
