// Synthetic victims for the cross-handle interference of r05, callable from Python next to the REAL engine (tools/concurrency_probe8.py):
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC victim_lib.hip -o libvictim.so
//   int victim_run(int variant, int reps, int lds_bytes): launches the victim `reps` times on its own non-blocking stream and returns the number of
//   launches whose output differs from the first (solo) launch made by victim_init().  Variants:
//     0  angle bins against COMPUTED boundaries (no constant memory)
//     1  angle bins against a __constant__ boundary array (scalar loads from the code object's data segment)
//     2  variant 1 without the dihedral: loads + compares only
#include <hip/hip_runtime.h>
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
__constant__ float c_bounds[23] = {
    -180.0f, -163.63636779785156f, -147.27273559570312f, -130.90908813476562f, -114.54545593261719f, -98.18182373046875f, -81.81818389892578f,
    -65.45454406738281f, -49.090911865234375f, -32.72727584838867f, -16.36363983154297f, 3.814697265625e-06f, 16.36363983154297f, 32.72727584838867f,
    49.090911865234375f, 65.45454406738281f, 81.81818389892578f, 98.18182373046875f, 114.54545593261719f, 130.90908813476562f, 147.27273559570312f,
    163.63636779785156f, 180.0f};
template <int V> __device__ inline int bin24(float a)
{
    int b = 0;
#pragma unroll
    for (int i = 0; i < 23; ++i) b += (a > (V == 0 ? -180.0f + 360.0f / 22.0f * (float)i : c_bounds[i])) ? 1 : 0;
    return b;
}
template <int V> __global__ __launch_bounds__(256) void k_victim(const float4 *__restrict__ n4, const float4 *__restrict__ ca4, const float4 *__restrict__ cb4,
                                                               const int *__restrict__ edges, long long total, int K, unsigned *__restrict__ codes)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e / K), j = edges[e];
    const float4 ni = n4[i], cai = ca4[i], caj = ca4[j], cbi = cb4[i], cbj = cb4[j];
    const v3 N{ni.x, ni.y, ni.z}, Ci{cai.x, cai.y, cai.z}, Cj{caj.x, caj.y, caj.z}, Bi{cbi.x, cbi.y, cbi.z}, Bj{cbj.x, cbj.y, cbj.z};
    float om, th;
    if (V == 2) { om = (ni.x + caj.y) * 4.0f - 90.0f; th = (cbi.z - cbj.x) * 4.0f; }
    else { om = dihedral_deg(Ci, Bi, Bj, Cj); th = dihedral_deg(N, Ci, Bi, Bj); }
    codes[e] = (unsigned)bin24<V>(om) | ((unsigned)bin24<V>(th) << 5);
}

static const int N = 15800, K = 60;
static const long long E = (long long)N * K;
static float4 *dn, *dca, *dcb; static int *de; static unsigned *dcodes;
static hipStream_t sa;
static std::vector<unsigned> solo[3];

template <int V> static void launch(int lds)
{
    hipLaunchKernelGGL(k_victim<V>, dim3((unsigned)((E + 255) / 256)), dim3(256), lds, sa, dn, dca, dcb, de, E, K, dcodes);
}
static void launch_v(int v, int lds) { if (v == 0) launch<0>(lds); else if (v == 1) launch<1>(lds); else launch<2>(lds); }

extern "C" int victim_init()
{
    std::vector<float4> n4(N), ca(N), cb(N);
    std::vector<int> edges(E);
    srand(7);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
    for (int i = 0; i < N; ++i) {
        ca[i] = make_float4(40 * rnd(), 40 * rnd(), 40 * rnd(), 0);
        n4[i] = make_float4(ca[i].x + rnd() - 0.5f, ca[i].y + 1.2f, ca[i].z + rnd() - 0.5f, 0);
        cb[i] = make_float4(ca[i].x + 1.1f, ca[i].y + rnd() - 0.5f, ca[i].z + rnd() - 0.5f, 0);
    }
    for (long long e = 0; e < E; ++e) edges[e] = rand() % N;
    if (hipMalloc(&dn, N * 16) || hipMalloc(&dca, N * 16) || hipMalloc(&dcb, N * 16) || hipMalloc(&de, E * 4) || hipMalloc(&dcodes, E * 4)) return -1;
    hipMemcpy(dn, n4.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(dca, ca.data(), N * 16, hipMemcpyHostToDevice);
    hipMemcpy(dcb, cb.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(de, edges.data(), E * 4, hipMemcpyHostToDevice);
    if (hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)) return -2;
    for (int v = 0; v < 3; ++v) {
        launch_v(v, 0);
        hipStreamSynchronize(sa);
        solo[v].resize(E);
        hipMemcpy(solo[v].data(), dcodes, E * 4, hipMemcpyDeviceToHost);
    }
    return 0;
}

extern "C" int victim_run(int variant, int reps, int lds_bytes, long long *worst_out)
{
    std::vector<unsigned> got(E);
    int bad = 0; long long worst = 0;
    for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(dcodes, 0, E * 4, sa);
        launch_v(variant, lds_bytes);
        hipStreamSynchronize(sa);
        hipMemcpy(got.data(), dcodes, E * 4, hipMemcpyDeviceToHost);
        long long nd = 0;
        for (long long e = 0; e < E; ++e) nd += got[e] != solo[variant][e];
        bad += nd != 0; worst = nd > worst ? nd : worst;
    }
    if (worst_out) *worst_out = worst;
    return bad;
}
