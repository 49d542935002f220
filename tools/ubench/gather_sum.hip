// Probe for DESIGN 14.2 (layer-0 table of intra-chain gated messages): how fast is the 60-row gather-sum it would need?
// table [N*N][256] fp16 (184 MB at N = 600), per (trajectory, node) 60 rows chosen like the engine's graph (20 nearest + 40 further
// ones of the same chain), summed in fp32 in slot order -> agg [B][N][256] fp32.  One wave per node: lane = 4 channels (8-byte loads,
// 512 B per row per wave), rows unrolled 4 deep.  Build: hipcc --offload-arch=gfx950 -O3 gather_sum.hip -o gather_sum
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k_gather_sum(const uint2 *__restrict__ table, const int *__restrict__ rows, float4 *__restrict__ agg, int nodes)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nodes) return;
    const int *r = rows + (size_t)w * 60;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < 60; s += 4) {
        uint2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = table[(size_t)r[s + q] * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const __half2 a = *reinterpret_cast<const __half2 *>(&v[q].x), b = *reinterpret_cast<const __half2 *>(&v[q].y);
            acc.x += __low2float(a); acc.y += __high2float(a); acc.z += __low2float(b); acc.w += __high2float(b);
        }
    }
    agg[(size_t)w * 64 + lane] = acc;
}
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 600, B = argc > 2 ? atoi(argv[2]) : 256, R = N / 2;
    const size_t trows = (size_t)N * N;
    uint2 *table; int *rows; float4 *agg;
    hipMalloc(&table, trows * 512); hipMalloc(&rows, (size_t)B * N * 60 * 4); hipMalloc(&agg, (size_t)B * N * 1024);
    hipMemset(table, 0, trows * 512);
    std::vector<int> h((size_t)B * N * 60);
    srand(1);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const int c0 = i < R ? 0 : R, cn = i < R ? R : N - R, il = i - c0;
            for (int s = 0; s < 60; ++s) {      // 20 sequence-near neighbours + 40 within +-100 of the same chain
                int j = s < 20 ? il + s - 10 : il + (rand() % 200) - 100;
                j = ((j % cn) + cn) % cn;
                h[((size_t)b * N + i) * 60 + s] = i * N + c0 + j;
            }
        }
    hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nodes = B * N;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather_sum, dim3((nodes + 3) / 4), dim3(256), 0, 0, table, rows, agg, nodes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("N=%d B=%d: gather-sum of %.2f GB from a %.0f MB table: %.3f ms = %.2f TB/s\n", N, B, (double)nodes * 60 * 512 / 1e9, trows * 512 / 1e6, ms,
               (double)nodes * 60 * 512 / 1e9 / ms);
    }
    return 0;
}
