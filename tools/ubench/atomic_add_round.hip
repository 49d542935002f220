// Does the L2 float atomic add round like v_add_f32?  z = x + y on the VALU vs atomicAdd(&z0 (= x), y), 1M random pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
__global__ void k(const float *x, const float *y, float *za, float *zv, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zv[i] = __fadd_rn(x[i], y[i]);
    za[i] = x[i];
    __threadfence();
    atomicAdd(za + i, y[i]);
}
int main()
{
    const int n = 1 << 20;
    float *hx = (float *)malloc(n * 4), *hy = (float *)malloc(n * 4), *ha = (float *)malloc(n * 4), *hv = (float *)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        hx[i] = (rand() / (float)RAND_MAX - 0.5f) * ldexpf(1.f, rand() % 28 - 22); hy[i] = (rand() / (float)RAND_MAX - 0.5f) * ldexpf(1.f, rand() % 28 - 22);
        if (i % 7 == 0) hy[i] = -hx[i] * (1.f + (rand() % 64) * 1e-7f);      // near cancellation
        if (i % 11 == 0) hx[i] = 0.f;
    }
    float *x, *y, *za, *zv;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&za, n * 4); hipMalloc(&zv, n * 4);
    hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(y, hy, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, y, za, zv, n);
    hipMemcpy(ha, za, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hv, zv, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, badhost = 0;
    for (int i = 0; i < n; ++i) { if (memcmp(ha + i, hv + i, 4)) ++bad; const float h = hx[i] + hy[i]; if (memcmp(&h, hv + i, 4)) ++badhost; }
    printf("atomicAdd vs v_add_f32: %d of %d differ; v_add_f32 vs host: %d differ\n", bad, n, badhost);
    return 0;
}
