// kernels_dense.hip - node-level dense layers on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact
// f32, bit-equal to an fmaf chain) and the GraphNorm statistics.
//
// One generic tile kernel C[M,Nout] = pro(A)[M,K] * W[Nout,K]^T + bias with
//   prologue  0: A as is | 1: A = concat(A0, A1) along K | 2: A = SiLU(GraphNorm(A)) per trajectory
//   epilogue  0: store   | 1: C = R + acc + bias (residual) | 2: split columns into C (<256) and C2 (+bf16)
// covers single_embed, node_mlp.0, node_mlp.3 (+ the next layer's [Wa|Wb] projection) and the energy
// head's two projections (egnn.py:106-116, score_net_mlsb.py:366,:386-388).
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles),
// K staged 16 at a time through LDS as [k][row] so that fragment reads are conflict-free ds_read_b32.
#include "dfm_device.h"
#include "dfm_internal.h"

namespace dfm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 1;   // +1: spread the transposing LDS writes

__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a)
{
    __shared__ float As[BK * LDT];
    __shared__ float Ws[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging assignment: thread loads 2 x float4 of A and of W per K-chunk
    const int lr = tid >> 2;          // 0..63 (+64)
    const int lk = (tid & 3) * 4;     // 0,4,8,12
    const int halfK = a.K >> 1;

    for (int k0 = 0; k0 < a.K; k0 += BK) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = lr + rr * 64;
            const int grow = row0 + r;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (grow < a.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    if (k < a.K) {
                        float x;
                        if (a.pro == 1) {
                            x = (k < halfK) ? a.A0[(size_t)grow * a.lda + k] : a.A1[(size_t)grow * a.lda + (k - halfK)];
                        } else {
                            x = a.A0[(size_t)grow * a.lda + k];
                            if (a.pro == 2) {
                                // GraphNorm (torch_geometric 2.6.0, batch=None) + SiLU: egnn.py:72-76
                                const int g = grow / a.rows_per_graph;
                                const float o = x - a.gn_shift[(size_t)g * H + k];
                                x = silu_exact(a.gn_w[k] * o / a.gn_den[(size_t)g * H + k] + a.gn_b[k]);
                            }
                        }
                        v[e] = x;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(lk + e) * LDT + r] = v[e];
            // W tile
            const int gcol = col0 + r;
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (gcol < a.Nout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    if (k < a.K) w[e] = a.W[(size_t)gcol * a.ldw + k];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[(lk + e) * LDT + r] = w[e];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + (lane >> 5);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[k * LDT + wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Ws[k * LDT + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= a.Nout) continue;
            const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= a.M) continue;
                float v = acc[i][j][r] + bias;
                if (a.epi == 1) {
                    a.C[(size_t)row * a.ldc + col] = a.R[(size_t)row * a.ldc + col] + v;
                } else if (a.epi == 2) {
                    if (col < H) a.C[(size_t)row * H + col] = v;
                    else {
                        a.C2[(size_t)row * H + (col - H)] = v;
                        if (a.C2b) a.C2b[(size_t)row * H + (col - H)] = f2bf(v);
                    }
                } else {
                    a.C[(size_t)row * a.ldc + col] = v;
                }
            }
        }
}

hipError_t launch_gemm_f32(const GemmArgs &a, hipStream_t s)
{
    const dim3 grid((a.M + BM - 1) / BM, (a.Nout + BN - 1) / BN);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GraphNorm statistics per trajectory and channel (torch_geometric 2.6.0 graph_norm.py, batch=None):
//   mean = mean_n u ;  shift = mean * mean_scale ;  var = mean_n (u - shift)^2 ;  den = sqrt(var + 1e-5)
// Two exact passes in float64 (no E[x^2]-E[x]^2 cancellation), deterministic (no atomics).
// grid (B, 4): each workgroup owns 64 channels of one trajectory; 256 threads = 4 row lanes x 64 channels.
__global__ __launch_bounds__(256) void k_gn_stats(const float *__restrict__ u, int N, const float *__restrict__ mean_scale,
                                                  float *__restrict__ shift, float *__restrict__ den)
{
    __shared__ double red[4][64];
    __shared__ float sh_shift[64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const float *U = u + (size_t)b * N * H;
    double s = 0;
    for (int n = rl; n < N; n += 4) s += U[(size_t)n * H + c];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0) {
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const float mean = (float)(t / N);
        sh_shift[threadIdx.x] = mean * mean_scale[c];
    }
    __syncthreads();
    const float sft = sh_shift[threadIdx.x & 63];
    double v = 0;
    for (int n = rl; n < N; n += 4) {
        const float o = U[(size_t)n * H + c] - sft;
        v += (double)o * o;
    }
    __syncthreads();
    red[rl][threadIdx.x & 63] = v;
    __syncthreads();
    if (rl == 0) {
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const float var = (float)(t / N);
        shift[(size_t)b * H + c] = sft;
        den[(size_t)b * H + c] = sqrtf(var + 1e-5f);
    }
}

hipError_t launch_gn_stats(const float *u, int B, int N, const float *mean_scale, float *shift, float *den, hipStream_t s)
{
    hipLaunchKernelGGL(k_gn_stats, dim3(B, 4), dim3(256), 0, s, u, N, mean_scale, shift, den);
    return hipGetLastError();
}

}  // namespace dfm
