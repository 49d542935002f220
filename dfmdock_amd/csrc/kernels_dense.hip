// kernels_dense.hip - node-level dense layers on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact
// f32, bit-equal to an fmaf chain) and the GraphNorm statistics.
//
// One generic tile kernel C[M,Nout] = pro(A)[M,K] * W[Nout,K]^T + bias with
//   prologue  0: A as is | 1: A = concat(A0, A1) along K | 2: A = SiLU(GraphNorm(A)) per trajectory
//   epilogue  0: store   | 1: C = R + acc + bias (residual) | 2: split columns into C (<256) and C2 (+fp16 copy)
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
                        if (a.C2b) a.C2b[(size_t)row * H + (col - H)] = f2h(v);
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
// Split-bf16 variant for the bf16 engine: x = hi + lo (two bf16 values, 16 mantissa bits), and
//   A W^T ~= Ahi Whi^T + Ahi Wlo^T + Alo Whi^T      (lo*lo dropped: ~2^-17 relative)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~1e-5 relative error at 3/16 of the f32-MFMA cost.
// Weights are pre-split on the host ([Nout][K] hi and lo); activations are split while they are staged.
// Same prologues / epilogues as k_gemm_f32, except that prologue 2 takes the folded GraphNorm affine
// (gn_den := w/den, gn_shift := b - w*shift/den per graph and channel, see k_gn_stats fold=1).  Needs K % 32 == 0.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union FragB { uint4 u; bf16x8 b; };
constexpr int SK = 32, SLD = 40;   // K per stage; LDS row stride in bf16 (80 B: conflict-free ds_read_b128)

struct GemmSplitArgs {
    GemmArgs g;
    const uint16_t *Whi, *Wlo;   // [Nout][ldw] bf16
};

__device__ inline uint32_t pack2(__bf16 a, __bf16 b)
{
    union { __bf16 h[2]; uint32_t u; } v;
    v.h[0] = a; v.h[1] = b;
    return v.u;
}

__global__ __launch_bounds__(256) void k_gemm_split(GemmSplitArgs sa)
{
    const GemmArgs &a = sa.g;
    __shared__ __attribute__((aligned(16))) uint16_t Ah[BM * SLD], Al[BM * SLD], Wh[BN * SLD], Wl[BN * SLD];
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

    const int lr = tid >> 1, lk = (tid & 1) * 16;    // staging: thread owns 16 consecutive k of one row
    const int halfK = a.K >> 1;
    const int grow = row0 + lr, gcol = col0 + lr;
    const int g = a.pro == 2 && grow < a.M ? grow / a.rows_per_graph : 0;

    for (int k0 = 0; k0 < a.K; k0 += SK) {
        float x[16];
        if (grow < a.M) {
            const int k = k0 + lk;
            const float *src = (a.pro == 1 && k >= halfK) ? a.A1 + (size_t)grow * a.lda + (k - halfK)
                                                          : a.A0 + (size_t)grow * a.lda + k;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(src + q * 4);
                x[q * 4] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
            }
            if (a.pro == 2) {   // GraphNorm + SiLU (egnn.py:72-76) as y = x * sc + sh with per-(graph, channel) sc, sh
                const float *sc = a.gn_den + (size_t)g * H + k, *sh = a.gn_shift + (size_t)g * H + k;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c4 = *reinterpret_cast<const float4 *>(sc + q * 4), h4 = *reinterpret_cast<const float4 *>(sh + q * 4);
                    const float y0 = fmaf(x[q * 4], c4.x, h4.x), y1 = fmaf(x[q * 4 + 1], c4.y, h4.y),
                                y2 = fmaf(x[q * 4 + 2], c4.z, h4.z), y3 = fmaf(x[q * 4 + 3], c4.w, h4.w);
                    x[q * 4] = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                    x[q * 4 + 1] = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                    x[q * 4 + 2] = y2 * __builtin_amdgcn_rcpf(1.0f + __expf(-y2));
                    x[q * 4 + 3] = y3 * __builtin_amdgcn_rcpf(1.0f + __expf(-y3));
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = 0.f;
        }
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 h0 = (__bf16)x[2 * e], h1 = (__bf16)x[2 * e + 1];
            hi[e] = pack2(h0, h1);
            lo[e] = pack2((__bf16)(x[2 * e] - (float)h0), (__bf16)(x[2 * e + 1] - (float)h1));
        }
        uint4 whi0 = make_uint4(0, 0, 0, 0), whi1 = whi0, wlo0 = whi0, wlo1 = whi0;
        if (gcol < a.Nout) {
            const uint16_t *ph = sa.Whi + (size_t)gcol * a.ldw + k0 + lk, *pl = sa.Wlo + (size_t)gcol * a.ldw + k0 + lk;
            whi0 = *reinterpret_cast<const uint4 *>(ph); whi1 = *reinterpret_cast<const uint4 *>(ph + 8);
            wlo0 = *reinterpret_cast<const uint4 *>(pl); wlo1 = *reinterpret_cast<const uint4 *>(pl + 8);
        }
        __syncthreads();   // previous stage fully consumed
        *reinterpret_cast<uint4 *>(&Ah[lr * SLD + lk]) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4 *>(&Ah[lr * SLD + lk + 8]) = make_uint4(hi[4], hi[5], hi[6], hi[7]);
        *reinterpret_cast<uint4 *>(&Al[lr * SLD + lk]) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4 *>(&Al[lr * SLD + lk + 8]) = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        *reinterpret_cast<uint4 *>(&Wh[lr * SLD + lk]) = whi0;
        *reinterpret_cast<uint4 *>(&Wh[lr * SLD + lk + 8]) = whi1;
        *reinterpret_cast<uint4 *>(&Wl[lr * SLD + lk]) = wlo0;
        *reinterpret_cast<uint4 *>(&Wl[lr * SLD + lk + 8]) = wlo1;
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < SK; ks += 16) {
            const int ko = ks + (lane >> 5) * 8;
            FragB ah[2], al[2], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + (lane & 31);
                ah[i].u = *reinterpret_cast<const uint4 *>(&Ah[r * SLD + ko]);
                al[i].u = *reinterpret_cast<const uint4 *>(&Al[r * SLD + ko]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + j * 32 + (lane & 31);
                wh[j].u = *reinterpret_cast<const uint4 *>(&Wh[c * SLD + ko]);
                wl[j].u = *reinterpret_cast<const uint4 *>(&Wl[c * SLD + ko]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i].b, wh[j].b, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].b, wl[j].b, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].b, wh[j].b, acc[i][j], 0, 0, 0);
                }
        }
    }
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
                const float v = acc[i][j][r] + bias;
                if (a.epi == 1) {
                    a.C[(size_t)row * a.ldc + col] = a.R[(size_t)row * a.ldc + col] + v;
                } else if (a.epi == 2) {
                    if (col < H) a.C[(size_t)row * H + col] = v;
                    else {
                        a.C2[(size_t)row * H + (col - H)] = v;
                        if (a.C2b) a.C2b[(size_t)row * H + (col - H)] = f2h(v);
                    }
                } else {
                    a.C[(size_t)row * a.ldc + col] = v;
                }
            }
        }
}

hipError_t launch_gemm_split(const GemmArgs &a, const uint16_t *Whi, const uint16_t *Wlo, hipStream_t s)
{
    if (a.K % SK != 0 || (a.pro == 1 && (a.K / 2) % 16 != 0)) return hipErrorInvalidValue;
    GemmSplitArgs sa;
    sa.g = a; sa.Whi = Whi; sa.Wlo = Wlo;
    const dim3 grid((a.M + BM - 1) / BM, (a.Nout + BN - 1) / BN);
    hipLaunchKernelGGL(k_gemm_split, grid, dim3(256), 0, s, sa);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GraphNorm statistics per trajectory and channel (torch_geometric 2.6.0 graph_norm.py, batch=None):
//   mean = mean_n u ;  shift = mean * mean_scale ;  var = mean_n (u - shift)^2 ;  den = sqrt(var + 1e-5)
// Two exact passes in float64 (no E[x^2]-E[x]^2 cancellation), deterministic (no atomics).
// grid (B, 4): each workgroup owns 64 channels of one trajectory; 256 threads = 4 row lanes x 64 channels.
__global__ __launch_bounds__(256) void k_gn_stats(const float *__restrict__ u, int N, const float *__restrict__ mean_scale,
                                                  float *__restrict__ shift, float *__restrict__ den,
                                                  const float *__restrict__ fold_w, const float *__restrict__ fold_b)
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
        const float dn = sqrtf(var + 1e-5f);
        if (fold_w) {   // y = w*(x - shift)/den + b  ==  x*sc + sh
            const float sc = fold_w[c] / dn;
            den[(size_t)b * H + c] = sc;
            shift[(size_t)b * H + c] = fold_b[c] - sc * sft;
        } else {
            shift[(size_t)b * H + c] = sft;
            den[(size_t)b * H + c] = dn;
        }
    }
}

hipError_t launch_gn_stats(const float *u, int B, int N, const float *mean_scale, float *shift, float *den,
                           const float *fold_w, const float *fold_b, hipStream_t s)
{
    hipLaunchKernelGGL(k_gn_stats, dim3(B, 4), dim3(256), 0, s, u, N, mean_scale, shift, den, fold_w, fold_b);
    return hipGetLastError();
}

}  // namespace dfm
