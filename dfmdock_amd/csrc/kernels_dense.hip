// kernels_dense.hip - node-level dense layers on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact
// f32, bit-equal to an fmaf chain) and the GraphNorm statistics.
//
// One generic tile kernel C[M,Nout] = pro(A)[M,K] * W[Nout,K]^T + bias with
//   prologue  0: A as is | 1: A = concat(A0, A1) along K | 2: A = SiLU(GraphNorm(A)) per trajectory | 3: A = SiLU(A)
//   epilogue  0: store   | 1: C = R + acc + bias (residual) | 2: split columns into C (<256) and C2 (+fp16 copy)
// covers single_embed, node_mlp.0, node_mlp.3 (+ the next layer's [Wa|Wb] projection) and the energy
// head's two projections (egnn.py:106-116, score_net_mlsb.py:366,:386-388).
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles),
// K staged 16 at a time through LDS as [k][row] so that fragment reads are conflict-free ds_read_b32.
#include <cstdlib>
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
            const int grow_a = a.a0_period ? grow % a.a0_period : grow;      // layer 0: the complex's own h0 rows (GemmArgs::a0_period)
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (grow < a.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    if (k < a.K) {
                        float x;
                        if (a.pro == 1) {
                            x = (k < halfK) ? a.A0[(size_t)grow_a * a.lda + k] : a.A1[(size_t)grow * a.lda + (k - halfK)];
                        } else {
                            x = a.A0[(size_t)grow_a * a.lda + k];
                            if (a.pro == 2) {
                                // GraphNorm (torch_geometric 2.6.0, batch=None) + SiLU: egnn.py:72-76
                                const int g = grow / a.rows_per_graph;
                                const float o = x - a.gn_shift[(size_t)g * H + k];
                                x = silu_exact(a.gn_w[k] * o / a.gn_den[(size_t)g * H + k] + a.gn_b[k]);
                            } else if (a.pro == 3) {
                                x = silu_exact(x);     // Sequential(..., SiLU, Linear): the activation rides on the next Linear's load
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
                    a.C[(size_t)row * a.ldc + col] = a.R[(size_t)(a.r_period ? row % a.r_period : row) * a.ldc + col] + v;
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

// The same GEMM for the shapes the engine actually launches (K a multiple of 16, 16-byte aligned rows): 16-byte loads, the next K-chunk's
// loads in registers under this chunk's MFMAs, double-buffered LDS tiles (one barrier per chunk).  Element for element the arithmetic of
// k_gemm_f32 - same prologue expressions, same MFMA order - so the results are bitwise the same; r01-r04's kernel staged synchronously with
// scalar loads and ran at a quarter of the fp32 matrix peak (0.79 ms per launch, 14.6 % of the fp32 engine's GPU time at C3).
__global__ __launch_bounds__(256) void k_gemm_f32v(GemmArgs a)
{
    __shared__ float As[2][BK * LDT];
    __shared__ float Ws[2][BK * LDT];
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
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const int halfK = a.K >> 1;
    // per-thread rows (two of the 128) and columns of the staging assignment
    int grow[2], gcol[2];
    size_t arow0[2], arow1[2];
    bool rv[2], cv[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        grow[rr] = row0 + lr + rr * 64; gcol[rr] = col0 + lr + rr * 64;
        rv[rr] = grow[rr] < a.M; cv[rr] = gcol[rr] < a.Nout;
        const int g = rv[rr] ? grow[rr] : 0;
        arow0[rr] = (size_t)(a.a0_period ? g % a.a0_period : g) * a.lda;
        arow1[rr] = (size_t)g * a.lda;
    }
    float4 ra[2], rw[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int k = k0 + lk;
            const float *src = (a.pro == 1 && k >= halfK) ? a.A1 + arow1[rr] + (k - halfK) : a.A0 + arow0[rr] + k;
            ra[rr] = *reinterpret_cast<const float4 *>(src);
            rw[rr] = *reinterpret_cast<const float4 *>(a.W + (size_t)(cv[rr] ? gcol[rr] : 0) * a.ldw + k);
        }
    };
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = lr + rr * 64;
            float v[4] = {ra[rr].x, ra[rr].y, ra[rr].z, ra[rr].w};
            if (a.pro == 2) {
                const int g = (rv[rr] ? grow[rr] : 0) / a.rows_per_graph;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    // GraphNorm (torch_geometric 2.6.0, batch=None) + SiLU: egnn.py:72-76
                    const float o = v[e] - a.gn_shift[(size_t)g * H + k];
                    v[e] = silu_exact(a.gn_w[k] * o / a.gn_den[(size_t)g * H + k] + a.gn_b[k]);
                }
            } else if (a.pro == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_exact(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) As[buf][(lk + e) * LDT + r] = rv[rr] ? v[e] : 0.f;
            const float w[4] = {rw[rr].x, rw[rr].y, rw[rr].z, rw[rr].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[buf][(lk + e) * LDT + r] = cv[rr] ? w[e] : 0.f;
        }
    };
    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < a.K; k0 += BK, buf ^= 1) {
        stage(k0, buf);
        __syncthreads();      // (the other buffer's readers are past it: they met this barrier after their MFMAs of the previous chunk)
        if (k0 + BK < a.K) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + (lane >> 5);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[buf][k * LDT + wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Ws[buf][k * LDT + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
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
                float v = acc[i][j][r] + bias;
                if (a.epi == 1) {
                    a.C[(size_t)row * a.ldc + col] = a.R[(size_t)(a.r_period ? row % a.r_period : row) * a.ldc + col] + v;
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
    static const bool scalar_env = [] { const char *e = getenv("DFM_GEMM_F32_SCALAR"); return e && atoi(e) != 0; }();      // A/B: the r01-r04 kernel
    const bool aligned = a.K % BK == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && (a.pro != 1 || (a.K / 2) % 4 == 0) &&
                         ((uintptr_t)a.A0 % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && (a.pro != 1 || (uintptr_t)a.A1 % 16 == 0);
    if (aligned && !scalar_env) hipLaunchKernelGGL(k_gemm_f32v, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant for the bf16 engine: x = hi + lo (two bf16 values, 16 mantissa bits), and
//   A W^T ~= Ahi Whi^T + Ahi Wlo^T + Alo Whi^T      (lo*lo dropped: ~2^-17 relative)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~1e-5 relative error at 3/16 of the f32-MFMA cost.
//
// The node-level layers are tall and skinny (M = B*N rows, K and Nout in {256, 512}) and HBM-bound, so one
// workgroup owns a block of rows x 256 output columns: every activation is read, normalised/activated and split
// exactly once per 256 outputs.  Weights are pre-split and pre-tiled on the host in stage order ([K/32][4 k-groups][Nout][8]
// bf16, hi and lo): coalesced 16-byte loads, L2-resident, and consecutive lanes write consecutive LDS rows.  Per K-stage of 32: the
// next stage's global loads are issued into registers, then the MFMAs run from the LDS tiles (80-byte row stride:
// conflict-free ds_read_b128).
// Same prologues / epilogues as k_gemm_f32, except that prologue 2 takes the folded GraphNorm affine
// (gn_den := w/den, gn_shift := b - w*shift/den per graph and channel, see k_gn_stats fold=1).
// Needs K % 32 == 0, Nout % 256 == 0.
// One channel of GraphNorm from the column statistics of k_gemm_split (stat_part [B][ceil(N / 32)][256][2] = (mean, M2 = sum of squared
// deviations) over each 32-row half of a tile - the unit every tile shape of the kernel shares, r04): halves merged in a fixed order with Chan's update in float64, then
// var = E[(u - shift)^2] = M2 / N + (mean - shift)^2 with shift = mean * mean_scale.  fold_w: returns the folded affine
// (den := w / den, shift := b - w * shift / den).  Runs in the prologue of the GEMM that consumes the normalised activations
// (r01-r03: a separate k_gn_finish launch per layer, 5 us each that small batches could not hide).
constexpr int GN_CH = 20;      // partials in flight per round trip (300+300: 19 partials = one round)
__device__ inline void gn_finish_col(const float *__restrict__ part, int b, int N, int c, const float *__restrict__ mean_scale,
                                     const float *__restrict__ fold_w, const float *__restrict__ fold_b, float &o_den, float &o_shift)
{
    const int tpt = (N + 31) / 32;
    // This loop is the exposed prologue of every workgroup of the consuming GEMM: the partials come GN_CH loads at a time (one L2 round
    // trip per GN_CH, not per partial: 164 -> 153 us per launch at C3), and the merge is the division-free pooled form in float64 - S1 = sum n_t mean_t,
    // S2 = sum n_t mean_t^2: mean = S1 / N, M2 = sum M2_t + (S2 - mean S1) - whose cancellation float64 absorbs (the partial means are
    // fp32 values).  Fixed order: the same bits for every trajectory and batch size.
    const float *base = part + (size_t)b * tpt * (H * 2) + c * 2;
    double S1 = 0, S2 = 0, SM = 0;
    for (int t0 = 0; t0 < tpt; t0 += GN_CH) {
        float2 v[GN_CH];
#pragma unroll
        for (int u = 0; u < GN_CH; ++u) {
            const int t = t0 + u < tpt ? t0 + u : tpt - 1;
            v[u] = *reinterpret_cast<const float2 *>(base + (size_t)t * (H * 2));
        }
#pragma unroll
        for (int u = 0; u < GN_CH; ++u) {
            const int t = t0 + u;
            if (t < tpt) {
                const double nt = (double)(N - t * 32 < 32 ? N - t * 32 : 32), m = (double)v[u].x, w = nt * m;
                S1 += w; S2 += w * m; SM += (double)v[u].y;
            }
        }
    }
    const double mean = S1 / N;
    const double M2 = SM + (S2 - mean * S1);
    const float sft = (float)mean * mean_scale[c];
    const double dm = mean - (double)sft;
    double var = M2 / N + dm * dm;
    var = var > 0 ? var : 0;
    const float dn = sqrtf((float)var + 1e-5f);
    if (fold_w) {
        const float sc = fold_w[c] / dn;
        o_den = sc;
        o_shift = fold_b[c] - sc * sft;
    } else {
        o_shift = sft;
        o_den = dn;
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union FragB { uint4 u; bf16x8 b; };
constexpr int SK = 32, SLD = 40;   // K per stage; LDS row stride in bf16

struct GemmSplitArgs {
    GemmArgs g;
    const uint16_t *Whi, *Wlo;   // [K/32][4][Nout][8] bf16 hi / lo (split_bf16 in api.hip)
};

__device__ inline uint32_t pack2(__bf16 a, __bf16 b)
{
    union { __bf16 h[2]; uint32_t u; } v;
    v.h[0] = a; v.h[1] = b;
    return v.u;
}

// 64 x 256 tile per workgroup, waves 1 x 4 of 64 x 64 (50 KiB LDS, 3 workgroups / CU: more independent phases in flight than the
// 128-row tile of r01).  Variants measured this round and dropped (same-box A/B, profiles/r03_exp_gemm_variants.txt): weight fragments
// straight from L2 into registers + double-buffered activation stages (200 registers, 2 workgroups / CU: 175 vs 162 us - the
// vector-memory path, not LDS, carries the 32 KiB of weights per stage either way, and occupancy matters more); that kernel made
// persistent (181 vs 186 us); two terms a_hi x (w_hi + w_lo) on fp16 (no faster: not MFMA-bound; 8.2e-3 on f of the 3x draw).
// What did pay: the fp16 outputs of the [Wa|Wb] projection converted by the hardware (f2h is 30 instructions of bit manipulation
// per value: half of that launch) and stored as 16-byte vectors: 200 -> 153 us for that launch, 166 -> 149 us over the three.
// HALF = 1: epilogue 2 with 16-bit outputs only (Cb and C2b set, no fp32 C2): the [Wa|Wb] projection of the 16-bit engine
// NJ = 1: 64 x 128 tiles (each wave 64 x 32) for launches with fewer workgroups than CUs: a lone workgroup is bound by what ONE CU
// can pull from L2 / HBM (~55 GB/s: 40 KiB per K-stage = 0.75 us; deeper prefetch or a pipelined loop buy nothing,
// profiles/r02_exp_gemm_deep.txt, r03_d notes) - twice the workgroups, each with half the weight bytes per stage.  Same MFMA sequence per
// output element and the same row order in the column statistics: bitwise the same results as NJ = 2.
// QT = 1 (with NJ = 1): 64 x 64 tiles, waves 2 x 2 of 32 x 32, for launches that do not even give every second CU a 64 x 128 workgroup
// (B <= 6 at 300+300): a lone workgroup is a serial chain of K-stages whose length follows the bytes and MFMAs of ONE stage (r03 stamps:
// 970 cycles of MFMA phase + 830 of fetch per stage at 64 x 128), so four times the workgroups with a third of each: node GEMMs
// 17.9 / 8.7 -> XX / XX us at B = 1 (profiles/r04_small_batch.txt).  Bitwise the same results again.
// RING (= QT): the K loop keeps four stages of loads in flight and double-buffers the operand tiles (see below).  The same loop on the
// 64 x 128 shape (RING = 1, QT = 0: 176 registers, 62 KiB of LDS) was measured at B = 8 and bought nothing - there the statistics epilogue
// and the GraphNorm + SiLU staging chains of a lone wave are what a launch waits for (profiles/r04_small_gemm_ring.txt) - and is not instantiated.
template <int HALF, int NJ, int QT = 0, int RING = QT> __global__ __launch_bounds__(256, RING ? 1 : 3) void k_gemm_split(GemmSplitArgs sa)
{
    static_assert(!QT || NJ == 1, "quarter tiles: one 32-column tile per wave");
    static_assert(!RING || NJ == 1, "the register ring is sized for the 64 x 128 and 64 x 64 shapes");
    static_assert(!QT || RING, "quarter tiles always run the ring");
    constexpr int MI = QT ? 1 : 2;                       // 32-row tiles per wave
    constexpr int SM = 64, SN = QT ? 64 : NJ * 128;      // rows; output columns per workgroup (QT 0: four waves along N, NJ 32-column tiles each)
    const GemmArgs &a = sa.g;
#ifdef DFM_GEMM_STAMP
    const unsigned long long g_entry = __builtin_amdgcn_s_memtime(), g_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // operand tiles; the epilogue reuses the space as 4 x 9216 B of transposition buffers
    constexpr int LDS_BUF = (2 * SM + 2 * SN) * SLD;      // one set of operand tiles (hi / lo activations, hi / lo weights)
    constexpr int LDS_OPER = (RING ? 2 : 1) * LDS_BUF, LDS_EPI = 4 * 32 * 72 * 2;      // operand tiles | epilogue staging (4 waves x 32 x ELD floats)
    constexpr int LDS_U16 = LDS_OPER > LDS_EPI ? LDS_OPER : LDS_EPI;
    __shared__ __attribute__((aligned(16))) uint16_t lds[LDS_U16];
    uint16_t *Ah = lds, *Al = lds + SM * SLD, *Wh = lds + 2 * SM * SLD, *Wl = lds + (2 * SM + SN) * SLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = QT ? wave >> 1 : 0, wn = QT ? wave & 1 : wave, l31 = lane & 31;      // wave's 32-row tile(s) start at wm * MI * 32
    // row tile: plain 64/128-row blocks of the [M] rows, or - when the GraphNorm column sums are wanted - blocks aligned to
    // the trajectory (rows_per_graph rows each, last block partial): every block then belongs to one trajectory and the
    // summation order is the same for every trajectory, whatever its position in the batch (batched == single, bitwise)
    // block -> (row tile, 256-column block).  With two column blocks (the [Wa|Wb] projection, Nout = 512) the launch is 1-D and the
    // two blocks of a row tile are 8 apart: workgroup g runs on XCD g % 8, so they share an L2 and run close in time - the second
    // read of the 64 x K activation tile hits L2 instead of HBM
    // (A persistent grid - three workgroups per CU walking the tiles, so that a tile's store burst drains under the next tile's K
    // loop - costs 68 spilled registers at this kernel's 168-register budget: 4x slower.)
    __shared__ __attribute__((aligned(16))) float gn_s[2 * H];
    const int vb = blockIdx.x;
    int bx = vb, by = blockIdx.y;
    if (NJ == 2 && gridDim.y == 1 && a.Nout == 2 * SN) {
        const int g16 = bx >> 4, j = bx & 15;
        bx = g16 * 8 + (j & 7); by = j >> 3;
        if (bx * SM >= a.M) return;          // tail of the last group of 8 tiles
    }
    int row0 = bx * SM, row_end = a.M, gn_tb = 0;
    const int col0 = by * SN;
    // GraphNorm prologue: the folded scale / shift of the tile's trajectory, 2 KiB in LDS for the whole K loop (a load
    // per K-stage inside the staging code would expose an L2 round trip per stage)
    if (a.stat_part || a.pro == 2) {
        const int tpt = (a.rows_per_graph + SM - 1) / SM, tb = bx / tpt;
        row0 = tb * a.rows_per_graph + (bx - tb * tpt) * SM;
        row_end = (tb + 1) * a.rows_per_graph;
        gn_tb = tb;
    }
    if (a.zbuf) {      // zero this tile's block of a [M][256] buffer nobody reads any more (agg: see GemmArgs::zbuf)
        const int zr = tid >> 2;
        if (row0 + zr < row_end) {
            float4 *z = reinterpret_cast<float4 *>(a.zbuf + (size_t)(row0 + zr) * H + col0 + (tid & 3) * (SN / 4));
#pragma unroll
            for (int q = 0; q < SN / 16; ++q) z[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // The residual tile, touched now (one dword per 128-byte line): the epilogue's loads of it then hit L2 instead of paying a round trip to
    // HBM at the end of the workgroup, where nothing is left to hide it.  Ordinary loads (the compiler's vmcnt bookkeeping covers them),
    // kept alive without arithmetic by an empty asm before the epilogue.
    // Quarter tiles only: at the 64 x 128 shape (B = 8) the epilogue got slower with it, and the 64 x 256 launches lose 2.5 %.
    constexpr int LPR = SN * 4 / 128, NTOUCH = QT ? (SM * LPR + 255) / 256 : 0;      // lines per tile row; lines per thread
    float touch[NTOUCH + 1];
#pragma unroll
    for (int q = 0; q < NTOUCH; ++q) {
        touch[q] = 0.f;
        const int t = tid + q * 256, row = row0 + t / LPR;
        if (a.epi == 1 && t < SM * LPR && row < row_end) {
            const int rrow = a.r_period ? row % a.r_period : row;
            touch[q] = a.R[(size_t)rrow * a.ldc + col0 + (t % LPR) * 32];
        }
    }
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread owns 8 consecutive k (kg) of row ar; four lanes cover one row's 128-byte line
    const int ar = tid >> 2, kg = (tid & 3) * 8;
    const int halfK = a.K >> 1;
    const bool rv0 = row0 + ar < row_end;
    const size_t gr0 = (size_t)(rv0 ? row0 + ar : 0);
    const size_t gr0a = a.a0_period ? gr0 % (size_t)a.a0_period : gr0;      // row of A0 (layer 0: the complex's own h0, GemmArgs::a0_period)

    // registers of the stage being fetched: activations (rows x 8 k) and 4 x 16 B of hi / lo weights
    float4 xa0, xa1;
    uint4 wh0, wh1, wh2, wh3, wl0, wl1, wl2, wl3;      // NJ = 1: thread = (column tid & 127, hi / lo tid >> 7), wh0..3 only;
                                                       // QT: thread = (column tid & 63, hi / lo (tid >> 6) & 1, k-groups 2 (tid >> 7) + {0, 1}), wh0..1 only
    const int wcol = QT ? (tid & 63) : (NJ == 2 ? tid : (tid & 127));
    const int wkq = QT ? (tid >> 7) * 2 : 0;
    const uint16_t *wsrc = QT ? (((tid >> 6) & 1) ? sa.Wlo : sa.Whi) : ((NJ == 2 || tid < 128) ? sa.Whi : sa.Wlo);
#define GEMM_SPLIT_FETCH(K0)                                                                                          \
    {                                                                                                                 \
        const int k_ = (K0) + kg;                                                                                     \
        const bool second_ = a.pro == 1 && k_ >= halfK;                                                               \
        const float *s0_ = second_ ? a.A1 + (k_ - halfK) + gr0 * a.lda : a.A0 + k_ + gr0a * a.lda;                    \
        xa0 = *reinterpret_cast<const float4 *>(s0_); xa1 = *reinterpret_cast<const float4 *>(s0_ + 4);              \
        const size_t wbase_ = ((size_t)((K0) / SK) * 4 * a.Nout + col0 + wcol) * 8;                                   \
        const size_t wq_ = (size_t)a.Nout * 8;                                                                        \
        const uint16_t *ph_ = wsrc + wbase_ + wkq * wq_, *pl_ = sa.Wlo + wbase_;                                      \
        wh0 = *reinterpret_cast<const uint4 *>(ph_); wh1 = *reinterpret_cast<const uint4 *>(ph_ + wq_);               \
        if constexpr (!QT) {                                                                                          \
            wh2 = *reinterpret_cast<const uint4 *>(ph_ + 2 * wq_); wh3 = *reinterpret_cast<const uint4 *>(ph_ + 3 * wq_); \
        }                                                                                                             \
        if constexpr (NJ == 2) {                                                                                      \
            wl0 = *reinterpret_cast<const uint4 *>(pl_); wl1 = *reinterpret_cast<const uint4 *>(pl_ + wq_);               \
            wl2 = *reinterpret_cast<const uint4 *>(pl_ + 2 * wq_); wl3 = *reinterpret_cast<const uint4 *>(pl_ + 3 * wq_); \
        }                                                                                                             \
    }
    auto stage_row = [&](const float4 &v0, const float4 &v1, bool /*valid*/, int k, int row, int boff = 0) {
        float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (a.pro == 2) {   // GraphNorm + SiLU (egnn.py:72-76) as y = x * sc + sh with per-(graph, channel) sc, sh
            // the tile's trajectory: scale / shift sit in LDS since the start of the kernel
            const float4 c0 = *reinterpret_cast<const float4 *>(&gn_s[k]), c1 = *reinterpret_cast<const float4 *>(&gn_s[k + 4]);
            const float4 h0 = *reinterpret_cast<const float4 *>(&gn_s[H + k]), h1 = *reinterpret_cast<const float4 *>(&gn_s[H + k + 4]);
            const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = fmaf(x[e], cc[e], hh[e]);
                x[e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            }
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (rows past the tile's last valid row hold a re-read of its first row: rows of a product are independent and the epilogue
            // neither stores nor counts them, so they are not zeroed - eight selects per thread and stage less)
            const float x0 = x[2 * e], x1 = x[2 * e + 1];
            const __bf16 b0 = (__bf16)x0, b1 = (__bf16)x1;
            hi[e] = pack2(b0, b1);
            lo[e] = pack2((__bf16)(x0 - (float)b0), (__bf16)(x1 - (float)b1));
        }
        *reinterpret_cast<uint4 *>(&Ah[boff + row * SLD + kg]) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4 *>(&Al[boff + row * SLD + kg]) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    };

    const int wu = wcol * SLD + wkq * 8;   // weights: thread = output column, q = 8-k group (consecutive lanes -> consecutive LDS rows: conflict-free)
    uint16_t *wdst = QT ? (((tid >> 6) & 1) ? Wl : Wh) : ((NJ == 2 || tid < 128) ? Wh : Wl);
#ifdef DFM_GEMM_STAMP
    unsigned long long gs[4] = {0, 0, 0, 0}, gprev = __builtin_amdgcn_s_memtime();
    const unsigned long long g_first = gprev;
#define GSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _n = __builtin_amdgcn_s_memtime(); gs[k] += _n - gprev; gprev = _n; __builtin_amdgcn_sched_barrier(0); }
#else
#define GSTAMP(k)
#endif
    // the epilogue's bias values, requested now: as a load in the epilogue they are a dependent L2 round trip at the head of every
    // tile's store phase (small launches have no other workgroup on the CU to hide it)
    float bias_r[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bias_r[j] = a.bias ? a.bias[col0 + (wn * NJ + j) * 32 + l31] : 0.f;
    // Quarter tiles run alone on their CU (launches of fewer workgroups than half the CUs): a K-stage there is ONE round trip to L2 / HBM
    // (~2 000 cycles, s_memtime stamps at B = 1: profiles/r04_small_gemm_ring.txt) however little it carries, so QD stages are kept in
    // flight in a static register ring (16 registers per stage) instead of one
    constexpr int QD = RING ? 4 : 1, RW = QT ? 2 : 4;      // stages in flight; 16-byte weight registers of a stage
    float4 rxa0[QD], rxa1[QD];
    uint4 rw[QD][RW];
    GEMM_SPLIT_FETCH(0)
#define RING_TAKE(d) { rxa0[d] = xa0; rxa1[d] = xa1; rw[d][0] = wh0; rw[d][1] = wh1; if constexpr (!QT) { rw[d][2] = wh2; rw[d][3] = wh3; } }
#define RING_WSTORE(d, nb) { _Pragma("unroll") for (int q_ = 0; q_ < RW; ++q_) *reinterpret_cast<uint4 *>(&wdst[(nb) + wu + 8 * q_]) = rw[d][q_]; }
    if constexpr (RING) {
        RING_TAKE(0)
#pragma unroll
        for (int d = 1; d < QD; ++d) {
            GEMM_SPLIT_FETCH((d * SK < a.K ? d : 0) * SK)
            RING_TAKE(d)
        }
    }
    // GraphNorm prologue, behind the first stage's loads (its L2 round trip rides under theirs instead of preceding it)
    if (a.pro == 2) {
        if (a.gn_part) {      // finish the statistics here (thread = channel): no separate launch between the two GEMMs
            float sc, sh;
            gn_finish_col(a.gn_part, gn_tb, a.rows_per_graph, tid, a.gn_ms, a.gn_w, a.gn_b, sc, sh);
            gn_s[tid] = sc; gn_s[H + tid] = sh;
        } else if (tid < 128) {
            const float *src = (tid < 64 ? a.gn_den : a.gn_shift) + (size_t)gn_tb * H + (tid & 63) * 4;
            *reinterpret_cast<float4 *>(&gn_s[(tid < 64 ? 0 : H) + (tid & 63) * 4]) = *reinterpret_cast<const float4 *>(src);
        }
        __syncthreads();
    }
    auto mfma_stage = [&](int boff = 0) {
#pragma unroll
        for (int ks = 0; ks < SK; ks += 16) {
            const int ko = ks + (lane >> 5) * 8;
            FragB ah[MI], al[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = (wm * MI + i) * 32 + l31;
                ah[i].u = *reinterpret_cast<const uint4 *>(&Ah[boff + r * SLD + ko]);
                al[i].u = *reinterpret_cast<const uint4 *>(&Al[boff + r * SLD + ko]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = (wn * NJ + j) * 32 + l31;
                FragB wh, wl;
                wh.u = *reinterpret_cast<const uint4 *>(&Wh[boff + c * SLD + ko]);
                wl.u = *reinterpret_cast<const uint4 *>(&Wl[boff + c * SLD + ko]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i].b, wh.b, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].b, wl.b, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].b, wh.b, acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (RING) {
        // ... and the operand tiles are double-buffered: stage k + 1 is converted and written (VALU, LDS stores) behind the MFMAs of
        // stage k in the same instruction stream, one barrier per stage - with a single wave per SIMD nothing else would fill the
        // matrix pipe's latency (stamps: 800 cycles of staging + 700 of MFMA phase per stage in sequence before)
        // The loop body is branch-free (the launcher takes this shape only when K / 32 is a multiple of QD; stages past the end are
        // clamped re-reads of the last one, staged into the idle buffer and never consumed): with a branch around the refill the
        // compiler's s_waitcnt insertion falls back to vmcnt(0) - it waits for the loads just issued and the ring is worth nothing.
        const int S = a.K / SK;
        stage_row(rxa0[0], rxa1[0], rv0, kg, ar, 0);
        RING_WSTORE(0, 0)
        {
            GEMM_SPLIT_FETCH((QD < S ? QD : S - 1) * SK)
            RING_TAKE(0)
        }
        __syncthreads();
        GSTAMP(1)
        for (int k0 = 0; k0 < S; k0 += QD) {
#pragma unroll
            for (int d = 0; d < QD; ++d) {
                const int k = k0 + d;
                const int dn = (d + 1) % QD, nb = ((d + 1) & 1) * LDS_BUF;
                const int kn = k + 1 < S ? k + 1 : S - 1, kf = k + 1 + QD < S ? k + 1 + QD : S - 1;
                mfma_stage((d & 1) * LDS_BUF);
                GSTAMP(0)
                stage_row(rxa0[dn], rxa1[dn], rv0, kn * SK + kg, ar, nb);
                RING_WSTORE(dn, nb)
                {      // this slot's next stage: QD - 1 others are already on their way
                    GEMM_SPLIT_FETCH(kf * SK)
                    RING_TAKE(dn)
                }
                GSTAMP(1)
                __syncthreads();
                GSTAMP(2)
            }
        }
    } else {
        for (int k0 = 0; k0 < a.K; k0 += SK) {
            if (k0) __syncthreads();   // previous stage fully consumed
            GSTAMP(0)                  // [0] MFMA phase + barrier wait
            stage_row(xa0, xa1, rv0, k0 + kg, ar);
            *reinterpret_cast<uint4 *>(&wdst[wu]) = wh0; *reinterpret_cast<uint4 *>(&wdst[wu + 8]) = wh1;
            if constexpr (!QT) { *reinterpret_cast<uint4 *>(&wdst[wu + 16]) = wh2; *reinterpret_cast<uint4 *>(&wdst[wu + 24]) = wh3; }
            if constexpr (NJ == 2) {
                *reinterpret_cast<uint4 *>(&Wl[wu]) = wl0; *reinterpret_cast<uint4 *>(&Wl[wu + 8]) = wl1;
                *reinterpret_cast<uint4 *>(&Wl[wu + 16]) = wl2; *reinterpret_cast<uint4 *>(&Wl[wu + 24]) = wl3;
            }
            GSTAMP(1)                  // [1] waiting for the fetched registers + conversion + LDS stores
            __syncthreads();
            GSTAMP(2)                  // [2] barrier after staging
            if (k0 + SK < a.K) GEMM_SPLIT_FETCH(k0 + SK)         // flies under the MFMAs below
            mfma_stage();
        }
    }
#pragma unroll
    for (int q = 0; q < NTOUCH; ++q) asm volatile("" :: "v"(touch[q]));
    GSTAMP(0)
    // epilogue: the C layout (lane = column, registers = rows) would need scalar stores, and the kernel is then bound
    // by store issue.  Each wave instead transposes its outputs through a private LDS region (the operand tiles are
    // dead by now) in 32 x 64 passes and stores 16-byte vectors, 256 B per row.
    __syncthreads();
    // GraphNorm statistics of this lane's rows (4 columns): shifted sums about the lane's first value - no E[u^2] - E[u]^2
    // cancellation when |mean| >> std - turned into (count, mean, M2) and merged Chan-style across lanes, tiles (gn_finish_col)
    // One set per 32-row half of the tile (r01-r03: per 64-row tile): the unit that every tile shape shares, so that the choice of shape -
    // a function of the launch size - never shows in a bit.
    constexpr int ELD = 72;                               // floats per staged row (64 + 8: the half-waves, 4 rows apart, hit disjoint banks)
    float *est = reinterpret_cast<float *>(lds) + wave * (32 * ELD);   // 9216 B per wave
    const int er = lane >> 4, ec = (lane & 15) * 4;       // read-back: 4 rows x 16 float4 per instruction
    float st_s[MI][4], st_q[MI][4], st_p[MI][4], st_n[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        st_n[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_s[i][e] = 0.f; st_q[i][e] = 0.f; st_p[i][e] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jp = 0; jp < (NJ + 1) / 2; ++jp) {
            // residual rows of this pass, requested before the transposition below instead of one dependent load per store
            float4 res[8];
            if (a.epi == 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t row = (size_t)row0 + (wm * MI + i) * 32 + q * 4 + er;
                    const size_t rrow = a.r_period ? row % (size_t)a.r_period : row;
                    const int col = col0 + (wn * NJ + jp * 2) * 32 + ec;
                    res[q] = (row < (size_t)row_end && ec < NJ * 32) ? *reinterpret_cast<const float4 *>(a.R + rrow * a.ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int jj = 0; jj < (NJ < 2 ? NJ : 2); ++jj) {
                const int j = jp * 2 + jj;
                const float bias = bias_r[j];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    est[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ELD + jj * 32 + l31] = acc[i][j][r] + bias;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 16-bit outputs only (the [Wa|Wb] projection of the 16-bit engines: A as fp16, Bm as fp16): 8 lanes x 16 B cover a
            // row's 64 columns (one 128-byte line), 8 rows = 1 KiB per store instruction - the 4 x 16 float4 read-back below
            // would emit 8-byte stores, twice the instructions for the same bytes (r02 stamps: 60 % of that launch was epilogue)
            if constexpr (HALF) {
                uint16_t *half_out = col0 < H ? a.Cb : a.C2b;
                const int er8 = lane >> 3, ec8 = (lane & 7) * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int lr = q * 8 + er8;
                    const float4 v0 = *reinterpret_cast<const float4 *>(est + lr * ELD + ec8);
                    const float4 v1 = *reinterpret_cast<const float4 *>(est + lr * ELD + ec8 + 4);
                    const size_t row = (size_t)row0 + (wm * MI + i) * 32 + lr;
                    const int col = (col0 < H ? col0 : col0 - H) + (wn * NJ + jp * 2) * 32 + ec8;
                    if (row >= (size_t)row_end || ec8 >= NJ * 32) continue;
                    uint4 o;
                    o.x = pack_h2_sat(v0.x, v0.y); o.y = pack_h2_sat(v0.z, v0.w);
                    o.z = pack_h2_sat(v1.x, v1.y); o.w = pack_h2_sat(v1.z, v1.w);
                    *reinterpret_cast<uint4 *>(half_out + row * H + col) = o;
                }
            } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int lr = q * 4 + er;
                const float4 v = *reinterpret_cast<const float4 *>(est + lr * ELD + ec);
                const size_t row = (size_t)row0 + (wm * MI + i) * 32 + lr;
                const int col = col0 + (wn * NJ + jp * 2) * 32 + ec;
                if (row >= (size_t)row_end || ec >= NJ * 32) continue;
                {
                    if (a.stat_part) {
#pragma clang fp contract(off)      // every tile shape must round these sums alike: no instantiation-dependent fma formation
                        if (st_n[i] == 0.f) { st_p[i][0] = v.x; st_p[i][1] = v.y; st_p[i][2] = v.z; st_p[i][3] = v.w; }
                        const float d0 = v.x - st_p[i][0], d1 = v.y - st_p[i][1], d2 = v.z - st_p[i][2], d3 = v.w - st_p[i][3];
                        st_s[i][0] += d0; st_s[i][1] += d1; st_s[i][2] += d2; st_s[i][3] += d3;
                        st_q[i][0] += d0 * d0; st_q[i][1] += d1 * d1; st_q[i][2] += d2 * d2; st_q[i][3] += d3 * d3;
                        st_n[i] += 1.f;
                    }
                }
                if (a.epi == 1) {
                    const float4 rr = res[q];
                    *reinterpret_cast<float4 *>(a.C + row * a.ldc + col) = make_float4(rr.x + v.x, rr.y + v.y, rr.z + v.z, rr.w + v.w);
                } else if (a.epi == 2) {
                    if (col0 < H) {
                        if (a.Cb) {
                            uint2 o;
                            o.x = pack_h2_sat(v.x, v.y);
                            o.y = pack_h2_sat(v.z, v.w);
                            *reinterpret_cast<uint2 *>(a.Cb + row * H + col) = o;
                        } else *reinterpret_cast<float4 *>(a.C + row * H + col) = v;
                    } else {
                        if (a.C2) *reinterpret_cast<float4 *>(a.C2 + row * H + (col - H)) = v;
                        if (a.C2b) {
                            uint2 o;
                            o.x = pack_h2_sat(v.x, v.y);
                            o.y = pack_h2_sat(v.z, v.w);
                            *reinterpret_cast<uint2 *>(a.C2b + row * H + (col - H)) = o;
                        }
                    }
                } else {
                    *reinterpret_cast<float4 *>(a.C + row * a.ldc + col) = v;
                }
            }
            }   // full-width outputs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    if (a.stat_part) {
#pragma clang fp contract(off)
        // each 32-row half's statistics -> partial (tile's first half index + wm * MI + i) of the trajectory's ceil(N / 32); a half past the
        // trajectory's last row has no slot.  (After BOTH row passes: emitted between them the block costs the large launches 8 %.)
        const int tpt64 = (a.rows_per_graph + 63) / 64, tpt32 = (a.rows_per_graph + 31) / 32;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int sub = (vb % tpt64) * 2 + wm * MI + i;
            float *sp = a.stat_part + ((size_t)(vb / tpt64) * tpt32 + sub) * (H * 2);
            const float inv_n = st_n[i] > 0.f ? 1.0f / st_n[i] : 0.f;
            // the four row groups (er) of a column, merged in a fixed order (Chan).  The counts are the same for the lane's four columns:
            // their exchange and the weight n2 / (n + n2) of each level are computed once (values unchanged: the same expression), and
            // the merge is branch-free (an empty partner enters with weight 0) so that the 16 exchanges of a half pipeline - at small
            // launches this block is exposed at the end of every workgroup (14 k cycles of node_mlp.0 at B = 8 before)
            float nl[2], fl[2];
            {
                float ne = st_n[i];
#pragma unroll
                for (int lv = 0; lv < 2; ++lv) {
                    const float n2 = __shfl_xor(ne, 16 << lv, 64), nt = ne + n2;
                    nl[lv] = ne;
                    fl[lv] = nt > 0.f ? n2 / nt : 0.f;
                    ne = nt;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mean = st_p[i][e] + st_s[i][e] * inv_n, M2 = st_q[i][e] - st_s[i][e] * st_s[i][e] * inv_n;
#pragma unroll
                for (int lv = 0; lv < 2; ++lv) {
                    const float mean2 = __shfl_xor(mean, 16 << lv, 64), M22 = __shfl_xor(M2, 16 << lv, 64);
                    const float d = mean2 - mean, f = fl[lv];
                    mean += d * f;
                    M2 += M22 + d * d * nl[lv] * f;
                }
                if (er == 0 && ec < NJ * 32 && sub < tpt32) {       // only this lane's merge order is ever read: deterministic, the same for every trajectory
                    const int c = col0 + wn * (NJ * 32) + ec + e;
                    sp[c * 2] = mean;
                    sp[c * 2 + 1] = M2;
                }
            }
        }
    }
#ifdef DFM_GEMM_STAMP
    GSTAMP(3)                      // [3] epilogue
    if (vb == (gridDim.x > 2048 ? 7 * 8 + 1536 : 3) && blockIdx.y == 0 && tid == 0 && a.C) {       // one mid-grid workgroup reports (debug buffer = first floats of ... stderr-free: printf)
        const unsigned long long g_rt1 = __builtin_amdgcn_s_memrealtime();
        printf("gemm stamp K=%d Nout=%d pro=%d epi=%d stats=%d gn=%d: entry->loop %llu  mfma+barrier %llu  fetch-wait+stage %llu  barrier2 %llu  epilogue %llu  total %llu shader cycles in %llu ticks of 100 MHz = %llu MHz\n",
               a.K, a.Nout, a.pro, a.epi, a.stat_part ? 1 : 0, a.gn_part ? 1 : 0, g_first - g_entry, gs[0], gs[1], gs[2], gs[3], gprev - g_entry, g_rt1 - g_rt0,
               (g_rt1 > g_rt0) ? 100ull * (gprev - g_entry) / (g_rt1 - g_rt0) : 0ull);
    }
#endif
}
#undef GEMM_SPLIT_FETCH

int gemm_rows_per_tile() { return 64; }

hipError_t launch_gemm_split(const GemmArgs &a, const uint16_t *Whi, const uint16_t *Wlo, hipStream_t s)
{
    constexpr int SN = 256;
    if (a.K % SK != 0 || a.Nout % SN != 0 || (a.pro == 1 && (a.K / 2) % SK != 0) || a.lda % 4 != 0 || a.ldc % 4 != 0)
        return hipErrorInvalidValue;
    GemmSplitArgs sa;
    sa.g = a; sa.Whi = Whi; sa.Wlo = Wlo;
    if (a.stat_part && (a.Nout != SN || a.rows_per_graph < 1 || a.M % a.rows_per_graph != 0)) return hipErrorInvalidValue;
    if (a.pro == 2 && (a.rows_per_graph < 1 || a.M % a.rows_per_graph != 0)) return hipErrorInvalidValue;
    if (a.zbuf && a.Nout != SN) return hipErrorInvalidValue;
    const int row_tiles = (a.stat_part || a.pro == 2) ? (a.M / a.rows_per_graph) * ((a.rows_per_graph + 63) / 64) : (a.M + 63) / 64;
    // fewer 64 x 256 workgroups than two per CU: 64 x 128 tiles (NJ = 1) - a lone workgroup is bound by its CU's own memory pipe
    static const int narrow_env = [] {
        const char *e = getenv("DFM_GEMM_NARROW_MAXWG");      // diagnostics: 64 x 256 workgroup count below which NJ = 1 is used (0 = never)
        return e ? atoi(e) : -1;
    }();
    // per call: device_cus() is the CURRENT device's count (a process may drive GPUs of different sizes).  2 per CU measured
    // (profiles/r03_d_small_narrow.txt): +17 % at B = 1, +8.5 % at B = 8, +1.5 % at B = 32, even at B = 64 (300+300)
    const int narrow_max = narrow_env >= 0 ? narrow_env : 2 * device_cus();
    const bool narrow = (long long)row_tiles * (a.Nout / SN) < narrow_max;
    const bool half = a.epi == 2 && a.Cb && a.C2b && !a.C2;
    if (narrow) {
        // ... and 64 x 64 tiles (QT) while even those leave more than half of the CUs without a workgroup (B <= 6 at 300+300)
        static const int quarter_env = [] {
            const char *e = getenv("DFM_GEMM_QUARTER_MAXWG");      // diagnostics: 64 x 128 workgroup count below which QT is used (0 = never)
            return e ? atoi(e) : -1;
        }();
        const int quarter_max = quarter_env >= 0 ? quarter_env : device_cus() / 2;
        if ((long long)row_tiles * (a.Nout / 128) < quarter_max && (a.K / 32) % 4 == 0) {      // (the quarter-tile K loop runs four stages per trip)
            const dim3 grid(row_tiles, a.Nout / 64);
            if (half) hipLaunchKernelGGL((k_gemm_split<1, 1, 1>), grid, dim3(256), 0, s, sa);
            else hipLaunchKernelGGL((k_gemm_split<0, 1, 1>), grid, dim3(256), 0, s, sa);
            return hipGetLastError();
        }
        const dim3 grid(row_tiles, a.Nout / 128);
        if (half) hipLaunchKernelGGL((k_gemm_split<1, 1>), grid, dim3(256), 0, s, sa);
        else hipLaunchKernelGGL((k_gemm_split<0, 1>), grid, dim3(256), 0, s, sa);
        return hipGetLastError();
    }
    dim3 grid;
    if (a.stat_part || a.pro == 2) grid = dim3(row_tiles, a.Nout / SN);   // row tiles aligned to the trajectories
    else if (a.Nout == 2 * SN) grid = dim3((((a.M + 63) / 64 + 7) / 8) * 16, 1);      // paired column blocks, see the block mapping in the kernel
    else grid = dim3((a.M + 63) / 64, a.Nout / SN);
    if (half) hipLaunchKernelGGL((k_gemm_split<1, 2>), grid, dim3(256), 0, s, sa);
    else hipLaunchKernelGGL((k_gemm_split<0, 2>), grid, dim3(256), 0, s, sa);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GraphNorm statistics per trajectory and channel (torch_geometric 2.6.0 graph_norm.py, batch=None):
//   mean = mean_n u ;  shift = mean * mean_scale ;  var = mean_n (u - shift)^2 ;  den = sqrt(var + 1e-5)
// Two exact passes in float64 (no E[x^2]-E[x]^2 cancellation), deterministic (no atomics).
// grid (B, 4): each workgroup owns 64 channels of one trajectory; 256 threads = 16 row lanes x 16 float4 channel quads
// (a wave instruction reads four 256-byte row segments; the second pass re-reads the 150 KB slab from L2).
__global__ __launch_bounds__(256) void k_gn_stats(const float *__restrict__ u, int N, const float *__restrict__ mean_scale,
                                                  float *__restrict__ shift, float *__restrict__ den,
                                                  const float *__restrict__ fold_w, const float *__restrict__ fold_b)
{
    __shared__ double red[16][64];
    __shared__ float sh_shift[64];
    const int b = blockIdx.x, cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.y * 64;
    const float *U = u + (size_t)b * N * H + c0 + cq * 4;
    auto reduce_rows = [&](const double (&s)[4]) {     // fixed-order sum over the 16 row lanes -> red[0][channel]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = s[e];
        __syncthreads();
        if (threadIdx.x < 64) {
            double t = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
            red[0][threadIdx.x] = t;
        }
        __syncthreads();
    };
    double s[4] = {0, 0, 0, 0};
    for (int n = rl; n < N; n += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(U + (size_t)n * H);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    reduce_rows(s);
    if (threadIdx.x < 64) sh_shift[threadIdx.x] = (float)(red[0][threadIdx.x] / N) * mean_scale[c0 + threadIdx.x];
    __syncthreads();
    const float4 sft = *reinterpret_cast<const float4 *>(&sh_shift[cq * 4]);
    double v2[4] = {0, 0, 0, 0};
    for (int n = rl; n < N; n += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(U + (size_t)n * H);
        const float o0 = v.x - sft.x, o1 = v.y - sft.y, o2 = v.z - sft.z, o3 = v.w - sft.w;
        v2[0] += (double)o0 * o0; v2[1] += (double)o1 * o1; v2[2] += (double)o2 * o2; v2[3] += (double)o3 * o3;
    }
    reduce_rows(v2);
    if (threadIdx.x < 64) {
        const int c = c0 + threadIdx.x;
        const float var = (float)(red[0][threadIdx.x] / N);
        const float dn = sqrtf(var + 1e-5f), sf = sh_shift[threadIdx.x];
        if (fold_w) {   // y = w*(x - shift)/den + b  ==  x*sc + sh
            const float sc = fold_w[c] / dn;
            den[(size_t)b * H + c] = sc;
            shift[(size_t)b * H + c] = fold_b[c] - sc * sf;
        } else {
            shift[(size_t)b * H + c] = sf;
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
