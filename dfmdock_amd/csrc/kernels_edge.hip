// kernels_edge.hip - the EGNN edge model, attention gate, fixed-degree segment sum and the last layer's
// coordinate update (reference: src/models/egnn.py:95-159).  This is ~92 % of the algorithmic FLOPs.
//
// Exact restructuring used by both kernels:
//   Linear_1([h_i, h_j, radial, e_ij]) = (Wa h_i + b1) + Wb h_j + w_r * radial + sum of 5 rows of T_l
// with A = Wa h + b1 and Bm = Wb h per NODE (kernels_dense.hip) and T_l = [S|P]^T We_l^T a per-layer
// lookup table (one-hot -> Linear == row gather).  Every node has exactly K out-edges stored
// contiguously, so scatter_add is a dense K-row reduction: no atomics anywhere.
//
//   k_edge_f32  : exact fp32 (VALU) - the parity-reference precision of the engine.
//   k_edge_msg / k_edge_coord : 256x256 contraction on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate; A-fragments are
//                 built in registers straight from the gathers, the weight matrix lives in LDS for the
//                 whole (persistent) workgroup.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "dfm_device.h"
#include "dfm_internal.h"
#include "dfm_edge_knobs.h"

namespace dfm {

// =================================================================================================
// fp32 kernel: one 256-thread workgroup per (trajectory, node); thread = channel.
constexpr int KF = 60;   // accumulator rows held in registers (K <= 60 always: knn 20 + sample 40)

struct EdgeKArgs {
    const float *A, *Bm;
    const uint16_t *Bmb;
    long long ab_bstride;
    const int32_t *edges;
    const uint32_t *codes;
    const float *radial;
    const float4 *ca4;
    int B, N, R, K, L;
    const float *w_r, *T, *W2t, *b2, *att_w;
    const uint16_t *T2b;
    const uint4 *Wf;
    float att_b;
    const float *Wc1t, *bc1, *wc2;
    float *agg;
    int last;
    float *fout;
    uint16_t *mbuf;
    // 16-bit MFMA kernels: everything upstream of a SiLU is pre-multiplied by -log2(e) on the host / in the producing GEMM
    // (see SILU_S), so SiLU is exp2 -> +1 -> rcp -> mul with no scaling multiply; biasp = the contraction's bias as packed
    // (hi, lo) 16-bit pairs laid out per (n-tile, lane) [8][64] with zeros for lanes 32..63, added by one extra MFMA k-step
    // instead of 128 accumulator moves
    const uint32_t *biasp;
    float inv_s;
    unsigned long long *stamp;   // DFM_EDGE_STAMP builds only: per-phase cycle sums of workgroup 0 (tools/edge_phases.py)
    const uint16_t *Ah;          // k_edge_msg<0, 1>: A as fp16
    int split;                   // k_edge_msg: 1 = a wave task is one TILE (small launches), agg is pre-zeroed and added to atomically
    int no_agg;                  // message kernels: nobody reads agg after this launch (ligand-only last layer): no segment-sum store, no atomics, no memset
    int node0, nodes;            // message kernels: the tasks cover nodes node0 .. node0 + nodes - 1 of every trajectory (all of them, or - last
                                 // layer when nobody reads the node outputs - the ligand nodes only: EdgeArgs::lig_only)
    uint32_t *range;             // k_edge_f32, dfm_complex_selfcheck only: [0] max |pre-activation of edge_mlp.0|, [1] of edge_mlp.2, as float bits
    // k_edge_msg<1, 1, 1> (row-list form, layer 0 only: A / Bm are the complex's own, pose-independent operands): the tasks are 32-row
    // tiles of a flat list of edges (i, j, code, radial bits); the gated messages go out row-major as fp16 [row][256]
    const uint4 *rows;
    const uint32_t *n_rows_dev;  // row count on the device (filled by k_edge_feat's classification), or nullptr: n_rows
    uint32_t n_rows;
    uint16_t *rows_out;
    float *rows_out32;           // k_edge_f32m<1>: gated messages of the row list, row-major fp32 [row][256]
    uint32_t *task_ctr;          // k_edge_msg, node tasks of large launches: [TASK_CTR_WGS] per-workgroup task counters + as many exit counters, all
                                 // zero at launch (the last wave out of a workgroup zeroes its pair again) - the waves of a workgroup take the
                                 // workgroup's tasks in order from here instead of by a fixed stride ("dynamic tasks" in the kernel), or nullptr
};

__device__ inline void row_dot(const float *lds_rows /*[KF][256]*/, const float *__restrict__ Wt /*[256][256]*/,
                               int c, float bias, float (&acc)[KF])
{
#pragma unroll
    for (int s = 0; s < KF; ++s) acc[s] = bias;
    for (int k = 0; k < H; k += 4) {
        const float w0 = Wt[(size_t)(k + 0) * H + c], w1 = Wt[(size_t)(k + 1) * H + c],
                    w2 = Wt[(size_t)(k + 2) * H + c], w3 = Wt[(size_t)(k + 3) * H + c];
#pragma unroll
        for (int s = 0; s < KF; ++s) {
            const float4 a = *reinterpret_cast<const float4 *>(lds_rows + s * H + k);   // broadcast read
            float t = acc[s];
            t = fmaf(a.x, w0, t); t = fmaf(a.y, w1, t); t = fmaf(a.z, w2, t); t = fmaf(a.w, w3, t);
            acc[s] = t;
        }
    }
}

__global__ __launch_bounds__(256) void k_edge_f32(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *rows = reinterpret_cast<float *>(smem);            // [KF][256]
    float *s_rad = rows + KF * H;                             // [64]
    float *s_gate = s_rad + 64;                               // [64]
    int *s_j = reinterpret_cast<int *>(s_gate + 64);          // [64]
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_j + 64);   // [64]

    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const long long node = blockIdx.x;
    const int b = (int)(node / p.N), i = (int)(node % p.N), K = p.K;
    if (i < p.node0) return;      // lig_only: receptor nodes are not needed (block-uniform)
    const size_t ebase = (size_t)node * K;
    if (c < 64) {
        const bool v = c < K;
        s_j[c] = v ? p.edges[ebase + c] : i;
        s_code[c] = v ? p.codes[ebase + c] : 0u;
        s_rad[c] = v ? p.radial[ebase + c] : 0.f;
    }
    __syncthreads();
    const size_t ab = (size_t)b * p.ab_bstride;
    const float Ai = p.A[ab + (size_t)i * H + c];
    const float wr = p.w_r[c];
    // edge_mlp.0 + SiLU  (egnn.py:95-101)
    float pre_max = 0.f;      // range telemetry (p.range): largest |pre-activation| this thread saw
    for (int s = 0; s < KF; ++s) {
        float v = 0.f;
        if (s < K) {
            const uint32_t code = s_code[s];
            const int j = s_j[s];
            float pre = Ai + p.Bm[ab + (size_t)j * H + c];
            pre += wr * s_rad[s];
            pre += p.T[(size_t)(code & 63u) * H + c];
            pre += p.T[(size_t)(40u + ((code >> 6) & 31u)) * H + c];
            pre += p.T[(size_t)(64u + ((code >> 11) & 31u)) * H + c];
            pre += p.T[(size_t)(88u + ((code >> 16) & 15u)) * H + c];
            pre += p.T[(size_t)(100u + ((code >> 20) & 127u)) * H + c];
            pre_max = fmaxf(pre_max, fabsf(pre));
            v = silu_exact(pre);
        }
        rows[s * H + c] = v;
    }
    __syncthreads();
    // edge_mlp.2 + SiLU
    float acc[KF];
    row_dot(rows, p.W2t, c, p.b2[c], acc);
    __syncthreads();
    if (p.range) {      // non-negative floats order like their bit patterns: one atomicMax per wave and quantity
        float acc_max = 0.f;
#pragma unroll
        for (int s = 0; s < KF; ++s) acc_max = s < K ? fmaxf(acc_max, fabsf(acc[s])) : acc_max;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            pre_max = fmaxf(pre_max, __shfl_xor(pre_max, m, 64));
            acc_max = fmaxf(acc_max, __shfl_xor(acc_max, m, 64));
        }
        if (lane == 0) { atomicMax(p.range, __float_as_uint(pre_max)); atomicMax(p.range + 1, __float_as_uint(acc_max)); }
    }
#pragma unroll
    for (int s = 0; s < KF; ++s) {
        acc[s] = silu_exact(acc[s]);
        rows[s * H + c] = acc[s];
    }
    __syncthreads();
    // attention gate (egnn.py:102-104): sigmoid(att_w . m + att_b) per edge
    for (int s = wave; s < KF; s += 4) {
        const float4 m4 = *reinterpret_cast<const float4 *>(rows + s * H + lane * 4);
        const float4 w4 = *reinterpret_cast<const float4 *>(p.att_w + lane * 4);
        float t = m4.x * w4.x + m4.y * w4.y + m4.z * w4.z + m4.w * w4.w;
        t = wave_sum(t);
        if (lane == 0) s_gate[s] = (s < K) ? sigmoid_exact(t + p.att_b) : 0.f;
    }
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KF; ++s) {
        acc[s] *= s_gate[s];
        sum += acc[s];   // unsorted_segment_sum over this node's edges, in edge order
    }
    p.agg[(size_t)node * H + c] = sum;

    if (p.last && i >= p.R) {
        // coord_model (egnn.py:118-137) for ligand nodes (lig_mask)
#pragma unroll
        for (int s = 0; s < KF; ++s) rows[s * H + c] = acc[s];
        __syncthreads();
        float cacc[KF];
        row_dot(rows, p.Wc1t, c, p.bc1[c], cacc);
        __syncthreads();
        const float w2 = p.wc2[c];
#pragma unroll
        for (int s = 0; s < KF; ++s) rows[s * H + c] = silu_exact(cacc[s]) * w2;
        __syncthreads();
        for (int s = wave; s < KF; s += 4) {
            const float4 m4 = *reinterpret_cast<const float4 *>(rows + s * H + lane * 4);
            float t = (m4.x + m4.y) + (m4.z + m4.w);
            t = wave_sum(t);
            if (lane == 0) s_gate[s] = fminf(fmaxf(t, -2.0f), 2.0f);   // clamp_(-2, 2)
        }
        __syncthreads();
        if (c < 3) {
            const float4 *ca = p.ca4 + (size_t)b * p.N;
            const float4 xi = ca[i];
            const float xi_d = c == 0 ? xi.x : (c == 1 ? xi.y : xi.z);
            float a = 0.f;
            for (int s = 0; s < K; ++s) {
                const float4 xj = ca[s_j[s]];
                const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
                const float r2 = (dx * dx + dy * dy) + dz * dz;
                const float nrm = sqrtf(r2 + 1e-8f) + 1.0f;      // coord2radial, normalize=True
                const float dd = (c == 0 ? dx : (c == 1 ? dy : dz)) / nrm;
                a += dd * s_gate[s];
            }
            a = a / (float)(K > 1 ? K : 1);                        // unsorted_segment_mean
            const float moved = xi_d + a;                          // coord + agg * lig_mask
            p.fout[((size_t)b * p.L + (i - p.R)) * 3 + c] = moved - xi_d;   // f = pos_out - r
        }
    }
}

// =================================================================================================
// 16-bit MFMA kernel (bf16 or fp16 operands).
//
// Workgroup = 8 waves, persistent, all 160 KiB of LDS: 128 KiB hold the 16-bit B-fragments of the 256x256
// weight matrix for the whole launch, 32 KiB are eight wave-private 4 KiB staging areas.  A wave owns one
// node at a time = two 32-row M-tiles (60 edges + 4 masked rows).  Per M-tile, in eight 32-channel chunks:
//   producer  (gather layout: 4 adjacent lanes cover one row's 32 channels = a 64-byte half line, 16 rows per pass):
//             A_i + Bm_j + w_r*radial + 2 merged T rows (fp16 gathers; see NTAB2) -> SiLU -> bf16/fp16 -> ds_write_b128 into the
//             staging buffer of the NEXT chunk ([8-channel unit][row ^ 4*unit] x 16 B: conflict-free both ways)
//   consumer  2 k-steps x 8 n-tiles of v_mfma_f32_32x32x16_{bf16,f16} on the CURRENT chunk's buffer, A-fragments and
//             the resident weight fragments by ds_read_b128 (reads one ahead).
// The producer arithmetic is cut into slices laid between the MFMAs in program order (scheduling barriers keep them
// there), so a wave's MFMAs run in the shadow of its own VALU work; the gathers of chunk c + 2 are in flight meanwhile.
// The chunk loop wraps around the tile boundary (chunk 6 requests, chunk 7 builds chunk 0 of the wave's next tile), so a tile
// has no prologue.  Epilogue in the C layout (lane = column, registers = rows): SiLU (bias added by one extra MFMA k-step),
// attention gate (in-lane dot + DPP row reduction), row mask, 60-row segment sum in registers -> agg; no atomics.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 b; f16x8 f; };
union H8 { uint4 u; __half2 h[4]; };   // eight fp16 values of one gathered 16-byte chunk

constexpr int LDS_WF_BYTES = 16 * 8 * 64 * 16;     // 131072: bf16 B-fragments of one 256x256 matrix
constexpr int LDS_STAGE_BYTES = 32 * 64 * 2;       // 4096 per wave: 32 rows x 64 channels bf16
constexpr int EDGE_WAVES = DFM_EDGE_WAVES;         // waves per workgroup: 8 = two per SIMD (256 registers each), 4 = one per SIMD (512)
constexpr int LDS_EDGE_BYTES = LDS_WF_BYTES + EDGE_WAVES * LDS_STAGE_BYTES;   // 163840 = the whole CU with 8 waves
// Diagnostic build (WRONG RESULTS BY DESIGN, r06): DFM_EDGE_HALF = the per-wave work of a kernel in which TWO waves share a 32-row tile
// (each: 16 of the 32 producer rows, 128 of the 256 output columns = 64 accumulators) without the pair's hand-shakes; DFM_MSG_WAVES waves
// per workgroup run it (12 = three per SIMD at <= 168 registers; staging areas of waves >= 8 alias those of waves 0..3: timing only).
#ifndef DFM_EDGE_HALF
#define DFM_EDGE_HALF 0
#endif
#ifndef DFM_MSG_WAVES
#define DFM_MSG_WAVES DFM_EDGE_WAVES
#endif
constexpr int MSG_WAVES = DFM_MSG_WAVES;
constexpr int MSG_NT = DFM_EDGE_HALF ? 4 : 8;      // n-tiles (32 columns) of the output a wave owns

typedef float f2 __attribute__((ext_vector_type(2)));   // packed fp32 pair -> v_pk_{mul,add,fma}_f32 (2 results / instr)
// a + (float)half of a packed fp16 pair in ONE plain-rate instruction (v_fma_mix_f32: f16 source 0 times 1.0 plus f32 source 2);
// hipcc otherwise emits v_cvt_f32_f16 x2 + v_pk_add_f32, twice the issue time (tools/ubench/valu_rate.hip)
__device__ inline float add_half_lo(float a, uint32_t h2)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(a));
    return r;
}
__device__ inline float add_half_hi(float a, uint32_t h2)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(a));
    return r;
}
// w * r + (float)a.lo and the .hi counterpart: fp32 sources 0 and 1, fp16 source 2
__device__ inline float fma_half_lo(float w, float r, uint32_t a2)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(w), "v"(r), "v"(a2));
    return d;
}
__device__ inline float fma_half_hi(float w, float r, uint32_t a2)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(w), "v"(r), "v"(a2));
    return d;
}
__device__ inline f2 add_half2(f2 a, __half2 h)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, h);
    return (f2){add_half_lo(a.x, u), add_half_hi(a.y, u)};
}

__device__ inline void acc8(float (&v)[8], const uint4 &q)
{
    v[0] += bflo(q.x); v[1] += bfhi(q.x); v[2] += bflo(q.y); v[3] += bfhi(q.y);
    v[4] += bflo(q.z); v[5] += bfhi(q.z); v[6] += bflo(q.w); v[7] += bfhi(q.w);
}
__device__ inline void acc8f(float (&v)[8], const float4 &a, const float4 &b)
{
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
}

// sum over the 32 lanes sharing lane>>5, result in every lane: 4 DPP adds inside each 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) + one ds_swizzle (xor 16) across the two rows
__device__ inline float half_sum_dpp(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // xor 0x10, and 0x1f
    return v;
}

// Row sums over the 32 lanes of a half as a REDUCE-SCATTER: v[0..15] are this lane's partial sums of 16 rows; each butterfly step
// (lane ^ 1, ^ 2 by DPP; ^ 4, ^ 8 by ds_swizzle) halves the values a lane carries - the lane keeps the half its bit selects and
// sends the other - and a last step (^ 16) adds the two 16-lane rows.  46 VALU instructions instead of the 80 of sixteen
// all-reduces, and lane l ends with the ONE sum of row index rs_index(l): whatever follows per row (the attention gate's
// exp / rcp) runs once per lane instead of sixteen times.
__device__ inline int rs_index(int lane) { return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3); }
__device__ inline float half_reduce_scatter(const float (&v)[16], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float a8[8], a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float keep = b0 ? v[8 + j] : v[j], send = b0 ? v[j] : v[8 + j];
        a8[j] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b1 ? a8[4 + j] : a8[j], send = b1 ? a8[j] : a8[4 + j];
        a4[j] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b2 ? a4[2 + j] : a4[j], send = b2 ? a4[j] : a4[2 + j];
        a2[j] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F));   // xor 4
    }
    const float keep = b3 ? a2[1] : a2[0], send = b3 ? a2[0] : a2[1];
    float a1 = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x201F));    // xor 8
    a1 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a1), 0x401F));                  // xor 16
    return a1;
}

// make every earlier LDS access of this wave visible/ordered before later ones (wave-private staging tile)
__device__ inline void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Transposing stores (last layer's gated messages, row-list form): the staging area is wave-private and a wave's LDS instructions execute in order, so neither
// the reads after the ds_write_b16 nor the next n-tile's writes after the reads need the counter drained - DFM_EDGE_MSTORE_NOFENCE keeps the compiler's order only
__device__ inline void mstore_fence() { if constexpr (DFM_EDGE_MSTORE_NOFENCE) asm volatile("" ::: "memory"); else wave_lds_fence(); }

#if DFM_TAB_MERGE
struct RawP { uint4 bm, t0, t1; };            // gathered fp16 operands of one producer pass (8 channels of one row)
#else
struct RawP { uint4 bm, t0, t1, t2; };
#endif

// one MFMA step on bf16 (F16 = 0) or fp16 (F16 = 1) operands, fp32 accumulate - same rate on gfx950
template <int F16> __device__ inline f32x16 mfma16(const Frag &a, const Frag &b, f32x16 c)
{
#if DFM_EDGE_KO & 1      // knock-out build (wrong results): no matrix instruction, operands still read
    asm volatile("" : "+v"(c) : "v"(a.u.x), "v"(a.u.w), "v"(b.u.x), "v"(b.u.w));      // (opaque: the epilogue is not folded away)
    return c;
#endif
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f, b.f, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c, 0, 0, 0);
}
// two fp32 -> packed fp16 (round to nearest even), -inf / below -65504 clamped to -65504 by ONE packed max on the converted pair
// (the values on this path are S * silu(.) [* gate]: bounded above by 0.41, unbounded below; no NaN from finite inputs)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack_f16_sat_lo(float a, float b)
{
    f16x2 v = {(_Float16)a, (_Float16)b};
    const f16x2 lo = {(_Float16)-65504.f, (_Float16)-65504.f};
    v = __builtin_elementwise_max(v, lo);
    return __builtin_bit_cast(uint32_t, v);
}
template <int F16> __device__ inline uint16_t to16(float x)
{
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(x, -65504.f), 65504.f));
    else return __builtin_bit_cast(uint16_t, (__bf16)x);
}

// Every input of a SiLU on this path is carried pre-multiplied by SILU_S = -log2(e): silu(x) = x * sigmoid(x) with
// x' = SILU_S * x is (x' / SILU_S) / (1 + exp2(x')), i.e. exp2 -> +1 -> rcp -> mul on x' and a constant factor 1 / SILU_S
// that the next linear stage absorbs:
//   producer   pre' = S * (A_i + Bm_j + w_r r^2 + table rows)   (tables, w_r, [Wa|Wb] and b1 are scaled on the host / in the
//              producing GEMM: api.hip)                          m' = pre' / (1 + exp2(pre')) = S * m
//   contraction acc' = W2 m' + S b2 = S * (W2 m + b2)            (the weights are untouched)
//   epilogue   m2' = acc' / (1 + exp2(acc')) = S * m2 ;  gate = 1 / (1 + exp2(att_w . m2' + S att_b)) ;
//              agg = (1 / S) * sum_rows gate * m2' ;  stored messages = gate * m2' = S * (gated message)
//   coord MLP  acc' = Wc1 (S m~) + S bc1 ;  c' = S * c ;  w = (wc2 / S) . c'

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// 16-byte buffer load: wave-uniform resource + base in SGPRs, per-lane byte offset in ONE VGPR, wave-uniform extra offset in
// an SGPR / the immediate field - no 64-bit VALU address arithmetic per gather
__device__ inline uint4 bload16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline float4 bload16f(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// Streams that pass through once (A_i rows, edge data, agg / message stores) carry the non-temporal hint, so that they do not
// push the lookup tables and the re-gathered Bm rows out of the XCD's 4 MiB L2 (cache-policy bit 1 of the buffer instructions)
constexpr int AUX_STREAM = DFM_EDGE_NT ? 2 : 0;
__device__ inline float4 bload16f_stream(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, AUX_STREAM);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ inline void store_stream(float *p, float v)
{
    if constexpr (DFM_EDGE_NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ inline void store_stream(uint4 *p, uint4 v)
{
    if constexpr (DFM_EDGE_NT) __builtin_nontemporal_store((u32x4v){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4v *>(p)); else *p = v;
}
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *base)
{   // raw buffer (stride 0), 2 GiB window, dword-format descriptor word 3 of gfx9
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00027000);
}
// knock-out builds (DFM_EDGE_KO; WRONG results by design): bit 0 no MFMA, bit 1 no producer transcendentals, bit 2 no epilogue transcendentals
template <int KO> __device__ inline float ko_exp2(float x) { if constexpr (KO) return x; else return __builtin_amdgcn_exp2f(x); }
template <int KO> __device__ inline float ko_rcp(float x) { if constexpr (KO) return x; else return __builtin_amdgcn_rcpf(x); }
// SiLU of two pre-scaled values (see SILU_S): 2 v_exp_f32 + 2 v_rcp_f32 + 2 packed ops
__device__ inline f2 silu2s(f2 x)
{
    f2 e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
    e = e + (f2){1.0f, 1.0f};
    const f2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return x * r;
}

// =================================================================================================
// fp32 engine on the matrix pipe: the same algebra as k_edge_f32, the two 256 x 256 contractions on v_mfma_f32_32x32x2_f32 (exact
// fp32 products and accumulation - the reference's own arithmetic, src/models/egnn.py:95-137 - at 16x the scalar-FMA rate of one
// lane).  fp32 weights do not fit the LDS (256 KiB), so the kernel is a synchronous tiled GEMM with the edge model fused around it:
//
//   workgroup = 4 waves = 2 nodes = four 32-row tiles (60 edges + 4 masked rows per node); 1 workgroup per CU (105 KiB LDS, 512 registers)
//   K loop in eight 32-channel chunks, double-buffered through LDS: the weight chunk W2t[32 k][256 n] (32 KiB, shared by the four
//   waves) and the producer's chunk m1 = SiLU(A_i + Bm_j + w_r r^2 + 5 table rows) [32 k][128 rows] (stride 129: conflict-free both
//   ways); the global loads of chunk c + 1 (weights + gathers, thread = one row x 16 channels) are in flight under chunk c's 128 MFMAs
//   epilogue in the C layout like the 16-bit kernel: bias, exact SiLU, attention logits by reduce-scatter, gate, 60-row segment sum
//   (two tiles of a node combined through LDS), agg store
//   last layer, ligand nodes: the coordinate MLP as a second K loop whose A operand is the gated message tile, transposed chunk by
//   chunk through the wave's rows of the staging buffer; clamp, normalised differences, mean -> f
//
// Summation order differs from k_edge_f32 (MFMA k order, tree reductions) at the 1e-7 level; DFM_EDGE_F32_SCALAR=1 selects the
// scalar kernel (A/B timing, profiles/r04_fp32_engine.txt).
constexpr int FM_LD = 129;                       // floats per k-row of the m1 staging ([k][128 rows] + 1)
constexpr int FM_W_FLOATS = 32 * 256;            // one weight chunk
constexpr int FM_M_FLOATS = 32 * FM_LD;          // one staging chunk
constexpr int LDS_F32M_BYTES = (2 * FM_W_FLOATS + 2 * FM_M_FLOATS + 4 * 256 + 16 + 4 * 128) * 4;

// SiLU on the hardware's 1-ulp exp2 / rcp: x / (1 + exp2(-x log2 e)), ~3 ulp; saturates correctly (exp2 -> inf: x * 0; -> 0: x * 1).
// silu_exact (expf + IEEE division, ~35 instructions) cost the kernel more VALU time than its MFMAs take
__device__ inline float silu_f32m(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }

// ROWS = 1 (layer 0 of the fp32 engine behind the per-complex message table, like k_edge_msg<1,1,1> for the 16-bit engine): a workgroup takes
// 128 consecutive rows of a flat edge list (i, j, code, radial per row; A_i gathered per row), and the gated messages go out row-major in
// fp32 instead of being summed per node.  No coordinate MLP (layer 0 is never the last layer on this path).
template <int ROWS> __global__ __launch_bounds__(256, 1) void k_edge_f32m(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ws = reinterpret_cast<float *>(smem);              // [2][32][256]
    float *Ms = Ws + 2 * FM_W_FLOATS;                         // [2][32][129]
    float *s_part = Ms + 2 * FM_M_FLOATS;                     // [4 waves][256] column sums of a tile
    float *s_cp = s_part + 4 * 256;                           // [4 waves][4] coordinate partials
    int *s_j = reinterpret_cast<int *>(s_cp + 16);            // [128]
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_j + 128);
    float *s_rad = reinterpret_cast<float *>(s_code + 128);
    int *s_i = reinterpret_cast<int *>(s_rad + 128);          // [128] row-list form: the row's own node

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int K = p.K;
#ifdef DFM_F32M_STAMP
    unsigned long long st[6]; int sti = 0;
#define FSTAMP() { __builtin_amdgcn_sched_barrier(0); st[sti++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define FSTAMP()
#endif
    FSTAMP()
    const long long ntask = (long long)p.B * p.nodes;
    const long long task0 = 2ll * blockIdx.x;
    // node of every 64-row half of the workgroup tile (the second may not exist: odd task count)
    auto node_of = [&](int half, int &b, int &i) -> bool {
        const long long t = task0 + half;
        const bool valid = t < ntask;
        const long long tt = valid ? t : task0;
        b = (int)(tt / p.nodes); i = p.node0 + (int)(tt % p.nodes);
        return valid;
    };
    uint32_t n_rows = 0;
    const uint32_t row_base = 128u * blockIdx.x;
    if constexpr (ROWS) {
        n_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.n_rows_dev ? *p.n_rows_dev : p.n_rows));
        if (row_base >= n_rows) return;      // the launch is sized for the list's capacity
        if (tid < 128) {      // rows past the end repeat the last row (valid addresses, finite values; gated off and never stored)
            const uint32_t r = row_base + (uint32_t)tid;
            const uint4 rec = p.rows[r < n_rows ? r : n_rows - 1u];
            s_i[tid] = (int)rec.x; s_j[tid] = (int)rec.y; s_code[tid] = rec.z; s_rad[tid] = __uint_as_float(rec.w);
        }
    } else if (tid < 128) {
        int b, i;
        const bool valid = node_of(tid >> 6, b, i);
        const int s = tid & 63;
        const bool v = valid && s < K;
        const size_t e = ((size_t)b * p.N + i) * K + (s < K ? s : 0);
        s_j[tid] = v ? p.edges[e] : i;
        s_code[tid] = v ? p.codes[e] : 0u;
        s_rad[tid] = v ? p.radial[e] : 0.f;
    }
    __syncthreads();
    // producer: thread = 4 rows (it * 32 + tid / 8) x 4 channels ((tid & 7) * 4) of every 32-channel chunk: eight lanes cover one row's
    // 128-byte line of each gathered operand (a lane per row would touch 64 lines per load instruction: the L1 then carries as many
    // cycles per chunk as the MFMAs).  LDS stores [k][row] with stride 129: bank (4 x + e + y) % 32 over lanes (x = tid & 7, y = tid / 8)
    // is conflict-free within each 32-lane pass.
    const int pr8 = tid >> 3, pch = (tid & 7) * 4;
    constexpr int NA = ROWS ? 4 : 2;      // A_i rows a thread reads per chunk: one per row (list) or one per node (two nodes per workgroup)
    uint32_t oA[NA], oBm[4], oT[5][4];
    float prad[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 32 + pr8;
        int nb = 0, ni = 0;
        if constexpr (ROWS) ni = s_i[row]; else node_of(row >> 6, nb, ni);
        const uint32_t ab = ROWS ? 0u : (uint32_t)((size_t)nb * p.ab_bstride);      // (the list's operands are the complex's own: no batch stride)
        if constexpr (ROWS) oA[it] = (uint32_t)ni * H + pch;
        else if ((it & 1) == 0) oA[it >> 1] = ab + (uint32_t)ni * H + pch;
        oBm[it] = ab + (uint32_t)s_j[row] * H + pch;
        const uint32_t code = s_code[row];
        oT[0][it] = (code & 63u) * H + pch; oT[1][it] = (40u + ((code >> 6) & 31u)) * H + pch;
        oT[2][it] = (64u + ((code >> 11) & 31u)) * H + pch; oT[3][it] = (88u + ((code >> 16) & 15u)) * H + pch;
        oT[4][it] = (100u + ((code >> 20) & 127u)) * H + pch;
        prad[it] = s_rad[row];
    }
    float pre_max = 0.f, acc_max = 0.f;      // range telemetry (p.range)

    float4 oa[NA], ob[4], ot[5][4], ow;      // operands of the m1 chunk in flight (4 rows x 4 channels)
    // weight chunk c of Wt ([256 k][256 n] fp32) straight from global memory into LDS buffer `buf` (global_load_lds_dwordx4: no
    // registers in between - 32 of them spilled otherwise; a wave's 64 lanes fill 1 KiB of contiguous LDS per instruction).  The
    // caller waits for vmcnt(0) before the barrier that publishes the buffer.
    auto fetch_w1 = [&](const float *Wt, int c, int buf, int q) {      // one of the eight 16-byte-per-lane pieces of a weight chunk
        const float4 *src = reinterpret_cast<const float4 *>(Wt + (size_t)c * FM_W_FLOATS) + tid;
        char *dst = reinterpret_cast<char *>(Ws + buf * FM_W_FLOATS) + wave * 1024;
        __builtin_amdgcn_global_load_lds(src + q * 256, (__attribute__((address_space(3))) void *)(dst + q * 4096), 16, 0, 0);
    };
    auto fetch_w = [&](const float *Wt, int c, int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) fetch_w1(Wt, c, buf, q);
    };
    auto wait_w = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // the 29 operand loads of a chunk, one per call (j = 0 .. 28): w_r, A_i of the two nodes, then per row Bm_j and five table rows
    constexpr int NOPS = 1 + NA + 24;      // operand loads of a chunk
    auto fetch_op1 = [&](int c, int j) {
        const int k0 = c * 32;
        if (j == 0) ow = *reinterpret_cast<const float4 *>(p.w_r + k0 + pch);
        else if (j <= NA) oa[j - 1] = *reinterpret_cast<const float4 *>(p.A + oA[j - 1] + k0);
        else {
            const int it = (j - 1 - NA) / 6, q = (j - 1 - NA) % 6;
            if (q == 0) ob[it] = *reinterpret_cast<const float4 *>(p.Bm + oBm[it] + k0);
            else ot[q - 1][it] = *reinterpret_cast<const float4 *>(p.T + oT[q - 1][it] + k0);
        }
    };
    auto fetch_ops = [&](int c) {
#pragma unroll
        for (int j = 0; j < NOPS; ++j) fetch_op1(c, j);
    };
    // edge_mlp.0 + SiLU (egnn.py:95-101) of element e of this thread's row `it`, same association as k_edge_f32; in two slices so that
    // it can be laid between MFMAs (a slice must stay below the 64 cycles an MFMA occupies the pipe)
    float pre_e = 0.f;
    auto build_slice = [&](int buf, int it, int e, int part) {
        if (part == 0) {
            const float4 a4 = oa[ROWS ? it : it >> 1], b4 = ob[it];
            const float a = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w)), bm = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
            const float w = e == 0 ? ow.x : (e == 1 ? ow.y : (e == 2 ? ow.z : ow.w));
            float pre = a + bm;
            pre += w * prad[it];
#pragma unroll
            for (int q = 0; q < 5; ++q) pre += e == 0 ? ot[q][it].x : (e == 1 ? ot[q][it].y : (e == 2 ? ot[q][it].z : ot[q][it].w));
            pre_e = pre;
        } else {
            pre_max = fmaxf(pre_max, fabsf(pre_e));
            Ms[buf * FM_M_FLOATS + (pch + e) * FM_LD + it * 32 + pr8] = silu_f32m(pre_e);
        }
    };
    auto build_row = [&](int buf, int it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { build_slice(buf, it, e, 0); build_slice(buf, it, e, 1); }
    };
    typedef float f32x16v __attribute__((ext_vector_type(16)));
    // 128 MFMAs of one chunk.  `slot(m)` runs after MFMA m = kk * 8 + nt (a few instructions each: requests and producer arithmetic of
    // the NEXT chunk ride in the shadow of this chunk's MFMAs; a wave issues in order, so whatever sits between two MFMAs must stay
    // below the 64 cycles one of them occupies the pipe).
    // The nine LDS reads of k-step kk + 1 go out right after the FIRST MFMA of k-step kk: hipcc waits for them with lgkmcnt(0) (it does
    // not count past LDS-DMA), so a wait placed directly behind a read drains the matrix pipe (~130 cycles of LDS latency); behind
    // seven more MFMAs the data is long there and the wait is free.  Left to itself the scheduler sinks every read to just before its
    // use (profiles/r04_fp32_engine.txt).
    auto mfma_chunk = [&](int buf, f32x16v (&acc)[8], auto slot) {
        const float *Mb = Ms + buf * FM_M_FLOATS + wave * 32 + l31 + h * FM_LD, *Wb = Ws + buf * FM_W_FLOATS + l31 + h * 256;
        float a_c = Mb[0], b_c[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) b_c[nt] = Wb[nt * 32];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a_n = 0.f, b_n[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[0], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < 16) {
                a_n = Mb[(2 * kk + 2) * FM_LD];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) b_n[nt] = Wb[(2 * kk + 2) * 256 + nt * 32];
            }
            slot(kk * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 1; nt < 8; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[nt], acc[nt], 0, 0, 0);
                slot(kk * 8 + nt);
                __builtin_amdgcn_sched_barrier(0);
            }
            a_c = a_n;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) b_c[nt] = b_n[nt];
        }
    };
    auto no_slot = [](int) {};

    f32x16v acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    fetch_w(p.W2t, 0, 0); fetch_ops(0);
#pragma unroll
    for (int it = 0; it < 4; ++it) build_row(0, it);
    wait_w();
    __syncthreads();
    FSTAMP()
#pragma unroll 1
    for (int c = 0; c < 7; ++c) {
        const int nb = (c + 1) & 1;
        // slots 1 .. 35 (first k-steps): the next chunk's 8 weight pieces and 27 operand loads, one per MFMA; slots 64 .. 127: its 16
        // elements in two slices each, one slice after every second MFMA
        mfma_chunk(c & 1, acc, [&](int m) {
            if (m >= 1 && m <= 8) fetch_w1(p.W2t, c + 1, nb, m - 1);
            else if (m >= 9 && m < 9 + NOPS) fetch_op1(c + 1, m - 9);
            else if (m >= 64 && (m & 1)) { const int j = (m - 64) >> 1; build_slice(nb, j >> 3, (j >> 1) & 3, j & 1); }
        });
        wait_w();
        __syncthreads();
    }
    mfma_chunk(1, acc, no_slot);
    __syncthreads();
    FSTAMP()

    // ---- epilogue of the wave's 32 x 256 tile: lane owns columns nt*32 + l31, rows rowof(r) = (r & 3) + 8 (r >> 2) + 4 h
    int wb = 0, wi = 0;
    const bool wvalid = ROWS ? true : node_of(wave >> 1, wb, wi);
    const int mt = wave & 1;
    const bool do_coord_wg = !ROWS && p.last != 0;
    if (do_coord_wg) fetch_w(p.Wc1t, 0, 0);      // flies under the epilogue (every wave is past the last chunk's reads: barrier above)
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const float bias = p.b2[nt * 32 + l31], av = p.att_w[nt * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = acc[nt][r] + bias;
            acc_max = fmaxf(acc_max, fabsf(x));
            const float m = silu_f32m(x);       // edge_mlp.2 + SiLU
            acc[nt][r] = m;
            part[r] = fmaf(m, av, part[r]);
        }
    }
    {   // attention gate (egnn.py:102-104): computed by the lane that ends up with the row's sum, handed back by ds_bpermute
        const float logit = half_reduce_scatter(part, lane);
        const int rs_j = rs_index(lane), rs_row = (rs_j & 3) + 8 * (rs_j >> 2) + 4 * h, bp_base = (lane & 32) * 4;
        const bool live = ROWS ? row_base + (uint32_t)(wave * 32 + rs_row) < n_rows : (wvalid && mt * 32 + rs_row < K);
        const float gate = live ? sigmoid_exact(logit + p.att_b) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int src = ((r & 8) >> 3) | ((r & 4) >> 1) | ((r & 2) << 1) | ((r & 1) << 3);
            part[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_base + src * 4, __builtin_bit_cast(int, gate)));
        }
    }
    if constexpr (ROWS) {      // gated messages, row-major: a store instruction writes 128 contiguous bytes of two rows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = row_base + (uint32_t)(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
            if (row < n_rows) {
                float *out = p.rows_out32 + (size_t)row * H + l31;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) out[nt * 32] = acc[nt][r] * part[r];
            }
        }
        if (p.range) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                pre_max = fmaxf(pre_max, __shfl_xor(pre_max, m, 64));
                acc_max = fmaxf(acc_max, __shfl_xor(acc_max, m, 64));
            }
            if (lane == 0) { atomicMax(p.range, __float_as_uint(pre_max)); atomicMax(p.range + 1, __float_as_uint(acc_max)); }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[nt][r] *= part[r]; cs += acc[nt][r]; }      // gated messages; unsorted_segment_sum over the tile's rows
        cs += __shfl_xor(cs, 32, 64);
        if (h == 0) s_part[wave * 256 + nt * 32 + l31] = cs;
    }
    __syncthreads();
    if (mt == 0 && wvalid) {
        float *out = p.agg + ((size_t)wb * p.N + wi) * H;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q * 64 + lane] = s_part[wave * 256 + q * 64 + lane] + s_part[(wave + 1) * 256 + q * 64 + lane];
    }
    if (p.range) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            pre_max = fmaxf(pre_max, __shfl_xor(pre_max, m, 64));
            acc_max = fmaxf(acc_max, __shfl_xor(acc_max, m, 64));
        }
        if (lane == 0) { atomicMax(p.range, __float_as_uint(pre_max)); atomicMax(p.range + 1, __float_as_uint(acc_max)); }
    }
    FSTAMP()
#ifdef DFM_F32M_STAMP
    if (blockIdx.x == gridDim.x / 2 + 7 && tid == 0)
        printf("f32m stamps (cycles of a 100 MHz clock x ~24): prologue %llu  K loop %llu  epilogue %llu\n", st[1] - st[0], st[2] - st[1], st[3] - st[2]);
#endif
    if (!do_coord_wg) return;
    {   // any ligand node in this workgroup?  (workgroup-uniform: the second K loop has barriers)
        int b0, i0, b1, i1;
        node_of(0, b0, i0);
        const bool v1 = node_of(1, b1, i1);
        if (!(i0 >= p.R || (v1 && i1 >= p.R))) return;
    }
    // ---- coord_model (egnn.py:118-137): second contraction, A operand = the gated message tile, chunk c = the wave's acc[c]
    f32x16v cacc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[nt][r] = 0.f;
    auto stage_gated = [&](int buf, const f32x16v &g) {      // [k = l31][row = wave*32 + rowof(r)]: banks (k + row) % 32
        float *dst = Ms + buf * FM_M_FLOATS + l31 * FM_LD + wave * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2)] = g[r];
    };
    stage_gated(0, acc[0]);
    wait_w();
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) { fetch_w(p.Wc1t, c + 1, (c + 1) & 1); stage_gated((c + 1) & 1, acc[(c + 1) & 7]); }
        mfma_chunk(c & 1, cacc, no_slot);
        wait_w();
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const float bias = p.bc1[nt * 32 + l31], w2 = p.wc2[nt * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = fmaf(silu_f32m(cacc[nt][r] + bias), w2, part[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = half_sum_dpp(part[r]);      // every lane of the half holds the row's sum
    float w = part[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) w = l31 == r ? part[r] : w;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    const int lrow = mt * 32 + (l31 & 3) + 8 * ((l31 >> 2) & 3) + 4 * h;      // one lane per row: lanes l31 < 16 take register row l31
    const float4 *ca = p.ca4 + (size_t)wb * p.N;
    const float4 xi = ca[wi];
    if (l31 < 16 && lrow < K) {
        const float4 xj = ca[s_j[(wave >> 1) * 64 + lrow]];
        w = fminf(fmaxf(w, -2.0f), 2.0f);                                       // clamp_(-2, 2)
        const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz + 1e-8f) + 1.0f;   // coord2radial, normalize=True
        c0 = dx / nrm * w; c1 = dy / nrm * w; c2 = dz / nrm * w;
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    if (lane == 0) { s_cp[wave * 4 + 0] = c0; s_cp[wave * 4 + 1] = c1; s_cp[wave * 4 + 2] = c2; }
    __syncthreads();
    if (mt == 0 && wvalid && wi >= p.R && lane < 3) {
        const float a = (s_cp[wave * 4 + lane] + s_cp[(wave + 1) * 4 + lane]) / (float)(K > 1 ? K : 1);      // unsorted_segment_mean
        const float xd = lane == 0 ? xi.x : (lane == 1 ? xi.y : xi.z);
        p.fout[((size_t)wb * p.L + (wi - p.R)) * 3 + lane] = (xd + a) - xd;      // f = pos_out - r
    }
}

// -------------------------------------------------------------------------------------------------
// Message kernel (edge_mlp + attention gate + segment sum; on the last layer also the store of the gated messages of the ligand
// nodes in the A-fragment order k_edge_coord reads).  Software-pipelined ACROSS tiles: with a per-tile prologue (wait for the edge
// indices, gather chunk 0, wait, build it, gather chunk 1, wait - about 2.2 k of the 32 k cycles of a tile,
// profiles/r02_exp_edge_phases.txt) the last chunk of a tile also ran its MFMAs with no producer work beside them.  Here the chunk
// loop wraps around the tile boundary: chunk 6 requests chunk 0 of the NEXT tile, chunk 7 builds it into the free staging buffer
// and requests that tile's chunk 1, which flies under this tile's epilogue; the next tile starts straight at its first MFMA.  The
// wave walks its (node, tile) sequence with a one-tile lookahead; after the last tile the lookahead repeats that tile (valid
// addresses, results never used) so that the loop body has no tail variant.
// AW16: A_i comes as fp16 (one 16-byte load per chunk instead of two; bf16 operands only).  w_r stays fp32: its product with the
// radial |x_i - x_j|^2 (thousands of A^2) is the one large term of the pre-activation, and an fp16 w_r moved the worst force
// deviation of the bf16 engine from 7.2e-3 to 8.8e-3
//
// ROWS = 1 (layer 0 behind the per-complex message table, see k_l0_gather): the same pipeline over a flat LIST of edges instead of the
// K edges of a node - the intra-chain edges whose table entry does not apply and every inter-chain edge of an evaluation, or all
// intra-chain pairs of the complex when the table is built.  A task is one 32-row tile of the list, every row carries its own
// (i, j, code, radial), A_i is gathered per row like Bm_j, and instead of the segment sum the gated messages are stored row-major
// as fp16 (S * gate * m, the unit of the last layer's message buffer).  Rows are independent in the contraction, so a row's result
// does not depend on which other rows share its tile.
template <int F16, int AW16, int ROWS = 0> __global__ __launch_bounds__(MSG_WAVES * 64) void k_edge_msg(EdgeKArgs p)
{
    static_assert(!ROWS || (F16 && AW16), "the row-list form exists for the shipped 16-bit plan only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *Wf = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform BY ANALYSIS too: the tile walk below stays in SGPRs
    char *stage = smem + LDS_WF_BYTES + (wave % EDGE_WAVES) * LDS_STAGE_BYTES;
    const int h = lane >> 5, l31 = lane & 31;
    uint32_t n_rows = 0, n_row_tiles = 0;
    if constexpr (ROWS) {
        n_rows = __builtin_amdgcn_readfirstlane(p.n_rows_dev ? *p.n_rows_dev : p.n_rows);
        n_row_tiles = (n_rows + 31u) >> 5;
        if (blockIdx.x >= n_row_tiles) return;      // tasks go wave-major (wave w of workgroup g starts at tile w * grid + g): nothing for this
                                                    // workgroup - leave before the 128 KiB weight fill (the launch is sized for the capacity)
    }
    // (the 128 KiB weight fill of the workgroup happens below, AFTER the wave has requested its first tile's operands)

    // XCD-aware task order (speed only): workgroup g runs on XCD g % 8; give every XCD whole trajectories so that the gathered
    // rows of Bm stay in that XCD's L2; few trajectories (B < 8): each is split over several XCDs
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    // A wave task is a node (its ceil(K / 32) tiles in sequence, segment sum in registers) or - p.split, launches too small to
    // give every wave a few nodes - a single tile, whose partial segment sum is added to the pre-zeroed agg atomically.  Both
    // forms produce bitwise the same agg: per tile x_t = inv_s * (sum over its rows), agg = x_0 + x_1 (two addends: commutative).
    const int K = p.K, ntile = (K + 31) >> 5;
    const bool split = ROWS || p.split != 0;
    const int NT = split ? p.nodes * ntile : p.nodes;
    const int nsplit = p.B >= 8 ? 1 : (8 + p.B - 1) / p.B;
    const int NTc = ROWS ? 1 : (NT + nsplit - 1) / nsplit;
    const int U = p.B * nsplit;
    const int nb_x = U > xcd ? (U - xcd + 7) >> 3 : 0;
    const unsigned ntask = ROWS ? n_row_tiles : (unsigned)nb_x * (unsigned)NTc;
    const unsigned tstride = ROWS ? gridDim.x * MSG_WAVES : (unsigned)wg_per_xcd * MSG_WAVES;
    auto task_tile = [&](unsigned tt, int &b, int &i, int &mt) -> bool {      // first tile of task tt
        if constexpr (ROWS) { b = 0; i = (int)tt; mt = 0; return true; }      // row-list form: "node" i = the tile of the list
        const unsigned tq = tt / (unsigned)NTc, tr = tt - tq * (unsigned)NTc;
        const int u = xcd + 8 * (int)tq;
        b = __builtin_amdgcn_readfirstlane(u / nsplit);
        const int idx = __builtin_amdgcn_readfirstlane((u % nsplit) * NTc + (int)tr);
        const int il = split ? idx / ntile : idx;
        mt = split ? idx - il * ntile : 0;
        i = p.node0 + il;
        return idx < NT;
    };
    auto next_task = [&](unsigned &tt, int &b, int &i, int &mt) -> bool {      // first valid task at or after tt
        while (tt < ntask) {
            if (task_tile(tt, b, i, mt)) return true;
            tt += tstride;
        }
        return false;
    };
    Frag onef;
    onef.u = make_uint4(h == 0 ? (F16 ? 0x3c003c00u : 0x3f803f80u) : 0u, 0u, 0u, 0u);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(p.T2b), rs_w = make_rsrc(p.w_r);
    const __amdgpu_buffer_rsrc_t rs_e = make_rsrc(ROWS ? (const void *)p.rows : (const void *)p.edges), rs_c = make_rsrc(p.codes), rs_r = make_rsrc(p.radial);
    const int r16 = lane >> 2, c4 = lane & 3;
    const uint32_t oc4 = c4 * 32;
    // epilogue constants: the row (inside a tile) whose sum this lane ends up with in half_reduce_scatter; byte address of lane 0 of
    // this lane's half for ds_bpermute
    const int rs_j = rs_index(lane), rs_row = (rs_j & 3) + 8 * (rs_j >> 2) + 4 * h, bp_base = (lane & 32) * 4;

    unsigned tt = ROWS ? (unsigned)wave * gridDim.x + blockIdx.x
                       : (unsigned)wave * (unsigned)wg_per_xcd + (unsigned)slot;      // wave-major: a launch with fewer tasks than waves spreads over ALL workgroups
                                                                               // (a few waves each, a SIMD to themselves) instead of filling the first ones
    int b = 0, i = 0, mt = 0;
    const bool has_task = next_task(tt, b, i, mt);      // (a wave without a task still helps to fill the weights and meets the barrier)
    // Dynamic tasks (r06).  The two waves of a SIMD do not run at the same speed: issue arbitration favours the OLDER wave, and the trace
    // of a C3 launch (tools/edge_trace.py, profiles/r06_edge_trace.txt) shows waves 0..3 of a workgroup finishing a tile every 21.4 k cycles
    // and waves 4..7 every 30.0 k.  With a fixed stride every wave gets the same number of nodes, so the older half is done after ~70 % of
    // the launch and the younger half finishes alone, one wave per SIMD.  Instead the WORKGROUP keeps its fixed share of the XCD's task
    // list - tasks slot, slot + wg_per_xcd, slot + 2 wg_per_xcd ... exactly the set its eight waves walk with the fixed stride, so the
    // XCD-aware order (an XCD walks whole trajectories, all its workgroups at the same pace) is untouched - and its waves take them in
    // order from a per-workgroup counter: a wave's first task is the static one (position = wave), every later position is one returning
    // atomic by lane 0, issued at the START of the epilogue of the tile before the task's last tile and read at the END of that epilogue
    // (the vector-memory counter completes in order: there it delays no gather, the only loads in flight are older).  Eight waves per
    // counter: a first version with one counter per XCD lost 3 - 22 % on launches with few tasks per wave, where all 256 waves of the XCD
    // reach their fetch at the same moment and queue on one address (profiles/r06_edge_trace.txt).
    // Results cannot depend on which wave runs a node: a task writes its own rows of agg / mbuf, nothing else.  The counters reset
    // themselves: every wave that had a task counts itself out on a second per-workgroup word when it leaves, and the last one out
    // stores zeros into both.
    // (AW16 = 0 - fp32 A_i: two more live registers per chunk - spills one register to scratch with the extra state: fixed stride there)
    const bool dyn = AW16 != 0 && !ROWS && !split && nsplit == 1 && p.task_ctr != nullptr;
    uint32_t dyn_next = ~0u;      // index of the task after the current one, valid from the end of the epilogue that fetched it
    auto fetch_task = [&]() -> uint32_t {      // per-lane result of lane 0's atomic; consumed through readfirstlane
        uint32_t v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(p.task_ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto dyn_index = [&](uint32_t pos) -> uint32_t { return (uint32_t)slot + (uint32_t)wg_per_xcd * ((uint32_t)MSG_WAVES + pos); };
    if (dyn && has_task && ntile == 1) dyn_next = dyn_index((uint32_t)__builtin_amdgcn_readfirstlane((int)fetch_task()));      // one-tile tasks: needed at once

    // raw edge data of the tile in lookahead (rows past K read the node's last edge and are masked in set_tile)
    int jqn[2]; uint32_t codeqn[2]; float radqn[2];
    int iqn[2] = {0, 0};      // row-list form: the row's own node
    auto load_idx = [&](int tb, int ti, int tm) {
        if constexpr (ROWS) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t s = (uint32_t)ti * 32u + (uint32_t)(q * 16 + r16);
                const u32x4v rec = __builtin_amdgcn_raw_buffer_load_b128(rs_e, (int)((s < n_rows ? s : n_rows - 1u) * 16u), 0, 0);
                iqn[q] = (int)rec.x; jqn[q] = (int)rec.y; codeqn[q] = rec.z; radqn[q] = __uint_as_float(rec.w);
            }
            return;
        }
        const uint32_t ebase = ((uint32_t)tb * (uint32_t)p.N + (uint32_t)ti) * (uint32_t)K;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = tm * 32 + q * 16 + r16;
            const uint32_t off = (ebase + (uint32_t)(s < K ? s : K - 1)) * 4u;
            jqn[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_e, (int)off, 0, AUX_STREAM);
            codeqn[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)off, 0, AUX_STREAM);
            radqn[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)off, 0, AUX_STREAM));
        }
    };
    // producer state: gather offsets / resources of the tile whose operands are being REQUESTED, radial of the tile being BUILT
    uint32_t obm[2], ot0[2], ot1[2];
#if !DFM_TAB_MERGE
    uint32_t ot2[2];
#endif
    uint32_t oa[2] = {0u, 0u};      // row-list form: byte offset of the row's own A_i
    float radq[2], radq_nx[2];
    __amdgpu_buffer_rsrc_t rs_bm = rs_t, rs_a = rs_t;
    auto set_tile = [&](int tb, int ti, int tm) {      // from jqn / codeqn / radqn of that tile
        const size_t ab = (size_t)tb * p.ab_bstride;
        rs_bm = make_rsrc(p.Bmb + ab);
        if constexpr (ROWS) rs_a = make_rsrc(p.Ah);
        else rs_a = AW16 ? make_rsrc(p.Ah + ab + (size_t)ti * H) : make_rsrc(p.A + ab + (size_t)ti * H);
#pragma unroll
        for (int q = 0; q < 2; ++q) {      // masked rows (>= K; row-list form: past the end of the list): self edge, zero features -> finite values, gate forced to 0
            const bool v = ROWS ? (uint32_t)ti * 32u + (uint32_t)(q * 16 + r16) < n_rows : tm * 32 + q * 16 + r16 < K;
            const int j = v ? jqn[q] : (DFM_EDGE_PAD0 || ROWS ? 0 : ti);
            const uint32_t code = v ? codeqn[q] : 0u;
            radq_nx[q] = v ? radqn[q] : 0.f;
            if constexpr (ROWS) oa[q] = (uint32_t)(v ? iqn[q] : 0) * (H * 2) + c4 * 16;
            obm[q] = (uint32_t)j * (H * 2) + c4 * 16;
#if DFM_TAB_MERGE
            ot0[q] = ((((code >> 6) & 31u) * 24u + ((code >> 11) & 31u)) * 12u + ((code >> 16) & 15u)) * (H * 2) + c4 * 16;
            ot1[q] = (6912u + ((code >> 20) & 127u) * 40u + (code & 63u)) * (H * 2) + c4 * 16;
#else
            ot0[q] = (((code >> 6) & 31u) * 24u + ((code >> 11) & 31u)) * (H * 2) + c4 * 16;
            ot1[q] = (576u + ((code >> 16) & 15u) * 40u + (code & 63u)) * (H * 2) + c4 * 16;
            ot2[q] = (1056u + ((code >> 20) & 127u)) * (H * 2) + c4 * 16;
#endif
#ifdef DFM_EDGE_SAMEROW      // diagnostic builds: rows gather row 0 (wrong results; loads issued, L1 hits): 1 everything, 2 Bm only, 3 tables only
            if (DFM_EDGE_SAMEROW != 3) obm[q] = (obm[q] & 1u) + c4 * 16;
            if (DFM_EDGE_SAMEROW != 2) {
                ot0[q] = (ot0[q] & 1u) + c4 * 16; ot1[q] = (ot1[q] & 1u) + c4 * 16;
#if !DFM_TAB_MERGE
                ot2[q] = (ot2[q] & 1u) + c4 * 16;
#endif
            }
#endif
        }
    };
    float4 a0, a1, w0, w1;      // fp32 A_i / w_r of this lane's 8 channels (AW16: a0 holds the 8 fp16 values of A_i, a1 unused)
#ifdef DFM_EDGE_NOGATHER      // diagnostic build: the gathered operands are whatever the registers hold (wrong results, no instruction
                              // issued for them) - the kernel's time with no gather in it
#define FAKE4(v) asm volatile("" : "=v"((v).x), "=v"((v).y), "=v"((v).z), "=v"((v).w))
    auto gather_chunk = [&](int) { FAKE4(a0); FAKE4(w0); FAKE4(w1); if constexpr (!AW16) FAKE4(a1); };
    auto gather = [&](int, int, RawP &r) { FAKE4(r.bm); FAKE4(r.t0); FAKE4(r.t1); };
#undef FAKE4
#else
    auto gather_chunk = [&](int c) {
        if constexpr (ROWS) { a0 = bload16f(rs_a, oa[0], c * 64); a1 = bload16f(rs_a, oa[1], c * 64); }      // a row of A per pass
        else if constexpr (AW16) a0 = DFM_EDGE_A_NT ? bload16f_stream(rs_a, c4 * 16, c * 64) : bload16f(rs_a, c4 * 16, c * 64);
        else { a0 = bload16f_stream(rs_a, oc4, c * 128); a1 = bload16f_stream(rs_a, oc4, c * 128 + 16); }
        w0 = bload16f(rs_w, oc4, c * 128); w1 = bload16f(rs_w, oc4, c * 128 + 16);
    };
    auto gather = [&](int c, int q, RawP &r) {
        r.bm = bload16(rs_bm, obm[q], c * 64);
        r.t0 = bload16(rs_t, ot0[q], c * 64);
        r.t1 = bload16(rs_t, ot1[q], c * 64);
#if !DFM_TAB_MERGE
        r.t2 = bload16(rs_t, ot2[q], c * 64);
#endif
    };
#endif
    H8 pt[2], pbm;
    f2 pv[2][4];
    f2 pex[2];      // DFM_EDGE_SKEW: exp2 of the pair in flight, issued at the end of the pair's pre-activation slice
    Frag pf[2];
    // The producer arithmetic of one pass (8 channels of one row per lane) cut into eight slices, so that it can be laid between
    // MFMAs in program order: even slice 2e = pre-activation of channel pair e, odd slice 2e + 1 = its SiLU + conversion; slice 7
    // also stores the finished 16 bytes.
    auto slice = [&](int q, int k, const RawP &r, char *buf) {
        const int e = k >> 1;
        if (k == 0) {
            H8 t1;
            pt[q].u = r.t0; t1.u = r.t1; pbm.u = r.bm;
#if !DFM_TAB_MERGE
            H8 t2;
            t2.u = r.t2;
#endif
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                pt[q].h[x] = __hadd2(pt[q].h[x], t1.h[x]);
#if !DFM_TAB_MERGE
                pt[q].h[x] = __hadd2(pt[q].h[x], t2.h[x]);
#endif
                pt[q].h[x] = __hadd2(pt[q].h[x], pbm.h[x]);      // Bm_j joins the table rows in the packed fp16 sum (one add for two channels)
            }
        }
        if ((k & 1) == 0) {
            const f2 wv = e == 0 ? (f2){w0.x, w0.y} : (e == 1 ? (f2){w0.z, w0.w} : (e == 2 ? (f2){w1.x, w1.y} : (f2){w1.z, w1.w}));
            if constexpr (AW16) {      // w * radial (fp32) + a (fp16), one v_fma_mix_f32 per channel
                const float4 &aq = (ROWS && q == 1) ? a1 : a0;
                const uint32_t ah = __float_as_uint(e == 0 ? aq.x : (e == 1 ? aq.y : (e == 2 ? aq.z : aq.w)));
                pv[q][e] = (f2){fma_half_lo(wv.x, radq[q], ah), fma_half_hi(wv.y, radq[q], ah)};
            } else {
                const f2 rad2 = {radq[q], radq[q]};
                const f2 av = e == 0 ? (f2){a0.x, a0.y} : (e == 1 ? (f2){a0.z, a0.w} : (e == 2 ? (f2){a1.x, a1.y} : (f2){a1.z, a1.w}));
                pv[q][e] = wv * rad2 + av;
            }
            pv[q][e] = add_half2(pv[q][e], pt[q].h[e]);
            if constexpr (DFM_EDGE_SKEW && F16) pex[q] = (f2){ko_exp2<DFM_EDGE_KO & 2>(pv[q][e].x), ko_exp2<DFM_EDGE_KO & 2>(pv[q][e].y)};
        } else {
            if constexpr (F16) {
                // fp16 operand from ONE v_cvt_pkrtz per pair: truncation saturates for free (no separate clamp), and the reciprocal
                // carries a (1 + 2^-12) bias - its addend and multiplier are (1 - 2^-12) instead of 1 - so that the truncated value is
                // within (-0.625, 0.375) ulp of the exact one: round-to-nearest-like (RMS 0.315 vs 0.289 ulp).  Same-box A/B with the
                // packed fp16 sum above: 2.211 vs 2.251 ms per launch, deviations unchanged (profiles/r03_exp_edge_trims.txt)
                const f2 x = pv[q][e];
                f2 ex;
                if constexpr (DFM_EDGE_SKEW) ex = pex[q];
                else ex = (f2){ko_exp2<DFM_EDGE_KO & 2>(x.x), ko_exp2<DFM_EDGE_KO & 2>(x.y)};
                ex = ex * (f2){0.999755859375f, 0.999755859375f} + (f2){0.999755859375f, 0.999755859375f};
                const f2 r = {ko_rcp<DFM_EDGE_KO & 2>(ex.x), ko_rcp<DFM_EDGE_KO & 2>(ex.y)};
                const f2 m = x * r;
                (&pf[q].u.x)[e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(m.x, m.y));
            } else {
                const f2 m = silu2s(pv[q][e]);
                pf[q].b[2 * e] = (__bf16)m.x; pf[q].b[2 * e + 1] = (__bf16)m.y;
            }
            if (k == 7) {
                const int row = q * 16 + r16;
                *reinterpret_cast<uint4 *>(buf + ((c4 * 32 + (row ^ (4 * c4))) << 4)) = pf[q].u;
#if DFM_EDGE_HALF      // (rows 16..31 belong to the partner wave of the pair form: filled with a copy so that the wrong results stay finite)
                *reinterpret_cast<uint4 *>(buf + ((c4 * 32 + ((row + 16) ^ (4 * c4))) << 4)) = pf[q].u;
#endif
            }
        }
    };
    auto compute_store = [&](int q, const RawP &r, char *buf) {
#pragma unroll
        for (int k = 0; k < 8; ++k) slice(q, k, r, buf);
    };
#ifdef DFM_EDGE_STAMP
    unsigned long long st_t[5] = {0, 0, 0, 0, 0}, st_prev = 0;
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _n = __builtin_amdgcn_s_memtime(); st_t[k] += _n - st_prev; st_prev = _n; __builtin_amdgcn_sched_barrier(0); }
#define STAMP0() { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(k)
#define STAMP0()
#endif
#ifdef DFM_EDGE_TRACE      // diagnostic build: absolute s_memtime of tile start / epilogue start of the first 64 tiles of every wave of workgroup 0
    int tr_n = 0;          // -> p.stamp[48 + wave * 130 ...] ([0] = HW_ID register: SIMD id in bits 5:4); tools/edge_trace.py
#define TRACE(w) { if (p.stamp && blockIdx.x == 0 && tr_n < 64) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                   if (lane == 0) p.stamp[48 + wave * 130 + 1 + 2 * tr_n + (w)] = _t; \
                   if ((w) == 0 && (tr_n == 0 || tr_n == 63) && lane == 0) p.stamp[wave * 2 + (tr_n ? 1 : 0)] = __builtin_amdgcn_s_memrealtime();      /* 100 MHz: the clock the kernel ran at */ \
                   __builtin_amdgcn_sched_barrier(0); } }
    if (p.stamp && blockIdx.x == 0 && lane == 0) p.stamp[48 + wave * 130] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#else
#define TRACE(w)
#endif

#if !defined(DFM_EDGE_STAMP) && !defined(DFM_EDGE_TRACE)
    // Profiled calls (DFM_F_PROFILE: p.stamp set): the shader clock this launch runs at.  The chip's power management holds the message kernel well below
    // the 2.4 GHz the peak figures assume (profiles/r06_clock.txt), so the bench reports the clock next to the fraction.  stamp[0] += shader cycles,
    // stamp[1] += 100 MHz ticks between this wave's start and its exit; one writer (launches of a handle are serialised on its stream).
    const bool clk_wave = p.stamp != nullptr && blockIdx.x == 0 && wave == 0;
    if (clk_wave && lane == 0) { p.stamp[2] = __builtin_amdgcn_s_memtime(); p.stamp[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // ---- the only prologue of the wave: first tile's chunk 0 built, its chunk 1 requested
    RawP r0, r1;
    if (has_task) {
        load_idx(b, i, mt);
        set_tile(b, i, mt);
        radq[0] = radq_nx[0]; radq[1] = radq_nx[1];
        gather_chunk(0);
        gather(0, 0, r0);
        if constexpr (!DFM_EDGE_HALF) gather(0, 1, r1);
    }
    // The workgroup's weight fragments, global -> LDS, with the first tile's index loads and gathers already in flight: a launch with
    // one round of tiles (small batches, the row-list launches) otherwise pays the two dependent round trips of its prologue AFTER
    // the 128 KiB fill instead of under it.
    for (int q = tid; q < LDS_WF_BYTES / 16; q += MSG_WAVES * 64) Wf[q] = p.Wf[q];
    __syncthreads();
    if (!has_task) return;
    if constexpr (DFM_EDGE_PRIO) { if (wave >= EDGE_WAVES / 2) __builtin_amdgcn_s_setprio(DFM_EDGE_PRIO); }      // the second wave of every SIMD
    compute_store(0, r0, stage); gather(1, 0, r0);
    if constexpr (!DFM_EDGE_HALF) { compute_store(1, r1, stage); gather(1, 1, r1); }
    gather_chunk(1);

    float colsum[MSG_NT];
#pragma unroll
    for (int nt = 0; nt < MSG_NT; ++nt) colsum[nt] = 0.f;
    const float *dot_v = p.att_w;

    while (true) {
        STAMP0();
        TRACE(0);
        // the tile after this one (lookahead); none left: this tile again, requested and built but never consumed
        unsigned ntt = tt;
        int nb = b, ni = i, nmt = mt + 1;
        bool have_next = true;
        if (split || nmt == ntile) {
            if (dyn) {
                ntt = dyn_next;
                have_next = ntt < ntask && task_tile(ntt, nb, ni, nmt);
            } else {
                ntt = tt + tstride;
                have_next = next_task(ntt, nb, ni, nmt);
            }
        }
        if (!have_next) { nb = b; ni = i; nmt = mt; }

        f32x16 acc[MSG_NT];
        Frag af0c, bq0c;      // DFM_EDGE_EARLYA: k-step 0's A fragment / first weight fragment of the next chunk, carried over the chunk boundary
        float dv[MSG_NT];
        uint32_t bp[MSG_NT];
        // one chunk: 16 MFMAs of chunk c; PRODUCE: the arithmetic of the next chunk (chunk 7: chunk 0 of the next tile), one slice after
        // every MFMA; the loads of the chunk after that go out at the slots DFM_EDGE_G0 / GC / G1; FIRST: opens the accumulators;
        // LAST: the epilogue's bias operand is requested instead of the (already built) staging being idle
        auto chunk = [&](int c, auto first, auto last) {
            char *bufc = stage + (c & 1) * 2048, *bufn = stage + ((c + 1) & 1) * 2048;
            const int cg = (c + 2) & 7;
            // (the staging area is wave-private and a wave's LDS instructions execute in order: the fragment reads below see the producer's
            // ds_write without waiting for it to complete - DFM_EDGE_NOFENCE keeps only the compiler from reordering them)
            if constexpr (DFM_EDGE_NOFENCE) asm volatile("" ::: "memory"); else wave_lds_fence();
            Frag af[2];
            constexpr bool carried = DFM_EDGE_EARLYA && !decltype(first)::value;      // k-step 0's fragments were requested at the end of the chunk before
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int un = ks * 2 + h;
                if (ks == 0 && carried) af[0] = af0c;
                else af[ks].u = *reinterpret_cast<const uint4 *>(bufc + ((un * 32 + (l31 ^ (4 * un))) << 4));
            }
            const uint4 *wq = Wf + (size_t)c * 16 * 64 + lane;
#if DFM_EDGE_HALF
            {
                Frag bq[2];
                bq[0].u = wq[0];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {      // 8 slots: k-step sl >> 2, n-tile sl & 3; one producer slice of the wave's ONE pass after each
                    if (sl < 7) bq[(sl + 1) & 1].u = wq[((((sl + 1) >> 2) * 8) + ((sl + 1) & 3)) * 64];
                    if constexpr (decltype(first)::value) {
                        if (sl < 4) acc[sl] = mfma16<F16>(af[0], bq[sl & 1], zero16);
                        else acc[sl & 3] = mfma16<F16>(af[1], bq[sl & 1], acc[sl & 3]);
                    } else {
                        acc[sl & 3] = mfma16<F16>(af[sl >> 2], bq[sl & 1], acc[sl & 3]);
                    }
                    if constexpr (decltype(last)::value) { if (sl < 4) bp[sl] = p.biasp[sl * 64 + lane]; }
                    slice(0, sl, r0, bufn);
                    if (sl == 3) gather(cg, 0, r0);
                    if constexpr (!decltype(last)::value || DFM_EDGE_DEFER < 1) { if (sl == 4) gather_chunk(cg); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#endif
            constexpr int BD = DFM_EDGE_BD;
            Frag bq[BD];
#pragma unroll
            for (int d = 0; d < BD - 1; ++d) { if (d == 0 && carried) bq[0] = bq0c; else bq[d].u = wq[d * 64]; }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
#if DFM_EDGE_KO & 8      // knock-out build (wrong results): ONE weight-fragment read per chunk instead of sixteen - what the 128 KiB of LDS reads per tile cost
                if (m + BD - 1 < 16) asm volatile("" : "+v"(bq[(m + BD - 1) % BD].u.x), "+v"(bq[(m + BD - 1) % BD].u.y), "+v"(bq[(m + BD - 1) % BD].u.z), "+v"(bq[(m + BD - 1) % BD].u.w));
#else
                if (m + BD - 1 < 16) bq[(m + BD - 1) % BD].u = wq[(m + BD - 1) * 64];
#endif
                auto do_slice = [&]() {
                    if constexpr (DFM_EDGE_ILV) { if (m & 1) slice(1, m >> 1, r1, bufn); else slice(0, m >> 1, r0, bufn); }      // passes interleaved slot by slot
                    else { if (m < 8) slice(0, m & 7, r0, bufn); else slice(1, m & 7, r1, bufn); }
                    if constexpr (DFM_EDGE_EARLYA && !decltype(last)::value) {
                        if (m == 15) {      // the next chunk is complete in bufn: its k-step 0 operands go out now, one MFMA (and the loop's back edge) ahead of their use
                            af0c.u = *reinterpret_cast<const uint4 *>(bufn + ((h * 32 + (l31 ^ (4 * h))) << 4));
                            bq0c.u = wq[16 * 64];
                        }
                    }
                };
                if constexpr (DFM_EDGE_ROT) do_slice();      // slice BEFORE the slot's MFMA: the fragment reads at the top of the chunk fly under slice 0
                if constexpr (decltype(first)::value) {
                    if (m < 8) acc[m] = mfma16<F16>(af[0], bq[m % BD], zero16);
                    else acc[m & 7] = mfma16<F16>(af[1], bq[m % BD], acc[m & 7]);
                } else {
                    acc[m & 7] = mfma16<F16>(af[m >> 3], bq[m % BD], acc[m & 7]);
                }
                if constexpr (decltype(last)::value) {
                    if (m < 8) bp[m] = p.biasp[m * 64 + lane];      // older than this chunk's gathers: the bias step does not wait for them
                }
                if constexpr (!DFM_EDGE_ROT) do_slice();
                if (m == DFM_EDGE_G0) gather(cg, 0, r0);
                if constexpr (!decltype(last)::value || DFM_EDGE_DEFER < 1) { if (m == DFM_EDGE_GC) gather_chunk(cg); }
                if constexpr (!decltype(last)::value || DFM_EDGE_DEFER < 2) { if (m == DFM_EDGE_G1) gather(cg, 1, r1); }
                if ((m + 1) % DFM_EDGE_SB == 0) __builtin_amdgcn_sched_barrier(0);
            }
        };
        chunk(0, std::true_type{}, std::false_type{});
        load_idx(nb, ni, nmt);            // lands long before chunk 6 needs it
#pragma unroll 1
        for (int c = 1; c < 6; ++c) chunk(c, std::false_type{}, std::false_type{});
        set_tile(nb, ni, nmt);            // requests switch to the next tile (this tile's last gathers went out in chunk 5)
        chunk(6, std::false_type{}, std::false_type{});     // builds chunk 7 of this tile, requests chunk 0 of the next
        STAMP(1);
        radq[0] = radq_nx[0]; radq[1] = radq_nx[1];
        chunk(7, std::false_type{}, std::true_type{});      // builds chunk 0 of the next tile, requests its chunk 1
#pragma unroll
        for (int nt = 0; nt < MSG_NT; ++nt) dv[nt] = dot_v[nt * 32 + l31];
        // bias k-step: acc += 1 * hi + 1 * lo (the accumulators were opened with C = 0)
#pragma unroll
        for (int nt = 0; nt < MSG_NT; ++nt) {
            Frag bb;
            bb.u = make_uint4(bp[nt], 0u, 0u, 0u);
            acc[nt] = mfma16<F16>(onef, bb, acc[nt]);
        }
        STAMP(2);
        TRACE(1);
#ifdef DFM_EDGE_TRACE
        ++tr_n;
#endif
        // dynamic tasks: the tile after this one is the last of its task -> its iteration will need the task after that one
        const bool fetch_now = dyn && have_next && nmt == ntile - 1;
        uint32_t fetched = 0;
        if (fetch_now) fetched = fetch_task();
        __builtin_amdgcn_sched_barrier(0);      // (the read of `fetched` stays at the END of the epilogue: hoisted next to the atomic it would expose its latency here)

        // ---- epilogue on the 32 x 256 tile: lane owns columns nt*32 + l31, rows rowof(r) = (r & 3) + 8 (r >> 2) + 4 h
        float part[16];
        {
            f2 part2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) part2[q] = (f2){0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < MSG_NT; ++nt) {
                const f2 vv = {dv[nt], dv[nt]};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f2 m;
                    if constexpr (DFM_EDGE_KO & 4) m = (f2){acc[nt][2 * q], acc[nt][2 * q + 1]} * (f2){0.5f, 0.5f};      // knock-out: no transcendentals
                    else m = silu2s((f2){acc[nt][2 * q], acc[nt][2 * q + 1]});
                    acc[nt][2 * q] = m.x; acc[nt][2 * q + 1] = m.y;
                    part2[q] = m * vv + part2[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { part[2 * q] = part2[q].x; part[2 * q + 1] = part2[q].y; }
        }
        {
            // attention gate sigmoid(logit) = 1 / (1 + exp2(S logit)) (att_b pre-scaled), rows >= K gated off: computed by the lane
            // that holds the row's sum, then handed to every lane of the half (ds_bpermute: no VALU issue slot)
            const float logit = half_reduce_scatter(part, lane);
            const int rown = mt * 32 + rs_row;
            const bool row_live = ROWS ? (uint32_t)i * 32u + (uint32_t)rs_row < n_rows : rown < K;
            const float gate = row_live ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(logit + p.att_b)) : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int src = ((r & 8) >> 3) | ((r & 4) >> 1) | ((r & 2) << 1) | ((r & 1) << 3);      // the lane (of 16) with rs_index == r
                part[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_base + src * 4, __builtin_bit_cast(int, gate)));
            }
        }
        if constexpr (ROWS) {
            // Row-list form: the gated messages of the tile's 32 rows, row-major fp16 [row][256] (512 B per row).  Same transposition
            // through the wave's free staging buffer as the last layer's store below; a unit (8 channels of one row, 16 B) goes to
            // uint4 slot row * 32 + nt * 4 + unit-of-the-n-tile.  Cached stores: k_l0_gather reads the rows back out of L2 / the
            // Infinity Cache within the same evaluation (the table build keeps its rows for the life of the complex).
            char *tb = stage + 2048;
            uint4 *Rout = reinterpret_cast<uint4 *>(p.rows_out) + (size_t)i * (32 * 32);
            const int u = l31 >> 3;
            int wbase[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) wbase[x] = u * 512 + ((x ^ u) + 4 * h) * 16 + (l31 & 7) * 2;
            int rd[2], wr[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int unit = lane + 64 * k2, uu = unit >> 5, row = unit & 31;
                rd[k2] = uu * 512 + (row ^ uu) * 16;
                wr[k2] = row * 32 + uu;
            }
#pragma unroll
            for (int nt = 0; nt < MSG_NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t pk = pack_f16_sat_lo(acc[nt][r] * part[r], acc[nt][r + 1] * part[r + 1]);
                    *reinterpret_cast<uint16_t *>(tb + wbase[r & 3] + (r >> 2) * 128) = (uint16_t)pk;
                    *reinterpret_cast<uint16_t *>(tb + wbase[(r + 1) & 3] + (r >> 2) * 128) = (uint16_t)(pk >> 16);
                }
                mstore_fence();
                const uint4 v0 = *reinterpret_cast<const uint4 *>(tb + rd[0]), v1 = *reinterpret_cast<const uint4 *>(tb + rd[1]);
                mstore_fence();      // the reads have returned before the next n-tile overwrites the buffer
                Rout[wr[0] + nt * 4] = v0;
                Rout[wr[1] + nt * 4] = v1;
            }
        } else {
        if (p.last && i >= p.R) {
            // Gated messages of a ligand node in the A-fragment order k_edge_coord reads: [k-step 16][lane half 2][row 32][8 channels].
            // An n-tile (32 channels) is one contiguous 2 KiB of that: 4 units (k-step, half) x 32 rows x 16 B.  A lane owns ONE
            // channel of 16 rows, so direct stores are 128 two-byte stores per tile (~100 cycles of issue each); instead every n-tile
            // goes through the wave's free staging buffer (buffer 1: chunk 7 has consumed it, buffer 0 already holds the next tile's
            // chunk 0): 16 ds_write_b16 (unit u, row ^ u: conflict-free), then two 16-byte reads + fully coalesced 16-byte stores.
#if DFM_EDGE_MSTORE_LDS
            char *tb = stage + 2048;
            uint4 *Mout = reinterpret_cast<uint4 *>(p.mbuf + (((size_t)b * p.L + (i - p.R)) * 2 + mt) * (32 * H));
            const int u = l31 >> 3;
            int wbase[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) wbase[x] = u * 512 + ((x ^ u) + 4 * h) * 16 + (l31 & 7) * 2;
            int rd[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int unit = lane + 64 * k2, uu = unit >> 5, row = unit & 31;
                rd[k2] = uu * 512 + (row ^ uu) * 16;
            }
            // (opaque copies of the gates: otherwise hipcc computes all 128 gate * message products once, for this store AND the
            // segment sums below, and spills them)
            float ps[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { ps[r] = part[r]; asm volatile("" : "+v"(ps[r])); }
#pragma unroll
            for (int nt = 0; nt < MSG_NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float g0 = acc[nt][r] * ps[r], g1 = acc[nt][r + 1] * ps[r + 1];
                    uint32_t pk;
                    if constexpr (F16) pk = pack_f16_sat_lo(g0, g1);
                    else pk = (uint32_t)to16<0>(g0) | ((uint32_t)to16<0>(g1) << 16);
                    *reinterpret_cast<uint16_t *>(tb + wbase[r & 3] + (r >> 2) * 128) = (uint16_t)pk;
                    *reinterpret_cast<uint16_t *>(tb + wbase[(r + 1) & 3] + (r >> 2) * 128) = (uint16_t)(pk >> 16);
                }
                mstore_fence();
                const uint4 v0 = *reinterpret_cast<const uint4 *>(tb + rd[0]), v1 = *reinterpret_cast<const uint4 *>(tb + rd[1]);
                mstore_fence();      // the reads have returned before the next n-tile overwrites the buffer
                store_stream(Mout + nt * 128 + lane, v0);
                store_stream(Mout + nt * 128 + 64 + lane, v1);
            }
#else
            uint16_t *Mout = p.mbuf + (((size_t)b * p.L + (i - p.R)) * 2 + mt) * (32 * H);
#pragma unroll
            for (int nt = 0; nt < MSG_NT; ++nt) {
                const int cbase = (((nt * 2 + (l31 >> 4)) * 2 + ((l31 >> 3) & 1)) * 32) * 8 + (l31 & 7);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowin = (r & 3) + 8 * (r >> 2) + 4 * h;
                    Mout[cbase + rowin * 8] = to16<F16>(acc[nt][r] * part[r]);
                }
            }
#endif
        }
        if (!p.no_agg) {      // (the ligand-only last layer has no reader for the segment sums: its launches skip them, r06)
#pragma unroll
        for (int nt = 0; nt < MSG_NT; ++nt) {
            f2 cs = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) cs = (f2){acc[nt][2 * q], acc[nt][2 * q + 1]} * (f2){part[2 * q], part[2 * q + 1]} + cs;
            const float t = cs.x + cs.y;
            // this tile's x_t (both lane halves); rounded on its own (the empty asm keeps hipcc from contracting the product into
            // an fma with the running sum) so that node tasks and tile tasks add the same two numbers
            float xt = (t + __shfl_xor(t, 32, 64)) * p.inv_s;
            asm volatile("" : "+v"(xt));
            colsum[nt] += xt;
        }
        }
        if (split || mt == ntile - 1) {
            // The lane term of this address is re-derived HERE (v_mbcnt_lo = the lane index for the storing half-wave h == 0; the
            // volatile asm keeps hipcc from hoisting `p.agg + l31` out of the task loop): hoisted, that 64-bit value was the one thing
            // this 256-register kernel spilled to SCRATCH - and two of these kernels running at once on different streams (two complex
            // handles, driver.run_set) then stored segment sums through each other's spilled pointer: rows of one handle's agg left
            // stale, rows of the other's overwritten (r05: tools/concurrency_probe*.py).  No kernel of this library may use scratch;
            // tests/test_abi_cpu.py checks the code objects' private_segment_fixed_size.
            uint32_t le;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0" : "=v"(le));
            float *out = p.agg + ((size_t)b * p.N + i) * H + le;
#pragma unroll
            for (int nt = 0; nt < MSG_NT; ++nt) {
                if (h == 0 && !p.no_agg) {
                    if (split) atomicAdd(out + nt * 32, colsum[nt]); else store_stream(out + nt * 32, colsum[nt]);
                }
                colsum[nt] = 0.f;
            }
        }
        }      // !ROWS
        STAMP(3);
        if (!have_next) break;
        if (fetch_now) {
            __builtin_amdgcn_sched_barrier(0);
            uint32_t got;
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(got) : "v"(fetched));
            dyn_next = dyn_index(got);
        }
        // the rest of the next tile's chunk 1 (kept out of the epilogue's register budget): the second pass is first used at slot 8
        if constexpr (DFM_EDGE_DEFER >= 2 && !DFM_EDGE_HALF) gather(1, 1, r1);
        if constexpr (DFM_EDGE_DEFER >= 1) gather_chunk(1);
        tt = ntt; b = nb; i = ni; mt = nmt;
    }
    if (dyn && lane == 0) {      // count this wave out; the last of the workgroup's waves that had a task leaves the counters zeroed for the next launch
        uint32_t active = 0;      // waves of this workgroup whose static first task exists
        for (int w = 0; w < MSG_WAVES; ++w) active += (uint32_t)w * (uint32_t)wg_per_xcd + (uint32_t)slot < ntask ? 1u : 0u;
        const uint32_t gone = __hip_atomic_fetch_add(p.task_ctr + TASK_CTR_WGS + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone + 1u == active) {
            __hip_atomic_store(p.task_ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.task_ctr + TASK_CTR_WGS + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef DFM_EDGE_STAMP
    if (p.stamp && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 4; ++k) p.stamp[wave * 4 + k] = st_t[k];
#endif
#if !defined(DFM_EDGE_STAMP) && !defined(DFM_EDGE_TRACE)
    if (clk_wave && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1c = __builtin_amdgcn_s_memrealtime();
        p.stamp[0] += t1 - p.stamp[2];
        p.stamp[1] += r1c - p.stamp[3];
    }
#endif
#undef STAMP
#undef STAMP0
#undef TRACE
}

// Coordinate MLP of the last layer (egnn.py:118-137) over the stored gated messages of the ligand nodes: same tile and epilogue
// layout as the message kernel, the A operand comes straight from HBM in fragment order (no producer).
template <int F16>   // F16 0: bf16 MFMA operands, 1: fp16 MFMA operands (3 more mantissa bits, same rate)
__global__ __launch_bounds__(EDGE_WAVES * 64) void k_edge_coord(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *Wf = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    for (int q = tid; q < LDS_WF_BYTES / 16; q += EDGE_WAVES * 64) Wf[q] = p.Wf[q];
    __syncthreads();

    // XCD-aware task order (speed only): workgroup g runs on XCD g % 8; give every XCD whole trajectories
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int NT = p.L;                                   // node tasks per trajectory: the ligand nodes
    const int nsplit = p.B >= 8 ? 1 : (8 + p.B - 1) / p.B;   // few trajectories: split each over several XCDs
    const int NTc = (NT + nsplit - 1) / nsplit;           // nodes per chunk
    const int U = p.B * nsplit;                           // chunks; chunk u lives on XCD u % 8
    const int nb = U > xcd ? (U - xcd + 7) >> 3 : 0;      // chunks owned by this XCD
    const unsigned ntask = (unsigned)nb * (unsigned)NTc;
    const unsigned tstride = (unsigned)wg_per_xcd * EDGE_WAVES;
    const int K = p.K, ntile = (K + 31) >> 5;
    const float *dot_v = p.wc2;                           // wc2 pre-divided by SILU_S

    // task tt of this XCD -> (trajectory, node); wave-uniform, kept in SGPRs
    auto task_node = [&](unsigned tt, int &b, int &i) -> bool {
        const unsigned tq = tt / (unsigned)NTc, tr = tt - tq * (unsigned)NTc;
        const int u = xcd + 8 * (int)tq;
        b = __builtin_amdgcn_readfirstlane(u / nsplit);
        const int idx = __builtin_amdgcn_readfirstlane((u % nsplit) * NTc + (int)tr);
        i = p.R + idx;
        return idx < NT;
    };
    // the constant A operand of the bias k-step: (1, 1, 0 ...) in k = 0, 1 of every row; bias = hi + lo in the B operand
    Frag onef;
    onef.u = make_uint4(h == 0 ? (F16 ? 0x3c003c00u : 0x3f803f80u) : 0u, 0u, 0u, 0u);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // The walk over (node, tile) runs ahead on the memory side: the tile is computed in two halves of 128 columns (accumulators 64
    // instead of 128; its A fragments stay in registers for both), and in the SECOND half every k-step's fragment register is
    // re-loaded with the same k-step of the wave's NEXT tile as soon as its last MFMA has been issued - so the HBM latency that
    // bounded this kernel in r02 (0.83 ms for 2.6 GB: 3.2 TB/s; every wave alternated between a load phase and a compute phase) lies
    // under the rest of the half, the epilogue and the next tile's first MFMAs, with no second register set.
    auto tile_ptr = [&](int tb, int ti, int tm) {
        return reinterpret_cast<const uint4 *>(p.mbuf + (((size_t)tb * p.L + (ti - p.R)) * 2 + tm) * (32 * H)) + lane;
    };
    auto load_tile = [&](const uint4 *Mt, uint4 (&a)[16]) {      // read once: non-temporal, like the stores that wrote them
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const u32x4v v = DFM_EDGE_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(Mt + kk * 64))
                                         : *reinterpret_cast<const u32x4v *>(Mt + kk * 64);
            a[kk] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    unsigned tt = (unsigned)wave * (unsigned)wg_per_xcd + (unsigned)slot;      // wave-major: a launch with fewer tasks than waves spreads over ALL workgroups
                                                                               // (a few waves each, a SIMD to themselves) instead of filling the first ones
    int b = 0, i = 0, mt = 0;
    while (tt < ntask && !task_node(tt, b, i)) tt += tstride;
    if (tt >= ntask) return;
    // dynamic tasks as in k_edge_msg (r06): the workgroup's share of the task list, taken in order from a per-workgroup counter; the fetch
    // for the task after the current one is issued right behind a tile's neighbour-coordinate load (the youngest load at that point, so no
    // earlier load waits for it) and read at the top of the next tile
    const bool dyn = nsplit == 1 && p.task_ctr != nullptr;
    uint32_t dyn_next = ~0u, fetched = 0;
    auto fetch_task = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(p.task_ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto dyn_index = [&](uint32_t pos) -> uint32_t { return (uint32_t)slot + (uint32_t)wg_per_xcd * ((uint32_t)EDGE_WAVES + pos); };
    if (dyn && ntile == 1) dyn_next = dyn_index((uint32_t)__builtin_amdgcn_readfirstlane((int)fetch_task()));
    bool fetch_pending = false;
    float cacc0 = 0.f, cacc1 = 0.f, cacc2 = 0.f;   // sum_s cdiff * w of the open node
    bool more = true;
    uint4 cur[16];
    // one tile from `cur`, which leaves holding the wave's next tile (if any)
    auto tile_step = [&]() {
        unsigned ntt = tt;
        int nb_ = b, ni = i, nmt = mt + 1;
        bool have_next = true;
        if (fetch_pending) {      // the fetch issued during the previous tile
            uint32_t got;
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(got) : "v"(fetched));
            dyn_next = dyn_index(got);
            fetch_pending = false;
        }
        if (nmt == ntile) {
            nmt = 0;
            if (dyn) {
                ntt = dyn_next;
                have_next = ntt < ntask && task_node(ntt, nb_, ni);
            } else {
                ntt = tt + tstride;
                while (ntt < ntask && !task_node(ntt, nb_, ni)) ntt += tstride;
                have_next = ntt < ntask;
            }
        }
        const bool fetch_now = dyn && have_next && nmt == ntile - 1;      // the next tile is the last of its task
        const uint4 *Mn = tile_ptr(have_next ? nb_ : b, have_next ? ni : i, have_next ? nmt : mt);      // (last tile: re-reads itself, unused)
        const size_t node = (size_t)b * p.N + i;
        const size_t ebase = node * K;
        const int lrow = mt * 32 + (l31 & 3) + 8 * ((l31 >> 2) & 3) + 4 * h;    // one lane per row (16 rows per half, lanes l31 < 16)
        const float4 c_xi = p.ca4[node];
        const int c_j = p.edges[ebase + (lrow < K ? lrow : 0)];
        float4 c_xj = c_xi;
        f2 part2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) part2[q] = (f2){0.f, 0.f};
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x16 acc[4];
            float dv[4];        // dot vector of the epilogue
            uint32_t bp[4];     // packed (hi, lo) bias of this lane's column per n-tile (biasp is [8][64]: lanes 32..63 hold 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dv[j] = dot_v[(hf * 4 + j) * 32 + l31];
                bp[j] = p.biasp[(hf * 4 + j) * 64 + lane];
            }
            constexpr int CDEPTH = 4;      // weight-fragment LDS reads run three ahead in a static register ring
            const uint4 *wq = Wf + lane + hf * 4 * 64;      // fragment (k-step kk, n-tile hf*4 + j) sits at (kk * 8 + hf * 4 + j) * 64
            Frag bq[CDEPTH];
#pragma unroll
            for (int d = 0; d < CDEPTH - 1; ++d) bq[d].u = wq[((d >> 2) * 8 + (d & 3)) * 64];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int mm = 0; mm < 32; ++mm) {
                    const int m = g * 32 + mm, mn = m + CDEPTH - 1;      // m = kk * 4 + j
                    if (mn < 64) bq[mn % CDEPTH].u = wq[((mn >> 2) * 8 + (mn & 3)) * 64];
                    Frag af;
                    af.u = cur[m >> 2];
                    if (m < 4) acc[m] = mfma16<F16>(af, bq[m % CDEPTH], zero16);
                    else acc[m & 3] = mfma16<F16>(af, bq[m % CDEPTH], acc[m & 3]);
                    if (hf == 1 && (m & 3) == 3) {      // k-step m >> 2 of this tile is done: its register takes the next tile's
                        const u32x4v v = DFM_EDGE_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(Mn + (m >> 2) * 64))
                                                     : *reinterpret_cast<const u32x4v *>(Mn + (m >> 2) * 64);
                        cur[m >> 2] = make_uint4(v.x, v.y, v.z, v.w);
                    }
                }
                if (hf == 0 && g == 0) {
                    c_xj = p.ca4[(size_t)b * p.N + c_j];     // the edge index has landed under the first 32 MFMAs
                    if (fetch_now) { fetched = fetch_task(); fetch_pending = true; }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting the next group's reads (spills)
            }
            // bias k-step: acc += 1 * hi + 1 * lo (the accumulators were opened with C = 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Frag bb;
                bb.u = make_uint4(bp[j], 0u, 0u, 0u);
                acc[j] = mfma16<F16>(onef, bb, acc[j]);
            }
            // epilogue of the half: lane owns columns (hf*4 + j)*32 + l31, rows rowof(r): w += sum_c silu(.) * wc2
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 vv = {dv[j], dv[j]};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f2 m = silu2s((f2){acc[j][2 * q], acc[j][2 * q + 1]});
                    part2[q] = m * vv + part2[q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // the halves in sequence: interleaved they would hold both accumulator sets
        }
        float part[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) { part[2 * q] = part2[q].x; part[2 * q + 1] = part2[q].y; }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = half_sum_dpp(part[r]);   // all 32 lanes of the half hold the row sum
        // coord_mlp: w = clamp(sum_c silu(.) * wc2, +-2); x_i += mean_s (x_i - x_j)/(|x_i - x_j| + 1) * w
        // one lane per row (lane l31 < 16 takes register row r = l31), so the 32 edge-index / coordinate loads of a tile
        // are issued together instead of as a 16-long dependent chain in one lane
        float w = part[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) w = l31 == r ? part[r] : w;
        if (l31 < 16 && lrow < K) {
            const float4 xi = c_xi, xj = c_xj;
            w = fminf(fmaxf(w, -2.0f), 2.0f);
            const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const float nrm = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f) + 1.0f;
            cacc0 += dx / nrm * w; cacc1 += dy / nrm * w; cacc2 += dz / nrm * w;
        }
        if (mt == ntile - 1) {      // the node is complete
            const float s0 = wave_sum(cacc0), s1 = wave_sum(cacc1), s2 = wave_sum(cacc2);
            if (lane == 0) {
                const float4 xi = c_xi;
                const float inv = 1.0f / (float)(K > 1 ? K : 1);
                float *fo = p.fout + ((size_t)b * p.L + (i - p.R)) * 3;
                fo[0] = (xi.x + s0 * inv) - xi.x;
                fo[1] = (xi.y + s1 * inv) - xi.y;
                fo[2] = (xi.z + s2 * inv) - xi.z;
            }
            cacc0 = cacc1 = cacc2 = 0.f;
        }
        tt = ntt; b = nb_; i = ni; mt = nmt;
        more = have_next;
    };
    load_tile(tile_ptr(b, i, 0), cur);
    while (more) tile_step();
    if (dyn && lane == 0) {      // count this wave out; the last one of the workgroup leaves the counters zeroed for the next launch
        uint32_t active = 0;
        for (int w = 0; w < EDGE_WAVES; ++w) active += (uint32_t)w * (uint32_t)wg_per_xcd + (uint32_t)slot < ntask ? 1u : 0u;
        const uint32_t gone = __hip_atomic_fetch_add(p.task_ctr + TASK_CTR_WGS + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone + 1u == active) {
            __hip_atomic_store(p.task_ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.task_ctr + TASK_CTR_WGS + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// -------------------------------------------------------------------------------------------------
static EdgeKArgs to_kargs(const EdgeArgs &a)
{
    EdgeKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.A = a.A; k.Bm = a.Bm; k.Bmb = a.Bmb; k.ab_bstride = a.ab_bstride;
    k.edges = a.edges; k.codes = a.codes; k.radial = a.radial; k.ca4 = a.ca4;
    k.B = a.B; k.N = a.N; k.R = a.R; k.K = a.K; k.L = a.N - a.R;
    const LayerDev *w = a.lw;
    k.w_r = w->w_r; k.T = w->T; k.W2t = w->W2t; k.b2 = w->b2; k.att_w = w->att_w; k.T2b = w->T2b;
    k.Wf = reinterpret_cast<const uint4 *>(w->W2f); k.att_b = w->att_b;
    k.Wc1t = w->Wc1t; k.bc1 = w->bc1; k.wc2 = w->wc2;
    k.agg = a.agg; k.last = a.last; k.fout = a.fout; k.mbuf = a.mbuf; k.stamp = a.stamp; k.split = 0;
    k.node0 = a.lig_only ? a.R : 0; k.nodes = a.lig_only ? a.N - a.R : a.N;
    k.no_agg = a.lig_only ? 1 : 0;      // the ligand-only last layer feeds the coordinate update alone (api.hip: no node model follows)
    k.Ah = a.Ah; k.range = a.range; k.task_ctr = nullptr;
    return k;
}
// the 16-bit MFMA kernels take the -log2(e)-scaled operands (SILU_S, api.hip)
static EdgeKArgs to_kargs_mfma(const EdgeArgs &a, int mode)
{
    EdgeKArgs k = to_kargs(a);
    const LayerDev *w = a.lw;
    k.w_r = w->w_r_s; k.att_b = w->att_b * SILU_S; k.inv_s = 1.0f / SILU_S; k.wc2 = w->wc2_s;
    if (mode == 0) { k.Wf = reinterpret_cast<const uint4 *>(a.f16 ? w->W2f16 : w->W2f); k.biasp = a.f16 ? w->b2p16 : w->b2p; }
    else { k.Wf = reinterpret_cast<const uint4 *>(a.f16 ? w->Wc1f16 : w->Wc1f); k.biasp = a.f16 ? w->bc1p16 : w->bc1p; }
    return k;
}

hipError_t launch_edge_f32(const EdgeArgs &a, hipStream_t s)
{
    static const bool scalar = [] { const char *e = getenv("DFM_EDGE_F32_SCALAR"); return e && atoi(e) != 0; }();      // diagnostics: the r01-r03 kernel
    if (!scalar) {
        static std::atomic<bool> attr_m[MAX_DEVICES];
        hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_f32m<0>), LDS_F32M_BYTES, attr_m);
        if (e != hipSuccess) return e;
        const EdgeKArgs k = to_kargs(a);
        const long long tasks = (long long)a.B * k.nodes;
        hipLaunchKernelGGL(k_edge_f32m<0>, dim3((unsigned)((tasks + 1) / 2)), dim3(256), LDS_F32M_BYTES, s, k);
        return hipGetLastError();
    }
    static std::atomic<bool> attr_done[MAX_DEVICES];
    const int lds = KF * H * 4 + 4 * 64 * 4;
    {
        hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_f32), lds, attr_done);
        if (e != hipSuccess) return e;
    }
    const EdgeKArgs k = to_kargs(a);
    hipLaunchKernelGGL(k_edge_f32, dim3((unsigned)((long long)a.B * a.N)), dim3(256), lds, s, k);
    return hipGetLastError();
}

static int persistent_grid(long long wave_tasks)
{
    const int cus = device_cus();
    long long wgs = wave_tasks;      // small launches: one workgroup per CU as soon as there is a task for it (tasks go wave-major)
    long long g = wgs < cus ? wgs : cus;
    g = (g + 7) / 8 * 8;   // multiple of the XCD count
    return (int)g;
}

template <int F16, int AW16> static hipError_t launch_msg_t(const EdgeKArgs &k, long long wave_tasks, hipStream_t s)
{
    static std::atomic<bool> attr_done[MAX_DEVICES];
    {
        hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_msg<F16, AW16>), LDS_EDGE_BYTES, attr_done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_edge_msg<F16, AW16>), dim3(persistent_grid(wave_tasks)), dim3(MSG_WAVES * 64), LDS_EDGE_BYTES, s, k);
    return hipGetLastError();
}
template <int F16> static hipError_t launch_coord_t(const EdgeKArgs &k, long long wave_tasks, hipStream_t s)
{
    static std::atomic<bool> attr_done[MAX_DEVICES];
    {
        hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_coord<F16>), LDS_EDGE_BYTES, attr_done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_edge_coord<F16>), dim3(persistent_grid(wave_tasks)), dim3(EDGE_WAVES * 64), LDS_EDGE_BYTES, s, k);
    return hipGetLastError();
}

// Small launches: with one node (two tiles) per task the last round of the persistent grid is mostly idle - e.g. B = 8 at
// N = 600: 4800 nodes over 2048 waves = 3 rounds of 2 tiles, but 9600 tiles = 5 rounds of 1.  Tile tasks when that saves a round;
// their partial segment sums are added atomically to an agg that must be zero (EdgeArgs::agg_is_zero, or a memset here).
bool edge_msg_tile_tasks(int B, int N /* nodes with a task per trajectory */, int K)
{
    const int ntile = (K + 31) / 32;
    if (ntile <= 1) return false;
    static const int env = [] { const char *e = getenv("DFM_EDGE_SPLIT"); return e ? atoi(e) : -1; }();      // diagnostics: 0 / 1 force
    if (env >= 0) return env != 0;
    const int cus = device_cus();
    const long long tasks = (long long)B * N, waves = (long long)cus * EDGE_WAVES;
    const long long rounds_node = (tasks + waves - 1) / waves * ntile, rounds_tile = (tasks * ntile + waves - 1) / waves;
    return rounds_tile < rounds_node;
}

hipError_t launch_edge_bf16(const EdgeArgs &a, hipStream_t s)
{
    EdgeKArgs k = to_kargs_mfma(a, 0);
    long long tasks = (long long)a.B * k.nodes;
    if (edge_msg_tile_tasks(a.B, k.nodes, a.K)) {
        k.split = 1; tasks *= (a.K + 31) / 32;
        if (!a.agg_is_zero && !k.no_agg) {
            hipError_t e = hipMemsetAsync(a.agg, 0, (size_t)a.B * a.N * H * sizeof(float), s);
            if (e != hipSuccess) return e;
        }
    }
    else if (a.task_ctr && a.B >= 8 && tasks >= 2 * (long long)device_cus() * MSG_WAVES) {
        // node tasks, more tasks than waves: every wave's tasks after its first come from the per-XCD counters (dynamic tasks, k_edge_msg)
        static const bool off = [] { const char *e = getenv("DFM_EDGE_DYNAMIC"); return e && atoi(e) == 0; }();      // diagnostics: the fixed stride of r01-r05
        if (!off && device_cus() <= TASK_CTR_WGS) k.task_ctr = a.task_ctr;      // zero at allocation, zeroed again by the last wave of every launch
    }
    if (a.f16) return a.Ah ? launch_msg_t<1, 1>(k, tasks, s) : launch_msg_t<1, 0>(k, tasks, s);
    return a.Ah ? launch_msg_t<0, 1>(k, tasks, s) : launch_msg_t<0, 0>(k, tasks, s);
}

hipError_t launch_coord_bf16(const EdgeArgs &a, hipStream_t s)
{
    EdgeKArgs k = to_kargs_mfma(a, 1);
    const long long tasks = (long long)a.B * (a.N - a.R);
    static const bool off = [] { const char *e = getenv("DFM_EDGE_DYNAMIC"); return e && atoi(e) == 0; }();
    if (!off && a.task_ctr && a.B >= 8 && tasks >= 2 * (long long)device_cus() * EDGE_WAVES && device_cus() <= TASK_CTR_WGS) k.task_ctr = a.task_ctr;
    return a.f16 ? launch_coord_t<1>(k, tasks, s) : launch_coord_t<0>(k, tasks, s);
}

// -------------------------------------------------------------------------------------------------
// Layer 0 behind the per-complex message table (src/models/egnn.py:95-104 evaluated once per intra-chain pair instead of once per
// edge, trajectory and step).  In layer 0 the node features are the embedding h0 of the complex (score_net_mlsb.py:365-366: no pose,
// no time in it), so the gated message of an edge (i, j) is a function of the pair's geometry alone - and for two residues of the
// SAME chain that geometry never changes under the rigid motion of the ligand.  M0 [pairs][256] (fp16, S * gate * m like the last
// layer's message buffer) holds it for every intra-chain ordered pair; an evaluation then needs
//   k_edge_feat      (kernels_geom.hip) classifies every edge: intra-chain AND its per-pose feature code equal to the table's code0 ->
//                    src = pair index; anything else (inter-chain edges; the rare pair whose fp32 features land in another bin in this
//                    pose than in the table's) -> appended to a row list, src = MISS | position
//   k_edge_msg<1,1,1> the edge model on the row list -> X [position][256]
//   k_l0_gather      agg[b][i] = (1 / S) * sum over the node's K slots IN SLOT ORDER of M0[src] or X[position]: no atomics, and the value
//                    of a row does not depend on its position in the list, so the result is a pure function of (pose, graph)
// At 300+300 that is 0.55 ms of gather-sum out of L2 / the Infinity Cache plus a message launch over 2 - 13 % of the edges instead of
// 2.2 ms (profiles/r04_l0_table.txt).  The only approximation against the direct kernel is the fp16 rounding of each stored message
// before the 60-row sum (the direct kernel sums fp32 registers).
constexpr uint32_t L0_MISS = 0x80000000u;

__global__ __launch_bounds__(256) void k_l0_gather(const uint2 *__restrict__ table, const uint2 *__restrict__ X, const uint32_t *__restrict__ src,
                                                   float4 *__restrict__ agg, int B, int N, int K, float inv_s, uint32_t *counter,
                                                   unsigned long long *miss_total)
{
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t == 0 && lane == 0) {      // this evaluation's row list is consumed: keep the count for the profile, reset it for the next evaluation
        *miss_total += *counter;
        *counter = 0u;
    }
    if (t >= (long long)B * N) return;
    // node-major task order: neighbouring waves handle the same node of neighbouring trajectories, whose kNN slots are the same
    // table rows (the chain is rigid) - they are re-read from this XCD's L2 instead of from the Infinity Cache
    const int i = (int)(t / B), b = (int)(t - (long long)i * B);
    const size_t node = (size_t)b * N + i;
    const uint32_t my = lane < K ? src[node * K + lane] : 0u;      // K <= 60: one slot per lane
    // r06: 16-byte loads, TWO rows per wave instruction (lanes 0..31 the even slot's 512-byte row, lanes 32..63 the odd slot's) - the L2
    // serves 16-byte requests at 1.4 - 1.8 x the rate of 8-byte ones (r01-r05: one row per instruction, 8 bytes per lane).  Each half
    // sums its own slots in slot order, the two partial sums are added at the end: still a fixed order per node, whatever the batch.
    const int half = lane >> 5, l = lane & 31;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < K; s0 += 12) {      // six instructions = twelve rows in flight
        uint4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int sa = s0 + 2 * q, sb = sa + 1;
            const uint32_t ida = (uint32_t)__builtin_amdgcn_readlane((int)my, sa < K ? sa : 0), idb = (uint32_t)__builtin_amdgcn_readlane((int)my, sb < K ? sb : 0);
            const uint32_t id = half ? idb : ida;
            const uint2 *row = (id & L0_MISS) ? X + (size_t)(id & ~L0_MISS) * 64 : table + (size_t)id * 64;
            v[q] = (half ? sb : sa) < K ? reinterpret_cast<const uint4 *>(row)[l] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {      // (rows past K were loaded as zeros: adding them changes nothing)
            acc[0] = add_half_lo(acc[0], v[q].x); acc[1] = add_half_hi(acc[1], v[q].x);
            acc[2] = add_half_lo(acc[2], v[q].y); acc[3] = add_half_hi(acc[3], v[q].y);
            acc[4] = add_half_lo(acc[4], v[q].z); acc[5] = add_half_hi(acc[5], v[q].z);
            acc[6] = add_half_lo(acc[6], v[q].w); acc[7] = add_half_hi(acc[7], v[q].w);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += __shfl_xor(acc[c], 32, 64);      // even slots + odd slots
    if (half == 0) {
        agg[node * 64 + 2 * l] = make_float4(acc[0] * inv_s, acc[1] * inv_s, acc[2] * inv_s, acc[3] * inv_s);
        agg[node * 64 + 2 * l + 1] = make_float4(acc[4] * inv_s, acc[5] * inv_s, acc[6] * inv_s, acc[7] * inv_s);
    }
}

// fp32 engine: the same gather-sum over fp32 rows (1 KiB each: a wave instruction covers one row with 16-byte loads), no scale
__global__ __launch_bounds__(256) void k_l0_gather32(const float4 *__restrict__ table, const float4 *__restrict__ X, const uint32_t *__restrict__ src,
                                                     float4 *__restrict__ agg, int B, int N, int K, uint32_t *counter, unsigned long long *miss_total)
{
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t == 0 && lane == 0) {
        *miss_total += *counter;
        *counter = 0u;
    }
    if (t >= (long long)B * N) return;
    const int i = (int)(t / B), b = (int)(t - (long long)i * B);
    const size_t node = (size_t)b * N + i;
    const uint32_t my = lane < K ? src[node * K + lane] : 0u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < K; s0 += 6) {
        float4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my, (s0 + q) < K ? s0 + q : 0);
            const float4 *row = (id & L0_MISS) ? X + (size_t)(id & ~L0_MISS) * 64 : table + (size_t)id * 64;
            v[q] = (s0 + q) < K ? row[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (s0 + q < K) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
        }
    }
    agg[node * 64 + lane] = acc;
}

hipError_t launch_l0_gather32(const float *table, const float *X, const uint32_t *src, float *agg, int B, int N, int K,
                              uint32_t *counter, unsigned long long *miss_total, hipStream_t s)
{
    const long long waves = (long long)B * N;
    hipLaunchKernelGGL(k_l0_gather32, dim3((unsigned)((waves + 3) / 4)), dim3(256), token_lds(), s, reinterpret_cast<const float4 *>(table),
                       reinterpret_cast<const float4 *>(X), src, reinterpret_cast<float4 *>(agg), B, N, K, counter, miss_total);
    return hipGetLastError();
}

hipError_t launch_l0_gather(const uint16_t *table, const uint16_t *X, const uint32_t *src, float *agg, int B, int N, int K,
                            uint32_t *counter, unsigned long long *miss_total, hipStream_t s)
{
    const long long waves = (long long)B * N;
    hipLaunchKernelGGL(k_l0_gather, dim3((unsigned)((waves + 3) / 4)), dim3(256), token_lds(), s, reinterpret_cast<const uint2 *>(table),
                       reinterpret_cast<const uint2 *>(X), src, reinterpret_cast<float4 *>(agg), B, N, K, 1.0f / SILU_S, counter, miss_total);
    return hipGetLastError();
}

// the edge model over a row list (a = layer 0's EdgeArgs: the complex's own A0h / Bmb0, ab_bstride 0); n_rows_dev = nullptr: exactly
// n_rows_cap rows.  `out` must hold the row count rounded up to 32 rows (the last tile stores all of its rows).
hipError_t launch_edge_rows(const EdgeArgs &a, const uint4 *rows, const uint32_t *n_rows_dev, uint32_t n_rows_cap, uint16_t *out, hipStream_t s)
{
    if (!a.Ah || !a.f16 || a.ab_bstride != 0 || n_rows_cap == 0) return hipErrorInvalidValue;
    EdgeKArgs k = to_kargs_mfma(a, 0);
    k.rows = rows; k.n_rows_dev = n_rows_dev; k.n_rows = n_rows_cap; k.rows_out = out; k.last = 0;
    static std::atomic<bool> attr_done[MAX_DEVICES];
    hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_msg<1, 1, 1>), LDS_EDGE_BYTES, attr_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_edge_msg<1, 1, 1>), dim3(persistent_grid(((long long)n_rows_cap + 31) / 32)), dim3(MSG_WAVES * 64), LDS_EDGE_BYTES, s, k);
    return hipGetLastError();
}

// fp32 engine: the edge model over a row list (a = layer 0's EdgeArgs with the complex's own fp32 A0 / Bm0, ab_bstride 0) -> fp32 rows
hipError_t launch_edge_rows32(const EdgeArgs &a, const uint4 *rows, const uint32_t *n_rows_dev, uint32_t n_rows_cap, float *out, hipStream_t s)
{
    if (a.ab_bstride != 0 || n_rows_cap == 0) return hipErrorInvalidValue;
    EdgeKArgs k = to_kargs(a);
    k.rows = rows; k.n_rows_dev = n_rows_dev; k.n_rows = n_rows_cap; k.rows_out32 = out; k.last = 0;
    static std::atomic<bool> attr_done[MAX_DEVICES];
    hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k_edge_f32m<1>), LDS_F32M_BYTES, attr_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_edge_f32m<1>, dim3((n_rows_cap + 127u) / 128u), dim3(256), LDS_F32M_BYTES, s, k);
    return hipGetLastError();
}

}  // namespace dfm
