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
//   k_edge_bf16 : 256x256 contraction on v_mfma_f32_32x32x16_bf16, fp32 accumulate; A-fragments are
//                 built in registers straight from the gathers, the weight matrix lives in LDS for the
//                 whole (persistent) workgroup.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_fp16.h>

#include "dfm_device.h"
#include "dfm_internal.h"

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
    long long *tl;   // DFM_TIMELINE debug: per-phase s_memtime stamps of wave 0 / workgroup 0 (nullptr normally)
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
            v = silu_exact(pre);
        }
        rows[s * H + c] = v;
    }
    __syncthreads();
    // edge_mlp.2 + SiLU
    float acc[KF];
    row_dot(rows, p.W2t, c, p.b2[c], acc);
    __syncthreads();
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
// Workgroup = 8 waves, persistent, all 160 KiB of LDS: 128 KiB hold the bf16 B-fragments of the 256x256
// weight matrix for the whole launch, 32 KiB are eight wave-private 4 KiB staging tiles.  A wave owns one
// node at a time = two 32-row M-tiles (60 edges + 4 masked rows).  Per M-tile and per 64-channel chunk:
//   producer  (gather layout: 8 adjacent lanes cover one row's 64 channels = whole 128-B lines):
//             A_i + Bm_j + w_r*radial + 3 merged T rows (fp16 gathers) -> SiLU -> bf16 -> ds_write_b128 (XOR-swizzled)
//   consumer  4 k-steps x 8 n-tiles of v_mfma_f32_32x32x16_bf16, A-fragments by ds_read_b128 from the
//             staging tile, B-fragments by ds_read_b128 from the resident weights.
// The two waves of a SIMD drift apart, so one wave's gathers/VALU run under the other's MFMAs.
// Epilogue in the C layout (lane = column, registers = rows): +b2, SiLU, attention gate (in-lane dot +
// 32-lane butterfly), row mask, 60-row segment sum in registers -> agg; no atomics.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 b; f16x8 f; };
union H8 { uint4 u; __half2 h[4]; };   // eight fp16 values of one gathered 16-byte chunk

constexpr int LDS_WF_BYTES = 16 * 8 * 64 * 16;     // 131072: bf16 B-fragments of one 256x256 matrix
constexpr int LDS_STAGE_BYTES = 32 * 64 * 2;       // 4096 per wave: 32 rows x 64 channels bf16
constexpr int EDGE_WAVES = 8;                      // waves per workgroup (two per SIMD, 256 registers each)
constexpr int LDS_EDGE_BYTES = LDS_WF_BYTES + 8 * LDS_STAGE_BYTES;   // 163840 = the whole CU

typedef float f2 __attribute__((ext_vector_type(2)));   // packed fp32 pair -> v_pk_{mul,add,fma}_f32 (2 results / instr)
__device__ inline float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// SiLU of two values: 3 packed ops + 2 v_exp_f32 + 2 v_rcp_f32
__device__ inline f2 silu2(f2 x)
{
    const f2 y = x * (f2){-1.44269504088896f, -1.44269504088896f};
    f2 e = {__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
    e = e + (f2){1.0f, 1.0f};
    const f2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return x * r;
}
__device__ inline float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

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

// make every earlier LDS access of this wave visible/ordered before later ones (wave-private staging tile)
__device__ inline void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct RawP { uint4 bm, t0, t1, t2; };        // gathered fp16 operands of one producer pass (8 channels of one row)

// one MFMA step on bf16 (F16 = 0) or fp16 (F16 = 1) operands, fp32 accumulate - same rate on gfx950
template <int F16> __device__ inline f32x16 mfma16(const Frag &a, const Frag &b, f32x16 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f, b.f, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c, 0, 0, 0);
}
template <int F16> __device__ inline uint16_t to16(float x)
{
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(x, -65504.f), 65504.f));
    else return __builtin_bit_cast(uint16_t, (__bf16)x);
}

template <int MODE, int F16, int TL = 0>   // MODE 0: edge messages (+ store of gated messages on the last layer), 1: coordinate MLP
                                 // F16 0: bf16 MFMA operands, 1: fp16 MFMA operands (3 more mantissa bits, same rate)
__global__ __launch_bounds__(EDGE_WAVES * 64) void k_edge_bf16(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *Wf = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *stage = smem + LDS_WF_BYTES + wave * LDS_STAGE_BYTES;
    const int h = lane >> 5, l31 = lane & 31;
    const int r8 = lane >> 3, c8 = lane & 7;            // producer layout: row-in-pass, 8-channel group
    for (int q = tid; q < LDS_WF_BYTES / 16; q += EDGE_WAVES * 64) Wf[q] = p.Wf[q];
    __syncthreads();

    // XCD-aware task order (speed only): workgroup g runs on XCD g % 8; give every XCD whole
    // trajectories so that the gathered rows of Bm stay in that XCD's L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int NT = MODE == 0 ? p.N : p.L;                 // node tasks per trajectory
    const int nsplit = p.B >= 8 ? 1 : (8 + p.B - 1) / p.B;   // few trajectories: split each over several XCDs
    const int NTc = (NT + nsplit - 1) / nsplit;           // nodes per chunk
    const int U = p.B * nsplit;                           // chunks; chunk u lives on XCD u % 8
    const int nb = U > xcd ? (U - xcd + 7) >> 3 : 0;      // chunks owned by this XCD
    const long long ntask = (long long)nb * NTc;
    const int K = p.K, ntile = (K + 31) >> 5;
    const float *bias_v = MODE == 0 ? p.b2 : p.bc1;       // bias of this contraction
    const float *dot_v = MODE == 0 ? p.att_w : p.wc2;     // att_w / wc2

    int tl_tile = 0;
    auto stamp = [&](int k) {
        if constexpr (TL) {
            if (blockIdx.x == 8 && wave == 0 && tl_tile < 64 && lane == 0) p.tl[tl_tile * 16 + k] = clock64();
        }
    };
    for (unsigned tt = (unsigned)slot * EDGE_WAVES + wave; tt < (unsigned)ntask; tt += (unsigned)wg_per_xcd * EDGE_WAVES) {
        const unsigned tq = tt / (unsigned)NTc, tr = tt - tq * (unsigned)NTc;
        const int u = xcd + 8 * (int)tq;
        const int b = __builtin_amdgcn_readfirstlane(u / nsplit);                    // wave-uniform -> SGPRs
        const int idx = __builtin_amdgcn_readfirstlane((u % nsplit) * NTc + (int)tr);
        if (idx >= NT) continue;
        const int i = (MODE == 0 ? 0 : p.R) + idx;
        const size_t node = (size_t)b * p.N + i;
        const size_t ebase = node * K;
        const size_t ab = (size_t)b * p.ab_bstride;
        float colsum[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) colsum[nt] = 0.f;
        float cacc0 = 0.f, cacc1 = 0.f, cacc2 = 0.f;   // MODE 1: sum_s cdiff * w

        for (int mt = 0; mt < ntile; ++mt) {
            stamp(0);
            f32x16 acc[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            float bv[8], dv[8];   // bias / dot vector of the epilogue, fetched under the last MFMA phase

            if (MODE == 0) {
                // per-pass row data: pass q handles rows mt*32 + q*8 + r8 (rows >= K: self edge, code 0 - finite filler)
                int jq[4]; uint32_t codeq[4]; float radq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = mt * 32 + q * 8 + r8;
                    const bool v = s < K;
                    jq[q] = v ? p.edges[ebase + s] : i;
                    codeq[q] = v ? p.codes[ebase + s] : 0u;
                    radq[q] = v ? p.radial[ebase + s] : 0.f;
                }
                const float *Arow = p.A + ab + (size_t)i * H + c8 * 8;
                float4 a0, a1, w0, w1;                 // per-chunk operands shared by the four passes
                auto gather_chunk = [&](int kc) {
                    a0 = *reinterpret_cast<const float4 *>(Arow + kc * 64);
                    a1 = *reinterpret_cast<const float4 *>(Arow + kc * 64 + 4);
                    w0 = *reinterpret_cast<const float4 *>(p.w_r + kc * 64 + c8 * 8);
                    w1 = *reinterpret_cast<const float4 *>(p.w_r + kc * 64 + c8 * 8 + 4);
                };
                auto gather = [&](int kc, int q, RawP &r) {
                    const uint32_t ch = kc * 64 + c8 * 8;
                    const uint32_t code = codeq[q];
                    r.bm = *reinterpret_cast<const uint4 *>(p.Bmb + ab + ((uint32_t)jq[q] * H + ch));
                    const uint32_t i0 = (((code >> 6) & 31u) * 24u + ((code >> 11) & 31u)) * H;
                    const uint32_t i1 = (576u + ((code >> 16) & 15u) * 40u + (code & 63u)) * H;
                    const uint32_t i2 = (1056u + ((code >> 20) & 127u)) * H;
                    r.t0 = *reinterpret_cast<const uint4 *>(p.T2b + (i0 + ch));
                    r.t1 = *reinterpret_cast<const uint4 *>(p.T2b + (i1 + ch));
                    r.t2 = *reinterpret_cast<const uint4 *>(p.T2b + (i2 + ch));
                };
                auto compute_store = [&](int q, const RawP &r) {
                    const f2 rad2 = {radq[q], radq[q]};
                    f2 v[4] = {(f2){w0.x, w0.y} * rad2 + (f2){a0.x, a0.y}, (f2){w0.z, w0.w} * rad2 + (f2){a0.z, a0.w},
                               (f2){w1.x, w1.y} * rad2 + (f2){a1.x, a1.y}, (f2){w1.z, w1.w} * rad2 + (f2){a1.z, a1.w}};
                    // table rows (and, with bf16 operands, Bm too) summed as packed fp16, then widened once
                    H8 t, t1, t2, bm;
                    t.u = r.t0; t1.u = r.t1; t2.u = r.t2; bm.u = r.bm;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t.h[e] = __hadd2(__hadd2(t.h[e], t1.h[e]), t2.h[e]);
                        if constexpr (!F16) t.h[e] = __hadd2(t.h[e], bm.h[e]);
                        else v[e] = v[e] + (f2){__low2float(bm.h[e]), __high2float(bm.h[e])};
                        v[e] = v[e] + (f2){__low2float(t.h[e]), __high2float(t.h[e])};
                    }
                    Frag f;   // rows >= K hold finite filler, gated to 0 below
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f2 m = silu2(v[e]);
                        if constexpr (F16) { f.f[2 * e] = (_Float16)fminf(m.x, 65504.f); f.f[2 * e + 1] = (_Float16)fminf(m.y, 65504.f); }
                        else { f.b[2 * e] = (__bf16)m.x; f.b[2 * e + 1] = (__bf16)m.y; }
                    }
                    const int row = q * 8 + r8;
                    *reinterpret_cast<uint4 *>(stage + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = f.u;
                };
                // every pass owns one raw buffer that is refilled in place for the NEXT chunk right after it is
                // consumed: a whole chunk of gathers (16 x 1 KiB per wave) flies under the 32 MFMAs of this chunk
                RawP r0, r1;
                gather_chunk(0);
                gather(0, 0, r0); gather(0, 1, r1);
                stamp(1);
                auto mfma_chunk = [&](int kc) {
                    wave_lds_fence();
                    stamp(3 + 2 * kc);
                    // 32 MFMAs on consecutive weight fragments; the fragment reads run BDEPTH - 1 ahead in a static
                    // register ring (one register set makes every MFMA wait a full LDS round trip)
                    constexpr int BDEPTH = 3;
                    const uint4 *wq = Wf + (size_t)kc * 32 * 64 + lane;
                    Frag bq[BDEPTH], af[2];
#pragma unroll
                    for (int d = 0; d < BDEPTH - 1; ++d) bq[d].u = wq[d * 64];
                    af[0].u = *reinterpret_cast<const uint4 *>(stage + l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
                    for (int m = 0; m < 32; ++m) {
                        if (m + BDEPTH - 1 < 32) bq[(m + BDEPTH - 1) % BDEPTH].u = wq[(m + BDEPTH - 1) * 64];
                        if ((m & 7) == 0 && m < 24)
                            af[((m >> 3) + 1) & 1].u = *reinterpret_cast<const uint4 *>(
                                stage + l31 * 128 + (((((m >> 3) + 1) * 2 + h) ^ ((l31 >> 1) & 7)) << 4));
                        acc[m & 7] = mfma16<F16>(af[(m >> 3) & 1], bq[m % BDEPTH], acc[m & 7]);
                    }
                    wave_lds_fence();
                    stamp(4 + 2 * kc);
                };
                // two raw buffers in a ring: the gathers of pass q+2 fly under the arithmetic of passes q, q+1 (and the
                // MFMA phase when they cross a chunk boundary)
#pragma unroll 1
                for (int kc = 0; kc < 3; ++kc) {
                    compute_store(0, r0); gather(kc, 2, r0);
                    if (kc == 0) stamp(2);
                    compute_store(1, r1); gather(kc, 3, r1);
                    compute_store(2, r0); gather(kc + 1, 0, r0);
                    compute_store(3, r1); gather(kc + 1, 1, r1);
                    gather_chunk(kc + 1);
                    mfma_chunk(kc);
                }
                compute_store(0, r0); gather(3, 2, r0);
                compute_store(1, r1); gather(3, 3, r1);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) { bv[nt] = bias_v[nt * 32 + l31]; dv[nt] = dot_v[nt * 32 + l31]; }
                compute_store(2, r0); compute_store(3, r1);
                mfma_chunk(3);
            } else {
                const int s = mt * 32 + l31;
                const bool valid = s < K;
                const uint16_t *Mrow = p.mbuf + (((size_t)b * p.L + (i - p.R)) * KPAD + (valid ? s : 0)) * H + h * 8;
                uint4 cur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const uint4 *>(Mrow + q * 16);
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    uint4 a4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) a4[q] = valid ? cur[q] : make_uint4(0, 0, 0, 0);
                    if (g < 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const uint4 *>(Mrow + ((g + 1) * 4 + q) * 16);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        Frag af;
                        af.u = a4[q];
#pragma unroll
                        for (int nt = 0; nt < 8; ++nt) {
                            Frag bf;
                            bf.u = Wf[((g * 4 + q) * 8 + nt) * 64 + lane];
                            acc[nt] = mfma16<F16>(af, bf, acc[nt]);
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) { bv[nt] = bias_v[nt * 32 + l31]; dv[nt] = dot_v[nt * 32 + l31]; }
            }

            // ---- epilogue on the 32 x 256 tile: lane owns columns nt*32 + l31, rows rowof(r) -------------
            float part[16];
            {
                f2 part2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) part2[q] = (f2){0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    const f2 vv = {dv[nt], dv[nt]}, bb = {bv[nt], bv[nt]};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f2 m = silu2((f2){acc[nt][2 * q], acc[nt][2 * q + 1]} + bb);
                        acc[nt][2 * q] = m.x; acc[nt][2 * q + 1] = m.y;
                        part2[q] = m * vv + part2[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { part[2 * q] = part2[q].x; part[2 * q + 1] = part2[q].y; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = half_sum_dpp(part[r]);   // all 32 lanes of the half hold the row sum

            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    part[r] = row < K ? sigmoid_fast(part[r] + p.att_b) : 0.f;   // attention gate; masked rows -> 0
                }
                const bool store_m = p.last && i >= p.R;
                if (store_m) {   // the last layer's launch is bound by these 2.5 GB of HBM writes, not by store issue (measured)
                    uint16_t *Mout = p.mbuf + (((size_t)b * p.L + (i - p.R)) * KPAD) * H;
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            Mout[(size_t)row * H + nt * 32 + l31] = to16<F16>(acc[nt][r] * part[r]);
                        }
                }
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    f2 cs = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) cs = (f2){acc[nt][2 * q], acc[nt][2 * q + 1]} * (f2){part[2 * q], part[2 * q + 1]} + cs;
                    colsum[nt] += cs.x + cs.y;
                }
            } else {
                // coord_mlp: w = clamp(sum_c silu(.) * wc2, +-2); x_i += mean_s (x_i - x_j)/(|x_i - x_j| + 1) * w
                if (l31 == 0) {
                    const float4 *ca = p.ca4 + (size_t)b * p.N;
                    const float4 xi = ca[i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row < K) {
                            const float w = fminf(fmaxf(part[r], -2.0f), 2.0f);
                            const float4 xj = ca[p.edges[ebase + row]];
                            const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
                            const float nrm = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f) + 1.0f;
                            cacc0 += dx / nrm * w; cacc1 += dy / nrm * w; cacc2 += dz / nrm * w;
                        }
                    }
                }
            }
            stamp(11);
            tl_tile += 1;
        }   // mt

        if (MODE == 0) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const float t = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
                if (h == 0) p.agg[node * H + nt * 32 + l31] = t;
            }
        } else {
            cacc0 += __shfl_xor(cacc0, 32, 64);
            cacc1 += __shfl_xor(cacc1, 32, 64);
            cacc2 += __shfl_xor(cacc2, 32, 64);
            if (lane == 0) {
                const float4 xi = p.ca4[node];
                const float inv = 1.0f / (float)(K > 1 ? K : 1);
                float *fo = p.fout + ((size_t)b * p.L + (i - p.R)) * 3;
                fo[0] = (xi.x + cacc0 * inv) - xi.x;
                fo[1] = (xi.y + cacc1 * inv) - xi.y;
                fo[2] = (xi.z + cacc2 * inv) - xi.z;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
static EdgeKArgs to_kargs(const EdgeArgs &a)
{
    EdgeKArgs k;
    k.A = a.A; k.Bm = a.Bm; k.Bmb = a.Bmb; k.ab_bstride = a.ab_bstride;
    k.edges = a.edges; k.codes = a.codes; k.radial = a.radial; k.ca4 = a.ca4;
    k.B = a.B; k.N = a.N; k.R = a.R; k.K = a.K; k.L = a.N - a.R;
    const LayerDev *w = a.lw;
    k.w_r = w->w_r; k.T = w->T; k.W2t = w->W2t; k.b2 = w->b2; k.att_w = w->att_w; k.T2b = w->T2b;
    k.Wf = reinterpret_cast<const uint4 *>(w->W2f); k.att_b = w->att_b;
    k.Wc1t = w->Wc1t; k.bc1 = w->bc1; k.wc2 = w->wc2;
    k.agg = a.agg; k.last = a.last; k.fout = a.fout; k.mbuf = a.mbuf; k.tl = nullptr;
    return k;
}

hipError_t launch_edge_f32(const EdgeArgs &a, hipStream_t s)
{
    static bool attr_set = false;
    const int lds = KF * H * 4 + 4 * 64 * 4;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_f32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const EdgeKArgs k = to_kargs(a);
    hipLaunchKernelGGL(k_edge_f32, dim3((unsigned)((long long)a.B * a.N)), dim3(256), lds, s, k);
    return hipGetLastError();
}

static int persistent_grid(long long wave_tasks)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    long long wgs = (wave_tasks + EDGE_WAVES - 1) / EDGE_WAVES;
    long long g = wgs < cus ? wgs : cus;
    g = (g + 7) / 8 * 8;   // multiple of the XCD count
    return (int)g;
}

// -------------------------------------------------------------------------------------------------
// Edge-message kernel, software-pipelined ACROSS tiles (k_edge_bf16<0,.> restarts cold on every tile:
// metadata load -> dependent gathers -> first SiLU, ~2 L2 latencies exposed per 32 rows).  Here the wave
// walks one flat sequence of tiles (node, m-tile): during the last K-chunk of tile t it loads the row
// metadata of tile t+1 into the registers the gathers no longer need, and it issues tile t+1's first
// chunk of gathers right before tile t's epilogue, so both latencies hide under VALU/MFMA work.
template <int F16, int NW>   // NW waves per workgroup: 8 (two per SIMD, 256 registers) or 4 (one per SIMD, 512 registers)
__global__ __launch_bounds__(NW * 64) void k_edge_msg(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *Wf = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *stage = smem + LDS_WF_BYTES + wave * LDS_STAGE_BYTES;
    const int h = lane >> 5, l31 = lane & 31;
    const int r8 = lane >> 3, c8 = lane & 7;
    for (int q = tid; q < LDS_WF_BYTES / 16; q += NW * 64) Wf[q] = p.Wf[q];
    __syncthreads();   // the only workgroup barrier: everything below is wave-private

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int NT = p.N;
    const int nsplit = p.B >= 8 ? 1 : (8 + p.B - 1) / p.B;
    const int NTc = (NT + nsplit - 1) / nsplit;
    const int U = p.B * nsplit;
    const int nb = U > xcd ? (U - xcd + 7) >> 3 : 0;
    const long long ntask = (long long)nb * NTc;
    const long long tstride = (long long)wg_per_xcd * NW;
    const int K = p.K, ntile = (K + 31) >> 5;

    struct Tile { int b, i, mt; };
    long long tt = (long long)slot * NW + wave - tstride;
    // next node task of this wave (wave-uniform); false when the list is exhausted
    auto next_node = [&](Tile &t) -> bool {
        for (;;) {
            tt += tstride;
            if (tt >= ntask) return false;
            const int u = xcd + 8 * (int)(tt / NTc);
            const int idx = (u % nsplit) * NTc + (int)(tt % NTc);
            if (idx >= NT) continue;
            t.b = __builtin_amdgcn_readfirstlane(u / nsplit);
            t.i = __builtin_amdgcn_readfirstlane(idx);
            t.mt = 0;
            return true;
        }
    };
    auto advance = [&](Tile &t) -> bool {
        if (t.mt + 1 < ntile) { t.mt += 1; return true; }
        return next_node(t);
    };

    Tile cur;
    if (!next_node(cur)) return;

    int jq[4]; uint32_t codeq[4]; float radq[4], radn[4];
    auto load_meta = [&](const Tile &t, float (&rad)[4]) {
        const size_t ebase = ((size_t)t.b * p.N + t.i) * K;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = t.mt * 32 + q * 8 + r8;
            const bool v = s < K;
            jq[q] = v ? p.edges[ebase + s] : t.i;
            codeq[q] = v ? p.codes[ebase + s] : 0u;
            rad[q] = v ? p.radial[ebase + s] : 0.f;
        }
    };
    float4 a0, a1, w0, w1;
    auto gather_chunk = [&](const Tile &t, int kc) {
        const float *Arow = p.A + (size_t)t.b * p.ab_bstride + (size_t)t.i * H + kc * 64 + c8 * 8;
        a0 = *reinterpret_cast<const float4 *>(Arow);
        a1 = *reinterpret_cast<const float4 *>(Arow + 4);
        w0 = *reinterpret_cast<const float4 *>(p.w_r + kc * 64 + c8 * 8);
        w1 = *reinterpret_cast<const float4 *>(p.w_r + kc * 64 + c8 * 8 + 4);
    };
    auto gather = [&](const Tile &t, int kc, int q, RawP &r) {
        const uint32_t ch = kc * 64 + c8 * 8;
        const uint32_t code = codeq[q];
        r.bm = *reinterpret_cast<const uint4 *>(p.Bmb + (size_t)t.b * p.ab_bstride + ((uint32_t)jq[q] * H + ch));
        const uint32_t i0 = (((code >> 6) & 31u) * 24u + ((code >> 11) & 31u)) * H;
        const uint32_t i1 = (576u + ((code >> 16) & 15u) * 40u + (code & 63u)) * H;
        const uint32_t i2 = (1056u + ((code >> 20) & 127u)) * H;
        r.t0 = *reinterpret_cast<const uint4 *>(p.T2b + (i0 + ch));
        r.t1 = *reinterpret_cast<const uint4 *>(p.T2b + (i1 + ch));
        r.t2 = *reinterpret_cast<const uint4 *>(p.T2b + (i2 + ch));
    };
    auto compute_store = [&](int q, const RawP &r) {
        const f2 rad2 = {radq[q], radq[q]};
        f2 v[4] = {(f2){w0.x, w0.y} * rad2 + (f2){a0.x, a0.y}, (f2){w0.z, w0.w} * rad2 + (f2){a0.z, a0.w},
                   (f2){w1.x, w1.y} * rad2 + (f2){a1.x, a1.y}, (f2){w1.z, w1.w} * rad2 + (f2){a1.z, a1.w}};
        H8 t, t1, t2, bm;
        t.u = r.t0; t1.u = r.t1; t2.u = r.t2; bm.u = r.bm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t.h[e] = __hadd2(__hadd2(t.h[e], t1.h[e]), t2.h[e]);
            if constexpr (!F16) t.h[e] = __hadd2(t.h[e], bm.h[e]);
            else v[e] = v[e] + (f2){__low2float(bm.h[e]), __high2float(bm.h[e])};
            v[e] = v[e] + (f2){__low2float(t.h[e]), __high2float(t.h[e])};
        }
        Frag f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f2 m = silu2(v[e]);
            if constexpr (F16) { f.f[2 * e] = (_Float16)fminf(m.x, 65504.f); f.f[2 * e + 1] = (_Float16)fminf(m.y, 65504.f); }
            else { f.b[2 * e] = (__bf16)m.x; f.b[2 * e + 1] = (__bf16)m.y; }
        }
        const int row = q * 8 + r8;
        *reinterpret_cast<uint4 *>(stage + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = f.u;
    };

    RawP r0, r1, r2, r3;
    load_meta(cur, radq);
    gather_chunk(cur, 0);
    gather(cur, 0, 0, r0); gather(cur, 0, 1, r1); gather(cur, 0, 2, r2); gather(cur, 0, 3, r3);
    float colsum[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) colsum[nt] = 0.f;

    for (;;) {
        f32x16 acc[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float bias = p.b2[nt * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = bias;
        }
        auto mfma_chunk = [&](int kc) {
            wave_lds_fence();
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                Frag af;
                af.u = *reinterpret_cast<const uint4 *>(stage + l31 * 128 + (((kq * 2 + h) ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    Frag bf;
                    bf.u = Wf[((kc * 4 + kq) * 8 + nt) * 64 + lane];
                    acc[nt] = mfma16<F16>(af, bf, acc[nt]);
                }
            }
            wave_lds_fence();
        };
#pragma unroll 1
        for (int kc = 0; kc < 3; ++kc) {
            compute_store(0, r0); gather(cur, kc + 1, 0, r0);
            compute_store(1, r1); gather(cur, kc + 1, 1, r1);
            compute_store(2, r2); gather(cur, kc + 1, 2, r2);
            compute_store(3, r3); gather(cur, kc + 1, 3, r3);
            gather_chunk(cur, kc + 1);
            mfma_chunk(kc);
        }
        // last chunk: jq / codeq are dead (all gathers of this tile are issued) -> fetch the next tile's metadata
        Tile nxt = cur;
        const bool has_next = advance(nxt);
        if (has_next) load_meta(nxt, radn);
        compute_store(0, r0); compute_store(1, r1); compute_store(2, r2); compute_store(3, r3);
        mfma_chunk(3);
        if (has_next) {   // half of the next tile's first chunk flies under this tile's epilogue (register budget)
            gather_chunk(nxt, 0);
            gather(nxt, 0, 0, r0); gather(nxt, 0, 1, r1);
            if constexpr (NW == 4) { gather(nxt, 0, 2, r2); gather(nxt, 0, 3, r3); }
        }

        // ---- epilogue: lane owns columns nt*32 + l31, rows rowof(r) ------------------------------------
        float part[16];
        {
            f2 part2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) part2[q] = (f2){0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const float vs = p.att_w[nt * 32 + l31];
                const f2 vv = {vs, vs};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f2 m = silu2((f2){acc[nt][2 * q], acc[nt][2 * q + 1]});
                    acc[nt][2 * q] = m.x; acc[nt][2 * q + 1] = m.y;
                    part2[q] = m * vv + part2[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { part[2 * q] = part2[q].x; part[2 * q + 1] = part2[q].y; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = cur.mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float g = half_sum_dpp(part[r]);
            part[r] = row < K ? sigmoid_fast(g + p.att_b) : 0.f;   // attention gate; masked rows -> 0
        }
        if (p.last && cur.i >= p.R) {
            uint16_t *Mout = p.mbuf + (((size_t)cur.b * p.L + (cur.i - p.R)) * KPAD) * H;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = cur.mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Mout[(size_t)row * H + nt * 32 + l31] = to16<F16>(acc[nt][r] * part[r]);
                }
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            f2 cs = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) cs = (f2){acc[nt][2 * q], acc[nt][2 * q + 1]} * (f2){part[2 * q], part[2 * q + 1]} + cs;
            colsum[nt] += cs.x + cs.y;
        }
        if (cur.mt == ntile - 1) {   // node complete: fixed-degree segment sum -> agg
            const size_t node = (size_t)cur.b * p.N + cur.i;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const float t = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
                if (h == 0) p.agg[node * H + nt * 32 + l31] = t;
                colsum[nt] = 0.f;
            }
        }
        if (!has_next) break;
        cur = nxt;
#pragma unroll
        for (int q = 0; q < 4; ++q) radq[q] = radn[q];
        if constexpr (NW != 4) { gather(cur, 0, 2, r2); gather(cur, 0, 3, r3); }
    }
}

// -------------------------------------------------------------------------------------------------
// "Weights in registers" edge-message kernel.
//
// The per-wave 32x256 tile of k_edge_bf16 needs 128 accumulator registers and the whole weight matrix in
// LDS, which leaves two waves per SIMD starved of registers (no read-ahead, exposed LDS latency).  Here the
// roles are turned around:
//   * wave w of the 8-wave workgroup owns 32 OUTPUT channels for every node; its slice of the weight
//     matrix (16 k-steps x 16 B/lane = 64 VGPRs) is loaded once and stays in registers for the launch;
//   * the workgroup processes one node (64 rows: 60 edges + 4 masked) at a time.  Every wave produces 8 rows
//     x 256 channels of the first activation (gather layout, whole 128-B lines) into a double-buffered,
//     XOR-swizzled LDS tile [64][256] that all eight waves then consume;
//   * the product is computed TRANSPOSED, D[channel][edge] = W[channel][k] * a1[edge][k]^T (weights as the
//     MFMA A operand, the staged tile as B), so a lane holds 16 contiguous channels of ONE edge: the
//     attention dot product is in-lane (+ one 8-way exchange through LDS), accumulators shrink to 32
//     registers, and the 60-row segment sum is 16 DPP row reductions per wave;
//   * one s_barrier per node: MFMAs of node n are issued interleaved with the producer VALU work of node
//     n+1 (other half of the double buffer); gathers run one node ahead, row metadata two nodes ahead.
constexpr int WR_WAVES = 8;
constexpr int WR_TILE_BYTES = 64 * 256 * 2;                       // one staged node
constexpr int WR_LDS_BYTES = 2 * WR_TILE_BYTES + 2 * 8 * 64 * 4;  // + double-buffered gate partials

template <int F16>
__global__ __launch_bounds__(WR_WAVES * 64) void k_edge_wr(EdgeKArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *gate_part = reinterpret_cast<float *>(smem + 2 * WR_TILE_BYTES);   // [2][8 waves][64 edges]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int prow = wave * 8 + r8;                  // the ONE row of the node this lane produces (all four passes)
    const int K = p.K;

    // resident weight slice: A-operand fragments, row rho <-> channel 32w + 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3)
    uint4 Wreg[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) Wreg[kk] = p.Wf[(kk * 8 + wave) * 64 + lane];
    const int ch0 = wave * 32 + h * 16;              // this lane's 16 contiguous output channels

    // node tasks of this workgroup (XCD-aware order, see k_edge_bf16)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int NT = p.N;
    const int nsplit = p.B >= 8 ? 1 : (8 + p.B - 1) / p.B;
    const int NTc = (NT + nsplit - 1) / nsplit;
    const int U = p.B * nsplit;
    const int nb = U > xcd ? (U - xcd + 7) >> 3 : 0;
    const unsigned ntask = (unsigned)nb * (unsigned)NTc;
    struct Node { int b, i; bool ok; };
    auto node_at = [&](unsigned it) -> Node {        // it-th node of this workgroup; ok = false past the end / padding
        const unsigned tt = (unsigned)slot + it * (unsigned)wg_per_xcd;
        Node n{0, 0, false};
        if (tt >= ntask) return n;
        const unsigned tq = tt / (unsigned)NTc, tr = tt - tq * (unsigned)NTc;
        const int u = xcd + 8 * (int)tq;
        const int idx = (u % nsplit) * NTc + (int)tr;
        n.b = u / nsplit; n.i = idx; n.ok = idx < NT;
        return n;
    };
    const unsigned niter = ntask > (unsigned)slot ? (ntask - slot + wg_per_xcd - 1) / wg_per_xcd : 0;
    if (niter == 0) return;

    struct Meta { int j; uint32_t code; float rad; };
    auto load_meta = [&](const Node &n) -> Meta {
        Meta m{n.i, 0u, 0.f};
        if (n.ok && prow < K) {
            const size_t e = ((size_t)n.b * p.N + n.i) * K + prow;
            m.j = p.edges[e]; m.code = p.codes[e]; m.rad = p.radial[e];
        }
        return m;
    };
    auto gather = [&](const Node &n, const Meta &m, int q, RawP &r) {
        const uint32_t ch = q * 64 + c8 * 8;
        const size_t ab = (size_t)n.b * p.ab_bstride;
        r.bm = *reinterpret_cast<const uint4 *>(p.Bmb + ab + ((uint32_t)m.j * H + ch));
        const uint32_t code = m.code;
        const uint32_t i0 = (((code >> 6) & 31u) * 24u + ((code >> 11) & 31u)) * H;
        const uint32_t i1 = (576u + ((code >> 16) & 15u) * 40u + (code & 63u)) * H;
        const uint32_t i2 = (1056u + ((code >> 20) & 127u)) * H;
        r.t0 = *reinterpret_cast<const uint4 *>(p.T2b + (i0 + ch));
        r.t1 = *reinterpret_cast<const uint4 *>(p.T2b + (i1 + ch));
        r.t2 = *reinterpret_cast<const uint4 *>(p.T2b + (i2 + ch));
    };
    // one producer pass: 8 channels of this lane's row -> SiLU -> 16-bit -> staged tile
    auto compute_store = [&](const Node &n, float rad, int q, const RawP &r, char *tile) {
        const float *Arow = p.A + (size_t)n.b * p.ab_bstride + (size_t)n.i * H + q * 64 + c8 * 8;
        const float4 a0 = *reinterpret_cast<const float4 *>(Arow), a1 = *reinterpret_cast<const float4 *>(Arow + 4);
        const float4 w0 = *reinterpret_cast<const float4 *>(p.w_r + q * 64 + c8 * 8);
        const float4 w1 = *reinterpret_cast<const float4 *>(p.w_r + q * 64 + c8 * 8 + 4);
        const f2 rad2 = {rad, rad};
        f2 v[4] = {(f2){w0.x, w0.y} * rad2 + (f2){a0.x, a0.y}, (f2){w0.z, w0.w} * rad2 + (f2){a0.z, a0.w},
                   (f2){w1.x, w1.y} * rad2 + (f2){a1.x, a1.y}, (f2){w1.z, w1.w} * rad2 + (f2){a1.z, a1.w}};
        H8 t, t1, t2, bm;
        t.u = r.t0; t1.u = r.t1; t2.u = r.t2; bm.u = r.bm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t.h[e] = __hadd2(__hadd2(t.h[e], t1.h[e]), t2.h[e]);
            if constexpr (!F16) t.h[e] = __hadd2(t.h[e], bm.h[e]);
            else v[e] = v[e] + (f2){__low2float(bm.h[e]), __high2float(bm.h[e])};
            v[e] = v[e] + (f2){__low2float(t.h[e]), __high2float(t.h[e])};
        }
        Frag f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f2 m = silu2(v[e]);
            if constexpr (F16) { f.f[2 * e] = (_Float16)fminf(m.x, 65504.f); f.f[2 * e + 1] = (_Float16)fminf(m.y, 65504.f); }
            else { f.b[2 * e] = (__bf16)m.x; f.b[2 * e + 1] = (__bf16)m.y; }
        }
        *reinterpret_cast<uint4 *>(tile + prow * 512 + (((q * 8 + c8) ^ (prow & 15)) << 4)) = f.u;
    };

    // ---- prologue: stage node 0, start the pipeline --------------------------------------------------
    Node n0 = node_at(0), n1 = node_at(1), n2 = node_at(2);
    Meta m0 = load_meta(n0), m1 = load_meta(n1), m2 = load_meta(n2);
    RawP r0, r1, r2, r3;
    gather(n0, m0, 0, r0); gather(n0, m0, 1, r1); gather(n0, m0, 2, r2); gather(n0, m0, 3, r3);
    compute_store(n0, m0.rad, 0, r0, smem); gather(n1, m1, 0, r0);
    compute_store(n0, m0.rad, 1, r1, smem); gather(n1, m1, 1, r1);
    compute_store(n0, m0.rad, 2, r2, smem); gather(n1, m1, 2, r2);
    compute_store(n0, m0.rad, 3, r3, smem); gather(n1, m1, 3, r3);
    __syncthreads();

    Node cur = n0, nx1 = n1, nx2 = n2;         // node being contracted, node being produced, node being gathered
    Meta mx1 = m1, mx2 = m2;
    for (unsigned it = 0; it < niter; ++it) {
        char *tile_c = smem + (it & 1) * WR_TILE_BYTES, *tile_n = smem + ((it + 1) & 1) * WR_TILE_BYTES;
        float *gp = gate_part + (it & 1) * 512;
        const Node nx3 = node_at(it + 3);
        const Meta mx3 = load_meta(nx3);            // metadata runs two nodes ahead of its gathers' consumers

        // ---- contraction of `cur` (32 MFMAs) interleaved with the producer passes of `nx1` -----------------
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        auto mfma_quarter = [&](int q4) {
#pragma unroll
            for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int row = mt * 32 + l31;
                    Frag bf, af;
                    bf.u = *reinterpret_cast<const uint4 *>(tile_c + row * 512 + (((kk * 2 + h) ^ (row & 15)) << 4));
                    af.u = Wreg[kk];
                    acc[mt] = mfma16<F16>(af, bf, acc[mt]);
                }
        };
        // (padding tasks can sit in the middle of the list, so every stage is predicated on its own node)
        if (nx1.ok) compute_store(nx1, mx1.rad, 0, r0, tile_n);
        gather(nx2, mx2, 0, r0); mfma_quarter(0);
        if (nx1.ok) compute_store(nx1, mx1.rad, 1, r1, tile_n);
        gather(nx2, mx2, 1, r1); mfma_quarter(1);
        if (nx1.ok) compute_store(nx1, mx1.rad, 2, r2, tile_n);
        gather(nx2, mx2, 2, r2); mfma_quarter(2);
        if (nx1.ok) compute_store(nx1, mx1.rad, 3, r3, tile_n);
        gather(nx2, mx2, 3, r3); mfma_quarter(3);

        // ---- epilogue part 1: bias, SiLU, partial attention logits of this wave's 32 channels --------------
        float4 bq[4], dq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bq[q] = *reinterpret_cast<const float4 *>(p.b2 + ch0 + q * 4);
            dq[q] = *reinterpret_cast<const float4 *>(p.att_w + ch0 + q * 4);
        }
        float pg[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f2 s2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 ma = silu2((f2){acc[mt][4 * q], acc[mt][4 * q + 1]} + (f2){bq[q].x, bq[q].y});
                const f2 mb = silu2((f2){acc[mt][4 * q + 2], acc[mt][4 * q + 3]} + (f2){bq[q].z, bq[q].w});
                acc[mt][4 * q] = ma.x; acc[mt][4 * q + 1] = ma.y; acc[mt][4 * q + 2] = mb.x; acc[mt][4 * q + 3] = mb.y;
                s2 = ma * (f2){dq[q].x, dq[q].y} + s2;
                s2 = mb * (f2){dq[q].z, dq[q].w} + s2;
            }
            pg[mt] = s2.x + s2.y;
            pg[mt] += __shfl_xor(pg[mt], 32, 64);
            if (h == 0) gp[wave * 64 + mt * 32 + l31] = pg[mt];
        }
        __syncthreads();   // tile_n complete, gate partials complete, every wave done reading tile_c

        // ---- epilogue part 2: gate, fixed-degree segment sum -----------------------------------------------
        if (cur.ok) {
            f2 s2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) s2[q] = (f2){0.f, 0.f};
            const bool store_m = p.last && cur.i >= p.R;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int edge = mt * 32 + l31;
                float g = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) g += gp[w2 * 64 + edge];
                g = edge < K ? sigmoid_fast(g + p.att_b) : 0.f;
                const f2 g2 = {g, g};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f2 mg = (f2){acc[mt][2 * q], acc[mt][2 * q + 1]} * g2;
                    acc[mt][2 * q] = mg.x; acc[mt][2 * q + 1] = mg.y;
                    s2[q] = s2[q] + mg;
                }
                if (store_m) {
                    Frag o0, o1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (F16) {
                            o0.f[e] = (_Float16)fminf(fmaxf(acc[mt][e], -65504.f), 65504.f);
                            o1.f[e] = (_Float16)fminf(fmaxf(acc[mt][8 + e], -65504.f), 65504.f);
                        } else { o0.b[e] = (__bf16)acc[mt][e]; o1.b[e] = (__bf16)acc[mt][8 + e]; }
                    }
                    uint16_t *Mo = p.mbuf + ((((size_t)cur.b * p.L + (cur.i - p.R)) * KPAD) + edge) * H + ch0;
                    *reinterpret_cast<uint4 *>(Mo) = o0.u;
                    *reinterpret_cast<uint4 *>(Mo + 8) = o1.u;
                }
            }
            float sum[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) { sum[2 * q] = half_sum_dpp(s2[q].x); sum[2 * q + 1] = half_sum_dpp(s2[q].y); }
            if (l31 == 0) {
                float *ao = p.agg + ((size_t)cur.b * p.N + cur.i) * H + ch0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(ao + q * 4) = make_float4(sum[4 * q], sum[4 * q + 1], sum[4 * q + 2], sum[4 * q + 3]);
            }
        }
        cur = nx1; nx1 = nx2; nx2 = nx3;
        mx1 = mx2; mx2 = mx3;
    }
}

template <int MODE, int F16> static hipError_t launch_mfma_t(const EdgeKArgs &k, long long wave_tasks, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_bf16<MODE, F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_EDGE_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_edge_bf16<MODE, F16>), dim3(persistent_grid(wave_tasks)), dim3(EDGE_WAVES * 64), LDS_EDGE_BYTES, s, k);
    return hipGetLastError();
}

static int persistent_grid_nw(long long wave_tasks, int nw)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    long long wgs = (wave_tasks + nw - 1) / nw;
    long long g = wgs < cus ? wgs : cus;
    return (int)((g + 7) / 8 * 8);
}

template <int F16, int NW> static hipError_t launch_msg_t(const EdgeKArgs &k, long long wave_tasks, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_msg<F16, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_EDGE_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_edge_msg<F16, NW>), dim3(persistent_grid_nw(wave_tasks, NW)), dim3(NW * 64), LDS_EDGE_BYTES, s, k);
    return hipGetLastError();
}

template <int F16> static hipError_t launch_wr_t(const EdgeKArgs &k, long long nodes, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_wr<F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    long long g = nodes < cus ? nodes : cus;
    g = (g + 7) / 8 * 8;
    hipLaunchKernelGGL((k_edge_wr<F16>), dim3((unsigned)g), dim3(WR_WAVES * 64), WR_LDS_BYTES, s, k);
    return hipGetLastError();
}

hipError_t launch_edge_bf16(const EdgeArgs &a, hipStream_t s)
{
    // DFM_EDGE_KERNEL (A/B timing): "wr" = weights-in-registers kernel, "tile" = per-wave 32x256 tile kernel,
    // "pipe8"/"pipe4" = cross-tile pipelined tile kernel
    static const int which = [] {
        const char *e = getenv("DFM_EDGE_KERNEL");
        if (!e) return 0;
        if (!strcmp(e, "wr")) return 1;
        if (!strcmp(e, "pipe8")) return 8;
        if (!strcmp(e, "pipe4")) return 4;
        return 0;
    }();
    EdgeKArgs k = to_kargs(a);
    const long long tasks = (long long)a.B * a.N;
    if (which == 1) {
        k.Wf = reinterpret_cast<const uint4 *>(a.f16 ? a.lw->W2t16 : a.lw->W2tb);
        return a.f16 ? launch_wr_t<1>(k, tasks, s) : launch_wr_t<0>(k, tasks, s);
    }
    if (a.f16) k.Wf = reinterpret_cast<const uint4 *>(a.lw->W2f16);
    if (which == 8) return a.f16 ? launch_msg_t<1, 8>(k, tasks, s) : launch_msg_t<0, 8>(k, tasks, s);
    if (which == 4) return a.f16 ? launch_msg_t<1, 4>(k, tasks, s) : launch_msg_t<0, 4>(k, tasks, s);
    return a.f16 ? launch_mfma_t<0, 1>(k, tasks, s) : launch_mfma_t<0, 0>(k, tasks, s);
}

hipError_t launch_coord_bf16(const EdgeArgs &a, hipStream_t s)
{
    EdgeKArgs k = to_kargs(a);
    k.Wf = reinterpret_cast<const uint4 *>(a.f16 ? a.lw->Wc1f16 : a.lw->Wc1f);
    const long long tasks = (long long)a.B * (a.N - a.R);
    return a.f16 ? launch_mfma_t<1, 1>(k, tasks, s) : launch_mfma_t<1, 0>(k, tasks, s);
}

}  // namespace dfm
