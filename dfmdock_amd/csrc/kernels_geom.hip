// kernels_geom.hip - pose preparation, kNN + inverse-cubic edge sampling, per-edge 6-D feature bins,
// initial pose randomisation, clash force.  HBM/latency-bound integer + fp32 geometry (no MFMA here).
//
// Compiled with -ffp-contract=off: the feature bins and the kNN order are discontinuous functions of
// fp32 geometry, so every operation is rounded exactly like the reference's op-by-op torch evaluation
// (src/utils/coords6d.py, src/models/score_net_mlsb.py:30-135).
#include "dfm_device.h"
#include "dfm_internal.h"

namespace dfm {

// ------------------------------------------------------------------------------------------------
// prep_pose: centre receptor + ligand on the ligand CA centroid (score_net_mlsb.py:353-359), build
// CA and virtual-CB arrays (coords6d.py:71-75).  One workgroup per trajectory.
__global__ __launch_bounds__(256) void k_prep_pose(const float *__restrict__ rec_pos, const float *__restrict__ lig_cur,
                                                   int R, int L, int all_atoms, float4 *__restrict__ n4,
                                                   float4 *__restrict__ ca4, float4 *__restrict__ cb4)
{
    __shared__ double scratch[8];
    __shared__ float center[3];
    const int b = blockIdx.x, N = R + L;
    prep_pose_block(rec_pos, lig_cur + (size_t)b * L * 9, R, L, all_atoms, n4 + (size_t)b * N, ca4 + (size_t)b * N, cb4 + (size_t)b * N,
                    scratch, center);
}

hipError_t launch_prep_pose(const float *rec_pos, const float *lig_cur, int B, int R, int L, int all_atoms, float4 *n4, float4 *ca4,
                            float4 *cb4, hipStream_t s)
{
    hipLaunchKernelGGL(k_prep_pose, dim3(B), dim3(256), 0, s, rec_pos, lig_cur, R, L, all_atoms, n4, ca4, cb4);
    return hipGetLastError();
}

// wave-wide unsigned minimum, result wave-uniform: 4 DPP steps inside each 16-lane row, ds_swizzle across the row
// pair, then the two 32-lane halves meet in scalar registers
__device__ inline uint32_t wave_min_u32(uint32_t v)
{
    auto step = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    v = step(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));                           // lane ^ 16
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    return a < b ? a : b;
}

// ------------------------------------------------------------------------------------------------
// kNN(20) + sample(40, p ~ 1/d^3, without replacement) (score_net_mlsb.py:85-135): the pairwise C-alpha distance scan and the
// radial-graph edge build.  One 64-lane wave per (trajectory, node); the candidates' coordinates are read straight from global
// memory (the C-alpha array of a trajectory is 9.6 KB at N = 600, 32 KB at N = 2000: L1 / L2 resident).  An LDS-tiled form of the
// scan was built, parity-tested and measured slower in r02 (profiles/r02_exp_knn_lds.txt).  The lane owns candidates
// j = 4*(lane + 64*q) + e (q < NPL/4, e < 4) in registers.
//
// Both selections - the knn nearest (ascending distance, lowest index first on ties, slot 0 = the node itself) and the nsamp
// smallest race keys key_j = Exp(1)_j * d_j^3 (= successive sampling without replacement with p ~ d^-3, the scheme
// torch.multinomial uses; Exp(1) from Philox4x32-10) - are THRESHOLD selections: non-negative floats order like their bit
// patterns, so the k-th smallest word is found bit by bit from the top with one wave-wide count (NPL compares + scalar popcounts)
// per bit, stopping as soon as a threshold cuts off exactly k words (about 15-20 bits in practice); ties at the k-th value go to the
// lowest candidate index.  The kNN winners are then ranked among themselves through LDS (20 x 20 comparisons) for the sorted order
// the reference's topk gives; sampled slots keep candidate order (their order carries no meaning).  r02 ran knn + nsamp = 60
// wave-wide arg-min passes per node (the kernel was VALU-issue-bound: 4.2 k instructions per node at N = 600, 8.5 k at N = 2000,
// profiles/r03_a_pmc_knn_*.txt); this form needs about a third of that.
template <int NPL> __device__ inline uint32_t knn_count_lt(const uint32_t (&key)[NPL], uint32_t t)
{
    uint32_t c = 0;
    if constexpr (NPL < 32) {      // ballots + scalar popcounts: the scalar unit works beside the vector pipe (kernel x0.50 at N = 600)
#pragma unroll
        for (int r = 0; r < NPL; ++r) c += (uint32_t)__builtin_popcountll(__ballot(key[r] < t));
        return c;
    }
    // NPL >= 32: 2 NPL live scalar registers spill (86 at NPL = 32); per-lane count on the vector pipe, then ONE wave sum
    // (kernel x0.70 at N = 2000 against x0.79 with ballots)
#pragma unroll
    for (int r = 0; r < NPL; ++r) c += key[r] < t ? 1u : 0u;
    c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true);    // row_half_mirror
    c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x140, 0xF, 0xF, true);    // row_mirror
    c += (uint32_t)__builtin_amdgcn_ds_swizzle((int)c, 0x401F);                      // lane ^ 16
    return (uint32_t)__builtin_amdgcn_readlane((int)c, 0) + (uint32_t)__builtin_amdgcn_readlane((int)c, 32);      // wave-uniform
}
__device__ inline uint32_t knn_cand(int lane, int r) { return 4u * (uint32_t)lane + 256u * (uint32_t)(r >> 2) + (uint32_t)(r & 3); }
// bit r of the result: slot r of this lane is among the k smallest (key, candidate index) pairs of the wave (k <= valid keys)
template <int NPL> __device__ inline uint64_t knn_select(const uint32_t (&key)[NPL], int k, int lane)
{
    uint32_t prefix = 0, thr = 0;
    bool exact = false;
    for (int bit = 30; bit >= 0; --bit) {      // keys are non-negative floats: bit 31 is clear
        const uint32_t t = prefix | (1u << bit);
        const uint32_t c = knn_count_lt<NPL>(key, t);
        if (c == (uint32_t)k) { thr = t; exact = true; break; }
        if (c < (uint32_t)k) prefix = t;       // the k-th smallest word is >= t
    }
    uint64_t sel = 0;
    if (exact) {
#pragma unroll
        for (int r = 0; r < NPL; ++r) sel |= (uint64_t)(key[r] < thr) << r;
        return sel;
    }
    // prefix = the k-th smallest word itself: everything below it, then the ties in ascending candidate order (rare)
    uint64_t tie = 0;
#pragma unroll
    for (int r = 0; r < NPL; ++r) { sel |= (uint64_t)(key[r] < prefix) << r; tie |= (uint64_t)(key[r] == prefix) << r; }
    asm volatile("" : "+v"(sel), "+v"(tie));
    int need = k - (int)knn_count_lt<NPL>(key, prefix);
    while (need > 0) {
        uint32_t jb = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < NPL; ++r) { const uint32_t j = knn_cand(lane, r); jb = ((tie >> r) & 1) && j < jb ? j : jb; }
        const uint32_t win = wave_min_u32(jb);
#pragma unroll
        for (int r = 0; r < NPL; ++r)
            if (((tie >> r) & 1) && knn_cand(lane, r) == win) { sel |= 1ull << r; tie &= ~(1ull << r); }
        --need;
    }
    return sel;
}
__device__ inline int lanes_below(unsigned long long m)      // set bits of m below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <int NPL>
__global__ __launch_bounds__(256) void k_knn_sample(const float4 *__restrict__ ca4, int B, int N, int knn, int nsamp,
                                                    uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                                                    int32_t *__restrict__ edges, const uint32_t *__restrict__ ctl)
{
    // ctl (replayed step graph, api.hip: StepCtl): the evaluation index and the call's seed come from device memory, so that the
    // captured launch is the same node in every step and every call
    if (ctl) { stream_id = ctl[0]; seed_lo = ctl[1]; seed_hi = ctl[2]; }
    __shared__ uint32_t sh_key[4][64], sh_j[4][64];
    __shared__ int32_t sh_out[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long node = (long long)blockIdx.x * 4 + w;
    if (node >= (long long)B * N) return;   // whole wave exits together; no block-level sync below
    const int b = (int)((unsigned long long)node / (unsigned)N), i = (int)(node - (long long)b * N);
    const float4 *ca = ca4 + (size_t)b * N;
    const float4 ci = ca[i];
    const int K = knn + nsamp;

    uint32_t key[NPL];      // first the distance bits, then (sampling) the race-key bits of the same candidates
#pragma unroll
    for (int q = 0; q < NPL / 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * (lane + 64 * q) + e;
            float d = __builtin_inff();
            if (j < N) {
                const float4 cj = ca[j];
                const float dx = ci.x - cj.x, dy = ci.y - cj.y, dz = ci.z - cj.z;
                d = sqrtf((dx * dx + dy * dy) + dz * dz);
            }
            key[q * 4 + e] = __float_as_uint(d);
        }
        // (keeps the scheduler from hoisting all NPL coordinate loads - 4 registers each - to the top: at NPL = 32 that alone is 128)
        if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- kNN: the knn smallest (distance, j), written in ascending order
    uint64_t near = knn_select<NPL>(key, knn, lane);
    asm volatile("" : "+v"(near));      // one packed mask, not NPL separate predicates kept alive across the sampling stage
    {
        int base = 0;
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
            const bool s = (near >> r) & 1;
            const unsigned long long m = __ballot(s);
            if (s) { const int pos = base + lanes_below(m); sh_key[w][pos] = key[r]; sh_j[w][pos] = knn_cand(lane, r); }
            base += (int)__builtin_popcountll(m);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (NPL ballots hoisted to the top spill scalar registers)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave: the writes above are visible to the reads below
        const uint32_t myk = sh_key[w][lane & 31], myj = sh_j[w][lane & 31];      // knn <= 20
        int rank = 0;
        for (int e = 0; e < knn; ++e) {
            const uint32_t ke = sh_key[w][e], je = sh_j[w][e];
            rank += (ke < myk || (ke == myk && je < myj)) ? 1 : 0;
        }
        if (lane < knn) sh_out[w][rank] = (int32_t)myj;
    }
    if (nsamp > 0) {
        // ---- sampling: race keys for every candidate not taken above; the nsamp smallest win
#pragma unroll
        for (int q = 0; q < NPL / 4; ++q) {
            const u32x4 rnd = philox4x32((uint32_t)node, (uint32_t)(node >> 32) ^ ((uint32_t)(lane + 64 * q) << 8),
                                         stream_id, RNG_EDGES, seed_lo, seed_hi);
            const uint32_t rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = q * 4 + e;
                float d = __uint_as_float(key[r]);
                float k2 = __builtin_inff();
                if (!((near >> r) & 1) && d != __builtin_inff()) {   // not taken by kNN, j < N
                    d = d < 1e-10f ? 1e-10f : d;
                    k2 = -__builtin_amdgcn_logf(u01(rr[e])) * ((d * d) * d);      // v_log_f32 (log2): a common factor ln 2 does not change the race
                }
                key[r] = __float_as_uint(k2);
            }
            __builtin_amdgcn_sched_barrier(0);      // one Philox block at a time: its temporaries are not multiplied by NPL / 4
        }
        uint64_t drawn = knn_select<NPL>(key, nsamp, lane);
        asm volatile("" : "+v"(drawn));
        int base = knn;
#pragma unroll 1
        for (int r = 0; r < NPL; ++r) {      // the mask is all this loop needs: no register array, no unrolling
            const bool s = (drawn >> r) & 1;
            const unsigned long long m = __ballot(s);
            if (s) sh_out[w][base + lanes_below(m)] = (int32_t)knn_cand(lane, r);
            base += (int)__builtin_popcountll(m);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < K) edges[((size_t)b * N + i) * K + lane] = sh_out[w][lane];
}

hipError_t launch_knn_sample(const float4 *ca4, int B, int N, int knn, int nsamp, uint64_t seed, uint32_t stream_id,
                             int32_t *edges, const uint32_t *ctl, hipStream_t s)
{
    const long long nodes = (long long)B * N;
    const dim3 grid((unsigned)((nodes + 3) / 4)), block(256);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
#define LAUNCH(NPL) hipLaunchKernelGGL(k_knn_sample<NPL>, grid, block, 0, s, ca4, B, N, knn, nsamp, lo, hi, stream_id, edges, ctl)
    if (N <= 256) LAUNCH(4);
    else if (N <= 512) LAUNCH(8);
    else if (N <= 768) LAUNCH(12);
    else if (N <= 1024) LAUNCH(16);
    else if (N <= 2048) LAUNCH(32);
    else LAUNCH(64);
#undef LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-edge trRosetta 6-D features -> bins (coords6d.py:10-103, score_net_mlsb.py:30-70), relpos
// (inference_base.py:255-292) and the EGNN radial |x_i - x_j|^2 (egnn.py:139-148).  One thread per edge.
struct v3 { float x, y, z; };
__device__ inline v3 vsub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline v3 vcross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ inline float vnorm(v3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ inline v3 vdivs(v3 a, float s) { return v3{a.x / s, a.y / s, a.z / s}; }

__device__ inline float dihedral_deg(v3 a, v3 b, v3 c, v3 d)
{   // coords6d.py:25-43
    const v3 b1 = vsub(a, b), b2 = vsub(b, c), b3 = vsub(c, d);
    v3 n1 = vcross(b1, b2); n1 = vdivs(n1, vnorm(n1));
    v3 n2 = vcross(b2, b3); n2 = vdivs(n2, vnorm(n2));
    const v3 m1 = vcross(n1, vdivs(b2, vnorm(b2)));
    return atan2f(vdot(m1, n2), vdot(n1, n2)) * 180.0f / 3.14159265358979323846f;
}
__device__ inline float planar_deg(v3 a, v3 b, v3 c)
{   // coords6d.py:46-58
    const v3 v1 = vsub(a, b), v2 = vsub(c, b);
    return acosf(vdot(v1, v2) / (vnorm(v1) * vnorm(v2))) * 180.0f / 3.14159265358979323846f;
}
// torch.linspace(-180, 180, 23) as float32 (golden: tests/golden/scalar_kats.npz angle boundaries)
__constant__ float c_angle_bounds[23] = {
    -180.0f, -163.63636779785156f, -147.27273559570312f, -130.90908813476562f, -114.54545593261719f,
    -98.18182373046875f, -81.81818389892578f, -65.45454406738281f, -49.090911865234375f,
    -32.72727584838867f, -16.36363983154297f, 3.814697265625e-06f, 16.36363983154297f,
    32.72727584838867f, 49.090911865234375f, 65.45454406738281f, 81.81818389892578f, 98.18182373046875f,
    114.54545593261719f, 130.90908813476562f, 147.27273559570312f, 163.63636779785156f, 180.0f};

__device__ inline int bin_angle(float a)
{   // #(a > boundary); NaN compares false -> bin 0
    int b = 0;
#pragma unroll
    for (int i = 0; i < 23; ++i) b += (a > c_angle_bounds[i]) ? 1 : 0;
    return b;
}

// features of one ordered pair (i, j) of trajectory `base / N`: packed bin code + radial
__device__ inline void edge_feature(const float4 *__restrict__ n4, const float4 *__restrict__ ca4, const float4 *__restrict__ cb4,
                                    size_t base, int i, int j, int R, float mask_dist, uint32_t &code, float &r2_out)
{
    const float4 cai4 = ca4[base + i], caj4 = ca4[base + j], cbi4 = cb4[base + i], cbj4 = cb4[base + j];
#ifdef DFM_N4_COHERENT      // experiment (r05): read N_i past L1 and L2 (system-coherent load)
    float4 ni4;
    {
        const float4 *ap = n4 + base + i;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(ni4) : "v"(ap) : "memory");
    }
#else
    const float4 ni4 = n4[base + i];
#endif
    const v3 Ni{ni4.x, ni4.y, ni4.z};
    const v3 Cai{cai4.x, cai4.y, cai4.z}, Caj{caj4.x, caj4.y, caj4.z}, Cbi{cbi4.x, cbi4.y, cbi4.z},
        Cbj{cbj4.x, cbj4.y, cbj4.z};
    const v3 dv = vsub(Cai, Caj);
    const float r2 = (dv.x * dv.x + dv.y * dv.y) + dv.z * dv.z;
    const float d = sqrtf(r2);
    int bd = 0;
#pragma unroll
    for (int q = 0; q < 39; ++q) bd += (d > (3.25f + 1.25f * (float)q)) ? 1 : 0;   // linspace(3.25, 50.75, 39)
    int bo = 0, bt = 0, bp = 0;
    if (d < mask_dist && i != j) {   // mask = dist < 22.0, fill_diagonal_(0)
        const float om = dihedral_deg(Cai, Cbi, Cbj, Caj);
        const float th = dihedral_deg(Ni, Cai, Cbi, Cbj);
        const float ph = planar_deg(Cai, Cbi, Cbj);
        bo = bin_angle(om);
        bt = bin_angle(th);
#pragma unroll
        for (int q = 0; q < 11; ++q) bp += (ph > 18.0f * (float)q) ? 1 : 0;     // linspace(0, 180, 11)
    }
    const bool same = (i < R) == (j < R);
    int off = i - j + 32;
    off = off < 0 ? 0 : (off > 64 ? 64 : off);
    const int rp = same ? off : 65;
    code = pack_code(bd, bo, bt, bp, rp);
#ifdef DFM_FEAT_PROBE      // experiment (r05): load N_i again and compute theta again - bit 30: the two loads differ; bit 31: same loads, different bin
    if (d < mask_dist && i != j) {
        float4 nb;
        const float4 *ap = n4 + base + i;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(nb) : "v"(ap) : "memory");
        const bool same = nb.x == ni4.x && nb.y == ni4.y && nb.z == ni4.z;
        const int bt2 = bin_angle(dihedral_deg(v3{nb.x, nb.y, nb.z}, Cai, Cbi, Cbj));
        if (!same) code |= 1u << 30;
        else if (bt2 != bt) code |= 1u << 31;
    }
#endif
    r2_out = r2;
}

// index of the intra-chain ordered pair (i, j) in the layer-0 message table: the receptor block [R][R], then the ligand block [L][L]
__device__ inline uint32_t l0_pair_index(int i, int j, int R, int L)
{
    return i < R ? (uint32_t)i * (uint32_t)R + (uint32_t)j : (uint32_t)R * (uint32_t)R + (uint32_t)(i - R) * (uint32_t)L + (uint32_t)(j - R);
}

// cls.code0 set (layer 0 behind the message table, kernels_edge.hip: k_l0_gather): the edge is a table HIT when both residues are of
// one chain and the code just computed from THIS pose equals the code the table entry was built with - the bins are discontinuous
// functions of fp32 geometry, and a rigidly moved chain can land on the other side of a boundary in the last bit; such an edge is
// evaluated like an inter-chain one, so the bins the engine uses are always those of the pose at hand.  Misses are appended to the
// row list (one atomic per wave; the position of a row has no influence on its value).
template <int CLS>      // CLS 1: with the table classification, 1024 threads per workgroup (one list reservation per workgroup)
__global__ __launch_bounds__(CLS ? 1024 : 256) void k_edge_feat(const float4 *__restrict__ n4, const float4 *__restrict__ ca4,
                                                   const float4 *__restrict__ cb4, const int32_t *__restrict__ edges,
                                                   long long total, int N, int R, int K, float mask_dist,
                                                   uint32_t *__restrict__ codes, float *__restrict__ radial, L0Classify cls,
                                                   uint32_t *__restrict__ eval_ctr)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // replayed step graph: the evaluation counter moves on here - after its reader of this evaluation's first half (k_knn_sample, an
    // earlier launch) and before k_heads, which reads counter - 1
    if (eval_ctr && e == 0) *eval_ctr += 1u;
    const bool live = e < total;
    if (!CLS && !live) return;
    int i = 0, j = 0;
    uint32_t code = 0u;
    float r2 = 0.f;
    if (live) {
        const long long node = e / K;
        const int b = (int)(node / N);
        i = (int)(node % N);
        j = edges[e];
        edge_feature(n4, ca4, cb4, (size_t)b * N, i, j, R, mask_dist, code, r2);
        codes[e] = code;
        radial[e] = r2;
    }
    if constexpr (CLS) {
        // positions in the row list: wave ballot -> one LDS add per wave -> ONE global add per workgroup.  (One global add per wave -
        // 144 k returning atomics on one address per evaluation at C3 - serialises in the L2: 1.2 ms, profiles/r04_l0_table.txt.)
        __shared__ uint32_t s_cnt, s_base;
        if (threadIdx.x == 0) s_cnt = 0u;
        __syncthreads();
        const bool same = (i < R) == (j < R);
        const uint32_t idx = (live && same) ? l0_pair_index(i, j, R, N - R) : 0u;
        // a HIT needs the pair's code AND its squared distance to be those the entry was built with: under the rigid motion of a chain
        // the distance moves by rounding only (|d r2| <= 2 d x ~1e-5 A), a different conformer or a perturbed backbone moves it by
        // >= 2 d x 1e-2 A - such an edge is a miss and goes through the edge model, so DFM_F_L0_TABLE is safe for ANY lig_pos
        // (the rigid-image precondition is now checked instead of assumed)
        const uint2 c0 = cls.code0[idx];
        const bool hit = live && same && c0.x == code && fabsf(__uint_as_float(c0.y) - r2) <= 1e-3f * fmaxf(sqrtf(r2), 1.0f);
        const bool is_miss = live && !hit;
        const unsigned long long miss = __ballot(is_miss);
        const int lane = threadIdx.x & 63;
        uint32_t wbase = 0u;
        if (miss && lane == 0) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(miss));
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(cls.counter, s_cnt) : 0u;
        __syncthreads();
        if (is_miss) {
            const uint32_t at = s_base + wbase + (uint32_t)__popcll(miss & ((1ull << lane) - 1ull));
            cls.rows[at] = make_uint4((uint32_t)i, (uint32_t)j, code, __float_as_uint(r2));
            cls.src[e] = 0x80000000u | at;
        } else if (hit) cls.src[e] = idx;
    }
}

hipError_t launch_edge_feat(const float4 *n4, const float4 *ca4, const float4 *cb4, const int32_t *edges, int B, int N,
                            int R, int K, float mask_dist, uint32_t *codes, float *radial, const L0Classify &cls, uint32_t *eval_ctr,
                            hipStream_t s)
{
    const long long total = (long long)B * N * K;
    if (cls.code0)
        hipLaunchKernelGGL(k_edge_feat<1>, dim3((unsigned)((total + 1023) / 1024)), dim3(1024), 0, s, n4, ca4, cb4, edges, total,
                           N, R, K, mask_dist, codes, radial, cls, eval_ctr);
    else
        hipLaunchKernelGGL(k_edge_feat<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), token_lds(), s, n4, ca4, cb4, edges, total,
                           N, R, K, mask_dist, codes, radial, cls, eval_ctr);
    return hipGetLastError();
}

// every intra-chain ordered pair of the complex (self pairs included: slot 0 of a node is the node itself) as a row list for the table
// build, features from the prepared pose of trajectory 0; code0 = the code each table entry is built with
__global__ __launch_bounds__(256) void k_l0_pairs(const float4 *__restrict__ n4, const float4 *__restrict__ ca4, const float4 *__restrict__ cb4,
                                                  int R, int L, float mask_dist, uint2 *__restrict__ code0, uint4 *__restrict__ rows)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x, RR = (uint32_t)R * (uint32_t)R, P = RR + (uint32_t)L * (uint32_t)L;
    if (q >= P) return;
    int i, j;
    if (q < RR) { i = (int)(q / (uint32_t)R); j = (int)(q - (uint32_t)i * (uint32_t)R); }
    else { const uint32_t w = q - RR; const int il = (int)(w / (uint32_t)L); i = R + il; j = R + (int)(w - (uint32_t)il * (uint32_t)L); }
    uint32_t code;
    float r2;
    edge_feature(n4, ca4, cb4, 0, i, j, R, mask_dist, code, r2);
    code0[q] = make_uint2(code, __float_as_uint(r2));
    rows[q] = make_uint4((uint32_t)i, (uint32_t)j, code, __float_as_uint(r2));
}

hipError_t launch_l0_pairs(const float4 *n4, const float4 *ca4, const float4 *cb4, int R, int L, float mask_dist, uint2 *code0,
                           uint4 *rows, hipStream_t s)
{
    const long long P = (long long)R * R + (long long)L * L;
    hipLaunchKernelGGL(k_l0_pairs, dim3((unsigned)((P + 255) / 256)), dim3(256), token_lds(), s, n4, ca4, cb4, R, L, mask_dist, code0, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// randomize_pose (inference_base.py:318-340): uniform random rotation about the ligand CA centroid and
// a N(0, 30^2) translation relative to the receptor centroid.  One workgroup per trajectory.
__device__ inline float normal_from(uint32_t a, uint32_t b)
{
    return sqrtf(-2.0f * logf(u01(a))) * cosf(6.283185307179586f * u01(b));
}

__global__ __launch_bounds__(256) void k_init_pose(const float *__restrict__ rec_pos, const float *__restrict__ lig0, int R,
                                                   int L, int all_atoms, const float *__restrict__ R0_inj,
                                                   const float *__restrict__ tr_inj, uint32_t seed_lo, uint32_t seed_hi,
                                                   float *__restrict__ lig_cur, float *__restrict__ tr_update,
                                                   float *__restrict__ rot_update)
{
    __shared__ double scratch[8];
    __shared__ float sh[24];   // c2[3], tr[3], R0[9]
    const int b = blockIdx.x;
    double a0 = 0, a1 = 0, a2 = 0, l0 = 0, l1 = 0, l2 = 0;
    if (all_atoms) {   // second family: torch.mean(x, dim=(0, 1)) over all backbone atoms (src/inference.py:224-225)
        for (int q = threadIdx.x; q < R * 3; q += blockDim.x) { a0 += rec_pos[q * 3]; a1 += rec_pos[q * 3 + 1]; a2 += rec_pos[q * 3 + 2]; }
        for (int q = threadIdx.x; q < L * 3; q += blockDim.x) { l0 += lig0[q * 3]; l1 += lig0[q * 3 + 1]; l2 += lig0[q * 3 + 2]; }
    } else {
        for (int q = threadIdx.x; q < R; q += blockDim.x) { a0 += rec_pos[q * 9 + 3]; a1 += rec_pos[q * 9 + 4]; a2 += rec_pos[q * 9 + 5]; }
        for (int q = threadIdx.x; q < L; q += blockDim.x) { l0 += lig0[q * 9 + 3]; l1 += lig0[q * 9 + 4]; l2 += lig0[q * 9 + 5]; }
    }
    const int nr = all_atoms ? R * 3 : R, nl = all_atoms ? L * 3 : L;
    a0 = block_sum_d(a0, scratch); a1 = block_sum_d(a1, scratch); a2 = block_sum_d(a2, scratch);
    l0 = block_sum_d(l0, scratch); l1 = block_sum_d(l1, scratch); l2 = block_sum_d(l2, scratch);
    if (threadIdx.x == 0) {
        const float c1[3] = {(float)(a0 / nr), (float)(a1 / nr), (float)(a2 / nr)};
        const float c2[3] = {(float)(l0 / nl), (float)(l1 / nl), (float)(l2 / nl)};
        float R0[9], draw[3];
        if (R0_inj) {
            for (int k = 0; k < 9; ++k) R0[k] = R0_inj[b * 9 + k];
        } else {
            // scipy Rotation.random(): normalised Gaussian quaternion (x,y,z,w), matrix in float64
            const u32x4 r1 = philox4x32((uint32_t)b, 0u, 0u, RNG_INIT, seed_lo, seed_hi);
            const u32x4 r2 = philox4x32((uint32_t)b, 1u, 0u, RNG_INIT, seed_lo, seed_hi);
            double q[4] = {(double)normal_from(r1.x, r1.y), (double)normal_from(r1.z, r1.w),
                           (double)normal_from(r2.x, r2.y), (double)normal_from(r2.z, r2.w)};
            const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
            R0[0] = (float)(1 - 2 * (y * y + z * z)); R0[1] = (float)(2 * (x * y - z * w)); R0[2] = (float)(2 * (x * z + y * w));
            R0[3] = (float)(2 * (x * y + z * w)); R0[4] = (float)(1 - 2 * (x * x + z * z)); R0[5] = (float)(2 * (y * z - x * w));
            R0[6] = (float)(2 * (x * z - y * w)); R0[7] = (float)(2 * (y * z + x * w)); R0[8] = (float)(1 - 2 * (x * x + y * y));
        }
        if (tr_inj) {
            for (int k = 0; k < 3; ++k) draw[k] = tr_inj[b * 3 + k];
        } else {
            const u32x4 r3 = philox4x32((uint32_t)b, 2u, 0u, RNG_INIT, seed_lo, seed_hi);
            const u32x4 r4 = philox4x32((uint32_t)b, 3u, 0u, RNG_INIT, seed_lo, seed_hi);
            draw[0] = 30.0f * normal_from(r3.x, r3.y); draw[1] = 30.0f * normal_from(r3.z, r3.w);
            draw[2] = 30.0f * normal_from(r4.x, r4.y);
        }
        float aa[3];
        mat_to_aa(R0, aa);
        for (int k = 0; k < 3; ++k) {
            const float tr = (draw[k] - c2[k]) + c1[k];
            sh[k] = c2[k]; sh[3 + k] = tr;
            tr_update[b * 3 + k] = tr;
            rot_update[b * 3 + k] = aa[k];
        }
        for (int k = 0; k < 9; ++k) sh[6 + k] = R0[k];
    }
    __syncthreads();
    float *out = lig_cur + (size_t)b * L * 9;
    for (int a = threadIdx.x; a < L * 3; a += blockDim.x) {
        const float v0 = lig0[a * 3] - sh[0], v1 = lig0[a * 3 + 1] - sh[1], v2 = lig0[a * 3 + 2] - sh[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s = 0;
            s += v0 * sh[6 + r * 3]; s += v1 * sh[6 + r * 3 + 1]; s += v2 * sh[6 + r * 3 + 2];
            out[a * 3 + r] = (s + sh[r]) + sh[3 + r];
        }
    }
}

hipError_t launch_init_pose(const float *rec_pos, const float *lig0, int B, int R, int L, int all_atoms, const float *R0,
                            const float *tr_draw, uint64_t seed, float *lig_cur, float *tr_update, float *rot_update,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_init_pose, dim3(B), dim3(256), 0, s, rec_pos, lig0, R, L, all_atoms, R0, tr_draw, (uint32_t)seed,
                       (uint32_t)(seed >> 32), lig_cur, tr_update, rot_update);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// get_clash_force (inference_base.py:366-384), closed form of the reference's autograd:
// E = -5 * sum_{d<4} (4-d)^1.5 / (0.75 d) over all backbone-atom pairs; the ligand is shifted rigidly
// by the mean over its 3L atoms of dE/dx.
// One workgroup of 1024 threads per trajectory, thread = ligand atom(s), the receptor's atoms in LDS; a pair goes through the float64 closed
// form only after a float32 distance test with a margin (d^2 < 16.5: pairs within 4 A are rare) - r01-r04 ran every one of the 810 k pairs of a
// 300+300 complex through float64 sqrt / division in 256 threads: 0.53 ms per step whatever the batch (+69 % on a B = 8 call,
// tools/option_costs.py).  Same expressions for the pairs that count; the float64 partial sums group differently (1e-16).
__global__ __launch_bounds__(1024) void k_clash_force(const float *__restrict__ rec_pos, int R, int L,
                                                      float *__restrict__ lig_cur, float *__restrict__ tr_update)
{
    constexpr int CH = 3072;      // receptor atoms per LDS chunk (36 KB)
    __shared__ float s_rec[CH * 3];
    __shared__ double scratch[16];
    __shared__ float shift[3];
    const int b = blockIdx.x;
    float *lig = lig_cur + (size_t)b * L * 9;
    const int nj = (L * 3 + (int)blockDim.x - 1) / (int)blockDim.x;      // ligand atoms per thread (1 up to 341 residues)
    double g0 = 0, g1 = 0, g2 = 0;
    for (int jj = 0; jj < nj; ++jj) {
        const int j = threadIdx.x + jj * blockDim.x;
        const bool jv = j < L * 3;
        const float fx = jv ? lig[j * 3] : 0.f, fy = jv ? lig[j * 3 + 1] : 0.f, fz = jv ? lig[j * 3 + 2] : 0.f;
        const double lx = fx, ly = fy, lz = fz;
        for (int c0 = 0; c0 < R * 3; c0 += CH) {      // (per thread the receptor atoms are still visited in ascending order)
            const int cn = R * 3 - c0 < CH ? R * 3 - c0 : CH;
            __syncthreads();
            for (int q = threadIdx.x; q < cn * 3; q += blockDim.x) s_rec[q] = rec_pos[(size_t)c0 * 3 + q];
            __syncthreads();
            if (!jv) continue;
            for (int i = 0; i < cn; ++i) {
                const float ex = s_rec[i * 3] - fx, ey = s_rec[i * 3 + 1] - fy, ez = s_rec[i * 3 + 2] - fz;
                if (ex * ex + ey * ey + ez * ez < 16.5f) {
                    const double dx = (double)s_rec[i * 3] - lx, dy = (double)s_rec[i * 3 + 1] - ly, dz = (double)s_rec[i * 3 + 2] - lz;
                    const double d = sqrt(dx * dx + dy * dy + dz * dz);
                    if (d < 4.0 && d > 0.0) {
                        const double u = 4.0 - d;
                        const double fp = (-1.5 * sqrt(u) * d - u * sqrt(u)) / (0.75 * d * d);
                        const double dE = -5.0 * fp;
                        g0 += dE * (-dx / d); g1 += dE * (-dy / d); g2 += dE * (-dz / d);
                    }
                }
            }
        }
    }
    g0 = block_sum_d(g0, scratch); g1 = block_sum_d(g1, scratch); g2 = block_sum_d(g2, scratch);
    if (threadIdx.x == 0) {
        shift[0] = (float)(g0 / (L * 3)); shift[1] = (float)(g1 / (L * 3)); shift[2] = (float)(g2 / (L * 3));
        for (int k = 0; k < 3; ++k) tr_update[b * 3 + k] += shift[k];
    }
    __syncthreads();
    for (int a = threadIdx.x; a < L * 9; a += blockDim.x) lig[a] += shift[a % 3];
}

hipError_t launch_clash_force(const float *rec_pos, int B, int R, int L, float *lig_cur, float *tr_update, hipStream_t s)
{
    hipLaunchKernelGGL(k_clash_force, dim3(B), dim3(1024), 0, s, rec_pos, R, L, lig_cur, tr_update);
    return hipGetLastError();
}

}  // namespace dfm
