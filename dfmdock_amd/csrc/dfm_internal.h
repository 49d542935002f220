// dfm_internal.h - shared declarations of the gfx950 engine (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include <atomic>

#include "../../include/dfmdock_amd.h"

namespace dfm {

constexpr int H = 256;        // node_dim (kernels are specialised for it)
constexpr int HE = 128;       // edge_dim
constexpr int HI = 128;       // inner_dim
constexpr int NTAB = 166;     // 100 spatial + 66 positional one-hot slots
constexpr int KPAD = 64;      // edges per node padded to two 32-row MFMA tiles
constexpr int MAX_NODES = 4096;
constexpr float SILU_S = -1.44269504088896340736f;   // -log2(e): scale carried by every SiLU input of the 16-bit MFMA edge kernels

// packed per-edge feature code: T-table row offsets are implied by the field
// (dist 0..39 | omega 40.. | theta 64.. | phi 88.. | relpos 100..)
__host__ __device__ inline uint32_t pack_code(int d, int om, int th, int ph, int rp)
{
    return (uint32_t)d | ((uint32_t)om << 6) | ((uint32_t)th << 11) | ((uint32_t)ph << 16) | ((uint32_t)rp << 20);
}

// channel handled by (k-step kk, lane-half h, element e) of the bf16 MFMA operands (natural order)
__host__ __device__ inline int frag_channel(int kk, int h, int e) { return kk * 16 + h * 8 + e; }

// merged edge-feature tables of the 16-bit MFMA kernel (stored as fp16): row gathers per edge and 32-channel chunk instead of 5.
// DFM_TAB_MERGE 1 (two gathers; 4.9 MB per layer, but the angle bins are zeroed beyond 22 A and most sequence offsets saturate,
// so the rows actually touched are few):
//   [0, 6912)     (omega*24 + theta)*12 + phi   = T[40+omega] + T[64+theta] + T[88+phi]
//   [6912, 9552)  6912 + relpos*40 + d          = T[100+relpos] + T[d]
// DFM_TAB_MERGE 0 (three gathers, 574 KiB per layer):
//   [0, 576)      omega*24 + theta              = T[40+omega] + T[64+theta]
//   [576, 1056)   576 + phi*40 + d              = T[88+phi]   + T[d]
//   [1056, 1122)  1056 + relpos                 = T[100+relpos]
#ifndef DFM_TAB_MERGE
#define DFM_TAB_MERGE 1
#endif
constexpr int NTAB2 = DFM_TAB_MERGE ? 6912 + 2640 : 576 + 480 + 66;

struct LayerDev {
    float *Wab;       // [512][256]   rows 0..255 = edge_mlp.0.weight[:, 0:256] (h_i), 256..511 = [:, 256:512] (h_j)
    float *bias_ab;   // [512]        [edge_mlp.0.bias ; 0]
    float *w_r;       // [256]        edge_mlp.0.weight[:, 512] (radial column)
    float *T;         // [166][256]   T = ([S|P]^T We^T): per-layer edge-feature lookup table, fp32
    uint16_t *T2b;    // [NTAB2][256]  merged tables (see NTAB2), fp16
    float *W2t;       // [256 in][256 out] edge_mlp.2.weight transposed (fp32 kernel)
    uint16_t *W2f;    // [16][8][64][8] bf16 MFMA B-fragments of edge_mlp.2.weight
    uint16_t *W2f16;  // same, fp16
    float *b2;        // [256]
    float *att_w;     // [256]
    float att_b;
    float *W3;        // [256][512]   node_mlp.0.weight
    float *b3;        // [256]
    float *gn_w, *gn_b, *gn_ms;   // GraphNorm
    float *W4;        // [256][256]   node_mlp.3.weight
    float *b4;
    uint16_t *Wab_hi, *Wab_lo, *W3_hi, *W3_lo, *W4_hi, *W4_lo;   // bf16 hi/lo splits for launch_gemm_split
    float *Wc1t;      // [256 in][256 out] coord_mlp.0.weight transposed (last layer)
    uint16_t *Wc1f;   // bf16 fragments of coord_mlp.0.weight
    uint16_t *Wc1f16; // fp16 fragments
    float *bc1;       // [256]
    float *wc2;       // [256]
    // operands of the 16-bit MFMA edge kernels, pre-multiplied by SILU_S = -log2(e) (kernels_edge.hip): T2b, Wab_hi/lo and
    // the arrays below.  The fp32 engine never reads them.
    float *w_r_s;     // [256]  SILU_S * w_r
    float *bias_ab_s; // [512]  SILU_S * bias_ab
    // positional_embed_dim 67: the same two biases with the homomer ("sym") channel's contribution We . P[:, 66] added to
    // the A half - a constant of every edge of a homomeric complex (dfm_complex_set_homomer); nullptr for 66 channels
    float *bias_ab_h, *bias_ab_h_s;
    uint32_t *b2p, *b2p16;    // [8][64] SILU_S * b2 as packed (hi, lo) bf16 / fp16 pairs per (n-tile, lane), 0 for lanes >= 32
    uint32_t *bc1p, *bc1p16;  // [8][64] SILU_S * bc1, same packing
    float *wc2_s;     // [256]  wc2 / SILU_S
};

struct HeadsDev {
    float *en_wa;     // [256][256] to_energy.0.weight[:, :256]
    float *en_wb;     // [256][256] to_energy.0.weight[:, 256:]
    float *en_ln_w, *en_ln_b, *en_w3;
    float *t_W;       // [64]
    float *t_lin;     // [128][128]
    float *trs0, *trs_ln_w, *trs_ln_b, *trs4;   // [128][129], [128], [128], [128]
    float *rots0, *rots_ln_w, *rots_ln_b, *rots4;
};

// one receptor x ligand pair head of model family 1 (egnn_net.py:329-358): Linear(513->256) split into the stacked node
// halves [W[:, :256]; W[:, 256:512]] and the distance column, LayerNorm affine, Linear(256->1)
struct PairHeadDev {
    float *wab;                 // [512][256] fp32 (fp32 engine)
    uint16_t *wab_hi, *wab_lo;  // split-bf16 tiles of the same (16-bit engines)
    float *w_d, *ln_w, *ln_b, *w3;
};

constexpr int MAX_DEVICES = 64;
// CU count of the current device, queried once per device (launch helpers ask on every launch)
inline int device_cus()
{
    static std::atomic<int> cache[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev].store(c, std::memory_order_relaxed);
    return c;
}
// The opt-in to more than 64 KiB of dynamic LDS is a per-device function attribute: remember it per device (atomic flags:
// two host threads may drive two GPUs)
inline hipError_t ensure_lds_attr(const void *fn, int bytes, std::atomic<bool> (&done)[MAX_DEVICES])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[dev].store(true, std::memory_order_release);
    return e;
}

// ---- kernel launchers (each returns the hipError of the launch) ---------------------------------
struct GemmArgs {
    const float *A0;      // [M][lda0]
    const float *A1;      // second half of a concatenated A (PRO_CONCAT) or nullptr
    int lda;              // leading dimension of A0 (and A1)
    int K;                // total K (for PRO_CONCAT: 2*256)
    const float *W;       // [Nout][ldw]
    int ldw;
    const float *bias;    // [Nout] or nullptr
    int M, Nout;
    // prologue: GraphNorm + SiLU applied to A on load (rows grouped by trajectory: b = row / rows_per_graph)
    int pro;              // 0 plain, 1 concat(A0,A1) each K/2 wide, 2 graphnorm+silu, 3 silu (k_gemm_f32 only)
    const float *gn_shift;   // [B][256]  mean*mean_scale
    const float *gn_den;     // [B][256]  sqrt(var+eps)
    const float *gn_w, *gn_b;
    int rows_per_graph;
    // epilogue
    int epi;              // 0 store, 1 residual add (C = R + acc + bias), 2 split into C (cols<256) and C2 (cols>=256, + bf16 copy)
    const float *R;       // residual [M][ldc]
    float *C;             // [M][ldc]
    int ldc;
    float *C2;            // [M][256]  (epi 2)
    uint16_t *C2b;        // [M][256]  fp16 copy of C2 (epi 2), may be nullptr
    uint16_t *Cb;         // [M][256]  epi 2: when set, the first 256 columns go here as fp16 INSTEAD of C (k_gemm_split only)
    // optional (k_gemm_split, 64-row tiles, Nout = 256): row tiles aligned to the trajectories (rows_per_graph rows each) and
    // per-tile column statistics of the output for GraphNorm, [M / rows_per_graph][tiles per trajectory][256][2] =
    // (mean, sum of squared deviations) of the tile's rows; merged in the prologue of the consuming GEMM (gn_part)
    float *stat_part;
    // optional (k_gemm_split, pro 2): the GraphNorm statistics are finished INSIDE the prologue - every workgroup merges the per-tile
    // (mean, M2) pairs of its trajectory (gn_part, written by the previous launch's stat_part) exactly as k_gn_finish does, so small
    // launches do not pay a separate 5 us kernel per layer; gn_ms = mean_scale [256]; gn_shift / gn_den are then unused
    const float *gn_part, *gn_ms;
    // optional (k_gemm_split, Nout = 256): every workgroup zeroes its rows x columns block of this [M][256] buffer, which no launch
    // reads any more (agg after node_mlp.0: the next layer's tile-task message launch adds into it atomically and would otherwise
    // need a memset launch per layer)
    float *zbuf;
    // optional row periods (0 = none): row r of A0 / of the residual R is read at r % period - layer 0's node embedding h0 [N][256] is
    // the same for every trajectory and is read in place instead of from a per-evaluation [B][N][256] copy
    int a0_period, r_period;
};
hipError_t launch_gemm_f32(const GemmArgs &a, hipStream_t s);
// split-bf16 (hi/lo) variant, ~1e-5 relative error; Whi/Wlo = pre-split weights [Nout][ldw] bf16
// W16 non-null: the two-term fp16 form (weights as ONE fp16 tile in the same [K/32][4][Nout][8] order); else three bf16 terms
hipError_t launch_gemm_split(const GemmArgs &a, const uint16_t *Whi, const uint16_t *Wlo, hipStream_t s);
int gemm_rows_per_tile();   // 64: the row tile of k_gemm_split (tiles of the fused GraphNorm statistics)

// Dynamic LDS bytes for kernels that need none.  r05 (tools/concurrency_probe*.py, profiles/r05_concurrency.txt): a wave of a kernel
// WITHOUT any LDS allocation can be placed on a CU whose LDS is entirely held by a 160 KiB workgroup of the message kernel of ANOTHER
// complex handle (another stream, another hardware queue) - and then computed wrong values: k_edge_feat<0> returned theta bins of a
// garbage N_i for scattered nodes in 11 of 12 concurrent calls, with GPU_MAX_HW_QUEUES <= 2 (both streams on one hardware queue) or
// with as little as 64 bytes of LDS on the victim in 0 of 12.  r06 (profiles/r06_concurrency.txt): the wrong values come from ONE
// instruction form - a packed fp32 VALU instruction with op_sel = [0,1], which hipcc's SLP vectoriser had made of the dihedral's cross
// product - that miscomputes in such a wave (reproduced stand-alone, tools/pkmul_probe.py).  The library no longer contains the form
// (-fno-slp-vectorize, audited by tests/test_abi_cpu.py); the token allocation stays as the second, independent fence: every kernel of
// this library holds at least 64 bytes of LDS, which keeps it off CUs whose LDS is full.  DFM_TOKEN_LDS=0 restores the old launches
// (for reproducing the effect together with tools/asm_variant.py).
inline unsigned token_lds()
{
    static const unsigned v = [] { const char *e = getenv("DFM_TOKEN_LDS"); return e ? (unsigned)atoi(e) : 64u; }();
    return v;
}

hipError_t launch_prep_pose(const float *rec_pos, const float *lig_cur, int B, int R, int L, int all_atoms, float4 *n4,
                            float4 *ca4, float4 *cb4, hipStream_t s);
// ctl (or nullptr): device words {evaluation index, seed lo, seed hi} that override seed / stream_id (replayed step graph)
hipError_t launch_knn_sample(const float4 *ca4, int B, int N, int knn, int nsamp, uint64_t seed, uint32_t stream_id,
                             int32_t *edges, const uint32_t *ctl, hipStream_t s);
// layer 0 behind the per-complex message table (kernels_edge.hip: k_l0_gather): k_edge_feat's classification of every edge into
// table hits (src = pair index) and row-list entries (src = 0x80000000 | position).  code0 == nullptr: no classification.
struct L0Classify {
    const uint2 *code0;      // [pairs] (feature code, bits of |x_i - x_j|^2) each table entry was built with
    uint32_t *src;           // [B][N][K]
    uint4 *rows;             // [capacity] (i, j, code, radial bits) of the misses
    uint32_t *counter;       // rows appended so far (zero at the start of an evaluation: k_l0_gather resets it)
};
hipError_t launch_edge_feat(const float4 *n4, const float4 *ca4, const float4 *cb4, const int32_t *edges, int B,
                            int N, int R, int K, float mask_dist, uint32_t *codes, float *radial, const L0Classify &cls,
                            uint32_t *eval_ctr /* or nullptr: incremented once per launch (replayed step graph) */, hipStream_t s);
hipError_t launch_l0_pairs(const float4 *n4, const float4 *ca4, const float4 *cb4, int R, int L, float mask_dist, uint2 *code0,
                           uint4 *rows, hipStream_t s);

constexpr int TASK_CTR_WGS = 1024;      // workgroups the message kernel's task counters are sized for (its persistent grid is one workgroup per CU)
struct EdgeArgs {
    const float *A;        // [Ab][N][256]  Wa h_i + b1   (Ab = 1 when a_bstride == 0)
    const float *Bm;       // [Ab][N][256]  Wb h_j        fp32
    const uint16_t *Bmb;   // same, fp16 (gathered operand of the bf16-MFMA kernel)
    const uint16_t *Ah;    // A as fp16 (bf16-operand message kernel: one load per chunk instead of two), or nullptr
    int64_t ab_bstride;    // elements between trajectories (0 for layer 0: pose independent)
    const int32_t *edges;  // [B][N][K]
    const uint32_t *codes; // [B][N][K]
    const float *radial;   // [B][N][K]
    const float4 *ca4;     // [B][N]
    int B, N, R, K;
    const LayerDev *lw;    // host copy of the layer's device pointers
    float *agg;            // [B][N][256]
    int last;              // last layer: also the coordinate update for ligand nodes
    float *fout;           // [B][L][3]   (last)
    uint16_t *mbuf;        // [B][L][64][256] 16-bit gated messages (MFMA path, last)
    int f16;               // MFMA operand type: 0 bf16, 1 fp16
    int lig_only;          // last layer, node outputs not wanted: messages of the LIGAND nodes only (the coordinate update reads nothing else)
    int agg_is_zero;       // 16-bit kernel with tile tasks: agg is known to be zero (zeroed by the previous layer's node_mlp.3 GEMM, GemmArgs::zbuf)
    unsigned long long *stamp;   // diagnostic builds (DFM_EDGE_STAMP): [8 waves][4 phases] cycle sums of workgroup 0, or nullptr
    uint32_t *range;             // fp32 kernel, selfcheck only: [2] running maxima of |pre-activation| of edge_mlp.0 / edge_mlp.2 (float bits)
    uint32_t *task_ctr;          // 16-bit message kernel: [2 * TASK_CTR_WGS] per-workgroup task + exit counters of this handle (zero between launches: the kernel resets them), or nullptr
};
hipError_t launch_edge_f32(const EdgeArgs &a, hipStream_t s);
hipError_t launch_edge_bf16(const EdgeArgs &a, hipStream_t s);
bool edge_msg_tile_tasks(int B, int N, int K);   // does launch_edge_bf16 run tile tasks (atomic adds into a zero agg) for this size?
hipError_t launch_coord_bf16(const EdgeArgs &a, hipStream_t s);
// row-list form of the message kernel and the slot-order gather-sum of layer 0 behind the message table (kernels_edge.hip)
hipError_t launch_edge_rows(const EdgeArgs &a, const uint4 *rows, const uint32_t *n_rows_dev, uint32_t n_rows_cap, uint16_t *out, hipStream_t s);
// fp32 engine: the same over k_edge_f32m<1> (fp32 rows, 1 KiB each) and the gather-sum of fp32 rows
hipError_t launch_edge_rows32(const EdgeArgs &a, const uint4 *rows, const uint32_t *n_rows_dev, uint32_t n_rows_cap, float *out, hipStream_t s);
hipError_t launch_l0_gather32(const float *table, const float *X, const uint32_t *src, float *agg, int B, int N, int K,
                              uint32_t *counter, unsigned long long *miss_total, hipStream_t s);
hipError_t launch_l0_gather(const uint16_t *table, const uint16_t *X, const uint32_t *src, float *agg, int B, int N, int K,
                            uint32_t *counter, unsigned long long *miss_total, hipStream_t s);

// fold_w / fold_b non-null: write the folded affine (den := w/den, shift := b - w*shift/den) for launch_gemm_split
// GraphNorm statistics from the per-tile column statistics the node_mlp.0 GEMM left in stat_part (no second pass over u)
hipError_t launch_gn_stats(const float *u, int B, int N, const float *mean_scale, float *shift, float *den,
                           const float *fold_w, const float *fold_b, hipStream_t s);

struct PairArgs {
    const float *P, *Q;      // [B][N][256] projections of every node (receptor rows of P, ligand rows of Q are used)
    const float4 *ca4;
    int B, R, L;
    const float *w_d, *ln_w, *ln_b, *w3;
    int mode;                // 0 force (+ clashes), 1 energy, 2 confidence
    int exact;               // fp32 engine: three-pass LayerNorm, expf
    float cut_off;
    float *fpart;            // [B][ceil(R/64)][L][3]
    float *spart;            // [B][ceil(R/64)*4][2]
    int32_t *clash_part;     // [B][ceil(R/64)*4]
    float *S; int Rp;        // 16-bit engines (k_pair_head_m): the pair scalars s(r, l) as [B][L][Rp], Rp = 32 ceil(R / 32)
};
hipError_t launch_pair_head(const PairArgs &a, hipStream_t s);
// 16-bit engines: the rank-4 part of the pre-activation on the fp32 matrix pipe -> S; then the reductions of one head (mode of a) from S
hipError_t launch_pair_head_m(const PairArgs &a, hipStream_t s);
hipError_t launch_pair_finish_s(const PairArgs &a, int n_part, float inv_pool, float *fvec, float *conf, hipStream_t s);
// dist_logits [B][R][L][64] = Linear(256 -> 64)(SiLU(LayerNorm(P_r + Q_l + w_d D)))  (egnn_net.py:347-352,:447); exact fp32
hipError_t launch_pair_dist(const float *P, const float *Q, const float4 *ca4, int B, int R, int L, const float *w_d, const float *ln_w,
                            const float *ln_b, const float *w3t /*[256][64]*/, float *out, hipStream_t s);
hipError_t launch_pair_finish(const float *fpart, int B, int R, int L, float inv_pool, float *fvec, const float *cpart,
                              float *conf, hipStream_t s);

// per-step scalars of the Euler-Maruyama update as k_heads reads them from device memory when the step loop is a replayed graph
struct StepParams { float g2_r, g_r, hg2_r, g2_t, g_t, hg2_t, dt, sqrt_dt, rot_noise, tr_noise; uint32_t step, pad; };

struct HeadArgs {
    const float *fvec;       // [B][L][3]
    const float4 *ca4;       // [B][N]
    int B, R, L;
    const float *hid_base;   // time-dependent half of the scale MLPs' first Linear (launch_time_embed): [.][2][128]
    int64_t hid_bstride;     // elements between trajectories in hid_base (0: the whole batch shares one time)
    const HeadsDev *hw;
    float *scores;           // [B][8] tr(3) rot(3) energy clashes
    // energy (optional)
    int want_energy;
    const float *en_part;    // [B][n_part][2]
    const int32_t *clash_part;  // [B][n_part]
    int n_part;              // partial sums per trajectory: R (family 0) or 4*ceil(R/64) (family 1)
    int en_mode;             // energy = 0: sum/(count + 1e-6) (score_net_mlsb.py:390); 1: sum/max(count, 1); 2: sum (egnn_net.py:438-441)
    float pool_div;          // tr / rot pooling divisor: L (mean) or 1 (`agg: sum`)
    // Euler-Maruyama update (optional)
    int do_update;
    float g2_r, g_r, hg2_r, g2_t, g_t, hg2_t;   // float32-rounded diffusion coefficients of this step
    float dt, sqrt_dt, rot_noise, tr_noise;
    int ode;
    const float *z_rot;      // [B][3] injected or nullptr (Philox)
    const float *z_tr;
    int64_t z_bstride;       // stride between trajectories in z arrays
    uint64_t seed;
    uint32_t step;
    int all_atoms;           // rotate about the all-backbone-atom centroid (second family, src/inference.py:244-254)
    float *lig_cur;          // [B][L][9] in/out
    float *tr_update;        // [B][3]
    float *rot_update;       // [B][3]
    float *trace_pose;       // [B][steps][L][9] or nullptr (already offset to this step)
    int64_t trace_bstride;
    float *trace_scores;     // [B][steps+1][8] or nullptr (already offset)
    int64_t trace_s_bstride;
    // replayed step graph: ctl = device words {evaluations started, seed lo, seed hi}; the step's scalars are step_params[ctl[0] - 1],
    // its time embedding hid_base + (ctl[0] - 1) * 256 (the by-value fields above are then ignored)
    const StepParams *step_params;
    const uint32_t *ctl;
    // prep_next: k_heads also prepares the pose it has just produced for the next evaluation (pos / ca4 / cb4 of its trajectory,
    // exactly what launch_prep_pose would write as that evaluation's first launch)
    int prep_next;
    const float *rec_pos;
    float4 *prep_pos;      // [B][N] centred backbone N
    float4 *prep_ca4, *prep_cb4;
};
hipError_t launch_heads(const HeadArgs &a, hipStream_t s);
// base[n][2][128] for the n times t_dev[n] (kernels_heads.hip: k_time_embed)
hipError_t launch_time_embed(const float *t_dev, int n, const HeadsDev *hw, float *base, hipStream_t s);

hipError_t launch_energy_pairs(const float *enA, const float *enB, const float4 *ca4, int B, int R, int L,
                               float cut_off, const HeadsDev *hw, int want_energy, float *en_part,
                               int32_t *clash_part, hipStream_t s);

// all_atoms: centroids over all backbone atoms (second family, src/inference.py:220-254) instead of the CA atoms
hipError_t launch_init_pose(const float *rec_pos, const float *lig0, int B, int R, int L, int all_atoms, const float *R0,
                            const float *tr_draw, uint64_t seed, float *lig_cur, float *tr_update,
                            float *rot_update, hipStream_t s);
hipError_t launch_clash_force(const float *rec_pos, int B, int R, int L, float *lig_cur, float *tr_update,
                              hipStream_t s);

}  // namespace dfm
