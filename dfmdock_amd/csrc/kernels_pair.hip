// kernels_pair.hip - receptor x ligand pair heads of the second model family (EGNN_Net).
// Reference: src/models/egnn_net.py:413-416 (vec, unit_vec, D), :430-441 (energy), :444 (confidence),
// :455-470 (force, pooling), :40-41 get_clashes; head shape :329-358:
//     s(r,l) = Linear(256->1, no bias)( SiLU( LayerNorm( Linear(513->256, no bias)( cat[h_r, h_l, D(r,l)] ))))
// The first Linear splits exactly into per-node halves and a distance column:
//     z_c = P[r][c] + Q[l][c] + w_d[c] * D          P = W[:, :256] h_r,  Q = W[:, 256:512] h_l,  w_d = W[:, 512]
// so the per-pair work is elementwise over the 256 channels (two statistics + SiLU + dot): VALU-bound, no MFMA.
//
// Layout: a workgroup owns one trajectory and a tile of 64 receptor residues whose P rows sit TRANSPOSED in LDS
// (Pt[c][r], 65-float rows: lane r reads without bank conflicts); lanes = receptor residues, the four waves stride
// over the ligand residues, and everything that depends on (l, c) only - Q, w_d, LayerNorm affine, w3 - is
// wave-uniform and comes through scalar loads.  No cross-lane traffic inside the channel loops; one wave
// reduction per (l, tile) for the force, fixed-order partials for the scalars (no atomics).
#include "dfm_device.h"
#include "dfm_internal.h"

namespace dfm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PT_LD = 65;
constexpr int PAIR_LDS_BYTES = H * PT_LD * 4;   // 66560

struct PairKArgs {
    const float *P, *Q;        // [B][N][H]; receptor rows of P and ligand rows of Q are read
    const float4 *ca4;         // [B][N] centred CA
    int R, L;
    const float *w_d, *ln_w, *ln_b, *w3;
    int mode;                  // 0 force (+ clash count), 1 energy (masked D < cut_off), 2 confidence
    float cut_off;
    float *fpart;              // mode 0: [B][RT][L][3]
    float *spart;              // mode 1: [B][RT*4][2] (sum, count)   mode 2: [B][RT*4][2] (sum, -)
    int32_t *clash_part;       // mode 0: [B][RT*4]
};

template <int EXACT>   // 1: three-pass LayerNorm, expf/division (fp32 engine); 0: sum / sum-of-squares, fast SiLU
__global__ __launch_bounds__(256) void k_pair_head(PairKArgs p)
{
    extern __shared__ float Pt[];
    const int rt = blockIdx.x, b = blockIdx.y, RT = gridDim.x, N = p.R + p.L;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = rt * 64 + lane;
    const bool valid = r < p.R;

    // stage the tile: coalesced float4 rows from global, transposed scalar writes to LDS.  A wave iteration covers one row (64 lanes
    // x 4 channels), so the row's channel moments sum P, sum P^2, sum P w_d - the LayerNorm statistics below are assembled from
    // moments and ONE dot product per pair instead of a pass over z - cost three wave reductions per row
    __shared__ float mom[3][64];
    const float4 wd4 = *reinterpret_cast<const float4 *>(p.w_d + lane * 4);
    for (int idx = threadIdx.x; idx < 64 * (H / 4); idx += 256) {
        const int row = idx >> 6, c4 = idx & 63, gr = rt * 64 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < p.R) v = *reinterpret_cast<const float4 *>(p.P + ((size_t)b * N + gr) * H + c4 * 4);
        Pt[(c4 * 4 + 0) * PT_LD + row] = v.x; Pt[(c4 * 4 + 1) * PT_LD + row] = v.y;
        Pt[(c4 * 4 + 2) * PT_LD + row] = v.z; Pt[(c4 * 4 + 3) * PT_LD + row] = v.w;
        if (!EXACT) {
            const float s1 = wave_sum((v.x + v.y) + (v.z + v.w));
            const float s2 = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
            const float sw = wave_sum((v.x * wd4.x + v.y * wd4.y) + (v.z * wd4.z + v.w * wd4.w));
            if (lane == 0) { mom[0][row] = s1; mom[1][row] = s2; mom[2][row] = sw; }
        }
    }
    __syncthreads();
    const float sum_w = wave_sum((wd4.x + wd4.y) + (wd4.z + wd4.w));
    const float sum_w2 = wave_sum((wd4.x * wd4.x + wd4.y * wd4.y) + (wd4.z * wd4.z + wd4.w * wd4.w));
    const float mP = mom[0][lane], mP2 = mom[1][lane], mPw = mom[2][lane];

    const float4 xr = p.ca4[(size_t)b * N + (valid ? r : 0)];
    const float *Pl = Pt + lane;
    double s_acc = 0, c_acc = 0;
    int clash = 0;
    for (int l = wave; l < p.L; l += 4) {
        const float4 xl = p.ca4[(size_t)b * N + p.R + l];
        const float dx = xr.x - xl.x, dy = xr.y - xl.y, dz = xr.z - xl.z;
        const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float *Ql = p.Q + ((size_t)b * N + p.R + l) * H;     // wave-uniform: scalar loads
        float mean, rstd;
        if (EXACT) {
            float s = 0.f;
#pragma unroll 16
            for (int c = 0; c < H; ++c) s += (Pl[c * PT_LD] + Ql[c]) + p.w_d[c] * D;
            mean = s * (1.0f / H);
            float v = 0.f;
#pragma unroll 16
            for (int c = 0; c < H; ++c) {
                const float d = ((Pl[c * PT_LD] + Ql[c]) + p.w_d[c] * D) - mean;
                v += d * d;
            }
            rstd = 1.0f / sqrtf(v * (1.0f / H) + 1e-5f);
        } else {
            // sum_c z and sum_c z^2 of z = P + Q + w_d D from the channel moments of the two rows and their dot product:
            //   sum z   = sP + sQ + D sw
            //   sum z^2 = sP2 + sQ2 + 2 P.Q + 2 D (sPw + sQw) + D^2 sw2
            const float4 q4 = *reinterpret_cast<const float4 *>(Ql + lane * 4);
            const float sQ = wave_sum((q4.x + q4.y) + (q4.z + q4.w));
            const float sQ2 = wave_sum((q4.x * q4.x + q4.y * q4.y) + (q4.z * q4.z + q4.w * q4.w));
            const float sQw = wave_sum((q4.x * wd4.x + q4.y * wd4.y) + (q4.z * wd4.z + q4.w * wd4.w));
            float dot0 = 0.f, dot1 = 0.f;
#pragma unroll 16
            for (int c = 0; c < H; c += 2) {
                dot0 = fmaf(Pl[c * PT_LD], Ql[c], dot0);
                dot1 = fmaf(Pl[(c + 1) * PT_LD], Ql[c + 1], dot1);
            }
            mean = ((mP + sQ) + D * sum_w) * (1.0f / H);
            const float ez2 = (((mP2 + sQ2) + 2.0f * (dot0 + dot1)) + D * (2.0f * (mPw + sQw) + D * sum_w2)) * (1.0f / H);
            rstd = __builtin_amdgcn_rsqf(fmaxf(ez2 - mean * mean, 0.f) + 1e-5f);
        }
        float o = 0.f;
        if (EXACT) {
#pragma unroll 16
            for (int c = 0; c < H; ++c) {
                const float z = (Pl[c * PT_LD] + Ql[c]) + p.w_d[c] * D;
                const float y = (z - mean) * rstd * p.ln_w[c] + p.ln_b[c];
                o += silu_exact(y) * p.w3[c];
            }
        } else {
            const float nm = -mean * rstd;
#pragma unroll 16
            for (int c = 0; c < H; ++c) {
                const float z = fmaf(p.w_d[c], D, Pl[c * PT_LD]) + Ql[c];
                const float y = fmaf(fmaf(z, rstd, nm), p.ln_w[c], p.ln_b[c]);
                const float e = __builtin_amdgcn_exp2f(y * -1.44269504088896f);
                o = fmaf(y * __builtin_amdgcn_rcpf(1.0f + e), p.w3[c], o);
            }
        }
        if (!valid) o = 0.f;
        if (p.mode == 0) {
            // fij = F.normalize(vec) * s ; the tile's share of sum_r fij for this ligand residue
            const float inv = 1.0f / fmaxf(D, 1e-12f);
            const float fx = wave_sum(dx * inv * o), fy = wave_sum(dy * inv * o), fz = wave_sum(dz * inv * o);
            if (lane == 0) {
                float *fo = p.fpart + (((size_t)b * RT + rt) * p.L + l) * 3;
                fo[0] = fx; fo[1] = fy; fo[2] = fz;
            }
            clash += (valid && D <= 3.0f) ? 1 : 0;
        } else if (p.mode == 1) {
            if (valid && D < p.cut_off) { s_acc += (double)o; c_acc += 1.0; }
        } else {
            s_acc += (double)o;
        }
    }
    if (p.mode == 0) {
        const int tot = (int)wave_sum((float)clash);       // <= 64 * L / 4 per wave: exact in fp32
        if (lane == 0) p.clash_part[(size_t)b * RT * 4 + rt * 4 + wave] = tot;
    } else {
        const double st = wave_sum_d(s_acc), ct = wave_sum_d(c_acc);
        if (lane == 0) {
            float *so = p.spart + ((size_t)b * RT * 4 + rt * 4 + wave) * 2;
            so[0] = (float)st; so[1] = (float)ct;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The pair head of every engine (r04; fp32 throughout): the rank-4 part of the pre-activation on the matrix pipe.  (DFM_PAIR_HEAD_VALU=1 selects
// k_pair_head<EXACT> above: all-VALU, three-pass LayerNorm + expf / division for the fp32 engine.)
// With S = -log2(e) folded in (SiLU as exp2 -> +1 -> rcp -> mul, see SILU_S in kernels_edge.hip) the LayerNorm output of pair (r, l) is
//     y'_c = rstd P''[r][c]  +  [rstd D] wd''_c + [rstd] Q''[l][c] + [-mean rstd] lnw''_c + [1] lnb''_c          ('' = times S ln_w_c; lnb'' = S ln_b)
// i.e. a per-row scaling of the resident P'' tile plus a K = 4 outer product of per-pair scalars (rstd D, rstd, -mean rstd, 1) with four
// channel vectors that depend on l only: two v_mfma_f32_32x32x2_f32 per 32 channels x 32 receptor residues (exact fp32) replace four
// VALU operations per pair and channel, and the channel-major C layout (lane = receptor residue, 16 channels per lane and block) keeps
// the w3 dot product inside the lane.  The pair's LayerNorm statistics come from row moments and the dot product P_r . Q_l as before
// (k_pair_head<0>), the dot products of a workgroup's 32 x 64 pairs from one fp32 MFMA pass over the raw tile before it is scaled in place.
// Per pair and channel the VALU is left with fma, exp2, add, rcp, mul, fma + a quarter of two LDS reads: 4.82 -> 2.26 ms per launch at
// 300+300, B = 256, second family 310 -> 358 traj/s (profiles/r04_pair_head.txt; VALU busy 0.77, 79 % of the issue floor of its mix).
//   workgroup = (32 receptor residues, 64 ligand residues, trajectory), four waves striding over the ligand residues; P'' sits in LDS as
//   [channel quad][r][4] (conflict-free 16-byte reads in the C layout, 32 KiB), three workgroups per CU.
//   Output: the scalar s(r, l) as S[b][l][Rp] - the reductions over r (force, clashes, masked energy sum, confidence) are
//   k_pair_finish_s's, in a fixed order: no per-pair wave reductions here, no atomics, batch-invariant.
constexpr int PM_RT = 32, PM_LC = 64, PM_LW = PM_LC / 4;

struct PairMArgs {
    const float *P, *Q;
    const float4 *ca4;
    int R, L, Rp;
    const float *w_d, *ln_w, *ln_b, *w3;
    float *S;                  // [B][L][Rp]
};

__global__ __launch_bounds__(256, 3) void k_pair_head_m(PairMArgs p)
{
    constexpr float SS = -1.44269504088896340736f;      // SILU_S
    __shared__ __attribute__((aligned(16))) float Pl[H * PM_RT];      // [c / 4][r][4]
    __shared__ float dots[4][PM_LW][PM_RT];
    __shared__ float momp[8][3][PM_RT];
    __shared__ __attribute__((aligned(16))) float w3s[H];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, r32 = lane & 31;
    const int rt = blockIdx.x, lc = blockIdx.y, b = blockIdx.z, N = p.R + p.L;
    const int l_begin = lc * PM_LC, l_end = l_begin + PM_LC < p.L ? l_begin + PM_LC : p.L;
    const int r = rt * PM_RT + r32;
    const bool valid = r < p.R;

    // ---- stage the raw tile (thread = (receptor residue, eight channel quads)), row moments on the way
    {
        const int grp = tid >> 5;
        const float *prow = p.P + ((size_t)b * N + (valid ? r : 0)) * H;
        float s1 = 0.f, s2 = 0.f, sw = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c4 = grp + 8 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) v = *reinterpret_cast<const float4 *>(prow + c4 * 4);
            const float4 wd = *reinterpret_cast<const float4 *>(p.w_d + c4 * 4);
            *reinterpret_cast<float4 *>(&Pl[(c4 * PM_RT + r32) * 4]) = v;
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            sw += (v.x * wd.x + v.y * wd.y) + (v.z * wd.z + v.w * wd.w);
        }
        momp[grp][0][r32] = s1; momp[grp][1][r32] = s2; momp[grp][2][r32] = sw;
        if (tid < 64) {
            const float4 w = *reinterpret_cast<const float4 *>(p.w3 + tid * 4);
            *reinterpret_cast<float4 *>(&w3s[tid * 4]) = make_float4(w.x * (1.0f / SS), w.y * (1.0f / SS), w.z * (1.0f / SS), w.w * (1.0f / SS));
        }
    }
    __syncthreads();
    float mP = 0.f, mP2 = 0.f, mPw = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) { mP += momp[g][0][r32]; mP2 += momp[g][1][r32]; mPw += momp[g][2][r32]; }
    float sum_w, sum_w2;
    {
        const float4 wd4 = *reinterpret_cast<const float4 *>(p.w_d + lane * 4);
        sum_w = wave_sum((wd4.x + wd4.y) + (wd4.z + wd4.w));
        sum_w2 = wave_sum((wd4.x * wd4.x + wd4.y * wd4.y) + (wd4.z * wd4.z + wd4.w * wd4.w));
    }

    // ---- dot products P_r . Q_l of the wave's 16 ligand residues (rows 16..31 of the 32 x 32 product idle) + the moments of those rows.
    // K-step (pg, e): k = 0 / 1 <-> channel ((2 pg + k) * 4 + e) for BOTH operands (any pairing of k with channels is a valid contraction order)
    float sQv, sQ2v, sQwv;
    {
        const int lrow = l_begin + wave + 4 * r32;
        const bool lv = r32 < PM_LW && lrow < l_end;
        const float *qrow = p.Q + ((size_t)b * N + p.R + (lv ? lrow : 0)) * H;
        f32x16 dacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dacc[i] = 0.f;
        float s1 = 0.f, s2 = 0.f, sw = 0.f;
#pragma unroll 4
        for (int pg = 0; pg < 32; ++pg) {
            const int c4 = pg * 2 + hh;
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lv) q4 = *reinterpret_cast<const float4 *>(qrow + c4 * 4);
            const float4 wd = *reinterpret_cast<const float4 *>(p.w_d + c4 * 4);
            const float4 p4 = *reinterpret_cast<const float4 *>(&Pl[(c4 * PM_RT + r32) * 4]);
            s1 += (q4.x + q4.y) + (q4.z + q4.w);
            s2 += (q4.x * q4.x + q4.y * q4.y) + (q4.z * q4.z + q4.w * q4.w);
            sw += (q4.x * wd.x + q4.y * wd.y) + (q4.z * wd.z + q4.w * wd.w);
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, p4.x, dacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, p4.y, dacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, p4.z, dacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, p4.w, dacc, 0, 0, 0);
        }
        sQv = s1 + __shfl_xor(s1, 32); sQ2v = s2 + __shfl_xor(s2, 32); sQwv = sw + __shfl_xor(sw, 32);
        // C layout: lane (hh, n = r32) holds rows m = 8 (i / 4) + 4 hh + i % 4
#pragma unroll
        for (int i = 0; i < 8; ++i) dots[wave][8 * (i >> 2) + 4 * hh + (i & 3)][r32] = dacc[i];
    }
    __syncthreads();
    // ---- scale the tile in place: P'' = S ln_w P
    {
        const int grp = tid >> 5;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c4 = grp + 8 * q;
            const float4 w = *reinterpret_cast<const float4 *>(p.ln_w + c4 * 4);
            float4 *dst = reinterpret_cast<float4 *>(&Pl[(c4 * PM_RT + r32) * 4]);
            float4 v = *dst;
            v.x *= SS * w.x; v.y *= SS * w.y; v.z *= SS * w.z; v.w *= SS * w.w;
            *dst = v;
        }
    }
    // per-block operand registers (A layout of 32x32x2: lane = (m = channel in block, k = hh))
    float A2[8], aux[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        const int c = cb * 32 + r32;
        const float lw = SS * p.ln_w[c];
        A2[cb] = hh ? SS * p.ln_b[c] : lw;              // k = 0: lnw'' (x -mean rstd), k = 1: lnb'' (x 1)
        aux[cb] = hh ? lw : lw * p.w_d[c];              // k = 0: wd'' (x rstd D), k = 1: the factor that turns Q into Q'' (x rstd)
    }
    __syncthreads();

    const float4 xr = p.ca4[(size_t)b * N + (valid ? r : 0)];
    const float4 *Pl4 = reinterpret_cast<const float4 *>(Pl);
    const float4 *w3s4 = reinterpret_cast<const float4 *>(w3s);
    float qn[8];
    {
        const int l0 = l_begin + wave < l_end ? l_begin + wave : l_end - 1;
        const float *Ql = p.Q + ((size_t)b * N + p.R + l0) * H;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) qn[cb] = Ql[cb * 32 + r32];
    }
    for (int it = 0; it < PM_LW; ++it) {
        const int l = l_begin + wave + 4 * it;
        if (l >= l_end) break;
        float A1[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) A1[cb] = hh ? qn[cb] * aux[cb] : aux[cb];
        {      // next ligand residue's row, in flight under this one's work
            const int ln = l + 4 < l_end ? l + 4 : l;
            const float *Ql = p.Q + ((size_t)b * N + p.R + ln) * H;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) qn[cb] = Ql[cb * 32 + r32];
        }
        const float4 xl = p.ca4[(size_t)b * N + p.R + l];
        const float dx = xr.x - xl.x, dy = xr.y - xl.y, dz = xr.z - xl.z;
        const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float sq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sQv), it));
        const float sq2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sQ2v), it));
        const float sqw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sQwv), it));
        const float dot = dots[wave][it][r32];
        const float mean = ((mP + sq) + D * sum_w) * (1.0f / H);
        const float ez2 = (((mP2 + sq2) + 2.0f * dot) + D * (2.0f * (mPw + sqw) + D * sum_w2)) * (1.0f / H);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(ez2 - mean * mean, 0.f) + 1e-5f);
        const float B1 = hh ? rstd : rstd * D, B2 = hh ? 1.0f : -mean * rstd;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[0], B2, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[0], B1, acc[0], 0, 0, 0);
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            if (cb + 1 < 8) {      // the next block's products run under this block's SiLUs
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[(cb + 1) & 1][i] = 0.f;
                acc[(cb + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[cb + 1], B2, acc[(cb + 1) & 1], 0, 0, 0);
                acc[(cb + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb + 1], B1, acc[(cb + 1) & 1], 0, 0, 0);
            }
            const f32x16 &a = acc[cb & 1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = (cb * 4 + g) * 2 + hh;
                const float4 p4 = Pl4[c4 * PM_RT + r32], w4 = w3s4[c4];
                const float y0 = fmaf(p4.x, rstd, a[g * 4 + 0]), y1 = fmaf(p4.y, rstd, a[g * 4 + 1]);
                const float y2 = fmaf(p4.z, rstd, a[g * 4 + 2]), y3 = fmaf(p4.w, rstd, a[g * 4 + 3]);
                const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y0)), r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y1));
                const float r2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y2)), r3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y3));
                o0 = fmaf(y0 * r0, w4.x, o0); o1 = fmaf(y1 * r1, w4.y, o1);
                o0 = fmaf(y2 * r2, w4.z, o0); o1 = fmaf(y3 * r3, w4.w, o1);
            }
        }
        float o = o0 + o1;
        o += __shfl_xor(o, 32);
        if (hh == 0 && valid) p.S[((size_t)b * p.L + l) * p.Rp + r] = o;
    }
}

// Reductions over the receptor residues of S[b][l][Rp] = s(r, l), one workgroup per trajectory, wave = ligand residues l = wave (mod 4),
// lanes = receptor residues in chunks of 64 (fixed order: per-lane running sums, then one wave reduction per l or per trajectory):
//   mode 0  fvec[b][l] = inv_pool sum_r normalize(x_r - x_l) s   (egnn_net.py:455-470) + clash count D <= 3 (:40-41)
//   mode 1  (sum, count) of s over D < cut_off                   (:430-441)
//   mode 2  conf[b] = mean of s                                  (:444)
// The scalar partials go to slots 0..3 of the n_part slots k_heads sums (the rest zeroed).
__global__ __launch_bounds__(256) void k_pair_finish_s(const float *__restrict__ S, const float4 *__restrict__ ca4, int R, int L, int Rp, int mode,
                                                       float cut_off, float inv_pool, int n_part, float *__restrict__ fvec,
                                                       float *__restrict__ spart, int32_t *__restrict__ clash_part, float *__restrict__ conf)
{
    // gridDim.y workgroups share a trajectory's ligand residues (modes 0 / 1: one partial slot per wave, 4 gridDim.y <= n_part; mode 2: 1)
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, N = R + L;
    const int slot = blockIdx.y * 4 + wave, nslot = gridDim.y * 4;
    __shared__ double red[4];
    double s_acc = 0, c_acc = 0;
    int clash = 0;
    for (int l = slot; l < L; l += nslot) {
        const float4 xl = ca4[(size_t)b * N + R + l];
        const float *Sl = S + ((size_t)b * L + l) * Rp;
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (int r = lane; r < R; r += 64) {
            const float4 xr = ca4[(size_t)b * N + r];
            const float s = Sl[r];
            const float dx = xr.x - xl.x, dy = xr.y - xl.y, dz = xr.z - xl.z;
            const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
            if (mode == 0) {
                const float w = s / fmaxf(D, 1e-12f);
                fx = fmaf(dx, w, fx); fy = fmaf(dy, w, fy); fz = fmaf(dz, w, fz);
                clash += D <= 3.0f ? 1 : 0;
            } else if (mode == 1) {
                if (D < cut_off) { s_acc += (double)s; c_acc += 1.0; }
            } else {
                s_acc += (double)s;
            }
        }
        if (mode == 0) {
            fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
            if (lane == 0) {
                float *fo = fvec + ((size_t)b * L + l) * 3;
                fo[0] = fx * inv_pool; fo[1] = fy * inv_pool; fo[2] = fz * inv_pool;
            }
        }
    }
    if (mode == 0) {
        const int tot = (int)wave_sum((float)clash);       // <= 64 * ceil(R / 64) * L / 4 per wave: exact in fp32 below 2^24
        if (lane == 0) clash_part[(size_t)b * n_part + slot] = tot;
        if (blockIdx.y == 0) for (int t = nslot + threadIdx.x; t < n_part; t += blockDim.x) clash_part[(size_t)b * n_part + t] = 0;
    } else {
        const double st = wave_sum_d(s_acc), ct = wave_sum_d(c_acc);
        if (mode == 1) {
            if (lane == 0) { spart[((size_t)b * n_part + slot) * 2] = (float)st; spart[((size_t)b * n_part + slot) * 2 + 1] = (float)ct; }
            if (blockIdx.y == 0) for (int t = nslot + threadIdx.x; t < n_part; t += blockDim.x) { spart[((size_t)b * n_part + t) * 2] = 0.f; spart[((size_t)b * n_part + t) * 2 + 1] = 0.f; }
        } else {
            if (lane == 0) red[wave] = st;
            __syncthreads();
            if (threadIdx.x == 0) conf[b] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / ((double)R * L));
        }
    }
}

// f[b][l] = agg over receptor tiles (fixed order) ; confidence[b] = mean over all pairs
__global__ __launch_bounds__(256) void k_pair_finish(const float *__restrict__ fpart, int RT, int L, float inv_pool,
                                                     float *__restrict__ fvec, const float *__restrict__ cpart, int R,
                                                     float *__restrict__ conf)
{
    const int b = blockIdx.x;
    if (fvec) {
        for (int q = threadIdx.x; q < L * 3; q += blockDim.x) {
            double s = 0;
            for (int t = 0; t < RT; ++t) s += fpart[((size_t)b * RT + t) * L * 3 + q];
            fvec[(size_t)b * L * 3 + q] = (float)(s * inv_pool);
        }
    }
    if (conf && threadIdx.x == 0) {
        double s = 0;
        for (int t = 0; t < RT * 4; ++t) s += cpart[((size_t)b * RT * 4 + t) * 2];
        conf[b] = (float)(s / ((double)R * L));
    }
}

// dist_logits (egnn_net.py:347-352,:447): the one pair head with a 64-wide output.  A training-loss input (DFMDock.py:196-215) that no
// sampler reads, evaluated only on request (DFM_F_DIST) and in exact fp32 in every engine: one workgroup per (trajectory, receptor
// residue), thread = channel, the ligand residues in a loop; LayerNorm as two exact block reductions, then the 256 -> 64 projection
// as four 64-channel partial dots per output (w3 transposed on the host: lanes read consecutive outputs).
__global__ __launch_bounds__(256) void k_pair_dist(const float *__restrict__ P, const float *__restrict__ Q, const float4 *__restrict__ ca4,
                                                   int R, int L, const float *__restrict__ w_d, const float *__restrict__ ln_w,
                                                   const float *__restrict__ ln_b, const float *__restrict__ w3t, float *__restrict__ out)
{
    __shared__ float y[H], red[4], part[4][64];
    const int r = blockIdx.x, b = blockIdx.y, N = R + L, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const float pc = P[((size_t)b * N + r) * H + c], wd = w_d[c], lw = ln_w[c], lb = ln_b[c];
    const float4 xr = ca4[(size_t)b * N + r];
    for (int l = 0; l < L; ++l) {
        const float4 xl = ca4[(size_t)b * N + R + l];
        const float dx = xr.x - xl.x, dy = xr.y - xl.y, dz = xr.z - xl.z;
        const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float z = (pc + Q[((size_t)b * N + R + l) * H + c]) + wd * D;
        float s = wave_sum(z);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / H);
        __syncthreads();
        const float d = z - mean;
        s = wave_sum(d * d);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float var = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / H);
        y[c] = silu_exact(d / sqrtf(var + 1e-5f) * lw + lb);
        __syncthreads();
        float acc = 0.f;
        for (int k = 0; k < 64; ++k) acc = fmaf(y[wave * 64 + k], w3t[(size_t)(wave * 64 + k) * 64 + lane], acc);
        part[wave][lane] = acc;
        __syncthreads();
        if (c < 64) out[(((size_t)b * R + r) * L + l) * 64 + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    }
}

hipError_t launch_pair_dist(const float *P, const float *Q, const float4 *ca4, int B, int R, int L, const float *w_d, const float *ln_w,
                            const float *ln_b, const float *w3t, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_pair_dist, dim3(R, B), dim3(256), 0, s, P, Q, ca4, R, L, w_d, ln_w, ln_b, w3t, out);
    return hipGetLastError();
}

hipError_t launch_pair_head(const PairArgs &a, hipStream_t s)
{
    static std::atomic<bool> done0[MAX_DEVICES], done1[MAX_DEVICES];
    {
        hipError_t e = a.exact ? ensure_lds_attr(reinterpret_cast<const void *>(k_pair_head<1>), PAIR_LDS_BYTES, done1)
                               : ensure_lds_attr(reinterpret_cast<const void *>(k_pair_head<0>), PAIR_LDS_BYTES, done0);
        if (e != hipSuccess) return e;
    }
    PairKArgs k;
    k.P = a.P; k.Q = a.Q; k.ca4 = a.ca4; k.R = a.R; k.L = a.L; k.w_d = a.w_d; k.ln_w = a.ln_w; k.ln_b = a.ln_b; k.w3 = a.w3;
    k.mode = a.mode; k.cut_off = a.cut_off; k.fpart = a.fpart; k.spart = a.spart; k.clash_part = a.clash_part;
    const dim3 grid((a.R + 63) / 64, a.B);
    if (a.exact) hipLaunchKernelGGL(k_pair_head<1>, grid, dim3(256), PAIR_LDS_BYTES, s, k);
    else hipLaunchKernelGGL(k_pair_head<0>, grid, dim3(256), PAIR_LDS_BYTES, s, k);
    return hipGetLastError();
}

hipError_t launch_pair_head_m(const PairArgs &a, hipStream_t s)
{
    PairMArgs k;
    k.P = a.P; k.Q = a.Q; k.ca4 = a.ca4; k.R = a.R; k.L = a.L; k.Rp = a.Rp; k.w_d = a.w_d; k.ln_w = a.ln_w; k.ln_b = a.ln_b; k.w3 = a.w3; k.S = a.S;
    const dim3 grid((a.R + PM_RT - 1) / PM_RT, (a.L + PM_LC - 1) / PM_LC, a.B);
    hipLaunchKernelGGL(k_pair_head_m, grid, dim3(256), 0, s, k);
    return hipGetLastError();
}

hipError_t launch_pair_finish_s(const PairArgs &a, int n_part, float inv_pool, float *fvec, float *conf, hipStream_t s)
{
    // split a trajectory's ligand residues over up to 8 workgroups (a partial slot per wave: 4 per workgroup of the n_part the heads sum; the
    // confidence is one number per trajectory: one workgroup).  A function of (R, L) only: batch-invariant.
    int ls = a.mode == 2 ? 1 : n_part / 4;
    ls = ls < 1 ? 1 : (ls > 8 ? 8 : ls);
    while (ls > 1 && ls * 4 > a.L) --ls;
    hipLaunchKernelGGL(k_pair_finish_s, dim3(a.B, ls), dim3(256), 0, s, a.S, a.ca4, a.R, a.L, a.Rp, a.mode, a.cut_off, inv_pool, n_part, fvec, a.spart,
                       a.clash_part, conf);
    return hipGetLastError();
}

hipError_t launch_pair_finish(const float *fpart, int B, int R, int L, float inv_pool, float *fvec, const float *cpart,
                              float *conf, hipStream_t s)
{
    hipLaunchKernelGGL(k_pair_finish, dim3(B), dim3(256), token_lds(), s, fpart, (R + 63) / 64, L, inv_pool, fvec, cpart, R, conf);
    return hipGetLastError();
}

}  // namespace dfm
