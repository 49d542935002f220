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

hipError_t launch_pair_finish(const float *fpart, int B, int R, int L, float inv_pool, float *fvec, const float *cpart,
                              float *conf, hipStream_t s)
{
    hipLaunchKernelGGL(k_pair_finish, dim3(B), dim3(256), 0, s, fpart, (R + 63) / 64, L, inv_pool, fvec, cpart, R, conf);
    return hipGetLastError();
}

}  // namespace dfm
