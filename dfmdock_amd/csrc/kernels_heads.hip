// kernels_heads.hip - energy head over receptor x ligand pairs, force/torque pooling, time embedding,
// learned score scales, and the fused Euler-Maruyama / SO(3) pose update.
// Reference: src/models/score_net_mlsb.py:362,:386-411,:162-172; src/inference_base.py:428-456,
// src/utils/r3_diffuser.py:40-55, src/utils/so3_diffuser.py:344-369.
#include "dfm_device.h"
#include "dfm_internal.h"

namespace dfm {

// ------------------------------------------------------------------------------------------------
// to_energy on the pairs with CA distance < cut_off: Linear(cat[h_r,h_l]) = enA[r] + enB[l] (no bias)
// -> LayerNorm(256) -> SiLU -> Linear(256 -> 1).  Also counts clashes (D <= 3.0, score_net_mlsb.py:72).
// enA / enB are [B][N][256] (projections of every node; receptor rows of enA and ligand rows of enB are used).
// grid (R, B); 4 waves stride over the ligand residues; one wave evaluates one pair (4 channels / lane).
__global__ __launch_bounds__(256) void k_energy_pairs(const float *__restrict__ enA, const float *__restrict__ enB,
                                                      const float4 *__restrict__ ca4, int R, int L, float cut_off,
                                                      const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                      const float *__restrict__ w3, int want_energy,
                                                      float *__restrict__ en_part, int32_t *__restrict__ clash_part)
{
    __shared__ double s_e[4];
    __shared__ int s_c[4], s_k[4];
    const int r = blockIdx.x, b = blockIdx.y, N = R + L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 xr = ca4[(size_t)b * N + r];
    float4 a4 = make_float4(0, 0, 0, 0), lw = a4, lb = a4, ww = a4;
    if (want_energy) {
        a4 = *reinterpret_cast<const float4 *>(enA + ((size_t)b * N + r) * H + lane * 4);
        lw = *reinterpret_cast<const float4 *>(ln_w + lane * 4);
        lb = *reinterpret_cast<const float4 *>(ln_b + lane * 4);
        ww = *reinterpret_cast<const float4 *>(w3 + lane * 4);
    }
    double esum = 0;
    int cnt = 0, clash = 0;
    for (int l = wave; l < L; l += 4) {
        const float4 xl = ca4[(size_t)b * N + R + l];
        const float dx = xr.x - xl.x, dy = xr.y - xl.y, dz = xr.z - xl.z;
        const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
        clash += (D <= 3.0f) ? 1 : 0;
        if (D < cut_off) {
            cnt += 1;
            if (want_energy) {
                const float4 b4 = *reinterpret_cast<const float4 *>(enB + ((size_t)b * N + R + l) * H + lane * 4);
                float v0 = a4.x + b4.x, v1 = a4.y + b4.y, v2 = a4.z + b4.z, v3 = a4.w + b4.w;
                const float mean = wave_sum((v0 + v1) + (v2 + v3)) * (1.0f / H);
                v0 -= mean; v1 -= mean; v2 -= mean; v3 -= mean;
                const float var = wave_sum((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3)) * (1.0f / H);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                const float y0 = v0 * rstd * lw.x + lb.x, y1 = v1 * rstd * lw.y + lb.y, y2 = v2 * rstd * lw.z + lb.z,
                            y3 = v3 * rstd * lw.w + lb.w;
                const float e = wave_sum((silu_exact(y0) * ww.x + silu_exact(y1) * ww.y) +
                                         (silu_exact(y2) * ww.z + silu_exact(y3) * ww.w));
                esum += (double)e;
            }
        }
    }
    if (lane == 0) { s_e[wave] = esum; s_c[wave] = cnt; s_k[wave] = clash; }
    __syncthreads();
    if (threadIdx.x == 0) {
        en_part[((size_t)b * R + r) * 2 + 0] = (float)(s_e[0] + s_e[1] + s_e[2] + s_e[3]);
        en_part[((size_t)b * R + r) * 2 + 1] = (float)(s_c[0] + s_c[1] + s_c[2] + s_c[3]);
        clash_part[(size_t)b * R + r] = s_k[0] + s_k[1] + s_k[2] + s_k[3];
    }
}

hipError_t launch_energy_pairs(const float *enA, const float *enB, const float4 *ca4, int B, int R, int L, float cut_off,
                               const HeadsDev *hw, int want_energy, float *en_part, int32_t *clash_part, hipStream_t s)
{
    hipLaunchKernelGGL(k_energy_pairs, dim3(R, B), dim3(256), 0, s, enA, enB, ca4, R, L, cut_off, hw->en_ln_w,
                       hw->en_ln_b, hw->en_w3, want_energy, en_part, clash_part);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Time-dependent half of the two score-scale MLPs, off the per-evaluation critical path: for every time t[n]
//   t_embed = Sigmoid(Linear(GaussianFourierProjection(t)))                  (score_net_mlsb.py:407, :162-172)
//   base[n][g][c] = sum_k t_embed[k] * W_g[c][1 + k]      g = 0 translation / 1 rotation scale net, first Linear(129 -> 128)
// so that k_heads only adds the norm column: hid = W_g[c][0] * |pred| + base.  dfm_sample calls it ONCE for its whole time grid
// (every trajectory of a step shares t), dfm_score once per call for its B times.  r01-r03 evaluated both Linears inside k_heads
// with one thread per output row: 256 dependent, uncoalesced row reads per evaluation (64 cache lines per wave instruction) - most
// of that kernel's 25-33 us, which small batches cannot hide.  Here a wave owns an output and reads its row coalesced.
// (1024 threads, four outputs per wave in flight: with 256 threads and one output per iteration the kernel was 96 dependent L2 round
// trips long - 50-60 us, 11 % of a dfm_score call at B = 1)
__global__ __launch_bounds__(1024) void k_time_embed(const float *__restrict__ t, HeadsDev hw, float *__restrict__ base)
{
    __shared__ float s_four[HI], s_temb[HI];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = 16;
    const float tv = t[n];
    if (tid < HI / 2) {
        const float xp = ((tv * hw.t_W[tid]) * 2.0f) * 3.14159265358979323846f;
        s_four[tid] = sinf(xp);
        s_four[HI / 2 + tid] = cosf(xp);
    }
    __syncthreads();
    {
        const float f0 = s_four[lane], f1 = s_four[64 + lane];
        float p[HI / 16];      // HI / NW outputs per wave, their row loads all in flight
#pragma unroll
        for (int u = 0; u < HI / 16; ++u) {
            const float *row = hw.t_lin + (size_t)(wave + u * NW) * HI;
            p[u] = fmaf(f0, row[lane], f1 * row[64 + lane]);
        }
#pragma unroll
        for (int u = 0; u < HI / 16; ++u) {
            const float v = wave_sum(p[u]);
            if (lane == 0) s_temb[wave + u * NW] = sigmoid_exact(v);
        }
    }
    __syncthreads();
    {
        const float e0 = s_temb[lane], e1 = s_temb[64 + lane];
        float p[2 * HI / 16];
#pragma unroll
        for (int u = 0; u < 2 * HI / 16; ++u) {
            const int q = wave + u * NW, g = q >> 7, c = q & (HI - 1);
            const float *row = (g ? hw.rots0 : hw.trs0) + (size_t)c * (HI + 1) + 1;
            p[u] = fmaf(e0, row[lane], e1 * row[64 + lane]);
        }
#pragma unroll
        for (int u = 0; u < 2 * HI / 16; ++u) {
            const float v = wave_sum(p[u]);
            if (lane == 0) base[(size_t)n * (2 * HI) + wave + u * NW] = v;
        }
    }
}

hipError_t launch_time_embed(const float *t_dev, int n, const HeadsDev *hw, float *base, hipStream_t s)
{
    hipLaunchKernelGGL(k_time_embed, dim3(n), dim3(1024), 0, s, t_dev, *hw, base);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
struct HeadKArgs {
    const float *fvec;
    const float4 *ca4;
    int R, L;
    const float *hid_base;       // [.][2][128] from k_time_embed; trajectory b reads entry b * hid_bstride (0: one time for the batch)
    long long hid_bstride;
    HeadsDev hw;
    float *scores;
    int want_energy;
    const float *en_part;
    const int32_t *clash_part;
    int n_part, en_mode;
    float pool_div;
    int do_update;
    float g2_r, g_r, hg2_r, g2_t, g_t, hg2_t, dt, sqrt_dt, rot_noise, tr_noise;
    int ode;
    const float *z_rot, *z_tr;
    long long z_bstride;
    uint32_t seed_lo, seed_hi, step;
    int all_atoms;
    float *lig_cur, *tr_update, *rot_update;
    float *trace_pose;
    long long trace_bstride;
    float *trace_scores;
    long long trace_s_bstride;
    const StepParams *step_params;      // replayed step graph (HeadArgs::ctl): this step's scalars, seed and time embedding from device memory
    const uint32_t *ctl;
    // prep_next: after the update, prepare the NEW pose for the next evaluation right here (what k_prep_pose would do as that
    // evaluation's first launch: same workgroup shape, same code) - one dependent launch less per step
    int prep_next;
    const float *rec_pos;
    float4 *prep_pos;
    float4 *prep_ca4, *prep_cb4;
};

// sum over the 128 threads of a group (two waves); scratch[4]
__device__ inline float group_sum(float v, float *scratch, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    const int g = tid >> 7;
    return scratch[g * 2] + scratch[g * 2 + 1];
}

// NV block sums (256 threads) behind ONE pair of barriers; per value the arithmetic of block_sum_d (wave butterfly, then the four
// wave sums in wave order)
template <int NV> __device__ inline void block_sum_dn(double (&v)[NV], double *scratch /*[4 * NV]*/)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) scratch[w * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((scratch[k] + scratch[NV + k]) + scratch[2 * NV + k]) + scratch[3 * NV + k];
}

__global__ __launch_bounds__(256) void k_heads(HeadKArgs p)
{
    __shared__ double dscr[4 * 12];
    __shared__ float s_red[4], s_pred[8], s_score[8], s_upd[16], s_center[3];
    const int b = blockIdx.x, tid = threadIdx.x, R = p.R, L = p.L, N = R + L;
    float *lig = p.lig_cur + (size_t)b * L * 9;
    const float *hid_base = p.hid_base + (size_t)b * p.hid_bstride;
    if (p.ctl) {      // the captured launch is the same in every step: what differs between steps is read here
        const uint32_t idx = p.ctl[0] - 1u;
        const StepParams q = p.step_params[idx];
        p.g2_r = q.g2_r; p.g_r = q.g_r; p.hg2_r = q.hg2_r; p.g2_t = q.g2_t; p.g_t = q.g_t; p.hg2_t = q.hg2_t;
        p.dt = q.dt; p.sqrt_dt = q.sqrt_dt; p.rot_noise = q.rot_noise; p.tr_noise = q.tr_noise; p.step = q.step;
        p.seed_lo = p.ctl[1]; p.seed_hi = p.ctl[2];
        hid_base = p.hid_base + (size_t)idx * (2 * HI);
    }

    // every reduction over the trajectory's residues in one pass and one exchange (r01-r03: fifteen block sums in sequence):
    //  [0..5]  :396-404  f = pos_out[lig] - r ;  tr_pred = mean f ; rot_pred = mean (r x f)
    //  [6..8]  centre of the pose BEFORE this step's update (modify_coords, inference_base.py:342-352)
    //  [9..11] energy = sum(e * mask) / (sum(mask) + 1e-6) ; num_clashes
    double a[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = tid; q < L; q += blockDim.x) {
        const float *f = p.fvec + ((size_t)b * L + q) * 3;
        const float4 r = p.ca4[(size_t)b * N + R + q];
        a[0] += f[0]; a[1] += f[1]; a[2] += f[2];
        a[3] += r.y * f[2] - r.z * f[1];
        a[4] += r.z * f[0] - r.x * f[2];
        a[5] += r.x * f[1] - r.y * f[0];
    }
    if (p.do_update) {
        if (p.all_atoms) {   // second family: centre = mean over all backbone atoms (src/inference.py:245, DFMDock.py:247)
            for (int q = tid; q < L * 3; q += blockDim.x) { a[6] += lig[q * 3]; a[7] += lig[q * 3 + 1]; a[8] += lig[q * 3 + 2]; }
        } else {
            for (int q = tid; q < L; q += blockDim.x) { a[6] += lig[q * 9 + 3]; a[7] += lig[q * 9 + 4]; a[8] += lig[q * 9 + 5]; }
        }
    }
    if (p.want_energy) {
        for (int r = tid; r < p.n_part; r += blockDim.x) {
            a[9] += p.en_part[((size_t)b * p.n_part + r) * 2];
            a[10] += p.en_part[((size_t)b * p.n_part + r) * 2 + 1];
            a[11] += p.clash_part[(size_t)b * p.n_part + r];
        }
    }
    block_sum_dn<12>(a, dscr);
    if (tid < 6) s_pred[tid] = (float)(a[tid] / p.pool_div);
    if (tid == 0) {
        if (p.want_energy) {
            s_score[6] = p.en_mode == 0 ? (float)a[9] / ((float)a[10] + 1e-6f)
                                        : (p.en_mode == 1 ? (float)a[9] / fmaxf((float)a[10], 1.0f) : (float)a[9]);
            s_score[7] = (float)a[11];
        } else {
            s_score[6] = 0.f; s_score[7] = 0.f;
        }
    }
    __syncthreads();

    // :408-411  two scale MLPs in parallel: threads 0..127 translation, 128..255 rotation; the t_embed half of the first Linear
    // comes from k_time_embed
    {
        const int g = tid >> 7, c = tid & 127;
        const float *pred = s_pred + g * 3;
        const float *w0 = g ? p.hw.rots0 : p.hw.trs0;
        const float *lw = g ? p.hw.rots_ln_w : p.hw.trs_ln_w, *lb = g ? p.hw.rots_ln_b : p.hw.trs_ln_b;
        const float *w4 = g ? p.hw.rots4 : p.hw.trs4;
        const float nrm = sqrtf((pred[0] * pred[0] + pred[1] * pred[1]) + pred[2] * pred[2]);
        const float hid = fmaf(w0[c * (HI + 1)], nrm, hid_base[tid]);
        const float mean = group_sum(hid, s_red, tid) * (1.0f / HI);
        const float d = hid - mean;
        const float var = group_sum(d * d, s_red, tid) * (1.0f / HI);
        const float y = d * (1.0f / sqrtf(var + 1e-5f)) * lw[c] + lb[c];
        const float o = group_sum(silu_exact(y) * w4[c], s_red, tid);
        if (c < 3) {
            const float sp = o > 20.0f ? o : log1pf(expf(o));   // Softplus(beta=1, threshold=20)
            s_score[g * 3 + c] = pred[c] / (nrm + 1e-6f) * sp;
        }
    }
    __syncthreads();
    if (tid < 8) {
        p.scores[(size_t)b * 8 + tid] = s_score[tid];
        if (p.trace_scores) p.trace_scores[(size_t)b * p.trace_s_bstride + tid] = s_score[tid];
    }
    if (!p.do_update) return;

    // ---- Euler-Maruyama step (inference_base.py:439-456) ---------------------------------------
    const double c0 = a[6], c1 = a[7], c2 = a[8];
    const int ncen = p.all_atoms ? L * 3 : L;
    if (tid == 0) {
        float zr[3], zt[3];
        if (p.z_rot) {
            for (int k = 0; k < 3; ++k) zr[k] = p.z_rot[(size_t)b * p.z_bstride + k];
        } else {
            const u32x4 r1 = philox4x32((uint32_t)b, p.step, 0u, RNG_NOISE, p.seed_lo, p.seed_hi);
            const float ra = sqrtf(-2.0f * logf(u01(r1.x))), rb = sqrtf(-2.0f * logf(u01(r1.z)));
            zr[0] = ra * cosf(6.283185307179586f * u01(r1.y));
            zr[1] = ra * sinf(6.283185307179586f * u01(r1.y));
            zr[2] = rb * cosf(6.283185307179586f * u01(r1.w));
        }
        if (p.z_tr) {
            for (int k = 0; k < 3; ++k) zt[k] = p.z_tr[(size_t)b * p.z_bstride + k];
        } else {
            const u32x4 r2 = philox4x32((uint32_t)b, p.step, 1u, RNG_NOISE, p.seed_lo, p.seed_hi);
            const float ra = sqrtf(-2.0f * logf(u01(r2.x))), rb = sqrtf(-2.0f * logf(u01(r2.z)));
            zt[0] = ra * cosf(6.283185307179586f * u01(r2.y));
            zt[1] = ra * sinf(6.283185307179586f * u01(r2.y));
            zt[2] = rb * cosf(6.283185307179586f * u01(r2.w));
        }
        // torch_reverse: float32 tensor ops with the float64 scalars g^2, g rounded to float32 (r3_diffuser.py:52-53)
        float rot[3], tr[3];
        for (int k = 0; k < 3; ++k) {
            if (!p.ode) {
                rot[k] = (p.g2_r * s_score[3 + k]) * p.dt + (p.g_r * p.sqrt_dt) * (p.rot_noise * zr[k]);
                tr[k] = (p.g2_t * s_score[k]) * p.dt + (p.g_t * p.sqrt_dt) * (p.tr_noise * zt[k]);
            } else {
                rot[k] = (p.hg2_r * s_score[3 + k]) * p.dt;
                tr[k] = (p.hg2_t * s_score[k]) * p.dt;
            }
        }
        float Rm[9];
        aa_to_mat(rot, Rm);
        for (int k = 0; k < 9; ++k) s_upd[k] = Rm[k];
        for (int k = 0; k < 3; ++k) s_upd[9 + k] = tr[k];
        s_upd[12] = (float)(c0 / ncen); s_upd[13] = (float)(c1 / ncen); s_upd[14] = (float)(c2 / ncen);
        // tr_update += tr ; rot_update = axis_angle(R(rot) @ R(rot_update))
        float ru[3] = {p.rot_update[b * 3], p.rot_update[b * 3 + 1], p.rot_update[b * 3 + 2]}, rn[3];
        rot_compose(ru, rot, rn);
        for (int k = 0; k < 3; ++k) {
            p.tr_update[b * 3 + k] += tr[k];
            p.rot_update[b * 3 + k] = rn[k];
        }
    }
    __syncthreads();
    // modify_coords (inference_base.py:342-352): x = (x - c) @ R^T + c ; x += tr
    float *tp = p.trace_pose ? p.trace_pose + (size_t)b * p.trace_bstride : nullptr;
    for (int at = tid; at < L * 3; at += blockDim.x) {
        const float v0 = lig[at * 3] - s_upd[12], v1 = lig[at * 3 + 1] - s_upd[13], v2 = lig[at * 3 + 2] - s_upd[14];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s = 0;
            s += v0 * s_upd[r * 3]; s += v1 * s_upd[r * 3 + 1]; s += v2 * s_upd[r * 3 + 2];
            const float o = (s + s_upd[12 + r]) + s_upd[9 + r];
            lig[at * 3 + r] = o;
            if (tp) tp[at * 3 + r] = o;
        }
    }
    if (p.prep_next) {
        __syncthreads();      // the workgroup's new pose is complete (and its reads of this evaluation's centred CA are long done)
        prep_pose_block(p.rec_pos, lig, R, L, p.all_atoms, p.prep_pos + (size_t)b * N, p.prep_ca4 + (size_t)b * N, p.prep_cb4 + (size_t)b * N,
                        dscr, s_center);
    }
}

hipError_t launch_heads(const HeadArgs &a, hipStream_t s)
{
    HeadKArgs k;
    k.fvec = a.fvec; k.ca4 = a.ca4; k.R = a.R; k.L = a.L; k.hid_base = a.hid_base; k.hid_bstride = a.hid_bstride; k.hw = *a.hw; k.scores = a.scores;
    k.want_energy = a.want_energy; k.en_part = a.en_part; k.clash_part = a.clash_part; k.do_update = a.do_update;
    k.n_part = a.n_part; k.en_mode = a.en_mode; k.pool_div = a.pool_div;
    k.g2_r = a.g2_r; k.g_r = a.g_r; k.hg2_r = a.hg2_r; k.g2_t = a.g2_t; k.g_t = a.g_t; k.hg2_t = a.hg2_t;
    k.dt = a.dt; k.sqrt_dt = a.sqrt_dt; k.rot_noise = a.rot_noise; k.tr_noise = a.tr_noise; k.ode = a.ode;
    k.z_rot = a.z_rot; k.z_tr = a.z_tr; k.z_bstride = a.z_bstride;
    k.seed_lo = (uint32_t)a.seed; k.seed_hi = (uint32_t)(a.seed >> 32); k.step = a.step; k.all_atoms = a.all_atoms;
    k.lig_cur = a.lig_cur; k.tr_update = a.tr_update; k.rot_update = a.rot_update;
    k.trace_pose = a.trace_pose; k.trace_bstride = a.trace_bstride;
    k.trace_scores = a.trace_scores; k.trace_s_bstride = a.trace_s_bstride;
    k.step_params = a.step_params; k.ctl = a.ctl;
    k.prep_next = a.prep_next; k.rec_pos = a.rec_pos; k.prep_pos = a.prep_pos; k.prep_ca4 = a.prep_ca4; k.prep_cb4 = a.prep_cb4;
    hipLaunchKernelGGL(k_heads, dim3(a.B), dim3(256), 0, s, k);
    return hipGetLastError();
}

}  // namespace dfm
