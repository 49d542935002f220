// dfm_device.h - device-side helpers: Philox4x32-10, bf16 pack/unpack, wave reductions, SO(3) maps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfm {

// ---- Philox4x32-10 (Salmon et al. 2011): counter-based, keyed by (seed) ---------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}
// uniform in (0,1): never 0 or 1
__host__ __device__ inline float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// RNG stream ids (counter word 3)
enum : uint32_t { RNG_EDGES = 1, RNG_NOISE = 2, RNG_INIT = 3 };

// ---- bf16 <-> f32 (round to nearest even) -----------------------------------------------------------
__host__ __device__ inline uint16_t f2bf(float f)
{
    union { float f; uint32_t u; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((v.u >> 16) | 0x40);   // NaN
    const uint32_t r = 0x7fffu + ((v.u >> 16) & 1u);
    return (uint16_t)((v.u + r) >> 16);
}
__host__ __device__ inline float bf2f(uint16_t b)
{
    union { float f; uint32_t u; } v;
    v.u = (uint32_t)b << 16;
    return v.f;
}
// float -> IEEE half (round to nearest even, saturating to +-65504: gathered operands must stay finite)
__host__ __device__ inline uint16_t f2h(float f)
{
    union { float f; uint32_t u; } v;
    v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u;
    uint32_t a = v.u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          // NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);         // >= 65520 rounds past max: saturate at 65504
    if (a < 0x33000001u) return (uint16_t)sign;                      // < 2^-25: zero
    if (a < 0x38800000u) {                                           // subnormal half
        const uint32_t shift = 126u - (a >> 23);                     // 14..24
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        m >>= shift;
        if (rem > halfway || (rem == halfway && (m & 1u))) m += 1u;
        return (uint16_t)(sign | m);
    }
    a += 0xc8000000u;                                                // rebias exponent (127 -> 15)
    const uint32_t rem = a & 0x1fffu;
    a >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (a & 1u))) a += 1u;
    return (uint16_t)(sign | a);
}

// IEEE half -> float (host side: weight packing)
__host__ __device__ inline float h2f(uint16_t hbits)
{
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16, e = (hbits >> 10) & 31u, m = hbits & 0x3ffu;
    union { float f; uint32_t u; } v;
    if (e == 0) {
        if (m == 0) { v.u = sign; return v.f; }
        float f = (float)m * (1.0f / 16777216.0f);   // m * 2^-24
        return sign ? -f : f;
    }
    if (e == 31) { v.u = sign | 0x7f800000u | (m << 13); return v.f; }
    v.u = sign | ((e + 112u) << 23) | (m << 13);
    return v.f;
}

__device__ inline float bflo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ inline float bfhi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// two fp32 -> packed fp16 on the hardware converter (round to nearest even), saturating at +-65504 like f2h (NaN stays NaN).
// f2h above is bit manipulation for host AND device (~30 instructions and several branches per value): fine for one-off uploads,
// not for a GEMM epilogue - the [Wa|Wb] projection spent half its time in it (profiles/r03_exp_gemm_variants.txt)
__device__ inline uint32_t pack_h2_sat(float a, float b)
{
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ inline uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

__device__ inline float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ inline float silu_exact(float x) { return x / (1.0f + expf(-x)); }
__device__ inline float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- wave (64-lane) reductions ------------------------------------------------------------------------
__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ inline double wave_sum_d(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// sum over the 32 lanes that share lane>>5
__device__ inline float half_sum(float v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// block-wide sum of a double via LDS scratch (>= blockDim/64 doubles); result broadcast to all threads
__device__ inline double block_sum_d(double v, double *scratch)
{
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// ---- pose preparation of ONE trajectory by one 256-thread workgroup: centre receptor + ligand on the ligand centroid
// (score_net_mlsb.py:353-359; all backbone atoms for the second family, DFMDock.py:254-257), build the CA and virtual-CB arrays
// (coords6d.py:71-75).  Used by k_prep_pose and - for the pose a step has just produced - by k_heads.  scratch: double[8], center: float[3]
__device__ inline void prep_pose_block(const float *__restrict__ rec_pos, const float *__restrict__ lig, int R, int L, int all_atoms,
                                       float4 *__restrict__ n4, float4 *__restrict__ ca4, float4 *__restrict__ cb4, double *scratch, float *center)
{
    const int N = R + L;
    double s0 = 0, s1 = 0, s2 = 0;
    if (all_atoms) {
        for (int q = threadIdx.x; q < L * 3; q += blockDim.x) { s0 += lig[q * 3]; s1 += lig[q * 3 + 1]; s2 += lig[q * 3 + 2]; }
    } else {
        for (int q = threadIdx.x; q < L; q += blockDim.x) { s0 += lig[q * 9 + 3]; s1 += lig[q * 9 + 4]; s2 += lig[q * 9 + 5]; }
    }
    s0 = block_sum_d(s0, scratch);
    s1 = block_sum_d(s1, scratch);
    s2 = block_sum_d(s2, scratch);
    if (threadIdx.x == 0) {
        const int cnt = all_atoms ? L * 3 : L;
        center[0] = (float)(s0 / cnt); center[1] = (float)(s1 / cnt); center[2] = (float)(s2 / cnt);
    }
    __syncthreads();
    const float cx = center[0], cy = center[1], cz = center[2];
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float *src = i < R ? rec_pos + (size_t)i * 9 : lig + (size_t)(i - R) * 9;
        float v[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v[a * 3 + 0] = src[a * 3 + 0] - cx;
            v[a * 3 + 1] = src[a * 3 + 1] - cy;
            v[a * 3 + 2] = src[a * 3 + 2] - cz;
        }
        // the centred backbone N (the one atom besides CA / CB the features read: theta = dih(N_i, CA_i, CB_i, CB_j)) as an ALIGNED
        // float4 like ca4 / cb4.  r01-r04 kept the whole centred backbone as [N][9] floats: 36-byte records written with 4-byte-
        // aligned 16-byte stores straddling cache lines - nothing but N was ever read back from it (r05).
        n4[i] = make_float4(v[0], v[1], v[2], 0.f);
        // Cb = -0.58273431*a + 0.56802827*b - 0.54067466*c + Ca ;  b = Ca - N, c = C - Ca, a = b x c
        const float bx = v[3] - v[0], by = v[4] - v[1], bz = v[5] - v[2];
        const float cx_ = v[6] - v[3], cy_ = v[7] - v[4], cz_ = v[8] - v[5];
        const float ax = by * cz_ - bz * cy_, ay = bz * cx_ - bx * cz_, az = bx * cy_ - by * cx_;
        float4 cb;
        cb.x = ((-0.58273431f * ax + 0.56802827f * bx) - 0.54067466f * cx_) + v[3];
        cb.y = ((-0.58273431f * ay + 0.56802827f * by) - 0.54067466f * cy_) + v[4];
        cb.z = ((-0.58273431f * az + 0.56802827f * bz) - 0.54067466f * cz_) + v[5];
        cb.w = 0.f;
        ca4[i] = make_float4(v[3], v[4], v[5], 0.f);
        cb4[i] = cb;
    }
}

// ---- SO(3) maps, float32, same operation order as the reference (src/utils/geometry.py) --------------
__device__ inline void aa_to_quat(const float aa[3], float q[4])
{   // geometry.py:154-183
    const float ang = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    const float half = 0.5f * ang;
    const float s = (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.0f) : (sinf(half) / ang);
    q[0] = cosf(half); q[1] = aa[0] * s; q[2] = aa[1] * s; q[3] = aa[2] * s;
}
__device__ inline void quat_to_mat(const float q[4], float R[9])
{   // geometry.py:18-45
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}
__device__ inline void aa_to_mat(const float aa[3], float R[9])
{
    float q[4];
    aa_to_quat(aa, q);
    quat_to_mat(q, R);
}
__device__ inline void mat_to_quat(const float R[9], float q[4])
{   // geometry.py:64-123
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
                m22 = R[8];
    float qa[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
#pragma unroll
    for (int c = 0; c < 4; ++c) qa[c] = qa[c] > 0 ? sqrtf(qa[c]) : 0.0f;
    int best = 0;
#pragma unroll
    for (int c = 1; c < 4; ++c) if (qa[c] > qa[best]) best = c;
    float cand[4];
    if (best == 0) { cand[0] = qa[0] * qa[0]; cand[1] = m21 - m12; cand[2] = m02 - m20; cand[3] = m10 - m01; }
    else if (best == 1) { cand[0] = m21 - m12; cand[1] = qa[1] * qa[1]; cand[2] = m10 + m01; cand[3] = m02 + m20; }
    else if (best == 2) { cand[0] = m02 - m20; cand[1] = m10 + m01; cand[2] = qa[2] * qa[2]; cand[3] = m12 + m21; }
    else { cand[0] = m10 - m01; cand[1] = m20 + m02; cand[2] = m21 + m12; cand[3] = qa[3] * qa[3]; }
    const float qb = qa[best];
    const float den = 2.0f * (qb > 0.1f ? qb : 0.1f);
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = cand[c] / den;
}
__device__ inline void quat_to_aa(const float q[4], float aa[3])
{   // geometry.py:126-151 (angle in [0, 2pi], not wrapped)
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float half = atan2f(n, q[0]);
    const float ang = 2.0f * half;
    const float s = (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.0f) : (sinf(half) / ang);
    aa[0] = q[1] / s; aa[1] = q[2] / s; aa[2] = q[3] / s;
}
__device__ inline void mat_to_aa(const float R[9], float aa[3])
{
    float q[4];
    mat_to_quat(R, q);
    quat_to_aa(q, aa);
}
// inference_base.py:311-316: axis_angle(R(r2) @ R(r1))
__device__ inline void rot_compose(const float r1[3], const float r2[3], float out[3])
{
    float R1[9], R2[9], Rm[9];
    aa_to_mat(r1, R1);
    aa_to_mat(r2, R2);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float s = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += R2[i * 3 + j] * R1[j * 3 + k];
            Rm[i * 3 + k] = s;
        }
    mat_to_aa(Rm, out);
}

}  // namespace dfm
