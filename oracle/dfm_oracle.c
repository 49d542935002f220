/*
 * dfm_oracle.c - CPU restatement (plain C, fp32, OpenMP) of the DFMDock sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY - see dfm_oracle.h.  It follows the reference
 * function by function, in the reference's own order of operations (dense
 * [E,641] edge MLP, concat node MLP), NOT the factorised form the HIP engine
 * uses, so that it is an independent check of the engine's algebra.  Every
 * function cites the reference lines it restates (paths relative to
 * /root/reference).  Pinned against golden vectors produced by running the
 * reference: tests/test_oracle_golden.py.
 */
#include "dfm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#include <malloc.h>
#endif

/* ------------------------------------------------------------------------- */
/* parameter blob (state_dict order: score_net_mlsb.py:249-341, egnn.py:37-93) */
typedef struct {
    const float *e1_w, *e1_b, *e2_w, *e2_b;      /* edge_mlp.0 / .2            */
    const float *n1_w, *n1_b;                    /* node_mlp.0                 */
    const float *gn_w, *gn_b, *gn_ms;            /* node_mlp.1 (GraphNorm)     */
    const float *n2_w, *n2_b;                    /* node_mlp.3                 */
    const float *c1_w, *c1_b, *c2_w;             /* coord_mlp (last layer)     */
    const float *att_w, *att_b;                  /* att_mlp.0                  */
} layer_w;

typedef struct {
    const float *single_embed, *spatial_embed, *positional_embed;
    layer_w layer[16];
    const float *en0_w, *en_ln_w, *en_ln_b, *en3_w;
    const float *fo0_w, *fo_ln_w, *fo_ln_b, *fo3_w;    /* family 1: to_force       */
    const float *cf0_w, *cf_ln_w, *cf_ln_b, *cf3_w;    /* family 1: to_confidence  */
    const float *di0_w, *di_ln_w, *di_ln_b, *di3_w;    /* family 1: to_dist (Linear(256 -> 64) last) */
    const float *ir0_w, *ir0_b, *ir2_w, *ir2_b, *ir4_w, *ir4_b;
    const float *t_W, *t_lin;
    const float *trs0_w, *trs_ln_w, *trs_ln_b, *trs4_w;
    const float *rots0_w, *rots_ln_w, *rots_ln_b, *rots4_w;
    int64_t total;
} net_w;

static void map_weights(const ora_hparams *hp, const float *blob, net_w *w)
{
    const int H = hp->node_dim, He = hp->edge_dim, Hi = hp->inner_dim;
    const float *p = blob;
#define TAKE(dst, n) do { (dst) = p; p += (int64_t)(n); } while (0)
    TAKE(w->single_embed, (int64_t)H * hp->lm_embed_dim);
    TAKE(w->spatial_embed, (int64_t)He * hp->spatial_embed_dim);
    TAKE(w->positional_embed, (int64_t)He * hp->positional_embed_dim);
    for (int l = 0; l < hp->depth; ++l) {
        layer_w *L = &w->layer[l];
        TAKE(L->e1_w, (int64_t)H * (2 * H + 1 + He)); TAKE(L->e1_b, H);
        TAKE(L->e2_w, (int64_t)H * H); TAKE(L->e2_b, H);
        TAKE(L->n1_w, (int64_t)H * 2 * H); TAKE(L->n1_b, H);
        TAKE(L->gn_w, H); TAKE(L->gn_b, H); TAKE(L->gn_ms, H);
        TAKE(L->n2_w, (int64_t)H * H); TAKE(L->n2_b, H);
        if (l == hp->depth - 1 && hp->family == 0) {
            TAKE(L->c1_w, (int64_t)H * H); TAKE(L->c1_b, H); TAKE(L->c2_w, H);
        } else {
            L->c1_w = L->c1_b = L->c2_w = NULL;
        }
        TAKE(L->att_w, H); TAKE(L->att_b, 1);
    }
    if (hp->family == 1) {   /* egnn_net.py:329-360: to_energy, to_force, to_dist, to_confidence on cat[h_r, h_l, D] */
        TAKE(w->en0_w, (int64_t)H * (2 * H + 1)); TAKE(w->en_ln_w, H); TAKE(w->en_ln_b, H); TAKE(w->en3_w, H);
        TAKE(w->fo0_w, (int64_t)H * (2 * H + 1)); TAKE(w->fo_ln_w, H); TAKE(w->fo_ln_b, H); TAKE(w->fo3_w, H);
        TAKE(w->di0_w, (int64_t)H * (2 * H + 1)); TAKE(w->di_ln_w, H); TAKE(w->di_ln_b, H); TAKE(w->di3_w, (int64_t)64 * H);
        TAKE(w->cf0_w, (int64_t)H * (2 * H + 1)); TAKE(w->cf_ln_w, H); TAKE(w->cf_ln_b, H); TAKE(w->cf3_w, H);
    } else {
        TAKE(w->en0_w, (int64_t)H * 2 * H); TAKE(w->en_ln_w, H); TAKE(w->en_ln_b, H); TAKE(w->en3_w, H);
        w->fo0_w = w->fo_ln_w = w->fo_ln_b = w->fo3_w = NULL;
        w->cf0_w = w->cf_ln_w = w->cf_ln_b = w->cf3_w = NULL;
        w->di0_w = w->di_ln_w = w->di_ln_b = w->di3_w = NULL;
    }
    TAKE(w->ir0_w, (int64_t)2 * H * H); TAKE(w->ir0_b, 2 * H);
    TAKE(w->ir2_w, (int64_t)4 * H * H); TAKE(w->ir2_b, 2 * H);
    TAKE(w->ir4_w, 2 * H); TAKE(w->ir4_b, 1);
    TAKE(w->t_W, Hi / 2); TAKE(w->t_lin, (int64_t)Hi * Hi);
    TAKE(w->trs0_w, (int64_t)Hi * (Hi + 1)); TAKE(w->trs_ln_w, Hi); TAKE(w->trs_ln_b, Hi); TAKE(w->trs4_w, Hi);
    TAKE(w->rots0_w, (int64_t)Hi * (Hi + 1)); TAKE(w->rots_ln_w, Hi); TAKE(w->rots_ln_b, Hi); TAKE(w->rots4_w, Hi);
#undef TAKE
    w->total = (int64_t)(p - blob);
}

int64_t ora_param_count(const ora_hparams *hp)
{
    net_w w;
    map_weights(hp, (const float *)0, &w);
    return w.total;
}

void ora_set_num_threads(int n)
{   /* bench.py's cpu_baseline: the all-cores and the 8-thread figures from one process */
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int ora_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* small RNG for the oracle's own (non-injected) draws: splitmix64 + Box-Muller */
typedef struct { uint64_t s; } rng_t;
static uint64_t rng_u64(rng_t *r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_uniform(rng_t *r) { return ((rng_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t *r)
{
    double u1 = rng_uniform(r), u2 = rng_uniform(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* ------------------------------------------------------------------------- */
/* dense helpers */
static inline float siluf(float x) { return x / (1.0f + expf(-x)); }   /* nn.SiLU */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Y[m][n] = bias[n] + sum_k X[m][k] * W[n][k]   (nn.Linear; W row-major [Nout,K]) */
static void linear(const float *X, int M, int K, int ldx, const float *W, int ldw, const float *bias,
                   int Nout, float *Y, int ldy, int parallel)
{
#pragma omp parallel for schedule(static) if (parallel)
    for (int mb = 0; mb < M; mb += 4) {
        const int mr = (M - mb) < 4 ? (M - mb) : 4;
        const float *x0 = X + (int64_t)mb * ldx;
        const float *x1 = X + (int64_t)(mb + (mr > 1 ? 1 : 0)) * ldx;
        const float *x2 = X + (int64_t)(mb + (mr > 2 ? 2 : 0)) * ldx;
        const float *x3 = X + (int64_t)(mb + (mr > 3 ? 3 : 0)) * ldx;
        for (int n = 0; n < Nout; n += 2) {
            const float *w0 = W + (int64_t)n * ldw;
            const float *w1 = W + (int64_t)(n + 1 < Nout ? n + 1 : n) * ldw;
            float a00 = 0, a01 = 0, a10 = 0, a11 = 0, a20 = 0, a21 = 0, a30 = 0, a31 = 0;
#pragma omp simd reduction(+ : a00, a01, a10, a11, a20, a21, a30, a31)
            for (int k = 0; k < K; ++k) {
                const float v0 = w0[k], v1 = w1[k];
                a00 += x0[k] * v0; a01 += x0[k] * v1;
                a10 += x1[k] * v0; a11 += x1[k] * v1;
                a20 += x2[k] * v0; a21 += x2[k] * v1;
                a30 += x3[k] * v0; a31 += x3[k] * v1;
            }
            const float b0 = bias ? bias[n] : 0.0f;
            const float b1 = (bias && n + 1 < Nout) ? bias[n + 1] : 0.0f;
            float r[4][2] = {{a00, a01}, {a10, a11}, {a20, a21}, {a30, a31}};
            for (int i = 0; i < mr; ++i) {
                Y[(int64_t)(mb + i) * ldy + n] = r[i][0] + b0;
                if (n + 1 < Nout) Y[(int64_t)(mb + i) * ldy + n + 1] = r[i][1] + b1;
            }
        }
    }
}

/* nn.LayerNorm(n, eps=1e-5) with affine */
static void layernorm(float *x, int n, const float *w, const float *b)
{
    double m = 0, v = 0;
    for (int i = 0; i < n; ++i) m += x[i];
    m /= n;
    for (int i = 0; i < n; ++i) { double d = x[i] - m; v += d * d; }
    v /= n;
    const float mean = (float)m, rstd = 1.0f / sqrtf((float)v + 1e-5f);
    for (int i = 0; i < n; ++i) x[i] = (x[i] - mean) * rstd * w[i] + b[i];
}

/* torch_geometric 2.6.0 GraphNorm, batch=None (third-party; call site egnn.py:74):
 * mean over nodes; out = x - mean*mean_scale; var = mean(out^2); weight*out/sqrt(var+eps)+bias */
static void ora_graphnorm(float *x, int N, int H, const float *w, const float *b, const float *ms)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < H; ++c) {
        double m = 0;
        for (int n = 0; n < N; ++n) m += x[(int64_t)n * H + c];
        const float mean = (float)(m / N);
        const float sh = mean * ms[c];
        double v = 0;
        for (int n = 0; n < N; ++n) { float o = x[(int64_t)n * H + c] - sh; v += (double)o * o; }
        const float var = (float)(v / N);
        const float den = sqrtf(var + 1e-5f);
        for (int n = 0; n < N; ++n) {
            float o = x[(int64_t)n * H + c] - sh;
            x[(int64_t)n * H + c] = w[c] * o / den + b[c];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a-3 / a-4 diffusers (float64 scalars, as numpy does them) */
double ora_r3_sigma(const ora_hparams *hp, double t)
{   /* r3_diffuser.py:20-21 */
    return (double)hp->r3_min_sigma * pow((double)hp->r3_max_sigma / (double)hp->r3_min_sigma, t);
}
double ora_r3_g(const ora_hparams *hp, double t)
{   /* r3_diffuser.py:23-24 */
    return ora_r3_sigma(hp, t) * sqrt(2.0 * (log((double)hp->r3_max_sigma) - log((double)hp->r3_min_sigma)));
}
double ora_so3_sigma(const ora_hparams *hp, double t)
{   /* so3_diffuser.py:210-217 (ValueError outside [0,1] -> NaN here) */
    if (t < 0.0 || t > 1.0) return NAN;
    return log(t * exp((double)hp->so3_max_sigma) + (1.0 - t) * exp((double)hp->so3_min_sigma));
}
double ora_so3_g(const ora_hparams *hp, double t)
{   /* so3_diffuser.py:219-227 */
    const double s = ora_so3_sigma(hp, t);
    return sqrt(2.0 * (exp((double)hp->so3_max_sigma) - exp((double)hp->so3_min_sigma)) * s / exp(s));
}
void ora_torch_reverse(double g, const float score[3], float dt, float noise_scale, const float z[3],
                       int ode, float out[3])
{   /* r3_diffuser.py:40-55 / so3_diffuser.py:344-369.  torch semantics: the float64 scalars g^2 and g
     * enter float32 tensor arithmetic rounded to float32; every tensor op rounds to float32. */
    const float g2 = (float)(g * g);
    const float hg2 = (float)(0.5 * (g * g));   /* ode: 0.5*(g**2) is a float64 product first */
    const float gf = (float)g;
    const float sdt = sqrtf(dt);
    for (int i = 0; i < 3; ++i) {
        if (!ode) {
            const float zz = noise_scale * z[i];
            const float a = (g2 * score[i]) * dt;
            const float b = (gf * sdt) * zz;
            out[i] = a + b;
        } else {
            out[i] = (hg2 * score[i]) * dt;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a-14 rotation conversions (geometry.py, pytorch3d-derived) */
void ora_axis_angle_to_quaternion(const float aa[3], float q[4])
{   /* geometry.py:154-183 */
    const float ang = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    const float half = 0.5f * ang;
    float s;
    if (fabsf(ang) < 1e-6f) s = 0.5f - (ang * ang) / 48.0f;
    else s = sinf(half) / ang;
    q[0] = cosf(half);
    q[1] = aa[0] * s; q[2] = aa[1] * s; q[3] = aa[2] * s;
}
void ora_quaternion_to_matrix(const float q[4], float R[9])
{   /* geometry.py:18-45 */
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}
void ora_axis_angle_to_matrix(const float aa[3], float R[9])
{   /* geometry.py:186-198 */
    float q[4];
    ora_axis_angle_to_quaternion(aa, q);
    ora_quaternion_to_matrix(q, R);
}
void ora_matrix_to_quaternion(const float R[9], float q[4])
{   /* geometry.py:64-123: best-conditioned of four candidates, denominator floored at 0.1 */
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6],
                m21 = R[7], m22 = R[8];
    float qa[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22,
                   1.0f - m00 - m11 + m22};
    for (int c = 0; c < 4; ++c) qa[c] = qa[c] > 0 ? sqrtf(qa[c]) : 0.0f;
    const float cand[4][4] = {
        {qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
        {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
        {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
        {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    int best = 0;
    for (int c = 1; c < 4; ++c) if (qa[c] > qa[best]) best = c;   /* argmax: first maximum */
    const float den = 2.0f * (qa[best] > 0.1f ? qa[best] : 0.1f);
    for (int c = 0; c < 4; ++c) q[c] = cand[best][c] / den;
}
void ora_quaternion_to_axis_angle(const float q[4], float aa[3])
{   /* geometry.py:126-151: angle = 2*atan2(|v|, w) in [0, 2pi], not wrapped */
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float half = atan2f(n, q[0]);
    const float ang = 2.0f * half;
    float s;
    if (fabsf(ang) < 1e-6f) s = 0.5f - (ang * ang) / 48.0f;
    else s = sinf(half) / ang;
    aa[0] = q[1] / s; aa[1] = q[2] / s; aa[2] = q[3] / s;
}
void ora_matrix_to_axis_angle(const float R[9], float aa[3])
{   /* geometry.py:48-61 */
    float q[4];
    ora_matrix_to_quaternion(R, q);
    ora_quaternion_to_axis_angle(q, aa);
}
void ora_rot_compose(const float r1[3], const float r2[3], float out[3])
{   /* inference_base.py:311-316: R = R(r2) @ R(r1) */
    float R1[9], R2[9], Rm[9];
    ora_axis_angle_to_matrix(r1, R1);
    ora_axis_angle_to_matrix(r2, R2);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            float s = 0;
            for (int j = 0; j < 3; ++j) s += R2[i * 3 + j] * R1[j * 3 + k];
            Rm[i * 3 + k] = s;
        }
    ora_matrix_to_axis_angle(Rm, out);
}
static void ca_mean(const float *x, int n, float c[3])
{   /* torch.mean(x[..., 1, :], dim=0) */
    double s[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) s[d] += x[i * 9 + 3 + d];
    for (int d = 0; d < 3; ++d) c[d] = (float)(s[d] / n);
}
static void atom_mean(const float *x, int n, float c[3])
{   /* torch.mean(x, dim=(0, 1)): all n x 3 backbone atoms (src/inference.py:224-225,:245; DFMDock.py:247) */
    double s[3] = {0, 0, 0};
    for (int a = 0; a < n * 3; ++a)
        for (int d = 0; d < 3; ++d) s[d] += x[a * 3 + d];
    for (int d = 0; d < 3; ++d) c[d] = (float)(s[d] / (n * 3));
}
static void rotate_about(float *x, int n, const float c[3], const float Rm[9])
{   /* (x - c) @ R.T + c on every backbone atom */
    for (int a = 0; a < n * 3; ++a) {
        float v[3] = {x[a * 3] - c[0], x[a * 3 + 1] - c[1], x[a * 3 + 2] - c[2]};
        for (int i = 0; i < 3; ++i) {
            float s = 0;
            for (int j = 0; j < 3; ++j) s += v[j] * Rm[i * 3 + j];
            x[a * 3 + i] = s + c[i];
        }
    }
}
void ora_modify_coords(float *x, int n, const float rot[3], const float tr[3])
{   /* inference_base.py:342-352 */
    float c[3], Rm[9];
    ca_mean(x, n, c);
    ora_axis_angle_to_matrix(rot, Rm);
    rotate_about(x, n, c, Rm);
    for (int a = 0; a < n * 3; ++a)
        for (int d = 0; d < 3; ++d) x[a * 3 + d] += tr[d];
}
void ora_modify_coords_all_atom(float *x, int n, const float rot[3], const float tr[3])
{   /* second family: src/inference.py:244-254 == DFMDock.py:246-252 (rotation about the all-atom centroid) */
    float c[3], Rm[9];
    atom_mean(x, n, c);
    ora_axis_angle_to_matrix(rot, Rm);
    rotate_about(x, n, c, Rm);
    for (int a = 0; a < n * 3; ++a)
        for (int d = 0; d < 3; ++d) x[a * 3 + d] += tr[d];
}
void ora_clash_force(const float *rec, int R, const float *lig, int L, float out[3])
{   /* inference_base.py:366-384.  rep = sum_{d<4} |4-d|^1.5 / (1.5*d*0.5); E = -5*rep;
     * force = dE/dlig, mean over the 3L ligand atoms.  dE/dd = -5 * d/dd[(4-d)^1.5/(0.75 d)]. */
    double acc[3] = {0, 0, 0};
    const int nr = R * 3, nl = L * 3;
    for (int j = 0; j < nl; ++j) {
        double g[3] = {0, 0, 0};
        for (int i = 0; i < nr; ++i) {
            const double dx = (double)rec[i * 3] - lig[j * 3], dy = (double)rec[i * 3 + 1] - lig[j * 3 + 1],
                         dz = (double)rec[i * 3 + 2] - lig[j * 3 + 2];
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (d < 4.0 && d > 0.0) {
                const double u = 4.0 - d;
                /* f(d) = u^1.5/(0.75 d);  f'(d) = (-1.5 u^0.5 * d - u^1.5) / (0.75 d^2) */
                const double fp = (-1.5 * sqrt(u) * d - u * sqrt(u)) / (0.75 * d * d);
                const double dE = -5.0 * fp;
                /* dd/dlig = -(rec-lig)/d */
                g[0] += dE * (-dx / d); g[1] += dE * (-dy / d); g[2] += dE * (-dz / d);
            }
        }
        acc[0] += g[0]; acc[1] += g[1]; acc[2] += g[2];
    }
    for (int d = 0; d < 3; ++d) out[d] = (float)(acc[d] / nl);
}

/* ------------------------------------------------------------------------- */
/* a-6: trRosetta 6-D coordinates (coords6d.py:10-103), fp32 op by op as torch evaluates them */
typedef struct { float x, y, z; } v3;
static inline v3 v_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v_cross(v3 a, v3 b)
{
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v_norm(v3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
static inline v3 v_divs(v3 a, float s) { v3 r = {a.x / s, a.y / s, a.z / s}; return r; }
static inline v3 ld3(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }

static v3 virtual_cb(const float *res /* [9] N,CA,C */)
{   /* coords6d.py:71-75 */
    v3 N = ld3(res), Ca = ld3(res + 3), C = ld3(res + 6);
    v3 b = v_sub(Ca, N), c = v_sub(C, Ca), a = v_cross(b, c);
    v3 r;
    r.x = ((-0.58273431f * a.x + 0.56802827f * b.x) - 0.54067466f * c.x) + Ca.x;
    r.y = ((-0.58273431f * a.y + 0.56802827f * b.y) - 0.54067466f * c.y) + Ca.y;
    r.z = ((-0.58273431f * a.z + 0.56802827f * b.z) - 0.54067466f * c.z) + Ca.z;
    return r;
}
static const float RAD2DEG_PI = 3.14159265358979323846f;
static float dihedral_deg(v3 a, v3 b, v3 c, v3 d)
{   /* coords6d.py:25-43 */
    v3 b1 = v_sub(a, b), b2 = v_sub(b, c), b3 = v_sub(c, d);
    v3 n1 = v_cross(b1, b2); n1 = v_divs(n1, v_norm(n1));
    v3 n2 = v_cross(b2, b3); n2 = v_divs(n2, v_norm(n2));
    v3 m1 = v_cross(n1, v_divs(b2, v_norm(b2)));
    return atan2f(v_dot(m1, n2), v_dot(n1, n2)) * 180.0f / RAD2DEG_PI;
}
static float planar_deg(v3 a, v3 b, v3 c)
{   /* coords6d.py:46-58 */
    v3 v1 = v_sub(a, b), v2 = v_sub(c, b);
    return acosf(v_dot(v1, v2) / (v_norm(v1) * v_norm(v2))) * 180.0f / RAD2DEG_PI;
}
static void pair6d(const float *pi, const float *pj, v3 cbi, v3 cbj, float *dist, float *omega, float *theta,
                   float *phi)
{   /* coords6d.py:77-100 */
    v3 Ni = ld3(pi), Cai = ld3(pi + 3), Caj = ld3(pj + 3);
    *dist = v_norm(v_sub(Cai, Caj));
    *omega = dihedral_deg(Cai, cbi, cbj, Caj);
    *theta = dihedral_deg(Ni, Cai, cbi, cbj);
    *phi = planar_deg(Cai, cbi, cbj);
}
void ora_coords6d_full(const float *pos, int N, float *dist, float *omega, float *theta, float *phi)
{
    v3 *cb = (v3 *)malloc(sizeof(v3) * N);
    for (int i = 0; i < N; ++i) cb[i] = virtual_cb(pos + i * 9);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            const int64_t o = (int64_t)i * N + j;
            pair6d(pos + i * 9, pos + j * 9, cb[i], cb[j], dist + o, omega + o, theta + o, phi + o);
        }
    free(cb);
}

/* a-7: get_bins / get_spatial_matrix (score_net_mlsb.py:30-70).  bin = #(x > boundary), NaN -> 0.
 * Angle boundaries are torch.linspace(-180,180,23) as float32 (tests/golden/scalar_kats: angle_bounds). */
static const float ANGLE_BOUNDS[23] = {
    -180.0f, -163.63636779785156f, -147.27273559570312f, -130.90908813476562f, -114.54545593261719f,
    -98.18182373046875f, -81.81818389892578f, -65.45454406738281f, -49.090911865234375f,
    -32.72727584838867f, -16.36363983154297f, 3.814697265625e-06f, 16.36363983154297f,
    32.72727584838867f, 49.090911865234375f, 65.45454406738281f, 81.81818389892578f, 98.18182373046875f,
    114.54545593261719f, 130.90908813476562f, 147.27273559570312f, 163.63636779785156f, 180.0f};
static int bin_dist(float d)
{   /* linspace(3.25, 50.75, 39): exact multiples of 1.25 */
    int b = 0;
    for (int i = 0; i < 39; ++i) b += d > (3.25f + 1.25f * (float)i);
    return b;
}
static int bin_angle(float a)
{
    int b = 0;
    for (int i = 0; i < 23; ++i) b += a > ANGLE_BOUNDS[i];
    return b;
}
static int bin_phi(float a)
{   /* linspace(0, 180, 11): exact multiples of 18 */
    int b = 0;
    for (int i = 0; i < 11; ++i) b += a > (18.0f * (float)i);
    return b;
}
static void pair_bins(const ora_hparams *hp, int i, int j, const float *pi, const float *pj, v3 cbi, v3 cbj,
                      int8_t out[4])
{
    float d, om, th, ph;
    pair6d(pi, pj, cbi, cbj, &d, &om, &th, &ph);
    const int masked = !(d < hp->mask_dist) || i == j;     /* mask = dist < 22.0; fill_diagonal_(0) */
    out[0] = (int8_t)bin_dist(d);
    out[1] = masked ? 0 : (int8_t)bin_angle(om);
    out[2] = masked ? 0 : (int8_t)bin_angle(th);
    out[3] = masked ? 0 : (int8_t)bin_phi(ph);
}
void ora_bins_full(const ora_hparams *hp, const float *pos, int N, int8_t *bins)
{
    v3 *cb = (v3 *)malloc(sizeof(v3) * N);
    for (int i = 0; i < N; ++i) cb[i] = virtual_cb(pos + i * 9);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            pair_bins(hp, i, j, pos + i * 9, pos + j * 9, cb[i], cb[j], bins + ((int64_t)i * N + j) * 4);
    free(cb);
}
/* a-8: relpos (inference_base.py:255-292) with res_id = 0..N-1, asym = 0 (rec) / 1 (lig) */
static int relpos_idx(int i, int j, int R)
{
    const int same = (i < R) == (j < R);
    int off = i - j + 32;
    off = off < 0 ? 0 : (off > 64 ? 64 : off);
    return same ? off : 65;
}
void ora_relpos_full(int R, int L, int8_t *rel)
{
    const int N = R + L;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) rel[(int64_t)i * N + j] = (int8_t)relpos_idx(i, j, R);
}

/* a-10: get_knn_and_sample (score_net_mlsb.py:85-135).  kNN: K smallest distances, ascending, lowest
 * index first on exact ties (slot 0 = self).  Sampling: n_sample draws without replacement with
 * p ~ 1/max(d,1e-10)^3 over the non-kNN nodes = exponential race (what torch.multinomial does:
 * top-k of p / Exp(1)); the oracle's stream is its own, so only the distribution matches. */
void ora_knn_sample(const ora_hparams *hp, const float *ca, int N, uint64_t seed, int32_t *edges, int *K_out)
{
    int knn = hp->knn, ns = hp->n_sample;
    if (N < knn) { knn = N; ns = 0; }
    if (N < knn + ns) ns = N - knn;
    const int K = knn + ns;
    *K_out = K;
#pragma omp parallel
    {
        float *d = (float *)malloc(sizeof(float) * N);
        double *key = (double *)malloc(sizeof(double) * N);
        char *used = (char *)malloc(N);
#pragma omp for schedule(static)
        for (int i = 0; i < N; ++i) {
            rng_t rng = {seed * 0x9E3779B97F4A7C15ull + (uint64_t)i * 0xD1B54A32D192ED03ull + 1};
            for (int j = 0; j < N; ++j) {
                const float dx = ca[i * 3] - ca[j * 3], dy = ca[i * 3 + 1] - ca[j * 3 + 1],
                            dz = ca[i * 3 + 2] - ca[j * 3 + 2];
                d[j] = sqrtf((dx * dx + dy * dy) + dz * dz);
            }
            memset(used, 0, N);
            for (int s = 0; s < knn; ++s) {
                int best = -1;
                for (int j = 0; j < N; ++j)
                    if (!used[j] && (best < 0 || d[j] < d[best])) best = j;
                used[best] = 1;
                edges[(int64_t)i * K + s] = best;
            }
            for (int j = 0; j < N; ++j) {
                if (used[j]) { key[j] = INFINITY; continue; }
                const double dd = d[j] < 1e-10f ? 1e-10 : (double)d[j];
                key[j] = -log(rng_uniform(&rng)) * (dd * dd * dd);   /* Exp(1)/w, w = d^-3 */
            }
            for (int s = 0; s < ns; ++s) {
                int best = -1;
                for (int j = 0; j < N; ++j)
                    if (!used[j] && (best < 0 || key[j] < key[best])) best = j;
                used[best] = 1;
                edges[(int64_t)i * K + knn + s] = best;
            }
        }
        free(d); free(key); free(used);
    }
}

/* ------------------------------------------------------------------------- */
/* a-5: Score_Net.forward(predict=True) (score_net_mlsb.py:343-425) */
/* positional_embed_dim = 67 (configs/model/DFMDock.yaml:5): the 67th position channel is the homomer flag of the complex
 * (`is_homomer`, datasets/docking_dataset.py:129), the same value on every residue pair: ora_hparams.homomer.  The reference
 * tree has NO producer that appends this channel (utils/crop.get_position_matrix returns 66): the layout [relpos66 | flag] is this
 * build's assumption, shared by the engine and the fwd2_sym goldens */

int ora_score(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x, const float *lig_x,
              const float *rec_pos, const float *lig_pos, float t, const int32_t *edges_in, uint64_t seed,
              int want_energy, ora_score_out *out, ora_debug *dbg)
{
    const int H = hp->node_dim, He = hp->edge_dim, Hi = hp->inner_dim, N = R + L;
    const int Kin1 = 2 * H + 1 + He;
    net_w w;
    map_weights(hp, blob, &w);

    /* :353-359 centre on the ligand CA centroid, concatenate.  Family 1: DFMDock.move_to_lig_center
     * (DFMDock.py:254-257) subtracts the mean over all L x 3 backbone atoms before the net is called. */
    float center[3];
    if (hp->family == 1) {
        for (int d = 0; d < 3; ++d) {
            double sacc = 0;
            for (int i = 0; i < L * 3; ++i) sacc += lig_pos[i * 3 + d];
            center[d] = (float)(sacc / (L * 3));
        }
    } else {
        ca_mean(lig_pos, L, center);
    }
    float *pos = (float *)malloc(sizeof(float) * N * 9);
    for (int i = 0; i < R * 3; ++i)
        for (int d = 0; d < 3; ++d) pos[i * 3 + d] = rec_pos[i * 3 + d] - center[d];
    for (int i = 0; i < L * 3; ++i)
        for (int d = 0; d < 3; ++d) pos[(R * 3 + i) * 3 + d] = lig_pos[i * 3 + d] - center[d];
    float *ca = (float *)malloc(sizeof(float) * N * 3);
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < 3; ++d) ca[i * 3 + d] = pos[i * 9 + 3 + d];

    /* :365-366 node embedding */
    float *x = (float *)malloc(sizeof(float) * (int64_t)N * hp->lm_embed_dim);
    memcpy(x, rec_x, sizeof(float) * (int64_t)R * hp->lm_embed_dim);
    memcpy(x + (int64_t)R * hp->lm_embed_dim, lig_x, sizeof(float) * (int64_t)L * hp->lm_embed_dim);
    float *h = (float *)malloc(sizeof(float) * (int64_t)N * H);
    linear(x, N, hp->lm_embed_dim, hp->lm_embed_dim, w.single_embed, hp->lm_embed_dim, NULL, H, h, H, 1);
    free(x);

    /* :373 graph */
    int K;
    int32_t *edges;
    {
        int knn = hp->knn, ns = hp->n_sample;
        if (N < knn) { knn = N; ns = 0; }
        if (N < knn + ns) ns = N - knn;
        K = knn + ns;
        edges = (int32_t *)malloc(sizeof(int32_t) * (int64_t)N * K);
        if (edges_in) memcpy(edges, edges_in, sizeof(int32_t) * (int64_t)N * K);
        else { int k2; ora_knn_sample(hp, ca, N, seed, edges, &k2); }
    }
    const int64_t E = (int64_t)N * K;

    /* :369-370 + :155: edge_attr of the selected pairs only.  one_hot @ W^T == sum of the selected
     * weight columns; spatial (dist, omega, theta, phi) first, then + positional. */
    v3 *cb = (v3 *)malloc(sizeof(v3) * N);
    for (int i = 0; i < N; ++i) cb[i] = virtual_cb(pos + i * 9);
    float *eattr = (float *)malloc(sizeof(float) * E * He);
    float *radial = (float *)malloc(sizeof(float) * E);
    float *cdiff = (float *)malloc(sizeof(float) * E * 3);
    const int Sd = hp->spatial_embed_dim, Pd = hp->positional_embed_dim;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i)
        for (int s = 0; s < K; ++s) {
            const int64_t e = (int64_t)i * K + s;
            const int j = edges[e];
            int8_t b[4];
            pair_bins(hp, i, j, pos + i * 9, pos + j * 9, cb[i], cb[j], b);
            if (dbg && dbg->bins_in) memcpy(b, dbg->bins_in + e * 4, 4);
            const int rp = relpos_idx(i, j, R);
            if (dbg && dbg->bins) memcpy(dbg->bins + e * 4, b, 4);
            if (dbg && dbg->relpos) dbg->relpos[e] = (int8_t)rp;
            for (int c = 0; c < He; ++c) {
                const float *ws = w.spatial_embed + (int64_t)c * Sd;
                float sp = ((ws[b[0]] + ws[40 + b[1]]) + ws[64 + b[2]]) + ws[88 + b[3]];
                float pe = w.positional_embed[(int64_t)c * Pd + rp];
                if (Pd == 67 && hp->homomer) pe += w.positional_embed[(int64_t)c * Pd + 66];
                eattr[e * He + c] = sp + pe;
            }
            /* egnn.py:139-148 coord2radial, normalize=True */
            const float dx = ca[i * 3] - ca[j * 3], dy = ca[i * 3 + 1] - ca[j * 3 + 1],
                        dz = ca[i * 3 + 2] - ca[j * 3 + 2];
            const float r2 = (dx * dx + dy * dy) + dz * dz;
            radial[e] = r2;
            const float nrm = sqrtf(r2 + 1e-8f) + 1.0f;
            cdiff[e * 3] = dx / nrm; cdiff[e * 3 + 1] = dy / nrm; cdiff[e * 3 + 2] = dz / nrm;
        }
    free(cb);
    if (dbg && dbg->edges) memcpy(dbg->edges, edges, sizeof(int32_t) * E);

    /* :380 EGNN: depth x E_GCL.forward (egnn.py:150-159) */
    float *coord = (float *)malloc(sizeof(float) * N * 3);
    memcpy(coord, ca, sizeof(float) * N * 3);
    float *agg = (float *)malloc(sizeof(float) * (int64_t)N * H);
    float *cat = (float *)malloc(sizeof(float) * (int64_t)N * 2 * H);
    float *u = (float *)malloc(sizeof(float) * (int64_t)N * H);
    float *o2 = (float *)malloc(sizeof(float) * (int64_t)N * H);
    float *cagg = (float *)calloc((size_t)N * 3, sizeof(float));
    for (int l = 0; l < hp->depth; ++l) {
        const layer_w *Lw = &w.layer[l];
        const int last = (l == hp->depth - 1) && hp->family == 0;   /* EGNN_Net: update_coords=False everywhere */
#pragma omp parallel
        {
            float *in = (float *)malloc(sizeof(float) * (int64_t)K * Kin1);
            float *m1 = (float *)malloc(sizeof(float) * (int64_t)K * H);
            float *m2 = (float *)malloc(sizeof(float) * (int64_t)K * H);
            float *c1 = (float *)malloc(sizeof(float) * (int64_t)K * H);
#pragma omp for schedule(dynamic, 4)
            for (int i = 0; i < N; ++i) {
                /* edge_model (egnn.py:95-104): cat[h_i, h_j, radial, edge_attr] -> Linear,SiLU,Linear,SiLU */
                for (int s = 0; s < K; ++s) {
                    const int64_t e = (int64_t)i * K + s;
                    const int j = edges[e];
                    float *row = in + (int64_t)s * Kin1;
                    memcpy(row, h + (int64_t)i * H, sizeof(float) * H);
                    memcpy(row + H, h + (int64_t)j * H, sizeof(float) * H);
                    row[2 * H] = radial[e];
                    memcpy(row + 2 * H + 1, eattr + e * He, sizeof(float) * He);
                }
                linear(in, K, Kin1, Kin1, Lw->e1_w, Kin1, Lw->e1_b, H, m1, H, 0);
                for (int q = 0; q < K * H; ++q) m1[q] = siluf(m1[q]);
                linear(m1, K, H, H, Lw->e2_w, H, Lw->e2_b, H, m2, H, 0);
                for (int q = 0; q < K * H; ++q) m2[q] = siluf(m2[q]);
                /* attention gate */
                for (int s = 0; s < K; ++s) {
                    float a = 0;
                    for (int c = 0; c < H; ++c) a += m2[s * H + c] * Lw->att_w[c];
                    a = sigmoidf_(a + Lw->att_b[0]);
                    for (int c = 0; c < H; ++c) m2[s * H + c] *= a;
                }
                if (last) {
                    /* coord_model (egnn.py:118-137): Linear,SiLU,Linear(no bias) -> clamp(+-2) -> mean */
                    linear(m2, K, H, H, Lw->c1_w, H, Lw->c1_b, H, c1, H, 0);
                    float acc[3] = {0, 0, 0};
                    for (int s = 0; s < K; ++s) {
                        float cw = 0;
                        for (int c = 0; c < H; ++c) cw += siluf(c1[s * H + c]) * Lw->c2_w[c];
                        cw = cw < -2.0f ? -2.0f : (cw > 2.0f ? 2.0f : cw);
                        const int64_t e = (int64_t)i * K + s;
                        for (int d = 0; d < 3; ++d) acc[d] += cdiff[e * 3 + d] * cw;
                    }
                    const float cnt = (float)(K > 1 ? K : 1);
                    for (int d = 0; d < 3; ++d) cagg[i * 3 + d] = acc[d] / cnt;
                }
                /* unsorted_segment_sum over the node's K edges */
                for (int c = 0; c < H; ++c) {
                    float sacc = 0;
                    for (int s = 0; s < K; ++s) sacc += m2[s * H + c];
                    agg[(int64_t)i * H + c] = sacc;
                }
            }
            free(in); free(m1); free(m2); free(c1);
        }
        if (last)
            for (int i = R; i < N; ++i)            /* lig_mask: only ligand nodes move */
                for (int d = 0; d < 3; ++d) coord[i * 3 + d] += cagg[i * 3 + d];
        /* node_model (egnn.py:106-116) */
        for (int i = 0; i < N; ++i) {
            memcpy(cat + (int64_t)i * 2 * H, h + (int64_t)i * H, sizeof(float) * H);
            memcpy(cat + (int64_t)i * 2 * H + H, agg + (int64_t)i * H, sizeof(float) * H);
        }
        linear(cat, N, 2 * H, 2 * H, Lw->n1_w, 2 * H, Lw->n1_b, H, u, H, 1);
        ora_graphnorm(u, N, H, Lw->gn_w, Lw->gn_b, Lw->gn_ms);
        for (int64_t q = 0; q < (int64_t)N * H; ++q) u[q] = siluf(u[q]);
        linear(u, N, H, H, Lw->n2_w, H, Lw->n2_b, H, o2, H, 1);
        for (int64_t q = 0; q < (int64_t)N * H; ++q) h[q] = h[q] + o2[q];
        if (dbg && dbg->h_layers) memcpy(dbg->h_layers + (int64_t)l * N * H, h, sizeof(float) * (int64_t)N * H);
    }
    if (dbg && dbg->pos_out) memcpy(dbg->pos_out, coord, sizeof(float) * N * 3);

    /* :383 ires head (unused by the sampler) */
    if (dbg && dbg->ires) {
        float *a1 = (float *)malloc(sizeof(float) * (int64_t)N * 2 * H);
        float *a2 = (float *)malloc(sizeof(float) * (int64_t)N * 2 * H);
        linear(h, N, H, H, w.ir0_w, H, w.ir0_b, 2 * H, a1, 2 * H, 1);
        for (int64_t q = 0; q < (int64_t)N * 2 * H; ++q) a1[q] = siluf(a1[q]);
        linear(a1, N, 2 * H, 2 * H, w.ir2_w, 2 * H, w.ir2_b, 2 * H, a2, 2 * H, 1);
        for (int64_t q = 0; q < (int64_t)N * 2 * H; ++q) a2[q] = siluf(a2[q]);
        linear(a2, N, 2 * H, 2 * H, w.ir4_w, 2 * H, w.ir4_b, 1, dbg->ires, 1, 1);
        free(a1); free(a2);
    }

    int64_t clashes = 0;
    double esum = 0, msum = 0;
    out->confidence = 0.f;
    float *fpair = NULL;   /* family 1: f [L,3] from the pair-force head */
    if (hp->family == 1) {
        /* egnn_net.py:430-470.  Linear(cat[h_r, h_l, D]) = W[:, :H] h_r + W[:, H:2H] h_l + W[:, 2H] D (no bias)
         * -> LayerNorm -> SiLU -> Linear(H -> 1).  energy: masked (D < cut_off) mean (clamp(min=1)) or sum;
         * confidence: mean over all pairs; force f_l = agg_r unit_vec(r,l) * to_force(.) ; clashes D <= 3 */
        const int Kp = 2 * H + 1;
        const float *W0[3] = {w.fo0_w, w.en0_w, w.cf0_w};
        const float *LW[3] = {w.fo_ln_w, w.en_ln_w, w.cf_ln_w}, *LB[3] = {w.fo_ln_b, w.en_ln_b, w.cf_ln_b};
        const float *W3[3] = {w.fo3_w, w.en3_w, w.cf3_w};
        const int nh = want_energy ? 3 : 1;
        float *Ar[3], *Bl[3];
        for (int q = 0; q < nh; ++q) {
            Ar[q] = (float *)malloc(sizeof(float) * (int64_t)R * H);
            Bl[q] = (float *)malloc(sizeof(float) * (int64_t)L * H);
            linear(h, R, H, H, W0[q], Kp, NULL, H, Ar[q], H, 1);
            linear(h + (int64_t)R * H, L, H, H, W0[q] + H, Kp, NULL, H, Bl[q], H, 1);
        }
        fpair = (float *)calloc((size_t)L * 3, sizeof(float));
        double csum = 0;
        double *facc = (double *)calloc((size_t)L * 3, sizeof(double));
#pragma omp parallel
        {
            float *v = (float *)malloc(sizeof(float) * H);
            double *fl = (double *)calloc((size_t)L * 3, sizeof(double));
#pragma omp for schedule(static) reduction(+ : esum, msum, csum, clashes)
            for (int r = 0; r < R; ++r) {
                for (int q = 0; q < L; ++q) {
                    const int j = R + q;
                    const float dx = ca[r * 3] - ca[j * 3], dy = ca[r * 3 + 1] - ca[j * 3 + 1],
                                dz = ca[r * 3 + 2] - ca[j * 3 + 2];
                    const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
                    if (D <= 3.0f) clashes += 1;
                    float sval[3] = {0, 0, 0};
                    for (int hd = 0; hd < nh; ++hd) {
                        for (int c = 0; c < H; ++c)
                            v[c] = (Ar[hd][(int64_t)r * H + c] + Bl[hd][(int64_t)q * H + c]) + W0[hd][(int64_t)c * Kp + 2 * H] * D;
                        layernorm(v, H, LW[hd], LB[hd]);
                        float o = 0;
                        for (int c = 0; c < H; ++c) o += siluf(v[c]) * W3[hd][c];
                        sval[hd] = o;
                    }
                    const float nrm = D > 1e-12f ? D : 1e-12f;     /* F.normalize(vec, dim=-1), eps 1e-12 */
                    fl[q * 3] += (double)(dx / nrm * sval[0]);
                    fl[q * 3 + 1] += (double)(dy / nrm * sval[0]);
                    fl[q * 3 + 2] += (double)(dz / nrm * sval[0]);
                    if (want_energy) {
                        if (D < hp->cut_off) { msum += 1.0; esum += sval[1]; }
                        csum += sval[2];
                    }
                }
            }
#pragma omp critical
            for (int q = 0; q < L * 3; ++q) facc[q] += fl[q];
            free(v); free(fl);
        }
        for (int q = 0; q < L * 3; ++q) fpair[q] = (float)(hp->agg_mean ? facc[q] / R : facc[q]);
        free(facc);
        if (dbg && dbg->dist) {   /* egnn_net.py:447 dist_logits = to_dist(interaction): [R, L, 64] (a training-loss input, :196-215) */
            float *Ad = (float *)malloc(sizeof(float) * (int64_t)R * H), *Bd = (float *)malloc(sizeof(float) * (int64_t)L * H);
            linear(h, R, H, H, w.di0_w, Kp, NULL, H, Ad, H, 1);
            linear(h + (int64_t)R * H, L, H, H, w.di0_w + H, Kp, NULL, H, Bd, H, 1);
#pragma omp parallel
            {
                float *v = (float *)malloc(sizeof(float) * H);
#pragma omp for schedule(static)
                for (int r = 0; r < R; ++r)
                    for (int q = 0; q < L; ++q) {
                        const int j = R + q;
                        const float dx = ca[r * 3] - ca[j * 3], dy = ca[r * 3 + 1] - ca[j * 3 + 1], dz = ca[r * 3 + 2] - ca[j * 3 + 2];
                        const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
                        for (int c = 0; c < H; ++c)
                            v[c] = (Ad[(int64_t)r * H + c] + Bd[(int64_t)q * H + c]) + w.di0_w[(int64_t)c * Kp + 2 * H] * D;
                        layernorm(v, H, w.di_ln_w, w.di_ln_b);
                        for (int c = 0; c < H; ++c) v[c] = siluf(v[c]);
                        for (int o = 0; o < 64; ++o) {
                            float acc = 0;
                            for (int c = 0; c < H; ++c) acc += v[c] * w.di3_w[(int64_t)o * H + c];
                            dbg->dist[((int64_t)r * L + q) * 64 + o] = acc;
                        }
                    }
                free(v);
            }
            free(Ad); free(Bd);
        }
        for (int q = 0; q < nh; ++q) { free(Ar[q]); free(Bl[q]); }
        if (want_energy) {
            out->energy = hp->agg_mean ? (float)((float)esum / (msum < 1.0 ? 1.0f : (float)msum)) : (float)esum;
            out->confidence = (float)(csum / ((double)R * L));
        } else {
            out->energy = 0.f;
        }
        out->num_clashes = clashes;
    } else
    /* :362, :386-390 energy over receptor x ligand pairs with CA distance < cut_off; :72 clashes */
    {
        /* Linear(cat[h_r,h_l]) = W[:, :H] h_r + W[:, H:] h_l (no bias) */
        float *Ar = (float *)malloc(sizeof(float) * (int64_t)R * H);
        float *Bl = (float *)malloc(sizeof(float) * (int64_t)L * H);
        if (want_energy) {
            linear(h, R, H, H, w.en0_w, 2 * H, NULL, H, Ar, H, 1);
            linear(h + (int64_t)R * H, L, H, H, w.en0_w + H, 2 * H, NULL, H, Bl, H, 1);
        }
#pragma omp parallel for schedule(static) reduction(+ : esum, msum, clashes)
        for (int r = 0; r < R; ++r) {
            float *v = (float *)malloc(sizeof(float) * H);
            for (int q = 0; q < L; ++q) {
                const int j = R + q;
                const float dx = ca[r * 3] - ca[j * 3], dy = ca[r * 3 + 1] - ca[j * 3 + 1],
                            dz = ca[r * 3 + 2] - ca[j * 3 + 2];
                const float D = sqrtf((dx * dx + dy * dy) + dz * dz);
                if (D <= 3.0f) clashes += 1;
                if (D < hp->cut_off) {
                    msum += 1.0;
                    if (want_energy) {
                        for (int c = 0; c < H; ++c) v[c] = Ar[(int64_t)r * H + c] + Bl[(int64_t)q * H + c];
                        layernorm(v, H, w.en_ln_w, w.en_ln_b);
                        float en = 0;
                        for (int c = 0; c < H; ++c) en += siluf(v[c]) * w.en3_w[c];
                        esum += en;
                    }
                }
            }
            free(v);
        }
        free(Ar); free(Bl);
    }
    if (hp->family == 0) {
        out->energy = (float)((float)esum / ((float)msum + 1e-6f));
        out->num_clashes = clashes;
    }

    /* :396-404 force, translation and torque pooling */
    double trp[3] = {0, 0, 0}, rtp[3] = {0, 0, 0};
    for (int q = 0; q < L; ++q) {
        const int i = R + q;
        const float r[3] = {ca[i * 3], ca[i * 3 + 1], ca[i * 3 + 2]};
        const float f[3] = {fpair ? fpair[q * 3] : coord[i * 3] - r[0], fpair ? fpair[q * 3 + 1] : coord[i * 3 + 1] - r[1],
                            fpair ? fpair[q * 3 + 2] : coord[i * 3 + 2] - r[2]};
        if (dbg && dbg->f) { dbg->f[q * 3] = f[0]; dbg->f[q * 3 + 1] = f[1]; dbg->f[q * 3 + 2] = f[2]; }
        for (int d = 0; d < 3; ++d) trp[d] += f[d];
        rtp[0] += r[1] * f[2] - r[2] * f[1];
        rtp[1] += r[2] * f[0] - r[0] * f[2];
        rtp[2] += r[0] * f[1] - r[1] * f[0];
    }
    float tr_pred[3], rot_pred[3];
    const double pool = (hp->family == 1 && !hp->agg_mean) ? 1.0 : (double)L;     /* egnn_net.py:459-470 */
    for (int d = 0; d < 3; ++d) { tr_pred[d] = (float)(trp[d] / pool); rot_pred[d] = (float)(rtp[d] / pool); }
    free(fpair);

    /* :407 t_embed: GaussianFourierProjection (:162-172) -> Linear(no bias) -> Sigmoid */
    float *four = (float *)malloc(sizeof(float) * Hi), *temb = (float *)malloc(sizeof(float) * Hi);
    for (int c = 0; c < Hi / 2; ++c) {
        const float xp = ((t * w.t_W[c]) * 2.0f) * 3.14159265358979323846f;
        four[c] = sinf(xp);
        four[Hi / 2 + c] = cosf(xp);
    }
    linear(four, 1, Hi, Hi, w.t_lin, Hi, NULL, Hi, temb, Hi, 0);
    for (int c = 0; c < Hi; ++c) temb[c] = sigmoidf_(temb[c]);
    /* :408-411 scale MLPs: Linear(129->128,no bias), LayerNorm, SiLU, Linear(128->1,no bias), Softplus */
    float *sin_ = (float *)malloc(sizeof(float) * (Hi + 1)), *hid = (float *)malloc(sizeof(float) * Hi);
    for (int which = 0; which < 2; ++which) {
        const float *pred = which ? rot_pred : tr_pred;
        const float *w0 = which ? w.rots0_w : w.trs0_w, *lw = which ? w.rots_ln_w : w.trs_ln_w,
                    *lb = which ? w.rots_ln_b : w.trs_ln_b, *w4 = which ? w.rots4_w : w.trs4_w;
        const float nrm = sqrtf((pred[0] * pred[0] + pred[1] * pred[1]) + pred[2] * pred[2]);
        sin_[0] = nrm;
        memcpy(sin_ + 1, temb, sizeof(float) * Hi);
        linear(sin_, 1, Hi + 1, Hi + 1, w0, Hi + 1, NULL, Hi, hid, Hi, 0);
        layernorm(hid, Hi, lw, lb);
        float o = 0;
        for (int c = 0; c < Hi; ++c) o += siluf(hid[c]) * w4[c];
        const float sp = o > 20.0f ? o : log1pf(expf(o));   /* nn.Softplus(beta=1, threshold=20) */
        float *dst = which ? out->rot_score : out->tr_score;
        for (int d = 0; d < 3; ++d) dst[d] = pred[d] / (nrm + 1e-6f) * sp;
    }
    free(four); free(temb); free(sin_); free(hid);
    free(pos); free(ca); free(h); free(edges); free(eattr); free(radial); free(cdiff); free(coord);
    free(agg); free(cat); free(u); free(o2); free(cagg);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a-1 / a-2: randomize_pose + Euler_Maruyama_sampler (inference_base.py:318-340, :390-468) */
static void haar_rotation(rng_t *r, double Rm[9])
{   /* scipy Rotation.random(): normalised 4-D Gaussian quaternion (x,y,z,w) */
    double q[4], n = 0;
    for (int i = 0; i < 4; ++i) { q[i] = rng_normal(r); n += q[i] * q[i]; }
    n = sqrt(n);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    Rm[0] = 1 - 2 * (y * y + z * z); Rm[1] = 2 * (x * y - z * w); Rm[2] = 2 * (x * z + y * w);
    Rm[3] = 2 * (x * y + z * w); Rm[4] = 1 - 2 * (x * x + z * z); Rm[5] = 2 * (y * z - x * w);
    Rm[6] = 2 * (x * z - y * w); Rm[7] = 2 * (y * z + x * w); Rm[8] = 1 - 2 * (x * x + y * y);
}

int ora_sample(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x, const float *lig_x,
               const float *rec_pos, const float *lig_pos0, int num_steps, float eps, float tr_noise_scale,
               float rot_noise_scale, int noise_annealing, int use_clash_force, int ode, int max_forwards,
               uint64_t seed, const ora_inject *inj, ora_traj_out *out)
{
    const int N = R + L;
    int knn = hp->knn, ns = hp->n_sample;
    if (N < knn) { knn = N; ns = 0; }
    if (N < knn + ns) ns = N - knn;
    const int K = knn + ns;
    rng_t rng = {seed ^ 0xA5A5A5A55A5A5A5Aull};
    float *lig = (float *)malloc(sizeof(float) * L * 9);
    memcpy(lig, lig_pos0, sizeof(float) * L * 9);

    /* :404-405 time grid: torch.linspace(1, eps, num_steps) float32; dt = t[0] - t[1] */
    float *ts = (float *)malloc(sizeof(float) * num_steps);
    {
        const float step = (eps - 1.0f) / (float)(num_steps - 1);
        for (int i = 0; i < num_steps; ++i)
            ts[i] = i < num_steps / 2 ? 1.0f + step * (float)i : eps - step * (float)(num_steps - 1 - i);
    }
    const float dt = num_steps > 1 ? ts[0] - ts[1] : 0.0f;

    /* :412 randomize_pose */
    float c1[3], c2[3], R0f[9], tr_update[3], rot_update[3];
    const int all_atom = hp->family == 1;      /* src/inference.py:220-254 (second family) vs inference_base.py:318-352 */
    if (all_atom) { atom_mean(rec_pos, R, c1); atom_mean(lig, L, c2); }
    else { ca_mean(rec_pos, R, c1); ca_mean(lig, L, c2); }
    {
        double R0[9];
        if (inj && inj->R0) memcpy(R0, inj->R0, sizeof(R0)); else haar_rotation(&rng, R0);
        for (int i = 0; i < 9; ++i) R0f[i] = (float)R0[i];
        float draw[3];
        for (int d = 0; d < 3; ++d)
            draw[d] = (inj && inj->tr_draw) ? inj->tr_draw[d] : (float)(30.0 * rng_normal(&rng));
        for (int d = 0; d < 3; ++d) tr_update[d] = (draw[d] - c2[d]) + c1[d];
        rotate_about(lig, L, c2, R0f);
        for (int a = 0; a < L * 3; ++a)
            for (int d = 0; d < 3; ++d) lig[a * 3 + d] += tr_update[d];
        ora_matrix_to_axis_angle(R0f, rot_update);
    }
    if (out->init_pose) memcpy(out->init_pose, lig, sizeof(float) * L * 9);

    ora_score_out so;
    memset(&so, 0, sizeof(so));
    int forwards = 0;
    for (int i = 0; i < num_steps; ++i) {
        if (max_forwards > 0 && forwards >= max_forwards) break;
        const int is_last = (i == num_steps - 1);
        const float t = ts[i];
        const int32_t *ed = (inj && inj->edges) ? inj->edges + (int64_t)i * N * K : NULL;
        ora_score(hp, blob, R, L, rec_x, lig_x, rec_pos, lig, t, ed, seed * 1315423911ull + (uint64_t)i,
                  out->trace_scores != NULL, &so, NULL);
        ++forwards;
        if (out->trace_scores) {
            float *tsr = out->trace_scores + i * 8;
            memcpy(tsr, so.tr_score, 12); memcpy(tsr + 3, so.rot_score, 12);
            tsr[6] = so.energy; tsr[7] = (float)so.num_clashes;
        }
        float trn, rotn;
        if (noise_annealing) { trn = t; rotn = t; }
        else { trn = is_last ? 0.0f : tr_noise_scale; rotn = is_last ? 0.0f : rot_noise_scale; }
        float zr[3], zt[3], rot[3], tr[3];
        for (int d = 0; d < 3; ++d) zr[d] = (inj && inj->z_rot) ? inj->z_rot[i * 3 + d] : (float)rng_normal(&rng);
        for (int d = 0; d < 3; ++d) zt[d] = (inj && inj->z_tr) ? inj->z_tr[i * 3 + d] : (float)rng_normal(&rng);
        ora_torch_reverse(ora_so3_g(hp, (double)t), so.rot_score, dt, rotn, zr, ode, rot);   /* :439-444 */
        ora_torch_reverse(ora_r3_g(hp, (double)t), so.tr_score, dt, trn, zt, ode, tr);       /* :446-451 */
        if (all_atom) ora_modify_coords_all_atom(lig, L, rot, tr);
        else ora_modify_coords(lig, L, rot, tr);                                             /* :453 */
        for (int d = 0; d < 3; ++d) tr_update[d] += tr[d];                                   /* :455 */
        { float tmp[3]; ora_rot_compose(rot_update, rot, tmp); memcpy(rot_update, tmp, 12); } /* :456 */
        if (use_clash_force) {                                                               /* :458-461 */
            float cf[3];
            ora_clash_force(rec_pos, R, lig, L, cf);
            for (int a = 0; a < L * 3; ++a)
                for (int d = 0; d < 3; ++d) lig[a * 3 + d] += cf[d];
            for (int d = 0; d < 3; ++d) tr_update[d] += cf[d];
        }
        if (out->trace_pose) memcpy(out->trace_pose + (int64_t)i * L * 9, lig, sizeof(float) * L * 9);
        if (is_last && !(max_forwards > 0 && forwards >= max_forwards)) {                    /* :463-466 */
            const int32_t *ed2 = (inj && inj->edges) ? inj->edges + (int64_t)num_steps * N * K : NULL;
            ora_score(hp, blob, R, L, rec_x, lig_x, rec_pos, lig, t, ed2, seed * 1315423911ull + (uint64_t)num_steps,
                      1, &so, NULL);
            ++forwards;
            if (out->trace_scores) {
                float *tsr = out->trace_scores + num_steps * 8;
                memcpy(tsr, so.tr_score, 12); memcpy(tsr + 3, so.rot_score, 12);
                tsr[6] = so.energy; tsr[7] = (float)so.num_clashes;
            }
        }
    }
    if (out->lig_pos) memcpy(out->lig_pos, lig, sizeof(float) * L * 9);
    memcpy(out->rot_update, rot_update, 12);
    memcpy(out->tr_update, tr_update, 12);
    out->energy = so.energy;
    out->num_clashes = so.num_clashes;
    free(lig); free(ts);
    return forwards;
}

/* Trajectory-parallel form of the sampler: n_traj INDEPENDENT trajectories (inference_base.py:644-657 runs them one after the other;
 * they interact only in the final arg-min over energies), one single-threaded trajectory per OpenMP thread, seeds seed0 + index.
 * This is how a CPU would run the headline metric (trajectories/s) on all its cores: no fork / join inside an evaluation, no shared
 * write traffic - every nested parallel region of ora_score runs on the calling thread.  bench.py's cpu_baseline times it on a
 * bounded number of evaluations per trajectory (max_forwards).  energy / clashes / forwards: per-trajectory outputs [n_traj] or NULL. */
int ora_sample_many(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x, const float *lig_x,
                    const float *rec_pos, const float *lig_pos0, int num_steps, float eps, float tr_noise_scale,
                    float rot_noise_scale, int max_forwards, uint64_t seed0, int n_traj, int n_threads,
                    float *energy, int64_t *clashes, int *forwards, float *updates)
{
    if (n_traj <= 0) return 0;
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (n_threads > n_traj) n_threads = n_traj;
    const int levels = omp_get_max_active_levels();
    omp_set_max_active_levels(1);      /* the evaluations' own parallel regions stay on the trajectory's thread */
#ifdef __GLIBC__
    /* every evaluation allocates and frees megabyte-sized temporaries: by default glibc serves those with mmap / munmap, and a
     * hundred threads of ONE process then queue on the address-space lock and re-fault their pages each time.  Keep them in the
     * per-thread arenas instead (first touch only). */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
#endif
    int total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : total)
    for (int k = 0; k < n_traj; ++k) {
        ora_traj_out out;
        memset(&out, 0, sizeof(out));
        const int nf = ora_sample(hp, blob, R, L, rec_x, lig_x, rec_pos, lig_pos0, num_steps, eps, tr_noise_scale, rot_noise_scale,
                                  0, 0, 0, max_forwards, seed0 + (uint64_t)k, NULL, &out);
        if (energy) energy[k] = out.energy;
        if (clashes) clashes[k] = out.num_clashes;
        if (forwards) forwards[k] = nf;
        if (updates) { memcpy(updates + 6 * k, out.rot_update, 12); memcpy(updates + 6 * k + 3, out.tr_update, 12); }
        total += nf;
    }
    omp_set_max_active_levels(levels);
    return total;
}
