/*
 * dfm_oracle.h - CPU restatement (plain C, fp32) of the DFMDock sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / CPU comparator.  The product path
 * (dfmdock_amd/csrc) never links or calls it.
 *
 * Parity pin: every function below is checked against golden vectors captured
 * by RUNNING the reference (tests/golden/make_golden.py) - see
 * tests/test_oracle_golden.py.  The one piece of arithmetic on the path that
 * does not live under /root/reference is torch_geometric==2.6.0 GraphNorm
 * (reference call site src/models/egnn.py:6,:74); it is restated from its
 * published formula (ora_graphnorm) and is unpinned by any reference test.
 */
#ifndef DFM_ORACLE_H
#define DFM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int lm_embed_dim, positional_embed_dim, spatial_embed_dim;
    int node_dim, edge_dim, inner_dim, depth, knn, n_sample;
    float cut_off, mask_dist;
    double r3_min_sigma, r3_max_sigma, so3_min_sigma, so3_max_sigma;   /* Python floats in the reference */
    int family;    /* 0: Score_Net (score_net_mlsb.py); 1: EGNN_Net behind DFMDock.forward (egnn_net.py, DFMDock.py:68-75) */
    int agg_mean;  /* family 1: `agg` == 'mean' (1) or 'sum' (0) */
    int homomer;   /* positional_embed_dim = 67: value of the 67th ("sym") position channel of this complex (default 0) */
} ora_hparams;

typedef struct {
    float tr_score[3], rot_score[3];
    float energy;
    int64_t num_clashes;
    float confidence;   /* family 1: confidence_logits (egnn_net.py:447); 0 otherwise */
} ora_score_out;

/* optional intermediates of one score evaluation (any pointer may be NULL) */
typedef struct {
    float *f;        /* [L,3]              */
    float *pos_out;  /* [N,3]  CA after the last layer's coordinate update */
    float *h_layers; /* [depth,N,H] node features after each layer         */
    int8_t *bins;    /* [N,K,4] dist/omega/theta/phi bins of the used edges */
    int8_t *relpos;  /* [N,K]                                               */
    int32_t *edges;  /* [N,K] edge list actually used                       */
    float *ires;     /* [N]                                                 */
    float *dist;     /* [R,L,64] family 1: dist_logits (egnn_net.py:447), or NULL */
    const int8_t *bins_in; /* [N,K,4] INPUT, or NULL: use these feature bins instead of the computed ones (tests isolate a
                              bin-boundary flip - one ulp of atan2 / acos between torch's kernels and libm - this way) */
} ora_debug;

typedef struct {
    const double *R0;      /* [9] initial rotation (row-major), or NULL = draw   */
    const float *tr_draw;  /* [3] the N(0,30^2) draw, or NULL                    */
    const float *z_rot;    /* [steps,3] N(0,1) draws for SO(3), or NULL          */
    const float *z_tr;     /* [steps,3] N(0,1) draws for R^3, or NULL            */
    const int32_t *edges;  /* [steps+1,N,K] per-forward edge lists, or NULL      */
} ora_inject;

typedef struct {
    float *lig_pos;       /* [L,9] final ligand backbone                         */
    float rot_update[3], tr_update[3];
    float energy;
    int64_t num_clashes;
    /* optional per-step traces (NULL to skip) */
    float *trace_pose;    /* [steps,L,9] pose after each step                    */
    float *trace_scores;  /* [steps+1,8]: tr_score, rot_score, energy, clashes   */
    float *init_pose;     /* [L,9]                                               */
} ora_traj_out;

int64_t ora_param_count(const ora_hparams *hp);
int ora_num_threads(void);
void ora_set_num_threads(int n);

/* a-3 / a-4: r3_diffuser.py:20-24, so3_diffuser.py:210-227 (float64, as numpy) */
double ora_r3_sigma(const ora_hparams *hp, double t);
double ora_r3_g(const ora_hparams *hp, double t);
double ora_so3_sigma(const ora_hparams *hp, double t);   /* NaN if t outside [0,1] (ValueError) */
double ora_so3_g(const ora_hparams *hp, double t);
/* torch_reverse (r3_diffuser.py:40-55 == so3_diffuser.py:344-369): z already scaled by nothing */
void ora_torch_reverse(double g, const float score[3], float dt, float noise_scale,
                       const float z[3], int ode, float out[3]);

/* a-14: geometry.py:18-200, inference_base.py:311-352 */
void ora_axis_angle_to_quaternion(const float aa[3], float q[4]);
void ora_quaternion_to_matrix(const float q[4], float R[9]);
void ora_axis_angle_to_matrix(const float aa[3], float R[9]);
void ora_matrix_to_quaternion(const float R[9], float q[4]);
void ora_quaternion_to_axis_angle(const float q[4], float aa[3]);
void ora_matrix_to_axis_angle(const float R[9], float aa[3]);
void ora_rot_compose(const float r1[3], const float r2[3], float out[3]);
void ora_modify_coords(float *x /*[n,9] in/out*/, int n, const float rot[3], const float tr[3]);
void ora_modify_coords_all_atom(float *x /*[n,9] in/out*/, int n, const float rot[3], const float tr[3]);  /* inference.py:244-254 */
/* a-17: inference_base.py:366-384 (closed-form gradient of the reference's autograd) */
void ora_clash_force(const float *rec /*[R,9]*/, int R, const float *lig /*[L,9]*/, int L, float out[3]);

/* a-6 / a-7 / a-8: coords6d.py:10-103, score_net_mlsb.py:30-70, inference_base.py:230-292 */
void ora_coords6d_full(const float *pos /*[N,9]*/, int N, float *dist, float *omega, float *theta,
                       float *phi /* each [N,N] */);
void ora_bins_full(const ora_hparams *hp, const float *pos, int N, int8_t *bins /*[N,N,4]*/);
void ora_relpos_full(int R, int L, int8_t *rel /*[N,N]*/);
/* a-10: score_net_mlsb.py:85-135; kNN exact; sampling by exponential race with the oracle's own RNG */
void ora_knn_sample(const ora_hparams *hp, const float *ca /*[N,3]*/, int N, uint64_t seed,
                    int32_t *edges /*[N,K]*/, int *K_out);

/* a-5 (+a-9, a-11, a-12, a-13): one score evaluation */
int ora_score(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x,
              const float *lig_x, const float *rec_pos, const float *lig_pos, float t,
              const int32_t *edges /*[N,K] or NULL*/, uint64_t seed, int want_energy,
              ora_score_out *out, ora_debug *dbg);

/* a-1 / a-2: one trajectory of the Euler-Maruyama sampler */
int ora_sample(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x,
               const float *lig_x, const float *rec_pos, const float *lig_pos, int num_steps,
               float eps, float tr_noise_scale, float rot_noise_scale, int noise_annealing,
               int use_clash_force, int ode, int max_forwards /* <=0: all */, uint64_t seed,
               const ora_inject *inj, ora_traj_out *out);

/* n_traj independent trajectories, one single-threaded trajectory per OpenMP thread (inference_base.py:644-657 runs them in
 * sequence; they only meet in the arg-min over energies): the all-cores CPU form of the headline metric.  Returns the number of
 * score evaluations done in total; per-trajectory outputs may be NULL. */
int ora_sample_many(const ora_hparams *hp, const float *blob, int R, int L, const float *rec_x, const float *lig_x,
                    const float *rec_pos, const float *lig_pos, int num_steps, float eps, float tr_noise_scale,
                    float rot_noise_scale, int max_forwards, uint64_t seed0, int n_traj, int n_threads,
                    float *energy /*[n_traj]*/, int64_t *clashes /*[n_traj]*/, int *forwards /*[n_traj]*/,
                    float *updates /*[n_traj,6]: rot_update, tr_update*/);

#ifdef __cplusplus
}
#endif
#endif
