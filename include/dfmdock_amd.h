/*
 * dfmdock_amd.h - C ABI of the MI355X-native DFMDock sampling engine.
 *
 * This is the drop-in boundary for the ONE hot path this project accelerates
 * (SURVEY.md section 8): the reverse-diffusion sampler over rigid-body poses and
 * the score network it calls.  The reference has no FFI layer - the path sits
 * behind plain Python call signatures - so each entry point below names the
 * reference call it replaces (paths relative to the reference checkout):
 *
 *   dfm_model_create    <- Score_Model.load_from_checkpoint / Score_Net.__init__
 *                          (src/inference_base.py:611-616, src/models/score_net_mlsb.py:249-341)
 *   dfm_complex_create  <- get_batch_from_inputs + get_position_matrix
 *                          (src/inference_base.py:192-253)
 *   dfm_score           <- Score_Model.forward(batch)  (src/models/score_model_mlsb.py:61-63 ->
 *                          src/models/score_net_mlsb.py:343-425); with dfm_hparams.family = 1:
 *                          DFMDock.forward(batch) (src/models/DFMDock.py:68-75 -> src/models/egnn_net.py:408-505)
 *   dfm_sample          <- Euler_Maruyama_sampler(model, batch, ...) (src/inference_base.py:390-468),
 *                          batched over B independent trajectories
 *   dfm_diffusion_coef  <- R3Diffuser.diffusion_coef / SO3Diffuser.diffusion_coef
 *                          (src/utils/r3_diffuser.py:23-24, src/utils/so3_diffuser.py:219-227)
 *
 * Conventions: plain pointers and sizes, caller owns every host buffer, the
 * library owns device memory behind opaque handles, no global state besides the
 * thread-local error string.  All entry points return 0 on success and a
 * negative dfm_status otherwise (the reference raises Python exceptions:
 * ValueError for t outside [0,1] -> DFM_E_INVALID).  A handle must not be used by
 * two host threads at once (no internal locking per handle); DIFFERENT complex
 * handles of one model may be driven from different host threads concurrently:
 * each complex owns a non-blocking HIP stream and every upload, launch and
 * release of the handle is ordered on that stream alone, so creating / checking
 * the next complex of a set overlaps the sampling of the current one
 * (dfmdock_amd/driver.py: run_set; the reference's loop is serial,
 * src/inference_mlsb.py:415-439).  One process per GPU.
 * Status of that guarantee (r05 / r06): concurrent handles once produced a silent miscompute - waves of an LDS-free geometry kernel
 * binning wrong dihedrals from correct inputs while another handle's 160 KiB message-kernel workgroups were resident.  r06 traced it to
 * one instruction form hipcc's SLP vectoriser had emitted (a packed fp32 multiply with op_sel = [0,1]: low result from the HIGH half of
 * source 1) and reproduced it without the engine: while one wave of a SIMD executes a 16-bit-input MFMA, such an instruction issued by
 * another wave of that SIMD returns wrong values (profiles/r06_concurrency.txt, tools/ubench/pk_erratum.hip) - hardware, not data flow.  Fenced three ways: no kernel of the library contains such an instruction
 * (built with -fno-slp-vectorize, disassembly audited by tests/test_abi_cpu.py: THE fence), every kernel launch holds some LDS (keeps
 * it off CUs that a message kernel's workgroup fills), and dfmdock_amd/driver.py re-samples one complex alone after an overlapped run and compares bit for bit
 * (falling back to the serial driver on a mismatch).  tests/test_gpu_concurrency.py holds the shipped build to 0 deviations in a
 * victim x aggressor matrix over every kernel of the path.  CALLERS that run their OWN kernels on the same GPU next to this library
 * should know the form (tools/ubench/pkmul_victim.hip).
 * A model lives on the device that was current at dfm_model_create (dfm_set_device), a
 * complex on its model's device; every entry point switches to the handle's device for
 * the duration of the call and restores the caller's current device.
 * Everything computes on the GPU: there is no CPU fallback in this library.
 */
#ifndef DFMDOCK_AMD_H
#define DFMDOCK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dfm_model dfm_model;
typedef struct dfm_complex dfm_complex;

typedef enum {
    DFM_OK = 0,
    DFM_E_INVALID = -1,   /* bad argument (shape, range, NULL)          */
    DFM_E_HIP = -2,       /* HIP runtime error (see dfm_last_error)      */
    DFM_E_OOM = -3,       /* device or host allocation failed            */
    DFM_E_NODEVICE = -4   /* no usable gfx950 device                     */
} dfm_status;

/* configs/model/score_model_mlsb.yaml:3-27 (+ score_net_mlsb.py:33,:85 constants) */
typedef struct {
    int lm_embed_dim;          /* 1301 = 1280 (ESM-2) + 21 (one-hot)   */
    int positional_embed_dim;  /* 66, or 67 = 66 relpos + 1 "sym" channel (configs/model/DFMDock.yaml:5):
                                  the homomer flag of the complex, see dfm_complex_set_homomer            */
    int spatial_embed_dim;     /* 100 = 40 + 24 + 24 + 12               */
    int node_dim;              /* 256 (only value supported by kernels) */
    int edge_dim;              /* 128                                   */
    int inner_dim;             /* 128                                   */
    int depth;                 /* 6                                     */
    int knn;                   /* 20                                    */
    int n_sample;              /* 40                                    */
    float cut_off;             /* 20.0  energy mask                     */
    float mask_dist;           /* 22.0  angle-feature mask              */
    double r3_min_sigma, r3_max_sigma;    /* 0.1, 30.0                  */
    double so3_min_sigma, so3_max_sigma;  /* 0.1, 1.5 (logarithmic)     */
    int family;                /* 0: Score_Net (src/models/score_net_mlsb.py:249-425, what inference_single.py loads);
                                  1: EGNN_Net behind DFMDock.forward (src/models/egnn_net.py:408-505,
                                     src/models/DFMDock.py:68-75, configs/model/DFMDock.yaml: mask_dist 20):
                                     no coordinate update, pair force / energy / confidence heads          */
    int agg_mean;              /* family 1: `agg` 'mean' (1, default) or 'sum' (0), egnn_net.py:438-474     */
} dfm_hparams;

/* flags for dfm_score / dfm_sample */
enum {
    DFM_F_MFMA16 = 1u << 0,          /* the 16-bit MFMA engine (default without flag: exact fp32): per-edge 256 x 256
                                        contractions on v_mfma_f32_32x32x16 with fp16 operands and fp32 accumulation in every
                                        layer, gathered operands (Wb h_j, lookup tables, A_i) stored as fp16, node-level GEMMs
                                        as three split-bf16 terms (~1e-5), geometry / GraphNorm statistics / heads / SDE step
                                        fp32.  dfm_config_string() describes the plan in force.                         */
    DFM_F_BF16 = DFM_F_MFMA16,       /* name of rounds 1-2 (the engine then took bf16 operands in layers 0..depth-2)    */
    DFM_F_ENERGY = 1u << 1,          /* dfm_score: also evaluate the energy head                 */
    DFM_F_NOISE_ANNEALING = 1u << 2, /* inference_base.py:428-430                                */
    DFM_F_CLASH_FORCE = 1u << 3,     /* inference_base.py:458-461                                */
    DFM_F_ODE = 1u << 4,             /* so3_diffuser.py:367-368                                  */
    DFM_F_PROFILE = 1u << 5,         /* time the dominant kernel with HIP events + its in-kernel clock stamps (dfm_get_profile) */
    DFM_F_STEP_ENERGY = 1u << 6,     /* dfm_sample: evaluate the energy head on every step (traces) */
    DFM_F_F16 = 1u << 7,             /* like DFM_F_MFMA16 but A_i = Wa h_i + b1 stays fp32 (one more load per chunk)     */
    DFM_F_IRES = 1u << 8,            /* dfm_score: also evaluate the interface-residue head (score_net_mlsb.py:383) */
    DFM_F_DIST = 1u << 10,           /* dfm_score, family 1: also evaluate dist_logits = to_dist(cat[h_r, h_l, D]) over all R x L pairs
                                        (egnn_net.py:347-352,:447; a training-loss input, never read by a sampler); fp32 in every engine */
    DFM_F_BF16_OPS = 1u << 9,        /* with DFM_F_MFMA16: bf16 instead of fp16 MFMA operands in layers 0..depth-2 (the r02
                                        plan; ~3 % faster).  OUTSIDE SURVEY 8(d)'s 1e-2 gate: measured up to 1.5e-2 on f /
                                        tr_score / rot_score over four weight draws (profiles/r03_tol_report.txt) - an opt-in
                                        for callers who accept that; tested at 2e-2                                      */
    /* Layer 0 behind the per-complex message table (DFM_F_MFMA16 engine and the fp32 engine, a table each; src/models/egnn.py:95-104 evaluated once per
       intra-chain residue pair instead of once per edge, trajectory and step - in layer 0 the node features are the pose-independent
       embedding, and the geometry of two residues of one chain does not change under the rigid motion of the ligand).  dfm_sample
       uses the table whenever the complex is eligible (depth >= 2, (R^2 + L^2) * 516 B and the per-batch row buffers within the
       budgets in api.hip), whatever the batch size; dfm_score is a pure function of its arguments and uses it only on request.
       Inter-chain edges, and intra-chain edges whose feature bins in the pose at hand differ from the table's, go through the edge
       model as before.  Against the direct evaluation the only difference is the fp16 rounding of each stored message before the
       K-row sum (fp32 engine: fp32 rows of 1 KiB, only the ORDER of the K-row sum differs, <= 2e-5; measured: tests/test_gpu_l0_table.py).                                                                  */
    DFM_F_L0_TABLE = 1u << 11,       /* dfm_score: use (and if necessary build) the table.  A hit needs the edge's bin code AND its squared
                                        C-alpha distance in the pose at hand to equal those the entry was built with on the handle's stored
                                        pose (the distance up to the rounding of a rigid motion, 1e-3 A x max(d, 1)): any lig_pos is safe - an
                                        intra-chain pair whose geometry differs (another conformer, a perturbed backbone) is a miss and goes
                                        through the edge model like an inter-chain edge.  Whether a complex uses the table depends on the
                                        complex alone (R^2 + L^2 <= 8 M pairs), never on B: a batch whose per-edge buffers (532 B per edge)
                                        do not fit fails with DFM_E_OOM instead of changing the arithmetic.                       */
    DFM_F_NO_L0_TABLE = 1u << 12,    /* dfm_sample: evaluate layer 0 directly                                            */
    DFM_F_GRAPH = 1u << 13           /* dfm_sample: capture ONE step (score evaluation + heads + update) as a hipGraph and replay it
                                        num_steps times instead of enqueueing every launch (ignored when anything is injected,
                                        traced or profiled).  Same kernels and arguments - the per-step / per-call scalars are
                                        read from device memory - so the trajectories are bitwise those of the plain path
                                        (tests/test_gpu_graph.py).  Opt-in: on MI355X the stream is paced by the device-side
                                        dispatch of ~30 dependent launches per evaluation, not by the host, and the replay
                                        measures 0.2 - 1.9 % SLOWER at B = 1 ... 120 (profiles/r04_graph_ab.txt); it is there
                                        for hosts that cannot keep a stream fed (DFM_GRAPH=1 in the environment: default on) */
};

/* Output of dfm_score.  Required: tr_score, rot_score.  Any other pointer may be NULL. */
typedef struct {
    float *tr_score;      /* [B,3]                                                      */
    float *rot_score;     /* [B,3]                                                      */
    float *energy;        /* [B]      (needs DFM_F_ENERGY)                              */
    int32_t *num_clashes; /* [B]      (needs DFM_F_ENERGY)                              */
    float *f;             /* [B,L,3]  per-ligand-residue force                          */
    /* debug / parity taps */
    float *h_last;        /* [B,N,H]  node features after the last layer.  They feed the energy / ires / dist heads only
                                      (score_net_mlsb.py:383-390): a call that asks for none of those and passes h_last = NULL
                                      computes the last layer for the ligand nodes alone (all that f needs, :396-398) -
                                      bitwise the same f / tr_score / rot_score                               */
    float *h_first;       /* [B,N,H]  node features after the first layer               */
    int32_t *edges;       /* [B,N,K]  edge list actually used                           */
    uint32_t *edge_codes; /* [B,N,K]  packed feature bins: d | omega<<6 | theta<<11 | phi<<16 | relpos<<20 */
    float *confidence;    /* [B]      family 1 + DFM_F_ENERGY: confidence_logits (egnn_net.py:444); may be NULL */
    float *ires;          /* [B,N]    needs DFM_F_IRES: to_ires(node_out) (score_net_mlsb.py:297-303,:383; family 1:
                                      ires_logits, egnn_net.py:362-368,:462); may be NULL                     */
    float *dist_logits;   /* [B,R,L,64] needs DFM_F_DIST and a family-1 model (egnn_net.py:447,:500); may be NULL  */
} dfm_score_out;

/* Injected randomness for parity tests (every pointer may be NULL = draw natively with Philox) */
typedef struct {
    const float *R0;       /* [B,9]  initial rotation matrices (row-major)              */
    const float *tr_draw;  /* [B,3]  the N(0,30^2) draws of randomize_pose               */
    const float *z_rot;    /* [B,steps,3] N(0,1) draws of the SO(3) update                */
    const float *z_tr;     /* [B,steps,3] N(0,1) draws of the R^3 update                  */
    const int32_t *edges;  /* [B,steps+1,N,K] edge lists, one per score evaluation       */
} dfm_inject;

typedef struct {
    float *lig_pos;       /* [B,L,9]  final ligand backbone (N,CA,C)                     */
    float *rot_update;    /* [B,3]    accumulated rotation, axis-angle                   */
    float *tr_update;     /* [B,3]    accumulated translation                            */
    float *energy;        /* [B]      energy of the final pose                           */
    int32_t *num_clashes; /* [B]                                                         */
    float *final_scores;  /* [B,6]    tr_score, rot_score of the final evaluation (may be NULL) */
    /* optional traces (NULL to skip) */
    float *trace_pose;    /* [B,steps,L,9]  pose after every step                        */
    float *trace_scores;  /* [B,steps+1,8]  tr_score, rot_score, energy, num_clashes per evaluation */
    float *init_pose;     /* [B,L,9]        pose after randomize_pose                    */
} dfm_traj_out;

typedef struct {
    double edge_kernel_ms;    /* summed HIP-event time of the per-edge message kernel     */
    int64_t edge_kernel_launches;
    int64_t edge_rows;        /* edge rows (B*N*K) processed by those launches           */
    double total_ms;          /* HIP-event time of the whole last dfm_sample / dfm_score  */
    double phase_cycles[4];   /* diagnostic builds only (-DDFM_EDGE_STAMP), else 0: shader cycles per 32-row tile of the
                                 message kernel's third launch, mean over the 8 waves of workgroup 0:
                                 prologue | chunks 0-6 | chunk 7 + bias | epilogue                */
    double slot_cycles[16];   /* diagnostic builds only: summed cycles between consecutive MFMA slots of chunk 3 (wave 0) */
    /* layer 0 behind the message table (DFM_F_L0_TABLE): its launches are NOT part of edge_kernel_* above */
    int64_t l0_evals;         /* evaluations whose layer 0 ran through the table                                 */
    int64_t l0_edges;         /* edges of those layer-0 passes (B*N*K each)                                       */
    int64_t l0_miss_rows;     /* of which evaluated by the edge model (inter-chain + bin mismatches)              */
    double l0_rows_ms;        /* summed HIP-event time of the row-list message launches                           */
    double l0_gather_ms;      /* ... of the gather-sum launches                                                  */
    double l0_build_ms;       /* table build of this call (0 when the table already existed)                      */
    int64_t edge_lig_launches;/* of edge_kernel_launches: last-layer launches over the ligand nodes only          */
    double edge_lig_ms;       /* their share of edge_kernel_ms                                                    */
    /* the shader clock the message kernel actually ran at (the chip's power management, not the kernel, sets it): workgroup 0's first wave
     * reads s_memtime (shader cycles) and s_memrealtime (100 MHz) when it starts and when it leaves, summed over the call's launches;
     * MHz = 100 * edge_shader_cycles / edge_ref_ticks.  0 when the call had no profiled message launch.                              */
    double edge_shader_cycles;
    double edge_ref_ticks;
} dfm_profile;

/* Output of dfm_complex_selfcheck: how far the 16-bit MFMA engine is from the fp32 engine (the reference's own arithmetic) on THIS
 * model and THIS complex, and how much of the fp16 range the model's activations use.  No reference call has a counterpart: the
 * reference computes in fp32 throughout; this is the runtime evidence SURVEY 8(d) gate (5) needs for weights the build has never
 * seen (src/inference_base.py:611-616 loads whatever checkpoint the user has).
 *
 * Deviations: L-inf over L-inf of f / tr_score / rot_score, |dE| / max(|E|, 0.1) of the energy, each the WORST over the n_eval
 * evaluations (same pose, n_eval engine-drawn graphs, n_eval times t); the gates are SURVEY 8(d)'s for 16-bit kernels.
 * The two scores are unit vectors of the pooled force mean_l f and torque mean_l (r_l x f_l) times a learned scale
 * (score_net_mlsb.py:396-411), so their deviation is governed by the force deviation over a cancellation ratio:
 * |d mean f| / |mean f| <= score_bound[0] = sqrt(3) dev_f max|f| / |mean_l f| and the same with r x f for the torque (rigorous for
 * the pooled vectors; the unit vector and the scale net add a model-dependent factor of order 1).  A small cancel_ratio means an
 * ill-conditioned pose - the reference's own fp32 scores are then sensitive to rounding as well - not a broken engine.
 * Range telemetry (from the fp32 pass, per layer l = 0 .. depth-1; the values as the 16-bit engine STORES them, i.e. times
 * log2(e) where the stored operand carries that factor): everything the 16-bit engine keeps in fp16 must stay below 65504 -
 * the engine saturates silently otherwise (conversions clamp).  range_ok = every entry below `limit` (6.0e4). */
typedef struct {
    int n_eval, depth;
    float dev_f, dev_tr_score, dev_rot_score, dev_energy;    /* worst over the evaluations                                   */
    float cancel_ratio[2];                                   /* |mean_l v| / mean_l |v| for v = f and v = r x f (fp32 pass): the
                                                                smallest over the evaluations                                */
    float score_bound[2];                                    /* bound of the relative deviation of the pooled force / torque  */
    float gate_f, gate_score, gate_energy;                   /* 1e-2, 1e-2, 3e-2                               */
    float limit;                                             /* 6.0e4                                                        */
    float max_h[9];       /* |h| entering layer l; [depth] = the final node features                                        */
    float max_A[8];       /* |log2e (Wa h_i + b1)|           stored fp16 (DFM_F_MFMA16) / fp32 (DFM_F_F16)                   */
    float max_Bm[8];      /* |log2e Wb h_j|                  stored fp16                                                     */
    float max_tab[8];     /* largest |entry| of the layer's merged lookup tables (log2e-scaled), stored fp16                 */
    float max_sum16[8];   /* bound of the packed fp16 sum Bm_j + two table rows: max_Bm + the two tables' maxima             */
    float max_pre[8];     /* |log2e pre-activation of edge_mlp.0|: the producer SiLU's output is stored fp16                 */
    float max_acc[8];     /* |log2e pre-activation of edge_mlp.2|: bounds the gated messages the last layer stores as fp16   */
    float headroom;       /* limit / largest of all the above (>= 1 when range_ok)                                          */
    int64_t saturated;    /* fp16 values found AT the saturation value (+-65504 or inf) in A / Bm of the 16-bit pass itself   */
    int range_ok;         /* 1: every tracked magnitude below limit and saturated == 0                                       */
    int dev_ok;           /* 1: dev_f <= gate_f, dev_energy <= gate_energy, each score deviation <= max(gate_score, 2 score_bound[.]) */
    int ok;               /* range_ok && dev_ok                                                                              */
} dfm_selfcheck_out;

const char *dfm_last_error(void);
/* One line describing the precision plan and every diagnostic environment switch / build knob in force in this process
 * (DFM_EDGE_SPLIT; DFM_LIB is the loader's): benches and tests print it, so that a run under
 * a stray variable cannot pass for the shipped engine.  The pointer stays valid for the life of the process. */
const char *dfm_config_string(void);
int dfm_device_count(int *count);
int dfm_set_device(int device);
/* fills hp with the reference configuration */
void dfm_default_hparams(dfm_hparams *hp);
/* number of floats the blob must hold for hp (state_dict order, see dfmdock_amd/weights.py) */
int64_t dfm_param_count(const dfm_hparams *hp);

dfm_model *dfm_model_create(const float *blob, size_t n_floats, const dfm_hparams *hp);
void dfm_model_destroy(dfm_model *m);

dfm_complex *dfm_complex_create(dfm_model *m, const float *rec_x /*[R,lm]*/, const float *lig_x /*[L,lm]*/,
                                const float *rec_pos /*[R,9]*/, const float *lig_pos /*[L,9]*/, int R, int L);
void dfm_complex_destroy(dfm_complex *cx);
/* Replace the poses stored by dfm_complex_create (either pointer may be NULL = keep): rec_pos [R,9] is what every later
 * dfm_score / dfm_sample call sees as the receptor, lig_pos [L,9] is the start pose of dfm_sample.  The node features and
 * everything derived from them stay resident - a caller that re-centres the complex every step (DFMDock.move_to_lig_center,
 * src/models/DFMDock.py:254-257) or docks several ligand conformations does not pay the feature upload again.  On an error
 * return the stored poses are unspecified (one of the two may have been replaced): call again. */
int dfm_complex_set_pose(dfm_complex *cx, const float *rec_pos_or_null, const float *lig_pos_or_null);
/* positional_embed_dim = 67 only: value of the 67th ("sym") position channel for this complex - 1 when receptor and ligand
 * have the same sequence (is_homomer, src/datasets/docking_dataset.py:129), 0 otherwise (the default).  DFM_E_INVALID for a
 * 66-channel model and flag != 0.  ASSUMPTION: the reference tree has no producer of a 67-channel position matrix
 * (utils/crop.get_position_matrix returns 66 channels and nothing concatenates is_homomer); the layout taken here is
 * [relpos one-hot 66 | flag], the same constant on every residue pair.  A checkpoint trained with another layout of that
 * channel needs this entry point revisited. */
int dfm_complex_set_homomer(dfm_complex *cx, int flag);
/* edges per node for this complex: min(N,20) + min(40, N-20) */
int dfm_complex_degree(const dfm_complex *cx);
/* Device blocks released by destroyed handles are parked per device for the next handle (a set driver creates and destroys a
 * complex every ~100 ms; hipMalloc / hipFree of gigabyte workspaces cost milliseconds and drain the device): at most
 * DFM_ALLOC_CACHE_FRAC (default 0.25) of the device's memory, DFM_ALLOC_CACHE=0 disables it.  dfm_trim_cache hands every parked
 * block of `device` (< 0: all devices; an index past the last device: nothing, returns 0) back to the driver - for processes that
 * share a GPU - and returns the bytes freed. */
long long dfm_trim_cache(int device);

/* B score evaluations of poses lig_pos[B,L,9] at times t[B] */
int dfm_score(dfm_complex *cx, int B, const float *lig_pos, const float *t, const int32_t *edges_or_null,
              uint64_t seed, uint32_t flags, dfm_score_out *out);

/* B independent trajectories of the Euler-Maruyama sampler */
int dfm_sample(dfm_complex *cx, int B, int num_steps, float eps, float tr_noise_scale, float rot_noise_scale,
               uint32_t flags, uint64_t seed, const dfm_inject *inj_or_null, dfm_traj_out *out);

int dfm_get_profile(const dfm_complex *cx, dfm_profile *p);

/* Runs the stored pose of the complex (dfm_complex_create / dfm_complex_set_pose) through the fp32 engine and through the 16-bit
 * engine `flags` selects (DFM_F_MFMA16, the default when neither is given, or DFM_F_F16; DFM_F_BF16_OPS is honoured) on the same
 * n_eval (1..16) engine-drawn graphs at times t[n_eval] (NULL: spread over [1, 0.001]) and fills `out` (see dfm_selfcheck_out).
 * Costs one fp32 + one 16-bit batched evaluation of n_eval poses (about 0.1 s at 300+300); the drivers call it once per complex. */
int dfm_complex_selfcheck(dfm_complex *cx, int n_eval, const float *t_or_null, uint64_t seed, uint32_t flags, dfm_selfcheck_out *out);

/* which: 0 = R^3 (VE), 1 = SO(3) (logarithmic).  Returns DFM_E_INVALID for t outside [0,1] on SO(3). */
int dfm_diffusion_coef(const dfm_hparams *hp, int which, double t, double *g_out, double *sigma_out);

#ifdef __cplusplus
}
#endif
#endif /* DFMDOCK_AMD_H */
