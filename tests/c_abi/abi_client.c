/*
 * abi_client.c - a plain C99 caller of the drop-in boundary (include/dfmdock_amd.h): no Python, no torch.
 *
 * usage: abi_client <in.bin> <out.bin>
 *   in.bin : int32 R, L, B, n_blob ; float blob[n_blob] ; rec_x[R*lm] ; lig_x[L*lm] ; rec_pos[R*9] ; lig_pos[L*9] ;
 *            poses[B*L*9] ; t[B]
 *            then the sampler half: int32 S ; float R0[9], tr_draw[3], z_rot[S*3], z_tr[S*3] ; int32 edges[(S+1)*N*K]
 *            (one injected trajectory: every random draw and every edge list of a reference sampler run)
 *   out.bin: float tr_score[B*3], rot_score[B*3], energy[B] ; int32 num_clashes[B] ; float f[B*L*3]
 *            then float lig_pos[L*9], rot_update[3], tr_update[3], energy ; int32 num_clashes ; float trace_pose[S*L*9] ;
 *            double g_r3, sigma_r3, g_so3, sigma_so3 at t = 0.487692297 (dfm_diffusion_coef)
 * tests/test_gpu_c_abi.py builds this with gcc, runs it on the GPU box and compares out.bin with what the ctypes
 * binding returns for the same inputs (must be bit-identical: it is the same library) and with the reference's rollout.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dfmdock_amd.h"

static void *xread(FILE *f, size_t n, size_t sz)
{
    void *p = malloc(n * sz + 1);
    if (!p || fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 2; }
    int32_t hdr[4];
    if (fread(hdr, 4, 4, fi) != 4) return 2;
    const int R = hdr[0], L = hdr[1], B = hdr[2];
    const size_t nb = (size_t)hdr[3];
    dfm_hparams hp;
    dfm_default_hparams(&hp);
    if ((int64_t)nb != dfm_param_count(&hp)) { fprintf(stderr, "blob size mismatch\n"); return 2; }
    float *blob = xread(fi, nb, 4);
    float *rec_x = xread(fi, (size_t)R * hp.lm_embed_dim, 4), *lig_x = xread(fi, (size_t)L * hp.lm_embed_dim, 4);
    float *rec_pos = xread(fi, (size_t)R * 9, 4), *lig_pos = xread(fi, (size_t)L * 9, 4);
    float *poses = xread(fi, (size_t)B * L * 9, 4), *t = xread(fi, (size_t)B, 4);
    int32_t S = 0;
    if (fread(&S, 4, 1, fi) != 1 || S < 1) { fprintf(stderr, "no sampler section\n"); return 2; }
    const int N = R + L, Kdeg = N < 60 ? N : 60;
    float *R0 = xread(fi, 9, 4), *tr_draw = xread(fi, 3, 4), *z_rot = xread(fi, (size_t)S * 3, 4), *z_tr = xread(fi, (size_t)S * 3, 4);
    int32_t *inj_edges = xread(fi, (size_t)(S + 1) * N * Kdeg, 4);
    fclose(fi);

    if (dfm_set_device(0) != DFM_OK) { fprintf(stderr, "dfm_set_device: %s\n", dfm_last_error()); return 3; }
    /* error convention: NULL handle + message, negative status */
    if (dfm_model_create(blob, nb - 1, &hp) != NULL) { fprintf(stderr, "short blob accepted\n"); return 4; }
    dfm_model *m = dfm_model_create(blob, nb, &hp);
    if (!m) { fprintf(stderr, "dfm_model_create: %s\n", dfm_last_error()); return 3; }
    dfm_complex *cx = dfm_complex_create(m, rec_x, lig_x, rec_pos, lig_pos, R, L);
    if (!cx) { fprintf(stderr, "dfm_complex_create: %s\n", dfm_last_error()); return 3; }

    /* pose / homomer setters: a 66-channel model has no sym channel; re-setting the same poses changes nothing */
    if (dfm_complex_set_homomer(cx, 1) != DFM_E_INVALID || dfm_complex_set_homomer(cx, 0) != DFM_OK) { fprintf(stderr, "set_homomer\n"); return 4; }
    if (dfm_complex_set_pose(cx, rec_pos, lig_pos) != DFM_OK || dfm_complex_set_pose(cx, NULL, NULL) != DFM_OK) { fprintf(stderr, "set_pose: %s\n", dfm_last_error()); return 4; }
    if (dfm_complex_degree(cx) != (R + L < 20 ? R + L : (R + L < 60 ? R + L : 60))) { fprintf(stderr, "degree\n"); return 4; }

    float *tr = calloc((size_t)B * 3, 4), *rot = calloc((size_t)B * 3, 4), *en = calloc(B, 4), *f = calloc((size_t)B * L * 3, 4);
    int32_t *cl = calloc(B, 4);
    dfm_score_out out = {0};
    out.tr_score = tr; out.rot_score = rot; out.energy = en; out.num_clashes = cl; out.f = f;
    if (dfm_score(cx, 0, poses, t, NULL, 7, DFM_F_ENERGY, &out) != DFM_E_INVALID) { fprintf(stderr, "B = 0 accepted\n"); return 4; }
    int rc = dfm_score(cx, B, poses, t, NULL, 7, DFM_F_ENERGY, &out);      /* fp32 engine, native graph from seed 7 */
    if (rc != DFM_OK) { fprintf(stderr, "dfm_score: %d %s\n", rc, dfm_last_error()); return 3; }

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 2; }
    fwrite(tr, 4, (size_t)B * 3, fo); fwrite(rot, 4, (size_t)B * 3, fo); fwrite(en, 4, B, fo);
    fwrite(cl, 4, B, fo); fwrite(f, 4, (size_t)B * L * 3, fo);

    /* sampler half of the boundary: one trajectory with every draw injected (fp32 engine), traced */
    if (dfm_complex_degree(cx) != Kdeg) { fprintf(stderr, "degree %d != %d\n", dfm_complex_degree(cx), Kdeg); return 4; }
    dfm_inject inj = {0};
    inj.R0 = R0; inj.tr_draw = tr_draw; inj.z_rot = z_rot; inj.z_tr = z_tr; inj.edges = inj_edges;
    float *fin_pose = calloc((size_t)L * 9, 4), *trace = calloc((size_t)S * L * 9, 4), rotu[3], tru[3], e1 = 0.f;
    int32_t c1 = 0;
    dfm_traj_out to = {0};
    to.lig_pos = fin_pose; to.rot_update = rotu; to.tr_update = tru; to.energy = &e1; to.num_clashes = &c1; to.trace_pose = trace;
    /* errors first: eps outside [0,1] is the reference's ValueError (so3_diffuser.py:212-213); steps < 1; NULL output */
    if (dfm_sample(cx, 1, S, -0.5f, 0.5f, 0.5f, 0, 0, &inj, &to) != DFM_E_INVALID) { fprintf(stderr, "eps < 0 accepted\n"); return 4; }
    if (dfm_sample(cx, 1, 0, 1e-3f, 0.5f, 0.5f, 0, 0, &inj, &to) != DFM_E_INVALID) { fprintf(stderr, "0 steps accepted\n"); return 4; }
    if (dfm_sample(cx, 1, S, 1e-3f, 0.5f, 0.5f, 0, 0, &inj, NULL) != DFM_E_INVALID) { fprintf(stderr, "NULL out accepted\n"); return 4; }
    if (dfm_last_error()[0] == 0) { fprintf(stderr, "no error message\n"); return 4; }
    rc = dfm_sample(cx, 1, S, 1e-3f, 0.5f, 0.5f, 0, 0, &inj, &to);
    if (rc != DFM_OK) { fprintf(stderr, "dfm_sample: %d %s\n", rc, dfm_last_error()); return 3; }
    fwrite(fin_pose, 4, (size_t)L * 9, fo); fwrite(rotu, 4, 3, fo); fwrite(tru, 4, 3, fo); fwrite(&e1, 4, 1, fo);
    fwrite(&c1, 4, 1, fo); fwrite(trace, 4, (size_t)S * L * 9, fo);
    double g[4];
    if (dfm_diffusion_coef(&hp, 0, 0.487692297, &g[0], &g[1]) != DFM_OK || dfm_diffusion_coef(&hp, 1, 0.487692297, &g[2], &g[3]) != DFM_OK) {
        fprintf(stderr, "dfm_diffusion_coef: %s\n", dfm_last_error()); return 3;
    }
    if (dfm_diffusion_coef(&hp, 1, 1.5, &g[2], &g[3]) != DFM_E_INVALID) { fprintf(stderr, "t = 1.5 accepted on SO(3)\n"); return 4; }
    if (dfm_diffusion_coef(&hp, 1, 0.487692297, &g[2], &g[3]) != DFM_OK) return 3;
    fwrite(g, 8, 4, fo);
    fclose(fo);
    dfm_complex_destroy(cx);
    dfm_model_destroy(m);
    printf("abi_client ok: B=%d energy[0]=%g clashes[0]=%d | sampler: %d steps, final energy %g, config: %s\n", B, en[0], cl[0], S, e1,
           dfm_config_string());
    return 0;
}
