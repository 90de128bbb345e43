/* C ABI of the MDMA vector field (model: "mdma") in libpfm_hip.so -- gfx950.
 *
 * Replaces, for `SetFlowMatchingLitModule(model="mdma", net_config=...)` (configs/model/flow_matching_mdma.yaml):
 *   - CNF.forward -> MDMA.forward            (particle_fm/models/flow_matching_module.py:163-167, 191-204;
 *                                             particle_fm/models/components/mdma.py:87-176, Block :7-84)
 *   - CNF.decode with the fixed-step solvers (flow_matching_module.py:245-299)
 *   - FlowMatchingLoss / ConditionalFlowMatchingLoss / DroidLoss forward + autograd (components/losses.py:38-77, 101-136)
 *
 * The network: particles are embedded (Linear on cat(time embedding, x), LeakyReLU, padded particles zeroed), a class token
 * per jet is made from their sum and count (embbed_cls, gated by `cond`), then `layers` blocks in which the token attends
 * to the particles (nn.MultiheadAttention, one query, padded keys masked) and is broadcast back into the particle stream
 * (fc1 on cat(particle, token) + residual).  The head is Linear(hidden, 1): the field has ONE output per particle, which
 * the reference's loss and solver broadcast over the features -- `v_out` below is that broadcast, [n_jets][N][F].
 *
 * Built: local_cat_cond = global_cat_cond = False, net_config.global_cond_dim = 0 (the shipped yaml); t_local_cat / t_global_cat
 * either way (desc.t_cat: the yaml has them off, MDMA.__init__'s own defaults on, mdma.py:101-102): the time embedding is the same
 * for every particle of a jet, so behind a particle Linear (embed, Block.fc0) its columns are a per-jet bias row (mdma_time_kernel),
 * behind a class-token Linear (fc0_cls, fc1_cls, fc2_cls) extra rows of the per-jet GEMV; hidden a multiple of 128 (<= 512) with
 * head_dim = hidden / num_heads in {8, 16}; latent a multiple of 4 (<= 64).  `cond` [n_jets] is the conditional variant's
 * ONE value per jet (desc.c_cat; mdma.py:157-169 appends global_cond_in.unsqueeze(-1)); NULL otherwise (the network never reads it).
 *
 * Weight formats (float offsets into one blob, gathered from the state_dict by particle_fm_amd/layout_mdma.py):
 *   MFMA_AK / MFMA_AKT: as include/pfm_tf.h.  KMAJOR [K][NO]: element (k, o) at k * NO + o.
 * All device pointers are fp32; every launch goes to `stream`; no host synchronisation.  Return codes as pfm_hip.h.
 */
#ifndef PFM_MDMA_H
#define PFM_MDMA_H

#include <stdint.h>

#include "pfm_tf.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PFM_MDMA_ABI_VERSION 3
#define PFM_MDMA_MAX_LAYERS 16
#define PFM_MDMA_F_BF16 32u /* bf16 operands in the particle-stream Linears (forward and dX), see PFM_TF_F_BF16 (pfm_tf.h) */
#define PFM_MDMA_F_TEMB_SINCOS 2u /* t_emb = "sincos" (flow_matching_module.py:208-211) instead of "cosine" */
#define PFM_MDMA_F_TEMB_GIVEN 64u /* as PFM_TF_F_TEMB_GIVEN (pfm_tf.h; t_emb = "gaussian"): the caller supplies the time EMBEDDING through `t` --
                                   * forward / loss forward: temb[n_jets][t_dim] (per_jet_t = 1) or one row for all jets; pfm_mdma_sample_rk:
                                   * the transposed table [t_dim][n_steps * stages]; the loss backward then also accumulates d loss / d temb in
                                   * its scratch (pfm_mdma_backward_dtemb) */

typedef struct {
    pfm_tf_lin fc0;     /* Block.fc0 columns 0..H [H][H] MFMA_AK (+ WT); Wt = its time columns H..H+T KMAJOR [T][H] (t_local, else -1);
                         * Wc = its condition column KMAJOR [1][H] behind them (local_cat_cond, else -1) */
    pfm_tf_lin kv;      /* attn.in_proj rows H..3H (k | v) [2H][H] MFMA_AK (+ WT), bias in_proj_bias[H..3H] */
    pfm_tf_lin fc1;     /* Block.fc1: W = columns 0..H MFMA_AK (+ WT); Wc = its class-token columns KMAJOR [L][H]; Wt = its condition column
                         * KMAJOR [1][H] (local_cat_cond: column H, the token columns then start at H + 1; else -1); b */
    int64_t fc0c_W, fc0c_b; /* fc0_cls  KMAJOR [L (+ T: t_global)][H], [H] */
    int64_t ln_g, ln_b;     /* ln       [H] */
    int64_t q_W, q_b;       /* attn.in_proj rows 0..H: KMAJOR [H][H], [H] */
    int64_t o_W, o_b;       /* attn.out_proj KMAJOR [H][H], [H] */
    int64_t fc1c_W, fc1c_b; /* fc1_cls  KMAJOR [H + 1 (+ T: t_global)][L] (rows: attention output, particle count, time embedding), [L] */
    int64_t fc2c_W, fc2c_b; /* fc2_cls  KMAJOR [L (+ T: t_global)][L], [L] */
} pfm_mdma_block;

typedef struct {
    int32_t abi_version; /* PFM_MDMA_ABI_VERSION */
    int32_t n_points;    /* N */
    int32_t features;    /* F (<= 16) */
    int32_t hidden;      /* H */
    int32_t latent;      /* L */
    int32_t layers;
    int32_t heads, head_dim;
    int32_t t_dim;          /* 2 * frequencies of the CNF's time embedding (<= 64) */
    int32_t time_in_input;  /* add_time_to_input: the embedding Linear sees cat(temb, x) */
    uint32_t flags;
    int32_t t_cat;          /* bit 0: t_local_cat (the time embedding concatenated to the inputs of embed and of every Block.fc0), bit 1:
                             * t_global_cat (to the class-token Linears fc0_cls, fc1_cls, fc2_cls); mdma.py:56-59, 71-78, 155-156 */
    float neg_slope; /* nn.LeakyReLU() default 0.01 */
    float ln_eps;
    float avg_n;     /* MDMA.avg_n: the particle sum is divided by it */
    int32_t c_cat;   /* the conditional variant (one condition value c per jet, `cond` [n_jets]; mdma.py:60-63, 79-82, 157-174): bit 0
                      * net_config.global_cond_dim = 1 (c behind the particle count in the inputs of embbed_cls, cond and fc1_cls), bit 1
                      * global_cat_cond (cl appended to the inputs of fc0_cls and fc2_cls), bit 2 local_cat_cond (c appended to the inputs of
                      * embed and out, cl to those of fc0 and fc1); cl = cond[..., -1:] = c with bit 0 or 1, else the particle count */
    int64_t blob_floats;
    int64_t freqs;                  /* [t_dim] */
    int64_t emb_Wx, emb_Wt, emb_b;  /* MDMA.embed: KMAJOR [F][H], KMAJOR [t_dim][H] (-1 without time_in_input), [H] */
    int64_t emb_Wt2;                /* MDMA.embed's trailing time columns KMAJOR [t_dim][H] (t_local, else -1) */
    int64_t emb_Wc, out_Wc;         /* local_cat_cond: embed's condition column KMAJOR [1][H], out's condition weight [1] (else -1) */
    int64_t ecls_W, ecls_b;         /* embbed_cls KMAJOR [H + 1][L], [L] */
    int64_t cond_W, cond_b;         /* MDMA.cond  KMAJOR [1][L], [L] */
    int64_t out_W, out_b;           /* MDMA.out   [H], [1] */
    pfm_mdma_block block[PFM_MDMA_MAX_LAYERS];
} pfm_mdma_desc;

/* Workspace / backward scratch sizes in floats (train != 0: every block keeps its activations). */
int64_t pfm_mdma_workspace_floats(const pfm_mdma_desc *desc, int32_t n_jets, int32_t train);
int64_t pfm_mdma_backward_scratch_floats(const pfm_mdma_desc *desc, int32_t n_jets);

/* v_out[n_jets][N][F] = broadcast over F of MDMA(t, x, mask).  t: [n_jets] (per_jet_t != 0) or one shared value.
 * mask [n_jets][N] must be given (MDMA.forward indexes with it, mdma.py:151). */
int pfm_mdma_forward(const pfm_mdma_desc *desc, const float *blob, const float *t, int32_t per_jet_t, const float *x,
                     const float *cond, const float *mask, float *v_out, int32_t n_jets, float *workspace, void *stream);

/* Fixed-step explicit Runge-Kutta sampler (as pfm_tf_sample_rk): t_eval [n_steps * stages], dt [n_steps];
 * state: (2 + stages) * n_jets * N * F floats. */
int pfm_mdma_sample_rk(const pfm_mdma_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                       const float *dt, int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                       int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);

/* Loss forward with the draws given (kind 0 FM-OT: a = z; 1 CFM: a = x0, b = eps; 2 droid: a = z).
 * loss_sums[0] += sum (v - u)^2 over [n_jets][N][F], loss_sums[1] += sum mask. */
int pfm_mdma_fm_loss_forward(const pfm_mdma_desc *desc, const float *blob, int32_t kind, float sigma, const float *t,
                             const float *x, const float *a, const float *b, const float *cond, const float *mask, float *y_out,
                             float *u_out, float *v_out, float *loss_sums, int32_t n_jets, float *workspace, void *stream);

/* gblob[blob_floats] += d(gscale * loss_sums[0]) / d blob, from the workspace the forward left behind. */
int pfm_mdma_fm_loss_backward(const pfm_mdma_desc *desc, const float *blob, const float *mask, const float *y,
                              const float *u, const float *v, const float *gscale, float *gblob, int32_t n_jets,
                              float *workspace, float *scratch, void *stream);

/* PFM_MDMA_F_TEMB_GIVEN: dtemb[n_jets][t_dim] = d(loss)/d(temb) * gscale of the pfm_mdma_fm_loss_backward call that has just filled
 * `scratch` (same descriptor and n_jets): through the time columns of embed and -- t_cat -- of Block.fc0 / fc0_cls / fc1_cls / fc2_cls. */
int pfm_mdma_backward_dtemb(const pfm_mdma_desc *desc, const float *scratch, int32_t n_jets, float *dtemb, void *stream);

#ifdef __cplusplus
}
#endif
#endif
