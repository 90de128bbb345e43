/*
 * pfm_ca.h -- C ABI of libpfm_hip.so, part 4: the cross-attention vector field (model "droid_fullcrossattention",
 * configs/model/fm_droid_crossattention.yaml).  Conventions as pfm_tf.h (shared types pfm_tf_norm / pfm_tf_lin).
 *
 * Reference interface replaced:
 *   pfm_ca_forward / pfm_ca_sample_midpoint / pfm_ca_fm_loss_forward / pfm_ca_fm_loss_backward
 *       CNF.forward / decode("midpoint") / losses with model="droid_fullcrossattention"
 *       -> FullCrossAttentionEncoder.forward        models/components/droid_transformer.py:685-711
 *       -> CrossAttentionEncoder.forward            droid_transformer.py:442-472 (global tokens <- sequence, sequence <- tokens)
 *       -> TransformerCrossAttentionLayer.forward   droid_transformer.py:380-397
 *       -> MultiHeadedAttentionBlock (q_linear / k_linear / v_linear)   droid_transformer.py:231-284
 *
 * Two row matrices: the particles (M = n_jets * n_points rows) and the global tokens (n_jets * tokens rows).  Linears,
 * LayerNorms and the context path are the kernels of pfm_tf.h; k_linear and v_linear of a layer are one GEMM ([2D][D]:
 * k rows, then v rows).  The two attention shapes -- a handful of token queries over all particles (key mask), and every
 * particle over a handful of tokens -- are small VALU kernels (one workgroup per jet), not MFMA: 2 * 2 * N * tokens * D
 * FLOP per jet and layer is < 1 % of the Linears.
 */
#ifndef PFM_CA_H
#define PFM_CA_H

#include "pfm_tf.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PFM_CA_ABI_VERSION 1
#define PFM_CA_MAX_LAYERS 16 /* from/to layer pairs */
#define PFM_CA_MAX_TOKENS 8
#define PFM_CA_F_F16X3 1        /* split-fp16 Linears, see PFM_TF_F_F16X3 */
#define PFM_CA_F_BF16 32        /* bf16 operands in the particle-side Linears (forward and dX), see PFM_TF_F_BF16; tokens, attention, dW: fp32 */
#define PFM_CA_F_TEMB_GIVEN 64  /* the `t` arguments hold the time EMBEDDING, see PFM_TF_F_TEMB_GIVEN (pfm_tf.h); no graph replay then */
#define PFM_CA_F_TEMB_SINCOS 2  /* see PFM_TF_F_TEMB_SINCOS */
#define PFM_CA_F_VALID_ROWS 4    /* see PFM_TF_F_VALID_ROWS: inference over the valid particles only */
#define PFM_CA_F_GRAPH_STEPS 8   /* pfm_ca_sample_midpoint on a non-null stream: step 0 is launched directly, the step body is captured
                                  * once (t / dt behind a device-side step counter) and replayed as a hipGraph for the other steps --
                                  * same kernels, same results; the graph is kept in a per-stream slot and released by the next such call
                                  * on that stream.  For callers that keep several sampler calls in flight from one thread and are
                                  * bound by the ~11 us the host needs per launch (~200 launches per step) */

typedef struct {
    pfm_tf_norm norm0, norm1, norm2, attn_norm, d_norm; /* norm0: keys/values input, norm1: query input, norm2: dense input */
    pfm_tf_lin q;   /* cross_attn.q_linear [D][D] MFMA_AK */
    pfm_tf_lin kv;  /* cross_attn.k_linear ; v_linear stacked: [2D][D] MFMA_AK, bias [2D] */
    pfm_tf_lin out; /* cross_attn.out_linear [D][D] */
    pfm_tf_lin d1;  /* dense.input_block [hidden][D (+ctxt)]: W over the D columns, Wc */
    pfm_tf_lin d2;  /* dense.output_block [D][hidden] */
} pfm_ca_layer;

typedef struct {
    int32_t abi_version, n_points, features, model_dim, hidden, layers, heads, head_dim, tokens, t_dim, cond_dim, ctxt_dim,
        ctxt_hidden, time_in_input, flags, pad_;
    float neg_slope, ln_eps;
    int64_t blob_floats, freqs;
    int64_t global_tokens; /* cae.global_tokens [tokens][D] */
    pfm_tf_lin c1; pfm_tf_norm c_norm; pfm_tf_lin c2;  /* ctxt_emdb (KMAJOR), as pfm_tf_desc */
    pfm_tf_lin n1; pfm_tf_norm n_norm; pfm_tf_lin n2;  /* node_embd */
    pfm_ca_layer from_layer[PFM_CA_MAX_LAYERS];          /* queries = tokens, keys = particles (masked) */
    pfm_ca_layer to_layer[PFM_CA_MAX_LAYERS];            /* queries = particles, keys = tokens */
    pfm_tf_lin o1; pfm_tf_norm o_norm; pfm_tf_lin o2;  /* outp_embd */
} pfm_ca_desc;

int64_t pfm_ca_workspace_floats(const pfm_ca_desc *desc, int32_t n_jets, int32_t train);
int pfm_ca_forward(const pfm_ca_desc *desc, const float *blob, const float *t, int32_t t_stride, const float *x,
                   const float *cond, const float *mask, float *v, int32_t n_jets, float *workspace, void *stream);
int pfm_ca_sample_midpoint(const pfm_ca_desc *desc, const float *blob, const float *t_eval, const float *dt,
                           int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                           int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);
/* see pfm_tf_sample_rk */
int pfm_ca_sample_rk(const pfm_ca_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                     const float *dt, int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                     int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);
int pfm_ca_fm_loss_forward(const pfm_ca_desc *desc, const float *blob, int32_t kind, float sigma, const float *t,
                           const float *x, const float *a, const float *b, const float *cond, const float *mask,
                           float *y_out, float *u_out, float *v_out, float *loss_sums, int32_t n_jets,
                           float *workspace, void *stream);
int64_t pfm_ca_backward_scratch_floats(const pfm_ca_desc *desc, int32_t n_jets);
int pfm_ca_fm_loss_backward(const pfm_ca_desc *desc, const float *blob, const float *cond, const float *mask,
                            const float *y, const float *u, const float *v, const float *gscale, float *gblob,
                            int32_t n_jets, float *workspace, float *scratch, void *stream);
/* the same, and grad_y[n_jets][N][F] = d(loss)/d(y) * gscale (see pfm_tf_fm_loss_backward_dx) */
int pfm_ca_fm_loss_backward_dx(const pfm_ca_desc *desc, const float *blob, const float *cond, const float *mask, const float *y,
                               const float *u, const float *v, const float *gscale, float *gblob, float *grad_y, int32_t n_jets,
                               float *workspace, float *scratch, void *stream);

/* PFM_CA_F_TEMB_GIVEN: dtemb[n_jets][t_dim] = d(loss)/d(temb) * gscale of the pfm_ca_fm_loss_backward call that has just filled
 * `scratch` (as pfm_tf_backward_dtemb). */
int pfm_ca_backward_dtemb(const pfm_ca_desc *desc, const float *blob, const float *scratch, int32_t n_jets, float *dtemb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFM_CA_H */
