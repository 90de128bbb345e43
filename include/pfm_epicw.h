/*
 * pfm_epicw.h -- C ABI of libpfm_hip.so, part 3: the EPiC vector field at widths the jet-resident kernel of
 * pfm_hip.h cannot hold in LDS (hidden_dim != 128, e.g. the JetClass configuration: hidden 300, latent 16,
 * 20 layers, 13 features, 12 conditioning values; configs/experiment/jetclass_cond.yaml:32-42).
 *
 * Same conventions as pfm_hip.h / pfm_tf.h.  Reference interface replaced:
 *   pfm_ew_forward           CNF.forward with model="epic"      flow_matching_module.py:191-233
 *                            -> EPiC_encoder.forward / EPiC_layer.forward   components/epic.py:304-391, 85-203
 *   pfm_ew_sample_midpoint   CNF.decode(ode_solver="midpoint")  flow_matching_module.py:245-259, 283-287, 668-671
 *
 * Data layout.  Particles of all jets form one row matrix (M = n_jets * n_points rows) of Hp = hidden rounded up
 * to a multiple of 64 columns (padding columns are exactly 0 everywhere: zero weight rows / columns, zero bias);
 * per-jet quantities live in P[n_jets][256 + Hp] = [temb | cond | 0 .. (128) ; g | 0 .. (128) ; g1 (Hp)] and
 * Q[n_jets][2 Hp] = [masked mean | masked sum * sum_scale].  Every Linear is the fp32-MFMA GEMM of pfm_tf.h
 * (MFMA_AK weights); the columns that multiply per-jet vectors (time, conditioning, broadcast global vector) are
 * evaluated once per jet by small GEMMs over the P rows into "jet bias" rows.
 *   fc_local1 / fc_local2 / fc_l2   [Hp][Hp]           particle block
 *   jb  (per layer)                 [2 Hp][256]        rows: fc_local1 extras | fc_local2 extras, cols: the first 256 of P; bias = the layers' biases
 *   sjb (once per evaluation)       [2 Hp + 128][256]  rows: fc_l1 extras | fc_l2 extras | fc_l3 extras (F rows)
 *   fc_global1 / fc_g1              [Hp][256 + 2 Hp]   cols: first 256 of P, then Q (the reference's (sum, mean) / (mean, sum) order is a column permutation)
 *   fc_global2 / fc_g2              [128][256 + Hp]    cols: P; rows 0..L-1
 */
#ifndef PFM_EPICW_H
#define PFM_EPICW_H

#include <stddef.h>
#include <stdint.h>
#include "pfm_hip.h" /* pfm_rk_tableau */

#ifdef __cplusplus
extern "C" {
#endif

#define PFM_EW_ABI_VERSION 1
#define PFM_EW_MAX_LAYERS 24
#define PFM_EW_F_TEMB_SINCOS 2 /* as PFM_TF_F_TEMB_SINCOS */
#define PFM_EW_F_F16X3 1 /* desc.flags: as PFM_TF_F_F16X3 (pfm_tf.h) */
#define PFM_EW_F_BF16 32 /* desc.flags: as PFM_TF_F_BF16 (pfm_tf.h): the particle Linears (forward and dX) on bf16 operands; the per-jet chain,
                          * the pooling and the dW GEMMs stay fp32 */
#define PFM_EW_F_TEMB_GIVEN 64 /* desc.flags: as PFM_TF_F_TEMB_GIVEN (pfm_tf.h): the caller supplies the time EMBEDDING rows through `t`
                                * (forward / loss forward: temb[n_jets][t_dim]; samplers: the transposed table [t_dim][evaluations]); the
                                * loss backward then also accumulates d loss / d temb in its scratch (pfm_ew_backward_dtemb) */

typedef struct { int64_t W, b, WT; } pfm_ew_lin; /* MFMA_AK weights, bias (-1: none), MFMA_AKT copy (-1: none) */

typedef struct { pfm_ew_lin g1, g2, jb, l1, l2; } pfm_ew_layer;

typedef struct {
    int32_t abi_version;
    int32_t n_points, features, hidden, hidden_pad, latent, layers, t_dim, cond_global, cond_local, flags, pad_;
    float sum_scale, neg_slope;
    int64_t blob_floats;
    int64_t freqs;  /* [t_dim] */
    int64_t l1x;    /* fc_l1 particle columns, KMAJOR [F][Hp] */
    int64_t l3;     /* fc_l3 particle block, ROWMAJOR [16][Hp] (rows >= F zero) */
    pfm_ew_lin sjb, l2, sg1, sg2;
    pfm_ew_layer layer[PFM_EW_MAX_LAYERS];
} pfm_ew_desc;

int64_t pfm_ew_workspace_floats(const pfm_ew_desc *desc, int32_t n_jets, int32_t train);

/* v[n_jets][N][F] = f(t, x); t_stride 1: one time per jet, 0: one time for all.  mask [n_jets][N] fp32 or NULL. */
int pfm_ew_forward(const pfm_ew_desc *desc, const float *blob, const float *t, int32_t t_stride, const float *x,
                   const float *cond, const float *mask, float *v, int32_t n_jets, float *workspace, void *stream);

/* see pfm_tf_sample_midpoint, except that a call stays on `stream` (no half-batch split) and, from the second step on, replays its
 * captured step body as a hipGraph (the host cannot enqueue ~210 launches per step fast enough); PFM_EW_GRAPH=0 in the environment:
 * direct launches (diagnostics) */
int pfm_ew_sample_midpoint(const pfm_ew_desc *desc, const float *blob, const float *t_eval, const float *dt,
                           int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                           int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);

/* see pfm_tf_sample_rk */
int pfm_ew_sample_rk(const pfm_ew_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                     const float *dt, int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                     int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);

/* Loss forward / backward, as pfm_tf_fm_loss_forward / pfm_tf_fm_loss_backward (losses.py:38-77, 101-136): the forward
 * keeps every stage's activations in `workspace` (train layout); the backward adds d loss / d(blob entry) * gscale
 * into gblob (zeroed by the caller; MFMA_AK blocks in MFMA_AK order; padding slots receive values nobody reads). */
int pfm_ew_fm_loss_forward(const pfm_ew_desc *desc, const float *blob, int32_t kind, float sigma, const float *t,
                           const float *x, const float *a, const float *b, const float *cond, const float *mask,
                           float *y_out, float *u_out, float *v_out, float *loss_sums, int32_t n_jets,
                           float *workspace, void *stream);
int64_t pfm_ew_backward_scratch_floats(const pfm_ew_desc *desc, int32_t n_jets);
int pfm_ew_fm_loss_backward(const pfm_ew_desc *desc, const float *blob, const float *mask, const float *y,
                            const float *u, const float *v, const float *gscale, float *gblob, int32_t n_jets,
                            float *workspace, float *scratch, void *stream);

/* pfm_ew_fm_loss_backward that also returns grad_y[n_jets][N][F] = d(loss)/d(y) * gscale, the gradient w.r.t. the network's particle input
 * (through fc_l1's particle columns): what a chain of flows needs (n_transforms > 1, flow_matching_module.py:421-443; losses.py:66-69 feeds
 * each flow's output to the next); see pfm_epic_fm_loss_backward_dx in pfm_hip.h. */
int pfm_ew_fm_loss_backward_dx(const pfm_ew_desc *desc, const float *blob, const float *mask, const float *y, const float *u,
                               const float *v, const float *gscale, float *gblob, float *grad_y, int32_t n_jets, float *workspace,
                               float *scratch, void *stream);

/* PFM_EW_F_TEMB_GIVEN: dtemb[n_jets][t_dim] = d(loss)/d(temb) * gscale of the pfm_ew_fm_loss_backward call that has just filled `scratch`
 * (same descriptor and n_jets): the time columns of every per-jet Linear (fc_l1 / fc_l2 / fc_l3 biases, fc_g1 / fc_g2 and, per layer,
 * fc_global1 / fc_global2 / fc_local1 / fc_local2), summed in launch order. */
int pfm_ew_backward_dtemb(const pfm_ew_desc *desc, const float *scratch, int32_t n_jets, float *dtemb, void *stream);

/* loss_type="diffusion" on this path (DiffusionLoss, models/components/losses.py:207-290; see pfm_epic_diffusion_loss_* in pfm_hip.h):
 * noisy = rates[b][0] x + rates[b][1] z, the field predicts z; loss_sums[0] = sum_b jet_weight[b] sum_n,f criterion(v - z),
 * loss_sums[1] = sum mask; criterion 0 = mse, 1 = huber (delta 1).  The backward is that of loss_sums[0] * gscale. */
int pfm_ew_diffusion_loss_forward(const pfm_ew_desc *desc, const float *blob, int32_t criterion, const float *rates,
                                  const float *jet_weight, const float *t, const float *x, const float *z, const float *cond,
                                  const float *mask, float *y_out, float *u_out, float *v_out, float *loss_sums,
                                  int32_t n_jets, float *workspace, void *stream);
int pfm_ew_diffusion_loss_backward(const pfm_ew_desc *desc, const float *blob, int32_t criterion, const float *jet_weight,
                                   const float *mask, const float *y, const float *u, const float *v, const float *gscale,
                                   float *gblob, int32_t n_jets, float *workspace, float *scratch, void *stream);
/* pfm_ew_sample_rk on the probability-flow ODE of a noise-predicting network: rhs[n_steps * stages][2] = (-0.5 beta, noise_rate) at
 * every stage time, the right-hand side is rhs0 (x - f(t, x) / rhs1) (ode_wrapper.forward, flow_matching_module.py:62-69). */
int pfm_ew_sample_rk_rhs(const pfm_ew_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                         const float *dt, int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                         int32_t n_jets, int32_t premask, float *state, float *workspace, const float *rhs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFM_EPICW_H */
