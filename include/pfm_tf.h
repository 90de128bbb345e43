/*
 * pfm_tf.h -- C ABI of libpfm_hip.so, part 2: the Full-Transformer vector field (model "droid_fulltransformer").
 *
 * Same conventions as pfm_hip.h: plain device pointers owned by the caller, `stream` is a hipStream_t as void*,
 * every call returns 0 or a hipError_t / PFM_E_* code (text: pfm_last_error()), nothing allocates or synchronises.
 *
 * Reference interface each entry point replaces (paths relative to the particle_fm repository):
 *   pfm_tf_forward             CNF.forward(t, x, cond, mask) with model="droid_fulltransformer", t_emb="cosine"
 *                              particle_fm/models/flow_matching_module.py:191-233
 *                              -> FullTransformerEncoder.forward   models/components/droid_transformer.py:529-548
 *                              -> TransformerEncoder(.Layer)       droid_transformer.py:331-344, 433-437
 *                              -> MultiHeadedAttentionBlock        droid_transformer.py:231-284 (torch SDPA, kv mask :16-52)
 *                              -> DenseNetwork / MLPBlock          droid_transformer.py:793-813, 958-981
 *   pfm_tf_sample_midpoint     CNF.decode(z, cond, mask, ode_solver="midpoint", ode_steps)
 *                              flow_matching_module.py:245-259, 283-287, incl. `z * mask` of sample (:668-671)
 *   pfm_tf_fm_loss_forward /   FlowMatchingLoss.forward / ConditionalFlowMatchingLoss.forward and their autograd
 *   pfm_tf_fm_loss_backward    models/components/losses.py:38-77, 101-136
 *
 * Data layout.  Particles of all jets form ONE row-major matrix of M = n_jets * n_points rows ("rows"); every
 * row is computed (the reference leaves queries unmasked; only keys are masked).  Activations live in a caller-owned
 * workspace of pfm_tf_workspace_floats() floats; they pass through HBM/MALL between kernels (the per-jet tile
 * 279 x 256 fp32 does not fit the 160 KiB LDS), every Linear is an fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32) whose
 * prologue applies the preceding LayerNorm and whose epilogue applies bias, the per-jet context bias,
 * LeakyReLU(0.1) and the residual.
 *
 * Weight blob formats (fp32):
 *   MFMA_AK [NO][K]   NO multiple of 32, K multiple of 64: 16-output x 64-k blocks, block (ob, st) at float
 *                     ((ob * (K/64) + st) * 4 + kt) * 256 + lane * 4 + r  holds
 *                     W[16*ob + (lane&15)][64*st + 16*kt + 4*(lane>>4) + r]       (A operand of the forward GEMM)
 *   MFMA_AKT          the MFMA_AK packing of W^T ([K][NO] seen as outputs x k): A operand of the backward dX GEMM
 *   KMAJOR [K][NO]    row k = column k of the nn.Linear weight (per-jet GEMVs: context / time columns, ctxt_emdb)
 *   ROWMAJOR [F][K]   the nn.Linear weight as stored (output head, F <= 16)
 * The context columns of a Linear whose input is cat(x, ctxt) (MLPBlock, droid_transformer.py:802) and the time
 * columns of node_embd's input cat(temb, x) (flow_matching_module.py:199-200) multiply per-jet vectors, so they
 * are evaluated once per jet into a bias row ("jet bias") instead of being concatenated to every particle.
 */
#ifndef PFM_TF_H
#define PFM_TF_H

#include <stddef.h>
#include <stdint.h>
#include "pfm_hip.h" /* pfm_rk_tableau */

#ifdef __cplusplus
extern "C" {
#endif

#define PFM_TF_ABI_VERSION 1
#define PFM_TF_MAX_LAYERS 12
#define PFM_TF_F_VALID_ROWS 4 /* inference entry points (forward, samplers) with a mask: evaluate the VALID particles only --
                                 rows are compacted to the valid particles in (jet, particle) order, attention runs over each
                                 jet's own keys, padded rows of the state are left as they are (z * mask) and padded rows of a raw
                                 field are 0.  Valid rows see the same arithmetic as without the flag (padded keys never
                                 contribute); the reference's (unmasked, meaningless) values at padded rows are not produced.
                                 The loss entry points ignore the flag: the reference's loss includes the padded rows. */
#define PFM_TF_F_ONE_STREAM 16 /* pfm_tf_sample_midpoint stays on the caller's stream (no half-batch split): for callers that keep several
                                * sampler calls in flight themselves (generate_data's batch pipeline, bench_secondary --overlap) */
#define PFM_TF_F_TEMB_SINCOS 2 /* t_emb="sincos": temb = [cos(f t) ; sin(f t)], freqs table = [f ; f] (flow_matching_module.py:208-211) */
#define PFM_TF_F_BF16 32 /* desc.flags: every Linear (forward and dX; the dW GEMMs, LayerNorm, softmax and attention products stay fp32) on
                          * v_mfma_f32_16x16x32_bf16 with both operands rounded to bf16, fp32 accumulate, fp32 activations: what
                          * trainer.precision="bf16-mixed" (configs/trainer/default.yaml:11-12: autocast around the same modules) asks of
                          * the nn.Linear layers.  Bar: no further from the fp32 vectors than the oracle under torch.autocast(bfloat16) */
#define PFM_TF_F_TEMB_GIVEN 64 /* desc.flags: the caller supplies the time EMBEDDING (t_emb="gaussian": a small trainable network in front of
                                * the field, flow_matching_module.py:178-181, 213-221) through the `t` argument of every entry point:
                                *   forward / loss forward: t = temb[n_jets][t_dim] (t_stride = 1) or one row for all jets (t_stride = 0);
                                *   samplers: t_eval = the table [t_dim][evaluations], TRANSPOSED (evaluation e starts at t_eval + e);
                                *   the loss forward's interpolation must then not depend on t: kind "droid" with a = 0, i.e. y = x -- the
                                *   forward-with-saved-activations of particle_fm_amd/fm_field.py, whose backward starts from an upstream
                                *   gradient; pfm_tf_backward_dtemb returns d loss / d temb from that backward's scratch. */
#define PFM_TF_F_F16X3 1 /* desc.flags: every Linear (forward and dX) as three fp16 MFMAs on (hi, lo) splits of both operands,
                            fp32 accumulate: fp32-grade products (see PFM_F_F16X3_MFMA in pfm_hip.h); needs |x| < 65504 */

typedef struct { int64_t gamma, beta; } pfm_tf_norm; /* LayerNorm weight / bias, [dim] each (eps = desc.ln_eps) */

typedef struct {
    int64_t W;   /* main block (format depends on the layer, see pfm_tf_desc) */
    int64_t Wc;  /* context columns, KMAJOR [ctxt_dim][NO]; -1 if the layer has no context input */
    int64_t Wt;  /* time-embedding columns, KMAJOR [t_dim][NO]; -1 unless node_embd with add_time_to_input */
    int64_t b;   /* bias [NO] */
    int64_t WT;  /* MFMA_AKT copy for the backward dX product; -1 where no dX is needed */
} pfm_tf_lin;

typedef struct {
    pfm_tf_norm norm1;     /* te.layers.k.norm1 */
    pfm_tf_lin qkv;        /* self_attn.all_linear  [3D][D] MFMA_AK; outputs ordered q | k | v, head h = columns 16h.. */
    pfm_tf_norm attn_norm; /* self_attn.layer_norm (do_layer_norm) ; gamma = -1 if absent */
    pfm_tf_lin out;        /* self_attn.out_linear  [D][D] MFMA_AK */
    pfm_tf_norm norm2;     /* te.layers.k.norm2 */
    pfm_tf_lin d1;         /* dense.input_block.block.0  [hidden][D (+ctxt)] : W MFMA_AK over the D columns, Wc */
    pfm_tf_norm d_norm;    /* dense.input_block.block.2 */
    pfm_tf_lin d2;         /* dense.output_block.block.0 [D][hidden] MFMA_AK */
} pfm_tf_layer;

typedef struct {
    int32_t abi_version;  /* PFM_TF_ABI_VERSION */
    int32_t n_points;     /* N: particles per jet */
    int32_t features;     /* F: particle features in and out (<= 16) */
    int32_t model_dim;    /* D: multiple of 128, <= 512 */
    int32_t hidden;       /* hddn_dim of node_embd / dense / outp_embd (the reference defaults all to 2D): multiple of 128, <= 512 */
    int32_t layers;
    int32_t heads;        /* model_dim / 16 (head_dim is 16 in this build) */
    int32_t head_dim;
    int32_t t_dim;        /* 2 * frequencies (<= 64) */
    int32_t cond_dim;     /* global_cond_dim (<= 16) */
    int32_t ctxt_dim;     /* ctxt_emdb outp_dim (<= 64, multiple of 4) */
    int32_t ctxt_hidden;  /* ctxt_emdb hddn_dim (<= 512, multiple of 4) */
    int32_t time_in_input; /* add_time_to_input */
    int32_t flags;
    float neg_slope;      /* 0.1 ("lrlu", droid_transformer.py:1022) */
    float ln_eps;         /* 1e-5 */
    int64_t blob_floats;
    int64_t freqs;        /* [t_dim] cosine-embedding frequency table */
    pfm_tf_lin c1;        /* ctxt_emdb.input_block : KMAJOR [t_dim + cond_dim][ctxt_hidden] */
    pfm_tf_norm c_norm;
    pfm_tf_lin c2;        /* ctxt_emdb.output_block: KMAJOR [ctxt_hidden][ctxt_dim] */
    pfm_tf_lin n1;        /* node_embd.input_block : W = KMAJOR [F][hidden] (particle columns), Wt, Wc */
    pfm_tf_norm n_norm;
    pfm_tf_lin n2;        /* node_embd.output_block: [D][hidden] MFMA_AK */
    pfm_tf_layer layer[PFM_TF_MAX_LAYERS];
    pfm_tf_norm final_norm;
    pfm_tf_lin o1;        /* outp_embd.input_block : [hidden][D (+ctxt)] MFMA_AK + Wc */
    pfm_tf_norm o_norm;
    pfm_tf_lin o2;        /* outp_embd.output_block: ROWMAJOR [F][hidden] */
} pfm_tf_desc;

/* Workspace size in floats for n_jets jets.  `train` != 0: every layer keeps its own activations (what
 * pfm_tf_fm_loss_backward re-reads); 0: layers share one set (and the size covers the two half-batch workspaces of
 * pfm_tf_sample_midpoint). */
int64_t pfm_tf_workspace_floats(const pfm_tf_desc *desc, int32_t n_jets, int32_t train);

/* v[n_jets][N][F] = f(t, x).  t: one time per jet (t_stride 1) or a single time for all jets (t_stride 0, the
 * 0-dim t of sampling).  cond [n_jets][cond_dim] (NULL iff cond_dim == 0).  mask [n_jets][N] fp32 {0,1} key mask,
 * NULL = all valid. */
int pfm_tf_forward(const pfm_tf_desc *desc, const float *blob, const float *t, int32_t t_stride, const float *x,
                   const float *cond, const float *mask, float *v, int32_t n_jets, float *workspace, void *stream);

/* Fixed-step midpoint over the 2*(ode_steps-1) times t_eval / ode_steps-1 steps dt (see pfm_epic_sample_midpoint).
 * x_out may alias z.  premask != 0 multiplies z by the mask first.  state: 2 * n_jets*N*F floats of scratch.
 * Calls on >= 64 jets run as two half-batches on two internal streams that fork from and join `stream` (same bits:
 * every kernel is row- or jet-local) unless desc.flags has PFM_TF_F_ONE_STREAM. */
int pfm_tf_sample_midpoint(const pfm_tf_desc *desc, const float *blob, const float *t_eval, const float *dt,
                           int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                           int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);

/* Fixed-step explicit Runge-Kutta (pfm_rk_tableau, pfm_hip.h): t_eval[n_steps * stages], dt[n_steps];
 * state: (2 + stages) * n_jets*N*F floats of scratch.  ode_solver "euler" / "rk4" of CNF.decode and the rk4 of CNF.encode
 * (flow_matching_module.py:235-243, 261-282).  Splits into two half-batches on two internal streams like pfm_tf_sample_midpoint. */
int pfm_tf_sample_rk(const pfm_tf_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                     const float *dt, int32_t n_steps, const float *z, const float *cond, const float *mask, float *x_out,
                     int32_t n_jets, int32_t premask, float *state, float *workspace, void *stream);

/* Loss forward: kind 0 = FM-OT (a = z), 1 = CFM (a = x0, b = eps), 2 = droid (DroidLoss, losses.py:304-342: y = x + t z, u = z mask).  Builds y, u; evaluates v = f(t, y) keeping
 * the activations in `workspace` (train layout); loss_sums[0] += sum (v-u)^2, loss_sums[1] += sum mask
 * (the caller zeroes loss_sums and divides).  y_out/u_out/v_out [n_jets][N][F] are written for the backward. */
int pfm_tf_fm_loss_forward(const pfm_tf_desc *desc, const float *blob, int32_t kind, float sigma, const float *t,
                           const float *x, const float *a, const float *b, const float *cond, const float *mask,
                           float *y_out, float *u_out, float *v_out, float *loss_sums, int32_t n_jets,
                           float *workspace, void *stream);

/* Loss backward: gblob (desc.blob_floats floats, zeroed by the caller) += dLoss/d(blob entry) * gscale for
 * every entry a parameter maps to (MFMA_AK blocks receive their gradient in the same MFMA_AK order).
 * gscale = grad_output / sum(mask).  scratch: pfm_tf_backward_scratch_floats() floats. */
int64_t pfm_tf_backward_scratch_floats(const pfm_tf_desc *desc, int32_t n_jets);
int pfm_tf_fm_loss_backward(const pfm_tf_desc *desc, const float *blob, const float *t, const float *cond,
                            const float *mask, const float *y, const float *u, const float *v, const float *gscale,
                            float *gblob, int32_t n_jets, float *workspace, float *scratch, void *stream);
/* the same, and grad_y[n_jets][N][F] = d(loss)/d(y) * gscale: the gradient w.r.t. the network's particle input through node_embd's particle
 * columns -- what a chain of flows needs (n_transforms > 1, flow_matching_module.py:421-443; losses.py:66-69 feeds each flow's output to the
 * next); see pfm_epic_fm_loss_backward_dx in pfm_hip.h */
int pfm_tf_fm_loss_backward_dx(const pfm_tf_desc *desc, const float *blob, const float *cond, const float *mask, const float *y,
                               const float *u, const float *v, const float *gscale, float *gblob, float *grad_y, int32_t n_jets,
                               float *workspace, float *scratch, void *stream);

/* PFM_TF_F_TEMB_GIVEN: dtemb[n_jets][t_dim] = d(loss)/d(temb) * gscale of the pfm_tf_fm_loss_backward call that has just filled
 * `scratch` (same descriptor, blob and n_jets): through the context network's first Linear and the time columns of node_embd. */
int pfm_tf_backward_dtemb(const pfm_tf_desc *desc, const float *blob, const float *scratch, int32_t n_jets, float *dtemb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFM_TF_H */
