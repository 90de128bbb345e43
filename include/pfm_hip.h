/*
 * pfm_hip.h -- C ABI of libpfm_hip.so: the MI355X (gfx950) flow-matching hot path of particle_fm.
 *
 * Plain pointers and sizes only; every pointer except `desc` is a DEVICE pointer owned by the
 * caller, `stream` is a hipStream_t passed as void*.  Every entry point returns 0 on success or a
 * hipError_t / PFM_E_* code; pfm_last_error() gives the text.  Nothing here allocates or
 * synchronises, so calls may be captured into a hipGraph.
 *
 * Reference interface each entry point replaces (paths relative to the particle_fm repository):
 *   pfm_epic_forward            CNF.forward(t, x, cond, mask) with model="epic", t_emb="cosine"
 *                               particle_fm/models/flow_matching_module.py:191-233
 *                               -> EPiC_encoder.forward   models/components/epic.py:304-391
 *                               -> EPiC_layer.forward     models/components/epic.py:85-203
 *                               -> cosine_encoding        models/components/time_emb.py:49-96
 *   pfm_epic_sample_midpoint    CNF.decode(z, cond, mask, ode_solver="midpoint", ode_steps)
 *                               flow_matching_module.py:245-259, 283-287 (torchdyn fixed-step midpoint)
 *                               incl. the `z * mask` of SetFlowMatchingLitModule.sample (:668-671)
 *   pfm_epic_sample_rk          CNF.decode(..., ode_solver="euler" | "rk4" | "midpoint") and CNF.encode (rk4, t: 0 -> 1)
 *                               flow_matching_module.py:235-243, 261-287 (torchdyn fixed-step solvers; tableau given by the caller)
 *   pfm_epic_fm_loss_forward /  FlowMatchingLoss.forward / ConditionalFlowMatchingLoss.forward
 *   pfm_epic_fm_loss_backward   models/components/losses.py:38-77, 101-136 and their autograd
 *   pfm_epic_diffusion_loss_*   DiffusionLoss.forward and its autograd   models/components/losses.py:207-290
 *   pfm_diffusion_update        the state update of ddim_sampler / euler_maruyama_sampler   models/components/solver.py:81-93, 126-132
 *   pfm_norm_update / _apply    IterativeNormLayer.update / fit / forward / reverse   models/components/norm_layer.py:98-152
 *   pfm_sample_epilogue         the per-batch post-processing of generate_data   utils/data_generation.py:94-123
 *   pfm_optim_step              clip_grad_norm_(gradient_clip_val) + AdamW + EMA
 *                               configs/experiment/jetnet/fm_tops150.yaml:24, configs/model/flow_matching.yaml:3-7,
 *                               particle_fm/callbacks/ema.py:73-81
 *
 * Weights travel as ONE fp32 "blob": the effective (weight-normalised) matrices W = g*v/||v||
 * re-ordered for the kernels.  The host builds it from the reference's state_dict
 * (weight_g / weight_v / bias per Linear) -- particle_fm_amd/layout.py -- and fills the offsets below.
 *
 * Blob formats (H = hidden = 128, T = time-embedding width, C = cond width, L = latent):
 *   KMAJOR[K][OUT]   row k holds column k of the nn.Linear weight for all OUT outputs (plain, small blocks).
 *   KM16             K-major block with OUT = 128, rows zero-padded to a multiple of 16 and stored in 16-row panels so
 *                    that thread t of 512 reads float4 number t of a panel: element (k, o) lives at float
 *                    ((k>>4)*32 + (o>>2))*64 + (k&15)*4 + (o&3).  Used for every per-jet GEMV with 128 outputs.
 *   KP16             [K16][16]: K-major with the OUT <= 16 outputs zero-padded to 16 columns, rows to a multiple of 16.
 *   KQ16             (round 4; the lean sampler's per-jet chains) OUT = 128, 16-row panels of 2048 floats, element (k, o) of panel
 *                    p = k >> 4 at float  p*2048 + o*16 + (k&15): thread t of 512 reads float4 number t of a panel and holds FOUR
 *                    CONSECUTIVE k of ONE output, W[o = t>>2][16p + 4(t&3) .. + 3] -- a thread's partial sum needs one add and a
 *                    4-lane (quad) reduction instead of a 16-lane tree over four accumulators.
 *   CH16             (round 4; bf16 descriptors) the per-jet blocks once more as bf16 A operands of v_mfma_f32_16x16x32_bf16 for the
 *                    lean bf16 sampler's chains, which run on the matrix pipe with the workgroup's JETS as the B operand's columns:
 *                    16-byte unit ((w*NK + kt)*64 + lane) holds W[16w + (lane&15)][32kt + 8(lane>>4) + e], e = 0..7 (rows >= K zero),
 *                    w = output slice (8 for OUT = 128, 1 for OUT <= 16), NK = ceil(K / 32).  Filled from the KQ16 / WQ16 copies by
 *                    pfm_epic_pack_a16 (round-to-nearest-even).
 *   WQ16             the same idea for OUT <= 16, K = 128: float (k>>4)*256 + o*16 + (k&15); wave w reads chunk w, lane l float4
 *                    number l = W[o = l>>2][16w + 4(l&3) .. + 3]  (outputs >= OUT are zero).
 *   MFMA_A           the H x H block that multiplies the per-particle activations, pre-arranged as the
 *                    A operand of v_mfma_f32_16x16x4_f32: float4 at ((w*8 + kt)*64 + lane) holds
 *                    W[16*w + (lane&15)][16*kt + 4*(lane>>4) + r], r = 0..3   (w = output slice 0..7).
 *   GRAD_D           (gradient blob only) a 128x128 block in the order of a 16-row-panel accumulator: float ((w*8 + it)*4 + r)*64 + lane
 *                    holds dW[16*w + 4*(lane>>4) + r][8*(lane&15) + it].
 *   MFMA_A16         the same block rounded to bf16 (round-to-nearest-even), the A operand of v_mfma_f32_16x16x32_bf16 in the kernels'
 *                    k order: the 16-byte unit ((w*4 + kt2)*64 + lane) holds the MFMA_A float4s of k-tiles 2*kt2 and 2*kt2 + 1 of that
 *                    (w, lane) as 8 bf16, i.e. W[16*w + (lane&15)][32*kt2 + 16*h + 4*(lane>>4) + r] at bf16 index 4*h + r.  H*H/2 floats.
 *                    Not gathered from the state_dict: pfm_epic_pack_a16 (device blobs) / layout.finish_blob (host blobs) derive it
 *                    from MFMA_A; only blobs of PFM_F_BF16_MFMA descriptors carry it (zeros otherwise).  Read by the lean bf16 sampler.
 *   MFMA_AT          the same block transposed (used by the backward dX products):
 *                    float4 at ((w*8 + kt)*64 + lane) holds W[16*kt + 4*(lane>>4) + r][16*w + (lane&15)].
 * The blob carries a copy of the descriptor in its tail: element blob[desc.blob_floats] starts
 * PFM_DESC_FLOATS floats holding the bytes of the pfm_epic_desc itself, so a device buffer of
 * desc.blob_floats + PFM_DESC_FLOATS floats is what every entry point expects (the kernels read
 * the offsets from there instead of a 2.5 KB kernel argument).
 * Column order of the "extras" (per-jet) inputs is [temb(T) ; cond(C) ; g(L)] (g only for fc_local1),
 * of the global MLP inputs [temb(T) ; cond(C) ; mean(H) ; sum*scale(H) ; g(L)].
 */
#ifndef PFM_HIP_H
#define PFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): the loss forward / backward entry points take a jet launch order; pfm_local_lin.A16 / pfm_epic_desc.l3_A16 (bf16 copies of the particle blocks in the blob); pfm_epic_fm_loss_backward / pfm_epic_diffusion_loss_backward take a `scratch` pointer in front of `stream` and WRITE
 * grad_blob (round 2 changed both under version 1: a caller or a stale library built against that header must be refused) */
/* 3 (round 4): pfm_epic_desc carries the KQ16 / WQ16 copies of the per-jet GEMV blocks (q_*) */
#define PFM_ABI_VERSION 3
#define PFM_MAX_LAYERS 24
#define PFM_HIDDEN 128

#define PFM_E_BADARG 10001   /* descriptor / shape the kernels are not built for */
#define PFM_E_LDS 10002      /* set does not fit the 160 KiB LDS tile */

/* flags */
#define PFM_F_SKIP_MASKED_TAIL 1u /* do not compute particle tiles that lie wholly behind the last valid particle */
#define PFM_F_PACK_JETS 16u /* pfm_epic_sample_midpoint (fp32 / bf16 kernels, with SKIP_MASKED_TAIL and a mask): two short jets may share a
                               workgroup -- rows one behind the other in the LDS tile, ONE weight stream and ONE set of phases for both
                               (the k-th longest jet takes the shortest remaining one if pad16(rows A) + rows B fit).  Same results
                               bit for bit; pays off for large batches of short jets, see DESIGN.md */
#define PFM_F_QUAD_JETS 64u /* pfm_epic_sample_midpoint with PFM_F_PACK_JETS on a descriptor of n_points = 128 (unconditioned jets, a mask, fp32 / bf16
                                operands): FOUR jets per workgroup in fixed 32-row slots.  The caller states that every jet's valid particles lie
                                in its first 32 rows -- particles behind row 32 are IGNORED (treated as masked) and come back as 0.  The host
                                side sets it when it pads a batch of sets of <= 32 particles to the 128-row tile (hip_ops.packed_tile_rows).
                                Same results bit for bit as one jet per workgroup */
#define PFM_F_F16X3_MFMA 4u /* inference kernels only: the particle Linears as three v_mfma_f32_16x16x16_f16 on (hi, lo) fp16 splits of
                               both operands -- fp32-grade products (error 2^-22) at 2.7x less matrix-pipe time; needs |x| < 65504 */
#define PFM_F_TEMB_SINCOS 8u /* t_emb="sincos" (flow_matching_module.py:208-211): temb = [cos(f t) ; sin(f t)], freqs table = [f ; f],
                               f = 2^k pi; default: t_emb="cosine" (time_emb.py:79-96) */
#define PFM_F_GENERIC_SAMPLER 32u /* pfm_epic_sample_midpoint: keep the generic kernel where the lean evaluation (csrc/epic_fast.h: unconditioned
                                     jets, T = 32, F <= 4) would run; results differ by fp32 re-association only */
#define PFM_F_BF16_MFMA 2u /* inference kernels, loss forward and the dX products of the backward: the 128x128 particle Linears run on v_mfma_f32_16x16x32_bf16 (operands
                              rounded to bf16 on the fly, fp32 accumulate, fp32 activations); everything else stays fp32 */

typedef struct pfm_local_lin {
    int64_t A;  /* MFMA_A  block, H*H floats */
    int64_t AT; /* MFMA_AT block, H*H floats (or -1 if the blob carries no backward copies) */
    int64_t We; /* KM16 [Ke][H] extras block */
    int64_t b;  /* [H] bias */
    int64_t A16; /* MFMA_A16 block, H*H/2 floats */
} pfm_local_lin;

typedef struct pfm_dense_lin {
    int64_t W; /* KM16 (OUT = 128) or KP16 (OUT = latent) */
    int64_t b; /* [OUT] */
} pfm_dense_lin;

typedef struct pfm_epic_layer {
    pfm_dense_lin gl1; /* fc_global1: K = T + C + 2H + L, OUT = H */
    pfm_dense_lin gl2; /* fc_global2: K = T + C + H,      OUT = L */
    pfm_local_lin lc1; /* fc_local1 : Ke = T + Cl + L */
    pfm_local_lin lc2; /* fc_local2 : Ke = T + Cl */
} pfm_epic_layer;

typedef struct pfm_epic_desc {
    int32_t abi_version; /* PFM_ABI_VERSION */
    int32_t n_points;    /* N: particles per jet (padded set size) */
    int32_t features;    /* F: per-particle features in and out */
    int32_t hidden;      /* H: must equal PFM_HIDDEN in this build */
    int32_t latent;      /* L <= 16 */
    int32_t layers;      /* EPiC layers <= PFM_MAX_LAYERS */
    int32_t t_dim;       /* T = 2*frequencies (t_local_cat and t_global_cat both on), <= 64 */
    int32_t cond_global; /* Cg <= 16 */
    int32_t cond_local;  /* Cl in {0, Cg} */
    uint32_t flags;
    float sum_scale;     /* epic.py sum_scale (1e-2) */
    float neg_slope;     /* leaky_relu slope (0.01) */
    int64_t blob_floats; /* total length of the blob */
    int64_t freqs;       /* [T] exp(arange(T)) as torch computes it (time_emb.py:90) */
    pfm_dense_lin l1x;   /* fc_l1 particle block: KMAJOR [F][H]; b unused (-1) */
    int64_t l1_We;       /* fc_l1 extras KM16 [T+Cl][H] */
    int64_t l1_b;        /* [H] */
    pfm_local_lin l2;    /* fc_l2 */
    pfm_dense_lin g1;    /* fc_g1: K = T + C + 2H (mean, sum order), OUT = H */
    pfm_dense_lin g2;    /* fc_g2: K = T + C + H, OUT = L */
    pfm_epic_layer layer[PFM_MAX_LAYERS];
    int64_t l3_W;        /* fc_l3 particle block, row-major [F][H] */
    int64_t l3_We;       /* fc_l3 extras KMAJOR [T+Cl][F] */
    int64_t l3_b;        /* [F] */
    int64_t l3_A;        /* fc_l3 particle block as ONE 16-row MFMA_A panel: float4 at (kt*64 + lane) holds
                            W3[lane&15][16*kt + 4*(lane>>4) + r] (rows >= F are zero), 2048 floats */
    int64_t l3_A16;      /* the same panel as MFMA_A16 (one output slice), 1024 floats */
    /* round 4: second copies of the per-jet GEMV blocks WITHOUT their time / conditioning rows, for the lean sampler's chains
       (csrc/epic_fast.h: unconditioned jets; the time rows are tabulated per evaluation) */
    int64_t q_g1;                    /* fc_g1      rows [mean ; sum]     : KQ16, 16 panels */
    int64_t q_g2;                    /* fc_g2      rows of g1 (128)      : WQ16 */
    int64_t q_gl1[PFM_MAX_LAYERS];   /* fc_global1 rows [mean ; sum ; g] : KQ16, 17 panels (g zero-padded to 16 rows) */
    int64_t q_gl2[PFM_MAX_LAYERS];   /* fc_global2 rows of g1            : WQ16 */
    int64_t q_we1[PFM_MAX_LAYERS];   /* fc_local1  extras rows of g      : KQ16, 1 panel */
    /* the same five blocks as CH16 (bf16 descriptors; 0 floats used otherwise): NK = 8 / 4 / 9 / 4 / 1 */
    int64_t b_g1, b_g2;
    int64_t b_gl1[PFM_MAX_LAYERS], b_gl2[PFM_MAX_LAYERS], b_we1[PFM_MAX_LAYERS];
} pfm_epic_desc;

#define PFM_DESC_FLOATS ((int64_t)((sizeof(pfm_epic_desc) + 15) / 16 * 4))

int pfm_abi_version(void);
const char *pfm_last_error(void);

/* Fills the MFMA_A16 blocks of a DEVICE blob from its MFMA_A blocks (one launch on `stream`); a no-op for descriptors without
 * PFM_F_BF16_MFMA.  Whoever writes the fp32 blocks of a bf16 model's blob (the gather of layout.py, pfm_wn_pack) calls it afterwards. */
int pfm_epic_pack_a16(const pfm_epic_desc *desc, float *blob, void *stream);

/* bytes of LDS one workgroup (= one jet) needs in the inference / loss-forward kernels and in the backward kernel; > 163840
 * (or n_points > 160) means PFM_E_LDS: such sets run on the row-matrix path of pfm_epicw.h (the host side picks it by these) */
int64_t pfm_epic_lds_bytes(const pfm_epic_desc *desc);
int64_t pfm_epic_backward_lds_bytes(const pfm_epic_desc *desc);
/* floats of activation workspace per jet that pfm_epic_fm_loss_forward writes for the backward */
int64_t pfm_epic_saved_floats_per_jet(const pfm_epic_desc *desc);

/* v[b,n,:] = EPiC(t[b], x[b], cond[b], mask[b]);  t[B], x[B,N,F], cond[B,Cg]|NULL, mask[B,N]|NULL (float 0/1) */
int pfm_epic_forward(const pfm_epic_desc *desc, const float *blob, const float *t, const float *x,
                     const float *cond, const float *mask, float *v, int32_t B, void *stream);

/* Same with the time embedding supplied by the caller: temb[B,T] (the reference's EPiC_encoder.forward takes the
 * embedded time, epic.py:304-310; only the row of the first particle is used there too, :342). */
int pfm_epic_forward_temb(const pfm_epic_desc *desc, const float *blob, const float *temb, const float *x,
                          const float *cond, const float *mask, float *v, int32_t B, void *stream);

/* Fixed-step explicit midpoint over n_intervals steps, all inside one launch:
 *   x <- z*mask;  for k: k1 = f(t_eval[2k], x); xm = x + 0.5*dt[k]*k1; x <- x + dt[k]*f(t_eval[2k+1], xm)
 * t_eval[2*n_intervals], dt[n_intervals] are the fp32 values the reference's driver visits.
 * scratch (pfm_epic_sample_scratch_floats(desc, n_intervals, B) floats, or NULL) enables two things that do not change results
 * beyond fp32 rounding: (1) every jet is evaluated at the same times, so the time columns of the per-jet Linears give
 * jet-independent vectors: they are tabulated once per call (a tiny launch) and the kernel's per-jet phase skips those weight rows;
 * (2) with a mask, workgroups take the jets in descending multiplicity (a one-workgroup ranking launch): a jet's run time grows
 * with its valid particles and workgroups are dispatched in order, so the short jets fill the tail of a launch -- and, when
 * several launches are in flight, the gaps of the previous one -- instead of a long jet starting last. */
int64_t pfm_epic_sample_scratch_floats(const pfm_epic_desc *desc, int32_t n_intervals, int32_t B);
/* 1 if pfm_epic_sample_midpoint(desc, ..., scratch != NULL, ...) runs the lean evaluation of csrc/epic_fast.h (unconditioned jets,
 * or conditioned ones with cond_local + latent <= 16 and no PFM_F_PACK_JETS: their conditioning terms come from a per-jet table;
 * t_dim = 32, features <= 4, fp32 / bf16 operands, no PFM_F_GENERIC_SAMPLER, the set leaves 1152 bytes of LDS
 * behind the activation tile: n_points <= 150 at features = 3), 0 if the generic kernel. */
int pfm_epic_sample_is_fast(const pfm_epic_desc *desc);
int pfm_epic_sample_midpoint(const pfm_epic_desc *desc, const float *blob, const float *t_eval,
                             const float *dt, int32_t n_intervals, const float *z, const float *cond,
                             const float *mask, float *x_out, int32_t B, float *scratch, void *stream);

/* The same with the time embedding of every evaluation supplied by the caller: temb_tab[2 * n_intervals][T], row e = the embedding
 * of the e-th evaluation time (t_emb="gaussian": the caller's trainable embedding network, flow_matching_module.py:213-221). */
int pfm_epic_sample_midpoint_temb(const pfm_epic_desc *desc, const float *blob, const float *temb_tab, const float *dt,
                                  int32_t n_intervals, const float *z, const float *cond, const float *mask, float *x_out,
                                  int32_t B, float *scratch, void *stream);

/* Explicit Runge-Kutta scheme with up to 4 stages (a strictly lower triangular; row s of `a` feeds stage s):
 *   k_s = f(t + c[s] dt, x + dt * (a[s][0] k_0 + ... + a[s][s-1] k_{s-1}));   x <- x + dt * (b[0] k_0 + ... + b[S-1] k_{S-1})
 * with the sums formed left to right in fp32 before the multiplication by dt (the op order of torchdyn's solver steps).
 * euler: S=1, b={1}.  midpoint: S=2, c={0,1/2}, a[1]={1/2}, b={0,1}.  torchdyn's "rk4" is the 3/8 rule:
 * c={0,1/3,2/3,1}, a[1]={1/3}, a[2]={-1/3,1}, a[3]={1,-1,1}, b={1/8,3/8,3/8,1/8}. */
#define PFM_RK_MAX_STAGES 4
typedef struct {
    int32_t stages, pad_;
    float c[PFM_RK_MAX_STAGES];
    float a[PFM_RK_MAX_STAGES][PFM_RK_MAX_STAGES];
    float b[PFM_RK_MAX_STAGES];
} pfm_rk_tableau;

/* Fixed-step explicit Runge-Kutta over n_intervals steps, all inside one launch: x <- z*mask, then the scheme above per
 * interval.  t_eval[n_intervals * stages] = the stage times t_k + c[s] dt_k, dt[n_intervals]: the fp32 values the reference's
 * driver visits.  kbuf: B * stages * N * F floats of scratch (the stage slopes of every jet) + B more (rounded up to 64) for the
 * jet launch order (see pfm_epic_sample_midpoint).
 * rhs (NULL for flow matching): [n_intervals * stages][2] = (-0.5 beta(t), noise_rate(t)) at every stage time: the ODE
 * right-hand side becomes rhs0 * (x - f(t, x) / rhs1), the probability-flow ODE of a noise-predicting network
 * (ode_wrapper.forward for loss_type="diffusion", flow_matching_module.py:62-69). */
int pfm_epic_sample_rk(const pfm_epic_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                       const float *dt, int32_t n_intervals, const float *z, const float *cond, const float *mask,
                       float *x_out, int32_t B, float *kbuf, const float *rhs, void *stream);

/* The same with the size of kbuf stated: with pfm_epic_sample_rk_scratch_floats(desc, stages, n_intervals, B) floats (stage slopes |
 * jet order | time-term table of every stage time) unconditioned jets (see pfm_epic_sample_is_fast) run the lean evaluation of
 * csrc/epic_fast.h; with less (at least the kbuf of pfm_epic_sample_rk) the generic kernel.  Results differ by fp32 re-association. */
int64_t pfm_epic_sample_rk_scratch_floats(const pfm_epic_desc *desc, int32_t stages, int32_t n_intervals, int32_t B);
int pfm_epic_sample_rk_sized(const pfm_epic_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *t_eval,
                             const float *dt, int32_t n_intervals, const float *z, const float *cond, const float *mask,
                             float *x_out, int32_t B, float *kbuf, int64_t kbuf_floats, const float *rhs, void *stream);

/* pfm_epic_sample_rk with caller-supplied embeddings: temb_tab[n_intervals * stages][T] (flow matching only: rhs = NULL). */
int pfm_epic_sample_rk_temb(const pfm_epic_desc *desc, const float *blob, const pfm_rk_tableau *tab, const float *temb_tab,
                            const float *dt, int32_t n_intervals, const float *z, const float *cond, const float *mask,
                            float *x_out, int32_t B, float *kbuf, void *stream);

/* Flow-matching loss, forward.  kind 0 = "FM-OT" (losses.py:56-62: y=(1-t)x+(sigma+(1-sigma)t)z, u=((1-sigma)z-x)*mask),
 * kind 1 = "CFM" (losses.py:115-119: y=(1-t)x+t*z+sigma*eps, u=(z-x)*mask; eps required).
 * Writes loss_parts[B] = sum_n,f (v-u)^2 of each jet, mask_count[B] = sum_n mask, and the activations the
 * backward needs into saved[B * pfm_epic_saved_floats_per_jet]. */
int pfm_epic_fm_loss_forward(const pfm_epic_desc *desc, const float *blob, int32_t kind, float sigma,
                             const float *t, const float *x, const float *z, const float *eps,
                             const float *cond, const float *mask, float *saved, float *loss_parts,
                             float *mask_count, int32_t B, const int32_t *order, void *stream);

/* Launch order of the jets of a training call: order[rank] = jet, descending multiplicity (ties by index; one small launch).
 * Every loss forward / backward entry point of this header takes `order` (device, B int32, or NULL = batch order): a jet's run
 * time grows with its valid particles and workgroups are dispatched in order, so with more jets than CUs the short jets fill the
 * tail of the launch instead of a long jet starting last.  Scheduling only: every jet writes its own records, results unchanged. */
int pfm_epic_jet_order(const float *mask, int32_t B, int32_t n_points, int32_t *order, void *stream);

/* Backward of the above w.r.t. the blob: grad_blob = d(loss)/d(blob) with loss = sum(loss_parts)/sum(mask_count)
 * scaled by grad_scale (the incoming dL).  inv_mask_total = 1/sum(mask_count) is passed as a device scalar.
 * grad_blob has the offsets (and length) of the blob; every position that carries a parameter's gradient is WRITTEN (not
 * accumulated; the rest is left alone): MFMA_A blocks come back in GRAD_D order, MFMA_AT blocks are not written, everything
 * else is in blob order.  `t` is unused (may be NULL).  scratch: pfm_epic_backward_scratch_floats(desc, B) floats (the
 * gradient rows of every 128x128 Linear, the per-jet rank-1 operands, the partial dW tiles).  Three launches -- the per-jet
 * chain, one GEMM launch for all dW over the rows of all jets, one fixed-order reduction -- and no atomics: the result is a
 * pure function of the inputs, bit for bit, run to run.  B <= 8192 per call. */
int64_t pfm_epic_backward_scratch_floats(const pfm_epic_desc *desc, int32_t B);
int pfm_epic_fm_loss_backward(const pfm_epic_desc *desc, const float *blob, const float *t,
                              const float *cond, const float *mask, const float *saved,
                              const float *inv_mask_total, const float *grad_scale, float *grad_blob,
                              int32_t B, float *scratch, const int32_t *order, void *stream);

/* The same backward in two halves, for a data-parallel caller that starts exchanging gradients before the backward has finished
 * (torch DDP's bucketed, overlapped all-reduce under configs/trainer/ddp.yaml:4-9):
 *   PFM_BWD_PHASE_CHAIN  the per-jet chain + the sums that need its records only: afterwards every gradient slot of grad_blob is final
 *                        EXCEPT the 128x128 particle blocks of fc_l2 / fc_local1 / fc_local2, i.e. the gradients of fc_l1, fc_l3, fc_g1,
 *                        fc_g2 and every fc_global1 / fc_global2 (51 % of the parameters at the JetNet shapes) can be unpacked and reduced
 *   PFM_BWD_PHASE_DW     the dW GEMM + its tile sums (needs the chain phase of the same scratch first)
 * Both bits = pfm_epic_fm_loss_backward, bit for bit.  criterion / jet_weight: 0 / NULL for FM-OT, CFM, droid; as
 * pfm_epic_diffusion_loss_backward otherwise. */
#define PFM_BWD_PHASE_CHAIN 1
#define PFM_BWD_PHASE_DW 2
int pfm_epic_fm_loss_backward_phases(const pfm_epic_desc *desc, const float *blob, const float *cond, const float *mask,
                                     const float *saved, const float *inv_mask_total, const float *grad_scale, float *grad_blob,
                                     int32_t criterion, const float *jet_weight, int32_t B, float *scratch, const int32_t *order,
                                     int32_t phases, void *stream);

/* The same backward, also returning grad_y[B][N][F] = d(loss)/d(y) * grad_scale, the gradient w.r.t. the network's particle input
 * (rows behind a jet's last valid particle: 0): for a caller that chains several flows (n_transforms > 1, flow_matching_module.py:421-443;
 * losses.py:66-69 feeds each flow's output to the next one).  The head of a jet's `saved` record is y | v | u, each round4(N * F)
 * floats (then the activations): a caller that wants the backward to start from an upstream gradient G instead of the loss's own
 * 2 (v - u) writes u := v - G / 2 there and passes inv_mask_total = grad_scale = 1. */
int pfm_epic_fm_loss_backward_dx(const pfm_epic_desc *desc, const float *blob, const float *cond, const float *mask,
                                 const float *saved, const float *inv_mask_total, const float *grad_scale, float *grad_blob,
                                 float *grad_y, int32_t B, float *scratch, const int32_t *order, void *stream);
/* ... for a `saved` record of pfm_epic_fm_loss_forward_temb (caller-supplied time embedding, below): grad_y and grad_temb[B][T] together */
int pfm_epic_fm_loss_backward_dx_temb(const pfm_epic_desc *desc, const float *blob, const float *cond, const float *mask,
                                      const float *saved, const float *inv_mask_total, const float *grad_scale, float *grad_blob,
                                      float *grad_y, float *grad_temb, int32_t B, float *scratch, const int32_t *order, void *stream);

/* The same two with the time embedding supplied by the caller, temb[B][T] (t_emb="gaussian": a small trainable network in front of
 * the field, flow_matching_module.py:178-181, 213-221; t is still needed for the interpolation y, u): the backward also returns
 * grad_temb[B][T] = d(loss)/d(temb) * grad_scale, from which the caller's autograd continues into that network. */
int pfm_epic_fm_loss_forward_temb(const pfm_epic_desc *desc, const float *blob, int32_t kind, float sigma, const float *t,
                                  const float *temb, const float *x, const float *z, const float *eps, const float *cond,
                                  const float *mask, float *saved, float *loss_parts, float *mask_count, int32_t B,
                                  const int32_t *order, void *stream);
int pfm_epic_fm_loss_backward_temb(const pfm_epic_desc *desc, const float *blob, const float *cond, const float *mask,
                                   const float *saved, const float *inv_mask_total, const float *grad_scale, float *grad_blob,
                                   float *grad_temb, int32_t B, float *scratch, const int32_t *order, void *stream);

/* The scalar tail of the losses above (losses.py:75-76: sum / mask.sum()) in one launch: out2[0] = sum_b w_b loss_parts[b] /
 * sum_b mask_count[b] (w = jet_weight, or 1 if NULL), out2[1] = 1 / sum_b mask_count[b] (the inv_mask_total of the backward).
 * Sums in a fixed order. */
int pfm_loss_finish(const float *loss_parts, const float *mask_count, const float *jet_weight, int32_t B, float *out2, void *stream);

/* DiffusionLoss (models/components/losses.py:207-290, configs/model/diffusion.yaml): noisy = rates[b][0] * x + rates[b][1] * z
 * (signal / noise rate of the jet's diffusion time, models/components/diffusion.py:21-52; z arrives multiplied by the mask),
 * v = f(t, noisy) predicts z.  criterion 0 = mse, 1 = huber (delta 1).  loss_parts[b] = sum_n,f criterion(v - z) of the jet
 * (the caller applies the per-jet weight 1 + 0.001 beta / noise_rate, losses.py:275-281); the rest as pfm_epic_fm_loss_forward. */
int pfm_epic_diffusion_loss_forward(const pfm_epic_desc *desc, const float *blob, int32_t criterion, const float *rates,
                                    const float *t, const float *x, const float *z, const float *cond, const float *mask,
                                    float *saved, float *loss_parts, float *mask_count, int32_t B, const int32_t *order, void *stream);

/* Backward of loss = sum_b jet_weight[b] * loss_parts[b] / sum(mask_count), as pfm_epic_fm_loss_backward. */
int pfm_epic_diffusion_loss_backward(const pfm_epic_desc *desc, const float *blob, int32_t criterion, const float *jet_weight,
                                     const float *cond, const float *mask, const float *saved, const float *inv_mask_total,
                                     const float *grad_scale, float *grad_blob, int32_t B, float *scratch, const int32_t *order,
                                     void *stream);

/* One in-place state update of the diffusion samplers (models/components/solver.py): mode 0 = ddim_sampler :81-93
 * (c = noise rate, signal rate, next signal rate, next noise rate; data_out, optional, receives the predicted data),
 * mode 1 = euler_maruyama_sampler :126-132 (c = noise rate, beta, delta_t, sqrt(beta delta_t); noise = the step's normal draw). */
int pfm_diffusion_update(int32_t mode, float *x, const float *pred, const float *noise, float c0, float c1, float c2, float c3,
                         float *data_out, int64_t n, void *stream);

/* IterativeNormLayer (models/components/norm_layer.py:17-155; SetFlowMatchingLitModule with use_normaliser=True,
 * flow_matching_module.py:467-473, 514-518, 666-677) on rows x[rows][features] (features <= 16), row mask fp32 {0,1} or NULL.
 * pfm_norm_update: one running-statistics step with the valid rows (fit() on the first batch, the batched Welford update()
 * afterwards, nothing once *n >= max_n); n (int64), means / vars / m2 [features] are the module's buffers, on the device.
 * pfm_norm_apply: out = (x - means) / (sqrt(vars) + 1e-8) on the valid rows (reverse != 0: x sqrt(vars) + means), others copied. */
int pfm_norm_update(const float *x, const float *mask, int64_t rows, int32_t features, int64_t *n, float *means, float *vars,
                    float *m2, int64_t max_n, void *stream);
int pfm_norm_apply(float *out, const float *x, const float *mask, int64_t rows, int32_t features, const float *means,
                   const float *vars, int32_t reverse, void *stream);

/* Optimiser tail on flat fp32 buffers of n elements:
 *   gnorm = ||grad * grad_mul||_2 ; c = min(1, max_norm/(gnorm+1e-6)) (clip_grad_norm_) ; g = grad*grad_mul*c
 *   AdamW (decoupled weight decay, torch.optim.AdamW defaults eps/betas passed in) ; ema = decay*ema + (1-decay)*p
 * `scratch` needs >= 1024 floats; on return (stream order) scratch[0] = gnorm^2 (0 without clipping).  The norm is reduced in a
 * fixed order, without atomics: ranks that hold the same all-reduced gradient apply bit-identical updates.  step is 1-based. */
int pfm_optim_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema,
                   float *scratch, int64_t n, float grad_mul, float max_norm, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float ema_decay, int32_t step,
                   void *stream);

/* Post-processing of a generated batch in place, on the device (utils/data_generation.py:94-123 does it on the host
 * after a D2H copy per batch): x[row][f] = x * scale[f] + shift[f] (inverse_normalize_tensor, data/components/utils.py:
 * 183-199; scale = std/sigma, shift = mean; NULL/NULL = skip), then x[row][log_pt_col] = 1 - exp(x) (log_pt_col < 0:
 * skip), then x *= mask[row] (NULL: skip).  x [rows][features], mask [rows], scale / shift [features] device pointers. */
int pfm_sample_epilogue(float *x, const float *mask, const float *scale, const float *shift, int32_t log_pt_col,
                        int64_t rows, int32_t features, void *stream);

/* Weight-norm reparametrisation on FLAT buffers (old-style nn.utils.weight_norm, epic.py:66-81, 262-300):
 *   pack:   W[o,:] = g[o] * v[o,:] / ||v[o,:]||  written to blob[dst1[s]] and blob[dst2[s]] (s = source index of the
 *           element, -1 = no destination), biases copied params[bias_from[i]] -> blob[bias_to[i]].
 *   unpack: grad[g] += <dW,v>/||v||, grad[v] += (g/||v||)(dW - v<dW,v>/||v||^2) with dW[s] = gblob[gsrc[s]],
 *           grad[bias_to[i]] += gblob[bias_from[i]].
 * rows[n_rows][4] = {v_off, g_off, in_dim, src_off} (int32, device), one entry per weight row. */
int pfm_wn_pack(const float *params, const int32_t *rows, int32_t n_rows, const int32_t *dst1, const int32_t *dst2,
                const int32_t *bias_from, const int32_t *bias_to, int32_t n_bias, float *blob, void *stream);
int pfm_wn_unpack_grad(const float *params, const float *gblob, const int32_t *rows, int32_t n_rows,
                       const int32_t *gsrc, const int32_t *bias_from, const int32_t *bias_to, int32_t n_bias,
                       float *grad, void *stream);
/* The same with `=` in place of `+=`: every g / v / bias element the tables name is WRITTEN (a caller whose tables cover the whole
 * flat buffer needs no zeroing launch in front). */
int pfm_wn_unpack_grad_set(const float *params, const float *gblob, const int32_t *rows, int32_t n_rows,
                           const int32_t *gsrc, const int32_t *bias_from, const int32_t *bias_to, int32_t n_bias,
                           float *grad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFM_HIP_H */
