// gfx950 kernels + C ABI (include/pfm_ca.h): cross-attention vector field (model "droid_fullcrossattention"), midpoint
// sampler, FM / CFM / droid loss forward and backward.  Linears, LayerNorm backward, dW / dX, the context path and the
// output head are the kernels of tf_fwd.h / tf_bwd.h on two row matrices (particles, global tokens); the attention
// shapes are in ca_attn.h.
#include <hip/hip_runtime.h>

#include "pfm_ca.h"
#include "tf_fwd.h"
#include "tf_bwd.h"
#include "ca_attn.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

namespace ca {
using namespace pfm::tf;

int validate(const pfm_ca_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_CA_ABI_VERSION) return set_err(PFM_E_BADARG, "ca desc.abi_version mismatch");
    if (d->model_dim < 128 || d->model_dim > MAXK || d->model_dim % 128) return set_err(PFM_E_BADARG, "model_dim must be a multiple of 128 in 128..512");
    if (d->hidden < 128 || d->hidden > MAXK || d->hidden % 128) return set_err(PFM_E_BADARG, "hidden must be a multiple of 128 in 128..512");
    if ((d->head_dim != 8 && d->head_dim != 16) || d->heads * d->head_dim != d->model_dim || d->heads > 64)
        return set_err(PFM_E_BADARG, "head_dim must be 8 or 16 with heads * head_dim = model_dim");
    if (d->tokens < 1 || d->tokens > PFM_CA_MAX_TOKENS || d->tokens * d->head_dim > 64)
        return set_err(PFM_E_BADARG, "tokens must be in 1..8 with tokens * head_dim <= 64");
    if (d->layers < 1 || d->layers > PFM_CA_MAX_LAYERS) return set_err(PFM_E_BADARG, "layers out of range");
    if (d->features < 1 || d->features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    if (d->t_dim < 1 || d->t_dim > 64 || d->cond_dim < 0 || d->cond_dim > 16) return set_err(PFM_E_BADARG, "t_dim / cond_dim out of range");
    if (d->ctxt_dim < 4 || d->ctxt_dim > 64 || d->ctxt_dim % 4 || d->ctxt_hidden < 4 || d->ctxt_hidden > 512 || d->ctxt_hidden % 4)
        return set_err(PFM_E_BADARG, "ctxt_dim / ctxt_hidden out of range");
    if (d->n_points < 1) return set_err(PFM_E_BADARG, "n_points must be >= 1");
    return 0;
}

// Workspace (floats).  Train: every layer keeps its buffers; inference: one set, streams updated in place.
struct Ws {
    int64_t temb, chid, ctxt, jb, h1, seq0, tok0, layer0, lstride;
    int64_t f_kv, f_q, f_att, f_mid, f_dh, f_out;  // from-layer: kv [M][2D], q / att / mid / out [Mt][D], dh [Mt][Hd]
    int64_t t_q, t_kv, t_att, t_mid, t_dh, t_out;  // to-layer:   q / att / mid / out [M][D], kv [Mt][2D], dh [M][Hd]
    int64_t oh, imaps, step, total;  // imaps: int32 row maps of the valid-rows-only evaluation; step: see ca_step_args_kernel
};

Ws make_ws(const pfm_ca_desc& d, int n_jets, bool train) {
    Ws w;
    const int64_t M = (int64_t)n_jets * d.n_points, Mt = (int64_t)n_jets * d.tokens, D = d.model_dim, Hd = d.hidden;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    w.temb = take((int64_t)n_jets * 64);
    w.chid = take((int64_t)n_jets * d.ctxt_hidden);
    w.ctxt = take((int64_t)n_jets * d.ctxt_dim);
    w.jb = take((int64_t)n_jets * (2 * d.layers + 2) * Hd);
    w.h1 = take(M * Hd);
    w.seq0 = take(M * D);
    w.tok0 = take(Mt * D);
    w.layer0 = o;
    int64_t p = 0;
    auto sub = [&](int64_t n) { const int64_t at = p; p += round64(n); return at; };
    w.f_kv = sub(M * 2 * D); w.f_q = sub(Mt * D); w.f_att = sub(Mt * D); w.f_dh = sub(Mt * Hd);
    w.t_q = sub(M * D); w.t_kv = sub(Mt * 2 * D); w.t_att = sub(M * D); w.t_dh = sub(M * Hd);
    if (train) {
        w.f_mid = sub(Mt * D); w.f_out = sub(Mt * D); w.t_mid = sub(M * D); w.t_out = sub(M * D);
        w.lstride = p;
        o += p * d.layers;
    } else {
        w.f_mid = w.f_out = w.tok0 - w.layer0;  // in place
        w.t_mid = w.t_out = w.seq0 - w.layer0;
        w.lstride = 0;
        o += p;
    }
    w.oh = take(M * Hd);
    w.imaps = take(row_maps_ints(n_jets, M));
    w.step = take(tf::STEP_SLOT_FLOATS);
    w.total = o;
    return w;
}

struct Plan {
    const pfm_ca_desc* d;
    const float* blob;
    float* ws;
    Ws w;
    int n_jets, M, Mt;
    hipStream_t s;
    int temb_k = 0;  // PFM_CA_F_TEMB_GIVEN: floats between the elements of a time-embedding row in the `t` argument (0: `t` holds times)
    // valid-rows-only evaluation (PFM_CA_F_VALID_ROWS, inference): the particle rows are the valid particles
    const int *rowsrc = nullptr, *rowjet = nullptr, *off = nullptr, *m_dev = nullptr, *cnt = nullptr, *order = nullptr;
};

// out[rows][ldo] = epi(LN?(A) W^T + b / jet bias) (+R); per_jet = rows per jet of this row matrix (jet-bias lookup)
int linear(const Plan& p, int rows, int per_jet, const float* A, int lda, int K, const pfm_tf_lin& lin, const pfm_tf_norm* ln, int NO,
           const float* jb, const float* R, int ldr, float* out, int ldo, int act) {
    LinArgs a;
    a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.blob = p.blob; a.jb = jb; a.R = R; a.Y = nullptr; a.ldy = 0;
    const bool prow = p.rowsrc && rows == p.M && per_jet == p.d->n_points;  // a particle-row GEMM of a compacted evaluation
    a.rowjet = (prow && jb) ? p.rowjet : nullptr; a.m_dev = prow ? p.m_dev : nullptr; a.part = nullptr; a.ksplit = 1; a.out = out;
    a.blob_floats = p.d->blob_floats; a.W = lin.W; a.b = lin.b;
    a.gamma = ln ? ln->gamma : -1; a.beta = ln ? ln->beta : -1;
    a.jb_stride = (int64_t)(2 * p.d->layers + 2) * p.d->hidden;
    a.lda = lda; a.ldr = ldr; a.ldo = ldo; a.M = rows; a.K = K; a.NO = NO; a.N = per_jet; a.act = act;
    a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
    const int ni = (ln && ln->gamma >= 0) ? K / 64 : 0;
    if (launch_linear_kernel(a, ni, (p.d->flags & PFM_CA_F_F16X3) ? 1 : ((p.d->flags & PFM_CA_F_BF16) ? 2 : 0), num_cus(), p.s))
        return set_err(PFM_E_BADARG, "LayerNorm width must be 128, 256, 384 or 512");
    return check_hip(hipGetLastError(), "tf_linear_kernel launch (ca)");
}

// two plain LayerNorm-Linears of the same particle rows (from.kv and to.q of a layer pair) in one launch (tf_linear_panel2_kernel); false: not launched
bool linear_pair(const Plan& p, int rows, int per_jet, const float* A, int D, const pfm_tf_lin& w1, const pfm_tf_norm* n1, int NO1, float* out1,
                 const pfm_tf_lin& w2, const pfm_tf_norm* n2, int NO2, float* out2) {
    if ((p.d->flags & (PFM_CA_F_F16X3 | PFM_CA_F_BF16)) || !n1 || !n2 || n1->gamma < 0 || n2->gamma < 0) return false;
    LinArgs a;
    a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = D; a.blob = p.blob; a.jb = nullptr; a.R = nullptr; a.Y = nullptr; a.ldy = 0; a.rowjet = nullptr;
    const bool prow = p.rowsrc && rows == p.M && per_jet == p.d->n_points;
    a.m_dev = prow ? p.m_dev : nullptr; a.part = nullptr; a.ksplit = 1;
    a.blob_floats = p.d->blob_floats; a.jb_stride = 0; a.lda = D; a.ldr = 0; a.M = rows; a.K = D; a.N = per_jet; a.act = 0;
    a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
    LinArgs b = a;
    a.W = w1.W; a.b = w1.b; a.gamma = n1->gamma; a.beta = n1->beta; a.NO = NO1; a.out = out1; a.ldo = NO1;
    b.W = w2.W; b.b = w2.b; b.gamma = n2->gamma; b.beta = n2->beta; b.NO = NO2; b.out = out2; b.ldo = NO2;
    return launch_panel2(a, b, num_cus(), p.s);
}

// the dense block of a particle row matrix  out = mid + d2(LN(lrelu(d1(LN(mid)) + jet bias)))  (droid_transformer.py:793-813 / 958-981): one launch
// (tf_mlp_panel_kernel) where the shapes and the row count allow it, else the two Linears through the hidden buffer `dh`
// (att != nullptr: with the Linear in front of it, mid = res + lo(LN(att)), as stage 0 of the same launch)
int dense_block(const Plan& p, int rows, int per_jet, float* mid, int D, int Hd, const pfm_tf_lin& d1, const pfm_tf_norm* n1, const float* jb,
                const pfm_tf_lin& d2, const pfm_tf_norm* n2, float* dh, float* out, const float* att = nullptr, const pfm_tf_lin* lo = nullptr,
                const pfm_tf_norm* no = nullptr, const float* res = nullptr) {
    bool pre_done = false;
    if (!(p.d->flags & (PFM_CA_F_F16X3 | PFM_CA_F_BF16)) && n1 && n2 && n1->gamma >= 0 && n2->gamma >= 0) {
        LinArgs a, b;
        a.A = mid; a.A2 = nullptr; a.lda2 = 0; a.K1 = D; a.blob = p.blob; a.jb = jb; a.R = nullptr; a.Y = nullptr; a.ldy = 0;
        const bool prow = p.rowsrc && rows == p.M && per_jet == p.d->n_points;
        a.rowjet = (prow && jb) ? p.rowjet : nullptr; a.m_dev = prow ? p.m_dev : nullptr; a.part = nullptr; a.ksplit = 1; a.out = nullptr;
        a.blob_floats = p.d->blob_floats; a.W = d1.W; a.b = d1.b; a.gamma = n1->gamma; a.beta = n1->beta;
        a.jb_stride = (int64_t)(2 * p.d->layers + 2) * p.d->hidden;
        a.lda = D; a.ldr = 0; a.ldo = Hd; a.M = rows; a.K = D; a.NO = Hd; a.N = per_jet; a.act = 1;
        a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
        b = a;
        b.A = nullptr; b.jb = nullptr; b.rowjet = nullptr; b.W = d2.W; b.b = d2.b; b.gamma = n2->gamma; b.beta = n2->beta;
        b.R = mid; b.ldr = D; b.out = out; b.ldo = D; b.K = Hd; b.NO = D; b.act = 0;
        if (att && no && no->gamma >= 0) {
            LinArgs z = a;
            z.A = att; z.jb = nullptr; z.rowjet = nullptr; z.W = lo->W; z.b = lo->b; z.gamma = no->gamma; z.beta = no->beta;
            z.R = res; z.ldr = D; z.out = mid; z.ldo = D; z.NO = D; z.act = 0;
            if (launch_mlp_panel(a, b, num_cus(), p.s, &z)) return check_hip(hipGetLastError(), "tf_mlp_panel_kernel launch (ca)");
        }
        if (att) {
            int rc = linear(p, rows, per_jet, att, D, D, *lo, no, D, nullptr, res, D, mid, D, 0);
            if (rc) return rc;
            pre_done = true;
        }
        if (launch_mlp_panel(a, b, num_cus(), p.s)) return check_hip(hipGetLastError(), "tf_mlp_panel_kernel launch (ca)");
    }
    if (att && !pre_done) {
        int rc = linear(p, rows, per_jet, att, D, D, *lo, no, D, nullptr, res, D, mid, D, 0);
        if (rc) return rc;
    }
    int rc = linear(p, rows, per_jet, mid, D, D, d1, n1, Hd, jb, nullptr, 0, dh, Hd, 1);
    if (rc) return rc;
    return linear(p, rows, per_jet, dh, Hd, Hd, d2, n2, D, nullptr, mid, D, out, D, 0);
}

// ------------------------------------------------------------------------------------------------
// The token side of a layer pair behind its attention, in ONE launch (inference): one workgroup per jet walks
//   mid = tok + out(LN_attn(att))                        droid_transformer.py:380-397 (TransformerCrossAttentionLayer)
//   dh  = lrelu(d1(LN_2(mid)) + ctxt bias)               :793-813 (DenseNetwork / MLPBlock)
//   tok = mid + d2(LN_d(dh))
//   kv  = kv_linear(LN_0(tok))      of the particles <- tokens layer that follows
//   q   = q_linear(LN_1(tok))       of the NEXT pair's tokens <- particles layer (if any)
// As row GEMMs over the 4 n_jets token rows these were five launches of a few workgroups each, ~7 us of launch + ramp latency
// apiece (40 of the ~100 launches of an evaluation).  Here the jet's Tk <= 8 token rows live in LDS and every Linear is a GEMV
// group straight on the MFMA_AK blocks (round 2's per-jet chain of the row-matrix EPiC path worked the same way): a wave takes a 16-output block, lane (i, q)
// multiplies the weight float4 it would feed the matrix pipe with by the matching float4 of each token row.
// ------------------------------------------------------------------------------------------------
constexpr int TKLD = MAXK + 8;          // floats per LDS row

struct TokArgs {
    const float *blob, *att, *jb;  // att [Mt][D]; jb: this layer's context-bias row of jet j at jb + j * jb_stride (Hd floats)
    float *tok, *kv_out, *q_out;   // tok [Mt][D] in / out; kv_out [Mt][2D]; q_out [Mt][D] or nullptr
    pfm_tf_norm attn_norm, norm2, d_norm, kv_norm, q_norm;
    pfm_tf_lin out, d1, d2, kv, q;
    int64_t jb_stride;
    int D, Hd, Tk;
    float slope, eps;
};

// LayerNorm of the workgroup's Tk rows src[Tk][TKLD] (width K) into dst; a norm with gamma < 0 copies.  One wave per row,
// the two-pass statistics of tf_fwd.h::ln_stats_tile.  Ends behind a barrier.
__device__ __forceinline__ void tok_layernorm(const float* __restrict__ blob, const pfm_tf_norm& nm, const float* __restrict__ src,
                                              float* __restrict__ dst, int K, int Tk, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = w; r < Tk; r += (int)(blockDim.x >> 6)) {
        const float* x = src + r * TKLD;
        float s = 0.f;
        for (int c = lane; c < K; c += 64) s += x[c];
        const float mean = wsum(s) / (float)K;
        float ss = 0.f;
        for (int c = lane; c < K; c += 64) { const float dl = x[c] - mean; ss += dl * dl; }
        const float rstd = 1.0f / sqrtf(wsum(ss) / (float)K + eps);
        for (int c = lane; c < K; c += 64)
            dst[r * TKLD + c] = nm.gamma >= 0 ? (x[c] - mean) * rstd * blob[nm.gamma + c] + blob[nm.beta + c] : x[c];
    }
    __syncthreads();
}

constexpr int TKT = 512;       // threads of a token-chain workgroup
constexpr int TKW = TKT / 64;  // waves: each takes the output blocks w, w + TKW, ...

// One Linear on the workgroup's token rows: out(ob, i, v[r]) is called by lanes i < 16 with v[r] = W[16 ob + i] . vin[r] (K = 64 ksteps,
// ksteps even: widths are multiples of 128).  The wave walks its (output block, 128-wide K chunk) pairs as one flat sequence with the
// NEXT pair's eight weight loads in flight behind the current pair's FMAs: one exposed L2 round trip per Linear instead of one per block.
template <int TK, typename Out>
__device__ __forceinline__ void tok_linear(const float* __restrict__ blob, int64_t W, int ksteps, int nob, const float* __restrict__ vin, Out out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane >> 4;
    const int nch = ksteps >> 1;
    const int mine = w < nob ? (nob - w + TKW - 1) / TKW : 0, total = mine * nch;
    if (total == 0) return;
    auto wptr = [&](int f) { return blob + W + ((int64_t)(w + TKW * (f / nch)) * ksteps + 2 * (f % nch)) * 1024 + lane * 4; };
    f32x4 wa[8], wb[8];
    f32x4 acc[TK];
    auto load = [&](f32x4 (&wv)[8], int f) {
        const float* wp = wptr(f);
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const f32x4*>(wp + 256 * i);
    };
    auto compute = [&](const f32x4 (&wv)[8], int f) {
        const int ch = f % nch;
        if (ch == 0) {
#pragma unroll
            for (int r = 0; r < TK; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* xp = vin + 128 * ch + 4 * q;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < TK; ++r) acc[r] += wv[i] * *reinterpret_cast<const f32x4*>(xp + r * TKLD + 16 * i);
        if (ch == nch - 1) {
            float v[TK];
#pragma unroll
            for (int r = 0; r < TK; ++r) {
                float t = hsum4(acc[r]);
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                v[r] = t;
            }
            if (lane < 16) out(w + TKW * (f / nch), lane, v);
        }
    };
    load(wa, 0);
    int f = 0;
    for (; f + 1 < total; f += 2) {
        load(wb, f + 1);
        compute(wa, f);
        if (f + 2 < total) load(wa, f + 2);
        compute(wb, f + 1);
    }
    if (f < total) compute(wa, f);
}

template <int TK>
__global__ __launch_bounds__(TKT) void ca_token_chain_kernel(TokArgs a) {
    __shared__ __attribute__((aligned(16))) float xa[TK * TKLD];  // LayerNorm output = GEMV input
    __shared__ __attribute__((aligned(16))) float xb[TK * TKLD];  // raw rows: att, then dh
    __shared__ __attribute__((aligned(16))) float mid[TK * TKLD]; // the token stream: tok, mid, new tok
    const int tid = threadIdx.x, jet = blockIdx.x;
    const int D = a.D, Hd = a.Hd, Tk = a.Tk;
    const int64_t row0 = (int64_t)jet * Tk;
    // rows beyond Tk stay zero (they are multiplied along and never stored)
    for (int i = tid; i < TK * TKLD; i += TKT) { xa[i] = 0.f; xb[i] = 0.f; mid[i] = 0.f; }
    __syncthreads();
    for (int i = tid; i < Tk * D; i += TKT) {
        const int r = i / D, c = i - r * D;
        xb[r * TKLD + c] = a.att[(row0 + r) * D + c];
        mid[r * TKLD + c] = a.tok[(row0 + r) * D + c];
    }
    __syncthreads();
    // ---- mid = tok + out(LN_attn(att)) ----
    tok_layernorm(a.blob, a.attn_norm, xb, xa, D, Tk, a.eps);
    tok_linear<TK>(a.blob, a.out.W, D / 64, D / 16, xa, [&](int ob, int i, const float (&v)[TK]) {
        const float b = a.blob[a.out.b + 16 * ob + i];
#pragma unroll
        for (int r = 0; r < TK; ++r)
            if (r < Tk) mid[r * TKLD + 16 * ob + i] += v[r] + b;  // each element has one owner: no race
    });
    __syncthreads();
    // ---- dh = lrelu(d1(LN_2(mid)) + ctxt bias) ----
    tok_layernorm(a.blob, a.norm2, mid, xa, D, Tk, a.eps);
    const float* jb = a.jb + (int64_t)jet * a.jb_stride;
    tok_linear<TK>(a.blob, a.d1.W, D / 64, Hd / 16, xa, [&](int ob, int i, const float (&v)[TK]) {
        const float b = jb[16 * ob + i];  // includes the Linear's bias (tf_ctxt_kernel)
#pragma unroll
        for (int r = 0; r < TK; ++r)
            if (r < Tk) xb[r * TKLD + 16 * ob + i] = lrelu(v[r] + b, a.slope);
    });
    __syncthreads();
    // ---- tok = mid + d2(LN_d(dh)) ----
    tok_layernorm(a.blob, a.d_norm, xb, xa, Hd, Tk, a.eps);
    tok_linear<TK>(a.blob, a.d2.W, Hd / 64, D / 16, xa, [&](int ob, int i, const float (&v)[TK]) {
        const float b = a.blob[a.d2.b + 16 * ob + i];
#pragma unroll
        for (int r = 0; r < TK; ++r)
            if (r < Tk) {
                const float t = mid[r * TKLD + 16 * ob + i] + (v[r] + b);
                mid[r * TKLD + 16 * ob + i] = t;
                a.tok[(row0 + r) * D + 16 * ob + i] = t;
            }
    });
    __syncthreads();
    // ---- kv = kv_linear(LN_0(tok)) of the particles <- tokens layer ----
    tok_layernorm(a.blob, a.kv_norm, mid, xa, D, Tk, a.eps);
    tok_linear<TK>(a.blob, a.kv.W, D / 64, 2 * D / 16, xa, [&](int ob, int i, const float (&v)[TK]) {
        const float b = a.blob[a.kv.b + 16 * ob + i];
#pragma unroll
        for (int r = 0; r < TK; ++r)
            if (r < Tk) a.kv_out[(row0 + r) * 2 * D + 16 * ob + i] = v[r] + b;
    });
    if (!a.q_out) return;
    __syncthreads();
    // ---- q = q_linear(LN_1(tok)) of the next pair ----
    tok_layernorm(a.blob, a.q_norm, mid, xa, D, Tk, a.eps);
    tok_linear<TK>(a.blob, a.q.W, D / 64, D / 16, xa, [&](int ob, int i, const float (&v)[TK]) {
        const float b = a.blob[a.q.b + 16 * ob + i];
#pragma unroll
        for (int r = 0; r < TK; ++r)
            if (r < Tk) a.q_out[(row0 + r) * D + 16 * ob + i] = v[r] + b;
    });
}

#define PFM_TRY(x) do { if ((rc = (x))) return rc; } while (0)
#define PFM_ATTN(KERNEL, grid, lds, ...)                                                                   \
    do {                                                                                                   \
        if (d.head_dim == 16) hipLaunchKernelGGL((KERNEL<16, 4>), grid, dim3(256), lds, p.s, __VA_ARGS__); \
        else if (d.tokens <= 4) hipLaunchKernelGGL((KERNEL<8, 4>), grid, dim3(256), lds, p.s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<8, 8>), grid, dim3(256), lds, p.s, __VA_ARGS__);                   \
    } while (0)

int run_nfe(const Plan& p, const float* t, int t_stride, const float* x, const float* cond, const float* mask, const HeadArgs& head_tpl) {
    const pfm_ca_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    const int D = d.model_dim, Hd = d.hidden, nb = 2 * d.layers + 2, N = d.n_points, Tk = d.tokens, heads = d.heads;
    int rc;
    {
        CtxtArgs a;
        a.blob = p.blob; a.t = t; a.cond = cond;
        a.temb = ws + w.temb; a.chid = ws + w.chid; a.ctxt = ws + w.ctxt; a.jb = ws + w.jb;
        a.t_stride = p.temb_k ? (t_stride ? d.t_dim : 0) : t_stride; a.temb_k = p.temb_k;
        a.T = d.t_dim; a.C = d.cond_dim; a.CH = d.ctxt_hidden; a.CO = d.ctxt_dim; a.Hd = Hd; a.nb = nb;
        a.slope = d.neg_slope; a.eps = d.ln_eps; a.sincos = (d.flags & PFM_CA_F_TEMB_SINCOS) ? 1 : 0;
        a.freqs = d.freqs; a.c1W = d.c1.W; a.c1b = d.c1.b; a.cg = d.c_norm.gamma; a.cb = d.c_norm.beta;
        a.c2W = d.c2.W; a.c2b = d.c2.b; a.n1Wt = d.time_in_input ? d.n1.Wt : -1;
        a.Wc[0] = d.n1.Wc; a.bb[0] = d.n1.b;
        for (int l = 0; l < d.layers; ++l) {
            a.Wc[1 + 2 * l] = d.from_layer[l].d1.Wc; a.bb[1 + 2 * l] = d.from_layer[l].d1.b;
            a.Wc[2 + 2 * l] = d.to_layer[l].d1.Wc; a.bb[2 + 2 * l] = d.to_layer[l].d1.b;
        }
        a.Wc[nb - 1] = d.o1.Wc; a.bb[nb - 1] = d.o1.b;
        launch_ctxt(a, p.n_jets, p.s);
        PFM_TRY(check_hip(hipGetLastError(), "tf_ctxt_kernel launch (ca)"));
    }
    const float* jb = ws + w.jb;
    hipLaunchKernelGGL(tf_embed_kernel, dim3((p.M + 31) / 32), dim3(256), 0, p.s, p.blob, d.n1.W, x, jb, (int64_t)nb * Hd, ws + w.h1, p.M,
                       N, d.features, Hd, d.neg_slope, p.rowsrc, p.rowjet, p.m_dev);
    PFM_TRY(check_hip(hipGetLastError(), "tf_embed_kernel launch (ca)"));
    PFM_TRY(linear(p, p.M, N, ws + w.h1, Hd, Hd, d.n2, &d.n_norm, D, nullptr, nullptr, 0, ws + w.seq0, D, 0));
    {
        const int64_t n = (int64_t)p.Mt * D;
        hipLaunchKernelGGL(ca_tokens_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, p.blob, d.global_tokens,
                           ws + w.tok0, n, Tk * D);
        PFM_TRY(check_hip(hipGetLastError(), "ca_tokens_init_kernel launch"));
    }
    const float *seq = ws + w.seq0, *tok = ws + w.tok0;
    // inference (one in-place buffer set): the token side of a layer pair behind its attention is one launch (ca_token_chain_kernel)
    const bool fused_tokens = w.lstride == 0 && !(d.flags & PFM_CA_F_F16X3);
    for (int l = 0; l < d.layers; ++l) {
        float* lb = ws + w.layer0 + w.lstride * l;
        const pfm_ca_layer& Fl = d.from_layer[l];
        const pfm_ca_layer& Tl = d.to_layer[l];
        // tokens <- particles
        // from.kv and to.q both read the particle rows of the pair's input (the tokens <- particles half does not change them): one launch at inference
        const bool kvq = fused_tokens && linear_pair(p, p.M, N, seq, D, Fl.kv, &Fl.norm0, 2 * D, lb + w.f_kv, Tl.q, &Tl.norm1, D, lb + w.t_q);
        if (kvq) PFM_TRY(check_hip(hipGetLastError(), "tf_linear_panel2_kernel launch (ca)"));
        if (!kvq) PFM_TRY(linear(p, p.M, N, seq, D, D, Fl.kv, &Fl.norm0, 2 * D, nullptr, nullptr, 0, lb + w.f_kv, 2 * D, 0));
        if (!fused_tokens || l == 0)  // (fused: the previous pair's token chain already wrote this pair's queries)
            PFM_TRY(linear(p, p.Mt, Tk, tok, D, D, Fl.q, &Fl.norm1, D, nullptr, nullptr, 0, lb + w.f_q, D, 0));
        PFM_ATTN(ca_attn_from_kernel, dim3(p.n_jets, (heads + 3) / 4), 0, (const float*)(lb + w.f_q), (const float*)(lb + w.f_kv), mask, lb + w.f_att, N, D,
                 heads, Tk, p.off, p.order);
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_from_kernel launch"));
        if (fused_tokens) {
            TokArgs a;
            a.blob = p.blob; a.att = lb + w.f_att; a.jb = jb + (int64_t)(1 + 2 * l) * Hd; a.jb_stride = (int64_t)nb * Hd;
            a.tok = ws + w.tok0; a.kv_out = lb + w.t_kv; a.q_out = l + 1 < d.layers ? lb + w.f_q : nullptr;
            a.attn_norm = Fl.attn_norm; a.norm2 = Fl.norm2; a.d_norm = Fl.d_norm; a.kv_norm = Tl.norm0;
            a.out = Fl.out; a.d1 = Fl.d1; a.d2 = Fl.d2; a.kv = Tl.kv;
            if (l + 1 < d.layers) { a.q_norm = d.from_layer[l + 1].norm1; a.q = d.from_layer[l + 1].q; }
            else { a.q_norm = Fl.norm1; a.q = Fl.q; }
            a.D = D; a.Hd = Hd; a.Tk = Tk; a.slope = d.neg_slope; a.eps = d.ln_eps;
            if (Tk <= 4) hipLaunchKernelGGL(ca_token_chain_kernel<4>, dim3(p.n_jets), dim3(TKT), 0, p.s, a);
            else hipLaunchKernelGGL(ca_token_chain_kernel<PFM_CA_MAX_TOKENS>, dim3(p.n_jets), dim3(TKT), 0, p.s, a);
            PFM_TRY(check_hip(hipGetLastError(), "ca_token_chain_kernel launch"));
            tok = ws + w.tok0;
        } else {
            PFM_TRY(linear(p, p.Mt, Tk, lb + w.f_att, D, D, Fl.out, &Fl.attn_norm, D, nullptr, tok, D, lb + w.f_mid, D, 0));
            PFM_TRY(linear(p, p.Mt, Tk, lb + w.f_mid, D, D, Fl.d1, &Fl.norm2, Hd, jb + (int64_t)(1 + 2 * l) * Hd, nullptr, 0, lb + w.f_dh, Hd, 1));
            PFM_TRY(linear(p, p.Mt, Tk, lb + w.f_dh, Hd, Hd, Fl.d2, &Fl.d_norm, D, nullptr, lb + w.f_mid, D, lb + w.f_out, D, 0));
            tok = lb + w.f_out;
        }
        // particles <- tokens
        if (!kvq) PFM_TRY(linear(p, p.M, N, seq, D, D, Tl.q, &Tl.norm1, D, nullptr, nullptr, 0, lb + w.t_q, D, 0));
        if (!fused_tokens) PFM_TRY(linear(p, p.Mt, Tk, tok, D, D, Tl.kv, &Tl.norm0, 2 * D, nullptr, nullptr, 0, lb + w.t_kv, 2 * D, 0));
        PFM_ATTN(ca_attn_to_kernel, dim3(p.n_jets, (N + TO_ROWS - 1) / TO_ROWS), (size_t)Tk * 2 * D * sizeof(float), (const float*)(lb + w.t_q),
                 (const float*)(lb + w.t_kv), lb + w.t_att, N, D, heads, Tk, p.off);
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_to_kernel launch"));
        if (fused_tokens) {  // inference: nothing reads the hidden rows again; to.out rides in front of the block (stage 0 of the same launch)
            PFM_TRY(dense_block(p, p.M, N, lb + w.t_mid, D, Hd, Tl.d1, &Tl.norm2, jb + (int64_t)(2 + 2 * l) * Hd, Tl.d2, &Tl.d_norm, lb + w.t_dh, lb + w.t_out,
                                lb + w.t_att, &Tl.out, &Tl.attn_norm, seq));
        } else {
            PFM_TRY(linear(p, p.M, N, lb + w.t_att, D, D, Tl.out, &Tl.attn_norm, D, nullptr, seq, D, lb + w.t_mid, D, 0));
            PFM_TRY(linear(p, p.M, N, lb + w.t_mid, D, D, Tl.d1, &Tl.norm2, Hd, jb + (int64_t)(2 + 2 * l) * Hd, nullptr, 0, lb + w.t_dh, Hd, 1));
            PFM_TRY(linear(p, p.M, N, lb + w.t_dh, Hd, Hd, Tl.d2, &Tl.d_norm, D, nullptr, lb + w.t_mid, D, lb + w.t_out, D, 0));
        }
        seq = lb + w.t_out;
    }
    // outp_embd: Linear on cat(seq, ctxt) without a LayerNorm in front (FullCrossAttentionEncoder has no final norm)
    PFM_TRY(linear(p, p.M, N, seq, D, D, d.o1, nullptr, Hd, jb + (int64_t)(nb - 1) * Hd, nullptr, 0, ws + w.oh, Hd, 1));
    HeadArgs h = head_tpl;
    h.A = ws + w.oh; h.blob = p.blob;
    h.gamma = d.o_norm.gamma; h.beta = d.o_norm.beta; h.W = d.o2.W; h.b = d.o2.b;
    h.M = p.M; h.Hd = Hd; h.F = d.features; h.eps = d.ln_eps;
    h.rowsrc = p.rowsrc; h.m_dev = p.m_dev;
    if (p.rowsrc && !h.base) {  // raw field: the rows the compacted evaluation never touches are 0
        const int64_t n = (int64_t)p.M * d.features;
        for (float* dst : {h.dst, h.v_out})
            if (dst)
                hipLaunchKernelGGL(rows_fill_masked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, mask, p.cnt,
                                   (const float*)nullptr, dst, (int64_t)p.M, N, d.features, 0);
    }
    const dim3 hg((p.M + 15) / 16), hb(256);
    switch (Hd / 64) {
        case 2: hipLaunchKernelGGL(tf_head_kernel<2>, hg, hb, 0, p.s, h); break;
        case 4: hipLaunchKernelGGL(tf_head_kernel<4>, hg, hb, 0, p.s, h); break;
        case 6: hipLaunchKernelGGL(tf_head_kernel<6>, hg, hb, 0, p.s, h); break;
        default: hipLaunchKernelGGL(tf_head_kernel<8>, hg, hb, 0, p.s, h); break;
    }
    return check_hip(hipGetLastError(), "tf_head_kernel launch (ca)");
}

int make_plan(Plan& p, const pfm_ca_desc* d, const float* blob, float* ws, int n_jets, bool train, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    p.d = d; p.blob = blob; p.ws = ws; p.n_jets = n_jets; p.M = n_jets * d->n_points; p.Mt = n_jets * d->tokens;
    p.s = (hipStream_t)stream;
    p.w = make_ws(*d, n_jets, train);
    p.temb_k = (d->flags & PFM_CA_F_TEMB_GIVEN) ? 1 : 0;  // rows [jet][T]; the samplers switch to their [T][evaluations] table
    return 0;
}

int setup_valid_rows(Plan& p, const float* mask) {
    if (!(p.d->flags & PFM_CA_F_VALID_ROWS) || !mask) return 0;
    if (p.d->n_points <= p.d->tokens) return 0;  // linear() tells particle rows from token rows by their count per jet
    const RowMaps m = build_row_maps(reinterpret_cast<int*>(p.ws + p.w.imaps), mask, p.n_jets, p.d->n_points, p.s);
    p.rowsrc = m.rowsrc; p.rowjet = m.rowjet; p.off = m.off; p.m_dev = m.m_dev; p.cnt = m.cnt; p.order = m.order;
    return check_hip(hipGetLastError(), "row compaction launch");
}

// ---- backward ------------------------------------------------------------------------------------------
struct Bs {
    int64_t dv, gh, gh2, gseq, gtok, ga, gat, gkv, gkvt, kvpart, gq, gqt, gatt, gattt, ght, rstat, djb, dctxt, dhn, dhnx, dpre, hn, dwpart, total;
};

// particles per workgroup of ca_attn_to_bwd_kernel: its p / ds staging ([rows][heads][2 TK] floats) stays under 24 KB
// (with the token keys / values, <= 32 KB, inside the 64 KB a kernel gets without opting in to more)
inline int to_bwd_rows(const pfm_ca_desc& d) {
    const int tk = (d.head_dim == 8 && d.tokens > 4) ? 8 : 4;
    int r = 6144 / (d.heads * 2 * tk);
    return r > 64 ? 64 : r;
}

Bs make_bs(const pfm_ca_desc& d, int n_jets) {
    Bs b;
    const int64_t M = (int64_t)n_jets * d.n_points, Mt = (int64_t)n_jets * d.tokens, D = d.model_dim, Hd = d.hidden;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    b.dv = take(M * d.features);
    const int64_t Mx = M > Mt ? M : Mt;  // gh2 and gkv double as scratch of the token rows and of the head backward
    b.gh = take(M * Hd); b.gh2 = take(Mx * Hd);
    b.gseq = take(M * D); b.gtok = take(Mt * D);
    b.ga = take(M * D); b.gat = take(Mt * D);
    b.gkv = take(M * (2 * D > Hd ? 2 * D : Hd)); b.gkvt = take(Mt * 2 * D);
    b.kvpart = take(Mt * 2 * D * ((d.n_points + to_bwd_rows(d) - 1) / to_bwd_rows(d)));
    b.gq = take(M * D); b.gqt = take(Mt * D);
    b.gatt = take(M * D); b.gattt = take(Mt * D);
    b.ght = take(Mt * Hd);
    b.rstat = take(Mx * 2);
    b.djb = take((int64_t)n_jets * (2 * d.layers + 2) * Hd);
    b.dctxt = take((int64_t)n_jets * d.ctxt_dim);
    b.dhn = take((int64_t)n_jets * d.ctxt_hidden); b.dhnx = take((int64_t)n_jets * d.ctxt_hidden);
    b.dpre = take((int64_t)n_jets * d.ctxt_hidden); b.hn = take((int64_t)n_jets * d.ctxt_hidden);
    b.dwpart = take((int64_t)DW_MAX_PARTS * 16384);
    b.total = o;
    return b;
}

struct Bwd {
    Plan p;
    float *gblob, *sc;
    Bs b;
    float* dy = nullptr;  // pfm_ca_fm_loss_backward_dx: also d loss / d y (chains of flows)

    int colsum(const float* Z, int ldz, int NO, int64_t rows, int group, const float* X, int F, float* jet_out, int64_t gb) const {
        ColsumArgs a;
        a.Z = Z; a.X = X; a.jet_out = jet_out; a.gblob = gblob; a.gb = gb;
        a.jet_stride = (int64_t)(2 * p.d->layers + 2) * p.d->hidden;
        a.ldz = ldz; a.NO = NO; a.N = group; a.F = F; a.rows = rows;
        const int ny = X ? F : 1, ngrp = (int)((rows + group - 1) / group);
        a.part = gb >= 0 ? sc + b.dwpart : nullptr;  // (free between two dW launches; every launch on p.s: stream order)
        hipLaunchKernelGGL(tf_colsum_kernel, dim3((unsigned)ngrp, ny, (NO + 767) / 768), dim3(256), 0, p.s, a);
        if (gb >= 0) launch_ordered_sum(p.s, a.part, ngrp, (int64_t)ny * NO, ny * NO, gblob + gb, ny * NO, nullptr);
        return check_hip(hipGetLastError(), "tf_colsum_kernel launch (ca)");
    }
    // dW += Z^T LN(A) over `rows` rows (ln == nullptr: no LayerNorm prologue)
    int dw(int rows, const float* Z, int NO, const float* A, int K, const pfm_tf_norm* ln, int64_t gW) const {
        float* st = sc + b.rstat;
        if (ln) {
            const dim3 g((rows + 15) / 16), bl(256);
            switch (K / 64) {
                case 2: hipLaunchKernelGGL(tf_rowstats_kernel<2>, g, bl, 0, p.s, A, rows, p.d->ln_eps, st); break;
                case 4: hipLaunchKernelGGL(tf_rowstats_kernel<4>, g, bl, 0, p.s, A, rows, p.d->ln_eps, st); break;
                case 6: hipLaunchKernelGGL(tf_rowstats_kernel<6>, g, bl, 0, p.s, A, rows, p.d->ln_eps, st); break;
                default: hipLaunchKernelGGL(tf_rowstats_kernel<8>, g, bl, 0, p.s, A, rows, p.d->ln_eps, st); break;
            }
        }
        DwArgs a;
        a.Z = Z; a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.stats = ln ? st : nullptr; a.blob = p.blob; a.part = sc + b.dwpart;
        a.gamma = ln ? ln->gamma : -1; a.beta = ln ? ln->beta : -1;
        a.ldz = NO; a.lda = K; a.M = rows; a.NO = NO; a.K = K;
        a.row_tiles = (rows + BM - 1) / BM;
        const int tiles = ((NO + 127) / 128) * ((K + 127) / 128);
        const int ns = dw_splits(a.row_tiles, tiles, num_cus());
        a.nsplit = ns;
        int rc;
        hipLaunchKernelGGL(tf_dw_kernel, dim3(tiles * ns), dim3(LT), 2 * 64 * DWS * sizeof(float), p.s, a);
        if ((rc = check_hip(hipGetLastError(), "tf_dw_kernel launch (ca)"))) return rc;
        launch_dw_reduce(p.s, a.part, gblob, gW, NO, K, tiles, ns);
        return check_hip(hipGetLastError(), "tf_dw_reduce_kernel launch (ca)");
    }
    int dx(int rows, const float* Z, int NO, const pfm_tf_lin& lin, int K, float* out) const {
        pfm_tf_lin t = lin;
        t.W = lin.WT;
        t.b = -1;
        return linear(p, rows, 1, Z, NO, NO, t, nullptr, K, nullptr, nullptr, 0, out, K, 0);
    }
    int lnbwd(int rows, const float* A, int K, const float* G, const float* add, float* out, const pfm_tf_norm& ln, bool act) const {
        LnBwdArgs a{};
        a.A = A; a.G = G; a.add = add; a.out = out; a.blob = p.blob; a.gblob = gblob;
        a.gamma = ln.gamma; a.beta = ln.beta; a.M = rows; a.K = K; a.act = act ? 1 : 0;
        a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
        a.part = sc + b.dwpart;  // d gamma | d beta: per-workgroup partials, summed in block order (no atomics)
        const dim3 g((rows + 63) / 64), bl(256);
        switch (K / 64) {
            case 2: hipLaunchKernelGGL(tf_ln_bwd_kernel<2>, g, bl, 0, p.s, a); break;
            case 4: hipLaunchKernelGGL(tf_ln_bwd_kernel<4>, g, bl, 0, p.s, a); break;
            case 6: hipLaunchKernelGGL(tf_ln_bwd_kernel<6>, g, bl, 0, p.s, a); break;
            default: hipLaunchKernelGGL(tf_ln_bwd_kernel<8>, g, bl, 0, p.s, a); break;
        }
        launch_ordered_sum(p.s, a.part, (int)g.x, 2 * (int64_t)K, 2 * K, gblob + ln.gamma, K, gblob + ln.beta);
        return check_hip(hipGetLastError(), "tf_ln_bwd_kernel launch (ca)");
    }
    // the block  q <- q + out(LN_a(att));  q <- q + d2(LN_d(lrelu(d1(LN2(q)) + jet bias)))  of one cross-attention layer,
    // backwards: in: g = d loss / d q_out (rows x D); out: g = d loss / d q_in excluding the query path of the attention,
    // gatt = d loss / d att.  `per_jet` rows per jet, jb_row = index of the layer's jet-bias row.
    int layer_tail_bwd(int rows, int per_jet, const pfm_ca_layer& L, int jb_row, const float* q_in, const float* att, const float* mid,
                       const float* dh, float* g, float* gh_, float* gh2_, float* ga_, float* gatt_) const {
        const int D = p.d->model_dim, Hd = p.d->hidden;
        (void)q_in;
        int rc;
        PFM_TRY(colsum(g, D, D, rows, per_jet, nullptr, 0, nullptr, L.d2.b));
        PFM_TRY(dw(rows, g, D, dh, Hd, &L.d_norm, L.d2.W));
        PFM_TRY(dx(rows, g, D, L.d2, Hd, gh2_));
        PFM_TRY(lnbwd(rows, dh, Hd, gh2_, nullptr, gh_, L.d_norm, true));
        PFM_TRY(colsum(gh_, Hd, Hd, rows, per_jet, nullptr, 0, sc + b.djb + (int64_t)jb_row * Hd, -1));
        PFM_TRY(dw(rows, gh_, Hd, mid, D, &L.norm2, L.d1.W));
        PFM_TRY(dx(rows, gh_, Hd, L.d1, D, ga_));
        PFM_TRY(lnbwd(rows, mid, D, ga_, g, g, L.norm2, false));  // g = d loss / d mid
        PFM_TRY(colsum(g, D, D, rows, per_jet, nullptr, 0, nullptr, L.out.b));
        PFM_TRY(dw(rows, g, D, att, D, &L.attn_norm, L.out.W));
        PFM_TRY(dx(rows, g, D, L.out, D, ga_));
        PFM_TRY(lnbwd(rows, att, D, ga_, nullptr, gatt_, L.attn_norm, false));
        return 0;
    }
    // a projection  y = lin(LN(x))  backwards: gy -> bias / weight gradients, and g_x += LN backward of (gy W)
    int proj_bwd(int rows, int per_jet, const pfm_tf_lin& lin, const pfm_tf_norm& ln, int NO, const float* x, const float* gy, float* tmp,
                 float* gx) const {
        const int D = p.d->model_dim;
        int rc;
        PFM_TRY(colsum(gy, NO, NO, rows, per_jet, nullptr, 0, nullptr, lin.b));
        PFM_TRY(dw(rows, gy, NO, x, D, &ln, lin.W));
        PFM_TRY(dx(rows, gy, NO, lin, D, tmp));
        return lnbwd(rows, x, D, tmp, gx, gx, ln, false);
    }
};

int run_backward(const Bwd& B, const float* cond, const float* mask, const float* y, const float* u, const float* v, const float* gscale) {
    const Plan& p = B.p;
    const pfm_ca_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    float* sc = B.sc;
    const Bs& b = B.b;
    const int D = d.model_dim, Hd = d.hidden, nb = 2 * d.layers + 2, F = d.features, N = d.n_points, Tk = d.tokens, heads = d.heads;
    const int rpb = to_bwd_rows(d);
    float *gh = sc + b.gh, *gh2 = sc + b.gh2, *gseq = sc + b.gseq, *gtok = sc + b.gtok, *ga = sc + b.ga, *gat = sc + b.gat;
    float *gkv = sc + b.gkv, *gkvt = sc + b.gkvt, *gq = sc + b.gq, *gqt = sc + b.gqt, *gatt = sc + b.gatt, *gattt = sc + b.gattt, *ght = sc + b.ght;
    float* djb = sc + b.djb;
    int rc;
    auto layer_base = [&](int l) { return ws + w.layer0 + w.lstride * l; };
    const float* seqL = layer_base(d.layers - 1) + w.t_out;
    // ---- output head + outp_embd input block (no LayerNorm in front of it) ----
    {
        HeadBwdArgs a;
        a.A = ws + w.oh; a.v = v; a.u = u; a.gscale = gscale; a.dv = sc + b.dv; a.dn = gkv; a.nout = gh2;  // gkv: M x 2D >= M x Hd scratch
        a.blob = p.blob; a.gblob = B.gblob;
        a.gamma = d.o_norm.gamma; a.beta = d.o_norm.beta; a.W3 = d.o2.W; a.b3 = d.o2.b;
        a.M = p.M; a.K = Hd; a.F = F; a.eps = d.ln_eps;
        a.part = sc + b.dwpart;
        const dim3 g((p.M + 15) / 16), bl(256);
        switch (Hd / 64) {
            case 2: hipLaunchKernelGGL(tf_head_bwd_kernel<2>, g, bl, 0, p.s, a); break;
            case 4: hipLaunchKernelGGL(tf_head_bwd_kernel<4>, g, bl, 0, p.s, a); break;
            case 6: hipLaunchKernelGGL(tf_head_bwd_kernel<6>, g, bl, 0, p.s, a); break;
            default: hipLaunchKernelGGL(tf_head_bwd_kernel<8>, g, bl, 0, p.s, a); break;
        }
        launch_ordered_sum(p.s, a.part, (int)g.x, 16, F, B.gblob + d.o2.b, F, nullptr);
        PFM_TRY(check_hip(hipGetLastError(), "tf_head_bwd_kernel launch (ca)"));
        PFM_TRY(B.colsum(gh2, Hd, Hd, p.M, N, sc + b.dv, F, nullptr, d.o2.W));
        PFM_TRY(B.lnbwd(p.M, ws + w.oh, Hd, gkv, nullptr, gh, d.o_norm, true));
        PFM_TRY(B.colsum(gh, Hd, Hd, p.M, N, nullptr, 0, djb + (int64_t)(nb - 1) * Hd, -1));
        PFM_TRY(B.dw(p.M, gh, Hd, seqL, D, nullptr, d.o1.W));
        PFM_TRY(B.dx(p.M, gh, Hd, d.o1, D, gseq));
    }
    PFM_TRY(check_hip(hipMemsetAsync(gtok, 0, (size_t)p.Mt * D * sizeof(float), p.s), "memset gtok"));
    for (int l = d.layers - 1; l >= 0; --l) {
        float* lb = layer_base(l);
        const pfm_ca_layer& Fl = d.from_layer[l];
        const pfm_ca_layer& Tl = d.to_layer[l];
        const float* seq_in = l ? layer_base(l - 1) + w.t_out : ws + w.seq0;
        const float* tok_in = l ? layer_base(l - 1) + w.f_out : ws + w.tok0;
        const float* tok_out = lb + w.f_out;
        // ---- to-layer: seq_out = f(seq_in, tok_out) ----
        PFM_TRY(B.layer_tail_bwd(p.M, N, Tl, 2 + 2 * l, seq_in, lb + w.t_att, lb + w.t_mid, lb + w.t_dh, gseq, gh, gh2, ga, gatt));
        {
            const int nblk = (N + rpb - 1) / rpb;
            const size_t lds = ((size_t)Tk * 2 * D + (size_t)rpb * heads * 2 * (d.head_dim == 8 && Tk > 4 ? 8 : 4)) * sizeof(float);
            PFM_ATTN(ca_attn_to_bwd_kernel, dim3(p.n_jets, nblk), lds, (const float*)(lb + w.t_q), (const float*)(lb + w.t_kv),
                     (const float*)gatt, gq, sc + b.kvpart, N, D, heads, Tk, rpb);
            const int64_t n = (int64_t)p.Mt * 2 * D;
            hipLaunchKernelGGL(ca_attn_to_bwd_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s,
                               (const float*)(sc + b.kvpart), gkvt, n, Tk * 2 * D, nblk);
        }
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_to_bwd_kernel launch"));
        PFM_TRY(B.proj_bwd(p.M, N, Tl.q, Tl.norm1, D, seq_in, gq, ga, gseq));           // gseq = d loss / d seq_in
        PFM_TRY(B.proj_bwd(p.Mt, Tk, Tl.kv, Tl.norm0, 2 * D, tok_out, gkvt, gat, gtok)); // gtok = d loss / d tok_out
        // ---- from-layer: tok_out = f(tok_in, seq_in) ----
        PFM_TRY(B.layer_tail_bwd(p.Mt, Tk, Fl, 1 + 2 * l, tok_in, lb + w.f_att, lb + w.f_mid, lb + w.f_dh, gtok, ght, gh2, gat, gattt));
        PFM_ATTN(ca_attn_from_bwd_kernel, dim3(p.n_jets, (heads + 3) / 4), 0, (const float*)(lb + w.f_q), (const float*)(lb + w.f_kv), mask,
                 (const float*)(lb + w.f_att), (const float*)gattt, gqt, gkv, N, D, heads, Tk);
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_from_bwd_kernel launch"));
        PFM_TRY(B.proj_bwd(p.Mt, Tk, Fl.q, Fl.norm1, D, tok_in, gqt, gat, gtok));        // gtok = d loss / d tok_in
        PFM_TRY(B.proj_bwd(p.M, N, Fl.kv, Fl.norm0, 2 * D, seq_in, gkv, ga, gseq));      // gseq += keys / values path
    }
    // ---- global tokens: every jet starts from the same rows ----
    PFM_TRY(B.colsum(gtok, Tk * D, Tk * D, p.n_jets, 16, nullptr, 0, nullptr, d.global_tokens));
    // ---- node_embd ----
    PFM_TRY(B.colsum(gseq, D, D, p.M, N, nullptr, 0, nullptr, d.n2.b));
    PFM_TRY(B.dw(p.M, gseq, D, ws + w.h1, Hd, &d.n_norm, d.n2.W));
    PFM_TRY(B.dx(p.M, gseq, D, d.n2, Hd, gh2));
    PFM_TRY(B.lnbwd(p.M, ws + w.h1, Hd, gh2, nullptr, gh, d.n_norm, true));
    PFM_TRY(B.colsum(gh, Hd, Hd, p.M, N, nullptr, 0, djb, -1));
    PFM_TRY(B.colsum(gh, Hd, Hd, p.M, N, y, F, nullptr, d.n1.W));
    if (B.dy) {
        hipLaunchKernelGGL(tf::tf_dy_kernel, dim3((unsigned)((p.M + 15) / 16)), dim3(256), 0, p.s, (const float*)gh, p.blob, d.n1.W, B.dy, (int64_t)p.M, F, Hd);
        PFM_TRY(check_hip(hipGetLastError(), "tf_dy_kernel launch (ca)"));
    }
    // ---- context path ----
    {
        CtxtBwdArgs a;
        a.blob = p.blob; a.djb = djb; a.chid = ws + w.chid;
        a.dctxt = sc + b.dctxt; a.dhn = sc + b.dhn; a.dhnx = sc + b.dhnx; a.dpre = sc + b.dpre; a.hn = sc + b.hn;
        a.CH = d.ctxt_hidden; a.CO = d.ctxt_dim; a.Hd = Hd; a.nb = nb; a.slope = d.neg_slope; a.eps = d.ln_eps;
        a.cg = d.c_norm.gamma; a.cb = d.c_norm.beta; a.c2W = d.c2.W;
        auto lin_of = [&](int c) -> const pfm_tf_lin& {
            if (c == 0) return d.n1;
            if (c == nb - 1) return d.o1;
            return ((c - 1) & 1) ? d.to_layer[(c - 1) / 2].d1 : d.from_layer[(c - 1) / 2].d1;
        };
        for (int c = 0; c < nb; ++c) a.Wc[c] = lin_of(c).Wc;
        hipLaunchKernelGGL(tf_ctxt_bwd_kernel, dim3(p.n_jets), dim3(512), 0, p.s, a);
        PFM_TRY(check_hip(hipGetLastError(), "tf_ctxt_bwd_kernel launch (ca)"));
        CtxtGradIn g;
        g.ctxt = ws + w.ctxt; g.djb = djb; g.temb = ws + w.temb; g.cond = cond;
        g.hn = sc + b.hn; g.dctxt = sc + b.dctxt; g.dhnx = sc + b.dhnx; g.dhn = sc + b.dhn; g.dpre = sc + b.dpre;
        g.gblob = B.gblob; g.n_jets = p.n_jets; g.nb = nb; g.Hd = Hd; g.CO = d.ctxt_dim; g.CH = d.ctxt_hidden; g.T = d.t_dim;
        g.C = d.cond_dim;
        for (int c = 0; c < nb; ++c) { g.gW[c] = lin_of(c).Wc; g.gb[c] = lin_of(c).b; }
        g.n1Wt = d.time_in_input ? d.n1.Wt : -1;
        g.c2W = d.c2.W; g.c2b = d.c2.b; g.cgamma = d.c_norm.gamma; g.cbeta = d.c_norm.beta; g.c1W = d.c1.W; g.c1b = d.c1.b;
        launch_ctxt_param_grads(g, p.s);
        PFM_TRY(check_hip(hipGetLastError(), "context parameter gradient launches"));
    }
    return 0;
}

}  // namespace ca
}  // namespace pfm

using namespace pfm;

extern "C" {

int64_t pfm_ca_workspace_floats(const pfm_ca_desc* d, int32_t n_jets, int32_t train) {
    if (ca::validate(d)) return -1;
    return ca::make_ws(*d, n_jets < 1 ? 1 : n_jets, train != 0).total;
}

int pfm_ca_forward(const pfm_ca_desc* d, const float* blob, const float* t, int32_t t_stride, const float* x, const float* cond,
                   const float* mask, float* v, int32_t n_jets, float* workspace, void* stream) {
    ca::Plan p;
    int rc = ca::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t || !x || !v || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    if ((rc = ca::setup_valid_rows(p, mask))) return rc;
    tf::HeadArgs h{};
    h.dst = v;
    return ca::run_nfe(p, t, t_stride ? 1 : 0, x, cond, mask, h);
}

int pfm_ca_sample_midpoint(const pfm_ca_desc* d, const float* blob, const float* t_eval, const float* dt, int32_t n_steps,
                           const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets, int32_t premask,
                           float* state, float* workspace, void* stream) {
    ca::Plan p;
    int rc = ca::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    if (p.temb_k) p.temb_k = 2 * n_steps;  // t_eval = the embedding table [T][2 n_steps]: evaluation e starts at t_eval + e
    const int64_t n = (int64_t)p.M * d->features;
    float* xs = state;
    float* xm = state + n;
    hipLaunchKernelGGL(tf::tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : nullptr, xs, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    if ((rc = ca::setup_valid_rows(p, mask))) return rc;
    if (p.rowsrc)  // x_mid's padded rows are never read; give them defined values once
        if ((rc = check_hip(hipMemcpyAsync(xm, xs, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_mid"))) return rc;
    auto one_step = [&](const float* t0, const float* t1, const float* h_dt) -> int {
        tf::HeadArgs h{};
        h.base = xs; h.dt = h_dt; h.coef = 0.5f; h.dst = xm;
        if (int e = ca::run_nfe(p, t0, 0, xs, cond, mask, h)) return e;
        h.coef = 1.0f; h.dst = xs;
        return ca::run_nfe(p, t1, 0, xm, cond, mask, h);
    };
    int k = 0;
    // the legacy null stream cannot be captured; two steps or fewer are not worth a graph
    // (a caller-supplied embedding table is addressed through t_eval itself: the replayed step body reads t from a staging slot: no graph)
    tf::ParkedGraph* gs = ((d->flags & PFM_CA_F_GRAPH_STEPS) && p.s != nullptr && n_steps > 2 && !p.temb_k) ? tf::park_graph(p.s) : nullptr;
    if (gs) {
        // step 0 runs directly: whatever the first launch of a kernel does lazily (module load, the 128-row Linear's LDS opt-in)
        // happens outside the capture
        if ((rc = one_step(t_eval, t_eval + 1, dt))) return rc;
        float* slot = p.ws + p.w.step;
        if ((rc = check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slot + 4), 1, 1, p.s), "step counter"))) return rc;
        if ((rc = check_hip(hipStreamBeginCapture(p.s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture"))) return rc;
        hipLaunchKernelGGL(tf::step_args_kernel, dim3(1), dim3(1), 0, p.s, t_eval, dt, slot);
        rc = one_step(slot, slot + 1, slot + 2);
        const hipError_t ce = hipStreamEndCapture(p.s, &gs->graph);  // always end the capture, also after a failed launch
        if (rc == 0) rc = check_hip(ce, "hipStreamEndCapture");
        if (rc == 0) rc = check_hip(hipGraphInstantiate(&gs->exec, gs->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        for (k = 1; rc == 0 && k < n_steps; ++k) rc = check_hip(hipGraphLaunch(gs->exec, p.s), "hipGraphLaunch");
        const hipError_t re = hipEventRecord(gs->done, p.s);  // behind the last launch: retire() waits for it
        if (rc == 0) rc = check_hip(re, "hipEventRecord (graph replay)");
        if (rc) return rc;
    }
    for (; k < n_steps; ++k)
        if ((rc = one_step(t_eval + 2 * k, t_eval + 2 * k + 1, dt + k))) return rc;
    return check_hip(hipMemcpyAsync(x_out, xs, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

int pfm_ca_sample_rk(const pfm_ca_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                     int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets,
                     int32_t premask, float* state, float* workspace, void* stream) {
    ca::Plan p;
    int rc = ca::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (const char* e = tf::rk_tableau_error(tab)) return set_err(PFM_E_BADARG, e);
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf::tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : nullptr, state, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    if ((rc = ca::setup_valid_rows(p, mask))) return rc;
    if (p.temb_k) p.temb_k = n_steps * tab->stages;  // t_eval = the embedding table [T][n_steps * stages]
    rc = tf::sample_rk_rows(*tab, t_eval, dt, n_steps, state, n, p.s, [&](const float* t, const float* x, float* v) {
        tf::HeadArgs h{};
        h.dst = v;
        return ca::run_nfe(p, t, 0, x, cond, mask, h);
    });
    if (rc) return rc;
    if ((rc = check_hip(hipGetLastError(), "tf_rk_combine_kernel launch"))) return rc;
    return check_hip(hipMemcpyAsync(x_out, state, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

int pfm_ca_fm_loss_forward(const pfm_ca_desc* d, const float* blob, int32_t kind, float sigma, const float* t, const float* x,
                           const float* a, const float* b, const float* cond, const float* mask, float* y_out, float* u_out,
                           float* v_out, float* loss_sums, int32_t n_jets, float* workspace, void* stream) {
    ca::Plan p;
    int rc = ca::make_plan(p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    if (!blob || !t || !x || !a || !y_out || !u_out || !v_out || !loss_sums || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (kind == 1 && !b) return set_err(PFM_E_BADARG, "CFM needs eps");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf::tf_yu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, kind, sigma, t, x, a, b, mask, y_out, u_out, n,
                       d->n_points * d->features, d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_yu_kernel launch"))) return rc;
    tf::HeadArgs h{};
    h.dst = v_out;
    if ((rc = ca::run_nfe(p, t, 1, y_out, cond, mask, h))) return rc;
    hipLaunchKernelGGL(tf::tf_loss_kernel, dim3(1), dim3(tf::LOSS_T), 0, p.s, (const float*)v_out, (const float*)u_out, mask, loss_sums, n,
                       (int64_t)p.M);
    return check_hip(hipGetLastError(), "tf_loss_kernel launch");
}

int64_t pfm_ca_backward_scratch_floats(const pfm_ca_desc* d, int32_t n_jets) {
    if (ca::validate(d)) return -1;
    return ca::make_bs(*d, n_jets < 1 ? 1 : n_jets).total;
}

static int ca_loss_backward(const pfm_ca_desc* d, const float* blob, const float* cond, const float* mask, const float* y,
                            const float* u, const float* v, const float* gscale, float* gblob, float* grad_y, int32_t n_jets, float* workspace,
                            float* scratch, void* stream) {
    ca::Bwd B;
    B.dy = grad_y;
    int rc = ca::make_plan(B.p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !y || !u || !v || !gscale || !gblob || !workspace || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    if (d->n2.WT < 0) return set_err(PFM_E_BADARG, "blob was packed without the transposed (backward) weight copies");
    B.gblob = gblob;
    B.sc = scratch;
    B.b = ca::make_bs(*d, n_jets);
    return ca::run_backward(B, cond, mask, y, u, v, gscale);
}

int pfm_ca_fm_loss_backward(const pfm_ca_desc* d, const float* blob, const float* cond, const float* mask, const float* y,
                            const float* u, const float* v, const float* gscale, float* gblob, int32_t n_jets, float* workspace,
                            float* scratch, void* stream) {
    return ca_loss_backward(d, blob, cond, mask, y, u, v, gscale, gblob, nullptr, n_jets, workspace, scratch, stream);
}

int pfm_ca_fm_loss_backward_dx(const pfm_ca_desc* d, const float* blob, const float* cond, const float* mask, const float* y,
                               const float* u, const float* v, const float* gscale, float* gblob, float* grad_y, int32_t n_jets,
                               float* workspace, float* scratch, void* stream) {
    if (!grad_y) return set_err(PFM_E_BADARG, "grad_y is NULL");
    return ca_loss_backward(d, blob, cond, mask, y, u, v, gscale, gblob, grad_y, n_jets, workspace, scratch, stream);
}

int pfm_ca_backward_dtemb(const pfm_ca_desc* d, const float* blob, const float* scratch, int32_t n_jets, float* dtemb, void* stream) {
    int rc = ca::validate(d);
    if (rc) return rc;
    if (!(d->flags & PFM_CA_F_TEMB_GIVEN)) return set_err(PFM_E_BADARG, "pfm_ca_backward_dtemb: the descriptor has no PFM_CA_F_TEMB_GIVEN");
    if (n_jets <= 0) return 0;
    if (!blob || !scratch || !dtemb) return set_err(PFM_E_BADARG, "NULL device pointer");
    const ca::Bs b = ca::make_bs(*d, n_jets);
    hipLaunchKernelGGL(tf::tf_dtemb_kernel, dim3(n_jets), dim3(64), 0, (hipStream_t)stream, blob, scratch + b.dpre, scratch + b.djb, dtemb,
                       d->c1.W, d->time_in_input ? d->n1.Wt : (int64_t)-1, d->t_dim, d->ctxt_hidden, d->hidden,
                       (int64_t)(2 * d->layers + 2) * d->hidden);
    return check_hip(hipGetLastError(), "tf_dtemb_kernel launch (ca)");
}

}  // extern "C"
