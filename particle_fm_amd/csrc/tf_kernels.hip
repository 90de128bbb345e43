// gfx950 kernels + C ABI (include/pfm_tf.h): Full-Transformer vector field, midpoint sampler, FM/CFM loss.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "tf_fwd.h"
#include "tf_bwd.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

namespace tf {

int validate(const pfm_tf_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_TF_ABI_VERSION) return set_err(PFM_E_BADARG, "tf desc.abi_version mismatch");
    if (d->model_dim < 128 || d->model_dim > MAXK || d->model_dim % 128)
        return set_err(PFM_E_BADARG, "model_dim must be a multiple of 128 in 128..512");
    if (d->hidden < 128 || d->hidden > MAXK || d->hidden % 128)
        return set_err(PFM_E_BADARG, "hidden must be a multiple of 128 in 128..512");
    if (d->head_dim != HD || d->heads * HD != d->model_dim)
        return set_err(PFM_E_BADARG, "this build is specialised for head_dim = 16 (heads = model_dim / 16)");
    if (d->layers < 1 || d->layers > PFM_TF_MAX_LAYERS) return set_err(PFM_E_BADARG, "layers out of range");
    if (d->features < 1 || d->features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    if (d->t_dim < 1 || d->t_dim > 64) return set_err(PFM_E_BADARG, "t_dim must be in 1..64");
    if (d->cond_dim < 0 || d->cond_dim > 16) return set_err(PFM_E_BADARG, "cond_dim must be in 0..16");
    if (d->ctxt_dim < 4 || d->ctxt_dim > 64 || d->ctxt_dim % 4) return set_err(PFM_E_BADARG, "ctxt_dim must be a multiple of 4 in 4..64");
    if (d->ctxt_hidden < 4 || d->ctxt_hidden > 512 || d->ctxt_hidden % 4)
        return set_err(PFM_E_BADARG, "ctxt_hidden must be a multiple of 4 in 4..512");
    if (d->n_points < 1 || d->n_points > 512) return set_err(PFM_E_BADARG, "n_points must be in 1..512");
    return 0;
}

struct Plan {
    const pfm_tf_desc* d;
    const float* blob;
    float* ws;
    Ws w;
    int n_jets, M;
    hipStream_t s;
    int temb_k = 0;  // PFM_TF_F_TEMB_GIVEN: floats between the elements of a time-embedding row in the `t` argument (0: `t` holds times)
    // valid-rows-only evaluation (PFM_TF_F_VALID_ROWS, inference): rows are the valid particles in (jet, particle) order
    const int *rowsrc = nullptr, *rowjet = nullptr, *off = nullptr, *m_dev = nullptr, *cnt = nullptr, *order = nullptr;
};

int launch_linear(const Plan& p, const float* A, int lda, int K, const pfm_tf_lin& lin, const pfm_tf_norm* ln, int NO,
                  const float* jb, const float* R, int ldr, float* out, int ldo, bool act) {
    LinArgs a;
    a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.blob = p.blob; a.jb = jb; a.R = R; a.Y = nullptr; a.ldy = 0; a.rowjet = jb ? p.rowjet : nullptr; a.m_dev = p.m_dev; a.part = nullptr; a.ksplit = 1; a.out = out;
    a.blob_floats = p.d->blob_floats; a.W = lin.W; a.b = lin.b;
    a.gamma = ln ? ln->gamma : -1; a.beta = ln ? ln->beta : -1;
    a.jb_stride = (int64_t)(p.d->layers + 2) * p.d->hidden;
    a.lda = lda; a.ldr = ldr; a.ldo = ldo; a.M = p.M; a.K = K; a.NO = NO; a.N = p.d->n_points; a.act = act ? 1 : 0;
    a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
    const int ni = (ln && ln->gamma >= 0) ? K / 64 : 0;
    if (launch_linear_kernel(a, ni, (p.d->flags & PFM_TF_F_F16X3) ? 1 : ((p.d->flags & PFM_TF_F_BF16) ? 2 : 0), num_cus(), p.s))
        return set_err(PFM_E_BADARG, "LayerNorm width must be 128, 256, 384 or 512");
    return check_hip(hipGetLastError(), "tf_linear_kernel launch");
}

int launch_attn(const Plan& p, const float* qkv, const float* mask, float* out) {
    const int N = p.d->n_points, D = p.d->model_dim, heads = p.d->heads;
    const size_t lds = (size_t)attn_lds_floats(N) * sizeof(float);
    const int nkt = attn_np32(N) / 16;
    const dim3 grid(p.n_jets * heads), block(256);
    if (nkt <= 12)
        hipLaunchKernelGGL(tf_attn_kernel<12>, grid, block, lds, p.s, qkv, mask, out, N, D, heads, p.off, p.order);
    else if (nkt <= 18)
        hipLaunchKernelGGL(tf_attn_kernel<18>, grid, block, lds, p.s, qkv, mask, out, N, D, heads, p.off, p.order);
    else
        hipLaunchKernelGGL(tf_attn_kernel<32>, grid, block, lds, p.s, qkv, mask, out, N, D, heads, p.off, p.order);
    return check_hip(hipGetLastError(), "tf_attn_kernel launch");
}

int set_attn_lds() {
    static bool done = false;
    if (done) return 0;
    const int big = attn_lds_floats(512) * 4;
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(tf_attn_kernel<32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, big), "hipFuncSetAttribute(attn)");
    if (rc) return rc;
    rc = check_hip((hipError_t)attn_bwd_set_lds(), "hipFuncSetAttribute(attn backward)");
    if (rc) return rc;
    done = true;
    return 0;
}

// One evaluation of the field for all jets.  x: network input [M][F].  The head either writes v (head.dst = v,
// base = nullptr) or applies a state update.
int run_nfe(const Plan& p, const float* t, int t_stride, const float* x, const float* cond, const float* mask,
            const HeadArgs& head_tpl) {
    const pfm_tf_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    const int D = d.model_dim, Hd = d.hidden, nb = d.layers + 2;
    int rc;
    {
        CtxtArgs a;
        a.blob = p.blob; a.t = t; a.cond = cond;
        a.temb = ws + w.temb; a.chid = ws + w.chid; a.ctxt = ws + w.ctxt; a.jb = ws + w.jb;
        a.t_stride = p.temb_k ? (t_stride ? d.t_dim : 0) : t_stride; a.temb_k = p.temb_k;
        a.T = d.t_dim; a.C = d.cond_dim; a.CH = d.ctxt_hidden; a.CO = d.ctxt_dim; a.Hd = Hd; a.nb = nb;
        a.slope = d.neg_slope; a.eps = d.ln_eps; a.sincos = (d.flags & PFM_TF_F_TEMB_SINCOS) ? 1 : 0;
        a.freqs = d.freqs; a.c1W = d.c1.W; a.c1b = d.c1.b; a.cg = d.c_norm.gamma; a.cb = d.c_norm.beta;
        a.c2W = d.c2.W; a.c2b = d.c2.b; a.n1Wt = d.time_in_input ? d.n1.Wt : -1;
        a.Wc[0] = d.n1.Wc; a.bb[0] = d.n1.b;
        for (int l = 0; l < d.layers; ++l) { a.Wc[1 + l] = d.layer[l].d1.Wc; a.bb[1 + l] = d.layer[l].d1.b; }
        a.Wc[nb - 1] = d.o1.Wc; a.bb[nb - 1] = d.o1.b;
        launch_ctxt(a, p.n_jets, p.s);
        if ((rc = check_hip(hipGetLastError(), "tf_ctxt_kernel launch"))) return rc;
    }
    const float* jb = ws + w.jb;
    hipLaunchKernelGGL(tf_embed_kernel, dim3((p.M + 31) / 32), dim3(256), 0, p.s, p.blob, d.n1.W, x, jb, (int64_t)nb * Hd,
                       ws + w.h1, p.M, d.n_points, d.features, Hd, d.neg_slope, p.rowsrc, p.rowjet, p.m_dev);
    if ((rc = check_hip(hipGetLastError(), "tf_embed_kernel launch"))) return rc;
    if ((rc = launch_linear(p, ws + w.h1, Hd, Hd, d.n2, &d.n_norm, D, nullptr, nullptr, 0, ws + w.x0, D, false))) return rc;
    const float* xin = ws + w.x0;
    for (int l = 0; l < d.layers; ++l) {
        const pfm_tf_layer& L = d.layer[l];
        float* lb = ws + w.layer0 + w.lstride * l;
        float *qkv = lb + w.o_qkv, *att = lb + w.o_att, *xmid = lb + w.o_xmid, *dh = lb + w.o_dh, *xout = lb + w.o_xout;
        if ((rc = launch_linear(p, xin, D, D, L.qkv, &L.norm1, 3 * D, nullptr, nullptr, 0, qkv, 3 * D, false))) return rc;
        if ((rc = launch_attn(p, qkv, mask, att))) return rc;
        if ((rc = launch_linear(p, att, D, D, L.out, &L.attn_norm, D, nullptr, xin, D, xmid, D, false))) return rc;
        if ((rc = launch_linear(p, xmid, D, D, L.d1, &L.norm2, Hd, jb + (int64_t)(1 + l) * Hd, nullptr, 0, dh, Hd, true))) return rc;
        if ((rc = launch_linear(p, dh, Hd, Hd, L.d2, &L.d_norm, D, nullptr, xmid, D, xout, D, false))) return rc;
        xin = xout;
    }
    if ((rc = launch_linear(p, xin, D, D, d.o1, &d.final_norm, Hd, jb + (int64_t)(nb - 1) * Hd, nullptr, 0, ws + w.oh, Hd, true)))
        return rc;
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
                                   (const float*)nullptr, dst, (int64_t)p.M, d.n_points, d.features, 0);
    }
    const dim3 hg((p.M + 15) / 16), hb(256);
    switch (Hd / 64) {
        case 2: hipLaunchKernelGGL(tf_head_kernel<2>, hg, hb, 0, p.s, h); break;
        case 4: hipLaunchKernelGGL(tf_head_kernel<4>, hg, hb, 0, p.s, h); break;
        case 6: hipLaunchKernelGGL(tf_head_kernel<6>, hg, hb, 0, p.s, h); break;
        default: hipLaunchKernelGGL(tf_head_kernel<8>, hg, hb, 0, p.s, h); break;
    }
    return check_hip(hipGetLastError(), "tf_head_kernel launch");
}

int make_plan(Plan& p, const pfm_tf_desc* d, const float* blob, float* ws, int n_jets, bool train, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if ((rc = set_attn_lds())) return rc;
    p.d = d; p.blob = blob; p.ws = ws; p.n_jets = n_jets; p.M = n_jets * d->n_points; p.s = (hipStream_t)stream;
    p.w = make_ws(*d, n_jets, train);
    p.temb_k = (d->flags & PFM_TF_F_TEMB_GIVEN) ? 1 : 0;  // rows [jet][T]; the samplers switch to their [T][evaluations] table
    return 0;
}

// PFM_TF_F_VALID_ROWS: evaluate the valid particles only (the mask is constant over an ODE solve: built once per call)
int setup_valid_rows(Plan& p, const float* mask) {
    if (!(p.d->flags & PFM_TF_F_VALID_ROWS) || !mask) return 0;
    const RowMaps m = build_row_maps(reinterpret_cast<int*>(p.ws + p.w.imaps), mask, p.n_jets, p.d->n_points, p.s);
    p.rowsrc = m.rowsrc; p.rowjet = m.rowjet; p.off = m.off; p.m_dev = m.m_dev; p.cnt = m.cnt; p.order = m.order;
    return check_hip(hipGetLastError(), "row compaction launch");
}


// ---- backward scratch (floats) -------------------------------------------------------------------

struct Bs {
    int64_t dv, gh, gh2, gx, ga, gqkv, gatt, stats, rstat, djb, dctxt, dhn, dhnx, dpre, hn, dwpart, total;
};

Bs make_bs(const pfm_tf_desc& d, int n_jets) {
    Bs b;
    const int64_t M = (int64_t)n_jets * d.n_points, D = d.model_dim, Hd = d.hidden;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    b.dv = take(M * d.features);
    b.gh = take(M * Hd);
    b.gh2 = take(M * Hd);
    b.gx = take(M * D);
    b.ga = take(M * D);
    b.gqkv = take(M * (3 * D > Hd ? 3 * D : Hd));  // also the head's M x hidden scratch
    b.gatt = take(M * D);
    b.stats = take((int64_t)n_jets * d.heads * 3 * attn_np16(d.n_points));
    b.rstat = take(M * 2);
    b.djb = take((int64_t)n_jets * (d.layers + 2) * Hd);
    b.dctxt = take((int64_t)n_jets * d.ctxt_dim);
    b.dhn = take((int64_t)n_jets * d.ctxt_hidden);
    b.dhnx = take((int64_t)n_jets * d.ctxt_hidden);
    b.dpre = take((int64_t)n_jets * d.ctxt_hidden);
    b.hn = take((int64_t)n_jets * d.ctxt_hidden);
    b.dwpart = take((int64_t)DW_MAX_PARTS * 16384);
    b.total = o;
    return b;
}

struct Bwd {
    Plan p;
    float* gblob;
    float* sc;
    Bs b;
    float* dy = nullptr;  // pfm_tf_fm_loss_backward_dx: also d loss / d y (chains of flows)

    int colsum(const float* Z, int ldz, int NO, const float* X, int F, float* jet_out, int64_t gb) const {
        ColsumArgs a;
        a.Z = Z; a.X = X; a.jet_out = jet_out; a.gblob = gblob; a.gb = gb;
        a.jet_stride = (int64_t)(p.d->layers + 2) * p.d->hidden;
        a.ldz = ldz; a.NO = NO; a.N = p.d->n_points; a.F = F; a.rows = 0;
        const int ny = X ? F : 1;
        a.part = gb >= 0 ? sc + b.dwpart : nullptr;  // (free between two dW launches; every launch on p.s: stream order)
        hipLaunchKernelGGL(tf_colsum_kernel, dim3(p.n_jets, ny, (NO + 767) / 768), dim3(256), 0, p.s, a);
        if (gb >= 0) launch_ordered_sum(p.s, a.part, p.n_jets, (int64_t)ny * NO, ny * NO, gblob + gb, ny * NO, nullptr);
        return check_hip(hipGetLastError(), "tf_colsum_kernel launch");
    }
    int rowstats(const float* A, int K) const {
        const dim3 g((p.M + 15) / 16), bl(256);
        float* st = sc + b.rstat;
        switch (K / 64) {
            case 2: hipLaunchKernelGGL(tf_rowstats_kernel<2>, g, bl, 0, p.s, A, p.M, p.d->ln_eps, st); break;
            case 4: hipLaunchKernelGGL(tf_rowstats_kernel<4>, g, bl, 0, p.s, A, p.M, p.d->ln_eps, st); break;
            case 6: hipLaunchKernelGGL(tf_rowstats_kernel<6>, g, bl, 0, p.s, A, p.M, p.d->ln_eps, st); break;
            default: hipLaunchKernelGGL(tf_rowstats_kernel<8>, g, bl, 0, p.s, A, p.M, p.d->ln_eps, st); break;
        }
        return check_hip(hipGetLastError(), "tf_rowstats_kernel launch");
    }
    // dW += Z^T LN(A)
    int dw(const float* Z, int NO, const float* A, int K, const pfm_tf_norm& ln, int64_t gW) const {
        int rc = rowstats(A, K);
        if (rc) return rc;
        DwArgs a;
        a.Z = Z; a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.stats = sc + b.rstat; a.blob = p.blob; a.part = sc + b.dwpart;
        a.gamma = ln.gamma; a.beta = ln.beta;
        a.ldz = NO; a.lda = K; a.M = p.M; a.NO = NO; a.K = K;
        a.row_tiles = (p.M + BM - 1) / BM;
        const int tiles = ((NO + 127) / 128) * ((K + 127) / 128);
        const int ns = dw_splits(a.row_tiles, tiles, num_cus());
        a.nsplit = ns;
        hipLaunchKernelGGL(tf_dw_kernel, dim3(tiles * ns), dim3(LT), 2 * 64 * DWS * sizeof(float), p.s, a);
        if ((rc = check_hip(hipGetLastError(), "tf_dw_kernel launch"))) return rc;
        launch_dw_reduce(p.s, a.part, gblob, gW, NO, K, tiles, ns);
        return check_hip(hipGetLastError(), "tf_dw_reduce_kernel launch");
    }
    // out[M][K] = Z[M][NO] W   (gradient w.r.t. the Linear's normalised input)
    int dx(const float* Z, int NO, const pfm_tf_lin& lin, int K, float* out) const {
        pfm_tf_lin t = lin;
        t.W = lin.WT;
        t.b = -1;
        return launch_linear(p, Z, NO, NO, t, nullptr, K, nullptr, nullptr, 0, out, K, false);
    }
    int lnbwd(const float* A, int K, const float* G, const float* add, float* out, const pfm_tf_norm& ln, bool act) const {
        LnBwdArgs a{};
        a.A = A; a.G = G; a.add = add; a.out = out; a.blob = p.blob; a.gblob = gblob;
        a.gamma = ln.gamma; a.beta = ln.beta; a.M = p.M; a.K = K; a.act = act ? 1 : 0;
        a.slope = p.d->neg_slope; a.eps = p.d->ln_eps;
        a.part = sc + b.dwpart;  // d gamma | d beta: per-workgroup partials, summed in block order (no atomics)
        const dim3 g((p.M + 63) / 64), bl(256);
        switch (K / 64) {
            case 2: hipLaunchKernelGGL(tf_ln_bwd_kernel<2>, g, bl, 0, p.s, a); break;
            case 4: hipLaunchKernelGGL(tf_ln_bwd_kernel<4>, g, bl, 0, p.s, a); break;
            case 6: hipLaunchKernelGGL(tf_ln_bwd_kernel<6>, g, bl, 0, p.s, a); break;
            default: hipLaunchKernelGGL(tf_ln_bwd_kernel<8>, g, bl, 0, p.s, a); break;
        }
        launch_ordered_sum(p.s, a.part, (int)g.x, 2 * (int64_t)K, 2 * K, gblob + ln.gamma, K, gblob + ln.beta);
        return check_hip(hipGetLastError(), "tf_ln_bwd_kernel launch");
    }
};

#define PFM_TRY(x) do { if ((rc = (x))) return rc; } while (0)

int run_backward(const Bwd& B, const float* cond, const float* mask, const float* y, const float* u, const float* v,
                 const float* gscale) {
    const Plan& p = B.p;
    const pfm_tf_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    float* sc = B.sc;
    const Bs& b = B.b;
    const int D = d.model_dim, Hd = d.hidden, nb = d.layers + 2, F = d.features;
    float *gh = sc + b.gh, *gh2 = sc + b.gh2, *gx = sc + b.gx, *ga = sc + b.ga, *gqkv = sc + b.gqkv, *gatt = sc + b.gatt;
    float* djb = sc + b.djb;
    int rc;
    // ---- output head: v = LN(oh) W3^T + b3 ----
    {
        HeadBwdArgs a;
        a.A = ws + w.oh; a.v = v; a.u = u; a.gscale = gscale; a.dv = sc + b.dv; a.dn = gqkv; a.nout = gh2;
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
        PFM_TRY(check_hip(hipGetLastError(), "tf_head_bwd_kernel launch"));
        PFM_TRY(B.colsum(gh2, Hd, Hd, sc + b.dv, F, nullptr, d.o2.W));
        PFM_TRY(B.lnbwd(ws + w.oh, Hd, gqkv, nullptr, gh, d.o_norm, true));
    }
    const float* xL = d.layers ? ws + w.layer0 + w.lstride * (d.layers - 1) + w.o_xout : ws + w.x0;
    // ---- outp_embd input block ----
    PFM_TRY(B.colsum(gh, Hd, Hd, nullptr, 0, djb + (int64_t)(nb - 1) * Hd, -1));
    PFM_TRY(B.dw(gh, Hd, xL, D, d.final_norm, d.o1.W));
    PFM_TRY(B.dx(gh, Hd, d.o1, D, ga));
    PFM_TRY(B.lnbwd(xL, D, ga, nullptr, gx, d.final_norm, false));
    // ---- encoder layers, last to first ----
    for (int l = d.layers - 1; l >= 0; --l) {
        const pfm_tf_layer& L = d.layer[l];
        float* lb = ws + w.layer0 + w.lstride * l;
        const float *qkv = lb + w.o_qkv, *att = lb + w.o_att, *xmid = lb + w.o_xmid, *dh = lb + w.o_dh;
        const float* xin = l ? ws + w.layer0 + w.lstride * (l - 1) + w.o_xout : ws + w.x0;
        // x_out = x_mid + d2(LN(dh))
        PFM_TRY(B.colsum(gx, D, D, nullptr, 0, nullptr, L.d2.b));
        PFM_TRY(B.dw(gx, D, dh, Hd, L.d_norm, L.d2.W));
        PFM_TRY(B.dx(gx, D, L.d2, Hd, gh2));
        PFM_TRY(B.lnbwd(dh, Hd, gh2, nullptr, gh, L.d_norm, true));
        // dh = lrelu(d1(LN2(x_mid)) + jet bias)
        PFM_TRY(B.colsum(gh, Hd, Hd, nullptr, 0, djb + (int64_t)(1 + l) * Hd, -1));
        PFM_TRY(B.dw(gh, Hd, xmid, D, L.norm2, L.d1.W));
        PFM_TRY(B.dx(gh, Hd, L.d1, D, ga));
        PFM_TRY(B.lnbwd(xmid, D, ga, gx, gx, L.norm2, false));
        // x_mid = x_in + out(LN(att))
        PFM_TRY(B.colsum(gx, D, D, nullptr, 0, nullptr, L.out.b));
        PFM_TRY(B.dw(gx, D, att, D, L.attn_norm, L.out.W));
        PFM_TRY(B.dx(gx, D, L.out, D, ga));
        PFM_TRY(B.lnbwd(att, D, ga, nullptr, gatt, L.attn_norm, false));
        // attention
        {
            const int N = d.n_points, heads = d.heads;
            const size_t lds = (size_t)attn_bwd_lds_floats(N) * sizeof(float);
            const int nkt = attn_np16(N) / 16;
            const dim3 grid(p.n_jets * heads), block(256);
            float* st = sc + b.stats;
            if (nkt <= 12)
                hipLaunchKernelGGL(tf_attn_bwd_q_kernel<12>, grid, block, lds, p.s, qkv, mask, att, gatt, gqkv, st, N, D, heads);
            else if (nkt <= 18)
                hipLaunchKernelGGL(tf_attn_bwd_q_kernel<18>, grid, block, lds, p.s, qkv, mask, att, gatt, gqkv, st, N, D, heads);
            else
                hipLaunchKernelGGL(tf_attn_bwd_q_kernel<32>, grid, block, lds, p.s, qkv, mask, att, gatt, gqkv, st, N, D, heads);
            PFM_TRY(check_hip(hipGetLastError(), "tf_attn_bwd_q_kernel launch"));
            hipLaunchKernelGGL(tf_attn_bwd_kv_kernel, grid, block, lds, p.s, qkv, mask, gatt, st, gqkv, N, D, heads);
            PFM_TRY(check_hip(hipGetLastError(), "tf_attn_bwd_kv_kernel launch"));
        }
        // qkv = all_linear(LN1(x_in))
        PFM_TRY(B.colsum(gqkv, 3 * D, 3 * D, nullptr, 0, nullptr, L.qkv.b));
        PFM_TRY(B.dw(gqkv, 3 * D, xin, D, L.norm1, L.qkv.W));
        PFM_TRY(B.dx(gqkv, 3 * D, L.qkv, D, ga));
        PFM_TRY(B.lnbwd(xin, D, ga, gx, gx, L.norm1, false));
    }
    // ---- node_embd ----
    PFM_TRY(B.colsum(gx, D, D, nullptr, 0, nullptr, d.n2.b));
    PFM_TRY(B.dw(gx, D, ws + w.h1, Hd, d.n_norm, d.n2.W));
    PFM_TRY(B.dx(gx, D, d.n2, Hd, gh2));
    PFM_TRY(B.lnbwd(ws + w.h1, Hd, gh2, nullptr, gh, d.n_norm, true));
    PFM_TRY(B.colsum(gh, Hd, Hd, nullptr, 0, djb, -1));
    PFM_TRY(B.colsum(gh, Hd, Hd, y, F, nullptr, d.n1.W));
    if (B.dy) {
        hipLaunchKernelGGL(tf_dy_kernel, dim3((unsigned)((p.M + 15) / 16)), dim3(256), 0, p.s, (const float*)gh, p.blob, d.n1.W, B.dy, (int64_t)p.M, F, Hd);
        PFM_TRY(check_hip(hipGetLastError(), "tf_dy_kernel launch"));
    }
    // ---- context path ----
    {
        CtxtBwdArgs a;
        a.blob = p.blob; a.djb = djb; a.chid = ws + w.chid;
        a.dctxt = sc + b.dctxt; a.dhn = sc + b.dhn; a.dhnx = sc + b.dhnx; a.dpre = sc + b.dpre; a.hn = sc + b.hn;
        a.CH = d.ctxt_hidden; a.CO = d.ctxt_dim; a.Hd = Hd; a.nb = nb; a.slope = d.neg_slope; a.eps = d.ln_eps;
        a.cg = d.c_norm.gamma; a.cb = d.c_norm.beta; a.c2W = d.c2.W;
        a.Wc[0] = d.n1.Wc;
        for (int l = 0; l < d.layers; ++l) a.Wc[1 + l] = d.layer[l].d1.Wc;
        a.Wc[nb - 1] = d.o1.Wc;
        hipLaunchKernelGGL(tf_ctxt_bwd_kernel, dim3(p.n_jets), dim3(512), 0, p.s, a);
        PFM_TRY(check_hip(hipGetLastError(), "tf_ctxt_bwd_kernel launch"));
        CtxtGradIn g;
        g.ctxt = ws + w.ctxt; g.djb = djb; g.temb = ws + w.temb; g.cond = cond;
        g.hn = sc + b.hn; g.dctxt = sc + b.dctxt; g.dhnx = sc + b.dhnx; g.dhn = sc + b.dhn; g.dpre = sc + b.dpre;
        g.gblob = B.gblob; g.n_jets = p.n_jets; g.nb = nb; g.Hd = Hd; g.CO = d.ctxt_dim; g.CH = d.ctxt_hidden; g.T = d.t_dim;
        g.C = d.cond_dim;
        for (int c = 0; c < nb; ++c) {
            const pfm_tf_lin& lin = c == 0 ? d.n1 : (c == nb - 1 ? d.o1 : d.layer[c - 1].d1);
            g.gW[c] = lin.Wc; g.gb[c] = lin.b;
        }
        g.n1Wt = d.time_in_input ? d.n1.Wt : -1;
        g.c2W = d.c2.W; g.c2b = d.c2.b; g.cgamma = d.c_norm.gamma; g.cbeta = d.c_norm.beta; g.c1W = d.c1.W; g.c1b = d.c1.b;
        launch_ctxt_param_grads(g, p.s);
        PFM_TRY(check_hip(hipGetLastError(), "context parameter gradient launches"));
    }
    return 0;
}

}  // namespace tf
}  // namespace pfm

using namespace pfm;
using namespace pfm::tf;

extern "C" {

#ifdef PFM_TF_DIAG
// diagnostics build only: copies the stamps of the last stamped panel launch to the host (synchronises the device)
int pfm_tf_diag_stamps(unsigned long long* host, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tf_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

int64_t pfm_tf_workspace_floats(const pfm_tf_desc* d, int32_t n_jets, int32_t train) {
    if (validate(d)) return -1;
    const int n = n_jets < 1 ? 1 : n_jets;
    const int64_t whole = make_ws(*d, n, train != 0).total;
    if (train || n < 2) return whole;
    const int64_t halves = make_ws(*d, n / 2, false).total + make_ws(*d, n - n / 2, false).total;  // two-stream midpoint sampler
    return whole > halves ? whole : halves;
}

int pfm_tf_forward(const pfm_tf_desc* d, const float* blob, const float* t, int32_t t_stride, const float* x,
                   const float* cond, const float* mask, float* v, int32_t n_jets, float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t || !x || !v || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    if ((rc = setup_valid_rows(p, mask))) return rc;
    HeadArgs h{};
    h.dst = v;
    return run_nfe(p, t, t_stride ? 1 : 0, x, cond, mask, h);
}

int pfm_tf_sample_midpoint(const pfm_tf_desc* d, const float* blob, const float* t_eval, const float* dt,
                           int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out,
                           int32_t n_jets, int32_t premask, float* state, float* workspace, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    // Two half-batches on two streams, their launches interleaved evaluation by evaluation (tf_common.h: side_stream): every kernel of an
    // evaluation is row- or jet-local, so the halves never meet, and one half's Linear workgroups stage their rows while the other half's
    // multiply (the workgroups of ONE launch start, stage and multiply together: DESIGN 4b)
    const int n_a = (d->flags & PFM_TF_F_ONE_STREAM) ? 0 : split_point(n_jets, 32);
    SideStream* ss = n_a ? side_stream((hipStream_t)stream) : nullptr;
    const int parts = ss ? 2 : 1;
    Plan p[2];
    float *xs[2], *xm[2];
    const float *cnd[2], *msk[2];
    int64_t n[2], r0[2];
    if (ss && (hipEventRecord(ss->fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess ||
               hipStreamWaitEvent(ss->s2, ss->fork, 0) != hipSuccess))
        return set_err(PFM_E_BADARG, "side stream fork failed");
    // everything between fork and join: an error return still joins (the caller stream must not overtake the side streams' work)
    rc = [&]() -> int {
        int rc = 0;
        for (int i = 0; i < parts; ++i) {
            const int j0 = i ? n_a : 0, nj = parts == 1 ? n_jets : (i ? n_jets - n_a : n_a);
            float* ws = workspace + (i ? make_ws(*d, n_a, false).total : 0);
            if ((rc = make_plan(p[i], d, blob, ws, nj, false, ss ? (void*)(i ? ss->s2 : ss->s) : stream))) return rc;
            if (p[i].temb_k) p[i].temb_k = 2 * n_steps;  // t_eval = the embedding table [T][2 n_steps]: evaluation e starts at t_eval + e
            r0[i] = (int64_t)j0 * d->n_points;
            n[i] = (int64_t)p[i].M * d->features;
            xs[i] = state + 2 * r0[i] * d->features;
            xm[i] = xs[i] + n[i];
            cnd[i] = cond ? cond + (int64_t)j0 * d->cond_dim : nullptr;
            msk[i] = mask ? mask + r0[i] : nullptr;
            hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n[i] + 255) / 256)), dim3(256), 0, p[i].s, z + r0[i] * d->features,
                               premask ? msk[i] : nullptr, xs[i], n[i], d->features);
            if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
            if ((rc = setup_valid_rows(p[i], msk[i]))) return rc;
            if (p[i].rowsrc)  // x_mid's padded rows are never read; give them defined values once
                if ((rc = check_hip(hipMemcpyAsync(xm[i], xs[i], n[i] * sizeof(float), hipMemcpyDeviceToDevice, p[i].s), "copy x_mid"))) return rc;
        }
        for (int k = 0; k < n_steps; ++k)
            for (int stage = 0; stage < 2; ++stage)
                for (int i = 0; i < parts; ++i) {
                    // k1 = f(t_k, x); x_mid = x + 0.5 dt k1; x <- x + dt f(t_k + dt/2, x_mid)
                    HeadArgs h{};
                    h.base = xs[i]; h.dt = dt + k; h.coef = stage ? 1.0f : 0.5f; h.dst = stage ? xs[i] : xm[i];
                    if ((rc = run_nfe(p[i], t_eval + 2 * k + stage, 0, stage ? xm[i] : xs[i], cnd[i], msk[i], h))) return rc;
                }
        for (int i = 0; i < parts; ++i)
            if ((rc = check_hip(hipMemcpyAsync(x_out + r0[i] * d->features, xs[i], n[i] * sizeof(float), hipMemcpyDeviceToDevice, p[i].s),
                                "copy x_out")))
                return rc;
        return 0;
    }();
    side_join(ss, (hipStream_t)stream);
    return rc;
}

// one fixed-step Runge-Kutta call on the jets [j0, j0 + nj) of the caller's arrays, everything on `stream` (state: this part's (2 + stages) x rows x F floats)
static int tf_sample_rk_part(const pfm_tf_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt, int32_t n_steps,
                             const float* z, const float* cond, const float* mask, float* x_out, int j0, int nj, int32_t premask, float* state,
                             float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, nj, false, stream);
    if (rc) return rc;
    if (p.temb_k) p.temb_k = n_steps * tab->stages;  // t_eval = the embedding table [T][n_steps * stages]
    const int64_t r0 = (int64_t)j0 * d->n_points, n = (int64_t)p.M * d->features;
    const float* cnd = cond ? cond + (int64_t)j0 * d->cond_dim : nullptr;
    const float* msk = mask ? mask + r0 : nullptr;
    hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z + r0 * d->features, premask ? msk : nullptr, state, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    if ((rc = setup_valid_rows(p, msk))) return rc;
    rc = sample_rk_rows(*tab, t_eval, dt, n_steps, state, n, p.s, [&](const float* t, const float* x, float* v) {
        HeadArgs h{};
        h.dst = v;
        return run_nfe(p, t, 0, x, cnd, msk, h);
    });
    if (rc) return rc;
    if ((rc = check_hip(hipGetLastError(), "tf_rk_combine_kernel launch"))) return rc;
    return check_hip(hipMemcpyAsync(x_out + r0 * d->features, state, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

int pfm_tf_sample_rk(const pfm_tf_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                     int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets,
                     int32_t premask, float* state, float* workspace, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (const char* e = rk_tableau_error(tab)) return set_err(PFM_E_BADARG, e);
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    // two half-batches on two side streams like the midpoint sampler (here one half is queued after the other: the host enqueues a call several
    // times faster than the GPU runs it, so the halves still run side by side)
    const int n_a = (d->flags & PFM_TF_F_ONE_STREAM) ? 0 : split_point(n_jets, 32);
    SideStream* ss = n_a ? side_stream((hipStream_t)stream) : nullptr;
    if (!ss) return tf_sample_rk_part(d, blob, tab, t_eval, dt, n_steps, z, cond, mask, x_out, 0, n_jets, premask, state, workspace, stream);
    if (hipEventRecord(ss->fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess ||
        hipStreamWaitEvent(ss->s2, ss->fork, 0) != hipSuccess)
        return set_err(PFM_E_BADARG, "side stream fork failed");
    const int64_t per_row = (int64_t)(2 + tab->stages) * d->features;
    rc = tf_sample_rk_part(d, blob, tab, t_eval, dt, n_steps, z, cond, mask, x_out, 0, n_a, premask, state, workspace, ss->s);
    if (!rc)
        rc = tf_sample_rk_part(d, blob, tab, t_eval, dt, n_steps, z, cond, mask, x_out, n_a, n_jets - n_a, premask,
                               state + per_row * n_a * d->n_points, workspace + make_ws(*d, n_a, false).total, ss->s2);
    side_join(ss, (hipStream_t)stream);  // also on an error return: the caller's stream must not overtake the side streams' work
    return rc;
}

int pfm_tf_fm_loss_forward(const pfm_tf_desc* d, const float* blob, int32_t kind, float sigma, const float* t,
                           const float* x, const float* a, const float* b, const float* cond, const float* mask,
                           float* y_out, float* u_out, float* v_out, float* loss_sums, int32_t n_jets,
                           float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    if (!blob || !t || !x || !a || !y_out || !u_out || !v_out || !loss_sums || !workspace)
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if (kind == 1 && !b) return set_err(PFM_E_BADARG, "CFM needs eps");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf_yu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, kind, sigma, t, x, a, b, mask, y_out,
                       u_out, n, d->n_points * d->features, d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_yu_kernel launch"))) return rc;
    HeadArgs h{};
    h.dst = v_out;
    if ((rc = run_nfe(p, t, 1, y_out, cond, mask, h))) return rc;
    hipLaunchKernelGGL(tf_loss_kernel, dim3(1), dim3(LOSS_T), 0, p.s, (const float*)v_out, (const float*)u_out, mask, loss_sums, n,
                       (int64_t)p.M);
    return check_hip(hipGetLastError(), "tf_loss_kernel launch");
}

int64_t pfm_tf_backward_scratch_floats(const pfm_tf_desc* d, int32_t n_jets) {
    if (validate(d)) return -1;
    return make_bs(*d, n_jets < 1 ? 1 : n_jets).total;
}

static int tf_loss_backward(const pfm_tf_desc* d, const float* blob, const float* cond, const float* mask, const float* y, const float* u,
                            const float* v, const float* gscale, float* gblob, float* grad_y, int32_t n_jets, float* workspace, float* scratch,
                            void* stream) {
    Bwd B;
    B.dy = grad_y;
    int rc = make_plan(B.p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !y || !u || !v || !gscale || !gblob || !workspace || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_dim > 0 && !cond) return set_err(PFM_E_BADARG, "cond_dim > 0 but cond is NULL");
    if (d->n2.WT < 0) return set_err(PFM_E_BADARG, "blob was packed without the transposed (backward) weight copies");
    B.gblob = gblob;
    B.sc = scratch;
    B.b = make_bs(*d, n_jets);
    return run_backward(B, cond, mask, y, u, v, gscale);
}

int pfm_tf_fm_loss_backward(const pfm_tf_desc* d, const float* blob, const float* t, const float* cond,
                            const float* mask, const float* y, const float* u, const float* v, const float* gscale,
                            float* gblob, int32_t n_jets, float* workspace, float* scratch, void* stream) {
    (void)t;  // the time embedding is part of the workspace
    return tf_loss_backward(d, blob, cond, mask, y, u, v, gscale, gblob, nullptr, n_jets, workspace, scratch, stream);
}

int pfm_tf_fm_loss_backward_dx(const pfm_tf_desc* d, const float* blob, const float* cond, const float* mask, const float* y, const float* u,
                               const float* v, const float* gscale, float* gblob, float* grad_y, int32_t n_jets, float* workspace,
                               float* scratch, void* stream) {
    if (!grad_y) return set_err(PFM_E_BADARG, "grad_y is NULL");
    return tf_loss_backward(d, blob, cond, mask, y, u, v, gscale, gblob, grad_y, n_jets, workspace, scratch, stream);
}

int pfm_tf_backward_dtemb(const pfm_tf_desc* d, const float* blob, const float* scratch, int32_t n_jets, float* dtemb, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (!(d->flags & PFM_TF_F_TEMB_GIVEN)) return set_err(PFM_E_BADARG, "pfm_tf_backward_dtemb: the descriptor has no PFM_TF_F_TEMB_GIVEN");
    if (n_jets <= 0) return 0;
    if (!blob || !scratch || !dtemb) return set_err(PFM_E_BADARG, "NULL device pointer");
    const Bs b = make_bs(*d, n_jets);
    hipLaunchKernelGGL(tf::tf_dtemb_kernel, dim3(n_jets), dim3(64), 0, (hipStream_t)stream, blob, scratch + b.dpre, scratch + b.djb, dtemb,
                       d->c1.W, d->time_in_input ? d->n1.Wt : (int64_t)-1, d->t_dim, d->ctxt_hidden, d->hidden,
                       (int64_t)(d->layers + 2) * d->hidden);
    return check_hip(hipGetLastError(), "tf_dtemb_kernel launch");
}

}  // extern "C"
