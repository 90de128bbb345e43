// gfx950 kernels + C ABI (include/pfm_epicw.h): EPiC vector field at widths beyond the jet-resident kernel.
//
// Reference graph: particle_fm/models/components/epic.py:304-391 (EPiC_encoder.forward), :85-203 (EPiC_layer.forward).
// The Linears run on tf_linear_kernel (fp32 MFMA, tf_fwd.h); this file adds the per-jet pieces: time embedding +
// conditioning rows, masked mean / sum pooling, the F-output head, and the launch sequence.
#include <hip/hip_runtime.h>

#include "pfm_epicw.h"
#include "tf_fwd.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

namespace ew {
using namespace pfm::tf;

// P[jet][0 .. 256 + Hp) = [temb | cond | 0 ; 0 (g) ; 0 (g1)]
__global__ __launch_bounds__(256) void ew_prep_kernel(const float* __restrict__ blob, int64_t freqs, const float* __restrict__ t,
                                                      int t_stride, const float* __restrict__ cond, float* __restrict__ P, int T,
                                                      int C, int ldp) {
    const int jet = blockIdx.x;
    float* row = P + (int64_t)jet * ldp;
    for (int c = threadIdx.x; c < ldp; c += 256) {
        float v = 0.f;
        if (c < T) {
            // time_emb.py:90-96, exact fp32 op order ((t + min) * f) * pi / (max + min)
            const float tj = t[(int64_t)jet * t_stride];
            v = cosf(__fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(tj, 0.0f), blob[freqs + c]), 3.14159274101257324f), 1.0f));
        } else if (c < T + C) {
            v = cond[(int64_t)jet * C + c - T];
        }
        row[c] = v;
    }
}

// Q[jet] = [ sum_p mask x / sum_p mask  |  sum_p mask x * scale ]   (epic.py:108-117 / :331-339)
__global__ __launch_bounds__(256) void ew_pool_kernel(const float* __restrict__ X, const float* __restrict__ mask,
                                                      float* __restrict__ Q, int N, int Hp, float scale) {
    __shared__ float red[4 * 512];
    __shared__ float cnt[4];
    const int tid = threadIdx.x, cg = tid & 63, rg = tid >> 6, jet = blockIdx.x;
    const int nc4 = Hp >> 2;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float n = 0.f;
    for (int r = rg; r < N; r += 4) {
        const int64_t row = (int64_t)jet * N + r;
        const float w = mask ? mask[row] : 1.0f;
        n += w;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (cg + 64 * i < nc4) acc[i] += w * *reinterpret_cast<const f32x4*>(X + row * Hp + 4 * (cg + 64 * i));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (cg + 64 * i < nc4) *reinterpret_cast<f32x4*>(red + rg * 512 + 4 * (cg + 64 * i)) = acc[i];
    if (cg == 0) cnt[rg] = n;
    __syncthreads();
    const float nv = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
    for (int c = tid; c < Hp; c += 256) {
        const float s = (red[c] + red[512 + c]) + (red[1024 + c] + red[1536 + c]);
        Q[(int64_t)jet * 2 * Hp + c] = s / nv;
        Q[(int64_t)jet * 2 * Hp + Hp + c] = s * scale;
    }
}

// v[row][f] = lrelu( W3[f] . X[row] + jb3[jet][f] ) * mask[row]   (epic.py:386-389); optional fused state update
struct HeadArgs {
    const float *X, *blob, *jb, *mask, *base, *dt;
    float* dst;
    int64_t W3, jb_stride;
    int M, N, F;
    float slope, coef;
};

template <int NI>
__global__ __launch_bounds__(256) void ew_head_kernel(HeadArgs a) {
    constexpr int Hp = 64 * NI;
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    const int rowc = min(row, a.M - 1);
    const float* xp = a.X + (int64_t)rowc * Hp + 4 * pl;
    f32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(xp + 64 * i);
    const float m = a.mask ? a.mask[rowc] : 1.0f;
    const int jet = rowc / a.N;
#pragma unroll 1
    for (int f = 0; f < a.F; ++f) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            d += hsum4(v[i] * *reinterpret_cast<const f32x4*>(a.blob + a.W3 + (int64_t)f * Hp + 4 * pl + 64 * i));
        d = row_sum16(d) + a.jb[(int64_t)jet * a.jb_stride + f];
        d = lrelu(d, a.slope) * m;
        if (pl == (f & 15) && row < a.M) {
            const int64_t e = (int64_t)row * a.F + f;
            if (a.base) a.dst[e] = __fadd_rn(a.base[e], __fmul_rn(__fmul_rn(a.coef, a.dt[0]), d));
            else a.dst[e] = d;
        }
    }
}

struct Ws {
    int64_t P, Q, SJB, JB, X1, X, L1, lstride, total;  // lstride: per-layer stride of (X, L1) in the train layout
};

__host__ inline Ws make_ws(const pfm_ew_desc& d, int n_jets, bool train) {
    Ws w;
    const int64_t M = (int64_t)n_jets * d.n_points, Hp = d.hidden_pad;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    w.P = take((int64_t)n_jets * (256 + Hp));
    w.Q = take((int64_t)n_jets * 2 * Hp);
    w.SJB = take((int64_t)n_jets * (2 * Hp + 128));
    w.JB = take((int64_t)n_jets * 2 * Hp);
    w.X1 = take(M * Hp);
    w.X = take(M * Hp);
    w.L1 = take(M * Hp);
    w.lstride = 0;
    (void)train;
    w.total = o;
    return w;
}

int validate(const pfm_ew_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_EW_ABI_VERSION) return set_err(PFM_E_BADARG, "epicw desc.abi_version mismatch");
    if (d->hidden < 1 || d->hidden_pad != (d->hidden + 127) / 128 * 128 || d->hidden_pad > 512)
        return set_err(PFM_E_BADARG, "hidden_pad must be hidden rounded up to a multiple of 128, at most 512");
    if (d->layers < 0 || d->layers > PFM_EW_MAX_LAYERS) return set_err(PFM_E_BADARG, "layers out of range");
    if (d->latent < 1 || d->latent > 128) return set_err(PFM_E_BADARG, "latent must be in 1..128");
    if (d->features < 1 || d->features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    if (d->t_dim < 0 || d->cond_global < 0 || d->t_dim + d->cond_global > 128)
        return set_err(PFM_E_BADARG, "t_dim + cond_global must be <= 128");
    if (d->cond_local != 0 && d->cond_local != d->cond_global) return set_err(PFM_E_BADARG, "cond_local must be 0 or cond_global");
    if (d->n_points < 1) return set_err(PFM_E_BADARG, "n_points must be >= 1");
    return 0;
}

struct Plan {
    const pfm_ew_desc* d;
    const float* blob;
    float* ws;
    Ws w;
    int n_jets, M;
    hipStream_t s;
};

// out[Mrows][ldo] = epi(A (+A2) W^T + b / jb)
int linear(const Plan& p, int Mrows, const float* A, int lda, int K1, const float* A2, int lda2, int K, const pfm_ew_lin& lin,
           int NO, const float* jb, int64_t jb_stride, int jbN, const float* R, int ldr, float* out, int ldo, int act) {
    LinArgs a;
    a.A = A; a.A2 = A2; a.lda = lda; a.lda2 = lda2; a.K1 = K1; a.blob = p.blob; a.jb = jb; a.R = R; a.out = out;
    a.blob_floats = p.d->blob_floats; a.W = lin.W; a.b = lin.b; a.gamma = -1; a.beta = -1; a.jb_stride = jb_stride;
    a.ldr = ldr; a.ldo = ldo; a.M = Mrows; a.K = K; a.NO = NO; a.N = jbN; a.act = act;
    a.row_tiles = (Mrows + BM - 1) / BM;
    a.slope = p.d->neg_slope; a.eps = 0.f;
    const int grid = ((a.row_tiles + 7) / 8) * 8 * (NO / BN);
    hipLaunchKernelGGL(tf_linear_kernel<0>, dim3(grid), dim3(LT), (BM * 128 + 2 * BM) * sizeof(float), p.s, a);
    return check_hip(hipGetLastError(), "tf_linear_kernel launch (epicw)");
}

#define PFM_TRY(x) do { if ((rc = (x))) return rc; } while (0)

int run_nfe(const Plan& p, const float* t, int t_stride, const float* x, const float* cond, const float* mask, const HeadArgs& head_tpl) {
    const pfm_ew_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    const int Hp = d.hidden_pad, ldp = 256 + Hp, B = p.n_jets, N = d.n_points;
    const int64_t sjbs = 2 * Hp + 128;
    float *P = ws + w.P, *Q = ws + w.Q, *SJB = ws + w.SJB, *JB = ws + w.JB, *X1 = ws + w.X1, *X = ws + w.X, *L1 = ws + w.L1;
    int rc;
    hipLaunchKernelGGL(ew_prep_kernel, dim3(B), dim3(256), 0, p.s, p.blob, d.freqs, t, t_stride, cond, P, d.t_dim, d.cond_global, ldp);
    PFM_TRY(check_hip(hipGetLastError(), "ew_prep_kernel launch"));
    PFM_TRY(linear(p, B, P, ldp, 256, nullptr, 0, 256, d.sjb, 2 * Hp + 128, nullptr, 0, 1, nullptr, 0, SJB, (int)sjbs, 0));
    // stem: fc_l1 (F columns on the VALU), fc_l2 (residual inside the activation, epic.py:327-328)
    hipLaunchKernelGGL(tf_embed_kernel, dim3((p.M + 31) / 32), dim3(256), 0, p.s, p.blob, d.l1x, x, (const float*)SJB, sjbs, X1, p.M, N,
                       d.features, Hp, d.neg_slope);
    PFM_TRY(check_hip(hipGetLastError(), "tf_embed_kernel launch (epicw)"));
    PFM_TRY(linear(p, p.M, X1, Hp, Hp, nullptr, 0, Hp, d.l2, Hp, SJB + Hp, sjbs, N, X1, Hp, X, Hp, 2));
    auto pool = [&]() {
        hipLaunchKernelGGL(ew_pool_kernel, dim3(B), dim3(256), 0, p.s, (const float*)X, mask, Q, N, Hp, d.sum_scale);
        return check_hip(hipGetLastError(), "ew_pool_kernel launch");
    };
    PFM_TRY(pool());
    PFM_TRY(linear(p, B, P, ldp, 256, Q, 2 * Hp, 256 + 2 * Hp, d.sg1, Hp, nullptr, 0, 1, nullptr, 0, P + 256, ldp, 1));
    PFM_TRY(linear(p, B, P, ldp, ldp, nullptr, 0, ldp, d.sg2, 128, nullptr, 0, 1, nullptr, 0, P + 128, ldp, 1));
    for (int l = 0; l < d.layers; ++l) {
        const pfm_ew_layer& L = d.layer[l];
        if (l) PFM_TRY(pool());
        PFM_TRY(linear(p, B, P, ldp, 256, Q, 2 * Hp, 256 + 2 * Hp, L.g1, Hp, nullptr, 0, 1, nullptr, 0, P + 256, ldp, 1));
        PFM_TRY(linear(p, B, P, ldp, ldp, nullptr, 0, ldp, L.g2, 128, nullptr, 0, 1, P + 128, ldp, P + 128, ldp, 2));
        PFM_TRY(linear(p, B, P, ldp, 256, nullptr, 0, 256, L.jb, 2 * Hp, nullptr, 0, 1, nullptr, 0, JB, 2 * Hp, 0));
        PFM_TRY(linear(p, p.M, X, Hp, Hp, nullptr, 0, Hp, L.l1, Hp, JB, 2 * Hp, N, nullptr, 0, L1, Hp, 1));
        PFM_TRY(linear(p, p.M, L1, Hp, Hp, nullptr, 0, Hp, L.l2, Hp, JB + Hp, 2 * Hp, N, X, Hp, X, Hp, 2));
    }
    HeadArgs h = head_tpl;
    h.X = X; h.blob = p.blob; h.jb = SJB + 2 * Hp; h.jb_stride = sjbs; h.mask = mask; h.W3 = d.l3;
    h.M = p.M; h.N = N; h.F = d.features; h.slope = d.neg_slope;
    const dim3 g((p.M + 15) / 16), bl(256);
    switch (Hp / 64) {
        case 2: hipLaunchKernelGGL(ew_head_kernel<2>, g, bl, 0, p.s, h); break;
        case 4: hipLaunchKernelGGL(ew_head_kernel<4>, g, bl, 0, p.s, h); break;
        case 6: hipLaunchKernelGGL(ew_head_kernel<6>, g, bl, 0, p.s, h); break;
        default: hipLaunchKernelGGL(ew_head_kernel<8>, g, bl, 0, p.s, h); break;
    }
    return check_hip(hipGetLastError(), "ew_head_kernel launch");
}

int make_plan(Plan& p, const pfm_ew_desc* d, const float* blob, float* ws, int n_jets, bool train, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    p.d = d; p.blob = blob; p.ws = ws; p.n_jets = n_jets; p.M = n_jets * d->n_points; p.s = (hipStream_t)stream;
    p.w = make_ws(*d, n_jets, train);
    return 0;
}

}  // namespace ew
}  // namespace pfm

using namespace pfm;
using namespace pfm::ew;

extern "C" {

int64_t pfm_ew_workspace_floats(const pfm_ew_desc* d, int32_t n_jets, int32_t train) {
    if (ew::validate(d)) return -1;
    return ew::make_ws(*d, n_jets < 1 ? 1 : n_jets, train != 0).total;
}

int pfm_ew_forward(const pfm_ew_desc* d, const float* blob, const float* t, int32_t t_stride, const float* x,
                   const float* cond, const float* mask, float* v, int32_t n_jets, float* workspace, void* stream) {
    ew::Plan p;
    int rc = ew::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t || !x || !v || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    ew::HeadArgs h{};
    h.dst = v;
    return ew::run_nfe(p, t, t_stride ? 1 : 0, x, cond, mask, h);
}

int pfm_ew_sample_midpoint(const pfm_ew_desc* d, const float* blob, const float* t_eval, const float* dt,
                           int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out,
                           int32_t n_jets, int32_t premask, float* state, float* workspace, void* stream) {
    ew::Plan p;
    int rc = ew::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    const int64_t n = (int64_t)p.M * d->features;
    float* xs = state;
    float* xm = state + n;
    hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : nullptr, xs, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    for (int k = 0; k < n_steps; ++k) {
        ew::HeadArgs h{};
        h.base = xs; h.dt = dt + k; h.coef = 0.5f; h.dst = xm;
        if ((rc = ew::run_nfe(p, t_eval + 2 * k, 0, xs, cond, mask, h))) return rc;
        h.coef = 1.0f; h.dst = xs;
        if ((rc = ew::run_nfe(p, t_eval + 2 * k + 1, 0, xm, cond, mask, h))) return rc;
    }
    return check_hip(hipMemcpyAsync(x_out, xs, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

}  // extern "C"
