// gfx950 kernels + C ABI (include/pfm_mdma.h): the MDMA vector field (model "mdma"), fixed-step samplers, FM / CFM / droid
// loss forward and backward.
//
// Reference graph: particle_fm/models/components/mdma.py:142-176 (MDMA.forward), :53-84 (Block.forward).
// The particle stream is a row matrix [n_jets * N][H] like the transformer paths: its three Linears per block (fc0 on
// LeakyReLU(x), the key | value half of attn.in_proj, the particle columns of fc1 + residual) are tf_linear_kernel launches
// (MFMA, tf_fwd.h), their weight gradients tf_dw_kernel (tf_bwd.h).  The class token is ONE row per jet: everything that
// touches only the token (fc0_cls, LayerNorm, the query projection, out_proj, fc1_cls, fc2_cls, the token columns of fc1)
// runs in two per-jet kernels per block (before / after the attention), GEMVs over KMAJOR weight blocks, and the
// attention itself is the tokens <- particles kernel of the cross-attention path with one token (ca_attn.h).
// Backward: the same steps in reverse; the per-jet kernels leave their gradient VECTORS per jet and the parameter
// gradients of the token path are fixed-order sums of outer products over the jets (tf_outer_jobs_kernel).
#include <hip/hip_runtime.h>

#include "pfm_mdma.h"
#include "tf_fwd.h"
#include "tf_bwd.h"
#include "ca_attn.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

namespace mdma {
using namespace pfm::tf;

constexpr int JT = 256;      // threads of a per-jet workgroup
constexpr int MAXH = 512;    // widest hidden
constexpr int MAXL = 64;     // widest latent

int validate(const pfm_mdma_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_MDMA_ABI_VERSION) return set_err(PFM_E_BADARG, "mdma desc.abi_version mismatch");
    if (d->hidden < 128 || d->hidden > MAXH || d->hidden % 128) return set_err(PFM_E_BADARG, "hidden must be a multiple of 128 in 128..512");
    if ((d->head_dim != 8 && d->head_dim != 16) || d->heads * d->head_dim != d->hidden || d->heads > 64)
        return set_err(PFM_E_BADARG, "head_dim must be 8 or 16 with heads * head_dim = hidden");
    if (d->latent < 4 || d->latent > MAXL || d->latent % 4) return set_err(PFM_E_BADARG, "latent must be a multiple of 4 in 4..64");
    if (d->layers < 1 || d->layers > PFM_MDMA_MAX_LAYERS) return set_err(PFM_E_BADARG, "layers out of range");
    if (d->features < 1 || d->features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    if (d->t_dim < 0 || d->t_dim > 64) return set_err(PFM_E_BADARG, "t_dim out of range");
    if (d->time_in_input && d->t_dim < 1) return set_err(PFM_E_BADARG, "time_in_input needs t_dim >= 1");
    if (d->t_cat < 0 || d->t_cat > 3 || (d->t_cat && d->t_dim < 1)) return set_err(PFM_E_BADARG, "t_cat must be 0..3 (and needs t_dim >= 1)");
    if (d->c_cat < 0 || d->c_cat > 7 || ((d->c_cat & 2) && !(d->c_cat & 1)))
        return set_err(PFM_E_BADARG, "c_cat must be 0..7, global_cat_cond (bit 1) only with global_cond_dim = 1 (bit 0)");
    if (d->n_points < 1) return set_err(PFM_E_BADARG, "n_points must be >= 1");
    if (!(d->avg_n > 0.f)) return set_err(PFM_E_BADARG, "avg_n must be positive");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-jet helpers: one 256-thread workgroup, vectors in LDS
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {  // red: 4 floats; result on every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y[o] = b[o] + sum_k W[k][o] x[k]  (W KMAJOR [K][O]; b < 0: no bias; x, y in LDS; part: JT floats of LDS).  The K rows are
// split over the JT / O thread groups and the partial sums are added in group order: the same bits every run.
__device__ __forceinline__ void jet_gemv(const float* __restrict__ blob, int64_t W, int64_t b, int K, int O, const float* x,
                                         float* y, float* part) {
    const int tid = threadIdx.x;
    if (O >= JT) {
        for (int o = tid; o < O; o += JT) {
            float acc = b >= 0 ? blob[b + o] : 0.f;
#pragma unroll 4
            for (int k = 0; k < K; ++k) acc = fmaf(blob[W + (int64_t)k * O + o], x[k], acc);
            y[o] = acc;
        }
        __syncthreads();
        return;
    }
    const int P = JT / O, o = tid % O, p = tid / O;
    float acc = 0.f;
    if (p < P)
#pragma unroll 4
        for (int k = p; k < K; k += P) acc = fmaf(blob[W + (int64_t)k * O + o], x[k], acc);
    part[tid] = acc;
    __syncthreads();
    if (tid < O) {
        float s = b >= 0 ? blob[b + tid] : 0.f;
        for (int pp = 0; pp < P; ++pp) s += part[pp * O + tid];
        y[tid] = s;
    }
    __syncthreads();
}

// out[k] = sum_o W[k][o] g[o]  (the same block read along its rows: one wave per row, lanes over o)
__device__ __forceinline__ void jet_gemv_t(const float* __restrict__ blob, int64_t W, int K, int O, const float* g, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = w; k < K; k += JT / 64) {
        float acc = 0.f;
        for (int o = lane; o < O; o += 64) acc = fmaf(blob[W + (int64_t)k * O + o], g[o], acc);
        acc = wave_sum(acc);
        if (lane == 0) out[k] = acc;
    }
    __syncthreads();
}

__device__ __forceinline__ float lrelu_d(float x, float slope) { return x > 0.f ? 1.f : slope; }

// ------------------------------------------------------------------------------------------------
// forward kernels
// ------------------------------------------------------------------------------------------------
// time embedding (time_emb.py:90-96 / flow_matching_module.py:208-211) and the per-jet bias rows it gives the particle Linears:
// jbt[jet][:] = b + Wt . temb   (MDMA.embed; x = cat(temb, x): the time columns are the same for every particle of a jet)
struct TimeArgs {
    const float *blob, *t, *cond, *mask;  // cond [B] (c_cat), mask [B][N]
    float *temb, *tact, *jbt, *jb0;  // temb / tact [B][64]: the embedding and LeakyReLU of it; jbt [B][H]; jb0 [layers][B][H] (t_local / local_cat_cond)
    float *cj, *hb;                  // cj [B][4] = c, cl, LeakyReLU(c), LeakyReLU(cl) (c_cat); hb [B]: the head's bias per jet (local_cat_cond)
    int64_t freqs, Wt, Wt2, Wc, b, out_b, out_Wc, fc0_Wt[PFM_MDMA_MAX_LAYERS], fc0_Wc[PFM_MDMA_MAX_LAYERS], fc0_b[PFM_MDMA_MAX_LAYERS];
    int c_cat, N;
    int t_stride, T, sincos, H, layers, B;  // layers: 0 without t_local / local_cat_cond
    int temb_k;  // PFM_MDMA_F_TEMB_GIVEN: floats between the elements of an embedding row in `t` (0: `t` holds times)
    float slope;
};
static __global__ __launch_bounds__(128) void mdma_time_kernel(TimeArgs a) {
    __shared__ float te[64], ta[64], red2[2], cs[4];
    const int tid = threadIdx.x, jet = blockIdx.x, T = a.T, H = a.H;
    if (a.c_cat) {  // the jet's condition value c and cl = cond[..., -1:] of the reference: c, or the particle count without global_cond_dim
        float m = 0.f;
        for (int n = tid; n < a.N; n += 128) m += a.mask[(int64_t)jet * a.N + n];
        m = wave_sum(m);
        if ((tid & 63) == 0) red2[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const float c = a.cond[jet], cl = (a.c_cat & 3) ? c : red2[0] + red2[1];
            cs[0] = c; cs[1] = cl; cs[2] = lrelu(c, a.slope); cs[3] = lrelu(cl, a.slope);
            if (a.out_Wc >= 0) a.hb[jet] = fmaf(a.blob[a.out_Wc], cs[2], a.blob[a.out_b]);  // out(act(cat(x, c))) (mdma.py:173-175)
        }
        __syncthreads();
        if (tid < 4) a.cj[(int64_t)jet * 4 + tid] = cs[tid];
    }
    if (tid < T) {
        const float tv = a.temb_k ? 0.f : a.t[(int64_t)jet * a.t_stride];
        const float f = a.blob[a.freqs + tid];
        float e;
        if (a.temb_k) {
            e = a.t[(int64_t)jet * a.t_stride + (int64_t)tid * a.temb_k];
        } else if (a.sincos) {
            const float arg = __fmul_rn(f, tv);
            e = 2 * tid < T ? cosf(arg) : sinf(arg);
        } else {
            e = cosf(__fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(tv, 0.0f), f), 3.14159274101257324f), 1.0f));
        }
        te[tid] = e;
        ta[tid] = lrelu(e, a.slope);
        a.temb[(int64_t)jet * 64 + tid] = e;
        a.tact[(int64_t)jet * 64 + tid] = ta[tid];
    }
    __syncthreads();
    for (int o = tid; o < H; o += 128) {
        float acc = a.blob[a.b + o];
        if (a.Wt >= 0)
#pragma unroll 8
            for (int k = 0; k < T; ++k) acc = fmaf(a.blob[a.Wt + (int64_t)k * H + o], te[k], acc);
        if (a.Wt2 >= 0)  // x = cat(x, t_in) in front of MDMA.embed (mdma.py:155-156): a second block of time columns
#pragma unroll 8
            for (int k = 0; k < T; ++k) acc = fmaf(a.blob[a.Wt2 + (int64_t)k * H + o], te[k], acc);
        if (a.Wc >= 0) acc = fmaf(a.blob[a.Wc + o], cs[0], acc);  // ... and cat(x, c) (mdma.py:157-158)
        a.jbt[(int64_t)jet * H + o] = acc;
    }
    // Block.fc0(act(cat(x, t_in))) (mdma.py:56-57, 65): the time columns give every particle of the jet the same bias row
    for (int l = 0; l < a.layers; ++l)
        for (int o = tid; o < H; o += 128) {
            float acc = a.blob[a.fc0_b[l] + o];
            if (a.fc0_Wt[l] >= 0)
#pragma unroll 8
                for (int k = 0; k < T; ++k) acc = fmaf(a.blob[a.fc0_Wt[l] + (int64_t)k * H + o], ta[k], acc);
            if (a.fc0_Wc[l] >= 0) acc = fmaf(a.blob[a.fc0_Wc[l] + o], cs[3], acc);  // act(cat(x, t_in, cond[..., -1:])) (mdma.py:62-65)
            a.jb0[((int64_t)l * a.B + jet) * H + o] = acc;
        }
}

// x = act(embed(cat(temb, x))); x[~mask] = 0   (mdma.py:150-151)
static __global__ __launch_bounds__(256) void mdma_embed_kernel(const float* __restrict__ blob, int64_t Wx, const float* __restrict__ x,
                                                               const float* __restrict__ jbt, const float* __restrict__ mask,
                                                               float* __restrict__ X0, int M, int N, int F, int H, float slope) {
    const int nc4 = H >> 2;
    const int row0 = blockIdx.x * 32;
    for (int idx = threadIdx.x; idx < 32 * nc4; idx += 256) {
        const int r = idx / nc4, c4 = idx - r * nc4;
        const int row = row0 + r;
        if (row >= M) break;
        f32x4 out = {0.f, 0.f, 0.f, 0.f};
        if (mask[row] != 0.f) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(jbt + (int64_t)(row / N) * H + 4 * c4);
            for (int f = 0; f < F; ++f) {
                const float xv = x[(int64_t)row * F + f];
                const f32x4 wv = *reinterpret_cast<const f32x4*>(blob + Wx + (int64_t)f * H + 4 * c4);
                acc.x = fmaf(wv.x, xv, acc.x); acc.y = fmaf(wv.y, xv, acc.y);
                acc.z = fmaf(wv.z, xv, acc.z); acc.w = fmaf(wv.w, xv, acc.w);
            }
            out = lrelu4(acc, slope);
        }
        *reinterpret_cast<f32x4*>(X0 + (int64_t)row * H + 4 * c4) = out;
    }
}

// the class token: x_cls = glu(cat(embbed_cls(cat(sum_n x / avg_n, n_valid)), cond(n_valid)))   (mdma.py:152-162)
struct ClsInitArgs {
    const float *blob, *X0, *mask, *cj;  // cj: the jet's condition scalars (global_cond_dim: gcd = 1)
    float *pooled, *nv, *ea, *eg, *xc0;
    int64_t ecls_W, ecls_b, cond_W, cond_b;
    int N, H, L, gcd;
    float avg_n;
};
static __global__ __launch_bounds__(JT) void mdma_cls_init_kernel(ClsInitArgs a) {
    __shared__ float xin[MAXH + 4], y[MAXL], part[JT], red[4];
    const int tid = threadIdx.x, jet = blockIdx.x;
    for (int k = tid; k < a.H; k += JT) {
        const float* col = a.X0 + (int64_t)jet * a.N * a.H + k;
        float s0 = 0.f, s1 = 0.f;
        int n = 0;
        for (; n + 2 <= a.N; n += 2) { s0 += col[(int64_t)n * a.H]; s1 += col[(int64_t)(n + 1) * a.H]; }
        if (n < a.N) s0 += col[(int64_t)n * a.H];
        const float p = __fdiv_rn(s0 + s1, a.avg_n);
        xin[k] = p;
        a.pooled[(int64_t)jet * a.H + k] = p;
    }
    float m = 0.f;
    for (int n = tid; n < a.N; n += JT) m += a.mask[(int64_t)jet * a.N + n];
    const float nv = block_sum256(m, red);
    const float cv = a.gcd ? a.cj[(int64_t)jet * 4] : 0.f;
    if (tid == 0) { xin[a.H] = nv; xin[a.H + 1] = cv; a.nv[jet] = nv; }
    __syncthreads();
    jet_gemv(a.blob, a.ecls_W, a.ecls_b, a.H + 1 + a.gcd, a.L, xin, y, part);
    if (tid < a.L) {
        const float av = y[tid];
        float g = fmaf(a.blob[a.cond_W + tid], nv, a.blob[a.cond_b + tid]);
        if (a.gcd) g = fmaf(a.blob[a.cond_W + a.L + tid], cv, g);  // cond(cat(n_valid, c)) (mdma.py:168-170)
        a.ea[(int64_t)jet * a.L + tid] = av;
        a.eg[(int64_t)jet * a.L + tid] = g;
        a.xc0[(int64_t)jet * a.L + tid] = av * (1.0f / (1.0f + __expf(-g)));
    }
}

// before the attention: x_cls = ln(fc0_cls(act(x_cls))); q = in_proj[:H] x_cls + b   (mdma.py:67 and the query half of :68)
struct ClsPreArgs {
    const float *blob, *xc_in, *tact, *cj;  // tact: LeakyReLU(time embedding) [B][64] (t_global: Tg > 0); cj: condition scalars (gcc)
    float *pre, *c, *q;
    int64_t fc0c_W, fc0c_b, ln_g, ln_b, q_W, q_b;
    int H, L, Tg, gcc;
    float slope, eps;
};
static __global__ __launch_bounds__(JT) void mdma_cls_pre_kernel(ClsPreArgs a) {
    __shared__ float al[MAXL + 64 + 4], pre[MAXH], c[MAXH], q[MAXH], part[JT], red[4];
    const int tid = threadIdx.x, jet = blockIdx.x;
    if (tid < a.L) al[tid] = lrelu(a.xc_in[(int64_t)jet * a.L + tid], a.slope);
    else if (tid < a.L + a.Tg) al[tid] = a.tact[(int64_t)jet * 64 + tid - a.L];  // act(cat(x_cls, t_in[:, :1])) (mdma.py:58-59, 67)
    else if (tid == a.L + a.Tg && a.gcc) al[tid] = a.cj[(int64_t)jet * 4 + 3];   // ... cat(.., cond[..., -1:]) (mdma.py:60-61)
    __syncthreads();
    jet_gemv(a.blob, a.fc0c_W, a.fc0c_b, a.L + a.Tg + a.gcc, a.H, al, pre, part);
    float s = 0.f;
    for (int k = tid; k < a.H; k += JT) s += pre[k];
    const float mean = block_sum256(s, red) / (float)a.H;
    float ss = 0.f;
    for (int k = tid; k < a.H; k += JT) {
        const float dl = pre[k] - mean;
        ss = fmaf(dl, dl, ss);
    }
    const float rstd = 1.0f / sqrtf(block_sum256(ss, red) / (float)a.H + a.eps);
    for (int k = tid; k < a.H; k += JT) {
        const float cv = (pre[k] - mean) * rstd * a.blob[a.ln_g + k] + a.blob[a.ln_b + k];
        c[k] = cv;
        a.pre[(int64_t)jet * a.H + k] = pre[k];
        a.c[(int64_t)jet * a.H + k] = cv;
    }
    __syncthreads();
    jet_gemv(a.blob, a.q_W, a.q_b, a.H, a.H, c, q, part);
    for (int k = tid; k < a.H; k += JT) a.q[(int64_t)jet * a.H + k] = q[k];
}

// after the attention: out_proj, fc1_cls(cat(x_cls, n_valid)), fc2_cls, and the token columns of fc1 as the jet's bias row
// (mdma.py:68-81: x = fc1(cat(x, x_cls.expand)) + res  ->  fc1.W[:, :H] x + (fc1.W[:, H:] x_cls + b))
struct ClsPostArgs {
    const float *blob, *att, *nv, *temb, *cj;  // temb: the time embedding [B][64] (t_global: Tg > 0); cj: condition scalars (c_cat)
    float *o, *c2, *xc_out, *jb;
    int64_t o_W, o_b, fc1c_W, fc1c_b, fc2c_W, fc2c_b, W1c, b1, W1k;  // W1k: fc1's condition column [H] (local_cat_cond, else -1)
    int H, L, Tg, gcd, gcc;
};
static __global__ __launch_bounds__(JT) void mdma_cls_post_kernel(ClsPostArgs a) {
    __shared__ float att[MAXH], o[MAXH + 4 + 64], c2[MAXL + 64 + 4], xo[MAXL], jb[MAXH], part[JT];
    const int tid = threadIdx.x, jet = blockIdx.x;
    for (int k = tid; k < a.H; k += JT) att[k] = a.att[(int64_t)jet * a.H + k];
    __syncthreads();
    jet_gemv(a.blob, a.o_W, a.o_b, a.H, a.H, att, o, part);
    if (tid == 0) {
        o[a.H] = a.nv[jet];
        if (a.gcd) o[a.H + 1] = a.cj[(int64_t)jet * 4];             // cond = cat(n_valid, c) (mdma.py:168-169)
        if (a.gcc) c2[a.L + a.Tg] = a.cj[(int64_t)jet * 4 + 1];     // cat(x_cls, t_in[:, :1], cond[..., -1:]) in front of fc2_cls (:78-79)
    }
    if (tid < a.Tg) {  // cat(x_cls, cond, t_in[:, :1]) in front of fc1_cls, cat(x_cls, t_in[:, :1]) in front of fc2_cls (mdma.py:70-78)
        const float e = a.temb[(int64_t)jet * 64 + tid];
        o[a.H + 1 + a.gcd + tid] = e;
        c2[a.L + tid] = e;
    }
    for (int k = tid; k < a.H; k += JT) a.o[(int64_t)jet * a.H + k] = o[k];
    __syncthreads();
    jet_gemv(a.blob, a.fc1c_W, a.fc1c_b, a.H + 1 + a.gcd + a.Tg, a.L, o, c2, part);
    jet_gemv(a.blob, a.fc2c_W, a.fc2c_b, a.L + a.Tg + a.gcc, a.L, c2, xo, part);
    if (tid < a.L) {
        a.c2[(int64_t)jet * a.L + tid] = c2[tid];
        a.xc_out[(int64_t)jet * a.L + tid] = xo[tid];
    }
    jet_gemv(a.blob, a.W1c, a.b1, a.L, a.H, xo, jb, part);
    const float cl = a.W1k >= 0 ? a.cj[(int64_t)jet * 4 + 1] : 0.f;  // fc1(cat(x, cond[..., -1:], x_cls.expand)) (mdma.py:81-83)
    for (int k = tid; k < a.H; k += JT) a.jb[(int64_t)jet * a.H + k] = a.W1k >= 0 ? fmaf(a.blob[a.W1k + k], cl, jb[k]) : jb[k];
}

// v = out(act(x)) * mask, written F times (the reference's loss / solver broadcast the single output over the features)
template <int NI>
__global__ __launch_bounds__(256) void mdma_head_kernel(const float* __restrict__ X, const float* __restrict__ blob, int64_t W, int64_t b,
                                                        const float* __restrict__ mask, float* __restrict__ dst, int M, int F, float slope,
                                                        const float* __restrict__ hb, int N) {  // hb: the bias per jet (local_cat_cond) or nullptr
    constexpr int H = 64 * NI;
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    const int rowc = min(row, M - 1);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 xv = lrelu4(*reinterpret_cast<const f32x4*>(X + (int64_t)rowc * H + 4 * pl + 64 * i), slope);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(blob + W + 4 * pl + 64 * i);
        d += hsum4(xv * w4);
    }
    d = (row_sum16(d) + (hb ? hb[rowc / N] : blob[b])) * mask[rowc];
    if (row < M && pl < F) dst[(int64_t)row * F + pl] = d;
}

// ------------------------------------------------------------------------------------------------
// backward kernels
// ------------------------------------------------------------------------------------------------
// d loss / d v per row (sum over the F broadcast copies, times the mask), the head's gradients, and d loss / d x_L
template <int NI>
__global__ __launch_bounds__(256) void mdma_head_bwd_kernel(const float* __restrict__ X, const float* __restrict__ v, const float* __restrict__ u,
                                                            const float* __restrict__ mask, const float* __restrict__ gscale,
                                                            const float* __restrict__ blob, int64_t W, float* __restrict__ gblob, int64_t gb,
                                                            float* __restrict__ dvrow, float* __restrict__ zact, float* __restrict__ gX,
                                                            int M, int F, float slope, float* __restrict__ part) {  // part[gridDim.x]: the
    // workgroups' partial sums of the head's bias gradient (added in block order by launch_ordered_sum: no atomics)
    constexpr int H = 64 * NI;
    __shared__ float red[16];
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    const int rowc = min(row, M - 1);
    float s = 0.f;
    if (pl < F) s = 2.0f * (v[(int64_t)rowc * F + pl] - u[(int64_t)rowc * F + pl]);
    const float dv = row < M ? gscale[0] * mask[rowc] * row_sum16(s) : 0.f;
    if (row < M) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t e = (int64_t)row * H + 4 * pl + 64 * i;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(X + e);
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(blob + W + 4 * pl + 64 * i);
            *reinterpret_cast<f32x4*>(zact + e) = lrelu4(xv, slope);
            f32x4 g = w4 * dv;
            g.x *= lrelu_d(xv.x, slope); g.y *= lrelu_d(xv.y, slope); g.z *= lrelu_d(xv.z, slope); g.w *= lrelu_d(xv.w, slope);
            *reinterpret_cast<f32x4*>(gX + e) = g;
        }
        if (pl == 0) dvrow[row] = dv;
    }
    if (pl == 0) red[tid >> 4] = dv;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        if (part) part[blockIdx.x] = t;
        else atomicAdd(gblob + gb, t);
    }
}

// the token path after the attention, backwards.  In: djb = sum over the jet's rows of d loss / d x_out (the bias row of fc1),
// gxc_next = d loss / d x_cls_out from the block above (nullptr: none).  Out, per jet: gxo = d / d x_cls_out (total),
// dc2 = d / d fc1_cls output, do = d / d out_proj output, datt = d / d attention output.
struct ClsPostBwdArgs {
    const float *blob, *djb, *gxc_next;
    float *gxo, *dc2, *dout, *datt;
    float* dtemb;  // PFM_MDMA_F_TEMB_GIVEN with t_global: [B][64], += the time rows of fc2_cls / fc1_cls times their output gradients
    int64_t W1c, fc2c_W, fc1c_W, o_W;
    int H, L, Tg, gcd;
};
static __global__ __launch_bounds__(JT) void mdma_cls_post_bwd_kernel(ClsPostBwdArgs a) {
    __shared__ float djb[MAXH], gxo[MAXL], dc2[MAXL], dov[MAXH + 4], datt[MAXH], dte[2][64];
    const int tid = threadIdx.x, jet = blockIdx.x;
    for (int k = tid; k < a.H; k += JT) djb[k] = a.djb[(int64_t)jet * a.H + k];
    __syncthreads();
    jet_gemv_t(a.blob, a.W1c, a.L, a.H, djb, gxo);
    if (tid < a.L) {
        if (a.gxc_next) gxo[tid] += a.gxc_next[(int64_t)jet * a.L + tid];
        a.gxo[(int64_t)jet * a.L + tid] = gxo[tid];
    }
    __syncthreads();
    jet_gemv_t(a.blob, a.fc2c_W, a.L, a.L, gxo, dc2);
    if (tid < a.L) a.dc2[(int64_t)jet * a.L + tid] = dc2[tid];
    if (a.dtemb && a.Tg) {
        jet_gemv_t(a.blob, a.fc2c_W + (int64_t)a.L * a.L, a.Tg, a.L, gxo, dte[0]);
        jet_gemv_t(a.blob, a.fc1c_W + (int64_t)(a.H + 1 + a.gcd) * a.L, a.Tg, a.L, dc2, dte[1]);
        if (tid < a.Tg) a.dtemb[(int64_t)jet * 64 + tid] += dte[0][tid] + dte[1][tid];
    }
    jet_gemv_t(a.blob, a.fc1c_W, a.H + 1, a.L, dc2, dov);  // row H (the particle count) has no upstream
    for (int k = tid; k < a.H; k += JT) a.dout[(int64_t)jet * a.H + k] = dov[k];
    jet_gemv_t(a.blob, a.o_W, a.H, a.H, dov, datt);
    for (int k = tid; k < a.H; k += JT) a.datt[(int64_t)jet * a.H + k] = datt[k];
}

// the token path before the attention, backwards.  In: gq = d loss / d q.  Out, per jet: dc = d / d ln output (= the ln.bias
// term), dgx = dc * xhat (the ln.weight term), dpre = d / d fc0_cls output, al = act(x_cls_in), gxc_in = d / d x_cls_in.
struct ClsPreBwdArgs {
    const float *blob, *gq, *pre, *xc_in, *temb;
    float *dc, *dgx, *dpre, *al, *gxc_in;
    float* dtemb;  // PFM_MDMA_F_TEMB_GIVEN with t_global: [B][64], += LeakyReLU'(temb) * (the time rows of fc0_cls times dpre)
    int64_t q_W, ln_g, fc0c_W;
    int H, L, Tg;
    float slope, eps;
};
static __global__ __launch_bounds__(JT) void mdma_cls_pre_bwd_kernel(ClsPreBwdArgs a) {
    __shared__ float gq[MAXH], dc[MAXH], dpre[MAXH], da[MAXL], red[4], dte[64];
    const int tid = threadIdx.x, jet = blockIdx.x;
    for (int k = tid; k < a.H; k += JT) gq[k] = a.gq[(int64_t)jet * a.H + k];
    __syncthreads();
    jet_gemv_t(a.blob, a.q_W, a.H, a.H, gq, dc);
    // LayerNorm backward over the H values of the jet's token
    float s = 0.f;
    for (int k = tid; k < a.H; k += JT) s += a.pre[(int64_t)jet * a.H + k];
    const float mean = block_sum256(s, red) / (float)a.H;
    float ss = 0.f;
    for (int k = tid; k < a.H; k += JT) {
        const float dl = a.pre[(int64_t)jet * a.H + k] - mean;
        ss = fmaf(dl, dl, ss);
    }
    const float rstd = 1.0f / sqrtf(block_sum256(ss, red) / (float)a.H + a.eps);
    float s1 = 0.f, s2 = 0.f;
    for (int k = tid; k < a.H; k += JT) {
        const float xh = (a.pre[(int64_t)jet * a.H + k] - mean) * rstd;
        const float dxh = dc[k] * a.blob[a.ln_g + k];
        s1 += dxh;
        s2 = fmaf(dxh, xh, s2);
    }
    const float m1 = block_sum256(s1, red) / (float)a.H;
    const float m2 = block_sum256(s2, red) / (float)a.H;
    for (int k = tid; k < a.H; k += JT) {
        const float xh = (a.pre[(int64_t)jet * a.H + k] - mean) * rstd;
        const float dxh = dc[k] * a.blob[a.ln_g + k];
        const float dp = rstd * (dxh - m1 - xh * m2);
        dpre[k] = dp;
        a.dpre[(int64_t)jet * a.H + k] = dp;
        a.dc[(int64_t)jet * a.H + k] = dc[k];
        a.dgx[(int64_t)jet * a.H + k] = dc[k] * xh;
    }
    __syncthreads();
    if (a.dtemb && a.Tg) {
        jet_gemv_t(a.blob, a.fc0c_W + (int64_t)a.L * a.H, a.Tg, a.H, dpre, dte);
        if (tid < a.Tg) a.dtemb[(int64_t)jet * 64 + tid] += dte[tid] * lrelu_d(a.temb[(int64_t)jet * 64 + tid], a.slope);
    }
    jet_gemv_t(a.blob, a.fc0c_W, a.L, a.H, dpre, da);
    if (tid < a.L) {
        const float xv = a.xc_in[(int64_t)jet * a.L + tid];
        a.al[(int64_t)jet * a.L + tid] = lrelu(xv, a.slope);
        a.gxc_in[(int64_t)jet * a.L + tid] = da[tid] * lrelu_d(xv, a.slope);
    }
}

// the gated token of mdma.py:156-162, backwards: x_cls = ea * sigmoid(eg)
struct ClsInitBwdArgs {
    const float *blob, *gxc0, *ea, *eg;
    float *da, *dg, *dpool;
    int64_t ecls_W;
    int H, L;
    float avg_n;
};
static __global__ __launch_bounds__(JT) void mdma_cls_init_bwd_kernel(ClsInitBwdArgs a) {
    __shared__ float da[MAXL], dp[MAXH + 4];
    const int tid = threadIdx.x, jet = blockIdx.x;
    if (tid < a.L) {
        const int64_t e = (int64_t)jet * a.L + tid;
        const float g = a.gxc0[e], sg = 1.0f / (1.0f + __expf(-a.eg[e]));
        const float dav = g * sg;
        da[tid] = dav;
        a.da[e] = dav;
        a.dg[e] = g * a.ea[e] * sg * (1.0f - sg);
    }
    __syncthreads();
    jet_gemv_t(a.blob, a.ecls_W, a.H, a.L, da, dp);
    for (int k = tid; k < a.H; k += JT) a.dpool[(int64_t)jet * a.H + k] = __fdiv_rn(dp[k], a.avg_n);
}

// out[jet] = sum of the jet's N values of v (local_cat_cond: the head's per-row output gradients, for its condition weight)
static __global__ __launch_bounds__(64) void mdma_jetsum_kernel(const float* __restrict__ v, float* __restrict__ out, int N) {
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) s += v[(int64_t)blockIdx.x * N + n];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// PFM_MDMA_F_TEMB_GIVEN: dtemb[jet][k] += f'(temb[jet][k]) * sum_o (W1[k][o] + W2[k][o]) g[jet][o] -- the time columns (KMAJOR [T][H]) of a
// particle Linear whose per-jet bias row they feed (embed: W1, W2 = its two time blocks, f = id; Block.fc0: f = LeakyReLU)
static __global__ __launch_bounds__(JT) void mdma_dtemb_kernel(const float* __restrict__ blob, int64_t W1, int64_t W2, const float* __restrict__ g,
                                                               const float* __restrict__ temb, float* __restrict__ dtemb, int T, int H,
                                                               int act, float slope) {
    __shared__ float gv[MAXH], o1[64], o2[64];
    const int tid = threadIdx.x, jet = blockIdx.x;
    for (int k = tid; k < H; k += JT) gv[k] = g[(int64_t)jet * H + k];
    __syncthreads();
    if (W1 >= 0) jet_gemv_t(blob, W1, T, H, gv, o1);
    if (W2 >= 0) jet_gemv_t(blob, W2, T, H, gv, o2);
    if (tid < T) {
        const float s = (W1 >= 0 ? o1[tid] : 0.f) + (W2 >= 0 ? o2[tid] : 0.f);
        dtemb[(int64_t)jet * 64 + tid] += act ? s * lrelu_d(temb[(int64_t)jet * 64 + tid], slope) : s;
    }
}

// d loss / d (embed output before the activation): the particle stream's gradient plus the pooled sum's, through the mask
// and the LeakyReLU (x0 = 0 at padded particles, where the gradient is 0 anyway)
static __global__ __launch_bounds__(256) void mdma_embed_bwd_kernel(const float* __restrict__ gX, const float* __restrict__ dpool,
                                                                   const float* __restrict__ X0, const float* __restrict__ mask,
                                                                   float* __restrict__ gE, int64_t n4, int N, int H, float slope) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int nc4 = H >> 2;
    const int64_t row = i / nc4;
    const int c4 = (int)(i - row * nc4);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (mask[row] != 0.f) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(X0 + row * H + 4 * c4);
        g = *reinterpret_cast<const f32x4*>(gX + row * H + 4 * c4) + *reinterpret_cast<const f32x4*>(dpool + (row / N) * H + 4 * c4);
        g.x *= lrelu_d(x0.x, slope); g.y *= lrelu_d(x0.y, slope); g.z *= lrelu_d(x0.z, slope); g.w *= lrelu_d(x0.w, slope);
    }
    *reinterpret_cast<f32x4*>(gE + row * H + 4 * c4) = g;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Ws {
    int64_t temb, tact, jbt, jb0, cj, hb, nv, pooled, ea, eg, xc, X, xstride, layer0, lstride;  // jb0: [layers][B][H] fc0 jet-bias rows (t_local / local_cat_cond)
    int64_t o_h, o_kv, o_pre, o_c, o_q, o_att, o_o, o_c2, o_jb, total;
};

Ws make_ws(const pfm_mdma_desc& d, int n_jets, bool train) {
    Ws w;
    const int64_t B = n_jets, M = B * d.n_points, H = d.hidden, L = d.latent;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    w.temb = take(B * 64); w.tact = take(B * 64); w.jbt = take(B * H); w.nv = take(B); w.pooled = take(B * H);
    w.jb0 = take(((d.t_cat & 1) || (d.c_cat & 4)) ? (int64_t)d.layers * B * H : 0);
    w.cj = take(d.c_cat ? B * 4 : 0); w.hb = take((d.c_cat & 4) ? B : 0);
    w.ea = take(B * L); w.eg = take(B * L);
    w.xc = take((int64_t)(d.layers + 1) * round64(B * L));
    w.xstride = train ? round64(M * H) : 0;
    w.X = take(train ? (int64_t)(d.layers + 1) * w.xstride : M * H);
    w.layer0 = o;
    int64_t p = 0;
    auto sub = [&](int64_t n) { const int64_t at = p; p += round64(n); return at; };
    w.o_h = sub(M * H); w.o_kv = sub(M * 2 * H);
    w.o_pre = sub(B * H); w.o_c = sub(B * H); w.o_q = sub(B * H); w.o_att = sub(B * H); w.o_o = sub(B * H);
    w.o_c2 = sub(B * L); w.o_jb = sub(B * H);
    w.lstride = train ? p : 0;
    o += train ? p * d.layers : p;
    w.total = o;
    return w;
}

struct Plan {
    const pfm_mdma_desc* d;
    const float* blob;
    float* ws;
    Ws w;
    int n_jets, M;
    hipStream_t s;
    int temb_k = 0;  // PFM_MDMA_F_TEMB_GIVEN: 1 (rows [jet][T]); the sampler switches to its [T][evaluations] table
    const float* cond = nullptr;  // [n_jets] condition values (desc.c_cat)
    float* X(int l) const { return ws + w.X + w.xstride * l; }
    float* xc(int l) const { return ws + w.xc + round64((int64_t)n_jets * d->latent) * l; }
    float* lay(int l) const { return ws + w.layer0 + w.lstride * l; }
};

int make_plan(Plan& p, const pfm_mdma_desc* d, const float* blob, float* ws, int n_jets, bool train, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    p.d = d; p.blob = blob; p.ws = ws; p.n_jets = n_jets; p.M = n_jets * d->n_points; p.s = (hipStream_t)stream;
    p.w = make_ws(*d, n_jets, train);
    p.temb_k = (d->flags & PFM_MDMA_F_TEMB_GIVEN) ? 1 : 0;
    return 0;
}

// out[M][NO] = epi(A(pre-activated?) W^T + b / jet bias) with the residual / derivative modes of LinArgs::act
int linear(const Plan& p, const float* A, int K, int64_t W, int64_t b, int NO, const float* jb, const float* R, const float* Y, float* out,
           int act, bool pre_act) {
    LinArgs a;
    a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.blob = p.blob; a.jb = jb; a.R = R; a.Y = Y; a.ldy = NO;
    a.rowjet = nullptr; a.m_dev = nullptr; a.part = nullptr; a.ksplit = 1; a.out = out;
    a.blob_floats = p.d->blob_floats; a.W = W; a.b = b; a.gamma = -1; a.beta = -1; a.jb_stride = p.d->hidden;
    a.lda = K; a.ldr = NO; a.ldo = NO; a.M = p.M; a.K = K; a.NO = NO; a.N = p.d->n_points; a.act = act;
    a.slope = p.d->neg_slope; a.eps = p.d->ln_eps; a.pre_act = pre_act ? 1 : 0;
    if (launch_linear_kernel(a, 0, (p.d->flags & PFM_MDMA_F_BF16) ? 2 : 0, num_cus(), p.s)) return set_err(PFM_E_BADARG, "tf_linear_kernel: unsupported shape");
    return check_hip(hipGetLastError(), "tf_linear_kernel launch (mdma)");
}

#define PFM_TRY(x) do { if ((rc = (x))) return rc; } while (0)
#define PFM_MDMA_ATTN(KERNEL, ...)                                                                                        \
    do {                                                                                                                  \
        const dim3 ag(p.n_jets, (d.heads + 3) / 4);                                                                       \
        if (d.head_dim == 16) hipLaunchKernelGGL((ca::KERNEL<16, 4>), ag, dim3(256), 0, p.s, __VA_ARGS__);                \
        else hipLaunchKernelGGL((ca::KERNEL<8, 4>), ag, dim3(256), 0, p.s, __VA_ARGS__);                                  \
    } while (0)
#define PFM_MDMA_NI(KERNEL, grid, ...)                                                               \
    do {                                                                                             \
        switch (d.hidden / 64) {                                                                     \
            case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, p.s, __VA_ARGS__); break;      \
            case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, p.s, __VA_ARGS__); break;      \
            case 6: hipLaunchKernelGGL(KERNEL<6>, grid, dim3(256), 0, p.s, __VA_ARGS__); break;      \
            default: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, p.s, __VA_ARGS__); break;     \
        }                                                                                            \
    } while (0)

// one evaluation: v_out[M][F] (broadcast over F)
int run_nfe(const Plan& p, const float* t, int t_stride, const float* x, const float* mask, float* v_out) {
    const pfm_mdma_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    const int H = d.hidden, L = d.latent, N = d.n_points, B = p.n_jets;
    int rc;
    const int Tl = (d.t_cat & 1) ? d.t_dim : 0, Tg = (d.t_cat & 2) ? d.t_dim : 0;
    const int gcd = d.c_cat & 1, gcc = (d.c_cat >> 1) & 1, lcc = (d.c_cat >> 2) & 1;
    if (d.c_cat && !p.cond) return set_err(PFM_E_BADARG, "the conditional MDMA (desc.c_cat) needs cond");
    const float* cj = ws + w.cj;
    {
        TimeArgs a;
        a.blob = p.blob; a.t = t; a.temb = ws + w.temb; a.tact = ws + w.tact; a.jbt = ws + w.jbt; a.jb0 = ws + w.jb0;
        a.cond = p.cond; a.mask = mask; a.cj = ws + w.cj; a.hb = ws + w.hb; a.c_cat = d.c_cat; a.N = N;
        a.Wc = lcc ? d.emb_Wc : (int64_t)-1; a.out_b = d.out_b; a.out_Wc = lcc ? d.out_Wc : (int64_t)-1;
        a.freqs = d.freqs; a.Wt = d.time_in_input ? d.emb_Wt : (int64_t)-1; a.Wt2 = Tl ? d.emb_Wt2 : (int64_t)-1; a.b = d.emb_b;
        a.t_stride = p.temb_k ? (t_stride ? d.t_dim : 0) : t_stride; a.temb_k = p.temb_k;
        a.T = (d.time_in_input || d.t_cat) ? d.t_dim : 0; a.sincos = (d.flags & PFM_MDMA_F_TEMB_SINCOS) ? 1 : 0;
        a.H = H; a.layers = (Tl || lcc) ? d.layers : 0; a.B = B; a.slope = d.neg_slope;
        for (int l = 0; l < a.layers; ++l) {
            a.fc0_Wt[l] = Tl ? d.block[l].fc0.Wt : (int64_t)-1; a.fc0_Wc[l] = lcc ? d.block[l].fc0.Wc : (int64_t)-1; a.fc0_b[l] = d.block[l].fc0.b;
        }
        hipLaunchKernelGGL(mdma_time_kernel, dim3(B), dim3(128), 0, p.s, a);
    }
    PFM_TRY(check_hip(hipGetLastError(), "mdma_time_kernel launch"));
    hipLaunchKernelGGL(mdma_embed_kernel, dim3((p.M + 31) / 32), dim3(256), 0, p.s, p.blob, d.emb_Wx, x, (const float*)(ws + w.jbt), mask,
                       p.X(0), p.M, N, d.features, H, d.neg_slope);
    PFM_TRY(check_hip(hipGetLastError(), "mdma_embed_kernel launch"));
    {
        ClsInitArgs a;
        a.blob = p.blob; a.X0 = p.X(0); a.mask = mask;
        a.pooled = ws + w.pooled; a.nv = ws + w.nv; a.ea = ws + w.ea; a.eg = ws + w.eg; a.xc0 = p.xc(0);
        a.ecls_W = d.ecls_W; a.ecls_b = d.ecls_b; a.cond_W = d.cond_W; a.cond_b = d.cond_b;
        a.N = N; a.H = H; a.L = L; a.avg_n = d.avg_n; a.cj = cj; a.gcd = gcd;
        hipLaunchKernelGGL(mdma_cls_init_kernel, dim3(B), dim3(JT), 0, p.s, a);
        PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_init_kernel launch"));
    }
    for (int l = 0; l < d.layers; ++l) {
        const pfm_mdma_block& k = d.block[l];
        float* lb = p.lay(l);
        const float* Xin = p.X(l);
        float* Xout = p.X(l + 1);
        PFM_TRY(linear(p, Xin, H, k.fc0.W, k.fc0.b, H, (Tl || lcc) ? ws + w.jb0 + (int64_t)l * B * H : nullptr, nullptr, nullptr, lb + w.o_h, 0, true));
        {
            ClsPreArgs a;
            a.blob = p.blob; a.xc_in = p.xc(l); a.tact = ws + w.tact; a.Tg = Tg; a.cj = cj; a.gcc = gcc; a.pre = lb + w.o_pre; a.c = lb + w.o_c; a.q = lb + w.o_q;
            a.fc0c_W = k.fc0c_W; a.fc0c_b = k.fc0c_b; a.ln_g = k.ln_g; a.ln_b = k.ln_b; a.q_W = k.q_W; a.q_b = k.q_b;
            a.H = H; a.L = L; a.slope = d.neg_slope; a.eps = d.ln_eps;
            hipLaunchKernelGGL(mdma_cls_pre_kernel, dim3(B), dim3(JT), 0, p.s, a);
            PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_pre_kernel launch"));
        }
        PFM_TRY(linear(p, lb + w.o_h, H, k.kv.W, k.kv.b, 2 * H, nullptr, nullptr, nullptr, lb + w.o_kv, 0, false));
        PFM_MDMA_ATTN(ca_attn_from_kernel, (const float*)(lb + w.o_q), (const float*)(lb + w.o_kv), mask, lb + w.o_att, N, H, d.heads, 1,
                      (const int*)nullptr, (const int*)nullptr);
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_from_kernel launch (mdma)"));
        {
            ClsPostArgs a;
            a.blob = p.blob; a.att = lb + w.o_att; a.nv = ws + w.nv; a.temb = ws + w.temb; a.Tg = Tg; a.cj = cj; a.gcd = gcd; a.gcc = gcc;
            a.W1k = lcc ? k.fc1.Wt : (int64_t)-1;
            a.o = lb + w.o_o; a.c2 = lb + w.o_c2; a.xc_out = p.xc(l + 1); a.jb = lb + w.o_jb;
            a.o_W = k.o_W; a.o_b = k.o_b; a.fc1c_W = k.fc1c_W; a.fc1c_b = k.fc1c_b; a.fc2c_W = k.fc2c_W; a.fc2c_b = k.fc2c_b;
            a.W1c = k.fc1.Wc; a.b1 = k.fc1.b; a.H = H; a.L = L;
            hipLaunchKernelGGL(mdma_cls_post_kernel, dim3(B), dim3(JT), 0, p.s, a);
            PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_post_kernel launch"));
        }
        PFM_TRY(linear(p, lb + w.o_h, H, k.fc1.W, -1, H, lb + w.o_jb, Xin, nullptr, Xout, 0, false));
    }
    PFM_MDMA_NI(mdma_head_kernel, dim3((p.M + 15) / 16), (const float*)p.X(d.layers), p.blob, d.out_W, d.out_b, mask, v_out, p.M,
                d.features, d.neg_slope, (d.c_cat & 4) ? (const float*)(ws + w.hb) : (const float*)nullptr, N);
    return check_hip(hipGetLastError(), "mdma_head_kernel launch");
}

// ---- backward ------------------------------------------------------------------------------------------
struct Bs {
    int64_t dvrow, zact, gX, gH, gkv, djb, gxc, gxo, dc2, dout, datt, gq, dc, dgx, dpre, al, da, dg, dpool, djbt, dwpart, dtemb, dvj, total;
};

Bs make_bs(const pfm_mdma_desc& d, int n_jets) {
    Bs b;
    const int64_t B = n_jets, M = B * d.n_points, H = d.hidden, L = d.latent;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    b.dvrow = take(M); b.zact = take(M * H); b.gX = take(M * H); b.gH = take(M * H); b.gkv = take(M * 2 * H);
    b.djb = take(B * H); b.gxc = take(2 * round64(B * L)); b.gxo = take(B * L); b.dc2 = take(B * L);
    b.dout = take(B * H); b.datt = take(B * H); b.gq = take(B * H); b.dc = take(B * H); b.dgx = take(B * H); b.dpre = take(B * H);
    b.al = take(B * L); b.da = take(B * L); b.dg = take(B * L); b.dpool = take(B * H); b.djbt = take(B * H);
    b.dwpart = take((int64_t)DW_MAX_PARTS * 16384);
    b.dtemb = take((d.flags & PFM_MDMA_F_TEMB_GIVEN) ? B * 64 : 0);
    b.dvj = take((d.c_cat & 4) ? B : 0);
    b.total = o;
    return b;
}

struct Bwd {
    Plan p;
    float *gblob, *sc;
    Bs b;

    int colsum(const float* Z, int NO, const float* X, int F, float* jet_out, int64_t gb) const {
        ColsumArgs a;
        a.Z = Z; a.X = X; a.jet_out = jet_out; a.gblob = gblob; a.gb = gb; a.jet_stride = NO;
        a.ldz = NO; a.NO = NO; a.N = p.d->n_points; a.F = F; a.rows = p.M;
        const int ny = X ? F : 1;
        a.part = gb >= 0 ? sc + b.dwpart : nullptr;  // (free between two dW launches; every launch on p.s: stream order)
        hipLaunchKernelGGL(tf_colsum_kernel, dim3(p.n_jets, ny, (NO + 767) / 768), dim3(256), 0, p.s, a);
        if (gb >= 0) launch_ordered_sum(p.s, a.part, p.n_jets, (int64_t)ny * NO, ny * NO, gblob + gb, ny * NO, nullptr);
        return check_hip(hipGetLastError(), "tf_colsum_kernel launch (mdma)");
    }
    int dw(const float* Z, int NO, const float* A, int K, int64_t gW, bool pre_act) const {
        DwArgs a;
        a.Z = Z; a.A = A; a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.stats = nullptr; a.blob = p.blob; a.part = sc + b.dwpart;
        a.gamma = -1; a.beta = -1; a.ldz = NO; a.lda = K; a.M = p.M; a.NO = NO; a.K = K;
        a.pre_act = pre_act ? 1 : 0; a.slope = p.d->neg_slope;
        a.row_tiles = (p.M + BM - 1) / BM;
        const int tiles = ((NO + 127) / 128) * ((K + 127) / 128);
        const int ns = dw_splits(a.row_tiles, tiles, num_cus());
        a.nsplit = ns;
        int rc;
        hipLaunchKernelGGL(tf_dw_kernel, dim3(tiles * ns), dim3(LT), 2 * 64 * DWS * sizeof(float), p.s, a);
        if ((rc = check_hip(hipGetLastError(), "tf_dw_kernel launch (mdma)"))) return rc;
        launch_dw_reduce(p.s, a.part, gblob, gW, NO, K, tiles, ns);
        return check_hip(hipGetLastError(), "tf_dw_reduce_kernel launch (mdma)");
    }
};

struct Jobs {
    OuterJobs j;
    int n = 0, most = 0;
    float* gblob;
    void add(const float* U, int64_t ldu, int K, const float* V, int64_t ldv, int NO, int64_t off) {
        j.job[n++] = OuterJob{U, V, gblob + off, ldu, ldv, K, NO};
        if (K * NO > most) most = K * NO;
    }
    int launch(int n_jets, hipStream_t s) {
        j.n_jets = n_jets;
        hipLaunchKernelGGL(tf_outer_jobs_kernel, dim3((most + 255) / 256, n), dim3(256), 0, s, j);
        return check_hip(hipGetLastError(), "tf_outer_jobs_kernel launch (mdma)");
    }
};

int run_backward(const Bwd& Bw, const float* mask, const float* y, const float* u, const float* v, const float* gscale) {
    const Plan& p = Bw.p;
    const pfm_mdma_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    float* sc = Bw.sc;
    const Bs& b = Bw.b;
    const int H = d.hidden, L = d.latent, N = d.n_points, B = p.n_jets, F = d.features;
    float *gX = sc + b.gX, *gH = sc + b.gH, *gkv = sc + b.gkv, *djb = sc + b.djb;
    float* gxc[2] = {sc + b.gxc, sc + b.gxc + round64((int64_t)B * L)};
    const int Tl = (d.t_cat & 1) ? d.t_dim : 0, Tg = (d.t_cat & 2) ? d.t_dim : 0;
    const float *temb = ws + w.temb, *tact = ws + w.tact;
    const int gcd = d.c_cat & 1, gcc = (d.c_cat >> 1) & 1, lcc = (d.c_cat >> 2) & 1;
    const float* cj = ws + w.cj;  // [B][4] = c, cl, LeakyReLU(c), LeakyReLU(cl) of the forward
    float* dtemb = (d.flags & PFM_MDMA_F_TEMB_GIVEN) ? sc + b.dtemb : nullptr;  // (zeroed by the caller of run_backward)
    int rc;
    PFM_MDMA_NI(mdma_head_bwd_kernel, dim3((p.M + 15) / 16), (const float*)p.X(d.layers), v, u, mask, gscale, p.blob, d.out_W, Bw.gblob,
                d.out_b, sc + b.dvrow, sc + b.zact, gX, p.M, F, d.neg_slope, sc + b.dwpart);
    launch_ordered_sum(p.s, sc + b.dwpart, (p.M + 15) / 16, 1, 1, Bw.gblob + d.out_b, 1, nullptr);
    PFM_TRY(check_hip(hipGetLastError(), "mdma_head_bwd_kernel launch"));
    PFM_TRY(Bw.colsum(sc + b.zact, H, sc + b.dvrow, 1, nullptr, d.out_W));
    if (lcc) {  // out(act(cat(x, c))): d W[H] = sum_jets LeakyReLU(c) * (sum of the jet's output gradients)
        hipLaunchKernelGGL(mdma_jetsum_kernel, dim3(B), dim3(64), 0, p.s, (const float*)(sc + b.dvrow), sc + b.dvj, N);
        PFM_TRY(check_hip(hipGetLastError(), "mdma_jetsum_kernel launch"));
        Jobs J;
        J.gblob = Bw.gblob;
        J.add(cj + 2, 4, 1, sc + b.dvj, 1, 1, d.out_Wc);
        PFM_TRY(J.launch(B, p.s));
    }
    int cur = 0;  // gxc[cur]: d loss / d x_cls_out of the block being processed (none for the last block)
    for (int l = d.layers - 1; l >= 0; --l) {
        const pfm_mdma_block& k = d.block[l];
        float* lb = p.lay(l);
        const float* Xin = p.X(l);
        const float* Hh = lb + w.o_h;
        // x_out = fc1.W[:, :H] h + jet bias + x_in
        PFM_TRY(Bw.colsum(gX, H, nullptr, 0, djb, -1));
        PFM_TRY(Bw.dw(gX, H, Hh, H, k.fc1.W, false));
        PFM_TRY(linear(p, gX, H, k.fc1.WT, -1, H, nullptr, nullptr, nullptr, gH, 0, false));
        {
            ClsPostBwdArgs a;
            a.blob = p.blob; a.djb = djb; a.gxc_next = (l == d.layers - 1) ? nullptr : gxc[cur];
            a.gxo = sc + b.gxo; a.dc2 = sc + b.dc2; a.dout = sc + b.dout; a.datt = sc + b.datt;
            a.W1c = k.fc1.Wc; a.fc2c_W = k.fc2c_W; a.fc1c_W = k.fc1c_W; a.o_W = k.o_W; a.H = H; a.L = L; a.Tg = Tg; a.gcd = gcd; a.dtemb = dtemb;
            hipLaunchKernelGGL(mdma_cls_post_bwd_kernel, dim3(B), dim3(JT), 0, p.s, a);
            PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_post_bwd_kernel launch"));
        }
        PFM_MDMA_ATTN(ca_attn_from_bwd_kernel, (const float*)(lb + w.o_q), (const float*)(lb + w.o_kv), mask, (const float*)(lb + w.o_att),
                      (const float*)(sc + b.datt), sc + b.gq, gkv, N, H, d.heads, 1);
        PFM_TRY(check_hip(hipGetLastError(), "ca_attn_from_bwd_kernel launch (mdma)"));
        PFM_TRY(Bw.colsum(gkv, 2 * H, nullptr, 0, nullptr, k.kv.b));
        PFM_TRY(Bw.dw(gkv, 2 * H, Hh, H, k.kv.W, false));
        PFM_TRY(linear(p, gkv, 2 * H, k.kv.WT, -1, H, nullptr, gH, nullptr, gH, 0, false));
        {
            ClsPreBwdArgs a;
            a.blob = p.blob; a.gq = sc + b.gq; a.pre = lb + w.o_pre; a.xc_in = p.xc(l);
            a.dc = sc + b.dc; a.dgx = sc + b.dgx; a.dpre = sc + b.dpre; a.al = sc + b.al; a.gxc_in = gxc[cur ^ 1];
            a.q_W = k.q_W; a.ln_g = k.ln_g; a.fc0c_W = k.fc0c_W; a.H = H; a.L = L; a.slope = d.neg_slope; a.eps = d.ln_eps;
            a.temb = temb; a.Tg = Tg; a.dtemb = dtemb;
            hipLaunchKernelGGL(mdma_cls_pre_bwd_kernel, dim3(B), dim3(JT), 0, p.s, a);
            PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_pre_bwd_kernel launch"));
        }
        {
            Jobs J;
            J.gblob = Bw.gblob;
            const float* nv = ws + w.nv;
            J.add(p.xc(l + 1), L, L, djb, H, H, k.fc1.Wc);
            J.add(nullptr, 0, 1, djb, H, H, k.fc1.b);
            J.add(lb + w.o_c2, L, L, sc + b.gxo, L, L, k.fc2c_W);
            J.add(nullptr, 0, 1, sc + b.gxo, L, L, k.fc2c_b);
            J.add(lb + w.o_o, H, H, sc + b.dc2, L, L, k.fc1c_W);
            J.add(nv, 1, 1, sc + b.dc2, L, L, k.fc1c_W + (int64_t)H * L);
            J.add(nullptr, 0, 1, sc + b.dc2, L, L, k.fc1c_b);
            J.add(lb + w.o_att, H, H, sc + b.dout, H, H, k.o_W);
            J.add(nullptr, 0, 1, sc + b.dout, H, H, k.o_b);
            J.add(lb + w.o_c, H, H, sc + b.gq, H, H, k.q_W);
            J.add(nullptr, 0, 1, sc + b.gq, H, H, k.q_b);
            J.add(nullptr, 0, 1, sc + b.dgx, H, H, k.ln_g);
            J.add(nullptr, 0, 1, sc + b.dc, H, H, k.ln_b);
            J.add(sc + b.al, L, L, sc + b.dpre, H, H, k.fc0c_W);
            J.add(nullptr, 0, 1, sc + b.dpre, H, H, k.fc0c_b);
            PFM_TRY(J.launch(B, p.s));
        }
        if (Tg || d.c_cat) {  // the time / condition rows of the class-token Linears (KMAJOR: behind their other input rows) and fc1's condition column
            Jobs J;
            J.gblob = Bw.gblob;
            if (Tg) {
                J.add(temb, 64, Tg, sc + b.gxo, L, L, k.fc2c_W + (int64_t)L * L);
                J.add(temb, 64, Tg, sc + b.dc2, L, L, k.fc1c_W + (int64_t)(H + 1 + gcd) * L);
                J.add(tact, 64, Tg, sc + b.dpre, H, H, k.fc0c_W + (int64_t)L * H);
            }
            if (gcd) J.add(cj, 4, 1, sc + b.dc2, L, L, k.fc1c_W + (int64_t)(H + 1) * L);
            if (gcc) {
                J.add(cj + 1, 4, 1, sc + b.gxo, L, L, k.fc2c_W + (int64_t)(L + Tg) * L);
                J.add(cj + 3, 4, 1, sc + b.dpre, H, H, k.fc0c_W + (int64_t)(L + Tg) * H);
            }
            if (lcc) J.add(cj + 1, 4, 1, djb, H, H, k.fc1.Wt);
            if (J.n) PFM_TRY(J.launch(B, p.s));
        }
        cur ^= 1;
        // h = fc0(act(x_in))
        if (Tl || lcc) {  // per-jet bias rows b + Wt . act(temb) + Wc act(cl): bias, time and condition columns get sums over the jets of per-jet column sums
            PFM_TRY(Bw.colsum(gH, H, nullptr, 0, djb, -1));
            Jobs J;
            J.gblob = Bw.gblob;
            J.add(nullptr, 0, 1, djb, H, H, k.fc0.b);
            if (Tl) J.add(tact, 64, Tl, djb, H, H, k.fc0.Wt);
            if (lcc) J.add(cj + 3, 4, 1, djb, H, H, k.fc0.Wc);
            PFM_TRY(J.launch(B, p.s));
            if (dtemb && Tl) {
                hipLaunchKernelGGL(mdma_dtemb_kernel, dim3(B), dim3(JT), 0, p.s, p.blob, k.fc0.Wt, (int64_t)-1, (const float*)djb, temb, dtemb,
                                   Tl, H, 1, d.neg_slope);
                PFM_TRY(check_hip(hipGetLastError(), "mdma_dtemb_kernel launch (fc0)"));
            }
        } else
        PFM_TRY(Bw.colsum(gH, H, nullptr, 0, nullptr, k.fc0.b));
        PFM_TRY(Bw.dw(gH, H, Xin, H, k.fc0.W, true));
        PFM_TRY(linear(p, gH, H, k.fc0.WT, -1, H, nullptr, gX, Xin, gX, 4, false));
    }
    {
        ClsInitBwdArgs a;
        a.blob = p.blob; a.gxc0 = gxc[cur]; a.ea = ws + w.ea; a.eg = ws + w.eg;
        a.da = sc + b.da; a.dg = sc + b.dg; a.dpool = sc + b.dpool; a.ecls_W = d.ecls_W; a.H = H; a.L = L; a.avg_n = d.avg_n;
        hipLaunchKernelGGL(mdma_cls_init_bwd_kernel, dim3(B), dim3(JT), 0, p.s, a);
        PFM_TRY(check_hip(hipGetLastError(), "mdma_cls_init_bwd_kernel launch"));
    }
    {
        const int64_t n4 = (int64_t)p.M * (H >> 2);
        hipLaunchKernelGGL(mdma_embed_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, p.s, (const float*)gX,
                           (const float*)(sc + b.dpool), (const float*)p.X(0), mask, gH, n4, N, H, d.neg_slope);
        PFM_TRY(check_hip(hipGetLastError(), "mdma_embed_bwd_kernel launch"));
    }
    PFM_TRY(Bw.colsum(gH, H, y, F, nullptr, d.emb_Wx));
    PFM_TRY(Bw.colsum(gH, H, nullptr, 0, sc + b.djbt, -1));
    {
        Jobs J;
        J.gblob = Bw.gblob;
        const float* nv = ws + w.nv;
        J.add(ws + w.pooled, H, H, sc + b.da, L, L, d.ecls_W);
        J.add(nv, 1, 1, sc + b.da, L, L, d.ecls_W + (int64_t)H * L);
        J.add(nullptr, 0, 1, sc + b.da, L, L, d.ecls_b);
        J.add(nv, 1, 1, sc + b.dg, L, L, d.cond_W);
        J.add(nullptr, 0, 1, sc + b.dg, L, L, d.cond_b);
        J.add(nullptr, 0, 1, sc + b.djbt, H, H, d.emb_b);
        if (d.time_in_input) J.add(temb, 64, d.t_dim, sc + b.djbt, H, H, d.emb_Wt);
        if (Tl) J.add(temb, 64, Tl, sc + b.djbt, H, H, d.emb_Wt2);
        if (lcc) J.add(cj, 4, 1, sc + b.djbt, H, H, d.emb_Wc);
        if (gcd) {
            J.add(cj, 4, 1, sc + b.da, L, L, d.ecls_W + (int64_t)(H + 1) * L);
            J.add(cj, 4, 1, sc + b.dg, L, L, d.cond_W + L);
        }
        PFM_TRY(J.launch(B, p.s));
    }
    if (dtemb && (d.time_in_input || Tl)) {
        hipLaunchKernelGGL(mdma_dtemb_kernel, dim3(B), dim3(JT), 0, p.s, p.blob, d.time_in_input ? d.emb_Wt : (int64_t)-1,
                           Tl ? d.emb_Wt2 : (int64_t)-1, (const float*)(sc + b.djbt), temb, dtemb, d.t_dim, H, 0, d.neg_slope);
        PFM_TRY(check_hip(hipGetLastError(), "mdma_dtemb_kernel launch (embed)"));
    }
    return 0;
}

}  // namespace mdma
}  // namespace pfm

using namespace pfm;
using namespace pfm::mdma;

extern "C" {

int64_t pfm_mdma_workspace_floats(const pfm_mdma_desc* d, int32_t n_jets, int32_t train) {
    if (validate(d)) return -1;
    return make_ws(*d, n_jets < 1 ? 1 : n_jets, train != 0).total;
}

int64_t pfm_mdma_backward_scratch_floats(const pfm_mdma_desc* d, int32_t n_jets) {
    if (validate(d)) return -1;
    return make_bs(*d, n_jets < 1 ? 1 : n_jets).total;
}

int pfm_mdma_forward(const pfm_mdma_desc* d, const float* blob, const float* t, int32_t per_jet_t, const float* x, const float* cond,
                     const float* mask, float* v_out, int32_t n_jets, float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    p.cond = cond;
    if (n_jets <= 0) return 0;
    if (!blob || !t || !x || !mask || !v_out || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer (MDMA needs the mask)");
    return run_nfe(p, t, per_jet_t ? 1 : 0, x, mask, v_out);
}

int pfm_mdma_sample_rk(const pfm_mdma_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                       int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets, int32_t premask,
                       float* state, float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    p.cond = cond;
    if (const char* e = rk_tableau_error(tab)) return set_err(PFM_E_BADARG, e);
    if (n_jets <= 0) return 0;
    if (n_steps < 1) return set_err(PFM_E_BADARG, "n_steps must be >= 1");
    if (!blob || !t_eval || !dt || !z || !mask || !x_out || !state || !workspace)
        return set_err(PFM_E_BADARG, "NULL device pointer (MDMA needs the mask)");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : (const float*)nullptr, state,
                       n, d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch (mdma)"))) return rc;
    if (p.temb_k) p.temb_k = n_steps * tab->stages;  // t_eval = the embedding table [T][n_steps * stages]: evaluation e starts at t_eval + e
    rc = sample_rk_rows(*tab, t_eval, dt, n_steps, state, n, p.s,
                        [&](const float* tt, const float* xin, float* vout) { return run_nfe(p, tt, 0, xin, mask, vout); });
    if (rc) return rc;
    return check_hip(hipMemcpyAsync(x_out, state, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy of the final state (mdma)");
}

int pfm_mdma_fm_loss_forward(const pfm_mdma_desc* d, const float* blob, int32_t kind, float sigma, const float* t, const float* x,
                             const float* a, const float* b, const float* cond, const float* mask, float* y_out, float* u_out, float* v_out,
                             float* loss_sums, int32_t n_jets, float* workspace, void* stream) {
    Plan p;
    int rc = make_plan(p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    p.cond = cond;
    if (n_jets <= 0) return 0;
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    if (!blob || !t || !x || !a || !mask || !y_out || !u_out || !v_out || !loss_sums || !workspace)
        return set_err(PFM_E_BADARG, "NULL device pointer (MDMA needs the mask)");
    if (kind == 1 && !b) return set_err(PFM_E_BADARG, "CFM needs eps");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf_yu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, kind, sigma, t, x, a, b, mask, y_out, u_out, n,
                       d->n_points * d->features, d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_yu_kernel launch (mdma)"))) return rc;
    if ((rc = run_nfe(p, t, 1, y_out, mask, v_out))) return rc;
    hipLaunchKernelGGL(tf_loss_kernel, dim3(1), dim3(LOSS_T), 0, p.s, (const float*)v_out, (const float*)u_out, mask, loss_sums, n,
                       (int64_t)p.M, 0, (const float*)nullptr, 1);
    return check_hip(hipGetLastError(), "tf_loss_kernel launch (mdma)");
}

int pfm_mdma_fm_loss_backward(const pfm_mdma_desc* d, const float* blob, const float* mask, const float* y, const float* u, const float* v,
                              const float* gscale, float* gblob, int32_t n_jets, float* workspace, float* scratch, void* stream) {
    Bwd B;
    int rc = make_plan(B.p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !mask || !y || !u || !v || !gscale || !gblob || !workspace || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->block[0].fc0.WT < 0) return set_err(PFM_E_BADARG, "blob was packed without the transposed (backward) weight copies");
    B.gblob = gblob;
    B.sc = scratch;
    B.b = make_bs(*d, n_jets);
    if ((d->flags & PFM_MDMA_F_TEMB_GIVEN) &&
        (rc = check_hip(hipMemsetAsync(scratch + B.b.dtemb, 0, (size_t)n_jets * 64 * sizeof(float), (hipStream_t)stream), "memset dtemb")))
        return rc;
    return run_backward(B, mask, y, u, v, gscale);
}

int pfm_mdma_backward_dtemb(const pfm_mdma_desc* d, const float* scratch, int32_t n_jets, float* dtemb, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (!(d->flags & PFM_MDMA_F_TEMB_GIVEN)) return set_err(PFM_E_BADARG, "pfm_mdma_backward_dtemb: the descriptor has no PFM_MDMA_F_TEMB_GIVEN");
    if (n_jets <= 0) return 0;
    if (!scratch || !dtemb) return set_err(PFM_E_BADARG, "NULL device pointer");
    // the scratch rows are 64 floats apart, the caller's t_dim
    return check_hip(hipMemcpy2DAsync(dtemb, (size_t)d->t_dim * sizeof(float), scratch + make_bs(*d, n_jets).dtemb, 64 * sizeof(float),
                                      (size_t)d->t_dim * sizeof(float), (size_t)n_jets, hipMemcpyDeviceToDevice, (hipStream_t)stream),
                     "copy dtemb");
}

}  // extern "C"
