// Backward kernels of the Full-Transformer vector field under the FM / CFM loss (gfx950, fp32 MFMA).
//
// The forward keeps, per layer, x_in, qkv, att (pre-norm attention output), x_mid, dh (dense hidden, post-
// activation) in the train workspace; LayerNorm statistics, attention probabilities and the normalised operands are
// recomputed.  For every Linear  Z = LN(A) W^T + b (+ jet bias)  with upstream gradient dZ:
//   db   column sums                      tf_colsum_kernel (one workgroup per jet; also the per-jet sums that feed the
//                                         context path, and the F-column products of node_embd / the output head)
//   dW   dZ^T LN(A)   (k = particles)     tf_dw_kernel: 128 x 128 tile of dW per workgroup, both operands staged
//                                         through LDS row-major so that ONE ds_read_b128 per operand feeds 16 MFMAs
//                                         (4 output groups x 4 feature groups); partial sums -> fp32 atomics on the
//                                         gradient blob, in the weight's own MFMA_AK order
//   dLN  dZ W          (k = outputs)      the forward Linear kernel on the MFMA_AKT copy of W
//   dA   LayerNorm backward (+ residual gradient, + LeakyReLU' of the producer)   tf_ln_bwd_kernel, 16 lanes per row
// Attention backward is two kernels per (jet, head): A recomputes S^T / P^T per query tile (as the forward), forms
// dS^T in registers and accumulates dQ; B walks key tiles with S / P in the transposed register layout so that dK
// and dV reduce over queries on MFMA.  Row statistics (max, sum, delta) travel from A to B through the scratch.
#pragma once
#include "tf_fwd.h"

namespace pfm {
namespace tf {

// ------------------------------------------------------------------------------------------------
// loss pieces (losses.py:38-77, 101-136)
// ------------------------------------------------------------------------------------------------
static __global__ void tf_yu_kernel(int kind, float sigma, const float* __restrict__ t, const float* __restrict__ x,
                             const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mask,
                             float* __restrict__ y, float* __restrict__ u, int64_t n, int NF, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float tj = t[i / NF];
    const float m = mask ? mask[i / F] : 1.0f;
    const float xv = x[i], zv = a[i];
    const float one_m_sigma = (float)(1.0 - (double)sigma);  // python computes (1 - sigma) in double
    if (kind == 0) {
        const float p = __fmul_rn(__fsub_rn(1.0f, tj), xv);
        const float q = __fmul_rn(__fadd_rn(sigma, __fmul_rn(one_m_sigma, tj)), zv);
        y[i] = __fadd_rn(p, q);
        u[i] = __fmul_rn(__fsub_rn(__fmul_rn(one_m_sigma, zv), xv), m);
    } else if (kind == 2) {  // DroidLoss, losses.py:332-336: y = x + t z, u = z mask
        y[i] = __fadd_rn(xv, __fmul_rn(tj, zv));
        u[i] = __fmul_rn(zv, m);
    } else if (kind == 3) {  // DiffusionLoss, losses.py:260, 272: noisy = signal_rate x + noise_rate z, target = z (z arrives masked);
        const int64_t jet = i / NF;  // b = rates[n_jets][2]
        y[i] = __fadd_rn(__fmul_rn(b[2 * jet], xv), __fmul_rn(b[2 * jet + 1], zv));
        u[i] = zv;
    } else {
        const float mu = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, tj), xv), __fmul_rn(tj, zv));
        y[i] = __fadd_rn(mu, __fmul_rn(sigma, b[i]));
        u[i] = __fmul_rn(__fsub_rn(zv, xv), m);
    }
}

// sums[0] += sum w_jet crit(v-u) ; sums[1] += sum mask.  crit 0: d^2, 1: huber (delta 1); jet_w (or NULL = 1): per-jet weight,
// NF = floats per jet (DiffusionLoss, losses.py:275-288).  ONE workgroup of 1024 threads, sums in a fixed order (thread-strided
// partial sums, wave tree, the 16 wave sums in wave order): the loss is a pure function of v, u, mask -- bit for bit, run to run.
// (Launch with grid 1: LOSS_T threads.)
constexpr int LOSS_T = 1024;
static __global__ __launch_bounds__(LOSS_T) void tf_loss_kernel(const float* __restrict__ v, const float* __restrict__ u,
                                                         const float* __restrict__ mask, float* __restrict__ sums,
                                                         int64_t n, int64_t rows, int crit = 0, const float* __restrict__ jet_w = nullptr,
                                                         int NF = 1) {
    __shared__ float red[2 * (LOSS_T / 64)];
    float sq[4] = {0.f, 0.f, 0.f, 0.f}, mc = 0.f;
    auto term = [&](int64_t i) {
        const float d = v[i] - u[i];
        if (crit || jet_w) {
            const float c = crit ? (fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f) : d * d;
            return c * (jet_w ? jet_w[i / NF] : 1.0f);
        }
        return d * d;
    };
    int64_t i = threadIdx.x;
    for (; i + 3 * LOSS_T < n; i += 4 * LOSS_T) {  // four independent chains: the loads of a round are in flight together
        sq[0] += term(i); sq[1] += term(i + LOSS_T); sq[2] += term(i + 2 * LOSS_T); sq[3] += term(i + 3 * LOSS_T);
    }
    for (; i < n; i += LOSS_T) sq[0] += term(i);
    for (int64_t r = threadIdx.x; r < rows; r += LOSS_T) mc += mask ? mask[r] : 1.0f;
    float s = wave_sum((sq[0] + sq[1]) + (sq[2] + sq[3]));
    mc = wave_sum(mc);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[LOSS_T / 64 + (threadIdx.x >> 6)] = mc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < LOSS_T / 64; ++w) { a += red[w]; b += red[LOSS_T / 64 + w]; }
        sums[0] += a;
        sums[1] += b;
    }
}

// out[c] += sum_{p < P} part[p * stride + c] for c < n, the partials added in a fixed order (P cut into `R` contiguous ranges, one
// per thread row; a range summed with 8 loads in flight; the R range sums joined through LDS in range order): what replaces the
// fp32 atomicAdd of per-workgroup partial sums -- LayerNorm gamma / beta, bias and column sums -- so that every parameter gradient
// is a pure function of the inputs.  Columns c < n0 go to out0[c], the others to out1[c - n0].  256 threads: C columns x R ranges.
template <int C>
static __global__ __launch_bounds__(256) void tf_ordered_sum_kernel(const float* __restrict__ part, int P, int64_t stride, int n,
                                                                    float* __restrict__ out0, int n0, float* __restrict__ out1) {
    constexpr int R = 256 / C;
    __shared__ float comb[256];
    const int cl = threadIdx.x % C, r = threadIdx.x / C;
    const int c = blockIdx.x * C + cl;
    const bool live = c < n;
    int p = (int)((int64_t)P * r / R);
    const int pend = live ? (int)((int64_t)P * (r + 1) / R) : p;
    const float* pp = part + (live ? c : 0);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; p + 8 <= pend; p += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += pp[(int64_t)(p + k) * stride];
    }
    for (; p < pend; ++p) s[0] += pp[(int64_t)p * stride];
    comb[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (r == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) t += comb[k * C + cl];
        if (c < n0) out0[c] += t;
        else out1[c - n0] += t;
    }
}
// (the caller checks hipGetLastError)
static inline void launch_ordered_sum(hipStream_t s, const float* part, int P, int64_t stride, int n, float* out0, int n0, float* out1) {
    if (n <= 0 || P <= 0) return;
    if (n <= 16) hipLaunchKernelGGL(tf_ordered_sum_kernel<16>, dim3((n + 15) / 16), dim3(256), 0, s, part, P, stride, n, out0, n0, out1);
    else hipLaunchKernelGGL(tf_ordered_sum_kernel<64>, dim3((n + 63) / 64), dim3(256), 0, s, part, P, stride, n, out0, n0, out1);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm row statistics (mean, rstd) of A[M][K]
// ------------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(256) void tf_rowstats_kernel(const float* __restrict__ A, int M, float eps,
                                                          float* __restrict__ stats) {
    constexpr int K = 64 * NI;
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    const float* ap = A + (int64_t)min(row, M - 1) * K + 4 * pl;
    f32x4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { v[i] = *reinterpret_cast<const f32x4*>(ap + 64 * i); s += hsum4(v[i]); }
    const float mean = row_sum16(s) / (float)K;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const f32x4 dl = v[i] - mean; ss += hsum4(dl * dl); }
    const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)K + eps);
    if (pl == 0 && row < M) { stats[2 * (int64_t)row] = mean; stats[2 * (int64_t)row + 1] = rstd; }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (row-wise)
// ------------------------------------------------------------------------------------------------
struct LnBwdArgs {
    const float* A;      // [M][K] input of the LayerNorm (a saved activation)
    const float* G;      // [M][K] gradient w.r.t. the LayerNorm output
    const float* add;    // [M][K] gradient joining from the residual path, or nullptr (may alias out)
    float* out;          // [M][K] gradient w.r.t. A, times LeakyReLU'(A) when act
    const float* blob;
    float* gblob;
    int64_t gamma, beta;
    int M, K, act;
    float slope, eps;
    float* part;         // [gridDim.x][2 K] per-workgroup partial sums of d gamma | d beta (summed in block order by
                         // launch_ordered_sum), or nullptr: fp32 atomics on gblob (order, hence the last bits, vary run to run)
};

template <int NI>
__global__ __launch_bounds__(256) void tf_ln_bwd_kernel(LnBwdArgs a) {
    __shared__ float red[16 * 64 * NI];
    const int tid = threadIdx.x, pl = tid & 15, rg = tid >> 4;
    f32x4 gsum[NI], bsum[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { gsum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float invK = 1.0f / (float)a.K;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int row = blockIdx.x * 64 + 16 * pass + rg;
        const bool ok = row < a.M;
        const int64_t ro = (int64_t)min(row, a.M - 1) * a.K + 4 * pl;
        f32x4 x[NI], dn[NI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            { x[i] = *reinterpret_cast<const f32x4*>(a.A + ro + 64 * i); s += hsum4(x[i]); }
        const float mean = row_sum16(s) * invK;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            { const f32x4 dl = x[i] - mean; ss += hsum4(dl * dl); }
        const float rstd = 1.0f / sqrtf(row_sum16(ss) * invK + a.eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            {
                dn[i] = *reinterpret_cast<const f32x4*>(a.G + ro + 64 * i);
                if (!ok) dn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 4 * pl + 64 * i);
                const f32x4 xh = (x[i] - mean) * rstd;
                const f32x4 g = dn[i] * g4;
                c1 += hsum4(g);
                c2 += hsum4(g * xh);
                gsum[i] += dn[i] * xh;
                bsum[i] += dn[i];
                dn[i] = g;
            }
        c1 = row_sum16(c1) * invK;
        c2 = row_sum16(c2) * invK;
        if (ok) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                {
                    const f32x4 xh = (x[i] - mean) * rstd;
                    f32x4 dA = (dn[i] - c1 - xh * c2) * rstd;
                    if (a.add) dA += *reinterpret_cast<const f32x4*>(a.add + ro + 64 * i);
                    if (a.act) {
                        dA.x *= x[i].x > 0.f ? 1.f : a.slope; dA.y *= x[i].y > 0.f ? 1.f : a.slope;
                        dA.z *= x[i].z > 0.f ? 1.f : a.slope; dA.w *= x[i].w > 0.f ? 1.f : a.slope;
                    }
                    *reinterpret_cast<f32x4*>(a.out + ro + 64 * i) = dA;
                }
        }
    }
    // d gamma / d beta: reduce the 16 row groups through LDS, one atomic per column per workgroup
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<f32x4*>(red + rg * a.K + 4 * pl + 64 * i) = which ? bsum[i] : gsum[i];
        __syncthreads();
        for (int c = tid; c < a.K; c += 256) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * a.K + c];
            if (a.part) a.part[(int64_t)blockIdx.x * 2 * a.K + which * a.K + c] = s;
            else atomicAdd(a.gblob + (which ? a.beta : a.gamma) + c, s);
        }
    }
}

// Output head backward, first half: dv = 2 (v-u) gscale, d b3, the gradient w.r.t. the head's (normalised) input
// dn = dv W3, and the normalised rows themselves (for d W3).  16 lanes per row.
struct HeadBwdArgs {
    const float* A;   // [M][K] outp_embd hidden (post-activation)
    const float *v, *u, *gscale;
    float *dv, *dn, *nout;
    const float* blob;
    float* gblob;
    int64_t gamma, beta, W3, b3;
    int M, K, F;
    float eps;
    float* part;      // [gridDim.x][16] per-workgroup partial sums of d b3 (launch_ordered_sum), or nullptr: atomics
};

template <int NI>
__global__ __launch_bounds__(256) void tf_head_bwd_kernel(HeadBwdArgs a) {
    __shared__ float db3[16];
    __shared__ float drow[16][16];  // [row of the workgroup][f]: summed over the rows in row order (no LDS atomics)
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    const bool ok = row < a.M;
    const int64_t ro = (int64_t)min(row, a.M - 1) * a.K + 4 * pl;
    const float gs = a.gscale[0];
    f32x4 x[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        { x[i] = *reinterpret_cast<const f32x4*>(a.A + ro + 64 * i); s += hsum4(x[i]); }
    const float mean = row_sum16(s) / (float)a.K;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        { const f32x4 dl = x[i] - mean; ss += hsum4(dl * dl); }
    const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)a.K + a.eps);
    if (ok) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 4 * pl + 64 * i);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.blob + a.beta + 4 * pl + 64 * i);
                *reinterpret_cast<f32x4*>(a.nout + ro + 64 * i) = (x[i] - mean) * rstd * g4 + b4;
            }
    }
    f32x4 dn[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) dn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int f = 0; f < a.F; ++f) {
        const int64_t e = (int64_t)min(row, a.M - 1) * a.F + f;
        const float d = ok ? 2.0f * (a.v[e] - a.u[e]) * gs : 0.f;
        if (pl == 0) { if (ok) a.dv[e] = d; drow[tid >> 4][f] = d; }
#pragma unroll
        for (int i = 0; i < NI; ++i)
                dn[i] += d * *reinterpret_cast<const f32x4*>(a.blob + a.W3 + (int64_t)f * a.K + 4 * pl + 64 * i);
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<f32x4*>(a.dn + ro + 64 * i) = dn[i];
    }
    __syncthreads();
    if (tid < a.F) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += drow[r][tid];
        if (a.part) a.part[(int64_t)blockIdx.x * 16 + tid] = t;
        else atomicAdd(a.gblob + a.b3 + tid, t);
    }
    (void)db3;
}

// ------------------------------------------------------------------------------------------------
// column sums per jet:  out[o] = sum_{rows of the jet} w(row) Z[row][o],  w = X[row][f] (f = blockIdx.y) or 1
// ------------------------------------------------------------------------------------------------
struct ColsumArgs {
    const float* Z;
    const float* X;     // [M][F] or nullptr
    float* jet_out;     // [jet][jet_stride] plain store, or nullptr
    float* gblob;
    int64_t gb;         // gblob offset of an [F or 1][NO] block receiving the sum over all jets (atomic), or -1
    int64_t jet_stride;
    int ldz, NO, N, F;
    int64_t rows;       // total rows of Z (the last group may be short); 0: every group has N rows
    float* part;        // [jets][gridDim.y][NO] per-jet sums for the sum over all jets (launch_ordered_sum), or nullptr: atomics
};

static __global__ __launch_bounds__(256) void tf_colsum_kernel(ColsumArgs a) {
    __shared__ float red[4 * 768];
    const int tid = threadIdx.x, cg = tid & 63, rg = tid >> 6;
    const int jet = blockIdx.x, f = blockIdx.y;
    const int col0 = blockIdx.z * 768;
    const int ncol = min(768, a.NO - col0), nc4 = ncol >> 2;
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nrows = a.rows ? (int)min((int64_t)a.N, a.rows - (int64_t)jet * a.N) : a.N;
    for (int r = rg; r < nrows; r += 4) {
        const int64_t row = (int64_t)jet * a.N + r;
        const float w = a.X ? a.X[row * a.F + f] : 1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (cg + 64 * i < nc4) acc[i] += w * *reinterpret_cast<const f32x4*>(a.Z + row * a.ldz + col0 + 4 * (cg + 64 * i));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (cg + 64 * i < nc4) *reinterpret_cast<f32x4*>(red + rg * 768 + 4 * (cg + 64 * i)) = acc[i];
    __syncthreads();
    for (int c = tid; c < ncol; c += 256) {
        const float s = (red[c] + red[768 + c]) + (red[2 * 768 + c] + red[3 * 768 + c]);
        if (a.jet_out) a.jet_out[(int64_t)jet * a.jet_stride + col0 + c] = s;
        if (a.gb >= 0) {
            if (a.part) a.part[((int64_t)jet * gridDim.y + f) * a.NO + col0 + c] = s;
            else atomicAdd(a.gblob + a.gb + (int64_t)f * a.NO + col0 + c, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dW += dZ^T LN(A): one 128 x 128 tile of dW per workgroup, particles split over `nsplit` workgroups
// ------------------------------------------------------------------------------------------------
struct DwArgs {
    const float* Z;      // [M][ldz] upstream gradient
    const float* A;      // [M][lda] the Linear's input (before its LayerNorm prologue, if any)
    const float* A2;     // optional second input segment [M][lda2]: input columns K1.. (see LinArgs)
    const float* stats;  // [M][2] mean, rstd of A's rows; nullptr: the Linear has no LayerNorm prologue
    const float* blob;
    float* part;         // [tiles][nsplit][128*128] partial tiles
    int64_t gamma, beta;
    int ldz, lda, lda2, K1, M, NO, K, nsplit, row_tiles;
    int pre_act = 0;     // 1: the Linear's input was LeakyReLU(A) (see LinArgs::pre_act)
    float slope = 0.f;
};

constexpr int DWS = 132;  // LDS row stride (floats)

static __global__ __launch_bounds__(LT, 2) void tf_dw_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const zt = lds;             // [64][DWS]
    float* const at = lds + 64 * DWS;  // [64][DWS]
    const int nkc = (a.K + 127) >> 7;  // 128-wide tiles over K and NO; partial last tiles are zero-filled
    const int split = blockIdx.x % a.nsplit, tile = blockIdx.x / a.nsplit;
    const int to = tile / nkc, tk = tile - to * nkc;
    const bool zin = 128 * to + 4 * (threadIdx.x & 31) < a.NO, ain = 128 * tk + 4 * (threadIdx.x & 31) < a.K;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    const int wo = w >> 1, wk = w & 1;
    const int sc4 = tid & 31, sr = tid >> 5;
    f32x4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool ln = a.stats != nullptr;
    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (ln && ain) {
        g4 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 128 * tk + 4 * sc4);
        b4 = *reinterpret_cast<const f32x4*>(a.blob + a.beta + 128 * tk + 4 * sc4);
    }
    const bool seg2 = a.A2 != nullptr && 128 * tk >= a.K1;
    const float* asrc = seg2 ? a.A2 + (128 * tk - a.K1) + 4 * sc4 : a.A + 128 * tk + 4 * sc4;
    const int ald = seg2 ? a.lda2 : a.lda;
#pragma unroll 1
    for (int rt = split; rt < a.row_tiles; rt += a.nsplit) {
        f32x4 zs[8], as[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rt * BM + sr + 8 * i;
            const int rc = min(row, a.M - 1);
            zs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < a.M && zin) zs[i] = *reinterpret_cast<const f32x4*>(a.Z + (int64_t)rc * a.ldz + 128 * to + 4 * sc4);
            const f32x4 av = ain ? *reinterpret_cast<const f32x4*>(asrc + (int64_t)rc * ald) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (ln) {
                const float mean = a.stats[2 * (int64_t)rc], rstd = a.stats[2 * (int64_t)rc + 1];
                as[i] = (av - mean) * rstd * g4 + b4;
            } else {
                as[i] = a.pre_act ? lrelu4(av, a.slope) : av;
            }
        }
        __syncthreads();  // the previous tile has been consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<f32x4*>(zt + (sr + 8 * i) * DWS + 4 * sc4) = zs[i];
            *reinterpret_cast<f32x4*>(at + (sr + 8 * i) * DWS + 4 * sc4) = as[i];
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 dz = *reinterpret_cast<const f32x4*>(zt + (4 * ks + q) * DWS + 64 * wo + 4 * pl);
            const f32x4 an = *reinterpret_cast<const f32x4*>(at + (4 * ks + q) * DWS + 64 * wk + 4 * pl);
#define PFM_DW_ROW(c, zc)                                                                   \
    acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.x, acc[c][0], 0, 0, 0);         \
    acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.y, acc[c][1], 0, 0, 0);         \
    acc[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.z, acc[c][2], 0, 0, 0);         \
    acc[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.w, acc[c][3], 0, 0, 0);
            PFM_DW_ROW(0, dz.x) PFM_DW_ROW(1, dz.y) PFM_DW_ROW(2, dz.z) PFM_DW_ROW(3, dz.w)
#undef PFM_DW_ROW
        }
    }
    // partial tile -> scratch, accumulator order: float4 (e = 0..3) at (((w*4 + c)*4 + r)*64 + lane); summed over the
    // splits and scattered into the gradient blob by tf_dw_reduce_kernel (atomics on the blob were 10x the GEMM time)
    float* pp = a.part + ((int64_t)tile * a.nsplit + split) * 16384;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v = {acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]};
            *reinterpret_cast<f32x4*>(pp + ((((w * 4 + c) * 4 + r) * 64 + lane) << 2)) = v;
        }
}

// gblob[W block] += sum over splits of the partial tiles.  Element p of a tile: w = p>>12, c = (p>>10)&3, r = (p>>8)&3,
// lane = (p>>2)&63, e = p&3  ->  dW[o][k], o = 128 to + 64 (w>>1) + 4 (4 (lane>>4) + r) + c, k = 128 tk + 64 (w&1) + 4 (lane&15) + e
// grid (256, tiles): a workgroup owns 64 consecutive floats of the tile; 16 thread groups stride over the splits with
// 16-byte loads (16 lanes = 256 contiguous bytes per split), LDS sums the groups.
static __global__ __launch_bounds__(256) void tf_dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ gblob,
                                                           int64_t gW, int NO, int K, int nsplit) {
    __shared__ f32x4 red[256];
    const int nkc = (K + 127) >> 7, nst = K >> 6;
    const int tile = blockIdx.y;
    const int c4 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const float* pp = part + (int64_t)tile * nsplit * 16384 + blockIdx.x * 64 + 4 * c4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int i = g;
    for (; i + 16 < nsplit; i += 32) {
        s0 += *reinterpret_cast<const f32x4*>(pp + (int64_t)i * 16384);
        s1 += *reinterpret_cast<const f32x4*>(pp + (int64_t)(i + 16) * 16384);
    }
    if (i < nsplit) s0 += *reinterpret_cast<const f32x4*>(pp + (int64_t)i * 16384);
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = threadIdx.x >> 2, e = threadIdx.x & 3;
        float s = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) s += red[gg * 16 + cc][e];
        const int p = blockIdx.x * 64 + threadIdx.x;
        const int to = tile / nkc, tk = tile - to * nkc;
        const int w = p >> 12, c = (p >> 10) & 3, r = (p >> 8) & 3, lane = (p >> 2) & 63;
        const int o = 128 * to + 64 * (w >> 1) + 4 * (4 * (lane >> 4) + r) + c;
        const int k = 128 * tk + 64 * (w & 1) + 4 * (lane & 15) + e;
        if (o < NO && k < K)
            gblob[gW + ((int64_t)((o >> 4) * nst + (k >> 6)) * 4 + ((k >> 4) & 3)) * 256 + (((k >> 2) & 3) * 16 + (o & 15)) * 4 + (k & 3)] += s;
    }
}

constexpr int DW_MAX_PARTS = 1024;  // partial 128 x 128 dW tiles in flight per Linear (scratch: DW_MAX_PARTS * 64 KB)

// number of row splits of one dW GEMM: ~2 workgroups per CU over all tiles, every split with the same number of row tiles
inline int dw_splits(int row_tiles, int tiles, int cus) {
    int target = 2 * cus / tiles;
    if (target > DW_MAX_PARTS / tiles) target = DW_MAX_PARTS / tiles;
    if (target < 1) target = 1;
    const int per = (row_tiles + target - 1) / target;
    return (row_tiles + per - 1) / per;
}

inline void launch_dw_reduce(hipStream_t s, const float* part, float* gblob, int64_t gW, int NO, int K, int tiles, int ns) {
    hipLaunchKernelGGL(tf_dw_reduce_kernel, dim3(256, tiles), dim3(256), 0, s, part, gblob, gW, NO, K, ns);
}

// ------------------------------------------------------------------------------------------------
// attention backward
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int attn_bwd_lds_floats(int N) {
    const int np = attn_np16(N);
    return 2 * HD * (np + 4) + 3 * np;
}

// A: per query tile, dQ and the row statistics (max, sum, delta = sum_d dO O)
template <int MAXKT>
__global__ __launch_bounds__(256, 2) void tf_attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                               const float* __restrict__ O, const float* __restrict__ dO,
                                                               float* __restrict__ dqkv, float* __restrict__ stats,
                                                               int N, int D, int heads) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int np = attn_np16(N), nkt = np >> 4, npv = np + 4;
    float* const Kt = lds;             // [HD][npv]
    float* const Vt = Kt + HD * npv;   // [HD][npv]
    float* const mb = Vt + HD * npv;   // [np]
    const int jet = blockIdx.x / heads, h = blockIdx.x - jet * heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)jet * N * ld + h * HD;
    for (int idx = tid; idx < np * 4; idx += 256) {
        const int key = idx >> 2, part = idx & 3;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (key < N) {
            kv = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + D + 4 * part);
            vv = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + 2 * D + 4 * part);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Kt[(4 * part + e) * npv + key] = kv[e];
            Vt[(4 * part + e) * npv + key] = vv[e];
        }
    }
    for (int key = tid; key < np; key += 256) {
        const bool ok = key < N && (mask == nullptr || mask[(int64_t)jet * N + key] != 0.f);
        mb[key] = ok ? 0.f : -__builtin_inff();
    }
    __syncthreads();
    float* const st = stats + (int64_t)blockIdx.x * 3 * np;
    for (int qt = w; qt < nkt; qt += 4) {
        const int qrow = min(qt * 16 + pl, N - 1);
        const int64_t grow = (int64_t)jet * N + qrow;
        f32x4 Qf = *reinterpret_cast<const f32x4*>(base + (int64_t)qrow * ld + 4 * q);
        Qf *= 0.25f;
        const f32x4 dOf = *reinterpret_cast<const f32x4*>(dO + grow * D + h * HD + 4 * q);
        const f32x4 Of = *reinterpret_cast<const f32x4*>(O + grow * D + h * HD + 4 * q);
        float delta = hsum4(dOf * Of);
        delta += __shfl_xor(delta, 16);
        delta += __shfl_xor(delta, 32);
        f32x4 s[MAXKT];
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt)
            if (kt < nkt) {
                f32x4 c = *reinterpret_cast<const f32x4*>(mb + 16 * kt + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[(4 * q + e) * npv + 16 * kt + pl], Qf[e], c, 0, 0, 0);
                s[kt] = c;
            }
        float m = -__builtin_inff();
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt)
            if (kt < nkt) m = fmaxf(fmaxf(fmaxf(s[kt].x, s[kt].y), fmaxf(s[kt].z, s[kt].w)), m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt)
            if (kt < nkt) {
                s[kt].x = __expf(s[kt].x - m); s[kt].y = __expf(s[kt].y - m);
                s[kt].z = __expf(s[kt].z - m); s[kt].w = __expf(s[kt].w - m);
                l += hsum4(s[kt]);
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        if (q == 0 && qt * 16 + pl < N) {
            st[qt * 16 + pl] = m;
            st[np + qt * 16 + pl] = l;
            st[2 * np + qt * 16 + pl] = delta;
        }
        f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt)
            if (kt < nkt) {
                // dP^T tile: rows = keys, column = query
                f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dp = __builtin_amdgcn_mfma_f32_16x16x4f32(Vt[(4 * q + e) * npv + 16 * kt + pl], dOf[e], dp, 0, 0, 0);
                const f32x4 ds = s[kt] * inv * (dp - delta) * 0.25f;
                const f32x4 Kf = *reinterpret_cast<const f32x4*>(Kt + pl * npv + 16 * kt + 4 * q);
                if (kt & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Kf[e], ds[e], dq1, 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Kf[e], ds[e], dq0, 0, 0, 0);
                }
            }
        if (qt * 16 + pl < N) *reinterpret_cast<f32x4*>(dqkv + grow * ld + h * HD + 4 * q) = dq0 + dq1;
    }
}

// B: per key tile, dK and dV (reductions over the queries)
static __global__ __launch_bounds__(256, 2) void tf_attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                                const float* __restrict__ dO, const float* __restrict__ stats,
                                                                float* __restrict__ dqkv, int N, int D, int heads) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int np = attn_np16(N), nkt = np >> 4, npv = np + 4;
    float* const Qt = lds;              // [HD][npv]
    float* const dOt = Qt + HD * npv;   // [HD][npv]
    float* const sm = dOt + HD * npv;   // [np] row max (+inf for padded queries: their P is 0)
    float* const sl = sm + np;          // [np] 1 / row sum
    float* const sd = sl + np;          // [np] delta
    const int jet = blockIdx.x / heads, h = blockIdx.x - jet * heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)jet * N * ld + h * HD;
    for (int idx = tid; idx < np * 4; idx += 256) {
        const int qr = idx >> 2, part = idx & 3;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        if (qr < N) {
            qv = *reinterpret_cast<const f32x4*>(base + (int64_t)qr * ld + 4 * part);
            dv = *reinterpret_cast<const f32x4*>(dO + ((int64_t)jet * N + qr) * D + h * HD + 4 * part);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Qt[(4 * part + e) * npv + qr] = qv[e];
            dOt[(4 * part + e) * npv + qr] = dv[e];
        }
    }
    const float* st = stats + (int64_t)blockIdx.x * 3 * np;
    for (int i = tid; i < np; i += 256) {
        const bool ok = i < N;
        sm[i] = ok ? st[i] : __builtin_inff();
        sl[i] = ok ? 1.0f / st[np + i] : 1.0f;
        sd[i] = ok ? st[2 * np + i] : 0.f;
    }
    __syncthreads();
    for (int kt = w; kt < nkt; kt += 4) {
        const int key = kt * 16 + pl;
        const int krow = min(key, N - 1);
        f32x4 Kf = *reinterpret_cast<const f32x4*>(base + (int64_t)krow * ld + D + 4 * q);
        Kf *= 0.25f;
        const f32x4 Vf = *reinterpret_cast<const f32x4*>(base + (int64_t)krow * ld + 2 * D + 4 * q);
        const bool kok = key < N && (mask == nullptr || mask[(int64_t)jet * N + key] != 0.f);
        const float kb = kok ? 0.f : -__builtin_inff();
        f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
#pragma unroll 2
        for (int qt = 0; qt < nkt; ++qt) {
            // S / dP tiles: rows = queries 16 qt + 4q + r, column = key pl
            f32x4 s = {kb, kb, kb, kb}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(Qt[(4 * q + e) * npv + 16 * qt + pl], Kf[e], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dOt[(4 * q + e) * npv + 16 * qt + pl], Vf[e], dp, 0, 0, 0);
            }
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(sm + 16 * qt + 4 * q);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + 16 * qt + 4 * q);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + 16 * qt + 4 * q);
            f32x4 p;
            p.x = __expf(s.x - m4.x) * l4.x; p.y = __expf(s.y - m4.y) * l4.y;
            p.z = __expf(s.z - m4.z) * l4.z; p.w = __expf(s.w - m4.w) * l4.w;
            const f32x4 ds = p * (dp - d4) * 0.25f;
            const f32x4 dOa = *reinterpret_cast<const f32x4*>(dOt + pl * npv + 16 * qt + 4 * q);
            const f32x4 Qa = *reinterpret_cast<const f32x4*>(Qt + pl * npv + 16 * qt + 4 * q);
            if (qt & 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dOa[e], p[e], dv1, 0, 0, 0);
                    dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Qa[e], ds[e], dk1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dOa[e], p[e], dv0, 0, 0, 0);
                    dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Qa[e], ds[e], dk0, 0, 0, 0);
                }
            }
        }
        if (key < N) {
            float* gp = dqkv + ((int64_t)jet * N + key) * ld + h * HD + 4 * q;
            *reinterpret_cast<f32x4*>(gp + D) = dk0 + dk1;
            *reinterpret_cast<f32x4*>(gp + 2 * D) = dv0 + dv1;
        }
    }
}

inline int attn_bwd_set_lds() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tf_attn_bwd_q_kernel<32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, attn_bwd_lds_floats(512) * 4);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(tf_attn_bwd_kv_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, attn_bwd_lds_floats(512) * 4);
    return (int)e;
}

// ------------------------------------------------------------------------------------------------
// context path: per-jet chain back to ctxt_emdb, then sums over jets of outer products
// ------------------------------------------------------------------------------------------------
struct CtxtBwdArgs {
    const float* blob;
    const float *djb, *chid;  // [jet][nb][Hd], [jet][CH]
    float *dctxt, *dhn, *dhnx, *dpre, *hn;  // [jet][CO], [jet][CH] x 4
    int CH, CO, Hd, nb;
    float slope, eps;
    int64_t cg, cb, c2W;
    int64_t Wc[CTXT_MAX_NB];
};

static __global__ __launch_bounds__(512) void tf_ctxt_bwd_kernel(CtxtBwdArgs a) {
    __shared__ float part[512];
    __shared__ float dcx[64];
    __shared__ float red[8];
    const int tid = threadIdx.x, jet = blockIdx.x;
    const float* __restrict__ blob = a.blob;
    // d ctxt[j] = sum_c sum_o Wc[c][j][o] djb[c][o]
    {
        const int j = tid & 63, p = tid >> 6;
        float acc = 0.f;
        if (j < a.CO) {
            const int o0 = p * (a.Hd >> 3), o1 = o0 + (a.Hd >> 3);
            for (int c = 0; c < a.nb; ++c) {
                const float* dj = a.djb + ((int64_t)jet * a.nb + c) * a.Hd;
                const float* wr = blob + a.Wc[c] + (int64_t)j * a.Hd;
                for (int o = o0; o < o1; ++o) acc = fmaf(wr[o], dj[o], acc);
            }
        }
        part[tid] = acc;
    }
    __syncthreads();
    if (tid < a.CO) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) acc += part[p * 64 + tid];
        dcx[tid] = acc;
        a.dctxt[(int64_t)jet * a.CO + tid] = acc;
    }
    __syncthreads();
    // back through the output Linear and the LayerNorm of ctxt_emdb (CH <= 512: one element per thread)
    const int o = tid;
    const bool ok = o < a.CH;
    const float h = ok ? a.chid[(int64_t)jet * a.CH + o] : 0.f;
    const float mean = block_sum512(h, red) / (float)a.CH;
    const float dl = ok ? h - mean : 0.f;
    const float rstd = 1.0f / sqrtf(block_sum512(dl * dl, red) / (float)a.CH + a.eps);
    const float xh = dl * rstd;
    float dn = 0.f;
    if (ok)
        for (int j = 0; j < a.CO; ++j) dn = fmaf(blob[a.c2W + (int64_t)o * a.CO + j], dcx[j], dn);
    const float gam = ok ? blob[a.cg + o] : 0.f;
    const float g = dn * gam;
    const float c1 = block_sum512(g, red) / (float)a.CH;
    const float c2 = block_sum512(g * xh, red) / (float)a.CH;
    if (ok) {
        const float dh = (g - c1 - xh * c2) * rstd;
        const int64_t e = (int64_t)jet * a.CH + o;
        a.dhn[e] = dn;
        a.dhnx[e] = dn * xh;
        a.hn[e] = xh * gam + blob[a.cb + o];
        a.dpre[e] = dh * (h > 0.f ? 1.f : a.slope);
    }
}

// G[k][o] += sum_jet U[jet][k] V[jet][o]   (U == nullptr: K = 1, weight 1).  Each element has one owner thread; four
// independent partial sums keep four loads of the (latency-bound) walk over the jets in flight.
__device__ __forceinline__ float outer_sum_elem(const float* __restrict__ U, int64_t ldu, int k, const float* __restrict__ V,
                                                int64_t ldv, int o, int n_jets) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    if (U) {
        for (; j + 4 <= n_jets; j += 4) {
            a0 = fmaf(U[j * ldu + k], V[j * ldv + o], a0);
            a1 = fmaf(U[(j + 1) * ldu + k], V[(j + 1) * ldv + o], a1);
            a2 = fmaf(U[(j + 2) * ldu + k], V[(j + 2) * ldv + o], a2);
            a3 = fmaf(U[(j + 3) * ldu + k], V[(j + 3) * ldv + o], a3);
        }
        for (; j < n_jets; ++j) a0 = fmaf(U[j * ldu + k], V[j * ldv + o], a0);
    } else {
        for (; j + 4 <= n_jets; j += 4) {
            a0 += V[j * ldv + o]; a1 += V[(j + 1) * ldv + o]; a2 += V[(j + 2) * ldv + o]; a3 += V[(j + 3) * ldv + o];
        }
        for (; j < n_jets; ++j) a0 += V[j * ldv + o];
    }
    return (a0 + a1) + (a2 + a3);
}

// a list of independent outer sums in one launch: grid (blocks of the largest job, jobs)
// d loss / d y[row][f] = sum_h dZ1[row][h] * Wx[f][h]: the gradient w.r.t. the network's particle input through the particle columns (KMAJOR [F][Hp])
// of its first Linear (EPiC fc_l1, node_embd.input_block) -- what a chain of flows (n_transforms > 1) hands to the flow in front; 16 lanes per row,
// 16 rows per workgroup
static __global__ __launch_bounds__(256) void tf_dy_kernel(const float* __restrict__ dZ1, const float* __restrict__ blob, int64_t l1x,
                                                    float* __restrict__ dy, int64_t M, int F, int Hp) {
    const int tid = threadIdx.x, pl = tid & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (tid >> 4);
    if (row >= M) return;  // (whole 16-lane groups leave together: the DPP row sums below stay inside a group)
    float acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc[f] = 0.f;
    for (int h = 4 * pl; h < Hp; h += 64) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dZ1 + row * Hp + h);
#pragma unroll
        for (int f = 0; f < 16; ++f)
            if (f < F) acc[f] += hsum4(g * *reinterpret_cast<const f32x4*>(blob + l1x + (int64_t)f * Hp + h));
    }
#pragma unroll
    for (int f = 0; f < 16; ++f)
        if (f < F) {
            const float s = row_sum16(acc[f]);
            if (pl == 0) dy[row * F + f] = s;
        }
}


struct OuterJob {
    const float* U;
    const float* V;
    float* G;
    int64_t ldu, ldv;
    int K, NO;
};
constexpr int OUTER_MAX_JOBS = 16;
struct OuterJobs {
    OuterJob job[OUTER_MAX_JOBS];
    int n_jets;
};
static __global__ __launch_bounds__(256) void tf_outer_jobs_kernel(OuterJobs a) {
    const OuterJob& jb = a.job[blockIdx.y];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)jb.K * jb.NO) return;
    const int k = (int)(e / jb.NO), o = (int)(e - (int64_t)k * jb.NO);
    jb.G[e] += outer_sum_elem(jb.U, jb.ldu, k, jb.V, jb.ldv, o, a.n_jets);
}

// the jet-bias rows: for every row c of djb [jet][nb][Hd]:  gblob[gW[c]] ([CO][Hd], K-major) += ctxt^T djb[:, c]  (z = 0)
// and gblob[gb[c]] ([Hd]) += sum_jet djb[:, c]  (z = 1).  grid (ceil(CO * Hd / 256), nb, 2)
struct OuterRowsArgs {
    const float* U;   // ctxt [jet][CO]
    const float* V;   // djb [jet][nb][Hd]
    float* gblob;
    int K, NO, nb, n_jets;
    int64_t gW[CTXT_MAX_NB], gb[CTXT_MAX_NB];
};
static __global__ __launch_bounds__(256) void tf_outer_rows_kernel(OuterRowsArgs a) {
    const int c = blockIdx.y;
    const bool bias = blockIdx.z == 1;
    const int K = bias ? 1 : a.K;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)K * a.NO) return;
    const int k = (int)(e / a.NO), o = (int)(e - (int64_t)k * a.NO);
    const float s = outer_sum_elem(bias ? nullptr : a.U, a.K, k, a.V + (int64_t)c * a.NO, (int64_t)a.nb * a.NO, o, a.n_jets);
    a.gblob[(bias ? a.gb[c] : a.gW[c]) + e] += s;
}

// d loss / d temb[jet][k] = sum_o c1W[k][o] dpre[jet][o] + sum_o n1Wt[k][o] djb[jet][0][o]: the gradient a caller-supplied time embedding
// (PFM_*_F_TEMB_GIVEN) receives through the context network's first Linear and the time columns of node_embd.  One workgroup per jet.
static __global__ __launch_bounds__(64) void tf_dtemb_kernel(const float* __restrict__ blob, const float* __restrict__ dpre,
                                                             const float* __restrict__ djb, float* __restrict__ dtemb, int64_t c1W,
                                                             int64_t n1Wt, int T, int CH, int Hd, int64_t jbs) {
    const int jet = blockIdx.x, k = threadIdx.x;
    if (k >= T) return;
    float a0 = 0.f, a1 = 0.f;
    const float* dp = dpre + (int64_t)jet * CH;
    for (int o = 0; o + 1 < CH; o += 2) {
        a0 = fmaf(blob[c1W + (int64_t)k * CH + o], dp[o], a0);
        a1 = fmaf(blob[c1W + (int64_t)k * CH + o + 1], dp[o + 1], a1);
    }
    if (CH & 1) a0 = fmaf(blob[c1W + (int64_t)k * CH + CH - 1], dp[CH - 1], a0);
    if (n1Wt >= 0) {
        const float* dj = djb + (int64_t)jet * jbs;
        for (int o = 0; o + 1 < Hd; o += 2) {
            a0 = fmaf(blob[n1Wt + (int64_t)k * Hd + o], dj[o], a0);
            a1 = fmaf(blob[n1Wt + (int64_t)k * Hd + o + 1], dj[o + 1], a1);
        }
    }
    dtemb[(int64_t)jet * T + k] = a0 + a1;
}

// Parameter gradients of the per-jet context path in two launches: the jet-bias rows (Wc, bias of every Linear that
// takes the context) and the context network itself (+ the time columns of node_embd).
struct CtxtGradIn {
    const float *ctxt, *djb, *temb, *cond;             // [jet][CO], [jet][nb][Hd], [jet][64], [jet][C]
    const float *hn, *dctxt, *dhnx, *dhn, *dpre;       // outputs of tf_ctxt_bwd_kernel
    float* gblob;
    int n_jets, nb, Hd, CO, CH, T, C;
    int64_t gW[CTXT_MAX_NB], gb[CTXT_MAX_NB];          // Wc / bias blocks of the nb jet-bias rows
    int64_t n1Wt, c2W, c2b, cgamma, cbeta, c1W, c1b;   // n1Wt < 0: no time columns
};

inline void launch_ctxt_param_grads(const CtxtGradIn& g, hipStream_t s) {
    OuterRowsArgs r;
    r.U = g.ctxt; r.V = g.djb; r.gblob = g.gblob; r.K = g.CO; r.NO = g.Hd; r.nb = g.nb; r.n_jets = g.n_jets;
    for (int c = 0; c < g.nb; ++c) { r.gW[c] = g.gW[c]; r.gb[c] = g.gb[c]; }
    hipLaunchKernelGGL(tf_outer_rows_kernel, dim3((g.CO * g.Hd + 255) / 256, g.nb, 2), dim3(256), 0, s, r);
    OuterJobs j;
    j.n_jets = g.n_jets;
    int n = 0, most = 0;
    auto add = [&](const float* U, int64_t ldu, int K, const float* V, int64_t ldv, int NO, int64_t off) {
        j.job[n++] = OuterJob{U, V, g.gblob + off, ldu, ldv, K, NO};
        if (K * NO > most) most = K * NO;
    };
    const int64_t jbs = (int64_t)g.nb * g.Hd;
    if (g.n1Wt >= 0) add(g.temb, 64, g.T, g.djb, jbs, g.Hd, g.n1Wt);
    add(g.hn, g.CH, g.CH, g.dctxt, g.CO, g.CO, g.c2W);
    add(nullptr, 0, 1, g.dctxt, g.CO, g.CO, g.c2b);
    add(nullptr, 0, 1, g.dhnx, g.CH, g.CH, g.cgamma);
    add(nullptr, 0, 1, g.dhn, g.CH, g.CH, g.cbeta);
    add(g.temb, 64, g.T, g.dpre, g.CH, g.CH, g.c1W);
    if (g.C > 0) add(g.cond, g.C, g.C, g.dpre, g.CH, g.CH, g.c1W + (int64_t)g.T * g.CH);
    add(nullptr, 0, 1, g.dpre, g.CH, g.CH, g.c1b);
    hipLaunchKernelGGL(tf_outer_jobs_kernel, dim3((most + 255) / 256, n), dim3(256), 0, s, j);
}

}  // namespace tf
}  // namespace pfm
