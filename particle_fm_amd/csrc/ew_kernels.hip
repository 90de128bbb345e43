// gfx950 kernels + C ABI (include/pfm_epicw.h): EPiC vector field at widths beyond the jet-resident kernel.
//
// Reference graph: particle_fm/models/components/epic.py:304-391 (EPiC_encoder.forward), :85-203 (EPiC_layer.forward).
// The Linears run on tf_linear_kernel (fp32 MFMA, tf_fwd.h); this file adds the per-jet pieces: time embedding +
// conditioning rows, masked mean / sum pooling, the F-output head, and the launch sequence.
#include <hip/hip_runtime.h>

#include "pfm_epicw.h"
#include "tf_fwd.h"
#include "tf_bwd.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

namespace ew {
using namespace pfm::tf;

// P[jet][0 .. 256 + Hp) = [temb | cond | 0 ; 0 (g) ; 0 (g1)]
__global__ __launch_bounds__(256) void ew_prep_kernel(const float* __restrict__ blob, int64_t freqs, const float* __restrict__ t,
                                                      int t_stride, const float* __restrict__ cond, float* __restrict__ P, int T,
                                                      int C, int ldp, int64_t pstride, int sincos, int temb_k) {
    const int jet = blockIdx.x;
    float* row = P + (int64_t)blockIdx.y * pstride + (int64_t)jet * ldp;  // blockIdx.y: stage copy (train layout)
    for (int c = threadIdx.x; c < ldp; c += 256) {
        float v = 0.f;
        if (c < T && temb_k) {  // PFM_EW_F_TEMB_GIVEN: `t` holds the embedding, element c of the jet's row at c * temb_k
            v = t[(int64_t)jet * t_stride + (int64_t)c * temb_k];
        } else if (c < T) {
            // time_emb.py:90-96, exact fp32 op order ((t + min) * f) * pi / (max + min)
            const float tj = t[(int64_t)jet * t_stride], f = blob[freqs + c];
            if (sincos) {  // flow_matching_module.py:208-211 (table = [f ; f], f = 2^k pi)
                const float arg = __fmul_rn(f, tj);
                v = 2 * c < T ? cosf(arg) : sinf(arg);
            } else {
                v = cosf(__fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(tj, 0.0f), f), 3.14159274101257324f), 1.0f));
            }
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

// compact pooling: the jet's rows are [off[jet], off[jet+1]), all valid
__global__ __launch_bounds__(256) void ew_pool_compact_kernel(const float* __restrict__ X, const int* __restrict__ off,
                                                              float* __restrict__ Q, int Hp, float scale) {
    __shared__ float red[4 * 512];
    const int tid = threadIdx.x, cg = tid & 63, rg = tid >> 6, jet = blockIdx.x;
    const int nc4 = Hp >> 2, r0 = off[jet], r1 = off[jet + 1];
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool c0 = cg < nc4, c1 = cg + 64 < nc4;
    int r = r0 + rg;
    for (; r + 28 < r1; r += 32) {  // eight rows in flight per thread
        f32x4 v[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* xp = X + (int64_t)(r + 4 * u) * Hp + 4 * cg;
            v[u][0] = c0 ? *reinterpret_cast<const f32x4*>(xp) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[u][1] = c1 ? *reinterpret_cast<const f32x4*>(xp + 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        acc[0] += ((v[0][0] + v[1][0]) + (v[2][0] + v[3][0])) + ((v[4][0] + v[5][0]) + (v[6][0] + v[7][0]));
        acc[1] += ((v[0][1] + v[1][1]) + (v[2][1] + v[3][1])) + ((v[4][1] + v[5][1]) + (v[6][1] + v[7][1]));
    }
    for (; r + 12 < r1; r += 16) {  // four
        f32x4 v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* xp = X + (int64_t)(r + 4 * u) * Hp + 4 * cg;
            v[u][0] = c0 ? *reinterpret_cast<const f32x4*>(xp) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[u][1] = c1 ? *reinterpret_cast<const f32x4*>(xp + 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        acc[0] += (v[0][0] + v[1][0]) + (v[2][0] + v[3][0]);
        acc[1] += (v[0][1] + v[1][1]) + (v[2][1] + v[3][1]);
    }
    for (; r < r1; r += 4) {
        const float* xp = X + (int64_t)r * Hp + 4 * cg;
        if (c0) acc[0] += *reinterpret_cast<const f32x4*>(xp);
        if (c1) acc[1] += *reinterpret_cast<const f32x4*>(xp + 256);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (cg + 64 * i < nc4) *reinterpret_cast<f32x4*>(red + rg * 512 + 4 * (cg + 64 * i)) = acc[i];
    __syncthreads();
    const float nv = (float)(r1 - r0);
    for (int c = tid; c < Hp; c += 256) {
        const float s = (red[c] + red[512 + c]) + (red[1024 + c] + red[1536 + c]);
        Q[(int64_t)jet * 2 * Hp + c] = s / nv;
        Q[(int64_t)jet * 2 * Hp + Hp + c] = s * scale;
    }
}
// ------------------------------------------------------------------------------------------------
// The per-jet chain of a stage at inference: pool -> fc_global1 -> fc_global2 -> jet-bias rows, on the matrix pipe, 16 jets per tile.
//   Q  = [masked mean | masked sum * scale] of the jet's rows of X   (ew_pool*_kernel)   epic.py:108-117 / :331-339
//   g1 = lrelu(Wg1 . [P256 | Q] + b)                                                    epic.py:180-182 / :375-377
//   g  = lrelu(Wg2 . [P256 | g1] + b (+ g))                                             epic.py:184-186 / :378-380
//   JB = Wjb . P256 + b   (the time / conditioning / broadcast-g columns of fc_local1 | fc_local2 for this jet)
// History: as row GEMMs over the B per-jet rows these were 6-7 launches per layer (round 1); round 2 gave every jet its own workgroup
// that ran the whole chain on the VALU (one launch, but each workgroup streamed the chain's ~1.9 MB of weights for ONE input vector:
// 25 us on as many CUs as there are jets, 29 % of cfg 5's wall time by a timing-only build without it).  Here the jets are the 16 columns of the MFMA's
// B operand: lane (n, q) of a wave reads 4 consecutive inputs of jet n straight from the jet's rows in global memory (16 columns of
// K = one "unit" = 4 v_mfma_f32_16x16x4_f32 on one MFMA_AK float4 of weights), no LDS staging, no barrier:
//   ew_g1_kernel    fc_global1 (fc_g1):  grid (jet groups, Hp / 64); two waves share a 16-output block, half of K each;
//                   K = [P256 | mean | sum] with the all-zero units of P256 skipped (only ceil((T + C) / 16) + ceil(latent / 16) of its
//                   16 units hold values);
//   ew_g2jb_kernel  fc_global2 (fc_g2) for the group -- its one to eight 16-output blocks, K split over the four waves, partials joined
//                   in wave order through LDS -- and the jet-bias rows of a slice of 128 outputs: the accumulator layout of fc_global2's
//                   result (lane (n, q): outputs 4 q .. 4 q + 3 of jet n) IS the B-operand layout of the unit that feeds it into the
//                   jet-bias GEMM, so the new latent vector never leaves registers.
// The latent state g lives in two [B][128] buffers (read G[l & 1], write G[(l + 1) & 1]: the workgroups of a group all read g_old while
// one of them writes g_new); the pooled vector comes from ew_pool*_kernel.  Three small launches per layer instead of one long one: the
// CUs stay free for the other half-batch's particle Linears.  A jet's values depend on its own column only: batch-independent bits.
// ------------------------------------------------------------------------------------------------
struct SkArgs {
    const float *blob, *P, *Q;  // P [B][ldp]: the prep row (temb | cond | 0 ...) and, from column 256 on, fc_global1's output; Q [B][2 Hp]
    const float* Gin;           // [B][128] latent state (nullptr: none -- the stem)
    float *Gout, *JB;
    pfm_ew_lin lin1, lin2, jb;
    int B, Hp, ldp, n_tc, nl, residual, do_jb;  // n_tc = ceil((T + C) / 16), nl = ceil(latent / 16)
    float slope;
};

#define PFM_SK_MFMA4(acc, wv, iv)                                             \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((wv).x, (iv).x, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((wv).y, (iv).y, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((wv).z, (iv).z, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((wv).w, (iv).w, acc, 0, 0, 0);

constexpr int G1U = 24;  // units a wave of ew_g1_kernel holds in registers: half of K <= 16 + 2 * 512 / 16 units -> Hp <= 256 in one pass
__global__ __launch_bounds__(512) void ew_g1_kernel(SkArgs a) {
    // eight waves: wave w and wave w + 4 share output block 4 blockIdx.y + (w & 3), each takes half of the K units with ALL of its
    // operand loads in flight before the first MFMA (the launch is one load round trip + ~100 MFMAs long); the upper half's partial
    // sums cross through LDS
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, q = lane >> 4;
    const int ob = 4 * blockIdx.y + (w & 3), half = w >> 2;
    const int jet0 = 16 * blockIdx.x + n, jet = min(jet0, a.B - 1);
    const int ks = (256 + 2 * a.Hp) >> 6;
    const float* wrow = a.blob + a.lin1.W + (int64_t)ob * ks * 1024 + lane * 4;
    const float* pP = a.P + (int64_t)jet * a.ldp + 4 * q;
    const float* pG = a.Gin ? a.Gin + (int64_t)jet * 128 + 4 * q : nullptr;
    const float* pQ = a.Q + (int64_t)jet * 2 * a.Hp + 4 * q;
    const int u1 = a.n_tc, u2 = u1 + (pG ? a.nl : 0), NU = u2 + (a.Hp >> 3);
    const int per = (NU + 1) >> 1, ua = half * per, ub = min(NU, ua + per);  // this wave's units [ua, ub)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (!half) acc0 = *reinterpret_cast<const f32x4*>(a.blob + a.lin1.b + 16 * ob + 4 * q);
#pragma unroll 1
    for (int u0 = ua; u0 < ub; u0 += G1U) {
        f32x4 wv[G1U], iv[G1U];
#pragma unroll
        for (int i = 0; i < G1U; ++i) {
            const int u = u0 + i;  // wave-uniform
            if (u >= ub) continue;
            int cu;
            const float* ip;
            if (u < u1) { cu = u; ip = pP + 16 * u; }
            else if (u < u2) { cu = 8 + (u - u1); ip = pG + 16 * (u - u1); }
            else { cu = 16 + (u - u2); ip = pQ + 16 * (u - u2); }
            wv[i] = *reinterpret_cast<const f32x4*>(wrow + cu * 256);
            iv[i] = *reinterpret_cast<const f32x4*>(ip);
        }
#pragma unroll
        for (int i = 0; i < G1U; i += 2) {
            if (u0 + i < ub) { PFM_SK_MFMA4(acc0, wv[i], iv[i]) }
            if (u0 + i + 1 < ub) { PFM_SK_MFMA4(acc1, wv[i + 1], iv[i + 1]) }
        }
    }
    acc0 += acc1;
    if (half) red[w & 3][lane] = acc0;
    __syncthreads();
    if (!half && jet0 < a.B)
        *reinterpret_cast<f32x4*>(const_cast<float*>(a.P) + (int64_t)jet * a.ldp + 256 + 16 * ob + 4 * q) = lrelu4(acc0 + red[w][lane], a.slope);
}

template <int NL>
__global__ __launch_bounds__(256) void ew_g2jb_kernel(SkArgs a) {
    __shared__ f32x4 red[NL][4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, q = lane >> 4;
    const int jet0 = 16 * blockIdx.x + n, jet = min(jet0, a.B - 1);
    const float* pP = a.P + (int64_t)jet * a.ldp + 4 * q;
    // the jet-bias GEMM's first operands do not depend on fc_global2: requested ahead of it
    const int nob_jb = a.Hp >> 3;  // 2 Hp / 16
    const int ob0 = 8 * blockIdx.y + 2 * w;
    f32x4 jw[2][8], ji[8], jacc[2];
    if (a.do_jb) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < a.n_tc) ji[u] = *reinterpret_cast<const f32x4*>(pP + 16 * u);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            const int ob = min(ob0 + bi, nob_jb - 1);
            const float* wrow = a.blob + a.jb.W + (int64_t)ob * 4 * 1024 + lane * 4;
            jacc[bi] = *reinterpret_cast<const f32x4*>(a.blob + a.jb.b + 16 * ob + 4 * q);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < a.n_tc) jw[bi][u] = *reinterpret_cast<const f32x4*>(wrow + u * 256);
        }
    }
    // ---- fc_global2: units = n_tc of the prep row + Hp / 16 of fc_global1's output; wave w takes units w, w + 4, ... ----
    const int ks2 = a.ldp >> 6, NU2 = a.n_tc + (a.Hp >> 4);
#pragma unroll
    for (int lb = 0; lb < NL; ++lb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (lb < a.nl) {
            const float* wrow = a.blob + a.lin2.W + (int64_t)lb * ks2 * 1024 + lane * 4;
#pragma unroll 1
            for (int u0 = w; u0 < NU2; u0 += 16) {
                f32x4 wv[4], iv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = u0 + 4 * i;
                    if (u >= NU2) continue;
                    const int cu = u < a.n_tc ? u : 16 + (u - a.n_tc);
                    wv[i] = *reinterpret_cast<const f32x4*>(wrow + cu * 256);
                    iv[i] = *reinterpret_cast<const f32x4*>(pP + 16 * cu);  // (column 16 cu of the row: 256 + ... for fc_global1's output)
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (u0 + 4 * i < NU2) { PFM_SK_MFMA4(acc, wv[i], iv[i]) }
            }
        }
        red[lb][w][lane] = acc;
    }
    __syncthreads();
    f32x4 g[NL];
#pragma unroll
    for (int lb = 0; lb < NL; ++lb) {
        if (lb >= a.nl) continue;
        f32x4 v = (red[lb][0][lane] + red[lb][1][lane]) + (red[lb][2][lane] + red[lb][3][lane]);
        v += *reinterpret_cast<const f32x4*>(a.blob + a.lin2.b + 16 * lb + 4 * q);
        if (a.residual) v += *reinterpret_cast<const f32x4*>(a.Gin + (int64_t)jet * 128 + 16 * lb + 4 * q);  // epic.py:184-186
        g[lb] = lrelu4(v, a.slope);
        if (blockIdx.y == 0 && w == 0 && jet0 < a.B) *reinterpret_cast<f32x4*>(a.Gout + (int64_t)jet * 128 + 16 * lb + 4 * q) = g[lb];
    }
    if (!a.do_jb) return;
    // ---- jet-bias rows: JB[jet] = Wjb . [temb | cond ; g_new] + b, two 16-output blocks per wave ----
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int ob = ob0 + bi;
        if (ob >= nob_jb) continue;
        const float* wrow = a.blob + a.jb.W + (int64_t)ob * 4 * 1024 + lane * 4;
        f32x4 gw[NL];
#pragma unroll
        for (int lb = 0; lb < NL; ++lb)
            if (lb < a.nl) gw[lb] = *reinterpret_cast<const f32x4*>(wrow + (8 + lb) * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < a.n_tc) { PFM_SK_MFMA4(jacc[bi], jw[bi][u], ji[u]) }
#pragma unroll
        for (int lb = 0; lb < NL; ++lb)
            if (lb < a.nl) { PFM_SK_MFMA4(jacc[bi], gw[lb], g[lb]) }
        if (jet0 < a.B) *reinterpret_cast<f32x4*>(a.JB + (int64_t)jet * 2 * a.Hp + 16 * ob + 4 * q) = jacc[bi];
    }
}
#undef PFM_SK_MFMA4

// masked rows of the output: 0 (or the state they start from), NaN for a jet without any valid particle (the
// reference's 0/0 mean poisons the whole jet, epic.py:331-339)
__global__ __launch_bounds__(256) void ew_fill_masked_kernel(const float* __restrict__ mask, const int* __restrict__ cnt,
                                                             const float* __restrict__ base, float* __restrict__ dst, int64_t M,
                                                             int N, int F) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * F) return;
    const int64_t row = i / F;
    if (mask[row] != 0.f) return;
    dst[i] = cnt[row / N] == 0 ? __builtin_nanf("") : (base ? base[i] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// ew_pair_kernel (round 4): the two local Linears of an EPiC layer for a 32-row tile in ONE workgroup (inference, fp32 operands)
//     l1 = lrelu(W1 x + jb1[jet]),   x' = lrelu(W2 l1 + jb2[jet] + x)                         epic.py:194-200
// Both are row-local, so the hidden rows never leave the CU: x sits in an LDS panel (the residual comes from there too), l1 in a second one.
// The two tf_linear_kernel launches it replaces spent 41 us each on 3.9 GFLOP (0.61 of the fp32 MFMA peak: 64-output workgroups, one
// 16-row A operand per wave -- 8 MFMAs per pair of operand reads -- and the hidden tensor's round trip through HBM between them).  Here a
// wave owns NSW = Hp / 64 sixteen-output groups (w, w + 4, ...) of both Linears: per 16 k it reads TWO B operands (the tile's two 16-row
// halves) and issues 8 NSW MFMAs, with the A operands of the next 16 k requested a step ahead (two register sets, static indices).
// Same products and sums in the same order as the two launches (accumulators start from the jet bias, k ascending): bit-identical.
// LDS: 2 x 32 x Hp floats (80 KB at Hp = 320: two workgroups per CU).  Weights: MFMA_AK blocks (pfm_tf.h).
// ------------------------------------------------------------------------------------------------
struct PairArgs {
    const float* blob;
    const float* X;      // [M][ldx] input rows (and the residual)
    float* out;          // [M][ldx] (may be X: a workgroup reads its rows before it writes them)
    const float* jb;     // [jets][jb_stride]: jb1 at 0, jb2 at Hp
    const int* rowjet;   // row -> jet (compacted rows) or nullptr: jet = row / N
    const int* m_dev;    // device-side row count or nullptr
    int64_t blob_floats, W1, W2, jb_stride;
    int ldx, M, N;
    float slope;
};

template <int NSW, int TPW>
__global__ __launch_bounds__(256, TPW == 1 ? 3 : 2) void ew_pair_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RB = 16 * TPW, Hp = 64 * NSW, NKT = Hp / 16, NST = Hp / 64;
    float* const XP = lds;                  // NSW slices of RB x 64, 16-byte slots XOR-swizzled with (row & 15)
    float* const HP_ = lds + NSW * RB * 64;  // the hidden rows, same layout
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.m_dev ? *a.m_dev : a.M;
    const int row0 = blockIdx.x * RB;
    if (row0 >= M) return;
    const blob_rsrc rs = make_blob_rsrc(a.blob, a.blob_floats);
    // A operand (g, kt): 16 outputs 16 g .. x k = 16 kt .. 16 kt + 15 of an MFMA_AK block
    auto request = [&](f32x4 (&af)[NSW], int64_t W, int kt) {
#pragma unroll
        for (int s = 0; s < NSW; ++s) af[s] = bload4(rs, W + ((int64_t)((w + 4 * s) * NST + (kt >> 2)) * 4 + (kt & 3)) * 256, lane * 16);
    };
    // FOUR operand sets, each requested three 16-k steps ahead of its use: with two (one step ahead) hipcc sank the requests into the middle
    // of the step before -- the set's previous reader is the MFMA block just issued -- and every step began with vmcnt waits
    f32x4 af[4][NSW];
    request(af[0], a.W1, 0);
    request(af[1], a.W1, 1);
    request(af[2], a.W1, 2);
    // ---- the tile's rows -> XP (coalesced float4 reads; rows past M repeat the last one, their results are never stored) ----
    for (int u = tid; u < RB * (Hp / 4); u += 256) {
        const int r = u / (Hp / 4), c4 = u - r * (Hp / 4);  // float4 c4 of row r: slice c4 >> 4, slot c4 & 15
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + (int64_t)min(row0 + r, M - 1) * a.ldx + 4 * c4);
        *reinterpret_cast<f32x4*>(XP + (c4 >> 4) * (RB * 64) + r * 64 + (((c4 & 15) ^ (r & 15)) << 2)) = v;
    }
    // jet of this lane's rows (tile parts t = 0 .. TPW - 1)
    int jet[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int row = min(row0 + 16 * t + pl, M - 1);
        jet[t] = a.rowjet ? a.rowjet[row] : row / a.N;
    }
    f32x4 acc[NSW][TPW];
    auto init = [&](int col0) {  // accumulators start from the jet-bias rows (they carry the bias)
#pragma unroll
        for (int s = 0; s < NSW; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[s][t] = *reinterpret_cast<const f32x4*>(a.jb + (int64_t)jet[t] * a.jb_stride + col0 + 16 * (w + 4 * s) + 4 * q);
    };
    auto bfrag = [&](f32x4 (&B)[TPW], const float* P, int kt) {
        const float* p0 = P + (kt >> 2) * (RB * 64) + pl * 64 + (((4 * (kt & 3) + q) ^ pl) << 2);
#pragma unroll
        for (int t = 0; t < TPW; ++t) B[t] = *reinterpret_cast<const f32x4*>(p0 + t * 16 * 64);
    };
    auto mma = [&](const f32x4 (&af)[NSW], const f32x4 (&B)[TPW]) {
#define PFM_EW_STEP(c)                                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < NSW; ++s)                                                                                      \
        _Pragma("unroll") for (int t = 0; t < TPW; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s].c, B[t].c, acc[s][t], 0, 0, 0);
        PFM_EW_STEP(x) PFM_EW_STEP(y) PFM_EW_STEP(z) PFM_EW_STEP(w)
#undef PFM_EW_STEP
    };
    // one Linear over the panel P: k ascending in steps of 16; on entry af[0..2] hold steps 0..2 of W.  Wnext (or -1): the block whose steps
    // 0..2 are requested behind the last ones of this
    static_assert(NKT % 4 == 0, "the step loop is unrolled by the four operand sets");
    auto gemm = [&](const float* P, int64_t W, int64_t Wnext) {
        f32x4 Ba[TPW], Bb[TPW];
        bfrag(Ba, P, 0);
#pragma unroll 1
        for (int kt = 0; kt < NKT; kt += 4) {
            const bool more = kt + 4 < NKT;
            request(af[3], W, kt + 3);
            bfrag(Bb, P, kt + 1);
            mma(af[0], Ba);
            if (more) request(af[0], W, kt + 4); else if (Wnext >= 0) request(af[0], Wnext, 0);
            bfrag(Ba, P, kt + 2);
            mma(af[1], Bb);
            if (more) request(af[1], W, kt + 5); else if (Wnext >= 0) request(af[1], Wnext, 1);
            bfrag(Bb, P, kt + 3);
            mma(af[2], Ba);
            if (more) { request(af[2], W, kt + 6); bfrag(Ba, P, kt + 4); } else if (Wnext >= 0) request(af[2], Wnext, 2);
            mma(af[3], Bb);
        }
    };
    init(0);
    __syncthreads();  // XP complete
    gemm(XP, a.W1, a.W2);
    // ---- hidden rows -> HP_ ----
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const int o = 16 * (w + 4 * s) + 4 * q;  // 4 consecutive hidden columns: slice o >> 6, slot (o & 63) >> 2
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int r = 16 * t + pl;
            *reinterpret_cast<f32x4*>(HP_ + (o >> 6) * (RB * 64) + r * 64 + ((((o & 63) >> 2) ^ (r & 15)) << 2)) = lrelu4(acc[s][t], a.slope);
        }
    }
    init(Hp);
    __syncthreads();  // HP_ complete
    gemm(HP_, a.W2, -1);
    // ---- x' = lrelu(acc + x): the residual from the X panel ----
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const int o = 16 * (w + 4 * s) + 4 * q;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int r = 16 * t + pl, row = row0 + r;
            f32x4 v = acc[s][t] + *reinterpret_cast<const f32x4*>(XP + (o >> 6) * (RB * 64) + r * 64 + ((((o & 63) >> 2) ^ (r & 15)) << 2));
            v = lrelu4(v, a.slope);
            if (row < M) *reinterpret_cast<f32x4*>(a.out + (int64_t)row * a.ldx + o) = v;
        }
    }
}

// the fused pair where it applies (inference, fp32 operands, Hp a multiple of 64 up to 512, enough row tiles to fill the chip); false: the
// caller launches the two Linears.  PFM_EW_PAIR=0 (diagnostics) switches it off.
inline bool launch_pair(const PairArgs& a, int Hp, int cus, hipStream_t s) {
    static int on = -1, tpw_env = 0;
    if (on < 0) {
        const char* e = getenv("PFM_EW_PAIR");      // diagnostics: 0 = the two launches
        const char* t = getenv("PFM_EW_PAIR_TPW");  // diagnostics: 1 / 2 = 16- / 32-row tiles
        on = e ? atoi(e) : 1;
        tpw_env = t ? atoi(t) : 0;
    }
    if (!on || Hp % 64 != 0 || (int64_t)(a.M + 31) / 32 < 2 * (int64_t)cus) return false;
    const int tpw = tpw_env ? tpw_env : 1;
    const dim3 grid((a.M + 16 * tpw - 1) / (16 * tpw)), block(256);
    const int lds = 2 * 16 * tpw * Hp * 4;
#define PFM_EW_PAIR_LAUNCH(NSW, TPW)                                                                                                       \
    {                                                                                                                                      \
        static bool attr = false;                                                                                                          \
        if (!attr) {                                                                                                                       \
            hipFuncSetAttribute(reinterpret_cast<const void*>(ew_pair_kernel<NSW, TPW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            attr = true;                                                                                                                   \
        }                                                                                                                                  \
        hipLaunchKernelGGL((ew_pair_kernel<NSW, TPW>), grid, block, lds, s, a);                                                            \
        return true;                                                                                                                       \
    }
#define PFM_EW_PAIR_CASE(NSW) \
    case NSW:                 \
        if (tpw == 1) PFM_EW_PAIR_LAUNCH(NSW, 1) else PFM_EW_PAIR_LAUNCH(NSW, 2)
    switch (Hp / 64) {
        PFM_EW_PAIR_CASE(2)
        PFM_EW_PAIR_CASE(3)
        PFM_EW_PAIR_CASE(4)
        PFM_EW_PAIR_CASE(5)
        default: return false;
    }
#undef PFM_EW_PAIR_CASE
#undef PFM_EW_PAIR_LAUNCH
}


// v[row][f] = lrelu( W3[f] . X[row] + jb3[jet][f] ) * mask[row]   (epic.py:386-389); optional fused state update
struct HeadArgs {
    const float *X, *blob, *jb, *mask, *base, *dt;
    const int *rowsrc, *rowjet, *m_dev;  // compacted rows (then mask == nullptr) or nullptr
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
    if (a.m_dev) a.M = *a.m_dev;
    if (blockIdx.x * 16 >= a.M) return;
    const int rowc = min(row, a.M - 1);
    const float* xp = a.X + (int64_t)rowc * Hp + 4 * pl;
    f32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(xp + 64 * i);
    const float m = a.mask ? a.mask[rowc] : 1.0f;
    const int jet = a.rowjet ? a.rowjet[rowc] : rowc / a.N;
    const int64_t orow = a.rowsrc ? a.rowsrc[rowc] : row;
#pragma unroll 1
    for (int f = 0; f < a.F; ++f) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            d += hsum4(v[i] * *reinterpret_cast<const f32x4*>(a.blob + a.W3 + (int64_t)f * Hp + 4 * pl + 64 * i));
        d = row_sum16(d) + a.jb[(int64_t)jet * a.jb_stride + f];
        d = lrelu(d, a.slope) * m;
        if (pl == (f & 15) && row < a.M) {
            const int64_t e = orow * a.F + f;
            if (a.base) a.dst[e] = __fadd_rn(a.base[e], __fmul_rn(__fmul_rn(a.coef, a.dt[0]), d));
            else a.dst[e] = d;
        }
    }
}

// Workspace (floats).  Inference: one P row set, one Q, one X / L1.  Train: every stage keeps its own copies
// (stage 0 = stem, stage l+1 = layer l): P_s, Q_s (pool of X_s), X_s, L1_l -- what the backward re-reads.
struct Ws {
    int64_t P, pstride, Q, qstride, SJB, JB, G, step, X1, X, xstride, L1, lstride, imaps, part, part_floats, total;  // G: two [B][128] latent states (inference); step: tf_common.h step_args_kernel; imaps: int32 cnt[B] off[B+1] m[1] rowsrc[M] rowjet[M]
};

__host__ inline Ws make_ws(const pfm_ew_desc& d, int n_jets, bool train) {
    Ws w;
    const int64_t M = (int64_t)n_jets * d.n_points, Hp = d.hidden_pad;
    const int stages = train ? d.layers + 1 : 1;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    w.pstride = train ? round64((int64_t)n_jets * (256 + Hp)) : 0;
    w.P = take((int64_t)n_jets * (256 + Hp)); o += w.pstride * (stages - 1);
    w.qstride = train ? round64((int64_t)n_jets * 2 * Hp) : 0;
    w.Q = take((int64_t)n_jets * 2 * Hp); o += w.qstride * (stages - 1);
    w.SJB = take((int64_t)n_jets * (2 * Hp + 128));
    w.JB = take((int64_t)n_jets * 2 * Hp);
    w.G = take(train ? 0 : 2 * (int64_t)n_jets * 128);
    w.step = take(STEP_SLOT_FLOATS);
    w.X1 = take(M * Hp);
    w.xstride = train ? round64(M * Hp) : 0;
    w.X = take(M * Hp); o += w.xstride * (stages - 1);
    w.lstride = train ? round64(M * Hp) : 0;
    w.L1 = take(M * Hp); o += w.lstride * (stages > 2 ? stages - 2 : 0);
    w.imaps = take(row_maps_ints(n_jets, M));
    w.part_floats = n_jets <= 1024 ? 8 * (int64_t)n_jets * (2 * Hp + 128) : 0;
    w.part = take(w.part_floats);
    w.total = o;
    return w;
}

int validate(const pfm_ew_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_EW_ABI_VERSION) return set_err(PFM_E_BADARG, "epicw desc.abi_version mismatch");
    if (d->hidden < 1 || d->hidden_pad != (d->hidden + 63) / 64 * 64 || d->hidden_pad > 512)
        return set_err(PFM_E_BADARG, "hidden_pad must be hidden rounded up to a multiple of 64, at most 512");
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
    // compacted rows (inference with a mask): only valid particles are rows; nullptr = dense rows
    const int *rowsrc = nullptr, *rowjet = nullptr, *off = nullptr, *m_dev = nullptr;
    float* part = nullptr;  // split-K partial sums of the per-jet GEMMs
    int64_t part_floats = 0;
    int temb_k = 0;  // PFM_EW_F_TEMB_GIVEN: floats between the elements of a time-embedding row in `t` (0: `t` holds times)
};

// out[Mrows][ldo] = epi(A (+A2) W^T + b / jb)
int linear(const Plan& p, int Mrows, const float* A, int lda, int K1, const float* A2, int lda2, int K, const pfm_ew_lin& lin,
           int NO, const float* jb, int64_t jb_stride, int jbN, const float* R, int ldr, float* out, int ldo, int act,
           const float* Y = nullptr, int ldy = 0) {
    LinArgs a;
    a.A = A; a.A2 = A2; a.lda = lda; a.lda2 = lda2; a.K1 = K1; a.blob = p.blob; a.jb = jb; a.R = R; a.Y = Y; a.ldy = ldy; a.out = out;
    a.rowjet = (jb && p.rowjet) ? p.rowjet : nullptr; a.m_dev = (jb && p.m_dev) ? p.m_dev : nullptr;  // particle-row GEMMs are the ones with a jet bias
    a.blob_floats = p.d->blob_floats; a.W = lin.W; a.b = lin.b; a.gamma = -1; a.beta = -1; a.jb_stride = jb_stride;
    a.ldr = ldr; a.ldo = ldo; a.M = Mrows; a.K = K; a.NO = NO; a.N = jbN; a.act = act;
    a.slope = p.d->neg_slope; a.eps = 0.f;
    // per-jet GEMMs (a few hundred rows) are latency chains over K: split K over workgroups, reduce in a second launch
    a.part = nullptr; a.ksplit = 1;
    if (!jb && Mrows <= 1024 && p.part) {
        const int nstep = K / 64;
        int ks = nstep < 8 ? nstep : 8;
        while (nstep % ks) --ks;
        if (ks > 1 && (int64_t)ks * Mrows * NO <= p.part_floats) { a.part = p.part; a.ksplit = ks; }
    }
    // particle-row GEMMs whose width is not a multiple of 128 outputs (H = 300 -> 320: 128 + 128 + 64): five equal 64-output workgroups
    // per row tile instead of two full ones and a half one -- more, smaller workgroups per CU (cfg 5: 50 -> 42 us per launch)
    if (jb && NO % BN != 0) a.bn = 64;
    launch_linear_kernel(a, 0, (p.d->flags & PFM_EW_F_F16X3) ? 1 : ((p.d->flags & PFM_EW_F_BF16) ? 2 : 0), num_cus(), p.s);
    int rc = check_hip(hipGetLastError(), "tf_linear_kernel launch (epicw)");
    if (rc || a.ksplit == 1) return rc;
    const int64_t n4 = (int64_t)Mrows * (NO / 4);
    hipLaunchKernelGGL(tf_splitk_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, p.s, a);
    return check_hip(hipGetLastError(), "tf_splitk_kernel launch");
}

#define PFM_TRY(x) do { if ((rc = (x))) return rc; } while (0)

int run_nfe(const Plan& p, const float* t, int t_stride, const float* x, const float* cond, const float* mask, const HeadArgs& head_tpl) {
    const pfm_ew_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    const int Hp = d.hidden_pad, ldp = 256 + Hp, B = p.n_jets, N = d.n_points;
    const int64_t sjbs = 2 * Hp + 128;
    float *SJB = ws + w.SJB, *JB = ws + w.JB, *X1 = ws + w.X1;
    auto Pst = [&](int s) { return ws + w.P + w.pstride * s; };
    auto Qst = [&](int s) { return ws + w.Q + w.qstride * s; };
    auto Xst = [&](int s) { return ws + w.X + w.xstride * s; };
    int rc;
    const int stages = w.pstride ? d.layers + 1 : 1;
    hipLaunchKernelGGL(ew_prep_kernel, dim3(B, stages), dim3(256), 0, p.s, p.blob, d.freqs, t,
                       p.temb_k ? (t_stride ? d.t_dim : 0) : t_stride, cond, Pst(0), d.t_dim, d.cond_global, ldp, w.pstride,
                       (d.flags & PFM_EW_F_TEMB_SINCOS) ? 1 : 0, p.temb_k);
    PFM_TRY(check_hip(hipGetLastError(), "ew_prep_kernel launch"));
    PFM_TRY(linear(p, B, Pst(0), ldp, 256, nullptr, 0, 256, d.sjb, 2 * Hp + 128, nullptr, 0, 1, nullptr, 0, SJB, (int)sjbs, 0));
    // stem: fc_l1 (F columns on the VALU), fc_l2 (residual inside the activation, epic.py:327-328)
    hipLaunchKernelGGL(tf_embed_kernel, dim3((p.M + 31) / 32), dim3(256), 0, p.s, p.blob, d.l1x, x, (const float*)SJB, sjbs, X1, p.M, N,
                       d.features, Hp, d.neg_slope, p.rowsrc, p.rowjet, p.m_dev);
    PFM_TRY(check_hip(hipGetLastError(), "tf_embed_kernel launch (epicw)"));
    PFM_TRY(linear(p, p.M, X1, Hp, Hp, nullptr, 0, Hp, d.l2, Hp, SJB + Hp, sjbs, N, X1, Hp, Xst(0), Hp, 2));
    auto pool = [&](int s) {
        if (p.off)
            hipLaunchKernelGGL(ew_pool_compact_kernel, dim3(B), dim3(256), 0, p.s, (const float*)Xst(s), p.off, Qst(s), Hp, d.sum_scale);
        else
            hipLaunchKernelGGL(ew_pool_kernel, dim3(B), dim3(256), 0, p.s, (const float*)Xst(s), mask, Qst(s), N, Hp, d.sum_scale);
        return check_hip(hipGetLastError(), "ew_pool_kernel launch");
    };
    // inference: the per-jet chain of a stage on 16-jet MFMA tiles (ew_g1_kernel, ew_g2jb_kernel); training keeps the row GEMMs, whose
    // per-stage P / Q the backward re-reads
#ifdef PFM_EW_AB_ROWCHAIN  // (diagnostic build: the per-jet chain as row GEMMs over the jets, the training path's launches)
    const bool fused_chain = false;
#else
    const bool fused_chain = !w.pstride;
#endif
    float* const G = ws + w.G;
    const int nl = (d.latent + 15) / 16, groups = (B + 15) / 16;
    // fc_global1 / fc_global2 (+ the jet-bias rows) of one stage; stage -1 = the stem (fc_g1 / fc_g2: no latent input, no residual, no
    // jet-bias rows).  The latent state alternates between the two halves of G.
    auto chain = [&](int l) {
        SkArgs a;
        a.blob = p.blob; a.P = Pst(0); a.Q = Qst(0); a.JB = JB;
        a.B = B; a.Hp = Hp; a.ldp = ldp; a.n_tc = (d.t_dim + d.cond_global + 15) / 16; a.nl = nl; a.slope = d.neg_slope;
        const bool stem = l < 0;
        a.lin1 = stem ? d.sg1 : d.layer[l].g1;
        a.lin2 = stem ? d.sg2 : d.layer[l].g2;
        a.jb = stem ? d.sjb : d.layer[l].jb;
        a.Gin = stem ? nullptr : G + (int64_t)(l & 1) * B * 128;
        a.Gout = G + (int64_t)(stem ? 0 : (l + 1) & 1) * B * 128;
        a.residual = stem ? 0 : 1;
        a.do_jb = stem ? 0 : 1;
        hipLaunchKernelGGL(ew_g1_kernel, dim3(groups, Hp / 64), dim3(512), 0, p.s, a);
        const dim3 g2(groups, stem ? 1 : (2 * Hp / 16 + 7) / 8);
        if (nl == 1) hipLaunchKernelGGL(ew_g2jb_kernel<1>, g2, dim3(256), 0, p.s, a);
        else if (nl == 2) hipLaunchKernelGGL(ew_g2jb_kernel<2>, g2, dim3(256), 0, p.s, a);
        else if (nl <= 4) hipLaunchKernelGGL(ew_g2jb_kernel<4>, g2, dim3(256), 0, p.s, a);
        else hipLaunchKernelGGL(ew_g2jb_kernel<8>, g2, dim3(256), 0, p.s, a);
        return check_hip(hipGetLastError(), "ew_g1_kernel / ew_g2jb_kernel launch");
    };
    if (fused_chain && d.layers > 0) {
        PFM_TRY(pool(0));
        PFM_TRY(chain(-1));
    } else {
        PFM_TRY(pool(0));
        PFM_TRY(linear(p, B, Pst(0), ldp, 256, Qst(0), 2 * Hp, 256 + 2 * Hp, d.sg1, Hp, nullptr, 0, 1, nullptr, 0, Pst(0) + 256, ldp, 1));
        PFM_TRY(linear(p, B, Pst(0), ldp, ldp, nullptr, 0, ldp, d.sg2, 128, nullptr, 0, 1, nullptr, 0, Pst(0) + 128, ldp, 1));
    }
    for (int l = 0; l < d.layers; ++l) {
        const pfm_ew_layer& L = d.layer[l];
        float *Pin = Pst(l), *Pout = Pst(l + 1);  // the same row set at inference
        if (fused_chain) {
#ifndef PFM_EW_AB_NOCHAIN  // (timing-only diagnostic build, tests/diag/build_ew_ab.sh: what the sampler costs without its per-jet chains)
            if (l) PFM_TRY(pool(l));
            PFM_TRY(chain(l));
#endif
        } else {
            if (l) PFM_TRY(pool(l));
            PFM_TRY(linear(p, B, Pin, ldp, 256, Qst(l), 2 * Hp, 256 + 2 * Hp, L.g1, Hp, nullptr, 0, 1, nullptr, 0, Pout + 256, ldp, 1));
            PFM_TRY(linear(p, B, Pout, ldp, ldp, nullptr, 0, ldp, L.g2, 128, nullptr, 0, 1, Pin + 128, ldp, Pout + 128, ldp, 2));
            PFM_TRY(linear(p, B, Pout, ldp, 256, nullptr, 0, 256, L.jb, 2 * Hp, nullptr, 0, 1, nullptr, 0, JB, 2 * Hp, 0));
        }
        float* L1 = ws + w.L1 + w.lstride * l;
        // inference, fp32 operands: both local Linears of a 32-row tile in one launch (ew_pair_kernel); training keeps the two launches
        // (the backward re-reads the hidden rows), as do the split-fp16 / bf16 flavours
        if (!w.lstride && !(d.flags & (PFM_EW_F_F16X3 | PFM_EW_F_BF16))) {
            PairArgs pa;
            pa.blob = p.blob; pa.X = Xst(l); pa.out = Xst(l + 1); pa.jb = JB; pa.rowjet = p.rowjet; pa.m_dev = p.m_dev;
            pa.blob_floats = d.blob_floats; pa.W1 = L.l1.W; pa.W2 = L.l2.W; pa.jb_stride = 2 * Hp; pa.ldx = Hp; pa.M = p.M; pa.N = N;
            pa.slope = d.neg_slope;
            if (launch_pair(pa, Hp, num_cus(), p.s)) {
                PFM_TRY(check_hip(hipGetLastError(), "ew_pair_kernel launch"));
                continue;
            }
        }
        PFM_TRY(linear(p, p.M, Xst(l), Hp, Hp, nullptr, 0, Hp, L.l1, Hp, JB, 2 * Hp, N, nullptr, 0, L1, Hp, 1));
        PFM_TRY(linear(p, p.M, L1, Hp, Hp, nullptr, 0, Hp, L.l2, Hp, JB + Hp, 2 * Hp, N, Xst(l), Hp, Xst(l + 1), Hp, 2));
    }
    HeadArgs h = head_tpl;
    h.X = Xst(w.xstride ? d.layers : 0); h.blob = p.blob; h.jb = SJB + 2 * Hp; h.jb_stride = sjbs; h.W3 = d.l3;
    h.mask = p.rowsrc ? nullptr : mask; h.rowsrc = p.rowsrc; h.rowjet = p.rowjet; h.m_dev = p.m_dev;
    h.M = p.M; h.N = N; h.F = d.features; h.slope = d.neg_slope;
    const dim3 g((p.M + 15) / 16), bl(256);
    if (p.rowsrc) {  // rows the compacted evaluation never touches
        const int64_t n = (int64_t)p.M * d.features;
        hipLaunchKernelGGL(ew_fill_masked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, mask,
                           reinterpret_cast<const int*>(ws + w.imaps), h.base, h.dst, (int64_t)p.M, N, d.features);
        PFM_TRY(check_hip(hipGetLastError(), "ew_fill_masked_kernel launch"));
    }
    switch (Hp / 64) {
        case 2: hipLaunchKernelGGL(ew_head_kernel<2>, g, bl, 0, p.s, h); break;
        case 1: hipLaunchKernelGGL(ew_head_kernel<1>, g, bl, 0, p.s, h); break;
        case 3: hipLaunchKernelGGL(ew_head_kernel<3>, g, bl, 0, p.s, h); break;
        case 4: hipLaunchKernelGGL(ew_head_kernel<4>, g, bl, 0, p.s, h); break;
        case 5: hipLaunchKernelGGL(ew_head_kernel<5>, g, bl, 0, p.s, h); break;
        case 6: hipLaunchKernelGGL(ew_head_kernel<6>, g, bl, 0, p.s, h); break;
        case 7: hipLaunchKernelGGL(ew_head_kernel<7>, g, bl, 0, p.s, h); break;
        default: hipLaunchKernelGGL(ew_head_kernel<8>, g, bl, 0, p.s, h); break;
    }
    return check_hip(hipGetLastError(), "ew_head_kernel launch");
}


// ---- backward pieces -------------------------------------------------------------------------------
// out[r][c] = (a[r][c] + b[r][c]) * lrelu'(y[r][c])   (b, y optional; row strides given)
__global__ __launch_bounds__(256) void ew_actbwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                        const float* __restrict__ y, int ldy, float* __restrict__ out, int ldo,
                                                        int64_t rows, int cols, float slope) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    float v = a[r * lda + c];
    if (b) v += b[r * ldb + c];
    if (y) v *= y[r * ldy + c] > 0.f ? 1.f : slope;
    out[r * ldo + c] = v;
}

// dpre[row][0..16) = crit'(v - u) w_jet gscale * mask * lrelu'(v)   (v = lrelu(pre) * mask: same sign as pre on valid rows)
// crit 0: 2 (v - u); 1 (huber, delta 1): clamp(v - u, -1, 1); jet_w (or NULL = 1): per-jet loss weight, N = rows per jet
__global__ __launch_bounds__(256) void ew_head_bwd_kernel(const float* __restrict__ v, const float* __restrict__ u,
                                                          const float* __restrict__ mask, const float* __restrict__ gscale,
                                                          float* __restrict__ dpre, int64_t M, int F, float slope, int crit,
                                                          const float* __restrict__ jet_w, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * 16) return;
    const int64_t row = i >> 4;
    const int f = (int)(i & 15);
    float d = 0.f;
    if (f < F) {
        const float vv = v[row * F + f], df = vv - u[row * F + f];
        const float dl = crit ? fminf(fmaxf(df, -1.0f), 1.0f) : 2.0f * df;
        d = dl * (jet_w ? jet_w[row / N] : 1.0f) * gscale[0] * (mask ? mask[row] : 1.0f) * (vv > 0.f ? 1.f : slope);
    }
    dpre[i] = d;
}

// K <- r0 * (x - K / r1): the probability-flow right-hand side of a noise-predicting network, -0.5 beta (x - eps_theta / noise_rate)
// (ode_wrapper.forward for loss_type="diffusion", flow_matching_module.py:62-69); rhs = (-0.5 beta, noise_rate) of this stage time
__global__ __launch_bounds__(256) void ew_diffusion_rhs_kernel(float* __restrict__ K, const float* __restrict__ x,
                                                               const float* __restrict__ rhs, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) K[i] = __fmul_rn(rhs[0], __fsub_rn(x[i], __fdiv_rn(K[i], rhs[1])));
}

// dX[row][c] += mask[row] * ( dmean[jet][c] / n_jet + dsum[jet][c] * scale )
__global__ __launch_bounds__(256) void ew_pool_bwd_kernel(const float* __restrict__ dQ, int ldq, const float* __restrict__ mask,
                                                          float* __restrict__ dX, int N, int Hp, float scale) {
    __shared__ float cnt[4];
    const int tid = threadIdx.x, jet = blockIdx.x;
    float n = 0.f;
    for (int r = tid; r < N; r += 256) n += mask ? mask[(int64_t)jet * N + r] : 1.0f;
    n = wave_sum(n);
    if ((tid & 63) == 0) cnt[tid >> 6] = n;
    __syncthreads();
    const float inv = 1.0f / ((cnt[0] + cnt[1]) + (cnt[2] + cnt[3]));
    const int nc4 = Hp >> 2;
    const float* dq = dQ + (int64_t)jet * ldq;
    for (int idx = tid; idx < N * nc4; idx += 256) {
        const int r = idx / nc4, c4 = idx - r * nc4;
        const int64_t row = (int64_t)jet * N + r;
        const float m = mask ? mask[row] : 1.0f;
        if (m != 0.f) {
            const f32x4 g = (*reinterpret_cast<const f32x4*>(dq + 4 * c4) * inv + *reinterpret_cast<const f32x4*>(dq + Hp + 4 * c4) * scale) * m;
            *reinterpret_cast<f32x4*>(dX + row * Hp + 4 * c4) += g;
        }
    }
}

__global__ __launch_bounds__(256) void ew_temb_acc_kernel(const float* __restrict__ dP, int ld, float* __restrict__ acc, int64_t n, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] += dP[(i / T) * ld + i % T];
}

struct Bs {
    int64_t dX, dZ, dT, dpre, DJB, DSJB, dPj, dP2, dP1, dG, dZg2, dZg1, zeros, dwpart, dtemb, total;
};

Bs make_bs(const pfm_ew_desc& d, int n_jets) {
    Bs b;
    const int64_t M = (int64_t)n_jets * d.n_points, Hp = d.hidden_pad, B = n_jets;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round64(n); return at; };
    b.dX = take(M * Hp); b.dZ = take(M * Hp); b.dT = take(M * Hp); b.dpre = take(M * 16);
    b.DJB = take(B * 2 * Hp); b.DSJB = take(B * (2 * Hp + 128)); b.dPj = take(B * 256);
    b.dP2 = take(B * (256 + Hp)); b.dP1 = take(B * (256 + 2 * Hp)); b.dG = take(B * 128);
    b.dZg2 = take(B * 128); b.dZg1 = take(B * Hp); b.zeros = take(B * Hp);
    b.dwpart = take((int64_t)DW_MAX_PARTS * 16384);
    b.dtemb = take((d.flags & PFM_EW_F_TEMB_GIVEN) ? B * d.t_dim : 0);
    b.total = o;
    return b;
}

struct Bwd {
    Plan p;
    float *gblob, *sc;
    Bs b;

    // column sums of Z[rows][NO]: per group of `group` rows into jet_out (stride), and/or over everything into gblob[gb]
    int colsum(const float* Z, int ldz, int NO, int64_t rows, int group, const float* X, int F, float* jet_out, int64_t jet_stride,
               int64_t gb) const {
        ColsumArgs a;
        a.Z = Z; a.X = X; a.jet_out = jet_out; a.gblob = gblob; a.gb = gb; a.jet_stride = jet_stride;
        a.ldz = ldz; a.NO = NO; a.N = group; a.F = X ? 16 : 1; a.rows = rows;
        const int ny = X ? F : 1, ngrp = (int)((rows + group - 1) / group);
        a.part = gb >= 0 ? sc + b.dwpart : nullptr;  // (free between two dW launches; every launch on p.s: stream order)
        hipLaunchKernelGGL(tf_colsum_kernel, dim3((unsigned)ngrp, ny, (NO + 767) / 768), dim3(256), 0, p.s, a);
        if (gb >= 0) launch_ordered_sum(p.s, a.part, ngrp, (int64_t)ny * NO, ny * NO, gblob + gb, ny * NO, nullptr);
        return check_hip(hipGetLastError(), "tf_colsum_kernel launch (epicw)");
    }
    // gblob[W] += Z^T [A | A2]
    int dw(int Mrows, const float* Z, int ldz, int NO, const float* A, int lda, int K1, const float* A2, int lda2, int K, int64_t gW) const {
        DwArgs a;
        a.Z = Z; a.A = A; a.A2 = A2; a.stats = nullptr; a.blob = p.blob; a.part = sc + b.dwpart; a.gamma = -1; a.beta = -1;
        a.ldz = ldz; a.lda = lda; a.lda2 = lda2; a.K1 = K1; a.M = Mrows; a.NO = NO; a.K = K;
        a.row_tiles = (Mrows + BM - 1) / BM;
        const int tiles = ((NO + 127) / 128) * ((K + 127) / 128);
        const int ns = dw_splits(a.row_tiles, tiles, num_cus());
        a.nsplit = ns;
        int rc;
        hipLaunchKernelGGL(tf_dw_kernel, dim3(tiles * ns), dim3(LT), 2 * 64 * DWS * sizeof(float), p.s, a);
        if ((rc = check_hip(hipGetLastError(), "tf_dw_kernel launch (epicw)"))) return rc;
        launch_dw_reduce(p.s, a.part, gblob, gW, NO, K, tiles, ns);
        return check_hip(hipGetLastError(), "tf_dw_reduce_kernel launch (epicw)");
    }
    // out[Mrows][K] = (Z W (+R)) (* lrelu'(Y))
    int dx(int Mrows, const float* Z, int ldz, int NO, const pfm_ew_lin& lin, int K, const float* R, int ldr, const float* Y, int ldy,
           float* out, int ldo) const {
        pfm_ew_lin t = lin;
        t.W = lin.WT;
        t.b = -1;
        return linear(p, Mrows, Z, ldz, NO, nullptr, 0, NO, t, K, nullptr, 0, 1, R, ldr, out, ldo, Y ? 3 : 0, Y, ldy);
    }
    // PFM_EW_F_TEMB_GIVEN: dtemb[jet][:T] += dP[jet][:T] (the time columns lead every P-row Linear's input)
    int temb_acc(const float* dP, int ld) const {
        if (!(p.d->flags & PFM_EW_F_TEMB_GIVEN)) return 0;
        const int64_t n = (int64_t)p.n_jets * p.d->t_dim;
        hipLaunchKernelGGL(ew_temb_acc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, dP, ld, sc + b.dtemb, n, p.d->t_dim);
        return check_hip(hipGetLastError(), "ew_temb_acc_kernel launch");
    }
    int actbwd(const float* a, int lda, const float* bb, int ldb, const float* y, int ldy, float* out, int ldo, int64_t rows, int cols) const {
        const int64_t n = rows * cols;
        hipLaunchKernelGGL(ew_actbwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, a, lda, bb, ldb, y, ldy, out, ldo, rows,
                           cols, p.d->neg_slope);
        return check_hip(hipGetLastError(), "ew_actbwd_kernel launch");
    }
};

int run_backward(const Bwd& W, const float* mask, const float* y, const float* u, const float* v, const float* gscale,
                 int crit = 0, const float* jet_w = nullptr, float* dy = nullptr) {
    const Plan& p = W.p;
    const pfm_ew_desc& d = *p.d;
    const Ws& w = p.w;
    float* ws = p.ws;
    float* sc = W.sc;
    const Bs& b = W.b;
    const int Hp = d.hidden_pad, ldp = 256 + Hp, B = p.n_jets, N = d.n_points, F = d.features, M = p.M;
    const int64_t sjbs = 2 * Hp + 128;
    float *dX = sc + b.dX, *dZ = sc + b.dZ, *dT = sc + b.dT, *dpre = sc + b.dpre, *DJB = sc + b.DJB, *DSJB = sc + b.DSJB;
    float *dPj = sc + b.dPj, *dP2 = sc + b.dP2, *dP1 = sc + b.dP1, *dG = sc + b.dG, *dZg2 = sc + b.dZg2, *dZg1 = sc + b.dZg1;
    float* zeros = sc + b.zeros;
    auto Pst = [&](int s) { return ws + w.P + w.pstride * s; };
    auto Qst = [&](int s) { return ws + w.Q + w.qstride * s; };
    auto Xst = [&](int s) { return ws + w.X + w.xstride * s; };
    float* X1 = ws + w.X1;
    int rc;
    PFM_TRY(check_hip(hipMemsetAsync(zeros, 0, (size_t)B * Hp * sizeof(float), p.s), "memset zeros"));
    PFM_TRY(check_hip(hipMemsetAsync(dG, 0, (size_t)B * 128 * sizeof(float), p.s), "memset dG"));
    // ---- head: v = lrelu(W3 X_L + jb3) mask ----
    hipLaunchKernelGGL(ew_head_bwd_kernel, dim3((unsigned)(((int64_t)M * 16 + 255) / 256)), dim3(256), 0, p.s, v, u, mask, gscale, dpre,
                       (int64_t)M, F, d.neg_slope, crit, jet_w, N);
    PFM_TRY(check_hip(hipGetLastError(), "ew_head_bwd_kernel launch"));
    const float* XL = Xst(d.layers);
    PFM_TRY(W.colsum(XL, Hp, Hp, M, N, dpre, F, nullptr, 0, d.l3));                       // d W3[f][k]
    PFM_TRY(W.colsum(dpre, 16, 16, M, N, nullptr, 0, DSJB + 2 * Hp, sjbs, -1));            // d jb3 per jet (16 of its 128 columns)
    hipLaunchKernelGGL(tf_embed_kernel, dim3((M + 31) / 32), dim3(256), 0, p.s, p.blob, d.l3, (const float*)dpre, (const float*)zeros,
                       (int64_t)Hp, dX, M, N, 16, Hp, 1.0f);                                // dX_L = dpre W3
    PFM_TRY(check_hip(hipGetLastError(), "tf_embed_kernel launch (head backward)"));
    // ---- layers, last to first ----
    for (int l = d.layers - 1; l >= 0; --l) {
        const pfm_ew_layer& L = d.layer[l];
        const float *Xin = Xst(l), *Xout = Xst(l + 1), *L1 = ws + w.L1 + w.lstride * l;
        const float *Pin = Pst(l), *Pout = Pst(l + 1);
        PFM_TRY(W.actbwd(dX, Hp, nullptr, 0, Xout, Hp, dZ, Hp, M, Hp));                                      // dZ2
        PFM_TRY(W.colsum(dZ, Hp, Hp, M, N, nullptr, 0, DJB + Hp, 2 * Hp, -1));
        PFM_TRY(W.dw(M, dZ, Hp, Hp, L1, Hp, Hp, nullptr, 0, Hp, L.l2.W));
        PFM_TRY(W.dx(M, dZ, Hp, Hp, L.l2, Hp, nullptr, 0, L1, Hp, dT, Hp));                                   // dZ1
        PFM_TRY(W.colsum(dT, Hp, Hp, M, N, nullptr, 0, DJB, 2 * Hp, -1));
        PFM_TRY(W.dw(M, dT, Hp, Hp, Xin, Hp, Hp, nullptr, 0, Hp, L.l1.W));
        PFM_TRY(W.dx(M, dT, Hp, Hp, L.l1, Hp, dZ, Hp, nullptr, 0, dX, Hp));                                   // dX_l = dZ2 + dZ1 W1
        // jet-bias GEMM
        PFM_TRY(W.dw(B, DJB, 2 * Hp, 2 * Hp, Pout, ldp, 256, nullptr, 0, 256, L.jb.W));
        PFM_TRY(W.colsum(DJB, 2 * Hp, 2 * Hp, B, 16, nullptr, 0, nullptr, 0, L.jb.b));
        PFM_TRY(W.dx(B, DJB, 2 * Hp, 2 * Hp, L.jb, 256, nullptr, 0, nullptr, 0, dPj, 256));
        PFM_TRY(W.temb_acc(dPj, 256));
        // fc_global2: g_new = lrelu(W [P | g1] + b + g_old)
        PFM_TRY(W.actbwd(dG, 128, dPj + 128, 256, Pout + 128, ldp, dZg2, 128, B, 128));
        PFM_TRY(W.dw(B, dZg2, 128, 128, Pout, ldp, ldp, nullptr, 0, ldp, L.g2.W));
        PFM_TRY(W.colsum(dZg2, 128, 128, B, 16, nullptr, 0, nullptr, 0, L.g2.b));
        PFM_TRY(W.dx(B, dZg2, 128, 128, L.g2, ldp, nullptr, 0, nullptr, 0, dP2, ldp));
        PFM_TRY(W.temb_acc(dP2, ldp));
        // fc_global1: g1 = lrelu(W [P256 | Q] + b)
        PFM_TRY(W.actbwd(dP2 + 256, ldp, nullptr, 0, Pout + 256, ldp, dZg1, Hp, B, Hp));
        PFM_TRY(W.dw(B, dZg1, Hp, Hp, Pin, ldp, 256, Qst(l), 2 * Hp, 256 + 2 * Hp, L.g1.W));
        PFM_TRY(W.colsum(dZg1, Hp, Hp, B, 16, nullptr, 0, nullptr, 0, L.g1.b));
        PFM_TRY(W.dx(B, dZg1, Hp, Hp, L.g1, 256 + 2 * Hp, nullptr, 0, nullptr, 0, dP1, 256 + 2 * Hp));
        PFM_TRY(W.temb_acc(dP1, 256 + 2 * Hp));
        PFM_TRY(W.actbwd(dZg2, 128, dP1 + 128, 256 + 2 * Hp, nullptr, 0, dG, 128, B, 128));                   // d g_old
        hipLaunchKernelGGL(ew_pool_bwd_kernel, dim3(B), dim3(256), 0, p.s, (const float*)(dP1 + 256), 256 + 2 * Hp, mask, dX, N, Hp,
                           d.sum_scale);
        PFM_TRY(check_hip(hipGetLastError(), "ew_pool_bwd_kernel launch"));
    }
    // ---- stem globals: g_0 = lrelu(fc_g2 [P | g1]), g1 = lrelu(fc_g1 [P256 | Q_0]) ----
    {
        const float* P0 = Pst(0);
        PFM_TRY(W.actbwd(dG, 128, nullptr, 0, P0 + 128, ldp, dZg2, 128, B, 128));
        PFM_TRY(W.dw(B, dZg2, 128, 128, P0, ldp, ldp, nullptr, 0, ldp, d.sg2.W));
        PFM_TRY(W.colsum(dZg2, 128, 128, B, 16, nullptr, 0, nullptr, 0, d.sg2.b));
        PFM_TRY(W.dx(B, dZg2, 128, 128, d.sg2, ldp, nullptr, 0, nullptr, 0, dP2, ldp));
        PFM_TRY(W.temb_acc(dP2, ldp));
        PFM_TRY(W.actbwd(dP2 + 256, ldp, nullptr, 0, P0 + 256, ldp, dZg1, Hp, B, Hp));
        PFM_TRY(W.dw(B, dZg1, Hp, Hp, P0, ldp, 256, Qst(0), 2 * Hp, 256 + 2 * Hp, d.sg1.W));
        PFM_TRY(W.colsum(dZg1, Hp, Hp, B, 16, nullptr, 0, nullptr, 0, d.sg1.b));
        PFM_TRY(W.dx(B, dZg1, Hp, Hp, d.sg1, 256 + 2 * Hp, nullptr, 0, nullptr, 0, dP1, 256 + 2 * Hp));
        PFM_TRY(W.temb_acc(dP1, 256 + 2 * Hp));
        hipLaunchKernelGGL(ew_pool_bwd_kernel, dim3(B), dim3(256), 0, p.s, (const float*)(dP1 + 256), 256 + 2 * Hp, mask, dX, N, Hp,
                           d.sum_scale);
        PFM_TRY(check_hip(hipGetLastError(), "ew_pool_bwd_kernel launch"));
    }
    // ---- stem locals: X_0 = lrelu(fc_l2 X1 + jb + X1), X1 = lrelu(fc_l1 y + jb) ----
    PFM_TRY(W.actbwd(dX, Hp, nullptr, 0, Xst(0), Hp, dZ, Hp, M, Hp));
    PFM_TRY(W.colsum(dZ, Hp, Hp, M, N, nullptr, 0, DSJB + Hp, sjbs, -1));
    PFM_TRY(W.dw(M, dZ, Hp, Hp, X1, Hp, Hp, nullptr, 0, Hp, d.l2.W));
    PFM_TRY(W.dx(M, dZ, Hp, Hp, d.l2, Hp, dZ, Hp, X1, Hp, dT, Hp));                                           // (dZ W + dZ) lrelu'(X1)
    PFM_TRY(W.colsum(dT, Hp, Hp, M, N, nullptr, 0, DSJB, sjbs, -1));
    if (dy) {  // the gradient w.r.t. the particle input (a chain of flows, n_transforms > 1: the next flow's backward starts from it)
        hipLaunchKernelGGL(tf_dy_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, p.s, (const float*)dT, p.blob, d.l1x, dy, (int64_t)M, F, Hp);
        PFM_TRY(check_hip(hipGetLastError(), "tf_dy_kernel launch (epicw)"));
    }
    {
        ColsumArgs a;  // d fc_l1 particle columns [F][Hp] = sum_rows y[row][f] dZ1[row][:]
        a.Z = dT; a.X = y; a.jet_out = nullptr; a.gblob = W.gblob; a.gb = d.l1x; a.jet_stride = 0; a.ldz = Hp; a.NO = Hp; a.N = N; a.F = F; a.rows = 0;
        a.part = W.sc + W.b.dwpart;  // per-jet sums, added over the jets in jet order (no atomics)
        hipLaunchKernelGGL(tf_colsum_kernel, dim3(B, F, 1), dim3(256), 0, p.s, a);
        launch_ordered_sum(p.s, a.part, B, (int64_t)F * Hp, F * Hp, W.gblob + d.l1x, F * Hp, nullptr);
        PFM_TRY(check_hip(hipGetLastError(), "tf_colsum_kernel launch (fc_l1)"));
    }
    // static jet-bias GEMM (the 112 padding columns of the fc_l3 block of DSJB were never written: clear them first)
    PFM_TRY(W.dw(B, DSJB, (int)sjbs, 2 * Hp + 128, Pst(0), ldp, 256, nullptr, 0, 256, d.sjb.W));
    PFM_TRY(W.colsum(DSJB, (int)sjbs, 2 * Hp + 128, B, 16, nullptr, 0, nullptr, 0, d.sjb.b));
    if (d.flags & PFM_EW_F_TEMB_GIVEN) {  // the time columns of fc_l1 / fc_l2 / fc_l3 (nobody else needs d P of the static jet biases)
        PFM_TRY(W.dx(B, DSJB, (int)sjbs, 2 * Hp + 128, d.sjb, 256, nullptr, 0, nullptr, 0, dPj, 256));
        PFM_TRY(W.temb_acc(dPj, 256));
    }
    return 0;
}

int make_plan(Plan& p, const pfm_ew_desc* d, const float* blob, float* ws, int n_jets, bool train, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    p.d = d; p.blob = blob; p.ws = ws; p.n_jets = n_jets; p.M = n_jets * d->n_points; p.s = (hipStream_t)stream;
    p.w = make_ws(*d, n_jets, train);
    p.part = ws + p.w.part;
    p.part_floats = p.w.part_floats;
    p.temb_k = (d->flags & PFM_EW_F_TEMB_GIVEN) ? 1 : 0;  // rows [jet][T]; the samplers switch to their [T][evaluations] table
    return 0;
}

// Inference with a mask: evaluate only the valid particles (they are the only rows that can influence an output the
// reference does not multiply by zero).  Row maps are built once per call; the mask is constant over an ODE solve.
int setup_compaction(Plan& p, const float* mask) {
    const RowMaps m = build_row_maps(reinterpret_cast<int*>(p.ws + p.w.imaps), mask, p.n_jets, p.d->n_points, p.s);
    p.rowsrc = m.rowsrc; p.rowjet = m.rowjet; p.off = m.off; p.m_dev = m.m_dev;
    return check_hip(hipGetLastError(), "row compaction launch");
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
    if (mask && (rc = ew::setup_compaction(p, mask))) return rc;
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
    if (p.temb_k) p.temb_k = 2 * n_steps;  // t_eval = the embedding table [T][2 n_steps]: evaluation e starts at t_eval + e
    // (Rounds 1-2 ran a call as two half-batches on two side streams.  With the per-jet chain on 16-jet MFMA tiles the halves fall into
    // lockstep -- both in their chains, then both in their particle Linears -- and one stream with 64-output workgroups is faster:
    // tests/diag/ew_sweep.sh, 256 jets x 100 steps, two calls in flight: 400 ms per call against 480-536 ms split.)
    const int64_t n = (int64_t)p.M * d->features;
    float *xs = state, *xm = state + n;
    hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : nullptr, xs, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    if (mask && (rc = ew::setup_compaction(p, mask))) return rc;
    // evaluation `stage` of a step, its time and step size behind the given addresses
    auto eval = [&](int stage, const float* t, const float* h_dt) -> int {
        ew::HeadArgs h{};
        h.base = xs; h.dt = h_dt; h.coef = stage ? 1.0f : 0.5f; h.dst = stage ? xs : xm;
        return ew::run_nfe(p, t, 0, stage ? xm : xs, cond, mask, h);
    };
    int k = 0;
    // Graph replay of the step body (tf_common.h: step_args_kernel): a step is ~210 launches of 5-45 us and the host needs ~9 us to
    // enqueue each (tests/diag/ew_host_rate.py: 370 ms of enqueueing for a 100-step sample of 256 jets) -- the enqueueing thread would
    // set the pace.  Step 0 runs directly (lazy first-launch work stays outside the capture); the body is captured once with its three
    // scalars behind a slot and replayed.  Not for the legacy stream, short solves, or a caller-supplied embedding table (addressed
    // through t_eval itself).  PFM_EW_GRAPH=0: direct launches (diagnostics).
    static const bool graph_on = [] { const char* e = getenv("PFM_EW_GRAPH"); return !e || atoi(e) != 0; }();
    ParkedGraph* gs = (graph_on && n_steps > 3 && !p.temb_k && p.s != nullptr) ? park_graph(p.s) : nullptr;
    if (gs) {
        for (int stage = 0; stage < 2; ++stage)
            if ((rc = eval(stage, t_eval + stage, dt))) return rc;
        float* slot = p.ws + p.w.step;
        if ((rc = check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slot + 4), 1, 1, p.s), "step counter"))) return rc;
        if ((rc = check_hip(hipStreamBeginCapture(p.s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture"))) return rc;
        hipLaunchKernelGGL(step_args_kernel, dim3(1), dim3(1), 0, p.s, t_eval, dt, slot);
        rc = eval(0, slot, slot + 2);
        if (rc == 0) rc = eval(1, slot + 1, slot + 2);
        const hipError_t ce = hipStreamEndCapture(p.s, &gs->graph);  // always end the capture, also after a failed launch
        if (rc == 0) rc = check_hip(ce, "hipStreamEndCapture");
        if (rc == 0) rc = check_hip(hipGraphInstantiate(&gs->exec, gs->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        for (k = 1; rc == 0 && k < n_steps; ++k) rc = check_hip(hipGraphLaunch(gs->exec, p.s), "hipGraphLaunch");
        const hipError_t re = hipEventRecord(gs->done, p.s);  // behind the last launch: park_graph's retire waits for it
        if (rc == 0) rc = check_hip(re, "hipEventRecord (graph replay)");
        if (rc) return rc;
    }
    for (; k < n_steps; ++k)
        for (int stage = 0; stage < 2; ++stage)
            if ((rc = eval(stage, t_eval + 2 * k + stage, dt + k))) return rc;
    return check_hip(hipMemcpyAsync(x_out, xs, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

static int ew_sample_rk(const pfm_ew_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                        int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets,
                        int32_t premask, float* state, float* workspace, void* stream, const float* rhs) {
    ew::Plan p;
    int rc = ew::make_plan(p, d, blob, workspace, n_jets, false, stream);
    if (rc) return rc;
    if (const char* e = rk_tableau_error(tab)) return set_err(PFM_E_BADARG, e);
    if (n_jets <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !state || !workspace) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_steps < 0) return set_err(PFM_E_BADARG, "n_steps < 0");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    if (p.temb_k && rhs) return set_err(PFM_E_BADARG, "the diffusion right-hand side is indexed by the time grid: no PFM_EW_F_TEMB_GIVEN");
    if (p.temb_k) p.temb_k = n_steps * tab->stages;  // t_eval = the embedding table [T][n_steps * stages]
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf_premask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, z, premask ? mask : nullptr, state, n,
                       d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_premask_kernel launch"))) return rc;
    if (mask && (rc = ew::setup_compaction(p, mask))) return rc;
    rc = sample_rk_rows(*tab, t_eval, dt, n_steps, state, n, p.s, [&](const float* t, const float* x, float* v) {
        ew::HeadArgs h{};
        h.dst = v;
        int r = ew::run_nfe(p, t, 0, x, cond, mask, h);
        if (r || !rhs) return r;
        hipLaunchKernelGGL(ew::ew_diffusion_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, v, x,
                           rhs + 2 * (t - t_eval), n);
        return check_hip(hipGetLastError(), "ew_diffusion_rhs_kernel launch");
    });
    if (rc) return rc;
    if ((rc = check_hip(hipGetLastError(), "tf_rk_combine_kernel launch"))) return rc;
    return check_hip(hipMemcpyAsync(x_out, state, n * sizeof(float), hipMemcpyDeviceToDevice, p.s), "copy x_out");
}

int pfm_ew_sample_rk(const pfm_ew_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                     int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets,
                     int32_t premask, float* state, float* workspace, void* stream) {
    return ew_sample_rk(d, blob, tab, t_eval, dt, n_steps, z, cond, mask, x_out, n_jets, premask, state, workspace, stream, nullptr);
}

int pfm_ew_sample_rk_rhs(const pfm_ew_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                         int32_t n_steps, const float* z, const float* cond, const float* mask, float* x_out, int32_t n_jets,
                         int32_t premask, float* state, float* workspace, const float* rhs, void* stream) {
    if (!rhs) return set_err(PFM_E_BADARG, "rhs is NULL");
    return ew_sample_rk(d, blob, tab, t_eval, dt, n_steps, z, cond, mask, x_out, n_jets, premask, state, workspace, stream, rhs);
}

static int ew_loss_forward(const pfm_ew_desc* d, const float* blob, int32_t kind, float sigma, const float* t,
                           const float* x, const float* a, const float* b, const float* cond, const float* mask,
                           float* y_out, float* u_out, float* v_out, float* loss_sums, int32_t n_jets,
                           float* workspace, void* stream, int crit, const float* jet_w) {
    ew::Plan p;
    int rc = ew::make_plan(p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (kind < 0 || kind > 3) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM), 2 (droid) or 3 (diffusion)");
    if (!blob || !t || !x || !a || !y_out || !u_out || !v_out || !loss_sums || !workspace)
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if ((kind == 1 || kind == 3) && !b) return set_err(PFM_E_BADARG, "CFM needs eps (diffusion: the rates)");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    const int64_t n = (int64_t)p.M * d->features;
    hipLaunchKernelGGL(tf_yu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.s, kind, sigma, t, x, a, b, mask, y_out,
                       u_out, n, d->n_points * d->features, d->features);
    if ((rc = check_hip(hipGetLastError(), "tf_yu_kernel launch"))) return rc;
    ew::HeadArgs h{};
    h.dst = v_out;
    if ((rc = ew::run_nfe(p, t, 1, y_out, cond, mask, h))) return rc;
    hipLaunchKernelGGL(tf_loss_kernel, dim3(1), dim3(LOSS_T), 0, p.s, (const float*)v_out, (const float*)u_out, mask, loss_sums, n,
                       (int64_t)p.M, crit, jet_w, d->n_points * d->features);
    return check_hip(hipGetLastError(), "tf_loss_kernel launch");
}

int pfm_ew_fm_loss_forward(const pfm_ew_desc* d, const float* blob, int32_t kind, float sigma, const float* t,
                           const float* x, const float* a, const float* b, const float* cond, const float* mask,
                           float* y_out, float* u_out, float* v_out, float* loss_sums, int32_t n_jets,
                           float* workspace, void* stream) {
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    return ew_loss_forward(d, blob, kind, sigma, t, x, a, b, cond, mask, y_out, u_out, v_out, loss_sums, n_jets, workspace, stream, 0,
                           nullptr);
}

int pfm_ew_diffusion_loss_forward(const pfm_ew_desc* d, const float* blob, int32_t criterion, const float* rates,
                                  const float* jet_weight, const float* t, const float* x, const float* z, const float* cond,
                                  const float* mask, float* y_out, float* u_out, float* v_out, float* loss_sums,
                                  int32_t n_jets, float* workspace, void* stream) {
    if (criterion < 0 || criterion > 1) return set_err(PFM_E_BADARG, "criterion must be 0 (mse) or 1 (huber)");
    if (!rates || !jet_weight) return set_err(PFM_E_BADARG, "the diffusion loss needs the signal / noise rates and the jet weights");
    return ew_loss_forward(d, blob, 3, 0.f, t, x, z, rates, cond, mask, y_out, u_out, v_out, loss_sums, n_jets, workspace, stream,
                           criterion, jet_weight);
}

int64_t pfm_ew_backward_scratch_floats(const pfm_ew_desc* d, int32_t n_jets) {
    if (ew::validate(d)) return -1;
    return ew::make_bs(*d, n_jets < 1 ? 1 : n_jets).total;
}

static int ew_loss_backward(const pfm_ew_desc* d, const float* blob, const float* mask, const float* y, const float* u,
                            const float* v, const float* gscale, float* gblob, int32_t n_jets, float* workspace,
                            float* scratch, void* stream, int crit, const float* jet_w, float* dy = nullptr) {
    ew::Bwd W;
    int rc = ew::make_plan(W.p, d, blob, workspace, n_jets, true, stream);
    if (rc) return rc;
    if (n_jets <= 0) return 0;
    if (!blob || !y || !u || !v || !gscale || !gblob || !workspace || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->l2.WT < 0) return set_err(PFM_E_BADARG, "blob was packed without the transposed (backward) weight copies");
    W.gblob = gblob;
    W.sc = scratch;
    W.b = ew::make_bs(*d, n_jets);
    if ((rc = check_hip(hipMemsetAsync(scratch + W.b.DSJB, 0, (size_t)n_jets * (2 * d->hidden_pad + 128) * sizeof(float),
                                       (hipStream_t)stream), "memset DSJB")))
        return rc;
    if ((d->flags & PFM_EW_F_TEMB_GIVEN) &&
        (rc = check_hip(hipMemsetAsync(scratch + W.b.dtemb, 0, (size_t)n_jets * d->t_dim * sizeof(float), (hipStream_t)stream), "memset dtemb")))
        return rc;
    return ew::run_backward(W, mask, y, u, v, gscale, crit, jet_w, dy);
}

int pfm_ew_fm_loss_backward(const pfm_ew_desc* d, const float* blob, const float* mask, const float* y, const float* u,
                            const float* v, const float* gscale, float* gblob, int32_t n_jets, float* workspace,
                            float* scratch, void* stream) {
    return ew_loss_backward(d, blob, mask, y, u, v, gscale, gblob, n_jets, workspace, scratch, stream, 0, nullptr);
}

int pfm_ew_fm_loss_backward_dx(const pfm_ew_desc* d, const float* blob, const float* mask, const float* y, const float* u,
                               const float* v, const float* gscale, float* gblob, float* grad_y, int32_t n_jets, float* workspace,
                               float* scratch, void* stream) {
    if (!grad_y) return set_err(PFM_E_BADARG, "grad_y is NULL");
    return ew_loss_backward(d, blob, mask, y, u, v, gscale, gblob, n_jets, workspace, scratch, stream, 0, nullptr, grad_y);
}

int pfm_ew_backward_dtemb(const pfm_ew_desc* d, const float* scratch, int32_t n_jets, float* dtemb, void* stream) {
    int rc = ew::validate(d);
    if (rc) return rc;
    if (!(d->flags & PFM_EW_F_TEMB_GIVEN)) return set_err(PFM_E_BADARG, "pfm_ew_backward_dtemb: the descriptor has no PFM_EW_F_TEMB_GIVEN");
    if (n_jets <= 0) return 0;
    if (!scratch || !dtemb) return set_err(PFM_E_BADARG, "NULL device pointer");
    return check_hip(hipMemcpyAsync(dtemb, scratch + ew::make_bs(*d, n_jets).dtemb, (size_t)n_jets * d->t_dim * sizeof(float),
                                    hipMemcpyDeviceToDevice, (hipStream_t)stream), "copy dtemb");
}

int pfm_ew_diffusion_loss_backward(const pfm_ew_desc* d, const float* blob, int32_t criterion, const float* jet_weight,
                                   const float* mask, const float* y, const float* u, const float* v, const float* gscale,
                                   float* gblob, int32_t n_jets, float* workspace, float* scratch, void* stream) {
    if (criterion < 0 || criterion > 1) return set_err(PFM_E_BADARG, "criterion must be 0 (mse) or 1 (huber)");
    if (!jet_weight) return set_err(PFM_E_BADARG, "jet_weight is NULL");
    return ew_loss_backward(d, blob, mask, y, u, v, gscale, gblob, n_jets, workspace, scratch, stream, criterion, jet_weight);
}

}  // extern "C"
