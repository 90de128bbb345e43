// Backward of the EPiC vector field + flow-matching loss for ONE jet by ONE 512-thread workgroup.
//
// Autograd of particle_fm/models/components/epic.py:85-203, 304-391 under losses.py:75-76
// (loss = sum (v-u)^2 / sum(mask)), restated by hand.  Works from the activations that
// pfm_epic_fm_loss_forward saved (SavedLayout in epic_nfe.h).
//
// On-chip mapping: the running gradient w.r.t. the hidden state (N x 128) stays in LDS buffer G for
// the whole backward; Hb holds the gradient at the inner activation of the current layer.
//   dX products   dl1 = W2^T da2,  dh = W1^T da1 + da2        -> same MFMA scheme as the forward with the
//                 transposed weight fragments (MFMA_AT) in registers and the gradient rows from LDS
//   dW products   dW[o][i] = sum_p da[p][o] * act[p][i]       -> A = da^T from LDS (one ds_read_b32 per
//                 k-step of 4 particles), B = saved activation rows straight from global/L2
//                 (lane j reads act[p][8j..8j+7], i.e. output tile `it` holds input feature 8j+it),
//                 accumulators 32 VGPRs per wave, then one coalesced 256-byte atomic per register
//   per-jet GEMV parts (global MLP, folded t/cond/g columns): rank-1 updates, atomics over K-major rows
// The weight gradient goes to `gblob`, which has the offsets of the weight blob; the 128x128 blocks
// are stored in the accumulator-native order documented in pfm_hip.h (GRAD_D format).
#pragma once
#include "epic_nfe.h"

namespace pfm {

struct BCarve {
    int G, Hb;         // N*H each
    int da3;           // N*F  gradient at the head pre-activation
    int maskf;         // N
    int w3;            // F*H
    int vin;           // 352  [temb ; cond ; mean ; sum*s ; g] of the current stage
    int vin2;          // 208  [temb ; cond ; g1]
    int dbj1, dbj2;    // H each: column sums of da1 / da2
    int dag1;          // H
    int dP;            // H   gradient w.r.t. the raw pooled sum
    int dg;            // MAXL running gradient w.r.t. g
    int dag2;          // MAXL
    int dg1;           // H
    int misc;          // 16
    int total;
};

__host__ __device__ inline BCarve make_bcarve(int N, int F) {
    BCarve c;
    int o = 0;
    c.G = o; o += N * H;
    c.Hb = o; o += N * H;
    c.da3 = o; o += round4(N * F);
    c.maskf = o; o += round4(N);
    c.w3 = o; o += F * H;
    c.vin = o; o += VIN_FLOATS;
    c.vin2 = o; o += 208;
    c.dbj1 = o; o += H;
    c.dbj2 = o; o += H;
    c.dag1 = o; o += H;
    c.dP = o; o += H;
    c.dg = o; o += MAXL;
    c.dag2 = o; o += MAXL;
    c.dg1 = o; o += H;
    c.misc = o; o += 16;
    c.total = o;
    return c;
}

__device__ __forceinline__ float dlrelu(float y, float slope) { return y > 0.f ? 1.0f : slope; }  // sign(y) = sign(pre-act)
__device__ __forceinline__ f32x4 dlrelu4(f32x4 y, float s) {
    f32x4 r;
    r.x = dlrelu(y.x, s); r.y = dlrelu(y.y, s); r.z = dlrelu(y.z, s); r.w = dlrelu(y.w, s);
    return r;
}

__device__ __forceinline__ f32x4 colsum16(f32x4 v) { return row_sum16(v); }

// dX product: for every row p < n_rows, epi(p, oslot, acc) with
//   acc[r] = sum_k AT-block[16w + 4q + r][k] * src[p][k]        (a = MFMA_AT fragments of this wave)
template <typename Epi>
__device__ __forceinline__ void gemm_dx(const f32x4 (&a)[8], const float* __restrict__ src, int n_rows, Epi epi) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pl = lane & 15, q = lane >> 4, oslot = 4 * w + q;
    const int npairs = (n_rows + 2 * TILE - 1) / (2 * TILE);
    for (int pair = 0; pair < npairs; ++pair) {
        const int p0 = pair * 2 * TILE + pl, p1 = p0 + TILE;
        const int pc0 = min(p0, n_rows - 1), pc1 = min(p1, n_rows - 1);
        f32x4 b0[8], b1[8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            b0[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc0, 4 * kt + q));
            b1[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc1, 4 * kt + q));
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) { PFM_MFMA_PAIR(acc0, acc1, a[kt], b0[kt], b1[kt]); }
        if (p0 < n_rows) epi(p0, oslot, acc0);
        if (p1 < n_rows) epi(p1, oslot, acc1);
    }
}

// dW product: gblock (GRAD_D order) += da^T . act, K = particles.
//   da  : LDS buffer (N x 128, swizzled rows), only rows < n_rows are meaningful
//   act : global (N x 128 row-major) saved activation
__device__ __forceinline__ void gemm_dw(const float* __restrict__ da, const float* __restrict__ act, int n_rows,
                                        float* __restrict__ gblock) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jl = lane & 15, q = lane >> 4;
    f32x4 acc[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (n_rows + 3) >> 2;
    // A: lane (o_local = jl, k = q) reads da[4s + q][16w + jl]
    const int acol_slot = 4 * w + (jl >> 2), acol_r = jl & 3;
#pragma unroll 2
    for (int s = 0; s < nsteps; ++s) {
        const int p = 4 * s + q;
        const int pc = min(p, n_rows - 1);
        float av = da[lds_off(pc, acol_slot) + acol_r];
        if (p >= n_rows) av = 0.f;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(act + pc * H + 8 * jl);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(act + pc * H + 8 * jl + 4);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.w, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.x, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.y, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.z, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.w, acc[7], 0, 0, 0);
    }
    // GRAD_D: float ((w*8 + it)*4 + r)*64 + lane  holds dW[16w + 4(lane>>4) + r][8(lane&15) + it]
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        float* g = gblock + ((w * 8 + it) * 4) * 64 + lane;
        atomicAdd(g, acc[it].x);
        atomicAdd(g + 64, acc[it].y);
        atomicAdd(g + 128, acc[it].z);
        atomicAdd(g + 192, acc[it].w);
    }
}

// Rank-1 gradient of a per-jet GEMV block: gW[(k,o)] += x[k] * dy[o] for k < K, gb[o] += dy[o]  (x, dy in LDS).
// OUT = 128: KM16 block, walked in storage order (coalesced atomics); OUT <= 16: KP16 block.
__device__ __forceinline__ void rank1_atomic(float* __restrict__ gW, float* __restrict__ gb, int K, int OUT,
                                             const float* __restrict__ x, const float* __restrict__ dy) {
    const int tid = threadIdx.x;
    if (OUT == H) {
        const int K16 = (K + 15) & ~15;
        for (int lin = tid; lin < K16 * H; lin += NT) {
            const int k = ((lin >> 11) << 4) + ((lin >> 2) & 15);
            const int o = (((lin >> 6) & 31) << 2) + (lin & 3);
            if (k < K) atomicAdd(gW + lin, x[k] * dy[o]);
        }
        if (tid < H && gb) atomicAdd(gb + tid, dy[tid]);
    } else {
        for (int i = tid; i < K * 16; i += NT) {
            const int o = i & 15;
            if (o < OUT) atomicAdd(gW + i, x[i >> 4] * dy[o]);
        }
        if (tid < OUT && gb) atomicAdd(gb + tid, dy[tid]);
    }
}

// dot of row k of a KM16 block with a 128-vector in LDS
__device__ __forceinline__ float km16_rowdot(const float* __restrict__ W, int k, const float* __restrict__ v) {
    const f32x4* row = reinterpret_cast<const f32x4*>(W) + (k >> 4) * 512 + (k & 15);
    float a = 0.f;
#pragma unroll 8
    for (int i = 0; i < H / 4; ++i) {
        const f32x4 wv = row[i * 16];
        const f32x4 dv = *reinterpret_cast<const f32x4*>(v + 4 * i);
        a += wv.x * dv.x + wv.y * dv.y + wv.z * dv.z + wv.w * dv.w;
    }
    return a;
}

// Backward of the global MLP of one stage (epic.py:180-186 / :375-380).
// In : c.dg = dL/dg_out (L), c.vin = [temb;cond;mean;sum;g_in], g1 / g_out (saved).
// Out: c.dP (gradient w.r.t. the raw pooled sum), c.dg = dL/dg_in (STEM: unused), weight grads.
template <bool STEM>
__device__ __forceinline__ void global_backward(const JetDims& j, const float* __restrict__ blob,
                                                float* __restrict__ gblob, const pfm_dense_lin& gl1,
                                                const pfm_dense_lin& gl2, float* __restrict__ lds, const BCarve& c,
                                                const float* __restrict__ sv_g1, const float* __restrict__ sv_gout) {
    const int tid = threadIdx.x;
    const int TC = j.T + j.C;
    const int K1 = TC + 2 * H + (STEM ? 0 : j.L), K2 = TC + H;
    const float nvalid = lds[c.misc];
    // dag2 = dg_out * phi'(g_out);  vin2 = [temb ; cond ; g1]
    if (tid < j.L) lds[c.dag2 + tid] = lds[c.dg + tid] * dlrelu(sv_gout[tid], j.slope);
    if (tid >= 64 && tid < 64 + TC) lds[c.vin2 + (tid - 64)] = lds[c.vin + (tid - 64)];
    if (tid >= 256 && tid < 256 + H) lds[c.vin2 + TC + (tid - 256)] = sv_g1[tid - 256];
    __syncthreads();
    // dW_gl2 += vin2 (x) dag2 ; db_gl2 += dag2 ; dg1 = W_gl2[g1 rows] . dag2
    rank1_atomic(gblob + gl2.W, gblob + gl2.b, K2, j.L, lds + c.vin2, lds + c.dag2);
    if (tid < H) {
        float a = 0.f;
        for (int jj = 0; jj < j.L; ++jj) a = fmaf(blob[gl2.W + (TC + tid) * 16 + jj], lds[c.dag2 + jj], a);
        lds[c.dag1 + tid] = a * dlrelu(sv_g1[tid], j.slope);  // dag1 = dg1 * phi'(g1)
    }
    __syncthreads();
    // dW_gl1 += vin (x) dag1 ; db_gl1 += dag1
    rank1_atomic(gblob + gl1.W, gblob + gl1.b, K1, H, lds + c.vin, lds + c.dag1);
    // dvin[k] = W_gl1[k][:] . dag1 for k >= TC  -> dmean, dsum (-> dP), dg_in
    auto rowdot = [&](int k) { return km16_rowdot(blob + gl1.W, k, lds + c.dag1); };
    if (tid < H) {
        // pooled mean = sum / n (epic.py:161), pooled sum * scale (:162)
        lds[c.dP + tid] = rowdot(TC + tid) / nvalid + rowdot(TC + H + tid) * j.sscale;
    } else if (!STEM && tid < H + j.L) {
        const int jj = tid - H;
        lds[c.dg + jj] = lds[c.dag2 + jj] + rowdot(TC + 2 * H + jj);  // residual path + vin path
    }
    __syncthreads();
}

}  // namespace pfm
