// One evaluation of the EPiC vector field for ONE jet by ONE 512-thread workgroup (gfx950).
//
// Reference graph: particle_fm/models/components/epic.py:304-391 (EPiC_encoder.forward) and
// :85-203 (EPiC_layer.forward); time embedding time_emb.py:79-96.
//
// Mapping to the hardware
//   * The jet's (N x 128) activations live in LDS for the whole evaluation (two buffers: 153.6 KB at
//     N = 150); nothing but the 3-feature input/output and the weights crosses the CU boundary.
//   * Every 128->128 Linear is out^T = W * x^T on v_mfma_f32_16x16x4_f32 (exact fp32):
//       A = W       : wave w keeps rows [16w,16w+16) x all 128 k in 32 VGPRs for the whole layer,
//       B = x^T     : read from LDS with one ds_read_b128 per four MFMAs (the k order inside the
//                     instruction is permuted so that one 16-byte read feeds four k-steps),
//       D           : lane (particle, q) ends up with 4 consecutive output features -> one
//                     ds_write_b128 back to LDS, bias / residual enter as the accumulator's C.
//   * The columns of each Linear that multiply per-jet quantities (time embedding, conditioning,
//     the broadcast global vector) are folded into a per-jet bias vector by a small GEMV, so the
//     MFMA K is 128 instead of 160/170.
//   * Masked mean/sum pooling: each lane accumulates its 4 features over the particle tiles, then a
//     16-lane xor-shuffle tree, no atomics.
#pragma once
#include "pfm_common.h"

namespace pfm {

struct JetDims {
    int N, F, T, C, Cl, L, layers;
    float slope, sscale;
};

__device__ __forceinline__ JetDims dims_of(const pfm_epic_desc& d) {
    JetDims j;
    j.N = d.n_points; j.F = d.features; j.T = d.t_dim; j.C = d.cond_global; j.Cl = d.cond_local;
    j.L = d.latent; j.layers = d.layers; j.slope = d.neg_slope; j.sscale = d.sum_scale;
    return j;
}

// ---- per-jet saved-activation layout (floats), shared by the loss forward and backward ---------
struct SavedLayout {
    int y, v, u;          // N*F each
    int x1, x2;           // N*H each: stem activations (x2 = input of layer 0)
    int l1, xo;           // base of per-layer l1 / x_out, stride 2*N*H per layer
    int lstride;
    int gstem1, gstem;    // H, MAXL
    int glayer, gstride;  // per layer: g1 (H) | g_new (MAXL)
    int pool, pstride;    // per stage (stem + layers): raw masked sum (H)
    int temb;             // MAXT
    int total;
};

__host__ __device__ inline SavedLayout make_saved(int N, int F, int layers) {
    SavedLayout s;
    int o = 0;
    s.y = o; o += round4(N * F);
    s.v = o; o += round4(N * F);
    s.u = o; o += round4(N * F);
    s.x1 = o; o += N * H;
    s.x2 = o; o += N * H;
    s.l1 = o; s.xo = o + N * H; s.lstride = 2 * N * H; o += layers * 2 * N * H;
    s.gstem1 = o; o += H;
    s.gstem = o; o += MAXL;
    s.glayer = o; s.gstride = H + MAXL; o += layers * (H + MAXL);
    s.pool = o; s.pstride = H; o += (layers + 1) * H;
    s.temb = o; o += MAXT;
    s.total = o;
    return s;
}

// A operand of one 128x128 block for this wave: 8 x float4 = 32 VGPRs (MFMA_A format of pfm_hip.h)
__device__ __forceinline__ void load_afrag(f32x4 (&a)[8], const float* __restrict__ A, int w, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(A) + (w * 8) * 64 + lane;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) a[kt] = p[kt * 64];
}

#define PFM_MFMA4(acc, av, bv)                                                        \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (bv).x, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (bv).y, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (bv).z, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (bv).w, acc, 0, 0, 0);

// dst[p][16w..16w+16) = lrelu( W[16w.., :] . src[p][:] + bj[16w..] (+ dst[p][16w..] if RESID) )
// for p < n_rows.  POOL: pooled[o] = sum_p mask[p] * dst[p][o].  SAVE: also store rows to `save`.
template <bool RESID, bool POOL, bool SAVE>
__device__ __forceinline__ void gemm_phase(const f32x4 (&a)[8], const float* __restrict__ src,
                                           float* __restrict__ dst, const float* __restrict__ bj,
                                           const float* __restrict__ maskf, float* __restrict__ pooled,
                                           float* __restrict__ save, int n_rows, float slope) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int oslot = 4 * w + q;  // 16-byte slot of this lane's 4 output features
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bj + 4 * oslot);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    const int ntiles = (n_rows + TILE - 1) / TILE;
    int tile = 0;
    for (; tile + 1 < ntiles; tile += 2) {
        const int p0 = tile * TILE + pl, p1 = p0 + TILE;
        const int pc1 = min(p1, n_rows - 1);
        f32x4 acc0 = bias, acc1 = bias;
        if (RESID) {
            acc0 += *reinterpret_cast<const f32x4*>(dst + lds_off(p0, oslot));
            acc1 += *reinterpret_cast<const f32x4*>(dst + lds_off(pc1, oslot));
        }
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + lds_off(p0, 4 * kt + q));
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(src + lds_off(pc1, 4 * kt + q));
            PFM_MFMA4(acc0, a[kt], b0);
            PFM_MFMA4(acc1, a[kt], b1);
        }
        acc0 = lrelu4(acc0, slope);
        acc1 = lrelu4(acc1, slope);
        *reinterpret_cast<f32x4*>(dst + lds_off(p0, oslot)) = acc0;
        if (SAVE) *reinterpret_cast<f32x4*>(save + p0 * H + 4 * oslot) = acc0;
        if (POOL) psum += acc0 * maskf[p0];
        if (p1 < n_rows) {
            *reinterpret_cast<f32x4*>(dst + lds_off(p1, oslot)) = acc1;
            if (SAVE) *reinterpret_cast<f32x4*>(save + p1 * H + 4 * oslot) = acc1;
            if (POOL) psum += acc1 * maskf[p1];
        }
    }
    if (tile < ntiles) {
        const int p0 = tile * TILE + pl;
        const int pc0 = min(p0, n_rows - 1);
        f32x4 acc0 = bias;
        if (RESID) acc0 += *reinterpret_cast<const f32x4*>(dst + lds_off(pc0, oslot));
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + lds_off(pc0, 4 * kt + q));
            PFM_MFMA4(acc0, a[kt], b0);
        }
        acc0 = lrelu4(acc0, slope);
        if (p0 < n_rows) {
            *reinterpret_cast<f32x4*>(dst + lds_off(p0, oslot)) = acc0;
            if (SAVE) *reinterpret_cast<f32x4*>(save + p0 * H + 4 * oslot) = acc0;
            if (POOL) psum += acc0 * maskf[p0];
        }
    }
    if (POOL) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            psum.x += __shfl_xor(psum.x, m);
            psum.y += __shfl_xor(psum.y, m);
            psum.z += __shfl_xor(psum.z, m);
            psum.w += __shfl_xor(psum.w, m);
        }
        if (pl == 0) *reinterpret_cast<f32x4*>(pooled + 4 * oslot) = psum;
    }
}

// part[pt*OUTP + o] = sum_{k = pt, pt+PARTS, ...} W[k*OUT + o] * vin[k]   for threads [t0, t0 + OUTP*PARTS)
template <int OUTP, int PARTS>
__device__ __forceinline__ void gemv_part(const float* __restrict__ W, int OUT, int K,
                                          const float* __restrict__ vin, float* __restrict__ part, int t) {
    const int o = t & (OUTP - 1), pt = t / OUTP;
    float acc = 0.f;
    if (o < OUT) {
        for (int k = pt; k < K; k += PARTS) acc = fmaf(W[k * OUT + o], vin[k], acc);
    }
    part[pt * OUTP + o] = acc;
}

// Per-jet bias of a layer's two local linears: bj = b + We^T . e, e = [temb ; cond_l ; (g)]
__device__ __forceinline__ void local_bias_part(const float* __restrict__ blob, const pfm_local_lin& l1,
                                                int K1, const pfm_local_lin& l2, int K2,
                                                const float* __restrict__ e, float* __restrict__ part) {
    const int tid = threadIdx.x;
    if (tid < 256) gemv_part<H, 2>(blob + l1.We, H, K1, e, part, tid);
    else gemv_part<H, 2>(blob + l2.We, H, K2, e, part + 2 * H, tid - 256);
}
__device__ __forceinline__ void local_bias_finish(const float* __restrict__ blob, const pfm_local_lin& l1,
                                                  const pfm_local_lin& l2, const float* __restrict__ part,
                                                  float* __restrict__ bj1, float* __restrict__ bj2) {
    const int tid = threadIdx.x;
    if (tid < H) bj1[tid] = blob[l1.b + tid] + (part[tid] + part[H + tid]);
    else if (tid < 2 * H) {
        const int o = tid - H;
        bj2[o] = blob[l2.b + o] + (part[2 * H + o] + part[3 * H + o]);
    }
}

// Global MLP of one stage.  In: pooled[0..H) raw masked sums, gvec (g_old, unused if STEM).
// Out: gvec = g_new, evec = [temb ; cond_l ; g_new] for the local-bias GEMV.
template <bool STEM, bool SAVE>
__device__ __forceinline__ void global_phase(const JetDims& j, const float* __restrict__ blob,
                                             const pfm_dense_lin& gl1, const pfm_dense_lin& gl2,
                                             float* __restrict__ lds, const Carve& c,
                                             float* __restrict__ evec, float* __restrict__ save_g1,
                                             float* __restrict__ save_g, float* __restrict__ save_pool) {
    const int tid = threadIdx.x;
    float* vin = lds + c.s_vin;
    float* part = lds + c.s_part;
    float* part2 = lds + c.s_part2;
    const int TC = j.T + j.C;
    const float nvalid = lds[c.misc];
    if (tid < H) {
        const float s = lds[c.pooled + tid];
        vin[TC + tid] = s / nvalid;            // epic.py:161 / :370
        vin[TC + H + tid] = s * j.sscale;      // epic.py:162 / :371
        if (SAVE) save_pool[tid] = s;
    } else if (tid < H + j.T) {
        const int k = tid - H;
        vin[k] = lds[c.temb + k];
        evec[k] = lds[c.temb + k];
    } else if (tid < H + j.T + j.C) {
        const int k = tid - H - j.T;
        vin[j.T + k] = lds[c.condv + k];
        if (k < j.Cl) evec[j.T + k] = lds[c.condv + k];
    } else if (!STEM && tid >= 256 && tid < 256 + j.L) {
        vin[TC + 2 * H + (tid - 256)] = lds[c.gvec + (tid - 256)];
    }
    __syncthreads();
    const int K1 = TC + 2 * H + (STEM ? 0 : j.L);
    gemv_part<H, 4>(blob + gl1.W, H, K1, vin, part, tid);
    __syncthreads();
    if (tid < H) {
        const float a = blob[gl1.b + tid] + ((part[tid] + part[H + tid]) + (part[2 * H + tid] + part[3 * H + tid]));
        const float g1 = lrelu(a, j.slope);
        vin[TC + tid] = g1;  // vin2 = [temb ; cond ; g1]
        if (SAVE) save_g1[tid] = g1;
    }
    __syncthreads();
    gemv_part<16, 32>(blob + gl2.W, j.L, TC + H, vin, part2, tid);
    __syncthreads();
    if (tid < j.L) {
        float a = blob[gl2.b + tid];
#pragma unroll
        for (int pt = 0; pt < 32; ++pt) a += part2[pt * 16 + tid];
        if (!STEM) a += lds[c.gvec + tid];  // residual before the activation, epic.py:184-186
        const float g = lrelu(a, j.slope);
        lds[c.gvec + tid] = g;
        evec[j.T + j.Cl + tid] = g;
        if (SAVE) save_g[tid] = g;
    }
    __syncthreads();
}

// Full network body up to (excluding) the fc_l3 head.  Preconditions (in LDS): yin (N x F input),
// maskf, condv, misc[0] = sum(mask), temb.  Postcondition: bufB holds the last hidden state; bj1
// holds nothing useful; evec (scratch in bufA) = [temb ; cond_l ; g].
template <bool SAVE>
__device__ __forceinline__ void epic_body(const pfm_epic_desc& d, const JetDims& j,
                                          const float* __restrict__ blob, float* __restrict__ lds,
                                          const Carve& c, int n_rows, float* __restrict__ saved,
                                          const SavedLayout& sl) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* bufA = lds + c.bufA;
    float* bufB = lds + c.bufB;
    float* bj1 = lds + c.bj1;
    float* bj2 = lds + c.bj2;
    float* part = lds + c.s_part;
    float* evec = lds + c.bufA + 1536;
    const float* maskf = lds + c.maskf;
    const int Ke = j.T + j.Cl;

    f32x4 a1[8], a2[8];
    // ---- stem: per-jet biases of fc_l1 / fc_l2 -------------------------------------------------
    if (tid < j.T) evec[tid] = lds[c.temb + tid];
    else if (tid < Ke) evec[tid] = lds[c.condv + (tid - j.T)];
    load_afrag(a2, blob + d.l2.A, w, lane);
    __syncthreads();
    {
        pfm_local_lin l1s; l1s.A = -1; l1s.AT = -1; l1s.We = d.l1_We; l1s.b = d.l1_b;
        local_bias_part(blob, l1s, Ke, d.l2, Ke, evec, part);
        __syncthreads();
        local_bias_finish(blob, l1s, d.l2, part, bj1, bj2);
        __syncthreads();
    }
    // ---- fc_l1 (K = F, VALU): bufA[p][o] = lrelu(bj1[o] + sum_f Wx[f][o] * y[p][f])  epic.py:360-362
    {
        const int slot = tid & 31;
        const float* Wx = blob + d.l1x.W;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bj1 + 4 * slot);
        for (int p = tid >> 5; p < n_rows; p += NT / 32) {
            f32x4 acc = b4;
            for (int f = 0; f < j.F; ++f) {
                const float yv = lds[c.yin + p * j.F + f];
                const f32x4 wv = *reinterpret_cast<const f32x4*>(Wx + f * H + 4 * slot);
                acc += wv * yv;
            }
            acc = lrelu4(acc, j.slope);
            *reinterpret_cast<f32x4*>(bufA + lds_off(p, slot)) = acc;
            if (SAVE) *reinterpret_cast<f32x4*>(saved + sl.x1 + p * H + 4 * slot) = acc;
        }
    }
    __syncthreads();
    // ---- fc_l2: bufB = lrelu(W.bufA + bj2 + bufA)  epic.py:364-366 ; residual comes from bufA, so
    //      copy-free trick: write into bufB, read residual from the *source* buffer.
    {
        const int pl = lane & 15, q = lane >> 4, oslot = 4 * w + q;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bj2 + 4 * oslot);
        f32x4 psum = {0.f, 0.f, 0.f, 0.f};
        const int ntiles = (n_rows + TILE - 1) / TILE;
        for (int tile = 0; tile < ntiles; ++tile) {
            const int p0 = tile * TILE + pl;
            const int pc0 = min(p0, n_rows - 1);
            f32x4 acc0 = bias + *reinterpret_cast<const f32x4*>(bufA + lds_off(pc0, oslot));
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bufA + lds_off(pc0, 4 * kt + q));
                PFM_MFMA4(acc0, a2[kt], b0);
            }
            acc0 = lrelu4(acc0, j.slope);
            if (p0 < n_rows) {
                *reinterpret_cast<f32x4*>(bufB + lds_off(p0, oslot)) = acc0;
                if (SAVE) *reinterpret_cast<f32x4*>(saved + sl.x2 + p0 * H + 4 * oslot) = acc0;
                psum += acc0 * maskf[p0];
            }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            psum.x += __shfl_xor(psum.x, m);
            psum.y += __shfl_xor(psum.y, m);
            psum.z += __shfl_xor(psum.z, m);
            psum.w += __shfl_xor(psum.w, m);
        }
        if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.pooled + 4 * oslot) = psum;
    }
    __syncthreads();
    // ---- fc_g1 / fc_g2 (epic.py:369-380) ---------------------------------------------------------
    global_phase<true, SAVE>(j, blob, d.g1, d.g2, lds, c, evec, saved + sl.gstem1, saved + sl.gstem,
                             saved + sl.pool);
    // ---- EPiC layers (epic.py:382-385 -> :159-203) -----------------------------------------------
    for (int k = 0; k < j.layers; ++k) {
        const pfm_epic_layer& ly = d.layer[k];
        load_afrag(a1, blob + ly.lc1.A, w, lane);
        load_afrag(a2, blob + ly.lc2.A, w, lane);
        // pooled still holds the masked sum of the current hidden state (bufB)
        global_phase<false, SAVE>(j, blob, ly.gl1, ly.gl2, lds, c, evec,
                                  saved + sl.glayer + k * sl.gstride, saved + sl.glayer + k * sl.gstride + H,
                                  saved + sl.pool + (k + 1) * sl.pstride);
        local_bias_part(blob, ly.lc1, Ke + j.L, ly.lc2, Ke, evec, part);
        __syncthreads();
        local_bias_finish(blob, ly.lc1, ly.lc2, part, bj1, bj2);
        __syncthreads();
        // phase 1: bufA = lrelu(W1 . bufB + bj1)                       epic.py:194-196
        gemm_phase<false, false, SAVE>(a1, bufB, bufA, bj1, maskf, nullptr,
                                       saved + sl.l1 + k * sl.lstride, n_rows, j.slope);
        __syncthreads();
        // phase 2: bufB = lrelu(W2 . bufA + bj2 + bufB), pooled = masked sum    epic.py:198-200, :160
        gemm_phase<true, true, SAVE>(a2, bufA, bufB, bj2, maskf, lds + c.pooled,
                                     saved + sl.xo + k * sl.lstride, n_rows, j.slope);
        __syncthreads();
    }
    // evec (in bufA scratch) was overwritten by phase 1; rebuild [temb ; cond_l] for the head
    if (tid < j.T) evec[tid] = lds[c.temb + tid];
    else if (tid < Ke) evec[tid] = lds[c.condv + (tid - j.T)];
    __syncthreads();
}

// fc_l3 head: emit(p, f, lrelu(b3[f] + We3.e + W3[f].x[p]) * mask[p]) for every p < N
// (rows >= n_rows are emitted as 0: they are masked).  epic.py:387-391
template <typename Emit>
__device__ __forceinline__ void epic_head(const pfm_epic_desc& d, const JetDims& j,
                                          const float* __restrict__ blob, float* __restrict__ lds,
                                          const Carve& c, int n_rows, Emit emit) {
    const int tid = threadIdx.x;
    const float* bufB = lds + c.bufB;
    const float* evec = lds + c.bufA + 1536;
    float* bj3 = lds + c.bj1;  // reuse
    const int Ke = j.T + j.Cl;
    if (tid < j.F) {
        float a = blob[d.l3_b + tid];
        for (int k = 0; k < Ke; ++k) a = fmaf(blob[d.l3_We + k * j.F + tid], evec[k], a);
        bj3[tid] = a;
    }
    __syncthreads();
    const int part = tid & 3;
    for (int base = 0; base < j.N; base += NT / 4) {
        const int p = base + (tid >> 2);
        const bool live = p < n_rows;
        const int pc = live ? p : 0;
        for (int f0 = 0; f0 < j.F; f0 += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int slot = 4 * s + part;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(bufB + lds_off(pc, slot));
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) {
                    if (f0 + jf < j.F) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(lds + c.w3 + (f0 + jf) * H + 4 * slot);
                        acc[jf] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                    }
                }
            }
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) {
                acc[jf] += __shfl_xor(acc[jf], 1);
                acc[jf] += __shfl_xor(acc[jf], 2);
            }
            if (part == 0 && p < j.N) {
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) {
                    if (f0 + jf < j.F) {
                        float val = 0.f;
                        if (live) val = lrelu(acc[jf] + bj3[f0 + jf], j.slope) * lds[c.maskf + p];
                        emit(p, f0 + jf, val);
                    }
                }
            }
        }
    }
}

// Loads that are constant over all evaluations of a jet: mask, cond, head weights; n_valid, n_rows.
// Returns n_rows (the number of leading rows that are computed).
__device__ __forceinline__ int epic_jet_setup(const pfm_epic_desc& d, const JetDims& j,
                                              const float* __restrict__ blob, float* __restrict__ lds,
                                              const Carve& c, const float* __restrict__ cond_jet,
                                              const float* __restrict__ mask_jet) {
    const int tid = threadIdx.x;
    int last = -1;
    float cnt = 0.f;
    for (int p = tid; p < j.N; p += NT) {
        const float m = mask_jet ? mask_jet[p] : 1.0f;
        lds[c.maskf + p] = m;
        cnt += m;
        if (m != 0.f) last = p;
    }
    for (int i = tid; i < j.F * H; i += NT) lds[c.w3 + i] = blob[d.l3_W + i];
    if (tid < j.C) lds[c.condv + tid] = cond_jet[tid];
    // workgroup reduction of cnt (sum) and last (max) through the (still unused) bufA scratch
    for (int m = 32; m >= 1; m >>= 1) {
        cnt += __shfl_xor(cnt, m);
        last = max(last, __shfl_xor(last, m));
    }
    float* red = lds + c.s_part;
    if ((tid & 63) == 0) { red[tid >> 6] = cnt; red[8 + (tid >> 6)] = (float)last; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, l = -1.f;
        for (int i = 0; i < NW; ++i) { s += red[i]; l = fmaxf(l, red[8 + i]); }
        lds[c.misc] = s;
        lds[c.misc + 1] = l;
    }
    __syncthreads();
    int n_rows = j.N;
    if (d.flags & PFM_F_SKIP_MASKED_TAIL) n_rows = max(1, (int)lds[c.misc + 1] + 1);
    return n_rows;
}

// temb[k] = cos(((t + 0) * freqs[k]) * pi / 1)  -- exact fp32 op order of time_emb.py:96
__device__ __forceinline__ void epic_time_embedding(const pfm_epic_desc& d, const JetDims& j,
                                                    const float* __restrict__ blob, float* __restrict__ lds,
                                                    const Carve& c, float t) {
    const int tid = threadIdx.x;
    if (tid < j.T) {
        const float f = blob[d.freqs + tid];
        const float arg = __fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(t, 0.0f), f), 3.14159274101257324f), 1.0f);
        lds[c.temb + tid] = cosf(arg);
    }
}

}  // namespace pfm
