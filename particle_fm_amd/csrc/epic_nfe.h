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

// two independent accumulator chains, alternated instruction by instruction (a dependent
// v_mfma_f32_16x16x4_f32 needs 40 cycles, the pipe issues one every 32)
#ifndef PFM_VAR
#define PFM_VAR 0
#endif
#define PFM_MFMA_PAIR(acc0, acc1, av, bv0, bv1)                                       \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (bv0).x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (bv1).x, acc1, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (bv0).y, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (bv1).y, acc1, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (bv0).z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (bv1).z, acc1, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (bv0).w, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (bv1).w, acc1, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);

template <bool SAVE>
__device__ __forceinline__ void pool_finish(f32x4 psum, const JetDims& j, float* __restrict__ lds, const Carve& c,
                                            int oslot, int pl, float* __restrict__ save_pool);

// One particle phase: for every row p < n_rows
//   dst[p][16w..16w+16) = lrelu( W[16w.., :] . src[p][:] + bj[16w..] (+ resid[p][16w..] if RESID) )
// Each wave walks the particle tiles two at a time (two accumulator chains).  The B operands are
// staged through registers in two halves of the K range so that the LDS reads of one half are in
// flight while the MFMAs of the other half issue (hipcc on its own serialises read -> wait -> 4 MFMAs).
// POOL: masked column sums -> vin (mean | sum*scale).  SAVE: rows also go to `save` (global).
template <bool RESID, bool POOL, bool SAVE>
__device__ __forceinline__ void gemm_phase(const f32x4 (&a)[8], const float* __restrict__ src,
                                           float* __restrict__ dst, const float* __restrict__ resid,
                                           const float* __restrict__ bj, const float* __restrict__ maskf,
                                           const JetDims& j, float* __restrict__ lds, const Carve& c,
                                           float* __restrict__ save, float* __restrict__ save_pool, int n_rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int oslot = 4 * w + q;  // 16-byte slot of this lane's 4 output features
    const float slope = j.slope;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bj + 4 * oslot);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    const int npairs = (n_rows + 2 * TILE - 1) / (2 * TILE);
    f32x4 bA0[4], bA1[4], bB0[4], bB1[4];
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
    // prologue: first half of pair 0
    {
        const int pc0 = min(pl, n_rows - 1), pc1 = min(pl + TILE, n_rows - 1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bA0[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc0, 4 * kt + q));
            bA1[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc1, 4 * kt + q));
        }
        if (RESID) {
            r0 = *reinterpret_cast<const f32x4*>(resid + lds_off(pc0, oslot));
            r1 = *reinterpret_cast<const f32x4*>(resid + lds_off(pc1, oslot));
        }
    }
    for (int pair = 0; pair < npairs; ++pair) {
        const int p0 = pair * 2 * TILE + pl, p1 = p0 + TILE;
        const int pc0 = min(p0, n_rows - 1), pc1 = min(p1, n_rows - 1);
        // second half of this pair: issue, then run the first half's MFMAs underneath
#if PFM_VAR == 3
        if (pair == 0)
#endif
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bB0[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc0, 4 * (kt + 4) + q));
            bB1[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(pc1, 4 * (kt + 4) + q));
        }
        f32x4 acc0 = bias, acc1 = bias;
        if (RESID) { acc0 += r0; acc1 += r1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { PFM_MFMA_PAIR(acc0, acc1, a[kt], bA0[kt], bA1[kt]); }
        __builtin_amdgcn_sched_barrier(0);
        // first half of the NEXT pair (rows clamp, so the last iteration just re-reads valid rows)
#if PFM_VAR == 3
        if (pair < 0)
#endif
        {
            const int n0 = min(p0 + 2 * TILE, n_rows - 1), n1 = min(p1 + 2 * TILE, n_rows - 1);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bA0[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(n0, 4 * kt + q));
                bA1[kt] = *reinterpret_cast<const f32x4*>(src + lds_off(n1, 4 * kt + q));
            }
            if (RESID) {
                r0 = *reinterpret_cast<const f32x4*>(resid + lds_off(n0, oslot));
                r1 = *reinterpret_cast<const f32x4*>(resid + lds_off(n1, oslot));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { PFM_MFMA_PAIR(acc0, acc1, a[kt + 4], bB0[kt], bB1[kt]); }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = lrelu4(acc0, slope);
        acc1 = lrelu4(acc1, slope);
#if PFM_VAR == 2
        if (acc0.x == 12345.678f) *reinterpret_cast<f32x4*>(dst + lds_off(p0, oslot)) = acc0 + acc1;
#else
        if (p0 < n_rows) {
            *reinterpret_cast<f32x4*>(dst + lds_off(p0, oslot)) = acc0;
            if (SAVE) *reinterpret_cast<f32x4*>(save + p0 * H + 4 * oslot) = acc0;
            if (POOL) psum += acc0 * maskf[p0];
        }
        if (p1 < n_rows) {
            *reinterpret_cast<f32x4*>(dst + lds_off(p1, oslot)) = acc1;
            if (SAVE) *reinterpret_cast<f32x4*>(save + p1 * H + 4 * oslot) = acc1;
            if (POOL) psum += acc1 * maskf[p1];
        }
#endif
    }
    if (POOL) pool_finish<SAVE>(psum, j, lds, c, oslot, pl, save_pool);
}

// ---- per-jet GEMVs (global MLP, per-jet biases) ------------------------------------------------
// All KMAJOR [K][128] matrices are read as float4 over 4 consecutive outputs: thread (og, pt) with
// og = tid & 31 (output group), pt = tid >> 5 (one of 16 k-partitions, k = pt, pt+16, ...).  The loads
// of a round are all issued before the first FMA so that one L2 round trip covers the whole GEMV.
// GEMV over a KMAJOR [K16][128] block (rows zero-padded to a multiple of 16, see pfm_hip.h).
// Thread (og, pt) owns outputs 4*og..4*og+3 and rows k = pt + 16*i.  Loads and FMAs are two steps so
// that several GEMVs can have their loads in flight together; `row` = pt*32 + og is the only per-lane
// address term, everything else is a wave-uniform offset.
template <int U>
__device__ __forceinline__ void gemv4_load(f32x4 (&wv)[U], const float* __restrict__ W, int K, int base, int row) {
    const f32x4* W4 = reinterpret_cast<const f32x4*>(W) + row;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = base + u;
        if (16 * i < K) wv[u] = W4[i * 16 * (H / 4)];  // uniform predicate
        else wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int U>
__device__ __forceinline__ void gemv4_fma(f32x4& acc, const f32x4 (&wv)[U], const float* __restrict__ vin, int K,
                                          int base, int pt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = pt + 16 * (base + u);
        const float x = k < K ? vin[k] : 0.f;
        acc += wv[u] * x;
    }
}

struct LocalBiasSrc {  // the two local linears whose per-jet bias a stage prepares
    int64_t We1, b1, We2, b2;
};

// t/cond part of both local biases (K = T + Cl <= 96) -> partials
__device__ __forceinline__ void local_bias_tc_part(const float* __restrict__ blob, const LocalBiasSrc& lb, int Ke,
                                                   const float* __restrict__ vin, float* __restrict__ lds,
                                                   const Carve& c) {
    const int tid = threadIdx.x, og = tid & 31, pt = tid >> 5;
    f32x4 w1[6], w2[6];
    gemv4_load<6>(w1, blob + lb.We1, Ke, 0, tid);
    gemv4_load<6>(w2, blob + lb.We2, Ke, 0, tid);
    f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
    gemv4_fma<6>(p1, w1, vin, Ke, 0, pt);
    gemv4_fma<6>(p2, w2, vin, Ke, 0, pt);
    *reinterpret_cast<f32x4*>(lds + c.s_pb1 + pt * H + 4 * og) = p1;
    *reinterpret_cast<f32x4*>(lds + c.s_pb2 + pt * H + 4 * og) = p2;
}

__device__ __forceinline__ float sum16(const float* __restrict__ part, int o) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 16; pt += 2) { s0 += part[pt * H + o]; s1 += part[(pt + 1) * H + o]; }
    return s0 + s1;
}

// The per-jet phase between two particle phases:
//   g1 = lrelu(Wg1.[temb;cond;mean;sum;g] + b)            epic.py:180-182 / :375-377
//   g  = lrelu(Wg2.[temb;cond;g1] + b (+ g))              epic.py:184-186 / :378-380
//   bj1 = b1 + We1.[temb;cond_l;g],  bj2 = b2 + We2.[temb;cond_l]     (folded t/cond/global columns)
// In : vin = [temb;cond;mean;sum;g_old] complete (the pooled part was written by the previous
//      particle phase), barrier already passed.  Out: vin.g = g_new, bj1/bj2 ready, barrier passed.
// STEM: fc_g1/fc_g2 (no g input, no residual) and no local biases.
template <bool STEM, bool SAVE>
__device__ __forceinline__ void per_jet_phase(const JetDims& j, const float* __restrict__ blob,
                                              const pfm_dense_lin& gl1, const pfm_dense_lin& gl2,
                                              const LocalBiasSrc& lb, float* __restrict__ lds, const Carve& c,
                                              float* __restrict__ save_g1, float* __restrict__ save_g) {
    const int tid = threadIdx.x, og = tid & 31, pt = tid >> 5;
    float* vin = lds + c.vin;
    float* vin2 = lds + c.s_vin2;
    const int TC = j.T + j.C, Ke = j.T + j.Cl;
    const int K1 = TC + 2 * H + (STEM ? 0 : j.L);
    const int K2 = TC + H;
    // ---- prefetch everything whose address does not depend on data ----
    float w2[7];  // fc_global2 weights of thread (o2 = tid & 15, pt2 = tid >> 4): k = pt2 + 32 i
    const int o2 = tid & 15, pt2 = tid >> 4;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = pt2 + 32 * i;
        w2[i] = (o2 < j.L && k < K2) ? blob[gl2.W + k * j.L + o2] : 0.f;
    }
    float wg[MAXL];  // g rows of local linear 1 for output o = tid (threads < H)
    float bias_pre = 0.f;
    if (!STEM) {
#pragma unroll
        for (int jj = 0; jj < MAXL; ++jj) wg[jj] = (tid < H && jj < j.L) ? blob[lb.We1 + (Ke + jj) * H + tid] : 0.f;
    }
    if (tid < H) bias_pre = blob[gl1.b + tid];
    else if (!STEM && tid < 2 * H) bias_pre = blob[lb.b1 + (tid - H)];
    else if (!STEM && tid < 3 * H) bias_pre = blob[lb.b2 + (tid - 2 * H)];
    // ---- S2: fc_global1 partials (+ t/cond part of the local biases); K1 <= 352 < 16 * 24 ----
    {
        f32x4 wa[8], wb[6];
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        gemv4_load<8>(wa, blob + gl1.W, K1, 0, tid);
        if (!STEM) gemv4_load<6>(wb, blob + lb.We1, Ke, 0, tid);
        gemv4_fma<8>(p, wa, vin, K1, 0, pt);
        gemv4_load<8>(wa, blob + gl1.W, K1, 8, tid);
        if (!STEM) {
            f32x4 p1 = {0.f, 0.f, 0.f, 0.f};
            gemv4_fma<6>(p1, wb, vin, Ke, 0, pt);
            *reinterpret_cast<f32x4*>(lds + c.s_pb1 + pt * H + 4 * og) = p1;
            gemv4_load<6>(wb, blob + lb.We2, Ke, 0, tid);
        }
        gemv4_fma<8>(p, wa, vin, K1, 8, pt);
        if (K1 > 256) gemv4_load<8>(wa, blob + gl1.W, K1, 16, tid);
        if (!STEM) {
            f32x4 p2 = {0.f, 0.f, 0.f, 0.f};
            gemv4_fma<6>(p2, wb, vin, Ke, 0, pt);
            *reinterpret_cast<f32x4*>(lds + c.s_pb2 + pt * H + 4 * og) = p2;
        }
        if (K1 > 256) gemv4_fma<8>(p, wa, vin, K1, 16, pt);
        *reinterpret_cast<f32x4*>(lds + c.s_part + pt * H + 4 * og) = p;
    }
    __syncthreads();
    // ---- S3: reduce ----
    if (tid < H) {
        const float g1 = lrelu(bias_pre + sum16(lds + c.s_part, tid), j.slope);
        vin2[TC + tid] = g1;
        if (SAVE) save_g1[tid] = g1;
    } else if (!STEM && tid < 2 * H) {
        lds[c.s_bj1p + (tid - H)] = bias_pre + sum16(lds + c.s_pb1, tid - H);
    } else if (!STEM && tid < 3 * H) {
        lds[c.bj2 + (tid - 2 * H)] = bias_pre + sum16(lds + c.s_pb2, tid - 2 * H);
    } else if (tid >= 3 * H && tid < 3 * H + TC) {
        vin2[tid - 3 * H] = vin[tid - 3 * H];
    }
    __syncthreads();
    // ---- S4: fc_global2 partials from the prefetched weights ----
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int k = pt2 + 32 * i;
            acc = fmaf(w2[i], k < K2 ? vin2[k] : 0.f, acc);
        }
        lds[c.s_part2 + pt2 * 16 + o2] = acc;
    }
    __syncthreads();
    // ---- S5: g_new ----
    if (tid < j.L) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < 32; p2 += 2) { a0 += lds[c.s_part2 + p2 * 16 + tid]; a1 += lds[c.s_part2 + (p2 + 1) * 16 + tid]; }
        float a = blob[gl2.b + tid] + (a0 + a1);
        if (!STEM) a += vin[TC + 2 * H + tid];  // residual before the activation, epic.py:184-186
        const float g = lrelu(a, j.slope);
        vin[TC + 2 * H + tid] = g;
        if (SAVE) save_g[tid] = g;
    }
    __syncthreads();
    // ---- S6: add the g part to bias 1 ----
    if (!STEM) {
        if (tid < H) {
            float a = lds[c.s_bj1p + tid];
#pragma unroll
            for (int jj = 0; jj < MAXL; ++jj) a = fmaf(wg[jj], jj < j.L ? vin[TC + 2 * H + jj] : 0.f, a);
            lds[c.bj1 + tid] = a;
        }
        __syncthreads();
    }
}

// masked pooling tail of a particle phase: 16-lane tree, then mean / scaled sum straight into vin
template <bool SAVE>
__device__ __forceinline__ void pool_finish(f32x4 psum, const JetDims& j, float* __restrict__ lds, const Carve& c,
                                            int oslot, int pl, float* __restrict__ save_pool) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        psum.x += __shfl_xor(psum.x, m);
        psum.y += __shfl_xor(psum.y, m);
        psum.z += __shfl_xor(psum.z, m);
        psum.w += __shfl_xor(psum.w, m);
    }
    if (pl == 0) {
        const float nvalid = lds[c.misc];
        const int TC = j.T + j.C;
        f32x4 mean;
        mean.x = psum.x / nvalid; mean.y = psum.y / nvalid; mean.z = psum.z / nvalid; mean.w = psum.w / nvalid;  // epic.py:161/:370
        *reinterpret_cast<f32x4*>(lds + c.vin + TC + 4 * oslot) = mean;
        *reinterpret_cast<f32x4*>(lds + c.vin + TC + H + 4 * oslot) = psum * j.sscale;  // epic.py:162/:371
        if (SAVE) *reinterpret_cast<f32x4*>(save_pool + 4 * oslot) = psum;
    }
}

template <int FM, bool SAVE>
__device__ __forceinline__ void stem_l1(const pfm_epic_desc& d, const JetDims& j, const float* __restrict__ blob,
                                        float* __restrict__ lds, const Carve& c, int n_rows,
                                        float* __restrict__ save_x1) {
    const int tid = threadIdx.x, slot = tid & 31;
    const float* Wx = blob + d.l1x.W;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + c.bj1 + 4 * slot);
    f32x4 wv[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) wv[f] = *reinterpret_cast<const f32x4*>(Wx + min(f, j.F - 1) * H + 4 * slot);
    for (int p = tid >> 5; p < n_rows; p += NT / 32) {
        f32x4 acc = b4;
#pragma unroll
        for (int f = 0; f < FM; ++f)
            if (f < j.F) acc += wv[f] * lds[c.yin + p * j.F + f];
        acc = lrelu4(acc, j.slope);
        *reinterpret_cast<f32x4*>(lds + c.bufA + lds_off(p, slot)) = acc;
        if (SAVE) *reinterpret_cast<f32x4*>(save_x1 + p * H + 4 * slot) = acc;
    }
}

// Full network body up to (excluding) the fc_l3 head.  Preconditions (in LDS): yin (N x F input),
// maskf, misc[0] = sum(mask), vin.temb, vin.cond.  Postcondition: bufB holds the last hidden state.
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
    const float* vin = lds + c.vin;
    const float* maskf = lds + c.maskf;
    const int Ke = j.T + j.Cl;
    const int TC = j.T + j.C;
    if (TC > j.T + j.Cl) { /* cond_local == 0 while cond_global > 0: local extras use only temb */ }

    f32x4 a1[8], a2[8];
    PFM_STAMP(1);
    // ---- stem: per-jet biases of fc_l1 / fc_l2 (t / cond columns) ------------------------------
    load_afrag(a2, blob + d.l2.A, w, lane);
    {
        LocalBiasSrc lb; lb.We1 = d.l1_We; lb.b1 = d.l1_b; lb.We2 = d.l2.We; lb.b2 = d.l2.b;
        local_bias_tc_part(blob, lb, Ke, vin, lds, c);
        __syncthreads();
        if (tid < H) bj1[tid] = blob[lb.b1 + tid] + sum16(lds + c.s_pb1, tid);
        else if (tid < 2 * H) bj2[tid - H] = blob[lb.b2 + (tid - H)] + sum16(lds + c.s_pb2, tid - H);
        __syncthreads();
    }
    PFM_STAMP(2);
    // ---- fc_l1 (K = F, VALU): bufA[p][o] = lrelu(bj1[o] + sum_f Wx[f][o] * y[p][f])  epic.py:360-362
    if (j.F <= 4) stem_l1<4, SAVE>(d, j, blob, lds, c, n_rows, saved + sl.x1);
    else stem_l1<MAXF, SAVE>(d, j, blob, lds, c, n_rows, saved + sl.x1);
    __syncthreads();
    PFM_STAMP(3);
    // ---- fc_l2: bufB = lrelu(W.bufA + bj2 + bufA)  epic.py:364-366 (residual from the source buffer)
    gemm_phase<true, true, SAVE>(a2, bufA, bufB, bufA, bj2, maskf, j, lds, c, saved + sl.x2, saved + sl.pool, n_rows);
    __syncthreads();
    PFM_STAMP(4);
    // ---- fc_g1 / fc_g2 (epic.py:369-380) ---------------------------------------------------------
    {
        LocalBiasSrc none; none.We1 = none.b1 = none.We2 = none.b2 = 0;
        per_jet_phase<true, SAVE>(j, blob, d.g1, d.g2, none, lds, c, saved + sl.gstem1, saved + sl.gstem);
    }
    // ---- EPiC layers (epic.py:382-385 -> :159-203) -----------------------------------------------
    for (int k = 0; k < j.layers; ++k) {
        const pfm_epic_layer& ly = d.layer[k];
        PFM_STAMP(10);
        // vin still holds mean / sum of the current hidden state (bufB) and g
        LocalBiasSrc lb; lb.We1 = ly.lc1.We; lb.b1 = ly.lc1.b; lb.We2 = ly.lc2.We; lb.b2 = ly.lc2.b;
        per_jet_phase<false, SAVE>(j, blob, ly.gl1, ly.gl2, lb, lds, c, saved + sl.glayer + k * sl.gstride,
                                   saved + sl.glayer + k * sl.gstride + H);
        PFM_STAMP(12);
        load_afrag(a1, blob + ly.lc1.A, w, lane);
        load_afrag(a2, blob + ly.lc2.A, w, lane);  // lands behind phase 1's MFMAs
        // phase 1: bufA = lrelu(W1 . bufB + bj1)                       epic.py:194-196
        gemm_phase<false, false, SAVE>(a1, bufB, bufA, nullptr, bj1, maskf, j, lds, c, saved + sl.l1 + k * sl.lstride,
                                       nullptr, n_rows);
        __syncthreads();
        PFM_STAMP(13);
        // phase 2: bufB = lrelu(W2 . bufA + bj2 + bufB), pooled -> vin    epic.py:198-200, :160-162
        gemm_phase<true, true, SAVE>(a2, bufA, bufB, bufB, bj2, maskf, j, lds, c, saved + sl.xo + k * sl.lstride,
                                     saved + sl.pool + (k + 1) * sl.pstride, n_rows);
        __syncthreads();
    }
    PFM_STAMP(20);
}

// fc_l3 head: emit(p, f, lrelu(b3[f] + We3.[temb;cond_l] + W3[f].x[p]) * mask[p]) for every p < N
// (rows >= n_rows are emitted as 0: they are masked).  epic.py:387-391
template <typename Emit>
__device__ __forceinline__ void epic_head(const pfm_epic_desc& d, const JetDims& j,
                                          const float* __restrict__ blob, float* __restrict__ lds,
                                          const Carve& c, int n_rows, Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* bufB = lds + c.bufB;
    const float* vin = lds + c.vin;
    float* bj3 = lds + c.bj1;  // reuse
    const int Ke = j.T + j.Cl;
    // bj3[f] = b3[f] + sum_k We3[k][f] * e[k]: wave w takes features w, w+8; lanes split k
    for (int f = w; f < j.F; f += NW) {
        float a = 0.f;
        for (int k = lane; k < Ke; k += 64) a = fmaf(blob[d.l3_We + k * j.F + f], vin[k], a);
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
        if (lane == 0) bj3[f] = a + blob[d.l3_b + f];
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
    if (tid < j.C) lds[c.vin + j.T + tid] = cond_jet[tid];
    if (tid >= 64 && tid < 64 + MAXL) lds[c.vin + j.T + j.C + 2 * H + (tid - 64)] = 0.f;
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

// vin.temb[k] = cos(((t + 0) * freqs[k]) * pi / 1)  -- exact fp32 op order of time_emb.py:96
__device__ __forceinline__ void epic_time_embedding(const pfm_epic_desc& d, const JetDims& j,
                                                    const float* __restrict__ blob, float* __restrict__ lds,
                                                    const Carve& c, float t) {
    const int tid = threadIdx.x;
    if (tid < j.T) {
        const float f = blob[d.freqs + tid];
        const float arg = __fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(t, 0.0f), f), 3.14159274101257324f), 1.0f);
        lds[c.vin + tid] = cosf(arg);
    }
}

}  // namespace pfm
