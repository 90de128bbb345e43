// Forward kernels of the Full-Transformer vector field (gfx950, fp32 MFMA).
//
// Reference graph: particle_fm/models/components/droid_transformer.py:529-548 (FullTransformerEncoder.forward),
// :331-344 (TransformerEncoderLayer), :231-284 (MultiHeadedAttentionBlock), :793-813 / :958-981 (MLPBlock /
// DenseNetwork); time embedding time_emb.py:79-96.
//
// Kernels (one NFE = ctxt, embed, then Linear / attention launches on the caller's stream):
//   tf_ctxt_kernel    one workgroup per jet: time embedding and ctxt_emdb.
//   tf_jetbias_kernel one workgroup per (jet, Linear that takes the context): its "jet bias" row = bias + the context
//                     (/ time) columns times the per-jet vectors.
//   tf_embed_kernel   node_embd input block: K = F particle columns on the VALU (+ jet bias), LeakyReLU.
//   tf_linear_kernel  out = epi( LN?(A) . W^T + b + jetbias ) (+ residual): 64 rows x 128 outputs per workgroup,
//                     K streamed in chunks of 128 through LDS, weights are the MFMA A operand straight from
//                     L2 (MFMA_AK blocks), activations the B operand via ds_read_b128 (k permuted so that one
//                     16-byte read feeds four v_mfma_f32_16x16x4_f32).
//   tf_attn_kernel    one workgroup per (jet, head): K and V^T of the head in LDS, S^T = K Q^T and O^T = V^T P^T
//                     on MFMA with the whole score row of a query kept in registers (exact two-pass softmax); the
//                     accumulator layout of S^T is already the B-operand layout of P^T, so P never leaves the VGPRs.
//   tf_head_kernel    outp_embd output block (F <= 16 outputs): LayerNorm + F dot products per row, 16 lanes per
//                     row, fused with the midpoint state update when sampling.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "tf_common.h"

namespace pfm {
namespace tf {

// ------------------------------------------------------------------------------------------------
// block-wide sum for the 512-thread per-jet kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_sum512(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i];
    return s;
}

constexpr int CTXT_MAX_NB = 34;  // jet-bias rows one context kernel can produce (transformer: layers + 2; cross-attention: 2 * pairs + 2)

struct CtxtArgs {
    const float* blob;
    const float* t;
    const float* cond;
    float *temb, *chid, *ctxt, *jb;
    int t_stride, T, C, CH, CO, Hd, nb, sincos;
    int temb_k = 0;  // > 0 (PFM_*_F_TEMB_GIVEN): `t` holds the time EMBEDDING -- element k of a jet's row at t[jet * t_stride + k * temb_k]
    float slope, eps;
    int64_t freqs, c1W, c1b, cg, cb, c2W, c2b, n1Wt;
    int64_t Wc[CTXT_MAX_NB], bb[CTXT_MAX_NB];
};

static __global__ __launch_bounds__(512) void tf_ctxt_kernel(CtxtArgs a) {
    __shared__ float cin[96];
    __shared__ float hb[512];
    __shared__ float part[512];
    __shared__ float red[8];
    const int tid = threadIdx.x, jet = blockIdx.x;
    const float* __restrict__ blob = a.blob;
    if (tid < a.T) {
        // time_emb.py:90-96, exact fp32 op order ((t + min) * f) * pi / (max + min)
        float e;
        if (a.temb_k > 0) {  // a caller-supplied embedding (t_emb="gaussian": a small trainable network in front of the field)
            e = a.t[(int64_t)jet * a.t_stride + (int64_t)tid * a.temb_k];
            cin[tid] = e;
            a.temb[(int64_t)jet * 64 + tid] = e;
        } else {
        const float t = a.t[(int64_t)jet * a.t_stride];
        const float f = blob[a.freqs + tid];
        if (a.sincos) {  // flow_matching_module.py:208-211 (table = [f ; f], f = 2^k pi)
            const float arg = __fmul_rn(f, t);
            e = 2 * tid < a.T ? cosf(arg) : sinf(arg);
        } else {
            e = cosf(__fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(t, 0.0f), f), 3.14159274101257324f), 1.0f));
        }
        cin[tid] = e;
        a.temb[(int64_t)jet * 64 + tid] = e;
        }
    } else if (tid < a.T + a.C) {
        cin[tid] = a.cond[(int64_t)jet * a.C + tid - a.T];
    }
    __syncthreads();
    const int Kc = a.T + a.C;
    // ctxt_emdb input block: Linear, LeakyReLU (the LayerNorm follows)
    for (int o = tid; o < a.CH; o += 512) {
        float acc = blob[a.c1b + o];
#pragma unroll 8
        for (int k = 0; k < Kc; ++k) acc = fmaf(blob[a.c1W + (int64_t)k * a.CH + o], cin[k], acc);
        acc = lrelu(acc, a.slope);
        hb[o] = acc;
        a.chid[(int64_t)jet * a.CH + o] = acc;
    }
    __syncthreads();
    float s = 0.f;
    for (int o = tid; o < a.CH; o += 512) s += hb[o];
    const float mean = block_sum512(s, red) / (float)a.CH;
    float ss = 0.f;
    for (int o = tid; o < a.CH; o += 512) {
        const float dlt = hb[o] - mean;
        ss = fmaf(dlt, dlt, ss);
    }
    const float rstd = 1.0f / sqrtf(block_sum512(ss, red) / (float)a.CH + a.eps);
    for (int o = tid; o < a.CH; o += 512) hb[o] = (hb[o] - mean) * rstd * blob[a.cg + o] + blob[a.cb + o];
    __syncthreads();
    // ctxt_emdb output block: CO <= 64 outputs, 8 partial sums each
    {
        const int j = tid & 63, p = tid >> 6;
        float acc = 0.f;
        if (j < a.CO)
#pragma unroll 8
            for (int o = p; o < a.CH; o += 8) acc = fmaf(blob[a.c2W + (int64_t)o * a.CO + j], hb[o], acc);
        part[tid] = acc;
    }
    __syncthreads();
    if (tid < a.CO) {
        float acc = blob[a.c2b + tid];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc += part[p * 64 + tid];
        a.ctxt[(int64_t)jet * a.CO + tid] = acc;
    }
}

// jet-bias rows: jb[jet][c][:] = b_c + Wc_c . ctxt (+ Wt . temb for node_embd, c = 0).  grid (n_jets, nb): one row per
// workgroup, so the nb * Hd dot products of a jet run side by side instead of one row after the other.
static __global__ __launch_bounds__(256) void tf_jetbias_kernel(CtxtArgs a) {
    __shared__ float cx[64];
    __shared__ float te[64];
    const int tid = threadIdx.x, jet = blockIdx.x, c = blockIdx.y;
    const float* __restrict__ blob = a.blob;
    const bool with_t = c == 0 && a.n1Wt >= 0;
    if (tid < a.CO) cx[tid] = a.ctxt[(int64_t)jet * a.CO + tid];
    if (with_t && tid >= 64 && tid < 64 + a.T) te[tid - 64] = a.temb[(int64_t)jet * 64 + tid - 64];
    __syncthreads();
    for (int o = tid; o < a.Hd; o += 256) {
        float acc = blob[a.bb[c] + o], acc2 = 0.f;
#pragma unroll 8
        for (int j = 0; j < a.CO; j += 2) {
            acc = fmaf(blob[a.Wc[c] + (int64_t)j * a.Hd + o], cx[j], acc);
            acc2 = fmaf(blob[a.Wc[c] + (int64_t)(j + 1) * a.Hd + o], cx[j + 1], acc2);
        }
        acc += acc2;
        if (with_t)
#pragma unroll 8
            for (int k = 0; k < a.T; ++k) acc = fmaf(blob[a.n1Wt + (int64_t)k * a.Hd + o], te[k], acc);
        a.jb[((int64_t)jet * a.nb + c) * a.Hd + o] = acc;
    }
}

inline void launch_ctxt(const CtxtArgs& a, int n_jets, hipStream_t s) {
    hipLaunchKernelGGL(tf_ctxt_kernel, dim3(n_jets), dim3(512), 0, s, a);
    hipLaunchKernelGGL(tf_jetbias_kernel, dim3(n_jets, a.nb), dim3(256), 0, s, a);
}

// node_embd input block (droid_transformer.py:793-813 on cat(temb, x, ctxt)): the F particle columns
static __global__ __launch_bounds__(256) void tf_embed_kernel(const float* __restrict__ blob, int64_t Wx,
                                                       const float* __restrict__ x, const float* __restrict__ jb,
                                                       int64_t jb_stride, float* __restrict__ h1, int M, int N, int F,
                                                       int Hd, float slope, const int* __restrict__ rowsrc = nullptr,
                                                       const int* __restrict__ rowjet = nullptr,
                                                       const int* __restrict__ m_dev = nullptr) {
    const int nc4 = Hd >> 2;
    const int row0 = blockIdx.x * 32;
    if (m_dev) M = *m_dev;
    for (int idx = threadIdx.x; idx < 32 * nc4; idx += 256) {
        const int r = idx / nc4, c4 = idx - r * nc4;
        const int row = row0 + r;
        if (row >= M) break;
        const int jet = rowjet ? rowjet[row] : row / N;
        const int64_t xrow = rowsrc ? rowsrc[row] : row;
        f32x4 acc = *reinterpret_cast<const f32x4*>(jb + (int64_t)jet * jb_stride + 4 * c4);
        for (int f = 0; f < F; ++f) {
            const float xv = x[xrow * F + f];
            const f32x4 wv = *reinterpret_cast<const f32x4*>(blob + Wx + (int64_t)f * Hd + 4 * c4);
            acc.x = fmaf(wv.x, xv, acc.x); acc.y = fmaf(wv.y, xv, acc.y);
            acc.z = fmaf(wv.z, xv, acc.z); acc.w = fmaf(wv.w, xv, acc.w);
        }
        *reinterpret_cast<f32x4*>(h1 + (int64_t)row * Hd + 4 * c4) = lrelu4(acc, slope);
    }
}

// ---- row compaction (inference): rows = valid particles only, in (jet, particle) order -----------------------
static __global__ __launch_bounds__(256) void rows_count_kernel(const float* __restrict__ mask, int* __restrict__ cnt, int N) {
    __shared__ int red[4];
    const int jet = blockIdx.x;
    int c = 0;
    for (int r = threadIdx.x; r < N; r += 256) c += mask[(int64_t)jet * N + r] != 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[jet] = red[0] + red[1] + red[2] + red[3];
}
// off[0..B] = exclusive scan of cnt; m_valid = off[B]   (one workgroup; B is a batch size)
static __global__ __launch_bounds__(1024) void rows_scan_kernel(const int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ m_valid, int B) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (B + 1023) / 1024;
    int s = 0;
    for (int i = 0; i < per; ++i) { const int j = tid * per + i; if (j < B) s += cnt[j]; }
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = 0; i < per; ++i) { const int j = tid * per + i; if (j < B) { off[j] = run; run += cnt[j]; } }
    if (tid == 1023) { off[B] = part[1023]; *m_valid = part[1023]; }
}
// rowsrc / rowjet of every valid particle (wave 0 of the jet's workgroup walks the mask in order)
static __global__ __launch_bounds__(64) void rows_map_kernel(const float* __restrict__ mask, const int* __restrict__ off,
                                                       int* __restrict__ rowsrc, int* __restrict__ rowjet, int N) {
    const int jet = blockIdx.x, lane = threadIdx.x;
    int base = off[jet];
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int r = r0 + lane;
        const bool ok = r < N && mask[(int64_t)jet * N + r] != 0.f;
        const unsigned long long bal = __ballot(ok);
        if (ok) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            rowsrc[pos] = jet * N + r;
            rowjet[pos] = jet;
        }
        base += __popcll(bal);
    }
}
// order[rank] = jet, descending multiplicity (ties by index): kernels with one workgroup per jet whose cost grows with the
// jet's size (attention ~ n^2) dispatch the long jets first and fill the tail with the short ones.  One workgroup, B <= 8192.
constexpr int RANK_MAX_JETS = 8192;
static __global__ __launch_bounds__(1024) void rows_rank_kernel(const int* __restrict__ cnt, int B, int* __restrict__ order) {
    __shared__ int c[RANK_MAX_JETS];
    for (int j = threadIdx.x; j < B; j += 1024) c[j] = cnt[j];
    __syncthreads();
    for (int j = threadIdx.x; j < B; j += 1024) {
        const int cj = c[j];
        int rank = 0;
        for (int k = 0; k < B; ++k) rank += (c[k] > cj) || (c[k] == cj && k < j);
        order[rank] = j;
    }
}
// int32 scratch for the maps of n_jets x N rows: cnt[B] | off[B+1] | m_valid (padded to 64) | rowsrc[M] | rowjet[M] | order[B]
__host__ __device__ inline int64_t row_maps_ints(int64_t n_jets, int64_t M) { return 3 * n_jets + 64 + 2 * M; }
struct RowMaps {
    int *cnt, *off, *m_dev, *rowsrc, *rowjet, *order;  // order: nullptr for batches beyond RANK_MAX_JETS
};
inline RowMaps build_row_maps(int* im, const float* mask, int B, int N, hipStream_t s) {
    RowMaps m;
    m.cnt = im; m.off = im + B; m.m_dev = im + 2 * B + 1; m.rowsrc = im + 2 * B + 64; m.rowjet = m.rowsrc + (int64_t)B * N;
    hipLaunchKernelGGL(rows_count_kernel, dim3(B), dim3(256), 0, s, mask, m.cnt, N);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)m.cnt, m.off, m.m_dev, B);
    hipLaunchKernelGGL(rows_map_kernel, dim3(B), dim3(64), 0, s, mask, (const int*)m.off, m.rowsrc, m.rowjet, N);
    m.order = nullptr;
    if (B >= 2 && B <= RANK_MAX_JETS) {
        m.order = m.rowjet + (int64_t)B * N;
        hipLaunchKernelGGL(rows_rank_kernel, dim3(1), dim3(1024), 0, s, (const int*)m.cnt, B, m.order);
    }
    return m;
}

// raw field of a compacted evaluation: the rows it never touches are 0 (NaN for a jet without any valid particle when
// `nan_empty`: EPiC's 0 / 0 mean poisons the whole jet, epic.py:331-339)
static __global__ __launch_bounds__(256) void rows_fill_masked_kernel(const float* __restrict__ mask, const int* __restrict__ cnt,
                                                               const float* __restrict__ base, float* __restrict__ dst, int64_t M,
                                                               int N, int F, int nan_empty) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * F) return;
    const int64_t row = i / F;
    if (mask[row] != 0.f) return;
    dst[i] = (nan_empty && cnt[row / N] == 0) ? __builtin_nanf("") : (base ? base[i] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// Linear
// ------------------------------------------------------------------------------------------------
struct LinArgs {
    const float* A;     // [M][lda], first K columns (first K1 when a second segment A2 is given)
    const float* A2;    // optional second input segment [M][lda2]: columns K1.. of the input come from A2[:, 0..K-K1)
    const float* blob;
    const float* jb;    // jet-bias rows (already contain the bias) or nullptr
    const float* R;     // residual [M][ldr] or nullptr
    const float* Y;     // act 3: activation output [M][ldy] whose LeakyReLU' scales the result
    const int* rowjet;  // row -> jet for the jet-bias lookup (compacted rows), or nullptr: jet = row / N
    const int* m_dev;   // device-side row count overriding M (compacted rows), or nullptr
    float* part;        // split-K: raw partial sums [ksplit][M][NO] (bias / activation / residual applied by tf_splitk_kernel)
    int ksplit;         // 1: the whole K in one workgroup
    float* out;         // [M][ldo]
    int64_t blob_floats, W, b, gamma, beta, jb_stride;
    int lda, lda2, K1, ldr, ldy, ldo, M, K, NO, N, act, row_tiles;  // act: 0 none (+R), 1 lrelu(acc) (+R after), 2 lrelu(acc + R), 3 (acc + R) * lrelu'(Y), 4 acc * lrelu'(Y) + R
    float slope, eps;
    int cpw = 1;        // panel kernel: column chunks (of BN outputs) per workgroup
#ifdef PFM_TF_DIAG
    int diag = 0;       // tests/diag/tf_panel_stamps.py: this launch records s_memtime stamps of wave 0 of every 8th workgroup
#endif
    int pre_act = 0;    // 1: the input rows pass through LeakyReLU(slope) on their way into LDS (MDMA: fc0(act(x)), mdma.py:65)
    int bn = BN;        // outputs per workgroup: BN (two 16-output operands per wave) or 64 (one): see launch_linear_kernel
};

// row statistics of a BM-row tile: 16 lanes per row, two-pass (mean, then centred sum of squares)
template <int NI, int ROWS>
__device__ __forceinline__ void ln_stats_tile(const float* __restrict__ A, int lda, int M, int row0, float eps,
                                              float* __restrict__ stat, int tid) {
    constexpr int K = 64 * NI;
    const int lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    constexpr int RPW = ROWS / (LT / 64);  // rows per wave
#pragma unroll 1
    for (int pass = 0; pass < RPW / 4; ++pass) {
        const int r = RPW * w + 4 * pass + q;
        const int row = min(row0 + r, M - 1);
        const float* ap = A + (int64_t)row * lda + 4 * pl;
        f32x4 v[NI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(ap + 64 * i);
            s += hsum4(v[i]);
        }
        const float mean = row_sum16(s) / (float)K;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 dl = v[i] - mean;
            ss += hsum4(dl * dl);
        }
        const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)K + eps);
        if (pl == 0) {
            stat[2 * r] = mean;
            stat[2 * r + 1] = rstd;
        }
    }
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (round-robin), so the column chunks of
// one row tile -- which re-read the same activation rows -- are given ids that are congruent mod 8 and share an L2.
__device__ __forceinline__ bool tile_of_block(int bid, int row_tiles, int nchunk, int& rt, int& ch) {
    const int xcd = bid & 7, slot = bid >> 3;
    rt = (slot / nchunk) * 8 + xcd;
    ch = slot % nchunk;
    return rt < row_tiles;
}

// NI = K / 64 of the LayerNorm prologue (0: no LayerNorm); TPW = 16-row tiles per wave: the workgroup's row tile is
// RB = 16 TPW rows (64, or 32 when 64-row tiles would leave the last round of workgroups mostly empty)
// X3: products on the fp16 matrix pipe with split operands (pfm_common.h: x = hi + lo 2^-11): per 64-wide step the
// activation slice is published as two fp16 planes (rows of 72 halfs: conflict-free ds_read_b128 of 8 consecutive k),
// the weights are split per wave from the same fp32 MFMA_AK block read in k8 order, three v_mfma_f32_16x16x32_f16
// per (output tile, row tile, 32 k) accumulate into a main and a correction accumulator.
constexpr int X3ROW = 72;  // halfs per LDS row of an X3 slice (64 + 8 pad)
// NS = 16-row A operands (output tiles) per wave: 2 in a full 128-output column chunk.  A last chunk of <= 64 outputs (NO = 320:
// 128 + 128 + 64) runs the NS = 1 body, every wave one operand, so that workgroup costs half a full one instead of idling two
// of its waves; the two bodies are separate code paths behind one workgroup-uniform branch (separate register allocations).
// MODE 2 (BF16; PFM_*_F_BF16: what trainer.precision = "bf16-mixed" means for the Linears of these models in the reference -- autocast
// around nn.Linear): operands rounded to bf16 (round-to-nearest-even), ONE v_mfma_f32_16x16x32_bf16 per (output tile, row tile, 32 k),
// fp32 accumulate, fp32 activations in HBM.  Same staging as X3 with the hi plane only.
template <int NI, int TPW, int MODE, int NS>
__device__ __forceinline__ void tf_linear_body(LinArgs& a, float* __restrict__ lds, int rt, int ch, int ks) {
    constexpr bool X3 = MODE == 1, BF = MODE == 2, HALF = MODE != 0;  // HALF: 16-bit operand planes in LDS, weights read in k8 order
    constexpr bool LN = NI > 0;
    constexpr int RB = 16 * TPW, SI = RB / 16;  // SI: float4 staged per thread and 64-wide step
    float* const tile = lds;             // two RB x 64 slices, 16-byte slots XOR-swizzled with (row & 15)
    float* const stat = lds + (HALF ? RB * X3ROW * 2 : RB * 128);  // RB x (mean, rstd), behind the slices
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the weight-block offsets below become SGPR offsets of the buffer loads
    const int row0 = rt * RB;
    const int ob = ch * a.bn + 16 * NS * w;  // this wave: outputs [ob, ob + 16 NS) as NS 16-row A operands
    const blob_rsrc rs = make_blob_rsrc(a.blob, a.blob_floats);
    // NO is a multiple of 32: in a partial last column chunk the waves past NO only help staging
    const bool active = ob < a.NO;

    if (LN) {
        ln_stats_tile<LN ? NI : 2, RB>(a.A, a.lda, a.M, row0, a.eps, stat, tid);
        __syncthreads();
    }

    // accumulators start from the bias (+ the jet-bias row of the particle's jet)
    f32x4 acc[NS][TPW];
    f32x4 cor[NS][X3 ? TPW : 1];  // X3: correction accumulators (hi.wlo + lo.whi), scaled by 2^-11 at the end
    if (X3) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < (X3 ? TPW : 1); ++t) cor[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = ob + 16 * s + 4 * q;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (a.ksplit > 1 || !active) {
                acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if (a.jb) {
                const int row = min(row0 + 16 * t + pl, a.M - 1);
                const int jet = a.rowjet ? a.rowjet[row] : row / a.N;
                acc[s][t] = *reinterpret_cast<const f32x4*>(a.jb + (int64_t)jet * a.jb_stride + o);
            } else if (a.b >= 0) {
                acc[s][t] = *reinterpret_cast<const f32x4*>(a.blob + a.b + o);
            } else {
                acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // Software pipeline over K in steps of 64: the weights (A operand: 2 x 4 float4 per lane) and the activation slice
    // (4 float4 per lane) of step s+1 are requested right after step s's slice has been published to LDS, so their
    // L2 / HBM latency runs under step s's 128 MFMAs.  Two LDS slices alternate: one barrier per step.  The
    // LayerNorm transform is applied on the way into LDS.
    const int sc4 = tid & 15, sr = tid >> 4;  // staging: 16-byte column slot (16 per 64-column step), first row
    const int nst_all = a.K >> 6;
    const int st0 = a.ksplit > 1 ? ks * (nst_all / a.ksplit) : 0;       // K / 64 is a multiple of ksplit (host)
    const int nst = a.ksplit > 1 ? st0 + nst_all / a.ksplit : nst_all;  // one past this workgroup's last step
    // gb: the LayerNorm gamma / beta columns of the step (requested with its operands, one step ahead like them: read at the head of
    // the step they exposed an L2 round trip per step)
    auto request = [&](f32x4 (&af)[NS][4], f32x4 (&st)[SI], f32x4 (&gb)[2], int step) {
        if (active) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t base = a.W + ((int64_t)((ob >> 4) + s) * nst_all + step) * 1024;
                if (HALF) {  // k8 order: af[2 kp + h] = k-tile 2 kp + (q >> 1), lane (pl, 2 (q & 1) + h)  (see load_afrag_k8)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            af[s][2 * kp + h] = bload4(rs, base + (2 * kp + (q >> 1)) * 256, (pl + 16 * (2 * (q & 1) + h)) * 16);
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) af[s][kt] = bload4(rs, base + kt * 256, lane * 16);
                }
            }
        }
        const int col = 64 * step + 4 * sc4;
        const bool seg2 = a.A2 != nullptr && 64 * step >= a.K1;  // wave-uniform
        const float* src = seg2 ? a.A2 + (col - a.K1) : a.A + col;
        const int ld = seg2 ? a.lda2 : a.lda;
#pragma unroll
        for (int i = 0; i < SI; ++i) {
            const int row = min(row0 + sr + 16 * i, a.M - 1);
            st[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)row * ld);
        }
        if (LN) {
            gb[0] = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + col);
            gb[1] = *reinterpret_cast<const f32x4*>(a.blob + a.beta + col);
        }
    };
    auto step_fn = [&](f32x4 (&af)[NS][4], f32x4 (&st)[SI], f32x4 (&gb)[2], f32x4 (&afn)[NS][4], f32x4 (&stn)[SI], f32x4 (&gbn)[2], int step) {
        float* const buf = tile + (step & 1) * (HALF ? RB * X3ROW : RB * 64);  // X3: hi plane, lo plane RB * X3ROW halfs later (BF16: unused)
        if (LN) {
            const f32x4 g4 = gb[0], b4 = gb[1];
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int r = sr + 16 * i;
                const float mean = stat[2 * r], rstd = stat[2 * r + 1];
                st[i] = (st[i] - mean) * rstd * g4 + b4;
            }
        }
        if (a.pre_act) {
#pragma unroll
            for (int i = 0; i < SI; ++i) st[i] = lrelu4(st[i], a.slope);
        }
        if (X3) {
            _Float16* hb = reinterpret_cast<_Float16*>(buf);
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                h4 hi, lo;
                x3_split(st[i], hi, lo);
                _Float16* dp = hb + (sr + 16 * i) * X3ROW + 4 * sc4;
                *reinterpret_cast<h4*>(dp) = hi;
                *reinterpret_cast<h4*>(dp + RB * X3ROW) = lo;
            }
        } else if (BF) {
            short* hb = reinterpret_cast<short*>(buf);
#pragma unroll
            for (int i = 0; i < SI; ++i) *reinterpret_cast<s16x4*>(hb + (sr + 16 * i) * X3ROW + 4 * sc4) = pack_bf16(st[i]);
        } else {
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int r = sr + 16 * i;
                *reinterpret_cast<f32x4*>(buf + r * 64 + ((sc4 ^ (r & 15)) << 2)) = st[i];
            }
        }
        __syncthreads();  // also orders this write after every wave's reads of the same slice two steps ago
        if (step + 1 < nst) request(afn, stn, gbn, step + 1);
        if (!active) return;
        if (X3) {
            const _Float16* hb = reinterpret_cast<const _Float16*>(buf);
            h8 wh[NS][2], wl[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) x3_split8(af[s][2 * kp], af[s][2 * kp + 1], wh[s][kp], wl[s][kp]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    const _Float16* bp = hb + (16 * t + pl) * X3ROW + 32 * kp + 8 * q;
                    const h8 bh = *reinterpret_cast<const h8*>(bp);
                    const h8 bl = *reinterpret_cast<const h8*>(bp + RB * X3ROW);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][kp], bh, acc[s][t], 0, 0, 0);
                        cor[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][kp], bl, cor[s][t], 0, 0, 0);
                        cor[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][kp], bh, cor[s][t], 0, 0, 0);
                    }
                }
            }
            return;
        }
        if (BF) {
            const short* hb = reinterpret_cast<const short*>(buf);
            bf16x8 wb[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) wb[s][kp] = pack_bf16x8(af[s][2 * kp], af[s][2 * kp + 1]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hb + (16 * t + pl) * X3ROW + 32 * kp + 8 * q);
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s][kp], bh, acc[s][t], 0, 0, 0);
                }
            }
            return;
        }
        // B fragments (two 16-row tiles x 16 k per ds_read_b128 pair) are read one item ahead of the 16 MFMAs that use them
        constexpr int NIT = (TPW / 2) * 4;
        auto bfrag = [&](f32x4 (&B)[2], int it) {
            const int tp = it >> 2, kt = it & 3;
            const float* b0p = buf + (32 * tp + pl) * 64 + (((4 * kt + q) ^ pl) << 2);
            B[0] = *reinterpret_cast<const f32x4*>(b0p);
            B[1] = *reinterpret_cast<const f32x4*>(b0p + 16 * 64);
        };
        f32x4 Bq[2][2];
        bfrag(Bq[0], 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) bfrag(Bq[(it + 1) & 1], it + 1);
            const int tp = it >> 2, kt = it & 3;
            const f32x4 B0 = Bq[it & 1][0], B1 = Bq[it & 1][1];
#define PFM_TF_STEP(c)                                                                                                    \
    acc[0][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B0.c, acc[0][2 * tp], 0, 0, 0);                    \
    if constexpr (NS == 2)                                                                                                \
        acc[NS - 1][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NS - 1][kt].c, B0.c, acc[NS - 1][2 * tp], 0, 0, 0); \
    acc[0][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B1.c, acc[0][2 * tp + 1], 0, 0, 0);            \
    if constexpr (NS == 2)                                                                                                \
        acc[NS - 1][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NS - 1][kt].c, B1.c, acc[NS - 1][2 * tp + 1], 0, 0, 0);
            PFM_TF_STEP(x) PFM_TF_STEP(y) PFM_TF_STEP(z) PFM_TF_STEP(w)
#undef PFM_TF_STEP
        }
    };
    {
        f32x4 afA[NS][4], afB[NS][4], stA[SI], stB[SI], gbA[2], gbB[2];
        request(afA, stA, gbA, st0);
        int step = st0;
#pragma unroll 1
        for (; step + 1 < nst; step += 2) {
            step_fn(afA, stA, gbA, afB, stB, gbB, step);
            step_fn(afB, stB, gbB, afA, stA, gbA, step + 1);
        }
        if (step < nst) step_fn(afA, stA, gbA, afB, stB, gbB, step);  // odd number of 64-wide steps
    }

    // epilogue: lane (particle pl of tile t, q) holds 4 consecutive outputs
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = ob + 16 * s + 4 * q;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int row = row0 + 16 * t + pl;
            if (!active) continue;
            if (X3) acc[s][t] += cor[s][X3 ? t : 0] * X3_DN;
            if (row < a.M && a.ksplit > 1) {
                *reinterpret_cast<f32x4*>(a.part + ((int64_t)ks * a.M + row) * a.NO + o) = acc[s][t];
            } else if (row < a.M) {
                f32x4 v = acc[s][t];
                if (a.act >= 3) {
                    if (a.act == 3 && a.R) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                    const f32x4 y = *reinterpret_cast<const f32x4*>(a.Y + (int64_t)row * a.ldy + o);
                    v.x *= y.x > 0.f ? 1.f : a.slope; v.y *= y.y > 0.f ? 1.f : a.slope;
                    v.z *= y.z > 0.f ? 1.f : a.slope; v.w *= y.w > 0.f ? 1.f : a.slope;
                    if (a.act == 4 && a.R) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                } else {
                    if (a.act == 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                    if (a.act) v = lrelu4(v, a.slope);
                    if (a.R && a.act != 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                }
                *reinterpret_cast<f32x4*>(a.out + (int64_t)row * a.ldo + o) = v;
            }
        }
    }
}

// B64: the launch runs 64-output workgroups only (LinArgs.bn = 64): an instantiation without the two-operand body, whose registers would
// otherwise set the kernel's occupancy
template <int NI, int TPW = 4, int MODE = 0, bool B64 = false>  // MODE: 0 fp32 MFMA, 1 split-fp16 (X3), 2 bf16 operands
__global__ __launch_bounds__(LT, 2) void tf_linear_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int rt, ch;
    const int ks = a.ksplit > 1 ? blockIdx.x % a.ksplit : 0;
    if (!tile_of_block(a.ksplit > 1 ? blockIdx.x / a.ksplit : blockIdx.x, a.row_tiles, (a.NO + a.bn - 1) / a.bn, rt, ch)) return;
    if (a.m_dev) {
        a.M = *a.m_dev;
        if (rt * 16 * TPW >= a.M) return;
    }
    if constexpr (B64) {
        tf_linear_body<NI, TPW, MODE, 1>(a, lds, rt, ch, ks);
    } else {
        if (a.NO - ch * a.bn <= 64 || a.bn == 64) tf_linear_body<NI, TPW, MODE, 1>(a, lds, rt, ch, ks);
        else tf_linear_body<NI, TPW, MODE, 2>(a, lds, rt, ch, ks);
    }
}

#ifdef PFM_TF_DIAG
// diagnostics build only (tests/diag/tf_panel_stamps.py): [workgroup / 8][8] s_memtime stamps of wave 0
__device__ unsigned long long g_tf_stamps[512 * 8];
#define PFM_TF_STAMP(k)                                                                              \
    if (a.diag && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 512 && tid == 0)                      \
        g_tf_stamps[(blockIdx.x >> 3) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define PFM_TF_STAMP(k)
#endif
// ------------------------------------------------------------------------------------------------
// tf_linear_panel_kernel: the LayerNorm-Linear of launches with many rows (no split-K, no second input segment, NO a multiple of 128).
// The fp32 MFMA shares its issue port with the VALU (tests/diag/mfma_coissue.hip: the two add up, they do not overlap), so what the
// kernel above spends per column chunk on row statistics, the LayerNorm transform of every 64-wide step, staging addresses and barriers
// is matrix-pipe time.  Here a workgroup owns RB rows for ALL outputs: the rows are read once, normalised once (statistics from the
// registers that hold the row) and kept in LDS as the B operand of every step -- K / 64 slices of RB x 64, the swizzle of the kernel
// above -- and the loop over (chunk, step) is weight requests (buffer loads at scalar offsets, one step ahead), ds_read_b128 and MFMAs:
// no barrier after the staging one.  Same products in the same order as tf_linear_kernel: the results are bit-identical.
// ------------------------------------------------------------------------------------------------
template <int NI, int TPW>
__global__ __launch_bounds__(LT, TPW == 2 && NI <= 4 ? 3 : 2) void tf_linear_panel_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RB = 16 * TPW, K = 64 * NI, NS = 2;
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.m_dev) a.M = *a.m_dev;
    // a.cpw chunks per workgroup: blockIdx -> (row tile, chunk group), the groups of a row tile on consecutive ids
    const int nchunk_all = a.NO / BN, ngrp = (nchunk_all + a.cpw - 1) / a.cpw;
    const int row0 = (blockIdx.x / ngrp) * RB, c0 = (blockIdx.x % ngrp) * a.cpw;
    if (row0 >= a.M) return;
    PFM_TF_STAMP(0)
    const blob_rsrc rs = make_blob_rsrc(a.blob, a.blob_floats);

    // weights of (chunk c, step): this wave's two 16-output operands, four k-tiles of 16 each
    auto request = [&](f32x4 (&af)[NS][4], int c, int step) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t base = a.W + ((int64_t)(((c * BN + 32 * w) >> 4) + s) * NI + step) * 1024;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) af[s][kt] = bload4(rs, base + kt * 256, lane * 16);
        }
    };
    f32x4 afA[NS][4], afB[NS][4];
    request(afA, c0, 0);  // in flight during the staging

    // staging: 16 lanes per row (lane pl: columns 4 pl + 64 i), four rows per wave and pass; two-pass statistics as ln_stats_tile
    constexpr int RPW = RB / (LT / 64);
#pragma unroll 1
    for (int pass = 0; pass < RPW / 4; ++pass) {
        const int r = RPW * w + 4 * pass + q;
        const int row = min(row0 + r, a.M - 1);
        const float* ap = a.A + (int64_t)row * a.lda + 4 * pl;
        f32x4 v[NI];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(ap + 64 * i);
            sm += hsum4(v[i]);
        }
        const float mean = row_sum16(sm) / (float)K;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 dl = v[i] - mean;
            ss += hsum4(dl * dl);
        }
        const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)K + a.eps);
        float* const dst = lds + r * 64 + ((pl ^ (r & 15)) << 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 64 * i + 4 * pl);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.blob + a.beta + 64 * i + 4 * pl);
            f32x4 x = (v[i] - mean) * rstd * g4 + b4;
            if (a.pre_act) x = lrelu4(x, a.slope);
            *reinterpret_cast<f32x4*>(dst + i * (RB * 64)) = x;
        }
    }
    PFM_TF_STAMP(1)
    __syncthreads();
    PFM_TF_STAMP(2)

    f32x4 acc[NS][TPW];
    auto mma = [&](f32x4 (&af)[NS][4], int step) {
        const float* const buf = lds + step * (RB * 64);
        constexpr int NIT = (TPW / 2) * 4;
        auto bfrag = [&](f32x4 (&B)[2], int it) {
            const int tp = it >> 2, kt = it & 3;
            const float* b0p = buf + (32 * tp + pl) * 64 + (((4 * kt + q) ^ pl) << 2);
            B[0] = *reinterpret_cast<const f32x4*>(b0p);
            B[1] = *reinterpret_cast<const f32x4*>(b0p + 16 * 64);
        };
        f32x4 Bq[2][2];
        bfrag(Bq[0], 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) bfrag(Bq[(it + 1) & 1], it + 1);
            const int tp = it >> 2, kt = it & 3;
            const f32x4 B0 = Bq[it & 1][0], B1 = Bq[it & 1][1];
#define PFM_TF_STEP(c)                                                                                         \
    acc[0][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B0.c, acc[0][2 * tp], 0, 0, 0);         \
    acc[1][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B0.c, acc[1][2 * tp], 0, 0, 0);         \
    acc[0][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B1.c, acc[0][2 * tp + 1], 0, 0, 0); \
    acc[1][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B1.c, acc[1][2 * tp + 1], 0, 0, 0);
            PFM_TF_STEP(x) PFM_TF_STEP(y) PFM_TF_STEP(z) PFM_TF_STEP(w)
#undef PFM_TF_STEP
        }
    };

    const int nchunk = min(nchunk_all, c0 + a.cpw);
    int nsteps = NI;
    if constexpr (NI == 2) asm volatile("" : "+s"(nsteps));  // opaque: the step loop stays a loop (unrolled into the chunk loop it spilled)
#pragma unroll 1
    for (int c = c0; c < nchunk; ++c) {
        const int ob = c * BN + 32 * w;
        // accumulators start from the bias (+ the jet-bias row of the particle's jet)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if (a.jb) {
                    const int row = min(row0 + 16 * t + pl, a.M - 1);
                    const int jet = a.rowjet ? a.rowjet[row] : row / a.N;
                    acc[s][t] = *reinterpret_cast<const f32x4*>(a.jb + (int64_t)jet * a.jb_stride + o);
                } else if (a.b >= 0) {
                    acc[s][t] = *reinterpret_cast<const f32x4*>(a.blob + a.b + o);
                } else {
                    acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#ifdef PFM_TF_DIAG
        if (c == c0) {
            __builtin_amdgcn_s_waitcnt(0);  // start values and first weights have arrived
            PFM_TF_STAMP(3)
        }
#endif
#pragma unroll 1
        for (int step = 0; step < nsteps; step += 2) {
            request(afB, c, step + 1);
            mma(afA, step);
            if (step + 2 < nsteps) request(afA, c, step + 2);
            else if (c + 1 < nchunk) request(afA, c + 1, 0);
            mma(afB, step + 1);
        }
#ifdef PFM_TF_DIAG
        if (c == c0) { PFM_TF_STAMP(4) }
#endif
        // epilogue: lane (particle pl of tile t, q) holds 4 consecutive outputs.  (Requesting the residual rows and the next chunk's start
        // values under the last steps was measured: no gain, the 32 registers cost the third wave per SIMD.)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int row = row0 + 16 * t + pl;
                if (row >= a.M) continue;
                f32x4 v = acc[s][t];
                if (a.act >= 3) {
                    if (a.act == 3 && a.R) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                    const f32x4 y = *reinterpret_cast<const f32x4*>(a.Y + (int64_t)row * a.ldy + o);
                    v.x *= y.x > 0.f ? 1.f : a.slope; v.y *= y.y > 0.f ? 1.f : a.slope;
                    v.z *= y.z > 0.f ? 1.f : a.slope; v.w *= y.w > 0.f ? 1.f : a.slope;
                    if (a.act == 4 && a.R) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                } else {
                    if (a.act == 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                    if (a.act) v = lrelu4(v, a.slope);
                    if (a.R && a.act != 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
                }
                *reinterpret_cast<f32x4*>(a.out + (int64_t)row * a.ldo + o) = v;
            }
        }
#ifdef PFM_TF_DIAG
        if (c == c0) { PFM_TF_STAMP(5) }
#endif
    }
    PFM_TF_STAMP(6)
}

template <int NI, int TPW>
inline void launch_panel(const LinArgs& a, hipStream_t s) {
    constexpr int lds = NI * 16 * TPW * 64 * 4;
    if (lds > 64 * 1024) {
        static bool big = false;
        if (!big) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(tf_linear_panel_kernel<NI, TPW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            big = true;
        }
    }
    const int ngrp = (a.NO / BN + a.cpw - 1) / a.cpw;
    hipLaunchKernelGGL((tf_linear_panel_kernel<NI, TPW>), dim3((a.M + 16 * TPW - 1) / (16 * TPW) * ngrp), dim3(LT), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// tf_mlp_panel_kernel: the dense block  out = R + W2 LN2( lrelu( W1 LN1(A) + jet bias ) )  of a row tile in ONE workgroup (inference; K1 = 128,
// hidden 256: the cross-attention layers).  The two Linears of the block are row-local, so the hidden rows never leave the CU: stage 1 is the panel
// kernel above with its epilogue writing the activated hidden values into a second LDS panel (same slice / swizzle layout), the rows are normalised
// there in place (statistics with the staging's own reduction order, from the same floats the two-launch path reads back from memory), and stage 2
// multiplies from that panel.  Same products and sums in the same order as two tf_linear(_panel)_kernel launches: bit-identical, minus the hidden
// tensor's trip through HBM (2 x M x 256 floats per block) and one launch.
// ------------------------------------------------------------------------------------------------
struct MlpArgs {
    LinArgs l0;      // PRE: the Linear in front of the block, mid = R0 + W0 LN0(A0) + b0 (one 128-output chunk; also stored to l0.out: stage 2's residual)
    LinArgs l1, l2;  // l1: A, lda, M, m_dev, W, gamma, beta, jb / b, rowjet, N, jb_stride, slope, eps, K, NO (= hidden);  l2: W, b, gamma, beta, R, ldr, out, ldo, NO
};

// PRE (cross-attention: to.out in front of the block): stage 0 multiplies LN0(A0) from the X panel, its epilogue adds the residual, stores the rows
// (the block's own residual reads them back, same thread) and parks them raw in the H region; they are normalised from there into X for stage 1.
template <int NI1, int NI2, bool PRE>
__global__ __launch_bounds__(LT, 3) void tf_mlp_panel_kernel(MlpArgs m) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TPW = 2, RB = 32, NS = 2, RPW = RB / (LT / 64), NP = RPW / 4;
    LinArgs& a = m.l1;
    LinArgs& b = m.l2;
    float* const X = lds;                   // NI1 slices of RB x 64: LN1(A)
    float* const H = lds + NI1 * RB * 64;   // NI2 slices: the hidden rows
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.m_dev) a.M = *a.m_dev;
    const int row0 = blockIdx.x * RB;
    if (row0 >= a.M) return;
    const blob_rsrc rs = make_blob_rsrc(a.blob, a.blob_floats);
    auto request = [&](f32x4 (&af)[NS][4], int64_t W, int nst, int c, int step) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t base = W + ((int64_t)(((c * BN + 32 * w) >> 4) + s) * nst + step) * 1024;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) af[s][kt] = bload4(rs, base + kt * 256, lane * 16);
        }
    };
    f32x4 afA[NS][4], afB[NS][4];
    request(afA, PRE ? m.l0.W : a.W, NI1, 0, 0);

    // rows -> LN -> panel (global rows into X; later the hidden rows in place in H): 16 lanes per row, two-pass statistics as ln_stats_tile
    auto normalise = [&](auto load, float* dstp, auto NIc, int64_t gamma, int64_t beta, float eps) {
        constexpr int NI = decltype(NIc)::value;
#pragma unroll 1
        for (int pass = 0; pass < NP; ++pass) {
            const int r = RPW * w + 4 * pass + q;
            f32x4 v[NI];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                v[i] = load(r, i);
                sm += hsum4(v[i]);
            }
            const float mean = row_sum16(sm) / (float)(64 * NI);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x4 dl = v[i] - mean;
                ss += hsum4(dl * dl);
            }
            const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)(64 * NI) + eps);
            float* const dst = dstp + r * 64 + ((pl ^ (r & 15)) << 2);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.blob + gamma + 64 * i + 4 * pl);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.blob + beta + 64 * i + 4 * pl);
                *reinterpret_cast<f32x4*>(dst + i * (RB * 64)) = (v[i] - mean) * rstd * g4 + b4;
            }
        }
    };
    {
        const LinArgs& f = PRE ? m.l0 : a;  // the Linear that reads the global rows
        normalise([&](int r, int i) { return *reinterpret_cast<const f32x4*>(f.A + (int64_t)min(row0 + r, a.M - 1) * f.lda + 4 * pl + 64 * i); }, X,
                  std::integral_constant<int, NI1>{}, f.gamma, f.beta, a.eps);
    }
    __syncthreads();

    f32x4 acc[NS][TPW];
    auto mma = [&](f32x4 (&af)[NS][4], const float* buf) {
        constexpr int NIT = (TPW / 2) * 4;
        auto bfrag = [&](f32x4 (&B)[2], int it) {
            const int tp = it >> 2, kt = it & 3;
            const float* b0p = buf + (32 * tp + pl) * 64 + (((4 * kt + q) ^ pl) << 2);
            B[0] = *reinterpret_cast<const f32x4*>(b0p);
            B[1] = *reinterpret_cast<const f32x4*>(b0p + 16 * 64);
        };
        f32x4 Bq[2][2];
        bfrag(Bq[0], 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) bfrag(Bq[(it + 1) & 1], it + 1);
            const int tp = it >> 2, kt = it & 3;
            const f32x4 B0 = Bq[it & 1][0], B1 = Bq[it & 1][1];
#define PFM_TF_STEP(c)                                                                                         \
    acc[0][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B0.c, acc[0][2 * tp], 0, 0, 0);         \
    acc[1][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B0.c, acc[1][2 * tp], 0, 0, 0);         \
    acc[0][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B1.c, acc[0][2 * tp + 1], 0, 0, 0); \
    acc[1][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B1.c, acc[1][2 * tp + 1], 0, 0, 0);
            PFM_TF_STEP(x) PFM_TF_STEP(y) PFM_TF_STEP(z) PFM_TF_STEP(w)
#undef PFM_TF_STEP
        }
    };

    int n1 = NI1, n2 = NI2;
    asm volatile("" : "+s"(n1), "+s"(n2));  // the step loops stay loops
    // Residual rows are requested BEFORE the MFMA loop whose epilogue adds them (every workgroup of a launch starts at once and walks the same
    // phases: a load issued in an epilogue is a round trip to L2 / HBM with the matrix pipes of the whole CU idle), and the rows stage 0 makes
    // stay in registers for stage 2's residual instead of being read back (same thread, same values): cross-attention 894 -> 912 jets/s.
    f32x4 resv[NS][TPW];
    bool mid_kept = false;
    if constexpr (PRE) {
        // ---- stage 0: mid = R0 + W0 LN0(A0) + b0 (NI1 columns = one chunk), to l0.out and, raw, into the H region ----------------------------
        const LinArgs& z = m.l0;
        const int ob = 32 * w;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[s][t] = z.b >= 0 ? *reinterpret_cast<const f32x4*>(a.blob + z.b + ob + 16 * s + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (z.R) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    resv[s][t] = *reinterpret_cast<const f32x4*>(z.R + (int64_t)min(row0 + 16 * t + pl, a.M - 1) * z.ldr + ob + 16 * s + 4 * q);
        }
#pragma unroll 1
        for (int step = 0; step < n1; step += 2) {
            request(afB, z.W, NI1, 0, step + 1);
            mma(afA, X + step * (RB * 64));
            if (step + 2 < n1) request(afA, z.W, NI1, 0, step + 2);
            else request(afA, a.W, NI1, 0, 0);  // first weights of stage 1
            mma(afB, X + (step + 1) * (RB * 64));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int r = 16 * t + pl, row = min(row0 + r, a.M - 1);
                f32x4 v = acc[s][t];
                if (z.R) v += resv[s][t];
                if (row0 + r < a.M) *reinterpret_cast<f32x4*>(z.out + (int64_t)row * z.ldo + o) = v;
                resv[s][t] = v;
                *reinterpret_cast<f32x4*>(H + (o >> 6) * (RB * 64) + r * 64 + ((((o & 63) >> 2) ^ (r & 15)) << 2)) = v;
            }
        }
        __syncthreads();  // every wave is done with LN0(A0) in X and has parked its mid columns
        normalise([&](int r, int i) { return *reinterpret_cast<const f32x4*>(H + i * (RB * 64) + r * 64 + ((pl ^ (r & 15)) << 2)); }, X,
                  std::integral_constant<int, NI1>{}, a.gamma, a.beta, a.eps);
        __syncthreads();
        mid_kept = b.R == z.out && b.ldr == z.ldo && b.NO == BN;  // (one output chunk: stage 2's (s, t) are stage 0's)
    }
    // ---- stage 1: hidden = lrelu(W1 LN1(mid) + jet bias), into H ------------------------------------------------------------------------
    const int nc1 = a.NO / BN;
#pragma unroll 1
    for (int c = 0; c < nc1; ++c) {
        const int ob = c * BN + 32 * w;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if (a.jb) {
                    const int row = min(row0 + 16 * t + pl, a.M - 1);
                    const int jet = a.rowjet ? a.rowjet[row] : row / a.N;
                    acc[s][t] = *reinterpret_cast<const f32x4*>(a.jb + (int64_t)jet * a.jb_stride + o);
                } else if (a.b >= 0) {
                    acc[s][t] = *reinterpret_cast<const f32x4*>(a.blob + a.b + o);
                } else {
                    acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#pragma unroll 1
        for (int step = 0; step < n1; step += 2) {
            request(afB, a.W, NI1, c, step + 1);
            mma(afA, X + step * (RB * 64));
            if (step + 2 < n1) request(afA, a.W, NI1, c, step + 2);
            else if (c + 1 < nc1) request(afA, a.W, NI1, c + 1, 0);
            else request(afA, b.W, NI2, 0, 0);  // first weights of stage 2
            mma(afB, X + (step + 1) * (RB * 64));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;  // 4 consecutive hidden columns: slice o / 64, 16-byte slot (o % 64) / 4
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int r = 16 * t + pl;
                f32x4 v = acc[s][t];
                if (a.act) v = lrelu4(v, a.slope);
                *reinterpret_cast<f32x4*>(H + (o >> 6) * (RB * 64) + r * 64 + ((((o & 63) >> 2) ^ (r & 15)) << 2)) = v;
            }
        }
    }
    __syncthreads();
    // ---- LN2 over the hidden rows, in place ---------------------------------------------------------------------------------------------
    normalise([&](int r, int i) { return *reinterpret_cast<const f32x4*>(H + i * (RB * 64) + r * 64 + ((pl ^ (r & 15)) << 2)); }, H,
              std::integral_constant<int, NI2>{}, b.gamma, b.beta, a.eps);
    __syncthreads();
    // ---- stage 2: out = R + W2 LN2(hidden) + b2 -----------------------------------------------------------------------------------------
    const int nc2 = b.NO / BN;
#pragma unroll 1
    for (int c = 0; c < nc2; ++c) {
        const int ob = c * BN + 32 * w;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[s][t] = b.b >= 0 ? *reinterpret_cast<const f32x4*>(a.blob + b.b + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (b.R && !mid_kept) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    resv[s][t] = *reinterpret_cast<const f32x4*>(b.R + (int64_t)min(row0 + 16 * t + pl, a.M - 1) * b.ldr + ob + 16 * s + 4 * q);
        }
#pragma unroll 1
        for (int step = 0; step < n2; step += 2) {
            request(afB, b.W, NI2, c, step + 1);
            mma(afA, H + step * (RB * 64));
            if (step + 2 < n2) request(afA, b.W, NI2, c, step + 2);
            else if (c + 1 < nc2) request(afA, b.W, NI2, c + 1, 0);
            mma(afB, H + (step + 1) * (RB * 64));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int row = row0 + 16 * t + pl;
                if (row >= a.M) continue;
                f32x4 v = acc[s][t];
                if (b.R) v += resv[s][t];
                *reinterpret_cast<f32x4*>(b.out + (int64_t)row * b.ldo + o) = v;
            }
        }
    }
}

// the dense block in one launch where the panel conditions hold (launch_linear_kernel) and the shapes are the cross-attention ones; false: the caller
// launches the two Linears
inline bool launch_mlp_panel(const LinArgs& l1, const LinArgs& l2, int cus, hipStream_t s, const LinArgs* l0 = nullptr) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("PFM_TF_MLP");  // diagnostics only (tests/diag): 0 = separate launches, 1 = the block alone, 2 (default) = with the Linear in front
        const char* p = getenv("PFM_TF_PANEL");
        on = (p && !atoi(p)) ? 0 : (e ? atoi(e) : 2);
    }
    if (!on || (l0 && on < 2) || l1.K != 128 || l1.NO != 256 || l2.K != 256 || l2.NO % BN != 0 || l1.gamma < 0 || l2.gamma < 0 || l1.A2 || l1.pre_act ||
        l2.pre_act || l1.act != 1 || l1.R || l2.act != 0 || l2.jb || (int64_t)(l1.M + 31) / 32 < 2 * (int64_t)cus)
        return false;
    if (l0 && (l0->K != 128 || l0->NO != 128 || l0->gamma < 0 || l0->act != 0 || l0->jb || l0->A2 || l0->pre_act || l0->out != l1.A || l0->ldo != l1.lda))
        return false;
    MlpArgs m{l0 ? *l0 : l1, l1, l2};
    if (l0) hipLaunchKernelGGL((tf_mlp_panel_kernel<2, 4, true>), dim3((l1.M + 31) / 32), dim3(LT), (2 + 4) * 32 * 64 * 4, s, m);
    else hipLaunchKernelGGL((tf_mlp_panel_kernel<2, 4, false>), dim3((l1.M + 31) / 32), dim3(LT), (2 + 4) * 32 * 64 * 4, s, m);
    return true;
}

// ------------------------------------------------------------------------------------------------
// tf_linear_panel2_kernel: TWO plain LayerNorm-Linears of the same rows (own LayerNorm, weights, bias and output each; no jet bias, residual or
// activation) in one launch: the rows are read and their statistics taken once, the two affine transforms fill two LDS panels, and the workgroup walks
// the chunks of the first Linear, then of the second.  Cross-attention: `from.kv` and `to.q` of a layer pair both read the particle rows.  Same
// arithmetic per output as two tf_linear_panel_kernel launches (bit-identical).
// ------------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(LT, 3) void tf_linear_panel2_kernel(MlpArgs m) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TPW = 2, RB = 32, K = 64 * NI, NS = 2, RPW = RB / (LT / 64), NP = RPW / 4;
    LinArgs& a = m.l1;
    LinArgs& b = m.l2;
    float* const X1 = lds;
    float* const X2 = lds + NI * RB * 64;
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.m_dev) a.M = *a.m_dev;
    const int row0 = blockIdx.x * RB;
    if (row0 >= a.M) return;
    const blob_rsrc rs = make_blob_rsrc(a.blob, a.blob_floats);
    auto request = [&](f32x4 (&af)[NS][4], int64_t W, int c, int step) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t base = W + ((int64_t)(((c * BN + 32 * w) >> 4) + s) * NI + step) * 1024;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) af[s][kt] = bload4(rs, base + kt * 256, lane * 16);
        }
    };
    f32x4 afA[NS][4], afB[NS][4];
    request(afA, a.W, 0, 0);
#pragma unroll 1
    for (int pass = 0; pass < NP; ++pass) {
        const int r = RPW * w + 4 * pass + q;
        const int row = min(row0 + r, a.M - 1);
        const float* ap = a.A + (int64_t)row * a.lda + 4 * pl;
        f32x4 v[NI];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(ap + 64 * i);
            sm += hsum4(v[i]);
        }
        const float mean = row_sum16(sm) / (float)K;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 dl = v[i] - mean;
            ss += hsum4(dl * dl);
        }
        const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)K + a.eps);
        const int so = r * 64 + ((pl ^ (r & 15)) << 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 64 * i + 4 * pl);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.blob + a.beta + 64 * i + 4 * pl);
            const f32x4 g2 = *reinterpret_cast<const f32x4*>(a.blob + b.gamma + 64 * i + 4 * pl);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.blob + b.beta + 64 * i + 4 * pl);
            *reinterpret_cast<f32x4*>(X1 + so + i * (RB * 64)) = (v[i] - mean) * rstd * g1 + b1;
            *reinterpret_cast<f32x4*>(X2 + so + i * (RB * 64)) = (v[i] - mean) * rstd * g2 + b2;
        }
    }
    __syncthreads();

    f32x4 acc[NS][TPW];
    auto mma = [&](f32x4 (&af)[NS][4], const float* buf) {
        constexpr int NIT = (TPW / 2) * 4;
        auto bfrag = [&](f32x4 (&B)[2], int it) {
            const int tp = it >> 2, kt = it & 3;
            const float* b0p = buf + (32 * tp + pl) * 64 + (((4 * kt + q) ^ pl) << 2);
            B[0] = *reinterpret_cast<const f32x4*>(b0p);
            B[1] = *reinterpret_cast<const f32x4*>(b0p + 16 * 64);
        };
        f32x4 Bq[2][2];
        bfrag(Bq[0], 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) bfrag(Bq[(it + 1) & 1], it + 1);
            const int tp = it >> 2, kt = it & 3;
            const f32x4 B0 = Bq[it & 1][0], B1 = Bq[it & 1][1];
#define PFM_TF_STEP(c)                                                                                         \
    acc[0][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B0.c, acc[0][2 * tp], 0, 0, 0);         \
    acc[1][2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B0.c, acc[1][2 * tp], 0, 0, 0);         \
    acc[0][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][kt].c, B1.c, acc[0][2 * tp + 1], 0, 0, 0); \
    acc[1][2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][kt].c, B1.c, acc[1][2 * tp + 1], 0, 0, 0);
            PFM_TF_STEP(x) PFM_TF_STEP(y) PFM_TF_STEP(z) PFM_TF_STEP(w)
#undef PFM_TF_STEP
        }
    };
    const int nc1 = a.NO / BN, nc = nc1 + b.NO / BN;
    int nsteps = NI;
    asm volatile("" : "+s"(nsteps));  // the step loop stays a loop
#pragma unroll 1
    for (int cc = 0; cc < nc; ++cc) {
        const bool second = cc >= nc1;  // workgroup-uniform
        const LinArgs& L = second ? b : a;
        const float* const X = second ? X2 : X1;
        const int c = second ? cc - nc1 : cc;
        const int ob = c * BN + 32 * w;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[s][t] = L.b >= 0 ? *reinterpret_cast<const f32x4*>(a.blob + L.b + ob + 16 * s + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int step = 0; step < nsteps; step += 2) {
            request(afB, L.W, c, step + 1);
            mma(afA, X + step * (RB * 64));
            if (step + 2 < nsteps) request(afA, L.W, c, step + 2);
            else if (cc + 1 < nc) request(afA, cc + 1 >= nc1 ? b.W : a.W, cc + 1 >= nc1 ? cc + 1 - nc1 : cc + 1, 0);
            mma(afB, X + (step + 1) * (RB * 64));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = ob + 16 * s + 4 * q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int row = row0 + 16 * t + pl;
                if (row < a.M) *reinterpret_cast<f32x4*>(L.out + (int64_t)row * L.ldo + o) = acc[s][t];
            }
        }
    }
}

// two plain LayerNorm-Linears of the same rows in one launch where the panel conditions hold; false: the caller launches them one by one
inline bool launch_panel2(const LinArgs& l1, const LinArgs& l2, int cus, hipStream_t s) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("PFM_TF_PANEL2");  // diagnostics only (tests/diag): 0 = two launches
        const char* p = getenv("PFM_TF_PANEL");
        on = (p && !atoi(p)) ? 0 : (e ? atoi(e) : 1);
    }
    auto plain = [](const LinArgs& l) { return l.gamma >= 0 && !l.jb && !l.R && !l.act && !l.pre_act && !l.A2 && l.ksplit == 1 && l.NO % BN == 0; };
    if (!on || !plain(l1) || !plain(l2) || l1.K != 128 || l2.K != 128 || l1.A != l2.A || l1.lda != l2.lda || l1.M != l2.M || l1.m_dev != l2.m_dev ||
        (int64_t)(l1.M + 31) / 32 < 2 * (int64_t)cus)
        return false;
    MlpArgs m{l1, l1, l2};
    hipLaunchKernelGGL((tf_linear_panel2_kernel<2>), dim3((l1.M + 31) / 32), dim3(LT), 2 * 2 * 32 * 64 * 4, s, m);
    return true;
}

// Row tile of a Linear launch: 32, 64 or 128 rows per workgroup, whichever gives the shortest schedule on this GPU's
// 2 x CU workgroup slots.  Work per round ~ rows x cost-per-row of the tile: a 128-row tile reads every weight block
// from L2 once per 128 rows instead of once per 64 (the L2 -> register weight stream is what limits the 64-row kernel),
// a 32-row tile fills the tail of launches that are only a little over a whole number of rounds.
inline int pick_row_tile(int64_t M, int chunks, int cus, bool allow128, bool b64 = false) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("PFM_TF_ROWTILE");  // diagnostics only (tests/diag): 32, 64 or 128
        forced = e ? atoi(e) : 0;
    }
    if (forced == 32 || forced == 64 || (forced == 128 && allow128)) return forced;
    // 64-output workgroups: 32 rows (84 VGPRs: five workgroups per CU) measured ahead of 64 (113: four) -- cfg 5, 256 jets x 100 steps:
    // 441 against 473 ms one call alone, 698 against 689 jets/s with three in flight
    if (b64) return 32;
    const int64_t slots = 2 * (int64_t)cus;
    auto rounds = [&](int rb) { return ((M + rb - 1) / rb * chunks + slots - 1) / slots; };
    static double f128 = 0.0;
    if (f128 == 0.0) {
        const char* e = getenv("PFM_TF_F128");
        f128 = e ? atof(e) : 0.85;
    }
    const double t32 = 32.0 * rounds(32), t64 = 64.0 * rounds(64), t128 = 128.0 * f128 * rounds(128);
    int rb = t32 < t64 ? 32 : 64;
    if (allow128 && t128 < (rb == 32 ? t32 : t64)) rb = 128;
    return rb;
}

// launches tf_linear_kernel for `a` (everything but row_tiles filled in); ni = K / 64 of the LayerNorm prologue or 0
// mode: 0 fp32 MFMA, 1 split-fp16 operands (X3), 2 bf16 operands (lin_mode() of the descriptor flags)
inline int launch_linear_kernel(LinArgs& a, int ni, int mode, int cus, hipStream_t s) {
    const bool x3 = mode != 0;  // 16-bit operand planes: no row-panel kernel, no 128-row tile, the X3 LDS layout
    const int chunks = (a.NO + BN - 1) / BN;
    {   // the row-panel kernel where a launch has the rows to fill the GPU with 32-row workgroups
        static int panel = -1;
        if (panel < 0) {
            const char* e = getenv("PFM_TF_PANEL");  // diagnostics only (tests/diag): 0 off, 32 / 64 force the row tile
            panel = e ? atoi(e) : 1;
        }
        if (panel && ni >= 2 && ni <= 6 && (ni & 1) == 0 && !x3 && a.ksplit == 1 && !a.A2 && a.NO % BN == 0 &&
            (int64_t)(a.M + 31) / 32 >= 2 * (int64_t)cus) {
            const bool r64 = panel == 64;
            // chunk groups per row tile: the fewest (dividing the chunk count) that give the launch two rounds of 3 workgroups per CU --
            // fewer, longer workgroups all stage their rows at once and leave the matrix pipes idle meanwhile
            const int64_t tiles = (a.M + (r64 ? 63 : 31)) / (r64 ? 64 : 32);
            int grp = 1;
            while (grp < chunks && (chunks % grp != 0 || tiles * grp < 6 * (int64_t)cus)) ++grp;
            static int cpw = -1;
            if (cpw < 0) { const char* e = getenv("PFM_TF_CPW"); cpw = e ? atoi(e) : 0; }
            a.cpw = cpw > 0 ? cpw : chunks / grp;
#ifdef PFM_TF_DIAG
            { const char* e = getenv("PFM_TF_DIAG_NO"); a.diag = e && atoi(e) == a.NO; }
#endif
            switch (ni) {
                case 2: r64 ? launch_panel<2, 4>(a, s) : launch_panel<2, 2>(a, s); break;
                case 4: r64 ? launch_panel<4, 4>(a, s) : launch_panel<4, 2>(a, s); break;
                default: r64 ? launch_panel<6, 4>(a, s) : launch_panel<6, 2>(a, s); break;
            }
            return 0;
        }
    }
    static int bn_env = -1;
    if (bn_env < 0) { const char* e = getenv("PFM_TF_BN"); bn_env = e ? atoi(e) : 0; }  // diagnostics only (tests/diag): 64 or 128
    if (bn_env == 64 || bn_env == BN) a.bn = bn_env;
    const int chunks_bn = (a.NO + a.bn - 1) / a.bn;
    const int rb = pick_row_tile(a.M, chunks_bn * a.ksplit, cus, !x3 && a.bn == BN, a.bn == 64 && ni == 0);
    a.row_tiles = (a.M + rb - 1) / rb;
    const int grid = ((a.row_tiles + 7) / 8) * 8 * chunks_bn * a.ksplit;
    const size_t lds = x3 ? (size_t)rb * X3ROW * 2 * 2 * 2 + 2 * rb * sizeof(float) : (size_t)(rb * 128 + 2 * rb) * sizeof(float);
#define PFM_LAUNCH_LIN(NI)                                                                                  \
    if (mode == 2) {                                                                                        \
        if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<NI, 2, 2>), dim3(grid), dim3(LT), lds, s, a);    \
        else hipLaunchKernelGGL((tf_linear_kernel<NI, 4, 2>), dim3(grid), dim3(LT), lds, s, a);             \
    } else if (x3) {                                                                                        \
        if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<NI, 2, 1>), dim3(grid), dim3(LT), lds, s, a);    \
        else hipLaunchKernelGGL((tf_linear_kernel<NI, 4, 1>), dim3(grid), dim3(LT), lds, s, a);             \
    } else if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<NI, 2>), dim3(grid), dim3(LT), lds, s, a);    \
    else if (rb == 64) hipLaunchKernelGGL((tf_linear_kernel<NI, 4>), dim3(grid), dim3(LT), lds, s, a);      \
    else {                                                                                                  \
        static bool big = false; /* 65 KB of dynamic LDS: opt in once per instantiation */                  \
        if (!big) {                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void*>(tf_linear_kernel<NI, 8>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)((128 * 128 + 256) * 4));  \
            big = true;                                                                                     \
        }                                                                                                   \
        hipLaunchKernelGGL((tf_linear_kernel<NI, 8>), dim3(grid), dim3(LT), lds, s, a);                     \
    }
    if (a.bn == 64 && ni == 0 && rb <= 64) {  // (the row-matrix EPiC path's particle Linears)
        if (mode == 2) {
            if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<0, 2, 2, true>), dim3(grid), dim3(LT), lds, s, a);
            else hipLaunchKernelGGL((tf_linear_kernel<0, 4, 2, true>), dim3(grid), dim3(LT), lds, s, a);
        } else if (x3) {
            if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<0, 2, 1, true>), dim3(grid), dim3(LT), lds, s, a);
            else hipLaunchKernelGGL((tf_linear_kernel<0, 4, 1, true>), dim3(grid), dim3(LT), lds, s, a);
        } else if (rb == 32) hipLaunchKernelGGL((tf_linear_kernel<0, 2, 0, true>), dim3(grid), dim3(LT), lds, s, a);
        else hipLaunchKernelGGL((tf_linear_kernel<0, 4, 0, true>), dim3(grid), dim3(LT), lds, s, a);
        return 0;
    }
    switch (ni) {
        case 0: PFM_LAUNCH_LIN(0) break;
        case 2: PFM_LAUNCH_LIN(2) break;
        case 4: PFM_LAUNCH_LIN(4) break;
        case 6: PFM_LAUNCH_LIN(6) break;
        case 8: PFM_LAUNCH_LIN(8) break;
        default: return -1;
    }
#undef PFM_LAUNCH_LIN
    return 0;
}

// split-K epilogue: out = epi( sum_ks part[ks] + b ) with the same activation / residual modes as the Linear kernel
static __global__ __launch_bounds__(256) void tf_splitk_kernel(LinArgs a) {
    const int nc4 = a.NO >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)a.M * nc4) return;
    const int row = (int)(i / nc4), o = 4 * (int)(i - (int64_t)row * nc4);
    f32x4 v = a.b >= 0 ? *reinterpret_cast<const f32x4*>(a.blob + a.b + o) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(a.part + ((int64_t)k * a.M + row) * a.NO + o);
    if (a.act == 3) {
        if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
        const f32x4 y = *reinterpret_cast<const f32x4*>(a.Y + (int64_t)row * a.ldy + o);
        v.x *= y.x > 0.f ? 1.f : a.slope; v.y *= y.y > 0.f ? 1.f : a.slope;
        v.z *= y.z > 0.f ? 1.f : a.slope; v.w *= y.w > 0.f ? 1.f : a.slope;
    } else {
        if (a.act == 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
        if (a.act) v = lrelu4(v, a.slope);
        if (a.R && a.act != 2) v += *reinterpret_cast<const f32x4*>(a.R + (int64_t)row * a.ldr + o);
    }
    *reinterpret_cast<f32x4*>(a.out + (int64_t)row * a.ldo + o) = v;
}

// ------------------------------------------------------------------------------------------------
// Attention: softmax(Q K^T / sqrt(hd) + keymask) V for one (jet, head); hd = 16
// ------------------------------------------------------------------------------------------------
// LDS strides of the forward attention's two ds_read_b128 operands.  A b128 read is served in four 16-lane groups -- {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- of 16-byte slots (address / 16) mod 16.  Lane (pl, q) reads slot
// (s * pl + q) mod 16 with s = stride / 4: conflict-free for s = 2, 6, 10, 14 (the group's q = 0 lanes land on even slots, its q = 1 lanes
// on odd ones).  K rows: 24 floats (s = 6); V^T rows: np + 8 (s = 2 or 10).  The strides 20 / np + 4 of rounds 1-2 were chosen for
// contiguous 16-lane groups and cost `SQ_LDS_BANK_CONFLICT` = 59 % of the kernel's busy cycles.
constexpr int KROW = 24;
constexpr int VPAD = 8;

__host__ __device__ inline int attn_np16(int N) { return (N + 15) & ~15; }
__host__ __device__ inline int attn_np32(int N) { return (N + 31) & ~31; }  // the forward walks key tiles in pairs
__host__ __device__ inline int attn_lds_floats(int N) {
    const int np = attn_np32(N);
    return np * KROW + HD * (np + VPAD) + np;
}

#define PFM_MFMA4(acc, A, B)                                          \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.x, B.x, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.y, B.y, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.z, B.z, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w, B.w, acc, 0, 0, 0);
// two independent accumulator chains, alternated instruction by instruction (a dependent MFMA stalls the pipe)
#define PFM_MFMA4x2(acc0, A0, B0, acc1, A1, B1)                          \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.x, B0.x, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.x, B1.x, acc1, 0, 0, 0); \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.y, B0.y, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.y, B1.y, acc1, 0, 0, 0); \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.z, B0.z, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.z, B1.z, acc1, 0, 0, 0); \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.w, B0.w, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.w, B1.w, acc1, 0, 0, 0);

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int MAXKT>
__global__ __launch_bounds__(256, 2) void tf_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                         float* __restrict__ out, int N, int D, int heads,
                                                         const int* __restrict__ off = nullptr, const int* __restrict__ order = nullptr) {
    static_assert(MAXKT % 2 == 0, "key tiles are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int slot = blockIdx.x / heads, h = blockIdx.x - slot * heads;
    const int jet = order ? order[slot] : slot;  // longest jets first (rows_rank_kernel)
    // compacted rows (off != nullptr): the jet owns rows [off[jet], off[jet + 1]), all valid, and N becomes its multiplicity
    int64_t row_base = (int64_t)jet * N;
    if (off) {
        row_base = off[jet];
        N = off[jet + 1] - off[jet];
        mask = nullptr;
        if (N == 0) return;
    }
    const int np = attn_np32(N), nkt = np >> 4, npv = np + VPAD, nqt = attn_np16(N) >> 4;
    float* const Ks = lds;              // [np][KROW]
    float* const Vt = Ks + np * KROW;   // [HD][npv]
    float* const mb = Vt + HD * npv;    // [np]: 0 for a valid key, -inf for a padded one
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    const int ld = 3 * D;
    const float* base = qkv + row_base * ld + h * HD;

    for (int idx = tid; idx < np * 4; idx += 256) {
        const int key = idx >> 2, part = idx & 3;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (key < N) {
            kv = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + D + 4 * part);
            vv = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + 2 * D + 4 * part);
        }
        *reinterpret_cast<f32x4*>(Ks + key * KROW + 4 * part) = kv;
        Vt[(4 * part + 0) * npv + key] = vv.x;
        Vt[(4 * part + 1) * npv + key] = vv.y;
        Vt[(4 * part + 2) * npv + key] = vv.z;
        Vt[(4 * part + 3) * npv + key] = vv.w;
    }
    for (int key = tid; key < np; key += 256) {
        const bool ok = key < N && (mask == nullptr || mask[row_base + key] != 0.f);
        mb[key] = ok ? 0.f : -__builtin_inff();
    }
    __syncthreads();

    for (int qt = w; qt < nqt; qt += 4) {
        const int qrow = min(qt * 16 + pl, N - 1);
        f32x4 Qf = *reinterpret_cast<const f32x4*>(base + (int64_t)qrow * ld + 4 * q);
        // 1/sqrt(16) and log2(e) folded into Q: the scores come out of the MFMAs in base-2 units and the softmax is exp2 of differences
        // (one v_exp_f32 per score, no multiply; the fp32 MFMA shares its issue port with the VALU, so softmax instructions are pipe time)
        Qf *= 0.25f * 1.44269504088896340736f;
        f32x4 s[MAXKT];
        // S^T tiles kt, kt+1: rows = keys 16kt + 4q + r, column = query pl
#pragma unroll
        for (int kt = 0; kt < MAXKT; kt += 2) {
            if (kt < nkt) {
                const f32x4 K0 = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + pl) * KROW + 4 * q);
                const f32x4 K1 = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + 16 + pl) * KROW + 4 * q);
                s[kt] = *reinterpret_cast<const f32x4*>(mb + 16 * kt + 4 * q);  // accumulate in place (no copies out of temporaries)
                s[kt + 1] = *reinterpret_cast<const f32x4*>(mb + 16 * kt + 16 + 4 * q);
                PFM_MFMA4x2(s[kt], K0, Qf, s[kt + 1], K1, Qf)
            }
        }
        float m = -__builtin_inff();
#pragma unroll
        for (int kt = 0; kt < MAXKT; kt += 2)
            if (kt < nkt) {  // v_max3_f32 written out: fmaxf() also emits a canonicalising v_max_f32 x, x, x per operand (113 + 18 instructions per query tile, now 36)
                m = max3f(m, s[kt].x, s[kt].y);
                m = max3f(m, s[kt].z, s[kt].w);
                m = max3f(m, s[kt + 1].x, s[kt + 1].y);
                m = max3f(m, s[kt + 1].z, s[kt + 1].w);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXKT; kt += 2)
            if (kt < nkt) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    s[kt + e].x = __builtin_amdgcn_exp2f(s[kt + e].x - m); s[kt + e].y = __builtin_amdgcn_exp2f(s[kt + e].y - m);
                    s[kt + e].z = __builtin_amdgcn_exp2f(s[kt + e].z - m); s[kt + e].w = __builtin_amdgcn_exp2f(s[kt + e].w - m);
                    l += hsum4(s[kt + e]);
                }
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        // O^T = V^T P^T: A = Vt rows (d = pl) x keys, B = this lane's own P values
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MAXKT; kt += 2) {
            if (kt < nkt) {
                const f32x4 V0 = *reinterpret_cast<const f32x4*>(Vt + pl * npv + 16 * kt + 4 * q);
                const f32x4 V1 = *reinterpret_cast<const f32x4*>(Vt + pl * npv + 16 * kt + 16 + 4 * q);
                PFM_MFMA4x2(o0, V0, s[kt], o1, V1, s[kt + 1])  // unnormalised weights: 1 / l scales the 4 outputs instead of the 72 weights
            }
        }
        const int orow = qt * 16 + pl;
        if (orow < N)
            *reinterpret_cast<f32x4*>(out + (row_base + orow) * D + h * HD + 4 * q) = (o0 + o1) * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// Output head: LayerNorm(hidden) then F dot products; optional fused state update  dst = base + coef*dt*v
// ------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* A;  // [M][Hd] outp_embd hidden (post-activation)
    const float* blob;
    const float* base;  // state the update starts from, or nullptr: dst = v
    const float* dt;    // device scalar (with base)
    float* dst;
    float* v_out;  // optional raw field
    const int* rowsrc;  // compacted rows: row -> row of base / dst / v_out (nullptr: identity)
    const int* m_dev;   // device-side row count overriding M, or nullptr
    int64_t gamma, beta, W, b;
    int M, Hd, F;
    float eps, coef;
};

template <int NI>
__global__ __launch_bounds__(256) void tf_head_kernel(HeadArgs a) {
    constexpr int Hd = 64 * NI;
    const int tid = threadIdx.x, pl = tid & 15;
    const int row = blockIdx.x * 16 + (tid >> 4);
    if (a.m_dev) a.M = *a.m_dev;
    if (blockIdx.x * 16 >= a.M) return;
    const int rowc = min(row, a.M - 1);
    const float* ap = a.A + (int64_t)rowc * Hd + 4 * pl;
    f32x4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(ap + 64 * i);
        s += hsum4(v[i]);
    }
    const float mean = row_sum16(s) / (float)Hd;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 dl = v[i] - mean;
        ss += hsum4(dl * dl);
    }
    const float rstd = 1.0f / sqrtf(row_sum16(ss) / (float)Hd + a.eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.blob + a.gamma + 4 * pl + 64 * i);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.blob + a.beta + 4 * pl + 64 * i);
        v[i] = (v[i] - mean) * rstd * g4 + b4;
    }
#pragma unroll 1
    for (int f = 0; f < a.F; ++f) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.blob + a.W + (int64_t)f * Hd + 4 * pl + 64 * i);
            d += hsum4(v[i] * w4);
        }
        d = row_sum16(d) + a.blob[a.b + f];
        if (pl == (f & 15) && row < a.M) {
            const int64_t e = (int64_t)(a.rowsrc ? a.rowsrc[row] : row) * a.F + f;
            if (a.v_out) a.v_out[e] = d;
            if (a.base) a.dst[e] = __fadd_rn(a.base[e], __fmul_rn(__fmul_rn(a.coef, a.dt[0]), d));
            else if (a.dst) a.dst[e] = d;
        }
    }
}

// x = z * mask (SetFlowMatchingLitModule.sample, flow_matching_module.py:668-671)
static __global__ void tf_premask_kernel(const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x,
                                  int64_t n, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = mask ? z[i] * mask[i / F] : z[i];
}

// out = x + dt * (c0 k0 + c1 k1 + ...): one stage input / the step update of an explicit Runge-Kutta scheme, products
// and sums rounded one by one, left to right (torchdyn's solver steps are unfused elementwise torch ops)
struct RkCombineArgs {
    const float* x;
    const float* dt;  // device scalar
    const float* k[PFM_RK_MAX_STAGES];
    float coef[PFM_RK_MAX_STAGES];
    float* out;
    int64_t n;
    int terms;
};
static __global__ void tf_rk_combine_kernel(RkCombineArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float acc = __fmul_rn(a.coef[0], a.k[0][i]);
    for (int j = 1; j < a.terms; ++j) acc = __fadd_rn(acc, __fmul_rn(a.coef[j], a.k[j][i]));
    a.out[i] = __fadd_rn(a.x[i], __fmul_rn(*a.dt, acc));
}

// Host loop of the fixed-step explicit Runge-Kutta samplers of the row-matrix paths.  state = xs | ys | k_0 .. k_{S-1}
// (n floats each; xs holds the start state).  nfe(t, x_in, v_out) queues one network evaluation.
template <class Nfe>
int sample_rk_rows(const pfm_rk_tableau& tab, const float* t_eval, const float* dt, int n_steps, float* state, int64_t n,
                   hipStream_t s, Nfe nfe) {
    const int S = tab.stages;
    float* xs = state;
    float* ys = state + n;
    float* K = state + 2 * n;
    for (int k = 0; k < n_steps; ++k)
        for (int st = 0; st < S; ++st) {
            int rc = nfe(t_eval + (int64_t)k * S + st, st ? ys : xs, K + st * n);
            if (rc) return rc;
            RkCombineArgs a;
            a.x = xs; a.dt = dt + k; a.n = n;
            const bool last = st == S - 1;
            a.out = last ? xs : ys;
            a.terms = st + 1;
            for (int j = 0; j <= st; ++j) {
                a.k[j] = K + j * n;
                a.coef[j] = last ? tab.b[j] : tab.a[st + 1][j];
            }
            hipLaunchKernelGGL(tf_rk_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        }
    return 0;
}

inline const char* rk_tableau_error(const pfm_rk_tableau* t) {
    if (!t) return "tableau is NULL";
    if (t->stages < 1 || t->stages > PFM_RK_MAX_STAGES) return "tableau.stages must be in 1..4";
    return nullptr;
}

}  // namespace tf
}  // namespace pfm
