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
//   dW products   dW[o][i] = sum_jets sum_p da[p][o] * act[p][i]  are NOT formed here: a sum over all jets needs a
//                 reduction across workgroups, and with one jet per CU nothing can be accumulated on chip.  (Round 1 added every
//                 jet's full dW with fp32 atomics: 2.2 MB of atomics per jet, 258x the gradient's size, atomic-rate bound and
//                 run-to-run non-deterministic.)  The chain kernel stores the gradient rows da (valid rows only, plain 16-byte
//                 stores) and epic_dw_kernel (epic_dw.h) forms each 128x128 dW as ONE GEMM over the rows of all jets, split over
//                 workgroups by row ranges, partial tiles summed in a fixed order: no atomics, bit-reproducible.
//   per-jet GEMV parts (global MLP, folded t/cond/g columns): the operands of every rank-1 update (x, dy) go to a per-jet
//                 record; epic_bwd_reduce_kernel sums x (x) dy over the jets in a fixed order.
// The weight gradient goes to `gblob`, which has the offsets of the weight blob; the 128x128 blocks
// are stored in the accumulator-native order documented in pfm_hip.h (GRAD_D format).
#pragma once
#include "epic_nfe.h"

namespace pfm {

// Diagnostic build only (-DPFM_BDIAG, tests/diag/bwd_stamps.py; never the shipped library): workgroup 0 of the backward chain kernel
// records (id, s_memtime) pairs at its step boundaries.  Expands to nothing otherwise.
#ifdef PFM_BDIAG
extern __device__ unsigned long long g_pfm_bstamps[1024];
extern __device__ int g_pfm_nbstamp;
#define PFM_BSTAMP(id)                                                      \
    do {                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                          \
            const int i_ = g_pfm_nbstamp;                                   \
            if (i_ < 512) {                                                 \
                g_pfm_bstamps[2 * i_] = (unsigned long long)(id);           \
                g_pfm_bstamps[2 * i_ + 1] = __builtin_amdgcn_s_memtime();   \
                g_pfm_nbstamp = i_ + 1;                                     \
            }                                                               \
        }                                                                   \
    } while (0)
#else
#define PFM_BSTAMP(id) do { } while (0)
#endif

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
    int dte;           // MAXT: gradient w.r.t. the time embedding (only when the caller supplies it: t_emb="gaussian")
    int tg;            // VIN_FLOATS: result of km16_tgemv (aliases da3 | w3, both dead after the head; padded where they are too small)
    int total;
};

__host__ __device__ inline BCarve make_bcarve(int N, int F) {
    BCarve c;
    int o = 0;
    c.G = o; o += N * H;
    c.Hb = o; o += N * H;
    c.da3 = o; o += round4(N * F);
    c.w3 = o; o += F * H;
    c.tg = c.da3;
    if (o - c.tg < VIN_FLOATS) o = c.tg + VIN_FLOATS;
    c.maskf = o; o += round4(N);
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
    c.dte = o; o += MAXT;
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

// dX product: for every row p < n_rows, epi(rowbase, valid, acc, pre(rowbase)) with p = rowbase + (lane & 15) and
//   acc[r] = sum_k AT-block[16w + 4q + r][k] * src[p][k]        (a = MFMA_AT fragments of this wave)
// pre(rowbase) fetches what the epilogue needs from global memory (the saved activation whose sign gates the gradient): it is
// called at the head of the tile pair and consumed one pair LATER, in the epilogue that runs next to the following pair's MFMAs.
// `valid` is the literal true for every pair but the last (its rows all lie below n_rows): the functors' predication folds away.
//
// fp32 pipe: the scheme of the forward's gemm_phase (epic_nfe.h) -- the pairs whose two tiles are real run as straight-line bodies
// with a COMPILE-TIME pair index (at most MAXPAIRS: the LDS tile holds <= 160 rows), so every LDS address is a per-lane constant
// plus an immediate and the callers' loads are buffer offsets (scalar base + lane constant): the rolled loop this replaces spent
// ~100 VALU instructions per pair and wave on addresses, register rotation and clamps next to its 64 MFMAs (the fp32 MFMA does not
// co-issue with VALU work) and recomputed a whole tile for every jet with an odd tile count; such a jet now ends with a body that
// issues the real tile's MFMAs only.  B operands are staged through two register sets of one K-quarter each (the ds_reads of a
// quarter fly behind the 16 MFMAs of the previous one, the first quarter of the NEXT pair behind the current pair's last MFMAs).
// Rows read stay inside [0, 16 ntiles).
// BF16 (v_mfma_f32_16x16x32_bf16 on operands rounded on the fly, fp32 accumulate -- what Lightning's precision="bf16-mixed" means for
// these products): the rolled loop (no half pair on that pipe), tiles clamped to the last computed one.
template <bool BF16 = false, typename Pre, typename Epi>
__device__ __forceinline__ void gemm_dx(const f32x4 (&a)[8], const float* __restrict__ src, int n_rows, Pre pre, Epi epi) {
    const int tid_ = launder(threadIdx.x);
    const int lane = tid_ & 63;
    const int pl = lane & 15, q = lane >> 4;
    const int ntiles = (n_rows + TILE - 1) / TILE, npairs = (ntiles + 1) >> 1;
    if (npairs <= 0) return;
    f32x4 X0[2], X1[2], Y0[2], Y1[2];
    f32x4 pacc0 = {0.f, 0.f, 0.f, 0.f}, pacc1 = pacc0, px0 = pacc0, px1 = pacc0;
    if constexpr (BF16) {
        int koff[8];  // (row & 15) == pl for every tile: the swizzled slot offsets are per-lane constants
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) koff[kt] = pl * H + (((4 * kt + q) ^ pl) << 2);
#define PFM_DX_LOADQ(B0, B1, t0p, t1p, qq)                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                 \
        B0[kk] = *reinterpret_cast<const f32x4*>((t0p) + koff[2 * (qq) + kk]);          \
        B1[kk] = *reinterpret_cast<const f32x4*>((t1p) + koff[2 * (qq) + kk]);          \
    }
#define PFM_DX_MFMAQ(B0, B1, qq)                                                                                   \
    {                                                                                                              \
        const bf16x8 ab = pack_bf16x8(a[2 * (qq)], a[2 * (qq) + 1]);                                               \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, pack_bf16x8(B0[0], B0[1]), acc0, 0, 0, 0);              \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, pack_bf16x8(B1[0], B1[1]), acc1, 0, 0, 0);              \
    }
        const float* t0p = src;
        const float* t1p = src + (ntiles > 1 ? TILE * H : 0);
        PFM_DX_LOADQ(X0, X1, t0p, t1p, 0);
        int prb = -1;  // row base of the pair whose epilogue is pending
        for (int pair = 0; pair < npairs; ++pair) {
            const int rb = pair * 2 * TILE;
            const f32x4 x0 = pre(rb), x1 = pre(rb + TILE);  // land during this pair's MFMAs
            PFM_DX_LOADQ(Y0, Y1, t0p, t1p, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (prb >= 0) {  // the previous pair's epilogue, next to this pair's MFMAs
                epi(prb, true, pacc0, px0);
                epi(prb + TILE, true, pacc1, px1);
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            PFM_DX_MFMAQ(X0, X1, 0);
            PFM_DX_LOADQ(X0, X1, t0p, t1p, 2);
            __builtin_amdgcn_sched_barrier(0);
            PFM_DX_MFMAQ(Y0, Y1, 1);
            PFM_DX_LOADQ(Y0, Y1, t0p, t1p, 3);
            __builtin_amdgcn_sched_barrier(0);
            PFM_DX_MFMAQ(X0, X1, 2);
            // first quarter of the next pair (the last pair re-reads its own: harmless)
            const int tn0 = min(2 * pair + 2, ntiles - 1), tn1 = min(2 * pair + 3, ntiles - 1);
            const float* n0p = src + tn0 * TILE * H;
            const float* n1p = src + tn1 * TILE * H;
            PFM_DX_LOADQ(X0, X1, n0p, n1p, 0);
            __builtin_amdgcn_sched_barrier(0);
            PFM_DX_MFMAQ(Y0, Y1, 3);
            t0p = n0p; t1p = n1p;
            pacc0 = acc0; pacc1 = acc1; px0 = x0; px1 = x1;
            prb = rb;
        }
        epi(prb, prb + pl < n_rows, pacc0, px0);
        epi(prb + TILE, prb + TILE + pl < n_rows, pacc1, px1);
#undef PFM_DX_LOADQ
#undef PFM_DX_MFMAQ
    } else {
        // four per-lane slot offsets: slot 4 kt + q of k-tile kt >= 4 is slot 4 (kt - 4) + q plus 16 (pl < 16 never touches bit 4), i.e. 64
        // floats further -- an immediate in the ds_read
        const float* kb[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kb[kt] = src + pl * H + (((4 * kt + q) ^ pl) << 2);
#define PFM_DX_K(kt, fo) (kb[(kt) & 3] + 64 * ((kt) >> 2) + (fo))
        // both tiles of a pair / the pair's first tile only (the last tile of a jet with an odd tile count)
#define PFM_DX_LOADQ(B0, B1, fo, qq)                                                            \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                         \
        B0[kk] = *reinterpret_cast<const f32x4*>(PFM_DX_K(2 * (qq) + kk, (fo)));                \
        B1[kk] = *reinterpret_cast<const f32x4*>(PFM_DX_K(2 * (qq) + kk, (fo) + TILE * H));     \
    }
#define PFM_DX_LOADQ1(B0, B1, fo, qq)                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) B0[kk] = *reinterpret_cast<const f32x4*>(PFM_DX_K(2 * (qq) + kk, (fo)));
#define PFM_DX_MFMAQ(B0, B1, qq)                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) { PFM_MFMA_PAIR(acc0, acc1, a[2 * (qq) + kk], B0[kk], B1[kk]); }
#define PFM_DX_MFMAQ1(B0, B1, qq)                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                         \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].x, B0[kk].x, acc0, 0, 0, 0); \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].y, B0[kk].y, acc0, 0, 0, 0); \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].z, B0[kk].z, acc0, 0, 0, 0); \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].w, B0[kk].w, acc0, 0, 0, 0); \
    }
        const int nfull = ntiles >> 1;  // pairs whose two tiles are real
        // the next pair's first quarter rides behind the current pair's last MFMAs (nothing behind the jet's last body)
#define PFM_DX_BODY(LQ, MF, NEXT_FO)                                                            \
    {                                                                                           \
        constexpr int rb = pair * 2 * TILE, fo = rb * H;                                        \
        const f32x4 x0 = pre(rb), x1 = pre(rb + TILE); /* land during this pair's MFMAs */     \
        LQ(Y0, Y1, fo, 1);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if constexpr (pair > 0) { /* the previous pair's epilogue (every row valid), next to this pair's MFMAs */ \
            epi(rb - 2 * TILE, true, pacc0, px0);                                               \
            epi(rb - TILE, true, pacc1, px1);                                                   \
        }                                                                                       \
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};                         \
        MF(X0, X1, 0);                                                                          \
        LQ(X0, X1, fo, 2);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        MF(Y0, Y1, 1);                                                                          \
        LQ(Y0, Y1, fo, 3);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        MF(X0, X1, 2);                                                                          \
        NEXT_FO                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        MF(Y0, Y1, 3);                                                                          \
        pacc0 = acc0; pacc1 = acc1; px0 = x0; px1 = x1;                                         \
    }
#define PFM_DX_NEXT                                                                             \
        if (pair + 1 < nfull) { PFM_DX_LOADQ(X0, X1, fo + 2 * TILE * H, 0); }                   \
        else if (pair + 1 < npairs) { PFM_DX_LOADQ1(X0, X1, fo + 2 * TILE * H, 0); }
#define PFM_DX_NONE
#define PFM_DX_PAIR_AT(P)                                                                       \
    if ((P) < nfull) {                                                                          \
        constexpr int pair = (P);                                                               \
        PFM_DX_BODY(PFM_DX_LOADQ, PFM_DX_MFMAQ, PFM_DX_NEXT)                                    \
    }
#define PFM_DX_TAIL_AT(P)                                                                       \
    if ((P) == nfull) {                                                                         \
        constexpr int pair = (P);                                                               \
        PFM_DX_BODY(PFM_DX_LOADQ1, PFM_DX_MFMAQ1, PFM_DX_NONE)                                  \
    }
        if (nfull > 0) { PFM_DX_LOADQ(X0, X1, 0, 0); } else { PFM_DX_LOADQ1(X0, X1, 0, 0); }
        static_assert(MAXPAIRS == 5, "unroll PFM_DX_PAIR_AT to MAXPAIRS");
        PFM_DX_PAIR_AT(0) PFM_DX_PAIR_AT(1) PFM_DX_PAIR_AT(2) PFM_DX_PAIR_AT(3) PFM_DX_PAIR_AT(4)
        if (nfull < npairs) {  // odd tile count: one real tile in the last pair
            PFM_DX_TAIL_AT(0) PFM_DX_TAIL_AT(1) PFM_DX_TAIL_AT(2) PFM_DX_TAIL_AT(3) PFM_DX_TAIL_AT(4)
        }
        const int rbl = (npairs - 1) * 2 * TILE;
        epi(rbl, rbl + pl < n_rows, pacc0, px0);
        if (nfull == npairs) epi(rbl + TILE, rbl + TILE + pl < n_rows, pacc1, px1);
#undef PFM_DX_TAIL_AT
#undef PFM_DX_PAIR_AT
#undef PFM_DX_NONE
#undef PFM_DX_NEXT
#undef PFM_DX_BODY
#undef PFM_DX_MFMAQ1
#undef PFM_DX_MFMAQ
#undef PFM_DX_LOADQ1
#undef PFM_DX_LOADQ
#undef PFM_DX_K
    }
}

// ---- per-jet record of the reductions over jets (floats) ------------------------------------------------------------
// stage s = 0 (stem: fc_g1 / fc_g2, fc_l1 / fc_l2 extras) and s = 1 + k (EPiC layer k): the operands of the rank-1 updates
//   dW_gl1 += vin (x) dag1, dW_gl2 += vin2 (x) dag2, dWe_lc1 += [vin[0..Ke) ; gout] (x) dbj1, dWe_lc2 += vin[0..Ke) (x) dbj2
// head: this jet's partial sums of dW3 (fc_l3 particle block) and dWx (fc_l1 particle block), db3.
struct BwdRec {
    static constexpr int VIN = 0, VIN2 = VIN_FLOATS, DAG1 = VIN2 + VIN2_FLOATS, DAG2 = DAG1 + H, DBJ1 = DAG2 + MAXL,
                         DBJ2 = DBJ1 + H, GOUT = DBJ2 + H, STAGE = GOUT + MAXL;
    int head;             // (layers + 1) * STAGE
    int dW3, dWx, db3;    // MAXF*H, MAXF*H, 16
    int total;
};
__host__ __device__ inline BwdRec make_bwd_rec(int layers) {
    BwdRec r;
    r.head = (layers + 1) * BwdRec::STAGE;
    r.dW3 = r.head;
    r.dWx = r.dW3 + MAXF * H;
    r.db3 = r.dWx + MAXF * H;
    r.total = r.db3 + 16;
    return r;
}

// ---- scratch of one backward call (floats) -----------------------------------------------------------------------------
//   nrows [B] (int32)            rows the chain kernel computed per jet (skip-masked-tail: last valid particle + 1)
//   rec   [B][BwdRec.total]
//   da    [B][nblk][N][H]        gradient rows; block 0: da2 of the stem (pairs with x1 -> dW of fc_l2),
//                                1 + 2k: da2 of layer k (pairs with l1_k -> fc_local2), 2 + 2k: da1 of layer k (pairs with h_k -> fc_local1)
//   part  [nblk][nsplit][H*H]    partial dW tiles of epic_dw_kernel
struct BwdWork {
    int64_t nrows, rec, da, part, total;
    int nblk, nsplit, rec_floats;
};
__host__ __device__ inline BwdWork make_bwd_work(int N, int layers, int B) {
    BwdWork w;
    w.nblk = 2 * layers + 1;
    // ~2 workgroups per CU of a 256-CU part over all blocks, at least ~8 16-row pieces per split (a fixed function of the
    // shapes, never of the device: the summation order, hence the result, is the same everywhere)
    int ns = 512 / w.nblk;
    const int cap = (B * ((N + 15) / 16) + 7) / 8;
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    w.nsplit = ns;
    w.rec_floats = make_bwd_rec(layers).total;
    int64_t o = 0;
    w.nrows = o; o += (B + 63) & ~63;
    w.rec = o; o += (int64_t)B * w.rec_floats;
    w.da = o; o += (int64_t)B * w.nblk * N * H;
    w.part = o; o += (int64_t)w.nblk * ns * H * H;
    w.total = o;
    return w;
}

// Transposed GEMV on a KM16 block: out[k - 16 p_lo] = sum_o W[k][o] * v[o] for every row k of the panels p_lo .. p_hi (16 rows
// each), v a 128-vector in LDS, out in LDS.  Wave w takes the panels p_lo + w, + 8, ...: a panel is 512 float4 = 8 coalesced
// 1-KiB loads per wave (lane l, step s: row l & 15, outputs 4 (4 s + (l >> 4)) ..+3); the 4 DPP rows of a wave are summed with two
// shuffles, so no cross-wave reduction and no barrier inside.  The caller puts a barrier before reading `out`.
#ifndef PFM_TGEMV_DB
#define PFM_TGEMV_DB 0  // (1 = the next panel's weights requested before this panel's FMAs: 15 VGPRs over the budget of the backward chain kernel, and no faster: 0.575 vs 0.572 ms per step)
#endif
__device__ __forceinline__ void km16_tgemv(const float* __restrict__ W, int p_lo, int p_hi, const float* __restrict__ v,
                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const f32x4* Wp = reinterpret_cast<const f32x4*>(W);
    // The vector's float4s of this lane are read per panel in two batches of four, each batch requested before its first use (hipcc
    // sinks a ds_read to its use: eight dependent LDS round trips per panel before round 4).  PFM_TGEMV_DB = 1 would also keep two
    // panels of weights in flight (the next panel's eight loads in front of this panel's FMAs): measured, not faster, and over the
    // register budget of the backward chain kernel.
    int p = p_lo + w;
    f32x4 wv[8], wn[8];
    if (p <= p_hi) {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) wv[s8] = Wp[(size_t)p * 512 + s8 * 64 + lane];
    }
    for (; p <= p_hi; p += NW) {
        const bool more = PFM_TGEMV_DB && p + NW <= p_hi;
        if (more) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) wn[s8] = Wp[(size_t)(p + NW) * 512 + s8 * 64 + lane];
        }
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 dv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dv[i] = *reinterpret_cast<const f32x4*>(v + 4 * (4 * (4 * h + i) + (lane >> 4)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ww = wv[4 * h + i];
                a += (ww.x * dv[i].x + ww.y * dv[i].y) + (ww.z * dv[i].z + ww.w * dv[i].w);
            }
        }
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (lane < 16) out[(p - p_lo) * 16 + lane] = a;
        if (more) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) wv[s8] = wn[s8];
        } else if (!PFM_TGEMV_DB && p + NW <= p_hi) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) wv[s8] = Wp[(size_t)(p + NW) * 512 + s8 * 64 + lane];
        }
    }
}

// dte[k] += We[k][:] . dy for the time rows k < T of a KM16 extras block (the gradient a caller-supplied time embedding receives
// through the per-jet bias of a local Linear).  Two barriers.
__device__ __forceinline__ void dtemb_from_extras(const float* __restrict__ We, const float* __restrict__ dy, int T,
                                                  float* __restrict__ tg, float* __restrict__ dte) {
    km16_tgemv(We, 0, (T - 1) >> 4, dy, tg);
    __syncthreads();
    if ((int)threadIdx.x < T) dte[threadIdx.x] += tg[threadIdx.x];
    __syncthreads();
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
// Out: c.dP (gradient w.r.t. the raw pooled sum), c.dg = dL/dg_in (STEM: unused), c.dag1 / c.dag2 / c.vin2 for the record.
template <bool STEM>
__device__ __forceinline__ void global_backward(const JetDims& j, const float* __restrict__ blob,
                                                const pfm_dense_lin& gl1,
                                                const pfm_dense_lin& gl2, float* __restrict__ lds, const BCarve& c,
                                                const float* __restrict__ sv_g1, const float* __restrict__ sv_gout,
                                                bool want_dt = false) {
    const int tid = threadIdx.x;
    const int TC = j.T + j.C;
    const float nvalid = lds[c.misc];
    // this thread's row of W_gl2 and its saved g1 entry: requested first, consumed behind the barrier below (the loads used to sit
    // right in front of their use: an exposed L2 / HBM round trip per layer)
    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, w2 = w0, w3 = w0;
    float g1v = 0.f;
    if (tid < H) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(blob + gl2.W + (TC + tid) * 16);  // KP16 row: 16 floats
        w0 = wr[0]; w1 = wr[1]; w2 = wr[2]; w3 = wr[3];
        g1v = sv_g1[tid];
    }
    // dag2 = dg_out * phi'(g_out);  vin2 = [temb ; cond ; g1]
    if (tid < j.L) lds[c.dag2 + tid] = lds[c.dg + tid] * dlrelu(sv_gout[tid], j.slope);
    if (tid >= 64 && tid < 64 + TC) lds[c.vin2 + (tid - 64)] = lds[c.vin + (tid - 64)];
    if (tid >= 256 && tid < 256 + H) lds[c.vin2 + TC + (tid - 256)] = sv_g1[tid - 256];
    __syncthreads();
    // (dW_gl2 = sum_jets vin2 (x) dag2, db_gl2 = sum dag2: epic_bwd_reduce_kernel)   dg1 = W_gl2[g1 rows] . dag2
    if (tid < H) {
        // dag2 as four float4s read up front, entries >= L selected to zero (the KP16 row is zero-padded there; the guarded form was a
        // branch + a dependent ds_read_b32 round trip per term).  Same terms in the same order: fma(w, 0, a) = a.
        const f32x4* dq4 = reinterpret_cast<const f32x4*>(lds + c.dag2);
        f32x4 d0 = dq4[0], d1 = dq4[1], d2 = dq4[2], d3 = dq4[3];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            d0[jj] = jj < j.L ? d0[jj] : 0.f;
            d1[jj] = 4 + jj < j.L ? d1[jj] : 0.f;
            d2[jj] = 8 + jj < j.L ? d2[jj] : 0.f;
            d3[jj] = 12 + jj < j.L ? d3[jj] : 0.f;
        }
        float a = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            a = fmaf(w0[jj], d0[jj], a);
            a = fmaf(w1[jj], d1[jj], a);
            a = fmaf(w2[jj], d2[jj], a);
            a = fmaf(w3[jj], d3[jj], a);
        }
        lds[c.dag1 + tid] = a * dlrelu(g1v, j.slope);  // dag1 = dg1 * phi'(g1)
    }
    __syncthreads();
    // (dW_gl1 = sum_jets vin (x) dag1, db_gl1 = sum dag1: epic_bwd_reduce_kernel)
    // dvin[k] = W_gl1[k][:] . dag1 for k >= TC  -> dmean, dsum (-> dP), dg_in
    // (with want_dt also the time rows k < T: d loss / d temb of a caller-supplied embedding gets W_gl1[k] . dag1 + W_gl2[k] . dag2)
    const int p_lo = want_dt ? 0 : TC >> 4, k_hi = TC + 2 * H + (STEM ? 0 : j.L) - 1;
    km16_tgemv(blob + gl1.W, p_lo, k_hi >> 4, lds + c.dag1, lds + c.tg);
    __syncthreads();
    const float* dv = lds + c.tg - 16 * p_lo;  // dv[k] = dvin[k]
    if (want_dt && tid >= 256 && tid < 256 + j.T) {
        const int k = tid - 256;
        float a = dv[k];
        for (int o = 0; o < j.L; ++o) a = fmaf(blob[gl2.W + k * 16 + o], lds[c.dag2 + o], a);
        lds[c.dte + k] += a;
    }
    if (tid < H) {
        // pooled mean = sum / n (epic.py:161), pooled sum * scale (:162)
        lds[c.dP + tid] = dv[TC + tid] / nvalid + dv[TC + H + tid] * j.sscale;
    } else if (!STEM && tid < H + j.L) {
        const int jj = tid - H;
        lds[c.dg + jj] = lds[c.dag2 + jj] + dv[TC + 2 * H + jj];  // residual path + vin path
    }
    __syncthreads();
}

}  // namespace pfm
