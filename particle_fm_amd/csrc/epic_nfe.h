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
__device__ __forceinline__ void load_afrag(f32x4 (&a)[8], blob_rsrc rs, int64_t A_off, int w, int lane) {
    const int lb = ((w * 8) * 64 + lane) * 16;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) a[kt] = bload4(rs, A_off + kt * 256, lb);
}

// two independent accumulator chains, alternated instruction by instruction (a dependent
// v_mfma_f32_16x16x4_f32 needs 40 cycles, the pipe issues one every 32)
#define PFM_MFMA_PAIR(acc0, acc1, av, bv0, bv1)                                       \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (bv0).x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (bv1).x, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (bv0).y, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (bv1).y, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (bv0).z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (bv1).z, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (bv0).w, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (bv1).w, acc1, 0, 0, 0);

// A operand of one 128x128 block for this wave from its MFMA_A16 copy (pfm_hip.h): 4 x 16 bytes = 16 VGPRs, already bf16
__device__ __forceinline__ void load_afrag16(f32x4 (&a)[4], blob_rsrc rs, int64_t A16_off, int w, int lane) {
    const int lb = ((w * 4) * 64 + lane) * 16;
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2) a[kt2] = bload4(rs, A16_off + kt2 * 256, lb);
}

// ---- split-fp16 operands (PFM_F_F16X3_MFMA): fp32-accurate products on the fp16 matrix pipe -----------------------
// x = hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11)  (22 significant bits); then
//   x . w  =  hi.whi  +  2^-11 (hi.wlo + lo.whi)  +  O(2^-22),
// three v_mfma_f32_16x16x32_f16 (16 cycles each, 32 k) instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each, 4 k):
// 5.3x less matrix-pipe time at the accuracy of fp32 re-association noise (measured 6e-7 on the NFE, like the fp32
// kernel).  The activation tiles live in LDS as two fp16 planes (hi, lo): same bytes as fp32, element index unchanged.
// Valid for |x| < 65504 (fp16 range); the network's activations are O(1..100).
// Element index inside an fp16 plane: rows of 16 sixteen-byte slots (8 halfs = 8 consecutive k), slot index XOR-ed
// with (row & 15): the ds_read_b128 of v_mfma_f32_16x16x32_f16's B operand (lane (row, q) reads slot 4 kt + q) and
// the 8-byte stores of the epilogue are bank-conflict free.  `slot4` counts groups of 4 features, like lds_off.
__device__ __forceinline__ int x3_off(int p, int slot4) { return p * H + (((slot4 >> 1) ^ (p & 15)) << 3) + ((slot4 & 1) << 2); }
// A operand of v_mfma_f32_16x16x32_f16 from the MFMA_A block: lane (i, q) needs k = 32 kt' + 8 q .. + 7, i.e. the two
// float4 that lanes (i, 2(q&1)) and (i, 2(q&1)+1) of k-tile 2 kt' + (q>>1) hold in the fp32 format: a[2 kt' + h].
__device__ __forceinline__ void load_afrag_k8(f32x4 (&a)[8], blob_rsrc rs, int64_t A_off, int w, int lane) {
    const int pl = lane & 15, q = lane >> 4;
#pragma unroll
    for (int kp = 0; kp < 4; ++kp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = 2 * kp + (q >> 1), src = pl + 16 * (2 * (q & 1) + h);
            a[2 * kp + h] = bload4(rs, A_off, (((w * 8 + kt) * 64) + src) * 16);
        }
}
template <int MODE>
__device__ __forceinline__ void load_afrag_m(f32x4 (&a)[8], blob_rsrc rs, int64_t A_off, int w, int lane) {
    if (MODE == 2) load_afrag_k8(a, rs, A_off, w, lane);
    else load_afrag(a, rs, A_off, w, lane);
}


template <bool SAVE, bool VB16 = false>
__device__ __forceinline__ void pool_finish(f32x4 psum, const JetDims& j, float* __restrict__ lds, int vin_off, int misc_off,
                                            int oslot, int pl, float* __restrict__ save_pool);

// The second jet of a two-jet workgroup as a particle phase sees it (NSEG == 2; see Segs / SegView in pfm_common.h)
struct Seg2Phase {
    const float* bj = nullptr;     // its per-jet bias vector for this Linear
    const float* mask1 = nullptr;  // mask of its rows, 0 on the first jet's rows
    int t1 = 1 << 20;              // its first 16-row tile
    int vin1 = 0, misc1 = 0;       // where its pooled mean / sum go, where its valid count is
};

// FOUR short jets in a workgroup (NSEG == 4, the quad mode of the lean sampler, epic_fast.h): fixed geometry -- jet s owns the 32-row
// slot [32 s, 32 s + 32) of the activation tiles = tile pair s of every particle phase, so a pair body knows its jet at compile time:
// its bias vector, and where its pooled mean / sum go.  LDS float offsets.
struct QuadPhase {
    int bj[4];    // per-jet bias of this Linear (all four the same offset if the bias does not depend on the jet)
    int vin[4];   // where the jet's pooled mean / sum go (POOL)
    int misc[4];  // where its valid count is (POOL)
};

// ---- prefetch lists -----------------------------------------------------------------------------------------------
// A particle phase keeps the matrix pipe busy for thousands of cycles while the CU's vector-memory path idles, so the weights the
// NEXT phases need (A fragments, per-jet GEMV panels) are requested during it.  A wave issues in order: 20 loads in a row at the
// head of a phase hold EVERY wave back ~2.5k cycles (8 waves x 20 KiB through the 64 B/clk path) before its first MFMA -- measured
// as the 2.9k + 3.5k cycles of load issue per layer in tests/diag/stamps.py.  gemm_phase therefore unrolls its pair loop (at most
// MAXPAIRS pairs: the LDS tile holds <= 160 rows) and issues ONE load of the list (two for lists longer than 20) behind every
// K-quarter of every pair: register indices are compile-time constants and a wave never queues more than that at a time.  What a
// short jet cannot spread over its few pairs is issued after the loop, as before.
constexpr int MAXPAIRS = 5;
struct PfSeg {  // load i of a segment: bload4(rs, off + i * stride, lane_bytes)
    int64_t off;
    int stride, lane_bytes;
};
template <int N0, int N1 = 0, int N2 = 0, int N3 = 0>
struct Prefetch {
    static constexpr int COUNT = N0 + N1 + N2 + N3;
    blob_rsrc rs;
    f32x4 *r0, *r1, *r2, *r3;
    PfSeg s0, s1, s2, s3;
    template <int I>
    __device__ __forceinline__ void issue() const {
#ifdef PFM_AB_NOPF  // (timing-only ablation: no weight loads at all; stale registers)
        return;
#endif
        if constexpr (I < N0) r0[I] = bload4(rs, s0.off + (int64_t)I * s0.stride, s0.lane_bytes);
        else if constexpr (I < N0 + N1) r1[I - N0] = bload4(rs, s1.off + (int64_t)(I - N0) * s1.stride, s1.lane_bytes);
        else if constexpr (I < N0 + N1 + N2) r2[I - N0 - N1] = bload4(rs, s2.off + (int64_t)(I - N0 - N1) * s2.stride, s2.lane_bytes);
        else if constexpr (I < COUNT) r3[I - N0 - N1 - N2] = bload4(rs, s3.off + (int64_t)(I - N0 - N1 - N2) * s3.stride, s3.lane_bytes);
    }
    template <int I0, int I1>
    __device__ __forceinline__ void issue_range() const {
        if constexpr (I0 < I1) {
            issue<I0>();
            issue_range<I0 + 1, I1>();
        }
    }
    // Slot S (= 4 * pair + K-quarter) of a particle phase carries LPS loads of the list: one while the list fits the 4 * MAXPAIRS slots
    // of a full-length jet, two for the longer lists (static: a runtime choice inside the pair bodies would cut their straight-line
    // blocks apart -- measured: +4 % on a 10-tile jet).  The pairs a shorter jet does not have issue theirs behind the loop.
    static constexpr int LPS = (COUNT + 19) / 20 > 0 ? (COUNT + 19) / 20 : 1;
    template <int S>
    __device__ __forceinline__ void issue_slot() const {
        issue_range<(S * LPS < COUNT ? S * LPS : COUNT), ((S + 1) * LPS < COUNT ? (S + 1) * LPS : COUNT)>();
    }
    template <int P>
    __device__ __forceinline__ void issue_pair() const {  // everything pair P would have carried
        issue_range<(4 * P * LPS < COUNT ? 4 * P * LPS : COUNT), (4 * (P + 1) * LPS < COUNT ? 4 * (P + 1) * LPS : COUNT)>();
    }
    // the part of the list a jet with `nfull` full tile pairs had no pairs for (behind the pair loop)
    __device__ __forceinline__ void issue_tail(int nfull) const {
        if (nfull <= 0) issue_pair<0>();
        if (nfull <= 1) issue_pair<1>();
        if (nfull <= 2) issue_pair<2>();
        if (nfull <= 3) issue_pair<3>();
        if (nfull <= 4) issue_pair<4>();
    }
};
using PfNone = Prefetch<0>;
// segment of an A-fragment load (load_afrag) / of a run of KM16 GEMV panels (gemv4_load) / of this thread's fc_global2 rows
__device__ __forceinline__ PfSeg seg_afrag(int64_t A_off, int w, int lane) { return PfSeg{A_off, 256, ((w * 8) * 64 + lane) * 16}; }
__device__ __forceinline__ PfSeg seg_afrag16(int64_t A16_off, int w, int lane) { return PfSeg{A16_off, 256, ((w * 4) * 64 + lane) * 16}; }
__device__ __forceinline__ PfSeg seg_panels(int64_t W_off, int first_panel, int tid) {
    return PfSeg{W_off + (int64_t)first_panel * (NT * 4), NT * 4, tid * 16};
}

// One particle phase: for every row p < n_rows
//   dst[p][16w..16w+16) = lrelu( W[16w.., :] . src[p][:] + bj[16w..] (+ resid[p][16w..] if RESID) )
// Each wave walks the particle tiles two at a time (two accumulator chains).  Software pipeline, one
// straight-line basic block per pair so that hipcc can interleave the three kinds of work:
//   * B operands are staged through registers in two halves of the K range: the ds_reads of one half are in
//     flight while the MFMAs of the other half issue;
//   * the epilogue of pair i-1 (leaky-relu, ds_write, pooling FMAs) sits next to the MFMAs of pair i.
// Rows are NOT clamped: tiles may run up to 31 rows past n_rows / N (the carve keeps that window inside LDS);
// such rows only produce garbage in their own output columns, which are never stored or pooled.
// POOL: masked column sums -> vin (mean | sum*scale).  SAVE: rows also go to `save` (global).
// NSEG == 2: the rows hold two jets (the second from tile s2.t1 on): each tile takes its own jet's bias, the pool keeps two sums.
// the number of full tile pairs the pair loop of gemm_phase runs for n_rows rows
template <bool BF16>
__device__ __forceinline__ int phase_full_pairs(int n_rows) {
    return BF16 ? (n_rows + 2 * TILE - 1) / (2 * TILE) : (n_rows / TILE + (n_rows % TILE >= 1 ? 1 : 0)) / 2;  // bf16: no half pair
}
// TAIL = false: the caller issues pf.issue_tail(phase_full_pairs<BF16>(n_rows)) itself (behind work of its own that must not wait for
// those loads: a vmcnt wait covers every load issued before it)
// AF: float4 registers of the A operand: 8 (fp32 MFMA_A fragment; BF16: rounded here) or, BF16 only, 4 (an MFMA_A16 fragment)
// RB16 (the lean bf16 samplers, round 4 -- "resident bf16"): src and dst are BF16 PLANES -- rows of 128 bf16 = 16 sixteen-byte slots,
// slot s of row r at float offset r * 64 + ((s ^ (r & 15)) << 2), slot 4 kt2 + q = the eight k (32 kt2 + 16 h + 4 q + r) a lane (row, q)
// feeds v_mfma_f32_16x16x32_bf16 for K-quarter kt2 (pack_bf16x8's order) -- written by the PRODUCING phase's epilogue (one
// v_cvt_pk_bf16_f32 pair + ds_write_b64 per tile and lane) and read as ONE ds_read_b128 per tile and quarter, no conversion: every
// activation was rounded eight times over before (once per consuming wave), 32 v_cvt_pk per tile pair and wave against 8 MFMAs.  dst32
// (or nullptr): where the same rows also go as fp32 (the residual input of the next local linear 2).  Same roundings of the same
// numbers: bit-identical to the fp32-resident bf16 flavour.
#ifndef PFM_PAIR_NEST
#define PFM_PAIR_NEST 1
#endif
template <bool RESID, bool POOL, bool SAVE, bool BF16 = false, typename PF = PfNone, int NSEG = 1, bool TAIL = true, int AF = 8, bool VB16 = false,
          bool RB16 = false>
__device__ __forceinline__ void gemm_phase(const f32x4 (&a)[AF], const float* __restrict__ src,
                                           float* __restrict__ dst, const float* __restrict__ resid,
                                           const float* __restrict__ bj, const float* __restrict__ maskf,
                                           const JetDims& j, float* __restrict__ lds, const Carve& c,
                                           float* __restrict__ save, float* __restrict__ save_pool, int n_rows,
                                           const PF& pf = PF{}, const Seg2Phase& s2 = Seg2Phase{}, const QuadPhase* qp = nullptr,
                                           float* __restrict__ dst32 = nullptr) {
    static_assert(!RB16 || (BF16 && !SAVE), "bf16-resident activations: the lean bf16 samplers");
    constexpr int SROW = RB16 ? H / 2 : H;  // floats per activation row of src / dst
    static_assert(NSEG == 1 || NSEG == 2 || NSEG == 4, "one jet, a packed pair, or four 32-row slots");
    static_assert(!(NSEG == 4 && SAVE), "quad mode: inference only");
    static_assert(AF == 8 || (BF16 && AF == 4), "A operand: 8 fp32 float4s, or 4 pre-packed bf16 ones for the bf16 pipe");
    const int tid_ = launder(threadIdx.x);
    const int lane = tid_ & 63, w = tid_ >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int oslot = 4 * w + q;  // 16-byte slot of this lane's 4 output features
    const float slope = j.slope;
    const f32x2 slope2 = {slope, slope};  // wave-uniform: an SGPR pair
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bj + 4 * oslot);
    f32x4 biasB = bias, psumB = {0.f, 0.f, 0.f, 0.f};
    if (NSEG == 2) biasB = *reinterpret_cast<const f32x4*>(s2.bj + 4 * oslot);
    const int t1 = s2.t1;
    const float* const mask1 = s2.mask1;
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    f32x4 psq[NSEG == 4 ? 4 : 1];  // quad mode: one pool sum per tile pair = per jet (static indices only)
    if (NSEG == 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) psq[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int npairs = (n_rows + 2 * TILE - 1) / (2 * TILE);
    // (row & 15) == pl for every tile, so the swizzled slot offsets are per-lane constants -- four of them: slot 4 kt + q of k-tile
    // kt >= 4 is slot 4 (kt - 4) + q plus 16 (pl < 16 never touches bit 4), i.e. 64 floats further: an immediate in the ds_read
    int koff4[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) koff4[kt] = RB16 ? pl * SROW + (((4 * kt + q) ^ pl) << 2) : pl * H + (((4 * kt + q) ^ pl) << 2);
#define PFM_KOFF(kt) (koff4[(kt) & 3] + 64 * ((kt) >> 2))
    const int ooff = pl * H + ((oslot ^ pl) << 2);
    // RB16: where this lane's four output features (16 w + 4 q ..) go in a bf16 plane: slot 4 (w >> 1) + q, half w & 1 (8 bytes = 2 floats)
    const int oofb = pl * SROW + (((4 * (w >> 1) + q) ^ pl) << 2) + 2 * (w & 1);
    float* const sink = lds + c.dummy;
    // operand staging: two register sets X / Y of one K-quarter (2 kt x 2 tiles = 16 VGPRs each)
    f32x4 X0[2], X1[2], Y0[2], Y1[2];
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pacc0 = {0.f, 0.f, 0.f, 0.f}, pacc1 = {0.f, 0.f, 0.f, 0.f};
#define PFM_LOADQ(B0, B1, base, qq)                                                        \
    if constexpr (RB16) { /* one 16-byte unit = the quarter's eight bf16 of this lane */   \
        B0[0] = *reinterpret_cast<const f32x4*>((base) + koff4[(qq)]);                     \
        B1[0] = *reinterpret_cast<const f32x4*>((base) + TILE * SROW + koff4[(qq)]);       \
    } else {                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                    \
        B0[kk] = *reinterpret_cast<const f32x4*>((base) + PFM_KOFF(2 * (qq) + kk));            \
        B1[kk] = *reinterpret_cast<const f32x4*>((base) + TILE * H + PFM_KOFF(2 * (qq) + kk)); \
    }                                                                                      \
    }
    // bf16 pipe: one v_mfma_f32_16x16x32_bf16 per tile and K-quarter (32 k: the two float4s of the quarter, see pack_bf16x8)
    bf16x8 ab[4];
    if constexpr (BF16) {
#pragma unroll
        for (int kt2 = 0; kt2 < 4; ++kt2) {
            if constexpr (AF == 4) ab[kt2] = __builtin_bit_cast(bf16x8, a[kt2]);
            else ab[kt2] = pack_bf16x8(a[2 * kt2], a[2 * kt2 + 1]);
        }
    }
#define PFM_MFMAQ(B0, B1, qq)                                                                                  \
    if constexpr (RB16) {                                                                                      \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(qq)], __builtin_bit_cast(bf16x8, B0[0]), acc0, 0, 0, 0); \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(qq)], __builtin_bit_cast(bf16x8, B1[0]), acc1, 0, 0, 0); \
    } else if constexpr (BF16) {                                                                               \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(qq)], pack_bf16x8(B0[0], B0[1]), acc0, 0, 0, 0);    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(qq)], pack_bf16x8(B1[0], B1[1]), acc1, 0, 0, 0);    \
    } else {                                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                    \
            PFM_MFMA_PAIR(acc0, acc1, a[2 * (qq) + kk], B0[kk], B1[kk]);                                        \
        }                                                                                                      \
    }
    // the last pair of a jet whose tile count is odd holds one real tile: only that tile's MFMAs are issued (fp32 pipe only)
#define PFM_MFMAQ1(B0, B1, qq)                                                                                 \
    if constexpr (!BF16) {                                                                                     \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                     \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].x, B0[kk].x, acc0, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].y, B0[kk].y, acc0, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].z, B0[kk].z, acc0, 0, 0, 0);          \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * (qq) + kk].w, B0[kk].w, acc0, 0, 0, 0);          \
        }                                                                                                      \
    }
    PFM_LOADQ(X0, X1, src, 0);
    if (RESID) {
        r0 = *reinterpret_cast<const f32x4*>(resid + ooff);
        r1 = *reinterpret_cast<const f32x4*>(resid + TILE * H + ooff);
    }
    // epilogue of an INTERIOR pair: all 32 rows are < n_rows, no predication at all
    auto epilogue_full = [&](f32x4 e0, f32x4 e1, int pair, f32x4& ps, auto late_tag) {  // ps: the pool sum this pair adds to
        constexpr bool late = decltype(late_tag)::value;
#ifdef PFM_AB_NOEPI  // (timing-only ablation: results dropped, the accumulators kept alive)
        asm volatile("" ::"v"(e0), "v"(e1));
        return;
#endif
        // packed, no canonicalising copies (pfm_common.h).  Interior pairs (LATE): called behind the first K-quarter of the next pair
        if (late) lrelu8_pk<false>(e0, e1, slope2);
        else lrelu8_pk<true>(e0, e1, slope2);
        if constexpr (RB16) {
            float* db = dst + pair * 2 * TILE * SROW + oofb;
            *reinterpret_cast<s16x4*>(db) = pack_bf16(e0);
            *reinterpret_cast<s16x4*>(db + TILE * SROW) = pack_bf16(e1);
            if constexpr (RESID) {  // (RB16: the phases with a residual input are the ones whose rows also stay as fp32: dst32 != nullptr)
                float* d0 = dst32 + pair * 2 * TILE * H + ooff;
                *reinterpret_cast<f32x4*>(d0) = e0;
                *reinterpret_cast<f32x4*>(d0 + TILE * H) = e1;
            }
        } else {
            float* d0 = dst + pair * 2 * TILE * H + ooff;
            *reinterpret_cast<f32x4*>(d0) = e0;
            *reinterpret_cast<f32x4*>(d0 + TILE * H) = e1;
        }
        if (SAVE) {
            const int p0 = pair * 2 * TILE + pl;
            *reinterpret_cast<f32x4*>(save + p0 * H + 4 * oslot) = e0;
            *reinterpret_cast<f32x4*>(save + (p0 + TILE) * H + 4 * oslot) = e1;
        }
        if (POOL) {
            const float* mp = maskf + pair * 2 * TILE + pl;
            if (NSEG == 2) {  // mask1 = the mask on the second jet's rows, 0 on the first's
                const float* mq = mask1 + pair * 2 * TILE + pl;
                const float q0 = mq[0], q1 = mq[TILE];
                psumB += e0 * q0;
                psumB += e1 * q1;
                ps += e0 * (mp[0] - q0);
                ps += e1 * (mp[TILE] - q1);
            } else {
                pool2_pk(ps, e0, e1, mp[0], mp[TILE]);  // ps = fma(e1, m1, fma(e0, m0, ps)): the same two FMAs per element
            }
        }
    };
    // epilogue of the LAST pair (straight-line too: rows >= n_rows store to a sink and add 0 to the pool)
    auto epilogue = [&](f32x4 e0, f32x4 e1, int pair) {
        const int p0 = pair * 2 * TILE + pl, p1 = p0 + TILE;
        const bool v0 = p0 < n_rows, v1 = p1 < n_rows;
#ifdef PFM_AB_NOEPI
        asm volatile("" ::"v"(e0), "v"(e1));
        return;
#endif
        lrelu8_pk(e0, e1, slope2);
        if constexpr (RB16) {
            float* b0 = v0 ? dst + pair * 2 * TILE * SROW + oofb : sink;
            float* b1 = v1 ? dst + (pair * 2 + 1) * TILE * SROW + oofb : sink;
            *reinterpret_cast<s16x4*>(b0) = pack_bf16(e0);
            *reinterpret_cast<s16x4*>(b1) = pack_bf16(e1);
            if constexpr (RESID) {  // (RB16: the phases with a residual input are the ones whose rows also stay as fp32: dst32 != nullptr)
                float* d0 = v0 ? dst32 + pair * 2 * TILE * H + ooff : sink;
                float* d1 = v1 ? dst32 + (pair * 2 + 1) * TILE * H + ooff : sink;
                *reinterpret_cast<f32x4*>(d0) = e0;
                *reinterpret_cast<f32x4*>(d1) = e1;
            }
        } else {
        float* d0 = v0 ? dst + pair * 2 * TILE * H + ooff : sink;
        float* d1 = v1 ? dst + (pair * 2 + 1) * TILE * H + ooff : sink;
        *reinterpret_cast<f32x4*>(d0) = e0;
        *reinterpret_cast<f32x4*>(d1) = e1;
        }
        if (SAVE) {
            if (v0) *reinterpret_cast<f32x4*>(save + p0 * H + 4 * oslot) = e0;
            if (v1) *reinterpret_cast<f32x4*>(save + p1 * H + 4 * oslot) = e1;
        }
        if (POOL) {
            const float m0 = maskf[v0 ? p0 : 0], m1 = maskf[v1 ? p1 : 0];
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (NSEG == 2) {
                const float q0 = mask1[v0 ? p0 : 0], q1 = mask1[v1 ? p1 : 0];
                psumB += v0 ? e0 * q0 : z;
                psumB += v1 ? e1 * q1 : z;
                psum += v0 ? e0 * (m0 - q0) : z;
                psum += v1 ? e1 * (m1 - q1) : z;
            } else {
                psum += v0 ? e0 * m0 : z;
                psum += v1 ? e1 * m1 : z;
            }
        }
    };
    // one pair: hipcc sinks every ds_read down to its first use (read -> wait -> MFMA); the sched_barriers pin each
    // quarter's reads BEFORE the MFMA block of the previous quarter so their latency hides behind 16 MFMAs.
    // PFM_PAIR_BODY(MF, PFI): `pair` in scope; PFI(q) issues the prefetch load that rides behind quarter q.
#define PFM_PAIR_BODY(MF, PFI)                                                                                  \
    {                                                                                                           \
        const float* s0 = src + pair * 2 * TILE * SROW;                                                         \
        PFM_LOADQ(Y0, Y1, s0, 1);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        f32x4 acc0 = bias, acc1 = bias;                                                                         \
        if (NSEG == 2) { /* wave-uniform: which jet each of the two tiles belongs to */                         \
            if (2 * pair >= t1) acc0 = biasB;                                                                   \
            if (2 * pair + 1 >= t1) acc1 = biasB;                                                               \
        }                                                                                                       \
        if constexpr (NSEG == 4) { /* this pair IS jet `pair` */                                                \
            acc0 = acc1 = *reinterpret_cast<const f32x4*>(lds + qp->bj[pair < 4 ? pair : 3] + 4 * oslot);       \
        }                                                                                                       \
        if (RESID) { acc0 += r0; acc1 += r1; }                                                                  \
        MF(X0, X1, 0);                                                                                          \
        PFI(0);                                                                                                 \
        PFM_LOADQ(X0, X1, s0, 2);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        /* the epilogue of pair - 1 (<= npairs - 2: every row valid) HERE, behind this pair's first K-quarter: its accumulators are   \
           16 MFMAs old (no hazard nop, nothing waits for the matrix pipe) and its ~20 VALU / LDS instructions issue between the     \
           MFMAs of quarter 1 instead of in front of an idle pipe (round 4: tests/diag/fixed_cost_table.py) */                  \
        if (pair > 0) epilogue_full(pacc0, pacc1, pair - 1, PFM_PSUM_OF(pair - 1), std::true_type{});                 \
        MF(Y0, Y1, 1);                                                                                          \
        PFI(1);                                                                                                 \
        PFM_LOADQ(Y0, Y1, s0, 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        MF(X0, X1, 2);                                                                                          \
        PFI(2);                                                                                                 \
        /* first quarter of the NEXT pair (one pair past the end on the last iteration: inside the LDS window) */ \
        PFM_LOADQ(X0, X1, s0 + 2 * TILE * SROW, 0);                                                             \
        if (RESID) {                                                                                            \
            const float* rn = resid + (pair + 1) * 2 * TILE * H;                                                \
            r0 = *reinterpret_cast<const f32x4*>(rn + ooff);                                                    \
            r1 = *reinterpret_cast<const f32x4*>(rn + TILE * H + ooff);                                         \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        MF(Y0, Y1, 3);                                                                                          \
        PFI(3);                                                                                                 \
        pacc0 = acc0;                                                                                           \
        pacc1 = acc1;                                                                                           \
    }
    // The pairs whose two tiles are both real run as straight-line bodies with a compile-time pair index P (the loop is unrolled
    // so that the prefetch list's loads get static registers: one load behind every K-quarter); a jet with an odd tile count
    // ends with one more body that issues the real tile's MFMAs only (runtime pair index, no prefetch slot).
    const int nfull = NSEG == 4 ? npairs : phase_full_pairs<BF16>(n_rows);  // = ntiles / 2 (quad mode: whole 32-row slots)
    // the pool sum pair P adds to: the jet's own in quad mode (P is a constant expression at every use)
#define PFM_PSUM_OF(P) (NSEG == 4 ? psq[NSEG == 4 ? ((P) > 0 ? (P) : 0) : 0] : psum)
#define PFM_NOPF(q)
    /* Quad mode runs ALL four pair bodies whatever the number of jets in the workgroup (only a call's last workgroup holds fewer than
       four): a missing jet's slot holds stale rows whose results stay inside that slot -- rows never mix in a particle phase, its pool sum is
       its own and never finished -- and the bodies become one straight-line block: the run-time `P < npairs` tests cost ~30 register
       copies per body at the merge points (psq[], the staged operands), as much VALU work as the bf16 body itself. */
    // (PFM_PAIR_NEST: body P + 1 sits INSIDE the `if` of body P -- control leaves the chain once, at its end, instead of re-joining
    // behind every body: the register copies of the merge points run once per phase, not once per pair)
#if PFM_PAIR_NEST
#define PFM_PAIR_AT(P)                                                                                          \
    if (NSEG == 4 ? (P) < 4 : (P) < nfull) {                                                                    \
        {                                                                                                       \
            constexpr int pair = (P);                                                                           \
            PFM_PAIR_BODY(PFM_MFMAQ, PFM_PFI_##P)                                                               \
        }
#define PFM_PAIR_END }
#else
#define PFM_PAIR_AT(P)                                                                                          \
    if (NSEG == 4 ? (P) < 4 : (P) < nfull) {                                                                    \
        constexpr int pair = (P);                                                                               \
        PFM_PAIR_BODY(PFM_MFMAQ, PFM_PFI_##P)                                                                   \
    }
#define PFM_PAIR_END
#endif
#define PFM_PFI_0(q) pf.template issue_slot<0 + (q)>()
#define PFM_PFI_1(q) pf.template issue_slot<4 + (q)>()
#define PFM_PFI_2(q) pf.template issue_slot<8 + (q)>()
#define PFM_PFI_3(q) pf.template issue_slot<12 + (q)>()
#define PFM_PFI_4(q) pf.template issue_slot<16 + (q)>()
    static_assert(MAXPAIRS == 5, "unroll PFM_PAIR_AT to MAXPAIRS");
    PFM_PAIR_AT(0) PFM_PAIR_AT(1) PFM_PAIR_AT(2) PFM_PAIR_AT(3) PFM_PAIR_AT(4)
    PFM_PAIR_END PFM_PAIR_END PFM_PAIR_END PFM_PAIR_END PFM_PAIR_END
    if constexpr (NSEG != 4) {
        if (nfull < npairs) {  // odd tile count: one real tile in the last pair
            const int pair = nfull;
            PFM_PAIR_BODY(PFM_MFMAQ1, PFM_NOPF)
        }
    }
#undef PFM_PFI_0
#undef PFM_PFI_1
#undef PFM_PFI_2
#undef PFM_PFI_3
#undef PFM_PFI_4
#undef PFM_PAIR_AT
#undef PFM_PAIR_END
#undef PFM_PSUM_OF
#undef PFM_NOPF
#undef PFM_PAIR_BODY
#undef PFM_MFMAQ1
#undef PFM_LOADQ
#undef PFM_MFMAQ
#undef PFM_KOFF
    if constexpr (NSEG == 4) {
        // the last jet's slot: every row of a slot is computed (holes carry zero input and zero mask), so no predication; the
        // branch is wave-uniform and the pool sum's index static in each arm
        epilogue_full(pacc0, pacc1, 3, psq[3], std::false_type{});
    } else {
        epilogue(pacc0, pacc1, npairs - 1);
    }
    // the part of the prefetch list a short jet had no pairs for
    if (TAIL) pf.issue_tail(nfull);
#ifdef PFM_AB_NOPOOLFIN
    asm volatile("" ::"v"(psum));
    return;
#endif
    if (POOL) {
        if constexpr (NSEG == 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < npairs) pool_finish<false, VB16>(psq[s], j, lds, qp->vin[s], qp->misc[s], oslot, pl, nullptr);
        } else {
            pool_finish<SAVE, VB16>(psum, j, lds, c.vin, c.misc, oslot, pl, save_pool);
            if (NSEG == 2) pool_finish<false, VB16>(psumB, j, lds, s2.vin1, s2.misc1, oslot, pl, nullptr);
        }
    }
}

// The same particle phase on split-fp16 operands.  src / dst / resid are (hi, lo) fp16 plane pairs (the lo plane
// starts j.N * H halfs after the hi plane); inference only (no SAVE).
template <bool RESID, bool POOL>
__device__ __forceinline__ void gemm_phase_x3(const f32x4 (&a)[8], const float* __restrict__ src, float* __restrict__ dst,
                                              const float* __restrict__ resid, const float* __restrict__ bj,
                                              const float* __restrict__ maskf, const JetDims& j, float* __restrict__ lds,
                                              const Carve& c, int n_rows) {
    const int tid_ = launder(threadIdx.x);
    const int lane = tid_ & 63, w = tid_ >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int oslot = 4 * w + q;
    const float slope = j.slope;
    const int plane = j.N * H;
    h8 ah[4], al[4];  // a[] arrives in the k8 order of load_afrag_k8
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) x3_split8(a[2 * kp], a[2 * kp + 1], ah[kp], al[kp]);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bj + 4 * oslot);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    const int npairs = (n_rows + 2 * TILE - 1) / (2 * TILE);
    int koff[4];  // (row & 15) == pl for every tile: per-lane constants
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) koff[kp] = pl * H + (((4 * kp + q) ^ pl) << 3);
    const int ooff = pl * H + (((oslot >> 1) ^ pl) << 3) + ((oslot & 1) << 2);
    const _Float16* sh = reinterpret_cast<const _Float16*>(src);
    const _Float16* rh = reinterpret_cast<const _Float16*>(resid);
    _Float16* dh = reinterpret_cast<_Float16*>(dst);
    _Float16* const sink = reinterpret_cast<_Float16*>(lds + c.dummy);
    // Software pipeline, one straight-line block per pair of tiles: the operand reads of one K half are in flight while
    // the 12 MFMAs of the other half issue, and the epilogue of pair i-1 (split, ds_write, pooling) sits between the
    // reads and the MFMAs of pair i.  Rows are not clamped (see gemm_phase).
    struct Half { h8 b0h[2], b1h[2], b0l[2], b1l[2]; };
    auto load_half = [&](Half& hf, const _Float16* s0, int h) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int o = koff[2 * h + kk];
            hf.b0h[kk] = *reinterpret_cast<const h8*>(s0 + o);
            hf.b1h[kk] = *reinterpret_cast<const h8*>(s0 + TILE * H + o);
            hf.b0l[kk] = *reinterpret_cast<const h8*>(s0 + plane + o);
            hf.b1l[kk] = *reinterpret_cast<const h8*>(s0 + plane + TILE * H + o);
        }
    };
    f32x4 m0, m1, c0, c1;
    auto mfma_half = [&](const Half& hf, int h) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kp = 2 * h + kk;
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kp], hf.b0h[kk], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kp], hf.b1h[kk], m1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kp], hf.b0l[kk], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kp], hf.b1l[kk], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kp], hf.b0h[kk], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kp], hf.b1h[kk], c1, 0, 0, 0);
        }
    };
    auto epilogue = [&](f32x4 e0, f32x4 e1, int pair) {
        const int p0 = pair * 2 * TILE + pl, p1 = p0 + TILE;
        const bool v0 = p0 < n_rows, v1 = p1 < n_rows;
        e0 = lrelu4(e0, slope);
        e1 = lrelu4(e1, slope);
        h4 h0, l0, h1, l1;
        x3_split(e0, h0, l0);
        x3_split(e1, h1, l1);
        _Float16* d0 = v0 ? dh + pair * 2 * TILE * H + ooff : sink;
        _Float16* d1 = v1 ? dh + (pair * 2 + 1) * TILE * H + ooff : sink;
        *reinterpret_cast<h4*>(d0) = h0;
        *reinterpret_cast<h4*>(d0 + (v0 ? plane : 4)) = l0;
        *reinterpret_cast<h4*>(d1) = h1;
        *reinterpret_cast<h4*>(d1 + (v1 ? plane : 4)) = l1;
        if (POOL) {
            const float mk0 = maskf[v0 ? p0 : 0], mk1 = maskf[v1 ? p1 : 0];
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            psum += v0 ? e0 * mk0 : z;
            psum += v1 ? e1 * mk1 : z;
        }
    };
    Half X, Y;
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, pe0 = r0, pe1 = r0;
    load_half(X, sh, 0);
    if (RESID) {
        r0 = x3_join(*reinterpret_cast<const h4*>(rh + ooff), *reinterpret_cast<const h4*>(rh + ooff + plane));
        r1 = x3_join(*reinterpret_cast<const h4*>(rh + TILE * H + ooff), *reinterpret_cast<const h4*>(rh + TILE * H + ooff + plane));
    }
    for (int pair = 0; pair < npairs; ++pair) {
        const _Float16* s0 = sh + pair * 2 * TILE * H;
        load_half(Y, s0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (pair > 0) epilogue(pe0, pe1, pair - 1);
        m0 = bias; m1 = bias;
        if (RESID) { m0 += r0; m1 += r1; }
        c0 = f32x4{0.f, 0.f, 0.f, 0.f}; c1 = c0;
        mfma_half(X, 0);
        load_half(X, s0 + 2 * TILE * H, 0);  // next pair (one pair past the end on the last iteration: inside the LDS window)
        if (RESID) {
            const _Float16* rn = rh + (pair + 1) * 2 * TILE * H + ooff;
            r0 = x3_join(*reinterpret_cast<const h4*>(rn), *reinterpret_cast<const h4*>(rn + plane));
            r1 = x3_join(*reinterpret_cast<const h4*>(rn + TILE * H), *reinterpret_cast<const h4*>(rn + TILE * H + plane));
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(Y, 1);
        pe0 = m0 + c0 * X3_DN;
        pe1 = m1 + c1 * X3_DN;
    }
    epilogue(pe0, pe1, npairs - 1);
    if (POOL) pool_finish<false>(psum, j, lds, c.vin, c.misc, oslot, pl, nullptr);
}

// ---- per-jet GEMVs (global MLP, per-jet biases) ------------------------------------------------
// Every 128-output block is KM16 (pfm_hip.h): thread t of 512 reads float4 number t of each 16-row panel, i.e.
// thread (og = t >> 4, pt = t & 15) owns outputs 4*og..4*og+3 and rows k = 16*i + pt.  The 16 partial sums of
// one output group sit in 16 adjacent lanes of ONE wave, so the reduction is four xor-shuffles -- no LDS round
// trip, no barrier -- and wave w ends up with exactly its own output slice [16w, 16w+16).
template <int U>
__device__ __forceinline__ void gemv4_load(f32x4 (&wv)[U], blob_rsrc rs, int64_t W_off, int K, int base, int tid) {
    // Unconditional: panels past K read whatever follows in the blob (finite weights; past the end the buffer
    // resource returns 0) and are multiplied by x = 0 in gemv4_fma.  A predicate here costs a branch, four v_mov
    // and, worse, a vmcnt(0) per load in hipcc's code.
    (void)K;
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = bload4(rs, W_off + (base + u) * (NT * 4), tid * 16);
}
template <int U>
__device__ __forceinline__ void gemv4_fma(f32x4& acc, const f32x4 (&wv)[U], const float* __restrict__ vin, int K,
                                          int base, int pt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = 16 * (base + u) + pt;
        const float x = k < K ? vin[k] : 0.f;
        acc += wv[u] * x;
    }
}
__device__ __forceinline__ f32x4 reduce_pt(f32x4 v) { return row_sum16(v); }  // the 16 lanes of an output group

struct LocalBiasSrc {  // the two local linears whose per-jet bias a stage prepares
    int64_t We1, b1, We2, b2;
};

// Stem: bj1 / bj2 = b + We^T [temb ; cond_l] for fc_l1 / fc_l2 (K = T + Cl <= 96).  Ends with a barrier.
template <int NSEG = 1>
__device__ __forceinline__ void stem_bias(const float* __restrict__ blob, blob_rsrc rs, const LocalBiasSrc& lb, int Ke,
                                          float* __restrict__ lds, const SegView (&sv)[2]) {
    const int tid = launder(threadIdx.x), og = tid >> 4, pt = tid & 15;
    f32x4 w1[2], w2[2];
    const f32x4 b1 = bload4(rs, lb.b1, og * 16);
    const f32x4 b2 = bload4(rs, lb.b2, og * 16);
    f32x4 p1[NSEG], p2[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) p1[s] = p2[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int base = 0; 16 * base < Ke; base += 2) {
        gemv4_load<2>(w1, rs, lb.We1, Ke, base, tid);
        gemv4_load<2>(w2, rs, lb.We2, Ke, base, tid);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            gemv4_fma<2>(p1[s], w1, lds + sv[s].vin, Ke, base, pt);
            gemv4_fma<2>(p2[s], w2, lds + sv[s].vin, Ke, base, pt);
        }
    }
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const f32x4 q1 = reduce_pt(p1[s]), q2 = reduce_pt(p2[s]);
        if (pt == 0) {
            *reinterpret_cast<f32x4*>(lds + sv[s].bj1 + 4 * og) = q1 + b1;
            *reinterpret_cast<f32x4*>(lds + sv[s].bj2 + 4 * og) = q2 + b2;
        }
    }
    __syncthreads();
}

// Register windows of the per-jet phase (all requested during the particle phases before it, see Prefetch):
constexpr int NGL = 20;   // panels of fc_global1 (K1 <= 16 * (tp + NGL); wider inputs: the rest the slow way)
constexpr int NGL1 = 12;  // ... of which the first NGL1 ride on particle phase 1, the rest on phase 2
constexpr int NWA = 3;    // panels of local linear 1's extras [temb ; cond_l ; g] (T + Cl + L <= 48; with TB what follows the time rows)
constexpr int NWB = 2;    // panels of local linear 2's extras [temb ; cond_l]

// Sampling evaluates every jet at the same time t, so the time columns of the per-jet Linears (the first T rows of every
// extras / fc_global block: inputs are ordered [temb ; ...]) give jet-independent vectors.  epic_time_table_kernel
// computes them once per evaluation for all layers; with TB the per-jet phase skips the tp = T / 16 time panels of every
// block (a quarter of its weight fetches, which is what bounds it) and adds the table entry to the bias instead.
constexpr int TB_SLOT = 512;  // floats per (evaluation, layer): fc_global1 | local-1 extras | local-2 extras | fc_global2 (16)
constexpr int TB_G1 = 0, TB_L1 = 128, TB_L2 = 256, TB_G2 = 384;

// row_ror:4 / row_ror:8 sums: every lane ends with the sum over the 4 lanes of its DPP row that share (lane & 3)
__device__ __forceinline__ float row_sum_stride4(float v) {
    v += dpp_move<0x124>(v);  // row_ror:4
    v += dpp_move<0x128>(v);  // row_ror:8
    return v;
}

// The per-jet phase between two particle phases:
//   g1 = lrelu(Wg1.[temb;cond;mean;sum;g] + b)            epic.py:180-182 / :375-377
//   g  = lrelu(Wg2.[temb;cond;g1] + b (+ g))              epic.py:184-186 / :378-380
//   bj1 = b1 + We1.[temb;cond_l;g],  bj2 = b2 + We2.[temb;cond_l]     (folded t/cond/global columns)
// In : vin = [temb;cond;mean;sum;g_old] complete (pooled part written by the previous particle phase, barrier
//      passed), vin2[0..T+C) = [temb;cond]; gl / wbA / wbB = this thread's rows of fc_global1 / the two extras blocks, requested
//      during the particle phases before (epic_body).
// fc_global1 and the bias GEMVs: thread (og = t >> 4, pt = t & 15) owns outputs 4 og.. of rows 16 i + pt; the 16 partial sums of
// an output group sit in one DPP row.  fc_global2 (L <= 16 outputs, K2 <= 208 rows): thread (kq = t >> 2, o4 = t & 3) takes rows
// kq and kq + 128 -- ONE or two 16-byte loads per thread instead of every wave fetching the whole block --, a wave sums its 16 rows
// with two DPP rotations and two cross-row shuffles, the 8 wave partials meet in LDS (second barrier) and every wave adds them up
// in the same fixed order.  From there on everything is wave-local: each wave finishes the bias slice [16w,16w+16) that its own
// MFMA phase reads, so the next particle phase starts without another barrier.  Out: vin.g = g_new (written by wave 0), bj1/bj2.
// STEM: fc_g1/fc_g2 (no g input, no residual, no local biases) and a trailing barrier (vin.g is read next).
// NSEG == 2: two jets in the workgroup -- the weights are in registers once, every per-jet step runs for both (views sv[0], sv[1]).
template <bool STEM, bool SAVE, bool TB = false, int NSEG = 1>
__device__ __forceinline__ void per_jet_phase(const JetDims& j, const float* __restrict__ blob, blob_rsrc rs,
                                              const pfm_dense_lin& gl1, const pfm_dense_lin& gl2,
                                              const LocalBiasSrc& lb, float* __restrict__ lds, const SegView (&sv)[2],
                                              float* __restrict__ save_g1, float* __restrict__ save_g,
                                              const f32x4 (&gl)[NGL], const f32x4 (&wbA)[NWA], const f32x4 (&wbB)[NWB],
                                              const float* __restrict__ tb = nullptr) {
    static_assert(!(STEM && TB), "the stem keeps its time rows");
    static_assert(!(SAVE && NSEG != 1), "training keeps one jet per workgroup");
    const int tp = TB ? (j.T >> 4) : 0;  // time panels skipped in every block
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int og = tid >> 4, pt = tid & 15;   // fc_global1 / bias GEMVs
    const int o4r = lane >> 4, part = lane & 15;  // after the second barrier: DPP row o4r owns outputs 4*o4r..
    const int TC = j.T + j.C, Ke = j.T + j.Cl;
    const int K1 = TC + 2 * H + (STEM ? 0 : j.L);
    const int K2 = TC + H;
    const int Ka = Ke + j.L;  // rows of local linear 1's extras: [temb ; cond_l ; g]
    constexpr int NGLu = TB ? NGL - 2 : NGL, KA = TB ? 1 : NWA, KBp = TB ? 1 : NWB;  // the part of each window a TB call uses
    const bool has_b = !TB || 16 * tp < Ke;  // local-2 extras beyond the time rows (cond_l): none for unconditioned jets
    if (!STEM) PFM_MARK(0);
    // ---- the few loads left to this phase: this thread's rows of fc_global2 and the bias vectors ----
    const int kq = tid >> 2, o4 = tid & 3;
    const int r0 = 16 * tp + kq, r1 = r0 + 128;  // rows of fc_global2 (KP16 [k][16]) this thread multiplies
    const f32x4 w2a = bload4(rs, gl2.W + (int64_t)16 * tp * 16, (kq * 16 + 4 * o4) * 4);
    f32x4 w2b = {0.f, 0.f, 0.f, 0.f};
    if (K2 > 16 * tp + 128) w2b = bload4(rs, gl2.W + (int64_t)(16 * tp + 128) * 16, (kq * 16 + 4 * o4) * 4);  // wave-uniform
    f32x4 bg1 = bload4(rs, gl1.b, og * 16);
    f32x4 bg2 = bload4(rs, gl2.b, o4r * 16);  // padded to 16 entries
    f32x4 bl1 = {0.f, 0.f, 0.f, 0.f}, bl2 = bl1;
    if (!STEM) {
        bl1 = bload4(rs, lb.b1, og * 16);
        bl2 = bload4(rs, lb.b2, og * 16);
    }
    if (TB) {  // the time terms of the four per-jet Linears, tabulated once per evaluation (epic_time_table_kernel)
        bg1 += *reinterpret_cast<const f32x4*>(tb + TB_G1 + 4 * og);
        bg2 += *reinterpret_cast<const f32x4*>(tb + TB_G2 + 4 * o4r);
        bl1 += *reinterpret_cast<const f32x4*>(tb + TB_L1 + 4 * og);
        bl2 += *reinterpret_cast<const f32x4*>(tb + TB_L2 + 4 * og);
    }
    f32x4 gold[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        gold[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!STEM) gold[s] = *reinterpret_cast<const f32x4*>(lds + sv[s].vin + TC + 2 * H + 4 * o4r);  // g_old, before anyone overwrites it
    }
    if (!STEM) PFM_MARK(1);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float* vin = lds + sv[s].vin;
        float* vin2 = lds + sv[s].vin2;
        // ---- fc_global1 ----
        // (round 4) the window's inputs read unconditionally and all before the first FMA, rows >= K1 selected to zero: the guarded
        // form `k < K1 ? vin[k] : 0` compiled to a branch + a dependent ds_read_b32 round trip per panel (tests/diag/isa_mix.py)
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        float xg[NGLu];
#pragma unroll
        for (int u = 0; u < NGLu; ++u) xg[u] = vin[16 * (tp + u) + pt];  // (< VIN_FLOATS: the windows cover whole 16-row panels)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NGLu; ++u) {
            const int k = 16 * (tp + u) + pt;
            p += gl[u] * (k < K1 ? xg[u] : 0.f);
        }
        for (int base = NGLu + tp; 16 * base < K1; base += 4) {  // wider models: the rest the slow way
            f32x4 wa[4];
            gemv4_load<4>(wa, rs, gl1.W, K1, base, tid);
            gemv4_fma<4>(p, wa, vin, K1, base, pt);
        }
        if (!STEM && s == 0) PFM_MARK(2);
        // 16-lane reductions; g1 -> vin2; local bias 2 (t / cond only) for this wave's own output slice
        p = reduce_pt(p);
        if (pt == 0) {
            const f32x4 g1 = lrelu4(p + bg1, j.slope);
            *reinterpret_cast<f32x4*>(vin2 + TC + 4 * og) = g1;
            if (SAVE) *reinterpret_cast<f32x4*>(save_g1 + 4 * og) = g1;
        }
        if (!STEM) {
            f32x4 p2 = {0.f, 0.f, 0.f, 0.f};
            if (has_b) {
                float xb[KBp];
#pragma unroll
                for (int u = 0; u < KBp; ++u) xb[u] = vin[16 * (tp + u) + pt];
#pragma unroll
                for (int u = 0; u < KBp; ++u) {
                    const int k = 16 * (tp + u) + pt;
                    p2 += wbB[u] * (k < Ke ? xb[u] : 0.f);
                }
                for (int base = KBp + tp; 16 * base < Ke; base += 1) {
                    f32x4 wx[1];
                    gemv4_load<1>(wx, rs, lb.We2, Ke, base, tid);
                    gemv4_fma<1>(p2, wx, vin, Ke, base, pt);
                }
                p2 = reduce_pt(p2);
            }
            if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].bj2 + 4 * og) = p2 + bl2;
        }
    }
    if (!STEM) PFM_MARK(3);
    __syncthreads();
    if (!STEM) PFM_MARK(4);
    // ---- fc_global2: rows split over all 512 threads, wave partials through LDS ----
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float* vin2 = lds + sv[s].vin2;
        const float v2a = vin2[r0], v2b = vin2[min(r1, VIN2_FLOATS - 1)];  // unconditional reads, selected below
        f32x4 gp = w2a * (r0 < K2 ? v2a : 0.f);
        gp += w2b * (r1 < K2 ? v2b : 0.f);
        gp.x = row_sum_stride4(gp.x); gp.y = row_sum_stride4(gp.y); gp.z = row_sum_stride4(gp.z); gp.w = row_sum_stride4(gp.w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gp[e] += __shfl_xor(gp[e], 16);
            gp[e] += __shfl_xor(gp[e], 32);
        }
        if (lane < 4) *reinterpret_cast<f32x4*>(lds + sv[s].g2p + MAXL * w + 4 * o4) = gp;  // lane = o4 here (kq & 15 == 0)
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float* vin = lds + sv[s].vin;
        f32x4 gpart[NW];  // the eight wave partials requested together, added in wave order
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) gpart[ww] = *reinterpret_cast<const f32x4*>(lds + sv[s].g2p + MAXL * ww + 4 * o4r);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 gn = gpart[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) gn += gpart[ww];
        if (!STEM && s == 0) PFM_MARK(5);
        gn += bg2;
        if (!STEM) gn += gold[s];  // residual before the activation, epic.py:184-186
        gn = lrelu4(gn, j.slope);
        // each wave keeps its own copy of g_new in LDS (read back below as the tail of the extras vector); wave 0's
        // copy is vin.g itself, the input of the next stage
        float* gcopy = (w == 0) ? lds + sv[s].vin + TC + 2 * H : lds + sv[s].gcopy + MAXL * w;
        if (part == 0) {
            *reinterpret_cast<f32x4*>(gcopy + 4 * o4r) = gn;
            if (SAVE && w == 0) *reinterpret_cast<f32x4*>(save_g + 4 * o4r) = gn;
        }
        if (!STEM) {
            // local bias 1 = b1 + We1 . [temb ; cond_l ; g_new] for this wave's slice: row k = 16 i + pt of the extras
            // takes temb/cond from vin (k < Ke) or g_new[k - Ke] from the wave's copy (same wave wrote it: LDS is in order)
            f32x4 p1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KA; ++i) {
                const int k = 16 * (i + tp) + pt;
                const float xv = vin[k], xc = gcopy[min(max(k - Ke, 0), MAXL - 1)];  // both read unconditionally, one selected
                const float x = k < Ke ? xv : (k < Ka ? xc : 0.f);
                p1 += wbA[i] * x;
            }
            for (int base = KA + tp; 16 * base < Ka; ++base) {  // wider extras than the register window
                f32x4 wx[1];
                gemv4_load<1>(wx, rs, lb.We1, Ka, base, tid);
                const int k = 16 * base + pt;
                p1 += wx[0] * (k < Ke ? vin[k] : (k < Ka ? gcopy[k - Ke] : 0.f));
            }
            p1 = reduce_pt(p1);
            if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].bj1 + 4 * og) = p1 + bl1;
        }
    }
    if (!STEM) PFM_MARK(6);
    if (STEM) __syncthreads();
}

// masked pooling tail of a particle phase: 16-lane tree, then mean / scaled sum straight into vin
// VB16 (the lean bf16 sampler's MFMA chains, epic_fast.h): mean / scaled sum go as bf16 into the jet's chain vector [mean (128) ; sum
// (128) ; g (16) ; 0 (16)] at vin_off (it takes the place of the fp32 [temb ; cond ; mean ; sum] slots, which that path does not read)
template <bool SAVE, bool VB16>
__device__ __forceinline__ void pool_finish(f32x4 psum, const JetDims& j, float* __restrict__ lds, int vin_off, int misc_off,
                                            int oslot, int pl, float* __restrict__ save_pool) {
    psum = row_sum16(psum);
    if (pl == 0) {
        const float nvalid = lds[misc_off], rinv = lds[misc_off + 2];
        const int TC = j.T + j.C;
        // mean = sum / n_valid (epic.py:161/:370), correctly rounded like the reference's division, without the 12-instruction
        // v_div_scale / v_div_fmas / v_div_fixup sequence per element: q = RN(a r) with r = RN(1 / n), one residual step
        // q' = RN(q + r RN(a - n q)) (Markstein: q' is the correctly rounded quotient when r is the correctly rounded reciprocal;
        // checked exhaustively over n = 1 .. 160 on 3.2 M random numerators, tests/test_x3_emulation_cpu.py::test_markstein_quotient).
        // n = 0 (no valid particle): r = inf, a = 0 -> NaN, as the reference's 0 / 0.
        f32x4 mean;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef PFM_AB_OLD_DIV  // (diagnostic A/B builds only)
            mean[i] = psum[i] / nvalid;
            continue;
#endif
            const float q = __fmul_rn(psum[i], rinv);
            mean[i] = __builtin_fmaf(__builtin_fmaf(-q, nvalid, psum[i]), rinv, q);
        }
        if constexpr (VB16) {
            *reinterpret_cast<s16x4*>(lds + vin_off + 2 * oslot) = pack_bf16(mean);                  // bf16 element 4 oslot ..
            *reinterpret_cast<s16x4*>(lds + vin_off + H / 2 + 2 * oslot) = pack_bf16(psum * j.sscale);  // bf16 element 128 + 4 oslot ..
        } else {
            *reinterpret_cast<f32x4*>(lds + vin_off + TC + 4 * oslot) = mean;
            *reinterpret_cast<f32x4*>(lds + vin_off + TC + H + 4 * oslot) = psum * j.sscale;  // epic.py:162/:371
        }
        if (SAVE) *reinterpret_cast<f32x4*>(save_pool + 4 * oslot) = psum;
    }
}

template <int FM, bool SAVE, int MODE = 0, int NSEG = 1>
__device__ __forceinline__ void stem_l1(const pfm_epic_desc& d, const JetDims& j, const float* __restrict__ blob,
                                        float* __restrict__ lds, const Carve& c, int n_rows,
                                        float* __restrict__ save_x1, int bj1_seg1 = 0, int r1 = 1 << 20) {
    const int tid = launder(threadIdx.x), slot = tid & 31;
    const float* Wx = blob + d.l1x.W;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + c.bj1 + 4 * slot);
    f32x4 b4B = b4;
    if (NSEG == 2) b4B = *reinterpret_cast<const f32x4*>(lds + bj1_seg1 + 4 * slot);
    f32x4 wv[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) wv[f] = *reinterpret_cast<const f32x4*>(Wx + min(f, j.F - 1) * H + 4 * slot);
    for (int p = tid >> 5; p < n_rows; p += NT / 32) {
        f32x4 acc = (NSEG == 2 && p >= r1) ? b4B : b4;
#pragma unroll
        for (int f = 0; f < FM; ++f)
            if (f < j.F) acc += wv[f] * lds[c.yin + p * j.F + f];
        acc = lrelu4(acc, j.slope);
        if (MODE == 2) {
            h4 hi, lo;
            x3_split(acc, hi, lo);
            _Float16* dp = reinterpret_cast<_Float16*>(lds + c.bufA) + x3_off(p, slot);
            *reinterpret_cast<h4*>(dp) = hi;
            *reinterpret_cast<h4*>(dp + j.N * H) = lo;
        } else {
            *reinterpret_cast<f32x4*>(lds + c.bufA + lds_off(p, slot)) = acc;
        }
        if (SAVE) *reinterpret_cast<f32x4*>(save_x1 + p * H + 4 * slot) = acc;
    }
}

// Full network body up to (excluding) the fc_l3 head.  Preconditions (in LDS): yin (N x F input),
// maskf, misc[0] = sum(mask), vin.temb, vin.cond.  Postcondition: bufB holds the last hidden state.
// Weight traffic (per layer ~330 KB through the CU's 64 B/clk path) is spread over the particle phases (Prefetch):
//   fc_l2 phase     : the stem's fc_g1 panels
//   per-jet phase k : A fragments of fc_local1[k] (issued at its head, consumed by phase 1 right behind it)
//   phase 1 of k    : A fragments of fc_local2[k], first NGL1 panels of fc_global1[k+1]
//   phase 2 of k    : the other panels of fc_global1[k+1], the extras panels of fc_local1/2[k+1]
// NSEG == 2 (the packed sampler): the rows [0, n_rows) hold two jets, `sg` says where the second starts; everything per-particle
// sees one set of n_rows rows with holes, everything per-jet runs for both (SegView).
template <bool SAVE, int MODE = 0, bool TB = false, int NSEG = 1>
__device__ __forceinline__ void epic_body(const pfm_epic_desc& d, const JetDims& j,
                                          const float* __restrict__ blob, float* __restrict__ lds,
                                          const Carve& c, int n_rows, float* __restrict__ saved,
                                          const SavedLayout& sl, const float* __restrict__ tb = nullptr,
                                          const Segs* sg = nullptr) {
    static_assert(NSEG == 1 || (MODE != 2 && !SAVE), "two jets per workgroup: fp32 / bf16 inference kernels only");
    const SegView sv[2] = {seg_view(c, j.N, 0), seg_view(c, j.N, NSEG == 2 ? 1 : 0)};
    Seg2Phase s2l1, s2l2;  // the second jet as the two local Linears of a stage see it
    if (NSEG == 2) {
        s2l1.bj = lds + sv[1].bj1; s2l2.bj = lds + sv[1].bj2;
        s2l1.mask1 = s2l2.mask1 = lds + sv[1].maskf;
        s2l1.t1 = s2l2.t1 = sg->r1 / TILE;
        s2l1.vin1 = s2l2.vin1 = sv[1].vin;
        s2l1.misc1 = s2l2.misc1 = sv[1].misc;
    }
    const int tp = TB ? (j.T >> 4) : 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* bufA = lds + c.bufA;
    float* bufB = lds + c.bufB;
    float* bj1 = lds + c.bj1;
    float* bj2 = lds + c.bj2;
    const float* maskf = lds + c.maskf;
    const int Ke = j.T + j.Cl;
    constexpr int NGLu = TB ? NGL - 2 : NGL, KA = TB ? 1 : NWA, KBp = TB ? 1 : NWB;  // as in per_jet_phase

    f32x4 a1[8], a2[8];
    f32x4 gl[NGL], wbA[NWA], wbB[NWB];  // per-jet GEMV windows of the NEXT per-jet phase (see per_jet_phase)
    PFM_STAMP(1);
    // ---- stem: per-jet biases of fc_l1 / fc_l2 (t / cond columns) ------------------------------
    const blob_rsrc rs = make_blob_rsrc(blob, d.blob_floats + PFM_DESC_FLOATS);
    load_afrag_m<MODE>(a2, rs, d.l2.A, w, lane);
    {
        LocalBiasSrc lb; lb.We1 = d.l1_We; lb.b1 = d.l1_b; lb.We2 = d.l2.We; lb.b2 = d.l2.b;
        stem_bias<NSEG>(blob, rs, lb, Ke, lds, sv);
    }
    PFM_STAMP(2);
    // ---- fc_l1 (K = F, VALU): bufA[p][o] = lrelu(bj1[o] + sum_f Wx[f][o] * y[p][f])  epic.py:360-362
    if (j.F <= 4) stem_l1<4, SAVE, MODE, NSEG>(d, j, blob, lds, c, n_rows, saved + sl.x1, sv[1].bj1, NSEG == 2 ? sg->r1 : 0);
    else stem_l1<MAXF, SAVE, MODE, NSEG>(d, j, blob, lds, c, n_rows, saved + sl.x1, sv[1].bj1, NSEG == 2 ? sg->r1 : 0);
    __syncthreads();
    PFM_STAMP(3);
    // ---- fc_l2: bufB = lrelu(W.bufA + bj2 + bufA)  epic.py:364-366 (residual from the source buffer)
    {
        Prefetch<NGL> pf{rs, gl, nullptr, nullptr, nullptr, seg_panels(d.g1.W, 0, tid), {}, {}, {}};
        if (MODE == 2) {
            pf.template issue_range<0, NGL>();
            gemm_phase_x3<true, true>(a2, bufA, bufB, bufA, bj2, maskf, j, lds, c, n_rows);
        } else {
            gemm_phase<true, true, SAVE, MODE == 1, decltype(pf), NSEG>(a2, bufA, bufB, bufA, bj2, maskf, j, lds, c, saved + sl.x2,
                                                                         saved + sl.pool, n_rows, pf, s2l2);
        }
    }
    __syncthreads();
    PFM_STAMP(4);
    // ---- fc_g1 / fc_g2 (epic.py:369-380) ---------------------------------------------------------
    {
        LocalBiasSrc none; none.We1 = none.b1 = none.We2 = none.b2 = 0;
        per_jet_phase<true, SAVE, false, NSEG>(j, blob, rs, d.g1, d.g2, none, lds, sv, saved + sl.gstem1, saved + sl.gstem, gl, wbA, wbB);
        if (j.layers > 0) {  // the first layer's windows: no particle phase in between, exposed once per evaluation
            const pfm_epic_layer& l0 = d.layer[0];
            Prefetch<NGLu, KA, KBp> pf{rs, gl, wbA, wbB, nullptr, seg_panels(l0.gl1.W, tp, tid), seg_panels(l0.lc1.We, tp, tid),
                                       seg_panels(l0.lc2.We, tp, tid), {}};
            pf.template issue_range<0, NGLu + KA + KBp>();
        }
    }
    // ---- EPiC layers (epic.py:382-385 -> :159-203) -----------------------------------------------
    for (int k = 0; k < j.layers; ++k) {
        const pfm_epic_layer ly = d.layer[k];  // by value: all 24 offset dwords in one batch of scalar loads
        const pfm_epic_layer& nx = d.layer[k + 1 < j.layers ? k + 1 : k];  // last layer: its own blocks again (harmless, hidden)
        PFM_STAMP(10);
        // vin still holds mean / sum of the current hidden state (bufB) and g
        load_afrag_m<MODE>(a1, rs, ly.lc1.A, w, lane);  // phase-1 weights: land behind the per-jet phase
        LocalBiasSrc lb; lb.We1 = ly.lc1.We; lb.b1 = ly.lc1.b; lb.We2 = ly.lc2.We; lb.b2 = ly.lc2.b;
        per_jet_phase<false, SAVE, TB, NSEG>(j, blob, rs, ly.gl1, ly.gl2, lb, lds, sv, saved + sl.glayer + k * sl.gstride,
                                             saved + sl.glayer + k * sl.gstride + H, gl, wbA, wbB, TB ? tb + k * TB_SLOT : nullptr);
        PFM_STAMP(12);
        // phase 1: bufA = lrelu(W1 . bufB + bj1)                       epic.py:194-196
        if (MODE == 2) {
            load_afrag_m<MODE>(a2, rs, ly.lc2.A, w, lane);
            Prefetch<NGL1> pf{rs, gl, nullptr, nullptr, nullptr, seg_panels(nx.gl1.W, tp, tid), {}, {}, {}};
            pf.template issue_range<0, NGL1>();
            gemm_phase_x3<false, false>(a1, bufB, bufA, nullptr, bj1, maskf, j, lds, c, n_rows);
        } else {
            Prefetch<8, NGL1> pf{rs, a2, gl, nullptr, nullptr, seg_afrag(ly.lc2.A, w, lane), seg_panels(nx.gl1.W, tp, tid), {}, {}};
            gemm_phase<false, false, SAVE, MODE == 1, decltype(pf), NSEG>(a1, bufB, bufA, nullptr, bj1, maskf, j, lds, c,
                                                                           saved + sl.l1 + k * sl.lstride, nullptr, n_rows, pf, s2l1);
        }
        __syncthreads();
        PFM_STAMP(13);
        // phase 2: bufB = lrelu(W2 . bufA + bj2 + bufB), pooled -> vin    epic.py:198-200, :160-162
        {
            Prefetch<NGLu - NGL1, KA, KBp> pf{rs, gl + NGL1, wbA, wbB, nullptr, seg_panels(nx.gl1.W, tp + NGL1, tid),
                                              seg_panels(nx.lc1.We, tp, tid), seg_panels(nx.lc2.We, tp, tid), {}};
            if (MODE == 2) {
                pf.template issue_range<0, NGLu - NGL1 + KA + KBp>();
                gemm_phase_x3<true, true>(a2, bufA, bufB, bufB, bj2, maskf, j, lds, c, n_rows);
            } else {
                gemm_phase<true, true, SAVE, MODE == 1, decltype(pf), NSEG>(a2, bufA, bufB, bufB, bj2, maskf, j, lds, c,
                                                                             saved + sl.xo + k * sl.lstride,
                                                                             saved + sl.pool + (k + 1) * sl.pstride, n_rows, pf, s2l2);
            }
        }
        __syncthreads();
    }
    PFM_STAMP(20);
}

// fc_l3 head: emit(p, f, lrelu(b3[f] + We3.[temb;cond_l] + W3[f].x[p]) * mask[p]) for every p < N
// (rows >= n_rows are emitted as 0: they are masked).  epic.py:387-391
// The F <= 16 outputs are one 16-row MFMA panel: wave w takes the particle tiles w, w+8, ...; lane (particle, q)
// ends up with features 4q..4q+3 of its particle.
template <int MODE = 0, int NSEG = 1, typename Emit>
__device__ __forceinline__ void epic_head(const pfm_epic_desc& d, const JetDims& j,
                                          const float* __restrict__ blob, float* __restrict__ lds,
                                          const Carve& c, int n_rows, Emit emit, const Segs* sg = nullptr) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const float* bufB = lds + c.bufB;
    const float* vin = lds + c.vin;
    float* bj3 = lds + c.bj1;  // reuse
    const int Ke = j.T + j.Cl;
    const blob_rsrc rs = make_blob_rsrc(blob, d.blob_floats + PFM_DESC_FLOATS);
    f32x4 a[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) a[kt] = bload4(rs, d.l3_A + kt * 256, lane * 16);
    // bj3[f] = b3[f] + sum_k We3[k][f] * e[k]: wave w takes features w, w+8; lanes split k
    const SegView sv1 = seg_view(c, j.N, NSEG == 2 ? 1 : 0);
    float* bj3B = lds + sv1.bj1;  // second jet: in its own bj1
    for (int f = w; f < j.F; f += NW) {
        float s = 0.f, sB = 0.f;
        for (int k = lane; k < Ke; k += 64) {
            const float wv = blob[d.l3_We + k * j.F + f];
            s = fmaf(wv, vin[k], s);
            if (NSEG == 2) sB = fmaf(wv, lds[sv1.vin + k], sB);
        }
        for (int m = 32; m >= 1; m >>= 1) {
            s += __shfl_xor(s, m);
            if (NSEG == 2) sB += __shfl_xor(sB, m);
        }
        if (lane == 0) {
            bj3[f] = s + blob[d.l3_b + f];
            if (NSEG == 2) bj3B[f] = sB + blob[d.l3_b + f];
        }
    }
    __syncthreads();
    int koff[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) koff[kt] = pl * H + (((4 * kt + q) ^ pl) << 2);
    const int ntiles = (n_rows + TILE - 1) / TILE;
    f32x4 b3 = {0.f, 0.f, 0.f, 0.f}, b3B = b3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b3[r] = (4 * q + r < j.F) ? bj3[4 * q + r] : 0.f;
        if (NSEG == 2) b3B[r] = (4 * q + r < j.F) ? bj3B[4 * q + r] : 0.f;
    }
    const int t1 = NSEG == 2 ? sg->r1 / TILE : (1 << 20);
    for (int tile = w; tile < ntiles; tile += NW) {
        const int p = tile * TILE + pl;
        const float* s0 = bufB + tile * TILE * H;
        f32x4 acc = (NSEG == 2 && tile >= t1) ? b3B : b3;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            f32x4 b;
            if (MODE == 2) {  // (hi, lo) fp16 planes -> fp32 (the head is 1 % of the work: keep its fp32 MFMA)
                const _Float16* hp = reinterpret_cast<const _Float16*>(bufB) + x3_off(tile * TILE + pl, 4 * kt + q);
                b = x3_join(*reinterpret_cast<const h4*>(hp), *reinterpret_cast<const h4*>(hp + j.N * H));
            } else {
                b = *reinterpret_cast<const f32x4*>(s0 + koff[kt]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].w, b.w, acc, 0, 0, 0);
        }
        if (p < j.N) {
            const bool live = p < n_rows;
            const float m = live ? lds[c.maskf + p] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * q + r;
                if (f < j.F) emit(p, f, live ? lrelu(acc[r], j.slope) * m : 0.f);
            }
        }
    }
    // rows behind the last computed tile (skip mode): masked, emit zeros
    for (int i = ntiles * TILE * j.F + tid; i < j.N * j.F; i += NT) emit(i / j.F, i % j.F, 0.f);
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
    if (tid < j.C) {  // (cond_jet == nullptr: the lean evaluation of conditioned jets keeps zeros there, epic_fast.h)
        const float cv = cond_jet ? cond_jet[tid] : 0.f;
        lds[c.vin + j.T + tid] = cv;
        lds[c.vin2 + j.T + tid] = cv;
    }
    if (tid >= 64 && tid < 64 + MAXL) lds[c.vin + j.T + j.C + 2 * H + (tid - 64)] = 0.f;
    for (int m = 32; m >= 1; m >>= 1) {
        cnt += __shfl_xor(cnt, m);
        last = max(last, __shfl_xor(last, m));
    }
    float* red = lds + c.misc + 8;
    if ((tid & 63) == 0) { red[tid >> 6] = cnt; red[8 + (tid >> 6)] = (float)last; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, l = -1.f;
        for (int i = 0; i < NW; ++i) { s += red[i]; l = fmaxf(l, red[8 + i]); }
        lds[c.misc] = s;
        lds[c.misc + 1] = l;
        lds[c.misc + 2] = 1.0f / s;  // correctly rounded reciprocal of the valid count, for pool_finish's quotient
    }
    __syncthreads();
    int n_rows = j.N;
    // a jet without any valid particle is NaN in the reference (epic.py:370 divides by 0): compute all rows then too
    if ((d.flags & PFM_F_SKIP_MASKED_TAIL) && lds[c.misc + 1] >= 0.f) n_rows = (int)lds[c.misc + 1] + 1;
    return n_rows;
}

// vin.temb = a row of a caller-supplied embedding (t_emb="gaussian": computed by the caller's trainable embedding network)
__device__ __forceinline__ void epic_time_embedding_from(const JetDims& j, float* __restrict__ lds, const Carve& c,
                                                         const float* __restrict__ row, bool second_jet = false) {
    const int tid = threadIdx.x;
    if (tid < j.T) {
        const float e = row[tid];
        lds[c.vin + tid] = e;
        lds[c.vin2 + tid] = e;
        if (second_jet) {
            const SegView v1 = seg_view(c, j.N, 1);
            lds[v1.vin + tid] = e;
            lds[v1.vin2 + tid] = e;
        }
    }
}

// vin.temb[k] = cos(((t + 0) * freqs[k]) * pi / 1)  -- exact fp32 op order of time_emb.py:96
__device__ __forceinline__ void epic_time_embedding(const pfm_epic_desc& d, const JetDims& j,
                                                    const float* __restrict__ blob, float* __restrict__ lds,
                                                    const Carve& c, float t, bool second_jet = false) {
    const int tid = threadIdx.x;
    if (tid < j.T) {
        const float f = blob[d.freqs + tid];
        float e;
        if (d.flags & PFM_F_TEMB_SINCOS) {  // flow_matching_module.py:208-211: cat(cos(f t), sin(f t)), f = 2^k pi (table holds [f ; f])
            const float arg = __fmul_rn(f, t);
            e = 2 * tid < j.T ? cosf(arg) : sinf(arg);
        } else {
            const float arg = __fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(t, 0.0f), f), 3.14159274101257324f), 1.0f);
            e = cosf(arg);
        }
        lds[c.vin + tid] = e;
        lds[c.vin2 + tid] = e;
        if (second_jet) {  // the packed sampler: both jets of the workgroup are at the same time
            const SegView v1 = seg_view(c, j.N, 1);
            lds[v1.vin + tid] = e;
            lds[v1.vin2 + tid] = e;
        }
    }
}

}  // namespace pfm
