// The samplers' lean evaluation ("fast path": the persistent midpoint and Runge-Kutta integrators of epic_kernels.hip) for the
// headline configuration of the EPiC network: unconditioned jets (cond_global = cond_local = 0), time embedding of width 32,
// F <= 4 features, fp32 or bf16 matrix operands.
//
// Reference graph: the same as epic_nfe.h (particle_fm/models/components/epic.py:304-391, :85-203).
//
// Why a second body.  Ablations on MI355X (tests/diag/ab_time.py on builds with pieces removed) showed where a 2-tile jet's
// 128k cycles per evaluation go: 53k matrix-pipe issue, 32k the six per-jet chains (pool -> fc_global1 -> fc_global2 -> local
// biases), 20k weight loads that stall wave issue, the rest stem / head / phase boundaries.  The fp32 MFMA does not co-issue
// with VALU work, so every VALU instruction is paid in full -- and the generic chain (any T / C / Cl / L, two jets per
// workgroup, SAVE for training) compiles to ~1000 instructions per thread: guards for rows past K, loops for wider models,
// exec-mask branches around every conditional store, serialized LDS round trips, SGPR spills.  Overlapping it with the MFMA
// phase changes nothing (measured: the chain issued behind the K-quarters of phase 1 took the same time as ahead of it).
// With everything per-jet that depends on time alone tabulated per evaluation, this configuration needs ~170:
//   * the table (epic_time_table_kernel, fast format) holds for every evaluation and layer the COMPLETE bias of each per-jet
//     Linear's time columns (b + W_t^T temb), and a stem slot with the per-jet biases of fc_l1 / fc_l2 / fc_l3 and the time terms
//     of fc_g1 / fc_g2: no time embedding (a Payne-Hanek cosf on arguments up to 1e13), no stem GEMVs, no head GEMV, and two
//     barriers fewer per evaluation;
//   * no guards: every input vector is zero-padded in LDS to the 16-row panels the weights are padded to;
//   * fc_global2's rows sit along the DPP row (16 k of one output group per row): its reduction is row_sum16, no bpermute;
//   * bias vectors that are constant per (evaluation, layer) are read by the MFMA phases straight from the table; the chain's own
//     three table rows are fetched a phase ahead by 68 threads and published to a 288-float LDS area behind the carve;
//   * fc_l1 (K = F <= 4) is ONE v_mfma_f32_16x16x4_f32 per tile and wave instead of a VALU layer;
//   * the head's weights and the next evaluation's fc_l2 weights are requested before they are needed.
// The layer chain keeps the generic chain's arithmetic order (same panels, same reduction trees, same wave-partial order), so
// a layer gives the same bits as the generic time-table path; the stem's tabulated terms and the MFMA fc_l1 differ from the
// generic kernels by fp32 re-association (~1e-7), inside every parity bar (tests/test_hip_forward.py, test_hip_fast.py).
// Two jets per workgroup (PFM_F_PACK_JETS): fast_eval<.., NSEG = 2> -- every tabulated bias is the same for both jets (no
// conditioning), only the chains run per jet; same bits as one jet per workgroup (tests/test_hip_packed.py).
#pragma once
#include "epic_nfe.h"

namespace pfm {

#ifndef PFM_BCHAIN
#define PFM_BCHAIN 1          // bf16 descriptors: the lean sampler's chains on the matrix pipe (diagnostic builds: 0 = the fp32 VALU chains)
#endif
#ifndef PFM_RB16
#define PFM_RB16 1            // bf16 lean samplers: 1 = activations resident as bf16 planes (gemm_phase<.., RB16>); 0 (diagnostic builds) = fp32 rows, converted at every read
#endif
#ifndef PFM_QG1_BATCH
#define PFM_QG1_BATCH 9       // fc_global1 of the KQ16 chains: input panels read per batch and jet (one jet; two jets: half); 0 = hipcc's own order
#endif
#ifndef PFM_QCHAIN_MINSEG
#define PFM_QCHAIN_MINSEG 1   // jets per workgroup from which the chains run on the KQ16 / WQ16 copies (diagnostic builds: 2, 5 = never)
#endif
constexpr int FT = 32;        // time-embedding width the fast path is built for
constexpr int FTP = FT / 16;  // time panels of every per-jet block (tabulated, skipped)
constexpr int FNG = 17;       // fc_global1 panels behind the time rows: [mean(128) ; sum(128) ; g(16)]
constexpr int FNGC = 18;      // ... of a conditioned model: [cond(C <= 16, multiplied by zeros) ; mean ; sum ; g(16)] = up to 288 rows
constexpr int FNGS = 16;      // fc_g1 (stem): [mean ; sum]
// stem slot of the fast table (slot index = layers): per-jet biases of fc_l1 / fc_l2, time terms (+ bias) of fc_g1 / fc_g2, fc_l3 bias
constexpr int TB_SJ1 = 0, TB_SJ2 = 128, TB_SG1 = 256, TB_SG2 = 384, TB_SB3 = 400;

// Conditioned jets (cond_global = C > 0): every conditioning column multiplies a per-jet constant, so its contribution to each per-jet
// Linear is one more table, per JET instead of per evaluation (epic_cond_table_kernel: W_c^T cond, same slot layout), added to the
// time table's row wherever that is read; the conditioning slots of the input vectors hold zeros (the weight panels run over them).
// One jet per workgroup, and cond_local + latent <= 16 (local linear 1's [cond_l ; g] rows fit one panel).
__host__ __device__ inline bool fast_path_ok(const pfm_epic_desc& d) {
    const bool cond_ok = d.cond_global == 0 || (d.cond_local + d.latent <= 16 && !(d.flags & PFM_F_PACK_JETS));
    return d.t_dim == FT && cond_ok && d.features <= 4 && d.layers > 0 &&
           !(d.flags & (PFM_F_F16X3_MFMA | PFM_F_GENERIC_SAMPLER)) &&
           ((int64_t)make_carve(d.n_points, d.features).total + 288) * 4 <= 163840;  // + TBL_FLOATS behind the carve
}

// ---- FOUR short jets per workgroup (quad mode; PFM_F_QUAD_JETS on a 128-row descriptor, unconditioned jets) -------------------------
// A jet of <= 32 particles is all fixed cost (two 16-row tiles against ~2 MB of weights per evaluation and seven serial per-jet chains),
// and even the packed pair leaves a 1024-jet batch at two rounds of 512 workgroups.  Quad mode gives every jet a fixed 32-row slot --
// jet s owns rows [32 s, 32 s + 32) = tile pair s of every particle phase -- so the pair bodies know their jet at compile time (bias,
// pool sum: gemm_phase<.., NSEG = 4>, epic_nfe.h), nothing is predicated (rows behind a jet's last particle are holes: zero input,
// zero mask, finite, never pooled) and the chains run for the four jets from ONE set of weight registers.  The per-jet vectors of jets
// 1..3 live behind the carve and the chain's table rows.  Same arithmetic per row and per jet as one jet per workgroup: same bits.
constexpr int QROWS = 32, QJETS = 4;
constexpr int QV_VIN = 0, QV_VIN2 = QV_VIN + VIN_FLOATS, QV_BJ1 = QV_VIN2 + VIN2_FLOATS, QV_GCOPY = QV_BJ1 + H, QV_G2P = QV_GCOPY + NW * MAXL,
              QV_MISC = QV_G2P + NW * MAXL, QV_FLOATS = QV_MISC + 8;
constexpr int QUAD_TILE_ROWS = QROWS * QJETS;  // desc.n_points of a quad call
__host__ __device__ inline int quad_lds_floats(int F) { return make_carve(QUAD_TILE_ROWS, F).total + 288 + (QJETS - 1) * QV_FLOATS; }
__host__ __device__ inline SegView quad_view(const Carve& c, int s) {
    if (s == 0) return SegView{c.vin, c.vin2, c.bj1, c.bj2, c.gcopy, c.g2p, c.misc, c.maskf};
    const int b = c.total + 288 + (s - 1) * QV_FLOATS;  // behind TBL_FLOATS
    return SegView{b + QV_VIN, b + QV_VIN2, b + QV_BJ1, c.bj2, b + QV_GCOPY, b + QV_G2P, b + QV_MISC, c.maskf};
}
__host__ __device__ inline bool quad_path_ok(const pfm_epic_desc& d) {
    return (d.flags & PFM_F_QUAD_JETS) && d.n_points == QUAD_TILE_ROWS && d.cond_global == 0 && fast_path_ok(d) &&
           (int64_t)quad_lds_floats(d.features) * 4 <= 163840;
}

// fc_l1: bufA[p][o] = lrelu(bj1[o] + sum_f Wx[f][o] * y[p][f])   epic.py:360-362, on the matrix pipe (K = F padded to 4).
// aw = this lane's element of the A operand (fast_l1_weight: constant over the call), bias = its slice of the per-jet bias (table).
__device__ __forceinline__ float fast_l1_weight(const pfm_epic_desc& d, const JetDims& j, const float* __restrict__ blob) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pl = lane & 15, q = lane >> 4;
    return q < j.F ? blob[d.l1x.W + q * H + 16 * w + pl] : 0.f;  // A[i = pl][k = q] = Wx[k][16 w + i]
}
// RB16 (bf16-resident activations, gemm_phase): x1 goes as fp32 to bufB (fc_l2's residual input, overwritten in place by its output) and
// as bf16 to plane A (fc_l2's matrix operand).
template <bool RB16 = false>
__device__ __forceinline__ void fast_stem_l1(const JetDims& j, float* __restrict__ lds, const Carve& c, int n_rows, float aw, f32x4 bias) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const int oslot = 4 * w + q;
    const bool kf = q < j.F;
    const float* yin = lds + c.yin;
    const int ntiles = (n_rows + TILE - 1) / TILE;
    for (int t = 0; t < ntiles; ++t) {
        const int p = t * TILE + pl;
        const float bv = (kf && p < n_rows) ? yin[p * j.F + q] : 0.f;  // B[k = q][j = pl]
        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv, bias, 0, 0, 0);
        acc = lrelu4(acc, j.slope);
        if (p < n_rows) {
            if constexpr (RB16) {
                *reinterpret_cast<f32x4*>(lds + c.bufB + lds_off(p, oslot)) = acc;
                *reinterpret_cast<s16x4*>(lds + c.bufA + p * (H / 2) + (((4 * (w >> 1) + q) ^ pl) << 2) + 2 * (w & 1)) = pack_bf16(acc);
            } else {
                *reinterpret_cast<f32x4*>(lds + c.bufA + lds_off(p, oslot)) = acc;
            }
        }
    }
}

// The few loads of a per-jet chain that are not weight panels: this thread's row of fc_global2, and the tabulated bias + time terms
// of the chain's three Linears.  Requested a whole particle phase ahead (a load waited for right behind its request costs an L2 round
// trip with the matrix pipe idle -- and a vmcnt wait also waits for every weight load queued before it).  The table rows (272 floats
// per chain) do not stay in registers for that phase: 68 threads fetch one float4 each (`stg`) and publish it to a small LDS area
// behind the carve (`tbl`) right before the barrier that precedes the chain, which reads its slices with ds_read (no vmcnt at all).
constexpr int TBL_G1 = 0, TBL_L1 = 128, TBL_G2 = 256, TBL_FLOATS = 288;  // LDS floats behind Carve::total
static_assert(TBL_FLOATS == 288, "quad_view / quad_lds_floats place the extra jets' vectors behind it");
constexpr bool L2LDS = true;  // measured: 0.3-1 % faster than the table read, same bits
struct ChainLoads {
    f32x4 w2;   // row FT + 16 w + pt of fc_global2 (KP16), outputs 4 o4..
    f32x4 stg;  // threads 0..67: one float4 of [G1 (128) | L1 (128) | G2 (16)]
};
// slot: the table slot of the chain; o_g1 / o_l1 / o_g2: where its three rows start inside the slot (layer slot: TB_G1 / TB_L1 /
// TB_G2; stem slot: TB_SG1 / (none: pass o_g1) / TB_SG2)
// Threads 68..99 fetch the local linear 2 row (o_l2), which the particle phase reads from LDS (c.bj2): straight from the time table it was a
// global load waited for at the head of the phase, an L2 round trip with the matrix pipe idle (L2LDS = false: the old way).
// ct (conditioned jets, or nullptr): the jet's cond-table slot with the same row offsets, added to every row.
__device__ __forceinline__ ChainLoads fast_chain_loads(blob_rsrc rs, int64_t gl2_W, int w2_row0, const float* __restrict__ slot, int o_g1,
                                                       int o_l1, int o_g2, const float* __restrict__ ct = nullptr, int o_l2 = 0) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int pt = tid & 15, o4 = lane >> 4;
    ChainLoads L;
    L.w2 = bload4(rs, gl2_W + (int64_t)w2_row0 * 16, ((16 * w + pt) * 16 + 4 * o4) * 4);  // row w2_row0 + 16 w + pt (KP16), outputs 4 o4..
    L.stg = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < ((ct || L2LDS) ? 100 : 68)) {
        const int off = tid < 32 ? o_g1 + 4 * tid : (tid < 64 ? o_l1 + 4 * (tid - 32) : (tid < 68 ? o_g2 + 4 * (tid - 64) : o_l2 + 4 * (tid - 68)));
        L.stg = *reinterpret_cast<const f32x4*>(slot + off);
        if (ct) L.stg += *reinterpret_cast<const f32x4*>(ct + off);
    }
    return L;
}
// a barrier must separate this from the previous chain's last read of tbl, and another one from the next chain's reads
template <typename CL>
__device__ __forceinline__ void fast_chain_publish(const CL& L, float* __restrict__ tbl) {
    const int tid = launder(threadIdx.x);
    if (tid < 68) *reinterpret_cast<f32x4*>(tbl + (tid < 64 ? 4 * tid : TBL_G2 + 4 * (tid - 64))) = L.stg;
}
// conditioned jets: the local-linear-2 bias row of threads 68..99 -> bj2.  Readers (the particle phase) must lie behind a later barrier,
// the previous phase's reads of bj2 behind an earlier one.
template <typename CL>
__device__ __forceinline__ void fast_chain_publish_l2(const CL& L, float* __restrict__ bj2) {
    const int tid = launder(threadIdx.x);
    if (tid >= 68 && tid < 100) *reinterpret_cast<f32x4*>(bj2 + 4 * (tid - 68)) = L.stg;
}

// sum of the eight wave partials of fc_global2 in wave order; all eight reads in flight before the first add
__device__ __forceinline__ f32x4 fast_sum_partials(const float* __restrict__ g2p_o4) {
    f32x4 part[NW];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) part[ww] = *reinterpret_cast<const f32x4*>(g2p_o4 + MAXL * ww);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 gn = part[0];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) gn += part[ww];
    return gn;
}

// Stem chain: g = lrelu(Wg2 . lrelu(Wg1 . [mean ; sum] + tg1) + tg2), tg* = bias + time term (table).  epic.py:369-380
// In: vin.mean / vin.sum (written by the fc_l2 phase, barrier passed), gl = this thread's rows of fc_g1 behind the time rows.
// after_fc1(): called once gl has been consumed (the caller requests the first layer's windows there).  Out: vin.g.  Ends with a barrier.
template <int NSEG, bool COND, int NSV, typename After, typename Publish>
__device__ __forceinline__ void fast_chain_stem(const JetDims& j, float* __restrict__ lds, const SegView (&sv)[NSV],
                                                const f32x4 (&gl)[COND ? FNGC : FNG], const ChainLoads& L, const float* __restrict__ tbl,
                                                After after_fc1, Publish publish_next) {
    static_assert(!(COND && NSEG != 1), "conditioned jets: one jet per workgroup");
    static_assert(NSEG <= NSV, "a view per jet");
    constexpr int NP = COND ? FNGS + 1 : FNGS;  // [cond (zeros in vin) ;] mean ; sum
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int og = tid >> 4, pt = tid & 15, o4 = lane >> 4;
    const int TC = COND ? FT + j.C : FT;
    f32x4 p[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float* vp = lds + sv[s].vin + FT + pt;
        p[s] = gl[0] * vp[0];
#pragma unroll
        for (int u = 1; u < NP; ++u) p[s] += gl[u] * vp[16 * u];
    }
    after_fc1();
    const f32x4 bg1 = *reinterpret_cast<const f32x4*>(tbl + TBL_G1 + 4 * og);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        p[s] = reduce_pt(p[s]);
        if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].vin2 + TC + 4 * og) = lrelu4(p[s] + bg1, j.slope);
    }
    wave_lds_sync();  // fc_g2 below reads the 16 values of g1 THIS wave has just written (rows 16 w .. 16 w + 15): no workgroup barrier
    const f32x4 bg2 = *reinterpret_cast<const f32x4*>(tbl + TBL_G2 + 4 * o4);  // before the next chain's rows replace these
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        f32x4 gp = L.w2 * lds[sv[s].vin2 + TC + 16 * w + pt];
        gp = row_sum16(gp);
        if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].g2p + MAXL * w + 4 * o4) = gp;
    }
    __syncthreads();
    publish_next();  // every read of this chain's table rows lies before the barrier above
    if (w == 0) {
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            f32x4 gn = fast_sum_partials(lds + sv[s].g2p + 4 * o4);
            gn = lrelu4(gn + bg2, j.slope);
            if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].vin + TC + 2 * H + 4 * o4) = gn;
        }
    }
    __syncthreads();
}

// Layer chain (epic.py:159-190): g1 = lrelu(Wg1.[mean;sum;g] + tg1); g = lrelu(Wg2.g1 + tg2 + g); bj1 = tl1 + We1.g
// (t* = bias + time term of the table slot).  Same panels, reduction trees and wave-partial order as PerJet phase of epic_nfe.h.
// In: vin = [.. ; mean ; sum ; g_old], gl = rows of fc_global1 behind the time rows, wbA = row FT + pt of local linear 1's extras.
// Out: vin.g = g_new, bj1 (each wave its own slice: the particle phase that follows needs no barrier).
// NSEG == 2 (two jets in the workgroup, the packed sampler): the weights are in registers once, every step runs for both jets.
// COND: the input vectors carry C zeroed conditioning slots behind the time slots (the pooled part starts at TC = FT + C).  L holds
// this layer's local-linear-2 bias row for c.bj2 (published at the end of the chain).
template <int NSEG, bool COND, int NSV>
__device__ __forceinline__ void fast_chain_layer(const JetDims& j, float* __restrict__ lds, const Carve& c, const SegView (&sv)[NSV],
                                                 const f32x4 (&gl)[COND ? FNGC : FNG], const f32x4& wbA, const ChainLoads& L,
                                                 const float* __restrict__ tbl) {
    static_assert(!(COND && NSEG != 1), "conditioned jets: one jet per workgroup");
    constexpr int NP = COND ? FNGC : FNG;
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int og = tid >> 4, pt = tid & 15, o4 = lane >> 4;
    const int TC = COND ? FT + j.C : FT;
    f32x4 gold[NSEG], p[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float* vin = lds + sv[s].vin;
        gold[s] = *reinterpret_cast<const f32x4*>(vin + TC + 2 * H + 4 * o4);  // before anyone overwrites it
        const float* vp = vin + FT + pt;
        p[s] = gl[0] * vp[0];
#pragma unroll
        for (int u = 1; u < NP; ++u) p[s] += gl[u] * vp[16 * u];
    }
    const f32x4 bg1 = *reinterpret_cast<const f32x4*>(tbl + TBL_G1 + 4 * og);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        p[s] = reduce_pt(p[s]);
        if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].vin2 + TC + 4 * og) = lrelu4(p[s] + bg1, j.slope);
    }
    // fc_global2's rows are split over the waves exactly as fc_global1's outputs are: wave w multiplies g1[16 w .. 16 w + 15], the
    // values its own lanes have just written.  LDS operations of one wave execute in order: no workgroup barrier here (round 4).
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        f32x4 gp = L.w2 * lds[sv[s].vin2 + TC + 16 * w + pt];
        gp = row_sum16(gp);
        if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].g2p + MAXL * w + 4 * o4) = gp;
    }
    __syncthreads();
    const f32x4 bg2 = *reinterpret_cast<const f32x4*>(tbl + TBL_G2 + 4 * o4);
    const f32x4 bl1 = *reinterpret_cast<const f32x4*>(tbl + TBL_L1 + 4 * og);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        f32x4 gn = fast_sum_partials(lds + sv[s].g2p + 4 * o4);
        gn += bg2;
        gn += gold[s];  // residual before the activation, epic.py:184-186
        gn = lrelu4(gn, j.slope);
        // each wave keeps its own copy of g_new (read back as the input of the bias GEMV: same wave, LDS is in order); wave 0's copy
        // is vin.g itself, the input of the next stage
        float* gcopy = (w == 0) ? lds + sv[s].vin + TC + 2 * H : lds + sv[s].gcopy + MAXL * w;
        if (pt == 0) *reinterpret_cast<f32x4*>(gcopy + 4 * o4) = gn;
        // local linear 1's extras behind the time rows: [cond_l (Cl, tabulated: multiplied by 0 here) ; g]; entries >= L of g are
        // lrelu(0) = 0 (zero-padded weights and biases)
        float x;
        if (COND) {
            const int gx = pt - j.Cl;
            x = gx >= 0 ? gcopy[gx] : 0.f;
        } else {
            x = gcopy[pt];
        }
        f32x4 p1 = wbA * x;
        p1 = reduce_pt(p1);
        if (pt == 0) *reinterpret_cast<f32x4*>(lds + sv[s].bj1 + 4 * og) = p1 + bl1;
    }
    // local linear 2's bias row -> c.bj2, last: its vmcnt wait covers the phase-1 weights requested behind the row, which the particle
    // phase right behind this chain waits for anyway (the previous particle phase read the old row two barriers ago)
    if (COND || L2LDS) fast_chain_publish_l2(L, lds + c.bj2);
}

// ---- round 4: the chains of UNCONDITIONED jets on the KQ16 / WQ16 copies of the per-jet blocks (include/pfm_hip.h) ------------------
// Why.  tests/diag/fixed_cost_table.py: the seven chains are 29 % (10 tiles) .. 53 % (2 tiles) of what an evaluation spends outside
// the matrix pipe, and 38 % of the whole cfg-2 bf16 sampler (four jets per workgroup: every VALU instruction of a chain runs once per
// jet).  In the KM16 layout a thread holds FOUR outputs of ONE input row: four accumulators, a 16-lane reduction tree over each
// (16 v_add_f32_dpp), a broadcast copy per second panel, LeakyReLU on four values -- ~137 VALU instructions per chain and jet, of which
// 34 are the packed FMAs that do the work.  KQ16: thread (o = t >> 2, kq = t & 3) holds four CONSECUTIVE input rows of ONE output, the
// input vector comes as a float4 (ds_read_b128), two packed FMAs per panel with both operands as vectors, ONE add and a quad
// reduction (2 DPP adds), LeakyReLU on one value: ~65.  Same weights, another summation order than fast_chain_* / per_jet_phase
// (fp32 re-association; inside every parity bar, and identical for one, two or four jets per workgroup).
__device__ __forceinline__ float quad_sum(float v) {  // every lane of a quad ends with the quad's sum
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
// w . x over this thread's four rows: (w.x x.x + w.z x.z) + (w.y x.y + w.w x.w), packed
__device__ __forceinline__ f32x2 pk_dot_step(f32x2 acc, const f32x4& wv, const f32x4& x) {
    acc = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{x.x, x.y}, acc);
    return __builtin_elementwise_fma(f32x2{wv.z, wv.w}, f32x2{x.z, x.w}, acc);
}
// the chain's loads that are not register windows: this thread's float4 of fc_global2 (WQ16: chunk w, float4 lane) and the staged
// table rows (as fast_chain_loads)
// ct (conditioned jets, or nullptr): the jet's cond-table slot with the same row offsets, added to every row (as fast_chain_loads)
__device__ __forceinline__ ChainLoads fastq_chain_loads(blob_rsrc rs, int64_t q_gl2, const float* __restrict__ slot, int o_g1, int o_l1,
                                                        int o_g2, int o_l2, const float* __restrict__ ct = nullptr) {
    const int tid = launder(threadIdx.x);
    ChainLoads L;
    L.w2 = bload4(rs, q_gl2, tid * 16);
    L.stg = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 100) {
        const int off = tid < 32 ? o_g1 + 4 * tid : (tid < 64 ? o_l1 + 4 * (tid - 32) : (tid < 68 ? o_g2 + 4 * (tid - 64) : o_l2 + 4 * (tid - 68)));
        L.stg = *reinterpret_cast<const f32x4*>(slot + off);
        if (ct) L.stg += *reinterpret_cast<const f32x4*>(ct + off);
    }
    return L;
}
// g1 = lrelu(W1 . v + t1) for NP panels of v (from vin + FT), written to vin2 + FT by the quad's first lane; wave w produces g1[16 w ..]
// Four jets per workgroup: the input vectors come through a two-deep register pipeline of one panel per jet (32 VGPRs); left to
// itself hipcc requests all 4 x 17 float4s up front.
template <int NSEG, int NP, int NSV, int S0 = 0>
__device__ __forceinline__ void fastq_g1_part(const JetDims& j, float* __restrict__ lds, const SegView (&sva)[NSV], const f32x4 (&gl)[FNG],
                                              const float* __restrict__ tbl) {
    const SegView* sv = sva + S0;  // jets S0 .. S0 + NSEG - 1 of the workgroup
    const int tid = launder(threadIdx.x), o = tid >> 2, kq = tid & 3;
    constexpr int G = NSEG == 4 ? 1 : (NSEG == 2 ? 2 : 4), NG = (NP + G - 1) / G;
    f32x4 xa[NSEG][G], xb[NSEG][G];
    f32x2 acc[NSEG];
    const float* vp[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        acc[s] = f32x2{0.f, 0.f};
        vp[s] = lds + sv[s].vin + FT + 4 * kq;
    }
#define PFM_Q_LOAD(X, g)                                                                            \
    _Pragma("unroll") for (int s = 0; s < NSEG; ++s)                                                \
        _Pragma("unroll") for (int i = 0; i < G; ++i)                                               \
            if ((g) * G + i < NP) X[s][i] = *reinterpret_cast<const f32x4*>(vp[s] + 16 * ((g) * G + i));
#define PFM_Q_FMA(X, g)                                                                             \
    _Pragma("unroll") for (int s = 0; s < NSEG; ++s)                                                \
        _Pragma("unroll") for (int i = 0; i < G; ++i)                                               \
            if ((g) * G + i < NP) acc[s] = pk_dot_step(acc[s], gl[(g) * G + i], X[s][i]);
    if constexpr (NSEG <= 2) {
        // One or two jets: the input panels in batches of QB float4 per jet, every read of a batch requested before its first FMA
        // (a scheduling fence pins that).  Left to itself hipcc (round 4, at 244 VGPRs) reuses ONE register quad for all 17 panels --
        // ds_read_b128, s_waitcnt lgkmcnt(0), two FMAs, 17 times over: 17 dependent LDS round trips, ~1.5 k of a chain's 2.7 k cycles
        // (tests/diag/isa_mix.py listing; the stamps of profiles/round4_chain_rider_experiment.txt).
        constexpr int QB = PFM_QG1_BATCH > 0 ? (NSEG == 1 ? PFM_QG1_BATCH : (PFM_QG1_BATCH + 1) / 2) : NP;
        if constexpr (PFM_QG1_BATCH > 0) {
            f32x4 xq[NSEG][QB];
#pragma unroll
            for (int u0 = 0; u0 < NP; u0 += QB) {
#pragma unroll
                for (int s = 0; s < NSEG; ++s)
#pragma unroll
                    for (int i = 0; i < QB; ++i)
                        if (u0 + i < NP) xq[s][i] = *reinterpret_cast<const f32x4*>(vp[s] + 16 * (u0 + i));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NSEG; ++s)
#pragma unroll
                    for (int i = 0; i < QB; ++i)
                        if (u0 + i < NP) acc[s] = pk_dot_step(acc[s], gl[u0 + i], xq[s][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSEG; ++s)
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[s] = pk_dot_step(acc[s], gl[u], *reinterpret_cast<const f32x4*>(vp[s] + 16 * u));
        }
    } else {
        PFM_Q_LOAD(xa, 0)
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            if (g + 1 < NG) { PFM_Q_LOAD(xb, g + 1) }
            __builtin_amdgcn_sched_barrier(0);
            PFM_Q_FMA(xa, g)
            if (g + 2 < NG) { PFM_Q_LOAD(xa, g + 2) }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) { PFM_Q_FMA(xb, g + 1) }
        }
    }
#undef PFM_Q_LOAD
#undef PFM_Q_FMA
    const float bg1 = tbl[TBL_G1 + o];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const float g1 = lrelu(quad_sum(acc[s].x + acc[s].y) + bg1, j.slope);
        if (kq == 0) lds[sv[s].vin2 + FT + o] = g1;
    }
    // fc_global2's rows are split over the waves exactly as these outputs are (wave w: g1[16 w .. 16 w + 15]): wave-local hand-over
    wave_lds_sync();
}
// four jets: two passes of two (the four-jet instantiation of the loop above put 400 bytes per lane into scratch)
template <int NSEG, int NP, int NSV>
__device__ __forceinline__ void fastq_g1(const JetDims& j, float* __restrict__ lds, const SegView (&sv)[NSV], const f32x4 (&gl)[FNG],
                                         const float* __restrict__ tbl) {
    if constexpr (NSEG == 4) {
        fastq_g1_part<2, NP, NSV, 0>(j, lds, sv, gl, tbl);
        __builtin_amdgcn_sched_barrier(0);
        fastq_g1_part<2, NP, NSV, 2>(j, lds, sv, gl, tbl);
    } else {
        fastq_g1_part<NSEG, NP, NSV, 0>(j, lds, sv, gl, tbl);
    }
}
// wave partial of fc_global2 over this wave's 16 rows of g1 -> g2p[w][o2]
template <int NSEG, int NSV>
__device__ __forceinline__ void fastq_g2_partial(float* __restrict__ lds, const SegView (&sv)[NSV], const f32x4& w2) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6, kq = tid & 3, o2 = lane >> 2;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(lds + sv[s].vin2 + FT + 16 * w + 4 * kq);
        const f32x2 acc = pk_dot_step(f32x2{0.f, 0.f}, w2, x);
        const float p = quad_sum(acc.x + acc.y);
        if (kq == 0) lds[sv[s].g2p + MAXL * w + o2] = p;
    }
}
// the eight wave partials of output (lane & 15), added in wave order; all eight reads in flight before the first add
__device__ __forceinline__ float fastq_sum_partials(const float* __restrict__ g2p_o) {
    float part[NW];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) part[ww] = g2p_o[MAXL * ww];
    __builtin_amdgcn_sched_barrier(0);
    float gn = part[0];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) gn += part[ww];
    return gn;
}

// Stem chain on the KQ16 / WQ16 copies (see fast_chain_stem for the contract): gl = the 16 [mean ; sum] panels of q_g1.
template <int NSEG, int NSV, typename After, typename Publish>
__device__ __forceinline__ void fastq_chain_stem(const JetDims& j, float* __restrict__ lds, const SegView (&sv)[NSV], const f32x4 (&gl)[FNG],
                                                 const ChainLoads& L, const float* __restrict__ tbl, After after_fc1, Publish publish_next) {
    const int tid = launder(threadIdx.x), lane = tid & 63;
    fastq_g1<NSEG, FNGS>(j, lds, sv, gl, tbl);
    after_fc1();
    const float bg2 = tbl[TBL_G2 + (lane & 15)];  // before the next chain's rows replace these
    fastq_g2_partial<NSEG>(lds, sv, L.w2);
    __syncthreads();
    publish_next();  // every read of this chain's table rows lies before the barrier above
    if (tid < 16) {
#pragma unroll
        for (int s = 0; s < NSEG; ++s) lds[sv[s].vin + FT + 2 * H + tid] = lrelu(fastq_sum_partials(lds + sv[s].g2p + tid) + bg2, j.slope);
    }
    __syncthreads();
}

// Layer chain on the KQ16 / WQ16 copies (see fast_chain_layer for the contract): gl = the 17 [mean ; sum ; g] panels of q_gl1,
// wbA = this thread's float4 of q_we1 (the g rows of local linear 1's extras).
template <int NSEG, int NSV>
__device__ __forceinline__ void fastq_chain_layer(const JetDims& j, float* __restrict__ lds, const Carve& c, const SegView (&sv)[NSV],
                                                  const f32x4 (&gl)[FNG], const f32x4& wbA, const ChainLoads& L, const float* __restrict__ tbl) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6, o = tid >> 2, kq = tid & 3;
    float gold[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) gold[s] = lds[sv[s].vin + FT + 2 * H + (lane & 15)];  // g_old, before anyone overwrites it
    fastq_g1<NSEG, FNG>(j, lds, sv, gl, tbl);
    fastq_g2_partial<NSEG>(lds, sv, L.w2);
    __syncthreads();
    const float bg2 = tbl[TBL_G2 + (lane & 15)], bl1 = tbl[TBL_L1 + o];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        float gn = fastq_sum_partials(lds + sv[s].g2p + (lane & 15));
        gn += bg2;
        gn += gold[s];  // residual before the activation, epic.py:184-186
        gn = lrelu(gn, j.slope);
        // every wave keeps its own copy of g_new (the input of its slice of the bias GEMV); wave 0's copy is vin.g, the next stage's input.
        // Entries >= L are lrelu(0) = 0 (zero-padded weights and biases).
        float* gcopy = (w == 0) ? lds + sv[s].vin + FT + 2 * H : lds + sv[s].gcopy + MAXL * w;
        if (lane < 16) gcopy[lane] = gn;
        wave_lds_sync();
        const f32x2 acc = pk_dot_step(f32x2{0.f, 0.f}, wbA, *reinterpret_cast<const f32x4*>(gcopy + 4 * kq));
        const float p1 = quad_sum(acc.x + acc.y);
        if (kq == 0) lds[sv[s].bj1 + o] = p1 + bl1;  // wave w: bj1[16 w .. 16 w + 15], the slice its own particle phase reads
    }
    if (L2LDS) fast_chain_publish_l2(L, lds + c.bj2);
    wave_lds_sync();
}

// ---- round 4: the chains of the bf16 flavour ON THE MATRIX PIPE, the workgroup's jets as columns (CH16 blocks, include/pfm_hip.h) ---------
// tests/diag/ab_cfg2.py: the seven chains were 4.3 of the 11.4 ms of the cfg-2 bf16 sampler (four jets per workgroup: every VALU
// instruction of a chain once per jet, its LDS round trips once per step).  Under the reference's bf16-mixed precision every nn.Linear
// runs in bf16 -- the global MLP's too -- so for bf16 descriptors the three per-jet Linears of a chain are v_mfma_f32_16x16x32_bf16
// products  out^T [16 outputs x 16 columns] += W [16 x 32 k] . v [32 k x 16 columns]  with column c = jet c of the workgroup (one, two or
// four real columns; the others repeat them and are never stored): fc_global1 is 9 MFMAs per wave for ALL jets instead of 34 packed FMAs +
// a 16-lane reduction tree per jet, fc_global2 4 MFMAs (every wave computes all 16 outputs: no wave partials), the bias GEMV 1.  A column's
// result does not depend on the others: one, two and four jets per workgroup give the same bits (tests/test_hip_packed.py).
// Operand vectors live in LDS as bf16 (fp32 accumulate; bias, residual, LeakyReLU in fp32):
//   vin  + [0, 144) floats : [mean (128) ; sum (128) ; g (16) ; 0 (16)] bf16, mean / sum written by pool_finish<.., VB16>, g by the chain
//   vin  + FT + 2H         : g as fp32 (the residual of the next chain)
//   vin2 + [0, 64)         : g1 (128) bf16
//   gcopy + 16 w + [0, 8)  : wave w's copy of g_new (16) bf16
constexpr int FNB = 9, FNBS = 8;  // K tiles of fc_global1 ([mean ; sum ; g] = 272 -> 288 rows) / fc_g1 ([mean ; sum] = 256 rows)
struct ChainLoadsB {
    f32x4 w2[4];  // fc_global2 as CH16: unit kt * 64 + lane (the same for every wave)
    f32x4 stg;    // as ChainLoads
};
__device__ __forceinline__ ChainLoadsB fastb_chain_loads(blob_rsrc rs, int64_t b_gl2, const float* __restrict__ slot, int o_g1, int o_l1,
                                                        int o_g2, int o_l2) {
    const int tid = launder(threadIdx.x), lane = tid & 63;
    ChainLoadsB L;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) L.w2[kt] = bload4(rs, b_gl2 + kt * 256, lane * 16);
    L.stg = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 100) {
        const int off = tid < 32 ? o_g1 + 4 * tid : (tid < 64 ? o_l1 + 4 * (tid - 32) : (tid < 68 ? o_g2 + 4 * (tid - 64) : o_l2 + 4 * (tid - 68)));
        L.stg = *reinterpret_cast<const f32x4*>(slot + off);
    }
    return L;
}
// LDS float offsets of the jet this lane's column stands for (column c -> jet c & (NSEG - 1)), and whether the column is a real one
struct ColView {
    int vin, vin2, gcopy, bj1;
    bool real;
};
template <int NSEG, int NSV>
__device__ __forceinline__ ColView fastb_col_view(const SegView (&sv)[NSV]) {
    const int col = launder(threadIdx.x) & 15;
    ColView v{sv[0].vin, sv[0].vin2, sv[0].gcopy, sv[0].bj1, col < NSEG};
    if constexpr (NSEG == 2) {
        if (col & 1) { v.vin = sv[1].vin; v.vin2 = sv[1].vin2; v.gcopy = sv[1].gcopy; v.bj1 = sv[1].bj1; }
    } else if constexpr (NSEG == 4) {
        // the views of jets 1 .. 3 are equally spaced (quad_view): arithmetic, not a select chain over sv[] -- hipcc turns that into a
        // lane-indexed load of the array and the views end up in scratch memory
        const int s = col & 3, st = sv[2].vin - sv[1].vin, k = (s > 0 ? s - 1 : 0) * st;
        if (s > 0) { v.vin = sv[1].vin + k; v.vin2 = sv[1].vin2 + k; v.gcopy = sv[1].gcopy + k; v.bj1 = sv[1].bj1 + k; }
    }
    return v;
}
// g1 = lrelu(W1 . v + t1) -> vin2 (bf16); NK K-tiles from `gl` (CH16 fragments of this wave's 16 outputs)
template <int NK>
__device__ __forceinline__ void fastb_g1(const JetDims& j, float* __restrict__ lds, const ColView& cv, const f32x4 (&gl)[FNB],
                                         const float* __restrict__ tbl) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6, q = lane >> 4;
    const float* vb = lds + cv.vin + 4 * q;  // bf16 element 32 kt + 8 q = float 16 kt + 4 q
    bf16x8 b[NK];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) b[kt] = *reinterpret_cast<const bf16x8*>(vb + 16 * kt);
    f32x4 acc0 = *reinterpret_cast<const f32x4*>(tbl + TBL_G1 + 16 * w + 4 * q), acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: a dependent MFMA waits
#if PFM_QG1_BATCH > 0
    __builtin_amdgcn_sched_barrier(0);  // every operand read requested before the first MFMA (hipcc sinks each ds_read to its use: NK dependent LDS round trips)
#endif
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        if (kt & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gl[kt]), b[kt], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gl[kt]), b[kt], acc0, 0, 0, 0);
    }
    const f32x4 g1 = lrelu4(acc0 + acc1, j.slope);
    if (cv.real) *reinterpret_cast<s16x4*>(lds + cv.vin2 + 8 * w + 2 * q) = pack_bf16(g1);  // bf16 element 16 w + 4 q
}
// W2 . g1 + t2 for this lane's four outputs (4 q ..) of its column's jet; every wave computes all sixteen
__device__ __forceinline__ f32x4 fastb_g2(float* __restrict__ lds, const ColView& cv, const f32x4 (&w2)[4], const f32x4& t2) {
    const int q = (launder(threadIdx.x) & 63) >> 4;
    const float* gb = lds + cv.vin2 + 4 * q;
    bf16x8 b[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) b[kt] = *reinterpret_cast<const bf16x8*>(gb + 16 * kt);
    f32x4 acc0 = t2, acc1 = {0.f, 0.f, 0.f, 0.f};
#if PFM_QG1_BATCH > 0
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2[kt]), b[kt], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2[kt]), b[kt], acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}
// wave 0 publishes g_new: fp32 (next residual) and bf16 (next fc_global1 operand)
__device__ __forceinline__ void fastb_publish_g(float* __restrict__ lds, const ColView& cv, const f32x4& gn) {
    const int tid = launder(threadIdx.x), q = (tid & 63) >> 4;
    if (tid < 64 && cv.real) {
        *reinterpret_cast<f32x4*>(lds + cv.vin + FT + 2 * H + 4 * q) = gn;
        *reinterpret_cast<s16x4*>(lds + cv.vin + 128 + 2 * q) = pack_bf16(gn);  // bf16 element 256 + 4 q
    }
}

// Stem chain on the matrix pipe (contract: fast_chain_stem).  gl = the 8 K-tiles of b_g1.
template <int NSEG, int NSV, typename After, typename Publish>
__device__ __forceinline__ void fastb_chain_stem(const JetDims& j, float* __restrict__ lds, const SegView (&sv)[NSV], const f32x4 (&gl)[FNB],
                                                 const ChainLoadsB& L, const float* __restrict__ tbl, After after_fc1, Publish publish_next) {
    const ColView cv = fastb_col_view<NSEG>(sv);
    const f32x4 t2 = *reinterpret_cast<const f32x4*>(tbl + TBL_G2 + 4 * ((launder(threadIdx.x) & 63) >> 4));  // every read of this chain's
    fastb_g1<FNBS>(j, lds, cv, gl, tbl);                                                                       // table rows: before the barrier
    after_fc1();
    __syncthreads();  // every wave's slice of g1
    publish_next();
    fastb_publish_g(lds, cv, lrelu4(fastb_g2(lds, cv, L.w2, t2), j.slope));
    __syncthreads();
}

// Layer chain on the matrix pipe (contract: fast_chain_layer).  gl = the 9 K-tiles of b_gl1, wbA = this wave's unit of b_we1.
template <int NSEG, int NSV>
__device__ __forceinline__ void fastb_chain_layer(const JetDims& j, float* __restrict__ lds, const Carve& c, const SegView (&sv)[NSV],
                                                  const f32x4 (&gl)[FNB], const f32x4& wbA, const ChainLoadsB& L, const float* __restrict__ tbl) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6, q = lane >> 4;
    const ColView cv = fastb_col_view<NSEG>(sv);
    const f32x4 gold = *reinterpret_cast<const f32x4*>(lds + cv.vin + FT + 2 * H + 4 * q);  // before wave 0 overwrites it (behind the barrier)
    const f32x4 t2 = *reinterpret_cast<const f32x4*>(tbl + TBL_G2 + 4 * q);
    f32x4 bl = *reinterpret_cast<const f32x4*>(tbl + TBL_L1 + 16 * w + 4 * q);
    fastb_g1<FNB>(j, lds, cv, gl, tbl);
    __syncthreads();  // every wave's slice of g1 -- the chain's only barrier
    f32x4 gn = fastb_g2(lds, cv, L.w2, t2);
    gn += gold;  // residual before the activation, epic.py:184-186 (t2 is the accumulator's start value)
    gn = lrelu4(gn, j.slope);
    fastb_publish_g(lds, cv, gn);
    // this wave's copy of g_new as the bias GEMV's operand: k = 8 q .. 8 q + 7 for q < 2, zeros behind row 16
    float* gc = lds + cv.gcopy + MAXL * w;
    if (cv.real) *reinterpret_cast<s16x4*>(gc + 2 * q) = pack_bf16(gn);
    wave_lds_sync();
    f32x4 braw = *reinterpret_cast<const f32x4*>(gc + 4 * (q & 1));
    if (q >= 2) braw = f32x4{0.f, 0.f, 0.f, 0.f};
    bl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wbA), __builtin_bit_cast(bf16x8, braw), bl, 0, 0, 0);
    if (cv.real) *reinterpret_cast<f32x4*>(lds + cv.bj1 + 16 * w + 4 * q) = bl;  // wave w: bj1[16 w ..], the slice its own particle phase reads
    if (L2LDS) fast_chain_publish_l2(L, lds + c.bj2);
    wave_lds_sync();
}

// fc_l3 head with its per-jet bias from the table and its weights already in registers (requested during the last particle
// phase).  emit(p, f, lrelu(b3[f] + W3[f].x[p]) * mask[p]) for the rows p < n_rows ONLY: the sampler's state rows behind a jet's
// last valid particle start as z * mask = 0 and an update by 0 would leave them there.  epic.py:387-391
// AF == 4: the panel's MFMA_A16 copy on the bf16 pipe (four v_mfma_f32_16x16x32_bf16 per tile)
template <int AF, bool RB16 = false, typename Emit>
__device__ __forceinline__ void fast_head(const JetDims& j, float* __restrict__ lds, const Carve& c, int n_rows,
                                          const f32x4 (&a)[AF], f32x4 b3, Emit emit) {
    const int tid = launder(threadIdx.x), lane = tid & 63, w = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    const float* bufB = lds + c.bufB;
    int koff[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) koff[kt] = pl * H + (((4 * kt + q) ^ pl) << 2);
    const int ntiles = (n_rows + TILE - 1) / TILE;
    for (int tile = w; tile < ntiles; tile += NW) {
        const int p = tile * TILE + pl;
        const float* s0 = bufB + tile * TILE * H;
        f32x4 b[8];
        // two accumulator chains over the K halves (a dependent fp32 MFMA waits 40 cycles, the pipe issues one every 32)
        f32x4 acc0 = b3, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (AF == 4 && !RB16) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) b[kt] = *reinterpret_cast<const f32x4*>(s0 + koff[kt]);
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kt2]), pack_bf16x8(b[2 * kt2], b[2 * kt2 + 1]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kt2 + 2]), pack_bf16x8(b[2 * kt2 + 4], b[2 * kt2 + 5]), acc1, 0, 0, 0);
            }
        } else if constexpr (AF == 4) {
            // bf16-resident activations (gemm_phase<.., RB16>): the last local linear 2 left its rows as bf16 in plane B, one 16-byte
            // unit per K-quarter and lane
            const float* pb = lds + c.bufA + (c.bufB - c.bufA) / 2 + (tile * TILE + pl) * (H / 2);
#pragma unroll
            for (int kt2 = 0; kt2 < 4; ++kt2) b[kt2] = *reinterpret_cast<const f32x4*>(pb + (((4 * kt2 + q) ^ pl) << 2));
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kt2]), __builtin_bit_cast(bf16x8, b[kt2]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kt2 + 2]), __builtin_bit_cast(bf16x8, b[kt2 + 2]), acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) b[kt] = *reinterpret_cast<const f32x4*>(s0 + koff[kt]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].x, b[kt].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt + 4].x, b[kt + 4].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].y, b[kt].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt + 4].y, b[kt + 4].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].z, b[kt].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt + 4].z, b[kt + 4].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].w, b[kt].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt + 4].w, b[kt + 4].w, acc1, 0, 0, 0);
            }
        }
        acc0 += acc1;
        if (p < n_rows) {
            const float m = lds[c.maskf + p];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * q + r;
                if (f < j.F) emit(p, f, lrelu(acc0[r], j.slope) * m);
            }
        }
    }
}

// Conditioned jets: the conditioning slots of the per-jet input vector and what the 18th panel reads behind g hold zeros for the whole
// call (the conditioning terms come from the cond table).  Call once behind epic_jet_setup(cond = nullptr); a barrier must follow.
__device__ __forceinline__ void fast_cond_zero(const JetDims& j, float* __restrict__ lds, const Carve& c) {
    const int tid = threadIdx.x;
    if (tid < j.C) lds[c.vin + FT + tid] = 0.f;
    const int tail0 = FT + j.C + 2 * H + MAXL;  // behind g's 16 slots
    if (tid < VIN_FLOATS - tail0) lds[c.vin + tail0 + tid] = 0.f;
}

// What an evaluation carries over from the one before it (requested behind that one's last particle phase / during its head):
// BF16: the A operands are MFMA_A16 fragments (4 float4 registers of bf16 instead of 8 of fp32: half the weight stream of the particle
// Linears and 32 VGPRs fewer -- with fp32 fragments rounded in the kernel the bf16 instantiations spilled 52-84 bytes per lane)
template <bool BF16>
struct FastCarry {
    static constexpr int AF = BF16 ? 4 : 8;
    f32x4 a1[AF], a2[AF];  // a2: fc_l2's weights on entry; a1: scratch (phase-1 weights, then the head's)
    f32x4 sj1;             // this lane's slice of fc_l1's per-jet bias for the coming evaluation
    float aw;              // this lane's element of fc_l1's A operand (constant over the call)
};
template <bool BF16>
__device__ __forceinline__ void load_afrag_lin(f32x4 (&a)[BF16 ? 4 : 8], blob_rsrc rs, const pfm_local_lin& l, int w, int lane) {
    if constexpr (BF16) load_afrag16(a, rs, l.A16, w, lane);
    else load_afrag(a, rs, l.A, w, lane);
}
template <bool BF16>
__device__ __forceinline__ PfSeg seg_afrag_lin(const pfm_local_lin& l, int w, int lane) {
    return BF16 ? seg_afrag16(l.A16, w, lane) : seg_afrag(l.A, w, lane);
}
// ctS (conditioned jets): the stem slot of the jet's cond table, or nullptr
template <bool BF16>
__device__ __forceinline__ void fast_carry_request(FastCarry<BF16>& cy, const pfm_epic_desc& d, blob_rsrc rs, const float* __restrict__ tbS_next,
                                                   const float* __restrict__ ctS = nullptr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    load_afrag_lin<BF16>(cy.a2, rs, d.l2, w, lane);
    cy.sj1 = *reinterpret_cast<const f32x4*>(tbS_next + TB_SJ1 + 4 * (4 * w + (lane >> 4)));
    if (ctS) cy.sj1 += *reinterpret_cast<const f32x4*>(ctS + TB_SJ1 + 4 * (4 * w + (lane >> 4)));
}

// One evaluation: yin -> emit(...).  tbE / tbE_next: the table rows of this and of the next evaluation (the last evaluation
// passes its own again).
// NSEG == 2: the rows [0, n_rows) hold two jets (sg: where the second starts); without conditioning every tabulated bias is the same
// for both, so only the chains (pooled vectors, g, bj1) run per jet.
// COND: ct = the jet's cond table ((layers + 1) slots, stem slot last); see fast_path_ok.
template <bool BF16, int NSEG, bool COND, typename Emit>
__device__ __forceinline__ void fast_eval(const pfm_epic_desc& d, const JetDims& j_in, const float* __restrict__ blob,
                                          float* __restrict__ lds, const Carve& c, int n_rows, const float* __restrict__ tbE,
                                          const float* __restrict__ tbE_next, FastCarry<BF16>& cy, Emit emit, const Segs* sg = nullptr,
                                          const float* __restrict__ ct = nullptr) {
    static_assert(!(COND && NSEG != 1), "conditioned jets: one jet per workgroup");
    static_assert(NSEG != 4 || L2LDS, "quad mode reads the shared biases from c.bj2");
    constexpr bool BCH = BF16 && !COND && PFM_BCHAIN;  // bf16 descriptors: the chains on the matrix pipe (fastb_chain_*)
    constexpr int NGL = BCH ? FNB : FNG, NGLS = BCH ? FNBS : FNGS;
    constexpr int AF = FastCarry<BF16>::AF;
    JetDims j = j_in;
    j.C = 0;  // (see QCH below: the lean path's per-jet vectors have no conditioning slots)
    using CLoads = std::conditional_t<BCH, ChainLoadsB, ChainLoads>;
    // Chains on the KQ16 / WQ16 copies (fastq_chain_*) for every unconditioned instantiation.  Same-box A/B (tests/diag/ab_cfg2.py,
    // ab_time.py; libraries built with -DPFM_QCHAIN_MINSEG=1 / 5): two jets per workgroup (cfg-2 fp32) 28.3 -> 27.0 ms; four jets
    // unchanged in time but without the scratch the KM16 chain's four-jet instantiation needed (44 B per lane bf16, 192 B fp32 -> 0);
    // ONE jet 1.4 % slower at 2 tiles, equal from 4 tiles on, +0.2 % on the bench mix: a single jet's chain is bound by its LDS
    // round trips, not by its instruction count, and the float4 reads move four times the bytes.  All of them take it, because one,
    // two and four jets per workgroup must give the same bits (tests/test_hip_packed.py).
    // Conditioned jets (round 4): the same chains -- the KQ16 / WQ16 copies carry neither time nor conditioning rows, the conditioning
    // terms come from the jet's cond table like the time terms from the time table, and the per-jet vectors are laid out as if C = 0
    // (`j` below: the pooled vectors start at vin + FT).  One panel less in the register window: the conditioned fp32 kernel builds
    // without scratch (round 3: 20 bytes per lane).
    constexpr bool QCH = !BCH && (COND || NSEG >= PFM_QCHAIN_MINSEG);
    static_assert(QCH || BCH || !COND, "the KM16 chains (fast_chain_*) serve diagnostic builds of unconditioned jets only");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NSV = NSEG == 4 ? 4 : 2;
    SegView sv[NSV];
    QuadPhase qp1, qpt;  // quad mode: the four jets as a phase with a per-jet LDS bias (bj1) / with a shared bias (c.bj2) sees them
    if constexpr (NSEG == 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sv[s] = quad_view(c, s);
            qp1.bj[s] = sv[s].bj1; qpt.bj[s] = c.bj2;
            qp1.vin[s] = qpt.vin[s] = sv[s].vin;
            qp1.misc[s] = qpt.misc[s] = sv[s].misc;
        }
    } else {
        sv[0] = seg_view(c, j.N, 0);
        sv[1] = seg_view(c, j.N, NSEG == 2 ? 1 : 0);
    }
    Seg2Phase s2p, s2t;  // the second jet as a phase with an LDS bias (bj1) / with a table bias sees it
    if (NSEG == 2) {
        s2p.bj = lds + sv[1].bj1;
        s2p.mask1 = s2t.mask1 = lds + sv[1].maskf;
        s2p.t1 = s2t.t1 = sg->r1 / TILE;
        s2p.vin1 = s2t.vin1 = sv[1].vin;
        s2p.misc1 = s2t.misc1 = sv[1].misc;
    }
    float* bufA = lds + c.bufA;
    float* bufB = lds + c.bufB;
    // bf16 flavour: the activations the MFMAs read live as BF16 PLANES (gemm_phase<.., RB16>): bufA's N x 128 floats hold two of them --
    // plane A (what local linear 1 / fc_l1 writes: read by the following phase only) and plane B (the bf16 copy of bufB's fp32 rows, the
    // residual stream x_local).  Same LDS bytes as before, half the operand reads, an eighth of the conversions.
    constexpr bool RB16 = BF16 && PFM_RB16;
    float* planeA = bufA;
    float* planeB = bufA + (c.bufB - c.bufA) / 2;
    const float* maskf = lds + c.maskf;
    const float* tbS = tbE + (size_t)j.layers * TB_SLOT;
    const float* ctS = COND ? ct + (size_t)j.layers * TB_SLOT : nullptr;
    const int w2r0 = FT;  // (KM16 chains of diagnostic builds) fc_global2's 128-row window starts behind the time rows
    const blob_rsrc rs = make_blob_rsrc(blob, d.blob_floats + PFM_DESC_FLOATS);
    f32x4 gl[NGL], wbA[1];
    PFM_STAMP(1);
    float* tbl = lds + c.total;  // TBL_FLOATS behind the carve (fast_path_ok: it fits)
    if constexpr (BCH) {  // rows 272 .. 287 of every jet's bf16 chain vector: zero weights there, but 0 x (stale NaN bits) is NaN
#pragma unroll
        for (int s = 0; s < NSEG; ++s)  // (static indices only: a lane-indexed sv[] would put the views into scratch memory)
            if ((tid >> 3) == s) lds[sv[s].vin + 136 + (tid & 7)] = 0.f;
    }
    // Unconditioned jets run their chains on the KQ16 / WQ16 copies of the per-jet blocks (fastq_chain_*, round 4), conditioned ones on
    // the KM16 / KP16 blocks behind the time panels (fast_chain_*): where a chain's register windows and loads come from
    auto gl1_seg = [&](int k) {
        if constexpr (BCH) return PfSeg{d.b_gl1[k], 256, ((w * FNB) * 64 + lane) * 16};
        else if constexpr (!QCH) return seg_panels(d.layer[k].gl1.W, FTP, tid);
        else return seg_panels(d.q_gl1[k], 0, tid);
    };
    auto we1_seg = [&](int k) {
        if constexpr (BCH) return PfSeg{d.b_we1[k], 256, (w * 64 + lane) * 16};
        else if constexpr (!QCH) return seg_panels(d.layer[k].lc1.We, FTP, tid);
        else return seg_panels(d.q_we1[k], 0, tid);
    };
    auto layer_chain_loads = [&](int k, const float* __restrict__ tb) {
        if constexpr (BCH) return fastb_chain_loads(rs, d.b_gl2[k], tb, TB_G1, TB_L1, TB_G2, TB_L2);
        else if constexpr (!QCH) return fast_chain_loads(rs, d.layer[k].gl2.W, w2r0, tb, TB_G1, TB_L1, TB_G2, nullptr, TB_L2);
        else return fastq_chain_loads(rs, d.q_gl2[k], tb, TB_G1, TB_L1, TB_G2, TB_L2, COND ? ct + (size_t)k * TB_SLOT : nullptr);
    };
    // the stem chain's own loads, two phases ahead of their use (conditioned jets: with fc_l2's bias row for c.bj2)
    CLoads L;
    if constexpr (BCH) L = fastb_chain_loads(rs, d.b_g2, tbS, TB_SG1, TB_SG1, TB_SG2, TB_SJ2);
    else if constexpr (!QCH) L = fast_chain_loads(rs, d.g2.W, w2r0, tbS, TB_SG1, TB_SG1, TB_SG2, nullptr, TB_SJ2);
    else L = fastq_chain_loads(rs, d.q_g2, tbS, TB_SG1, TB_SG1, TB_SG2, TB_SJ2, ctS);
#ifndef PFM_AB_NOL1  // (PFM_AB_*: timing-only ablation builds of tests/diag/fixed_cost_table.sh; results are garbage)
    fast_stem_l1<RB16>(j, lds, c, n_rows, cy.aw, cy.sj1);
#endif
    if (COND || L2LDS) {  // fc_l2 reads its per-jet bias (time (+ conditioning) term) from LDS: in place before the barrier in front of it
        fast_chain_publish(L, tbl);
        fast_chain_publish_l2(L, lds + c.bj2);
    }
    __syncthreads();
    PFM_STAMP(3);
    // ---- fc_l2: bufB = lrelu(W.bufA + bj2 + bufA), pooled -> vin   epic.py:364-371; carries the stem chain's fc_g1 rows
    {
        Prefetch<NGLS> pf{rs, gl, nullptr, nullptr, nullptr, BCH ? PfSeg{d.b_g1, 256, ((w * FNBS) * 64 + lane) * 16} : (!QCH ? seg_panels(d.g1.W, FTP, tid) : seg_panels(d.q_g1, 0, tid)), {}, {}, {}};
        const float* bj = (COND || L2LDS) ? lds + c.bj2 : tbS + TB_SJ2;
        s2t.bj = bj;
        if constexpr (RB16)
            gemm_phase<true, true, false, BF16, decltype(pf), NSEG, true, AF, BCH, true>(cy.a2, planeA, planeB, bufB, bj, maskf, j, lds, c, nullptr, nullptr,
                                                                                         n_rows, pf, s2t, &qpt, bufB);
        else
            gemm_phase<true, true, false, BF16, decltype(pf), NSEG, true, AF, BCH>(cy.a2, bufA, bufB, bufA, bj, maskf, j, lds, c, nullptr, nullptr, n_rows, pf,
                                                                                   s2t, &qpt);
    }
    if (!(COND || L2LDS)) fast_chain_publish(L, tbl);  // (the previous evaluation's last chain read tbl many barriers ago)
    __syncthreads();
    PFM_STAMP(4);
    {
        const pfm_epic_layer& l0 = d.layer[0];
        CLoads L0;
        // the first layer's windows, phase-1 weights and chain loads: no particle phase to ride on; they land behind the rest of the stem chain
        auto first_layer_requests = [&]() {
            Prefetch<NGL, 1> pf{rs, gl, wbA, nullptr, nullptr, gl1_seg(0), we1_seg(0), {}, {}};
            pf.template issue_range<0, NGL + 1>();
            load_afrag_lin<BF16>(cy.a1, rs, l0.lc1, w, lane);
            L0 = layer_chain_loads(0, tbE);
        };
#ifdef PFM_AB_NOCHAIN
        first_layer_requests();
        fast_chain_publish(L0, tbl);
        __syncthreads();
#else
        if constexpr (BCH) fastb_chain_stem<NSEG>(j, lds, sv, gl, L, tbl, first_layer_requests, [&]() { fast_chain_publish(L0, tbl); });
        else if constexpr (!QCH) fast_chain_stem<NSEG, false>(j, lds, sv, gl, L, tbl, first_layer_requests, [&]() { fast_chain_publish(L0, tbl); });
        else fastq_chain_stem<NSEG>(j, lds, sv, gl, L, tbl, first_layer_requests, [&]() { fast_chain_publish(L0, tbl); });
#endif
        L = L0;
    }
    f32x4 b3 = {0.f, 0.f, 0.f, 0.f};  // head bias (zero for f >= F): requested in front of the last particle phase
    for (int k = 0; k < j.layers; ++k) {
        const pfm_epic_layer ly = d.layer[k];  // by value: the offset dwords in one batch of scalar loads
        const bool last = k + 1 == j.layers;
        const int kn = last ? k : k + 1;                       // last layer: its own blocks again (harmless, hidden)
        const pfm_epic_layer& nx = d.layer[kn];
        const float* tbK = tbE + (size_t)k * TB_SLOT;
        const float* tbN = tbE + (size_t)(last ? k : k + 1) * TB_SLOT;
        PFM_STAMP(10);
#ifndef PFM_AB_NOCHAIN
        if constexpr (BCH) fastb_chain_layer<NSEG>(j, lds, c, sv, gl, wbA[0], L, tbl);
        else if constexpr (!QCH) fast_chain_layer<NSEG, false>(j, lds, c, sv, gl, wbA[0], L, tbl);
        else fastq_chain_layer<NSEG>(j, lds, c, sv, gl, wbA[0], L, tbl);
#else
        if (COND || L2LDS) fast_chain_publish_l2(L, lds + c.bj2);
#endif
        PFM_STAMP(12);
        // phase 1: bufA = lrelu(W1 . bufB + bj1)   epic.py:194-196.  Riders: phase 2's weights and ALL per-jet windows of the next
        // layer (gl / wbA were consumed by the chain above), so that nothing the next chain waits for is requested late
        {
            Prefetch<AF, NGL, 1> pf{rs, cy.a2, gl, wbA, nullptr, seg_afrag_lin<BF16>(ly.lc2, w, lane), gl1_seg(kn), we1_seg(kn), {}};
            if constexpr (RB16)
                gemm_phase<false, false, false, BF16, decltype(pf), NSEG, true, AF, false, true>(cy.a1, planeB, planeA, nullptr, lds + c.bj1, maskf, j, lds, c,
                                                                                                 nullptr, nullptr, n_rows, pf, s2p, &qp1);
            else
                gemm_phase<false, false, false, BF16, decltype(pf), NSEG>(cy.a1, bufB, bufA, nullptr, lds + c.bj1, maskf, j, lds, c, nullptr,
                                                                           nullptr, n_rows, pf, s2p, &qp1);
        }
#ifndef PFM_AB_NOBAR
        __syncthreads();
#endif
        PFM_STAMP(13);
        // the next chain's loads, a phase ahead
        L = layer_chain_loads(kn, tbN);
        if (last) {
            b3 = *reinterpret_cast<const f32x4*>(tbS + TB_SB3 + 4 * (lane >> 4));
            if (COND) b3 += *reinterpret_cast<const f32x4*>(ctS + TB_SB3 + 4 * (lane >> 4));
        }
        // phase 2: bufB = lrelu(W2 . bufA + bj2 + bufB), pooled -> vin    epic.py:198-200, :160-162
        // (bj2 = bias + time term: constant per evaluation and layer, read from the table; conditioned jets: + the jet's term, from
        // c.bj2).  Riders: the next layer's phase-1 weights, or the head's one 16-row panel behind the last layer, into a1 (free since phase 1)
        {
            const PfSeg sa = last ? PfSeg{BF16 ? d.l3_A16 : d.l3_A, 256, lane * 16} : seg_afrag_lin<BF16>(nx.lc1, w, lane);
            Prefetch<AF> pf{rs, cy.a1, nullptr, nullptr, nullptr, sa, {}, {}, {}};
            const float* bj = (COND || L2LDS) ? lds + c.bj2 : tbK + TB_L2;
            s2t.bj = bj;
            if constexpr (RB16)
                gemm_phase<true, true, false, BF16, decltype(pf), NSEG, false, AF, BCH, true>(cy.a2, planeA, planeB, bufB, bj, maskf, j, lds, c, nullptr, nullptr,
                                                                                              n_rows, pf, s2t, &qpt, bufB);
            else
                gemm_phase<true, true, false, BF16, decltype(pf), NSEG, false, AF, BCH>(cy.a2, bufA, bufB, bufB, bj, maskf, j, lds, c, nullptr, nullptr, n_rows,
                                                                                pf, s2t, &qpt);
            // this layer's chain read tbl two barriers ago; the next one reads it behind the barrier below.  The publish waits for the
            // staged row with a vmcnt that covers every load issued before it: the riders a short jet had no K-quarter for go behind it
            fast_chain_publish(L, tbl);
            pf.issue_tail(NSEG == 4 ? n_rows / (2 * TILE) : phase_full_pairs<BF16>(n_rows));
        }
#ifndef PFM_AB_NOBAR
        __syncthreads();
#endif
    }
    PFM_STAMP(20);
    fast_carry_request(cy, d, rs, tbE_next + (size_t)j.layers * TB_SLOT, ctS);  // lands behind the head
#ifndef PFM_AB_NOHEAD
    fast_head<AF, RB16>(j, lds, c, n_rows, cy.a1, b3, emit);
#endif
}

}  // namespace pfm
