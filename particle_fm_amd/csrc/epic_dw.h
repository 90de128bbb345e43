// Weight gradients of the jet-resident EPiC network: the sums over jets (included by epic_train.hip).
//
//   epic_dw_kernel          dW_b[o][k] = sum over ALL jets' valid rows of da_b[row][o] * act_b[row][k] for the 2*layers + 1
//                           128x128 Linears (fc_l2, fc_local1/2 of every layer): ONE launch, grid (nsplit, nblk).  The rows of a
//                           block are the 16-row pieces (jet, tile) of every jet in jet order; a workgroup takes a contiguous
//                           range of pieces, stages 64 rows of both operands in LDS per step and runs the 128 x 128 x 64 product on
//                           v_mfma_f32_16x16x4_f32 (4 waves, 64 x 64 outputs each, ONE ds_read_b128 per operand feeds 16 MFMAs).
//                           Partial tiles go to scratch.
//   epic_bwd_reduce_kernel  (a) sums the partial tiles in split order into the gradient blob (GRAD_D order of pfm_hip.h),
//                           (b) forms the rank-1 sums over jets  dW[k][o] = sum_jets x_jet[k] * dy_jet[o]  (+ bias = sum dy)  of
//                           the per-jet Linears from the records of the chain kernel, (c) sums the per-jet partials of the two
//                           F-wide particle blocks (fc_l1, fc_l3).  Fixed summation order everywhere: the gradient is a pure
//                           function of its inputs, bit for bit (no atomics).
// Autograd of particle_fm/models/components/epic.py:85-203, 304-391 (the nn.Linear weight gradients), restated.
#pragma once
#include "epic_bwd.h"

namespace pfm {

constexpr int DW_T = 256;    // threads of epic_dw_kernel: 4 waves, wave (wo, wk) owns outputs [64 wo, +64) x inputs [64 wk, +64)
constexpr int DW_S = 128;    // LDS row stride (floats).  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...:
                             // MI355X_MICROARCH.md, LDS): a group takes lanes of two q (two consecutive rows), columns [0,16) + [48,64) of one and
                             // [16,48) of the other -- disjoint banks exactly when the rows are a multiple of 64 floats apart.  The 132 of rounds 2-3
                             // (padding for contiguous groups) shifted the second row by one slot: 2-way conflicts, SQ_LDS_BANK_CONFLICT 32 % of busy
constexpr int DW_MAXB = 8192;  // jets per call (the piece scan lives in LDS)

// saved-activation offset (inside a jet's record) of the input of block b
__device__ __forceinline__ int dw_act_off(const SavedLayout& sl, int b) {
    if (b == 0) return sl.x1;                                   // fc_l2: input x1
    const int k = (b - 1) >> 1;
    if (b & 1) return sl.l1 + k * sl.lstride;                   // fc_local2 of layer k: input l1_k
    return k > 0 ? sl.xo + (k - 1) * sl.lstride : sl.x2;        // fc_local1 of layer k: input h_k
}

__global__ __launch_bounds__(DW_T, 2) void epic_dw_kernel(const float* __restrict__ blob, int64_t desc_off,
                                                          const float* __restrict__ saved, float* __restrict__ work, BwdWork bw,
                                                          int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const zt = lds;               // [64][DW_S]  da rows
    float* const at = lds + 64 * DW_S;   // [64][DW_S]  activation rows
    int* const toff = reinterpret_cast<int*>(lds + 128 * DW_S);  // [B + 1] exclusive scan of the jets' tile counts
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const int N = d.n_points, layers = d.layers;
    const SavedLayout sl = make_saved(N, d.features, layers);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pl = lane & 15, q = lane >> 4;
    const int wo = w >> 1, wk = w & 1;
    const int sc4 = tid & 31, sr = tid >> 5;
    const int split = blockIdx.x, b = blockIdx.y;
    const int* nrows = reinterpret_cast<const int*>(work + bw.nrows);
    // ---- exclusive scan of ceil(nrows / 16) over the jets (every workgroup does its own: B ints, a few hundred cycles) ----
    {
        const int per = (B + DW_T - 1) / DW_T;
        const int j0 = tid * per, j1 = min(B, j0 + per);
        int s = 0;
        for (int jj = j0; jj < j1; ++jj) s += (nrows[jj] + 15) >> 4;
        int* part = reinterpret_cast<int*>(lds);  // zt is not in use yet
        part[tid] = s;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int i = 0; i < DW_T; ++i) { const int v = part[i]; part[i] = run; run += v; }
            toff[B] = run;
        }
        __syncthreads();
        int run = part[tid];
        for (int jj = j0; jj < j1; ++jj) { toff[jj] = run; run += (nrows[jj] + 15) >> 4; }
        __syncthreads();
    }
    const int P = toff[B];
    const int p0 = (int)((int64_t)P * split / bw.nsplit), p1 = (int)((int64_t)P * (split + 1) / bw.nsplit);
    f32x4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread stages rows r = sr + 8 i (i = 0..7) of every 64-row step: piece r >> 4 = i >> 1, row r & 15 of that piece
    // jet of piece p0: last jet with toff[jet] <= p0
    int jet = 0;
    {
        int lo = 0, hi = B - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (toff[mid] <= p0) lo = mid; else hi = mid - 1;
        }
        jet = lo;
    }
    const float* const da = work + bw.da + ((int64_t)b * N) * H + 4 * sc4;  // + jet * nblk * N * H
    const int64_t da_jet = (int64_t)bw.nblk * N * H;
    const float* const act = saved + dw_act_off(sl, b) + 4 * sc4;           // + jet * sl.total
    f32x4 zs[8], as[8];
    unsigned okbits = 0;  // bit i: row i of this thread's 8 is a valid row (applied when the registers go to LDS, so that
                          // nothing consumes a load before the MFMA block it is meant to fly behind)
    // global loads of one 64-row step into registers; `jet` walks along (jet of the step's first piece on entry, of the next
    // step's first piece on exit)
    auto load_step = [&](int pc) {
        int jj = jet;
        okbits = 0;
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int piece = pc + pi;
            const bool live = piece < p1;
            if (live)
                while (toff[jj + 1] <= piece) ++jj;
            const int tile = piece - toff[jj];
            const int nr = live ? nrows[jj] : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * pi + h;
                const int p = tile * 16 + ((sr + 8 * i) & 15);
                const bool ok = live && p < nr;
                const int pcl = ok ? p : 0;
                const int64_t jo = ok ? jj : 0;
                zs[i] = *reinterpret_cast<const f32x4*>(da + jo * da_jet + (int64_t)pcl * H);
                as[i] = *reinterpret_cast<const f32x4*>(act + jo * (int64_t)sl.total + (int64_t)pcl * H);
                okbits |= (ok ? 1u : 0u) << i;
            }
        }
        const int nextp = pc + 4;
        if (nextp < p1)
            while (toff[jet + 1] <= nextp) ++jet;
    };
    if (p0 < p1) load_step(p0);
#pragma unroll 1
    for (int pc = p0; pc < p1; pc += 4) {
        __syncthreads();  // the previous step's tiles have been consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // rows behind a jet's last valid particle / behind this split's range contribute nothing (their saved activations
            // were never written: select, do not multiply)
            const bool ok = (okbits >> i) & 1u;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(zt + (sr + 8 * i) * DW_S + 4 * sc4) = ok ? zs[i] : zero;
            *reinterpret_cast<f32x4*>(at + (sr + 8 * i) * DW_S + 4 * sc4) = ok ? as[i] : zero;
        }
        __syncthreads();
        if (pc + 4 < p1) load_step(pc + 4);  // the next step's rows fly while this step's 256 MFMAs per wave issue
        // operands of k-step ks + 1 requested in front of the 16 MFMAs of ks (a fence pins that: hipcc sinks each ds_read_b128 down to
        // its first use, i.e. an LDS round trip in front of every MFMA block -- 12 of the 16 k-steps waited like that in round 3)
        f32x4 dz = *reinterpret_cast<const f32x4*>(zt + q * DW_S + 64 * wo + 4 * pl);
        f32x4 an = *reinterpret_cast<const f32x4*>(at + q * DW_S + 64 * wk + 4 * pl);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            f32x4 dzn = dz, ann = an;
            if (ks + 1 < 16) {
                dzn = *reinterpret_cast<const f32x4*>(zt + (4 * (ks + 1) + q) * DW_S + 64 * wo + 4 * pl);
                ann = *reinterpret_cast<const f32x4*>(at + (4 * (ks + 1) + q) * DW_S + 64 * wk + 4 * pl);
            }
            __builtin_amdgcn_sched_barrier(0);
#define PFM_DW_ROW(c, zc)                                                                   \
    acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.x, acc[c][0], 0, 0, 0);         \
    acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.y, acc[c][1], 0, 0, 0);         \
    acc[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.z, acc[c][2], 0, 0, 0);         \
    acc[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, an.w, acc[c][3], 0, 0, 0);
            PFM_DW_ROW(0, dz.x) PFM_DW_ROW(1, dz.y) PFM_DW_ROW(2, dz.z) PFM_DW_ROW(3, dz.w)
#undef PFM_DW_ROW
            __builtin_amdgcn_sched_barrier(0);
            dz = dzn;
            an = ann;
        }
    }
    // partial tile -> scratch in accumulator order: float4 (e = 0..3) at (((w*4 + c)*4 + r)*64 + lane) holds
    //   dW[o][k],  o = 64 wo + 4 (4 (lane>>4) + r) + c,  k = 64 wk + 4 (lane&15) + e
    // (A operand = da: MFMA row index i = lane&15 of operand <-> output 64 wo + 4 pl + c; B operand = act: column j = lane&15 <->
    //  input 64 wk + 4 pl + e; the accumulator register r of lane holds D[i = 4 (lane>>4) + r][j = lane&15])
    float* pp = work + bw.part + ((int64_t)b * bw.nsplit + split) * (H * H);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v = {acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]};
            *reinterpret_cast<f32x4*>(pp + ((((w * 4 + c) * 4 + r) * 64 + lane) << 2)) = v;
        }
}

// ---- reduce -------------------------------------------------------------------------------------------------------------
// One job of the rank-1 part: out[k][o] = sum_jets x[k] * dy[o] for k < K, bias[o] = sum_jets dy[o] (o < OUT).
struct R1Job {
    int x0, x0n;     // x = rec[x0 .. x0 + x0n) followed by rec[x1 ..]   (the extras vector [temb ; cond_l] ++ g lives in two places)
    int x1;
    int dy;          // rec offset of dy
    int K, OUT;      // OUT = 128 (KM16 destination) or <= 16 (KP16: [K16][16]; fmt 2: plain K-major [K][OUT])
    int fmt;         // 0 KM16, 1 KP16, 2 KMAJOR [K][OUT]
    int64_t W, b;    // gradient-blob offsets
};

// the rank-1 jobs of stage s (0 = stem, 1 + k = layer k), i = 0..3; stage == layers + 1: the head (i = 0)
__device__ __forceinline__ bool r1_job(const pfm_epic_desc& d, int stage, int i, R1Job& jb) {
    const int T = d.t_dim, C = d.cond_global, Cl = d.cond_local, L = d.latent, TC = T + C, Ke = T + Cl;
    if (stage > d.layers) {  // head: dWe3[k][f] = sum e[k] db3[f]; e = [temb ; cond_l] = prefix of the stem's vin
        if (i != 0) return false;
        const BwdRec br = make_bwd_rec(d.layers);
        jb = R1Job{BwdRec::VIN, Ke, 0, br.db3, Ke, d.features, 2, d.l3_We, d.l3_b};
        return true;
    }
    const int s0 = stage * BwdRec::STAGE;
    const bool stem = stage == 0;
    const pfm_epic_layer& ly = d.layer[stem ? 0 : stage - 1];
    switch (i) {
        case 0:  // fc_g1 / fc_global1: x = vin
            jb = R1Job{s0 + BwdRec::VIN, TC + 2 * H + (stem ? 0 : L), 0, s0 + BwdRec::DAG1, TC + 2 * H + (stem ? 0 : L), H, 0,
                       stem ? d.g1.W : ly.gl1.W, stem ? d.g1.b : ly.gl1.b};
            return true;
        case 1:  // fc_g2 / fc_global2: x = vin2 = [temb ; cond ; g1]
            jb = R1Job{s0 + BwdRec::VIN2, TC + H, 0, s0 + BwdRec::DAG2, TC + H, L, 1, stem ? d.g2.W : ly.gl2.W, stem ? d.g2.b : ly.gl2.b};
            return true;
        case 2:  // fc_l1 extras (stem) / fc_local1 extras: x = [temb ; cond_l] ++ g_out
            jb = R1Job{s0 + BwdRec::VIN, Ke, s0 + BwdRec::GOUT, s0 + BwdRec::DBJ1, Ke + (stem ? 0 : L), H, 0, stem ? d.l1_We : ly.lc1.We,
                       stem ? d.l1_b : ly.lc1.b};
            return true;
        default:  // fc_l2 extras (stem) / fc_local2 extras: x = [temb ; cond_l]
            jb = R1Job{s0 + BwdRec::VIN, Ke, 0, s0 + BwdRec::DBJ2, Ke, H, 0, stem ? d.l2.We : ly.lc2.We, stem ? d.l2.b : ly.lc2.b};
            return true;
    }
}

__device__ __forceinline__ float r1_x(const float* __restrict__ rj, const R1Job& jb, int k) {
    return k < jb.x0n ? rj[jb.x0 + k] : rj[jb.x1 + (k - jb.x0n)];
}

constexpr int RED_T = 256;
// grid.x enumerates work items of three kinds (host computes the counts, all functions of the descriptor):
//   [0, n_tile)                partial-tile sums: item = (block b, 1024-float slice of the 16384-float tile)
//   [n_tile, n_tile + n_r1)    rank-1 panels: item = (stage, job, 16-row panel)
//   rest                       the F-wide particle blocks (dW3, dWx): item = 256 floats of the 2 x MAXF*H sums
struct RedArgs {
    int n_tile, n_r1, n_small, panels_per_job;  // panels_per_job: max 16-row panels of any job (ceil(VIN_FLOATS / 16))
    int item0;                                  // first work item of this launch (a launch may cover parts (b) + (c) or part (a) alone)
};

// RED_G groups of RED_T threads per workgroup: the rank-1 sums over jets (part (b)) are cut into RED_G contiguous jet ranges, one per
// group -- a chain of dependent L2 round trips a quarter as long -- and joined through LDS in group order; parts (a) and (c) run on
// the first group alone (the other waves end at once: ended waves do not take part in a barrier).
constexpr int RED_G = 4;
__global__ __launch_bounds__(RED_T * RED_G) void epic_bwd_reduce_kernel(const float* __restrict__ blob, int64_t desc_off,
                                                                        const float* __restrict__ work, BwdWork bw, int B, RedArgs ra,
                                                                        float* __restrict__ gblob) {
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const int grp = threadIdx.x / RED_T, tid = threadIdx.x % RED_T;
    int item = blockIdx.x + ra.item0;
    if (item < ra.n_tile) {
        if (grp != 0) return;
        // ---- (a) dW tile b = sum over splits, to GRAD_D order ----
        const int b = item >> 4, slice = item & 15;
        const int p4 = slice * 1024 + tid * 4;  // float index inside the tile (accumulator order), 4 consecutive = e 0..3
        const float* pp = work + bw.part + (int64_t)b * bw.nsplit * (H * H) + p4;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        int i = 0;
        for (; i + 4 <= bw.nsplit; i += 4) {
            s0 += *reinterpret_cast<const f32x4*>(pp + (int64_t)i * (H * H));
            s1 += *reinterpret_cast<const f32x4*>(pp + (int64_t)(i + 1) * (H * H));
            s2 += *reinterpret_cast<const f32x4*>(pp + (int64_t)(i + 2) * (H * H));
            s3 += *reinterpret_cast<const f32x4*>(pp + (int64_t)(i + 3) * (H * H));
        }
        for (; i < bw.nsplit; ++i) s0 += *reinterpret_cast<const f32x4*>(pp + (int64_t)i * (H * H));
        const f32x4 s = (s0 + s1) + (s2 + s3);
        const int idx = p4 >> 2;  // ((w*4 + c)*4 + r)*64 + lane
        const int lane = idx & 63, r = (idx >> 6) & 3, c = (idx >> 8) & 3, w = idx >> 10;
        const int o = 64 * (w >> 1) + 4 * (4 * (lane >> 4) + r) + c;
        const int k0 = 64 * (w & 1) + 4 * (lane & 15);
        int64_t gA;
        if (b == 0) gA = d.l2.A;
        else gA = (b & 1) ? d.layer[(b - 1) >> 1].lc2.A : d.layer[(b - 1) >> 1].lc1.A;
        // GRAD_D: float ((w'*8 + it)*4 + r')*64 + lane' holds dW[16 w' + 4 (lane'>>4) + r'][8 (lane'&15) + it]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            const int wq = o >> 4, rq = o & 3, lq = (((o >> 2) & 3) << 4) | (k >> 3), it = k & 7;
            gblob[gA + ((wq * 8 + it) * 4 + rq) * 64 + lq] = s[e];
        }
        return;
    }
    item -= ra.n_tile;
    const BwdRec br = make_bwd_rec(d.layers);
    const float* rec = work + bw.rec;
    if (item < ra.n_r1) {
        // ---- (b) one 16-row panel of a rank-1 job: thread (kk = tid >> 5, o4 = tid & 31) -> rows 16 panel + kk, + kk + 8; 4 outputs ----
        // (every exit up to the join is uniform over the workgroup; per-thread conditions only mask the work)
        __shared__ f32x4 comb[(RED_G - 1) * RED_T * 2];  // partial sums of groups 1 .. RED_G - 1 (two float4 per thread)
        const int panel = item % ra.panels_per_job, ji = item / ra.panels_per_job;
        R1Job jb;
        if (!r1_job(d, ji >> 2, ji & 3, jb)) return;
        const int K16 = (jb.K + 15) & ~15;
        if (16 * panel >= K16 + 16) return;  // one extra "panel" carries the bias row
        const bool bias_panel = 16 * panel >= K16;
        const int kk = tid >> 5, o4 = tid & 31;
        const bool idle = 4 * o4 >= ((jb.OUT + 3) & ~3) || (bias_panel && kk != 0);
        const int ka = 16 * panel + kk, kb = ka + 8;
        const bool va = !bias_panel && ka < jb.K, vb = !bias_panel && kb < jb.K;
        // record offsets of this thread's two x entries (rows past K: any valid float, multiplied by 0)
        const int xa = va ? (ka < jb.x0n ? jb.x0 + ka : jb.x1 + (ka - jb.x0n)) : jb.x0;
        const int xb = vb ? (kb < jb.x0n ? jb.x0 + kb : jb.x1 + (kb - jb.x0n)) : jb.x0;
        const float ma = va ? 1.f : 0.f, mb = vb ? 1.f : 0.f;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
        // this group's jets in order, 8 per step: all 24 loads of a step are issued before its FMAs (the loop is latency-bound
        // otherwise); even jets accumulate in (a0, b0), odd ones in (a1, b1): a fixed order
        constexpr int U = 8;
        const int64_t rt = br.total;
        const float* rbase = rec + jb.dy + 4 * (idle ? 0 : o4);
        int jet = (int)((int64_t)B * grp / RED_G);
        const int jend = idle ? jet : (int)((int64_t)B * (grp + 1) / RED_G);
        for (; jet + U <= jend; jet += U) {
            f32x4 dy[U];
            float xav[U], xbv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* r = rec + (int64_t)(jet + u) * rt;
                dy[u] = *reinterpret_cast<const f32x4*>(rbase + (int64_t)(jet + u) * rt);
                xav[u] = r[xa];
                xbv[u] = r[xb];
            }
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                if (bias_panel) { a0 += dy[u]; a1 += dy[u + 1]; continue; }
                a0 += dy[u] * (xav[u] * ma);
                b0 += dy[u] * (xbv[u] * mb);
                a1 += dy[u + 1] * (xav[u + 1] * ma);
                b1 += dy[u + 1] * (xbv[u + 1] * mb);
            }
        }
        for (; jet < jend; ++jet) {
            const float* r = rec + (int64_t)jet * rt;
            const f32x4 dy0 = *reinterpret_cast<const f32x4*>(rbase + (int64_t)jet * rt);
            if (bias_panel) { a0 += dy0; continue; }
            a0 += dy0 * (r[xa] * ma);
            b0 += dy0 * (r[xb] * mb);
        }
        if (grp > 0) {
            comb[((grp - 1) * RED_T + tid) * 2 + 0] = a0 + a1;
            comb[((grp - 1) * RED_T + tid) * 2 + 1] = b0 + b1;
        }
        __syncthreads();
        if (grp != 0 || idle) return;
        static_assert(RED_G == 4, "four partials are joined below");
        const f32x4 sa = ((a0 + a1) + comb[(0 * RED_T + tid) * 2]) + (comb[(1 * RED_T + tid) * 2] + comb[(2 * RED_T + tid) * 2]);
        const f32x4 sb = ((b0 + b1) + comb[(0 * RED_T + tid) * 2 + 1]) + (comb[(1 * RED_T + tid) * 2 + 1] + comb[(2 * RED_T + tid) * 2 + 1]);
        if (bias_panel) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * o4 + e < jb.OUT) gblob[jb.b + 4 * o4 + e] = sa[e];
            return;
        }
        auto put = [&](int k, f32x4 v) {
            if (k >= jb.K) return;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = 4 * o4 + e;
                if (o >= jb.OUT) continue;
                int64_t pos;
                if (jb.fmt == 0) pos = km16(k, o);
                else if (jb.fmt == 1) pos = (int64_t)k * 16 + o;
                else pos = (int64_t)k * jb.OUT + o;
                gblob[jb.W + pos] = v[e];
            }
        };
        put(ka, sa);
        put(kb, sb);
        return;
    }
    item -= ra.n_r1;
    if (item < ra.n_small) {
        // ---- (c) dW3 (row-major [F][H]) and dWx (K-major [F][H]): sums of the per-jet partials.  A workgroup takes 64 outputs; wave g
        //      sums the jets of its quarter [B g / 4, B (g + 1) / 4) with 8 loads in flight (the loop is a chain of dependent L2 round
        //      trips: one thread walking all B jets took 130 us at 1024 jets), the four partials meet in LDS in wave order ----
        if (grp != 0) return;
        __shared__ float combc[3 * 64];
        const int g = tid >> 6, l = tid & 63;
        const int e = item * 64 + l;  // 0 .. 2 * MAXF * H
        const int which = e / (MAXF * H), fe = e - which * (MAXF * H);
        const bool live = which <= 1 && fe < d.features * H;
        const int off = (which == 0 ? br.dW3 : br.dWx) + (live ? fe : 0);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int jet = (int)((int64_t)B * g / 4);
        const int jend = live ? (int)((int64_t)B * (g + 1) / 4) : jet;
        for (; jet + 8 <= jend; jet += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += rec[(int64_t)(jet + u) * br.total + off];
        }
        for (; jet < jend; ++jet) s[0] += rec[(int64_t)jet * br.total + off];
        const float part = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        if (g > 0) combc[(g - 1) * 64 + l] = part;
        __syncthreads();
        if (g == 0 && live) gblob[(which == 0 ? d.l3_W : d.l1x.W) + fe] = (part + combc[l]) + (combc[64 + l] + combc[128 + l]);
    }
}

}  // namespace pfm
