// The backward kernel body (included by epic_train.hip).  See epic_bwd.h for the on-chip mapping.
#pragma once
#include "epic_bwd.h"

namespace pfm {

// vin = [temb ; cond ; mean ; sum*s ; g_in]; temb / cond are already in place
__device__ __forceinline__ void build_vin(const JetDims& j, float* __restrict__ lds, const BCarve& c,
                                          const float* __restrict__ pool_raw, const float* __restrict__ g_in,
                                          bool has_g) {
    const int tid = threadIdx.x, TC = j.T + j.C;
    const float nvalid = lds[c.misc];
    if (tid < H) {
        const float s = pool_raw[tid];
        lds[c.vin + TC + tid] = s / nvalid;
        lds[c.vin + TC + H + tid] = s * j.sscale;
    } else if (has_g && tid < H + j.L) {
        lds[c.vin + TC + 2 * H + (tid - H)] = g_in[tid - H];
    }
}

// copy n floats (n % 4 == 0, 16-byte aligned both sides) from LDS to the jet's record: threads [t0, t0 + n/4)
__device__ __forceinline__ void rec_put(float* __restrict__ dst, const float* __restrict__ src, int n, int t0) {
    const int i = (int)threadIdx.x - t0;
    if (i >= 0 && 4 * i < n) *reinterpret_cast<f32x4*>(dst + 4 * i) = *reinterpret_cast<const f32x4*>(src + 4 * i);
}

// The per-jet chain of d(loss)/d(blob), loss = sum(loss_parts) / sum(mask_count), times *grad_scale: the gradient w.r.t. the
// hidden state walks back through the layers inside LDS; what has to be summed over jets leaves the CU as plain stores --
// the gradient rows `da` of every 128x128 Linear (epic_dw_kernel forms dW from them), the rank-1 operands and the small
// per-jet partial sums in `rec` (epic_bwd_reduce_kernel).  No atomics.
template <bool BF16>
__global__ __launch_bounds__(NT, 2) void epic_fm_loss_backward_kernel(
    const float* __restrict__ blob, int64_t desc_off, const float* __restrict__ cond,
    const float* __restrict__ mask, const float* __restrict__ saved, const float* __restrict__ inv_mask_total,
    const float* __restrict__ grad_scale, float* __restrict__ work, BwdWork bw, int crit, const float* __restrict__ jet_w,
    float* __restrict__ dtemb,        // dtemb (or NULL): [B][T] gradient w.r.t. a caller-supplied time embedding
    const int* __restrict__ order,    // order (or NULL): launch order of the jets, longest first (pfm_epic_jet_order)
    float* __restrict__ dy = nullptr) {  // dy (or NULL): [B][N][F] gradient w.r.t. the network's particle input (chained flows)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const BCarve c = make_bcarve(j.N, j.F);
    const SavedLayout sl = make_saved(j.N, j.F, j.layers);
    const int jet = order ? order[blockIdx.x] : blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pl = lane & 15, q = lane >> 4, oslot = 4 * w + q;
    const float* sv = saved + (size_t)jet * sl.total;
    const BwdRec br = make_bwd_rec(j.layers);
    float* rec = work + bw.rec + (size_t)jet * br.total;
    float* daj = work + bw.da + (size_t)jet * bw.nblk * j.N * H;  // this jet's gradient rows, [nblk][N][H]
    const float slope = j.slope;
    const int Ke = j.T + j.Cl;
    float* G = lds + c.G;
    float* Hb = lds + c.Hb;
    PFM_BSTAMP(0);
    // ---- setup: mask, n_valid, n_rows, temb, cond, head weights ----
    {
        const float* mj = mask ? mask + (size_t)jet * j.N : nullptr;
        int last = -1;
        float cnt = 0.f;
        for (int p = tid; p < j.N; p += NT) {
            const float m = mj ? mj[p] : 1.0f;
            lds[c.maskf + p] = m;
            cnt += m;
            if (m != 0.f) last = p;
        }
        for (int i = tid; i < j.F * H; i += NT) lds[c.w3 + i] = blob[d.l3_W + i];
        if (tid < j.T) lds[c.vin + tid] = sv[sl.temb + tid];
        if (tid >= 64 && tid < 64 + j.C) lds[c.vin + j.T + (tid - 64)] = cond[(size_t)jet * j.C + (tid - 64)];
        if (tid >= 128 && tid < 128 + MAXL) lds[c.dg + (tid - 128)] = 0.f;
        if (tid >= 192 && tid < 192 + MAXT) lds[c.dte + (tid - 192)] = 0.f;
        for (int m = 32; m >= 1; m >>= 1) {
            cnt += __shfl_xor(cnt, m);
            last = max(last, __shfl_xor(last, m));
        }
        float* red = lds + c.dg1;  // scratch
        if (lane == 0) { red[w] = cnt; red[8 + w] = (float)last; }
        __syncthreads();
        if (tid == 0) {
            float s = 0.f, l = -1.f;
            for (int i = 0; i < NW; ++i) { s += red[i]; l = fmaxf(l, red[8 + i]); }
            lds[c.misc] = s;
            lds[c.misc + 1] = l;
        }
        __syncthreads();
    }
    int n_rows = j.N;
    // a jet without any valid particle is NaN in the reference (epic.py:370 divides by 0): compute all rows then too
    if ((d.flags & PFM_F_SKIP_MASKED_TAIL) && lds[c.misc + 1] >= 0.f) n_rows = (int)lds[c.misc + 1] + 1;
    const int ntiles = (n_rows + TILE - 1) / TILE;
    if (tid == 0) reinterpret_cast<int*>(work + bw.nrows)[jet] = n_rows;
    // d/dv of  w_jet * sum crit(v - u) / M:  mse (crit 0) 2 (v - u);  huber (crit 1, delta 1) clamp(v - u, -1, 1)
    const float gscale = (crit ? 1.0f : 2.0f) * inv_mask_total[0] * grad_scale[0] * (jet_w ? jet_w[jet] : 1.0f);
    const float* maskf = lds + c.maskf;
    const float* evec = lds + c.vin;  // [temb ; cond_l] is a prefix of vin (Cl in {0, C})
    PFM_BSTAMP(1);

    // the first dX product's weights (step (3) of the last layer, or the stem's): in flight across the head
    f32x4 a2[8];
    const blob_rsrc rs = make_blob_rsrc(blob, d.blob_floats + PFM_DESC_FLOATS);
    load_afrag(a2, rs, j.layers > 0 ? d.layer[j.layers - 1].lc2.AT : d.l2.AT, w, lane);
#ifndef PFM_AB_NOHEAD  // (timing-only ablation: head skipped, G left as it is)
    // ---- head backward (epic.py:387-391): da3 = dv * mask * phi'(v);  G = W3^T da3 ----
    for (int i = tid; i < j.N * j.F; i += NT) {
        const int p = i / j.F;
        float val = 0.f;
        if (p < n_rows) {
            const float v = sv[sl.v + i], u = sv[sl.u + i];
            const float dl = crit ? fminf(fmaxf(v - u, -1.0f), 1.0f) : v - u;
            val = gscale * dl * maskf[p] * dlrelu(v, slope);
        }
        lds[c.da3 + i] = val;
    }
    __syncthreads();
    {
        const float* hL = sv + (j.layers > 0 ? sl.xo + (j.layers - 1) * sl.lstride : sl.x2);
        // this jet's part of dW3[f][k] = sum_p da3[p][f] * hL[p][k]: k = tid & 127, particles split 4 ways, the 4 partial sums
        // joined through LDS (Hb is free here) in a fixed order -> rec.dW3
        for (int f0 = 0; f0 < j.F; f0 += 4) {
            const int k = tid & (H - 1), pt = tid >> 7;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            // Eight rows per step, their saved activations (global) AND their da3 entries (LDS) requested before the first FMA; rows
            // behind n_rows and features behind F are selected to zero instead of branched around (round 4: the guarded form was a
            // branch + a dependent ds_read_b32 round trip per term, ~24 in a row per step).  fma(0, h, a) = a: same sums, same order.
            for (int p = pt; p < n_rows; p += 32) {
                float hv[8], dv[8][4];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = p + 4 * u;
                    ok[u] = row < n_rows;
                    const int rc = ok[u] ? row : p;
                    hv[u] = hL[rc * H + k];
#pragma unroll
                    for (int jf = 0; jf < 4; ++jf) dv[u][jf] = lds[c.da3 + rc * j.F + f0 + jf];  // (reads past a row's F entries stay inside the carve)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int jf = 0; jf < 4; ++jf) a[jf] = fmaf((ok[u] && f0 + jf < j.F) ? dv[u][jf] : 0.f, hv[u], a[jf]);
            }
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) Hb[(jf * 4 + pt) * H + k] = a[jf];
            __syncthreads();
            {
                const int jf = tid >> 7;  // 4 features x 128 k = 512 threads
                if (f0 + jf < j.F) {
                    const float* q4 = Hb + (jf * 4) * H + k;
                    rec[br.dW3 + (f0 + jf) * H + k] = (q4[0] + q4[H]) + (q4[2 * H] + q4[3 * H]);
                }
            }
            __syncthreads();
        }
        // db3j[f] = sum_p da3[p][f]   (wave f); dWe3 = sum_jets e (x) db3j: epic_bwd_reduce_kernel
        for (int f = w; f < MAXF; f += NW) {
            float a = 0.f;
            if (f < j.F)
                for (int p = lane; p < n_rows; p += 64) a += lds[c.da3 + p * j.F + f];
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
            if (lane == 0) {
                rec[br.db3 + f] = a;
                lds[c.dag2 + f] = a;  // (scratch until the first global_backward) for the time rows of fc_l3's extras below
            }
        }
        // G[p][4slot..] = sum_f W3[f][4slot..] * da3[p][f]
        const int slot = tid & 31;
        for (int p = tid >> 5; p < n_rows; p += NT / 32) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int f0 = 0; f0 < j.F; f0 += 4) {  // four features per step, all eight reads up front, features >= F selected to zero
                f32x4 wr[4];
                float dq[4];
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) {
                    wr[jf] = *reinterpret_cast<const f32x4*>(lds + c.w3 + min(f0 + jf, j.F - 1) * H + 4 * slot);
                    dq[jf] = lds[c.da3 + p * j.F + f0 + jf];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) acc += wr[jf] * (f0 + jf < j.F ? dq[jf] : 0.f);
            }
            *reinterpret_cast<f32x4*>(G + lds_off(p, slot)) = acc;
        }
    }
    __syncthreads();

#endif
    PFM_BSTAMP(2);
    const bool want_dt = dtemb != nullptr;
    if (want_dt && tid < j.T) {  // fc_l3 extras, KMAJOR [Ke][F]: d temb[k] = sum_f We3[k][f] db3j[f]   (db3j written before the barrier above)
        float a = 0.f;
        for (int f = 0; f < j.F; ++f) a = fmaf(blob[d.l3_We + tid * j.F + f], lds[c.dag2 + f], a);
        lds[c.dte + tid] += a;
    }
    if (want_dt) __syncthreads();
    f32x4 a1[8];
    // the dX products address this jet's saved record and gradient rows as buffers: wave-uniform block / row-base offset + a per-lane
    // constant (row pl of a tile, the lane's four output features), LDS rows as a per-lane constant + an immediate
    const blob_rsrc rs_sv = make_blob_rsrc(sv, sl.total);
#ifdef PFM_BWD_BUF_ST
    const blob_rsrc rs_da = make_blob_rsrc(daj, (int64_t)bw.nblk * j.N * H);
#endif
    const int gofs = (pl * H + 4 * oslot) * 4;          // bytes: [row pl][features 4 oslot ..] of a row-major (rows, H) block
    const int ooff = pl * H + ((oslot ^ pl) << 2);      // floats: the same element in a swizzled LDS tile (lds_off(pl, oslot))
    const float* const mrow = maskf + pl;
    // Gradient rows leave through ordinary global stores.  buffer_store_dwordx4 with an SGPR soffset gave WRONG rows on gfx950 (lanes
    // 12..15 of a tile, dwords 1 and 3, in the bodies of jets with an odd tile count: tests/diag/ab_bwd.py against the previous build):
    // hipcc assumes that form has no store-data hazard and reuses the data registers at once; the global_store form gets the hazard's
    // wait states.  (-DPFM_BWD_BUF_ST keeps the buffer form for that diagnosis; PFM_AB_*: timing-only ablations, tests/diag.)
#if defined(PFM_AB_NOSTORE)
#define PFM_DA_STORE(off, v) ((void)(v))
#elif defined(PFM_BWD_BUF_ST)
#define PFM_DA_STORE(off, v) bstore4(rs_da, (off), gofs, (v))
#else
#define PFM_DA_STORE(off, v) (*reinterpret_cast<f32x4*>(daj + (off) + (gofs >> 2)) = (v))
#endif
#if defined(PFM_AB_NOLOAD)
#define PFM_SV_LOAD(off) (f32x4{1.f, 1.f, 1.f, 1.f})
#elif defined(PFM_BWD_PLAIN_LD)
#define PFM_SV_LOAD(off) (*reinterpret_cast<const f32x4*>(sv + (off) + (gofs >> 2)))
#else
#define PFM_SV_LOAD(off) bload4(rs_sv, (off), gofs)
#endif
    // ---- EPiC layers, last to first ----
    for (int k = j.layers - 1; k >= 0; --k) {
        const pfm_epic_layer& ly = d.layer[k];
        const float* xo = sv + sl.xo + k * sl.lstride;                           // h_{k+1}
        const float* g1 = sv + sl.glayer + k * sl.gstride;
        const float* gout = g1 + H;
        const float* gin = (k > 0) ? sv + sl.glayer + (k - 1) * sl.gstride + H : sv + sl.gstem;
        PFM_BSTAMP(10);
        float* rstage = rec + (1 + k) * BwdRec::STAGE;
        float* da2 = daj + (size_t)(1 + 2 * k) * j.N * H;  // pairs with l1_k: dW of fc_local2 (epic.py:198-200)
        // da block 2 + 2k: da1, pairs with h_k: dW of fc_local1 (epic.py:194-196)
        // (1) da2 = G * phi'(h_{k+1}) in place (and to `da`); db2j = column sums.  Only for the last layer: for the others
        //     step (7) of the layer above has already done it in its epilogue.
        if (k == j.layers - 1) {
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < ntiles; t0 += 10) {  // every tile's global load in flight at once (the saved activations come from
                                                       // HBM: tile by tile, or five at a time, the pass was a chain of DRAM round trips)
                f32x4 hv[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) hv[u] = *reinterpret_cast<const f32x4*>(xo + min((t0 + u) * TILE + pl, n_rows - 1) * H + 4 * oslot);
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int p = (t0 + u) * TILE + pl;
                    if (t0 + u < ntiles && p < n_rows) {
                        f32x4 gv = *reinterpret_cast<f32x4*>(G + lds_off(p, oslot));
                        gv *= dlrelu4(hv[u], slope);
                        *reinterpret_cast<f32x4*>(G + lds_off(p, oslot)) = gv;
                        *reinterpret_cast<f32x4*>(da2 + p * H + 4 * oslot) = gv;
                        ps += gv;
                    }
                }
            }
            ps = colsum16(ps);
            if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.dbj2 + 4 * oslot) = ps;
            __syncthreads();
        }
        PFM_BSTAMP(11);
        // (3) da1 = (W_lc2^T da2) * phi'(l1) -> Hb (and to `da`); db1j = column sums
        {
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
            const int64_t l1_off = sl.l1 + (int64_t)k * sl.lstride, da1_off = (int64_t)(2 + 2 * k) * j.N * H;
            gemm_dx<BF16>(a2, G, n_rows,
                    [&](int rb) { return PFM_SV_LOAD(l1_off + rb * H); },
                    [&](int rb, bool valid, f32x4 acc, f32x4 lv) {
                        acc *= dlrelu4(lv, slope);
                        if (valid) {
                            *reinterpret_cast<f32x4*>(Hb + rb * H + ooff) = acc;
                            PFM_DA_STORE(da1_off + rb * H, acc);
                            ps += acc;
                        }
                    });
            ps = colsum16(ps);
            if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.dbj1 + 4 * oslot) = ps;
        }
        PFM_BSTAMP(12);
        // step (7)'s weights and the NEXT dX product's (step (3) of layer k - 1, or the stem's): they land behind
        // the per-jet steps below
        load_afrag(a1, rs, ly.lc1.AT, w, lane);
        load_afrag(a2, rs, k > 0 ? d.layer[k - 1].lc2.AT : d.l2.AT, w, lane);
#ifndef PFM_AB_NOCHAIN  // (timing-only ablation: the per-jet steps between the two dX products skipped)
        // vin of this stage (for the global backward): [temb;cond;mean_k;sum_k;g_k]
        build_vin(j, lds, c, sv + sl.pool + k * sl.pstride, gin, true);
        __syncthreads();
        // (4) per-jet pieces: dg_{k+1} += We_lc1[g rows] . db1j  (the extras / bias gradients of both local linears are the
        //     rank-1 sums  [temb ; cond_l ; g_{k+1}] (x) db1j,  [temb ; cond_l] (x) db2j  over jets: operands -> rec)
        km16_tgemv(blob + ly.lc1.We, Ke >> 4, (Ke + j.L - 1) >> 4, lds + c.dbj1, lds + c.tg);
        __syncthreads();
        if (tid >= 128 && tid < 128 + j.L) {
            const int jj = tid - 128;
            lds[c.dg + jj] += lds[c.tg + (Ke + jj) - 16 * (Ke >> 4)];
        }
        rec_put(rstage + BwdRec::DBJ1, lds + c.dbj1, H, 0);
        rec_put(rstage + BwdRec::DBJ2, lds + c.dbj2, H, 32);
        if (tid >= 64 && tid < 64 + MAXL) rstage[BwdRec::GOUT + tid - 64] = (tid - 64 < j.L) ? gout[tid - 64] : 0.f;
        __syncthreads();
        PFM_BSTAMP(13);
        if (want_dt) {  // time rows of the two extras blocks
            dtemb_from_extras(blob + ly.lc1.We, lds + c.dbj1, j.T, lds + c.tg, lds + c.dte);
            dtemb_from_extras(blob + ly.lc2.We, lds + c.dbj2, j.T, lds + c.tg, lds + c.dte);
        }
        // (5) global MLP backward -> dP_k, dg_k; its rank-1 operands -> rec
        global_backward<false>(j, blob, ly.gl1, ly.gl2, lds, c, g1, gout, want_dt);
        rec_put(rstage + BwdRec::VIN, lds + c.vin, VIN_FLOATS, 0);
        rec_put(rstage + BwdRec::VIN2, lds + c.vin2, VIN2_FLOATS, 128);
        rec_put(rstage + BwdRec::DAG1, lds + c.dag1, H, 192);
        rec_put(rstage + BwdRec::DAG2, lds + c.dag2, MAXL, 224);
#else
        __syncthreads();
#endif
        PFM_BSTAMP(14);
        // (7) dh_k = W_lc1^T da1 + da2 (residual) + mask * dP_k (pooling) -> G in place; for k > 0 times phi'(h_k) right away:
        //     that is da2 of layer k - 1 (step (1) of the next iteration, fused here: one pass over G and one L2 round trip less)
        {
            const f32x4 dP4 = *reinterpret_cast<const f32x4*>(lds + c.dP + 4 * oslot);
            if (k > 0) {
                f32x4 ps = {0.f, 0.f, 0.f, 0.f};
                const int64_t hin_off = sl.xo + (int64_t)(k - 1) * sl.lstride, da2n_off = (int64_t)(1 + 2 * (k - 1)) * j.N * H;
                gemm_dx<BF16>(a1, Hb, n_rows,
                        [&](int rb) { return PFM_SV_LOAD(hin_off + rb * H); },
                        [&](int rb, bool valid, f32x4 acc, f32x4 hv) {
                            if (valid) {
                                f32x4 gv = *reinterpret_cast<f32x4*>(G + rb * H + ooff);
                                gv = (gv + acc + dP4 * mrow[rb]) * dlrelu4(hv, slope);
                                *reinterpret_cast<f32x4*>(G + rb * H + ooff) = gv;
                                PFM_DA_STORE(da2n_off + rb * H, gv);
                                ps += gv;
                            }
                        });
                ps = colsum16(ps);
                if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.dbj2 + 4 * oslot) = ps;  // db2j of layer k - 1
            } else {
                gemm_dx<BF16>(a1, Hb, n_rows,
                        [&](int) { return f32x4{0.f, 0.f, 0.f, 0.f}; },
                        [&](int rb, bool valid, f32x4 acc, f32x4) {
                            if (valid) {
                                f32x4 gv = *reinterpret_cast<f32x4*>(G + rb * H + ooff);
                                gv += acc + dP4 * mrow[rb];
                                *reinterpret_cast<f32x4*>(G + rb * H + ooff) = gv;
                            }
                        });
            }
        }
        __syncthreads();
    }

    PFM_BSTAMP(20);
    // ---- stem ----
    // global stem backward (fc_g1 / fc_g2): dg_0 -> dP (the pool of x2 as seen by the stem MLP)
    build_vin(j, lds, c, sv + sl.pool, nullptr, false);
    __syncthreads();
    global_backward<true>(j, blob, d.g1, d.g2, lds, c, sv + sl.gstem1, sv + sl.gstem, want_dt);
    rec_put(rec + BwdRec::VIN, lds + c.vin, VIN_FLOATS, 0);
    rec_put(rec + BwdRec::VIN2, lds + c.vin2, VIN2_FLOATS, 128);
    rec_put(rec + BwdRec::DAG1, lds + c.dag1, H, 192);
    rec_put(rec + BwdRec::DAG2, lds + c.dag2, MAXL, 224);
    // (a2 = fc_l2's transposed block: requested behind step (3) of layer 0, or ahead of the head when there is no layer)
    // da2s = (G + mask * dP) * phi'(x2) in place; db2j
    {
        const float* x2 = sv + sl.x2;
        const f32x4 dP4 = *reinterpret_cast<const f32x4*>(lds + c.dP + 4 * oslot);
        f32x4 ps = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < ntiles; t0 += 10) {  // every tile's global load in flight at once
            f32x4 hv[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) hv[u] = *reinterpret_cast<const f32x4*>(x2 + min((t0 + u) * TILE + pl, n_rows - 1) * H + 4 * oslot);
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int p = (t0 + u) * TILE + pl;
                if (t0 + u < ntiles && p < n_rows) {
                    f32x4 gv = *reinterpret_cast<f32x4*>(G + lds_off(p, oslot));
                    gv = (gv + dP4 * maskf[p]) * dlrelu4(hv[u], slope);
                    *reinterpret_cast<f32x4*>(G + lds_off(p, oslot)) = gv;
                    *reinterpret_cast<f32x4*>(daj + p * H + 4 * oslot) = gv;  // block 0: pairs with x1 -> dW of fc_l2
                    ps += gv;
                }
            }
        }
        ps = colsum16(ps);
        if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.dbj2 + 4 * oslot) = ps;
    }
    __syncthreads();
    PFM_BSTAMP(21);
    // da1s = (W_l2^T da2s + da2s) * phi'(x1) -> Hb   (epic.py:364-366, 360-362)
    {
        f32x4 ps = {0.f, 0.f, 0.f, 0.f};
        gemm_dx<BF16>(a2, G, n_rows,
                [&](int rb) { return PFM_SV_LOAD(sl.x1 + rb * H); },
                [&](int rb, bool valid, f32x4 acc, f32x4 xv) {
                    if (valid) {
                        acc += *reinterpret_cast<const f32x4*>(G + rb * H + ooff);
                        acc *= dlrelu4(xv, slope);
                        *reinterpret_cast<f32x4*>(Hb + rb * H + ooff) = acc;
                        ps += acc;
                    }
                });
        ps = colsum16(ps);
        if (pl == 0) *reinterpret_cast<f32x4*>(lds + c.dbj1 + 4 * oslot) = ps;
    }
    __syncthreads();
    PFM_BSTAMP(22);
    rec_put(rec + BwdRec::DBJ1, lds + c.dbj1, H, 0);   // dWe of fc_l1 / fc_l2 = sum_jets [temb ; cond_l] (x) db1j / db2j
    rec_put(rec + BwdRec::DBJ2, lds + c.dbj2, H, 32);
    if (want_dt) {
        dtemb_from_extras(blob + d.l1_We, lds + c.dbj1, j.T, lds + c.tg, lds + c.dte);
        dtemb_from_extras(blob + d.l2.We, lds + c.dbj2, j.T, lds + c.tg, lds + c.dte);
        if (tid < j.T) dtemb[(size_t)jet * j.T + tid] = lds[c.dte + tid];
    }
    // this jet's part of dWx_l1[f][o] = sum_p y[p][f] * da1s[p][o]   (K-major [F][H]) -> rec.dWx
    // (y through LDS -- da3 is dead since the head, its alias tg since the last tgemv: scalar global loads inside the tile loop
    //  were an exposed L2 round trip per tile)
    __syncthreads();
    for (int i = tid; i < n_rows * j.F; i += NT) lds[c.da3 + i] = sv[sl.y + i];
    __syncthreads();
    for (int f0 = 0; f0 < j.F; f0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) acc[jf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tile = 0; tile < ntiles; ++tile) {
            const int p = tile * TILE + pl;
            if (p < n_rows) {
                const f32x4 dv = *reinterpret_cast<const f32x4*>(Hb + lds_off(p, oslot));
#pragma unroll
                for (int jf = 0; jf < 4; ++jf)
                    if (f0 + jf < j.F) acc[jf] += dv * lds[c.da3 + p * j.F + f0 + jf];
            }
        }
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) {
            if (f0 + jf < j.F) {
                const f32x4 s4 = colsum16(acc[jf]);
                if (pl == 0) *reinterpret_cast<f32x4*>(rec + br.dWx + (f0 + jf) * H + 4 * oslot) = s4;
            }
        }
    }
    if (dy) {
        // d loss / d y[p][f] = sum_o Wx[f][o] da1s[p][o]  (fc_l1's particle block, epic.py:360-362): what a flow in front of this one
        // needs (n_transforms > 1, flow_matching_module.py:421-443; losses.py:66-69 feeds each flow's output to the next).  da1s is
        // still in Hb; rows behind the last computed particle (and masked rows: their da1s is 0) get 0.
        __syncthreads();
        for (int i = tid; i < j.F * H; i += NT) lds[c.w3 + i] = blob[d.l1x.W + i];  // K-major [F][H]; w3 is dead since the head
        __syncthreads();
        float* dyj = dy + (size_t)jet * j.N * j.F;
        for (int i = tid; i < j.N * j.F; i += NT) {
            const int p = i / j.F, f = i - p * j.F;
            float a = 0.f;
            if (p < n_rows) {
                f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int s4 = 0; s4 < H / 4; ++s4)
                    a4 += *reinterpret_cast<const f32x4*>(Hb + lds_off(p, s4)) * *reinterpret_cast<const f32x4*>(lds + c.w3 + f * H + 4 * s4);
                a = (a4.x + a4.y) + (a4.z + a4.w);
            }
            dyj[i] = a;
        }
    }
    PFM_BSTAMP(30);
}

}  // namespace pfm
