// gfx950 kernels + C ABI: EPiC vector-field evaluation and the persistent midpoint sampler.
// One workgroup (512 threads) per jet; see epic_nfe.h for the on-chip mapping.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "epic_nfe.h"
#include "epic_fast.h"

#ifdef PFM_DIAG
namespace pfm {
__device__ unsigned long long g_pfm_stamps[512];
__device__ int g_pfm_nstamp;
}
extern "C" int pfm_diag_read_stamps(unsigned long long* out, int* n) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(n, HIP_SYMBOL(pfm::g_pfm_nstamp), sizeof(int));
    hipMemcpyFromSymbol(out, HIP_SYMBOL(pfm::g_pfm_stamps), sizeof(unsigned long long) * 512);
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(pfm::g_pfm_nstamp), &zero, sizeof(int));
    return 0;
}
#endif

// Diagnostic build only (-DPFM_WGT, tests/diag/wg_times.py; never the shipped library): every workgroup of the lean midpoint sampler
// records its start and end on the 100 MHz constant clock -- is a pipeline of launches short of its CU-time bound because workgroups
// wait for a CU, or because they run longer next to each other?
#ifdef PFM_WGT
namespace pfm {
constexpr int WGT_MAX = 32768;
__device__ unsigned long long g_pfm_wgt[2 * WGT_MAX];
__device__ int g_pfm_nwgt;
}
extern "C" int pfm_diag_read_wgt(unsigned long long* out, int cap, int* n) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(n, HIP_SYMBOL(pfm::g_pfm_nwgt), sizeof(int));
    const int m = *n < cap ? *n : cap;
    hipMemcpyFromSymbol(out, HIP_SYMBOL(pfm::g_pfm_wgt), sizeof(unsigned long long) * 2 * (m < pfm::WGT_MAX ? m : pfm::WGT_MAX));
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(pfm::g_pfm_nwgt), &zero, sizeof(int));
    return 0;
}
#define PFM_WGT_BEGIN                                         \
    unsigned long long wgt0_ = 0;                             \
    if (threadIdx.x == 0) wgt0_ = __builtin_amdgcn_s_memrealtime();
#define PFM_WGT_END                                                       \
    if (threadIdx.x == 0) {                                               \
        const int i_ = atomicAdd(&g_pfm_nwgt, 1);                         \
        if (i_ < WGT_MAX) {                                               \
            g_pfm_wgt[2 * i_] = wgt0_;                                    \
            g_pfm_wgt[2 * i_ + 1] = __builtin_amdgcn_s_memrealtime();     \
        }                                                                 \
    }
#else
#define PFM_WGT_BEGIN
#define PFM_WGT_END
#endif

namespace pfm {

thread_local char g_err[512] = "";

int set_err(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}
int check_hip(hipError_t e, const char* where) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

int validate(const pfm_epic_desc* d) {
    if (!d) return set_err(PFM_E_BADARG, "desc is NULL");
    if (d->abi_version != PFM_ABI_VERSION) return set_err(PFM_E_BADARG, "desc.abi_version mismatch");
    if (d->hidden != H) return set_err(PFM_E_BADARG, "this build is specialised for hidden = 128");
    if (d->layers < 0 || d->layers > PFM_MAX_LAYERS) return set_err(PFM_E_BADARG, "layers out of range");
    if (d->latent < 1 || d->latent > MAXL) return set_err(PFM_E_BADARG, "latent must be in 1..16");
    if (d->features < 1 || d->features > MAXF) return set_err(PFM_E_BADARG, "features must be in 1..16");
    if (d->t_dim < 0 || d->t_dim > MAXT) return set_err(PFM_E_BADARG, "t_dim must be in 0..64");
    if (d->cond_global < 0 || d->cond_global > MAXC) return set_err(PFM_E_BADARG, "cond_global must be in 0..16");
    if (d->cond_local != 0 && d->cond_local != d->cond_global)
        return set_err(PFM_E_BADARG, "cond_local must be 0 or cond_global");
    if (d->n_points < 13) return set_err(PFM_E_BADARG, "n_points must be >= 13");
    if (d->n_points > 2 * TILE * MAXPAIRS) return set_err(PFM_E_LDS, "set does not fit the LDS tile (n_points > 160: the particle phases are unrolled for at most 5 tile pairs)");
    if ((int64_t)make_carve(d->n_points, d->features).total * 4 > 163840)
        return set_err(PFM_E_LDS, "set does not fit the 160 KiB LDS tile (n_points too large for fp32, hidden 128)");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// v = f(t, x): one evaluation per jet
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void epic_forward_kernel(const float* __restrict__ blob, int64_t desc_off,
                                                             const float* __restrict__ t, const float* __restrict__ temb,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ cond,
                                                             const float* __restrict__ mask, float* __restrict__ v) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const Carve c = make_carve(j.N, j.F);
    const int jet = blockIdx.x, tid = threadIdx.x;
    const int n_rows = epic_jet_setup(d, j, blob, lds, c, cond ? cond + (size_t)jet * j.C : nullptr,
                                      mask ? mask + (size_t)jet * j.N : nullptr);
    const float* xj = x + (size_t)jet * j.N * j.F;
    for (int i = tid; i < j.N * j.F; i += NT) lds[c.yin + i] = xj[i];
    if (temb) {  // caller-supplied embedding (EPiC_encoder.forward(t_in, ...) signature, epic.py:304-310)
        if (tid < j.T) {
            lds[c.vin + tid] = temb[(size_t)jet * j.T + tid];
            lds[c.vin2 + tid] = lds[c.vin + tid];
        }
    } else {
        epic_time_embedding(d, j, blob, lds, c, t[jet]);
    }
    __syncthreads();
    const SavedLayout sl = make_saved(j.N, j.F, j.layers);
    PFM_STAMP(0);
    epic_body<false, MODE>(d, j, blob, lds, c, n_rows, nullptr, sl);
    float* vj = v + (size_t)jet * j.N * j.F;
    const int F = j.F;
    epic_head<MODE>(d, j, blob, lds, c, n_rows, [=](int p, int f, float val) { vj[p * F + f] = val; });
    PFM_STAMP(30);
}

// One evaluation of the vector field inside the persistent sampler + the integrator update.  (Tried as a real,
// non-inlined function to isolate its register allocation: the call ABI's callee-saved spills made it 25 % slower.)
//   stage 0: x_mid = x + 0.5*dt*k1 -> next input;   stage 1: x = x + dt*f(t+dt/2, x_mid)
template <int MODE, bool TB, int NSEG = 1>
static __device__ __forceinline__ void sampler_eval(const float* __restrict__ blob, int64_t desc_off, int n_rows,
                                                      float t, float hs, int stage, const float* __restrict__ tb,
                                                      const Segs* sg = nullptr, const float* __restrict__ temb_row = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const Carve c = make_carve(j.N, j.F);
    const SavedLayout sl = make_saved(j.N, j.F, j.layers);
    float* xs = lds + c.xs;
    float* yin = lds + c.yin;
    const int F = j.F;
    if (temb_row) epic_time_embedding_from(j, lds, c, temb_row, NSEG == 2);
    else epic_time_embedding(d, j, blob, lds, c, t, NSEG == 2);
    __syncthreads();
    epic_body<false, MODE, TB, NSEG>(d, j, blob, lds, c, n_rows, nullptr, sl, tb, sg);
    epic_head<MODE, NSEG>(d, j, blob, lds, c, n_rows, [=](int p, int f, float val) {
        const float xn = __fadd_rn(xs[p * F + f], __fmul_rn(hs, val));
        yin[p * F + f] = xn;
        if (stage) xs[p * F + f] = xn;
    }, sg);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Two jets in one workgroup (packed sampler).  A jet's evaluation costs a fixed ~100k cycles (per-jet GEMV phases, stem, head,
// phase fill / drain: ~330 KB of weights through the CU's 64 B/clk path per layer) plus ~30k per 16-row tile; for a 2-tile jet the
// fixed part is two thirds of the time.  Two short jets whose rows fit the LDS tile together share ONE weight stream and ONE set of
// phases: rows [0, n0) = jet A, [r1, r1 + n1) = jet B with r1 = n0 rounded up to a tile; rows in between are holes (zero input,
// zero mask: finite, never pooled).  Results are those of the one-jet kernel (same arithmetic per row and per jet).
// epic_pair_setup: masks, valid counts, conditioning, start state of both jets; returns the segment description.
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ Segs epic_pair_setup(const pfm_epic_desc& d, const JetDims& j, float* __restrict__ lds, const Carve& c,
                                                         int jetA, int jetB, const float* __restrict__ z, const float* __restrict__ cond,
                                                         const float* __restrict__ mask) {
    const int tid = threadIdx.x;
    const SegView v0 = seg_view(c, j.N, 0), v1 = seg_view(c, j.N, 1);
    const float* mA = mask + (size_t)jetA * j.N;
    const float* mB = mask + (size_t)jetB * j.N;
    // last valid particle and valid count of both jets (wave 0..3: jet A, 4..7: jet B would do; simpler: everyone scans both)
    int lastA = -1, lastB = -1;
    float cntA = 0.f, cntB = 0.f;
    for (int p = tid; p < j.N; p += NT) {
        const float a = mA[p], b = mB[p];
        cntA += a; cntB += b;
        if (a != 0.f) lastA = p;
        if (b != 0.f) lastB = p;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        cntA += __shfl_xor(cntA, m); cntB += __shfl_xor(cntB, m);
        lastA = max(lastA, __shfl_xor(lastA, m)); lastB = max(lastB, __shfl_xor(lastB, m));
    }
    float* red = lds + c.misc + 8;  // 16 floats of scratch + 8 more in the second view
    float* red2 = lds + v1.misc;
    if ((tid & 63) == 0) {
        red[tid >> 6] = cntA; red[8 + (tid >> 6)] = (float)lastA;
        red2[tid >> 6] = cntB;
    }
    __syncthreads();
    float sA = 0.f, sB = 0.f, lA = -1.f;
    for (int i = 0; i < NW; ++i) { sA += red[i]; lA = fmaxf(lA, red[8 + i]); sB += red2[i]; }
    __syncthreads();
    // lastB through a second round (red2 holds 8 floats only)
    if ((tid & 63) == 0) red2[tid >> 6] = (float)lastB;
    __syncthreads();
    float lB = -1.f;
    for (int i = 0; i < NW; ++i) lB = fmaxf(lB, red2[i]);
    __syncthreads();
    Segs sg;
    sg.nseg = 2;
    sg.n0 = (int)lA + 1;
    sg.n1 = (int)lB + 1;
    sg.r1 = (sg.n0 + TILE - 1) / TILE * TILE;
    sg.rows = sg.r1 + sg.n1;
    if (tid == 0) {
        lds[v0.misc] = sA; lds[v0.misc + 1] = lA; lds[v1.misc] = sB;
        lds[v0.misc + 2] = 1.0f / sA; lds[v1.misc + 2] = 1.0f / sB;  // pool_finish's reciprocals
    }
    const int F = j.F;
    const float* zA = z + (size_t)jetA * j.N * F;
    const float* zB = z + (size_t)jetB * j.N * F;
    for (int p = tid; p < j.N; p += NT) {  // true mask of every row; second jet's mask on its own rows
        float m = 0.f, m1 = 0.f;
        if (p < sg.n0) m = mA[p];
        else if (p >= sg.r1 && p < sg.rows) m = m1 = mB[p - sg.r1];
        lds[v0.maskf + p] = m;
        lds[v1.maskf + p] = m1;
    }
    for (int i = tid; i < j.N * F; i += NT) {
        const int p = i / F, f = i - p * F;
        float z0 = 0.f;
        if (p < sg.n0) z0 = zA[i] * mA[p];                                               // flow_matching_module.py:668-671
        else if (p >= sg.r1 && p < sg.rows) z0 = zB[(p - sg.r1) * F + f] * mB[p - sg.r1];
        lds[c.xs + i] = z0;
        lds[c.yin + i] = z0;
    }
    if (tid < j.C) {
        const float a = cond[(size_t)jetA * j.C + tid], b = cond[(size_t)jetB * j.C + tid];
        lds[v0.vin + j.T + tid] = a; lds[v0.vin2 + j.T + tid] = a;
        lds[v1.vin + j.T + tid] = b; lds[v1.vin2 + j.T + tid] = b;
    }
    if (tid >= 64 && tid < 64 + MAXL) {
        lds[v0.vin + j.T + j.C + 2 * H + (tid - 64)] = 0.f;
        lds[v1.vin + j.T + j.C + 2 * H + (tid - 64)] = 0.f;
    }
    __syncthreads();
    return sg;
}

// ------------------------------------------------------------------------------------------------
// Persistent fixed-step midpoint integrator: all 2*n_intervals evaluations of a jet in one launch,
// state and activations never leave the CU.  (torchdyn Midpoint.step restated in oracle/fm_ref.py)
// ------------------------------------------------------------------------------------------------
template <int MODE, bool TB>
__global__ __launch_bounds__(NT, 2) void epic_sample_midpoint_kernel(
    const float* __restrict__ blob, int64_t desc_off, const float* __restrict__ t_eval,
    const float* __restrict__ dt, int n_intervals, const float* __restrict__ z, const float* __restrict__ cond,
    const float* __restrict__ mask, float* __restrict__ x_out, const float* __restrict__ table, const int* __restrict__ pack,
    const float* __restrict__ temb_tab) {  // temb_tab (or NULL): [2 n_intervals][T] embedding of every evaluation time, from the caller
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d0 = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d0);
    const Carve c = make_carve(j.N, j.F);
    const int tid = threadIdx.x;
    // pack (epic_jet_pack_kernel): [0] = workgroups in use, then (jet A, jet B or -1) per workgroup, longest first
    int jetA = blockIdx.x, jetB = -1;
    if (pack) {
        if ((int)blockIdx.x >= pack[0]) return;
        jetA = pack[1 + 2 * blockIdx.x];
        jetB = pack[2 + 2 * blockIdx.x];
    }
    float* xs = lds + c.xs;
    const int F = j.F;
    if constexpr (MODE != 2) if (jetB >= 0) {
        // ---- two jets ----
        const Segs sg = epic_pair_setup(d0, j, lds, c, jetA, jetB, z, cond, mask);
        for (int e = 0; e < 2 * n_intervals; ++e) {
#ifdef PFM_DIAG
            if (e == 2 * n_intervals - 1 && blockIdx.x == 0 && threadIdx.x == 0) g_pfm_nstamp = 0;  // keep the last NFE
            PFM_STAMP(0);
#endif
            const int stage = e & 1;
            const float h = dt[e >> 1];
            const float hs = stage ? h : __fmul_rn(0.5f, h);
            sampler_eval<MODE, TB, 2>(blob, desc_off, sg.rows, t_eval[e], hs, stage, TB ? table + (size_t)e * (j.layers + 1) * TB_SLOT : nullptr, &sg,
                                      temb_tab ? temb_tab + (size_t)e * j.T : nullptr);
            PFM_STAMP(30);
        }
        float* oA = x_out + (size_t)jetA * j.N * F;
        float* oB = x_out + (size_t)jetB * j.N * F;
        for (int i = tid; i < j.N * F; i += NT) {
            const int p = i / F;
            oA[i] = p < sg.n0 ? xs[i] : 0.f;                      // rows behind a jet's last valid particle are masked: 0
            oB[i] = p < sg.n1 ? xs[sg.r1 * F + i] : 0.f;
        }
        return;
    }
    const int jet = jetA;
    const int n_rows = epic_jet_setup(d0, j, blob, lds, c, cond ? cond + (size_t)jet * j.C : nullptr,
                                      mask ? mask + (size_t)jet * j.N : nullptr);
    const float* zj = z + (size_t)jet * j.N * j.F;
    for (int i = tid; i < j.N * j.F; i += NT) {
        const float z0 = zj[i] * lds[c.maskf + i / j.F];  // flow_matching_module.py:668-671
        lds[c.xs + i] = z0;
        lds[c.yin + i] = z0;
    }
    // 2*n_intervals evaluations; even = k1 at t_k, odd = slope at the midpoint (one inlined body)
    for (int e = 0; e < 2 * n_intervals; ++e) {
#ifdef PFM_DIAG
        if (e == 2 * n_intervals - 1 && blockIdx.x == 0 && threadIdx.x == 0) g_pfm_nstamp = 0;  // keep the last NFE
        PFM_STAMP(0);
#endif
        const int stage = e & 1;
        const float h = dt[e >> 1];
        const float hs = stage ? h : __fmul_rn(0.5f, h);
        sampler_eval<MODE, TB>(blob, desc_off, n_rows, t_eval[e], hs, stage, TB ? table + (size_t)e * (j.layers + 1) * TB_SLOT : nullptr, nullptr,
                               temb_tab ? temb_tab + (size_t)e * j.T : nullptr);
        PFM_STAMP(30);
    }
    float* oj = x_out + (size_t)jet * j.N * j.F;
    for (int i = tid; i < j.N * j.F; i += NT) oj[i] = xs[i];
}

// ------------------------------------------------------------------------------------------------
// The same integrator on the lean evaluation of epic_fast.h (unconditioned jets, T = 32, F <= 4; fp32 or bf16 operands): one jet per
// workgroup, jets in descending multiplicity (`pack`), every time-only term from the fast-format table.
// ------------------------------------------------------------------------------------------------
// PAIRS: the launch may hold two-jet workgroups (PFM_F_PACK_JETS); the one-jet instantiation does not carry that path (its register
// allocation and code size are the single jet's own).
// COND: conditioned jets (one per workgroup); ctab = the call's cond table [n_jets][layers + 1][TB_SLOT] (epic_cond_table_kernel).
template <int MODE, bool PAIRS, bool COND = false>
__global__ __launch_bounds__(NT, 2) void epic_sample_midpoint_fast_kernel(
    const float* __restrict__ blob, int64_t desc_off, const float* __restrict__ dt, int n_intervals, const float* __restrict__ z,
    const float* __restrict__ mask, float* __restrict__ x_out, const float* __restrict__ table, const int* __restrict__ pack,
    const float* __restrict__ ctab) {
    static_assert(!(PAIRS && COND), "conditioned jets: one jet per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d0 = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d0);
    const Carve c = make_carve(j.N, j.F);
    const int tid = threadIdx.x;
    // pack (epic_jet_pack_kernel): [0] = workgroups in use, then (jet A, jet B or -1) per workgroup, longest first
    int jetA = blockIdx.x, jetB = -1;
    if (pack) {
        if ((int)blockIdx.x >= pack[0]) return;
        jetA = pack[1 + 2 * blockIdx.x];
        jetB = pack[2 + 2 * blockIdx.x];
    }
    PFM_WGT_BEGIN
    float* xs = lds + c.xs;
    float* yin = lds + c.yin;
    const int F = j.F;
    const size_t estride = (size_t)(j.layers + 1) * TB_SLOT;
    const int n_evals = 2 * n_intervals;
    FastCarry<MODE == 1> cy;
    cy.aw = fast_l1_weight(d0, j, blob);
    if constexpr (PAIRS) if (jetB >= 0) {
        // ---- two jets: rows [0, n0) = jet A, [r1, r1 + n1) = jet B (epic_pair_setup), one weight stream and one set of phases ----
        const Segs sg = epic_pair_setup(d0, j, lds, c, jetA, jetB, z, nullptr, mask);
        fast_carry_request(cy, d0, make_blob_rsrc(blob, d0.blob_floats + PFM_DESC_FLOATS), table + (size_t)j.layers * TB_SLOT);
        for (int e = 0; e < n_evals; ++e) {
            const int stage = e & 1;
            const float h = dt[e >> 1];
            const float hs = stage ? h : __fmul_rn(0.5f, h);
            fast_eval<MODE == 1, 2, false>(d0, j, blob, lds, c, sg.rows, table + e * estride, table + (e + 1 < n_evals ? e + 1 : e) * estride, cy,
                                    [=](int p, int f, float val) {
                                        const float xn = __fadd_rn(xs[p * F + f], __fmul_rn(hs, val));
                                        yin[p * F + f] = xn;
                                        if (stage) xs[p * F + f] = xn;
                                    }, &sg);
            __syncthreads();
        }
        float* oA = x_out + (size_t)jetA * j.N * F;
        float* oB = x_out + (size_t)jetB * j.N * F;
        for (int i = tid; i < j.N * F; i += NT) {
            const int p = i / F;
            oA[i] = p < sg.n0 ? xs[i] : 0.f;  // rows behind a jet's last valid particle are masked: 0
            oB[i] = p < sg.n1 ? xs[sg.r1 * F + i] : 0.f;
        }
        PFM_WGT_END
        return;
    }
    const int jet = jetA;
    const int n_rows = epic_jet_setup(d0, j, blob, lds, c, nullptr, mask ? mask + (size_t)jet * j.N : nullptr);
    const float* zj = z + (size_t)jet * j.N * j.F;
    for (int i = tid; i < j.N * j.F; i += NT) {
        const float z0 = zj[i] * lds[c.maskf + i / j.F];  // flow_matching_module.py:668-671
        lds[c.xs + i] = z0;
        lds[c.yin + i] = z0;
    }
    const float* ct = COND ? ctab + (size_t)jet * estride : nullptr;
    if (COND) fast_cond_zero(j, lds, c);
    __syncthreads();
    fast_carry_request(cy, d0, make_blob_rsrc(blob, d0.blob_floats + PFM_DESC_FLOATS), table + (size_t)j.layers * TB_SLOT,
                       COND ? ct + (size_t)j.layers * TB_SLOT : nullptr);
    for (int e = 0; e < n_evals; ++e) {
#ifdef PFM_DIAG
        if (e == n_evals - 1 && blockIdx.x == 0 && threadIdx.x == 0) g_pfm_nstamp = 0;  // keep the last NFE
        PFM_STAMP(0);
#endif
        const int stage = e & 1;
        const float h = dt[e >> 1];
        const float hs = stage ? h : __fmul_rn(0.5f, h);
        // stage 0: x_mid = x + 0.5*dt*k1 -> next input;   stage 1: x = x + dt*f(t+dt/2, x_mid)
        fast_eval<MODE == 1, 1, COND>(d0, j, blob, lds, c, n_rows, table + e * estride, table + (e + 1 < n_evals ? e + 1 : e) * estride, cy,
                                      [=](int p, int f, float val) {
                                          const float xn = __fadd_rn(xs[p * F + f], __fmul_rn(hs, val));
                                          yin[p * F + f] = xn;
                                          if (stage) xs[p * F + f] = xn;
                                      }, nullptr, ct);
        __syncthreads();
        PFM_STAMP(30);
    }
    float* oj = x_out + (size_t)jet * j.N * j.F;
    for (int i = tid; i < j.N * j.F; i += NT) oj[i] = xs[i];
    PFM_WGT_END
}

// ------------------------------------------------------------------------------------------------
// The midpoint integrator on the lean evaluation with FOUR jets per workgroup (quad mode, epic_fast.h): workgroup k takes the jets
// 4 k .. 4 k + 3 (the last one possibly fewer), jet s in the fixed 32-row slot [32 s, 32 s + 32).  Only the first QROWS rows of a jet
// are read (PFM_F_QUAD_JETS: the caller pads sets of <= 32 particles to the 128-row tile); rows behind them come back as 0.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void epic_sample_midpoint_quad_kernel(
    const float* __restrict__ blob, int64_t desc_off, const float* __restrict__ dt, int n_intervals, const float* __restrict__ z,
    const float* __restrict__ mask, float* __restrict__ x_out, const float* __restrict__ table, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d0 = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d0);
    const Carve c = make_carve(j.N, j.F);
    const int tid = threadIdx.x, F = j.F;
    const int jet0 = QJETS * blockIdx.x;
    const int nseg = min(QJETS, B - jet0);  // jets in this workgroup (wave-uniform)
    float* xs = lds + c.xs;
    float* yin = lds + c.yin;
    // ---- setup: masks, valid counts, start state of the (up to) four jets ----
    if (tid < QUAD_TILE_ROWS) {  // one thread per row: waves 0, 1 hold two jets each (32 lanes per jet)
        const int s = tid >> 5, p = tid & 31;
        const float m = s < nseg ? mask[(size_t)(jet0 + s) * j.N + p] : 0.f;
        lds[c.maskf + tid] = m;
        float cnt = m;
        for (int sh = 16; sh >= 1; sh >>= 1) cnt += __shfl_xor(cnt, sh);  // over the jet's 32 lanes
        if (p == 0) {
            lds[quad_view(c, s).misc] = cnt;
            lds[quad_view(c, s).misc + 2] = 1.0f / cnt;  // pool_finish's reciprocal
        }
    }
    for (int i = tid; i < QUAD_TILE_ROWS * F; i += NT) {
        const int r = i / F, f = i - r * F, s = r >> 5, p = r & 31;
        float z0 = 0.f;
        if (s < nseg) z0 = z[((size_t)(jet0 + s) * j.N + p) * F + f] * mask[(size_t)(jet0 + s) * j.N + p];  // flow_matching_module.py:668-671
        xs[i] = z0;
        yin[i] = z0;
    }
    if (tid < QJETS * MAXL) {  // every jet's g starts at 0 (the stem chain overwrites it; the padding entries must be 0)
        const SegView v = quad_view(c, tid / MAXL);
        lds[v.vin + FT + 2 * H + (tid % MAXL)] = 0.f;
    }
    __syncthreads();
    const int n_rows = QROWS * nseg;
    const size_t estride = (size_t)(j.layers + 1) * TB_SLOT;
    const int n_evals = 2 * n_intervals;
    FastCarry<MODE == 1> cy;
    cy.aw = fast_l1_weight(d0, j, blob);
    fast_carry_request(cy, d0, make_blob_rsrc(blob, d0.blob_floats + PFM_DESC_FLOATS), table + (size_t)j.layers * TB_SLOT);
    for (int e = 0; e < n_evals; ++e) {
        const int stage = e & 1;
        const float h = dt[e >> 1];
        const float hs = stage ? h : __fmul_rn(0.5f, h);
        fast_eval<MODE == 1, 4, false>(d0, j, blob, lds, c, n_rows, table + e * estride, table + (e + 1 < n_evals ? e + 1 : e) * estride, cy,
                                       [=](int p, int f, float val) {
                                           const float xn = __fadd_rn(xs[p * F + f], __fmul_rn(hs, val));
                                           yin[p * F + f] = xn;
                                           if (stage) xs[p * F + f] = xn;
                                       });
        __syncthreads();
    }
    for (int i = tid; i < nseg * j.N * F; i += NT) {  // rows behind the slot are masked: 0
        const int s = i / (j.N * F), r = i - s * j.N * F, p = r / F;
        x_out[(size_t)(jet0 + s) * j.N * F + r] = p < QROWS ? xs[s * QROWS * F + r] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent fixed-step explicit Runge-Kutta integrator (pfm_rk_tableau: euler, midpoint, torchdyn's rk4 = 3/8 rule):
// as above, with the stage slopes of the jet parked in global scratch (every element is written and read back by the
// same lane; the LDS carve has no room for four more N x F tiles at N = 150).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void epic_sample_rk_kernel(
    const float* __restrict__ blob, int64_t desc_off, pfm_rk_tableau tab, const float* __restrict__ t_eval,
    const float* __restrict__ dt, int n_intervals, const float* __restrict__ z, const float* __restrict__ cond,
    const float* __restrict__ mask, float* __restrict__ x_out, float* __restrict__ kbuf, const float* __restrict__ rhs,
    const int* __restrict__ order, const float* __restrict__ temb_tab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d0 = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d0);
    const Carve c = make_carve(j.N, j.F);
    const int jet = order ? order[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
    const int n_rows = epic_jet_setup(d0, j, blob, lds, c, cond ? cond + (size_t)jet * j.C : nullptr,
                                      mask ? mask + (size_t)jet * j.N : nullptr);
    const int NF = j.N * j.F, F = j.F, S = tab.stages;
    const float* zj = z + (size_t)jet * NF;
    for (int i = tid; i < NF; i += NT) {
        const float z0 = zj[i] * lds[c.maskf + i / F];
        lds[c.xs + i] = z0;
        lds[c.yin + i] = z0;
    }
    const SavedLayout sl = make_saved(j.N, j.F, j.layers);
    float* xs = lds + c.xs;
    float* yin = lds + c.yin;
    float* kj = kbuf + (size_t)jet * S * NF;
    int st = 0;
    for (int e = 0; e < S * n_intervals; ++e) {
        const float h = dt[e / S];
        const bool last = st == S - 1;
        const float* coef = last ? tab.b : tab.a[st + 1 < PFM_RK_MAX_STAGES ? st + 1 : 0];
        if (temb_tab) epic_time_embedding_from(j, lds, c, temb_tab + (size_t)e * j.T);
        else epic_time_embedding(d0, j, blob, lds, c, t_eval[e]);
        __syncthreads();
        epic_body<false, MODE>(d0, j, blob, lds, c, n_rows, nullptr, sl);
        const float r0 = rhs ? rhs[2 * e] : 0.f, r1 = rhs ? rhs[2 * e + 1] : 1.f;
        epic_head<MODE>(d0, j, blob, lds, c, n_rows, [=](int p, int f, float val) {
            const int i = p * F + f;
            if (rhs) val = __fmul_rn(r0, __fsub_rn(yin[i], __fdiv_rn(val, r1)));  // -0.5 beta (x - eps_theta / noise_rate)
            if (!last) kj[st * NF + i] = val;
            float acc = __fmul_rn(coef[0], st == 0 ? val : kj[i]);
            for (int q = 1; q <= st; ++q) acc = __fadd_rn(acc, __fmul_rn(coef[q], q == st ? val : kj[q * NF + i]));
            const float xn = __fadd_rn(xs[i], __fmul_rn(h, acc));
            yin[i] = xn;
            if (last) xs[i] = xn;
        });
        __syncthreads();
        st = last ? 0 : st + 1;
    }
    float* oj = x_out + (size_t)jet * NF;
    for (int i = tid; i < NF; i += NT) oj[i] = xs[i];
}

// ------------------------------------------------------------------------------------------------
// The Runge-Kutta integrator on the lean evaluation of epic_fast.h (unconditioned jets, T = 32, F <= 4): euler / rk4 / any tableau,
// and the probability-flow ODE of a diffusion model (`rhs`).  table: fast-format time table of all stages * n_intervals evaluation
// times.  Rows behind a jet's last valid particle are never evaluated: their state is z * mask = 0 and stays 0.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void epic_sample_rk_fast_kernel(
    const float* __restrict__ blob, int64_t desc_off, pfm_rk_tableau tab, const float* __restrict__ dt, int n_intervals,
    const float* __restrict__ z, const float* __restrict__ mask, float* __restrict__ x_out, float* __restrict__ kbuf,
    const float* __restrict__ rhs, const int* __restrict__ order, const float* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d0 = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d0);
    const Carve c = make_carve(j.N, j.F);
    const int jet = order ? order[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
    const int n_rows = epic_jet_setup(d0, j, blob, lds, c, nullptr, mask ? mask + (size_t)jet * j.N : nullptr);
    const int NF = j.N * j.F, F = j.F, S = tab.stages;
    const float* zj = z + (size_t)jet * NF;
    for (int i = tid; i < NF; i += NT) {
        const float z0 = zj[i] * lds[c.maskf + i / F];
        lds[c.xs + i] = z0;
        lds[c.yin + i] = z0;
    }
    __syncthreads();
    float* xs = lds + c.xs;
    float* yin = lds + c.yin;
    float* kj = kbuf + (size_t)jet * S * NF;
    const size_t estride = (size_t)(j.layers + 1) * TB_SLOT;
    const int n_evals = S * n_intervals;
    FastCarry<MODE == 1> cy;
    cy.aw = fast_l1_weight(d0, j, blob);
    fast_carry_request(cy, d0, make_blob_rsrc(blob, d0.blob_floats + PFM_DESC_FLOATS), table + (size_t)j.layers * TB_SLOT);
    int st = 0;
    for (int e = 0; e < n_evals; ++e) {
        const float h = dt[e / S];
        const bool last = st == S - 1;
        const float* coef = last ? tab.b : tab.a[st + 1 < PFM_RK_MAX_STAGES ? st + 1 : 0];
        const float r0 = rhs ? rhs[2 * e] : 0.f, r1 = rhs ? rhs[2 * e + 1] : 1.f;
        fast_eval<MODE == 1, 1, false>(d0, j, blob, lds, c, n_rows, table + e * estride, table + (e + 1 < n_evals ? e + 1 : e) * estride, cy,
                                [=](int p, int f, float val) {
                                    const int i = p * F + f;
                                    if (rhs) val = __fmul_rn(r0, __fsub_rn(yin[i], __fdiv_rn(val, r1)));  // -0.5 beta (x - eps_theta / noise_rate)
                                    if (!last) kj[st * NF + i] = val;
                                    float acc = __fmul_rn(coef[0], st == 0 ? val : kj[i]);
                                    for (int q = 1; q <= st; ++q) acc = __fadd_rn(acc, __fmul_rn(coef[q], q == st ? val : kj[q * NF + i]));
                                    const float xn = __fadd_rn(xs[i], __fmul_rn(h, acc));
                                    yin[i] = xn;
                                    if (last) xs[i] = xn;
                                });
        __syncthreads();
        st = last ? 0 : st + 1;
    }
    float* oj = x_out + (size_t)jet * NF;
    for (int i = tid; i < NF; i += NT) oj[i] = xs[i];
}

// ------------------------------------------------------------------------------------------------
// Time-term table of a sampling call (epic_nfe.h: TB): table[e][layer][TB_SLOT] = W_t^T temb(t_eval[e]) for the four per-jet
// Linears of every EPiC layer (fc_global1, local-1 extras, local-2 extras: KM16 blocks; fc_global2: KP16), time rows = the
// first T rows of each block; slot `layers` of an evaluation is the stem slot of the fast format (epic_fast.h).
// grid (n_evals, layers + 1), 512 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void epic_time_table_kernel(const float* __restrict__ blob, int64_t desc_off,
                                                              const float* __restrict__ t_eval, float* __restrict__ table,
                                                              const float* __restrict__ temb_tab, int fast) {
    __shared__ float temb[MAXT];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const int e = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    if (temb_tab) {
        if (tid < j.T) temb[tid] = temb_tab[(size_t)e * j.T + tid];
    } else if (tid < j.T) {  // the op order of epic_time_embedding
        const float t = t_eval[e], f = blob[d.freqs + tid];
        float v;
        if (d.flags & PFM_F_TEMB_SINCOS) {
            const float arg = __fmul_rn(f, t);
            v = 2 * tid < j.T ? cosf(arg) : sinf(arg);
        } else {
            v = cosf(__fdiv_rn(__fmul_rn(__fmul_rn(__fadd_rn(t, 0.0f), f), 3.14159274101257324f), 1.0f));
        }
        temb[tid] = v;
    }
    __syncthreads();
    float* out = table + ((size_t)e * (j.layers + 1) + k) * TB_SLOT;
    const int which = tid >> 7, o = tid & 127;
    if (k == j.layers) {
        // stem slot (fast format only; epic_fast.h): complete per-jet biases of fc_l1 / fc_l2 / fc_l3, bias + time term of fc_g1 / fc_g2
        if (!fast) return;
        if (which < 3) {  // 0: fc_l1 extras, 1: fc_l2 extras, 2: fc_g1 time rows
            const int64_t W = which == 0 ? d.l1_We : (which == 1 ? d.l2.We : d.g1.W);
            const int64_t b = which == 0 ? d.l1_b : (which == 1 ? d.l2.b : d.g1.b);
            float s = 0.f;
            for (int r = 0; r < j.T; ++r) s = fmaf(blob[W + km16(r, o)], temb[r], s);
            out[which * 128 + o] = s + blob[b + o];
        } else if (o < 16) {  // fc_g2 time rows (KP16)
            float s = 0.f;
            for (int r = 0; r < j.T; ++r) s = fmaf(blob[d.g2.W + (r >> 4) * 256 + (r & 15) * 16 + o], temb[r], s);
            out[TB_SG2 + o] = s + blob[d.g2.b + o];  // bias padded to 16 entries
        } else if (o < 32) {  // fc_l3: b3[f] + We3[:, f] . temb, zero for f >= F
            const int f = o - 16;
            float s = 0.f;
            if (f < j.F) {
                for (int r = 0; r < j.T; ++r) s = fmaf(blob[d.l3_We + r * j.F + f], temb[r], s);
                s += blob[d.l3_b + f];
            }
            out[TB_SB3 + f] = s;
        }
        return;
    }
    // layer slot: the time terms of fc_global1 | local-1 extras | local-2 extras (KM16 blocks) | fc_global2 (KP16);
    // fast format: with the Linear's bias added (b + t, the sum the generic chain forms in registers)
    const pfm_epic_layer& ly = d.layer[k];
    if (which < 3) {
        const int64_t W = which == 0 ? ly.gl1.W : (which == 1 ? ly.lc1.We : ly.lc2.We);
        const int64_t b = which == 0 ? ly.gl1.b : (which == 1 ? ly.lc1.b : ly.lc2.b);
        float s = 0.f;
        for (int r = 0; r < j.T; ++r) s = fmaf(blob[W + km16(r, o)], temb[r], s);
        out[which * 128 + o] = fast ? blob[b + o] + s : s;
    } else if (o < 16) {
        float s = 0.f;
        for (int r = 0; r < j.T; ++r) s = fmaf(blob[ly.gl2.W + (r >> 4) * 256 + (r & 15) * 16 + o], temb[r], s);
        out[TB_G2 + o] = fast ? blob[ly.gl2.b + o] + s : s;
    }
}

// ------------------------------------------------------------------------------------------------
// Cond table of a sampling call (conditioned jets on the lean evaluation, epic_fast.h): ctab[jet][slot][TB_SLOT] = W_c^T cond[jet] for
// the conditioning rows (rows T .. T + C of every global block, T .. T + Cl of every extras block) of the Linears the time table covers,
// same slot layout (layer slots, stem slot last), no bias.  grid (n_jets, layers + 1), 512 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void epic_cond_table_kernel(const float* __restrict__ blob, int64_t desc_off, const float* __restrict__ cond,
                                                              float* __restrict__ ctab) {
    __shared__ float cj[MAXC];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const int jet = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    if (tid < j.C) cj[tid] = cond[(size_t)jet * j.C + tid];
    __syncthreads();
    float* out = ctab + ((size_t)jet * (j.layers + 1) + k) * TB_SLOT;
    const int which = tid >> 7, o = tid & 127;
    const bool stem = k == j.layers;
    if (which < 3) {
        // layer slot: 0 fc_global1 (C rows) | 1 local-1 extras (Cl) | 2 local-2 extras (Cl); stem slot: 0 fc_l1 extras (Cl) | 1 fc_l2
        // extras (Cl) | 2 fc_g1 (C)
        int64_t W;
        int rows;
        if (stem) { W = which == 0 ? d.l1_We : (which == 1 ? d.l2.We : d.g1.W); rows = which == 2 ? j.C : j.Cl; }
        else { const pfm_epic_layer& ly = d.layer[k]; W = which == 0 ? ly.gl1.W : (which == 1 ? ly.lc1.We : ly.lc2.We); rows = which == 0 ? j.C : j.Cl; }
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s = fmaf(blob[W + km16(j.T + r, o)], cj[r], s);
        out[which * 128 + o] = s;
    } else if (o < 16) {  // fc_global2 / fc_g2 (KP16), C rows
        const int64_t W = stem ? d.g2.W : d.layer[k].gl2.W;
        float s = 0.f;
        for (int r = 0; r < j.C; ++r) {
            const int kr = j.T + r;
            s = fmaf(blob[W + (kr >> 4) * 256 + (kr & 15) * 16 + o], cj[r], s);
        }
        out[(stem ? TB_SG2 : TB_G2) + o] = s;
    } else if (o < 32 && stem) {  // fc_l3 extras (KMAJOR [T + Cl][F]), Cl rows
        const int f = o - 16;
        float s = 0.f;
        if (f < j.F)
            for (int r = 0; r < j.Cl; ++r) s = fmaf(blob[d.l3_We + (j.T + r) * j.F + f], cj[r], s);
        out[TB_SB3 + f] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Launch order of the jets of a sampling call: descending multiplicity (ties by index).  A jet's run time grows with its
// number of valid particles (masked tail tiles are skipped) and workgroups are dispatched in blockIdx order as CUs free up, so
// "longest first" is the classic LPT schedule: with more jets than CUs, or several launches in flight, the short jets fill the
// tail instead of a long jet starting last.  order[rank] = jet; every jet still writes its own rows: results do not change.
// One workgroup; B <= ORDER_MAX_JETS (larger batches keep the identity order).
// ------------------------------------------------------------------------------------------------
constexpr int ORDER_MAX_JETS = 8192;
// (a) valid particles per jet, one WAVE per jet over a grid of workgroups (coalesced reads; the one-thread-per-jet walk of a single
// workgroup this replaces took 134 us at 1024 jets x 150 particles -- twice per training step) -> cnt[jet];
// (b) one workgroup ranks the counts (stable by jet index) in place: every count is in LDS before the first order entry is written.
__global__ __launch_bounds__(1024) void epic_jet_count_kernel(const float* __restrict__ mask, int B, int N, int* __restrict__ cnt) {
    const int lane = threadIdx.x & 63, jet = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (jet >= B) return;
    int c = 0;
    for (int r = lane; r < N; r += 64) c += mask[(int64_t)jet * N + r] != 0.f;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    if (lane == 0) cnt[jet] = c;
}
__global__ __launch_bounds__(1024) void epic_jet_order_kernel(int B, int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int cnt[ORDER_MAX_JETS + 4];
    const int tid = threadIdx.x;
    for (int jet = tid; jet < B; jet += 1024) cnt[jet] = order[jet];  // (the counts, written there by epic_jet_count_kernel)
    if (tid < 4) cnt[B + tid] = -1;                                   // pad to a multiple of 4: never larger than a real count
    __syncthreads();
    const int B4 = (B + 3) >> 2;
    for (int jet = tid; jet < B; jet += 1024) {
        const int c = cnt[jet];
        // rank = jets with a larger count + jets with the same count and a smaller index; four counts per (broadcast) LDS read, two
        // accumulators (the scalar walk of rounds 2-3 took 42 us at 1024 jets)
        int r0 = 0, r1 = 0;
        for (int k4 = 0; k4 < B4; ++k4) {
            const int4 v = *reinterpret_cast<const int4*>(cnt + 4 * k4);
            const int k = 4 * k4;
            r0 += (v.x > c) || (v.x == c && k < jet);
            r1 += (v.y > c) || (v.y == c && k + 1 < jet);
            r0 += (v.z > c) || (v.z == c && k + 2 < jet);
            r1 += (v.w > c) || (v.w == c && k + 3 < jet);
        }
        order[r0 + r1] = jet;
    }
}

// ------------------------------------------------------------------------------------------------
// Workgroup list of a packed sampling call: jets in descending row count (rows = last valid particle + 1, what the kernel
// computes), the k-th longest paired with the shortest remaining jet if both fit the LDS tile together
// (pad16(rows A) + rows B <= seg2_rows(N)), else alone.  pack[0] = number of workgroups, pack[1 + 2k], pack[2 + 2k] = jets of
// workgroup k (second = -1: alone).  One workgroup; the pairing walk is sequential (B steps of one thread: ~50 us at B = 1024).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void epic_jet_pack_kernel(const float* __restrict__ mask, int B, int N, int pair_ok,
                                                             int* __restrict__ pack) {
    // 56 KB of static LDS: rows per jet (<= 160: a byte), jets in descending order / tiles per workgroup, the workgroup list
    __shared__ unsigned char cnt[ORDER_MAX_JETS];
    __shared__ unsigned short sorted[ORDER_MAX_JETS];
    __shared__ unsigned short pa[ORDER_MAX_JETS], pb[ORDER_MAX_JETS];  // pb = 0xFFFF: alone
    __shared__ int nwg_s;
    const int tid = threadIdx.x;
    for (int jet = tid >> 6; jet < B; jet += 16) {  // one wave per jet: coalesced reads (a thread per jet walked N strided floats)
        int last = -1;
        for (int r = tid & 63; r < N; r += 64)
            if (mask[(int64_t)jet * N + r] != 0.f) last = r;
        for (int m = 32; m >= 1; m >>= 1) last = max(last, __shfl_xor(last, m));
        if ((tid & 63) == 0) cnt[jet] = (unsigned char)(last >= 0 ? last + 1 : N);  // no valid particle: every row is computed (NaN like the reference)
    }
    __syncthreads();
    for (int jet = tid; jet < B; jet += 1024) {
        const int c = cnt[jet];
        int rank = 0;
        for (int k = 0; k < B; ++k) rank += (cnt[k] > c) || (cnt[k] == c && k < jet);
        sorted[rank] = (unsigned short)jet;
    }
    __syncthreads();
    if (tid == 0) {
        const int cap = seg2_rows(N);
        int i = 0, jj = B - 1, nwg = 0;
        while (i <= jj) {
            const int a = sorted[i++];
            int b = 0xFFFF;
            if (pair_ok && i <= jj) {
                const int cand = sorted[jj];
                if ((cnt[a] + TILE - 1) / TILE * TILE + cnt[cand] <= cap) { b = cand; --jj; }
            }
            pa[nwg] = (unsigned short)a;
            pb[nwg] = (unsigned short)b;
            ++nwg;
        }
        pack[0] = nwg;
        nwg_s = nwg;
    }
    // Workgroups are dispatched in list order and a pair's run time goes with the rows of BOTH its jets: the list is ordered by 16-row
    // tiles of the whole workgroup, most first, ties in pairing order (every thread ranks its workgroups against all others, like the jets
    // above), so that a 9-tile pair does not start behind the 6-tile singles (round 4).  `sorted` is free again: it holds the tile counts.
    __syncthreads();
    const int nwg = nwg_s;
    for (int k = tid; k < nwg; k += 1024) {
        const int a = pa[k], b = pb[k];
        sorted[k] = (unsigned short)((cnt[a] + TILE - 1) / TILE + (b != 0xFFFF ? (cnt[b] + TILE - 1) / TILE : 0));
    }
    __syncthreads();
    for (int k = tid; k < nwg; k += 1024) {
        const int t = sorted[k];
        int rank = 0;
        for (int m = 0; m < nwg; ++m) rank += (sorted[m] > t) || (sorted[m] == t && m < k);
        pack[1 + 2 * rank] = pa[k];
        pack[2 + 2 * rank] = pb[k] == 0xFFFF ? -1 : (int)pb[k];
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA_A16 blocks of a blob from its MFMA_A blocks (pfm_hip.h): thread (block b, w, kt2, lane) rounds the two float4s of k-tiles
// 2 kt2, 2 kt2 + 1 to bf16 (round-to-nearest-even) and stores them as one 16-byte unit.  grid (2 layers + 2): the particle blocks of
// fc_l2, of every layer's two local Linears, and the head's one-slice panel; 512 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void epic_a16_pack_kernel(float* __restrict__ blob, int64_t desc_off) {
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const int b = blockIdx.x, nl = 2 * d.layers + 1;
    int64_t A, A16;
    int nw = 8;
    if (b == 0) { A = d.l2.A; A16 = d.l2.A16; }
    else if (b < nl) { const pfm_epic_layer& ly = d.layer[(b - 1) >> 1]; A = (b & 1) ? ly.lc1.A : ly.lc2.A; A16 = (b & 1) ? ly.lc1.A16 : ly.lc2.A16; }
    else { A = d.l3_A; A16 = d.l3_A16; nw = 1; }
    for (int u = threadIdx.x; u < nw * 4 * 64; u += NT) {  // u = (w * 4 + kt2) * 64 + lane
        const int lane = u & 63, wk = u >> 6, w = wk >> 2, kt2 = wk & 3;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(blob + A + ((int64_t)((w * 8 + 2 * kt2) * 64 + lane)) * 4);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(blob + A + ((int64_t)((w * 8 + 2 * kt2 + 1) * 64 + lane)) * 4);
        *reinterpret_cast<bf16x8*>(blob + A16 + (int64_t)u * 4) = pack_bf16x8(lo, hi);
    }
}

// CH16 copies (pfm_hip.h) of the per-jet blocks, from their KQ16 / WQ16 copies: grid 2 + 3 layers (fc_g1, fc_g2, then fc_global1,
// fc_global2 and the g rows of fc_local1's extras of every layer); 512 threads.
__global__ __launch_bounds__(NT) void epic_ch16_pack_kernel(float* __restrict__ blob, int64_t desc_off) {
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const int b = blockIdx.x;
    int64_t src, dst;
    int rows, nk, nw;
    bool kq;
    if (b == 0) { src = d.q_g1; dst = d.b_g1; rows = 2 * H; nk = 8; nw = 8; kq = true; }
    else if (b == 1) { src = d.q_g2; dst = d.b_g2; rows = H; nk = 4; nw = 1; kq = false; }
    else {
        const int k = (b - 2) / 3, which = (b - 2) % 3;
        if (which == 0) { src = d.q_gl1[k]; dst = d.b_gl1[k]; rows = 2 * H + 16; nk = 9; nw = 8; kq = true; }
        else if (which == 1) { src = d.q_gl2[k]; dst = d.b_gl2[k]; rows = H; nk = 4; nw = 1; kq = false; }
        else { src = d.q_we1[k]; dst = d.b_we1[k]; rows = 16; nk = 1; nw = 8; kq = true; }
    }
    for (int u = threadIdx.x; u < nw * nk * 64; u += NT) {  // u = (w * nk + kt) * 64 + lane
        const int lane = u & 63, wk = u >> 6, w = wk / nk, kt = wk - w * nk;
        const int o = 16 * w + (lane & 15), k0 = 32 * kt + 8 * (lane >> 4);  // eight consecutive rows: inside one 16-row panel
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (k0 < rows) {
            const float* p = blob + src + (kq ? (int64_t)(k0 >> 4) * 2048 + o * 16 + (k0 & 15) : (int64_t)(k0 >> 4) * 256 + o * 16 + (k0 & 15));
            lo = *reinterpret_cast<const f32x4*>(p);
            hi = *reinterpret_cast<const f32x4*>(p + 4);
        }
        *reinterpret_cast<bf16x8*>(blob + dst + (int64_t)u * 4) = pack_bf16x8(lo, hi);
    }
}

// which matrix-pipe flavour the inference kernels use (descriptor flags): 0 fp32, 1 bf16 operands, 2 split fp16
int mfma_mode(const pfm_epic_desc* d) {
    if (!d) return 0;
    if (d->flags & PFM_F_F16X3_MFMA) return 2;
    return (d->flags & PFM_F_BF16_MFMA) ? 1 : 0;
}

template <typename K>
int prepare(K kernel, const pfm_epic_desc* d, int* lds_bytes) {
    int rc = validate(d);
    if (rc) return rc;
    *lds_bytes = make_carve(d->n_points, d->features).total * 4;
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, *lds_bytes),
                     "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

}  // namespace pfm

using namespace pfm;

extern "C" {

int pfm_abi_version(void) { return PFM_ABI_VERSION; }
const char* pfm_last_error(void) { return g_err; }

int pfm_epic_pack_a16(const pfm_epic_desc* d, float* blob, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (!(d->flags & PFM_F_BF16_MFMA)) return 0;
    if (!blob) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->l2.A16 <= 0 || d->l3_A16 <= 0) return set_err(PFM_E_BADARG, "descriptor without MFMA_A16 offsets");
    hipLaunchKernelGGL(epic_a16_pack_kernel, dim3(2 * d->layers + 2), dim3(NT), 0, (hipStream_t)stream, blob, d->blob_floats);
    if (d->b_g1 > 0)
        hipLaunchKernelGGL(epic_ch16_pack_kernel, dim3(2 + 3 * d->layers), dim3(NT), 0, (hipStream_t)stream, blob, d->blob_floats);
    return check_hip(hipGetLastError(), "epic_a16_pack_kernel launch");
}

int pfm_epic_jet_order(const float* mask, int32_t B, int32_t n_points, int32_t* order, void* stream) {
    if (!mask || !order) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (B < 1 || B > ORDER_MAX_JETS) return set_err(PFM_E_BADARG, "1 <= B <= 8192 jets");
    hipLaunchKernelGGL(epic_jet_count_kernel, dim3((B + 15) / 16), dim3(1024), 0, (hipStream_t)stream, mask, B, n_points, order);
    hipLaunchKernelGGL(epic_jet_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, order);
    return check_hip(hipGetLastError(), "epic_jet_order_kernel launch");
}

int64_t pfm_epic_lds_bytes(const pfm_epic_desc* d) {
    if (!d) return -1;
    return (int64_t)make_carve(d->n_points, d->features).total * 4;
}

int64_t pfm_epic_saved_floats_per_jet(const pfm_epic_desc* d) {
    if (!d) return -1;
    return make_saved(d->n_points, d->features, d->layers).total;
}

int pfm_epic_forward(const pfm_epic_desc* d, const float* blob, const float* t, const float* x,
                     const float* cond, const float* mask, float* v, int32_t B, void* stream) {
    int lds = 0;
    const int mode = mfma_mode(d);
    int rc = mode == 2 ? prepare(epic_forward_kernel<2>, d, &lds)
                       : (mode == 1 ? prepare(epic_forward_kernel<1>, d, &lds) : prepare(epic_forward_kernel<0>, d, &lds));
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!blob || !t || !x || !v) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
#define PFM_LAUNCH_FWD(M)                                                                                             \
    hipLaunchKernelGGL(epic_forward_kernel<M>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, t, \
                       (const float*)nullptr, x, cond, mask, v)
    if (mode == 2) PFM_LAUNCH_FWD(2); else if (mode == 1) PFM_LAUNCH_FWD(1); else PFM_LAUNCH_FWD(0);
#undef PFM_LAUNCH_FWD
    return check_hip(hipGetLastError(), "epic_forward_kernel launch");
}

int pfm_epic_forward_temb(const pfm_epic_desc* d, const float* blob, const float* temb, const float* x,
                          const float* cond, const float* mask, float* v, int32_t B, void* stream) {
    int lds = 0;
    const int mode = mfma_mode(d);
    int rc = mode == 2 ? prepare(epic_forward_kernel<2>, d, &lds)
                       : (mode == 1 ? prepare(epic_forward_kernel<1>, d, &lds) : prepare(epic_forward_kernel<0>, d, &lds));
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!blob || !temb || !x || !v) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
#define PFM_LAUNCH_FWD(M)                                                                                          \
    hipLaunchKernelGGL(epic_forward_kernel<M>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, \
                       (const float*)nullptr, temb, x, cond, mask, v)
    if (mode == 2) PFM_LAUNCH_FWD(2); else if (mode == 1) PFM_LAUNCH_FWD(1); else PFM_LAUNCH_FWD(0);
#undef PFM_LAUNCH_FWD
    return check_hip(hipGetLastError(), "epic_forward_kernel launch");
}

// scratch of a sampling call: time-term table [2 n_intervals][layers + 1][TB_SLOT] | cond table [B][layers + 1][TB_SLOT] (conditioned
// models) | workgroup list [1 + 2 B] (int32)
static int64_t cond_table_floats(const pfm_epic_desc* d, int B) { return d->cond_global > 0 ? (int64_t)B * (d->layers + 1) * TB_SLOT : 0; }
int64_t pfm_epic_sample_scratch_floats(const pfm_epic_desc* d, int32_t n_intervals, int32_t B) {
    if (!d || n_intervals < 0 || B < 0) return -1;
    return (int64_t)2 * n_intervals * (d->layers + 1) * TB_SLOT + cond_table_floats(d, B) + ((2 * (int64_t)B + 1 + 63) & ~63);
}

static bool sample_fast(const pfm_epic_desc* d, int mode) { return d && d->layers > 0 && fast_path_ok(*d) && mode != 2; }
int pfm_epic_sample_is_fast(const pfm_epic_desc* d) { return validate(d) == 0 && sample_fast(d, mfma_mode(d)) ? 1 : 0; }

// Two jets per workgroup: opt-in (PFM_F_PACK_JETS; the diagnostic tests/diag/pack_time.py also switches it with the environment
// variable PFM_PACK=1 / 0).  Measured on MI355X: a pair saves one jet's fixed cost but pays for the doubled per-jet work,
// and the workgroups of a launch become few and uniformly long -- +1 % at 1024 jets, +-0 at 256 with two launches in flight, -4 % at
// 512: not on by default.  ONE place decides it, so that the workgroup list and the kernel instantiation can never disagree.
static bool pairs_wanted(const pfm_epic_desc* d) {
    static const char* env = getenv("PFM_PACK");
    return env ? env[0] == '1' : (d->flags & PFM_F_PACK_JETS) != 0;
}

// device pointer of the workgroup list inside `scratch`, after queueing its computation; nullptr: one jet per workgroup, in order.
// allow_pairs: the kernel that will read the list has a two-jet path (fp32 / bf16 kernels with tail-skipping on: a jet then occupies
// rows up to its last valid particle only); false: singles in descending multiplicity, whatever the flag / environment says.
static const int* queue_jet_pack(const pfm_epic_desc* d, float* scratch, int64_t table_floats, const float* mask, int B, int mode,
                                 hipStream_t s, bool allow_pairs) {
    if (!scratch || !mask || B < 2 || B > ORDER_MAX_JETS) return nullptr;
    int* pack = reinterpret_cast<int*>(scratch + table_floats);
    const int pair_ok = allow_pairs && pairs_wanted(d) && (mode != 2) && (d->flags & PFM_F_SKIP_MASKED_TAIL) && seg2_rows(d->n_points) >= 2 * TILE;
    hipLaunchKernelGGL(epic_jet_pack_kernel, dim3(1), dim3(1024), 0, s, mask, B, d->n_points, pair_ok, pack);
    return pack;
}

// device pointer of the jet order inside `scratch`, after queueing its computation; nullptr: identity order
static const int* queue_jet_order(const pfm_epic_desc* d, float* scratch, int64_t table_floats, const float* mask, int B, hipStream_t s) {
    if (!scratch || !mask || B < 2 || B > ORDER_MAX_JETS) return nullptr;
    int* order = reinterpret_cast<int*>(scratch + table_floats);
    hipLaunchKernelGGL(epic_jet_count_kernel, dim3((B + 15) / 16), dim3(1024), 0, s, mask, B, d->n_points, order);
    hipLaunchKernelGGL(epic_jet_order_kernel, dim3(1), dim3(1024), 0, s, B, order);
    return order;
}

static int sample_midpoint(const pfm_epic_desc* d, const float* blob, const float* t_eval, const float* dt,
                           int32_t n_intervals, const float* z, const float* cond, const float* mask,
                           float* x_out, int32_t B, float* scratch, void* stream, const float* temb_tab) {
    int lds = 0;
    const int mode = mfma_mode(d);
    // the time-term table needs whole 16-row time panels and at least one EPiC layer
    const bool tb = scratch && d && d->layers > 0 && d->t_dim % 16 == 0 && d->t_dim > 0;
    int rc;
#define PFM_PREP(M, T) prepare(epic_sample_midpoint_kernel<M, T>, d, &lds)
    if (tb) rc = mode == 2 ? PFM_PREP(2, true) : (mode == 1 ? PFM_PREP(1, true) : PFM_PREP(0, true));
    else rc = mode == 2 ? PFM_PREP(2, false) : (mode == 1 ? PFM_PREP(1, false) : PFM_PREP(0, false));
#undef PFM_PREP
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_intervals < 0) return set_err(PFM_E_BADARG, "n_intervals < 0");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    const int64_t table_floats = (int64_t)2 * n_intervals * (d->layers + 1) * TB_SLOT;
    const int64_t list_off = table_floats + cond_table_floats(d, B);  // the workgroup list sits behind both tables
    // the lean evaluation of epic_fast.h: unconditioned jets, T = 32, F <= 4, fp32 / bf16 operands, one jet per workgroup
    const bool fast = tb && sample_fast(d, mode);
    // four jets per workgroup in fixed 32-row slots (epic_fast.h, quad mode): the caller's descriptor says the sets fit them
    const bool quad = fast && mask && quad_path_ok(*d) && pairs_wanted(d);
    if (quad) {
        const void* qk = mode == 1 ? (const void*)epic_sample_midpoint_quad_kernel<1> : (const void*)epic_sample_midpoint_quad_kernel<0>;
        lds = quad_lds_floats(d->features) * 4;
        if ((rc = check_hip(hipFuncSetAttribute(qk, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)")))
            return rc;
    } else if (fast) {
        const bool cnd = d->cond_global > 0;  // conditioned jets: one per workgroup (the PFM_PACK override must not pair them either)
        const bool pairs = !cnd && pairs_wanted(d);
        const void* fk = cnd ? (mode == 1 ? (const void*)epic_sample_midpoint_fast_kernel<1, false, true> : (const void*)epic_sample_midpoint_fast_kernel<0, false, true>)
                       : pairs ? (mode == 1 ? (const void*)epic_sample_midpoint_fast_kernel<1, true> : (const void*)epic_sample_midpoint_fast_kernel<0, true>)
                               : (mode == 1 ? (const void*)epic_sample_midpoint_fast_kernel<1, false> : (const void*)epic_sample_midpoint_fast_kernel<0, false>);
        if ((rc = validate(d))) return rc;
        lds = (make_carve(d->n_points, d->features).total + TBL_FLOATS) * 4;  // + the chain's table rows behind the carve (fast_path_ok: it fits)
        if ((rc = check_hip(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)")))
            return rc;
    }
    if (tb && n_intervals > 0) {
        hipLaunchKernelGGL(epic_time_table_kernel, dim3(2 * n_intervals, d->layers + 1), dim3(NT), 0, (hipStream_t)stream, blob,
                           d->blob_floats, t_eval, scratch, temb_tab, fast ? 1 : 0);
        if ((rc = check_hip(hipGetLastError(), "epic_time_table_kernel launch"))) return rc;
    }
    if (quad) {
        if (mode == 1)
            hipLaunchKernelGGL(epic_sample_midpoint_quad_kernel<1>, dim3((B + QJETS - 1) / QJETS), dim3(NT), lds, (hipStream_t)stream, blob,
                               d->blob_floats, dt, n_intervals, z, mask, x_out, (const float*)scratch, B);
        else
            hipLaunchKernelGGL(epic_sample_midpoint_quad_kernel<0>, dim3((B + QJETS - 1) / QJETS), dim3(NT), lds, (hipStream_t)stream, blob,
                               d->blob_floats, dt, n_intervals, z, mask, x_out, (const float*)scratch, B);
        return check_hip(hipGetLastError(), "epic_sample_midpoint_quad_kernel launch");
    }
    if (fast) {
        const bool cnd = d->cond_global > 0;
        const bool pairs = !cnd && pairs_wanted(d);
        const int* jet_order = queue_jet_pack(d, scratch, list_off, mask, B, mode, (hipStream_t)stream, pairs);  // singles unless `pairs`
        float* ctab = nullptr;
        if (cnd) {  // the jets' conditioning terms, once per call
            ctab = scratch + table_floats;
            hipLaunchKernelGGL(epic_cond_table_kernel, dim3(B, d->layers + 1), dim3(NT), 0, (hipStream_t)stream, blob, d->blob_floats, cond, ctab);
            if ((rc = check_hip(hipGetLastError(), "epic_cond_table_kernel launch"))) return rc;
        }
#define PFM_LAUNCH_FAST(M, P, C)                                                                                                          \
    hipLaunchKernelGGL((epic_sample_midpoint_fast_kernel<M, P, C>), dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, dt, \
                       n_intervals, z, mask, x_out, (const float*)scratch, jet_order, (const float*)ctab)
        if (cnd) { if (mode == 1) PFM_LAUNCH_FAST(1, false, true); else PFM_LAUNCH_FAST(0, false, true); }
        else if (pairs) { if (mode == 1) PFM_LAUNCH_FAST(1, true, false); else PFM_LAUNCH_FAST(0, true, false); }
        else { if (mode == 1) PFM_LAUNCH_FAST(1, false, false); else PFM_LAUNCH_FAST(0, false, false); }
#undef PFM_LAUNCH_FAST
        return check_hip(hipGetLastError(), "epic_sample_midpoint_fast_kernel launch");
    }
#define PFM_LAUNCH_SMP(M, T)                                                                                                  \
    hipLaunchKernelGGL((epic_sample_midpoint_kernel<M, T>), dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, \
                       t_eval, dt, n_intervals, z, cond, mask, x_out, (const float*)scratch, order, temb_tab)
    const int* order = queue_jet_pack(d, scratch, list_off, mask, B, mode, (hipStream_t)stream, true);  // the generic kernel pairs conditioned jets too
    if (tb) { if (mode == 2) PFM_LAUNCH_SMP(2, true); else if (mode == 1) PFM_LAUNCH_SMP(1, true); else PFM_LAUNCH_SMP(0, true); }
    else { if (mode == 2) PFM_LAUNCH_SMP(2, false); else if (mode == 1) PFM_LAUNCH_SMP(1, false); else PFM_LAUNCH_SMP(0, false); }
#undef PFM_LAUNCH_SMP
    return check_hip(hipGetLastError(), "epic_sample_midpoint_kernel launch");
}

int pfm_epic_sample_midpoint(const pfm_epic_desc* d, const float* blob, const float* t_eval, const float* dt,
                             int32_t n_intervals, const float* z, const float* cond, const float* mask,
                             float* x_out, int32_t B, float* scratch, void* stream) {
    return sample_midpoint(d, blob, t_eval, dt, n_intervals, z, cond, mask, x_out, B, scratch, stream, nullptr);
}

int pfm_epic_sample_midpoint_temb(const pfm_epic_desc* d, const float* blob, const float* temb_tab, const float* dt,
                                  int32_t n_intervals, const float* z, const float* cond, const float* mask,
                                  float* x_out, int32_t B, float* scratch, void* stream) {
    if (!temb_tab) return set_err(PFM_E_BADARG, "temb_tab is NULL");
    return sample_midpoint(d, blob, temb_tab /* stands in for t_eval: never read */, dt, n_intervals, z, cond, mask, x_out, B, scratch,
                           stream, temb_tab);
}

// kbuf: [stage slopes B * stages * N * F | jet order, B rounded up to 64 | fast-format time table of all stage times]
static int64_t rk_order_off(const pfm_epic_desc* d, int stages, int B) { return (int64_t)B * stages * d->n_points * d->features; }
static int64_t rk_table_off(const pfm_epic_desc* d, int stages, int B) { return rk_order_off(d, stages, B) + (((int64_t)B + 63) & ~(int64_t)63); }

static int sample_rk(const pfm_epic_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                     int32_t n_intervals, const float* z, const float* cond, const float* mask, float* x_out, int32_t B,
                     float* kbuf, int64_t kbuf_floats, const float* rhs, void* stream, const float* temb_tab) {
    int lds = 0;
    const int mode = mfma_mode(d);
    int rc = mode == 2 ? prepare(epic_sample_rk_kernel<2>, d, &lds)
                       : (mode == 1 ? prepare(epic_sample_rk_kernel<1>, d, &lds) : prepare(epic_sample_rk_kernel<0>, d, &lds));
    if (rc) return rc;
    if (!tab || tab->stages < 1 || tab->stages > PFM_RK_MAX_STAGES) return set_err(PFM_E_BADARG, "tableau.stages must be in 1..4");
    if (B <= 0) return 0;
    if (!blob || !t_eval || !dt || !z || !x_out || !kbuf) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_intervals < 0) return set_err(PFM_E_BADARG, "n_intervals < 0");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    const int64_t order_off = rk_order_off(d, tab->stages, B);
    const int* order = queue_jet_order(d, kbuf, order_off, mask, B, (hipStream_t)stream);
    // the lean evaluation (epic_fast.h) when the caller's scratch has room for the time table of every stage time
    const int n_evals = tab->stages * n_intervals;
    const int64_t table_off = rk_table_off(d, tab->stages, B), table_floats = (int64_t)n_evals * (d->layers + 1) * TB_SLOT;
    if (sample_fast(d, mode) && d->cond_global == 0 && !(d->flags & PFM_F_PACK_JETS) && kbuf_floats >= table_off + table_floats && n_evals > 0) {
        float* table = kbuf + table_off;
        hipLaunchKernelGGL(epic_time_table_kernel, dim3(n_evals, d->layers + 1), dim3(NT), 0, (hipStream_t)stream, blob, d->blob_floats,
                           t_eval, table, temb_tab, 1);
        if ((rc = check_hip(hipGetLastError(), "epic_time_table_kernel launch"))) return rc;
        const void* fk = mode == 1 ? (const void*)epic_sample_rk_fast_kernel<1> : (const void*)epic_sample_rk_fast_kernel<0>;
        lds = (make_carve(d->n_points, d->features).total + TBL_FLOATS) * 4;
        if ((rc = check_hip(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)")))
            return rc;
        if (mode == 1)
            hipLaunchKernelGGL(epic_sample_rk_fast_kernel<1>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, *tab, dt,
                               n_intervals, z, mask, x_out, kbuf, rhs, order, (const float*)table);
        else
            hipLaunchKernelGGL(epic_sample_rk_fast_kernel<0>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, *tab, dt,
                               n_intervals, z, mask, x_out, kbuf, rhs, order, (const float*)table);
        return check_hip(hipGetLastError(), "epic_sample_rk_fast_kernel launch");
    }
#define PFM_LAUNCH_RK(M)                                                                                                    \
    hipLaunchKernelGGL(epic_sample_rk_kernel<M>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, *tab, \
                       t_eval, dt, n_intervals, z, cond, mask, x_out, kbuf, rhs, order, temb_tab)
    if (mode == 2) PFM_LAUNCH_RK(2); else if (mode == 1) PFM_LAUNCH_RK(1); else PFM_LAUNCH_RK(0);
#undef PFM_LAUNCH_RK
    return check_hip(hipGetLastError(), "epic_sample_rk_kernel launch");
}

int64_t pfm_epic_sample_rk_scratch_floats(const pfm_epic_desc* d, int32_t stages, int32_t n_intervals, int32_t B) {
    if (!d || stages < 1 || stages > PFM_RK_MAX_STAGES || n_intervals < 0 || B < 0) return -1;
    return rk_table_off(d, stages, B) + (int64_t)stages * n_intervals * (d->layers + 1) * TB_SLOT;
}

int pfm_epic_sample_rk_sized(const pfm_epic_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                             int32_t n_intervals, const float* z, const float* cond, const float* mask, float* x_out, int32_t B,
                             float* kbuf, int64_t kbuf_floats, const float* rhs, void* stream) {
    return sample_rk(d, blob, tab, t_eval, dt, n_intervals, z, cond, mask, x_out, B, kbuf, kbuf_floats, rhs, stream, nullptr);
}

int pfm_epic_sample_rk(const pfm_epic_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* t_eval, const float* dt,
                       int32_t n_intervals, const float* z, const float* cond, const float* mask, float* x_out, int32_t B,
                       float* kbuf, const float* rhs, void* stream) {
    return sample_rk(d, blob, tab, t_eval, dt, n_intervals, z, cond, mask, x_out, B, kbuf, 0, rhs, stream, nullptr);
}

int pfm_epic_sample_rk_temb(const pfm_epic_desc* d, const float* blob, const pfm_rk_tableau* tab, const float* temb_tab, const float* dt,
                            int32_t n_intervals, const float* z, const float* cond, const float* mask, float* x_out, int32_t B,
                            float* kbuf, void* stream) {
    if (!temb_tab) return set_err(PFM_E_BADARG, "temb_tab is NULL");
    return sample_rk(d, blob, tab, temb_tab, dt, n_intervals, z, cond, mask, x_out, B, kbuf, 0, nullptr, stream, temb_tab);
}

}  // extern "C"
