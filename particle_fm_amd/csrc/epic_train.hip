// gfx950 kernels + C ABI: flow-matching loss forward (with saved activations) and backward.
//
// Reference: particle_fm/models/components/losses.py:38-77 (FlowMatchingLoss.forward, "FM-OT"),
// :101-136 (ConditionalFlowMatchingLoss.forward, "CFM") and the autograd of the EPiC network
// (epic.py:85-203, 304-391).  The random draws (t per jet, z, eps) are inputs: the host draws them
// exactly as the reference does (t on the CPU generator, z on the device generator).
#include <hip/hip_runtime.h>

#include "epic_bwd_kernel.h"
#include "epic_dw.h"

#ifdef PFM_BDIAG
namespace pfm {
__device__ unsigned long long g_pfm_bstamps[1024];
__device__ int g_pfm_nbstamp;
}
extern "C" int pfm_diag_read_bwd_stamps(unsigned long long* out, int* n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(pfm::g_pfm_nbstamp), sizeof(int));
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pfm::g_pfm_bstamps), sizeof(unsigned long long) * 1024);
    int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pfm::g_pfm_nbstamp), &zero, sizeof(int));
    return 0;
}
#endif

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);
int validate(const pfm_epic_desc* d);
int mfma_mode(const pfm_epic_desc* d);

// ------------------------------------------------------------------------------------------------
// forward: y, u from (x, z, t); v = f(t, y); loss_parts[jet] = sum (v-u)^2; activations -> saved
// ------------------------------------------------------------------------------------------------
// MODE 0: fp32 MFMA; 1: bf16 MFMA operands in the particle Linears (PFM_F_BF16_MFMA; fp32 accumulate, fp32 activations and saves)
template <int MODE>
__global__ __launch_bounds__(NT, 2) void epic_fm_loss_forward_kernel(
    const float* __restrict__ blob, int64_t desc_off, int kind, float sigma, const float* __restrict__ t,
    const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ eps,
    const float* __restrict__ cond, const float* __restrict__ mask, float* __restrict__ saved,
    float* __restrict__ loss_parts, float* __restrict__ mask_count, int crit, const float* __restrict__ rates,
    const float* __restrict__ temb,   // temb (or NULL): [B][T] time embedding supplied by the caller (t_emb="gaussian")
    const int* __restrict__ order) {  // order (or NULL): launch order of the jets, longest first (pfm_epic_jet_order)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const pfm_epic_desc& d = *reinterpret_cast<const pfm_epic_desc*>(blob + desc_off);
    const JetDims j = dims_of(d);
    const Carve c = make_carve(j.N, j.F);
    const SavedLayout sl = make_saved(j.N, j.F, j.layers);
    const int jet = order ? order[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
    const int n_rows = epic_jet_setup(d, j, blob, lds, c, cond ? cond + (size_t)jet * j.C : nullptr,
                                      mask ? mask + (size_t)jet * j.N : nullptr);
    float* sv = saved + (size_t)jet * sl.total;
    const size_t base = (size_t)jet * j.N * j.F;
    const float tj = t[jet];
    const float one_m_sigma = (float)(1.0 - (double)sigma);  // python computes (1 - sigma) in double
    for (int i = tid; i < j.N * j.F; i += NT) {
        const float xv = x[base + i], zv = z[base + i], m = lds[c.maskf + i / j.F];
        float y, u;
        if (kind == 0) {
            // losses.py:56  y = (1 - t) * x + (sigma + (1 - sigma) * t) * z ; :61-62 u = ((1 - sigma) * z - x) * mask
            const float a = __fmul_rn(__fsub_rn(1.0f, tj), xv);
            const float b = __fmul_rn(__fadd_rn(sigma, __fmul_rn(one_m_sigma, tj)), zv);
            y = __fadd_rn(a, b);
            u = __fmul_rn(__fsub_rn(__fmul_rn(one_m_sigma, zv), xv), m);
        } else if (kind == 2) {
            // DroidLoss, losses.py:332-336:  y = x + t * z ; u = z * mask
            y = __fadd_rn(xv, __fmul_rn(tj, zv));
            u = __fmul_rn(zv, m);
        } else if (kind == 3) {
            // DiffusionLoss, losses.py:260, 272:  noisy = signal_rate * x + noise_rate * z ; target = z (z arrives masked, :244)
            y = __fadd_rn(__fmul_rn(rates[2 * jet], xv), __fmul_rn(rates[2 * jet + 1], zv));
            u = zv;
        } else {
            // losses.py:115-119  mu = (1 - t) * x + t * x0 ; y = mu + sigma * eps ; u = (x0 - x) * mask
            const float mu = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, tj), xv), __fmul_rn(tj, zv));
            y = __fadd_rn(mu, __fmul_rn(sigma, eps[base + i]));
            u = __fmul_rn(__fsub_rn(zv, xv), m);
        }
        lds[c.yin + i] = y;
        sv[sl.y + i] = y;
        sv[sl.u + i] = u;
    }
    if (temb) {
        if (tid < j.T) lds[c.vin + tid] = lds[c.vin2 + tid] = temb[(size_t)jet * j.T + tid];
    } else {
        epic_time_embedding(d, j, blob, lds, c, tj);
    }
    __syncthreads();
    if (tid < j.T) sv[sl.temb + tid] = lds[c.vin + tid];
    epic_body<true, MODE>(d, j, blob, lds, c, n_rows, sv, sl);
    float sq = 0.f;
    const int F = j.F;
    float* svv = sv + sl.v;
    const float* svu = sv + sl.u;
    epic_head(d, j, blob, lds, c, n_rows, [&](int p, int f, float val) {
        svv[p * F + f] = val;
        const float dlt = val - svu[p * F + f];
        // crit 1 = nn.HuberLoss(delta = 1): d^2 / 2 inside the knee, |d| - 1/2 outside
        sq += crit ? (fabsf(dlt) < 1.0f ? 0.5f * dlt * dlt : fabsf(dlt) - 0.5f) : dlt * dlt;
    });
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    __syncthreads();
    float* red = lds + c.misc + 8;
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < NW; ++i) s += red[i];
        loss_parts[jet] = s;
        mask_count[jet] = lds[c.misc];
    }
}

}  // namespace pfm

using namespace pfm;

static int loss_forward(const pfm_epic_desc* d, const float* blob, int kind, float sigma, const float* t, const float* x,
                        const float* z, const float* eps, const float* cond, const float* mask, float* saved,
                        float* loss_parts, float* mask_count, int crit, const float* rates, int B, const int32_t* order, void* stream,
                        const float* temb = nullptr) {
    int rc = validate(d);
    if (rc) return rc;
    const int lds = make_carve(d->n_points, d->features).total * 4;
    const bool bf16 = mfma_mode(d) == 1;  // the split-fp16 flavour is for the inference kernels only: training stays fp32 there
    rc = check_hip(hipFuncSetAttribute(bf16 ? reinterpret_cast<const void*>(epic_fm_loss_forward_kernel<1>)
                                            : reinterpret_cast<const void*>(epic_fm_loss_forward_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds),
                   "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!blob || !t || !x || !z || !saved || !loss_parts || !mask_count)
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if (kind == 1 && !eps) return set_err(PFM_E_BADARG, "CFM needs eps");
    if (kind == 3 && !rates) return set_err(PFM_E_BADARG, "the diffusion loss needs the signal / noise rates");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    if (bf16)
        hipLaunchKernelGGL(epic_fm_loss_forward_kernel<1>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, kind,
                           sigma, t, x, z, eps, cond, mask, saved, loss_parts, mask_count, crit, rates, temb, order);
    else
        hipLaunchKernelGGL(epic_fm_loss_forward_kernel<0>, dim3(B), dim3(NT), lds, (hipStream_t)stream, blob, d->blob_floats, kind,
                           sigma, t, x, z, eps, cond, mask, saved, loss_parts, mask_count, crit, rates, temb, order);
    return check_hip(hipGetLastError(), "epic_fm_loss_forward_kernel launch");
}

extern "C" int pfm_epic_fm_loss_forward(const pfm_epic_desc* d, const float* blob, int32_t kind, float sigma,
                                        const float* t, const float* x, const float* z, const float* eps,
                                        const float* cond, const float* mask, float* saved, float* loss_parts,
                                        float* mask_count, int32_t B, const int32_t* order, void* stream) {
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    return loss_forward(d, blob, kind, sigma, t, x, z, eps, cond, mask, saved, loss_parts, mask_count, 0, nullptr, B, order, stream);
}

extern "C" int pfm_epic_fm_loss_forward_temb(const pfm_epic_desc* d, const float* blob, int32_t kind, float sigma, const float* t,
                                             const float* temb, const float* x, const float* z, const float* eps,
                                             const float* cond, const float* mask, float* saved, float* loss_parts,
                                             float* mask_count, int32_t B, const int32_t* order, void* stream) {
    if (kind < 0 || kind > 2) return set_err(PFM_E_BADARG, "kind must be 0 (FM-OT), 1 (CFM) or 2 (droid)");
    if (!temb) return set_err(PFM_E_BADARG, "temb is NULL");
    return loss_forward(d, blob, kind, sigma, t, x, z, eps, cond, mask, saved, loss_parts, mask_count, 0, nullptr, B, order, stream, temb);
}

extern "C" int pfm_epic_diffusion_loss_forward(const pfm_epic_desc* d, const float* blob, int32_t criterion, const float* rates,
                                               const float* t, const float* x, const float* z, const float* cond,
                                               const float* mask, float* saved, float* loss_parts, float* mask_count,
                                               int32_t B, const int32_t* order, void* stream) {
    if (criterion < 0 || criterion > 1) return set_err(PFM_E_BADARG, "criterion must be 0 (mse) or 1 (huber)");
    return loss_forward(d, blob, 3, 0.f, t, x, z, nullptr, cond, mask, saved, loss_parts, mask_count, criterion, rates, B, order, stream);
}

// phases: PFM_BWD_PHASE_CHAIN (the per-jet chain kernel + the rank-1 / F-wide parts of the reduction: afterwards every gradient slot of
// the blob is final except the 128x128 particle blocks) | PFM_BWD_PHASE_DW (the dW GEMM + the tile part of the reduction)
static int loss_backward(const pfm_epic_desc* d, const float* blob, const float* cond, const float* mask, const float* saved,
                         const float* inv_mask_total, const float* grad_scale, float* grad_blob, int crit, const float* jet_w,
                         int B, float* scratch, const int32_t* order, void* stream, float* dtemb = nullptr,
                         int phases = PFM_BWD_PHASE_CHAIN | PFM_BWD_PHASE_DW, float* dy = nullptr) {
    int rc = validate(d);
    if (rc) return rc;
    const int64_t lds = (int64_t)make_bcarve(d->n_points, d->features).total * 4;
    if (lds > 163840) return set_err(PFM_E_LDS, "set does not fit the 160 KiB LDS tile of the backward kernel");
    if (d->l2.AT < 0) return set_err(PFM_E_BADARG, "blob was packed without the transposed (backward) weight copies");
    if (B > DW_MAXB) return set_err(PFM_E_BADARG, "at most 8192 jets per backward call (split the batch)");
    const bool bf16 = mfma_mode(d) == 1;  // dX products on bf16 operands like the forward; the dW GEMM keeps fp32 operands
    rc = check_hip(hipFuncSetAttribute(bf16 ? reinterpret_cast<const void*>(epic_fm_loss_backward_kernel<true>)
                                            : reinterpret_cast<const void*>(epic_fm_loss_backward_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                   "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    const int dw_lds = (128 * DW_S + (B > 0 ? B : 0) + 1) * 4;
    rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(epic_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds),
                   "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!blob || !saved || !inv_mask_total || !grad_scale || !grad_blob || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (d->cond_global > 0 && !cond) return set_err(PFM_E_BADARG, "cond_global > 0 but cond is NULL");
    const BwdWork bw = make_bwd_work(d->n_points, d->layers, B);
    hipStream_t s = (hipStream_t)stream;
    RedArgs ra;
    ra.n_tile = bw.nblk * 16;
    ra.panels_per_job = (VIN_FLOATS + 15) / 16 + 1;
    ra.n_r1 = (d->layers + 2) * 4 * ra.panels_per_job;
    ra.n_small = (2 * MAXF * H + 63) / 64;  // 64 outputs per workgroup (epic_bwd_reduce_kernel, part (c))
    const bool whole = (phases & PFM_BWD_PHASE_CHAIN) && (phases & PFM_BWD_PHASE_DW);
    if (phases & PFM_BWD_PHASE_CHAIN) {
        // 1. per-jet chain: gradient rows + rank-1 operands -> scratch
        if (bf16)
            hipLaunchKernelGGL(epic_fm_loss_backward_kernel<true>, dim3(B), dim3(NT), (int)lds, s, blob, d->blob_floats, cond, mask, saved,
                               inv_mask_total, grad_scale, scratch, bw, crit, jet_w, dtemb, order, dy);
        else
            hipLaunchKernelGGL(epic_fm_loss_backward_kernel<false>, dim3(B), dim3(NT), (int)lds, s, blob, d->blob_floats, cond, mask, saved,
                               inv_mask_total, grad_scale, scratch, bw, crit, jet_w, dtemb, order, dy);
        if ((rc = check_hip(hipGetLastError(), "epic_fm_loss_backward_kernel launch"))) return rc;
        if (!whole) {  // the sums that need the chain's records only: rank-1 sums over jets, the F-wide particle blocks
            ra.item0 = ra.n_tile;
            hipLaunchKernelGGL(epic_bwd_reduce_kernel, dim3(ra.n_r1 + ra.n_small), dim3(RED_T * RED_G), 0, s, blob, d->blob_floats,
                               (const float*)scratch, bw, B, ra, grad_blob);
            if ((rc = check_hip(hipGetLastError(), "epic_bwd_reduce_kernel launch (records)"))) return rc;
        }
    }
    if (phases & PFM_BWD_PHASE_DW) {
        // 2. the 2 * layers + 1 dW GEMMs over the rows of all jets, split by row ranges
        hipLaunchKernelGGL(epic_dw_kernel, dim3(bw.nsplit, bw.nblk), dim3(DW_T), dw_lds, s, blob, d->blob_floats, saved, scratch, bw, B);
        if ((rc = check_hip(hipGetLastError(), "epic_dw_kernel launch"))) return rc;
        // 3. fixed-order sums: partial tiles (whole call: + the rank-1 sums and the F-wide blocks, one launch)
        ra.item0 = 0;
        hipLaunchKernelGGL(epic_bwd_reduce_kernel, dim3(whole ? ra.n_tile + ra.n_r1 + ra.n_small : ra.n_tile), dim3(RED_T * RED_G), 0, s,
                           blob, d->blob_floats, (const float*)scratch, bw, B, ra, grad_blob);
        if ((rc = check_hip(hipGetLastError(), "epic_bwd_reduce_kernel launch"))) return rc;
    }
    return 0;
}

// loss = sum_b w_b parts_b / sum_b count_b and 1 / sum count from the per-jet outputs of the loss forward: one workgroup, sums in a
// fixed order (a pure function of its inputs)
namespace pfm {
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ parts, const float* __restrict__ count,
                                                          const float* __restrict__ jet_w, int B, float* __restrict__ out) {
    __shared__ float red[2 * 256];
    const int tid = threadIdx.x;
    float sp = 0.f, sc = 0.f;
    for (int i = tid; i < B; i += 256) {
        sp += parts[i] * (jet_w ? jet_w[i] : 1.0f);
        sc += count[i];
    }
    red[tid] = sp;
    red[256 + tid] = sc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) { red[tid] += red[tid + s]; red[256 + tid] += red[256 + tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = red[0] / red[256];
        out[1] = 1.0f / red[256];
    }
}
}  // namespace pfm

extern "C" int pfm_loss_finish(const float* loss_parts, const float* mask_count, const float* jet_weight, int32_t B, float* out2,
                               void* stream) {
    if (B <= 0) return set_err(PFM_E_BADARG, "B must be positive");
    if (!loss_parts || !mask_count || !out2) return set_err(PFM_E_BADARG, "NULL device pointer");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_parts, mask_count, jet_weight, B, out2);
    return check_hip(hipGetLastError(), "loss_finish_kernel launch");
}

extern "C" int64_t pfm_epic_backward_lds_bytes(const pfm_epic_desc* d) {
    if (!d) return -1;
    return (int64_t)make_bcarve(d->n_points, d->features).total * 4;
}

extern "C" int64_t pfm_epic_backward_scratch_floats(const pfm_epic_desc* d, int32_t B) {
    if (!d || B < 0) return -1;
    return make_bwd_work(d->n_points, d->layers, B).total;
}

extern "C" int pfm_epic_fm_loss_backward(const pfm_epic_desc* d, const float* blob, const float* t, const float* cond,
                                         const float* mask, const float* saved, const float* inv_mask_total,
                                         const float* grad_scale, float* grad_blob, int32_t B, float* scratch, const int32_t* order,
                                         void* stream) {
    (void)t;  // the time embedding is part of `saved`
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, 0, nullptr, B, scratch, order, stream);
}

extern "C" int pfm_epic_fm_loss_backward_phases(const pfm_epic_desc* d, const float* blob, const float* cond, const float* mask,
                                                const float* saved, const float* inv_mask_total, const float* grad_scale,
                                                float* grad_blob, int32_t criterion, const float* jet_weight, int32_t B, float* scratch,
                                                const int32_t* order, int32_t phases, void* stream) {
    if (criterion < 0 || criterion > 1) return set_err(PFM_E_BADARG, "criterion must be 0 (mse) or 1 (huber)");
    if (!(phases & (PFM_BWD_PHASE_CHAIN | PFM_BWD_PHASE_DW)) || (phases & ~(PFM_BWD_PHASE_CHAIN | PFM_BWD_PHASE_DW)))
        return set_err(PFM_E_BADARG, "phases must be PFM_BWD_PHASE_CHAIN, PFM_BWD_PHASE_DW or both");
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, criterion, jet_weight, B, scratch, order, stream,
                         nullptr, phases);
}

extern "C" int pfm_epic_fm_loss_backward_dx(const pfm_epic_desc* d, const float* blob, const float* cond, const float* mask,
                                            const float* saved, const float* inv_mask_total, const float* grad_scale, float* grad_blob,
                                            float* grad_y, int32_t B, float* scratch, const int32_t* order, void* stream) {
    if (!grad_y) return set_err(PFM_E_BADARG, "grad_y is NULL");
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, 0, nullptr, B, scratch, order, stream, nullptr,
                         PFM_BWD_PHASE_CHAIN | PFM_BWD_PHASE_DW, grad_y);
}

extern "C" int pfm_epic_fm_loss_backward_dx_temb(const pfm_epic_desc* d, const float* blob, const float* cond, const float* mask,
                                                 const float* saved, const float* inv_mask_total, const float* grad_scale, float* grad_blob,
                                                 float* grad_y, float* grad_temb, int32_t B, float* scratch, const int32_t* order,
                                                 void* stream) {
    if (!grad_y || !grad_temb) return set_err(PFM_E_BADARG, "grad_y / grad_temb is NULL");
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, 0, nullptr, B, scratch, order, stream, grad_temb,
                         PFM_BWD_PHASE_CHAIN | PFM_BWD_PHASE_DW, grad_y);
}

extern "C" int pfm_epic_fm_loss_backward_temb(const pfm_epic_desc* d, const float* blob, const float* cond, const float* mask,
                                              const float* saved, const float* inv_mask_total, const float* grad_scale,
                                              float* grad_blob, float* grad_temb, int32_t B, float* scratch, const int32_t* order,
                                              void* stream) {
    if (!grad_temb) return set_err(PFM_E_BADARG, "grad_temb is NULL");
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, 0, nullptr, B, scratch, order, stream, grad_temb);
}

extern "C" int pfm_epic_diffusion_loss_backward(const pfm_epic_desc* d, const float* blob, int32_t criterion, const float* jet_weight,
                                                const float* cond, const float* mask, const float* saved,
                                                const float* inv_mask_total, const float* grad_scale, float* grad_blob,
                                                int32_t B, float* scratch, const int32_t* order, void* stream) {
    if (criterion < 0 || criterion > 1) return set_err(PFM_E_BADARG, "criterion must be 0 (mse) or 1 (huber)");
    return loss_backward(d, blob, cond, mask, saved, inv_mask_total, grad_scale, grad_blob, criterion, jet_weight, B, scratch, order, stream);
}
