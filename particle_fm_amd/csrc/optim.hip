// Optimiser tail on flat fp32 buffers: global-norm clip + AdamW + EMA in two launches.
//
// Replaces, per training step, torch.nn.utils.clip_grad_norm_(params, 0.5)
// (configs/experiment/jetnet/fm_tops150.yaml:24), torch.optim.AdamW(lr=1e-3, weight_decay=5e-5)
// (configs/model/flow_matching.yaml:3-7) and EMA.apply_ema (particle_fm/callbacks/ema.py:73-81:
// ema -= (ema - w) * (1 - decay)), which in the reference are ~90 parameter tensors x several
// tiny launches.  HBM-bound: 561 330 parameters x (read p,g,m,v,ema + write p,m,v,ema) = 20 MB.
#include <hip/hip_runtime.h>

#include "pfm_common.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

// ||g * mul||^2 in two deterministic stages (no atomics: every rank of a data-parallel job must derive the SAME clip factor
// from the same all-reduced gradient, or the replicas drift apart by an ulp per step): block b writes its partial sum to
// part[b]; the optimiser kernel adds the partials in a fixed order.
constexpr int SUMSQ_MAX_BLOCKS = 1023;  // partials live in scratch[1 .. 1024), scratch[0] receives the total

__device__ __forceinline__ float block_sum256(float acc) {
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float mul,
                                                    float* __restrict__ part) {
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = g4[i] * mul;
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x] * mul;
        acc += v * v;
    }
    const float tot = block_sum256(acc);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

struct AdamArgs {
    float grad_mul, max_norm, lr, beta1, beta2, eps, weight_decay, ema_decay, bc1, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float& e, bool has_ema,
                                         const AdamArgs& a, float clip) {
    g = g * a.grad_mul * clip;
    p = p * (1.0f - a.lr * a.weight_decay);           // decoupled weight decay (torch AdamW)
    m = m + (g - m) * (1.0f - a.beta1);               // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - (a.lr / a.bc1) * (m / denom);
    if (has_ema) e = e - (e - p) * (1.0f - a.ema_decay);  // ema.py:78-81
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, float* __restrict__ scratch,
                                                        int nparts, int64_t n, AdamArgs a) {
    float clip = 1.0f;
    if (a.max_norm > 0.f) {
        // every block adds the same partials in the same order: one value for the whole grid, bit-identical on every rank
        float acc = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) acc += scratch[1 + i];
        const float sumsq = block_sum256(acc);
        if (blockIdx.x == 0 && threadIdx.x == 0) scratch[0] = sumsq;  // read back by the host side (grad_norm)
        clip = fminf(1.0f, a.max_norm / (sqrtf(sumsq) + 1e-6f));  // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
    }
    const bool has_ema = ema != nullptr;
    const int64_t n4 = n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    f32x4* e4 = reinterpret_cast<f32x4*>(ema);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        f32x4 ee = has_ema ? e4[i] : pp;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ps = pp[c], ms = mm[c], vs = vv[c], es = ee[c];
            adam_one(ps, gg[c], ms, vs, es, has_ema, a, clip);
            pp[c] = ps; mm[c] = ms; vv[c] = vs; ee[c] = es;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (has_ema) e4[i] = ee;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i], ee = has_ema ? ema[i] : 0.f;
        adam_one(pp, g[i], mm, vv, ee, has_ema, a, clip);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (has_ema) ema[i] = ee;
    }
}
}  // namespace pfm

using namespace pfm;

extern "C" int pfm_optim_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                              float* scratch, int64_t n, float grad_mul, float max_norm, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float ema_decay, int32_t step,
                              void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !scratch) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n <= 0) return 0;
    if (step < 1) return set_err(PFM_E_BADARG, "step is 1-based");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema)) & 15)
        return set_err(PFM_E_BADARG, "flat buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (!(max_norm > 0.f)) {  // with clipping the optimiser kernel itself writes scratch[0] = ||g||^2; without, it reads as 0
        int rc = check_hip(hipMemsetAsync(scratch, 0, sizeof(float), s), "hipMemsetAsync(scratch)");
        if (rc) return rc;
    }
    const int64_t n4 = (n + 3) / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    const int nparts = blocks < SUMSQ_MAX_BLOCKS ? blocks : SUMSQ_MAX_BLOCKS;
    if (max_norm > 0.f) hipLaunchKernelGGL(sumsq_kernel, dim3(nparts), dim3(256), 0, s, grad, n, grad_mul, scratch + 1);
    AdamArgs a;
    a.grad_mul = grad_mul; a.max_norm = max_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.weight_decay = weight_decay; a.ema_decay = ema_decay;
    a.bc1 = 1.0f - powf(beta1, (float)step);
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, ema,
                       scratch, nparts, n, a);
    return check_hip(hipGetLastError(), "pfm_optim_step launch");
}
