// gfx950 kernel + C ABI: post-processing of generated jets, on the device.
//
// Reference: particle_fm/utils/data_generation.py:94-123 (generate_data: per batch, after model.sample(...).cpu()):
//   inverse_normalize_tensor  x[..., i] = x[..., i] * (std[i] / sigma) + mean[i]     data/components/utils.py:183-199
//   log_pt                    x[..., 2] = 1 - exp(x[..., 2])
//   variable_set_sizes        x = x * mask
// The reference does this on the host with one D2H copy per batch inside the timed loop; here the batch stays in HBM.
#include <hip/hip_runtime.h>

#include "pfm_hip.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

__global__ __launch_bounds__(256) void sample_epilogue_kernel(float* __restrict__ x, const float* __restrict__ mask,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int log_col, int64_t n, int F) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t row = i / F;
    const int f = (int)(i - row * F);
    float v = x[i];
    if (scale) v = __fadd_rn(__fmul_rn(v, scale[f]), shift[f]);  // two roundings, like the torch ops
    if (f == log_col) v = 1.0f - expf(v);
    if (mask) v = __fmul_rn(v, mask[row]);
    x[i] = v;
}
}  // namespace pfm

extern "C" int pfm_sample_epilogue(float* x, const float* mask, const float* scale, const float* shift, int32_t log_pt_col,
                                   int64_t rows, int32_t features, void* stream) {
    using namespace pfm;
    if (rows <= 0) return 0;
    if (!x) return set_err(PFM_E_BADARG, "x is NULL");
    if ((scale == nullptr) != (shift == nullptr)) return set_err(PFM_E_BADARG, "scale and shift must be given together");
    if (features < 1 || log_pt_col >= features) return set_err(PFM_E_BADARG, "bad features / log_pt_col");
    const int64_t n = rows * features;
    hipLaunchKernelGGL(sample_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, scale,
                       shift, (int)log_pt_col, n, (int)features);
    return check_hip(hipGetLastError(), "sample_epilogue_kernel launch");
}

// ------------------------------------------------------------------------------------------------
// One state update of the reference's diffusion samplers (particle_fm/models/components/solver.py), op for op:
//   mode 0 = ddim_sampler :81-93   data = (x - nr * pred) / sr;  x <- nsr * data + nnr * pred     c = (nr, sr, nsr, nnr)
//   mode 1 = euler_maruyama :126-132   s = -pred / nr;  x <- x + 0.5 * beta * (x + 2 s) * dt;  x <- x + sqrt(beta dt) * noise
//                                                                                                c = (nr, beta, dt, sqrt(beta dt))
// ------------------------------------------------------------------------------------------------
namespace pfm {
__global__ __launch_bounds__(256) void diffusion_update_kernel(int mode, float* __restrict__ x, const float* __restrict__ pred,
                                                               const float* __restrict__ noise, float c0, float c1, float c2,
                                                               float c3, float* __restrict__ data_out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], p = pred[i];
    if (mode == 0) {
        const float data = __fdiv_rn(__fsub_rn(xv, __fmul_rn(c0, p)), c1);
        if (data_out) data_out[i] = data;
        x[i] = __fadd_rn(__fmul_rn(c2, data), __fmul_rn(c3, p));
    } else {
        const float s = __fdiv_rn(-p, c0);
        const float drift = __fmul_rn(__fmul_rn(__fmul_rn(0.5f, c1), __fadd_rn(xv, __fmul_rn(2.0f, s))), c2);
        x[i] = __fadd_rn(__fadd_rn(xv, drift), __fmul_rn(c3, noise[i]));
    }
}
}  // namespace pfm

extern "C" int pfm_diffusion_update(int32_t mode, float* x, const float* pred, const float* noise, float c0, float c1, float c2,
                                    float c3, float* data_out, int64_t n, void* stream) {
    using namespace pfm;
    if (n <= 0) return 0;
    if (mode < 0 || mode > 1) return set_err(PFM_E_BADARG, "mode must be 0 (ddim) or 1 (euler-maruyama)");
    if (!x || !pred || (mode == 1 && !noise)) return set_err(PFM_E_BADARG, "NULL device pointer");
    hipLaunchKernelGGL(diffusion_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)mode, x, pred,
                       noise, c0, c1, c2, c3, data_out, n);
    return check_hip(hipGetLastError(), "diffusion_update_kernel launch");
}

// ------------------------------------------------------------------------------------------------
// IterativeNormLayer (particle_fm/models/components/norm_layer.py:84-155): running per-feature standardisation of the
// valid particles.  One workgroup: the batch is a few 1e5 numbers and the update is two dependent reductions.
//   first batch (n == 0), fit():  vars, means = var_mean (unbiased);  n = len;  m2 = vars * n              :98-104
//   later batches, update():      n += len;  delta = x - means;  means += sum(delta) / n;  delta2 = x - means;
//                                 m2 += sum(delta * delta2);  vars = m2 / n                                 :137-152
// Nothing happens once n >= max_n (frozen, :154).
// ------------------------------------------------------------------------------------------------
namespace pfm {
__global__ __launch_bounds__(1024) void norm_update_kernel(const float* __restrict__ x, const float* __restrict__ mask, int64_t M,
                                                           int F, int64_t* __restrict__ n, float* __restrict__ means,
                                                           float* __restrict__ vars, float* __restrict__ m2, int64_t max_n) {
    __shared__ float red[64][17];
    __shared__ float mean_new[16];
    __shared__ float len_s;
    const int tid = threadIdx.x, f = tid & 15, g = tid >> 4;
    const int64_t n_old = *n;
    if (n_old >= max_n) return;  // frozen
    const bool first = n_old == 0;
    const float m_old = (f < F && !first) ? means[f] : 0.f;
    float s = 0.f, cnt = 0.f;
    for (int64_t r = g; r < M; r += 64) {
        if (mask && mask[r] == 0.f) continue;
        cnt += 1.f;
        if (f < F) s += x[r * F + f] - m_old;
    }
    red[g][f] = s;
    if (f == 0) red[g][16] = cnt;
    __syncthreads();
    if (tid < 17) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += red[i][tid];
        if (tid == 16) len_s = t;
        else mean_new[tid] = t;  // sum of (x - m_old) for now
    }
    __syncthreads();
    const float len = len_s;
    if (len == 0.f) return;
    const float n_new = (float)n_old + len;
    const float mean = m_old + mean_new[f] / n_new;  // first batch: m_old = 0, n_new = len
    __syncthreads();
    const float ref = first ? mean : m_old;  // first batch: sum (x - mean)^2; later: sum (x - old mean) (x - new mean)
    float q = 0.f;
    for (int64_t r = g; r < M; r += 64) {
        if (mask && mask[r] == 0.f) continue;
        if (f < F) {
            const float xv = x[r * F + f];
            q += (xv - ref) * (xv - mean);
        }
    }
    red[g][f] = q;
    __syncthreads();
    if (tid < F) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += red[i][tid];
        means[tid] = mean;  // tid == f here
        if (first) {
            const float v = t / (len - 1.f);
            vars[tid] = v;
            m2[tid] = v * len;
        } else {
            const float mm = m2[tid] + t;
            m2[tid] = mm;
            vars[tid] = mm / n_new;
        }
    }
    if (tid == 0) *n = n_old + (int64_t)len;
}

// forward (:116-126): (x - means) / (sqrt(vars) + 1e-8) on the valid rows, the rest untouched;  reverse (:128-139): x sqrt(vars) + means
__global__ __launch_bounds__(256) void norm_apply_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ mask,
                                                         int64_t n, int F, const float* __restrict__ means, const float* __restrict__ vars,
                                                         int reverse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t row = i / F;
    const int f = (int)(i - row * F);
    float v = x[i];
    if (!mask || mask[row] != 0.f) {
        const float sd = sqrtf(vars[f]);
        v = reverse ? __fadd_rn(__fmul_rn(v, sd), means[f]) : __fdiv_rn(__fsub_rn(v, means[f]), __fadd_rn(sd, 1e-8f));
    }
    out[i] = v;
}
}  // namespace pfm

extern "C" int pfm_norm_update(const float* x, const float* mask, int64_t rows, int32_t features, int64_t* n, float* means, float* vars,
                               float* m2, int64_t max_n, void* stream) {
    using namespace pfm;
    if (rows <= 0) return 0;
    if (!x || !n || !means || !vars || !m2) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (features < 1 || features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    hipLaunchKernelGGL(norm_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, mask, rows, (int)features, n, means, vars, m2,
                       max_n);
    return check_hip(hipGetLastError(), "norm_update_kernel launch");
}

extern "C" int pfm_norm_apply(float* out, const float* x, const float* mask, int64_t rows, int32_t features, const float* means,
                              const float* vars, int32_t reverse, void* stream) {
    using namespace pfm;
    if (rows <= 0) return 0;
    if (!out || !x || !means || !vars) return set_err(PFM_E_BADARG, "NULL device pointer");
    if (features < 1 || features > 16) return set_err(PFM_E_BADARG, "features must be in 1..16");
    const int64_t n = rows * features;
    hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, mask, n,
                       (int)features, means, vars, (int)reverse);
    return check_hip(hipGetLastError(), "norm_apply_kernel launch");
}
