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
