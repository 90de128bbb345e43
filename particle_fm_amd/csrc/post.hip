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
