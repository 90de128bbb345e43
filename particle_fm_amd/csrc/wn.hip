// Weight-norm reparametrisation as two gfx950 kernels over FLAT parameter / gradient buffers.
//
// Reference: old-style torch.nn.utils.weight_norm on every Linear of the EPiC network
// (particle_fm/models/components/epic.py:66-81, 262-300): W[o,:] = g[o] * v[o,:] / ||v[o,:]||_2, recomputed by a
// forward pre-hook on every call and differentiated by autograd (29 Linears x ~10 tiny launches each way).
// Here: one launch packs all effective weights + biases straight into the kernel blob (MFMA_A, MFMA_AT and
// K-major positions), one launch turns the gradient blob into d weight_g / d weight_v / d bias.
// One 64-lane wave per weight row; HBM traffic = parameters once (2.2 MB) -- launch-latency bound.
#include <hip/hip_runtime.h>

#include "pfm_common.h"

namespace pfm {
int set_err(int code, const char* what);
int check_hip(hipError_t e, const char* where);

struct WnRow {
    int32_t v_off;    // row of weight_v in the flat parameter buffer
    int32_t g_off;    // weight_g element
    int32_t in_dim;   // row length
    int32_t src_off;  // index of W[o][0] in the layout's source vector (indexes dst / gsrc maps)
};

__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void wn_pack_kernel(const float* __restrict__ params, const WnRow* __restrict__ rows,
                                                      int n_rows, const int32_t* __restrict__ dst1,
                                                      const int32_t* __restrict__ dst2,
                                                      const int32_t* __restrict__ bias_from,
                                                      const int32_t* __restrict__ bias_to, int n_bias,
                                                      float* __restrict__ blob) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < n_rows) {
        const WnRow r = rows[row];
        const float* v = params + r.v_off;
        float ss = 0.f;
        for (int c = lane; c < r.in_dim; c += 64) ss = fmaf(v[c], v[c], ss);
        const float scale = params[r.g_off] / sqrtf(wave_sum(ss));
        for (int c = lane; c < r.in_dim; c += 64) {
            const float w = v[c] * scale;
            const int d1 = dst1[r.src_off + c], d2 = dst2[r.src_off + c];
            if (d1 >= 0) blob[d1] = w;
            if (d2 >= 0) blob[d2] = w;
        }
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = gid; i < n_bias; i += gridDim.x * blockDim.x) blob[bias_to[i]] = params[bias_from[i]];
}

// dW -> dg = <dW, v>/||v||,  dv = (g/||v||) (dW - v <dW, v>/||v||^2);  grad += (accumulates like autograd)
// ACC: grad += (autograd's accumulate semantics); !ACC: grad = (the fused trainer, whose tables cover every parameter: no zeroing launch)
template <bool ACC>
__global__ __launch_bounds__(256) void wn_unpack_grad_kernel(const float* __restrict__ params,
                                                             const float* __restrict__ gblob,
                                                             const WnRow* __restrict__ rows, int n_rows,
                                                             const int32_t* __restrict__ gsrc,
                                                             const int32_t* __restrict__ bias_from,
                                                             const int32_t* __restrict__ bias_to, int n_bias,
                                                             float* __restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < n_rows) {
        const WnRow r = rows[row];
        const float* v = params + r.v_off;
        float ss = 0.f, dot = 0.f;
        for (int c = lane; c < r.in_dim; c += 64) {
            const float vc = v[c];
            ss = fmaf(vc, vc, ss);
            dot = fmaf(gblob[gsrc[r.src_off + c]], vc, dot);
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float inv_n = 1.0f / sqrtf(ss);
        const float g = params[r.g_off];
        if (lane == 0) grad[r.g_off] = (ACC ? grad[r.g_off] : 0.f) + dot * inv_n;
        const float k1 = g * inv_n, k2 = dot / ss;
        for (int c = lane; c < r.in_dim; c += 64)
            grad[r.v_off + c] = (ACC ? grad[r.v_off + c] : 0.f) + k1 * (gblob[gsrc[r.src_off + c]] - v[c] * k2);
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = gid; i < n_bias; i += gridDim.x * blockDim.x) grad[bias_to[i]] = (ACC ? grad[bias_to[i]] : 0.f) + gblob[bias_from[i]];
}
}  // namespace pfm

using namespace pfm;

extern "C" int pfm_wn_pack(const float* params, const int32_t* rows, int32_t n_rows, const int32_t* dst1,
                           const int32_t* dst2, const int32_t* bias_from, const int32_t* bias_to, int32_t n_bias,
                           float* blob, void* stream) {
    if (!params || !rows || !dst1 || !dst2 || !blob || (n_bias > 0 && (!bias_from || !bias_to)))
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(wn_pack_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<const WnRow*>(rows), n_rows, dst1, dst2, bias_from, bias_to, n_bias, blob);
    return check_hip(hipGetLastError(), "wn_pack_kernel launch");
}

extern "C" int pfm_wn_unpack_grad(const float* params, const float* gblob, const int32_t* rows, int32_t n_rows,
                                  const int32_t* gsrc, const int32_t* bias_from, const int32_t* bias_to,
                                  int32_t n_bias, float* grad, void* stream) {
    if (!params || !gblob || !rows || !gsrc || !grad || (n_bias > 0 && (!bias_from || !bias_to)))
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(wn_unpack_grad_kernel<true>, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, gblob,
                       reinterpret_cast<const WnRow*>(rows), n_rows, gsrc, bias_from, bias_to, n_bias, grad);
    return check_hip(hipGetLastError(), "wn_unpack_grad_kernel launch");
}

extern "C" int pfm_wn_unpack_grad_set(const float* params, const float* gblob, const int32_t* rows, int32_t n_rows,
                                      const int32_t* gsrc, const int32_t* bias_from, const int32_t* bias_to,
                                      int32_t n_bias, float* grad, void* stream) {
    if (!params || !gblob || !rows || !gsrc || !grad || (n_bias > 0 && (!bias_from || !bias_to)))
        return set_err(PFM_E_BADARG, "NULL device pointer");
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(wn_unpack_grad_kernel<false>, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, gblob,
                       reinterpret_cast<const WnRow*>(rows), n_rows, gsrc, bias_from, bias_to, n_bias, grad);
    return check_hip(hipGetLastError(), "wn_unpack_grad_kernel launch");
}
