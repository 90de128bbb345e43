// microbenchmark: does VALU work co-issue with v_mfma_f32_16x16x4_f32 on a SIMD?  512 threads = two waves per SIMD, one workgroup per CU.
//   mode 0: waves 0-3 run MFMAs (two accumulator chains), waves 4-7 idle          -> T_mfma
//   mode 1: waves 0-3 idle, waves 4-7 run NV independent v_fma_f32 per iteration   -> T_valu
//   mode 2: both at once                                                            -> max(T_mfma, T_valu) if they co-issue, the sum if not
//   mode 3: every wave runs both, interleaved in one instruction stream
// everything inside an iteration is unrolled (no inner loops), so the loop overhead is one s_add/s_cmp/s_cbranch per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_16x16x4_f32 (8 passes), KIND 1: v_mfma_f32_16x16x32_f16 (8 passes)
template <int KIND>
__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 acc) {
    if (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)a; y[i] = (_Float16)b; }
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
}

template <int KIND, int NV, bool DM, bool DV>
__device__ __forceinline__ void body(int iters, float a, float b, f32x4& acc0, f32x4& acc1, float (&v)[8]) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (DM) {
                acc0 = mm<KIND>(a, b, acc0);
                acc1 = mm<KIND>(b, a, acc1);
            }
            if (DV) {
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
            }
        }
    }
}

template <int KIND, int NV>
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, unsigned long long* cyc) {
    const int w = threadIdx.x >> 6;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 3) body<KIND, NV, true, true>(iters, a, b, acc0, acc1, v);
    else if (w < 4) { if (mode == 0 || mode == 2) body<KIND, NV, true, false>(iters, a, b, acc0, acc1, v); }
    else { if (mode == 1 || mode == 2) body<KIND, NV, false, true>(iters, a, b, acc0, acc1, v); }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = acc0.x + acc1.y;
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; }
}
// the same issue stream on operands that toggle (eight pseudo-random values per lane and operand, accumulators random-walk): the clock
// the power management leaves under a realistic fp32 MFMA load
__global__ void __launch_bounds__(512) k_data(float* out, int iters, unsigned long long* cyc) {
    float a[8], b[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u; b[i] = (int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + u) & 7], acc[r & 3], 0, 0, 0);
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NV>
void run(float* out, unsigned long long* cyc) {
    const int iters = 4000;
    printf("8 x %s + %3d v_fma_f32 per iteration:", KIND ? "mfma_f32_16x16x32_f16" : "mfma_f32_16x16x4_f32 ", NV);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(512), 0, 0, out, iters, mode, cyc);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("  mode %d: %6.1f", mode, (double)h / iters);
    }
    printf("   (clocks per iteration)\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    run<0, 0>(out, cyc); run<0, 16>(out, cyc); run<0, 32>(out, cyc); run<0, 64>(out, cyc); run<0, 128>(out, cyc);
    run<1, 0>(out, cyc); run<1, 16>(out, cyc); run<1, 32>(out, cyc); run<1, 64>(out, cyc); run<1, 128>(out, cyc);
    // sustained whole-chip rate of the fp32 MFMA alone (two waves per SIMD on every CU), wall clock: what the clock under load leaves of the
    // nominal 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 100000 << rep;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, out, iters, 3, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 8 * 8.0 * iters * 2048.0;
        printf("fp32 MFMA alone, %d iterations: %.2f ms, %.1f TFLOP/s, %.3f GHz (cycle counter / wall)\n", iters, ms, flop / ms * 1e-9, h / (ms * 1e6));
    }
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 100000 << rep;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_data, dim3(256), dim3(512), 0, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 8 * 8.0 * iters * 2048.0;
        printf("fp32 MFMA on toggling operands, %d iterations: %.2f ms, %.1f TFLOP/s, %.3f GHz, %.1f clocks per 8 MFMAs of each of the two waves of a SIMD\n", iters, ms, flop / ms * 1e-9, h / (ms * 1e6), (double)h / iters);
    }
    return 0;
}
