// microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 for waves-per-SIMD x chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void k16(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int c = 1; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(cyc, t0); atomicMax(cyc + 1, t1); }
}
template <int CH>
__global__ void k32(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(cyc, t0); atomicMax(cyc + 1, t1); }
}
template <typename K>
void run(const char* name, K kern, int ch, int threads, int per_mfma_flop) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
    int iters = 2000;
    unsigned long long init[2] = {~0ull, 0ull};
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
    unsigned long long h = hh[1] - hh[0];
    int waves_per_simd = threads / 256;
    double n = (double)iters * 8 * ch;  // mfma per wave
    printf("%-10s chains=%d waves/SIMD=%d : %.1f cycles per MFMA per wave, %.1f cycles per MFMA per SIMD\n", name, ch,
           waves_per_simd, h / n, h / n / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        run("16x16x4", k16<1>, 1, threads, 2048);
        run("16x16x4", k16<2>, 2, threads, 2048);
        run("16x16x4", k16<4>, 4, threads, 2048);
        run("32x32x2", k32<1>, 1, threads, 4096);
        run("32x32x2", k32<2>, 2, threads, 4096);
    }
    return 0;
}
