// microbenchmark: issue rate of the fp16 / bf16 MFMA shapes on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));

template <int KIND, int CH>
__global__ void kern(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    h4 a4, b4; h8 a8, b8; s4 sa, sb;
    for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(threadIdx.x * 0.001f + i); b4[i] = (_Float16)(1.0f + i); sa[i] = (short)(threadIdx.x + i); sb[i] = (short)(i + 3); }
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(1.0f + i); }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[c], 0, 0, 0);
                if (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[c], 0, 0, 0);
                if (KIND == 2) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[c], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int c = 1; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(cyc, t0); atomicMax(cyc + 1, t1); }
}
template <typename K>
void run(const char* name, K k, int ch, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
    int iters = 2000;
    unsigned long long init[2] = {~0ull, 0ull};
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
    double n = (double)iters * 8 * ch;
    int wps = threads / 256;
    printf("%-14s chains=%d waves/SIMD=%d : %.1f memtime-cycles per MFMA per SIMD\n", name, ch, wps, (hh[1] - hh[0]) / n / wps);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512}) {
        run("f16 16x16x16", kern<0, 4>, 4, threads);
        run("f16 16x16x32", kern<1, 4>, 4, threads);
        run("bf16 16x16x16", kern<2, 4>, 4, threads);
        run("f16 16x16x16", kern<0, 1>, 1, threads);
        run("f16 16x16x32", kern<1, 1>, 1, threads);
    }
    return 0;
}
