// Diagnostic: what do op_sel / op_sel_hi select on v_pk_fma_f32's 64-bit operands (gfx950)?   hipcc --offload-arch=gfx950 pk_opsel.hip -o pk_opsel && ./pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* a, const f32x2* m, const f32x2* c, f32x2* o) {
    const int t = threadIdx.x;
    f32x2 r0 = c[t], r1 = c[t], r2 = c[t];
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r0) : "v"(a[t]), "v"(m[t]));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(r1) : "v"(a[t]), "v"(m[t]));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r2) : "v"(a[t]), "v"(m[t]));
    o[3 * t] = r0; o[3 * t + 1] = r1; o[3 * t + 2] = r2;
}
int main() {
    f32x2 ha[64], hm[64], hc[64], ho[192], *a, *m, *c, *o;
    for (int i = 0; i < 64; ++i) { ha[i] = f32x2{1.f + i, 100.f + i}; hm[i] = f32x2{2.f, 3.f}; hc[i] = f32x2{0.5f, 0.25f}; }
    hipMalloc(&a, sizeof ha); hipMalloc(&m, sizeof hm); hipMalloc(&c, sizeof hc); hipMalloc(&o, sizeof ho);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(m, hm, sizeof hm, hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, m, c, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const float a0 = ha[i].x, a1 = ha[i].y;
        const f32x2 w0 = {a0 * 2 + 0.5f, a1 * 2 + 0.25f}, w1 = {a0 * 3 + 0.5f, a1 * 3 + 0.25f}, w2 = {a0 * 2 + 0.5f, a1 * 3 + 0.25f};
        if (ho[3*i].x != w0.x || ho[3*i].y != w0.y || ho[3*i+1].x != w1.x || ho[3*i+1].y != w1.y || ho[3*i+2].x != w2.x || ho[3*i+2].y != w2.y) {
            if (bad++ < 3) printf("lane %d: lo-bcast (%g %g) want (%g %g); hi-bcast (%g %g) want (%g %g); plain (%g %g) want (%g %g)\n", i, ho[3*i].x, ho[3*i].y, w0.x, w0.y, ho[3*i+1].x, ho[3*i+1].y, w1.x, w1.y, ho[3*i+2].x, ho[3*i+2].y, w2.x, w2.y);
        }
    }
    printf("op_sel check: %d lanes wrong\n", bad);
    return 0;
}
