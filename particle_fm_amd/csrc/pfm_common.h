// Shared device helpers for the gfx950 EPiC flow-matching kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfm_hip.h"

namespace pfm {

constexpr int H = PFM_HIDDEN;  // hidden width this build is specialised for
constexpr int NT = 512;        // threads per workgroup: 8 waves, two per SIMD
constexpr int NW = NT / 64;    // waves; wave w owns output features [16w, 16w+16)
constexpr int TILE = 16;       // particles per MFMA tile (v_mfma_f32_16x16x4_f32)
constexpr int MAXT = 64;       // max time-embedding width
constexpr int MAXC = 16;       // max conditioning width
constexpr int MAXL = 16;       // max latent width
constexpr int MAXF = 16;       // max particle features

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- LDS carve (floats).  One workgroup = one jet. -------------------------------------------
// bufA/bufB hold the (N x 128) activation tiles, rows of 32 16-byte slots, slot index XOR-ed with
// (row & 15) so that the ds_read_b128 / ds_write_b128 lane groups of the MFMA operand pattern are
// bank-conflict free (cdna_hip_programming.md T2).  The global-MLP scratch aliases bufA, which is
// dead between a layer's phase 2 and the next layer's phase 1.
struct Carve {
    int bufA, bufB;  // N*H each
    int xs, yin;     // N*F : ODE state, network input
    int maskf;       // N (rounded to 4)
    int w3;          // F*H  head weights
    int bj1, bj2;    // H each: per-jet bias of the two local linears of the current layer
    int pooled;      // 2*H : mean | sum*scale
    int temb;        // MAXT
    int condv;       // MAXC
    int gvec;        // MAXL
    int misc;        // 16 : [0]=n_valid, [1]=1/n_valid
    int total;
    // scratch inside bufA while it is dead
    int s_vin, s_part, s_g1, s_part2;
};

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline Carve make_carve(int N, int F) {
    Carve c;
    int o = 0;
    c.bufA = o; o += N * H;
    c.bufB = o; o += N * H;
    c.xs = o; o += round4(N * F);
    c.yin = o; o += round4(N * F);
    c.maskf = o; o += round4(N);
    c.w3 = o; o += F * H;
    c.bj1 = o; o += H;
    c.bj2 = o; o += H;
    c.pooled = o; o += 2 * H;
    c.temb = o; o += MAXT;
    c.condv = o; o += MAXC;
    c.gvec = o; o += MAXL;
    c.misc = o; o += 16;
    c.total = o;
    // scratch (needs N*H >= 1408 floats, i.e. N >= 11)
    c.s_vin = c.bufA;                 // up to MAXT+MAXC+2H+MAXL = 352
    c.s_part = c.bufA + 384;          // 4*H = 512
    c.s_g1 = c.bufA + 896;            // H
    c.s_part2 = c.bufA + 1024;        // 32*16 = 512
    return c;
}

__device__ __forceinline__ int lds_off(int p, int slot) { return p * H + ((slot ^ (p & 15)) << 2); }

__device__ __forceinline__ float lrelu(float x, float slope) { return fmaxf(x, x * slope); }

__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float s) {
    f32x4 r;
    r.x = lrelu(v.x, s); r.y = lrelu(v.y, s); r.z = lrelu(v.z, s); r.w = lrelu(v.w, s);
    return r;
}

}  // namespace pfm
