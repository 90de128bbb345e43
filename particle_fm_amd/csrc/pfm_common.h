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

// Diagnostic build only (-DPFM_DIAG, never the shipped library): workgroup 0 records s_memtime at
// phase boundaries into a device array that tests/diag reads back.  Expands to nothing otherwise.
#ifdef PFM_DIAG
extern __device__ unsigned long long g_pfm_stamps[512];
extern __device__ int g_pfm_nstamp;
#define PFM_STAMP(id)                                                                       \
    do {                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                          \
            const int i_ = g_pfm_nstamp;                                                    \
            if (i_ < 256) {                                                                 \
                g_pfm_stamps[2 * i_] = ((unsigned long long)(id) << 48) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFFFull); \
                g_pfm_stamps[2 * i_ + 1] = __builtin_amdgcn_s_memtime();                    \
                g_pfm_nstamp = i_ + 1;                                                      \
            }                                                                               \
        }                                                                                   \
    } while (0)
#else
#define PFM_STAMP(id) do { } while (0)
#endif

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
    int vin;         // 352: [temb(T) ; cond(C) ; mean(H) ; sum*scale(H) ; g(L)]  input of the global MLP
    int misc;        // 16 : [0]=n_valid, [1]=last valid index
    int total;
    // scratch inside bufA while it is dead (between a layer's phase 2 and the next phase 1)
    int s_part;      // [16][H]  GEMV partials of fc_global1
    int s_pb1, s_pb2;  // [16][H] each: partials of the two local-bias GEMVs
    int s_vin2;      // 208: [temb ; cond ; g1]
    int s_part2;     // [32][16] partials of fc_global2
    int s_bj1p;      // H : bias of local linear 1 without its g part
};

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

constexpr int VIN_FLOATS = MAXT + MAXC + 2 * H + MAXL;  // 352
constexpr int SCRATCH_FLOATS = 3 * 16 * H + 208 + 512 + H;  // 6992 -> n_points >= 55? no: see below

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
    c.vin = o; o += VIN_FLOATS;
    c.misc = o; o += 16;
    // the scratch wants SCRATCH_FLOATS; bufA provides N*H.  For small sets (N < 55) the scratch
    // simply extends past bufA into a dedicated tail so that any N >= 1 works.
    const int need = SCRATCH_FLOATS > N * H ? SCRATCH_FLOATS - N * H : 0;
    int sbase = c.bufA;
    if (need > 0) { sbase = o; o += SCRATCH_FLOATS; }
    c.total = o;
    c.s_part = sbase;
    c.s_pb1 = sbase + 16 * H;
    c.s_pb2 = sbase + 32 * H;
    c.s_vin2 = sbase + 48 * H;
    c.s_part2 = c.s_vin2 + 208;
    c.s_bj1p = c.s_part2 + 512;
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
