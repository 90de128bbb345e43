// Shared device helpers for the gfx950 EPiC flow-matching kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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
// light stamp: one fire-and-forget store to a fixed slot (last writer wins), no counter traffic
#define PFM_MARK(slot)                                                                      \
    do {                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_pfm_stamps[384 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PFM_STAMP(id) do { } while (0)
#define PFM_MARK(slot) do { } while (0)
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
    int g2p;         // NW*MAXL: per-wave partial sums of fc_global2 (the k range is split over the waves)
    int bj1, bj2;    // H each: per-jet bias of the two local linears of the current layer
    int vin;         // 352: [temb(T) ; cond(C) ; mean(H) ; sum*scale(H) ; g(L)]  input of fc_global1
    int vin2;        // 208: [temb ; cond ; g1]  input of fc_global2
    int gcopy;       // NW*MAXL: per-wave copy of g_new (wave 0 uses vin.g)
    int misc;        // 32 : [0]=n_valid, [1]=last valid index, [8..24) reduction scratch
    int dummy;       // 4 : sink for the (predicated-off) stores of rows >= N
    int total;
};

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

constexpr int VIN_FLOATS = MAXT + MAXC + 2 * H + MAXL;  // 352
constexpr int VIN2_FLOATS = MAXT + MAXC + H;            // 208

__host__ __device__ inline Carve make_carve(int N, int F) {
    Carve c;
    int o = 0;
    c.bufA = o; o += N * H;
    c.bufB = o; o += N * H;
    c.xs = o; o += round4(N * F);
    c.yin = o; o += round4(N * F);
    c.maskf = o; o += round4(N);
    c.g2p = o; o += NW * MAXL;
    c.bj1 = o; o += H;
    c.bj2 = o; o += H;
    c.vin = o; o += VIN_FLOATS;
    c.vin2 = o; o += VIN2_FLOATS;
    c.gcopy = o; o += NW * MAXL;
    c.misc = o; o += 32;
    c.dummy = o; o += 4;
    // the MFMA phases read particle rows in pairs of 16-row tiles WITHOUT clamping: up to 31 rows past the end of
    // bufB may be touched (read only; their results are never stored).  Keep that window inside the allocation.
    {
        const int over = ((32 - N % 32) % 32) * H;
        const int tail = o - (c.bufB + N * H);
        if (over > tail) o += over - tail;
    }
    c.total = o;
    return c;
}

// ---- a SECOND jet in the workgroup (the packed sampler of epic_kernels.hip) ---------------------------------------------
// Two short jets may share a workgroup: their rows sit one behind the other in bufA / bufB (the second jet starts on a 16-row
// tile boundary), the weight streams are fetched once for both.  The second jet's per-jet vectors live in the tail of bufB, which
// such a workgroup never fills: it is formed only if  pad16(rows of jet 0) + rows of jet 1 <= seg2_rows(N).
constexpr int X2_VIN = 0, X2_VIN2 = X2_VIN + VIN_FLOATS, X2_BJ1 = X2_VIN2 + VIN2_FLOATS, X2_BJ2 = X2_BJ1 + H,
              X2_GCOPY = X2_BJ2 + H, X2_G2P = X2_GCOPY + NW * MAXL, X2_MISC = X2_G2P + NW * MAXL, X2_MASK = X2_MISC + 8;
__host__ __device__ inline int x2_floats(int N) { return X2_MASK + round4(N); }
__host__ __device__ inline int seg2_rows(int N) { return N - (x2_floats(N) + H - 1) / H; }
// LDS offsets (floats) of one jet's per-jet vectors: segment 0 = the carve's own, segment 1 = the tail of bufB.
// maskf: segment 0 -> the TRUE mask of every row of the workgroup; segment 1 -> the mask of segment 1's rows, 0 elsewhere.
struct SegView {
    int vin, vin2, bj1, bj2, gcopy, g2p, misc, maskf;
};
__host__ __device__ inline SegView seg_view(const Carve& c, int N, int s) {
    if (s == 0) return SegView{c.vin, c.vin2, c.bj1, c.bj2, c.gcopy, c.g2p, c.misc, c.maskf};
    const int b = c.bufB + N * H - x2_floats(N);
    return SegView{b + X2_VIN, b + X2_VIN2, b + X2_BJ1, b + X2_BJ2, b + X2_GCOPY, b + X2_G2P, b + X2_MISC, b + X2_MASK};
}
// The jets of a workgroup (wave-uniform).  One jet: nseg = 1, rows = its computed rows.
struct Segs {
    int nseg;    // 1 or 2
    int r1;      // first LDS row of segment 1 (a multiple of 16; segment 0 starts at row 0); 1 << 20 if there is none
    int n0, n1;  // rows computed for each segment (last valid particle + 1)
    int rows;    // r1 + n1, or n0
};

// index of element (k, o) of a KM16 block (K-major, OUT = 128, rows in blocks of 16; see pfm_hip.h)
__host__ __device__ inline int km16(int k, int o) { return ((k >> 4) * 32 + (o >> 2)) * 64 + (k & 15) * 4 + (o & 3); }

// Opaque copy of a per-lane value.  Every phase of the kernel derives a dozen per-lane constants (swizzled LDS
// offsets, GEMV row indices) from threadIdx.x; without this hipcc hoists ALL of them out of the layer loop and
// keeps them live (and spilled) for the whole kernel.  Laundering the seed makes each phase recompute its own.
__device__ __forceinline__ int launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// LDS hand-over between lanes of ONE wave (a value written by some lanes, read by others of the same wave): the LDS unit executes a
// wave's operations in issue order, so all that is needed is that the compiler keeps the write in front of the read.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- DPP row reductions: sum over the 16 lanes of a DPP row with four v_add_f32_dpp (no LDS, no address math)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return v;                 // every lane of the row holds the row's sum
}
// Four components at once, the add fused into the DPP instruction (hipcc pairs the components into v_pk_add_f32, which cannot take
// a DPP operand, and emits v_mov_b32_dpp + v_pk_add_f32: 24 instructions where 16 do).  Same tree, same bits as the scalar version.
// A DPP source written by the VALU instruction right before needs two wait states: the four independent chains are interleaved (three
// instructions between a write and its DPP read) and the block starts with an s_nop for whatever produced the inputs.
__device__ __forceinline__ f32x4 row_sum16(f32x4 v) {
    float x = v.x, y = v.y, z = v.z, w = v.w;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    return f32x4{x, y, z, w};
}

// ---- weight blob through a buffer resource: address = SGPR base + SGPR offset + one per-lane VGPR + immediate,
//      so streaming a panel costs no per-load 64-bit VALU address arithmetic (cdna_hip_programming.md T8)
typedef __amdgpu_buffer_rsrc_t blob_rsrc;
__device__ __forceinline__ blob_rsrc make_blob_rsrc(const float* blob, int64_t total_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blob), 0, (int)(total_floats * 4), 0x00020000);
}
// 16 bytes at  blob + 4*elem_off (wave-uniform) + lane_bytes (per lane)
__device__ __forceinline__ f32x4 bload4(blob_rsrc rs, int64_t elem_off, int lane_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, (int)(elem_off << 2), 0));
}
// 16 bytes to  base + 4*elem_off (wave-uniform) + lane_bytes (per lane); a lane whose offset lies outside the resource stores nothing
__device__ __forceinline__ void bstore4(blob_rsrc rs, int64_t elem_off, int lane_bytes, f32x4 v) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rs, lane_bytes, (int)(elem_off << 2), 0);
}

// split-fp16 operand helpers (PFM_F_F16X3_MFMA; see epic_nfe.h): x = hi + lo * 2^-11, hi = fp16(x), lo = fp16((x - hi) * 2^11)
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float X3_UP = 2048.0f, X3_DN = 1.0f / 2048.0f;
__device__ __forceinline__ void x3_split(f32x4 x, h4& hi, h4& lo) {
    hi = __builtin_convertvector(x, h4);
    lo = __builtin_convertvector((x - __builtin_convertvector(hi, f32x4)) * X3_UP, h4);
}
__device__ __forceinline__ f32x4 x3_join(h4 hi, h4 lo) {
    return __builtin_convertvector(hi, f32x4) + __builtin_convertvector(lo, f32x4) * X3_DN;
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void x3_split8(f32x4 x0, f32x4 x1, h8& hi, h8& lo) {
    h4 h0, l0, h1, l1;
    x3_split(x0, h0, l0);
    x3_split(x1, h1, l1);
    hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// bf16 operands (PFM_F_BF16_MFMA; the kernels issue v_mfma_f32_16x16x32_bf16, see pack_bf16x8): the fp32 float4 a lane already holds -- four
// consecutive k of one row / column -- is exactly that instruction's operand after rounding, so one MFMA replaces four.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// (plain vector conversion, not inline asm: the compiler must see the VALU write to insert the MFMA read hazard nop)
__device__ __forceinline__ s16x4 pack_bf16(f32x4 v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};  // one v_cvt_pk_bf16_f32 each (round-to-nearest-even, gfx950)
    const u32x2 u = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
    return __builtin_bit_cast(s16x4, u);
}

// gfx950's v_mfma_f32_16x16x32_bf16 contracts 32 k per instruction (lane (i, q) holds 8 of them) in 16 cycles.  The two float4
// a lane holds for k-tiles 2 kt2 and 2 kt2 + 1 -- k = 32 kt2 + 16 h + 4 q + r -- are, rounded and concatenated, a valid operand: the
// instruction sums over its 32 k slots whatever their order, as long as A and B use the same one.  One MFMA replaces eight fp32 ones.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack_bf16x8(f32x4 lo, f32x4 hi) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 a = pack_bf16(lo), b = pack_bf16(hi);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int lds_off(int p, int slot) { return p * H + ((slot ^ (p & 15)) << 2); }

__device__ __forceinline__ float lrelu(float x, float slope) { return fmaxf(x, x * slope); }

__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float s) {
    return __builtin_elementwise_max(v, v * s);  // two v_pk_mul_f32 + four v_max_f32 (+ four canonicalising v_max_f32 v, v, v)
}

// ---- packed-fp32 epilogue of the MFMA particle phases, as inline asm ------------------------------------------------------
// Why asm.  VALU work does not co-issue with v_mfma_f32_16x16x4_f32 (tests/diag/mfma_coissue.hip: the times add), so every VALU
// instruction of a pair body is matrix-pipe time -- and hipcc emits twice as many as the arithmetic needs (tests/diag/isa_mix.py on
// the round-3 build: 25 - 63 VALU instructions per 64 MFMAs):
//   * llvm.maxnum quiets signalling NaNs, and an MFMA result is not known to be canonical: every max(x, s x) comes with a
//     `v_max_f32 x, x, x` in front (8 per tile pair);
//   * the pre-emit peephole UNPACKS v_pk_mul / v_pk_add / v_pk_fma_f32 that sit in the shadow of an MFMA into two VOP3 instructions
//     each ("so that they may co-issue with the matrix pipe": true for the XDL shapes, not for the fp32 one).
// The hazards the compiler cannot see through an asm statement are handled by construction:
//   * MFMA write -> VALU read of the accumulators needs 10 wait states for the 8-pass v_mfma_f32_16x16x4_f32 (7 for the 4-pass XDL
//     shapes): lrelu8_pk starts with `s_nop 11` and multiplies all eight inputs in ONE statement, every later statement consumes its
//     outputs (in-order issue: whatever follows a dependent instruction is past the window too);
//   * nothing here feeds an MFMA operand: results go to LDS / global stores and to VALU sums.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vmax_raw(float a, float b) {  // v_max_f32 without the canonicalising copy (inputs: VALU results)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// e0, e1 <- max(e, s e) of two accumulator tiles; s2 = (s, s) in an SGPR pair.  4 v_pk_mul + 8 v_max.
// FRESH = true: the accumulators may come straight from the matrix pipe (s_nop 11 in front).  FRESH = false: the CALLER guarantees that
// at least ten wait states of other instructions (e.g. a K-quarter of MFMAs of another accumulator) lie between the MFMA that wrote
// them and this call -- gemm_phase's interior pairs, whose epilogue sits behind the first K-quarter of the next pair.
template <bool FRESH = true>
__device__ __forceinline__ void lrelu8_pk(f32x4& e0, f32x4& e1, f32x2 s2) {
#ifdef PFM_AB_OLD_LRELU  // (diagnostic A/B builds only)
    e0 = lrelu4(e0, s2.x); e1 = lrelu4(e1, s2.x);
    return;
#endif
    f32x2 m0, m1, m2, m3;
    const f32x2 a0 = {e0.x, e0.y}, a1 = {e0.z, e0.w}, a2 = {e1.x, e1.y}, a3 = {e1.z, e1.w};
    if constexpr (FRESH) {
        asm("s_nop 11\n\t"
            "v_pk_mul_f32 %0, %4, %8\n\t"
            "v_pk_mul_f32 %1, %5, %8\n\t"
            "v_pk_mul_f32 %2, %6, %8\n\t"
            "v_pk_mul_f32 %3, %7, %8"
            : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(s2));
    } else {
        asm("v_pk_mul_f32 %0, %4, %8\n\t"
            "v_pk_mul_f32 %1, %5, %8\n\t"
            "v_pk_mul_f32 %2, %6, %8\n\t"
            "v_pk_mul_f32 %3, %7, %8"
            : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(s2));
    }
    e0 = f32x4{vmax_raw(a0.x, m0.x), vmax_raw(a0.y, m0.y), vmax_raw(a1.x, m1.x), vmax_raw(a1.y, m1.y)};
    e1 = f32x4{vmax_raw(a2.x, m2.x), vmax_raw(a2.y, m2.y), vmax_raw(a3.x, m3.x), vmax_raw(a3.y, m3.y)};
}
// ps += e0 * m0 + e1 * m1 (masked pool sum of two tiles; e0 / e1: outputs of lrelu8_pk): four v_pk_fma_f32.  Each mask value sits in
// the LOW half of a register pair of its own and is broadcast by op_sel_hi:[1,0,1], the form hipcc itself emits (the high halves are
// never read: left undefined, no copy).  NOT one pair (m0, m1) with the high half selected by op_sel:[0,1,0] op_sel_hi:[1,1,1]: that form
// passes in isolation (tests/diag/pk_opsel.hip) and gave run-to-run different pool sums inside the quad kernel (tests/diag/ab_quad.py;
// explicit (m, m) pairs or this form: bit-identical to the compiler's code) -- cause not found, form avoided.
__device__ __forceinline__ void pool2_pk(f32x4& ps, const f32x4& e0, const f32x4& e1, float m0, float m1) {
#ifdef PFM_AB_OLD_POOL  // (diagnostic A/B builds only)
    ps += e0 * m0; ps += e1 * m1;
    return;
#endif
    f32x2 lo = {ps.x, ps.y}, hi = {ps.z, ps.w};
    const f32x2 a0 = {e0.x, e0.y}, a1 = {e0.z, e0.w}, a2 = {e1.x, e1.y}, a3 = {e1.z, e1.w};
    f32x2 p0, p1;  // .y deliberately not set
    p0.x = m0;
    p1.x = m1;
    asm("v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %3, %6, %1 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %7, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %5, %7, %1 op_sel_hi:[1,0,1]"
        : "+v"(lo), "+v"(hi)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p0), "v"(p1));
    ps = f32x4{lo.x, lo.y, hi.x, hi.y};
}

}  // namespace pfm
