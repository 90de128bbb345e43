// Attention kernels of the cross-attention encoder (droid_transformer.py:442-472): a handful of global tokens attend to
// all particles of the jet (key mask), then every particle attends to the tokens.  Both shapes are tiny next to the
// Linears (2 * 2 * N * tokens * D FLOP per jet and layer), so they run on the VALU:
//   *_from_* : grid (jets, heads / 4): wave per head, lanes over the keys (particles), ONE pass with a running softmax
//              per lane (max, sum, weighted values for every token), merged over the wave by shuffles.  The four heads
//              of a workgroup read neighbouring 32..64-byte pieces of the same key rows, so every fetched line is used.
//   *_to_*   : grid (jets, ceil(N / 64)): thread per (particle, head); the jet's token keys / values sit in LDS
// HD = head_dim (8 in fm_droid_crossattention.yaml, 16 supported), TK = compile-time bound on the number of tokens (4 or 8).
#pragma once
#include "tf_common.h"

namespace pfm {
namespace ca {

using pfm::f32x4;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <int HD>
__device__ __forceinline__ void load_row(float (&r)[HD], const float* __restrict__ p) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}
template <int HD>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&r)[HD]) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) *reinterpret_cast<f32x4*>(p + 4 * i) = f32x4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
}

constexpr int TO_ROWS = 64;  // particles per workgroup of the *_to_* kernels

// running softmax of one lane over its keys, for TK query tokens
template <int HD, int TK>
struct Running {
    float m[TK], l[TK], o[TK][HD];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            m[tk] = -__builtin_inff();
            l[tk] = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[tk][d] = 0.f;
        }
    }
};

template <int HD, int TK>
__device__ __forceinline__ void load_queries(float (&qr)[TK][HD], const float* __restrict__ q, int Tk, int D, float scale) {
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        if (tk < Tk) {
            load_row<HD>(qr[tk], q + (int64_t)tk * D);
#pragma unroll
            for (int d = 0; d < HD; ++d) qr[tk][d] *= scale;
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) qr[tk][d] = 0.f;
        }
    }
}

// ---- tokens <- particles ---------------------------------------------------------------------------
// q [n_jets*Tk][D]; kv [n_jets*N][2D] (k | v); mask [n_jets][N] or nullptr; out [n_jets*Tk][D]
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_from_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                           const float* __restrict__ mask, float* __restrict__ out, int N,
                                                           int D, int heads, int Tk, const int* __restrict__ off = nullptr,
                                                           const int* __restrict__ order = nullptr) {
    const int jet = order ? order[blockIdx.x] : blockIdx.x;  // longest jets first (tf_fwd.h: rows_rank_kernel)
    const int lane = threadIdx.x & 63, h = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (h >= heads) return;
    // compacted particle rows (off != nullptr): the jet's keys are rows [off[jet], off[jet + 1]), all valid
    int64_t row_base = (int64_t)jet * N;
    if (off) {
        row_base = off[jet];
        N = off[jet + 1] - off[jet];
        mask = nullptr;
    }
    float qr[TK][HD];  // pre-scaled by 1 / sqrt(HD)
    load_queries<HD, TK>(qr, q + (int64_t)jet * Tk * D + h * HD, Tk, D, 1.0f / sqrtf((float)HD));
    Running<HD, TK> st;
    st.init();
    const float* base = kv + row_base * 2 * D + h * HD;
#pragma unroll 2
    for (int n = lane; n < N; n += 64) {
        if (mask && mask[row_base + n] == 0.f) continue;
        float kr[HD], vr[HD];
        load_row<HD>(kr, base + (int64_t)n * 2 * D);
        load_row<HD>(vr, base + (int64_t)n * 2 * D + D);
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
            const float mn = fmaxf(st.m[tk], s);
            const float corr = __expf(st.m[tk] - mn), p = __expf(s - mn);  // first key: exp(-inf) = 0
            st.m[tk] = mn;
            st.l[tk] = fmaf(st.l[tk], corr, p);
#pragma unroll
            for (int d = 0; d < HD; ++d) st.o[tk][d] = fmaf(st.o[tk][d], corr, p * vr[d]);
        }
    }
    // merge the 64 lanes; all keys masked: max = -inf -> NaN like torch's softmax over an empty row
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        const float M = wmax(st.m[tk]);
        const float w = st.m[tk] == -__builtin_inff() ? 0.f : __expf(st.m[tk] - M);
        const float L = wsum(st.l[tk] * w);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float o = wsum(st.o[tk][d] * w);
            if (lane == tk * HD + d && tk < Tk)
                out[((int64_t)jet * Tk + tk) * D + h * HD + d] = (M == -__builtin_inff()) ? __builtin_nanf("") : o / L;
        }
    }
}

// backward: dq [n_jets*Tk][D], dkv [n_jets*N][2D] (every row written: masked keys get 0)
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_from_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                               const float* __restrict__ mask, const float* __restrict__ O,
                                                               const float* __restrict__ dO, float* __restrict__ dq,
                                                               float* __restrict__ dkv, int N, int D, int heads, int Tk) {
    const int jet = blockIdx.x, lane = threadIdx.x & 63, h = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (h >= heads) return;
    const float scale = 1.0f / sqrtf((float)HD);
    float qr[TK][HD], dor[TK][HD], delta[TK];
    load_queries<HD, TK>(qr, q + (int64_t)jet * Tk * D + h * HD, Tk, D, scale);
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        delta[tk] = 0.f;
        if (tk < Tk) {
            const int64_t e = ((int64_t)jet * Tk + tk) * D + h * HD;
            load_row<HD>(dor[tk], dO + e);
            float orow[HD];
            load_row<HD>(orow, O + e);
#pragma unroll
            for (int d = 0; d < HD; ++d) delta[tk] = fmaf(dor[tk][d], orow[d], delta[tk]);
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) dor[tk][d] = 0.f;
        }
    }
    const float* base = kv + (int64_t)jet * N * 2 * D + h * HD;
    // pass 1: softmax statistics (running max and sum per lane, merged over the wave)
    float m[TK], l[TK];
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) { m[tk] = -__builtin_inff(); l[tk] = 0.f; }
#pragma unroll 2
    for (int n = lane; n < N; n += 64) {
        if (mask && mask[(int64_t)jet * N + n] == 0.f) continue;
        float kr[HD];
        load_row<HD>(kr, base + (int64_t)n * 2 * D);
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
            const float mn = fmaxf(m[tk], s);
            l[tk] = fmaf(l[tk], __expf(m[tk] - mn), __expf(s - mn));
            m[tk] = mn;
        }
    }
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        const float M = wmax(m[tk]);
        const float w = m[tk] == -__builtin_inff() ? 0.f : __expf(m[tk] - M);
        l[tk] = 1.0f / wsum(l[tk] * w);
        m[tk] = M;
    }
    // pass 2: gradients
    float dqa[TK][HD];
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
        for (int d = 0; d < HD; ++d) dqa[tk][d] = 0.f;
    float* gbase = dkv + (int64_t)jet * N * 2 * D + h * HD;
#pragma unroll 2
    for (int n = lane; n < N; n += 64) {
        float dk[HD], dv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        if (!(mask && mask[(int64_t)jet * N + n] == 0.f)) {
            float kr[HD], vr[HD];
            load_row<HD>(kr, base + (int64_t)n * 2 * D);
            load_row<HD>(vr, base + (int64_t)n * 2 * D + D);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) { s = fmaf(qr[tk][d], kr[d], s); dp = fmaf(dor[tk][d], vr[d], dp); }
                const float p = __expf(s - m[tk]) * l[tk];
                const float ds = p * (dp - delta[tk]);  // d loss / d (scaled score); qr carries the 1 / sqrt(HD)
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dqa[tk][d] = fmaf(ds, kr[d], dqa[tk][d]);
                    dk[d] = fmaf(ds, qr[tk][d], dk[d]);
                    dv[d] = fmaf(p, dor[tk][d], dv[d]);
                }
            }
        }
        store_row<HD>(gbase + (int64_t)n * 2 * D, dk);
        store_row<HD>(gbase + (int64_t)n * 2 * D + D, dv);
    }
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float s = wsum(dqa[tk][d]) * scale;
            if (lane == tk * HD + d && tk < Tk) dq[((int64_t)jet * Tk + tk) * D + h * HD + d] = s;
        }
}

// ---- particles <- tokens ---------------------------------------------------------------------------
// q [n_jets*N][D]; kv [n_jets*Tk][2D]; out [n_jets*N][D]; no key mask (droid_transformer.py:470)
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_to_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                         float* __restrict__ out, int N, int D, int heads, int Tk,
                                                         const int* __restrict__ off = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [Tk][2D]
    const int jet = blockIdx.x, r0 = blockIdx.y * TO_ROWS;
    int64_t row_base = (int64_t)jet * N;
    if (off) {  // compacted particle rows
        row_base = off[jet];
        N = off[jet + 1] - off[jet];
    }
    if (r0 >= N) return;
    const int nr = min(TO_ROWS, N - r0);
    for (int i = threadIdx.x; i < Tk * 2 * D; i += 256) lds[i] = kv[(int64_t)jet * Tk * 2 * D + i];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    for (int idx = threadIdx.x; idx < nr * heads; idx += 256) {
        const int r = r0 + idx / heads, h = idx % heads;
        const int64_t e = (row_base + r) * D + h * HD;
        float qr[HD];
        load_row<HD>(qr, q + e);
        float s[TK], mx = -__builtin_inff();
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            s[tk] = -__builtin_inff();
            if (tk < Tk) {
                float kr[HD];
                load_row<HD>(kr, lds + tk * 2 * D + h * HD);
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a = fmaf(qr[d], kr[d], a);
                s[tk] = a * scale;
            }
            mx = fmaxf(mx, s[tk]);
        }
        float l = 0.f, o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
            if (tk < Tk) {
                const float p = __expf(s[tk] - mx);
                l += p;
                float vr[HD];
                load_row<HD>(vr, lds + tk * 2 * D + D + h * HD);
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] = fmaf(p, vr[d], o[d]);
            }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= inv;
        store_row<HD>(out + e, o);
    }
}

// backward, workgroup = (jet, block of `rpb` particles):
//   phase 1, thread per (particle, head): softmax weights p and score gradients ds -> LDS, dq -> global
//   phase 2, thread per model column: dk / dv of the block = sums over its particles of ds * q and p * dO (coalesced
//            re-reads of the q / dO rows, p / ds broadcast from LDS) -> part [jet][block][Tk][2D]; no atomics.
// ca_attn_to_bwd_sum_kernel adds the blocks of a jet.  dynamic LDS: (Tk * 2D + rpb * heads * 2 * TK) floats
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_to_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             const float* __restrict__ dO, float* __restrict__ dq,
                                                             float* __restrict__ part, int N, int D, int heads, int Tk, int rpb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // kv [Tk][2D] | pd [rpb][heads][2 TK] (p | ds)
    float* pd = lds + Tk * 2 * D;
    const int jet = blockIdx.x, r0 = blockIdx.y * rpb, nr = min(rpb, N - r0);
    for (int i = threadIdx.x; i < Tk * 2 * D; i += 256) lds[i] = kv[(int64_t)jet * Tk * 2 * D + i];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    for (int idx = threadIdx.x; idx < nr * heads; idx += 256) {
        const int rl = idx / heads, h = idx % heads;
        const int64_t e = ((int64_t)jet * N + r0 + rl) * D + h * HD;
        float qr[HD], dor[HD];
        load_row<HD>(qr, q + e);
        load_row<HD>(dor, dO + e);
        float s[TK], dp[TK], mx = -__builtin_inff();
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            s[tk] = -__builtin_inff();
            dp[tk] = 0.f;
            if (tk < Tk) {
                float kr[HD], vr[HD];
                load_row<HD>(kr, lds + tk * 2 * D + h * HD);
                load_row<HD>(vr, lds + tk * 2 * D + D + h * HD);
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) { a = fmaf(qr[d], kr[d], a); b = fmaf(dor[d], vr[d], b); }
                s[tk] = a * scale;
                dp[tk] = b;
            }
            mx = fmaxf(mx, s[tk]);
        }
        float l = 0.f, p[TK], ds[TK];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) { p[tk] = tk < Tk ? __expf(s[tk] - mx) : 0.f; l += p[tk]; }
        const float inv = 1.0f / l;
        float delta = 0.f;
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) { p[tk] *= inv; delta = fmaf(p[tk], dp[tk], delta); }
        float dqr[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dqr[d] = 0.f;
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            ds[tk] = p[tk] * (dp[tk] - delta) * scale;
            if (tk < Tk) {
                float kr[HD];
                load_row<HD>(kr, lds + tk * 2 * D + h * HD);
#pragma unroll
                for (int d = 0; d < HD; ++d) dqr[d] = fmaf(ds[tk], kr[d], dqr[d]);
            }
        }
        store_row<HD>(dq + e, dqr);
        store_row<TK>(pd + (rl * heads + h) * 2 * TK, p);
        store_row<TK>(pd + (rl * heads + h) * 2 * TK + TK, ds);
    }
    __syncthreads();
    float* out = part + ((int64_t)jet * gridDim.y + blockIdx.y) * Tk * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        const int h = c / HD;
        float dk[TK], dv[TK];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) { dk[tk] = 0.f; dv[tk] = 0.f; }
        const float* qc = q + ((int64_t)jet * N + r0) * D + c;
        const float* doc = dO + ((int64_t)jet * N + r0) * D + c;
#pragma unroll 4
        for (int rl = 0; rl < nr; ++rl) {
            const float qv = qc[(int64_t)rl * D], dov = doc[(int64_t)rl * D];
            float pr[TK], dsr[TK];
            load_row<TK>(pr, pd + (rl * heads + h) * 2 * TK);
            load_row<TK>(dsr, pd + (rl * heads + h) * 2 * TK + TK);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) { dk[tk] = fmaf(dsr[tk], qv, dk[tk]); dv[tk] = fmaf(pr[tk], dov, dv[tk]); }
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
            if (tk < Tk) { out[tk * 2 * D + c] = dk[tk]; out[tk * 2 * D + D + c] = dv[tk]; }
    }
}

// dkv [jet][Tk * 2D] = sum over the nblk row blocks of part [jet][nblk][Tk * 2D]
static __global__ void ca_attn_to_bwd_sum_kernel(const float* __restrict__ part, float* __restrict__ dkv, int64_t n, int per_jet, int nblk) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t jet = i / per_jet, e = i - jet * per_jet;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(jet * nblk + b) * per_jet + e];
    dkv[i] = s;
}

// tok[jet][k][:] = global_tokens[k][:]   (droid_transformer.py:465)
static __global__ void ca_tokens_init_kernel(const float* __restrict__ blob, int64_t off, float* __restrict__ tok, int64_t n, int per_jet) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tok[i] = blob[off + i % per_jet];
}

}  // namespace ca
}  // namespace pfm
