// Attention kernels of the cross-attention encoder (droid_transformer.py:442-472): a handful of global tokens attend to
// all particles of the jet (key mask), then every particle attends to the tokens.  Both shapes are tiny next to the
// Linears (2 * 2 * N * tokens * D FLOP per jet and layer), so they run on the VALU, one workgroup per jet:
//   *_from_* : wave per head, lanes over the keys (particles); softmax statistics by wave reductions
//   *_to_*   : thread per (particle, head); the jet's token keys / values sit in LDS
// HD = head_dim (8 in fm_droid_crossattention.yaml, 16 supported), TK = compile-time bound on the number of tokens.
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

// ---- tokens <- particles ---------------------------------------------------------------------------
// q [n_jets*Tk][D]; kv [n_jets*N][2D] (k | v); mask [n_jets][N] or nullptr; out [n_jets*Tk][D]
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_from_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                           const float* __restrict__ mask, float* __restrict__ out, int N,
                                                           int D, int heads, int Tk) {
    const int jet = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float scale = 1.0f / sqrtf((float)HD);
    for (int h = w; h < heads; h += 4) {
        float qr[TK][HD];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            if (tk < Tk) load_row<HD>(qr[tk], q + ((int64_t)jet * Tk + tk) * D + h * HD);
            else {
#pragma unroll
                for (int d = 0; d < HD; ++d) qr[tk][d] = 0.f;
            }
        }
        float m[TK];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) m[tk] = -__builtin_inff();
        for (int n = lane; n < N; n += 64) {
            if (mask && mask[(int64_t)jet * N + n] == 0.f) continue;
            float kr[HD];
            load_row<HD>(kr, kv + ((int64_t)jet * N + n) * 2 * D + h * HD);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
                m[tk] = fmaxf(m[tk], s * scale);
            }
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) m[tk] = wmax(m[tk]);
        float l[TK], o[TK][HD];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            l[tk] = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[tk][d] = 0.f;
        }
        for (int n = lane; n < N; n += 64) {
            if (mask && mask[(int64_t)jet * N + n] == 0.f) continue;
            float kr[HD], vr[HD];
            const float* row = kv + ((int64_t)jet * N + n) * 2 * D + h * HD;
            load_row<HD>(kr, row);
            load_row<HD>(vr, row + D);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
                const float p = __expf(s * scale - m[tk]);
                l[tk] += p;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[tk][d] = fmaf(p, vr[d], o[tk][d]);
            }
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            l[tk] = wsum(l[tk]);
#pragma unroll
            for (int d = 0; d < HD; ++d) o[tk][d] = wsum(o[tk][d]);
        }
        // lane (tk, d) writes one element; all keys masked: m = -inf, l = 0 -> NaN like torch's softmax
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
#pragma unroll
            for (int d = 0; d < HD; ++d)
                if (lane == tk * HD + d && tk < Tk) {
                    const float mm = m[tk];
                    out[((int64_t)jet * Tk + tk) * D + h * HD + d] = (mm == -__builtin_inff()) ? __builtin_nanf("") : o[tk][d] / l[tk];
                }
    }
}

// backward: dq [n_jets*Tk][D], dkv [n_jets*N][2D] (every row written: masked keys get 0)
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_from_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                               const float* __restrict__ mask, const float* __restrict__ O,
                                                               const float* __restrict__ dO, float* __restrict__ dq,
                                                               float* __restrict__ dkv, int N, int D, int heads, int Tk) {
    const int jet = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float scale = 1.0f / sqrtf((float)HD);
    for (int h = w; h < heads; h += 4) {
        float qr[TK][HD], dor[TK][HD], delta[TK];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            delta[tk] = 0.f;
            if (tk < Tk) {
                const int64_t e = ((int64_t)jet * Tk + tk) * D + h * HD;
                load_row<HD>(qr[tk], q + e);
                load_row<HD>(dor[tk], dO + e);
                float orow[HD];
                load_row<HD>(orow, O + e);
#pragma unroll
                for (int d = 0; d < HD; ++d) delta[tk] = fmaf(dor[tk][d], orow[d], delta[tk]);
            } else {
#pragma unroll
                for (int d = 0; d < HD; ++d) { qr[tk][d] = 0.f; dor[tk][d] = 0.f; }
            }
        }
        float m[TK], l[TK];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) { m[tk] = -__builtin_inff(); l[tk] = 0.f; }
        for (int n = lane; n < N; n += 64) {
            if (mask && mask[(int64_t)jet * N + n] == 0.f) continue;
            float kr[HD];
            load_row<HD>(kr, kv + ((int64_t)jet * N + n) * 2 * D + h * HD);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
                m[tk] = fmaxf(m[tk], s * scale);
            }
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) m[tk] = wmax(m[tk]);
        for (int n = lane; n < N; n += 64) {
            if (mask && mask[(int64_t)jet * N + n] == 0.f) continue;
            float kr[HD];
            load_row<HD>(kr, kv + ((int64_t)jet * N + n) * 2 * D + h * HD);
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s = fmaf(qr[tk][d], kr[d], s);
                l[tk] += __expf(s * scale - m[tk]);
            }
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) l[tk] = 1.0f / wsum(l[tk]);
        float dqa[TK][HD];
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
#pragma unroll
            for (int d = 0; d < HD; ++d) dqa[tk][d] = 0.f;
        for (int n = lane; n < N; n += 64) {
            float* grow = dkv + ((int64_t)jet * N + n) * 2 * D + h * HD;
            float dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
            if (!(mask && mask[(int64_t)jet * N + n] == 0.f)) {
                float kr[HD], vr[HD];
                const float* row = kv + ((int64_t)jet * N + n) * 2 * D + h * HD;
                load_row<HD>(kr, row);
                load_row<HD>(vr, row + D);
#pragma unroll
                for (int tk = 0; tk < TK; ++tk) {
                    float s = 0.f, dp = 0.f;
#pragma unroll
                    for (int d = 0; d < HD; ++d) { s = fmaf(qr[tk][d], kr[d], s); dp = fmaf(dor[tk][d], vr[d], dp); }
                    const float p = __expf(s * scale - m[tk]) * l[tk];
                    const float ds = p * (dp - delta[tk]) * scale;
#pragma unroll
                    for (int d = 0; d < HD; ++d) {
                        dqa[tk][d] = fmaf(ds, kr[d], dqa[tk][d]);
                        dk[d] = fmaf(ds, qr[tk][d], dk[d]);
                        dv[d] = fmaf(p, dor[tk][d], dv[d]);
                    }
                }
            }
            store_row<HD>(grow, dk);
            store_row<HD>(grow + D, dv);
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float s = wsum(dqa[tk][d]);
                if (lane == tk * HD + d && tk < Tk) dq[((int64_t)jet * Tk + tk) * D + h * HD + d] = s;
            }
    }
}

// ---- particles <- tokens ---------------------------------------------------------------------------
// q [n_jets*N][D]; kv [n_jets*Tk][2D]; out [n_jets*N][D]; no key mask (droid_transformer.py:470)
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_to_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                         float* __restrict__ out, int N, int D, int heads, int Tk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [Tk][2D]
    const int jet = blockIdx.x;
    for (int i = threadIdx.x; i < Tk * 2 * D; i += 256) lds[i] = kv[(int64_t)jet * Tk * 2 * D + i];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    for (int idx = threadIdx.x; idx < N * heads; idx += 256) {
        const int r = idx / heads, h = idx - r * heads;
        const int64_t e = ((int64_t)jet * N + r) * D + h * HD;
        float qr[HD];
        load_row<HD>(qr, q + e);
        float s[TK], mx = -__builtin_inff();
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            s[tk] = -__builtin_inff();
            if (tk < Tk) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a = fmaf(qr[d], lds[tk * 2 * D + h * HD + d], a);
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
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] = fmaf(p, lds[tk * 2 * D + D + h * HD + d], o[d]);
            }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= inv;
        store_row<HD>(out + e, o);
    }
}

// backward: dq [n_jets*N][D]; dkv [n_jets*Tk][2D] (sum over the jet's particles, accumulated in LDS)
template <int HD, int TK>
__global__ __launch_bounds__(256) void ca_attn_to_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             const float* __restrict__ dO, float* __restrict__ dq,
                                                             float* __restrict__ dkv, int N, int D, int heads, int Tk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // kv [Tk][2D] | dkv accumulators [Tk][2D]
    float* acc = lds + Tk * 2 * D;
    const int jet = blockIdx.x;
    for (int i = threadIdx.x; i < Tk * 2 * D; i += 256) {
        lds[i] = kv[(int64_t)jet * Tk * 2 * D + i];
        acc[i] = 0.f;
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    // thread (head h, row group g): every thread of a head walks a different subset of the rows
    const int h = threadIdx.x % heads, g = threadIdx.x / heads, ng = 256 / heads;
    float dk[TK][HD], dv[TK][HD];
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
        for (int d = 0; d < HD; ++d) { dk[tk][d] = 0.f; dv[tk][d] = 0.f; }
    if (g < ng) {
        for (int r = g; r < N; r += ng) {
            const int64_t e = ((int64_t)jet * N + r) * D + h * HD;
            float qr[HD], dor[HD];
            load_row<HD>(qr, q + e);
            load_row<HD>(dor, dO + e);
            float s[TK], dp[TK], mx = -__builtin_inff();
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                s[tk] = -__builtin_inff();
                dp[tk] = 0.f;
                if (tk < Tk) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int d = 0; d < HD; ++d) {
                        a = fmaf(qr[d], lds[tk * 2 * D + h * HD + d], a);
                        b = fmaf(dor[d], lds[tk * 2 * D + D + h * HD + d], b);
                    }
                    s[tk] = a * scale;
                    dp[tk] = b;
                }
                mx = fmaxf(mx, s[tk]);
            }
            float l = 0.f, p[TK];
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
            for (int tk = 0; tk < TK; ++tk)
                if (tk < Tk) {
                    const float ds = p[tk] * (dp[tk] - delta) * scale;
#pragma unroll
                    for (int d = 0; d < HD; ++d) {
                        dqr[d] = fmaf(ds, lds[tk * 2 * D + h * HD + d], dqr[d]);
                        dk[tk][d] = fmaf(ds, qr[d], dk[tk][d]);
                        dv[tk][d] = fmaf(p[tk], dor[d], dv[tk][d]);
                    }
                }
            store_row<HD>(dq + e, dqr);
        }
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
            if (tk < Tk) {
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    atomicAdd(acc + tk * 2 * D + h * HD + d, dk[tk][d]);
                    atomicAdd(acc + tk * 2 * D + D + h * HD + d, dv[tk][d]);
                }
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tk * 2 * D; i += 256) dkv[(int64_t)jet * Tk * 2 * D + i] = acc[i];
}

// tok[jet][k][:] = global_tokens[k][:]   (droid_transformer.py:465)
static __global__ void ca_tokens_init_kernel(const float* __restrict__ blob, int64_t off, float* __restrict__ tok, int64_t n, int per_jet) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tok[i] = blob[off + i % per_jet];
}

}  // namespace ca
}  // namespace pfm
