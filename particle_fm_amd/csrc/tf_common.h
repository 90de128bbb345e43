// Shared pieces of the Full-Transformer kernels (gfx950): workspace layout, launch geometry, small device helpers.
#pragma once
#include <stdlib.h>

#include <mutex>

#include "pfm_common.h"
#include "pfm_tf.h"

namespace pfm {
namespace tf {

constexpr int BM = 64;    // rows (particles) per workgroup tile of the Linear kernels
constexpr int BN = 128;   // outputs per workgroup tile: 4 waves x 32
constexpr int LT = 256;   // threads of a Linear workgroup (one wave per SIMD; two workgroups share a CU)
constexpr int HD = 16;    // head_dim this build is specialised for
constexpr int MAXK = 512; // widest Linear input

// Float offsets of the activation workspace.  Rows = n_jets * n_points.  In the train layout every layer owns
// its buffers (lstride > 0) and the residual stream is written out of place; at inference the layers share
// one set and the stream is updated in place.
struct Ws {
    int64_t temb, chid, ctxt, jb;  // per jet: [T] | [CH] (post-activation, pre-norm) | [CO] | [(layers+2)][hidden]
    int64_t h1;                    // rows x hidden: node_embd hidden (post-activation)
    int64_t x0;                    // rows x D: input of layer 0
    int64_t layer0, lstride;       // per layer: qkv | att | xmid | dh | xout
    int64_t o_qkv, o_att, o_xmid, o_dh, o_xout;
    int64_t oh;                    // rows x hidden: outp_embd hidden
    int64_t imaps;                 // int32 row maps of the valid-rows-only evaluation (tf_fwd.h: row_maps_ints)
    int64_t total;
};

// compute units of the current device (launch geometry: Linear row-tile height, dW splits)
inline int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Two half-batches on two streams: the jets of a sampling call are independent, so the latency-bound stretches of one half
// (per-jet GEMMs, context path, global tokens: a few workgroups each) run under the particle-row GEMMs of the other.
// One side stream + fork / join events per device, created on first use and kept for the life of the process.
// A HIP stream is bound to one of a few hardware queues when it is created (least-referenced queue, first one on a tie) and two
// streams on one queue run strictly one after the other, so "two new streams" are not automatically two queues: which queue a
// stream gets depends on every stream the process made before.  streams_overlap() measures it with a one-thread spin kernel.
static __global__ void spin_kernel(long long ticks) {  // ticks of the 100 MHz constant clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
inline bool streams_overlap(hipStream_t a, hipStream_t b) {
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) return false;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, 100LL);  // the hardware queue is set up at the first launch
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, b, 100LL);
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    bool ok = false;
    // serial streams can never look concurrent (b's kernel starts after a's has ended); concurrent ones may look serial once in a
    // while (another process on the card): best of three
    for (int attempt = 0; attempt < 3 && !ok; ++attempt) {
        hipEventRecord(e0, a);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, 30000LL);  // 0.3 ms
        hipEventRecord(e1, a);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, b, 30000LL);
        hipEventRecord(e2, b);
        float ta = 0.f, tab = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventSynchronize(e2) != hipSuccess) break;
        if (hipEventElapsedTime(&ta, e0, e1) != hipSuccess || hipEventElapsedTime(&tab, e0, e2) != hipSuccess) break;
        ok = tab < 1.5f * ta;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipEventDestroy(e2);
    (void)hipGetLastError();
    return ok;
}

struct SideStream {
    hipStream_t s = nullptr, s2 = nullptr;  // two streams measured to run side by side (the caller's stream only forks / joins)
    hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
};
// One PRIVATE pair of side streams per CALLER stream (up to SIDE_PAIRS per device): two sampler calls in flight from different caller
// streams (generate_data's batch pipeline, bench_secondary --overlap) then run side by side instead of queueing on one pair.  A pair and
// its fork / join events are never shared between caller streams: with a shared pair, one call's side streams could wait on the OTHER
// caller's fork record and start before their own caller stream's earlier work (z H2D copy, blob pack) has finished.  A caller stream
// beyond SIDE_PAIRS gets nullptr: its call runs unsplit on its own stream.  Creation is serialised by a mutex (two host threads may enter
// a sampler at once; two threads on the SAME caller stream are the caller's race, as with any stream).
constexpr int SIDE_PAIRS = 4;
inline SideStream* side_stream(hipStream_t caller) {
    struct PerDev {
        SideStream pair[SIDE_PAIRS];
        hipStream_t owner[SIDE_PAIRS] = {};
        int n = 0;
    };
    static PerDev st[16];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    PerDev& pd = st[dev];
    for (int i = 0; i < pd.n; ++i)
        if (pd.owner[i] == caller) return &pd.pair[i];
    if (pd.n == SIDE_PAIRS) return nullptr;  // no private pair left: single-stream path
    SideStream e;
    {
        // candidates are kept alive until both choices are made (a destroyed stream's queue would be dealt to the next one again);
        // s must overlap with the null stream (where callers without a stream of their own run their other work), s2 with both
        hipStream_t cand[8] = {};
        int n = 0, a = -1, b = -1;
        for (; n < 8 && b < 0; ++n) {
            if (hipStreamCreateWithFlags(&cand[n], hipStreamNonBlocking) != hipSuccess) { cand[n] = nullptr; break; }
            if (!streams_overlap(nullptr, cand[n])) continue;
            if (a < 0) a = n;
            else if (streams_overlap(cand[a], cand[n])) b = n;
        }
        // fewer independent queues than hoped for: still correct, just not concurrent
        if (a < 0 && cand[0]) a = 0;
        if (b < 0)
            for (int i = 0; i < 8 && b < 0; ++i)
                if (cand[i] && i != a) b = i;
        for (int i = 0; i < 8; ++i)
            if (cand[i] && i != a && i != b) hipStreamDestroy(cand[i]);
        e.s = a >= 0 ? cand[a] : nullptr;
        e.s2 = b >= 0 ? cand[b] : nullptr;
        if (!e.s || !e.s2 || hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e.join2, hipEventDisableTiming) != hipSuccess) {
            if (e.s) hipStreamDestroy(e.s);  // nothing half-made is kept (or leaked)
            if (e.s2) hipStreamDestroy(e.s2);
            if (e.fork) hipEventDestroy(e.fork);
            if (e.join) hipEventDestroy(e.join);
            if (e.join2) hipEventDestroy(e.join2);
            (void)hipGetLastError();
            return nullptr;
        }
    }
    pd.pair[pd.n] = e;
    pd.owner[pd.n] = caller;
    return &pd.pair[pd.n++];
}
// the caller's stream waits for whatever the two side streams have queued so far (the normal end of a call, and every error return
// after the fork: the caller stream must never run ahead of half-finished side work)
inline void side_join(SideStream* ss, hipStream_t caller) {
    if (!ss) return;
    hipEventRecord(ss->join, ss->s);
    hipEventRecord(ss->join2, ss->s2);
    hipStreamWaitEvent(caller, ss->join, 0);
    hipStreamWaitEvent(caller, ss->join2, 0);
}
// ---- graph replay of a sampler's step body (PFM_CA_F_GRAPH_STEPS) ------------------------------------------------------------
// One midpoint step of the row-matrix models is hundreds of launches of 5-20 us and the host needs ~10 us to enqueue each, so
// the enqueueing thread, not the GPU, sets the pace once two calls are in flight (cross-attention: 225 of 276 ms per call).  The
// step body is the same for every k except for three scalars, so it is captured once per call with those scalars behind fixed
// addresses -- slot = {t_eval[2k], t_eval[2k+1], dt[k]}, refreshed by this one-thread kernel at the head of the body from a
// device-side step counter (slot[4], as int) -- and the captured graph is replayed for the remaining steps.
static __global__ void step_args_kernel(const float* __restrict__ t_eval, const float* __restrict__ dt, float* __restrict__ slot) {
    int* counter = reinterpret_cast<int*>(slot + 4);
    const int k = *counter;
    slot[0] = t_eval[2 * k];
    slot[1] = t_eval[2 * k + 1];
    slot[2] = dt[k];
    *counter = k + 1;
}
constexpr int STEP_SLOT_FLOATS = 64;
// The executable graph of a call must outlive its launches, and the call must not wait for them (the caller goes on enqueueing
// other streams): the graph is parked in a small per-stream ring together with an event recorded behind its last launch; a later
// call on that stream retires the oldest entry (complete long ago unless the caller runs more than two calls ahead).
struct ParkedGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;
};
struct GraphRing {
    hipStream_t s = nullptr;
    ParkedGraph ring[2];
    int next = 0;
};
struct GraphRings {
    std::mutex m;
    GraphRing r[16];
};
inline GraphRings& graph_rings() {
    static GraphRings g;
    return g;
}
// a free entry for a graph launched on stream s (its previous occupant retired), or nullptr: no graph this call
inline ParkedGraph* park_graph(hipStream_t s) {
    GraphRings& G = graph_rings();
    std::lock_guard<std::mutex> lock(G.m);
    GraphRing* g = nullptr;
    for (auto& e : G.r)
        if (e.s == s) g = &e;
    if (!g)
        for (auto& e : G.r)
            if (!e.s && !g) g = &e;
    if (!g) return nullptr;
    g->s = s;
    ParkedGraph& pg = g->ring[g->next];
    g->next ^= 1;
    if (!pg.done && hipEventCreateWithFlags(&pg.done, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (pg.exec) {
        (void)hipEventSynchronize(pg.done);
        (void)hipGraphExecDestroy(pg.exec);
        pg.exec = nullptr;
    }
    if (pg.graph) {
        (void)hipGraphDestroy(pg.graph);
        pg.graph = nullptr;
    }
    return &pg;
}

// jets of the first half (0: do not split).  PFM_SPLIT_STREAMS=0 turns the split off (diagnostics).
inline int split_point(int n_jets, int min_half) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("PFM_SPLIT_STREAMS");
        on = e ? atoi(e) : 1;
    }
    return (on && n_jets >= 2 * min_half) ? n_jets / 2 : 0;
}
__host__ __device__ inline int64_t round64(int64_t x) { return (x + 63) & ~(int64_t)63; }

__host__ inline Ws make_ws(const pfm_tf_desc& d, int n_jets, bool train) {
    Ws w;
    const int64_t M = (int64_t)n_jets * d.n_points, D = d.model_dim, Hd = d.hidden;
    int64_t o = 0;
    w.temb = o; o += round64((int64_t)n_jets * 64);
    w.chid = o; o += round64((int64_t)n_jets * d.ctxt_hidden);
    w.ctxt = o; o += round64((int64_t)n_jets * d.ctxt_dim);
    w.jb = o; o += round64((int64_t)n_jets * (d.layers + 2) * Hd);
    w.h1 = o; o += round64(M * Hd);
    w.x0 = o; o += round64(M * D);
    w.layer0 = o;
    w.o_qkv = 0;
    w.o_att = round64(M * 3 * D);
    w.o_dh = w.o_att + round64(M * D);
    if (train) {
        w.o_xmid = w.o_dh + round64(M * Hd);
        w.o_xout = w.o_xmid + round64(M * D);
        w.lstride = w.o_xout + round64(M * D);
        o += w.lstride * d.layers;
    } else {
        w.o_xmid = w.o_xout = w.x0 - w.layer0;  // in place
        w.lstride = 0;
        o += w.o_dh + round64(M * Hd);
    }
    w.oh = o; o += round64(M * Hd);
    w.imaps = o; o += round64(3 * (int64_t)n_jets + 64 + 2 * M);
    w.total = o;
    return w;
}

// 16 lanes of a DPP row cooperate on one matrix row: lane pl holds columns 4*pl + 64*i.
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, dpp_move<CTRL>(v));
}

__device__ __forceinline__ float hsum4(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }

}  // namespace tf
}  // namespace pfm
