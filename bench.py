#!/usr/bin/env python3
"""Headline benchmark: jets/sec of EPiC-FM JetNet N=150 -- one training step plus one 100-step midpoint
ODE sample per "step", on N MI355X of one node (BASELINE.json `metric`; workload = BASELINE `configs[2]`,
the configuration the metric is quoted on: N=150, F=3, H=128, L=10, 6 EPiC layers, batch 256 per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step, per GPU (SURVEY.md §8d definitions):
  (T) train step on 256 jets: FM-OT loss forward + hand-written backward (HIP), flat gradient all-reduce (RCCL,
      N>1), clip_grad_norm 0.5 + AdamW(1e-3, wd 5e-5) + EMA 0.999 (HIP);
  (S) decode of 256 jets: z -> x by 99 explicit-midpoint intervals = 198 network evaluations (HIP, one launch).
Inputs are synthetic (x ~ N(0,1)*mask, multiplicities U{30..150}, seeds 12345 / 9999), resident in HBM before
the timed region.  value = (jets per step over all ranks) / (max-over-ranks step time); weak scaling.
The JSON line also carries `roofline` (dominant kernel = the persistent sampler, against the fp32 MFMA peak) and,
at N=1, `cpu_baseline` (the eager-PyTorch oracle on the host cores, on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NFE_FLOP_PER_JET = 84.22e6     # algorithmic fwd FLOP / jet, dense over the padded N=150
FP32_MFMA_PEAK = 157.3e12      # MI355X_MICROARCH.md: FP32 matrix peak (v_mfma_f32_16x16x4_f32)
# profiles/: rocprofv3 --pmc passes of this command, newest first (tests/diag/collect_pmc_sq.sh: SQ / GRBM / TCC counters per launch;
# tests/diag/collect_bench_profiles.sh: FETCH_SIZE / WRITE_SIZE only)
PMC_SUMMARIES = ("round4_pmc_summary.json", "round3_pmc_summary.json", "round2_fast_pmc_hbm_summary.json", "round2_pmc_hbm_summary.json", "round1_pmc_hbm_summary.json")
HP = dict(model="epic", features=3, hidden_dim=128, num_particles=150, frequencies=16, layers=6, latent=10,
          activation="leaky_relu", wrapper_func="weight_norm", t_local_cat=True, t_global_cat=True,
          add_time_to_input=False, t_emb="cosine", loss_type="FM-OT", sigma=1e-4, global_cond_dim=0,
          local_cond_dim=0, dropout=0.0, sum_scale=1e-2)  # configs/model/flow_matching.yaml + fm_tops150.yaml


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


T_START = time.perf_counter()


class stage:
    """`with stage("rendezvous", 180): ...` -- logs entry / exit per rank and, if the block has not finished after `limit` seconds
    (a rank that never arrives, a collective that hangs), prints what it was waiting in and ends the PROCESS with a non-zero code:
    torch.distributed.run then tears the other ranks down, and the driver's record shows the stage instead of a silent time-out."""

    def __init__(self, what, limit):
        self.what, self.limit = what, float(os.environ.get("PFM_BENCH_STAGE_TIMEOUT", limit))

    def _expired(self):
        rank = os.environ.get("RANK", "0")
        print(f"[bench +{time.perf_counter() - T_START:7.1f}s] rank {rank}: stage '{self.what}' did not finish within {self.limit:.0f} s "
              "-- giving up (exit 124)", file=sys.stderr, flush=True)
        os._exit(124)

    def __enter__(self):
        import threading
        self.t0 = time.perf_counter()
        self.timer = threading.Timer(self.limit, self._expired)
        self.timer.daemon = True
        self.timer.start()
        log(f"rank {os.environ.get('RANK', '0')}: {self.what} ...")
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        if exc[0] is None:
            log(f"rank {os.environ.get('RANK', '0')}: {self.what} done in {time.perf_counter() - self.t0:.2f} s")
        return False


def synthetic_batch(B, N, F, seed):
    gen = torch.Generator().manual_seed(seed)
    n = torch.randint(30, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).to(torch.int64).unsqueeze(-1)  # JetNet masks are int64
    x = torch.randn(B, N, F, generator=gen) * mask
    return x, mask, torch.zeros(B)


def usable_cores() -> int:
    """Host cores this process may actually run on: affinity mask, capped by the cgroup CPU quota (a GPU box hands
    a 1-GPU job a 16-core share of a 128-thread host; spinning 128 OpenMP threads on it is 100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("PFM_BENCH_CPU_THREADS", "32"))))


def cpu_baseline(state, freqs, ode_steps, jets=256, sample_nfe=20, warmup=3, reps=5):
    """The oracle (eager PyTorch restatement of the reference graph) on the host cores, SURVEY.md §8d protocol:
    the bench batch (256 jets), `warmup` untimed + `reps` timed iterations of each leg, median.  The train leg is the
    full (T) step.  The sample leg integrates `sample_nfe` network evaluations (sample_nfe / 2 midpoint intervals of the
    same t-grid spacing) and is scaled to the 2 (ode_steps - 1) evaluations of the GPU's sample: the cost of an evaluation
    does not depend on t, and the scaling is stated in `sample`."""
    import statistics
    from oracle.fm_ref import EpicVectorField, fm_ot_loss, midpoint_trajectory_end
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"  cpu: {cores} threads (os.cpu_count()={os.cpu_count()})")
    ohp = dict(HP)
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in state.items()}
    params = [v for v in st.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-5)
    x, mask, _ = synthetic_batch(jets, HP["num_particles"], HP["features"], 12345)  # the GPU leg's batch of rank 0
    maskf = mask.float()
    vf = EpicVectorField(st, "flows.0.net", ohp, freqs=freqs)

    def train_step():
        t = torch.rand(jets)
        z = torch.randn_like(x)
        opt.zero_grad()
        loss, *_ = fm_ot_loss(vf, x, maskf, None, t, z, sigma=1e-4)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()

    def timed(fn):
        for _ in range(warmup):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), ts

    t_train, tr_all = timed(train_step)
    log(f"  cpu: train step median {t_train*1e3:.1f} ms over {reps} (after {warmup} warm-up)")
    n_nfe = 2 * (ode_steps - 1)
    n_int = max(1, min(ode_steps - 1, sample_nfe // 2))
    t_span = torch.linspace(1.0, 0.0, ode_steps)[: n_int + 1]  # the first n_int intervals of the reference's grid
    z = torch.randn(jets, HP["num_particles"], HP["features"], generator=torch.Generator().manual_seed(9999)) * maskf
    vf_ng = EpicVectorField({k: v.detach() for k, v in st.items()}, "flows.0.net", ohp, freqs=freqs)

    def sample_leg():
        with torch.no_grad():
            midpoint_trajectory_end(lambda t, xx: vf_ng(t, xx, mask=maskf, cond=None), z, t_span)

    t_part, sm_all = timed(sample_leg)
    scale = n_nfe / (2 * n_int)
    t_sample = t_part * scale
    log(f"  cpu: {2 * n_int} NFE median {t_part*1e3:.1f} ms -> x{scale:.2f} = {t_sample:.2f} s per {n_nfe}-NFE sample")
    return {
        "value": jets / (t_train + t_sample), "unit": "jets/s", "cores": cores, "kind": "port",
        "sample": f"{jets} jets (the bench batch), {warmup} warm-up + {reps} timed iterations per leg, median: (T) full train step "
                  f"(FM-OT fwd+bwd, clip 0.5, AdamW); (S) {2 * n_int} of the {n_nfe} network evaluations of the midpoint sample "
                  f"(first {n_int} intervals of linspace(1,0,{ode_steps})), time SCALED x{scale:.2f} to {n_nfe} NFE; eager-PyTorch "
                  f"oracle, fp32, torch threads = {cores}",
        "train_jets_per_s": jets / t_train, "sample_jets_per_s": jets / t_sample,
        "train_ms_median": 1e3 * t_train, "sample_s_scaled": t_sample, "sample_nfe_timed": 2 * n_int, "sample_scale": scale,
        "train_ms_all": [1e3 * v for v in tr_all], "sample_part_ms_all": [1e3 * v for v in sm_all],
    }


class StepLoop:
    """The bench's step pipeline.  Step i = train step i + sample i with the weights of step i (a snapshot blob, so the next train
    step may update the parameters meanwhile).  With D = overlap > 1 the sample of step i runs on stream i % D while step i+1 trains
    on a stream of its own and the next sample is queued; with D = 1 everything is on the caller's stream.  The results do not
    depend on D (tests/test_hip_bench_pipeline.py compares the parameters and the samples bit for bit)."""

    def __init__(self, model, trainer, batch, z, ode_steps, overlap, dev):
        from particle_fm_amd.utils.streams import concurrent_streams
        self.model, self.trainer, self.batch, self.z, self.ode_steps, self.dev = model, trainer, batch, z, ode_steps, dev
        N = batch[0].shape[1]
        D = self.D = max(1, overlap)
        # D + 1 streams on hardware queues of their own (a stream is bound to a queue at creation and two streams on one queue
        # serialise); with D > 1 the train step runs on the last of them rather than on the default stream
        pool = concurrent_streams(D + 1, dev) if D > 1 else []  # verified by measurement to run side by side (utils/streams.py)
        self.streams = pool[:D] if D > 1 else [torch.cuda.current_stream(dev)]
        self.main = pool[D] if D > 1 else torch.cuda.current_stream(dev)
        if D > 1:
            self.main.wait_stream(torch.cuda.current_stream(dev))
        # D + 1 snapshot slots: train step i may run while samples i-1 .. i-D are still in flight (it only has to wait for sample
        # i-(D+1), the last reader of its slot), so the train chain -- which includes the gradient all-reduce when N > 1 -- has a
        # whole step of slack and a late collective delays nothing
        S = self.S = D + 1
        with torch.cuda.stream(self.main):
            self.snaps = [trainer.snapshot_blob(N) for _ in range(S)]
        self.outs = [None] * S
        self.done = [torch.cuda.Event() for _ in range(S)]
        self.N = N

    def step(self, i, ev=None):
        s, q, main = i % self.S, self.streams[i % self.D], self.main
        x, mask, cond = self.batch
        main.wait_event(self.done[s])          # snapshot slot s is free again (sample i-S has finished)
        with torch.cuda.stream(main):
            if ev:
                ev[0].record(main)
            self.trainer.step((x, mask, cond))
            self.trainer.snapshot_blob(self.N, out=self.snaps[s])
            if ev:
                ev[1].record(main)
        q.wait_stream(main)
        with torch.cuda.stream(q), torch.no_grad():
            if ev:
                ev[2].record(q)
            self.outs[s] = self.model(self.z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=self.ode_steps,
                                      weights=self.snaps[s])
            if ev:
                ev[3].record(q)
            self.done[s].record(q)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="jets per GPU (BASELINE config: 256)")
    ap.add_argument("--ode-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pack-jets", dest="pack_jets", action="store_false",
                    help="sampler: one jet per workgroup.  Default (round 4): two jets share a workgroup where their rows fit the LDS tile "
                         "together (PFM_F_PACK_JETS: the k-th longest jet takes the shortest remaining one that fits, 64 pairs in the bench "
                         "batch; one weight stream, one set of phases and barriers for both; workgroups dispatched by their total tiles) -- "
                         "same results bit for bit (tests/test_hip_packed.py), 20.6 -> 19.9 ms per step")
    ap.add_argument("--overlap", type=int, default=2,
                    help="sampling launches in flight: step i's sample runs on its own HIP stream with a weight snapshot "
                         "while step i+1 trains and the next sample starts (default 2: +13 %% jets/s on one MI355X -- a launch lasts "
                         "as long as its largest jet, the CUs that finish early pick up the next launch's jets; 1 = strictly "
                         "sequential, one launch at a time)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    # PFM_BENCH_REHEARSAL=gloo: rehearse the N > 1 control flow on a box with fewer GPUs than ranks (ranks share the cards, the
    # gradient all-reduce goes through gloo); never set by the driver, and the line it prints says so in `config`
    rehearsal = os.environ.get("PFM_BENCH_REHEARSAL", "")
    if rehearsal not in ("", "gloo"):
        raise SystemExit(f"PFM_BENCH_REHEARSAL={rehearsal!r}: only 'gloo' is known")
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        with stage(f"rendezvous ({'gloo rehearsal' if rehearsal else 'RCCL'}, world {world}, MASTER {os.environ.get('MASTER_ADDR')}:"
                   f"{os.environ.get('MASTER_PORT')})", 300):
            if rehearsal:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))  # RCCL over xGMI
        with stage("first all-reduce (communicator set-up)", 300):
            probe = torch.ones(4, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev)
            if float(probe[0].item()) != float(world):
                raise SystemExit(f"rank {rank}: all-reduce of ones gave {float(probe[0].item())}, expected {world}")

    from particle_fm_amd.engine import FusedFMTrainer
    from particle_fm_amd.models import SetFlowMatchingLitModule

    torch.manual_seed(12345)  # fm_tops150.yaml:19 -- identical replicas on every rank
    model = SetFlowMatchingLitModule(optimizer=None, **HP).to(dev)
    state_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith("flows.")}
    trainer = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    if args.pack_jets:
        model.flows[0].net.set_jet_packing(True)
    B, N, F = args.batch, HP["num_particles"], HP["features"]
    x, mask, cond = synthetic_batch(B, N, F, 12345 + rank)
    x, mask, cond = x.to(dev), mask.to(dev), cond.to(dev)
    gz = torch.Generator().manual_seed(9999 + rank)  # jetnet_eval.py:146
    z = (torch.randn(B, N, F, generator=gz) * mask.cpu()).to(dev)  # sample(): CPU draw, masked (:659-671)
    n_nfe = 2 * (args.ode_steps - 1)

    loop = StepLoop(model, trainer, (x, mask, cond), z, args.ode_steps, args.overlap, dev)
    D, S, main, snaps, outs, step = loop.D, loop.S, loop.main, loop.snaps, loop.outs, loop.step

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"rank {rank}/{world}: model + data ready, {args.warmup} warm-up steps, overlap depth {D}")
    with stage(f"warm-up ({args.warmup} steps; the first one loads the kernels and, N > 1, runs the first gradient all-reduce on the "
               "train stream)", 600):
        for i in range(args.warmup):
            step(i)
            if i == 0:
                torch.cuda.synchronize(dev)
                log(f"rank {rank}: first step done")
        torch.cuda.synchronize(dev)
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    with stage(f"timed region ({args.steps} steps)", 1200):
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, ev[i])
        fence()
        elapsed = time.perf_counter() - t0
    log(f"rank {rank}: timed {args.steps} steps in {elapsed:.3f}s")
    out = outs[(args.steps - 1) % S]
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    def check_replicas(where):
        """data-parallel invariant (rehearsal only): identical replicas after synchronised steps (bitwise)"""
        if not (rehearsal and world > 1):
            return
        torch.cuda.synchronize(dev)
        flat = trainer.fp.flat.detach()
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"rank {rank}: parameter replicas diverged {where} (max spread {float((hi - lo).abs().max()):.3e})")

    check_replicas("after the timed region")
    train_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    sample_ms = sum(e[2].elapsed_time(e[3]) for e in ev) / args.steps  # on the stream the sampler was launched on
    assert torch.isfinite(out).all()

    # the dominant kernel alone on the GPU (outside the timed region): with --overlap > 1 the launches of the timed region
    # share the machine, so their individual durations say little about the kernel itself
    excl_ms = sample_ms
    if D > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(main), torch.no_grad():
            e0.record(main)
            for _ in range(3):
                model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=args.ode_steps, weights=snaps[0])
            e1.record(main)
        torch.cuda.synchronize(dev)
        excl_ms = e0.elapsed_time(e1) / 3

    # the train step alone on the GPU (after the timed region; N > 1: includes the gradient all-reduce): inside the timed region
    # its kernels queue behind sampler workgroups for a free CU, so `train_ms` there is mostly waiting
    t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    TRAIN_ALONE_REPS = 20  # (3 were too few: the first step's host-side enqueue time sat in the average, 0.80 instead of 0.65 ms)
    with torch.cuda.stream(main):
        for _ in range(3):
            trainer.step((x, mask, cond))
        t0e.record(main)
        for _ in range(TRAIN_ALONE_REPS):
            trainer.step((x, mask, cond))
        t1e.record(main)
    torch.cuda.synchronize(dev)
    train_alone_ms = t0e.elapsed_time(t1e) / TRAIN_ALONE_REPS
    # N > 1: the train step alone under BOTH gradient exchanges -- one flat all-reduce behind the whole backward (the default) and
    # the two-bucket exchange whose first all-reduce runs next to the dW GEMM (PFM_DP_OVERLAP=1, engine.FusedFMTrainer
    # .fused_loss_and_grad) -- so that a SCALE record separates overlap gain from collective cost (the comparison SURVEY 5.8 asks for)
    train_alone_modes = {}
    if world > 1:
        keep = trainer.split_backward
        for mode, split in (("flat_allreduce", False), ("bucketed_overlap", True)):
            with stage(f"train step alone, gradient exchange: {mode}", 300):
                trainer.split_backward = split
                with torch.cuda.stream(main):
                    for _ in range(3):
                        trainer.step((x, mask, cond))
                    t0e.record(main)
                    for _ in range(TRAIN_ALONE_REPS):
                        trainer.step((x, mask, cond))
                    t1e.record(main)
                torch.cuda.synchronize(dev)
                train_alone_modes[mode] = t0e.elapsed_time(t1e) / TRAIN_ALONE_REPS
            check_replicas(f"after {3 + TRAIN_ALONE_REPS} steps with the {mode} exchange")
        trainer.split_backward = keep
    # the gradient exchange alone (N > 1): the flat 2.2 MB all-reduce on the train stream, 10 in a row
    allreduce_alone_ms = 0.0
    if world > 1:
        with stage("all-reduce timing", 300):
            with torch.cuda.stream(main):
                trainer.sync.sync(trainer.fp.grad)
                t0e.record(main)
                for _ in range(10):
                    trainer.sync.sync(trainer.fp.grad)
                t1e.record(main)
            torch.cuda.synchronize(dev)
            allreduce_alone_ms = t0e.elapsed_time(t1e) / 10
    per_rank = torch.tensor([train_alone_ms, allreduce_alone_ms, train_ms, sample_ms, excl_ms], device=dev, dtype=torch.float64)
    if world > 1:
        gathered = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        per_rank = torch.stack(gathered).cpu()
    else:
        per_rank = per_rank[None].cpu()

    if rank == 0:
        jets_per_step = B * world
        value = jets_per_step * args.steps / elapsed
        # What the matrix cores execute (rank 0's batch; every rank has the same shape of work): the 13 128x128 Linears of an
        # evaluation on the 16-row tiles up to the last valid particle of each jet.  The t / cond / g columns of the reference's
        # concatenated inputs are folded into per-jet biases (GEMVs on the VALU) and fully masked tiles are skipped, so this is
        # LESS than SURVEY 8d's dense count (84.22 MFLOP/jet/NFE); the train step executes 3x the forward (fwd, dX, dW).
        n_valid = mask.sum((1, 2)).cpu()
        rows_exec = int(((n_valid + 15) // 16 * 16).sum().item())
        lin_flop_per_row = 13 * 2 * 128 * 128
        exec_sample = rows_exec * lin_flop_per_row * n_nfe            # per sampler launch
        exec_train = rows_exec * lin_flop_per_row * 3                 # per train step
        dense_sample = B * n_nfe * NFE_FLOP_PER_JET                   # SURVEY 8d algorithmic count per launch
        executed = (exec_sample + exec_train) * args.steps / elapsed  # FLOP/s on the MFMA pipe over the timed wall time
        n_pad = (N + 15) // 16 * 16
        traffic = None
        pmc_file = None
        # which kernel the sampler launches for this descriptor (csrc/epic_fast.h: the lean evaluation of unconditioned jets)
        import ctypes
        from particle_fm_amd import _lib
        lay = model.flows[0].net.layout()
        sampler_kernel = ((f"epic_sample_midpoint_fast_kernel<0, {'true' if args.pack_jets else 'false'}, false>")
                          if _lib.load().pfm_epic_sample_is_fast(ctypes.byref(lay.desc)) else "epic_sample_midpoint_kernel<0, true>")
        mfma_busy, mfma_busy_step, sq_file = None, None, None
        for cand in PMC_SUMMARIES:  # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs, gfx950 correction)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                pmc = pmc["pfm::" + sampler_kernel] if "pfm::" + sampler_kernel in pmc else next(
                    v for k, v in pmc.items() if k.startswith("pfm::" + sampler_kernel.split("<")[0] + "<0"))
                if traffic is None and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    traffic, pmc_file = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0, cand
                if mfma_busy is None and "mfma_busy" in pmc:
                    mfma_busy, sq_file = pmc["mfma_busy"], cand
                    if "cycles" in pmc and pmc.get("duration_ns"):  # busy cycles of one launch over the cycles of one timed step
                        mfma_busy_step = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * pmc["cycles"] / pmc["duration_ns"] * 1e9 * elapsed / args.steps)
            except Exception:
                continue
        res = {
            "metric": "jets/sec (train step + 100-step ODE sample), EPiC-FM JetNet N=150",
            "value": value, "unit": "jets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "EPiC-FM JetNet150 (N=150, F=3, H=128, L=10, 6 EPiC layers, 561330 params): per step "
                            "1 train step (FM-OT fwd+bwd, grad all-reduce, clip 0.5, AdamW, EMA) + 1 midpoint "
                            f"ODE sample (ode_steps={args.ode_steps}, {n_nfe} NFE) on the same number of jets; the sample starts from a "
                            "z that is already resident in HBM (drawn once on the CPU generator and copied before the timed region: "
                            "sample()'s own per-call CPU randn + H2D copy, flow_matching_module.py:659-663, is NOT in the timed "
                            "region -- generate_data, which performs it, reaches the same rate, DESIGN.md section 5)",
                "jets_per_gpu": B, "global_batch": jets_per_step,
                "parallelism": f"dp{world}" + (f" (REHEARSAL: {rehearsal} collectives, ranks share {torch.cuda.device_count()} GPU(s))" if rehearsal else ""),
                "overlap": f"{D} sampling launches in flight (sample of step i on its own HIP stream with a weight "
                           "snapshot while step i+1 trains); every launch of the K steps is inside the timed region",
                "multiplicity": "U{30..150} per jet (masked tail tiles are skipped; results identical)",
                "jets_per_workgroup": "1 or 2 (PFM_F_PACK_JETS: two jets whose rows fit the 150-row LDS tile together share a workgroup; "
                                      "bit-identical results)" if args.pack_jets else "1",
            },
            "train_ms": train_ms, "sample_ms": sample_ms, "train_ms_alone": train_alone_ms, "allreduce_ms_alone": allreduce_alone_ms,
            "grad_exchange": None if world == 1 else {
                "mode": "timed region: " + ("two buckets, the first all-reduce next to the dW GEMM (PFM_DP_OVERLAP=1)" if trainer.split_backward
                                            else "one flat RCCL all-reduce (SUM, then x 1/world in the optimiser kernel) behind the whole "
                                                 "backward (default; PFM_DP_OVERLAP=1 = two buckets, the gradients the backward's chain phase "
                                                 "finishes -- 51 % of the parameters -- reduced next to the dW GEMM of the rest)"),
                "overlapped": bool(trainer.split_backward),
                "train_ms_alone_flat_allreduce": train_alone_modes.get("flat_allreduce"),
                "train_ms_alone_bucketed_overlap": train_alone_modes.get("bucketed_overlap"),
                "two_phase_direct": None},  # filled by the last leg of the run (below)
            "per_rank": {"train_ms_alone": [float(v) for v in per_rank[:, 0]], "allreduce_ms_alone": [float(v) for v in per_rank[:, 1]],
                         "train_ms_in_timed_region": [float(v) for v in per_rank[:, 2]],
                         "sample_ms_in_timed_region": [float(v) for v in per_rank[:, 3]],
                         "sample_ms_alone": [float(v) for v in per_rank[:, 4]],
                         "note": "one entry per rank; train_ms_alone includes the gradient all-reduce when N > 1, allreduce_ms_alone is "
                                 "that collective by itself (flat 2.245 MB fp32 buffer, 10 in a row on the train stream): a SCALE record "
                                 "separates collective cost from compute with these"},
            "train_jets_per_s": B * world / (train_alone_ms * 1e-3), "sample_jets_per_s": B * world / (sample_ms * 1e-3),
            "timing_note": "train_ms / sample_ms: HIP events around the train step / one sampler launch INSIDE the timed region, where "
                           "they share the GPU (the train step's kernels wait for CUs held by sampler workgroups); train_ms_alone and "
                           "roofline.kernel_alone_ms: the same work alone on the GPU after the timed region; train_jets_per_s uses "
                           "train_ms_alone",
            "roofline": {
                "bound": "mfma", "kernel": sampler_kernel,
                "achieved": executed / 1e12, "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": executed / FP32_MFMA_PEAK, "traffic": traffic,
                "executed_flop_per_launch": exec_sample, "executed_flop_per_train_step": exec_train,
                "valid_row_fraction": float(n_valid.sum().item()) / (B * N),
                "executed_row_fraction": rows_exec / float(B * n_pad),
                "kernel_ms_in_timed_region": sample_ms, "kernel_alone_ms": excl_ms, "concurrent_launches": D,
                "algorithmic_dense_flop_per_launch": dense_sample,
                "dense_equiv_over_peak": dense_sample / (excl_ms * 1e-3) / FP32_MFMA_PEAK,
                "mfma_busy": mfma_busy, "mfma_busy_per_timed_step": mfma_busy_step,
                "dense_flop_not_executed_share": 1.0 - exec_sample / dense_sample,
                "note": "frac = achieved / peak with achieved = FLOP the matrix cores EXECUTE (13 Linears of 128x128 per evaluation on "
                        "the 16-row tiles up to each jet's last valid particle; sampler launches + train steps, train = 3x forward) "
                        "over the timed wall time: it cannot exceed 1.  mfma_busy = the hardware's own count for ONE sampler launch alone on "
                        f"the GPU: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of profiles/{sq_file} (rocprofv3 --pmc, "
                        "tests/diag/collect_pmc_sq.sh; null if that file has no SQ pass for this kernel); mfma_busy_per_timed_step (derived, two runs combined) = that launch's "
                        "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the cycles of ONE timed step at the clock of the counter run): the share of "
                        "the matrix pipes' cycles one sampler launch keeps busy when launches overlap as in the timed region (alone, a launch "
                        "with two jets per workgroup occupies 192 of the 256 CUs: the second launch in flight uses the rest); the train "
                        "step's MFMAs are not in it.  "
                        "dense_equiv_over_peak (NOT a "
                        "utilisation: it may exceed 1) = SURVEY 8d's dense count "
                        f"({NFE_FLOP_PER_JET/1e6:.2f} MFLOP/jet/NFE x {n_nfe} NFE x {B} jets: padded N, concatenated t/cond/g columns "
                        "counted) / duration of ONE launch alone on the GPU (kernel_alone_ms) / peak; dense_flop_not_executed_share of "
                        "that count is never executed (masked tail tiles skipped, t/cond/g columns folded into per-jet bias GEMVs on "
                        "the VALU), so the dense figure is a throughput-equivalent, not MFMA utilisation.  traffic = HBM-side bytes per "
                        f"launch, (2*FETCH_SIZE + WRITE_SIZE) KiB of profiles/{pmc_file} (each of the 8 XCD L2s fetches the "
                        "weights and the time-term table once, Infinity-Cache hits included)",
            },
        }
        if world == 1:
            # informational, outside the timed region and NOT part of `value`: the same sampler launch with bf16 MFMA
            # operands (PFM_F_BF16_MFMA; fp32 accumulate and activations), and how far its result is from the fp32 one
            net = model.flows[0].net
            with torch.no_grad():
                ref32 = model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=args.ode_steps)
                for prec, key, what in (
                    ("f16x3", "f16x3_mfma_sampler",
                     "particle Linears as three v_mfma_f32_16x16x32_f16 on (hi, lo) fp16 splits of both operands (22 "
                     "significant bits each), fp32 accumulate / activations in two fp16 planes; passes the SAME fp32 parity "
                     "tests as the fp32-MFMA kernel (tests/test_hip_f16x3.py)"),
                    ("bf16", "bf16_mfma_sampler",
                     "particle Linears AND the per-jet chains on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32 accumulate, fp32 "
                     "activations; what the reference's bf16-mixed precision does to every nn.Linear); tests/test_hip_bf16.py bounds it by "
                     "the reference's autocast-bf16 error"),
                ):
                    net.set_precision(prec)
                    o = model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=args.ode_steps)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        o = model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=args.ode_steps)
                    e1.record()
                    torch.cuda.synchronize(dev)
                    ms = e0.elapsed_time(e1) / 3
                    res[key] = {"sample_ms": ms, "sample_jets_per_s": B / (ms * 1e-3),
                                "max_abs_dev_from_f32": float((o - ref32).abs().max()),
                                "note": "informational, outside the timed region, NOT part of `value`: the same sampler launch, " + what}
                net.set_precision("fp32")
        if world == 1 and not args.no_cpu_baseline:
            freqs = model.flows[0].net.layout().default_freqs()
            log("cpu baseline (oracle on the host cores) ...")
            res["cpu_baseline"] = cpu_baseline(state_cpu, freqs, args.ode_steps)
            log("cpu baseline done")
    # N > 1, LAST leg of the run, informational: the direct two-phase gradient exchange (engine.GradSync.two_phase_sum: all-to-all of 1/N
    # slices over the point-to-point links, local sum, all-gather; SURVEY 5.8 / 8e's comparison against RCCL's ring at 2.2 MB) -- the
    # exchange alone and the train step alone with it.  It has never run on more than one RCCL rank, so it comes behind everything the
    # line depends on and cannot cost the line: an exception is recorded in it, and if the leg has not finished after 90 s rank 0 prints
    # the line without it and every rank ends with exit code 0.
    if world > 1:
        import threading
        two_phase = {"train_ms_alone": None, "exchange_ms_alone": None, "error": None,
                     "note": "informational, outside the timed region: all-to-all of 1/N slices + local sum + all-gather "
                             "(PFM_DP_EXCHANGE=two_phase) instead of the ring all-reduce; exchange_ms_alone compares with "
                             "allreduce_ms_alone, train_ms_alone with grad_exchange.train_ms_alone_flat_allreduce"}

        def give_up():
            if rank == 0:
                two_phase["error"] = "did not finish within 90 s"
                res["grad_exchange"]["two_phase_direct"] = two_phase
                print(json.dumps(res), flush=True)
            os._exit(0)

        guard = threading.Timer(float(os.environ.get("PFM_BENCH_TWO_PHASE_TIMEOUT", 90)), give_up)
        guard.daemon = True
        guard.start()
        keep_x = trainer.sync.exchange
        try:
            log(f"rank {rank}: two-phase exchange timing ...")
            trainer.sync.exchange = "two_phase"
            with torch.cuda.stream(main):
                trainer.sync.sync(trainer.fp.grad)
                t0e.record(main)
                for _ in range(10):
                    trainer.sync.sync(trainer.fp.grad)
                t1e.record(main)
            torch.cuda.synchronize(dev)
            two_phase["exchange_ms_alone"] = t0e.elapsed_time(t1e) / 10
            with torch.cuda.stream(main):
                for _ in range(3):
                    trainer.step((x, mask, cond))
                t0e.record(main)
                for _ in range(TRAIN_ALONE_REPS):
                    trainer.step((x, mask, cond))
                t1e.record(main)
            torch.cuda.synchronize(dev)
            two_phase["train_ms_alone"] = t0e.elapsed_time(t1e) / TRAIN_ALONE_REPS
            check_replicas(f"after {3 + TRAIN_ALONE_REPS} steps with the two-phase exchange")
        except SystemExit:
            raise
        except Exception as exc:  # noqa: BLE001 -- diagnostic leg only
            two_phase["error"] = f"{type(exc).__name__}: {exc}"[:300]
        finally:
            trainer.sync.exchange = keep_x
            guard.cancel()
        if rank == 0:
            res["grad_exchange"]["two_phase_direct"] = two_phase
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        if two_phase["error"]:
            os._exit(0)  # ranks may be out of step behind a failed collective: do not wait for each other in the teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
