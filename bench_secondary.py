#!/usr/bin/env python3
"""Secondary benchmark lines: the BASELINE configurations bench.py does not run (it measures cfg 3, the one the
metric is quoted on).  One JSON line per workload, same schema as bench.py, single GPU:

    python bench_secondary.py [--workload jetnet30|lhco_transformer|lhco_crossattention|jetclass|all] [--steps K] [--warmup W] [--precision fp32|f16x3]

step = 1 train step (loss forward + backward through the HIP kernels, clip 0.5 + AdamW + EMA) + 1 midpoint sample
(ode_steps = 100, 198 NFE) on the configuration's batch; inputs synthetic and resident in HBM.  `roofline.achieved` is
the FLOP rate the matrix cores executed in the sampling launches over the timed wall time (padded particles skipped), against
the fp32 MFMA peak; `dense_equiv_over_peak` keeps SURVEY.md section 8's dense count for comparison.
`cpu_baseline` = the oracle on the host cores on a bounded sample (a few jets, 3-step sample scaled to 100 steps: stated).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench import FP32_MFMA_PEAK, usable_cores  # noqa: E402

EPIC = dict(model="epic", frequencies=16, activation="leaky_relu", wrapper_func="weight_norm", t_local_cat=True,
            t_global_cat=True, add_time_to_input=False, t_emb="cosine", loss_type="FM-OT", sigma=1e-4, dropout=0.0, sum_scale=1e-2)
TF_NET = dict(
    node_embd_config=dict(act_h="lrlu", nrm="layer"), ctxt_embd_config=dict(outp_dim=64, act_h="lrlu", nrm="layer"),
    te_config=dict(model_dim=256, num_layers=3, mha_config=dict(num_heads=16, init_zeros=True, do_layer_norm=True),
                   dense_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True)),
    outp_embd_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True))
CA_NET = dict(
    node_embd_config=dict(act_h="lrlu", nrm="layer"), ctxt_embd_config=dict(outp_dim=64, act_h="lrlu", nrm="layer"),
    cae_config=dict(model_dim=128, num_layers=8, mha_config=dict(num_heads=16, init_zeros=True, do_layer_norm=True),
                    dense_config=dict(hddn_dim=256, act_h="lrlu", nrm="layer", output_init_zeros=True)),
    outp_embd_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True))
WORKLOADS = {
    # name: (hparams, batch, n_min, Cg, algorithmic fwd FLOP / jet / NFE, description)
    "jetnet30": (dict(EPIC, features=3, hidden_dim=128, num_particles=30, layers=6, latent=10, global_cond_dim=0, local_cond_dim=0),
                 1024, 10, 0, 17.29e6, "cfg 2: EPiC-FM JetNet30 (N=30, F=3, H=128, 6 layers), batch 1024 (fp32 here; bf16 operands: bench.py / tests/diag/bf16_time.py)"),
    "lhco_transformer": (dict(model="droid_fulltransformer", features=3, num_particles=279, frequencies=16, global_cond_dim=5,
                              add_time_to_input=True, t_emb="cosine", loss_type="FM-OT", sigma=1e-4, net_config=TF_NET),
                         128, 20, 5, 1365e6, "cfg 4: Full-Transformer LHCO (N=279, D=256, 3 layers, 16 heads, 2088515 params), batch 128"),
    "lhco_crossattention": (dict(model="droid_fullcrossattention", features=3, num_particles=279, frequencies=16, global_cond_dim=5,
                                 add_time_to_input=True, t_emb="cosine", loss_type="FM-OT", sigma=1e-4, net_config=CA_NET),
                            128, 20, 5, 644.6e6, "SURVEY 8(f)-4: cross-attention encoder, configs/model/fm_droid_crossattention.yaml on LHCO "
                            "(N=279, D=128, 8 layer pairs, 16 heads, 4 global tokens, 2535107 params), batch 128"),
    "jetclass": (dict(EPIC, features=13, hidden_dim=300, num_particles=128, layers=20, latent=16, global_cond_dim=12, local_cond_dim=0),
                 256, 20, 12, 1083.1e6, "cfg 5: EPiC-FM JetClass (N=128, F=13, H=300, L=16, 20 layers, Cg=12, 8504698 params), batch 256"),
}


def make_batch(B, N, F, C, n_min, seed):
    gen = torch.Generator().manual_seed(seed)
    n = torch.randint(n_min, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, F, generator=gen) * mask
    cond = torch.randn(B, C, generator=gen) if C else torch.zeros(B)
    return x, mask, cond


def cpu_baseline(name, hp, state, jets, C, n_min):
    """Oracle (eager PyTorch restatement of the reference graph) on the host cores: 1 train step + a 3-step midpoint
    sample (4 NFE) scaled to the 198 NFE of ode_steps = 100."""
    from oracle.fm_ref import EpicVectorField, fm_ot_loss, sample_midpoint
    from oracle.ca_ref import CrossAttentionVectorField
    from oracle.tf_ref import TransformerVectorField
    cores = usable_cores()
    torch.set_num_threads(cores)
    N, F = hp["num_particles"], hp["features"]
    x, mask, cond = make_batch(jets, N, F, C, n_min, 4242)
    cond = cond if C else None
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in state.items()}
    params = [v for v in st.values() if v.requires_grad]
    if hp["model"] == "epic":
        vf = EpicVectorField(st, "flows.0.net", dict(hp, sum_scale=1e-2))
    elif hp["model"] == "droid_fullcrossattention":
        vf = CrossAttentionVectorField(st, "flows.0.", hp)
    else:
        vf = TransformerVectorField(st, "flows.0.", hp)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-5)
    import statistics

    def train_step():
        opt.zero_grad()
        loss, *_ = fm_ot_loss(vf, x, mask, cond, torch.rand(jets), torch.randn_like(x), sigma=1e-4)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()

    def sample_part():
        with torch.no_grad():
            sample_midpoint(vf, z0, cond, mask, ode_steps=3)

    def timed(fn, warmup=1, reps=3):
        for _ in range(warmup):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    z0 = torch.randn(jets, N, F)
    t_train = timed(train_step)
    t_sample = timed(sample_part) * 198 / 4
    return {"value": jets / (t_train + t_sample), "unit": "jets/s", "cores": cores, "kind": "port",
            "sample": f"{jets} jets, 1 warm-up + 3 timed iterations per leg, median: 1 train step; a 3-step midpoint sample (4 NFE) "
                      f"SCALED x 198/4 to ode_steps=100; eager-PyTorch oracle, fp32, torch threads = {cores}",
            "train_jets_per_s": jets / t_train, "sample_jets_per_s": jets / t_sample}


def run(name, args):
    from particle_fm_amd.engine import FusedFMTrainer
    from particle_fm_amd.utils.streams import concurrent_streams
    from particle_fm_amd.models import SetFlowMatchingLitModule
    hp, B, n_min, C, flop, what = WORKLOADS[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(12345)
    model = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(hp))
    if hp["model"] != "epic":  # init_zeros / output_init_zeros make an untrained transformer's field identically 0
        with torch.no_grad():
            for p in model.parameters():
                if float(p.abs().sum()) == 0.0 and p.dim() == 2:
                    p.uniform_(-1.0, 1.0).div_(p.shape[1] ** 0.5)
    model = model.to(dev)
    state_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith("flows.")}
    net = model.flows[0].net
    if args.precision != "fp32":
        net.set_precision(args.precision)
    # cfg 2: 1024 jets of <= 30 particles -- two jets per workgroup on an 80-row LDS tile (hip_ops.packed_tile_rows; same numbers bit for
    # bit): a jet this short is all fixed cost, which a pair shares
    packed = name == "jetnet30" and not args.no_pack
    if packed:
        net.set_jet_packing(True)
    valid_rows = hasattr(net, "set_valid_rows_only") and not args.dense_rows
    if valid_rows:  # what generate_data does with variable_set_sizes: the sampler skips padded particles (EPiC always does)
        net.set_valid_rows_only(True)
    # sampler calls in flight: 2 everywhere, 3 for the cross-attention path (its ~200 short launches per step leave most CUs idle,
    # a third call still finds room: 443 jets/s one call at a time, 621 with two, 708 with three in flight).  From two calls on the
    # host's launch rate, not the GPU, would set the pace (485 with two): the sampler then replays its captured step body
    # (cfg 5 likewise: 105 launches per evaluation, a dependent launch costs ~5 us whatever it does: 533 / 645 / 670 jets/s with 1 / 2 / 3;
    # cfg 4: 728 / 746 with 2 / 3).  cfg 2's jet-resident sampler is one launch per call: two in flight, like bench.py
    overlap = args.overlap if args.overlap is not None else (2 if hp["model"] == "epic" and name != "jetclass" else 3)
    graph = overlap > 1 and getattr(net, "_GRAPH_FLAG", 0) != 0 and not args.no_graph
    if graph:
        net.set_graph_replay(True)
    if overlap > 1 and getattr(net, "_ONE_STREAM_FLAG", 0):
        net.set_stream_split(False)  # whole calls overlap here; a call that splits itself over two more streams only adds launches
    trainer = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    N, F = hp["num_particles"], hp["features"]
    x, mask, cond = (a.to(dev) for a in make_batch(B, N, F, C, n_min, 12345))
    z = (torch.randn(B, N, F, generator=torch.Generator().manual_seed(9999)) * mask.cpu()).to(dev)
    cnd = cond if C else None

    # step i = train step i (default stream) + sample i with the weights of step i, on stream i % D; D = 2: the sample runs
    # while step i+1 trains and the next sample is queued (bench.py does the same; all launches are inside the timed region)
    D = max(1, overlap)
    main = torch.cuda.current_stream(dev)
    # the sampling streams must sit on hardware queues of their own (checked by measurement, utils/streams.py)
    streams = concurrent_streams(D, dev, also=[main]) if D > 1 else [main]
    done = [torch.cuda.Event() for _ in range(D)]
    outs = [None] * D

    def step(i, ev=None):
        s = i % D
        main.wait_event(done[s])
        if ev:
            ev[0].record(main)
        if not args.sample_only:
            trainer.step((x, mask, cond))
        with torch.no_grad():
            blob = net.packed_weights(N)  # this step's weights, packed on the training stream
        if ev:
            ev[1].record(main)
        streams[s].wait_stream(main)
        with torch.cuda.stream(streams[s]), torch.no_grad():
            blob.record_stream(streams[s])
            if ev:
                ev[2].record(streams[s])
            outs[s] = model(z, cond=cnd, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=args.ode_steps, weights=blob)
            if ev:
                ev[3].record(streams[s])
            done[s].record(streams[s])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i, e in enumerate(evs):
        step(i, e)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    out = outs[(args.steps - 1) % D]
    assert torch.isfinite(out[mask.squeeze(-1) > 0]).all()
    train_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps
    sample_ms = sum(e[2].elapsed_time(e[3]) for e in evs) / args.steps  # per sample, on its own stream (samples may overlap)
    n_nfe = 2 * (args.ode_steps - 1)
    dense_launch = B * n_nfe * flop / (sample_ms * 1e-3)          # SURVEY 8d dense count / HIP-event time of ONE sampler call
    dense_aggregate = B * n_nfe * flop * args.steps / elapsed     # the same count over the timed wall time (calls overlap)
    # what the matrix cores really ran: the samplers skip padded particles (row work ~ n, self-attention ~ n^2)
    nv = mask.sum(dim=(1, 2)).double().cpu()
    att_share = {"lhco_transformer": 3 * 4 * 279 ** 2 * 256 / 1365e6}.get(name, 0.0)
    skips = valid_rows or hp["model"] == "epic"
    exec_share = ((1 - att_share) * float(nv.mean()) / N + att_share * float((nv ** 2).mean()) / N ** 2) if skips else 1.0
    executed = dense_aggregate * exec_share  # over the timed wall time: cannot exceed the peak
    # the dense MFMA peak of the operand type (MI355X_MICROARCH.md): bf16 / fp16 ~2.5 PFLOP/s; f16x3 runs 3 fp16 MFMAs per product
    # block, so its useful-FLOP peak is a third of that
    peak = {"fp32": FP32_MFMA_PEAK, "bf16": 2500e12, "f16x3": 2500e12 / 3}[args.precision]
    # the kernel the sampler really launches for this descriptor, and its HBM traffic from the committed rocprofv3 --pmc passes
    kernel = {"lhco_transformer": "sampling launches (tf_linear_panel_kernel<4, 2> dominates)",
              "lhco_crossattention": "sampling launches (tf_mlp_panel_kernel<2, 4, true> dominates)",
              "jetclass": "sampling launches (ew_pair_kernel<5, 1> dominates)" if args.precision == "fp32"
                          else "sampling launches (tf_linear_kernel<0, 2, 0, true> dominates)"}.get(name, "sampling launches")
    if name == "jetnet30":
        from particle_fm_amd import hip_ops
        from particle_fm_amd.layout import PFM_F_QUAD_JETS
        lay30 = net.layout(N)
        big = hip_ops.packed_layout(lay30, N)
        mode = {"fp32": 0, "bf16": 1, "f16x3": 2}[args.precision]
        if big is not None and int(big.desc.flags) & PFM_F_QUAD_JETS:
            kernel = f"epic_sample_midpoint_quad_kernel<{mode}>"
        elif args.precision == "f16x3":
            kernel = "epic_sample_midpoint_kernel<2, true>"
        else:
            kernel = f"epic_sample_midpoint_fast_kernel<{mode}, {'true' if packed else 'false'}, false>"
    traffic, traffic_note = None, "no rocprofv3 --pmc pass on file for this workload / precision"
    pmc_path = os.path.join(ROOT, "profiles", f"round4_pmc_{name}_{args.precision}.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        tot = pmc.get("_totals", {})
        if name != "jetnet30":  # the launch that takes the largest share of an evaluation, by the profiler's own durations
            per = {k: v["duration_ns"] * v["launches_seen"] for k, v in pmc.items()
                   if isinstance(v, dict) and "launches_seen" in v and "duration_ns" in v and "spin_kernel" not in k}
            if per:
                top = max(per, key=per.get)
                kernel = f"sampling launches ({top.split('::')[-1]} dominates: {100.0 * per[top] / sum(per.values()):.0f} % of an evaluation's kernel time)"
        if tot.get("hbm_bytes_per_evaluation"):
            traffic = tot["hbm_bytes_per_evaluation"] * n_nfe
            traffic_note = (f"traffic = HBM-side bytes of ONE sampler call ({n_nfe} evaluations): (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over every "
                            f"launch of a sampling-only run of {tot.get('evaluations')} evaluations under rocprofv3 --pmc (separate passes, gfx950 "
                            f"correction; tests/diag/collect_pmc_rowmatrix.sh), per evaluation x {n_nfe}: profiles/{os.path.basename(pmc_path)}; "
                            f"algorithmic bytes per call (inputs, outputs, masks, ONE read of the weights): {tot.get('algorithmic_note', 'see DESIGN.md section 5')}")
    res = {
        "metric": "jets/sec (train step + 100-step ODE sample)", "value": B * args.steps / elapsed, "unit": "jets/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        # dtype = the arithmetic type of the matrix products: "bf16" = bf16 MFMA operands with fp32 accumulate (the reference's
        # bf16-mixed); "f16x3" = split-fp16 operands (22-bit products, NOT fp32: never reported as f32)
        "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "f16x3": "f16x3"}[args.precision],
        "data": "synthetic",
        "config": {"workload": what, "jets_per_gpu": B, "parallelism": "dp1", "ode_steps": args.ode_steps, "overlap": D,
                   "multiplicity": f"U{{{n_min}..{N}}} per jet",
                   "sampler_rows": "valid particles only" if (valid_rows or hp["model"] == "epic") else "all N rows (padded included)",
                   "sampler_launches": "step body captured once per call, replayed as a hipGraph" if graph else "every launch enqueued by the host",
                   "jets_per_workgroup": 4 if "quad" in kernel else (2 if packed else 1)},
        "train_ms": train_ms, "sample_ms": sample_ms, "train_jets_per_s": B / (train_ms * 1e-3),
        "sample_jets_per_s": B / (sample_ms * 1e-3),
        "roofline": {"bound": "mfma", "kernel": kernel,
                     "achieved": executed / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": executed / peak,
                     "traffic": traffic, "traffic_note": traffic_note, "concurrent_launches": D, "valid_row_fraction": float(nv.mean()) / N,
                     "executed_share_of_dense": exec_share,
                     "dense_equiv_over_peak": dense_launch / peak,
                     "dense_equiv_over_peak_aggregate": dense_aggregate / peak,
                     "note": "frac = achieved / peak, achieved = estimate of the FLOP the matrix cores executed in the sampling launches over "
                             f"the timed wall time: SURVEY 8d's dense count ({flop/1e6:.1f} MFLOP/jet/NFE over the padded N x {n_nfe} NFE x {B} "
                             "jets) scaled by executed_share_of_dense (padded particles are skipped: row work ~ mean(n)/N, self-attention ~ "
                             "mean(n^2)/N^2); dense_equiv_over_peak = the dense count / HIP-event time of ONE sampler call (calls overlap: "
                             "a throughput-equivalent, not MFMA utilisation)"},
    }
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(name, hp, state_cpu, 8 if name != "jetnet30" else 64, C, n_min)
    print(json.dumps(res), flush=True)
    del trainer, model
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=list(WORKLOADS) + ["all"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ode-steps", type=int, default=100)
    ap.add_argument("--no-pack", action="store_true", help="jetnet30: one jet per workgroup (default: two, PFM_F_PACK_JETS)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3", "bf16"],
                    help="matrix operands: f16x3 = split fp16, fp32-grade accuracy (jet-resident EPiC: the sampler; row-matrix paths: "
                         "every Linear, training included); bf16 = bf16 operands in every Linear of the model, training included (BASELINE cfg 2 is quoted in bf16; "
                         "training stays fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=None, help="sampler calls in flight (1 = strictly sequential; default 2, "
                    "cross-attention 3)")
    ap.add_argument("--no-graph", action="store_true", help="cross-attention path: enqueue every launch of the sampler from the host "
                    "instead of replaying the captured step body (hipGraph)")
    ap.add_argument("--dense-rows", action="store_true", help="transformer paths: sample all N rows like the reference (padded included)")
    ap.add_argument("--sample-only", action="store_true", help="diagnostics (tests/diag/collect_pmc_rowmatrix.sh): no train step, so that every "
                    "launch of the run belongs to the sampler; the line it prints is NOT a benchmark line")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_secondary.py needs an MI355X (the HIP path has no CPU fallback)")
    for name in (WORKLOADS if args.workload == "all" else [args.workload]):
        run(name, args)


if __name__ == "__main__":
    main()
