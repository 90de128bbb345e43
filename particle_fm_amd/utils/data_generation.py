"""Drop-in for particle_fm/utils/data_generation.py::generate_data (utils/data_generation.py:17-176).

Same signature, same batching (full batches, then the remainder with the LAST rows of cond / mask), same timing
convention (the clock starts with the second batch and stops before the remainder batch), same return value
(numpy array, seconds).  Differences, all on purpose: the batches stay in HBM -- inverse normalisation, log_pt and the
mask multiply run as one HIP launch per batch (pfm_sample_epilogue) instead of after a D2H copy -- and there is a
single D2H copy of the concatenated result at the end; the clock is bracketed by a device synchronisation so that it
measures the generation, not the enqueue.
"""
from __future__ import annotations

import contextlib
import ctypes
import time
from typing import Optional

import numpy as np
import torch

from .. import _lib
from ..hip_ops import _ptr, _stream_ptr
from .streams import concurrent_streams


def sample_epilogue_(x: torch.Tensor, mask: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
                     shift: Optional[torch.Tensor] = None, log_pt_col: int = -1) -> torch.Tensor:
    """In place on a device tensor x (B,N,F): x*scale+shift, 1-exp on one column, *mask."""
    if not x.is_cuda:
        raise RuntimeError("sample_epilogue_ needs a ROCm device tensor (no CPU fallback)")
    lib = _lib.load()
    assert x.is_contiguous() and x.dtype == torch.float32
    B, N, F = x.shape
    dev = x.device
    m = None if mask is None else mask.reshape(B, N).to(device=dev, dtype=torch.float32).contiguous()
    sc = None if scale is None else scale.to(device=dev, dtype=torch.float32).contiguous()
    sh = None if shift is None else shift.to(device=dev, dtype=torch.float32).contiguous()
    rc = lib.pfm_sample_epilogue(_ptr(x), _ptr(m), _ptr(sc), _ptr(sh), int(log_pt_col), ctypes.c_int64(B * N), F, _stream_ptr(dev))
    _lib.check(rc, "pfm_sample_epilogue")
    return x


def _affine(means, stds, F, normalize_sigma, pt_standardization):
    """(scale, shift) of inverse_normalize_tensor as fp32 tensors: tensor * (std / sigma) + mean, the quotient taken in
    double like the reference's python arithmetic (data/components/utils.py:197-198)."""
    scale, shift = np.ones(F, dtype=np.float64), np.zeros(F, dtype=np.float64)
    if pt_standardization:  # data_generation.py:98-105: sigma 10 for (eta, phi), 5 for pt
        for i in range(2):
            scale[i], shift[i] = float(stds[i]) / 10, float(means[i])
        scale[2], shift[2] = float(stds[2]) / 5, float(means[2])
    else:
        for i in range(len(means)):
            scale[i], shift[i] = float(stds[i]) / normalize_sigma, float(means[i])
    return torch.from_numpy(scale.astype(np.float32)), torch.from_numpy(shift.astype(np.float32))


def generate_data(model, num_jet_samples: int, batch_size: int = 256, cond: torch.Tensor = None, device: str = "cuda",
                  variable_set_sizes: bool = False, mask: torch.Tensor = None, normalized_data: bool = False,
                  normalize_sigma: int = 5, means=None, stds=None, log_pt: bool = False, pt_standardization: bool = False,
                  shuffle_mask: bool = False, verbose: bool = True, ode_solver: str = "midpoint", ode_steps: int = 100,
                  valid_rows_only: bool = True, pipeline: bool = True, _shard=(0, 1)):
    if variable_set_sizes and mask is None:
        raise ValueError("Please use mask when using variable_set_sizes=True")  # data_generation.py:62-63
    if mask is not None and len(mask) != num_jet_samples:
        raise ValueError(f"Mask should have the same length as num_jet_samples ({len(mask)} != {num_jet_samples})")
    dev = torch.device(device)
    if verbose:
        print(f"Generating data ({num_jet_samples} samples). Device: {dev}")
    model = model.to(dev)
    # With variable_set_sizes the result is multiplied by the mask below, so the transformer paths may skip the padded
    # particles (extension `valid_rows_only`; EPiC does so anyway): same output, about half the work at LHCO multiplicities.
    switched = []
    if variable_set_sizes and valid_rows_only:
        for f in getattr(model, "flows", []):
            net = getattr(f, "net", None)
            if hasattr(net, "set_valid_rows_only") and not net.valid_rows_only:
                net.set_valid_rows_only(True)
                switched.append(net)
    # Two batches in flight from this one thread: the cross-attention sampler (~200 launches per step) is then bound by the host's
    # launch rate unless it replays its captured step body (PFM_CA_F_GRAPH_STEPS); same kernels, same results.
    replay, unsplit = [], []
    if pipeline and dev.type == "cuda" and _pipelined(model) and ode_solver == "midpoint":
        net = model.flows[0].net
        if getattr(net, "_GRAPH_FLAG", 0) and hasattr(net, "set_graph_replay") and not net.graph_replay:
            net.set_graph_replay(True)
            replay.append(net)
        if getattr(net, "_ONE_STREAM_FLAG", 0) and getattr(net, "stream_split", False):
            net.set_stream_split(False)  # the pipeline overlaps whole calls; a call splitting itself as well only adds launches
            unsplit.append(net)
    try:
        return _generate(model, num_jet_samples, cond, batch_size, dev, variable_set_sizes, mask, normalized_data, normalize_sigma,
                         means, stds, log_pt, pt_standardization, shuffle_mask, ode_solver, ode_steps, pipeline, _shard)
    finally:
        for net in switched:
            net.set_valid_rows_only(False)
        for net in replay:
            net.set_graph_replay(False)
        for net in unsplit:
            net.set_stream_split(True)


_PIPE_STREAMS = {}


def _pipeline_streams(dev, n=2):
    """the side streams of the batch pipeline, created once per device and checked to run side by side (the stream ->
    hardware-queue mapping is fixed at creation; streams that land on one queue lose the overlap: utils/streams.py)"""
    key = (str(dev), n)
    if key not in _PIPE_STREAMS:
        _PIPE_STREAMS[key] = concurrent_streams(n, dev)
    return _PIPE_STREAMS[key]


def _pipelined(model) -> bool:
    """a single flow on a HIP network: the weights can be packed once for all batches, and launches on different streams share no
    activation workspace (the jet-resident EPiC kernel has none, the row-matrix paths keep one per stream)"""
    flows = list(getattr(model, "flows", []))
    return len(flows) == 1 and hasattr(flows[0].net, "packed_weights")


def _generate(model, num_jet_samples, cond, batch_size, dev, variable_set_sizes, mask, normalized_data, normalize_sigma, means, stds,
              log_pt, pt_standardization, shuffle_mask, ode_solver, ode_steps, pipeline, shard=(0, 1)):
    rank, world = shard
    dev_sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)  # (the sampler itself refuses CPU tensors)
    n_full = num_jet_samples // batch_size
    rem = num_jet_samples - n_full * batch_size
    scale = shift = None
    outs = []
    start_time = 0.0
    # Extension `pipeline`: the parameters are packed once for all batches, and consecutive batches alternate between two streams.
    # Jet-resident EPiC: a sampler launch lasts as long as its largest jet, and with the jets taken longest-first the next batch
    # starts in the gaps of the current one; row-matrix models (transformer, cross-attention, wide EPiC): the short dependent
    # launches of one batch run in the gaps of the other's.  Same draws in the same order (CPU generator), same results.
    blob, streams = None, None
    if pipeline and dev.type == "cuda" and _pipelined(model) and ode_solver == "midpoint":
        with torch.no_grad():
            blob = model.flows[0].net.packed_weights(getattr(model.hparams, "num_particles", None))
        # batches in flight: 2; 3 for the cross-attention model, whose short launches leave room for a third (bench_secondary.py)
        streams = _pipeline_streams(dev, 3 if getattr(model.flows[0].net, "_GRAPH_FLAG", 0) else 2)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))

    P = len(streams) if streams else 1
    pinned, staged = None, [None] * P
    if streams and not getattr(model.hparams, "use_normaliser", False):
        # sample() inlined (flow_matching_module.py:656-674) so that nothing in the loop blocks the host: z is drawn by the same
        # torch.randn call into a pinned buffer (one per stream, reused once its copy has been consumed) and copied asynchronously
        N_, F_ = model.hparams.num_particles, model.hparams.features
        pinned = [torch.empty(batch_size, N_, F_, pin_memory=True) for _ in range(P)]

    n_seen = 0

    def one_batch(n, cond_b, mask_b):
        nonlocal scale, shift, n_seen
        n_seen += 1
        if (n_seen - 1) % world != rank:
            # another rank's batch (generate_data_sharded): consume the same numbers of the CPU generator as sample() would
            # (flow_matching_module.py:659-663), so that every rank's z is the slice the single-process run would have drawn
            torch.randn(n, getattr(model.hparams, "num_particles"), getattr(model.hparams, "features"))
            return
        k = len(outs) % P
        ctx = torch.cuda.stream(streams[k]) if streams else contextlib.nullcontext()
        with ctx, torch.no_grad():
            if pinned is not None:
                if staged[k] is not None:
                    staged[k].synchronize()  # the H2D copy that last read this pinned buffer
                zh = pinned[k][:n]
                torch.randn(n, zh.shape[1], zh.shape[2], out=zh)
                z = zh.to(dev, non_blocking=True)
                staged[k] = torch.cuda.Event()
                staged[k].record()
                cd = None if cond_b is None else cond_b.to(dev, non_blocking=True)
                md = None if mask_b is None else mask_b[:n].to(dev, non_blocking=True)
                if md is not None:
                    z = z * md
                x = model.forward(z, cond=cd, mask=md, reverse=True, ode_solver=ode_solver, ode_steps=ode_steps, weights=blob)
                mask_b = md  # the epilogue below must not copy the host mask again: a pageable copy behind the sampler would
                             # park the host until the sampler is done
            else:
                kw = {"weights": blob} if blob is not None else {}
                x = model.sample(n_samples=n, cond=cond_b, mask=mask_b, ode_solver=ode_solver, ode_steps=ode_steps, **kw)
            x = x.contiguous()
            if normalized_data and scale is None:
                scale, shift = (t.to(dev) for t in _affine(means, stds, x.shape[-1], normalize_sigma, pt_standardization))
                if streams:  # made on this stream, read by the other one from the next batch on
                    torch.cuda.current_stream(dev).synchronize()
            sample_epilogue_(x, mask_b if variable_set_sizes else None, scale if normalized_data else None,
                             shift if normalized_data else None, 2 if (normalized_data and log_pt) else -1)
        outs.append(x)

    for i in range(n_full):
        cond_b = None if cond is None else cond[i * batch_size:(i + 1) * batch_size]
        if i == 1:  # the reference's convention: the first (warm-up) batch is not timed (data_generation.py:82-83)
            dev_sync()
            start_time = time.time()
        if variable_set_sizes:
            if shuffle_mask:
                mask = mask[np.random.permutation(len(mask))]
                mask_b = mask[:batch_size]
            else:
                mask_b = mask[i * batch_size:(i + 1) * batch_size]
        else:
            mask_b = None
        one_batch(batch_size, cond_b, mask_b)
    dev_sync()
    end_time = time.time()
    if rem:
        cond_b = None if cond is None else cond[-rem:]
        if variable_set_sizes:
            if shuffle_mask:
                mask = mask[np.random.permutation(len(mask))]
            mask_b = mask[-rem:]
        else:
            mask_b = None
        one_batch(rem, cond_b, mask_b)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)  # the remainder batch may have run on a side stream
    if world > 1:
        return outs, end_time - start_time  # this rank's batches (round-robin), still on the device
    data = torch.cat(outs).cpu().numpy() if outs else np.zeros((0,), dtype=np.float32)
    return data, end_time - start_time


def shard_plan(num_jet_samples: int, batch_size: int, world: int):
    """Row ranges (start, stop) of the batches generate_data forms -- full batches in order, then the remainder -- and the rank
    that takes each (round-robin over batches: consecutive batches go to different GPUs, every rank sees the same mix of the
    caller's jet order)."""
    n_full = num_jet_samples // batch_size
    spans = [(i * batch_size, (i + 1) * batch_size) for i in range(n_full)]
    if num_jet_samples - n_full * batch_size:
        spans.append((n_full * batch_size, num_jet_samples))
    return [(a, b, i % world) for i, (a, b) in enumerate(spans)]


def generate_data_sharded(model, num_jet_samples: int, batch_size: int = 256, process_group=None, gather: bool = True,
                          rank: Optional[int] = None, world: Optional[int] = None, **kw):
    """``generate_data`` with the jet list split across the ranks of ``process_group`` (sampling is embarrassingly
    parallel; one process per GPU, no data-path collective).  The reference's helper is single-process
    (utils/data_generation.py:17-176); this one returns the SAME array bit for bit:

    * every rank walks the same batch list (:func:`shard_plan`) and runs batch i iff ``i % world == rank``;
    * z is drawn from the CPU generator exactly as the single process draws it -- every rank draws every batch's z (cheap next to
      198 network evaluations) and drops the ones it does not run -- so all ranks must enter with the same CPU RNG state
      (``seed_everything`` / ``torch.manual_seed``, as the reference's evaluation callbacks do) and leave with the state the
      single-process run leaves; ``shuffle_mask`` likewise relies on identical numpy RNG state;
    * ``cond`` / ``mask`` are the FULL arrays on every rank (a rank slices its rows itself).

    ``gather=True``: the shards are exchanged with one all_gather of equal-sized (padded) device tensors (RCCL for backend
    "nccl", gloo in the CPU tests) and every rank returns the full (num_jet_samples, N, F) array; ``gather=False``: returns this
    rank's rows and their indices ``(rows, index)``.  The time returned is this rank's (same convention as generate_data:
    from its second batch to the end of its full batches)."""
    import torch.distributed as dist

    if world is None:
        world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
    if world == 1:
        return generate_data(model, num_jet_samples, batch_size=batch_size, **kw)
    outs, seconds = generate_data(model, num_jet_samples, batch_size=batch_size, _shard=(rank, world), **kw)
    plan = shard_plan(num_jet_samples, batch_size, world)
    mine = [(a, b) for a, b, r in plan if r == rank]
    assert len(mine) == len(outs) and all(o.shape[0] == b - a for o, (a, b) in zip(outs, mine))
    index = np.concatenate([np.arange(a, b) for a, b in mine]) if mine else np.zeros((0,), dtype=np.int64)
    if not gather:
        rows = torch.cat(outs).cpu().numpy() if outs else np.zeros((0,), dtype=np.float32)
        return (rows, index), seconds
    N, F = model.hparams.num_particles, model.hparams.features
    dev = outs[0].device if outs else torch.device(kw.get("device", "cuda"))
    counts = [sum(b - a for a, b, r in plan if r == q) for q in range(world)]
    pad = max(counts)
    buf = torch.zeros(pad, N, F, device=dev, dtype=torch.float32)
    if outs:
        buf[:counts[rank]] = torch.cat(outs)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=process_group)
    full = torch.empty(num_jet_samples, N, F, dtype=torch.float32)
    for q in range(world):
        rows_q = np.concatenate([np.arange(a, b) for a, b, r in plan if r == q]) if counts[q] else np.zeros((0,), dtype=np.int64)
        full[torch.from_numpy(rows_q)] = parts[q][:counts[q]].cpu()
    return full.numpy(), seconds
