"""Data-parallel training step around the HIP loss: flat parameter/gradient buffers, one RCCL
all-reduce, fused clip + AdamW + EMA.

What it replaces in the reference's stack (all host orchestration, none of it in particle_fm itself):
Lightning's automatic optimisation for one step = ``loss.backward()``, torch DDP's bucketed gradient
all-reduce (configs/trainer/ddp.yaml:4-9), ``clip_grad_norm_(0.5)`` (experiment/jetnet/fm_tops150.yaml:24),
``AdamW(lr=1e-3, weight_decay=5e-5)`` (configs/model/flow_matching.yaml:3-7) and the EMA callback's per-step
update (callbacks/ema.py:73-81).

Multi-GPU mapping: one process per GPU, full weight replica, each rank takes its own jets;
the 561 330-element fp32 gradient (2.2 MB) lives in ONE flat buffer, so the exchange is a single
``all_reduce(SUM)`` over RCCL/xGMI followed by a multiply with 1/world (DDP's mean-of-per-rank-gradients
semantics: each rank normalises its loss by its own mask count, losses.py:75-76).  At 2.2 MB the collective
is latency-bound on the 7 xGMI links; one flat call lets RCCL pick its low-latency protocol instead of
~90 tiny per-tensor reductions.  Sampling needs no collective.
"""
from __future__ import annotations

import ctypes
import contextlib
import math
from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, hip_ops


class FlatParams:
    """Re-homes every parameter of ``module`` as a view into one flat fp32 buffer (and its .grad into a second
    one).  ``state_dict`` / ``load_state_dict`` keep working (they copy in place); after ``module.to(device)``
    call :meth:`rebuild`."""

    def __init__(self, params: Iterable[nn.Parameter], first: Optional[set] = None):
        """``first``: ids of the parameters whose gradients are final EARLY in the backward; they are laid out in front of the others
        (``n_first`` floats) so that their share of the gradient is one contiguous range that can be reduced while the rest is still
        being computed.  The order inside each group stays the registration order."""
        ps = [p for p in params if p.requires_grad]
        if first:
            ps = [p for p in ps if id(p) in first] + [p for p in ps if id(p) not in first]
        self.params: List[nn.Parameter] = ps
        self._first = set(first or ())
        self.rebuild()

    def rebuild(self):
        ps = self.params
        dev = ps[0].device
        sizes = [p.numel() for p in ps]
        # 16-byte alignment of every view keeps float4 access legal in the optimiser kernel's tails
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += (n + 3) & ~3
        self.numel = o
        self.offsets = offs
        nf = sum(1 for p in ps if id(p) in self._first)
        self.n_first = (offs[nf] if nf < len(ps) else o) if nf else 0  # floats of the early group (a multiple of 4)
        flat = torch.zeros(o, device=dev, dtype=torch.float32)
        grad = torch.zeros(o, device=dev, dtype=torch.float32)
        for p, off in zip(ps, offs):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p.data)
            p.grad = grad[off:off + n].view_as(p.data)
        self.flat, self.grad = flat, grad

    def is_intact(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off in zip(self.params, self.offsets))

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # a previous backward may have replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view_as(p.data)


class GradSync:
    """Gradient exchange of the data-parallel step: one all-reduce of the flat buffer.  Backend "nccl" is RCCL
    on ROCm; "gloo" serves the CPU tests."""

    def __init__(self, process_group=None, exchange: Optional[str] = None):
        import os
        self.group = process_group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        # "allreduce" (default): one RCCL all-reduce (a ring: 2 (N - 1) steps of 1/N of the buffer each).
        # "two_phase" (opt-in, PFM_DP_EXCHANGE=two_phase): the direct exchange SURVEY 5.8 / 8e names for the 2.2 MB gradient on
        # point-to-point xGMI -- (1) every rank sends slice r of its buffer straight to rank r (one all-to-all: all 7 links of a GPU at
        # once, 1/N of the buffer per link), (2) rank r adds the N copies of its slice (one reduction kernel), (3) one all-gather hands the summed
        # slices round: two communication steps instead of 2 (N - 1).  Each slice is summed by exactly one rank, so every replica receives
        # the same bits.  Never run on more than one RCCL rank (no multi-GPU node in this pool): covered by the gloo world-2 test and
        # timed next to the all-reduce by bench.py --gpus N, not used in its timed region.
        self.exchange = exchange or os.environ.get("PFM_DP_EXCHANGE", "allreduce")
        if self.exchange not in ("allreduce", "two_phase"):
            raise ValueError(f"PFM_DP_EXCHANGE / exchange: 'allreduce' or 'two_phase', not {self.exchange!r}")
        self._tp = None

    def sync(self, flat_grad: torch.Tensor) -> float:
        """Sums the buffer over ranks in place; returns the factor (1/world) that turns it into the mean."""
        if self.enabled:
            if self.exchange == "two_phase":
                self.two_phase_sum(flat_grad)
            else:
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def two_phase_sum(self, flat: torch.Tensor) -> None:
        """flat <- sum over ranks, by all-to-all of slices + local sum + all-gather (see __init__)."""
        W, n = self.world, flat.numel()
        per = ((n + W - 1) // W + 3) // 4 * 4  # floats per slice (16-byte multiples; the padding behind n stays zero)
        key = (n, flat.device, flat.dtype)
        if self._tp is None or self._tp[0] != key:
            self._tp = (key, torch.zeros(W * per, device=flat.device, dtype=flat.dtype), torch.empty(W * per, device=flat.device, dtype=flat.dtype),
                        torch.empty(per, device=flat.device, dtype=flat.dtype))
        _, send, recv, shard = self._tp
        send[:n].copy_(flat)
        if dist.get_backend(self.group) == "gloo":  # (CPU tests; gloo has no all_to_all_single: the same slices by way of an all_gather)
            every = [torch.empty_like(send) for _ in range(W)]
            dist.all_gather(every, send, group=self.group)
            for s_ in range(W):
                recv.view(W, per)[s_].copy_(every[s_].view(W, per)[self.rank])
        else:
            dist.all_to_all_single(recv, send, group=self.group)  # recv[s] = slice `rank` of rank s's buffer
        torch.sum(recv.view(W, per), dim=0, out=shard)              # one kernel; only this rank sums this slice
        if dist.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(shard) for _ in range(W)]
            dist.all_gather(parts, shard, group=self.group)
            torch.cat(parts, out=send)
        else:
            dist.all_gather_into_tensor(send, shard, group=self.group)
        flat.copy_(send[:n])

    def start(self, part: torch.Tensor):
        """Begins the sum of one contiguous part of the flat gradient (ordered behind what the current stream has queued so far) and
        returns at once: kernels queued afterwards run next to the collective (RCCL: on the process group's own stream).  Pair with
        :meth:`finish` before anything reads ``part``."""
        return dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.enabled else None

    @staticmethod
    def finish(work) -> None:
        if work is not None:
            work.wait()  # RCCL: the current stream waits for the collective; the host does not block

    def mean_scalar(self, value: torch.Tensor) -> torch.Tensor:
        """``self.log(..., sync_dist=True)`` of the reference (flow_matching_module.py:524)."""
        if self.enabled:
            value = value.detach().clone()
            dist.all_reduce(value, op=dist.ReduceOp.SUM, group=self.group)
            value /= self.world
        return value


def early_linear(name: str) -> bool:
    """Linears of the EPiC network without a 128x128 particle block: their weight gradient needs the backward's chain phase only."""
    leaf = name.rsplit(".", 1)[-1]
    return leaf in ("fc_l1", "fc_l3", "fc_g1", "fc_g2", "fc_global1", "fc_global2")


class FusedEpicTables:
    """Device-side index tables that let the weight-norm kernels read the FLAT parameter buffer and write the
    kernel blob (and back for gradients).  Built once per (module, FlatParams)."""

    def __init__(self, net, layout, fp: "FlatParams"):
        import numpy as np

        dev = fp.flat.device
        off = {id(p): o for p, o in zip(fp.params, fp.offsets)}
        named = dict(net.named_parameters())
        rows, bfrom, bsrc = [], [], []
        # the Linears whose gradient is final after the backward's chain phase (no 128x128 particle block: pfm_hip.h,
        # PFM_BWD_PHASE_CHAIN) first, so that each half of the backward has one contiguous run of table rows
        lins = sorted(layout.linears, key=lambda l: not early_linear(l[0]))
        n_early_rows = n_early_bias = 0
        for name, in_dim, out_dim in lins:
            vo, go, bo = (off[id(named[f"{name}.{k}"])] for k in ("weight_v", "weight_g", "bias"))
            o = np.arange(out_dim)
            rows.append(np.stack([vo + o * in_dim, go + o, np.full(out_dim, in_dim), layout.w_off[name] + o * in_dim], 1))
            bfrom.append(bo + o)
            bsrc.append(layout.b_off[name] + o)
            if early_linear(name):
                n_early_rows += out_dim
                n_early_bias += out_dim
        rows = np.concatenate(rows).astype(np.int32)
        bfrom = np.concatenate(bfrom).astype(np.int32)
        bsrc = np.concatenate(bsrc)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.rows, self.n_rows = t(rows), len(rows)
        self.n_early_rows, self.n_early_bias = n_early_rows, n_early_bias
        self.dst1, self.dst2, self.gsrc = t(layout.src_dst1), t(layout.src_dst2), t(layout.src_gpos)
        self.bias_param = t(bfrom)                                  # flat-parameter offsets of the biases
        self.bias_blob = t(layout.src_dst1[bsrc].astype(np.int32))  # their place in the blob
        self.bias_gblob = t(layout.src_gpos[bsrc].astype(np.int32))  # and in the gradient blob
        self.n_bias = len(bfrom)
        # do the tables name EVERY element of the flat buffer that belongs to a parameter?  Then the unpack kernel may write (=)
        # instead of accumulating and the step needs no zeroing launch (padding between parameters is never read as a gradient:
        # the optimiser multiplies it by nothing that matters -- it stays at its initial 0).
        self.covers_all = (int(sum(i * o for _, i, o in layout.linears)) + int(sum(o for _, _, o in layout.linears)) + self.n_bias
                           == sum(p.numel() for p in fp.params))


class FusedFMTrainer:
    """One training step = HIP loss forward + backward, flat all-reduce, fused clip/AdamW/EMA.

    ``module`` is a ``SetFlowMatchingLitModule`` (or anything with ``.loss(x, mask=, cond=)`` and parameters)
    living on a ROCm device."""

    def __init__(self, module: nn.Module, lr: float = 1e-3, weight_decay: float = 5e-5, betas=(0.9, 0.999),
                 eps: float = 1e-8, max_grad_norm: float = 0.5, ema_decay: Optional[float] = 0.999,
                 process_group=None, lr_schedule: Optional[Callable[[int], float]] = None):
        """``lr_schedule(k)`` -> factor on ``lr`` for the k-th optimiser step (k = 0 for the first one), e.g.
        :func:`cosine_warmup`; None keeps the rate constant."""
        self.module = module
        self.lr_schedule = lr_schedule
        # fully fused path (no autograd): single EPiC flow with an FM-OT / CFM loss.  The transformer's parameters are
        # plain (no weight norm): its autograd node already is two launches + two gathers, so it takes the generic path
        flows = getattr(module, "flows", None)
        fusable = flows is not None and len(flows) == 1 and hasattr(flows[0], "net") and hasattr(flows[0].net, "source_vector") \
            and not getattr(flows[0].net, "wide", False) and getattr(flows[0], "t_emb", None) != "gaussian" \
            and not getattr(flows[0].net, "add_time_to_input", False)  # (its fc_l1 is a folded matrix: autograd path)
        # the parameters whose gradient the backward finishes first lie in front of the flat buffers (FlatParams.n_first): the
        # data-parallel step reduces that range while the dW GEMM of the others runs (fused_loss_and_grad)
        early = {id(p) for n, p in flows[0].net.named_parameters() if early_linear(n.rsplit(".", 1)[0])} if fusable else None
        self.fp = FlatParams(module.parameters(), first=early)
        dev = self.fp.flat.device
        if dev.type != "cuda":
            raise RuntimeError("FusedFMTrainer needs the module on a ROCm device (no CPU fallback)")
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.max_grad_norm = max_grad_norm
        self.ema_decay = ema_decay
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.ema = self.fp.flat.clone() if ema_decay is not None else None
        self.scratch = torch.zeros(1024, device=dev, dtype=torch.float32)
        self.sync = GradSync(process_group)
        self.step_count = 0
        # the reference draws t on the CPU generator (losses.py:46); a pageable host tensor lands on the device through a copy that
        # SYNCHRONISES the stream -- every step would drain the GPU before its first kernel is even queued.  A small ring of pinned
        # staging buffers + asynchronous copies keeps the host a few steps ahead (same numbers: the draw itself is unchanged).
        self._t_ring = []
        self._t_next = 0
        self._fused = None
        # False (default): ONE flat all-reduce behind the whole backward -- the exchange that has run on RCCL.  True (opt-in,
        # PFM_DP_OVERLAP=1): split the backward and overlap the first bucket's all-reduce with the dW GEMM (DDP's bucketed overlap);
        # bit-identical by tests/test_hip_trainer_split.py and the gloo world-2 tests, but no N > 1 RCCL record exists for it yet, so
        # it is not the default (True without a process group: the same two-phase launches, nothing to exchange -- for tests).
        import os
        self.split_backward: bool = os.environ.get("PFM_DP_OVERLAP", "0") == "1"
        self._grad_synced = False
        if fusable:
            self._fused = {}
            flows[0].net._fast_pack = self.packed_blob  # sampling re-packs with one HIP launch instead of ~100 torch ops

    def _land(self, t_cpu: torch.Tensor) -> torch.Tensor:
        """A small host tensor (the per-jet times) -> device, without a stream synchronisation: through one of 8 pinned slots, each
        guarded by an event recorded behind its last copy."""
        dev = self.fp.flat.device
        n = t_cpu.numel()
        if not self._t_ring or self._t_ring[0][0].numel() < n:
            self._t_ring = [(torch.empty(max(n, 1024), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        buf, ev = self._t_ring[self._t_next % 8]
        self._t_next += 1
        ev.synchronize()  # the copy that last read this slot (8 steps ago) has finished
        view = buf[:n].view(t_cpu.shape)
        view.copy_(t_cpu.to(torch.float32))
        out = view.to(dev, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        return out

    def current_lr(self) -> float:
        """learning rate of the NEXT optimiser step"""
        return self.lr * (float(self.lr_schedule(self.step_count)) if self.lr_schedule is not None else 1.0)

    def optimizer_step(self, grad_mul: float = 1.0):
        lr = self.current_lr()
        self.step_count += 1
        fp = self.fp
        P = hip_ops._ptr
        rc = _lib.load().pfm_optim_step(
            P(fp.flat), P(fp.grad), P(self.exp_avg), P(self.exp_avg_sq), P(self.ema), P(self.scratch),
            ctypes.c_int64(fp.numel), grad_mul, self.max_grad_norm if self.max_grad_norm else 0.0, lr,
            self.betas[0], self.betas[1], self.eps, self.weight_decay,
            self.ema_decay if self.ema_decay is not None else 0.0, self.step_count,
            hip_ops._stream_ptr(fp.flat.device))
        _lib.check(rc, "pfm_optim_step")

    def _fused_state(self, n_points: int):
        net = self.module.flows[0].net
        # a new frequency table (set_freq_table) means a new blob; the precision switch (set_precision) another descriptor
        key = (n_points, getattr(net, "_freq_version", 0), getattr(net, "mfma_dtype", "fp32"))
        st = self._fused.get(key)
        if st is None:
            from . import fm_loss
            lay = net.layout(n_points)
            st = {"layout": lay, "tables": FusedEpicTables(net, lay, self.fp),
                  # initial blob the slow (torch) way: fixes freqs + descriptor tail; the pack kernel rewrites the rest
                  "blob": fm_loss.pack_blob_from_source(lay, net.source_vector(lay).detach()).contiguous(),
                  "one": torch.ones(1, device=self.fp.flat.device)}
            # the backward WRITES every gradient slot the unpack kernel reads (no atomics; tests/test_hip_train.py fills it with NaN
            # first and checks): allocated once, never zeroed again
            st["gblob"] = torch.zeros_like(st["blob"])
            self._fused[key] = st
        return st

    def _pack(self, st):
        tb = st["tables"]
        P = hip_ops._ptr
        _lib.check(_lib.load().pfm_wn_pack(P(self.fp.flat), P(tb.rows), tb.n_rows, P(tb.dst1), P(tb.dst2),
                                           P(tb.bias_param), P(tb.bias_blob), tb.n_bias, P(st["blob"]),
                                           hip_ops._stream_ptr(self.fp.flat.device)), "pfm_wn_pack")
        return st["layout"].finish_blob(st["blob"])  # bf16 layouts: + the MFMA_A16 copies (one more launch)

    def packed_blob(self, n_points: int):
        """The kernel blob for the CURRENT flat parameters (weight-norm pack kernel); None if the parameters no
        longer alias the flat buffer (e.g. after module.to()), in which case the caller packs the slow way."""
        if self._fused is None or not self.fp.is_intact():
            return None
        return self._pack(self._fused_state(n_points))

    def snapshot_blob(self, n_points: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A private copy of the kernel blob for the current parameters (e.g. to sample with on another stream while
        training goes on).  Pass the previous snapshot as ``out`` to refresh it in place (one launch)."""
        st = self._fused_state(n_points)
        if out is None:
            out = st["blob"].clone()  # carries the frequency table and the descriptor tail
        tb = st["tables"]
        P = hip_ops._ptr
        _lib.check(_lib.load().pfm_wn_pack(P(self.fp.flat), P(tb.rows), tb.n_rows, P(tb.dst1), P(tb.dst2),
                                           P(tb.bias_param), P(tb.bias_blob), tb.n_bias, P(out),
                                           hip_ops._stream_ptr(self.fp.flat.device)), "pfm_wn_pack")
        return st["layout"].finish_blob(out)

    def fused_loss_and_grad(self, x, mask, cond) -> torch.Tensor:
        """pack -> loss forward -> backward -> d(weight_g, weight_v, bias) accumulated into the flat gradient;
        six launches plus three scalar torch ops, no autograd graph."""
        lib = _lib.load()
        P, S = hip_ops._ptr, hip_ops._stream_ptr(self.fp.flat.device)
        loss_mod = self.module.loss
        kind = {"ConditionalFlowMatchingLoss": "CFM", "DroidLoss": "droid", "DiffusionLoss": "diffusion"}.get(
            type(loss_mod).__name__, "FM-OT")
        if kind in ("CFM", "diffusion") and mask is None:
            raise TypeError(f"{type(loss_mod).__name__} needs a mask (losses.py:119, 247)")
        st = self._fused_state(x.shape[1])
        lay, tb, gblob = st["layout"], st["tables"], st["gblob"]
        blob = self._pack(st)
        B = x.shape[0]
        condf = None if lay.cfg.global_cond_dim == 0 else cond.to(torch.float32).contiguous()
        maskf = None if mask is None else mask.reshape(B, -1).to(torch.float32).contiguous()  # converted once, used by both kernels
        # [loss, 1 / sum(mask)] of THIS step in a buffer of its own (the caching allocator hands one out without a launch): the caller may
        # keep the returned loss -- a view of it -- as long as it likes (tests/test_hip_trainer_state.py holds 40 of them), and the step
        # needs neither a copy of the scalar nor a ring whose slots come round again
        fin = torch.empty(2, device=x.device, dtype=torch.float32)
        if kind == "diffusion":
            from .fm_loss import MLE_LOSS_WEIGHT
            t, z = loss_mod.draw(x, mask, land=self._land)
            sr, nr, beta = hip_ops.diffusion_schedule(t.to(torch.float32), **loss_mod.diff_config)
            jet_w = (1.0 + MLE_LOSS_WEIGHT * (beta / nr)).contiguous()
            parts, count, saved = hip_ops.epic_diffusion_loss_forward(lay, blob, x, t, z, torch.stack([sr, nr], dim=1), cond, maskf,
                                                                      loss_mod.criterion)
            _lib.check(lib.pfm_loss_finish(P(parts), P(count), P(jet_w), B, P(fin), S), "pfm_loss_finish")
            bw_kw = dict(criterion=loss_mod.criterion, jet_w=jet_w)
        else:
            if kind == "CFM":
                t, z, eps = loss_mod.draw(x, land=self._land)
            else:
                (t, z), eps = loss_mod.draw(x, land=self._land), None
            parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, cond, maskf, loss_mod.sigma, kind, eps)
            _lib.check(lib.pfm_loss_finish(P(parts), P(count), P(None), B, P(fin), S), "pfm_loss_finish")
            bw_kw = {}
        loss = fin[0]
        unpack = lib.pfm_wn_unpack_grad_set if tb.covers_all else lib.pfm_wn_unpack_grad

        def unpack_rows(r0, r1, b0, b1):
            _lib.check(unpack(P(self.fp.flat), P(gblob), P(tb.rows[r0:r1]), r1 - r0, P(tb.gsrc), P(tb.bias_gblob[b0:b1]),
                              P(tb.bias_param[b0:b1]), b1 - b0, P(self.fp.grad), S), "pfm_wn_unpack_grad")

        split = bool(self.split_backward)
        nf = self.fp.n_first
        if split and tb.covers_all and 0 < nf < self.fp.numel and B <= hip_ops.BWD_CHUNK_JETS:
            # DDP's overlap of the gradient exchange with the backward (configs/trainer/ddp.yaml:4-9), two buckets: the chain phase
            # finishes the gradients of the Linears without a 128x128 particle block -- they sit in front of the flat buffer -- and
            # their all-reduce runs next to the dW GEMM of the others
            hip_ops.epic_loss_backward_phase(lay, blob, condf, maskf, saved, fin[1:], st["one"], gblob, hip_ops.BWD_PHASE_CHAIN, **bw_kw)
            unpack_rows(0, tb.n_early_rows, 0, tb.n_early_bias)
            w_early = self.sync.start(self.fp.grad[:nf])
            hip_ops.epic_loss_backward_phase(lay, blob, condf, maskf, saved, fin[1:], st["one"], gblob, hip_ops.BWD_PHASE_DW, **bw_kw)
            unpack_rows(tb.n_early_rows, tb.n_rows, tb.n_early_bias, tb.n_bias)
            w_late = self.sync.start(self.fp.grad[nf:])
            self.sync.finish(w_early)
            self.sync.finish(w_late)
            self._grad_synced = True
        else:
            hip_ops.epic_loss_backward(lay, blob, condf, maskf, saved, fin[1:], st["one"], gblob, **bw_kw)
            unpack_rows(0, tb.n_rows, 0, tb.n_bias)
        return loss

    def step(self, batch, fused: bool = True) -> torch.Tensor:
        x, mask, cond = batch
        if not self.fp.is_intact():
            self._rebuild()
        if getattr(getattr(self.module, "hparams", None), "use_normaliser", False):  # training_step's pre-processing (:514-518)
            x, cond = self.module._normalise(x, mask, cond)
        if fused and self._fused is not None and not self.module.flows[0].net.is_wide(x.shape[1]):
            if not self._fused_state(x.shape[1])["tables"].covers_all:
                self.fp.grad.zero_()  # (never for the EPiC network: its unpack kernel writes every gradient element)
            loss = self.fused_loss_and_grad(x, mask, cond)
        else:
            self.fp.zero_grad()
            loss = self.module.loss(x, mask=mask, cond=cond)
            loss.backward()
        if self._grad_synced:  # fused_loss_and_grad has exchanged the two halves already
            self._grad_synced = False
            mul = 1.0 / self.sync.world
        else:
            mul = self.sync.sync(self.fp.grad)
        self.optimizer_step(mul)
        return loss.detach()

    def _rebuild(self):
        """The parameters no longer alias the flat buffer (module.to(device), a parameter replaced): re-home them, and move the
        optimiser / EMA state along (same element order: FlatParams keeps the parameter list and the offsets)."""
        self.fp.rebuild()
        dev = self.fp.flat.device
        if dev.type != "cuda":
            raise RuntimeError("FusedFMTrainer needs the module on a ROCm device (no CPU fallback)")
        self.exp_avg, self.exp_avg_sq, self.scratch = self.exp_avg.to(dev), self.exp_avg_sq.to(dev), self.scratch.to(dev)
        if self.ema is not None:
            self.ema = self.ema.to(dev)
        if self._fused is not None:
            self._fused = {}

    # ---- checkpoint / resume (what Lightning saves for the reference: optimizer state + the EMA callback's weights) ----
    def _names(self) -> List[str]:
        by_id = {id(p): n for n, p in self.module.named_parameters()}
        return [by_id[id(p)] for p in self.fp.params]

    def state_dict(self) -> Dict[str, object]:
        """Everything a resume needs besides ``module.state_dict()``: Adam moments, step count (bias correction, schedule),
        EMA weights, hyper-parameters; flat tensors in the order of ``param_names`` (CPU copies)."""
        if not self.fp.is_intact():
            self._rebuild()
        return {
            "step_count": self.step_count, "param_names": self._names(), "offsets": list(self.fp.offsets), "numel": self.fp.numel,
            "exp_avg": self.exp_avg.detach().cpu().clone(), "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(),
            "ema": None if self.ema is None else self.ema.detach().cpu().clone(),
            "hparams": {"lr": self.lr, "weight_decay": self.weight_decay, "betas": tuple(self.betas), "eps": self.eps,
                        "max_grad_norm": self.max_grad_norm, "ema_decay": self.ema_decay},
        }

    def load_state_dict(self, sd: Dict[str, object], load_hparams: bool = True) -> None:
        if not self.fp.is_intact():
            self._rebuild()
        names = self._names()
        if sorted(sd["param_names"]) != sorted(names):
            raise ValueError("trainer state was saved for a different parameter list")
        dev = self.fp.flat.device

        def load(dst: torch.Tensor, src: torch.Tensor):
            if list(sd["param_names"]) == names and int(sd["numel"]) == self.fp.numel:
                dst.copy_(src.to(dev))
                return
            # saved under another order of the flat buffer (an earlier build, or a module without the early / late grouping): by name
            so = dict(zip(sd["param_names"], sd["offsets"]))
            for n, p, off in zip(names, self.fp.params, self.fp.offsets):
                dst[off:off + p.numel()].copy_(src[so[n]:so[n] + p.numel()].to(dev))

        load(self.exp_avg, sd["exp_avg"])
        load(self.exp_avg_sq, sd["exp_avg_sq"])
        if sd.get("ema") is not None and self.ema is not None:
            load(self.ema, sd["ema"])
        self.step_count = int(sd["step_count"])
        if load_hparams:
            hp = sd["hparams"]
            self.lr, self.weight_decay, self.betas, self.eps = hp["lr"], hp["weight_decay"], tuple(hp["betas"]), hp["eps"]
            self.max_grad_norm = hp["max_grad_norm"]
            if self.ema is not None:
                self.ema_decay = hp["ema_decay"]

    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        """``module.state_dict()`` with every trainable parameter replaced by its EMA value: what the reference's
        ``-EMA.ckpt`` holds under "state_dict" (callbacks/ema.py:145-157, EMAModelCheckpoint)."""
        if self.ema is None:
            raise RuntimeError("EMA is off (ema_decay=None)")
        out = {k: v.detach().clone() for k, v in self.module.state_dict().items()}
        names = self._names()
        for name, p, off in zip(names, self.fp.params, self.fp.offsets):
            out[name] = self.ema[off:off + p.numel()].view_as(p).detach().clone()
        return out

    @contextlib.contextmanager
    def swap_ema(self):
        """``with trainer.swap_ema(): validate / sample`` -- the EMA callback's replace_model_weights / restore_original_weights
        (callbacks/ema.py:145-176): inside the block the module's parameters ARE the EMA weights (one device copy each way)."""
        if self.ema is None:
            raise RuntimeError("EMA is off (ema_decay=None)")
        if not self.fp.is_intact():
            self._rebuild()
        keep = self.fp.flat.detach().clone()
        self.fp.flat.copy_(self.ema)
        try:
            yield self.module
        finally:
            self.fp.flat.copy_(keep)

    def grad_norm(self) -> torch.Tensor:
        """global L2 norm of the last (scaled) gradient, as clip_grad_norm_ saw it"""
        return self.scratch[0].sqrt()


def cosine_warmup(warmup: int, max_iters: int) -> Callable[[int], float]:
    """The reference's CosineWarmupScheduler.get_lr_factor (schedulers/lr_scheduler.py:17-21) as an ``lr_schedule`` for
    :class:`FusedFMTrainer`: 0.5 (1 + cos(pi k / max_iters)), times k / warmup while k <= warmup.  The reference steps it once per
    EPOCH (flow_matching_module.py:626-633, ``interval: "epoch"``): pass ``lambda k: f(k // steps_per_epoch)`` for that."""
    def factor(k: int) -> float:
        f = 0.5 * (1.0 + math.cos(math.pi * k / max_iters))
        if k <= warmup:
            f *= k * 1.0 / warmup
        return f
    return factor
