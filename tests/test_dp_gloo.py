"""Multi-rank path on CPU (gloo, world_size 2): the flat gradient all-reduce reproduces DDP's
mean-of-per-rank-gradients, with every rank normalising its loss by its own mask count (losses.py:75-76)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.fm_ref import EpicVectorField, fm_ot_loss
from tests.conftest import load_golden
from tests.test_modules_cpu import _yaml_kwargs


def _rank_grads(g, sl):
    tag = "loss_f32/"
    x, t, z, mask = g.get(tag + "x")[sl], g.get(tag + "t")[sl], g.get(tag + "z")[sl], g.get(tag + "mask")[sl]
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
    vf = EpicVectorField(st, "flows.0.net", g.hp, freqs=g.freqs)
    loss, *_ = fm_ot_loss(vf, x, mask, None, t, z, sigma=1e-4)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in st.items() if v.grad is not None}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from particle_fm_amd.engine import FlatParams, GradSync
        from particle_fm_amd.models import SetFlowMatchingLitModule
        torch.set_num_threads(2)
        g = load_golden("jetnet30")
        m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **_yaml_kwargs(g.hp))
        full = dict(g.state)
        full.update({"loss." + k: v for k, v in g.state.items()})
        m.load_state_dict(full)
        fp = FlatParams(m.parameters())
        sync = GradSync()
        assert sync.enabled and sync.world == world
        loss, grads = _rank_grads(g, slice(2 * rank, 2 * rank + 2))
        names = [k for k, _ in m.named_parameters()]
        fp.zero_grad()
        for k, p in m.named_parameters():
            p.grad.copy_(grads[k])
        mul = sync.sync(fp.grad)
        fp.grad.mul_(mul)
        mean_loss = sync.mean_scalar(loss)
        if rank == 0:
            l0, g0 = _rank_grads(g, slice(0, 2))
            l1, g1 = _rank_grads(g, slice(2, 4))
            for k, p in m.named_parameters():
                torch.testing.assert_close(p.grad, 0.5 * (g0[k] + g1[k]), atol=1e-7, rtol=1e-5)
            torch.testing.assert_close(mean_loss, 0.5 * (l0 + l1))
            assert fp.is_intact() and len(names) == 87
        # the two-bucket exchange of the fused step (engine.FusedFMTrainer.fused_loss_and_grad): early group in front of the flat buffer,
        # its all-reduce started (async) before the rest is there; same means as the one flat call
        from particle_fm_amd.engine import early_linear
        net = m.flows[0].net
        early = {id(p) for n, p in net.named_parameters() if early_linear(n.rsplit(".", 1)[0])}
        fp2 = FlatParams(m.parameters(), first=early)
        assert 0 < fp2.n_first < fp2.numel
        fp2.zero_grad()
        for k, p in m.named_parameters():
            if id(p) in early:
                p.grad.copy_(grads[k])
        w_early = sync.start(fp2.grad[: fp2.n_first])       # the late half is still zero on every rank
        for k, p in m.named_parameters():
            if id(p) not in early:
                p.grad.copy_(grads[k])
        w_late = sync.start(fp2.grad[fp2.n_first:])
        sync.finish(w_early)
        sync.finish(w_late)
        fp2.grad.mul_(1.0 / sync.world)
        if rank == 0:
            for k, p in m.named_parameters():
                torch.testing.assert_close(p.grad, 0.5 * (g0[k] + g1[k]), atol=1e-7, rtol=1e-5)
        # the direct two-phase exchange (PFM_DP_EXCHANGE=two_phase: all-to-all of slices, local sum, all-gather): the same means, and
        # the SAME BITS on every rank (each slice is summed by exactly one rank)
        tp = GradSync(exchange="two_phase")
        fp.zero_grad()
        for k, p in m.named_parameters():
            p.grad.copy_(grads[k])
        mul = tp.sync(fp.grad)
        fp.grad.mul_(mul)
        both = [torch.empty_like(fp.grad) for _ in range(world)]
        dist.all_gather(both, fp.grad)
        assert torch.equal(both[0], both[1])
        if rank == 0:
            for k, p in m.named_parameters():
                torch.testing.assert_close(p.grad, 0.5 * (g0[k] + g1[k]), atol=1e-7, rtol=1e-5)
            out.put("ok")
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_is_ddp_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"
