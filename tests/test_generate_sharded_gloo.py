"""generate_data_sharded on CPU (gloo, world_size 2): the rank-sharded sampling driver returns the array the single-process
generate_data returns, bit for bit -- batch plan, z drawn in lock-step from the CPU generator, cond / mask row slices, remainder
batch, gather order.  The network is a stand-in here (the HIP sampler has no CPU path; tests/test_hip_generate.py runs the real
one, rank by rank, on the GPU); the epilogue is the oracle's CPU restatement of data_generation.py:94-123."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N, F, TOTAL, BS = 12, 3, 103, 16  # 6 full batches + a remainder of 7


class StubModel(torch.nn.Module):
    """SetFlowMatchingLitModule's sampling surface: .hparams, .to(), .sample(n, cond, mask, ...) drawing z on the CPU generator."""

    def __init__(self):
        super().__init__()
        self.hparams = SimpleNamespace(num_particles=N, features=F, use_normaliser=False)
        self.w = torch.nn.Parameter(torch.linspace(-1, 1, F))

    @torch.no_grad()
    def sample(self, n_samples, cond=None, mask=None, ode_solver="midpoint", ode_steps=100, **kw):
        z = torch.randn(n_samples, N, F)  # flow_matching_module.py:659-663
        if mask is not None:
            z = z * mask[:n_samples]
        return torch.tanh(z * self.w) + (0 if cond is None else cond[:, None, :1]) - z  # anything deterministic in (z, cond, mask)


def _cpu_epilogue(x, mask=None, scale=None, shift=None, log_pt_col=-1):
    from oracle.fm_ref import generate_epilogue
    means = None if shift is None else shift.tolist()
    stds = None if scale is None else (scale * 5).tolist()
    y = generate_epilogue(x, mask if mask is not None else torch.ones_like(x[..., :1]), scale is not None, 5, means, stds,
                          log_pt_col >= 0, False, mask is not None)
    x.copy_(y)
    return x


def _inputs():
    gen = torch.Generator().manual_seed(4)
    n = torch.randint(2, N + 1, (TOTAL,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    cond = torch.randn(TOTAL, 2, generator=gen)
    return mask, cond


KW = dict(device="cpu", variable_set_sizes=True, normalized_data=True, means=[0.1, -0.2, 0.3], stds=[1.5, 0.7, 2.0], log_pt=True,
          verbose=False, ode_steps=5)


def _single():
    from particle_fm_amd.utils import data_generation as dg
    dg.sample_epilogue_ = _cpu_epilogue
    mask, cond = _inputs()
    torch.manual_seed(9999)
    data, _ = dg.generate_data(StubModel(), TOTAL, batch_size=BS, cond=cond, mask=mask, **KW)
    return data, torch.get_rng_state()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from particle_fm_amd.utils import data_generation as dg
        dg.sample_epilogue_ = _cpu_epilogue
        torch.set_num_threads(1)
        mask, cond = _inputs()
        torch.manual_seed(9999)  # every rank enters with the same CPU generator state (seed_everything in the reference)
        full, _ = dg.generate_data_sharded(StubModel(), TOTAL, batch_size=BS, cond=cond, mask=mask, **KW)
        state = torch.get_rng_state()
        torch.manual_seed(9999)
        (rows, index), _ = dg.generate_data_sharded(StubModel(), TOTAL, batch_size=BS, cond=cond, mask=mask, gather=False, **KW)
        q.put((rank, full.tobytes(), state.numpy().tobytes(), rows.tobytes(), index.tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_generation_equals_single_process_bitwise():
    want, want_state = _single()
    assert want.shape == (TOTAL, N, F)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    seen = np.zeros(TOTAL, dtype=int)
    for rank, full, state, rows, index in got:
        assert full == want.tobytes(), f"rank {rank}: gathered array differs from the single-process one"
        assert state == want_state.numpy().tobytes(), "CPU generator left in a different state"
        assert rows == want[index].tobytes()
        seen[index] += 1
    assert (seen == 1).all()  # the shards partition the jet list


def test_shard_plan_partitions_rows():
    from particle_fm_amd.utils.data_generation import shard_plan
    for total, bs, world in ((103, 16, 2), (1000, 256, 8), (5, 256, 4), (512, 256, 3), (0, 16, 2)):
        plan = shard_plan(total, bs, world)
        rows = [i for a, b, _ in plan for i in range(a, b)]
        assert rows == list(range(total))
        assert all(r == i % world for i, (_, _, r) in enumerate(plan))
