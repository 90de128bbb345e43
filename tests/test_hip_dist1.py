"""RCCL smoke on the one-GPU box: a world-size-1 "nccl" process group carries the flat gradient all-reduce of the
data-parallel step (the N > 1 arithmetic is covered on CPU by tests/test_dp_gloo.py)."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_flat_allreduce_over_rccl_world1():
    from particle_fm_amd.engine import GradSync
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sync = GradSync()
        assert not sync.enabled and sync.world == 1
        buf = torch.arange(561330, dtype=torch.float32, device="cuda")
        ref = buf.clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)  # the collective FusedFMTrainer issues when world > 1
        torch.cuda.synchronize()
        assert torch.equal(buf, ref)
        assert sync.sync(buf) == 1.0
        dist.barrier()
    finally:
        dist.destroy_process_group()
