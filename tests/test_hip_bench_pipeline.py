"""bench.py's overlapped step pipeline computes what the sequential one does.

`bench.StepLoop` keeps `--overlap` sampler launches in flight on streams of their own, each reading a snapshot of the weights of
its step while the next train step already updates the parameters (event-guarded snapshot slots).  Nothing of that may change a
number: from the same seed, `--overlap 2` and `--overlap 1` must end with bit-identical parameters, optimiser state and samples
(the train step is deterministic: its gradient is reduced in a fixed order, no atomics).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(overlap, steps, B=48, ode_steps=6):
    import bench
    from particle_fm_amd.engine import FusedFMTrainer
    from particle_fm_amd.models import SetFlowMatchingLitModule

    dev = torch.device("cuda", 0)
    torch.manual_seed(12345)
    torch.cuda.manual_seed_all(12345)
    model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
    trainer = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    N, F = bench.HP["num_particles"], bench.HP["features"]
    x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(B, N, F, 12345))
    z = (torch.randn(B, N, F, generator=torch.Generator().manual_seed(9999)) * mask.cpu()).to(dev)
    loop = bench.StepLoop(model, trainer, (x, mask, cond), z, ode_steps, overlap, dev)
    samples = []
    for i in range(steps):
        loop.step(i)
        # reading sample i-1 here (behind its own event) keeps the pipeline as deep as in the bench
        if i > 0:
            loop.done[(i - 1) % loop.S].synchronize()
            samples.append(loop.outs[(i - 1) % loop.S].clone())
    torch.cuda.synchronize(dev)
    samples.append(loop.outs[(steps - 1) % loop.S].clone())
    return trainer.fp.flat.clone(), trainer.exp_avg.clone(), trainer.ema.clone(), samples


def test_overlapped_pipeline_equals_sequential_bitwise():
    steps = 4
    p1, m1, e1, s1 = _run(1, steps)
    p2, m2, e2, s2 = _run(2, steps)
    p3, m3, e3, s3 = _run(3, steps)
    for (p, m, e, s) in ((p2, m2, e2, s2), (p3, m3, e3, s3)):
        assert torch.equal(p1, p), f"parameters differ: max {float((p1 - p).abs().max()):.3e}"
        assert torch.equal(m1, m) and torch.equal(e1, e)
        assert len(s1) == len(s) == steps
        for i, (a, b) in enumerate(zip(s1, s)):
            assert torch.equal(a, b), f"sample of step {i} differs: max {float((a - b).abs().max()):.3e}"
    # and the samples do depend on the step's weights (the test would be vacuous otherwise)
    assert not torch.equal(s1[0], s1[-1])
