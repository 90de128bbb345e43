"""The data-parallel step's two-bucket gradient exchange (engine.FusedFMTrainer.fused_loss_and_grad; torch DDP's overlap of the
all-reduce with the backward under configs/trainer/ddp.yaml:4-9): pfm_epic_fm_loss_backward_phases splits the backward into the chain
phase (every gradient final except the 128x128 particle blocks) and the dW phase; the trainer reduces the first bucket while the dW
GEMM runs.  The split must not change a bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B=24, n_points=60, seed=3):
    import bench
    from particle_fm_amd.models import SetFlowMatchingLitModule
    torch.manual_seed(12345)
    hp = dict(bench.HP, num_particles=n_points)
    m = SetFlowMatchingLitModule(optimizer=None, **hp).cuda()
    x, mask, cond = (a.cuda() for a in bench.synthetic_batch(B, n_points, 3, seed))
    return m, (x, mask, cond)


@pytest.mark.parametrize("crit", [None, "huber"])
def test_two_phases_equal_the_whole_backward(crit):
    from particle_fm_amd import hip_ops
    m, (x, mask, cond) = _setup()
    net = m.flows[0].net
    lay, blob = net.layout(60), net.packed_weights(60)
    B = x.shape[0]
    maskf = mask.reshape(B, -1).float().contiguous()
    g = torch.Generator(device="cuda").manual_seed(5)
    t = torch.rand(B, device="cuda", generator=g)
    z = torch.randn(x.shape, device="cuda", generator=g)
    parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
    inv = (1.0 / count.sum()).reshape(1)
    one = torch.ones(1, device="cuda")
    jw = (0.5 + torch.rand(B, device="cuda", generator=g)) if crit else None
    kw = dict(criterion=crit, jet_w=jw) if crit else {}
    whole = torch.full_like(blob, float("nan"))
    hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, whole, **kw)
    halves = torch.full_like(blob, float("nan"))
    hip_ops.epic_loss_backward_phase(lay, blob, None, maskf, saved, inv, one, halves, hip_ops.BWD_PHASE_CHAIN, **kw)
    gp = torch.as_tensor(lay.src_gpos, device="cuda").long()
    gp = gp[gp >= 0]
    after_chain = halves[gp].clone()
    # what is final after the chain phase: everything but the 128x128 particle blocks (still NaN here)
    frac_final = float(torch.isfinite(after_chain).float().mean())
    assert 0.4 < frac_final < 0.65, frac_final
    fin = torch.isfinite(after_chain)
    assert torch.equal(after_chain[fin], whole[gp][fin])
    hip_ops.epic_loss_backward_phase(lay, blob, None, maskf, saved, inv, one, halves, hip_ops.BWD_PHASE_DW, **kw)
    assert torch.isfinite(halves[gp]).all()
    assert torch.equal(halves[gp], whole[gp])


def test_split_step_equals_the_single_call_step():
    """Three optimiser steps with the backward split in two (what every rank of a data-parallel job runs) against the single-call
    step: identical parameters, Adam moments and EMA; the early bucket is the front of the flat buffer."""
    from particle_fm_amd.engine import FusedFMTrainer, early_linear
    outs = []
    for split in (False, True):
        m, batch = _setup()
        tr = FusedFMTrainer(m, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
        tr.split_backward = split
        torch.manual_seed(7)
        losses = [tr.step(batch).item() for _ in range(3)]
        names = tr._names()
        n_early = sum(1 for n in names if early_linear(n.rsplit(".", 1)[0]))
        assert all(early_linear(n.rsplit(".", 1)[0]) for n in names[:n_early]) and not any(early_linear(n.rsplit(".", 1)[0]) for n in names[n_early:])
        assert 0 < tr.fp.n_first < tr.fp.numel and tr.fp.n_first == tr.fp.offsets[n_early]
        outs.append((losses, tr.fp.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.ema.clone(), tr.state_dict()))
    (l0, p0, a0, s0, e0, sd0), (l1, p1, a1, s1, e1, _) = outs
    assert l0 == l1
    for a, b in ((p0, p1), (a0, a1), (s0, s1), (e0, e1)):
        assert torch.equal(a, b)
    # a trainer state saved under another order of the flat buffer (an earlier build) is loaded by name
    m, batch = _setup()
    tr = FusedFMTrainer(m, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    names, offs = list(sd0["param_names"]), list(sd0["offsets"])
    perm = list(reversed(range(len(names))))
    sizes = {n: p.numel() for n, p in zip(tr._names(), tr.fp.params)}
    new_offs, o = {}, 0
    for i in perm:
        new_offs[names[i]] = o
        o += (sizes[names[i]] + 3) & ~3
    shuffled = {k: v for k, v in sd0.items()}
    for key in ("exp_avg", "exp_avg_sq", "ema"):
        buf = torch.zeros(o)
        for n, off in zip(names, offs):
            buf[new_offs[n]:new_offs[n] + sizes[n]] = sd0[key][off:off + sizes[n]]
        shuffled[key] = buf
    shuffled["param_names"] = [names[i] for i in perm]
    shuffled["offsets"] = [new_offs[names[i]] for i in perm]
    shuffled["numel"] = o
    tr.load_state_dict(shuffled)
    for n, p, off in zip(tr._names(), tr.fp.params, tr.fp.offsets):
        so = dict(zip(names, offs))[n]
        assert torch.equal(tr.exp_avg[off:off + p.numel()].cpu(), sd0["exp_avg"][so:so + p.numel()])
