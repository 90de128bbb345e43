"""The MDMA drop-in modules on the GPU: reference call signatures in, reference numbers out."""
import copy

import pytest
import torch

from oracle.fm_ref import fm_ot_loss, sample_midpoint
from oracle.mdma_ref import MdmaVectorField, broadcast_field

pytestmark = pytest.mark.gpu


def _module(g, **over):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    kw = copy.deepcopy(g.hp)
    kw.update(over)
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **kw)
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    return m.cuda()


def _oracle(g, state=None):
    from particle_fm_amd.layout_mdma import default_freqs
    # the product fixes the frequency table; give the oracle the same one
    return MdmaVectorField(state or g.state, "flows.0.", g.hp, freqs=default_freqs(2 * g.hp["frequencies"], g.hp.get("t_emb", "cosine")))


def test_cnf_forward_reference_signature(mdma_golden):
    g = mdma_golden
    m = _module(g)
    tag = "nfe_int64/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))  # cond: the conditional fixtures' (B, 1)
    N = x.shape[1]
    vf = _oracle(g)
    with torch.no_grad():
        ref = vf(t[:, None].expand(-1, N), x, cond=cond, mask=mask)
        ref_s = vf(t[0], x, cond=cond, mask=mask)
    tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)  # losses.py:47 shape (B,N)
    dcond = torch.zeros(x.shape[0], 0).cuda() if cond is None else cond.cuda()
    v = m.flows[0](tt.cuda(), x.cuda(), cond=dcond, mask=mask.cuda()).cpu()
    assert v.shape == (x.shape[0], N, 1)  # ONE output per particle, like the reference
    torch.testing.assert_close(v, ref, atol=2e-5, rtol=2e-4)
    vs = m.flows[0](t[0].cuda(), x.cuda(), cond=None if cond is None else dcond, mask=mask.cuda()).cpu()  # 0-dim t (sampling)
    torch.testing.assert_close(vs, ref_s, atol=2e-5, rtol=2e-4)
    # with the recording host's frequency table the recorded vectors themselves come back
    m.set_freq_table(g.freqs)
    v = m.flows[0](tt.cuda(), x.cuda(), cond=None if cond is None else dcond, mask=mask.cuda()).cpu()
    torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)


def test_sample_matches_oracle():
    from tests.conftest import load_mdma_golden
    g = load_mdma_golden("small")
    m = _module(g)
    mask = g.get("midpoint_10/mask")
    B, N, F = mask.shape[0], g.hp["num_particles"], g.hp["features"]
    vf = broadcast_field(_oracle(g))
    for solver, steps in (("midpoint", 20), ("euler", 12), ("rk4", 8)):
        torch.manual_seed(9999)
        out = m.sample(B, mask=mask, ode_solver=solver, ode_steps=steps).cpu()
        torch.manual_seed(9999)
        z = torch.randn(B, N, F)
        with torch.no_grad():
            if solver == "midpoint":
                want = sample_midpoint(vf, z, None, mask, steps)
            else:
                from oracle.fm_ref import rk_trajectory_end
                want = rk_trajectory_end(lambda tt, xx: vf(tt, xx, None, mask), z * mask, torch.linspace(1.0, 0.0, steps), solver)
        torch.testing.assert_close(out, want, atol=2e-4, rtol=1e-3)
    with pytest.raises(NotImplementedError):
        m.sample(B, mask=mask, ode_solver="ieuler")


def test_training_step_replays_reference_draws(mdma_golden):
    """training_step -> FlowMatchingLoss.forward draws t on the CPU generator and z on x's device (losses.py:46-53); with the
    same seed the oracle's loss on the same draws is what comes back, and backward fills every used parameter's .grad."""
    g = mdma_golden
    m = _module(g)
    tag = "loss_f32/"
    x, mask = g.get(tag + "x").cuda(), g.get(tag + "mask").cuda()
    cond = g.get(tag + "cond")
    torch.manual_seed(77)
    loss = m.training_step((x, mask, None if cond is None else cond.cuda()), 0)["loss"]
    torch.manual_seed(77)
    t = torch.rand_like(torch.ones(x.shape[0]))
    z = torch.randn_like(x).cpu()
    ref, *_ = fm_ot_loss(broadcast_field(_oracle(g)), x.cpu(), mask.cpu(), cond, t, z, 1e-4)
    torch.testing.assert_close(loss.detach().cpu(), ref, rtol=2e-5, atol=1e-6)
    loss.backward()
    for k, p in m.flows[0].net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        assert ("cond_cls" in k) == (float(p.grad.abs().sum()) == 0.0 and p.numel() > 0 or p.numel() == 0), k


def test_fused_trainer_trains():
    """engine.FusedFMTrainer on the MDMA module (generic autograd path + fused clip / AdamW / EMA): the loss of a fixed batch with
    fixed draws goes down, checkpoint round trip included."""
    from particle_fm_amd.engine import FusedFMTrainer
    from tests.conftest import load_mdma_golden
    g = load_mdma_golden("small")
    m = _module(g)
    tr = FusedFMTrainer(m, lr=2e-3, weight_decay=0.0, max_grad_norm=0.5, ema_decay=0.999)
    x, mask = g.get("loss_f32/x").cuda(), g.get("loss_f32/mask").cuda()
    losses = []
    for _ in range(12):
        torch.manual_seed(5)
        losses.append(float(tr.step((x, mask, None))))
    assert all(l == l for l in losses) and losses[-1] < 0.9 * losses[0], losses
    sd, msd = tr.state_dict(), copy.deepcopy(m.state_dict())
    m2 = _module(g)
    m2.load_state_dict(msd)
    tr2 = FusedFMTrainer(m2, lr=2e-3, weight_decay=0.0, max_grad_norm=0.5, ema_decay=0.999)
    tr2.load_state_dict(sd)
    torch.manual_seed(5)
    a = float(tr.step((x, mask, None)))
    torch.manual_seed(5)
    b = float(tr2.step((x, mask, None)))
    assert abs(a - b) <= 1e-6 * abs(a)
