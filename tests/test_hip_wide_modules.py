"""SetFlowMatchingLitModule at the JetClass configuration (hidden 300: the wide HIP path behind the same classes)."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, fm_ot_loss, sample_midpoint
from tests.test_modules_cpu import _yaml_kwargs

pytestmark = pytest.mark.gpu


def _module(g):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **_yaml_kwargs(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    return m.cuda()


def _oracle(g, m, state=None):
    return EpicVectorField(state or g.state, "flows.0.net", g.hp, freqs=m.flows[0].net.layout().default_freqs())


def test_forward_and_sample(wide_golden):
    g = wide_golden
    m = _module(g)
    # sincos fixture: hidden 128, N = 24 -> jet-resident kernel; lhco128: hidden 128 but N = 279 does not fit the LDS tile
    assert m.flows[0].net.wide == (g.hp["hidden_dim"] != 128 or g.hp["num_particles"] > 150)
    tag = "nfe_f32/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    N = x.shape[1]
    vf = _oracle(g, m)
    with torch.no_grad():
        ref = vf(t[:, None].expand(-1, N), x, cond=cond, mask=mask)
    tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
    v = m.flows[0](tt.cuda(), x.cuda(), cond=cond.cuda(), mask=mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=2e-5, rtol=2e-4)
    B, F = x.shape[0], x.shape[2]
    torch.manual_seed(9999)
    out = m.sample(B, cond=cond, mask=mask, ode_solver="midpoint", ode_steps=6).cpu()
    torch.manual_seed(9999)
    z = torch.randn(B, N, F)
    refs = sample_midpoint(vf, z, cond, mask, ode_steps=6)
    torch.testing.assert_close(out, refs, atol=2e-4, rtol=1e-3)
    assert torch.all(out[mask.squeeze(-1) == 0] == 0)


def test_training_step_and_fused_optimizer():
    from particle_fm_amd.engine import FusedFMTrainer
    from tests.conftest import load_wide_golden
    g = load_wide_golden("small")
    m = _module(g)
    tag = "loss_f32/"
    x, mask, cond = (g.get(tag + k) for k in ("x", "mask", "cond"))
    tr = FusedFMTrainer(m, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    torch.manual_seed(4321)
    loss = tr.step((x.cuda(), mask.cuda(), cond.cuda()))
    torch.manual_seed(4321)
    t = torch.rand_like(torch.ones(x.shape[0]))
    z = torch.randn_like(x.cuda()).cpu()
    ref = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if "frequencies" not in k}
    params = list(ref.values())
    l_ref, *_ = fm_ot_loss(_oracle(g, m, ref), x, mask, cond, t, z, sigma=1e-4)
    l_ref.backward()
    torch.testing.assert_close(loss.cpu(), l_ref.detach(), atol=2e-6, rtol=2e-5)
    gn = torch.nn.utils.clip_grad_norm_(params, 0.5)
    torch.testing.assert_close(tr.grad_norm().cpu(), gn, atol=1e-5, rtol=1e-3)


@pytest.mark.parametrize("activation", ["relu", "none"])
def test_other_activations_on_the_row_matrix_path(activation):
    """activation="relu" / a name torch.nn.functional lacks (= none, epic.py:180) at hidden 136 (the row-matrix kernels: the descriptor's
    slope 0 / 1): forward, FM-OT loss and every gradient against the oracle's autograd on the module's own (default-initialised) weights."""
    from particle_fm_amd.models import SetFlowMatchingLitModule
    torch.manual_seed(5)
    hp = dict(model="epic", features=3, hidden_dim=136, num_particles=20, frequencies=6, layers=2, latent=12, activation=activation,
              wrapper_func="weight_norm", t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="sincos", global_cond_dim=2,
              local_cond_dim=2, dropout=0.0, sum_scale=1e-2)
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **hp).cuda()
    assert m.flows[0].net.wide
    state = {"flows.0." + k: v.detach().cpu().clone() for k, v in m.flows[0].state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    B, N = 3, 20
    mask = (torch.arange(N)[None] < torch.tensor([20, 13, 7])[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, 3, generator=gen) * mask
    cond, t, z = torch.randn(B, 2, generator=gen), torch.rand(B, generator=gen), torch.randn(B, N, 3, generator=gen)
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in state.items()}
    vf = EpicVectorField(st, "flows.0.net", hp, freqs=state["flows.0.frequencies"])
    with torch.no_grad():
        want = vf(t[:, None].expand(-1, N), x, cond=cond, mask=mask)
        got = m.flows[0](t.cuda(), x.cuda(), cond=cond.cuda(), mask=mask.cuda()).cpu()
    torch.testing.assert_close(got, want, atol=2e-5, rtol=2e-4)
    ref, *_ = fm_ot_loss(vf, x, mask, cond, t, z, 1e-4)
    ref.backward()
    loss = m.flows[0].fm_loss(x.cuda(), t.cuda(), z.cuda(), mask=mask.cuda(), cond=cond.cuda(), sigma=1e-4, kind="FM-OT")
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=3e-5, atol=1e-6)
    loss.backward()
    for k, p in m.flows[0].named_parameters():
        want_g = st["flows.0." + k].grad
        assert float((p.grad.cpu() - want_g).norm()) <= 2e-3 * float(want_g.norm()) + 1e-6, k
