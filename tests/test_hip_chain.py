"""n_transforms = 2 (flow_matching_module.py:421-443): two EPiC flows.  The losses feed the first flow's output to the second at the same
time t (losses.py:66-69, 125-128), sampling decodes through the flows in reverse order (:485-487).  No fused loss kernel for a chain:
every flow is the differentiable field of fm_field.py, whose backward (pfm_epic_fm_loss_backward_dx) returns the gradient w.r.t. the
parameters and w.r.t. the particle input.  Against vectors recorded from the reference (tests/golden/epic_chain2.npz; epic_chain2w.npz: the
same chain at hidden 136, i.e. on the row-matrix EPiC kernels, pfm_ew_fm_loss_backward_dx)."""
import copy

import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _module(g):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    hp = copy.deepcopy(g.hp)
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **hp)
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    m = m.cuda()
    m.set_freq_table(g.freqs)
    return m


@pytest.mark.parametrize("fixture", ["chain2", "chain2w"])
def test_field_gradient_wrt_input_matches_the_oracle(fixture):
    """d <G, f(t, x)> / d x and / d parameters of ONE flow against the oracle's autograd (the piece the chain is built from)."""
    from oracle.fm_ref import EpicVectorField
    g = load_golden(fixture)
    m = _module(g)
    assert m.flows[1].net.is_wide(g.hp["num_particles"]) == (fixture == "chain2w")
    tag = "loss_fm/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    gen = torch.Generator().manual_seed(3)
    G = torch.randn(x.shape, generator=gen)
    xr = x.clone().requires_grad_(True)
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items() if k.startswith("flows.1.")}
    vf = EpicVectorField(st, "flows.1.net", g.hp, freqs=g.freqs)
    v_ref = vf(t[:, None].expand(-1, x.shape[1]), xr, cond=cond, mask=mask)
    (v_ref * G).sum().backward()
    xc = x.cuda().requires_grad_(True)
    v = m.flows[1].field(t.cuda(), xc, cond=cond.cuda(), mask=mask.cuda())
    torch.testing.assert_close(v.detach().cpu(), v_ref.detach(), atol=2e-5, rtol=2e-4)
    (v * G.cuda()).sum().backward()
    torch.testing.assert_close(xc.grad.cpu(), xr.grad, atol=2e-5, rtol=2e-3)
    assert torch.all(xc.grad.cpu()[mask.squeeze(-1) == 0] == 0)
    named = dict(m.flows[1].named_parameters())
    for k, p in st.items():
        if p.grad is None:
            continue
        got = named[k[len("flows.1."):]].grad.cpu()
        assert float((got - p.grad).norm()) <= 5e-4 * float(p.grad.norm()) + 1e-6, k


@pytest.mark.parametrize("fixture", ["chain2", "chain2w"])
@pytest.mark.parametrize("name", ["fm", "cfm"])
def test_chained_loss_and_all_parameter_gradients(name, fixture):
    g = load_golden(fixture)
    m = _module(g)
    assert len(m.flows) == 2
    tag = f"loss_{name}/"
    x, t, a, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "a", "mask", "cond"))
    from particle_fm_amd.models.components.losses import _chained_loss
    eps = g.get(tag + "eps").cuda() if name == "cfm" else None
    loss = _chained_loss(m.flows, {"fm": "FM-OT", "cfm": "CFM"}[name], x, t, a, eps, mask, cond, 1e-4)
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=3e-5, atol=1e-6)
    loss.backward()
    named = dict(m.flows.named_parameters())
    bad = []
    for k, want in g.grads(tag).items():
        got = named[k[len("flows."):]].grad.cpu()
        rel = float((got - want).norm()) / max(float(want.norm()), 1e-12)
        if not rel < 1e-3:
            bad.append((k, rel))
    assert not bad, bad[:8]
    assert len(g.grads(tag)) == 2 * len(list(m.flows[0].parameters()))  # every parameter of both flows


@pytest.mark.parametrize("fixture", ["chain2", "chain2w"])
def test_training_step_and_sampling_through_both_flows(fixture):
    g = load_golden(fixture)
    m = _module(g)
    tag = "loss_fm/"
    x, mask, cond = (g.get(tag + k).cuda() for k in ("x", "mask", "cond"))
    torch.manual_seed(2468)  # the recording's seed: t on the CPU generator as the reference draws it; z on the device differs
    loss = m.training_step((x, mask, cond), 0)["loss"]
    assert torch.isfinite(loss)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.flows.parameters())
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mk, c = (g.get(tag + k) for k in ("z", "mask", "cond"))
        out = m((z * mk).cuda(), cond=c.cuda(), mask=mk.cuda(), reverse=True, ode_solver="midpoint", ode_steps=steps).cpu()
        torch.testing.assert_close(out, g.get(tag + "x_end"), atol=5e-5, rtol=1e-3)
