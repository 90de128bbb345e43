"""t_emb="gaussian" (flow_matching_module.py:178-181, 213-221): the learned time embedding in front of the EPiC field.
The embedding network runs as host-side torch ops on the device, its output goes to the kernels (pfm_epic_*_temb) and the loss
backward returns d loss / d temb, so its parameters train.  Checked against the reference's recorded vectors (epic_gauss.npz)."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, sample_fixed_step
from tests.conftest import load_golden
from tests.test_modules_cpu import _yaml_kwargs


def _module(g, cuda=True):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **_yaml_kwargs(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    return m.cuda() if cuda else m


def test_state_dict_layout_is_the_reference_one():
    g = load_golden("gauss")
    m = _module(g, cuda=False)
    keys = [k for k in m.state_dict() if k.startswith("flows.")]
    assert keys == g.keys
    assert not m.flows[0].embed[0].W.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_forward_matches_reference(mk):
    g = load_golden("gauss")
    m = _module(g)
    tag = f"nfe_{mk}/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    dev = lambda a: None if a is None else a.cuda()
    N = x.shape[1]
    with torch.no_grad():
        v = m.flows[0](dev(t.unsqueeze(-1).repeat_interleave(N, dim=1)), dev(x), cond=dev(cond), mask=dev(mask)).cpu()
        vs = m.flows[0](dev(t[0]), dev(x), cond=dev(cond), mask=dev(mask)).cpu()
        temb = m.flows[0].time_embedding(dev(t.unsqueeze(-1).repeat_interleave(N, dim=1)), dev(x), "gaussian")[:, 0, :].cpu()
    torch.testing.assert_close(temb, g.get(tag + "temb"), atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(vs, g.get(tag + "v_scalar_t"), atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("mk", ["f32", "none"])
def test_loss_and_all_gradients_including_the_embedding_network(mk):
    g = load_golden("gauss")
    m = _module(g)
    tag = f"loss_{mk}/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    dev = lambda a: None if a is None else a.cuda()
    loss = m.flows[0].fm_loss(dev(x), dev(t), dev(z), mask=dev(mask), cond=dev(cond), sigma=1e-4, kind="FM-OT")
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), atol=2e-6, rtol=2e-5)
    loss.backward()
    ref = g.grads(tag)
    named = {"flows.0." + k: p for k, p in m.flows[0].named_parameters()}
    assert set(ref) == {k for k, p in named.items() if p.requires_grad}
    for k, gref in ref.items():
        got = named[k].grad.cpu()
        scale = max(gref.abs().max().item(), 1e-8)
        assert (got - gref).abs().max().item() / scale <= 2e-4, k
    for k in ("flows.0.embed.1.weight", "flows.0.linear.weight"):
        assert ref[k].abs().max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [3, 10, 100])
def test_midpoint_matches_reference(steps):
    g = load_golden("gauss")
    m = _module(g)
    tag = f"midpoint_{steps}/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    with torch.no_grad():
        out = m(z.cuda() * mask.cuda(), cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="midpoint", ode_steps=steps).cpu()
    torch.testing.assert_close(out, g.get(tag + "x_end"), atol=5e-5, rtol=1e-4)


@pytest.mark.gpu
def test_rk4_and_trainer_step():
    from particle_fm_amd.engine import FusedFMTrainer
    g = load_golden("gauss")
    m = _module(g)
    tag = "midpoint_10/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    vf = EpicVectorField(g.state, "flows.0.net", g.hp)
    ref = sample_fixed_step(vf, z, cond, mask, ode_steps=6, solver="rk4")
    with torch.no_grad():
        out = m(z.cuda() * mask.cuda(), cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="rk4", ode_steps=6).cpu()
    torch.testing.assert_close(out, ref, atol=5e-5, rtol=1e-4)
    # the (non-fused) trainer moves the embedding network's parameters
    tr = FusedFMTrainer(m, lr=1e-3, max_grad_norm=0.5)
    assert tr._fused is None
    w0 = m.flows[0].linear.weight.detach().clone()
    x = g.get("loss_f32/x").cuda()
    tr.step((x, g.get("loss_f32/mask").cuda(), g.get("loss_f32/cond").cuda()))
    assert not torch.equal(w0, m.flows[0].linear.weight.detach())
