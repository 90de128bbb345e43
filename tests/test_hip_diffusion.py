"""loss_type="diffusion" (configs/model/diffusion.yaml) on the jet-resident EPiC path: DiffusionLoss forward + every parameter
gradient, the probability-flow ODE samplers, DDIM and Euler-Maruyama, against the reference's recorded vectors."""
import copy

import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("crit", ["huber", "mse"])
def test_loss_and_all_parameter_gradients(crit):
    from particle_fm_amd.fm_loss import epic_diffusion_loss
    from particle_fm_amd.layout import EpicLayout
    from tests.test_layout_cpu import cfg_of
    g = load_golden("diffusion")
    lay = EpicLayout(cfg_of(g.hp))
    state = {k: v.clone().cuda().requires_grad_(k != "flows.0.frequencies") for k, v in g.state.items()}
    src = lay.source_vector(state, "flows.0.net.", freqs=g.freqs.cuda())
    tag = f"loss_{crit}/"
    x, t, z, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "z", "mask", "cond"))
    loss = epic_diffusion_loss(lay, src, x, t, z, cond=cond, mask=mask, criterion=crit, diff_config=g.hp["diff_config"])
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
    loss.backward()
    ref = g.grads(tag)
    for k, want in ref.items():
        got = state[k].grad.cpu()
        assert float((got - want).norm()) <= 5e-4 * float(want.norm()) + 1e-6, k


def test_samplers_match_reference_vectors():
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.test_layout_cpu import cfg_of
    g = load_golden("diffusion")
    lay = EpicLayout(cfg_of(g.hp))
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    dc = g.hp["diff_config"]
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k).cuda() for k in ("z", "mask", "cond"))
        xe = hip_ops.epic_sample_rk(lay, blob, z, cond, mask, ode_steps=steps, solver="midpoint", diff_config=dc).cpu()
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=2e-4)


def test_module_surface_ddim_em_and_training_step():
    from oracle import diffusion_ref as dr
    from oracle.fm_ref import EpicVectorField
    from particle_fm_amd.layout import EpicLayout
    from tests.test_layout_cpu import cfg_of
    g = load_golden("diffusion")
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, criterion="huber", **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    m = m.cuda()
    freqs = EpicLayout(cfg_of(g.hp)).default_freqs()  # the product's table; the oracle gets the same one
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=freqs)
    dc, n = g.hp["diff_config"], int(g.z["n_steps"])
    z, mask, cond = (g.get("ddim/" + k) for k in ("z", "mask", "cond"))
    zc = (z * mask).cuda()
    out = m(zc, cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="ddim", ode_steps=n).cpu()
    torch.testing.assert_close(out, dr.ddim_sample(vf, z * mask, cond, mask, n, dc), rtol=1e-3, atol=2e-4)
    # Euler-Maruyama: same device generator state -> same per-step draws as the module made
    torch.manual_seed(31)
    out = m(zc, cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="em", ode_steps=n).cpu()
    torch.manual_seed(31)
    noises = [torch.randn_like(zc).cpu() for _ in range(n)]
    torch.testing.assert_close(out, dr.em_sample(vf, z * mask, cond, mask, n, dc, noises), rtol=1e-3, atol=2e-4)
    # probability-flow ODE with the tuned names
    out = m(zc, cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="midpoint", ode_steps=8).cpu()
    from oracle.fm_ref import midpoint_trajectory_end
    with torch.no_grad():
        ref = midpoint_trajectory_end(lambda tt, xx: dr.diffusion_rhs(vf, tt, xx, cond, mask, dc), z * mask, torch.linspace(1.0, 0.0, 8))
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=2e-4)
    # training_step: the reference's draws replayed (t from the CPU generator, z on x's device)
    x = g.get("loss_huber/x").cuda()
    torch.manual_seed(77)
    loss = m.training_step((x, mask.cuda(), cond.cuda()), 0)["loss"]
    torch.manual_seed(77)
    t = torch.rand_like(torch.ones(x.shape[0]))
    zz = (torch.randn_like(x) * mask.cuda()).cpu()
    ref_loss, *_ = dr.diffusion_loss(vf, x.cpu(), mask, cond, t, zz, "huber", dc)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss, rtol=2e-5, atol=1e-6)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.flows[0].net.parameters())


def test_fused_trainer_step_equals_autograd_path():
    """FusedFMTrainer's graph-free diffusion step (pack, loss, backward, weight-norm unpack) against the autograd path."""
    from particle_fm_amd.engine import FusedFMTrainer
    from particle_fm_amd.models import SetFlowMatchingLitModule
    g = load_golden("diffusion")
    x, mask, cond = (g.get("loss_huber/" + k).cuda() for k in ("x", "mask", "cond"))
    grads = []
    for fused in (True, False):
        m = SetFlowMatchingLitModule(optimizer=None, criterion="huber", **copy.deepcopy(g.hp))
        full = dict(g.state)
        full.update({"loss." + k: v for k, v in g.state.items()})
        m.load_state_dict(full)
        tr = FusedFMTrainer(m.cuda(), lr=1e-3, weight_decay=0.0, max_grad_norm=None, ema_decay=None)
        torch.manual_seed(5)
        loss = tr.step((x, mask, cond), fused=fused)
        grads.append((loss.cpu(), tr.fp.grad.clone().cpu()))
    torch.testing.assert_close(grads[0][0], grads[1][0], rtol=1e-6, atol=1e-7)
    assert float((grads[0][1] - grads[1][1]).norm()) <= 1e-4 * float(grads[1][1].norm())


@pytest.mark.parametrize("crit", ["huber", "mse"])
def test_row_matrix_path_loss_and_gradients(crit):
    """The same reference vectors through the row-matrix EPiC path (what a diffusion model with a set beyond the LDS tile, or a
    hidden width other than 128, runs on): DiffusionLoss forward and every parameter gradient."""
    from particle_fm_amd.fm_loss_wide import epic_wide_diffusion_loss
    from particle_fm_amd.layout_wide import EpicWideLayout
    from tests.test_layout_cpu import cfg_of
    g = load_golden("diffusion")
    lay = EpicWideLayout(cfg_of(g.hp))
    state = {k: v.clone().cuda().requires_grad_(k != "flows.0.frequencies") for k, v in g.state.items()}
    src = lay.source_vector(state, "flows.0.net.", freqs=g.freqs.cuda())
    tag = f"loss_{crit}/"
    x, t, z, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "z", "mask", "cond"))
    loss = epic_wide_diffusion_loss(lay, src, x, t, z, cond=cond, mask=mask, criterion=crit, diff_config=g.hp["diff_config"])
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
    loss.backward()
    for k, want in g.grads(tag).items():
        got = state[k].grad.cpu()
        assert float((got - want).norm()) <= 5e-4 * float(want.norm()) + 1e-6, k


def test_row_matrix_path_samplers():
    from oracle import diffusion_ref as dr
    from oracle.fm_ref import EpicVectorField
    from particle_fm_amd import hip_ops_wide
    from particle_fm_amd.layout_wide import EpicWideLayout
    from tests.test_layout_cpu import cfg_of
    g = load_golden("diffusion")
    lay = EpicWideLayout(cfg_of(g.hp))
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    dc = g.hp["diff_config"]
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k).cuda() for k in ("z", "mask", "cond"))
        xe = hip_ops_wide.ew_sample_rk(lay, blob, z, cond, mask, ode_steps=steps, solver="midpoint", diff_config=dc).cpu()
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=2e-4)
    # module surface at a set size beyond the LDS tile: construct, train a step, sample with ddim / the ODE (finite, masked)
    import copy
    from particle_fm_amd.models import SetFlowMatchingLitModule
    hp = copy.deepcopy(g.hp)
    hp["num_particles"] = 170
    m = SetFlowMatchingLitModule(optimizer=None, criterion="huber", **hp).cuda()
    assert m.flows[0].net.wide
    gen = torch.Generator().manual_seed(3)
    B, N = 3, 170
    n = torch.tensor([170, 40, 99])
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, hp["features"], generator=gen) * mask
    cond = torch.randn(B, hp["global_cond_dim"], generator=gen) if hp["global_cond_dim"] else torch.zeros(B)
    loss = m.training_step((x.cuda(), mask.cuda(), cond.cuda()), 0)["loss"]
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.flows[0].net.parameters())
    state = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("flows.")}
    vf = EpicVectorField(state, "flows.0.net", hp, freqs=m.flows[0].net.layout().default_freqs())
    zc = (torch.randn(B, N, hp["features"], generator=gen) * mask)
    cc = cond if hp["global_cond_dim"] else None
    with torch.no_grad():
        out = m(zc.cuda(), cond=None if cc is None else cc.cuda(), mask=mask.cuda(), reverse=True, ode_solver="ddim", ode_steps=4).cpu()
    torch.testing.assert_close(out, dr.ddim_sample(vf, zc, cc, mask, 4, hp["diff_config"]), rtol=1e-3, atol=2e-4)
