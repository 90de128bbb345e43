"""loss_type="diffusion" on the Full-Transformer / cross-attention / MDMA models (flow_matching_module.py:452-458 builds
DiffusionLoss for any `model`): no fused loss kernel there -- the field is a differentiable function of the parameters built from the
paths' loss entry points (particle_fm_amd/fm_field.py), criterion / rates / masks are element-wise device ops.  Against the
reference's recorded vectors (tests/golden/{tf,ca,mdma}_diffusion.npz, oracle/make_golden.py --only diffusion_rows): DiffusionLoss +
sub-sampled parameter gradients for both criteria, the probability-flow right-hand side, midpoint on it, DDIM, Euler-Maruyama.
The "_gauss" paths: the same with t_emb="gaussian" (tests/golden/{tf,ca,mdma,epic,epicw}_diffusion_gauss.npz) -- there the EPiC models
(jet-resident "epic", row-matrix "epicw") take this route too: their fused diffusion kernels embed the time themselves."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


ALL = ["tf", "ca", "mdma", "tf_gauss", "ca_gauss", "mdma_gauss", "epic_gauss", "epicw_gauss"]


def _load(path):
    from tests.conftest import load_ca_golden, load_epic_seeded_golden, load_mdma_golden, load_tf_golden, load_wide_golden
    base, _, gauss = path.partition("_")
    loader = {"tf": load_tf_golden, "ca": load_ca_golden, "mdma": load_mdma_golden, "epic": load_epic_seeded_golden, "epicw": load_wide_golden}[base]
    return loader("diffusion_gauss" if gauss else "diffusion")


def _module(g):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, criterion="huber", **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full, strict=False)
    return m.cuda()


def _cond(g, tag):
    c = g.get(tag + "cond")
    return None if c is None or c.numel() == 0 else c


@pytest.mark.parametrize("path", ALL)
@pytest.mark.parametrize("crit", ["huber", "mse"])
def test_loss_and_parameter_gradients(path, crit):
    g = _load(path)
    m = _module(g)
    if not path.endswith("_gauss"):  # (a learned embedding reads no frequency table)
        m.set_freq_table(g.freqs)  # the table of the recording machine (the embedding is bit-sensitive to it: DESIGN 2)
    tag = f"loss_{crit}/"
    x, t, z, mask = (g.get(tag + k).cuda() for k in ("x", "t", "z", "mask"))
    cond = _cond(g, tag)
    loss = m.flows[0].diffusion_loss(x, t.cuda(), z, mask=mask, cond=None if cond is None else cond.cuda(), criterion=crit,
                                     diff_config=g.hp["diff_config"])
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=3e-5, atol=1e-6)
    loss.backward()
    named = dict(m.flows[0].named_parameters())
    bad = []
    for k, want in g.grads(tag).items():
        got = g.pick(named[k[len("flows.0."):]].grad.cpu())
        if float(want.abs().max()) < 2e-6:
            # (the k_linear bias shifts every score of a softmax row alike: its gradient is 0 in exact arithmetic, rounding noise in
            # both implementations
            # -- as in tests/test_hip_ca.py; the sub-sampled picks of a few MDMA tensors are that small too)
            assert float(got.abs().max()) < 1e-5, k
            continue
        l2 = float((got - want).norm()) / max(float(want.norm()), 1e-12)
        if not l2 < 2e-3:
            bad.append((k, l2))
    assert not bad, bad[:8]


@pytest.mark.parametrize("path", ALL)
def test_samplers_match_reference_vectors(path):
    g = _load(path)
    m = _module(g)
    if not path.endswith("_gauss"):  # (a learned embedding reads no frequency table)
        m.set_freq_table(g.freqs)  # the table of the recording machine (the embedding is bit-sensitive to it: DESIGN 2)
    n = int(g.z["n_steps"])
    dev = lambda a: None if a is None else a.cuda()
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask = g.get(tag + "z"), g.get(tag + "mask")
        cond = _cond(g, tag)
        xe = m((z * mask).cuda(), cond=dev(cond), mask=mask.cuda(), reverse=True, ode_solver="midpoint", ode_steps=steps).cpu()
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=3e-4)
    z, mask = g.get("ddim/z"), g.get("ddim/mask")
    cond = _cond(g, "ddim/")
    out = m((z * mask).cuda(), cond=dev(cond), mask=mask.cuda(), reverse=True, ode_solver="ddim", ode_steps=n).cpu()
    torch.testing.assert_close(out, g.get("ddim/x_end"), rtol=1e-3, atol=3e-4)
    # Euler-Maruyama: the reference's draws (recorded) are CPU draws; the module draws on the device -- replay the module's own
    # draws through the oracle instead
    from oracle import diffusion_ref as dr
    vf = _oracle(path, g)
    zc = (z * mask).cuda()
    torch.manual_seed(31)
    out = m(zc, cond=dev(cond), mask=mask.cuda(), reverse=True, ode_solver="em", ode_steps=n).cpu()
    torch.manual_seed(31)
    noises = [torch.randn_like(zc).cpu() for _ in range(n)]
    torch.testing.assert_close(out, dr.em_sample(vf, z * mask, cond, mask, n, g.hp["diff_config"], noises), rtol=1e-3, atol=3e-4)
    # ... and the oracle itself reproduces the reference's recorded Euler-Maruyama run from the recorded draws
    ref = dr.em_sample(vf, g.get("em/z") * g.get("em/mask"), _cond(g, "em/"), g.get("em/mask"), n, g.hp["diff_config"],
                       list(g.get("em/noise")))
    torch.testing.assert_close(ref, g.get("em/x_end"), rtol=1e-3, atol=2e-4)


def _oracle(path, g):
    base = path.partition("_")[0]
    if base == "tf":
        from oracle.tf_ref import TransformerVectorField
        return TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    if base == "ca":
        from oracle.ca_ref import CrossAttentionVectorField
        return CrossAttentionVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    if base in ("epic", "epicw"):
        from oracle.fm_ref import EpicVectorField
        return EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    from oracle.mdma_ref import MdmaVectorField
    return MdmaVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)


@pytest.mark.parametrize("path", ["tf", "mdma"])
def test_training_step_replays_reference_draws(path):
    from oracle import diffusion_ref as dr
    g = _load(path)
    m = _module(g)
    if not path.endswith("_gauss"):  # (a learned embedding reads no frequency table)
        m.set_freq_table(g.freqs)  # the table of the recording machine (the embedding is bit-sensitive to it: DESIGN 2)
    tag = "loss_huber/"
    x, mask = g.get(tag + "x").cuda(), g.get(tag + "mask").cuda()
    cond = _cond(g, tag)
    c = torch.zeros(x.shape[0], device="cuda") if cond is None else cond.cuda()
    torch.manual_seed(77)
    loss = m.training_step((x, mask, c), 0)["loss"]
    torch.manual_seed(77)
    t = torch.rand_like(torch.ones(x.shape[0]))
    zz = (torch.randn_like(x) * mask).cpu()
    vf = _oracle(path, g)
    field = vf if path != "mdma" else (lambda tt, xx, mask=None, cond=None: vf(tt, xx, cond, mask))
    ref_loss, *_ = dr.diffusion_loss(field, x.cpu(), mask.cpu(), cond, t, zz, "huber", g.hp["diff_config"])
    torch.testing.assert_close(loss.detach().cpu(), ref_loss, rtol=3e-5, atol=1e-6)
    loss.backward()
    used = [p for n, p in m.flows[0].net.named_parameters() if "cond_cls" not in n]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in used)
