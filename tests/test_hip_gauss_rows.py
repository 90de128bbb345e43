"""t_emb="gaussian" (flow_matching_module.py:178-181, 213-221: a small trainable network of the CNF in front of the field) on the
Full-Transformer, cross-attention and row-matrix EPiC ("ew": hidden 300) models: the kernels take the embedding rows through their
`t` argument (PFM_*_F_TEMB_GIVEN), pfm_*_backward_dtemb returns d loss / d temb and autograd continues into embed.1.* / linear.*.
Against vectors recorded from the reference (tests/golden/{tf,ca,epicw}_gauss.npz): forward (vector and scalar t), FM-OT / CFM loss
+ sub-sampled gradients of every tensor including the embedding network, midpoint 3 / 10."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(path):
    from tests.conftest import load_ca_golden, load_tf_golden, load_wide_golden
    return {"tf": load_tf_golden, "ca": load_ca_golden, "ew": load_wide_golden}[path]("gauss")


def _module(g):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    return m.cuda()


@pytest.mark.parametrize("path", ["tf", "ca", "ew"])
def test_forward_matches_reference_vectors(path):
    g = _load(path)
    m = _module(g)
    for mk in ("f32", "int64", "none" if path == "ew" else "ones"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        N = x.shape[1]
        tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
        dmask = None if mask is None else mask.cuda()
        with torch.no_grad():
            v = m.flows[0](tt.cuda(), x.cuda(), cond=cond.cuda(), mask=dmask).cpu()
            vs = m.flows[0](t[0].cuda(), x.cuda(), cond=cond.cuda(), mask=dmask).cpu()
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)
        torch.testing.assert_close(vs, g.get(tag + "v_scalar_t"), atol=2e-5, rtol=2e-4)


@pytest.mark.parametrize("path", ["tf", "ca", "ew"])
@pytest.mark.parametrize("kind", ["FM-OT", "CFM"])
def test_loss_and_gradients_including_the_embedding_network(path, kind):
    g = _load(path)
    if path == "ew" and kind == "CFM":
        pytest.skip("the epicw recorder holds FM-OT and droid losses")
    m = _module(g)
    tag = "loss_f32/" if kind == "FM-OT" else "cfm/"
    x, t, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond"))
    if kind == "FM-OT":
        a, eps = g.get(tag + "z").cuda(), None
    else:
        a, eps = g.get(tag + "x0").cuda(), g.get(tag + "eps").cuda()
    loss = m.flows[0].fm_loss(x, t, a, mask=mask, cond=cond, sigma=1e-4, kind=kind, eps=eps)
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=3e-5, atol=1e-6)
    loss.backward()
    named = dict(m.flows[0].named_parameters())
    ref = g.grads(tag)
    assert all(("flows.0." + k) in ref for k in ("embed.1.weight", "embed.1.bias", "linear.weight", "linear.bias"))
    bad = []
    for k, want in ref.items():
        got = g.pick(named[k[len("flows.0."):]].grad.cpu())
        if float(want.abs().max()) < 2e-6:  # (a k_linear bias: zero in exact arithmetic)
            assert float(got.abs().max()) < 1e-5, k
            continue
        l2 = float((got - want).norm()) / max(float(want.norm()), 1e-12)
        if not l2 < 2e-3:
            bad.append((k, l2))
    assert not bad, bad[:8]
    assert named["embed.0.W"].grad is None  # GaussianFourierProjection.W is frozen (time_emb.py:15)


@pytest.mark.parametrize("path", ["tf", "ca", "ew"])
def test_samplers_and_training_step(path):
    g = _load(path)
    m = _module(g)
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        out = m((z * mask).cuda(), cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="midpoint", ode_steps=steps).cpu()
        keep = mask.squeeze(-1) != 0
        torch.testing.assert_close(out[keep], g.get(tag + "x_end")[keep], atol=2e-4, rtol=1e-3)
    # euler / rk4 through the same table mechanism: against the oracle's restated integrators
    from oracle.fm_ref import sample_fixed_step
    if path == "tf":
        from oracle.tf_ref import TransformerVectorField as VF
    elif path == "ca":
        from oracle.ca_ref import CrossAttentionVectorField as VF
    if path == "ew":
        from oracle.fm_ref import EpicVectorField
        vf = EpicVectorField(g.state, "flows.0.net", g.hp)
    else:
        vf = VF(g.state, "flows.0.", g.hp, freqs=g.freqs)
    z, mask, cond = (g.get("midpoint_10/" + k) for k in ("z", "mask", "cond"))
    for solver in ("euler", "rk4"):
        out = m((z * mask).cuda(), cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver=solver, ode_steps=6).cpu()
        ref = sample_fixed_step(vf, z, cond, mask, ode_steps=6, solver=solver)
        keep = mask.squeeze(-1) != 0
        torch.testing.assert_close(out[keep], ref[keep], atol=2e-4, rtol=1e-3)
    x, mask, cond = (g.get("loss_f32/" + k).cuda() for k in ("x", "mask", "cond"))
    loss = m.training_step((x, mask, cond), 0)["loss"]
    assert torch.isfinite(loss)
    loss.backward()
    for k in ("embed.1.weight", "linear.bias"):
        assert dict(m.flows[0].named_parameters())[k].grad.abs().max() > 0


def test_droid_loss_and_the_encoder_forward_with_a_given_embedding_on_the_row_matrix_path():
    """(a) DroidLoss (losses.py:304-342) with the gaussian embedding on the row-matrix EPiC path, against the recorded loss and
    gradients; (b) EPiC_encoder.forward(t_emb, x, cond, mask) (epic.py:304: the reference's own signature takes the EMBEDDING) on the
    row-matrix path of a cosine configuration equals vector_field(t, ...) with the in-kernel embedding."""
    g = _load("ew")
    m = _module(g)
    tag = "droid/"
    x, t, z, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "z", "mask", "cond"))
    loss = m.flows[0].fm_loss(x, t, z, mask=mask, cond=cond, sigma=1e-4, kind="droid")
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=3e-5, atol=1e-6)
    loss.backward()
    named = dict(m.flows[0].named_parameters())
    for k, want in g.grads(tag).items():
        got = g.pick(named[k[len("flows.0."):]].grad.cpu())
        assert float((got - want).norm()) / max(float(want.norm()), 1e-12) < 2e-3, k
    from tests.conftest import load_wide_golden
    from tests.test_hip_wide_modules import _module as wide_module
    gc = load_wide_golden("small")
    mc = wide_module(gc)
    net = mc.flows[0].net
    x, t, mask, cond = (gc.get("nfe_f32/" + k).cuda() for k in ("x", "t", "mask", "cond"))
    with torch.no_grad():
        want = net.vector_field(t, x, cond, mask)
        temb = mc.flows[0].time_embedding(t[:, None].expand(-1, x.shape[1]), x, "cosine")  # (B, N, T), as CNF.forward hands it over
        got = net(temb, x, cond, mask)
    torch.testing.assert_close(got, want, atol=2e-6, rtol=1e-5)


def test_mdma_with_the_gaussian_embedding_and_every_time_concatenation():
    """tests/golden/mdma_gauss.npz: t_emb="gaussian" in front of an MDMA with t_local_cat = t_global_cat = True and add_time_to_input --
    the embedding reaches embed twice, every Block.fc0 and the three class-token Linears of every block; pfm_mdma_backward_dtemb
    collects d loss / d temb from all of them.  Forward, FM-OT / CFM loss + every gradient (embedding network included), midpoint vs the
    reference's vectors; euler / rk4 vs the oracle."""
    from oracle.fm_ref import sample_fixed_step
    from oracle.mdma_ref import MdmaVectorField, broadcast_field
    from tests.conftest import load_mdma_golden
    g = load_mdma_golden("gauss")
    m = _module(g)
    cnf = m.flows[0]
    for mk in ("f32", "int64", "ones"):
        tag = f"nfe_{mk}/"
        x, t, mask = (g.get(tag + k) for k in ("x", "t", "mask"))
        tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1)
        with torch.no_grad():
            v = cnf(tt.cuda(), x.cuda(), mask=mask.cuda()).cpu()
            vs = cnf(t[0].cuda(), x.cuda(), mask=mask.cuda()).cpu()
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)
        torch.testing.assert_close(vs, g.get(tag + "v_scalar_t"), atol=2e-5, rtol=2e-4)
    for kind, tag in (("FM-OT", "loss_f32/"), ("CFM", "cfm/")):
        x, t, mask = (g.get(tag + k).cuda() for k in ("x", "t", "mask"))
        a, eps = (g.get(tag + "z").cuda(), None) if kind == "FM-OT" else (g.get(tag + "x0").cuda(), g.get(tag + "eps").cuda())
        m.zero_grad()
        loss = cnf.fm_loss(x, t, a, mask=mask, sigma=1e-4, kind=kind, eps=eps)
        torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=3e-5, atol=1e-6)
        loss.backward()
        named = dict(cnf.named_parameters())
        ref = g.grads(tag)
        assert all(("flows.0." + k) in ref for k in ("embed.1.weight", "embed.1.bias", "linear.weight", "linear.bias"))
        bad = []
        for k, want in ref.items():
            got = named[k[len("flows.0."):]].grad
            if got is None:
                assert "cond_cls" in k, k  # (constructed and never used, mdma.py:37)
                continue
            got = g.pick(got.cpu())
            if float(want.abs().max()) < 2e-6:
                assert float(got.abs().max()) < 1e-5, k
                continue
            l2 = float((got - want).norm()) / max(float(want.norm()), 1e-12)
            if not l2 < 2e-3:
                bad.append((k, l2))
        assert not bad, bad[:8]
    vf = broadcast_field(MdmaVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs))
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask = (g.get(tag + k) for k in ("z", "mask"))
        out = m((z * mask).cuda(), mask=mask.cuda(), reverse=True, ode_solver="midpoint", ode_steps=steps).cpu()
        torch.testing.assert_close(out, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)
    z, mask = (g.get("midpoint_10/" + k) for k in ("z", "mask"))
    for solver in ("euler", "rk4"):
        out = m((z * mask).cuda(), mask=mask.cuda(), reverse=True, ode_solver=solver, ode_steps=6).cpu()
        torch.testing.assert_close(out, sample_fixed_step(vf, z, None, mask, ode_steps=6, solver=solver), atol=2e-4, rtol=1e-3)
    x, mask = (g.get("loss_f32/" + k).cuda() for k in ("x", "mask"))
    loss = m.training_step((x, mask, None), 0)["loss"]
    assert torch.isfinite(loss)
