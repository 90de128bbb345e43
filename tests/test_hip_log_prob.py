"""CNF.log_prob (flow_matching_module.py:330-347) on the HIP path: the right-hand side of its augmented ODE -- the field and, per particle,
the sum over the features of the batched vector-Jacobian products -- against torch autograd through the ORACLE's field with the
reference's own formula (:331-343), and the integral against a fixed-step rk4 solution of the oracle's augmented system.  The adaptive
integrator itself is PARITY UNPINNED (zuko is not in the image; tests/test_hip_adaptive.py)."""
import pytest
import torch


pytestmark = pytest.mark.gpu


def _augmented(vf):
    """the reference's ``augmented`` (flow_matching_module.py:334-343), on the oracle's field (no cond, no mask: self(t, x))"""

    def f(t, x):
        i = torch.eye(x.shape[-1]).to(x)
        i = i.expand(x.shape + x.shape[-1:]).movedim(-1, 0)  # :331-332
        with torch.enable_grad():
            x = x.detach().requires_grad_()
            dx = vf(t, x)
            jac = torch.autograd.grad(dx, x, i, is_grads_batched=True)[0]
        return dx.detach(), torch.einsum("i...i", jac)

    return f


def _plain(loader, name, field):
    def make():
        from tests.test_hip_adaptive import _module
        g = loader(name)  # no conditioning, t_emb="sincos": oracle/make_golden.py "plain"
        return g, _module(g), field(g)
    return make


def _families():
    from oracle.fm_ref import EpicVectorField
    from tests.conftest import load_wide_golden
    epic = lambda g: EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    return {"epic": _plain(load_wide_golden, "plain", epic), "epicw": _plain(load_wide_golden, "plainw", epic)}


@pytest.mark.parametrize("family", ["epic", "epicw"])
def test_field_and_trace_match_autograd_through_the_oracle(family):
    g, m, vf = _families()[family]()
    assert m.flows[0].net.is_wide(g.hp["num_particles"]) == (family == "epicw")
    x = g.get("midpoint_10/z")
    torch.manual_seed(5)
    for t in (0.0, 0.37, 1.0):
        want_dx, want_tr = _augmented(vf)(torch.tensor(t), x)
        dx, tr = m.flows[0].field_and_trace(torch.tensor(t).cuda(), x.cuda())
        torch.testing.assert_close(dx.cpu(), want_dx, atol=2e-4, rtol=2e-4)
        torch.testing.assert_close(tr.cpu(), want_tr, atol=5e-4, rtol=5e-4)


@pytest.mark.parametrize("family", ["epic", "epicw"])
def test_log_prob_reaches_the_fine_solution(family):
    from oracle.fm_ref import rk_trajectory_end
    g, m, vf = _families()[family]()
    x = g.get("midpoint_10/z")[:2]
    F = x.shape[-1]
    aug = _augmented(vf)

    def rhs(t, s):
        dx, tr = aug(t, s[..., :F])
        return torch.cat([dx, (tr * 1e-2).unsqueeze(-1)], dim=-1)

    with torch.no_grad():
        s1 = rk_trajectory_end(rhs, torch.cat([x, torch.zeros_like(x[..., :1])], dim=-1), torch.linspace(0.0, 1.0, 40), "rk4")
    z, ladj = s1[..., :F], s1[..., F]
    want = torch.distributions.Normal(0.0, 1.0).log_prob(z).sum(dim=-1) + ladj * 1e2  # :347
    got = m.flows[0].log_prob(x.cuda()).cpu()
    assert got.shape == x.shape[:2]
    torch.testing.assert_close(got, want, atol=2e-2, rtol=1e-2)


def test_log_prob_refusals():
    """the models whose log_prob does not run in the reference either: MDMA's one-output field under the batched vector-Jacobian product
    (:339), the transformer encoders with mask=None (droid_transformer.py:539)"""
    from tests.conftest import load_ca_golden, load_mdma_golden, load_tf_golden
    from tests.test_hip_adaptive import _module
    for g, msg in ((load_mdma_golden("tglob"), "one output per particle"), (load_tf_golden("plain"), "mask=None"), (load_ca_golden("plain"), "mask=None")):
        with pytest.raises(NotImplementedError, match=msg):
            _module(g).flows[0].log_prob(torch.zeros(2, g.hp["num_particles"], 3).cuda())
