"""The transformer oracle (oracle/tf_ref.py) against vectors recorded from the reference's own modules."""
import torch

from oracle.fm_ref import cfm_loss, fm_ot_loss, sample_midpoint
from oracle.tf_ref import TransformerVectorField

TOL = dict(rtol=2e-4, atol=2e-5)


def _vf(g, state=None):
    return TransformerVectorField(state or g.state, "flows.0.", g.hp, freqs=g.freqs)


def test_nfe(tf_golden):
    g = tf_golden
    vf = _vf(g)
    for mk in ("f32", "int64", "ones"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1)
        with torch.no_grad():
            torch.testing.assert_close(vf(tt, x, cond=cond, mask=mask), g.get(tag + "v_vec_t"), **TOL)
            torch.testing.assert_close(vf(t[0], x, cond=cond, mask=mask), g.get(tag + "v_scalar_t"), **TOL)


def _grad_check(g, state, tag):
    ref = g.grads(tag)
    assert len(ref) == len(state)
    for k, p in state.items():
        torch.testing.assert_close(g.pick(p.grad), ref[k], rtol=2e-3, atol=2e-5, msg=lambda m, k=k: f"{k}: {m}")


def test_fm_loss_and_grads(tf_golden):
    g = tf_golden
    state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    loss, *_ = fm_ot_loss(_vf(g, state), x, mask, cond, t, z, 1e-4)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grad_check(g, state, tag)


def test_cfm_loss_and_grads(tf_golden):
    g = tf_golden
    state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
    tag = "cfm/"
    x, t, x0, eps, mask, cond = (g.get(tag + k) for k in ("x", "t", "x0", "eps", "mask", "cond"))
    loss, *_ = cfm_loss(_vf(g, state), x, mask, cond, t, x0, eps, 1e-4)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grad_check(g, state, tag)


def test_midpoint(tf_golden):
    g = tf_golden
    vf = _vf(g)
    for steps in (3, 10, 100):
        tag = f"midpoint_{steps}/"
        if g.get(tag + "z") is None:
            continue
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        xe = sample_midpoint(vf, z, cond, mask, steps)
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=1e-4)
