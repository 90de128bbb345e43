"""The cross-attention oracle (oracle/ca_ref.py) against vectors recorded from the reference's own modules."""
import torch

from oracle.ca_ref import CrossAttentionVectorField
from oracle.fm_ref import cfm_loss, droid_loss, fm_ot_loss, sample_midpoint
from oracle.seeded import subsample

TOL = dict(rtol=2e-4, atol=2e-5)


def _vf(g, state=None):
    return CrossAttentionVectorField(state or g.state, "flows.0.", g.hp, freqs=g.freqs)


def test_nfe(ca_golden):
    g = ca_golden
    vf = _vf(g)
    for mk in ("f32", "int64", "ones"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1)
        with torch.no_grad():
            torch.testing.assert_close(vf(tt, x, cond=cond, mask=mask), g.get(tag + "v_vec_t"), **TOL)
            torch.testing.assert_close(vf(t[0], x, cond=cond, mask=mask), g.get(tag + "v_scalar_t"), **TOL)


def test_losses_and_grads(ca_golden):
    g = ca_golden
    for tag, fn in (("loss_f32/", "fm"), ("cfm/", "cfm"), ("droid/", "droid")):
        state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        if fn == "fm":
            loss, *_ = fm_ot_loss(_vf(g, state), x, mask, cond, t, g.get(tag + "z"), 1e-4)
        elif fn == "cfm":
            loss, *_ = cfm_loss(_vf(g, state), x, mask, cond, t, g.get(tag + "x0"), g.get(tag + "eps"), 1e-4)
        else:
            loss, *_ = droid_loss(_vf(g, state), x, mask, cond, t, g.get(tag + "z"))
        torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
        loss.backward()
        for k, want in g.grads(tag).items():
            got = torch.from_numpy(subsample(state[k].grad.numpy()))
            assert float((got - want).norm()) <= 2e-3 * float(want.norm()) + 2e-6, (tag, k)  # d k_linear.bias == 0 analytically: noise


def test_midpoint(ca_golden):
    g = ca_golden
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        torch.testing.assert_close(sample_midpoint(_vf(g), z, cond, mask, steps), g.get(tag + "x_end"), rtol=1e-3, atol=1e-4)
